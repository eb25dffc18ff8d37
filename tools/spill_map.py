#!/usr/bin/env python3
"""Where a kernel's spilled scalars are moved: v_readlane_b32 / v_writelane_b32 (SGPR spills live in VGPR lanes) and scratch
loads / stores of one kernel, attributed to the source lines the compiler's line table names (static; no GPU), and to the loops
(label .. backward branch) they sit in.  A lane move counts as a spill when its vector register is touched by lane moves ONLY (the
registers the compiler parks scalars in); the broadcasts the algorithm itself asks for use registers that ordinary instructions touch too.  The lean form of graph_search_kernel was found this way round from counters (half of a
launch's vector instructions were such moves); this tool answers "which lines" before a change goes to the GPU.
    python tools/spill_map.py traverse_sq_insert_lat.hip 'graph_insert_search_kernel<0, 4, false, 1>' [--top 25]"""
import argparse
import importlib.util
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "hnswindex.net_amd" / "csrc"
spec = importlib.util.spec_from_file_location("hnsw_build", ROOT / "hnswindex.net_amd" / "build.py")
build = importlib.util.module_from_spec(spec)
spec.loader.exec_module(build)

LANE = re.compile(r"^\s*v_(readlane|writelane)_b32\s+([vs]\d+),\s*([vs]\d+)")
SCRATCH = re.compile(r"^\s*scratch_(load|store)_\w+")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def spill_vgprs(body):
    """VGPRs that are touched by v_writelane / v_readlane ONLY: the lanes the compiler keeps spilled scalars in (a register the
    algorithm broadcasts from is also written or read by ordinary vector instructions)."""
    lane_regs, other = set(), set()
    for ln in body:
        if not INSTR.match(ln) or ln.strip().startswith("."):
            continue
        code = ln.split(";")[0]
        m = LANE.match(code)
        if m:
            lane_regs.add(int((m.group(2) if m.group(1) == "writelane" else m.group(3))[1:]))
            continue
        for a, lo, hi in VREG.findall(code):
            other.update([int(a)] if a else range(int(lo), int(hi) + 1))
    return lane_regs - other
INSTR = re.compile(r"^\s+([a-z][a-z0-9_]+)\b")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("unit")
    ap.add_argument("kernel", help="demangled name as tools/kernel_resources.py prints it")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        asm = Path(tmp) / "k.s"
        cmd = [build.hipcc(), *build.FLAGS, "-gline-tables-only", f"-I{ROOT / 'include'}", f"-I{CSRC}", "--cuda-device-only", "-S", str(CSRC / a.unit), "-o", str(asm)]
        subprocess.run(cmd, check=True, capture_output=True)
        lines = asm.read_text().split("\n")
    files, body, inside = {}, [], False
    want = a.kernel.replace(" ", "")
    for ln in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = Path(m.group(3) or m.group(2)).name
        m = re.match(r"^(_ZN\w+):", ln)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            inside = re.sub(r"\(.*", "", name).replace("void ", "").replace(" ", "").endswith(want)
            continue
        if inside:
            body.append(ln)
            if "s_endpgm" in ln:
                inside = False
    if not body:
        sys.exit(f"kernel {a.kernel!r} not found in {a.unit}")
    parked = spill_vgprs(body)
    loc, per_line, per_line_all, n_instr, n_spill = None, Counter(), Counter(), 0, 0
    labels, loops, idx = {}, [], 0
    spill_at, loc_at = [], []
    for ln in body:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = idx
            continue
        m = INSTR.match(ln)
        if not m or ln.strip().startswith("."):
            continue
        idx += 1
        n_instr += 1
        per_line_all[loc] += 1
        loc_at.append(loc)
        lm = LANE.match(ln)
        if SCRATCH.match(ln) or (lm and int((lm.group(2) if lm.group(1) == "writelane" else lm.group(3))[1:]) in parked):
            n_spill += 1
            per_line[loc] += 1
            spill_at.append(idx)
        b = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if b and b.group(1) in labels:       # backward branch: a loop from the label to here
            loops.append((labels[b.group(1)], idx, b.group(1)))
    print(f"{a.kernel}: {n_instr} instructions, {n_spill} spill moves ({100.0 * n_spill / n_instr:.1f} %); scalars parked in v{sorted(parked)}")
    print(f"\nsource lines by spill moves (of {len(per_line)} lines that carry any):")
    for (f, l), c in per_line.most_common(a.top):
        print(f"  {c:5d}  {f}:{l}   ({per_line_all[(f, l)]} instructions on that line)")
    print("\nloops (label .. backward branch) by spill moves inside; nested loops count in their parents too:")
    rows = []
    for lo, hi, lab in loops:
        c = sum(1 for s in spill_at if lo < s <= hi)
        rows.append((c, hi - lo, lab, lo, hi))
    def show(rs):
        seen = set()
        for c, n, lab, lo, hi in rs:
            if (lab, n // 16) in seen:      # the same loop reached by several back edges
                continue
            seen.add((lab, n // 16))
            inside = Counter(loc_at[lo:hi]).most_common(3)
            print(f"  {c:5d} spill moves in {n:6d} instructions ({100.0 * c / max(1, n):4.1f} %)  {lab}   mostly " + ", ".join(f"{f}:{l} ({k})" for (f, l), k in inside))
    show(sorted(rows, reverse=True)[:a.top])
    print("\nthe same for loops of at most 1 500 instructions (the inner loops a wave spends its time in):")
    show(sorted((r for r in rows if r[1] <= 1500 and r[0] > 0), reverse=True)[:a.top])


if __name__ == "__main__":
    main()
