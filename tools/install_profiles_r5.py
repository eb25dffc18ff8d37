#!/usr/bin/env python3
"""Copies the summaries produced by tools/run_profiles_r5.sh (gpurun_out/profiles5/) into profiles/ under the r5 prefix and
derives
  r5_gather_ceilings.json   what random row gathers reach on this chip per row size (tools/gather_bench): from a table far beyond the
                            Infinity Cache ("rows", with the FETCH_SIZE calibration factor: known bytes / counter bytes) AND from a table
                            of each configuration's own size ("by_table": what bench.py prices frac_of_measured_gather against)
  r5_pmc_traffic.json       per configuration: PMC traffic per launch of graph_search_kernel (and over the whole build for the
                            two Add kernels) against the algorithmic bytes, stamped with the build id of the library that was measured --
                            bench.py quotes roofline.traffic only when that id is the running library's
  r5_issue_counters.json    instruction-issue counters of the product search kernels (C5-size at 12 500 and 65 536 per call, C4-size, C2)
  r5_mfma_utilisation.json  C3 insert kernel: MFMA flop/s against the fp32 matrix peak
python tools/install_profiles_r5.py"""
import json
import re
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src, dst, r = ROOT / "gpurun_out" / "profiles5", ROOT / "profiles", "r5"
FP32_MATRIX_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD


def json_line(path):
    return "".join(l for l in open(path) if l.startswith('{"metric"'))


def bench(path):
    t = json_line(path)
    return json.loads(t) if t else None


for f, name in (("bench_plain", "bench_1m_plain"), ("bench_under_rocprof", "bench_under_rocprof"), ("bench_c3", "bench_c3_1m_768_ucosine"),
                ("bench_c4_size", "bench_c4_10m_1gpu"), ("bench_c5_size", "bench_c5_10m_96_int8_1gpu"), ("bench_clustered", "bench_1m_clustered"),
                ("bench_host_cabi", "bench_host_cabi_200k"), ("bench_2rank_gloo", "bench_2rank_gloo_rehearsal_200k"),
                ("bench_native2", "bench_native_2contexts_rehearsal_200k")):
    if (src / f"{f}.log").exists() and json_line(src / f"{f}.log"):
        (dst / f"{r}_{name}.json").write_text(json_line(src / f"{f}.log"))
for f, name in (("kernel_stats.json", "bench_1m_kernel_stats.json"), ("s_kernel_stats.csv", "bench_1m_kernel_stats.csv"),
                ("kernel_stats_host_cabi.json", "host_cabi_kernel_stats.json")):
    if (src / f).exists():
        shutil.copy(src / f, dst / f"{r}_{name}")

# ---- gather ceilings + FETCH_SIZE calibration ----
ceil = {"tool": "tools/gather_bench.hip: 16.8M random rows of a 4-GiB table (16x the Infinity Cache), the product's lane mapping "
                "(8 lanes per row, strided dwords, up to 64 loads in flight per lane)", "rows": {}}
for rb in (128, 512, 3072):
    log = src / f"gather_{rb}.log"
    if not log.exists():
        continue
    best = max(float(m.group(1)) for m in re.finditer(r"([0-9.]+) GB/s", log.read_text()))
    e = {"row_bytes": rb, "best_GBps": best, "rows_per_s": best * 1e9 / rb}
    pm = src / f"gather_pmc_{rb}.json"
    if pm.exists():
        c = list(json.load(open(pm))["counters"].values())[0]["FETCH_SIZE"]["avg_per_dispatch"]
        known = 16777216 * rb
        e.update({"FETCH_SIZE_KB_per_launch": c, "known_bytes_per_launch": known, "calibration_factor": known / (c * 1024)})
    ceil["rows"][str(rb)] = e
v1 = src / "gather_128_v1.log"
if v1.exists():
    ceil["rows"]["128"]["one_16_byte_load_per_lane_GBps"] = max(float(m.group(1)) for m in re.finditer(r"([0-9.]+) GB/s", v1.read_text()))
prev_r5 = {}
if (dst / f"{r}_gather_ceilings.json").exists():   # a partial re-run (one configuration's passes) must not drop what an earlier run measured
    prev_r5 = json.load(open(dst / f"{r}_gather_ceilings.json"))
for k, v in prev_r5.get("rows", {}).items():
    ceil["rows"].setdefault(k, v)
ceil["by_table"] = []
for cfg, rb, rows in (("c2", 512, 1_000_000), ("c3", 3072, 1_000_000), ("c4", 512, 10_000_000), ("c5", 128, 10_000_000)):
    log = src / f"gather_own_{cfg}.log"
    if log.exists():
        best = max(float(m.group(1)) for m in re.finditer(r"([0-9.]+) GB/s", log.read_text()))
        ceil["by_table"].append({"config": cfg, "row_bytes": rb, "table_bytes": rows * rb, "best_GBps": best, "rows_per_s": best * 1e9 / rb,
                                 "note": "uniform random rows of a table of this configuration's own size (a table near the 256-MiB Infinity Cache is served partly from it)"})
have = {e["config"] for e in ceil["by_table"]}
ceil["by_table"] += [e for e in prev_r5.get("by_table", []) if e["config"] not in have]
if not ceil["rows"]:  # the >> cache pass was not re-run this round: keep last round's figures
    prev = dst / "r4_gather_ceilings.json"
    if prev.exists():
        ceil["rows"] = json.load(open(prev))["rows"]
        ceil["rows_measured_in"] = "round 4 (profiles/r4_gather_ceilings.json)"
json.dump(ceil, open(dst / f"{r}_gather_ceilings.json", "w"), indent=1)


def factor(rb):
    return ceil["rows"].get(str(rb), {}).get("calibration_factor", 2.0)


def counter(path, cname, kernel):
    cs = json.load(open(path))["counters"]
    keys = [k for k in cs if kernel in k]
    if not keys:
        return None
    tot = {"sum": 0.0, "dispatches": 0}
    for k in keys:
        tot["sum"] += cs[k][cname]["sum"]
        tot["dispatches"] += cs[k][cname]["dispatches"]
    return tot


traffic = {"round": 5, "build_id": None, "note": "FETCH_SIZE (KB) x 1024 x calibration factor of that row size + WRITE_SIZE (KB) x 1024, per launch; the counters tally "
                                "memory-side requests of the L2s: Infinity-Cache hits are counted (guide, HBM section). Every configuration carries the build id "
                                "of the library its passes ran on (bench.py quotes a figure only for the running library's own id); the top-level build_id is "
                                "the id of the passes installed last", "configs": {}}
prev_traffic = {}
if (dst / f"{r}_pmc_traffic.json").exists():   # a partial re-run keeps the other configurations' entries, each under the build id it was measured on
    pt = json.load(open(dst / f"{r}_pmc_traffic.json"))
    prev_traffic = {k: {"build_id": pt.get("build_id"), **v} for k, v in pt.get("configs", {}).items()}
for cfg, rb in (("c2", 512), ("c3", 3072), ("c4", 512), ("c5", 128), ("c5L", 128)):
    fj, wj, bl = src / f"{cfg}_fetch.json", src / f"{cfg}_write.json", src / f"bench_{cfg}_fetch.log"
    if not (fj.exists() and wj.exists() and bl.exists()):
        continue
    b, bw = bench(bl), bench(src / f"bench_{cfg}_write.log") if (src / f"bench_{cfg}_write.log").exists() else None
    if not b:
        continue
    if bw and bw.get("build_id") != b.get("build_id"):
        print(f"{cfg}: FETCH_SIZE pass on build {str(b.get('build_id'))[:16]}, WRITE_SIZE pass on {str(bw.get('build_id'))[:16]}: skipped (re-run both)")
        continue
    traffic["build_id"] = b.get("build_id")
    f = factor(rb)
    e = {"build_id": b.get("build_id"), "workload": {k: b["config"][k] for k in ("n", "dim", "queries_per_gpu_per_step", "ef_search", "k", "max_edges")}, "row_bytes_fetched": rb,
         "calibration_factor": f}
    fs, ws = counter(fj, "FETCH_SIZE", "graph_search_kernel"), counter(wj, "WRITE_SIZE", "graph_search_kernel")
    alg = b["roofline"]["evals_per_launch"] * b["roofline"]["bytes_per_eval"]   # (no CPU leg under the profiler: rows measured by the device)
    t = fs["sum"] / fs["dispatches"] * 1024 * f + ws["sum"] / ws["dispatches"] * 1024
    e["graph_search_kernel"] = {"FETCH_SIZE_KB_per_launch": fs["sum"] / fs["dispatches"], "WRITE_SIZE_KB_per_launch": ws["sum"] / ws["dispatches"],
                                "traffic_bytes_per_launch": t, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": t / alg}
    ra = b.get("roofline_add")
    if ra:
        for kern, part in (("graph_insert_search_kernel", "insert_search"), ("graph_link_kernel", "link_half")):
            fa, wa = counter(fj, "FETCH_SIZE", kern), counter(wj, "WRITE_SIZE", kern)
            if not fa:
                continue
            a = ra[part]["evals"] * ra["bytes_per_eval"]
            ta = fa["sum"] * 1024 * f + wa["sum"] * 1024
            e[part] = {"FETCH_SIZE_KB_total": fa["sum"], "WRITE_SIZE_KB_total": wa["sum"], "traffic_bytes_total": ta, "algorithmic_bytes_total": a,
                       "traffic_over_algorithmic": ta / a}
    traffic["configs"][cfg] = e
for k, v in prev_traffic.items():
    traffic["configs"].setdefault(k, v)
if traffic["build_id"] is None and prev_traffic:
    traffic["build_id"] = pt.get("build_id")
json.dump(traffic, open(dst / f"{r}_pmc_traffic.json", "w"), indent=1)

# ---- instruction issue of the product search kernels ----
issue = {"round": 5, "note": "rocprofv3 --pmc, one pass per counter group, bench.py --steps 2 (every dispatch of the search kernel in the run averaged: the timed steps, "
                             "the resident and recall calls -- all of the same size); SQ_* cycle counters are in quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs. "
                             "The launches run with shadow traversals (idle waves of a draining launch start exact traversals of the jobs still running): their instructions are in these counts",
         "kernels": {}}
for cfg, what in (("c5", "C5-size: 10M x 96 int8 records, 12 500 queries per call"), ("c5L", "C5-size, 65 536 queries per call"),
                  ("c4", "C4-size: 10M x 128 f32, 12 500 queries per call"), ("c2", "C2: 1M x 128 f32, 65 536 queries per call")):
    ja, jb = src / f"{cfg}_issue_a.json", src / f"{cfg}_issue_b.json"
    if not (ja.exists() and jb.exists()):
        continue
    ca, cb = json.load(open(ja))["counters"], json.load(open(jb))["counters"]
    ka = [k for k in ca if "graph_search_kernel" in k]
    if not ka:
        continue
    k = max(ka, key=lambda x: ca[x]["SQ_WAVE_CYCLES"]["sum"])
    A = {c: v["avg_per_dispatch"] for c, v in ca[k].items()}
    B = {c: v["avg_per_dispatch"] for c, v in cb.get(k, {}).items()}
    cyc = A["GRBM_GUI_ACTIVE"] / 8.0
    wave_cycles = A["SQ_WAVE_CYCLES"] * 4.0
    e = {"config": what, "kernel": k, "counters_per_launch": {**A, **B}, "kernel_cycles": cyc,
         "avg_waves_per_simd": wave_cycles / cyc / 1024.0,
         "valu_instructions_per_launch": A["SQ_INSTS_VALU"], "salu_instructions_per_launch": A["SQ_INSTS_SALU"],
         "instructions_per_vmem_read": (A["SQ_INSTS_VALU"] + A["SQ_INSTS_SALU"] + A["SQ_INSTS_LDS"]) / max(1.0, A["SQ_INSTS_VMEM_RD"])}
    if B:
        e.update({"valu_pipe_busy_fraction": B["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0),
                  "scalar_issue_fraction_per_cu": B["SQ_ACTIVE_INST_SCA"] * 4.0 / (cyc * 1024.0),
                  "wave_time_executing": B["SQ_ACTIVE_INST_ANY"] / A["SQ_WAVE_CYCLES"],
                  "wave_time_waiting_for_issue": B["SQ_WAIT_INST_ANY"] / A["SQ_WAVE_CYCLES"],
                  "wave_time_waiting_on_memory_or_dependencies": 1.0 - (B["SQ_ACTIVE_INST_ANY"] + B["SQ_WAIT_INST_ANY"]) / A["SQ_WAVE_CYCLES"]})
    bi = bench(src / f"bench_{cfg}_issue_a.log")
    if bi:
        e["build_id"] = bi.get("build_id")
    issue["kernels"][cfg] = e
if (dst / f"{r}_issue_counters.json").exists():   # a partial re-run keeps the other configurations' entries (each names the build it ran on where known)
    for k, v in json.load(open(dst / f"{r}_issue_counters.json")).get("kernels", {}).items():
        issue["kernels"].setdefault(k, v)
if issue["kernels"]:
    json.dump(issue, open(dst / f"{r}_issue_counters.json", "w"), indent=1)
    print("issue", {k: (round(v["avg_waves_per_simd"], 2), round(v.get("valu_pipe_busy_fraction", 0), 3), round(v.get("wave_time_waiting_on_memory_or_dependencies", 0), 3)) for k, v in issue["kernels"].items()})

# ---- MFMA utilisation of the C3 insert kernel ----
mj, bl = src / "c3_mfma.json", src / "bench_c3_mfma.log"
if mj.exists() and bl.exists() and bench(bl):
    b = bench(bl)
    cs = json.load(open(mj))["counters"]
    ks = [x for x in cs if "graph_insert_search_kernel" in x]  # the loaded form and the latency variant: summed (seconds are both's)
    k = " + ".join(sorted(ks))
    mops, busy, gui = (sum(cs[x][c]["sum"] for x in ks) for c in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
    secs = b["roofline_add"]["insert_search"]["seconds"]
    flops = mops * 512.0                      # the counter's unit: 512 floating-point operations
    out = {"kernel": k, "config": "C3 build (1M x 768 ucosine, M=32, efConstruction=400): gram_tile inside RelativeNeighborPruning, v_mfma_f32_32x32x2_f32",
           "SQ_INSTS_VALU_MFMA_MOPS_F32": mops, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_summed_over_8_XCDs": gui,
           "mfma_instructions": busy / 64.0, "flops": flops, "kernel_seconds_hip_events_same_run": secs,
           "mfma_TFLOPs": flops / secs / 1e12, "fp32_matrix_peak_TFLOPs": FP32_MATRIX_PEAK_TFLOPS,
           "mfma_utilisation_vs_peak": flops / secs / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
           "mfma_pipe_busy_fraction": busy / (gui / 8.0 * 1024.0),
           "note": "a prefilter inside an HBM-bound kernel: the matrix core decides 99.8 % of the heuristic's comparisons from one tile per 32 x 32 pairs, "
                   "the kernel's time is its row traffic (roofline_add)"}
    json.dump(out, open(dst / f"{r}_mfma_utilisation.json", "w"), indent=1)
    print("MFMA", out["mfma_TFLOPs"], out["mfma_utilisation_vs_peak"], out["mfma_pipe_busy_fraction"])
print(json.dumps({k: v.get("graph_search_kernel", {}).get("traffic_over_algorithmic") for k, v in traffic["configs"].items()}))
print(json.dumps({k: (v["best_GBps"], v.get("calibration_factor")) for k, v in ceil["rows"].items()}))
