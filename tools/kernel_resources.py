#!/usr/bin/env python3
"""Static resource usage of every device kernel of the library, from the compiler's own remarks
(hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed): VGPRs, SGPRs, spills, scratch, LDS, occupancy per kernel, stamped with the
build id of the sources -- the baseline a kernel change is compared with before it goes to the GPU (round 5: the lean form of
graph_search_kernel was found by its spilled SGPRs).
    python tools/kernel_resources.py [-j 8] [-o profiles/r5_kernel_resources.json]"""
import argparse
import importlib.util
import json
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "hnswindex.net_amd" / "csrc"

spec = importlib.util.spec_from_file_location("hnsw_build", ROOT / "hnswindex.net_amd" / "build.py")
build = importlib.util.module_from_spec(spec)
spec.loader.exec_module(build)

FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
          "Occupancy [waves/SIMD]": "waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
          "LDS Size [bytes/block]": "lds_bytes_per_block"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void ", "") for o in out[:len(names)]]


def unit(src, tmp):
    cmd = [build.hipcc(), *build.FLAGS, f"-I{ROOT / 'include'}", f"-I{CSRC}", "--cuda-device-only", "-c", str(src), "-o", str(Path(tmp) / (src.stem + ".o")),
           "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"{src.name}: {r.stderr[-800:]}")
    kernels, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (?:\s*)(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"mangled": t.split(":", 1)[1].strip(), "unit": src.name}
            kernels.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.rsplit(":", 1)
            if k.strip() in FIELDS:
                cur[FIELDS[k.strip()]] = int(v)
    return kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=8)
    ap.add_argument("-o", default=str(ROOT / "profiles" / "r5_kernel_resources.json"))
    a = ap.parse_args()
    srcs = sorted(CSRC.glob("*.hip"))
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(a.j) as ex:
        ks = [k for res in ex.map(lambda s: unit(s, tmp), srcs) for k in res]
    for k, name in zip(ks, demangle([k["mangled"] for k in ks])):
        k["kernel"] = name
        del k["mangled"]
    ks.sort(key=lambda k: (k["kernel"], k["unit"]))
    out = {"build_id": build.source_id(), "flags": build.FLAGS,
           "note": "hipcc -Rpass-analysis=kernel-resource-usage per translation unit (static; no GPU). graph_search_kernel / graph_insert_search_kernel"
                   "<METRIC (0 sq_euclid, 1 cosine, 2 ucosine, 3 int8 records), NS (register sets of 64 beam entries), HASHED (visited set as an id hash "
                   "table: graphs too large for a bitset per resident wave), FORM>: form 0 plain, 1 latency variant (two waves per job), 2 lean "
                   "(launches without visited sets). SGPR spills go to VGPR lanes (v_writelane / v_readlane), not to scratch, while "
                   "scratch_bytes_per_lane is 0.",
           "kernels": ks}
    Path(a.o).write_text(json.dumps(out, indent=1))
    hot = [k for k in ks if re.search(r"graph_(search|insert_search|link|range|relink)_kernel", k["kernel"])]
    for k in hot:
        print(f"{k['kernel']:<70} vgprs {k.get('vgprs'):>3} waves/SIMD {k.get('waves_per_simd')} sgpr spills {k.get('sgpr_spills'):>3} "
              f"vgpr spills {k.get('vgpr_spills')} scratch {k.get('scratch_bytes_per_lane')} lds {k.get('lds_bytes_per_block')}")
    print(f"{len(ks)} kernels -> {a.o}", file=sys.stderr)


if __name__ == "__main__":
    main()
