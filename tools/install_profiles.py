#!/usr/bin/env python3
"""Copies the summaries produced by tools/run_profiles.sh (gpurun_out/profiles/) into profiles/
under the round's prefix and derives rN_pmc_traffic.json.   python tools/install_profiles.py r2"""
import json
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src, dst, r = ROOT / "gpurun_out" / "profiles", ROOT / "profiles", sys.argv[1]


def json_line(path):
    return "".join(l for l in open(path) if l.startswith('{"metric"'))


shutil.copy(src / "s_kernel_stats.csv", dst / f"{r}_bench_1m_kernel_stats.csv")
shutil.copy(src / "kernel_stats.json", dst / f"{r}_bench_1m_kernel_stats.json")
shutil.copy(src / "pmc_fetch.json", dst / f"{r}_pmc_fetch_size.json")
shutil.copy(src / "pmc_write.json", dst / f"{r}_pmc_write_size.json")
shutil.copy(src / "pmc_calibration.json", dst / f"{r}_pmc_calibration_kbench.json")
(dst / f"{r}_pmc_calibration_kbench.log").write_text("".join(l for l in open(src / "kbench_calibration.log") if l.startswith(("N=", "v1"))))
for f, name in (("bench_under_rocprof", "bench_under_rocprof"), ("bench_pmc_fetch", "bench_pmc_fetch"), ("bench_pmc_write", "bench_pmc_write"),
                ("bench_plain", "bench_1m_plain"), ("bench_host_cabi", "bench_host_cabi_200k"), ("bench_c3", "bench_c3_1m_768_ucosine"),
                ("bench_c4_size", "bench_c4_10m_1gpu"), ("bench_c5_size", "bench_c5_10m_96_int8_1gpu"), ("bench_clustered", "bench_1m_clustered"),
                ("bench_2rank_gloo", "bench_2rank_gloo_rehearsal_200k")):
    if (src / f"{f}.log").exists() and json_line(src / f"{f}.log"):
        (dst / f"{r}_{name}.json").write_text(json_line(src / f"{f}.log"))
if (src / "kernel_stats_host_cabi.json").exists():
    shutil.copy(src / "kernel_stats_host_cabi.json", dst / f"{r}_host_cabi_kernel_stats.json")


def counter(path, name, kernel):
    cs = json.load(open(path))["counters"]
    key = [k for k in cs if kernel in k][0]  # sq_euclid instantiation (any register-set count)
    return cs[key][name]["avg_per_dispatch"], cs[key][name]["dispatches"], cs[key][name]["sum"]


cal = json.load(open(dst / f"{r}_pmc_calibration_kbench.json"))["counters"]["v1"]["FETCH_SIZE"]["avg_per_dispatch"]
m = re.search(r"single ([0-9.]+) us \(([0-9.]+) GB/s\)", [l for l in open(dst / f"{r}_pmc_calibration_kbench.log") if l.startswith("v1")][0])
known = float(m.group(2)) * 1e9 * float(m.group(1)) * 1e-6
factor = known / (cal * 1024)
b = json.load(open(dst / f"{r}_bench_pmc_fetch.json"))
fetch, _, _ = counter(dst / f"{r}_pmc_fetch_size.json", "FETCH_SIZE", "graph_search_kernel<0")
write, _, _ = counter(dst / f"{r}_pmc_write_size.json", "WRITE_SIZE", "graph_search_kernel<0")
alg = b["roofline"]["evals_per_launch"] * b["roofline"]["bytes_per_eval"]
out = {"round": int(r[1:]), "kernel": "graph_search_kernel<sq_euclid>",
       "workload": {"n": b["config"]["n"], "dim": b["config"]["dim"], "nq": b["config"]["queries_per_gpu_per_step"],
                    "ef_search": b["config"]["ef_search"], "k": b["config"]["k"], "max_edges": b["config"]["max_edges"]},
       "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
       "fetch_calibration": {"method": "tools/kbench (same 8-lane strided-dword row gather), 32768 slots x ~21 distinct random 512-B rows of a 2 GB matrix",
                             "known_bytes_per_launch": known, "FETCH_SIZE_KB_per_launch": cal, "factor": factor},
       "traffic_bytes_per_launch": fetch * 1024 * factor + write * 1024, "algorithmic_bytes_per_launch": alg}
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / alg
# the Add kernels: counter sums over the whole build against the build's algorithmic bytes
ra = b["roofline_add"]
add = {}
for kern, part in (("graph_insert_search_kernel<0", "insert_search"), ("graph_link_kernel<0", "link_half")):
    _, _, fs = counter(dst / f"{r}_pmc_fetch_size.json", "FETCH_SIZE", kern)
    _, _, ws = counter(dst / f"{r}_pmc_write_size.json", "WRITE_SIZE", kern)
    a = ra[part]["evals"] * ra["bytes_per_eval"]
    add[part] = {"FETCH_SIZE_KB_total": fs, "WRITE_SIZE_KB_total": ws, "traffic_bytes_total": fs * 1024 * factor + ws * 1024,
                 "algorithmic_bytes_total": a, "traffic_over_algorithmic": (fs * 1024 * factor + ws * 1024) / a}
out["add_kernels_whole_build"] = add
json.dump(out, open(dst / f"{r}_pmc_traffic.json", "w"), indent=1)
ks = json.load(open(dst / f"{r}_bench_1m_kernel_stats.json"))
for k in ks["kernel_stats"]:
    print(k["name"][:60], k["calls"], round(k["total_ns"] / 1e9, 3), "s  avg", round(k["avg_ns"] / 1e3, 1), "us")
for f in (f"{r}_bench_under_rocprof", f"{r}_bench_1m_plain"):
    j = json.load(open(dst / f"{f}.json"))
    print(f, j["value"], j["add_per_sec"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"], j.get("cpu_baseline") and j["cpu_baseline"]["value"])
print("traffic/algorithmic", out["traffic_over_algorithmic"], "factor", factor, {k: round(v["traffic_over_algorithmic"], 3) for k, v in add.items()})
