"""CPU tier: `bench.py --gpus N --dry-run` validates the plan of an N-rank run (launcher, rendezvous, shard bounds, memory per rank)
without touching a GPU.  No multi-GPU measurement exists yet; this keeps the first real one from dying of a plan mistake."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _plan(*args):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args, "--dry-run"], capture_output=True, text=True, timeout=300)
    assert r.stdout.strip(), r.stderr[-2000:]
    return r.returncode, json.loads(r.stdout.strip().splitlines()[-1])


def test_eight_rank_plan_of_the_drivers_scaling_bench():
    rc, d = _plan("--gpus", "8")
    assert rc == 0 and d["ok"] and d["dry_run"] and not d["problems"]
    p = d["plan"]
    assert p["processes"] == 8 and len(p["shards"]) == 8 and p["scaling"] == "weak"
    assert all(hi - lo == 65536 for lo, hi in p["shards"]) and p["shards"][0][0] == 0 and p["shards"][-1][1] == 8 * 65536
    assert "torch.distributed.run" in p["launch"] and "--nproc-per-node=8" in p["launch"] and "127.0.0.1" in p["launch"] and "--dry-run" not in p["launch"]
    assert p["collective_per_step"]["bytes_gathered"] == 8 * 65536 * 10 * 8
    assert d["measured_on_more_than_one_gpu"] is False
    assert d["memory"]["hbm_GiB_per_gpu"] < 288 and d["memory"]["host_GiB_per_rank"] > 1.0


def test_strong_scaling_shards_tile_an_uneven_query_set():
    import importlib.util
    rc, d = _plan("--gpus", "3", "--nq", "1000", "--scaling", "strong", "--n", "50000")
    assert rc == 0 and d["ok"]
    sh = d["plan"]["shards"]
    assert [hi - lo for lo, hi in sh] == [334, 333, 333] and sh[0][0] == 0 and sh[-1][1] == 1000
    assert all(sh[i][1] == sh[i + 1][0] for i in range(2))


def test_a_plan_that_cannot_fit_is_refused():
    rc, d = _plan("--gpus", "8", "--n", "400000000", "--dim", "256")     # 410 GB of rows per replica: beyond one MI355X's 288 GB
    assert rc == 1 and not d["ok"] and any("HBM" in p for p in d["problems"])


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pmc_traffic_is_quoted_only_for_the_running_build_and_workload(tmp_path):
    """roofline.traffic comes from a committed --pmc profile: only the entry of the very workload, and only if its passes ran on the
    library that is running (an entry without an id of its own falls under the id of the whole file)."""
    b = _bench_module()
    wl = {"n": 1000000, "dim": 128, "queries_per_gpu_per_step": 65536, "ef_search": 128, "k": 10, "max_edges": 16}
    entry = {"workload": wl, "row_bytes_fetched": 512, "graph_search_kernel": {"traffic_bytes_per_launch": 1.4e11}, "insert_search": {"x": 1}}
    other = {"workload": {**wl, "n": 10000000}, "row_bytes_fetched": 512, "build_id": "B", "graph_search_kernel": {"traffic_bytes_per_launch": 7.0}}
    f = tmp_path / "r5_pmc_traffic.json"
    f.write_text(json.dumps({"build_id": "OLD", "configs": {"c2": {**entry, "build_id": "A"}, "c4": other}}))
    key = tuple(wl.values())
    t, add, note = b.quote_pmc_traffic(f, key, 512, "A")
    assert t == 140000000000 and add == {"insert_search": {"x": 1}} and "this build" in note
    t, add, note = b.quote_pmc_traffic(f, key, 512, "B")                 # the same workload on another library: not quoted, and the note says why
    assert t is None and add == {} and "measured on build A" in note
    assert b.quote_pmc_traffic(f, (10000000, 128, 65536, 128, 10, 16), 512, "B")[0] == 7
    assert b.quote_pmc_traffic(f, key, 128, "A")[0] is None              # another row size
    assert b.quote_pmc_traffic(f, None, 512, "A")[0] is None             # host traversal / clustered data: no profile of that
    f.write_text(json.dumps({"build_id": "A", "configs": {"c2": entry}}))   # a file from before the per-configuration ids
    assert b.quote_pmc_traffic(f, key, 512, "A")[0] == 140000000000 and b.quote_pmc_traffic(f, key, 512, "B")[0] is None
    assert b.quote_pmc_traffic(tmp_path / "missing.json", key, 512, "A") == (None, {}, "no PMC profile of this workload under profiles/")


def test_the_committed_pmc_profile_names_a_build_per_configuration():
    pm = json.loads((ROOT / "profiles" / "r5_pmc_traffic.json").read_text())
    assert pm["configs"] and all(len(c.get("build_id", "")) == 64 for c in pm["configs"].values())
