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
