"""GPU tier: hnsw_knn_query on a query set large enough for the STREAMED upload (32 768 queries and more: the launch starts
on the first rows and the rest arrives behind it, Device::set_queries_streamed) where the kernel outruns the copy -- a small
index at the default MinNN, so a traversal is a few expansions long and the waves reach the gate (graph_search_kernel's
`ready`) all the time.  Shadows must never run ahead of a row that has not landed, a gate time-out must go through the job
word; answers are those of the plain upload and of the oracle, bit for bit.  Each configuration is a process of its own:
the stream switch is read once per process."""
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from common import default_cap

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, {root!r})
import hnswindex
dim, n, nq = {dim}, 3000, {nq}
x = np.random.default_rng(5).random((n, dim), dtype=np.float32)
ix = hnswindex.Index(dim); ix.set_collection_size(n)
ix.add(x)
out = []
for rep in range(3):   # fresh host buffers each call, like bench.py's rotating sets
    q = np.random.default_rng(100 + rep).random((nq, dim), dtype=np.float32)
    ids, d = ix.knn_query(q, 10)
    out.append([hashlib.sha256(ids.tobytes()).hexdigest(), hashlib.sha256(d.tobytes()).hexdigest()])
np.save({sample!r}, np.concatenate([ids[:400].astype(np.float64), d[:400].astype(np.float64)], axis=1))
print(json.dumps({{"digests": out, "graph_hash": int(ix.graph_hash())}}))
"""


def _run(tmp_path, dim, nq, env_extra, tag):
    sample = str(tmp_path / f"sample_{tag}.npy")
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", WORKER.format(root=str(ROOT), dim=dim, nq=nq, sample=sample)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), np.load(sample)


@pytest.mark.parametrize("dim,nq", [(128, 40000), (512, 33000)])
def test_streamed_upload_answers_like_the_plain_one_and_the_oracle(tmp_path, dim, nq):
    plain, s_plain = _run(tmp_path, dim, nq, {"HNSW_MI355X_DIAG": "stream_queries=0"}, "plain")
    for shadow in ("1", "0"):
        got, s_got = _run(tmp_path, dim, nq, {"HNSW_MI355X_DIAG": "stream_queries=1,shadow=" + shadow}, "s" + shadow)
        assert got["graph_hash"] == plain["graph_hash"]
        assert got["digests"] == plain["digests"], f"streamed upload (shadows {shadow}) answered differently"
        assert s_got.tobytes() == s_plain.tobytes()
    # and the plain answers are the reference algorithm's
    import oracle
    x = np.random.default_rng(5).random((3000, dim), dtype=np.float32)
    ref = oracle.OracleIndex(dim, "sq_euclid", collection_size=3000)
    ref.add_batched(x, default_cap())  # the default schedule of one hnsw_add call (orc_add_batched, cap = the host's hardware threads)
    assert ref.graph_hash() == plain["graph_hash"]
    q = np.random.default_rng(102).random((nq, dim), dtype=np.float32)[:400]
    ids, d = ref.knn_query(q, 10)
    assert (ids.astype(np.float64) == s_plain[:, :10]).all() and (d.astype(np.float64) == s_plain[:, 10:]).all()
