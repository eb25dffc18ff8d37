#!/usr/bin/env python3
"""Writes the inputs tools/dotnet_fixture/Program.cs runs the real reference on: the vectors / queries / parameters of
tests/golden/*.json's sequential cases (same seeds), a tie-heavy integer-grid case (equal distances everywhere: Span.Sort's
and the heaps' order among equal keys decides ids), a case that grows from a tiny CollectionSize, and one with removals.
numpy's generator is not reproducible from C#: the vectors travel as raw little-endian float32."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
OUT = Path(__file__).resolve().parent / "inputs"


def data(n, dim, metric, seed):
    x = np.random.default_rng(seed).random((n, dim), dtype=np.float32)
    if metric == "ucosine":  # Utils.Normalize (src/HNSWIndex.Tests/Utils.cs:23-30), in float32
        x = x / np.sqrt((x * x).sum(axis=1, dtype=np.float32, keepdims=True))
    return x.astype(np.float32)


def grid(n, dim, seed):      # coordinates on a 1/4 grid: distances collide all the time
    return (np.random.default_rng(seed).integers(0, 4, (n, dim)) / 4).astype(np.float32)


def cases():
    out = []
    for name, n, dim, metric, params, nq, k in (
            ("c1_sq_euclid_seq", 1500, 64, "sq_euclid", dict(max_edges=16, max_candidates=100, min_nn=5), 64, 10),
            ("cosine_seq", 800, 128, "cosine", dict(max_edges=16, max_candidates=100, min_nn=5), 32, 10),
            ("ucosine_m8_seq", 800, 96, "ucosine", dict(max_edges=8, max_candidates=40, min_nn=16), 32, 5),
            ("sq_euclid_dim127_seq", 500, 127, "sq_euclid", dict(max_edges=12, max_candidates=60, min_nn=1), 32, 3)):
        out.append((dict(name=name, n=n, dim=dim, metric=metric, nq=nq, k=k, collection_size=n, random_seed=31337, **params),
                    data(n, dim, metric, 65537), data(nq, dim, metric, 65538)))
    out.append((dict(name="grid_ties", n=600, dim=6, metric="sq_euclid", nq=48, k=10, max_edges=6, max_candidates=30, min_nn=12,
                     collection_size=600, random_seed=7, range=0.5), grid(600, 6, 11), grid(48, 6, 12)))
    out.append((dict(name="resize_from_10", n=700, dim=16, metric="sq_euclid", nq=32, k=5, max_edges=8, max_candidates=50, min_nn=5,
                     collection_size=10, random_seed=31337), data(700, 16, "sq_euclid", 21), data(32, 16, "sq_euclid", 22)))
    out.append((dict(name="removals", n=500, dim=24, metric="cosine", nq=32, k=8, max_edges=8, max_candidates=60, min_nn=8,
                     collection_size=512, random_seed=99, remove=[int(i) for i in np.random.default_rng(5).permutation(500)[:120]]),
                data(500, 24, "cosine", 31), data(32, 24, "cosine", 32)))
    # C2-shaped (BASELINE configs[1] at a size one core builds in seconds): i.i.d. uniform 128-d rows, M = 16, efConstruction = 200,
    # ef = 128.  At this beam width ~2 % of the inserts find two EQUAL float distances among their 200 candidates, and for ~0.3 % the
    # selection depends on the order Span.Sort leaves equal keys in (Heuristic.cs:22; DESIGN.md 4.2 rule 3): 20 000 sequential inserts
    # exercise a few dozen of those -- the BCL behaviour the oracle restates from memory.  (Its snapshot is ~12 MB: rows included.)
    out.append((dict(name="c2_shape_20k", n=20000, dim=128, metric="sq_euclid", nq=64, k=10, max_edges=16, max_candidates=200, min_nn=128,
                     collection_size=20000, random_seed=31337, allow_removals=False), data(20000, 128, "sq_euclid", 65537), data(64, 128, "sq_euclid", 65538)))
    return out


if __name__ == "__main__":
    OUT.mkdir(exist_ok=True)
    listing = []
    for c, x, q in cases():
        x.tofile(OUT / f"{c['name']}.x.f32")
        q.tofile(OUT / f"{c['name']}.q.f32")
        listing.append(c)
    (OUT / "cases.json").write_text(json.dumps(listing, indent=1))
    print(f"{len(listing)} cases under {OUT}")
    sys.exit(0)
