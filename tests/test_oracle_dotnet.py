"""Pins the oracle's restatements of .NET BCL behaviour that lives OUTSIDE /root/reference
(System.Random(seed), Span.Sort, BinaryHeap semantics) -- SURVEY.md 8c."""
import math

import numpy as np

import oracle


def test_random_known_answers():
    # Publicly known outputs of .NET's seeded System.Random (Knuth subtractive generator).
    assert oracle.dotnet_random_next(0, 3).tolist() == [1559595546, 1755192844, 1649316166]
    assert oracle.dotnet_random_next(42, 3).tolist() == [1434747710, 302596119, 269548474]
    assert oracle.dotnet_random_next(1, 2).tolist() == [534011718, 237820880]
    d = oracle.dotnet_random_double(42, 2)
    assert abs(d[0] - 0.668106465911542) < 1e-15  # new Random(42).NextDouble()
    assert (d == oracle.dotnet_random_next(42, 2) * (1.0 / 2147483647)).all()  # Sample() = InternalSample() * (1.0 / int.MaxValue)


def test_random_negative_seed_is_abs():
    # new Random(-5) == new Random(5) in .NET (Math.Abs(seed))
    assert (oracle.dotnet_random_next(-5, 8) == oracle.dotnet_random_next(5, 8)).all()


def test_next_single_is_float_of_sample():
    d = oracle.dotnet_random_double(31337, 1000)
    s = oracle.dotnet_random_single(31337, 1000)
    assert (d.astype(np.float32) == s).all()
    assert (s >= 0).all() and (s <= 1).all()


def test_levels_follow_reference_formula():
    # GraphData.cs:211-219: (int)(-Math.Log(random) * distRate), one draw per insert
    rate = 1.0 / math.log(16)
    s = oracle.dotnet_random_single(31337, 5000)
    want = np.array([int(-math.log(float(x)) * rate) for x in s], dtype=np.int32)
    got = oracle.random_levels(31337, rate, 5000)
    assert (got == want).all()
    # exponential with base 16: ~15/16 of the nodes sit on layer 0
    frac0 = (oracle.random_levels(7, rate, 200000) == 0).mean()
    assert abs(frac0 - 15 / 16) < 0.005


def test_sort_is_a_sort_and_matches_numpy_on_distinct_keys():
    rng = np.random.default_rng(1)
    for n in (0, 1, 2, 3, 5, 16, 17, 33, 100, 200, 401, 1000):
        d = rng.permutation(n).astype(np.float32)  # distinct keys: any correct sort agrees
        ids = np.arange(n, dtype=np.int32)
        sid, sd = oracle.dotnet_sort(ids, d)
        order = np.argsort(d, kind="stable")
        assert (sid == ids[order]).all() and (sd == d[order]).all()


def test_sort_with_ties_is_a_permutation_and_sorted():
    rng = np.random.default_rng(2)
    for n in (17, 64, 300):
        d = rng.integers(0, 5, n).astype(np.float32)
        ids = np.arange(n, dtype=np.int32)
        sid, sd = oracle.dotnet_sort(ids, d)
        assert (np.diff(sd) >= 0).all()
        assert sorted(sid.tolist()) == list(range(n))
        assert (d[sid] == sd).all()


def _py_heap_script(closer_first, ops, dists):
    """Pure-Python BinaryHeap (src/HNSWIndex/BinaryHeap.cs:30-107) for small cases."""
    def cmp(x, y):  # DistanceComparer / ReverseDistanceComparer
        a, b = (y[1], x[1]) if closer_first else (x[1], y[1])
        return -1 if a < b else (1 if a > b else 0)
    buf, popped = [], []
    for op, d in zip(ops, dists):
        if op >= 0:
            item = (int(op), float(d))
            buf.append(item)
            i = len(buf) - 1
            while i > 0:
                p = (i - 1) >> 1
                if cmp(item, buf[p]) <= 0:
                    break
                buf[i] = buf[p]
                i = p
            buf[i] = item
        elif buf:
            popped.append(buf[0][0])
            last = buf.pop()
            n = len(buf)
            if n:
                i, half = 0, n >> 1
                while i < half:
                    left, right = 2 * i + 1, 2 * i + 2
                    mc = right if (right < n and cmp(buf[left], buf[right]) < 0) else left
                    if cmp(buf[mc], last) <= 0:
                        break
                    buf[i] = buf[mc]
                    i = mc
                buf[i] = last
    return [b[0] for b in buf], popped


def test_heap_matches_python_restatement_including_ties():
    rng = np.random.default_rng(3)
    for closer in (0, 1):
        for trial in range(20):
            n = 200
            ops = np.where(rng.random(n) < 0.7, np.arange(n), -1).astype(np.int32)
            d = rng.integers(0, 12, n).astype(np.float32)  # many ties
            ids, dd, popped = oracle.heap_script(closer, ops, d)
            want_ids, want_popped = _py_heap_script(closer, ops, d)
            assert ids.tolist() == want_ids
            assert popped.tolist() == want_popped
