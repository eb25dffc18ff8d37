"""Pins the oracle's metric arithmetic with the reference's own tolerances:
src/HNSWIndex.Tests/MetricsTests.cs:7-92 (1e-6 vs a scalar loop at dim 127 and 128) and
bindings/__tests__/metric_test.py:34-96 (atol 1e-5 vs float64)."""
import numpy as np
import pytest

import oracle

DIMS = [1, 7, 8, 9, 15, 16, 64, 96, 127, 128, 768]


def _vecs(dim, seed=65537, normalize=False):
    rng = np.random.default_rng(seed + dim)
    a = rng.random(dim, dtype=np.float32)
    b = rng.random(dim, dtype=np.float32)
    if normalize:
        a = a / np.float32(np.sqrt(np.sum(a * a, dtype=np.float32)))
        b = b / np.float32(np.sqrt(np.sum(b * b, dtype=np.float32)))
    return a, b


def _scalar_f32(metric, a, b):
    """MetricsTests.cs:94-136 reference loops, float32 accumulators."""
    if metric == "sq_euclid":
        s = np.float32(0)
        for x, y in zip(a, b):
            d = np.float32(x - y)
            s = np.float32(s + np.float32(d * d))
        return s
    dot = na = nb = np.float32(0)
    for x, y in zip(a, b):
        dot = np.float32(dot + np.float32(x * y))
        na = np.float32(na + np.float32(x * x))
        nb = np.float32(nb + np.float32(y * y))
    if metric == "ucosine":
        return np.float32(1) - dot
    return np.float32(1) - dot / np.float32(np.sqrt(np.float64(na)) * np.sqrt(np.float64(nb)))


@pytest.mark.parametrize("dim", DIMS)
@pytest.mark.parametrize("metric", ["sq_euclid", "cosine", "ucosine"])
def test_spec_and_avx_forms_agree_bitwise(metric, dim):
    if not oracle.lib().orc_has_avx2():
        pytest.skip("host CPU without AVX2+FMA")
    for seed in range(20):
        a, b = _vecs(dim, seed)
        s, v = oracle.metric(metric, a, b, False), oracle.metric(metric, a, b, True)
        assert s.tobytes() == v.tobytes()


@pytest.mark.parametrize("dim", [127, 128])
@pytest.mark.parametrize("metric", ["sq_euclid", "cosine", "ucosine"])
def test_reference_metric_unit_tolerance(metric, dim):
    a, b = _vecs(dim, normalize=(metric == "ucosine"))
    # identical vectors, as the reference's test effectively uses (same seed twice)
    assert abs(float(oracle.metric(metric, a, a)) - float(_scalar_f32(metric, a, a))) < 1e-6
    # and genuinely different ones, against float64 (1e-5: metric_test.py) --
    # float32 accumulation order differs from the scalar loop by more than 1e-6 at this size
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    want = {"sq_euclid": ((a64 - b64) ** 2).sum(),
            "cosine": 1 - a64 @ b64 / np.linalg.norm(a64) / np.linalg.norm(b64),
            "ucosine": 1 - a64 @ b64}[metric]
    assert abs(float(oracle.metric(metric, a, b)) - want) < 1e-5


def test_lane_order_is_the_avx_order_not_the_scalar_order():
    # A vector built so that summation order matters: the 8-lane order must give the
    # hand-computed value.
    dim = 16
    a = np.zeros(dim, dtype=np.float32)
    b = np.zeros(dim, dtype=np.float32)
    a[0], a[8] = 4096.0, 1e-3       # lane 0 chain: fma(1e-3,1e-3, 4096^2)
    a[4] = 1.0                      # lane 4
    acc0 = np.float32(np.float64(np.float32(1e-3)) ** 2 + np.float64(4096.0) ** 2)  # one rounding (fma)
    want = np.float32(np.float32(acc0 + np.float32(1.0)))
    assert oracle.metric("sq_euclid", a, b).tobytes() == want.tobytes()


def test_cosine_zero_vector_returns_one():
    z = np.zeros(128, dtype=np.float32)
    a, _ = _vecs(128)
    assert oracle.metric("cosine", z, a) == np.float32(1.0)  # CosineMetric.cs:89-90
    assert oracle.metric("cosine", a, z) == np.float32(1.0)


def test_symmetry_bitwise():
    for dim in (64, 127, 128):
        a, b = _vecs(dim)
        for m in ("sq_euclid", "cosine", "ucosine"):
            assert oracle.metric(m, a, b).tobytes() == oracle.metric(m, b, a).tobytes()


def test_tail_path_dim_below_8():
    a, b = _vecs(5)
    s = np.float32(0)
    for x, y in zip(a, b):
        d = np.float32(x - y)
        s = np.float32(s + np.float32(d * d))
    assert oracle.metric("sq_euclid", a, b).tobytes() == s.tobytes()
