"""
oracle -- TEST INFRASTRUCTURE ONLY (see oracle/hnsw_oracle.c header).

ctypes front-end for the C restatement of the reference CPU path.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the
product package.
"""
import ctypes as ct
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "_build" / "libhnsw_oracle.so"

METRICS = {"sq_euclid": 0, "cosine": 1, "ucosine": 2, "sq_euclid_i8": 3}

_F = ct.POINTER(ct.c_float)
_I = ct.POINTER(ct.c_int)


def build(force: bool = False) -> Path:
    """Compile oracle/hnsw_oracle.c with the committed Makefile (gcc)."""
    src = _DIR / "hnsw_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR), "-B"], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ct.CDLL(str(_SO))
        L.orc_create.restype = ct.c_void_p
        L.orc_create.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_double, ct.c_int, ct.c_int, ct.c_int, ct.c_int,
                                 ct.c_int, ct.c_int]
        L.orc_free.argtypes = [ct.c_void_p]
        L.orc_free.restype = None
        L.orc_add.argtypes = [ct.c_void_p, _F, ct.c_int, _I]
        L.orc_add_batched.argtypes = [ct.c_void_p, _F, ct.c_int, _I, ct.c_int]
        L.orc_add_batched_mt.argtypes = [ct.c_void_p, _F, ct.c_int, _I, ct.c_int, ct.c_int]
        L.orc_add_ticks.argtypes = [ct.c_void_p, _F, ct.c_int, _I, ct.c_int, ct.POINTER(ct.c_uint64)]
        L.orc_i8_pitch.argtypes = [ct.c_int]
        L.orc_i8_quantize.argtypes = [_F, ct.c_int, _I]
        L.orc_i8_quantize.restype = None
        L.orc_rng_skip.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_rng_skip.restype = None
        L.orc_import_nodes.argtypes = [ct.c_void_p, _F, _I, ct.c_int, ct.c_int]
        L.orc_import_edges.argtypes = [ct.c_void_p, ct.c_int, _I, _I, ct.c_int, ct.c_int]
        L.orc_range_query.argtypes = [ct.c_void_p, _F, ct.c_int, ct.c_float, ct.c_int, _I, _I, _F]
        L.orc_remove.argtypes = [ct.c_void_p, _I, ct.c_int]
        L.orc_remove_batched.argtypes = [ct.c_void_p, _I, ct.c_int, ct.c_int]
        L.orc_active_ids.argtypes = [ct.c_void_p, _I, ct.c_int]
        L.orc_length.argtypes = [ct.c_void_p]
        L.orc_set_remove_max_candidates.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_set_remove_max_candidates.restype = None
        L.orc_knn_query.argtypes = [ct.c_void_p, _F, ct.c_int, ct.c_int, _I, _F, ct.c_int]
        for name in ("orc_count", "orc_entry_point", "orc_capacity"):
            getattr(L, name).argtypes = [ct.c_void_p]
        L.orc_node_max_layer.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_get_edges.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, _I, ct.c_int]
        L.orc_n_eval.argtypes = [ct.c_void_p]
        L.orc_n_eval.restype = ct.c_uint64
        L.orc_reset_n_eval.argtypes = [ct.c_void_p]
        L.orc_reset_n_eval.restype = None
        L.orc_graph_hash.argtypes = [ct.c_void_p]
        L.orc_graph_hash.restype = ct.c_uint64
        L.orc_metric.argtypes = [ct.c_int, _F, _F, ct.c_int, ct.c_int]
        L.orc_metric.restype = ct.c_float
        L.orc_dist_query_rows.argtypes = [ct.c_int, _F, ct.c_int, _F, _I, ct.c_int, _F, ct.c_int]
        L.orc_dist_query_rows.restype = None
        L.orc_dist_pairs.argtypes = [ct.c_int, _F, ct.c_int, _I, _I, ct.c_int, _F, ct.c_int]
        L.orc_dist_pairs.restype = None
        L.orc_random_next.argtypes = [ct.c_int, ct.c_int, _I]
        L.orc_random_next_double.argtypes = [ct.c_int, ct.c_int, ct.POINTER(ct.c_double)]
        L.orc_random_next_single.argtypes = [ct.c_int, ct.c_int, _F]
        L.orc_random_levels.argtypes = [ct.c_int, ct.c_double, ct.c_int, _I]
        L.orc_next_single_from_samples.argtypes = [_I, ct.c_int, _I]
        L.orc_next_single_from_samples.restype = ct.c_float
        L.orc_sort_nd.argtypes = [_I, _F, ct.c_int]
        L.orc_heap_script.argtypes = [ct.c_int, _I, _F, ct.c_int, _I, _F, _I, _I]
        L.orc_search_layer.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, _F, _I, _F]
        L.orc_find_entry_point.argtypes = [ct.c_void_p, ct.c_int, _F]
        L.orc_alloc_only.argtypes = [ct.c_void_p, _F, ct.c_int, _I]
        L.orc_connect_allocated.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_connect_allocated.restype = None
        L.orc_window_search.argtypes = [ct.c_void_p, ct.c_int, _I, _I, ct.c_int]
        L.orc_window_dry.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, _I, _I]
        L.orc_window_link.argtypes = [ct.c_void_p, ct.c_int, _I, _I, ct.c_int]
        L.orc_window_link.restype = None
        L.orc_dist_ids.argtypes = [ct.c_void_p, ct.c_int, ct.c_int]
        L.orc_dist_ids.restype = ct.c_float
        L.orc_access_log.argtypes = [ct.c_void_p, ct.c_longlong]
        L.orc_access_log.restype = None
        L.orc_access_log_fetch.argtypes = [ct.c_void_p, ct.POINTER(ct.c_int64), ct.c_longlong]
        L.orc_access_log_fetch.restype = ct.c_longlong
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _pf(a):
    return a.ctypes.data_as(_F)


def _pi(a):
    return a.ctypes.data_as(_I)


# ---------------------------------------------------------------- unit pieces
def metric(name, a, b, use_avx=False) -> np.float32:
    a, b = _f32(a), _f32(b)
    assert a.shape == b.shape and a.ndim == 1
    return np.float32(lib().orc_metric(METRICS[name], _pf(a), _pf(b), a.size, int(use_avx)))


def dist_query_rows(name, rows, q, ids, use_avx=False):
    rows, q, ids = _f32(rows), _f32(q), _i32(ids)
    out = np.empty(ids.size, dtype=np.float32)
    lib().orc_dist_query_rows(METRICS[name], _pf(rows), rows.shape[1], _pf(q), _pi(ids), ids.size, _pf(out),
                              int(use_avx))
    return out


def dist_pairs(name, rows, a, b, use_avx=False):
    rows, a, b = _f32(rows), _i32(a), _i32(b)
    out = np.empty(a.size, dtype=np.float32)
    lib().orc_dist_pairs(METRICS[name], _pf(rows), rows.shape[1], _pi(a), _pi(b), a.size, _pf(out), int(use_avx))
    return out


def i8_quantize(x):
    """The int8 record of each row of x (see oracle/hnsw_oracle.c "int8 rows"): (q int8 [n, dim], scale float32 [n],
    sumsq int32 [n])."""
    x = _f32(x).reshape(-1, np.shape(x)[-1])
    n, dim = x.shape
    pitch = lib().orc_i8_pitch(dim)
    rec = np.empty((n, pitch), dtype=np.int32)
    for i in range(n):
        lib().orc_i8_quantize(_pf(x[i]), dim, _pi(rec[i]))
    q = rec[:, :pitch - 2].copy().view(np.int8)[:, :dim]
    return q, rec[:, pitch - 2].copy().view(np.float32), rec[:, pitch - 1].copy()


def dotnet_random_next(seed, n):
    out = np.empty(n, dtype=np.int32)
    lib().orc_random_next(seed, n, _pi(out))
    return out


def dotnet_random_double(seed, n):
    out = np.empty(n, dtype=np.float64)
    lib().orc_random_next_double(seed, n, out.ctypes.data_as(ct.POINTER(ct.c_double)))
    return out


def dotnet_random_single(seed, n):
    out = np.empty(n, dtype=np.float32)
    lib().orc_random_next_single(seed, n, _pf(out))
    return out


def random_levels(seed, rate, n):
    out = np.empty(n, dtype=np.int32)
    lib().orc_random_levels(seed, float(rate), n, _pi(out))
    return out


def dotnet_sort(ids, dists):
    ids, dists = _i32(ids).copy(), _f32(dists).copy()
    lib().orc_sort_nd(_pi(ids), _pf(dists), ids.size)
    return ids, dists


def heap_script(closer_first, ops, dists):
    ops, dists = _i32(ops), _f32(dists)
    n = ops.size
    out_ids = np.empty(n, dtype=np.int32)
    out_d = np.empty(n, dtype=np.float32)
    popped = np.empty(n, dtype=np.int32)
    npop = ct.c_int(0)
    c = lib().orc_heap_script(int(closer_first), _pi(ops), _pf(dists), n, _pi(out_ids), _pf(out_d), _pi(popped),
                              ct.byref(npop))
    return out_ids[:c].copy(), out_d[:c].copy(), popped[:npop.value].copy()


# ---------------------------------------------------------------- index
class OracleIndex:
    """Same surface as the reference's Python `Index` (bindings/bindings.py:172-521) for
    the float32 Add/KnnQuery path, plus graph introspection for parity checks."""

    def __init__(self, dim, metric="sq_euclid", *, max_edges=16, distribution_rate=None, min_nn=5,
                 max_candidates=100, collection_size=65536, random_seed=31337, allow_removals=True, use_avx=True,
                 remove_max_candidates=100):
        import math
        if distribution_rate is None:
            distribution_rate = 1.0 / math.log(16)  # HNSWParameters.cs:19
        self.dim, self.metric = dim, metric
        self._h = lib().orc_create(dim, METRICS[metric], max_edges, float(distribution_rate), min_nn, max_candidates,
                                   collection_size, random_seed, int(allow_removals), int(use_avx))
        if not self._h:
            raise RuntimeError("orc_create failed")
        lib().orc_set_remove_max_candidates(self._h, int(remove_max_candidates))

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().orc_free(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None

    def add(self, vecs):
        """Sequential HNSWIndex.Add(item) per row, in order."""
        a = _f32(vecs).reshape(-1, self.dim)
        ids = np.empty(a.shape[0], dtype=np.int32)
        lib().orc_add(self._h, _pf(a), a.shape[0], _pi(ids))
        return ids

    def add_batched(self, vecs, max_batch=4096, threads=1):
        """The product's snapshot-batched schedule (see orc_add_batched); threads > 1 spreads each
        batch's searches over that many host threads (same graph)."""
        a = _f32(vecs).reshape(-1, self.dim)
        ids = np.empty(a.shape[0], dtype=np.int32)
        lib().orc_add_batched_mt(self._h, _pf(a), a.shape[0], _pi(ids), int(max_batch), int(threads))
        return ids

    def add_ticks(self, vecs, slots=256):
        """The tick schedule (orc_add_ticks; the CPU model behind DESIGN.md 4.3, not a product path): at most `slots` items in
        flight, started in id order; a multi-layer item searches its top layer one tick ahead of the rest and links everything in
        its last tick.  -> (ids, {ticks, steps (item-ticks), max_in_flight, alone, long_ticks})"""
        a = _f32(vecs).reshape(-1, self.dim)
        ids = np.empty(a.shape[0], dtype=np.int32)
        st = (ct.c_uint64 * 5)()
        lib().orc_add_ticks(self._h, _pf(a), a.shape[0], _pi(ids), int(slots), st)
        return ids, {"ticks": int(st[0]), "steps": int(st[1]), "max_in_flight": int(st[2]), "alone": int(st[3]), "long_ticks": int(st[4])}

    # ---- the exact-window schedule taken apart (tests/test_window_model.py) ----
    def alloc_only(self, vecs):
        a = _f32(vecs).reshape(-1, self.dim)
        ids = np.empty(a.shape[0], dtype=np.int32)
        lib().orc_alloc_only(self._h, _pf(a), a.shape[0], _pi(ids))
        return ids

    def connect_allocated(self, i):
        lib().orc_connect_allocated(self._h, int(i))

    def window_search(self, i, stride=130):
        """Selections per layer of item i on the graph as it stands (nothing written): list of id arrays, layer 0 first."""
        sel = np.zeros((64, stride), dtype=np.int32)
        cnt = np.zeros(64, dtype=np.int32)
        top = lib().orc_window_search(self._h, int(i), _pi(sel), _pi(cnt), stride)
        return [sel[l, :cnt[l]].copy() for l in range(top + 1)]

    def window_dry(self, nb, layer, item):
        lost = np.zeros(8, dtype=np.int32)
        n = ct.c_int(0)
        code = lib().orc_window_dry(self._h, int(nb), int(layer), int(item), _pi(lost), ct.byref(n))
        return code, (None if n.value == 255 else lost[:n.value].tolist())

    def window_link(self, i, sels, stride=130):
        sel = np.zeros((64, stride), dtype=np.int32)
        cnt = np.zeros(64, dtype=np.int32)
        for l, s in enumerate(sels):
            cnt[l] = len(s)
            sel[l, :len(s)] = s
        lib().orc_window_link(self._h, int(i), _pi(sel), _pi(cnt), stride)

    def dist_ids(self, a, b):
        return np.float32(lib().orc_dist_ids(self._h, int(a), int(b)))

    def access_log(self, cap):
        """Record which adjacency lists the sequential Add reads / writes from now on (cap entries; 0 = off)."""
        lib().orc_access_log(self._h, int(cap))

    def access_log_fetch(self):
        """(kind, layer, node) arrays of the recorded entries: kind 0 read, 1 write, 2 item start."""
        n = lib().orc_access_log_fetch(self._h, None, 0)
        out = np.empty(max(1, n), dtype=np.int64)
        lib().orc_access_log_fetch(self._h, out.ctypes.data_as(ct.POINTER(ct.c_int64)), n)
        out = out[:n]
        return (out >> 60).astype(np.int32), ((out >> 40) & 0xFFFFF).astype(np.int32), (out & 0xFFFFFFFF).astype(np.int64)

    def rng_skip(self, n):
        """Advance the level generator by n draws (after import_graph: one per imported node)."""
        lib().orc_rng_skip(self._h, int(n))

    def import_graph(self, items, levels, entry, layers):
        """layers: list of (counts[n], edges[n, stride]) per layer 0.. as produced by the
        product's Index.export_edges.  In-edge lists are not rebuilt (Add after an import
        is only meaningful with allow_removals=False)."""
        items, levels = _f32(items), _i32(levels)
        n = levels.size
        if lib().orc_import_nodes(self._h, _pf(items), _pi(levels), n, int(entry)) != n:
            raise RuntimeError("orc_import_nodes failed")
        for layer, (counts, edges) in enumerate(layers):
            counts, edges = _i32(counts), _i32(edges)
            if lib().orc_import_edges(self._h, layer, _pi(counts), _pi(edges), edges.shape[1], n) != n:
                raise RuntimeError("orc_import_edges failed")

    def knn_query(self, queries, k, threads=1):
        q = _f32(queries).reshape(-1, self.dim)
        n = q.shape[0]
        ids = np.empty((n, k), dtype=np.int32)
        d = np.empty((n, k), dtype=np.float32)
        lib().orc_knn_query(self._h, _pf(q), n, k, _pi(ids), _pf(d), threads)
        return ids, d

    def range_query(self, queries, radius, cap=None):
        """HNSWIndex.RangeQuery per query: (list of id arrays, list of distance arrays)."""
        q = _f32(queries).reshape(-1, self.dim)
        n = q.shape[0]
        cap = cap or max(1, self.count)
        cnt = np.empty(n, dtype=np.int32)
        ids = np.empty((n, cap), dtype=np.int32)
        d = np.empty((n, cap), dtype=np.float32)
        if lib().orc_range_query(self._h, _pf(q), n, float(radius), cap, _pi(cnt), _pi(ids), _pf(d)) != 0:
            raise RuntimeError("orc_range_query: cap too small")
        return [ids[i, :cnt[i]].copy() for i in range(n)], [d[i, :cnt[i]].copy() for i in range(n)]

    @property
    def count(self):
        return lib().orc_count(self._h)

    @property
    def entry_point(self):
        return lib().orc_entry_point(self._h)

    def max_layer(self, i):
        return lib().orc_node_max_layer(self._h, int(i))

    def remove(self, ids):
        """HNSWIndex.Remove per id, in order (HNSWIndex.cs:83-102)."""
        a = _i32(ids).ravel()
        if lib().orc_remove(self._h, _pi(a), a.size) != 0:
            raise RuntimeError("orc_remove: removals disabled or invalid id")

    def remove_batched(self, ids, batch):
        """The builder's snapshot-batched removal schedule (hnsw_mi355x_set_remove_batch; NOT a reference code path)."""
        a = _i32(ids).ravel()
        if lib().orc_remove_batched(self._h, _pi(a), a.size, int(batch)) != 0:
            raise RuntimeError("orc_remove_batched: removals disabled or invalid id")

    def active_ids(self):
        out = np.empty(max(1, self.count), dtype=np.int32)
        n = lib().orc_active_ids(self._h, _pi(out), out.size)
        return out[:n].copy()

    @property
    def length(self):
        """Slots ever allocated (GraphData.Length)."""
        return lib().orc_length(self._h)

    def levels(self):
        return np.array([self.max_layer(i) for i in range(self.length)], dtype=np.int32)

    def edges(self, i, layer, incoming=False):
        buf = np.empty(4096, dtype=np.int32)
        n = lib().orc_get_edges(self._h, int(i), int(layer), int(incoming), _pi(buf), buf.size)
        if n < 0:
            raise IndexError((i, layer))
        return buf[:n].copy()

    def graph_hash(self):
        return int(lib().orc_graph_hash(self._h))

    @property
    def n_eval(self):
        return int(lib().orc_n_eval(self._h))

    def reset_n_eval(self):
        lib().orc_reset_n_eval(self._h)

    def search_layer(self, entry_id, layer, k, q):
        q = _f32(q)
        ids = np.empty(k, dtype=np.int32)
        d = np.empty(k, dtype=np.float32)
        n = lib().orc_search_layer(self._h, int(entry_id), int(layer), int(k), _pf(q), _pi(ids), _pf(d))
        return ids[:n].copy(), d[:n].copy()

    def find_entry_point(self, dst_layer, q):
        q = _f32(q)
        return lib().orc_find_entry_point(self._h, int(dst_layer), _pf(q))
