"""
ctypes bindings over the C-ABI library -- the host-side mirror of the reference's Python
interface (/root/reference/bindings/bindings.py): same `Index` class, method names, argument
meaning, dtypes/shapes and error behaviour (RuntimeError carrying the native last-error
string on a negative status).  `DeviceBackend` exposes the inner hnswdev_* boundary.
"""
import ctypes as ct
import os
from pathlib import Path
from typing import List, Tuple

import numpy as np
import numpy.typing as npt

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "artifacts" / "native" / "linux-x64" / "HNSWIndex.Native.so"  # bindings.py:27-41


def _load_lib():
    # torch ships its own libamdhip64.so.7; load it first so that this process holds ONE
    # HIP runtime (the library's DT_NEEDED then resolves to the copy already mapped).
    if os.environ.get("HNSW_MI355X_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch absent: /opt/rocm's runtime is used
            pass
    override = os.environ.get("HNSW_MI355X_LIB")  # diagnostic builds only (e.g. the phase-clock variant, tools/)
    if override:
        return ct.CDLL(override)
    if not LIB_PATH.exists():
        raise FileNotFoundError(
            f"Native library missing {LIB_PATH}: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no pure-Python or CPU fallback")
    return ct.CDLL(str(LIB_PATH))


lib = _load_lib()

_F = ct.POINTER(ct.c_float)
_I = ct.POINTER(ct.c_int)


class Stats(ct.Structure):
    """hnswdev_stats (include/hnsw_mi355x.h)."""
    _fields_ = [("launches", ct.c_uint64), ("evals", ct.c_uint64), ("timed_launches", ct.c_uint64),
                ("timed_evals", ct.c_uint64), ("kernel_ms", ct.c_double), ("row_bytes", ct.c_uint64),
                ("search_launches", ct.c_uint64), ("search_evals", ct.c_uint64), ("search_timed_launches", ct.c_uint64),
                ("search_timed_evals", ct.c_uint64), ("search_kernel_ms", ct.c_double), ("search_overflows", ct.c_uint64),
                ("search_repeats", ct.c_uint64),
                ("insert_launches", ct.c_uint64), ("insert_evals", ct.c_uint64), ("insert_timed_launches", ct.c_uint64),
                ("insert_timed_evals", ct.c_uint64), ("insert_kernel_ms", ct.c_double),
                ("link_launches", ct.c_uint64), ("link_evals", ct.c_uint64), ("link_timed_launches", ct.c_uint64),
                ("link_timed_evals", ct.c_uint64), ("link_kernel_ms", ct.c_double), ("visited_hash_launches", ct.c_uint64),
                ("range_launches", ct.c_uint64), ("range_evals", ct.c_uint64), ("range_timed_launches", ct.c_uint64),
                ("range_timed_evals", ct.c_uint64), ("range_kernel_ms", ct.c_double), ("range_handbacks", ct.c_uint64),
                ("replica_bytes", ct.c_uint64),
                ("tie_windows", ct.c_uint64), ("peer_direct_copies", ct.c_uint64), ("peer_staged_copies", ct.c_uint64), ("lat_launches", ct.c_uint64),
                ("range_device_ordered", ct.c_uint64), ("range_host_ordered", ct.c_uint64), ("insert_tie_reruns", ct.c_uint64), ("lean_launches", ct.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# ---- (A) the reference's 16 exports: bindings.py:45-119 ---------------------------------
lib.hnsw_create.restype = ct.c_void_p
lib.hnsw_create.argtypes = [ct.c_char_p]
lib.hnsw_free.restype = None
lib.hnsw_free.argtypes = [ct.c_void_p]
lib.hnsw_add.restype = ct.c_int
lib.hnsw_add.argtypes = [ct.c_void_p, _F, ct.c_int, ct.c_int, _I]
lib.hnsw_remove.restype = ct.c_int
lib.hnsw_remove.argtypes = [ct.c_void_p, _I, ct.c_int]
lib.hnsw_knn_query.restype = ct.c_int
lib.hnsw_knn_query.argtypes = [ct.c_void_p, _F, ct.c_int, ct.c_int, ct.c_int, _I, _F]
lib.hnsw_range_query.restype = ct.c_int
lib.hnsw_range_query.argtypes = [ct.c_void_p, _F, ct.c_int, ct.c_int, ct.c_float, ct.POINTER(ct.c_void_p),
                                 ct.POINTER(ct.c_void_p), _I]
lib.hnsw_free_results.restype = None
lib.hnsw_free_results.argtypes = [ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_void_p), ct.c_int]
for _name in ("hnsw_set_collection_size", "hnsw_set_max_edges", "hnsw_set_max_candidates",
              "hnsw_set_remove_max_candidates", "hnsw_set_random_seed", "hnsw_set_min_nn",
              "hnsw_mi355x_set_device", "hnsw_mi355x_set_insert_batch", "hnsw_mi355x_set_remove_batch", "hnsw_mi355x_set_search_slots",
              "hnsw_mi355x_set_host_threads", "hnsw_mi355x_set_device_traversal", "hnsw_mi355x_set_devices"):
    getattr(lib, _name).restype = ct.c_int
    getattr(lib, _name).argtypes = [ct.c_int]
lib.hnsw_set_distribution_rate.restype = ct.c_int
lib.hnsw_set_distribution_rate.argtypes = [ct.c_float]
lib.hnsw_set_allow_removals.restype = ct.c_int
lib.hnsw_set_allow_removals.argtypes = [ct.c_bool]
lib.hnsw_get_last_error_utf8.restype = ct.c_int
lib.hnsw_get_last_error_utf8.argtypes = [ct.c_void_p, ct.c_int]

# ---- additions ---------------------------------------------------------------------------
for _name in ("hnsw_mi355x_count", "hnsw_mi355x_length", "hnsw_mi355x_entry_point", "hnsw_mi355x_reset_stats", "hnsw_mi355x_resident_count"):
    getattr(lib, _name).restype = ct.c_int
    getattr(lib, _name).argtypes = [ct.c_void_p]
lib.hnsw_mi355x_set_queries.restype = ct.c_int
lib.hnsw_mi355x_set_queries.argtypes = [ct.c_void_p, _F, ct.c_int, ct.c_int]
lib.hnsw_mi355x_knn_query_resident.restype = ct.c_int
lib.hnsw_mi355x_knn_query_resident.argtypes = [ct.c_void_p, ct.c_int, _I, _F]
lib.hnsw_mi355x_active_ids.restype = ct.c_int
lib.hnsw_mi355x_active_ids.argtypes = [ct.c_void_p, _I, ct.c_int]
lib.hnsw_mi355x_node_max_layer.restype = ct.c_int
lib.hnsw_mi355x_node_max_layer.argtypes = [ct.c_void_p, ct.c_int]
lib.hnsw_mi355x_get_out_edges.restype = ct.c_int
lib.hnsw_mi355x_get_out_edges.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, _I, ct.c_int]
lib.hnsw_mi355x_export_levels.restype = ct.c_int
lib.hnsw_mi355x_export_levels.argtypes = [ct.c_void_p, _I, ct.c_int]
lib.hnsw_mi355x_export_edges.restype = ct.c_int
lib.hnsw_mi355x_export_edges.argtypes = [ct.c_void_p, ct.c_int, _I, _I, ct.c_int, ct.c_int]
lib.hnsw_mi355x_dim.restype = ct.c_int
lib.hnsw_mi355x_dim.argtypes = [ct.c_void_p]
lib.hnsw_mi355x_serialize.restype = ct.c_int
lib.hnsw_mi355x_serialize.argtypes = [ct.c_void_p, ct.c_char_p]
lib.hnsw_mi355x_deserialize.restype = ct.c_void_p
lib.hnsw_mi355x_deserialize.argtypes = [ct.c_char_p, ct.c_char_p]
lib.hnsw_mi355x_graph_hash.restype = ct.c_uint64
lib.hnsw_mi355x_graph_hash.argtypes = [ct.c_void_p]
lib.hnsw_mi355x_get_stats.restype = ct.c_int
lib.hnsw_mi355x_get_stats.argtypes = [ct.c_void_p, ct.POINTER(Stats)]
lib.hnsw_mi355x_set_profiling.restype = ct.c_int
lib.hnsw_mi355x_set_profiling.argtypes = [ct.c_void_p, ct.c_int]
lib.hnsw_mi355x_index_set_insert_batch.restype = ct.c_int
lib.hnsw_mi355x_index_set_insert_batch.argtypes = [ct.c_void_p, ct.c_int]
lib.hnsw_mi355x_index_insert_batch.restype = ct.c_int
lib.hnsw_mi355x_index_insert_batch.argtypes = [ct.c_void_p]
lib.hnsw_mi355x_host_parallelism.restype = ct.c_int
lib.hnsw_mi355x_host_parallelism.argtypes = []
lib.hnsw_mi355x_device_count.restype = ct.c_int
lib.hnsw_mi355x_device_count.argtypes = [ct.c_void_p]
lib.hnsw_mi355x_get_stats_at.restype = ct.c_int
lib.hnsw_mi355x_get_stats_at.argtypes = [ct.c_void_p, ct.c_int, ct.POINTER(Stats)]
lib.hnsw_mi355x_build_id.restype = ct.c_char_p
lib.hnsw_mi355x_build_id.argtypes = []
lib.hnsw_mi355x_exact_window_stats.restype = ct.c_int
lib.hnsw_mi355x_exact_window_stats.argtypes = [ct.c_void_p, ct.POINTER(ct.c_uint64)]

# ---- (B) hnswdev_* -------------------------------------------------------------------------
lib.hnsw_mi355x_import_nodes.argtypes = [ct.c_void_p, _F, ct.c_int, ct.c_int, _I, ct.c_int]
lib.hnsw_mi355x_import_edges.argtypes = [ct.c_void_p, ct.c_int, _I, _I, ct.c_int]
lib.hnswdev_device_count.restype = ct.c_int
lib.hnswdev_create.restype = ct.c_int
lib.hnswdev_create.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_longlong, ct.POINTER(ct.c_void_p)]
lib.hnswdev_destroy.argtypes = [ct.c_void_p]
lib.hnswdev_reserve.argtypes = [ct.c_void_p, ct.c_longlong]
lib.hnswdev_upload_rows.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, _F]
lib.hnswdev_download_rows.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, _F]
lib.hnswdev_dist_query_batch.argtypes = [ct.c_void_p, _F, ct.c_int, _I, _I, _F]
lib.hnswdev_dist_pair_batch.argtypes = [ct.c_void_p, _I, _I, ct.c_int, _F]
lib.hnswdev_graph_begin.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, _I]
lib.hnswdev_graph_set_layer.argtypes = [ct.c_void_p, ct.c_int, _I, _I, ct.c_int]
lib.hnswdev_graph_commit.argtypes = [ct.c_void_p]
lib.hnswdev_knn_search.argtypes = [ct.c_void_p, _F, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _I, _F, _I]
lib.hnswdev_range_search.restype = ct.c_int
lib.hnswdev_range_search.argtypes = [ct.c_void_p, _F, ct.c_int, ct.c_int, ct.c_float, _I, _I]
lib.hnswdev_range_results.restype = ct.c_int
lib.hnswdev_range_results.argtypes = [ct.c_void_p, _I, _F]
lib.hnswdev_sync.argtypes = [ct.c_void_p]
lib.hnswdev_set_profiling.argtypes = [ct.c_void_p, ct.c_int]
lib.hnswdev_get_stats.argtypes = [ct.c_void_p, ct.POINTER(Stats)]
lib.hnswdev_reset_stats.argtypes = [ct.c_void_p]
lib.hnswdev_last_error.argtypes = [ct.c_char_p, ct.c_int]
lib.hnswdev_ctx_last_error.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int]
lib.hnswdev_set_queries.argtypes = [ct.c_void_p, _F, ct.c_int]
lib.hnswdev_step_buffers.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.POINTER(_I), ct.POINTER(_F)]
lib.hnswdev_step_submit.argtypes = [ct.c_void_p, ct.c_int, ct.c_int]
lib.hnswdev_step_wait.argtypes = [ct.c_void_p, ct.c_int]
lib.hnswdev_test_sqrt_rn.argtypes = [ct.c_int, ct.POINTER(ct.c_double), ct.POINTER(ct.c_double), ct.c_int]

METRICS = {"sq_euclid": 0, "cosine": 1, "ucosine": 2, "sq_euclid_i8": 3}


class Options(ct.Structure):
    """hnsw_mi355x_options, version 1 (include/hnsw_mi355x.h): every knob of the backend."""
    _fields_ = [("struct_size", ct.c_uint32), ("device", ct.c_int32), ("devices", ct.c_int32), ("insert_batch", ct.c_int32),
                ("remove_batch", ct.c_int32), ("host_threads", ct.c_int32), ("search_slots", ct.c_int32), ("device_traversal", ct.c_int32),
                ("diagnostics", ct.c_char_p)]


lib.hnsw_mi355x_default_options.restype = ct.c_int
lib.hnsw_mi355x_default_options.argtypes = [ct.POINTER(Options)]
lib.hnsw_mi355x_set_options.restype = ct.c_int
lib.hnsw_mi355x_set_options.argtypes = [ct.POINTER(Options)]


def default_options() -> Options:
    o = Options()
    if lib.hnsw_mi355x_default_options(ct.byref(o)) != 0:
        raise RuntimeError("hnsw_mi355x_default_options failed")
    return o


def set_options(**knobs) -> None:
    """Sets the pending backend knobs for the next Index (defaults for whatever is not named): device, devices, insert_batch,
    remove_batch, host_threads, search_slots, device_traversal, diagnostics (str or None)."""
    o = default_options()
    for k, v in knobs.items():
        if k not in dict(Options._fields_) or k == "struct_size":
            raise TypeError(f"unknown option {k!r}")
        setattr(o, k, v.encode() if isinstance(v, str) else v)
    if lib.hnsw_mi355x_set_options(ct.byref(o)) != 0:
        raise RuntimeError(last_error())


def host_parallelism() -> int:
    """Hardware threads this process may run on = the default cap of Add's snapshot batches (include/hnsw_mi355x.h)."""
    return int(lib.hnsw_mi355x_host_parallelism())


def last_error() -> str:
    """The library's last error message (what the reference's `_last_error` returns, bindings.py:122-128):
    hnsw_get_last_error_utf8 reports the byte count it needs when asked with no buffer."""
    need = int(lib.hnsw_get_last_error_utf8(None, 0))
    if need < 1:
        return ""
    raw = (ct.c_char * (need + 1))()
    lib.hnsw_get_last_error_utf8(raw, need + 1)
    return bytes(raw).split(b"\0", 1)[0].decode("utf-8", "replace")


def _dev_error() -> str:
    buf = ct.create_string_buffer(1024)
    lib.hnswdev_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def _as_2d_f32(x: npt.ArrayLike, dim_expected=None):
    """One vector or a batch of them as a C-contiguous float32 matrix (n, dim) -- the input contract of the
    reference's wrapper (bindings.py:131-139): a single vector becomes a batch of one, anything else that is not
    two-dimensional or has the wrong row length is a ValueError."""
    m = np.atleast_2d(np.asarray(x, dtype=np.float32))
    if m.ndim > 2 or np.ndim(x) == 0:
        raise ValueError("expected a 2D array of shape (n, dim) or a 1D vector")
    if dim_expected is not None and m.shape[1] != dim_expected:
        raise ValueError(f"expected dim={dim_expected}, got {m.shape[1]}")
    return np.require(m, dtype=np.float32, requirements="C")


class Index:
    """
    Python binding for the native HNSW index -- drop-in for the reference's `Index`
    (bindings/bindings.py:142-597) on the float32 add / knn_query path.

    The native index is created lazily on first insertion; configuration setters must be
    called before it (they mutate the process-global pending parameters that the next
    `hnsw_create` consumes, exactly as in the reference).
    """

    def __init__(self, dim: int, metric="sq_euclid"):
        self.dim = dim
        self.metric = metric
        self._initialized = False
        self._h = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.hnsw_free(self._h)
            self._h = None

    def _initialize(self):
        h = lib.hnsw_create(self.metric.encode("utf-8"))
        if not h:
            raise RuntimeError("hnsw_create failed: " + last_error())
        self._h = h
        self._initialized = True

    @staticmethod
    def _check(status):
        if status < 0:
            raise RuntimeError(last_error())

    # ---- the reference's setters (bindings.py:200-398) ----
    def set_collection_size(self, init_size: int):
        self._check(lib.hnsw_set_collection_size(init_size))

    def set_max_edges(self, max_conn: int):
        self._check(lib.hnsw_set_max_edges(max_conn))

    def set_max_candidates(self, max_candidates: int):
        self._check(lib.hnsw_set_max_candidates(max_candidates))

    def set_remove_max_candidates(self, rem_max_candidates: int):
        self._check(lib.hnsw_set_remove_max_candidates(rem_max_candidates))

    def set_distribution_rate(self, dist_rate: float):
        self._check(lib.hnsw_set_distribution_rate(dist_rate))

    def set_random_seed(self, random_seed: int):
        self._check(lib.hnsw_set_random_seed(random_seed))

    def set_min_nn(self, min_nn: int):
        self._check(lib.hnsw_set_min_nn(min_nn))

    def set_allow_removals(self, allow_removals: bool):
        self._check(lib.hnsw_set_allow_removals(allow_removals))

    # ---- backend knobs (not in the reference) ----
    def set_device(self, device: int):
        self._check(lib.hnsw_mi355x_set_device(device))

    def set_devices(self, n: int):
        """Device contexts of the index: knn_query shards its queries over n GPUs inside this process (include/hnsw_mi355x.h)."""
        self._check(lib.hnsw_mi355x_set_devices(n))

    def stats_at(self, context: int):
        st = Stats()
        if not self._h or lib.hnsw_mi355x_get_stats_at(self._h, context, ct.byref(st)) != 0:
            raise RuntimeError(last_error() or "no such device context")
        return st.as_dict()

    def set_insert_batch(self, max_batch: int):
        """Cap of Add's snapshot batches.  0 (default) = the host's hardware threads: what the reference's Parallel.For can hold
        in flight here; 1 = strictly sequential inserts (HNSWIndex.Add(item)); -W = the same graph built through speculative
        windows of W items; larger caps (65536: rounds 1-4's schedule) are opt-in (see include/hnsw_mi355x.h)."""
        self._check(lib.hnsw_mi355x_set_insert_batch(max_batch))

    def set_insert_batch_live(self, max_batch: int):
        """The insert-batch knob on the index as it stands (pending like the others while nothing has been added)."""
        if not self._h:
            return self.set_insert_batch(max_batch)
        self._check(lib.hnsw_mi355x_index_set_insert_batch(self._h, max_batch))

    @property
    def insert_batch_cap(self) -> int:
        """The cap this index's Add runs under (the resolved default when nothing was set)."""
        return int(lib.hnsw_mi355x_index_insert_batch(self._h)) if self._h else host_parallelism()

    def exact_window_stats(self):
        """Counters of the exact-window Add: rounds, searches run, items inserted alone, items linked through windows."""
        out = (ct.c_uint64 * 4)()
        if not self._h or lib.hnsw_mi355x_exact_window_stats(self._h, out) != 0:
            return {"rounds": 0, "searches": 0, "alone": 0, "linked": 0}
        return {"rounds": int(out[0]), "searches": int(out[1]), "alone": int(out[2]), "linked": int(out[3])}

    def set_remove_batch(self, max_batch: int):
        """1 (default): remove() takes the ids one after the other; B > 1: removals with disjoint neighbourhoods together."""
        self._check(lib.hnsw_mi355x_set_remove_batch(max_batch))

    def set_search_slots(self, slots: int):
        self._check(lib.hnsw_mi355x_set_search_slots(slots))

    def set_host_threads(self, threads: int):
        self._check(lib.hnsw_mi355x_set_host_threads(threads))

    def set_device_traversal(self, enabled: bool):
        """True (default): knn_query traverses on the device; False: host lock-step traversal."""
        self._check(lib.hnsw_mi355x_set_device_traversal(int(enabled)))

    # ---- data path ----
    def add(self, vecs: npt.ArrayLike) -> npt.NDArray[np.int32]:
        """bindings.py:400-441."""
        if not self._initialized:
            self._initialize()
        a = _as_2d_f32(vecs, self.dim)
        n, d = a.shape
        out_ids = np.empty(n, dtype=np.int32)
        rc = lib.hnsw_add(self._h, a.ctypes.data_as(_F), int(n), int(d), out_ids.ctypes.data_as(_I))
        if rc < 0:
            raise RuntimeError(last_error())
        return out_ids[:rc].copy()

    def remove(self, ids: npt.ArrayLike) -> None:
        """bindings.py:443-472."""
        arr = np.asarray(ids, dtype=np.int32).ravel()
        if arr.size == 0:
            return
        result = lib.hnsw_remove(self._h, arr.ctypes.data_as(_I), int(arr.size))
        if result < 0:
            raise RuntimeError(last_error())

    def knn_query(self, queries: npt.ArrayLike, k: int) -> Tuple[npt.NDArray[np.int32], npt.NDArray[np.float32]]:
        """bindings.py:474-521."""
        q = _as_2d_f32(queries, self.dim)
        n = int(q.shape[0])
        ids = np.empty((n, k), dtype=np.int32)
        dists = np.empty((n, k), dtype=np.float32)
        status = lib.hnsw_knn_query(self._h, q.ctypes.data_as(_F), n, self.dim, k, ids.ctypes.data_as(_I),
                                    dists.ctypes.data_as(_F))
        if status < 0:
            raise RuntimeError(last_error())
        return ids, dists  # freshly allocated above (the reference returns copies of equally fresh arrays, bindings.py:521)

    def range_query(self, queries: npt.ArrayLike, radius: float) -> Tuple[List[npt.NDArray[np.int32]], List[npt.NDArray[np.float32]]]:
        """bindings.py:523-597."""
        q = _as_2d_f32(queries, self.dim)
        n = int(q.shape[0])
        ids_pp = (ct.c_void_p * n)()
        dists_pp = (ct.c_void_p * n)()
        counts = (ct.c_int * n)()
        status = lib.hnsw_range_query(self._h, q.ctypes.data_as(_F), n, self.dim, radius, ids_pp, dists_pp, counts)
        if status < 0:
            raise RuntimeError(last_error())
        ids, dists = [], []
        try:
            for i in range(n):
                m = counts[i]
                if m == 0:
                    ids.append(np.empty(0, dtype=np.int32))
                    dists.append(np.empty(0, dtype=np.float32))
                    continue
                ids.append(np.ctypeslib.as_array(ct.cast(ids_pp[i], _I), shape=(m,)).copy())
                dists.append(np.ctypeslib.as_array(ct.cast(dists_pp[i], _F), shape=(m,)).copy())
        finally:
            lib.hnsw_free_results(ids_pp, dists_pp, n)
        return ids, dists

    # ---- measurement aid: query set resident in HBM across calls ----
    def set_resident_queries(self, queries: npt.ArrayLike):
        q = _as_2d_f32(queries, self.dim)
        if lib.hnsw_mi355x_set_queries(self._h, q.ctypes.data_as(_F), int(q.shape[0]), self.dim) < 0:
            raise RuntimeError(last_error())

    def knn_query_resident(self, k: int):
        n = int(lib.hnsw_mi355x_resident_count(self._h)) if self._h else 0  # asked, not remembered: knn_query / range_query replace the set
        if n <= 0:
            raise RuntimeError("no resident query set: call set_resident_queries first (range_query discards it)")
        ids = np.empty((n, k), dtype=np.int32)
        dists = np.empty((n, k), dtype=np.float32)
        if lib.hnsw_mi355x_knn_query_resident(self._h, k, ids.ctypes.data_as(_I), dists.ctypes.data_as(_F)) < 0:
            raise RuntimeError(last_error())
        return ids, dists

    # ---- introspection (parity checks, measurement) ----
    @property
    def count(self) -> int:
        return lib.hnsw_mi355x_count(self._h) if self._h else 0

    @property
    def entry_point(self) -> int:
        return lib.hnsw_mi355x_entry_point(self._h) if self._h else -1

    def max_layer(self, i: int) -> int:
        return lib.hnsw_mi355x_node_max_layer(self._h, int(i))

    @property
    def length(self) -> int:
        """Slots ever allocated; every id is < length (removed slots are reused by later adds)."""
        return lib.hnsw_mi355x_length(self._h) if self._h else 0

    def ids(self):
        """HNSWIndex.Ids(): live ids."""
        out = np.empty(max(1, self.count), dtype=np.int32)
        n = lib.hnsw_mi355x_active_ids(self._h, out.ctypes.data_as(_I), out.size) if self._h else 0
        return out[:n].copy()

    def levels(self):
        out = np.empty(self.length, dtype=np.int32)
        if out.size:
            lib.hnsw_mi355x_export_levels(self._h, out.ctypes.data_as(_I), out.size)
        return out

    def export_edges(self, layer: int, stride: int):
        """(counts[n], edges[n, stride]) of one layer; counts == -1 where the node is absent."""
        n = self.length
        counts = np.empty(n, dtype=np.int32)
        edges = np.zeros((n, stride), dtype=np.int32)
        if n and lib.hnsw_mi355x_export_edges(self._h, int(layer), counts.ctypes.data_as(_I), edges.ctypes.data_as(_I),
                                              int(stride), n) < 0:
            raise RuntimeError("export_edges: stride too small")
        return counts, edges

    def edges(self, i: int, layer: int):
        buf = np.empty(4096, dtype=np.int32)
        n = lib.hnsw_mi355x_get_out_edges(self._h, int(i), int(layer), buf.ctypes.data_as(_I), buf.size)
        if n < 0:
            raise IndexError((i, layer))
        return buf[:n].copy()

    # ---- HNSWIndex.Serialize / Deserialize (src/HNSWIndex/HNSWIndex.cs:210-229) ----
    def serialize(self, path) -> None:
        """Write the reference's protobuf-net snapshot of this index to `path`."""
        if not self._initialized:
            self._initialize()
        self._check(lib.hnsw_mi355x_serialize(self._h, os.fsencode(path)))

    @classmethod
    def deserialize(cls, path, metric="sq_euclid") -> "Index":
        """Reconstruct an index from a snapshot written by `serialize` or by the reference."""
        h = lib.hnsw_mi355x_deserialize(metric.encode("utf-8"), os.fsencode(path))
        if not h:
            raise RuntimeError("deserialize failed: " + last_error())
        ix = cls(0, metric)
        ix._h = h
        ix._initialized = True
        ix.dim = int(lib.hnsw_mi355x_dim(h))
        return ix

    def import_graph(self, rows, levels, entry_point: int, layers):
        """Load a graph built elsewhere into this (still empty) index: rows [n, dim], levels [n], the entry point and per
        layer a (counts[n], edges[n, stride]) pair as `export_edges` returns them.  Afterwards the index answers and
        grows exactly like the one the graph came from."""
        if not self._initialized:
            self._initialize()
        a = _as_2d_f32(rows, self.dim)
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        self._check(lib.hnsw_mi355x_import_nodes(self._h, a.ctypes.data_as(_F), int(a.shape[0]), int(a.shape[1]),
                                                 lv.ctypes.data_as(_I), int(entry_point)))
        for layer, (counts, edges) in enumerate(layers):
            c = np.ascontiguousarray(counts, dtype=np.int32)
            e = np.ascontiguousarray(edges, dtype=np.int32)
            self._check(lib.hnsw_mi355x_import_edges(self._h, layer, c.ctypes.data_as(_I), e.ctypes.data_as(_I), int(e.shape[1])))

    def graph_hash(self) -> int:
        return int(lib.hnsw_mi355x_graph_hash(self._h))

    def set_profiling(self, on: bool):
        if not self._initialized:
            self._initialize()
        lib.hnsw_mi355x_set_profiling(self._h, int(on))

    def stats(self) -> dict:
        s = Stats()
        if self._h:
            lib.hnsw_mi355x_get_stats(self._h, ct.byref(s))
        return s.as_dict()

    def reset_stats(self):
        if self._h:
            lib.hnsw_mi355x_reset_stats(self._h)


class DeviceBackend:
    """The inner boundary (hnswdev_*): HBM-resident row matrix + batched distance calls."""

    def __init__(self, dim: int, metric="sq_euclid", capacity=1024, device=0):
        self.dim, self.metric = dim, metric
        ctx = ct.c_void_p()
        if lib.hnswdev_create(device, dim, METRICS[metric], capacity, ct.byref(ctx)) != 0:
            raise RuntimeError("hnswdev_create failed: " + _dev_error())
        self._ctx = ctx

    def __del__(self):
        if getattr(self, "_ctx", None):
            lib.hnswdev_destroy(self._ctx)
            self._ctx = None

    def last_error(self) -> str:
        """This context's own last error (hnswdev_ctx_last_error)."""
        buf = ct.create_string_buffer(4096)
        lib.hnswdev_ctx_last_error(self._ctx, buf, len(buf))
        return buf.value.decode("utf-8", "replace")

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.last_error())

    def reserve(self, capacity: int):
        self._check(lib.hnswdev_reserve(self._ctx, capacity))

    def upload_rows(self, first_id: int, rows):
        a = _as_2d_f32(rows, self.dim)
        self._check(lib.hnswdev_upload_rows(self._ctx, first_id, a.shape[0], a.ctypes.data_as(_F)))

    def download_rows(self, first_id: int, n: int):
        out = np.empty((n, self.dim), dtype=np.float32)
        self._check(lib.hnswdev_download_rows(self._ctx, first_id, n, out.ctypes.data_as(_F)))
        return out

    def set_queries(self, queries):
        """Uploads the query set of a batch of searches once; records name queries by row index."""
        q = _as_2d_f32(queries, self.dim)
        self._check(lib.hnswdev_set_queries(self._ctx, q.ctypes.data_as(_F), q.shape[0]))

    def dist_query_batch(self, queries, cand_offsets, cand_ids):
        """queries=None: measure against the resident set (set_queries), nothing is re-uploaded."""
        off = np.ascontiguousarray(cand_offsets, dtype=np.int32)
        ids = np.ascontiguousarray(cand_ids, dtype=np.int32)
        out = np.empty(ids.size, dtype=np.float32)
        if queries is None:
            qp, nq = None, off.size - 1
        else:
            q = _as_2d_f32(queries, self.dim)
            assert off.size == q.shape[0] + 1
            qp, nq = q.ctypes.data_as(_F), q.shape[0]
        self._check(lib.hnswdev_dist_query_batch(self._ctx, qp, nq, off.ctypes.data_as(_I), ids.ctypes.data_as(_I),
                                                 out.ctypes.data_as(_F)))
        return out

    def step_buffers(self, which: int, nslots: int, stride: int):
        """The context's pinned step-buffer set `which` (0 | 1) as numpy views: records
        [nslots, stride + 2] = (cnt, qidx, ids...) and distances [nslots, stride]."""
        rec, dist = _I(), _F()
        self._check(lib.hnswdev_step_buffers(self._ctx, which, nslots, stride, ct.byref(rec), ct.byref(dist)))
        r = np.ctypeslib.as_array(rec, shape=(nslots, stride + 2))
        d = np.ctypeslib.as_array(dist, shape=(nslots, stride))
        return r, d

    def step_submit(self, which: int, nslots_used: int):
        self._check(lib.hnswdev_step_submit(self._ctx, which, nslots_used))

    def step_wait(self, which: int):
        self._check(lib.hnswdev_step_wait(self._ctx, which))

    def dist_pair_batch(self, a_ids, b_ids):
        a = np.ascontiguousarray(a_ids, dtype=np.int32)
        b = np.ascontiguousarray(b_ids, dtype=np.int32)
        assert a.size == b.size
        out = np.empty(a.size, dtype=np.float32)
        self._check(lib.hnswdev_dist_pair_batch(self._ctx, a.ctypes.data_as(_I), b.ctypes.data_as(_I), a.size,
                                                out.ctypes.data_as(_F)))
        return out

    def set_graph(self, levels, layers, max_edges: int):
        """layers: per layer a (counts[n], edges[n, stride]) pair, EdgeList order (as Index.export_edges)."""
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        self._check(lib.hnswdev_graph_begin(self._ctx, lv.size, int(max_edges), lv.ctypes.data_as(_I)))
        for layer, (counts, edges) in enumerate(layers):
            c = np.ascontiguousarray(counts, dtype=np.int32)
            e = np.ascontiguousarray(edges, dtype=np.int32)
            self._check(lib.hnswdev_graph_set_layer(self._ctx, layer, c.ctypes.data_as(_I), e.ctypes.data_as(_I), e.shape[1]))
        self._check(lib.hnswdev_graph_commit(self._ctx))

    def knn_search(self, queries, entry_point: int, k_beam: int, k_out: int):
        q = _as_2d_f32(queries, self.dim)
        n = q.shape[0]
        ids = np.empty((n, k_out), dtype=np.int32)
        d = np.empty((n, k_out), dtype=np.float32)
        flags = np.empty(n, dtype=np.int32)
        self._check(lib.hnswdev_knn_search(self._ctx, q.ctypes.data_as(_F), n, int(entry_point), int(k_beam), int(k_out),
                                           ids.ctypes.data_as(_I), d.ctypes.data_as(_F), flags.ctypes.data_as(_I)))
        return ids, d, flags

    def range_search(self, queries, entry_point: int, radius: float):
        """Per query the ids / distances within `radius`, ascending by distance; flags[i] = 1: handed back (empty)."""
        q = _as_2d_f32(queries, self.dim)
        n = q.shape[0]
        counts = np.zeros(n, dtype=np.int32)
        flags = np.zeros(n, dtype=np.int32)
        self._check(lib.hnswdev_range_search(self._ctx, q.ctypes.data_as(_F), n, int(entry_point), float(radius),
                                             counts.ctypes.data_as(_I), flags.ctypes.data_as(_I)))
        total = int(counts.sum())
        ids = np.empty(max(total, 1), dtype=np.int32)
        d = np.empty(max(total, 1), dtype=np.float32)
        self._check(lib.hnswdev_range_results(self._ctx, ids.ctypes.data_as(_I), d.ctypes.data_as(_F)))
        cuts = np.cumsum(counts)[:-1]
        return np.split(ids[:total], cuts), np.split(d[:total], cuts), flags

    def set_profiling(self, on: bool):
        self._check(lib.hnswdev_set_profiling(self._ctx, int(on)))

    def stats(self) -> dict:
        s = Stats()
        self._check(lib.hnswdev_get_stats(self._ctx, ct.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._check(lib.hnswdev_reset_stats(self._ctx))


def device_sqrt_rn(x, device=0):
    """Test hook: the device's correctly rounded double sqrt (cosine epilogue)."""
    a = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(a)
    rc = lib.hnswdev_test_sqrt_rn(device, a.ctypes.data_as(ct.POINTER(ct.c_double)),
                                  out.ctypes.data_as(ct.POINTER(ct.c_double)), a.size)
    if rc != 0:
        raise RuntimeError(_dev_error())
    return out
