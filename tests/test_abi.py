"""CPU tier: the C-ABI library loads, exports every symbol include/*.h declares, and keeps
the reference's error conventions (HNSWIndexExports.cs) without touching a GPU."""
import ctypes as ct
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def net():
    import hnswindex
    return hnswindex.net_amd


def _declared_symbols():
    text = (ROOT / "include" / "hnsw_mi355x.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hnsw(?:dev|_mi355x)?_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_16_reference_exports():
    want = {"hnsw_get_last_error_utf8", "hnsw_create", "hnsw_free", "hnsw_add", "hnsw_remove", "hnsw_knn_query",
            "hnsw_range_query", "hnsw_free_results", "hnsw_set_collection_size", "hnsw_set_max_edges",
            "hnsw_set_max_candidates", "hnsw_set_remove_max_candidates", "hnsw_set_distribution_rate",
            "hnsw_set_random_seed", "hnsw_set_min_nn", "hnsw_set_allow_removals"}
    assert want <= set(_declared_symbols())


def test_library_exports_every_declared_symbol(net):
    syms = _declared_symbols()
    assert len(syms) >= 16 + 12
    for s in syms:
        assert hasattr(net.lib, s), s


def test_every_declared_entry_point_is_named_in_integration_md():
    """INTEGRATION.md shows the reference-side binding for the calls a maintainer binds first and lists every other entry point
    of the header by group (section 4): no export may be undocumented."""
    text = (ROOT / "INTEGRATION.md").read_text()
    missing = [s for s in _declared_symbols() if s not in text]
    assert not missing, missing


def test_library_is_at_the_reference_loader_path(net):
    # /root/reference/bindings/bindings.py:27-41
    assert net.LIB_PATH.parts[-4:] == ("artifacts", "native", "linux-x64", "HNSWIndex.Native.so")


def test_loaded_library_was_built_from_this_tree(net):
    """A stale binary (it is git-ignored and travels with the tree) must not pass for the sources beside it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hnsw_build", ROOT / "hnswindex.net_amd" / "build.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    want = b.source_id()
    assert len(want) == 64
    got = net.lib.hnsw_mi355x_build_id().decode()
    if net.LIB_PATH == b.LIB:          # (HNSW_MI355X_LIB may point the bindings at a diagnostic variant)
        assert got == want, "the loaded library was compiled from other sources than the tree's: run __graft_entry__.build()"
        assert b.embedded_id() == want and not b.needs_build()
    else:
        assert got.endswith("+variant") or got == want


def _has_gpu(net):
    return net.lib.hnswdev_device_count() > 0


def test_null_handle_conventions(net):
    lib = net.lib
    v = np.zeros((2, 4), dtype=np.float32)
    ids = np.zeros(2, dtype=np.int32)
    F, I = ct.POINTER(ct.c_float), ct.POINTER(ct.c_int)
    assert lib.hnsw_add(None, v.ctypes.data_as(F), 2, 4, ids.ctypes.data_as(I)) == 0      # Exports.cs:78
    assert lib.hnsw_remove(None, ids.ctypes.data_as(I), 2) == 0                            # :105
    d = np.zeros((2, 3), dtype=np.float32)
    o = np.zeros((2, 3), dtype=np.int32)
    assert lib.hnsw_knn_query(None, v.ctypes.data_as(F), 2, 4, 3, o.ctypes.data_as(I), d.ctypes.data_as(F)) == 0  # :122
    lib.hnsw_free(None)                                                                     # :69
    lib.hnsw_free_results(None, None, 3)                                                    # :202


def test_setters_return_zero_and_unknown_metric_fails_with_message(net):
    lib = net.lib
    for name, val in (("hnsw_set_collection_size", 128), ("hnsw_set_max_edges", 8), ("hnsw_set_max_candidates", 50),
                      ("hnsw_set_remove_max_candidates", 50), ("hnsw_set_random_seed", 7), ("hnsw_set_min_nn", 3)):
        assert getattr(lib, name)(val) == 0
    assert lib.hnsw_set_distribution_rate(0.5) == 0
    assert lib.hnsw_set_allow_removals(False) == 0
    assert not lib.hnsw_create(b"manhattan")                      # Exports.cs:58-59,64
    msg = net.last_error()
    assert "Unsupported distance metric" in msg and "manhattan" in msg
    # last-error getter: returns needed byte count, truncates and NUL-terminates (:27-39)
    need = lib.hnsw_get_last_error_utf8(None, 0)
    assert need == len(msg.encode())
    buf = ct.create_string_buffer(8)
    assert lib.hnsw_get_last_error_utf8(buf, 8) == need
    assert buf.value == msg.encode()[:7]
    # a failed create must NOT consume the pending parameters; reset them for other tests
    for name, val in (("hnsw_set_collection_size", 65536), ("hnsw_set_max_edges", 16), ("hnsw_set_max_candidates", 100),
                      ("hnsw_set_remove_max_candidates", 100), ("hnsw_set_random_seed", 31337), ("hnsw_set_min_nn", 5)):
        getattr(lib, name)(val)
    lib.hnsw_set_distribution_rate(float(1 / np.log(16)))
    lib.hnsw_set_allow_removals(True)


def test_no_gpu_means_loud_failure_not_cpu_fallback(net):
    if _has_gpu(net):
        pytest.skip("a HIP device is present")
    assert not net.lib.hnsw_create(b"sq_euclid")
    assert "no HIP device" in net.last_error() and "no CPU fallback" in net.last_error()
    with pytest.raises(RuntimeError, match="no HIP device"):
        net.Index(8).add(np.zeros((1, 8), dtype=np.float32))
    with pytest.raises(RuntimeError, match="no HIP device"):
        net.DeviceBackend(8)


def test_product_never_references_the_oracle():
    # the oracle is test infrastructure: nothing under the package may import, link, load
    # or call it (comments may mention it)
    pat = re.compile(r"import\s+oracle|from\s+oracle|oracle/|oracle\\.|libhnsw_oracle|\borc_[a-z]")
    files = [p for p in (ROOT / "hnswindex.net_amd").rglob("*") if p.is_file() and p.suffix in (".py", ".cpp", ".hip", ".h")]
    files.append(ROOT / "hnswindex" / "__init__.py")
    assert len(files) > 8
    for p in files:
        assert not pat.search(p.read_text()), p


def test_options_struct_is_the_documented_surface(net, monkeypatch):
    """hnsw_mi355x_options (version 1): defaults, range checks, a shorter (older) struct, and the one diagnostics mechanism --
    the options' string while set, else the environment variable HNSW_MI355X_DIAG, read on every use (csrc/diag.h)."""
    import ctypes as ct
    L = net.lib
    o = net.default_options()
    assert o.struct_size == ct.sizeof(net.Options) and (o.device, o.devices, o.insert_batch, o.remove_batch) == (0, 1, 0, 1)
    assert (o.host_threads, o.search_slots, o.device_traversal) == (0, 16384, 1) and o.diagnostics is None
    L.hnswhost_test_diag.restype = ct.c_int
    L.hnswhost_test_diag.argtypes = [ct.c_char_p, ct.c_int]
    monkeypatch.delenv("HNSW_MI355X_DIAG", raising=False)
    assert L.hnswhost_test_diag(b"lat", 1) == 1
    monkeypatch.setenv("HNSW_MI355X_DIAG", "novis=0, lat=2,vis_hash_cap=64")
    assert (L.hnswhost_test_diag(b"lat", 1), L.hnswhost_test_diag(b"novis", 2), L.hnswhost_test_diag(b"vis_hash_cap", 0)) == (2, 0, 64)
    assert L.hnswhost_test_diag(b"vis_hash", -1) == -1 and L.hnswhost_test_diag(b"la", 7) == 7      # a prefix of a name is not the name
    net.set_options(insert_batch=256, devices=2, diagnostics="lat=0")                                 # the struct's string wins while set ...
    assert L.hnswhost_test_diag(b"lat", 1) == 0 and L.hnswhost_test_diag(b"novis", 2) == 2
    net.set_options()                                                                                 # ... NULL: back to the environment
    assert L.hnswhost_test_diag(b"lat", 1) == 2
    for bad in (dict(insert_batch=-1), dict(remove_batch=0), dict(devices=65), dict(host_threads=-3)):
        with pytest.raises(RuntimeError):
            net.set_options(**bad)
    with pytest.raises(TypeError):
        net.set_options(no_such_knob=1)
    short = net.default_options(); short.struct_size = 16; short.insert_batch = 64          # an older caller: four fields only
    assert L.hnsw_mi355x_set_options(ct.byref(short)) == 0
    zero = net.Options()                                                                     # struct_size not set
    assert L.hnsw_mi355x_set_options(ct.byref(zero)) == -1
    net.set_options()                                                                        # leave the pending block at its defaults


def test_no_environment_switch_but_the_documented_one():
    """The 27 HNSW_MI355X_* environment switches of rounds 1-4 are one: HNSW_MI355X_DIAG.  (HNSW_MI355X_LIB / _NO_TORCH belong to
    the Python loader, _EXTRA_FLAGS / _REBUILD to the build recipe.)"""
    import re
    root = Path(__file__).resolve().parent.parent
    names = set()
    for f in [f for f in (root / "hnswindex.net_amd" / "csrc").glob("*") if f.is_file()] + list((root / "include").glob("*.h")):
        names |= set(re.findall(r'getenv\("(HNSW_MI355X_[A-Z_0-9]+)"\)', f.read_text()))
    assert names == {"HNSW_MI355X_DIAG"}, names

