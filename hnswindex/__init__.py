"""
Import name of the reference's Python package (`from hnswindex import Index`,
/root/reference/bindings/__tests__/*.py).  The implementation lives in the directory
`hnswindex.net_amd/` (not importable under that spelling); it is loaded here as the module
`hnswindex_net_amd`.
"""
import importlib.util
import sys
from pathlib import Path

_root = Path(__file__).resolve().parent.parent / "hnswindex.net_amd"
if "hnswindex_net_amd" not in sys.modules:
    _spec = importlib.util.spec_from_file_location("hnswindex_net_amd", _root / "__init__.py",
                                                   submodule_search_locations=[str(_root)])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["hnswindex_net_amd"] = _mod
    _spec.loader.exec_module(_mod)
net_amd = sys.modules["hnswindex_net_amd"]
Index = net_amd.Index
DeviceBackend = net_amd.DeviceBackend
__all__ = ["Index", "DeviceBackend", "net_amd"]
