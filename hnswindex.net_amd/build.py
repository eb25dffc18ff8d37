"""Build recipe for the gfx950 shared library (run by __graft_entry__.build())."""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
# Same relative location and file name the reference's ctypes loader expects
# (/root/reference/bindings/bindings.py:27-41), so its bindings.py binds unchanged.
LIB = PKG / "artifacts" / "native" / "linux-x64" / "HNSWIndex.Native.so"
SOURCES = ["device_backend.hip", "search_engine.cpp", "hnsw_index.cpp", "exports.cpp"]
# -ffp-contract=off: the kernels fuse a*b+c only where __builtin_fmaf is written -- the
# reference's AVX path fuses in sq_euclid (Fma.MultiplyAdd) and nowhere else.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall",
         "-shared"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob("*")) + [PKG.parent / "include" / "hnsw_mi355x.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    LIB.parent.mkdir(parents=True, exist_ok=True)
    extra = os.environ.get("HNSW_MI355X_EXTRA_FLAGS", "").split()  # kernel experiments (-D...)
    cmd = [hipcc(), *FLAGS, *extra, *[str(CSRC / s) for s in SOURCES], "-o", str(LIB), "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
