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
# device_backend.hip: host side + the small kernels; traverse_<metric>_<kernel>.hip: the instantiations of the two
# big traversal kernel templates (device code in device_kernels.h) -- separate units so that they
# compile in parallel (one unit took two minutes).
SOURCES = ["device_backend.hip", *[f"traverse_{m}_{k}{v}.hip" for m in ("sq", "cos", "ucos", "i8") for k in ("insert", "search") for v in ("", "_lat")],
           *[f"traverse_{m}_search_lean.hip" for m in ("sq", "cos", "ucos", "i8")],
           "search_engine.cpp", "hnsw_index.cpp", "exports.cpp"]
# -ffp-contract=off: the kernels fuse a*b+c only where __builtin_fmaf is written -- the
# reference's AVX path fuses in sq_euclid (Fma.MultiplyAdd) and nowhere else.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall"]
OBJ = PKG / "artifacts" / "obj"


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found")


BUILD_ID_TAG = b"HNSW_MI355X_BUILD_ID="


def source_id(extra=()) -> str:
    """sha256 over everything the library is compiled from: csrc/* and include/*, by name and content, plus the
    compiler flags (and the extra flags of a diagnostic build).  The same string is compiled into the library
    (hnsw_mi355x_build_id()), so a binary says which sources it came from -- file times say nothing once a tree has
    been copied."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(list(CSRC.glob("*")) + list((PKG.parent / "include").glob("*.h")), key=lambda f: f.name)
    for f in files:
        if f.is_file():
            h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    h.update(" ".join(list(FLAGS) + sorted(extra)).encode())
    return h.hexdigest()


def embedded_id(lib: Path = None) -> str:
    """The build id inside a built library (read from the file: nothing is loaded), '' when it carries none."""
    lib = lib or LIB
    try:
        blob = lib.read_bytes()
    except OSError:
        return ""
    i = blob.find(BUILD_ID_TAG)
    if i < 0:
        return ""
    j = blob.find(b"\0", i)   # the WHOLE string up to its terminator: "<sha256>+variant" is not "<sha256>"
    return blob[i + len(BUILD_ID_TAG):j if j >= 0 else i + len(BUILD_ID_TAG) + 64].decode("ascii", "replace")


def needs_build() -> bool:
    return not LIB.exists() or embedded_id() != source_id()


def build(force: bool = False, verbose: bool = False, out: Path = None) -> Path:
    """out: build a diagnostic variant (HNSW_MI355X_EXTRA_FLAGS) beside the product library instead of replacing it;
    load it with HNSW_MI355X_LIB=<out>."""
    global LIB, OBJ
    if out is not None:
        LIB, OBJ = Path(out).resolve(), PKG / "artifacts" / ("obj_" + Path(out).stem)
    elif not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    LIB.parent.mkdir(parents=True, exist_ok=True)
    OBJ.mkdir(parents=True, exist_ok=True)
    extra = os.environ.get("HNSW_MI355X_EXTRA_FLAGS", "").split()  # kernel experiments (-D...)
    sources = list(SOURCES)
    if "-DHNSW_SINGLE_TU" in extra:  # diagnostic builds: every kernel in one unit
        extra = sorted(set(extra) | {"-DHNSW_SINGLE_TU"})
        sources = [s for s in sources if not s.startswith("traverse_")]

    sid = source_id() if out is None and not extra else source_id(extra) + "+variant"

    def compile_one(src):
        obj = OBJ / (Path(src).stem + ".o")
        ident = [f'-DHNSW_MI355X_BUILD_ID_STR="{sid}"'] if src == "exports.cpp" else []  # one unit carries the id
        cmd = [hipcc(), *FLAGS, *extra, *ident, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout + r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, sources))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *[str(o) for o in objs], "-o", str(LIB) + ".tmp", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    os.replace(str(LIB) + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, out=Path(sys.argv[1]) if len(sys.argv) > 1 else None))
