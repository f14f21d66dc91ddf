"""Builds libmimir_gpu_shuffle.so (gfx950 HIP kernels + C ABI) in-tree.

    python -m mimir_amd.build [--force] [-v]

hipcc cross-compiles for gfx950 without a GPU, so this runs anywhere the ROCm
toolchain is installed.  Every .hip file under csrc/ is one translation unit,
compiled to an object under mimir_amd/lib/obj/ (several at a time) and rebuilt
only when it or a header it includes has changed (hipcc's own dependency
files); the objects are linked into mimir_amd/lib/, which `mimir_amd.capi` loads.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libmimir_gpu_shuffle.so")
SOURCES = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
ARCH = "gfx950"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _obj(src):
    return os.path.join(OBJDIR, os.path.splitext(os.path.basename(src))[0] + ".o")


def _deps(obj):
    """Files named by the compiler's dependency file of an object, or None."""
    try:
        with open(obj[:-2] + ".d") as f:
            text = f.read().replace("\\\n", " ")
    except OSError:
        return None
    return [t for t in text.split(":", 1)[1].split() if t] if ":" in text else None


def _obj_stale(src, flags_tag):
    obj = _obj(src)
    if not os.path.exists(obj):
        return True
    try:
        with open(obj[:-2] + ".flags") as f:
            if f.read() != flags_tag:
                return True
    except OSError:
        return True
    deps = _deps(obj)
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps + [src])


def stale(extra_flags=()):
    tag = " ".join(extra_flags)
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(_obj_stale(s, tag) or os.path.getmtime(_obj(s)) > t for s in SOURCES)


def build_library(force=False, verbose=False, extra_flags=(), jobs=None):
    tag = " ".join(extra_flags)
    if not force and not stale(extra_flags):
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    common = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
              "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC, *extra_flags]

    def compile_one(src):
        obj = _obj(src)
        cmd = common + ["-c", "-MD", "-MF", obj[:-2] + ".d", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(obj[:-2] + ".flags", "w") as f:
            f.write(tag)

    todo = [s for s in SOURCES if force or _obj_stale(s, tag)]
    jobs = jobs or int(os.environ.get("MAX_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=max(1, min(jobs, len(todo) or 1))) as ex:
        list(ex.map(compile_one, todo))
    link = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB,
            *[_obj(s) for s in SOURCES]]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a not in ("--force", "-v")]
    print(build_library(force="--force" in sys.argv, verbose="-v" in sys.argv or True,
                        extra_flags=flags))
