"""Builds libmimir_gpu_shuffle.so (gfx950 HIP kernels + C ABI) in-tree.

    python -m mimir_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs anywhere the ROCm
toolchain is installed.  The output lands in mimir_amd/lib/ next to the
sources and is what `mimir_amd.capi` loads.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmimir_gpu_shuffle.so")
SOURCES = [os.path.join(CSRC, "mgs_api.hip"), os.path.join(CSRC, "mgs_stager.hip")]
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
    os.path.join(ROOT, "include", "mimir_gpu_shuffle.h")]
ARCH = "gfx950"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False, extra_flags=()):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
           "-shared", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           *extra_flags, "-o", LIB, *SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a not in ("--force", "-v")]
    print(build_library(force="--force" in sys.argv, verbose=True,
                        extra_flags=flags))
