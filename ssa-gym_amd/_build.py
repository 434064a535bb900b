"""Builds libssa_hip.so (the C-ABI library of include/ssa_hip.h) with hipcc for gfx950.

The library is built IN-TREE next to this file so that it travels with the repo snapshot
to the GPU box.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ssa_kernels.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "ssa_math.hpp"), os.path.join(HERE, "csrc", "ssa_conics.hpp"),
        os.path.join(os.path.dirname(HERE), "include", "ssa_hip.h")]
LIB = os.path.join(HERE, "libssa_hip.so")
ARCH = "gfx950"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build_library(force=False, verbose=False, extra_flags=()):
    """hipcc --offload-arch=gfx950 -shared -fPIC -> ssa-gym_amd/libssa_hip.so"""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libssa_hip.so")
    # -disable-machine-licm: the step kernel is one long straight-line body inside a tile loop (plus Newton /
    # ladder loops); MachineLICM hoists its literals, argument scalars and LDS addresses out of those loops
    # and the register allocator then spills them (16 VGPR + 48 SGPR spills, 220 B scratch per lane with it;
    # none without at 4 waves/SIMD) -- see DESIGN.md section 6.  -amdgpu-kernarg-preload-count: the step kernel's leading pointer
    # arguments arrive in SGPRs at wavefront launch (no scalar-memory round trip before the tile loads)
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-shared",
           "-ffp-contract=fast", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-kernarg-preload-count=8",
           *extra_flags, "-o", LIB + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
