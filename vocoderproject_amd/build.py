"""Builds libvp_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m vocoderproject_amd.build [--force]

-ffp-contract=off is part of the numerics contract: the kernels reproduce the reference's IEEE
double arithmetic operation by operation (no FMA contraction), see csrc/vp_kernels.hip.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libvp_amd.so")
SOURCES = ["vp_kernels.hip", "vp_capi.hip"]
DEPS = SOURCES + ["vp_common.h", "vp_kernels.h"]
ARCH = "gfx950"


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in DEPS] + [os.path.join(ROOT, "include", "vp_amd.h")]
    return any(os.path.getmtime(p) > t for p in deps)


LIB_STAMPS = os.path.join(HERE, "libvp_amd_stamps.so")
LIB_POISON = os.path.join(HERE, "libvp_amd_poison.so")


def build(force=False, verbose=False, stamps=False, poison=False):
    """stamps=True builds the DIAGNOSTIC library (in-kernel phase timers, -DVP_STAMPS) next to the
    product one; it is only ever loaded through VP_AMD_LIB by tools/phase_stamps.py.
    poison=True builds the other diagnostic library (-DVP_POISON_LDS: every kernel first fills its LDS with
    NaNs, so a read of LDS the launch has not written fails the parity tests deterministically instead of
    depending on what the previous kernel left on that CU); tests/test_gpu_parity.py runs the suite on it."""
    lib = LIB_STAMPS if stamps else LIB_POISON if poison else LIB
    if not stamps and not poison and not force and not needs_build():
        return LIB
    cmd = [hipcc(), "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
           f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    if stamps:
        cmd.append("-DVP_STAMPS")
    if poison:
        cmd.append("-DVP_POISON_LDS")
    cmd += [os.path.join(CSRC, f) for f in SOURCES]
    cmd += ["-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, stamps="--stamps" in sys.argv, poison="--poison" in sys.argv))
