"""Builds libvp_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m vocoderproject_amd.build [--force]

-ffp-contract=off is part of the numerics contract: the kernels reproduce the reference's IEEE
double arithmetic operation by operation (no FMA contraction), see csrc/vp_kernels.hip.
-fdenormal-fp-math=preserve-sign is the other part: the reference's processBlock() runs under
juce::ScopedNoDenormals (PluginProcessor.cpp:205: MXCSR FTZ|DAZ for float AND double SSE arithmetic), so every
kernel is built with the wave's denormal modes (float and double) set to "flush inputs and results": an f32-denormal
input sample is widened to 0.0, a result in the f32-denormal range leaves as (signed) 0.0f, like on the CPU.
"""
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libvp_amd.so")
SOURCES = ["vp_kernels.hip", "vp_voc2.hip", "vp_stft.hip", "vp_capi.hip"]
PARTS = ["vp_fft.inc", "vp_filters.inc", "vp_vocoder_wg.inc", "vp_pitch.inc", "vp_pitch_ws.inc", "vp_pitch_ws_body.inc"]      # included by vp_kernels.hip
DEPS = SOURCES + PARTS + ["vp_common.h", "vp_kernels.h", "vp_voc2.h", "vp_stft.h", "vp_fft32.inc"]
ARCH = "gfx950"
NUM_TUS = 9          # groups of kernels in vp_kernels.hip (VP_TU)


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


_HIPCC_VERSION = None


def hipcc_version():
    """`hipcc --version`, part of the object-cache key: an upgraded compiler at the same path must not be served stale objects."""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            _HIPCC_VERSION = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
        except (OSError, subprocess.SubprocessError):
            _HIPCC_VERSION = "unknown"
    return _HIPCC_VERSION


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in DEPS] + [os.path.join(ROOT, "include", "vp_amd.h")]
    return any(os.path.getmtime(p) > t for p in deps)


LIB_STAMPS = os.path.join(HERE, "libvp_amd_stamps.so")
LIB_POISON = os.path.join(HERE, "libvp_amd_poison.so")


CACHE_CAP_BYTES = 600 << 20      # (one full set of product, stamps, poison and side-build objects is ~200 MB: room for three generations)


def prune_cache(cap=CACHE_CAP_BYTES):
    """Least-recently-used objects go until the object cache is below `cap` (it once grew to 1.2 GB of experiment objects)."""
    cache = os.path.join(HERE, ".build_cache")
    try:
        ents = []
        now = time.time()
        for f in os.listdir(cache):
            p = os.path.join(cache, f)
            try:
                if ".part" in f:
                    # a half-written object of a build that is gone (a live one is minutes old at most): not part of the cache
                    if now - os.path.getmtime(p) > 3600:
                        os.remove(p)
                    continue
                ents.append((os.path.getmtime(p), os.path.getsize(p), p))
            except OSError:
                pass                                               # (another build pruned or replaced it meanwhile)
    except OSError:
        return
    total = sum(e[1] for e in ents)
    for _, size, path in sorted(ents):
        if total <= cap:
            break
        try:
            os.remove(path)
            total -= size
        except OSError:
            pass


def build(force=False, verbose=False, stamps=False, poison=False):
    """stamps=True builds the DIAGNOSTIC library (in-kernel phase timers, -DVP_STAMPS) next to the
    product one; it is only ever loaded through VP_AMD_LIB by tools/phase_stamps.py.
    poison=True builds the other diagnostic library (-DVP_POISON_LDS: every kernel first fills its LDS with
    NaNs, so a read of LDS the launch has not written fails the parity tests deterministically instead of
    depending on what the previous kernel left on that CU); tests/test_gpu_parity.py runs the suite on it."""
    # (VP_LIB_OUT: experiment builds of the PRODUCT library, tools/ab.sh; the diagnostic twins keep their own paths)
    lib = LIB_STAMPS if stamps else LIB_POISON if poison else (os.environ.get("VP_LIB_OUT") or LIB)
    if not stamps and not poison and not force and lib == LIB and not needs_build():
        return LIB
    # vp_kernels.hip can be compiled as NUM_TUS translation units side by side (each keeps one group of kernels, -DVP_TU=k),
    # vp_capi.hip is one more; then one link.  (One translation unit takes 90 s, the groups 40 s.)
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    common = [hipcc(), "-std=c++17", "-O3", "-ffp-contract=off", "-fdenormal-fp-math=preserve-sign", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
              "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    common += os.environ.get("VP_EXTRA_HIPCC_FLAGS", "").split()     # experiments (tools/ab.sh): e.g. scheduler options
    if stamps:
        common.append("-DVP_STAMPS")
    if poison:
        common.append("-DVP_POISON_LDS")
    # The PRODUCT library keeps vp_kernels.hip in one translation unit: measured on the same box, the same kernels run
    # 1.3 % slower out of per-group code objects (instruction placement; tools/ab.sh).  The diagnostic twins take the
    # fast build.
    # (VP_SPLIT_BUILD=1: the development loop -- the product library from per-group objects too, 40 s instead of 3 min)
    groups = [0] if not (stamps or poison or os.environ.get("VP_SPLIT_BUILD") == "1") else list(range(1, NUM_TUS + 1))
    with tempfile.TemporaryDirectory(prefix="vp_build_") as tmp:
        jobs = [(os.path.join(CSRC, "vp_kernels.hip"), os.path.join(tmp, f"k{k}.o"), [f"-DVP_TU={k}"]) for k in groups]
        jobs.append((os.path.join(CSRC, "vp_voc2.hip"), os.path.join(tmp, "voc2.o"), []))     # the batched vocoder pipeline (includes vp_kernels.hip's helpers)
        # the fused STFT round trip (self-contained).  No SLP vectorisation: left to itself the compiler packs the single-precision kernel's
        # butterflies into v_pk_add_f32 / v_pk_mul_f32 (224 + 68 of them, 230 v_mov to form the pairs, 128 registers and spills), and packed f32
        # is no faster than scalar f32 on this chip (MI355X_MICROARCH.md); the double-precision kernels are unaffected.
        # -ffp-contract=fast (round 5), for THIS translation unit only: the STFT kernels have no reference arithmetic to reproduce,
        # so their butterflies, twiddle products and split/merge may fuse multiply-adds -- fewer vector instructions, one rounding less per
        # pair; the pitch-corrector / vocoder units keep -ffp-contract=off (the numerics contract above).
        # (VP_STFT_EXTRA_FLAGS: experiment builds, tools/ab.sh)
        jobs.append((os.path.join(CSRC, "vp_stft.hip"), os.path.join(tmp, "stft.o"), os.environ.get("VP_STFT_EXTRA_FLAGS", "-fno-slp-vectorize -ffp-contract=fast").split()))
        jobs.append((os.path.join(CSRC, "vp_capi.hip"), os.path.join(tmp, "capi.o"), []))

        def compile_one(job):
            # objects are cached by a hash of everything that goes into them (the big translation unit takes minutes)
            import hashlib
            src, obj, extra = job
            cmd = common + extra + ["-c", src, "-o", obj]
            hsh = hashlib.sha256((hipcc_version() + " ".join(cmd[:-1]).replace(tmp, "")).encode())
            deps = {"vp_kernels.hip": ["vp_kernels.hip", "vp_common.h"] + PARTS,
                    "vp_voc2.hip": ["vp_voc2.hip", "vp_voc2.h", "vp_kernels.hip", "vp_common.h"] + PARTS,
                    "vp_stft.hip": ["vp_stft.hip", "vp_stft.h", "vp_fft.inc", "vp_fft32.inc"]}.get(os.path.basename(src))
            if deps is None:
                deps = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.basename(src), os.path.join(ROOT, "include", "vp_amd.h")]
            for dep in deps:
                with open(dep if os.path.isabs(dep) else os.path.join(CSRC, dep), "rb") as f:
                    hsh.update(f.read())
            cache = os.path.join(HERE, ".build_cache")
            os.makedirs(cache, exist_ok=True)
            cached = os.path.join(cache, os.path.basename(obj) + "." + hsh.hexdigest()[:20])
            # force=True (what __graft_entry__.build() passes) COMPILES: the cache only serves the development loop
            if os.path.exists(cached) and not force and not os.environ.get("VP_NO_OBJ_CACHE"):
                try:
                    shutil.copy(cached, obj)
                    os.utime(cached)                               # (LRU: served objects are the young ones)
                    return obj
                except OSError:
                    pass                                           # (a concurrent build's prune_cache() took it between the test and the copy: compile)
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            part = cached + ".part%d" % os.getpid()                # (atomic: a concurrent build must never see half an object)
            shutil.copy(obj, part)
            os.replace(part, cached)
            return obj

        with ThreadPoolExecutor(len(jobs)) as ex:
            objs = list(ex.map(compile_one, jobs))
        prune_cache()
        link = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}"] + objs + ["-o", lib]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, stamps="--stamps" in sys.argv, poison="--poison" in sys.argv))
