"""CPU-side check of the built library's code objects (no GPU, no compiler run): the kernels that the BASELINE configs launch
must not use scratch memory -- a spilled register in a serial phase is a memory round trip, and 428 bytes per lane of it was
35 MB of traffic per launch in round 2.  The figures come from the AMDGPU metadata notes of
libvp_amd.so (tools/kernel_resources.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# what `python bench.py` (configs[1] + the configs2/3/4 legs, FAST and EXACT IIR) launches
BASELINE_KERNELS = [
    "vp_k_pitch_ws", "vp_k_pitch_ws_x",                       # configs[1]: 256 streams, pitch corrector (FAST / bit-exact), round 5
    "vp_k_pitch_ws_o24", "vp_k_pitch_ws_x_o24",               # ... at lpcPitch 16 .. 24 (round 6)
    "vp_k_pitch_fast_c", "vp_k_pitch_c",                      # ... and the phase kernels they replaced there (still what larger host blocks launch)
    "vp_k_vocoder",                                           # configs[2]: 256 streams, vocoder, LPC order 24
    "vp_k_pitch_lite_fast_c", "vp_k_v2_ingest_stage", "vp_k_v2_autocorr<4, true>", "vp_k_v2_autocorr<8, true>", "vp_k_v2_autocorr<4, false>", "vp_k_v2_autocorr<8, false>", "vp_k_v2_levinson2<40, 8, true>", "vp_k_v2_levinson2<40, 8, false>",
    "vp_k_v2_fir2<40, 8, true>", "vp_k_v2_fir2<40, 8, false>", "vp_k_v2_iir_fast<3, 1>", "vp_k_v2_ola",    # configs[3]
    "vp_k_pitch_fast", "vp_k_v2_levinson2<48, 32, true>", "vp_k_v2_fir2<48, 32, true>",                      # configs[4] geometry
    "vp_k_pitch_fast_multi_c", "vp_k_emit", "vp_k_ingest_gate",
    "vp_k_stft_fused<false, false>", "vp_k_stft_fused<true, false>", "vp_k_stft_fused2k<false>", "vp_k_stft_fused32<false>", "vp_k_stft_fused2k32<false>",      # the standalone STFT figures
]


@pytest.fixture(scope="module")
def resources():
    from vocoderproject_amd import build
    import kernel_resources
    if not os.path.exists(os.path.join(kernel_resources.LLVM, "llvm-readelf")):
        pytest.skip("no llvm-readelf in this image")
    return kernel_resources.kernel_resources(build.build())


def test_metadata_lists_the_kernels(resources):
    assert len(resources) > 40
    for k in BASELINE_KERNELS:
        assert k in resources, f"{k} not found in the code objects: {sorted(resources)[:8]} ..."


def test_no_scratch_in_the_kernels_the_baseline_configs_launch(resources):
    bad = {k: resources[k]["scratch"] for k in BASELINE_KERNELS if resources[k]["scratch"] > 0}
    assert not bad, f"kernels with scratch (bytes per lane): {bad}"


def test_scratch_of_every_product_kernel(resources):
    """Round 4: every kernel of the product library is free of scratch except the ones listed here with their ceilings -- the exact
    multi-block general build, the register-light GENERAL-geometry builds (128 registers; the common-case build the BASELINE configs
    launch has none) and the order-48 workgroup vocoder (never launched by a BASELINE config: the pipeline takes its batches).
    What was removed in round 4 were uniform values the compiler computed on the vector ALU once per kernel and then carried
    (an integer modulo by a run-time value, (double) of the frame length, the constant 1.0 of the DPP forms): DESIGN.md section 4.12."""
    allowed = {"vp_k_pitch_multi": 48, "vp_k_pitch_lite": 16, "vp_k_pitch_lite_fast": 32, "vp_k_pitch_lite_fast_multi": 72,
               "vp_k_pitch_lite_fast_multi_c": 48, "vp_k_vocoder_o48": 444,
               # (round 6: the multi-block builds of the wave-specialised kernel, 8 .. 19 spilled registers around their block loops)
               "vp_k_pitch_ws_mb": 40, "vp_k_pitch_ws_x_mb": 56, "vp_k_pitch_ws_mb_o24": 76, "vp_k_pitch_ws_x_mb_o24": 64}
    bad = {k: r["scratch"] for k, r in resources.items() if r["scratch"] > allowed.get(k, 0)}
    assert not bad, f"scratch bytes per lane above the ceilings: {bad}"
    for k in ("vp_k_pitch", "vp_k_pitch_fast", "vp_k_pitch_fast_multi", "vp_k_pitch_c", "vp_k_pitch_fast_c", "vp_k_pitch_fast_multi_c", "vp_k_pitch_lite_fast_c"):
        assert resources[k]["scratch"] == 0, (k, resources[k])


def test_occupancy_two_for_the_full_register_builds(resources):
    """512-thread workgroups need two wavefronts per SIMD: VGPRs + AGPRs must stay within 256."""
    for k in ("vp_k_pitch_fast_c", "vp_k_pitch_c", "vp_k_pitch_fast", "vp_k_vocoder", "vp_k_vocoder_o48"):
        r = resources[k]
        assert r["vgpr"] + r["agpr"] <= 256, (k, r)
    for k in ("vp_k_pitch_lite_fast_c", "vp_k_vocoder_lite"):
        assert resources[k]["vgpr"] + resources[k]["agpr"] <= 128, (k, resources[k])
    # the wave-specialised kernels run twelve wavefronts per workgroup: three per SIMD
    for k in ("vp_k_pitch_ws", "vp_k_pitch_ws_x", "vp_k_pitch_ws_o24", "vp_k_pitch_ws_x_o24"):
        assert resources[k]["vgpr"] + resources[k]["agpr"] <= 168 and resources[k]["scratch"] == 0, (k, resources[k])


def test_stft_kernel_fp64_instruction_count_matches_bench(resources):
    """bench.py prices the fused STFT kernel's fp64 vector share from a count of its ISA; the count must be the built kernel's."""
    import kernel_resources
    from vocoderproject_amd import build
    sys.path.insert(0, ROOT)
    import bench
    got = kernel_resources.fp64_op_counts(build.build(), "vp_k_stft_fused<false, false>")
    assert got == bench.STFT_FP64_OPS_PER_FRAME, got
    r = resources["vp_k_stft_fused<false, false>"]
    assert r["vgpr"] + r["agpr"] <= 256 and r["scratch"] == 0        # two wavefronts per SIMD (two workgroups per CU)
    r = resources["vp_k_stft_fused32<false>"]
    assert r["vgpr"] + r["agpr"] <= 128 and r["scratch"] == 0        # the single-precision build: four wavefronts per SIMD
