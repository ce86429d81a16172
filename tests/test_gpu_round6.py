"""GPU tests added in round 6 (through the C ABI).

* a bounded inter-wavefront wait that runs out is an ERROR: VP_ERR_TIMEOUT from the call that synchronises behind the launch, the
  handle poisoned until it is prepared again (round-5 verdict, weak item 2; the reference asserts on impossible state,
  PitchProcess.cpp:824,828);
* the exact mode's two recursion wavefronts on blocks of 2 cpf - 1 chunk steps and more (the circular wait the round-5 advisor
  described: prepareExplicit(44100, 1028 .. 1160, F = 1024, H = 512));
* vp_set_wave_specialised / vp_set_time_parallel toggled in the middle of a run on ONE handle (the frame in flight crosses from
  one kernel family to the other through HBM), in both arithmetic modes, entry states nChunk0 != 0 included.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 44100.0
VP_ERR_TIMEOUT = -9


def _streams(S, T, **kw):
    from vocoderproject_amd.synth import make_streams
    return np.ascontiguousarray(make_streams(S, T, **kw).numpy())


def _edge_streams(T, fs=FS):
    from test_gpu_round5 import _edge_streams as e
    return e(T, fs)


def _assert_equal(got, ref, what=""):
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"{what}: {len(bad)} samples differ, first at {bad[0]}, max abs {np.abs(got - ref).max()}"


def _state_key(p, s):
    d = p.pitch_state(s)
    d["a"] = d["a"].tobytes()
    return sorted(d.items())


def _timeouts(p):
    v = p.debug_stamps(reset=False)
    return [round(v[i] * 100.0) for i in (59, 60, 61)]


# ---- timeouts are errors -----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("iir", ["fast", "exact"])
@pytest.mark.parametrize("ws", [True, False])
def test_forced_wait_timeout_is_an_error_and_poisons_the_handle(ws, iir):
    """With the bounded waits cut to ONE poll the kernels' inter-wavefront waits run out (the wave-specialised kernel has dozens per
    block; the phase kernels wait for the prefix sums / the FFT's partial spectra / the grain table built ahead).  The host-pointer call
    synchronises behind its launch and must return VP_ERR_TIMEOUT, every later call must too, vp_last_error must say why, and a new
    prepare must give a working handle whose output equals the oracle's."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    N, S = 1024, 6
    x = _streams(S, N * 8)
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(FS, N, S)
    p.set_iir_mode(iir)
    p.set_yin_mode("xcorr")
    p.set_wave_specialised(ws)
    assert p.pitch_kernel_name().startswith("vp_k_pitch_ws") == ws
    p.debug_set_spin_limit(1)
    codes = []
    for b in range(6):                                       # (the first frames of a run need no waits that can lose: keep going)
        try:
            p.process(np.ascontiguousarray(x[:, :, b * N:(b + 1) * N]))
            codes.append(0)
        except VpError as e:
            codes.append(e.code)
            assert "wait" in str(e) or "timed out" in str(e), str(e)
    assert VP_ERR_TIMEOUT in codes, codes
    first = codes.index(VP_ERR_TIMEOUT)
    assert all(c == VP_ERR_TIMEOUT for c in codes[first:]), codes           # poisoned: every later call reports it
    with pytest.raises(VpError) as ei:
        p.synchronize()
    assert ei.value.code == VP_ERR_TIMEOUT
    assert sum(_timeouts(p)) > 0                                            # (the debug counters agree)
    # the device-pointer entry points report it too (on the call after the launch)
    import torch
    d_in = torch.zeros((S, 3, N), dtype=torch.float32, device="cuda")
    d_out = torch.empty((S, 2, N), dtype=torch.float32, device="cuda")
    with pytest.raises(VpError) as ei:
        p.process_device(d_in, d_out)
    assert ei.value.code == VP_ERR_TIMEOUT
    # prepare again: a clean handle
    p.debug_set_spin_limit(1 << 22)
    p.prepareToPlay(FS, N, S)
    p.set_iir_mode("exact")
    got = p.run(x)
    assert _timeouts(p)[2] >= 0
    for s in (0, S - 1):
        o = O.OracleStream(vocBool=0)
        o.prepare_to_play(FS, N)
        _assert_equal(got[s], o.run(x[s]), f"after re-prepare, stream {s}")
    p.close()


def test_timeout_on_device_entry_point_is_reported_by_the_next_call():
    """vp_process_block_device does not synchronise: the launch that timed out returns VP_OK, vp_synchronize (or the next process
    call) returns VP_ERR_TIMEOUT."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    N, S = 1024, 4
    x = torch.from_numpy(_streams(S, N * 6)).cuda()
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(FS, N, S)
    p.set_iir_mode("fast")
    p.set_yin_mode("xcorr")
    p.debug_set_spin_limit(1)
    d_out = torch.empty((S, 2, N), dtype=torch.float32, device="cuda")
    raised = None
    for b in range(6):
        try:
            p.process_device(x[:, :, b * N:(b + 1) * N].contiguous(), d_out)
        except VpError as e:
            raised = e.code
            break
    if raised is None:
        with pytest.raises(VpError) as ei:
            p.synchronize()
        raised = ei.value.code
    assert raised == VP_ERR_TIMEOUT
    p.close()


# ---- EXACT mode, blocks of 2 cpf - 1 steps and more ------------------------------------------------------------------------------

@pytest.mark.parametrize("N", [1028, 1100, 1160])
def test_exact_mode_two_chunks_per_frame_three_steps_per_block(N):
    """F = 1024, H = 512 (two chunks per frame), host blocks just above 1024 samples: every block runs three chunk steps = 2 cpf - 1,
    i.e. both recursion wavefronts of the exact mode hold chunks of different frames whose additions wait for each other's turn while
    the next Start waits for the older frame's last addition (the round-5 advisor's circular wait).  Bit-exact against the oracle and
    the phase kernels; no wait runs out."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    F, H, W, h = 1024, 512, 512, 128
    x = _edge_streams(N * 24)
    S = x.shape[0]
    outs = {}
    for ws in (True, False):
        p = BatchVocoderProcessor(vocBool=0)
        p.prepareExplicit(FS, N, S, F, H, W, h)
        p.set_iir_mode("exact")
        p.set_yin_mode("xcorr")
        p.set_wave_specialised(ws)
        if ws and not p.pitch_kernel_name().startswith("vp_k_pitch_ws"):
            p.close()
            pytest.skip(f"N={N}: the block's voice window does not fit the wave-specialised kernel's carve")
        outs[ws] = (p.run(x), [_state_key(p, s) for s in range(S)], p.ub_counters())
        assert _timeouts(p) == [0, 0, 0]
        p.close()
    _assert_equal(outs[True][0], outs[False][0], f"N={N}: wave-specialised vs phase kernels")
    assert outs[True][1:] == outs[False][1:]
    for s in range(S):
        o = O.OracleStream(vocBool=0)
        o.prepare_explicit(FS, N, F, H, W, h)
        _assert_equal(outs[True][0][s], o.run(x[s]), f"N={N} stream {s} vs oracle")


# ---- kernel families switched mid-run on one handle ------------------------------------------------------------------------------

@pytest.mark.parametrize("iir", ["exact", "fast"])
@pytest.mark.parametrize("N,every", [(1024, 1), (1024, 3), (512, 2), (256, 5)])
def test_wave_specialised_toggled_mid_run_on_one_handle(N, every, iir):
    """include/vp_amd.h promises the same bits from vp_k_pitch_ws* and the phase kernels and allows vp_set_wave_specialised at any
    time: the frame in flight (eFrame, outEFrame / yFrame ranges, impulse response, tracker state) then crosses from one family to the
    other through HBM.  Toggle every `every` blocks -- entry states nChunk0 = 1, 2, 3 all occur -- and compare with an untoggled run
    (bit for bit, both modes) and, in the exact mode, with the oracle."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    B = 36 if N >= 512 else 96
    x = _edge_streams(N * B)
    S = x.shape[0]

    def run(toggle):
        p = BatchVocoderProcessor(vocBool=0)
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode(iir)
        p.set_yin_mode("xcorr")
        out = np.empty((S, 2, N * B), np.float32)
        names = set()
        on = True
        for b in range(B):
            if toggle and b % every == 0:
                on = not on
                p.set_wave_specialised(on)
            names.add(p.pitch_kernel_name())
            out[:, :, b * N:(b + 1) * N] = p.process(np.ascontiguousarray(x[:, :, b * N:(b + 1) * N]))
        res = (out, [_state_key(p, s) for s in range(S)], p.ub_counters())
        assert _timeouts(p) == [0, 0, 0]
        p.close()
        return res, names

    (a, na), (b_, nb) = run(True), run(False)
    assert len(na) == 2 and len(nb) == 1, (na, nb)                          # both families really ran in the toggled run
    _assert_equal(a[0], b_[0], f"N={N} every={every} {iir}: toggled vs untoggled")
    assert a[1:] == b_[1:]
    if iir == "exact":
        for s in range(S):
            o = O.OracleStream(vocBool=0)
            o.prepare_to_play(FS, N)
            _assert_equal(a[0][s], o.run(x[s]), f"stream {s} vs oracle")


@pytest.mark.parametrize("iir", ["exact", "fast"])
def test_single_block_and_multi_block_calls_mixed_with_kernel_switches(iir):
    """One handle, device-pointer calls: single blocks on the wave-specialised kernel, a four-block call with vp_set_time_parallel
    (the one-launch phase kernel behind the analysis front end), single blocks with the wave-specialised kernel switched off, a
    three-block call without the front end ... The output must equal a plain block-by-block run bit for bit."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    N, S = 1024, 9
    plan = [("one", True, False), ("multi", 4, True), ("one", False, False), ("one", True, False), ("multi", 3, False),
            ("one", True, False), ("multi", 2, True), ("one", False, False), ("one", True, False)]
    B = sum(k[1] if k[0] == "multi" else 1 for k in plan)
    x = _edge_streams(N * B)
    ref_p = BatchVocoderProcessor(vocBool=0)
    ref_p.prepareToPlay(FS, N, S)
    ref_p.set_iir_mode(iir)
    ref_p.set_yin_mode("xcorr")
    ref = ref_p.run(x)
    ref_state = [_state_key(ref_p, s) for s in range(S)]
    ref_p.close()
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(FS, N, S)
    p.set_iir_mode(iir)
    p.set_yin_mode("xcorr")
    p.reserve_blocks(4)
    xd = torch.from_numpy(x).cuda()
    out = np.empty_like(ref)
    b = 0
    for kind, arg, tp in plan:
        if kind == "one":
            p.set_wave_specialised(arg)
            p.set_time_parallel(False)
            d_out = torch.empty((S, 2, N), dtype=torch.float32, device="cuda")
            p.process_device(xd[:, :, b * N:(b + 1) * N].contiguous(), d_out)
            out[:, :, b * N:(b + 1) * N] = d_out.cpu().numpy()
            b += 1
        else:
            p.set_wave_specialised(True)
            p.set_time_parallel(tp)
            d_in = torch.stack([xd[:, :, (b + k) * N:(b + k + 1) * N] for k in range(arg)]).contiguous()
            d_out = torch.empty((arg, S, 2, N), dtype=torch.float32, device="cuda")
            p.process_blocks_device(d_in, d_out)
            o = d_out.cpu().numpy()
            for k in range(arg):
                out[:, :, (b + k) * N:(b + k + 1) * N] = o[k]
            b += arg
    p.synchronize()
    _assert_equal(out, ref, f"{iir}: mixed calls vs block by block")
    assert [_state_key(p, s) for s in range(S)] == ref_state
    assert _timeouts(p) == [0, 0, 0]
    p.close()


def test_stft_plain_roundtrip_does_not_depend_on_the_phase_vocoder_attribute():
    """vp_stft_create validates its arguments before it touches the device, and a plain handle works whatever happened to the
    phase-vocoder build's LDS attribute (round-5 advisor): bad arguments -> VP_ERR_INVALID_ARG / VP_ERR_GEOMETRY without a sticky HIP error."""
    import ctypes as C
    import torch
    from vocoderproject_amd import load_library
    from vocoderproject_amd.processor import StftRoundTrip
    L = load_library()
    h = C.c_void_p()
    assert L.vp_stft_create(0, 0, 4096, 1024, 256, C.byref(h)) == -1
    assert L.vp_stft_create(0, 2, 4096, 1000, 250, C.byref(h)) == -4
    st = StftRoundTrip(2, 8192)
    x = torch.randn(2, 8192, device="cuda")
    y = torch.empty_like(x)
    st(x, y)
    torch.cuda.synchronize()
    assert float((x[:, 1024:-1024] - y[:, 1024:-1024]).abs().max()) < 1e-5
    st.close()


def test_cpp_sharded_batch_processor_device_resident_forms(tmp_path):
    """vp::ShardedBatchProcessor's device-pointer forms (round-5 verdict, item 7): processBlockDevice / processBlockMonoDevice /
    processBlocksDevice with every shard's buffers resident on its device, enqueued from the shards' worker threads, then
    synchronize().  G = 2 and 3 handles on device 0 against ONE handle driven through the host-pointer entry point: bit for bit,
    ragged shards, both processes on (single blocks, then three blocks per call), and the pitch corrector alone from mono buffers."""
    import os
    import shutil
    import subprocess
    from vocoderproject_amd import build
    gxx = shutil.which("g++")
    if not gxx or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no g++ / HIP headers")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = build.build()
    src = tmp_path / "t.cpp"
    src.write_text(r"""
#include <hip/hip_runtime_api.h>
#include "vp_amd.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#define HIPOK(x) do { if ((x) != hipSuccess) { std::printf("HIP error at %d\n", __LINE__); return 9; } } while (0)
int main() {
    const int S = 7, N = 1024, B = 12;
    std::vector<float> x((size_t)B * S * 3 * N);
    unsigned lcg = 777u;
    for (int b = 0; b < B; b++) for (int s = 0; s < S; s++) for (int c = 0; c < 3; c++) for (int i = 0; i < N; i++) {
        const double t = (double)(b * N + i) / 44100.0, f0 = 131.0 + 29.0 * s;
        lcg = lcg * 1664525u + 1013904223u;
        const double nz = ((double)(lcg >> 8) / 16777216.0 - 0.5) * 0.004;
        double v = 0.0;
        if (c == 0) { for (int h = 1; h <= 8; h++) v += std::sin(2.0 * M_PI * h * f0 * t) / h; v = 0.25 * v + nz; }
        else v = 0.15 * (2.0 * std::fmod(t * (110.0 + 13.0 * s), 1.0) - 1.0);
        x[(((size_t)b * S + s) * 3 + c) * N + i] = (float)v;
    }
    try {
        for (int mono = 0; mono < 2; mono++) {
            vp::BatchVocoderProcessor one(0);
            if (mono) one.setParameter("vocBool", 0);
            one.prepareToPlay(44100.0, N, S);
            one.setStreamParameter(2, "keyPitch", 5);
            std::vector<float> ref((size_t)B * S * 2 * N), got(ref.size(), -1.0f);
            for (int b = 0; b < B; b++) one.processBlock(&x[(size_t)b * S * 3 * N], &ref[(size_t)b * S * 2 * N]);
            for (int G = 2; G <= 3; G++) {
                vp::ShardedBatchProcessor sh(std::vector<int>(G, 0));
                if (mono) sh.setParameter("vocBool", 0);
                sh.prepareToPlay(44100.0, N, S);
                sh.setStreamParameter(2, "keyPitch", 5);
                sh.reserveBlocks(3);
                const int C = mono ? 1 : 3;
                std::vector<float *> dIn(G), dOut(G);
                std::vector<const float *> dInC(G);
                for (int g = 0; g < G; g++) {
                    HIPOK(hipSetDevice(sh.device(g)));
                    HIPOK(hipMalloc((void **)&dIn[g], (size_t)3 * sh.shardRange(g).second * C * N * sizeof(float)));
                    HIPOK(hipMalloc((void **)&dOut[g], (size_t)3 * sh.shardRange(g).second * 2 * N * sizeof(float)));
                    dInC[g] = dIn[g];
                }
                // blocks 0 .. 5 one at a time, 6 .. 11 three per call
                for (int b = 0; b < B; ) {
                    const int nb = b < 6 ? 1 : 3;
                    for (int g = 0; g < G; g++) {
                        const int lo = sh.shardRange(g).first, n = sh.shardRange(g).second;
                        for (int k = 0; k < nb; k++) for (int s = 0; s < n; s++) for (int c = 0; c < C; c++)
                            HIPOK(hipMemcpy(dIn[g] + (((size_t)k * n + s) * C + c) * N, &x[((((size_t)(b + k)) * S + lo + s) * 3 + c) * N], N * sizeof(float), hipMemcpyHostToDevice));
                    }
                    if (nb == 1) { if (mono) sh.processBlockMonoDevice(dInC.data(), dOut.data()); else sh.processBlockDevice(dInC.data(), dOut.data()); }
                    else { if (mono) sh.processBlocksMonoDevice(dInC.data(), dOut.data(), nb); else sh.processBlocksDevice(dInC.data(), dOut.data(), nb); }
                    sh.synchronize();
                    for (int g = 0; g < G; g++) {
                        const int lo = sh.shardRange(g).first, n = sh.shardRange(g).second;
                        for (int k = 0; k < nb; k++)
                            HIPOK(hipMemcpy(&got[((size_t)(b + k) * S + lo) * 2 * N], dOut[g] + (size_t)k * n * 2 * N, (size_t)n * 2 * N * sizeof(float), hipMemcpyDeviceToHost));
                    }
                    b += nb;
                }
                for (int g = 0; g < G; g++) { HIPOK(hipFree(dIn[g])); HIPOK(hipFree(dOut[g])); }
                if (std::memcmp(ref.data(), got.data(), ref.size() * sizeof(float)) != 0) { std::printf("mono %d G = %d differs\n", mono, G); return 5; }
            }
            double e = 0.0;
            for (float v : ref) e += (double)v * v;
            if (!(e > 0.0)) return 6;
        }
        std::printf("sharded device forms == single handle\n");
        return 0;
    } catch (const vp::Error &e) {
        std::printf("vp::Error %d: %s\n", e.code, e.what());
        return 1;
    }
}
""")
    exe = tmp_path / "t"
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-Wall", "-Werror", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(root, "include"),
                           str(src), "-o", str(exe), lib, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.dirname(lib) + ":/opt/rocm/lib", "-lpthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


# ---- the plugin's geometry at 48 kHz as bench.py times it --------------------------------------------------------------------------

@pytest.mark.parametrize("mode", ["pitch", "both"])
def test_48k_plugin_geometry_fast_mode_as_benched(mode):
    """prepareToPlay(48000, 1024) x 256 streams in VP_IIR_FAST / certified YIN -- bench.py's fs48k legs (round-5 verdict, item 3): frames
    1112 / 834 (chunks of 278 samples: the block-form recursion's ragged last block, round 6), vocoder 556 / 139.  Sampled streams
    against the oracle within the north_star tolerance with identical tracker states; the edge corpus (gate crossings, unvoiced
    bursts, silence, an octave jump, clipping) rides in the batch's first streams; the exact mode on the same input is bit-exact."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    fs, N, B, S = 48000.0, 1024, 24, 256
    x = _streams(S, N * B, fs=fs)
    e = _edge_streams(N * B, fs=fs)
    x[:e.shape[0]] = e
    x = np.ascontiguousarray(x)
    if mode == "pitch":
        x[:, 1:] = 0
    pick = list(range(e.shape[0])) + [37, 128, 255]
    kw = dict(vocBool=0) if mode == "pitch" else {}
    outs = {}
    for iir in ("fast", "exact"):
        p = BatchVocoderProcessor(**kw)
        p.prepareToPlay(fs, N, S)
        p.set_iir_mode(iir)
        p.set_yin_mode("xcorr")
        g = p.geometry()
        assert (g["F"], g["H"], g["C"], g["W"], g["h"]) == (1112, 834, 278, 556, 139)
        def decisions(s):                                                   # (the LPC coefficients are tolerance-mode arithmetic in VP_IIR_FAST)
            d = p.pitch_state(s)
            return [(k, d[k]) for k in ("period", "prevPeriod", "prevVoicedPeriod", "periodNew", "pitch", "prevPitch", "beta", "closestFreq", "gateOpen", "stMarkIdx", "anMarks", "stMarks")]
        outs[iir] = (p.run(x), [decisions(s) for s in pick])
        assert _timeouts(p) == [0, 0, 0]
        p.close()
    assert outs["fast"][1] == outs["exact"][1]                               # no decision depends on the recursions' or the LPC's arithmetic
    for i, s in enumerate(pick):
        o = O.OracleStream(**kw)
        o.prepare_to_play(fs, N)
        ref = o.run(x[s])
        _assert_equal(outs["exact"][0][s], ref, f"{mode} stream {s} exact vs oracle")
        err = outs["fast"][0][s].astype(np.float64) - ref
        assert np.sqrt((err ** 2).mean()) < 1e-4 and np.abs(err).max() < 1e-3, (mode, s, np.sqrt((err ** 2).mean()), np.abs(err).max())
    assert np.abs(outs["fast"][0]).max() > 0.05


# ---- the wave-specialised kernel at lpcPitch 16 .. 24 ------------------------------------------------------------------------------

@pytest.mark.parametrize("N", [1024, 512, 256])
@pytest.mark.parametrize("order", [16, 17, 24])
@pytest.mark.parametrize("iir,yin", [("exact", "xcorr"), ("fast", "xcorr"), ("exact", "direct")])
def test_wave_specialised_kernel_orders_16_to_24(order, iir, yin, N):
    """lpcPitch up to 24 on vp_k_pitch_ws*_o24 (round-5 verdict, item 2c; SURVEY section 8 names 24 beside the default 15;
    PluginProcessor.cpp:53-60 allows 2 .. 100): two groups of sixteen lags on the two LPC wavefronts, levinson_fast64 / levinson_row48,
    128 samples of impulse response, the block recursion with its history matrix / the one-lane exact chain.  Same bits as the phase
    kernels of that order (output, tracker state of every stream incl. the coefficients, UB-site counters), and in the exact mode
    as the oracle."""
    from vocoderproject_amd import BatchVocoderProcessor
    x = _edge_streams(1024 * 24)
    S = x.shape[0]
    runs = {}
    for ws in (True, False):
        p = BatchVocoderProcessor(vocBool=0, lpcPitch=order)
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode(iir)
        p.set_yin_mode(yin)
        p.set_wave_specialised(ws)
        name = p.pitch_kernel_name()
        assert name == (("vp_k_pitch_ws_o24" if iir == "fast" else "vp_k_pitch_ws_x_o24") if ws else ("vp_k_pitch_fast" if iir == "fast" else "vp_k_pitch")), name
        runs[ws] = (p.run(x), [_state_key(p, s) for s in range(S)], p.ub_counters())
        assert _timeouts(p) == [0, 0, 0]
        p.close()
    _assert_equal(runs[True][0], runs[False][0], f"order {order} N={N} {iir}/{yin}: wave-specialised vs phase kernels")
    assert runs[True][1:] == runs[False][1:]
    assert np.abs(runs[True][0]).max() > 0.05
    if iir == "exact":
        from oracle import oracle_py as O
        for s in range(S):
            o = O.OracleStream(vocBool=0, lpcPitch=order)
            o.prepare_to_play(FS, N)
            _assert_equal(runs[True][0][s], o.run(x[s]), f"order {order} N={N} stream {s} vs oracle")


# ---- several queued blocks in one launch of the wave-specialised kernel ---------------------------------------------------------------

@pytest.mark.parametrize("name,prepare,mono,dry", [
    ("plugin_geometry_mono", None, True, False),
    ("plugin_geometry_3ch_dry_paths", None, False, True),
    ("N512_two_steps_per_block", (44100.0, 512, 1024, 768, 512, 128), True, False),
    ("N256_one_step_per_block", (44100.0, 256, 1024, 768, 512, 128), False, False),
    ("two_chunks_per_frame", (44100.0, 1024, 1024, 512, 512, 256), True, False),
    ("eight_chunks_per_frame_N512", (44100.0, 512, 1024, 896, 512, 128), False, True),
])
def test_multi_block_launch_of_the_wave_specialised_kernel(name, prepare, mono, dry):
    _multi_block_ws_case(name, prepare, mono, dry)


@pytest.mark.parametrize("iir,order", [("exact", 15), ("fast", 24), ("exact", 24), ("exact", 16), ("fast", 17)])
@pytest.mark.parametrize("name,prepare,mono,dry", [
    ("plugin_geometry_mono", None, True, False),
    ("N512_3ch_dry_paths", (44100.0, 512, 1024, 768, 512, 128), False, True),
    ("two_chunks_per_frame", (44100.0, 1024, 1024, 512, 512, 256), True, False),
])
def test_multi_block_launch_exact_arithmetic_and_orders_to_24(name, prepare, mono, dry, iir, order):
    """The same in the exact arithmetic (vp_k_pitch_ws_x_mb) and for lpcPitch 16 .. 24 (vp_k_pitch_ws_mb_o24 / _x_mb_o24); the exact
    runs are, besides, bit-identical to the CPU oracle."""
    _multi_block_ws_case(name, prepare, mono, dry, iir=iir, order=order)


def _multi_block_ws_case(name, prepare, mono, dry, iir="fast", order=15):
    """vp_process_blocks*_device, pitch corrector alone: groups of up to sixteen queued blocks in ONE launch of
    vp_k_pitch_ws_mb (round-5 verdict, item 4) -- state, frame in flight, voice window and accumulator slice stay in LDS between the
    blocks, the gate is updated incrementally.  Calls of 5, 16, 19 (16 + 3) and 2 blocks, single-block calls in between: the output,
    the tracker state of every stream and the UB-site counters must equal a block-by-block run's bit for bit (edge corpus: gate
    crossings, silence, unvoiced bursts), and no wait may run out."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    fs, N = (prepare[0], prepare[1]) if prepare else (FS, 1024)
    plan = [5, 1, 16, 19, 1, 1, 2, 7]
    B = sum(plan)
    x = _edge_streams(N * B, fs=fs)
    S = x.shape[0]
    kw = dict(vocBool=0, lpcPitch=order)
    if dry:
        kw.update(gainVoice=-12.0, gainSynth=-20.0)
    kernel = "vp_k_pitch_ws" + ("" if iir == "fast" else "_x") + ("_o24" if order > 15 else "")
    keys = [12, 0, 5, 7, 12, 3, 9, 1, 11]

    def make():
        p = BatchVocoderProcessor(**kw)
        if prepare:
            p.prepareExplicit(fs, N, S, *prepare[2:])
        else:
            p.prepareToPlay(fs, N, S)
        p.set_iir_mode(iir)
        p.set_yin_mode("xcorr")
        for s_ in range(S):
            p.setStreamParameter(s_, "keyPitch", keys[s_ % 9])
        assert p.pitch_kernel_name() == kernel
        return p

    xd = torch.from_numpy(x).cuda()

    def blocks(b0, n):
        t = torch.stack([xd[:, :, (b0 + k) * N:(b0 + k + 1) * N] for k in range(n)])          # [n][S][3][N]
        return t[:, :, 0, :].contiguous() if mono else t.contiguous()

    ref_p = make()
    ref = np.empty((S, 2, N * B), np.float32)
    d_out = torch.empty((S, 2, N), dtype=torch.float32, device="cuda")
    for b in range(B):
        xb = blocks(b, 1)[0]
        if mono:
            ref_p.process_mono_device(xb, d_out)
        else:
            ref_p.process_device(xb, d_out)
        ref[:, :, b * N:(b + 1) * N] = d_out.cpu().numpy()
    ref_state = [_state_key(ref_p, s) for s in range(S)]
    ref_ub = ref_p.ub_counters()
    ref_p.close()

    p = make()
    p.reserve_blocks(max(plan))
    p.profile_enable(1)
    out = np.empty_like(ref)
    b = 0
    for n in plan:
        xin = blocks(b, n)
        if n == 1:
            if mono:
                p.process_mono_device(xin[0], d_out)
            else:
                p.process_device(xin[0], d_out)
            out[:, :, b * N:(b + 1) * N] = d_out.cpu().numpy()
        else:
            yo = torch.empty((n, S, 2, N), dtype=torch.float32, device="cuda")
            if mono:
                p.process_blocks_mono_device(xin, yo)
            else:
                p.process_blocks_device(xin, yo)
            o = yo.cpu().numpy()
            for k in range(n):
                out[:, :, (b + k) * N:(b + k + 1) * N] = o[k]
        b += n
    p.synchronize()
    launches = p.profile_read()[kernel][1]
    if name == "eight_chunks_per_frame_N512":
        assert launches >= B - 2          # seven chunk steps per frame, four per block: seven distinct schedules > WS_MB_SCHEDS -> block by block
    else:
        assert launches == sum(1 if n == 1 else (n + 15) // 16 for n in plan), launches      # ONE launch per group of up to sixteen blocks
    _assert_equal(out, ref, f"{name}: multi-block launches vs block by block")
    assert [_state_key(p, s) for s in range(S)] == ref_state
    assert p.ub_counters() == ref_ub
    assert _timeouts(p) == [0, 0, 0]
    assert np.abs(ref).max() > 0.05
    p.close()
    if iir == "exact":
        from oracle import oracle_py as O
        for s_ in (0, S - 1):
            o = O.OracleStream(**kw)
            if prepare:
                o.prepare_explicit(*[prepare[0], prepare[1]] + list(prepare[2:]))
            else:
                o.prepare_to_play(fs, N)
            o.set_param("keyPitch", keys[s_ % 9])
            xo = x[s_].copy()
            if mono:
                xo[1:] = 0.0                                   # (the mono entry points: null side-chain pointers, MyBuffer.cpp:93-102)
            _assert_equal(out[s_], o.run(xo), f"{name} {iir} order {order}: stream {s_} vs oracle")


def test_multi_block_launch_timeout_is_an_error():
    """The multi-block launch keeps the timeouts-are-errors contract: one poll per wait -> VP_ERR_TIMEOUT at the next synchronisation."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    N, S = 1024, 4
    x = torch.from_numpy(_streams(S, N * 8)).cuda()
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(FS, N, S)
    p.set_iir_mode("fast")
    p.set_yin_mode("xcorr")
    p.reserve_blocks(8)
    p.debug_set_spin_limit(1)
    xin = torch.stack([x[:, 0, k * N:(k + 1) * N] for k in range(8)]).contiguous()
    yo = torch.empty((8, S, 2, N), dtype=torch.float32, device="cuda")
    code = None
    try:
        p.process_blocks_mono_device(xin, yo)
        p.synchronize()
    except VpError as e:
        code = e.code
    assert code == VP_ERR_TIMEOUT
    p.close()


# ---- the device's note lookup against the reference's own Notes.cpp, compiled (oracle/_ref) -------------------------------------------

@pytest.mark.parametrize("ws", [True, False])
@pytest.mark.parametrize("iir", ["exact", "fast"])
def test_device_note_lookup_equals_the_compiled_reference(iir, ws):
    """Notes::getClosestFreq (Notes.cpp:79-110) is the one piece of the path whose REFERENCE SOURCE compiles here (standard library only):
    oracle/_ref/libnotes_ref.so is /root/reference/Source/Notes.cpp itself behind a C binding, built in the build container and carried
    to this box.  After every block, for every stream whose last frame start was voiced, the tracker's closestFreq (place_st_marks /
    notes_closest on the device, tables built by the host in vp_capi.hip) must be what the reference's Notes object returns for the
    tracker's pitch in the stream's key -- all 13 keys, both kernel families, both arithmetic modes -- and beta their quotient."""
    import torch
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    if O.build_ref() is None:
        pytest.skip("oracle/_ref/libnotes_ref.so did not travel and /root/reference is not here")
    S, N, B = 26, 1024, 14
    x = _streams(S, N * B)
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(FS, N, S)
    p.set_iir_mode(iir)
    p.set_yin_mode("xcorr")
    p.set_wave_specialised(ws)
    keys = [s_ % 13 for s_ in range(S)]
    refs = [O.RefNotes(k, 100.0, 800.0) for k in keys]
    for s_ in range(S):
        p.setStreamParameter(s_, "keyPitch", keys[s_])
    xd = torch.from_numpy(x).cuda()
    y = torch.empty((S, 2, N), dtype=torch.float32, device="cuda")
    seen = 0
    for b in range(B):
        p.process_device(xd[:, :, b * N:(b + 1) * N].contiguous(), y)
        p.synchronize()
        for s_ in range(S):
            st = p.pitch_state(s_)
            if st["gateOpen"] and st["pitch"] > 1:
                want = refs[s_].closest(st["pitch"], keys[s_])
                assert st["closestFreq"] == want, f"block {b} stream {s_} key {keys[s_]}: pitch {st['pitch']!r}: device {st['closestFreq']!r}, reference {want!r}"
                assert st["beta"] == want / st["pitch"]
                seen += 1
    assert seen > S * B // 2, seen
    assert _timeouts(p) == [0, 0, 0]
    p.close()


def test_pipeline_multi_block_plans_follow_the_batch_size(monkeypatch):
    """vp_process_blocks_device on the lane-per-window pipeline: where ONE block's windows already fill the chip (a lane per window: 8192
    windows and more per block, 1040 streams here) more blocks per launch only add the plans' fixed costs -- measured: the vocoder-only plan loses at every
    group size, the combined plan breaks even at eight blocks (profiles/r06_blocks_per_call.txt) -- so short calls go block by block there;
    smaller batches (260 streams) are short of wavefronts and take the plans from two blocks on.  The environment variable the suite sets
    (conftest.py) forces the plans.  Same output either way."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    N, nb = 1024, 4
    base = _streams(13, N * 2 * nb)

    def run(S, pitch):
        x = np.ascontiguousarray(np.tile(base, (S // 13, 1, 1)))
        xd = torch.from_numpy(x).cuda()
        p = BatchVocoderProcessor(pitchBool=int(pitch))
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode("fast")
        p.set_yin_mode("xcorr")
        p.reserve_blocks(nb)
        p.profile_enable(1)
        outs = []
        for c_ in range(2):
            xin = torch.stack([xd[:, :, (c_ * nb + k) * N:(c_ * nb + k + 1) * N] for k in range(nb)]).contiguous()
            yo = torch.empty((nb, S, 2, N), dtype=torch.float32, device="cuda")
            p.process_blocks_device(xin, yo)
            outs.append(yo.cpu().numpy())
        p.synchronize()
        prof = p.profile_read()
        launches = prof[p.pitch_kernel_name()][1] if pitch else prof["vp_k_vocoder"][1]
        p.close()
        return np.concatenate(outs), launches

    for pitch in (True, False):
        monkeypatch.setenv("VP_BOTH_MB_MIN", "2")
        y_plan, n_plan = run(1040, pitch)
        monkeypatch.delenv("VP_BOTH_MB_MIN")
        y_def, n_def = run(1040, pitch)
        _, n_small = run(260, pitch)
        assert n_plan < n_def, (pitch, n_plan, n_def)              # the plan: one pass per call; the default at this size: block by block
        assert n_small == n_plan, (pitch, n_small, n_plan)         # a batch that does not fill the chip takes the plan by default
        assert np.abs(y_plan).max() > 0.01
        d = np.abs(y_plan.astype(np.float64) - y_def)
        assert d.max() <= 4e-7 * max(1.0, float(np.abs(y_def).max())), (pitch, d.max())
