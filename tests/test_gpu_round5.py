"""GPU tests added in round 5 (through the C ABI).

* the wave-specialised pitch kernel (csrc/vp_pitch_ws.inc) against the CPU oracle and, bit for bit (output, tracker state,
  undefined-behaviour counters), against the phase kernels it replaces -- on the edge-case corpus, for every arithmetic mode,
  host block sizes from below the chunk to the largest it serves, and chunk grids other than the plugin's;
* the three workloads bench.py times that had no test at their own size and mode (round-4 verdict, weak item 2): BASELINE
  configs[2] at 256 streams with a stereo side chain on both windows, the +-12-semitone shift at 256 mono streams, the
  phase-vocoder stage at 256 x 65 536 samples.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 44100.0
RMS_TOL = 1e-4          # BASELINE.json north_star: "per-sample RMS error < 1e-4 vs reference"


def _streams(S, T, **kw):
    from vocoderproject_amd.synth import make_streams
    return np.ascontiguousarray(make_streams(S, T, **kw).numpy())


def _edge_streams(T, fs=FS):
    """gate crossings, unvoiced bursts, silence, an octave jump, clipping (as tests/test_gpu_parity.py's corpus) + plain streams"""
    rng = np.random.default_rng(11)
    base = _streams(9, T, fs=fs)
    x = base.copy()
    env = np.where((np.arange(T) // 9000) % 2 == 0, 1.0, 2e-5).astype(np.float32)
    x[0, 0] *= env
    noise = (rng.standard_normal(T) * 0.08).astype(np.float32)
    x[1, 0] = np.where((np.arange(T) // 7000) % 2 == 0, noise, base[1, 0])
    x[2, 0] = 0
    x[3, 1:] = 0
    t = np.arange(T) / fs
    x[4, 0] = (0.3 * np.sin(2 * np.pi * np.where(t < t[T // 2], 101.0, 640.0) * t)).astype(np.float32)
    x[5, 0] = np.clip(base[5, 0] * 8, -1, 1)
    return np.ascontiguousarray(x)


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


# ---- the wave-specialised kernel ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("N", [1024, 512, 256, 100])
@pytest.mark.parametrize("iir,yin", [("exact", "direct"), ("exact", "xcorr"), ("fast", "xcorr"), ("exact", "xcorr_force_fallback")])
def test_wave_specialised_kernel_equals_phase_kernels_and_oracle(N, iir, yin):
    """Same bits as the phase kernels in every mode (output, tracker state of every stream, UB-site counters), and -- in the
    exact mode -- as the oracle; no bounded wait ever timed out.  N = 100 and 256: blocks of at most one chunk step (launches with
    one instance or none), 512: two steps, 1024: four."""
    from vocoderproject_amd import BatchVocoderProcessor
    T = 1024 * 30 if N != 100 else 100 * 300
    x = _edge_streams(T)
    S = x.shape[0]
    runs = {}
    for ws in (True, False):
        p = BatchVocoderProcessor(vocBool=0)
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode(iir)
        p.set_yin_mode(yin)
        p.set_wave_specialised(ws)
        name = p.pitch_kernel_name()
        assert name.startswith("vp_k_pitch_ws") == ws, name
        y = p.run(x)
        runs[ws] = (y, [_state_key(p, s) for s in range(S)], p.ub_counters(), p.yin_certified_counts(reset=False))
        assert _timeouts(p) == [0, 0, 0]
        p.close()
    _assert_equal(runs[True][0], runs[False][0], f"N={N} {iir}/{yin}: wave-specialised vs phase kernels")
    assert runs[True][1] == runs[False][1]
    assert runs[True][2] == runs[False][2]
    assert runs[True][3] == runs[False][3]                                   # the same frames certified / handed to the fallback
    if yin == "xcorr_force_fallback":
        assert runs[True][3][0] == 0 and runs[True][3][1] > 0
    assert np.abs(runs[True][0]).max() > 0.05
    if iir == "exact":
        from oracle import oracle_py as O
        for s in range(S):
            o = O.OracleStream(vocBool=0)
            o.prepare_to_play(FS, N)
            _assert_equal(runs[True][0][s], o.run(x[s]), f"N={N} stream {s} vs oracle")


@pytest.mark.parametrize("name,prepare", [
    ("two_chunks_per_frame", (44100.0, 1024, 1024, 512, 512, 256)),          # chunk 512: a frame starts every step
    ("eight_chunks_per_frame", (44100.0, 512, 1024, 896, 512, 128)),         # chunk 128
    ("sixteen_chunks_per_frame", (44100.0, 256, 1024, 960, 512, 128)),       # chunk 64: a block is four steps of 64 samples
    ("chunk_128_block_1024", (44100.0, 1024, 1024, 896, 1024, 256)),         # eight steps per block, a start every seventh
])
def test_wave_specialised_kernel_other_chunk_grids(name, prepare):
    """Frames of 1024 samples on chunk grids other than the plugin's (PitchProcess::prepare is public: PitchProcess.h:40): the
    schedule's segments, the two parity buffers and the early residual must hold for 2, 8 and 16 chunks per frame."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    fs, N, F, H, W, h = prepare
    x = _edge_streams(N * (40 if N >= 1024 else 96))
    S = x.shape[0]
    for iir in ("exact", "fast"):
        outs = {}
        for ws in (True, False):
            p = BatchVocoderProcessor(vocBool=0)
            p.prepareExplicit(fs, N, S, F, H, W, h)
            p.set_iir_mode(iir)
            p.set_yin_mode("xcorr")
            p.set_wave_specialised(ws)
            if ws:
                assert p.pitch_kernel_name().startswith("vp_k_pitch_ws"), (name, p.pitch_kernel_name())
            outs[ws] = (p.run(x), [_state_key(p, s) for s in range(S)], p.ub_counters())
            assert _timeouts(p) == [0, 0, 0]
            p.close()
        _assert_equal(outs[True][0], outs[False][0], f"{name} {iir}")
        assert outs[True][1:] == outs[False][1:]
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareExplicit(fs, N, S, F, H, W, h)
    got = p.run(x)
    for s in range(S):
        o = O.OracleStream(vocBool=0)
        o.prepare_explicit(fs, N, F, H, W, h)
        _assert_equal(got[s], o.run(x[s]), f"{name} stream {s} vs oracle")


def test_wave_specialised_kernel_beside_the_vocoder_and_with_switches_mid_run():
    """Both processes on (the pitch kernel then neither ingests nor decides the gate itself), dry paths, per-stream keys and a
    pitchBool that is switched off and on again mid-run (cohorts: launches through a stream map): bit-exact against the oracle."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    N, B = 1024, 30
    x = _edge_streams(N * B)
    S = x.shape[0]
    p = BatchVocoderProcessor(gainVoice=-20.0, gainSynth=-30.0)
    p.prepareToPlay(FS, N, S)
    assert p.pitch_kernel_name() == "vp_k_pitch_ws_x"
    keys = [12, 0, 5, 7, 12, 3, 9, 1, 11]
    os_ = []
    for s in range(S):
        p.setStreamParameter(s, "keyPitch", keys[s])
        o = O.OracleStream(gainVoice=-20.0, gainSynth=-30.0, keyPitch=keys[s])
        o.prepare_to_play(FS, N)
        os_.append(o)
    for b in range(B):
        if b == 7:
            for s in (1, 4):
                p.setStreamParameter(s, "pitchBool", 0)
                os_[s].set_param("pitchBool", 0)
        if b == 13:
            p.setStreamParameter(1, "pitchBool", 1)
            os_[1].set_param("pitchBool", 1)
        if b == 19:
            p.setStreamParameter(6, "vocBool", 0)
            os_[6].set_param("vocBool", 0)
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        got = p.process(blk)
        for s in range(S):
            io = blk[s].copy()
            os_[s].process_block(io)
            _assert_equal(got[s], io[:2], f"block {b} stream {s}")
    assert _timeouts(p) == [0, 0, 0]


# ---- the workloads bench.py times, at their size and in their mode ----------------------------------------------------------------

@pytest.mark.parametrize("prepare", [None, (44100.0, 1024, 1024, 768, 1024, 256)], ids=["window_512_128", "window_1024_256"])
def test_config2_as_benched_256_streams_stereo_side_chain(prepare):
    """BASELINE configs[2] as bench.py's `configs2` leg runs it: 256 streams, vocoder only, lpcVoice 24, VP_IIR_FAST, [S][3][N] input
    through vp_process_block_device, on the reference's window (512/128) and on the metric's (1024/256): sampled streams against the
    oracle within the north_star tolerance, and the copies of a stream agree wherever they sit in the batch."""
    import torch
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B, U = 256, 1024, 10, 16
    params = dict(pitchBool=0, lpcVoice=24)
    base = _streams(U, N * B)
    idx = np.arange(S) % U
    x = np.ascontiguousarray(base[idx])
    p = BatchVocoderProcessor(**params)
    if prepare:
        p.prepareExplicit(prepare[0], N, S, *prepare[2:])
    else:
        p.prepareToPlay(FS, N, S)
    p.set_iir_mode("fast")
    assert p.vocoder_kernel_name() == "vp_k_vocoder"
    xd = torch.from_numpy(x).cuda().view(S, 3, B, N).permute(2, 0, 1, 3).contiguous()
    yd = torch.empty((B, S, 2, N), dtype=torch.float32, device="cuda")
    for b in range(B):
        p.process_device(xd[b], yd[b])
    torch.cuda.synchronize()
    got = yd.permute(1, 2, 0, 3).reshape(S, 2, B * N).cpu().numpy()
    for u in range(U):
        assert np.all(got[u::U] == got[u]), u
    pick = [0, 1, U + 3, 77, 128, 200, 255]
    ref = []
    for s in pick:
        o = O.OracleStream(**params)
        if prepare:
            o.prepare_explicit(prepare[0], N, *prepare[2:])
        else:
            o.prepare_to_play(FS, N)
        ref.append(o.run(base[idx[s]]))
    ref = np.stack(ref)
    err = got[pick].astype(np.float64) - ref
    rms = float(np.sqrt((err ** 2).mean()))
    scale = max(1.0, float(np.abs(ref).max()))
    print(f"configs[2] as benched ({'1024/256' if prepare else '512/128'}): rms err {rms:.3e}, max abs {np.abs(err).max():.3e}")
    assert rms < RMS_TOL and np.abs(err).max() <= 4e-6 * scale, (rms, np.abs(err).max(), scale)
    assert np.abs(ref).max() > 0.05


def test_pm12_semitone_shift_as_benched_256_mono_streams():
    """`value_pm12_semitone_shift` of bench.py: 256 mono streams, FAST IIR, certified YIN, fixed shift of +12 / -12 semitones on alternate
    streams, through vp_process_block_mono_device -- now on the wave-specialised kernel: decisions identical to the oracle's version of the
    extension, audio within the north_star tolerance."""
    import torch
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 256, 1024, 12
    x = _streams(S, N * B)
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(FS, N, S)
    p.set_iir_mode("fast")
    p.set_yin_mode("xcorr")
    assert p.pitch_kernel_name() == "vp_k_pitch_ws"
    for s in range(S):
        p.setPitchShift(12.0 if s % 2 == 0 else -12.0, stream=s)
    xm = torch.from_numpy(np.ascontiguousarray(x[:, 0])).cuda().view(S, B, N).permute(1, 0, 2).contiguous()
    yd = torch.empty((B, S, 2, N), dtype=torch.float32, device="cuda")
    for b in range(B):
        p.process_mono_device(xm[b], yd[b])
    torch.cuda.synchronize()
    got = yd.permute(1, 2, 0, 3).reshape(S, 2, B * N).cpu().numpy()
    pick = [0, 1, 2, 63, 64, 101, 128, 191, 254, 255]
    errs = []
    for s in pick:
        o = O.OracleStream(vocBool=0)
        o.prepare_to_play(FS, N)
        o.set_pitch_shift(12.0 if s % 2 == 0 else -12.0)
        ref = np.concatenate([o.process_block_mono(np.ascontiguousarray(x[s, 0, b * N:(b + 1) * N])) for b in range(B)], axis=1)
        errs.append(got[s].astype(np.float64) - ref)
        st, f = p.pitch_state(s), o.traces()[-1]
        assert (st["period"], st["anMarks"], st["stMarks"], st["beta"]) == (f["period"], f["anMarks"], f["stMarks"], f["beta"]), s
        assert np.abs(ref).max() > 0.02
    err = np.stack(errs)
    rms = float(np.sqrt((err ** 2).mean()))
    print(f"+-12 semitones as benched: rms err {rms:.3e}, max abs {np.abs(err).max():.3e}")
    assert rms < RMS_TOL and np.abs(err).max() <= 4e-6
    assert _timeouts(p) == [0, 0, 0]


def test_phase_vocoder_stage_as_benched_256_by_65536():
    """`stft_kernel.phase_vocoder_frames_per_s` of bench.py is timed at 256 streams x 65 536 samples: that launch, sampled streams
    against the NumPy restatement (tests/stft_reference.py).  No reference counterpart: parity unpinned by nature."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import stft_reference as R
    from vocoderproject_amd import StftRoundTrip
    S, T, F, hop = 256, 65536, 1024, 256
    x = _streams(S, T)[:, 0].copy()
    st = StftRoundTrip(S, T, F, hop)
    xd = torch.from_numpy(x).cuda()
    for semis in (12.0, -5.0):
        yd = torch.full_like(xd, float("nan"))                       # every output sample must be written
        st.pitch_shift(xd, yd, semis)
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        assert not np.isnan(got).any()
        ratio = 2.0 ** (semis / 12.0)
        for s in (0, 100, 255):
            ref = R.stft_roundtrip(x[s], F, hop, ratio=ratio)
            err = got[s].astype(np.float64) - ref
            rms = float(np.sqrt((err ** 2).mean()))
            assert rms < RMS_TOL and (np.abs(err) > 1e-5).mean() < 1e-3, (semis, s, rms)
            assert np.sqrt((ref ** 2).mean()) > 0.02
    st.close()


# ---- the C++-level shard helper ------------------------------------------------------------------------------------------------------

def test_cpp_sharded_batch_processor_equals_one_handle(tmp_path):
    """include/vp_amd.hpp vp::ShardedBatchProcessor (round-4 verdict, missing item 2): the batch split by stream over G handles, one
    worker thread per handle.  On this one-GPU box: G = 2 and G = 3 handles on device 0 against ONE handle on the same input, both
    processes on, a per-stream key and a fixed shift routed to their owners -- bit for bit, ragged shard sizes included."""
    import os
    import shutil
    import subprocess
    from vocoderproject_amd import build
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = build.build()
    src = tmp_path / "t.cpp"
    src.write_text(r"""
#include "vp_amd.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
int main() {
    const int S = 7, N = 1024, B = 14;
    std::vector<float> x((size_t)B * S * 3 * N);
    unsigned lcg = 12345u;
    for (int b = 0; b < B; b++) for (int s = 0; s < S; s++) for (int c = 0; c < 3; c++) for (int i = 0; i < N; i++) {
        const double t = (double)(b * N + i) / 44100.0, f0 = 120.0 + 37.0 * s;
        lcg = lcg * 1664525u + 1013904223u;
        const double nz = ((double)(lcg >> 8) / 16777216.0 - 0.5) * 0.004;
        double v = 0.0;
        if (c == 0) { for (int h = 1; h <= 8; h++) v += std::sin(2.0 * M_PI * h * f0 * t) / h; v = 0.25 * v + nz; }
        else v = 0.15 * (2.0 * std::fmod(t * (110.0 + 13.0 * s), 1.0) - 1.0);
        x[(((size_t)b * S + s) * 3 + c) * N + i] = (float)v;
    }
    try {
        vp::BatchVocoderProcessor one(0);
        one.setParameter("lpcVoice", 24);
        one.prepareToPlay(44100.0, N, S);
        one.setStreamParameter(4, "keyPitch", 3);
        one.setPitchShift(7.0, true, 6);
        std::vector<float> ref((size_t)B * S * 2 * N), got(ref.size());
        for (int b = 0; b < B; b++) one.processBlock(&x[(size_t)b * S * 3 * N], &ref[(size_t)b * S * 2 * N]);
        for (int G = 2; G <= 3; G++) {
            vp::ShardedBatchProcessor sh(std::vector<int>(G, 0));
            sh.setParameter("lpcVoice", 24);
            sh.prepareToPlay(44100.0, N, S);
            sh.setStreamParameter(4, "keyPitch", 3);
            sh.setPitchShift(7.0, true, 6);
            int total = 0;
            for (int g = 0; g < G; g++) { if (sh.shardRange(g).first != total) return 3; total += sh.shardRange(g).second; }
            if (total != S || sh.getLatencySamples() != one.getLatencySamples()) return 4;
            for (int b = 0; b < B; b++) sh.processBlock(&x[(size_t)b * S * 3 * N], &got[(size_t)b * S * 2 * N]);
            if (std::memcmp(ref.data(), got.data(), ref.size() * sizeof(float)) != 0) { std::printf("G = %d differs\n", G); return 5; }
        }
        double e = 0.0;
        for (float v : ref) e += (double)v * v;
        std::printf("sharded == single, out rms %.4f\n", std::sqrt(e / ref.size()));
        return e > 0.0 ? 0 : 6;
    } catch (const vp::Error &e) {
        std::printf("vp::Error %d: %s\n", e.code, e.what());
        return 1;
    }
}
""")
    exe = tmp_path / "t"
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), lib,
                           "-Wl,-rpath," + os.path.dirname(lib), "-lpthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
