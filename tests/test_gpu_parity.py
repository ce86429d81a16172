"""GPU parity tests: libvp_amd.so (through the C ABI) against the CPU oracle on identical inputs.
The kernels reproduce the reference's double arithmetic operation by operation, so the bar is
BIT-EXACT float32 output and identical pitch-tracker state (period, marks, LPC coefficients).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 44100.0


def _streams(S, T, **kw):
    from vocoderproject_amd.synth import make_streams
    return np.ascontiguousarray(make_streams(S, T, **kw).numpy())


def _oracle_run(x, N, params, prepare=None, trace=False):
    from oracle import oracle_py as O
    outs, traces = [], []
    for s in range(x.shape[0]):
        o = O.OracleStream(**params)
        if prepare:
            o.prepare_explicit(*prepare)
        else:
            o.prepare_to_play(FS, N)
        if trace:
            y, tr = o.run(x[s], trace=True)
            traces.append(tr)
        else:
            y = o.run(x[s])
        outs.append(y)
    return (np.stack(outs), traces) if trace else np.stack(outs)


def _gpu_run(x, N, params, prepare=None):
    from vocoderproject_amd import BatchVocoderProcessor
    p = BatchVocoderProcessor(**params)
    if prepare:
        fs, n, F, H, W, h = prepare
        p.prepareExplicit(fs, n, x.shape[0], F, H, W, h)
    else:
        p.prepareToPlay(FS, N, x.shape[0])
    y = p.run(x)
    return y, p


@pytest.mark.parametrize("mode", ["pitch", "voc", "both"])
def test_bit_exact_default_geometry(mode):
    S, N, B = 6, 1024, 24
    x = _streams(S, N * B)
    params = dict(pitchBool=int(mode != "voc"), vocBool=int(mode != "pitch"))
    ref = _oracle_run(x, N, params)
    got, p = _gpu_run(x, N, params)
    assert p.getLatencySamples() == 1024
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"{len(bad)} samples differ, first at {bad[0]}, max abs {np.abs(got - ref).max()}"
    assert np.abs(ref).max() > 0.05       # the comparison is not vacuous


def _assert_equal(got, ref, what=""):
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"{what}: {len(bad)} samples differ, first at {bad[0]}, max abs {np.abs(got - ref).max()}"


# ---- geometries -------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name,prepare,params,S,B", [
    # BASELINE configs[2] metric geometry: vocoder 1024/256, lpcVoice 24, lpcSynth 5
    ("cfg3_voc_1024_256", (44100.0, 1024, 1024, 768, 1024, 256), dict(pitchBool=0, lpcVoice=24, lpcSynth=5), 4, 16),
    # BASELINE configs[4]: 48 kHz, 2048-pt frames hop 512, orders 48 / 30 (lpcSynth max), both processes
    ("cfg5_48k_2048", (48000.0, 2048, 2048, 1536, 2048, 512), dict(lpcVoice=48, lpcPitch=48, lpcSynth=30), 3, 10),
    # 50 % overlap vocoder, half-frame pitch hop (chunksPerFrame = 2)
    ("half_overlap", (44100.0, 512, 1024, 512, 512, 256), dict(), 3, 24),
])
def test_bit_exact_explicit_geometry(name, prepare, params, S, B):
    fs, N = prepare[0], prepare[1]
    x = _streams(S, N * B, fs=fs)
    ref = _oracle_run(x, N, params, prepare=prepare)
    got, p = _gpu_run(x, N, params, prepare=prepare)
    _assert_equal(got, ref, name)
    assert np.abs(ref).max() > 0.05


def test_bit_exact_prepare_to_play_48k_non_power_of_two():
    # prepareToPlay(48000): vocoder 556/139, pitch 1112/834 (chunk 278), tauMax 480
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 3, 480, 40
    x = _streams(S, N * B, fs=48000.0)
    p = BatchVocoderProcessor()
    p.prepareToPlay(48000.0, N, S)
    g = p.geometry()
    assert (g["W"], g["h"], g["F"], g["H"], g["C"], g["tauMax"], g["latency"]) == (556, 139, 1112, 834, 278, 480, 1112)
    got = p.run(x)
    ref = []
    for s in range(S):
        o = O.OracleStream()
        o.prepare_to_play(48000.0, N)
        ref.append(o.run(x[s]))
    _assert_equal(got, np.stack(ref), "48k")


@pytest.mark.parametrize("N", [100, 128, 256, 512, 2048, 4096])
def test_bit_exact_block_sizes(N):
    S = 3
    T = (1024 * 20 // N) * N
    x = _streams(S, T)
    ref = _oracle_run(x, N, {})
    got, _ = _gpu_run(x, N, {})
    _assert_equal(got, ref, f"N={N}")


# ---- parameters ----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("params", [
    dict(keyPitch=8),                                            # F major instead of chromatic
    dict(keyPitch=0, lpcPitch=24, lpcVoice=24),
    dict(lpcVoice=100, lpcPitch=100, lpcSynth=30),               # parameter maxima (generic IIR path)
    dict(lpcVoice=2, lpcPitch=2, lpcSynth=2),                    # parameter minima
    dict(gainVoice=0.0, gainSynth=-6.0, gainVoc=-3.0, gainPitch=3.0),   # dry voice + dry carrier mixed in: L != R
    dict(gainVoc=-60.0, gainPitch=-60.0, gainVoice=-12.0),
])
def test_bit_exact_parameters(params):
    S, N, B = 3, 1024, 14
    x = _streams(S, N * B)
    x[:, 2] *= -0.5                                              # make the carrier channels differ
    ref = _oracle_run(x, N, params)
    got, _ = _gpu_run(x, N, params)
    _assert_equal(got, ref, str(params))
    if params.get("gainSynth", -60) > -59:
        assert not np.array_equal(got[:, 0], got[:, 1])


def test_parameter_changes_between_blocks():
    # treeState values are re-read per window/frame (VocoderProcess.cpp:193-194, PitchProcess.cpp:206);
    # pitchBool off calls silence() and freezes the chunk counters (PluginProcessor.cpp:218-221)
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 2, 512, 40
    x = _streams(S, N * B)
    sched = {6: ("keyPitch", 3), 10: ("lpcVoice", 16), 14: ("pitchBool", 0), 19: ("pitchBool", 1), 22: ("vocBool", 0),
             27: ("vocBool", 1), 30: ("gainVoice", -3.0), 33: ("lpcSynth", 12), 35: ("gainVoc", -20.0)}
    p = BatchVocoderProcessor()
    p.prepareToPlay(FS, N, S)
    os_ = [O.OracleStream() for _ in range(S)]
    for o in os_:
        o.prepare_to_play(FS, N)
    for b in range(B):
        if b in sched:
            k, v = sched[b]
            p.setParameter(k, v)
            for o in os_:
                o.set_param(k, v)
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        got = p.process(blk)
        for s in range(S):
            io = blk[s].copy()
            os_[s].process_block(io)
            _assert_equal(got[s], io[:2], f"block {b} stream {s}")


# ---- signal edge cases ---------------------------------------------------------------------------------------------------

def _edge_streams(T):
    rng = np.random.default_rng(11)
    base = _streams(6, T)
    x = base.copy()
    # 0: crosses the -60 dB gate back and forth
    env = np.where((np.arange(T) // 9000) % 2 == 0, 1.0, 2e-5).astype(np.float32)
    x[0, 0] *= env
    # 1: unvoiced noise bursts alternating with voiced segments
    noise = (rng.standard_normal(T) * 0.08).astype(np.float32)
    x[1, 0] = np.where((np.arange(T) // 7000) % 2 == 0, noise, base[1, 0])
    # 2: digital silence on the voice, 3: silence on the carrier
    x[2, 0] = 0
    x[3, 1:] = 0
    # 4: low voice (110 Hz region) with an octave jump in the middle
    t = np.arange(T) / FS
    x[4, 0] = (0.3 * np.sin(2 * np.pi * np.where(t < t[T // 2], 101.0, 640.0) * t)).astype(np.float32)
    # 5: hard-clipped full-scale input
    x[5, 0] = np.clip(base[5, 0] * 8, -1, 1)
    return np.ascontiguousarray(x)


@pytest.mark.parametrize("N", [1024, 256])
def test_bit_exact_gate_unvoiced_silence_edges(N):
    from oracle import oracle_py as O
    T = 1024 * 44
    x = _edge_streams(T)
    ref, traces = _oracle_run(x, N, {}, trace=True)
    got, p = _gpu_run(x, N, {})
    _assert_equal(got, ref, f"edge N={N}")
    assert any(t["gated"] for t in traces[0]) and any(not t["gated"] for t in traces[0])
    assert any(t["period"] == 0 and not t["gated"] for t in traces[1])          # unvoiced frames exist
    assert not got[2].any()                                                          # silent voice: nothing comes out
    # the undefined-behaviour sites of the reference are reached equally often on both sides
    ub_ref = np.zeros(5, int)
    for s in range(x.shape[0]):
        o = O.OracleStream()
        o.prepare_to_play(FS, N)
        o.run(x[s])
        ub_ref += np.array(o.ub_counters())
    assert list(ub_ref) == p.ub_counters()


def test_pitch_tracker_state_matches_frame_by_frame():
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 3, 256, 64                     # one chunk per block: a new frame every third block
    x = _edge_streams(N * B)[:S]
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(FS, N, S)
    os_ = [O.OracleStream(vocBool=0) for _ in range(S)]
    for o in os_:
        o.prepare_to_play(FS, N)
    checked = 0
    for b in range(B):
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        p.process(blk)
        for s in range(S):
            io = blk[s].copy()
            os_[s].process_block(io)
            tr = os_[s].traces()
            if tr and not tr[-1]["gated"]:
                st = p.pitch_state(s)
                f = tr[-1]
                assert (st["period"], st["periodNew"], st["prevPeriod"]) == (f["period"], f["periodNew"], f["prevPeriod"])
                assert st["anMarks"] == f["anMarks"] and st["stMarks"] == f["stMarks"]
                assert st["pitch"] == f["pitch"] and st["beta"] == f["beta"] and st["closestFreq"] == f["closestFreq"]
                np.testing.assert_array_equal(st["a"][:16], f["a"][:16])
                checked += 1
    assert checked > 20


# ---- API surface ------------------------------------------------------------------------------------------------------------

def test_inplace_processblock_zeroes_ch2_and_matches_out_of_place():
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 2, 1024, 6
    x = _streams(S, N * B)
    a = BatchVocoderProcessor(); a.prepareToPlay(FS, N, S)
    b = BatchVocoderProcessor(); b.prepareToPlay(FS, N, S)
    for k in range(B):
        blk = np.ascontiguousarray(x[:, :, k * N:(k + 1) * N])
        out = a.process(blk)
        io = blk.copy()
        b.processBlock(io)
        np.testing.assert_array_equal(io[:, :2], out)
        assert not io[:, 2].any()                                   # MyBuffer.cpp:115


def test_device_pointer_path_matches_host_path():
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 5, 1024, 8
    x = _streams(S, N * B)
    a = BatchVocoderProcessor(); a.prepareToPlay(FS, N, S)
    b = BatchVocoderProcessor(); b.prepareToPlay(FS, N, S)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty((S, 2, N), dtype=torch.float32, device="cuda")
    for k in range(B):
        blk = np.ascontiguousarray(x[:, :, k * N:(k + 1) * N])
        out = a.process(blk)
        b.process_device(xd[:, :, k * N:(k + 1) * N].contiguous(), yd)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(yd.cpu().numpy(), out)


def test_prepare_errors_mirror_reference_asserts():
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    p = BatchVocoderProcessor()
    with pytest.raises(VpError) as e:
        p.prepareExplicit(FS, 1024, 2, 1024, 768, 512, 100)        # VocoderProcess.cpp:110-114
    assert e.value.code == -3
    with pytest.raises(VpError) as e:
        p.prepareExplicit(FS, 1024, 2, 1000, 700, 512, 128)        # PitchProcess.cpp:91-92
    assert e.value.code == -4
    with pytest.raises(VpError):
        p.setParameter("lpcSynth", 31)                             # PluginProcessor.cpp:59 range end 30
    q = BatchVocoderProcessor()
    q.n_streams, q.N = 1, 16
    with pytest.raises(VpError) as e:
        q.process(np.zeros((1, 3, 16), np.float32))                # processBlock before prepareToPlay
    assert e.value.code == -2
    # the entry points added on top of the plugin's surface reject nonsense the same way
    import ctypes as C
    r = BatchVocoderProcessor()
    assert r.L.vp_process_blocks_device(r.h, C.c_void_p(16), C.c_void_p(16), 2, None) == -2          # not prepared
    r.prepareToPlay(FS, 256, 2)
    assert r.L.vp_process_blocks_device(r.h, C.c_void_p(16), C.c_void_p(16), 0, None) == -1          # VP_ERR_INVALID_ARG
    assert r.L.vp_process_blocks_device(r.h, None, C.c_void_p(16), 1, None) == -1
    assert r.L.vp_set_yin_mode(r.h, 7) == -1 and r.L.vp_set_iir_mode(r.h, 5) == -1
    assert r.L.vp_set_stream_params(r.h, 2, C.byref(r._p)) == -1 and r.L.vp_set_stream_params(r.h, -1, C.byref(r._p)) == -1


# ---- full-size properties (BASELINE configs[1] size: 256 streams) ------------------------------------------------------------

def test_full_batch_matches_oracle_on_sampled_streams_and_is_batch_invariant():
    S, N, B = 256, 1024, 10
    x = _streams(S, N * B)
    got, _ = _gpu_run(x, N, {})
    pick = [0, 1, 63, 64, 127, 200, 255]
    ref = _oracle_run(x[pick], N, {})
    _assert_equal(got[pick], ref, "256-stream batch vs oracle")
    # batch invariance: a stream's output does not depend on which batch it is processed in
    alone, _ = _gpu_run(np.ascontiguousarray(x[100:104]), N, {})
    _assert_equal(alone, got[100:104], "batch invariance")
    # dry path linearity at full size: pitch/vocoder off -> out = in delayed by the latency, exactly
    dry, p = _gpu_run(x, N, dict(pitchBool=0, vocBool=0, gainVoice=0.0))
    lat = p.getLatencySamples()
    np.testing.assert_array_equal(dry[:, 0, lat:], x[:, 0, :-lat])


# ---- VP_IIR_FAST: transposed-form synthesis filters (floating-point tolerance, stated by north_star) ----------------

RMS_TOL = 1e-4          # BASELINE.json north_star: "per-sample RMS error < 1e-4 vs reference"


@pytest.mark.parametrize("name,prepare,params", [
    ("default", None, dict()),
    ("voc_1024_256", (44100.0, 1024, 1024, 768, 1024, 256), dict(lpcVoice=24, lpcSynth=5)),
    ("cfg5_48k", (48000.0, 2048, 2048, 1536, 2048, 512), dict(lpcVoice=48, lpcPitch=48, lpcSynth=30)),
    ("order100", None, dict(lpcVoice=100, lpcPitch=100, lpcSynth=30)),
])
def test_fast_iir_mode_within_tolerance_and_decisions_identical(name, prepare, params):
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S = 4
    N = prepare[1] if prepare else 1024
    fs = prepare[0] if prepare else FS
    B = 14
    x = _edge_streams(N * B)[:S] if prepare is None else _streams(S, N * B, fs=fs)
    ref = _oracle_run(x, N, params, prepare=prepare)
    p = BatchVocoderProcessor(**params)
    if prepare:
        p.prepareExplicit(fs, N, S, *prepare[2:])
    else:
        p.prepareToPlay(FS, N, S)
    p.set_iir_mode("fast")
    assert p.get_iir_mode() == "fast"
    got = p.run(x)
    err = got.astype(np.float64) - ref
    rms = np.sqrt((err ** 2).mean())
    scale = max(1.0, float(np.abs(ref).max()))
    assert rms < RMS_TOL, rms
    # in practice the difference is a rare last-bit flip of the float32 cast
    assert np.abs(err).max() <= 4e-7 * scale, (np.abs(err).max(), scale)
    assert (got != ref).mean() < 0.02
    # the pitch tracker (every discrete decision) is bit-identical to the oracle's
    for s in range(S):
        o = O.OracleStream(vocBool=0, **{k: v for k, v in params.items()})
        if prepare:
            o.prepare_explicit(*prepare)
        else:
            o.prepare_to_play(FS, N)
        _, tr = o.run(x[s], trace=True)
        if tr and not tr[-1]["gated"]:
            st = p.pitch_state(s)
            f = tr[-1]
            assert (st["period"], st["anMarks"], st["stMarks"]) == (f["period"], f["anMarks"], f["stMarks"])
            assert st["beta"] == f["beta"]


# ---- VP_YIN_FFT (round 4): the certified form's cross-correlations by FFT (wavefront-level 512-point transforms) -------------------

@pytest.mark.parametrize("iir,fs,params,N,S", [("exact", 44100.0, dict(vocBool=0), 256, 8), ("fast", 44100.0, dict(vocBool=0), 1024, 8), ("fast", 44100.0, dict(), 1024, 8),
                                               ("exact", 22050.0, dict(), 512, 8), ("exact", 44100.0, dict(vocBool=0), 1024, 300)])
def test_fft_cross_correlation_yin_is_certified_bit_identical(iir, fs, params, N, S):
    """SURVEY.md 8f(1), as finished in round 4: the FFT evaluation of the YIN difference function's cross-correlations sits inside
    the certified form (every comparison of the pitch decision checked against the error bound, the reference's arithmetic as the
    fallback).  The full-register common-case builds evaluate the certified form no other way (frames of one or two 512-sample
    segments: 22.05 / 44.1 kHz); batches above 256 streams run the register-light builds, which keep the fused-multiply-add form.
    Period, marks and OUTPUT must be the oracle's bit for bit in the exact IIR mode (no flip rate to measure any more), and -- the
    forms being exchangeable -- identical between a batch on the FFT form and the same streams on the fused-multiply-add form."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    B = max(6, 96 * 256 // N)
    U = min(S, 8)
    base = np.concatenate([_streams(5, N * B, fs=fs), _edge_streams(N * B)[[0, 1, 4]]])[:U]
    x = np.ascontiguousarray(base[np.arange(S) % U])
    p = BatchVocoderProcessor(**params)
    p.prepareToPlay(fs, N, S)
    p.set_iir_mode(iir)
    p.set_yin_mode("xcorr")
    name = p.pitch_kernel_name()
    # (the builds that carry the FFT form: the common-case phase kernels and, round 5, the wave-specialised kernel that serves 44.1 kHz)
    assert ("lite" in name) == (S > 256) and (S > 256 or name.endswith("_c") or name.startswith("vp_k_pitch_ws"))
    got = p.run(x)
    st = [p.pitch_state(s) for s in range(U)]
    cert, fb = p.yin_certified_counts()
    assert cert > 3 * max(fb, 1), (cert, fb)
    if S > 256:                                                      # lite (fused multiply-adds) against a small batch (FFT): same bits
        q = BatchVocoderProcessor(**params)
        q.prepareToPlay(fs, N, U)
        q.set_iir_mode(iir)
        q.set_yin_mode("fft")
        assert "lite" not in q.pitch_kernel_name()
        small = q.run(base)
        _assert_equal(got[:U], small, "lite (fused multiply-adds) vs full-register (FFT) build")
        for s in range(U):
            for k in ("period", "anMarks", "stMarks", "beta", "pitch"):
                assert np.array_equal(st[s][k], q.pitch_state(s)[k]), (s, k)
    if iir == "exact":
        for s in range(U):
            o = O.OracleStream(**params)
            o.prepare_to_play(fs, N)
            ref, tr = o.run(base[s], trace=True)
            _assert_equal(got[s], ref, f"stream {s} vs oracle")
            if tr and not tr[-1]["gated"]:
                f = tr[-1]
                assert (st[s]["period"], st[s]["anMarks"], st[s]["stMarks"], st[s]["beta"]) == (f["period"], f["anMarks"], f["stMarks"], f["beta"])


def test_standalone_stft_roundtrip_against_numpy_fft():
    """The STFT kernel has NO reference counterpart (the reference has no FFT): it is checked against
    numpy.fft (parity unpinned) -- perfect reconstruction in the interior and the magnitude spectrum."""
    import torch
    from vocoderproject_amd import StftRoundTrip
    S, F, hop, T = 5, 1024, 256, 1024 * 24
    x = _streams(S, T)[:, 0].copy()
    st = StftRoundTrip(S, T, F, hop)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    md = torch.empty((S, st.n_frames, F // 2 + 1), dtype=torch.float32, device="cuda")
    st(xd, yd, md)
    torch.cuda.synchronize()
    y, mag = yd.cpu().numpy(), md.cpu().numpy()
    # interior samples are covered by F/hop frames: exact reconstruction up to float32 rounding
    np.testing.assert_allclose(y[:, F:T - F], x[:, F:T - F], rtol=0, atol=2e-6)
    w = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * np.arange(F) / F))
    for s in (0, S - 1):
        for f in (0, 7, st.n_frames - 1):
            ref = np.abs(np.fft.rfft(x[s, f * hop:f * hop + F].astype(np.float64) * w))
            np.testing.assert_allclose(mag[s, f], ref, rtol=1e-5, atol=1e-5)


def test_iir_mode_can_change_mid_frame():
    """The block-form fast IIR keeps the frame's impulse response across calls; switching modes between
    blocks (frames span several 256-sample calls here) must stay within the fast-mode tolerance."""
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 3, 256, 60
    x = _streams(S, N * B)
    ref = _oracle_run(x, N, {})
    p = BatchVocoderProcessor()
    p.prepareToPlay(FS, N, S)
    got = np.empty((S, 2, N * B), np.float32)
    for b in range(B):
        p.set_iir_mode("fast" if (b // 2) % 2 == 0 else "exact")
        got[:, :, b * N:(b + 1) * N] = p.process(np.ascontiguousarray(x[:, :, b * N:(b + 1) * N]))
    err = got.astype(np.float64) - ref
    assert np.abs(err).max() <= 4e-7 * max(1.0, float(np.abs(ref).max()))


def test_register_light_kernel_variant_for_large_batches():
    """Above 256 streams the host launches vp_k_pitch_lite (two workgroups per CU).  Same bar: bit-exact in
    the default mode, and independent of which variant processed a stream."""
    S, N, B = 320, 1024, 6
    x = _streams(S, N * B)
    got, _ = _gpu_run(x, N, {})                               # lite variant, exact IIR (lpcPitch 15 <= 16)
    pick = [0, 5, 128, 255, 256, 319]
    ref = _oracle_run(x[pick], N, {})
    _assert_equal(got[pick], ref, "lite variant vs oracle")
    small, _ = _gpu_run(np.ascontiguousarray(x[250:260]), N, {})      # regular variant
    _assert_equal(small, got[250:260], "lite vs regular variant")
    # lpcPitch 24 in exact mode must fall back to the regular kernel and stay exact
    got24, _ = _gpu_run(x, N, dict(lpcPitch=24, vocBool=0))
    ref24 = _oracle_run(x[[3, 300]], N, dict(lpcPitch=24, vocBool=0))
    _assert_equal(got24[[3, 300]], ref24, "order 24, large batch")


# ---- randomised sweep over sample rates, block sizes, orders, keys, gains and modes ---------------------------------------

def _fuzz_case(seed):
    rng = np.random.default_rng(seed)
    fs = float(rng.choice([8000.0, 11025.0, 16000.0, 22050.0, 32000.0, 44100.0, 48000.0, 44099.0, 88200.0]))
    N = int(rng.choice([64, 100, 278, 441, 512, 1000, 1024, 1536, 3000]))
    params = dict(lpcVoice=int(rng.integers(2, 101)), lpcPitch=int(rng.integers(2, 101)), lpcSynth=int(rng.integers(2, 31)),
                  keyPitch=int(rng.integers(0, 13)), gainPitch=float(rng.uniform(-20, 6)), gainVoc=float(rng.uniform(-20, 6)),
                  gainVoice=float(rng.choice([-60.0, -30.0, 0.0])), gainSynth=float(rng.choice([-60.0, -12.0])),
                  pitchBool=int(rng.random() < 0.85), vocBool=int(rng.random() < 0.7))
    return fs, N, params


@pytest.mark.parametrize("seed", list(range(16)))
def test_randomised_configurations_bit_exact(seed):
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    fs, N, params = _fuzz_case(1000 + seed)
    S = 3
    T = max(6, int(26000 * fs / 44100.0) // N) * N
    x = _streams(S, T, fs=fs)
    if seed % 3 == 0:
        x[0, 0] *= np.where((np.arange(T) // 7000) % 2 == 0, 1.0, 2e-5).astype(np.float32)      # gate crossings
    p = BatchVocoderProcessor(**params)
    try:
        p.prepareToPlay(fs, N, S)
    except VpError as e:
        assert e.code == -4, e                     # frame does not fit LDS at this sample rate: reported, not silent
        pytest.skip(f"geometry for fs={fs} exceeds the LDS budget (VP_ERR_GEOMETRY)")
    got = p.run(x)
    ref = []
    for s in range(S):
        o = O.OracleStream(**params)
        o.prepare_to_play(fs, N)
        ref.append(o.run(x[s]))
    _assert_equal(got, np.stack(ref), f"seed {seed}: fs={fs} N={N} {params}")


def test_pitch_kernel_build_selection():
    """The builds of the pitch kernel (IIR mode x register budget x common-case geometry x FFT) are selected as documented."""
    from vocoderproject_amd import BatchVocoderProcessor
    p = BatchVocoderProcessor(vocBool=0)
    assert p.pitch_kernel_name() == ""
    p.prepareToPlay(FS, 1024, 8)
    assert p.pitch_kernel_name() == "vp_k_pitch_ws_x"            # the plugin's own geometry, up to 256 streams: the wave-specialised kernel (round 5)
    p.set_iir_mode("fast")
    assert p.pitch_kernel_name() == "vp_k_pitch_ws"
    p.set_wave_specialised(False)
    assert p.pitch_kernel_name() == "vp_k_pitch_fast_c"          # chunk of 256 samples, lpcPitch 15, tauMax 441: common case of the phase kernels
    p.set_iir_mode("exact")
    assert p.pitch_kernel_name() == "vp_k_pitch_c"
    w = BatchVocoderProcessor(vocBool=0)
    w.prepareToPlay(FS, 4096, 8)                                 # sixteen chunk steps per block: their voice window does not fit beside two frames
    assert w.pitch_kernel_name() == "vp_k_pitch_c"
    g = BatchVocoderProcessor(vocBool=0, lpcPitch=24)            # an order the common-case builds do not cover: the wave-specialised _o24 builds (round 6)
    g.prepareToPlay(FS, 1024, 8)
    assert g.pitch_kernel_name() == "vp_k_pitch_ws_x_o24"
    g.set_wave_specialised(False)
    assert g.pitch_kernel_name() == "vp_k_pitch"
    g.set_iir_mode("fast")
    assert g.pitch_kernel_name() == "vp_k_pitch_fast"
    g.set_wave_specialised(True)
    assert g.pitch_kernel_name() == "vp_k_pitch_ws_o24"
    g2 = BatchVocoderProcessor(vocBool=0, lpcPitch=25)           # beyond WS_ORDER_MAX: the general phase kernels
    g2.prepareToPlay(FS, 1024, 8)
    assert g2.pitch_kernel_name() == "vp_k_pitch"
    q = BatchVocoderProcessor(vocBool=0)
    q.prepareToPlay(FS, 1024, 300)
    assert q.pitch_kernel_name() == "vp_k_pitch_lite"
    q.set_iir_mode("fast")
    assert q.pitch_kernel_name() == "vp_k_pitch_lite_fast_c"
    q.set_yin_mode("fft")                                        # (certified FFT cross-correlations: a path of the common-case builds, no builds of its own)
    assert q.pitch_kernel_name() == "vp_k_pitch_lite_fast_c"


def test_long_run_with_random_parameter_schedule():
    """Nine seconds of audio per stream (400 blocks), parameters changed at random blocks, edge-case signals:
    every block bit-exact, and the tracker state and the undefined-behaviour counters equal at the end."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 6, 1024, 400
    x = _edge_streams(N * B)[:S]
    rng = np.random.default_rng(2024)
    choices = [("keyPitch", lambda: int(rng.integers(0, 13))), ("lpcVoice", lambda: int(rng.integers(2, 101))),
               ("lpcSynth", lambda: int(rng.integers(2, 31))), ("gainPitch", lambda: float(rng.uniform(-30, 6))),
               ("gainVoc", lambda: float(rng.uniform(-30, 6))), ("gainVoice", lambda: float(rng.choice([-60.0, -20.0, 0.0]))),
               ("gainSynth", lambda: float(rng.choice([-60.0, -20.0]))), ("pitchBool", lambda: int(rng.random() < 0.8)),
               ("vocBool", lambda: int(rng.random() < 0.7))]
    sched = {}
    for b in sorted(rng.choice(np.arange(3, B), size=36, replace=False)):
        k, f = choices[int(rng.integers(0, len(choices)))]
        sched[int(b)] = (k, f())
    p = BatchVocoderProcessor()
    p.prepareToPlay(FS, N, S)
    os_ = [O.OracleStream() for _ in range(S)]
    for o in os_:
        o.prepare_to_play(FS, N)
    for b in range(B):
        if b in sched:
            k, v = sched[b]
            p.setParameter(k, v)
            for o in os_:
                o.set_param(k, v)
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        got = p.process(blk)
        for s in range(S):
            io = blk[s].copy()
            os_[s].process_block(io)
            _assert_equal(got[s], io[:2], f"block {b} stream {s} (schedule {sched})")
    ub = np.sum([o.ub_counters() for o in os_], axis=0)
    assert list(p.ub_counters()) == list(ub)


def test_per_stream_parameters():
    """Every stream of a batch is its own plugin instance: per-stream key / orders / gains, set before and between
    blocks, against one oracle per stream with that stream's values."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    S, N, B = 5, 1024, 30
    x = _streams(S, N * B)
    per = [dict(), dict(keyPitch=3, gainPitch=-6.0), dict(lpcVoice=16, lpcSynth=12, gainVoc=-12.0),
           dict(keyPitch=0, gainVoice=-10.0, gainSynth=-20.0), dict(keyPitch=7, lpcVoice=100, lpcSynth=30)]
    later = {12: (1, "keyPitch", 9), 17: (2, "gainVoc", 3.0), 21: (4, "lpcVoice", 8)}
    p = BatchVocoderProcessor()
    with pytest.raises(VpError):
        p.setStreamParameter(0, "keyPitch", 2)                      # before prepare
    p.prepareToPlay(FS, N, S)
    os_ = []
    for s_, kv in enumerate(per):
        o = O.OracleStream()
        o.prepare_to_play(FS, N)
        for k, v in kv.items():
            p.setStreamParameter(s_, k, v)
            o.set_param(k, v)
        os_.append(o)
    assert p.getStreamParameter(1, "keyPitch") == 3 and p.getStreamParameter(0, "keyPitch") == 12
    with pytest.raises(VpError):
        p.setStreamParameter(0, "lpcPitch", 20)                     # per handle only (read at prepare, selects the kernel build)
    with pytest.raises(VpError):
        p.setStreamParameter(S, "keyPitch", 1)                      # no such stream
    for b in range(B):
        if b in later:
            s_, k, v = later[b]
            p.setStreamParameter(s_, k, v)
            os_[s_].set_param(k, v)
        if b == 25:                                                  # the handle-wide set puts every stream on one set again
            p.setParameter("keyPitch", 5)
            for o in os_:
                for k, v in dict(gainPitch=0.0, gainVoice=-60.0, gainSynth=-60.0, gainVoc=0.0, lpcVoice=40, lpcSynth=5).items():
                    o.set_param(k, v)
                o.set_param("keyPitch", 5)
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        got = p.process(blk)
        for s_ in range(S):
            io = blk[s_].copy()
            os_[s_].process_block(io)
            _assert_equal(got[s_], io[:2], f"block {b} stream {s_}")


@pytest.mark.parametrize("iir", ["exact", "fast"])
def test_multi_block_launch_equals_block_by_block(iir):
    """vp_process_blocks_device: B blocks in one launch give exactly what B single-block calls give (and, in exact
    mode, what the oracle gives), gate crossings and unvoiced stretches included; other plans fall back to single calls
    (both processes in the fast IIR mode: the combined plan, equal to rounding level with identical decisions)."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 6, 1024, 24
    x = _edge_streams(N * B)[:S]
    xb = torch.from_numpy(np.ascontiguousarray(x.reshape(S, 3, B, N).transpose(2, 0, 1, 3))).cuda()      # [B][S][3][N]

    def run(split, voc="auto", **params):
        p = BatchVocoderProcessor(**params)
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode(iir)
        p.set_vocoder_path(voc)
        p.reserve_blocks(max(split))
        y = torch.empty((B, S, 2, N), dtype=torch.float32, device="cuda")
        b = 0
        for n in split:
            p.process_blocks_device(xb[b:b + n].contiguous(), y[b:b + n])
            b += n
        assert b == B
        torch.cuda.synchronize()
        return y.cpu().numpy(), [p.pitch_state(s_) for s_ in range(S)], p.ub_counters()

    # pitch only (one launch per group) / both on the default path (six streams: the workgroup vocoder, so block by block) /
    # both with the lane-per-window pipeline forced, which single-block calls then run too (FAST: the combined plan)
    for params, voc in ((dict(vocBool=0), "auto"), (dict(), "auto"), (dict(), "batched")):
        ref, st_ref, ub_ref = run([1] * B, voc, **params)
        for split in ([B], [5, 1, 7, 11], [2] * 12):
            got, st, ub = run(split, voc, **params)
            if iir == "fast" and not params and voc == "batched":
                # both processes, tolerance mode, pipeline: the combined multi-block plan adds the call's chunks before
                # its windows instead of windows-first per block -- the same vocoder arithmetic as the single-block calls (the plan only
                # runs where they take the pipeline too), so what differs is the rounding of the additions into the accumulator
                dlt = got.astype(np.float64) - ref
                assert np.abs(dlt).max() <= 2e-6 * max(1.0, float(np.abs(ref).max())), (split, np.abs(dlt).max())
            else:
                _assert_equal(got, ref, f"{params} {voc} split {split}")
            assert ub == ub_ref
            for s_ in range(S):
                for k in st_ref[s_]:
                    assert np.array_equal(st[s_][k], st_ref[s_][k]), (s_, k)
    if iir == "exact":
        want = _oracle_run(x, N, dict(vocBool=0))
        got, _, _ = run([B], vocBool=0)
        _assert_equal(got.transpose(1, 2, 0, 3).reshape(S, 2, B * N), want, "multi-block vs oracle")


def test_certified_cross_correlation_yin_is_bit_identical():
    """VP_YIN_XCORR computes the YIN difference function as energies minus a cross-correlation (fused multiply-adds, a
    third of the arithmetic), certifies every comparison of the pitch decision against a rounding-error bound and falls
    back to the reference's arithmetic when one is too close to call: the OUTPUT must therefore be bit-identical to
    VP_YIN_DIRECT (not merely within tolerance), for every signal, and the forced-fallback diagnostic mode as well."""
    from vocoderproject_amd import BatchVocoderProcessor
    N, B = 1024, 30
    x = np.concatenate([_edge_streams(N * B), _streams(10, N * B)], axis=0)
    S = x.shape[0]
    outs, states, counts = {}, {}, {}
    for mode in ("direct", "xcorr", "fft", "xcorr_force_fallback"):
        p = BatchVocoderProcessor()
        p.prepareToPlay(FS, N, S)
        p.set_yin_mode(mode)
        assert p.get_yin_mode() == mode
        p.yin_certified_counts(reset=True)
        outs[mode] = p.run(x)
        counts[mode] = p.yin_certified_counts()                                      # (certified, fallback) frames
        states[mode] = [p.pitch_state(s_) for s_ in range(S)]
    for mode in ("xcorr", "fft", "xcorr_force_fallback"):
        _assert_equal(outs[mode], outs["direct"], mode)
        for s_ in range(S):
            for k in states["direct"][s_]:
                assert np.array_equal(states[mode][s_][k], states["direct"][s_][k]), (mode, s_, k)
    cert, fb = counts["xcorr"]
    print(f"certified {cert} frames, fell back on {fb} (silence / start-up frames have a zero running sum)")
    assert counts["direct"] == (0, 0)
    assert cert > 0 and fb < 0.25 * (cert + fb)
    assert counts["fft"] == counts["xcorr"]                      # (two names of the certified form: which evaluation runs is the build's)
    assert counts["xcorr_force_fallback"][0] == 0 and counts["xcorr_force_fallback"][1] == cert + fb


@pytest.mark.parametrize("seed", [3, 8, 13])
def test_certified_yin_random_configurations(seed):
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    fs, N, params = _fuzz_case(1000 + seed)
    S = 3
    T = max(6, int(26000 * fs / 44100.0) // N) * N
    x = _streams(S, T, fs=fs)
    outs = []
    for mode in ("direct", "xcorr", "fft"):
        p = BatchVocoderProcessor(**params)
        try:
            p.prepareToPlay(fs, N, S)
        except VpError:
            pytest.skip("geometry exceeds the LDS budget")
        p.set_yin_mode(mode)
        outs.append(p.run(x))
    _assert_equal(outs[1], outs[0], f"seed {seed}: fs={fs} N={N} {params}")
    _assert_equal(outs[2], outs[0], f"fft, seed {seed}: fs={fs} N={N} {params}")


@pytest.mark.parametrize("seed", [0, 2, 5, 7, 11, 13, 14])
def test_randomised_configurations_fast_modes_against_exact(seed):
    """The bench's modes (FAST IIR + certified XCORR YIN) at random sample rates / block sizes / orders: output within the
    tolerance of, and every pitch decision identical to, the library's default modes (which are bit-exact to the oracle)."""
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    fs, N, params = _fuzz_case(1000 + seed)
    S = 3
    T = max(6, int(26000 * fs / 44100.0) // N) * N
    x = _streams(S, T, fs=fs)
    outs, states = [], []
    for iir, yin in (("exact", "direct"), ("fast", "xcorr")):
        p = BatchVocoderProcessor(**params)
        try:
            p.prepareToPlay(fs, N, S)
        except VpError:
            pytest.skip("geometry exceeds the LDS budget")
        p.set_iir_mode(iir)
        p.set_yin_mode(yin)
        outs.append(p.run(x).astype(np.float64))
        states.append([p.pitch_state(s_) for s_ in range(S)])
    err = outs[1] - outs[0]
    rms = np.sqrt((err ** 2).mean())
    assert rms < RMS_TOL, (rms, fs, N, params)
    for s_ in range(S):
        for k in ("period", "prevPeriod", "prevVoicedPeriod", "periodNew", "pitch", "beta", "anMarks", "stMarks", "gateOpen"):
            assert np.array_equal(states[1][s_][k], states[0][s_][k]), (seed, s_, k)


def test_no_read_of_unwritten_lds():
    """LDS is not cleared between kernels: a read of a location the launch has not written returns what the previous
    kernel on that CU left there -- usually a harmless finite number, once in a while a NaN (this is how a 16th history
    tap with a zero coefficient made the FAST IIR emit NaNs in one run out of five).  The -DVP_POISON_LDS build fills
    the dynamic LDS with NaNs at the top of every kernel; the whole GPU suite must pass on it."""
    import subprocess
    import sys
    from vocoderproject_amd import build
    if not os.path.exists(build.LIB_POISON):
        pytest.skip("libvp_amd_poison.so not built (python -m vocoderproject_amd.build --poison)")
    if os.environ.get("VP_AMD_LIB"):
        pytest.skip("already running on a diagnostic library")
    env = dict(os.environ, VP_AMD_LIB=build.LIB_POISON)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.dirname(os.path.abspath(__file__)), "-m", "gpu", "-q", "-x",
                        "-k", "not unwritten_lds", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_two_handles_of_different_geometry_coexist():
    """Per-function launch attributes (the dynamic-LDS ceiling) are process-wide: preparing a second, smaller handle must
    not break the first one."""
    from vocoderproject_amd import BatchVocoderProcessor
    big = BatchVocoderProcessor()
    big.prepareExplicit(48000.0, 2048, 2, 2048, 1536, 2048, 512)       # cfg5 geometry: ~150 KB of LDS per workgroup
    small = BatchVocoderProcessor()
    small.prepareToPlay(22050.0, 256, 2)
    xb = _streams(2, 2048 * 6, fs=48000.0)
    xs = _streams(2, 256 * 24, fs=22050.0)
    yb = big.run(xb)                                                     # launched AFTER the small handle was prepared
    ys = small.run(xs)
    _assert_equal(yb, _oracle_run(xb, 2048, dict(), prepare=(48000.0, 2048, 2048, 1536, 2048, 512)), "big")
    ref = []
    from oracle import oracle_py as O
    for s_ in range(2):
        o = O.OracleStream()
        o.prepare_to_play(22050.0, 256)
        ref.append(o.run(xs[s_]))
    _assert_equal(ys, np.stack(ref), "small")


@pytest.mark.parametrize("order", [2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 14, 15, 16, 17, 24])
def test_exact_iir_every_small_order(order):
    """The EXACT pitch-path recursion has one instantiation per order up to 16 (DPP-row form) and the general form above:
    bit-exact against the oracle for each, at two block sizes (chunk boundaries fall differently)."""
    from vocoderproject_amd import BatchVocoderProcessor
    for N in (1024, 300):
        S, B = 3, 12 if N == 1024 else 40
        x = _streams(S, N * B)
        params = dict(lpcPitch=order, vocBool=0)
        p = BatchVocoderProcessor(**params)
        p.prepareToPlay(FS, N, S)
        _assert_equal(p.run(x), _oracle_run(x, N, params), f"lpcPitch={order} N={N}")


# ---- extension: fixed pitch-shift interval (BASELINE configs[1] "+-12-semitone pitch shift") -----------------------------

@pytest.mark.parametrize("mode", ["pitch", "both"])
def test_fixed_pitch_shift_extension_bit_exact(mode):
    """vp_set_pitch_shift: beta = 2^(semitones/12) in placeStMarks instead of the key's nearest note, per stream, set
    before and between blocks.  No reference counterpart: the bar is GPU == oracle (the oracle's own version of the
    extension is checked musically in tests/test_oracle_behaviour.py), bit-exact like everything else."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    S, N, B = 7, 1024, 26
    x = np.concatenate([_streams(4, N * B), _edge_streams(N * B)[:3]])
    shifts = [12.0, -12.0, 7.0, -5.0, None, 0.37, -12.0]
    later = {9: (4, 3.0), 14: (0, None), 18: (6, 12.0)}
    params = dict(vocBool=0) if mode == "pitch" else dict()
    p = BatchVocoderProcessor(**params)
    with pytest.raises(VpError):
        p.setPitchShift(3.0)                                         # before prepare
    p.prepareToPlay(FS, N, S)
    with pytest.raises(VpError):
        p.setPitchShift(12.5)
    with pytest.raises(VpError):
        p.setPitchShift(1.0, stream=S)
    os_ = []
    for s_ in range(S):
        o = O.OracleStream(**params)
        o.prepare_to_play(FS, N)
        if shifts[s_] is not None:
            p.setPitchShift(shifts[s_], stream=s_)
            o.set_pitch_shift(shifts[s_])
        os_.append(o)
    assert p.getPitchShift(0) == (True, 12.0) and p.getPitchShift(4)[0] is False
    diff_from_plain = 0
    plain = _oracle_run(x, N, params)
    for b in range(B):
        if b in later:
            s_, v = later[b]
            p.setPitchShift(v if v is not None else 0.0, on=v is not None, stream=s_)
            os_[s_].set_pitch_shift(v if v is not None else 0.0, on=v is not None)
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        got = p.process(blk)
        for s_ in range(S):
            io = blk[s_].copy()
            os_[s_].process_block(io)
            _assert_equal(got[s_], io[:2], f"block {b} stream {s_} shift {shifts[s_]}")
            diff_from_plain += int(np.any(io[:2] != plain[s_][:, b * N:(b + 1) * N]))
    assert diff_from_plain > B                                       # the shift does something
    ub = np.sum([o.ub_counters() for o in os_], axis=0)
    assert list(p.ub_counters()) == list(ub)
    # the tracker holds the fixed factor
    st = p.pitch_state(1)
    if st["period"] > 0:
        assert st["beta"] == 0.5
    # a new prepare switches the extension off
    p.prepareToPlay(FS, N, S)
    assert p.getPitchShift(0)[0] is False


def test_fixed_pitch_shift_fast_modes_and_large_batch():
    """The shift through the arithmetic modes bench.py runs (FAST IIR, certified cross-correlation YIN) and through the
    register-light build for large batches: decisions identical to the exact default, samples within tolerance."""
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 260, 1024, 10                                           # > 256 streams: vp_k_pitch_lite*
    x = np.tile(_streams(13, N * B), (20, 1, 1))
    semis = [12.0 if s_ % 2 == 0 else -12.0 for s_ in range(S)]
    outs, states = {}, {}
    for tag, iir, yin in [("exact", "exact", "direct"), ("fast", "fast", "xcorr")]:
        p = BatchVocoderProcessor(vocBool=0)
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode(iir); p.set_yin_mode(yin)
        for s_ in range(S):
            p.setPitchShift(semis[s_], stream=s_)
        outs[tag] = p.run(x)
        states[tag] = [p.pitch_state(s_) for s_ in (0, 1, 14, 259)]
    err = outs["fast"].astype(np.float64) - outs["exact"]
    assert np.sqrt((err ** 2).mean()) < RMS_TOL
    for a, b_ in zip(states["exact"], states["fast"]):
        assert (a["period"], a["anMarks"], a["stMarks"], a["beta"]) == (b_["period"], b_["anMarks"], b_["stMarks"], b_["beta"])
    # streams 0 and 26 carry the same input and the same shift: batch position must not matter
    np.testing.assert_array_equal(outs["exact"][0], outs["exact"][26])
    # against the oracle on a sample of streams
    from oracle import oracle_py as O
    for s_ in (0, 1, 259):
        o = O.OracleStream(vocBool=0)
        o.prepare_to_play(FS, N)
        o.set_pitch_shift(semis[s_])
        _assert_equal(outs["exact"][s_], o.run(x[s_]), f"stream {s_}")


def test_fixed_pitch_shift_rejected_when_marks_would_not_fit():
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    p = BatchVocoderProcessor()
    p.prepareExplicit(48000.0, 2048, 2, 2048, 1536, 2048, 512)        # 2048 / round(60 / 2) + 2 = 70 marks > 64
    with pytest.raises(VpError):
        p.setPitchShift(12.0)
    p.setPitchShift(7.0)                                              # 2048 / 40 + 2 = 53: fine


# ---- offline (file-to-file) front end: SURVEY.md section 8f item 4 -------------------------------------------------------

def _oracle_offline(v, c, N, params, shift=None, latency=1024):
    """What offline.render must return for one recording: the oracle over the zero-padded recording, latency removed."""
    from oracle import oracle_py as O
    T = ((len(v) + latency + N - 1) // N) * N
    x = np.zeros((3, T), np.float32)
    x[0, :len(v)] = v
    if c is not None:
        x[1:3, :min(T, c.shape[-1])] = c[..., :T]
    o = O.OracleStream(**params)
    o.prepare_to_play(FS, N)
    if shift is not None:
        o.set_pitch_shift(shift)
    return o.run(x)[:, latency:latency + len(v)]


def test_offline_pitch_corrector_ragged_batch_matches_oracle():
    from vocoderproject_amd import offline
    N = 1024
    x = _streams(4, N * 20)
    voices = [x[0, 0, :20000], x[1, 0, :7777], x[2, 0, :N * 20], x[3, 0, :1500]]
    shift = [None, 12.0, -7.0, None]
    outs = offline.pitch_corrector(voices, FS, key=5, shift=shift, blocks_per_call=8)
    for s, (v, o) in enumerate(zip(voices, outs)):
        ref = _oracle_offline(v, None, N, dict(vocBool=0, keyPitch=5), shift[s])
        assert o.shape == (len(v),)
        _assert_equal(o, ref[0], f"recording {s}")
    assert np.abs(outs[0]).max() > 0.05
    # block-by-block calls give the same files as eight blocks per call
    outs1 = offline.pitch_corrector(voices, FS, key=5, shift=shift, blocks_per_call=1)
    for a, b in zip(outs, outs1):
        np.testing.assert_array_equal(a, b)


def test_offline_vocode_and_command_line(tmp_path):
    import subprocess
    import sys
    from vocoderproject_amd import offline
    N = 1024
    x = _streams(2, N * 12)
    f_v = [str(tmp_path / "v0.wav"), str(tmp_path / "v1.wav")]
    f_c = str(tmp_path / "carrier.wav")
    offline.write_wav(f_v[0], 44100, x[0, 0, :11000])
    offline.write_wav(f_v[1], 44100, x[1, 0, :6000])
    offline.write_wav(f_c, 44100, x[0, 1:3, :12000])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "vocoderproject_amd.offline", "vocode", *f_v, "--carrier", f_c,
                        "--out-dir", str(tmp_path / "out"), "--lpc-voice", "24"], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    _, car = offline.read_wav(f_c)
    for s in range(2):
        fs, v = offline.read_wav(f_v[s])
        fs2, got = offline.read_wav(str(tmp_path / "out" / f"v{s}_vocode.wav"))
        assert fs == fs2 == 44100 and got.shape == (2, v.shape[1])
        ref = _oracle_offline(v[0], car, N, dict(pitchBool=0, lpcVoice=24, lpcSynth=5, keyPitch=12))
        # the file holds the output rounded to 16-bit PCM
        q = np.rint(np.clip(ref.astype(np.float64), -1, 1) * 32767.0) / 32767.0
        assert np.abs(got - q.astype(np.float32)).max() <= 1e-7
        assert np.abs(ref).max() > 0.02


# ---- buffers without the side-chain bus (MyBuffer.cpp:93-102: null pointers -> zeros) -----------------------------------

@pytest.mark.parametrize("params", [dict(), dict(vocBool=0), dict(gainSynth=-10.0, gainVoice=-8.0)])
def test_mono_entry_matches_null_sidechain(params):
    """vp_process_block_mono: voice only, the synth ring takes zeros like fillInputBuffers with null side-chain pointers;
    mixed with three-channel calls so that the ring goes through 'holds carrier', 'partly zeroed' and 'all zero'."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 5, 1024, 22
    x = _streams(S, N * B)
    kinds = "33333mmmmmmm333mmmmm3m"                                  # 3 = three channels, m = mono
    assert len(kinds) == B
    p = BatchVocoderProcessor(**params)
    p.prepareToPlay(FS, N, S)
    os_ = []
    for s_ in range(S):
        o = O.OracleStream(**params)
        o.prepare_to_play(FS, N)
        os_.append(o)
    nonzero = 0
    for b in range(B):
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        if kinds[b] == "3":
            got = p.process(blk)
        else:
            got = p.process_mono(np.ascontiguousarray(blk[:, 0]))
        for s_ in range(S):
            if kinds[b] == "3":
                io = blk[s_].copy()
                os_[s_].process_block(io)
                ref = io[:2]
            else:
                ref = os_[s_].process_block_mono(np.ascontiguousarray(blk[s_, 0]))
            _assert_equal(got[s_], ref, f"block {b} ({kinds[b]}) stream {s_}")
        nonzero += int(np.abs(got).max() > 0.01)
    assert nonzero > B // 2


def test_mono_entry_device_and_multi_block_forms():
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 6, 1024, 16
    x = _streams(S, N * B)
    v = np.ascontiguousarray(x[:, 0].reshape(S, B, N).transpose(1, 0, 2))            # [B][S][N]
    x0 = x.copy()
    x0[:, 1:] = 0
    ref, _ = _gpu_run(x0, N, dict(vocBool=0))                                       # three channels, zeroed side chain
    dev = torch.device("cuda", 0)
    dv = torch.from_numpy(v).to(dev)
    # block by block on the device
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(FS, N, S)
    dout = torch.empty((S, 2, N), dtype=torch.float32, device=dev)
    for b in range(B):
        p.process_mono_device(dv[b], dout)
        torch.cuda.synchronize()
        _assert_equal(dout.cpu().numpy(), ref[:, :, b * N:(b + 1) * N], f"mono device block {b}")
    # eight blocks per call (one launch)
    p2 = BatchVocoderProcessor(vocBool=0)
    p2.prepareToPlay(FS, N, S)
    dout8 = torch.empty((8, S, 2, N), dtype=torch.float32, device=dev)
    for b0 in range(0, B, 8):
        p2.process_blocks_mono_device(dv[b0:b0 + 8], dout8)
        torch.cuda.synchronize()
        y = dout8.cpu().numpy()
        for j in range(8):
            _assert_equal(y[j], ref[:, :, (b0 + j) * N:(b0 + j + 1) * N], f"mono multi block {b0 + j}")


@pytest.mark.parametrize("seed", list(range(10)))
def test_randomised_configurations_with_extensions_bit_exact(seed):
    """The randomised sweep again with this round's additions switched on at random: per-stream fixed pitch shifts
    (set, changed and switched off between blocks), mono and three-channel calls mixed, block by block against one
    oracle per stream.  (The vocoder's several-wavefronts-per-window launch comes in through the small block sizes.)"""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    fs, N, params = _fuzz_case(2000 + seed)
    rng = np.random.default_rng(7000 + seed)
    S = 4
    T = max(8, int(24000 * fs / 44100.0) // N) * N
    x = _streams(S, T, fs=fs)
    p = BatchVocoderProcessor(**params)
    try:
        p.prepareToPlay(fs, N, S)
    except VpError as e:
        assert e.code == -4, e
        pytest.skip(f"geometry for fs={fs} exceeds the LDS budget (VP_ERR_GEOMETRY)")
    os_ = []
    for s_ in range(S):
        o = O.OracleStream(**params)
        o.prepare_to_play(fs, N)
        os_.append(o)
    nb = T // N
    for b in range(nb):
        if rng.random() < 0.25:                                      # a shift event on a random stream
            s_ = int(rng.integers(0, S))
            on = rng.random() < 0.8
            semi = float(rng.choice([-12.0, -7.0, -2.5, 0.0, 3.0, 5.0, 12.0]))
            try:
                p.setPitchShift(semi, on=on, stream=s_)
            except VpError as e:
                assert e.code == -4, e                               # more synthesis marks than the arrays hold: refused
            else:
                os_[s_].set_pitch_shift(semi, on=on)
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        mono = rng.random() < 0.4
        got = p.process_mono(np.ascontiguousarray(blk[:, 0])) if mono else p.process(blk)
        for s_ in range(S):
            if mono:
                ref = os_[s_].process_block_mono(np.ascontiguousarray(blk[s_, 0]))
            else:
                io = blk[s_].copy()
                os_[s_].process_block(io)
                ref = io[:2]
            _assert_equal(got[s_], ref, f"seed {seed}: fs={fs} N={N} {params} block {b} mono={mono} stream {s_}")
    ub = np.sum([o.ub_counters() for o in os_], axis=0)
    assert list(p.ub_counters()) == list(ub)


@pytest.mark.parametrize("params,N", [(dict(), 1024), (dict(pitchBool=0, lpcVoice=64, lpcSynth=30), 1024), (dict(pitchBool=0), 300)])
def test_register_light_vocoder_for_large_batches(params, N):
    """Above 256 streams in FAST IIR mode the host launches vp_k_vocoder_lite: half the window slots, two wavefronts per
    slot, two workgroups per CU.  Same filters as the regular FAST vocoder (whose recursion and Levinson-Durbin take other,
    register-heavier forms since round 2: rounding-level differences): the streams must come out as a small batch (regular
    build) gives them up to last-bit flips of the float32 cast, and within the stated tolerance of the oracle."""
    from vocoderproject_amd import BatchVocoderProcessor
    S, B = 300, 8
    x = _streams(S, N * B)

    def run(xs):
        p = BatchVocoderProcessor(**params)
        p.prepareToPlay(FS, N, xs.shape[0])
        p.set_vocoder_path("workgroup")                               # (the default above 256 streams is the lane-per-window pipeline)
        # (the regular build comes in two register budgets: vp_k_vocoder for orders up to 32 and above 48, vp_k_vocoder_o48 between)
        full = "vp_k_vocoder_o48" if 32 < params.get("lpcVoice", 40) <= 48 else "vp_k_vocoder"
        assert p.vocoder_kernel_name() == full                        # exact IIR: always a regular build
        p.set_iir_mode("fast")
        assert p.vocoder_kernel_name() == ("vp_k_vocoder_lite" if xs.shape[0] > 256 else full)
        return p.run(xs)

    big = run(x)
    pick = [0, 7, 255, 256, 299]
    small = run(np.ascontiguousarray(x[pick]))                       # <= 256 streams: vp_k_vocoder
    dlt = np.abs(big[pick].astype(np.float64) - small)
    assert dlt.max() <= 4e-7 * max(1.0, float(np.abs(small).max())) and (big[pick] != small).mean() < 0.02, "lite vs regular FAST vocoder"
    ref = _oracle_run(x[pick], N, params)
    err = big[pick].astype(np.float64) - ref
    assert np.sqrt((err ** 2).mean()) < RMS_TOL
    assert np.abs(ref).max() > 0.02


@pytest.mark.parametrize("S,iir,voc", [(1024, "exact", "workgroup"), (1024, "fast", "workgroup"), (4096, "fast", "workgroup"),
                                       (1024, "exact", "auto"), (1024, "fast", "auto"), (4096, "fast", "auto")])
def test_config3_scale_batches(S, iir, voc):
    """BASELINE configs[3] per GPU (1024 streams, pitch corrector + vocoder) and four times that: sampled streams against
    the oracle (bit-exact in exact mode, within tolerance in FAST mode, where the large-batch builds of both kernels
    run), and a size-independent property at full size: permuting the streams of the batch permutes the output."""
    from vocoderproject_amd import BatchVocoderProcessor
    N, B, U = 1024, 5, 16
    base = _streams(U, N * B)
    idx = np.arange(S) % U
    x = np.ascontiguousarray(base[idx])

    def run(xs):
        p = BatchVocoderProcessor()
        p.prepareToPlay(FS, N, xs.shape[0])
        p.set_iir_mode(iir)
        p.set_vocoder_path(voc)                  # "workgroup": the large-batch builds of vp_k_vocoder; "auto": the lane-per-window pipeline here
        if voc == "auto":
            assert p.vocoder_kernel_name() == "vp_k_v2_pipeline"
        elif iir == "fast":
            assert p.vocoder_kernel_name() == "vp_k_vocoder_lite" and p.pitch_kernel_name().startswith("vp_k_pitch_lite")
        return p.run(xs)

    got = run(x)
    ref = _oracle_run(base, N, {})
    pick = [0, 1, U + 3, S // 2 + 5, S - 1]
    if iir == "exact":
        for s_ in pick:
            _assert_equal(got[s_], ref[idx[s_]], f"stream {s_}")
        # every copy of a stream comes out the same, wherever it sits in the batch
        for u in range(U):
            assert np.all(got[u::U] == got[u]), u
    else:
        err = got[pick].astype(np.float64) - ref[idx[pick]]
        assert np.sqrt((err ** 2).mean()) < RMS_TOL
        for u in range(U):
            assert np.all(got[u::U] == got[u]), u
    perm = np.random.default_rng(S).permutation(S)
    got_p = run(np.ascontiguousarray(x[perm]))
    np.testing.assert_array_equal(got_p, got[perm])
