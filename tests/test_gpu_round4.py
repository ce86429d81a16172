"""GPU tests added in round 4 (through the C ABI): every workload bench.py times, in the mode, on the path and at the batch size
it is timed in, against the CPU oracle; the reference-derived fixtures fed straight through libvp_amd.so; the allocation
contract of the process calls; the fused STFT kernel and its phase-vocoder stage."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 44100.0
RMS_TOL = 1e-4          # BASELINE.json north_star: "per-sample RMS error < 1e-4 vs reference"


def _streams(S, T, **kw):
    from vocoderproject_amd.synth import make_streams
    return np.ascontiguousarray(make_streams(S, T, **kw).numpy())


def _oracle(x, N, params, prepare=None, fs=FS, mono=False):
    from oracle import oracle_py as O
    outs = []
    for s in range(x.shape[0]):
        o = O.OracleStream(**params)
        if prepare:
            o.prepare_explicit(fs, N, *prepare)
        else:
            o.prepare_to_play(fs, N)
        if mono:
            outs.append(np.concatenate([o.process_block_mono(np.ascontiguousarray(x[s, 0, b * N:(b + 1) * N]))
                                        for b in range(x.shape[2] // N)], axis=1))
        else:
            outs.append(o.run(x[s]))
    return np.stack(outs)


# ---- BASELINE configs[4] as bench.py times it: FAST IIR + the lane-per-window pipeline, 48 kHz 2048/512, orders 48/48/30, 512 streams ----

def test_config4_as_benched_fast_pipeline_512_streams():
    """round-3 verdict, weak item 1: `vp_k_v2_autocorr<4,true>`, `vp_k_v2_fir2<48,32,true>`, `vp_k_v2_levinson2<48,32,true>`,
    `vp_k_v2_iir_fast` at W = 2048 and `vp_k_pitch_fast` at order 48 -- the instantiations the configs4 leg of bench.py launches --
    against the oracle on sampled streams (north_star tolerance), plus the size-independent property at full size: permuting
    the streams of the batch permutes the output, and every copy of a stream comes out the same wherever it sits."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B, U, fs = 512, 2048, 5, 16, 48000.0
    prepare = (2048, 1536, 2048, 512)
    params = dict(lpcVoice=48, lpcPitch=48, lpcSynth=30)
    base = _streams(U, N * B, fs=fs)
    idx = np.arange(S) % U
    x = np.ascontiguousarray(base[idx])

    def run(xs):
        p = BatchVocoderProcessor(**params)
        p.prepareExplicit(fs, N, xs.shape[0], *prepare)
        p.set_iir_mode("fast")
        p.set_yin_mode("xcorr")
        assert p.vocoder_kernel_name() == "vp_k_v2_pipeline" and p.pitch_kernel_name() == "vp_k_pitch_fast"     # what BENCH_r03.configs4 names
        xd = torch.from_numpy(xs).cuda().view(xs.shape[0], 3, B, N).permute(2, 0, 1, 3).contiguous()
        yd = torch.empty((B, xs.shape[0], 2, N), dtype=torch.float32, device="cuda")
        for b in range(B):
            p.process_device(xd[b], yd[b])                # the device entry bench.py times
        torch.cuda.synchronize()
        st = [p.pitch_state(s) for s in (0, 1, U + 3, S - 1)]
        return yd.permute(1, 2, 0, 3).reshape(xs.shape[0], 2, B * N).cpu().numpy(), st

    got, st = run(x)
    ref = _oracle(base, N, params, prepare=prepare, fs=fs)
    pick = [0, 1, U + 3, S // 2 + 5, S - 1, 77, 300, 411]
    err = got[pick].astype(np.float64) - ref[idx[pick]]
    rms = float(np.sqrt((err ** 2).mean()))
    scale = max(1.0, float(np.abs(ref).max()))
    print(f"configs[4] as benched: rms err {rms:.3e}, max abs {np.abs(err).max():.3e}, ref rms {np.sqrt((ref.astype(np.float64) ** 2).mean()):.3f}")
    assert rms < RMS_TOL, rms
    assert np.abs(err).max() <= 2e-6 * scale, (np.abs(err).max(), scale)
    assert np.abs(ref).max() > 0.05
    for u in range(U):
        assert np.all(got[u::U] == got[u]), u
    # every decision is the oracle's (the tracker state after the last block)
    from oracle import oracle_py as O
    for s_, gst in zip((0, 1, U + 3, S - 1), st):
        o = O.OracleStream(vocBool=0, **params)
        o.prepare_explicit(fs, N, *prepare)
        _, tr = o.run(base[idx[s_]], trace=True)
        f = tr[-1]
        assert not f["gated"]
        assert (gst["period"], gst["anMarks"], gst["stMarks"], gst["beta"]) == (f["period"], f["anMarks"], f["stMarks"], f["beta"]), s_
    perm = np.random.default_rng(4).permutation(S)
    got_p, _ = run(np.ascontiguousarray(x[perm]))
    np.testing.assert_array_equal(got_p, got[perm])


# ---- BASELINE configs[1] as bench.py times it: 256 mono streams, FAST IIR, certified cross-correlation YIN (vp_k_pitch_fast_c) ----

@pytest.mark.parametrize("iir", ["fast", "exact"])
def test_config1_as_benched_256_mono_streams(iir):
    """round-3 verdict, weak item 2: the headline workload at its full size and through the entry point that is timed
    (vp_process_block_mono_device, S = 256), sampled streams against the oracle: within the north_star tolerance in the
    tolerance mode (and within the few-ulp bound the small-batch tests hold it to), bit for bit in the exact mode; decisions
    identical in both."""
    import torch
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 256, 1024, 12
    x = _streams(S, N * B)
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(FS, N, S)
    p.set_iir_mode(iir)
    p.set_yin_mode("xcorr")
    assert p.pitch_kernel_name() == ("vp_k_pitch_ws" if iir == "fast" else "vp_k_pitch_ws_x")     # (round 5: what bench.py's headline launches)
    xm = torch.from_numpy(np.ascontiguousarray(x[:, 0])).cuda().view(S, B, N).permute(1, 0, 2).contiguous()
    yd = torch.empty((B, S, 2, N), dtype=torch.float32, device="cuda")
    pick = [0, 1, 31, 63, 64, 100, 127, 128, 190, 200, 254, 255]
    orc = []
    for s in pick:
        o = O.OracleStream(vocBool=0)
        o.prepare_to_play(FS, N)
        orc.append(o)
    ref = np.empty((len(pick), 2, N * B), np.float32)
    frames = 0
    for b in range(B):
        p.process_mono_device(xm[b], yd[b])
        for i, s in enumerate(pick):
            ref[i, :, b * N:(b + 1) * N] = orc[i].process_block_mono(np.ascontiguousarray(x[s, 0, b * N:(b + 1) * N]))
            tr = orc[i].traces()
            if tr and not tr[-1]["gated"]:
                st, f = p.pitch_state(s), tr[-1]
                frames += 1
                assert (st["period"], st["anMarks"], st["stMarks"], st["beta"]) == (f["period"], f["anMarks"], f["stMarks"], f["beta"]), (s, b)
    torch.cuda.synchronize()
    got = yd.permute(1, 2, 0, 3).reshape(S, 2, B * N).cpu().numpy()[pick]
    assert frames > 100
    if iir == "exact":
        np.testing.assert_array_equal(got, ref)
    else:
        err = got.astype(np.float64) - ref
        rms = float(np.sqrt((err ** 2).mean()))
        print(f"configs[1] as benched: rms err {rms:.3e}, max abs {np.abs(err).max():.3e}")
        assert rms < RMS_TOL, rms
        assert np.abs(err).max() <= 4e-7 * max(1.0, float(np.abs(ref).max()))
        assert (got != ref).mean() < 0.02
    assert np.abs(ref).max() > 0.05
    cert, fallback = p.yin_certified_counts()
    assert cert > 3 * max(fallback, 1)                   # the certified form decided most frames of the batch (start-up frames fall back)


# ---- the reference-derived fixtures, straight through libvp_amd.so ----------------------------------------------------------

def test_notebook_pitch_corrector_recordings_through_the_hip_path(golden_dir, capsys):
    """round-3 verdict, weak item 3: the six recordings of tests/golden/pitch_corrector_vectors.npz (inputs, outputs and per-frame
    pitch / marks produced by running the reference notebook's own pitch_corrector loop) as ONE batch of six streams, each with
    its own key, through libvp_amd.so -- held to the agreement table tests/test_oracle_golden.py holds the CPU oracle to
    (tests/_agreement.py: same thresholds).  The tracker is read after every block of 256 samples in which a frame started."""
    from _agreement import KEYS, agreement_row, print_report
    from vocoderproject_amd import BatchVocoderProcessor
    G3 = np.load(os.path.join(golden_dir, "pitch_corrector_vectors.npz"))
    names = [str(n) for n in G3["names"]]
    S, T = len(names), len(G3[f"{names[0]}_x"])
    x = np.zeros((S, 3, T), np.float32)
    for i, n in enumerate(names):
        assert len(G3[f"{n}_x"]) == T
        x[i, 0] = G3[f"{n}_x"]

    def make(N):
        p = BatchVocoderProcessor(vocBool=0)
        p.prepareToPlay(FS, N, S)
        for i, n in enumerate(names):
            p.setStreamParameter(i, "keyPitch", KEYS.index(str(G3[f"{n}_key"])))
        return p

    p = make(256)
    y = np.empty((S, 2, T), np.float32)
    tr = [[] for _ in range(S)]
    for b in range(T // 256):
        y[:, :, b * 256:(b + 1) * 256] = p.process(np.ascontiguousarray(x[:, :, b * 256:(b + 1) * 256]))
        if b % 3 == 0:                               # PitchProcess.cpp:171-189: a frame starts every third chunk step
            for i in range(S):
                st = p.pitch_state(i)
                tr[i].append(dict(pitch=st["pitch"], anMarks=st["anMarks"], stMarks=st["stMarks"]))
    # the host block size does not matter while the gate stays open: the plugin's N = 1024 gives the same samples
    y1024 = make(1024).run(x)
    np.testing.assert_array_equal(y1024, y)
    report = [agreement_row(G3, n, y[i, 0].astype(np.float64), tr[i]) for i, n in enumerate(names)]
    with capsys.disabled():
        print_report(report, "libvp_amd.so")
    # and the oracle says the same, bit for bit
    from oracle import oracle_py as O
    for i, n in enumerate(names):
        o = O.OracleStream(vocBool=0, keyPitch=KEYS.index(str(G3[f"{n}_key"])))
        o.prepare_to_play(FS, 1024)
        np.testing.assert_array_equal(o.run(x[i]), y[i])


# ---- the boundary's allocation contract (include/vp_amd.h: "No allocation happens in any vp_process_*() call") ---------------------

@pytest.mark.parametrize("S,voc", [(6, "auto"), (300, "auto"), (6, "batched")])
def test_no_device_allocation_inside_any_process_call(S, voc):
    """round-3 verdict, weak item 9: prepare and vp_reserve_blocks are the only allocation sites.  Drives every process entry point
    (single, in place, mono, device, multi-block device / mono / host; all three switch settings; both IIR modes, so that the
    vocoder-only and the combined multi-block plans of the pipeline run) through the raw C ABI and watches the handle's
    allocation count.  Without a reservation the multi-block calls must still work (block by block) and give the same samples
    in the exact mode."""
    import ctypes as C
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    N, B = 512, 6
    x = _streams(min(S, 6), N * B * 2)
    x = np.ascontiguousarray(x[np.arange(S) % x.shape[0]])
    xb = np.ascontiguousarray(x.reshape(S, 3, 2 * B, N).transpose(2, 0, 1, 3))                  # [2B][S][3][N]
    xd = torch.from_numpy(xb).cuda()
    xm = xd[:, :, 0, :].contiguous()
    outs = {}
    for reserve in (0, B):
        for iir in ("exact", "fast"):
            p = BatchVocoderProcessor()
            p.prepareToPlay(FS, N, S)
            p.set_iir_mode(iir)
            p.set_vocoder_path(voc)
            if reserve:
                p.reserve_blocks(reserve)
                assert p.L.vp_get_reserved_blocks(p.h) == reserve
            n0 = p.alloc_count()
            assert n0 > 10
            L, h = p.L, p.h
            yd = torch.empty((B, S, 2, N), dtype=torch.float32, device="cuda")
            yh = np.empty((B, S, 2, N), np.float32)
            got = []
            for sw in (dict(pitchBool=1, vocBool=1), dict(pitchBool=0, vocBool=1), dict(pitchBool=1, vocBool=0)):
                for k, v in sw.items():
                    p.setParameter(k, v)
                assert L.vp_process_blocks_device(h, xd[:B].data_ptr(), yd.data_ptr(), B, None) == 0
                torch.cuda.synchronize()
                got.append(yd.cpu().numpy().copy())
                assert L.vp_process_blocks(h, xb[B:].ctypes.data, yh.ctypes.data, B) == 0
                got.append(yh.copy())
                assert L.vp_process_blocks_mono_device(h, xm[:B].data_ptr(), yd.data_ptr(), B, None) == 0
                assert L.vp_process_block_device(h, xd[0].data_ptr(), yd[0].data_ptr(), None) == 0
                assert L.vp_process_block_mono_device(h, xm[1].data_ptr(), yd[1].data_ptr(), None) == 0
                torch.cuda.synchronize()
                io = xb[2].copy()
                assert L.vp_process_block_inplace(h, io.ctypes.data) == 0
                assert L.vp_process_block(h, xb[3].ctypes.data, yh[0].ctypes.data) == 0
                assert L.vp_process_block_mono(h, np.ascontiguousarray(xb[4][:, 0]).ctypes.data, yh[1].ctypes.data) == 0
                got.append(yh[:2].copy())
            assert p.alloc_count() == n0, (reserve, iir, p.alloc_count(), n0)
            outs[(reserve, iir)] = got
    # exact mode: the same samples with and without the reservation (the plans that need scratch only change HOW the blocks are issued)
    for a, b in zip(outs[(0, "exact")], outs[(B, "exact")]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(outs[(0, "fast")], outs[(B, "fast")]):
        assert np.abs(a.astype(np.float64) - b).max() <= 2e-6 * max(1.0, float(np.abs(a).max()))


# ---- the fused STFT round trip (csrc/vp_stft.hip) and its phase-vocoder stage: no reference counterpart, checked against NumPy ---------

def _stft_run(x, F, hop, runs=0, mag=False, semitones=None, precision="f64"):
    import torch
    from vocoderproject_amd import StftRoundTrip
    S, T = x.shape
    st = StftRoundTrip(S, T, F, hop)
    st.set_runs(runs)
    st.set_precision(precision)
    xd = torch.from_numpy(x).cuda()
    yd = torch.full_like(xd, float("nan"))                       # every output sample must be written
    md = torch.empty((S, st.n_frames, F // 2 + 1), dtype=torch.float32, device="cuda") if mag else None
    if semitones is None:
        st(xd, yd, md)
    else:
        st.pitch_shift(xd, yd, semitones)
    torch.cuda.synchronize()
    return yd.cpu().numpy(), (md.cpu().numpy() if mag else None), st


@pytest.mark.parametrize("F,hop,T", [(1024, 256, 1024 * 24), (1024, 256, 1024 * 9 + 300), (1024, 512, 1024 * 12), (1024, 128, 1024 * 6 + 128), (1024, 256, 4097),
                                     (1024, 64, 5000), (2048, 512, 2048 * 12), (2048, 512, 2048 * 5 + 1234), (2048, 1024, 2048 * 6), (2048, 256, 9001)])
def test_fused_stft_round_trip_against_numpy(F, hop, T):
    """Whole output (edges included) against tests/stft_reference.py, the magnitude spectrum against numpy.fft.rfft, every sample
    written exactly once (NaN prefill), odd lengths (the unaligned load path), overlap factors 2 / 4 / 8 / 16 -- and the partition of a
    stream into runs of frames must not change a single bit (a run recomputes the frames in front of it in the same order)."""
    import stft_reference as R
    S = 5
    x = _streams(S, T)[:, 0].copy()
    x[1] *= 3.0
    y, mag, st = _stft_run(x, F, hop, mag=True)
    assert st.fused
    assert not np.isnan(y).any()
    for s in range(S):
        ref = R.stft_roundtrip(x[s], F, hop)
        np.testing.assert_allclose(y[s], ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))
    nF = st.n_frames
    np.testing.assert_allclose(y[:, F:(nF - 1) * hop], x[:, F:(nF - 1) * hop], rtol=0, atol=1e-6)      # interior: perfect reconstruction
    w = R.window(F)
    for s in (0, S - 1):
        for f in (0, min(7, nF - 1), nF - 1):
            ref = np.abs(np.fft.rfft(x[s, f * hop:f * hop + F].astype(np.float64) * w))
            np.testing.assert_allclose(mag[s, f], ref, rtol=1e-6, atol=1e-6)
    for runs in (1, 2, 3, 7):
        y2, _, _ = _stft_run(x, F, hop, runs=runs)
        np.testing.assert_array_equal(y2, y, err_msg=f"runs={runs}")


@pytest.mark.parametrize("F,hop,T", [(1024, 256, 1024 * 24), (1024, 256, 1024 * 9 + 300), (1024, 512, 1024 * 12), (1024, 128, 1024 * 6 + 128), (1024, 256, 4097),
                                     (1024, 64, 5000), (2048, 512, 2048 * 12), (2048, 512, 2048 * 5 + 1234), (2048, 1024, 2048 * 6), (2048, 256, 9001)])
def test_single_precision_stft_round_trip_against_numpy(F, hop, T):
    """vp_stft_set_precision(VP_STFT_F32): the same kernel with transform, split and merge in f32 (vp_k_stft_fused32).  Same checks as the
    default build's, with the tolerance single precision earns: the whole output within 2e-6 of the NumPy restatement's scale (the
    north_star's bound is 1e-4 RMS; measured rms ~1e-7), magnitudes to 2e-5 relative, every sample written once, and -- being the
    same deterministic order of additions -- bit-identical across partitions into runs."""
    import stft_reference as R
    S = 5
    x = _streams(S, T)[:, 0].copy()
    x[1] *= 3.0
    y, mag, st = _stft_run(x, F, hop, mag=True, precision="f32")
    assert st.precision == "f32" and not np.isnan(y).any()
    worst = 0.0
    for s in range(S):
        ref = R.stft_roundtrip(x[s], F, hop)
        np.testing.assert_allclose(y[s], ref, rtol=0, atol=2e-6 * max(1.0, np.abs(ref).max()))
        worst = max(worst, float(np.sqrt(((y[s] - ref) ** 2).mean()) / max(1e-30, np.sqrt((ref ** 2).mean()))))
    print(f"f32 STFT {F}/{hop} T {T}: worst relative rms error {worst:.2e}")
    assert worst < 1e-6
    nF = st.n_frames
    w = R.window(F)
    for s in (0, S - 1):
        for f in (0, min(7, nF - 1), nF - 1):
            ref = np.abs(np.fft.rfft(x[s, f * hop:f * hop + F].astype(np.float64) * w))
            np.testing.assert_allclose(mag[s, f], ref, rtol=2e-5, atol=2e-5 * max(1.0, ref.max()))
    for runs in (1, 2, 3, 7):
        y2, _, _ = _stft_run(x, F, hop, runs=runs, precision="f32")
        np.testing.assert_array_equal(y2, y, err_msg=f"runs={runs}")
    y64, _, _ = _stft_run(x, F, hop)                                   # and the default is still the double-precision kernel
    assert np.abs(y64 - y).max() < 4e-6 and not np.array_equal(y64, y)


def test_stft_precision_switch_through_the_c_abi():
    """vp_stft_set_precision / vp_stft_get_precision: argument checking, default, and that the phase-vocoder stage ignores the switch (its
    phases accumulate over the whole stream: double in either setting -- same bits)."""
    import ctypes as C
    import torch
    from vocoderproject_amd import StftRoundTrip
    st = StftRoundTrip(3, 1024 * 10, 1024, 256)
    L = st.L
    assert L.vp_stft_get_precision(st.h) == 0 and st.precision == "f64"                    # VP_STFT_F64 is the default
    assert L.vp_stft_set_precision(st.h, 2) != 0 and L.vp_stft_set_precision(None, 1) != 0  # invalid value / null handle
    assert L.vp_stft_get_precision(st.h) == 0
    x = torch.randn((3, 1024 * 10), dtype=torch.float32, device="cuda") * 0.1
    y64, y32 = torch.empty_like(x), torch.empty_like(x)
    st.pitch_shift(x, y64, 5.0)
    st.set_precision("f32")
    assert L.vp_stft_get_precision(st.h) == 1
    st.pitch_shift(x, y32, 5.0)
    torch.cuda.synchronize()
    assert torch.equal(y64, y32)
    st.close()


def test_stft_rejects_frame_lengths_the_fused_kernel_is_not_built_for():
    from vocoderproject_amd import StftRoundTrip, VpError
    for F, hop in ((512, 128), (4096, 1024), (1024, 1024), (1024, 48), (2048, 96)):
        with pytest.raises(VpError) as e:
            StftRoundTrip(2, 8192, F, hop)
        assert e.value.code == -4                                    # VP_ERR_GEOMETRY


@pytest.mark.parametrize("semitones", [3.0, -5.0, 12.0, -12.0, 0.0])
def test_phase_vocoder_stage_against_the_numpy_restatement(semitones):
    """The north_star's "per-bin phase unwrap/accumulate" stage (vp_stft_pitch_shift): PARITY UNPINNED -- the reference has no phase
    vocoder; the checker is the build's own NumPy restatement (tests/stft_reference.py).  Both compute in double; libm differences
    (atan2, sincos) are at the 1e-15 level, a wrap or bin-rounding decision on an exact tie could differ, hence an rms bound and a
    bound on the fraction of samples beyond a tight absolute one.  Also checks what the stage is FOR: a steady tone comes out at
    ratio times its frequency."""
    import stft_reference as R
    F, hop, S, T = 1024, 256, 4, 1024 * 20
    ratio = 2.0 ** (semitones / 12.0)
    x = _streams(S, T)[:, 0].copy()
    t = np.arange(T) / FS
    x[3] = (0.4 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    y, _, st = _stft_run(x, F, hop, semitones=semitones)
    assert st.fused and not np.isnan(y).any()
    ref = np.stack([R.stft_roundtrip(x[s], F, hop, ratio=ratio) for s in range(S)])
    err = y.astype(np.float64) - ref
    rms, rel = float(np.sqrt((err ** 2).mean())), float(np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean()))
    print(f"phase vocoder {semitones:+.0f} st: rms err {rms:.3e} (relative {rel:.3e}), max abs {np.abs(err).max():.3e}, out rms {np.sqrt((ref ** 2).mean()):.3f}")
    assert rms < RMS_TOL, rms
    assert (np.abs(err) > 1e-5).mean() < 1e-3
    assert np.sqrt((ref ** 2).mean()) > 0.02
    # the tone: spectral peak of the interior of the output at ratio * 440 Hz (within a bin of a long transform)
    seg = y[3, 4 * F:4 * F + 8192].astype(np.float64) * np.hanning(8192)
    peak = np.argmax(np.abs(np.fft.rfft(seg))) * FS / 8192
    assert abs(peak - 440.0 * ratio) < 2.0 * FS / 8192 + 0.01 * 440.0 * ratio, (peak, 440.0 * ratio)
