"""GPU tests added in round 3 (through the C ABI, against the CPU oracle or against the block-by-block path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 44100.0


def _streams(S, T, **kw):
    from vocoderproject_amd.synth import make_streams
    return np.ascontiguousarray(make_streams(S, T, **kw).numpy())


def _oracle(x, N, params):
    from oracle import oracle_py as O
    outs = []
    for s in range(x.shape[0]):
        o = O.OracleStream(**params)
        o.prepare_to_play(FS, N)
        outs.append(o.run(x[s]))
    return np.stack(outs)


@pytest.mark.parametrize("S,N,B,iir,yin,lpc", [(6, 1024, 8, "exact", "direct", 15), (6, 1024, 8, "exact", "xcorr", 15), (5, 1024, 3, "fast", "xcorr", 15),
                                               (4, 256, 7, "exact", "xcorr", 15), (300, 1024, 8, "fast", "xcorr", 15), (7, 2048, 2, "exact", "xcorr", 15),
                                               (5, 1024, 3, "fast", "xcorr", 48), (4, 1024, 4, "fast", "xcorr", 40), (4, 1024, 3, "fast", "direct", 33),
                                               (4, 1024, 3, "exact", "xcorr", 24), (3, 1024, 2, "exact", "xcorr", 48)])
def test_time_parallel_front_end_equals_block_by_block(S, N, B, iir, yin, lpc):
    """with vp_set_time_parallel on, a multi-block pitch-only call computes yin() and the LPC of every frame that starts
    inside it in vp_k_pitch_front (a workgroup per stream and frame) and the serial kernel consumes the records.  Same routines,
    same modes: the output must be the block-by-block path's bit for bit (and, in the exact modes, the oracle's), over gate
    crossings, cold starts, frames that straddle calls, blocks shorter than a frame, and the register-light builds above 256
    streams.  LPC orders above 15: the serial kernel takes the fused three-group autocorrelation on one wavefront and the recursion on
    another (fast mode: levinson_fast64), the front end the group-by-group form on one -- same sums, same order, same bits."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    calls = 3
    T = N * B * calls
    U = min(S, 6)
    base = _streams(U, T)
    base[0, 0] *= np.where((np.arange(T) // 7000) % 2 == 0, 1.0, 1e-5).astype(np.float32)          # gate crossings
    if U > 2:
        base[2, 0, :N * 3] = 0                                                                     # a silent start
    x = np.ascontiguousarray(base[np.arange(S) % U])

    def run(tp):
        p = BatchVocoderProcessor(vocBool=0, lpcPitch=lpc)
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode(iir)
        p.set_yin_mode(yin)
        p.set_time_parallel(tp)
        xs = torch.from_numpy(x).cuda().view(S, 3, calls * B, N).permute(2, 0, 1, 3).contiguous()
        ys = []
        for i in range(calls):
            y = torch.empty((B, S, 2, N), dtype=torch.float32, device="cuda")
            if tp is None:
                for b in range(B):
                    p.process_device(xs[i * B + b], y[b])
            else:
                p.process_blocks_device(xs[i * B:(i + 1) * B], y)
            ys.append(y)
        torch.cuda.synchronize()
        out = torch.cat(ys, 0).permute(1, 2, 0, 3).reshape(S, 2, T).cpu().numpy()
        return out, [p.pitch_state(s) for s in range(min(S, U))], p.debug_stamps()[59:62]

    ref, st_ref, _ = run(None)
    got, st_got, timeouts = run(True)
    assert list(timeouts) == [0, 0, 0]
    assert np.isfinite(got).all()
    bad = np.nonzero(got != ref)
    assert bad[0].size == 0, f"{bad[0].size} samples differ, first at stream {bad[0][0]}, sample {bad[2][0]}"
    for a, b_ in zip(st_ref, st_got):
        for k in ("period", "pitch", "beta", "anMarks", "stMarks", "gateOpen"):
            assert np.array_equal(a[k], b_[k]), k
    if iir == "exact":
        want = _oracle(base, N, dict(vocBool=0, lpcPitch=lpc))
        assert np.array_equal(got[:U], want)
    else:
        want = _oracle(base, N, dict(vocBool=0, lpcPitch=lpc))
        rms = np.sqrt(np.mean((got[:U].astype(np.float64) - want) ** 2))
        assert rms < 1e-4, rms                                                                     # BASELINE.json north_star tolerance
    assert np.abs(got).max() > 0.05


@pytest.mark.parametrize("S,N,B", [(6, 1024, 8), (5, 256, 16), (300, 1024, 4), (3, 512, 5)])
def test_combined_multi_block_plan_equals_block_by_block(S, N, B):
    """SURVEY 8(f2), combined plan: with both processes on (VP_IIR_FAST, batched vocoder pipeline) a multi-block call runs the serial
    pitch kernel once over all the blocks and the vocoder pipeline once over all their windows.  Against the block-by-block path
    only the order of the additions into the output accumulator differs: every pitch decision must be identical, the audio equal
    to rounding level, and the result within the tolerance of the tolerance mode against the oracle -- over gate crossings (both
    gates), a silent start, calls that follow each other (overhang carried across calls) and a side-chain that goes quiet."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    calls = 3
    T = N * B * calls
    U = min(S, 6)
    base = _streams(U, T)
    base[0, 0] *= np.where((np.arange(T) // 7000) % 2 == 0, 1.0, 1e-5).astype(np.float32)          # voice gate crossings
    base[1, 1:] *= np.where((np.arange(T) // 5000) % 3 == 0, 1e-6, 1.0).astype(np.float32)         # synth gate crossings
    if U > 2:
        base[2, 0, :N * 3] = 0
    x = np.ascontiguousarray(base[np.arange(S) % U])

    def run(multi):
        p = BatchVocoderProcessor()
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode("fast")
        p.set_vocoder_path("batched")
        p.profile_enable(True)
        xs = torch.from_numpy(x).cuda().view(S, 3, calls * B, N).permute(2, 0, 1, 3).contiguous()
        ys = []
        for i in range(calls):
            y = torch.empty((B, S, 2, N), dtype=torch.float32, device="cuda")
            if multi:
                p.process_blocks_device(xs[i * B:(i + 1) * B], y)
            else:
                for b in range(B):
                    p.process_device(xs[i * B + b], y[b])
            ys.append(y)
        torch.cuda.synchronize()
        out = torch.cat(ys, 0).permute(1, 2, 0, 3).reshape(S, 2, T).cpu().numpy()
        return out, [p.pitch_state(s) for s in range(U)], p.debug_stamps()[59:62], p.profile_read()

    ref, st_ref, _, _ = run(False)
    got, st_got, timeouts, kern = run(True)
    assert list(timeouts) == [0, 0, 0]
    launches = sorted(n for (_, n) in kern.values() if n)
    assert launches == [calls, calls], kern             # one launch of the pitch kernel and one of the pipeline per CALL
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 2e-6, np.abs(got - ref).max()
    for a, b_ in zip(st_ref, st_got):
        for k in ("period", "pitch", "anMarks", "stMarks", "gateOpen"):
            assert np.array_equal(a[k], b_[k]), k
    want = _oracle(base, N, dict())
    rms = np.sqrt(np.mean((got[:U].astype(np.float64) - want) ** 2))
    assert rms < 1e-4, rms                                                                         # BASELINE.json north_star tolerance
    assert np.abs(got).max() > 0.05
