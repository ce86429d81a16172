"""GPU parity tests added in round 2 (through the C ABI, against the CPU oracle on identical inputs):
per-stream pitchBool/vocBool (PluginProcessor.cpp:214-221), ScopedNoDenormals (:205), multi-block launches with host
blocks smaller than the chunk, low sample rates (LDS scratch sizing), STFT geometry validation, and the
scatter/gather helpers over backend nccl (= RCCL)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 44100.0


def _streams(S, T, **kw):
    from vocoderproject_amd.synth import make_streams
    return np.ascontiguousarray(make_streams(S, T, **kw).numpy())


def _assert_equal(got, ref, what=""):
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"{what}: {len(bad)} samples differ, first at {bad[0]}, max abs {np.abs(got - ref).max()}"


# ---- per-stream pitchBool / vocBool ---------------------------------------------------------------------------------

@pytest.mark.parametrize("N,iir", [(1024, "exact"), (300, "exact"), (1024, "fast")])
def test_per_stream_switches_mixed_batch(N, iir):
    """Every stream is its own plugin instance with its own pitchBool / vocBool (PluginProcessor.cpp:214-221): a process
    that is switched off does not advance its startSample / nChunk, pitchBool off calls silence().  Streams are switched
    at different blocks, so their window and chunk grids drift apart; each must equal its own oracle instance."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, B = 7, 40 if N == 1024 else 90
    x = _streams(S, N * B)
    init = [dict(), dict(pitchBool=0), dict(vocBool=0), dict(pitchBool=0, vocBool=0), dict(), dict(vocBool=0), dict(pitchBool=0)]
    sched = {3: (4, "vocBool", 0), 5: (1, "pitchBool", 1), 8: (4, "vocBool", 1), 9: (2, "vocBool", 1), 11: (3, "pitchBool", 1),
             14: (0, "pitchBool", 0), 15: (3, "vocBool", 1), 17: (0, "pitchBool", 1), 20: (5, "pitchBool", 0), 22: (6, "keyPitch", 3),
             23: (5, "pitchBool", 1), 27: (2, "pitchBool", 0), 31: (2, "pitchBool", 1)}
    p = BatchVocoderProcessor()
    p.prepareToPlay(FS, N, S)
    p.set_iir_mode(iir)
    os_ = []
    for s_, kv in enumerate(init):
        o = O.OracleStream()
        o.prepare_to_play(FS, N)
        for k, v in kv.items():
            p.setStreamParameter(s_, k, v)
            o.set_param(k, v)
        os_.append(o)
    worst = 0.0
    for b in range(B):
        if b in sched:
            s_, k, v = sched[b]
            p.setStreamParameter(s_, k, v)
            os_[s_].set_param(k, v)
        if b == 34:                                                   # handle-wide set: one set of switches again, grids stay apart
            p.setParameter("vocBool", 1)
            for o in os_:
                for k, v in dict(pitchBool=1, vocBool=1, keyPitch=12).items():
                    o.set_param(k, v)
        blk = np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])
        got = p.process(blk)
        for s_ in range(S):
            io = blk[s_].copy()
            os_[s_].process_block(io)
            if iir == "exact":
                _assert_equal(got[s_], io[:2], f"block {b} stream {s_}")
            else:
                worst = max(worst, float(np.sqrt(np.mean((got[s_].astype(np.float64) - io[:2]) ** 2))))
    if iir == "fast":
        assert worst < 1e-4, worst                                    # north_star's tolerance (per-sample RMS)
    ub = np.sum([o.ub_counters() for o in os_], axis=0)
    assert list(p.ub_counters()) == list(ub)


def test_per_stream_switches_large_batch_cohorts():
    """The same on a batch large enough for the register-light builds, with two cohorts of very different size."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 300, 1024, 10
    x = _streams(S, N * B)
    p = BatchVocoderProcessor()
    p.prepareToPlay(FS, N, S)
    odd = [5, 77, 299]
    for s_ in odd:
        p.setStreamParameter(s_, "vocBool", 0)
    ys = []
    for b in range(B):
        if b == 4:
            p.setStreamParameter(77, "vocBool", 1)
            p.setStreamParameter(8, "pitchBool", 0)
        ys.append(p.process(np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])))
    got = np.concatenate(ys, axis=2)
    for s_ in (0, 5, 8, 77, 150, 299):
        o = O.OracleStream()
        o.prepare_to_play(FS, N)
        if s_ in odd:
            o.set_param("vocBool", 0)
        ref = []
        for b in range(B):
            if b == 4 and s_ == 77:
                o.set_param("vocBool", 1)
            if b == 4 and s_ == 8:
                o.set_param("pitchBool", 0)
            io = np.ascontiguousarray(x[s_, :, b * N:(b + 1) * N]).copy()
            o.process_block(io)
            ref.append(io[:2])
        _assert_equal(got[s_], np.concatenate(ref, axis=1), f"stream {s_}")


# ---- ScopedNoDenormals ----------------------------------------------------------------------------------------------

def test_denormal_inputs_and_outputs_flush_like_scoped_no_denormals():
    """PluginProcessor.cpp:205: processBlock runs with FTZ|DAZ.  f32-denormal input samples enter as 0.0 and results in
    the f32-denormal range leave as 0.0f; the kernels are built with the matching denormal mode.  Dry voice/carrier on,
    so input samples reach the output directly."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 4, 1024, 24
    T = N * B
    x = _streams(S, T)
    rng = np.random.default_rng(7)
    den = (10.0 ** rng.uniform(-45, -38.2, size=T)).astype(np.float32) * rng.choice([-1.0, 1.0], size=T).astype(np.float32)
    assert np.count_nonzero((np.abs(den) > 0) & (np.abs(den) < 1.1754944e-38)) > T // 2
    # stream 0: nothing but denormals (gate closed: only the dry paths speak)
    x[0, 0] = den; x[0, 1] = den[::-1]; x[0, 2] = -den
    # stream 1: a normal voice with every 5th sample replaced by a denormal, carrier decaying through the denormal range
    x[1, 0, ::5] = den[::5]
    decay = (0.3 * np.exp(-np.arange(T) / 180.0)).astype(np.float32)
    x[1, 1] = decay; x[1, 2] = decay * np.float32(0.5)
    x[1, 0, 14000:] = 0.0                                            # the voice stops: the carrier's denormal tail is alone in the output
    # stream 2: the voice fades out into the denormal range and comes back (gate crossings on the way)
    env = np.exp(-np.abs(((np.arange(T) % 12000) - 6000)) / 60.0).astype(np.float64)
    x[2, 0] = (x[2, 0].astype(np.float64) * (1.0 - env) * 1e-3 + 1e-41 * np.sin(np.arange(T))).astype(np.float32)
    params = dict(gainVoice=0.0, gainSynth=-1.0, gainPitch=-3.0)
    p = BatchVocoderProcessor(**params)
    p.prepareToPlay(FS, N, S)
    got = p.run(x)
    ref, raw = [], []
    for s_ in range(S):
        o = O.OracleStream(**params)
        o.prepare_to_play(FS, N)
        ref.append(o.run(x[s_]))
        o2 = O.OracleStream(**params)                                   # the same WITHOUT flushing: must differ, or the test is vacuous
        o2.prepare_to_play(FS, N)
        o2.set_ftz(False)
        raw.append(o2.run(x[s_]))
    ref, raw = np.stack(ref), np.stack(raw)
    assert not np.array_equal(ref[0], raw[0]) and not np.array_equal(ref[1], raw[1])
    _assert_equal(got, ref, "denormal inputs")
    tiny = (np.abs(got) > 0) & (np.abs(got) < 1.1754944e-38)
    assert not tiny.any()                                             # nothing denormal leaves
    assert np.abs(got[3]).max() > 0.05


# ---- multi-block launches with host blocks smaller than the chunk ------------------------------------------------------

@pytest.mark.parametrize("N", [64, 100, 128])
def test_multi_block_launch_with_blocks_smaller_than_the_chunk(N):
    """with N < C a launch of several blocks can start on a block that has no chunk step of its own; every
    block of the launch must still be ingested, processed and emitted."""
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    S = 4
    B = (1024 * 14) // N
    x = _streams(S, N * B)
    xb = torch.from_numpy(np.ascontiguousarray(x.reshape(S, 3, B, N).transpose(2, 0, 1, 3))).cuda()      # [B][S][3][N]

    def run(split):
        p = BatchVocoderProcessor(vocBool=0)
        p.prepareToPlay(FS, N, S)
        y = torch.full((B, S, 2, N), float("nan"), dtype=torch.float32, device="cuda")
        b = 0
        for n in split:
            p.process_blocks_device(xb[b:b + n].contiguous(), y[b:b + n])
            b += n
        assert b == B
        torch.cuda.synchronize()
        return y.cpu().numpy().transpose(1, 2, 0, 3).reshape(S, 2, B * N)

    from oracle import oracle_py as O
    ref = []
    for s_ in range(S):
        o = O.OracleStream(vocBool=0)
        o.prepare_to_play(FS, N)
        ref.append(o.run(x[s_]))
    ref = np.stack(ref)
    splits = [[1] * B, [3] * (B // 3) + [B % 3] * (1 if B % 3 else 0), [7, 2, 5] * (B // 14) + [B - 14 * (B // 14)] * (1 if B % 14 else 0), [B]]
    for split in splits:
        split = [n for n in split if n > 0]
        _assert_equal(run(split), ref, f"N={N} split {split[:6]}...")


# ---- low sample rates: the scratch that lives in the yinTemp / running-sum LDS regions -----------------------------------

@pytest.mark.parametrize("fs,N,prepare,params", [
    (8000.0, 256, None, dict()),
    (11025.0, 256, None, dict()),                                            # C = 64, tauMax = 111: common-case build
    (11025.0, 64, None, dict(vocBool=0)),
    (16000.0, 512, None, dict()),
    (16000.0, 256, (16000.0, 256, 512, 384, 256, 64), dict()),               # explicit: C = 128
    (9000.0, 200, None, dict(lpcPitch=100, lpcVoice=60)),                     # order above tauMax + 1
    (12000.0, 128, (12000.0, 128, 256, 192, 128, 32), dict(lpcPitch=30)),
])
@pytest.mark.parametrize("iir", ["exact", "fast"])
def test_low_sample_rates(fs, N, prepare, params, iir):
    """at low sample rates tauMax + 1 is smaller than the scratch parked in the yinTemp / running-sum regions
    (grain table, recursion history, block-form IIR impulse response); they are now sized for it."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S = 3
    T = max(8, int(3.0 * fs) // N) * N
    x = _streams(S, T, fs=fs)
    p = BatchVocoderProcessor(**params)
    if prepare:
        p.prepareExplicit(prepare[0], prepare[1], S, *prepare[2:])
    else:
        p.prepareToPlay(fs, N, S)
    p.set_iir_mode(iir)
    got = p.run(x)
    ref = []
    for s_ in range(S):
        o = O.OracleStream(**params)
        if prepare:
            o.prepare_explicit(*prepare)
        else:
            o.prepare_to_play(fs, N)
        ref.append(o.run(x[s_]))
    ref = np.stack(ref)
    assert np.abs(ref).max() > 0.02
    if iir == "exact":
        _assert_equal(got, ref, f"fs={fs} N={N} {params} [{p.pitch_kernel_name()}]")
    else:
        rms = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
        assert rms < 1e-4, (rms, p.pitch_kernel_name())              # north_star's tolerance
        st = [p.pitch_state(s_) for s_ in range(S)]
        assert all(np.isfinite(s_["a"]).all() for s_ in st)


# ---- STFT geometry validation -----------------------------------------------------------------------------------------

def test_stft_geometry_validation_and_largest_frame():
    import torch
    from vocoderproject_amd import StftRoundTrip, VpError
    with pytest.raises(VpError) as e:
        StftRoundTrip(2, 4096, 1024, 1024)                           # hop == frame: no overlap to normalise
    assert e.value.code == -4
    with pytest.raises(VpError) as e:
        StftRoundTrip(2, 4096 * 6, 4096, 1024)                       # round 4: the fused kernels are built for 1024 and 2048 points
    assert e.value.code == -4
    T = 2048 * 6
    st = StftRoundTrip(2, T, 2048, 512)                               # the largest frame: sixteen complex points per lane
    assert st.fused
    x = torch.randn((2, T), dtype=torch.float32, device="cuda") * 0.1
    y = torch.empty_like(x)
    st(x, y)
    torch.cuda.synchronize()
    a, b = x.cpu().numpy()[:, 2048:-2048], y.cpu().numpy()[:, 2048:-2048]
    assert np.abs(a - b).max() < 1e-5


# ---- RCCL: scatter / gather over backend nccl ---------------------------------------------------------------------------

def test_scatter_gather_over_nccl_world_size_1():
    """the root fan-out / fan-in helpers over torch.distributed backend "nccl" (= RCCL).  One GPU per box
    here, so world_size = 1: the process group is real RCCL, the exchange degenerates to the local copy."""
    import torch
    import torch.distributed as dist
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    from vocoderproject_amd.dist import gather_streams, scatter_streams
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        S, N, B = 5, 1024, 10
        x = _streams(S, N * B)
        xr = torch.from_numpy(x).to(dev)
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)                                            # RCCL really up
        assert float(t.item()) == 1.0
        p = BatchVocoderProcessor()
        p.prepareToPlay(FS, N, S)
        outs = []
        for b in range(B):
            mine = scatter_streams(xr[:, :, b * N:(b + 1) * N].contiguous(), S, (3, N), torch.float32, dev)
            y = torch.empty((S, 2, N), dtype=torch.float32, device=dev)
            p.process_device(mine, y)
            outs.append(gather_streams(y, S).cpu().numpy())
        got = np.concatenate(outs, axis=2)
    finally:
        dist.destroy_process_group()
    ref = []
    for s_ in range(S):
        o = O.OracleStream()
        o.prepare_to_play(FS, N)
        ref.append(o.run(x[s_]))
    _assert_equal(got, np.stack(ref), "nccl world 1")


def _nccl_two_rank_worker(rank, world, port, S, N, B, ret):
    import torch
    import torch.distributed as dist
    from vocoderproject_amd import BatchVocoderProcessor
    from vocoderproject_amd.dist import exchange_steps, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (see bench.py)
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        lo, hi = shard_range(S, rank, world)
        p = BatchVocoderProcessor(device=rank)
        p.prepareToPlay(FS, N, hi - lo)
        st = torch.cuda.current_stream(dev)
        xr = torch.from_numpy(_streams(S, N * B)).to(dev).view(S, 3, B, N).permute(2, 0, 1, 3).contiguous() if rank == 0 else None
        outs = [torch.empty((S, 2, N), dtype=torch.float32, device=dev) for _ in range(B)] if rank == 0 else None
        exchange_steps(B, S, (lambda i: xr[i]), (lambda i: outs[i]), (3, N), (2, N), torch.float32, dev,
                       lambda i_, o_: p.process_device(i_, o_, st.cuda_stream))
        torch.cuda.synchronize(dev)
        if rank == 0:
            ret["y"] = torch.stack(outs, 0).permute(1, 2, 0, 3).reshape(S, 2, B * N).cpu().numpy()
    finally:
        dist.destroy_process_group()


def test_exchange_over_nccl_two_ranks():
    """SURVEY 8(e) on real RCCL traffic: rank 0 holds the batch, two ranks (two GPUs) each process their shard per step,
    double-buffered scatter / gather (dist.exchange_steps); the reassembled output must be the oracle's bit for bit.  Needs two
    GPUs: skips itself on the one-GPU boxes this build has had (there the gloo tests and the world-size-1 test above cover
    the same code)."""
    import torch
    import torch.multiprocessing as mp
    from oracle import oracle_py as O
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    S, N, B = 7, 1024, 8                                              # ragged shards: 4 + 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    with ctx.Manager() as man:
        ret = man.dict()
        procs = [ctx.Process(target=_nccl_two_rank_worker, args=(r, 2, port, S, N, B, ret)) for r in range(2)]
        for q in procs:
            q.start()
        for q in procs:
            q.join(600)
        assert all(q.exitcode == 0 for q in procs), [q.exitcode for q in procs]
        got = ret["y"]
    x = _streams(S, N * B)
    ref = []
    for s_ in range(S):
        o = O.OracleStream()
        o.prepare_to_play(FS, N)
        ref.append(o.run(x[s_]))
    _assert_equal(got, np.stack(ref), "nccl two ranks")


# ---- large batches from a cold start, certified-YIN fallback path -------------------------------------------------------

@pytest.mark.parametrize("yin", ["xcorr", "xcorr_force_fallback"])
@pytest.mark.parametrize("mode", ["pitch", "both"])
def test_large_batch_cold_start_fallback_frames(yin, mode):
    """Round-2 finding: at 1024 streams (two workgroups per CU, second-round workgroups starting beside a running one)
    the first block -- whose all-zero first frame sends every stream through the certified YIN's fallback -- came out
    wrong (NaN) for a few streams and some in-kernel flag waits ran into their bound: a wavefront could read the
    certification verdict after it had been reset and leave the others one barrier out of step.  Every copy of a stream
    must come out identical wherever it sits in the batch, no flag wait may time out, and sampled streams must match
    the oracle (decisions exactly; samples within the FAST tolerance)."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B, U = 1024, 1024, 4, 16
    base = _streams(U, N * B)
    x = np.ascontiguousarray(base[np.arange(S) % U])
    params = dict(vocBool=int(mode == "both"))
    for rep in range(3):                                              # the fault was intermittent
        p = BatchVocoderProcessor(**params)
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode("fast")
        p.set_yin_mode(yin)
        assert p.pitch_kernel_name() == "vp_k_pitch_lite_fast_c"
        p.debug_stamps(reset=True)
        got = p.run(x)
        st = p.debug_stamps(reset=False)
        assert st[59] == 0 and st[60] == 0 and st[61] == 0, ("flag waits timed out", st[59:62])
        assert np.isfinite(got).all()
        for u in range(U):
            assert np.all(got[u::U] == got[u]), (rep, u)
    ref = []
    for u in range(U):
        o = O.OracleStream(**params)
        o.prepare_to_play(FS, N)
        ref.append(o.run(base[u]))
    err = got[:U].astype(np.float64) - np.stack(ref)
    assert np.sqrt((err ** 2).mean()) < 1e-4


# ---- the batched lane-per-window vocoder pipeline (vp_voc2.hip) ------------------------------------------------------------

def _run_blocks(p, x, N):
    ys = [p.process(np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])) for b in range(x.shape[2] // N)]
    return np.concatenate(ys, axis=2)


@pytest.mark.parametrize("name,prepare,params,S,B", [
    ("default_both", None, dict(), 9, 16),
    ("voc_only", None, dict(pitchBool=0), 70, 10),
    ("cfg3_1024_256_order24", (44100.0, 1024, 1024, 768, 1024, 256), dict(pitchBool=0, lpcVoice=24, lpcSynth=5), 5, 12),
    ("cfg5_48k_2048", (48000.0, 2048, 2048, 1536, 2048, 512), dict(lpcVoice=48, lpcPitch=48, lpcSynth=30), 3, 8),
    ("half_overlap_N300", (44100.0, 300, 1024, 512, 512, 256), dict(pitchBool=0, lpcVoice=7, lpcSynth=2), 4, 40),
    ("48k_556_139", "p2p48", dict(lpcVoice=33, lpcSynth=11), 3, 30),
])
def test_batched_vocoder_pipeline_bit_exact(name, prepare, params, S, B):
    """The lane-per-window pipeline (VP_VOC_BATCHED) against the oracle, bit for bit in VP_IIR_EXACT mode, and against the
    workgroup-per-stream kernel; gate crossings, windows that are not a multiple of 4 or 64 samples, carried startSample."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    fs = 48000.0 if prepare == "p2p48" else (prepare[0] if prepare else FS)
    N = 480 if prepare == "p2p48" else (prepare[1] if prepare else 1024)
    x = _streams(S, N * B, fs=fs)
    x[0, 0] *= np.where((np.arange(N * B) // 6000) % 2 == 0, 1.0, 1e-5).astype(np.float32)          # gate crossings
    x[:, 2] *= -0.5
    outs = {}
    for path in ("batched", "workgroup"):
        p = BatchVocoderProcessor(**params)
        if prepare == "p2p48":
            p.prepareToPlay(fs, N, S)
        elif prepare:
            p.prepareExplicit(prepare[0], prepare[1], S, *prepare[2:])
        else:
            p.prepareToPlay(fs, N, S)
        p.set_vocoder_path(path)
        assert (p.vocoder_kernel_name() == "vp_k_v2_pipeline") == (path == "batched")
        outs[path] = _run_blocks(p, x, N)
    ref = []
    for s_ in range(S):
        o = O.OracleStream(**params)
        if prepare == "p2p48" or not prepare:
            o.prepare_to_play(fs, N)
        else:
            o.prepare_explicit(*prepare)
        ref.append(o.run(x[s_]))
    ref = np.stack(ref)
    assert np.abs(ref).max() > 0.02
    _assert_equal(outs["batched"], ref, f"{name}: batched vs oracle")
    _assert_equal(outs["workgroup"], ref, f"{name}: workgroup vs oracle")


def test_batched_vocoder_pipeline_fast_mode_per_stream_orders_and_auto_selection():
    """FAST mode (block-form recursion) within the tolerance; per-stream orders / gains / switches in one batch; the automatic
    choice takes the pipeline above 256 streams (>= 1024 windows per block) and falls back above order 48."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S, N, B = 384, 1024, 6
    U = 12
    base = _streams(U, N * B)
    x = np.ascontiguousarray(base[np.arange(S) % U])
    per = {3: dict(lpcVoice=16, lpcSynth=12, gainVoc=-12.0), 40: dict(lpcVoice=48, lpcSynth=30), 77: dict(vocBool=0), 300: dict(lpcVoice=2, lpcSynth=2)}
    for iir in ("exact", "fast"):
        p = BatchVocoderProcessor()
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode(iir)
        assert p.vocoder_kernel_name() == "vp_k_v2_pipeline"          # 384 streams > 256, 3072 windows
        for s_, kv in per.items():
            for k, v in kv.items():
                p.setStreamParameter(s_, k, v)
        got = _run_blocks(p, x, N)
        for s_ in (0, 3, 40, 77, 300, 383):
            o = O.OracleStream()
            o.prepare_to_play(FS, N)
            for k, v in per.get(s_, {}).items():
                o.set_param(k, v)
            want = o.run(x[s_])
            if iir == "exact":
                _assert_equal(got[s_], want, f"stream {s_}")
            else:
                assert np.sqrt(np.mean((got[s_].astype(np.float64) - want) ** 2)) < 1e-4, s_
        # one stream above the pipeline's orders: it becomes a cohort of its own on the workgroup kernel, the other 383 keep the
        # pipeline (round 2 demoted the whole batch) -- and all of them still come out right
        p.setStreamParameter(5, "lpcVoice", 64)
        assert p.vocoder_kernel_name() == "vp_k_v2_pipeline"
        x2 = np.ascontiguousarray(x[:, :, :N * 3])
        got2 = _run_blocks(p, x2, N)
        for s_ in (0, 5, 300):
            o = O.OracleStream()
            o.prepare_to_play(FS, N)
            for k, v in per.get(s_, {}).items():
                o.set_param(k, v)
            want = o.run(x[s_])                                       # (the oracle stream replays the first B blocks with the old order ...)
            if s_ == 5:
                o.set_param("lpcVoice", 64)
            want2 = o.run(x2[s_])                                     # ... then these three
            if iir == "exact":
                _assert_equal(got2[s_], want2, f"stream {s_} after one stream's order went to 64")
            else:
                assert np.sqrt(np.mean((got2[s_].astype(np.float64) - want2) ** 2)) < 1e-4, s_
        for s_ in range(S):
            p.setStreamParameter(s_, "lpcVoice", 64)                  # every stream above: nothing is left for the pipeline
        assert p.vocoder_kernel_name() in ("vp_k_vocoder", "vp_k_vocoder_o48", "vp_k_vocoder_lite")
    q = BatchVocoderProcessor()
    q.prepareToPlay(FS, N, 100)
    assert q.vocoder_kernel_name() == "vp_k_vocoder_o48"                # default lpcVoice 40: the build for orders 33..48


@pytest.mark.parametrize("S,path", [(6, "batched"), (1024, "auto")])
def test_pitch_beside_vocoder_equals_pitch_behind_vocoder(S, path):
    """Round-1 verdict item 1(a): in combined mode the two processes need not run back to back.  VP_IIR_FAST: the pitch kernel
    runs beside the vocoder pipeline and adds into its own accumulator (merged at emit) -- only the order of the additions
    into the output changes, so the result must stay within the FAST tolerance of the sequential plan (and of the oracle),
    with identical pitch decisions; switching the overlap on and off between blocks must leave nothing behind in the second
    accumulator.  VP_IIR_EXACT never overlaps: bit-identical to the oracle whatever the switch says."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    N, B, U = 1024, 14, 6
    base = _streams(U, N * B)
    base[1, 0] *= np.where((np.arange(N * B) // 5000) % 2 == 0, 1.0, 1e-5).astype(np.float32)          # gate crossings
    x = np.ascontiguousarray(base[np.arange(S) % U])
    ref = []
    for u in range(U):
        o = O.OracleStream()
        o.prepare_to_play(FS, N)
        ref.append(o.run(base[u]))
    ref = np.stack(ref)

    def run(iir, overlap):
        p = BatchVocoderProcessor()
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode(iir)
        p.set_vocoder_path(path)
        ys = []
        for b in range(B):
            p.set_overlap(overlap(b))
            ys.append(p.process(np.ascontiguousarray(x[:, :, b * N:(b + 1) * N])))
        return np.concatenate(ys, axis=2), [p.pitch_state(s_) for s_ in range(min(S, U))]

    assert BatchVocoderProcessor().L.vp_get_overlap(BatchVocoderProcessor().h) == 2      # VP_OVERLAP_AUTO by default
    seq, st_seq = run("fast", lambda b: False)
    par, st_par = run("fast", lambda b: True)
    mix, _ = run("fast", lambda b: (b // 3) % 2 == 0)
    for got in (seq, par, mix):
        assert np.isfinite(got).all()
        for u in range(U):
            assert np.all(got[u::U] == got[u]), u                    # every copy of a stream identical, wherever it sits
        err = got[:U].astype(np.float64) - ref
        assert np.sqrt((err ** 2).mean()) < 1e-4
    assert np.sqrt(((par[:U].astype(np.float64) - seq[:U]) ** 2).mean()) < 1e-4
    assert np.abs(par[:U] - seq[:U]).max() < 1e-5                     # rounding-level, not merely "within tolerance"
    for a, b_ in zip(st_seq, st_par):
        for k in ("period", "prevPeriod", "periodNew", "pitch", "beta", "anMarks", "stMarks", "gateOpen"):
            assert np.array_equal(a[k], b_[k]), k
    ex, _ = run("exact", lambda b: True)
    _assert_equal(ex[:U], ref, "exact mode with the overlap switch on")


@pytest.mark.parametrize("seed", list(range(12)))
def test_batched_vocoder_pipeline_randomised_configurations(seed):
    """Random sample rate, host block size, orders (up to the pipeline's 48/30), gains and switches on the lane-per-window
    pipeline (forced), a gate-crossing stream among them: bit-exact against the oracle in exact mode."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    rng = np.random.default_rng(7000 + seed)
    fs = float(rng.choice([16000.0, 22050.0, 32000.0, 44100.0, 48000.0, 44099.0]))
    N = int(rng.choice([64, 100, 278, 441, 512, 1000, 1024, 1536, 2048]))
    params = dict(lpcVoice=int(rng.integers(2, 49)), lpcPitch=int(rng.integers(2, 60)), lpcSynth=int(rng.integers(2, 31)),
                  gainPitch=float(rng.uniform(-20, 6)), gainVoc=float(rng.uniform(-20, 6)),
                  gainVoice=float(rng.choice([-60.0, -30.0])), gainSynth=float(rng.choice([-60.0, -12.0])),
                  pitchBool=int(rng.random() < 0.5), vocBool=1)
    S = int(rng.integers(3, 70))
    T = max(6, int(16000 * fs / 44100.0) // N) * N
    x = _streams(S, T, fs=fs)
    x[0, 0] *= np.where((np.arange(T) // 5000) % 2 == 0, 1.0, 2e-5).astype(np.float32)
    p = BatchVocoderProcessor(**params)
    try:
        p.prepareToPlay(fs, N, S)
    except VpError as e:
        assert e.code == -4, e
        pytest.skip("geometry exceeds the LDS budget")
    p.set_vocoder_path("batched")
    if p.vocoder_kernel_name() != "vp_k_v2_pipeline":
        pytest.skip("more than 64 windows per block at this geometry: the pipeline does not take it")
    got = p.run(x)
    pick = sorted(set([0, 1, S // 2, S - 1]))
    for s_ in pick:
        o = O.OracleStream(**params)
        o.prepare_to_play(fs, N)
        _assert_equal(got[s_], o.run(x[s_]), f"seed {seed}: fs={fs} N={N} S={S} {params} stream {s_}")


@pytest.mark.parametrize("fs,N,params", [
    (8000.0, 64, dict(lpcVoice=100, lpcPitch=4, lpcSynth=11, gainVoc=-12.0)),        # vocoder window 92 samples, order 100 (soak seed 505)
    (8000.0, 100, dict(lpcVoice=93, lpcSynth=30, pitchBool=0)),
    (11025.0, 128, dict(lpcVoice=100, lpcPitch=100, lpcSynth=30)),                    # window 128, frame 256
])
def test_lpc_order_beyond_the_window_length(fs, N, params):
    """Found by the round-2 soak: at low sample rates the vocoder's window can be SHORTER than lpcVoice allows (8 kHz:
    92 samples against orders up to 100).  Lags beyond the window sum nothing in the reference (LPC.cpp:65 `n < wlen - m`);
    the workgroup kernel's tail loop ran with a negative trip start instead.  Bit-exact on both vocoder paths."""
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S = 4
    T = max(40, int(1.2 * fs) // N) * N
    x = _streams(S, T, fs=fs)
    ref = []
    for s_ in range(S):
        o = O.OracleStream(**params)
        o.prepare_to_play(fs, N)
        ref.append(o.run(x[s_]))
    ref = np.stack(ref)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0.01
    for path in ("workgroup", "batched"):
        p = BatchVocoderProcessor(**params)
        p.prepareToPlay(fs, N, S)
        p.set_vocoder_path(path)
        _assert_equal(p.run(x), ref, f"fs={fs} N={N} {params} [{p.vocoder_kernel_name()}]")


@pytest.mark.parametrize("N,iir", [(1024, "exact"), (300, "exact"), (100, "exact"), (1024, "fast"), (512, "fast")])
def test_vocoder_multi_block_launch_equals_block_by_block(N, iir):
    """SURVEY 8(f2) for the vocoder: vp_process_blocks_device with the vocoder-only plan runs up to 16 blocks as ONE launch of
    the lane-per-window pipeline (B times the windows = B times the lanes).  The window grid carries across blocks, every
    block keeps its own ring ingest, silence gate and output slab: B blocks in one call must give exactly what B single
    calls give (and, in exact mode, what the oracle gives) -- gate crossings (whole blocks gated in the middle of a
    launch), dry voice and dry carrier on (their samples come from the ring as it stood before the call, or from the
    input slabs), host blocks smaller than the hop (blocks without a window of their own), odd splits."""
    import torch
    from oracle import oracle_py as O
    from vocoderproject_amd import BatchVocoderProcessor
    S = 6
    B = 48 if N >= 512 else 96
    x = _streams(S, N * B)
    x[0, 0] *= np.where((np.arange(N * B) // 5000) % 2 == 0, 1.0, 1e-5).astype(np.float32)          # gate crossings
    x[1, 1] *= np.where((np.arange(N * B) // 7000) % 2 == 0, 1.0, 1e-5).astype(np.float32)          # ... of the carrier
    x[:, 2] *= -0.5
    params = dict(pitchBool=0, gainVoice=-6.0, gainSynth=-9.0, lpcVoice=24, lpcSynth=7)
    xb = torch.from_numpy(np.ascontiguousarray(x.reshape(S, 3, B, N).transpose(2, 0, 1, 3))).cuda()      # [B][S][3][N]

    def run(split, path="batched"):
        p = BatchVocoderProcessor(**params)
        p.prepareToPlay(FS, N, S)
        p.set_iir_mode(iir)
        p.set_vocoder_path(path)
        y = torch.full((B, S, 2, N), float("nan"), dtype=torch.float32, device="cuda")
        b = 0
        for n in split:
            p.process_blocks_device(xb[b:b + n].contiguous(), y[b:b + n])
            b += n
        assert b == B
        torch.cuda.synchronize()
        return y.cpu().numpy().transpose(1, 2, 0, 3).reshape(S, 2, B * N)

    single = run([1] * B)
    assert np.isfinite(single).all()
    for split in ([B], [16] * (B // 16), [5, 1, 7, 3] * (B // 16), [2] * (B // 2), [1, 15] * (B // 16)):
        got = run(split)
        if iir == "exact":
            _assert_equal(got, single, f"N={N} split {split[:4]}...")
        else:
            assert np.abs(got - single).max() < 1e-5 and np.isfinite(got).all()     # FAST: slice energies sum in the same order: identical or rounding-level
    if iir == "exact":
        ref = []
        for s_ in range(S):
            o = O.OracleStream(**params)
            o.prepare_to_play(FS, N)
            ref.append(o.run(x[s_]))
        _assert_equal(single, np.stack(ref), "single calls vs oracle")
        _assert_equal(run([B]), np.stack(ref), "one launch group vs oracle")
        _assert_equal(run([B], path="workgroup"), np.stack(ref), "workgroup path (block by block)")


@pytest.mark.parametrize("order", [17, 24, 32, 33, 40, 47, 48])
def test_fast_block_recursion_for_orders_17_to_48(order):
    """FAST mode, pitch orders 17..48: the chunk's all-pole recursion runs as zero-state response of every 64-sample block
    plus a 64 x order matrix applied to the previous outputs (iir_block_wave_hc).  Same tolerance as every FAST path
    (RMS < 1e-4, north_star; in practice last-bit flips of the float32 cast), from a cold start, across frames, gate
    closings and a block size that makes launches begin in the middle of a frame."""
    from vocoderproject_amd import BatchVocoderProcessor
    import test_gpu_parity as T
    S = 5
    for N in (1024, 256):
        B = 16 * (1024 // N)
        x = T._streams(S, N * B)
        x[1, 0] *= np.where((np.arange(N * B) // 5000) % 2 == 0, 1.0, 1e-6).astype(np.float32)     # gate closes and reopens
        params = dict(lpcPitch=order, vocBool=0)
        ref = T._oracle_run(x, N, params)
        p = BatchVocoderProcessor(**params)
        p.prepareToPlay(T.FS, N, S)
        p.set_iir_mode("fast")
        p.set_yin_mode("xcorr")
        got = p.run(x)
        err = got.astype(np.float64) - ref
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.isfinite(got).all()
        assert np.sqrt((err ** 2).mean()) < 1e-4
        assert np.abs(err).max() <= 4e-7 * scale, (order, N, np.abs(err).max(), scale)
        assert np.abs(ref).max() > 0.01, "vacuous comparison"


@pytest.mark.parametrize("order", [2, 7, 15, 16])
def test_fast_block_recursion_of_the_register_light_builds(order):
    """Above 256 streams the FAST recursion of pitch orders up to 16 runs as four 16-tap zero-state passes plus a 16-entry
    history row (iir_block_wave_hc16; the 128-VGPR builds cannot hold the 64 taps of the regular form).  Same tolerance as every
    FAST path, against the oracle on a sample of streams; batch position must not matter."""
    from vocoderproject_amd import BatchVocoderProcessor
    import test_gpu_parity as T
    S, N, B = 260, 1024, 12
    base = T._streams(13, N * B)
    base[1, 0] *= np.where((np.arange(N * B) // 5000) % 2 == 0, 1.0, 1e-6).astype(np.float32)     # gate closes and reopens
    x = np.ascontiguousarray(np.tile(base, (20, 1, 1)))
    params = dict(lpcPitch=order, vocBool=0)
    p = BatchVocoderProcessor(**params)
    p.prepareToPlay(T.FS, N, S)
    p.set_iir_mode("fast")
    p.set_yin_mode("xcorr")
    assert p.pitch_kernel_name().startswith("vp_k_pitch_lite_fast")
    got = p.run(x)
    pick = [0, 1, 5, 12]
    ref = T._oracle_run(base[pick], N, params)
    err = got[pick].astype(np.float64) - ref
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(got).all()
    assert np.sqrt((err ** 2).mean()) < 1e-4
    assert np.abs(err).max() <= 4e-7 * scale, (order, np.abs(err).max(), scale)
    assert np.abs(ref).max() > 0.01, "vacuous comparison"
    np.testing.assert_array_equal(got[1], got[1 + 13 * 19])
