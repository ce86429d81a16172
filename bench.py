#!/usr/bin/env python3
"""bench.py -- throughput of the pitch-corrector / vocoder hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode pitch|voc|both] [--streams S] [--block N]

A "step" is one processBlock() over the whole stream batch: S streams x N samples (default the
BASELINE.json configs[1] workload: 256 mono streams, pitch corrector only, 1024-sample frames,
256-sample hop, 44.1 kHz, host block N = 1024).  The metric unit "frame" is one 256-sample hop of
one stream through the enabled path, so a step is S*N/256 frames.

Inputs are synthetic (vocoderproject_amd.synth) and already resident in HBM when the timed region
starts.  One process per GPU; for N > 1 every rank owns its own S streams (weak scaling, streams
are independent: no data-path collective), rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline     -- dominant kernel, ALGORITHMIC bytes (3072 B per pitch frame, 5120 B with the
                  vocoder) / its HIP-event duration vs the 8 TB/s HBM peak (the path is
                  ALU/latency-bound, the fraction is tiny by construction)
  cpu_baseline -- the CPU oracle (the build's restatement of the reference, kind "port") timed on
                  this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 44100.0
HOP = 256
ALG_BYTES_PER_FRAME = {"pitch": 3072, "voc": 5120, "both": 5120}
PROFILE_EVERY = 8       # kernel durations are sampled live inside the timed region (two event records cost the stream 2-3 us)
# fp64 operations per hop-frame of the reference's arithmetic at the default geometry:
# the secondary, ALU-side sanity line (no MFMA: nothing on this path is a dense contraction)
ALG_FLOP_PER_FRAME = {"pitch": 0.51e6, "voc": 0.36e6, "both": 0.87e6}
FP64_VALU_PEAK_TFLOPS = 78.6                                            # vector fp64 peak used by SURVEY.md section 8d (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz)
HBM_PEAK_GBS = 8000.0                                                  # MI355X_MICROARCH.md
UNIQUE_BLOCKS = 16                                                      # synthetic input ring, cycled


def usable_cores():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(mode, N, seconds_target=12.0):
    """Time the CPU oracle on all host cores: one stream per task (ctypes releases the GIL)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_py as O
    from vocoderproject_amd.synth import make_streams
    cores = usable_cores()
    params = dict(pitchBool=int(mode != "voc"), vocBool=int(mode != "pitch"))
    # calibrate on one stream
    x1 = np.ascontiguousarray(make_streams(1, N * 8).numpy())[0]
    o = O.OracleStream(**params)
    o.prepare_to_play(FS, N)
    t = time.perf_counter()
    o.run(x1)
    per_block = (time.perf_counter() - t) / 8
    blocks = 128
    reps = int(max(1, min(512, round(seconds_target / max(per_block * blocks, 1e-6)))))
    n_streams = cores
    x = np.ascontiguousarray(make_streams(n_streams, N * blocks).numpy())

    def work(s):
        oo = O.OracleStream(**params)
        oo.prepare_to_play(FS, N)
        for _ in range(reps):            # the stream simply continues: state carries over
            oo.run(x[s])

    t = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, range(n_streams)))
    dt = time.perf_counter() - t
    blocks *= reps
    frames = n_streams * blocks * N / HOP
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_streams} streams x {blocks} blocks of {N} samples, mode={mode}, one oracle stream per thread, "
                      f"{dt:.1f} s wall"}


def measured_traffic(mode, S, N, iir, mono=False):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_traffic.json:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this very command); None when the
    configuration being benched is not the one that was profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f)
        key = f"{mode}{'-mono' if mono else ''}/S{S}/N{N}/{iir}"
        return t["per_launch_bytes"].get(key)
    except (OSError, ValueError, KeyError):
        return None


def stft_figure(dev, S, T=1024 * 16, F=1024, hop=256, reps=20):
    """Standalone STFT->iSTFT kernel (Hann, radix-2 FFT, iFFT, OLA): NO reference counterpart, reported apart
    from the metric."""
    import torch
    from vocoderproject_amd import StftRoundTrip
    st = StftRoundTrip(S, T, F, hop, device=dev.index or 0)
    x = torch.randn((S, T), dtype=torch.float32, device=dev) * 0.1
    y = torch.empty_like(x)
    for _ in range(3):
        st(x, y)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        st(x, y)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    frames = S * st.n_frames
    return {"frames_per_s": frames / dt, "note": "standalone STFT round trip 1024/256, fp64 radix-2 FFT in LDS; no reference counterpart",
            "hbm_gbs": (2 * S * T * 4 + 2 * frames * F * 4) / dt / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="pitch", choices=["pitch", "voc", "both"])
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--block", type=int, default=1024, help="samplesPerBlock N")
    ap.add_argument("--iir", default="fast", choices=["fast", "exact"],
                    help="arithmetic of the two synthesis IIRs: 'fast' (VP_IIR_FAST, tolerance-tested) is the measured "
                         "configuration; 'exact' (bit-identical to the oracle) is timed beside it as value_exact_mode")
    ap.add_argument("--single-mode", action="store_true",
                    help="do not time the other IIR mode or the STFT kernel (for profiler runs: one kernel population)")
    ap.add_argument("--yin", default="xcorr", choices=["direct", "xcorr", "fft"],
                    help="evaluation of the YIN difference function: 'xcorr' (default) is certified to take the same "
                         "decisions as 'direct' (the reference's arithmetic) and falls back to it otherwise: same output bits")
    ap.add_argument("--blocks-per-step", type=int, default=1,
                    help="host blocks of N samples handed over per step (vp_process_blocks_device; one launch in pitch mode)")
    ap.add_argument("--shift", type=float, default=None,
                    help="extension (no reference counterpart): fixed pitch-shift interval in semitones, +x on even and -x "
                         "on odd streams (vp_set_pitch_shift), instead of the correction to the key's nearest note")
    ap.add_argument("--three-channel", action="store_true",
                    help="pitch mode: hand over [S][3][N] buffers (voice + zero side chain) instead of the mono voice buffers "
                         "configs[1] describes (vp_process_block_mono_device; same output, a third of the input bytes)")
    ap.add_argument("--cfg5", action="store_true",
                    help="BASELINE configs[4] geometry instead of the default one: 48 kHz, 2048-point frames hop 512 (pitch "
                         "2048/1536, vocoder 2048/512), LPC orders 48/48/30, host block 2048 (a documentation figure: use with --mode both --no-cpu)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vocoderproject_amd import BatchVocoderProcessor
    from vocoderproject_amd.synth import make_streams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("VP_BENCH_FORCE_DIST") == "1"   # the latter: exercise the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    n_gpus = world

    S, N, mode = args.streams, args.block, args.mode
    global FS, HOP
    if args.cfg5:
        FS, HOP, N = 48000.0, 512, 2048
        p = BatchVocoderProcessor(device=local_rank, pitchBool=int(mode != "voc"), vocBool=int(mode != "pitch"),
                                  lpcVoice=48, lpcPitch=48, lpcSynth=30)
        p.prepareExplicit(FS, N, S, 2048, 1536, 2048, 512)
    else:
        p = BatchVocoderProcessor(device=local_rank, pitchBool=int(mode != "voc"), vocBool=int(mode != "pitch"))
        p.prepareToPlay(FS, N, S)
    p.set_yin_mode(args.yin)

    def set_shift(semi):
        for s_ in range(S):
            p.setPitchShift(semi if s_ % 2 == 0 else -semi, on=True, stream=s_)

    if args.shift is not None and mode != "voc":
        set_shift(args.shift)

    # synthetic inputs in HBM: [U][S][3][N], stream ids unique across ranks
    U = UNIQUE_BLOCKS
    x = make_streams(S, N * U, fs=FS, first_stream=rank * S, device=dev)          # [S][3][U*N]
    x = x.view(S, 3, U, N).permute(2, 0, 1, 3).contiguous()                     # [U][S][3][N]
    BPS = args.blocks_per_step
    assert 1 <= BPS <= U and U % BPS == 0
    y = torch.empty((BPS, S, 2, N), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    # configs[1] is "256 MONO streams": the pitch-only workload hands over the voice alone (the plugin's null side-chain
    # pointers, MyBuffer.cpp:93-102); vocoder workloads carry the stereo side chain
    mono = mode == "pitch" and not args.three_channel
    xm = x[:, :, 0, :].contiguous() if mono else None                          # [U][S][N]

    def step(i):
        if BPS == 1:
            if mono:
                p.process_mono_device(xm[i % U], y[0], stream.cuda_stream)
            else:
                p.process_device(x[i % U], y[0], stream.cuda_stream)
        else:
            b0 = (i * BPS) % U
            if mono:
                p.process_blocks_mono_device(xm[b0:b0 + BPS], y, stream.cuda_stream)
            else:
                p.process_blocks_device(x[b0:b0 + BPS], y, stream.cuda_stream)

    def timed(mode_iir, steps, warmup):
        p.set_iir_mode(mode_iir)
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize(dev)
        p.profile_read(reset=True)
        p.profile_enable(PROFILE_EVERY)       # HIP events around every PROFILE_EVERY-th launch of the timed region
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt_ = time.perf_counter() - t0
        p.profile_enable(False)
        return dt_, p.profile_read(reset=True)

    # secondary figure first (shorter), then THE timed region: W warmup steps, exactly K timed steps
    other = "exact" if args.iir == "fast" else "fast"
    k2 = max(10, args.steps // 4)
    dt_other = float("nan")
    if not args.single_mode:
        dt_other, _ = timed(other, k2, max(2, args.warmup // 4))
    # a second secondary figure: the same blocks handed over eight at a time (vp_process_blocks_device: one launch per eight
    # blocks in pitch mode, state stays on chip in between); reported beside `value`, never as it
    MB, k3 = 8, max(4, args.steps // 16)
    dt_mb = float("nan")
    if not args.single_mode and BPS == 1 and mode == "pitch" and U % MB == 0:
        ymb = torch.empty((MB, S, 2, N), dtype=torch.float32, device=dev)
        p.set_iir_mode(args.iir)
        def step_mb(b0):
            if mono:
                p.process_blocks_mono_device(xm[b0:b0 + MB], ymb, stream.cuda_stream)
            else:
                p.process_blocks_device(x[b0:b0 + MB], ymb, stream.cuda_stream)

        for i in range(2):
            step_mb(0)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(k3):
            step_mb((i * MB) % U)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        dt_mb = time.perf_counter() - t0
    dt, prof = timed(args.iir, args.steps, args.warmup)

    tt = torch.tensor([dt, dt_other, dt_mb], dtype=torch.float64, device=dev)
    chk = y.double().abs().sum().view(1)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)          # the only collective: a checksum of the outputs
    dt, dt_other, dt_mb = float(tt[0].item()), float(tt[1].item()), float(tt[2].item())
    # a third secondary figure, AFTER the headline region: BASELINE configs[1]'s "+-12-semitone pitch shift" (fixed
    # interval, +12 on even and -12 on odd streams; extension without a reference counterpart, parity GPU <-> oracle)
    dt_shift = float("nan")
    if not args.single_mode and args.shift is None and mode != "voc" and BPS == 1:
        set_shift(12.0)
        dt_shift, _ = timed(args.iir, k2, max(4, args.warmup // 2))
        ts = torch.tensor([dt_shift], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        dt_shift = float(ts[0].item())

    frames_per_step_gpu = S * N * BPS // HOP
    total_frames = frames_per_step_gpu * args.steps * n_gpus
    value = total_frames / dt

    if rank == 0:
        dom = "vp_k_vocoder" if mode == "voc" else "vp_k_pitch"
        # the dominant kernel by measured time
        dom = max(prof.items(), key=lambda kv: kv[1][0])[0] if prof else dom
        ms, n = prof[dom]
        avg_s = (ms / max(n, 1)) * 1e-3
        alg_bytes = ALG_BYTES_PER_FRAME[mode] * (HOP // 256) * frames_per_step_gpu      # f32 I/O per hop-frame (hop 512: twice the samples)
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        out = {
            "metric": "STFT-geometry frames/sec (1024-pt frames, hop 256) through the pitch-corrector/vocoder path",
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"configs[4] geometry: {S} streams per GPU @48 kHz, 2048-pt frames hop 512, LPC orders 48/48/30, mode {mode}, host block N={N}" if args.cfg5 else "") or f"configs[{1 if mode == 'pitch' else 2 if mode == 'voc' else 3}]: {S} mono streams per GPU @44.1 kHz, "
                                   f"{'pitch corrector (YIN+PSOLA on LPC residual, key=Chrom)' if mode == 'pitch' else 'LPC vocoder' if mode == 'voc' else 'pitch corrector + vocoder'}"
                                   f", 1024-pt frames hop 256, host block N={N}" + (f", {BPS} blocks per step" if BPS > 1 else ""),
                       "streams_per_gpu": S, "block": N, "blocks_per_step": BPS, "mode": mode, "iir_mode": args.iir, "yin_mode": args.yin, "fixed_shift_semitones": args.shift, "input": "mono voice [S][N]" if mono else "[S][3][N]", "frames_per_step": frames_per_step_gpu * n_gpus,
                       "kernel_builds": {"pitch": p.pitch_kernel_name() if mode != "voc" else None, "vocoder": p.vocoder_kernel_name() if mode != "pitch" else None},
                       "parallelism": f"streams sharded over {n_gpus} GPU(s), no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "avg_kernel_us": avg_s * 1e6, "alg_bytes_per_launch": alg_bytes,
                         "note": "path is fp64-VALU/latency-bound (DESIGN.md); HBM fraction is reported as the contract asks",
                         "alu_sanity": {"achieved": ALG_FLOP_PER_FRAME[mode] * value / n_gpus / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": ALG_FLOP_PER_FRAME[mode] * value / n_gpus / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                        "what": "reference-arithmetic fp64 ops per hop-frame x frames/s per GPU, default orders"}},
            "kernel_us": {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof.items() if v[1]},
            "checksum": float(chk.item()),
            f"value_{other}_mode": (frames_per_step_gpu * k2 * n_gpus / dt_other) if dt_other == dt_other else None,
            "value_8_blocks_per_call": ((S * N * MB // HOP) * k3 * n_gpus / dt_mb) if dt_mb == dt_mb else None,
            "value_pm12_semitone_shift": (frames_per_step_gpu * k2 * n_gpus / dt_shift) if dt_shift == dt_shift else None,
        }
        if not args.single_mode:
            out["stft_kernel"] = stft_figure(dev, S)
        out["roofline"]["traffic"] = measured_traffic(mode, S, N, args.iir, mono) if BPS == 1 else None
        if n_gpus == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(mode, N, args.cpu_seconds)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
