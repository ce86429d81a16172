#!/usr/bin/env python3
"""bench.py -- throughput of the pitch-corrector / vocoder hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode pitch|voc|both] [--streams S] [--block N] [--exchange]

A "step" is one processBlock() over the whole stream batch: S streams x N samples per GPU (default the
BASELINE.json configs[1] workload: 256 mono streams, pitch corrector only, 1024-sample frames,
256-sample hop, 44.1 kHz, host block N = 1024).  The metric unit "frame" is one 256-sample hop of
one stream through the enabled path, so a step is S*N/256 frames per GPU.

Inputs are synthetic (vocoderproject_amd.synth) and already resident in HBM when the timed region
starts.  One process per GPU.  `--gpus N` with N > 1 and no launcher (WORLD_SIZE unset) makes this
process spawn N rank processes itself (before anything here touches a GPU) and wait for them; under
`python -m torch.distributed.run ...` the ranks are the launcher's.  Every rank owns its own S streams
(weak scaling, streams are independent: no data-path collective in `value`), rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline     -- dominant kernel, ALGORITHMIC bytes (3072 B per pitch frame, 5120 B with the
                  vocoder) / its HIP-event duration vs the 8 TB/s HBM peak (the path is
                  ALU/latency-bound, the fraction is tiny by construction); `traffic` and
                  `valu` come from the committed rocprofv3 PMC passes of THIS kernel build (matched by kernel
                  name and by a hash of the kernel sources, null when either differs)
  cpu_baseline -- the CPU oracle (the build's restatement of the reference, kind "port") timed on
                  this host's cores on a bounded sample of the same workload (rank 0, one GPU only)
  parity       -- the accuracy half of the metric ("...; RMS err vs CPU ref"): after the timed regions, sampled streams of a FRESH
                  processor of the same configuration (same builds, modes, batch) against the CPU oracle on the same input:
                  rms_err, max_abs_err, rms_rel, decision_mismatch_frames (tracker states that differ), streams, blocks.  One per
                  leg: the headline (`parity`, `parity_exact_mode`, `parity_pm12_semitone_shift`) and configs2/3/4 (`.parity`)
  rccl         -- (N > 1, or --exchange) what torch.distributed backend "nccl" (= RCCL) saw: world size, the
                  all-reduced sum of ranks
  exchange     -- SURVEY 8(e)'s root fan-out/fan-in: rank 0 holds the whole batch, per step scatter_streams ->
                  processBlock -> gather_streams, double-buffered; frames/s beside the resident-input `value`
  configs2     -- BASELINE configs[2]: 256 streams, vocoder, lpcVoice 24, on the reference's 512/128 window and on the
                  metric's 1024/256 window
  configs3     -- the same hot path at BASELINE configs[3]'s per-GPU share (1024 streams, pitch + vocoder)
  configs4     -- BASELINE configs[4]'s per-GPU share: 48 kHz, 2048-pt frames hop 512, orders 48/48/30, 512 streams, both
                  processes (its frames are 512-sample hops)
  value_long   -- the headline workload again for about --long-seconds (6 s: a 20-step driver run times 1.3 ms; this is the stable
                  figure, and the stretch in which an outside observer's rocm-smi samples see the GPU busy)
The per-kernel durations (`kernel_us`, `roofline.avg_kernel_us`) are HIP events around every 8th launch of THE timed region;
`value_long` runs without any event record, and `roofline.step_us_without_events` is its time per step (one launch per step: an
upper bound of the kernel's duration that carries no event overhead).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 44100.0
HOP = 256
ALG_BYTES_PER_FRAME = {"pitch": 3072, "voc": 5120, "both": 5120}
PROFILE_EVERY = 8       # kernel durations are sampled live inside the timed region (two event records cost the stream 2-3 us)
# fp64 vector issue peak: 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz (an FMA counts once here: these are lane-operations)
VALU_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9
HBM_PEAK_GBS = 8000.0                                                  # MI355X_MICROARCH.md
UNIQUE_BLOCKS = 16                                                      # synthetic input ring, cycled
COUNTERS_JSON = os.path.join(ROOT, "profiles", "r06_counters.json")


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


def kernel_source_hash():
    """sha256 over what the kernels are built from: a counter file measured on other sources is not this build's."""
    h = hashlib.sha256()
    for rel in ("vocoderproject_amd/csrc/vp_kernels.hip", "vocoderproject_amd/csrc/vp_filters.inc", "vocoderproject_amd/csrc/vp_vocoder_wg.inc",
                "vocoderproject_amd/csrc/vp_pitch.inc", "vocoderproject_amd/csrc/vp_pitch_ws.inc", "vocoderproject_amd/csrc/vp_pitch_ws_body.inc", "vocoderproject_amd/csrc/vp_fft.inc", "vocoderproject_amd/csrc/vp_fft32.inc", "vocoderproject_amd/csrc/vp_voc2.hip", "vocoderproject_amd/csrc/vp_common.h",
                "vocoderproject_amd/csrc/vp_kernels.h", "vocoderproject_amd/csrc/vp_voc2.h", "vocoderproject_amd/csrc/vp_stft.hip",
                "vocoderproject_amd/csrc/vp_stft.h", "vocoderproject_amd/build.py"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def workload_key(cfg5, mode, mono, S, N, iir, yin, lpc_voice=None, voc_window=None, shift=None, bps=1):
    """Key of a (kernel, workload) pair in profiles/<tag>_counters.json; tools/summarize_counters.py builds the same one from the
    bench line's `config`.  The non-default vocoder order / window and the fixed shift are part of it: the profiled build must be
    the benched one (round-5 verdict, weak item 8)."""
    k = f"{'cfg5' if cfg5 else 'cfg'}/{mode}{'-mono' if mono else ''}/S{S}/N{N}/{iir}/{yin}"
    if lpc_voice:
        k += f"/lpcVoice{lpc_voice}"
    if voc_window:
        k += "/w" + voc_window.replace("/", "-")
    if shift is not None:
        k += f"/shift{shift:g}"
    if bps != 1:
        k += f"/bps{bps}"                     # blocks per call (vp_process_blocks*_device): a launch then covers that many blocks
    return k


def start_mix(F, H, N, n_blocks, first_block=0):
    """Frame starts (processChunkStart calls, PitchProcess.cpp:171-189) per host block, for blocks first_block .. first_block + n_blocks - 1
    since prepare: with the plugin's geometry (four chunk steps per 1024-sample block, a start every third step) blocks carry
    2, 1, 1, 2, 1, 1, ... starts, and a block with two starts costs the pitch kernel a third more.  Returns (period in blocks,
    [starts per block])."""
    C = F - H
    cpf = F // C
    per = []
    pS, nCh = 0, 0
    for b in range(first_block + n_blocks):
        n = 0
        while pS < N:
            if nCh == 0 or nCh == cpf - 1:
                n += 1
                nCh = 0
            nCh += 1
            pS += C
        pS -= N
        per.append(n)
    # period of the (pStart, nChunk) state
    import math
    steps_cycle = (cpf - 1) * C
    period = steps_cycle // math.gcd(steps_cycle, N)
    return period, per[first_block:]


def committed_counters(kernel, workload_key):
    """Per-launch PMC figures of `kernel` at `workload_key` from profiles/r06_counters.json (tools/collect_counters.sh:
    separate rocprofv3 --pmc passes of this very command); None unless kernel build name AND source hash match."""
    try:
        with open(COUNTERS_JSON) as f:
            t = json.load(f)
        if t.get("kernel_source_hash") != kernel_source_hash():
            return None
        return t["kernels"].get(f"{kernel}@{workload_key}")
    except (OSError, ValueError, KeyError):
        return None


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


def parity_vs_oracle(make, S_, N_, fs_, prepare, params, mono_, first_stream, iir, shift=None, streams=8, blocks=12, dev=None):
    """The accuracy half of the metric ("frames/s ...; RMS err vs CPU ref"), for ONE leg, outside every timed region: a FRESH
    processor of exactly the leg's configuration (`make()`: same mode switches, IIR/YIN mode, vocoder path, geometry, batch size
    -- hence the same kernel builds) takes `blocks` blocks of the leg's synthetic streams through the same device entry point
    the leg times; `streams` streams sampled across the batch are run through the CPU oracle on the same float32 input.
    Reports the per-sample RMS / max-abs error of the float32 output, the RMS error relative to the output's RMS, and the
    number of (stream, block) tracker states -- period, analysis and synthesis marks, beta of the block's last frame -- that
    differ from the oracle's (discrete decisions; SURVEY.md section 8d "Accuracy")."""
    import numpy as np
    import torch
    from oracle import oracle_py as O
    from vocoderproject_amd.synth import make_streams
    q = make()
    q.set_iir_mode(iir)
    pitch_on = bool(q.getParameter("pitchBool"))
    pick = sorted(set(int(round(i * (S_ - 1) / max(streams - 1, 1))) for i in range(min(streams, S_))))
    x = make_streams(S_, N_ * blocks, fs=fs_, first_stream=first_stream, device=dev)               # [S][3][blocks*N]
    if shift is not None:
        for s_ in range(S_):
            q.setPitchShift(shift if s_ % 2 == 0 else -shift, on=True, stream=s_)
    xb = x.view(S_, 3, blocks, N_).permute(2, 0, 1, 3).contiguous()                              # [blocks][S][3][N]
    xm = xb[:, :, 0, :].contiguous() if mono_ else None
    y = torch.empty((S_, 2, N_), dtype=torch.float32, device=dev)
    orc = []
    for s_ in pick:
        o = O.OracleStream(**dict(params, pitchBool=int(pitch_on), vocBool=int(q.getParameter("vocBool"))))
        if prepare:
            o.prepare_explicit(fs_, N_, *prepare)
        else:
            o.prepare_to_play(fs_, N_)
        if shift is not None:
            o.set_pitch_shift(shift if s_ % 2 == 0 else -shift, on=True)
        orc.append(o)
    xh = x[pick].cpu().numpy()
    got = np.empty((len(pick), 2, N_ * blocks), np.float32)
    ref = np.empty_like(got)
    frames = mism = 0
    st_cur = torch.cuda.current_stream(dev).cuda_stream
    for b in range(blocks):
        if mono_:
            q.process_mono_device(xm[b], y, st_cur)
        else:
            q.process_device(xb[b], y, st_cur)
        got[:, :, b * N_:(b + 1) * N_] = y[pick].cpu().numpy()
        for i, s_ in enumerate(pick):
            if mono_:
                ref[i, :, b * N_:(b + 1) * N_] = orc[i].process_block_mono(np.ascontiguousarray(xh[i, 0, b * N_:(b + 1) * N_]))
            else:
                io = np.ascontiguousarray(xh[i, :, b * N_:(b + 1) * N_])
                orc[i].process_block(io)
                ref[i, :, b * N_:(b + 1) * N_] = io[:2]
            if pitch_on:
                tr = orc[i].traces()
                if tr and not tr[-1]["gated"]:
                    st = q.pitch_state(s_)
                    f = tr[-1]
                    frames += 1
                    mism += int((st["period"], st["anMarks"], st["stMarks"], st["beta"]) != (f["period"], f["anMarks"], f["stMarks"], f["beta"]))
    err = got.astype(np.float64) - ref
    ref_rms = float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
    rms = float(np.sqrt((err ** 2).mean()))
    out = {"rms_err": rms, "max_abs_err": float(np.abs(err).max()), "rms_rel": rms / ref_rms if ref_rms > 0 else None, "ref_rms": ref_rms,
           "decision_mismatch_frames": mism, "decision_frames_compared": frames, "bit_identical": bool(np.array_equal(got, ref)),
           "streams": len(pick), "stream_ids": [first_stream + s_ for s_ in pick], "blocks": blocks, "iir_mode": iir,
           "kernel_builds": {"pitch": q.pitch_kernel_name() if pitch_on else None,
                             "vocoder": q.vocoder_kernel_name() if q.getParameter("vocBool") else None},
           "vs": "oracle/vp_oracle.c (CPU restatement of the reference processBlock), same float32 input, fresh state on both sides"}
    q.close()
    return out


# fp64 vector instructions one wavefront executes per frame in vp_k_stft_fused<false, false> (its loop body is straight-line: one pass
# per frame), counted in the disassembly of the built library by tools/kernel_resources.py fp64_op_counts();
# tests/test_kernel_resources.py fails when the build and these figures part.
STFT_FP64_OPS_PER_FRAME = {"v_mul_f64": 143, "v_add_f64": 338, "v_fma_f64": 40, "v_fmac_f64": 72}      # (round 5: the STFT unit is compiled with -ffp-contract=fast)
FP64_VECTOR_PEAK_TFLOPS = 78.6                                          # MI355X: 256 CUs x 4 SIMDs x 16 lanes/clk x 2 (FMA) x 2.4 GHz


def stft_figure(dev, S, T=1024 * 64, F=1024, hop=256, reps=30, pv=True, precision="f64"):
    """Standalone fused STFT->iSTFT kernel (Hann, FFT, iFFT, OLA in one launch; csrc/vp_stft.hip): NO reference counterpart, reported
    apart from the metric, with its own roofline: HBM on the algorithmic 2048 B per frame (every input
    sample in once, every output sample out once) and the fp64 vector share."""
    import torch
    from vocoderproject_amd import StftRoundTrip
    st = StftRoundTrip(S, T, F, hop, device=dev.index or 0)
    st.set_precision(precision)
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
    fps = frames / dt
    alg = 2 * hop * 4                                                   # bytes per frame: hop samples in, hop samples out, f32
    if precision == "f32":                                              # (--stft-only --stft-precision f32: the rocprofv3 passes of the single-precision build)
        return {"frames_per_s": fps, "kernel": "vp_k_stft_fused32<false>", "us_per_call": dt * 1e6, "dtype": "f32",
                "workload": f"{S} streams x {T} samples, {F}-pt frames hop {hop}, {st.n_frames} frames per stream, one launch per call, single precision",
                "roofline": {"bound": "hbm", "achieved": fps * alg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fps * alg / 1e9 / HBM_PEAK_GBS,
                             "alg_bytes_per_frame": alg, "alg_bytes_per_launch": alg * frames, "traffic": None}}
    n = STFT_FP64_OPS_PER_FRAME
    insts = sum(n.values())
    flops = (n["v_add_f64"] + n["v_mul_f64"] + 2 * (n["v_fmac_f64"] + n.get("v_fma_f64", 0))) * 64
    out = {"frames_per_s": fps, "kernel": "vp_k_stft_fused<false, false>" if st.fused else "vp_k_stft_frames + vp_k_stft_ola",
           "workload": f"{S} streams x {T} samples, {F}-pt frames hop {hop}, {st.n_frames} frames per stream, one launch per call",
           "us_per_call": dt * 1e6,
           "note": "standalone fused STFT round trip (sqrt-Hann, 512-pt complex register FFT per wavefront, iFFT, overlap-add in LDS); fp64; no reference counterpart",
           "roofline": {"bound": "hbm", "achieved": fps * alg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fps * alg / 1e9 / HBM_PEAK_GBS,
                        "alg_bytes_per_frame": alg, "alg_bytes_per_launch": alg * frames, "traffic": None},
           "fp64_valu": {"insts_per_frame_per_lane": insts, "flop_per_frame": flops, "tflops": fps * flops / 1e12,
                         "frac_of_peak_flops": fps * flops / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "frac_of_issue_slots": fps * insts * 64 / VALU_LANE_OPS_PEAK,
                         "what": "fp64 vector instructions per frame counted in the kernel's ISA (straight-line loop body) x frames/s, against "
                                 "78.6 TFLOP/s (FMA = 2) and against the fp64 issue slots (256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz)"}}
    # what really bounds the kernel: the issue of its fp64 vector instructions (HBM traffic is 1.03x algorithmic at 8 % of the peak)
    out["roofline"]["real_bound"] = {"bound": "valu_issue", "frac": out["fp64_valu"]["frac_of_issue_slots"],
                                     "what": "fp64 vector instructions per frame x frames/s against 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz"}
    ctr = committed_counters("vp_k_stft_fused<false, false>", f"stft/S{S}/T{T}/F{F}/hop{hop}")
    if ctr:                                                     # HBM bytes per launch from the committed PMC passes of this build (FETCH x 2 + WRITE)
        out["roofline"]["traffic"] = ctr.get("hbm_bytes_per_launch")
        out["roofline"]["rocprof_avg_us"] = ctr.get("rocprof_avg_us")
    if not pv:
        return out
    # BASELINE configs[4]'s geometry for this kernel: 2048-point frames, hop 512 (sixteen complex points per lane: vp_k_stft_fused2k)
    st2 = StftRoundTrip(S, T, 2048, 512, device=dev.index or 0)
    for _ in range(2):
        st2(x, y)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(max(4, reps // 3)):
        st2(x, y)
    torch.cuda.synchronize(dev)
    dt2 = (time.perf_counter() - t0) / max(4, reps // 3)
    out["frames_2048_hop_512"] = {"frames_per_s": S * st2.n_frames / dt2, "us_per_call": dt2 * 1e6, "kernel": "vp_k_stft_fused2k<false>",
                                  "hbm_gbs_algorithmic": S * st2.n_frames * 2 * 512 * 4 / dt2 / 1e9}
    st2.set_precision("f32")
    for _ in range(2):
        st2(x, y)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(max(4, reps // 3)):
        st2(x, y)
    torch.cuda.synchronize(dev)
    out["frames_2048_hop_512"]["single_precision_frames_per_s"] = S * st2.n_frames / ((time.perf_counter() - t0) / max(4, reps // 3))    # vp_k_stft_fused2k32
    st2.close()
    # the single-precision build of the same kernel (vp_stft_set_precision(VP_STFT_F32): transform, split and merge in f32; I/O is f32 either way)
    st.set_precision("f32")
    for _ in range(3):
        st(x, y)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        st(x, y)
    torch.cuda.synchronize(dev)
    dt3 = (time.perf_counter() - t0) / reps
    out["single_precision"] = {"frames_per_s": frames / dt3, "us_per_call": dt3 * 1e6, "kernel": "vp_k_stft_fused32<false>", "dtype": "f32",
                               "roofline": {"bound": "hbm", "achieved": frames / dt3 * alg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": frames / dt3 * alg / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_frame": alg},
                               "note": "opt-in; differs from the default (fp64) build by rounding only, ~1e-7 relative rms (tests/test_gpu_round4.py)"}
    ctr = committed_counters("vp_k_stft_fused32<false>", f"stft/S{S}/T{T}/F{F}/hop{hop}/f32")
    if ctr:
        out["single_precision"]["roofline"]["traffic"] = ctr.get("hbm_bytes_per_launch")
        out["single_precision"]["roofline"]["alg_bytes_per_launch"] = alg * frames
        out["single_precision"]["roofline"]["rocprof_avg_us"] = ctr.get("rocprof_avg_us")
    st.set_precision("f64")
    # the phase-vocoder stage between the transforms (vp_stft_pitch_shift, +7 semitones): one workgroup per stream
    if st.fused:
        for _ in range(2):
            st.pitch_shift(x, y, 7.0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(max(4, reps // 4)):
            st.pitch_shift(x, y, 7.0)
        torch.cuda.synchronize(dev)
        out["phase_vocoder_frames_per_s"] = frames / ((time.perf_counter() - t0) / max(4, reps // 4))
    return out


def count_gpus_without_hip():
    """GPUs this process may use, counted WITHOUT touching HIP (the parent of the rank processes must not initialise the
    runtime): the KFD topology lists every agent, GPU nodes are the ones with SIMDs; HIP_/ROCR_VISIBLE_DEVICES narrow it."""
    n = 0
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in sorted(os.listdir(top)):
            with open(os.path.join(top, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        return 0                                     # no KFD: no AMD GPU this process could open
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: N rank processes, one per GPU, started from a parent that never
    touches HIP (the GPUs are counted from the KFD topology in sysfs).  Rank 0's stdout is this process's stdout."""
    have = count_gpus_without_hip()
    if have < n:
        print(f"bench.py --gpus {n}: this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as s:                       # (a free port at this instant; the ranks rendezvous on it a moment later)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; without it RCCL's intra-node
        # transport (hipIpcGetMemHandle) fails with "invalid argument".  The image exports it already; kept explicit for
        # environments built from scratch.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 3600
    for p in procs:
        try:
            rc = max(rc, abs(p.wait(timeout=max(1.0, deadline - time.time()))))
        except subprocess.TimeoutExpired:
            rc = max(rc, 124)
    for p in procs:                       # our own children only, by pid
        if p.poll() is None:
            p.kill()
    return rc


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
                    help="do not time the other IIR mode, the STFT kernel or the secondary configurations (for profiler runs: one kernel population)")
    ap.add_argument("--yin", default="xcorr", choices=["direct", "xcorr", "fft"],
                    help="evaluation of the YIN difference function: 'xcorr' (default) is certified to take the same "
                         "decisions as 'direct' (the reference's arithmetic) and falls back to it otherwise: same output bits")
    ap.add_argument("--blocks-per-step", type=int, default=1,
                    help="host blocks of N samples handed over per step (vp_process_blocks_device)")
    ap.add_argument("--shift", type=float, default=None,
                    help="extension (no reference counterpart): fixed pitch-shift interval in semitones, +x on even and -x "
                         "on odd streams (vp_set_pitch_shift), instead of the correction to the key's nearest note")
    ap.add_argument("--three-channel", action="store_true",
                    help="pitch mode: hand over [S][3][N] buffers (voice + zero side chain) instead of the mono voice buffers "
                         "configs[1] describes (vp_process_block_mono_device; same output, a third of the input bytes)")
    ap.add_argument("--cfg5", action="store_true",
                    help="BASELINE configs[4] geometry instead of the default one: 48 kHz, 2048-point frames hop 512 (pitch "
                         "2048/1536, vocoder 2048/512), LPC orders 48/48/30, host block 2048 (a documentation figure: use with --mode both --no-cpu)")
    ap.add_argument("--exchange", action="store_true",
                    help="also time SURVEY 8(e)'s exchange: rank 0 holds the whole batch; per step scatter -> processBlock -> gather "
                         "(double-buffered, RCCL point-to-point).  On by default when N > 1.")
    ap.add_argument("--voc-path", default="auto", choices=["auto", "workgroup", "batched"],
                    help="vocoder implementation (vp_set_vocoder_path): one workgroup per stream, or the lane-per-window pipeline "
                         "(auto: the pipeline above 256 streams)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off"],
                    help="combined mode, FAST, batched vocoder: the pitch corrector beside the vocoder pipeline's tail instead of behind it "
                         "(vp_set_overlap; auto = the library's default: where the pitch build leaves registers beside it)")
    ap.add_argument("--lpc-voice", type=int, default=None,
                    help="lpcVoice (the plugin's default is 40; BASELINE configs[2] names 24)")
    ap.add_argument("--voc-window", default=None, choices=["512/128", "1024/256"],
                    help="vocoder window/hop: the reference's own 512/128 (default) or the metric's 1024/256 (SURVEY section 8, cfg 3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--stft-only", action="store_true",
                    help="time the standalone fused STFT kernel alone and print its figure (one kernel population: for the rocprofv3 passes)")
    ap.add_argument("--stft-precision", default="f64", choices=["f64", "f32"], help="with --stft-only: which build of the kernel")
    ap.add_argument("--no-parity", action="store_true", help="skip the per-leg comparison with the CPU oracle")
    ap.add_argument("--parity-streams", type=int, default=8, help="streams of each leg's batch that are run through the CPU oracle")
    ap.add_argument("--parity-blocks", type=int, default=12, help="blocks per stream of that comparison")
    ap.add_argument("--long-seconds", type=float, default=6.0,
                    help="value_long: the headline workload again for about this long (so that an outside observer -- the driver's "
                         "rocm-smi samples -- sees the GPU busy)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    # (VP_BENCH_SPAWN=1: take the spawning route even for one rank -- how the route is tested on a one-GPU box)
    if (args.gpus > 1 or os.environ.get("VP_BENCH_SPAWN") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))      # nothing above touched a GPU

    # stdout carries ONE JSON line and nothing else: libraries that write to the C-level stdout (RCCL prints a version
    # banner there, flushed when the process exits, i.e. after the line) get stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from vocoderproject_amd import BatchVocoderProcessor
    from vocoderproject_amd.dist import exchange_steps
    from vocoderproject_amd.synth import make_streams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    do_exchange = (args.exchange or world > 1) and not args.single_mode and args.blocks_per_step == 1
    use_dist = world > 1 or do_exchange or os.environ.get("VP_BENCH_FORCE_DIST") == "1"   # the latter: exercise the RCCL path on one GPU
    rccl = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
        rs = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
        dist.all_reduce(rs, op=dist.ReduceOp.SUM)
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank_sum": float(rs.item()),
                "rank_sum_expected": world * (world - 1) / 2.0}
    n_gpus = world

    if args.stft_only:
        fig = stft_figure(dev, args.streams, reps=max(args.steps, 10), pv=False, precision=args.stft_precision)
        T = 1024 * 64
        line = {"stft_only": True, "value": fig["frames_per_s"], "unit": "frames/s", "kernel_us": {fig["kernel"]: fig["us_per_call"]},
                "config": {"workload_key": f"stft/S{args.streams}/T{T}/F1024/hop256" + ("/f32" if args.stft_precision == "f32" else ""),
                           "workload": fig["workload"]}, "stft_kernel": fig}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        return
    S, N, mode = args.streams, args.block, args.mode
    global FS, HOP
    if args.cfg5:
        FS, HOP, N = 48000.0, 512, 2048

    # the headline workload's explicit geometry (None: what prepareToPlay picks) and parameters -- shared by the timed processor,
    # the parity check and the oracle
    if args.cfg5:
        hl_prepare, hl_params = (2048, 1536, 2048, 512), {"lpcVoice": 48, "lpcPitch": 48, "lpcSynth": 30}
    else:
        hl_prepare = (1024, 768, 1024, 256) if args.voc_window == "1024/256" else None   # the pitch geometry prepareToPlay(44100) picks + the metric's vocoder window
        hl_params = {"lpcVoice": args.lpc_voice} if args.lpc_voice else {}

    def make_processor(mode_, S_):
        q = BatchVocoderProcessor(device=local_rank, pitchBool=int(mode_ != "voc"), vocBool=int(mode_ != "pitch"), **hl_params)
        if hl_prepare:
            q.prepareExplicit(FS, N, S_, *hl_prepare)
        else:
            q.prepareToPlay(FS, N, S_)
        q.set_yin_mode(args.yin)
        q.set_vocoder_path(args.voc_path)
        q.set_overlap({"auto": "auto", "on": True, "off": False}[args.overlap])
        return q

    p = make_processor(mode, S)

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

    blocks_done = [0]                                  # host blocks the headline processor has taken since prepare

    def step(i):
        blocks_done[0] += BPS
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

    def region(fn, steps, warmup):
        """W untimed calls of fn, then exactly K timed ones between barrier + synchronize on both sides"""
        for i in range(warmup):
            fn(i)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            fn(warmup + i)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    def max_over_ranks(*vals):
        t = torch.tensor(list(vals), dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    # The pitch corrector's blocks are not all alike: with the plugin's geometry they carry 2, 1, 1, 2, 1, 1, ... frame starts
    # (start_mix), and a block with two starts costs a third more.  Every timed region of the headline therefore BEGINS at the same
    # phase of that cycle -- a block with the cycle's first (two-start) block -- by a few extra untimed steps behind the W warm-up
    # ones, and the line states the mix of the K timed blocks beside the long-run mix (round-5 verdict, weak item 9).
    geom = p.geometry()
    mix_period = start_mix(geom["F"], geom["H"], N, 1)[0] if mode != "voc" else 1
    align_info = {"period_blocks": mix_period, "extra_untimed_steps": 0}

    def align_phase():
        n = 0
        while mix_period > 1 and mix_period <= 64 and BPS == 1 and blocks_done[0] % mix_period != 0:
            step(blocks_done[0])
            n += 1
        return n

    def timed(mode_iir, steps, warmup):
        p.set_iir_mode(mode_iir)
        for i in range(warmup):
            step(i)
        align_info["extra_untimed_steps"] = align_phase()
        torch.cuda.synchronize(dev)
        p.profile_read(reset=True)
        p.profile_enable(PROFILE_EVERY)       # HIP events around every PROFILE_EVERY-th launch of the timed region
        dt_ = region(lambda i: step(warmup + i), steps, 0)
        p.profile_enable(False)
        return dt_, p.profile_read(reset=True)

    # secondary figure first (shorter), then THE timed region: W warmup steps, exactly K timed steps
    other = "exact" if args.iir == "fast" else "fast"
    k2 = max(10, args.steps // 4)
    dt_other = float("nan")
    if not args.single_mode:
        dt_other, _ = timed(other, k2, max(2, args.warmup // 4))
    # a second secondary figure: the same blocks handed over eight at a time (vp_process_blocks_device: one launch per eight
    # blocks in pitch mode, state stays on chip in between; one launch of the lane-per-window pipeline with eight times the
    # windows in vocoder mode); reported beside `value`, never as it
    MB, k3 = 8, max(4, args.steps // 16)
    dt_mb = float("nan")
    if not args.single_mode and BPS == 1 and mode in ("pitch", "voc") and U % MB == 0:
        ymb = torch.empty((MB, S, 2, N), dtype=torch.float32, device=dev)
        p.set_iir_mode(args.iir)

        def step_mb(i):
            blocks_done[0] += MB
            b0 = (i * MB) % U
            if mono:
                p.process_blocks_mono_device(xm[b0:b0 + MB], ymb, stream.cuda_stream)
            else:
                p.process_blocks_device(x[b0:b0 + MB], ymb, stream.cuda_stream)

        dt_mb = region(step_mb, k3, 2)
    # the headline workload over a long stretch without any event record, BEFORE the timed region: the stable figure beside a
    # 20-step driver run, the stretch an outside observer (rocm-smi samples) sees the GPU busy in, and what brings the chip to the
    # clocks it holds under sustained load before the K timed steps are taken
    frames_per_step_gpu = S * N * BPS // HOP
    value_long = None
    if not args.single_mode and BPS == 1:
        p.set_iir_mode(args.iir)
        region(step, 50, 8)
        torch.cuda.synchronize(dev)
        t_probe = time.perf_counter()
        region(step, 200, 0)
        (per_step,) = max_over_ranks((time.perf_counter() - t_probe) / 200)
        kl = max(2000, args.steps, int(args.long_seconds / max(per_step, 1e-6)))
        dtl = region(step, kl, 8)
        (dtl,) = max_over_ranks(dtl)
        value_long = {"value": frames_per_step_gpu * kl * n_gpus / dtl, "steps": kl, "ms_per_step": dtl / kl * 1e3}

    dt, prof = timed(args.iir, args.steps, args.warmup)

    chk = y.double().abs().sum().view(1)
    if use_dist:
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)          # a checksum of the outputs (not on the data path)
    dt, dt_other, dt_mb = max_over_ranks(dt, dt_other, dt_mb)

    # a third secondary figure, AFTER the headline region: BASELINE configs[1]'s "+-12-semitone pitch shift" (fixed
    # interval, +12 on even and -12 on odd streams; extension without a reference counterpart, parity GPU <-> oracle)
    dt_shift = float("nan")
    if not args.single_mode and args.shift is None and mode != "voc" and BPS == 1:
        set_shift(12.0)
        dt_shift, _ = timed(args.iir, k2, max(4, args.warmup // 2))
        (dt_shift,) = max_over_ranks(dt_shift)
        for s_ in range(S):
            p.setPitchShift(0.0, on=False, stream=s_)

    # The accuracy half of the metric for the headline workload (and for the secondary modes timed above), on rank 0, outside the
    # timed regions: sampled streams of a fresh processor of the same configuration against the CPU oracle.
    parity = parity_other = parity_shift = None
    if rank == 0 and not args.no_parity and BPS == 1:
        def mk_hl():
            return make_processor(mode, S)
        pk = dict(dev=dev, streams=args.parity_streams, blocks=min(args.parity_blocks, U))
        parity = parity_vs_oracle(mk_hl, S, N, FS, hl_prepare, hl_params, mono, rank * S, args.iir, shift=args.shift, **pk)
        if not args.single_mode:
            parity_other = parity_vs_oracle(mk_hl, S, N, FS, hl_prepare, hl_params, mono, rank * S, other, shift=args.shift, **pk)
            if dt_shift == dt_shift:
                parity_shift = parity_vs_oracle(mk_hl, S, N, FS, hl_prepare, hl_params, mono, rank * S, args.iir, shift=12.0, **pk)

    # the batch lives on rank 0; per step root fan-out (scatter_streams), processBlock on every rank, root
    # fan-in (gather_streams).  Double-buffered: step i+1's scatter and step i-1's gather ride RCCL's stream beside step
    # i's kernels (exchange_steps enqueues on torch's CURRENT stream: the processor is handed that same stream).
    def exchange_region(q, S_, N_, fs_, hop_, mono_, ke):
        C_in = 1 if mono_ else 3
        tail_in = (N_,) if mono_ else (3, N_)
        Sg = S_ * world
        xr = yr = None
        if rank == 0:
            xr = make_streams(Sg, N_ * 4, fs=fs_, first_stream=0, device=dev).view(Sg, 3, 4, N_).permute(2, 0, 1, 3)
            xr = (xr[:, :, 0, :] if mono_ else xr).contiguous()                  # [4][Sg][N] or [4][Sg][3][N]
            yr = [torch.empty((Sg, 2, N_), dtype=torch.float32, device=dev) for _ in range(2)]

        def run(steps):
            def proc(i_, o_):
                if mono_:
                    q.process_mono_device(i_, o_, stream.cuda_stream)
                else:
                    q.process_device(i_, o_, stream.cuda_stream)
            exchange_steps(steps, Sg, (lambda i: xr[i % 4]), (lambda i: yr[i & 1]), tail_in, (2, N_), torch.float32, dev, proc)

        run(4)
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(ke)
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        (dte,) = max_over_ranks(time.perf_counter() - t0)
        return {"value": (S_ * N_ // hop_) * ke * n_gpus / dte, "unit": "frames/s", "ms_per_step": dte / ke * 1e3, "steps": ke,
                "bytes_scattered_per_step": (Sg - S_) * C_in * N_ * 4, "bytes_gathered_per_step": (Sg - S_) * 2 * N_ * 4,
                "what": "rank 0 holds the batch: scatter_streams -> processBlock -> gather_streams per step, double-buffered, "
                        "RCCL point-to-point (batch_isend_irecv); root's own shard is a local copy"}

    exch = None
    if do_exchange:
        p.set_iir_mode(args.iir)
        exch = exchange_region(p, S, N, FS, HOP, mono, max(16, args.steps // 2))

    # The other BASELINE configs at their per-GPU share, so that one driver run (at every N) carries a figure for each of them.
    def leg(what, mode_, S_, fs_, N_, hop_, prepare, params, steps_, with_exchange=False, blocks=0, exact_too=True, mono_=False):
        def mk():
            q_ = BatchVocoderProcessor(device=local_rank, pitchBool=int(mode_ != "voc"), vocBool=int(mode_ != "pitch"), **params)
            if prepare:
                q_.prepareExplicit(fs_, N_, S_, *prepare)
            else:
                q_.prepareToPlay(fs_, N_, S_)
            q_.set_yin_mode(args.yin)
            return q_

        q = mk()
        q.set_iir_mode(args.iir)
        xl = make_streams(S_, N_ * 4, fs=fs_, first_stream=rank * S_, device=dev).view(S_, 3, 4, N_).permute(2, 0, 1, 3).contiguous()
        yl = torch.empty((S_, 2, N_), dtype=torch.float32, device=dev)
        xlm = xl[:, :, 0, :].contiguous() if mono_ else None

        def run_leg(i):
            if mono_:
                q.process_mono_device(xlm[i % 4], yl, stream.cuda_stream)
            else:
                q.process_device(xl[i % 4], yl, stream.cuda_stream)

        dtl_ = region(run_leg, steps_, 12)
        (dtl_,) = max_over_ranks(dtl_)
        out_ = {"workload": what, "value": (S_ * N_ // hop_) * steps_ * n_gpus / dtl_, "unit": "frames/s", "frame_hop": hop_,
                "ms_per_step": dtl_ / steps_ * 1e3, "steps": steps_, "streams_per_gpu": S_, "mode": mode_, "iir_mode": args.iir,
                "alg_bytes_per_step_per_gpu": (S_ * N_ * 4 * (1 + 2) if (mode_ == "pitch" and mono_) else ALG_BYTES_PER_FRAME[mode_] * S_ * N_ // 256),
                "kernel_builds": {"pitch": q.pitch_kernel_name() if mode_ != "voc" else None,
                                  "vocoder": q.vocoder_kernel_name() if mode_ != "pitch" else None}}
        out_["hbm_frac_algorithmic"] = out_["alg_bytes_per_step_per_gpu"] / (dtl_ / steps_) / 1e9 / HBM_PEAK_GBS
        if exact_too:                                  # the bit-identical mode's figure for this configuration
            q.set_iir_mode(other)
            ke = max(12, steps_ // 3)
            dte_ = region(run_leg, ke, 2)
            (dte_,) = max_over_ranks(dte_)
            out_[f"value_{other}_mode"] = (S_ * N_ // hop_) * ke * n_gpus / dte_
            q.set_iir_mode(args.iir)
        if rank == 0 and not args.no_parity:           # the accuracy half of the metric, in the mode and on the builds just timed
            out_["parity"] = parity_vs_oracle(mk, S_, N_, fs_, prepare, params, mono_, rank * S_, args.iir, dev=dev,
                                              streams=args.parity_streams, blocks=args.parity_blocks)
        if do_exchange and with_exchange:
            out_["exchange"] = exchange_region(q, S_, N_, fs_, hop_, False, max(8, steps_ // 2))
        if blocks > 1:                              # the same blocks handed over `blocks` at a time (vp_process_blocks_device)
            xb = xl.repeat(blocks // 4 + 1, 1, 1, 1)[:blocks].contiguous()
            yb = torch.empty((blocks, S_, 2, N_), dtype=torch.float32, device=dev)
            kb = max(6, steps_ // blocks)
            dtb = region(lambda i: q.process_blocks_device(xb, yb, stream.cuda_stream), kb, 2)
            (dtb,) = max_over_ranks(dtb)
            out_[f"value_{blocks}_blocks_per_call"] = (S_ * N_ // hop_) * blocks * kb * n_gpus / dtb
            del xb, yb
        del q, xl, yl
        return out_

    cfg2 = cfg3 = cfg4 = fs48 = lpc24 = None
    k4 = max(120, args.steps // 2)                     # (a leg is tens of milliseconds: long enough for a stable figure whatever --steps is)
    if not args.single_mode and not args.cfg5 and BPS == 1:
        if not (mode == "voc" and args.lpc_voice == 24):
            cfg2 = {"window_512_128": leg("configs[2]: 256 streams, vocoder, lpcVoice 24, the reference's 512/128 window", "voc", 256, 44100.0, 1024, 256,
                                          None, {"lpcVoice": 24}, k4),
                    "window_1024_256": leg("configs[2]: 256 streams, vocoder, lpcVoice 24, the metric's 1024/256 window", "voc", 256, 44100.0, 1024, 256,
                                           (1024, 768, 1024, 256), {"lpcVoice": 24}, k4)}
        if not (mode == "both" and S == 1024):
            cfg3 = leg("configs[3] per GPU: 1024 streams, pitch corrector + vocoder", "both", 1024, 44100.0, 1024, 256, None, {}, k4, with_exchange=True, blocks=8)
        cfg4 = leg("configs[4] per GPU: 512 streams @48 kHz, 2048-pt frames hop 512, orders 48/48/30, pitch corrector + vocoder", "both", 512,
                   48000.0, 2048, 512, (2048, 1536, 2048, 512), {"lpcVoice": 48, "lpcPitch": 48, "lpcSynth": 30}, max(64, args.steps // 3))

        # configs[1] with lpcPitch 24 (cfg 2: "pP = 15 (also 24)"): the wave-specialised kernel's _o24 build (round 6)
        lpc24 = leg("configs[1] with lpcPitch 24: 256 mono streams, pitch corrector", "pitch", 256, 44100.0, 1024, 256, None, {"lpcPitch": 24}, k4, mono_=True)

        # The reference's OTHER real geometry (round-5 verdict, item 3): what prepareToPlay(48000, 1024) picks (PluginProcessor.cpp:159-173:
        # pitch frames 1112 / hop 834 / chunk 278, vocoder window 556 / hop 139), 256 streams; a frame here is one 278-sample chunk.
        fs48 = {"pitch": leg("prepareToPlay(48000, 1024): 256 mono streams, pitch corrector, frames 1112/834 (chunk 278)", "pitch", 256, 48000.0, 1024, 278,
                             None, {}, k4, mono_=True),
                "both": leg("prepareToPlay(48000, 1024): 256 streams, pitch corrector (1112/834) + vocoder (556/139)", "both", 256, 48000.0, 1024, 278,
                            None, {}, k4)}

    total_frames = frames_per_step_gpu * args.steps * n_gpus
    value = total_frames / dt

    if rank == 0:
        dom = "vp_k_vocoder" if mode == "voc" else "vp_k_pitch"
        # the dominant kernel by measured time
        dom = max(prof.items(), key=lambda kv: kv[1][0])[0] if prof else dom
        ms, n = prof[dom]
        avg_s = (ms / max(n, 1)) * 1e-3
        dom_build = p.vocoder_kernel_name() if dom == "vp_k_vocoder" else p.pitch_kernel_name() if dom == "vp_k_pitch" else dom
        alg_bytes = ALG_BYTES_PER_FRAME[mode] * (HOP // 256) * frames_per_step_gpu      # f32 I/O per hop-frame (hop 512: twice the samples)
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        wkey = workload_key(args.cfg5, mode, mono, S, N, args.iir, args.yin, args.lpc_voice, args.voc_window, args.shift, BPS)
        ctr = committed_counters(dom_build, wkey) if BPS == 1 else None
        roof = {"bound": "hbm", "kernel": dom_build, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": ctr["hbm_bytes_per_launch"] if ctr else None,
                "avg_kernel_us": avg_s * 1e6, "alg_bytes_per_launch": alg_bytes,
                "note": "path is fp64-VALU/latency-bound (DESIGN.md); HBM fraction is reported as the contract asks"}
        if ctr:
            # what the kernel EXECUTED (rocprofv3 --pmc SQ_INSTS_VALU of this build, per launch), against the fp64 vector issue peak
            lane_ops = ctr["SQ_INSTS_VALU"] * 64.0
            roof["valu"] = {"insts_per_launch": ctr["SQ_INSTS_VALU"], "lane_ops_per_s": lane_ops / avg_s, "peak_lane_ops_per_s": VALU_LANE_OPS_PEAK,
                            "frac": lane_ops / avg_s / VALU_LANE_OPS_PEAK, "valu_busy_frac_of_wave_cycles": ctr.get("valu_active_over_wave_cycles"),
                            "what": "executed vector instructions x 64 lanes per second vs 256 CUs x 4 SIMDs x 16 fp64 lanes/clk x 2.4 GHz "
                                    "(upper bound on useful work: serial phases run all 64 lanes redundantly)"}
        out = {
            "metric": "STFT-geometry frames/sec (1024-pt frames, hop 256) through the pitch-corrector/vocoder path",
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"configs[4] geometry: {S} streams per GPU @48 kHz, 2048-pt frames hop 512, LPC orders 48/48/30, mode {mode}, host block N={N}" if args.cfg5 else "") or f"configs[{1 if mode == 'pitch' else 2 if mode == 'voc' else 3}]: {S} mono streams per GPU @44.1 kHz, "
                                   f"{'pitch corrector (YIN+PSOLA on LPC residual, key=Chrom)' if mode == 'pitch' else 'LPC vocoder' if mode == 'voc' else 'pitch corrector + vocoder'}"
                                   f", 1024-pt frames hop 256, host block N={N}" + (f", {BPS} blocks per step" if BPS > 1 else "")
                                   + (f", lpcVoice {args.lpc_voice}" if args.lpc_voice else "") + (f", vocoder window {args.voc_window}" if args.voc_window else ""),
                       "streams_per_gpu": S, "block": N, "blocks_per_step": BPS, "mode": mode, "iir_mode": args.iir, "yin_mode": args.yin, "fixed_shift_semitones": args.shift, "lpc_voice": args.lpc_voice, "voc_window": args.voc_window, "workload_key": workload_key(args.cfg5, mode, mono, S, N, args.iir, args.yin, args.lpc_voice, args.voc_window, args.shift, BPS), "input": "mono voice [S][N]" if mono else "[S][3][N]", "frames_per_step": frames_per_step_gpu * n_gpus,
                       "kernel_builds": {"pitch": p.pitch_kernel_name() if mode != "voc" else None, "vocoder": p.vocoder_kernel_name() if mode != "pitch" else None},
                       "kernel_source_hash": kernel_source_hash(),
                       "parallelism": f"streams sharded over {n_gpus} GPU(s), one process per GPU, no data-path collective"},
            "roofline": roof,
            "kernel_us": {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof.items() if v[1]},
            "parity": parity,
            f"parity_{other}_mode": parity_other,
            "parity_pm12_semitone_shift": parity_shift,
            "checksum": float(chk.item()),
            f"value_{other}_mode": (frames_per_step_gpu * k2 * n_gpus / dt_other) if dt_other == dt_other else None,
            "value_8_blocks_per_call": ((S * N * MB // HOP) * k3 * n_gpus / dt_mb) if dt_mb == dt_mb else None,
            "value_pm12_semitone_shift": (frames_per_step_gpu * k2 * n_gpus / dt_shift) if dt_shift == dt_shift else None,
        }
        if rccl:
            out["rccl"] = rccl
        if exch:
            out["exchange"] = exch
        if value_long:
            out["value_long"] = value_long
            roof["step_us_without_events"] = value_long["ms_per_step"] * 1e3
        if cfg2:
            out["configs2"] = cfg2
        if cfg3:
            out["configs3"] = cfg3
        if cfg4:
            out["configs4"] = cfg4
        if fs48:
            out["fs48k"] = fs48
        if lpc24:
            out["configs1_lpcPitch24"] = lpc24
        if not args.single_mode:
            out["stft_kernel"] = stft_figure(dev, S)
        if n_gpus == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(mode, N, args.cpu_seconds)

        # The certificate in brief, INSIDE the objects every consumer of the line keeps (`config`, `roofline`): per leg the accuracy half
        # of the metric -- rms / max-abs error against the CPU oracle, tracker states that differ -- and the long-run figure.
        def brief(pr):
            return None if not pr else {"rms_err": pr["rms_err"], "max_abs_err": pr["max_abs_err"], "decision_mismatch_frames": pr["decision_mismatch_frames"],
                                        "bit_identical": pr["bit_identical"], "streams": pr["streams"], "blocks": pr["blocks"]}
        legs = {"headline": brief(parity), f"headline_{other}_mode": brief(parity_other), "headline_pm12_semitones": brief(parity_shift)}
        if cfg2:
            legs["configs2_512_128"] = brief(cfg2["window_512_128"].get("parity"))
            legs["configs2_1024_256"] = brief(cfg2["window_1024_256"].get("parity"))
        if cfg3:
            legs["configs3"] = brief(cfg3.get("parity"))
        if cfg4:
            legs["configs4"] = brief(cfg4.get("parity"))
        if lpc24:
            legs["lpcPitch24"] = brief(lpc24.get("parity"))
        if fs48:
            legs["fs48k_pitch"] = brief(fs48["pitch"].get("parity"))
            legs["fs48k_both"] = brief(fs48["both"].get("parity"))
        legs = {k: v for k, v in legs.items() if v}
        out["config"]["parity_vs_cpu_oracle"] = legs
        out["config"]["parity_rms_err_max"] = max((v["rms_err"] for v in legs.values()), default=None)
        out["config"]["value_long"] = value_long["value"] if value_long else None
        # ... and as SCALAR keys of `config` (a consumer that keeps only scalars keeps every leg's value and error; round-5 verdict item 5)
        cf = out["config"]
        for k_, v_ in legs.items():
            cf[f"parity_{k_}_rms"] = v_["rms_err"]
            cf[f"parity_{k_}_mismatch_frames"] = v_["decision_mismatch_frames"]
        cf["value_exact_mode" if other == "exact" else "value_fast_mode"] = out.get(f"value_{other}_mode")
        cf["value_8_blocks_per_call"] = out.get("value_8_blocks_per_call")
        cf["value_pm12_semitone_shift"] = out.get("value_pm12_semitone_shift")
        if cfg2:
            cf["configs2_512_128_value"] = cfg2["window_512_128"]["value"]
            cf["configs2_1024_256_value"] = cfg2["window_1024_256"]["value"]
            cf["configs2_kernel"] = cfg2["window_512_128"]["kernel_builds"]["vocoder"]
        if cfg3:
            cf["configs3_value"] = cfg3["value"]
            cf["configs3_value_exact_mode"] = cfg3.get("value_exact_mode")
            cf["configs3_value_8_blocks_per_call"] = cfg3.get("value_8_blocks_per_call")
            cf["configs3_pitch_kernel"] = cfg3["kernel_builds"]["pitch"]
        if cfg4:
            cf["configs4_value"] = cfg4["value"]
            cf["configs4_value_exact_mode"] = cfg4.get("value_exact_mode")
            cf["configs4_pitch_kernel"] = cfg4["kernel_builds"]["pitch"]
        if lpc24:
            cf["lpcPitch24_value"] = lpc24["value"]
            cf["lpcPitch24_value_exact_mode"] = lpc24.get("value_exact_mode")
            cf["lpcPitch24_kernel"] = lpc24["kernel_builds"]["pitch"]
        if fs48:
            cf["fs48k_pitch_value"] = fs48["pitch"]["value"]
            cf["fs48k_pitch_kernel"] = fs48["pitch"]["kernel_builds"]["pitch"]
            cf["fs48k_both_value"] = fs48["both"]["value"]
            cf["fs48k_frame_hop"] = 278
        sk = out.get("stft_kernel")
        if sk:
            cf["stft_frames_per_s"] = sk["frames_per_s"]
            cf["stft_hbm_frac"] = sk["roofline"]["frac"]
            cf["stft_valu_issue_frac"] = sk.get("fp64_valu", {}).get("frac_of_issue_slots")
            cf["stft_f32_frames_per_s"] = sk.get("single_precision", {}).get("frames_per_s")
            cf["pv_frames_per_s"] = sk.get("phase_vocoder_frames_per_s")
        # the K timed blocks' mix of one- and two-start blocks beside the long-run mix (the timed region begins at a fixed phase)
        if mode != "voc" and BPS == 1:
            per_, mix_ = start_mix(geom["F"], geom["H"], N, args.steps, 0)
            two_ = sum(1 for m_ in mix_ if m_ >= 2) / max(len(mix_), 1)
            _, cyc_ = start_mix(geom["F"], geom["H"], N, max(per_, 1), 0)
            cf["timed_blocks_with_two_starts_frac"] = two_
            cf["long_run_blocks_with_two_starts_frac"] = sum(1 for m_ in cyc_ if m_ >= 2) / max(len(cyc_), 1)
            cf["timed_region_phase"] = dict(align_info, begins_at="the cycle's first block (two frame starts)")
            cf["value_over_value_long"] = (value / value_long["value"]) if value_long else None
        roof["parity_rms_err"] = parity["rms_err"] if parity else None
        roof["parity_decision_mismatch_frames"] = parity["decision_mismatch_frames"] if parity else None
        # the contract's own keys LAST: whoever keeps only the tail of the line keeps them (and the roofline / cpu_baseline / config objects)
        tail_keys = ["roofline", "cpu_baseline", "config", "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                     "scaling", "vs_baseline", "dtype", "data"]
        out = {**{k: v for k, v in out.items() if k not in tail_keys}, **{k: out[k] for k in tail_keys if k in out}}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
