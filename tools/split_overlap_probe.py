#!/usr/bin/env python3
"""Probe (round 3): does a large batch run faster as TWO half batches on two HIP streams (the halves' kernels interleave:
one half's latency-bound recursion beside the other half's throughput-bound autocorrelation) than as one?
    python tools/split_overlap_probe.py [--mode voc|both|pitch] [--streams 1024]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vocoderproject_amd import BatchVocoderProcessor
from vocoderproject_amd.synth import make_streams

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="voc")
ap.add_argument("--streams", type=int, default=1024)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--parts", type=int, default=2)
a = ap.parse_args()
N, S = 1024, a.streams
dev = torch.device("cuda", 0)


def mk(S_):
    p = BatchVocoderProcessor(pitchBool=int(a.mode != "voc"), vocBool=int(a.mode != "pitch"))
    p.prepareToPlay(44100.0, N, S_)
    p.set_iir_mode("fast")
    p.set_yin_mode("xcorr")
    return p


def run(parts):
    Sp = S // parts
    ps = [mk(Sp) for _ in range(parts)]
    xs = [make_streams(Sp, N * 4, first_stream=i * Sp, device=dev).view(Sp, 3, 4, N).permute(2, 0, 1, 3).contiguous() for i in range(parts)]
    ys = [torch.empty((Sp, 2, N), dtype=torch.float32, device=dev) for _ in range(parts)]
    sts = [torch.cuda.Stream(dev) for _ in range(parts)]

    def step(i):
        for k in range(parts):
            ps[k].process_device(xs[k][i % 4], ys[k], sts[k].cuda_stream)
    for i in range(10):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / a.steps
    print(f"mode={a.mode} S={S} as {parts} part(s) on {parts} stream(s): {dt * 1e6:.1f} us per block, {S * N / 256 / dt / 1e6:.2f} M frames/s")


run(1)
run(a.parts)
run(4)
