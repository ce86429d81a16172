"""Synthetic 44.1/48 kHz streams for tests and bench.

Per stream s (seed 0x5EED0000 + s): voice = 0.25 * sum_{h=1..12} sin(2 pi h phi)/h with
f0 ~ U[110, 380] Hz, 5 Hz vibrato of depth 1 %, + N(0, 0.002^2) noise; carrier L = R = three
band-unlimited saws (110 / 138.59 / 164.81 Hz * 2^U[-1,1]) * 0.15.  float32, planar
[stream][channel][sample] with channel 0 = voice, 1 = carrier L, 2 = carrier R -- the layout of the
reference's in-place processBlock() buffer (PluginProcessor.cpp:19-23, 209-210).

The same array is handed to the GPU path and to the CPU oracle, so "identical inputs" holds by
construction.  Synthesis runs in torch (CPU or GPU) in float64 and is rounded to float32 once.
"""
import math

import numpy as np
import torch

SEED_BASE = 0x5EED0000
SAW_BASE = (110.0, 138.59, 164.81)


def stream_params(s):
    """Random per-stream constants (host side, numpy PCG64)."""
    rng = np.random.default_rng(SEED_BASE + int(s))
    f0 = rng.uniform(110.0, 380.0)
    vib_phase = rng.uniform(0.0, 2.0 * math.pi)
    octave = rng.uniform(-1.0, 1.0)
    saw_phase = rng.uniform(0.0, 1.0, size=3)
    noise_seed = int(rng.integers(0, 2**31 - 1))
    return f0, vib_phase, octave, saw_phase, noise_seed


def make_streams(n_streams, n_samples, fs=44100.0, first_stream=0, device="cpu", t0=0,
                 voice_gain=1.0, carrier_gain=1.0):
    """Returns float32 tensor [n_streams][3][n_samples] on `device`."""
    dev = torch.device(device)
    S = n_streams
    prm = [stream_params(first_stream + s) for s in range(S)]
    f0 = torch.tensor([p[0] for p in prm], dtype=torch.float64, device=dev).view(S, 1)
    vph = torch.tensor([p[1] for p in prm], dtype=torch.float64, device=dev).view(S, 1)
    octv = torch.tensor([p[2] for p in prm], dtype=torch.float64, device=dev).view(S, 1)
    sph = torch.tensor(np.stack([p[3] for p in prm]), dtype=torch.float64, device=dev)  # [S,3]

    out = torch.empty((S, 3, n_samples), dtype=torch.float32, device=dev)
    # chunk over time to bound memory: S x chunk x float64 temporaries
    chunk = max(1024, min(n_samples, (1 << 24) // max(S, 1)))
    vib_f, depth = 5.0, 0.01
    for c0 in range(0, n_samples, chunk):
        c1 = min(n_samples, c0 + chunk)
        t = (torch.arange(c0 + t0, c1 + t0, dtype=torch.float64, device=dev) / fs).view(1, -1)
        # phase of a tone whose instantaneous frequency is f0 (1 + depth sin(2 pi vib_f t + vph))
        phi = f0 * (t - depth / (2 * math.pi * vib_f) * (torch.cos(2 * math.pi * vib_f * t + vph) - torch.cos(vph)))
        v = torch.zeros_like(phi)
        for h in range(1, 13):
            v += torch.sin(2 * math.pi * h * phi) / h
        v *= 0.25
        out[:, 0, c0:c1] = (v * voice_gain).to(torch.float32)
        car = torch.zeros_like(phi)
        for j, fb in enumerate(SAW_BASE):
            fj = fb * torch.pow(torch.tensor(2.0, dtype=torch.float64, device=dev), octv)
            ph = fj * t + sph[:, j:j + 1]
            car += 2.0 * (ph - torch.floor(ph)) - 1.0
        car *= 0.15 * carrier_gain
        cf = car.to(torch.float32)
        out[:, 1, c0:c1] = cf
        out[:, 2, c0:c1] = cf
    # additive noise, per-stream generator so a stream's samples do not depend on the batch it is in
    for s in range(S):
        g = torch.Generator(device="cpu")
        g.manual_seed(prm[s][4] + t0)
        nz = torch.randn(n_samples, generator=g, dtype=torch.float32) * 0.002 * voice_gain
        out[s, 0] += nz.to(dev)
    return out
