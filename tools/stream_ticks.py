#!/usr/bin/env python3
"""Diagnostic (-DVP_STAMPS library): microseconds each stream's workgroup spends inside the pitch kernel per block --
which streams pace a launch, and what distinguishes them."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VP_AMD_LIB", os.path.join(ROOT, "vocoderproject_amd", "libvp_amd_stamps.so"))
import numpy as np, torch
from vocoderproject_amd import BatchVocoderProcessor
from vocoderproject_amd.synth import make_streams
S, N, U, steps = 256, 1024, 16, 96
p = BatchVocoderProcessor(vocBool=0); p.prepareToPlay(44100.0, N, S); p.set_iir_mode("fast"); p.set_yin_mode(sys.argv[1] if len(sys.argv) > 1 else "xcorr")
x = make_streams(S, N * U, device="cuda").view(S, 3, U, N).permute(2, 0, 1, 3).contiguous(); y = torch.empty((S, 2, N), device="cuda")
buf = (C.c_ulonglong * S)()
for i in range(16): p.process_device(x[i % U], y)
p.L.vp_debug_read_stream_ticks(p.h, buf, S, 1)
for i in range(steps): p.process_device(x[(16 + i) % U], y)
p.L.vp_debug_read_stream_ticks(p.h, buf, S, 1)
t = np.array(buf[:], dtype=np.float64) / 100.0 / steps
per = [p.pitch_state(s)["period"] for s in range(S)]
order = np.argsort(t)
print(f"per-block microseconds in the kernel: min {t.min():.1f}  median {np.median(t):.1f}  p90 {np.percentile(t, 90):.1f}  max {t.max():.1f}")
print("slowest streams (stream, us, last period):", [(int(s), round(float(t[s]), 1), per[s]) for s in order[-8:]])
print("fastest streams (stream, us, last period):", [(int(s), round(float(t[s]), 1), per[s]) for s in order[:8]])
pp = np.array(per, dtype=float)
ok = pp > 0
print("correlation of time with period:", round(float(np.corrcoef(t[ok], pp[ok])[0, 1]), 3))
