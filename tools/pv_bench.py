#!/usr/bin/env python3
"""Diagnostic: the phase-vocoder stage (vp_stft_pitch_shift) at the bench's shape, a few intervals.

    [VP_AMD_LIB=...] python tools/pv_bench.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from vocoderproject_amd import StftRoundTrip
    S, T, F, hop = 256, 65536, 1024, 256
    x = torch.randn(S, T, device="cuda", dtype=torch.float32) * 0.1
    y = torch.empty_like(x)
    st = StftRoundTrip(S, T, F, hop)
    frames = S * ((T - F) // hop + 1)
    for semis in (7.0, 12.0, -5.0):
        for _ in range(3):
            st.pitch_shift(x, y, semis)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            st.pitch_shift(x, y, semis)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{semis:+5.1f} semitones: {frames / dt / 1e6:7.1f} M frames/s  {dt * 1e6:8.1f} us per call")
    st.close()


if __name__ == "__main__":
    main()
