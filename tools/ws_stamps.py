#!/usr/bin/env python3
"""Diagnostic: the wave-specialised pitch kernel's per-wavefront timeline (workgroup 0), per block type.

    VP_AMD_LIB=vocoderproject_amd/libvp_amd_stamps.so python tools/ws_stamps.py [--iir fast] [--steps 96]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VP_AMD_LIB", os.path.join(ROOT, "vocoderproject_amd", "libvp_amd_stamps.so"))

REC = {0: "start", 1: "wait producers (sum)", 2: "recursion (sum)", 3: "wait turn (sum)", 4: "windowed add (sum)", 5: "KERNEL", 6: "end", 7: "end / launches"}
PROD = {0: "start", 1: "wait background (sum)", 2: "adopt+residual+tables (sum)", 3: "group barriers (sum)", 4: "grain table (sum)", 5: "gather pass (sum)", 7: "end"}
BG0 = {0: "start", 1: "transform done", 2: "prefix sums done", 5: "-", 6: "-"}
BG1 = {0: "start", 1: "transform done", 2: "autocorr half done"}
BG2 = {0: "start", 1: "transform done", 2: "autocorr complete", 3: "levinson done", 4: "impulse response, lpc published"}
LEAD = {0: "start", 1: "cross-correlations done", 2: "prefix sums seen", 3: "picked", 4: "marks placed", 5: "published (1st start)", 6: "published (2nd start)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iir", default="fast")
    ap.add_argument("--yin", default="xcorr")
    ap.add_argument("--steps", type=int, default=96)
    a = ap.parse_args()
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    from vocoderproject_amd.synth import make_streams
    S, N, U = 256, 1024, 16
    dev = torch.device("cuda", 0)
    p = BatchVocoderProcessor(vocBool=0)
    p.prepareToPlay(44100.0, N, S)
    p.set_iir_mode(a.iir)
    p.set_yin_mode(a.yin)
    x = make_streams(S, N * U, device=dev).view(S, 3, U, N).permute(2, 0, 1, 3).contiguous()
    xm = x[:, :, 0, :].contiguous()
    y = torch.empty((S, 2, N), dtype=torch.float32, device=dev)
    for i in range(12):
        p.process_mono_device(xm[i % U], y)
    p.synchronize()
    v = (C.c_ulonglong * 512)()
    p.L.vp_debug_read_ws_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    p._chk(p.L.vp_debug_read_ws_stamps(p.h, v, 1))
    for i in range(a.steps):
        p.process_mono_device(xm[i % U], y)
    p.synchronize()
    p._chk(p.L.vp_debug_read_ws_stamps(p.h, v, 1))
    NW = int(os.environ.get("VP_WS_WAVES", "12"))
    print(f"kernel {p.pitch_kernel_name()}  iir={a.iir} yin={a.yin}: microseconds since kernel entry / summed per block, workgroup 0, by block type (nChunk at entry)")
    fast = a.iir == "fast"
    M = 1 << 64
    if v[0]:
        print(f"gather (workgroup 0, all launches): {int(v[0])} trips, {v[1] / max(int(v[0]), 1) / 100.0:.2f} us each; {int(v[2])} grain pairs, {int(v[3])} by the general form")
    for typ in range(4):
        n = int(v[typ * 128 + 0 * 8 + 7])
        if n == 0:
            continue
        t0 = int(v[typ * 128 + 15 * 8 + 7])                    # the launches' entry clocks, summed

        def ev(w, e):                                          # an EVENT slot: absolute clocks summed -> mean microseconds since entry
            raw = int(v[typ * 128 + w * 8 + e])
            return ((raw - t0) % M) / n / 100.0 if raw else 0.0

        def du(w, e):                                          # a DURATION slot
            return int(v[typ * 128 + w * 8 + e]) / n / 100.0

        print(f"--- type {typ}: {n} launches, kernel {ev(0, 5):.1f} us")
        for w in range(NW):
            if w == 1 and fast:                                # (the windowed-add wavefront has no timers: its slots carry wavefront 0's prologue)
                print(f"  prologue (wave 0)      requests issued: {ev(1, 0):.1f} | consumed: {ev(1, 1):.1f} | barrier passed: {ev(1, 2):.1f}")
                continue
            names = REC if w < 2 else PROD if w < NW - 4 else (BG0, BG1, BG2, LEAD)[w - (NW - 4)]
            role = "recursion" if names is REC else "producer" if names is PROD else "background" + (" (lead)" if w == NW - 1 else "")
            events = {0, 6, 7} if names is REC else {0, 7} if names is PROD else set(range(8))
            cells = []
            for e in range(8):
                if w == 0 and e in (5, 7):
                    continue
                if w == 1 and not fast and e in (0, 1, 2):     # (EXACT: the second recursion wavefront's slots 0-2 also take wavefront 0's prologue stamps)
                    continue
                val = ev(w, e) if e in events else du(w, e)
                if val or e == 0:
                    cells.append(f"{names.get(e, str(e))}: {val:.1f}")
            print(f"  wave {w} {role:18s} " + " | ".join(cells))
        fine = ["loop entry", "state rolled", "notes requested", "analysis marks", "coefficients adopted", "zeroing seen"]
        print("  lead, first start in detail   " + " | ".join(f"{nm}: {ev(12, k):.1f}" for k, nm in enumerate(fine)))
        finep = ["published seen", "adopted (0) / -", "grain table (0) / quotient table (1)", "barrier", "first chunk gathered", "barrier", "rest gathered"]
        for r in (0, 1):
            print(f"  producer {r}, first Start's segment   " + " | ".join(f"{nm}: {ev(13 + r, k):.1f}" for k, nm in enumerate(finep)))
        print("  recursion, instance taken up at   " + " | ".join(f"{ev(15, k):.1f}" for k in range(7) if v[typ * 128 + 15 * 8 + k]))


if __name__ == "__main__":
    main()
