#!/usr/bin/env python3
"""Diagnostic: where does a kernel spend its time?  Builds/loads the -DVP_STAMPS library and prints,
per phase, the microseconds workgroup 0 spent (100 MHz wall clock) averaged per step.

    VP_AMD_LIB=vocoderproject_amd/libvp_amd_stamps.so python tools/phase_stamps.py [--mode both] [--steps 50]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VP_AMD_LIB", os.path.join(ROOT, "vocoderproject_amd", "libvp_amd_stamps.so"))

PHASES = {15: "pitch: prologue (ingest, gate, state/frame/window loads)", 0: "pitch: loop top", 1: "pitch: YIN diff + autocorr", 2: "pitch: cum/normalise/pick", 3: "pitch: an marks",
          4: "pitch: st marks", 5: "pitch: levinson", 6: "pitch: FIR start", 7: "pitch: psola", 8: "pitch: IIR",
          9: "pitch: fill output", 12: "pitch:   (cum sum)", 13: "pitch:   (normalise / block IIR: wait for the zero-state responses)", 14: "pitch:   (psola qtab+grain table)", 10: "pitch: FIR cont", 11: "pitch: state out",
          24: "pitch:   (block IIR: carry-in / history matrix, all chunks)", 25: "pitch:   (block IIR: 64-term dot / walk over the blocks, all chunks)",
          28: "pitch:   (an marks: roll + first mark)", 29: "pitch:   (an marks: walk to the right)",
          26: "pitch:   (last wave: LPC autocorrelation)", 27: "pitch:   (last wave: Levinson-Durbin)",
          30: "pitch:   (last wave: impulse response)",
          32: "pitch:   (marks phase, wave 0 busy)", 33: "pitch:   (marks phase, wave 1 busy)", 34: "pitch:   (marks phase, wave 2 busy)",
          35: "pitch:   (marks phase, wave 3 busy)", 36: "pitch:   (marks phase, wave 4 busy)", 37: "pitch:   (marks phase, wave 5 busy)",
          38: "pitch:   (marks phase, wave 6 busy)", 39: "pitch:   (marks phase, wave 7 busy)",
          40: "pitch:   (YIN phase, wave 0 busy)", 41: "pitch:   (YIN phase, wave 1 busy)", 42: "pitch:   (YIN phase, wave 2 busy)",
          43: "pitch:   (YIN phase, wave 3 busy)", 44: "pitch:   (YIN phase, wave 4 busy)", 45: "pitch:   (YIN phase, wave 5 busy)",
          46: "pitch:   (YIN phase, wave 6 busy)", 47: "pitch:   (YIN phase, wave 7 busy)",
          48: "pitch:   (scan phase, wave 0 busy)", 49: "pitch:   (scan phase, wave 1 busy)", 50: "pitch:   (scan phase, wave 2 busy)",
          54: "pitch:   (scan phase, wave 6 busy)", 55: "pitch:   (scan phase, wave 7 busy)",
          16: "voc: load", 17: "voc: autocorr", 18: "voc: levinson", 19: "voc: FIR", 20: "voc: energies",
          21: "voc: gains", 22: "voc: IIR", 23: "voc: scale+OLA"}


PITCH_ONLY = {16 + w: f"pitch:   (IIR phase, wave {w} busy)" for w in range(8)}     # (the vocoder's ids, free in a pitch-only run)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="both")
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--block", type=int, default=1024)
    ap.add_argument("--iir", default="exact")
    ap.add_argument("--yin", default="direct")
    ap.add_argument("--fs", type=float, default=44100.0, help="sample rate handed to prepareToPlay (48000: the plugin's other geometry, 1112/834 + 556/139)")
    ap.add_argument("--cfg5", action="store_true", help="BASELINE configs[4] geometry: 48 kHz, 2048/1536 + 2048/512, orders 48/48/30, N = 2048")
    ap.add_argument("--lpc-voice", type=int, default=None, help="lpcVoice (BASELINE configs[2]: 24)")
    a = ap.parse_args()
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    from vocoderproject_amd.synth import make_streams
    S, N = a.streams, a.block
    if a.cfg5:
        N = 2048
        p = BatchVocoderProcessor(pitchBool=int(a.mode != "voc"), vocBool=int(a.mode != "pitch"), lpcVoice=48, lpcPitch=48, lpcSynth=30)
        p.prepareExplicit(48000.0, N, S, 2048, 1536, 2048, 512)
    else:
        kw = dict(lpcVoice=a.lpc_voice) if a.lpc_voice else {}
        p = BatchVocoderProcessor(pitchBool=int(a.mode != "voc"), vocBool=int(a.mode != "pitch"), **kw)
        p.prepareToPlay(a.fs, N, S)
    p.set_iir_mode(a.iir)
    p.set_yin_mode(a.yin)
    U = 16
    x = make_streams(S, N * U, fs=48000.0 if a.cfg5 else a.fs, device="cuda").view(S, 3, U, N).permute(2, 0, 1, 3).contiguous()
    y = torch.empty((S, 2, N), dtype=torch.float32, device="cuda")
    for i in range(8):
        p.process_device(x[i % U], y)
    p.debug_stamps(reset=True)
    for i in range(a.steps):
        p.process_device(x[(8 + i) % U], y)
    st = p.debug_stamps()
    # ids < 24 are consecutive stretches of thread 0's timeline (they add up to the kernels' duration); ids >= 24 are
    # timers of OTHER wavefronts or nested stretches, concurrent with the former
    names = dict(PHASES)
    if a.mode == "pitch":
        names.update(PITCH_ONLY)
    timeline = lambda i: i < 16 or (i < 24 and a.mode != "pitch")
    tot = sum(t for i, t in enumerate(st) if timeline(i))
    print(f"mode={a.mode} S={S} N={N} iir={a.iir}: per-step microseconds of workgroup 0, thread 0's timeline (sum {tot / a.steps:.1f})")
    for i, t in enumerate(st):
        if t and timeline(i):
            print(f"  {str(names.get(i, i)):44s} {t / a.steps:9.1f} us  {100 * t / tot:5.1f} %")
    if any(st[16:]):
        print("concurrent / nested timers:")
        for i, t in enumerate(st):
            if t and not timeline(i) and i < 62:         # 62/63 are frame counters of VP_YIN_XCORR, not timers
                print(f"  {str(names.get(i, i)):44s} {t / a.steps:9.1f} us")

if __name__ == "__main__":
    main()
