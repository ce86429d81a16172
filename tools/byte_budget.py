#!/usr/bin/env python3
"""Byte budget of the lane-per-window vocoder pipeline (csrc/vp_voc2.hip) and of the pitch kernel beside it: per kernel of one
configs[3] / configs[4] step, the bytes it HAS to move (every input of the stage read once, every output written once, in the
types the stage stores them in) against the HBM bytes the counters saw (profiles/<tag>_counters.json: 2 x FETCH_SIZE + WRITE_SIZE per
launch, the gfx950 correction of MI355X_MICROARCH.md).  Prints a markdown table (profiles/<tag>_byte_budget.md).

    python tools/byte_budget.py r04 > profiles/r04_byte_budget.md
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stage_bytes(S, N, W, hop, oV, oS, F, C):
    """{kernel prefix: (bytes, what)} for one block of S streams: nWin = N / hop windows per stream."""
    nw = S * (N // hop)                       # windows per launch
    MB = 1.0e6
    smp_in = S * 3 * N * 4                    # the block's input, f32, three channels
    return {
        "vp_k_v2_ingest_stage": ((smp_in + smp_in + 2 * S * N * 4) / MB,
                                 "input in once, the three rings' new samples out, voice + carrier-L samples of the block staged once (f32)"),
        "vp_k_v2_autocorr": ((2 * S * (N + W) * 4 + nw * (oV + 1 + oS + 1) * 8) / MB,
                             "every staged sample of the block's windows once (f32), r[] out (f64)"),
        "vp_k_v2_levinson2": ((2 * nw * (oV + 1 + oS + 1) * 8) / MB, "r[] in, a[] out (f64)"),
        "vp_k_v2_fir2": ((2 * S * (N + W) * 4 + nw * (oV + 1 + oS + 1) * 8 + nw * W * 4 + nw * 2 * 8) / MB,
                         "staged samples once, a[] in, the carrier's residual of every window out (f32 [W] per window in FAST mode: windows overlap W/hop times), energies"),
        "vp_k_v2_energy": ((nw * 4 * 8) / MB, "partial energies in, window energies out"),
        "vp_k_v2_iir_fast": ((2 * nw * W * 4 + nw * (oV + 1) * 8) / MB, "residual in, a[] in, the window's all-pole output out (both f32 [W] per window, round 4)"),
        "vp_k_v2_iir_exact": ((2 * nw * W * 8 + nw * (oV + 1) * 8) / MB, "residual in, a[] in, the window's all-pole output out (f64 [W] per window)"),
        "vp_k_v2_ola": ((nw * W * 4 + 2 * S * N * 8 + S * 2 * N * 4) / MB,
                        "every window's output in once (f32 in FAST mode), the accumulator's block slice read and written (f64), the block's output out (f32) when it emits"),
        "vp_k_pitch": ((S * N * 4 * 2 + S * 2 * N * 4 + S * 2 * 2048) / MB,
                       "voice in (slab + ring write), output out (f32), tracker state in and out -- the frame in flight would not have to leave the chip"),
    }


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_counters.json")))
    cfgs = [("configs[3] per GPU: 1024 streams, 44.1 kHz, N = 1024, vocoder 512/128, orders 40/5, pitch 1024/768", "cfg/both/S1024/N1024/fast/xcorr",
             dict(S=1024, N=1024, W=512, hop=128, oV=40, oS=5, F=1024, C=256)),
            ("configs[4] per GPU: 512 streams, 48 kHz, N = 2048, vocoder 2048/512, orders 48/30, pitch 2048/1536", "cfg5/both/S512/N2048/fast/xcorr",
             dict(S=512, N=2048, W=2048, hop=512, oV=48, oS=30, F=2048, C=512))]
    print(f"# Byte budget of one step, per kernel (`profiles/{tag}_counters.json`; tools/byte_budget.py)\n")
    print("\"has to move\" = every input of the stage read once and every output written once in the types the stage stores; "
          "\"counters\" = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch.  The external f32 I/O of the whole step (SURVEY 8d's algorithmic bytes) is the last line.\n")
    for title, key, geo in cfgs:
        need = stage_bytes(**geo)
        print(f"## {title}\n")
        print("| kernel | us (rocprofv3) | has to move (MB) | counters (MB) | ratio | GB/s on the counters | what it has to move |")
        print("|---|---|---|---|---|---|---|")
        tot_need = tot_ctr = tot_us = 0.0
        for k, e in sorted(d["kernels"].items()):
            if not k.endswith("@" + key) or "hbm_bytes_per_launch" not in e:
                continue
            name = k.split("@")[0]
            pref = next((p for p in need if name.startswith(p)), None)
            if pref is None:
                continue
            nb, what = need[pref]
            mb = e["hbm_bytes_per_launch"] / 1e6
            us = e.get("rocprof_avg_us", 0.0)
            tot_need += nb; tot_ctr += mb; tot_us += us
            print(f"| `{name}` | {us:.1f} | {nb:.1f} | {mb:.1f} | {mb / nb:.1f}x | {mb / us * 1e3:.0f} | {what} |")
        alg = geo["S"] * geo["N"] * 4 * 5 / 1e6
        print(f"| **sum** | {tot_us:.1f} | {tot_need:.1f} | {tot_ctr:.1f} | {tot_ctr / tot_need:.1f}x | {tot_ctr / tot_us * 1e3:.0f} | external f32 I/O of the step: {alg:.1f} MB |\n")


if __name__ == "__main__":
    main()
