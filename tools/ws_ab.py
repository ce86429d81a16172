#!/usr/bin/env python3
"""A/B of the wave-specialised pitch kernel (vp_k_pitch_ws*) against the phase kernels (VP_NO_WS=1) on one box:
bit-equality of the output and of the tracker states on a corpus with gate crossings and unvoiced stretches, timing of both,
and the bounded-wait timeout counter (must stay 0).

    python tools/ws_ab.py [--streams 256] [--blocks 48]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import time
    import numpy as np
    import torch
    from vocoderproject_amd import BatchVocoderProcessor
    from vocoderproject_amd.synth import make_streams
    S, N, B = a.streams, a.block, a.blocks
    dev = torch.device("cuda", 0)
    x = make_streams(S, N * B).clone()
    # gate crossings and unvoiced stretches on some streams
    g = torch.Generator().manual_seed(7)
    for s in range(0, S, 5):
        t0 = int(torch.randint(0, N * B // 2, (1,), generator=g))
        x[s, :, t0:t0 + 6 * N] *= 1e-6
    for s in range(2, S, 7):
        t0 = int(torch.randint(0, N * B // 2, (1,), generator=g))
        x[s, 0, t0:t0 + 4 * N] = 0.05 * torch.randn(4 * N, generator=g)
    res = {}
    for iir in ("fast", "exact"):
        for yin in ("xcorr", "direct"):
            p = BatchVocoderProcessor(vocBool=0)
            p.prepareToPlay(44100.0, N, S)
            p.set_iir_mode(iir)
            p.set_yin_mode(yin)
            name = p.pitch_kernel_name()
            y = p.run(x.numpy())
            states = []
            for s in range(min(S, 64)):
                d = p.pitch_state(s)
                d["a"] = d["a"].tobytes()
                states.append(repr(sorted(d.items())))
            stamps = [round(v * 100.0) for v in p.debug_stamps(reset=False)]
            np.save(os.path.join(a.out, f"y_{iir}_{yin}.npy"), y)
            res[f"{iir}_{yin}"] = dict(kernel=name, timeouts=[int(stamps[i]) for i in (59, 60, 61)], cert=[int(stamps[62]), int(stamps[63])],
                                        states=[hashlib.sha256(s.encode()).hexdigest()[:16] for s in states], rms=float(np.sqrt((y.astype(np.float64) ** 2).mean())))
            if yin == "xcorr":
                U = 16
                xd = make_streams(S, N * U, device=dev).view(S, 3, U, N).permute(2, 0, 1, 3).contiguous()
                xm = xd[:, :, 0, :].contiguous()
                yo = torch.empty((S, 2, N), dtype=torch.float32, device=dev)
                for i in range(32):
                    p.process_mono_device(xm[i % U], yo)
                torch.cuda.synchronize()
                K = 400
                t0 = time.perf_counter()
                for i in range(K):
                    p.process_mono_device(xm[i % U], yo)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                res[f"{iir}_{yin}"]["us_per_block"] = dt / K * 1e6
                res[f"{iir}_{yin}"]["Mframes_s"] = S * N / 256 * K / dt / 1e6
                res[f"{iir}_{yin}"]["timeouts_after_timing"] = [round(p.debug_stamps(reset=False)[i] * 100.0) for i in (59, 60, 61)]
            p.close()
    with open(os.path.join(a.out, "res.json"), "w") as f:
        json.dump(res, f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--block", type=int, default=1024)
    ap.add_argument("--blocks", type=int, default=48)
    ap.add_argument("--out", default="")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    import numpy as np
    out = {}
    with tempfile.TemporaryDirectory(prefix="ws_ab_") as tmp:
        for tag, env in (("ws", {}), ("phase", {"VP_NO_WS": "1"})):
            d = os.path.join(tmp, tag)
            os.makedirs(d)
            e = dict(os.environ, **env)
            subprocess.check_call([sys.executable, __file__, "--child", "--out", d, "--streams", str(a.streams), "--block", str(a.block), "--blocks", str(a.blocks)], env=e)
            with open(os.path.join(d, "res.json")) as f:
                out[tag] = json.load(f)
        ok = True
        for key in out["ws"]:
            ya, yb = np.load(os.path.join(tmp, "ws", f"y_{key}.npy")), np.load(os.path.join(tmp, "phase", f"y_{key}.npy"))
            same = bool(np.array_equal(ya, yb))
            st_same = out["ws"][key]["states"] == out["phase"][key]["states"]
            w, ph = out["ws"][key], out["phase"][key]
            diff = float(np.abs(ya.astype(np.float64) - yb).max())
            print(f"{key:14s} {w['kernel']:18s} vs {ph['kernel']:18s} output equal: {same} (max diff {diff:.3g}, rms {w['rms']:.4f})  states equal: {st_same}  "
                  f"timeouts {w['timeouts']} cert {w['cert']} / {ph['cert']}")
            if "us_per_block" in w:
                print(f"{'':14s} {w['us_per_block']:.1f} us/block = {w['Mframes_s']:.2f} M frames/s   against {ph['us_per_block']:.1f} us = {ph['Mframes_s']:.2f} M"
                      f"   timeouts after timing {w['timeouts_after_timing']}")
            ok = ok and same and st_same and not any(w["timeouts"])
        print("OK" if ok else "MISMATCH")
        return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
