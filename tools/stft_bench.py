"""Timing of the standalone STFT round trip (fused kernel) over batch shapes: frames/s, HBM GB/s on the algorithmic bytes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vocoderproject_amd import StftRoundTrip

def run(S, T, F=1024, hop=256, reps=30, runs=0, semitones=None, precision="f64"):
    st = StftRoundTrip(S, T, F, hop)
    st.set_runs(runs)
    st.set_precision(precision)
    x = torch.randn((S, T), dtype=torch.float32, device="cuda") * 0.1
    y = torch.empty_like(x)
    f = (lambda: st(x, y)) if semitones is None else (lambda: st.pitch_shift(x, y, semitones))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    frames = S * st.n_frames
    print(f"S={S:5d} T={T:7d} F={F} hop={hop} runs={runs} pv={semitones} {precision}: {dt*1e6:9.1f} us  {frames/dt/1e6:8.1f} M frames/s  "
          f"{2*S*T*4/dt/1e9:7.1f} GB/s (in+out once)  fused={st.fused}", flush=True)

if __name__ == "__main__":
    for S, T in ((256, 16384), (256, 65536), (1024, 65536), (4096, 32768), (64, 262144)):
        run(S, T)
    for runs in (1, 2, 4, 8, 16):
        run(256, 65536, runs=runs)
    run(256, 65536, semitones=7.0)
    run(1024, 65536, semitones=7.0)
    run(256, 65536, hop=512)
    run(256, 65536, hop=128)
    run(256, 65536, F=2048, hop=512)          # BASELINE configs[4]'s "2048-pt FFT hop 512"
    run(1024, 65536, F=2048, hop=512)
    # single precision (vp_stft_set_precision: vp_k_stft_fused32)
    for S, T in ((256, 16384), (256, 65536), (1024, 65536), (4096, 32768), (64, 262144)):
        run(S, T, precision="f32")
    for runs in (1, 4, 8, 16, 32):
        run(256, 65536, runs=runs, precision="f32")
    run(256, 65536, hop=512, precision="f32")
    run(256, 65536, hop=128, precision="f32")
    run(256, 65536, F=2048, hop=512, precision="f32")
    run(1024, 65536, F=2048, hop=512, precision="f32")
