"""Generate tests/golden/filter_psola_vectors.npz by RUNNING the reference's own Python module
(/root/reference/Notebook/methods.py) and the scipy.signal.lfilter calls the reference's notebook makes
(Notebook/"Pitch Corrector and Vocoder.ipynb", cells 9 and 25), in the build container.

Only inputs and outputs (data) are stored -- no reference source.  The reference tree does not exist on the GPU box, so this
script is run once here and its output committed.

    python tests/golden/gen_filter_psola_vectors.py

What it pins (round-1 verdict, "extend the oracle pin by the same committed-script route"):
  * methods.pitch_shift (methods.py:374-476): grain extraction, Hann(2T+1), x-positions mark + t/beta, linear
    interpolation onto the integer grid, accumulation in mark order
                              <-> PitchProcess::psola / interp / getClosestAnMarkIdx (PitchProcess.cpp:665-741, 842-870, 788-831)
    The notebook writes each grain onto [mark - T_new, mark + T_new] with T_new = int(fs / (beta pitch)) and zero outside
    the grain's x-range; the plugin onto [floor(x0), ceil(xN)) inside the x-range.  Because T/beta - T_new lies in
    (-1/beta, 1), the two supports hold the same integers, so whole frames are comparable -- for frames whose first and
    last marks' grains lie inside the frame and the residual buffer (their half-windowing is the same in both: the
    unwindowed half is the outer one, hann(2T+1)[T] = 1).
  * sp.lfilter(a, [1], x[i - tau_max : i + w_len]) and sp.lfilter([1], a, out_window)  (cell 9)
                              <-> PitchProcess::filterFIR / filterIIR (PitchProcess.cpp:280-322)
  * sp.lfilter(a, [1], x_frame) and sp.lfilter([1], a, g * y_frame) with x_frame = x * window  (cell 25)
                              <-> VocoderProcess::filterFIR / filterIIR (VocoderProcess.cpp:235-286)
    with a = methods.lpc(frame, p) (methods.py:141-150).  lfilter is transposed direct form II: same filter, other
    summation order, so the comparison carries a tolerance (1e-9 relative to the signal's peak).
  * methods.build_notes_vector(key) for the 12 major keys (methods.py:22-77) <-> Notes::buildFreqVect (Notes.cpp:43-70)
    and the nearest note (np.argmin(np.abs(pitch - notes)), cell 9) <-> Notes::getClosestFreq (Notes.cpp:79-110)
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/Notebook")

import numpy as np  # noqa: E402
import scipy.signal as sp  # noqa: E402

import methods as M  # noqa: E402  (the reference module)

HERE = os.path.dirname(os.path.abspath(__file__))
FS = 44100.0
KEYS = ("A", "A#", "B", "C", "C#", "D", "D#", "E", "F", "F#", "G", "G#")      # PluginProcessor.cpp:64 / Notes.h:18 order


def voiced(rng, n, f0, fs=FS, harmonics=8, noise=0.002):
    t = np.arange(n) / fs
    ph = rng.uniform(0, 2 * np.pi)
    x = sum(np.sin(2 * np.pi * h * f0 * t + h * ph) / h for h in range(1, harmonics + 1)) * 0.25
    x = x + rng.normal(0, noise, n)
    return x.astype(np.float32).astype(np.float64)  # the plugin's inputs are float32


def main():
    rng = np.random.default_rng(20261002)
    out = {}

    # ---- PSOLA ----------------------------------------------------------------------------------------------------------
    w_len, tau_max = 1024, 441
    cases = []
    for ci, (T, beta) in enumerate([(100, 1.02), (100, 0.97), (157, 1.0293), (292, 0.985), (60, 1.0), (211, 1.0595), (333, 0.944),
                                    (80, 2.0), (120, 0.5), (147, 1.3348)]):
        pitch = FS / (T + 0.5)                       # int(f_s / pitch) = T inside pitch_shift
        assert int(FS / pitch) == T
        T_new = int(FS / (beta * pitch))
        e = rng.normal(0, 0.05, tau_max + w_len)
        e += 0.3 * np.sin(2 * np.pi * np.arange(e.size) / T)
        # analysis marks about T apart with jitter; first and last far enough from the frame's ends for whole grains
        an = [T + int(rng.integers(0, 8))]
        while an[-1] + T + int(2) < w_len - T - 2:
            an.append(an[-1] + T + int(rng.integers(-3, 4)))
        an = np.array([m for m in an if T <= m and m + T + 1 <= w_len], dtype=np.int64)
        # synthesis marks T_new apart; keep those whose written range [mark - T_new, mark + T_new] lies inside the frame
        first = int(an[0]) + int(rng.integers(-5, 6))
        st = first + np.arange(0, 64) * T_new
        st = st[(st - max(T_new, int(np.ceil(T / beta))) - 1 >= 0) & (st + max(T_new, int(np.ceil(T / beta))) + 2 <= w_len)]
        # no ties between a mark's two neighbouring analysis marks (argmin takes the lower one, the plugin the upper)
        st = np.array([m for m in st if np.sum(np.abs(m - an) == np.min(np.abs(m - an))) == 1], dtype=np.int64)
        if st.size < 3:
            continue
        ow = M.pitch_shift(e.copy(), np.zeros(w_len), pitch, pitch, an, st, beta, w_len, FS, tau_max)
        out[f"psola{ci}_e"] = e
        out[f"psola{ci}_an"] = an
        out[f"psola{ci}_st"] = st
        out[f"psola{ci}_out"] = ow
        cases.append((ci, T, T_new))
        out[f"psola{ci}_beta"] = np.array([beta])
    out["psola_cases"] = np.array(cases)

    # ---- pitch path filters (cell 9) ---------------------------------------------------------------------------------------
    pf = []
    for ci, (p, f0) in enumerate([(15, 151.0), (24, 233.0), (48, 110.0), (2, 300.0), (100, 190.0)]):
        x = voiced(rng, 2 * w_len, f0)                                  # idx -toKeep .. F-1 with toKeep = F = 1024
        a = M.lpc(x[w_len:], p)                                          # lpc(x_frame, p)
        e = sp.lfilter(a, [1], x)                                        # zero state at the left end of the kept samples
        ow = rng.normal(0, 0.1, w_len)
        y = sp.lfilter([1], a, ow)
        out[f"pf{ci}_x"], out[f"pf{ci}_a"], out[f"pf{ci}_e"], out[f"pf{ci}_ow"], out[f"pf{ci}_y"] = x, a, e, ow, y
        pf.append((ci, p))
    out["pf_cases"] = np.array(pf)

    # ---- vocoder window (cell 25) --------------------------------------------------------------------------------------------
    vw = []
    for ci, (W, hop, pv, ps, f0) in enumerate([(512, 128, 40, 5, 140.0), (1024, 256, 24, 5, 205.0), (2048, 512, 48, 30, 99.0),
                                               (512, 256, 16, 8, 330.0)]):
        window = M.create_window(W, overlap=(W - hop) / W, type='sine')
        if isinstance(window, tuple):
            window = window[0]
        x = voiced(rng, W, f0)
        t = np.arange(W) / FS
        y = (0.15 * (2 * ((110.0 * t) % 1.0) - 1) + 0.15 * (2 * ((164.81 * t + 0.3) % 1.0) - 1)).astype(np.float32).astype(np.float64)
        x_frame, y_frame = x * window, y * window
        a = M.lpc(x_frame, pv)
        a_s = M.lpc(y_frame, ps)
        e = sp.lfilter(a, [1], x_frame)
        e_s = sp.lfilter(a_s, [1], y_frame)
        for k, v in dict(x=x, y=y, win=window, a=a, as_=a_s, e=e, es=e_s).items():
            out[f"vw{ci}_{k}"] = v
        vw.append((ci, W, hop, pv, ps))
    out["vw_cases"] = np.array(vw)

    # ---- note tables of the 12 major keys, and nearest notes on a pitch grid ---------------------------------------------------
    grid = np.concatenate([np.linspace(101.0, 790.0, 400), rng.uniform(100.5, 795.0, 200)])
    out["notes_grid"] = grid
    for k, name in enumerate(KEYS):
        notes, _ = M.build_notes_vector(name, n_oct=4)
        out[f"notes_key{k}"] = notes
        out[f"notes_key{k}_closest"] = np.array([notes[np.argmin(np.abs(p - notes))] for p in grid])

    path = os.path.join(HERE, "filter_psola_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays; psola cases", cases)


if __name__ == "__main__":
    main()
