"""Generate tests/golden/methods_vectors.npz by RUNNING the reference's own Python module
(/root/reference/Notebook/methods.py) in the build container.

Only inputs and outputs (data) are stored -- no reference source.  The reference tree does not
exist on the GPU box, so this script is run once here and its output committed.

    python tests/golden/gen_methods_vectors.py

What methods.py can pin (it is the authors' NumPy proof-of-concept of the same algorithms, with a
different parametrisation -- SURVEY.md section 2 row 7 and section 8c):
  * biased_auto_corr          <-> LPC.cpp:44-97  (rectangular window)
  * levinson_durbin / lpc     <-> LPC.cpp:107-148
  * yin_algo                  <-> PitchProcess.cpp:350-448 (tau_max = round vs ceil: equal at 44.1 kHz/100 Hz)
  * pitch_marks (first voiced frame after an unvoiced one: search right + left; runs of frames: continuation
    through the overlap's marks, unvoiced extrapolation, restart)
                              <-> PitchProcess.cpp:455-567
  * hann(2T+1) as pitch_shift uses it <-> PitchProcess.cpp:878-882 (JUCE symmetric Hann)
  * create_window('sine')     <-> VocoderProcess.cpp:95-135 (np.pi vs the literal 3.14159265)
  * build_notes_vector('chromatic') <-> Notes.cpp:43-70
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/Notebook")

import numpy as np  # noqa: E402

import methods as M  # noqa: E402  (the reference module)

HERE = os.path.dirname(os.path.abspath(__file__))
FS = 44100.0


def voiced(rng, n, f0, fs=FS, harmonics=8, noise=0.002):
    t = np.arange(n) / fs
    ph = rng.uniform(0, 2 * np.pi)
    x = sum(np.sin(2 * np.pi * h * f0 * t + h * ph) / h for h in range(1, harmonics + 1)) * 0.25
    x = x + rng.normal(0, noise, n)
    return x.astype(np.float32).astype(np.float64)  # the plugin's inputs are float32


def main():
    rng = np.random.default_rng(20201002)
    out = {}

    # ---- autocorrelation + Levinson-Durbin -------------------------------------------------------
    lpc_cases = []
    for ci, (L, p, f0) in enumerate([(512, 5, 130.0), (512, 40, 210.0), (1024, 15, 151.0), (1024, 24, 333.0),
                                     (2048, 48, 97.0), (1024, 100, 262.0), (556, 2, 440.0)]):
        x = voiced(rng, L, f0)
        r = M.biased_auto_corr(x, p)              # sequential sum over n for each lag, like the C++
        a = M.levinson_durbin(r, p)
        out[f"lpc{ci}_x"] = x
        out[f"lpc{ci}_r"] = r
        out[f"lpc{ci}_a"] = a
        lpc_cases.append((L, p))
    out["lpc_cases"] = np.array(lpc_cases)
    # all-zero frame: the |r0| < 1e-9 branch
    out["lpc_zero_a"] = M.levinson_durbin(np.zeros(16), 15)

    # ---- YIN ------------------------------------------------------------------------------------------
    yin_cases = []
    w_len = 1024
    tau_max = int(np.round(1 / 100.0 * FS))       # 441, same as ceil(fs/fMin) in the C++
    for ci, f0 in enumerate([110.0, 151.0, 233.3, 347.0, 520.0, 790.0, 60.0]):
        x = voiced(rng, w_len + tau_max + 8, f0)
        yt = np.zeros(tau_max)
        pitch = M.yin_algo(x, tau_max, yt, w_len, FS, 100.0, 800.0, 0.25)
        # yin_algo re-binds yin_temp internally; recompute its normalised difference for the record
        xf = x[0: tau_max + w_len]
        d = np.array([np.sum((xf[0:w_len] - xf[tau: w_len + tau]) ** 2) for tau in range(tau_max)])
        d[0] = 1
        tmp = 0.0
        for tau in range(1, tau_max):
            tmp += d[tau]
            d[tau] = d[tau] * tau / tmp
        out[f"yin{ci}_x"] = x
        out[f"yin{ci}_d"] = d
        out[f"yin{ci}_pitch"] = np.float64(pitch)
        yin_cases.append(f0)
    out["yin_cases"] = np.array(yin_cases)
    # noise: unvoiced -> methods returns the sentinel 5.0
    xn = rng.normal(0, 0.1, w_len + tau_max + 8).astype(np.float32).astype(np.float64)
    out["yin_noise_x"] = xn
    out["yin_noise_pitch"] = np.float64(M.yin_algo(xn, tau_max, np.zeros(tau_max), w_len, FS, 100.0, 800.0, 0.25))

    # ---- analysis pitch marks, first voiced frame (prev unvoiced): search right and left ----------------------
    pm_cases = []
    for ci, f0 in enumerate([120.0, 151.0, 260.0, 410.0]):
        x = voiced(rng, w_len, f0, noise=0.0005)
        T = int(FS / f0)
        pitch = FS / T
        marks = M.pitch_marks(x, pitch, np.array([], dtype=int), 0.0, 100.0, w_len, 768, FS, 0.94, valley=True)
        out[f"pm{ci}_x"] = x
        out[f"pm{ci}_marks"] = np.asarray(marks, dtype=np.int64)
        out[f"pm{ci}_period"] = np.int64(int(FS / pitch))     # the T methods.pitch_marks itself derives
        pm_cases.append(f0)
    out["pm_cases"] = np.array(pm_cases)

    # ---- analysis pitch marks over RUNS of frames: voiced->voiced continuation through the marks of the overlap,
    #      voiced->unvoiced periodic extrapolation (with and without marks in the overlap), unvoiced->voiced restart
    #      (search right + left).  The tracker state rolls as in the notebook's driver loop: prev_marks = last marks,
    #      prev_pitch = last frame's pitch, prev_voiced_pitch = last voiced pitch.  (The voiced->voiced branch WITHOUT
    #      marks in the overlap is left out: methods.pitch_marks returns the arg-extremum relative to the search
    #      slice there, the plugin an absolute index.)
    H = 768
    seqs = [(130.0, "VVVUUVVUV"), (233.0, "VVUUUVVVU"), (300.0, "VUVVUUVVV"), (180.0, "VVUUVVV"), (420.0, "VVVUVVUUV")]
    for si, (f0, pat) in enumerate(seqs):
        nf = len(pat)
        x = voiced(rng, w_len + (nf - 1) * H, f0, noise=0.0005)
        T = int(FS / f0)
        pitch = FS / T
        assert int(FS / pitch) == T                       # methods derives T back from the pitch
        prev_marks = np.array([], dtype=int)
        prev_st = np.array([], dtype=int)
        prev_pitch, prev_voiced = 0.0, 100.0
        periods, all_marks, all_st, betas = [], [], [], []
        # synthesis marks (methods.synthesis_pitch_marks <-> placeStMarks, PitchProcess.cpp:573-658) with beta = nearest
        # chromatic note / pitch, the notes being the reference module's own table.  methods takes T_new =
        # int(fs / (beta pitch)), the plugin round(period / beta): the cases keep frac(fs / note) < 0.5 where the two
        # agree; on unvoiced frames the plugin ignores beta (periodNew = prevVoicedPeriod), so methods gets beta = 1.
        notes_tab, _ = M.build_notes_vector("chromatic", n_oct=4)
        closest = float(notes_tab[np.argmin(np.abs(notes_tab - pitch))])
        assert (FS / closest) % 1.0 < 0.5, (f0, FS / closest)
        for f, c in enumerate(pat):
            cur = pitch if c == "V" else 0.0
            xf = x[f * H: f * H + w_len]
            if cur > 10 and prev_pitch > 10:
                assert np.sum(prev_marks - H >= 0) > 0, "voiced->voiced without overlap marks: not comparable"
            marks = np.asarray(M.pitch_marks(xf, cur, prev_marks, prev_pitch, prev_voiced, w_len, H, FS, 0.94, valley=True),
                               dtype=np.int64)
            all_marks.append(marks)
            periods.append(T if c == "V" else 0)
            beta = closest / pitch if cur > 10 else 1.0
            if marks.size == 0:
                st = np.array([], dtype=np.int64)             # placeStMarks returns with no marks (:580-581); methods would index an_marks[0]
            else:
                st = np.asarray(M.synthesis_pitch_marks(cur, prev_pitch, prev_voiced, marks, prev_st, beta, w_len, H, FS),
                                dtype=np.int64)
                if cur > 10 and not prev_pitch > 10:
                    # restart: methods also lays marks to the LEFT of an_marks[0] (arange(-3, ...)), the plugin starts at it
                    assert marks[0] - int(FS / (beta * cur)) < 0, "restart case not comparable"
            all_st.append(st)
            betas.append(beta)
            prev_st = st
            prev_marks = marks
            prev_pitch = cur
            if cur > 10:
                prev_voiced = cur
        out[f"pmseq{si}_x"] = x
        out[f"pmseq{si}_periods"] = np.array(periods, dtype=np.int64)
        out[f"pmseq{si}_counts"] = np.array([len(m) for m in all_marks], dtype=np.int64)
        out[f"pmseq{si}_marks"] = np.concatenate(all_marks) if all_marks else np.array([], dtype=np.int64)
        out[f"pmseq{si}_st_counts"] = np.array([len(m) for m in all_st], dtype=np.int64)
        out[f"pmseq{si}_st_marks"] = np.concatenate(all_st)
        out[f"pmseq{si}_beta"] = np.array(betas)
    out["pmseq_n"] = np.int64(len(seqs))

    # ---- the PSOLA grain window: methods.pitch_shift takes hann(2T+1) (methods.py:401, scipy's symmetric Hann, the
    #      function object the reference module itself imported) <-> fillPsolaWindow's JUCE symmetric Hann
    #      (PitchProcess.cpp:878-882)
    for T in (55, 292, 441):
        out[f"hann_{2 * T + 1}"] = np.asarray(M.hann(2 * T + 1), dtype=np.float64)

    # ---- windows and note table ----------------------------------------------------------------------------------
    for W, ov in [(512, 0.75), (1024, 0.75), (2048, 0.75), (556, 0.75), (512, 0.5)]:
        out[f"sine_{W}_{int(ov * 100)}"] = M.create_window(W, overlap=ov, type="sine")
    notes, _ = M.build_notes_vector("chromatic", n_oct=4)
    out["notes_chromatic"] = notes

    path = os.path.join(HERE, "methods_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
