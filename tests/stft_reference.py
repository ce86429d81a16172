"""Build-authored NumPy restatement of the standalone STFT kernels (csrc/vp_stft.hip): the checker of a path that has NO reference
counterpart (the reference contains no FFT, no STFT and no phase vocoder: SURVEY.md section 0) -- PARITY UNPINNED by nature.
Test infrastructure only.

stft_roundtrip: periodic sqrt-Hann analysis window, rfft, [stage], irfft, sqrt-Hann synthesis window, overlap-add normalised by the
sum of w^2 over one hop grid (frames f = 0 .. (T - F) / hop; samples the frames do not cover stay 0).

pv_stage: the classic phase-vocoder pitch shift between the two transforms, stated the way the kernel computes it:
  per frame and bin k (0..F/2): magnitude m, phase p;
  d = p - p_prev[k] - k 2 pi hop / F, wrapped: d -= 2 pi rint(d / 2 pi)          (the unwrap)
  true frequency in bins: fk = k + d (F / hop) / (2 pi)
  synthesis bin kk gathers the bins k with floor(k ratio + 0.5) == kk (increasing k): magnitudes add, the frequency of the last one
  scaled by the ratio stays; phase increment inc = (2 pi hop / F) fk' ; the accumulator runs over the frames -- in rounds of four
  frames: sp_w = carry + inc_0 + ... + inc_w (left to right), carry' = sp_last - 2 pi rint(sp_last / 2 pi);
  output bin = m' (cos sp + i sin sp); bins 0 and F/2 keep their real parts only (a real frame).
"""
import numpy as np

TWO_PI = 6.283185307179586476925286766559
ROUND = 4            # frames per round of the kernel (VP_STFT_WAVES)


def window(F):
    return np.sqrt(0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(F) / F))


def stft_roundtrip(x, F=1024, hop=256, ratio=None):
    """x: float array [T] -> float64 [T]."""
    x = np.asarray(x, np.float64)
    T = len(x)
    w = window(F)
    scale = 1.0 / np.sum(w[::hop] ** 2)
    nF = (T - F) // hop + 1
    nb = F // 2 + 1
    y = np.zeros(T)
    p_prev = np.zeros(nb)
    carry = np.zeros(nb)
    sp = np.zeros(nb)
    k = np.arange(nb)
    O = F // hop
    expct = TWO_PI / O
    for f in range(nF):
        X = np.fft.rfft(x[f * hop:f * hop + F] * w)
        if ratio is not None:
            m, p = np.abs(X), np.arctan2(X.imag, X.real)
            d = p - p_prev - k * expct
            d -= TWO_PI * np.rint(d * (1.0 / TWO_PI))
            fk = k + d * (O * (1.0 / TWO_PI))
            p_prev = p
            idx = np.floor(k * ratio + 0.5).astype(np.int64)
            sm, sf = np.zeros(nb), np.zeros(nb)
            for kk in range(nb):                     # increasing k: magnitudes add, the last frequency stays
                t = idx[kk]
                if 0 <= t < nb:
                    sm[t] += m[kk]
                    sf[t] = fk[kk] * ratio
            inc = expct * sf
            if f % ROUND == 0:
                sp = carry + inc
            else:
                sp = sp + inc
            if f % ROUND == ROUND - 1 or f == nF - 1:
                carry = sp - TWO_PI * np.rint(sp * (1.0 / TWO_PI))
            X = sm * (np.cos(sp) + 1j * np.sin(sp))
            X[0] = X[0].real
            X[-1] = X[-1].real
        y[f * hop:f * hop + F] += np.fft.irfft(X, F) * w
    return y * scale
