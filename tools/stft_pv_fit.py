#!/usr/bin/env python3
"""The polynomials of the phase-vocoder stage's elementary functions in turns (csrc/vp_stft.hip: PV_AT, PV_SN, PV_CS): Chebyshev-node
fits, converted to the power basis of the reduced argument's square, with their maximum errors on a dense grid.

    python tools/stft_pv_fit.py
"""
import numpy as np
from numpy.polynomial import chebyshev as C


def fit(f, lo, hi, deg, n=4000):
    k = np.arange(n)
    x = np.cos(np.pi * (k + 0.5) / n)
    u = (lo + hi) / 2 + (hi - lo) / 2 * x
    c = C.chebfit(x, np.asarray(f(u.astype(np.longdouble)), float), deg)
    p = C.cheb2poly(c)
    a, b = 2 / (hi - lo), -(lo + hi) / (hi - lo)
    base, acc, out = np.array([b, a]), np.array([1.0]), np.zeros(deg + 1)
    for ci in p:                                   # substitute x = a u + b
        out[:len(acc)] += ci * acc
        acc = np.convolve(acc, base)
    return out


def max_err(coef, f, lo, hi):
    u = np.linspace(lo, hi, 200001).astype(np.longdouble)
    pv = np.zeros_like(u)
    for c in coef[::-1]:
        pv = pv * u + c
    return float(np.max(np.abs(pv - f(u))))


def main():
    t8 = np.tan(np.pi / 8)
    tiny = np.longdouble(1e-300)
    fa = lambda u: np.where(u > 0, np.arctan(np.sqrt(np.maximum(u, tiny))) / np.sqrt(np.maximum(u, tiny)), 1.0) / (2 * np.pi)
    fs = lambda v: np.where(v > 0, np.sin(2 * np.pi * np.sqrt(np.maximum(v, tiny))) / np.sqrt(np.maximum(v, tiny)), 2 * np.pi)
    fc = lambda v: np.cos(2 * np.pi * np.sqrt(np.maximum(v, 0)))
    for name, f, lo, hi, deg, what in (("PV_AT", fa, 0.0, t8 * t8, 10, "atan(r) / (2 pi r), u = r^2 in [0, tan^2(pi/8)]"),
                                       ("PV_SN", fs, 0.0, 1 / 64, 6, "sin(2 pi t) / t, v = t^2 in [0, 1/64]"),
                                       ("PV_CS", fc, 0.0, 1 / 64, 6, "cos(2 pi t), v = t^2 in [0, 1/64]")):
        c = fit(f, lo, hi, deg)
        print(f"{name}[{len(c)}] = {{" + ", ".join(repr(float(v)) for v in c) + f"}};   // {what}: max error {max_err(c, f, lo, hi):.2e}")


if __name__ == "__main__":
    main()
