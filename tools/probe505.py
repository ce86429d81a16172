"""Probe of soak seed 505 (fs 8000, N 64, lpcVoice 100 > window length 92): GPU vs oracle with and without FTZ/DAZ, per mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as T
from oracle import oracle_py as O
from vocoderproject_amd import BatchVocoderProcessor
fs, N = 8000.0, 64
for params in (dict(lpcVoice=100, lpcPitch=4, lpcSynth=11, keyPitch=10, gainPitch=-1.15, gainVoc=-12.76, pitchBool=1, vocBool=1),
               dict(lpcVoice=100, lpcSynth=11, gainVoc=-12.76, pitchBool=0, vocBool=1),
               dict(lpcVoice=60, lpcSynth=11, pitchBool=0, vocBool=1),
               dict(lpcVoice=100, lpcPitch=4, pitchBool=1, vocBool=0)):
    S = 4
    Tn = 68 * N
    x = T._streams(S, Tn, fs=fs)
    p = BatchVocoderProcessor(**params); p.prepareToPlay(fs, N, S)
    print(params, p.geometry())
    got = p.run(x)
    for ftz in (True, False):
        ref = []
        for s_ in range(S):
            o = O.OracleStream(**params); o.prepare_to_play(fs, N); o.set_ftz(ftz); ref.append(o.run(x[s_]))
        ref = np.stack(ref)
        bad = np.argwhere(got != ref)
        print("   oracle ftz", ftz, "differing samples", len(bad), "first", bad[0] if len(bad) else None, "max abs", float(np.abs(got - ref).max()), "finite", bool(np.isfinite(ref).all()))
