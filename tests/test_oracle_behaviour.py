"""Behavioural checks of the CPU oracle that mirror what the survey observed on the compiled
reference: block-size independence with an open gate, block-size
dependence across the -60 dB gate, FTZ independence on audio-range inputs, sanitizer cleanliness."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle_py as O
from vocoderproject_amd.synth import make_streams

FS = 44100.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(x, N, **params):
    o = O.OracleStream(**params)
    o.prepare_to_play(FS, N)
    return o.run(x), o


def test_open_gate_output_is_block_size_independent():
    # "an always-open gate gives bit-identical outputs for N in {100,128,256,512,1024}"
    x = make_streams(1, 25600).numpy()[0]
    ref, _ = run(x, 1024)
    for N in (100, 128, 256, 512):
        y, _ = run(x[:, : (25600 // N) * N], N)
        np.testing.assert_array_equal(y, ref[:, : y.shape[1]])


def test_gate_crossing_depends_on_block_size():
    # alternating -100 dB / -14 dB every 30000 samples gives different outputs per N
    x = make_streams(1, 30000 * 4 + 1024 * 3).numpy()[0]
    env = np.where((np.arange(x.shape[1]) // 30000) % 2 == 0, 1e-5 / 0.2, 1.0).astype(np.float32)
    x = (x * env).astype(np.float32)
    T = (x.shape[1] // 1024) * 1024
    y1, o1 = run(x[:, :T], 1024)
    y2, _ = run(x[:, :T], 512)
    assert np.abs(y1).max() > 0.05
    assert not np.array_equal(y1, y2)
    assert any(t["gated"] for t in _traces(x[:, :T], 1024))


def _traces(x, N):
    o = O.OracleStream(vocBool=0)
    o.prepare_to_play(FS, N)
    return o.run(x, trace=True)[1]


def test_ftz_daz_makes_no_difference_on_audio_inputs():
    x = make_streams(1, 1024 * 24).numpy()[0]
    a, _ = run(x, 1024)
    o = O.OracleStream()
    o.prepare_to_play(FS, 1024)
    o.set_ftz(0)
    np.testing.assert_array_equal(o.run(x), a)


def test_unvoiced_and_silent_streams_do_not_trip_ub_sites():
    rng = np.random.default_rng(3)
    x = np.zeros((3, 1024 * 40), np.float32)
    x[0] = (rng.standard_normal(x.shape[1]) * 0.05).astype(np.float32)          # noise: mostly unvoiced
    x[0, 1024 * 10:1024 * 20] = 0                                                # a silent gap
    x[0, 1024 * 20:] += make_streams(1, 1024 * 20).numpy()[0, 0]                # then voiced
    x[1] = x[2] = make_streams(1, x.shape[1]).numpy()[0, 1]
    y, o = run(x, 1024)
    ub = o.ub_counters()
    assert ub[1] == 0 and ub[3] == 0 and ub[4] == 0          # Q3, yinTemp[tauMax], mark overflow never reached
    assert np.isfinite(y).all()


def test_explicit_geometry_and_rejections():
    o = O.OracleStream(lpcVoice=48, lpcPitch=48, lpcSynth=30)
    o.prepare_explicit(48000.0, 2048, 2048, 1536, 2048, 512)       # BASELINE configs[4] geometry
    g = o.geometry()
    assert (g["tauMax"], g["latency"], g["inSize"], g["C"]) == (480, 2048, 6144, 512)
    with pytest.raises(ValueError):
        O.OracleStream().prepare_explicit(44100.0, 1024, 1024, 768, 512, 100)    # Invalid overlap
    with pytest.raises(ValueError):
        O.OracleStream().prepare_explicit(44100.0, 1024, 1000, 700, 512, 128)    # F % (F-H) != 0
    with pytest.raises(ValueError):
        O.OracleStream().set_param("lpcSynth", 48)                                # range end is 30


def test_oracle_is_clean_under_asan_ubsan():
    lib = O.build(asan=True)
    code = (
        "import ctypes, numpy as np, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from oracle import oracle_py as O\n"
        f"O._LIB = None; O.build = lambda force=False, asan=False: {lib!r}\n"
        "from vocoderproject_amd.synth import make_streams\n"
        "x = make_streams(1, 1024 * 12).numpy()[0]\n"
        "o = O.OracleStream(); o.prepare_to_play(44100.0, 1024); y = o.run(x)\n"
        "o = O.OracleStream(lpcVoice=100, lpcPitch=100, lpcSynth=30); o.prepare_to_play(48000.0, 480); "
        "y = o.run(x[:, :480 * 20]); print('ok', float(abs(y).max()))\n")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]


def _first_strong_lag(y, lo=40, hi=700):
    y = y - y.mean()
    ac = np.array([np.dot(y[:-k], y[k:]) for k in range(lo, hi)])
    ac /= np.dot(y, y)
    k = int(np.argmax(ac > 0.8 * ac.max()))
    while k + 1 < ac.size and ac[k + 1] > ac[k]:
        k += 1
    return lo + k


@pytest.mark.parametrize("semitones,ratio", [(12.0, 0.5), (-12.0, 2.0), (7.0, 2.0 ** (-7 / 12))])
def test_fixed_pitch_shift_extension_moves_the_fundamental(semitones, ratio):
    # Extension (no reference counterpart, BASELINE configs[1]): beta = 2^(semitones/12) instead of the key's note.
    # A 147 Hz harmonic tone (period 300) must come out with its period scaled by 1/beta.
    T, n = 300, 1024 * 40
    t = np.arange(n)
    v = sum(np.sin(2 * np.pi * h * t / T) / h for h in range(1, 9)) * 0.2
    x = np.zeros((3, n), np.float32)
    x[0] = v
    o = O.OracleStream(vocBool=0)
    o.prepare_to_play(FS, 1024)
    o.set_pitch_shift(semitones)
    y = o.run(x)[0]
    got = _first_strong_lag(y[1024 * 10:].astype(np.float64))
    assert abs(got - T * ratio) <= 0.02 * T * ratio + 1, (got, T * ratio)
    with pytest.raises(ValueError):
        o.set_pitch_shift(12.5)
    o2 = O.OracleStream(vocBool=0)                       # switched off again: the plugin's own correction
    o2.prepare_to_play(FS, 1024)
    o2.set_pitch_shift(5.0)
    o2.set_pitch_shift(0.0, on=False)
    o3 = O.OracleStream(vocBool=0)
    o3.prepare_to_play(FS, 1024)
    np.testing.assert_array_equal(o2.run(x[:, :1024 * 8]), o3.run(x[:, :1024 * 8]))
