"""The offline (file-to-file) front end, SURVEY.md section 8f item 4: WAV I/O, batching of recordings of different
lengths, latency compensation.  The DSP itself needs the GPU (tests/test_gpu_parity.py::test_offline_*); here the
processor is a stand-in that only delays, so that the plumbing around the hot path is what gets checked."""
import os
import subprocess
import sys

import numpy as np
import pytest

from vocoderproject_amd import offline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("width,tol", [(2, 1.0 / 32767), (3, 1.0 / 8388607)])
def test_wav_round_trip(tmp_path, width, tol):
    rng = np.random.default_rng(1)
    y = rng.uniform(-1, 1, (2, 3001)).astype(np.float32)
    f = str(tmp_path / "a.wav")
    offline.write_wav(f, 44100, y, width=width)
    fs, z = offline.read_wav(f)
    assert fs == 44100 and z.shape == y.shape and z.dtype == np.float32
    assert np.abs(z - y).max() <= 0.5 * tol + 1e-7
    # the notebook's convention (cell 3): int16 / 32767.0
    if width == 2:
        offline.write_wav(f, 8000, np.array([1.0, -1.0, 0.0, 2.0]))           # mono, clipped
        fs, z = offline.read_wav(f)
        assert fs == 8000 and z.shape == (1, 4) and list(z[0]) == [1.0, -1.0, 0.0, 1.0]


def test_pack_unpack_ragged_batch():
    v = [np.arange(1, 2501, dtype=np.float32), np.arange(1, 701, dtype=np.float32), np.zeros(0, np.float32)]
    c = [np.ones(3000, np.float32), None, np.stack([np.full(10, 2.0), np.full(10, 3.0)]).astype(np.float32)]
    x, lens = offline.pack_batch(v, c, 1024, 1024)
    assert x.shape == (3, 3, 4096) and lens == [2500, 700, 0]                  # 2500 + 1024 -> next multiple of 1024
    assert x[0, 0, 2499] == 2500 and x[0, 0, 2500] == 0 and np.all(x[0, 1:, :3000] == 1) and np.all(x[0, 1:, 3000:] == 0)
    assert np.all(x[1, 1:] == 0)                                              # no side chain: zeros (MyBuffer.cpp:93-102)
    assert np.all(x[2, 1, :10] == 2) and np.all(x[2, 2, :10] == 3)
    y = np.zeros((3, 2, 4096), np.float32)
    y[:, :, 1024:] = x[:, :2, :-1024]                                         # a pure delay of `latency`
    out = offline.unpack_batch(y, lens, 1024)
    assert [o.shape for o in out] == [(2, 2500), (2, 700), (2, 0)]
    np.testing.assert_array_equal(out[0][0], v[0])
    with pytest.raises(ValueError):
        offline.pack_batch([], None, 1024, 1024)
    with pytest.raises(ValueError):
        offline.pack_batch([np.zeros((2, 5))], None, 1024, 1024)


class _DelayProcessor:
    """BatchVocoderProcessor's interface, DSP replaced by a delay of `latency` samples (voice -> both channels)."""

    def __init__(self, latency=1024):
        self._lat, self.calls, self.shifts, self.sparams = latency, [], {}, {}

    def prepareToPlay(self, fs, N, S):
        self.fs, self.N, self.n_streams = fs, N, S
        self.hist = np.zeros((S, self._lat), np.float32)

    def set_iir_mode(self, m): self.iir = m
    def set_yin_mode(self, m): self.yin = m
    def setStreamParameter(self, s, k, v): self.sparams[(s, k)] = v
    def setPitchShift(self, semi, on=True, stream=-1): self.shifts[stream] = semi

    @property
    def latency(self):
        return self._lat

    def process(self, x):
        assert x.shape == (self.n_streams, 3, self.N) and x.flags.c_contiguous
        self.calls.append(1)
        cat = np.concatenate([self.hist, x[:, 0]], axis=1)
        self.hist = cat[:, -self._lat:]
        d = cat[:, :self.N]
        return np.stack([d, d], axis=1)



class _DelayProcessorBlocks(_DelayProcessor):
    def process_blocks(self, xb):
        assert xb.shape[1:] == (self.n_streams, 3, self.N) and xb.flags.c_contiguous
        k = xb.shape[0]
        out = np.stack([self.process(np.ascontiguousarray(xb[j])) for j in range(k)])
        self.calls[-k:] = [k]
        return out


@pytest.mark.parametrize("with_blocks", [True, False])
def test_render_aligns_output_with_input(with_blocks):
    rng = np.random.default_rng(2)
    voices = [rng.normal(0, 0.1, n).astype(np.float32) for n in (5000, 1024, 1, 12288)]
    fake = (_DelayProcessorBlocks if with_blocks else _DelayProcessor)(latency=1024)
    outs = offline.render(voices, None, 44100.0, processor=fake, blocks_per_call=4, shift=[None, 12.0, -3.0, None],
                          stream_params=[dict(keyPitch=3), None, None, dict(gainPitch=-6.0)], iir_mode="fast", yin_mode="xcorr")
    for v, o in zip(voices, outs):
        assert o.shape == (2, v.size)
        np.testing.assert_array_equal(o[0], v)                                # delay removed: output t <-> input t
        np.testing.assert_array_equal(o[1], v)
    assert fake.shifts == {1: 12.0, 2: -3.0} and fake.sparams == {(0, "keyPitch"): 3, (3, "gainPitch"): -6.0}
    assert (fake.iir, fake.yin, fake.n_streams, fake.N) == ("fast", "xcorr", 4, 1024)
    nb = (12288 + 1024) // 1024
    assert sum(fake.calls) == nb and (max(fake.calls) == 4 if with_blocks else max(fake.calls) == 1)
    with pytest.raises(ValueError):
        offline.render(voices, None, 44100.0, processor=_DelayProcessor(), shift=[1.0])


def test_cli_fails_loudly_without_gpu(tmp_path):
    # no CPU fallback anywhere in the product: on a box without a GPU the command line must fail, not emit silence
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    f = str(tmp_path / "v.wav")
    offline.write_wav(f, 44100, np.zeros(2048))
    r = subprocess.run([sys.executable, "-m", "vocoderproject_amd.offline", "pitch", f, "--out-dir", str(tmp_path / "o")],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0, r.stdout + r.stderr
    assert not os.path.exists(str(tmp_path / "o" / "v_pitch.wav"))
