"""CPU check of the wavefront FFT's index algebra (tools/stft_fft_model.py): the exchange addresses of csrc/vp_fft.inc (16-byte elements,
additive skews) and csrc/vp_fft32.inc (8-byte elements, XOR layouts) replayed lane by lane against numpy.fft, with the LDS bank model of
MI355X_MICROARCH.md -- every lane group of every exchange must be conflict-free -- and the real-input split / merge with the lane
ownership of the bin pairs.  No GPU: this is the model the kernels were written from, kept honest by running it."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft_exchange_layouts_are_exact_and_conflict_free():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stft_fft_model.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    for key in ("fft err", "split err", "merge err", "roundtrip err"):
        m = re.search(key + r"\s+([0-9.eE+-]+)", out)
        assert m and float(m.group(1)) < 1e-10, (key, out)
    m = re.search(r"conflicts wr1 rd1 wr2 rd2:\s+(\d+) (\d+) (\d+) (\d+)", out)
    assert m and [int(x) for x in m.groups()] == [1, 1, 1, 1], out          # fp64: ds_write_b128 / ds_read_b128 groups
    m = re.search(r"f32 layout: fft512 err ([0-9.eE+-]+) worst bank multiplicity .* (\d+)\s*$", out, re.M)
    assert m and float(m.group(1)) < 1e-10 and int(m.group(2)) == 1, out      # f32: ds_write_b64 / ds_read_b64 groups
    assert "exchange 1 in registers: transposition ok" in out, out            # the permlane / DPP stages of fft_exchange1_regs
