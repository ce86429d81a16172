#!/usr/bin/env python3
"""Soak: the randomised parity tests of tests/test_gpu_parity.py over many more seeds than the suite carries
(GPU box only; prints a summary line, exits non-zero on the first mismatch).

    python tools/soak_fuzz.py [first_seed] [count]
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pytest  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def lite_case(seed):
    """Large-batch builds (two workgroups per CU) against the regular ones: a 260-stream batch in FAST mode must give, for
    the streams looked at, exactly what a small batch gives (same arithmetic, different wavefront layout)."""
    import numpy as np
    from vocoderproject_amd import BatchVocoderProcessor, VpError
    fs, N, params = T._fuzz_case(9000 + seed)
    S = 260
    nb = max(4, int(9000 * fs / 44100.0) // N)
    base = T._streams(13, N * nb, fs=fs)
    x = np.ascontiguousarray(np.tile(base, (20, 1, 1)))

    def run(xs):
        p = BatchVocoderProcessor(**params)
        p.prepareToPlay(fs, N, xs.shape[0])
        p.set_iir_mode("fast")
        p.set_yin_mode("xcorr")
        return p.run(xs)

    try:
        big = run(x)
    except VpError as e:
        assert e.code == -4, e
        raise pytest.skip.Exception("geometry beyond the LDS budget")
    pick = [0, 12, 130, 259]
    small = run(np.ascontiguousarray(x[pick]))
    T._assert_equal(big[pick], small, f"lite vs regular, seed {seed}: fs={fs} N={N} {params}")
    if params["pitchBool"] or params["vocBool"]:
        assert np.abs(big).max() > 0.01, "vacuous comparison"


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ok = skipped = 0
    for seed in range(first, first + count):
        fns = (lite_case,) if os.environ.get("VP_SOAK_LITE") else (T.test_randomised_configurations_bit_exact, T.test_randomised_configurations_with_extensions_bit_exact)
        for fn in fns:
            try:
                fn(seed)
                ok += 1
            except pytest.skip.Exception:
                skipped += 1
            except Exception:
                traceback.print_exc()
                print(f"FAILED: {fn.__name__}({seed})")
                return 1
    print(f"soak ok: {ok} cases bit-exact, {skipped} skipped (geometry beyond the LDS budget), seeds {first}..{first + count - 1}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
