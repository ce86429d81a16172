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


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ok = skipped = 0
    for seed in range(first, first + count):
        for fn in (T.test_randomised_configurations_bit_exact, T.test_randomised_configurations_with_extensions_bit_exact):
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
