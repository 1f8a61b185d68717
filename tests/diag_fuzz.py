"""Diagnostic (not a test): the seeded random-configuration parity check of tests/test_gpu_parity.py over many seeds."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import test_gpu_parity as T

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    try:
        T.test_random_configurations(seed)
    except Exception as e:                                   # noqa
        bad.append(seed)
        print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    if seed % 20 == 0:
        print(f"... seed {seed}", flush=True)
print("failed seeds:", bad)
