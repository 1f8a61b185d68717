"""Diagnostic (not a test): the seeded random-configuration parity check of tests/test_gpu_parity.py over many seeds."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import test_gpu_parity as T

if os.environ.get("FUZZ_FRAG"):                           # widen the oracle's fragile-pixel allowance (the test's is 3 %)
    _check = T.check
    def check(*a, **kw):
        kw["frag_frac"] = float(os.environ["FUZZ_FRAG"])
        return _check(*a, **kw)
    T.check = check
seeds = [int(a) for a in sys.argv[1:]] if len(sys.argv) > 3 else range(int(sys.argv[1]), int(sys.argv[2]))
bad = []
for seed in seeds:
    try:
        T.test_random_configurations(seed)
    except Exception as e:                                   # noqa
        bad.append(seed)
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]} at {os.path.basename(tb.filename)}:{tb.lineno}: {tb.line}", flush=True)
    if seed % 20 == 0:
        print(f"... seed {seed}", flush=True)
print("failed seeds:", bad)
