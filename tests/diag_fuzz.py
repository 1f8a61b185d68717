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
tot = dict(px=0, fragile=0, need_b=0, neither=0, worst_neither=0.0, worst_solid=0.0, runs=0, all_fragile=0)
for seed in seeds:
    try:
        T.check.last_yardstick = None
        T.test_random_configurations(seed)
    except Exception as e:                                   # noqa
        bad.append(seed)
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]} at {os.path.basename(tb.filename)}:{tb.lineno}: {tb.line}", flush=True)
    y = getattr(T.check, "last_yardstick", None)
    if y is not None:                                        # the float32 yardstick of this draw (round 4)
        tot["runs"] += 1
        tot["px"] += y["n"]
        tot["fragile"] += round(y["fragile"] * y["n"])
        tot["need_b"] += round(y["need_b"] * y["n"])
        tot["neither"] += y["neither_px"]
        tot["worst_neither"] = max(tot["worst_neither"], y["worst_neither"])
        tot["worst_solid"] = max(tot["worst_solid"], y["worst_solid"])
        tot["all_fragile"] += 1 if y["fragile"] >= 1.0 else 0
    if seed % 20 == 0:
        print(f"... seed {seed}", flush=True)
print("failed seeds:", bad)
print(f"float32 yardstick over {tot['runs']} draws: {tot['px']} pixels, {tot['fragile']} fragile ({tot['fragile'] / max(tot['px'], 1):.4f}), "
      f"{tot['need_b']} need the float32 outcome (clause B), {tot['neither']} fragile pixels on neither clause "
      f"({tot['neither'] / max(tot['fragile'], 1):.5f} of the fragile ones, worst {tot['worst_neither']:.2e}); worst solid-pixel error "
      f"{tot['worst_solid']:.2e}; draws with every pixel fragile: {tot['all_fragile']}")
