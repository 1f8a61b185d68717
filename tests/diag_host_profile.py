"""Diagnostic (not a test): where the HOST's time per view goes in the pipelined regime (round 5: four streams run at the
rate the single Python thread can enqueue -- 0.68 ms of host time per 0.70 ms view).  cProfile over N views dealt over four
streams + wall time inside the two C entry points.

    python tests/diag_host_profile.py [views=200]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa: F401,E402
import diff_gaussian_rasterization as D  # noqa: E402
from gsplat_attack.renderer import PipelineParams, render  # noqa: E402
from gsplat_attack.scenes import make_scene  # noqa: E402

NV = int(sys.argv[1]) if len(sys.argv) > 1 else 200
MODE = sys.argv[2] if len(sys.argv) > 2 else "fused"          # "dropin": the classic surface + object channels, one stream
dev = torch.device("cuda:0")
model, cams, _ = make_scene("nyc-1M", device=dev, n_views=8)
bg = torch.zeros(3, device=dev)
gc = torch.randn(3, cams[0].image_height, cams[0].image_width, device=dev)
pipe = PipelineParams(skip_objects=True) if MODE == "fused" else PipelineParams(skip_objects=False, fused_activations=False)
if MODE == "dropin0":
    with torch.no_grad():
        model._objects_dc.zero_()
streams = [torch.cuda.Stream(device=dev) for _ in range(4 if MODE == "fused" else 1)]
lib = D._load()
acc = {}


class Timed:
    """wall time inside one ctypes entry point"""

    def __init__(self, fn, name):
        self.fn, self.name = fn, name

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        d = acc.setdefault(self.name, [0.0, 0])
        d[0] += time.perf_counter() - t0
        d[1] += 1
        return r


def run(n):
    for s_ in streams:
        s_.wait_stream(torch.cuda.current_stream(dev))
    for i in range(n):
        with torch.cuda.stream(streams[i % len(streams)]):
            model.zero_grad()
            render(cams[i % 8], model, pipe, bg)["render"].backward(gc)
    for s_ in streams:
        torch.cuda.current_stream(dev).wait_stream(s_)


run(16)
torch.cuda.synchronize()
for name in ("gsr_forward_raw", "gsr_backward_raw", "gsr_backward_raw_into", "gsr_forward", "gsr_backward"):
    setattr(lib, name, Timed(getattr(lib, name), name))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
run(NV)
pr.disable()
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{MODE}: {NV} views over {len(streams)} stream(s): {NV / wall:.0f} views/s, host {host / NV * 1e3:.3f} ms per view to enqueue (profiler on)")
for k, (t, c) in acc.items():
    print(f"  inside {k}: {t / NV * 1e3:.3f} ms per view ({c} calls)")
st = pstats.Stats(pr)
st.sort_stats("tottime")
st.print_stats(28)
