"""Diagnostic (not a test): what shares the chip with what when views are pipelined over S streams,
WITHOUT a profiler -- rocprofv3's kernel trace changes the picture (profiles/coresidency.py on the round-4 trace: a
compositor and an HBM-bound kernel are never in flight together under the profiler, and a view takes 0.86 ms instead of
0.70).  The library's own per-stage HIP events (gsr_profile_timeline: start / end of every stage on its stream) give the
spans; classes: compositor = render_fwd / render_bwd (VALU-bound), hbm = preprocess_bwd (K8+K9), chain = preprocess /
depth_sort / bin / tile_sort (latency-bound front end; `preprocess` includes the wait for the colour kernel's side stream).

    python tests/diag_coresidency.py [streams=4] [views=64]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa: F401,E402
import diff_gaussian_rasterization as D  # noqa: E402
from gsplat_attack.renderer import PipelineParams, render  # noqa: E402
from gsplat_attack.scenes import make_scene  # noqa: E402
from gsplat_attack.streams import StreamRing  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NV = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
model, cams, spec = make_scene("nyc-1M", device=dev, n_views=8)
pipe = PipelineParams(skip_objects=True)
bg = torch.zeros(3, device=dev)
gc = torch.randn(3, cams[0].image_height, cams[0].image_width, device=dev)
lib = D._load()
lib.gsr_profile_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.gsr_profile_timeline.restype = ctypes.c_int
ring = StreamRing(S, dev)


def run(n):
    for i in range(n):
        with ring.next():
            model.zero_grad()
            render(cams[i % 8], model, pipe, bg)["render"].backward(gc)
    ring.join()
    torch.cuda.synchronize()


run(3 * S)
D.profile(True)
run(NV)
cap = 16 * NV
buf = (ctypes.c_float * (3 * cap))()
n = lib.gsr_profile_timeline(buf, cap)
D.profile(False)
KL = {"render_fwd": "compositor", "render_bwd": "compositor", "preprocess_bwd": "hbm"}
spans = [(buf[3 * i + 1], buf[3 * i + 2], D.GSR_STAGES[int(buf[3 * i])]) for i in range(n)]
t_end = max(s[1] for s in spans)
lo, hi = 0.2 * t_end, 0.8 * t_end
pts = []
for a, b, name in spans:
    a, b = max(a, lo), min(b, hi)
    if b > a:
        k = KL.get(name, "chain")
        pts.append((a, 1, k))
        pts.append((b, -1, k))
pts.sort()
live = {"compositor": 0, "hbm": 0, "chain": 0}
acc, t_prev, infl = {}, pts[0][0], 0.0
for t, d, k in pts:
    dt = t - t_prev
    if dt > 0:
        key = tuple(sorted(c for c, v in live.items() if v > 0))
        acc[key] = acc.get(key, 0.0) + dt
        infl += dt * sum(live.values())
    live[k] += d
    t_prev = t
total = sum(acc.values())


def share(pred):
    return sum(v for k, v in acc.items() if pred(k)) / total


print(f"== {S} streams, {NV} views, no profiler: {t_end / NV:.3f} ms/view = {NV / t_end * 1e3:.0f} views/s; steady-state window "
      f"{total:.1f} ms, mean stages in flight {infl / total:.2f}")
print(f"  compositor in flight            {share(lambda k: 'compositor' in k):.3f}")
print(f"  hbm-bound stage in flight       {share(lambda k: 'hbm' in k):.3f}")
print(f"  chain stage in flight           {share(lambda k: 'chain' in k):.3f}")
print(f"  compositor AND hbm              {share(lambda k: 'compositor' in k and 'hbm' in k):.3f}")
print(f"  compositor AND chain            {share(lambda k: 'compositor' in k and 'chain' in k):.3f}")
print(f"  hbm with NO compositor          {share(lambda k: 'hbm' in k and 'compositor' not in k):.3f}")
print(f"  chain only                      {share(lambda k: k == ('chain',)):.3f}")
print(f"  nothing in flight               {share(lambda k: k == ()):.3f}")
dur = {}
for a, b, name in spans:
    d = dur.setdefault(name, [0.0, 0])
    d[0] += b - a
    d[1] += 1
print("  mean span per stage (ms): " + ", ".join(f"{k} {v[0] / v[1]:.3f}" for k, v in dur.items()))
