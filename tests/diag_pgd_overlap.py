"""Diagnostic (not a test): does the success re-render overlap with the next iteration's forward?  Times, on S-nyc-1M:
two forward-only renders one after the other on one stream / on two streams at once, and BASELINE config 3's PGD iteration
(colour L2, one view, re-render after every step) with overlap_success off / on, with and without a per-iteration log.
    python tests/diag_pgd_overlap.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from gsplat_attack.attack import SurrogateDetector, pgd_attack
from gsplat_attack.renderer import PipelineParams, render
from gsplat_attack.scenes import make_scene

dev = torch.device("cuda:0")
model, cams, _ = make_scene("nyc-1M", device=dev, n_views=2)
pipe = PipelineParams(skip_objects=True)
bg = torch.zeros(3, device=dev)
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def two_serial():
    with torch.no_grad(), torch.cuda.stream(s0):
        render(cams[0], model, pipe, bg)
        render(cams[1], model, pipe, bg)


def two_streams():
    with torch.no_grad():
        with torch.cuda.stream(s0):
            render(cams[0], model, pipe, bg)
        with torch.cuda.stream(s1):
            render(cams[1], model, pipe, bg)


print(f"two forwards, one stream {timed(two_serial):.3f} ms; two streams {timed(two_streams):.3f} ms")
det = SurrogateDetector().to(dev)
never = lambda im, i: False   # noqa
for overlap in (False, True):
    for with_log in (True, False):
        m = model.clone()
        kw = dict(groups=("color",), loss_fn=det, streams=1, alpha=0.5, epsilon=5.0, success_fn=never, background=None,
                  overlap_success=overlap)
        pgd_attack(m, cams[:1], iters=3, **kw)
        torch.cuda.synchronize()
        recs = []
        t0 = time.perf_counter()
        pgd_attack(m, cams[:1], iters=20, log=recs.append if with_log else None, **kw)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        med = sorted(r["seconds"] for r in recs)[10] * 1e3 if recs else float("nan")
        print(f"overlap {overlap!s:5} log {with_log!s:5}: {wall:.3f} ms per iteration (median logged iteration {med:.3f} ms)")
