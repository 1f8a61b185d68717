"""Diagnostic (not a test): where a view's time goes in the DROP-IN regime -- the reference's unchanged
render() (gaussian_renderer/__init__.py:53-95): activation getters and their backward as PyTorch kernels, GaussianRasterizer
with 16 object channels -- on S-nyc-1M at 1080p.  For each part: host time to enqueue it (perf_counter, no synchronise) and
device time (events), per view, one stream.

    python tests/diag_dropin.py [views=30]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa: F401,E402
import diff_gaussian_rasterization as D  # noqa: E402
from gsplat_attack.renderer import PipelineParams, render, _settings  # noqa: E402
from gsplat_attack.scenes import make_scene  # noqa: E402

NV = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
model, cams, _ = make_scene("nyc-1M", device=dev, n_views=8)
bg = torch.zeros(3, device=dev)
gc = torch.randn(3, cams[0].image_height, cams[0].image_width, device=dev)


def timed(fn, n):
    """-> (host ms per call to enqueue, device ms per call)"""
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    host = (time.perf_counter() - t0) / n * 1e3
    e1.record()
    torch.cuda.synchronize()
    return host, e0.elapsed_time(e1) / n


def getters_only(i=0):
    """the reference model's getters and their backward (exp / sigmoid / normalize / cat), nothing rasterised"""
    model.zero_grad()
    loss = (model.get_scaling.sum() + model.get_rotation.sum() + model.get_opacity.sum() + model.get_features.sum()
            + model.get_objects.sum())
    loss.backward()


pipes = {
    "drop-in: classic surface + 16 object channels": PipelineParams(skip_objects=False, fused_activations=False),
    "classic surface, objects off": PipelineParams(skip_objects=True, fused_activations=False),
    "fused raw-parameter path + 16 object channels": PipelineParams(skip_objects=False),
    "fused raw-parameter path, objects off (the headline)": PipelineParams(skip_objects=True),
}


def view(pipe):
    def f(i=0):
        model.zero_grad()
        render(cams[i % 8], model, pipe, bg)["render"].backward(gc)
    return f


# the rasteriser alone on the classic surface: activated tensors made once, as leaves
act = {k: v.detach().clone().requires_grad_(True) for k, v in dict(
    means3D=model.get_xyz, opacities=model.get_opacity, shs=model.get_features, sh_objs=model.get_objects,
    scales=model.get_scaling, rotations=model.get_rotation).items()}
m2d = torch.zeros_like(act["means3D"], requires_grad=True)


def raster_only(i=0):
    for t in list(act.values()) + [m2d]:
        t.grad = None
    st = _settings(cams[i % 8], model, PipelineParams(), bg, 1.0)
    color, radii, objs = D.GaussianRasterizer(raster_settings=st)(means2D=m2d, colors_precomp=None, cov3D_precomp=None, **act)
    color.backward(gc)


print(f"S-nyc-1M 1080p, one stream, {NV} views each: host ms to enqueue / device ms, per view")
h, d = timed(getters_only, NV)
print(f"  getters + their backward only (PyTorch kernels)      host {h:.3f}  device {d:.3f}")
h, d = timed(raster_only, NV)
print(f"  GaussianRasterizer fwd+bwd on ready activated tensors host {h:.3f}  device {d:.3f}")
for name, pipe in pipes.items():
    h, d = timed(view(pipe), NV)
    print(f"  {name:52s} host {h:.3f}  device {d:.3f}  -> {1e3 / max(h, d):.0f} views/s on one stream")

# ---- the same views dealt over four streams, with and without the host's wait for the pair count ---------------------
from gsplat_attack.streams import StreamRing  # noqa: E402
ring = StreamRing(4, dev)


def piped(pipe, n):
    def run(k):
        for i in range(k):
            with ring.next():
                model.zero_grad()
                render(cams[i % 8], model, pipe, bg)["render"].backward(gc)
        ring.join()
    run(12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0), host / n * 1e3


print("four streams: views/s (host ms per view to enqueue)")
for flags, tag in ((0, "pair count awaited by the host"), (D.FLAG_ASYNC_COUNT, "asynchronous pair count")):
    D.set_flags(flags)
    for name, pipe in pipes.items():
        r, h = piped(pipe, 2 * NV)
        print(f"  {tag:32s} {name:52s} {r:7.0f} views/s  (host {h:.3f})")
D.set_flags(0)
