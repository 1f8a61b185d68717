"""Diagnostic (not a test): a few thousand pipelined views; workspace pool and torch allocator must stay flat."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import diff_gaussian_rasterization as D
from gsplat_attack.scenes import make_scene
from gsplat_attack.renderer import PipelineParams, render
from gsplat_attack.streams import StreamRing

dev = torch.device("cuda:0")
model, cams, spec = make_scene("nyc-1M", device=dev, n_views=8)
pipe = PipelineParams(skip_objects=True)
bg = torch.zeros(3, device=dev)
gc = torch.randn(3, cams[0].image_height, cams[0].image_width, device=dev)
ring = StreamRing(4, dev)
def run(n):
    for i in range(n):
        with ring.next():
            model.zero_grad()
            out = render(cams[i % 8], model, pipe, bg)
            out["render"].backward(gc)
    ring.join(); torch.cuda.synchronize()
run(30)
p0, t0 = D.pool_bytes(), torch.cuda.memory_reserved()
for rep in range(4):
    t = time.perf_counter(); run(750); dt = time.perf_counter() - t
    print(f"round {rep}: {750 / dt:7.1f} views/s  pool {D.pool_bytes() / 2**20:8.1f} MiB  torch reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB", flush=True)
assert D.pool_bytes() <= p0 * 1.3 + (64 << 20), (p0, D.pool_bytes())
assert torch.cuda.memory_reserved() <= t0 * 1.3 + (256 << 20)
print("flat: ok")

# the same with batches of views of changing size through one launch chain (round 6): summed gradients and per-view buckets
from gsplat_attack.renderer import render_batch
P = int(model.get_xyz.shape[0])
H, W = cams[0].image_height, cams[0].image_width
bucket = D.GradBucket(P, dev)
bset = D.GradBucketSet(8, P, dev)
def run_batches(n):
    for i in range(n):
        B = (5, 8, 3, 1, 8)[i % 5]
        group = [cams[(i + k) % 8] for k in range(B)]
        tgt = bset if i % 2 else bucket
        if tgt is bucket:
            bucket.reset()
        out = render_batch(group, model, PipelineParams(skip_objects=True, grad_bucket=tgt), bg)
        out["render"].backward(gc.unsqueeze(0).expand(B, 3, H, W))
    torch.cuda.synchronize()
run_batches(10)
p0, t0 = D.pool_bytes(), torch.cuda.memory_reserved()
for rep in range(3):
    t = time.perf_counter(); run_batches(200); dt = time.perf_counter() - t
    print(f"batches round {rep}: {200 * 5.0 / dt:7.1f} views/s  pool {D.pool_bytes() / 2**20:8.1f} MiB  torch reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB", flush=True)
assert D.pool_bytes() <= p0 * 1.3 + (64 << 20), (p0, D.pool_bytes())
assert torch.cuda.memory_reserved() <= t0 * 1.3 + (256 << 20)
print("batches flat: ok")

# a colour attack on a batch with kept contexts (round 6, second session): the batch's re-render + colour-only backward and the
# forward-only pair batch of the success renders, the colours stepped in place between iterations
from gsplat_attack.renderer import render_pair_batch
cache = D.RenderCache()
cpipe = PipelineParams(skip_objects=True, viewspace_grad=False, render_cache=cache)
for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
    getattr(model, n).requires_grad_(False)
back = model.clone()
gcb = gc.unsqueeze(0).expand(8, 3, H, W).contiguous()
def run_colour(n):
    for i in range(n):
        model.zero_grad()
        out = render_batch(cams, model, cpipe, bg)
        out["render"].backward(gcb)
        with torch.no_grad():
            model._features_dc.add_(1e-3)
            if i % 4 == 0:
                render_pair_batch(cams, model, back, cpipe, bg)
    torch.cuda.synchronize()
run_colour(5)
p0, t0 = D.pool_bytes(), torch.cuda.memory_reserved()
for rep in range(3):
    t = time.perf_counter(); run_colour(100); dt = time.perf_counter() - t
    print(f"kept batch round {rep}: {100 * 8.0 / dt:7.1f} views/s (re-render + SH backward)  pool {D.pool_bytes() / 2**20:8.1f} MiB  "
          f"torch reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB  cache hits {cache.hits} misses {cache.misses}", flush=True)
assert D.pool_bytes() <= p0 * 1.3 + (64 << 20), (p0, D.pool_bytes())
assert torch.cuda.memory_reserved() <= t0 * 1.3 + (256 << 20)
assert cache.misses == 2
print("kept batch flat: ok")
