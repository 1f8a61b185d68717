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
