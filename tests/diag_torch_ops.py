"""Diagnostic: which PyTorch-side device ops run per view (torch profiler, one stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
from torch.profiler import profile, ProfilerActivity
from gsplat_attack.scenes import make_scene
from gsplat_attack.renderer import PipelineParams, render
dev = torch.device("cuda:0")
model, cams, _ = make_scene("nyc-1M", device=dev, P=200000, n_views=2)
pipe = PipelineParams(skip_objects=True); bg = torch.zeros(3, device=dev)
gc = torch.randn(3, cams[0].image_height, cams[0].image_width, device=dev)
def step():
    model.zero_grad(); out = render(cams[0], model, pipe, bg); out["render"].backward(gc)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0]
for k, c, t in sorted(rows, key=lambda r: -r[2])[:25]:
    print(f"{c/5:5.1f}/view {t/5:8.1f} us/view  {k[:100]}")
