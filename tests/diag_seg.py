"""Diagnostic (not a test): segmented vs whole-list backward walks, the per-group difference as a fraction of the scale,
and where the largest 'scaling' difference sits."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import diff_gaussian_rasterization as D
from gsplat_attack.scenes import make_scene
from gsplat_attack.renderer import PipelineParams, render
from test_gpu_segments import _run

D._load()
dev = torch.device("cuda:0")
model, cams, _ = make_scene("nyc-1M", device=dev, n_views=1, P=400000, width=960, height=544)
cam = cams[0]
bg = torch.tensor([0.2, 0.3, 0.1], device=dev)
gc = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(3)).to(dev)
whole = _run(model, cam, bg, gc, D.FLAG_NO_SEGMENTS)
seg = _run(model, cam, bg, gc, 0)
print({k: round(((seg[3][k] - whole[3][k]).abs().max() / whole[3][k].abs().max().clamp_min(1e-30)).item(), 7) for k in whole[3]})
img = render(cam, model, PipelineParams(skip_objects=True), bg)["render"]
offg = D.export_state(img, "offg").long()
G = D.export_state(img, "G").view(-1, 12)
rg = D.export_state(img, "ranges").view(-1, 2).long()
d = (seg[3]["scaling"] - whole[3]["scaling"]).abs()
top = torch.topk(d.flatten(), 5)
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    g = i // 3
    rows = int(offg[g + 1] - offg[g])
    print(f"g={g} axis={i % 3} diff={v:.3e} whole={whole[3]['scaling'][g].tolist()} seg={seg[3]['scaling'][g].tolist()} rows={rows} "
          f"rec={[round(x, 4) for x in G[g, :10].tolist()]} scale_raw={model._scaling[g].tolist()}")
print("longest list", int((rg[:, 1] - rg[:, 0]).max()))
