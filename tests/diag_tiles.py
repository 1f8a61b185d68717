"""Diagnostic (not a test): tile-list statistics of the benchmark scene + per-stage times."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import diff_gaussian_rasterization as D
from gsplat_attack.scenes import make_scene
from gsplat_attack.renderer import PipelineParams, render

dev = torch.device("cuda:0")
model, cams, spec = make_scene("nyc-1M", device=dev, n_views=8)
pipe = PipelineParams(skip_objects=True)
bg = torch.zeros(3, device=dev)
for ci in (0, 3):
    cam = cams[ci]
    out = render(cam, model, pipe, bg)
    img = out["render"]
    rg = D.export_state(img, "ranges").view(-1, 2).long().cpu()
    ln = (rg[:, 1] - rg[:, 0])
    nc = D.export_state(img, "n_contrib").long().cpu().view(cam.image_height, cam.image_width)
    N = D.last_num_rendered(img)
    print(f"cam {ci}: N={N} tiles={ln.numel()} empty={(ln==0).sum().item()} mean={ln.float().mean():.1f} "
          f"p50={ln.float().median():.0f} p90={ln.float().quantile(0.9):.0f} p99={ln.float().quantile(0.99):.0f} max={ln.max().item()}")
    # per tile: max n_contrib (entries actually walked) 
    H, W = nc.shape
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = torch.zeros(gy * 16, gx * 16, dtype=torch.long); pad[:H, :W] = nc
    tmax = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(dim=1).values
    print(f"   walked entries/tile: sum={tmax.sum().item()} ({tmax.sum().item()/max(N,1):.2f} of N) mean={tmax.float().mean():.1f} "
          f"p99={tmax.float().quantile(0.99):.0f} max={tmax.max().item()}")
    srt, _ = torch.sort(tmax, descending=True)
    print("   top-10 walked:", srt[:10].tolist(), " sum of top 256:", srt[:256].sum().item())
