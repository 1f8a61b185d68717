"""Diagnostic (not a test): what fraction of the lanes of a compositor strip evaluation carry a pixel that passes the
alpha test?  Evaluated in torch from the forward's exported lists on a sample of tiles of the benchmark scene, for three
ways of cutting a 16x16 tile into four 64-pixel parts."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import diff_gaussian_rasterization as D
from gsplat_attack.scenes import make_scene
from gsplat_attack.renderer import PipelineParams, render

dev = torch.device("cuda:0")
model, cams, spec = make_scene(os.environ.get("DIAG_SCENE", "nyc-1M"), device=dev, n_views=1)
cam = cams[0]
out = render(cam, model, PipelineParams(skip_objects=True), torch.zeros(3, device=dev))
img = out["render"]
H, W = cam.image_height, cam.image_width
gx = (W + 15) // 16
rg = D.export_state(img, "ranges").view(-1, 2).long()
pairs = D.export_state(img, "pair_rank").long()
G = D.export_state(img, "G").view(-1, 12)
ncon = D.export_state(img, "n_contrib").view(H, W).long()
gen = torch.Generator().manual_seed(0)
tiles = torch.randperm(rg.shape[0], generator=gen)[:300].tolist()
yy, xx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
parts = {"16x4 strips": (yy // 4), "8x8 quadrants": (yy // 8) * 2 + (xx // 8), "4x16 columns": (xx // 4)}
stat = {k: [0, 0] for k in parts}          # evaluated lanes, useful lanes
for t in tiles:
    s, e = int(rg[t, 0]), int(rg[t, 1])
    if e <= s:
        continue
    tx, ty = t % gx, t // gx
    px = (tx * 16 + xx).float(); py = (ty * 16 + yy).float()
    inside = (px < W) & (py < H)
    nc = torch.zeros(16, 16, dtype=torch.long, device=dev)
    nc[inside] = ncon[(ty * 16 + yy)[inside], (tx * 16 + xx)[inside]]
    g = pairs[s:e] & ((1 << 28) - 1)
    rec = G[g]
    dx = rec[:, 0, None, None] - px[None]; dy = rec[:, 1, None, None] - py[None]
    power = -0.5 * (rec[:, 2, None, None] * dx * dx + rec[:, 4, None, None] * dy * dy) - rec[:, 3, None, None] * dx * dy
    alpha = torch.clamp(rec[:, 5, None, None] * torch.exp(power), max=0.99)
    pos = torch.arange(1, e - s + 1, device=dev)[:, None, None]
    useful = (power <= 0) & (alpha >= 1 / 255.0) & inside[None] & (pos <= nc[None])      # what the backward needs
    for name, part in parts.items():
        for k in range(4):
            m = (part == k)[None]
            hit = (useful & m).flatten(1).any(dim=1)              # parts some pixel of which is useful: they get evaluated
            stat[name][0] += int(hit.sum()) * 64
            stat[name][1] += int((useful & m).flatten(1).sum(dim=1)[hit].sum())
for name, (ev, us) in stat.items():
    print(f"{name:14s}: {ev // 64:9d} part evaluations, useful lanes {us / max(ev, 1):.3f}")
