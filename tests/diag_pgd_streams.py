"""Diagnostic (not a test): one PGD iteration of config 4 on one GPU (8 views, five groups) over 1 and 4 streams, with the
surrogate detector and with a trivial loss (image . fixed tensor) -- which part of the iteration the streams do not hide.
Run: python tests/diag_pgd_streams.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-gaussian-splat-attack_amd"))
import torch
from gsplat_attack.scenes import make_scene
from gsplat_attack.attack import pgd_attack, SurrogateDetector

dev = torch.device("cuda:0")
model, cams, _ = make_scene("nyc-1M", device=dev, n_views=8)
H, W = cams[0].image_height, cams[0].image_width
gc = torch.randn(3, H, W, generator=torch.Generator().manual_seed(99)).to(dev)
det = SurrogateDetector().to(dev)
groups = ("color", "position", "scaling", "rotation", "opacity")
for name, loss in (("surrogate", det), ("dot", lambda im: (im[0] * gc).sum())):
    for streams in (1, 4):
        m = model.clone()
        pgd_attack(m, cams, iters=2, groups=groups, loss_fn=loss, streams=streams)
        torch.cuda.synchronize()
        recs = []
        pgd_attack(m, cams, iters=6, groups=groups, loss_fn=loss, streams=streams, log=recs.append)
        ms = sorted(r["seconds"] * 1e3 for r in recs)
        print(f"{name:10s} streams {streams}: median iteration {ms[len(ms) // 2]:.3f} ms (8 views)", flush=True)
        del m
