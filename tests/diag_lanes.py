"""Diagnostic (CPU only, not a test): how many 64-pixel sub-blocks of a 16x16 tile the compositors would have to evaluate
per (tile, Gaussian) pair under different lane layouts -- the 16x4 strips K6/K7 use, 8x8 quadrants
and 4x16 column strips -- on the benchmark scene, counted exactly: a sub-block is evaluated when at least one of its pixel
centres passes the reference's alpha >= 1/255 test (that is the wave ballot in front of the strip body).

    python tests/diag_lanes.py [scene] [view] [max_pairs]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "3d-gaussian-splat-attack_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gsplat_attack.scenes import make_scene  # noqa: E402
from oracle import oracle_r as O  # noqa: E402


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "nyc-1M"
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    max_pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_000
    torch.manual_seed(0)
    model, cams, _ = make_scene(key, n_views=max(view + 1, 1))
    cam = cams[view]
    H, W = cam.image_height, cam.image_width
    st = O.Settings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3), 1.0,
                    cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    with torch.no_grad():
        g = O.preprocess(model.get_xyz.float(), model.get_scaling.float(), model.get_rotation.float(), None, st)
        op = model.get_opacity.float().view(-1)
    idx = torch.nonzero(g.valid).view(-1)
    rmin, rmax = g.rect_min[idx].long(), g.rect_max[idx].long()
    wx, wy = (rmax[:, 0] - rmin[:, 0]), (rmax[:, 1] - rmin[:, 1])
    cnt = wx * wy
    keep = cnt > 0
    idx, rmin, wx, cnt = idx[keep], rmin[keep], wx[keep], cnt[keep]
    N = int(cnt.sum())
    print(f"{key} view {view}: {H}x{W}, visible {idx.numel()}, reference pairs {N}")
    owner = torch.repeat_interleave(torch.arange(idx.numel()), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(N) - start[owner]
    if N > max_pairs:                                    # an unbiased sample of the pairs
        sel = torch.randperm(N)[:max_pairs]
        owner, local = owner[sel], local[sel]
    tx = rmin[owner, 0] + local % wx[owner]
    ty = rmin[owner, 1] + local // wx[owner]
    gi = idx[owner]
    px, py = g.xy[gi, 0], g.xy[gi, 1]
    A, B, C = g.conic[gi, 0], g.conic[gi, 1], g.conic[gi, 2]
    o = op[gi]
    lx = torch.arange(16).view(1, 1, 16).float()
    ly = torch.arange(16).view(1, 16, 1).float()
    layouts = {"16x4 strips (K6/K7)": (16, 4), "8x8 quadrants": (8, 8), "4x16 columns": (4, 16), "32x2 (n/a: tile is 16 wide)": None}
    tot = {k: [0, 0] for k in layouts if layouts[k]}
    pairs_hit = 0
    px_hit = 0
    n = owner.numel()
    for s in range(0, n, 200_000):
        e = slice(s, min(s + 200_000, n))
        dx = px[e].view(-1, 1, 1) - (tx[e].view(-1, 1, 1) * 16 + lx)
        dy = py[e].view(-1, 1, 1) - (ty[e].view(-1, 1, 1) * 16 + ly)
        inside = ((tx[e].view(-1, 1, 1) * 16 + lx) < W) & ((ty[e].view(-1, 1, 1) * 16 + ly) < H)
        power = -0.5 * (A[e].view(-1, 1, 1) * dx * dx + C[e].view(-1, 1, 1) * dy * dy) - B[e].view(-1, 1, 1) * dx * dy
        alpha = torch.clamp(o[e].view(-1, 1, 1) * torch.exp(power), max=0.99)
        hit = (power <= 0) & (alpha >= 1.0 / 255.0) & inside          # [n,16(y),16(x)]
        pairs_hit += int(hit.flatten(1).any(1).sum())
        px_hit += int(hit.sum())
        for name, wh in layouts.items():
            if not wh:
                continue
            bw, bh = wh
            blk = hit.view(-1, 16 // bh, bh, 16 // bw, bw).any(4).any(2)     # [n, by, bx]
            tot[name][0] += int(blk.sum())
    print(f"pairs sampled {n}; pairs with any passing pixel {pairs_hit} ({pairs_hit / n:.3f}); passing pixels per pair {px_hit / n:.1f}")
    base = tot["16x4 strips (K6/K7)"][0]
    for name, (blocks, _) in tot.items():
        print(f"{name:24s} sub-blocks evaluated per pair {blocks / n:.3f}  useful lanes {px_hit / (64.0 * blocks):.3f}  vs strips {blocks / base:.3f}")


if __name__ == "__main__":
    main()
