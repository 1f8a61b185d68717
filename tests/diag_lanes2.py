"""Diagnostic (CPU only, not a test): wave-level evaluations of the compositors if the 64 lanes of a 16x4 strip walked SEVERAL
entry streams at once -- lanes split into S groups, each group compositing its own compacted list of
the tile's entries onto its own sub-block of the strip, so that in one issue slot different groups work on different entries.

Counted exactly on whole tile lists (every (tile, Gaussian) pair of a sample of tiles, the reference's alpha >= 1/255 test on
every pixel centre): per tile and strip k, n_k = entries with a passing pixel in the strip (today's evaluations: one per
entry and strip, all 64 lanes), n_ks = entries with a passing pixel in sub-block s; a wave whose S lane groups walk their own
lists needs max_s n_ks slots for the strip (the groups execute ONE instruction stream: they are at the same strip k at any
time, but each at its own entry).  Reported: sum of slots / sum of n_k per layout, the useful-lane share, and the same with
the transmittance stop (an entry behind a pixel's last contributor is not evaluated for that pixel).

    python tests/diag_lanes2.py [scene] [view] [n_tiles] [scale_modifier]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "3d-gaussian-splat-attack_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gsplat_attack.scenes import make_scene  # noqa: E402
from oracle import oracle_r as O  # noqa: E402

# sub-blocks of a 16x4 strip: name -> (block width, block height); S = (16 / bw) * (4 / bh) lane groups
LAYOUTS = {"16x4 (today, 1 stream)": (16, 4), "16x2 rows (2 streams)": (16, 2), "8x4 halves (2 streams)": (8, 4),
           "8x2 (4 streams)": (8, 2), "4x4 (4 streams)": (4, 4), "16x1 rows (4 streams)": (16, 1)}


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "nyc-1M"
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n_tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    mod = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    torch.manual_seed(0)
    model, cams, _ = make_scene(key, n_views=max(view + 1, 1))
    cam = cams[view]
    H, W = cam.image_height, cam.image_width
    st = O.Settings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3), mod,
                    cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    with torch.no_grad():
        g = O.preprocess(model.get_xyz.float(), model.get_scaling.float(), model.get_rotation.float(), None, st)
        op = model.get_opacity.float().view(-1)
    idx = torch.nonzero(g.valid).view(-1)
    rmin, rmax = g.rect_min[idx].long(), g.rect_max[idx].long()
    depth = g.depth[idx]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = torch.randperm(gx * gy)[:n_tiles]
    tot = {k: [0, 0] for k in LAYOUTS}            # [slots without the T stop, slots with it]
    lanes_useful = [0, 0]
    pairs = 0
    for t in tiles.tolist():
        tx, ty = t % gx, t // gx
        m = (rmin[:, 0] <= tx) & (tx < rmax[:, 0]) & (rmin[:, 1] <= ty) & (ty < rmax[:, 1])
        sel = torch.nonzero(m).view(-1)
        if sel.numel() == 0:
            continue
        sel = sel[torch.argsort(depth[sel], stable=True)]
        gi = idx[sel]
        pairs += sel.numel()
        lx = (tx * 16 + torch.arange(16)).view(1, 1, 16).float()
        ly = (ty * 16 + torch.arange(16)).view(1, 16, 1).float()
        dx = g.xy[gi, 0].view(-1, 1, 1) - lx
        dy = g.xy[gi, 1].view(-1, 1, 1) - ly
        A, B, C = g.conic[gi, 0].view(-1, 1, 1), g.conic[gi, 1].view(-1, 1, 1), g.conic[gi, 2].view(-1, 1, 1)
        power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
        alpha = torch.clamp(op[gi].view(-1, 1, 1) * torch.exp(power), max=0.99)
        inside = (lx < W) & (ly < H)
        hit = (power <= 0) & (alpha >= 1.0 / 255.0) & inside                   # [n, 16(y), 16(x)], list order
        # transmittance stop: a pixel is done once T' < 1e-4 (that entry and all behind it are not blended)
        a_eff = torch.where(hit, alpha, torch.zeros_like(alpha))
        Tn = torch.cumprod(1.0 - a_eff, dim=0)
        live = torch.cat([torch.ones_like(Tn[:1], dtype=torch.bool), (Tn[:-1] >= 1e-4)], dim=0) & (Tn >= 1e-4)
        hit_T = hit & live
        for j, h in enumerate((hit, hit_T)):
            lanes_useful[j] += int(h.sum())
            for name, (bw, bh) in LAYOUTS.items():
                # [n, strip k (4), rows in strip (4), 16] -> blocks [n, 4, 4/bh, 16/bw]
                blk = h.view(-1, 4, 4 // bh, bh, 16 // bw, bw).any(5).any(3)
                n_ks = blk.sum(0).view(4, -1)                                 # entries per (strip, sub-block)
                tot[name][j] += int(n_ks.max(dim=1).values.sum())
    print(f"{key} view {view} scale_modifier {mod}: {len(tiles)} tiles, {pairs} (tile, Gaussian) pairs of the reference's rects")
    for j, label in enumerate(("alpha test only", "alpha test + transmittance stop")):
        base = tot["16x4 (today, 1 stream)"][j]
        print(f"-- {label}: {base / max(pairs, 1):.3f} strip evaluations per pair today, useful lanes {lanes_useful[j] / (64.0 * base):.3f}")
        for name in LAYOUTS:
            s = tot[name][j]
            print(f"   {name:26s} wave-level slots {s:9d}  vs today {s / base:.3f}  useful lanes {lanes_useful[j] / (64.0 * s):.3f}")


if __name__ == "__main__":
    main()
