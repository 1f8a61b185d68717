"""Shared helpers for the parity tests (oracle side + HIP side)."""
import math

import torch

from oracle import oracle_r as O


def settings_for(cam, bg, sh_degree=3, scale_modifier=1.0, cls=O.Settings, device=None):
    def dev(t):
        return t if device is None else t.to(device)
    return cls(int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
               dev(bg), scale_modifier, dev(cam.world_view_transform), dev(cam.full_proj_transform), sh_degree,
               dev(cam.camera_center), False, False)


def model_inputs(model, with_objs=True):
    """Activated attributes exactly as render() hands them to the rasteriser."""
    d = dict(means3D=model.get_xyz.detach(), shs=model.get_features.detach(), opacities=model.get_opacity.detach(),
             scales=model.get_scaling.detach(), rotations=model.get_rotation.detach())
    if with_objs:
        d["sh_objs"] = model.get_objects.detach()
    return d


def grad_error(g, ref, elem_tol=5e-3, yard=None):
    """(normwise max error / max|ref|,  fraction of significant elements whose relative error exceeds
    elem_tol).  Significant = within 3 decades of the group's largest magnitude.
    yard: the same gradient from the float32 oracle.  An element then counts as off only if its error also exceeds TWICE
    the float32 oracle's own error on that element (the per-element form of the float32 yardstick: a component that is
    the residue of a cancellation is as wrong in the oracle's float32 run as in any float32 implementation)."""
    g = g.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    scale = ref.abs().max().item()
    if scale == 0.0:
        return g.abs().max().item(), 0.0
    err = (g - ref).abs()
    norm = (err.max() / scale).item()
    big = ref.abs() > 1e-3 * scale
    off = (err / ref.abs().clamp_min(1e-300)) > elem_tol
    if yard is not None:
        off = off & (err > 2.0 * (yard.detach().double().cpu().reshape(-1) - ref).abs())
    frac = off[big].double().mean().item() if big.any() else 0.0
    return norm, frac


# ---------------------------------------------------------------------------------------------------------------------
# The float32 yardstick (round 4).  oracle-R flags as "fragile" the pixels where a threshold test (alpha >= 1/255,
# T' < 1e-4, power > 0, an integer rect / radius decision) sits within float32 rounding of its edge.  Rounds 1-3 held those
# pixels to 1e-2 only.  Here every pixel -- fragile or not -- is measured against what float32 arithmetic itself does
# to the published algorithm: oracle-R is run in float64 AND in float32 on the same inputs, and the implementation's value
# must satisfy, per pixel (max over channels),
#     |hip - r64| <= max(tol, 2 |r32 - r64|)      (A: no further from the float64 result than float32 rounding puts the
#                                                     oracle's own float32 run, with a factor 2)
#  or |hip - r32| <= tol                           (B: it IS the float32 outcome)
# Reported: the share of pixels that is fragile, the share that needs clause B, and the share of FRAGILE pixels on neither.
# ---------------------------------------------------------------------------------------------------------------------
def pixel_yardstick(hip, r64, r32, fragile, mask=None, tol=1e-4):
    """hip, r64, r32: [C,H,W] images; fragile, mask: [H,W] bool (mask = pixels compared; None: all).
    -> dict(n, fragile, need_b, neither, neither_solid, worst_neither, worst_solid, e64 map, ok map)."""
    hip, r64, r32 = hip.detach().double().cpu(), r64.detach().double().cpu(), r32.detach().double().cpu()
    e64 = (hip - r64).abs().amax(dim=0)
    e32 = (hip - r32).abs().amax(dim=0)
    d = (r32 - r64).abs().amax(dim=0)
    ok_a = e64 <= torch.clamp_min(2.0 * d, tol)
    ok_b = e32 <= tol
    m = torch.ones_like(fragile) if mask is None else mask
    frag = fragile & m
    solid = m & ~fragile
    neither = m & ~ok_a & ~ok_b
    n = max(int(m.sum()), 1)
    nf = max(int(frag.sum()), 1)
    return dict(n=int(m.sum()), fragile=int(frag.sum()) / n, need_b=int((m & ~ok_a & ok_b).sum()) / n,
                neither=int((neither & frag).sum()) / nf, neither_px=int((neither & frag).sum()),
                neither_solid=int((neither & solid).sum()),
                worst_neither=float(torch.minimum(e64, e32)[neither].max()) if bool(neither.any()) else 0.0,
                worst_solid=float(e64[solid].max()) if bool(solid.any()) else 0.0,
                worst_any=float(e64[m].max()) if bool(m.any()) else 0.0,
                f32_vs_f64=float(d[m].max()) if bool(m.any()) else 0.0, e64=e64, ok=ok_a | ok_b)


def yardstick_line(tag, y):
    return (f"[{tag}] px {y['n']}, fragile {y['fragile']:.4f}, need the float32 outcome (clause B) {y['need_b']:.5f}, "
            f"fragile px on neither {y['neither']:.5f} ({y['neither_px']} px, worst {y['worst_neither']:.2e}), solid px on "
            f"neither {y['neither_solid']}, solid err {y['worst_solid']:.2e}, worst err {y['worst_any']:.2e}, "
            f"oracle f32 vs f64 {y['f32_vs_f64']:.2e}")


def check_tile_lists_depth_order(D, img):
    """Every tile's span of the sorted pair list is in (float32 depth key, Gaussian index) order -- strictly: what a stable
    depth sort of storage-ordered Gaussians in front of a stable tile sort gives.  Returns (ranges, Gaussian ids)."""
    dev = img.device
    ranges = D.export_state(img, "ranges").view(-1, 2).long()
    pairs = D.export_state(img, "pair_rank").long()
    assert int((pairs >> 28).max() if pairs.numel() else 0) <= 15
    g = pairs & ((1 << 28) - 1)
    depth = D.export_state(img, "G").view(-1, 12)[:, 9]
    key = depth[g]
    assert bool((key > 0).all())                              # every listed Gaussian has its record (and depth) written
    lens = ranges[:, 1] - ranges[:, 0]
    assert int(lens.min()) >= 0 and int(lens.sum()) <= pairs.numel()
    nz = lens > 0
    starts = torch.zeros(pairs.numel() + 1, dtype=torch.bool, device=dev)
    starts[ranges[nz, 0]] = True
    covered = torch.zeros(pairs.numel() + 1, dtype=torch.long, device=dev)
    covered.index_add_(0, ranges[nz, 0], torch.ones(int(nz.sum()), dtype=torch.long, device=dev))
    covered.index_add_(0, ranges[nz, 1], -torch.ones(int(nz.sum()), dtype=torch.long, device=dev))
    inside = torch.cumsum(covered, 0)[:-1] > 0
    if pairs.numel() > 1:
        inc = (key[1:] > key[:-1]) | ((key[1:] == key[:-1]) & (g[1:] > g[:-1]))
        must = inside[1:] & inside[:-1] & ~starts[1:-1]
        assert bool(inc[must].all()), f"{int((~inc[must]).sum())} list neighbours out of (depth, index) order"
    return ranges, g
