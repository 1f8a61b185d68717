"""BASELINE.json's full-size configurations against oracle-R, on the kernels and sizes the benchmark times.

oracle-R cannot composite 2 M pixels x 1 M Gaussians, but it can run its per-Gaussian stages over all P in float64 and
composite a few TILE WINDOWS (oracle_r.rasterize(tile_windows=...)).  The HIP path renders the whole image; dL/dC is
zero outside the windows, so every attribute gradient is owned by the windows' pixels and comparable to the oracle's:
RGB <= 1e-4 on the windows, all five gradient groups <= 1e-3 relative (BASELINE.json tolerances).  Windows: the tile
with the longest list, a dense one, one on the ragged image border, one mid-image.  At these sizes the library takes the
launch shapes the headline number is measured on (>= 4096 tiles: k_render_fwd<.,2>, k_render_bwd<.,4,.>, segmented
long lists), which the small oracle cases never reach.
"""
import math

import pytest
import os

import torch

from oracle import oracle_r as O
from util import settings_for, grad_error, pixel_yardstick, yardstick_line

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4
GRAD_TOL = 1e-3
RAW = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")
# share of the FRAGILE window pixels that may sit on neither clause of the float32 yardstick (util.pixel_yardstick)
NEITHER_CAP = 0.005           # a constant (round 6): a tolerance that the environment could loosen is not a tolerance
_R32 = {}          # id(float64 RenderOut) -> the float32 oracle's RenderOut of the same windows
_ALLPX = {}        # id(float64 RenderOut) -> the all-pixel loss (fragile pixels INCLUDED) and both oracles' gradients of it


def _hip():
    import diff_gaussian_rasterization as D
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    D._load()
    return D


def pick_windows(ranges, gx, gy, half=1, extra=0, seed=0):
    """Tile windows (tx0, ty0, tx1, ty1) from the HIP forward's per-tile list lengths: around the longest list, around a
    dense tile (99th percentile), the bottom-right border, the bottom border under the densest column, mid-image, plus
    `extra` seeded random 2x2 windows among the non-empty tiles."""
    lens = (ranges[:, 1] - ranges[:, 0]).view(gy, gx).cpu()

    def around(ty, tx):
        return (max(tx - half, 0), max(ty - half, 0), min(tx + half + 1, gx), min(ty + half + 1, gy))
    flat = lens.flatten()
    t_long = int(flat.argmax())
    order = torch.argsort(flat)
    t_dense = int(order[int(0.99 * (flat.numel() - 1))])
    col = int(lens.sum(dim=0).argmax())
    wins = [around(t_long // gx, t_long % gx), around(t_dense // gx, t_dense % gx),
            (gx - 2, gy - 2, gx, gy), (max(col - 1, 0), gy - 1, min(col + 1, gx), gy),
            around(gy // 2, gx // 2)]
    if extra:
        g = torch.Generator().manual_seed(seed)
        nz = torch.nonzero(flat > 0).flatten()
        for t in nz[torch.randperm(nz.numel(), generator=g)[:extra]].tolist():
            ty, tx = t // gx, t % gx
            wins.append((tx, ty, min(tx + 2, gx), min(ty + 2, gy)))
    return wins, int(flat.max())


def window_mask(wins, H, W):
    m = torch.zeros(H, W, dtype=torch.bool)
    for (x0, y0, x1, y1) in wins:
        m[16 * y0:16 * y1, 16 * x0:16 * x1] = True
    return m


def oracle_raw(key, cam_i, bg, gc, wins, scale_kw=None, objects=False, go=None, keys=None, scale=1.0, n_views=None):
    """oracle-R float64 on the CPU twin of the scene, autograd down to the RAW parameters through the getters.
    keys = (depth keys, radii) exported from the HIP forward: the oracle composites in the order of those float32 keys
    once they are within a few ulps of its own float64 depth (oracle_r.check_depth_keys)."""
    from gsplat_attack.scenes import make_scene
    # (the ring cameras depend on the ring's size: the twin must be built with the same number of views)
    ref, rcams, _ = make_scene(key, device="cpu", n_views=n_views or cam_i + 1, **(scale_kw or {}))
    st = settings_for(rcams[cam_i], bg, 3, scale)
    depth_key = None
    if keys is not None:
        depth_key = keys[0]
        O.check_depth_keys(depth_key, ref.get_xyz, st, keys[1])
    ro = O.rasterize(ref.get_xyz, None, ref.get_opacity, st, shs=ref.get_features,
                     sh_objs=ref.get_objects if objects else None, scales=ref.get_scaling, rotations=ref.get_rotation,
                     tile_windows=wins, depth_key=depth_key)
    # the same oracle in float32 on the same windows: the yardstick of compare() (what float32 arithmetic itself does).
    # It is differentiated too (round 5): compare_all_pixels() measures the backward on EVERY window pixel, the fragile
    # ones included, against what float32 costs the oracle's own gradients.
    _R32.clear()
    _ALLPX.clear()
    r32 = O.rasterize(ref.get_xyz, None, ref.get_opacity, st, shs=ref.get_features,
                      sh_objs=ref.get_objects if objects else None, scales=ref.get_scaling, rotations=ref.get_rotation,
                      tile_windows=wins, depth_key=depth_key, dtype=torch.float32)
    _R32[id(ro)] = r32
    params = ref.named_parameters()
    wpx = ro.window_px if ro.window_px is not None else torch.ones_like(ro.fragile_px)
    gc_all = gc * wpx.to(gc.dtype)
    go_all = None if go is None else go * wpx.to(go.dtype)

    names = [n for n, p in params.items() if p.requires_grad]

    def grads_of(r, gcm, gom):
        """d/d(raw parameters) of sum(colour * gcm) [+ sum(objects * gom)] from oracle run r (its graph is kept)."""
        loss = (r.color * gcm.to(r.color.dtype)).sum()
        if gom is not None:
            loss = loss + (r.objects * gom.to(r.color.dtype)).sum()
        gs = torch.autograd.grad(loss, [params[n] for n in names], retain_graph=True, allow_unused=True)
        return {n: g.detach().double() for n, g in zip(names, gs) if g is not None}

    def both(keep=None):
        """(float64, float32) oracle gradients of the loss over the window pixels in `keep` (None: all of them)."""
        k = wpx if keep is None else (wpx & keep)
        gcm = gc_all * k.to(gc_all.dtype)
        gom = None if go_all is None else go_all * k.to(go_all.dtype)
        return grads_of(ro, gcm, gom), grads_of(r32, gcm, gom), gcm, gom
    _ALLPX[id(ro)] = dict(both=both, wpx=wpx)
    # the loss of the first comparison ignores the pixels oracle-R flags as fragile (a float32 threshold test may flip there)
    gc, go = O.solid_grads(ro, gc, go)
    grads = grads_of(ro, gc, go)
    return ro, grads, gc, go


def hip_raw(model, cam, bg, gc, flags=0, fused=True, objects=False, go=None, color_only=False, scale=1.0):
    from gsplat_attack.renderer import PipelineParams, render
    D = _hip()
    model.zero_grad()
    frozen = []
    if color_only:
        for n in ("_xyz", "_scaling", "_rotation", "_opacity", "_objects_dc"):
            p = getattr(model, n)
            if p.requires_grad:
                p.requires_grad_(False)
                frozen.append(p)
    try:
        D.set_flags(flags)
        out = render(cam, model, PipelineParams(fused_activations=fused, skip_objects=not objects,
                                                viewspace_grad=not color_only), bg, scale)
        loss = (out["render"] * gc).sum()
        if go is not None:
            loss = loss + (out["render_object"] * go).sum()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        D.set_flags(0)
        for p in frozen:
            p.requires_grad_(True)
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters().items() if p.grad is not None}
    return out, grads


def _note(line):
    """Observed figures of the parity runs (fragile shares, errors): printed, and appended to gpurun_out/parity_notes.txt
    when that directory exists (DESIGN.md quotes them)."""
    print(line)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_notes.txt"), "a") as f:
            f.write(line + "\n")


def compare(out, grads, ro, rgrads, m, names=RAW, frag_frac=0.08, objects=False, tag=""):
    """frag_frac: cap on the share of window pixels the oracle may flag fragile -- at most twice what this binary shows on
    the case (observed in round 3: cfg 3 0.036, classic surface 0.026, dense point 0.039, cfg 5
    0.078, cfg 2 0.169 -- long lists of faint splats --, cfg 3 under the oracle's own depth order 0.186); the observed
    share is printed and recorded."""
    color = out["render"].detach().cpu().double()
    err = (color - ro.color.detach()).abs().max(dim=0).values
    solid = m & ~ro.fragile_px
    frag = m & ro.fragile_px
    share = frag.sum().item() / max(m.sum().item(), 1)
    _note(f"[{tag or 'windows'}] fragile share {share:.4f} (cap {frag_frac}), solid RGB err {err[solid].max().item():.2e}"
          + (f", fragile RGB err {err[frag].max().item():.2e}" if frag.any() else ""))
    assert share <= frag_frac, f"{int(frag.sum())} fragile pixels of {int(m.sum())}"
    r32 = _R32[id(ro)]
    # Solid pixels: 1e-4 (BASELINE) at every size, 4K included.  (With the published float32 pixel centre that did not hold
    # at 4K -- one ulp of a coordinate beyond 2048 is 2.4e-4 px: over 2 % of S-airport-4K's tiles the float32 ORACLE is above
    # 1e-4 on 85 of 167 608 solid pixels, max 1.95e-4, and this implementation was on 40, max 1.79e-4.  Since the compositors
    # measure distances from a tile-relative centre projected in double -- csrc/gsr_math.h Splat::pxd -- the same sweep reads
    # max 1.2e-6: profiles/r05_fullsize_sweep.txt.)
    assert err[solid].max().item() <= RGB_TOL, f"RGB max abs err {err[solid].max().item():.3e} on the windows"
    # EVERY window pixel, the fragile ones included, against the float32 yardstick (round 4): within
    # max(1e-4, 2 |r32 - r64|) of the float64 oracle, or on the float32 oracle's own outcome
    y = pixel_yardstick(color, ro.color, r32.color, ro.fragile_px, mask=m, tol=RGB_TOL)
    _note(yardstick_line(f"yardstick {tag or 'windows'}", y))
    _ALLPX[id(ro)]["ok"] = y["ok"] | ~m          # compare_all_pixels: the pixels whose VALUE is a float32 / float64 outcome
    assert y["neither_solid"] == 0
    assert y["neither_px"] <= max(5, NEITHER_CAP * y["fragile"] * y["n"]), yardstick_line(tag, y)
    assert err[m].max().item() <= 1e-2                     # backstop only
    if objects:
        eo = (out["render_object"].detach().cpu().double() - ro.objects.detach()).abs().max(dim=0).values
        assert eo[solid].max().item() <= 3 * RGB_TOL
    bad_r = (out["radii"].cpu() != ro.radii) & ~ro.fragile_gauss
    assert int(bad_r.sum()) == 0, f"{int(bad_r.sum())} radii differ on non-fragile Gaussians"
    rep = {}
    for n in names:
        norm, frac = grad_error(grads[n], rgrads[n], elem_tol=5 * GRAD_TOL)
        rep[n] = (norm, frac)
        assert rgrads[n].abs().max().item() > 0, n
        assert norm <= GRAD_TOL, f"grad {n}: normwise rel err {norm:.3e}"
        assert frac <= 3e-3, f"grad {n}: {frac:.2e} of the significant elements off by more than {5 * GRAD_TOL}"
    return rep


def compare_all_pixels(model, cam, bg, ro, names=RAW, tag="", **hip_kw):
    """Round 5: the backward on ALL window pixels, the fragile ones included.  compare() above
    differentiates a loss from which the fragile pixels are removed on both sides -- 3.6 % ... 19 % of the compared pixels
    never contributed a checked gradient.  Here the loss is over every window pixel; oracle-R is differentiated in float64
    AND in float32 on it, and per attribute group the implementation must satisfy
        |g_hip - g64|_inf <= max(1e-3 |g64|_inf, 2 |g32 - g64|_inf)
    (no further from the float64 gradient than BASELINE's tolerance or twice what float32 arithmetic costs the oracle's
    own gradient), plus the element criterion of util.grad_error with the float32 oracle as its yardstick.

    A threshold test that flips is a DISCONTINUITY of the gradient, not a rounding error: an entry with alpha ~ 1/255 that
    one float32 arithmetic blends and another skips changes dL/dopacity of that Gaussian by G T (c - C_behind) . g -- not
    scaled by alpha.  The float32 oracle flips on its own pixels, the implementation on its own; where the two sets
    differ by a pixel the norm-wise comparison sees that pixel, not the arithmetic.  Such pixels are known exactly: they
    are the ones on NEITHER clause of compare()'s image yardstick (the implementation's colour there is neither within
    twice the float32 oracle's deviation from float64 nor the float32 oracle's own value; compare() caps their number at
    max(5, 0.5 % of the fragile pixels)).  The comparison is run (a) on every window pixel -- logged as 'raw', and it
    must hold as it stands when no pixel is on neither clause -- and, when some are, (b) with exactly those pixels
    removed from the loss on all three sides, which must hold; the log shows how many pixels that was and what (a) gave,
    so that a failure of (a) is attributed to those pixels by measurement, not by assumption."""
    a = _ALLPX[id(ro)]
    dev = next(iter(model.named_parameters().values())).device

    def one(keep, label):
        g64s, g32s, gcm, gom = a["both"](keep)
        kw = dict(hip_kw, objects=True, go=gom.to(dev)) if gom is not None else hip_kw
        _, grads = hip_raw(model, cam, bg.to(dev), gcm.to(dev), **kw)
        rep, bad = {}, []
        for n in names:
            g64, g32 = g64s[n], g32s[n]
            s = g64.abs().max().item()
            assert s > 0, n
            e_hip = (grads[n].double() - g64).abs().max().item() / s
            e_32 = (g32 - g64).abs().max().item() / s
            _, frac = grad_error(grads[n], g64, elem_tol=5 * GRAD_TOL)
            _, frac_y = grad_error(grads[n], g64, elem_tol=5 * GRAD_TOL, yard=g32)
            _, frac32 = grad_error(g32, g64, elem_tol=5 * GRAD_TOL)
            rep[n] = (e_hip, e_32, frac, frac32, frac_y)
            _note(f"[all-pixel grads {tag}{label}] {n}: HIP {e_hip:.2e}, float32 oracle {e_32:.2e} (normwise, of |g64|_inf); "
                  f"elements off by > {5 * GRAD_TOL}: HIP {frac:.2e}, float32 oracle {frac32:.2e}, HIP beyond twice the "
                  f"float32 oracle {frac_y:.2e}")
            if e_hip > max(GRAD_TOL, 2 * e_32):
                bad.append(f"{n}: normwise {e_hip:.3e} > max({GRAD_TOL}, 2 x {e_32:.3e})")
            if not (frac <= max(3e-3, 2 * frac32) or frac_y <= 3e-3):
                bad.append(f"{n}: {frac:.2e} of the significant elements off (float32 oracle {frac32:.2e}, beyond its "
                           f"yardstick {frac_y:.2e})")
        return rep, bad
    third = a["wpx"] & ~a["ok"]                 # window pixels whose colour is a third outcome (compare() has capped them)
    n_third, n_px = int(third.sum()), int(a["wpx"].sum())
    rep, bad = one(None, ", raw" if n_third else "")
    if n_third:
        _note(f"[all-pixel grads {tag}] {n_third} of {n_px} window pixels are on neither clause of the image yardstick; raw "
              f"comparison: {'holds' if not bad else 'misses: ' + '; '.join(bad)}")
        rep, bad = one(a["ok"], f", without the {n_third} third-outcome px")
    assert not bad, f"all-pixel backward [{tag}] ({n_third} third-outcome pixels removed): " + "; ".join(bad)
    return rep


def _scene_on_gpu(key, n_views):
    from gsplat_attack.scenes import make_scene
    _hip()
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene(key, device=dev, n_views=n_views)
    return dev, model, cams


def _windows_for(D, model, cam, bg, scale=1.0, extra=0):
    from gsplat_attack.renderer import PipelineParams, render
    H, W = cam.image_height, cam.image_width
    gx, gy = (W + 15) // 16, (H + 15) // 16
    out = render(cam, model, PipelineParams(skip_objects=True), bg, scale)
    img = out["render"]
    ranges = D.export_state(img, "ranges").view(-1, 2).long()
    wins, longest = pick_windows(ranges, gx, gy, extra=extra, seed=gx * gy)
    radii = out["radii"].cpu()
    # depth keys from a forward over the full 3-sigma rects: every Gaussian with radius > 0 then has its record written
    with D.extra_flags(D.FLAG_NO_CULL):
        full = render(cam, model, PipelineParams(skip_objects=True), bg, scale)
    depth = D.export_state(full["render"], "G").view(-1, 12)[:, 9].cpu()
    assert torch.equal(full["radii"].cpu(), radii)
    del full
    depth = torch.where(radii > 0, depth, torch.zeros_like(depth))       # records of culled Gaussians are not written
    return wins, longest, gx, gy, (depth, radii)


def test_cfg3_nyc_1m_1080p_all_gradients_vs_windowed_oracle():
    """BASELINE config 3/4 shape: S-nyc-1M, 1920x1080 (8160 tiles => k_render_fwd<.,2>, k_render_bwd<.,4,true>), all
    five attribute groups, fused raw-parameter path -- the exact kernels and sizes bench.py times."""
    D = _hip()
    dev, model, cams = _scene_on_gpu("nyc-1M", 3)
    cam = cams[2]
    bg = torch.tensor([0.1, 0.2, 0.3])
    wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev), extra=10)   # 5 chosen + 10 random windows
    assert gx * gy >= 4096 and longest > 256, (gx * gy, longest)
    m = window_mask(wins, 1080, 1920)
    gc = torch.randn(3, 1080, 1920, generator=torch.Generator().manual_seed(99)) * m
    ro, rgrads, gc, _ = oracle_raw("nyc-1M", 2, bg, gc, wins, keys=keys)
    out, grads = hip_raw(model, cam, bg.to(dev), gc.to(dev))
    rep = compare(out, grads, ro, rgrads, m, tag="cfg3 nyc-1M fused")
    print("cfg3 windows", wins, "longest list", longest, rep)
    print("cfg3 all pixels", compare_all_pixels(model, cam, bg, ro, tag="cfg3 nyc-1M fused"))
    compare_all_pixels(model, cam, bg, ro, tag="cfg3 nyc-1M fused, whole lists", flags=D.FLAG_NO_SEGMENTS)
    # the same view with long lists NOT split over waves and the other tile splits: same numbers within rounding
    for flags in (D.FLAG_NO_SEGMENTS, D.FLAG_FWD_SHARED, D.flag_fwd_split(4) | D.flag_bwd_split(2), D.flag_fwd_split(1) | D.flag_tile_map(0)):
        out2, grads2 = hip_raw(model, cam, bg.to(dev), gc.to(dev), flags=flags)
        assert (out2["render"] - out["render"]).abs().max().item() <= 2e-6, flags
        compare(out2, grads2, ro, rgrads, m, tag=f"cfg3 nyc-1M flags {flags:#x}")


def test_cfg3_windows_with_the_oracles_own_depth_order():
    """The same comparison WITHOUT handing the HIP path's depth keys to the oracle: the oracle
    sorts on its own float32(float64 depth) keys and flags, per pixel, the list neighbours whose keys may sort the other
    way round in another float32 arithmetic.  More pixels are fragile; the solid ones must agree all the same."""
    D = _hip()
    dev, model, cams = _scene_on_gpu("nyc-1M", 3)
    cam = cams[2]
    bg = torch.tensor([0.1, 0.2, 0.3])
    wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev))
    wins = wins[1:4]                                       # a 99th-percentile tile window, the ragged corner, the border
    m = window_mask(wins, 1080, 1920)
    gc = torch.randn(3, 1080, 1920, generator=torch.Generator().manual_seed(17)) * m
    ro, rgrads, gc, _ = oracle_raw("nyc-1M", 2, bg, gc, wins, keys=None)
    out, grads = hip_raw(model, cam, bg.to(dev), gc.to(dev))
    rep = compare(out, grads, ro, rgrads, m, frag_frac=0.38, tag="cfg3 nyc-1M, oracle's own depth order")
    # (no all-pixel pass here: with the oracle's own depth order the fragile pixels include every pixel two near-tied
    # splats share -- the float32 oracle sorts them its way, the implementation its own)
    print("own-order windows", wins, rep)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg5"])
def test_cfg2_and_cfg5_windows_with_the_oracles_own_depth_order(cfg):
    """Configs 2 and 5 once each WITHOUT the HIP path's depth keys: the oracle sorts on its own
    float32(float64 depth) keys, flags the pixels whose list neighbours may swap in another float32 arithmetic, and the
    solid pixels -- image and the gradients of a loss over them -- must agree all the same.  Fewer windows than the
    keyed tests (the own-order oracle flags more pixels fragile; what is checked is that agreement does not hinge on
    borrowed keys)."""
    D = _hip()
    if cfg == "cfg2":
        key, view, H, W = "hydrant-full", 0, 800, 800
        bg = torch.tensor([0.0, 0.0, 0.0])
        dev, model, cams = _scene_on_gpu(key, 1)
    else:
        key, view, H, W = "airport-4K", 1, 2160, 3840
        bg = torch.tensor([0.0, 0.0, 0.0])
        dev, model, cams = _scene_on_gpu(key, 2)
    cam = cams[view]
    wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev))
    wins = wins[1:4]
    m = window_mask(wins, H, W)
    gc = torch.randn(3, H, W, generator=torch.Generator().manual_seed(23)) * m
    ro, rgrads, gc, _ = oracle_raw(key, view, bg, gc, wins, keys=None, n_views=view + 1)
    out, grads = hip_raw(model, cam, bg.to(dev), gc.to(dev))
    # (observed fragile shares: cfg 2 0.59 -- a dense blob: most pixels see two near-tied splats somewhere in their list --,
    # cfg 5 see profiles/r06_parity_notes.txt; the solid AND the fragile pixels agree to 5e-7 on cfg 2 all the same)
    rep = compare(out, grads, ro, rgrads, m, frag_frac=0.75 if cfg == "cfg2" else 0.5, tag=f"{cfg} {key}, oracle's own depth order")
    _note(f"[own depth order] {cfg} {key}: windows {wins}, longest list {longest}, {rep}")
    print(cfg, "own-order windows", wins, rep)


def test_dense_10m_pairs_vs_windowed_oracle():
    """The benchmark's second data point: S-nyc-1M with every splat scaled by 2.43 (scale_modifier, render()'s own
    argument) so that a view emits ~10 M (tile, Gaussian) pairs -- lists of several thousand entries, many segments per
    tile -- against the windowed oracle."""
    D = _hip()
    dev, model, cams = _scene_on_gpu("nyc-1M", 2)
    cam = cams[1]
    bg = torch.tensor([0.0, 0.1, 0.0])
    scale = 2.43
    wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev), scale)
    assert longest > 2000, longest
    wins = wins[:3]                                                       # three 3x3 windows of ~4000-entry lists
    m = window_mask(wins, 1080, 1920)
    gc = torch.randn(3, 1080, 1920, generator=torch.Generator().manual_seed(31)) * m
    ro, rgrads, gc, _ = oracle_raw("nyc-1M", 1, bg, gc, wins, keys=keys, scale=scale)
    assert ro.num_rendered > 20000
    out, grads = hip_raw(model, cam, bg.to(dev), gc.to(dev), scale=scale)
    rep = compare(out, grads, ro, rgrads, m, frag_frac=0.08, tag="dense 10M pairs")
    print("dense all pixels", compare_all_pixels(model, cam, bg, ro, tag="dense 10M pairs", scale=scale))
    print("dense windows", wins, "longest list", longest, rep)


def test_cfg3_classic_activated_surface_vs_windowed_oracle():
    """The surface the reference's own render() hits (getters + GaussianRasterizer.forward), at full size."""
    D = _hip()
    dev, model, cams = _scene_on_gpu("nyc-1M", 1)
    cam = cams[0]
    bg = torch.zeros(3)
    wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev))
    m = window_mask(wins, 1080, 1920)
    gc = torch.randn(3, 1080, 1920, generator=torch.Generator().manual_seed(7)) * m
    ro, rgrads, gc, _ = oracle_raw("nyc-1M", 0, bg, gc, wins, keys=keys)
    out, grads = hip_raw(model, cam, bg.to(dev), gc.to(dev), fused=False)
    compare(out, grads, ro, rgrads, m, tag="cfg3 classic surface")
    print("classic all pixels", compare_all_pixels(model, cam, bg, ro, tag="cfg3 classic surface", fused=False))


def test_cfg2_hydrant_full_800px_sh_gradients_vs_windowed_oracle():
    """BASELINE config 2: 300 k Gaussians at 800x800 (2500 tiles => the finer tile split), gradients on the SH
    coefficients only (colour-only backward kernels), then the full backward on the same windows."""
    D = _hip()
    dev, model, cams = _scene_on_gpu("hydrant-full", 1)
    cam = cams[0]
    bg = torch.tensor([0.0, 0.0, 0.0])
    wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev))
    assert gx * gy < 4096
    m = window_mask(wins, 800, 800)
    gc = torch.randn(3, 800, 800, generator=torch.Generator().manual_seed(2)) * m
    ro, rgrads, gc, _ = oracle_raw("hydrant-full", 0, bg, gc, wins, keys=keys)
    out, grads = hip_raw(model, cam, bg.to(dev), gc.to(dev), color_only=True)
    assert set(grads) == {"f_dc", "f_rest"}
    compare(out, grads, ro, rgrads, m, names=("f_dc", "f_rest"), frag_frac=0.25, tag="cfg2 hydrant-full SH only")
    out, grads = hip_raw(model, cam, bg.to(dev), gc.to(dev))
    compare(out, grads, ro, rgrads, m, frag_frac=0.25, tag="cfg2 hydrant-full all")
    compare_all_pixels(model, cam, bg, ro, names=("f_dc", "f_rest"), tag="cfg2 hydrant-full SH only", color_only=True)
    print("cfg2 all pixels", compare_all_pixels(model, cam, bg, ro, tag="cfg2 hydrant-full all"))


def test_cfg5_airport_4k_full_backward_vs_windowed_oracle():
    """BASELINE config 5: 2 M Gaussians at 3840x2160 (32400 tiles), all gradients, object channels composited and
    differentiated."""
    D = _hip()
    dev, model, cams = _scene_on_gpu("airport-4K", 1)
    cam = cams[0]
    bg = torch.tensor([0.2, 0.1, 0.0])
    wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev))
    m = window_mask(wins, 2160, 3840)
    g = torch.Generator().manual_seed(5)
    gc = torch.randn(3, 2160, 3840, generator=g) * m
    go = torch.randn(16, 2160, 3840, generator=g) * 0.2 * m
    ro, rgrads, gc, go = oracle_raw("airport-4K", 0, bg, gc, wins, keys=keys, objects=True, go=go)
    out, grads = hip_raw(model, cam, bg.to(dev), gc.to(dev), objects=True, go=go.to(dev))
    rep = compare(out, grads, ro, rgrads, m, names=RAW + ("objects_dc",), objects=True, frag_frac=0.155, tag="cfg5 airport-4K objects")
    print("cfg5 windows", wins, "longest list", longest, rep)
    print("cfg5 all pixels", compare_all_pixels(model, cam, bg, ro, names=RAW + ("objects_dc",), tag="cfg5 airport-4K objects"))


def test_cfg4_views_of_a_batch_accumulate_at_full_size():
    """BASELINE config 4's single-GPU logic at full size: three ring views of S-nyc-1M rendered and differentiated one
    after the other (all five attribute groups, three HIP streams), .grad accumulating across them -- what each rank
    of the view-sharded loop does before the all-reduce -- against the SUM of the windowed oracle's per-view gradients."""
    from gsplat_attack.renderer import PipelineParams, render
    from gsplat_attack.streams import StreamRing
    D = _hip()
    dev, model, cams = _scene_on_gpu("nyc-1M", 6)
    bg = torch.tensor([0.05, 0.0, 0.1])
    views = [cams[1], cams[3], cams[5]]
    per_view = []
    for vi, cam in zip((1, 3, 5), views):
        wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev))
        wins = wins[:3]
        m = window_mask(wins, 1080, 1920)
        gc = torch.randn(3, 1080, 1920, generator=torch.Generator().manual_seed(40 + vi)) * m
        ro, rgrads, gc, _ = oracle_raw("nyc-1M", vi, bg, gc, wins, keys=keys, n_views=6)
        per_view.append((gc.to(dev), rgrads))
    model.zero_grad()
    ring = StreamRing(3, dev)
    for cam, (gcd, _) in zip(views, per_view):
        with ring.next():
            render(cam, model, PipelineParams(skip_objects=True), bg.to(dev))["render"].backward(gcd)
    ring.join()
    torch.cuda.synchronize()
    for n in RAW:
        want = sum(rg[n] for _, rg in per_view)
        norm, frac = grad_error(getattr_grad(model, n), want, elem_tol=5 * GRAD_TOL)
        assert norm <= GRAD_TOL and frac <= 3e-3, (n, norm, frac)


def test_cfg4_batch_of_four_views_through_one_launch_chain_vs_windowed_oracle():
    """BASELINE config 4's rasterisation as ONE batch (gsr_forward_raw_batch / gsr_backward_raw_batch_into): four ring views
    of S-nyc-1M at 1080p rendered by one launch chain -- every view's image against the windowed oracle on its own windows,
    and the batch's summed gradients (written once per Gaussian by k_pre_bwd_batch) against the SUM of the oracle's per-view
    gradients, all five attribute groups."""
    import diff_gaussian_rasterization as Dm
    from gsplat_attack.renderer import PipelineParams, render_batch
    D = _hip()
    dev, model, cams = _scene_on_gpu("nyc-1M", 6)
    bg = torch.tensor([0.05, 0.0, 0.1])
    idx = (0, 2, 3, 5)
    views = [cams[i] for i in idx]
    per_view = []
    for vi, cam in zip(idx, views):
        wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev))
        wins = wins[:3]
        m = window_mask(wins, 1080, 1920)
        gc = torch.randn(3, 1080, 1920, generator=torch.Generator().manual_seed(60 + vi)) * m
        ro, rgrads, gc, _ = oracle_raw("nyc-1M", vi, bg, gc, wins, keys=keys, n_views=6)
        per_view.append((gc.to(dev), rgrads, ro, m))
    bucket = Dm.GradBucket(int(model.get_xyz.shape[0]), dev)
    out = render_batch(views, model, PipelineParams(skip_objects=True, grad_bucket=bucket), bg.to(dev))
    out["render"].backward(torch.stack([pv[0] for pv in per_view]))
    torch.cuda.synchronize()
    for v, (_, _, ro, m) in enumerate(per_view):
        err = (out["render"][v].detach().cpu().double() - ro.color.detach()).abs().max(dim=0).values
        solid = m & ~ro.fragile_px
        assert err[solid].max().item() <= RGB_TOL, (v, err[solid].max().item())
        _note(f"[cfg4 batch of four, view {idx[v]}] solid RGB err {err[solid].max().item():.2e}")
    got = bucket.views()
    names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "scaling": "_scaling", "rotation": "_rotation",
             "opacity": "_opacity"}
    for n in RAW:
        want = sum(pv[1][n] for pv in per_view)
        norm, frac = grad_error(got[names[n]].detach().cpu().reshape(want.shape), want, elem_tol=5 * GRAD_TOL)
        _note(f"[cfg4 batch of four] {n}: normwise {norm:.2e}, elements off {frac:.1e}")
        assert norm <= GRAD_TOL and frac <= 3e-3, (n, norm, frac)


def getattr_grad(model, name):
    return model.named_parameters()[name].grad.detach().cpu()


def test_cfg3_pgd20_colour_attack_at_full_size():
    """BASELINE config 3: PGD-20, L2 on the SH colour (alpha 0.5, eps 5.0: configs/config.yaml:47-48) of S-nyc-1M at
    1080p.  Iteration 0 is checked against oracle-R: with a loss that is linear on a few tile windows the stepped
    colour tensors must equal the reference update rule (gsplat_attack.pgd on the CPU, pinned to attack.py:138-173 by the
    golden fixtures) applied to the oracle's gradients.  Then 20 iterations against the surrogate detector: the loss
    goes down, every Gaussian stays in its eps ball, geometry is untouched."""
    from gsplat_attack import pgd
    from gsplat_attack.attack import pgd_attack
    from gsplat_attack.scenes import make_scene
    D = _hip()
    dev, model, cams = _scene_on_gpu("nyc-1M", 1)
    cam = cams[0]
    bg = torch.zeros(3)
    wins, longest, gx, gy, keys = _windows_for(D, model, cam, bg.to(dev))
    m = window_mask(wins, 1080, 1920)
    gc = torch.randn(3, 1080, 1920, generator=torch.Generator().manual_seed(11)) * m
    ro, rgrads, gc, _ = oracle_raw("nyc-1M", 0, bg, gc, wins, keys=keys)
    gcd = gc.to(dev)
    orig = {n: getattr(model, n).detach().clone() for n in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc",
                                                           "_features_rest")}
    hist = pgd_attack(model, [cam], iters=1, alpha=0.5, epsilon=5.0, groups=("color",), norm="l2", bg=bg.to(dev),
                      loss_fn=lambda imgs: (imgs[0] * gcd).sum(), streams=1)
    torch.cuda.synchronize()
    ref, _, _ = make_scene("nyc-1M", device="cpu", n_views=1)
    ref._features_dc.grad = rgrads["f_dc"].float()
    ref._features_rest.grad = rgrads["f_rest"].float()
    pgd.gaussian_color_l2_attack(ref, 0.5, 5.0, ref._features_rest.detach().clone(), ref._features_dc.detach().clone())
    for n in ("_features_dc", "_features_rest"):
        step_ref = getattr(ref, n).detach() - orig[n].cpu()
        step_hip = getattr(model, n).detach().cpu() - orig[n].cpu()
        assert step_ref.abs().max().item() > 0
        assert (step_hip - step_ref).abs().max().item() <= 2e-3 * step_ref.abs().max().item(), n
    assert abs(hist[0] - float((ro.color * gc.double()).sum())) <= 1e-3 * max(1.0, abs(hist[0]))
    # the attack proper: 20 iterations, surrogate detector, whole image
    hist = pgd_attack(model, [cam], iters=20, alpha=0.5, epsilon=5.0, groups=("color",), norm="l2", bg=bg.to(dev), streams=1)
    assert len(hist) == 20 and hist[-1] < hist[0] and all(math.isfinite(h) for h in hist)
    for n in ("_features_dc", "_features_rest"):
        d = (getattr(model, n).detach() - orig[n]).reshape(orig[n].shape[0], -1).norm(dim=1)
        assert d.max().item() <= 5.0 * (1 + 1e-5) and d.max().item() > 0
    for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
        assert torch.equal(getattr(model, n).detach(), orig[n]), n


@pytest.mark.parametrize("bwd", [2, 4])
@pytest.mark.parametrize("fwd", [1, 2, 4])
def test_every_tile_split_and_tile_map_against_the_oracle(fwd, bwd):
    """GSR_FLAG_FWD_SPLIT x GSR_FLAG_BWD_SPLIT x GSR_FLAG_TILE_MAP on a scene small enough for a full oracle render:
    every kernel instantiation (k_render_fwd<.,1|2|4>, k_render_bwd<.,2|4,.>) and block->tile map gives the oracle's
    numbers; the image is bitwise independent of all of them."""
    from gsplat_attack.scenes import make_scene
    from util import model_inputs
    from test_gpu_parity import run_hip
    D = _hip()
    model, cams, _ = make_scene("nyc-1M", P=20000, width=320, height=180, n_views=1)
    cam = cams[0]
    inp = model_inputs(model, with_objs=False)
    bg = torch.tensor([0.3, 0.2, 0.1])
    gc = torch.randn(3, 180, 320, generator=torch.Generator().manual_seed(1))
    from test_gpu_parity import hip_depth_keys
    ref, rg = O.forward_backward(inp, settings_for(cam, bg), gc, drop_fragile=True,
                                 depth_key=hip_depth_keys(inp, cam, bg))
    gc, _ = O.solid_grads(ref, gc)
    base = None
    for mode in (0, 1, 2, 3, 4, 5):                       # 4, 5: the tile's forward waves as one workgroup (maps 3, 0)
        try:
            shared = D.FLAG_FWD_SHARED if mode >= 4 else 0
            D.set_flags(D.flag_fwd_split(fwd) | D.flag_bwd_split(bwd) | D.flag_tile_map((3, 0)[mode - 4] if mode >= 4 else mode) | shared)
            color, radii, _, grads = run_hip(inp, cam, bg, gc)
        finally:
            D.set_flags(0)
        if base is None:
            base = color
            err = (color.double() - ref.color).abs().max(dim=0).values
            assert err[~ref.fragile_px].max().item() <= RGB_TOL
        assert torch.equal(color, base), (fwd, bwd, mode)
        for k in ("means3D", "shs", "opacities", "scales", "rotations", "means2D"):
            norm, frac = grad_error(grads[k], rg[k], elem_tol=5 * GRAD_TOL)
            assert norm <= GRAD_TOL and frac <= 3e-3, (k, fwd, bwd, mode, norm, frac)
