"""oracle-R checked against itself: float32 vs float64, finite differences, compositing invariants."""
import math

import pytest
import torch

from gsplat_attack.cameras import look_at_camera
from gsplat_attack.scenes import make_scene
from oracle import oracle_r as O
from util import model_inputs, settings_for


@pytest.fixture(scope="module")
def small():
    model, cams, _ = make_scene("hydrant-1k", P=300, width=64, height=48, n_views=1)
    return model, cams[0]


def test_float32_agrees_with_float64(small):
    model, cam = small
    gc = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(1))
    st = settings_for(cam, torch.tensor([0.2, 0.3, 0.4]))
    o64, g64 = O.forward_backward(model_inputs(model), st, gc, dtype=torch.float64)
    o32, g32 = O.forward_backward(model_inputs(model), st, gc, dtype=torch.float32)
    solid = ~o64.fragile_px
    assert ((o32.color.double() - o64.color).abs().max(dim=0).values[solid].max().item()) < 1e-4
    assert int(((o32.radii != o64.radii) & ~o64.fragile_gauss).sum()) == 0
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        rel = ((g32[k].double() - g64[k]).abs().max() / g64[k].abs().max()).item()
        assert rel < 1e-3, (k, rel)


def test_background_is_blended_with_final_transmittance(small):
    """color(bg) - color(0) == final_T * bg for every pixel: sum_i alpha_i T_i + T_final accounts for all light."""
    model, cam = small
    inp = model_inputs(model)
    bg = torch.tensor([0.7, 0.1, 0.9])
    a = O.rasterize(inp["means3D"], None, inp["opacities"], settings_for(cam, torch.zeros(3)), shs=inp["shs"],
                    scales=inp["scales"], rotations=inp["rotations"])
    b = O.rasterize(inp["means3D"], None, inp["opacities"], settings_for(cam, bg), shs=inp["shs"],
                    scales=inp["scales"], rotations=inp["rotations"])
    assert torch.allclose(b.color - a.color, a.final_T[None] * bg.double()[:, None, None], atol=1e-12)
    assert float(a.final_T.min()) >= 0.0 and float(a.final_T.max()) <= 1.0
    assert torch.equal(a.n_contrib, b.n_contrib)


def test_storage_order_does_not_matter(small):
    """Permuting the Gaussians (no exact depth ties in this scene) leaves the image unchanged."""
    model, cam = small
    inp = model_inputs(model)
    P = inp["means3D"].shape[0]
    perm = torch.randperm(P, generator=torch.Generator().manual_seed(3))
    st = settings_for(cam, torch.zeros(3))
    a = O.rasterize(inp["means3D"], None, inp["opacities"], st, shs=inp["shs"], scales=inp["scales"], rotations=inp["rotations"])
    b = O.rasterize(inp["means3D"][perm], None, inp["opacities"][perm], st, shs=inp["shs"][perm],
                    scales=inp["scales"][perm], rotations=inp["rotations"][perm])
    assert torch.allclose(a.color, b.color, atol=1e-12)
    assert torch.equal(a.radii[perm], b.radii)


def test_gradients_match_central_differences():
    """float64 autograd vs central finite differences on a 6-Gaussian scene where no pixel sits near a
    threshold and no alpha reaches the 0.99 cap (the two straight-through constructions are inactive)."""
    g = torch.Generator().manual_seed(11)
    P = 6
    means = torch.randn(P, 3, generator=g, dtype=torch.float64) * 0.25
    scales = torch.rand(P, 3, generator=g, dtype=torch.float64) * 0.08 + 0.05
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g, dtype=torch.float64))
    opac = torch.rand(P, 1, generator=g, dtype=torch.float64) * 0.5 + 0.2
    shs = torch.randn(P, 16, 3, generator=g, dtype=torch.float64) * 0.3
    cam = look_at_camera((0.0, -0.2, -2.2), (0.0, 0.0, 0.0), fovx=0.7, fovy=0.7, width=32, height=32)
    st = settings_for(cam, torch.tensor([0.1, 0.2, 0.3]))
    gc = torch.randn(3, 32, 32, generator=g, dtype=torch.float64)

    def loss(m, s, r, o, h):
        return (O.rasterize(m, None, o, st, shs=h, scales=s, rotations=r).color * gc).sum()

    params = [p.clone().requires_grad_(True) for p in (means, scales, rots, opac, shs)]
    out = O.rasterize(params[0], None, params[3], st, shs=params[4], scales=params[1], rotations=params[2])
    assert not bool(out.fragile_px.any())
    (out.color * gc).sum().backward()
    eps = 1e-6
    pick = torch.Generator().manual_seed(5)
    for idx, p in enumerate(params):
        flat = p.detach().reshape(-1)
        for j in torch.randperm(flat.numel(), generator=pick)[:6].tolist():
            args_p = [q.detach().clone() for q in params]
            args_m = [q.detach().clone() for q in params]
            args_p[idx].reshape(-1)[j] += eps
            args_m[idx].reshape(-1)[j] -= eps
            fd = (loss(*args_p) - loss(*args_m)).item() / (2 * eps)
            an = p.grad.reshape(-1)[j].item()
            assert abs(fd - an) <= 2e-5 * max(1.0, abs(an)), (idx, j, fd, an)


def test_argument_contract():
    z = torch.zeros
    cam = look_at_camera((0.0, 0.0, -2.0), (0.0, 0.0, 0.0), fovx=0.7, fovy=0.7, width=16, height=16)
    st = settings_for(cam, z(3))
    with pytest.raises(Exception):
        O.rasterize(z(2, 3), None, z(2, 1), st, scales=z(2, 3), rotations=z(2, 4))
    with pytest.raises(Exception):
        O.rasterize(z(2, 3), None, z(2, 1), st, shs=z(2, 16, 3))
    vis = O.mark_visible(torch.tensor([[0.0, 0.0, 0.0], [0.0, 0.0, -5.0]], dtype=torch.float64), st)
    assert vis.tolist() == [True, False]


def test_tile_windows_equal_the_full_render_on_their_pixels():
    """rasterize(tile_windows=...) composites only the chosen tiles: image, transmittance, positions and -- with dL/dC
    zero outside the windows -- every gradient equal those of the full render."""
    model, cams, _ = make_scene("nyc-1M", P=3000, width=150, height=91, n_views=1)   # ragged: 10 x 6 tiles, 5.7 rows
    cam = cams[0]
    inp = model_inputs(model, with_objs=True)
    st = settings_for(cam, torch.tensor([0.3, 0.1, 0.2]))
    wins = [(2, 1, 5, 3), (8, 4, 12, 9), (4, 2, 6, 4)]            # overlapping, and one running over the border
    gen = torch.Generator().manual_seed(8)
    gc = torch.randn(3, 91, 150, generator=gen)
    go = torch.randn(O.NUM_OBJECTS, 91, 150, generator=gen) * 0.2
    full, _ = O.forward_backward(inp, st, gc, go)
    assert full.window_px is None
    win, gw = O.forward_backward(inp, st, gc, go, tile_windows=wins)
    m = win.window_px
    want = torch.zeros(91, 150, dtype=torch.bool)
    for (x0, y0, x1, y1) in wins:
        want[16 * y0:16 * y1, 16 * x0:16 * x1] = True
    assert torch.equal(m, want) and 0 < int(m.sum()) < m.numel()
    assert torch.equal(win.color[:, m], full.color[:, m]) and float(win.color.detach()[:, ~m].abs().max()) == 0.0
    assert torch.equal(win.objects[:, m], full.objects[:, m])
    assert torch.equal(win.final_T[m], full.final_T[m]) and torch.equal(win.n_contrib[m], full.n_contrib[m])
    assert torch.equal(win.fragile_px[m], full.fragile_px[m]) and torch.equal(win.radii, full.radii)
    _, gm = O.forward_backward(inp, st, gc * m, go * m)           # full render, gradient masked to the windows
    for k in gm:
        assert torch.allclose(gw[k], gm[k], rtol=1e-12, atol=1e-14 * float(gm[k].abs().max() + 1)), k
        assert float(gm[k].abs().max()) > 0 or k == "means2D"


def test_fragile_flags_are_per_pixel_and_ghosts_never_contribute():
    """A Gaussian whose radius sits on an integer is fragile: the oracle walks its widened tile rect as ghost entries.
    Ghosts change nothing in the image; only pixels that the fragile Gaussian reaches (alpha >= 1/255) are flagged, not
    the whole tile."""
    cam = look_at_camera((0.0, 0.0, -2.0), (0.0, 0.0, 0.0), fovx=0.7, fovy=0.7, width=64, height=64)
    st = settings_for(cam, torch.zeros(3))
    means = torch.tensor([[0.0, 0.0, 0.0], [0.35, 0.3, 0.2]], dtype=torch.float64)
    rots = torch.tensor([[1.0, 0, 0, 0]] * 2, dtype=torch.float64)
    opac = torch.tensor([[0.02], [0.6]], dtype=torch.float64)             # the fragile one is faint: small footprint
    shs = torch.zeros(2, 16, 3, dtype=torch.float64)
    shs[:, 0] = 1.0

    def radius_of(s0):
        sc = torch.tensor([[s0, s0, s0], [0.03, 0.03, 0.03]], dtype=torch.float64)
        return sc, O.preprocess(means, sc, rots, None, st)
    # bisect the isotropic scale until 3*sqrt(lambda) of Gaussian 0 is an integer to 1e-12 (ceil flips across it)
    lo, hi = 0.05, 0.06
    r_lo = int(radius_of(lo)[1].radii[0])
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if int(radius_of(mid)[1].radii[0]) == r_lo:
            lo = mid
        else:
            hi = mid
    sc, geo = radius_of(lo)
    assert bool(geo.fragile[0]) and not bool(geo.fragile[1])
    assert bool((geo.ghost_max[0] - geo.ghost_min[0] >= geo.rect_max[0] - geo.rect_min[0]).all())
    out = O.rasterize(means, None, opac, st, shs=shs, scales=sc, rotations=rots)
    gid, ranges, ghost = O.build_tile_lists(geo, 64, 64, ghosts=True)
    gid0, ranges0, _ = O.build_tile_lists(geo, 64, 64)
    assert int(ghost.sum()) > 0 and gid0.numel() == int((~ghost).sum()) == out.num_rendered
    # flagged pixels: a strict, non-empty subset of the fragile Gaussian's tiles; every one reached by it
    fp = out.fragile_px
    assert 0 < int(fp.sum()) < 256
    ys, xs = torch.nonzero(fp, as_tuple=True)
    d = torch.stack([geo.xy[0, 0] - xs.double(), geo.xy[0, 1] - ys.double()], 1)
    A, B, C = geo.conic[0]
    power = -0.5 * (A * d[:, 0] ** 2 + C * d[:, 1] ** 2) - B * d[:, 0] * d[:, 1]
    assert bool((opac[0, 0] * torch.exp(power) >= (1 - 1e-3) / 255.0).all())
    # the image is what the two real Gaussians give: the same scene with the scale nudged off the edge renders equally
    sc2 = sc.clone(); sc2[0] *= 1.0 - 1e-9
    out2 = O.rasterize(means, None, opac, st, shs=shs, scales=sc2, rotations=rots)
    assert torch.allclose(out.color, out2.color, atol=1e-7)
