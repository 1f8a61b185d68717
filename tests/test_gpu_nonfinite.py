"""Non-finite and extreme attribute values (include/gsraster.h, "Non-finite inputs").  A position or
scale step of the attack (reference attack.py:500-511) can drive exp(_scaling) to inf or a mean to NaN.  What must hold:
the call returns (no hang), a Gaussian with a non-finite footprint is culled, every OTHER Gaussian renders and
differentiates bit for bit as in the scene without it, num_rendered stays within (visible Gaussians) x tiles, and a
non-finite colour stays inside the tiles its Gaussian touches."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
NAN, INF = float("nan"), float("inf")


def _setup(P=3000, W=256, H=160):
    from gsplat_attack.scenes import make_scene
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=W, height=H, n_views=1)
    gc = torch.randn(3, H, W, generator=torch.Generator().manual_seed(4)).to(dev)
    return dev, model, cams[0], gc


def _render(model, cam, gc, flags=0):
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import PipelineParams, render
    model.zero_grad()
    with D.extra_flags(flags):
        out = render(cam, model, PipelineParams(skip_objects=True), torch.tensor([0.1, 0.2, 0.3], device=gc.device))
        out["render"].backward(gc)
    torch.cuda.synchronize()
    n = D.last_num_rendered(out["render"])
    return out["render"].detach().clone(), out["radii"].clone(), {k: getattr(model, k).grad.clone() for k in NAMES}, n


def _without(model, idx):
    """The scene without Gaussians `idx` (same storage order otherwise)."""
    from gsplat_attack.gaussian_model import GaussianModel
    keep = torch.ones(model._xyz.shape[0], dtype=torch.bool, device=model._xyz.device)
    keep[idx] = False
    return GaussianModel.from_tensors(model._xyz[keep], model._features_dc[keep], model._features_rest[keep],
                                      model._scaling[keep], model._rotation[keep], model._opacity[keep],
                                      model._objects_dc[keep], device=model._xyz.device), keep


GEOMETRY_POISONS = [
    ("_xyz", (NAN, 0.0, 0.0)), ("_xyz", (INF, 0.0, 1.0)), ("_xyz", (0.0, -INF, 0.0)), ("_xyz", (1e30, 1e30, 1e30)),
    ("_scaling", (NAN, 0.0, 0.0)), ("_scaling", (INF, 0.0, 0.0)), ("_scaling", (1e30, 1e30, 1e30)), ("_scaling", (45.0, 45.0, 45.0)),
    # (a quaternion of 1e30s is not a poison: its float32 norm overflows to inf and F.normalize -- like the fused
    # activation -- turns it into the zero quaternion, the last test's case)
    ("_rotation", (NAN, 0.0, 0.0, 1.0)), ("_rotation", (INF, 0.0, 0.0, 0.0)),
    ("_opacity", (NAN,)),
]


@pytest.mark.parametrize("attr,value", GEOMETRY_POISONS)
@pytest.mark.parametrize("flags", [0, 1])          # with and without the footprint cull (GSR_FLAG_NO_CULL)
def test_a_gaussian_with_a_non_finite_footprint_is_culled_and_harms_nobody(attr, value, flags):
    dev, model, cam, gc = _setup()
    bad = [7, 1500, 2999]
    poisoned = model.clone()
    with torch.no_grad():
        getattr(poisoned, attr)[bad] = torch.tensor(value, device=dev).view(1, -1).expand(len(bad), -1).reshape(
            getattr(poisoned, attr)[bad].shape)
    img, radii, grads, n = _render(poisoned, cam, gc, flags)
    clean, keep = _without(model, bad)
    img0, radii0, grads0, n0 = _render(clean, cam, gc, flags)
    assert int(radii[bad].abs().max()) == 0, "the poisoned Gaussians must be culled"
    assert n == n0
    assert torch.equal(img, img0), "the image must be that of the scene without the poisoned Gaussians"
    assert torch.equal(radii[keep], radii0)
    for k in NAMES:
        assert torch.equal(grads[k][keep], grads0[k]), k
        assert float(grads[k][bad].abs().max()) == 0.0, (k, "a culled Gaussian receives zero gradients")


@pytest.mark.parametrize("log_scale", [3.0, 8.0, 20.0])
def test_enormous_finite_splats_stay_within_the_pair_bound(log_scale):
    """exp(3) = 20, exp(8) = 2981, exp(20) = 4.9e8 world units: footprints far larger than the image.  The rect is the image,
    the radius saturates, the pair count stays within visible x tiles, everything is finite."""
    dev, model, cam, gc = _setup()
    big = model.clone()
    with torch.no_grad():
        big._scaling[[5, 900, 2100]] = log_scale
    img, radii, grads, n = _render(big, cam, gc)
    tiles = ((256 + 15) // 16) * ((160 + 15) // 16)
    assert n <= int((radii > 0).sum()) * tiles
    assert int(radii.max()) <= 1 << 24
    assert torch.isfinite(img).all()
    for k in NAMES:
        assert torch.isfinite(grads[k]).all(), k


def test_non_finite_colours_stay_inside_their_gaussians_tiles():
    """NaN SH coefficients composite as colour 0; infinite ones poison the pixels their Gaussian reaches and no others."""
    import diff_gaussian_rasterization as D
    dev, model, cam, gc = _setup()
    img0, radii0, grads0, n0 = _render(model.clone(), cam, gc)
    vis = torch.nonzero(radii0 > 0).flatten()
    bad = vis[[3, len(vis) // 2]]
    m = model.clone()
    with torch.no_grad():
        m._features_dc[bad] = NAN
    img, radii, grads, n = _render(m, cam, gc)
    assert n == n0 and torch.equal(radii, radii0)
    assert torch.isfinite(img).all(), "a NaN colour composites as 0"
    keep = torch.ones(radii.numel(), dtype=torch.bool, device=dev)
    keep[bad] = False
    for k in NAMES:
        assert torch.isfinite(grads[k][keep]).all(), k
    m2 = model.clone()
    with torch.no_grad():
        m2._features_dc[bad] = INF
    img2, radii2, _, n2 = _render(m2, cam, gc)
    assert n2 == n0
    # pixels outside the 3-sigma squares of the two Gaussians are untouched
    from gsplat_attack.renderer import PipelineParams, render
    out = render(cam, model, PipelineParams(skip_objects=True), torch.tensor([0.1, 0.2, 0.3], device=dev))
    G = D.export_state(out["render"], "G").view(-1, 12)
    touched = torch.zeros(160, 256, dtype=torch.bool, device=dev)
    for b in bad.tolist():
        rx, ry = int(G[b, 10].view(torch.int32)), int(G[b, 11].view(torch.int32))
        x0, x1 = (rx & 0xFFF) - (rx >> 24 & 0xFF), (rx >> 12) & 0xFFF
        y0, y1 = (ry & 0xFFF) - (ry >> 24 & 0xFF), (ry >> 12) & 0xFFF
        touched[16 * y0:16 * y1, 16 * x0:16 * x1] = True
    assert torch.equal(img2[:, ~touched], img0[:, ~touched])


def test_zero_quaternion_and_zero_scale_render():
    dev, model, cam, gc = _setup()
    m = model.clone()
    with torch.no_grad():
        m._rotation[[11]] = 0.0
        m._rotation[[12]] = torch.tensor([1e30, -1e30, 1e30, 1e30], device=dev)     # norm overflows: normalises to zero
        m._scaling[[13, 14]] = -INF                      # exp(-inf) = 0: a point, rendered through the 0.3 px^2 dilation
    img, radii, grads, n = _render(m, cam, gc)
    assert torch.isfinite(img).all()
    for k in NAMES:
        g = grads[k]
        ok = torch.ones(g.shape[0], dtype=torch.bool, device=dev)
        ok[[11, 12, 13, 14]] = False
        assert torch.isfinite(g[ok]).all(), k


def test_batch_of_views_with_a_poisoned_gaussian():
    """The batch entry points apply the same semantics per view."""
    from gsplat_attack.scenes import make_scene
    from gsplat_attack.renderer import PipelineParams, render_batch
    dev, model, cam, gc = _setup()
    _, cams, _ = make_scene("nyc-1M", device=dev, P=3000, width=256, height=160, n_views=3)
    bad = [7, 1500]
    m = model.clone()
    with torch.no_grad():
        m._xyz[bad] = NAN
        m._scaling[[2999]] = INF
    clean, keep = _without(model, bad + [2999])
    pipe = PipelineParams(skip_objects=True)
    bg = torch.zeros(3, device=dev)
    a = render_batch(cams, m, pipe, bg)
    b = render_batch(cams, clean, pipe, bg)
    assert torch.equal(a["render"], b["render"])
    assert int(a["radii"][:, bad + [2999]].abs().max()) == 0
    a["render"].backward(gc.unsqueeze(0).expand(3, -1, -1, -1).contiguous())
    b["render"].backward(gc.unsqueeze(0).expand(3, -1, -1, -1).contiguous())
    for k in NAMES:
        assert torch.equal(getattr(m, k).grad[keep], getattr(clean, k).grad), k
