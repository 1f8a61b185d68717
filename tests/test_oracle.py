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
