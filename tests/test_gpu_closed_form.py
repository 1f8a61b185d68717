"""Closed-form pins that do not go through the oracle: one Gaussian, two overlapping Gaussians, the near plane -- camera,
footprint, pixel values and gradients derived by hand from SURVEY.md section 8(a) and written as literals."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hip():
    import diff_gaussian_rasterization as D
    D._load()
    return D


SH_C0 = 0.28209479177387814


def _closed_form_settings(D, dev, bg):
    n, f = 0.01, 100.0
    P = torch.zeros(4, 4)
    P[0, 0] = 1.0
    P[1, 1] = 1.0
    P[2, 2] = f / (f - n)
    P[2, 3] = -(f * n) / (f - n)
    P[3, 2] = 1.0
    view = torch.eye(4)
    full = view @ P.t()                                    # p_hom = [p, 1] . full
    return D.GaussianRasterizationSettings(32, 32, 1.0, 1.0, torch.tensor(bg, device=dev), 1.0, view.to(dev), full.to(dev), 0,
                                           torch.zeros(3, device=dev), False, False)


def _render_closed(D, dev, xyz, scale, opac, rgb, bg):
    n = len(xyz)
    means = torch.tensor(xyz, dtype=torch.float32, device=dev, requires_grad=True)
    sh = torch.tensor([[[(c - 0.5) / SH_C0 for c in col]] for col in rgb], dtype=torch.float32, device=dev, requires_grad=True)
    op = torch.tensor(opac, dtype=torch.float32, device=dev).view(n, 1).requires_grad_(True)
    sc = torch.tensor([[s, s, s] for s in scale], dtype=torch.float32, device=dev, requires_grad=True)
    rot = torch.tensor([[1.0, 0.0, 0.0, 0.0]] * n, dtype=torch.float32, device=dev, requires_grad=True)
    st = _closed_form_settings(D, dev, bg)
    color, radii, _ = D.GaussianRasterizer(raster_settings=st)(
        means3D=means, means2D=torch.zeros(n, 3, device=dev), opacities=op, shs=sh, scales=sc, rotations=rot)
    return color, radii, dict(means=means, sh=sh, op=op, sc=sc, rot=rot)


def test_closed_form_one_gaussian_pixel_values_and_gradients():
    D = _hip()
    dev = torch.device("cuda:0")
    bg = [0.1, 0.2, 0.3]
    color, radii, leaf = _render_closed(D, dev, [[0.0, 0.0, 4.0]], [0.5], [0.6], [[0.8, 0.4, 0.2]], bg)
    assert radii.tolist() == [7]
    # literals (see the derivation above): alpha = 0.6 * 0.943518284537 = 0.566110970722
    px = color[:, 15, 15].detach().cpu().double()
    want = torch.tensor([0.49627767950558, 0.31322219414445, 0.24338890292777], dtype=torch.float64)
    assert (px - want).abs().max().item() <= 2e-6
    # the whole image from the same formula, pixel by pixel: alpha >= 1/255 <=> dx^2 + dy^2 <= 2 * 4.3 * ln(255 * 0.6) = 43.26
    ys, xs = torch.meshgrid(torch.arange(32.0, dtype=torch.float64), torch.arange(32.0, dtype=torch.float64), indexing="ij")
    r2 = (15.5 - xs) ** 2 + (15.5 - ys) ** 2
    alpha = 0.6 * torch.exp(-0.5 * r2 / 4.3)
    alpha = torch.where(alpha >= 1.0 / 255.0, alpha, torch.zeros_like(alpha))
    assert int((alpha > 0).sum()) == int((r2 <= 43.261766).sum())
    rgb = torch.tensor([0.8, 0.4, 0.2], dtype=torch.float64)
    bgt = torch.tensor(bg, dtype=torch.float64)
    img = rgb[:, None, None] * alpha[None] + bgt[:, None, None] * (1.0 - alpha[None])
    assert (color.detach().cpu().double() - img).abs().max().item() <= 3e-6
    # gradients of L = red channel of pixel (15, 15):  dL/do = G (r - bg_r),  dL/dsh_r = C0 alpha,
    # dL/dX = (r - bg_r) o G (-dx / 4.3) dpx/dX with dpx/dX = W / (2 z tan) = 4;  by symmetry dL/dY is the same number
    color[0, 15, 15].backward()
    assert abs(leaf["op"].grad.item() - 0.660462799176) <= 2e-6
    assert abs(leaf["sh"].grad[0, 0, 0].item() - 0.159696956407) <= 1e-6
    assert leaf["sh"].grad[0, 0, 1:].abs().max().item() == 0.0
    assert abs(leaf["means"].grad[0, 0].item() - (-0.184315199770)) <= 2e-6
    assert abs(leaf["means"].grad[0, 1].item() - (-0.184315199770)) <= 2e-6


def test_closed_form_two_overlapping_gaussians_blend_front_to_back():
    """Front splat (z = 4, o = 0.6, rgb (.8,.4,.2)) over a back splat with the same screen footprint (z = 8, s = 1:
    16 * 1 / 8 = 2, o = 0.5, rgb (.2,.9,.5)), STORED back first: the depth sort must put the z = 4 one in front.
    Pixel (15,15): a1 = 0.566111, a2 = 0.5 * 0.943518 = 0.471759;  C = c1 a1 + c2 a2 (1 - a1) + bg (1 - a1)(1 - a2)."""
    D = _hip()
    dev = torch.device("cuda:0")
    bg = [0.1, 0.2, 0.3]
    color, radii, leaf = _render_closed(D, dev, [[0.0, 0.0, 8.0], [0.0, 0.0, 4.0]], [1.0, 0.5], [0.5, 0.6],
                                        [[0.2, 0.9, 0.5], [0.8, 0.4, 0.2]], bg)
    assert radii.tolist() == [7, 7]
    px = color[:, 15, 15].detach().cpu().double()
    want = torch.tensor([0.516746791134765, 0.4565059755487128, 0.2843271261861336], dtype=torch.float64)
    assert (px - want).abs().max().item() <= 3e-6
    # dL/d(o_back) for L = red of that pixel: G (1 - a1) (r2 - bg_r) = 0.943518 * 0.433889 * 0.1
    color[0, 15, 15].backward()
    assert abs(leaf["op"].grad[0].item() - 0.943518284537 * (1 - 0.566110970722) * (0.2 - 0.1)) <= 2e-6
    # dL/d(o_front) = G (r1 - [c2 a2 + bg (1 - a2)]) = 0.943518 * (0.8 - (0.2 * 0.471759 + 0.1 * 0.528241))
    a2 = 0.5 * 0.943518284537
    assert abs(leaf["op"].grad[1].item() - 0.943518284537 * (0.8 - (0.2 * a2 + 0.1 * (1 - a2)))) <= 2e-6


def test_closed_form_near_plane_keeps_z_above_0p2_only():
    """View-space z <= 0.2 is culled (radius 0, no contribution, zero gradients); z slightly above is kept.  The kept
    splat: z = 0.21, s = 0.02625 => 16 s / z = 2, the footprint of the cases above."""
    D = _hip()
    dev = torch.device("cuda:0")
    bg = [0.1, 0.2, 0.3]
    color, radii, leaf = _render_closed(D, dev, [[0.0, 0.0, 0.2], [0.0, 0.0, 0.21], [0.0, 0.0, -3.0]],
                                        [0.025, 0.02625, 0.5], [0.9, 0.6, 0.9], [[1.0, 1.0, 1.0], [0.8, 0.4, 0.2], [1.0, 0.0, 1.0]], bg)
    assert radii.tolist() == [0, 7, 0]
    px = color[:, 15, 15].detach().cpu().double()
    want = torch.tensor([0.49627767950558, 0.31322219414445, 0.24338890292777], dtype=torch.float64)
    assert (px - want).abs().max().item() <= 5e-6
    color.sum().backward()
    for k in ("means", "sh", "op", "sc", "rot"):
        g = leaf[k].grad
        assert g[0].abs().max().item() == 0.0 and g[2].abs().max().item() == 0.0, k
    assert leaf["op"].grad[1].abs().item() > 1.0
