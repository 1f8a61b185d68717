"""The float32 yardstick on hard cases: fragile pixels must land on a float32 or a float64 outcome; strongly anisotropic
splats and needles against oracle-R in float32 and float64, with and without GSR_FLAG_NEEDLE_DOUBLE."""
import math
import pytest
import torch
from oracle import oracle_r as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _hip():
    import diff_gaussian_rasterization as D
    D._load()
    return D


NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _scene(P=40000, W=320, H=192, n_views=2):
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=W, height=H, n_views=n_views)
    return dev, model, cams


@pytest.mark.parametrize("scene,kw", [("hydrant-1k", {}), ("nyc-1M", dict(P=60000, width=640, height=360))])
def test_fragile_pixels_land_on_a_float32_or_float64_outcome(scene, kw):
    import test_gpu_parity as T
    D = _hip()
    from gsplat_attack.renderer import PipelineParams, render
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene(scene, device=dev, n_views=1, **kw)
    cam = cams[0]
    H, W = cam.image_height, cam.image_width
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    with D.extra_flags(D.FLAG_NO_CULL):
        full = render(cam, model, PipelineParams(skip_objects=True), bg)
    depth = D.export_state(full["render"], "G").view(-1, 12)[:, 9].cpu()
    radii = full["radii"].cpu()
    depth = torch.where(radii > 0, depth, torch.zeros_like(depth))
    with torch.no_grad():
        hip = render(cam, model, PipelineParams(skip_objects=True), bg)["render"].cpu().double()
    cpu = lambda t: t.detach().cpu()
    st = O.Settings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), cpu(bg), 1.0, cpu(cam.world_view_transform),
                    cpu(cam.full_proj_transform), 3, cpu(cam.camera_center), False, False)
    O.check_depth_keys(depth, cpu(model.get_xyz), st, radii)
    args = (cpu(model.get_xyz), None, cpu(model.get_opacity), st)
    kws = dict(shs=cpu(model.get_features), scales=cpu(model.get_scaling), rotations=cpu(model.get_rotation), depth_key=depth)
    with torch.no_grad():
        r64 = O.rasterize(*args, dtype=torch.float64, **kws)
        r32 = O.rasterize(*args, dtype=torch.float32, **kws)
    # every pixel, fragile or not, against the float32 yardstick (tests/util.py: pixel_yardstick); round 3 accepted 10 %
    # of the fragile pixels on neither outcome and held them to 1e-2
    from util import pixel_yardstick, yardstick_line
    y = pixel_yardstick(hip, r64.color, r32.color, r64.fragile_px | r32.fragile_px, tol=1e-4)
    print(yardstick_line(scene, y))
    assert y["worst_solid"] <= 1e-4 and y["neither_solid"] == 0
    assert y["neither_px"] <= max(T.NEITHER_MIN_PX, T.NEITHER_CAP * y["fragile"] * y["n"]), yardstick_line(scene, y)


def test_needle_splats_keep_their_geometry_gradients():
    """Splats 100:1 long that cross the whole image: cov2D ~ l1 u u^T, and the published dL/dconic -> dL/dcov2D formula
    (-c^2 dA + b c dB - b^2 dC) / det^2 cancels to first order.  In float32 it lost l1 / l2 times its rounding -- a random
    configuration (tests/diag_fuzz.py seed 1192) had one needle whose rotation gradient was 4.5 % off --; K9 forms the
    cancelling sums in double (gsr_math.h project_splat_bwd).  Six such needles among ordinary splats, against oracle-R
    float64 on the solid pixels (the oracle flags a band along every needle as fragile: a float32 conic of that shape is
    uncertain by 1e-4, times terms of 1e4 in the exponent); the float32 formulation fails on the rotation gradients."""
    import test_gpu_parity as T
    from gsplat_attack.cameras import look_at_camera
    g = torch.Generator().manual_seed(5)
    P = 300
    xyz = torch.randn(P, 3, generator=g) * torch.tensor([0.5, 0.4, 0.5])
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.3 + math.log(0.03))
    scales[:6, 0] = 2.0                                      # six needles among ordinary splats (the oracle flags the pixels
    scales[:6, 1:] = 0.02                                    # of a needle-only scene as fragile: their float32 conics differ)
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) + 0.5)
    shs = torch.randn(P, 16, 3, generator=g) * 0.2
    shs[:, 0] += torch.randn(P, 3, generator=g)
    inp = dict(means3D=xyz, shs=shs, opacities=opac, scales=scales, rotations=rots)
    cam = look_at_camera((2.1, 0.8, 2.4), (0.0, 0.0, 0.0), fovx=0.9, width=120, height=72)
    rep = T.check(inp, cam, torch.tensor([0.1, 0.3, 0.2]), sh_degree=2, scale_modifier=1.0, seed=5, frag_frac=0.6,
                  elem_frac=5e-3)
    print("needle splats, normwise gradient error vs float64:", {k: f"{v[0]:.2e}" for k, v in rep.items()})
    assert rep["rotations"][0] <= 2e-4 and rep["scales"][0] <= 3e-4       # float32 formulation: rotations 5.2e-4


@pytest.mark.parametrize("seed", [4, 45, 102, 106, 0, 6, 8, 13])
def test_anisotropic_splats_against_the_float32_yardstick(seed):
    import test_gpu_parity as T
    from fuzz_cases import aniso_case
    inp, cam, bg, kw, desc = aniso_case(seed)
    rep = T.check(inp, cam, bg, frag_frac=1.0, elem_frac=5e-3, f32_grads=True, **kw)
    y = T.check.last_yardstick
    print(f"aniso seed {seed} {desc}: hip image err {y['worst_any']:.2e}, float32 oracle {y['f32_vs_f64']:.2e}; "
          + ", ".join(f"{k} {v[0]:.1e}/{v[2]:.1e}" for k, v in rep.items()))


@pytest.mark.parametrize("seed,elem_frac", [(106, 1e-3), (102, 0.035), (45, 1e-3), (41, 7e-3)])
def test_needle_splats_under_the_double_chain_flag(seed, elem_frac):
    """GSR_FLAG_NEEDLE_DOUBLE: the anisotropic draws whose gradient elements float32 cannot hold (round 3 / 4: a 1500:1
    needle's dL/dmean2D 2.3 % off, a 2300:1 needle's dL/dmean3D 8-11 % off, while the float32 oracle is itself 0.4-5 % off)
    against the float64 oracle OUTRIGHT -- no float32 yardstick: solid pixels to 1e-4, every gradient group to 1e-3, at most
    one significant element in a thousand off by more than 5e-3 (seed 102: 21 Gaussians, 84 % of the pixels fragile, 32
    significant screen-space elements: one of them may be).  Seed 41: the 440:1 needle whose rotation gradient was 0.64 % off
    with the double conic rounded entry by entry to its float32 record (gsr_math.h needle_conic_to_float); one of its 168
    significant scale elements -- the needle's SHORT axis, 0.68 % off here, 0.64 % in the float32 oracle -- is the
    compositor's float32 sums of dL/dconic, which the flag does not touch."""
    import diff_gaussian_rasterization as D
    import test_gpu_parity as T
    from fuzz_cases import aniso_case
    inp, cam, bg, kw, desc = aniso_case(seed)
    with D.extra_flags(D.FLAG_NEEDLE_DOUBLE):
        rep = T.check(inp, cam, bg, frag_frac=1.0, elem_frac=elem_frac, f32_grads=False, **kw)
    assert rep, "nothing compared"
    print(f"needle seed {seed} {desc}: " + ", ".join(f"{k} {v[0]:.1e}" for k, v in rep.items()))


def test_needle_flag_leaves_ordinary_splats_alone():
    """Splats whose covariance eigenvalues are less than 256 apart keep the published float32 conic under the flag: a scene
    without needles renders and differentiates to the same numbers (to rounding: the flag selects another instantiation
    of the geometry kernel, whose float32 chain the compiler may contract differently)."""
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams = _scene(n_views=1)
    with torch.no_grad():
        model._scaling.copy_(model._scaling.mean(dim=1, keepdim=True).expand(-1, 3) + 0.05 * torch.randn_like(model._scaling))
    bg = torch.zeros(3, device=dev)
    gc = torch.randn(3, 192, 320, generator=torch.Generator().manual_seed(8)).to(dev)
    res = []
    for flags in (0, D.FLAG_NEEDLE_DOUBLE):
        with D.extra_flags(flags):
            model.zero_grad()
            out = render(cams[0], model, PipelineParams(skip_objects=True), bg)
            out["render"].backward(gc)
            torch.cuda.synchronize()
            res.append((out["render"].detach().clone(), {n: getattr(model, n).grad.clone() for n in NAMES}))
    assert (res[0][0] - res[1][0]).abs().max().item() <= 2e-6
    for n in NAMES:
        scale = res[0][1][n].abs().max().item()
        assert (res[0][1][n] - res[1][1][n]).abs().max().item() <= 2e-5 * scale, n
