"""Round 5: the L2 step's global norms out of the raster backward (gsr_ctx_request_sumsq / GradNorms / gsr_pgd_step_normed),
reference attack.py:53-119, 138-173."""
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _scene(P=40000, W=320, H=192, n_views=2):
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=W, height=H, n_views=n_views)
    return dev, model, cams


@pytest.mark.parametrize("color_only", [False, True])
@pytest.mark.parametrize("bucket", [False, True])
def test_backward_leaves_the_sums_of_squares_of_the_gradients_it_writes(color_only, bucket):
    from diff_gaussian_rasterization import GradBucket, GradNorms
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams = _scene()
    if color_only and bucket:
        pytest.skip("a bucket takes all 59 floats")
    if color_only:
        for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
            getattr(model, n).requires_grad_(False)
    P = model.get_xyz.shape[0]
    norms = GradNorms(dev)
    b = GradBucket(P, dev) if bucket else None
    pipe = PipelineParams(skip_objects=True, viewspace_grad=not color_only, grad_norms=norms, grad_bucket=b)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    gc = torch.randn(3, 192, 320, generator=torch.Generator().manual_seed(3)).to(dev)
    norms.begin()
    model.zero_grad()
    render(cams[0], model, pipe, bg)["render"].backward(gc)
    if b is not None:
        b.assign_to(model)
    torch.cuda.synchronize()
    want = ("_features_dc", "_features_rest") if color_only else NAMES
    assert norms.writes == 1 and set(norms.names) == set(want)
    for n in want:
        g = getattr(model, n).grad
        ref = float((g.double() ** 2).sum())
        got = float(norms.sumsq_of(n))
        assert ref > 0 and abs(got - ref) <= 2e-6 * ref, (n, got, ref)
    for n in set(NAMES) - set(want):
        assert norms.sumsq_of(n) is None
    # a second view's gradients on top: the sums no longer describe what .grad holds
    render(cams[1], model, pipe, bg)["render"].backward(gc)
    assert norms.writes == 2 and norms.sumsq_of("_features_dc") is None
    # a new iteration: valid again, and equal to the new gradient's
    norms.begin()
    if b is not None:
        b.reset()
    model.zero_grad()
    render(cams[1], model, pipe, bg)["render"].backward(gc)
    if b is not None:
        b.assign_to(model)
    g = model._features_rest.grad
    ref = float((g.double() ** 2).sum())
    assert abs(float(norms.sumsq_of("_features_rest")) - ref) <= 2e-6 * ref


def test_normed_step_equals_the_step_that_sums_the_gradient_itself():
    from gsplat_attack import pgd
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    for rows, cols in ((50000, 45), (50000, 3), (777, 4), (4097, 1)):
        x0 = torch.randn(rows, cols, generator=g).to(dev)
        x = (x0 + 0.3 * torch.randn(rows, cols, generator=g).to(dev)).contiguous()
        grad = torch.randn(rows, cols, generator=g).to(dev)
        grad[::3] = 0.0
        ss = (grad.double() ** 2).sum().reshape(1)
        a, b = x.clone(), x.clone()
        pgd.l2_step_(a, grad, 0.5, 0.4, x0)
        pgd.l2_step_(b, grad, 0.5, 0.4, x0, sumsq=ss)
        assert (a - b).abs().max().item() <= 2e-6, (rows, cols)
        assert (b - x).abs().max().item() > 0


def test_colour_attack_with_fused_norms_follows_the_same_trajectory():
    """BASELINE config 3's shape (one view per iteration, L2 on the SH colour): the loop whose steps take their norms from
    the raster backward against the loop whose steps sum the gradient themselves."""
    from gsplat_attack.attack import pgd_attack
    dev, model, cams = _scene(n_views=1)
    ref = model.clone()
    bg = torch.zeros(3, device=dev)
    h1 = pgd_attack(model, cams[:1], iters=4, groups=("color",), bg=bg, streams=1, fused_norms=True)
    h0 = pgd_attack(ref, cams[:1], iters=4, groups=("color",), bg=bg, streams=1, fused_norms=False)
    assert h1 == pytest.approx(h0, rel=1e-5, abs=1e-7)
    for n in ("_features_dc", "_features_rest"):
        a, b = getattr(model, n).detach(), getattr(ref, n).detach()
        assert (a - b).abs().max().item() <= 1e-5 and (a - b).abs().max().item() < 0.1 * (a - cams[0].camera_center.new_zeros(1)).abs().max().item()
    # all five groups, one view: the bucket path
    m2, r2 = model.clone(), model.clone()
    groups = ("color", "position", "scaling", "rotation", "opacity")
    g1 = pgd_attack(m2, cams[:1], iters=3, groups=groups, bg=bg, streams=1, fused_norms=True)
    g0 = pgd_attack(r2, cams[:1], iters=3, groups=groups, bg=bg, streams=1, fused_norms=False)
    assert g1 == pytest.approx(g0, rel=1e-4, abs=1e-6)
    for n in NAMES:
        assert (getattr(m2, n).detach() - getattr(r2, n).detach()).abs().max().item() <= 2e-5, n


def test_backward_without_object_gradients_after_an_object_forward_walks_segments():
    """The reference's render() always composites the 16 object channels and the attack never differentiates them
    (gaussian_renderer/__init__.py:80-95, attack.py:486-494).  The forward with objects now stores the segment-boundary
    records too, so that backward is the segmented K7 of the objects-off path: same image, bit-equal gradients -- on a scene
    whose lists are long enough to be split."""
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams = _scene(P=150000, W=480, H=270, n_views=1)
    import diff_gaussian_rasterization as D
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)
    gc = torch.randn(3, 270, 480, generator=torch.Generator().manual_seed(4)).to(dev)
    out = {}
    for objects in (False, True):
        model.zero_grad()
        r = render(cams[0], model, PipelineParams(skip_objects=not objects), bg, 2.0)
        if not objects:
            lens = D.export_state(r["render"], "ranges").view(-1, 2).long()
            assert int((lens[:, 1] - lens[:, 0]).max()) > 512          # split lists exist
        r["render"].backward(gc)
        torch.cuda.synchronize()
        out[objects] = (r["render"].detach().clone(), {n: getattr(model, n).grad.clone() for n in NAMES})
        if objects:
            assert float(r["render_object"].abs().max()) > 0
    assert torch.equal(out[False][0], out[True][0])
    for n in NAMES:
        assert torch.equal(out[False][1][n], out[True][1][n]), n


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
