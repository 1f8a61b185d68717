"""Gradient buckets (gsr_backward_raw_into, GradBucket) and the L2 steps' norms out of the backward (gsr_ctx_request_sumsq,
GradNorms): accumulation over views, per-view outputs, unwritten buckets, the fused and the normed step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hip():
    import diff_gaussian_rasterization as D
    D._load()
    return D


def _small_scene(n_views=3, P=20000, w=320, h=192):
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=w, height=h, n_views=n_views)
    return dev, model, cams


def _hip():
    import diff_gaussian_rasterization as D
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    D._load()
    return D


def _small_scene(n_views=3, P=20000, w=320, h=192):
    from gsplat_attack.scenes import make_scene
    _hip()
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=w, height=h, n_views=n_views)
    return dev, model, cams


NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _scene(P=40000, W=320, H=192, n_views=2):
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=W, height=H, n_views=n_views)
    return dev, model, cams


def test_backward_into_a_bucket_adds_views_like_autograd_does():
    D = _hip()
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams = _small_scene()
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    gcs = [torch.randn(3, 192, 320, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(3)]
    pipe = PipelineParams(skip_objects=True)
    model.zero_grad()
    per_view = []
    for cam, gc in zip(cams, gcs):                         # reference behaviour: autograd accumulates in .grad
        model.zero_grad()
        render(cam, model, pipe, bg)["render"].backward(gc)
        per_view.append({n: getattr(model, n).grad.clone() for n in D.GradBucket.NAMES})
    want = {n: sum(v[n] for v in per_view) for n in D.GradBucket.NAMES}
    model.zero_grad()
    bucket = D.GradBucket(model.get_xyz.shape[0], dev)
    bucket.flat.fill_(float("nan"))                        # the first backward must overwrite, not add
    pipe_b = PipelineParams(skip_objects=True, grad_bucket=bucket)
    for cam, gc in zip(cams, gcs):
        render(cam, model, pipe_b, bg)["render"].backward(gc)
    assert all(getattr(model, n).grad is None for n in D.GradBucket.NAMES)       # autograd got nothing for them
    got = bucket.views()
    for n in D.GradBucket.NAMES:
        scale = want[n].abs().max().clamp_min(1e-30)
        assert torch.isfinite(got[n]).all(), n
        assert ((got[n].view(want[n].shape) - want[n]).abs().max() / scale).item() <= 2e-6, n
    # first view alone: bitwise what the plain backward writes
    bucket.reset()
    render(cams[0], model, pipe_b, bg)["render"].backward(gcs[0])
    for n in D.GradBucket.NAMES:
        assert torch.equal(bucket.views()[n].view(per_view[0][n].shape), per_view[0][n]), n
    bucket.assign_to(model)
    assert model._features_rest.grad.data_ptr() == bucket.views()["_features_rest"].data_ptr()


def test_pgd_attack_with_buckets_takes_the_same_steps():
    from gsplat_attack.attack import pgd_attack
    dev, model, cams = _small_scene(n_views=4)
    a, b = model.clone(), model.clone()
    kw = dict(iters=3, groups=("color", "position", "scaling", "rotation", "opacity"), alpha=0.05, epsilon=0.5)
    ha = pgd_attack(a, cams, use_buckets=False, streams=1, **kw)
    hb = pgd_attack(b, cams, use_buckets=True, streams=1, **kw)
    hc_model = model.clone()
    hc = pgd_attack(hc_model, cams, use_buckets=True, streams=3, batched=False, **kw)     # per-view loop over three streams
    assert max(abs(x - y) for x, y in zip(ha, hb)) <= 1e-5 * max(1.0, max(abs(x) for x in ha))
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        ref = getattr(a, n).detach()
        for other in (b, hc_model):
            d = (getattr(other, n).detach() - ref).abs().max().item()
            assert d <= 2e-5 * max(1.0, ref.abs().max().item()), (n, d)
    assert len(hc) == 3


def test_unwritten_bucket_of_a_rank_with_views_raises():
    from gsplat_attack.attack import pgd_attack
    from gsplat_attack.renderer import PipelineParams
    import gsplat_attack.attack as A
    dev, model, cams = _small_scene(n_views=1)
    # a pipe that passes the predicate when the buckets are made and then renders through the classic surface
    pipe = PipelineParams(skip_objects=True)
    real = A.render

    def classic_render(cam, pc, p, bg, *a, **k):
        q = PipelineParams(skip_objects=True, fused_activations=False)
        return real(cam, pc, q, bg, *a, **k)
    A.render = classic_render
    try:
        with pytest.raises(RuntimeError, match="bucket was not written"):
            pgd_attack(model, cams, iters=1, groups=("color", "position"), pipe=pipe, streams=1)
    finally:
        A.render = real


def test_per_view_gradients_are_overwritten_when_a_bucket_accumulates():
    """(medium) dmeans2D (viewspace_points.grad) and dL/dobjects belong to ONE view: with a bucket that already holds
    another view's gradients (accumulate mode) they must come out exactly as without a bucket -- they used to be
    uninitialised memory plus the gradient."""
    D = _hip()
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams = _small_scene(n_views=2)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    H, W = cams[0].image_height, cams[0].image_width
    g = torch.Generator().manual_seed(8)
    gc = torch.randn(3, H, W, generator=g).to(dev)
    go = (torch.randn(16, H, W, generator=g) * 0.2).to(dev)
    with torch.no_grad():
        model._objects_dc.copy_(torch.randn(model._objects_dc.shape, generator=g).to(dev))

    def run(pipe, cam):
        out = render(cam, model, pipe, bg)
        ((out["render"] * gc).sum() + (out["render_object"] * go).sum()).backward()
        torch.cuda.synchronize()
        return out["viewspace_points"].grad.detach().clone()
    model.zero_grad()
    vs_plain = run(PipelineParams(), cams[1])
    obj_plain = model._objects_dc.grad.detach().clone()
    plain = {n: getattr(model, n).grad.detach().clone() for n in D.GradBucket.NAMES}
    model.zero_grad()
    bucket = D.GradBucket(int(model.get_xyz.shape[0]), dev)
    pipe_b = PipelineParams(grad_bucket=bucket)
    run(pipe_b, cams[0])                                        # first view: overwrites the bucket
    first = bucket.flat.clone()
    model._objects_dc.grad = None
    # poison what the caching allocator will hand out next: the per-view outputs are torch.empty
    junk = [torch.full((int(model.get_xyz.shape[0]), k), float("nan"), device=dev) for k in (3, 16)]
    del junk
    vs_b = run(pipe_b, cams[1])                                 # second view: ADDS into the bucket
    assert torch.equal(vs_b, vs_plain)
    assert torch.equal(model._objects_dc.grad, obj_plain)
    want = first + torch.cat([plain[n].reshape(-1) for n in D.GradBucket.NAMES])
    assert (bucket.flat - want).abs().max().item() <= 1e-6 * want.abs().max().item()


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
