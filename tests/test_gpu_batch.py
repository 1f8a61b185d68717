"""A batch of views through ONE launch chain (gsr_forward_raw_batch / gsr_backward_raw_batch_into).

The reference's batch is a loop of render() calls on one set of attributes whose backward passes add up in .grad
(reference attack.py:476-494).  The batch entry points render the B views as one virtual scene; what they must give:
every view's image and radii bit for bit those of the single-view call, every view's screen-space gradient likewise, and
the 59 attribute gradients equal to the B single-view backward passes accumulated in view order (gsr_backward_raw_into:
the first overwrites, the others add).  One view of every batch is also held against oracle-R directly."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


PARAMS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _single_views(model, cams, pipe, bg_of, gcs, scale=1.0):
    """The reference's loop: one render() + backward per view, gradients accumulated in a GradBucket in view order."""
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import render
    import copy
    P = int(model.get_xyz.shape[0])
    bucket = D.GradBucket(P, model.get_xyz.device)
    pipe = copy.copy(pipe)
    pipe.grad_bucket = bucket
    imgs, radii, vs = [], [], []
    exact = torch.zeros(59 * P, dtype=torch.float64, device=model.get_xyz.device)   # the views' gradients summed in double
    own = D.GradBucket(P, model.get_xyz.device)
    pipe_own = copy.copy(pipe)
    pipe_own.grad_bucket = own
    for v, cam in enumerate(cams):
        out = render(cam, model, pipe, bg_of(v), scale)
        out["render"].backward(gcs[v])
        imgs.append(out["render"].detach())
        radii.append(out["radii"])
        vs.append(out["viewspace_points"].grad.clone() if out["viewspace_points"] is not None and
                  out["viewspace_points"].grad is not None else None)
        own.reset()
        render(cam, model, pipe_own, bg_of(v), scale)["render"].backward(gcs[v])
        exact += own.flat.double()
    bucket.exact = exact
    return torch.stack(imgs), torch.stack(radii), vs, bucket


def _batched(model, cams, pipe, bg, gcs, scale=1.0, per_view_bg=None):
    import diff_gaussian_rasterization as D
    from gsplat_attack import renderer as R
    import copy
    P = int(model.get_xyz.shape[0])
    bucket = D.GradBucket(P, model.get_xyz.device)
    pipe = copy.copy(pipe)
    pipe.grad_bucket = bucket
    if per_view_bg is None:
        out = R.render_batch(cams, model, pipe, bg, scale)
    else:
        sts = [R._settings(c, model, pipe, per_view_bg[v], scale) for v, c in enumerate(cams)]
        vsp = torch.zeros(len(cams), P, 3, device=model.get_xyz.device, requires_grad=True)
        image, radii = D.rasterize_gaussians_raw_batch(model._xyz, vsp, model._features_dc, model._features_rest,
                                                       model._opacity, model._scaling, model._rotation, sts, grad_bucket=bucket)
        out = dict(render=image, radii=radii, viewspace_points=vsp)
    out["render"].backward(torch.stack(list(gcs)))
    return out, bucket


def _check_equal(model, cams, gcs, bg_list=None, scale=1.0):
    from gsplat_attack.renderer import PipelineParams
    dev = model.get_xyz.device
    pipe = PipelineParams(skip_objects=True)
    bgs = bg_list if bg_list is not None else [torch.tensor([0.1, 0.2, 0.3], device=dev)] * len(cams)
    imgs, radii, vs, b1 = _single_views(model, cams, pipe, lambda v: bgs[v], gcs, scale)
    out, b2 = _batched(model, cams, pipe, bgs[0], gcs, scale, per_view_bg=bgs if bg_list is not None else None)
    torch.cuda.synchronize()
    assert torch.equal(out["render"].detach(), imgs), "batched images differ from the single-view renders"
    assert torch.equal(out["radii"], radii)
    g = out["viewspace_points"].grad
    for v in range(len(cams)):
        assert torch.equal(g[v], vs[v]), f"view {v}: screen-space gradient differs"
    assert b1.used and b2.used
    # The fused per-Gaussian kernel of the batch sums the views in registers and pushes dL/dSigma3D through scale and
    # rotation once (k_pre_bwd_batch): the same sum in another association -- float32 rounding of a sum of B terms, per
    # tensor.  (GSR_BATCH_K9=0 runs one k_pre_bwd launch per view instead and is bit-equal to the single-view loop.)
    import os
    exact = os.environ.get("GSR_BATCH_K9", "1") == "0"
    # Yardstick: the B single-view gradients summed in double.  The batch must be no further from that sum than the
    # float32 accumulation of the single-view loop itself (factor 3), or within `floor` of the tensor's largest gradient.
    # floor: 1e-5, and 1e-4 for scale and rotation -- the batch pushes the SUM of dL/dSigma3D through scale and rotation once
    # (k_pre_bwd_batch), the loop pushes every view's share and adds the results: two float32 roundings of one linear map
    # whose error is ~1e-7 |dL/dSigma3D|, which for an anisotropic splat is 10-100x the scale gradient of its long axis
    # (tests/diag_batch_err.py: worst element 7e-6 of ITS OWN value off, 4e-6 of the tensor's largest at 60 k Gaussians,
    # 5e-5 at 1 M).  The per-view values in the double sum carry the loop's rounding of that map, so this comparison
    # cannot say which of the two is closer to the exact gradient; the oracle comparison below can, at small size.
    P = b1.P
    for name, s1, s2, c0, c1 in zip(b1.NAMES, b1.slices(), b2.slices(), b1.CUTS[:-1], b1.CUTS[1:]):
        if exact:
            assert torch.equal(s1, s2), f"{name}: batched gradients differ from the accumulated single-view ones"
        else:
            ex = b1.exact[c0 * P:c1 * P]
            scale_ = ex.abs().max().item()
            e_seq = (s1.double() - ex).abs().max().item()
            e_bat = (s2.double() - ex).abs().max().item()
            floor = 1e-4 if name in ("_scaling", "_rotation") else 1e-5
            assert e_bat <= max(3.0 * e_seq, floor * scale_), f"{name}: batch {e_bat:.3e}, loop {e_seq:.3e}, scale {scale_:.3e}"
    return out, b2


def test_small_scene_batch_equals_single_views_and_oracle():
    """S-hydrant-1k, 5 cameras at 128x128 (fewer Gaussians than tiles of the batch: the memset path of the ranges)."""
    from gsplat_attack.scenes import make_scene
    from oracle import oracle_r as O
    from util import settings_for, model_inputs
    dev = _dev()
    model, cams, _ = make_scene("hydrant-1k", device=dev, n_views=5)
    g = torch.Generator().manual_seed(5)
    gcs = [torch.randn(3, 128, 128, generator=g).to(dev) for _ in cams]
    out, _ = _check_equal(model, cams, gcs)
    # view 3 of the batch against oracle-R (float64) on its solid pixels
    ref, rcams, _ = make_scene("hydrant-1k", device="cpu", n_views=5)
    st = settings_for(rcams[3], torch.tensor([0.1, 0.2, 0.3]))
    ro = O.rasterize(ref.get_xyz, None, ref.get_opacity, st, shs=ref.get_features, scales=ref.get_scaling,
                     rotations=ref.get_rotation)
    solid = ~ro.fragile_px
    err = (out["render"][3].detach().cpu().double() - ro.color.detach()).abs().amax(dim=0)[solid].max().item()
    assert err <= 1e-4, err
    # the batch's summed gradients against oracle-R differentiated in float64 on all five views (dL/dC zero on each view's
    # fragile pixels, on both sides): BASELINE's 1e-3 per attribute group
    from util import grad_error
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import PipelineParams, render_batch
    leaves = [p_ for p_ in ref.parameters()]
    for p_ in leaves:
        p_.grad = None
    gcs_solid = []
    for v in range(5):
        stv = settings_for(rcams[v], torch.tensor([0.1, 0.2, 0.3]))
        rov = O.rasterize(ref.get_xyz, None, ref.get_opacity, stv, shs=ref.get_features, scales=ref.get_scaling,
                          rotations=ref.get_rotation)
        gk = gcs[v].cpu().double() * (~rov.fragile_px).double()
        (rov.color * gk).sum().backward()
        gcs_solid.append(gk.float().to(dev))
    bucket = D.GradBucket(1000, dev)
    render_batch(cams, model, PipelineParams(skip_objects=True, grad_bucket=bucket),
                 torch.tensor([0.1, 0.2, 0.3], device=dev))["render"].backward(torch.stack(gcs_solid))
    views = bucket.views()
    for name in PARAMS:
        norm, frac = grad_error(views[name], getattr(ref, name).grad)
        assert norm <= 1e-3 and frac <= 0.01, (name, norm, frac)


@pytest.mark.parametrize("P,W,H,B", [(60_000, 640, 360, 4), (20_001, 333, 190, 3), (70_000, 512, 512, 16), (300, 64, 48, 2)])
def test_mid_scenes_ragged_sizes(P, W, H, B):
    """Gaussian counts that are no multiple of anything, image sizes that are no multiple of a tile, 2..16 views,
    split tile lists (segments) at the larger sizes."""
    from gsplat_attack.scenes import make_scene
    dev = _dev()
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=W, height=H, n_views=B)
    g = torch.Generator().manual_seed(P)
    gcs = [torch.randn(3, H, W, generator=g).to(dev) for _ in cams]
    _check_equal(model, cams, gcs)


def test_per_view_backgrounds_and_fov():
    """Backgrounds and fields of view are per view; image size is shared."""
    from gsplat_attack.scenes import make_scene
    from gsplat_attack.cameras import look_at_camera
    dev = _dev()
    model, cams, _ = make_scene("nyc-1M", device=dev, P=30_000, width=320, height=200, n_views=3)
    cams[1] = look_at_camera((25.0, 5.0, 9.0), (0.0, 0.0, 3.0), up=(0.0, 0.0, 1.0), fovx=0.6, width=320, height=200, uid=9,
                             device=dev)
    bgs = [torch.tensor(b, device=dev) for b in ([0.0, 0.0, 0.0], [1.0, 0.5, 0.25], [0.3, 0.3, 0.9, 7.0])]
    g = torch.Generator().manual_seed(11)
    gcs = [torch.randn(3, 200, 320, generator=g).to(dev) for _ in cams]
    _check_equal(model, cams, gcs, bg_list=bgs)


def test_batch_of_one_and_empty_scene():
    import diff_gaussian_rasterization as D
    from gsplat_attack.scenes import make_scene
    from gsplat_attack.renderer import PipelineParams, render, render_batch
    from gsplat_attack.gaussian_model import GaussianModel
    dev = _dev()
    model, cams, _ = make_scene("hydrant-1k", device=dev, n_views=2)
    pipe = PipelineParams(skip_objects=True)
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)
    one = render_batch(cams[:1], model, pipe, bg)
    ref = render(cams[0], model, pipe, bg)
    assert torch.equal(one["render"][0], ref["render"]) and torch.equal(one["radii"][0], ref["radii"])
    # no Gaussians at all: B backgrounds
    z = lambda *s: torch.zeros(*s, device=dev)
    empty = GaussianModel.from_tensors(xyz=z(0, 3), features_dc=z(0, 1, 3), features_rest=z(0, 15, 3), scaling=z(0, 3),
                                       rotation=z(0, 4), opacity=z(0, 1), objects_dc=z(0, 1, 16), device=dev)
    out = render_batch(cams, empty, pipe, bg)
    assert out["render"].shape == (2, 3, 128, 128)
    assert torch.equal(out["render"], bg.view(1, 3, 1, 1).expand(2, 3, 128, 128))
    out["render"].sum().backward()


def test_batch_rejects_mixed_sizes_and_objects():
    import diff_gaussian_rasterization as D
    from gsplat_attack.scenes import make_scene
    from gsplat_attack.renderer import PipelineParams, render_batch, _settings
    dev = _dev()
    model, cams, _ = make_scene("hydrant-1k", device=dev, n_views=2)
    _, cams2, _ = make_scene("hydrant-1k", device=dev, n_views=2, width=64, height=64)
    bg = torch.zeros(3, device=dev)
    with pytest.raises(ValueError):
        render_batch([cams[0], cams2[1]], model, PipelineParams(skip_objects=True), bg)
    with pytest.raises(ValueError):
        render_batch(cams, model, PipelineParams(skip_objects=False), bg)
    pipe = PipelineParams(skip_objects=True)
    sts = [_settings(cams[0], model, pipe, bg, 1.0), _settings(cams2[1], model, pipe, bg, 1.0)]
    with pytest.raises(Exception, match="differs from view 0"):
        D.rasterize_gaussians_raw_batch(model._xyz, None, model._features_dc, model._features_rest, model._opacity,
                                        model._scaling, model._rotation, sts)


def test_fullsize_batch_of_four_equals_single_views():
    """S-nyc-1M at 1080p, B = 4: images, radii and screen-space gradients bit for bit, the gradient bucket equal to the
    four single-view backward passes accumulated in view order."""
    from gsplat_attack.scenes import make_scene
    dev = _dev()
    model, cams, _ = make_scene("nyc-1M", device=dev, n_views=8)
    g = torch.Generator().manual_seed(99)
    gcs = [torch.randn(3, 1080, 1920, generator=g).to(dev) for _ in range(4)]
    _check_equal(model, [cams[i] for i in (0, 3, 5, 6)], gcs)


def test_pgd_attack_batched_matches_per_view_loop():
    """pgd_attack with the rank's views as one batch per iteration against the per-view loop: same images, gradients within
    rounding -> loss histories and stepped parameters agree (all five attribute groups, L2 steps with the norms out of the
    batch's one backward)."""
    from gsplat_attack.scenes import make_scene
    from gsplat_attack.attack import pgd_attack, SurrogateDetector
    dev = _dev()
    model, cams, _ = make_scene("nyc-1M", device=dev, P=40_000, width=480, height=272, n_views=4)
    det = SurrogateDetector().to(dev)
    groups = ("color", "position", "scaling", "rotation", "opacity")
    outs = []
    for batched in (True, False):
        m = model.clone()
        recs = []
        hist = pgd_attack(m, cams, iters=3, groups=groups, loss_fn=det, streams=1, batched=batched, log=recs.append)
        outs.append((hist, {n: getattr(m, n).detach().clone() for n in PARAMS}))
    (h1, p1), (h2, p2) = outs
    assert len(h1) == len(h2) == 3
    for a, b in zip(h1, h2):
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-3), (h1, h2)
    for n in PARAMS:
        d = (p1[n] - p2[n]).abs().max().item()
        assert d <= 2e-4, (n, d)
    # and the batch really ran: the first iteration's loss equals the per-view one exactly (same images, same detector)
    assert h1[0] == pytest.approx(h2[0], rel=1e-6)


@pytest.mark.parametrize("P,W,H,B", [(60_000, 640, 360, 5), (1000, 128, 128, 3)])
def test_per_view_gradients_of_a_batch_equal_the_single_view_backward(P, W, H, B):
    """gsr_backward_raw_batch_views / GradBucketSet: the views share one launch chain and one backward composite, and every
    view's OWN 59 gradient floats per Gaussian are bit for bit those of the single-view call."""
    import diff_gaussian_rasterization as D
    from gsplat_attack.scenes import make_scene
    from gsplat_attack.renderer import PipelineParams, render, render_batch
    dev = _dev()
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=W, height=H, n_views=B)
    g = torch.Generator().manual_seed(B)
    gcs = [torch.randn(3, H, W, generator=g).to(dev) for _ in cams]
    bg = torch.tensor([0.3, 0.2, 0.1], device=dev)
    bset = D.GradBucketSet(B, P, dev)
    bset.flat.fill_(float("nan"))                       # every float of every view's bucket must be written
    out = render_batch(cams, model, PipelineParams(skip_objects=True, grad_bucket=bset), bg)
    out["render"].backward(torch.stack(gcs))
    torch.cuda.synchronize()
    assert bset.used == B and torch.isfinite(bset.flat).all()
    for v, cam in enumerate(cams):
        one = D.GradBucket(P, dev)
        o = render(cam, model, PipelineParams(skip_objects=True, grad_bucket=one), bg)
        o["render"].backward(gcs[v])
        assert torch.equal(out["render"][v].detach(), o["render"].detach())
        assert torch.equal(bset.bucket(v).flat, one.flat), f"view {v}"
        assert torch.equal(out["viewspace_points"].grad[v], o["viewspace_points"].grad)
    # the set's buckets are views of its memory, shaped like the model's parameters
    assert bset.bucket(1).views()["_features_rest"].shape == (P, 15, 3)


@pytest.mark.parametrize("seed", [125, 43])
def test_needles_seen_by_several_views_match_the_oracle_in_loop_and_batch(seed):
    """Found by tests/diag_fuzz_batch.py (round 6): draws with 100:1+ needles where the batch kernel's scale gradient sat
    0.5 % from the single-view kernel's and 30x further from oracle-R.  Cause: the backward kernels recompute the 2D
    covariance (a, b, c) and differentiate its inversion there -- a derivative that cancels by the eigenvalue ratio -- and
    the compiler had contracted a*b + c to fma differently in k_pre_bwd_batch than in k_pre_geom / k_pre_bwd, so the batch
    kernel's (a, b, c) were not the forward's bits.  The projection chain is now evaluated without contraction in every
    kernel (gsr_math.h GSR_FP_STRICT).  Here: both paths within 5e-5 (normwise) of the float64 oracle on the solid pixels,
    and within 5e-5 of each other."""
    import os
    import sys
    import diff_gaussian_rasterization as D
    from gsplat_attack import renderer as R
    from gsplat_attack.gaussian_model import GaussianModel
    from gsplat_attack.renderer import PipelineParams, render
    from oracle import oracle_r as O
    from util import settings_for
    sys.path.insert(0, os.path.dirname(__file__))
    import diag_fuzz_batch as F
    dev = _dev()
    model, cams, bgs, gcs, scale, _ = F.draw(seed, dev)
    P, B = int(model.get_xyz.shape[0]), len(cams)
    ref = GaussianModel.from_tensors(model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation,
                                     model._opacity, device="cpu")
    ref.active_sh_degree = model.active_sh_degree
    solid = []
    for v, cam in enumerate(cams):
        st = settings_for(cam, bgs[v], sh_degree=model.active_sh_degree, scale_modifier=scale, device="cpu")
        ro = O.rasterize(ref.get_xyz, None, ref.get_opacity, st, shs=ref.get_features, scales=ref.get_scaling,
                         rotations=ref.get_rotation)
        gk = gcs[v].cpu().double() * (~ro.fragile_px).double()
        if ro.color.requires_grad:
            (ro.color * gk).sum().backward()
        solid.append(gk.float().to(dev))
    loop = D.GradBucket(P, dev)
    for v, cam in enumerate(cams):
        render(cam, model, PipelineParams(skip_objects=True, grad_bucket=loop), bgs[v], scale)["render"].backward(solid[v])
    bat = D.GradBucket(P, dev)
    pipe = PipelineParams(skip_objects=True)
    sts = [R._settings(c, model, pipe, bgs[v], scale) for v, c in enumerate(cams)]
    vsp = torch.zeros(B, P, 3, device=dev, requires_grad=True)
    image, _ = D.rasterize_gaussians_raw_batch(model._xyz, vsp, model._features_dc, model._features_rest, model._opacity,
                                               model._scaling, model._rotation, sts, grad_bucket=bat)
    image.backward(torch.stack(solid))
    torch.cuda.synchronize()
    for name, gl, gb in zip(loop.NAMES, loop.slices(), bat.slices()):
        go = getattr(ref, name).grad.reshape(-1).double()
        gl, gb = gl.double().cpu(), gb.double().cpu()
        s_ = go.abs().max().item()
        assert ((gl - go).abs().max() / s_).item() <= 5e-5, (name, "loop")
        assert ((gb - go).abs().max() / s_).item() <= 5e-5, (name, "batch")
        assert ((gb - gl).abs().max() / s_).item() <= 5e-5, (name, "batch vs loop")


def test_per_view_gradients_stay_bit_equal_without_segments_on_a_batch_of_many_tiles():
    """Found by tests/diag_fuzz_batch.py with random extension flags (round 6): under GSR_FLAG_NO_SEGMENTS the backward
    splits a tile over two waves when the image has fewer than 4096 tiles -- the rule looked at the batch's tiles (13 x 396
    here), chose one wave per tile for the batch and two for the single view, and K9 summed a different number of partial
    rows per pair: equal within float32 rounding, not bit for bit.  The rule looks at one view's tiles now."""
    import diff_gaussian_rasterization as D
    from gsplat_attack.scenes import make_scene
    from gsplat_attack.renderer import PipelineParams, render, render_batch
    dev = _dev()
    P, W, H, B = 3000, 338, 274, 13
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=W, height=H, n_views=B)
    g = torch.Generator().manual_seed(13)
    gcs = [torch.randn(3, H, W, generator=g).to(dev) for _ in cams]
    bg = torch.tensor([0.3, 0.2, 0.1], device=dev)
    for flags in (D.FLAG_NO_SEGMENTS, D.FLAG_NO_SEGMENTS | D.FLAG_FWD_SHARED | D.flag_tile_map(0), 0):
        with D.extra_flags(flags):
            bset = D.GradBucketSet(B, P, dev)
            out = render_batch(cams, model, PipelineParams(skip_objects=True, grad_bucket=bset), bg)
            out["render"].backward(torch.stack(gcs))
            for v in (0, 7, 12):
                one = D.GradBucket(P, dev)
                o = render(cams[v], model, PipelineParams(skip_objects=True, grad_bucket=one), bg)
                o["render"].backward(gcs[v])
                torch.cuda.synchronize()
                assert torch.equal(out["render"][v].detach(), o["render"].detach()), (flags, v)
                assert torch.equal(bset.bucket(v).flat, one.flat), (flags, v)
