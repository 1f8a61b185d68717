"""Kept contexts of a BATCH of views (gsr_ctx_rerender on a gsr_forward_raw_batch context; RenderCache behind render_batch).

A colour attack on several views (reference attack.py:476-494 with the colour rules of :25-49) renders the same B cameras
iteration after iteration with the same means / scales / rotations / opacities: the batch's binning -- one scan, two sorts,
one emission, one schedule for all B views -- is kept, and a later render of the batch is the batch's colour kernel (every SH
row read once for all views) plus ONE compositor launch over the kept lists.  Everything here is BIT equality against the
uncached batch: images, radii, every gradient, PGD histories, success flags, stepped parameters.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

COL = ("_features_dc", "_features_rest")
ALL = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _scene(P, W, H, n_views, key="hydrant-full"):
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene(key, device=dev, P=P, width=W, height=H, n_views=n_views)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    return dev, model, cams, bg


def _fwd_bwd(cams, model, pipe, bg, gcs, names):
    from gsplat_attack.renderer import render_batch
    model.zero_grad()
    out = render_batch(cams, model, pipe, bg)
    out["render"].backward(gcs)
    torch.cuda.synchronize()
    vs = out["viewspace_points"]
    return (out["render"].detach().clone(), out["radii"].clone(),
            {n: getattr(model, n).grad.detach().clone() for n in names},
            None if vs is None or vs.grad is None else vs.grad.detach().clone())


def _step_colours(model, gen, dev):
    with torch.no_grad():                            # in place, like the fused PGD update
        model._features_dc.add_(0.05 * torch.randn(model._features_dc.shape, generator=gen).to(dev))
        model._features_rest.add_(0.02 * torch.randn(model._features_rest.shape, generator=gen).to(dev))


@pytest.mark.parametrize("color_only", [True, False])
@pytest.mark.parametrize("P,W,H,B", [(30000, 320, 240, 3), (120000, 480, 270, 5), (300, 64, 48, 2)])
def test_batch_rerender_is_bit_equal_to_a_fresh_batch(color_only, P, W, H, B):
    """Three colour steps on a batch: every cached batch render + backward equals the uncached one bit for bit -- with the
    geometry frozen (colour-only backward, d colour / d direction skipped) and with every attribute differentiated."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams
    dev, model, cams, bg = _scene(P, W, H, B)
    names = COL if color_only else ALL
    if color_only:
        for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
            getattr(model, n).requires_grad_(False)
    plain = PipelineParams(skip_objects=True, viewspace_grad=not color_only)
    cached = PipelineParams(skip_objects=True, viewspace_grad=not color_only, render_cache=RenderCache())
    gen = torch.Generator().manual_seed(7)
    gcs = torch.randn(B, 3, H, W, generator=gen).to(dev)
    for it in range(4):
        want = _fwd_bwd(cams, model, plain, bg, gcs, names)
        got = _fwd_bwd(cams, model, cached, bg, gcs, names)
        assert torch.equal(want[0], got[0]), ("image", it)
        assert torch.equal(want[1], got[1]), ("radii", it)
        for n in names:
            assert torch.equal(want[2][n], got[2][n]), (n, it)
        if not color_only:
            assert torch.equal(want[3], got[3]), ("viewspace", it)
        _step_colours(model, gen, dev)
    c = cached.render_cache
    assert c.misses == 1 and c.hits == 3


def test_batch_rerender_equals_the_single_view_renders():
    """The cached batch's images are the single-view renders of its cameras, bit for bit, after a colour step."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams, render, render_batch
    dev, model, cams, bg = _scene(50000, 400, 300, 4)
    cached = PipelineParams(skip_objects=True, render_cache=RenderCache())
    plain = PipelineParams(skip_objects=True)
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        render_batch(cams, model, cached, bg)
        _step_colours(model, gen, dev)
        got = render_batch(cams, model, cached, bg)["render"]
        for v, cam in enumerate(cams):
            assert torch.equal(got[v], render(cam, model, plain, bg)["render"]), v
    assert cached.render_cache.hits == 1


def test_backgrounds_cameras_and_geometry_behind_a_batch_key():
    """A kept batch never serves stale state: new background VALUES in the same tensor are read again, other background
    tensors are taken, and a changed camera set, an edited geometry tensor or another scale modifier take the full forward."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams, render_batch
    dev, model, cams, bg = _scene(20000, 256, 192, 4)
    cache = RenderCache()
    cached = PipelineParams(skip_objects=True, render_cache=cache)
    plain = PipelineParams(skip_objects=True)

    def both(cs, b, scale=1.0):
        with torch.no_grad():
            return (render_batch(cs, model, cached, b, scale)["render"].clone(),
                    render_batch(cs, model, plain, b, scale)["render"].clone())
    a, w = both(cams[:3], bg)
    assert torch.equal(a, w) and cache.misses == 1
    bg.copy_(torch.tensor([0.7, 0.1, 0.4], device=dev))             # same tensor, new values
    a, w = both(cams[:3], bg)
    assert torch.equal(a, w) and cache.hits == 1
    bg2 = torch.tensor([0.3, 0.9, 0.2], device=dev)                  # another tensor
    a, w = both(cams[:3], bg2)
    assert torch.equal(a, w) and cache.hits == 2
    a, w = both(cams[:3], bg)                                        # and back (the context reads a packed copy now)
    assert torch.equal(a, w) and cache.hits == 3
    a, w = both(cams[1:], bg)                                        # other cameras: another key
    assert torch.equal(a, w) and cache.misses == 2
    a, w = both(cams[:3], bg, 1.5)                                   # same key, another scale modifier: full forward
    assert torch.equal(a, w) and cache.misses == 3
    with torch.no_grad():
        model._xyz.add_(0.01)                                        # stepped geometry: full forward
    a, w = both(cams[:3], bg, 1.5)
    assert torch.equal(a, w) and cache.misses == 4
    a, w = both(cams[:3], bg, 1.5)
    assert torch.equal(a, w) and cache.hits == 4


def test_a_batch_key_waiting_for_its_backward_is_left_alone():
    """Two live differentiable renders of one batch key: the second takes a context of its own, both backward passes give
    the uncached gradients."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams, render_batch
    dev, model, cams, bg = _scene(15000, 200, 160, 3)
    for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
        getattr(model, n).requires_grad_(False)
    cache = RenderCache()
    cached = PipelineParams(skip_objects=True, viewspace_grad=False, render_cache=cache)
    plain = PipelineParams(skip_objects=True, viewspace_grad=False)
    gcs = torch.randn(3, 3, 160, 200, generator=torch.Generator().manual_seed(2)).to(dev)
    want = _fwd_bwd(cams, model, plain, bg, gcs, COL)
    model.zero_grad()
    o1 = render_batch(cams, model, cached, bg)
    o2 = render_batch(cams, model, cached, bg)                       # the key is busy: bypassed
    assert cache.bypassed == 1
    o2["render"].backward(gcs)
    g2 = {n: getattr(model, n).grad.detach().clone() for n in COL}
    model.zero_grad()
    o1["render"].backward(gcs)
    for n in COL:
        assert torch.equal(g2[n], want[2][n]) and torch.equal(getattr(model, n).grad, want[2][n]), n


@pytest.mark.parametrize("with_background", [True, False])
def test_batched_colour_attack_with_kept_contexts_equals_the_uncached_batched_loop(with_background, tmp_path):
    """pgd_attack(groups=("color",), batched=True) with cache_binning on and off: bit-equal history, success flags, stepped
    parameters and saved model (the fused L2 norms come out of the batch's backward in both)."""
    from gsplat_attack.attack import pgd_attack
    dev, model, cams, bg = _scene(8000, 160, 128, 3)
    base, back = model.clone(), model.clone()
    runs = []
    for cache_on in (False, True):
        m = base.clone()
        calls = []

        def success(im, i, calls=calls):
            calls.append(float(im.double().sum()))
            return len(calls) > 3 * len(cams)
        recs = []
        path = str(tmp_path / f"m_{cache_on}.ply")
        hist = pgd_attack(m, cams, iters=6, groups=("color",), success_fn=success,
                          background=back if with_background else None, log=recs.append, save_path=path,
                          cache_binning=cache_on, batched=True)
        torch.cuda.synchronize()
        runs.append((hist, [r.get("successes") for r in recs], calls,
                     {n: getattr(m, n).detach().clone() for n in COL}, open(path, "rb").read()))
    (h0, f0, c0, p0, s0), (h1, f1, c1, p1, s1) = runs
    assert len(h0) == 4 and h0 == h1 and f0 == f1 and c0 == c1 and s0 == s1
    for n in COL:
        assert torch.equal(p0[n], p1[n]), n


def test_c_abi_batch_rerender_arguments():
    """gsr_ctx_rerender on a batch context: object channels and second-segment coefficients are refused; bg is [B,3]."""
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import _settings, PipelineParams
    lib = D._load()
    dev, model, cams, bg = _scene(3000, 96, 64, 2)
    H, W, B = 64, 96, 2
    packs = [D._SettingsPack(_settings(c, model, PipelineParams(), bg, 1.0), dev) for c in cams]
    carr = (D._CSettings * B)()
    for v, pk in enumerate(packs):
        carr[v] = pk.c
    P = model._xyz.shape[0]
    raw = [t.detach().contiguous() for t in (model._xyz, model._features_dc, model._features_rest, model._opacity,
                                              model._scaling, model._rotation)]
    color = torch.empty(B, 3, H, W, device=dev)
    radii = torch.empty(B, P, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    h, n = ctypes.c_void_p(None), ctypes.c_int64(0)
    ptr = D._ptr
    assert lib.gsr_forward_raw_batch(carr, B, P, *[ptr(t) for t in raw], ptr(color), ptr(radii), ctypes.byref(h),
                                     ctypes.byref(n), stream) == 0
    two = torch.empty_like(color)
    objs = torch.empty(16, H, W, device=dev)
    assert lib.gsr_ctx_rerender(h, None, None, None, None, None, ptr(two), ptr(objs), 0, stream) == 1
    assert b"batch" in lib.gsr_last_error()
    assert lib.gsr_ctx_rerender(h, None, None, ptr(raw[1]), None, None, ptr(two), None, 0, stream) == 1
    assert lib.gsr_ctx_rerender(h, None, None, None, None, None, ptr(two), None, 0, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(two, color)
    # per-view backgrounds through the [B,3] argument: each view's image moves by (1 - alpha) * (new - old)
    bgs = torch.tensor([[0.9, 0.8, 0.7], [0.0, 0.5, 1.0]], device=dev)
    assert lib.gsr_ctx_rerender(h, None, None, None, None, ptr(bgs), ptr(two), None, 0, stream) == 0
    torch.cuda.synchronize()
    for v in range(B):
        pk = D._SettingsPack(_settings(cams[v], model, PipelineParams(), bgs[v].clone(), 1.0), dev)
        one = torch.empty(3, H, W, device=dev)
        r1 = torch.empty(P, dtype=torch.int32, device=dev)
        assert lib.gsr_forward_raw(ctypes.byref(pk.c), P, ptr(raw[0]), ptr(raw[1]), ptr(raw[2]), None, ptr(raw[3]), ptr(raw[4]),
                                   ptr(raw[5]), ptr(one), None, ptr(r1), None, ctypes.byref(n), stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(two[v], one), v
    # a geometry backward after a colour-gradients-only re-render is refused, a colour one is served
    assert lib.gsr_ctx_rerender(h, None, None, None, None, None, ptr(two), None, 1, stream) == 0
    g = torch.zeros(B, 3, H, W, device=dev)
    d_x, d_dc, d_rest = torch.empty(P, 3, device=dev), torch.empty(P, 1, 3, device=dev), torch.empty(P, 15, 3, device=dev)
    assert lib.gsr_backward_raw_batch_into(h, ptr(g), ptr(d_x), None, ptr(d_dc), ptr(d_rest), None, None, None, 0, stream) == 4
    assert b"COLOR_GRADS_ONLY" in lib.gsr_last_error()
    assert lib.gsr_backward_raw_batch_into(h, ptr(g), None, None, ptr(d_dc), ptr(d_rest), None, None, None, 0, stream) == 0
    torch.cuda.synchronize()
    lib.gsr_ctx_free(h)


def test_full_size_batch_rerender_on_the_benchmark_scene():
    """S-nyc-1M at 1080p, four ring cameras: cached == fresh for the images and the SH gradients after a colour step."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams
    dev, model, cams, bg = _scene(None, None, None, 4, key="nyc-1M")
    for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
        getattr(model, n).requires_grad_(False)
    H, W = cams[0].image_height, cams[0].image_width
    plain = PipelineParams(skip_objects=True, viewspace_grad=False)
    cached = PipelineParams(skip_objects=True, viewspace_grad=False, render_cache=RenderCache())
    gcs = torch.randn(4, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    for it in range(2):
        want = _fwd_bwd(cams, model, plain, bg, gcs, COL)
        got = _fwd_bwd(cams, model, cached, bg, gcs, COL)
        assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
        for n in COL:
            assert torch.equal(want[2][n], got[2][n]), n
        with torch.no_grad():
            model._features_dc.add_(0.1)
            model._features_rest.mul_(1.05)
    assert cached.render_cache.hits == 1


@pytest.mark.parametrize("Pa,Pb,W,H,B", [(20000, 30000, 320, 240, 3), (300, 70000, 400, 300, 5), (50000, 100, 256, 256, 2)])
def test_pair_batch_equals_the_per_camera_pair_renders(Pa, Pb, W, H, B):
    """render_pair_batch (gsr_forward_raw2_batch): target + background from B cameras through one launch chain -- images and
    radii bit for bit render_pair's; through a cache, over colour steps of the target and then of the background too."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams, render_pair, render_pair_batch
    dev, model, cams, bg = _scene(Pa + Pb, W, H, B)
    from gsplat_attack.gaussian_model import GaussianModel

    def part(sl):
        return GaussianModel.from_tensors(model._xyz[sl].detach().clone(), model._features_dc[sl].detach().clone(),
                                          model._features_rest[sl].detach().clone(), model._scaling[sl].detach().clone(),
                                          model._rotation[sl].detach().clone(), model._opacity[sl].detach().clone(),
                                          model._objects_dc[sl].detach().clone(), sh_degree=model.max_sh_degree,
                                          device=dev, requires_grad=False)
    a, b = part(slice(0, Pa)), part(slice(Pa, Pa + Pb))
    plain = PipelineParams(skip_objects=True)
    cache = RenderCache()
    cached = PipelineParams(skip_objects=True, render_cache=cache)
    gen = torch.Generator().manual_seed(11)
    for it in range(5):
        want = [render_pair(c, a, b, plain, bg) for c in cams]
        got = render_pair_batch(cams, a, b, plain, bg)
        got_c = render_pair_batch(cams, a, b, cached, bg)
        for v in range(B):
            assert torch.equal(got["render"][v], want[v]["render"]), ("image", it, v)
            assert torch.equal(got["radii"][v], want[v]["radii"]), ("radii", it, v)
            assert torch.equal(got_c["render"][v], want[v]["render"]), ("cached image", it, v)
            assert torch.equal(got_c["radii"][v], want[v]["radii"]), ("cached radii", it, v)
        _step_colours(a, gen, dev)
        if it >= 2:
            _step_colours(b, gen, dev)                # the frozen background is not frozen any more: its colours are redone
    assert cache.misses == 1 and cache.hits == 4


@pytest.mark.parametrize("groups", [("color",), ("color", "position", "opacity")])
def test_attack_with_batched_success_renders_equals_the_per_camera_checks(groups, tmp_path):
    """pgd_attack with a background model: the success renders of the rank's views through render_pair_batch against one
    render_pair per camera (PipelineParams.batched_checks): bit-equal history, flags, what the success function saw."""
    from gsplat_attack.attack import pgd_attack
    from gsplat_attack.renderer import PipelineParams
    dev, model, cams, bg = _scene(8000, 160, 128, 3)
    base, back = model.clone(), model.clone()
    with torch.no_grad():
        back._xyz.add_(torch.tensor([0.05, -0.02, 0.1], device=dev))
    runs = []
    for batched_checks in (False, True):
        m = base.clone()
        calls = []

        def success(im, i, calls=calls):
            calls.append(float(im.double().sum()))
            return len(calls) > 3 * len(cams)
        recs = []
        pipe = PipelineParams(skip_objects=True)
        pipe.batched_checks = batched_checks
        hist = pgd_attack(m, cams, iters=6, groups=groups, success_fn=success, background=back, log=recs.append, pipe=pipe)
        torch.cuda.synchronize()
        runs.append((hist, [r.get("successes") for r in recs], calls, {n: getattr(m, n).detach().clone() for n in ALL}))
    (h0, f0, c0, p0), (h1, f1, c1, p1) = runs
    assert len(h0) == 4 and h0 == h1 and f0 == f1 and c0 == c1
    for n in ALL:
        assert torch.equal(p0[n], p1[n]), n


def test_full_size_pair_batch_on_the_benchmark_scene():
    """S-nyc-1M plus a shifted copy of itself as the background (2 M Gaussians per view), four ring cameras at 1080p: the pair
    batch's images and radii are bit for bit the four render_pair calls'."""
    from gsplat_attack.renderer import PipelineParams, render_pair, render_pair_batch
    dev, model, cams, bg = _scene(None, None, None, 4, key="nyc-1M")
    back = model.clone()
    with torch.no_grad():
        back._xyz.add_(torch.tensor([0.3, 0.1, -0.2], device=dev))
        pipe = PipelineParams(skip_objects=True)
        got = render_pair_batch(cams, model, back, pipe, bg)
        for v, c in enumerate(cams):
            want = render_pair(c, model, back, pipe, bg)
            assert torch.equal(got["render"][v], want["render"]) and torch.equal(got["radii"][v], want["radii"]), v
