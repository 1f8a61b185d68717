"""Kept contexts for frozen-geometry renders (gsr_ctx_rerender, RenderCache, pgd_attack(cache_binning=...)).

A colour attack (reference attack.py:25-49) steps the SH coefficients and nothing else, so every iteration renders the
same cameras with the same means / scales / rotations / opacities.  A render through a RenderCache re-uses the binning of
the camera's previous render and runs the colour kernel + the compositor only.  Everything here is BIT equality against
the uncached path: image, radii, every gradient, PGD histories, success flags, saved parameters.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(P=30000, W=320, H=240, n_views=3, key="hydrant-full"):
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene(key, device=dev, P=P, width=W, height=H, n_views=n_views)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    return dev, model, cams, bg


def _grads(model, names):
    return {n: getattr(model, n).grad.detach().clone() for n in names if getattr(model, n).grad is not None}


COL = ("_features_dc", "_features_rest")
ALL = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _fwd_bwd(cam, model, pipe, bg, gc, names):
    from gsplat_attack.renderer import render
    model.zero_grad()
    out = render(cam, model, pipe, bg)
    out["render"].backward(gc)
    torch.cuda.synchronize()
    return out["render"].detach().clone(), out["radii"].clone(), _grads(model, names)


@pytest.mark.parametrize("color_only", [True, False])
@pytest.mark.parametrize("P,W,H", [(30000, 320, 240), (200000, 480, 270)])
def test_rerender_is_bit_equal_to_a_fresh_forward(color_only, P, W, H):
    """Three colour steps on two cameras: every cached render + backward equals the uncached one bit for bit -- with the
    geometry tensors frozen (colour-only backward kernels, d colour / d direction skipped) and with all of them
    differentiated (full backward from a re-rendered context)."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams
    dev, model, cams, bg = _scene(P, W, H, n_views=2)
    names = COL if color_only else ALL
    if color_only:
        for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
            getattr(model, n).requires_grad_(False)
    plain = PipelineParams(skip_objects=True, viewspace_grad=not color_only)
    cached = PipelineParams(skip_objects=True, viewspace_grad=not color_only, render_cache=RenderCache())
    gen = torch.Generator().manual_seed(5)
    gcs = [torch.randn(3, H, W, generator=gen).to(dev) for _ in cams]
    for it in range(4):
        for cam, gc in zip(cams, gcs):
            want = _fwd_bwd(cam, model, plain, bg, gc, names)
            got = _fwd_bwd(cam, model, cached, bg, gc, names)
            assert torch.equal(want[0], got[0]), ("image", it)
            assert torch.equal(want[1], got[1]), ("radii", it)
            assert set(want[2]) == set(got[2]) == set(names)
            for n in names:
                assert torch.equal(want[2][n], got[2][n]), (n, it)
        with torch.no_grad():                        # the colour step: in place, like the fused PGD update
            model._features_dc.add_(0.05 * torch.randn(model._features_dc.shape, generator=gen).to(dev))
            model._features_rest.add_(0.02 * torch.randn(model._features_rest.shape, generator=gen).to(dev))
    c = cached.render_cache
    assert c.misses == len(cams) and c.hits == 3 * len(cams)


def test_changed_geometry_camera_or_settings_take_the_full_forward():
    """The cache never serves a stale binning: an in-place edit of a geometry tensor, another camera behind the key's
    object id, another scale modifier or background all give the uncached result."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams, bg = _scene(20000, 256, 192, n_views=2)
    cache = RenderCache()
    plain, cached = PipelineParams(skip_objects=True), PipelineParams(skip_objects=True, render_cache=cache)
    cam = cams[0]

    def same(**kw):
        with torch.no_grad():
            a = render(cam, model, plain, bg, **kw)["render"]
            b = render(cam, model, cached, bg, **kw)["render"]
        return torch.equal(a, b)
    assert same() and same()
    assert (cache.misses, cache.hits) == (1, 1)
    for n, amp in (("_xyz", 0.01), ("_opacity", 0.3), ("_scaling", 0.1), ("_rotation", 0.1)):
        with torch.no_grad():
            getattr(model, n).add_(amp * torch.randn_like(getattr(model, n)))
        h = cache.hits
        assert same(), n
        assert cache.hits == h, n                   # a miss: the kept context was replaced
        assert same() and cache.hits == h + 1
    # another background: still a hit (bg is read by the compositor alone), other bits
    h = cache.hits
    bg2 = torch.tensor([0.9, 0.0, 0.4], device=dev)
    with torch.no_grad():
        a = render(cam, model, plain, bg2)["render"]
        b = render(cam, model, cached, bg2)["render"]
    assert torch.equal(a, b) and cache.hits == h + 1
    # another scale modifier: a miss
    assert same(scaling_modifier=1.3) and cache.hits == h + 1
    # the camera's tensors replaced in place of the old ones under the same Python object
    other = cams[1]
    cam.world_view_transform, cam.full_proj_transform, cam.camera_center = (
        other.world_view_transform.clone(), other.full_proj_transform.clone(), other.camera_center.clone())
    h = cache.hits
    assert same() and cache.hits == h
    assert same() and cache.hits == h + 1
    # ... and edited in place (same objects, same storage, another version)
    third = cams[0] if cams[0] is not cam else cams[1]
    with torch.no_grad():
        cam.world_view_transform.copy_(third.world_view_transform * 1.0)
        cam.full_proj_transform.mul_(1.0)
    h = cache.hits
    assert same() and cache.hits == h


def test_a_key_waiting_for_its_backward_is_left_alone():
    """Two renders of one camera before either backward (a batched loss with a repeated view): the second takes the full
    forward with a context of its own, both backwards give the uncached gradients.  Only a SECOND backward through a graph
    whose context has been rendered again since raises."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams, bg = _scene(5000, 128, 96, n_views=1)
    cache = RenderCache()
    plain, pipe = PipelineParams(skip_objects=True), PipelineParams(skip_objects=True, render_cache=cache)
    gen = torch.Generator().manual_seed(2)
    g1, g2 = (torch.randn(3, 96, 128, generator=gen).to(dev) for _ in range(2))
    model.zero_grad()
    render(cams[0], model, plain, bg)["render"].backward(g1 + g2)
    want = _grads(model, ALL)
    model.zero_grad()
    first = render(cams[0], model, pipe, bg)["render"]
    second = render(cams[0], model, pipe, bg)["render"]
    assert cache.bypassed == 1 and cache.hits == 0
    first.backward(g1)
    second.backward(g2)
    got = _grads(model, ALL)
    for n in ALL:
        scale = want[n].abs().max().item() + 1e-20
        assert (want[n] - got[n]).abs().max().item() <= 2e-6 * scale, n      # (g1 + g2) vs two backwards: rounding only
    third = render(cams[0], model, pipe, bg)["render"]          # both are differentiated: the kept context serves again
    assert cache.hits == 1
    third.sum().backward(retain_graph=True)
    render(cams[0], model, pipe, bg)["render"].sum().backward()
    with pytest.raises(RuntimeError, match="rendered again"):
        third.sum().backward()


@pytest.mark.parametrize("objects", [False, True])
def test_cached_pair_render_equals_the_uncached_one(objects):
    """render_pair (target + frozen background as one scene, reference attack.py:513-530) through the cache, while the
    target's colours are stepped."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams, render_pair
    dev, model, cams, bg = _scene(12000, 256, 192, n_views=2)
    back = model.clone()
    with torch.no_grad():
        back._xyz.add_(torch.tensor([0.4, 0.0, 0.2], device=dev))
    cache = RenderCache()
    plain = PipelineParams(skip_objects=not objects)
    cached = PipelineParams(skip_objects=not objects, render_cache=cache)
    for it in range(4):
        for cam in cams:
            a = render_pair(cam, model, back, plain, bg)
            b = render_pair(cam, model, back, cached, bg)
            assert torch.equal(a["render"], b["render"]) and torch.equal(a["radii"], b["radii"])
            if objects:
                assert torch.equal(a["render_object"], b["render_object"])
        with torch.no_grad():
            model._features_dc.mul_(0.9)
            model._features_rest.add_(0.01)
            if it == 1:                               # the background's colours too, once: its colour words are redone
                back._features_dc.add_(0.05)
    assert cache.hits == 3 * len(cams) and cache.misses == len(cams)


@pytest.mark.parametrize("streams,with_background", [(1, True), (3, True), (2, False)])
def test_colour_attack_with_kept_contexts_equals_the_plain_loop(streams, with_background, tmp_path):
    """pgd_attack(groups=("color",)) with cache_binning on and off: bit-equal history, success flags, parameters and saved
    model; the success check (pair render, or the target alone) goes through kept contexts as well."""
    from gsplat_attack.attack import pgd_attack
    dev, model, cams, bg = _scene(8000, 160, 128, n_views=3)
    base, back = model.clone(), model.clone()
    runs = []
    for cache_on in (False, True):
        m = base.clone()
        calls = []

        def success(im, i, calls=calls):
            calls.append(float(im.double().sum()))
            return len(calls) > 3 * len(cams)         # fooled from the fourth iteration on
        recs = []
        path = str(tmp_path / f"m_{cache_on}.ply")
        # (batched=False: without kept contexts the rank's views would go through one launch chain per iteration, whose
        # summed gradient equals the per-view accumulation within rounding, not bit for bit -- tests/test_gpu_batch.py)
        hist = pgd_attack(m, cams, iters=6, groups=("color",), streams=streams, success_fn=success,
                          background=back if with_background else None, log=recs.append, save_path=path,
                          cache_binning=cache_on, batched=False)
        torch.cuda.synchronize()
        runs.append((hist, [r.get("successes") for r in recs], calls,
                     {n: getattr(m, n).detach().clone() for n in COL}, open(path, "rb").read()))
    (h0, f0, c0, p0, s0), (h1, f1, c1, p1, s1) = runs
    assert len(h0) == 4 and h0 == h1 and f0 == f1 and c0 == c1 and s0 == s1
    for n in COL:
        assert torch.equal(p0[n], p1[n]), n


def test_attack_on_all_groups_makes_no_cache():
    """Geometry is stepped: pgd_attack does not create a cache, and one handed in never hits."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.attack import pgd_attack
    from gsplat_attack.renderer import PipelineParams
    dev, model, cams, bg = _scene(6000, 128, 96, n_views=2)
    cache = RenderCache()
    m0, m1 = model.clone(), model.clone()
    # (batched=False on both sides: a pipe that carries a cache keeps the per-view loop, and the comparison is bit for bit)
    h0 = pgd_attack(m0, cams, iters=3, groups=("color", "position"), streams=1, batched=False)
    h1 = pgd_attack(m1, cams, iters=3, groups=("color", "position"), streams=1, batched=False,
                    pipe=PipelineParams(skip_objects=True, render_cache=cache))
    assert h0 == h1 and cache.hits == 0
    assert torch.equal(m0._xyz, m1._xyz)


def test_c_abi_refusals():
    """gsr_backward* on a re-render-only context, gsr_ctx_rerender on a context without a colour stage, geometry
    gradients after a colour-gradients-only re-render: error codes, not results."""
    import diff_gaussian_rasterization as D
    lib = D._load()
    dev, model, cams, bg = _scene(3000, 96, 64, n_views=1)
    H, W = 64, 96
    from gsplat_attack.renderer import _settings, PipelineParams
    rs = _settings(cams[0], model, PipelineParams(), bg, 1.0)
    pack = D._SettingsPack(rs, dev)
    P = model._xyz.shape[0]
    raw = [model._xyz, model._features_dc, model._features_rest, None, model._opacity, model._scaling, model._rotation]
    raw = [None if t is None else t.detach().contiguous() for t in raw]
    color = torch.empty(3, H, W, device=dev)
    radii = torch.empty(2 * P, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    h = ctypes.c_void_p(None)
    n = ctypes.c_int64(0)
    ptr = D._ptr
    rc = lib.gsr_forward_raw2_keep(ctypes.byref(pack.c), P, *[ptr(t) for t in raw], P, *[ptr(t) for t in raw], ptr(color), None,
                                   ptr(radii), ctypes.byref(h), ctypes.byref(n), stream)
    assert rc == 0 and h.value
    g = torch.zeros(3, H, W, device=dev)
    d_dc, d_rest = torch.empty(P, 1, 3, device=dev), torch.empty(P, 15, 3, device=dev)
    rc = lib.gsr_backward_raw(h, ptr(g), None, None, None, ptr(d_dc), ptr(d_rest), None, None, None, None, stream)
    assert rc == 4 and b"re-render only" in lib.gsr_last_error()
    two = color.clone()
    assert lib.gsr_ctx_rerender(h, None, None, None, None, None, ptr(two), None, 0, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(two, color)
    lib.gsr_ctx_free(h)
    # a single-segment context: geometry backward refused after a colour-gradients-only re-render, fine after a full one
    h = ctypes.c_void_p(None)
    rc = lib.gsr_forward_raw(ctypes.byref(pack.c), P, *[ptr(t) for t in raw], ptr(color), None, ptr(radii), ctypes.byref(h),
                             ctypes.byref(n), stream)
    assert rc == 0
    assert lib.gsr_ctx_rerender(h, None, None, ptr(raw[1]), None, None, ptr(two), None, 0, stream) == 1   # no second segment
    assert lib.gsr_ctx_rerender(h, None, None, None, None, None, ptr(two), None, 1, stream) == 0
    d_x = torch.empty(P, 3, device=dev)
    rc = lib.gsr_backward_raw(h, ptr(g), None, ptr(d_x), None, ptr(d_dc), ptr(d_rest), None, None, None, None, stream)
    assert rc == 4 and b"COLOR_GRADS_ONLY" in lib.gsr_last_error()
    assert lib.gsr_backward_raw(h, ptr(g), None, None, None, ptr(d_dc), ptr(d_rest), None, None, None, None, stream) == 0
    assert lib.gsr_ctx_rerender(h, None, None, None, None, None, ptr(two), None, 0, stream) == 0
    assert lib.gsr_backward_raw(h, ptr(g), None, ptr(d_x), None, ptr(d_dc), ptr(d_rest), None, None, None, None, stream) == 0
    torch.cuda.synchronize()
    lib.gsr_ctx_free(h)
    # precomputed colours: there is no colour stage to run again
    cols = torch.rand(P, 3, device=dev)
    op, sc, ro = torch.sigmoid(raw[4]).contiguous(), torch.exp(raw[5]).contiguous(), torch.nn.functional.normalize(raw[6]).contiguous()
    h = ctypes.c_void_p(None)
    rc = lib.gsr_forward(ctypes.byref(pack.c), P, 0, ptr(raw[0]), None, None, ptr(cols), ptr(op), ptr(sc), ptr(ro), None, ptr(color),
                         None, ptr(radii), ctypes.byref(h), ctypes.byref(n), stream)
    assert rc == 0
    assert lib.gsr_ctx_rerender(h, None, None, None, None, None, ptr(two), None, 0, stream) == 1
    lib.gsr_ctx_free(h)


def test_full_size_rerender_bit_equal_on_the_benchmark_scene():
    """S-nyc-1M at 1080p (split tile lists, boundary records, two waves per tile forward): cached == fresh, image and SH
    gradients, after a colour step; and the cached forward is the shorter one."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams
    dev, model, cams, bg = _scene(None, None, None, n_views=2, key="nyc-1M")
    for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
        getattr(model, n).requires_grad_(False)
    H, W = cams[0].image_height, cams[0].image_width
    plain = PipelineParams(skip_objects=True, viewspace_grad=False)
    cached = PipelineParams(skip_objects=True, viewspace_grad=False, render_cache=RenderCache())
    gc = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    for it in range(2):
        for cam in cams:
            want = _fwd_bwd(cam, model, plain, bg, gc, COL)
            got = _fwd_bwd(cam, model, cached, bg, gc, COL)
            assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
            for n in COL:
                assert torch.equal(want[2][n], got[2][n]), n
        with torch.no_grad():
            model._features_dc.add_(0.1)
            model._features_rest.mul_(1.05)
    assert cached.render_cache.hits == len(cams)


def test_overflowed_async_forward_does_not_poison_its_cache_key():
    """under FLAG_ASYNC_COUNT a forward that overflows its guessed pair capacity returns rc 0 with a NaN image,
    and its context lands in the RenderCache.  The next render of the key must not fail on it forever: gsr_ctx_rerender
    reports the overflow, the entry is dropped, and the SAME call takes the full forward (which counts synchronously
    again); the render after that is an ordinary hit."""
    import diff_gaussian_rasterization as D
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams, bg = _scene(50000, 640, 360, n_views=1, key="nyc-1M")
    cam = cams[0]
    plain = PipelineParams(skip_objects=True)
    with torch.no_grad():
        small = render(cam, model, plain, bg)["render"].clone()           # seeds the capacity table for this (P, H, W)
    assert torch.isfinite(small).all()
    cache = RenderCache()
    cached = PipelineParams(skip_objects=True, render_cache=cache)
    try:
        D.set_flags(D.FLAG_ASYNC_COUNT)
        with torch.no_grad():
            # four times larger splats: far more pairs than 1.25 x the 1x count + 64K the buffers are sized for
            first = render(cam, model, cached, bg, 4.0)["render"].clone()
            assert torch.isnan(first).all() and len(cache.entries) == 1
            second = render(cam, model, cached, bg, 4.0)["render"].clone()  # hit -> overflow -> dropped -> full forward
            assert cache.dropped_overflow == 1 and len(cache.entries) == 1
            assert torch.isfinite(second).all()
            hits = cache.hits
            third = render(cam, model, cached, bg, 4.0)["render"].clone()   # an ordinary hit on the replaced entry
        assert cache.dropped_overflow == 1 and cache.hits == hits + 1
    finally:
        D.set_flags(0)
    with torch.no_grad():
        want = render(cam, model, plain, bg, 4.0)["render"].clone()       # counted synchronously, no cache: the right image
    assert torch.equal(second, want) and torch.equal(third, want)
