"""BASELINE.json's full-size configurations (too large for oracle-R) checked through size-independent properties of
the path: bitwise reproducibility, linearity of the backward in dL/dC, background linearity against the exported
final transmittance, per-tile depth ordering of the binned pairs, equality of the culled / un-culled and fused /
un-fused paths."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(key, n_views=1, **kw):
    import diff_gaussian_rasterization as D
    from gsplat_attack.scenes import make_scene
    assert torch.cuda.is_available()
    D._load()
    dev = torch.device("cuda:0")
    model, cams, spec = make_scene(key, device=dev, n_views=n_views, **kw)
    return D, dev, model, cams


def _fwd_bwd(model, cam, bg, gc, fused=True, objects=False):
    from gsplat_attack.renderer import PipelineParams, render
    model.zero_grad()
    out = render(cam, model, PipelineParams(fused_activations=fused, skip_objects=not objects), bg)
    out["render"].backward(gc)
    grads = {k: v.grad.detach().clone() for k, v in model.named_parameters().items() if v.grad is not None}
    return out, grads


@pytest.fixture(scope="module")
def nyc():
    D, dev, model, cams = _setup("nyc-1M", n_views=8)
    cam = cams[2]
    gc = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(99)).to(dev)
    return D, dev, model, cam, gc


def test_nyc_1m_1080p_is_bitwise_reproducible_and_linear(nyc):
    D, dev, model, cam, gc = nyc
    bg = torch.zeros(3, device=dev)
    out1, g1 = _fwd_bwd(model, cam, bg, gc)
    img1 = out1["render"].detach().clone()
    N = D.last_num_rendered(out1["render"])
    assert (cam.image_height, cam.image_width) == (1080, 1920) and model.get_xyz.shape[0] == 1_000_000
    assert N > 1_000_000 and int((out1["radii"] > 0).sum()) > 300_000
    out2, g2 = _fwd_bwd(model, cam, bg, gc)
    assert torch.equal(img1, out2["render"])
    for k in g1:
        assert torch.equal(g1[k], g2[k]), f"{k}: two identical runs differ (no atomics => must be bitwise equal)"
    # backward is linear in dL/dC: scaling by 2 is exact in floating point
    _, g3 = _fwd_bwd(model, cam, bg, 2.0 * gc)
    for k in g1:
        assert torch.equal(2.0 * g1[k], g3[k]), f"{k}: backward(2g) != 2 backward(g)"
    assert all(torch.isfinite(v).all() for v in g1.values())


def test_nyc_background_enters_through_final_transmittance(nyc):
    D, dev, model, cam, gc = nyc
    from gsplat_attack.renderer import PipelineParams, render
    with torch.no_grad():
        pass
    pipe = PipelineParams(skip_objects=True)
    a = render(cam, model, pipe, torch.zeros(3, device=dev))["render"]
    bgv = torch.tensor([0.9, 0.3, 0.6], device=dev)
    b_out = render(cam, model, pipe, bgv)
    b = b_out["render"]
    Tf = D.export_state(b, "final_T").view(cam.image_height, cam.image_width)
    assert float(Tf.min()) >= 0.0 and float(Tf.max()) <= 1.0
    assert torch.allclose(b - a, Tf[None] * bgv[:, None, None], atol=2e-6)


def model_radii(model, cam, dev):
    from gsplat_attack.renderer import PipelineParams, render
    with torch.no_grad():
        return render(cam, model, PipelineParams(skip_objects=True), torch.zeros(3, device=dev))["radii"]


def test_nyc_pair_lists_are_depth_sorted_per_tile(nyc):
    D, dev, model, cam, gc = nyc
    from gsplat_attack.renderer import PipelineParams, render
    img = render(cam, model, PipelineParams(skip_objects=True), torch.zeros(3, device=dev))["render"]
    # inside every tile's span the entries are in (depth key, Gaussian index) order, strictly
    from util import check_tile_lists_depth_order
    ranges, g = check_tile_lists_depth_order(D, img)
    assert bool((model_radii(model, cam, dev)[g] > 0).all())       # every listed Gaussian is a visible one


def test_nyc_cull_and_fused_paths_agree_at_full_size(nyc):
    D, dev, model, cam, gc = nyc
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    # Unsplit lists: dropping the pairs the footprint test rejects changes NOTHING, bit for bit.
    try:
        D.set_flags(D.FLAG_NO_SEGMENTS)
        out0, g0 = _fwd_bwd(model, cam, bg, gc, fused=True)
        img0 = out0["render"].detach().clone()
        D.set_flags(D.FLAG_NO_SEGMENTS | D.FLAG_NO_CULL)
        out1, g1 = _fwd_bwd(model, cam, bg, gc, fused=True)
    finally:
        D.set_flags(0)
    assert torch.equal(img0, out1["render"])
    for k in g0:
        # the same partial rows are summed per Gaussian; only a Gaussian whose full rect has more rows than K9 stages at
        # once (summed by the whole wave) and whose tightened rect has fewer (summed by one lane) adds them in another order
        scale = g0[k].abs().max().clamp_min(1e-30)
        assert ((g0[k] - g1[k]).abs().max() / scale).item() <= 2e-6, k
        assert (g0[k] != g1[k]).float().mean().item() <= 1e-3, k
    # Default (long lists walked as segments by the backward): the forward is the same bits; the backward starts each
    # segment from the forward's stored (T, C) instead of dividing T back through the whole list -- same numbers
    # within float32 rounding, and the segment boundaries move with the culled pairs.
    outs, gs = _fwd_bwd(model, cam, bg, gc, fused=True)
    assert torch.equal(img0, outs["render"])
    try:
        D.set_flags(D.FLAG_NO_CULL)
        outn, gn = _fwd_bwd(model, cam, bg, gc, fused=True)
    finally:
        D.set_flags(0)
    assert torch.equal(img0, outn["render"])
    for k in g0:
        scale = g0[k].abs().max().clamp_min(1e-30)
        assert ((gs[k] - g0[k]).abs().max() / scale).item() <= 1e-4, k
        assert ((gn[k] - g0[k]).abs().max() / scale).item() <= 1e-4, k
    # Fused vs PyTorch activations differ by an ulp in scale / rotation / opacity, which flips a threshold test
    # (alpha >= 1/255, T < 1e-4) on a handful of the 2M pixels: all but 1e-4 of the pixels must agree to 2e-6, the
    # rest are bounded by one skipped/added contribution.
    out2, g2 = _fwd_bwd(model, cam, bg, gc, fused=False)
    diff = (img0 - out2["render"]).abs().max(dim=0).values
    assert (diff > 2e-6).float().mean().item() <= 1e-4
    assert diff.max().item() <= 1e-2
    for k in g0:
        rel = ((gs[k] - g2[k]).abs().max() / gs[k].abs().max().clamp_min(1e-30)).item()
        assert rel <= 2e-3, (k, rel)


def test_airport_4k_full_backward():
    """Config 5: 2M Gaussians at 3840x2160, all attribute gradients, object channels on."""
    D, dev, model, cams = _setup("airport-4K", n_views=1)
    cam = cams[0]
    gc = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(5)).to(dev)
    bg = torch.zeros(3, device=dev)
    out1, g1 = _fwd_bwd(model, cam, bg, gc, objects=True)
    img = out1["render"].detach().clone()
    assert tuple(img.shape) == (3, 2160, 3840) and tuple(out1["render_object"].shape) == (16, 2160, 3840)
    assert torch.isfinite(img).all() and all(torch.isfinite(v).all() for v in g1.values())
    out2, g2 = _fwd_bwd(model, cam, bg, gc, objects=True)
    assert torch.equal(img, out2["render"])
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    assert g1["objects_dc"].abs().max().item() == 0.0      # nothing flowed into the object channels


def test_hydrant_full_800px_sh_gradients():
    """Config 2: 300k Gaussians, 800x800, gradients consumed on the SH coefficients."""
    D, dev, model, cams = _setup("hydrant-full", n_views=1)
    cam = cams[0]
    gc = torch.randn(3, 800, 800, generator=torch.Generator().manual_seed(2)).to(dev)
    out, g = _fwd_bwd(model, cam, torch.zeros(3, device=dev), gc)
    assert g["f_dc"].abs().max().item() > 0 and g["f_rest"].abs().max().item() > 0
    # DC colour gradient of a visible Gaussian is C0 * dL/drgb: the three channels of f_dc relate to f_rest's
    # degree-1 terms through the same dL/drgb -> both vanish on exactly the same (invisible) Gaussians
    vis = out["radii"] > 0
    assert g["f_dc"][~vis].abs().max().item() == 0.0 and g["f_rest"][~vis].abs().max().item() == 0.0


def test_nyc_two_parameter_sets_render_as_the_concatenated_scene_at_full_size(nyc):
    """gsr_forward_raw2 at BASELINE size: an attacked target (400 003 Gaussians) beside a frozen background (the other
    599 997) gives the image and the radii of the one-million-Gaussian scene they concatenate to, bit for bit, with
    and without the object channels (reference attack.py:513-530 builds that scene with seven torch.cat per
    iteration and never differentiates its render)."""
    D, dev, model, cam, gc = nyc
    from gsplat_attack.renderer import PipelineParams, render, render_pair
    P = model.get_xyz.shape[0]
    cut = 400_003                                          # not a multiple of 64: one wave reads both sets
    mask = torch.zeros(P, dtype=torch.bool, device=dev)
    mask[:cut] = True
    target, background = model.clone(), model.clone()
    target.removal_setup(~mask)
    background.removal_setup(mask)
    bg = torch.tensor([0.2, 0.4, 0.1], device=dev)
    for objects in (False, True):
        pipe = PipelineParams(skip_objects=not objects)
        with torch.no_grad():
            ref = render(cam, model, pipe, bg)
            got = render_pair(cam, target, background, pipe, bg)
        assert torch.equal(ref["render"], got["render"])
        assert torch.equal(ref["radii"], got["radii"]) and got["radii"].numel() == P
        if objects:
            assert torch.equal(ref["render_object"], got["render_object"])
