"""All-zero object features (the attack's combined scenes, reference scene/gaussian_model.py:528; the reference's render()
passes them all the same, gaussian_renderer/__init__.py:81): the classic surface composites without the 16 object channels
(GSR_FLAG_OBJECTS_FOR_BACKWARD_ONLY) and must give what the object variant of the compositor gives -- the same image bit
for bit, an object map of zeros, the same gradients, dL/dsh_objs included when the object map IS differentiated."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(model, cam, bg, gc, go, shortcut, zero=True, fused=False):
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import PipelineParams, render
    m = model.clone()
    if zero:
        with torch.no_grad():
            m._objects_dc.zero_()
    old = D._OBJ_SHORTCUT
    D._OBJ_SHORTCUT = shortcut
    D._OBJ_ZERO.clear()
    try:
        out = render(cam, m, PipelineParams(skip_objects=False, fused_activations=fused), bg)
        loss = (out["render"] * gc).sum()
        if go is not None:
            loss = loss + (out["render_object"] * go).sum()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        D._OBJ_SHORTCUT = old
    grads = {n: (None if getattr(m, n).grad is None else getattr(m, n).grad.clone())
             for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_objects_dc")}
    return out["render"].detach().clone(), out["render_object"].detach().clone(), out["radii"].clone(), grads


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("with_go", [False, True])
def test_zero_object_features_take_the_plain_compositor_with_equal_results(with_go, fused):
    # fused: the raw-parameter path (what gsplat_attack.patch_reference() / GSR_PATCH_REFERENCE=1 give the reference's render())
    from gsplat_attack.scenes import make_scene
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=50_000, width=512, height=288, n_views=1)
    cam = cams[0]
    bg = torch.tensor([0.2, 0.1, 0.4], device=dev)
    g = torch.Generator().manual_seed(3)
    gc = torch.randn(3, 288, 512, generator=g).to(dev)
    go = torch.randn(16, 288, 512, generator=g).to(dev) if with_go else None
    img0, obj0, rad0, g0 = _run(model, cam, bg, gc, go, shortcut=False, fused=fused)
    img1, obj1, rad1, g1 = _run(model, cam, bg, gc, go, shortcut=True, fused=fused)
    assert torch.equal(img0, img1) and torch.equal(rad0, rad1)
    assert float(obj0.abs().max()) == 0.0 and float(obj1.abs().max()) == 0.0
    assert obj1.shape == obj0.shape == (16, 288, 512)
    for n in g0:
        if n == "_objects_dc" and not with_go:
            # nobody differentiated the object map: zeros (object variant) or no gradient at all (shortcut)
            assert g1[n] is None or float(g1[n].abs().max()) == 0.0
            continue
        assert g0[n] is not None and g1[n] is not None, n
        assert torch.equal(g0[n], g1[n]), n
    if with_go:
        assert float(g1["_objects_dc"].abs().max()) > 0.0           # zero features still receive their gradient


@pytest.mark.parametrize("fused", [False, True])
def test_nonzero_object_features_are_composited_as_before(fused):
    import diff_gaussian_rasterization as D
    from gsplat_attack.scenes import make_scene
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("hydrant-1k", device=dev, n_views=1)
    bg = torch.zeros(3, device=dev)
    gc = torch.ones(3, 128, 128, device=dev)
    img0, obj0, _, g0 = _run(model, cams[0], bg, gc, None, shortcut=False, zero=False, fused=fused)
    img1, obj1, _, g1 = _run(model, cams[0], bg, gc, None, shortcut=True, zero=False, fused=fused)
    assert torch.equal(img0, img1) and torch.equal(obj0, obj1) and float(obj1.abs().max()) > 0.0
    # a tensor seen non-zero is not read again
    key = [k for k, v in D._OBJ_ZERO.items() if not v[2]]
    assert len(key) == 1


def test_zero_object_features_through_a_kept_context():
    """The fused path with a RenderCache: a model whose object features are all zero keeps a context without object
    channels; images and SH gradients over colour steps equal the uncached renders; making the features non-zero takes the
    full forward with object channels again."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams, render
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=30_000, width=320, height=192, n_views=1)
    with torch.no_grad():
        model._objects_dc.zero_()
    for n in ("_xyz", "_scaling", "_rotation", "_opacity", "_objects_dc"):
        getattr(model, n).requires_grad_(False)
    bg = torch.tensor([0.2, 0.1, 0.4], device=dev)
    gc = torch.randn(3, 192, 320, generator=torch.Generator().manual_seed(4)).to(dev)
    cache = RenderCache()
    plain = PipelineParams(skip_objects=False, viewspace_grad=False)
    cached = PipelineParams(skip_objects=False, viewspace_grad=False, render_cache=cache)

    def one(pipe):
        model.zero_grad()
        out = render(cams[0], model, pipe, bg)
        out["render"].backward(gc)
        return out["render"].detach().clone(), out["render_object"].detach().clone(), model._features_rest.grad.clone()
    for it in range(3):
        a, b = one(plain), one(cached)
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and float(b[1].abs().max()) == 0.0
        with torch.no_grad():
            model._features_dc.add_(0.05)
    assert cache.hits == 2
    with torch.no_grad():
        model._objects_dc.add_(0.5)
    a, b = one(plain), one(cached)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and float(b[1].abs().max()) > 0.0 and cache.hits == 2
