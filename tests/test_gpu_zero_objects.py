"""All-zero object features (the attack's combined scenes, reference scene/gaussian_model.py:528; the reference's render()
passes them all the same, gaussian_renderer/__init__.py:81): the classic surface composites without the 16 object channels
(GSR_FLAG_OBJECTS_FOR_BACKWARD_ONLY) and must give what the object variant of the compositor gives -- the same image bit
for bit, an object map of zeros, the same gradients, dL/dsh_objs included when the object map IS differentiated."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(model, cam, bg, gc, go, shortcut, zero=True):
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
        out = render(cam, m, PipelineParams(skip_objects=False, fused_activations=False), bg)
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


@pytest.mark.parametrize("with_go", [False, True])
def test_zero_object_features_take_the_plain_compositor_with_equal_results(with_go):
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
    img0, obj0, rad0, g0 = _run(model, cam, bg, gc, go, shortcut=False)
    img1, obj1, rad1, g1 = _run(model, cam, bg, gc, go, shortcut=True)
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


def test_nonzero_object_features_are_composited_as_before():
    import diff_gaussian_rasterization as D
    from gsplat_attack.scenes import make_scene
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("hydrant-1k", device=dev, n_views=1)
    bg = torch.zeros(3, device=dev)
    gc = torch.ones(3, 128, 128, device=dev)
    img0, obj0, _, g0 = _run(model, cams[0], bg, gc, None, shortcut=False, zero=False)
    img1, obj1, _, g1 = _run(model, cams[0], bg, gc, None, shortcut=True, zero=False)
    assert torch.equal(img0, img1) and torch.equal(obj0, obj1) and float(obj1.abs().max()) > 0.0
    # a tensor seen non-zero is not read again
    key = [k for k, v in D._OBJ_ZERO.items() if not v[2]]
    assert len(key) == 1
