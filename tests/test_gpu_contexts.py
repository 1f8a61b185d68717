"""What a forward keeps for its backward: nothing under no_grad, segment records also after a forward with object channels."""
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


NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _scene(P=40000, W=320, H=192, n_views=2):
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=W, height=H, n_views=n_views)
    return dev, model, cams


def test_forward_without_grad_keeps_no_backward_state():
    """a torch.no_grad() render of a model whose parameters require grad must not keep the rasteriser's
    backward state (segment boundary records, d colour / d direction)."""
    D = _hip()
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams = _small_scene(n_views=1, P=200000, w=960, h=544)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams(skip_objects=True)
    with torch.no_grad():
        out = render(cams[0], model, pipe, bg)
    assert out["render"].grad_fn is None and not out["render"].requires_grad
    out_g = render(cams[0], model, pipe, bg)
    assert out_g["render"].grad_fn is not None and out_g["render"].grad_fn.holder is not None
    assert torch.equal(out["render"], out_g["render"].detach())
    # plain tensors without requires_grad: no context either
    frozen = model.clone()
    for p in frozen.parameters():
        p.requires_grad_(False)
    out_f = render(cams[0], frozen, PipelineParams(skip_objects=True, viewspace_grad=False), bg)
    assert out_f["render"].grad_fn is None


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
