"""The reference's own A/B hooks on the device: pipe.convert_SHs_python / pipe.compute_cov3D_python
(reference gaussian_renderer/__init__.py:62-63,70-78) swap K1's colour and covariance stages for the PyTorch twins the golden
fixtures pin."""
import pytest
import torch
from util import grad_error

pytestmark = pytest.mark.gpu


RAW = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")


def _hip():
    import diff_gaussian_rasterization as D
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    D._load()
    return D


def _render_grads(model, cam, pipe, bg, gc):
    from gsplat_attack.renderer import render
    model.zero_grad()
    out = render(cam, model, pipe, bg)
    out["render"].backward(gc)
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters().items() if p.grad is not None}
    return out, grads, out["viewspace_points"].grad.detach().clone()


def _small_scene(n_views=3, P=20000, w=320, h=192):
    from gsplat_attack.scenes import make_scene
    _hip()
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=w, height=h, n_views=n_views)
    return dev, model, cams


@pytest.mark.parametrize("scene,kw", [("hydrant-1k", {}), ("nyc-1M", {})])
@pytest.mark.parametrize("switch", ["convert_SHs_python", "compute_cov3D_python", "both"])
def test_reference_ab_hooks_agree_with_the_kernels_own_stages(scene, kw, switch):
    """render() with the reference's Python twin of a stage == render() with the HIP stage: image <= 1e-5, every raw
    attribute gradient <= 1e-3 (SH path: dL/df_dc, dL/df_rest and the view-direction term of dL/dxyz come from autograd
    through eval_sh instead of K9; covariance path: dL/dscaling and dL/drotation through build_scaling_rotation)."""
    from gsplat_attack.renderer import PipelineParams
    from gsplat_attack.scenes import make_scene
    _hip()
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene(scene, device=dev, n_views=1, **kw)
    cam = cams[0]
    H, W = cam.image_height, cam.image_width
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    gc = torch.randn(3, H, W, generator=torch.Generator().manual_seed(41)).to(dev)
    base_out, base, base_vs = _render_grads(model, cam, PipelineParams(skip_objects=True), bg, gc)
    classic_out, classic, _ = _render_grads(model, cam, PipelineParams(skip_objects=True, fused_activations=False), bg, gc)
    pipe = PipelineParams(skip_objects=True, convert_SHs_python=switch in ("convert_SHs_python", "both"),
                          compute_cov3D_python=switch in ("compute_cov3D_python", "both"))
    out, grads, vs = _render_grads(model, cam, pipe, bg, gc)
    # The Python twin's float32 rounding differs from the kernel stage's by an ulp here and there (torch's matrix products
    # for the covariance, exp / sigmoid / normalize for the activations the fused path applies in-kernel): on a couple of
    # the 2 M pixels of the full-size scene that flips a threshold test (alpha >= 1/255, T < 1e-4) -- one contribution
    # added or skipped, <= 5e-3.  All but 1e-4 of the pixels agree to 1e-5; the colour twin alone changes no geometry
    # and must agree everywhere with the classic surface (same activated inputs).
    d_f = (out["render"] - base_out["render"]).abs().amax(dim=0)
    d_c = (out["render"] - classic_out["render"]).abs().amax(dim=0)
    print(f"{scene} {switch}: image vs fused {float(d_f.max()):.2e} ({float((d_f > 1e-5).float().mean()):.1e} of the pixels "
          f"> 1e-5), vs classic {float(d_c.max()):.2e} ({float((d_c > 1e-5).float().mean()):.1e})")
    for d in (d_f, d_c):
        assert float((d > 1e-5).float().mean()) <= 1e-4 and float(d.max()) <= 5e-3
    if switch == "convert_SHs_python":
        assert float(d_c.max()) <= 1e-5
        assert torch.equal(out["radii"], classic_out["radii"])
    assert float((out["radii"] != base_out["radii"]).float().mean()) <= 1e-5
    for n in RAW:
        norm, frac = grad_error(grads[n], base[n], elem_tol=5e-3)
        assert base[n].abs().max().item() > 0, n
        assert norm <= 1e-3, f"{switch}: grad {n} normwise rel err {norm:.3e}"
        assert frac <= 1e-3, f"{switch}: grad {n}: {frac:.2e} of the significant elements off by more than 5e-3"
    norm, _ = grad_error(vs, base_vs)
    assert norm <= 1e-3, f"{switch}: viewspace gradient {norm:.3e}"


def test_python_colours_keep_the_reference_clamp_gradient():
    """The convert_SHs_python branch clamps with clamp_min(. + 0.5, 0) (gaussian_renderer/__init__.py:78): a Gaussian
    whose colour is clamped gets no SH gradient in that channel -- on both paths."""
    from gsplat_attack.renderer import PipelineParams
    from gsplat_attack.scenes import make_scene
    _hip()
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("hydrant-1k", device=dev, n_views=1)
    with torch.no_grad():
        model._features_dc[::3] = -40.0                         # a third of the splats: every channel far below zero
    cam = cams[0]
    bg = torch.zeros(3, device=dev)
    gc = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(3)).to(dev)
    _, a, _ = _render_grads(model, cam, PipelineParams(skip_objects=True), bg, gc)
    _, b, _ = _render_grads(model, cam, PipelineParams(skip_objects=True, convert_SHs_python=True), bg, gc)
    assert float(a["f_dc"][::3].abs().max()) == 0.0 and float(b["f_dc"][::3].abs().max()) == 0.0
    assert float(a["f_rest"][::3].abs().max()) == 0.0 and float(b["f_rest"][::3].abs().max()) == 0.0
    for n in ("f_dc", "f_rest", "xyz"):
        assert grad_error(b[n], a[n])[0] <= 1e-3, n


@pytest.mark.parametrize("switch", ["convert_SHs_python", "compute_cov3D_python"])
def test_pgd_attack_with_a_python_switch_still_steps(switch):
    """(high) With a reference switch on, render() takes the classic surface, which never writes a GradBucket: the
    loop must not create buckets then (it used to zero .grad and take no step at all)."""
    from gsplat_attack.attack import pgd_attack
    from gsplat_attack.renderer import PipelineParams
    dev, model, cams = _small_scene(n_views=3)
    kw = dict(iters=3, groups=("color", "position", "scaling", "rotation", "opacity"), alpha=0.05, epsilon=0.5, streams=1)
    pipe = PipelineParams(skip_objects=True, **{switch: True})
    a, b, c = model.clone(), model.clone(), model.clone()
    ha = pgd_attack(a, cams, pipe=pipe, use_buckets=True, **kw)
    hb = pgd_attack(b, cams, pipe=pipe, use_buckets=False, **kw)
    hc = pgd_attack(c, cams, use_buckets=True, **kw)                    # default fused path with buckets
    assert max(abs(x - y) for x, y in zip(ha, hb)) <= 1e-6 * max(1.0, max(abs(x) for x in ha))
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        moved = (getattr(a, n).detach() - getattr(model, n).detach()).abs().max().item()
        assert moved > 0, f"{n} did not move: the attack took no step"
        assert torch.equal(getattr(a, n).detach(), getattr(b, n).detach()), n
        ref = getattr(c, n).detach()
        d = (getattr(a, n).detach() - ref).abs().max().item()
        assert d <= 1e-3 * max(1.0, ref.abs().max().item()), (n, d)
    assert max(abs(x - y) for x, y in zip(ha, hc)) <= 1e-4 * max(1.0, max(abs(x) for x in hc))
