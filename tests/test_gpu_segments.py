"""Long tile lists walked as segments by the backward (north_star: prefix-scan for transmittance) against the whole-list
walk they replace: same image bits, same gradients within float32 rounding -- at sizes where many tiles are split, for
both backward tile splits."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(model, cam, bg, gc, flags):
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import PipelineParams, render
    model.zero_grad()
    try:
        D.set_flags(flags)
        out = render(cam, model, PipelineParams(skip_objects=True), bg)
        out["render"].backward(gc)
        torch.cuda.synchronize()
        nc = D.export_state(out["render"], "n_contrib").clone()
        ft = D.export_state(out["render"], "final_T").clone()
    finally:
        D.set_flags(0)
    grads = {k: v.grad.detach().clone() for k, v in model.named_parameters().items() if v.grad is not None}
    return out["render"].detach().clone(), nc, ft, grads


@pytest.mark.parametrize("scene,kw", [("hydrant-full", dict(P=120000, width=480, height=400)),
                                      ("nyc-1M", dict(P=400000, width=960, height=544))])
def test_segmented_walks_equal_the_whole_list_walks(scene, kw):
    import diff_gaussian_rasterization as D
    from gsplat_attack.scenes import make_scene
    D._load()
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene(scene, device=dev, n_views=1, **kw)
    cam = cams[0]
    bg = torch.tensor([0.2, 0.3, 0.1], device=dev)
    gc = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(3)).to(dev)
    whole = _run(model, cam, bg, gc, D.FLAG_NO_SEGMENTS)
    rg = None
    from gsplat_attack.renderer import PipelineParams, render
    img = render(cam, model, PipelineParams(skip_objects=True), bg)["render"]
    rg = D.export_state(img, "ranges").view(-1, 2).long()
    lens = rg[:, 1] - rg[:, 0]
    assert int((lens > 256).sum()) > 5, "the scene must have split tiles"
    for name, flags in (("library's choice", 0), ("two waves per tile", D.flag_bwd_split(2)),
                        ("one wave per tile, image-order tile map", D.flag_bwd_split(4) | D.flag_tile_map(0))):
        img, nc, ft, grads = _run(model, cam, bg, gc, flags)
        assert torch.equal(img, whole[0]) and torch.equal(nc, whole[1]) and torch.equal(ft, whole[2]), name
        for k in whole[3]:
            # Both walks are float32 recursions over ~1000 list entries (T divided back vs restarted from the stored
            # (T, C)): their per-tile sums differ by float32 rounding, and so do the gradients -- ~1e-6 of the gradient
            # scale (observed <= 1.3e-6).  (Until K9 formed dL/dconic -> dL/dcov2D in double, needle-shaped splats
            # amplified that rounding to 1.2e-4 of the scale here: EXPERIMENTS.md, round 3.)
            scale = whole[3][k].abs().max().clamp_min(1e-30)
            err = ((grads[k] - whole[3][k]).abs().max() / scale).item()
            print(f"segments vs whole list [{scene}, {name}] {k}: {err:.2e}")
            assert err <= 2e-5, (name, k)
