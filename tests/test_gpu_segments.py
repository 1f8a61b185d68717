"""Long tile lists composited as segments (north_star: prefix-scan for transmittance): the backward's segments, the
segmented forward of small images, and the whole-list walks they replace must give the same numbers within float32
rounding -- at a size where most tiles are split, with the library's own choice and with each mode forced."""
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
                                      ("nyc-1M", dict(P=150000, width=1280, height=832))])
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
    assert int((lens > 256).sum()) > 20, "the scene must have split tiles"
    for name, flags in (("library's choice", 0), ("forward segments on", D.flag_fwd_segments(1)),
                        ("forward segments off", D.flag_fwd_segments(2)),
                        ("forward segments on, 4 waves per tile", D.flag_fwd_segments(1) | D.flag_fwd_split(1)),
                        ("forward segments on, 1 wave per tile", D.flag_fwd_segments(1) | D.flag_fwd_split(4) | D.flag_bwd_split(2))):
        img, nc, ft, grads = _run(model, cam, bg, gc, flags)
        d = (img - whole[0]).abs().max(dim=0).values
        # a pixel whose stop test sits within rounding of 1e-4 may stop one entry earlier or later
        assert (d > 3e-6).float().mean().item() <= 2e-5, (name, (d > 3e-6).float().mean().item())
        assert d.max().item() <= 2e-3, (name, d.max().item())
        same = nc == whole[1]
        assert (~same).float().mean().item() <= 2e-5, name
        assert (ft - whole[2]).abs()[same].max().item() <= 1e-6, name
        for k in whole[3]:
            scale = whole[3][k].abs().max().clamp_min(1e-30)
            assert ((grads[k] - whole[3][k]).abs().max() / scale).item() <= 2e-4, (name, k)
    # forward segments on / off give the SAME image bits when forced on two different tile splits (the per-pixel
    # arithmetic of a segment does not depend on which wave owns the pixel)
    a = _run(model, cam, bg, gc, D.flag_fwd_segments(1) | D.flag_fwd_split(1))[0]
    b = _run(model, cam, bg, gc, D.flag_fwd_segments(1) | D.flag_fwd_split(2))[0]
    assert torch.equal(a, b)
