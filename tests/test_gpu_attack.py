"""The PGD loop counterpart on the HIP path: the surrogate loss goes down, perturbations stay inside the eps-ball,
and a batch of views gives the same gradient as the sum of the single-view gradients."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(**kw):
    from gsplat_attack.scenes import make_scene
    import diff_gaussian_rasterization as D
    D._load()
    return make_scene("nyc-1M", device=torch.device("cuda:0"), P=20000, width=320, height=192, **kw)


def test_pgd_colour_attack_lowers_the_surrogate_loss_and_respects_epsilon():
    from gsplat_attack.attack import pgd_attack
    model, cams, _ = _scene(n_views=3)
    orig_rest = model._features_rest.detach().clone()
    orig_dc = model._features_dc.detach().clone()
    hist = pgd_attack(model, cams, iters=8, alpha=0.5, epsilon=0.75, groups=("color",), norm="l2")
    assert hist[-1] < hist[0]
    d_rest = (model._features_rest.detach() - orig_rest).reshape(orig_rest.shape[0], -1).norm(dim=1)
    d_dc = (model._features_dc.detach() - orig_dc).reshape(orig_dc.shape[0], -1).norm(dim=1)
    assert float(d_rest.max()) <= 0.75 + 1e-5 and float(d_dc.max()) <= 0.75 + 1e-5
    assert float(d_rest.max()) > 0


def test_all_five_attribute_groups_step():
    from gsplat_attack.attack import pgd_attack
    model, cams, _ = _scene(n_views=2)
    before = {n: p.detach().clone() for n, p in model.named_parameters().items()}
    pgd_attack(model, cams, iters=2, alpha=0.01, epsilon=0.05, norm="linf",
               groups=("color", "position", "scaling", "rotation", "opacity"))
    for n in ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity"):
        delta = (model.named_parameters()[n].detach() - before[n]).abs().max().item()
        assert 0 < delta <= 0.05 + 1e-6, (n, delta)


def test_batch_gradient_is_the_sum_of_view_gradients():
    """What loss.backward() over a batch accumulates (reference attack.py:476-494) == what view-sharded ranks
    all-reduce: sum over views of the per-view attribute gradients (bitwise: each view's backward is deterministic)."""
    from gsplat_attack.attack import SurrogateDetector
    from gsplat_attack.renderer import PipelineParams, render
    model, cams, _ = _scene(n_views=3)
    det = SurrogateDetector().cuda()
    pipe = PipelineParams(skip_objects=True)
    bg = torch.zeros(3, device="cuda")
    per_view = []
    for cam in cams:
        model.zero_grad()
        det(render(cam, model, pipe, bg)["render"][None]).backward()
        per_view.append({n: p.grad.detach().clone() for n, p in model.named_parameters().items() if p.grad is not None})
    model.zero_grad()
    loss = sum(det(render(cam, model, pipe, bg)["render"][None]) for cam in cams)
    loss.backward()
    for n, p in model.named_parameters().items():
        if p.grad is None:
            continue
        ref = per_view[0][n] + per_view[1][n] + per_view[2][n]
        assert torch.allclose(p.grad, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max())), n


def test_fused_backward_hands_out_one_flat_gradient_bucket():
    """After render()+backward() on the fused path the six attack gradients are back-to-back slices of one
    buffer of 59 floats per Gaussian: the data-parallel step sums them with a single collective."""
    from gsplat_attack import dist as gdist
    from gsplat_attack.renderer import PipelineParams, render
    model, cams, _ = _scene(n_views=2)
    for cam in cams:                       # two views accumulate into the same bucket
        render(cam, model, PipelineParams(skip_objects=True), torch.zeros(3, device="cuda"))["render"].sum().backward()
    grads = [getattr(model, n).grad for n in gdist.ATTACK_PARAMS]
    flat = gdist._flat_view_of(grads)
    assert flat is not None and flat.numel() == 59 * model.get_xyz.shape[0]
    assert flat.data_ptr() == model._xyz.grad.data_ptr()


def test_combined_scene_render_equals_render_of_the_concatenated_attributes():
    """target + frozen background (reference attack.py:513-530): splitting a scene in two and re-combining it must
    reproduce the render of the whole scene bit for bit (pairs only depend on the multiset of splats and their order)."""
    from gsplat_attack.attack import render_combined
    from gsplat_attack.renderer import PipelineParams, render
    model, cams, _ = _scene(n_views=1)
    P = model.get_xyz.shape[0]
    mask = torch.zeros(P, dtype=torch.bool, device="cuda")
    mask[: P // 3] = True
    target, background = model.clone(), model.clone()
    background.removal_setup(~mask)            # keeps the first third ...
    target.removal_setup(mask)                 # ... and the rest
    bg = torch.tensor([0.0, 0.1, 0.2], device="cuda")
    whole = model.clone()
    whole.removal_setup(torch.zeros(P, dtype=torch.bool, device="cuda"))
    # same storage order as the combination: target rows first, background rows after
    order = torch.cat([torch.nonzero(~mask).flatten(), torch.nonzero(mask).flatten()])
    for n in whole._PARAM_ATTRS:
        setattr(whole, n, torch.nn.Parameter(getattr(model, n).detach()[order].clone()))
    with torch.no_grad():
        ref = render(cams[0], whole, PipelineParams(skip_objects=True), bg)["render"]
    got = render_combined(target, background, cams[:1], bg)[0]
    assert torch.equal(ref, got)


@pytest.mark.parametrize("shape", [(5000, 3), (4097, 4), (3000, 1), (2500, 15, 3), (1, 1, 3), (64, 45),
                                   (5022, 3), (4106, 15, 3), (94, 4), (30, 3), (10, 45)])
@pytest.mark.parametrize("kind", ["linf", "l2"])
def test_fused_pgd_step_matches_the_tensor_formulation(shape, kind):
    """gsr_pgd_step (HIP) against the PyTorch statement of the reference's update rules (gsplat_attack/pgd.py on CPU,
    itself pinned to attack.py:25-173 by tests/test_golden_twins.py), including rows inside and outside the eps ball
    and a zero gradient."""
    from gsplat_attack import pgd
    g = torch.Generator().manual_seed(sum(shape) + len(kind))
    x = torch.randn(*shape, generator=g)
    x0 = x + torch.randn(*shape, generator=g) * torch.rand(shape[0], *([1] * (len(shape) - 1)), generator=g) * 2.0
    grad = torch.randn(*shape, generator=g) * 3.0
    step = pgd.l2_step_ if kind == "l2" else pgd.linf_step_
    for gr in (grad, torch.zeros_like(grad)):
        want = x.clone()
        step(want, gr, 0.5, 0.8, x0)                       # CPU tensors: the formulation as written
        got = x.clone().cuda()
        step(got, gr.cuda(), 0.5, 0.8, x0.cuda())           # device tensors: the fused kernel
        torch.cuda.synchronize()
        assert torch.allclose(got.cpu(), want, atol=2e-6, rtol=1e-5), (shape, kind, (got.cpu() - want).abs().max())
        if kind == "linf":
            assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("kind", ["linf", "l2", "l2_normed"])
@pytest.mark.parametrize("rows", [5000, 4097, 37])
def test_all_stepped_tensors_in_one_launch_equal_the_per_tensor_steps(kind, rows):
    """gsr_pgd_step_multi: the six attribute tensors of a model (45, 3, 3, 1, 3, 4 floats per row; the gradients slices of
    one flat bucket, so most start off a 16-byte boundary for odd rows) stepped in one launch -- bit for bit the per-tensor
    gsr_pgd_step / gsr_pgd_step_normed, with norms already on the device for some, all or none of the tensors."""
    from gsplat_attack import pgd
    g = torch.Generator().manual_seed(rows)
    shapes = [(rows, 15, 3), (rows, 1, 3), (rows, 3), (rows, 1), (rows, 3), (rows, 4)]
    flat = (torch.randn(59 * rows, generator=g) * 2.0).cuda()
    xs, x0s, grads, o = [], [], [], 0
    for sh in shapes:
        n = math.prod(sh)
        x = torch.randn(*sh, generator=g)
        xs.append(x.cuda())
        x0s.append((x + torch.randn(*sh, generator=g) * torch.rand(rows, *([1] * (len(sh) - 1)), generator=g) * 2.0).cuda())
        grads.append(flat[o:o + n].view(*sh))
        o += n
    grads[3] = torch.zeros_like(grads[3])                   # a zero gradient: no step for that tensor, projection only
    l2 = kind != "linf"
    ss = [None] * 6
    if kind == "l2_normed":
        ss = [(gr.double() ** 2).sum().reshape(1) if i != 4 else None for i, gr in enumerate(grads)]     # one without
    want = [x.clone() for x in xs]
    for w, gr, x0, s_ in zip(want, grads, x0s, ss):
        if l2:
            pgd.l2_step_(w, gr, 0.5, 0.8, x0, sumsq=s_)
        else:
            pgd.linf_step_(w, gr, 0.5, 0.8, x0)
    got = [x.clone() for x in xs]
    vers = [t._version for t in got]
    assert pgd.multi_step_(list(zip(got, grads, x0s, ss)), 0.5, 0.8, l2)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), (kind, rows, i, (a - b).abs().max().item())
        assert a._version > vers[i]
    # tensors of different lengths in one launch (the table of first workgroups per tensor)
    cut = [rows, max(rows // 3, 1), rows, 1, max(rows - 5, 1), rows]
    want2 = [x[:c].clone() for x, c in zip(xs, cut)]
    for w, gr, x0, c in zip(want2, grads, x0s, cut):
        (pgd.l2_step_ if l2 else pgd.linf_step_)(w, gr[:c].contiguous(), 0.5, 0.8, x0[:c].contiguous())
    got2 = [x[:c].clone() for x, c in zip(xs, cut)]
    assert pgd.multi_step_([(g_, gr[:c].contiguous(), x0[:c].contiguous(), None) for g_, gr, x0, c in zip(got2, grads, x0s, cut)],
                           0.5, 0.8, l2)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got2, want2)):
        assert torch.equal(a, b), (kind, rows, i, "ragged")
    # tensors the fused update does not take: nothing touched, the caller steps tensor by tensor
    keep = got[0].clone()
    assert not pgd.multi_step_([(got[0], grads[0], x0s[0], None), (got[1].double(), grads[1].double(), x0s[1].double(), None)],
                               0.5, 0.8, l2)
    assert not pgd.multi_step_([(got[0], None, x0s[0], None)], 0.5, 0.8, l2)
    assert torch.equal(got[0], keep)


def test_pgd_attack_steps_all_groups_in_one_launch_with_the_same_result():
    from gsplat_attack import attack as A
    from gsplat_attack.attack import pgd_attack, SurrogateDetector
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("hydrant-1k", device=dev, n_views=3)
    det = SurrogateDetector().to(dev)
    res = []
    for multi in (True, False):
        for norm in ("l2", "linf"):
            m = model.clone()
            old = A.MULTI_STEP
            A.MULTI_STEP = multi
            try:
                pgd_attack(m, cams, iters=3, groups=("color", "position", "scaling", "rotation", "opacity"), loss_fn=det,
                           alpha=0.05, epsilon=0.3, norm=norm)
            finally:
                A.MULTI_STEP = old
            res.append({n: getattr(m, n).detach().clone() for n in ("_xyz", "_features_dc", "_features_rest", "_opacity",
                                                                    "_scaling", "_rotation")})
    for a, b in ((res[0], res[2]), (res[1], res[3])):       # same arithmetic per tensor: bit-equal
        for n in a:
            assert torch.equal(a[n], b[n]), n
    assert not torch.equal(res[0]["_xyz"], model._xyz.detach()) and not torch.equal(res[1]["_rotation"], model._rotation.detach())


@pytest.mark.parametrize("kind", ["linf", "l2"])
def test_fused_pgd_step_on_tensors_that_start_off_a_16_byte_boundary(kind):
    """Gradients that are slices of a flat bucket start at a multiple of P floats -- 4-byte aligned only for odd P: the
    kernels' 16-byte accesses must fall back, not fault or read the wrong floats."""
    from gsplat_attack import pgd
    g = torch.Generator().manual_seed(11)
    rows, cols = 4099, 45
    step = pgd.l2_step_ if kind == "l2" else pgd.linf_step_
    for off in (1, 2, 3):
        x, grad = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g) * 2.0
        x0 = x + torch.randn(rows, cols, generator=g)
        want = x.clone()
        step(want, grad, 0.5, 0.8, x0)

        def shifted(t):
            buf = torch.zeros(t.numel() + 4, device="cuda")
            v = buf[off:off + t.numel()].view(rows, cols)
            v.copy_(t)
            assert v.data_ptr() % 16 == 4 * off and v.is_contiguous()
            return v
        got = shifted(x)
        step(got, shifted(grad), 0.5, 0.8, shifted(x0))
        torch.cuda.synchronize()
        assert torch.allclose(got.cpu(), want, atol=2e-6, rtol=1e-5), (off, kind)


def test_render_with_a_reference_style_pipe_object_takes_the_fused_path():
    """A pipe object with only the reference's three switches (attack.py:254-256) -- what render() sees after
    gsplat_attack.patch_reference() -- still gets the fused raw-parameter path: the image equals the classic surface's
    and gradients land on the raw parameters."""
    from gsplat_attack.renderer import PipelineParams, render
    from gsplat_attack.scenes import make_scene

    class GroupParams:                       # the reference's bare attribute bag
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("hydrant-1k", device=dev, n_views=1)
    bg = torch.zeros(4, device=dev)          # the reference passes four zeros for black (attack.py:396)
    out = render(cams[0], model, GroupParams(), bg)
    assert type(out["render"].grad_fn).__name__ == "_RasterizeGaussiansRawBackward"
    ref = render(cams[0], model, PipelineParams(fused_activations=False), bg)
    assert (out["render"] - ref["render"]).abs().max().item() < 2e-6
    assert (out["render_object"] - ref["render_object"]).abs().max().item() < 2e-6
    out["render"].sum().backward()
    assert model._scaling.grad is not None and out["viewspace_points"].grad is not None


def test_render_pair_equals_the_concatenated_scene_with_objects_and_radii():
    """gsr_forward_raw2 (two parameter sets side by side) against render() of the explicitly concatenated model:
    image, object map and radii bit for bit, at a size where the lists are long (segments, both tile splits)."""
    from gsplat_attack.attack import combine_with_background
    from gsplat_attack.renderer import PipelineParams, render, render_pair
    from gsplat_attack.scenes import make_scene
    model, cams, _ = make_scene("nyc-1M", device="cuda", P=60000, width=640, height=360, n_views=2)
    P = model.get_xyz.shape[0]
    cut = 2 * P // 5 + 3                                   # not a multiple of 64: a wave straddles the two segments
    mask = torch.zeros(P, dtype=torch.bool, device="cuda")
    mask[:cut] = True
    target, background = model.clone(), model.clone()
    target.removal_setup(~mask)                            # first `cut` rows
    background.removal_setup(mask)                         # the rest
    bg = torch.tensor([0.3, 0.1, 0.2], device="cuda")
    whole = combine_with_background(target, background)
    for objects in (False, True):
        pipe = PipelineParams(skip_objects=not objects)
        with torch.no_grad():
            ref = render(cams[1], whole, pipe, bg)
        got = render_pair(cams[1], target, background, pipe, bg)
        assert torch.equal(ref["render"], got["render"])
        assert torch.equal(ref["radii"], got["radii"]) and got["radii"].numel() == P
        assert torch.equal(ref["render_object"], got["render_object"])
    assert float(got["render_object"].abs().max()) > 0
    # degenerate splits
    empty = target.clone()
    empty.removal_setup(torch.ones(cut, dtype=torch.bool, device="cuda"))
    assert empty.get_xyz.shape[0] == 0
    with torch.no_grad():
        ref = render(cams[0], target, PipelineParams(skip_objects=True), bg)["render"]
    assert torch.equal(render_pair(cams[0], target, empty, PipelineParams(skip_objects=True), bg)["render"], ref)
    assert torch.equal(render_pair(cams[0], empty, target, PipelineParams(skip_objects=True), bg)["render"], ref)


def test_reference_loop_switches_match_a_straight_line_transcription():
    """pgd_attack(accumulate_grads=True, batch_loss=True) against the reference loop written out with plain PyTorch
    calls (attack.py:476-499 + :602-604 as it actually behaves): B renders stacked, ONE loss, one backward, the L2
    colour step of gsplat_attack.pgd (pinned to attack.py:138-173 by the golden fixtures), .grad never cleared."""
    from gsplat_attack import pgd
    from gsplat_attack.attack import SurrogateDetector, pgd_attack
    from gsplat_attack.renderer import PipelineParams, render
    model, cams, _ = _scene(n_views=3)
    twin = model.clone()
    det = SurrogateDetector().to("cuda")
    bg = torch.tensor([0.1, 0.1, 0.1], device="cuda")
    hist = pgd_attack(model, cams, iters=2, alpha=0.5, epsilon=5.0, groups=("color",), norm="l2", bg=bg, loss_fn=det,
                      accumulate_grads=True, batch_loss=True, streams=1)
    # transcription (geometry frozen like pgd_attack does for a colour attack: same gradients on the colour tensors)
    o_rest, o_dc = twin._features_rest.detach().clone(), twin._features_dc.detach().clone()
    pipe = PipelineParams(skip_objects=True)
    want = []
    for it in range(2):
        renders = torch.stack([render(cam, twin, pipe, bg)["render"] for cam in cams])
        loss = det(renders)
        loss.backward()                                    # .grad accumulates: nothing zeroes it (attack.py:602 is a no-op)
        want.append(float(loss))
        pgd.gaussian_color_l2_attack(twin, 0.5, 5.0, o_rest, o_dc)
    torch.cuda.synchronize()
    assert hist == pytest.approx(want, rel=1e-5)
    for n in ("_features_dc", "_features_rest"):
        a, b = getattr(model, n).detach(), getattr(twin, n).detach()
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item()), n
    # and the switches matter: zeroing the gradients each iteration gives a different second step
    third = twin.clone()
    for n, o in (("_features_rest", o_rest), ("_features_dc", o_dc)):
        getattr(third, n).data.copy_(o)
    pgd_attack(third, cams, iters=2, alpha=0.5, epsilon=5.0, groups=("color",), norm="l2", bg=bg, loss_fn=det,
               batch_loss=True, streams=1)
    assert (third._features_rest.detach() - twin._features_rest.detach()).abs().max().item() > 1e-4


def test_success_bookkeeping_stops_the_loop_and_saves_the_model(tmp_path):
    """Success flags per view after every step from the target + background re-render; the loop stops at B-1 successes
    and writes the attacked model (attack.py:556-569)."""
    from gsplat_attack.attack import pgd_attack, run_attack
    from gsplat_attack.gaussian_model import GaussianModel
    model, cams, _ = _scene(n_views=3)
    P = model.get_xyz.shape[0]
    mask = torch.zeros(P, dtype=torch.bool, device="cuda")
    mask[: P // 2] = True
    target, background = model.clone(), model.clone()
    target.removal_setup(~mask)
    background.removal_setup(mask)
    seen = []

    def success(img, view):                                # "fooled" once the image moved away from its first version
        seen.append((view, tuple(img.shape)))
        first.setdefault(view, img.clone())
        return bool((img - first[view]).abs().mean() > 2e-4)
    first = {}
    recs = []
    path = str(tmp_path / "adv.ply")
    hist = pgd_attack(target, cams, iters=12, background=background, success_fn=success, save_path=path, log=recs.append,
                      streams=1)
    assert 1 < len(hist) < 12 and sum(recs[-1]["successes"]) >= 2 and not any(recs[0]["successes"])
    assert all(s == (3, 192, 320) for _, s in seen)
    back = GaussianModel.load_ply(path, device="cuda")
    assert torch.allclose(back._features_dc, target._features_dc.detach(), atol=1e-6)
    # the batch schedule around it: 3 views in batches of 2
    first.clear()
    rep = run_attack(target.clone(), cams, background=background, batch_size=2, max_iters=10, success_fn=success, streams=1,
                     truncate=False)
    assert [b["views"] for b in rep["batches"]] == [[0, 1], [2]] and rep["all_succeeded"] and rep["saved"]
    first.clear()
    rep = run_attack(target.clone(), cams, background=background, batch_size=2, max_iters=10, success_fn=success, streams=1)
    assert [b["views"] for b in rep["batches"]] == [[0, 1]] and rep["all_succeeded"]      # the third view is truncated


def test_run_attack_with_yawed_views_and_benign_boxes():
    """run_attack(add_cams=3, benign=True): the two yawed copies of view 0 join the batch (attack.py:404-415) and the benign
    pass (black background, luma > 20, bounding box: attack.py:434-461) returns one box per view, equal to the box of a
    direct black-background render."""
    from gsplat_attack.attack import bbox_from_render, run_attack
    from gsplat_attack.renderer import PipelineParams, render
    model, cams, _ = _scene(n_views=1)
    want = bbox_from_render(render(cams[0], model, PipelineParams(skip_objects=True), torch.zeros(3, device="cuda"))["render"])
    seen = []
    rep = run_attack(model, cams, batch_size=3, max_iters=3, add_cams=3, benign=True, streams=1,
                     bg=torch.ones(3, device="cuda"), success_fn=lambda im, i: (seen.append(i), True)[1])
    assert sorted(set(seen)) == [0, 1, 2] and rep["all_succeeded"] and len(rep["gt_bboxes"]) == 3
    assert rep["gt_bboxes"][0] is not None and rep["gt_bboxes"][0] == want    # taken before the first step
    l, u, r, b = rep["gt_bboxes"][0]
    assert 0 <= l < r <= cams[0].image_width and 0 <= u < b <= cams[0].image_height
    # (the city block fills the frame from every one of these views: the boxes of the yawed views need not differ)
    for box in rep["gt_bboxes"][1:]:
        assert box is not None and 0 <= box[0] < box[2] <= cams[0].image_width and 0 <= box[1] < box[3] <= cams[0].image_height


def test_benign_boxes_cover_the_kept_views_only():
    """The reference truncates viewpoint_stack to a multiple of the batch size BEFORE its benign pass (attack.py:417-423,
    434-461): four views at batch size 3 give three boxes; and the benign pass keeps no rasteriser context even when the
    pipe it is handed carries a RenderCache."""
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.attack import benign_bboxes, run_attack
    from gsplat_attack.renderer import PipelineParams
    model, cams, _ = _scene(n_views=1)
    rep = run_attack(model, cams, batch_size=3, max_iters=3, add_cams=4, benign=True, streams=1,
                     bg=torch.ones(3, device="cuda"), success_fn=lambda im, i: True)
    assert len(rep["gt_bboxes"]) == 3 and [b["views"] for b in rep["batches"]] == [[0, 1, 2]]
    cache = RenderCache()
    boxes = benign_bboxes(model, cams, PipelineParams(skip_objects=True, render_cache=cache))
    assert len(boxes) == 1 and boxes[0] == rep["gt_bboxes"][0] and len(cache.entries) == 0
