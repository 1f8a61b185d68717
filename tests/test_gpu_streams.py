"""Views pipelined over several HIP streams give exactly what the one-stream order gives."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(P=30000, W=320, H=240, n_views=6):
    from gsplat_attack.scenes import make_scene
    from gsplat_attack.renderer import PipelineParams
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("hydrant-full", device=dev, P=P, width=W, height=H, n_views=n_views)
    pipe = PipelineParams(skip_objects=True)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    gcs = [torch.randn(3, H, W, generator=torch.Generator().manual_seed(10 + i)).to(dev) for i in range(n_views)]
    return dev, model, cams, pipe, bg, gcs


def _grads(model):
    from gsplat_attack import dist as gdist
    return {n: getattr(model, n).grad.detach().clone() for n in gdist.ATTACK_PARAMS}


def test_per_view_results_identical_across_streams():
    from gsplat_attack.renderer import render
    from gsplat_attack.streams import StreamRing
    dev, model, cams, pipe, bg, gcs = _setup()
    ref = []
    for cam, gc in zip(cams, gcs):
        model.zero_grad()
        out = render(cam, model, pipe, bg)
        out["render"].backward(gc)
        ref.append((out["render"].detach().clone(), out["radii"].clone(), _grads(model)))
    torch.cuda.synchronize()
    for n_streams in (2, 3):
        ring = StreamRing(n_streams, dev)
        got = []
        for rep in range(2):                      # second round re-uses each stream's workspace blocks
            got.clear()
            for cam, gc in zip(cams, gcs):
                with ring.next():
                    model.zero_grad()
                    out = render(cam, model, pipe, bg)
                    out["render"].backward(gc)
                    got.append((out["render"].detach(), out["radii"], _grads(model)))
            ring.join()
        torch.cuda.synchronize()
        for (img0, rad0, g0), (img1, rad1, g1) in zip(ref, got):
            assert torch.equal(img0, img1)
            assert torch.equal(rad0, rad1)
            for n in g0:
                assert torch.equal(g0[n], g1[n]), n


def test_accumulated_batch_gradient_matches():
    from gsplat_attack.renderer import render
    from gsplat_attack.streams import StreamRing
    dev, model, cams, pipe, bg, gcs = _setup()
    model.zero_grad()
    for cam, gc in zip(cams, gcs):
        render(cam, model, pipe, bg)["render"].backward(gc)
    torch.cuda.synchronize()
    want = _grads(model)
    ring = StreamRing(3, dev)
    model.zero_grad()
    for cam, gc in zip(cams, gcs):
        with ring.next():
            render(cam, model, pipe, bg)["render"].backward(gc)
    ring.join()
    torch.cuda.synchronize()
    got = _grads(model)
    for n in want:
        scale = want[n].abs().max().item() + 1e-12
        assert (want[n] - got[n]).abs().max().item() <= 1e-5 * scale, n     # summation order over views may differ


def test_pgd_attack_same_history_with_and_without_streams():
    from gsplat_attack.attack import pgd_attack
    dev, model, cams, pipe, bg, _ = _setup(P=8000, W=160, H=128, n_views=4)
    m1, m3 = model.clone(), model.clone()
    # (batched=False: the stream ring deals the per-view loop's views; a batch goes through one launch chain on one stream)
    h1 = pgd_attack(m1, cams, iters=3, groups=("color", "position"), streams=1, batched=False)
    h3 = pgd_attack(m3, cams, iters=3, groups=("color", "position"), streams=3, batched=False)
    torch.cuda.synchronize()
    for a, b in zip(h1, h3):
        assert abs(a - b) <= 1e-4 * max(abs(a), 1.0)
    # same per-view gradients summed in another order (one bucket per stream, folded once): the normalised steps agree to
    # a small fraction of the distance the positions moved
    moved = (m1._xyz - model._xyz).abs().max().item()
    assert moved > 1e-3
    assert (m1._xyz - m3._xyz).abs().max().item() <= 2e-2 * moved


@pytest.mark.parametrize("n_views,streams", [(1, 1), (3, 1), (3, 2)])
def test_overlapped_success_check_equals_the_serial_loop(n_views, streams, tmp_path):
    """pgd_attack(overlap_success=True): the success re-render of iteration i runs on a side stream beside the forward of
    iteration i + 1, which is dropped if the batch turns out to be done.  Same iteration count, bit-equal history, flags
    and parameters as the strictly serial loop -- with a detector that is never fooled, one fooled on iteration 3, and one
    fooled at once."""
    from gsplat_attack.attack import pgd_attack
    dev, model, cams, pipe, bg, _ = _setup(P=8000, W=160, H=128, n_views=n_views)
    base = model.clone()
    bg_model = model.clone()
    for stop_at in (None, 3, 1):
        runs = []
        for overlap in (False, True):
            m = base.clone()
            calls = []

            def success(im, i, calls=calls):
                assert torch.isfinite(im).all()
                calls.append(i)
                it = (len(calls) - 1) // n_views + 1          # the iteration whose renders these are
                return stop_at is not None and it >= stop_at
            recs = []
            path = str(tmp_path / f"m_{overlap}_{stop_at}.ply")
            # (one stream: the views of an iteration as ONE batch; more: the per-view loop dealt over the stream ring)
            hist = pgd_attack(m, cams, iters=5, groups=("color", "position"), streams=streams, success_fn=success,
                              background=bg_model, overlap_success=overlap, log=recs.append, save_path=path,
                              batched=streams == 1)
            torch.cuda.synchronize()
            runs.append((hist, [r.get("successes") for r in recs], {n: getattr(m, n).detach().clone() for n in
                                                                      ("_xyz", "_features_dc", "_features_rest")},
                         len(calls), pgd_attack.last_successes, __import__("os").path.exists(path)))
        (h0, f0, p0, c0, l0, s0), (h1, f1, p1, c1, l1, s1) = runs
        assert len(h0) == (5 if stop_at is None else stop_at)
        assert h0 == h1 and f0 == f1 and c0 == c1 and l0 == l1 and s0 == s1 == (stop_at is not None)
        for n in p0:
            assert torch.equal(p0[n], p1[n]), (n, stop_at)
