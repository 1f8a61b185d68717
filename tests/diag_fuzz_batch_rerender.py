"""Diagnostic (not a test): kept contexts of a BATCH of views (gsr_ctx_rerender on gsr_forward_raw_batch /
gsr_forward_raw2_batch contexts) on the random draws of tests/diag_fuzz_batch.py, with random extension flags and per-view
backgrounds.  Per draw, over three colour steps:
  * the cached batch render + backward equals the uncached batch bit for bit -- images, radii, every gradient, the per-view
    screen-space gradients -- with the geometry frozen (colour-only backward) or every attribute differentiated, gradients
    summed (GradBucket-free autograd path) or per view (GradBucketSet);
  * every image of the cached batch equals the single-view render of its camera;
  * the scene split in two models: the cached pair batch (render of target + background, forward only) equals the
    per-camera pair renders, the second model's colours stepped in the last iteration.

    python tests/diag_fuzz_batch_rerender.py first_seed last_seed
"""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa: F401,E402
import diag_fuzz_batch as F  # noqa: E402

COL = ("_features_dc", "_features_rest")
ALL = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def batch_fwd_bwd(model, sts, gcs, names, cache, key, per_view):
    import diff_gaussian_rasterization as D
    dev = model.get_xyz.device
    P, B = int(model.get_xyz.shape[0]), len(sts)
    model.zero_grad()
    geom = "_xyz" in names
    vsp = torch.zeros(B, P, 3, device=dev, requires_grad=True) if geom else None
    bset = D.GradBucketSet(B, P, dev) if (per_view and geom) else None
    if bset is not None:
        bset.flat.fill_(float("nan"))
    image, radii = D.rasterize_gaussians_raw_batch(model._xyz, vsp, model._features_dc, model._features_rest, model._opacity,
                                                   model._scaling, model._rotation, sts, grad_bucket=bset, cache=cache,
                                                   cache_key=key)
    image.backward(torch.stack(gcs))
    torch.cuda.synchronize()
    if bset is not None:
        grads = {"bucket_set": bset.flat.clone()}
    else:
        grads = {n: getattr(model, n).grad.detach().clone() for n in names}
    if vsp is not None:
        grads["viewspace"] = vsp.grad.detach().clone()
    return image.detach().clone(), radii.clone(), grads


def one(seed, dev):
    import diff_gaussian_rasterization as D
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack import renderer as R
    from gsplat_attack.gaussian_model import GaussianModel
    from gsplat_attack.renderer import PipelineParams, render
    model, cams, bgs, gcs, scale, desc = F.draw(seed, dev)
    B = len(cams)
    g = torch.Generator().manual_seed(seed + 4242)
    r = lambda n: int(torch.randint(0, n, (), generator=g))   # noqa: E731
    color_only, per_view = bool(r(2)), bool(r(2))
    flags = F.draw_flags(seed)
    names = COL if color_only else ALL
    if color_only:
        for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
            getattr(model, n).requires_grad_(False)
    pipe = PipelineParams(skip_objects=True)
    cache = RenderCache()
    P = int(model.get_xyz.shape[0])
    Pa = max(1, min(P - 1, r(max(P, 1)))) if P >= 2 else 0
    with D.extra_flags(flags):
        sts = [R._settings(c, model, pipe, bgs[v], scale) for v, c in enumerate(cams)]
        for it in range(3):
            want = batch_fwd_bwd(model, sts, gcs, names, None, None, per_view)
            got = batch_fwd_bwd(model, sts, gcs, names, cache, "k", per_view)
            assert torch.equal(want[0], got[0]), f"iteration {it}: images"
            assert torch.equal(want[1], got[1]), f"iteration {it}: radii"
            assert want[2].keys() == got[2].keys(), f"iteration {it}: gradient set"
            for n in want[2]:
                assert torch.equal(want[2][n], got[2][n]), f"iteration {it}: {n}"
            with torch.no_grad():
                v = r(B)
                one_v = render(cams[v], model, pipe, bgs[v], scale)["render"]
                assert torch.equal(got[0][v], one_v), f"iteration {it}: view {v} against its single-view render"
                if Pa:
                    # target + background as two models of the same scene, forward only, through a kept pair-batch context
                    def part(sl):
                        return tuple(t[sl].detach().clone() for t in (model._xyz, model._features_dc, model._features_rest,
                                                                      model._opacity, model._scaling, model._rotation))
                    if it == 0:
                        pa, pb = part(slice(0, Pa)), part(slice(Pa, P))
                    else:
                        pa[1].copy_(model._features_dc[:Pa]); pa[2].copy_(model._features_rest[:Pa])
                        if it == 2:
                            pb[1].copy_(model._features_dc[Pa:]); pb[2].copy_(model._features_rest[Pa:])
                    pim, prad = D.rasterize_gaussians_raw2_batch(pa, pb, sts, cache=cache, cache_key="p")
                    if it < 2:
                        # (the second model still holds the colours of iteration 0: compare with the single-view pair render)
                        def with_obj(p):
                            return (p[0], p[1], p[2], None, p[3], p[4], p[5])
                        o = D.rasterize_gaussians_raw2(with_obj(pa), with_obj(pb), sts[v], objects=False)
                        assert torch.equal(pim[v], o[0]), f"iteration {it}: pair batch view {v} against its pair render"
                        assert torch.equal(prad[v], o[1]), f"iteration {it}: pair batch radii"
                    else:
                        assert torch.equal(pim, got[0]), "pair batch of the split scene against the batch of the whole scene"
                        assert torch.equal(prad, got[1]), "pair batch radii against the whole scene's"
            with torch.no_grad():                          # a colour step: the geometry tensors stay untouched
                model._features_dc.add_(torch.randn(model._features_dc.shape, generator=g).to(dev) * 0.05)
                model._features_rest.add_(torch.randn(model._features_rest.shape, generator=g).to(dev) * 0.02)
    assert cache.hits == (4 if Pa else 2), f"cache hits {cache.hits}"
    return f"{desc} color_only={color_only} per_view={per_view} Pa={Pa} flags={flags:#x}"


def main():
    dev = torch.device("cuda:0")
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    for seed in range(lo, hi):
        desc = "?"
        try:
            desc = one(seed, dev)
        except Exception as e:                               # noqa: BLE001
            bad.append(seed)
            tb = traceback.extract_tb(e.__traceback__)[-1]
            print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]} at {os.path.basename(tb.filename)}:{tb.lineno}", flush=True)
        if seed % 20 == 0:
            print(f"... seed {seed} ({desc})", flush=True)
    print(f"batch rerender fuzz seeds [{lo}, {hi}): {hi - lo - len(bad)} of {hi - lo} draws clean")
    print("failed seeds:", bad)


if __name__ == "__main__":
    main()
