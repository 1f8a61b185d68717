"""Diagnostic (not a test): kept rasteriser contexts (gsr_ctx_rerender / RenderCache) on the random draws of
tests/diag_fuzz_batch.py, with random extension flags: over three colour steps every cached render + backward must equal
the uncached one bit for bit -- image, radii, every gradient -- with the geometry frozen (colour-only backward) or all
attributes differentiated.

    python tests/diag_fuzz_rerender.py first_seed last_seed
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


def fwd_bwd(cam, model, pipe, bg, gc, names, scale):
    from gsplat_attack.renderer import render
    model.zero_grad()
    out = render(cam, model, pipe, bg, scale)
    out["render"].backward(gc)
    torch.cuda.synchronize()
    return (out["render"].detach().clone(), out["radii"].clone(),
            {n: getattr(model, n).grad.detach().clone() for n in names if getattr(model, n).grad is not None})


def one(seed, dev):
    import diff_gaussian_rasterization as D
    from diff_gaussian_rasterization import RenderCache
    from gsplat_attack.renderer import PipelineParams
    model, cams, bgs, gcs, scale, desc = F.draw(seed, dev)
    cams, bgs, gcs = cams[:4], bgs[:4], gcs[:4]
    g = torch.Generator().manual_seed(seed + 99)
    color_only = bool(int(torch.randint(0, 2, (), generator=g)))
    flags = F.draw_flags(seed)
    names = COL if color_only else ALL
    if color_only:
        for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
            getattr(model, n).requires_grad_(False)
    plain = PipelineParams(skip_objects=True, viewspace_grad=not color_only)
    cached = PipelineParams(skip_objects=True, viewspace_grad=not color_only, render_cache=RenderCache())
    with D.extra_flags(flags):
        for it in range(3):
            for v, cam in enumerate(cams):
                want = fwd_bwd(cam, model, plain, bgs[v], gcs[v], names, scale)
                got = fwd_bwd(cam, model, cached, bgs[v], gcs[v], names, scale)
                assert torch.equal(want[0], got[0]), f"iteration {it} view {v}: image"
                assert torch.equal(want[1], got[1]), f"iteration {it} view {v}: radii"
                assert want[2].keys() == got[2].keys(), f"iteration {it} view {v}: gradient set"
                for n in want[2]:
                    assert torch.equal(want[2][n], got[2][n]), f"iteration {it} view {v}: {n}"
            with torch.no_grad():                          # a colour step: the geometry tensors stay untouched
                model._features_dc.add_(torch.randn(model._features_dc.shape, generator=g).to(dev) * 0.05)
                model._features_rest.add_(torch.randn(model._features_rest.shape, generator=g).to(dev) * 0.02)
    return f"{desc} views={len(cams)} color_only={color_only} flags={flags:#x}"


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
    print(f"rerender fuzz seeds [{lo}, {hi}): {hi - lo - len(bad)} of {hi - lo} draws clean")
    print("failed seeds:", bad)


if __name__ == "__main__":
    main()
