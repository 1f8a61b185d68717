"""Diagnostic (not a test): draws of tests/diag_fuzz_batch.py whose SUMMED batch gradients left the suite's yardstick (the
double sum of the float32 single-view gradients), held against oracle-R in float64: is the batch further from the exact
gradient than the per-view loop is, or are both equally far and merely rounded differently?

Per seed and attribute group: |loop - oracle|, |batch - oracle|, |batch - loop| (largest element, relative to the group's
largest gradient), and at the element where batch and loop differ most: the three values.  dL/dC is zeroed on every view's
fragile pixels on all three sides (the oracle's own flag), as the suite does.

    python tests/diag_fuzz_batch_oracle.py seed [seed ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa: F401,E402
import diag_fuzz_batch as F  # noqa: E402
from oracle import oracle_r as O  # noqa: E402
from util import settings_for  # noqa: E402


def main():
    import diff_gaussian_rasterization as D
    from gsplat_attack import renderer as R
    from gsplat_attack.renderer import PipelineParams, render
    dev = torch.device("cuda:0")
    torch.set_num_threads(16)
    for seed in [int(a) for a in sys.argv[1:]]:
        model, cams, bgs, gcs, scale, desc = F.draw(seed, dev)
        P, B = int(model.get_xyz.shape[0]), len(cams)
        from gsplat_attack.gaussian_model import GaussianModel
        ref = GaussianModel.from_tensors(model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation,
                                         model._opacity, device="cpu")
        ref.active_sh_degree = model.active_sh_degree
        solid = []
        for v, cam in enumerate(cams):
            st = settings_for(cam, bgs[v], sh_degree=model.active_sh_degree, scale_modifier=scale, device="cpu")
            ro = O.rasterize(ref.get_xyz, None, ref.get_opacity, st, shs=ref.get_features, scales=ref.get_scaling,
                             rotations=ref.get_rotation)
            gk = gcs[v].cpu().double() * (~ro.fragile_px).double()
            if ro.color.requires_grad:                        # (a view that sees nothing has no graph)
                (ro.color * gk).sum().backward()
            solid.append(gk.float().to(dev))
        pipe = PipelineParams(skip_objects=True)
        loop = D.GradBucket(P, dev)
        for v, cam in enumerate(cams):
            render(cam, model, PipelineParams(skip_objects=True, grad_bucket=loop), bgs[v], scale)["render"].backward(solid[v])
        bat = D.GradBucket(P, dev)
        sts = [R._settings(c, model, pipe, bgs[v], scale) for v, c in enumerate(cams)]
        vsp = torch.zeros(B, P, 3, device=dev, requires_grad=True)
        image, _ = D.rasterize_gaussians_raw_batch(model._xyz, vsp, model._features_dc, model._features_rest, model._opacity,
                                                   model._scaling, model._rotation, sts, grad_bucket=bat)
        image.backward(torch.stack(solid))
        torch.cuda.synchronize()
        print(f"seed {seed} ({desc}, scale_modifier {scale:.2f}):")
        for name, gl, gb in zip(loop.NAMES, loop.slices(), bat.slices()):
            go = getattr(ref, name).grad
            go = torch.zeros(gl.numel(), dtype=torch.float64) if go is None else go.reshape(-1).double()
            gl, gb = gl.double().cpu(), gb.double().cpu()
            s = go.abs().max().item() or 1.0
            i = int((gb - gl).abs().argmax())
            print(f"  {name:15s} max|g| {s:9.3e}   loop-oracle {((gl - go).abs().max() / s):.2e}   batch-oracle "
                  f"{((gb - go).abs().max() / s):.2e}   batch-loop {((gb - gl).abs().max() / s):.2e}   at its worst element: "
                  f"oracle {go[i]:+.6e} loop {gl[i]:+.6e} batch {gb[i]:+.6e} (element {i})", flush=True)


if __name__ == "__main__":
    main()
