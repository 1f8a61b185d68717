"""Diagnostic (not a test): the OTHER ways a batch's backward can be asked for, on the random draws of tests/diag_fuzz_batch.py.

Per draw one of:
  subset      a random subset of the attribute groups requires a gradient (colour only = the batch kernel without geometry,
              position only, ...): autograd's .grad after one batched backward against the per-view loop's accumulated .grad;
  accumulate  the batch's one backward ADDS to a bucket that already holds numbers (gsr_backward_raw_batch_into, accumulate = 1)
              against the loop adding view by view to the same start;
  chunks      the batch's per-Gaussian stage in 2 .. 5 ranges (gsr_backward_raw_chunked on a batch context, the all-reduce
              overlap's form) against the one-launch form;
  norms       the sums of squares the batch's backward leaves for the L2 steps (gsr_ctx_request_sumsq) against the gradient's.
The yardstick is tests/test_gpu_batch.py's: the double sum of the per-view float32 gradients; the batch may be no further from
it than three times the loop's own float32 accumulation error, or 1e-5 (1e-4: scale, rotation) of the tensor's largest gradient.

    python tests/diag_fuzz_batch_modes.py first_seed last_seed
"""
import copy
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa: F401,E402
import diag_fuzz_batch as F  # noqa: E402

NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
GROUPS = {"position": ("_xyz",), "color": ("_features_dc", "_features_rest"), "opacity": ("_opacity",),
          "scaling": ("_scaling",), "rotation": ("_rotation",)}


def yard(name, got, seq, exact):
    scale = exact.abs().max().item()
    e_seq = (seq.double() - exact).abs().max().item()
    e_bat = (got.double() - exact).abs().max().item()
    floor = 1e-4 if name in ("_scaling", "_rotation") else 1e-5
    assert e_bat <= max(3.0 * e_seq, floor * scale), f"{name}: batch {e_bat:.3e}, loop {e_seq:.3e}, scale {scale:.3e}"


def loop_grads(model, cams, bgs, gcs, scale, start=None):
    """Per-view single-view backward passes: (float32 accumulation in view order [from `start`], double sum [+ start])."""
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import PipelineParams, render
    P = int(model.get_xyz.shape[0])
    dev = model.get_xyz.device
    seq = D.GradBucket(P, dev)
    exact = torch.zeros(59 * P, dtype=torch.float64, device=dev)
    if start is not None:
        seq.flat.copy_(start)
        seq.fresh, seq.used = False, True
        exact += start.double()
    for v, cam in enumerate(cams):
        render(cam, model, PipelineParams(skip_objects=True, grad_bucket=seq), bgs[v], scale)["render"].backward(gcs[v])
        own = D.GradBucket(P, dev)
        render(cam, model, PipelineParams(skip_objects=True, grad_bucket=own), bgs[v], scale)["render"].backward(gcs[v])
        exact += own.flat.double()
    return seq, exact


def batch_call(model, cams, bgs, gcs, scale, bucket=None, norms=None):
    import diff_gaussian_rasterization as D
    from gsplat_attack import renderer as R
    from gsplat_attack.renderer import PipelineParams
    P, B = int(model.get_xyz.shape[0]), len(cams)
    pipe = PipelineParams(skip_objects=True)
    sts = [R._settings(c, model, pipe, bgs[v], scale) for v, c in enumerate(cams)]
    vsp = torch.zeros(B, P, 3, device=model.get_xyz.device, requires_grad=True)
    image, _ = D.rasterize_gaussians_raw_batch(model._xyz, vsp, model._features_dc, model._features_rest, model._opacity,
                                               model._scaling, model._rotation, sts, grad_bucket=bucket, grad_norms=norms)
    image.backward(torch.stack(gcs))
    torch.cuda.synchronize()


def one(seed, dev):
    import diff_gaussian_rasterization as D
    model, cams, bgs, gcs, scale, desc = F.draw(seed, dev)
    P = int(model.get_xyz.shape[0])
    g = torch.Generator().manual_seed(seed + 4242)
    mode = ("subset", "accumulate", "chunks", "norms")[int(torch.randint(0, 4, (), generator=g))]
    if mode == "subset":
        from gsplat_attack.renderer import PipelineParams, render
        keys = list(GROUPS)
        pick = [k for k in keys if int(torch.randint(0, 2, (), generator=g))] or [keys[int(torch.randint(0, 5, (), generator=g))]]
        want = {n for k in pick for n in GROUPS[k]}
        for n in NAMES:
            getattr(model, n).requires_grad_(n in want)
        # the loop: autograd accumulates in .grad view by view; the exact sum from per-view clones
        exact = {n: torch.zeros_like(getattr(model, n), dtype=torch.float64) for n in want}
        for v, cam in enumerate(cams):
            m1 = model.clone()
            for n in NAMES:
                getattr(m1, n).requires_grad_(n in want)
            render(cam, m1, PipelineParams(skip_objects=True), bgs[v], scale)["render"].backward(gcs[v])
            render(cam, model, PipelineParams(skip_objects=True), bgs[v], scale)["render"].backward(gcs[v])
            for n in want:
                if getattr(m1, n).grad is not None:
                    exact[n] += getattr(m1, n).grad.double()
        seq = {n: getattr(model, n).grad.clone() for n in want}
        for n in NAMES:
            getattr(model, n).grad = None
        batch_call(model, cams, bgs, gcs, scale)
        for n in NAMES:
            gr = getattr(model, n).grad
            if n in want:
                assert gr is not None, f"{n}: no gradient from the batch"
                yard(n, gr, seq[n], exact[n])
            else:
                assert gr is None, f"{n}: a gradient nobody asked for"
        return f"{desc} subset={'+'.join(pick)}"
    if mode == "accumulate":
        start = torch.randn(59 * P, generator=g).to(dev) * 3.0
        seq, exact = loop_grads(model, cams, bgs, gcs, scale, start=start)
        b = D.GradBucket(P, dev)
        b.flat.copy_(start)
        b.fresh, b.used = False, True
        batch_call(model, cams, bgs, gcs, scale, bucket=b)
        for name, s1, s2, c0, c1 in zip(b.NAMES, seq.slices(), b.slices(), b.CUTS[:-1], b.CUTS[1:]):
            yard(name, s2, s1, exact[c0 * P:c1 * P])
        return f"{desc} accumulate"
    if mode == "chunks":
        k = int(torch.randint(2, 6, (), generator=g))
        seq, exact = loop_grads(model, cams, bgs, gcs, scale)
        b = D.GradBucket(P, dev)
        b.flat.fill_(float("nan"))
        seen = []
        b.chunks, b.on_chunk = k, (lambda c, g0, g1: seen.append((c, g0, g1)))
        batch_call(model, cams, bgs, gcs, scale, bucket=b)
        assert torch.isfinite(b.flat).all(), "a float of the chunked batch's bucket was not written"
        if P > 0:
            assert seen and seen[0][1] == 0 and seen[-1][2] == P and all(a[2] == c[1] for a, c in zip(seen, seen[1:])), seen
        for name, s1, s2, c0, c1 in zip(b.NAMES, seq.slices(), b.slices(), b.CUTS[:-1], b.CUTS[1:]):
            yard(name, s2, s1, exact[c0 * P:c1 * P])
        return f"{desc} chunks={k}"
    # norms
    norms = D.GradNorms(dev)
    norms.begin()
    b = D.GradBucket(P, dev)
    batch_call(model, cams, bgs, gcs, scale, bucket=b, norms=norms)
    for n, sl in zip(b.NAMES, b.slices()):
        ss = norms.sumsq_of(n)
        assert ss is not None, f"{n}: the batch's backward left no sum of squares"
        ref = (sl.double() ** 2).sum().item()
        got = float(ss.item())
        assert abs(got - ref) <= 1e-5 * max(ref, 1e-30), f"{n}: sum of squares {got:.9e} against {ref:.9e}"
    return f"{desc} norms"


def main():
    dev = torch.device("cuda:0")
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad, count = [], {}
    for seed in range(lo, hi):
        try:
            desc = one(seed, dev)
            count[desc.split()[-1].split("=")[0]] = count.get(desc.split()[-1].split("=")[0], 0) + 1
        except Exception as e:                               # noqa: BLE001
            bad.append(seed)
            tb = traceback.extract_tb(e.__traceback__)[-1]
            print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]} at {os.path.basename(tb.filename)}:{tb.lineno}", flush=True)
            desc = "?"
        if seed % 20 == 0:
            print(f"... seed {seed} ({desc})", flush=True)
    print(f"batch-mode fuzz seeds [{lo}, {hi}): {hi - lo - len(bad)} of {hi - lo} draws clean; modes {count}")
    print("failed seeds:", bad)


if __name__ == "__main__":
    main()
