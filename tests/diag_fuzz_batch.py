"""Diagnostic (not a test): random configurations of a BATCH of views against the single-view calls.

Per draw: P (1 .. 40 000, every sixth draw next to a multiple of the batch's 16 384-Gaussian padding), image size (mostly not
multiples of 16), B (1 .. 16), per-view camera / field of view / background, active SH degree, scale modifier, some cameras
that see nothing.  Checked, as tests/test_gpu_batch.py does on its fixed cases:
  * images, radii and screen-space gradients of every view bit-equal to the single-view render,
  * per-view gradients (GradBucketSet, gsr_backward_raw_batch_views) bit-equal to the single-view backward,
  * summed gradients (gsr_backward_raw_batch_into) against the double sum of the single-view gradients (test's yardstick).

    python tests/diag_fuzz_batch.py first_seed last_seed          (FUZZ_FLAGS=1: random extension flags per draw, see draw_flags)
"""
import math
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa: F401,E402
import test_gpu_batch as TB  # noqa: E402


def draw(seed: int, dev):
    from gsplat_attack.cameras import look_at_camera
    from gsplat_attack.gaussian_model import GaussianModel
    g = torch.Generator().manual_seed(seed)

    def u(lo, hi):
        return lo + (hi - lo) * torch.rand((), generator=g).item()
    P = int(round(math.exp(u(0.0, math.log(40000.0)))))
    if seed % 6 == 0:
        P = (16383, 16384, 16385, 32767, 32768, 32769)[(seed // 6) % 6]
    W, H = int(u(17, 420)), int(u(17, 300))
    if seed % 5 == 0:
        W, H = 16 * (W // 16 + 1), 16 * (H // 16 + 1)
    B = int(u(1, 16.999))
    spread = torch.tensor([u(0.1, 0.8), u(0.1, 0.8), u(0.1, 0.8)])
    xyz = torch.randn(P, 3, generator=g) * spread
    log_s = torch.randn(P, 3, generator=g) * u(0.2, 1.2) + math.log(u(0.004, 0.06))
    rot = torch.randn(P, 4, generator=g)
    opac = torch.randn(P, 1, generator=g) * u(0.5, 3.0) + u(-2.0, 2.0)
    f_dc = torch.randn(P, 1, 3, generator=g)
    f_rest = torch.randn(P, 15, 3, generator=g) * u(0.02, 0.4)
    model = GaussianModel.from_tensors(xyz, f_dc, f_rest, log_s.clamp(max=math.log(1.5)), rot, opac, device=dev)
    model.active_sh_degree = int(u(0, 3.999))
    cams, bgs = [], []
    for v in range(B):
        dist = u(1.2, 5.0)
        th, ph = u(0, 2 * math.pi), u(-0.7, 0.7)
        eye = (dist * math.cos(th) * math.cos(ph), dist * math.sin(ph), dist * math.sin(th) * math.cos(ph))
        target = (u(-0.2, 0.2), u(-0.2, 0.2), u(-0.2, 0.2))
        if u(0, 1) < 0.08:                                    # a camera that looks away from the scene
            target = tuple(2.0 * e for e in eye)
        cams.append(look_at_camera(eye, target, fovx=u(0.3, 1.5), width=W, height=H, uid=v, device=dev))
        bgs.append(torch.rand(3, generator=g).to(dev))
    gcs = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(B)]
    return model, cams, bgs, gcs, u(0.5, 1.6), f"P={P} {W}x{H} B={B} deg={model.active_sh_degree}"


def draw_flags(seed: int) -> int:
    """FUZZ_FLAGS=1: extension flags for the draw -- the same for the batch and for the single-view calls it is held to (which
    must stay bit-equal under every combination): footprint cull off, segment records off, the forward's staging shared by
    a tile's waves, no side stream for the colour kernel, forced forward / backward splits and tile maps."""
    import diff_gaussian_rasterization as D
    g = torch.Generator().manual_seed(seed + 77_777)
    r = lambda n: int(torch.randint(0, n, (), generator=g))   # noqa: E731
    f = 0
    if r(3) == 0: f |= D.FLAG_NO_CULL
    if r(3) == 0: f |= D.FLAG_NO_SEGMENTS
    if r(4) == 0: f |= D.FLAG_FWD_SHARED
    if r(3) == 0: f |= D.FLAG_NO_SIDE_STREAM
    f |= D.flag_fwd_split((0, 0, 1, 2, 4)[r(5)])
    f |= D.flag_bwd_split((0, 0, 2, 4)[r(4)])
    if r(2) == 0: f |= D.flag_tile_map(r(4))
    return f


def per_view(model, cams, bgs, gcs, scale):
    import diff_gaussian_rasterization as D
    from gsplat_attack import renderer as R
    from gsplat_attack.renderer import PipelineParams, render
    dev = model.get_xyz.device
    P, B = int(model.get_xyz.shape[0]), len(cams)
    pipe = PipelineParams(skip_objects=True)
    bset = D.GradBucketSet(B, P, dev)
    bset.flat.fill_(float("nan"))
    sts = [R._settings(c, model, pipe, bgs[v], scale) for v, c in enumerate(cams)]
    vsp = torch.zeros(B, P, 3, device=dev, requires_grad=True)
    image, radii = D.rasterize_gaussians_raw_batch(model._xyz, vsp, model._features_dc, model._features_rest, model._opacity,
                                                   model._scaling, model._rotation, sts, grad_bucket=bset)
    image.backward(torch.stack(gcs))
    torch.cuda.synchronize()
    assert torch.isfinite(bset.flat).all(), "a float of a per-view bucket was not written"
    for v, cam in enumerate(cams):
        one = D.GradBucket(P, dev)
        o = render(cam, model, PipelineParams(skip_objects=True, grad_bucket=one), bgs[v], scale)
        o["render"].backward(gcs[v])
        assert torch.equal(image[v].detach(), o["render"].detach()), f"view {v}: image"
        assert torch.equal(radii[v], o["radii"]), f"view {v}: radii"
        assert torch.equal(bset.bucket(v).flat, one.flat), f"view {v}: per-view gradients"
        assert torch.equal(vsp.grad[v], o["viewspace_points"].grad), f"view {v}: screen-space gradient"
    return int((radii > 0).sum())


def main():
    dev = torch.device("cuda:0")
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad, seen, views = [], 0, 0
    for seed in range(lo, hi):
        desc = "?"
        try:
            import diff_gaussian_rasterization as D
            model, cams, bgs, gcs, scale, desc = draw(seed, dev)
            flags = draw_flags(seed) if os.environ.get("FUZZ_FLAGS") else 0
            if flags:
                desc += f" flags={flags:#x}"
            with D.extra_flags(flags):
                seen += per_view(model, cams, bgs, gcs, scale)
                TB._check_equal(model, cams, gcs, bg_list=bgs, scale=scale)
            views += len(cams)
        except Exception as e:                               # noqa: BLE001
            bad.append(seed)
            tb = traceback.extract_tb(e.__traceback__)[-1]
            print(f"seed {seed} ({desc}): {type(e).__name__}: {str(e)[:300]} at {os.path.basename(tb.filename)}:{tb.lineno}", flush=True)
        if seed % 20 == 0:
            print(f"... seed {seed} ({desc})", flush=True)
    print(f"batch fuzz seeds [{lo}, {hi}): {hi - lo - len(bad)} of {hi - lo} draws clean, {views} views, {seen} visible (view, Gaussian) pairs")
    print("failed seeds:", bad)


if __name__ == "__main__":
    main()
