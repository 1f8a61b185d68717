"""Seeded random configurations shared by the suite and the diagnostic sweeps (tests/diag_fuzz_*.py)."""
import math

import torch


def aniso_case(seed: int):
    """Strongly anisotropic splats: log-normal scales with sigma 1.2 .. 2.2 per axis -- needles and discs of 10:1 to
    1000:1 -- where the published float32 arithmetic itself gives out (EXPERIMENTS.md, round 3).
    -> (inputs dict, camera, background, kwargs for test_gpu_parity.check, description)."""
    from gsplat_attack.cameras import look_at_camera
    g = torch.Generator().manual_seed(seed)

    def u(lo, hi):
        return lo + (hi - lo) * torch.rand((), generator=g).item()
    P = int(round(math.exp(u(math.log(20.0), math.log(3000.0)))))
    W, H = int(u(40, 230)), int(u(40, 170))
    xyz = torch.randn(P, 3, generator=g) * torch.tensor([u(0.1, 0.6), u(0.1, 0.6), u(0.1, 0.6)])
    scales = torch.exp(torch.randn(P, 3, generator=g) * u(1.2, 2.2) + math.log(u(0.01, 0.05))).clamp(max=1.5)
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * u(0.5, 3.0) + u(-2.0, 2.0))
    shs = torch.randn(P, 16, 3, generator=g) * u(0.05, 0.5)
    shs[:, 0] += torch.randn(P, 3, generator=g)
    inp = dict(means3D=xyz, shs=shs, opacities=opac, scales=scales, rotations=rots)
    dist = u(1.5, 3.5)
    th, ph = u(0, 2 * math.pi), u(-0.6, 0.6)
    eye = (dist * math.cos(th) * math.cos(ph), dist * math.sin(ph), dist * math.sin(th) * math.cos(ph))
    cam = look_at_camera(eye, (u(-0.1, 0.1), u(-0.1, 0.1), u(-0.1, 0.1)), fovx=u(0.3, 1.4), width=W, height=H)
    bg = torch.rand(3, generator=g)
    kw = dict(sh_degree=int(u(0, 3.999)), scale_modifier=u(0.5, 1.8), with_gobj=False, seed=seed)
    return inp, cam, bg, kw, f"P={P} {W}x{H}"
