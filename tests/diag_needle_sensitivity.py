"""Diagnostic (CPU only, oracle-R in float64): what a float32 CONIC RECORD costs a needle's gradients.

    [RNG=2] python tests/diag_needle_sensitivity.py [SEED GAUSSIAN]      (default: aniso fuzz seed 41, Gaussian 59 -- 440:1)

The conic of ONE Gaussian is perturbed (detached: the backward still differentiates the exact chain, like needle_bwd_d) and
the change of that Gaussian's gradients is printed:
  round  every entry rounded to float32 independently (<= half an ulp each)
  axis   the float32 neighbour (+-RNG ulps per entry) that keeps u^T K u along the long axis u (gsr_math.h needle_conic_to_float)
  scale  a common factor 1 + 3e-6 (what the float32 chain's error in the determinant does)
Round 5 finding: 'round' moves the rotation gradient by 2e-3, 'scale' by 4e-4 (what the float32 oracle and the float32 HIP
chain both show on this splat), 'axis' by 2e-4 -- a float32 triple cannot hold the conic's small eigenvalue (2e-3 next to
entries ~1) to better than ~1e-5, and a needle's rotation gradient amplifies that by ~250."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa
import torch
from oracle import oracle_r as O
from util import settings_for
from fuzz_cases import aniso_case
seed, gi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (41, 59)
inp, cam, bg, kw, desc = aniso_case(seed)
H, W = cam.image_height, cam.image_width
g = torch.Generator().manual_seed(kw["seed"])
gc = torch.randn(3, H, W, generator=g)
st = settings_for(cam, bg, kw["sh_degree"], kw["scale_modifier"])
ref, rg = O.forward_backward(inp, st, gc, None, dtype=torch.float64, drop_fragile=True)
gc2, _ = O.solid_grads(ref, gc, None)
orig = O._project
def run(mode):
    def proj(*a, **k):
        q = orig(*a, **k)
        c = q["conic"]
        if c.dtype == torch.float64 and mode:
            d = torch.zeros_like(c)
            if mode == "round":       # every entry of Gaussian gi's conic rounded to float32
                d[gi] = (c[gi].detach().float().double() - c[gi].detach())
            elif mode == "axis":      # float32 neighbours of the rounded triple: the one that keeps u^T K u along the needle's axis
                import itertools, numpy as np
                K = c[gi].detach()
                Km = torch.tensor([[K[0], K[1]], [K[1], K[2]]])
                w, V = torch.linalg.eigh(Km)
                u = V[:, 0]                       # eigenvector of the SMALL conic eigenvalue = the long axis
                f = K.float()
                best = None
                def step(x, n):
                    x = np.float32(x)
                    for _ in range(abs(n)):
                        x = np.nextafter(x, np.float32(np.inf if n > 0 else -np.inf))
                    return float(x)
                RNG = int(os.environ.get("RNG", "2"))
                for da, db, dc in itertools.product(range(-RNG, RNG + 1), repeat=3):
                    cand = torch.tensor([step(f[0], da), step(f[1], db), step(f[2], dc)], dtype=torch.float64)
                    e = cand - K
                    err = abs(u[0] ** 2 * e[0] + 2 * u[0] * u[1] * e[1] + u[1] ** 2 * e[2])
                    if best is None or err < best[0]:
                        best = (err, e)
                e0 = f.double() - K
                print("axis: plain rounding u^T dK u", float(u[0] ** 2 * e0[0] + 2 * u[0] * u[1] * e0[1] + u[1] ** 2 * e0[2]), "best", float(best[0]), "k_b", float(w[0]))
                d[gi] = best[1]
            elif mode == "scale":     # a common factor (what an error in det does)
                d[gi] = c[gi].detach() * 3e-6
            q = dict(q); q["conic"] = c + d
        return q
    O._project = proj
    try:
        return O.forward_backward(inp, st, gc2, None, dtype=torch.float64, drop_fragile=False)[1]
    finally:
        O._project = orig
base = run(None)
for mode in ("round", "axis", "scale"):
    gr = run(mode)
    for k in ("rotations", "scales", "means3D", "opacities"):
        b, p = base[k][gi].flatten(), gr[k][gi].flatten()
        print(mode, k, [f"{abs((p[j]-b[j])/b[j]).item():.2e}" for j in range(b.numel())])
