"""Diagnostic (not a test): the random-configuration parity check at sizes where tile lists are hundreds to thousands of
entries long (split lists, segment records, the whole-wave K9 path) -- 5 k .. 60 k Gaussians, images up to 500x350.
    python tests/diag_fuzz_mid.py SEED_LO SEED_HI
"""
import math, os, sys, time, traceback, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import test_gpu_parity as T
import diff_gaussian_rasterization as D
from gsplat_attack.cameras import look_at_camera

bad = []
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    g = torch.Generator().manual_seed(seed)

    def u(lo, hi):
        return lo + (hi - lo) * torch.rand((), generator=g).item()
    P = int(round(math.exp(u(math.log(5000.0), math.log(60000.0)))))
    W, H = int(u(200, 500)), int(u(150, 350))
    xyz = torch.randn(P, 3, generator=g) * torch.tensor([u(0.2, 0.6), u(0.2, 0.6), u(0.2, 0.6)])
    scales = torch.exp(torch.randn(P, 3, generator=g) * u(0.2, 0.8) + math.log(u(0.01, 0.06)))
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * u(0.5, 3.0) + u(-3.0, 1.0))
    shs = torch.randn(P, 16, 3, generator=g) * u(0.05, 0.5)
    shs[:, 0] += torch.randn(P, 3, generator=g)
    inp = dict(means3D=xyz, shs=shs, opacities=opac, scales=scales, rotations=rots)
    with_objs = seed % 3 == 0
    if with_objs:
        inp["sh_objs"] = torch.randn(P, 1, 16, generator=g) * 0.3
    dist = u(1.2, 3.5)
    th, ph = u(0, 2 * math.pi), u(-0.6, 0.6)
    eye = (dist * math.cos(th) * math.cos(ph), dist * math.sin(ph), dist * math.sin(th) * math.cos(ph))
    cam = look_at_camera(eye, (u(-0.1, 0.1), u(-0.1, 0.1), u(-0.1, 0.1)), fovx=u(0.4, 1.2), width=W, height=H)
    bg = torch.rand(3, generator=g)
    t0 = time.time()
    try:
        rep = T.check(inp, cam, bg, sh_degree=int(u(0, 3.999)), scale_modifier=u(0.7, 1.5), with_gobj=with_objs, seed=seed,
                      frag_frac=1.0, elem_frac=5e-3)
        worst = max(v[0] for v in rep.values())
        print(f"seed {seed}: P={P} {W}x{H} ok, worst normwise gradient error {worst:.2e} ({time.time() - t0:.1f} s)", flush=True)
    except Exception as e:                                   # noqa
        bad.append(seed)
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print(f"seed {seed}: P={P} {W}x{H} {type(e).__name__}: {str(e)[:300]} at {os.path.basename(tb.filename)}:{tb.lineno}", flush=True)
print("failed seeds:", bad)
