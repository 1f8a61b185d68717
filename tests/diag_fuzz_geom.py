"""Diagnostic: for one seed of the random-configuration check, which Gaussians carry the geometry-gradient error
(HIP vs oracle float64), and what is special about them."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import test_gpu_parity as T
from util import settings_for
from oracle import oracle_r as O

seed = int(sys.argv[1])
cap = {}
def spy(inp, cam, bg, **kw):
    cap.update(inp=inp, cam=cam, bg=bg, kw=kw)
    raise KeyboardInterrupt
T.check = spy
try:
    T.test_random_configurations(seed)
except KeyboardInterrupt:
    pass
inp, cam, bg, kw = cap["inp"], cap["cam"], cap["bg"], cap["kw"]
H, W = cam.image_height, cam.image_width
g = torch.Generator().manual_seed(kw["seed"])
gc = torch.randn(3, H, W, generator=g)
go = torch.randn(O.NUM_OBJECTS, H, W, generator=g) * 0.3 if kw.get("with_gobj") else None
st = settings_for(cam, bg, kw["sh_degree"], kw["scale_modifier"])
keys = T.hip_depth_keys(inp, cam, bg, kw["sh_degree"], kw["scale_modifier"])
ref, rg = O.forward_backward(inp, st, gc, go, dtype=torch.float64, drop_fragile=True, depth_key=keys)
gc2, go2 = O.solid_grads(ref, gc, go)
r32, g32 = O.forward_backward(inp, st, gc2, go2, dtype=torch.float32, depth_key=keys)
color, radii, objects, gh = T.run_hip(inp, cam, bg, gc2, go2, kw["sh_degree"], kw["scale_modifier"])
geom = O.preprocess(inp["means3D"].double(), inp["scales"].double() if inp.get("scales") is not None else None,
                    inp["rotations"].double() if inp.get("rotations") is not None else None,
                    inp.get("cov3D_precomp"), st)
for k in ("scales", "rotations", "means3D"):
    if rg.get(k) is None:
        continue
    e = (gh[k].double().cpu() - rg[k]).norm(dim=-1)
    e32 = (g32[k].double() - rg[k]).norm(dim=-1)
    tot = rg[k].norm()
    top = torch.argsort(e, descending=True)[:6]
    print(f"{k}: |ref| {tot:.3e}  HIP err norm {e.norm():.3e}  f32-oracle err norm {e32.norm():.3e}")
    for i in top.tolist():
        A, B, C = geom.conic[i].tolist()
        det = A * C - B * B
        print(f"   g {i:5d} err {e[i]:.3e} (f32 oracle {e32[i]:.3e}) |ref| {rg[k][i].norm():.3e} radius {int(geom.radii[i])} depth {geom.depth[i]:.3f} "
              f"xy ({geom.xy[i,0]:.1f},{geom.xy[i,1]:.1f}) conic det {det:.3e} A {A:.3e} C {C:.3e} scales {[round(float(x),4) for x in inp['scales'][i]] if inp.get('scales') is not None else None} fragile {bool(geom.fragile[i])}")
