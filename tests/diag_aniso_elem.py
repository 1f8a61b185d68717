"""Diagnostic (not a test): for an anisotropic fuzz seed, the gradient elements of a group that are off by more than
5e-3 relative -- the implementation's, the float32 oracle's and the float64 oracle's values, and the Gaussian they
belong to (scales, conic, radius, fragile flags).
    python tests/diag_aniso_elem.py SEED [GROUP]
"""
import os, sys
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import torch
import test_gpu_parity as T
from oracle import oracle_r as O
from util import settings_for
from fuzz_cases import aniso_case

seed = int(sys.argv[1])
group = sys.argv[2] if len(sys.argv) > 2 else "means2D"
inp, cam, bg, kw, desc = aniso_case(seed)
H, W = cam.image_height, cam.image_width
g = torch.Generator().manual_seed(kw["seed"])
gc = torch.randn(3, H, W, generator=g)
st = settings_for(cam, bg, kw["sh_degree"], kw["scale_modifier"])
keys = T.hip_depth_keys(inp, cam, bg, kw["sh_degree"], kw["scale_modifier"])
ref, rg = O.forward_backward(inp, st, gc, None, dtype=torch.float64, drop_fragile=True, depth_key=keys)
gc, _ = O.solid_grads(ref, gc, None)
r32, rg32 = O.forward_backward(inp, st, gc, None, dtype=torch.float32, drop_fragile=False, depth_key=keys)
import diff_gaussian_rasterization as D
color, radii, _, grads = T.run_hip(inp, cam, bg, gc, None, kw["sh_degree"], kw["scale_modifier"])
variants = {}
for name, fl in (("no segments", D.FLAG_NO_SEGMENTS), ("no cull", D.FLAG_NO_CULL), ("bwd split 2", D.flag_bwd_split(2)),
                 ("fwd split 4", D.flag_fwd_split(4)), ("fwd split 1, bwd split 2", D.flag_fwd_split(1) | D.flag_bwd_split(2))):
    with D.extra_flags(fl):
        variants[name] = T.run_hip(inp, cam, bg, gc, None, kw["sh_degree"], kw["scale_modifier"])[3]
print(f"seed {seed} {desc} sh_degree {kw['sh_degree']} scale_modifier {kw['scale_modifier']:.3f} fragile px {float(ref.fragile_px.float().mean()):.3f}")
a, b, c = grads[group].detach().double().cpu(), rg[group].double(), rg32[group].double()
scale = b.abs().max().item()
big = b.abs() > 1e-3 * scale
off = big & (((a - b).abs() / b.abs()) > 5e-3)
geo = O.preprocess(inp["means3D"].double(), inp["scales"].double(), inp["rotations"].double(), None, st)
print(f"group {group}: scale {scale:.3e}, significant {int(big.sum())}, off {int(off.sum())}")
for idx in torch.nonzero(off).tolist():
    i = idx[0]
    j = tuple(idx)
    sc = (inp["scales"][i] * kw["scale_modifier"]).tolist()
    rect = (geo.rect_min[i].tolist(), geo.rect_max[i].tolist())
    con = geo.conic[i].tolist()
    det = con[0] * con[2] - con[1] ** 2
    print(f"  element {j}: hip {a[j].item():+.6e}  f64 {b[j].item():+.6e}  f32 {c[j].item():+.6e}  | rel err hip {abs(a[j]-b[j]).item()/abs(b[j]).item():.2e} "
          f"f32 {abs(c[j]-b[j]).item()/abs(b[j]).item():.2e}")
    print(f"    Gaussian {i}: scales {sc[0]:.4f} {sc[1]:.4f} {sc[2]:.4f}  opacity {inp['opacities'][i].item():.3f} radius {int(geo.radii[i])} rect {rect} "
          f"conic ({con[0]:.3e}, {con[1]:.3e}, {con[2]:.3e}) det {det:.3e}  aniso {max(con[0], con[2]) / max(det / max(con[0], con[2]), 1e-30):.1e} fragile {bool(geo.fragile[i])}")
    for name, gv in variants.items():
        print(f"    [{name}] hip {gv[group].detach().double().cpu()[j].item():+.6e}")
    full = {k: (grads[k][i].detach().cpu().flatten()[:4].tolist(), rg[k][i].flatten()[:4].tolist()) for k in ("means3D", "means2D", "opacities")}
    print("    its other gradients (hip, f64):", full)

# DIAG_GAUSSIAN=i: every element of Gaussian i's gradients without and with GSR_FLAG_NEEDLE_DOUBLE, next to both oracles
if os.environ.get("DIAG_GAUSSIAN"):
    i = int(os.environ["DIAG_GAUSSIAN"])
    with D.extra_flags(D.FLAG_NEEDLE_DOUBLE):
        g_on = T.run_hip(inp, cam, bg, gc, None, kw["sh_degree"], kw["scale_modifier"])[3]
    for k in ("means3D", "scales", "rotations", "opacities", "means2D"):
        f64, f32 = rg[k][i].double().flatten(), rg32[k][i].double().flatten()
        off_, on_ = grads[k][i].detach().double().cpu().flatten(), g_on[k][i].detach().double().cpu().flatten()
        for j in range(f64.numel()):
            den = max(abs(f64[j].item()), 1e-30)
            print(f"  G{i} {k}[{j}]: f64 {f64[j].item():+.6e} | rel err: f32 oracle {abs(f32[j] - f64[j]).item() / den:.2e}, "
                  f"hip {abs(off_[j] - f64[j]).item() / den:.2e}, hip with the flag {abs(on_[j] - f64[j]).item() / den:.2e}")
