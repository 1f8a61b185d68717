"""Diagnostic (not a test): distribution of element-wise gradient errors, HIP vs oracle f64, and oracle f32 vs f64."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
from oracle import oracle_r as O
from util import settings_for, model_inputs
from gsplat_attack.scenes import make_scene
from test_gpu_parity import run_hip

model, cams, _ = make_scene("nyc-1M", P=20000, width=320, height=180, n_views=2)
cam = cams[0]; bg = torch.zeros(3)
inp = model_inputs(model, with_objs=False)
gc = torch.randn(3, 180, 320, generator=torch.Generator().manual_seed(99))
st = settings_for(cam, bg)
ref, rg = O.forward_backward(inp, st, gc, dtype=torch.float64)
r32, rg32 = O.forward_backward(inp, st, gc, dtype=torch.float32)
color, radii, objects, grads = run_hip(inp, cam, bg, gc)
print("fragile px", int(ref.fragile_px.sum()), "N", ref.num_rendered)
err = (color.double() - ref.color).abs().max(0).values
print("rgb err solid", err[~ref.fragile_px].max().item(), "fragile", err[ref.fragile_px].max().item() if ref.fragile_px.any() else 0)
e32 = (r32.color.double() - ref.color).abs().max(0).values
print("oracle32 rgb err solid", e32[~ref.fragile_px].max().item(), "n>1e-4:", int((e32 > 1e-4).sum()), "hip n>1e-4:", int((err > 1e-4).sum()))
for k in rg:
    if rg[k] is None or grads.get(k) is None: continue
    a = grads[k].detach().double().cpu().reshape(-1); b = rg[k].reshape(-1); c = rg32[k].double().reshape(-1)
    sc = b.abs().max()
    if sc == 0: continue
    big = b.abs() > 1e-3 * sc
    ea = ((a - b).abs() / b.abs())[big]; ec = ((c - b).abs() / b.abs())[big]
    print(f"{k:10s} sig {int(big.sum()):7d} hip: max {ea.max():.2e} n>5e-3 {int((ea>5e-3).sum())} n>1e-3 {int((ea>1e-3).sum())} | oracle32: max {ec.max():.2e} n>5e-3 {int((ec>5e-3).sum())} n>1e-3 {int((ec>1e-3).sum())}")
