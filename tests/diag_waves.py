"""Diagnostic (not a test): per-wave start/end clocks of the backward composite (K7; `fwd` argument: of K6) on the
benchmark scene."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import diff_gaussian_rasterization as D
from gsplat_attack.scenes import make_scene
from gsplat_attack.renderer import PipelineParams, render

dev = torch.device("cuda:0")
SCENE = os.environ.get("DIAG_SCENE", "nyc-1M")
model, cams, spec = make_scene(SCENE, device=dev, n_views=8)
pipe = PipelineParams(skip_objects=True)
bg = torch.zeros(3, device=dev)
cam = cams[0]
lib = D._load()
lib.gsr_debug_wave_clock.argtypes = [ctypes.c_void_p]
gc = torch.randn(3, cam.image_height, cam.image_width, device=dev)
for _ in range(3):
    model.zero_grad(); out = render(cam, model, pipe, bg); out["render"].backward(gc)
torch.cuda.synchronize()
model.zero_grad()
out = render(cam, model, pipe, bg)
img = out["render"]
rg = D.export_state(img, "ranges").view(-1, 2).long().cpu()
ln = (rg[:, 1] - rg[:, 0])
nt = ln.numel()
FWD = len(sys.argv) > 1 and sys.argv[1] == "fwd"
if FWD:
    lib.gsr_debug_wave_clock_fwd.argtypes = [ctypes.c_void_p]
    NW = 4 if nt < 4096 else 2                     # forward waves per tile (the library's choice from the tile count)
    clk = torch.zeros(NW * nt, 2, dtype=torch.int64, device=dev)
    lib.gsr_debug_wave_clock_fwd(ctypes.c_void_p(clk.data_ptr()))
    out2 = render(cam, model, pipe, bg)
    torch.cuda.synchronize()
    lib.gsr_debug_wave_clock_fwd(None)
    ln = ln.repeat_interleave(NW)
    nt = NW * nt
else:
    clk = torch.zeros(nt, 2, dtype=torch.int64, device=dev)
    lib.gsr_debug_wave_clock(ctypes.c_void_p(clk.data_ptr()))
    img.backward(gc)
    torch.cuda.synchronize()
    lib.gsr_debug_wave_clock(None)
SLOTS = 8192 if FWD else 5120
c = clk.cpu().double() / 100.0          # microseconds
t0 = c[:, 0].min()
start, end = c[:, 0] - t0, c[:, 1] - t0
dur = end - start
print(f"tiles {nt}  kernel span {end.max():.1f} us   wave duration: mean {dur.mean():.1f} p50 {dur.median():.1f} p99 {dur.quantile(0.99):.1f} max {dur.max():.1f} us")
print(f"sum of wave durations / ({SLOTS} slots x span) = {dur.sum() / (SLOTS * end.max()):.3f}")
order = torch.argsort(ln, descending=True)
print("longest lists: len, start, end, dur, us/entry")
for t in order[:8].tolist():
    print(f"   {ln[t].item():5d} {start[t]:8.1f} {end[t]:8.1f} {dur[t]:8.1f} {dur[t] / max(ln[t].item(), 1):.3f}")
late = torch.argsort(end, descending=True)
print("last waves to finish: len, start, end, dur")
for t in late[:8].tolist():
    print(f"   {ln[t].item():5d} {start[t]:8.1f} {end[t]:8.1f} {dur[t]:8.1f}")
for q in (0.25, 0.5, 0.75, 0.9, 1.0):
    tq = end.max() * q
    print(f"   waves running at {q:.2f} of span: {((start <= tq) & (end > tq)).sum().item()}")
print("us per entry by start-time quartile:", [f"{(dur[m] / ln[m].clamp(min=1)).mean():.3f}" for m in
      [(start >= end.max() * a) & (start < end.max() * b) for a, b in ((0, .25), (.25, .5), (.5, .75), (.75, 1.01))] if m.any()])
