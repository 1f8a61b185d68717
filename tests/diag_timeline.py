"""Diagnostic (not a test): GPU timeline of the library's stages with views pipelined over S streams (no profiler)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import diff_gaussian_rasterization as D
from gsplat_attack.scenes import make_scene
from gsplat_attack.renderer import PipelineParams, render
from gsplat_attack.streams import StreamRing

S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
model, cams, spec = make_scene("nyc-1M", device=dev, n_views=8)
pipe = PipelineParams(skip_objects=True)
bg = torch.zeros(3, device=dev)
cam = cams[0]
gc = torch.randn(3, cam.image_height, cam.image_width, device=dev)
lib = D._load()
lib.gsr_profile_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.gsr_profile_timeline.restype = ctypes.c_int
ring = StreamRing(S, dev)

def run(n):
    for i in range(n):
        with ring.next():
            model.zero_grad()
            out = render(cam, model, pipe, bg)
            out["render"].backward(gc)
    ring.join()
    torch.cuda.synchronize()

run(9)
D.profile(True)
run(12)
buf = (ctypes.c_float * (3 * 4096))()
n = lib.gsr_profile_timeline(buf, 4096)
D.profile(False)
spans = sorted((buf[3 * i + 1], buf[3 * i + 2], D.GSR_STAGES[int(buf[3 * i])]) for i in range(n))
t_end = max(s[1] for s in spans)
print(f"{n} spans, {t_end:.3f} ms for 12 views = {t_end / 12:.3f} ms/view")
for a, b, name in spans:
    if 3.0 <= a <= 6.5:
        print(f"{a:8.3f} {b:8.3f} {b - a:7.3f} {name}")
