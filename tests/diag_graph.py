"""Diagnostic (not a test): one view's forward + backward recorded into a hipGraph (asynchronous pair count) and replayed,
against the same work launched eagerly on one stream.  VERDICT r02 item 1d."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(__file__))
import conftest  # noqa
import diff_gaussian_rasterization as D
from gsplat_attack.scenes import make_scene
from gsplat_attack.renderer import PipelineParams, render

D._load()
dev = torch.device("cuda:0")
model, cams, _ = make_scene("nyc-1M", device=dev, n_views=8)
cam = cams[0]
bg = torch.zeros(3, device=dev)
gc = torch.randn(3, cam.image_height, cam.image_width, device=dev)
pipe = PipelineParams(skip_objects=True)


def one_view():
    model.zero_grad()
    render(cam, model, pipe, bg)["render"].backward(gc)


def rate(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


print(f"eager, synchronous count : {rate(one_view):8.1f} views/s")
D.set_flags(D.FLAG_ASYNC_COUNT)
print(f"eager, asynchronous count: {rate(one_view):8.1f} views/s")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5):
        one_view()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        one_view()
    ref = model._features_rest.grad.detach().clone()
    print(f"graph replay             : {rate(g.replay):8.1f} views/s")
    one = model._features_rest.grad.detach().clone()
    print("replayed gradient equals the captured run's:", torch.equal(ref, one), "finite:", bool(torch.isfinite(one).all()))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:500])
