"""Diagnostic: a batch of B views through one launch chain (render_batch) against the same views one after another
(render) on one stream -- views/s and the library's per-stage times.  python tests/diag_batch.py [scene] [B] [reps]"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splat-attack_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import diff_gaussian_rasterization as D  # noqa: E402
from gsplat_attack.renderer import PipelineParams, render, render_batch  # noqa: E402
from gsplat_attack.scenes import make_scene  # noqa: E402


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "nyc-1M"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    only = sys.argv[4] if len(sys.argv) > 4 else ""
    if len(sys.argv) > 5:
        D.set_flags(int(sys.argv[5], 0))               # launch overrides (diff_gaussian_rasterization.flag_*), e.g. 0x30 = FWD_SPLIT(4)
    dev = torch.device("cuda:0")
    model, cams, spec = make_scene(scene, device=dev, n_views=max(B, 8))
    cams = cams[:B]
    H, W = cams[0].image_height, cams[0].image_width
    P = int(model.get_xyz.shape[0])
    bg = torch.zeros(3, device=dev)
    gc = torch.randn(3, H, W, generator=torch.Generator().manual_seed(99)).to(dev)
    gcb = gc.unsqueeze(0).expand(B, 3, H, W).contiguous()
    bucket = D.GradBucket(P, dev)
    pipe = PipelineParams(skip_objects=True, grad_bucket=bucket)

    def seq():
        bucket.reset()
        for c in cams:
            render(c, model, pipe, bg)["render"].backward(gc)

    def bat():
        bucket.reset()
        render_batch(cams, model, pipe, bg)["render"].backward(gcb)

    # two batches in flight on two streams (each with its own bucket): the front end of one beside the compositors of the other
    streams2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
    buckets2 = [D.GradBucket(P, dev) for _ in range(2)]
    pipes2 = [PipelineParams(skip_objects=True, grad_bucket=b) for b in buckets2]
    flip = [0]

    def bat2():
        i = flip[0] = flip[0] ^ 1
        with torch.cuda.stream(streams2[i]):
            buckets2[i].reset()
            render_batch(cams, model, pipes2[i], bg)["render"].backward(gcb)

    for s_ in streams2:
        s_.wait_stream(torch.cuda.current_stream(dev))
    for name, fn in (("sequential", seq), ("batched", bat), ("batched-2streams", bat2)):
        if only and name != only:
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / reps)
        t = sorted(ts)[1]
        D.profile(True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        st = {k: round(ms / 3 / B, 4) for k, (ms, _) in D.profile_read().items()}
        D.profile(False)
        print(f"{name:10s} {spec.name} B={B}: {t * 1e3:8.3f} ms per batch, {t / B * 1e3:7.4f} ms per view, "
              f"{B / t:8.1f} views/s; stages ms/view {st} sum {sum(st.values()):.4f}", flush=True)


if __name__ == "__main__":
    main()
