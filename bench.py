#!/usr/bin/env python
"""bench.py -- fwd+bwd views/sec @1080p, 1M Gaussians (BASELINE.json metric) on N MI355X GPUs of one node.

One "step" = one view: ``render()`` forward of the S-nyc-1M synthetic scene (1,000,000 Gaussians, SH degree 3,
1920x1080) through the drop-in ``diff_gaussian_rasterization`` package (libgsraster.so,
hand-written HIP) + one backward from a fixed dL/dC[3,H,W] down to .grad on all seven raw attribute tensors
(59 attack-relevant floats per Gaussian), object channels off.  Inputs are resident in HBM before a timed region
starts.  A timed region is EXACTLY --steps steps between barrier + synchronise on both sides; --regions of them are
run back to back and the MEDIAN region is reported (every region's time is listed).

N = 1 (the BASELINE metric): `value` = independent views -- each with its own forward, its own dL/dC and its own 59
attribute-gradient floats per Gaussian -- rendered several at a time through ONE launch chain on one HIP stream
(gsr_forward_raw_batch + gsr_backward_raw_batch_views: the views of a group share the storage scan, both sorts, the
emission, the schedule and the two compositor launches; view v's gradients are bit for bit the single-view backward's;
views per group = the largest divisor of --steps up to 8, so a region is exactly --steps views in whole groups);
`pipelined_streams` = the same views as one render() + backward per view dealt over --streams HIP streams (the headline
regime of rounds 2-5), `sequential` = strictly one view after another on one stream (what the reference's default loop,
batch_mode false, and config 4's one view per rank see), `batched` = the eight ring cameras as one batch whose gradients are
SUMMED into one bucket (what reference attack.py:476-494 accumulates in .grad), `pgd` = whole PGD iterations (config 3 and
config 4 on one GPU) split into their phases.
N > 1 (config 4): a step is one PGD iteration's worth of rasterisation per rank -- --views-per-rank forward+backward
passes into the rank's gradient bucket, ONE sum all-reduce of the bucket (59 floats per Gaussian) over RCCL, the fused
projected-gradient step on all six attribute tensors -- strictly in that order, as a PGD loop needs it; `allreduce_ms`
is the collective alone.  --independent-views restores round 2's behaviour (views pipelined across steps, no update).

    python bench.py                       # N=1, defaults finish in ~2 minutes incl. the CPU baseline
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel stage against the 8 TB/s HBM peak with the
algorithmic bytes of SURVEY.md section 8(d) and its live HIP-event duration over the timed region;
`cpu_baseline` times oracle-R (pure PyTorch, CPU) on a bounded sample of the same scene.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splat-attack_amd"))
sys.path.insert(0, ROOT)
# the HIP runtime multiplexes a process's streams onto 4 hardware queues unless told otherwise (read at its first
# call): views pipelined over four streams (+ torch's own) need more, or kernels of independent views serialise
# behind each other (same box: 4 queues / 3 streams 1276, 4 / 4 1200, 8 / 4 1310 views/s)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from benchparts.common import HBM_PEAK_GBS, log, sources_digest, stage_bytes  # noqa: E402
from benchparts.cpu import cpu_baseline, parity_and_cfg1  # noqa: E402
from benchparts.extras import extras_and_pgd, masked_streams  # noqa: E402
from benchparts.fanout import fan_out  # noqa: E402

PMC_FILE = "r06_pmc_traffic.json"   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (profiles/collect_r06.sh)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--regions", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--scene", default="nyc-1M")
    ap.add_argument("--P", type=int, default=None, help="override the Gaussian count (parity-size runs)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--objects", action="store_true", help="composite the 16 object channels too")
    ap.add_argument("--color-only", action="store_true",
                    help="gradients on the SH coefficients only (BASELINE configs 2/3): geometry parameters frozen")
    ap.add_argument("--classic", action="store_true",
                    help="activated tensors through GaussianRasterizer (what the reference's own render() does) instead of "
                         "the fused raw-parameter path")
    ap.add_argument("--no-cull", action="store_true", help="keep the full 3-sigma tile rects (A/B of the footprint cull)")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams independent views are dealt over (view i runs on stream i %% S)")
    ap.add_argument("--views-per-rank", type=int, default=1, help="N > 1: views each rank renders per PGD step")
    ap.add_argument("--ar-chunks", type=int, default=1,
                    help="N > 1, one bucket per rank: all-reduce the bucket in this many ranges of Gaussians, each issued "
                         "while K9 still computes the next (1 = one collective after the backward)")
    ap.add_argument("--independent-views", action="store_true",
                    help="N > 1: round-2 behaviour (views pipelined across steps, all-reduce but no parameter update)")
    ap.add_argument("--batch", type=int, default=8,
                    help="N = 1: views per batch of the `batched` block (one launch chain per batch, gsr_forward_raw_batch); "
                         "0 = skip.  N > 1: the rank's --views-per-rank views go through one launch chain unless --no-batch")
    ap.add_argument("--no-batch", action="store_true", help="N > 1: one render() + backward per view, as in round 5")
    ap.add_argument("--headline", choices=("auto", "batched-views", "streams"), default="auto",
                    help="N = 1: how the independent views of the headline `value` are run -- batched-views: several views "
                         "at a time through ONE launch chain, every view's 59 gradient floats per Gaussian written to its "
                         "own buffer (gsr_backward_raw_batch_views); streams: one render() + backward per view dealt over "
                         "--streams HIP streams (rounds 2-5's headline, reported as `pipelined_streams` otherwise); auto: "
                         "batched-views when the configuration allows it")
    ap.add_argument("--views-per-chain", type=int, default=0,
                    help="headline regime: at most this many views per launch chain (default 8; the largest divisor of "
                         "--steps up to it is used)")
    ap.add_argument("--headline-only", action="store_true",
                    help="profiling runs: only the headline regime (warm-up included) launches kernels, so that per-kernel "
                         "averages of a profiler describe one kind of launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (dense point, forward-only, PGD blocks)")
    ap.add_argument("--scale-modifier", type=float, default=1.0,
                    help="render()'s scaling_modifier for the main timed region (diagnostic: 2.43 = the dense data point)")
    ap.add_argument("--one-camera", action="store_true",
                    help="render the same camera every step instead of cycling the 8 ring cameras")
    ap.add_argument("--dense-pairs", type=float, default=10e6,
                    help="target pair count of the second, denser data point (scale_modifier is searched for it); 0 = skip")
    ap.add_argument("--cpu-sample", default="100000,1920,1080")
    ap.add_argument("--flags", type=lambda v: int(v, 0), default=0,
                    help="extra GsrSettings.flags for every call (launch overrides: diff_gaussian_rasterization.flag_*)")
    ap.add_argument("--cu-masks", default=None,
                    help="experiment: one CU mask per view stream, ';'-separated, each a ','-separated list of 32-bit hex "
                         "words (hipExtStreamCreateWithCUMask); replaces --streams")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(fan_out(args.gpus))              # before anything touches a GPU in this (parent) process
    if os.environ.get("BENCH_TEST_HANG_RANK") == os.environ.get("RANK", "-"):
        # test hook (tests/test_dist_cpu.py): this rank behaves like one stuck in a collective -- it ignores SIGTERM
        import signal
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        time.sleep(3600)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the raster path has no CPU fallback")
    from gsplat_attack import dist as gdist
    from gsplat_attack import pgd as gpgd
    from gsplat_attack.renderer import PipelineParams, can_batch, render, render_batch
    from gsplat_attack.scenes import make_scene
    import diff_gaussian_rasterization as D

    # BENCH_REHEARSE_GLOO=1: rehearsal of the N > 1 code path on a box with ONE GPU (every rank on cuda:0, gloo
    # collectives through the host) -- checks the distributed logic, measures nothing meaningful
    rehearse = os.environ.get("BENCH_REHEARSE_GLOO", "0") == "1"
    rank, world, local = gdist.init_from_env("gloo" if rehearse else "nccl")
    if rehearse:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: start {args.gpus} ranks (torchrun "
                         f"--nproc-per-node {args.gpus}), or run `python bench.py --gpus {args.gpus}` as a plain process "
                         "and it starts them itself")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"[bench] rank {rank}: LOCAL_RANK {local} but this node shows {torch.cuda.device_count()} HIP "
                         f"device(s) (--gpus {args.gpus})")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    D._load()
    if args.no_cull or args.flags:
        D.set_flags((D.FLAG_NO_CULL if args.no_cull else 0) | args.flags)

    n_views = max(8, world * max(args.views_per_rank, 1))
    if rank == 0:
        log(f"building scene {args.scene} on {dev}")
    model, cams, spec = make_scene(args.scene, device=dev, P=args.P, width=args.width, height=args.height,
                                   n_views=n_views)
    cam = cams[rank % n_views]
    H, W = cam.image_height, cam.image_width
    P = model.get_xyz.shape[0]
    pipe = PipelineParams(skip_objects=not args.objects, viewspace_grad=not args.color_only,
                          fused_activations=not args.classic)
    if args.color_only:
        for n_ in ("_xyz", "_scaling", "_rotation", "_opacity", "_objects_dc"):
            getattr(model, n_).requires_grad_(False)
    bg = torch.zeros(3, device=dev)
    gc = torch.randn(3, H, W, generator=torch.Generator().manual_seed(99)).to(dev)
    scale_mod = [args.scale_modifier]
    step_no = [0]
    pgd_loop = world > 1 and not args.independent_views
    B = max(args.views_per_rank, 1) if pgd_loop else 1

    def next_cam():
        # every step renders the NEXT camera of the ring (rank r: views (i * world + r) mod 8): no step re-renders what
        # the previous one left in the caches
        if args.one_camera:
            return cam
        c = cams[(step_no[0] * world + rank) % n_views]
        step_no[0] += 1
        return c

    def step():
        model.zero_grad()
        out = render(next_cam(), model, pipe, bg, scale_mod[0])
        out["render"].backward(gc)
        if world > 1:
            gdist.allreduce_attribute_grads(model)
        return out

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    if args.cu_masks:
        # experiment: the view streams as hipExtStreamCreateWithCUMask streams, one mask per stream
        streams = masked_streams(dev, args.cu_masks)
        args.streams = len(streams)

    def run_steps(n, use_streams=True):
        sts = streams if use_streams else None
        if sts is None:
            for _ in range(n):
                step()
            return
        for s_ in sts:
            s_.wait_stream(torch.cuda.current_stream(dev))
        for i in range(n):
            with torch.cuda.stream(sts[i % len(sts)]):
                step()
        for s_ in sts:
            torch.cuda.current_stream(dev).wait_stream(s_)

    # ---- N = 1 headline: independent views, several at a time through ONE launch chain --------------------------------
    # Every view keeps its own gradient: view v's 59 floats per Gaussian land in bucket v of a GradBucketSet, bit for bit
    # what the single-view backward writes (tests/test_gpu_batch.py).  The views of a step group share the storage scan,
    # both sorts, the emission, the tile schedule and the two compositor launches; the per-Gaussian backward runs once per
    # view.  Views per group: the largest divisor of --steps up to 8, so that a region is EXACTLY --steps views in whole
    # groups and every launch of a kernel covers the same number of views.
    Bh = 0
    if world == 1 and args.headline != "streams" and not (args.classic or args.objects or args.color_only or args.cu_masks):
        cap = args.views_per_chain if args.views_per_chain else 8
        Bh = max([d for d in range(2, min(cap, D.MAX_BATCH) + 1) if args.steps % d == 0], default=0)
    if args.headline == "batched-views" and Bh == 0:
        raise SystemExit("[bench] --headline batched-views needs N = 1, the fused path without object channels and a "
                         "--steps with a divisor in 2..8")
    if Bh:
        bset_h = D.GradBucketSet(Bh, P, dev)
        pipe_h = PipelineParams(skip_objects=True, grad_bucket=bset_h)
        gcb_h = gc.unsqueeze(0).expand(Bh, 3, H, W).contiguous()

        def run_headline(n):
            assert n % Bh == 0
            for _ in range(n // Bh):
                out_h = render_batch([next_cam() for _ in range(Bh)], model, pipe_h, bg, scale_mod[0])
                out_h["render"].backward(gcb_h)
            return out_h

    # ---- N > 1: one PGD iteration per step (config 4) ---------------------------------------------------------------
    ar_events = []
    bytes_reduced = [0]
    if pgd_loop:
        from gsplat_attack.streams import StreamRing
        ring = StreamRing(min(args.streams, B), dev)
        buckets = [D.GradBucket(P, dev) for _ in range(ring.n)]
        pipe_b = PipelineParams(skip_objects=not args.objects, grad_bucket=lambda: buckets[ring.current])
        originals = {n_: getattr(model, n_).detach().clone() for n_ in gdist.ATTACK_PARAMS}
        rank_cams = [cams[(rank * B + v) % n_views] for v in range(B)]
        batch_views = (not args.no_batch) and B >= 2 and can_batch(rank_cams, model, pipe_b)
        gcb_rank = gc.unsqueeze(0).expand(B, 3, H, W).contiguous() if batch_views else None
        pipe_one = PipelineParams(skip_objects=not args.objects, grad_bucket=buckets[0])

        def step_all():
            # the identical projected L2 step on every rank (attack.py:53-173): all six tensors in one launch (gsr_pgd_step_multi,
            # the norms summed in front), bit for bit the six per-tensor steps it falls back to
            items = [(getattr(model, n_), getattr(model, n_).grad, originals[n_], None) for n_ in gdist.ATTACK_PARAMS]
            if not gpgd.multi_step_(items, 0.5, 5.0, True):
                for p_, g_, o_, _ in items:
                    gpgd.l2_step_(p_, g_, 0.5, 5.0, o_)

        def pgd_step_batched():
            # the rank's B views through ONE launch chain (gsr_forward_raw_batch / gsr_backward_raw_batch_into): the bucket
            # receives the summed gradient of the B views, written once
            buckets[0].reset()
            ar = None
            out_ = render_batch(rank_cams, model, pipe_one, bg, scale_mod[0])["render"]
            if args.ar_chunks > 1:
                ar = gdist.BucketAllReduce(buckets[0], args.ar_chunks)
            out_.backward(gcb_rank)
            return ar

        def pgd_step():
            if batch_views:
                ar = pgd_step_batched()
                tot = buckets[0]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if ar is None:
                    ar = gdist.BucketAllReduce(tot, 1)
                bytes_reduced[0] = ar.wait()
                e1.record()
                ar_events.append((e0, e1))
                tot.assign_to(model)
                step_all()
                return
            for b_ in buckets:
                b_.reset()
            ar = None
            for vi, c_ in enumerate(rank_cams):
                with ring.next():
                    out_ = render(c_, model, pipe_b, bg, scale_mod[0])["render"]
                    if vi == len(rank_cams) - 1 and len(buckets) == 1 and args.ar_chunks > 1:
                        # the step's last backward fills the bucket in ranges; each range is all-reduced as soon as its
                        # K9 launch is enqueued, while the following ranges are still being computed
                        ar = gdist.BucketAllReduce(buckets[0], args.ar_chunks)
                    out_.backward(gc)
            ring.join()
            tot = buckets[0]
            for b_ in buckets[1:]:
                tot.add_(b_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                                    # every kernel of the step's rasterisation is enqueued before this
            if ar is None:
                ar = gdist.BucketAllReduce(tot, 1)
            bytes_reduced[0] = ar.wait()                   # ONE collective of 59 floats per Gaussian (or its ranges)
            e1.record()
            ar_events.append((e0, e1))
            tot.assign_to(model)
            step_all()

        def run_steps(n, use_streams=True):                # noqa: F811 -- the N > 1 definition of a step
            for _ in range(n):
                pgd_step()

    if rank == 0:
        log(f"scene ready: P={P}, {W}x{H}; warmup x{args.warmup}")
    for i in range(0 if (Bh and args.headline_only) else args.warmup):
        if pgd_loop:
            pgd_step()
        else:
            out = step()
        torch.cuda.synchronize()
        if rank == 0 and not pgd_loop:
            log(f"warmup step {i} done, N={D.last_num_rendered(out['render'])}")
    if Bh:
        # the headline regime's own warm-up: at least --warmup views, in whole groups
        oh = run_headline(max(-(-args.warmup // Bh), 2) * Bh)
        torch.cuda.synchronize()
        if rank == 0:
            log(f"headline warm-up done ({Bh} views per launch chain), pairs per group N={D.last_num_rendered(oh['render'])}")
        del oh
    out = None
    ar_events.clear()
    info = {}
    # pair / visible counts of every camera of the ring (the roofline figures use their means)
    Ns, Vs, Es = [], [], []
    if Bh and args.headline_only:
        # (profiling runs: counted from forward-only GROUPS, so that every K6 launch of the process is a group launch)
        with torch.no_grad():
            pass
        for g0 in range(0, n_views, Bh):
            grp = [cams[(g0 + k) % n_views] for k in range(Bh)]
            o_ = render_batch(grp, model, PipelineParams(skip_objects=True), bg, scale_mod[0])
            Ns.extend([D.last_num_rendered(o_["render"]) // Bh] * Bh)
            Vs.extend([int(D.export_state(o_["render"], "dv")[1].item()) // Bh] * Bh)
            Es.extend([int(D.export_state(o_["render"], "n_contrib").to(torch.int64).sum().item()) // Bh] * Bh)
            del o_
    for c_ in ([] if (Bh and args.headline_only) else (cams[:n_views] if not args.one_camera else [cam])):
        o_ = render(c_, model, pipe, bg, scale_mod[0])
        Ns.append(D.last_num_rendered(o_["render"]))
        Vs.append(int(D.export_state(o_["render"], "dv")[1].item()))
        # (pixel, entry) evaluations of the reference's per-pixel walk: every pixel visits its tile's list up to its last
        # contributor ("Flops (secondary)": E, measured here from the forward's n_contrib)
        Es.append(int(D.export_state(o_["render"], "n_contrib").to(torch.int64).sum().item()))
        del o_
    info["N_per_camera"], info["V_per_camera"] = Ns, Vs
    info["N"], info["V"] = int(round(sum(Ns) / len(Ns))), int(round(sum(Vs) / len(Vs)))
    info["E"] = int(round(sum(Es) / len(Es)))
    step_no[0] = 0

    def timed_regions(n_regions, use_streams=True, profile_stage=None, runner=None):
        """n_regions x exactly --steps steps, each between barrier + synchronise; per-region seconds (max over ranks)."""
        secs = []
        for _ in range(n_regions):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if runner is not None:
                runner(args.steps)
            else:
                run_steps(args.steps, use_streams)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                if rehearse:
                    t = t.cpu()
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            secs.append(dt)
        return secs

    # Timed regions: only the dominant kernel (K7, the backward composite) is timed inside them, by two HIP events that
    # hipExtLaunchKernelGGL stamps with the dispatch's own start and end on its launch stream.
    DOMINANT = "render_bwd"
    pipelined = None
    if Bh:
        D.profile(True, stages=[DOMINANT])
        secs = timed_regions(args.regions, runner=run_headline)
        dom_calls = D.profile_read()[DOMINANT]
        D.profile(False)
        if not args.headline_only and streams is not None:
            # rounds 2-5's headline regime beside it: one render() + backward per view, dealt over the streams
            run_steps(2 * max(args.streams, 1))
            torch.cuda.synchronize()
            secs_p = timed_regions(args.regions)
            mp_ = sorted(secs_p)[len(secs_p) // 2]
            pipelined = {"value": round(args.steps / mp_, 2), "unit": "views/s", "ms_per_step": round(mp_ / args.steps * 1e3, 4),
                         "regions_ms": [round(x * 1e3, 2) for x in secs_p], "streams": args.streams,
                         "what": "the same independent views as one render() + backward per view dealt round-robin over "
                                 "--streams HIP streams: the headline regime of rounds 2-5"}
    else:
        run_steps(2 * max(args.streams, 1))                # untimed: lets every stream build its own workspace blocks
        torch.cuda.synchronize()
        D.profile(True, stages=[DOMINANT])
        secs = timed_regions(args.regions)
        dom_calls = D.profile_read()[DOMINANT]
    dom_ms_timed = dom_calls[0] / max(dom_calls[1], 1)
    # views one launch of the dominant kernel covers in the timed regions
    views_per_launch = Bh if Bh else (B if (pgd_loop and batch_views) else 1)
    med = sorted(secs)[len(secs) // 2]
    if rank == 0:
        log(f"timed regions ({args.steps} steps each): " + ", ".join(f"{x * 1e3:.1f} ms" for x in secs))
    ar_ms = None
    if ar_events:
        torch.cuda.synchronize()
        ar_ms = sum(a.elapsed_time(b) for a, b in ar_events) / len(ar_events)
    # N > 1, PGD loop: the N = 1-comparable figure in the same line -- every rank renders --steps INDEPENDENT views
    # (forward + backward, no collective, no update) over its streams, exactly what the N = 1 line's `value` times on one
    # GPU; one region between barrier + synchronise, max over ranks
    indep = None
    if pgd_loop:
        def indep_steps(n):
            for s_ in (streams or []):
                s_.wait_stream(torch.cuda.current_stream(dev))
            for i in range(n):
                ctx_ = torch.cuda.stream(streams[i % len(streams)]) if streams is not None else contextlib.nullcontext()
                with ctx_:
                    model.zero_grad()
                    render(next_cam(), model, pipe, bg, scale_mod[0])["render"].backward(gc)
            for s_ in (streams or []):
                torch.cuda.current_stream(dev).wait_stream(s_)
        indep_steps(2 * max(args.streams, 1))
        keep_run, run_steps = run_steps, (lambda n, use_streams=True: indep_steps(n))
        si = timed_regions(1)[0]
        run_steps = keep_run
        indep = {"value": round(world * args.steps / si, 2), "unit": "views/s", "per_gpu": round(args.steps / si, 2),
                 "ms_per_step": round(si / args.steps * 1e3, 4), "streams": args.streams,
                 "what": "every rank renders --steps independent views (fwd+bwd, no all-reduce, no update) over its "
                         "streams: N x what the N = 1 line's `value` measures; one region, max over ranks"}
    # the same views strictly one after another on one stream
    seq = None
    if world == 1 and streams is not None and not args.headline_only:
        D.profile(False)
        secs1 = timed_regions(args.regions, use_streams=False)
        m1 = sorted(secs1)[len(secs1) // 2]
        seq = {"value": round(args.steps / m1, 2), "unit": "views/s", "ms_per_step": round(m1 / args.steps * 1e3, 4),
               "regions_ms": [round(x * 1e3, 2) for x in secs1], "streams": 1,
               "what": "the same fwd+bwd views strictly one after another on ONE stream: what the reference's default "
                       "loop (configs/config.yaml:56 batch_mode false, attack.py:486-494) and config 4's one view per "
                       "rank per step see"}
    # A batch of views through ONE launch chain (gsr_forward_raw_batch / gsr_backward_raw_batch_into): what the reference's
    # batch loop (attack.py:476-494: B render() calls, B backward passes summed in .grad) becomes when the B views are one
    # virtual scene -- one scan, one depth sort, one emission, one tile sort, one schedule, one forward and one backward
    # composite for all of them, every SH row read once, the 59 gradient floats per Gaussian written once.  ONE stream.
    batched = None
    if world == 1 and args.batch >= 2 and not args.classic and not args.objects and not args.color_only and not args.headline_only:
        Bb = min(args.batch, len(cams), D.MAX_BATCH)
        bcams = cams[:Bb]
        bucket_b = D.GradBucket(P, dev)
        pipe_bb = PipelineParams(skip_objects=True, grad_bucket=bucket_b)
        gcb = gc.unsqueeze(0).expand(Bb, 3, H, W).contiguous()

        def batch_step():
            bucket_b.reset()
            out_b = render_batch(bcams, model, pipe_bb, bg, scale_mod[0])
            out_b["render"].backward(gcb)
            bucket_b.assign_to(model)                      # .grad = views of the bucket (no copy)
            return out_b
        D.profile(False)
        nbt = max(args.steps // Bb, 3)
        for _ in range(3):
            ob = batch_step()
        torch.cuda.synchronize()
        n_batch = D.last_num_rendered(ob["render"])
        del ob
        tb = []
        for _ in range(args.regions):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nbt):
                batch_step()
            torch.cuda.synchronize()
            tb.append(time.perf_counter() - t0)
        mb = sorted(tb)[len(tb) // 2]
        D.profile(True)
        for _ in range(3):
            batch_step()
        torch.cuda.synchronize()
        st_b = {k: round(ms / 3 / Bb, 4) for k, (ms, _) in D.profile_read().items()}
        D.profile(False)
        batched = {"value": round(Bb * nbt / mb, 2), "unit": "views/s", "views_per_batch": Bb, "batches_per_region": nbt,
                   "ms_per_batch": round(mb / nbt * 1e3, 4), "ms_per_view": round(mb / nbt / Bb * 1e3, 4),
                   "regions_ms": [round(x * 1e3, 2) for x in tb], "streams": 1, "N_pairs_per_batch": n_batch,
                   "stages_ms_per_view": st_b,
                   "what": f"the {Bb} ring cameras as ONE batch per step on ONE stream: gsr_forward_raw_batch + "
                           "gsr_backward_raw_batch_into from a fixed dL/dC per view, the gradients of the batch's views summed "
                           "into one 59-float-per-Gaussian bucket (what reference attack.py:476-494 accumulates in .grad); "
                           "every image bit for bit the single-view render (tests/test_gpu_batch.py)"}
        model.zero_grad()
    # Untimed extra pass with every stage bracketed: the per-stage breakdown reported under "stages".
    D.profile(True)
    nb = max(3, min(args.steps, 10))
    if pgd_loop:
        for _ in range(nb):
            pgd_step()
        nb_views = nb * B
    elif Bh:
        run_headline(3 * Bh)                               # the headline regime's stages (per view: / views)
        nb_views = 3 * Bh
    else:
        for _ in range(nb):
            step()
        nb_views = nb
    torch.cuda.synchronize()
    stages = {k: (ms / max(nb_views, 1), calls) for k, (ms, calls) in D.profile_read().items()}
    D.profile(False)

    ranks_seen = 1
    if world > 1:
        # every rank adds one: what the collective path really spans (a line that says n_gpus: N while the ranks never
        # met would be the silent one-GPU measurement again)
        one = torch.ones(1, dtype=torch.int32, device="cpu" if rehearse else dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        assert ranks_seen == world, (ranks_seen, world)
    if rank == 0:
        N, V, HW = info["N"], info["V"], H * W
        sb = stage_bytes(P, V, N, HW)
        per = {}
        for name, (avg_ms, calls) in stages.items():
            per[name] = {"avg_ms": round(avg_ms, 4),
                         "GBps": round(sb[name] / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 and sb[name] else None}
        dom = max(("render_fwd", "render_bwd", "preprocess", "preprocess_bwd", "tile_sort", "bin"),
                  key=lambda n: per[n]["avg_ms"])
        # (stage times are per VIEW; a launch of the dominant kernel covers views_per_launch views in the headline regime)
        vpl = views_per_launch
        dom_ms = round(per[dom]["avg_ms"] * vpl, 4)
        if dom == DOMINANT:
            dom_ms = round(dom_ms_timed, 4)      # the duration measured inside the timed regions themselves
        achieved = sb[dom] * vpl / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes of the dominant kernel from the committed PMC passes (profiles/collect_r03.sh: FETCH_SIZE and
        # WRITE_SIZE in separate runs, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950) -- only when they
        # were taken on this very workload
        traffic = None
        valu = None
        traffic_note = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
            if pmc.get("sources_sha256") != sources_digest():
                traffic_note = (f"profiles/{PMC_FILE} was collected from other kernel sources than this build's (digest "
                                "mismatch): not quoted; run profiles/collect_r06.sh again")
                raise LookupError(traffic_note)
            kern = {"render_bwd": "void gsr::k_render_bwd<false, 4, true>", "render_fwd": "void gsr::k_render_fwd<false, 2, 1>",
                    "preprocess_bwd": "void gsr::k_pre_bwd<true, true, false>",
                    "preprocess": "void gsr::k_pre_color<true>"}.get(dom)
            if int(pmc.get("views_per_launch", 1)) != vpl:
                traffic_note = (f"profiles/{PMC_FILE} counts launches of {pmc.get('views_per_launch', 1)} view(s), this run's "
                                f"cover {vpl}: not quoted")
                raise LookupError(traffic_note)
            if (args.scene, args.P, args.width, args.height, args.objects) == ("nyc-1M", None, None, None, False):
                traffic = round(pmc["per_kernel"][kern]["hbm_bytes_fetch_x2"])
                n_valu = pmc["per_kernel"][kern].get("SQ_INSTS_VALU")
                if n_valu:
                    peak = 1024 * 2.4e9 / 2 / 1e9
                    ach = n_valu / (per[dom]["avg_ms"] * vpl * 1e-3) / 1e9
                    valu = {"bound": "fp32 VALU issue", "wave_instr_per_launch": round(n_valu),
                            "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instr/s",
                            "frac": round(ach / peak, 4), "launch_ms": round(per[dom]["avg_ms"] * vpl, 4)}
                    if dom == "render_bwd":
                        # what a kernel made of NOTHING but K7's strip-body instruction mix sustains on this chip at six
                        # waves per SIMD (tests/ubench/valu_rate.hip `kmix`: 2 transcendentals, 1 min, 4 compares, 2 selects,
                        # 23 mul / add / fma per strip evaluation; profiles/r06_valu_rate.txt)
                        valu["mix_ceiling"] = 765.9
                        valu["frac_of_mix_ceiling"] = round(ach / 765.9, 4)
                        valu["mix_source"] = "profiles/r06_valu_rate.txt (K7 strip-body mix, 6 waves/SIMD)"
        except Exception as e:                             # noqa: BLE001
            traffic = None
            valu = None
            traffic_note = traffic_note or f"profiles/{PMC_FILE}: {type(e).__name__}"
        B_total = 304 * P + 548 * V + 116 * N + 40 * HW
        views_per_step = world * B
        t_view = med / (args.steps * B)
        workload = (f"{spec.name}: {P} Gaussians (SH degree 3), {W}x{H}, "
                    + (f"one PGD iteration per step: {B} view(s) per GPU fwd+bwd into a gradient bucket, one all-reduce of "
                       "59 floats/Gaussian, fused L2 step on all six attribute tensors" if pgd_loop else
                       (f"one view per step: independent views, {Bh} at a time through ONE launch chain (gsr_forward_raw_batch "
                        "+ gsr_backward_raw_batch_views), fwd + bwd, every view's 59 attribute-gradient floats per Gaussian "
                        "written to its own buffer" if Bh else
                        "one view per step per GPU, render() fwd + bwd to all attribute grads"))
                    + (", 16 object channels on" if args.objects else ", object channels off")
                    + (", gradients on SH coefficients only" if args.color_only else "")
                    + (", classic activated-tensor surface" if args.classic else ""))
        result = {
            "metric": "fwd+bwd views/sec @1080p, 1M Gaussians",
            "value": round(views_per_step * args.steps / med, 3),
            "unit": "views/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(med / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "regions_ms": [round(x * 1e3, 2) for x in secs],
            "timing": f"median of {args.regions} regions of exactly {args.steps} steps, each between barrier + synchronise",
            "config": {"workload": workload,
                       "P": P, "V_visible": V, "N_pairs": N, "width": W, "height": H,
                       "loop": "pgd" if pgd_loop else "independent views",
                       "views_per_rank": B,
                       "views_per_launch_chain": views_per_launch,
                       "cameras": "one fixed camera" if args.one_camera else
                                  f"the {n_views} ring cameras in turn (V and N are means over them)",
                       "N_pairs_per_camera": info["N_per_camera"], "V_visible_per_camera": info["V_per_camera"],
                       "streams": args.streams,
                       "parallelism": f"views sharded {B}/GPU, dp{world}"
                                      + (f", independent views {Bh} per launch chain on one HIP stream" if Bh else
                                         (f", independent views pipelined over {args.streams} HIP streams per GPU"
                                          if args.streams > 1 and not pgd_loop else ""))
                                      + (", RCCL all-reduce of 59 floats/Gaussian per step" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": traffic_note if traffic is None else
                         f"profiles/{PMC_FILE}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on "
                         "this workload (not measured inside this run; FETCH_SIZE doubled per MI355X_MICROARCH.md)",
                         "algorithmic_bytes_per_launch": sb[dom] * vpl, "views_per_launch": vpl, "avg_launch_ms": dom_ms,
                         # the same kernel in the untimed per-stage pass (stage events around it)
                         "avg_launch_ms_alone": round(per[dom]["avg_ms"] * vpl, 4),
                         "note": "K6/K7 are bound by VALU instruction issue (256*N alpha evaluations), not by HBM; "
                                 "HBM is the reporting roofline BASELINE.md section 3 prescribes"},
            "pipeline": {"bytes_per_view": B_total, "achieved": round(B_total / t_view / 1e9, 1), "unit": "GB/s",
                         "frac": round(B_total / t_view / 1e9 / HBM_PEAK_GBS, 5)},
            "stages": per,
        }
        if valu is not None:
            result["roofline"]["valu"] = valu
        # SURVEY.md section 8d's secondary figure: the (pixel, entry) alpha evaluations E of the reference's walk (sum over
        # pixels of the last contributor's list position, mean over the cameras) against the compositors' own durations,
        # priced at the survey's 14 FMA + exp forward / 45 FMA + exp backward per evaluation (29 / 91 flop) -- what the
        # kernels deliver in the reference's own unit of work, whatever strips and lanes they evaluate to get there
        k6, k7 = per["render_fwd"]["avg_ms"], per["render_bwd"]["avg_ms"]
        if k6 > 0 and k7 > 0 and info.get("E"):
            E = info["E"]
            result["roofline"]["alpha_evaluations"] = {
                "per_view": E, "per_pixel": round(E / HW, 1),
                "fwd_Geval_per_s": round(E / (k6 * 1e-3) / 1e9, 1), "bwd_Geval_per_s": round(E / (k7 * 1e-3) / 1e9, 1),
                "fwd_TFLOPs_equiv": round(29 * E / (k6 * 1e-3) / 1e12, 1), "bwd_TFLOPs_equiv": round(91 * E / (k7 * 1e-3) / 1e12, 1),
                "fp32_vector_peak_TFLOPs": 157.3,
                "note": "render_fwd includes the tile-schedule kernel; one-stream stage durations of the untimed pass"}
        if pipelined is not None:
            result["pipelined_streams"] = pipelined
        if seq is not None:
            result["sequential"] = seq
        if batched is not None:
            result["batched"] = batched
        if world > 1:
            result["ranks_seen"] = ranks_seen
            try:
                result["rccl"] = None if rehearse else ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:                         # noqa: BLE001 -- a version string must not cost the line
                result["rccl"] = f"unavailable: {e}"
            result["collective_backend"] = dist.get_backend()
            result["allreduce_ms"] = None if ar_ms is None else round(ar_ms, 4)
            result["bytes_reduced"] = bytes_reduced[0] or 59 * 4 * P
            result["views_per_rank"] = B
            result["independent_views"] = indep
            result["allreduce_chunks"] = args.ar_chunks if (pgd_loop and min(args.streams, B) == 1) else 1
            result["allreduce_ms_is"] = ("time the compute stream waits for the collective after the step's last "
                                         "rasteriser kernel (with --ar-chunks > 1 part of it ran behind K9)")
            result["scaling_note"] = ("N > 1 lines time whole PGD iterations (serialised render -> backward -> all-reduce -> "
                                      "step); compare them with `pgd.cfg4_one_gpu` of the N = 1 line, not with its `value` "
                                      "(independent views, no update)")
        if world == 1 and not args.no_extras and args.dense_pairs > 0:
            # Second data point: the same scene with the splats scaled up until a view emits ~10 M pairs (BASELINE.md's
            # nominal N; the SURVEY section 8d distributions give 2.5 M).  scale_modifier is render()'s own argument.
            log("dense data point: searching scale_modifier ...")
            lo, hi = 1.0, 6.0
            for _ in range(7):
                mid = 0.5 * (lo + hi)
                n_mid = D.last_num_rendered(render(cam, model, pipe, bg, mid)["render"])
                lo, hi = (mid, hi) if n_mid < args.dense_pairs else (lo, mid)
            scale_mod[0] = 0.5 * (lo + hi)
            n_dense = []
            for c_ in cams[:n_views]:
                o_ = render(c_, model, pipe, bg, scale_mod[0])
                n_dense.append(D.last_num_rendered(o_["render"]))
                del o_
            step_no[0] = 0
            run_steps(12)
            torch.cuda.synchronize()
            sd = timed_regions(3)
            sd1 = timed_regions(3, use_streams=False)
            dt, dt1 = sorted(sd)[1], sorted(sd1)[1]
            Nd = sum(n_dense) / len(n_dense)
            Bd = 304 * P + 548 * V + 116 * Nd + 40 * HW            # V of the unscaled scene: a lower bound
            result["dense"] = {"value": round(args.steps / dt, 2), "unit": "views/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
                               "sequential_views_per_s": round(args.steps / dt1, 2),
                               "steps": args.steps, "scale_modifier": round(scale_mod[0], 4), "N_pairs": int(round(Nd)),
                               "N_pairs_per_camera": n_dense,
                               "pipeline_GBps": round(Bd / (dt / args.steps) / 1e9, 1),
                               "workload": "same scene and cameras, every splat scaled by scale_modifier so that a view "
                                           "emits ~10 M (tile, Gaussian) pairs (BASELINE.md section 3 nominal)"}
            scale_mod[0] = args.scale_modifier
            step_no[0] = 0
        if world == 1 and not args.no_extras:
            result["extras"], result["pgd"] = extras_and_pgd(args, D, dev, model, cams, pipe, bg, gc, streams)
        if world == 1 and not args.no_cpu_baseline:
            sp, sw, sh = (int(x) for x in args.cpu_sample.split(","))
            log(f"cpu baseline (oracle-R) on a {sp}-Gaussian {sw}x{sh} sample ...")
            result["cpu_baseline"] = cpu_baseline(sp, sw, sh)
            log("parity vs oracle-R and CPU time on BASELINE config 1 ...")
            extra = parity_and_cfg1(dev)
            result["cpu_baseline"]["cfg1"] = extra["cfg1_cpu"]
            result["grad_max_rel_err_vs_ref"] = extra["parity"]["grad_max_rel_err"]
            result["parity"] = extra["parity"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
