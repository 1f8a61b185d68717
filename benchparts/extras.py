"""The untimed extras of the N = 1 line and its PGD blocks.

Part of bench.py (the repo-root benchmark driver), split out in round 6: bench.py keeps the command line, the timed regions
of the headline metric and the assembly of the ONE JSON line; this module holds `extras_and_pgd` and the CU-mask stream experiment."""
from __future__ import annotations

import contextlib
import time

import torch

from .common import log


def masked_streams(dev, spec: str):
    """HIP streams restricted to sets of compute units (hipExtStreamCreateWithCUMask), wrapped for torch."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    out = []
    with torch.cuda.device(dev):
        torch.cuda.current_stream(dev).synchronize()       # the runtime is initialised
        for m in spec.split(";"):
            words = [int(w, 16) for w in m.split(",") if w.strip()]
            arr = (ctypes.c_uint32 * len(words))(*words)
            h = ctypes.c_void_p(None)
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(len(words)), arr)
            if rc != 0 or not h.value:
                raise SystemExit(f"hipExtStreamCreateWithCUMask failed ({rc}) for mask {m}")
            out.append(torch.cuda.ExternalStream(h.value, device=dev))
    return out


def extras_and_pgd(args, D, dev, model, cams, pipe, bg, gc, streams):
    """Untimed extras of the N = 1 line (SURVEY.md section 8d "also report"): forward-only and SH-only rates, and whole
    PGD iterations split into phases."""
    from gsplat_attack.attack import PhaseTimer, SurrogateDetector, pgd_attack
    from gsplat_attack.renderer import PipelineParams, render
    cam = cams[0]
    log("extras: forward-only views, SH-only gradients ...")
    with torch.no_grad():
        run_fwd = min(args.steps, 100)
        for s_ in (streams or []):
            s_.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(run_fwd):
            if streams is None:
                render(cam, model, pipe, bg)
            else:
                with torch.cuda.stream(streams[i % len(streams)]):
                    render(cam, model, pipe, bg)
        torch.cuda.synchronize()
        fwd_rate = run_fwd / (time.perf_counter() - t0)
    col_rate = None
    if not args.color_only:
        # gradients on the SH coefficients only (BASELINE configs 2/3): geometry frozen, lighter K7 / K8+K9
        frozen = [getattr(model, n_) for n_ in ("_xyz", "_scaling", "_rotation", "_opacity")]
        for p_ in frozen:
            p_.requires_grad_(False)
        pipe_c = PipelineParams(skip_objects=not args.objects, viewspace_grad=False)
        n_col = min(args.steps, 150)

        def col_steps(n):
            for i in range(n):
                ctx_ = torch.cuda.stream(streams[i % len(streams)]) if streams is not None else contextlib.nullcontext()
                with ctx_:
                    model.zero_grad()
                    render(cam, model, pipe_c, bg)["render"].backward(gc)
        for s_ in (streams or []):
            s_.wait_stream(torch.cuda.current_stream(dev))
        col_steps(6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        col_steps(n_col)
        torch.cuda.synchronize()
        col_rate = n_col / (time.perf_counter() - t0)
        # the same over the ring of cameras (what a batch of attack views is), without and with each camera's binning
        # kept across renders (RenderCache -> gsr_ctx_rerender: the geometry is frozen, so only the colour kernel and the
        # compositor run from a camera's second render on); bit-equal results (tests/test_gpu_rerender.py)
        from diff_gaussian_rasterization import RenderCache
        ring_rates = []
        for cache_ in (None, RenderCache()):
            pipe_r = PipelineParams(skip_objects=not args.objects, viewspace_grad=False, render_cache=cache_)

            def ring_steps(n):
                for i in range(n):
                    ctx_ = torch.cuda.stream(streams[i % len(streams)]) if streams is not None else contextlib.nullcontext()
                    with ctx_:
                        model.zero_grad()
                        render(cams[i % len(cams)], model, pipe_r, bg)["render"].backward(gc)
            ring_steps(2 * len(cams))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ring_steps(n_col)
            torch.cuda.synchronize()
            ring_rates.append(n_col / (time.perf_counter() - t0))
            del pipe_r, cache_
        for p_ in frozen:
            p_.requires_grad_(True)
    extras = {"fwd_only_views_per_s": round(fwd_rate, 1),
              "sh_grads_only_views_per_s": None if col_rate is None else round(col_rate, 1)}
    if not args.color_only:
        # The drop-in regime: what the reference's UNCHANGED render() reaches (gaussian_renderer/__init__.py:53-95) when a
        # user installs this package without gsplat_attack.patch_reference() -- activated tensors (exp / sigmoid /
        # normalize / cat and their backward as PyTorch kernels) through GaussianRasterizer.forward, and ALWAYS the 16
        # object channels (`sh_objs = pc.get_objects`).  Same views, same dL/dC, gradients to all raw parameters.
        log("extras: the drop-in regime (classic activated-tensor surface + 16 object channels) ...")
        pipe_d = PipelineParams(skip_objects=False, fused_activations=False)
        n_d = min(args.steps, 60)

        dmodel = [model]

        pipe_cur = [pipe_d]

        def dropin_steps(n, sts):
            m_ = dmodel[0]
            pipe_d = pipe_cur[0]
            for s_ in (sts or []):
                s_.wait_stream(torch.cuda.current_stream(dev))
            for i in range(n):
                ctx_ = torch.cuda.stream(sts[i % len(sts)]) if sts else contextlib.nullcontext()
                with ctx_:
                    m_.zero_grad()
                    render(cams[i % len(cams)], m_, pipe_d, bg)["render"].backward(gc)
            for s_ in (sts or []):
                torch.cuda.current_stream(dev).wait_stream(s_)

        def dropin_rates():
            rates_ = []
            for sts in (streams, None):
                dropin_steps(8, sts)
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    dropin_steps(n_d, sts)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                rates_.append(n_d / sorted(ts)[1])
            D.profile(True)
            dropin_steps(6, None)
            torch.cuda.synchronize()
            st_ = {k: round(ms / 6, 4) for k, (ms, _) in D.profile_read().items()}
            D.profile(False)
            return rates_, st_
        rates, st_d = dropin_rates()
        # the attack's case: every object feature zero (reference scene/gaussian_model.py:528, combine_splats) -- the classic
        # binding then composites without the object channels (GSR_FLAG_OBJECTS_FOR_BACKWARD_ONLY), same image and gradients
        dmodel[0] = model.clone()
        with torch.no_grad():
            dmodel[0]._objects_dc.zero_()
        rates_z, st_z = dropin_rates()
        # GSR_PATCH_REFERENCE=1 (or gsplat_attack.patch_reference()): the reference's render() rebound to the fused one -- the raw
        # parameters straight into the kernels, the reference's own pipe (objects on); same two models
        pipe_cur[0] = PipelineParams(skip_objects=False)
        rates_pz, st_pz = dropin_rates()
        dmodel[0] = model
        rates_p, st_p = dropin_rates()
        pipe_cur[0] = pipe_d
        extras["dropin"] = {"value": round(rates[0], 1), "sequential_views_per_s": round(rates[1], 1), "unit": "views/s",
                            "steps": n_d, "streams": len(streams) if streams else 1, "stages_ms": st_d,
                            "what": "render() on the classic surface with object channels on (--classic --objects): the "
                                    "configuration the reference's unchanged gaussian_renderer.render() reaches; median "
                                    "of 3 regions; `value` pipelined over the streams, `sequential` on one",
                            "zero_object_features": {
                                "value": round(rates_z[0], 1), "sequential_views_per_s": round(rates_z[1], 1), "stages_ms": st_z,
                                "what": "the same with every object feature zero, as in the attack's combined scenes "
                                        "(reference scene/gaussian_model.py:528): the binding finds that out once per tensor "
                                        "version and composites without the 16 object channels; image, object map (zeros) "
                                        "and gradients equal the object variant's (tests/test_gpu_zero_objects.py)"},
                            "env_patched": {
                                "value": round(rates_p[0], 1), "sequential_views_per_s": round(rates_p[1], 1), "stages_ms": st_p,
                                "zero_object_features": {"value": round(rates_pz[0], 1),
                                                         "sequential_views_per_s": round(rates_pz[1], 1), "stages_ms": st_pz},
                                "what": "the reference UNCHANGED but started with GSR_PATCH_REFERENCE=1 in the environment "
                                        "(diff_gaussian_rasterization._auto_patch_reference rebinds gaussian_renderer.render "
                                        "to the fused render(): raw parameters into the kernels, no activated copies, no "
                                        "PyTorch backward of the getters; tests/test_auto_patch.py); object channels on as "
                                        "the reference's render() asks, and the same with all-zero object features"}}
    if col_rate is not None:
        extras["sh_grads_only_ring_views_per_s"] = round(ring_rates[0], 1)
        extras["sh_grads_only_ring_binning_kept_views_per_s"] = round(ring_rates[1], 1)

    log("pgd: config 3 (colour L2, PGD-20, B = 1) and config 4 on one GPU (8 views, five groups) ...")
    det = SurrogateDetector().to(dev)
    never = lambda im, i: False                            # noqa: E731 -- the success check runs, the loop never stops on it

    def measure(name, views, groups, iters, n_streams, rerender, cache_binning=True, batched=True, checks_batched=True):
        m = model.clone()
        kw = dict(groups=groups, loss_fn=det, streams=n_streams, alpha=0.5, epsilon=5.0, cache_binning=cache_binning,
                  batched=batched)
        if not checks_batched:
            from gsplat_attack.renderer import PipelineParams
            kw["pipe"] = PipelineParams(skip_objects=True)
            kw["pipe"].batched_checks = False
        if rerender:
            kw.update(success_fn=never, background=None)
        pgd_attack(m, views, iters=3, **kw)                # warm-up
        torch.cuda.synchronize()
        recs = []
        pgd_attack(m, views, iters=iters, log=recs.append, **kw)
        torch.cuda.synchronize()
        per_it = sorted(r["seconds"] for r in recs)
        wall = per_it[len(per_it) // 2] * 1e3              # median iteration (each one ends with a synchronise)
        out = {"iteration_ms": round(wall, 3), "iteration_ms_all": [round(x * 1e3, 3) for x in per_it],
               "views": len(views), "groups": list(groups), "streams": n_streams, "what": name,
               # colour-only attacks: each camera's rasteriser context (projection, sorts, tile lists) kept in HBM after its
               # first render and re-used while the geometry tensors are untouched (gsr_ctx_rerender); same bits
               "binning_kept": bool(cache_binning and tuple(groups) == ("color",)),
               # all-attribute attacks on two or more views: the views of an iteration go through one launch chain
               # (a colour attack's batch keeps ITS context: the batch's colour kernel + one compositor launch per render)
               "views_batched": bool(batched and len(views) >= 2),
               "success_renders_batched": bool(rerender and checks_batched and len(views) >= 2)}
        if n_streams == 1:
            # phase split on one stream: HIP events at the phase boundaries + the library's own stage events
            tm = PhaseTimer()
            D.profile(True)
            pgd_attack(m, views, iters=iters, timer=tm, **kw)
            ph = {k: v / iters for k, v in tm.totals_ms().items()}
            st = {k: ms / iters for k, (ms, _) in D.profile_read().items()}
            D.profile(False)
            r_bwd = st["render_bwd"] + st["preprocess_bwd"]
            out["phases_ms"] = {
                "raster_forward": round(ph["render"], 3),
                "raster_backward": round(r_bwd, 3),
                "detector": round(ph["loss"] + max(ph["backward"] - r_bwd, 0.0), 3),
                "gradient_accumulation": round(ph["reduce"], 3),
                "step": round(ph["step"], 3),
                "rerender": round(ph["rerender"], 3),
            }
            tot = sum(out["phases_ms"].values())
            in_scope_overhead = out["phases_ms"]["gradient_accumulation"] + out["phases_ms"]["step"]
            out["overhead_frac"] = round(in_scope_overhead / max(tot, 1e-9), 4)
            out["phases_note"] = ("HIP events on the one stream (with the library's per-stage events on, which add a few "
                                  "microseconds per stage); detector = surrogate forward + its share of backward; "
                                  "overhead_frac = (gradient_accumulation + step) / sum of phases")
        del m
        return out
    pgd = {
        "cfg3": measure("BASELINE config 3: DAGGER PGD-20, L2 on the SH colour only, ONE view per iteration, forward-only "
                        "re-render after every step (attack.py:522-530), surrogate detector; the camera's binning is kept "
                        "across iterations (the geometry is frozen)", cams[:1], ("color",), 20, 1, True),
        "cfg3_rebinned_every_render": measure("config 3 with every render running the whole forward (cache_binning=False): "
                                              "what a rasteriser without kept contexts does", cams[:1], ("color",), 20, 1, True,
                                              cache_binning=False),
        "cfg3_8views": measure("config 3's colour attack on a batch of 8 views: the views as ONE batch whose binning is kept "
                               "(gsr_ctx_rerender on the batch context: one colour kernel reading every SH row once + one "
                               "compositor launch per render, one backward composite, the SH gradients written once)", cams[:8],
                               ("color",), 6, 1, True),
        "cfg3_8views_per_view_contexts": measure("the same with one kept context and one render() + backward per view, the "
                                                 "views dealt over 4 streams (round 5's form)", cams[:8], ("color",), 6,
                                                 max(args.streams, 1), True, batched=False),
        "cfg3_8views_rebinned_every_render": measure("the batch with every render the whole forward", cams[:8], ("color",), 6,
                                                     1, True, cache_binning=False),
        "cfg4_one_gpu": measure("BASELINE config 4 on one GPU: 8 views per iteration, L2 on {colour, position, scaling, "
                                "rotation, opacity}, one stream, the 8 views as ONE batch (gsr_forward_raw_batch)", cams[:8],
                                ("color", "position", "scaling", "rotation", "opacity"), 6, 1, False),
        "cfg4_one_gpu_with_success_renders": measure("config 4 on one GPU with the forward-only success render of every view "
                                                     "after the step (reference attack.py:522-530; SURVEY 8d: B x (fwd + bwd) + B x "
                                                     "fwd): the 8 success renders as ONE forward-only batch", cams[:8],
                                                     ("color", "position", "scaling", "rotation", "opacity"), 6, 1, True),
        "cfg4_one_gpu_with_success_renders_per_view": measure("the same with one forward per success render", cams[:8],
                                                              ("color", "position", "scaling", "rotation", "opacity"), 6, 1,
                                                              True, checks_batched=False),
        "cfg4_one_gpu_pipelined": measure("the same with the 8 views dealt over 4 streams, one render() per view", cams[:8],
                                          ("color", "position", "scaling", "rotation", "opacity"), 6, max(args.streams, 1), False,
                                          batched=False),
        "cfg4_one_gpu_per_view": measure("the same on one stream with one render() + backward per view (round 5's loop)", cams[:8],
                                         ("color", "position", "scaling", "rotation", "opacity"), 6, 1, False, batched=False),
    }
    return extras, pgd
