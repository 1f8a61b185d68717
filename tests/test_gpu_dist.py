"""The N > 1 code path with the REAL rasteriser: two processes (gloo; both on the box's one GPU) shard a batch of views,
run the fused backward, all-reduce the flat 59-floats-per-Gaussian bucket it hands out and take the PGD step; the
replicas end up identical and equal to the single-process run over the whole batch.  (RCCL itself needs N GPUs: this is
everything short of the transport.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

GROUPS = ("color", "position", "scaling", "rotation", "opacity")
KW = dict(P=20000, width=320, height=192)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _attack(model, cams, **kw):
    from gsplat_attack.attack import pgd_attack
    return pgd_attack(model, cams, iters=2, alpha=0.05, epsilon=0.5, groups=GROUPS, norm="l2", streams=1, **kw)


def _worker(rank, world, port, n_views, out_dir, accumulate):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from gsplat_attack import dist as gdist
    from gsplat_attack.renderer import PipelineParams, render
    from gsplat_attack.scenes import make_scene
    r, w, _ = gdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model, cams, _ = make_scene("nyc-1M", device=dev, n_views=n_views, **KW)
    # the fused backward hands out ONE flat bucket: the collective is a single all-reduce of 59 floats per Gaussian
    render(cams[rank], model, PipelineParams(skip_objects=True), torch.zeros(3, device=dev))["render"].sum().backward()
    flat = gdist._flat_view_of([getattr(model, n).grad for n in gdist.ATTACK_PARAMS])
    assert flat is not None and flat.numel() == 59 * model.get_xyz.shape[0]
    assert gdist.allreduce_attribute_grads(model) == flat.numel() * 4
    model.zero_grad()
    hist = _attack(model, cams, accumulate_grads=accumulate)
    torch.save({n: getattr(model, n).detach().cpu() for n in gdist.ATTACK_PARAMS} | {"hist": hist},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_views,accumulate", [(4, False), (3, True)])
def test_two_rank_attack_equals_the_single_process_batch(tmp_path, n_views, accumulate):
    from gsplat_attack import dist as gdist
    from gsplat_attack.scenes import make_scene
    import diff_gaussian_rasterization as D
    D._load()
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_views, str(tmp_path), accumulate), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    for k in gdist.ATTACK_PARAMS:
        assert torch.equal(r0[k], r1[k]), f"replicas diverged on {k}"      # an all-reduce returns the same bits everywhere
    assert r0["hist"] == r1["hist"]
    model, cams, _ = make_scene("nyc-1M", device=torch.device("cuda:0"), n_views=n_views, **KW)
    start = {n: getattr(model, n).detach().clone() for n in gdist.ATTACK_PARAMS}
    hist = _attack(model, cams, accumulate_grads=accumulate)
    assert hist == pytest.approx(r0["hist"], rel=1e-4)
    for n in gdist.ATTACK_PARAMS:
        moved = (getattr(model, n).detach() - start[n]).abs().max().item()
        assert moved > 0, n
        # same sum of per-view gradients in another order: float32 rounding of a normalised step
        assert (getattr(model, n).detach().cpu() - r0[n]).abs().max().item() <= 2e-3 * moved, n


def _nccl_worker(rank, world, port, n_views, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from gsplat_attack import dist as gdist
    from gsplat_attack.scenes import make_scene
    r, w, local = gdist.init_from_env("nccl")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    model, cams, _ = make_scene("nyc-1M", device=dev, n_views=n_views, **KW)
    hist = _attack(model, cams)
    torch.save({n: getattr(model, n).detach().cpu() for n in gdist.ATTACK_PARAMS} | {"hist": hist},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI); lights up by itself on a bigger box")
def test_two_ranks_on_two_devices_over_rccl(tmp_path):
    """The same attack with one rank per DEVICE and the bucket all-reduced by RCCL ("nccl" backend): replicas bitwise
    equal, equal to the single-process batch within float32 rounding of a normalised step."""
    from gsplat_attack import dist as gdist
    from gsplat_attack.scenes import make_scene
    world, port, n_views = 2, _free_port(), 4
    mp.spawn(_nccl_worker, args=(world, port, n_views, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    for k in gdist.ATTACK_PARAMS:
        assert torch.equal(r0[k], r1[k]), f"replicas diverged on {k}"
    model, cams, _ = make_scene("nyc-1M", device=torch.device("cuda:0"), n_views=n_views, **KW)
    start = {n: getattr(model, n).detach().clone() for n in gdist.ATTACK_PARAMS}
    hist = _attack(model, cams)
    assert hist == pytest.approx(r0["hist"], rel=1e-4)
    for n in gdist.ATTACK_PARAMS:
        moved = (getattr(model, n).detach() - start[n]).abs().max().item()
        assert (getattr(model, n).detach().cpu() - r0[n]).abs().max().item() <= 2e-3 * moved, n


def test_bench_pgd_loop_rehearsal_prints_the_multi_gpu_fields(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run), rehearsed on the one GPU over gloo: the N > 1
    line times whole PGD iterations and carries allreduce_ms / bytes_reduced / views_per_rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_REHEARSE_GLOO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--regions", "2", "--views-per-rank", "2", "--P", "20000", "--width", "320", "--height", "192", "--no-cpu-baseline",
           "--no-extras"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["views_per_rank"] == 2 and d["config"]["loop"] == "pgd"
    assert d["bytes_reduced"] == 59 * 4 * 20000 and d["allreduce_ms"] > 0
    assert len(d["regions_ms"]) == 2 and d["value"] > 0 and d["steps"] == 2
    assert abs(d["value"] - 2 * 2 * 2 / (sorted(d["regions_ms"])[1] * 1e-3)) / d["value"] < 0.02
