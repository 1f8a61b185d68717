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
