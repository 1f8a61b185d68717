"""View-sharded data parallelism on CPU (gloo, world_size 2): the all-reduced attribute gradients equal the
single-process sum over the batch's views, and replicas stay identical after the PGD step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gsplat_attack import dist as gdist
from gsplat_attack import pgd
from gsplat_attack.scenes import make_scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _view_grads(model, view: int):
    """A stand-in for render+backward on CPU: any deterministic per-view gradient will do for the plumbing."""
    g = torch.Generator().manual_seed(100 + view)
    return {n: torch.randn(getattr(model, n).shape, generator=g) for n in gdist.ATTACK_PARAMS}


def _as_flat_bucket(model):
    """Re-home the gradients as slices of one buffer, like the fused backward does."""
    names = gdist.ATTACK_PARAMS
    flat = torch.cat([getattr(model, n).grad.reshape(-1) for n in names])
    pos = 0
    for n in names:
        p = getattr(model, n)
        p.grad = flat[pos:pos + p.numel()].view(p.shape)
        pos += p.numel()


def _worker(rank, world, port, n_views, out_dir, flat=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = gdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    model, _, _ = make_scene("hydrant-1k", P=200, n_views=1)
    orig = {n: getattr(model, n).detach().clone() for n in gdist.ATTACK_PARAMS}
    for step in range(2):
        model.zero_grad()
        for v in gdist.views_of_rank(n_views, rank, world):
            for n, gr in _view_grads(model, v + 10 * step).items():
                p = getattr(model, n)
                p.grad = gr if p.grad is None else p.grad + gr
        if flat:
            _as_flat_bucket(model)
            assert gdist._flat_view_of([getattr(model, n).grad for n in gdist.ATTACK_PARAMS]) is not None
        nbytes = gdist.allreduce_attribute_grads(model)
        assert nbytes == sum(getattr(model, n).numel() * 4 for n in gdist.ATTACK_PARAMS)
        pgd.gaussian_color_l2_attack(model, 0.5, 5.0, orig["_features_rest"], orig["_features_dc"])
        pgd.gaussian_position_linf_attack(model, 0.01, 0.05, orig["_xyz"])
    torch.save({n: getattr(model, n).detach() for n in gdist.ATTACK_PARAMS} |
               {"grad_" + n: getattr(model, n).grad for n in gdist.ATTACK_PARAMS}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_views,flat", [(2, False), (5, False), (3, True)])
def test_allreduce_equals_single_process_sum(tmp_path, n_views, flat):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_views, str(tmp_path), flat), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), f"replicas diverged on {k}"
    # single-process reference of the same two steps
    model, _, _ = make_scene("hydrant-1k", P=200, n_views=1)
    orig = {n: getattr(model, n).detach().clone() for n in gdist.ATTACK_PARAMS}
    for step in range(2):
        model.zero_grad()
        for v in range(n_views):
            for n, gr in _view_grads(model, v + 10 * step).items():
                p = getattr(model, n)
                p.grad = gr if p.grad is None else p.grad + gr
        pgd.gaussian_color_l2_attack(model, 0.5, 5.0, orig["_features_rest"], orig["_features_dc"])
        pgd.gaussian_position_linf_attack(model, 0.01, 0.05, orig["_xyz"])
    for n in gdist.ATTACK_PARAMS:
        assert torch.allclose(r0["grad_" + n], getattr(model, n).grad, atol=1e-5), n
        assert torch.allclose(r0[n], getattr(model, n).detach(), atol=1e-5), n


def test_view_partition_covers_batch_once():
    for world in (1, 2, 4, 8):
        for n in (1, 5, 8, 13):
            seen = sorted(v for r in range(world) for v in gdist.views_of_rank(n, r, world))
            assert seen == list(range(n))


def test_single_process_allreduce_is_a_noop():
    model, _, _ = make_scene("hydrant-1k", P=50, n_views=1)
    assert gdist.allreduce_attribute_grads(model) == 0


def test_flat_bucket_detection():
    """Gradients carved out of one buffer are recognised (=> one collective); separate tensors are not."""
    P = 7
    flat = torch.arange(59 * P, dtype=torch.float32)
    cuts = [0, 3 * P, 6 * P, 51 * P, 52 * P, 55 * P, 59 * P]
    views = [flat[cuts[i]:cuts[i + 1]] for i in range(6)]
    shaped = [views[0].view(P, 3), views[1].view(P, 1, 3), views[2].view(P, 15, 3), views[3].view(P, 1),
              views[4].view(P, 3), views[5].view(P, 4)]
    got = gdist._flat_view_of(shaped)
    assert got is not None and got.numel() == 59 * P and got.data_ptr() == flat.data_ptr()
    got.mul_(2.0)
    assert float(shaped[5][-1, -1]) == 2.0 * (59 * P - 1)
    assert gdist._flat_view_of([torch.zeros(3), torch.zeros(3)]) is None
    assert gdist._flat_view_of([flat[0:3], flat[4:8]]) is None            # a hole


def test_run_attack_follows_the_references_global_iteration_budget(monkeypatch):
    """run_attack is the reference's batch schedule (attack.py:463-475, 556-569): one global iteration counter, the slot
    (it + 1) % max_iters == 0 drops the current batch without attacking, a batch that starts late only gets the rest of
    the window, and the model is saved when the LAST pending batch succeeds even if an earlier one was dropped."""
    import torch
    from gsplat_attack import attack as A

    class M:
        def __init__(self):
            for n in A.gdist.ATTACK_PARAMS:
                setattr(self, n, torch.zeros(2, 3))
            self.saved = []

        def save_ply(self, path):
            self.saved.append(path)

    plan = {}      # views of the batch -> (iterations it needs to succeed, or None = never)
    calls = []

    def fake_pgd(model, batch, *, iters, **kw):
        need = plan[tuple(batch)]
        n = iters if need is None or need > iters else need
        calls.append((tuple(batch), iters, n))
        ok = need is not None and need <= iters
        fake_pgd.last_successes = [ok] * len(batch)
        return [0.0] * n
    fake_pgd.last_successes = None
    monkeypatch.setattr(A, "pgd_attack", fake_pgd)

    cams = list(range(7))                                  # 7 views, B = 2: the reference truncates to 6 = 3 batches
    plan.update({(0, 1): 3, (2, 3): None, (4, 5): 2})
    m = M()
    rep = A.run_attack(m, cams, batch_size=2, max_iters=10, success_fn=lambda im, i: True, save_path="x.ply")
    # batch (0,1): 9 slots left in window 0, succeeds after 3 (it = 3); batch (2,3) gets the REST of window 0 (6 attack
    # iterations, it = 9), fails; slot it = 9 drops it (it = 10); batch (4,5) starts window 1 with 9, succeeds after 2
    assert calls == [((0, 1), 9, 3), ((2, 3), 6, 6), ((4, 5), 9, 2)]
    assert [b["views"] for b in rep["batches"]] == [[0, 1], [2, 3], [4, 5]]
    assert [b["success"] for b in rep["batches"]] == [True, False, True] and rep["batches"][1].get("dropped")
    assert rep["iterations"] == 12 and m.saved == ["x.ply"] and rep["saved"] and not rep["all_succeeded"]
    # a last batch that never succeeds: nothing is saved
    calls.clear()
    plan.update({(0, 1): 1, (2, 3): 1, (4, 5): None})
    m2 = M()
    rep = A.run_attack(m2, cams, batch_size=2, max_iters=4, success_fn=lambda im, i: True, save_path="y.ply")
    assert m2.saved == [] and not rep["saved"] and rep["batches"][-1]["views"] == [4, 5] and not rep["batches"][-1]["success"]


def test_a_single_view_batch_must_succeed_itself(monkeypatch):
    """run_attack and pgd_attack share ONE stopping rule (attack.batch_done).  With batch_size = 1 the
    reference's literal `successes >= B - 1` is 0 >= 0: a view whose detector is never fooled would be reported as a
    success and the model saved.  It has to run its window, be dropped, and nothing is saved."""
    from gsplat_attack import attack as A
    assert A.batch_done([False], 1) is False and A.batch_done([True], 1) is True
    assert A.batch_done([True, False], 2) and not A.batch_done([False, False], 2)
    assert A.batch_done([True, True, False], 3) and not A.batch_done([True, False, False], 3)

    class M:
        def __init__(self):
            for n in A.gdist.ATTACK_PARAMS:
                setattr(self, n, torch.zeros(2, 3))
            self.saved = []

        def save_ply(self, path):
            self.saved.append(path)

    def fake_pgd(model, batch, *, iters, success_fn, **kw):
        fake_pgd.last_successes = [bool(success_fn(None, j)) for j in range(len(batch))]
        return [0.0] * (1 if A.batch_done(fake_pgd.last_successes, len(batch)) else iters)
    fake_pgd.last_successes = None
    monkeypatch.setattr(A, "pgd_attack", fake_pgd)
    m = M()
    rep = A.run_attack(m, [0, 1], batch_size=1, max_iters=5, success_fn=lambda im, i: False, save_path="z.ply")
    assert [b["success"] for b in rep["batches"]] == [False, False]
    assert not rep["all_succeeded"] and not rep["saved"] and m.saved == []
    assert all(b.get("dropped") for b in rep["batches"])
    m2 = M()
    rep = A.run_attack(m2, [0, 1], batch_size=1, max_iters=5, success_fn=lambda im, i: True, save_path="z.ply")
    assert rep["all_succeeded"] and rep["saved"] and m2.saved == ["z.ply"]


def _bucket_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    gdist.init_from_env("gloo")
    from diff_gaussian_rasterization import GradBucket
    P = 1000
    res = {}
    for chunks in (1, 4):
        b = GradBucket(P, "cpu")
        b.flat.copy_(torch.randn(59 * P, generator=torch.Generator().manual_seed(7 + rank)))
        ar = gdist.BucketAllReduce(b, chunks)
        if chunks > 1:
            # what gsr_backward_raw_chunked's callback does after each range of K9 is enqueued (ranges end on multiples of 64)
            assert b.chunks == chunks and b.on_chunk is not None
            edges = [0, 256, 512, 768, P]
            for c in range(chunks):
                b.on_chunk(c, edges[c], edges[c + 1])
        assert ar.wait() == 59 * 4 * P
        res[chunks] = b.flat.clone()
    torch.save(res, os.path.join(out_dir, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_all_reduce_by_ranges_on_cpu(tmp_path):
    """BucketAllReduce (gsplat_attack/dist.py): the six slices of every range all-reduced separately give the sum of the
    whole flat bucket, identically on both ranks (world size 2, gloo)."""
    world, port = 2, _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "b0.pt"), torch.load(tmp_path / "b1.pt")
    want = sum(torch.randn(59 * 1000, generator=torch.Generator().manual_seed(7 + r)) for r in range(2))
    for chunks in (1, 4):
        assert torch.equal(r0[chunks], r1[chunks])
        assert torch.equal(r0[chunks], want)


def test_bench_fan_out_parent_reports_a_failing_rank():
    """bench.py --gpus 2 as a plain command starts two ranks itself and touches no GPU in the parent; here (no HIP device)
    every rank exits with bench.py's "needs a HIP device" error and the parent must hand a non-zero code on."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("only meaningful without a device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_REHEARSE_GLOO"] = "1"                          # skips the device-count check: the children must fail themselves
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs a HIP device" in out.stderr and "stopping the other ranks" in out.stderr or out.stderr.count("needs a HIP device") == 2
    # without the rehearsal switch: the parent asks the runtime nothing (even counting devices may initialise
    # HIP before the fork); the ranks find out themselves and the parent hands their failure on
    env.pop("BENCH_REHEARSE_GLOO")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "HIP device" in out.stderr
    # a rank that does not act on SIGTERM (stuck in a collective) is ended by its exact PID after a bounded grace period
    # (no wall-clock bound on the run itself: three cold interpreter starts importing torch take what the machine's load
    # makes them take -- round 5's `< 60 s` failed once in three runs of the suite.  What is asserted is the ORDER of
    # events the parent reports: the failing rank, the SIGTERM, and the SIGKILL of the exact PID after the 2 s grace period;
    # a parent that never kills the hanging rank runs into subprocess's own 300 s timeout, which fails the test.)
    env.update(BENCH_REHEARSE_GLOO="1", BENCH_FANOUT_GRACE_S="2", BENCH_TEST_HANG_RANK="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "SIGKILL" in out.stderr
    assert "still running 2 s after SIGTERM" in out.stderr
    assert out.stderr.index("stopping the other ranks") < out.stderr.index("SIGKILL")
