"""world_size-2 gloo test of the data-parallel gradient exchange (the only collective on the path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mop_amd.parallel import FlatGradBucket, shard_batch
    torch.manual_seed(0)  # same init on every rank
    from mop_amd.nn import EdgewiseMSA
    m = EdgewiseMSA(32, 2, n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2)
    # synthetic per-rank gradients: rank-dependent but known
    for i, p in enumerate(m.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    list(m.parameters())[3].grad = None  # a parameter that received no gradient on this rank
    b = FlatGradBucket(m.parameters())
    b.allreduce_(average=True)
    ok = True
    for i, p in enumerate(m.parameters()):
        exp = sum((r + 1) * (i + 1) for r in range(world)) / world if i != 3 else 0.0
        ok &= bool(torch.allclose(p.grad, torch.full_like(p, exp)))
    s, e = shard_batch(10, rank, world)
    q.put((rank, ok, (s, e)))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert [r[2] for r in res] == [(0, 5), (5, 10)]


def test_shard_batch_covers_everything():
    from mop_amd.parallel import shard_batch
    for n in (1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_batch(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mop_amd.parallel import shard_batch
    from mop_amd.training import DataParallelStep, make_optimizer_and_schedule
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    x, y = torch.randn(12, 8), torch.randn(12, 3)
    opt, sched = make_optimizer_and_schedule(m, lr=1e-2, weight_decay=0.0, steps=4, warmup_frac=0.25)
    step = DataParallelStep(m, opt, torch.nn.functional.mse_loss, sched)
    s, e = shard_batch(12, rank, world)
    for _ in range(4):
        step(x[s:e], y[s:e])
    q.put((rank, [p.detach().numpy().copy() for p in m.parameters()]))      # by value: the process exits before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_matches_single_process_world2():
    """2 ranks x half batch with the flat gradient all-reduce == 1 process on the full batch (equal shards, mean loss)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from mop_amd.training import DataParallelStep, make_optimizer_and_schedule
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    x, y = torch.randn(12, 8), torch.randn(12, 3)
    opt, sched = make_optimizer_and_schedule(m, lr=1e-2, weight_decay=0.0, steps=4, warmup_frac=0.25)
    step = DataParallelStep(m, opt, torch.nn.functional.mse_loss, sched)
    for _ in range(4):
        step(x, y)
    for a, b, c in zip(m.parameters(), res[0], res[1]):
        assert (b == c).all()                                       # ranks stay bit-identical
        assert torch.allclose(a.detach(), torch.from_numpy(b), atol=1e-6)


def _bench(*argv, env=None, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("workload,nbytes", [("layer", 4 * 596001), ("vit", 4 * 5397972), ("quartet", 4 * (6 * 768 * 768 + 2)),
                                             ("whisper", 4 * 4 * 384 * 384)])
def test_bench_self_launch_two_ranks_gloo(workload, nbytes):
    """`python bench.py --gpus 2` as a plain command: bench.py starts the ranks itself (torch.distributed.run, 127.0.0.1), every rank
    all-reduces ONE flat gradient bucket of the workload's real size per step, rank 0 prints one JSON line (CPU rehearsal: gloo)."""
    import json
    r = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--workload", workload)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["config"]["parallelism"] == "dp2" and j["scaling"] == "weak"
    assert j["config"]["grad_allreduce_bytes"] == nbytes and j["config"]["allreduce_ok"] is True


def test_bench_refuses_a_world_size_mismatch():
    r = _bench("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)
