#!/usr/bin/env python3
"""bench.py -- MoP-attention fwd+bwd images/s on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: forward + backward of one
EdgewiseMSA layer (qkv GEMM -> attention core in libmopk -> proj GEMM), per-GPU batch
256 x 197 tokens x 384 dims, 6 heads, 5 views, low-rank gates r=4 (BASELINE.json
configs[1]); for N>1 ranks each process its own batch shard (weak scaling) and the
parameter gradients are summed with ONE flat RCCL all-reduce per step.
Inputs are synthetic and already resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# the qkv / proj Linear layers are hipBLASLt GEMMs (PyTorch plumbing, 9 % of the step): let PyTorch's TunableOp pick their
# kernels during the untimed warm-up (-2 % step time; opt out with MOPK_BENCH_TUNABLEOP=0)
if os.environ.get("MOPK_BENCH_TUNABLEOP", "1") != "0":
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/mopk_bench_tunableop_%d.csv")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NS = dict(B=256, N=197, D=384, H=6, V=5, r=4)
# algorithmic FLOPs per image per layer, attention core only
CORE_FLOP_FWD = 1069.27e6
CORE_FLOP_BWD = 2.0 * CORE_FLOP_FWD
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md chip table (spec)


def build_layer(dtype):
    from mop_amd.nn import EdgewiseMSA
    torch.manual_seed(0)
    m = EdgewiseMSA(NS["D"], NS["H"], n_views=NS["V"], share_qkv=True, gate_mode="lowrank",
                    gate_rank=NS["r"], gate_init="mix5")
    with torch.no_grad():  # de-degenerate the init: distinct views, live gates
        for n, p in m.named_parameters():
            if n.endswith("_scale"):
                p.add_(0.1 * torch.randn_like(p))
            elif "edge_head" in n and n.endswith("weight"):
                p.mul_(3.0)
    return m.cuda().to(dtype)


def cpu_baseline(sample_b=32, reps=5):
    """The numpy oracle (a port of the reference path) timed on this box's host cores."""
    import numpy as np
    from oracle import edgewise as oe
    m = build_layer_cpu()
    params = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(0)
    x = rng.standard_normal((sample_b, NS["N"], NS["D"]), dtype=np.float32)
    w = rng.standard_normal((sample_b, NS["N"], NS["D"]), dtype=np.float32)
    ts = []
    for i in range(reps + 1):
        t0 = time.perf_counter()
        out, cache = oe.module_fwd(x, params, NS["H"], NS["V"], True, 0.5)
        oe.module_bwd(w, cache)
        ts.append(time.perf_counter() - t0)
    t = sorted(ts[1:])[len(ts[1:]) // 2]
    return dict(value=sample_b / t, unit="images/s", cores=os.cpu_count(), kind="port",
                sample=f"oracle/edgewise.py module_fwd+module_bwd, float32, B={sample_b} images, "
                       f"median of {reps} after 1 warm-up ({t:.2f} s per pass)")


def build_layer_cpu():
    from mop_amd.nn import EdgewiseMSA
    torch.manual_seed(0)
    m = EdgewiseMSA(NS["D"], NS["H"], n_views=NS["V"], share_qkv=True, gate_mode="lowrank",
                    gate_rank=NS["r"], gate_init="mix5")
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("_scale"):
                p.add_(0.1 * torch.randn_like(p))
            elif "edge_head" in n and n.endswith("weight"):
                p.mul_(3.0)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=NS["B"], help="per-GPU batch (default = BASELINE config)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from mop_amd import ops
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    layer = build_layer(dtype)
    params = [p for p in layer.parameters()]
    B = args.batch
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    x = torch.randn(B, NS["N"], NS["D"], device="cuda", dtype=dtype, generator=g).requires_grad_(True)
    dy = torch.randn(B, NS["N"], NS["D"], device="cuda", dtype=dtype, generator=g)
    from mop_amd.parallel import FlatGradBucket
    bucket = FlatGradBucket(params)

    def step():
        for p in params:
            p.grad = None
        x.grad = None
        y = layer(x)
        y.backward(dy)
        if world > 1:  # one flat gradient bucket, one RCCL all-reduce over xGMI
            bucket.allreduce_(average=True)

    # one untimed initialisation pass (code-object loading, hipBLASLt kernel selection by TunableOp, allocator growth), so that the
    # W warm-up steps -- and with --warmup 0 the timed steps -- run the steady-state path
    step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    ops.enable_timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tim = ops.timing_results()
    ops.enable_timing(False)

    if rank == 0:
        import ctypes as C
        from mop_amd import _lib as L
        a = L.EdgewiseArgs()
        a.B, a.H, a.N, a.dk, a.V, a.r = B, NS["H"], NS["N"], NS["D"] // NS["H"], NS["V"], NS["r"]
        a.precision = L.PREC_BF16 if args.dtype == "bf16" else L.PREC_FP32
        a.io_dtype = L.MOPK_BF16 if args.dtype == "bf16" else L.MOPK_F32
        a.path = ops.LAST_PATH.get("edgewise_bwd", L.PATH_AUTO)
        dom_bwd = L.lib().mopk_edgewise_dominant_kernel(C.byref(a), 1).decode()
        traffic = None  # HBM bytes per launch from rocprofv3 PMC passes (profiles/*hbm_traffic.json), if collected
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tpath) and B == NS["B"]:
            traffic = json.load(open(tpath)).get(dom_bwd, {}).get("hbm_bytes_per_launch")
        fwd_ms = sum(tim.get("edgewise_fwd", [0.0])) / max(1, len(tim.get("edgewise_fwd", [])))
        bwd_ms = sum(tim.get("edgewise_bwd", [0.0])) / max(1, len(tim.get("edgewise_bwd", [])))
        # dominant launch = the backward core (2/3 of the algorithmic FLOPs)
        ach = B * CORE_FLOP_BWD / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else 0.0
        ach_fwd = B * CORE_FLOP_FWD / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
        imgs = B * world * args.steps / dt
        out = {
            "metric": "MoP-attention fwd+bwd images/sec", "value": imgs, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "EdgewiseMSA layer fwd+bwd (qkv GEMM + libmopk attention core + proj GEMM), "
                                   "BASELINE.json configs[1] shape", "per_gpu_batch": B, "tokens": NS["N"],
                       "dim": NS["D"], "heads": NS["H"], "views": NS["V"], "gate_rank": NS["r"],
                       "share_qkv": True, "gate_mode": "lowrank",
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": {"bound": "mfma", "kernel": dom_bwd, "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "launch_ms": bwd_ms, "algorithmic_flop_per_launch": B * CORE_FLOP_BWD,
                         "fwd": {"launch_ms": fwd_ms, "achieved": ach_fwd, "frac": ach_fwd / PEAK_BF16_TFLOPS},
                         "core_fwd_bwd_frac": (B * (CORE_FLOP_FWD + CORE_FLOP_BWD) / ((fwd_ms + bwd_ms) * 1e-3) / 1e12
                                               / PEAK_BF16_TFLOPS) if fwd_ms + bwd_ms > 0 else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
