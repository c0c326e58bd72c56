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


def host_cores():
    """CPU cores this process may actually use: min(logical CPUs, affinity mask, cgroup CPU quota).  On the GPU box a 1-GPU lease is a
    16-CPU share of a 256-thread host (cgroup cpu.max = 1600000/100000); 256 torch threads under that quota run 70 x slower than 16."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sample_b=16, reps=5, warm=2):
    """The reference path restated for the CPU (oracle/edgewise_torch.py, pinned against the reference fixtures), timed on this
    box's host cores as SURVEY.md section 8(d) specifies: torch CPU ops on all cores this process may use (host_cores()), B = 16 images,
    forward + autograd backward,
    median of `reps` after `warm` warm-ups.  The numpy oracle (single-threaded algebra + BLAS) is timed beside it."""
    import numpy as np
    from oracle import edgewise as oe
    from oracle import edgewise_torch as oet
    cores = host_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        m = build_layer_cpu()
        p = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        g = torch.Generator().manual_seed(0)
        x = torch.randn(sample_b, NS["N"], NS["D"], generator=g).requires_grad_(True)
        w = torch.randn(sample_b, NS["N"], NS["D"], generator=g)
        ts = []
        for i in range(warm + reps):
            for t in list(p.values()) + [x]:
                t.grad = None
            t0 = time.perf_counter()
            y = oet.edgewise_layer(x, p, NS["H"], NS["V"], 0.5)
            (y * w).sum().backward()
            ts.append(time.perf_counter() - t0)
        t_torch = sorted(ts[warm:])[reps // 2]
    finally:
        torch.set_num_threads(prev)
    # second figure: the numpy oracle (what the parity tests run), one warm-up, median of 3
    params = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    xn, wn = x.detach().numpy(), w.numpy()
    tn = []
    for i in range(4):
        t0 = time.perf_counter()
        out, cache = oe.module_fwd(xn, params, NS["H"], NS["V"], True, 0.5)
        oe.module_bwd(wn, cache)
        tn.append(time.perf_counter() - t0)
    t_np = sorted(tn[1:])[1]
    return dict(value=sample_b / t_torch, unit="images/s", cores=cores, kind="port",
                sample=f"oracle/edgewise_torch.py (torch CPU restatement, means-only formulation) fwd+bwd, float32, "
                       f"torch.set_num_threads({cores}), B={sample_b} images, median of {reps} after {warm} warm-ups "
                       f"({t_torch:.3f} s per pass)",
                numpy_oracle={"value": sample_b / t_np, "unit": "images/s",
                              "sample": f"oracle/edgewise.py module_fwd+module_bwd, float32, B={sample_b}, median of 3 after 1 warm-up "
                                        f"({t_np:.2f} s per pass); BLAS threads as numpy defaults"})


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


VIT = dict(dim=384, depth=3, heads=6, n_classes=100, img=32)      # BASELINE.json configs[2]: ViT-MoP ~5.4 M parameters


def _self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start one fresh process per GPU through
    torch.distributed.run (this process has not touched the GPU), relay rank 0's JSON line and exit with the children's status."""
    import socket
    import subprocess
    with socket.socket() as sk:                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL / tensor sharing across processes)
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="per-GPU batch (default: the BASELINE config's -- 256 images for layer / vit, 8 sequences for quartet / whisper)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="layer", choices=["layer", "vit", "quartet", "whisper"],
                    help="layer: one EdgewiseMSA layer fwd+bwd (BASELINE.json configs[1], the metric's workload at every N); "
                         "vit: ViT-MoP 5.4 M training step with AdamW (configs[2], 256 images per GPU); "
                         "quartet: GPT-MoP CausalSelfAttention (Quartet) fwd+bwd at T=1024, d=768, 12 heads (configs[3]); "
                         "whisper: Whisper-MoP encoder MultiheadSelfAttention fwd+bwd at T=3000, d=384, 6 heads, bf16 (configs[4], "
                         "fixed per-GPU batch for the 1 -> N leg)")
    ap.add_argument("--graph", action="store_true",
                    help="vit workload only: replay the model's forward + backward as HIP graphs (torch.cuda.make_graphed_callables); "
                         "that step is launch-bound in eager mode (~270 launches for 3.3 ms of kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the multi-process protocol on CPU (gloo): rendezvous, one flat gradient all-reduce per step, "
                         "barrier + max-over-ranks timing, rank-0 JSON; no GPU work, the value is not a measurement")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); they must agree")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist = None
    if args.batch is None:
        args.batch = NS["B"] if args.workload in ("layer", "vit") else 8
    if args.dry_run:
        return dry_run(args, world, rank)
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        # first RCCL contact: a failure names the rank, the device and the error and ends this (fresh) process with a non-zero status
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            assert int(probe.item()) == world, f"all-reduce of ones gave {probe.item()} on {world} ranks"
        except Exception as e:      # noqa: BLE001 -- reported, then fatal
            print(f"bench.py: RCCL start-up failed on rank {rank} (cuda:{local}, {torch.cuda.get_device_name(local)}): {e!r}",
                  file=sys.stderr, flush=True)
            raise SystemExit(3)

    from mop_amd import ops
    from mop_amd.parallel import FlatGradBucket
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    B = args.batch
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    if args.workload == "layer":
        layer = build_layer(dtype)
        params = [p for p in layer.parameters()]
        x = torch.randn(B, NS["N"], NS["D"], device="cuda", dtype=dtype, generator=g).requires_grad_(True)
        dy = torch.randn(B, NS["N"], NS["D"], device="cuda", dtype=dtype, generator=g)
        bucket = FlatGradBucket(params)

        def step():
            for p in params:
                p.grad = None
            x.grad = None
            y = layer(x)
            y.backward(dy)
            if world > 1:  # one flat gradient bucket, one RCCL all-reduce over xGMI
                bucket.allreduce_(average=True)
    elif args.workload in ("quartet", "whisper"):
        from mop_amd.nn import CausalSelfAttention, MultiheadSelfAttention, TransformerConfig
        torch.manual_seed(0)
        if args.workload == "quartet":
            SIB = dict(T=1024, D=768, H=12)
            mod = CausalSelfAttention(TransformerConfig(n_head=SIB["H"], n_embd=SIB["D"], block_size=SIB["T"], dropout=0.0))
            with torch.no_grad():
                mod.mixture.fill_(0.0)                 # de-degenerate the init: sigmoid(-5) makes Quartet a single path
        else:
            SIB = dict(T=3000, D=384, H=6)
            mod = MultiheadSelfAttention(SIB["D"], SIB["H"], 0.0, False, causal=False)
        mod = mod.cuda().to(dtype)
        params = [p for p in mod.parameters()]
        x = torch.randn(B, SIB["T"], SIB["D"], device="cuda", dtype=dtype, generator=g).requires_grad_(True)
        dy = torch.randn(B, SIB["T"], SIB["D"], device="cuda", dtype=dtype, generator=g)
        bucket = FlatGradBucket(params)

        def step():
            for p in params:
                p.grad = None
            x.grad = None
            y = mod(x)
            y.backward(dy)
            if world > 1:
                bucket.allreduce_(average=True)
    else:
        import torch.nn.functional as F
        from mop_amd.nn import ViT_MoP
        from mop_amd.training import DataParallelStep, make_optimizer_and_schedule
        torch.manual_seed(0)                           # identical initial weights on every rank
        model = ViT_MoP(dim=VIT["dim"], depth=VIT["depth"], heads=VIT["heads"], n_classes=VIT["n_classes"], drop_path=0.0).cuda().to(dtype)
        opt, sched = make_optimizer_and_schedule(model, 3e-3, 5e-2, steps=max(args.steps + args.warmup + 1, 2))
        params = [p for p in model.parameters()]
        xi = torch.randn(B, 3, VIT["img"], VIT["img"], device="cuda", generator=g).to(dtype)
        yi = torch.randint(0, VIT["n_classes"], (B,), device="cuda", generator=g)
        net = torch.cuda.make_graphed_callables(model, (xi,)) if args.graph else model
        dp = DataParallelStep(net, opt, lambda out, tgt: F.cross_entropy(out.float(), tgt), sched,
                              params=[p for p in params if p.requires_grad])                              # flat all-reduce inside

        def step():
            dp(xi, yi)

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
    ar_ms = None
    if world > 1 and args.workload != "vit":           # the step's one collective alone (HIP events on the compute stream), for the scaling read-out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            bucket.allreduce_(average=True)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 10

    if rank == 0:
        imgs = B * world * args.steps / dt
        out = {
            "metric": "MoP-attention fwd+bwd images/sec", "value": imgs, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        }
        par = f"dp{world}" if world > 1 else "single"
        if args.workload == "layer":
            out["config"] = {"workload": "EdgewiseMSA layer fwd+bwd (qkv GEMM + libmopk attention core + proj GEMM), "
                                         "BASELINE.json configs[1] shape", "per_gpu_batch": B, "tokens": NS["N"],
                             "dim": NS["D"], "heads": NS["H"], "views": NS["V"], "gate_rank": NS["r"],
                             "share_qkv": True, "gate_mode": "lowrank", "parallelism": par,
                             "grad_allreduce_bytes": 4 * sum(p.numel() for p in params) if world > 1 else 0}
            out["config"]["grad_allreduce_ms"] = ar_ms
            out["roofline"] = roofline(tim, B, args.dtype)
        elif args.workload in ("quartet", "whisper"):
            T, D, H = SIB["T"], SIB["D"], SIB["H"]
            dh = D // H
            if args.workload == "quartet":
                # two full score maps (the row z-normalisation needs every key of a row: 2 x 2 T^2 dh) + the causal half of P V (T^2 dh)
                flop_fwd, key, kname = B * H * 5.0 * T * T * dh, "quartet", "qt_{stats,fwd}_kernel / qt_{rowsum,dq,dkv}_kernel (quartet_flash.hip)"
                out["metric"] = "GPT-MoP QuartetAttention fwd+bwd sequences/sec"
                wl = ("CausalSelfAttention(use_quartet) module fwd+bwd: 5 Linear projections + libmopk Quartet core + o_proj, T=1024, d=768, "
                      "12 heads, causal (BASELINE.json configs[3])")
            else:
                flop_fwd, key, kname = B * H * 4.0 * T * T * dh, "sdpa", "sdpa_flash_{fwd,dq,dkv}_kernel (sdpa_flash.hip)"
                out["metric"] = "Whisper-MoP encoder attention fwd+bwd sequences/sec"
                wl = ("MultiheadSelfAttention module fwd+bwd: q/k/v/o Linears + libmopk SDPA core, T=3000 (80 x 3000 mel frames), d=384, "
                      "6 heads, non-causal (BASELINE.json configs[4]); fixed per-GPU batch")
            out["unit"] = "sequences/s"
            out["config"] = {"workload": wl, "per_gpu_batch": B, "tokens": T, "dim": D, "heads": H, "parallelism": par,
                             "grad_allreduce_bytes": 4 * sum(p.numel() for p in params) if world > 1 else 0, "grad_allreduce_ms": ar_ms}
            fm = sum(tim.get(key + "_fwd", [0.0])) / max(1, len(tim.get(key + "_fwd", [])))
            bm = sum(tim.get(key + "_bwd", [0.0])) / max(1, len(tim.get(key + "_bwd", [])))
            ach = (flop_fwd * 3.5) / ((fm + bm) * 1e-3) / 1e12 if fm + bm > 0 else 0.0
            out["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_BF16_TFLOPS, "traffic": None, "launch_ms": fm + bm,
                               "algorithmic_flop_per_launch": flop_fwd * 3.5,
                               "note": "attention core forward + backward (backward = 2.5 x the forward's flops), HIP events on the launch stream",
                               "fwd": {"launch_ms": fm, "achieved": flop_fwd / (fm * 1e-3) / 1e12 if fm > 0 else 0.0}}
        else:
            out["metric"] = "ViT-MoP 5.4M training images/sec"
            out["config"] = {"workload": "ViT_MoP(dim 384, depth 3, heads 6, 100 classes) training step on 32x32 images: fwd + bwd + "
                                         "one flat gradient all-reduce + AdamW (BASELINE.json configs[2])", "per_gpu_batch": B,
                             "params": sum(p.numel() for p in params), "parallelism": par,
                             "grad_allreduce_bytes": 4 * sum(p.numel() for p in params) if world > 1 else 0,
                             "hip_graph": bool(args.graph)}
        if world == 1 and not args.no_cpu_baseline and args.workload == "layer":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def roofline(tim, B, dtype):
    """MFMA roofline of the dominant pass -- the Edgewise backward core, three launches of ew_fused_bwd_kernel (template parameter
    0 / 1 / 2: mix backward, D-chains, per-view gradients; their rocprofv3 kernel-trace averages add up to `launch_ms`) -- from HIP
    events on the launch stream; `traffic` = fabric bytes of those launches from the committed PMC passes (profiles/)."""
    fwd_ms = sum(tim.get("edgewise_fwd", [0.0])) / max(1, len(tim.get("edgewise_fwd", [])))
    bwd_ms = sum(tim.get("edgewise_bwd", [0.0])) / max(1, len(tim.get("edgewise_bwd", [])))
    # `traffic` comes from separate rocprofv3 --pmc passes (tools/collect_profiles.sh), not from this run: the newest committed file is
    # used only while it is not older than the kernel sources it describes, and its name travels in `traffic_source`
    traffic, tsrc = None, None
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if cands and B == NS["B"] and dtype == "bf16":
        tpath = cands[-1]
        srcs = [os.path.join(ROOT, "mop_amd", "csrc", f) for f in ("edgewise_fused_bwd.hip", "bwd_common.h", "fused_common.h")]
        stale = os.environ.get("MOPK_TRAFFIC_ANY") is None and any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(tpath) + 1 for f in srcs)
        if not stale:
            tj = json.load(open(tpath))
            traffic = sum(v.get("hbm_bytes_per_launch", 0.0) for k, v in tj.items() if "ew_fused_bwd_kernel" in k) or None
            tsrc = os.path.basename(tpath)
    ach = B * CORE_FLOP_BWD / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else 0.0
    ach_fwd = B * CORE_FLOP_FWD / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
    return {"bound": "mfma", "kernel": "ew_fused_bwd_kernel<.., 0|1|2> (backward core: 3 launches)", "achieved": ach,
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
            "launch_ms": bwd_ms, "algorithmic_flop_per_launch": B * CORE_FLOP_BWD,
            "fwd": {"kernel": "ew_fused_fwd_kernel", "launch_ms": fwd_ms, "achieved": ach_fwd, "frac": ach_fwd / PEAK_BF16_TFLOPS},
            "core_fwd_bwd_frac": (B * (CORE_FLOP_FWD + CORE_FLOP_BWD) / ((fwd_ms + bwd_ms) * 1e-3) / 1e12
                                  / PEAK_BF16_TFLOPS) if fwd_ms + bwd_ms > 0 else 0.0}


def dry_run(args, world, rank):
    """CPU rehearsal of the N-rank protocol (tests/test_dist_gloo.py): gloo rendezvous on 127.0.0.1, per step one flat all-reduce of
    a gradient bucket of the workload's real size, barrier + max-over-ranks timing, rank 0 prints the JSON line."""
    import torch.distributed as dist
    from mop_amd.parallel import FlatGradBucket
    if world > 1:
        dist.init_process_group("gloo")
    if args.workload == "layer":
        model = build_layer_cpu()
    elif args.workload == "quartet":
        from mop_amd.nn import CausalSelfAttention, TransformerConfig
        model = CausalSelfAttention(TransformerConfig(n_head=12, n_embd=768, block_size=1024, dropout=0.0))
    elif args.workload == "whisper":
        from mop_amd.nn import MultiheadSelfAttention
        model = MultiheadSelfAttention(384, 6, 0.0, False, causal=False)
    else:
        from mop_amd.nn import ViT_MoP
        torch.manual_seed(0)
        model = ViT_MoP(dim=VIT["dim"], depth=VIT["depth"], heads=VIT["heads"], n_classes=VIT["n_classes"], drop_path=0.0)
    params = [p for p in model.parameters()]
    bucket = FlatGradBucket(params)
    g = torch.Generator().manual_seed(1234 + rank)

    def step():
        for p in params:                               # stand-in for forward/backward: rank-dependent synthetic gradients
            p.grad = torch.full_like(p, float(rank + 1))
        bucket.allreduce_(average=True)

    for _ in range(args.warmup + 1):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    expect = (world + 1) / 2.0                          # mean of 1..world
    ok = all(bool(torch.allclose(p.grad, torch.full_like(p, expect))) for p in params)
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work)", "value": args.batch * world * args.steps / dt, "unit": "images/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "dry-run",
                          "config": {"workload": args.workload, "backend": "gloo", "parallelism": f"dp{world}",
                                     "grad_allreduce_bytes": 4 * sum(p.numel() for p in params), "allreduce_ok": ok}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("dry-run: averaged gradients differ from the expected mean")


if __name__ == "__main__":
    main()
