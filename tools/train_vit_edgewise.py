#!/usr/bin/env python3
"""Data-parallel ViTEdgewise training on synthetic CIFAR-shaped data (one process per GPU, RCCL gradient all-reduce).

    python tools/train_vit_edgewise.py --steps 50                                   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/train_vit_edgewise.py --steps 50                                           # 8 GPUs, 256 images each

Recipe of the reference's drivers (AdamW, linear warm-up -> cosine; experiments/cifar100_ab5_param_budgets.py:464-479);
the attention layers run the fused gfx950 Edgewise kernels.  Writes a reference-format checkpoint with --ckpt.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.nn.functional as F

from mop_amd.nn import ViTEdgewise
from mop_amd.training import DataParallelStep, make_optimizer_and_schedule, save_checkpoint


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--heads", type=int, default=6)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--lr", type=float, default=3e-3)
    ap.add_argument("--weight-decay", type=float, default=5e-2)
    ap.add_argument("--warmup-frac", type=float, default=0.1)
    ap.add_argument("--ckpt", type=str, default="")
    ap.add_argument("--graph", action="store_true",
                    help="capture the model's forward + backward into HIP graphs (torch.cuda.make_graphed_callables) and replay them; "
                         "the optimizer step stays eager")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl")                    # RCCL over xGMI
    torch.manual_seed(0)                                   # identical initial weights on every rank
    model = ViTEdgewise(dim=args.dim, depth=args.depth, heads=args.heads, n_classes=100, n_views=args.views, share_qkv=True,
                        gate_mode="lowrank", gate_rank=4, gate_init="mix5", drop_path=0.0).cuda().to(torch.bfloat16)
    opt, sched = make_optimizer_and_schedule(model, args.lr, args.weight_decay, args.steps, args.warmup_frac)
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)            # each rank draws its own shard
    x = torch.randn(args.batch, 3, 32, 32, device="cuda", generator=g).to(torch.bfloat16)
    y = torch.randint(0, 100, (args.batch,), device="cuda", generator=g)
    net = torch.cuda.make_graphed_callables(model, (x,)) if args.graph else model   # same parameters; kernels are stream-ordered, no host syncs
    step = DataParallelStep(net, opt, lambda out, tgt: F.cross_entropy(out.float(), tgt), sched,
                            params=[p for p in model.parameters() if p.requires_grad])
    for _ in range(3):
        step(x, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(x, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"world={world} {args.steps} steps: {dt / args.steps * 1e3:.2f} ms/step, {world * args.batch * args.steps / dt:.0f} img/s, "
              f"loss {float(loss):.3f}")
        if args.ckpt:
            save_checkpoint(model, opt, args.steps, float(loss), args.ckpt)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
