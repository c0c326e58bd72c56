"""Data-parallel training utilities around the hot path.

* `make_optimizer_and_schedule` -- AdamW + linear warm-up -> cosine, the recipe of the reference's experiment drivers
  (experiments/cifar100_ab5_param_budgets.py:464-479).
* `save_checkpoint` / `load_checkpoint` -- the reference's checkpoint dictionary (mop/training/utils.py:120-170), so files
  are interchangeable with it.
* `DataParallelStep` -- one process per GPU: forward/backward on the local batch shard, ONE flat RCCL all-reduce of all
  gradients (`mop_amd.parallel.FlatGradBucket`), optimizer step.  There is no data-path collective: the attention path
  shards over batch only.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple, Union

import torch
import torch.distributed as dist
from torch import nn, optim

from .parallel import FlatGradBucket


def make_optimizer_and_schedule(model: nn.Module, lr: float, weight_decay: float, steps: int, warmup_frac: float = 0.0
                                ) -> Tuple[optim.Optimizer, optim.lr_scheduler.LRScheduler]:
    opt = optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
    warm = int(max(steps, 1) * max(warmup_frac, 0.0))
    if warm > 0:
        sched = optim.lr_scheduler.SequentialLR(
            opt, [optim.lr_scheduler.LinearLR(opt, start_factor=1e-3, total_iters=warm),
                  optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max(steps - warm, 1))], milestones=[warm])
    else:
        sched = optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max(steps, 1))
    return opt, sched


def save_checkpoint(model: nn.Module, optimizer: optim.Optimizer, epoch: int, loss: float, filepath: str) -> None:
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                "loss": loss}, filepath)


def load_checkpoint(model: nn.Module, optimizer: Optional[optim.Optimizer], filepath: str,
                    device: Union[str, torch.device] = "cpu") -> Dict:
    ckpt = torch.load(filepath, map_location=device)
    model.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None and "optimizer_state_dict" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return ckpt


class DataParallelStep:
    """loss = loss_fn(model(x), y) on the local shard; gradients averaged over ranks with one flat all-reduce."""

    def __init__(self, model: nn.Module, optimizer: optim.Optimizer, loss_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                 scheduler: Optional[optim.lr_scheduler.LRScheduler] = None, params=None):
        """`model` may be any callable (e.g. the graphed callable of `torch.cuda.make_graphed_callables`); then pass `params`."""
        self.model, self.opt, self.loss_fn, self.sched = model, optimizer, loss_fn, scheduler
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        params = [p for p in model.parameters() if p.requires_grad] if params is None else list(params)
        self.bucket = FlatGradBucket(params) if self.world > 1 else None

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(x), y)
        loss.backward()
        if self.bucket is not None:
            self.bucket.allreduce_(average=True)
        self.opt.step()
        if self.sched is not None:
            self.sched.step()
        return loss.detach()
