"""Token-matrix Linear: same parameters / state_dict keys / forward as ``torch.nn.Linear`` (the reference's qkv, proj and
MLP projections, e.g. attention_variants.py:374-383), with a weight-gradient GEMM shaped for MI355X.

The projections around the attention cores see M = batch x tokens rows (50 432 at the bench shape) and dim <= 1152 columns, so
``dW = dY^T X`` is a GEMM whose *reduction* dimension is M and whose output is tiny.  hipBLASLt runs that as one skinny GEMM at
~110-270 TFLOP/s (162 us for 1152x384, 121 us for 384x384, measured with TunableOp on).  Cutting M into S slices, taking
the S partial products as one batched GEMM (fp32 accumulation inside each slice) and summing them in fp32 is 2.4-2.7x faster
(68 / 45 us) and slightly more accurate.  Forward and dX stay plain hipBLASLt GEMMs.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

_MIN_ROWS = 4096          # below this the single GEMM is not reduction-bound


def weight_grad_splits(rows: int, out_features: int) -> int:
    """number of M-slices for the batched weight-gradient GEMM: the largest power of two that divides ``rows``, capped at
    16 (32 for narrow outputs, where more slices are needed to fill 256 CUs)"""
    if rows < _MIN_ROWS:
        return 1
    cap = 32 if out_features <= 512 else 16
    s = 1
    while s < cap and rows % (2 * s) == 0:
        s *= 2
    return s


def _weight_grad(dy2: torch.Tensor, x2: torch.Tensor, n_out: int) -> torch.Tensor:
    rows = x2.shape[0]
    s = weight_grad_splits(rows, n_out)
    if s > 1:
        part = torch.bmm(dy2.reshape(s, rows // s, n_out).transpose(1, 2), x2.reshape(s, rows // s, x2.shape[1]))
        return part.sum(0, dtype=torch.float32).to(dy2.dtype)
    return dy2.t() @ x2


class _ResidualLinearFn(torch.autograd.Function):
    """``res + x @ W^T``: the residual add rides in the GEMM epilogue (hipBLASLt beta = 1), its gradient is the incoming one."""

    @staticmethod
    def forward(ctx, res, x, weight):
        ctx.save_for_backward(x, weight)
        n_out, n_in = weight.shape
        return torch.addmm(res.reshape(-1, n_out), x.reshape(-1, n_in), weight.t()).view(res.shape)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        n_out, n_in = weight.shape
        dy2 = dy.reshape(-1, n_out)
        dx = dw = None
        if ctx.needs_input_grad[1]:
            dx = (dy2 @ weight).view(x.shape)
        if ctx.needs_input_grad[2]:
            dw = _weight_grad(dy2, x.reshape(-1, n_in), n_out).to(weight.dtype)
        return (dy if ctx.needs_input_grad[0] else None), dx, dw


def residual_linear(res: torch.Tensor, x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``res + F.linear(x, weight)`` as one GEMM (reference: the `x + self.dp1(self.attn(...))` add of
    experiments/cifar100_edgewise_gates.py:372 folded into `proj`, attention_variants.py:564)."""
    return _ResidualLinearFn.apply(res, x, weight)


class _TokenLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        n_out, n_in = weight.shape
        dy2 = dy.reshape(-1, n_out)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = (dy2 @ weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, n_in)
            rows = x2.shape[0]
            s = weight_grad_splits(rows, n_out)
            if s > 1:
                part = torch.bmm(dy2.reshape(s, rows // s, n_out).transpose(1, 2), x2.reshape(s, rows // s, n_in))
                dw = part.sum(0, dtype=torch.float32).to(weight.dtype)
            else:
                dw = dy2.t() @ x2
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0, dtype=torch.float32).to(dy.dtype)
        return dx, dw, db


class TokenLinear(nn.Linear):
    """drop-in ``nn.Linear`` (identical parameters and forward values); see the module docstring for the backward."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (x.is_cuda and x.dtype == self.weight.dtype and torch.is_grad_enabled() and not torch.is_autocast_enabled()
                and (x.requires_grad or self.weight.requires_grad) and x.numel() // max(1, x.shape[-1]) >= _MIN_ROWS):
            return _TokenLinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
