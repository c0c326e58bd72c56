"""Quartet dual-path causal attention with the reference surface (mop/models/quartet_attn_patch.py:19-127).

The five/three Linear projections stay in PyTorch (hipBLASLt); scores, row z-normalisation, product
mix, causal + additive mask, softmax and AV are one libmopk call (mopk_quartet_*).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .linear import TokenLinear


@dataclass
class TransformerConfig:
    n_layer: int = 6
    n_head: int = 8
    n_embd: int = 512
    dropout: float = 0.1
    block_size: int = 512
    bias: bool = False
    use_quartet: bool = True
    quartet_scale: float = 1.0
    quartet_gate_init: float = -5.0
    score_norm_eps: float = 1e-5
    use_abs_pos_emb: bool = True


class CausalSelfAttention(nn.Module):
    def __init__(self, config: TransformerConfig):
        super().__init__()
        assert config.n_embd % config.n_head == 0
        self.config = config
        self.n_head = config.n_head
        self.head_dim = config.n_embd // config.n_head
        self.scale = 1.0 / math.sqrt(self.head_dim)
        C = config.n_embd
        self.q_proj = TokenLinear(C, C, bias=config.bias)
        self.k_proj = TokenLinear(C, C, bias=config.bias)
        self.v_proj = TokenLinear(C, C, bias=config.bias)
        self.o_proj = TokenLinear(C, C, bias=config.bias)
        if config.use_quartet:
            self.q2_proj = TokenLinear(C, C, bias=config.bias)
            self.k2_proj = TokenLinear(C, C, bias=config.bias)
            self.mixture = nn.Parameter(torch.tensor([config.quartet_gate_init], dtype=torch.float32))
            self.quartet_scale = nn.Parameter(torch.tensor([config.quartet_scale], dtype=torch.float32))
        else:
            self.q2_proj = self.k2_proj = None
            self.register_parameter("mixture", None)
            self.register_parameter("quartet_scale", None)
        self.attn_drop = nn.Dropout(config.dropout)
        self.resid_drop = nn.Dropout(config.dropout)
        # kept for state/buffer parity with the reference (non-persistent); the kernel masks j > i itself
        self.register_buffer("causal_mask", torch.tril(torch.ones(config.block_size, config.block_size))
                             .view(1, 1, config.block_size, config.block_size), persistent=False)

    def forward(self, x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, need_weights: bool = False):
        B, T, C = x.shape
        H, Dh = self.n_head, self.head_dim
        pdrop = float(self.attn_drop.p) if self.training else 0.0      # attn_dropout on the probabilities (:119), inside the kernels
        if T > self.config.block_size:
            raise ValueError("Sequence length > block size")
        q = self.q_proj(x).view(B, T, H, Dh)
        k = self.k_proj(x).view(B, T, H, Dh)
        v = self.v_proj(x).view(B, T, H, Dh)
        uq = bool(self.config.use_quartet)
        q2 = self.q2_proj(x).view(B, T, H, Dh) if uq else None
        k2 = self.k2_proj(x).view(B, T, H, Dh) if uq else None
        out = ops.quartet_core(q, k, v, q2, k2, self.mixture, self.quartet_scale, attention_mask,
                               self.config.score_norm_eps, uq, need_weights, dropout_p=pdrop)
        y, attn = out if need_weights else (out, None)
        y = self.resid_drop(self.o_proj(y))
        return (y, attn) if need_weights else y
