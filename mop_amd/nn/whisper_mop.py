"""Encoder side of Whisper-MoP with the reference's parameter names (mop/models/whisper_mop.py).

`MultiheadSelfAttention` (reference :137-177) runs its attention core in libmopk (plain SDPA with
the causal flag and the additive `attn_bias`); the mel-map gate `MoP2D`
(:91-124, row a17) and the MLP are stock PyTorch-ROCm layers.  The decoder and its
cross-attention (:180-228, :291-317) are outside the hot path and are not mirrored.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .linear import TokenLinear


@dataclass
class WhisperConfig:
    """reference :18-37 (same field names and defaults)."""
    n_mels: int = 80
    n_audio_ctx: int = 1500
    vocab_size: int = 51865
    n_text_ctx: int = 448
    n_embd: int = 1024
    n_head: int = 16
    n_layer_enc: int = 12
    n_layer_dec: int = 12
    dropout: float = 0.0
    bias: bool = False
    use_abs_pos_emb: bool = True
    n_views: int = 5
    n_kernels: int = 3
    kernel_size: int = 5


class ViewsConv2D(nn.Module):
    def __init__(self, n_views: int):
        super().__init__()
        self.conv = nn.Conv2d(1, n_views, kernel_size=1, bias=False)

    def forward(self, mel2d):            # (B,1,T,F) -> (B,V,T,F)
        return self.conv(mel2d)


class Kernels2D(nn.Module):
    def __init__(self, in_ch: int, n_kernels: int, kernel_size: int):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, n_kernels, kernel_size, padding=kernel_size // 2, bias=False)

    def forward(self, x):                # (B,V,T,F) -> (B,K,T,F)
        return self.conv(x)


class FuseExcInh2D(nn.Module):
    def __init__(self, in_ch: int):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, 2, kernel_size=1, bias=False)
        self.alpha = nn.Parameter(torch.ones(2))          # (alpha_pos, alpha_neg), no squashing here (:84-88)

    def forward(self, x):
        g = self.conv(x)
        return g[:, 0:1], g[:, 1:2], self.alpha[0], self.alpha[1]


class MoP2D(nn.Module):
    """per-time-step gate 1 + a+ mean_F(g+) - a- mean_F(g-) from the raw mel map (reference :91-124)."""

    def __init__(self, n_views: int, n_kernels: int, kernel_size: int):
        super().__init__()
        self.views = ViewsConv2D(n_views)
        self.kernels = Kernels2D(n_views, n_kernels, kernel_size)
        self.fuse = FuseExcInh2D(n_views + n_kernels)

    def forward(self, mel2d) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        V = self.views(mel2d)
        K = self.kernels(V)
        g_pos, g_neg, a_pos, a_neg = self.fuse(torch.cat([V, K], dim=1))
        gate_t = 1 + a_pos * g_pos.mean(dim=3) - a_neg * g_neg.mean(dim=3)      # (B,1,T)
        return gate_t.transpose(1, 2), V, K                                     # (B,T,1)


class MultiheadSelfAttention(nn.Module):
    """separate q/k/v/o Linears; softmax(q k^T / sqrt(dh) [causal] [+ attn_bias]) v in libmopk (reference :137-177)."""

    def __init__(self, dim: int, n_head: int, dropout: float, bias: bool, causal: bool):
        super().__init__()
        assert dim % n_head == 0
        self.dim, self.n_head, self.head_dim, self.causal = dim, n_head, dim // n_head, causal
        self.scale = self.head_dim ** -0.5
        self.q_proj = TokenLinear(dim, dim, bias=bias)
        self.k_proj = TokenLinear(dim, dim, bias=bias)
        self.v_proj = TokenLinear(dim, dim, bias=bias)
        self.o_proj = TokenLinear(dim, dim, bias=bias)
        self.attn_drop = nn.Dropout(dropout)
        self.resid_drop = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor, attn_bias: Optional[torch.Tensor] = None):
        pdrop = float(self.attn_drop.p) if self.training else 0.0      # dropout on the probabilities (:172), inside the kernels
        B, T, D = x.shape
        H, Dh = self.n_head, self.head_dim
        q, k, v = (p(x).view(B, T, H, Dh) for p in (self.q_proj, self.k_proj, self.v_proj))
        y = ops.sdpa_core(q, k, v, bias=attn_bias, causal=self.causal, dropout_p=pdrop)
        return self.resid_drop(self.o_proj(y))


class MLP(nn.Module):
    def __init__(self, dim: int, dropout: float, bias: bool):
        super().__init__()
        self.fc = TokenLinear(dim, 4 * dim, bias=bias)
        self.proj = TokenLinear(4 * dim, dim, bias=bias)
        self.drop = nn.Dropout(dropout)

    def forward(self, x):
        return self.drop(self.proj(F.gelu(self.fc(x), approximate="tanh")))


class EncoderBlock(nn.Module):
    """x + SA(ln1 x); x * gate_t(mel); x + MLP(ln2 x)   (reference :250-275)."""

    def __init__(self, cfg: WhisperConfig):
        super().__init__()
        D = cfg.n_embd
        self.ln1 = nn.LayerNorm(D)
        self.attn = MultiheadSelfAttention(D, cfg.n_head, cfg.dropout, cfg.bias, causal=False)
        self.ln2 = nn.LayerNorm(D)
        self.mlp = MLP(D, cfg.dropout, cfg.bias)
        self.mop = MoP2D(cfg.n_views, cfg.n_kernels, cfg.kernel_size)

    def forward(self, x, mel2d):
        x = x + self.attn(self.ln1(x))
        gate_t, _, _ = self.mop(mel2d)
        x = x * gate_t
        x = x + self.mlp(self.ln2(x))
        return x, gate_t.squeeze(-1)
