"""ViT building blocks with the reference's parameter names (mop/models/components.py).

Only `MSA` touches libmopk (plain SDPA core, reference components.py:56-66); the rest are stock
PyTorch-ROCm layers kept so that `ViT_MoP` state_dicts load unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .linear import TokenLinear, residual_linear


class DropPath(nn.Module):
    """Per-sample stochastic depth (reference components.py:14-26)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if not self.training or self.drop_prob == 0.0:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


class PatchEmbed(nn.Module):
    def __init__(self, in_ch=3, dim=256, patch=4):
        super().__init__()
        self.proj = nn.Conv2d(in_ch, dim, kernel_size=patch, stride=patch, bias=False)

    def forward(self, x):
        x = self.proj(x)
        return x.flatten(2).transpose(1, 2), (x.shape[2], x.shape[3])


class MSA(nn.Module):
    """softmax(q k^T / sqrt(dk)) v ; attention core = mopk_sdpa_* (reference components.py:43-66)."""

    def __init__(self, dim, heads=4, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        assert dim % heads == 0
        self.h, self.dk = heads, dim // heads
        self.qkv = TokenLinear(dim, dim * 3, bias=False)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = TokenLinear(dim, dim, bias=False)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, residual=None):
        """`residual` (default None = reference behaviour): returns `residual + attn(x)`, the add folded into proj's GEMM"""
        B, N, D = x.shape
        pdrop = float(self.attn_drop.p) if self.training else 0.0      # self.attn_drop(A) (:62), inside the kernels
        qkv = self.qkv(x).view(B, N, 3, self.h, self.dk)
        y = ops.sdpa_core(qkv, dropout_p=pdrop)                       # packed q | k | v: one gradient tensor back
        if residual is None:
            return self.proj_drop(self.proj(y))
        if y.is_cuda and y.dtype == residual.dtype == self.proj.weight.dtype and not torch.is_autocast_enabled() \
                and not (self.training and self.proj_drop.p > 0):
            return residual_linear(residual, y, self.proj.weight)
        return residual + self.proj_drop(self.proj(y))


class MLP(nn.Module):
    def __init__(self, dim, mlp_ratio=4.0, drop=0.0):
        super().__init__()
        hid = int(dim * mlp_ratio)
        self.fc1 = TokenLinear(dim, hid, bias=False)
        self.fc2 = TokenLinear(hid, dim, bias=False)
        self.act = nn.GELU(approximate="tanh")
        self.drop = nn.Dropout(drop)

    def forward(self, x, residual=None):
        """`residual` (default None = reference behaviour): returns `residual + mlp(x)`, the add folded into fc2's GEMM"""
        hid = self.act(self.fc1(x))
        if residual is None:
            return self.drop(self.fc2(hid))
        if hid.is_cuda and hid.dtype == residual.dtype == self.fc2.weight.dtype and not torch.is_autocast_enabled() \
                and not (self.training and self.drop.p > 0):
            return residual_linear(residual, hid, self.fc2.weight)
        return residual + self.drop(self.fc2(hid))


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4.0, drop=0.0, attn_drop=0.0, drop_path=0.0):
        super().__init__()
        self.ln1 = nn.LayerNorm(dim)
        self.attn = MSA(dim, heads, attn_drop, drop)
        self.dp1 = DropPath(drop_path)
        self.ln2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, mlp_ratio, drop)
        self.dp2 = DropPath(drop_path)

    def forward(self, x):
        idle = lambda dp: not (self.training and dp.drop_prob > 0.0)
        if not (x.is_cuda and idle(self.dp1) and idle(self.dp2) and ops.layernorm_supported(x, self.ln1.weight)):
            x = x + self.dp1(self.attn(self.ln1(x)))
            return x + self.dp2(self.mlp(self.ln2(x)))
        # LayerNorm prologue + residual epilogue (mopk_layernorm_*, the add in the projection GEMM), as in BlockEdgewise
        odt = torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else x.dtype
        xr, h = ops.layernorm_residual(x, self.ln1.weight, self.ln1.bias, self.ln1.eps, odt)
        x = self.attn(h, residual=xr)
        xr, h = ops.layernorm_residual(x, self.ln2.weight, self.ln2.bias, self.ln2.eps, odt)
        return self.mlp(h, residual=xr)


class ViTEncoder(nn.Module):
    def __init__(self, dim=256, depth=6, heads=4, mlp_ratio=4.0, drop=0.0, drop_path=0.1, patch=4, num_tokens=64):
        super().__init__()
        self.patch = PatchEmbed(dim=dim, patch=patch)
        self.pos = nn.Parameter(torch.zeros(1, num_tokens, dim))
        rates = torch.linspace(0, drop_path, depth).tolist()
        self.blocks = nn.ModuleList(Block(dim, heads, mlp_ratio, drop, 0.0, rates[i]) for i in range(depth))
        self.ln_f = nn.LayerNorm(dim)
        nn.init.normal_(self.pos, mean=0.0, std=0.02)

    def forward(self, x):
        tok, grid = self.patch(x)
        tok = tok + self.pos
        for blk in self.blocks:
            tok = blk(tok)
        return self.ln_f(tok), grid


class ViewsLinear(nn.Module):
    def __init__(self, dim, n_views=5):
        super().__init__()
        self.proj = TokenLinear(dim, n_views, bias=False)
        self.n_views = n_views

    def forward(self, tok, grid):
        B = tok.shape[0]
        return self.proj(tok).transpose(1, 2).reshape(B, self.n_views, grid[0], grid[1])


class Kernels3(nn.Module):
    def __init__(self, in_ch, n_kernels=3):
        super().__init__()
        self.k = nn.Sequential(nn.Conv2d(in_ch, 16, kernel_size=3, padding=1, bias=False), nn.SiLU(inplace=True),
                               nn.Conv2d(16, n_kernels, kernel_size=1, bias=False))

    def forward(self, maps):
        return self.k(maps)


class FuseExcInh(nn.Module):
    def __init__(self, in_ch):
        super().__init__()
        hid = max(8, in_ch)
        self.fuse = nn.Sequential(nn.Conv2d(in_ch, hid, kernel_size=1, bias=False), nn.SiLU(inplace=True),
                                  nn.Conv2d(hid, 2, kernel_size=1, bias=True))
        self.alpha_pos = nn.Parameter(torch.tensor(0.8))
        self.alpha_neg = nn.Parameter(torch.tensor(0.8))

    def forward(self, x):
        g = torch.sigmoid(self.fuse(x))
        return g[:, :1], g[:, 1:], F.softplus(self.alpha_pos), F.softplus(self.alpha_neg)
