"""The caller of the Edgewise hot path: `BlockEdgewise` / `ViTEdgewise` with the reference's parameter names
(experiments/cifar100_edgewise_gates.py:326-452; the same block shape as experiments/voc_localization_vit.py:217-241).

Everything here is stock PyTorch-ROCm except `EdgewiseMSA` (libmopk).  SURVEY.md 8b "what calls it", 8f rank 4.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .attention_variants import EdgewiseMSA
from .components import MLP, DropPath, PatchEmbed


class BlockEdgewise(nn.Module):
    """x + dp(attn(ln1 x)); x + dp(mlp(ln2 x))   (reference :371-374)."""

    def __init__(self, dim: int, heads: int, mlp_ratio: float = 4.0, drop: float = 0.0, attn_drop: float = 0.0,
                 drop_path: float = 0.0, beta_not: float = 0.5, use_k3: bool = False, n_views: int = 2,
                 share_qkv: bool = False, gate_mode: str = "dense", gate_rank: int = 4, gate_init: str = "neutral",
                 use_lens_bank_qk: bool = False, lens_qk_kernel_size: int = 3,
                 lens_qk_dilations: Optional[Tuple[int, ...]] = None, lens_qk_causal: bool = False):
        super().__init__()
        self.ln1 = nn.LayerNorm(dim)
        self.attn = EdgewiseMSA(dim, heads, attn_drop, drop, beta_not=beta_not, use_k3=use_k3, n_views=n_views,
                                share_qkv=share_qkv, gate_mode=gate_mode, gate_rank=gate_rank, gate_init=gate_init,
                                use_lens_bank_qk=use_lens_bank_qk, lens_qk_kernel_size=lens_qk_kernel_size,
                                lens_qk_dilations=lens_qk_dilations, lens_qk_causal=lens_qk_causal)
        self.dp1 = DropPath(drop_path)
        self.ln2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, mlp_ratio, drop)
        self.dp2 = DropPath(drop_path)

    def _fused_edges(self, x: torch.Tensor) -> bool:
        """LayerNorm prologue + residual epilogue in libmopk / the GEMM epilogue: on the GPU, when the
        stochastic-depth branches are identities (eval, or drop_path == 0 as in the benchmarked configs)"""
        idle = lambda dp: not (self.training and dp.drop_prob > 0.0)
        return x.is_cuda and idle(self.dp1) and idle(self.dp2) and ops.layernorm_supported(x, self.ln1.weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self._fused_edges(x):
            x = x + self.dp1(self.attn(self.ln1(x)))
            return x + self.dp2(self.mlp(self.ln2(x)))
        odt = torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else x.dtype
        xr, h = ops.layernorm_residual(x, self.ln1.weight, self.ln1.bias, self.ln1.eps, odt)      # one pass: LN + cast
        x = self.attn(h, residual=xr)                                                              # proj GEMM adds xr
        xr, h = ops.layernorm_residual(x, self.ln2.weight, self.ln2.bias, self.ln2.eps, odt)
        return self.mlp(h, residual=xr)


class ViTEdgewise(nn.Module):
    """patch embedding + learned positions + `depth` BlockEdgewise + LayerNorm + mean-pool head (reference :377-452)."""

    def __init__(self, dim: int = 256, depth: int = 8, heads: int = 4, n_classes: int = 100, mlp_ratio: float = 4.0,
                 drop: float = 0.0, drop_path: float = 0.1, patch: int = 4, num_tokens: int = 64, beta_not: float = 0.5,
                 use_k3: bool = False, n_views: int = 2, share_qkv: bool = False, gate_mode: str = "dense",
                 gate_rank: int = 4, gate_init: str = "neutral", use_lens_bank_qk: bool = False,
                 lens_qk_kernel_size: int = 3, lens_qk_dilations: Optional[Tuple[int, ...]] = None,
                 lens_qk_causal: bool = False):
        super().__init__()
        self.patch = PatchEmbed(in_ch=3, dim=dim, patch=patch)
        self.pos = nn.Parameter(torch.zeros(1, num_tokens, dim))
        rates = torch.linspace(0, drop_path, depth).tolist()
        self.blocks = nn.ModuleList(
            BlockEdgewise(dim, heads, mlp_ratio, drop, 0.0, rates[i], beta_not=beta_not, use_k3=use_k3, n_views=n_views,
                          share_qkv=share_qkv, gate_mode=gate_mode, gate_rank=gate_rank, gate_init=gate_init,
                          use_lens_bank_qk=use_lens_bank_qk, lens_qk_kernel_size=lens_qk_kernel_size,
                          lens_qk_dilations=lens_qk_dilations, lens_qk_causal=lens_qk_causal)
            for i in range(depth))
        self.ln_f = nn.LayerNorm(dim)
        self.head = nn.Linear(dim, n_classes, bias=False)
        nn.init.normal_(self.pos, mean=0.0, std=0.02)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        tok, _ = self.patch(x)
        tok = tok + self.pos
        for blk in self.blocks:
            tok = blk(tok)
        return self.head(self.ln_f(tok).mean(dim=1))
