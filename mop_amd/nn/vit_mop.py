"""ViT_MoP with the reference's constructor, forward and state_dict (mop/models/vit_mop.py:15-140).

The encoder's attention runs on libmopk (plain SDPA core); the excitatory/inhibitory token gate
`1 + a+ sigma(G+) - a- sigma(G-)` (:98-118) is a <1 % FLOP epilogue kept in PyTorch-ROCm.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .components import FuseExcInh, Kernels3, ViewsLinear, ViTEncoder


class ViT_MoP(nn.Module):
    def __init__(self, dim=256, depth=6, heads=4, mlp_ratio=4.0, n_classes=10, n_views=5, n_kernels=3,
                 drop_path=0.1, patch=4, img_size=32, use_moe: bool = False, moe_experts: int = 4):
        super().__init__()
        assert dim % heads == 0, f"dim {dim} not divisible by heads {heads}"
        if use_moe:
            raise NotImplementedError("ViT_MoP(use_moe=True): the dense-MoE MLP is outside the hot path (SURVEY.md section 2 row 4)")
        self.enc = ViTEncoder(dim=dim, depth=depth, heads=heads, mlp_ratio=mlp_ratio, drop_path=drop_path,
                              patch=patch, num_tokens=(img_size // patch) ** 2)
        self.views = ViewsLinear(dim, n_views=n_views)
        self.kerns = Kernels3(in_ch=n_views, n_kernels=n_kernels)
        self.fuse = FuseExcInh(in_ch=n_views + n_kernels)
        self.cls = nn.Linear(dim, n_classes, bias=False)
        self.n_views, self.n_kernels = n_views, n_kernels

    def _gate(self, tok, grid):
        V = self.views(tok, grid)
        K = self.kerns(V)
        g_pos, g_neg, a_pos, a_neg = self.fuse(torch.cat([V, K], dim=1))
        return 1 + a_pos * g_pos - a_neg * g_neg, V, K

    def forward(self, x):
        tok, grid = self.enc(x)
        B, N, _ = tok.shape
        gate, _, _ = self._gate(tok, grid)
        tok = tok * gate.reshape(B, N, 1)
        return self.cls(tok.mean(dim=1))

    def get_gate_maps(self, x):
        with torch.no_grad():
            tok, grid = self.enc(x)
            return self._gate(tok, grid)
