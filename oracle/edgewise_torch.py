"""torch-CPU restatement of EdgewiseMSA (shared QKV, low-rank gate head)  --  TEST INFRASTRUCTURE ONLY.

Same algebra as oracle/edgewise.py (reference `mop/models/attention_variants.py:453-564`), written with torch ops so that it
runs multi-threaded on the host cores and gets its backward from autograd: this is the CPU baseline `bench.py` times on the GPU
box (`torch.set_num_threads(all cores)`, B = 16, fwd + bwd), and `tests/test_oracle_golden.py` pins it
against the reference fixtures.  It is the *means-only* formulation: the gate head consumes only row / column means of the
feature stack (:323-324), so the (B*H, 2V+2, N, N) stack of :534 is never built (the reference spends half its CPU time there).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

EPS_CHAIN = 1e-6  # attention_variants.py:516


def edgewise_layer(x: torch.Tensor, p: Dict[str, torch.Tensor], heads: int, n_views: int, beta_not: float = 0.5) -> torch.Tensor:
    """x: (B, N, D); p: reference state_dict entries (shared-QKV, low-rank head) as tensors.  Returns proj(y): (B, N, D)."""
    B, N, D = x.shape
    H, V = heads, max(2, n_views)                                           # :362
    dk = D // H
    qkv = (x @ p["qkv.weight"].t()).view(B, N, 3, H, dk).permute(2, 0, 3, 1, 4)     # :458-460  (3,B,H,N,dk)
    q, k, v = qkv[0], qkv[1], qkv[2]
    qs, ks, vs = p["q_scale"], p["k_scale"], p["v_scale"]                   # (V,H,1,dk)  :461-463
    scale = 1.0 / math.sqrt(dk)
    S = torch.stack([((q * qs[i]) @ (k * ks[i]).transpose(-1, -2)) * scale for i in range(V)])      # :500-503 (V,B,H,N,N)
    A = torch.softmax(S, dim=-1)                                            # :507
    Cf = A[0]
    for i in range(1, V):                                                   # :508-512
        Cf = Cf @ A[i]
    Cb = A[V - 1]
    for i in range(V - 2, -1, -1):                                          # :513-515
        Cb = Cb @ A[i]
    Cr, Cl = torch.log(Cf + EPS_CHAIN), torch.log(Cb + EPS_CHAIN)           # :520-521
    rS, cS = S.mean(-1), S.mean(-2)                                         # (V,B,H,N): row means of S_v; row means of S_v^T = col means
    row_feat = torch.cat([rS.permute(1, 2, 0, 3), cS.permute(1, 2, 0, 3), Cr.mean(-1)[:, :, None], Cl.mean(-1)[:, :, None]], 2)   # (B,H,C,N)
    col_feat = torch.cat([cS.permute(1, 2, 0, 3), rS.permute(1, 2, 0, 3), Cr.mean(-2)[:, :, None], Cl.mean(-2)[:, :, None]], 2)
    Wr, Wc = p["edge_head.row_proj.weight"].squeeze(-1), p["edge_head.col_proj.weight"].squeeze(-1)     # (4r, C)
    a = torch.einsum("oc,bhcn->bhon", Wr, row_feat) + p["edge_head.row_proj.bias"][None, None, :, None]    # :325
    b = torch.einsum("oc,bhcn->bhon", Wc, col_feat) + p["edge_head.col_proj.bias"][None, None, :, None]    # :326
    r = Wr.shape[0] // 4
    G = torch.sigmoid(torch.einsum("bhgkn,bhgkm->bhgnm", a.view(B, H, 4, r, N), b.view(B, H, 4, r, N)))    # :330-331
    S0, Ssum = S[0], S.sum(0)
    lse = torch.logsumexp(S, dim=0)                                         # :541
    O = Ssum - S0
    nb = beta_not / max(1, V - 1)
    Smix = S0 + G[:, :, 0] * O + G[:, :, 1] * (lse - S0) - G[:, :, 2] * (nb * O) + G[:, :, 3] * Cr      # :543-547
    P = torch.softmax(Smix, dim=-1)                                         # :551
    y = P @ (v * vs[0])                                                     # :554
    t = v * vs[V - 1]                                                       # :556-560 value transport
    for i in range(V - 1, -1, -1):
        t = A[i] @ t
    y = y + torch.sigmoid(p["chain_value_logit"]) * t                       # :561-562
    return y.transpose(1, 2).reshape(B, N, D) @ p["proj.weight"].t()        # :563-564
