#!/usr/bin/env python3
"""dev tool (GPU box): fused vs generic core gradients wrt packed qkv, per component, on a seeded random case: B N H dk V r"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mop_amd import ops, _lib as L
B, N, H, dk, V, r = map(int, sys.argv[1:7])
torch.manual_seed(0)
dev = "cuda"
qkv0 = torch.randn(B, N, 1, 3, H, dk, device=dev).to(torch.bfloat16)
sqk = (1 + 0.1 * torch.randn(V, H, dk, device=dev)) / dk ** 0.5
vs0 = 1 + 0.1 * torch.randn(H, dk, device=dev); vsL = 1 + 0.1 * torch.randn(H, dk, device=dev)
C = 2 * V + 2
Wr = 0.3 * torch.randn(4 * r, C, device=dev); Wc = 0.3 * torch.randn(4 * r, C, device=dev)
br = 0.1 * torch.randn(4 * r, device=dev); bc = 0.1 * torch.randn(4 * r, device=dev)
lg = torch.tensor(-0.5, device=dev)
dy = torch.randn(B, N, H * dk, device=dev).to(torch.bfloat16)
res = {}
for path in (L.PATH_GENERIC, L.PATH_FUSED):
    q = qkv0.clone().requires_grad_(True)
    ps = [t.clone().requires_grad_(True) for t in (sqk, vs0, vsL, Wr, br, Wc, bc, lg)]
    y = ops.edgewise_lowrank_core(q, *ps, 0.5, V, precision=L.PREC_BF16, path=path)
    y.backward(dy)
    res[path] = (y.float(), q.grad.float(), [p.grad for p in ps])
yg, gg, pg = res[L.PATH_GENERIC]; yf, gf, pf = res[L.PATH_FUSED]
print("y", (yg - yf).abs().max().item(), yg.abs().max().item())
for i, nm in enumerate("qkv"):
    a, b = gg[:, :, 0, i], gf[:, :, 0, i]
    e = (a - b).abs()
    idx = torch.nonzero(e == e.max())[0].tolist()
    print(f"d{nm}: maxerr {e.max().item():.3e} ref max {a.abs().max().item():.3e} at (b,n,h,d)={idx}; per-n max err:", [f"{v:.1e}" for v in e.amax(dim=(0, 2, 3))[:N].tolist()][:64])
for nm, a, b in zip("sqk vs0 vsL Wr br Wc bc lg".split(), pg, pf):
    print(nm, f"{(a - b).abs().max().item():.3e} / {a.abs().max().item():.3e}")
