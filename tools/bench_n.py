#!/usr/bin/env python3
"""dev tool (GPU box): fused Edgewise core time vs token count (NT buckets) at B=256, H=6, dk=64, V=5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mop_amd import ops
from mop_amd.nn import EdgewiseMSA
torch.manual_seed(0)
m = EdgewiseMSA(384, 6, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4, gate_init="mix5").cuda().to(torch.bfloat16)
for N in (32, 64, 128, 197, 224):
    x = torch.randn(256, N, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    for _ in range(3):
        m(x).sum().backward()
    ops.enable_timing(True)
    for _ in range(5):
        m(x).sum().backward()
    torch.cuda.synchronize()
    t = {k: sum(v) / len(v) for k, v in ops.timing_results().items()}
    ops.enable_timing(False)
    nt = (N + 31) // 32
    print(f"N={N:4d} NT={nt} fwd {t['edgewise_fwd']:.3f} ms  bwd {t['edgewise_bwd']:.3f} ms   per NT^3: fwd {t['edgewise_fwd']/nt**3*1e3:.2f} bwd {t['edgewise_bwd']/nt**3*1e3:.2f} us", flush=True)
