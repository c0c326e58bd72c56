#!/usr/bin/env python3
"""dev tool (GPU box): a few fused EdgewiseMSA fwd+bwd steps at the bench shape, for rocprofv3 runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mop_amd.nn import EdgewiseMSA
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
m = EdgewiseMSA(384, 6, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4, gate_init="mix5").cuda().to(torch.bfloat16)
x = torch.randn(B, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
for _ in range(steps):
    m(x).sum().backward()
torch.cuda.synchronize()
