#!/usr/bin/env python3
"""dev tool (GPU box): timeline of one chain part of the fused forward (MOPK_STAMPS3 build): wave 0 and its SIMD partner wave 4."""
import os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mop_amd import ops
layer = bench.build_layer(torch.bfloat16)
x = torch.randn(256, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
for _ in range(2):
    y = layer(x)
torch.cuda.synchronize()
st = struct.unpack("32Q", ops.LAST_PATH["_fwd_ws"][:256].cpu().numpy().tobytes())
names = ["start", "loads issued", "export mfma", "S mfma", "k=1", "k=4", "k=7", "k=10", "k=12", "gemm done", "packed", "barrier out"]
t0 = min(st[0], st[16])
for wv, base in ((0, 0), (4, 16)):
    print("wave", wv, " ".join(f"{names[i]}:{st[base+i]-t0}" for i in range(12)))
