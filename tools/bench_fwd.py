#!/usr/bin/env python3
"""Forward-only timing of the EdgewiseMSA core at the NS shape (dev tool, GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mop_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layer = bench.build_layer(torch.bfloat16)
x = torch.randn(B, 197, 384, device="cuda", dtype=torch.bfloat16)
with torch.no_grad():
    for _ in range(3):
        layer(x)
    ops.enable_timing(True)
    for _ in range(10):
        layer(x)
torch.cuda.synchronize()
t = ops.timing_results()["edgewise_fwd"]
ms = sorted(t)[len(t) // 2]
fl = B * bench.CORE_FLOP_FWD
print(f"fused fwd core: median {ms:.3f} ms  min {min(t):.3f} ms  -> {B/ms*1e3:.0f} img/s  {fl/ms/1e9:.1f} TFLOP/s "
      f"({fl/ms/1e9/2500*100:.2f}% of 2.5 PF)  path={ops.LAST_PATH}")
