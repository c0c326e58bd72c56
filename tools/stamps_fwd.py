#!/usr/bin/env python3
"""dev tool (GPU box): per-phase cycle shares of the fused forward from a MOPK_STAMPS build (workgroup 0)."""
import os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mop_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layer = bench.build_layer(torch.bfloat16)
x = torch.randn(B, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
infer = len(sys.argv) > 2 and sys.argv[2] == "infer"     # inference forward (nothing exported) instead of the training forward
for _ in range(2):
    if infer:
        with torch.no_grad():
            y = layer(x)
    else:
        y = layer(x)
torch.cuda.synchronize()
st = struct.unpack("32Q", ops.LAST_PATH["_fwd_ws"][:256].cpu().numpy().tobytes())
names = ["P0 stage + means", "row constants", "chain <-", "chain -> (+exports, V^T, y_chain)", "gate vectors", "mix", "softmax + P V0"]
vals = list(st[:len(names) + 1])            # the kernel writes one stamp per phase boundary; the rest of the buffer is uninitialised
tot = vals[-1] - vals[0]
for i in range(len(vals) - 1):
    print(f"{(names[i] if i < len(names) else str(i)):36s} {vals[i+1]-vals[i]:10d} cyc  {100.0*(vals[i+1]-vals[i])/tot:5.1f}%")
print("total", tot)
