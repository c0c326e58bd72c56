#!/usr/bin/env python3
"""dev tool (GPU box): per-section cycle shares of one launch of the split fused backward from a MOPK_STAMPS build
(MOPK_STAMP_PH=A|B|C selects the launch at build time).  Stamps are the first 512 bytes of the backward workspace."""
import os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MOPK_STAMPS", "1")
import torch, bench
from mop_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layer = bench.build_layer(torch.bfloat16)
x = torch.randn(B, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
dy = torch.randn_like(x)
for _ in range(2):
    layer(x).backward(dy)
torch.cuda.synchronize()
ws = ops.LAST_PATH["_bwd_ws"]
st = struct.unpack("64Q", ws[:512].cpu().numpy().tobytes())
names = {"A": ["P0 stage", "means", "P3 gates", "P4 mix state", "-", "P6 mix bwd", "P7 gate grads", "P8 dv", "end"],
         "B": ["stage", "init D"] + [x for m in range(4) for x in ("export + q frags", "barrier", "A image", "barrier", "GEMM")] + ["last export", "item 2 ..."],
         "C": ["P0 stage", "GEMM1 (v=V-1)", "GEMM2", "rowdot", "dS pass", "dK", "rest of the views", "P11 out"]}[os.environ.get("MOPK_STAMP_PH", "C")]
vals = [s for s in st if s]
tot_c = vals[-1] - vals[0]
for i in range(len(vals) - 1):
    print(f"{(names[i] if i < len(names) else str(i)):22s} {vals[i+1]-vals[i]:10d} cyc  {100.0*(vals[i+1]-vals[i])/tot_c:5.1f}%")
print("total", tot_c)
