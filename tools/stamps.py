#!/usr/bin/env python3
"""dev tool (GPU box): per-phase cycle shares of the fused backward from a MOPK_STAMPS build."""
import os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
# stamp offset: find via the known layout -- last 512 bytes before stride end of WG 0
import ctypes as C
from mop_amd import _lib as L
a = L.EdgewiseArgs(); a.B, a.H, a.N, a.dk, a.V, a.r = B, 6, 197, 64, 5, 4; a.path = L.PATH_FUSED; a.precision = L.PREC_BF16
a.io_dtype = L.MOPK_BF16; a.save_for_backward = 1
tot = L.lib().mopk_edgewise_workspace_bytes(C.byref(a))
stride = (tot - 256) // min(B * 6, 256)
raw = ws[stride - 512: stride].cpu().numpy().tobytes()
st = struct.unpack("64Q", raw)
names = ["P0 stage", "P1/2 fwd chains", "P3 gates", "P4 mix", "P5 delta", "P6 mix bwd", "P7 gate grads", "P8 dv", "P9 <-D chain",
         "P10a dA (v=V-1)", "P10b dS", "P10c dQe", "P10d dK", "P10e rest of P10", "P11 out"]
vals = [s for s in st if s]
tot_c = vals[-1] - vals[0]
for i in range(len(vals) - 1):
    print(f"{(names[i] if i < len(names) else str(i)):22s} {vals[i+1]-vals[i]:10d} cyc  {100.0*(vals[i+1]-vals[i])/tot_c:5.1f}%")
print("total", tot_c)
