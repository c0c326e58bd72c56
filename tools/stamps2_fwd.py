import os, sys, struct
sys.path.insert(0, os.getcwd())
import torch, bench
from mop_amd import ops
layer = bench.build_layer(torch.bfloat16)
x = torch.randn(256, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
for _ in range(2):
    y = layer(x)
torch.cuda.synchronize()
st = struct.unpack("32Q", ops.LAST_PATH["_fwd_ws"][:256].cpu().numpy().tobytes())
print([st[i+1]-st[i] for i in range(20)])
