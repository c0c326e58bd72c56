"""dev tool (GPU box): a few training steps of EdgewiseMSA (low-rank head) on the GENERIC path at the bench shape, for
rocprofv3 --kernel-trace --stats"""
import sys, torch
sys.path.insert(0, ".")
from mop_amd import ops
from mop_amd.nn import EdgewiseMSA
torch.manual_seed(0)
ops.set_path("generic")
m = EdgewiseMSA(384, 6, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4).cuda().to(torch.bfloat16)
x = torch.randn(256, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
w = torch.randn_like(x)
for _ in range(4):
    m(x).backward(w)
torch.cuda.synchronize()
