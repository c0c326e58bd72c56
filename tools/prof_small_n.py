"""dev tool (GPU box): EdgewiseMSA layer at N = 65 (the reference's CIFAR sequence length), B = 256, for rocprofv3 --kernel-trace --stats"""
import sys, torch
sys.path.insert(0, ".")
from mop_amd.nn import EdgewiseMSA
torch.manual_seed(0)
m = EdgewiseMSA(384, 6, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4).cuda().to(torch.bfloat16)
x = torch.randn(256, 65, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
w = torch.randn_like(x)
for _ in range(6):
    m(x).backward(w)
torch.cuda.synchronize()
