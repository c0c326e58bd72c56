"""dev tool (GPU box): fused SDPA forward + backward at T=3000, B=8, H=6, dk=64 (bf16) for rocprofv3 --kernel-trace --stats"""
import sys, torch
sys.path.insert(0, ".")
from mop_amd import ops
B, T, H, dk = 8, 3000, 6, 64
q, k, v = (torch.randn(B, T, H, dk, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
w = torch.randn(B, T, H * dk, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    ops.sdpa_core(q, k, v).backward(w)
torch.cuda.synchronize()
