"""dev tool (GPU box): cProfile of the host side of sdpa_core forward + backward"""
import cProfile, pstats, sys, torch
sys.path.insert(0, ".")
from mop_amd import ops
B, T, H, dk = 4, 65, 6, 64
q, k, v = (torch.randn(B, T, H, dk, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
w = torch.randn(B, T, H * dk, device="cuda", dtype=torch.bfloat16)
for _ in range(20):
    ops.sdpa_core(q, k, v).backward(w)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    ops.sdpa_core(q, k, v).backward(w)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
