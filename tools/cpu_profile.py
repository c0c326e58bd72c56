"""dev tool (GPU box): cProfile of the host side of one EdgewiseMSA layer's forward + backward (small shape: the GPU is idle, the host
path is what is measured)"""
import cProfile, pstats, sys, torch
sys.path.insert(0, ".")
from mop_amd.nn import EdgewiseMSA
m = EdgewiseMSA(256, 4, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4).cuda().to(torch.bfloat16)
x = torch.randn(8, 64, 256, device="cuda", dtype=torch.bfloat16, requires_grad=True)
w = torch.randn_like(x)
for _ in range(20):
    m.zero_grad(set_to_none=True)
    m(x).backward(w)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    m.zero_grad(set_to_none=True)
    m(x).backward(w)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
