"""dev tool (GPU box): the low-rank + S-lens-bank EdgewiseMSA layer at the bench shape on the fused route: kernel table of one training
step (torch profiler), to see what the closed-form lens features (ops._lens_means_fwd / _bwd, torch ops) cost beside the fused kernels."""
import sys, time, torch
sys.path.insert(0, ".")
from mop_amd.nn import EdgewiseMSA
from torch.profiler import profile, ProfilerActivity

torch.manual_seed(0)
m = EdgewiseMSA(384, 6, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4, use_lens_bank=True).cuda().to(torch.bfloat16)
x = torch.randn(256, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
w = torch.randn_like(x)
for _ in range(3):
    m(x).backward(w)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as p:
    m(x).backward(w)
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=90))
