"""dev tool (GPU box): operator-level view (torch.profiler) of one ViT-MoP training step: which aten ops launch the small kernels"""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from mop_amd.nn import ViT_MoP
from mop_amd.training import DataParallelStep, make_optimizer_and_schedule
torch.manual_seed(0)
model = ViT_MoP(dim=384, depth=3, heads=6, n_classes=100, drop_path=0.0).cuda().to(torch.bfloat16)
opt, sched = make_optimizer_and_schedule(model, 3e-3, 5e-2, steps=100)
dp = DataParallelStep(model, opt, lambda out, tgt: F.cross_entropy(out.float(), tgt), sched)
x = torch.randn(256, 3, 32, 32, device="cuda").to(torch.bfloat16)
y = torch.randint(0, 100, (256,), device="cuda")
for _ in range(5):
    dp(x, y)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        dp(x, y)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
