#!/usr/bin/env python3
"""dev tool (GPU box): one ViT-MoP 5M training step (BASELINE.json configs[2] per-GPU slice: B=256, 32x32 images, CE loss,
AdamW) on one MI355X, bf16 autocast-free (module in bf16).  Attention core = fused SDPA kernels (mop_amd MSA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mop_amd.nn import ViT_MoP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = ViT_MoP(dim=384, depth=3, heads=6, n_classes=100, n_views=5, n_kernels=3, drop_path=0.0).cuda().to(torch.bfloat16)
opt = torch.optim.AdamW(m.parameters(), lr=3e-3, weight_decay=5e-2)
x = torch.randn(B, 3, 32, 32, device="cuda", dtype=torch.bfloat16)
y = torch.randint(0, 100, (B,), device="cuda")
print("params", sum(p.numel() for p in m.parameters()))


def step():
    opt.zero_grad(set_to_none=True)
    loss = F.cross_entropy(m(x).float(), y)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    loss = step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"ViT-MoP 5M train step B={B}: {ms:.3f} ms/step  {B / ms * 1e3:.0f} img/s  loss {float(loss):.3f}")
