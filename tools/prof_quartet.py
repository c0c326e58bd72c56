"""dev tool (GPU box): Quartet CausalSelfAttention fwd+bwd at T=1024, d=768, H=12, B=8 (bf16) for rocprofv3 --kernel-trace --stats"""
import sys, torch
sys.path.insert(0, ".")
from mop_amd.nn import CausalSelfAttention, TransformerConfig
torch.manual_seed(0)
cfg = TransformerConfig(n_embd=768, n_head=12, block_size=1024, dropout=0.0, bias=True, use_quartet=True)
m = CausalSelfAttention(cfg).cuda().to(torch.bfloat16)
x = torch.randn(8, 1024, 768, device="cuda", dtype=torch.bfloat16, requires_grad=True)
w = torch.randn_like(x)
for _ in range(6):
    out = m(x)
    out = out[0] if isinstance(out, tuple) else out
    out.backward(w)
torch.cuda.synchronize()
