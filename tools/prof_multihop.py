"""dev tool (GPU box): MultiHopMSA and CrossViewMixerMSA fwd+bwd at N=197, B=256 for rocprofv3 --kernel-trace --stats"""
import sys, torch
sys.path.insert(0, ".")
from mop_amd.nn import MultiHopMSA, CrossViewMixerMSA
torch.manual_seed(0)
for ctor in (lambda: MultiHopMSA(384, 6), lambda: CrossViewMixerMSA(384, 6)):
    m = ctor().cuda().to(torch.bfloat16)
    x = torch.randn(256, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    w = torch.randn_like(x)
    for _ in range(4):
        m(x).backward(w)
torch.cuda.synchronize()
