"""dev tool (GPU box): EdgewiseMSA layer (low-rank head, V=5, r=4, d=384, H=6, bf16) at the small sequence lengths of the reference's
CIFAR experiments (N = 65: 32x32 images, patch 4) and at N = 128 / 197: training step and inference forward"""
import sys, time, torch
sys.path.insert(0, ".")
from mop_amd.nn import EdgewiseMSA


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


for N, B in ((65, 256), (65, 1024), (128, 256), (145, 256), (170, 256), (197, 256)):
    torch.manual_seed(0)
    m = EdgewiseMSA(384, 6, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4).cuda().to(torch.bfloat16)
    x = torch.randn(B, N, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    w = torch.randn_like(x)
    step = timed(lambda: m(x).backward(w))
    def infer():
        with torch.no_grad():
            m(x)
    print(f"N={N:4d} B={B:5d}: train step {step:7.3f} ms ({B / step:8.1f} k img/s) | inference fwd {timed(infer):6.3f} ms", flush=True)
