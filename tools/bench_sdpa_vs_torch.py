"""dev tool (GPU box): the fused plain-SDPA core against torch's scaled_dot_product_attention at the Whisper (T=3000), ViT (N=197)
and Quartet-sized (T=1024, causal) shapes, bf16, forward and forward+backward (core only, no Linears)."""
import sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from mop_amd import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


for (B, T, H, dk, causal) in ((8, 3000, 6, 64, False), (256, 197, 6, 64, False), (8, 1024, 12, 64, True)):
    q, k, v = (torch.randn(B, T, H, dk, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    w = torch.randn(B, T, H * dk, device="cuda", dtype=torch.bfloat16)
    def ours(bwd):
        y = ops.sdpa_core(q, k, v, causal=causal)
        if bwd:
            y.backward(w)
    def ref(bwd):
        y = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=causal).transpose(1, 2).reshape(B, T, H * dk)
        if bwd:
            y.backward(w)
    flops = 4.0 * B * H * T * T * dk * (0.5 if causal else 1.0)
    with torch.no_grad():
        fo, fr = timed(lambda: ours(False)), timed(lambda: ref(False))
    bo, br = timed(lambda: ours(True)), timed(lambda: ref(True))
    print(f"B={B} T={T} H={H} dk={dk} causal={causal}: fwd ours {fo:.3f} ms ({flops / fo / 1e9:.0f} TF) torch {fr:.3f} ms | fwd+bwd ours {bo:.3f} ms torch {br:.3f} ms", flush=True)
