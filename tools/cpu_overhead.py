"""dev tool (GPU box): host-side cost of enqueueing one op (no synchronisation inside the loop): the time the Python wrapper, ctypes
marshalling and the launch take, against torch's own SDPA / LayerNorm calls"""
import sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from mop_amd import ops

def host_us(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return dt / n * 1e6

B, T, H, dk = 4, 65, 6, 64
q, k, v = (torch.randn(B, T, H, dk, device="cuda", dtype=torch.bfloat16) for _ in range(3))
x = torch.randn(B * T, 384, device="cuda", dtype=torch.bfloat16)
g = torch.ones(384, device="cuda", dtype=torch.bfloat16); b = torch.zeros_like(g)
with torch.no_grad():
    print("sdpa_core fwd (no grad)     %.1f us" % host_us(lambda: ops.sdpa_core(q, k, v)))
    print("torch SDPA fwd (no grad)    %.1f us" % host_us(lambda: F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))))
    print("ops.layernorm fwd           %.1f us" % host_us(lambda: ops.layernorm(x, g, b, 1e-5)))
    print("torch layer_norm fwd        %.1f us" % host_us(lambda: F.layer_norm(x, (384,), g, b, 1e-5)))
qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
w = torch.randn(B, T, H * dk, device="cuda", dtype=torch.bfloat16)
print("sdpa_core fwd+bwd           %.1f us" % host_us(lambda: ops.sdpa_core(qg, kg, vg).backward(w)))
print("torch SDPA fwd+bwd          %.1f us" % host_us(lambda: F.scaled_dot_product_attention(qg.transpose(1, 2), kg.transpose(1, 2), vg.transpose(1, 2)).transpose(1, 2).reshape(B, T, H * dk).backward(w)))
