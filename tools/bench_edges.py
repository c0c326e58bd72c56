"""dev tool (GPU box): the edges of the attention core at the bench shape (B=256, N=197, D=384, bf16) -- torch LayerNorm + separate
residual add vs the libmopk LN prologue + residual GEMM epilogue.  Prints per-op milliseconds (HIP events, median of 20)."""
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from mop_amd import ops                                  # noqa: E402
from mop_amd.nn.linear import residual_linear           # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def main():
    B, N, D = 256, 197, 384
    dt = torch.bfloat16
    x = torch.randn(B, N, D, device="cuda", dtype=dt, requires_grad=True)
    g = torch.ones(D, device="cuda", dtype=dt, requires_grad=True)
    b = torch.zeros(D, device="cuda", dtype=dt, requires_grad=True)
    W = (torch.randn(D, D, device="cuda", dtype=dt) * 0.05).requires_grad_(True)
    yc = torch.randn(B, N, D, device="cuda", dtype=dt, requires_grad=True)
    w = torch.randn(B, N, D, device="cuda", dtype=dt)
    gb = B * N * D * 2 / 1e9

    def torch_edges():
        h = F.layer_norm(x, (D,), g, b, 1e-5)
        out = x + F.linear(yc, W) + 0 * h.sum()          # keeps LN in the graph without a qkv GEMM
        out.backward(w)

    def torch_ln_fwd(): return F.layer_norm(x, (D,), g, b, 1e-5)
    def mopk_ln_fwd(): return ops.layernorm(x, g, b, 1e-5)
    print(f"LN forward   torch {timeit(torch_ln_fwd):.4f} ms   libmopk {timeit(mopk_ln_fwd):.4f} ms   (2 x {gb:.3f} GB algorithmic)")

    def torch_ln_fb():
        h = F.layer_norm(x, (D,), g, b, 1e-5); h.backward(w)
    def mopk_ln_fb():
        h = ops.layernorm(x, g, b, 1e-5); h.backward(w)
    print(f"LN fwd+bwd   torch {timeit(torch_ln_fb):.4f} ms   libmopk {timeit(mopk_ln_fb):.4f} ms")

    def torch_branch():
        h = F.layer_norm(x, (D,), g, b, 1e-5)
        out = x + F.linear(h, W)
        out.backward(w)
    def mopk_branch():
        xr, h = ops.layernorm_residual(x, g, b, 1e-5)
        out = residual_linear(xr, h, W)
        out.backward(w)
    print(f"x + lin(ln(x)) fwd+bwd   torch {timeit(torch_branch):.4f} ms   fused edges {timeit(mopk_branch):.4f} ms")


if __name__ == "__main__":
    main()
