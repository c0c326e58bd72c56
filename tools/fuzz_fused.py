#!/usr/bin/env python3
"""dev tool (GPU box): random-shape comparison of the fused attention kernels against the generic path (fwd + all gradients)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mop_amd
from mop_amd import ops

mop_amd.set_precision("bf16")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = {}


def cmp(tag, shape, a, b, tol):
    # gradients are judged against the largest gradient of the call (a degenerate N=1 softmax has analytically zero dq/dk,
    # where both paths only hold bf16 rounding noise); scalar parameter gradients are cancellation-dominated sums
    gscale = max(float(y.detach().abs().max()) for y in b[1:])
    for i, (x, y) in enumerate(zip(a, b)):
        x, y = x.detach(), y.detach()
        den = max(float(y.abs().max()), 1.0) if i == 0 else max(float(y.abs().max()), 0.05 * gscale)
        err = float((x - y).abs().max()) / den
        lim = 0.3 if y.numel() == 1 else tol
        worst[tag] = max(worst.get(tag, 0.0), err if y.numel() > 1 else 0.0)
        if not (err <= lim) or not torch.isfinite(x).all():
            print(f"FAIL {tag} {shape} output {i}: rel err {err:.3e}", flush=True)
            return False
    return True


def run(fn, tensors, dy, extra_params=()):
    out = {}
    for path in ("fused", "generic"):
        ops.set_path(path)
        ts = [t.clone().requires_grad_(True) for t in tensors]
        ps = [p.clone().requires_grad_(True) for p in extra_params]
        y = fn(ts, ps)
        y.backward(dy)
        out[path] = [y.float()] + [t.grad.float() for t in ts] + [p.grad.float() for p in ps]
    ops.set_path("auto")
    return out


ok = True
for case in range(n_cases):
    kind = random.choice(["sdpa", "dual", "quartet"])
    B, H = random.randint(1, 3), random.randint(1, 4)
    N = random.choice([1, 2, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 197, 255, 256, 300, 500])
    dk = random.choice([32, 64])
    dt = random.choice([torch.bfloat16, torch.float32])
    causal = random.random() < 0.5
    g = torch.Generator(device="cuda").manual_seed(case)
    mk = lambda: torch.randn(B, N, H, dk, device="cuda", generator=g).to(dt)
    dy = torch.randn(B, N, H * dk, device="cuda", generator=g).to(dt)
    shape = (kind, B, N, H, dk, str(dt).split(".")[-1], causal)
    if kind == "sdpa":
        r = run(lambda ts, ps: ops.sdpa_core(ts[0], ts[1], ts[2], causal=causal), [mk(), mk(), mk()], dy)
    elif kind == "dual":
        hops, gates = random.choice([2, 3, 4]), (random.choice([1.0, 0.6]), random.choice([0.0, 0.5]), random.choice([0.0, 0.4]))
        shape += (hops, gates)
        r = run(lambda ts, ps: ops.dualpath_core(*ts, ps[0], gates[0], gates[1], gates[2], 0.0, 0.5, hops, causal=causal),
                [mk() for _ in range(6)], dy, [torch.tensor(-0.4, device="cuda")])
    else:
        uq = random.random() < 0.7
        shape += (uq,)
        if uq:
            r = run(lambda ts, ps: ops.quartet_core(ts[0], ts[1], ts[2], ts[3], ts[4], ps[0], ps[1], None, 1e-5, True),
                    [mk() for _ in range(5)], dy, [torch.tensor([0.2], device="cuda"), torch.tensor([0.9], device="cuda")])
        else:
            r = run(lambda ts, ps: ops.quartet_core(ts[0], ts[1], ts[2], None, None, None, None, None, 1e-5, False), [mk() for _ in range(3)], dy)
    ok &= cmp(kind, shape, r["fused"], r["generic"], 6e-2)
print("worst relative differences:", {k: f"{v:.2e}" for k, v in worst.items()}, "ALL OK" if ok else "FAILURES", flush=True)
