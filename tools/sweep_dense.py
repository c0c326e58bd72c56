"""dev tool (GPU box): fused dense-head forward + backward over a grid of shapes / dtypes (every NT and DK instantiation, V = 2..6,
ragged N), compared loosely with the generic path -- a smoke sweep for faults and gross errors, not a parity test."""
import itertools, sys, torch
sys.path.insert(0, ".")
import mop_amd
from mop_amd import ops
from mop_amd.nn import EdgewiseMSA

torch.manual_seed(0)
mop_amd.set_precision("bf16")
bad = 0
cases = [(B, N, H, dk, V, dt) for (N, B) in ((5, 3), (32, 2), (33, 1), (64, 2), (65, 2), (96, 1), (100, 1), (129, 1), (145, 1), (170, 1), (192, 1), (197, 2), (224, 1))
         for (H, dk) in ((2, 16), (1, 32), (2, 64)) for V in (2, 3, 5, 6) for dt in (torch.bfloat16, torch.float32)]
for i, (B, N, H, dk, V, dt) in enumerate(cases):
    D = H * dk
    m = EdgewiseMSA(D, H, n_views=V, share_qkv=True, gate_mode="dense", use_k3=False, gate_init="and").cuda().to(dt)
    with torch.no_grad():
        m.edge_head.conv2.bias.copy_(0.5 * torch.randn(4))
    x = torch.randn(B, N, D, device="cuda", dtype=dt)
    w = torch.randn_like(x)
    res = {}
    for path in ("auto", "generic"):
        ops.set_path(path)
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        y.backward(w)
        torch.cuda.synchronize()
        res[path] = (y.detach().float(), xi.grad.float(), {k: p.grad.float().clone() for k, p in m.named_parameters()}, dict(ops.LAST_PATH))
    ops.set_path("auto")
    (yf, dxf, gf, pf), (yg, dxg, gg, _) = res["auto"], res["generic"]
    if pf["edgewise_fwd"] != 2 or pf["edgewise_bwd"] != 2:          # outside the fused kernels (e.g. LDS budget at V = 6, N > 128, dk = 64)
        print(f"generic  B={B} N={N} H={H} dk={dk} V={V}", flush=True)
        continue
    ey = float((yf - yg).abs().max() / yg.abs().max().clamp_min(1e-6))
    ex = float((dxf - dxg).abs().max() / dxg.abs().max().clamp_min(1e-6))
    scale = max(float(v.abs().max()) for v in gg.values())
    eg = max(float((gf[k] - gg[k]).abs().max()) / max(float(gg[k].abs().max()), 1e-2 * scale) for k in gg)
    ok = all(torch.isfinite(t).all() for t in (yf, dxf)) and ey < 3e-2 and ex < 6e-2 and eg < 5e-1       # both sides are bf16 runs: the small gate-head gradients differ by their own noise
    if not ok:
        bad += 1
        print(f"BAD  B={B} N={N} H={H} dk={dk} V={V} {dt}: y {ey:.2e} dx {ex:.2e} grads {eg:.2e}", flush=True)
    if i % 24 == 0:
        print(f"[{i}/{len(cases)}] B={B} N={N} H={H} dk={dk} V={V} {dt}: y {ey:.2e} dx {ex:.2e} grads {eg:.2e}", flush=True)
print("cases", len(cases), "bad", bad)
sys.exit(1 if bad else 0)
