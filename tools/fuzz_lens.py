import sys, random, torch
sys.path.insert(0, ".")
import mop_amd
from mop_amd import ops, _lib
from mop_amd.nn import EdgewiseMSA
random.seed(1); torch.manual_seed(1)
mop_amd.set_precision("bf16")
bad = 0
for it in range(40):
    H = random.choice([1, 2, 4]); dk = random.choice([16, 32, 64]); D = H * dk
    V = random.randint(2, 5); r = random.randint(1, 4); N = random.choice([1, 2, 7, 31, 32, 33, 65, 100, 128, 160, 197, 224]); B = random.randint(1, 3)
    L = random.randint(1, 4)
    while 2 * V + 2 + L * V > 26: L -= 1
    dil = tuple(random.choice([1, 2, 3, 5, 9, 40]) for _ in range(L))
    m = EdgewiseMSA(D, H, n_views=V, share_qkv=True, gate_mode="lowrank", gate_rank=r, gate_init="mix5", use_lens_bank=True, lens_dilations=dil).cuda().to(torch.bfloat16)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("_scale"): p.add_(0.1 * torch.randn_like(p))
    x = torch.randn(B, N, D, device="cuda", dtype=torch.bfloat16)
    w = torch.randn_like(x)
    res = {}
    for path in ("generic", "auto"):
        ops.set_path(path)
        xi = x.clone().requires_grad_(True); m.zero_grad()
        y = m(xi); y.backward(w)
        res[path] = (y.detach().float(), xi.grad.float(), m.lens_bank[0].weight.grad.float().clone(), ops.LAST_PATH["edgewise_fwd"])
    ops.set_path("auto")
    g, f = res["generic"], res["auto"]
    ey = float((g[0] - f[0]).abs().max()); ex = float((g[1] - f[1]).abs().max()) / max(1e-6, float(g[1].abs().max()))
    el = float((g[2] - f[2]).abs().max()) / max(1e-6, float(g[2].abs().max()))
    ok = ey <= 2e-2 and ex <= 6e-2 and torch.isfinite(f[0]).all() and torch.isfinite(f[1]).all()
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} B{B} N{N} H{H} dk{dk} V{V} r{r} dil{dil} path {f[3]}: y {ey:.2e} dx {ex:.2e} dlens {el:.2e}", flush=True)
print("bad", bad)
