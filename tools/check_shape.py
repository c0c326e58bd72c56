#!/usr/bin/env python3
"""dev tool: fused vs generic(bf16) vs oracle(float64) errors on a seeded random shape: B N D H V r"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mop_amd
from mop_amd import ops
from oracle import edgewise as oe
from gpu_util import run_fwd_bwd, rel_err, max_abs
from test_gpu_edgewise import _mk
B, N, D, H, V, r = map(int, sys.argv[1:7])
m = _mk(D, H, V, r, seed=B * 1000 + N)
params = {k: v.detach().numpy().astype(np.float64) for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(N)
x = torch.randn(B, N, D, generator=g).numpy(); w = torch.randn(B, N, D, generator=g).numpy()
out, cache = oe.module_fwd(x.astype(np.float64), params, H, V, True, 0.5)
dx_ref, g_ref = oe.module_bwd(w.astype(np.float64), cache)
mop_amd.set_precision("bf16")
for path in ("generic", "auto"):
    ops.set_path(path)
    mm = _mk(D, H, V, r, seed=B * 1000 + N).cuda().eval()
    y, dx, grads = run_fwd_bwd(mm, x, w)
    print(f"{path:8s} y {max_abs(y,out):.2e} dx {rel_err(dx,dx_ref):.2e} " + " ".join(f"{k[-22:]}={rel_err(grads[k].reshape(g_ref[k].shape), g_ref[k]):.1e}|ref{np.abs(g_ref[k]).max():.1e}" for k in g_ref))
ops.set_save_chain_state(False)
mm = _mk(D, H, V, r, seed=B * 1000 + N).cuda().eval()
y, dx, grads = run_fwd_bwd(mm, x, w)
print(f"{'recompute':8s} y {max_abs(y,out):.2e} dx {rel_err(dx,dx_ref):.2e} " + " ".join(f"{k[-22:]}={rel_err(grads[k].reshape(g_ref[k].shape), g_ref[k]):.1e}|ref{np.abs(g_ref[k]).max():.1e}" for k in g_ref))
