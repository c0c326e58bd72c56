#!/usr/bin/env python3
"""dev tool (GPU box): per-gradient error of the fused path vs the reference golden vectors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import mop_amd
from mop_amd import ops, _lib
from mop_amd.nn import EdgewiseMSA
from conftest import golden_names, load_golden
from gpu_util import module_from_golden, run_fwd_bwd, rel_err, max_abs

ops.set_save_chain_state(os.environ.get("MOPK_SAVE", "1") != "0")
names = sys.argv[1:] or [n for n in golden_names("ew_") if "unshared" not in n]
for name in names:
    d, params, gref, meta = load_golden(name)
    mop_amd.set_precision("bf16")
    ctor = dict(dim=meta["dim"], heads=meta["heads"], n_views=meta["n_views"], share_qkv=bool(meta["share_qkv"]),
                gate_mode="lowrank", gate_rank=meta["gate_rank"], beta_not=meta["beta_not"])
    res = {}
    for path in ("generic", "auto"):
        ops.set_path(path)
        m = module_from_golden(EdgewiseMSA, params, **ctor)
        y, dx, grads = run_fwd_bwd(m, d["x"], d["w"])
        res[path] = (y, dx, grads, dict(ops.LAST_PATH))
    print(f"== {name}  paths: {res['auto'][3]}")
    for path in ("generic", "auto"):
        y, dx, grads, _ = res[path]
        line = f"  {path:8s} y {max_abs(y, d['y']):.2e} dx {rel_err(dx, d['dx']):.2e} "
        line += " ".join(f"{k.split('.')[-2][:6] if '.' in k else k[:8]}.{k.split('.')[-1][:1]} {rel_err(grads[k].reshape(gref[k].shape), gref[k]):.1e}" for k in gref)
        print(line)
