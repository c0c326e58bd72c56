#!/usr/bin/env python3
"""dev tool (GPU box): fused EdgewiseMSA core fwd/bwd kernel times with and without forward-saved chain state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mop_amd
from mop_amd import ops
from mop_amd.nn import EdgewiseMSA

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = EdgewiseMSA(384, 6, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4, gate_init="mix5").cuda().to(torch.bfloat16)
x = torch.randn(B, 197, 384, device="cuda", dtype=torch.bfloat16, requires_grad=True)
for save in (False, True, False, True):
    ops.set_save_chain_state(save)
    for _ in range(3):
        m(x).sum().backward()
    ops.enable_timing(True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        m(x).sum().backward()
    e1.record(); torch.cuda.synchronize()
    t = ops.timing_results()
    ops.enable_timing(False)
    print(f"save={save}: step {e0.elapsed_time(e1)/10:.3f} ms  " + "  ".join(f"{k} {sum(v)/len(v):.3f} ms" for k, v in t.items()), flush=True)
