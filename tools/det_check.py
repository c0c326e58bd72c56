#!/usr/bin/env python3
"""dev tool: run-to-run bitwise determinism of the fused path at several batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
layer = bench.build_layer(torch.bfloat16)
for B in (4, 42, 43, 64, 128, 256):
    g = torch.Generator(device="cuda").manual_seed(B)
    x = torch.randn(B, 197, 384, device="cuda", dtype=torch.bfloat16, generator=g)
    dy = torch.randn(B, 197, 384, device="cuda", dtype=torch.bfloat16, generator=g)
    outs = []
    for rep in range(3):
        layer.zero_grad()
        xx = x.clone().requires_grad_(True)
        y = layer(xx); y.backward(dy); torch.cuda.synchronize()
        outs.append((y.detach().clone(), xx.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}))
    msg = []
    for rep in (1, 2):
        dxe = (outs[rep][1] != outs[0][1])
        bad_b = dxe.flatten(1).any(1).nonzero().flatten().tolist()
        msg.append(f"rep{rep}: y_eq={torch.equal(outs[rep][0], outs[0][0])} dx_neq_samples={bad_b[:12]} n={len(bad_b)} "
                   + " ".join(k.split('.')[-2][:5] + ":" + str(bool(torch.equal(outs[rep][2][k], outs[0][2][k]))) for k in outs[0][2] if '.' in k))
    print(f"B={B}", " | ".join(msg))
