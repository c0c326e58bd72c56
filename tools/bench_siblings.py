#!/usr/bin/env python3
"""dev tool (GPU box): fwd / fwd+bwd times of the sibling attention modules at the BASELINE.json config shapes (bf16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mop_amd
from mop_amd.nn import (BaselineMSA, CausalSelfAttention, CrossViewMixerMSA, MultiHopMSA, MultiheadSelfAttention,
                        TransformerConfig)


def timeit(fn, n=10, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(name, m, x, flops_fwd):
    m = m.cuda().to(torch.bfloat16)
    x = x.cuda().to(torch.bfloat16).requires_grad_(True)
    with torch.no_grad():
        tf = timeit(lambda: m(x))
    tb = timeit(lambda: m(x).sum().backward())
    print(f"{name:58s} fwd {tf:8.3f} ms  fwd+bwd {tb:8.3f} ms   core fwd {flops_fwd / tf / 1e9:7.1f} TFLOP/s", flush=True)


torch.manual_seed(0)
B = int(os.environ.get("SB", "8"))
# config 2: ViT-MoP MSA at N=65, d=384, 6 heads (B=256)
run("BaselineMSA N=65 d=384 H=6 B=256", BaselineMSA(384, 6), torch.randn(256, 65, 384), 256 * 6 * 4 * 65 * 65 * 64)
# config 1: N=197
run("BaselineMSA N=197 d=384 H=6 B=256", BaselineMSA(384, 6), torch.randn(256, 197, 384), 256 * 6 * 4 * 197 * 197 * 64)
run("MultiHopMSA N=197 d=384 H=6 B=256 (default gates, hops=3)", MultiHopMSA(384, 6), torch.randn(256, 197, 384), 256 * 6 * (4 + 4 + 6) * 197 * 197 * 64)
run("CrossViewMixerMSA N=197 d=384 H=6 B=256", CrossViewMixerMSA(384, 6), torch.randn(256, 197, 384), 256 * 6 * 10 * 197 * 197 * 64)
# config 4: Quartet T=1024 d=768 H=12
cfg = TransformerConfig(n_head=12, n_embd=768, block_size=1024, dropout=0.0)
run(f"CausalSelfAttention(Quartet) T=1024 d=768 H=12 B={B}", CausalSelfAttention(cfg), torch.randn(B, 1024, 768), B * 12 * 6 * 1024 * 1024 * 64)
# config 5: Whisper encoder self-attention T=3000 d=384 H=6
run(f"MultiheadSelfAttention T=3000 d=384 H=6 B={B}", MultiheadSelfAttention(384, 6, 0.0, False, causal=False), torch.randn(B, 3000, 384), B * 6 * 4 * 3000 * 3000 * 64)
