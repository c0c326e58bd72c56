#!/usr/bin/env python3
"""dev tool (CPU): which bf16 / fp16 roundings inside the fused Edgewise path limit the accuracy of the gate-head weight gradients.

Runs the float64 oracle on a reference fixture, then repeats the gate-gradient contraction with ONE intermediate rounded the way
the kernels round it, and prints max-abs error / max|ref| for dWr, dWc.  The last rows round the kernel INPUTS (q, k to bf16, the
scaled queries Qe = bf16(q * sqk)): that error is shared by every bf16-input implementation, the reference's own bf16 run included."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # run as: python tests/probe_grad_noise.py (lives under tests/: it imports the oracle)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from oracle import edgewise as oe


def bf16(x):
    x = np.ascontiguousarray(x, dtype=np.float32); u = x.view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7fff)) & 0xffff0000).view(np.float32).astype(np.float64)


def fp16(x):
    return np.asarray(x, dtype=np.float64).astype(np.float16).astype(np.float64)


def gate_grads(c, dyc, P, Cr, L, rnd):
    S, G = c["S"], c["G"]
    B, H, N = S.shape[1], S.shape[2], S.shape[3]
    dP = np.matmul(dyc, np.swapaxes(c["v0"], -1, -2))
    dSmix = P * (dP - (P * dP).sum(-1, keepdims=True))
    O, nb = c["O"], c["nb"]
    dZ = rnd(np.stack([dSmix * O, dSmix * L, -dSmix * (nb * O), dSmix * Cr], axis=2) * G * (1 - G))
    a4, b4 = c["a4"], c["b4"]
    r = a4.shape[3]
    da = np.einsum("bhgnm,bhgkm->bhgkn", dZ, b4).reshape(B, H, 4 * r, N)
    db = np.einsum("bhgnm,bhgkn->bhgkm", dZ, a4).reshape(B, H, 4 * r, N)
    return np.einsum("bhon,bhcn->oc", da, c["row_feat"]), np.einsum("bhon,bhcn->oc", db, c["col_feat"])


def main(name):
    d, params, gref, meta = load_golden(name)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    H = meta["heads"]
    x, w = d["x"].astype(np.float64), d["w"].astype(np.float64)
    out, c = oe.module_fwd(x, p64, H, meta["n_views"], bool(meta["share_qkv"]), meta["beta_not"])
    B, N, D = w.shape
    dyc = (w @ p64["proj.weight"]).reshape(B, N, H, D // H).transpose(0, 2, 1, 3)
    ident = lambda t: t
    L = c["lse"] - c["S"][0]
    ref = gate_grads(c, dyc, c["P"], c["Cr"], L, ident)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    rep = lambda tag, g: print(f"  {tag:58s} dWr {rel(g[0], ref[0]):.1e}  dWc {rel(g[1], ref[1]):.1e}")
    print(f"{name}: |dWr|max {np.abs(ref[0]).max():.3e}  (largest gradient of the module {max(np.abs(v).max() for v in gref.values()):.3e})")
    rep("dZ rounded to bf16 before the two contractions", gate_grads(c, dyc, c["P"], c["Cr"], L, bf16))
    rep("dZ as bf16 value + bf16 remainder", gate_grads(c, dyc, c["P"], c["Cr"], L, lambda z: bf16(z) + bf16(z - bf16(z))))
    Sm = fp16(c["Smix"]); e = np.exp(Sm - Sm.max(-1, keepdims=True))
    rep("Smix stored as fp16", gate_grads(c, dyc, e / e.sum(-1, keepdims=True), c["Cr"], L, ident))
    rep("C-> stored as bf16 (log taken afterwards)", gate_grads(c, dyc, c["P"], np.log(bf16(c["Cf"]) + oe.EPS_CHAIN), L, ident))
    rep("L = lse - S0 stored as fp16", gate_grads(c, dyc, c["P"], c["Cr"], fp16(L * 1.4426950408889634) / 1.4426950408889634, ident))
    # inputs rounded: rerun the whole oracle on bf16(x)-derived q, k (module level: x and the qkv weight rounded to bf16)
    pb = dict(p64)
    for k in pb:
        if k.endswith("qkv.weight") or "qkv_list" in k:
            pb[k] = bf16(pb[k])
    out_b, cb = oe.module_fwd(bf16(x), pb, H, meta["n_views"], bool(meta["share_qkv"]), meta["beta_not"])
    dyb = (w @ p64["proj.weight"]).reshape(B, N, H, D // H).transpose(0, 2, 1, 3)
    rep("INPUTS: x and qkv.weight rounded to bf16 (exact arithmetic)", gate_grads(cb, dyb, cb["P"], cb["Cr"], cb["lse"] - cb["S"][0], ident))


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["ew_ns_shared_v5_r4_mix5", "ew_mid_shared_v5_r4_chain", "ew_odd_shared_v5_r4_mix5"]):
        main(n)
