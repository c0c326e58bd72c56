"""numpy oracle for plain scaled-dot-product attention  --  TEST INFRASTRUCTURE ONLY.

Restates the three identical SDPA cores of the reference:
  * `BaselineMSA.forward`            mop/models/attention_variants.py:36-48
  * `MSA.forward`                    mop/models/components.py:56-66
  * `MultiheadSelfAttention.forward` mop/models/whisper_mop.py:163-175 (bool mask / additive bias)
"""
from __future__ import annotations

import math

import numpy as np

from .edgewise import _heads
from .multihop import _masked_softmax


def core_fwd(q, k, v, blocked=None, bias=None, drop=None):
    """drop: None or the dropout multiplier keep / (1 - p) per edge, broadcastable to (B,H,N,N): `self.attn_drop(A)` of
    attention_variants.py:45 / components.py:62 / whisper_mop.py:172 with the mask made explicit"""
    dk = q.shape[-1]
    scale = 1.0 / math.sqrt(dk)
    S = np.matmul(q, np.swapaxes(k, -1, -2)) * scale
    if bias is not None:
        S = S + bias
    if blocked is not None:
        blocked = np.broadcast_to(blocked, S.shape)
    P = _masked_softmax(S, blocked)
    Pd = P if drop is None else P * drop
    y = np.matmul(Pd, v)
    return y, dict(q=q, k=k, v=v, P=P, Pd=Pd, drop=drop, scale=scale)


def core_bwd(dy, c):
    P = c["P"]
    dP = np.matmul(dy, np.swapaxes(c["v"], -1, -2))
    if c.get("drop") is not None:
        dP = dP * c["drop"]
    dv = np.matmul(np.swapaxes(c.get("Pd", P), -1, -2), dy)
    dS = P * (dP - (P * dP).sum(-1, keepdims=True)) * c["scale"]
    return dict(dq=np.matmul(dS, c["k"]), dk=np.matmul(np.swapaxes(dS, -1, -2), c["q"]), dv=dv)


def baseline_module_fwd(x, params, heads, attn_mask=None):
    """BaselineMSA: qkv.weight (3D,D), proj.weight (D,D), no biases."""
    B, N, D = x.shape
    H, dk = heads, D // heads
    t = _heads(x @ params["qkv.weight"].T, B, N, H, dk)
    blocked = None if attn_mask is None else (np.asarray(attn_mask) == 0)
    y, cache = core_fwd(t[0], t[1], t[2], blocked)
    ycat = np.transpose(y, (0, 2, 1, 3)).reshape(B, N, D)
    cache.update(x=x, ycat=ycat, params=params, H=H)
    return ycat @ params["proj.weight"].T, cache


def baseline_module_bwd(dout, cache):
    p, x, H = cache["params"], cache["x"], cache["H"]
    B, N, D = x.shape
    dk = D // H
    grads = {"proj.weight": np.einsum("bno,bni->oi", dout, cache["ycat"])}
    dy = np.transpose((dout @ p["proj.weight"]).reshape(B, N, H, dk), (0, 2, 1, 3))
    g = core_bwd(dy, cache)
    d = np.transpose(np.stack([g["dq"], g["dk"], g["dv"]]), (1, 3, 0, 2, 4)).reshape(B, N, 3 * D)
    grads["qkv.weight"] = np.einsum("bno,bni->oi", d, x)
    return d @ p["qkv.weight"], grads
