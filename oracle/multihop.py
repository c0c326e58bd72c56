"""numpy oracle for MultiHopMSA (dual-path, scalar gates)  --  TEST INFRASTRUCTURE ONLY.

Restates reference `mop/models/attention_variants.py:163-231` (`MultiHopMSA`) and
`_lse` (:159-160); with the default gates it is exactly
``softmax(S1 + S2) v1 + sigmoid(w) * A1 A2^(hops-1) v2``.
Backward is hand-derived and pinned against reference autograd (tests/golden).

attn_mask: broadcastable to (B,H,N,N), 0 = blocked (:202-205, :219-220).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

from .edgewise import _heads, _sigmoid, _softmax

EPS_CHAIN = 1e-6  # :217
DEFAULT_GATES = dict(and_=1.0, or_=0.0, not_=0.0, chain=0.0, base=1.0)  # :188


def _masked_softmax(S, blocked):
    if blocked is None:
        return _softmax(S, -1)
    Sm = np.where(blocked, -np.inf, S)
    m = Sm.max(-1, keepdims=True)
    e = np.where(blocked, 0.0, np.exp(Sm - m))
    return e / e.sum(-1, keepdims=True)


def core_fwd(q1, k1, v1, q2, k2, v2, gates, beta_not, hops, chain_logit, blocked=None, drop=None):
    """q*,k*,v*: (B,H,N,dk); blocked: bool array broadcastable to (B,H,N,N) or None;
    drop: None or the attn_drop multiplier keep / (1 - p) per edge on the mixed weights (:222 with the mask made explicit)."""
    dk = q1.shape[-1]
    scale = 1.0 / math.sqrt(dk)
    S1 = np.matmul(q1, np.swapaxes(k1, -1, -2)) * scale          # :200
    S2 = np.matmul(q2, np.swapaxes(k2, -1, -2)) * scale          # :201
    if blocked is not None:
        blocked = np.broadcast_to(blocked, S1.shape)
    A1 = _masked_softmax(S1, blocked)                             # :206
    A2 = _masked_softmax(S2, blocked)                             # :207
    g_and = gates.get("and_", 1.0)
    g_or = gates.get("or_", 0.0)
    g_not = gates.get("not_", 0.0)
    g_ch = gates.get("chain", 0.0)
    mx = np.maximum(S1, S2)
    lse = mx + np.log(np.exp(S1 - mx) + np.exp(S2 - mx))          # :159-160
    T = [A1]                                                      # :214-216
    for _ in range(hops - 1):
        T.append(np.matmul(T[-1], A2))
    C = T[-1]
    Smix = S1 + g_and * S2 + g_or * (lse - S1) - g_not * (beta_not * S2) \
        + g_ch * np.log(C + EPS_CHAIN)                            # :209-218
    P = _masked_softmax(Smix, blocked)                            # :219-221
    tr = [v2]                                                     # :224-226
    for _ in range(hops - 1):
        tr.append(np.matmul(A2, tr[-1]))
    y_chain = np.matmul(A1, tr[-1])                               # :227
    w = _sigmoid(chain_logit)
    Pd = P if drop is None else P * drop                          # :222
    y = np.matmul(Pd, v1) + w * y_chain                           # :229
    cache = dict(q1=q1, k1=k1, v1=v1, q2=q2, k2=k2, v2=v2, S1=S1, S2=S2, A1=A1, A2=A2, T=T,
                 C=C, P=P, tr=tr, y_chain=y_chain, w=w, lse=lse, blocked=blocked, scale=scale,
                 g=(g_and, g_or, g_not, g_ch), beta=beta_not, hops=hops, Pd=Pd, drop=drop)
    return y, cache


def core_bwd(dy, c):
    A1, A2, P, T, tr = c["A1"], c["A2"], c["P"], c["T"], c["tr"]
    g_and, g_or, g_not, g_ch = c["g"]
    w, hops, blocked = c["w"], c["hops"], c["blocked"]
    dlogit = (dy * c["y_chain"]).sum() * w * (1 - w)
    dP = np.matmul(dy, np.swapaxes(c["v1"], -1, -2))
    if c.get("drop") is not None:
        dP = dP * c["drop"]
    dv1 = np.matmul(np.swapaxes(c.get("Pd", P), -1, -2), dy)
    g = w * dy
    dA1 = np.matmul(g, np.swapaxes(tr[-1], -1, -2))
    gt = np.matmul(np.swapaxes(A1, -1, -2), g)
    dA2 = np.zeros_like(A2)
    for i in range(hops - 1, 0, -1):
        dA2 += np.matmul(gt, np.swapaxes(tr[i - 1], -1, -2))
        gt = np.matmul(np.swapaxes(A2, -1, -2), gt)
    dv2 = gt
    dSmix = P * (dP - (P * dP).sum(-1, keepdims=True))
    pi1 = np.exp(c["S1"] - c["lse"])
    pi2 = np.exp(c["S2"] - c["lse"])
    dS1 = dSmix * (1.0 - g_or + g_or * pi1)
    dS2 = dSmix * (g_and - g_not * c["beta"] + g_or * pi2)
    D = g_ch * dSmix / (c["C"] + EPS_CHAIN)
    for i in range(hops - 1, 0, -1):
        dA2 += np.matmul(np.swapaxes(T[i - 1], -1, -2), D)
        D = np.matmul(D, np.swapaxes(A2, -1, -2))
    dA1 += D
    dS1 = dS1 + A1 * (dA1 - (A1 * dA1).sum(-1, keepdims=True))
    dS2 = dS2 + A2 * (dA2 - (A2 * dA2).sum(-1, keepdims=True))
    if blocked is not None:
        dS1 = np.where(blocked, 0.0, dS1)
        dS2 = np.where(blocked, 0.0, dS2)
    dS1 *= c["scale"]
    dS2 *= c["scale"]
    return dict(dq1=np.matmul(dS1, c["k1"]), dk1=np.matmul(np.swapaxes(dS1, -1, -2), c["q1"]),
                dq2=np.matmul(dS2, c["k2"]), dk2=np.matmul(np.swapaxes(dS2, -1, -2), c["q2"]),
                dv1=dv1, dv2=dv2, dlogit=dlogit)


def module_fwd(x, params: Dict[str, np.ndarray], heads: int, gates: Optional[dict] = None,
               beta_not: float = 0.5, hops: int = 3, attn_mask=None):
    B, N, D = x.shape
    H, dk = heads, D // heads
    t1 = _heads(x @ params["qkv1.weight"].T, B, N, H, dk)        # :196
    t2 = _heads(x @ params["qkv2.weight"].T, B, N, H, dk)        # :197
    blocked = None if attn_mask is None else (np.asarray(attn_mask) == 0)
    y, cache = core_fwd(t1[0], t1[1], t1[2], t2[0], t2[1], t2[2], gates or DEFAULT_GATES,
                        beta_not, hops, params["chain_value_logit"], blocked)
    ycat = np.transpose(y, (0, 2, 1, 3)).reshape(B, N, D)
    out = ycat @ params["proj.weight"].T
    cache.update(x=x, ycat=ycat, params=params, H=H)
    return out, cache


def module_bwd(dout, cache):
    p, x, H = cache["params"], cache["x"], cache["H"]
    B, N, D = x.shape
    dk = D // H
    grads = {"proj.weight": np.einsum("bno,bni->oi", dout, cache["ycat"])}
    dy = np.transpose((dout @ p["proj.weight"]).reshape(B, N, H, dk), (0, 2, 1, 3))
    g = core_bwd(dy, cache)
    grads["chain_value_logit"] = np.asarray(g["dlogit"])

    def unheads(a, b, c_):
        return np.transpose(np.stack([a, b, c_]), (1, 3, 0, 2, 4)).reshape(B, N, 3 * D)

    d1 = unheads(g["dq1"], g["dk1"], g["dv1"])
    d2 = unheads(g["dq2"], g["dk2"], g["dv2"])
    grads["qkv1.weight"] = np.einsum("bno,bni->oi", d1, x)
    grads["qkv2.weight"] = np.einsum("bno,bni->oi", d2, x)
    dx = d1 @ p["qkv1.weight"] + d2 @ p["qkv2.weight"]
    return dx, grads
