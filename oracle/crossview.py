"""numpy oracle for CrossViewMixerMSA  --  TEST INFRASTRUCTURE ONLY.

Restates reference `mop/models/attention_variants.py`:
  * `CrossViewMixerMSA._compute_logits`   :90-110  (four score maps, 2x2 mix, transpose cues)
  * `CrossViewMixerMSA.forward`           :120-156 (mask, softmax, optional per-key prior sharpening)
with a hand-derived backward pinned against the reference's autograd (tests/golden/cv_*.npz).
params: qkv1.weight, qkv2.weight (3D,D), proj.weight (D,D), mix (2,2).
"""
from __future__ import annotations

import math

import numpy as np

from .edgewise import _heads
from .multihop import _masked_softmax


def anchor_rows(A2, mode: str, fixed_k_star: int):
    """k_star (B,H) of :131-145.  `argmax_row_sum` takes the argmax over row sums of a row-stochastic matrix, i.e. it is
    decided by rounding noise; callers that need the reference's choice pass it explicitly (k_star=...)."""
    B, H, N, _ = A2.shape
    if mode == "fixed":
        return np.full((B, H), max(0, min(N - 1, fixed_k_star)), dtype=np.int64)
    if mode == "argmax_row_sum":
        return A2.sum(-1).argmax(-1)
    return np.zeros((B, H), dtype=np.int64)


def core_fwd(q1, k1, v1, q2, k2, mix, t1=0.0, t2=0.0, blocked=None, prior_weight=0.0, anchor_mode="argmax_row_sum",
             fixed_k_star=0, k_star=None):
    dk = q1.shape[-1]
    scale = 1.0 / math.sqrt(dk)
    tr = lambda a: np.swapaxes(a, -1, -2)
    S1 = np.matmul(q1, tr(k1)) * scale                            # :99-102
    S2 = np.matmul(q2, tr(k2)) * scale
    S12 = np.matmul(q1, tr(k2)) * scale
    S21 = np.matmul(q2, tr(k1)) * scale
    S = mix[0, 0] * S1 + mix[0, 1] * S12 + mix[1, 0] * S21 + mix[1, 1] * S2     # :105
    S = S + t1 * tr(S1) + t2 * tr(S2)                             # :106-110 (t = 0 when cues are off)
    if blocked is not None:
        blocked = np.broadcast_to(blocked, S.shape)
    P = _masked_softmax(S, blocked)                               # :124-125
    c = dict(q1=q1, k1=k1, v1=v1, q2=q2, k2=k2, mix=mix, t1=t1, t2=t2, S1=S1, S2=S2, S12=S12, S21=S21, P=P, scale=scale,
             pw=prior_weight, blocked=blocked)
    A = P
    if prior_weight > 0.0:                                        # :126-150
        A1 = _masked_softmax(S1, blocked)
        A2 = _masked_softmax(S2, blocked)
        if k_star is None:
            k_star = anchor_rows(A2, anchor_mode, fixed_k_star)
        anc = np.take_along_axis(A2, k_star[:, :, None, None].repeat(A2.shape[-1], -1), axis=2)   # (B,H,1,N)   :147
        u = A1 * anc
        s = u.sum(-1, keepdims=True) + 1e-9
        As = u / s                                                # :149
        A = (1.0 - prior_weight) * P + prior_weight * As          # :150
        c.update(A1=A1, A2=A2, anc=anc, s=s, As=As, k_star=k_star)
    y = np.matmul(A, v1)
    c["A"] = A
    return y, c


def _softmax_bwd(P, dP):
    return P * (dP - (P * dP).sum(-1, keepdims=True))


def core_bwd(dy, c):
    tr = lambda a: np.swapaxes(a, -1, -2)
    A, P, pw, mix = c["A"], c["P"], c["pw"], c["mix"]
    dA = np.matmul(dy, tr(c["v1"]))
    dv1 = np.matmul(tr(A), dy)
    dS1 = np.zeros_like(P)
    dS2 = np.zeros_like(P)
    dPm = dA
    if pw > 0.0:
        dPm = (1.0 - pw) * dA
        dAs = pw * dA
        As, A1, A2, anc, s = c["As"], c["A1"], c["A2"], c["anc"], c["s"]
        du = (dAs - (dAs * As).sum(-1, keepdims=True)) / s
        dA1 = du * anc
        danc = (du * A1).sum(-2, keepdims=True)                   # (B,H,1,N)
        dA2 = np.zeros_like(A2)
        np.put_along_axis(dA2, c["k_star"][:, :, None, None].repeat(A2.shape[-1], -1), danc, axis=2)
        dS1 += _softmax_bwd(A1, dA1)
        dS2 += _softmax_bwd(A2, dA2)
    dS = _softmax_bwd(P, dPm)
    dmix = np.array([[(dS * c["S1"]).sum(), (dS * c["S12"]).sum()], [(dS * c["S21"]).sum(), (dS * c["S2"]).sum()]])
    dS1 += mix[0, 0] * dS + c["t1"] * tr(dS)
    dS2 += mix[1, 1] * dS + c["t2"] * tr(dS)
    dS12, dS21 = mix[0, 1] * dS, mix[1, 0] * dS
    sc = c["scale"]
    q1, k1, q2, k2 = c["q1"], c["k1"], c["q2"], c["k2"]
    return dict(dq1=(np.matmul(dS1, k1) + np.matmul(dS12, k2)) * sc, dk1=(np.matmul(tr(dS1), q1) + np.matmul(tr(dS21), q2)) * sc,
                dq2=(np.matmul(dS2, k2) + np.matmul(dS21, k1)) * sc, dk2=(np.matmul(tr(dS2), q2) + np.matmul(tr(dS12), q1)) * sc,
                dv1=dv1, dmix=dmix.astype(P.dtype))


def module_fwd(x, params, heads, attn_mask=None, use_transpose_cues=True, t1=0.0, t2=0.0, enable_per_key_prior=False,
               prior_weight=0.5, anchor_mode="argmax_row_sum", fixed_k_star=0, k_star=None):
    B, N, D = x.shape
    H, dk = heads, D // heads
    a = _heads(x @ params["qkv1.weight"].T, B, N, H, dk)
    b = _heads(x @ params["qkv2.weight"].T, B, N, H, dk)
    blocked = None if attn_mask is None else (np.asarray(attn_mask) == 0)
    pw = prior_weight if (enable_per_key_prior and prior_weight > 0.0) else 0.0
    y, c = core_fwd(a[0], a[1], a[2], b[0], b[1], params["mix"], t1 if use_transpose_cues else 0.0,
                    t2 if use_transpose_cues else 0.0, blocked, pw, anchor_mode, fixed_k_star, k_star)
    ycat = np.transpose(y, (0, 2, 1, 3)).reshape(B, N, D)
    c.update(x=x, ycat=ycat, params=params, H=H)
    return ycat @ params["proj.weight"].T, c


def module_bwd(dout, c):
    p, x, H = c["params"], c["x"], c["H"]
    B, N, D = x.shape
    dk = D // H
    grads = {"proj.weight": np.einsum("bno,bni->oi", dout, c["ycat"])}
    dy = np.transpose((dout @ p["proj.weight"]).reshape(B, N, H, dk), (0, 2, 1, 3))
    g = core_bwd(dy, c)
    pack = lambda q, k, v: np.transpose(np.stack([q, k, v]), (1, 3, 0, 2, 4)).reshape(B, N, 3 * D)
    d1 = pack(g["dq1"], g["dk1"], g["dv1"])
    d2 = pack(g["dq2"], g["dk2"], np.zeros_like(g["dv1"]))       # v2 is unused (:98)
    grads["qkv1.weight"] = np.einsum("bno,bni->oi", d1, x)
    grads["qkv2.weight"] = np.einsum("bno,bni->oi", d2, x)
    grads["mix"] = g["dmix"]
    return d1 @ p["qkv1.weight"] + d2 @ p["qkv2.weight"], grads
