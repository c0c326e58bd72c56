"""numpy oracle for the Quartet `CausalSelfAttention`  --  TEST INFRASTRUCTURE ONLY.

Restates reference `mop/models/quartet_attn_patch.py:75-127`:
row z-normalisation of two score maps over the FULL un-masked row with the
unbiased std (:95-98), mix ``(1-m) z1 + m z1 z2 * quartet_scale`` (:103-106),
causal mask (:112-113), additive attention_mask (:115-116), softmax, AV, o_proj.
`use_quartet=False` still z-normalises the single map (:108-110).
"""
from __future__ import annotations

import math

import numpy as np

from .edgewise import _sigmoid
from .multihop import _masked_softmax


def _znorm(S, eps):
    T = S.shape[-1]
    mu = S.mean(-1, keepdims=True)
    d = S - mu
    sd = np.sqrt((d * d).sum(-1, keepdims=True) / max(T - 1, 1))   # torch.std: unbiased
    return d / (sd + eps), sd


def _znorm_bwd(dz, z, sd, eps):
    """z = (S-mu)/(sd+eps), sd unbiased over T entries."""
    T = z.shape[-1]
    den = sd + eps
    # dS = (dz - mean(dz))/den  -  (S-mu) * sum(dz*(S-mu)) / ((T-1) sd den^2)
    d = z * den                                                   # S - mu
    t1 = (dz - dz.mean(-1, keepdims=True)) / den
    s = (dz * d).sum(-1, keepdims=True)
    t2 = d * s / (max(T - 1, 1) * np.maximum(sd, 1e-30) * den * den)
    return t1 - t2


def core_fwd(q, k, v, q2, k2, mixture, quartet_scale, eps=1e-5, use_quartet=True, causal=True,
             add_mask=None, drop=None):
    """q..: (B,H,T,dh). mixture/quartet_scale: scalars. add_mask: additive, broadcastable.
    drop: None or the dropout multiplier keep / (1 - p) per edge (`self.attn_dropout(att)` :119 with the mask made explicit)."""
    dh = q.shape[-1]
    scale = 1.0 / math.sqrt(dh)
    T = q.shape[-2]
    S1 = np.matmul(q, np.swapaxes(k, -1, -2)) * scale             # :88
    z1, sd1 = _znorm(S1, eps if use_quartet else 1e-5)
    c = dict(q=q, k=k, v=v, z1=z1, sd1=sd1, scale=scale, use_quartet=use_quartet, eps=eps)
    if use_quartet:
        S2 = np.matmul(q2, np.swapaxes(k2, -1, -2)) * scale       # :93
        z2, sd2 = _znorm(S2, eps)
        m = _sigmoid(mixture)
        scores = (1.0 - m) * z1 + m * (z1 * z2) * quartet_scale   # :104-106
        c.update(q2=q2, k2=k2, z2=z2, sd2=sd2, m=m, qs=quartet_scale)
    else:
        scores = z1                                               # :108-110
    blocked = None
    if causal:
        blocked = np.broadcast_to(~np.tril(np.ones((T, T), dtype=bool)), scores.shape)  # :112-113
    if add_mask is not None:
        scores = scores + add_mask                                # :115-116
    P = _masked_softmax(scores, blocked)
    Pd = P if drop is None else P * drop
    y = np.matmul(Pd, v)
    c.update(P=P, Pd=Pd, drop=drop)
    return y, c


def core_bwd(dy, c):
    P = c["P"]
    dP = np.matmul(dy, np.swapaxes(c["v"], -1, -2))
    if c.get("drop") is not None:
        dP = dP * c["drop"]
    dv = np.matmul(np.swapaxes(c.get("Pd", P), -1, -2), dy)
    dsc = P * (dP - (P * dP).sum(-1, keepdims=True))
    out = dict(dv=dv)
    if c["use_quartet"]:
        m, qs, z1, z2 = c["m"], c["qs"], c["z1"], c["z2"]
        dz1 = dsc * ((1.0 - m) + m * qs * z2)
        dz2 = dsc * (m * qs * z1)
        dm = (dsc * (-z1 + z1 * z2 * qs)).sum()
        out["dmixture"] = dm * m * (1.0 - m)
        out["dquartet_scale"] = (dsc * (m * z1 * z2)).sum()
        dS2 = _znorm_bwd(dz2, z2, c["sd2"], c["eps"]) * c["scale"]
        out["dq2"] = np.matmul(dS2, c["k2"])
        out["dk2"] = np.matmul(np.swapaxes(dS2, -1, -2), c["q2"])
        dS1 = _znorm_bwd(dz1, z1, c["sd1"], c["eps"]) * c["scale"]
    else:
        dS1 = _znorm_bwd(dsc, c["z1"], c["sd1"], 1e-5) * c["scale"]
    out["dq"] = np.matmul(dS1, c["k"])
    out["dk"] = np.matmul(np.swapaxes(dS1, -1, -2), c["q"])
    return out


def module_fwd(x, params, n_head, use_quartet=True, eps=1e-5, add_mask=None):
    """params keyed like the reference state_dict: {q,k,v,o,q2,k2}_proj.weight[/bias],
    mixture (1,), quartet_scale (1,)."""
    B, T, C = x.shape
    H, dh = n_head, C // n_head

    def lin(name):
        y = x @ params[f"{name}.weight"].T
        if f"{name}.bias" in params:
            y = y + params[f"{name}.bias"]
        return np.transpose(y.reshape(B, T, H, dh), (0, 2, 1, 3))  # :84-86

    q, k, v = lin("q_proj"), lin("k_proj"), lin("v_proj")
    if use_quartet:
        q2, k2 = lin("q2_proj"), lin("k2_proj")
        y, c = core_fwd(q, k, v, q2, k2, params["mixture"][0], params["quartet_scale"][0], eps,
                        True, True, add_mask)
    else:
        y, c = core_fwd(q, k, v, None, None, None, None, eps, False, True, add_mask)
    ycat = np.transpose(y, (0, 2, 1, 3)).reshape(B, T, C)         # :122
    out = ycat @ params["o_proj.weight"].T
    if "o_proj.bias" in params:
        out = out + params["o_proj.bias"]
    c.update(x=x, ycat=ycat, params=params, H=H)
    return out, c


def module_bwd(dout, c):
    p, x, H = c["params"], c["x"], c["H"]
    B, T, C = x.shape
    dh = C // H
    grads = {"o_proj.weight": np.einsum("bno,bni->oi", dout, c["ycat"])}
    if "o_proj.bias" in p:
        grads["o_proj.bias"] = dout.sum((0, 1))
    dy = np.transpose((dout @ p["o_proj.weight"]).reshape(B, T, H, dh), (0, 2, 1, 3))
    g = core_bwd(dy, c)
    dx = np.zeros_like(x)
    names = [("q_proj", "dq"), ("k_proj", "dk"), ("v_proj", "dv")]
    if c["use_quartet"]:
        names += [("q2_proj", "dq2"), ("k2_proj", "dk2")]
        grads["mixture"] = np.asarray([g["dmixture"]])
        grads["quartet_scale"] = np.asarray([g["dquartet_scale"]])
    for name, key in names:
        d = np.transpose(g[key], (0, 2, 1, 3)).reshape(B, T, C)
        grads[f"{name}.weight"] = np.einsum("bno,bni->oi", d, x)
        if f"{name}.bias" in p:
            grads[f"{name}.bias"] = d.sum((0, 1))
        dx = dx + d @ p[f"{name}.weight"]
    return dx, grads
