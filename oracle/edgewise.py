"""numpy oracle for EdgewiseMSA (low-rank gate head)  --  TEST INFRASTRUCTURE ONLY.

Restates reference `mop/models/attention_variants.py`:
  * EdgewiseGateHead low-rank branch            :273-309 (init), :319-331 (forward)
  * EdgewiseMSA.__init__ parameter set          :335-451
  * EdgewiseMSA.forward                         :453-564
and adds a hand-derived backward pass (the reference relies on autograd); the
backward is pinned against autograd gradients of the reference in
tests/golden/*.npz (tools/gen_golden.py).

Conventions
-----------
All tensors are numpy arrays, dtype taken from the inputs (float32 or float64).
`params` is a dict keyed exactly like the reference ``state_dict``:
  shared   : qkv.weight (3D,D), q_scale/k_scale/v_scale (V,H,1,dk)
  unshared : qkv_list.{i}.weight (3D,D)
  always   : proj.weight (D,D), chain_value_logit (),
             edge_head.row_proj.{weight (4r,C,1), bias (4r)}, edge_head.col_proj.{...}
Only `attn_mask=None`, dropout p=0, no lens banks (the configuration every
BASELINE.json config uses; see SURVEY.md section 8a note on masked NaNs).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

EPS_CHAIN = 1e-6  # attention_variants.py:516


def _softmax(x: np.ndarray, axis: int = -1) -> np.ndarray:
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _logsumexp(x: np.ndarray, axis: int) -> np.ndarray:
    m = x.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(axis=axis, keepdims=True))).squeeze(axis)


# ----------------------------------------------------------------------------
# attention core: everything between the qkv projection and the out projection
# ----------------------------------------------------------------------------
def core_fwd(qv, kv, v0, vL, Wr, br, Wc, bc, beta_not, chain_logit):
    """EdgewiseMSA.forward :500-562 for per-view queries/keys.

    qv, kv : (V,B,H,N,dk)  per-view q_i, k_i   (:461-470)
    v0, vL : (B,H,N,dk)    vs[0] and vs[v_idx_last] (:553-557)
    Wr, Wc : (4r, C) with C = 2V+2 ; br, bc : (4r,)
    returns y (B,H,N,dk) and a cache for core_bwd.
    """
    V, B, H, N, dk = qv.shape
    scale = 1.0 / math.sqrt(dk)
    S = np.matmul(qv, np.swapaxes(kv, -1, -2)) * scale          # :500-503  (V,B,H,N,N)
    A = _softmax(S, -1)                                           # :507
    T = [A[0]]                                                    # :508-512 prefix products
    for i in range(1, V):
        T.append(np.matmul(T[-1], A[i]))
    U = [A[V - 1]]                                                # :513-515
    for i in range(V - 2, -1, -1):
        U.append(np.matmul(U[-1], A[i]))
    Cf, Cb = T[-1], U[-1]
    Cr = np.log(Cf + EPS_CHAIN)                                   # :520
    Cl = np.log(Cb + EPS_CHAIN)                                   # :521
    # feature stack [S_v, S_v^T, Cr, Cl] (:522) is only consumed through its
    # row/col means (:323-324); mean over rows of S_v^T == col-mean of S_v.
    rS = S.mean(-1)                                               # (V,B,H,N)
    cS = S.mean(-2)
    row_feat = np.concatenate(
        [np.moveaxis(rS, 0, 2), np.moveaxis(cS, 0, 2),
         Cr.mean(-1)[:, :, None], Cl.mean(-1)[:, :, None]], axis=2)   # (B,H,C,N)
    col_feat = np.concatenate(
        [np.moveaxis(cS, 0, 2), np.moveaxis(rS, 0, 2),
         Cr.mean(-2)[:, :, None], Cl.mean(-2)[:, :, None]], axis=2)
    a = np.einsum("oc,bhcn->bhon", Wr, row_feat) + br[None, None, :, None]   # :325
    b = np.einsum("oc,bhcn->bhon", Wc, col_feat) + bc[None, None, :, None]   # :326
    r = Wr.shape[0] // 4
    a4 = a.reshape(B, H, 4, r, N)
    b4 = b.reshape(B, H, 4, r, N)
    Z = np.einsum("bhgkn,bhgkm->bhgnm", a4, b4)                  # :330
    G = _sigmoid(Z)                                               # :331
    S0 = S[0]
    Ssum = S.sum(0)                                               # :538-540
    lse = _logsumexp(S, 0)                                        # :541
    O = Ssum - S0
    nb = beta_not / max(1, V - 1)                                 # :542,546
    Smix = S0 + G[:, :, 0] * O + G[:, :, 1] * (lse - S0) - G[:, :, 2] * (nb * O) \
        + G[:, :, 3] * Cr                                         # :543-547
    P = _softmax(Smix, -1)                                        # :551
    y_base = np.matmul(P, v0)                                     # :554
    t = [None] * V                                                # :557-560 value transport
    t[V - 1] = vL
    for i in range(V - 1, 0, -1):
        t[i - 1] = np.matmul(A[i], t[i])
    y_chain = np.matmul(A[0], t[0])
    w = _sigmoid(chain_logit)                                     # :561
    y = y_base + w * y_chain                                      # :562
    cache = dict(qv=qv, kv=kv, v0=v0, vL=vL, Wr=Wr, Wc=Wc, S=S, A=A, T=T, U=U, Cf=Cf, Cb=Cb,
                 Cr=Cr, Cl=Cl, row_feat=row_feat, col_feat=col_feat, a4=a4, b4=b4, G=G,
                 lse=lse, O=O, nb=nb, P=P, t=t, y_chain=y_chain, w=w, scale=scale,
                 Smix=Smix, a=a, b=b, y_base=y_base)
    return y, cache


def core_bwd(dy, c):
    """Hand-derived gradient of core_fwd (the reference uses autograd)."""
    qv, kv, v0, vL = c["qv"], c["kv"], c["v0"], c["vL"]
    S, A, T, U, P, G = c["S"], c["A"], c["T"], c["U"], c["P"], c["G"]
    V, B, H, N, dk = qv.shape
    w = c["w"]
    # y = P v0 + w * A0 t0
    dlogit = (dy * c["y_chain"]).sum() * w * (1.0 - w)
    dP = np.matmul(dy, np.swapaxes(v0, -1, -2))
    dv0 = np.matmul(np.swapaxes(P, -1, -2), dy)
    dA = np.zeros_like(A)
    g = w * dy                                                    # grad wrt A0 t0
    t = c["t"]
    dA[0] += np.matmul(g, np.swapaxes(t[0], -1, -2))
    g = np.matmul(np.swapaxes(A[0], -1, -2), g)                   # grad wrt t0
    for i in range(1, V):
        dA[i] += np.matmul(g, np.swapaxes(t[i], -1, -2))
        g = np.matmul(np.swapaxes(A[i], -1, -2), g)
    dvL = g
    # softmax backward of the mixed logits
    dSmix = P * (dP - (P * dP).sum(-1, keepdims=True))
    S0, O, lse, nb, Cr = S[0], c["O"], c["lse"], c["nb"], c["Cr"]
    dG = np.stack([dSmix * O, dSmix * (lse - S0), -dSmix * (nb * O), dSmix * Cr], axis=2)
    dZ = dG * G * (1.0 - G)                                       # (B,H,4,N,N)
    a4, b4 = c["a4"], c["b4"]
    da4 = np.einsum("bhgnm,bhgkm->bhgkn", dZ, b4)
    db4 = np.einsum("bhgnm,bhgkn->bhgkm", dZ, a4)
    r = a4.shape[3]
    da = da4.reshape(B, H, 4 * r, N)
    db = db4.reshape(B, H, 4 * r, N)
    dWr = np.einsum("bhon,bhcn->oc", da, c["row_feat"])
    dWc = np.einsum("bhon,bhcn->oc", db, c["col_feat"])
    dbr = da.sum((0, 1, 3))
    dbc = db.sum((0, 1, 3))
    drow = np.einsum("oc,bhon->bhcn", c["Wr"], da)                # (B,H,C,N)
    dcol = np.einsum("oc,bhon->bhcn", c["Wc"], db)
    drS = np.moveaxis(drow[:, :, 0:V] + dcol[:, :, V:2 * V], 2, 0)       # (V,B,H,N)
    dcS = np.moveaxis(drow[:, :, V:2 * V] + dcol[:, :, 0:V], 2, 0)
    # logits: direct terms + mean terms
    pi = np.exp(S - lse[None])                                    # softmax over views
    G0, G1, G2, G3 = G[:, :, 0], G[:, :, 1], G[:, :, 2], G[:, :, 3]
    dS = np.empty_like(S)
    dS[0] = dSmix * (1.0 - G1 + G1 * pi[0])
    for v_ in range(1, V):
        dS[v_] = dSmix * (G0 - nb * G2 + G1 * pi[v_])
    dS += drS[..., :, None] / N + dcS[..., None, :] / N
    dCr = dSmix * G3 + drow[:, :, 2 * V][..., :, None] / N + dcol[:, :, 2 * V][..., None, :] / N
    dCl = drow[:, :, 2 * V + 1][..., :, None] / N + dcol[:, :, 2 * V + 1][..., None, :] / N
    dCf = dCr / (c["Cf"] + EPS_CHAIN)
    dCb = dCl / (c["Cb"] + EPS_CHAIN)
    # forward chain T_m = T_{m-1} A_m
    D = dCf
    for m in range(V - 1, 0, -1):
        dA[m] += np.matmul(np.swapaxes(T[m - 1], -1, -2), D)
        D = np.matmul(D, np.swapaxes(A[m], -1, -2))
    dA[0] += D
    # backward chain U_m = U_{m-1} A_{V-1-m}
    D = dCb
    for m in range(V - 1, 0, -1):
        dA[V - 1 - m] += np.matmul(np.swapaxes(U[m - 1], -1, -2), D)
        D = np.matmul(D, np.swapaxes(A[V - 1 - m], -1, -2))
    dA[V - 1] += D
    dS += A * (dA - (A * dA).sum(-1, keepdims=True))
    dS *= c["scale"]
    dqv = np.matmul(dS, kv)
    dkv = np.matmul(np.swapaxes(dS, -1, -2), qv)
    return dict(dqv=dqv, dkv=dkv, dv0=dv0, dvL=dvL, dWr=dWr, dbr=dbr, dWc=dWc, dbc=dbc,
                dlogit=dlogit)


# ----------------------------------------------------------------------------
# module level: projections + view construction (:458-470, :563-564)
# ----------------------------------------------------------------------------
def _heads(t, B, N, H, dk):
    # (B,N,3,H,dk) -> (3,B,H,N,dk)   attention_variants.py:459
    return np.transpose(t.reshape(B, N, 3, H, dk), (2, 0, 3, 1, 4))


def module_fwd(x, params: Dict[str, np.ndarray], heads: int, n_views: int, share_qkv: bool,
               beta_not: float = 0.5):
    B, N, D = x.shape
    H, dk = heads, D // heads
    V = max(2, int(n_views))                                      # :362
    if share_qkv:
        qkv = _heads(x @ params["qkv.weight"].T, B, N, H, dk)
        qb, kb, vb = qkv[0], qkv[1], qkv[2]
        qs = params["q_scale"][:, None]                           # (V,1,H,1,dk)
        ks = params["k_scale"][:, None]
        vs = params["v_scale"][:, None]
        qv, kv = qb[None] * qs, kb[None] * ks                     # :462-463
        v0, vL = vb * vs[0], vb * vs[V - 1]                       # :464, :556-557
    else:
        qkvs = [_heads(x @ params[f"qkv_list.{i}.weight"].T, B, N, H, dk) for i in range(V)]
        qv = np.stack([t[0] for t in qkvs])
        kv = np.stack([t[1] for t in qkvs])
        v0, vL = qkvs[0][2], qkvs[V - 1][2]
    Wr = params["edge_head.row_proj.weight"][:, :, 0]
    Wc = params["edge_head.col_proj.weight"][:, :, 0]
    y, cache = core_fwd(qv, kv, v0, vL, Wr, params["edge_head.row_proj.bias"], Wc,
                        params["edge_head.col_proj.bias"], beta_not,
                        params["chain_value_logit"])
    ycat = np.transpose(y, (0, 2, 1, 3)).reshape(B, N, D)         # :563
    out = ycat @ params["proj.weight"].T                          # :564
    cache.update(x=x, ycat=ycat, share=share_qkv, H=H, V=V, params=params)
    if share_qkv:
        cache.update(qb=qb, kb=kb, vb=vb)
    return out, cache


def module_bwd(dout, cache) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    p = cache["params"]
    x = cache["x"]
    B, N, D = x.shape
    H, V = cache["H"], cache["V"]
    dk = D // H
    grads: Dict[str, np.ndarray] = {}
    grads["proj.weight"] = np.einsum("bno,bni->oi", dout, cache["ycat"])
    dycat = dout @ p["proj.weight"]
    dy = np.transpose(dycat.reshape(B, N, H, dk), (0, 2, 1, 3))
    g = core_bwd(dy, cache)
    grads["chain_value_logit"] = np.asarray(g["dlogit"])
    grads["edge_head.row_proj.weight"] = g["dWr"][:, :, None]
    grads["edge_head.row_proj.bias"] = g["dbr"]
    grads["edge_head.col_proj.weight"] = g["dWc"][:, :, None]
    grads["edge_head.col_proj.bias"] = g["dbc"]

    def unheads(dq, dk_, dv):  # (B,H,N,dk)x3 -> (B,N,3D)
        t = np.stack([dq, dk_, dv])                               # (3,B,H,N,dk)
        return np.transpose(t, (1, 3, 0, 2, 4)).reshape(B, N, 3 * D)

    if cache["share"]:
        qb, kb, vb = cache["qb"], cache["kb"], cache["vb"]
        qs, ks, vs = p["q_scale"], p["k_scale"], p["v_scale"]
        dqv, dkv = g["dqv"], g["dkv"]
        grads["q_scale"] = (dqv * qb[None]).sum((1, 3))[:, :, None, :]
        grads["k_scale"] = (dkv * kb[None]).sum((1, 3))[:, :, None, :]
        dvs = np.zeros_like(vs)
        dvs[0] += (g["dv0"] * vb).sum((0, 2))[:, None, :]
        dvs[V - 1] += (g["dvL"] * vb).sum((0, 2))[:, None, :]
        grads["v_scale"] = dvs
        dqb = (dqv * qs[:, None]).sum(0)
        dkb = (dkv * ks[:, None]).sum(0)
        dvb = g["dv0"] * vs[0][None] + g["dvL"] * vs[V - 1][None]
        dqkv = unheads(dqb, dkb, dvb)
        grads["qkv.weight"] = np.einsum("bno,bni->oi", dqkv, x)
        dx = dqkv @ p["qkv.weight"]
    else:
        dx = np.zeros_like(x)
        zero = np.zeros_like(g["dv0"])
        for i in range(V):
            dv = zero
            if i == 0:
                dv = dv + g["dv0"]
            if i == V - 1:
                dv = dv + g["dvL"]
            dqkv = unheads(g["dqv"][i], g["dkv"][i], dv)
            grads[f"qkv_list.{i}.weight"] = np.einsum("bno,bni->oi", dqkv, x)
            dx = dx + dqkv @ p[f"qkv_list.{i}.weight"]
    return dx, grads
