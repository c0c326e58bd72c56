"""numpy oracle for EdgewiseMSA (low-rank and dense gate heads, lens banks)  --  TEST INFRASTRUCTURE ONLY.

Restates reference `mop/models/attention_variants.py`:
  * EdgewiseGateHead low-rank branch            :273-309 (init), :319-331 (forward)
  * EdgewiseGateHead dense branch (+ use_k3)    :250-272 (init), :312-318 (forward)
  * S lens bank (depthwise dilated 3x3 over the score planes)   :425-442, :523-533
  * Q/K lens bank (depthwise dilated conv over tokens)          :392-423, :472-498
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
             dense head: edge_head.conv1.{weight (16,C,1,1), bias}, edge_head.mid3.{weight (16,16,3,3), bias}
             (use_k3), edge_head.conv2.{weight (4,16,1,1), bias};
             lens banks: lens_bank.{l}.weight (S,1,3,3), q_lens.{l}.weight / k_lens.{l}.weight (dk,1,k)
Only `attn_mask=None`, dropout p=0 (see SURVEY.md section 8a note on masked NaNs).
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


_K0 = math.sqrt(2.0 / math.pi)


def _gelu_tanh(x):
    """nn.GELU(approximate="tanh")   attention_variants.py:252"""
    return 0.5 * x * (1.0 + np.tanh(_K0 * (x + 0.044715 * x ** 3)))


def _gelu_tanh_grad(x):
    t = np.tanh(_K0 * (x + 0.044715 * x ** 3))
    return 0.5 * (1.0 + t) + 0.5 * x * (1.0 - t * t) * _K0 * (1.0 + 3 * 0.044715 * x * x)


def _shift2(x, di, dj):
    """y[..., i, j] = x[..., i+di, j+dj], zero outside (zero-padded cross-correlation tap)."""
    N, M = x.shape[-2], x.shape[-1]
    y = np.zeros_like(x)
    i0, i1 = max(0, -di), min(N, N - di)
    j0, j1 = max(0, -dj), min(M, M - dj)
    if i0 < i1 and j0 < j1:
        y[..., i0:i1, j0:j1] = x[..., i0 + di:i1 + di, j0 + dj:j1 + dj]
    return y


def lens_fwd(S, lens_w, dils):
    """S lens bank :523-531.  S (V,B,H,N,N); lens_w (L,V,3,3) depthwise taps; dilation = padding = dils[l].
    returns (L,V,B,H,N,N):  out[i,j] = sum_ab w[a,b] S[i+(a-1)d, j+(b-1)d]."""
    L = len(dils)
    out = np.zeros((L,) + S.shape, dtype=S.dtype)
    for l, d in enumerate(dils):
        for a in range(3):
            for b in range(3):
                out[l] += lens_w[l][:, a, b][:, None, None, None, None] * _shift2(S, (a - 1) * d, (b - 1) * d)
    return out


def lens_bwd(dLz, S, lens_w, dils):
    """gradient of lens_fwd wrt S and the taps."""
    dS = np.zeros_like(S)
    dw = np.zeros_like(lens_w)
    for l, d in enumerate(dils):
        for a in range(3):
            for b in range(3):
                dS += lens_w[l][:, a, b][:, None, None, None, None] * _shift2(dLz[l], -(a - 1) * d, -(b - 1) * d)
                dw[l][:, a, b] = (dLz[l] * _shift2(S, (a - 1) * d, (b - 1) * d)).sum((1, 2, 3, 4))
    return dS, dw


def dense_head_fwd(feat, hp):
    """EdgewiseGateHead dense branch :312-318.  feat (B,H,C,N,N); hp: W1 (16,C), b1, [W3 (16,16,3,3), b3], W2 (4,16), b2."""
    x1 = np.einsum("oc,bhcij->bhoij", hp["W1"], feat) + hp["b1"][None, None, :, None, None]
    h = _gelu_tanh(x1)
    c = dict(feat=feat, x1=x1, h=h)
    last = h
    if "W3" in hp:
        h2 = _gelu_tanh(h)                                        # `self.mid3(self.act(x))` :315-316: GELU applied twice
        x3 = np.zeros_like(h2)
        for a in range(3):
            for b in range(3):
                x3 += np.einsum("oc,bhcij->bhoij", hp["W3"][:, :, a, b], _shift2(h2, a - 1, b - 1))
        x3 += hp["b3"][None, None, :, None, None]
        c.update(h2=h2, x3=x3)
        last = x3
    Z = np.einsum("go,bhoij->bhgij", hp["W2"], last) + hp["b2"][None, None, :, None, None]
    c["last"] = last
    return _sigmoid(Z), c


def dense_head_bwd(dZ, c, hp):
    g = dict(dW2=np.einsum("bhgij,bhoij->go", dZ, c["last"]), db2=dZ.sum((0, 1, 3, 4)))
    dlast = np.einsum("go,bhgij->bhoij", hp["W2"], dZ)
    if "W3" in hp:
        h2 = c["h2"]
        dW3 = np.zeros_like(hp["W3"])
        dh2 = np.zeros_like(h2)
        for a in range(3):
            for b in range(3):
                dW3[:, :, a, b] = np.einsum("bhoij,bhcij->oc", dlast, _shift2(h2, a - 1, b - 1))
                dh2 += np.einsum("oc,bhoij->bhcij", hp["W3"][:, :, a, b], _shift2(dlast, -(a - 1), -(b - 1)))
        g.update(dW3=dW3, db3=dlast.sum((0, 1, 3, 4)))
        dh = dh2 * _gelu_tanh_grad(c["h"])
    else:
        dh = dlast
    dx1 = dh * _gelu_tanh_grad(c["x1"])
    g.update(dW1=np.einsum("bhoij,bhcij->oc", dx1, c["feat"]), db1=dx1.sum((0, 1, 3, 4)))
    g["dfeat"] = np.einsum("oc,bhoij->bhcij", hp["W1"], dx1)
    return g


# ----------------------------------------------------------------------------
# attention core: everything between the qkv projection and the out projection
# ----------------------------------------------------------------------------
def core_fwd(qv, kv, v0, vL, Wr, br, Wc, bc, beta_not, chain_logit, dense=None, lens=None, drop=None, blocked=None):
    """EdgewiseMSA.forward :500-562 for per-view queries/keys.

    qv, kv : (V,B,H,N,dk)  per-view q_i, k_i   (:461-470)
    v0, vL : (B,H,N,dk)    vs[0] and vs[v_idx_last] (:553-557)
    Wr, Wc : (4r, C) with C = 2V+2 ; br, bc : (4r,)
    dense  : None (low-rank head, Wr.. used) or dict W1,b1,[W3,b3],W2,b2 (dense head, Wr.. ignored)
    lens   : None or (lens_w (L,V,3,3), dils) -- extra feature channels l*V+v after [S, S^T, Cr, Cl]  (:533)
    blocked: None or bool (.., N, N), True = masked edge.  EXTENSION, not reference behaviour: the reference fills the scores with
             -inf before they enter the feature stack (:504-506 feed :518-546), which makes every output NaN for any blocking mask
             (SURVEY.md 8a note).  Here the mask acts on the attention probabilities only -- the per-view softmaxes (:507) and the final
             one (:549-551) -- while the gate features (scores, their means, lse, chain logs) see the unmasked finite scores.
    returns y (B,H,N,dk) and a cache for core_bwd.
    """
    V, B, H, N, dk = qv.shape
    scale = 1.0 / math.sqrt(dk)
    S = np.matmul(qv, np.swapaxes(kv, -1, -2)) * scale          # :500-503  (V,B,H,N,N)
    if blocked is not None:
        blocked = np.broadcast_to(np.asarray(blocked, dtype=bool), S.shape[1:])
        Sm = np.where(blocked[None], -np.inf, S)
        e = np.where(blocked[None], 0.0, np.exp(Sm - Sm.max(-1, keepdims=True)))
        A = e / e.sum(-1, keepdims=True)                          # :504-507
    else:
        A = _softmax(S, -1)                                       # :507
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
    Lz = lens_fwd(S, lens[0], lens[1]) if lens is not None else None     # (L,V,B,H,N,N)
    hc = None
    row_feat = col_feat = a4 = b4 = a = b = None
    if dense is not None:
        chans = [S[v] for v in range(V)] + [np.swapaxes(S[v], -1, -2) for v in range(V)] + [Cr, Cl]
        if Lz is not None:
            chans += [Lz[l, v] for l in range(Lz.shape[0]) for v in range(V)]
        feat = np.stack(chans, axis=2)                            # (B,H,C,N,N)  :534
        G, hc = dense_head_fwd(feat, dense)
    else:
        rS = S.mean(-1)                                           # (V,B,H,N)
        cS = S.mean(-2)
        rows = [np.moveaxis(rS, 0, 2), np.moveaxis(cS, 0, 2), Cr.mean(-1)[:, :, None], Cl.mean(-1)[:, :, None]]
        cols = [np.moveaxis(cS, 0, 2), np.moveaxis(rS, 0, 2), Cr.mean(-2)[:, :, None], Cl.mean(-2)[:, :, None]]
        if Lz is not None:
            Lf = Lz.reshape((-1,) + Lz.shape[2:])                 # (L*V,B,H,N,N), channel l*V+v
            rows.append(np.moveaxis(Lf.mean(-1), 0, 2))
            cols.append(np.moveaxis(Lf.mean(-2), 0, 2))
        row_feat = np.concatenate(rows, axis=2)                   # (B,H,C,N)
        col_feat = np.concatenate(cols, axis=2)
        a = np.einsum("oc,bhcn->bhon", Wr, row_feat) + br[None, None, :, None]   # :325
        b = np.einsum("oc,bhcn->bhon", Wc, col_feat) + bc[None, None, :, None]   # :326
        r = Wr.shape[0] // 4
        a4 = a.reshape(B, H, 4, r, N)
        b4 = b.reshape(B, H, 4, r, N)
        Z = np.einsum("bhgkn,bhgkm->bhgnm", a4, b4)              # :330
        G = _sigmoid(Z)                                           # :331
    S0 = S[0]
    Ssum = S.sum(0)                                               # :538-540
    lse = _logsumexp(S, 0)                                        # :541
    O = Ssum - S0
    nb = beta_not / max(1, V - 1)                                 # :542,546
    Smix = S0 + G[:, :, 0] * O + G[:, :, 1] * (lse - S0) - G[:, :, 2] * (nb * O) \
        + G[:, :, 3] * Cr                                         # :543-547
    if blocked is not None:
        Sx = np.where(blocked, -np.inf, Smix)                     # :549-550
        e = np.where(blocked, 0.0, np.exp(Sx - Sx.max(-1, keepdims=True)))
        P = e / e.sum(-1, keepdims=True)
    else:
        P = _softmax(Smix, -1)                                    # :551
    Pd = P if drop is None else P * drop                          # :552 attn_drop with the mask made explicit: drop = keep / (1 - p)
    y_base = np.matmul(Pd, v0)                                    # :554
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
                 Smix=Smix, a=a, b=b, y_base=y_base, dense=dense, lens=lens, hc=hc, Lz=Lz, Pd=Pd, drop=drop)
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
    if c.get("drop") is not None:
        dP = dP * c["drop"]
    dv0 = np.matmul(np.swapaxes(c.get("Pd", P), -1, -2), dy)
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
    dense, lens = c["dense"], c["lens"]
    out = {}
    dLz = None
    if dense is not None:
        hg = dense_head_bwd(dZ, c["hc"], dense)
        dfeat = hg.pop("dfeat")                                   # (B,H,C,N,N)
        out.update(hg)
        dS_feat = np.stack([dfeat[:, :, v_] + np.swapaxes(dfeat[:, :, V + v_], -1, -2) for v_ in range(V)])
        dCr_feat, dCl_feat = dfeat[:, :, 2 * V], dfeat[:, :, 2 * V + 1]
        if lens is not None:
            L = len(lens[1])
            dLz = np.stack([np.stack([dfeat[:, :, 2 * V + 2 + l * V + v_] for v_ in range(V)]) for l in range(L)])
    else:
        a4, b4 = c["a4"], c["b4"]
        da4 = np.einsum("bhgnm,bhgkm->bhgkn", dZ, b4)
        db4 = np.einsum("bhgnm,bhgkn->bhgkm", dZ, a4)
        r = a4.shape[3]
        da = da4.reshape(B, H, 4 * r, N)
        db = db4.reshape(B, H, 4 * r, N)
        out["dWr"] = np.einsum("bhon,bhcn->oc", da, c["row_feat"])
        out["dWc"] = np.einsum("bhon,bhcn->oc", db, c["col_feat"])
        out["dbr"] = da.sum((0, 1, 3))
        out["dbc"] = db.sum((0, 1, 3))
        drow = np.einsum("oc,bhon->bhcn", c["Wr"], da)            # (B,H,C,N)
        dcol = np.einsum("oc,bhon->bhcn", c["Wc"], db)
        drS = np.moveaxis(drow[:, :, 0:V] + dcol[:, :, V:2 * V], 2, 0)   # (V,B,H,N)
        dcS = np.moveaxis(drow[:, :, V:2 * V] + dcol[:, :, 0:V], 2, 0)
        dS_feat = drS[..., :, None] / N + dcS[..., None, :] / N
        dCr_feat = drow[:, :, 2 * V][..., :, None] / N + dcol[:, :, 2 * V][..., None, :] / N
        dCl_feat = drow[:, :, 2 * V + 1][..., :, None] / N + dcol[:, :, 2 * V + 1][..., None, :] / N
        if lens is not None:
            L = len(lens[1])
            dl = (drow[:, :, 2 * V + 2:][..., :, None] + dcol[:, :, 2 * V + 2:][..., None, :]) / N   # (B,H,L*V,N,N)
            dLz = np.moveaxis(dl, 2, 0).reshape((L, V) + dl.shape[:2] + dl.shape[3:])
    # logits: direct terms + mean terms
    pi = np.exp(S - lse[None])                                    # softmax over views
    G0, G1, G2, G3 = G[:, :, 0], G[:, :, 1], G[:, :, 2], G[:, :, 3]
    dS = np.empty_like(S)
    dS[0] = dSmix * (1.0 - G1 + G1 * pi[0])
    for v_ in range(1, V):
        dS[v_] = dSmix * (G0 - nb * G2 + G1 * pi[v_])
    dS += dS_feat
    if dLz is not None:
        dS_l, out["dlens_w"] = lens_bwd(dLz, S, lens[0], lens[1])
        dS += dS_l
    dCr = dSmix * G3 + dCr_feat
    dCl = dCl_feat
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
    out.update(dqv=dqv, dkv=dkv, dv0=dv0, dvL=dvL, dlogit=dlogit)
    return out


# ----------------------------------------------------------------------------
# module level: projections + view construction (:458-470, :563-564)
# ----------------------------------------------------------------------------
def _heads(t, B, N, H, dk):
    # (B,N,3,H,dk) -> (3,B,H,N,dk)   attention_variants.py:459
    return np.transpose(t.reshape(B, N, 3, H, dk), (2, 0, 3, 1, 4))


def _conv1d_dw_fwd(x, w, d, causal):
    """depthwise Conv1d over the last axis (:396-421): x (M,C,N), w (C,k); causal = left pad (k-1)d (:484-486),
    else symmetric pad d(k-1)//2.  out[m,c,n] = sum_t w[c,t] xpad[m,c,n+t*d]."""
    k = w.shape[1]
    N = x.shape[-1]
    left = (k - 1) * d if causal else d * (k - 1) // 2
    right = 0 if causal else d * (k - 1) // 2
    xp = np.pad(x, ((0, 0), (0, 0), (left, right)))
    No = xp.shape[-1] - d * (k - 1)
    out = np.zeros(x.shape[:2] + (No,), dtype=x.dtype)
    for t in range(k):
        out += w[None, :, t, None] * xp[:, :, t * d:t * d + No]
    assert No == N, "lens conv must preserve the token count"
    return out, (xp, left)


def _conv1d_dw_bwd(dout, w, d, ctx):
    xp, left = ctx
    k = w.shape[1]
    No = dout.shape[-1]
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for t in range(k):
        dxp[:, :, t * d:t * d + No] += w[None, :, t, None] * dout
        dw[:, t] = (dout * xp[:, :, t * d:t * d + No]).sum((0, 2))
    return dxp[:, :, left:left + No], dw


def _head_params(params):
    """(lowrank tuple | None, dense dict | None) from reference state_dict keys."""
    if "edge_head.conv1.weight" in params:
        hp = dict(W1=params["edge_head.conv1.weight"][:, :, 0, 0], b1=params["edge_head.conv1.bias"],
                  W2=params["edge_head.conv2.weight"][:, :, 0, 0], b2=params["edge_head.conv2.bias"])
        if "edge_head.mid3.weight" in params:
            hp.update(W3=params["edge_head.mid3.weight"], b3=params["edge_head.mid3.bias"])
        return None, hp
    return (params["edge_head.row_proj.weight"][:, :, 0], params["edge_head.row_proj.bias"],
            params["edge_head.col_proj.weight"][:, :, 0], params["edge_head.col_proj.bias"]), None


def module_fwd(x, params: Dict[str, np.ndarray], heads: int, n_views: int, share_qkv: bool,
               beta_not: float = 0.5, lens_dilations=None, lens_qk=None, drop=None, attn_mask=None):
    """lens_dilations: dilations of the S lens bank (params lens_bank.{l}.weight) or None;
    lens_qk: None or (dilations, causal) for the Q/K lens bank (params q_lens.{l}.weight, k_lens.{l}.weight);
    drop: None or the attn_drop multiplier keep / (1 - p) per edge, (B,H,N,N) (:552 with the mask made explicit)."""
    B, N, D = x.shape
    H, dk = heads, D // heads
    V = max(2, int(n_views))                                      # :362
    qk_ctx = None
    if share_qkv:
        qkv = _heads(x @ params["qkv.weight"].T, B, N, H, dk)
        qb, kb, vb = qkv[0], qkv[1], qkv[2]
        qs = params["q_scale"][:, None]                           # (V,1,H,1,dk)
        ks = params["k_scale"][:, None]
        vs = params["v_scale"][:, None]
        qv, kv = qb[None] * qs, kb[None] * ks                     # :462-463
        v_last = V - 1
        if lens_qk is not None:                                   # :472-498: the score views are rebuilt from view 0
            dils, causal = lens_qk
            qf = np.ascontiguousarray(qv[0]).reshape(B * H, dk, N)    # `.reshape(B*H, dk, N)` of a (B,H,N,dk) tensor :477-478
            kf = np.ascontiguousarray(kv[0]).reshape(B * H, dk, N)
            ql, kl, qk_ctx = [], [], []
            for l, dl in enumerate(dils):
                qo, cq = _conv1d_dw_fwd(qf, params[f"q_lens.{l}.weight"][:, 0], dl, causal)
                ko, ck = _conv1d_dw_fwd(kf, params[f"k_lens.{l}.weight"][:, 0], dl, causal)
                ql.append(np.swapaxes(qo.reshape(B, H, dk, N), 2, 3))     # :491-492
                kl.append(np.swapaxes(ko.reshape(B, H, dk, N), 2, 3))
                qk_ctx.append((cq, ck))
            qv, kv = np.stack(ql), np.stack(kl)
            v_last = min(V - 1, len(dils) - 1)                    # :556
        v0, vL = vb * vs[0], vb * vs[v_last]                      # :464, :556-557
    else:
        qkvs = [_heads(x @ params[f"qkv_list.{i}.weight"].T, B, N, H, dk) for i in range(V)]
        qv = np.stack([t[0] for t in qkvs])
        kv = np.stack([t[1] for t in qkvs])
        v0, vL = qkvs[0][2], qkvs[V - 1][2]
        v_last = V - 1
    low, dense = _head_params(params)
    Wr, br, Wc, bc = low if low is not None else (None, None, None, None)
    lens = None
    if lens_dilations is not None:
        lens = (np.stack([params[f"lens_bank.{l}.weight"][:, 0] for l in range(len(lens_dilations))]), tuple(lens_dilations))
    blocked = None if attn_mask is None else (np.asarray(attn_mask) == 0)       # reference convention: 0 = blocked (:504)
    y, cache = core_fwd(qv, kv, v0, vL, Wr, br, Wc, bc, beta_not, params["chain_value_logit"], dense=dense, lens=lens, drop=drop,
                        blocked=blocked)
    ycat = np.transpose(y, (0, 2, 1, 3)).reshape(B, N, D)         # :563
    out = ycat @ params["proj.weight"].T                          # :564
    cache.update(x=x, ycat=ycat, share=share_qkv, H=H, V=V, params=params, lens_qk=lens_qk, qk_ctx=qk_ctx,
                 v_last=v_last, lens_dilations=lens_dilations)
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
    if cache["dense"] is not None:
        grads["edge_head.conv1.weight"] = g["dW1"][:, :, None, None]
        grads["edge_head.conv1.bias"] = g["db1"]
        grads["edge_head.conv2.weight"] = g["dW2"][:, :, None, None]
        grads["edge_head.conv2.bias"] = g["db2"]
        if "dW3" in g:
            grads["edge_head.mid3.weight"] = g["dW3"]
            grads["edge_head.mid3.bias"] = g["db3"]
    else:
        grads["edge_head.row_proj.weight"] = g["dWr"][:, :, None]
        grads["edge_head.row_proj.bias"] = g["dbr"]
        grads["edge_head.col_proj.weight"] = g["dWc"][:, :, None]
        grads["edge_head.col_proj.bias"] = g["dbc"]
    if cache["lens_dilations"] is not None:
        for l in range(len(cache["lens_dilations"])):
            grads[f"lens_bank.{l}.weight"] = g["dlens_w"][l][:, None]

    def unheads(dq, dk_, dv):  # (B,H,N,dk)x3 -> (B,N,3D)
        t = np.stack([dq, dk_, dv])                               # (3,B,H,N,dk)
        return np.transpose(t, (1, 3, 0, 2, 4)).reshape(B, N, 3 * D)

    if cache["share"]:
        qb, kb, vb = cache["qb"], cache["kb"], cache["vb"]
        qs, ks, vs = p["q_scale"], p["k_scale"], p["v_scale"]
        dqv, dkv = g["dqv"], g["dkv"]
        if cache["lens_qk"] is not None:                          # back through the Q/K lens convolutions to view 0
            dils, causal = cache["lens_qk"]
            dqf = np.zeros((B * H, dk, N), dtype=x.dtype)
            dkf = np.zeros_like(dqf)
            for l, dl in enumerate(dils):
                cq, ck = cache["qk_ctx"][l]
                do_q = np.ascontiguousarray(np.swapaxes(dqv[l], 2, 3)).reshape(B * H, dk, N)
                do_k = np.ascontiguousarray(np.swapaxes(dkv[l], 2, 3)).reshape(B * H, dk, N)
                dq_in, dwq = _conv1d_dw_bwd(do_q, p[f"q_lens.{l}.weight"][:, 0], dl, cq)
                dk_in, dwk = _conv1d_dw_bwd(do_k, p[f"k_lens.{l}.weight"][:, 0], dl, ck)
                dqf += dq_in
                dkf += dk_in
                grads[f"q_lens.{l}.weight"] = dwq[:, None]
                grads[f"k_lens.{l}.weight"] = dwk[:, None]
            dqv = np.zeros((V,) + qb.shape, dtype=x.dtype)
            dkv = np.zeros_like(dqv)
            dqv[0] = dqf.reshape(B, H, N, dk)
            dkv[0] = dkf.reshape(B, H, N, dk)
        vl = cache["v_last"]
        grads["q_scale"] = (dqv * qb[None]).sum((1, 3))[:, :, None, :]
        grads["k_scale"] = (dkv * kb[None]).sum((1, 3))[:, :, None, :]
        dvs = np.zeros_like(vs)
        dvs[0] += (g["dv0"] * vb).sum((0, 2))[:, None, :]
        dvs[vl] += (g["dvL"] * vb).sum((0, 2))[:, None, :]
        grads["v_scale"] = dvs
        dqb = (dqv * qs[:, None]).sum(0)
        dkb = (dkv * ks[:, None]).sum(0)
        dvb = g["dv0"] * vs[0][None] + g["dvL"] * vs[vl][None]
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
