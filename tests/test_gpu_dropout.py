"""-m gpu: attention dropout inside the fused SDPA / Quartet kernels (reference: `self.attn_drop(A)` attention_variants.py:45,
components.py:62, whisper_mop.py:172; `self.attn_dropout(att)` quartet_attn_patch.py:119).  The mask is a counter-based function of
(seed, b, h, query, key) that `mop_amd.ops.dropout_keep_mask` restates on the host, so values and gradients are compared with the
oracle run under exactly the same mask; the reference's own RNG stream is not reproducible, its semantics (keep / (1 - p) on the
probabilities after the softmax) are."""
import numpy as np
import pytest
import torch

from gpu_util import max_abs, rel_err

pytestmark = pytest.mark.gpu


def _bhnd(t):                      # (B,N,H,d) torch -> (B,H,N,d) float64 numpy
    return t.detach().float().cpu().numpy().transpose(0, 2, 1, 3).astype(np.float64)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("N,dk,p", [(197, 64, 0.1), (70, 32, 0.5), (300, 64, 0.25)])
def test_sdpa_dropout_matches_the_oracle_under_the_same_mask(N, dk, p, causal):
    from mop_amd import ops
    from oracle import sdpa as O
    torch.manual_seed(N + dk)
    B, H, seed = 2, 3, 0xC0FFEE1234567
    q, k, v = (torch.randn(B, N, H, dk, device="cuda", dtype=torch.bfloat16).requires_grad_(True) for _ in range(3))
    w = torch.randn(B, N, H * dk, device="cuda", dtype=torch.bfloat16)
    y = ops.sdpa_core(q, k, v, causal=causal, dropout_p=p, seed=seed)
    assert ops.LAST_PATH["sdpa_fwd"] == 2
    y.backward(w)
    torch.cuda.synchronize()
    keep = ops.dropout_keep_mask(seed, p, B, H, N).numpy()
    drop = keep.astype(np.float64) / (1.0 - p)
    blocked = np.broadcast_to(~np.tril(np.ones((N, N), dtype=bool)), (B, H, N, N)) if causal else None
    yr, c = O.core_fwd(_bhnd(q), _bhnd(k), _bhnd(v), blocked, None, drop)
    g = O.core_bwd(w.float().cpu().numpy().reshape(B, N, H, dk).transpose(0, 2, 1, 3).astype(np.float64), c)
    got = y.detach().float().cpu().numpy().reshape(B, N, H, dk).transpose(0, 2, 1, 3)
    assert max_abs(got, yr) <= 1e-2 * max(1.0, float(np.abs(yr).max())), max_abs(got, yr)
    for name, t in (("dq", q), ("dk", k), ("dv", v)):
        assert rel_err(_bhnd(t.grad), g[name]) <= 3e-2, (name, rel_err(_bhnd(t.grad), g[name]))
    # the mask really bites: the undropped output differs
    y0 = ops.sdpa_core(q, k, v, causal=causal)
    assert float((y0 - y).detach().abs().max()) > 1e-2


def test_sdpa_dropout_is_reproducible_and_unbiased():
    from mop_amd import ops
    torch.manual_seed(1)
    B, N, H, dk = 2, 128, 4, 64
    q, k, v = (torch.randn(B, N, H, dk, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    with torch.no_grad():
        a = ops.sdpa_core(q, k, v, dropout_p=0.2, seed=11)
        b = ops.sdpa_core(q, k, v, dropout_p=0.2, seed=11)
        c = ops.sdpa_core(q, k, v, dropout_p=0.2, seed=12)
        y0 = ops.sdpa_core(q, k, v).float()
        mean = torch.stack([ops.sdpa_core(q, k, v, dropout_p=0.2, seed=100 + s).float() for s in range(64)]).mean(0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float((mean - y0).abs().max()) <= 0.15 * float(y0.abs().max())          # E[dropout(P)] = P
    torch.manual_seed(5)
    s1 = ops.dropout_seed()
    torch.manual_seed(5)
    assert ops.dropout_seed() == s1                                                 # torch.manual_seed reproduces the draw


def test_modules_train_with_attention_dropout():
    """the module-level contract: training mode applies dropout (outputs differ between calls and from eval), eval does not"""
    from mop_amd.nn import BaselineMSA, CausalSelfAttention, TransformerConfig
    from mop_amd.nn.components import MSA
    from mop_amd.nn.whisper_mop import MultiheadSelfAttention
    torch.manual_seed(0)
    x = torch.randn(2, 64, 128, device="cuda", dtype=torch.bfloat16)
    mods = [BaselineMSA(128, 4, attn_drop=0.2), MSA(128, 4, attn_drop=0.2), MultiheadSelfAttention(128, 4, 0.2, False, causal=False),
            CausalSelfAttention(TransformerConfig(n_head=4, n_embd=128, block_size=64, dropout=0.2))]
    for m in mods:
        m = m.cuda().to(torch.bfloat16)
        for d in m.modules():                                   # isolate the attention dropout: residual / projection dropouts off
            if isinstance(d, torch.nn.Dropout) and d is not getattr(m, "attn_drop", None):
                d.p = 0.0
        m.eval()
        with torch.no_grad():
            e1, e2 = m(x), m(x)
        assert torch.equal(e1, e2)
        m.train()
        xi = x.clone().requires_grad_(True)
        t1 = m(xi)
        t1.float().square().sum().backward()
        with torch.no_grad():
            t2 = m(x)
        assert not torch.equal(t1.detach(), t2) and not torch.equal(t1.detach(), e1), type(m).__name__
        assert torch.isfinite(xi.grad).all() and float(xi.grad.abs().max()) > 0


@pytest.mark.parametrize("use_quartet", [True, False])
def test_quartet_dropout_matches_the_oracle_under_the_same_mask(use_quartet):
    from mop_amd import ops
    from oracle import quartet as O
    torch.manual_seed(3)
    B, T, H, dh, p, seed = 2, 160, 3, 64, 0.2, 987654321012345
    ts = [torch.randn(B, T, H, dh, device="cuda", dtype=torch.bfloat16).requires_grad_(True) for _ in range(5)]
    q, k, v, q2, k2 = ts
    mix = torch.tensor([0.3], device="cuda", requires_grad=True)
    qs = torch.tensor([1.2], device="cuda", requires_grad=True)
    w = torch.randn(B, T, H * dh, device="cuda", dtype=torch.bfloat16)
    y = ops.quartet_core(q, k, v, q2 if use_quartet else None, k2 if use_quartet else None, mix, qs, None, 1e-5, use_quartet,
                         dropout_p=p, seed=seed)
    assert ops.LAST_PATH["quartet_fwd"] == 2
    y.backward(w)
    torch.cuda.synchronize()
    drop = ops.dropout_keep_mask(seed, p, B, H, T).numpy().astype(np.float64) / (1.0 - p)
    yr, c = O.core_fwd(_bhnd(q), _bhnd(k), _bhnd(v), _bhnd(q2), _bhnd(k2), float(mix.detach()), float(qs.detach()), 1e-5, use_quartet, True, None, drop)
    g = O.core_bwd(w.float().cpu().numpy().reshape(B, T, H, dh).transpose(0, 2, 1, 3).astype(np.float64), c)
    got = y.detach().float().cpu().numpy().reshape(B, T, H, dh).transpose(0, 2, 1, 3)
    assert max_abs(got, yr) <= 1e-2 * max(1.0, float(np.abs(yr).max())), max_abs(got, yr)
    names = [("dq", q), ("dk", k), ("dv", v)] + ([("dq2", q2), ("dk2", k2)] if use_quartet else [])
    for name, t in names:
        assert rel_err(_bhnd(t.grad), g[name]) <= 4e-2, (name, rel_err(_bhnd(t.grad), g[name]))
    if use_quartet:
        assert abs(float(mix.grad) - float(g["dmixture"])) <= 5e-2 * max(abs(float(g["dmixture"])), 1e-3)
        assert abs(float(qs.grad) - float(g["dquartet_scale"])) <= 5e-2 * max(abs(float(g["dquartet_scale"])), 1e-3)


@pytest.mark.parametrize("save_chain", [True, False, "dense"])
def test_edgewise_dropout_matches_the_oracle_under_the_same_mask(save_chain):
    """EdgewiseMSA(lowrank) in training mode with attn_drop (:552) on the fused kernels; both backward modes (saved chain state /
    forward re-run inside the backward, which must reproduce the same mask)."""
    import mop_amd
    from mop_amd import ops
    from mop_amd.nn import EdgewiseMSA
    from oracle import edgewise as O
    from gpu_util import check_grads, oracle_bf16_noise
    D, Hh, V, N, B, p = 128, 2, 3, 50, 3, 0.2
    torch.manual_seed(21)
    dense = save_chain == "dense"
    save_chain = bool(save_chain)
    kw = dict(gate_mode="dense", use_k3=False, gate_init="and") if dense else dict(gate_mode="lowrank", gate_rank=2, gate_init="mix5")
    m = EdgewiseMSA(D, Hh, attn_drop=p, n_views=V, share_qkv=True, **kw)
    with torch.no_grad():
        for n_, prm in m.named_parameters():
            prm.add_(0.1 * torch.randn_like(prm))
            if n_.endswith("conv2.bias"):
                prm.copy_(0.7 * torch.randn_like(prm))
    params = {k: v.detach().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    x = torch.randn(B, N, D)
    w = torch.randn(B, N, D)
    mop_amd.set_precision("bf16")
    ops.set_save_chain_state(save_chain)
    try:
        mg = m.cuda().to(torch.bfloat16).train()
        xg = x.cuda().to(torch.bfloat16).requires_grad_(True)
        torch.manual_seed(77)
        seed = ops.dropout_seed()
        torch.manual_seed(77)                              # the module draws exactly this seed
        y = mg(xg)
        assert ops.LAST_PATH["edgewise_fwd"] == 2
        y.backward(w.cuda().to(torch.bfloat16))
        torch.cuda.synchronize()
    finally:
        ops.set_save_chain_state(True)
        mop_amd.set_precision("auto")
    drop = ops.dropout_keep_mask(seed, p, B, Hh, N).numpy().astype(np.float64) / (1.0 - p)
    fwd = lambda xx, pp, *a: O.module_fwd(xx, pp, *a, drop=drop)
    rb = lambda t: torch.as_tensor(t).to(torch.bfloat16).double().numpy()       # what the kernels were given
    yr, cb = fwd(rb(x), {k: rb(v) for k, v in params.items()}, Hh, V, True, 0.5)
    dxr, _ = O.module_bwd(rb(w), cb)
    assert max_abs(y.detach().float().cpu().numpy(), yr) <= 1e-2 * max(1.0, float(np.abs(yr).max()))
    assert rel_err(xg.grad.float().cpu().numpy(), dxr) <= 3e-2
    _, c = fwd(x.double().numpy(), params, Hh, V, True, 0.5)
    _, gr = O.module_bwd(w.double().numpy(), c)
    grads = {k: v.grad.detach().float().cpu().numpy() for k, v in mg.named_parameters()}
    # per tensor 3e-2, or 4 x the amount the exact gradient itself moves when the inputs are rounded to bf16 (gate-head gradients are
    # differences of N^2 terms, with or without dropout: DESIGN.md section 5 "Accuracy")
    noise = oracle_bf16_noise(fwd, O.module_bwd, x.numpy(), w.numpy(), params, Hh, V, True, 0.5)
    check_grads(grads, gr, 3e-2, d=noise)
    with torch.no_grad():                                   # eval mode ignores attn_drop
        mg.eval()
        e1, e2 = mg(xg), mg(xg)
    assert torch.equal(e1, e2)


def test_multihop_dropout_matches_the_oracle_under_the_same_mask():
    """MultiHopMSA: attn_drop on the mixed weights (:222), transport term undropped (:224-227)"""
    from mop_amd import ops
    from oracle import multihop as O
    torch.manual_seed(9)
    B, N, H, dk, p, seed, hops = 2, 100, 2, 64, 0.3, 424242424242, 3
    ts = [torch.randn(B, N, H, dk, device="cuda", dtype=torch.bfloat16).requires_grad_(True) for _ in range(6)]
    logit = torch.tensor(-0.5, device="cuda", requires_grad=True)
    w = torch.randn(B, N, H * dk, device="cuda", dtype=torch.bfloat16)
    y = ops.dualpath_core(*ts, logit, 1.0, 0.3, 0.2, 0.0, 0.5, hops, dropout_p=p, seed=seed)
    assert ops.LAST_PATH["dualpath_fwd"] == 2
    y.backward(w)
    torch.cuda.synchronize()
    drop = ops.dropout_keep_mask(seed, p, B, H, N).numpy().astype(np.float64) / (1.0 - p)
    gates = dict(and_=1.0, or_=0.3, not_=0.2, chain=0.0)
    yr, c = O.core_fwd(*[_bhnd(t) for t in ts], gates, 0.5, hops, float(logit.detach()), None, drop=drop)
    g = O.core_bwd(w.float().cpu().numpy().reshape(B, N, H, dk).transpose(0, 2, 1, 3).astype(np.float64), c)
    got = y.detach().float().cpu().numpy().reshape(B, N, H, dk).transpose(0, 2, 1, 3)
    assert max_abs(got, yr) <= 1e-2 * max(1.0, float(np.abs(yr).max()))
    for name, t in zip(("dq1", "dk1", "dv1", "dq2", "dk2", "dv2"), ts):
        assert rel_err(_bhnd(t.grad), g[name]) <= 4e-2, (name, rel_err(_bhnd(t.grad), g[name]))


def test_crossview_and_multihop_modules_train_with_attention_dropout():
    from mop_amd.nn import CrossViewMixerMSA, MultiHopMSA
    torch.manual_seed(0)
    x = torch.randn(2, 64, 128, device="cuda", dtype=torch.bfloat16)
    for m in (MultiHopMSA(128, 2, attn_drop=0.2), CrossViewMixerMSA(128, 2, attn_drop=0.2, use_transpose_cues=False)):
        m = m.cuda().to(torch.bfloat16).train()
        xi = x.clone().requires_grad_(True)
        t1 = m(xi)
        t1.float().square().sum().backward()
        with torch.no_grad():
            t2 = m(x)
            m.eval()
            e1, e2 = m(x), m(x)
        assert torch.equal(e1, e2) and not torch.equal(t1.detach(), t2) and torch.isfinite(xi.grad).all()



# ---- generic paths (round 3): the same mask from the same seed, applied to the N x N map in a workspace plane -------------------------
def _pair(fn):
    """run fn() on the generic path and on the path the library picks; -> (generic, picked) results"""
    from mop_amd import ops
    out = []
    for path in ("generic", "auto"):
        ops.set_path(path)
        try:
            out.append(fn())
        finally:
            ops.set_path("auto")
    return out


def _close(a, b, tol):
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_generic_paths_draw_the_fused_kernels_mask():
    """SDPA, dual-path (MultiHopMSA) and Quartet cores: with one seed the generic multi-kernel path and the fused kernels drop the same
    weights -- outputs and input gradients agree to bf16 accuracy (they would differ by O(1) under different masks)."""
    import mop_amd
    from mop_amd import ops
    mop_amd.set_precision("bf16")
    try:
        torch.manual_seed(9)
        B, N, H, dk, p, seed = 2, 96, 2, 64, 0.3, 424242424242
        mk = lambda: torch.randn(B, N, H, dk, device="cuda", dtype=torch.bfloat16).requires_grad_(True)
        w = torch.randn(B, N, H * dk, device="cuda", dtype=torch.bfloat16)

        def run(core, ts, *args, **kw):
            for t in ts:
                t.grad = None
            y = core(*ts, *args, dropout_p=p, seed=seed, **kw)
            y.backward(w)
            return [y] + [t.grad.clone() for t in ts]

        q, k, v = mk(), mk(), mk()
        g, f = _pair(lambda: (run(ops.sdpa_core, [q, k, v], causal=True), ops.LAST_PATH["sdpa_fwd"]))
        assert (g[1], f[1]) == (1, 2) and all(_close(x, y, 3e-2) for x, y in zip(g[0], f[0]))
        y0 = ops.sdpa_core(q, k, v, causal=True)
        assert not _close(g[0][0], y0, 3e-2)                      # and the mask bites on the generic path

        ts = [mk() for _ in range(6)]
        logit = torch.tensor(-0.5, device="cuda")
        g, f = _pair(lambda: (run(lambda *a, **kw: ops.dualpath_core(*a[:6], logit, 1.0, 0.3, 0.2, 0.0, 0.5, 3, **kw), ts), ops.LAST_PATH["dualpath_fwd"]))
        assert (g[1], f[1]) == (1, 2) and all(_close(x, y, 4e-2) for x, y in zip(g[0], f[0]))

        ts = [mk() for _ in range(5)]
        mix, qs = torch.tensor([0.3], device="cuda"), torch.tensor([1.2], device="cuda")
        g, f = _pair(lambda: (run(lambda q_, k_, v_, q2, k2, **kw: ops.quartet_core(q_, k_, v_, q2, k2, mix, qs, None, 1e-5, True, **kw), ts),
                              ops.LAST_PATH["quartet_fwd"]))
        assert (g[1], f[1]) == (1, 2) and all(_close(x, y, 4e-2) for x, y in zip(g[0], f[0]))
    finally:
        mop_amd.set_precision("auto")


@pytest.mark.parametrize("variant", ["lowrank", "dense_k3_lens_mask"])
def test_edgewise_generic_path_dropout(variant):
    """EdgewiseMSA with attn_drop > 0 in training mode on the generic path (:552).  Low-rank head: the generic path against the fused
    kernels under one seed.  Dense head + use_k3 + S lens bank + attn_mask (generic only): against the float64 oracle run under the
    same mask."""
    import mop_amd
    from mop_amd import ops
    from mop_amd.nn import EdgewiseMSA
    from oracle import edgewise as O
    D, Hh, V, N, B, p = 64, 2, 3, 40, 2, 0.25
    torch.manual_seed(33)
    if variant == "lowrank":
        m = EdgewiseMSA(D, Hh, attn_drop=p, n_views=V, share_qkv=True, gate_mode="lowrank", gate_rank=2, gate_init="mix5")
    else:
        m = EdgewiseMSA(D, Hh, attn_drop=p, n_views=V, share_qkv=True, gate_mode="dense", use_k3=True, gate_init="and", use_lens_bank=True,
                        lens_dilations=(1, 2))
    with torch.no_grad():
        for n_, prm in m.named_parameters():
            prm.add_(0.1 * torch.randn_like(prm))
            if n_.endswith("conv2.bias"):
                prm.copy_(0.7 * torch.randn_like(prm))
    x, w = torch.randn(B, N, D), torch.randn(B, N, D)
    if variant == "lowrank":
        mop_amd.set_precision("bf16")
        try:
            mg = m.cuda().to(torch.bfloat16).train()

            def step():
                xg = x.cuda().to(torch.bfloat16).requires_grad_(True)
                mg.zero_grad()
                torch.manual_seed(5)                        # the module draws its mask seed from torch's generator
                y = mg(xg)
                y.backward(w.cuda().to(torch.bfloat16))
                return [y, xg.grad] + [prm.grad.clone() for prm in mg.parameters()], ops.LAST_PATH["edgewise_fwd"]
            g, f = _pair(step)
        finally:
            mop_amd.set_precision("auto")
        assert (g[1], f[1]) == (1, 2)
        assert _close(g[0][0], f[0][0], 2e-2) and _close(g[0][1], f[0][1], 4e-2)
        return
    mop_amd.set_precision("fp32")
    try:
        params = {k: v.detach().numpy().astype(np.float64) for k, v in m.state_dict().items()}
        mask = torch.ones(N, N).tril_()
        mg = m.cuda().train()
        xg = x.cuda().requires_grad_(True)
        torch.manual_seed(77)
        seed = ops.dropout_seed()
        torch.manual_seed(77)
        y = mg(xg, attn_mask=mask.cuda())
        assert ops.LAST_PATH["edgewise_fwd"] == 1
        y.backward(w.cuda())
    finally:
        mop_amd.set_precision("auto")
    drop = ops.dropout_keep_mask(seed, p, B, Hh, N).numpy().astype(np.float64) / (1.0 - p)
    yr, c = O.module_fwd(x.double().numpy(), params, Hh, V, True, 0.5, lens_dilations=(1, 2), attn_mask=mask.numpy(), drop=drop)
    dxr, gr = O.module_bwd(w.double().numpy(), c)
    assert max_abs(y.detach().cpu().numpy(), yr) <= 1e-4 and rel_err(xg.grad.cpu().numpy(), dxr) <= 1e-3
    for k_, prm in mg.named_parameters():
        assert rel_err(prm.grad.cpu().numpy().reshape(gr[k_].shape), gr[k_]) <= 1e-3 or max_abs(prm.grad.cpu().numpy().reshape(gr[k_].shape), gr[k_]) <= 1e-6, k_


def test_crossview_cues_and_prior_dropout_vs_autograd():
    """CrossViewMixerMSA core with transpose cues / the per-key prior (generic path) and attn_drop: against the reference's formula
    (:99-110, :124-152) written in torch float64 with the library's mask, gradients by autograd."""
    import mop_amd
    from mop_amd import ops
    torch.manual_seed(4)
    B, N, H, dk, p, seed = 2, 33, 2, 16, 0.2, 1357913579
    mop_amd.set_precision("fp32")
    try:
        for t1, t2, pw in ((0.3, -0.2, 0.0), (0.0, 0.0, 0.5)):
            ts = [torch.randn(B, N, H, dk, device="cuda").requires_grad_(True) for _ in range(5)]
            mix = torch.tensor([[1.0, 0.2], [-0.3, 0.8]], device="cuda", requires_grad=True)
            w = torch.randn(B, N, H * dk, device="cuda")
            y = ops.crossview_core(*ts, mix, t1=t1, t2=t2, prior_weight=pw, anchor_mode="fixed", fixed_k_star=3, dropout_p=p, seed=seed)
            assert ops.LAST_PATH["crossview_fwd"] == 1
            y.backward(w)
            got = [y.detach()] + [t.grad.clone() for t in ts] + [mix.grad.clone()]
            q1, k1, v1, q2, k2 = (t.detach().double().permute(0, 2, 1, 3).requires_grad_(True) for t in ts)
            mx = mix.detach().double().requires_grad_(True)
            sc = dk ** -0.5
            S1, S2 = q1 @ k1.transpose(-1, -2) * sc, q2 @ k2.transpose(-1, -2) * sc
            S12, S21 = q1 @ k2.transpose(-1, -2) * sc, q2 @ k1.transpose(-1, -2) * sc
            S = mx[0, 0] * S1 + mx[0, 1] * S12 + mx[1, 0] * S21 + mx[1, 1] * S2 + t1 * S1.transpose(-1, -2) + t2 * S2.transpose(-1, -2)
            A = torch.softmax(S, -1)
            if pw > 0:
                A1, A2 = torch.softmax(S1, -1), torch.softmax(S2, -1)
                sharp = A1 * A2[:, :, 3:4, :]
                A = (1 - pw) * A + pw * sharp / (sharp.sum(-1, keepdim=True) + 1e-9)
            keep = ops.dropout_keep_mask(seed, p, B, H, N).to("cuda").double() / (1 - p)
            yr = ((A * keep) @ v1).permute(0, 2, 1, 3).reshape(B, N, H * dk)
            yr.backward(w.double())
            ref = [yr.detach()] + [t.grad.permute(0, 2, 1, 3) for t in (q1, k1, v1, q2, k2)] + [mx.grad]
            for a_, b_ in zip(got, ref):
                assert float((a_.double() - b_).abs().max()) <= 1e-3 * max(1.0, float(b_.abs().max()))
    finally:
        mop_amd.set_precision("auto")
