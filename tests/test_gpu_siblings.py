"""MI355X parity of the sibling attention cores (SDPA, MultiHop dual-path, Quartet) against the
reference's golden vectors; forward and gradients, fp32 arithmetic (<=1e-3) and bf16 MFMA (<=1e-2)."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from gpu_util import check_grads, max_abs, module_from_golden, rel_err, run_fwd_bwd

pytestmark = pytest.mark.gpu
TOL = {"fp32": (1e-3, 1e-3), "bf16": (1e-2, 3e-2)}


@pytest.fixture(autouse=True)
def _reset():
    import mop_amd
    yield
    mop_amd.set_precision("auto")


def _check(y, dx, grads, d, gref, prec):
    tol, gtol = TOL[prec]
    assert max_abs(y, d["y"]) <= tol, f"y {max_abs(y, d['y']):.3e}"
    assert rel_err(dx, d["dx"]) <= gtol, f"dx {rel_err(dx, d['dx']):.3e}"
    check_grads(grads, gref, gtol, scalar_tol=5e-2 if prec == "bf16" else None, floor=1e-2 if prec == "bf16" else 1e-3,
                d=d if prec == "bf16" else None)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", golden_names("sdpa_"))
def test_baseline_msa(name, prec):
    import mop_amd
    from mop_amd.nn import BaselineMSA
    d, params, gref, meta = load_golden(name)
    mop_amd.set_precision(prec)
    m = module_from_golden(BaselineMSA, params, dim=meta["dim"], heads=meta["heads"])
    fk = {}
    if "attn_mask" in d:
        fk["attn_mask"] = torch.from_numpy(d["attn_mask"]).cuda()
    _check(*run_fwd_bwd(m, d["x"], d["w"], **fk), d, gref, prec)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", golden_names("mh_"))
def test_multihop(name, prec):
    import mop_amd
    from mop_amd.nn import MultiHopMSA
    d, params, gref, meta = load_golden(name)
    mop_amd.set_precision(prec)
    gates = dict(and_=meta["g_and"], or_=meta["g_or"], not_=meta["g_not"], chain=meta["g_chain"])
    m = module_from_golden(MultiHopMSA, params, dim=meta["dim"], heads=meta["heads"], beta_not=meta["beta_not"],
                           gates=gates, hops=meta["hops"])
    fk = {}
    if "attn_mask" in d:
        fk["attn_mask"] = torch.from_numpy(d["attn_mask"]).cuda()
    _check(*run_fwd_bwd(m, d["x"], d["w"], **fk), d, gref, prec)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", golden_names("qt_"))
def test_quartet(name, prec):
    import mop_amd
    from mop_amd.nn import CausalSelfAttention, TransformerConfig
    d, params, gref, meta = load_golden(name)
    mop_amd.set_precision(prec)
    T = d["x"].shape[1]
    cfg = TransformerConfig(n_head=meta["heads"], n_embd=meta["dim"], block_size=max(T, 16), dropout=0.0,
                            bias=any(k.endswith(".bias") for k in params), use_quartet=bool(meta["use_quartet"]),
                            score_norm_eps=meta["eps"])
    m = module_from_golden(CausalSelfAttention, params, config=cfg)
    fk = {}
    if "attention_mask" in d:
        fk["attention_mask"] = torch.from_numpy(d["attention_mask"]).cuda()
    _check(*run_fwd_bwd(m, d["x"], d["w"], **fk), d, gref, prec)


def _cv_ctor(meta):
    return dict(dim=meta["dim"], heads=meta["heads"], use_transpose_cues=bool(meta["use_transpose_cues"]), t1=meta["t1"],
                t2=meta["t2"], enable_per_key_prior=bool(meta["enable_per_key_prior"]), prior_weight=meta["prior_weight"],
                anchor_mode=meta["anchor_mode"], fixed_k_star=meta["fixed_k_star"])


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", golden_names("cv_"))
def test_crossview_mixer(name, prec):
    """CrossViewMixerMSA (reference attention_variants.py:51-156): 2x2 mix, transpose cues, masks, per-key prior."""
    import mop_amd
    from mop_amd import ops
    from mop_amd.nn import CrossViewMixerMSA
    d, params, gref, meta = load_golden(name)
    mop_amd.set_precision(prec)
    m = module_from_golden(CrossViewMixerMSA, params, **_cv_ctor(meta))
    fk = {}
    if "attn_mask" in d:
        fk["attn_mask"] = torch.from_numpy(d["attn_mask"]).cuda()
    out = run_fwd_bwd(m, d["x"], d["w"], **fk)
    if "k_star" in d:
        # anchor_mode="argmax_row_sum": argmax over row sums that are all 1 up to rounding, i.e. the anchor rows are decided by
        # rounding noise.  Where this device picked the reference's rows the fixture is the target; where it did not, the target is
        # the oracle (pinned to the reference at the reference's rows by tests/test_oracle_golden.py) evaluated AT THE DEVICE'S rows
        mine = ops.LAST_PATH["crossview_k_star"].cpu().numpy()
        if not np.array_equal(mine, d["k_star"]):
            from oracle import crossview as oc
            p64 = {k: v.astype(np.float64) for k, v in params.items()}
            kw = _cv_ctor(meta)
            yo, c = oc.module_fwd(d["x"].astype(np.float64), p64, kw["heads"], fk.get("attn_mask").cpu().numpy() if fk else None,
                                  kw["use_transpose_cues"], kw["t1"], kw["t2"], kw["enable_per_key_prior"], kw["prior_weight"],
                                  "fixed", 0, k_star=mine.astype(np.int64))
            dxo, go = oc.module_bwd(d["w"].astype(np.float64), c)
            y, dx, grads = out
            tol, gtol = TOL[prec]
            assert max_abs(y, yo) <= tol and rel_err(dx, dxo) <= gtol
            check_grads(grads, go, gtol, floor=1e-2 if prec == "bf16" else 1e-3)
            return
    _check(*out, d, gref, prec)


def test_crossview_argmax_anchor_is_self_consistent():
    """whatever row argmax_row_sum picks on this device, the result equals anchor_mode='fixed' with that row (single b,h)."""
    import mop_amd
    from mop_amd import ops
    from mop_amd.nn import CrossViewMixerMSA
    mop_amd.set_precision("fp32")
    torch.manual_seed(3)
    x = torch.randn(1, 40, 32, device="cuda")
    ma = CrossViewMixerMSA(32, 1, enable_per_key_prior=True, prior_weight=0.5).cuda().eval()
    ya = ma(x)
    k = int(ops.LAST_PATH["crossview_k_star"][0, 0])
    mf = CrossViewMixerMSA(32, 1, enable_per_key_prior=True, prior_weight=0.5, anchor_mode="fixed", fixed_k_star=k).cuda().eval()
    mf.load_state_dict(ma.state_dict())
    assert torch.equal(ya, mf(x))


def test_unified_mode_c_runs():
    from mop_amd.nn import UnifiedMSA
    torch.manual_seed(0)
    m = UnifiedMSA("C", 64, 4, t1=0.1).cuda()
    x = torch.randn(2, 9, 64, device="cuda", requires_grad=True)
    m(x).sum().backward()
    assert x.grad.shape == x.shape and torch.isfinite(x.grad).all() and m.impl.mix.grad.shape == (2, 2)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", golden_names("wh_"))
def test_whisper_encoder_block(name, prec):
    """Whisper-MoP EncoderBlock (non-causal MultiheadSelfAttention core in libmopk, MoP2D gate, MLP) vs the reference."""
    import mop_amd
    from mop_amd.nn import EncoderBlock, WhisperConfig
    d, params, gref, meta = load_golden(name)
    mop_amd.set_precision(prec)
    T = d["x"].shape[1]
    cfg = WhisperConfig(n_mels=meta["n_mels"], n_audio_ctx=T, n_embd=meta["dim"], n_head=meta["heads"], n_layer_enc=1,
                        n_layer_dec=1, bias=bool(meta["bias"]), n_views=meta["n_views"], n_kernels=meta["n_kernels"],
                        kernel_size=meta["kernel_size"])
    mel = torch.from_numpy(d["mel"]).cuda()

    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blk = EncoderBlock(cfg)

        def forward(self, x):
            return self.blk(x, mel)[0]

    m = module_from_golden(Wrap, params)
    y, dx, grads = run_fwd_bwd(m, d["x"], d["w"])
    tol, gtol = TOL[prec]
    assert max_abs(y, d["y"]) / float(np.abs(d["y"]).max()) <= tol
    assert rel_err(dx, d["dx"]) <= gtol
    check_grads(grads, gref, gtol, floor=1e-2 if prec == "bf16" else 1e-3, d=d if prec == "bf16" else None)


def test_whisper_self_attention_causal_bias_matches_torch():
    """causal flag + additive attn_bias of MultiheadSelfAttention (reference :163-170) against the same math in torch."""
    import mop_amd
    from mop_amd.nn import MultiheadSelfAttention
    mop_amd.set_precision("fp32")
    torch.manual_seed(1)
    m = MultiheadSelfAttention(64, 4, 0.0, True, causal=True).cuda().eval()
    x = torch.randn(2, 19, 64, device="cuda")
    bias = 0.5 * torch.randn(2, 1, 19, 19, device="cuda")
    q, k, v = (p(x).view(2, 19, 4, 16).transpose(1, 2) for p in (m.q_proj, m.k_proj, m.v_proj))
    att = (q @ k.transpose(-2, -1)) * m.scale
    att = att.masked_fill(~torch.tril(torch.ones(19, 19, dtype=torch.bool, device="cuda")), float("-inf")) + bias
    ref = m.o_proj((att.softmax(-1) @ v).transpose(1, 2).reshape(2, 19, 64))
    assert float((m(x, attn_bias=bias) - ref).detach().abs().max()) <= 1e-4


@pytest.mark.parametrize("shape", [
    # B, N, H, dk, causal, io dtype
    (2, 65, 6, 64, False, torch.bfloat16), (1, 197, 2, 64, False, torch.float32), (2, 300, 3, 32, True, torch.bfloat16),
    (1, 1000, 2, 64, True, torch.bfloat16), (1, 128, 1, 64, False, torch.bfloat16), (3, 1, 2, 32, False, torch.float32),
    (1, 129, 2, 64, True, torch.float32)])
def test_sdpa_flash_matches_generic_and_torch(shape):
    """fused (flash-style) SDPA kernels: forward and gradients against the generic HIP path and a plain torch fp32 reference."""
    import mop_amd
    from mop_amd import ops, _lib
    B, N, H, dk, causal, dt = shape
    mop_amd.set_precision("bf16")
    g = torch.Generator(device="cuda").manual_seed(N)
    qkv = torch.randn(B, N, 3, H, dk, device="cuda", generator=g)
    dy = torch.randn(B, N, H * dk, device="cuda", generator=g)
    res = {}
    for path in ("fused", "generic"):
        ops.set_path(path)
        t = qkv.to(dt).requires_grad_(True)
        y = ops.sdpa_core(t[:, :, 0], t[:, :, 1], t[:, :, 2], causal=causal)
        y.backward(dy.to(dt))
        res[path] = (y.float(), t.grad.float())
        assert ops.LAST_PATH["sdpa_fwd"] == (_lib.PATH_FUSED if path == "fused" else _lib.PATH_GENERIC)
    ops.set_path("auto")
    t = qkv.to(dt).float().requires_grad_(True)                  # torch fp32 reference on the same (rounded) inputs
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    att = (q @ k.transpose(-2, -1)) / dk ** 0.5
    if causal:
        att = att.masked_fill(~torch.tril(torch.ones(N, N, dtype=torch.bool, device="cuda")), float("-inf"))
    yr = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * dk)
    yr.backward(dy.to(dt).float())
    for path in ("fused", "generic"):
        y, gr = res[path]
        # north_star bf16 bound on outputs; |y| reaches ~3 here (y_0 = v_0 under the causal mask), where one bf16 ulp is 1.6e-2
        assert float((y - yr).detach().abs().max()) <= 1e-2 * max(1.0, float(yr.detach().abs().max())), path
        assert float((gr - t.grad).abs().max()) / float(t.grad.abs().max()) <= 3e-2, path
    assert float((res["fused"][0] - res["generic"][0]).abs().max()) <= 1e-2 * max(1.0, float(yr.detach().abs().max()))


@pytest.mark.parametrize("cfg", [
    # B, N, H, dk, hops, (and, or, not), causal, io dtype
    (2, 197, 2, 64, 3, (1.0, 0.0, 0.0), False, torch.bfloat16), (1, 130, 3, 32, 2, (0.7, 0.4, 0.3), True, torch.bfloat16),
    (2, 64, 2, 64, 4, (0.9, 0.5, 0.0), False, torch.float32), (1, 300, 1, 64, 3, (1.0, 0.0, 0.6), True, torch.float32),
    # "mask": an explicit, non-triangular mask tensor (attention_variants.py:202-207, :219-220), per batch / broadcast over heads
    (2, 150, 2, 64, 3, (1.0, 0.3, 0.0), "mask", torch.bfloat16), (1, 197, 3, 32, 2, (0.8, 0.0, 0.4), "mask", torch.float32)])
def test_dualpath_fused_matches_generic(cfg):
    """MultiHopMSA core on the fused kernels (mixed logits in one pass, transport as chained passes) vs the generic path."""
    import mop_amd
    from mop_amd import ops, _lib
    B, N, H, dk, hops, (g_and, g_or, g_not), causal, dt = cfg
    mop_amd.set_precision("bf16")
    g = torch.Generator(device="cuda").manual_seed(N + hops)
    base = [torch.randn(B, N, H, dk, device="cuda", generator=g) for _ in range(6)]
    dy = torch.randn(B, N, H * dk, device="cuda", generator=g)
    mask = None
    if causal == "mask":
        mask = (torch.rand(B, 1, N, N, device="cuda", generator=g) > 0.35)
        mask |= torch.eye(N, dtype=torch.bool, device="cuda")            # every row keeps at least one key
        causal = False
    res = {}
    for path in ("fused", "generic"):
        ops.set_path(path)
        ts = [t.to(dt).requires_grad_(True) for t in base]
        lg = torch.tensor(-0.3, device="cuda", requires_grad=True)
        y = ops.dualpath_core(*ts, lg, g_and, g_or, g_not, 0.0, 0.6, hops, mask, causal=causal)
        y.backward(dy.to(dt))
        res[path] = [y.float()] + [t.grad.float() for t in ts] + [lg.grad.float()]
        assert ops.LAST_PATH["dualpath_fwd"] == (_lib.PATH_FUSED if path == "fused" else _lib.PATH_GENERIC)
    ops.set_path("auto")
    names = ["y", "dq1", "dk1", "dv1", "dq2", "dk2", "dv2", "dlogit"]
    for n_, a_, b_ in zip(names, res["fused"], res["generic"]):
        den = max(1.0, float(b_.abs().max())) if n_ == "y" else float(b_.abs().max())
        assert float((a_ - b_).abs().max()) / den <= (1e-2 if n_ == "y" else 4e-2), n_


@pytest.mark.parametrize("cfg", [
    # B, T, H, dh, use_quartet, io dtype
    (2, 64, 2, 64, True, torch.bfloat16), (1, 200, 3, 32, True, torch.float32), (2, 130, 2, 64, False, torch.bfloat16),
    (1, 513, 2, 64, True, torch.bfloat16), (1, 1, 2, 32, True, torch.float32)])
def test_quartet_fused_matches_generic(cfg):
    """Quartet core on the fused kernels (row statistics pass + online-softmax pass; dense z-norm corrections in the backward)."""
    import mop_amd
    from mop_amd import ops, _lib
    B, T, H, dh, uq, dt = cfg
    mop_amd.set_precision("bf16")
    g = torch.Generator(device="cuda").manual_seed(T)
    base = [torch.randn(B, T, H, dh, device="cuda", generator=g) for _ in range(5)]
    dy = torch.randn(B, T, H * dh, device="cuda", generator=g)
    res = {}
    for path in ("fused", "generic"):
        ops.set_path(path)
        ts = [t.to(dt).requires_grad_(True) for t in base]
        mix = torch.tensor([0.3], device="cuda", requires_grad=True)
        qs = torch.tensor([0.8], device="cuda", requires_grad=True)
        y = ops.quartet_core(ts[0], ts[1], ts[2], ts[3] if uq else None, ts[4] if uq else None, mix if uq else None, qs if uq else None,
                             None, 1e-5, uq)
        y.backward(dy.to(dt))
        n = 5 if uq else 3
        res[path] = [y.float()] + [t.grad.float() for t in ts[:n]] + ([mix.grad.float(), qs.grad.float()] if uq else [])
        assert ops.LAST_PATH["quartet_fwd"] == (_lib.PATH_FUSED if path == "fused" else _lib.PATH_GENERIC)
    ops.set_path("auto")
    names = ["y", "dq", "dk", "dv", "dq2", "dk2", "dmixture", "dquartet_scale"]
    for n_, a_, b_ in zip(names, res["fused"], res["generic"]):
        den = max(1.0, float(b_.abs().max())) if n_ == "y" else max(float(b_.abs().max()), 1e-6)
        assert float((a_ - b_).abs().max()) / den <= (1e-2 if n_ == "y" else 5e-2), n_


@pytest.mark.parametrize("causal", [False, True, "mask"])
def test_crossview_fused_default_matches_generic(causal):
    """CrossViewMixerMSA without cues / prior: the 2x2 mix folds into two mixed key tensors + the fused two-score kernels."""
    import mop_amd
    from mop_amd import ops, _lib
    from mop_amd.nn import CrossViewMixerMSA
    mop_amd.set_precision("bf16")
    torch.manual_seed(5)
    m = CrossViewMixerMSA(128, 2).cuda()
    with torch.no_grad():
        m.mix.add_(0.3 * torch.randn(2, 2, device="cuda"))
    x = torch.randn(2, 150, 128, device="cuda")
    mask = torch.tril(torch.ones(150, 150, device="cuda")).view(1, 1, 150, 150) if causal else None
    if causal == "mask":                             # an explicit non-triangular mask tensor runs on the fused path too
        mask = ((torch.rand(2, 1, 150, 150, device="cuda") > 0.3) | torch.eye(150, dtype=torch.bool, device="cuda")).float()
    res = {}
    for path in ("auto", "generic"):
        ops.set_path(path)
        m.zero_grad()
        xg = x.clone().requires_grad_(True)
        y = m(xg, attn_mask=mask)
        y.square().sum().backward()
        res[path] = [y.detach(), xg.grad, m.mix.grad.clone(), m.qkv2.weight.grad.clone()]
        assert ops.LAST_PATH["crossview_fwd"] == (_lib.PATH_FUSED if path == "auto" else _lib.PATH_GENERIC)
    ops.set_path("auto")
    for a_, b_ in zip(res["auto"], res["generic"]):
        assert float((a_ - b_).abs().max()) / max(float(b_.abs().max()), 1e-6) <= 4e-2


@pytest.mark.parametrize("causal", [False, True])
def test_sdpa_flash_mask_and_bias_tensors(causal):
    """explicit uint8 mask + additive bias applied inside the fused kernels (Whisper attn_bias, arbitrary BaselineMSA masks)."""
    import mop_amd
    from mop_amd import ops, _lib
    mop_amd.set_precision("bf16")
    B, N, H, dk = 2, 150, 3, 64
    g = torch.Generator(device="cuda").manual_seed(7)
    qkv = torch.randn(B, N, 3, H, dk, device="cuda", generator=g)
    dy = torch.randn(B, N, H * dk, device="cuda", generator=g)
    mask = (torch.rand(B, 1, N, N, device="cuda", generator=g) > 0.3)
    mask[..., torch.arange(N), torch.arange(N)] = True                      # keep the diagonal: no fully blocked row
    bias = 0.5 * torch.randn(1, H, N, N, device="cuda", generator=g)
    res = {}
    for path in ("fused", "generic"):
        ops.set_path(path)
        t = qkv.to(torch.bfloat16).requires_grad_(True)
        y = ops.sdpa_core(t[:, :, 0], t[:, :, 1], t[:, :, 2], attn_mask=mask, bias=bias, causal=causal)
        y.backward(dy.to(torch.bfloat16))
        res[path] = (y.float(), t.grad.float())
        assert ops.LAST_PATH["sdpa_fwd"] == (_lib.PATH_FUSED if path == "fused" else _lib.PATH_GENERIC)
    ops.set_path("auto")
    t = qkv.to(torch.bfloat16).float().requires_grad_(True)
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    att = (q @ k.transpose(-2, -1)) / dk ** 0.5 + bias
    keep = mask & (torch.tril(torch.ones(N, N, dtype=torch.bool, device="cuda")) if causal else True)
    yr = (att.masked_fill(~keep, float("-inf")).softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * dk)
    yr.backward(dy.to(torch.bfloat16).float())
    for path in ("fused", "generic"):
        y, gr = res[path]
        assert float((y - yr.detach()).abs().max()) <= 1e-2 * max(1.0, float(yr.detach().abs().max())), path
        assert float((gr - t.grad).abs().max()) / float(t.grad.abs().max()) <= 3e-2, path


def test_quartet_fused_additive_mask_matches_generic():
    """attention_mask (additive, reference :115-116) applied inside the fused Quartet kernels."""
    import mop_amd
    from mop_amd import ops, _lib
    mop_amd.set_precision("bf16")
    B, T, H, dh = 2, 150, 2, 64
    g = torch.Generator(device="cuda").manual_seed(11)
    base = [torch.randn(B, T, H, dh, device="cuda", generator=g) for _ in range(5)]
    dy = torch.randn(B, T, H * dh, device="cuda", generator=g)
    am = 0.7 * torch.randn(B, 1, T, T, device="cuda", generator=g)
    res = {}
    for path in ("fused", "generic"):
        ops.set_path(path)
        ts = [t.to(torch.bfloat16).requires_grad_(True) for t in base]
        mix = torch.tensor([0.3], device="cuda", requires_grad=True)
        qs = torch.tensor([0.8], device="cuda", requires_grad=True)
        y = ops.quartet_core(*ts, mix, qs, am, 1e-5, True)
        y.backward(dy.to(torch.bfloat16))
        res[path] = [y.float()] + [t.grad.float() for t in ts]
        assert ops.LAST_PATH["quartet_fwd"] == (_lib.PATH_FUSED if path == "fused" else _lib.PATH_GENERIC)
    ops.set_path("auto")
    for i, (a_, b_) in enumerate(zip(res["fused"], res["generic"])):
        den = max(1.0, float(b_.abs().max())) if i == 0 else float(b_.abs().max())
        assert float((a_ - b_).detach().abs().max()) / den <= (1e-2 if i == 0 else 5e-2), i


def test_quartet_need_weights_rows_sum_to_one():
    from mop_amd.nn import CausalSelfAttention, TransformerConfig
    torch.manual_seed(0)
    m = CausalSelfAttention(TransformerConfig(n_head=2, n_embd=32, block_size=16, dropout=0.0)).cuda().eval()
    y, attn = m(torch.randn(2, 12, 32, device="cuda"), need_weights=True)
    assert y.shape == (2, 12, 32) and attn.shape == (2, 2, 12, 12)
    assert torch.allclose(attn.sum(-1), torch.ones_like(attn.sum(-1)), atol=1e-5)
    assert float(attn.triu(1).abs().max()) == 0.0


def test_vit_mop_shapes_and_gate_api():
    """reference tests/test_forward_shapes.py::{test_vit_shapes,test_gate_api} on the GPU path."""
    from mop_amd.nn import ViT_MoP
    torch.manual_seed(0)
    x = torch.randn(2, 3, 32, 32, device="cuda")
    m = ViT_MoP(dim=256, depth=2, heads=2, n_classes=10, n_views=2, n_kernels=1).cuda().eval()
    with torch.no_grad():
        assert m(x).shape == (2, 10)
    gates, views, kernels = m.get_gate_maps(x)
    assert gates.ndim == 4 and gates.shape[1] == 1 and views.shape[1] == 2 and kernels.shape[1] == 1


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", golden_names("vit_"))
def test_vit_mop_vs_reference_golden(name, prec):
    """ViT_MoP end to end against the reference (vit_mop.py:84-140): logits, the three gate maps, dx and every parameter gradient
    (strided sample + L2 norm).  `vit_cfg0_5m` is BASELINE.json configs[0]: 5,397,972 parameters, d = 384, 6 heads, 5 views.
    The parameters are regenerated from the fixture's seed (tests/vit_fixture.py) and loaded with strict=True."""
    import mop_amd
    from mop_amd.nn import ViT_MoP
    from vit_fixture import fill_params, grad_sample
    d, _, _, meta = load_golden(name)
    shapes = {k[6:]: tuple(int(v) for v in d[k]) for k in d if k.startswith("shape:")}        # the reference's state_dict order
    vals = fill_params(shapes, int(meta["param_seed"]))
    m = ViT_MoP(dim=int(meta["dim"]), depth=int(meta["depth"]), heads=int(meta["heads"]), n_classes=int(meta["n_classes"]),
                n_views=int(meta["n_views"]), n_kernels=int(meta["n_kernels"]), drop_path=0.0)
    m.load_state_dict({k: torch.from_numpy(v).reshape(shapes[k]) for k, v in vals.items()}, strict=True)
    assert sum(p.numel() for p in m.parameters()) == int(meta["n_params"])
    mop_amd.set_precision(prec)
    m = m.cuda().eval()
    x = torch.from_numpy(d["x"]).cuda().requires_grad_(True)
    y = m(x)
    y.backward(torch.from_numpy(d["w"]).cuda())
    torch.cuda.synchronize()
    tol, gtol = TOL[prec]
    assert max_abs(y.detach().cpu().numpy(), d["y"]) <= tol, f"logits {max_abs(y.detach().cpu().numpy(), d['y']):.3e}"
    gate, views, kernels = m.get_gate_maps(x.detach())
    for got, key in ((gate, "gate"), (views, "views"), (kernels, "kernels")):
        assert got.shape == d[key].shape and max_abs(got.cpu().numpy(), d[key]) <= tol * max(1.0, float(np.abs(d[key]).max())), key
    assert rel_err(x.grad.cpu().numpy(), d["dx"]) <= gtol, f"dx {rel_err(x.grad.cpu().numpy(), d['dx']):.3e}"
    gscale = max(float(d[k]) for k in d if k.startswith("gnorm:"))
    for k, p in m.named_parameters():
        smp, nrm = grad_sample(p.grad.detach().float().cpu().numpy())
        ref_s, ref_n = d["gsample:" + k], float(d["gnorm:" + k])
        assert abs(float(nrm) - ref_n) <= gtol * max(ref_n, 1e-3 * gscale), f"|grad {k}| {float(nrm):.4e} vs {ref_n:.4e}"
        assert max_abs(smp, ref_s) <= gtol * max(float(np.abs(ref_s).max()), 1e-3 * gscale / max(1.0, np.sqrt(p.numel()))), f"grad sample {k}"


def test_training_steps_match_the_reference_recipe():
    """six steps of the reference's training recipe (AdamW + LinearLR -> cosine, experiments/cifar100_ab5_param_budgets.py:464-479,
    loop :793-804) on ViT_MoP through `DataParallelStep`: loss and learning-rate sequences and the parameters after the last step
    against the sequence recorded from the reference (tests/golden/train_vit_tiny_adamw6.npz)."""
    import mop_amd
    from mop_amd.nn import ViT_MoP
    from mop_amd.training import DataParallelStep, make_optimizer_and_schedule
    from vit_fixture import fill_params, grad_sample
    d, _, _, meta = load_golden("train_vit_tiny_adamw6")
    shapes = {k[6:]: tuple(int(v) for v in d[k]) for k in d if k.startswith("shape:")}
    vals = fill_params(shapes, int(meta["param_seed"]))
    m = ViT_MoP(dim=int(meta["dim"]), depth=int(meta["depth"]), heads=int(meta["heads"]), n_classes=int(meta["n_classes"]),
                n_views=int(meta["n_views"]), n_kernels=int(meta["n_kernels"]), drop_path=0.0)
    m.load_state_dict({k: torch.from_numpy(v).reshape(shapes[k]) for k, v in vals.items()}, strict=True)
    mop_amd.set_precision("fp32")
    m = m.cuda().train()
    steps = int(meta["steps"])
    opt, sched = make_optimizer_and_schedule(m, lr=float(meta["lr"]), weight_decay=float(meta["weight_decay"]), steps=steps,
                                             warmup_frac=float(meta["warmup_frac"]))
    step = DataParallelStep(m, opt, torch.nn.functional.cross_entropy, sched)
    xs, ys = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["labels"]).cuda()
    losses, lrs = [], []
    for i in range(steps):
        lrs.append(opt.param_groups[0]["lr"])
        losses.append(float(step(xs[i], ys[i])))
    assert np.allclose(lrs, d["lr"], rtol=1e-12, atol=0.0)
    assert np.allclose(losses, d["loss"], rtol=2e-4, atol=0.0), (losses, list(d["loss"]))
    # parameters after the last step: Adam's update is lr * m / (sqrt(v) + eps), i.e. +-lr for a coordinate whose gradient is
    # rounding noise, so single entries may differ by a few lr; norms and the bulk of the sample must agree
    lr_sum = float(np.sum(d["lr"]))
    for k, v in m.state_dict().items():
        smp, nrm = grad_sample(v.detach().float().cpu().numpy())
        ref_s, ref_n = d["psample:" + k], float(d["pnorm:" + k])
        assert abs(float(nrm) - ref_n) <= 1e-3 * max(ref_n, 1e-3), k
        diff = np.abs(smp - ref_s)
        assert float(np.median(diff)) <= 1e-5 and float(diff.max()) <= 2.0 * lr_sum, (k, float(diff.max()))


# ---- BASELINE.json config sizes: size-independent properties of the fused sibling kernels (no oracle at these sizes)
def test_sdpa_whisper_size_key_permutation_and_value_linearity():
    """T=3000, d=384, H=6 (config 5): non-causal attention is invariant to a joint permutation of keys/values and linear in v."""
    from mop_amd import ops
    torch.manual_seed(0)
    B, T, H, dk = 2, 3000, 6, 64
    q, k, v, v2 = (torch.randn(B, T, H, dk, device="cuda", dtype=torch.bfloat16) for _ in range(4))
    with torch.no_grad():
        y = ops.sdpa_core(q, k, v).float()
        perm = torch.randperm(T, device="cuda")
        yp = ops.sdpa_core(q, k[:, perm].contiguous(), v[:, perm].contiguous()).float()
        ys = ops.sdpa_core(q, k, (v.float() + 2 * v2.float()).to(torch.bfloat16)).float()
        y2 = ops.sdpa_core(q, k, v2).float()
    assert float((y - yp).abs().max()) <= 2e-2                     # summation order changes with the tiling of the permuted keys
    assert float((ys - (y + 2 * y2)).abs().max()) <= 6e-2          # bf16 rounding of v + 2 v2 and of three outputs
    assert torch.isfinite(y).all()


def test_quartet_gpt_size_first_row_and_value_linearity():
    """T=1024, d=768, H=12 (config 4): row 0 sees only key 0 (y_0 = v_0); the output is linear in v; gradients are finite."""
    from mop_amd import ops
    torch.manual_seed(1)
    B, T, H, dh = 2, 1024, 12, 64
    q, k, v, q2, k2, v2 = (torch.randn(B, T, H, dh, device="cuda", dtype=torch.bfloat16) for _ in range(6))
    mix, qs = torch.tensor([0.2], device="cuda"), torch.tensor([0.9], device="cuda")
    with torch.no_grad():
        y = ops.quartet_core(q, k, v, q2, k2, mix, qs, None, 1e-5, True).float().view(B, T, H, dh)
        yb = ops.quartet_core(q, k, v2, q2, k2, mix, qs, None, 1e-5, True).float().view(B, T, H, dh)
        ys = ops.quartet_core(q, k, (v.float() - v2.float()).to(torch.bfloat16), q2, k2, mix, qs, None, 1e-5, True).float().view(B, T, H, dh)
    assert float((y[:, 0] - v[:, 0].float()).abs().max()) <= 2e-2
    assert float((ys - (y - yb)).abs().max()) <= 6e-2
    qg = q.clone().requires_grad_(True)
    ops.quartet_core(qg, k, v, q2, k2, mix, qs, None, 1e-5, True).sum().backward()
    assert torch.isfinite(qg.grad).all() and float(qg.grad.abs().max()) > 0


def test_multihop_full_size_is_deterministic():
    """N=197, d=384, H=6, B=32: two runs of the fused dual-path forward+backward are bit-identical (no atomics anywhere)."""
    from mop_amd.nn import MultiHopMSA
    torch.manual_seed(2)
    m = MultiHopMSA(384, 6).cuda().to(torch.bfloat16)
    x = torch.randn(32, 197, 384, device="cuda", dtype=torch.bfloat16)
    outs = []
    for _ in range(2):
        xg = x.clone().requires_grad_(True)
        m.zero_grad()
        y = m(xg)
        y.float().square().sum().backward()
        outs.append((y.detach().clone(), xg.grad.clone(), m.qkv2.weight.grad.clone()))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)


# ---- BASELINE.json config sizes: HIP vs the float64 oracle on full-length slices (B = 1, two heads) ----
def _bf16_np(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy().astype(np.float64)


def test_quartet_vs_oracle_at_gpt_sequence_length():
    """config 4 slice: T = 1024, dh = 64, two heads, causal dual-path z-normalised scores (quartet_attn_patch.py:88-120).
    The oracle gets the bf16-rounded operands the kernel reads, so the comparison isolates the kernel's arithmetic."""
    from oracle import quartet as oq
    from mop_amd import ops, _lib
    torch.manual_seed(4)
    B, T, H, dh = 1, 1024, 2, 64
    t = [torch.randn(B, T, H, dh, device="cuda").to(torch.bfloat16) for _ in range(5)]
    mix, qs = torch.tensor([0.3], device="cuda"), torch.tensor([0.8], device="cuda")
    dy = torch.randn(B, T, H * dh, device="cuda").to(torch.bfloat16)
    tg = [a.clone().requires_grad_(True) for a in t]
    mixg, qsg = mix.clone().requires_grad_(True), qs.clone().requires_grad_(True)
    y = ops.quartet_core(tg[0], tg[1], tg[2], tg[3], tg[4], mixg, qsg, None, 1e-5, True)
    y.backward(dy)
    assert ops.LAST_PATH["quartet_fwd"] == _lib.PATH_FUSED
    hp = lambda a: np.transpose(a.detach().float().cpu().numpy().astype(np.float64), (0, 2, 1, 3))     # (B,T,H,dh) -> (B,H,T,dh)
    q, k, v, q2, k2 = (hp(a) for a in t)
    yo, c = oq.core_fwd(q, k, v, q2, k2, float(mix), float(qs), 1e-5, True, True, None)
    g = oq.core_bwd(hp(dy.view(B, T, H, dh)), c)
    yk = hp(y.view(B, T, H, dh))
    assert max_abs(yk, yo) <= 1e-2 * max(1.0, float(np.abs(yo).max())), f"y {max_abs(yk, yo):.3e}"
    for got, ref, nm in ((tg[0].grad, g["dq"], "dq"), (tg[1].grad, g["dk"], "dk"), (tg[2].grad, g["dv"], "dv"),
                         (tg[3].grad, g["dq2"], "dq2"), (tg[4].grad, g["dk2"], "dk2")):
        assert rel_err(hp(got), ref) <= 3e-2, f"{nm} {rel_err(hp(got), ref):.3e}"
    assert abs(float(mixg.grad) - float(g["dmixture"])) <= 5e-2 * max(abs(float(g["dmixture"])), 1e-3 * float(np.abs(g["dv"]).max()) * T)
    assert abs(float(qsg.grad) - float(g["dquartet_scale"])) <= 5e-2 * max(abs(float(g["dquartet_scale"])), 1e-3 * float(np.abs(g["dv"]).max()) * T)


def test_multihop_vs_oracle_at_gpt_sequence_length():
    """config 4, second reading: the literal softmax(S1 + S2) kernel -- MultiHopMSA(768, 12) core with a causal mask at
    seq 1024 (attention_variants.py:200-229), here one sequence and two heads of dk = 64, against the float64 oracle on the bf16-rounded
    operands the kernel reads."""
    from oracle import multihop as om
    from mop_amd import ops, _lib
    torch.manual_seed(6)
    B, T, H, dk, hops = 1, 1024, 2, 64, 3
    t = [torch.randn(B, T, H, dk, device="cuda").to(torch.bfloat16) for _ in range(6)]
    dy = torch.randn(B, T, H * dk, device="cuda").to(torch.bfloat16)
    logit = torch.tensor(-2.0, device="cuda")
    tg = [a.clone().requires_grad_(True) for a in t]
    lg = logit.clone().requires_grad_(True)
    y = ops.dualpath_core(tg[0], tg[1], tg[2], tg[3], tg[4], tg[5], lg, 1.0, 0.0, 0.0, 0.0, 0.5, hops, None, True)
    y.backward(dy)
    assert ops.LAST_PATH["dualpath_fwd"] == _lib.PATH_FUSED
    hp = lambda a: np.transpose(a.detach().float().cpu().numpy().astype(np.float64), (0, 2, 1, 3))     # (B,T,H,dk) -> (B,H,T,dk)
    blocked = ~np.tril(np.ones((T, T), dtype=bool))
    yo, c = om.core_fwd(*(hp(a) for a in t), dict(and_=1.0, or_=0.0, not_=0.0, chain=0.0), 0.5, hops, -2.0, blocked)
    g = om.core_bwd(hp(dy.view(B, T, H, dk)), c)
    yk = hp(y.view(B, T, H, dk))
    assert max_abs(yk, yo) <= 1e-2 * max(1.0, float(np.abs(yo).max())), f"y {max_abs(yk, yo):.3e}"
    for got, nm in ((tg[0].grad, "dq1"), (tg[1].grad, "dk1"), (tg[2].grad, "dv1"), (tg[3].grad, "dq2"), (tg[4].grad, "dk2"), (tg[5].grad, "dv2")):
        assert rel_err(hp(got), g[nm]) <= 3e-2, f"{nm} {rel_err(hp(got), g[nm]):.3e}"
    assert abs(float(lg.grad) - float(g["dlogit"])) <= 5e-2 * max(abs(float(g["dlogit"])), 1e-3 * float(np.abs(g["dv1"]).max()) * T)


def test_sdpa_vs_oracle_at_whisper_sequence_length():
    """config 5 slice: T = 3000, dk = 64, two heads, non-causal SDPA (whisper_mop.py:137-177) against the float64 oracle."""
    from oracle import sdpa as osd
    from mop_amd import ops, _lib
    torch.manual_seed(5)
    B, T, H, dk = 1, 3000, 2, 64
    t = [torch.randn(B, T, H, dk, device="cuda").to(torch.bfloat16) for _ in range(3)]
    dy = torch.randn(B, T, H * dk, device="cuda").to(torch.bfloat16)
    tg = [a.clone().requires_grad_(True) for a in t]
    y = ops.sdpa_core(tg[0], tg[1], tg[2])
    y.backward(dy)
    assert ops.LAST_PATH["sdpa_fwd"] == _lib.PATH_FUSED
    hp = lambda a: np.transpose(a.detach().float().cpu().numpy().astype(np.float64), (0, 2, 1, 3))
    yo, c = osd.core_fwd(*(hp(a) for a in t))
    g = osd.core_bwd(hp(dy.view(B, T, H, dk)), c)
    assert max_abs(hp(y.view(B, T, H, dk)), yo) <= 1e-2
    for got, ref, nm in ((tg[0].grad, g["dq"], "dq"), (tg[1].grad, g["dk"], "dk"), (tg[2].grad, g["dv"], "dv")):
        assert rel_err(hp(got), ref) <= 3e-2, f"{nm} {rel_err(hp(got), ref):.3e}"


def test_empty_batch_returns_an_empty_result_like_the_torch_ops_of_the_reference():
    """B = 0: the reference's torch ops return an empty (0, N, D) tensor (and empty input gradients); the C ABI rejects empty shapes,
    so the core wrappers answer before calling it."""
    from mop_amd.nn import EdgewiseMSA, BaselineMSA, MultiHopMSA, CrossViewMixerMSA
    mods = [EdgewiseMSA(128, 4, n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2),
            EdgewiseMSA(128, 4, n_views=3, share_qkv=True, gate_mode="dense", use_k3=True),
            BaselineMSA(128, 4), MultiHopMSA(128, 4), CrossViewMixerMSA(128, 4)]
    for m in mods:
        m = m.cuda().to(torch.bfloat16)
        x = torch.zeros(0, 17, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        y = m(x)
        assert tuple(y.shape) == (0, 17, 128), type(m).__name__
        y.sum().backward()
        assert tuple(x.grad.shape) == (0, 17, 128)


def test_float16_modules_run_through_the_float32_arithmetic():
    """`module.half()`: the kernels take float32 / bfloat16; half tensors are computed in float32 (a superset) and cast back, so the
    result equals the float32 module's up to the fp16 rounding of parameters, activations and the output."""
    import mop_amd
    from mop_amd.nn import EdgewiseMSA, BaselineMSA, MultiHopMSA
    mop_amd.set_precision("auto")
    for ctor in (lambda: EdgewiseMSA(128, 4, n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2, gate_init="mix5"),
                 lambda: EdgewiseMSA(128, 4, n_views=3, share_qkv=True, gate_mode="dense"),
                 lambda: BaselineMSA(128, 4), lambda: MultiHopMSA(128, 4)):
        torch.manual_seed(11)
        m32 = ctor().cuda()
        x = torch.randn(2, 33, 128, device="cuda")
        y32 = m32(x)
        m16 = ctor().cuda()
        m16.load_state_dict(m32.state_dict())
        m16 = m16.half()
        xh = x.half().requires_grad_(True)
        y16 = m16(xh)
        assert y16.dtype == torch.float16
        y16.float().square().sum().backward()
        assert torch.isfinite(xh.grad).all() and xh.grad.dtype == torch.float16
        assert float((y16.detach().float() - y32.detach()).abs().max()) <= 2e-2 * max(1.0, float(y32.detach().abs().max())), type(m32).__name__
