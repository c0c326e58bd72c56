"""Host-side surface (no GPU): constructor signatures, state_dict keys/shapes and init presets
match what the reference produced (golden param sets came from the reference's own modules)."""
import inspect
import math

import numpy as np
import pytest
import os

import torch

from conftest import golden_names, load_golden


@pytest.mark.parametrize("name", golden_names("ew_"))
def test_edgewise_state_dict_matches_reference(name):
    from mop_amd.nn import EdgewiseMSA
    d, params, gref, meta = load_golden(name)
    m = EdgewiseMSA(meta["dim"], meta["heads"], n_views=meta["n_views"], share_qkv=bool(meta["share_qkv"]),
                    gate_mode="lowrank", gate_rank=meta["gate_rank"])
    sd = m.state_dict()
    assert set(sd) == set(params)
    for k, v in params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)


def test_edgewise_ctor_signature_is_the_reference_one():
    from mop_amd.nn import EdgewiseMSA
    names = list(inspect.signature(EdgewiseMSA.__init__).parameters)[1:]
    assert names == ["dim", "heads", "attn_drop", "proj_drop", "beta_not", "use_k3", "n_views", "share_qkv",
                     "gate_mode", "gate_rank", "gate_init", "use_lens_bank", "lens_kernel_size", "lens_dilations",
                     "use_lens_bank_qk", "lens_qk_kernel_size", "lens_qk_dilations", "lens_qk_causal"]
    d = {k: p.default for k, p in inspect.signature(EdgewiseMSA.__init__).parameters.items() if k != "self"}
    assert d["heads"] == 4 and d["beta_not"] == 0.5 and d["n_views"] == 2 and d["gate_mode"] == "dense"
    assert d["gate_rank"] == 4 and d["gate_init"] == "neutral" and d["share_qkv"] is False


@pytest.mark.parametrize("init,gates", [("and", [0]), ("or", [1]), ("not", [2]), ("chain", [3]), ("nor", [2]),
                                         ("xor", [1]), ("mix5", [0, 1, 2]), ("neutral", [])])
def test_lowrank_gate_presets(init, gates):
    from mop_amd.nn import EdgewiseGateHead
    r = 4
    h = EdgewiseGateHead(12, gate_mode="lowrank", gate_rank=r, gate_init=init)
    c = math.sqrt(2.0 / r)
    exp = np.zeros(4 * r, dtype=np.float32)
    for g in gates:
        exp[g * r:(g + 1) * r] = c
    np.testing.assert_allclose(h.row_proj.bias.detach().numpy(), exp, rtol=1e-6)
    np.testing.assert_allclose(h.col_proj.bias.detach().numpy(), exp, rtol=1e-6)


def test_dense_head_params_and_presets():
    from mop_amd.nn import EdgewiseMSA
    m = EdgewiseMSA(64, 4, use_k3=True, n_views=3, gate_mode="dense", gate_init="or")
    sd = m.state_dict()
    assert tuple(sd["edge_head.conv1.weight"].shape) == (16, 8, 1, 1)
    assert tuple(sd["edge_head.mid3.weight"].shape) == (16, 16, 3, 3)
    assert tuple(sd["edge_head.conv2.weight"].shape) == (4, 16, 1, 1)
    np.testing.assert_allclose(sd["edge_head.conv2.bias"].numpy(), [-5, 2, -5, -5])
    assert "qkv_list.2.weight" in sd and "qkv.weight" not in sd


def test_n_views_floor_and_lens_validation():
    from mop_amd.nn import EdgewiseMSA
    assert EdgewiseMSA(32, 2, n_views=1, gate_mode="lowrank").n_views == 2
    with pytest.raises(ValueError):
        EdgewiseMSA(32, 2, use_lens_bank_qk=True, share_qkv=False)
    m = EdgewiseMSA(64, 4, share_qkv=True, gate_mode="lowrank", gate_rank=2, use_lens_bank=True,
                    use_lens_bank_qk=True, lens_qk_causal=True)
    sd = m.state_dict()
    assert tuple(sd["q_lens.0.weight"].shape) == (16, 1, 3) and tuple(sd["lens_bank.1.weight"].shape) == (2, 1, 3, 3)
    assert tuple(sd["edge_head.row_proj.weight"].shape) == (8, 2 * 2 + 2 + 2 * 2, 1)


def test_unified_switch():
    from mop_amd.nn import BaselineMSA, CrossViewMixerMSA, EdgewiseMSA, MultiHopMSA, UnifiedMSA
    assert isinstance(UnifiedMSA("a", 64, 4).impl, BaselineMSA)
    assert isinstance(UnifiedMSA("B", 64, 4).impl, BaselineMSA)
    assert isinstance(UnifiedMSA("C", 64, 4).impl, CrossViewMixerMSA)
    assert isinstance(UnifiedMSA("D", 64, 4, hops=2).impl, MultiHopMSA)
    e = UnifiedMSA("E", 64, 4, n_views=5, share_qkv=True, gate_mode="lowrank", gate_rank=4, gate_init="mix5")
    assert isinstance(e.impl, EdgewiseMSA) and "impl.q_scale" in e.state_dict()
    with pytest.raises(ValueError):
        UnifiedMSA("Z", 64)


def test_unsupported_variants_raise_instead_of_falling_back():
    from mop_amd.nn import EdgewiseMSA
    x = torch.randn(1, 8, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # masked Edgewise (an extension, generic path) -- still never on the CPU
        EdgewiseMSA(64, 4, gate_mode="lowrank", share_qkv=True)(x, attn_mask=torch.ones(8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # ... with attn_drop in training mode too (the generic path carries dropout)
        EdgewiseMSA(64, 4, attn_drop=0.1, gate_mode="lowrank", share_qkv=True).train()(x, attn_mask=torch.ones(8, 8))
    with pytest.raises(ValueError):      # reference: torch.stack fails for lens_kernel_size != 3 (:534)
        EdgewiseMSA(64, 4, gate_mode="lowrank", share_qkv=True, use_lens_bank=True, lens_kernel_size=5)(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # dense head is built, but never on the CPU: no fallback path
        EdgewiseMSA(64, 4, gate_mode="dense")(x)


def test_attention_dropout_in_training_is_never_silently_skipped():
    """the reference applies attn_drop to the attention weights (attention_variants.py:45, :153, :222, :552).  Every path of the library
    carries it (tests/test_gpu_dropout.py); a module in training mode with attn_drop > 0 therefore reaches the library (and, on a CPU
    tensor, its loud no-fallback error) instead of quietly training a different model."""
    from mop_amd.nn import EdgewiseMSA
    m = EdgewiseMSA(64, 4, attn_drop=0.1, gate_mode="dense", use_k3=True).train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 8, 64))


def test_dropout_mask_restatement_matches_the_library():
    """`ops.dropout_keep_mask` (numpy) == `mopk_dropout_keep` (the C function the kernels share their hash with), and its keep rate"""
    from mop_amd import _lib as L, ops
    seed, p, B, H, N = 0x1234_5678_9ABC_DEF, 0.3, 2, 3, 37
    m = ops.dropout_keep_mask(seed, p, B, H, N)
    lib = L.lib()
    for bh, i, j in ((0, 0, 0), (5, 36, 1), (3, 17, 36), (1, 2, 3), (4, 30, 30)):
        assert bool(m[bh // H, bh % H, i, j]) == bool(lib.mopk_dropout_keep(seed, p, bh, i, j))
    big = ops.dropout_keep_mask(7, 0.1, 4, 4, 256).float()
    assert abs(float(big.mean()) - 0.9) < 2e-3
    assert abs(float(big.mean(dim=-1).std()) - (0.1 * 0.9 / 256) ** 0.5) < 5e-3          # rows are independent draws
    assert bool(ops.dropout_keep_mask(7, 0.0, 1, 1, 8).all()) and lib.mopk_dropout_keep(7, 0.0, 0, 1, 2) == 1


def test_causal_mask_detection_is_keyed_by_tensor_identity_not_address():
    """a non-causal mask allocated at the address of a freed causal one must not inherit its cached verdict"""
    from mop_amd.nn.attention_variants import _is_causal_mask
    n = 16
    verdicts = []
    for i in range(8):                       # same shape, alternating content; CPU allocations are recycled as freely as device ones
        m = torch.ones(1, 1, n, n).tril_() if i % 2 == 0 else torch.ones(1, 1, n, n)
        verdicts.append(_is_causal_mask(m, n))
        del m
    assert verdicts == [True, False] * 4
    m = torch.ones(n, n).tril_()
    assert _is_causal_mask(m, n)
    m[0, n - 1] = 1                           # in-place edit bumps the version: re-validated
    assert not _is_causal_mask(m, n)


@pytest.mark.parametrize("kw", [
    dict(gate_mode="dense", use_k3=True, n_views=3, share_qkv=True),
    dict(gate_mode="lowrank", gate_rank=2, n_views=3, share_qkv=True, use_lens_bank=True, lens_dilations=(1, 2)),
    dict(gate_mode="dense", n_views=4, share_qkv=True, use_lens_bank=True, lens_dilations=(1, 2), use_lens_bank_qk=True,
         lens_qk_dilations=(2, 3), lens_qk_causal=True),
])
def test_variant_state_dict_matches_golden_layout(kw):
    """dense head / lens bank parameter names and shapes = the reference's (tests/golden ewx_* fixtures hold its state_dict)."""
    from mop_amd.nn import EdgewiseMSA
    m = EdgewiseMSA(64, 4, **kw)
    sd = m.state_dict()
    eh = "edge_head."
    n_s = len(kw["lens_qk_dilations"]) if kw.get("use_lens_bank_qk") else kw["n_views"]
    C = 2 * n_s + 2 + (n_s * len(kw["lens_dilations"]) if kw.get("use_lens_bank") else 0)
    if kw["gate_mode"] == "dense":
        assert sd[eh + "conv1.weight"].shape == (16, C, 1, 1) and sd[eh + "conv2.weight"].shape == (4, 16, 1, 1)
        assert (eh + "mid3.weight" in sd) == bool(kw.get("use_k3"))
        assert float(sd[eh + "conv2.bias"][0]) == -5.0
    else:
        assert sd[eh + "row_proj.weight"].shape == (4 * kw["gate_rank"], C, 1)
    if kw.get("use_lens_bank"):
        assert sd["lens_bank.1.weight"].shape == (n_s, 1, 3, 3)
    if kw.get("use_lens_bank_qk"):
        assert sd["q_lens.0.weight"].shape == (16, 1, 3) and sd["k_lens.1.weight"].shape == (16, 1, 3)


@pytest.mark.parametrize("name", golden_names("wh_"))
def test_whisper_encoder_block_state_dict(name):
    from mop_amd.nn import EncoderBlock, WhisperConfig
    d, params, gref, meta = load_golden(name)
    cfg = WhisperConfig(n_mels=meta["n_mels"], n_embd=meta["dim"], n_head=meta["heads"], bias=bool(meta["bias"]),
                        n_views=meta["n_views"], n_kernels=meta["n_kernels"], kernel_size=meta["kernel_size"])
    sd = EncoderBlock(cfg).state_dict()
    ref = {k[len("blk."):]: v for k, v in params.items()}
    assert set(sd) == set(ref)
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k


def test_vit_edgewise_state_dict_layout_and_checkpoint_roundtrip(tmp_path):
    """parameter names of the reference's ViTEdgewise (experiments/cifar100_edgewise_gates.py:377-452) and its checkpoint dict."""
    from mop_amd.nn import ViTEdgewise
    from mop_amd.training import load_checkpoint, make_optimizer_and_schedule, save_checkpoint
    m = ViTEdgewise(dim=64, depth=2, heads=4, n_classes=10, n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2)
    keys = set(m.state_dict())
    for k in ("patch.proj.weight", "pos", "ln_f.weight", "head.weight", "blocks.1.ln1.bias", "blocks.0.mlp.fc1.weight",
              "blocks.1.attn.qkv.weight", "blocks.0.attn.edge_head.row_proj.weight", "blocks.0.attn.chain_value_logit"):
        assert k in keys, k
    assert m.pos.shape == (1, 64, 64) and m.blocks[1].dp1.drop_prob > m.blocks[0].dp1.drop_prob == 0.0
    opt, sched = make_optimizer_and_schedule(m, lr=3e-3, weight_decay=5e-2, steps=10, warmup_frac=0.2)
    lrs = []
    for _ in range(10):
        opt.step(); sched.step(); lrs.append(opt.param_groups[0]["lr"])
    assert lrs[0] < lrs[1] <= 3e-3 + 1e-12 and lrs[-1] < lrs[2]          # linear warm-up, then cosine decay
    path = str(tmp_path / "ck.pt")
    save_checkpoint(m, opt, 3, 0.5, path)
    m2 = ViTEdgewise(dim=64, depth=2, heads=4, n_classes=10, n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2)
    ck = load_checkpoint(m2, None, path)
    assert ck["epoch"] == 3 and set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_training_recipe_matches_the_reference_fixture(tmp_path):
    """learning-rate sequence of the reference recipe (experiments/cifar100_ab5_param_budgets.py:464-479) and the checkpoint
    dictionary of mop/training/utils.py:120-144, as recorded from the reference in tests/golden/train_vit_tiny_adamw6.npz"""
    import numpy as np
    from conftest import GOLDEN
    from mop_amd.training import make_optimizer_and_schedule, save_checkpoint
    d = np.load(os.path.join(GOLDEN, "train_vit_tiny_adamw6.npz"))
    m = torch.nn.Linear(4, 4)
    opt, sched = make_optimizer_and_schedule(m, lr=float(d["meta:lr"]), weight_decay=float(d["meta:weight_decay"]),
                                             steps=int(d["meta:steps"]), warmup_frac=float(d["meta:warmup_frac"]))
    lrs = []
    for _ in range(int(d["meta:steps"])):
        lrs.append(opt.param_groups[0]["lr"])
        m(torch.ones(1, 4)).sum().backward()
        opt.step(); sched.step()
    assert np.allclose(lrs, d["lr"], rtol=1e-12, atol=0.0), (lrs, d["lr"])
    assert opt.defaults["weight_decay"] == float(d["meta:weight_decay"]) and type(opt).__name__ == "AdamW"
    path = str(tmp_path / "c.pt")
    save_checkpoint(m, opt, 3, 0.25, path)
    ck = torch.load(path, map_location="cpu")
    assert sorted(ck) == list(d["ckpt_keys"]) and sorted(ck["optimizer_state_dict"]) == list(d["ckpt_opt_keys"])


def test_token_linear_matches_nn_linear_forward_and_gradients():
    """TokenLinear = nn.Linear (same keys, same forward); its weight gradient is a batched GEMM over row slices summed in fp32."""
    import torch.nn as nn
    from mop_amd.nn.linear import TokenLinear, _TokenLinearFn, weight_grad_splits
    assert weight_grad_splits(256 * 197, 1152) == 16 and weight_grad_splits(256 * 197, 384) == 32
    assert weight_grad_splits(1000, 384) == 1 and weight_grad_splits(4097, 384) == 1 and weight_grad_splits(197 * 64, 64) == 32
    torch.manual_seed(0)
    ref, lin = nn.Linear(48, 64, bias=True), TokenLinear(48, 64, bias=True)
    lin.load_state_dict(ref.state_dict())
    assert set(lin.state_dict()) == {"weight", "bias"} and isinstance(lin, nn.Linear)
    x = torch.randn(8, 1024, 48, requires_grad=True)           # 8192 rows -> 16 slices... (cap 32: 8192 = 2^13 -> 32)
    x2 = x.detach().clone().requires_grad_(True)
    w = torch.randn(8, 1024, 64)
    (ref(x) * w).sum().backward()
    y = _TokenLinearFn.apply(x2, lin.weight, lin.bias)         # the CUDA forward path, exercised on the CPU
    (y * w).sum().backward()
    assert torch.equal(y, ref(x2))
    assert torch.allclose(x2.grad, x.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(lin.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(lin.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-4)
    assert torch.equal(lin(x2), ref(x2))                       # CPU tensors take the plain F.linear branch


@pytest.mark.parametrize("name", golden_names("vit_"))
def test_vit_mop_state_dict_matches_the_reference_layout(name):
    """same keys, shapes and ORDER as the reference's ViT_MoP state_dict (recorded in the fixture); 5,397,972 parameters at configs[0]"""
    from mop_amd.nn import ViT_MoP
    d, _, _, meta = load_golden(name)
    shapes = {k[6:]: tuple(int(v) for v in d[k]) for k in d if k.startswith("shape:")}
    m = ViT_MoP(dim=int(meta["dim"]), depth=int(meta["depth"]), heads=int(meta["heads"]), n_classes=int(meta["n_classes"]),
                n_views=int(meta["n_views"]), n_kernels=int(meta["n_kernels"]), drop_path=0.0)
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(ours.items()) == list(shapes.items())
    assert sum(p.numel() for p in m.parameters()) == int(meta["n_params"])


@pytest.mark.parametrize("shape", [(2, 13, 3, 8, 4, (1, 2)), (1, 5, 2, 4, 3, (1, 7, 5)), (2, 33, 2, 16, 2, (3,)), (1, 4, 1, 4, 2, (4, 2))])
def test_lens_mean_features_closed_form(shape):
    """S lens bank row / column means (reference attention_variants.py:523-533 followed by the head's means :323-326) in closed form:
    ops.lens_mean_features against the planes convolved with F.conv2d, and the hand-written forward / backward the fused route uses
    (ops._lens_means_fwd / _lens_means_bwd) against autograd of the same."""
    import torch.nn.functional as F
    from mop_amd import ops
    B, N, H, dk, V, dil = shape
    torch.manual_seed(N)
    qkv = torch.randn(B, N, 1, 3, H, dk, dtype=torch.float64, requires_grad=True)
    sqk = torch.randn(V, H, dk, dtype=torch.float64, requires_grad=True)
    lw = torch.randn(len(dil), V, 3, 3, dtype=torch.float64, requires_grad=True)
    q, k = qkv[:, :, 0, 0].permute(0, 2, 1, 3), qkv[:, :, 0, 1].permute(0, 2, 1, 3)
    S = torch.einsum("bhid,vhd,bhjd->bhvij", q, sqk, k).reshape(B * H, V, N, N)
    planes = [F.conv2d(S, lw[l][:, None], padding=d, dilation=d, groups=V) for l, d in enumerate(dil)]
    r0 = torch.cat([o.mean(3).view(B, H, V, N) for o in planes], 2)
    c0 = torch.cat([o.mean(2).view(B, H, V, N) for o in planes], 2)
    gr, gc = torch.randn_like(r0), torch.randn_like(c0)
    g0 = torch.autograd.grad((r0 * gr).sum() + (c0 * gc).sum(), (qkv, sqk, lw))
    orig_float = torch.Tensor.float
    torch.Tensor.float = lambda t: t.double()            # keep the check in float64 (the product code computes in float32)
    try:
        r1, c1 = ops.lens_mean_features(qkv, sqk, lw, dil)
        r2, c2, st = ops._lens_means_fwd(qkv.detach(), sqk.detach(), lw.detach(), dil)
        dq, dk_, dsqk, dlw = ops._lens_means_bwd(gr, gc, qkv.detach(), sqk.detach(), lw.detach(), dil, st)
    finally:
        torch.Tensor.float = orig_float
    for a_, b_ in ((r1, r0), (c1, c0), (r2, r0), (c2, c0)):
        assert float((a_ - b_).detach().abs().max()) <= 1e-10 * max(1.0, float(b_.detach().abs().max()))
    assert float((dq.permute(0, 2, 1, 3) - g0[0][:, :, 0, 0]).abs().max()) <= 1e-9
    assert float((dk_.permute(0, 2, 1, 3) - g0[0][:, :, 0, 1]).abs().max()) <= 1e-9
    assert float((dsqk - g0[1]).abs().max()) <= 1e-9 and float((dlw - g0[2]).abs().max()) <= 1e-9
