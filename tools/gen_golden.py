#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation on CPU.

Run only in the build container (the reference tree is not shipped anywhere):

    PYTHONDONTWRITEBYTECODE=1 MOP_REFERENCE=/root/reference python tools/gen_golden.py

Imports `mop.models` from $MOP_REFERENCE, builds each module under a fixed seed,
perturbs degenerate initialisations (identical views under
share_qkv, zero low-rank biases, mixture=-5), runs forward and autograd backward
of L = sum(y * w) in float32 on CPU, and writes inputs / parameters / outputs /
gradients to tests/golden/<case>.npz.  Only data is written; no reference source.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REF = os.environ.get("MOP_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from mop.models.attention_variants import (BaselineMSA, CrossViewMixerMSA, EdgewiseMSA,  # noqa: E402
                                           MultiHopMSA)
from mop.models.quartet_attn_patch import (CausalSelfAttention,  # noqa: E402
                                           TransformerConfig)

from mop.models.whisper_mop import EncoderBlock, WhisperConfig  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _perturb(mod: torch.nn.Module, seed: int):
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if name.endswith(("q_scale", "k_scale", "v_scale")):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif "row_proj.weight" in name or "col_proj.weight" in name:
                p.mul_(3.0)
            elif "row_proj.bias" in name or "col_proj.bias" in name:
                p.add_(0.3 * torch.randn(p.shape, generator=g))
            elif name.endswith("conv2.bias"):                     # dense head: bias -5 leaves every gate (and its gradient) ~0
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif name.endswith("mixture"):
                p.fill_(0.3)
            elif name.endswith("quartet_scale"):
                p.fill_(0.8)
            elif name.endswith("chain_value_logit"):
                p.fill_(-0.5)


def _bf16_self_error(mod, x, w, fwd_kwargs, ref):
    """The reference module run end to end in bfloat16 (parameters, input, every op) against its own float32 run: max-abs error /
    max|fp32 value| per gradient tensor ('bf16err:<name>').  The bf16 parity tests use it as the noise floor of bf16 arithmetic on
    this case: an implementation cannot be asked to be closer to the fp32 gradients than the reference's own bf16 run is."""
    import copy
    mb = copy.deepcopy(mod).to(torch.bfloat16)
    xb = x.detach().clone().to(torch.bfloat16).requires_grad_(True)
    fwd_kwargs = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in fwd_kwargs.items()}
    yb = mb(xb, **fwd_kwargs)
    (yb * w.to(torch.bfloat16)).sum().backward()
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    out = {"bf16err:y": np.float32(np.abs(yb.detach().float().numpy() - ref["y"]).max()),
           "bf16err:dx": np.float32(rel(xb.grad.float().numpy(), ref["dx"]))}
    for k, p in mb.named_parameters():
        if p.grad is not None:
            out["bf16err:" + k] = np.float32(rel(p.grad.float().numpy(), ref["grad:" + k]))
    return out


def _run(mod, x, fwd_kwargs=None, extra=None, hook=None, bf16_self=False):
    fwd_kwargs = fwd_kwargs or {}
    x = x.clone().requires_grad_(True)
    inter = {}
    y = mod(x, **fwd_kwargs)
    g = torch.Generator().manual_seed(4242)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    out = {"x": x.detach().numpy(), "y": y.detach().numpy(), "w": w.numpy(),
           "dx": x.grad.numpy()}
    for k, v in mod.state_dict().items():
        out["param:" + k] = v.detach().numpy()
    for k, p in mod.named_parameters():
        out["grad:" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    if extra:
        out.update(extra)
    out.update(inter)
    if bf16_self:
        out.update(_bf16_self_error(mod, x, w, fwd_kwargs, out))
    return out


def _save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **d)
    ymax = f"|y|max={np.abs(d['y']).max():.4f}" if "y" in d else ""
    print(f"{name:40s} {os.path.getsize(path) / 1e6:7.3f} MB  {ymax}")


def edgewise_cases():
    cases = [
        # name, dim, heads, B, N, kwargs
        ("ew_tiny_shared_v2_r2_neutral", 64, 4, 2, 8, dict(n_views=2, share_qkv=True, gate_rank=2, gate_init="neutral")),
        ("ew_tiny_shared_v3_r4_mix5", 64, 4, 2, 8, dict(n_views=3, share_qkv=True, gate_rank=4, gate_init="mix5")),
        ("ew_tiny_unshared_v2_r4_xor", 64, 4, 2, 8, dict(n_views=2, share_qkv=False, gate_rank=4, gate_init="xor")),
        ("ew_tiny_unshared_v3_r2_and", 64, 4, 2, 8, dict(n_views=3, share_qkv=False, gate_rank=2, gate_init="and")),
        ("ew_odd_shared_v5_r4_mix5", 32, 2, 1, 6, dict(n_views=5, share_qkv=True, gate_rank=4, gate_init="mix5")),
        ("ew_mid_shared_v5_r4_chain", 128, 2, 2, 50, dict(n_views=5, share_qkv=True, gate_rank=4, gate_init="chain", beta_not=0.7)),
        ("ew_mid_shared_v4_r3_not", 96, 3, 3, 33, dict(n_views=4, share_qkv=True, gate_rank=3, gate_init="not")),
        ("ew_ns_shared_v5_r4_mix5", 384, 6, 1, 197, dict(n_views=5, share_qkv=True, gate_rank=4, gate_init="mix5")),
    ]
    for i, (name, dim, heads, B, N, kw) in enumerate(cases):
        torch.manual_seed(100 + i)
        mod = EdgewiseMSA(dim, heads, gate_mode="lowrank", **kw).eval()
        _perturb(mod, 100 + i)
        x = torch.randn(B, N, dim)
        meta = dict(kind="edgewise", dim=dim, heads=heads, beta_not=kw.get("beta_not", 0.5),
                    n_views=kw["n_views"], share_qkv=kw["share_qkv"], gate_rank=kw["gate_rank"])
        extra = {"meta:" + k: np.asarray(v) for k, v in meta.items()}
        _save(name, _run(mod, x, extra=extra, bf16_self=True))


def edgewise_variant_cases():
    """dense gate head (+use_k3), S lens bank, Q/K lens bank  (attention_variants.py:250-272, :392-442, :472-533)."""
    cases = [
        # name, dim, heads, B, N, kwargs
        ("ewx_tiny_dense_v2", 64, 4, 2, 8, dict(n_views=2, share_qkv=True, gate_mode="dense", gate_init="and")),
        ("ewx_tiny_dense_k3_v3", 64, 4, 2, 8, dict(n_views=3, share_qkv=True, gate_mode="dense", use_k3=True)),
        ("ewx_odd_dense_k3_unshared", 32, 2, 1, 6, dict(n_views=2, share_qkv=False, gate_mode="dense", use_k3=True, gate_init="xor")),
        ("ewx_mid_dense_k3_v4", 96, 3, 2, 33, dict(n_views=4, share_qkv=True, gate_mode="dense", use_k3=True, beta_not=0.7)),
        ("ewx_tiny_lowrank_lens", 64, 4, 2, 8, dict(n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2, gate_init="mix5",
                                                   use_lens_bank=True, lens_dilations=(1, 2))),
        ("ewx_tiny_lowrank_qklens_causal", 64, 4, 2, 8, dict(n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2,
                                                            use_lens_bank_qk=True, lens_qk_dilations=(1, 2, 3), lens_qk_causal=True)),
        ("ewx_tiny_lowrank_lens_qklens", 64, 4, 2, 8, dict(n_views=4, share_qkv=True, gate_mode="lowrank", gate_rank=2, gate_init="mix5",
                                                          use_lens_bank=True, lens_dilations=(1, 2), use_lens_bank_qk=True,
                                                          lens_qk_dilations=(2, 3), lens_qk_causal=True)),
        ("ewx_mid_dense_k3_lens_qk", 96, 3, 2, 33, dict(n_views=3, share_qkv=True, gate_mode="dense", use_k3=True,
                                                       use_lens_bank=True, lens_dilations=(1, 3), use_lens_bank_qk=True,
                                                       lens_qk_dilations=(1, 2), lens_qk_causal=False)),
    ]
    for i, (name, dim, heads, B, N, kw) in enumerate(cases):
        torch.manual_seed(500 + i)
        mod = EdgewiseMSA(dim, heads, **kw).eval()
        _perturb(mod, 500 + i)
        x = torch.randn(B, N, dim)
        meta = dict(kind="edgewise", dim=dim, heads=heads, beta_not=kw.get("beta_not", 0.5), n_views=kw["n_views"],
                    share_qkv=kw["share_qkv"], gate_rank=kw.get("gate_rank", 4), gate_mode=kw["gate_mode"],
                    use_k3=kw.get("use_k3", False), use_lens_bank=kw.get("use_lens_bank", False),
                    lens_dilations=np.asarray(kw.get("lens_dilations", ()), dtype=np.int64),
                    use_lens_bank_qk=kw.get("use_lens_bank_qk", False),
                    lens_qk_dilations=np.asarray(kw.get("lens_qk_dilations", ()), dtype=np.int64),
                    lens_qk_causal=kw.get("lens_qk_causal", False))
        extra = {"meta:" + k: np.asarray(v) for k, v in meta.items()}
        _save(name, _run(mod, x, extra=extra, bf16_self=True))


def crossview_cases():
    """CrossViewMixerMSA (attention_variants.py:51-156)."""
    cases = [
        # name, dim, heads, B, N, ctor kwargs, causal mask
        ("cv_tiny_default", 64, 4, 2, 8, dict(), False),
        ("cv_tiny_cues_fixed_prior", 64, 4, 2, 8, dict(t1=0.3, t2=-0.2, enable_per_key_prior=True, prior_weight=0.4,
                                                      anchor_mode="fixed", fixed_k_star=3), False),
        ("cv_mid_cues_causal_prior0", 96, 3, 2, 33, dict(t1=0.25, t2=0.1, enable_per_key_prior=True, prior_weight=0.6,
                                                        anchor_mode="first"), True),
        ("cv_mid_argmax_prior", 128, 2, 2, 50, dict(enable_per_key_prior=True, prior_weight=0.5), False),
        ("cv_odd_nocues", 32, 2, 1, 6, dict(use_transpose_cues=False, t1=0.7, t2=0.7), False),
    ]
    for i, (name, dim, heads, B, N, kw, causal) in enumerate(cases):
        torch.manual_seed(600 + i)
        mod = CrossViewMixerMSA(dim, heads, **kw).eval()
        with torch.no_grad():
            mod.mix.add_(0.3 * torch.randn(2, 2))                 # identity init makes S12/S21 and dmix off-diagonals untested
        x = torch.randn(B, N, dim)
        fk, extra = {}, {}
        if causal:
            mask = torch.tril(torch.ones(N, N)).view(1, 1, N, N)
            fk["attn_mask"] = mask
            extra["attn_mask"] = mask.numpy()
        meta = dict(kind="crossview", dim=dim, heads=heads, use_transpose_cues=kw.get("use_transpose_cues", True),
                    t1=kw.get("t1", 0.0), t2=kw.get("t2", 0.0), enable_per_key_prior=kw.get("enable_per_key_prior", False),
                    prior_weight=kw.get("prior_weight", 0.5), anchor_mode=kw.get("anchor_mode", "argmax_row_sum"),
                    fixed_k_star=kw.get("fixed_k_star", 0))
        extra.update({"meta:" + k: np.asarray(v) for k, v in meta.items()})
        if meta["enable_per_key_prior"] and meta["anchor_mode"] == "argmax_row_sum":
            with torch.no_grad():                                 # the anchor the reference picked (:139-140), rounding-noise dependent
                _, _, S2, _ = mod._compute_logits(x)
                extra["k_star"] = torch.softmax(mod._apply_mask(S2, fk.get("attn_mask")), -1).sum(-1).argmax(-1).numpy()
        _save(name, _run(mod, x, fk, extra, bf16_self=True))


def whisper_cases():
    """Whisper-MoP encoder block: non-causal MultiheadSelfAttention + MoP2D mel gate + MLP (whisper_mop.py:91-124, :137-177, :250-275)."""
    cases = [("wh_enc_tiny", 64, 4, 2, 12, 10, False), ("wh_enc_mid_bias", 96, 3, 2, 40, 16, True)]
    for i, (name, dim, heads, B, T, n_mels, bias) in enumerate(cases):
        torch.manual_seed(700 + i)
        cfg = WhisperConfig(n_mels=n_mels, n_audio_ctx=T, n_embd=dim, n_head=heads, n_layer_enc=1, n_layer_dec=1, bias=bias,
                            n_views=3, n_kernels=2, kernel_size=3)
        blk = EncoderBlock(cfg).eval()
        mel = torch.randn(B, 1, T, n_mels)

        class Wrap(torch.nn.Module):          # the block returns (x, gate); L = sum(x * w) sees both paths
            def __init__(self):
                super().__init__()
                self.blk = blk

            def forward(self, x):
                return self.blk(x, mel)[0]

        x = torch.randn(B, T, dim)
        meta = dict(kind="whisper_enc", dim=dim, heads=heads, n_mels=n_mels, bias=bias, n_views=3, n_kernels=2, kernel_size=3)
        extra = {"meta:" + k: np.asarray(v) for k, v in meta.items()}
        extra["mel"] = mel.numpy()
        _save(name, _run(Wrap(), x, extra=extra))


def multihop_cases():
    cases = [
        ("mh_tiny_default", 64, 4, 2, 8, dict(), False),
        ("mh_tiny_allgates_h2", 64, 4, 2, 8, dict(gates=dict(and_=0.7, or_=0.4, not_=0.3, chain=0.5), hops=2, beta_not=0.6), False),
        ("mh_mid_allgates_h3_causal", 96, 3, 2, 33, dict(gates=dict(and_=0.9, or_=0.5, not_=0.2, chain=0.3), hops=3), True),
        ("mh_mid_default_causal", 128, 2, 2, 50, dict(), True),
        ("mh_n197_default", 128, 2, 1, 197, dict(), False),
    ]
    for i, (name, dim, heads, B, N, kw, causal) in enumerate(cases):
        torch.manual_seed(200 + i)
        mod = MultiHopMSA(dim, heads, **kw).eval()
        _perturb(mod, 200 + i)
        x = torch.randn(B, N, dim)
        fk, extra = {}, {}
        if causal:
            mask = torch.tril(torch.ones(N, N)).view(1, 1, N, N)
            fk["attn_mask"] = mask
            extra["attn_mask"] = mask.numpy()
        g = kw.get("gates", dict(and_=1.0, or_=0.0, not_=0.0, chain=0.0))
        meta = dict(kind="multihop", dim=dim, heads=heads, beta_not=kw.get("beta_not", 0.5),
                    hops=kw.get("hops", 3), g_and=g["and_"], g_or=g["or_"], g_not=g["not_"],
                    g_chain=g["chain"])
        extra.update({"meta:" + k: np.asarray(v) for k, v in meta.items()})
        _save(name, _run(mod, x, fk, extra, bf16_self=True))


def quartet_cases():
    cases = [
        ("qt_tiny_quartet", 64, 4, 2, 16, True, False, False),
        ("qt_tiny_plain", 64, 4, 2, 16, False, False, False),
        ("qt_mid_quartet_bias_addmask", 96, 3, 2, 40, True, True, True),
        ("qt_slice_quartet", 128, 2, 1, 256, True, False, False),
    ]
    for i, (name, dim, heads, B, T, uq, bias, addmask) in enumerate(cases):
        torch.manual_seed(300 + i)
        cfg = TransformerConfig(n_head=heads, n_embd=dim, block_size=max(T, 16), dropout=0.0,
                                bias=bias, use_quartet=uq)
        mod = CausalSelfAttention(cfg).eval()
        _perturb(mod, 300 + i)
        x = torch.randn(B, T, dim)
        fk, extra = {}, {}
        if addmask:
            am = 0.5 * torch.randn(B, 1, T, T)
            fk["attention_mask"] = am
            extra["attention_mask"] = am.numpy()
        meta = dict(kind="quartet", dim=dim, heads=heads, use_quartet=uq, eps=cfg.score_norm_eps)
        extra.update({"meta:" + k: np.asarray(v) for k, v in meta.items()})
        _save(name, _run(mod, x, fk, extra, bf16_self=True))


def sdpa_cases():
    cases = [("sdpa_tiny", 64, 4, 2, 8, False), ("sdpa_mid_causal", 96, 3, 2, 33, True),
             ("sdpa_n196", 128, 2, 1, 196, False)]
    for i, (name, dim, heads, B, N, causal) in enumerate(cases):
        torch.manual_seed(400 + i)
        mod = BaselineMSA(dim, heads).eval()
        x = torch.randn(B, N, dim)
        fk, extra = {}, {}
        if causal:
            mask = torch.tril(torch.ones(N, N)).view(1, 1, N, N)
            fk["attn_mask"] = mask
            extra["attn_mask"] = mask.numpy()
        meta = dict(kind="sdpa", dim=dim, heads=heads)
        extra.update({"meta:" + k: np.asarray(v) for k, v in meta.items()})
        _save(name, _run(mod, x, fk, extra, bf16_self=True))


def vit_cases():
    """ViT_MoP end to end (vit_mop.py:84-140; reference tests/test_forward_shapes.py:12-28): logits, gate maps, dx and sampled
    parameter gradients.  Parameters come from tests/vit_fixture.py (numpy stream, not stored): BASELINE.json configs[0] is the
    5,397,972-parameter model (dim 384, depth 3, heads 6, 5 views) on 32x32 images (N = 64 tokens)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from vit_fixture import fill_params, grad_sample
    from mop.models.vit_mop import ViT_MoP
    cases = [("vit_tiny_d64", dict(dim=64, depth=2, heads=4, n_classes=10, n_views=3, n_kernels=2, drop_path=0.0), 3, 900),
             ("vit_cfg0_5m", dict(dim=384, depth=3, heads=6, n_classes=100, n_views=5, n_kernels=3, drop_path=0.0), 2, 901)]
    for name, kw, B, seed in cases:
        torch.manual_seed(seed)
        mod = ViT_MoP(**kw).eval()
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        vals = fill_params(shapes, seed)
        mod.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(shapes[k]) for k, v in vals.items()}, strict=True)
        x = torch.randn(B, 3, 32, 32).requires_grad_(True)
        y = mod(x)
        g = torch.Generator().manual_seed(4242)
        w = torch.randn(y.shape, generator=g)
        (y * w).sum().backward()
        gate, Vm, Km = mod.get_gate_maps(x.detach())
        out = {"x": x.detach().numpy(), "y": y.detach().numpy(), "w": w.numpy(), "dx": x.grad.numpy(),
               "gate": gate.numpy(), "views": Vm.numpy(), "kernels": Km.numpy()}
        for k, p in mod.named_parameters():
            smp, nrm = grad_sample(p.grad.numpy())
            out["gsample:" + k], out["gnorm:" + k] = smp, nrm
        for k, v in shapes.items():
            out["shape:" + k] = np.asarray(v, dtype=np.int64)
        meta = dict(kind="vit_mop", param_seed=seed, n_params=sum(p.numel() for p in mod.parameters()), **kw)
        out.update({"meta:" + k: np.asarray(v) for k, v in meta.items()})
        _save(name, out)


def train_cases():
    """k optimizer steps of the reference's training recipe on the reference's ViT_MoP: AdamW + LinearLR warm-up ->
    CosineAnnealingLR exactly as experiments/cifar100_ab5_param_budgets.py:464-479 builds them, the step loop of :793-804
    (zero_grad, cross-entropy, backward, opt.step, sched.step), and the checkpoint dictionary of mop/training/utils.py:120-144.
    Recorded: per-step batches, losses, learning rates, a strided sample + norm of every parameter after the last step, the
    checkpoint's keys."""
    import tempfile
    from torch import nn, optim
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from vit_fixture import fill_params, grad_sample
    from mop.models.vit_mop import ViT_MoP
    from mop.training.utils import save_checkpoint
    kw = dict(dim=64, depth=2, heads=4, n_classes=10, n_views=3, n_kernels=2, drop_path=0.0)
    steps, warmup_frac, lr, wd, B, seed = 6, 0.34, 3e-3, 5e-2, 4, 910
    torch.manual_seed(seed)
    mod = ViT_MoP(**kw).train()
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    vals = fill_params(shapes, seed)
    mod.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(shapes[k]) for k, v in vals.items()}, strict=True)
    opt = optim.AdamW(mod.parameters(), lr=lr, weight_decay=wd)                                   # :465
    warm = int(max(steps, 1) * max(warmup_frac, 0.0))                                             # :466
    sched = optim.lr_scheduler.SequentialLR(                                                      # :468-476
        opt, [optim.lr_scheduler.LinearLR(opt, start_factor=1e-3, total_iters=warm),
              optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max(steps - warm, 1))], milestones=[warm])
    g = torch.Generator().manual_seed(seed + 1)
    xs = torch.randn(steps, B, 3, 32, 32, generator=g)
    ys = torch.randint(0, kw["n_classes"], (steps, B), generator=g)
    losses, lrs = [], []
    for i in range(steps):                                                                        # :793-804
        lrs.append(opt.param_groups[0]["lr"])
        opt.zero_grad(set_to_none=True)
        loss = nn.functional.cross_entropy(mod(xs[i]), ys[i])
        loss.backward()
        opt.step()
        sched.step()
        losses.append(float(loss))
    out = {"x": xs.numpy(), "labels": ys.numpy(), "loss": np.asarray(losses, dtype=np.float64), "lr": np.asarray(lrs, dtype=np.float64)}
    for k, v in mod.state_dict().items():
        smp, nrm = grad_sample(v.detach().numpy())
        out["psample:" + k], out["pnorm:" + k] = smp, nrm
    for k, v in shapes.items():
        out["shape:" + k] = np.asarray(v, dtype=np.int64)
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "c.pt")
        save_checkpoint(mod, opt, 3, losses[-1], f)
        ck = torch.load(f, map_location="cpu")
    out["ckpt_keys"] = np.asarray(sorted(ck.keys()))
    out["ckpt_opt_keys"] = np.asarray(sorted(ck["optimizer_state_dict"].keys()))
    meta = dict(kind="train_vit_mop", param_seed=seed, steps=steps, warmup_frac=warmup_frac, lr=lr, weight_decay=wd, **kw)
    out.update({"meta:" + k: np.asarray(v) for k, v in meta.items()})
    _save("train_vit_tiny_adamw6", out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    groups = dict(train=train_cases, vit=vit_cases, ew=edgewise_cases, ewx=edgewise_variant_cases, cv=crossview_cases, wh=whisper_cases, mh=multihop_cases, qt=quartet_cases, sdpa=sdpa_cases)
    for name in (sys.argv[1:] or list(groups)):               # e.g. `gen_golden.py ewx` regenerates one group only
        groups[name]()
