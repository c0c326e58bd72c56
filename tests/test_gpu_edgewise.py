"""MI355X parity of the EdgewiseMSA kernels (through the nn.Module -> ctypes -> C ABI path)
against golden vectors produced by the reference itself, and against the numpy oracle on
fresh seeded inputs.  Tolerances are BASELINE.json's: <=1e-3 (fp32 arithmetic), <=1e-2 (bf16)."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, ref_bf16_error
from gpu_util import check_grads, max_abs, module_from_golden, oracle_bf16_noise, rel_err, run_fwd_bwd

pytestmark = pytest.mark.gpu

TOL_FP32, TOL_BF16 = 1e-3, 1e-2          # max-abs on y (north_star)
GTOL_FP32, GTOL_BF16 = 1e-3, 3e-2          # gradients: max-abs / max|ref| (bf16 bound is ours; north_star bounds y only)


def _ctor(meta):
    return dict(dim=meta["dim"], heads=meta["heads"], n_views=meta["n_views"], share_qkv=bool(meta["share_qkv"]),
                gate_mode="lowrank", gate_rank=meta["gate_rank"], beta_not=meta["beta_not"])


@pytest.fixture(autouse=True)
def _reset():
    import mop_amd
    from mop_amd import ops
    yield
    mop_amd.set_precision("auto")
    ops.set_path("auto")


@pytest.mark.parametrize("path", ["generic", "auto"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", golden_names("ew_"))
def test_edgewise_vs_reference_golden(name, prec, path):
    import mop_amd
    from mop_amd import ops
    from mop_amd.nn import EdgewiseMSA
    d, params, gref, meta = load_golden(name)
    mop_amd.set_precision(prec)
    ops.set_path(path)
    m = module_from_golden(EdgewiseMSA, params, **_ctor(meta))
    y, dx, grads = run_fwd_bwd(m, d["x"], d["w"])
    tol, gtol = (TOL_FP32, GTOL_FP32) if prec == "fp32" else (TOL_BF16, GTOL_BF16)
    assert max_abs(y, d["y"]) <= tol, f"y max-abs {max_abs(y, d['y']):.3e}"
    assert rel_err(dx, d["dx"]) <= gtol, f"dx rel {rel_err(dx, d['dx']):.3e}"
    assert set(grads) == set(gref)
    if prec == "fp32":
        for k in gref:
            assert rel_err(grads[k].reshape(gref[k].shape), gref[k]) <= gtol, f"grad {k}: rel {rel_err(grads[k].reshape(gref[k].shape), gref[k]):.3e}"
    else:       # 3e-2 per tensor, or NOISE_FACTOR x the error of the REFERENCE's own all-bf16 run where bf16 cannot resolve the tensor (gpu_util.check_grads)
        check_grads(grads, gref, gtol, d=d, noise_factor=NOISE_FACTOR.get(name, 2.0))


# Multiple of the reference's own all-bf16 error (one sample of rounding noise, recorded per tensor in the fixture) a bf16 run may show on
# a gradient tensor.  At the benchmark shape (N = 197) the kernels are held to that error itself: the gate-head gradients there are 19-32 %
# off in the reference's bf16 run, so a factor above 1 would accept almost anything.  The N = 6 case is the one where the fused path is
# still noisier than two samples of that noise explain (row_proj 5.7e-2 against 1.9e-2 recorded, generic path 1.3e-2): DESIGN section 5.
NOISE_FACTOR = {"ew_ns_shared_v5_r4_mix5": 1.0, "ew_odd_shared_v5_r4_mix5": 4.0}


def test_fused_gate_gradients_vs_generic_bf16():
    """The fused backward against the repo's own bf16 baseline (the generic path: same bf16 MFMA inputs, every N x N map in fp32 between the
    kernels) at the benchmark shape: per gate-head tensor the fused error may be at most twice the generic path's."""
    import mop_amd
    from mop_amd import ops, _lib
    from mop_amd.nn import EdgewiseMSA
    d, params, gref, meta = load_golden("ew_ns_shared_v5_r4_mix5")
    mop_amd.set_precision("bf16")
    errs = {}
    for path in ("generic", "auto"):
        ops.set_path(path)
        m = module_from_golden(EdgewiseMSA, params, **_ctor(meta))
        _, _, grads = run_fwd_bwd(m, d["x"], d["w"])
        errs[path] = {k: rel_err(grads[k].reshape(gref[k].shape), gref[k]) for k in gref if "edge_head" in k}
    assert ops.LAST_PATH["edgewise_bwd"] == _lib.PATH_FUSED
    for k in errs["auto"]:
        assert errs["auto"][k] <= 2.0 * max(errs["generic"][k], 1e-2), f"{k}: fused {errs['auto'][k]:.3e} vs generic {errs['generic'][k]:.3e}"


def _ctor_variant(meta):
    kw = dict(dim=meta["dim"], heads=meta["heads"], n_views=meta["n_views"], share_qkv=bool(meta["share_qkv"]),
              gate_mode=meta["gate_mode"], gate_rank=meta["gate_rank"], beta_not=meta["beta_not"], use_k3=bool(meta["use_k3"]))
    if meta["use_lens_bank"]:
        kw.update(use_lens_bank=True, lens_dilations=tuple(int(v) for v in meta["lens_dilations"]))
    if meta["use_lens_bank_qk"]:
        kw.update(use_lens_bank_qk=True, lens_qk_dilations=tuple(int(v) for v in meta["lens_qk_dilations"]),
                  lens_qk_causal=bool(meta["lens_qk_causal"]))
    return kw


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", golden_names("ewx_"))
def test_edgewise_variants_vs_reference_golden(name, prec):
    """dense gate head (+use_k3), S lens bank, Q/K lens bank: reference outputs and autograd gradients."""
    import mop_amd
    from mop_amd import ops, _lib
    from mop_amd.nn import EdgewiseMSA
    d, params, gref, meta = load_golden(name)
    mop_amd.set_precision(prec)
    m = module_from_golden(EdgewiseMSA, params, **_ctor_variant(meta))
    y, dx, grads = run_fwd_bwd(m, d["x"], d["w"])
    tol, gtol = (TOL_FP32, GTOL_FP32) if prec == "fp32" else (TOL_BF16, GTOL_BF16)
    assert max_abs(y, d["y"]) <= tol, f"y max-abs {max_abs(y, d['y']):.3e}"
    assert rel_err(dx, d["dx"]) <= gtol, f"dx rel {rel_err(dx, d['dx']):.3e}"
    check_grads(grads, gref, gtol, floor=1e-3 if prec == "fp32" else 1e-2, d=d if prec == "bf16" else None)
    if meta["gate_mode"] == "dense" or meta["use_lens_bank"]:
        # on the fused bf16 kernels (shared qkv, no Q/K lens bank): the plain dense head (no 3x3, no lens bank) and the low-rank head with
        # the S lens bank (its planes enter as row / column means); every other variant on the generic path
        shared = bool(meta["share_qkv"]) and not meta["use_lens_bank_qk"]
        plain = meta["gate_mode"] == "dense" and not meta["use_k3"] and not meta["use_lens_bank"]
        lr_lens = meta["gate_mode"] == "lowrank" and meta["use_lens_bank"]
        want = _lib.PATH_FUSED if (shared and (plain or lr_lens) and prec == "bf16") else _lib.PATH_GENERIC
        assert ops.LAST_PATH["edgewise_fwd"] == want and ops.LAST_PATH["edgewise_bwd"] == want


@pytest.mark.parametrize("shape", [(3, 17, 64, 4, 3, 2), (2, 64, 128, 2, 5, 4), (1, 197, 128, 2, 5, 4),
                                   (2, 100, 64, 4, 2, 1), (1, 224, 64, 1, 4, 8), (1, 1, 32, 2, 2, 2)])
def test_edgewise_vs_oracle_seeded(shape):
    """fresh inputs: HIP (fp32 arithmetic) vs the numpy oracle in float64."""
    from oracle import edgewise as oe
    import mop_amd
    from mop_amd.nn import EdgewiseMSA
    B, N, D, H, V, r = shape
    mop_amd.set_precision("fp32")
    torch.manual_seed(B * 1000 + N)
    m = EdgewiseMSA(D, H, n_views=V, share_qkv=True, gate_mode="lowrank", gate_rank=r, gate_init="mix5")
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("_scale"):
                p.add_(0.1 * torch.randn_like(p))
            if "proj.weight" in n_ and "edge_head" in n_:
                p.mul_(3.0)
    params = {k: v.detach().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    x = torch.randn(B, N, D).numpy()
    w = torch.randn(B, N, D).numpy()
    out, cache = oe.module_fwd(x.astype(np.float64), params, H, V, True, 0.5)
    dx_ref, g_ref = oe.module_bwd(w.astype(np.float64), cache)
    y, dx, grads = run_fwd_bwd(m.cuda().eval(), x, w)
    assert max_abs(y, out) <= 1e-4
    assert rel_err(dx, dx_ref) <= 1e-3
    for k in g_ref:
        assert rel_err(grads[k].reshape(g_ref[k].shape), g_ref[k]) <= 1e-3, k


def test_bf16_tensors_end_to_end():
    """bf16 I/O + bf16 MFMA arithmetic stays within the bf16 tolerance of the fp32 reference."""
    import mop_amd
    from mop_amd.nn import EdgewiseMSA
    d, params, gref, meta = load_golden("ew_ns_shared_v5_r4_mix5")
    mop_amd.set_precision("auto")
    m = module_from_golden(EdgewiseMSA, params, **_ctor(meta))
    y, dx, grads = run_fwd_bwd(m, d["x"], d["w"], dtype=torch.bfloat16)
    assert max_abs(y, d["y"]) <= TOL_BF16
    assert rel_err(dx, d["dx"]) <= 3e-2
    check_grads(grads, gref, GTOL_BF16, d=d, noise_factor=1.0)


def test_cpu_tensor_fails_loudly():
    from mop_amd.nn import EdgewiseMSA
    m = EdgewiseMSA(64, 4, n_views=2, share_qkv=True, gate_mode="lowrank")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 8, 64))


def test_deterministic_bitwise():
    """fixed reduction order: two runs give identical bits (fwd and grads)."""
    import mop_amd
    from mop_amd.nn import EdgewiseMSA
    d, params, gref, meta = load_golden("ew_mid_shared_v5_r4_chain")
    mop_amd.set_precision("bf16")
    m = module_from_golden(EdgewiseMSA, params, **_ctor(meta))
    r1 = run_fwd_bwd(m, d["x"], d["w"])
    m.zero_grad()
    r2 = run_fwd_bwd(m, d["x"], d["w"])
    assert np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1])
    for k in r1[2]:
        assert np.array_equal(r1[2][k], r2[2][k]), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", [n for n in golden_names("ew_") if "unshared" not in n])
def test_fused_forward_vs_reference_golden(name, dtype):
    """forward-only (no_grad) takes the fused gfx950 kernel; bf16 MFMA tolerance 1e-2."""
    import ctypes as C
    import mop_amd
    from mop_amd import ops
    from mop_amd.nn import EdgewiseMSA
    d, params, gref, meta = load_golden(name)
    mop_amd.set_precision("bf16")
    m = module_from_golden(EdgewiseMSA, params, **_ctor(meta)).to(dtype)
    x = torch.from_numpy(d["x"]).cuda().to(dtype)
    ops.enable_timing(True)
    with torch.no_grad():
        y = m(x)
    torch.cuda.synchronize()
    ops.enable_timing(False)
    from mop_amd import _lib
    assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_FUSED, "fused kernel was not selected"
    err = max_abs(y.float().cpu().numpy(), d["y"])
    print(f"{name} {dtype}: fused fwd max-abs err {err:.3e}")
    assert err <= TOL_BF16, f"fused fwd y max-abs {err:.3e}"


# ----------------------------------------------------------------------------------------------
# fused gfx950 kernels: shape sweep vs the oracle, and size-independent properties at the full
# BASELINE.json size (B=256, N=197, D=384, H=6, V=5, r=4)
# ----------------------------------------------------------------------------------------------
def _mk(D, H, V, r, seed, init="mix5"):
    from mop_amd.nn import EdgewiseMSA
    torch.manual_seed(seed)
    m = EdgewiseMSA(D, H, n_views=V, share_qkv=True, gate_mode="lowrank", gate_rank=r, gate_init=init)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("_scale"):
                p.add_(0.1 * torch.randn_like(p))
            elif "edge_head" in n_ and n_.endswith("weight"):
                p.mul_(3.0)
        m.chain_value_logit.fill_(-0.5)
    return m


@pytest.mark.parametrize("shape", [
    # B, N, D, H, V, r        (dk = D/H in {16, 32, 64}; N hits every NT bucket and its edges)
    (2, 1, 32, 2, 2, 1), (3, 31, 64, 4, 3, 2), (2, 32, 64, 2, 5, 4), (2, 33, 128, 2, 2, 3), (1, 64, 64, 1, 8, 4),
    (2, 65, 96, 3, 4, 2), (2, 65, 384, 6, 5, 4), (1, 96, 64, 1, 5, 4), (2, 97, 64, 2, 3, 2), (1, 128, 64, 4, 5, 4), (1, 129, 128, 2, 3, 1), (2, 145, 128, 2, 5, 4), (1, 160, 64, 1, 4, 2), (1, 161, 32, 2, 3, 1), (2, 170, 64, 1, 5, 4), (1, 192, 64, 2, 2, 1), (2, 196, 128, 2, 5, 4), (1, 224, 64, 1, 5, 4), (2, 100, 64, 1, 8, 4)])
def test_fused_vs_oracle_shape_sweep(shape):
    """fused fwd+bwd (bf16 MFMA) vs the float64 oracle on fresh seeded inputs."""
    from oracle import edgewise as oe
    import mop_amd
    from mop_amd import ops, _lib
    B, N, D, H, V, r = shape
    mop_amd.set_precision("bf16")
    m = _mk(D, H, V, r, seed=B * 1000 + N)
    params = {k: v.detach().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(N)
    x = torch.randn(B, N, D, generator=g).numpy()
    w = torch.randn(B, N, D, generator=g).numpy()
    out, cache = oe.module_fwd(x.astype(np.float64), params, H, V, True, 0.5)
    dx_ref, g_ref = oe.module_bwd(w.astype(np.float64), cache)
    y, dx, grads = run_fwd_bwd(m.cuda().eval(), x, w)
    assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_FUSED and ops.LAST_PATH["edgewise_bwd"] == _lib.PATH_FUSED
    assert max_abs(y, out) <= TOL_BF16
    assert rel_err(dx, dx_ref) <= 3e-2
    # per tensor: 3e-2, or 4 x the amount the exact (float64) gradient itself moves when the inputs are rounded to bf16 -- e.g. at
    # (1,129,128,2,3,1) the rank-1 row_proj gradient is 1e-4 of the module's scale and pure cancellation noise in any bf16 arithmetic
    noise = oracle_bf16_noise(oe.module_fwd, oe.module_bwd, x, w, params, H, V, True, 0.5)
    check_grads(grads, g_ref, GTOL_BF16, d=noise)


def test_shapes_outside_the_fused_kernels_take_the_generic_path():
    import mop_amd
    from mop_amd import ops, _lib
    mop_amd.set_precision("bf16")
    m = _mk(64, 2, 3, 2, seed=5).cuda().eval()
    x = torch.randn(1, 225, 64, device="cuda", requires_grad=True)      # N > 224
    m(x).sum().backward()
    assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_GENERIC and ops.LAST_PATH["edgewise_bwd"] == _lib.PATH_GENERIC
    m3 = _mk(64, 1, 8, 4, seed=7).cuda().eval()                           # V = 8 at N = 224: LDS budget of the fused backward exceeded
    x3 = torch.randn(1, 224, 64, device="cuda", requires_grad=True)
    m3(x3).sum().backward()
    assert ops.LAST_PATH["edgewise_fwd"] == ops.LAST_PATH["edgewise_bwd"] == _lib.PATH_GENERIC
    m2 = _mk(96, 2, 3, 2, seed=6).cuda().eval()                           # dk = 48
    m2(torch.randn(1, 16, 96, device="cuda"))
    assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_GENERIC


@pytest.fixture(scope="module")
def ns_full():
    """one fwd+bwd of the full BASELINE.json configs[1] size through the fused kernels (bf16)."""
    import mop_amd
    mop_amd.set_precision("auto")
    m = _mk(384, 6, 5, 4, seed=0).cuda().to(torch.bfloat16).eval()
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(256, 197, 384, device="cuda", dtype=torch.bfloat16, generator=g)
    dy = torch.randn(256, 197, 384, device="cuda", dtype=torch.bfloat16, generator=g)

    def run(xx, dd):
        m.zero_grad()
        xx = xx.clone().requires_grad_(True)
        y = m(xx)
        y.backward(dd)
        torch.cuda.synchronize()
        return y.detach(), xx.grad.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    return m, x, dy, run


def test_full_size_is_deterministic_and_batch_independent(ns_full):
    """bit-identical across runs, and a sample's result does not depend on the batch it rides in
    (every (b,h) is one workgroup with a fixed reduction order)."""
    from mop_amd import ops, _lib
    m, x, dy, run = ns_full
    y1, dx1, g1 = run(x, dy)
    assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_FUSED and ops.LAST_PATH["edgewise_bwd"] == _lib.PATH_FUSED
    y2, dx2, g2 = run(x, dy)
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    assert torch.isfinite(y1.float()).all() and torch.isfinite(dx1.float()).all()
    # batch independence, checked on the attention core itself (the surrounding hipBLASLt GEMMs may pick a
    # different tiling, hence summation order, for a different batch size)
    H, dk, V = 6, 64, 5
    g = torch.Generator(device="cuda").manual_seed(11)
    qkv = torch.randn(256, 197, 1, 3, H, dk, device="cuda", dtype=torch.bfloat16, generator=g)
    gy = torch.randn(256, 197, H * dk, device="cuda", dtype=torch.bfloat16, generator=g)
    eh = m.edge_head

    def core(q):
        q = q.clone().requires_grad_(True)
        sqk = (m.q_scale * m.k_scale).squeeze(2).float() / 8.0
        y = ops.edgewise_lowrank_core(q, sqk, m.v_scale[0, :, 0].float(), m.v_scale[V - 1, :, 0].float(),
                                      eh.row_proj.weight.squeeze(-1).float(), eh.row_proj.bias.float(),
                                      eh.col_proj.weight.squeeze(-1).float(), eh.col_proj.bias.float(),
                                      m.chain_value_logit.float(), 0.5, V)
        y.backward(gy[: q.shape[0]] if q.shape[0] == 256 else gy[37:40])
        return y.detach(), q.grad.detach()
    yb, gb = core(qkv)
    ys, gs = core(qkv[37:40])
    assert torch.equal(ys, yb[37:40]) and torch.equal(gs, gb[37:40])


def test_full_size_token_permutation_equivariance(ns_full):
    """no positional term anywhere on the path: permuting the tokens permutes y and dx."""
    m, x, dy, run = ns_full
    perm = torch.randperm(197, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    y, dx, _ = run(x[:4], dy[:4])
    yp, dxp, _ = run(x[:4][:, perm], dy[:4][:, perm])
    assert max_abs(yp.float().cpu().numpy(), y[:, perm].float().cpu().numpy()) <= 2e-2
    assert rel_err(dxp.float().cpu().numpy(), dx[:, perm].float().cpu().numpy()) <= 5e-2


def test_full_size_gradient_linearity(ns_full):
    """backward is linear in dy: grad(2 dy) == 2 grad(dy) (exactly, powers of two commute with rounding)."""
    m, x, dy, run = ns_full
    _, dx1, g1 = run(x[:8], dy[:8])
    _, dx2, g2 = run(x[:8], 2 * dy[:8])
    assert torch.equal(dx2, 2 * dx1)
    for k in g1:
        assert torch.equal(g2[k], 2 * g1[k]), k


def test_constant_values_give_constant_rows():
    """every attention map on the path is row-stochastic: with v_j = c for all tokens,
    y_i = c*vs0 + sigmoid(w) * c*vsL for every query (reference :554-562)."""
    import mop_amd
    mop_amd.set_precision("bf16")
    D, H, V = 128, 2, 5
    m = _mk(D, H, V, 4, seed=11).cuda().eval()
    with torch.no_grad():
        W = m.qkv.weight                       # zero the v projection, then bias it through x's last channel
        W[2 * D:].zero_()
        W[2 * D:, -1] = torch.linspace(-1, 1, D)
        m.proj.weight.copy_(torch.eye(D))
        x = torch.randn(2, 197, D, device="cuda")
        x[..., -1] = 1.0
        y = m(x)
        c = torch.linspace(-1, 1, D, device="cuda").view(H, D // H)
        exp = c * m.v_scale[0, :, 0] + torch.sigmoid(m.chain_value_logit) * c * m.v_scale[V - 1, :, 0]
    assert max_abs(y.cpu().numpy(), exp.reshape(1, 1, D).expand_as(y).cpu().numpy()) <= 1e-2


def test_vit_edgewise_trains_on_the_fused_path():
    """ViTEdgewise (the reference experiments' caller of the hot path) + AdamW/warm-up-cosine: the loss falls on a fixed batch."""
    import torch.nn.functional as F
    from mop_amd import ops, _lib
    from mop_amd.nn import ViTEdgewise
    from mop_amd.training import DataParallelStep, make_optimizer_and_schedule
    torch.manual_seed(0)
    m = ViTEdgewise(dim=128, depth=2, heads=2, n_classes=10, n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2,
                    gate_init="mix5", drop_path=0.0).cuda().to(torch.bfloat16)
    x = torch.randn(32, 3, 32, 32, device="cuda", dtype=torch.bfloat16)
    y = torch.randint(0, 10, (32,), device="cuda")
    opt, sched = make_optimizer_and_schedule(m, lr=2e-3, weight_decay=0.0, steps=30, warmup_frac=0.1)
    step = DataParallelStep(m, opt, lambda out, tgt: F.cross_entropy(out.float(), tgt), sched)
    losses = [float(step(x, y)) for _ in range(30)]
    assert ops.LAST_PATH["edgewise_bwd"] == _lib.PATH_FUSED
    assert all(l == l for l in losses) and losses[-1] < 0.6 * losses[0], losses[::5]


def test_token_linear_gradients_match_nn_linear_on_gpu():
    """TokenLinear's batched weight-gradient GEMM (bf16 slices, fp32 sum) against nn.Linear's single GEMM at the bench shape."""
    import torch.nn as nn
    from mop_amd.nn.linear import TokenLinear
    torch.manual_seed(0)
    ref = nn.Linear(384, 1152, bias=True).cuda().to(torch.bfloat16)
    lin = TokenLinear(384, 1152, bias=True).cuda().to(torch.bfloat16)
    lin.load_state_dict(ref.state_dict())
    x = torch.randn(64, 197, 384, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(64, 197, 1152, device="cuda", dtype=torch.bfloat16)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = ref(xa), lin(xb)
    assert torch.equal(ya, yb)
    (ya * w).sum().backward(); (yb * w).sum().backward()
    gw64 = (w.double().flatten(0, 1).t() @ x.double().flatten(0, 1))          # fp64 reference of dW
    err_ref = (ref.weight.grad.double() - gw64).abs().max() / gw64.abs().max()
    err_new = (lin.weight.grad.double() - gw64).abs().max() / gw64.abs().max()
    assert float(err_new) <= max(2.0 * float(err_ref), 5e-3)                  # at least as accurate as the single GEMM
    assert torch.allclose(xa.grad.float(), xb.grad.float(), rtol=0, atol=0)  # dX is the same hipBLASLt GEMM
    assert float((lin.bias.grad.double() - w.double().sum((0, 1))).abs().max() / w.double().sum((0, 1)).abs().max()) <= 1e-2


def test_reduce_parts_matches_a_host_side_sum():
    """mopk_edgewise_reduce_parts (one launch, fixed order) against torch sums of the same partial buffers; bitwise repeatable."""
    import ctypes as C
    from mop_amd import _lib as L
    lib = L.lib()
    B, V, H, dk = 37, 5, 6, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    parts = [torch.randn(B, V, H, dk, device="cuda", generator=g), torch.randn(B, H, dk, device="cuda", generator=g),
             torch.randn(B, H, dk, device="cuda", generator=g), torch.randn(B, H, device="cuda", generator=g)]
    a = L.EdgewiseArgs()
    a.B, a.H, a.V, a.dk = B, H, V, dk
    a.dsqk_part, a.dvs0_part, a.dvsL_part, a.dlogit_part = (p.data_ptr() for p in parts)
    outs = []
    for _ in range(2):
        o = [torch.empty(V, H, dk, device="cuda"), torch.empty(H, dk, device="cuda"), torch.empty(H, dk, device="cuda"), torch.empty(1, device="cuda")]
        rc = lib.mopk_edgewise_reduce_parts(C.byref(a), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        outs.append(o)
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    exp = [parts[0].double().sum(0), parts[1].double().sum(0), parts[2].double().sum(0), parts[3].double().sum().reshape(1)]
    for x, e in zip(outs[0], exp):
        assert float((x.double() - e).abs().max()) <= 1e-4 * max(1.0, float(e.abs().max()))
    assert lib.mopk_edgewise_reduce_parts(None, 0, 0, 0, 0, None) == -2 or lib.mopk_edgewise_reduce_parts(None, 0, 0, 0, 0, None) < 0


@pytest.mark.parametrize("variant", ["dense_k3", "lowrank_lens", "dense_k3_lens"])
def test_variants_vs_oracle_at_full_sequence_length(variant):
    """N = 197 (the config's token count), one head of dk = 64: dense gate head with the 3x3 mid convolution and the dilated S lens
    bank (attention_variants.py:250-272, :312-318, :425-442, :523-533) on the generic path in exact fp32 arithmetic vs the float64 oracle.
    (The reference fixtures for these variants stop at N = 33.)"""
    from oracle import edgewise as oe
    import mop_amd
    from mop_amd.nn import EdgewiseMSA
    N, D, H, V = 197, 64, 1, 3
    kw = dict(n_views=V, share_qkv=True)
    dil = None
    if variant == "dense_k3":
        kw.update(gate_mode="dense", use_k3=True, gate_init="and")
    elif variant == "lowrank_lens":
        dil = (1, 2)
        kw.update(gate_mode="lowrank", gate_rank=2, gate_init="mix5", use_lens_bank=True, lens_dilations=dil)
    else:
        dil = (2,)
        kw.update(gate_mode="dense", use_k3=True, gate_init="or", use_lens_bank=True, lens_dilations=dil)
    mop_amd.set_precision("fp32")
    torch.manual_seed(197)
    m = EdgewiseMSA(D, H, **kw)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("_scale"):
                p.add_(0.1 * torch.randn_like(p))
            elif n_.endswith("conv2.bias"):
                p.copy_(0.5 * torch.randn_like(p))          # the -5 preset leaves every gate (and its gradient) ~0
        m.chain_value_logit.fill_(-0.5)
    params = {k: v.detach().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    x = torch.randn(1, N, D).numpy()
    w = torch.randn(1, N, D).numpy()
    out, cache = oe.module_fwd(x.astype(np.float64), params, H, V, True, 0.5, lens_dilations=dil)
    dx_ref, g_ref = oe.module_bwd(w.astype(np.float64), cache)
    y, dx, grads = run_fwd_bwd(m.cuda().eval(), x, w)
    assert max_abs(y, out) <= 1e-4, f"y {max_abs(y, out):.3e}"
    assert rel_err(dx, dx_ref) <= 1e-3, f"dx {rel_err(dx, dx_ref):.3e}"
    check_grads(grads, g_ref, 1e-3, floor=1e-3)


@pytest.mark.parametrize("shape", [(2, 197, 384, 6, 5, 2, (1, 2)), (3, 50, 128, 2, 3, 4, (1, 2, 3)), (1, 8, 64, 4, 2, 1, (1, 9)),
                                   (2, 129, 64, 1, 4, 3, (2,)), (1, 224, 128, 4, 8, 2, (3,)), (2, 65, 64, 4, 2, 4, (1, 2, 4, 8))])
def test_fused_lowrank_lens_bank_vs_oracle_and_generic(shape):
    """Low-rank gate head + S lens bank (attention_variants.py:425-442, :523-533, :323-326) on the FUSED kernels: the lens planes reach the
    head only as row / column means, which ops.lens_mean_features evaluates in closed form and the kernels take as extra feature channels
    (MopkEdgewiseExt.n_extra).  Output and every gradient (lens weights included) against the float64 oracle on bf16-rounded inputs, next to
    the generic path (which convolves the planes) on the same inputs."""
    from oracle import edgewise as oe
    import mop_amd
    from mop_amd import ops, _lib
    from mop_amd.nn import EdgewiseMSA
    B, N, D, H, V, r, dil = shape
    torch.manual_seed(N * 11 + V)
    m = EdgewiseMSA(D, H, n_views=V, share_qkv=True, gate_mode="lowrank", gate_rank=r, gate_init="mix5", use_lens_bank=True, lens_dilations=dil)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("_scale"):
                p.add_(0.1 * torch.randn_like(p))
            if "proj.weight" in n_ and "edge_head" in n_:
                p.mul_(3.0)
        m.chain_value_logit.fill_(-0.5)
    x, w = torch.randn(B, N, D), torch.randn(B, N, D)
    rb = lambda t: torch.as_tensor(t).to(torch.bfloat16).double().numpy()
    params = {k: rb(v) for k, v in m.state_dict().items()}
    out, cache = oe.module_fwd(rb(x), params, H, V, True, 0.5, lens_dilations=dil)
    dx_ref, g_ref = oe.module_bwd(rb(w), cache)
    mop_amd.set_precision("bf16")
    res = {}
    for path in ("generic", "auto"):
        ops.set_path(path)
        mg = m.cuda().to(torch.bfloat16).eval()
        mg.zero_grad()
        res[path] = run_fwd_bwd(mg, rb(x), rb(w), dtype=torch.bfloat16)
        want = _lib.PATH_FUSED if path == "auto" else _lib.PATH_GENERIC
        assert ops.LAST_PATH["edgewise_fwd"] == want and ops.LAST_PATH["edgewise_bwd"] == want
    y, dx, grads = res["auto"]
    yg, dxg, gg = res["generic"]
    assert max_abs(y, out) <= TOL_BF16, f"y {max_abs(y, out):.3e}"
    assert rel_err(dx, dx_ref) <= GTOL_BF16, f"dx {rel_err(dx, dx_ref):.3e}"
    assert set(grads) == set(g_ref)
    for k in g_ref:          # per tensor: the bf16 bound, or twice what the generic path (same inputs, planes convolved in fp32) shows
        e, eg = rel_err(grads[k].reshape(g_ref[k].shape), g_ref[k]), rel_err(gg[k].reshape(g_ref[k].shape), g_ref[k])
        assert e <= max(GTOL_BF16, 2.0 * eg), f"{k}: fused {e:.3e} generic {eg:.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 197, 6, 64, 5, (1, 2)), (3, 50, 2, 32, 3, (1, 2, 3)), (1, 8, 4, 16, 2, (1, 9)), (1, 224, 1, 64, 4, (2, 7, 1, 10)), (1, 100, 1, 32, 2, (150, 60)),
                                   (2, 1, 2, 64, 2, (1,))])
def test_lens_means_kernels_vs_torch_statement(shape, dtype):
    """mopk_lens_means_fwd / _bwd (mop_amd/csrc/lens_means.hip) against the torch statement of the same closed form (ops._lens_means_fwd /
    _lens_means_bwd, itself pinned against F.conv2d planes and autograd on the CPU): means, q / k gradients (added into an existing
    buffer), scale and lens-weight gradients."""
    from mop_amd import ops
    B, N, H, dk, V, dil = shape
    torch.manual_seed(N + V)
    qkv = torch.randn(B, N, 1, 3, H, dk, device="cuda").to(dtype)
    sqk = (0.125 + 0.05 * torch.randn(V, H, dk, device="cuda")).contiguous()
    lw = torch.randn(len(dil), V, 3, 3, device="cuda").contiguous()
    qb = qkv.to(torch.bfloat16).float()                      # the kernel works on bf16-rounded q / k, like the fused Edgewise kernels
    r0, c0, st = ops._lens_means_fwd(qb, sqk, lw, dil)
    r1, c1 = ops.lens_means_hip(qkv, sqk, lw, dil)
    scale = float(r0.abs().max()) + 1e-6
    assert float((r1 - r0).abs().max()) <= 2e-5 * scale and float((c1 - c0).abs().max()) <= 2e-5 * (float(c0.abs().max()) + 1e-6)
    gr, gc = torch.randn_like(r0), torch.randn_like(c0)
    dq0, dk0, dsqk0, dlw0 = ops._lens_means_bwd(gr, gc, qb, sqk, lw, dil, st)
    base = torch.randn_like(qkv)
    dqkv = base.clone()
    dsqk1, dlw1 = ops.lens_means_bwd_hip(gr, gc, qkv, dqkv, sqk, lw, dil)
    exp = base.float().clone()
    exp[:, :, 0, 0] += dq0.permute(0, 2, 1, 3)
    exp[:, :, 0, 1] += dk0.permute(0, 2, 1, 3)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-5            # bf16 buffer: one rounding of the sum
    assert float((dqkv.float() - exp).abs().max()) <= tol * (float(exp.abs().max()) + 1e-6)
    assert torch.equal(dqkv[:, :, 0, 2], base[:, :, 0, 2])     # v is not touched
    assert float((dsqk1 - dsqk0).abs().max()) <= 1e-4 * (float(dsqk0.abs().max()) + 1e-6)
    assert float((dlw1 - dlw0).abs().max()) <= 1e-4 * (float(dlw0.abs().max()) + 1e-6)


def test_fused_lowrank_lens_bank_takes_dropout():
    """attn_drop in training mode with the S lens bank: carried by the fused kernels (the generic path would refuse)"""
    import mop_amd
    from mop_amd import ops, _lib
    from mop_amd.nn import EdgewiseMSA
    torch.manual_seed(5)
    m = EdgewiseMSA(128, 2, n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2, use_lens_bank=True, attn_drop=0.2).cuda().to(torch.bfloat16).train()
    x = torch.randn(2, 40, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    m(x).float().sum().backward()
    assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_FUSED and torch.isfinite(x.grad).all()
    assert all(torch.isfinite(c.weight.grad).all() and float(c.weight.grad.abs().max()) > 0 for c in m.lens_bank)


@pytest.mark.parametrize("shape", [(2, 197, 384, 6, 5), (3, 50, 128, 2, 3), (1, 8, 64, 4, 2), (2, 129, 64, 1, 4), (1, 224, 128, 4, 8), (2, 65, 128, 2, 5)])
def test_fused_dense_head_forward_vs_oracle_and_generic(shape):
    """dense gate head without the 3x3 convolution (attention_variants.py:250-272 minus :253-254, :312-318) evaluated inside the fused
    forward's mix loop (bf16 score tiles, fp32 MLP): vs the float64 oracle on bf16-rounded inputs, and vs the generic path."""
    from oracle import edgewise as oe
    import mop_amd
    from mop_amd import ops, _lib
    from mop_amd.nn import EdgewiseMSA
    B, N, D, H, V = shape
    torch.manual_seed(N * 7 + V)
    m = EdgewiseMSA(D, H, n_views=V, share_qkv=True, gate_mode="dense", use_k3=False, gate_init="and")
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("_scale"):
                p.add_(0.1 * torch.randn_like(p))
            elif n_.endswith("conv2.bias"):
                p.copy_(0.7 * torch.randn_like(p))          # the -5 preset leaves every gate ~0
            elif "conv1" in n_ or "conv2" in n_:
                p.mul_(1.5)
        m.chain_value_logit.fill_(-0.5)
    x = torch.randn(B, N, D)
    rb = lambda t: torch.as_tensor(t).to(torch.bfloat16).double().numpy()
    params = {k: rb(v) for k, v in m.state_dict().items()}
    out, _ = oe.module_fwd(rb(x), params, H, V, True, 0.5)
    mop_amd.set_precision("bf16")
    try:
        mg = m.cuda().to(torch.bfloat16).eval()
        xg = x.cuda().to(torch.bfloat16)
        with torch.no_grad():
            y = mg(xg)
        assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_FUSED
        ops.set_path("generic")
        with torch.no_grad():
            yg = mg(xg)
        assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_GENERIC
        ops.set_path("auto")
        xr = xg.clone().requires_grad_(True)
        mg(xr).float().sum().backward()                     # a differentiated call: fused too while dW1[k][:] + db1[k] fit one 16-slot row (V <= 6)
        want = _lib.PATH_FUSED if V <= 6 else _lib.PATH_GENERIC
        assert ops.LAST_PATH["edgewise_fwd"] == want and ops.LAST_PATH["edgewise_bwd"] == want and torch.isfinite(xr.grad).all()
    finally:
        ops.set_path("auto")
        mop_amd.set_precision("auto")
    scale = max(1.0, float(np.abs(out).max()))
    assert max_abs(y.float().cpu().numpy(), out) <= 1e-2 * scale, f"fused vs oracle {max_abs(y.float().cpu().numpy(), out):.3e}"
    assert max_abs(y.float().cpu().numpy(), yg.float().cpu().numpy()) <= 1.5e-2 * scale


@pytest.mark.parametrize("shape", [(2, 197, 384, 6, 5), (3, 50, 128, 2, 3), (1, 8, 64, 4, 2), (2, 129, 64, 1, 4), (2, 33, 64, 2, 6), (2, 65, 128, 2, 5), (1, 96, 64, 1, 3), (1, 145, 64, 1, 5), (1, 170, 128, 2, 4)])
def test_fused_dense_head_backward_vs_oracle(shape):
    """training through the fused dense head (launch A: per-edge MLP backward, feature gradients into the hand-off slabs, weight
    gradients by wave butterflies; launch B: log C<- gradient in the <- chain's seed; launch C: S_v^T feature gradients added
    transposed) vs the float64 oracle's hand-derived backward."""
    from oracle import edgewise as oe
    import mop_amd
    from mop_amd import ops, _lib
    from mop_amd.nn import EdgewiseMSA
    B, N, D, H, V = shape
    torch.manual_seed(N * 11 + V)
    m = EdgewiseMSA(D, H, n_views=V, share_qkv=True, gate_mode="dense", use_k3=False, gate_init="and")
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("_scale"):
                p.add_(0.1 * torch.randn_like(p))
            elif n_.endswith("conv2.bias"):
                p.copy_(0.7 * torch.randn_like(p))
            elif "conv1" in n_ or "conv2" in n_:
                p.mul_(1.5)
        m.chain_value_logit.fill_(-0.5)
    params = {k: v.detach().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(N)
    x = torch.randn(B, N, D, generator=g).numpy()
    w = torch.randn(B, N, D, generator=g).numpy()
    out, cache = oe.module_fwd(x.astype(np.float64), params, H, V, True, 0.5)
    dx_ref, g_ref = oe.module_bwd(w.astype(np.float64), cache)
    mop_amd.set_precision("bf16")
    try:
        y, dx, grads = run_fwd_bwd(m.cuda().eval(), x, w)
        assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_FUSED and ops.LAST_PATH["edgewise_bwd"] == _lib.PATH_FUSED
    finally:
        mop_amd.set_precision("auto")
    assert max_abs(y, out) <= 1e-2 * max(1.0, float(np.abs(out).max())), f"y {max_abs(y, out):.3e}"
    assert rel_err(dx, dx_ref) <= 3e-2, f"dx {rel_err(dx, dx_ref):.3e}"
    noise = oracle_bf16_noise(oe.module_fwd, oe.module_bwd, x, w, params, H, V, True, 0.5)
    check_grads(grads, g_ref, GTOL_BF16, d=noise)


def test_fused_dense_head_forward_vs_reference_fixture():
    import mop_amd
    from mop_amd import ops, _lib
    from mop_amd.nn import EdgewiseMSA
    d, params, _, meta = load_golden("ewx_tiny_dense_v2")
    m = module_from_golden(EdgewiseMSA, params, dim=int(meta["dim"]), heads=int(meta["heads"]), n_views=int(meta["n_views"]),
                           share_qkv=bool(meta["share_qkv"]), gate_mode="dense", use_k3=False)
    mop_amd.set_precision("bf16")
    try:
        with torch.no_grad():
            y = m(torch.from_numpy(d["x"]).cuda())
        assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_FUSED
    finally:
        mop_amd.set_precision("auto")
    assert max_abs(y.cpu().numpy(), d["y"]) <= TOL_BF16


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("variant", ["lowrank", "dense", "dense_k3"])
@pytest.mark.parametrize("mask_kind", ["causal", "random"])
def test_masked_edgewise_extension_vs_oracle(variant, mask_kind, prec):
    """attn_mask on EdgewiseMSA: the reference is NaN there (SURVEY.md 8a note); the documented extension (mask on the probabilities
    only, gate features from the unmasked scores; generic path) against the float64 oracle, whose masked backward is pinned by finite
    differences in tests/test_oracle_golden.py."""
    from oracle import edgewise as oe
    import mop_amd
    from mop_amd import ops, _lib
    from mop_amd.nn import EdgewiseMSA
    B, N, D, H, V = 2, 37, 64, 2, 3
    kw = dict(gate_mode="lowrank", gate_rank=2, gate_init="mix5") if variant == "lowrank" else \
        dict(gate_mode="dense", use_k3=variant == "dense_k3", gate_init="and")
    torch.manual_seed(41)
    m = EdgewiseMSA(D, H, n_views=V, share_qkv=True, **kw)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("_scale"):
                p.add_(0.1 * torch.randn_like(p))
            elif n_.endswith("conv2.bias"):
                p.copy_(0.5 * torch.randn_like(p))
        m.chain_value_logit.fill_(-0.5)
    if mask_kind == "causal":
        mask = torch.ones(N, N).tril_()
    else:
        mask = (torch.rand(B, 1, N, N) > 0.4).float()
        mask[..., torch.arange(N), torch.arange(N)] = 1.0            # every query keeps at least one key
    params = {k: v.detach().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    x = torch.randn(B, N, D).numpy()
    w = torch.randn(B, N, D).numpy()
    out, cache = oe.module_fwd(x.astype(np.float64), params, H, V, True, 0.5, attn_mask=mask.numpy())
    dx_ref, g_ref = oe.module_bwd(w.astype(np.float64), cache)
    mop_amd.set_precision(prec)
    try:
        y, dx, grads = run_fwd_bwd(m.cuda().eval(), x, w, attn_mask=mask.cuda())
        assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_GENERIC and ops.LAST_PATH["edgewise_bwd"] == _lib.PATH_GENERIC
    finally:
        mop_amd.set_precision("auto")
    assert np.isfinite(y).all() and np.isfinite(dx).all()
    tol, gtol = (1e-4, 1e-3) if prec == "fp32" else (TOL_BF16, GTOL_BF16)
    assert max_abs(y, out) <= tol * max(1.0, float(np.abs(out).max())), f"y {max_abs(y, out):.3e}"
    assert rel_err(dx, dx_ref) <= gtol, f"dx {rel_err(dx, dx_ref):.3e}"
    noise = oracle_bf16_noise(lambda xx, pp, *a_: oe.module_fwd(xx, pp, *a_, attn_mask=mask.numpy()), oe.module_bwd, x, w, params,
                              H, V, True, 0.5) if prec == "bf16" else None
    check_grads(grads, g_ref, gtol, floor=1e-3, d=noise)


def test_fused_lowrank_batch_of_three_at_full_size_vs_oracle():
    """B = 3 images at the north-star layer shape (N = 197, D = 384, 6 heads, 5 views, r = 4), fused bf16 kernels vs the float64 oracle:
    the reference fixture at this shape is B = 1, so batch striding of every launch (saved records, hand-off regions, partials) is
    only covered here."""
    from oracle import edgewise as oe
    import mop_amd
    from mop_amd import ops, _lib
    B, N, D, H, V, r = 3, 197, 384, 6, 5, 4
    mop_amd.set_precision("bf16")
    m = _mk(D, H, V, r, seed=3197)
    params = {k: v.detach().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(N + B)
    x = torch.randn(B, N, D, generator=g).numpy()
    w = torch.randn(B, N, D, generator=g).numpy()
    out, cache = oe.module_fwd(x.astype(np.float64), params, H, V, True, 0.5)
    dx_ref, g_ref = oe.module_bwd(w.astype(np.float64), cache)
    y, dx, grads = run_fwd_bwd(m.cuda().eval(), x, w)
    assert ops.LAST_PATH["edgewise_fwd"] == _lib.PATH_FUSED and ops.LAST_PATH["edgewise_bwd"] == _lib.PATH_FUSED
    assert max_abs(y, out) <= TOL_BF16
    # every image separately: a stride bug would hit some images and not others
    for b in range(B):
        assert rel_err(dx[b], dx_ref[b]) <= 3e-2, f"dx[{b}] {rel_err(dx[b], dx_ref[b]):.3e}"
    noise = oracle_bf16_noise(oe.module_fwd, oe.module_bwd, x, w, params, H, V, True, 0.5, samples=2)
    check_grads(grads, g_ref, GTOL_BF16, d=noise)
