"""MI355X parity of the EdgewiseMSA kernels (through the nn.Module -> ctypes -> C ABI path)
against golden vectors produced by the reference itself, and against the numpy oracle on
fresh seeded inputs.  Tolerances are BASELINE.json's: <=1e-3 (fp32 arithmetic), <=1e-2 (bf16)."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from gpu_util import max_abs, module_from_golden, rel_err, run_fwd_bwd

pytestmark = pytest.mark.gpu

TOL_FP32, TOL_BF16 = 1e-3, 1e-2          # max-abs on y (north_star)
GTOL_FP32, GTOL_BF16 = 1e-3, 1.5e-1        # gradients: max-abs / max|ref| (bf16 bound is ours; north_star bounds y only)


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
    for k in gref:
        assert rel_err(grads[k].reshape(gref[k].shape), gref[k]) <= gtol, \
            f"grad {k}: rel {rel_err(grads[k].reshape(gref[k].shape), gref[k]):.3e}"


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
    assert rel_err(dx, d["dx"]) <= 5e-2


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
