"""-m gpu: the LayerNorm prologue / residual epilogue (mopk_layernorm_*) against torch's fp32 LayerNorm on the
CPU (the op the reference block calls: experiments/cifar100_edgewise_gates.py:371-374), and BlockEdgewise with the fused edges
against the same block on the unfused route."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(7, 64), (1970, 384), (33, 1152), (5, 4096), (129, 72), (4, 8)]


def _ref(x, g, b, eps, dy, dres):
    """fp64 LayerNorm forward/backward on the CPU"""
    x = x.double().requires_grad_(True)
    g = g.double().requires_grad_(True)
    b = b.double().requires_grad_(True)
    y = F.layer_norm(x, (x.shape[-1],), g, b, eps)
    (y * dy.double()).sum().backward()
    return y.detach(), x.grad + dres.double(), g.grad, b.grad


@pytest.mark.parametrize("rows,dim", SHAPES)
@pytest.mark.parametrize("xdt,ydt,pdt", [(torch.float32, torch.float32, torch.float32),
                                         (torch.float32, torch.bfloat16, torch.float32),
                                         (torch.bfloat16, torch.bfloat16, torch.bfloat16)])
def test_layernorm_forward_backward_vs_torch(rows, dim, xdt, ydt, pdt):
    from mop_amd import ops
    gen = torch.Generator().manual_seed(rows * 131 + dim)
    x = (torch.randn(rows, dim, generator=gen) * 1.7 + 0.3).to(xdt)
    g = (1.0 + 0.2 * torch.randn(dim, generator=gen)).to(pdt)
    b = (0.1 * torch.randn(dim, generator=gen)).to(pdt)
    dy = torch.randn(rows, dim, generator=gen).to(ydt)
    dres = torch.randn(rows, dim, generator=gen).to(xdt)
    yr, dxr, dgr, dbr = _ref(x.float(), g.float(), b.float(), 1e-5, dy.float(), dres.float())

    xg = x.cuda().requires_grad_(True)
    gg, bg = g.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    xres, y = ops.layernorm_residual(xg, gg, bg, 1e-5, ydt)
    assert y.dtype == ydt and xres.data_ptr() == xg.data_ptr()
    torch.autograd.backward([xres, y], [dres.cuda(), dy.cuda()])
    torch.cuda.synchronize()
    ytol = 1e-5 if ydt == torch.float32 else 1.6e-2           # bf16 output: half an ulp of |y| <= 4
    assert (y.float().cpu().double() - yr).abs().max().item() <= ytol * max(1.0, yr.abs().max().item())
    # gradients: fp32 statistics in the kernel; the bf16 cases round dx / the parameter gradients once on the way out
    gt = 2e-5 if xdt == torch.float32 else 8e-3
    for name, got, ref in (("dx", xg.grad, dxr), ("dgamma", gg.grad, dgr), ("dbeta", bg.grad, dbr)):
        err = (got.float().cpu().double() - ref).abs().max().item() / max(1e-6, ref.abs().max().item())
        assert err <= gt, f"{name} {err:.2e}"


def test_layernorm_without_residual_branch_and_bias():
    from mop_amd import ops
    x = torch.randn(3, 50, 384, device="cuda", requires_grad=True)
    g = torch.rand(384, device="cuda", requires_grad=True)
    y = ops.layernorm(x, g, None, 1e-6)
    yr = F.layer_norm(x.detach().cpu(), (384,), g.detach().cpu(), None, 1e-6)
    assert (y.detach().cpu() - yr).abs().max().item() <= 1e-5
    y.square().sum().backward()
    xc = x.detach().cpu().requires_grad_(True)
    gc = g.detach().cpu().requires_grad_(True)
    F.layer_norm(xc, (384,), gc, None, 1e-6).square().sum().backward()
    assert (x.grad.cpu() - xc.grad).abs().max().item() <= 1e-4 * xc.grad.abs().max().item()
    assert (g.grad.cpu() - gc.grad).abs().max().item() <= 1e-4 * gc.grad.abs().max().item()


def test_layernorm_rejects_what_the_kernel_does_not_cover():
    from mop_amd import _lib as L
    x = torch.randn(4, 20, device="cuda")
    y = torch.empty_like(x)
    g = torch.ones(20, device="cuda")
    a = L.LayerNormArgs(rows=4, dim=20, x_dtype=0, y_dtype=0, p_dtype=0, eps=1e-5, x_ld=20, y_ld=20,
                        x=x.data_ptr(), gamma=g.data_ptr(), y=y.data_ptr())
    assert L.lib().mopk_layernorm_fwd(C.byref(a), None) == -3            # MOPK_ERR_UNSUPPORTED: dim % 8 != 0
    a.dim, a.rows = 16, 0
    assert L.lib().mopk_layernorm_fwd(C.byref(a), None) == -1            # MOPK_ERR_BAD_SHAPE
    assert L.lib().mopk_layernorm_fwd(None, None) == -2                  # MOPK_ERR_BAD_ARG
    from mop_amd import ops
    assert not ops.layernorm_supported(x, g) and ops.layernorm_supported(torch.empty(2, 16, device="cuda"), g)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 4e-2)])
def test_block_edgewise_fused_edges_match_the_unfused_block(dtype, tol):
    """same parameters, same input: LN prologue + residual GEMM epilogue vs torch LayerNorm + separate adds"""
    from mop_amd.nn.vit_edgewise import BlockEdgewise
    torch.manual_seed(3)
    blk = BlockEdgewise(128, 4, n_views=3, share_qkv=True, gate_mode="lowrank", gate_rank=2, gate_init="mix5").cuda().to(dtype)
    with torch.no_grad():
        for p in blk.parameters():
            p.add_(0.05 * torch.randn_like(p))
    x = torch.randn(3, 50, 128, device="cuda", dtype=dtype)
    w = torch.randn_like(x)
    outs = []
    for fused in (True, False):
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        if not fused:
            blk._fused_edges = lambda _x: False
        y = blk(xi)
        y.backward(w)
        outs.append((y.detach().float(), xi.grad.float(), {k: p.grad.float().clone() for k, p in blk.named_parameters()}))
    del blk._fused_edges
    (yf, dxf, gf), (yu, dxu, gu) = outs
    assert (yf - yu).abs().max().item() <= tol * yu.abs().max().item()
    assert (dxf - dxu).abs().max().item() <= tol * dxu.abs().max().item()
    scale = max(v.abs().max().item() for v in gu.values())
    for k in gu:
        assert (gf[k] - gu[k]).abs().max().item() <= tol * max(gu[k].abs().max().item(), 1e-2 * scale), k
