"""torch.autograd bridges from tensors to the C ABI (include/mopk.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the current
HIP stream and the autograd graph.  Every attention core runs in libmopk.so; a CPU
tensor or a missing library raises -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

_PRECISION = "auto"  # "auto": fp32 tensors -> exact fp32 kernels, bf16 tensors -> bf16 MFMA


def set_precision(p: str) -> None:
    """'auto' | 'fp32' | 'bf16' -- arithmetic of the contractions (MopkPrecision)."""
    global _PRECISION
    if p not in ("auto", "fp32", "bf16"):
        raise ValueError(p)
    _PRECISION = p


def get_precision() -> str:
    return _PRECISION


_PATH = L.PATH_AUTO


def set_path(p: str) -> None:
    global _PATH
    _PATH = {"auto": L.PATH_AUTO, "generic": L.PATH_GENERIC, "fused": L.PATH_FUSED}[p]


def _prec_for(dtype: torch.dtype) -> int:
    if _PRECISION == "fp32":
        return L.PREC_FP32
    if _PRECISION == "bf16":
        return L.PREC_BF16
    return L.PREC_BF16 if dtype == torch.bfloat16 else L.PREC_FP32


def _io_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.MOPK_F32
    if t.dtype == torch.bfloat16:
        return L.MOPK_BF16
    raise TypeError(f"mop_amd supports float32 / bfloat16 tensors, got {t.dtype}")


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: mop_amd runs on MI355X (ROCm) tensors only; got a {t.device} tensor. "
            "There is no CPU fallback -- move the module and inputs to 'cuda'.")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


# ---- optional kernel timing (bench.py): HIP events on the stream the kernels are launched on ----
LAST_PATH = {}  # entry point -> MopkPath actually requested on the last call (tests assert on it)
_TIMING = None  # dict name -> list[(start_event, stop_event)] when enabled


def enable_timing(on: bool = True) -> None:
    global _TIMING
    _TIMING = {} if on else None


def timing_results() -> dict:
    """name -> list of elapsed milliseconds (call after torch.cuda.synchronize())."""
    if _TIMING is None:
        return {}
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in _TIMING.items()}


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _TIMING is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()  # current stream == the stream passed to libmopk

    def __exit__(self, *exc):
        if _TIMING is not None:
            self.b.record()
            _TIMING.setdefault(self.name, []).append((self.a, self.b))
        return False


def _bytes(n: int, dev) -> torch.Tensor:
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)


# --------------------------------------------------------------------------------------
# EdgewiseMSA low-rank core
# --------------------------------------------------------------------------------------
def _ew_views(args: L.EdgewiseArgs, qkv: torch.Tensor, prefix: str):
    """qkv: (B,N,Vq,3,H,dk) contiguous. Fills q/k/v0/vL (or dq/dk_/dv0/dvL) views."""
    B, N, Vq, three, H, dk = qkv.shape
    es = 1  # strides below are in elements
    s_n = Vq * 3 * H * dk
    s_b = N * s_n
    s_v = 3 * H * dk if Vq > 1 else 0
    base = qkv.data_ptr()
    isz = qkv.element_size()
    qn, kn, v0n, vLn = (("q", "k", "v0", "vL") if prefix == "" else ("dq", "dk_", "dv0", "dvL"))
    setattr(args, qn, L.View5(base, s_v, s_b, dk * es, s_n))
    setattr(args, kn, L.View5(base + H * dk * isz, s_v, s_b, dk, s_n))
    setattr(args, v0n, L.View4(base + 2 * H * dk * isz, s_b, dk, s_n))
    last = (Vq - 1) * 3 * H * dk
    setattr(args, vLn, L.View4(base + (last + 2 * H * dk) * isz, s_b, dk, s_n))


class _EdgewiseLowrankFn(torch.autograd.Function):
    """y = EdgewiseMSA core(qkv, ...) ; reference attention_variants.py:500-562."""

    @staticmethod
    def forward(ctx, qkv, sqk, vs0, vsL, Wr, br, Wc, bc, logit, beta_not, V, prec, path, want_bwd):
        _require_gpu(qkv, "EdgewiseMSA")
        lib = L.lib()
        B, N, Vq, _, H, dk = qkv.shape
        qkv = qkv.contiguous()
        dev = qkv.device
        f = dict(sqk=_f32c(sqk), vs0=_f32c(vs0), vsL=_f32c(vsL), Wr=_f32c(Wr), br=_f32c(br),
                 Wc=_f32c(Wc), bc=_f32c(bc), logit=_f32c(logit).reshape(1))
        r = f["Wr"].shape[0] // 4
        a = L.EdgewiseArgs()
        a.B, a.H, a.N, a.dk, a.V, a.r = B, H, N, dk, V, r
        a.io_dtype, a.precision, a.path, a.beta_not = _io_dtype(qkv), prec, path, float(beta_not)
        _ew_views(a, qkv, "")
        a.sqk, a.vs0, a.vsL = f["sqk"].data_ptr(), f["vs0"].data_ptr(), f["vsL"].data_ptr()
        a.Wr, a.br, a.Wc, a.bc = (f["Wr"].data_ptr(), f["br"].data_ptr(), f["Wc"].data_ptr(),
                                  f["bc"].data_ptr())
        a.chain_logit = f["logit"].data_ptr()
        y = torch.empty(B, N, H, dk, dtype=qkv.dtype, device=dev)
        a.y = L.View4(y.data_ptr(), N * H * dk, dk, H * dk)
        if path == L.PATH_AUTO and not want_bwd and lib.mopk_edgewise_fused_supported(C.byref(a)):
            a.path = path = L.PATH_FUSED  # forward-only: fused kernel (the fused backward is not built yet)
        LAST_PATH["edgewise_fwd"] = path
        saved = _bytes(lib.mopk_edgewise_saved_bytes(C.byref(a)), dev)
        ws = _bytes(lib.mopk_edgewise_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("edgewise_fwd"):
            rc = lib.mopk_edgewise_lowrank_fwd(C.byref(a), _stream())
        L.check(rc, "mopk_edgewise_lowrank_fwd")
        ctx.save_for_backward(qkv, saved, *f.values())
        ctx.meta = (beta_not, V, prec, path, r)
        return y.view(B, N, H * dk)

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        qkv, saved, sqk, vs0, vsL, Wr, br, Wc, bc, logit = ctx.saved_tensors
        beta_not, V, prec, path, r = ctx.meta
        B, N, Vq, _, H, dk = qkv.shape
        dev = qkv.device
        dy = dy.contiguous()
        if dy.dtype != qkv.dtype:
            dy = dy.to(qkv.dtype)
        a = L.EdgewiseArgs()
        a.B, a.H, a.N, a.dk, a.V, a.r = B, H, N, dk, V, r
        a.io_dtype, a.precision, a.path, a.beta_not = _io_dtype(qkv), prec, path, float(beta_not)
        _ew_views(a, qkv, "")
        a.sqk, a.vs0, a.vsL = sqk.data_ptr(), vs0.data_ptr(), vsL.data_ptr()
        a.Wr, a.br, a.Wc, a.bc = Wr.data_ptr(), br.data_ptr(), Wc.data_ptr(), bc.data_ptr()
        a.chain_logit = logit.data_ptr()
        a.y = L.View4(dy.data_ptr(), N * H * dk, dk, H * dk)  # unused by bwd, must be non-null
        a.dy = L.View4(dy.data_ptr(), N * H * dk, dk, H * dk)
        # unshared: only v of view 0 and V-1 receive gradient -> zero-fill the rest
        dqkv = (torch.empty_like(qkv) if Vq == 1 else torch.zeros_like(qkv))
        _ew_views(a, dqkv, "d")
        f32 = dict(dtype=torch.float32, device=dev)
        dsqk = torch.empty(B, V, H, dk, **f32)
        dvs0 = torch.empty(B, H, dk, **f32)
        dvsL = torch.empty(B, H, dk, **f32)
        C_ = 2 * V + 2
        dWr = torch.empty(4 * r, C_, **f32)
        dWc = torch.empty(4 * r, C_, **f32)
        dbr = torch.empty(4 * r, **f32)
        dbc = torch.empty(4 * r, **f32)
        dlg = torch.empty(B, H, **f32)
        a.dsqk_part, a.dvs0_part, a.dvsL_part = dsqk.data_ptr(), dvs0.data_ptr(), dvsL.data_ptr()
        a.dWr, a.dbr, a.dWc, a.dbc = dWr.data_ptr(), dbr.data_ptr(), dWc.data_ptr(), dbc.data_ptr()
        a.dlogit_part = dlg.data_ptr()
        ws = _bytes(lib.mopk_edgewise_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("edgewise_bwd"):
            rc = lib.mopk_edgewise_lowrank_bwd(C.byref(a), _stream())
        L.check(rc, "mopk_edgewise_lowrank_bwd")
        return (dqkv, dsqk.sum(0), dvs0.sum(0), dvsL.sum(0), dWr, dbr, dWc, dbc,
                dlg.sum().reshape(()), None, None, None, None, None)


def edgewise_lowrank_core(qkv, sqk, vs0, vsL, Wr, br, Wc, bc, chain_logit, beta_not: float,
                          n_views: int, precision: Optional[int] = None, path: Optional[int] = None):
    """qkv: (B,N,Vq,3,H,dk) with Vq in {1 (share_qkv), n_views}; returns (B,N,H*dk)."""
    prec = _prec_for(qkv.dtype) if precision is None else precision
    want_bwd = torch.is_grad_enabled() and any(
        t.requires_grad for t in (qkv, sqk, vs0, vsL, Wr, br, Wc, bc, chain_logit))
    return _EdgewiseLowrankFn.apply(qkv, sqk, vs0, vsL, Wr, br, Wc, bc, chain_logit, beta_not,
                                    n_views, prec, _PATH if path is None else path, want_bwd)


def sdpa_core(qkv, attn_mask=None):
    raise NotImplementedError("mopk_sdpa_* kernels are not built yet")


def dualpath_core(qkv1, qkv2, chain_logit, g_and, g_or, g_not, g_chain, beta_not, hops, attn_mask=None):
    raise NotImplementedError("mopk_dualpath_* kernels are not built yet")
