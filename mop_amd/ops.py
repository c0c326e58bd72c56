"""torch.autograd bridges from tensors to the C ABI (include/mopk.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the current
HIP stream and the autograd graph.  Every attention core runs in libmopk.so; a CPU
tensor or a missing library raises -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn.functional as F

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


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> C.c_void_p:
    """the current stream's handle.  torch.cuda.current_stream() costs ~20 us of host time per call (device-index plumbing + a Stream
    object); the raw accessor behind it ~1 us -- these wrappers are called once or twice per layer and step"""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _f32_pack(ts):
    """fp32 contiguous copies of several small parameter tensors.  fp32 inputs are used as they are; otherwise ONE
    concatenation + ONE conversion kernel serve all of them (views of a single buffer) instead of one cast kernel each."""
    if all(t.dtype == torch.float32 for t in ts):
        return [t.detach().contiguous() for t in ts]
    if len({t.dtype for t in ts}) != 1:
        return [_f32c(t) for t in ts]
    flat = torch.cat([t.detach().reshape(-1) for t in ts]).to(torch.float32)
    out, o = [], 0
    for t in ts:
        out.append(flat[o:o + t.numel()].view(t.shape))
        o += t.numel()
    return out


# ---- optional kernel timing (bench.py): HIP events on the stream the kernels are launched on ----
LAST_PATH = {}  # entry point -> MopkPath actually requested on the last call (tests assert on it)
_KEEP_WS = bool(__import__("os").environ.get("MOPK_STAMPS"))  # stamp builds only: keep the last workspace tensors alive for read-back
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


_SAVE_CHAIN_STATE = True


def set_save_chain_state(on: bool) -> None:
    """Fused EdgewiseMSA training forward: export chain prefix products / softmax constants for the backward
    (faster backward, ~1 MB per (b,h) of extra activation memory at N=197, V=5) or let the backward recompute them."""
    global _SAVE_CHAIN_STATE
    _SAVE_CHAIN_STATE = bool(on)


class _EdgewiseLowrankFn(torch.autograd.Function):
    """y = EdgewiseMSA core(qkv, ...) ; reference attention_variants.py:500-562."""

    @staticmethod
    def forward(ctx, qkv, sqk, vs0, vsL, Wr, br, Wc, bc, logit, beta_not, V, prec, path, want_bwd, drop=(0.0, 0),
                lens_w=None, lens_dil=()):
        """lens_w (L,V,3,3) + lens_dil: the S lens bank.  Its planes reach the head as E = L V extra feature channels given by their
        row / column means (Wr / Wc then have 2V + 2 + E input channels): fused path only (MopkEdgewiseExt.n_extra)."""
        _require_gpu(qkv, "EdgewiseMSA")
        lib = L.lib()
        B, N, Vq, _, H, dk = qkv.shape
        qkv = qkv.contiguous()
        dev = qkv.device
        n_extra = 0 if lens_w is None else int(lens_w.shape[0] * lens_w.shape[1])
        f = dict(zip(("sqk", "vs0", "vsL", "Wr", "br", "Wc", "bc", "logit"),
                     _f32_pack([sqk, vs0, vsL, Wr, br, Wc, bc, logit.reshape(1)])))
        ctx.small_dtype = sqk.dtype if len({t.dtype for t in (sqk, vs0, vsL, Wr, br, Wc, bc, logit)}) == 1 else None
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
        extras = ()
        ctx.lens_dil, ctx.lens_dtype = tuple(lens_dil), (None if lens_w is None else lens_w.dtype)
        if n_extra:
            lens_w = _f32c(lens_w)
            row_x, col_x = lens_means_hip(qkv, f["sqk"], lens_w, lens_dil)
            extras = (row_x, col_x, lens_w)
            ext = L.EdgewiseExt()
            ext.n_extra, ext.row_extra, ext.col_extra = n_extra, extras[0].data_ptr(), extras[1].data_ptr()
            a.ext = C.pointer(ext)
            if path == L.PATH_GENERIC or not lib.mopk_edgewise_fused_supported(C.byref(a)):
                raise NotImplementedError("extra feature channels are an input of the fused Edgewise kernels only")
            path = L.PATH_FUSED
        if path == L.PATH_AUTO:   # AUTO: fused gfx950 kernels when they cover the shape, generic otherwise
            path = L.PATH_FUSED if lib.mopk_edgewise_fused_supported(C.byref(a)) else L.PATH_GENERIC
        a.path = path
        a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
        # training forward of the fused path also exports the chain state its backward would otherwise recompute
        a.save_for_backward = int(bool(want_bwd) and path == L.PATH_FUSED and _SAVE_CHAIN_STATE)
        LAST_PATH["edgewise_fwd"] = path
        saved = _bytes(lib.mopk_edgewise_saved_bytes(C.byref(a)), dev)
        ws = _bytes(256 if path == L.PATH_FUSED else lib.mopk_edgewise_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        if _KEEP_WS:
            LAST_PATH["_fwd_ws"] = ws  # diagnostics only (stamp builds read it back): pins the buffer until the next call
        with _timed("edgewise_fwd"):
            rc = lib.mopk_edgewise_lowrank_fwd(C.byref(a), _stream())
        L.check(rc, "mopk_edgewise_lowrank_fwd")
        ctx.save_for_backward(qkv, saved, *f.values(), *extras)
        ctx.meta = (beta_not, V, prec, path, r, int(a.save_for_backward), drop)
        return y.view(B, N, H * dk)

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        qkv, saved, sqk, vs0, vsL, Wr, br, Wc, bc, logit, *extras = ctx.saved_tensors
        beta_not, V, prec, path, r, sfb, drop = ctx.meta
        B, N, Vq, _, H, dk = qkv.shape
        dev = qkv.device
        dy = dy.contiguous()
        if dy.dtype != qkv.dtype:
            dy = dy.to(qkv.dtype)
        a = L.EdgewiseArgs()
        a.B, a.H, a.N, a.dk, a.V, a.r = B, H, N, dk, V, r
        a.io_dtype, a.precision, a.path, a.beta_not = _io_dtype(qkv), prec, path, float(beta_not)
        a.save_for_backward = sfb
        a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
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
        n_extra = extras[0].shape[2] if extras else 0
        C_ = 2 * V + 2 + n_extra
        d_extras = None
        if n_extra:
            d_extras = (torch.empty_like(extras[0]), torch.empty_like(extras[1]))
            ext = L.EdgewiseExt()
            ext.n_extra, ext.row_extra, ext.col_extra = n_extra, extras[0].data_ptr(), extras[1].data_ptr()
            ext.d_row_extra, ext.d_col_extra = d_extras[0].data_ptr(), d_extras[1].data_ptr()
            a.ext = C.pointer(ext)
        # per-batch partials of the small gradients, and ONE buffer for their final values (reduced by the library in a single
        # launch, cast to the parameters' dtype in a single kernel, handed to autograd as views)
        n_sqk, n_vs, n_w, n_b = V * H * dk, H * dk, 4 * r * C_, 4 * r
        parts = torch.empty(B * (n_sqk + 2 * n_vs + H), **f32)
        dsqk_p, dvs0_p, dvsL_p, dlg_p = torch.split(parts, [B * n_sqk, B * n_vs, B * n_vs, B * H])
        small = torch.empty(n_sqk + 2 * n_vs + 2 * n_w + 2 * n_b + 1, **f32)
        dsqk, dvs0, dvsL, dWr, dbr, dWc, dbc, dlg = torch.split(small, [n_sqk, n_vs, n_vs, n_w, n_b, n_w, n_b, 1])
        a.dsqk_part, a.dvs0_part, a.dvsL_part = dsqk_p.data_ptr(), dvs0_p.data_ptr(), dvsL_p.data_ptr()
        a.dWr, a.dbr, a.dWc, a.dbc = dWr.data_ptr(), dbr.data_ptr(), dWc.data_ptr(), dbc.data_ptr()
        a.dlogit_part = dlg_p.data_ptr()
        LAST_PATH["edgewise_bwd"] = path
        ws = _bytes(lib.mopk_edgewise_workspace_bytes(C.byref(a)), dev)
        if _KEEP_WS:
            LAST_PATH["_bwd_ws"] = ws  # diagnostics only (stamp builds read it back): pins the buffer until the next call
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("edgewise_bwd"):
            rc = lib.mopk_edgewise_lowrank_bwd(C.byref(a), _stream())
        L.check(rc, "mopk_edgewise_lowrank_bwd")
        L.check(lib.mopk_edgewise_reduce_parts(C.byref(a), dsqk.data_ptr(), dvs0.data_ptr(), dvsL.data_ptr(), dlg.data_ptr(),
                                               _stream()), "mopk_edgewise_reduce_parts")
        dlens = None
        if n_extra:     # the lens means are functions of q, k, sqk and the lens weights: their gradients join the kernels' own
            dsqk_x, dlens = lens_means_bwd_hip(d_extras[0], d_extras[1], qkv, dqkv, sqk, extras[2], ctx.lens_dil)
            dsqk.add_(dsqk_x.reshape(-1))
            dlens = dlens.to(ctx.lens_dtype)
        if ctx.small_dtype is not None and ctx.small_dtype != torch.float32:
            small = small.to(ctx.small_dtype)
            dsqk, dvs0, dvsL, dWr, dbr, dWc, dbc, dlg = torch.split(small, [n_sqk, n_vs, n_vs, n_w, n_b, n_w, n_b, 1])
        return (dqkv, dsqk.view(V, H, dk), dvs0.view(H, dk), dvsL.view(H, dk), dWr.view(4 * r, C_), dbr, dWc.view(4 * r, C_), dbc,
                dlg.reshape(()), None, None, None, None, None, None, dlens, None)


class EdgewiseVariant:
    """Static description of a non-default gate head / lens bank (MopkEdgewiseExt, generic path)."""

    def __init__(self, dense: bool = False, use_k3: bool = False, lens_dilations=()):
        self.dense, self.use_k3, self.lens_dilations = bool(dense), bool(use_k3), tuple(int(d) for d in lens_dilations)
        if len(self.lens_dilations) > L.MAX_LENS:
            raise ValueError(f"at most {L.MAX_LENS} lens dilations are supported")


def _fill_ext(ext: L.EdgewiseExt, var: EdgewiseVariant, t: dict):
    ext.gate_mode, ext.use_k3, ext.n_lens = int(var.dense), int(var.use_k3), len(var.lens_dilations)
    for i, d in enumerate(var.lens_dilations):
        ext.lens_dil[i] = d
    for k in ("lens_w", "W1", "b1", "W3", "b3", "W2", "b2"):
        if t.get(k) is not None:
            setattr(ext, k, t[k].data_ptr())


class _EdgewiseGeneralFn(torch.autograd.Function):
    """EdgewiseMSA core with a dense gate head and/or an S lens bank (reference :250-272, :312-318, :425-442, :523-533).

    tensor inputs: qkv, sqk, vs0, vsL, logit, head (4 tensors: low-rank Wr,br,Wc,bc | dense W1,b1,W2,b2), W3, b3, lens_w
    (unused ones are passed as empty tensors)."""

    @staticmethod
    def forward(ctx, qkv, sqk, vs0, vsL, logit, h0, h1, h2, h3, W3, b3, lens_w, beta_not, V, prec, var, wants_grad=True, drop=(0.0, 0),
                mask=None):
        _require_gpu(qkv, "EdgewiseMSA")
        lib = L.lib()
        B, N, Vq, _, H, dk = qkv.shape
        qkv = qkv.contiguous()
        dev = qkv.device
        f = dict(sqk=_f32c(sqk), vs0=_f32c(vs0), vsL=_f32c(vsL), logit=_f32c(logit).reshape(1),
                 h0=_f32c(h0), h1=_f32c(h1), h2=_f32c(h2), h3=_f32c(h3), W3=_f32c(W3), b3=_f32c(b3), lens_w=_f32c(lens_w))
        a, ext = L.EdgewiseArgs(), L.EdgewiseExt()
        a.B, a.H, a.N, a.dk, a.V = B, H, N, dk, V
        a.r = 1 if var.dense else f["h0"].shape[0] // 4
        a.io_dtype, a.precision, a.path, a.beta_not = _io_dtype(qkv), prec, L.PATH_GENERIC, float(beta_not)
        _ew_views(a, qkv, "")
        a.sqk, a.vs0, a.vsL, a.chain_logit = f["sqk"].data_ptr(), f["vs0"].data_ptr(), f["vsL"].data_ptr(), f["logit"].data_ptr()
        if var.dense:
            _fill_ext(ext, var, dict(lens_w=f["lens_w"] if var.lens_dilations else None, W1=f["h0"], b1=f["h1"], W2=f["h2"], b2=f["h3"],
                                     W3=f["W3"] if var.use_k3 else None, b3=f["b3"] if var.use_k3 else None))
        else:
            a.Wr, a.br, a.Wc, a.bc = f["h0"].data_ptr(), f["h1"].data_ptr(), f["h2"].data_ptr(), f["h3"].data_ptr()
            _fill_ext(ext, var, dict(lens_w=f["lens_w"] if var.lens_dilations else None))
        a.ext = C.pointer(ext)
        y = torch.empty(B, N, H, dk, dtype=qkv.dtype, device=dev)
        a.y = L.View4(y.data_ptr(), N * H * dk, dk, H * dk)
        # dense head without the 3x3 convolution / lens bank: the fused kernels evaluate it inside their mix loops (the backward
        # keeps dW1[k][:] and db1[k] in one 16-slot row, i.e. covers V <= 6)
        m8, ms = _mask_u8(mask, B, H, N, dev)
        a.mask, (a.mask_sb, a.mask_sh, a.mask_si) = _ptr(m8), ms
        if var.dense and not var.use_k3 and not var.lens_dilations and m8 is None and _PATH != L.PATH_GENERIC and (not wants_grad or V <= 6):
            a.path, a.save_for_backward = L.PATH_FUSED, 1
            if not lib.mopk_edgewise_fused_supported(C.byref(a)):
                a.path, a.save_for_backward = L.PATH_GENERIC, 0
        a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
        LAST_PATH["edgewise_fwd"] = int(a.path)
        saved = _bytes(lib.mopk_edgewise_saved_bytes(C.byref(a)), dev)
        ws = _bytes(256 if a.path == L.PATH_FUSED else lib.mopk_edgewise_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("edgewise_fwd"):
            rc = lib.mopk_edgewise_fwd(C.byref(a), _stream())
        L.check(rc, "mopk_edgewise_fwd")
        ctx.save_for_backward(qkv, saved, *f.values())
        ctx.keys = list(f.keys())
        ctx.meta = (beta_not, V, prec, var, int(a.r))
        ctx.fwd_path, ctx.drop, ctx.mask = int(a.path), drop, (m8, ms)
        return y.view(B, N, H * dk)

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        qkv, saved, *rest = ctx.saved_tensors
        f = dict(zip(ctx.keys, rest))
        beta_not, V, prec, var, r = ctx.meta
        path = ctx.fwd_path
        B, N, Vq, _, H, dk = qkv.shape
        dev = qkv.device
        dy = dy.contiguous()
        if dy.dtype != qkv.dtype:
            dy = dy.to(qkv.dtype)
        a, ext = L.EdgewiseArgs(), L.EdgewiseExt()
        a.B, a.H, a.N, a.dk, a.V, a.r = B, H, N, dk, V, r
        a.io_dtype, a.precision, a.path, a.beta_not = _io_dtype(qkv), prec, path, float(beta_not)
        a.save_for_backward = int(path == L.PATH_FUSED)
        a.dropout_p, a.dropout_seed = float(ctx.drop[0]), int(ctx.drop[1])
        a.mask, (a.mask_sb, a.mask_sh, a.mask_si) = _ptr(ctx.mask[0]), ctx.mask[1]
        _ew_views(a, qkv, "")
        a.sqk, a.vs0, a.vsL, a.chain_logit = f["sqk"].data_ptr(), f["vs0"].data_ptr(), f["vsL"].data_ptr(), f["logit"].data_ptr()
        a.y = L.View4(dy.data_ptr(), N * H * dk, dk, H * dk)
        a.dy = L.View4(dy.data_ptr(), N * H * dk, dk, H * dk)
        dqkv = (torch.empty_like(qkv) if Vq == 1 else torch.zeros_like(qkv))
        _ew_views(a, dqkv, "d")
        f32 = dict(dtype=torch.float32, device=dev)
        dsqk, dvs0, dvsL, dlg = torch.empty(B, V, H, dk, **f32), torch.empty(B, H, dk, **f32), torch.empty(B, H, dk, **f32), torch.empty(B, H, **f32)
        a.dsqk_part, a.dvs0_part, a.dvsL_part, a.dlogit_part = dsqk.data_ptr(), dvs0.data_ptr(), dvsL.data_ptr(), dlg.data_ptr()
        g = {k: torch.zeros_like(f[k]) for k in ("h0", "h1", "h2", "h3", "W3", "b3", "lens_w")}
        lens = f["lens_w"] if var.lens_dilations else None
        if var.dense:
            _fill_ext(ext, var, dict(lens_w=lens, W1=f["h0"], b1=f["h1"], W2=f["h2"], b2=f["h3"],
                                     W3=f["W3"] if var.use_k3 else None, b3=f["b3"] if var.use_k3 else None))
            ext.dW1, ext.db1, ext.dW2, ext.db2 = g["h0"].data_ptr(), g["h1"].data_ptr(), g["h2"].data_ptr(), g["h3"].data_ptr()
            if var.use_k3:
                ext.dW3, ext.db3 = g["W3"].data_ptr(), g["b3"].data_ptr()
        else:
            a.Wr, a.br, a.Wc, a.bc = f["h0"].data_ptr(), f["h1"].data_ptr(), f["h2"].data_ptr(), f["h3"].data_ptr()
            a.dWr, a.dbr, a.dWc, a.dbc = g["h0"].data_ptr(), g["h1"].data_ptr(), g["h2"].data_ptr(), g["h3"].data_ptr()
            _fill_ext(ext, var, dict(lens_w=lens))
        if var.lens_dilations:
            ext.dlens_w = g["lens_w"].data_ptr()
        a.ext = C.pointer(ext)
        LAST_PATH["edgewise_bwd"] = path
        ws = _bytes(lib.mopk_edgewise_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("edgewise_bwd"):
            rc = lib.mopk_edgewise_bwd(C.byref(a), _stream())
        L.check(rc, "mopk_edgewise_bwd")
        return (dqkv, dsqk.sum(0), dvs0.sum(0), dvsL.sum(0), dlg.sum().reshape(()), g["h0"], g["h1"], g["h2"], g["h3"],
                g["W3"], g["b3"], g["lens_w"], None, None, None, None, None, None, None)


def _half_via_fp32(fn):
    """float16 callers (`module.half()`, fp16 autocast): the kernels take float32 / bfloat16 only, so half tensors go through the
    float32 arithmetic (a superset of fp16) and the result is cast back; autograd carries the casts."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kw):
        def is_h(a):
            return isinstance(a, torch.Tensor) and a.dtype == torch.float16
        flat = list(args) + list(kw.values())
        if not any(is_h(a) or (isinstance(a, (tuple, list)) and any(is_h(b) for b in a)) for a in flat):
            return fn(*args, **kw)
        def up(a):
            if is_h(a):
                return a.float()
            if isinstance(a, (tuple, list)):
                return type(a)(up(b) for b in a)
            return a
        out = fn(*[up(a) for a in args], **{k: up(v) for k, v in kw.items()})
        def down(o):
            return o.half() if isinstance(o, torch.Tensor) and o.dtype == torch.float32 and o.dim() == 3 else o
        return tuple(down(o) for o in out) if isinstance(out, tuple) else down(out)
    return wrapped


def _empty_batch(t: torch.Tensor, *shape) -> torch.Tensor:
    """B = 0: the reference's torch ops return an empty result; the C ABI rejects empty shapes.  An empty tensor that still hangs
    on `t` in the autograd graph (its backward hands `t` an empty gradient)."""
    return t.new_zeros(shape) + t.sum() * 0


@_half_via_fp32
def edgewise_general_core(qkv, sqk, vs0, vsL, chain_logit, head, beta_not: float, n_views: int, variant: EdgewiseVariant,
                          W3=None, b3=None, lens_w=None, precision: Optional[int] = None, dropout_p: float = 0.0,
                          seed: Optional[int] = None, attn_mask=None):
    """EdgewiseMSA core for the dense gate head and/or the S lens bank.  head = (Wr, br, Wc, bc) for the low-rank head
    with C = 2V+2+L*V input channels, or (W1 (16,C), b1, W2 (4,16), b2) for the dense head; W3/b3 with use_k3;
    lens_w (L,V,3,3).  qkv as in edgewise_lowrank_core."""
    if qkv.shape[0] == 0:
        return _empty_batch(qkv, 0, qkv.shape[1], qkv.shape[-2] * qkv.shape[-1])
    prec = _prec_for(qkv.dtype) if precision is None else precision
    e = qkv.new_zeros(0, dtype=torch.float32)
    drop = (float(dropout_p), (dropout_seed() if seed is None else int(seed))) if dropout_p > 0 else (0.0, 0)
    # decided here: inside Function.forward autograd is already switched off
    wants_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                 for t in (qkv, sqk, vs0, vsL, chain_logit, *head, W3, b3, lens_w))
    return _EdgewiseGeneralFn.apply(qkv, sqk, vs0, vsL, chain_logit, *head, e if W3 is None else W3, e if b3 is None else b3,
                                    e if lens_w is None else lens_w, beta_not, n_views, prec, variant, wants_grad, drop, attn_mask)


def lens_mean_features(qkv, sqk, lens_w, dilations):
    """Row / column means of the S lens bank's planes (reference attention_variants.py:523-533) WITHOUT the planes.

    The bank convolves each score plane S_v = (q * sqk_v) k^T with a depthwise dilated 3x3 kernel w (zero padding = dilation d); the
    low-rank head only ever sees row and column means of the result (:323-326), and those are linear functionals of q and k:
        row mean [i] = 1/N sum_a sum_b w[a][b] R_b[i + (a-1) d],   R_b[i'] = q_v[i'] . sum_{j' in J_b} k[j']
        col mean [j] = 1/N sum_a sum_b w[a][b] C_a[j + (b-1) d],   C_a[j'] = k[j'] . sum_{i' in J_a} q_v[i']
    with J_0 = [0, N-d), J_1 = [0, N), J_2 = [d, N) the source rows / columns a kernel tap can reach.  O(N dk) work per (b, h, v)
    in plain torch ops (autograd carries their backward into q, k, sqk and the lens weights).

    qkv (B,N,1,3,H,dk) shared q / k; sqk (V,H,dk); lens_w (L,V,3,3); -> row, col (B,H,L*V,N) float32, channels l-major as :531."""
    B, N, _, _, H, dk = qkv.shape
    V = sqk.shape[0]
    q = qkv[:, :, 0, 0].permute(0, 2, 1, 3).float()           # (B,H,N,dk)
    k = qkv[:, :, 0, 1].permute(0, 2, 1, 3).float()
    sq = sqk.float()
    L_ = len(dilations)
    dil = [int(d) for d in dilations]
    dmax = max(dil)
    qsum, ksum = q.sum(2), k.sum(2)
    # sums over J_0 = all minus the last d tokens, J_1 = all, J_2 = all minus the first d (slices clamp at N: d >= N leaves nothing)
    ks = torch.stack([torch.stack([ksum - k[:, :, max(N - d, 0):].sum(2), ksum, ksum - k[:, :, :d].sum(2)], 2) for d in dil], 2)    # (B,H,L,3,dk)
    qs = torch.stack([torch.stack([qsum - q[:, :, max(N - d, 0):].sum(2), qsum, qsum - q[:, :, :d].sum(2)], 2) for d in dil], 2)
    w = lens_w.float() / N                                                                  # (L,V,3,3) [a][b]
    # the taps are folded into the (tiny) left operands first: ONE batched GEMM per side then yields, per lens, view and shift, the
    # vector that the shift-and-add below turns into the mean
    ua = torch.einsum("lvst,vhd,nhltd->nhlvsd", w, sq, ks).reshape(B, H, L_ * V * 3, dk)        # row side: shift index a (= s)
    ub = torch.einsum("lvst,vhd,nhlsd->nhlvtd", w, sq, qs).reshape(B, H, L_ * V * 3, dk)        # col side: shift index b (= t)
    Tr = F.pad(torch.matmul(ua, q.transpose(2, 3)).view(B, H, L_, V, 3, N), (dmax, dmax))      # [.., a, dmax + i]
    Tc = F.pad(torch.matmul(ub, k.transpose(2, 3)).view(B, H, L_, V, 3, N), (dmax, dmax))
    def shift_add(T, l, d):                                                                  # sum_a T_a[i + (a-1) d]
        return T[:, :, l, :, 0, dmax - d:dmax - d + N] + T[:, :, l, :, 1, dmax:dmax + N] + T[:, :, l, :, 2, dmax + d:dmax + d + N]
    return (torch.cat([shift_add(Tr, l, d) for l, d in enumerate(dil)], 2),
            torch.cat([shift_add(Tc, l, d) for l, d in enumerate(dil)], 2))


def _lens_set_sums(x, dil):
    """x (B,H,N,dk) -> (B,H,L,3,dk): sums of x over the tokens J_0 = [0, N-d), J_1 = all, J_2 = [d, N) per dilation (all minus the d edge rows)"""
    N = x.shape[2]
    tot = x.sum(2)
    return torch.stack([torch.stack([tot - x[:, :, N - d:].sum(2), tot, tot - x[:, :, :d].sum(2)], 2) for d in dil], 2)


def _lens_set_sums_adjoint(dx, g, dil):
    """dx (B,H,N,dk) += adjoint of _lens_set_sums applied to g (B,H,L,3,dk), in place"""
    N = dx.shape[2]
    dx += g.sum((2, 3))[:, :, None, :]
    for l, d in enumerate(dil):
        dx[:, :, N - d:] -= g[:, :, l, 0][:, :, None, :]
        dx[:, :, :d] -= g[:, :, l, 2][:, :, None, :]
    return dx


def _lens_shift_add(T, dil, dmax, N):
    """T (B,H,L,V,3,N + 2 dmax), zero padded by dmax: -> (B,H,L*V,N)  sum_a T[l,v,a][i + (a-1) d_l]"""
    return torch.cat([T[:, :, l, :, 0, dmax - d:dmax - d + N] + T[:, :, l, :, 1, dmax:dmax + N] + T[:, :, l, :, 2, dmax + d:dmax + d + N]
                      for l, d in enumerate(dil)], 2)


def _lens_unshift(dM, dil, dmax, N, V):
    """adjoint of _lens_shift_add: dM (B,H,L*V,N) -> (B,H,L*V*3,N)  dT[l,v,a][i'] = dM[l,v][i' - (a-1) d_l] (0 outside)"""
    B, H = dM.shape[:2]
    P = F.pad(dM.view(B, H, len(dil), V, N), (dmax, dmax))
    return torch.stack([torch.stack([P[:, :, l, :, dmax + d:dmax + d + N], P[:, :, l, :, dmax:dmax + N], P[:, :, l, :, dmax - d:dmax - d + N]], 3)
                        for l, d in enumerate(dil)], 2).reshape(B, H, -1, N)


def _lens_args(qkv, sqk, lens_w, dilations) -> L.LensMeansArgs:
    B, N, _, _, H, dk = qkv.shape
    a = L.LensMeansArgs()
    a.B, a.H, a.N, a.dk, a.V, a.L = B, H, N, dk, sqk.shape[0], len(dilations)
    a.io_dtype = _io_dtype(qkv)
    for i, d in enumerate(dilations):
        a.dil[i] = int(d)
    s_n = 3 * H * dk
    a.q = L.View4(qkv.data_ptr(), N * s_n, dk, s_n)
    a.k = L.View4(qkv.data_ptr() + H * dk * qkv.element_size(), N * s_n, dk, s_n)
    a.sqk, a.lens_w = sqk.data_ptr(), lens_w.data_ptr()
    return a


def lens_means_hip(qkv, sqk, lens_w, dilations):
    """row / col means of the S lens bank's planes (B,H,L*V,N) fp32 from libmopk's closed-form kernel (mopk_lens_means_fwd);
    qkv (B,N,1,3,H,dk) contiguous on the GPU, sqk (V,H,dk) / lens_w (L,V,3,3) float32 contiguous."""
    _require_gpu(qkv, "lens means")
    B, N, _, _, H, dk = qkv.shape
    E = lens_w.shape[0] * lens_w.shape[1]
    a = _lens_args(qkv, sqk, lens_w, dilations)
    row, col = torch.empty(B, H, E, N, dtype=torch.float32, device=qkv.device), torch.empty(B, H, E, N, dtype=torch.float32, device=qkv.device)
    a.row, a.col = row.data_ptr(), col.data_ptr()
    L.check(L.lib().mopk_lens_means_fwd(C.byref(a), _stream()), "mopk_lens_means_fwd")
    return row, col


def lens_means_bwd_hip(d_row, d_col, qkv, dqkv, sqk, lens_w, dilations):
    """backward of `lens_means_hip`: ADDS the q / k gradients into dqkv (same layout as qkv) and returns dsqk (V,H,dk), dlens_w (L,V,3,3)"""
    B, N, _, _, H, dk = qkv.shape
    V, Ln = sqk.shape[0], lens_w.shape[0]
    a = _lens_args(qkv, sqk, lens_w, dilations)
    s_n = 3 * H * dk
    a.dq = L.View4(dqkv.data_ptr(), N * s_n, dk, s_n)
    a.dk_ = L.View4(dqkv.data_ptr() + H * dk * dqkv.element_size(), N * s_n, dk, s_n)
    dsqk_p = torch.empty(B, V, H, dk, dtype=torch.float32, device=qkv.device)
    dlens_p = torch.empty(B * H, Ln, V, 3, 3, dtype=torch.float32, device=qkv.device)
    a.d_row, a.d_col, a.dsqk_part, a.dlens_part = d_row.data_ptr(), d_col.data_ptr(), dsqk_p.data_ptr(), dlens_p.data_ptr()
    L.check(L.lib().mopk_lens_means_bwd(C.byref(a), _stream()), "mopk_lens_means_bwd")
    return dsqk_p.sum(0), dlens_p.sum(0)


def _lens_means_fwd(qkv, sqk, lens_w, dilations):
    """`lens_mean_features` without an autograd graph, in torch ops: -> row, col (B,H,L*V,N), state for `_lens_means_bwd`.  The statement
    the HIP kernels (`lens_means_hip`, mop_amd/csrc/lens_means.hip) are tested against; the product path calls the kernels."""
    B, N, _, _, H, dk = qkv.shape
    V, L_ = sqk.shape[0], len(dilations)
    dil = [min(int(d), N) for d in dilations]
    dmax = max(dil)
    q = qkv[:, :, 0, 0].permute(0, 2, 1, 3).float()
    k = qkv[:, :, 0, 1].permute(0, 2, 1, 3).float()
    ks, qs = _lens_set_sums(k, dil), _lens_set_sums(q, dil)
    w = lens_w.float() / N
    ua = torch.einsum("lvst,vhd,nhltd->nhlvsd", w, sqk, ks).reshape(B, H, L_ * V * 3, dk)
    ub = torch.einsum("lvst,vhd,nhlsd->nhlvtd", w, sqk, qs).reshape(B, H, L_ * V * 3, dk)
    Tr = F.pad(torch.matmul(ua, q.transpose(2, 3)).view(B, H, L_, V, 3, N), (dmax, dmax))
    Tc = F.pad(torch.matmul(ub, k.transpose(2, 3)).view(B, H, L_, V, 3, N), (dmax, dmax))
    return _lens_shift_add(Tr, dil, dmax, N), _lens_shift_add(Tc, dil, dmax, N), (ks, qs, ua, ub)


def _lens_means_bwd(d_row, d_col, qkv, sqk, lens_w, dilations, state):
    """-> dq, dk (B,H,N,dk) float32, dsqk (V,H,dk), dlens_w (L,V,3,3)"""
    B, N, _, _, H, dk = qkv.shape
    V, L_ = sqk.shape[0], len(dilations)
    dil = [min(int(d), N) for d in dilations]
    dmax = max(dil)
    ks, qs, ua, ub = state
    q = qkv[:, :, 0, 0].permute(0, 2, 1, 3).float()
    k = qkv[:, :, 0, 1].permute(0, 2, 1, 3).float()
    w = lens_w.float() / N
    dTr, dTc = _lens_unshift(d_row, dil, dmax, N, V), _lens_unshift(d_col, dil, dmax, N, V)       # (B,H,L*V*3,N)
    dua = torch.matmul(dTr, q).view(B, H, L_, V, 3, dk)
    dub = torch.matmul(dTc, k).view(B, H, L_, V, 3, dk)
    dq = torch.matmul(dTr.transpose(2, 3), ua)
    dk_ = torch.matmul(dTc.transpose(2, 3), ub)
    # through ua = w . sqk . ks and ub = w . sqk . qs
    gs_a, gs_b = dua * sqk.permute(1, 0, 2)[None, :, None, :, None, :], dub * sqk.permute(1, 0, 2)[None, :, None, :, None, :]   # (B,H,L,V,3,dk)
    # contractions over (batch, head, d) with a 3 x 3 result per (l, v): a broadcast product + one reduction (as a GEMM they are
    # K = B H dk deep with a 30 x 3 output -- one workgroup's worth of parallelism)
    dw = ((gs_a[:, :, :, :, :, None, :] * ks[:, :, :, None, None, :, :]).sum((0, 1, 6)) +
          (gs_b[:, :, :, :, None, :, :] * qs[:, :, :, None, :, None, :]).sum((0, 1, 6))) / N
    wks = torch.einsum("lvst,nhltd->nhlvsd", w, ks)
    wqs = torch.einsum("lvst,nhlsd->nhlvtd", w, qs)
    dsqk = (dua * wks + dub * wqs).sum((0, 2, 4)).permute(1, 0, 2)                               # (V,H,dk)
    dks = torch.einsum("nhlvsd,lvst->nhltd", gs_a, w).reshape(B, H, L_ * 3, dk)
    dqs = torch.einsum("nhlvtd,lvst->nhlsd", gs_b, w).reshape(B, H, L_ * 3, dk)
    return _lens_set_sums_adjoint(dq, dqs.view(B, H, L_, 3, dk), dil), _lens_set_sums_adjoint(dk_, dks.view(B, H, L_, 3, dk), dil), dsqk, dw


def lowrank_lens_fused_supported(qkv, n_views: int, rank: int, dilations, precision: Optional[int] = None) -> bool:
    """True when the fused kernels take this low-rank call with an S lens bank of these dilations (as extra mean-feature channels) and
    libmopk's closed-form kernels (mopk_lens_means_*) take the bank"""
    n_lens = len(dilations)
    if not qkv.is_cuda or qkv.shape[2] != 1 or qkv.shape[0] == 0 or _PATH == L.PATH_GENERIC or qkv.dtype == torch.float16:
        return False
    if not 1 <= n_lens <= L.MAX_LENS:
        return False
    B, N, _, _, H, dk = qkv.shape
    lm = L.LensMeansArgs()
    lm.B, lm.H, lm.N, lm.dk, lm.V, lm.L = B, H, N, dk, n_views, n_lens
    for i, d in enumerate(dilations):
        lm.dil[i] = int(d)
    if not L.lib().mopk_lens_means_supported(C.byref(lm), 1):
        return False
    a, ext = L.EdgewiseArgs(), L.EdgewiseExt()
    a.B, a.H, a.N, a.dk, a.V, a.r = B, H, N, dk, n_views, rank
    a.io_dtype, a.path = _io_dtype(qkv), L.PATH_FUSED
    a.precision = _prec_for(qkv.dtype) if precision is None else precision
    _ew_views(a, qkv, "")
    a.y = L.View4(qkv.data_ptr(), N * H * dk, dk, H * dk)
    ext.n_extra, ext.row_extra, ext.col_extra = n_lens * n_views, qkv.data_ptr(), qkv.data_ptr()      # non-null stand-ins: nothing is read
    a.ext = C.pointer(ext)
    return bool(L.lib().mopk_edgewise_fused_supported(C.byref(a)))


@_half_via_fp32
def edgewise_lowrank_core(qkv, sqk, vs0, vsL, Wr, br, Wc, bc, chain_logit, beta_not: float,
                          n_views: int, precision: Optional[int] = None, path: Optional[int] = None,
                          dropout_p: float = 0.0, seed: Optional[int] = None, lens_w=None, lens_dilations=()):
    """qkv: (B,N,Vq,3,H,dk) with Vq in {1 (share_qkv), n_views}; returns (B,N,H*dk).  dropout_p > 0: attn_drop on the mixed
    attention weights (:552) inside the fused kernels (see `sdpa_core`).  lens_w (L,V,3,3) + lens_dilations: the S lens bank, fed to
    the fused kernels as extra mean-feature channels (`lens_mean_features` states the closed form; the Function evaluates it and its
    backward by hand, `_lens_means_fwd` / `_lens_means_bwd`; callers check `lowrank_lens_fused_supported` first)."""
    if qkv.shape[0] == 0:
        return _empty_batch(qkv, 0, qkv.shape[1], qkv.shape[-2] * qkv.shape[-1])
    drop = (float(dropout_p), (dropout_seed() if seed is None else int(seed))) if dropout_p > 0 else (0.0, 0)
    prec = _prec_for(qkv.dtype) if precision is None else precision
    want_bwd = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (qkv, sqk, vs0, vsL, Wr, br, Wc, bc, chain_logit, lens_w))
    return _EdgewiseLowrankFn.apply(qkv, sqk, vs0, vsL, Wr, br, Wc, bc, chain_logit, beta_not,
                                    n_views, prec, _PATH if path is None else path, want_bwd, drop, lens_w, tuple(lens_dilations))


# --------------------------------------------------------------------------------------
# sibling cores: plain SDPA, dual-path (MultiHopMSA), Quartet
# --------------------------------------------------------------------------------------
def _v4(t: torch.Tensor) -> L.View4:
    """t: (B,N,H,dk) view with unit inner stride -> MopkView4 (strides b,h,n in elements)."""
    assert t.dim() == 4 and t.stride(3) == 1, "inner (head_dim) stride must be 1"
    return L.View4(t.data_ptr(), t.stride(0), t.stride(2), t.stride(1))


def _heads_view(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


def _mask_u8(attn_mask, B, H, N, dev):
    """reference convention: broadcastable to (B,H,N,N), 0 = blocked (attention_variants.py:43-44)."""
    if attn_mask is None:
        return None, (0, 0, 0)
    m = (attn_mask.to(dev) != 0).to(torch.uint8)
    while m.dim() < 4:
        m = m.unsqueeze(0)
    m = m.contiguous().expand(B, H, N, N)
    assert m.stride(3) == 1 or N == 1
    return m, (m.stride(0), m.stride(1), m.stride(2))


def _bias_f32(bias, B, H, N, dev):
    if bias is None:
        return None, (0, 0, 0)
    b = bias.to(dev, torch.float32)
    while b.dim() < 4:
        b = b.unsqueeze(0)
    b = b.contiguous().expand(B, H, N, N)
    return b, (b.stride(0), b.stride(1), b.stride(2))


class _SdpaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, bias, causal, prec, path, drop=(0.0, 0)):
        _require_gpu(q, "SDPA")
        lib = L.lib()
        # packed form: q is the (B,N,3,H,dk) output of one qkv projection and k = v = None.  The kernels read the three strided views and
        # the backward writes ONE packed gradient -- autograd then has no per-view zero-fill + copy + add to assemble it from three
        ctx.packed = k is None
        if ctx.packed:
            q, k, v = q.contiguous().unbind(2)
        q, k, v = _heads_view(q), _heads_view(k), _heads_view(v)
        B, N, H, dk = q.shape
        dev = q.device
        a = L.SdpaArgs()
        a.B, a.H, a.N, a.dk = B, H, N, dk
        a.io_dtype, a.precision, a.path, a.causal = _io_dtype(q), prec, path, int(bool(causal))
        a.q, a.k, a.v = _v4(q), _v4(k), _v4(v)
        m8, ms = _mask_u8(mask, B, H, N, dev)
        bf, bs = _bias_f32(bias, B, H, N, dev)
        a.mask, (a.mask_sb, a.mask_sh, a.mask_si) = _ptr(m8), ms
        a.bias, (a.bias_sb, a.bias_sh, a.bias_si) = _ptr(bf), bs
        a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
        y = torch.empty(B, N, H, dk, dtype=q.dtype, device=dev)
        a.y = _v4(y)
        if path == L.PATH_AUTO:       # resolve once so forward, backward and the size queries agree
            path = L.PATH_FUSED if lib.mopk_sdpa_fused_supported(C.byref(a)) else L.PATH_GENERIC
            a.path = path
        LAST_PATH["sdpa_fwd"] = path
        saved = _bytes(lib.mopk_sdpa_saved_bytes(C.byref(a)), dev)
        ws = _bytes(lib.mopk_sdpa_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("sdpa_fwd"):
            rc = lib.mopk_sdpa_fwd(C.byref(a), _stream())
        L.check(rc, "mopk_sdpa_fwd")
        ctx.save_for_backward(q, k, v, y, saved)
        ctx.meta = (causal, prec, path, m8, ms, bf, bs, drop)
        return y.view(B, N, H * dk)

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        q, k, v, y, saved = ctx.saved_tensors
        causal, prec, path, m8, ms, bf, bs, drop = ctx.meta
        B, N, H, dk = q.shape
        dev = q.device
        dy = dy.contiguous().to(q.dtype).view(B, N, H, dk)
        a = L.SdpaArgs()
        a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
        a.B, a.H, a.N, a.dk = B, H, N, dk
        a.io_dtype, a.precision, a.path, a.causal = _io_dtype(q), prec, path, int(bool(causal))
        a.q, a.k, a.v, a.y, a.dy = _v4(q), _v4(k), _v4(v), _v4(y), _v4(dy)
        a.mask, (a.mask_sb, a.mask_sh, a.mask_si) = _ptr(m8), ms
        a.bias, (a.bias_sb, a.bias_sh, a.bias_si) = _ptr(bf), bs
        if ctx.packed:
            dqkv = torch.empty(B, N, 3, H, dk, dtype=q.dtype, device=dev)
            dq, dk_, dv = dqkv.unbind(2)
        else:
            dq, dk_, dv = (torch.empty(B, N, H, dk, dtype=q.dtype, device=dev) for _ in range(3))
        a.dq, a.dk_, a.dv = _v4(dq), _v4(dk_), _v4(dv)
        LAST_PATH["sdpa_bwd"] = path
        ws = _bytes(lib.mopk_sdpa_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("sdpa_bwd"):
            rc = lib.mopk_sdpa_bwd(C.byref(a), _stream())
        L.check(rc, "mopk_sdpa_bwd")
        if ctx.packed:
            return dqkv, None, None, None, None, None, None, None, None
        return dq, dk_, dv, None, None, None, None, None, None


def dropout_seed() -> int:
    """a fresh 63-bit seed for the in-kernel dropout mask, drawn from torch's CPU generator (so `torch.manual_seed` makes runs
    reproducible, like it does for `nn.Dropout`)"""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def dropout_keep_mask(seed: int, p: float, B: int, H: int, N: int) -> torch.Tensor:
    """the kernels' keep mask as a (B,H,N,N) bool tensor (host restatement of `mopk_dropout_keep`, vectorised; tests / debugging)"""
    import numpy as np
    def h32(x):
        x = x.astype(np.uint64)
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xffffffff)
        x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xffffffff)
        x ^= x >> np.uint64(16)
        return x
    M = np.uint64(0xffffffff)
    lo, hi = np.uint64(seed & 0xffffffff), np.uint64((seed >> 32) & 0xffffffff)
    t = float(p) * 4294967296.0
    thresh = np.uint64(0xffffffff if t >= 4294967295.0 else (1 if t < 1.0 else int(t)))
    bh = np.arange(B * H, dtype=np.uint64)[:, None, None]
    i = np.arange(N, dtype=np.uint64)[None, :, None]
    j = np.arange(N, dtype=np.uint64)[None, None, :]
    row = h32((i * np.uint64(0x9E3779B1) + bh * np.uint64(0x85EBCA77) + hi) & M) ^ lo
    keep = h32((row ^ ((j * np.uint64(0xC2B2AE3D)) & M)) & M) >= thresh
    return torch.from_numpy(keep.reshape(B, H, N, N)) if p > 0 else torch.ones(B, H, N, N, dtype=torch.bool)


@_half_via_fp32
def sdpa_core(q, k=None, v=None, attn_mask=None, bias=None, causal=False, dropout_p: float = 0.0, seed: Optional[int] = None):
    """q,k,v: (B,N,H,dk) views -- or packed: q = the (B,N,3,H,dk) output of one qkv projection, k = v = None (one packed gradient comes
    back).  Returns (B,N,H*dk).  attn_mask: 0 = blocked; bias: additive.  dropout_p > 0: the probabilities are multiplied by
    keep / (1 - p) (mask = `dropout_keep_mask(seed, ...)`, seed drawn when None)."""
    if q.shape[0] == 0:
        if k is None:
            return _empty_batch(q, 0, q.shape[1], q.shape[-2] * q.shape[-1])
        return _empty_batch(q, 0, q.shape[1], q.shape[2] * q.shape[3]) + (k.sum() + v.sum()) * 0
    drop = (float(dropout_p), (dropout_seed() if seed is None else int(seed))) if dropout_p > 0 else (0.0, 0)
    return _SdpaFn.apply(q, k, v, attn_mask, bias, causal, _prec_for(q.dtype), _PATH, drop)


_ANCHOR_MODES = {"fixed": 0, "argmax_row_sum": 1}          # any other string -> row 0 (reference :141-145)


class _CrossViewFn(torch.autograd.Function):
    """CrossViewMixerMSA core, reference attention_variants.py:90-153."""

    @staticmethod
    def forward(ctx, q1, k1, v1, q2, k2, mix, cfg, mask, causal, prec, drop=(0.0, 0)):
        _require_gpu(q1, "CrossViewMixerMSA")
        lib = L.lib()
        ts = [_heads_view(t) for t in (q1, k1, v1, q2, k2)]
        B, N, H, dk = ts[0].shape
        dev = ts[0].device
        mx = _f32c(mix).reshape(4)
        a = L.CrossViewArgs()
        a.B, a.H, a.N, a.dk = B, H, N, dk
        a.io_dtype, a.precision, a.path, a.causal = _io_dtype(ts[0]), prec, L.PATH_GENERIC, int(bool(causal))
        a.t1, a.t2, a.prior_weight, a.use_prior, a.anchor_mode, a.fixed_k_star = cfg
        a.q1, a.k1, a.v1, a.q2, a.k2 = (_v4(t) for t in ts)
        a.mix = mx.data_ptr()
        m8, ms = _mask_u8(mask, B, H, N, dev)
        a.mask, (a.mask_sb, a.mask_sh, a.mask_si) = _ptr(m8), ms
        a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
        y = torch.empty(B, N, H, dk, dtype=ts[0].dtype, device=dev)
        a.y = _v4(y)
        kst = torch.zeros(B, H, dtype=torch.int32, device=dev)
        a.k_star = kst.data_ptr()
        saved = _bytes(lib.mopk_crossview_saved_bytes(C.byref(a)), dev)
        ws = _bytes(lib.mopk_crossview_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("crossview_fwd"):
            rc = lib.mopk_crossview_fwd(C.byref(a), _stream())
        L.check(rc, "mopk_crossview_fwd")
        LAST_PATH["crossview_k_star"] = kst
        ctx.save_for_backward(*ts, mx, saved)
        ctx.meta = (cfg, causal, prec, m8, ms, drop)
        return y.view(B, N, H * dk)

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        *ts, mx, saved = ctx.saved_tensors
        cfg, causal, prec, m8, ms, drop = ctx.meta
        B, N, H, dk = ts[0].shape
        dev = ts[0].device
        dy = dy.contiguous().to(ts[0].dtype).view(B, N, H, dk)
        a = L.CrossViewArgs()
        a.B, a.H, a.N, a.dk = B, H, N, dk
        a.io_dtype, a.precision, a.path, a.causal = _io_dtype(ts[0]), prec, L.PATH_GENERIC, int(bool(causal))
        a.t1, a.t2, a.prior_weight, a.use_prior, a.anchor_mode, a.fixed_k_star = cfg
        a.q1, a.k1, a.v1, a.q2, a.k2 = (_v4(t) for t in ts)
        a.mix = mx.data_ptr()
        a.mask, (a.mask_sb, a.mask_sh, a.mask_si) = _ptr(m8), ms
        a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
        a.y, a.dy = _v4(dy), _v4(dy)
        outs = [torch.empty(B, N, H, dk, dtype=ts[0].dtype, device=dev) for _ in range(5)]
        a.dq1, a.dk1, a.dv1, a.dq2, a.dk2 = (_v4(t) for t in outs)
        dmix = torch.empty(B, H, 4, dtype=torch.float32, device=dev)
        a.dmix_part = dmix.data_ptr()
        ws = _bytes(lib.mopk_crossview_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("crossview_bwd"):
            rc = lib.mopk_crossview_bwd(C.byref(a), _stream())
        L.check(rc, "mopk_crossview_bwd")
        return (*outs, dmix.sum((0, 1)).view(2, 2), None, None, None, None, None)


@_half_via_fp32
def crossview_core(q1, k1, v1, q2, k2, mix, t1=0.0, t2=0.0, prior_weight=0.0, anchor_mode="argmax_row_sum", fixed_k_star=0,
                   attn_mask=None, causal=False, dropout_p: float = 0.0, seed: Optional[int] = None):
    """q*,k*,v1: (B,N,H,dk) views; mix (2,2).  prior_weight = 0 disables the per-key prior.  Returns (B,N,H*dk)."""
    if q1.shape[0] == 0:
        return _empty_batch(q1, 0, q1.shape[1], q1.shape[2] * q1.shape[3])
    prec = _prec_for(q1.dtype)
    drop = (float(dropout_p), (dropout_seed() if seed is None else int(seed))) if dropout_p > 0 else (0.0, 0)
    if (prior_weight <= 0.0 and t1 == 0.0 and t2 == 0.0 and _PATH != L.PATH_GENERIC and prec == L.PREC_BF16
            and q1.shape[-1] in (32, 64)):
        # S = q1 (m11 k1 + m12 k2)^T + q2 (m21 k1 + m22 k2)^T: the 2x2 mix folds into two mixed key tensors (autograd carries
        # d mix, d k1, d k2) and the core is the fused two-score attention (dual-path kernels without the transport term)
        m = mix.to(k1.dtype)
        k1p = (m[0, 0] * k1 + m[0, 1] * k2).contiguous()
        k2p = (m[1, 0] * k1 + m[1, 1] * k2).contiguous()
        zero = q1.new_zeros((), dtype=torch.float32)
        LAST_PATH["crossview_fwd"] = L.PATH_FUSED
        return _DualPathFn.apply(q1, k1p, v1, q2, k2p, v1, zero, (1.0, 0.0, 0.0, 0.0), 0.0, 0, attn_mask, causal, prec, L.PATH_FUSED, drop)
    LAST_PATH["crossview_fwd"] = L.PATH_GENERIC
    cfg = (float(t1), float(t2), float(prior_weight), int(prior_weight > 0.0), _ANCHOR_MODES.get(anchor_mode, 2), int(fixed_k_star))
    return _CrossViewFn.apply(q1, k1, v1, q2, k2, mix, cfg, attn_mask, causal, prec, drop)


@_half_via_fp32
def crossview_core_packed(qkv1, qkv2, mix, t1=0.0, t2=0.0, prior_weight=0.0, anchor_mode="argmax_row_sum", fixed_k_star=0,
                          attn_mask=None, causal=False, dropout_p: float = 0.0, seed: Optional[int] = None):
    """`crossview_core` from the packed (B,N,3,H,dk) outputs of the two qkv projections: on the fused route (no transpose cues, no
    prior, bf16 arithmetic, dk 32 / 64) one Function builds the mixed keys, runs the two-score kernels and hands back ONE gradient per
    projection; otherwise the views go to `crossview_core`."""
    if qkv1.shape[0] == 0:
        return _empty_batch(qkv1, 0, qkv1.shape[1], qkv1.shape[-2] * qkv1.shape[-1])
    prec = _prec_for(qkv1.dtype)
    if (prior_weight <= 0.0 and t1 == 0.0 and t2 == 0.0 and _PATH != L.PATH_GENERIC and prec == L.PREC_BF16
            and qkv1.shape[-1] in (32, 64) and qkv1.dtype != torch.float16):
        drop = (float(dropout_p), (dropout_seed() if seed is None else int(seed))) if dropout_p > 0 else (0.0, 0)
        LAST_PATH["crossview_fwd"] = L.PATH_FUSED
        return _CrossViewFoldedFn.apply(qkv1, qkv2, mix, attn_mask, causal, prec, drop)
    return crossview_core(qkv1[:, :, 0], qkv1[:, :, 1], qkv1[:, :, 2], qkv2[:, :, 0], qkv2[:, :, 1], mix, t1=t1, t2=t2,
                          prior_weight=prior_weight, anchor_mode=anchor_mode, fixed_k_star=fixed_k_star, attn_mask=attn_mask, causal=causal,
                          dropout_p=dropout_p, seed=seed)


def _dp_fwd(ts, lg, gates, beta_not, hops, m8, ms, causal, prec, path, drop):
    """one mopk_dualpath_fwd call on six (B,N,H,dk) views -> (y (B,N,H,dk), saved, resolved path)"""
    lib = L.lib()
    B, N, H, dk = ts[0].shape
    dev = ts[0].device
    a = L.DualPathArgs()
    a.B, a.H, a.N, a.dk, a.hops = B, H, N, dk, hops
    a.io_dtype, a.precision, a.path, a.causal = _io_dtype(ts[0]), prec, path, int(bool(causal))
    a.g_and, a.g_or, a.g_not, a.g_chain = (float(g) for g in gates)
    a.beta_not = float(beta_not)
    a.q1, a.k1, a.v1, a.q2, a.k2, a.v2 = (_v4(t) for t in ts)
    a.mask, (a.mask_sb, a.mask_sh, a.mask_si) = _ptr(m8), ms
    a.chain_logit = lg.data_ptr()
    y = torch.empty(B, N, H, dk, dtype=ts[0].dtype, device=dev)
    a.y = _v4(y)
    if path == L.PATH_AUTO:
        path = L.PATH_FUSED if lib.mopk_dualpath_fused_supported(C.byref(a)) else L.PATH_GENERIC
        a.path = path
    a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
    LAST_PATH["dualpath_fwd"] = path
    saved = _bytes(lib.mopk_dualpath_saved_bytes(C.byref(a)), dev)
    ws = _bytes(lib.mopk_dualpath_workspace_bytes(C.byref(a)), dev)
    a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
    with _timed("dualpath_fwd"):
        rc = lib.mopk_dualpath_fwd(C.byref(a), _stream())
    L.check(rc, "mopk_dualpath_fwd")
    return y, saved, path


def _dp_bwd(ts, lg, y, saved, dy, gs, gates, beta_not, hops, m8, ms, causal, prec, path, drop):
    """one mopk_dualpath_bwd call: gradients into the six (B,N,H,dk) views `gs`; -> d chain_logit partials (B,H)"""
    lib = L.lib()
    B, N, H, dk = ts[0].shape
    dev = ts[0].device
    a = L.DualPathArgs()
    a.B, a.H, a.N, a.dk, a.hops = B, H, N, dk, hops
    a.io_dtype, a.precision, a.path, a.causal = _io_dtype(ts[0]), prec, path, int(bool(causal))
    a.g_and, a.g_or, a.g_not, a.g_chain = (float(g) for g in gates)
    a.beta_not = float(beta_not)
    a.q1, a.k1, a.v1, a.q2, a.k2, a.v2 = (_v4(t) for t in ts)
    a.mask, (a.mask_sb, a.mask_sh, a.mask_si) = _ptr(m8), ms
    a.chain_logit = lg.data_ptr()
    a.y, a.dy = _v4(y), _v4(dy)
    a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
    a.dq1, a.dk1, a.dv1, a.dq2, a.dk2, a.dv2 = (_v4(g) for g in gs)
    dlg = torch.empty(B, H, dtype=torch.float32, device=dev)
    a.dlogit_part = dlg.data_ptr()
    LAST_PATH["dualpath_bwd"] = path
    ws = _bytes(lib.mopk_dualpath_workspace_bytes(C.byref(a)), dev)
    a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
    with _timed("dualpath_bwd"):
        rc = lib.mopk_dualpath_bwd(C.byref(a), _stream())
    L.check(rc, "mopk_dualpath_bwd")
    return dlg


class _DualPathFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q1, k1, v1, q2, k2, v2, logit, gates, beta_not, hops, mask, causal, prec, path, drop=(0.0, 0)):
        _require_gpu(q1, "MultiHopMSA")
        ctx.packed = k1 is None            # packed form: q1 / q2 are the (B,N,3,H,dk) outputs of the two qkv projections (see _SdpaFn)
        if ctx.packed:
            q1, k1, v1 = q1.contiguous().unbind(2)
            q2, k2, v2 = q2.contiguous().unbind(2)
        ts = [_heads_view(t) for t in (q1, k1, v1, q2, k2, v2)]
        B, N, H, dk = ts[0].shape
        lg = _f32c(logit).reshape(1)
        m8, ms = _mask_u8(mask, B, H, N, ts[0].device)
        y, saved, path = _dp_fwd(ts, lg, gates, beta_not, hops, m8, ms, causal, prec, path, drop)
        ctx.save_for_backward(*ts, lg, y, saved)
        ctx.meta = (gates, beta_not, hops, causal, prec, path, m8, ms, drop)
        return y.view(B, N, H * dk)

    @staticmethod
    def backward(ctx, dy):
        *ts, lg, y, saved = ctx.saved_tensors
        gates, beta_not, hops, causal, prec, path, m8, ms, drop = ctx.meta
        B, N, H, dk = ts[0].shape
        dev = ts[0].device
        dy = dy.contiguous().to(ts[0].dtype).view(B, N, H, dk)
        mk = torch.zeros if hops == 0 else torch.empty          # hops == 0: dv2 is never written
        if ctx.packed:
            packs = [mk(B, N, 3, H, dk, dtype=ts[0].dtype, device=dev) for _ in range(2)]
            gs = [*packs[0].unbind(2), *packs[1].unbind(2)]
        else:
            gs = [mk(B, N, H, dk, dtype=ts[0].dtype, device=dev) for _ in range(6)]
        dlg = _dp_bwd(ts, lg, y, saved, dy, gs, gates, beta_not, hops, m8, ms, causal, prec, path, drop)
        if ctx.packed:
            return (packs[0], None, None, packs[1], None, None, dlg.sum().reshape(()), None, None, None, None, None, None, None, None)
        return (*gs, dlg.sum().reshape(()), None, None, None, None, None, None, None, None)


class _CrossViewFoldedFn(torch.autograd.Function):
    """CrossViewMixerMSA without transpose cues / prior on the fused two-score kernels, from the PACKED projections:
    S = q1 (m11 k1 + m12 k2)^T + q2 (m21 k1 + m22 k2)^T (reference :99-105): the 2x2 mix folds into two mixed key tensors built here
    (no autograd graph), the core is the dual-path kernel pair without the transport term (hops = 0, v2 unused), and the backward
    un-mixes dk1' / dk2' into dk1, dk2 and d mix and writes ONE gradient per projection."""

    @staticmethod
    def forward(ctx, qkv1, qkv2, mix, mask, causal, prec, drop):
        _require_gpu(qkv1, "CrossViewMixerMSA")
        q1, k1, v1 = qkv1.contiguous().unbind(2)
        q2, k2, _ = qkv2.contiguous().unbind(2)
        B, N, H, dk = q1.shape
        m = mix.detach().to(k1.dtype)
        k1p = (m[0, 0] * k1 + m[0, 1] * k2).contiguous()
        k2p = (m[1, 0] * k1 + m[1, 1] * k2).contiguous()
        ts = [q1, k1p, v1, q2, k2p, v1]
        lg = torch.zeros(1, dtype=torch.float32, device=q1.device)
        m8, ms = _mask_u8(mask, B, H, N, q1.device)
        y, saved, path = _dp_fwd(ts, lg, (1.0, 0.0, 0.0, 0.0), 0.0, 0, m8, ms, causal, prec, L.PATH_FUSED, drop)
        ctx.save_for_backward(qkv1, qkv2, mix, k1p, k2p, lg, y, saved)
        ctx.meta = (causal, prec, path, m8, ms, drop)
        return y.view(B, N, H * dk)

    @staticmethod
    def backward(ctx, dy):
        qkv1, qkv2, mix, k1p, k2p, lg, y, saved = ctx.saved_tensors
        causal, prec, path, m8, ms, drop = ctx.meta
        q1, k1, v1 = qkv1.unbind(2)
        q2, k2, _ = qkv2.unbind(2)
        B, N, H, dk = q1.shape
        dev = q1.device
        dy = dy.contiguous().to(q1.dtype).view(B, N, H, dk)
        d1, d2 = torch.empty_like(qkv1), torch.zeros_like(qkv2)            # v2 receives no gradient (:98)
        dk1p, dk2p = torch.empty(B, N, H, dk, dtype=q1.dtype, device=dev), torch.empty(B, N, H, dk, dtype=q1.dtype, device=dev)
        dv2 = torch.zeros(B, N, H, dk, dtype=q1.dtype, device=dev)          # hops == 0: never written
        gs = [d1[:, :, 0], dk1p, d1[:, :, 2], d2[:, :, 0], dk2p, dv2]
        _dp_bwd([q1, k1p, v1, q2, k2p, v1], lg, y, saved, dy, gs, (1.0, 0.0, 0.0, 0.0), 0.0, 0, m8, ms, causal, prec, path, drop)
        m = mix.detach().to(q1.dtype)
        d1k, d2k = d1[:, :, 1], d2[:, :, 1]                                  # k1' = m11 k1 + m12 k2, k2' = m21 k1 + m22 k2
        torch.mul(dk1p, m[0, 0], out=d1k); d1k.addcmul_(dk2p, m[1, 0])
        torch.mul(dk1p, m[0, 1], out=d2k); d2k.addcmul_(dk2p, m[1, 1])
        f = lambda x, y_: (x.float() * y_.float()).sum()
        dmix = torch.stack([torch.stack([f(dk1p, k1), f(dk1p, k2)]), torch.stack([f(dk2p, k1), f(dk2p, k2)])]).to(mix.dtype)
        return d1, d2, dmix, None, None, None, None


@_half_via_fp32
def dualpath_core(q1, k1, v1, q2, k2, v2, chain_logit, g_and, g_or, g_not, g_chain, beta_not, hops,
                  attn_mask=None, causal=False, dropout_p: float = 0.0, seed: Optional[int] = None):
    """q*, k*, v*: (B,N,H,dk) views -- or packed: q1 / q2 = the (B,N,3,H,dk) outputs of the two qkv projections, k1 = v1 = k2 = v2 = None"""
    if q1.shape[0] == 0:
        return _empty_batch(q1, 0, q1.shape[1], q1.shape[-2] * q1.shape[-1])
    drop = (float(dropout_p), (dropout_seed() if seed is None else int(seed))) if dropout_p > 0 else (0.0, 0)
    return _DualPathFn.apply(q1, k1, v1, q2, k2, v2, chain_logit, (g_and, g_or, g_not, g_chain), beta_not,
                             int(hops), attn_mask, causal, _prec_for(q1.dtype), _PATH, drop)


class _QuartetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, q2, k2, mixture, qscale, add_mask, eps, use_quartet, need_weights, prec, path, drop=(0.0, 0)):
        _require_gpu(q, "CausalSelfAttention")
        lib = L.lib()
        ts = [_heads_view(t) for t in ((q, k, v, q2, k2) if use_quartet else (q, k, v))]
        B, T, H, dh = ts[0].shape
        dev = ts[0].device
        a = L.QuartetArgs()
        a.B, a.H, a.T, a.dh = B, H, T, dh
        a.io_dtype, a.precision, a.path = _io_dtype(ts[0]), prec, path
        a.use_quartet, a.eps = int(bool(use_quartet)), float(eps)
        a.q, a.k, a.v = _v4(ts[0]), _v4(ts[1]), _v4(ts[2])
        sc = []
        if use_quartet:
            a.q2, a.k2 = _v4(ts[3]), _v4(ts[4])
            sc = [_f32c(mixture).reshape(1), _f32c(qscale).reshape(1)]
            a.mixture, a.quartet_scale = sc[0].data_ptr(), sc[1].data_ptr()
        am, ams = _bias_f32(add_mask, B, H, T, dev)
        a.add_mask, (a.am_sb, a.am_sh, a.am_si) = _ptr(am), ams
        y = torch.empty(B, T, H, dh, dtype=ts[0].dtype, device=dev)
        a.y = _v4(y)
        attn = torch.empty(B, H, T, T, dtype=torch.float32, device=dev) if need_weights else None
        a.attn = _ptr(attn)
        a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
        if path == L.PATH_AUTO:
            path = L.PATH_FUSED if lib.mopk_quartet_fused_supported(C.byref(a)) else L.PATH_GENERIC
            a.path = path
        LAST_PATH["quartet_fwd"] = path
        saved = _bytes(lib.mopk_quartet_saved_bytes(C.byref(a)), dev)
        ws = _bytes(lib.mopk_quartet_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("quartet_fwd"):
            rc = lib.mopk_quartet_fwd(C.byref(a), _stream())
        L.check(rc, "mopk_quartet_fwd")
        ctx.save_for_backward(*ts, *sc, y, saved)
        ctx.meta = (eps, use_quartet, prec, path, am, ams, drop)
        out = y.view(B, T, H * dh)
        if need_weights:
            ctx.mark_non_differentiable(attn)
            return out, attn
        return out

    @staticmethod
    def backward(ctx, dy, *unused):
        lib = L.lib()
        eps, use_quartet, prec, path, am, ams, drop = ctx.meta
        if use_quartet:
            q, k, v, q2, k2, mix, qs, y, saved = ctx.saved_tensors
        else:
            q, k, v, y, saved = ctx.saved_tensors
        B, T, H, dh = q.shape
        dev = q.device
        dy = dy.contiguous().to(q.dtype).view(B, T, H, dh)
        a = L.QuartetArgs()
        a.B, a.H, a.T, a.dh = B, H, T, dh
        a.io_dtype, a.precision, a.path = _io_dtype(q), prec, path
        a.use_quartet, a.eps = int(bool(use_quartet)), float(eps)
        a.q, a.k, a.v, a.y, a.dy = _v4(q), _v4(k), _v4(v), _v4(y), _v4(dy)
        a.add_mask, (a.am_sb, a.am_sh, a.am_si) = _ptr(am), ams
        a.dropout_p, a.dropout_seed = float(drop[0]), int(drop[1])
        n = 5 if use_quartet else 3
        gs = [torch.empty(B, T, H, dh, dtype=q.dtype, device=dev) for _ in range(n)]
        a.dq, a.dk_, a.dv = _v4(gs[0]), _v4(gs[1]), _v4(gs[2])
        dmix = dqs = None
        if use_quartet:
            a.q2, a.k2, a.dq2, a.dk2 = _v4(q2), _v4(k2), _v4(gs[3]), _v4(gs[4])
            a.mixture, a.quartet_scale = mix.data_ptr(), qs.data_ptr()
            dmix = torch.empty(B, H, dtype=torch.float32, device=dev)
            dqs = torch.empty(B, H, dtype=torch.float32, device=dev)
            a.dmixture_part, a.dqscale_part = dmix.data_ptr(), dqs.data_ptr()
        ws = _bytes(lib.mopk_quartet_workspace_bytes(C.byref(a)), dev)
        a.saved, a.workspace = saved.data_ptr(), ws.data_ptr()
        with _timed("quartet_bwd"):
            rc = lib.mopk_quartet_bwd(C.byref(a), _stream())
        L.check(rc, "mopk_quartet_bwd")
        if use_quartet:
            return (gs[0], gs[1], gs[2], gs[3], gs[4], dmix.sum().reshape(1), dqs.sum().reshape(1),
                    None, None, None, None, None, None, None)
        return (gs[0], gs[1], gs[2], None, None, None, None, None, None, None, None, None, None, None)


@_half_via_fp32
def quartet_core(q, k, v, q2, k2, mixture, quartet_scale, add_mask, eps, use_quartet, need_weights=False,
                 dropout_p: float = 0.0, seed: Optional[int] = None):
    if q.shape[0] == 0:                   # q: (B,T,H,dk)
        out = _empty_batch(q, 0, q.shape[1], q.shape[2] * q.shape[3])
        return (out, q.new_zeros((0, q.shape[2], q.shape[1], q.shape[1]), dtype=torch.float32)) if need_weights else out
    drop = (float(dropout_p), (dropout_seed() if seed is None else int(seed))) if dropout_p > 0 else (0.0, 0)
    return _QuartetFn.apply(q, k, v, q2, k2, mixture, quartet_scale, add_mask, eps, use_quartet, need_weights,
                            _prec_for(q.dtype), _PATH, drop)


# ---- LayerNorm prologue / residual epilogue around the cores (mopk_layernorm_*) ----
def layernorm_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """shapes the HIP prologue covers (anything else: the caller keeps torch's LayerNorm)"""
    d = x.shape[-1]
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and weight.dtype in (torch.float32, torch.bfloat16)
            and d % 8 == 0 and d <= 4096 and x.numel() > 0)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype, with_residual):
        _require_gpu(x, "layernorm")
        xc = x.contiguous()
        d = xc.shape[-1]
        rows = xc.numel() // d
        y = torch.empty(xc.shape, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        w = weight.detach().contiguous()
        b = None if bias is None else bias.detach().to(w.dtype).contiguous()
        a = L.LayerNormArgs(rows=rows, dim=d, x_dtype=_io_dtype(xc), y_dtype=_io_dtype(y), p_dtype=_io_dtype(w), eps=float(eps),
                            x_ld=d, y_ld=d, x=_ptr(xc), gamma=_ptr(w), beta=_ptr(b), y=_ptr(y), mean=_ptr(mean), rstd=_ptr(rstd))
        with _timed("layernorm_fwd"):
            L.check(L.lib().mopk_layernorm_fwd(C.byref(a), _stream()), "mopk_layernorm_fwd")
        ctx.save_for_backward(xc, w, mean, rstd)
        ctx.meta = (float(eps), bias is not None, with_residual, weight.dtype, None if bias is None else bias.dtype)
        if with_residual:
            return xc.view_as(xc), y
        return y

    @staticmethod
    def backward(ctx, *grads):
        xc, w, mean, rstd = ctx.saved_tensors
        eps, has_bias, with_residual, wdt, bdt = ctx.meta
        dres, dy = (grads if with_residual else (None, grads[0]))
        d = xc.shape[-1]
        rows = xc.numel() // d
        if dy is None:                                    # only the residual branch carries a gradient
            return (dres, None, None, None, None, None)
        dy = dy.contiguous()
        if dres is not None:
            dres = dres.to(xc.dtype).contiguous()
        dx = torch.empty_like(xc)
        dg = torch.empty(d, dtype=torch.float32, device=xc.device)
        db = torch.empty(d, dtype=torch.float32, device=xc.device) if has_bias else None
        a = L.LayerNormArgs(rows=rows, dim=d, x_dtype=_io_dtype(xc), y_dtype=_io_dtype(dy), p_dtype=_io_dtype(w), eps=eps,
                            x_ld=d, y_ld=d, x=_ptr(xc), gamma=_ptr(w), mean=_ptr(mean), rstd=_ptr(rstd),
                            dy=_ptr(dy), dres=_ptr(dres), dx=_ptr(dx), dgamma=_ptr(dg), dbeta=_ptr(db))
        ws = torch.empty(max(1, L.lib().mopk_layernorm_workspace_bytes(C.byref(a))), dtype=torch.uint8, device=xc.device)
        a.workspace = _ptr(ws)
        with _timed("layernorm_bwd"):
            L.check(L.lib().mopk_layernorm_bwd(C.byref(a), _stream()), "mopk_layernorm_bwd")
        return dx, dg.to(wdt), (db.to(bdt) if has_bias else None), None, None, None


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float = 1e-5,
              out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``F.layer_norm(x, (dim,), weight, bias, eps)`` written in ``out_dtype`` (default: x.dtype) by one HIP pass."""
    return _LayerNormFn.apply(x, weight, bias, eps, out_dtype or x.dtype, False)


def layernorm_residual(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float = 1e-5,
                       out_dtype: Optional[torch.dtype] = None):
    """(x_res, ln(x)) for a pre-norm residual branch ``x_res + f(ln(x))``: ``x_res`` is ``x`` itself, routed through the same
    autograd node so that the backward returns ``d x_res + LN'(d ln)`` from ONE kernel instead of LN backward + an add."""
    return _LayerNormFn.apply(x, weight, bias, eps, out_dtype or x.dtype, True)
