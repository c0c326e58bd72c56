"""Drop-in attention modules backed by libmopk (gfx950 HIP kernels).

Same constructor signatures, forward signatures, parameter names/shapes and
initialisation as reference `mop/models/attention_variants.py`, so reference
checkpoints load with `load_state_dict` and call sites need no change:

    BaselineMSA      <- attention_variants.py:23-48
    MultiHopMSA      <- :163-231
    EdgewiseGateHead <- :234-331   (parameter container; its arithmetic runs inside the kernels)
    EdgewiseMSA      <- :334-564
    UnifiedMSA       <- :567-629

Only the Linear projections run in PyTorch (hipBLASLt); everything between the
qkv projection and the output projection is one C-ABI call (mop_amd/ops.py).
The plain dense gate head runs on the fused kernels like the low-rank head; `use_k3` and the S lens bank run inside the library's
generic path; the Q/K lens bank's depthwise token convolutions (:472-498) are torch ops feeding per-view q/k to the same core.
An `attn_mask` on EdgewiseMSA is an extension (the reference is NaN there, SURVEY.md 8a note): the mask acts on the probabilities
only, generic path.  Attention dropout (`attn_drop > 0` in training mode) is a counter-based keep mask that both paths of every
core evaluate from one seed: inside the fused kernels, and on the generic paths as a multiply of the N x N map ("Dropout").
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .linear import TokenLinear, residual_linear

# gate order inside the low-rank head: 0=and 1=or 2=not 3=chain   (:284)
_LOWRANK_PRESET = {
    "and": (0,), "or": (1,), "not": (2,), "chain": (3,),
    "nor": (2,), "xor": (1,),          # :292-297
    "mix5": (0, 1, 2),                 # :302-308
}
_DENSE_PRESET = {"and": 0, "or": 1, "not": 2, "nor": 2, "xor": 1, "chain": 3}   # :259-272


_CAUSAL_CACHE: Dict[int, tuple] = {}  # id(mask) -> (weakref to the mask tensor, version, is_causal)


def _is_causal_mask(mask: Optional[torch.Tensor], n: int) -> bool:
    """True iff `mask` (0 = blocked) is exactly the lower-triangular causal mask shared by every batch and head.

    The comparison synchronises, so its result is cached per mask TENSOR OBJECT and version: an entry holds a weak reference to the
    tensor and counts only while that very object is alive (never keyed by the data address -- the caching allocator hands a freed
    mask's address to the next mask of the same shape), and it is re-validated whenever the tensor was written in place."""
    import weakref
    if mask is None or mask.shape[-2:] != (n, n) or any(d != 1 for d in mask.shape[:-2]):
        return False
    hit = _CAUSAL_CACHE.get(id(mask))
    if hit is None or hit[0]() is not mask or hit[1] != mask._version:
        tri = torch.ones(n, n, dtype=torch.bool, device=mask.device).tril_()
        verdict = bool(torch.equal(mask.reshape(n, n) != 0, tri))
        if len(_CAUSAL_CACHE) > 64:
            for k in [k for k, v in _CAUSAL_CACHE.items() if v[0]() is None]:
                del _CAUSAL_CACHE[k]
            if len(_CAUSAL_CACHE) > 64:
                _CAUSAL_CACHE.clear()
        hit = (weakref.ref(mask), mask._version, verdict)
        _CAUSAL_CACHE[id(mask)] = hit
    return hit[2]


class EdgewiseGateHead(nn.Module):
    """Parameters of the per-edge gate head (reference :234-309).

    low-rank mode: row_proj / col_proj Conv1d(in_ch, 4*rank, 1) consumed by the
    kernels as (4r, C) matrices.  dense mode: conv1 / [mid3] / conv2 with the
    reference's names and presets, consumed as (16,C) / (16,16,3,3) / (4,16).
    """

    def __init__(self, in_ch: int, hidden: int = 16, use_k3: bool = False, gate_mode: str = "dense",
                 gate_rank: int = 4, gate_init: str = "neutral"):
        super().__init__()
        self.use_k3 = bool(use_k3)
        self.gate_mode = str(gate_mode)
        self.gate_rank = int(gate_rank)
        self.gate_init = str(gate_init)
        if self.gate_mode == "dense":
            self.conv1 = nn.Conv2d(in_ch, hidden, kernel_size=1, bias=True)
            self.act = nn.GELU(approximate="tanh")
            if self.use_k3:
                self.mid3 = nn.Conv2d(hidden, hidden, kernel_size=3, padding=1, bias=True)
            self.conv2 = nn.Conv2d(hidden, 4, kernel_size=1, bias=True)
            with torch.no_grad():
                self.conv2.bias.fill_(-5.0)
                if self.gate_init in _DENSE_PRESET:
                    self.conv2.bias[_DENSE_PRESET[self.gate_init]] = 2.0
        else:
            r = self.gate_rank
            self.row_proj = nn.Conv1d(in_ch, 4 * r, kernel_size=1, bias=True)
            self.col_proj = nn.Conv1d(in_ch, 4 * r, kernel_size=1, bias=True)
            c = math.sqrt(2.0 / max(1, r))
            with torch.no_grad():
                self.row_proj.bias.zero_()
                self.col_proj.bias.zero_()
                for g in _LOWRANK_PRESET.get(self.gate_init, ()):
                    self.row_proj.bias[g * r:(g + 1) * r] = c
                    self.col_proj.bias[g * r:(g + 1) * r] = c

    def forward(self, feat: torch.Tensor) -> torch.Tensor:  # pragma: no cover - never materialised
        raise NotImplementedError(
            "EdgewiseGateHead is evaluated inside the fused kernels from row/col means; the "
            "(BH,C,N,N) feature stack of the reference is never built")


class EdgewiseMSA(nn.Module):
    def __init__(self, dim: int, heads: int = 4, attn_drop: float = 0.0, proj_drop: float = 0.0,
                 beta_not: float = 0.5, use_k3: bool = False, n_views: int = 2, share_qkv: bool = False,
                 gate_mode: str = "dense", gate_rank: int = 4, gate_init: str = "neutral",
                 use_lens_bank: bool = False, lens_kernel_size: int = 3,
                 lens_dilations: Optional[Tuple[int, ...]] = None, use_lens_bank_qk: bool = False,
                 lens_qk_kernel_size: int = 3, lens_qk_dilations: Optional[Tuple[int, ...]] = None,
                 lens_qk_causal: bool = False):
        super().__init__()
        assert dim % heads == 0
        self.h, self.dk = heads, dim // heads
        self.beta_not = beta_not
        self.n_views = max(2, int(n_views))
        self.share_qkv = bool(share_qkv)
        self.use_lens_bank = bool(use_lens_bank)
        self.lens_kernel_size = int(lens_kernel_size)
        self.lens_dilations = tuple(lens_dilations) if lens_dilations is not None else (1, 2)
        self.use_lens_bank_qk = bool(use_lens_bank_qk)
        self.lens_qk_kernel_size = int(lens_qk_kernel_size)
        self.lens_qk_dilations = tuple(lens_qk_dilations) if lens_qk_dilations is not None else (1, 2)
        self.lens_qk_causal = bool(lens_qk_causal)
        if self.use_lens_bank_qk and not self.share_qkv:
            raise ValueError("use_lens_bank_qk=True requires share_qkv=True for now")
        V, H, dk = self.n_views, self.h, self.dk
        if self.share_qkv:
            self.qkv = TokenLinear(dim, 3 * dim, bias=False)
            self.q_scale = nn.Parameter(torch.ones(V, H, 1, dk))
            self.k_scale = nn.Parameter(torch.ones(V, H, 1, dk))
            self.v_scale = nn.Parameter(torch.ones(V, H, 1, dk))
        else:
            self.qkv_list = nn.ModuleList(TokenLinear(dim, 3 * dim, bias=False) for _ in range(V))
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = TokenLinear(dim, dim, bias=False)
        self.proj_drop = nn.Dropout(proj_drop)
        num_s = len(self.lens_qk_dilations) if self.use_lens_bank_qk else V
        in_ch = 2 * num_s + 2
        if self.use_lens_bank_qk:
            k = self.lens_qk_kernel_size

            def dw(d):
                pad = 0 if self.lens_qk_causal else d * (k - 1) // 2
                return nn.Conv1d(dk, dk, k, padding=pad, dilation=d, groups=dk, bias=False)

            self.q_lens = nn.ModuleList(dw(d) for d in self.lens_qk_dilations)
            self.k_lens = nn.ModuleList(dw(d) for d in self.lens_qk_dilations)
            self._lens_qk_num = num_s
        if self.use_lens_bank:
            self.lens_bank = nn.ModuleList(
                nn.Conv2d(num_s, num_s, self.lens_kernel_size, padding=d, dilation=d, groups=num_s, bias=False)
                for d in self.lens_dilations)
            in_ch += num_s * len(self.lens_dilations)
        self.edge_head = EdgewiseGateHead(in_ch=in_ch, hidden=16, use_k3=use_k3, gate_mode=gate_mode,
                                          gate_rank=gate_rank, gate_init=gate_init)
        self.chain_value_logit = nn.Parameter(torch.tensor(-2.0))

    def _check_supported(self, attn_mask):
        # attn_mask: an EXTENSION.  The reference fills the scores with -inf before they enter the feature stack (:504-506 feed :518-546),
        # so its output is NaN for any blocking mask (SURVEY.md 8a note).  Here the mask (0 = blocked) acts on the attention
        # probabilities only -- the per-view softmaxes and the final one -- while the gate features see the unmasked scores; it runs on
        # the generic path, and every query must keep at least one key (a causal mask does).
        if self.use_lens_bank and self.lens_kernel_size != 3:
            raise ValueError("lens_kernel_size must be 3: with padding = dilation any other size changes the plane size and "
                             "the reference's feature stack (:534) cannot be built")

    def _qk_lens_views(self, qkv: torch.Tensor) -> torch.Tensor:
        """Q/K lens bank (:472-498): depthwise dilated convolutions over the token axis of view-0 q and k build one
        (q, k) pair per dilation.  qkv: (B,N,3,H,dk) -> packed (B,N,L,3,H,dk) for the core (v of every slot = raw v)."""
        B, N, _, H, dk = qkv.shape
        k_sz = self.lens_qk_kernel_size
        q_base = qkv[:, :, 0].permute(0, 2, 1, 3) * self.q_scale[0].to(qkv.dtype)      # (B,H,N,dk)   :462
        k_base = qkv[:, :, 1].permute(0, 2, 1, 3) * self.k_scale[0].to(qkv.dtype)
        q_flat = q_base.reshape(B * H, dk, N)              # the reference reshapes (not transposes) to (B*H, dk, N)   :477-478
        k_flat = k_base.reshape(B * H, dk, N)
        slots = []
        for i, (qc, kc) in enumerate(zip(self.q_lens, self.k_lens)):
            q_in, k_in = q_flat, k_flat
            if self.lens_qk_causal:
                left = (k_sz - 1) * self.lens_qk_dilations[i]                            # :484-486
                q_in, k_in = F.pad(q_flat, (left, 0)), F.pad(k_flat, (left, 0))
            q_l = qc(q_in).view(B, H, dk, N).permute(0, 3, 1, 2)                         # -> (B,N,H,dk)   :491-492
            k_l = kc(k_in).view(B, H, dk, N).permute(0, 3, 1, 2)
            slots.append(torch.stack([q_l, k_l, qkv[:, :, 2]], dim=2))                   # (B,N,3,H,dk)
        return torch.stack(slots, dim=2)

    def _project(self, y: torch.Tensor, residual: Optional[torch.Tensor]) -> torch.Tensor:
        """proj (+ proj_drop) of the core's output; with `residual` the block's `x + ...` add rides in the proj GEMM's epilogue"""
        if residual is None:
            return self.proj_drop(self.proj(y))
        if y.is_cuda and y.dtype == residual.dtype == self.proj.weight.dtype and not torch.is_autocast_enabled() \
                and not (self.training and self.proj_drop.p > 0):
            return residual_linear(residual, y, self.proj.weight)
        return residual + self.proj_drop(self.proj(y))

    def forward(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`residual` (not in the reference signature, default None = reference behaviour): returns `residual + attn(x)` with
        the add fused into the output projection -- what `BlockEdgewise` wraps around this module (:371-374)."""
        self._check_supported(attn_mask)
        B, N, D = x.shape
        H, dk, V = self.h, self.dk, self.n_views
        inv = 1.0 / math.sqrt(dk)
        eh = self.edge_head
        dense = eh.gate_mode == "dense"
        n_s = V                                            # number of score views the core sees
        if self.share_qkv and self.use_lens_bank_qk:
            qkv = self._qk_lens_views(self.qkv(x).view(B, N, 3, H, dk))
            n_s = len(self.lens_qk_dilations)
            sqk = torch.full((n_s, H, dk), inv, device=x.device, dtype=torch.float32)
            vs0, vsL = self.v_scale[0, :, 0], self.v_scale[min(V - 1, n_s - 1), :, 0]     # :556
        elif self.share_qkv:
            qkv = self.qkv(x).view(B, N, 1, 3, H, dk)
            sqk = (self.q_scale * self.k_scale).squeeze(2) * inv          # (V,H,dk)
            vs0, vsL = self.v_scale[0, :, 0], self.v_scale[V - 1, :, 0]   # (H,dk)
        else:
            w = torch.cat([lin.weight for lin in self.qkv_list], dim=0)   # one GEMM for all views
            qkv = F.linear(x, w).view(B, N, V, 3, H, dk)
            sqk = torch.full((V, H, dk), inv, device=x.device, dtype=torch.float32)
            vs0 = vsL = torch.ones(H, dk, device=x.device, dtype=torch.float32)
        # low-rank head + S lens bank: the head reads row / column means only, and those of the lens planes have a closed form in q, k
        # (ops.lens_mean_features) -- the fused kernels take them as extra feature channels, no N x N plane is ever convolved
        lens_fused = (not dense and self.use_lens_bank and attn_mask is None and qkv.shape[2] == 1 and
                      ops.lowrank_lens_fused_supported(qkv, n_s, eh.row_proj.weight.shape[0] // 4, self.lens_dilations))
        if not dense and attn_mask is None and (lens_fused or not self.use_lens_bank):
            lens_w = torch.stack([c.weight[:, 0] for c in self.lens_bank]) if lens_fused else None           # (L,V,3,3)
            y = ops.edgewise_lowrank_core(qkv, sqk, vs0, vsL, eh.row_proj.weight.squeeze(-1), eh.row_proj.bias,
                                          eh.col_proj.weight.squeeze(-1), eh.col_proj.bias,
                                          self.chain_value_logit, float(self.beta_not), n_s,
                                          dropout_p=float(self.attn_drop.p) if self.training else 0.0,       # :552
                                          lens_w=lens_w, lens_dilations=self.lens_dilations if lens_fused else ())
            return self._project(y, residual)
        # dense gate head and / or S lens bank: the library's generic path (MopkEdgewiseExt)
        lens_w = torch.stack([c.weight[:, 0] for c in self.lens_bank]) if self.use_lens_bank else None   # (L,S,3,3)
        var = ops.EdgewiseVariant(dense=dense, use_k3=dense and eh.use_k3,
                                  lens_dilations=self.lens_dilations if self.use_lens_bank else ())
        if dense:
            head = (eh.conv1.weight.flatten(1), eh.conv1.bias, eh.conv2.weight.flatten(1), eh.conv2.bias)
            W3, b3 = (eh.mid3.weight, eh.mid3.bias) if eh.use_k3 else (None, None)
        else:
            head = (eh.row_proj.weight.squeeze(-1), eh.row_proj.bias, eh.col_proj.weight.squeeze(-1), eh.col_proj.bias)
            W3 = b3 = None
        y = ops.edgewise_general_core(qkv, sqk, vs0, vsL, self.chain_value_logit, head, float(self.beta_not), n_s, var,
                                      W3=W3, b3=b3, lens_w=lens_w, dropout_p=float(self.attn_drop.p) if self.training else 0.0,
                                      attn_mask=attn_mask)
        return self._project(y, residual)


class BaselineMSA(nn.Module):
    def __init__(self, dim: int, heads: int = 4, attn_drop: float = 0.0, proj_drop: float = 0.0):
        super().__init__()
        assert dim % heads == 0
        self.h, self.dk = heads, dim // heads
        self.qkv = TokenLinear(dim, 3 * dim, bias=False)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = TokenLinear(dim, dim, bias=False)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        pdrop = float(self.attn_drop.p) if self.training else 0.0      # reference :45: dropout on the attention weights, inside the kernels
        B, N, D = x.shape
        qkv = self.qkv(x).view(B, N, 3, self.h, self.dk)
        causal = _is_causal_mask(attn_mask, N)      # tril mask -> in-kernel causal flag (keeps the call on the fused kernels)
        y = ops.sdpa_core(qkv, attn_mask=None if causal else attn_mask, causal=causal, dropout_p=pdrop)     # packed q | k | v: one gradient tensor back
        return self.proj_drop(self.proj(y))


class MultiHopMSA(nn.Module):
    """Dual-path logits with scalar gates; default gates give softmax(S1+S2) (:209-221)."""

    def __init__(self, dim: int, heads: int = 4, attn_drop: float = 0.0, proj_drop: float = 0.0,
                 beta_not: float = 0.5, gates: Optional[Dict[str, float]] = None, hops: int = 3):
        super().__init__()
        assert dim % heads == 0
        assert hops >= 2
        self.h, self.dk, self.hops = heads, dim // heads, int(hops)
        self.qkv1 = TokenLinear(dim, 3 * dim, bias=False)
        self.qkv2 = TokenLinear(dim, 3 * dim, bias=False)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = TokenLinear(dim, dim, bias=False)
        self.proj_drop = nn.Dropout(proj_drop)
        self.beta_not = float(beta_not)
        self.gates = gates or dict(and_=1.0, or_=0.0, not_=0.0, chain=0.0, base=1.0)
        self.chain_value_logit = nn.Parameter(torch.tensor(-2.0))

    def forward(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        pdrop = float(self.attn_drop.p) if self.training else 0.0      # reference :222: dropout on the mixed attention weights
        B, N, D = x.shape
        qkv1 = self.qkv1(x).view(B, N, 3, self.h, self.dk)
        qkv2 = self.qkv2(x).view(B, N, 3, self.h, self.dk)
        g = self.gates
        causal = _is_causal_mask(attn_mask, N)      # a lower-triangular mask becomes the in-kernel causal flag (fused path)
        y = ops.dualpath_core(qkv1, None, None, qkv2, None, None,      # packed q | k | v of each projection: two gradient tensors back
                              self.chain_value_logit, g.get("and_", 1.0), g.get("or_", 0.0),
                              g.get("not_", 0.0), g.get("chain", 0.0), self.beta_not, self.hops,
                              None if causal else attn_mask, causal=causal, dropout_p=pdrop)
        return self.proj_drop(self.proj(y))


class CrossViewMixerMSA(nn.Module):
    """Cross-view binding: four score maps mixed 2x2, transpose cues, optional per-key prior (reference :51-156)."""

    def __init__(self, dim: int, heads: int = 4, attn_drop: float = 0.0, proj_drop: float = 0.0,
                 use_transpose_cues: bool = True, t1: float = 0.0, t2: float = 0.0,
                 enable_per_key_prior: bool = False, prior_weight: float = 0.5,
                 anchor_mode: str = "argmax_row_sum", fixed_k_star: int = 0):
        super().__init__()
        assert dim % heads == 0
        self.h, self.dk = heads, dim // heads
        self.qkv1 = TokenLinear(dim, 3 * dim, bias=False)
        self.qkv2 = TokenLinear(dim, 3 * dim, bias=False)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = TokenLinear(dim, dim, bias=False)
        self.proj_drop = nn.Dropout(proj_drop)
        self.mix = nn.Parameter(torch.eye(2))
        self.use_transpose_cues, self.t1, self.t2 = bool(use_transpose_cues), float(t1), float(t2)
        self.enable_per_key_prior, self.prior_weight = bool(enable_per_key_prior), float(prior_weight)
        self.anchor_mode, self.fixed_k_star = str(anchor_mode), int(fixed_k_star)

    def forward(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        pdrop = float(self.attn_drop.p) if self.training else 0.0      # reference :153, carried by the fused path only
        B, N, D = x.shape
        a = self.qkv1(x).view(B, N, 3, self.h, self.dk)
        b = self.qkv2(x).view(B, N, 3, self.h, self.dk)           # v2 is unused by the reference too (:98)
        cues = self.use_transpose_cues
        pw = self.prior_weight if (self.enable_per_key_prior and self.prior_weight > 0.0) else 0.0     # :126
        causal = _is_causal_mask(attn_mask, N)
        y = ops.crossview_core_packed(a, b, self.mix,
                               t1=self.t1 if cues else 0.0, t2=self.t2 if cues else 0.0, prior_weight=pw,
                               anchor_mode=self.anchor_mode, fixed_k_star=self.fixed_k_star,
                               attn_mask=None if causal else attn_mask, causal=causal, dropout_p=pdrop)
        return self.proj_drop(self.proj(y))


class UnifiedMSA(nn.Module):
    """mode 'A'/'B' -> BaselineMSA, 'C' -> CrossViewMixerMSA, 'D' -> MultiHopMSA, 'E' -> EdgewiseMSA.
    Like the reference (:609-622), mode E does not forward the lens kwargs."""

    def __init__(self, mode: str, dim: int, heads: int = 4, **kwargs):
        super().__init__()
        mode = str(mode).upper()
        self.mode = mode
        kw = kwargs.get
        if mode in ("A", "B"):
            self.impl = BaselineMSA(dim, heads, kw("attn_drop", 0.0), kw("proj_drop", 0.0))
        elif mode == "C":
            self.impl = CrossViewMixerMSA(
                dim, heads, kw("attn_drop", 0.0), kw("proj_drop", 0.0),
                use_transpose_cues=kw("use_transpose_cues", True), t1=kw("t1", 0.0), t2=kw("t2", 0.0),
                enable_per_key_prior=kw("enable_per_key_prior", False), prior_weight=kw("prior_weight", 0.5),
                anchor_mode=kw("anchor_mode", "argmax_row_sum"), fixed_k_star=kw("fixed_k_star", 0))
        elif mode == "D":
            self.impl = MultiHopMSA(dim, heads, kw("attn_drop", 0.0), kw("proj_drop", 0.0),
                                    beta_not=kw("beta_not", 0.5), gates=kw("gates", None), hops=kw("hops", 3))
        elif mode == "E":
            self.impl = EdgewiseMSA(dim, heads, kw("attn_drop", 0.0), kw("proj_drop", 0.0),
                                    beta_not=kw("beta_not", 0.5), use_k3=kw("use_k3", False),
                                    n_views=kw("n_views", 2), share_qkv=kw("share_qkv", False),
                                    gate_mode=kw("gate_mode", "dense"), gate_rank=kw("gate_rank", 4),
                                    gate_init=kw("gate_init", "neutral"))
        else:
            raise ValueError(f"Unknown attention mode: {mode}")

    def forward(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.impl(x, attn_mask)
