/*
 * mopk.h -- C ABI of libmopk.so, the MI355X (gfx950) MoP-attention kernel library.
 *
 * The reference (Eran-BA/MoP) is pure PyTorch and has no FFI of its own; the
 * drop-in boundary is the nn.Module surface (mop_amd/nn mirrors it).  This
 * header is the boundary BELOW those modules: one entry point per reference
 * attention core, each citing the reference code it replaces.  Plain pointers,
 * sizes and element strides only; no torch types.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer unless stated; the caller owns every
 *    buffer (inputs, outputs, `saved`, `workspace`); the library never
 *    allocates or frees device memory and keeps no state between calls.
 *  - Kernels are enqueued on `stream` (a hipStream_t passed as void*); no call
 *    synchronises the host.  All entry points are graph-capturable.
 *  - Strides are in ELEMENTS of the tensor's dtype; the innermost (dk) dimension
 *    must be contiguous.
 *  - Return value: 0 = ok, negative = MopkStatus; mopk_strerror() names it.
 *  - `*_part` outputs are per-batch partial sums with leading dimension B; the
 *    caller reduces them over dim 0 (keeps the kernels free of global atomics
 *    and bit-reproducible).
 */
#ifndef MOPK_H
#define MOPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOPK_VERSION 118 /* 118: attention dropout on every generic path, MopkCrossViewArgs.{dropout_p,dropout_seed}; 117: mopk_lens_means_{fwd,bwd}; 116: MopkEdgewiseExt.{n_extra,row_extra,col_extra,d_row_extra,d_col_extra}; 115: mask tensors on the fused dual-path kernels; 114: MopkEdgewiseArgs.mask (generic path), fused dense gate head; 113: attention dropout in the fused SDPA / Quartet kernels (dropout_p, dropout_seed, mopk_dropout_keep); 112: mopk_layernorm_*; 0.1.1: MopkEdgewiseArgs.{save_for_backward, ext}, MopkCrossViewArgs, *_fused_supported, y read by the sibling _bwd; 111: mopk_edgewise_reduce_parts */

typedef enum MopkStatus {
    MOPK_OK = 0,
    MOPK_ERR_BAD_SHAPE = -1,    /* non-positive or unsupported dimension */
    MOPK_ERR_BAD_ARG = -2,      /* null pointer / bad stride / bad enum */
    MOPK_ERR_UNSUPPORTED = -3,  /* variant not implemented by the requested path */
    MOPK_ERR_LAUNCH = -4,       /* hipGetLastError() != hipSuccess after a launch */
    MOPK_ERR_NO_DEVICE = -5     /* no gfx950 device visible */
} MopkStatus;

typedef enum MopkDtype { MOPK_F32 = 0, MOPK_BF16 = 1 } MopkDtype;

/* Arithmetic of the contractions.  FP32: exact fp32 FMA accumulation (parity
 * <= 1e-3 vs the reference CPU path, in practice ~1e-5).  BF16: bf16 MFMA
 * operands, fp32 accumulate, fp32 softmax/log/LSE/sigmoid (parity <= 1e-2). */
typedef enum MopkPrecision { MOPK_PREC_FP32 = 0, MOPK_PREC_BF16 = 1 } MopkPrecision;

/* Which implementation to run.  AUTO picks FUSED when the shape is supported
 * by the fused gfx950 kernels and GENERIC (multi-kernel, any shape) otherwise. */
typedef enum MopkPath { MOPK_PATH_AUTO = 0, MOPK_PATH_GENERIC = 1, MOPK_PATH_FUSED = 2 } MopkPath;

/* A (B,H,N,dk) tensor view: element (b,h,n,d) at ptr + b*sb + h*sh + n*sn + d. */
typedef struct MopkView4 {
    void *ptr;
    int64_t sb, sh, sn;
} MopkView4;

/* A per-view (V,B,H,N,dk) tensor view; sv == 0 means "one tensor shared by all views". */
typedef struct MopkView5 {
    void *ptr;
    int64_t sv, sb, sh, sn;
} MopkView5;

/* --------------------------------------------------------------------------
 * EdgewiseMSA attention core (low-rank gate head; dense head / lens banks through MopkEdgewiseExt).
 * Replaces reference mop/models/attention_variants.py:
 *   EdgewiseMSA.forward :500-562 (scores, per-view softmax, chain products,
 *   log features, gate head :319-331, score-space mix, re-normalise, value
 *   aggregation + transport) for attn_mask=None, no lens banks, dropout 0.
 * The qkv / proj Linear layers (:459, :564) stay outside (hipBLASLt GEMMs).
 *
 * Scale folding (verified identity):
 *   S_v = ((q * q_scale_v) (k * k_scale_v)^T) / sqrt(dk) = ((q * sqk_v) k^T),
 *   sqk_v = q_scale_v * k_scale_v / sqrt(dk).
 * For share_qkv=False pass per-view q/k (sv != 0) and sqk = 1/sqrt(dk).
 * -------------------------------------------------------------------------- */
/* Optional gate-head / feature variants (NULL ext = low-rank head, no lens bank).  The plain dense head (use_k3 = 0, n_lens = 0,
 * shared q/k, V <= 8 forward / V <= 6 backward, save_for_backward = 1) also runs on the fused kernels (MOPK_PATH_FUSED); the 3x3
 * convolution and the lens banks are generic-path only.
 *   dense head  : reference EdgewiseGateHead dense branch :250-272, :312-318 (Conv2d 1x1 C->16, GELU(tanh),
 *                 [use_k3: GELU again, Conv2d 3x3 16->16 pad 1], Conv2d 1x1 16->4, sigmoid)
 *   S lens bank : depthwise dilated 3x3 convolutions of the score planes appended to the feature stack :425-442, :523-533
 * Feature channel order (C = 2V + 2 + n_lens*V): S_0..S_{V-1}, S_0^T..S_{V-1}^T, Cr, Cl, lens[l*V+v]   :522-533
 * With the low-rank head and a lens bank, Wr/Wc (and dWr/dWc) are (4r, C) with this C. */
#define MOPK_MAX_LENS 4
#define MOPK_DENSE_HIDDEN 16
typedef struct MopkEdgewiseExt {
    int32_t gate_mode;          /* 0 = low-rank (row_proj/col_proj), 1 = dense                     :243 */
    int32_t use_k3;             /* dense only                                                       :253 */
    int32_t n_lens;             /* number of lens dilations L (0 = no lens bank), <= MOPK_MAX_LENS */
    int32_t lens_dil[MOPK_MAX_LENS]; /* dilation == padding of each 3x3 depthwise conv             :430-436 */
    const float *lens_w;        /* (L,V,3,3) = lens_bank.{l}.weight[:,0]                            */
    const float *W1, *b1;       /* edge_head.conv1 (16,C) / (16)                                    :251 */
    const float *W3, *b3;       /* edge_head.mid3 (16,16,3,3) / (16), use_k3 only                   :254 */
    const float *W2, *b2;       /* edge_head.conv2 (4,16) / (4)                                     :255 */
    /* backward outputs, fully reduced on device (any may be NULL when the matching input is unused) */
    float *dlens_w, *dW1, *db1, *dW3, *db3, *dW2, *db2;
    /* Extra feature channels of the LOW-RANK head on the FUSED path (gate_mode = 0, n_lens = 0): n_extra channels appended to the head's
     * input behind the 2V + 2 built-in ones, given directly as their row / column means -- row_extra, col_extra: (B,H,n_extra,N) fp32,
     * contiguous; Wr / Wc (dWr / dWc) are (4r, 2V + 2 + n_extra).  _bwd writes the gradient with respect to those values into
     * d_row_extra / d_col_extra (same shape).  This is how the S lens bank :425-442, :523-533 reaches the fused kernels with the
     * low-rank head: the row / column means of a depthwise dilated 3x3 convolution of S_v = Qe_v k^T are linear functionals of q and k
     * (row sums of S over three column ranges, shifted by the dilation), which the caller evaluates without ever forming a plane
     * (mop_amd/ops.py: lens_mean_features).  2V + 2 + n_extra <= 26. */
    int32_t n_extra;
    const float *row_extra, *col_extra;
    float *d_row_extra, *d_col_extra;
} MopkEdgewiseExt;

typedef struct MopkEdgewiseArgs {
    int32_t B, H, N, dk;
    int32_t V;          /* number of score views (>= 2)            :362 */
    int32_t r;          /* gate_rank                               :277 */
    int32_t io_dtype;   /* MopkDtype of q,k,v,y,dy,dq,dk,dv        */
    int32_t precision;  /* MopkPrecision                           */
    int32_t path;       /* MopkPath                                */
    int32_t save_for_backward; /* fused path: 1 = fwd also exports the chain state (prefix products, softmax constants,
                                * log-means) and the mix state (mixed logits, view log-sum-exp in fp32, softmax row statistics,
                                * P v0) into `saved`, and _bwd reads them.  0 = small `saved` (w * y_chain only); _bwd then
                                * re-runs the forward kernel into its workspace first to obtain that record (workspace_bytes()
                                * grows by the record's size).  Must have the same value in the fwd call, the bwd call and
                                * both *_bytes() queries.  The fused _bwd is three launches on `stream` (mix backward, the two
                                * D-chains, per-view gradients) that hand packed N x N slabs to each other through `workspace`. */
    float beta_not;     /* :361, used at :546 */

    MopkView5 q, k;          /* per-view (sv!=0) or shared (sv==0) queries / keys  :461-470 */
    MopkView4 v0, vL;        /* values of view 0 and of the last view :553-557 (before v_scale) */
    const float *sqk;        /* (V,H,dk) fp32, see above */
    const float *vs0, *vsL;  /* (H,dk) fp32: v_scale[0], v_scale[V-1] (ones when unshared) */
    const float *Wr, *br;    /* edge_head.row_proj weight (4r, 2V+2) / bias (4r)  :277 */
    const float *Wc, *bc;    /* edge_head.col_proj                                 :278 */
    const float *chain_logit;/* chain_value_logit, 1 fp32 on device                :451 */

    MopkView4 y;             /* out: (B,H,N,dk) view of the (B,N,D) tensor fed to proj :563 */

    void *saved;             /* out (fwd) / in (bwd): mopk_edgewise_saved_bytes() bytes */
    void *workspace;         /* scratch: mopk_edgewise_workspace_bytes() bytes */

    /* ---- backward only (ignored by _fwd) ---- */
    MopkView4 dy;            /* in : dL/dy, same geometry as y */
    MopkView5 dq, dk_;       /* out: dL/dq, dL/dk; sv==0 -> summed over views */
    MopkView4 dv0, dvL;      /* out: dL/dv0, dL/dvL (v_scale applied); dv0.ptr == dvL.ptr -> their sum is written once */
    float *dsqk_part;        /* out: (B,V,H,dk) */
    float *dvs0_part, *dvsL_part; /* out: (B,H,dk) */
    float *dWr, *dbr, *dWc, *dbc; /* out: (4r,2V+2),(4r),(4r,2V+2),(4r) -- fully reduced */
    float *dlogit_part;      /* out: (B,H) */

    const MopkEdgewiseExt *ext; /* host pointer; NULL = low-rank head without lens bank (the only form the fused path takes) */
    float dropout_p;         /* attn_drop on the mixed attention weights (:552); see MopkSdpaArgs.dropout_p */
    uint64_t dropout_seed;
    /* Optional attention mask, 1 = keep (generic path only).  EXTENSION: the reference's masked EdgewiseMSA is NaN for any blocking
     * mask (-inf scores enter the feature stack, :504-506 -> :518-546).  Here the mask acts on the probabilities only -- the per-view
     * softmaxes (:507) and the final one (:549-551); gate features see the unmasked scores.  Every row must keep at least one key. */
    const uint8_t *mask;
    int64_t mask_sb, mask_sh, mask_si;
} MopkEdgewiseArgs;

size_t mopk_edgewise_saved_bytes(const MopkEdgewiseArgs *a);
size_t mopk_edgewise_workspace_bytes(const MopkEdgewiseArgs *a);
int mopk_edgewise_lowrank_fwd(const MopkEdgewiseArgs *a, void *stream);   /* requires ext == NULL or ext->gate_mode == 0 */
int mopk_edgewise_lowrank_bwd(const MopkEdgewiseArgs *a, void *stream);
int mopk_edgewise_fwd(const MopkEdgewiseArgs *a, void *stream);           /* any head / lens variant (ext) */
int mopk_edgewise_bwd(const MopkEdgewiseArgs *a, void *stream);

/* --------------------------------------------------------------------------
 * MultiHopMSA (dual-path, scalar gates) attention core.
 * Replaces reference mop/models/attention_variants.py:200-229
 * (and experiments/cifar10_twohop_gates.py:55-99 `dual_path_mix`).
 * mask: optional uint8 (1 = keep, 0 = blocked), element (b,h,i,j) at
 * mask + b*mask_sb + h*mask_sh + i*mask_si + j (strides may be 0 to broadcast);
 * `causal` != 0 adds the lower-triangular mask without reading memory.
 * -------------------------------------------------------------------------- */
typedef struct MopkDualPathArgs {
    int32_t B, H, N, dk;
    int32_t hops;            /* >= 2  :174 ; 0 = no value-transport term (two-score attention only; fused kernels only,
                              * v2 / dv2 / chain_logit unused) -- CrossViewMixerMSA's 2x2 mix folds into (k1', k2') */
    int32_t io_dtype, precision, path;
    int32_t causal;
    float g_and, g_or, g_not, g_chain; /* python-float gates :188 */
    float beta_not;
    MopkView4 q1, k1, v1, q2, k2, v2;
    const uint8_t *mask;
    int64_t mask_sb, mask_sh, mask_si;
    const float *chain_logit;
    MopkView4 y;
    void *saved, *workspace;
    /* backward */
    MopkView4 dy, dq1, dk1, dv1, dq2, dk2, dv2;
    float *dlogit_part;      /* (B,H) */
    float dropout_p;         /* attn_drop on the mixed attention weights (attention_variants.py:222; the transport term uses the
                              * undropped A1, A2 :224-227); fused path only, see MopkSdpaArgs.dropout_p */
    uint64_t dropout_seed;
} MopkDualPathArgs;

size_t mopk_dualpath_saved_bytes(const MopkDualPathArgs *a);
size_t mopk_dualpath_workspace_bytes(const MopkDualPathArgs *a);
int mopk_dualpath_fwd(const MopkDualPathArgs *a, void *stream);
int mopk_dualpath_bwd(const MopkDualPathArgs *a, void *stream);
/* 1 if MOPK_PATH_AUTO runs this call on the fused kernels: bf16 arithmetic, dk 32/64, chain gate 0 (causal flag and mask tensor ok) */
int mopk_dualpath_fused_supported(const MopkDualPathArgs *a);

/* --------------------------------------------------------------------------
 * Quartet CausalSelfAttention core.
 * Replaces reference mop/models/quartet_attn_patch.py:88-121 (scores, row
 * z-norm with unbiased std over the full row, product mix, causal mask,
 * additive attention_mask, softmax, AV).  use_quartet==0 -> :108-110.
 * add_mask: optional fp32 additive mask, element (b,h,i,j) at
 * add_mask + b*am_sb + h*am_sh + i*am_si + j.
 * -------------------------------------------------------------------------- */
typedef struct MopkQuartetArgs {
    int32_t B, H, T, dh;
    int32_t io_dtype, precision, path;
    int32_t use_quartet;
    float eps;               /* score_norm_eps :31 */
    MopkView4 q, k, v, q2, k2;
    const float *mixture;        /* 1 fp32 on device :52 */
    const float *quartet_scale;  /* 1 fp32 on device :55 */
    const float *add_mask;
    int64_t am_sb, am_sh, am_si;
    MopkView4 y;
    float *attn;             /* optional out (B,H,T,T) fp32 for need_weights=True :125 */
    void *saved, *workspace;
    /* backward */
    MopkView4 dy, dq, dk_, dv, dq2, dk2;
    float *dmixture_part, *dqscale_part; /* (B,H) */
    float dropout_p;         /* attn_dropout on the probabilities (quartet_attn_patch.py:119); see MopkSdpaArgs.dropout_p */
    uint64_t dropout_seed;
} MopkQuartetArgs;

size_t mopk_quartet_saved_bytes(const MopkQuartetArgs *a);
size_t mopk_quartet_workspace_bytes(const MopkQuartetArgs *a);
int mopk_quartet_fwd(const MopkQuartetArgs *a, void *stream);
int mopk_quartet_bwd(const MopkQuartetArgs *a, void *stream);  /* needs q,k,v,(q2,k2), y (forward output), dy */
/* 1 if MOPK_PATH_AUTO runs this call on the fused kernels: bf16 arithmetic, dh 32/64, no add_mask, attn == NULL */
int mopk_quartet_fused_supported(const MopkQuartetArgs *a);

/* --------------------------------------------------------------------------
 * Plain scaled-dot-product attention core.
 * Replaces BaselineMSA.forward attention_variants.py:42-46, MSA.forward
 * components.py:61-64 and MultiheadSelfAttention.forward whisper_mop.py:163-175.
 * path: MOPK_PATH_AUTO picks the fused kernels (sdpa_flash.hip: no N x N map in HBM) when they cover the call.
 * -------------------------------------------------------------------------- */
typedef struct MopkSdpaArgs {
    int32_t B, H, N, dk;
    int32_t io_dtype, precision, path;
    int32_t causal;
    MopkView4 q, k, v;
    const uint8_t *mask;     /* optional, 1 = keep */
    int64_t mask_sb, mask_sh, mask_si;
    const float *bias;       /* optional additive fp32 */
    int64_t bias_sb, bias_sh, bias_si;
    MopkView4 y;
    void *saved, *workspace;
    MopkView4 dy, dq, dk_, dv;
    /* attention dropout (`self.attn_drop(A)` components.py:62, attention_variants.py:45, `self.attn_dropout` quartet_attn_patch.py:119):
     * probabilities are multiplied by keep(b,h,i,j) / (1 - dropout_p) after the softmax, keep = mopk_dropout_keep(); the same
     * function is evaluated again in _bwd (nothing is stored), so pass the same p and seed to both.  0 = off.  Both paths draw the
     * same mask from a seed (the generic path multiplies its N x N map by it in a workspace plane). */
    float dropout_p;
    uint64_t dropout_seed;
} MopkSdpaArgs;

size_t mopk_sdpa_saved_bytes(const MopkSdpaArgs *a);
size_t mopk_sdpa_workspace_bytes(const MopkSdpaArgs *a);
int mopk_sdpa_fwd(const MopkSdpaArgs *a, void *stream);
int mopk_sdpa_bwd(const MopkSdpaArgs *a, void *stream);  /* needs q,k,v, y (forward output), dy; mask/bias as in the forward */
/* 1 if the fused (flash-style) gfx950 kernels take this call under MOPK_PATH_AUTO: bf16 arithmetic, dk 32/64, no mask/bias tensor */
int mopk_sdpa_fused_supported(const MopkSdpaArgs *a);
/* The dropout mask of the fused kernels as a host function (tests / reproducibility): 1 if edge (query i, key j) of head-batch
 * index bh = b * H + h is kept under (seed, p).  Counter-based: lowbias32(lowbias32(i * 0x9E3779B1 + bh * 0x85EBCA77 + seed_hi)
 * ^ seed_lo ^ j * 0xC2B2AE3D) >= p * 2^32. */
int mopk_dropout_keep(uint64_t seed, float p, int64_t bh, int64_t i, int64_t j);

/* --------------------------------------------------------------------------
 * CrossViewMixerMSA attention core.
 * Replaces reference mop/models/attention_variants.py:90-110 (`_compute_logits`: S1, S2, S12, S21, 2x2 mix,
 * transpose cues) and :120-153 (mask, softmax, optional per-key prior sharpening, A v1).
 * anchor_mode: 0 = "fixed" (row clamp(fixed_k_star)), 1 = "argmax_row_sum" (argmax over the row sums of
 * softmax(S2) -- all ~1, i.e. decided by rounding; reproduced as written), 2 = any other string (row 0).
 * -------------------------------------------------------------------------- */
typedef struct MopkCrossViewArgs {
    int32_t B, H, N, dk;
    int32_t io_dtype, precision, path;
    int32_t causal;
    int32_t use_prior;       /* enable_per_key_prior && prior_weight > 0   :126 */
    int32_t anchor_mode, fixed_k_star;
    float t1, t2;            /* transpose-cue weights; 0 when use_transpose_cues is False :106-110 */
    float prior_weight;
    MopkView4 q1, k1, v1, q2, k2;
    const float *mix;        /* (2,2) fp32 on device :78 */
    const uint8_t *mask;     /* optional, 1 = keep */
    int64_t mask_sb, mask_sh, mask_si;
    MopkView4 y;
    void *saved, *workspace;
    int32_t *k_star;         /* optional out (B,H) int32: anchor row used by the prior */
    /* backward */
    MopkView4 dy, dq1, dk1, dv1, dq2, dk2;
    float *dmix_part;        /* (B,H,4) */
    float dropout_p;         /* attn_drop on the final attention weights (:151), see MopkSdpaArgs.dropout_p */
    uint64_t dropout_seed;
} MopkCrossViewArgs;

size_t mopk_crossview_saved_bytes(const MopkCrossViewArgs *a);
size_t mopk_crossview_workspace_bytes(const MopkCrossViewArgs *a);
int mopk_crossview_fwd(const MopkCrossViewArgs *a, void *stream);
int mopk_crossview_bwd(const MopkCrossViewArgs *a, void *stream);

/* Sum the per-batch partial gradients an Edgewise backward left in a->dsqk_part (B,V,H,dk), a->dvs0_part, a->dvsL_part
 * (B,H,dk) and a->dlogit_part (B,H) over the batch, in a fixed order (bitwise reproducible), into dsqk (V,H,dk), dvs0, dvsL
 * (H,dk) and dlogit (1): one launch in place of the four reductions `q_scale.grad`, `v_scale.grad` and
 * `chain_value_logit.grad` would otherwise need on the host side (attention_variants.py:374-378, :451 are the parameters).
 * Reads only B, V, H, dk and the four *_part pointers of `a`. */
int mopk_edgewise_reduce_parts(const MopkEdgewiseArgs *a, float *dsqk, float *dvs0, float *dvsL, float *dlogit, void *stream);

/* -------------------------------------------------------------------------- */
/* LayerNorm prologue / residual epilogue of the attention path.
 *
 * Replaces, around the core, the `self.ln1(x)` of `x = x + self.dp1(self.attn(self.ln1(x)))`
 * (experiments/cifar100_edgewise_gates.py:371-374; mop/models/components.py:96-98 is the same block): the forward applies
 * torch.nn.LayerNorm (biased variance, eps inside the sqrt, fp32 statistics) to `rows` token rows of width `dim` and writes the
 * result in `y_dtype` -- the dtype the qkv GEMM consumes -- in one pass; mean / rstd are kept for the backward.  The backward
 * returns dx = dres + LN'(dy): `dres` is the gradient arriving on the residual branch (NULL: none), so the `x + ...` add of the
 * block costs no extra pass.  dgamma / dbeta are summed over rows in a fixed order (bitwise reproducible).
 * Rows are contiguous with leading dimensions x_ld / y_ld (elements, multiples of 8); dim % 8 == 0, dim <= 4096, pointers
 * 16-byte aligned; anything else returns MOPK_ERR_UNSUPPORTED / MOPK_ERR_BAD_ARG. */
typedef struct MopkLayerNormArgs {
    int64_t rows;
    int32_t dim;
    int32_t x_dtype, y_dtype, p_dtype;   /* MopkDtype of x/dx/dres, of y/dy, of gamma/beta */
    float eps;
    int64_t x_ld, y_ld;
    const void *x, *gamma, *beta;        /* beta may be NULL */
    void *y;                             /* forward out */
    float *mean, *rstd;                  /* (rows) fp32: forward out (both or neither), backward in */
    const void *dy, *dres;               /* backward in; dres may be NULL */
    void *dx;                            /* backward out (may alias dres) */
    float *dgamma, *dbeta;               /* (dim) fp32 backward out; either may be NULL */
    void *workspace;                     /* backward: mopk_layernorm_workspace_bytes() */
} MopkLayerNormArgs;
size_t mopk_layernorm_workspace_bytes(const MopkLayerNormArgs *a);
int mopk_layernorm_fwd(const MopkLayerNormArgs *a, void *stream);
int mopk_layernorm_bwd(const MopkLayerNormArgs *a, void *stream);

/* --------------------------------------------------------------------------
 * Row / column means of the S lens bank's planes in closed form -- the values MopkEdgewiseExt.row_extra / col_extra take.
 * The bank (attention_variants.py:425-442, :523-533) convolves every score plane S_v = (q * sqk_v) k^T with a depthwise dilated
 * 3x3 kernel (zero padding = dilation); the low-rank head reads only row / column means of the result (:323-326), which are
 * linear functionals of q and k (O(N dk) per (b, h, view); see mop_amd/csrc/lens_means.hip).  N <= 224, dk in {16, 32, 64},
 * L * V <= 16, and a working set (grows with the largest dilation) within the CU's 160 KB of LDS: mopk_lens_means_supported.  q / k: the shared (sv == 0) queries / keys of the Edgewise call.
 *   _fwd: row, col (B,H,L*V,N) fp32, channel l * V + v (the reference's order, :531).
 *   _bwd: given d_row / d_col, ADDS the q / k gradients into dq / dk_ (the buffers the Edgewise backward has already written,
 *         io_dtype elements) and writes per-(b) / per-(b,h) partials of the scale and weight gradients for the caller to sum:
 *         dsqk_part (B,V,H,dk), dlens_part (B*H,L,V,3,3). */
typedef struct MopkLensMeansArgs {
    int32_t B, H, N, dk, V, L;
    int32_t io_dtype;              /* MopkDtype of q, k, dq, dk_ */
    int32_t dil[MOPK_MAX_LENS];    /* dilation == padding of each lens   :430-436 */
    MopkView4 q, k;
    const float *sqk;              /* (V,H,dk) fp32, as MopkEdgewiseArgs.sqk */
    const float *lens_w;           /* (L,V,3,3) fp32: lens_bank[l].weight[:, 0] */
    float *row, *col;              /* fwd out */
    const float *d_row, *d_col;    /* bwd in  */
    MopkView4 dq, dk_;             /* bwd in/out: += */
    float *dsqk_part, *dlens_part; /* bwd out */
} MopkLensMeansArgs;
int mopk_lens_means_supported(const MopkLensMeansArgs *a, int backward);   /* 1 if the kernels take this shape (dimensions and dil only) */
int mopk_lens_means_fwd(const MopkLensMeansArgs *a, void *stream);
int mopk_lens_means_bwd(const MopkLensMeansArgs *a, void *stream);

/* -------------------------------------------------------------------------- */
int mopk_version(void);
const char *mopk_strerror(int status);
/* 1 if the fused gfx950 kernels cover this Edgewise shape (N, dk, V, r). */
int mopk_edgewise_fused_supported(const MopkEdgewiseArgs *a);
/* name prefix of the dominant kernel(s) an edgewise fwd/bwd with these arguments launches (static string; bench.py matches
 * rocprofv3 kernel-trace rows with it -- the fused backward is three launches of one template, `ew_fused_bwd_kernel<.., 0|1|2>`,
 * and all three rows are summed). */
const char *mopk_edgewise_dominant_kernel(const MopkEdgewiseArgs *a, int backward);

#ifdef __cplusplus
}
#endif
#endif /* MOPK_H */
