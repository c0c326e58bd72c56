// extern "C" entry points of libmopk.so (see include/mopk.h).
#include "common.h"
#include "flash_common.h"

namespace mopk {
size_t ew_generic_saved_bytes(const MopkEdgewiseArgs *a);
size_t ew_generic_workspace_bytes(const MopkEdgewiseArgs *a);
int ew_generic_fwd(const MopkEdgewiseArgs *a, hipStream_t st);
int ew_generic_bwd(const MopkEdgewiseArgs *a, hipStream_t st);
int ew_reduce_parts(const MopkEdgewiseArgs *a, float *dsqk, float *dvs0, float *dvsL, float *dlogit, hipStream_t st);
int ew_fused_fwd_supported(const MopkEdgewiseArgs *a);
int ew_fused_fwd(const MopkEdgewiseArgs *a, hipStream_t st);
size_t ew_fused_saved_bytes(const MopkEdgewiseArgs *a);
int ew_fused_bwd_supported(const MopkEdgewiseArgs *a);
size_t ew_fused_bwd_ws_bytes(const MopkEdgewiseArgs *a);
int ew_fused_bwd(const MopkEdgewiseArgs *a, hipStream_t st);
size_t sdpa_saved_bytes(const MopkSdpaArgs *a); size_t sdpa_ws_bytes(const MopkSdpaArgs *a);
int sdpa_fwd(const MopkSdpaArgs *a, hipStream_t st); int sdpa_bwd(const MopkSdpaArgs *a, hipStream_t st);
int sdpa_flash_supported(const MopkSdpaArgs *a, bool bwd); size_t sdpa_flash_saved_bytes(const MopkSdpaArgs *a); size_t sdpa_flash_ws_bytes(const MopkSdpaArgs *a);
int sdpa_flash_fwd(const MopkSdpaArgs *a, hipStream_t st); int sdpa_flash_bwd(const MopkSdpaArgs *a, hipStream_t st);
size_t dp_saved_bytes(const MopkDualPathArgs *a); size_t dp_ws_bytes(const MopkDualPathArgs *a);
int dp_flash_supported(const MopkDualPathArgs *a, bool bwd); size_t dp_flash_saved_bytes(const MopkDualPathArgs *a); size_t dp_flash_ws_bytes(const MopkDualPathArgs *a);
int dp_flash_fwd(const MopkDualPathArgs *a, hipStream_t st); int dp_flash_bwd(const MopkDualPathArgs *a, hipStream_t st);
int dp_fwd(const MopkDualPathArgs *a, hipStream_t st); int dp_bwd(const MopkDualPathArgs *a, hipStream_t st);
size_t qt_saved_bytes(const MopkQuartetArgs *a); size_t qt_ws_bytes(const MopkQuartetArgs *a);
int qt_flash_supported(const MopkQuartetArgs *a, bool bwd); size_t qt_flash_saved_bytes(const MopkQuartetArgs *a); size_t qt_flash_ws_bytes(const MopkQuartetArgs *a);
int qt_flash_fwd(const MopkQuartetArgs *a, hipStream_t st); int qt_flash_bwd(const MopkQuartetArgs *a, hipStream_t st);
size_t cv_saved_bytes(const MopkCrossViewArgs *a); size_t cv_ws_bytes(const MopkCrossViewArgs *a);
int cv_fwd(const MopkCrossViewArgs *a, hipStream_t st); int cv_bwd(const MopkCrossViewArgs *a, hipStream_t st);
int qt_fwd(const MopkQuartetArgs *a, hipStream_t st); int qt_bwd(const MopkQuartetArgs *a, hipStream_t st);
int lens_means_run(const MopkLensMeansArgs *a, bool bwd, hipStream_t st);
int lens_means_supported(const MopkLensMeansArgs *a, bool bwd);

static int ew_validate(const MopkEdgewiseArgs *a, bool bwd) {
    if (!a) return MOPK_ERR_BAD_ARG;
    if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return MOPK_ERR_BAD_ARG;
    if (a->B <= 0 || a->H <= 0 || a->N <= 0 || a->dk <= 0 || a->V < 2 || a->r < 1) return MOPK_ERR_BAD_SHAPE;
    if (a->io_dtype != MOPK_F32 && a->io_dtype != MOPK_BF16) return MOPK_ERR_BAD_ARG;
    if (a->precision != MOPK_PREC_FP32 && a->precision != MOPK_PREC_BF16) return MOPK_ERR_BAD_ARG;
    if (a->mask && a->path == MOPK_PATH_FUSED) return MOPK_ERR_UNSUPPORTED;      // attention mask: generic path only
    const MopkEdgewiseExt *x = a->ext;
    const bool dense = x && x->gate_mode == 1;
    if (x) {
        if ((x->gate_mode != 0 && x->gate_mode != 1) || x->n_lens < 0 || x->n_lens > MOPK_MAX_LENS) return MOPK_ERR_BAD_ARG;
        for (int l = 0; l < x->n_lens; ++l) if (x->lens_dil[l] < 1) return MOPK_ERR_BAD_ARG;
        if (x->n_lens > 0 && (!x->lens_w || (bwd && !x->dlens_w))) return MOPK_ERR_BAD_ARG;
        if (dense && (!x->W1 || !x->b1 || !x->W2 || !x->b2 || (x->use_k3 && (!x->W3 || !x->b3)))) return MOPK_ERR_BAD_ARG;
        if (dense && bwd && (!x->dW1 || !x->db1 || !x->dW2 || !x->db2 || (x->use_k3 && (!x->dW3 || !x->db3)))) return MOPK_ERR_BAD_ARG;
        // lens banks, the 3x3 convolution and every dense-head backward: generic path only (the fused forward takes the plain dense head)
        if (a->path == MOPK_PATH_FUSED && (x->n_lens > 0 || (dense && x->use_k3))) return MOPK_ERR_UNSUPPORTED;
        // extra feature channels (given as row / column means): low-rank head on the fused path only
        if (x->n_extra < 0 || (x->n_extra > 0 && (!x->row_extra || !x->col_extra || (bwd && (!x->d_row_extra || !x->d_col_extra))))) return MOPK_ERR_BAD_ARG;
        if (x->n_extra > 0 && (dense || x->n_lens > 0 || a->path != MOPK_PATH_FUSED)) return MOPK_ERR_UNSUPPORTED;
    }
    if (!a->q.ptr || !a->k.ptr || !a->v0.ptr || !a->vL.ptr || !a->sqk || !a->vs0 || !a->vsL || !a->chain_logit || !a->saved ||
        !a->workspace)
        return MOPK_ERR_BAD_ARG;
    if (!dense && (!a->Wr || !a->br || !a->Wc || !a->bc)) return MOPK_ERR_BAD_ARG;
    if (!bwd && !a->y.ptr) return MOPK_ERR_BAD_ARG;
    if (bwd && (!a->dy.ptr || !a->dq.ptr || !a->dk_.ptr || !a->dv0.ptr || !a->dvL.ptr || !a->dsqk_part ||
                !a->dvs0_part || !a->dvsL_part || !a->dlogit_part))
        return MOPK_ERR_BAD_ARG;
    if (bwd && !dense && (!a->dWr || !a->dbr || !a->dWc || !a->dbc)) return MOPK_ERR_BAD_ARG;
    if ((a->q.sv == 0) != (a->k.sv == 0)) return MOPK_ERR_BAD_ARG;
    if (bwd && ((a->dq.sv == 0) != (a->q.sv == 0) || (a->dk_.sv == 0) != (a->k.sv == 0))) return MOPK_ERR_BAD_ARG;
    return MOPK_OK;
}
}  // namespace mopk

using namespace mopk;

extern "C" {

int mopk_version(void) { return MOPK_VERSION; }
int mopk_dropout_keep(uint64_t seed, float p, int64_t bh, int64_t i, int64_t j) {
    const FaDrop d = fa_drop(p, seed);
    if (!d.thresh) return 1;
    return fa_drop_keep(d, fa_drop_row(d, (int)bh, (int)i), (int)j) ? 1 : 0;
}

const char *mopk_strerror(int s) {
    switch (s) {
        case MOPK_OK: return "ok";
        case MOPK_ERR_BAD_SHAPE: return "mopk: bad or unsupported shape";
        case MOPK_ERR_BAD_ARG: return "mopk: bad argument (null pointer, stride or enum)";
        case MOPK_ERR_UNSUPPORTED: return "mopk: variant not supported by the requested path";
        case MOPK_ERR_LAUNCH: return "mopk: HIP kernel launch failed";
        case MOPK_ERR_NO_DEVICE: return "mopk: no gfx950 device";
        default: return "mopk: unknown status";
    }
}

int mopk_edgewise_fused_supported(const MopkEdgewiseArgs *a) { return a ? ew_fused_fwd_supported(a) : 0; }

const char *mopk_edgewise_dominant_kernel(const MopkEdgewiseArgs *a, int backward) {
    const bool fused = a && a->path != MOPK_PATH_GENERIC && ew_fused_fwd_supported(a);
    if (!fused) return a && a->precision == MOPK_PREC_BF16 ? "bgemm_mfma_kernel" : "bgemm_kernel";
    return backward ? "ew_fused_bwd_kernel" : "ew_fused_fwd_kernel";
}

size_t mopk_edgewise_saved_bytes(const MopkEdgewiseArgs *a) {
    if (!a || a->B <= 0 || a->H <= 0 || a->N <= 0 || a->dk <= 0 || a->V < 2 || a->r < 1) return 0;
    if (a->path == MOPK_PATH_FUSED) return ew_fused_fwd_supported(a) ? ew_fused_saved_bytes(a) : 0;
    return ew_generic_saved_bytes(a);
}
size_t mopk_edgewise_workspace_bytes(const MopkEdgewiseArgs *a) {
    if (!a || a->B <= 0 || a->H <= 0 || a->N <= 0 || a->dk <= 0 || a->V < 2 || a->r < 1) return 0;
    if (a->path == MOPK_PATH_FUSED) return ew_fused_fwd_supported(a) ? ew_fused_bwd_ws_bytes(a) : 0;   // backward scratch (forward needs none)
    return ew_generic_workspace_bytes(a);
}
int mopk_edgewise_fwd(const MopkEdgewiseArgs *a, void *stream) {
    int rc = ew_validate(a, false);
    if (rc) return rc;
    if (a->path == MOPK_PATH_FUSED) return ew_fused_fwd(a, (hipStream_t)stream);
    return ew_generic_fwd(a, (hipStream_t)stream);
}
int mopk_edgewise_bwd(const MopkEdgewiseArgs *a, void *stream) {
    int rc = ew_validate(a, true);
    if (rc) return rc;
    if (a->path == MOPK_PATH_FUSED) return ew_fused_bwd(a, (hipStream_t)stream);
    return ew_generic_bwd(a, (hipStream_t)stream);
}
int mopk_lens_means_supported(const MopkLensMeansArgs *a, int backward) { return lens_means_supported(a, backward != 0); }
int mopk_lens_means_fwd(const MopkLensMeansArgs *a, void *stream) { return lens_means_run(a, false, (hipStream_t)stream); }
int mopk_lens_means_bwd(const MopkLensMeansArgs *a, void *stream) { return lens_means_run(a, true, (hipStream_t)stream); }
int mopk_edgewise_lowrank_fwd(const MopkEdgewiseArgs *a, void *stream) {
    if (a && a->ext && a->ext->gate_mode != 0) return MOPK_ERR_BAD_ARG;
    return mopk_edgewise_fwd(a, stream);
}
int mopk_edgewise_reduce_parts(const MopkEdgewiseArgs *a, float *dsqk, float *dvs0, float *dvsL, float *dlogit, void *stream) {
    if (!a || !dsqk || !dvs0 || !dvsL || !dlogit || !a->dsqk_part || !a->dvs0_part || !a->dvsL_part || !a->dlogit_part)
        return MOPK_ERR_BAD_ARG;
    if (a->B <= 0 || a->H <= 0 || a->V <= 0 || a->dk <= 0) return MOPK_ERR_BAD_SHAPE;
    return ew_reduce_parts(a, dsqk, dvs0, dvsL, dlogit, (hipStream_t)stream);
}
int mopk_edgewise_lowrank_bwd(const MopkEdgewiseArgs *a, void *stream) {
    if (a && a->ext && a->ext->gate_mode != 0) return MOPK_ERR_BAD_ARG;
    return mopk_edgewise_bwd(a, stream);
}

// ---- sibling cores (generic multi-kernel paths, attn_generic.hip) ----
static bool v4ok(const MopkView4 &v) { return v.ptr != nullptr; }
static int base_ok(int B, int H, int N, int dk, int io, int prec) {
    if (B <= 0 || H <= 0 || N <= 0 || dk <= 0) return MOPK_ERR_BAD_SHAPE;
    if ((io != MOPK_F32 && io != MOPK_BF16) || (prec != MOPK_PREC_FP32 && prec != MOPK_PREC_BF16)) return MOPK_ERR_BAD_ARG;
    return MOPK_OK;
}
// path: AUTO = fused flash kernels when they cover the call (bf16 arithmetic, dk 32/64, no mask/bias tensor), else generic
static bool sdpa_use_flash(const MopkSdpaArgs *a, bool bwd) {
    return a->path != MOPK_PATH_GENERIC && sdpa_flash_supported(a, bwd);
}
int mopk_sdpa_fused_supported(const MopkSdpaArgs *a) { return (a && a->B > 0 && a->H > 0 && a->N > 0 && a->dk > 0) ? sdpa_flash_supported(a, false) : 0; }
size_t mopk_sdpa_saved_bytes(const MopkSdpaArgs *a) {
    if (!(a && a->B > 0 && a->H > 0 && a->N > 0 && a->dk > 0)) return 0;
    return sdpa_use_flash(a, false) ? sdpa_flash_saved_bytes(a) : sdpa_saved_bytes(a);
}
size_t mopk_sdpa_workspace_bytes(const MopkSdpaArgs *a) {
    if (!(a && a->B > 0 && a->H > 0 && a->N > 0 && a->dk > 0)) return 0;
    return sdpa_use_flash(a, false) ? sdpa_flash_ws_bytes(a) : sdpa_ws_bytes(a);
}
int mopk_sdpa_fwd(const MopkSdpaArgs *a, void *stream) {
    if (!a) return MOPK_ERR_BAD_ARG;
    int rc = base_ok(a->B, a->H, a->N, a->dk, a->io_dtype, a->precision); if (rc) return rc;
    if (!v4ok(a->q) || !v4ok(a->k) || !v4ok(a->v) || !v4ok(a->y) || !a->saved || !a->workspace) return MOPK_ERR_BAD_ARG;
    if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return MOPK_ERR_BAD_ARG;
    if (sdpa_use_flash(a, false)) return sdpa_flash_fwd(a, (hipStream_t)stream);
    if (a->path == MOPK_PATH_FUSED) return MOPK_ERR_UNSUPPORTED;
    return sdpa_fwd(a, (hipStream_t)stream);
}
int mopk_sdpa_bwd(const MopkSdpaArgs *a, void *stream) {      // the fused path also reads `y` (the forward's output)
    if (!a) return MOPK_ERR_BAD_ARG;
    int rc = base_ok(a->B, a->H, a->N, a->dk, a->io_dtype, a->precision); if (rc) return rc;
    if (!v4ok(a->q) || !v4ok(a->k) || !v4ok(a->v) || !v4ok(a->y) || !v4ok(a->dy) || !v4ok(a->dq) || !v4ok(a->dk_) || !v4ok(a->dv) ||
        !a->saved || !a->workspace)
        return MOPK_ERR_BAD_ARG;
    if (sdpa_use_flash(a, false)) return sdpa_flash_bwd(a, (hipStream_t)stream);   // same decision as the forward (saved layout)
    if (a->path == MOPK_PATH_FUSED) return MOPK_ERR_UNSUPPORTED;
    return sdpa_bwd(a, (hipStream_t)stream);
}
size_t mopk_crossview_saved_bytes(const MopkCrossViewArgs *a) { return (a && a->B > 0 && a->H > 0 && a->N > 0 && a->dk > 0) ? cv_saved_bytes(a) : 0; }
size_t mopk_crossview_workspace_bytes(const MopkCrossViewArgs *a) { return (a && a->B > 0 && a->H > 0 && a->N > 0 && a->dk > 0) ? cv_ws_bytes(a) : 0; }
static int cv_validate(const MopkCrossViewArgs *a, bool bwd) {
    if (!a) return MOPK_ERR_BAD_ARG;
    int rc = base_ok(a->B, a->H, a->N, a->dk, a->io_dtype, a->precision); if (rc) return rc;
    if (!v4ok(a->q1) || !v4ok(a->k1) || !v4ok(a->v1) || !v4ok(a->q2) || !v4ok(a->k2) || !a->mix || !a->saved || !a->workspace) return MOPK_ERR_BAD_ARG;
    if (a->anchor_mode < 0 || a->anchor_mode > 2 || a->dropout_p < 0.f || a->dropout_p >= 1.f) return MOPK_ERR_BAD_ARG;
    if (!bwd && !v4ok(a->y)) return MOPK_ERR_BAD_ARG;
    if (bwd && (!v4ok(a->dy) || !v4ok(a->dq1) || !v4ok(a->dk1) || !v4ok(a->dv1) || !v4ok(a->dq2) || !v4ok(a->dk2) || !a->dmix_part)) return MOPK_ERR_BAD_ARG;
    if (a->path == MOPK_PATH_FUSED) return MOPK_ERR_UNSUPPORTED;
    return MOPK_OK;
}
int mopk_crossview_fwd(const MopkCrossViewArgs *a, void *stream) { int rc = cv_validate(a, false); return rc ? rc : cv_fwd(a, (hipStream_t)stream); }
int mopk_crossview_bwd(const MopkCrossViewArgs *a, void *stream) { int rc = cv_validate(a, true); return rc ? rc : cv_bwd(a, (hipStream_t)stream); }
static bool dp_ok(const MopkDualPathArgs *a) { return a && a->B > 0 && a->H > 0 && a->N > 0 && a->dk > 0 && (a->hops >= 2 || a->hops == 0); }
// path: AUTO = fused kernels (sdpa_flash.hip) when the chain gate is 0 and there is no mask tensor, else generic
static bool dp_use_flash(const MopkDualPathArgs *a) { return a->path != MOPK_PATH_GENERIC && dp_flash_supported(a, false); }
int mopk_dualpath_fused_supported(const MopkDualPathArgs *a) { return dp_ok(a) ? dp_flash_supported(a, false) : 0; }
size_t mopk_dualpath_saved_bytes(const MopkDualPathArgs *a) { return !dp_ok(a) ? 0 : (dp_use_flash(a) ? dp_flash_saved_bytes(a) : dp_saved_bytes(a)); }
size_t mopk_dualpath_workspace_bytes(const MopkDualPathArgs *a) { return !dp_ok(a) ? 0 : (dp_use_flash(a) ? dp_flash_ws_bytes(a) : dp_ws_bytes(a)); }
int mopk_dualpath_fwd(const MopkDualPathArgs *a, void *stream) {
    if (!a) return MOPK_ERR_BAD_ARG;
    int rc = base_ok(a->B, a->H, a->N, a->dk, a->io_dtype, a->precision); if (rc) return rc;
    if (a->hops < 2 && a->hops != 0) return MOPK_ERR_BAD_SHAPE;
    if (!v4ok(a->q1) || !v4ok(a->k1) || !v4ok(a->v1) || !v4ok(a->q2) || !v4ok(a->k2) || !v4ok(a->y) || !a->saved || !a->workspace)
        return MOPK_ERR_BAD_ARG;
    if (a->hops > 0 && (!v4ok(a->v2) || !a->chain_logit)) return MOPK_ERR_BAD_ARG;
    if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return MOPK_ERR_BAD_ARG;
    if (dp_use_flash(a)) return dp_flash_fwd(a, (hipStream_t)stream);
    if (a->path == MOPK_PATH_FUSED || a->hops == 0) return MOPK_ERR_UNSUPPORTED;   // hops == 0 (no transport term) exists on the fused kernels only
    return dp_fwd(a, (hipStream_t)stream);
}
int mopk_dualpath_bwd(const MopkDualPathArgs *a, void *stream) {
    if (!a) return MOPK_ERR_BAD_ARG;
    int rc = base_ok(a->B, a->H, a->N, a->dk, a->io_dtype, a->precision); if (rc) return rc;
    if (a->hops < 2 && a->hops != 0) return MOPK_ERR_BAD_SHAPE;
    if (!v4ok(a->q1) || !v4ok(a->k1) || !v4ok(a->v1) || !v4ok(a->q2) || !v4ok(a->k2) || !v4ok(a->y) ||
        !v4ok(a->dy) || !v4ok(a->dq1) || !v4ok(a->dk1) || !v4ok(a->dv1) || !v4ok(a->dq2) || !v4ok(a->dk2) ||
        !a->dlogit_part || !a->saved || !a->workspace) return MOPK_ERR_BAD_ARG;
    if (a->hops > 0 && (!v4ok(a->v2) || !v4ok(a->dv2) || !a->chain_logit)) return MOPK_ERR_BAD_ARG;
    if (dp_use_flash(a)) return dp_flash_bwd(a, (hipStream_t)stream);        // same decision as the forward (saved layout)
    if (a->path == MOPK_PATH_FUSED || a->hops == 0) return MOPK_ERR_UNSUPPORTED;
    return dp_bwd(a, (hipStream_t)stream);
}
static bool qt_ok(const MopkQuartetArgs *a) { return a && a->B > 0 && a->H > 0 && a->T > 0 && a->dh > 0; }
// path: AUTO = fused kernels (quartet_flash.hip) without an additive mask / returned attention weights, else generic
static bool qt_use_flash(const MopkQuartetArgs *a) { return a->path != MOPK_PATH_GENERIC && qt_flash_supported(a, false); }
int mopk_quartet_fused_supported(const MopkQuartetArgs *a) { return qt_ok(a) ? qt_flash_supported(a, false) : 0; }
size_t mopk_quartet_saved_bytes(const MopkQuartetArgs *a) { return !qt_ok(a) ? 0 : (qt_use_flash(a) ? qt_flash_saved_bytes(a) : qt_saved_bytes(a)); }
size_t mopk_quartet_workspace_bytes(const MopkQuartetArgs *a) { return !qt_ok(a) ? 0 : (qt_use_flash(a) ? qt_flash_ws_bytes(a) : qt_ws_bytes(a)); }
int mopk_quartet_fwd(const MopkQuartetArgs *a, void *stream) {
    if (!a) return MOPK_ERR_BAD_ARG;
    int rc = base_ok(a->B, a->H, a->T, a->dh, a->io_dtype, a->precision); if (rc) return rc;
    if (!v4ok(a->q) || !v4ok(a->k) || !v4ok(a->v) || !v4ok(a->y) || !a->saved || !a->workspace) return MOPK_ERR_BAD_ARG;
    if (a->use_quartet && (!v4ok(a->q2) || !v4ok(a->k2) || !a->mixture || !a->quartet_scale)) return MOPK_ERR_BAD_ARG;
    if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return MOPK_ERR_BAD_ARG;
    if (qt_use_flash(a)) return qt_flash_fwd(a, (hipStream_t)stream);
    if (a->path == MOPK_PATH_FUSED) return MOPK_ERR_UNSUPPORTED;
    return qt_fwd(a, (hipStream_t)stream);
}
int mopk_quartet_bwd(const MopkQuartetArgs *a, void *stream) {    // the fused path also reads `y` (the forward's output)
    if (!a) return MOPK_ERR_BAD_ARG;
    int rc = base_ok(a->B, a->H, a->T, a->dh, a->io_dtype, a->precision); if (rc) return rc;
    if (!v4ok(a->q) || !v4ok(a->k) || !v4ok(a->v) || !v4ok(a->y) || !v4ok(a->dy) || !v4ok(a->dq) || !v4ok(a->dk_) || !v4ok(a->dv) ||
        !a->saved || !a->workspace) return MOPK_ERR_BAD_ARG;
    if (a->use_quartet && (!v4ok(a->q2) || !v4ok(a->k2) || !v4ok(a->dq2) || !v4ok(a->dk2) || !a->dmixture_part || !a->dqscale_part ||
                           !a->mixture || !a->quartet_scale)) return MOPK_ERR_BAD_ARG;
    if (qt_use_flash(a)) return qt_flash_bwd(a, (hipStream_t)stream);
    if (a->path == MOPK_PATH_FUSED) return MOPK_ERR_UNSUPPORTED;
    return qt_bwd(a, (hipStream_t)stream);
}

}  // extern "C"
