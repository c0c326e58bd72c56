// extern "C" entry points of libmopk.so (see include/mopk.h).
#include "common.h"

namespace mopk {
size_t ew_generic_saved_bytes(const MopkEdgewiseArgs *a);
size_t ew_generic_workspace_bytes(const MopkEdgewiseArgs *a);
int ew_generic_fwd(const MopkEdgewiseArgs *a, hipStream_t st);
int ew_generic_bwd(const MopkEdgewiseArgs *a, hipStream_t st);
int ew_fused_fwd_supported(const MopkEdgewiseArgs *a);
int ew_fused_fwd(const MopkEdgewiseArgs *a, hipStream_t st);

static int ew_validate(const MopkEdgewiseArgs *a, bool bwd) {
    if (!a) return MOPK_ERR_BAD_ARG;
    if (a->B <= 0 || a->H <= 0 || a->N <= 0 || a->dk <= 0 || a->V < 2 || a->r < 1) return MOPK_ERR_BAD_SHAPE;
    if (a->io_dtype != MOPK_F32 && a->io_dtype != MOPK_BF16) return MOPK_ERR_BAD_ARG;
    if (a->precision != MOPK_PREC_FP32 && a->precision != MOPK_PREC_BF16) return MOPK_ERR_BAD_ARG;
    if (!a->q.ptr || !a->k.ptr || !a->v0.ptr || !a->vL.ptr || !a->sqk || !a->vs0 || !a->vsL || !a->Wr || !a->br ||
        !a->Wc || !a->bc || !a->chain_logit || !a->saved || !a->workspace)
        return MOPK_ERR_BAD_ARG;
    if (!bwd && !a->y.ptr) return MOPK_ERR_BAD_ARG;
    if (bwd && (!a->dy.ptr || !a->dq.ptr || !a->dk_.ptr || !a->dv0.ptr || !a->dvL.ptr || !a->dsqk_part ||
                !a->dvs0_part || !a->dvsL_part || !a->dWr || !a->dbr || !a->dWc || !a->dbc || !a->dlogit_part))
        return MOPK_ERR_BAD_ARG;
    if ((a->q.sv == 0) != (a->k.sv == 0)) return MOPK_ERR_BAD_ARG;
    if (bwd && ((a->dq.sv == 0) != (a->q.sv == 0) || (a->dk_.sv == 0) != (a->k.sv == 0))) return MOPK_ERR_BAD_ARG;
    return MOPK_OK;
}
}  // namespace mopk

using namespace mopk;

extern "C" {

int mopk_version(void) { return MOPK_VERSION; }

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
    (void)a; (void)backward;
    return "bgemm_kernel";
}

size_t mopk_edgewise_saved_bytes(const MopkEdgewiseArgs *a) {
    if (!a || a->B <= 0 || a->H <= 0 || a->N <= 0 || a->dk <= 0 || a->V < 2 || a->r < 1) return 0;
    if (a->path == MOPK_PATH_FUSED) return (size_t)a->B * a->H * a->N * a->dk * sizeof(float) + 256;  // w * y_chain
    return ew_generic_saved_bytes(a);
}
size_t mopk_edgewise_workspace_bytes(const MopkEdgewiseArgs *a) {
    if (!a || a->B <= 0 || a->H <= 0 || a->N <= 0 || a->dk <= 0 || a->V < 2 || a->r < 1) return 0;
    if (a->path == MOPK_PATH_FUSED) return 256;
    return ew_generic_workspace_bytes(a);
}
int mopk_edgewise_lowrank_fwd(const MopkEdgewiseArgs *a, void *stream) {
    int rc = ew_validate(a, false);
    if (rc) return rc;
    if (a->path == MOPK_PATH_FUSED) return ew_fused_fwd(a, (hipStream_t)stream);
    return ew_generic_fwd(a, (hipStream_t)stream);
}
int mopk_edgewise_lowrank_bwd(const MopkEdgewiseArgs *a, void *stream) {
    int rc = ew_validate(a, true);
    if (rc) return rc;
    if (a->path == MOPK_PATH_FUSED) return MOPK_ERR_UNSUPPORTED;
    return ew_generic_bwd(a, (hipStream_t)stream);
}

// ---- not yet implemented cores: report UNSUPPORTED (callers fail loudly) ----
size_t mopk_dualpath_saved_bytes(const MopkDualPathArgs *) { return 0; }
size_t mopk_dualpath_workspace_bytes(const MopkDualPathArgs *) { return 0; }
int mopk_dualpath_fwd(const MopkDualPathArgs *, void *) { return MOPK_ERR_UNSUPPORTED; }
int mopk_dualpath_bwd(const MopkDualPathArgs *, void *) { return MOPK_ERR_UNSUPPORTED; }
size_t mopk_quartet_saved_bytes(const MopkQuartetArgs *) { return 0; }
size_t mopk_quartet_workspace_bytes(const MopkQuartetArgs *) { return 0; }
int mopk_quartet_fwd(const MopkQuartetArgs *, void *) { return MOPK_ERR_UNSUPPORTED; }
int mopk_quartet_bwd(const MopkQuartetArgs *, void *) { return MOPK_ERR_UNSUPPORTED; }
size_t mopk_sdpa_saved_bytes(const MopkSdpaArgs *) { return 0; }
size_t mopk_sdpa_workspace_bytes(const MopkSdpaArgs *) { return 0; }
int mopk_sdpa_fwd(const MopkSdpaArgs *, void *) { return MOPK_ERR_UNSUPPORTED; }
int mopk_sdpa_bwd(const MopkSdpaArgs *, void *) { return MOPK_ERR_UNSUPPORTED; }

}  // extern "C"
