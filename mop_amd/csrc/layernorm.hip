// LayerNorm prologue of the attention path: `ln1(x)` applied and cast to the dtype the qkv GEMM consumes in
// ONE pass over the residual stream, and its backward fused with the residual-branch add (`x + attn(ln1(x))`,
// experiments/cifar100_edgewise_gates.py:371-374 -- the same block shape as mop/models/components.py:96-98).
//
// HBM-bound row work: one 64-lane wave per token row, the row lives in registers (16-byte vector loads, lane l owns the 8-element
// chunks l, l+64, ...), mean / variance are the exact two-pass fp32 values (what torch.nn.LayerNorm computes), no LDS in the
// forward.  The backward keeps per-lane column sums of dgamma / dbeta in registers across the rows a persistent workgroup
// processes and leaves one partial row per workgroup; a second tiny kernel sums the partials in a fixed order (bitwise
// reproducible).  Algorithmic traffic: forward rows*dim*(sizeof x + sizeof y), backward rows*dim*(dy + x + dres + dx).
#include "common.h"

namespace mopk {

constexpr int LN_WAVES = 4;            // rows per workgroup pass
constexpr int LN_MAX_PARTS = 512;      // persistent workgroups of the backward (2 per CU)

template <typename T> __device__ __forceinline__ void ld8(const T *p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float *p, float (&v)[8]) {
    const float4 a = *(const float4 *)p, b = *(const float4 *)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<unsigned short>(const unsigned short *p, float (&v)[8]) {
    const uint4 u = *(const uint4 *)p;
    const unsigned int w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __builtin_bit_cast(float, w[i] << 16); v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ void st8(T *p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<float>(float *p, const float (&v)[8]) {
    *(float4 *)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4 *)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void st8<unsigned short>(unsigned short *p, const float (&v)[8]) {
    *(uint4 *)p = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}

template <typename XT, typename YT, typename PT, int VPL>
__global__ __launch_bounds__(LN_WAVES * 64) void ln_fwd_kernel(MopkLayerNormArgs a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * LN_WAVES + w;
    if (row >= a.rows) return;
    const int nvec = a.dim >> 3;
    const XT *xp = (const XT *)a.x + row * a.x_ld;
    float x[VPL][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) ld8<XT>(xp + 8 * c, x[i]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[i][e];
    }
    const float mean = wave_sum(s) / (float)a.dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        if (lane + 64 * i < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = x[i][e] - mean; q = fmaf(d, d, q); }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)a.dim + a.eps);
    if (lane == 0 && a.mean) { a.mean[row] = mean; a.rstd[row] = rstd; }
    YT *yp = (YT *)a.y + row * a.y_ld;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            float g[8], b[8], o[8];
            ld8<PT>((const PT *)a.gamma + 8 * c, g);
            if (a.beta) ld8<PT>((const PT *)a.beta + 8 * c, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf((x[i][e] - mean) * rstd, g[e], a.beta ? b[e] : 0.f);
            st8<YT>(yp + 8 * c, o);
        }
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ;  dgamma += dy * xhat ; dbeta += dy
template <typename XT, typename YT, typename PT, int VPL>
__global__ __launch_bounds__(LN_WAVES * 64) void ln_bwd_kernel(MopkLayerNormArgs a, float *part) {
    __shared__ float red[LN_WAVES][64 * 8 + 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nvec = a.dim >> 3;
    float dg[VPL][8], db[VPL][8], gam[VPL][8];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; gam[i][e] = 0.f; }
        if (c < nvec) ld8<PT>((const PT *)a.gamma + 8 * c, gam[i]);
    }
    const float invd = 1.0f / (float)a.dim;
    for (int64_t row = (int64_t)blockIdx.x * LN_WAVES + w; row < a.rows; row += (int64_t)gridDim.x * LN_WAVES) {
        const XT *xp = (const XT *)a.x + row * a.x_ld;
        const YT *dyp = (const YT *)a.dy + row * a.y_ld;
        const float mean = a.mean[row], rstd = a.rstd[row];
        float xh[VPL][8], g[VPL][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float xv[8], dv[8];
                ld8<XT>(xp + 8 * c, xv);
                ld8<YT>(dyp + 8 * c, dv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[i][e] = (xv[e] - mean) * rstd;
                    g[i][e] = dv[e] * gam[i][e];
                    s1 += g[i][e];
                    s2 = fmaf(g[i][e], xh[i][e], s2);
                    dg[i][e] = fmaf(dv[e], xh[i][e], dg[i][e]);
                    db[i][e] += dv[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[i][e] = 0.f; g[i][e] = 0.f; }
            }
        }
        s1 = wave_sum(s1) * invd;
        s2 = wave_sum(s2) * invd;
        XT *dxp = (XT *)a.dx + row * a.x_ld;
        const XT *drp = a.dres ? (const XT *)a.dres + row * a.x_ld : nullptr;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float o[8], rr[8];
                if (drp) ld8<XT>(drp + 8 * c, rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(rstd, g[i][e] - s1 - xh[i][e] * s2, drp ? rr[e] : 0.f);
                st8<XT>(dxp + 8 * c, o);
            }
        }
    }
    // column partials: sum the workgroup's waves through LDS, one partial row per workgroup  [part][2][dim]
    float *prow = part + (size_t)blockIdx.x * 2 * a.dim;
    for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[w][lane * 8 + e] = which ? db[i][e] : dg[i][e];
            __syncthreads();
            for (int c = threadIdx.x; c < 64 * 8; c += LN_WAVES * 64) {
                const int col = 64 * 8 * i + c;
                if (col < a.dim) {
                    float s = 0.f;
#pragma unroll
                    for (int ww = 0; ww < LN_WAVES; ++ww) s += red[ww][c];
                    prow[which * a.dim + col] = s;
                }
            }
        }
    }
}

// one block per 64 columns of [dgamma | dbeta]: 16 waves each sum every 16th partial row (independent, coalesced loads), then
// the 16 sums are added in a fixed order through LDS
__global__ __launch_bounds__(1024) void ln_reduce_kernel(const float *part, int nparts, int dim, float *dgamma, float *dbeta) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float acc = 0.f;
    if (col < 2 * dim)
        for (int p = s; p < nparts; p += 16) acc += part[(size_t)p * 2 * dim + col];
    red[s][c] = acc;
    __syncthreads();
    if (s == 0 && col < 2 * dim) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][c];
        if (col < dim) { if (dgamma) dgamma[col] = t; } else if (dbeta) dbeta[col - dim] = t;
    }
}

static int ln_parts(const MopkLayerNormArgs *a) {
    const int64_t wgs = (a->rows + LN_WAVES - 1) / LN_WAVES;
    return (int)(wgs < LN_MAX_PARTS ? wgs : LN_MAX_PARTS);
}
static bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }
static int ln_validate(const MopkLayerNormArgs *a, bool bwd) {
    if (!a) return MOPK_ERR_BAD_ARG;
    if (a->rows <= 0 || a->dim <= 0) return MOPK_ERR_BAD_SHAPE;
    if (a->dim % 8 != 0 || a->dim > 4096) return MOPK_ERR_UNSUPPORTED;
    for (int dt : {a->x_dtype, a->y_dtype, a->p_dtype}) if (dt != MOPK_F32 && dt != MOPK_BF16) return MOPK_ERR_BAD_ARG;
    if (!a->x || !a->gamma || a->x_ld < a->dim || a->y_ld < a->dim || a->x_ld % 8 || a->y_ld % 8) return MOPK_ERR_BAD_ARG;
    if (!al16(a->x) || !al16(a->gamma) || (a->beta && !al16(a->beta))) return MOPK_ERR_BAD_ARG;
    if (!bwd) {
        if (!a->y || !al16(a->y) || ((a->mean == nullptr) != (a->rstd == nullptr))) return MOPK_ERR_BAD_ARG;
    } else {
        if (!a->dy || !a->dx || !a->mean || !a->rstd || !a->workspace) return MOPK_ERR_BAD_ARG;
        if (!al16(a->dy) || !al16(a->dx) || (a->dres && !al16(a->dres))) return MOPK_ERR_BAD_ARG;
    }
    return MOPK_OK;
}

template <typename XT, typename YT, typename PT>
static void ln_launch(const MopkLayerNormArgs *a, bool bwd, hipStream_t st) {
    const int vpl = (a->dim / 8 + 63) / 64;
    const dim3 block(LN_WAVES * 64);
#define MOPK_LN(V_) do {                                                                                              \
        if (!bwd) hipLaunchKernelGGL((ln_fwd_kernel<XT, YT, PT, V_>), dim3((unsigned)((a->rows + LN_WAVES - 1) / LN_WAVES)), block, 0, st, *a); \
        else hipLaunchKernelGGL((ln_bwd_kernel<XT, YT, PT, V_>), dim3(ln_parts(a)), block, 0, st, *a, (float *)a->workspace);        \
    } while (0)
    switch (vpl) { case 1: MOPK_LN(1); break; case 2: MOPK_LN(2); break; case 3: MOPK_LN(3); break; case 4: MOPK_LN(4); break;
                   case 5: MOPK_LN(5); break; case 6: MOPK_LN(6); break; case 7: MOPK_LN(7); break; default: MOPK_LN(8); break; }
#undef MOPK_LN
}
template <typename XT, typename YT>
static void ln_dispatch_p(const MopkLayerNormArgs *a, bool bwd, hipStream_t st) {
    if (a->p_dtype == MOPK_BF16) ln_launch<XT, YT, unsigned short>(a, bwd, st); else ln_launch<XT, YT, float>(a, bwd, st);
}
static void ln_dispatch(const MopkLayerNormArgs *a, bool bwd, hipStream_t st) {
    if (a->x_dtype == MOPK_BF16) {
        if (a->y_dtype == MOPK_BF16) ln_dispatch_p<unsigned short, unsigned short>(a, bwd, st); else ln_dispatch_p<unsigned short, float>(a, bwd, st);
    } else {
        if (a->y_dtype == MOPK_BF16) ln_dispatch_p<float, unsigned short>(a, bwd, st); else ln_dispatch_p<float, float>(a, bwd, st);
    }
}

}  // namespace mopk

using namespace mopk;

extern "C" {

size_t mopk_layernorm_workspace_bytes(const MopkLayerNormArgs *a) {
    if (!a || a->rows <= 0 || a->dim <= 0) return 0;
    return (size_t)ln_parts(a) * 2 * a->dim * sizeof(float);
}
int mopk_layernorm_fwd(const MopkLayerNormArgs *a, void *stream) {
    const int rc = ln_validate(a, false);
    if (rc != MOPK_OK) return rc;
    ln_dispatch(a, false, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? MOPK_OK : MOPK_ERR_LAUNCH;
}
int mopk_layernorm_bwd(const MopkLayerNormArgs *a, void *stream) {
    const int rc = ln_validate(a, true);
    if (rc != MOPK_OK) return rc;
    ln_dispatch(a, true, (hipStream_t)stream);
    if (a->dgamma || a->dbeta) {
        const int n = 2 * a->dim;
        hipLaunchKernelGGL(ln_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const float *)a->workspace,
                           ln_parts(a), a->dim, a->dgamma, a->dbeta);
    }
    return hipGetLastError() == hipSuccess ? MOPK_OK : MOPK_ERR_LAUNCH;
}

}  // extern "C"
