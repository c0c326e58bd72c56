// Plain scaled-dot-product attention -- fused gfx950 kernels (bf16 MFMA, fp32 accumulate, no N x N map in HBM).
//
// Replaces the attention core of reference BaselineMSA (attention_variants.py:42-46), MSA (components.py:61-64) and
// MultiheadSelfAttention (whisper_mop.py:163-175) when there is no explicit mask / bias tensor (the causal flag is
// handled in-kernel); everything else runs the generic path (attn_generic.hip).
//
// Layout ("X layout", as in edgewise_fused.hip): a wave owns 32 queries; a 32x32 score tile is computed TRANSPOSED,
//   S^T[key, query] = K_tile . Q^T            (v_mfma_f32_32x32x16_bf16: A = K rows from LDS, B = q fragments in registers)
// so a lane holds one query and its registers run over keys: the online softmax is a per-lane register loop plus one
// exchange between the lane halves, and the packed probabilities are directly the B operand of
//   O^T[d, query] += V^T[d, key] . P^T[key, query]
// with V^T staged in LDS (key columns permuted to the accumulator's k order).  The backward is two kernels (dQ per
// query block, dK/dV per key block), each recomputing P from the saved row log-sum-exp: deterministic, no atomics.
#include "fused_common.h"

namespace mopk {

namespace {
constexpr int FA_NW = 4, FA_QB = 32 * FA_NW, FA_KT = 64, FA_LDT = FA_KT + 8;
constexpr float FA_NEG = -1e30f, FA_LOG2E = 1.4426950408889634f, FA_LN2 = 0.6931471805599453f;

__device__ __forceinline__ f32x16 fa_zero() { return f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ void fa_pack(bf16x8 &lo, bf16x8 &hi, const f32x16 &x) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (short)f2bf(x[j]); hi[j] = (short)f2bf(x[8 + j]); }
}
// row fragments of one token: 8 contiguous features at 16 s + 8 h, optionally scaled (rounded to bf16 once)
template <int DK, typename IOT>
__device__ __forceinline__ void fa_frags(bf16x8 (&f)[DK / 16], const IOT *row, bool ok, int h, float scale) {
#pragma unroll
    for (int s = 0; s < DK / 16; ++s) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) v = load8_bf16<IOT>(row + 16 * s + 8 * h);
        if (scale != 1.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(bf2f((unsigned short)v[j]) * scale);
        }
        f[s] = v;
    }
}
// stage a tile of TK tokens: row-major image rows[tok][DK+8] and/or transposed image cols[d][perm(tok)]
template <int DK, typename IOT, bool ROWS, bool COLS>
__device__ __forceinline__ void fa_stage(unsigned short *rows, unsigned short *cols, const IOT *base, int64_t sn, int t0, int N,
                                         float scale, int tid) {
    constexpr int CH = DK / 8, LDK = DK + 8;
    for (int c = tid; c < FA_KT * CH; c += FA_NW * 64) {
        const int j = c / CH, dc = c % CH;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 + j < N) v = load8_bf16<IOT>(base + (int64_t)(t0 + j) * sn + dc * 8);
        if (scale != 1.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (short)f2bf(bf2f((unsigned short)v[e]) * scale);
        }
        if (ROWS) *(bf16x8 *)&rows[j * LDK + dc * 8] = v;
        if (COLS) {
            const int col = (j & ~15) + kperm16(j & 15);
#pragma unroll
            for (int e = 0; e < 8; ++e) cols[(dc * 8 + e) * FA_LDT + col] = (unsigned short)v[e];
        }
    }
}
// 32x32 tile: sum_s A[(row0 + r)][16 s + 8 h ..] x B[s]   (A from a row-major LDS image with stride DK+8)
template <int DK>
__device__ __forceinline__ f32x16 fa_mm_rows(const unsigned short *img, int row0, int r, int h, const bf16x8 (&B)[DK / 16]) {
    f32x16 acc = fa_zero();
#pragma unroll
    for (int s = 0; s < DK / 16; ++s) {
        const bf16x8 af = *(const bf16x8 *)&img[(row0 + r) * (DK + 8) + 16 * s + 8 * h];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, B[s], acc, 0, 0, 0);
    }
    return acc;
}
// acc[dt] += T[d = 32 dt + r][32 half + ..] x packed tile (lo: k 0-15, hi: k 16-31 of this 32-token half)
template <int DK>
__device__ __forceinline__ void fa_mm_cols(f32x16 (&acc)[DK / 32], const unsigned short *timg, int half, int r, int h, bf16x8 lo, bf16x8 hi) {
#pragma unroll
    for (int dt = 0; dt < DK / 32; ++dt) {
        const unsigned short *p = &timg[(32 * dt + r) * FA_LDT + 32 * half + 8 * h];
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)p, lo, acc[dt], 0, 0, 0);
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(p + 16), hi, acc[dt], 0, 0, 0);
    }
}
// write a transposed accumulator (lane = token, registers = features) to a (.., token, feature) tensor row
template <int DK, typename IOT>
__device__ __forceinline__ void fa_store_rows(IOT *row, const f32x16 (&acc)[DK / 32], int h, float scale) {
#pragma unroll
    for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            store4<IOT>(row + 32 * dt + 8 * g4 + 4 * h, acc[dt][4 * g4] * scale, acc[dt][4 * g4 + 1] * scale, acc[dt][4 * g4 + 2] * scale,
                        acc[dt][4 * g4 + 3] * scale);
}
}  // namespace

// ------------------------------------------------------------------ forward
template <int DK, typename IOT, bool CAUSAL>
__global__ void __launch_bounds__(FA_NW * 64) sdpa_flash_fwd_kernel(MopkSdpaArgs a, float *lse) {
    constexpr int DT = DK / 32, LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[FA_KT * LDK], Vt[DK * FA_LDT];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.N, bh = blockIdx.y, b = bh / a.H, hh = bh % a.H;
    const int q0 = blockIdx.x * FA_QB, qi = q0 + 32 * w + r;
    const bool qok = qi < N;
    const float c = rsqrtf((float)DK) * FA_LOG2E;                  // logits in base-2 units: exp2 without a multiply
    const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh, *vp = (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh;
    bf16x8 qe[DK / 16];
    fa_frags<DK, IOT>(qe, (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn, qok, h, c);
    float m = FA_NEG, l = 0.f;
    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) O[dt] = fa_zero();
    int nkt = (N + FA_KT - 1) / FA_KT;
    if (CAUSAL) nkt = min(nkt, (min(q0 + FA_QB, N) + FA_KT - 1) / FA_KT);   // keys beyond the block's last query are never seen
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * FA_KT;
        __syncthreads();
        fa_stage<DK, IOT, true, false>(Ks, nullptr, kp, a.k.sn, k0, N, 1.f, tid);
        fa_stage<DK, IOT, false, true>(nullptr, Vt, vp, a.v.sn, k0, N, 1.f, tid);
        __syncthreads();
        f32x16 S[2];
        float mx = FA_NEG;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            S[s2] = fa_mm_rows<DK>(Ks, 32 * s2, r, h, qe);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int j = k0 + 32 * s2 + tile_row(g, h);
                if (j >= N || (CAUSAL && j > qi)) S[s2][g] = FA_NEG;
                mx = fmaxf(mx, S[s2][g]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx), alpha = __builtin_amdgcn_exp2f(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int g = 0; g < 16; ++g) { const float p = __builtin_amdgcn_exp2f(S[s2][g] - mn); S[s2][g] = p; ps += p; }
        ps += __shfl_xor(ps, 32, 64);
        l = fmaf(l, alpha, ps);
        m = mn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 16; ++g) O[dt][g] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 lo, hi;
            fa_pack(lo, hi, S[s2]);
            fa_mm_cols<DK>(O, Vt, s2, r, h, lo, hi);
        }
    }
    if (qok) {
        fa_store_rows<DK, IOT>((IOT *)a.y.ptr + b * a.y.sb + hh * a.y.sh + (int64_t)qi * a.y.sn, O, h, 1.f / l);
        if (h == 0) lse[(int64_t)bh * N + qi] = m + __builtin_amdgcn_logf(l);     // log2 of the row sum of 2^(logit)
    }
}

// ------------------------------------------------------------------ backward
// delta_i = sum_d dy[i,d] y[i,d]  ( = sum_j P_ij dP_ij ); one wave per row
template <typename IOT>
__global__ void sdpa_flash_delta_kernel(MopkSdpaArgs a, float *delta) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)a.B * a.H * a.N) return;
    const int lane = threadIdx.x & 63, i = row % a.N;
    const int64_t bh = row / a.N;
    const int b = bh / a.H, hh = bh % a.H;
    const IOT *yp = (const IOT *)a.y.ptr + b * a.y.sb + hh * a.y.sh + (int64_t)i * a.y.sn;
    const IOT *gp = (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh + (int64_t)i * a.dy.sn;
    float s = 0.f;
    for (int d = lane; d < a.dk; d += 64) s = fmaf(ld_as_f32(yp + d), ld_as_f32(gp + d), s);
    s = wave_sum(s);
    if (lane == 0) delta[row] = s;
}

// dQ: one workgroup per 128 queries, loop over key tiles
template <int DK, typename IOT, bool CAUSAL>
__global__ void __launch_bounds__(FA_NW * 64) sdpa_flash_dq_kernel(MopkSdpaArgs a, const float *lse, const float *delta) {
    constexpr int DT = DK / 32, LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[FA_KT * LDK], Vs[FA_KT * LDK], Kt[DK * FA_LDT];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.N, bh = blockIdx.y, b = bh / a.H, hh = bh % a.H;
    const int q0 = blockIdx.x * FA_QB, qi = q0 + 32 * w + r;
    const bool qok = qi < N;
    const float sc = rsqrtf((float)DK), c = sc * FA_LOG2E;
    const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh, *vp = (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh;
    bf16x8 qe[DK / 16], dof[DK / 16];
    fa_frags<DK, IOT>(qe, (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn, qok, h, c);
    fa_frags<DK, IOT>(dof, (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh + (int64_t)qi * a.dy.sn, qok, h, 1.f);
    const float Li = qok ? lse[(int64_t)bh * N + qi] : 0.f, di = qok ? delta[(int64_t)bh * N + qi] : 0.f;
    f32x16 dQ[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dQ[dt] = fa_zero();
    int nkt = (N + FA_KT - 1) / FA_KT;
    if (CAUSAL) nkt = min(nkt, (min(q0 + FA_QB, N) + FA_KT - 1) / FA_KT);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * FA_KT;
        __syncthreads();
        fa_stage<DK, IOT, true, true>(Ks, Kt, kp, a.k.sn, k0, N, 1.f, tid);
        fa_stage<DK, IOT, true, false>(Vs, nullptr, vp, a.v.sn, k0, N, 1.f, tid);
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const f32x16 S = fa_mm_rows<DK>(Ks, 32 * s2, r, h, qe);
            const f32x16 dP = fa_mm_rows<DK>(Vs, 32 * s2, r, h, dof);
            f32x16 dS;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int j = k0 + 32 * s2 + tile_row(g, h);
                const bool ok = qok && j < N && (!CAUSAL || j <= qi);
                const float p = ok ? __builtin_amdgcn_exp2f(S[g] - Li) : 0.f;
                dS[g] = p * (dP[g] - di) * sc;                    // d logits / sqrt(dk)
            }
            bf16x8 lo, hi;
            fa_pack(lo, hi, dS);
            fa_mm_cols<DK>(dQ, Kt, s2, r, h, lo, hi);
        }
    }
    if (qok) fa_store_rows<DK, IOT>((IOT *)a.dq.ptr + b * a.dq.sb + hh * a.dq.sh + (int64_t)qi * a.dq.sn, dQ, h, 1.f);
}

// dK, dV: one workgroup per 128 keys (a lane owns a key), loop over query tiles
template <int DK, typename IOT, bool CAUSAL>
__global__ void __launch_bounds__(FA_NW * 64) sdpa_flash_dkv_kernel(MopkSdpaArgs a, const float *lse, const float *delta) {
    constexpr int DT = DK / 32, LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Qs[FA_KT * LDK], Gs[FA_KT * LDK], Qt[DK * FA_LDT], Gt[DK * FA_LDT];
    __shared__ float Ls[FA_KT], Ds[FA_KT];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.N, bh = blockIdx.y, b = bh / a.H, hh = bh % a.H;
    const int k0 = blockIdx.x * FA_QB, kj = k0 + 32 * w + r;
    const bool kok = kj < N;
    const float c = rsqrtf((float)DK) * FA_LOG2E;
    const IOT *qp = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh, *gp = (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh;
    bf16x8 kf[DK / 16], vf[DK / 16];
    fa_frags<DK, IOT>(kf, (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh + (int64_t)kj * a.k.sn, kok, h, 1.f);
    fa_frags<DK, IOT>(vf, (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh + (int64_t)kj * a.v.sn, kok, h, 1.f);
    f32x16 dK[DT], dV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dK[dt] = fa_zero(); dV[dt] = fa_zero(); }
    const int nqt = (N + FA_KT - 1) / FA_KT;
    for (int qt = CAUSAL ? k0 / FA_KT : 0; qt < nqt; ++qt) {      // causal: queries before this key block see none of its keys
        const int i0 = qt * FA_KT;
        __syncthreads();
        fa_stage<DK, IOT, true, true>(Qs, Qt, qp, a.q.sn, i0, N, c, tid);       // q pre-scaled exactly as the forward's fragments
        fa_stage<DK, IOT, true, true>(Gs, Gt, gp, a.dy.sn, i0, N, 1.f, tid);
        if (tid < FA_KT) {
            const bool ok = i0 + tid < N;
            Ls[tid] = ok ? lse[(int64_t)bh * N + i0 + tid] : 0.f;
            Ds[tid] = ok ? delta[(int64_t)bh * N + i0 + tid] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const f32x16 S = fa_mm_rows<DK>(Qs, 32 * s2, r, h, kf);       // rows = queries, lane = key
            const f32x16 dP = fa_mm_rows<DK>(Gs, 32 * s2, r, h, vf);
            f32x16 P, dS;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int il = 32 * s2 + tile_row(g, h), i = i0 + il;
                const bool ok = kok && i < N && (!CAUSAL || kj <= i);
                const float p = ok ? __builtin_amdgcn_exp2f(S[g] - Ls[il]) : 0.f;
                P[g] = p;
                dS[g] = p * (dP[g] - Ds[il]) * FA_LN2;            // Q' = q log2(e)/sqrt(dk)  ->  dK = (dS ln2)^T Q'
            }
            bf16x8 lo, hi;
            fa_pack(lo, hi, P);
            fa_mm_cols<DK>(dV, Gt, s2, r, h, lo, hi);
            fa_pack(lo, hi, dS);
            fa_mm_cols<DK>(dK, Qt, s2, r, h, lo, hi);
        }
    }
    if (kok) {
        fa_store_rows<DK, IOT>((IOT *)a.dk_.ptr + b * a.dk_.sb + hh * a.dk_.sh + (int64_t)kj * a.dk_.sn, dK, h, 1.f);
        fa_store_rows<DK, IOT>((IOT *)a.dv.ptr + b * a.dv.sb + hh * a.dv.sh + (int64_t)kj * a.dv.sn, dV, h, 1.f);
    }
}

// ------------------------------------------------------------------ host side
static bool fa_aligned(const MopkView4 &v, int es) {
    const int64_t al = 16 / es;
    return ((uintptr_t)v.ptr & 15) == 0 && v.sb % al == 0 && v.sh % al == 0 && v.sn % al == 0;
}
int sdpa_flash_supported(const MopkSdpaArgs *a, bool bwd) {
    if (a->precision != MOPK_PREC_BF16) return 0;                 // fp32-exact arithmetic stays on the generic path
    if (a->mask || a->bias) return 0;                             // explicit mask / bias tensors: generic path
    if (a->dk != 32 && a->dk != 64) return 0;
    const int es = a->io_dtype == MOPK_BF16 ? 2 : 4;
    if (!fa_aligned(a->q, es) || !fa_aligned(a->k, es) || !fa_aligned(a->v, es) || !fa_aligned(a->y, es)) return 0;
    if (bwd && (!fa_aligned(a->dy, es) || !fa_aligned(a->dq, es) || !fa_aligned(a->dk_, es) || !fa_aligned(a->dv, es))) return 0;
    return 1;
}
size_t sdpa_flash_saved_bytes(const MopkSdpaArgs *a) { return (size_t)a->B * a->H * a->N * sizeof(float) + 256; }   // row log-sum-exp
size_t sdpa_flash_ws_bytes(const MopkSdpaArgs *a) { return (size_t)a->B * a->H * a->N * sizeof(float) + 256; }      // delta

#define FA_DISPATCH(KERNEL, GRID, ...)                                                                         \
    do {                                                                                                       \
        const dim3 blk(FA_NW * 64);                                                                            \
        if (a->io_dtype == MOPK_BF16) {                                                                        \
            if (a->dk == 64) { if (a->causal) hipLaunchKernelGGL((KERNEL<64, unsigned short, true>), GRID, blk, 0, st, __VA_ARGS__);   \
                               else hipLaunchKernelGGL((KERNEL<64, unsigned short, false>), GRID, blk, 0, st, __VA_ARGS__); }          \
            else { if (a->causal) hipLaunchKernelGGL((KERNEL<32, unsigned short, true>), GRID, blk, 0, st, __VA_ARGS__);              \
                   else hipLaunchKernelGGL((KERNEL<32, unsigned short, false>), GRID, blk, 0, st, __VA_ARGS__); }                     \
        } else {                                                                                               \
            if (a->dk == 64) { if (a->causal) hipLaunchKernelGGL((KERNEL<64, float, true>), GRID, blk, 0, st, __VA_ARGS__);            \
                               else hipLaunchKernelGGL((KERNEL<64, float, false>), GRID, blk, 0, st, __VA_ARGS__); }                   \
            else { if (a->causal) hipLaunchKernelGGL((KERNEL<32, float, true>), GRID, blk, 0, st, __VA_ARGS__);                       \
                   else hipLaunchKernelGGL((KERNEL<32, float, false>), GRID, blk, 0, st, __VA_ARGS__); }                              \
        }                                                                                                      \
    } while (0)

int sdpa_flash_fwd(const MopkSdpaArgs *a, hipStream_t st) {
    if (!sdpa_flash_supported(a, false)) return MOPK_ERR_UNSUPPORTED;
    const dim3 grid((a->N + FA_QB - 1) / FA_QB, a->B * a->H);
    FA_DISPATCH(sdpa_flash_fwd_kernel, grid, *a, (float *)a->saved);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}
int sdpa_flash_bwd(const MopkSdpaArgs *a, hipStream_t st) {
    if (!sdpa_flash_supported(a, true)) return MOPK_ERR_UNSUPPORTED;
    const int64_t rows = (int64_t)a->B * a->H * a->N;
    float *delta = (float *)a->workspace;
    if (a->io_dtype == MOPK_BF16) hipLaunchKernelGGL((sdpa_flash_delta_kernel<unsigned short>), dim3((rows + 3) / 4), dim3(256), 0, st, *a, delta);
    else hipLaunchKernelGGL((sdpa_flash_delta_kernel<float>), dim3((rows + 3) / 4), dim3(256), 0, st, *a, delta);
    MOPK_CHECK_LAUNCH();
    const dim3 grid((a->N + FA_QB - 1) / FA_QB, a->B * a->H);
    FA_DISPATCH(sdpa_flash_dq_kernel, grid, *a, (const float *)a->saved, (const float *)delta);
    MOPK_CHECK_LAUNCH();
    FA_DISPATCH(sdpa_flash_dkv_kernel, grid, *a, (const float *)a->saved, (const float *)delta);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

}  // namespace mopk
