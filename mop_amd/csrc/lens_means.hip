// Row / column means of the S lens bank's planes in closed form (reference attention_variants.py:425-442, :523-533 followed by the low-rank
// head's means :323-326), forward and backward.  The bank convolves every score plane S_v = (q * sqk_v) k^T with a depthwise dilated 3x3
// kernel w (zero padding = dilation d); the head reads only the row / column means of the result, and those are linear in q and k:
//     row[l,v][i] = sum_a  q[i + (a-1) d] . ua[l,v,a],    ua[l,v,a][c] = sqk_v[c] / N  sum_b w[l,v,a,b] ks[l,b][c],   ks[l,b] = sum_{j in J_b} k[j]
//     col[l,v][j] = sum_b  k[j + (b-1) d] . ub[l,v,b],    ub[l,v,b][c] = sqk_v[c] / N  sum_a w[l,v,a,b] qs[l,a][c],   qs[l,a] = sum_{i in J_a} q[i]
// with J_0 = [0, N-d), J_1 = [0, N), J_2 = [d, N): the source rows / columns a tap reaches.  O(N dk) per (b, h, view), no plane.
// One workgroup per (b, h); q and k live in LDS as bf16 (the values the fused Edgewise kernels multiply), everything else in fp32.
// The outputs feed the fused Edgewise kernels as extra feature channels (MopkEdgewiseExt.n_extra); mop_amd/ops.py holds the same
// algebra in torch ops (lens_mean_features) -- the CPU-tested statement these kernels are checked against.
#include "common.h"

namespace mopk {
namespace {

constexpr int LM_THREADS = 512;       // 8 waves: the loops below are chains of dependent LDS reads, two waves per SIMD hide each other's
constexpr int LM_MAXN = 224, LM_MAXE = 16;      // what the fused Edgewise kernels take: N <= 224, 2V + 2 + L V <= 26

struct LmSmem {
    unsigned short *q, *k;     // [N][LDK] bf16
    float *ks, *qs;            // [L][3][DK]
    float *ua, *ub;            // [E][3][DK]
    float *g;                  // backward: d_row | d_col  [2][E][GS], each row zero padded by GP = max dilation on both sides
    float *dua, *dub;          // backward: [E][3][DK]
    float *dks, *dqs;          // backward: [L][3][DK]
};
template <int DK> __host__ __device__ constexpr int lm_ldk() { return DK + 8; }
__host__ __device__ inline int lm_gpad(const MopkLensMeansArgs &a) {            // largest dilation, clamped to N (a tap further out reaches nothing)
    int m = 1;
    for (int l = 0; l < a.L; ++l) m = a.dil[l] > m ? a.dil[l] : m;
    return m < a.N ? m : a.N;
}
__host__ __device__ inline int lm_gstride(int N, int gp) { return (N + 2 * gp + 3) & ~3; }
template <int DK> __host__ __device__ inline size_t lm_smem_bytes(int N, int L, int E, bool bwd, int gs) {
    size_t b = (size_t)2 * N * lm_ldk<DK>() * 2;
    b = (b + 15) & ~(size_t)15;
    b += (size_t)(2 * L * 3 * DK + 2 * E * 3 * DK) * 4;
    if (bwd) b += (size_t)(2 * E * gs + 2 * E * 3 * DK + 2 * L * 3 * DK) * 4;
    return b;
}
template <int DK> __device__ inline LmSmem lm_carve(unsigned char *smem, int N, int L, int E, int gs) {
    LmSmem s;
    s.q = (unsigned short *)smem;
    s.k = s.q + N * lm_ldk<DK>();
    size_t o = ((size_t)2 * N * lm_ldk<DK>() * 2 + 15) & ~(size_t)15;
    float *f = (float *)(smem + o);
    s.ks = f; f += L * 3 * DK;
    s.qs = f; f += L * 3 * DK;
    s.ua = f; f += E * 3 * DK;
    s.ub = f; f += E * 3 * DK;
    s.g = f; f += 2 * E * gs;
    s.dua = f; f += E * 3 * DK;
    s.dub = f; f += E * 3 * DK;
    s.dks = f; f += L * 3 * DK;
    s.dqs = f;
    return s;
}

// q, k rows of this (b, h) -> LDS (bf16), then the token sums over J_0 / J_1 / J_2 per dilation and the folded operands ua, ub
template <int DK, typename IOT>
__device__ void lm_stage(const MopkLensMeansArgs &a, const LmSmem &s, int b, int hh) {
    constexpr int LDK = lm_ldk<DK>(), CH = DK / 8;
    const int tid = threadIdx.x, N = a.N, V = a.V, L = a.L;
    const IOT *qp = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh, *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh;
    for (int c = tid; c < N * CH; c += LM_THREADS) {
        const int n = c / CH, dc = c % CH;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s.q[n * LDK + dc * 8 + e] = f2bf(ld_as_f32<IOT>(qp + (int64_t)n * a.q.sn + dc * 8 + e));
            s.k[n * LDK + dc * 8 + e] = f2bf(ld_as_f32<IOT>(kp + (int64_t)n * a.k.sn + dc * 8 + e));
        }
    }
    __syncthreads();
    // thread = (tensor, channel): total, and per dilation the sums of the first / last d rows (J_0 = all - last d, J_2 = all - first d)
    if (tid < 2 * DK) {
        const int c = tid % DK;
        const unsigned short *x = tid < DK ? s.k : s.q;
        float *out = tid < DK ? s.ks : s.qs;
        float t4[4] = {0.f, 0.f, 0.f, 0.f};
        int n0 = 0;
        for (; n0 + 4 <= N; n0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) t4[u] += bf2f(x[(n0 + u) * LDK + c]);
        }
        for (; n0 < N; ++n0) t4[0] += bf2f(x[n0 * LDK + c]);
        const float tot = (t4[0] + t4[1]) + (t4[2] + t4[3]);
        for (int l = 0; l < L; ++l) {
            const int d = a.dil[l] < N ? a.dil[l] : N;
            float head = 0.f, tail = 0.f;
            for (int n = 0; n < d; ++n) { head += bf2f(x[n * LDK + c]); tail += bf2f(x[(N - 1 - n) * LDK + c]); }
            out[(l * 3 + 0) * DK + c] = tot - tail;
            out[(l * 3 + 1) * DK + c] = tot;
            out[(l * 3 + 2) * DK + c] = tot - head;
        }
    }
    __syncthreads();
    const float invN = 1.f / (float)N;
    for (int idx = tid; idx < L * V * 3 * DK; idx += LM_THREADS) {          // idx = ((l V + v) 3 + t) DK + c; t = a (row side) / b (col side)
        const int c = idx % DK, t = (idx / DK) % 3, e = idx / (3 * DK), l = e / V, v = e % V;
        const float *w = a.lens_w + (size_t)e * 9;                           // (L,V,3,3) [a][b]
        const float sq = a.sqk[((size_t)v * a.H + hh) * DK + c] * invN;
        float ra = 0.f, cb = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) { ra = fmaf(w[t * 3 + u], s.ks[(l * 3 + u) * DK + c], ra); cb = fmaf(w[u * 3 + t], s.qs[(l * 3 + u) * DK + c], cb); }
        s.ua[idx] = sq * ra;
        s.ub[idx] = sq * cb;
    }
    __syncthreads();
}

template <int DK> __device__ __forceinline__ float lm_dot(const unsigned short *row, const float *u) {
    float acc = 0.f;
#pragma unroll
    for (int c8 = 0; c8 < DK / 8; ++c8) {
        const bf16x8 x = *(const bf16x8 *)(row + 8 * c8);
        const float4 u0 = *(const float4 *)(u + 8 * c8), u1 = *(const float4 *)(u + 8 * c8 + 4);
        acc = fmaf(bf2f((unsigned short)x[0]), u0.x, acc); acc = fmaf(bf2f((unsigned short)x[1]), u0.y, acc);
        acc = fmaf(bf2f((unsigned short)x[2]), u0.z, acc); acc = fmaf(bf2f((unsigned short)x[3]), u0.w, acc);
        acc = fmaf(bf2f((unsigned short)x[4]), u1.x, acc); acc = fmaf(bf2f((unsigned short)x[5]), u1.y, acc);
        acc = fmaf(bf2f((unsigned short)x[6]), u1.z, acc); acc = fmaf(bf2f((unsigned short)x[7]), u1.w, acc);
    }
    return acc;
}

template <int DK, typename IOT>
__global__ void __launch_bounds__(LM_THREADS) lens_means_fwd_kernel(MopkLensMeansArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LDK = lm_ldk<DK>();
    const int b = blockIdx.x / a.H, hh = blockIdx.x % a.H, N = a.N, V = a.V, E = a.L * a.V;
    const LmSmem s = lm_carve<DK>(smem, N, a.L, E, 0);
    lm_stage<DK, IOT>(a, s, b, hh);
    // thread = (token, side): the E channel means of its token; the three shifted rows come straight from the LDS rows
    for (int item = threadIdx.x; item < 2 * N; item += LM_THREADS) {
        const int side = item / N, n = item % N;
        const unsigned short *x = side ? s.k : s.q;
        const float *u = side ? s.ub : s.ua;
        float *out = (side ? a.col : a.row) + (size_t)blockIdx.x * E * N + n;
        for (int e = 0; e < E; ++e) {
            const int d = a.dil[e / V] < N ? a.dil[e / V] : N;
            float acc = lm_dot<DK>(x + n * LDK, u + (e * 3 + 1) * DK);
            if (n - d >= 0) acc += lm_dot<DK>(x + (n - d) * LDK, u + (e * 3 + 0) * DK);
            if (n + d < N) acc += lm_dot<DK>(x + (n + d) * LDK, u + (e * 3 + 2) * DK);
            out[(size_t)e * N] = acc;
        }
    }
}

// Backward.  With g_r = d_row, g_c = d_col:   dT_r[l,v,a][i'] = g_r[l,v][i' - (a-1) d],  dT_c[l,v,b][j'] = g_c[l,v][j' - (b-1) d]   (0 outside)
//   dq[i'] = sum dT_r[.][i'] ua[.]  + set-sum adjoint of dqs ;  dk[j'] = sum dT_c[.][j'] ub[.] + set-sum adjoint of dks
//   dua[l,v,a] = sum_i' dT_r[l,v,a][i'] q[i'] ;  dub[l,v,b] = sum_j' dT_c[l,v,b][j'] k[j']
//   dks[l,b] = sum_{v,a} w[a][b] sqk_v / N * dua[l,v,a] ;  dqs[l,a] = sum_{v,b} w[a][b] sqk_v / N * dub[l,v,b]
//   dsqk_v = sum_{l,a,b} w[a][b] / N (dua[l,v,a] * ks[l,b] + dub[l,v,b] * qs[l,a]) ;  dw[l,v,a,b] = 1/N sum_c sqk_v (dua ks + dub qs)
template <int DK, typename IOT>
__global__ void __launch_bounds__(LM_THREADS) lens_means_bwd_kernel(MopkLensMeansArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LDK = lm_ldk<DK>();
    const int tid = threadIdx.x, b = blockIdx.x / a.H, hh = blockIdx.x % a.H, N = a.N, V = a.V, L = a.L, E = L * V;
    const int GP = lm_gpad(a), GS = lm_gstride(N, GP);
    const LmSmem s = lm_carve<DK>(smem, N, L, E, GS);
    for (int idx = tid; idx < 2 * E * GS; idx += LM_THREADS) {          // upstream gradients -> LDS rows [pad | N values | pad]: shifted reads need no bounds
        const int row = idx / GS, n = idx % GS - GP, side = row / E, e = row % E;
        s.g[idx] = (n >= 0 && n < N) ? (side ? a.d_col : a.d_row)[((size_t)blockIdx.x * E + e) * N + n] : 0.f;
    }
    lm_stage<DK, IOT>(a, s, b, hh);
    const float invN = 1.f / (float)N;
    // dua / dub [rho = (e, t)][c] = sum_n g[e][n - (t-1) d] x[n][c]: thread = (channel pair, rho, side), four independent partial sums per channel
    for (int item = tid; item < 2 * E * 3 * (DK / 2); item += LM_THREADS) {
        const int cp = item % (DK / 2), rho = (item / (DK / 2)) % (E * 3), side = item / ((DK / 2) * E * 3);
        const int e = rho / 3, t = rho % 3, d = a.dil[e / V] < N ? a.dil[e / V] : N;
        const unsigned short *x = (side ? s.k : s.q) + 2 * cp;
        const float *g = s.g + (side * E + e) * GS + GP - (t - 1) * d;          // g[n] = upstream gradient at n - (t-1) d (0 outside)
        float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
        int n = 0;
        for (; n + 4 <= N; n += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned int xv = *(const unsigned int *)(x + (n + u) * LDK);
                const float gv = g[n + u];
                a0[u] = fmaf(gv, __builtin_bit_cast(float, xv << 16), a0[u]);
                a1[u] = fmaf(gv, __builtin_bit_cast(float, xv & 0xffff0000u), a1[u]);
            }
        }
        for (; n < N; ++n) {
            const unsigned int xv = *(const unsigned int *)(x + n * LDK);
            a0[0] = fmaf(g[n], __builtin_bit_cast(float, xv << 16), a0[0]);
            a1[0] = fmaf(g[n], __builtin_bit_cast(float, xv & 0xffff0000u), a1[0]);
        }
        float *du = (side ? s.dub : s.dua) + rho * DK + 2 * cp;
        du[0] = (a0[0] + a0[1]) + (a0[2] + a0[3]);
        du[1] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    }
    __syncthreads();
    // dks / dqs, dsqk partial of this (b, h), dlens partial
    for (int idx = tid; idx < 2 * L * 3 * DK; idx += LM_THREADS) {
        const int c = idx % DK, u = (idx / DK) % 3, l = (idx / (3 * DK)) % L, side = idx / (L * 3 * DK);     // side 0: dks[l][b = u], 1: dqs[l][a = u]
        float acc = 0.f;
        for (int v = 0; v < V; ++v) {
            const float *w = a.lens_w + (size_t)(l * V + v) * 9;
            const float sq = a.sqk[((size_t)v * a.H + hh) * DK + c] * invN;
            const float *du = (side ? s.dub : s.dua) + (size_t)(l * V + v) * 3 * DK + c;
#pragma unroll
            for (int t = 0; t < 3; ++t) acc = fmaf((side ? w[u * 3 + t] : w[t * 3 + u]) * sq, du[t * DK], acc);
        }
        (side ? s.dqs : s.dks)[(l * 3 + u) * DK + c] = acc;
    }
    for (int idx = tid; idx < V * DK; idx += LM_THREADS) {
        const int c = idx % DK, v = idx / DK;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) {
            const float *w = a.lens_w + (size_t)(l * V + v) * 9;
            const float *dua = s.dua + (size_t)(l * V + v) * 3 * DK + c, *dub = s.dub + (size_t)(l * V + v) * 3 * DK + c;
#pragma unroll
            for (int ta = 0; ta < 3; ++ta)
#pragma unroll
                for (int tb = 0; tb < 3; ++tb)
                    acc = fmaf(w[ta * 3 + tb], dua[ta * DK] * s.ks[(l * 3 + tb) * DK + c] + dub[tb * DK] * s.qs[(l * 3 + ta) * DK + c], acc);
        }
        a.dsqk_part[(((size_t)b * V + v) * a.H + hh) * DK + c] = acc * invN;
    }
    for (int idx = tid; idx < E * 9; idx += LM_THREADS) {
        const int e = idx / 9, ta = (idx % 9) / 3, tb = idx % 3, l = e / V, v = e % V;
        const float *sq = a.sqk + ((size_t)v * a.H + hh) * DK;
        float acc = 0.f;
#pragma unroll 8
        for (int c = 0; c < DK; ++c)
            acc = fmaf(sq[c], s.dua[(e * 3 + ta) * DK + c] * s.ks[(l * 3 + tb) * DK + c] + s.dub[(e * 3 + tb) * DK + c] * s.qs[(l * 3 + ta) * DK + c], acc);
        a.dlens_part[(size_t)blockIdx.x * E * 9 + idx] = acc * invN;
    }
    __syncthreads();
    // dq / dk rows, ADDED to what the caller's buffers hold (the Edgewise backward wrote its own dq / dk there): thread = (token, side, quarter of dk)
    constexpr int QW = DK / 4;             // channels per thread
    for (int item = tid; item < 2 * N * 4; item += LM_THREADS) {
        const int qd = item % 4, n = (item / 4) % N, side = item / (4 * N);
        const float *u = (side ? s.ub : s.ua) + qd * QW, *g = s.g + side * E * GS + GP + n, *ds = (side ? s.dks : s.dqs) + qd * QW;
        float acc[QW];
#pragma unroll
        for (int c = 0; c < QW; ++c) acc[c] = 0.f;
        for (int l = 0; l < L; ++l) {
            const int d = a.dil[l] < N ? a.dil[l] : N;
            const float in0 = n < N - d ? 1.f : 0.f, in2 = n >= d ? 1.f : 0.f;
#pragma unroll
            for (int c = 0; c < QW; ++c) acc[c] += fmaf(in0, ds[(l * 3 + 0) * DK + c], fmaf(in2, ds[(l * 3 + 2) * DK + c], ds[(l * 3 + 1) * DK + c]));
            for (int v = 0; v < V; ++v) {
                const int e = l * V + v;
                const float c0 = g[e * GS + d], c1 = g[e * GS], c2 = g[e * GS - d];     // dT[e][t][n] = g[e][n - (t-1) d]
                const float *u0 = u + (e * 3 + 0) * DK, *u1 = u0 + DK, *u2 = u1 + DK;
#pragma unroll
                for (int c = 0; c < QW; ++c) acc[c] = fmaf(c0, u0[c], fmaf(c1, u1[c], fmaf(c2, u2[c], acc[c])));
            }
        }
        const MopkView4 &dv = side ? a.dk_ : a.dq;
        IOT *dst = (IOT *)dv.ptr + b * dv.sb + hh * dv.sh + (int64_t)n * dv.sn + qd * QW;
        if constexpr (sizeof(IOT) == 2 && QW % 8 == 0) {               // bf16 rows, 16-byte aligned (checked on the host): 16-byte read-modify-write
#pragma unroll
            for (int c8 = 0; c8 < QW / 8; ++c8) {
                bf16x8 x = *(const bf16x8 *)(dst + 8 * c8);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (short)f2bf(bf2f((unsigned short)x[e]) + acc[8 * c8 + e]);
                *(bf16x8 *)(dst + 8 * c8) = x;
            }
        } else {
#pragma unroll
            for (int c = 0; c < QW; ++c) st_from_f32<IOT>(dst + c, ld_as_f32<IOT>(dst + c) + acc[c]);
        }
    }
}

template <int DK> int lm_launch(const MopkLensMeansArgs *a, bool bwd, hipStream_t st) {
    const int E = a->L * a->V;
    const size_t lds = lm_smem_bytes<DK>(a->N, a->L, E, bwd, lm_gstride(a->N, lm_gpad(*a)));
    if (lds > 160 * 1024) return MOPK_ERR_UNSUPPORTED;
    const dim3 grid(a->B * a->H), block(LM_THREADS);
#define LM_GO(KERNEL, IOT_) do {                                                                                                   \
        auto kfn = KERNEL<DK, IOT_>;                                                                                               \
        static size_t lds_set = 0;      /* sticky per instantiation; may not be set during stream capture */                       \
        if (lds_set < lds) { if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MOPK_ERR_LAUNCH; lds_set = lds; } \
        hipLaunchKernelGGL(kfn, grid, block, lds, st, *a);                                                                          \
    } while (0)
    if (bwd) { if (a->io_dtype == MOPK_BF16) LM_GO(lens_means_bwd_kernel, unsigned short); else LM_GO(lens_means_bwd_kernel, float); }
    else { if (a->io_dtype == MOPK_BF16) LM_GO(lens_means_fwd_kernel, unsigned short); else LM_GO(lens_means_fwd_kernel, float); }
#undef LM_GO
    return hipGetLastError() == hipSuccess ? MOPK_OK : MOPK_ERR_LAUNCH;
}

}  // namespace

// shapes the kernels take (no pointer is looked at): N <= 224, dk 16 / 32 / 64, L V <= 16, and the working set within the CU's LDS
int lens_means_supported(const MopkLensMeansArgs *a, bool bwd) {
    if (!a || a->B <= 0 || a->H <= 0 || a->N <= 0 || a->V < 1 || a->L < 1 || a->L > MOPK_MAX_LENS) return 0;
    if (a->N > LM_MAXN || a->L * a->V > LM_MAXE || (a->dk != 16 && a->dk != 32 && a->dk != 64)) return 0;
    for (int l = 0; l < a->L; ++l) if (a->dil[l] < 1) return 0;
    const int E = a->L * a->V, gs = lm_gstride(a->N, lm_gpad(*a));
    const size_t lds = a->dk == 16 ? lm_smem_bytes<16>(a->N, a->L, E, bwd, gs) : a->dk == 32 ? lm_smem_bytes<32>(a->N, a->L, E, bwd, gs)
                                                                                               : lm_smem_bytes<64>(a->N, a->L, E, bwd, gs);
    return lds <= 160 * 1024;
}

int lens_means_run(const MopkLensMeansArgs *a, bool bwd, hipStream_t st) {
    if (!a) return MOPK_ERR_BAD_ARG;
    if (a->B <= 0 || a->H <= 0 || a->N <= 0 || a->V < 1 || a->L < 1 || a->L > MOPK_MAX_LENS) return MOPK_ERR_BAD_SHAPE;
    if (a->N > LM_MAXN || a->L * a->V > LM_MAXE || (a->dk != 16 && a->dk != 32 && a->dk != 64)) return MOPK_ERR_UNSUPPORTED;
    if (a->io_dtype != MOPK_F32 && a->io_dtype != MOPK_BF16) return MOPK_ERR_BAD_ARG;
    for (int l = 0; l < a->L; ++l) if (a->dil[l] < 1) return MOPK_ERR_BAD_ARG;
    if (!a->q.ptr || !a->k.ptr || !a->sqk || !a->lens_w) return MOPK_ERR_BAD_ARG;
    if (!bwd && (!a->row || !a->col)) return MOPK_ERR_BAD_ARG;
    if (bwd && (!a->d_row || !a->d_col || !a->dq.ptr || !a->dk_.ptr || !a->dsqk_part || !a->dlens_part)) return MOPK_ERR_BAD_ARG;
    if (bwd && a->io_dtype == MOPK_BF16 && a->dk >= 32) {       // the 16-byte read-modify-write of dq / dk rows
        const MopkView4 *vs[2] = {&a->dq, &a->dk_};
        for (const MopkView4 *v : vs) if (((uintptr_t)v->ptr & 15) || v->sb % 8 || v->sh % 8 || v->sn % 8) return MOPK_ERR_BAD_ARG;
    }
    switch (a->dk) {
        case 16: return lm_launch<16>(a, bwd, st);
        case 32: return lm_launch<32>(a, bwd, st);
        default: return lm_launch<64>(a, bwd, st);
    }
}

}  // namespace mopk
