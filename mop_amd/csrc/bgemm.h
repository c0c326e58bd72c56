// Generic strided batched GEMM used by the multi-kernel ("generic") path.
//   C[b0,b1] = alpha * op(A)[b0,b1] * op(B)[b0,b1] + beta * C[b0,b1]
// fp32 in global memory; arithmetic is either exact fp32 FMA (PREC_FP32) or
// bf16 MFMA 16x16x32 with fp32 accumulation (PREC_BF16: operands are rounded to
// bf16 while being staged into LDS).  Any M/N/K/strides; edges are zero-filled.
// This is the any-shape fallback and the fp32-exact path -- the NS hot shape
// runs the fused kernels in edgewise_fused.hip instead.
#pragma once
#include "common.h"

namespace mopk {

struct GemmDesc {
    int M, N, K, nb0, nb1;
    const float *A;
    int64_t a_rs, a_cs, a_b0, a_b1;  // A(m,k)
    const float *B;
    int64_t b_rs, b_cs, b_b0, b_b1;  // B(k,n)
    float *C;
    int64_t c_rs, c_b0, c_b1;        // C(m,n), n contiguous
    float alpha;
    const float *alpha_dev;          // optional extra device-side scalar factor
    float beta;
};

constexpr int GB_M = 64, GB_N = 64, GB_K = 32, GB_LD = GB_K + 8;

// One operand tile of the bf16 path: ROWS (64 or 128) rows x 32 k of X(r, k) = X[r * rs + k * cs] -> registers (ROWS / 8 values per
// thread) -> LDS image [row][k] bf16 (row stride 40: 16-byte fragment reads without bank conflicts).  KC: k is the contiguous index
// (16-byte loads along k, 16-byte LDS stores); otherwise rows are (four rows x EPT / 4 k per thread: 16-byte loads along the rows,
// packed 4- or 8-byte stores).  `vec` = the host checked stride-1 / alignment; chunks that cross the matrix edge, and unaligned
// operands, take the scalar loads.
template <bool KC, int ROWS>
struct BgStage {
    static constexpr int EPT = ROWS * GB_K / 256, TPR = 256 / ROWS, RG = ROWS / 4, KP = EPT / 4;
    float v[EPT];
    __device__ __forceinline__ void load(const float *X, int64_t rs, int64_t cs, int r0, int k0, int R, int K, bool vec) {
        const int t = threadIdx.x;
        if (KC) {
            const int r = r0 + t / TPR, k = k0 + (t % TPR) * EPT;
            const float *p = X + (int64_t)r * rs + (int64_t)k * cs;
            if (vec && r < R && k + EPT - 1 < K) {
#pragma unroll
                for (int j = 0; j < EPT / 4; ++j) { const float4 a = *(const float4 *)(p + 4 * j); v[4 * j] = a.x; v[4 * j + 1] = a.y; v[4 * j + 2] = a.z; v[4 * j + 3] = a.w; }
            } else {
#pragma unroll
                for (int j = 0; j < EPT; ++j) v[j] = (r < R && k + j < K) ? p[(int64_t)j * cs] : 0.f;
            }
        } else {
            const int r = r0 + (t % RG) * 4, k = k0 + (t / RG) * KP;
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const float *p = X + (int64_t)r * rs + (int64_t)(k + j) * cs;
                if (vec && r + 3 < R && k + j < K) {
                    const float4 a = *(const float4 *)p;
                    v[4 * j] = a.x; v[4 * j + 1] = a.y; v[4 * j + 2] = a.z; v[4 * j + 3] = a.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[4 * j + i] = (r + i < R && k + j < K) ? p[(int64_t)i * rs] : 0.f;
                }
            }
        }
    }
    __device__ __forceinline__ void store(unsigned short *Xh) const {
        const int t = threadIdx.x;
        if (KC) {
#pragma unroll
            for (int hh = 0; hh < EPT / 8; ++hh) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(v[8 * hh + j]);
                *(bf16x8 *)&Xh[(t / TPR) * GB_LD + (t % TPR) * EPT + 8 * hh] = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {                // row r + i: k .. k + KP - 1 = v[i], v[4 + i], ...
                unsigned short *dst = &Xh[((t % RG) * 4 + i) * GB_LD + (t / RG) * KP];
                if (KP == 2) *(unsigned int *)dst = (unsigned int)f2bf(v[i]) | ((unsigned int)f2bf(v[4 + i]) << 16);
                else { bf16x4 o = {(short)f2bf(v[i]), (short)f2bf(v[4 + i]), (short)f2bf(v[8 + i]), (short)f2bf(v[12 + i])}; *(bf16x4 *)dst = o; }
            }
        }
    }
};

// bf16 MFMA variant, TM x TN output tile per workgroup (four waves, 2 x 2; 64 x 64 or 128 x 128 -- the larger tile halves the operand
// traffic per flop, which is what the N x N x N products are bound by: 8 TB/s of L2 reads with 64 x 64 tiles); the next k-tile is
// fetched into registers while the matrix core works on the current one
template <bool A_KC, bool B_NC, int TM, int TN>
__global__ __launch_bounds__(256) void bgemm_mfma_kernel(GemmDesc d, bool vecA, bool vecB) {
    __shared__ __attribute__((aligned(16))) unsigned short Ah[TM * GB_LD], Bh[TN * GB_LD];
    constexpr int MI = TM / 32, NI = TN / 32;
    const int t = threadIdx.x, bz = blockIdx.z, i0 = bz / d.nb1, i1 = bz % d.nb1;
    const float *A = d.A + i0 * d.a_b0 + i1 * d.a_b1;
    const float *B = d.B + i0 * d.b_b0 + i1 * d.b_b1;
    float *C = d.C + i0 * d.c_b0 + i1 * d.c_b1;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1;
    f32x4 macc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) macc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    BgStage<A_KC, TM> sa;
    BgStage<!B_NC, TN> sb;               // B(k, n): row index n, "k contiguous" when n is not
    sa.load(A, d.a_rs, d.a_cs, m0, 0, d.M, d.K, vecA);
    sb.load(B, d.b_cs, d.b_rs, n0, 0, d.N, d.K, vecB);
    for (int k0 = 0; k0 < d.K; k0 += GB_K) {
        sa.store(Ah);
        sb.store(Bh);
        __syncthreads();
        if (k0 + GB_K < d.K) {
            sa.load(A, d.a_rs, d.a_cs, m0, k0 + GB_K, d.M, d.K, vecA);
            sb.load(B, d.b_cs, d.b_rs, n0, k0 + GB_K, d.N, d.K, vecB);
        }
        bf16x8 af[MI], bfr[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8 *)&Ah[(wm * (TM / 2) + i * 16 + (lane & 15)) * GB_LD + 8 * (lane >> 4)];
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[j] = *(const bf16x8 *)&Bh[(wn * (TN / 2) + j * 16 + (lane & 15)) * GB_LD + 8 * (lane >> 4)];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], macc[i][j], 0, 0, 0);
        __syncthreads();
    }
    float alpha = d.alpha;
    if (d.alpha_dev) alpha *= *d.alpha_dev;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (TM / 2) + i * 16 + (lane >> 4) * 4 + r, n = n0 + wn * (TN / 2) + j * 16 + (lane & 15);
                if (m < d.M && n < d.N) {
                    float *p = C + (int64_t)m * d.c_rs + n;
                    float v = alpha * macc[i][j][r];
                    if (d.beta != 0.f) v += d.beta * *p;
                    *p = v;
                }
            }
}

// exact fp32 FMA variant
template <bool A_KC, bool B_NC>
__global__ __launch_bounds__(256) void bgemm_kernel(GemmDesc d) {
    __shared__ __attribute__((aligned(16))) float smem[2 * GB_K * (GB_M + 4)];
    const int t = threadIdx.x;
    const int bz = blockIdx.z;
    const int i0 = bz / d.nb1, i1 = bz % d.nb1;
    const float *A = d.A + i0 * d.a_b0 + i1 * d.a_b1;
    const float *B = d.B + i0 * d.b_b0 + i1 * d.b_b1;
    float *C = d.C + i0 * d.c_b0 + i1 * d.c_b1;
    const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
    float *As = smem;                           // [GB_K][GB_M+4]
    float *Bs = smem + GB_K * (GB_M + 4);       // [GB_K][GB_N+4]
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int tx = t & 15, ty = t >> 4;
    for (int k0 = 0; k0 < d.K; k0 += GB_K) {
#pragma unroll
        for (int i = 0; i < (GB_M * GB_K) / 256; ++i) {
            int idx = t + 256 * i;
            int m, k;
            if (A_KC) { k = idx % GB_K; m = idx / GB_K; } else { m = idx % GB_M; k = idx / GB_M; }
            float v = 0.f;
            if (m0 + m < d.M && k0 + k < d.K) v = A[(int64_t)(m0 + m) * d.a_rs + (int64_t)(k0 + k) * d.a_cs];
            As[k * (GB_M + 4) + m] = v;
        }
#pragma unroll
        for (int i = 0; i < (GB_N * GB_K) / 256; ++i) {
            int idx = t + 256 * i;
            int n, k;
            if (B_NC) { n = idx % GB_N; k = idx / GB_N; } else { k = idx % GB_K; n = idx / GB_K; }
            float v = 0.f;
            if (n0 + n < d.N && k0 + k < d.K) v = B[(int64_t)(k0 + k) * d.b_rs + (int64_t)(n0 + n) * d.b_cs];
            Bs[k * (GB_N + 4) + n] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < GB_K; ++k) {
            const float4 a4 = *(const float4 *)&As[k * (GB_M + 4) + ty * 4];
            const float4 b4 = *(const float4 *)&Bs[k * (GB_N + 4) + tx * 4];
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    float alpha = d.alpha;
    if (d.alpha_dev) alpha *= *d.alpha_dev;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < d.M && n < d.N) {
                float *p = C + (int64_t)m * d.c_rs + n;
                float v = alpha * acc[i][j];
                if (d.beta != 0.f) v += d.beta * *p;
                *p = v;
            }
        }
}

inline int bgemm(const GemmDesc &d, bool mfma, hipStream_t st) {
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.nb0 * d.nb1 <= 0) return MOPK_ERR_BAD_SHAPE;
    dim3 grid((d.N + GB_N - 1) / GB_N, (d.M + GB_M - 1) / GB_M, d.nb0 * d.nb1);
    const bool akc = d.a_cs == 1, bnc = d.b_cs == 1;
    if (mfma) {
        // 16-byte loads need the contiguous index to be stride 1 and every other stride / the base to keep 16-byte alignment
        auto al = [](const float *p, int64_t s0, int64_t s1, int64_t s2) { return ((uintptr_t)p % 16 == 0) && s0 % 4 == 0 && s1 % 4 == 0 && s2 % 4 == 0; };
        const bool vecA = akc ? al(d.A, d.a_rs, d.a_b0, d.a_b1) : (d.a_rs == 1 && al(d.A, d.a_cs, d.a_b0, d.a_b1));
        const bool vecB = bnc ? al(d.B, d.b_rs, d.b_b0, d.b_b1) : (d.b_rs == 1 && al(d.B, d.b_cs, d.b_b0, d.b_b1));
        const bool big = d.M > 96 && d.N > 96;             // N x N outputs: 128 x 128 tiles; N x dk / dk x N ones stay at 64 x 64
        const dim3 gbig((d.N + 127) / 128, (d.M + 127) / 128, d.nb0 * d.nb1);
#define MOPK_BG(AK, BN_) do { if (big) hipLaunchKernelGGL((bgemm_mfma_kernel<AK, BN_, 128, 128>), gbig, dim3(256), 0, st, d, vecA, vecB); \
                              else hipLaunchKernelGGL((bgemm_mfma_kernel<AK, BN_, 64, 64>), grid, dim3(256), 0, st, d, vecA, vecB); } while (0)
        if (akc && bnc) MOPK_BG(true, true); else if (akc) MOPK_BG(true, false);
        else if (bnc) MOPK_BG(false, true); else MOPK_BG(false, false);
#undef MOPK_BG
    } else {
#define MOPK_BG(AK, BN_) hipLaunchKernelGGL((bgemm_kernel<AK, BN_>), grid, dim3(256), 0, st, d)
        if (akc && bnc) MOPK_BG(true, true); else if (akc) MOPK_BG(true, false);
        else if (bnc) MOPK_BG(false, true); else MOPK_BG(false, false);
#undef MOPK_BG
    }
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

}  // namespace mopk
