// Quartet causal attention core -- fused gfx950 kernels (bf16 MFMA, fp32 accumulate, no T x T map in HBM).
//
// Replaces reference mop/models/quartet_attn_patch.py:88-121 when the attention weights are not requested (the additive
// attention_mask is applied in-kernel):  qk and q2k2 scores are z-normalised over the FULL row (all T keys, unbiased std), mixed
//   scores = (1 - m) z1 + m (z1 z2) quartet_scale          (use_quartet)        |   scores = z1   (otherwise, eps 1e-5)
// causally masked, soft-maxed and applied to v.  Because the row statistics need every key, the forward is two passes
// (statistics, then an online-softmax pass as in sdpa_flash.hip); the backward of the z-norm adds dense row corrections,
//   dS = (dz - mean(dz)) / den - (S - mu) sum(dz (S - mu)) / ((T-1) sd den^2),
// so it is: delta, a row-sum pass over the causal region, a dQ pass and a dK/dV pass over ALL tiles.  Same X layout as
// sdpa_flash.hip (lane = query, registers = keys; in the dK/dV pass lane = key, registers = queries).
#include "flash_common.h"

namespace mopk {

namespace {
struct QtScal { float m, mq, qs, eps1, eps2; };        // sigmoid(mixture), m * quartet_scale, quartet_scale, eps of the two z-norms
template <bool DUAL>
__device__ __forceinline__ QtScal qt_scal(const MopkQuartetArgs &a) {
    QtScal s;
    s.m = DUAL ? 1.f / (1.f + __expf(-*a.mixture)) : 0.f;         // :103
    s.qs = DUAL ? *a.quartet_scale : 0.f;
    s.mq = s.m * s.qs;
    s.eps1 = DUAL ? a.eps : 1e-5f;                                // :98 / :110
    s.eps2 = a.eps;
    return s;
}
// per-row quantities of one z-normalised map
struct QtNorm { float mu, inv, cc, am, bs; };          // mean, 1/(sd+eps), 1/((T-1) sd (sd+eps)^2), mean(dz), sum(dz (S-mu))
__device__ __forceinline__ QtNorm qt_norm(float mu, float sd, float eps, int T, float A, float B) {
    QtNorm n;
    n.mu = mu; n.inv = 1.f / (sd + eps);
    // sd == 0 means every score of the row equals mu: the (S - mu) term it multiplies is 0, and 1/sd would overflow to inf (0 * inf = NaN)
    n.cc = sd > 1e-12f ? n.inv * n.inv / ((float)max(T - 1, 1) * sd) : 0.f;
    n.am = A / (float)T; n.bs = B;
    return n;
}
template <bool DUAL>
__device__ __forceinline__ float qt_logit(float s1, float s2, const QtNorm &n1, const QtNorm &n2, const QtScal &q, float &z1, float &z2) {
    z1 = (s1 - n1.mu) * n1.inv;
    if (!DUAL) { z2 = 0.f; return z1; }
    z2 = (s2 - n2.mu) * n2.inv;
    return z1 * ((1.f - q.m) + q.mq * z2);                        // :104-106
}
// additive attention_mask (:115-116), element (b,h,i,j); indices clamped so the load is unconditional (no per-lane branch)
template <bool AM>
__device__ __forceinline__ float qt_am(const MopkQuartetArgs &a, const float *base, int i, int j) {
    return AM ? base[(int64_t)min(i, a.T - 1) * a.am_si + min(j, a.T - 1)] : 0.f;
}
}  // namespace

// ------------------------------------------------------------------ pass 1: row statistics over all T keys
template <int DK, typename IOT, bool DUAL>
__global__ void __launch_bounds__(FA_NW * 64) qt_stats_kernel(MopkQuartetArgs a, float *stats) {
    constexpr int LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[FA_KT * LDK], K2s[DUAL ? FA_KT * LDK : 8];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.T;
    int qb, bh;
    fa_block_id((N + FA_QB - 1) / FA_QB, qb, bh);
    const int b = bh / a.H, hh = bh % a.H;
    const int qi = qb * FA_QB + 32 * w + r;
    const bool qok = qi < N;
    const float sc = rsqrtf((float)DK);
    bf16x8 qe[DK / 16], q2e[DUAL ? DK / 16 : 1];
    fa_frags<DK, IOT>(qe, (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn, qok, h, sc);
    const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh, *k2p = nullptr;
    if (DUAL) {
        fa_frags<DK, IOT>(*(bf16x8(*)[DK / 16]) & q2e, (const IOT *)a.q2.ptr + b * a.q2.sb + hh * a.q2.sh + (int64_t)qi * a.q2.sn, qok, h, sc);
        k2p = (const IOT *)a.k2.ptr + b * a.k2.sb + hh * a.k2.sh;
    }
    float pv1 = 0.f, pv2 = 0.f, s1 = 0.f, ss1 = 0.f, s2 = 0.f, ss2 = 0.f;   // sums of (S - pivot), pivot = S[i, 0]: no cancellation
    const int nkt = (N + FA_KT - 1) / FA_KT;
    FaTile<DK> fk, fk2;
    fa_fetch<DK, IOT>(fk, kp, a.k.sn, 0, N, 1.f, tid);
    if (DUAL) fa_fetch<DK, IOT>(fk2, k2p, a.k2.sn, 0, N, 1.f, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * FA_KT;
        __syncthreads();
        fa_put<DK, true, false>(Ks, nullptr, fk, tid);
        if (DUAL) fa_put<DK, true, false>(K2s, nullptr, fk2, tid);
        if (kt + 1 < nkt) {
            fa_fetch<DK, IOT>(fk, kp, a.k.sn, k0 + FA_KT, N, 1.f, tid);
            if (DUAL) fa_fetch<DK, IOT>(fk2, k2p, a.k2.sn, k0 + FA_KT, N, 1.f, tid);
        }
        __syncthreads();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x16 S1 = fa_mm_rows<DK>(Ks, 32 * hf, r, h, qe);
            f32x16 S2 = fa_zero();
            if (DUAL) S2 = fa_mm_rows<DK>(K2s, 32 * hf, r, h, *(const bf16x8(*)[DK / 16]) & q2e);
            if (kt == 0 && hf == 0) { pv1 = __shfl(S1[0], r, 64); pv2 = __shfl(S2[0], r, 64); }   // key 0 sits in register 0 of half 0
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (k0 + 32 * hf + tile_row(g, h) < N) {
                    const float d1 = S1[g] - pv1, d2 = S2[g] - pv2;
                    s1 += d1; ss1 = fmaf(d1, d1, ss1); s2 += d2; ss2 = fmaf(d2, d2, ss2);
                }
            }
        }
    }
    s1 += __shfl_xor(s1, 32, 64); ss1 += __shfl_xor(ss1, 32, 64); s2 += __shfl_xor(s2, 32, 64); ss2 += __shfl_xor(ss2, 32, 64);
    if (qok && h == 0) {
        const float T = (float)N, tm1 = (float)max(N - 1, 1);
        float4 o;
        o.x = pv1 + s1 / T; o.y = sqrtf(fmaxf((ss1 - s1 * s1 / T) / tm1, 0.f));          // mean, unbiased std   :96-97
        o.z = pv2 + s2 / T; o.w = sqrtf(fmaxf((ss2 - s2 * s2 / T) / tm1, 0.f));
        ((float4 *)stats)[(int64_t)bh * N + qi] = o;
    }
}

// ------------------------------------------------------------------ pass 2: mixed logits, causal online softmax, A v
template <int DK, typename IOT, bool DUAL, bool AM>
__global__ void __launch_bounds__(FA_NW * 64, 2) qt_fwd_kernel(MopkQuartetArgs a, const float *stats, float *lse) {
    constexpr int DT = DK / 32, LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[FA_KT * LDK], Vt[DK * FA_LDT], K2s[DUAL ? FA_KT * LDK : 8];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.T;
    int qb, bh;
    fa_block_id((N + FA_QB - 1) / FA_QB, qb, bh);
    const int b = bh / a.H, hh = bh % a.H;
    const int q0 = qb * FA_QB, qi = q0 + 32 * w + r;
    const bool qok = qi < N;
    const float sc = rsqrtf((float)DK);
    const QtScal qs = qt_scal<DUAL>(a);
    const float *amp = AM ? a.add_mask + b * a.am_sb + hh * a.am_sh : nullptr;
    bf16x8 qe[DK / 16], q2e[DUAL ? DK / 16 : 1];
    fa_frags<DK, IOT>(qe, (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn, qok, h, sc);
    const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh, *vp = (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh, *k2p = nullptr;
    if (DUAL) {
        fa_frags<DK, IOT>(*(bf16x8(*)[DK / 16]) & q2e, (const IOT *)a.q2.ptr + b * a.q2.sb + hh * a.q2.sh + (int64_t)qi * a.q2.sn, qok, h, sc);
        k2p = (const IOT *)a.k2.ptr + b * a.k2.sb + hh * a.k2.sh;
    }
    const float4 st = qok ? ((const float4 *)stats)[(int64_t)bh * N + qi] : make_float4(0.f, 1.f, 0.f, 1.f);
    const QtNorm n1 = qt_norm(st.x, st.y, qs.eps1, N, 0.f, 0.f), n2 = qt_norm(st.z, st.w, qs.eps2, N, 0.f, 0.f);
    const FaDrop drop = fa_drop(a.dropout_p, a.dropout_seed);
    const uint32_t rowh = fa_drop_row(drop, bh, qi);
    float m = FA_NEG, l = 0.f;
    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) O[dt] = fa_zero();
    const int nkt = min((N + FA_KT - 1) / FA_KT, (min(q0 + FA_QB, N) + FA_KT - 1) / FA_KT);      // causal   :112-113
    FaTile<DK> fk, fv, fk2;
    fa_fetch<DK, IOT>(fk, kp, a.k.sn, 0, N, 1.f, tid);
    fa_fetch<DK, IOT>(fv, vp, a.v.sn, 0, N, 1.f, tid);
    if (DUAL) fa_fetch<DK, IOT>(fk2, k2p, a.k2.sn, 0, N, 1.f, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * FA_KT;
        __syncthreads();
        fa_put<DK, true, false>(Ks, nullptr, fk, tid);
        fa_put<DK, false, true>(nullptr, Vt, fv, tid);
        if (DUAL) fa_put<DK, true, false>(K2s, nullptr, fk2, tid);
        if (kt + 1 < nkt) {
            fa_fetch<DK, IOT>(fk, kp, a.k.sn, k0 + FA_KT, N, 1.f, tid);
            fa_fetch<DK, IOT>(fv, vp, a.v.sn, k0 + FA_KT, N, 1.f, tid);
            if (DUAL) fa_fetch<DK, IOT>(fk2, k2p, a.k2.sn, k0 + FA_KT, N, 1.f, tid);
        }
        __syncthreads();
        f32x16 S[2];
        float mx = FA_NEG;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            S[hf] = fa_mm_rows<DK>(Ks, 32 * hf, r, h, qe);
            f32x16 T2 = fa_zero();
            if (DUAL) T2 = fa_mm_rows<DK>(K2s, 32 * hf, r, h, *(const bf16x8(*)[DK / 16]) & q2e);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int j = k0 + 32 * hf + tile_row(g, h);
                float z1, z2;
                const float lg = (qt_logit<DUAL>(S[hf][g], T2[g], n1, n2, qs, z1, z2) + qt_am<AM>(a, amp, qi, j)) * FA_LOG2E;
                S[hf][g] = (j >= N || j > qi) ? FA_NEG : lg;
                mx = fmaxf(mx, S[hf][g]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx), alpha = __builtin_amdgcn_exp2f(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int g = 0; g < 16; ++g) { const float p = __builtin_amdgcn_exp2f(S[hf][g] - mn); S[hf][g] = p; ps += p; }
        ps += __shfl_xor(ps, 32, 64);
        l = fmaf(l, alpha, ps);
        m = mn;
        if (drop.thresh) {                      // attn_dropout (:119): the row sum is of the undropped probabilities
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int g = 0; g < 16; ++g) S[hf][g] = fa_drop_keep(drop, rowh, k0 + 32 * hf + tile_row(g, h)) ? S[hf][g] * drop.inv_keep : 0.f;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 16; ++g) O[dt][g] *= alpha;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bf16x8 lo, hi;
            fa_pack(lo, hi, S[hf]);
            fa_mm_cols<DK>(O, Vt, hf, r, h, lo, hi);
        }
    }
    if (qok) {
        fa_store_rows<DK, IOT>((IOT *)a.y.ptr + b * a.y.sb + hh * a.y.sh + (int64_t)qi * a.y.sn, O, h, 1.f / l);
        if (h == 0) lse[(int64_t)bh * N + qi] = m + __builtin_amdgcn_logf(l);
    }
}

// ------------------------------------------------------------------ backward
template <typename IOT>
__global__ void qt_delta_kernel(MopkQuartetArgs a, float *delta) {          // delta_i = dy_i . y_i
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)a.B * a.H * a.T) return;
    const int lane = threadIdx.x & 63, i = row % a.T;
    const int64_t bh = row / a.T;
    const int b = bh / a.H, hh = bh % a.H;
    const IOT *yp = (const IOT *)a.y.ptr + b * a.y.sb + hh * a.y.sh + (int64_t)i * a.y.sn;
    const IOT *gp = (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh + (int64_t)i * a.dy.sn;
    float s = 0.f;
    for (int d = lane; d < a.dh; d += 64) s = fmaf(ld_as_f32(yp + d), ld_as_f32(gp + d), s);
    s = wave_sum(s);
    if (lane == 0) delta[row] = s;
}

// row sums of the z-norm backward over the causal region + per-block partials of the two scalar gradients
template <int DK, typename IOT, bool DUAL, bool AM>
__global__ void __launch_bounds__(FA_NW * 64, 2) qt_rowsum_kernel(MopkQuartetArgs a, const float *stats, const float *lse, const float *delta,
                                                               float *rows, float *spart) {
    constexpr int LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[FA_KT * LDK], Vs[FA_KT * LDK], K2s[DUAL ? FA_KT * LDK : 8];
    __shared__ float red[2][FA_NW];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.T;
    int qb, bh;
    fa_block_id((N + FA_QB - 1) / FA_QB, qb, bh);
    const int b = bh / a.H, hh = bh % a.H;
    const int q0 = qb * FA_QB, qi = q0 + 32 * w + r;
    const bool qok = qi < N;
    const float sc = rsqrtf((float)DK);
    const QtScal qs = qt_scal<DUAL>(a);
    const float *amp = AM ? a.add_mask + b * a.am_sb + hh * a.am_sh : nullptr;
    bf16x8 qe[DK / 16], q2e[DUAL ? DK / 16 : 1], dof[DK / 16];
    fa_frags<DK, IOT>(qe, (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn, qok, h, sc);
    fa_frags<DK, IOT>(dof, (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh + (int64_t)qi * a.dy.sn, qok, h, 1.f);
    const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh, *vp = (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh, *k2p = nullptr;
    if (DUAL) {
        fa_frags<DK, IOT>(*(bf16x8(*)[DK / 16]) & q2e, (const IOT *)a.q2.ptr + b * a.q2.sb + hh * a.q2.sh + (int64_t)qi * a.q2.sn, qok, h, sc);
        k2p = (const IOT *)a.k2.ptr + b * a.k2.sb + hh * a.k2.sh;
    }
    const float4 st = qok ? ((const float4 *)stats)[(int64_t)bh * N + qi] : make_float4(0.f, 1.f, 0.f, 1.f);
    const QtNorm n1 = qt_norm(st.x, st.y, qs.eps1, N, 0.f, 0.f), n2 = qt_norm(st.z, st.w, qs.eps2, N, 0.f, 0.f);
    const float Li = qok ? lse[(int64_t)bh * N + qi] : 0.f, di = qok ? delta[(int64_t)bh * N + qi] : 0.f;
    const FaDrop drop = fa_drop(a.dropout_p, a.dropout_seed);
    const uint32_t rowh = fa_drop_row(drop, bh, qi);
    float A1 = 0.f, B1 = 0.f, A2 = 0.f, B2 = 0.f, gm = 0.f, gq = 0.f;
    const int nkt = min((N + FA_KT - 1) / FA_KT, (min(q0 + FA_QB, N) + FA_KT - 1) / FA_KT);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * FA_KT;
        __syncthreads();
        fa_stage<DK, IOT, true, false>(Ks, nullptr, kp, a.k.sn, k0, N, 1.f, tid);
        fa_stage<DK, IOT, true, false>(Vs, nullptr, vp, a.v.sn, k0, N, 1.f, tid);
        if (DUAL) fa_stage<DK, IOT, true, false>(K2s, nullptr, k2p, a.k2.sn, k0, N, 1.f, tid);
        __syncthreads();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x16 S1 = fa_mm_rows<DK>(Ks, 32 * hf, r, h, qe);
            const f32x16 dP = fa_mm_rows<DK>(Vs, 32 * hf, r, h, dof);
            f32x16 S2 = fa_zero();
            if (DUAL) S2 = fa_mm_rows<DK>(K2s, 32 * hf, r, h, *(const bf16x8(*)[DK / 16]) & q2e);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int j = k0 + 32 * hf + tile_row(g, h);
                if (qok && j < N && j <= qi) {
                    float z1, z2;
                    const float lg = qt_logit<DUAL>(S1[g], S2[g], n1, n2, qs, z1, z2) + qt_am<AM>(a, amp, qi, j);
                    const float dpd = drop.thresh ? (fa_drop_keep(drop, rowh, j) ? dP[g] * drop.inv_keep : 0.f) : dP[g];
                    const float dsc = __builtin_amdgcn_exp2f(lg * FA_LOG2E - Li) * (dpd - di);
                    const float dz1 = DUAL ? dsc * ((1.f - qs.m) + qs.mq * z2) : dsc;
                    A1 += dz1; B1 = fmaf(dz1, S1[g] - n1.mu, B1);
                    if (DUAL) {
                        const float dz2 = dsc * qs.mq * z1;
                        A2 += dz2; B2 = fmaf(dz2, S2[g] - n2.mu, B2);
                        gm = fmaf(dsc, z1 * (z2 * qs.qs - 1.f), gm);          // d scores / d m
                        gq = fmaf(dsc, qs.m * z1 * z2, gq);                   // d scores / d quartet_scale
                    }
                }
            }
        }
    }
    A1 += __shfl_xor(A1, 32, 64); B1 += __shfl_xor(B1, 32, 64); A2 += __shfl_xor(A2, 32, 64); B2 += __shfl_xor(B2, 32, 64);
    if (qok && h == 0) ((float4 *)rows)[(int64_t)bh * N + qi] = make_float4(A1, B1, A2, B2);
    gm = wave_sum(gm); gq = wave_sum(gq);
    if (lane == 0) { red[0][w] = gm; red[1][w] = gq; }
    __syncthreads();
    if (tid == 0) {
        float x = 0.f, y = 0.f;
        for (int i = 0; i < FA_NW; ++i) { x += red[0][i]; y += red[1][i]; }
        const int nq = (N + FA_QB - 1) / FA_QB;
        spart[((int64_t)bh * nq + qb) * 2] = x; spart[((int64_t)bh * nq + qb) * 2 + 1] = y;
    }
}
// per (b,h): fixed-order sum of the block partials; dmixture carries sigmoid'(mixture)
__global__ void qt_scalar_sum_kernel(const float *spart, int nblk, int nbh, const float *mixture, float *dmix, float *dqs) {
    const int bh = blockIdx.x * blockDim.x + threadIdx.x;
    if (bh >= nbh) return;
    float x = 0.f, y = 0.f;
    for (int i = 0; i < nblk; ++i) { x += spart[((int64_t)bh * nblk + i) * 2]; y += spart[((int64_t)bh * nblk + i) * 2 + 1]; }
    const float m = 1.f / (1.f + __expf(-*mixture));
    dmix[bh] = x * m * (1.f - m); dqs[bh] = y;
}

// gradient of one z-normalised map at one element: dense row correction included
__device__ __forceinline__ float qt_dscore(float dz, float s, const QtNorm &n) {
    return (dz - n.am) * n.inv - (s - n.mu) * n.bs * n.cc;
}

// dQ (and dQ2): per query block over ALL key tiles (the z-norm correction is dense)
template <int DK, typename IOT, bool DUAL, bool AM>
__global__ void __launch_bounds__(FA_NW * 64, 2) qt_dq_kernel(MopkQuartetArgs a, const float *stats, const float *lse, const float *delta,
                                                           const float *rows) {
    constexpr int DT = DK / 32, LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[FA_KT * LDK], Vs[FA_KT * LDK], Kt[DK * FA_LDT];
    __shared__ __attribute__((aligned(16))) unsigned short K2s[DUAL ? FA_KT * LDK : 8], K2t[DUAL ? DK * FA_LDT : 8];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.T;
    int qb, bh;
    fa_block_id((N + FA_QB - 1) / FA_QB, qb, bh);
    const int b = bh / a.H, hh = bh % a.H;
    const int q0 = qb * FA_QB, qi = q0 + 32 * w + r;
    const bool qok = qi < N;
    const float sc = rsqrtf((float)DK);
    const QtScal qs = qt_scal<DUAL>(a);
    const float *amp = AM ? a.add_mask + b * a.am_sb + hh * a.am_sh : nullptr;
    bf16x8 qe[DK / 16], q2e[DUAL ? DK / 16 : 1], dof[DK / 16];
    fa_frags<DK, IOT>(qe, (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn, qok, h, sc);
    fa_frags<DK, IOT>(dof, (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh + (int64_t)qi * a.dy.sn, qok, h, 1.f);
    const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh, *vp = (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh, *k2p = nullptr;
    if (DUAL) {
        fa_frags<DK, IOT>(*(bf16x8(*)[DK / 16]) & q2e, (const IOT *)a.q2.ptr + b * a.q2.sb + hh * a.q2.sh + (int64_t)qi * a.q2.sn, qok, h, sc);
        k2p = (const IOT *)a.k2.ptr + b * a.k2.sb + hh * a.k2.sh;
    }
    const float4 st = qok ? ((const float4 *)stats)[(int64_t)bh * N + qi] : make_float4(0.f, 1.f, 0.f, 1.f);
    const float4 rw = qok ? ((const float4 *)rows)[(int64_t)bh * N + qi] : make_float4(0.f, 0.f, 0.f, 0.f);
    const QtNorm n1 = qt_norm(st.x, st.y, qs.eps1, N, rw.x, rw.y), n2 = qt_norm(st.z, st.w, qs.eps2, N, rw.z, rw.w);
    const float Li = qok ? lse[(int64_t)bh * N + qi] : 0.f, di = qok ? delta[(int64_t)bh * N + qi] : 0.f;
    const FaDrop drop = fa_drop(a.dropout_p, a.dropout_seed);
    const uint32_t rowh = fa_drop_row(drop, bh, qi);
    f32x16 dQ[DT], dQ2[DUAL ? DT : 1];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dQ[dt] = fa_zero(); if (DUAL) dQ2[dt] = fa_zero(); }
    const int nkt = (N + FA_KT - 1) / FA_KT;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * FA_KT;
        __syncthreads();
        fa_stage<DK, IOT, true, true>(Ks, Kt, kp, a.k.sn, k0, N, 1.f, tid);
        fa_stage<DK, IOT, true, false>(Vs, nullptr, vp, a.v.sn, k0, N, 1.f, tid);
        if (DUAL) fa_stage<DK, IOT, true, true>(K2s, K2t, k2p, a.k2.sn, k0, N, 1.f, tid);
        __syncthreads();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x16 S1 = fa_mm_rows<DK>(Ks, 32 * hf, r, h, qe);
            const f32x16 dP = fa_mm_rows<DK>(Vs, 32 * hf, r, h, dof);
            f32x16 S2 = fa_zero();
            if (DUAL) S2 = fa_mm_rows<DK>(K2s, 32 * hf, r, h, *(const bf16x8(*)[DK / 16]) & q2e);
            f32x16 g1, g2;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int j = k0 + 32 * hf + tile_row(g, h);
                float z1, z2, dz1 = 0.f, dz2 = 0.f;
                const float lg = qt_logit<DUAL>(S1[g], S2[g], n1, n2, qs, z1, z2) + qt_am<AM>(a, amp, qi, j);
                if (j <= qi) {                                    // causal region: the soft-max sees this edge
                    const float dpd = drop.thresh ? (fa_drop_keep(drop, rowh, j) ? dP[g] * drop.inv_keep : 0.f) : dP[g];
                    const float dsc = __builtin_amdgcn_exp2f(lg * FA_LOG2E - Li) * (dpd - di);
                    dz1 = DUAL ? dsc * ((1.f - qs.m) + qs.mq * z2) : dsc;
                    dz2 = dsc * qs.mq * z1;
                }
                const bool ok = qok && j < N;
                g1[g] = ok ? qt_dscore(dz1, S1[g], n1) * sc : 0.f;
                g2[g] = (DUAL && ok) ? qt_dscore(dz2, S2[g], n2) * sc : 0.f;
            }
            bf16x8 lo, hi;
            fa_pack(lo, hi, g1);
            fa_mm_cols<DK>(dQ, Kt, hf, r, h, lo, hi);
            if (DUAL) { fa_pack(lo, hi, g2); fa_mm_cols<DK>(*(f32x16(*)[DT]) & dQ2, K2t, hf, r, h, lo, hi); }
        }
    }
    if (qok) {
        fa_store_rows<DK, IOT>((IOT *)a.dq.ptr + b * a.dq.sb + hh * a.dq.sh + (int64_t)qi * a.dq.sn, dQ, h, 1.f);
        if (DUAL) fa_store_rows<DK, IOT>((IOT *)a.dq2.ptr + b * a.dq2.sb + hh * a.dq2.sh + (int64_t)qi * a.dq2.sn, *(const f32x16(*)[DT]) & dQ2, h, 1.f);
    }
}

// dK, dK2, dV: per key block (lane = key) over ALL query tiles
template <int DK, typename IOT, bool DUAL, bool AM>
__global__ void __launch_bounds__(FA_NW * 64, 2) qt_dkv_kernel(MopkQuartetArgs a, const float *stats, const float *lse, const float *delta,
                                                            const float *rows) {
    constexpr int DT = DK / 32, LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Qs[FA_KT * LDK], Gs[FA_KT * LDK], Qt[DK * FA_LDT], Gt[DK * FA_LDT];
    __shared__ __attribute__((aligned(16))) unsigned short Q2s[DUAL ? FA_KT * LDK : 8], Q2t[DUAL ? DK * FA_LDT : 8];
    __shared__ float4 Rs[FA_KT][3];                    // per query: (mu1, inv1, cc1, am1) (bs1, mu2, inv2, cc2) (am2, bs2, L, delta)
    __shared__ uint32_t Hs[FA_KT];                     // dropout row hashes of the tile's queries
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.T;
    int kb, bh;
    fa_block_id((N + FA_QB - 1) / FA_QB, kb, bh);
    const int b = bh / a.H, hh = bh % a.H;
    const int k0 = kb * FA_QB, kj = k0 + 32 * w + r;
    const bool kok = kj < N;
    const float sc = rsqrtf((float)DK);
    const QtScal qs = qt_scal<DUAL>(a);
    const float *amp = AM ? a.add_mask + b * a.am_sb + hh * a.am_sh : nullptr;
    const IOT *qp = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh, *gp = (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh, *q2p = nullptr;
    bf16x8 kf[DK / 16], vf[DK / 16], k2f[DUAL ? DK / 16 : 1];
    fa_frags<DK, IOT>(kf, (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh + (int64_t)kj * a.k.sn, kok, h, 1.f);
    fa_frags<DK, IOT>(vf, (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh + (int64_t)kj * a.v.sn, kok, h, 1.f);
    if (DUAL) {
        fa_frags<DK, IOT>(*(bf16x8(*)[DK / 16]) & k2f, (const IOT *)a.k2.ptr + b * a.k2.sb + hh * a.k2.sh + (int64_t)kj * a.k2.sn, kok, h, 1.f);
        q2p = (const IOT *)a.q2.ptr + b * a.q2.sb + hh * a.q2.sh;
    }
    f32x16 dK[DT], dV[DT], dK2[DUAL ? DT : 1];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dK[dt] = fa_zero(); dV[dt] = fa_zero(); if (DUAL) dK2[dt] = fa_zero(); }
    const FaDrop drop = fa_drop(a.dropout_p, a.dropout_seed);
    const int nqt = (N + FA_KT - 1) / FA_KT;
    for (int qt = 0; qt < nqt; ++qt) {
        const int i0 = qt * FA_KT;
        __syncthreads();
        fa_stage<DK, IOT, true, true>(Qs, Qt, qp, a.q.sn, i0, N, sc, tid);        // q / sqrt(dk), rounded as in the forward
        fa_stage<DK, IOT, true, true>(Gs, Gt, gp, a.dy.sn, i0, N, 1.f, tid);
        if (DUAL) fa_stage<DK, IOT, true, true>(Q2s, Q2t, q2p, a.q2.sn, i0, N, sc, tid);
        if (tid < FA_KT) {
            const int i = i0 + tid;
            float4 st = make_float4(0.f, 1.f, 0.f, 1.f), rw = make_float4(0.f, 0.f, 0.f, 0.f);
            float L = 0.f, dl = 0.f;
            if (i < N) { st = ((const float4 *)stats)[(int64_t)bh * N + i]; rw = ((const float4 *)rows)[(int64_t)bh * N + i];
                         L = lse[(int64_t)bh * N + i]; dl = delta[(int64_t)bh * N + i]; }
            const QtNorm n1 = qt_norm(st.x, st.y, qs.eps1, N, rw.x, rw.y), n2 = qt_norm(st.z, st.w, qs.eps2, N, rw.z, rw.w);
            Rs[tid][0] = make_float4(n1.mu, n1.inv, n1.cc, n1.am);
            Rs[tid][1] = make_float4(n1.bs, n2.mu, n2.inv, n2.cc);
            Rs[tid][2] = make_float4(n2.am, n2.bs, L, dl);
            Hs[tid] = fa_drop_row(drop, bh, i);
        }
        __syncthreads();
        const bool live = i0 + FA_KT > k0;               // causal: some query of this tile can see a key of this block
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x16 S1 = fa_mm_rows<DK>(Qs, 32 * hf, r, h, kf);               // rows = queries, lane = key
            f32x16 dP = fa_zero(), S2 = fa_zero();
            if (live) dP = fa_mm_rows<DK>(Gs, 32 * hf, r, h, vf);
            if (DUAL) S2 = fa_mm_rows<DK>(Q2s, 32 * hf, r, h, *(const bf16x8(*)[DK / 16]) & k2f);
            f32x16 P, g1, g2;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int il = 32 * hf + tile_row(g, h), i = i0 + il;
                const float4 ra = Rs[il][0], rb = Rs[il][1], rc = Rs[il][2];
                const QtNorm n1{ra.x, ra.y, ra.z, ra.w, rb.x}, n2{rb.y, rb.z, rb.w, rc.x, rc.y};
                float z1, z2, dz1 = 0.f, dz2 = 0.f, p = 0.f;
                const float lg = qt_logit<DUAL>(S1[g], S2[g], n1, n2, qs, z1, z2) + qt_am<AM>(a, amp, i, kj);
                const bool ok = kok && i < N;
                if (ok && kj <= i) {
                    p = __builtin_amdgcn_exp2f(lg * FA_LOG2E - rc.z);
                    float dpd = dP[g], pk = 1.f;
                    if (drop.thresh) { pk = fa_drop_keep(drop, Hs[il], kj) ? drop.inv_keep : 0.f; dpd *= pk; }
                    const float dsc = p * (dpd - rc.w);
                    p *= pk;                                      // dV sees the dropped probabilities
                    dz1 = DUAL ? dsc * ((1.f - qs.m) + qs.mq * z2) : dsc;
                    dz2 = dsc * qs.mq * z1;
                }
                P[g] = p;
                g1[g] = ok ? qt_dscore(dz1, S1[g], n1) : 0.f;     // Qt holds q / sqrt(dk): the scale is already in the operand
                g2[g] = (DUAL && ok) ? qt_dscore(dz2, S2[g], n2) : 0.f;
            }
            bf16x8 lo, hi;
            if (live) { fa_pack(lo, hi, P); fa_mm_cols<DK>(dV, Gt, hf, r, h, lo, hi); }
            fa_pack(lo, hi, g1);
            fa_mm_cols<DK>(dK, Qt, hf, r, h, lo, hi);
            if (DUAL) { fa_pack(lo, hi, g2); fa_mm_cols<DK>(*(f32x16(*)[DT]) & dK2, Q2t, hf, r, h, lo, hi); }
        }
    }
    if (kok) {
        fa_store_rows<DK, IOT>((IOT *)a.dk_.ptr + b * a.dk_.sb + hh * a.dk_.sh + (int64_t)kj * a.dk_.sn, dK, h, 1.f);
        fa_store_rows<DK, IOT>((IOT *)a.dv.ptr + b * a.dv.sb + hh * a.dv.sh + (int64_t)kj * a.dv.sn, dV, h, 1.f);
        if (DUAL) fa_store_rows<DK, IOT>((IOT *)a.dk2.ptr + b * a.dk2.sb + hh * a.dk2.sh + (int64_t)kj * a.dk2.sn, *(const f32x16(*)[DT]) & dK2, h, 1.f);
    }
}

// ------------------------------------------------------------------ host side
namespace {
bool qt_al(const MopkView4 &v, int es) {
    const int64_t al = 16 / es;
    return ((uintptr_t)v.ptr & 15) == 0 && v.sb % al == 0 && v.sh % al == 0 && v.sn % al == 0;
}
struct QtBufs { float *stats, *lse, *delta, *rows, *spart; size_t nsaved, nwork; };
QtBufs qt_carve_flash(void *saved, void *ws, const MopkQuartetArgs *a) {
    Carver cs(saved), cw(ws);
    const size_t rows = (size_t)a->B * a->H * a->T, nblk = (a->T + FA_QB - 1) / FA_QB;
    QtBufs b;
    b.stats = cs.take<float>(rows * 4); b.lse = cs.take<float>(rows);
    b.delta = cw.take<float>(rows); b.rows = cw.take<float>(rows * 4); b.spart = cw.take<float>((size_t)a->B * a->H * nblk * 2);
    b.nsaved = cs.off; b.nwork = cw.off;
    return b;
}
}  // namespace

int qt_flash_supported(const MopkQuartetArgs *a, bool bwd) {
    if (a->precision != MOPK_PREC_BF16) return 0;
    if (a->attn) return 0;                                         // returned attention weights: generic path
    if (a->dh != 32 && a->dh != 64) return 0;
    const int es = a->io_dtype == MOPK_BF16 ? 2 : 4;
    if (!qt_al(a->q, es) || !qt_al(a->k, es) || !qt_al(a->v, es) || !qt_al(a->y, es)) return 0;
    if (a->use_quartet && (!qt_al(a->q2, es) || !qt_al(a->k2, es))) return 0;
    if (bwd) {
        if (!qt_al(a->dy, es) || !qt_al(a->dq, es) || !qt_al(a->dk_, es) || !qt_al(a->dv, es)) return 0;
        if (a->use_quartet && (!qt_al(a->dq2, es) || !qt_al(a->dk2, es))) return 0;
    }
    return 1;
}
size_t qt_flash_saved_bytes(const MopkQuartetArgs *a) { return qt_carve_flash(nullptr, nullptr, a).nsaved + 256; }
size_t qt_flash_ws_bytes(const MopkQuartetArgs *a) { return qt_carve_flash(nullptr, nullptr, a).nwork + 256; }

#define QT_LAUNCH_T(KERNEL, DK_, IOT_, GRID, ...)                                                                \
    do { if (a->use_quartet) hipLaunchKernelGGL((KERNEL<DK_, IOT_, true>), GRID, dim3(FA_NW * 64), 0, st, __VA_ARGS__);      \
         else hipLaunchKernelGGL((KERNEL<DK_, IOT_, false>), GRID, dim3(FA_NW * 64), 0, st, __VA_ARGS__); } while (0)
#define QT_LAUNCH_TA(KERNEL, DK_, IOT_, GRID, ...)                                                               \
    do { if (a->use_quartet) { if (a->add_mask) hipLaunchKernelGGL((KERNEL<DK_, IOT_, true, true>), GRID, dim3(FA_NW * 64), 0, st, __VA_ARGS__);   \
                               else hipLaunchKernelGGL((KERNEL<DK_, IOT_, true, false>), GRID, dim3(FA_NW * 64), 0, st, __VA_ARGS__); }           \
         else { if (a->add_mask) hipLaunchKernelGGL((KERNEL<DK_, IOT_, false, true>), GRID, dim3(FA_NW * 64), 0, st, __VA_ARGS__);                \
                else hipLaunchKernelGGL((KERNEL<DK_, IOT_, false, false>), GRID, dim3(FA_NW * 64), 0, st, __VA_ARGS__); } } while (0)
#define QT_DISPATCH(MACRO, KERNEL, GRID, ...)                                                                    \
    do {                                                                                                         \
        if (a->io_dtype == MOPK_BF16) { if (a->dh == 64) MACRO(KERNEL, 64, unsigned short, GRID, __VA_ARGS__);   \
                                        else MACRO(KERNEL, 32, unsigned short, GRID, __VA_ARGS__); }             \
        else { if (a->dh == 64) MACRO(KERNEL, 64, float, GRID, __VA_ARGS__); else MACRO(KERNEL, 32, float, GRID, __VA_ARGS__); }   \
    } while (0)
#define QT_LAUNCH(KERNEL, GRID, ...) QT_DISPATCH(QT_LAUNCH_TA, KERNEL, GRID, __VA_ARGS__)
#define QT_LAUNCH_STATS(KERNEL, GRID, ...) QT_DISPATCH(QT_LAUNCH_T, KERNEL, GRID, __VA_ARGS__)

int qt_flash_fwd(const MopkQuartetArgs *a, hipStream_t st) {
    if (!qt_flash_supported(a, false)) return MOPK_ERR_UNSUPPORTED;
    const QtBufs b = qt_carve_flash(a->saved, a->workspace, a);
    const int nq = (a->T + FA_QB - 1) / FA_QB;
    const dim3 grid(nq * a->B * a->H);
    QT_LAUNCH_STATS(qt_stats_kernel, grid, *a, b.stats);
    QT_LAUNCH(qt_fwd_kernel, grid, *a, (const float *)b.stats, b.lse);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}
int qt_flash_bwd(const MopkQuartetArgs *a, hipStream_t st) {
    if (!qt_flash_supported(a, true)) return MOPK_ERR_UNSUPPORTED;
    const QtBufs b = qt_carve_flash(a->saved, a->workspace, a);
    const int64_t rows = (int64_t)a->B * a->H * a->T;
    if (a->io_dtype == MOPK_BF16) hipLaunchKernelGGL((qt_delta_kernel<unsigned short>), dim3((rows + 3) / 4), dim3(256), 0, st, *a, b.delta);
    else hipLaunchKernelGGL((qt_delta_kernel<float>), dim3((rows + 3) / 4), dim3(256), 0, st, *a, b.delta);
    const int nq = (a->T + FA_QB - 1) / FA_QB;
    const dim3 grid(nq * a->B * a->H);
    QT_LAUNCH(qt_rowsum_kernel, grid, *a, (const float *)b.stats, (const float *)b.lse, (const float *)b.delta, b.rows, b.spart);
    if (a->use_quartet)
        hipLaunchKernelGGL(qt_scalar_sum_kernel, dim3((a->B * a->H + 63) / 64), dim3(64), 0, st, (const float *)b.spart, nq, a->B * a->H,
                           a->mixture, a->dmixture_part, a->dqscale_part);
    QT_LAUNCH(qt_dq_kernel, grid, *a, (const float *)b.stats, (const float *)b.lse, (const float *)b.delta, (const float *)b.rows);
    QT_LAUNCH(qt_dkv_kernel, grid, *a, (const float *)b.stats, (const float *)b.lse, (const float *)b.delta, (const float *)b.rows);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

}  // namespace mopk
