// Plain scaled-dot-product attention -- fused gfx950 kernels (bf16 MFMA, fp32 accumulate, no N x N map in HBM).
//
// Replaces the attention core of reference BaselineMSA (attention_variants.py:42-46), MSA (components.py:61-64) and
// MultiheadSelfAttention (whisper_mop.py:163-175): causal flag, explicit uint8 mask and additive fp32 bias are applied
// in-kernel.  fp32-exact arithmetic and head dims other than 32 / 64 run the generic path (attn_generic.hip).
//
// Layout ("X layout", as in edgewise_fused.hip): a wave owns 32 queries; a 32x32 score tile is computed TRANSPOSED,
//   S^T[key, query] = K_tile . Q^T            (v_mfma_f32_32x32x16_bf16: A = K rows from LDS, B = q fragments in registers)
// so a lane holds one query and its registers run over keys: the online softmax is a per-lane register loop plus one
// exchange between the lane halves, and the packed probabilities are directly the B operand of
//   O^T[d, query] += V^T[d, key] . P^T[key, query]
// with V^T staged in LDS (key columns permuted to the accumulator's k order).  The backward is two kernels (dQ per
// query block, dK/dV per key block), each recomputing P from the saved row log-sum-exp: deterministic, no atomics.
#include "flash_common.h"

namespace mopk {

namespace {
// second score pair of the dual-path (MultiHopMSA) logits  z = s1 + a2 s2 + g_or (lse(s1,s2) - s1)   (:209-213, chain gate 0)
struct FaDual {
    MopkView4 q2, k2, dq2, dk2;
    float a2, g_or;          // a2 = g_and - beta_not * g_not
};
// logits are kept in base-2 units (scores pre-scaled by log2 e): lse2(s1,s2) = log2(2^s1 + 2^s2) is homogeneous in that scaling
__device__ __forceinline__ float fa_mix(float s1, float s2, float a2, float g_or) {
    float z = fmaf(a2, s2, s1);
    if (g_or != 0.f) {
        const float mx = fmaxf(s1, s2);
        const float l2 = mx + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(s1 - mx) + __builtin_amdgcn_exp2f(s2 - mx));
        z = fmaf(g_or, l2 - s1, z);
    }
    return z;
}
// d z / d s1, d z / d s2
__device__ __forceinline__ void fa_mix_grad(float s1, float s2, float a2, float g_or, float &c1, float &c2) {
    c1 = 1.f; c2 = a2;
    if (g_or != 0.f) {
        const float mx = fmaxf(s1, s2);
        const float e1 = __builtin_amdgcn_exp2f(s1 - mx), e2 = __builtin_amdgcn_exp2f(s2 - mx);
        const float p1 = e1 * __builtin_amdgcn_rcpf(e1 + e2);
        c1 = 1.f - g_or + g_or * p1; c2 = a2 + g_or * (1.f - p1);
    }
}
// optional explicit mask (uint8, 0 = blocked) and additive fp32 bias, element (b,h,i,j)      (whisper_mop.py:166-170, :202-205)
struct FaMB { const uint8_t *mask; const float *bias; };
__device__ __forceinline__ FaMB fa_mb(const MopkSdpaArgs &a, int b, int hh) {
    FaMB m;
    m.mask = a.mask ? a.mask + b * a.mask_sb + hh * a.mask_sh : nullptr;
    m.bias = a.bias ? a.bias + b * a.bias_sb + hh * a.bias_sh : nullptr;
    return m;
}
// one element: logit (base-2 units) -> logit + bias, or FA_NEG when blocked.  Written as selects on unconditionally loaded
// values (indices clamped into range): a per-lane branch around an element write of the accumulator vector is miscompiled
// by hipcc 7.2 (the taken path clobbers the other 15 elements).
__device__ __forceinline__ float fa_apply_mb(float z, const FaMB &m, const MopkSdpaArgs &a, int i, int j, bool &blocked) {
    const int ic = min(i, a.N - 1), jc = min(j, a.N - 1);
    float bz = 0.f;
    unsigned int keep = 1;
    if (m.bias) bz = m.bias[(int64_t)ic * a.bias_si + jc] * FA_LOG2E;          // wave-uniform pointer tests
    if (m.mask) keep = m.mask[(int64_t)ic * a.mask_si + jc];
    blocked = keep == 0;
    return blocked ? FA_NEG : z + bz;
}
}  // namespace

// ------------------------------------------------------------------ forward
template <int DK, typename IOT, bool CAUSAL, bool DUAL, bool MB>
__global__ void __launch_bounds__(FA_NW * 64, DUAL ? 2 : 3) sdpa_flash_fwd_kernel(MopkSdpaArgs a, float *lse, FaDual u) {
    constexpr int DT = DK / 32, LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[FA_KT * LDK], Vt[DK * FA_LDT], K2s[DUAL ? FA_KT * LDK : 8];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.N;
    int qb, bh;
    fa_block_id((N + FA_QB - 1) / FA_QB, qb, bh);
    const int b = bh / a.H, hh = bh % a.H;
    const int q0 = qb * FA_QB, qi = q0 + 32 * w + r;
    const int wu = __builtin_amdgcn_readfirstlane(w);              // the wave index as a scalar (uniform branches on it)
    const bool qok = qi < N;
    const float c = rsqrtf((float)DK) * FA_LOG2E;                  // logits in base-2 units: exp2 without a multiply
    const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh, *vp = (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh;
    bf16x8 qe[DK / 16];
    fa_frags<DK, IOT>(qe, (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn, qok, h, c);
    bf16x8 q2e[DUAL ? DK / 16 : 1];
    const IOT *k2p = nullptr;
    if (DUAL) {
        fa_frags<DK, IOT>(*(bf16x8(*)[DK / 16]) & q2e, (const IOT *)u.q2.ptr + b * u.q2.sb + hh * u.q2.sh + (int64_t)qi * u.q2.sn, qok, h, c);
        k2p = (const IOT *)u.k2.ptr + b * u.k2.sb + hh * u.k2.sh;
    }
    const FaMB mb = fa_mb(a, b, hh);
    const FaDrop drop = fa_drop(a.dropout_p, a.dropout_seed);
    const uint32_t rowh = fa_drop_row(drop, bh, qi);
    float m = FA_NEG, l = 0.f;
    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) O[dt] = fa_zero();
    int nkt = (N + FA_KT - 1) / FA_KT;
    if (CAUSAL) nkt = min(nkt, (min(q0 + FA_QB, N) + FA_KT - 1) / FA_KT);   // keys beyond the block's last query are never seen
    FaTile<DK> fk, fv, fk2;               // next tile's K / V (/ K2) rows, in flight while the current tile is computed
    fa_fetch<DK, IOT>(fk, kp, a.k.sn, 0, N, 1.f, tid);
    fa_fetch<DK, IOT>(fv, vp, a.v.sn, 0, N, 1.f, tid);
    if (DUAL) fa_fetch<DK, IOT>(fk2, k2p, u.k2.sn, 0, N, 1.f, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * FA_KT;
        __syncthreads();
        fa_put<DK, true, false>(Ks, nullptr, fk, tid);
        fa_put<DK, false, true>(nullptr, Vt, fv, tid);
        if (DUAL) fa_put<DK, true, false>(K2s, nullptr, fk2, tid);
        if (kt + 1 < nkt) {
            fa_fetch<DK, IOT>(fk, kp, a.k.sn, k0 + FA_KT, N, 1.f, tid);
            fa_fetch<DK, IOT>(fv, vp, a.v.sn, k0 + FA_KT, N, 1.f, tid);
            if (DUAL) fa_fetch<DK, IOT>(fk2, k2p, u.k2.sn, k0 + FA_KT, N, 1.f, tid);
        }
        __syncthreads();
        f32x16 S[2];
        float mx = FA_NEG;
        const bool edge = k0 + FA_KT > N || (CAUSAL && k0 + FA_KT - 1 > q0 + 32 * wu);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            S[s2] = fa_mm_rows<DK>(Ks, 32 * s2, r, h, qe);
            if (DUAL) {
                const f32x16 T2 = fa_mm_rows<DK>(K2s, 32 * s2, r, h, *(const bf16x8(*)[DK / 16]) & q2e);
#pragma unroll
                for (int g = 0; g < 16; ++g) S[s2][g] = fa_mix(S[s2][g], T2[g], u.a2, u.g_or);
            }
            if (MB) {
#pragma unroll
                for (int g = 0; g < 16; ++g) { bool blk; S[s2][g] = fa_apply_mb(S[s2][g], mb, a, qi, k0 + 32 * s2 + tile_row(g, h), blk); }
            }
            if (edge) {                         // wave-uniform: only tiles that touch the end of the keys or this wave's diagonal pay for the mask
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int j = k0 + 32 * s2 + tile_row(g, h);
                    S[s2][g] = (j >= N || (CAUSAL && j > qi)) ? FA_NEG : S[s2][g];      // select, not a branch around the element write
                }
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) mx = fmaxf(mx, S[s2][g]);
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
        if (drop.thresh) {                      // the row sum above is of the undropped probabilities; P V sees keep / (1 - p)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int g = 0; g < 16; ++g) S[s2][g] = fa_drop_keep(drop, rowh, k0 + 32 * s2 + tile_row(g, h)) ? S[s2][g] * drop.inv_keep : 0.f;
        }
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
// delta_i = sum_d dy[i,d] y[i,d]  ( = sum_j P_ij dP_ij ); dk / 8 lanes per row, 16-byte loads (one wave per row moved 2 bytes per lane)
__device__ __forceinline__ void fa_ld8(const unsigned short *p, float (&v)[8]) {
    const bf16x8 x = *(const bf16x8 *)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f((unsigned short)x[e]);
}
__device__ __forceinline__ void fa_ld8(const float *p, float (&v)[8]) {
    const float4 a = *(const float4 *)p, b = *(const float4 *)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename IOT>
__global__ void sdpa_flash_delta_kernel(MopkSdpaArgs a, float *delta) {
    const int cpr = a.dk >> 3;                                  // 16-byte chunks (= lanes) per row: 4 or 8
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / cpr;
    const int ch = threadIdx.x % cpr;
    float s = 0.f;
    if (row < (int64_t)a.B * a.H * a.N) {
        const int i = row % a.N;
        const int64_t bh = row / a.N;
        const int b = bh / a.H, hh = bh % a.H;
        float y[8], g[8];
        fa_ld8((const IOT *)a.y.ptr + b * a.y.sb + hh * a.y.sh + (int64_t)i * a.y.sn + 8 * ch, y);
        fa_ld8((const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh + (int64_t)i * a.dy.sn + 8 * ch, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(y[e], g[e], s);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (cpr == 8) s += __shfl_xor(s, 4, 64);
    if (ch == 0 && row < (int64_t)a.B * a.H * a.N) delta[row] = s;
}

// dQ: one workgroup per 128 queries, loop over key tiles
template <int DK, typename IOT, bool CAUSAL, bool DUAL, bool MB>
__global__ void __launch_bounds__(FA_NW * 64, (DUAL || MB) ? 2 : 3) sdpa_flash_dq_kernel(MopkSdpaArgs a, const float *lse, const float *delta, FaDual u) {
    constexpr int DT = DK / 32, LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[FA_KT * LDK], Vs[FA_KT * LDK], Kt[DK * FA_LDT];
    __shared__ __attribute__((aligned(16))) unsigned short K2s[DUAL ? FA_KT * LDK : 8], K2t[DUAL ? DK * FA_LDT : 8];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.N;
    int qb, bh;
    fa_block_id((N + FA_QB - 1) / FA_QB, qb, bh);
    const int b = bh / a.H, hh = bh % a.H;
    const int q0 = qb * FA_QB, qi = q0 + 32 * w + r;
    const bool qok = qi < N;
    const float sc = rsqrtf((float)DK), c = sc * FA_LOG2E;
    const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh, *vp = (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh;
    bf16x8 qe[DK / 16], dof[DK / 16];
    fa_frags<DK, IOT>(qe, (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn, qok, h, c);
    fa_frags<DK, IOT>(dof, (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh + (int64_t)qi * a.dy.sn, qok, h, 1.f);
    bf16x8 q2e[DUAL ? DK / 16 : 1];
    const IOT *k2p = nullptr;
    if (DUAL) {
        fa_frags<DK, IOT>(*(bf16x8(*)[DK / 16]) & q2e, (const IOT *)u.q2.ptr + b * u.q2.sb + hh * u.q2.sh + (int64_t)qi * u.q2.sn, qok, h, c);
        k2p = (const IOT *)u.k2.ptr + b * u.k2.sb + hh * u.k2.sh;
    }
    const float Li = qok ? lse[(int64_t)bh * N + qi] : 0.f, di = qok ? delta[(int64_t)bh * N + qi] : 0.f;
    const FaMB mb = fa_mb(a, b, hh);
    const FaDrop drop = fa_drop(a.dropout_p, a.dropout_seed);
    const uint32_t rowh = fa_drop_row(drop, bh, qi);
    f32x16 dQ[DT], dQ2[DUAL ? DT : 1];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dQ[dt] = fa_zero(); if (DUAL) dQ2[dt] = fa_zero(); }
    int nkt = (N + FA_KT - 1) / FA_KT;
    if (CAUSAL) nkt = min(nkt, (min(q0 + FA_QB, N) + FA_KT - 1) / FA_KT);
    FaTile<DK> fk, fv, fk2;
    fa_fetch<DK, IOT>(fk, kp, a.k.sn, 0, N, 1.f, tid);
    fa_fetch<DK, IOT>(fv, vp, a.v.sn, 0, N, 1.f, tid);
    if (DUAL) fa_fetch<DK, IOT>(fk2, k2p, u.k2.sn, 0, N, 1.f, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * FA_KT;
        __syncthreads();
        fa_put<DK, true, true>(Ks, Kt, fk, tid);
        fa_put<DK, true, false>(Vs, nullptr, fv, tid);
        if (DUAL) fa_put<DK, true, true>(K2s, K2t, fk2, tid);
        if (kt + 1 < nkt) {
            fa_fetch<DK, IOT>(fk, kp, a.k.sn, k0 + FA_KT, N, 1.f, tid);
            fa_fetch<DK, IOT>(fv, vp, a.v.sn, k0 + FA_KT, N, 1.f, tid);
            if (DUAL) fa_fetch<DK, IOT>(fk2, k2p, u.k2.sn, k0 + FA_KT, N, 1.f, tid);
        }
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const f32x16 S = fa_mm_rows<DK>(Ks, 32 * s2, r, h, qe);
            const f32x16 dP = fa_mm_rows<DK>(Vs, 32 * s2, r, h, dof);
            f32x16 T2 = fa_zero();
            if (DUAL) T2 = fa_mm_rows<DK>(K2s, 32 * s2, r, h, *(const bf16x8(*)[DK / 16]) & q2e);
            f32x16 dS, dS2;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int j = k0 + 32 * s2 + tile_row(g, h);
                bool ok = qok && j < N && (!CAUSAL || j <= qi);
                float z = DUAL ? fa_mix(S[g], T2[g], u.a2, u.g_or) : S[g];
                if (MB) { bool blk; z = fa_apply_mb(z, mb, a, qi, j, blk); ok = ok && !blk; }
                const float p = ok ? __builtin_amdgcn_exp2f(z - Li) : 0.f;
                float dp = dP[g];                                // dropout: dP = (dy v^T) keep / (1 - p); delta = dy . y already has it
                if (drop.thresh) dp = fa_drop_keep(drop, rowh, j) ? dp * drop.inv_keep : 0.f;
                const float dz = p * (dp - di) * sc;             // d logits / sqrt(dk)
                if (DUAL) { float c1, c2; fa_mix_grad(S[g], T2[g], u.a2, u.g_or, c1, c2); dS[g] = dz * c1; dS2[g] = dz * c2; }
                else dS[g] = dz;
            }
            bf16x8 lo, hi;
            fa_pack(lo, hi, dS);
            fa_mm_cols<DK>(dQ, Kt, s2, r, h, lo, hi);
            if (DUAL) { fa_pack(lo, hi, dS2); fa_mm_cols<DK>(*(f32x16(*)[DT]) & dQ2, K2t, s2, r, h, lo, hi); }
        }
    }
    if (qok) {
        fa_store_rows<DK, IOT>((IOT *)a.dq.ptr + b * a.dq.sb + hh * a.dq.sh + (int64_t)qi * a.dq.sn, dQ, h, 1.f);
        if (DUAL) fa_store_rows<DK, IOT>((IOT *)u.dq2.ptr + b * u.dq2.sb + hh * u.dq2.sh + (int64_t)qi * u.dq2.sn, *(const f32x16(*)[DT]) & dQ2, h, 1.f);
    }
}

// dK, dV: one workgroup per 128 keys (a lane owns a key), loop over query tiles
template <int DK, typename IOT, bool CAUSAL, bool DUAL, bool MB>
__global__ void __launch_bounds__(FA_NW * 64, DUAL ? 1 : 2) sdpa_flash_dkv_kernel(MopkSdpaArgs a, const float *lse, const float *delta, FaDual u) {
    constexpr int DT = DK / 32, LDK = DK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short Qs[FA_KT * LDK], Gs[FA_KT * LDK], Qt[DK * FA_LDT], Gt[DK * FA_LDT];
    __shared__ __attribute__((aligned(16))) unsigned short Q2s[DUAL ? FA_KT * LDK : 8], Q2t[DUAL ? DK * FA_LDT : 8];
    __shared__ __attribute__((aligned(16))) float Ls[FA_KT], Ds[FA_KT];      // 16-byte aligned: the four consecutive rows of a register quad are one ds_read_b128
    __shared__ __attribute__((aligned(16))) uint32_t Hs[FA_KT];              // dropout row hashes of the tile's queries
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.N;
    int kb, bh;
    fa_block_id((N + FA_QB - 1) / FA_QB, kb, bh);
    const int b = bh / a.H, hh = bh % a.H;
    const int k0 = kb * FA_QB, kj = k0 + 32 * w + r;
    const bool kok = kj < N;
    const float c = rsqrtf((float)DK) * FA_LOG2E;
    const IOT *qp = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh, *gp = (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh;
    bf16x8 kf[DK / 16], vf[DK / 16];
    fa_frags<DK, IOT>(kf, (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh + (int64_t)kj * a.k.sn, kok, h, 1.f);
    fa_frags<DK, IOT>(vf, (const IOT *)a.v.ptr + b * a.v.sb + hh * a.v.sh + (int64_t)kj * a.v.sn, kok, h, 1.f);
    bf16x8 k2f[DUAL ? DK / 16 : 1];
    const IOT *q2p = nullptr;
    if (DUAL) {
        fa_frags<DK, IOT>(*(bf16x8(*)[DK / 16]) & k2f, (const IOT *)u.k2.ptr + b * u.k2.sb + hh * u.k2.sh + (int64_t)kj * u.k2.sn, kok, h, 1.f);
        q2p = (const IOT *)u.q2.ptr + b * u.q2.sb + hh * u.q2.sh;
    }
    f32x16 dK[DT], dV[DT], dK2[DUAL ? DT : 1];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dK[dt] = fa_zero(); dV[dt] = fa_zero(); if (DUAL) dK2[dt] = fa_zero(); }
    const FaMB mb = fa_mb(a, b, hh);
    const FaDrop drop = fa_drop(a.dropout_p, a.dropout_seed);
    const int nqt = (N + FA_KT - 1) / FA_KT, qt0 = CAUSAL ? k0 / FA_KT : 0;   // causal: queries before this key block see none of its keys
    FaTile<DK> fq, fg, fq2;               // q is pre-scaled exactly as the forward's fragments
    fa_fetch<DK, IOT>(fq, qp, a.q.sn, qt0 * FA_KT, N, c, tid);
    fa_fetch<DK, IOT>(fg, gp, a.dy.sn, qt0 * FA_KT, N, 1.f, tid);
    if (DUAL) fa_fetch<DK, IOT>(fq2, q2p, u.q2.sn, qt0 * FA_KT, N, c, tid);
    // the tile's row statistics travel with the prefetch too (loaded between the barriers they cost one exposed round trip per tile)
    float nl = 0.f, nd = 0.f;
    auto fetch_stats = [&](int i0) {
        const bool ok = tid < FA_KT && i0 + tid < N;
        nl = ok ? lse[(int64_t)bh * N + i0 + tid] : 0.f;
        nd = ok ? delta[(int64_t)bh * N + i0 + tid] : 0.f;
    };
    fetch_stats(qt0 * FA_KT);
    for (int qt = qt0; qt < nqt; ++qt) {
        const int i0 = qt * FA_KT;
        __syncthreads();
        fa_put<DK, true, true>(Qs, Qt, fq, tid);
        fa_put<DK, true, true>(Gs, Gt, fg, tid);
        if (DUAL) fa_put<DK, true, true>(Q2s, Q2t, fq2, tid);
        if (tid < FA_KT) { Ls[tid] = nl; Ds[tid] = nd; Hs[tid] = fa_drop_row(drop, bh, i0 + tid); }
        if (qt + 1 < nqt) {
            fa_fetch<DK, IOT>(fq, qp, a.q.sn, i0 + FA_KT, N, c, tid);
            fa_fetch<DK, IOT>(fg, gp, a.dy.sn, i0 + FA_KT, N, 1.f, tid);
            if (DUAL) fa_fetch<DK, IOT>(fq2, q2p, u.q2.sn, i0 + FA_KT, N, c, tid);
            fetch_stats(i0 + FA_KT);
        }
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const f32x16 S = fa_mm_rows<DK>(Qs, 32 * s2, r, h, kf);       // rows = queries, lane = key
            const f32x16 dP = fa_mm_rows<DK>(Gs, 32 * s2, r, h, vf);
            f32x16 T2 = fa_zero();
            if (DUAL) T2 = fa_mm_rows<DK>(Q2s, 32 * s2, r, h, *(const bf16x8(*)[DK / 16]) & k2f);
            f32x16 P, dS, dS2;
            float lrow[16], drow[16];                                // row statistics of the tile's 16 registers: rows 8 q + 4 h + {0..3} per quad q
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 l4 = *(const float4 *)&Ls[32 * s2 + 8 * q4 + 4 * h], d4 = *(const float4 *)&Ds[32 * s2 + 8 * q4 + 4 * h];
                lrow[4 * q4] = l4.x; lrow[4 * q4 + 1] = l4.y; lrow[4 * q4 + 2] = l4.z; lrow[4 * q4 + 3] = l4.w;
                drow[4 * q4] = d4.x; drow[4 * q4 + 1] = d4.y; drow[4 * q4 + 2] = d4.z; drow[4 * q4 + 3] = d4.w;
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int il = 32 * s2 + tile_row(g, h), i = i0 + il;
                bool ok = kok && i < N && (!CAUSAL || kj <= i);
                float z = DUAL ? fa_mix(S[g], T2[g], u.a2, u.g_or) : S[g];
                if (MB) { bool blk; z = fa_apply_mb(z, mb, a, i, kj, blk); ok = ok && !blk; }
                const float p = ok ? __builtin_amdgcn_exp2f(z - lrow[g]) : 0.f;
                float dp = dP[g], pd = p;                         // dropout: dV sees P keep / (1 - p), dP = (dy v^T) keep / (1 - p)
                if (drop.thresh) { const float kp = fa_drop_keep(drop, Hs[il], kj) ? drop.inv_keep : 0.f; dp *= kp; pd *= kp; }
                P[g] = pd;
                const float dz = p * (dp - drow[g]) * FA_LN2;     // Q' = q log2(e)/sqrt(dk)  ->  dK = (dS ln2)^T Q'
                if (DUAL) { float c1, c2; fa_mix_grad(S[g], T2[g], u.a2, u.g_or, c1, c2); dS[g] = dz * c1; dS2[g] = dz * c2; }
                else dS[g] = dz;
            }
            bf16x8 lo, hi;
            fa_pack(lo, hi, P);
            fa_mm_cols<DK>(dV, Gt, s2, r, h, lo, hi);
            fa_pack(lo, hi, dS);
            fa_mm_cols<DK>(dK, Qt, s2, r, h, lo, hi);
            if (DUAL) { fa_pack(lo, hi, dS2); fa_mm_cols<DK>(*(f32x16(*)[DT]) & dK2, Q2t, s2, r, h, lo, hi); }
        }
    }
    if (kok) {
        fa_store_rows<DK, IOT>((IOT *)a.dk_.ptr + b * a.dk_.sb + hh * a.dk_.sh + (int64_t)kj * a.dk_.sn, dK, h, 1.f);
        fa_store_rows<DK, IOT>((IOT *)a.dv.ptr + b * a.dv.sb + hh * a.dv.sh + (int64_t)kj * a.dv.sn, dV, h, 1.f);
        if (DUAL) fa_store_rows<DK, IOT>((IOT *)u.dk2.ptr + b * u.dk2.sb + hh * u.dk2.sh + (int64_t)kj * u.dk2.sn, *(const f32x16(*)[DT]) & dK2, h, 1.f);
    }
}

// ------------------------------------------------------------------ host side
static bool fa_aligned(const MopkView4 &v, int es) {
    const int64_t al = 16 / es;
    return ((uintptr_t)v.ptr & 15) == 0 && v.sb % al == 0 && v.sh % al == 0 && v.sn % al == 0;
}
int sdpa_flash_supported(const MopkSdpaArgs *a, bool bwd) {
    if (a->precision != MOPK_PREC_BF16) return 0;                 // fp32-exact arithmetic stays on the generic path
    if (a->dk != 32 && a->dk != 64) return 0;
    const int es = a->io_dtype == MOPK_BF16 ? 2 : 4;
    if (!fa_aligned(a->q, es) || !fa_aligned(a->k, es) || !fa_aligned(a->v, es) || !fa_aligned(a->y, es)) return 0;
    if (bwd && (!fa_aligned(a->dy, es) || !fa_aligned(a->dq, es) || !fa_aligned(a->dk_, es) || !fa_aligned(a->dv, es))) return 0;
    return 1;
}
size_t sdpa_flash_saved_bytes(const MopkSdpaArgs *a) { return (size_t)a->B * a->H * a->N * sizeof(float) + 256; }   // row log-sum-exp
size_t sdpa_flash_ws_bytes(const MopkSdpaArgs *a) { return (size_t)a->B * a->H * a->N * sizeof(float) + 256; }      // delta

#define FA_LAUNCH4(KERNEL, DK_, IOT_, DUAL_, MB_, GRID, ...)                                                   \
    do { if (a->causal) hipLaunchKernelGGL((KERNEL<DK_, IOT_, true, DUAL_, MB_>), GRID, dim3(FA_NW * 64), 0, st, __VA_ARGS__);   \
         else hipLaunchKernelGGL((KERNEL<DK_, IOT_, false, DUAL_, MB_>), GRID, dim3(FA_NW * 64), 0, st, __VA_ARGS__); } while (0)
#define FA_DISPATCH5(KERNEL, DUAL_, MB_, GRID, ...)                                                            \
    do {                                                                                                       \
        if (a->io_dtype == MOPK_BF16) { if (a->dk == 64) FA_LAUNCH4(KERNEL, 64, unsigned short, DUAL_, MB_, GRID, __VA_ARGS__);   \
                                        else FA_LAUNCH4(KERNEL, 32, unsigned short, DUAL_, MB_, GRID, __VA_ARGS__); }            \
        else { if (a->dk == 64) FA_LAUNCH4(KERNEL, 64, float, DUAL_, MB_, GRID, __VA_ARGS__);                    \
               else FA_LAUNCH4(KERNEL, 32, float, DUAL_, MB_, GRID, __VA_ARGS__); }                              \
    } while (0)
// explicit mask / bias tensors: MB instantiations (plain and DUAL: the mask acts on the mixed logits, attention_variants.py:219-220)
#define FA_DISPATCH(KERNEL, DUAL_, GRID, ...)                                                                  \
    do { if (a->mask || a->bias) FA_DISPATCH5(KERNEL, DUAL_, true, GRID, __VA_ARGS__);                          \
         else FA_DISPATCH5(KERNEL, DUAL_, false, GRID, __VA_ARGS__); } while (0)

int sdpa_flash_fwd(const MopkSdpaArgs *a, hipStream_t st) {
    if (!sdpa_flash_supported(a, false)) return MOPK_ERR_UNSUPPORTED;
    const dim3 grid(((a->N + FA_QB - 1) / FA_QB) * a->B * a->H);
    FA_DISPATCH(sdpa_flash_fwd_kernel, false, grid, *a, (float *)a->saved, FaDual{});
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}
int sdpa_flash_bwd(const MopkSdpaArgs *a, hipStream_t st) {
    if (!sdpa_flash_supported(a, true)) return MOPK_ERR_UNSUPPORTED;
    const int64_t rows = (int64_t)a->B * a->H * a->N;
    float *delta = (float *)a->workspace;
    if (a->io_dtype == MOPK_BF16) hipLaunchKernelGGL((sdpa_flash_delta_kernel<unsigned short>), dim3((rows * (a->dk / 8) + 255) / 256), dim3(256), 0, st, *a, delta);
    else hipLaunchKernelGGL((sdpa_flash_delta_kernel<float>), dim3((rows * (a->dk / 8) + 255) / 256), dim3(256), 0, st, *a, delta);
    MOPK_CHECK_LAUNCH();
    const dim3 grid(((a->N + FA_QB - 1) / FA_QB) * a->B * a->H);
    FA_DISPATCH(sdpa_flash_dq_kernel, false, grid, *a, (const float *)a->saved, (const float *)delta, FaDual{});
    MOPK_CHECK_LAUNCH();
    FA_DISPATCH(sdpa_flash_dkv_kernel, false, grid, *a, (const float *)a->saved, (const float *)delta, FaDual{});
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

// ------------------------------------------------------------------ dual path (MultiHopMSA) on the fused kernels
// y = softmax(S1 + a2 S2 + g_or (lse(S1,S2) - S1)) v1 + sigmoid(w) A1 A2^(hops-1) v2        (attention_variants.py:200-229)
// with the chain gate at 0 and no mask tensor: one DUAL pass for the mixed logits, the transport term as chained
// plain passes (A2 v2, A2 (A2 v2), ..., A1 (..)); backward = the same passes in reverse plus sums of the q/k gradients.
namespace {
struct DpLayout { float *lse_m, *lse_1, *lse_2; void *O1, *yc, *tr; size_t bytes; };   // tr: (hops-1) tensors
static DpLayout dp_saved(void *p, const MopkDualPathArgs *a) {
    Carver c(p);
    const size_t rows = (size_t)a->B * a->H * a->N, nd = rows * a->dk, es = a->io_dtype == MOPK_BF16 ? 2 : 4;
    DpLayout L;
    L.lse_m = c.take<float>(rows); L.lse_1 = c.take<float>(rows); L.lse_2 = c.take<float>(rows);
    L.O1 = c.take<char>(nd * es); L.yc = c.take<char>(nd * es); L.tr = c.take<char>((size_t)(a->hops > 1 ? a->hops - 1 : 0) * ((nd * es + 255) & ~(size_t)255));
    L.bytes = c.off;
    return L;
}
struct DpWork { float *delta; void *g, *tq, *tk, *ta, *tb; size_t bytes; };
static DpWork dp_work(void *p, const MopkDualPathArgs *a) {
    Carver c(p);
    const size_t rows = (size_t)a->B * a->H * a->N, nd = rows * a->dk, es = a->io_dtype == MOPK_BF16 ? 2 : 4;
    DpWork W;
    W.delta = c.take<float>(rows);
    W.g = c.take<char>(nd * es); W.tq = c.take<char>(nd * es); W.tk = c.take<char>(nd * es); W.ta = c.take<char>(nd * es); W.tb = c.take<char>(nd * es);
    W.bytes = c.off;
    return W;
}
static MopkView4 dp_tmp_view(void *p, const MopkDualPathArgs *a) {     // contiguous (B,N,H,dk)
    return MopkView4{p, (int64_t)a->N * a->H * a->dk, (int64_t)a->dk, (int64_t)a->H * a->dk};
}
static MopkSdpaArgs dp_sdpa(const MopkDualPathArgs *a) {
    MopkSdpaArgs s{};
    s.B = a->B; s.H = a->H; s.N = a->N; s.dk = a->dk; s.io_dtype = a->io_dtype; s.precision = MOPK_PREC_BF16; s.path = MOPK_PATH_FUSED;
    s.causal = a->causal;
    s.mask = a->mask; s.mask_sb = a->mask_sb; s.mask_sh = a->mask_sh; s.mask_si = a->mask_si;      // every pass (A1, A2 and the mixed weights) sees it: :202-207, :219-220
    return s;
}
// element (b,h,n,d) of a strided view
template <typename T> __device__ __forceinline__ T *dp_at(const MopkView4 &v, int64_t idx, int H, int N, int dk) {
    const int d = idx % dk; const int n = (idx / dk) % N; const int64_t bh = idx / ((int64_t)dk * N);
    return (T *)v.ptr + (bh / H) * v.sb + (bh % H) * v.sh + (int64_t)n * v.sn + d;
}
// the three element-wise passes of the dual path work on 8 consecutive features per thread (16-byte accesses; dk is 32 or 64 and the
// views are 16-byte aligned on this path): idx8 indexes the (b,h,n,d/8) chunks
__device__ __forceinline__ void fa_st8(unsigned short *p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (short)f2bf(v[e]);
    *(bf16x8 *)p = x;
}
__device__ __forceinline__ void fa_st8(float *p, const float (&v)[8]) {
    *(float4 *)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4 *)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <typename T>
__global__ void dp_combine_kernel(MopkView4 y, MopkView4 o1, MopkView4 yc, const float *logit, int H, int N, int dk, int64_t total) {
    const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (idx >= total) return;
    const float w = 1.f / (1.f + __expf(-*logit));
    float a[8], c[8];
    fa_ld8(dp_at<T>(o1, idx, H, N, dk), a);
    fa_ld8(dp_at<T>(yc, idx, H, N, dk), c);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = a[e] + w * c[e];                                                                    // :229
    fa_st8(dp_at<T>(y, idx, H, N, dk), a);
}
template <typename T>
__global__ void dp_scale_kernel(MopkView4 out, MopkView4 in, const float *logit, int H, int N, int dk, int64_t total) {
    const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (idx >= total) return;
    const float w = 1.f / (1.f + __expf(-*logit));
    float a[8];
    fa_ld8(dp_at<T>(in, idx, H, N, dk), a);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= w;
    fa_st8(dp_at<T>(out, idx, H, N, dk), a);
}
template <typename T>
__global__ void dp_add_kernel(MopkView4 out, MopkView4 in, int H, int N, int dk, int64_t total) {
    const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (idx >= total) return;
    T *o = dp_at<T>(out, idx, H, N, dk);
    float a[8], c[8];
    fa_ld8(o, a);
    fa_ld8(dp_at<T>(in, idx, H, N, dk), c);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += c[e];
    fa_st8(o, a);
}
template <typename T>
__global__ void dp_dlogit_kernel(MopkView4 dy, MopkView4 yc, const float *logit, float *out, int H, int N, int dk) {
    __shared__ float red[256];
    const int64_t bh = blockIdx.x, per = (int64_t)N * dk;
    float sm = 0.f;
    for (int64_t i = threadIdx.x; i < per; i += 256) sm += ld_as_f32(dp_at<T>(dy, bh * per + i, H, N, dk)) * ld_as_f32(dp_at<T>(yc, bh * per + i, H, N, dk));
    red[threadIdx.x] = sm; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { const float w = 1.f / (1.f + __expf(-*logit)); out[bh] = red[0] * w * (1.f - w); }
}
#define DP_ELEM(KERNEL, ...)                                                                                        \
    do { const int64_t tot_ = (int64_t)a->B * a->H * a->N * a->dk, thr_ = tot_ / 8;                                \
         if (a->io_dtype == MOPK_BF16) hipLaunchKernelGGL((KERNEL<unsigned short>), dim3((thr_ + 255) / 256), dim3(256), 0, st, __VA_ARGS__, a->H, a->N, a->dk, tot_); \
         else hipLaunchKernelGGL((KERNEL<float>), dim3((thr_ + 255) / 256), dim3(256), 0, st, __VA_ARGS__, a->H, a->N, a->dk, tot_); } while (0)
}  // namespace

int dp_flash_supported(const MopkDualPathArgs *a, bool bwd) {
    if (a->g_chain != 0.f) return 0;                               // chain gate needs the N x N product A1 A2^(h-1): generic path
    MopkSdpaArgs s = dp_sdpa(a);
    s.q = a->q1; s.k = a->k1; s.v = a->v1; s.y = a->y;
    if (bwd) { s.dy = a->dy; s.dq = a->dq1; s.dk_ = a->dk1; s.dv = a->dv1; }
    if (a->precision != MOPK_PREC_BF16 || !sdpa_flash_supported(&s, bwd)) return 0;
    s.q = a->q2; s.k = a->k2;
    if (bwd) { s.dq = a->dq2; s.dk_ = a->dk2; }
    if (a->hops > 0) { s.v = a->v2; if (bwd) s.dv = a->dv2; }      // hops == 0: no transport term, v2 / dv2 are not touched
    return sdpa_flash_supported(&s, bwd);
}
size_t dp_flash_saved_bytes(const MopkDualPathArgs *a) { return dp_saved(nullptr, a).bytes + 256; }
size_t dp_flash_ws_bytes(const MopkDualPathArgs *a) { return dp_work(nullptr, a).bytes + 256; }

static int dp_plain_fwd(const MopkDualPathArgs *d, const MopkView4 &q, const MopkView4 &k, const MopkView4 &v, const MopkView4 &y, float *lse, hipStream_t st) {
    MopkSdpaArgs s = dp_sdpa(d);
    s.q = q; s.k = k; s.v = v; s.y = y;
    const MopkSdpaArgs *a = &s;
    const dim3 grid(((a->N + FA_QB - 1) / FA_QB) * a->B * a->H);
    FA_DISPATCH(sdpa_flash_fwd_kernel, false, grid, *a, lse, FaDual{});
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}
static int dp_plain_bwd(const MopkDualPathArgs *d, const MopkView4 &q, const MopkView4 &k, const MopkView4 &v, const MopkView4 &y,
                        const MopkView4 &dy, const MopkView4 &dq, const MopkView4 &dk, const MopkView4 &dv, const float *lse, float *delta,
                        hipStream_t st) {
    MopkSdpaArgs s = dp_sdpa(d);
    s.q = q; s.k = k; s.v = v; s.y = y; s.dy = dy; s.dq = dq; s.dk_ = dk; s.dv = dv;
    const MopkSdpaArgs *a = &s;
    const int64_t rows = (int64_t)a->B * a->H * a->N;
    if (a->io_dtype == MOPK_BF16) hipLaunchKernelGGL((sdpa_flash_delta_kernel<unsigned short>), dim3((rows * (a->dk / 8) + 255) / 256), dim3(256), 0, st, *a, delta);
    else hipLaunchKernelGGL((sdpa_flash_delta_kernel<float>), dim3((rows * (a->dk / 8) + 255) / 256), dim3(256), 0, st, *a, delta);
    const dim3 grid(((a->N + FA_QB - 1) / FA_QB) * a->B * a->H);
    FA_DISPATCH(sdpa_flash_dq_kernel, false, grid, *a, lse, (const float *)delta, FaDual{});
    FA_DISPATCH(sdpa_flash_dkv_kernel, false, grid, *a, lse, (const float *)delta, FaDual{});
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

int dp_flash_fwd(const MopkDualPathArgs *a, hipStream_t st) {
    if (!dp_flash_supported(a, false)) return MOPK_ERR_UNSUPPORTED;
    const DpLayout L = dp_saved(a->saved, a);
    const size_t es = a->io_dtype == MOPK_BF16 ? 2 : 4, trs = ((size_t)a->B * a->H * a->N * a->dk * es + 255) & ~(size_t)255;
    const MopkView4 o1 = a->hops > 0 ? dp_tmp_view(L.O1, a) : a->y, yc = dp_tmp_view(L.yc, a);
    {   // mixed logits: S1 + a2 S2 + g_or (lse - S1)                                           :209-213, :219-221
        MopkSdpaArgs s = dp_sdpa(a);
        s.q = a->q1; s.k = a->k1; s.v = a->v1; s.y = o1;
        s.dropout_p = a->dropout_p; s.dropout_seed = a->dropout_seed;             // attn_drop on the mixed weights only (:222)
        FaDual u{a->q2, a->k2, MopkView4{}, MopkView4{}, a->g_and - a->beta_not * a->g_not, a->g_or};
        const dim3 grid(((a->N + FA_QB - 1) / FA_QB) * a->B * a->H);
        const MopkSdpaArgs *keep = a ? &s : nullptr;
        { const MopkSdpaArgs *a = keep; FA_DISPATCH(sdpa_flash_fwd_kernel, true, grid, *a, L.lse_m, u); }
        MOPK_CHECK_LAUNCH();
    }
    if (a->hops == 0) return MOPK_OK;                              // two-score attention only (CrossViewMixerMSA's fused path)
    MopkView4 t = a->v2;                                           // value transport A1 A2^(hops-1) v2   :224-227
    for (int i = 1; i < a->hops; ++i) {
        const MopkView4 nt = dp_tmp_view((char *)L.tr + (size_t)(i - 1) * trs, a);
        int rc = dp_plain_fwd(a, a->q2, a->k2, t, nt, L.lse_2, st); if (rc) return rc;
        t = nt;
    }
    int rc = dp_plain_fwd(a, a->q1, a->k1, t, yc, L.lse_1, st); if (rc) return rc;
    DP_ELEM(dp_combine_kernel, a->y, o1, yc, a->chain_logit);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

int dp_flash_bwd(const MopkDualPathArgs *a, hipStream_t st) {
    if (!dp_flash_supported(a, true)) return MOPK_ERR_UNSUPPORTED;
    const DpLayout L = dp_saved(a->saved, a);
    const DpWork W = dp_work(a->workspace, a);
    const size_t es = a->io_dtype == MOPK_BF16 ? 2 : 4, trs = ((size_t)a->B * a->H * a->N * a->dk * es + 255) & ~(size_t)255;
    const MopkView4 o1 = a->hops > 0 ? dp_tmp_view(L.O1, a) : a->y, yc = dp_tmp_view(L.yc, a), g = dp_tmp_view(W.g, a),
                    tq = dp_tmp_view(W.tq, a), tk = dp_tmp_view(W.tk, a);
    auto tr = [&](int i) { return i == 0 ? a->v2 : dp_tmp_view((char *)L.tr + (size_t)(i - 1) * trs, a); };
    if (a->hops == 0) hipMemsetAsync(a->dlogit_part, 0, sizeof(float) * a->B * a->H, st);
    else if (a->io_dtype == MOPK_BF16) hipLaunchKernelGGL((dp_dlogit_kernel<unsigned short>), dim3(a->B * a->H), dim3(256), 0, st, a->dy, yc, a->chain_logit, a->dlogit_part, a->H, a->N, a->dk);
    else hipLaunchKernelGGL((dp_dlogit_kernel<float>), dim3(a->B * a->H), dim3(256), 0, st, a->dy, yc, a->chain_logit, a->dlogit_part, a->H, a->N, a->dk);
    if (a->hops > 0) DP_ELEM(dp_scale_kernel, g, a->dy, a->chain_logit);   // gradient reaching the transport term
    MOPK_CHECK_LAUNCH();
    {   // mixed-logit pass: dq1, dk1, dv1, dq2, dk2 straight into the caller's tensors
        MopkSdpaArgs s = dp_sdpa(a);
        s.q = a->q1; s.k = a->k1; s.v = a->v1; s.y = o1; s.dy = a->dy; s.dq = a->dq1; s.dk_ = a->dk1; s.dv = a->dv1;
        s.dropout_p = a->dropout_p; s.dropout_seed = a->dropout_seed;
        FaDual u{a->q2, a->k2, a->dq2, a->dk2, a->g_and - a->beta_not * a->g_not, a->g_or};
        const int64_t rows = (int64_t)a->B * a->H * a->N;
        if (a->io_dtype == MOPK_BF16) hipLaunchKernelGGL((sdpa_flash_delta_kernel<unsigned short>), dim3((rows * (s.dk / 8) + 255) / 256), dim3(256), 0, st, s, W.delta);
        else hipLaunchKernelGGL((sdpa_flash_delta_kernel<float>), dim3((rows * (s.dk / 8) + 255) / 256), dim3(256), 0, st, s, W.delta);
        const dim3 grid(((a->N + FA_QB - 1) / FA_QB) * a->B * a->H);
        const MopkSdpaArgs *keep = &s;
        { const MopkSdpaArgs *a = keep; FA_DISPATCH(sdpa_flash_dq_kernel, true, grid, *a, (const float *)L.lse_m, (const float *)W.delta, u);
          FA_DISPATCH(sdpa_flash_dkv_kernel, true, grid, *a, (const float *)L.lse_m, (const float *)W.delta, u); }
        MOPK_CHECK_LAUNCH();
    }
    if (a->hops == 0) return MOPK_OK;
    // transport term backwards: yc = A1 t_{h-1}, t_i = A2 t_{i-1}, t_0 = v2
    const int hm = a->hops - 1;
    MopkView4 cur = hm >= 1 ? dp_tmp_view(W.ta, a) : a->dv2;       // gradient wrt t_{h-1}
    int rc = dp_plain_bwd(a, a->q1, a->k1, tr(hm), yc, g, tq, tk, cur, L.lse_1, W.delta, st); if (rc) return rc;
    DP_ELEM(dp_add_kernel, a->dq1, tq); DP_ELEM(dp_add_kernel, a->dk1, tk);
    for (int i = hm; i >= 1; --i) {
        const MopkView4 nxt = i == 1 ? a->dv2 : dp_tmp_view((hm - i) % 2 == 0 ? W.tb : W.ta, a);
        rc = dp_plain_bwd(a, a->q2, a->k2, tr(i - 1), tr(i), cur, tq, tk, nxt, L.lse_2, W.delta, st); if (rc) return rc;
        DP_ELEM(dp_add_kernel, a->dq2, tq); DP_ELEM(dp_add_kernel, a->dk2, tk);
        cur = nxt;
    }
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

}  // namespace mopk
