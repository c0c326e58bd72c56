// EdgewiseMSA low-rank core -- fused gfx950 forward kernel, 16-query waves (bf16 MFMA, fp32 accumulate).
//
// Same algorithm, on-chip data flow and `saved` record as edgewise_fused.hip (reference attention_variants.py:495-560), but
// every wave owns 16 queries instead of 32 and works on v_mfma_f32_16x16x32_bf16 tiles:
//   * one workgroup per (batch, head), NW = ceil(N/16) waves (13 at N = 197): 3-4 waves per SIMD instead of 2, and the
//     per-wave chain state is 28 VGPRs (packed bf16) instead of 56, so the kernel fits the 128-register budget of that
//     occupancy without scratch;
//   * "X layout": lane (n, g) = (lane & 15, lane >> 4) holds, for query 16w + n, the keys 16t + 4g + i (i = 0..3) of tile t
//     -- the accumulator layout of a 16x16 MFMA whose N index is the query.  Two consecutive tiles packed to bf16 are the
//     B operand of the next chain step (k slot 8g + j <-> key 32s + 16(j>>2) + 4g + (j&3)); the LDS image of A_m^T stores
//     its columns in that slot order (kperm32), so one ds_read_b128 per MFMA feeds the A operand.
// The training exports (prefix products, C->/C<- slabs) are written in exactly the order the 32-query backward kernel reads
// them: the transposing MFMA (X . selector) lands each 16-query wave's values in 8-byte halves of that order's 16-byte
// chunks, and a wave's two stores per step cover whole 256-byte segments.
#include "fused_common.h"

namespace mopk {

__device__ __forceinline__ int kperm32(int k) { return ((k >> 2) & 3) * 8 + (k >> 4) * 4 + (k & 3); }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// register-array element with a wave-uniform runtime index (scalar branches; keeps rolled loops out of scratch)
template <typename T, int K> __device__ __forceinline__ T vsel_get(const T (&X)[K], int s) {
    T r = X[0];
    switch (s) {
#define MOPK_VG(I_) case I_: if (I_ < K) r = X[I_ < K ? I_ : 0]; break;
        MOPK_VG(1) MOPK_VG(2) MOPK_VG(3) MOPK_VG(4) MOPK_VG(5) MOPK_VG(6) MOPK_VG(7)
#undef MOPK_VG
        default: break;
    }
    return r;
}
template <typename T, int K> __device__ __forceinline__ void vsel_set(T (&X)[K], int s, const T &v) {
    switch (s) {
#define MOPK_VS(I_) case I_: if (I_ < K) X[I_ < K ? I_ : 0] = v; break;
        MOPK_VS(0) MOPK_VS(1) MOPK_VS(2) MOPK_VS(3) MOPK_VS(4) MOPK_VS(5) MOPK_VS(6) MOPK_VS(7)
#undef MOPK_VS
        default: break;
    }
}

template <int KS32, int DK>
struct Cfg16 {
    // LDS rows are read as MFMA A operands: lane (n, g) takes 16 bytes of row n at chunk 4s + g.  ds_read_b128 is served in
    // four 16-lane groups {n in 0-3,12-15 of g | n in 4-11 of g+1} (MI355X_MICROARCH.md, LDS table), which is conflict-free
    // when the row stride is 2 x odd 16-byte slots: 480 B for the N x N image (NP + 16 columns), 96 B for K at dk = 32.
    // At dk = 64 the K rows stay unpadded (128 B) and chunk c of row j is stored at c ^ ((j >> 1) & 7) instead.
    static constexpr int NP = 32 * KS32, NT16 = 2 * KS32, LDA = NP + 16, LDK = DK == 64 ? 64 : DK + 16, KQ = DK / 32, DT16 = DK / 16;
    static constexpr bool KSWZ = DK == 64;
    static constexpr int BTS16 = 16;      // gate b-vectors: [b_hi | b_hi | b_lo | b_lo], 32-byte rows (2 slots: conflict-free)
    static constexpr int R_BYTES = imax(NP * LDA * 2, 2 * DK * LDA * 2 + 4 * NP * BTS16 * 2);
    static constexpr int K_BYTES = NP * LDK * 2;
    static constexpr int WSM_FLOATS = 2 * 16 * 19;
    // fp32 scratch: sqk sqk2 [8][DK] | qbar kbar vs0 vsL [DK] | rCr rCl cCr cCl [NP] | Z | wsig[8]
    // Z during the chains: colpart [NT16][NP]; afterwards rS cS [V][NP] + gate-head weights
    static __host__ __device__ constexpr int z_floats(int V) { return imax(NT16 * NP, 2 * V * NP + WSM_FLOATS); }
    static __host__ __device__ constexpr int small_floats(int V) { return 16 * DK + 4 * DK + 4 * NP + z_floats(V) + 8; }
    static __host__ __device__ constexpr int lds_bytes(int V) { return R_BYTES + K_BYTES + 4 * small_floats(V); }
};

template <int KS32, int DK, typename IOT, bool SAVE>
__global__ void __launch_bounds__(KS32 * 128) ew16_fwd_kernel(MopkEdgewiseArgs a) {
    using Cfg = Cfg16<KS32, DK>;
    constexpr int NP = Cfg::NP, NT16 = Cfg::NT16, LDA = Cfg::LDA, LDK = Cfg::LDK, KQ = Cfg::KQ, DT16 = Cfg::DT16, BTS16 = Cfg::BTS16;
    constexpr int PLD = NP + 8;           // row length of the exported prefix-product images (layout of the 32-query kernels)
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short *AT = (unsigned short *)smem;                   // [NP][LDA]   A_m^T, kperm32 columns
    unsigned short *VT0 = AT;                                      // [DK][LDA]   (aliases AT after the chains)
    unsigned short *VTL = AT + DK * LDA;                           // [DK][LDA]
    unsigned short *bT = AT + 2 * DK * LDA;                        // [4][NP][BTS16]
    unsigned short *Ksm = (unsigned short *)(smem + Cfg::R_BYTES); // [NP][LDK]
    float *fs = (float *)(smem + Cfg::R_BYTES + Cfg::K_BYTES);
    float *sqk = fs, *sqk2 = sqk + 8 * DK, *qbar = sqk2 + 8 * DK, *kbar = qbar + DK, *vs0 = kbar + DK, *vsL = vs0 + DK;
    float *rCr = vsL + DK, *rCl = rCr + NP, *cCr = rCl + NP, *cCl = cCr + NP;
    float *Z = cCl + NP;
    float *colpart = Z;                                            // chains
    float *rS = Z, *cS = Z + a.V * NP, *Wsm = Z + 2 * a.V * NP;    // gate phase
    float *wsig = Z + Cfg::z_floats(a.V);

    const int tid = threadIdx.x, w = tid >> 6, NTHR = blockDim.x, NW = NTHR >> 6;
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int N = a.N, V = a.V, H = a.H, R = a.r;
    const int b = blockIdx.x / H, hh = blockIdx.x % H;
    const int qi = 16 * w + n;
    const bool qok = qi < N;
    const float invN = 1.f / (float)N;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

#ifdef MOPK_STAMPS
    unsigned long long *stamps = (unsigned long long *)a.workspace;      // 256-byte forward workspace: stamps of workgroup 0 (tools/stamps_fwd.py)
    int stamp_i = 0;
#define FSTAMP() do { if (blockIdx.x == 0 && tid == 0 && stamp_i < 32) stamps[stamp_i] = __builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#else
#define FSTAMP() do { } while (0)
#endif
#ifdef MOPK_STAMPS2
#define FSTAMP2(c_) do { if (c_) FSTAMP(); } while (0)
#else
#define FSTAMP2(c_) do { } while (0)
#endif
    FSTAMP();
    const IOT *qbase = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh;
    const IOT *qrow = qbase + (int64_t)qi * a.q.sn;
    // ---------------- P0: stage K, scales, means ----------------
    {
        const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh;
        constexpr int CH = DK / 8;
        for (int c = tid; c < NP * CH; c += NTHR) {
            const int j = c / CH, dc = c % CH;
            bf16x8 v = zero8;
            if (j < N) v = load8_bf16<IOT>(kp + (int64_t)j * a.k.sn + dc * 8);
            *(bf16x8 *)&Ksm[j * LDK + (Cfg::KSWZ ? (dc ^ ((j >> 1) & 7)) : dc) * 8] = v;
        }
        for (int c = tid; c < V * DK; c += NTHR) { const float t = a.sqk[((c / DK) * H + hh) * DK + (c % DK)]; sqk[c] = t; sqk2[c] = t * 1.4426950408889634f; }
        for (int c = tid; c < DK; c += NTHR) { vs0[c] = a.vs0[hh * DK + c]; vsL[c] = a.vsL[hh * DK + c]; }
        if (tid == 0) wsig[0] = 1.f / (1.f + __expf(-*a.chain_logit));
        // image columns of the queries no wave owns (16 NW .. NP-1) stay zero through every chain step
        const int miss = NP - 16 * NW;
        for (int c = tid; c < NP * miss; c += NTHR) {
            const int row = c / miss, kq = 16 * NW + c % miss;
            AT[row * LDA + (kq & ~31) + kperm32(kq & 31)] = 0;
        }
        // qbar partials straight from global q (bf16-rounded like the fragments), [NPQ][DK] in Z
        const int d = tid % DK, p = tid / DK, NPQ = NTHR / DK;
        float s = 0.f;
        for (int i = p; i < N; i += NPQ) s += bf2f(f2bf(ld_as_f32<IOT>(qbase + (int64_t)i * a.q.sn + d)));
        Z[p * DK + d] = s;
    }
    __syncthreads();                                    // Ksm staged, qbar partials written
    {
        const int d = tid % DK, p = tid / DK, NPQ = NTHR / DK;
        float s = 0.f;
        for (int j = p; j < N; j += NPQ) s += bf2f(Ksm[j * LDK + (Cfg::KSWZ ? (((d >> 3) ^ ((j >> 1) & 7)) * 8 + (d & 7)) : d)]);
        Z[(NPQ + p) * DK + d] = s;
    }
    __syncthreads();
    if (tid < DK) {
        const int NPQ = NTHR / DK;
        float sk = 0.f, sq = 0.f;
        for (int p = 0; p < NPQ; ++p) { sq += Z[p * DK + tid]; sk += Z[(NPQ + p) * DK + tid]; }
        kbar[tid] = sk * invN; qbar[tid] = sq * invN;
    }
    __syncthreads();

    const FusedSavedLayout SL = fused_saved_layout<KS32, DK>(N, V, SAVE);
    unsigned char *svb = (unsigned char *)a.saved + (size_t)blockIdx.x * SL.stride;    // this (b,h)'s record
    // ---------------- helpers ----------------
    auto load_q = [&](bf16x8 (&qf)[KQ]) {
#pragma unroll
        for (int s = 0; s < KQ; ++s) { bf16x8 v = zero8; if (qok) v = load8_bf16<IOT>(qrow + 32 * s + 8 * g); qf[s] = v; }
    };
    auto scale_q = [&](bf16x8 (&qe)[KQ], const bf16x8 (&qf)[KQ], const float *tab, int v) {
#pragma unroll
        for (int s = 0; s < KQ; ++s) {
            const float4 s0 = *(const float4 *)&tab[v * DK + 32 * s + 8 * g], s1 = *(const float4 *)&tab[v * DK + 32 * s + 8 * g + 4];
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) qe[s][j] = (short)f2bf(bf2f((unsigned short)qf[s][j]) * sc[j]);
        }
    };
    // this lane's K-row fragment offset within a 16-key tile (rows 16t + n keep the swizzle term: 16t is even-aligned)
    auto krow_off = [&](int s) -> int { return n * LDK + (Cfg::KSWZ ? ((4 * s + g) ^ ((n >> 1) & 7)) : (4 * s + g)) * 8; };
    auto s_tile = [&](const bf16x8 (&qe)[KQ], int t) -> f32x4 {      // S^T tile: keys 16t + 4g + i, query n
        f32x4 acc = zero4;
#pragma unroll
        for (int s = 0; s < KQ; ++s) {
            const bf16x8 af = *(const bf16x8 *)&Ksm[(16 * t) * LDK + krow_off(s)];
            acc = mfma16(af, qe[s], acc);
        }
        return acc;
    };
    constexpr float NEG = -1e30f;
    auto row_const = [&](const bf16x8 (&qe)[KQ]) -> float {        // c = log2 sum_j 2^(S'[i,j])      :500-507
        float m = NEG, l = 0.f;
#pragma unroll 2
        for (int t = 0; t < NT16; ++t) {
            f32x4 S = s_tile(qe, t);
#pragma unroll
            for (int i = 0; i < 4; ++i) S[i] = (16 * t + 4 * g + i >= N) ? NEG : S[i];
            const float mn = fmaxf(fmaxf(m, fmaxf(S[0], S[1])), fmaxf(S[2], S[3]));
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) sm += __builtin_amdgcn_exp2f(S[i] - mn);
            l = fmaf(l, __builtin_amdgcn_exp2f(m - mn), sm);
            m = mn;
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
            const float mx = fmaxf(m, m2);
            l = l * __builtin_amdgcn_exp2f(m - mx) + l2 * __builtin_amdgcn_exp2f(m2 - mx);
            m = mx;
        }
        return m + __builtin_amdgcn_logf(l);
    };
    auto a_tile = [&](const bf16x8 (&qe)[KQ], int t, float c) -> f32x4 {   // A_v^T tile (keys >= N -> 0)
        f32x4 S = s_tile(qe, t);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float e = __builtin_amdgcn_exp2f(S[i] - c); S[i] = (16 * t + 4 * g + i >= N) ? 0.f : e; }
        return S;
    };
    // two consecutive tiles -> one packed B fragment
    auto pack2 = [&](const f32x4 &x0, const f32x4 &x1) -> bf16x8 {
        bf16x8 p;
#pragma unroll
        for (int i = 0; i < 4; ++i) { p[i] = (short)f2bf(x0[i]); p[4 + i] = (short)f2bf(x1[i]); }
        return p;
    };
    unsigned short *at_col = AT + 32 * (w >> 1) + kperm32(16 * (w & 1) + n);     // this query's image column
    unsigned short *at_col_a = at_col + (4 * g + 2 * (g & 1)) * LDA, *at_col_b = at_col + (4 * g + 2 - 2 * (g & 1)) * LDA;
    auto store_AT_tile = [&](int t, const f32x4 &x) {
        // odd lane groups store their rows in the order 2,3,0,1: with 480-byte rows, groups g and g+1 would otherwise
        // hit the same banks (4 rows = 1920 B = 0 mod 128 B)
        const float y0 = (g & 1) ? x[2] : x[0], y1 = (g & 1) ? x[3] : x[1], y2 = (g & 1) ? x[0] : x[2], y3 = (g & 1) ? x[1] : x[3];
        at_col_a[16 * t * LDA] = f2bf(y0); at_col_a[(16 * t + 1) * LDA] = f2bf(y1);
        at_col_b[16 * t * LDA] = f2bf(y2); at_col_b[(16 * t + 1) * LDA] = f2bf(y3);
    };
    const unsigned short *at_row = AT + n * LDA + 8 * g;
    auto gemm_tile = [&](int to, const bf16x8 (&Xp)[KS32]) -> f32x4 {        // (A_m^T . X)[tile to]
        f32x4 acc = zero4;
#pragma unroll
        for (int s = 0; s < KS32; ++s) {
            const bf16x8 af = *(const bf16x8 *)&at_row[(16 * to) * LDA + 32 * s];
            acc = mfma16(af, Xp[s], acc);
        }
        return acc;
    };
    // v = log(C + eps) for one tile; row sum and per-wave column partials (butterfly over the 16 lanes of a group)
    auto log_tile = [&](f32x4 &X, int t, float &rs) {
        float c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = __logf(X[i] + EPSC);
            X[i] = v;
            rs += (16 * t + 4 * g + i < N) ? v : 0.f;
            c[i] = qok ? v : 0.f;
        }
        const bool up = (n >> 3) & 1, up2 = (n >> 2) & 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float keep = up ? c[k + 2] : c[k], send = up ? c[k] : c[k + 2];
            c[k] = keep + __shfl_xor(send, 8, 64);
        }
        {
            const float keep = up2 ? c[1] : c[0], send = up2 ? c[0] : c[1];
            c[0] = keep + __shfl_xor(send, 4, 64);
        }
        c[0] += __shfl_xor(c[0], 2, 64);
        c[0] += __shfl_xor(c[0], 1, 64);
        if ((n & 3) == 0) colpart[w * NP + 16 * t + 4 * g + 2 * (int)up + (int)up2] = c[0];
    };
    // training export of a prefix product in the backward's "row slab" order (see the header)
    auto export_prefix = [&](const bf16x8 (&Xp)[KS32], u2 *out) {
        bf16x8 id0, id1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool hit = 4 * g + (j & 3) == n;
            id0[j] = (short)((hit && j < 4) ? 0x3f80 : 0);
            id1[j] = (short)((hit && j >= 4) ? 0x3f80 : 0);
        }
        for (int wc = w; wc < NT16; wc += NW) {                  // chunks of missing waves are written as zeros
            const bool real = wc == w;
#pragma unroll
            for (int s = 0; s < KS32; ++s)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    f32x4 tr = mfma16(Xp[s], h2 ? id1 : id0, zero4);      // lane (n, g): T[query 16w + 4g + i][key 32s + 16 h2 + n]
                    u2 pk;
                    pk[0] = real ? pack_bf16(tr[0], tr[1]) : 0u;
                    pk[1] = real ? pack_bf16(tr[2], tr[3]) : 0u;
                    const int L = n + 16 * h2 + 32 * (g & 1);
                    __builtin_nontemporal_store(pk, &out[(((size_t)s * 2 * KS32 + wc) * 64 + L) * 2 + (g >> 1)]);
                }
        }
    };
    // C-> / C<- slab (accumulator order of the 32-query layout): tile t16 of this wave -> 8-byte halves
    auto export_slab_tile = [&](u2 *slab, int t16, const f32x4 &x) {
        for (int wc = w; wc < NT16; wc += NW) {
            u2 pk;
            pk[0] = wc == w ? pack_bf16(x[0], x[1]) : 0u;
            pk[1] = wc == w ? pack_bf16(x[2], x[3]) : 0u;
            const int L = 16 * (wc & 1) + n + 32 * (g & 1);
            __builtin_nontemporal_store(pk, &slab[((((size_t)(wc >> 1) * KS32 * 2) + t16) * 64 + L) * 2 + (g >> 1)]);
        }
    };

    // same for a tile kept as packed fp16 (the backward's Smix / L slabs)
    auto export_slab_tile_h = [&](u2 *slab, int t16, const f32x4 &x) {
        for (int wc = w; wc < NT16; wc += NW) {
            u2 pk;
            pk[0] = wc == w ? pack_h2(x[0], x[1]) : 0u;
            pk[1] = wc == w ? pack_h2(x[2], x[3]) : 0u;
            const int L = 16 * (wc & 1) + n + 32 * (g & 1);
            __builtin_nontemporal_store(pk, &slab[((((size_t)(wc >> 1) * KS32 * 2) + t16) * 64 + L) * 2 + (g >> 1)]);
        }
    };
    // chain product (transposed, row-block local):  X <- A_{o[V-1]}^T .. A_{o[1]}^T A_{o[0]}^T[:, I]
    float cstr[8];                        // softmax constants c_v of this lane's query (computed by the <- chain, reused by ->)
#pragma unroll
    for (int v = 0; v < 8; ++v) cstr[v] = 0.f;
    auto run_chain = [&](bool forward, auto &&epi) {
        bf16x8 Xp[KS32];
        auto view_const = [&](const bf16x8 (&qe)[KQ], int v) -> float {
            if (forward) return vsel_get(cstr, v);
            const float c = row_const(qe);
            vsel_set(cstr, v, c);
            if (SAVE && g == 0) ((float *)(svb + SL.oCst))[v * NP + qi] = c;
            return c;
        };
        {
            const int v = forward ? 0 : V - 1;
            bf16x8 qe[KQ];
            load_q(qe);                   // q re-read from L2 per step (keeps the fragments out of the loop-carried state)
            scale_q(qe, qe, sqk2, v);
            const float c = view_const(qe, v);
#pragma nounroll
            for (int s = 0; s < KS32; ++s) { const f32x4 A0 = a_tile(qe, 2 * s, c), A1 = a_tile(qe, 2 * s + 1, c); vsel_set(Xp, s, pack2(A0, A1)); }
        }
        for (int m = 1; m < V; ++m) {
            if (SAVE) export_prefix(Xp, (u2 *)(svb + (forward ? SL.oT : SL.oU) + (size_t)(m - 1) * NP * PLD * 2));
            {
                const int v = forward ? m : V - 1 - m;
                bf16x8 qe[KQ];
                load_q(qe);
                scale_q(qe, qe, sqk2, v);
                const float c = view_const(qe, v);
                    __syncthreads();              // previous step's readers of AT are done
    #pragma unroll 2
                for (int t = 0; t < NT16; ++t) { const f32x4 A = a_tile(qe, t, c); store_AT_tile(t, A); }
                    __syncthreads();
                }
            if (m < V - 1) {
                bf16x8 Xn[KS32];
#pragma nounroll
                for (int s = 0; s < KS32; ++s) { const f32x4 c0 = gemm_tile(2 * s, Xp), c1 = gemm_tile(2 * s + 1, Xp); vsel_set(Xn, s, pack2(c0, c1)); }
#pragma unroll
                for (int s = 0; s < KS32; ++s) Xp[s] = Xn[s];
                } else {
#pragma nounroll
                for (int s = 0; s < KS32; ++s) { f32x4 c0 = gemm_tile(2 * s, Xp), c1 = gemm_tile(2 * s + 1, Xp); epi(s, c0, c1); }
            }
            FSTAMP2(true);
        }
    };

    FSTAMP();
    // ---------------- chain <- : only its log-means survive           :513-515, :521
    {
        float rs = 0.f;
        run_chain(false, [&](int s, f32x4 &c0, f32x4 &c1) {
            if (SAVE) { export_slab_tile((u2 *)(svb + SL.oCB), 2 * s, c0); export_slab_tile((u2 *)(svb + SL.oCB), 2 * s + 1, c1); }
            log_tile(c0, 2 * s, rs);
            log_tile(c1, 2 * s + 1, rs);
        });
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        if (g == 0) rCl[qi] = rs * invN;
    }
    __syncthreads();
    if (tid < NP) { float c = 0.f; for (int ww = 0; ww < NW; ++ww) c += colpart[ww * NP + tid]; cCl[tid] = c * invN; }
    for (int c = tid + 16 * NW; c < NP; c += NTHR) rCl[c] = 0.f;       // rows of the queries no wave owns
    __syncthreads();                          // colpart is rewritten by the -> chain's epilogue
    FSTAMP();
    // ---------------- chain -> : C-> kept as packed bf16 (for y_chain) and log C-> as packed fp16 (for the mix)
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    u4 crp[KS32];                         // log C-> (later Smix) of tiles 2s, 2s+1 as packed fp16; filled after y_chain
    IOT *yp = (IOT *)a.y.ptr + b * a.y.sb + hh * a.y.sh + (int64_t)qi * a.y.sn;
    float *ych = (float *)(svb + SL.oYch);
    {
        bf16x8 Xc[KS32];
        float rs = 0.f;
        // log C-> from the bf16-rounded product (the same rounding the backward sees): means + packed fp16 copy
        run_chain(true, [&](int s, f32x4 &a0, f32x4 &a1) {
            const bf16x8 xc = pack2(a0, a1);
            vsel_set(Xc, s, xc);
            f32x4 c0, c1;
#pragma unroll
            for (int i = 0; i < 4; ++i) { c0[i] = bf2f((unsigned short)xc[i]); c1[i] = bf2f((unsigned short)xc[4 + i]); }
            if (SAVE) { export_slab_tile((u2 *)(svb + SL.oCF), 2 * s, c0); export_slab_tile((u2 *)(svb + SL.oCF), 2 * s + 1, c1); }
            log_tile(c0, 2 * s, rs);
            log_tile(c1, 2 * s + 1, rs);
        });
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        if (g == 0) rCr[qi] = rs * invN;
        for (int c = tid + 16 * NW; c < NP; c += NTHR) rCr[c] = 0.f;
        FSTAMP2(true);
        __syncthreads();                      // AT free: build V0^T, VL^T (kperm32 key columns); colpart complete
        FSTAMP2(true);
        {
            const IOT *v0p = (const IOT *)a.v0.ptr + b * a.v0.sb + hh * a.v0.sh;
            const IOT *vLp = (const IOT *)a.vL.ptr + b * a.vL.sb + hh * a.vL.sh;
            constexpr int CH = DK / 8;
            for (int c = tid; c < NP * CH; c += NTHR) {
                const int j = c % NP, dc = c / NP;       // consecutive lanes = consecutive keys: their 2-byte stores spread over the banks
                bf16x8 x0 = zero8, xL = zero8;
                if (j < N) { x0 = load8_bf16<IOT>(v0p + (int64_t)j * a.v0.sn + dc * 8); xL = load8_bf16<IOT>(vLp + (int64_t)j * a.vL.sn + dc * 8); }
                const int col = (j & ~31) + kperm32(j & 31);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int d = dc * 8 + e;
                    VT0[d * LDA + col] = f2bf(bf2f((unsigned short)x0[e]) * vs0[d]);
                    VTL[d * LDA + col] = f2bf(bf2f((unsigned short)xL[e]) * vsL[d]);
                }
            }
            if (tid < NP) { float c = 0.f; for (int ww = 0; ww < NW; ++ww) c += colpart[ww * NP + tid]; cCr[tid] = c * invN; }
            __syncthreads();              // colpart fully consumed before rS / cS overwrite it; cCr complete
            if (SAVE) {                   // the log-means; softmax constants of the queries no wave owns
                float *gc = (float *)(svb + SL.oCst), *gm = (float *)(svb + SL.oMeans);
                if (tid < NP) { gm[tid] = rCr[tid]; gm[NP + tid] = rCl[tid]; gm[2 * NP + tid] = cCr[tid]; gm[3 * NP + tid] = cCl[tid]; }
                for (int c = tid; c < V * (NP - 16 * NW); c += NTHR) gc[(c / (NP - 16 * NW)) * NP + 16 * NW + c % (NP - 16 * NW)] = 0.f;
            }
            // row / col means of S_v are linear in q, k:  rS_v[i] = Qe_v[i,:].kbar ; cS_v[j] = k[j,:].(sqk_v*qbar)
            {
                bf16x8 qf[KQ];
                load_q(qf);
                for (int v = 0; v < V; ++v) {
                    float p = 0.f;
#pragma unroll
                    for (int s = 0; s < KQ; ++s)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int d = 32 * s + 8 * g + j;
                            p = fmaf(bf2f((unsigned short)qf[s][j]) * sqk[v * DK + d], kbar[d], p);
                        }
                    p += __shfl_xor(p, 16, 64);
                    p += __shfl_xor(p, 32, 64);
                    if (g == 0) rS[v * NP + qi] = p;
                }
                // cS on the matrix core: B column n < 8 carries the bf16 "hi" part of u_v = sqk_v * qbar (v = n), columns 8-15 the remainder
                const int vv = n & 7;
                const bool act = vv < V, lo_part = n >= 8;
                f32x4 acc = zero4;
#pragma unroll
                for (int s = 0; s < KQ; ++s) {
                    bf16x8 bf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int d = 32 * s + 8 * g + e;
                        const float u = act ? sqk[vv * DK + d] * qbar[d] : 0.f;
                        const unsigned short hi = f2bf(u);
                        bf[e] = (short)(lo_part ? f2bf(u - bf2f(hi)) : hi);
                    }
                    const bf16x8 af = *(const bf16x8 *)&Ksm[(16 * w) * LDK + krow_off(s)];
                    acc = mfma16(af, bf, acc);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float tot = acc[i] + __shfl_xor(acc[i], 8, 64);
                    if (n < 8 && n < V) cS[n * NP + 16 * w + 4 * g + i] = tot;
                }
                for (int c = tid; c < V * (NP - 16 * NW); c += NTHR) {         // keys / queries no wave owns
                    const int v = c / (NP - 16 * NW), j = 16 * NW + c % (NP - 16 * NW);
                    cS[v * NP + j] = 0.f; rS[v * NP + j] = 0.f;
                }
            }
        }
        __syncthreads();                      // VT*, cCr, rS, cS complete
        FSTAMP2(true);
        const float wv = wsig[0];
#pragma unroll
        for (int dt = 0; dt < DT16; ++dt) {   // y_chain^T = VL^T C->^T       :556-560 (as C-> vL); parked in `saved`
            f32x4 acc = zero4;
#pragma unroll
            for (int s = 0; s < KS32; ++s) {
                const bf16x8 af = *(const bf16x8 *)&VTL[(16 * dt + n) * LDA + 32 * s + 8 * g];
                acc = mfma16(af, Xc[s], acc);
            }
            if (qok) *(float4 *)&ych[(size_t)qi * DK + 16 * dt + 4 * g] = make_float4(wv * acc[0], wv * acc[1], wv * acc[2], wv * acc[3]);
        }
        FSTAMP2(true);
#pragma unroll
        for (int s = 0; s < KS32; ++s) {      // log C-> (bf16-rounded product) as packed fp16 for the mix
            float c[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __logf(bf2f((unsigned short)Xc[s][j]) + EPSC);
            crp[s][0] = pack_h2(c[0], c[1]); crp[s][1] = pack_h2(c[2], c[3]); crp[s][2] = pack_h2(c[4], c[5]); crp[s][3] = pack_h2(c[6], c[7]);
        }
    }
    FSTAMP();
    // ---------------- gate vectors                                     :323-326
    const int C = 2 * V + 2;
    for (int c = tid; c < 2 * 4 * R * (C + 1); c += NTHR) {
        const int side = c / (4 * R * (C + 1)), rem = c % (4 * R * (C + 1)), o = rem / (C + 1), cc = rem % (C + 1);
        const float *Wg = side ? a.Wc : a.Wr, *bg = side ? a.bc : a.br;
        Wsm[(side * 16 + o) * 19 + (cc < C ? cc : 18)] = cc < C ? Wg[o * C + cc] : bg[o];
    }
    __syncthreads();
    for (int p = tid; p < 4 * NP; p += NTHR) {          // b[g,k,j] -> bT[g][j][slots] = [b_hi | b_hi | b_lo | b_lo]
        const int j = p % NP, gg = p / NP;
        unsigned short hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
        if (j < N)
            for (int k = 0; k < R; ++k) {
                const float *Wo = Wsm + (16 + gg * R + k) * 19;
                float s = Wo[18];
                for (int c = 0; c < V; ++c) s = fmaf(Wo[c], cS[c * NP + j], fmaf(Wo[V + c], rS[c * NP + j], s));
                s = fmaf(Wo[2 * V], cCr[j], fmaf(Wo[2 * V + 1], cCl[j], s));
                hi[k] = f2bf(s); lo[k] = f2bf(s - bf2f(hi[k]));
            }
        unsigned short *row = bT + (gg * NP + j) * BTS16;
#pragma unroll
        for (int k = 0; k < 4; ++k) { row[k] = hi[k]; row[4 + k] = hi[k]; row[8 + k] = lo[k]; row[12 + k] = lo[k]; }
    }
    bf16x8 af4[4];                        // a[g,k,i] as B fragments: slots [a_hi | a_lo] in lane groups 0 and 1, zero above
    for (int gg = 0; gg < 4; ++gg) {
        float av[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < R; ++k) {
            const float *Wo = Wsm + (gg * R + k) * 19;
            float s = Wo[18];
            for (int c = 0; c < V; ++c) s = fmaf(Wo[c], rS[c * NP + qi], fmaf(Wo[V + c], cS[c * NP + qi], s));
            s = fmaf(Wo[2 * V], rCr[qi], fmaf(Wo[2 * V + 1], rCl[qi], s));
            av[k] = s;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a2 = av[k] * 1.4426950408889634f;               // gate logits pre-scaled by log2(e): sigmoid = 1/(1+2^-z')
            const unsigned short hi = f2bf(a2), lo = f2bf(a2 - bf2f(hi));
            af4[gg][k] = g < 2 ? (short)hi : (short)0;
            af4[gg][4 + k] = g < 2 ? (short)lo : (short)0;
        }
    }
    __syncthreads();                      // bT complete
    FSTAMP();
    // ---------------- score-space mix, tile by tile                    :537-547
    const float nb = a.beta_not / (float)(V > 1 ? V - 1 : 1);
    float mxrow = -INFINITY;
    const unsigned short *bt_row = bT + n * BTS16 + 8 * (g & 1);
    auto gate_tile = [&](int t, int g4) -> f32x4 {        // sigmoid(a_g^T b_g) for one 16x16 tile
        const bf16x8 bfrag = *(const bf16x8 *)&bt_row[(g4 * NP + 16 * t) * BTS16];
        f32x4 z = mfma16(bfrag, af4[g4], zero4);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-z[i]));
        return z;
    };
    {
        bf16x8 qraw[KQ];
        load_q(qraw);
#pragma nounroll
        for (int s = 0; s < KS32; ++s) {          // two key tiles per iteration share the per-view Qe scaling
            u4 cw = vsel_get(crp, s);
            f32x4 S0[2], O[2], L[2];
            {
                bf16x8 qe[KQ];
                scale_q(qe, qraw, sqk, 0);
                f32x4 mx[2], se[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) { S0[h2] = s_tile(qe, 2 * s + h2); O[h2] = zero4; mx[h2] = zero4; se[h2] = f32x4{1.f, 1.f, 1.f, 1.f}; }
                for (int v = 1; v < V; ++v) {     // online logsumexp over views, relative to S_0
                    scale_q(qe, qraw, sqk, v);
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const f32x4 Sv = s_tile(qe, 2 * s + h2);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            O[h2][i] += Sv[i];
                            const float d = Sv[i] - S0[h2][i];
                            const float e = __expf(-fabsf(d - mx[h2][i]));
                            se[h2][i] = d > mx[h2][i] ? fmaf(se[h2][i], e, 1.f) : se[h2][i] + e;
                            mx[h2][i] = fmaxf(mx[h2][i], d);
                        }
                    }
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int i = 0; i < 4; ++i) L[h2][i] = mx[h2][i] + __logf(se[h2][i]);   // lse_v S_v - S_0
                if (SAVE) {
                    export_slab_tile_h((u2 *)(svb + SL.oL), 2 * s, L[0] * 1.4426950408889634f);
                    export_slab_tile_h((u2 *)(svb + SL.oL), 2 * s + 1, L[1] * 1.4426950408889634f);
                }
            }
            // Smix = S0 + (G_and - nb G_not) O + G_or L + G_chain Cr
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int t = 2 * s + h2;
                const float cr[4] = {h2_lo(cw[2 * h2]), h2_hi(cw[2 * h2]), h2_lo(cw[2 * h2 + 1]), h2_hi(cw[2 * h2 + 1])};
                const f32x4 G0 = gate_tile(t, 0), G1 = gate_tile(t, 1), G2 = gate_tile(t, 2), G3 = gate_tile(t, 3);
                f32x4 sm;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = fmaf(G0[i] - nb * G2[i], O[h2][i], S0[h2][i]);
                    x = fmaf(G1[i], L[h2][i], x);
                    x = fmaf(G3[i], cr[i], x);
                    x = (16 * t + 4 * g + i >= N) ? -INFINITY : x;
                    mxrow = fmaxf(mxrow, x);
                    sm[i] = x;
                }
                cw[2 * h2] = pack_h2(sm[0], sm[1]);
                cw[2 * h2 + 1] = pack_h2(sm[2], sm[3]);
                if (SAVE) export_slab_tile_h((u2 *)(svb + SL.oSm), t, sm);
            }
            vsel_set(crp, s, cw);
        }
    }
    FSTAMP();
    // ---------------- softmax over keys + P V0                         :551-554
    mxrow = fmaxf(mxrow, __shfl_xor(mxrow, 16, 64));
    mxrow = fmaxf(mxrow, __shfl_xor(mxrow, 32, 64));
    float l = 0.f;
    bf16x8 Pp[KS32];
#pragma unroll
    for (int s = 0; s < KS32; ++s)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const float e0 = __expf(h2_lo(crp[s][2 * h2]) - mxrow), e1 = __expf(h2_hi(crp[s][2 * h2]) - mxrow);
            const float e2 = __expf(h2_lo(crp[s][2 * h2 + 1]) - mxrow), e3 = __expf(h2_hi(crp[s][2 * h2 + 1]) - mxrow);
            l += (e0 + e1) + (e2 + e3);
            Pp[s][4 * h2] = (short)f2bf(e0); Pp[s][4 * h2 + 1] = (short)f2bf(e1);
            Pp[s][4 * h2 + 2] = (short)f2bf(e2); Pp[s][4 * h2 + 3] = (short)f2bf(e3);
        }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float invl = 1.f / l;
    if (SAVE) {
        float *rw = (float *)(svb + SL.oRow);
        if (g == 0) { rw[qi] = mxrow; rw[NP + qi] = invl; }
        for (int c = tid + 16 * NW; c < NP; c += NTHR) { rw[c] = 0.f; rw[NP + c] = 1.f; }      // queries no wave owns
    }
#pragma unroll
    for (int dt = 0; dt < DT16; ++dt) {
        f32x4 acc = zero4;
#pragma unroll
        for (int s = 0; s < KS32; ++s) {
            const bf16x8 af = *(const bf16x8 *)&VT0[(16 * dt + n) * LDA + 32 * s + 8 * g];
            acc = mfma16(af, Pp[s], acc);
        }
        if (qok) {
            const int d0 = 16 * dt + 4 * g;
            const float4 yc = *(const float4 *)&ych[(size_t)qi * DK + d0];      // w * y_chain (own earlier store)
            store4<IOT>(yp + d0, fmaf(acc[0], invl, yc.x), fmaf(acc[1], invl, yc.y), fmaf(acc[2], invl, yc.z), fmaf(acc[3], invl, yc.w));
            if (SAVE) *(float4 *)&((float *)(svb + SL.oYb))[(size_t)qi * DK + d0] = make_float4(acc[0] * invl, acc[1] * invl, acc[2] * invl, acc[3] * invl);
        }
    }
    FSTAMP();
}

// ------------------------------------------------------------------ host side
template <int KS32, int DK>
static int ew16_launch(const MopkEdgewiseArgs *a, hipStream_t st) {
    const int lds = Cfg16<KS32, DK>::lds_bytes(a->V);
    if (lds > 160 * 1024) return MOPK_ERR_UNSUPPORTED;
    const int NW = (a->N + 15) / 16;
    const dim3 grid(a->B * a->H), block(NW * 64);
#define MOPK_LAUNCH(IOT_, SAVE_) do {                                                                             \
        auto kfn = ew16_fwd_kernel<KS32, DK, IOT_, SAVE_>;                                                        \
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MOPK_ERR_LAUNCH; \
        hipLaunchKernelGGL(kfn, grid, block, lds, st, *a);                                                        \
    } while (0)
    if (a->io_dtype == MOPK_BF16) { if (a->save_for_backward) MOPK_LAUNCH(unsigned short, true); else MOPK_LAUNCH(unsigned short, false); }
    else { if (a->save_for_backward) MOPK_LAUNCH(float, true); else MOPK_LAUNCH(float, false); }
#undef MOPK_LAUNCH
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}

// 16-query-wave forward: N in (128, 224], dk in {32, 64}; the caller has already checked ew_fused_fwd_supported()
int ew16_fwd_supported(const MopkEdgewiseArgs *a) {
    if (a->N <= 128 || a->N > 224) return 0;
    if (a->dk != 32 && a->dk != 64) return 0;
    const int lds = a->dk == 64 ? Cfg16<7, 64>::lds_bytes(a->V) : Cfg16<7, 32>::lds_bytes(a->V);
    return lds <= 160 * 1024;
}
int ew16_fwd(const MopkEdgewiseArgs *a, hipStream_t st) {
    return a->dk == 64 ? ew16_launch<7, 64>(a, st) : ew16_launch<7, 32>(a, st);
}

}  // namespace mopk
