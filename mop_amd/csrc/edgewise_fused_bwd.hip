// EdgewiseMSA low-rank core -- fused gfx950 BACKWARD kernels (bf16 MFMA, fp32 accumulate).
//
// One workgroup per (batch, head), NT = ceil(N/32) waves, wave w owns queries I = [32w, 32w+32).
// The backward is THREE launches of one kernel template (template parameter PH), each with its own register allocation
// (as one kernel the phases' live ranges overlapped: 83-213 spilled VGPRs, dq / dk accumulators and the chain state
// parked in memory -- 16.6 GB of traffic per launch at B = 256, 61 x the algorithmic I/O):
//   PH_A  mix backward: dSmix, gate-head gradients, mean gradients, dv; exports the direct score gradients per view,
//         G_chain * dSmix and the mean-gradient vectors into the per-(b,h) hand-off region;
//   PH_B  the two D-chains (D_{m-1}^T = A_m D_m^T, row-block local), one work item per (b,h, chain); every D_m is
//         exported as a packed per-wave slab;
//   PH_C  per view: dA_v = T_{v-1}^T D_v + U^T D' (two GEMMs), softmax backward + direct + mean terms -> dS_v,
//         dQe_v, dK_v; dq / dk accumulate over the views IN REGISTERS.
// It needs the chain / mix state the training forward exports (`saved`, edgewise_fused.hip); when the caller asked for the
// small `saved` (save_for_backward = 0) the host re-runs the forward in export mode into the workspace first.
// The only N x N data that touches memory are packed bf16 / fp16 per-wave register slabs (one coalesced 1 KiB store per
// fragment) and the forward's prefix products T_m, U_m in the backward's load order.
// Gradient flow (oracle/edgewise.py::core_bwd): dy -> dP, dSmix -> gate grads (da via MFMA on the
// register tile, db via a 32x32 LDS transpose + MFMA) -> mean grads -> dC->, dC<- -> the two D-chains
// (D_{m-1}^T = A_m D_m^T, row-block local) -> per view: dA_v (two GEMMs) -> softmax backward + direct +
// mean terms -> dS_v -> dQe_v (K^T dS^T) and dK (dS^T Q through LDS) -> dq, dk, dsqk; dv from P^T dy, C->^T dy.
#include "bwd_common.h"

namespace mopk {

// HEAD: 0 = low-rank gate head, 1 = dense gate head without the 3x3 convolution (launch A: per-edge MLP backward; launch B: log C<-
//       gradient slab in the <- chain's seed; launch C: the S_v^T feature gradients added transposed)
template <int NT, int DK, typename IOT, int PH, int HEAD = 0>
__global__ void __launch_bounds__(NT * 64, NT <= 3 ? 2 : 1) ew_fused_bwd_kernel(MopkEdgewiseArgs a, BwdWs W, FusedDenseW dw) {
    using Cfg = BwdCfg<NT, DK>;
    constexpr int NP = Cfg::NP, LDA = Cfg::LDA, LDK = DK + 8, KS = DK / 16, DT = Cfg::DT, DP = Cfg::DP;
    constexpr int NTH = NT * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short *R = (unsigned short *)smem;                       // [NP][LDA] matrix region
    unsigned short *bT = R;                                           // [4][NP][BTS]           (mix phases)
    unsigned short *bmat = bT + 4 * NP * BTS;                         // [32][LDA] rows rho: b_hi (0-15) / b_lo (16-31), k-permuted cols
    unsigned short *amat = bmat + 32 * LDA;                           // [32][LDA] rows rho: a_hi / a_lo, k-permuted cols
    unsigned short *tbuf = amat + 32 * LDA;                           // [NT][32][40] per-wave 32x32 transpose buffer
    float *Wsm = (float *)(smem + Cfg::GATE_BYTES);                    // [2][16][WST] gate-head weights (+bias at [WST-1]) staged per (b,h) for P3..P7
    float *dav = (float *)R;                                          // [16][NP]  (after the mix-backward loop)
    float *dbv = dav + 16 * NP;                                       // [16][NP]
    unsigned short *Ksm = (unsigned short *)(smem + Cfg::R_BYTES);    // [NP][LDK]
    float *fs = (float *)(smem + Cfg::R_BYTES + Cfg::K_BYTES);
    float *sqk = fs, *sqk2 = sqk + 8 * DK, *qbar = sqk2 + 8 * DK, *kbar = qbar + DK, *vs0 = kbar + DK, *vsL = vs0 + DK;   // sqk2 = sqk * log2(e)
    float *rCr = vsL + DK, *rCl = rCr + NP, *cCr = rCl + NP, *cCl = cCr + NP;
    float *colpart = cCl + NP;                                        // [NT][NP]
    float *rS = colpart + NT * NP, *cS = rS + a.V * NP;               // [V][NP]
    float *dmean = colpart;                                           // [(2V+4)][NP] aliases colpart+rS+cS after the gate phase
    float *redbuf = colpart + imax(NT * NP + 2 * a.V * NP, (2 * a.V + 4) * NP);   // [2][NT][DK]
    float *misc = redbuf + 2 * NT * DK;                               // wsig, ...

    const int tid = threadIdx.x, w = tid >> 6;
    int lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.N, V = a.V, H = a.H, RK = a.r;
    int qi = 32 * w + r;
    bool qok = qi < N;
    // REFRESH(): make the lane id opaque so address arithmetic derived from it is recomputed per phase instead of
    // being hoisted to the kernel prologue and spilled (hipcc LICM + rematerialisation failure: ~1000 spills)
#define REFRESH() do { asm volatile("" : "+v"(lane)); r = lane & 31; h = lane >> 5; qi = 32 * w + r; qok = qi < N; } while (0)
    const float invN = 1.f / (float)N;
    const int E = (HEAD == 0 && PH == PH_A) ? dw.E : 0;     // extra feature channels of the low-rank head (row / column means from the caller)
    const int C = 2 * V + 2 + E;

    unsigned char *ws = W.base + (size_t)blockIdx.x * W.stride;
    const FusedSavedLayout SL = fused_saved_layout<NT, DK>(a.N, a.V, true);
    const unsigned char *svb = (const unsigned char *)a.saved;          // forward's record of the current (b,h) (set per iteration):
    const unsigned short *Tg = nullptr, *Ug = nullptr;                  //   prefix products, final products, softmax constants, log-means, mix state
    unsigned char *xf = W.xbase;                                        // hand-off region of the current (b,h)
    unsigned short *KT = (unsigned short *)(ws + W.oKT), *QT = (unsigned short *)(ws + W.oQT), *DYT = (unsigned short *)(ws + W.oDYT);
    unsigned short *V0s = (unsigned short *)(ws + W.oV0s), *VLs = (unsigned short *)(ws + W.oVLs);
    const float *cstats = nullptr;                                    // [V][NP] softmax constants c_v[i] (log2 of the row sum of 2^S'), from the forward
    float *dbp = (float *)(ws + W.oDbp);                              // [NT][16][NP]
    float *dwp = (float *)(ws + W.oDW);
    auto slot = [&](int s) -> u32x4 * {
        if (s < X_C3)                                                        // read-only: the forward's copies
            return (u32x4 *)(svb + (s == S_CF ? SL.oCF : s == S_CB ? SL.oCB : s == S_SM ? SL.oSm : SL.oL) + (size_t)w * Cfg::SLOT) + lane;
#ifdef MOPK_DEBUG_SLOTS      // diagnostic builds (tools/build_variant.py ... "-DMOPK_DEBUG_SLOTS"): a slab id outside this launch's hand-off region traps here
        if (s < 0 || s - X_C3 >= X_COUNT(V, HEAD == 1)) __builtin_trap();
#endif
        const size_t idx = (size_t)(unsigned)(s - X_C3) * (size_t)NT + (size_t)w;      // every slab offset in size_t
        return (u32x4 *)(xf + W.xSlots + idx * Cfg::SLOT) + lane;
    };
    // slot layout: [(t*2+s)][lane] u32x4  -> one coalesced 1 KiB store per (t,s)

#ifdef MOPK_STAMPS
    unsigned long long *stamps = (unsigned long long *)(ws + W.oStamp);
    int stamp_i = 0;
#ifndef MOPK_STAMP_PH
#define MOPK_STAMP_PH 2
#endif
#define STAMP() do { if (PH == MOPK_STAMP_PH && blockIdx.x == 0 && tid == 0 && stamp_i < 60) stamps[stamp_i] = __builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#else
#define STAMP() do { } while (0)
#endif
#ifdef MOPK_STAMPS2
#define STAMP2() STAMP()
#else
#define STAMP2() do { } while (0)
#endif
    // persistent workgroup: scratch is indexed by blockIdx, work items are strided.  PH_B has two items per (b,h): the two chains
    constexpr int ITEMS = PH == PH_B ? 2 : 1;
    for (int item = blockIdx.x; item < a.B * H * ITEMS; item += gridDim.x) {
    const int bh = item / ITEMS, chain_id = item % ITEMS;
    const int b = bh / H, hh = bh % H;
    const bool first_pass = item == (int)blockIdx.x;
    svb = (const unsigned char *)a.saved + (size_t)bh * SL.stride;
    xf = W.xbase + (size_t)bh * W.xstride;
    float *xdmean = (float *)(xf + W.xDmean);                          // [(2V+4)][NP] mean gradients (PH_A -> PH_B, PH_C)
    const float *ych = (const float *)(svb + SL.oYch);                 // w * y_chain from the fused forward
    Tg = (const unsigned short *)(svb + SL.oT); Ug = (const unsigned short *)(svb + SL.oU);
    cstats = (const float *)(svb + SL.oCst);
    STAMP();
    REFRESH();                           // per (b,h): nothing lane-derived may be hoisted out of the persistent loop (it would be spilled)
    const IOT *qrow = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn;
    const IOT *dyrow = (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh + (int64_t)qi * a.dy.sn;

    // ================= P0: stage operands (each launch stages only what it reads) =================
    //   PH_A: K (LDS), dy^T image, V0 rows (scaled), q^T in LDS for the query mean; row / col means of the scores
    //   PH_B: K (LDS), VL rows (scaled)          PH_C: K (LDS), k^T and q^T images
    constexpr bool NEED_KT = PH == PH_C, NEED_QT = PH == PH_C, NEED_QT_LDS = PH != PH_B, NEED_DYT = PH == PH_A;
    constexpr bool NEED_V0 = PH == PH_A, NEED_VL = PH == PH_B;
    {
        const int tl = 64 * w + lane;      // == tid, but derived from the refreshed lane id
        const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh;
        const IOT *qp = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh;
        const IOT *dp = (const IOT *)a.dy.ptr + b * a.dy.sb + hh * a.dy.sh;
        const IOT *v0p = (const IOT *)a.v0.ptr + b * a.v0.sb + hh * a.v0.sh;
        const IOT *vLp = (const IOT *)a.vL.ptr + b * a.vL.sb + hh * a.vL.sh;
        for (int c = tl; c < V * DK; c += NTH) { const float t = a.sqk[((c / DK) * H + hh) * DK + (c % DK)]; sqk[c] = t; sqk2[c] = t * 1.4426950408889634f; }
        for (int c = tl; c < DK; c += NTH) { vs0[c] = a.vs0[hh * DK + c]; vsL[c] = a.vsL[hh * DK + c]; }
        if (tid == 0) misc[0] = 1.f / (1.f + __expf(-*a.chain_logit));
        if (PH != PH_A) {                  // mean gradients of this (b,h) from PH_A
            for (int c = tl; c < (2 * V + 4) * NP; c += NTH) dmean[c] = xdmean[c];
        }
        constexpr int CH = DK / 8, ITER = (NP * CH + NTH - 1) / NTH;
        // every thread requests all its chunks first (padded rows read row 0 and are zeroed afterwards): one exposed memory round trip
        // for the whole staging pass instead of one per chunk
        bf16x8 kvv[ITER], qvv[ITER], dvv[ITER], x0v[ITER], xLv[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int c = tl + it * NTH, j = c / CH, dc = c % CH, jc = (c < NP * CH && j < N) ? j : 0;
            kvv[it] = load8_bf16<IOT>(kp + (int64_t)jc * a.k.sn + dc * 8);
            qvv[it] = dvv[it] = x0v[it] = xLv[it] = kvv[it];
            if (NEED_QT_LDS) qvv[it] = load8_bf16<IOT>(qp + (int64_t)jc * a.q.sn + dc * 8);
            if (NEED_DYT) dvv[it] = load8_bf16<IOT>(dp + (int64_t)jc * a.dy.sn + dc * 8);
            if (NEED_V0) x0v[it] = load8_bf16<IOT>(v0p + (int64_t)jc * a.v0.sn + dc * 8);
            if (NEED_VL) xLv[it] = load8_bf16<IOT>(vLp + (int64_t)jc * a.vL.sn + dc * 8);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int c = tl + it * NTH, j = c / CH, dc = c % CH;
            if (c >= NP * CH) continue;
            const bool ok = j < N;
            bf16x8 kv, qv, dv, x0, xL;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kv[e] = ok ? kvv[it][e] : (short)0; qv[e] = ok ? qvv[it][e] : (short)0; dv[e] = ok ? dvv[it][e] : (short)0;
                x0[e] = ok ? x0v[it][e] : (short)0; xL[e] = ok ? xLv[it][e] : (short)0;
            }
            *(bf16x8 *)&Ksm[j * LDK + dc * 8] = kv;
            const int col = (j & ~15) + kperm16(j & 15);
            bf16x8 s0, sL;
            float sc0[8], scL[8];
            *(float4 *)&sc0[0] = *(const float4 *)&a.vs0[hh * DK + dc * 8]; *(float4 *)&sc0[4] = *(const float4 *)&a.vs0[hh * DK + dc * 8 + 4];
            *(float4 *)&scL[0] = *(const float4 *)&a.vsL[hh * DK + dc * 8]; *(float4 *)&scL[4] = *(const float4 *)&a.vsL[hh * DK + dc * 8 + 4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = dc * 8 + e;
                // the transposed images are assembled in LDS (R is free here) and exported with 16-byte stores
                if (NEED_KT) R[d * LDA + col] = (unsigned short)kv[e];
                if (NEED_QT_LDS) R[(DP + d) * LDA + col] = (unsigned short)qv[e];
                if (NEED_DYT) R[(2 * DP + d) * LDA + col] = (unsigned short)dv[e];
                s0[e] = (short)f2bf(bf2f((unsigned short)x0[e]) * sc0[e]);
                sL[e] = (short)f2bf(bf2f((unsigned short)xL[e]) * scL[e]);
            }
            if (NEED_V0) *(bf16x8 *)&V0s[j * DK + dc * 8] = s0;
            if (NEED_VL) *(bf16x8 *)&R[j * LDK + dc * 8] = sL;        // PH_B: scaled vL rows in LDS ([NP][LDK] at the start of R, free until the first A image)
        }
        if (DK < DP && PH != PH_B)          // zero rows of the transposed images (PH_B builds none: its R holds the vL rows here)
            for (int c = tl; c < (DP - DK) * LDA; c += NTH) { R[DK * LDA + c] = 0; R[(DP + DK) * LDA + c] = 0; R[(2 * DP + DK) * LDA + c] = 0; }
        __syncthreads();
        for (int c = tl; c < DP * LDA / 8; c += NTH) {
            if (NEED_KT) ((u32x4 *)KT)[c] = ((const u32x4 *)R)[c];
            if (NEED_QT) ((u32x4 *)QT)[c] = ((const u32x4 *)(R + DP * LDA))[c];
            if (NEED_DYT) ((u32x4 *)DYT)[c] = ((const u32x4 *)(R + 2 * DP * LDA))[c];
        }
        // qbar partials from the q^T image (row d = all tokens, padded ones are zero): thread (p, d) sums 32 tokens of row d with
        // four 16-byte LDS reads.  (32 five-step lane reductions of the q fragments cost ~80 k cycles here: every shuffle waited
        // on its own scratch-reloaded address.)
        if (PH == PH_A) {
            const int d = tl % DK, pp = tl / DK;            // NT partials per d (32 tokens each)
            if (pp < NT) {
                const unsigned short *qr = R + (DP + d) * LDA + 32 * pp;
                float sacc = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bf16x8 v8 = *(const bf16x8 *)&qr[8 * c];
#pragma unroll
                    for (int e = 0; e < 8; ++e) sacc += bf2f((unsigned short)v8[e]);
                }
                colpart[pp * DK + d] = sacc;
            }
        }
    }
    if (PH == PH_A) {
        bf16x8 qf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0}; if (qok) v = load8_bf16<IOT>(qrow + 16 * s + 8 * h); qf[s] = v; }
        __syncthreads();                                // Ksm staged, qbar partials written
        key_mean_partials<NT, DK>(rS, Ksm, N, tid);     // rS is overwritten with the row means right below
        __syncthreads();
        if (tid < DK) {
            float sk = 0.f, sq = 0.f;
            for (int p = 0; p < NT * 64 / DK; ++p) sk += rS[p * DK + tid];
            for (int ww = 0; ww < NT; ++ww) sq += colpart[ww * DK + tid];
            kbar[tid] = sk * invN; qbar[tid] = sq * invN;
        }
        __syncthreads();
        for (int v = 0; v < V; ++v) {
            float p = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int d = 16 * s + 8 * h + j; p = fmaf(bf2f((unsigned short)qf[s][j]) * sqk[v * DK + d], kbar[d], p); }
            p += __shfl_xor(p, 32, 64);
            if (h == 0) rS[v * NP + qi] = p;
        }
        col_means_mfma<NT, DK>(cS, Ksm, sqk, qbar, V, w, r, h);
    }
    __syncthreads();                     // P0 global images + LDS complete
    const float wv = misc[0];

    // ================= helpers =================
    auto make_frag = [&](bf16x8 (&qe)[KS], const IOT *row, const float *scale) {   // B fragments of a (q|dy) row, optional d-scale
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 qv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qok) qv = load8_bf16<IOT>(row + 16 * s + 8 * h);
            if (scale) {
                const float4 s0 = *(const float4 *)&scale[16 * s + 8 * h], s1 = *(const float4 *)&scale[16 * s + 8 * h + 4];
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) qv[j] = (short)f2bf(bf2f((unsigned short)qv[j]) * sc[j]);
            }
            qe[s] = qv;
        }
    };
    // Qe_v fragments from raw q fragments already in registers (no global round trip inside tile loops)
    auto scale_frag = [&](bf16x8 (&qe)[KS], const bf16x8 (&qraw)[KS], const float *scale) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 s0 = *(const float4 *)&scale[16 * s + 8 * h], s1 = *(const float4 *)&scale[16 * s + 8 * h + 4];
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) qe[s][j] = (short)f2bf(bf2f((unsigned short)qraw[s][j]) * sc[j]);
        }
    };
    auto s_tile = [&](const bf16x8 (&qe)[KS], int t) -> f32x16 {                   // S^T tile [key, query] from K in LDS
        const unsigned short *kbase = Ksm + r * LDK + 8 * h;   // lane base + compile-time offsets
        f32x16 acc = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 af = *(const bf16x8 *)&kbase[(32 * t) * LDK + 16 * s];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qe[s], acc, 0, 0, 0);
        }
        return acc;
    };
    auto g_tile = [&](const unsigned short *Arows, const bf16x8 (&fr)[KS], int t) -> f32x16 {   // (Arows[j,:] . frag) tile, A from global [NP][DK]
        f32x16 acc = zero16();
        bf16x8 af[KS];                    // all A fragments requested first (one round trip), then the MFMAs
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = *(const bf16x8 *)&(Arows + r * DK + 8 * h)[(32 * t) * DK + 16 * s];
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(af[s]));
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], fr[s], acc, 0, 0, 0);
        return acc;
    };
    auto a_tile = [&](const bf16x8 (&qe)[KS], int t, float c) -> f32x16 {   // one tile of A_v^T (keys >= N -> 0)
        f32x16 S = s_tile(qe, t);
#pragma unroll
        for (int g = 0; g < 16; ++g) S[g] = __builtin_amdgcn_exp2f(S[g] - c);
        if (32 * t + 32 > N) {
#pragma unroll
            for (int g = 0; g < 16; ++g) S[g] = (32 * t + tile_row(g, h) >= N) ? 0.f : S[g];
        }
        return S;
    };
    // form (i): dst[j][perm(i)] = X^T tile (A operand for products contracting over QUERIES)
    auto store_i_tile = [&](unsigned short *dst, int t, bf16x8 lo, bf16x8 hi) {
        unsigned short *base = dst + (32 * t + 4 * h) * LDA + 32 * w + 16 * (r >> 4) + kperm16(r & 15);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            base[((g & 3) + 8 * (g >> 2)) * LDA] = (unsigned short)lo[g];
            base[((g & 3) + 8 * (g >> 2) + 16) * LDA] = (unsigned short)hi[g];
        }
    };
    // form (ii): dst[i][perm(j)] = tile rows (A operand for products contracting over KEYS)
    auto store_ii_tile = [&](unsigned short *dst, int t, const f32x16 &X) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)   // regs 4g4..4g4+3 <-> keys 32t + 8g4 + 4h + {0..3}; permuted position: 16-group base + 8h + 4(g4&1)
            *(uint2 *)&(dst + qi * LDA + 8 * h)[32 * t + 16 * (g4 >> 1) + 4 * (g4 & 1)] =
                make_uint2(pack_bf16(X[4 * g4], X[4 * g4 + 1]), pack_bf16(X[4 * g4 + 2], X[4 * g4 + 3]));
    };
    // one 32x32 output tile of (LDS image rows) . (packed slab):  acc += Am[32to + r][:] . Xp
    auto gemm_tile = [&](f32x16 acc, const unsigned short *Am, int to, const bf16x8 (&Xp)[NT][2]) -> f32x16 {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 af = *(const bf16x8 *)&(Am + r * LDA + 8 * h)[(32 * to) * LDA + 32 * t + 16 * s];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, Xp[t][s], acc, 0, 0, 0);
            }
        return acc;
    };
    // two output tiles at once: two independent accumulator chains, so a wave issues an MFMA every 32 cycles instead of waiting
    // out the 64-cycle dependent-accumulate latency (and its own LDS read) between consecutive MFMAs of one chain
    auto gemm_tile2 = [&](f32x16 &acc0, f32x16 &acc1, const unsigned short *Am, int to0, const bf16x8 (&Xp)[NT][2]) {
        const unsigned short *ap = Am + r * LDA + 8 * h + (32 * to0) * LDA;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 af0 = *(const bf16x8 *)&ap[32 * t + 16 * s];
                const bf16x8 af1 = *(const bf16x8 *)&ap[32 * LDA + 32 * t + 16 * s];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0, Xp[t][s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, Xp[t][s], acc1, 0, 0, 0);
            }
    };
    // Xn = Am . Xp for all output tiles, result re-packed (chain state never exists as an fp32 slab)
    // Out[to] (+)= Am[32 to + r][:] . Bf  with packed bf16 output tiles: the NT x 2NT A-fragment reads form ONE stream that runs PF
    // fragments ahead of the MFMAs.  Reads and their waits are inline asm (hipcc sinks each compiler-visible read next to its MFMA
    // and waits lgkmcnt(0) behind it: read -> wait -> MFMA per k-step); asm statements keep their order and LDS returns in order, so
    // before fragment f only the younger reads may be outstanding.
    const bool klast = N > NP - 16;        // N <= NP - 16: the last 16-wide k-step of every contraction over tokens is all padding
    // ... and so is the last 16-B chunk of every packed slab (keys / queries NP - 16 .. NP - 1): it is neither stored nor loaded (7 % of
    // every N x N stream at N = 197).  Loads of it re-read the neighbouring chunk (cache-hot, branch-free) and are zeroed.
    const bool ctrim = HEAD == 0 && !klast;
    auto trim_idx = [&](int c) -> int { return (ctrim && c == 2 * NT - 1) ? c - 1 : c; };         // chunk actually loaded
    auto trim_val = [&](int c, u32x4 v) -> u32x4 { return (ctrim && c == 2 * NT - 1) ? u32x4{0, 0, 0, 0} : v; };
    auto gemm_stream = [&](bf16x8 (&Out)[NT][2], const unsigned short *Am, const bf16x8 (&Bf)[NT][2], bool accumulate) {
        gemm_stream_epi<NT>(Am + r * LDA + 8 * h, Bf, klast,
                            [&](int to) { return accumulate ? unpack_tile_bf(Out[to][0], Out[to][1]) : zero16(); },
                            [&](int to, const f32x16 &acc) { pack_tile_bf(Out[to][0], Out[to][1], acc); });
    };
    auto gemm_packed = [&](bf16x8 (&Xp)[NT][2], const unsigned short *Am) {      // Xp <- Am . Xp
        bf16x8 Xn[NT][2];
        gemm_stream(Xn, Am, Xp, false);
#pragma unroll
        for (int t = 0; t < NT; ++t) { Xp[t][0] = Xn[t][0]; Xp[t][1] = Xn[t][1]; }
    };
    auto load_rows = [&](bf16x8 (&Bf)[NT][2], const unsigned short *Bm) {      // this lane's row of an exported image ("row slab" order:
        const u32x4 *p = (const u32x4 *)Bm + (size_t)w * 2 * NT * 64 + lane;  //  [wave][2t+s][lane] -> coalesced 1 KiB per fragment)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) Bf[t][s] = as_b8(trim_val(2 * t + s, __builtin_nontemporal_load(&p[trim_idx(2 * t + s) * 64])));
    };
    // Pk[to] (+)= Am rows . Bf   with the running sum kept as packed bf16 tiles
    auto gemm_acc_packed = [&](bf16x8 (&Pk)[NT][2], const unsigned short *Am, const bf16x8 (&Bf)[NT][2], bool accumulate) {
#pragma unroll
        for (int to = 0; to < NT; ++to) {
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc = accumulate ? unpack_tile_bf(Pk[to][0], Pk[to][1]) : zero16();
            acc = gemm_tile(acc, Am, to, Bf);
            pack_tile_bf(Pk[to][0], Pk[to][1], acc);
        }
    };
    // barrier for LDS-only hand-offs: global loads / stores stay in flight across it (a __syncthreads() fence
    // would drain vmcnt and expose every prefetch)
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // out[dt] = Am[d][:] . Xp   (A rows d from a [DP][LDA] image, contraction over the slab's rows)
    auto gemm_small = [&](f32x16 (&out)[DT], const unsigned short *Am, const bf16x8 (&Xp)[NT][2]) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc = zero16();
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 af = *(const bf16x8 *)&(Am + r * LDA + 8 * h)[(32 * dt) * LDA + 32 * t + 16 * s];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, Xp[t][s], acc, 0, 0, 0);
                }
            out[dt] = acc;
        }
    };
    // out[dt] = sum_i Am[32w + r][i] * Bm[d][i]   (rows of this wave's tile of an LDS AT image, B rows d from global)
    auto gemm_rows_glob = [&](f32x16 (&out)[DT], const unsigned short *Am, const unsigned short *Bm) {
        // the B fragments come from global memory (L2): they are requested GLD k-steps ahead through a register ring, so a step's
        // MFMAs never wait for a load issued in the same step (one exposed L2 round trip per step otherwise: 28 per call)
        constexpr int NK = 2 * NT, GLD = 4;
        const unsigned short *ap = Am + (32 * w + r) * LDA + 8 * h, *bp = Bm + r * LDA + 8 * h;
        bf16x8 ring[GLD][DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) out[dt] = zero16();
#pragma unroll
        for (int k = 0; k < GLD && k < NK; ++k)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) ring[k][dt] = *(const bf16x8 *)&bp[(32 * dt) * LDA + 16 * k];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 af = *(const bf16x8 *)&ap[16 * k];
            bf16x8 cur[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) cur[dt] = ring[k % GLD][dt];
            if (k + GLD < NK) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) ring[k % GLD][dt] = *(const bf16x8 *)&bp[(32 * dt) * LDA + 16 * (k + GLD)];
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, cur[dt], out[dt], 0, 0, 0);
        }
    };
    auto slot_st = [&](int s, const bf16x8 (&Xp)[NT][2]) {
        __builtin_amdgcn_sched_barrier(0);
        u32x4 *p = slot(s);
#pragma unroll
        for (int t = 0; t < NT; ++t) { __builtin_nontemporal_store(as_u4(Xp[t][0]), &p[(2 * t) * 64]); if (!(ctrim && t == NT - 1)) __builtin_nontemporal_store(as_u4(Xp[t][1]), &p[(2 * t + 1) * 64]); }
    };
    auto slot_ld = [&](int s, bf16x8 (&Xp)[NT][2]) {
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 *p = slot(s);
#pragma unroll
        for (int t = 0; t < NT; ++t) { Xp[t][0] = as_b8(__builtin_nontemporal_load(&p[(2 * t) * 64])); Xp[t][1] = as_b8(trim_val(2 * t + 1, __builtin_nontemporal_load(&p[trim_idx(2 * t + 1) * 64]))); }
    };
    // "all fragments of this slab are needed here": without it hipcc sinks each load of a slot_ld next to the tile that consumes
    // it (to shorten live ranges), i.e. one exposed memory round trip per tile instead of one per slab
    auto pin_slab = [&](bf16x8 (&Xp)[NT][2]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(Xp[t][0]), "+v"(Xp[t][1]));
    };
    // image of A_v (form ii) or A_v^T (form i) streamed tile by tile into dst (LDS)
    auto a_image = [&](unsigned short *dst, int v, bool form_ii) {
        bf16x8 qe[KS];
        make_frag(qe, qrow, sqk2 + v * DK);
        const float c = cstats[v * NP + qi];
        STAMP2();
        lds_barrier();                    // previous readers of dst are done
        STAMP2();
#pragma nounroll
        for (int t = 0; t < NT; ++t) {
            const f32x16 A = a_tile(qe, t, c);
            if (form_ii) store_ii_tile(dst, t, A);
            else { bf16x8 lo, hi; pack_tile_bf(lo, hi, A); store_i_tile(dst, t, lo, hi); }
        }
        STAMP2();
        lds_barrier();
        STAMP2();
    };
    STAMP();
    REFRESH();
    if constexpr (PH == PH_A) {
    // ================= log-means of the chain products (from the forward) =================
    {
        const float *gm = (const float *)(svb + SL.oMeans);
        if (tid < NP) { rCr[tid] = gm[tid]; rCl[tid] = gm[NP + tid]; cCr[tid] = gm[2 * NP + tid]; cCl[tid] = gm[3 * NP + tid]; }
    }
    __syncthreads();
    STAMP();
    REFRESH();
    // ================= P4: mix state: Smix / L slabs and the final softmax row statistics come from the forward =================
    const float nb = a.beta_not / (float)(V > 1 ? V - 1 : 1);
    bf16x8 dyf[KS];
    make_frag(dyf, dyrow, nullptr);
    float mxrow, invl, delta;
    {
        // delta_i = sum_j P_ij dP_ij = dy_i . (P v0)_i with the forward's fp32 y_base = P v0
        const float *rw = (const float *)(svb + SL.oRow), *yb = (const float *)(svb + SL.oYb);
        mxrow = rw[qi]; invl = rw[NP + qi];
        float d = 0.f;
        if (qok) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 c0 = *(const float4 *)&yb[(size_t)qi * DK + 16 * s + 8 * h], c1 = *(const float4 *)&yb[(size_t)qi * DK + 16 * s + 8 * h + 4];
                const float yc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(bf2f((unsigned short)dyf[s][e]), yc[e], d);
            }
        }
        delta = d + __shfl_xor(d, 32, 64);
    }
    const FaDrop drop = fa_drop(a.dropout_p, a.dropout_seed);
    const uint32_t rowh = fa_drop_row(drop, bh, qi);
    auto p_tile = [&](int t) -> f32x16 {      // P tile from the parked Smix
        const u32x4 *p = slot(S_SM);
        u32x4 xh = __builtin_nontemporal_load(&p[trim_idx(2 * t + 1) * 64]);
        if (ctrim && t == NT - 1) xh = u32x4{0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u};       // trimmed chunk: Smix = -inf (padding keys)
        f32x16 x = unpack_tile_h(__builtin_nontemporal_load(&p[(2 * t) * 64]), xh);
#pragma unroll
        for (int g = 0; g < 16; ++g) x[g] = __expf(x[g] - mxrow) * invl;
        return x;
    };
    STAMP();
    REFRESH();
    STAMP();
    REFRESH();
    if constexpr (HEAD == 0) {
    // ================= P3: gate vectors =================
    // gate-head weights -> LDS (row side [16][WST], col side [16][WST]; slot WST-1 = bias); R's tail is free from here to P7
    const float *rowx = dw.rowx + (size_t)bh * (E * N), *colx = dw.colx + (size_t)bh * (E * N);
    for (int c = tid; c < 2 * 4 * RK * (C + 1); c += NTH) {
        const int side = c / (4 * RK * (C + 1)), rem = c % (4 * RK * (C + 1)), o = rem / (C + 1), cc = rem % (C + 1);
        const float *Wg = side ? a.Wc : a.Wr, *bg = side ? a.bc : a.br;
        Wsm[(side * 16 + o) * WST + (cc < C ? cc : WST - 1)] = cc < C ? Wg[o * C + cc] : bg[o];
    }
    __syncthreads();
    for (int p = tid; p < 4 * NP; p += NTH) {                  // b side: one (key j, gate g) pair per thread-iteration
        const int j = p % NP, g = p / NP;
        const int col = (j & ~15) + kperm16(j & 15);
        unsigned short hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
        if (j < N) {
            // the (up to) four rank channels of this gate side by side: four independent fma chains, every feature read once
            const float *Wg4 = Wsm + (16 + g * RK) * WST;
            float sk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sk[k] = Wg4[(k < RK ? k : 0) * WST + WST - 1];
            for (int c = 0; c < V; ++c) {
                const float fc = cS[c * NP + j], fr = rS[c * NP + j];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float *Wo = Wg4 + (k < RK ? k : 0) * WST; sk[k] = fmaf(Wo[c], fc, fmaf(Wo[V + c], fr, sk[k])); }
            }
            {
                const float f0 = cCr[j], f1 = cCl[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float *Wo = Wg4 + (k < RK ? k : 0) * WST; sk[k] = fmaf(Wo[2 * V], f0, fmaf(Wo[2 * V + 1], f1, sk[k])); }
            }
            for (int e0 = 0; e0 < E; e0 += XU) {
                float fx[XU];
                load_extra(fx, colx, N, j, e0, E, true);
                for (int u = 0; u < XU && e0 + u < E; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) sk[k] = fmaf(Wg4[(k < RK ? k : 0) * WST + 2 * V + 2 + e0 + u], fx[u], sk[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < RK) { hi[k] = f2bf(sk[k]); lo[k] = f2bf(sk[k] - bf2f(hi[k])); }
        }
        unsigned short *row = bT + (g * NP + j) * BTS;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            row[k] = hi[k]; row[4 + k] = hi[k]; row[8 + k] = lo[k]; row[12 + k] = 0;
            bmat[(4 * g + k) * LDA + col] = hi[k];
            bmat[(16 + 4 * g + k) * LDA + col] = lo[k];
        }
    }
    bf16x8 af4[4];
    float av16[4][4];                      // a[g,k] for this lane's query: 16 independent fma chains over the feature channels
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) av16[g][k] = Wsm[(g * RK + (k < RK ? k : 0)) * WST + WST - 1];
    for (int c = 0; c < V; ++c) {
        const float fr = rS[c * NP + qi], fc = cS[c * NP + qi];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float *Wo = Wsm + (g * RK + (k < RK ? k : 0)) * WST; av16[g][k] = fmaf(Wo[c], fr, fmaf(Wo[V + c], fc, av16[g][k])); }
    }
    {
        const float f0 = rCr[qi], f1 = rCl[qi];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float *Wo = Wsm + (g * RK + (k < RK ? k : 0)) * WST; av16[g][k] = fmaf(Wo[2 * V], f0, fmaf(Wo[2 * V + 1], f1, av16[g][k])); }
    }
    for (int e0 = 0; e0 < E; e0 += XU) {
        float fx[XU];
        load_extra(fx, rowx, N, qi, e0, E, qok);
        for (int u = 0; u < XU && e0 + u < E; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) av16[g][k] = fmaf(Wsm[(g * RK + (k < RK ? k : 0)) * WST + 2 * V + 2 + e0 + u], fx[u], av16[g][k]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float av[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) av[k] = k < RK ? av16[g][k] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned short hi = f2bf(av[k]), lo = f2bf(av[k] - bf2f(hi));
            const float a2 = av[k] * 1.4426950408889634f;           // Z MFMA operand pre-scaled: sigmoid = 1/(1+2^-z')
            const unsigned short hi2 = f2bf(a2), lo2 = f2bf(a2 - bf2f(hi2));
            af4[g][k] = (short)hi2;
            af4[g][4 + k] = h == 0 ? (short)lo2 : (short)0;
            if (h == 0) { const int col = 32 * w + 16 * (r >> 4) + kperm16(r & 15); amat[(4 * g + k) * LDA + col] = qok ? hi : (unsigned short)0; amat[(16 + 4 * g + k) * LDA + col] = qok ? lo : (unsigned short)0; }
        }
    }
    __syncthreads();
    STAMP();
    REFRESH();
    auto gate_tile = [&](int t, int g4) -> f32x16 {
        const bf16x8 bfrag = *(const bf16x8 *)&bT[(g4 * NP + 32 * t + r) * BTS + 8 * h];
        f32x16 z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfrag, af4[g4], zero16(), 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) z[g] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-z[g]));
        return z;
    };
    // ================= P6: mix backward =================
    f32x16 daacc = zero16();
#pragma nounroll
    for (int t = 0; t < NT; ++t) {
        f32x16 dS;                                   // dSmix tile
        bf16x8 idl, idh;                              // B fragments of the 32 x 32 identity in the accumulator's k order (for the transposes below)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            idl[e] = (short)(r == tile_row(e, h) ? 0x3f80 : 0);          // bf16(1.0)
            idh[e] = (short)(r == 16 + tile_row(e, h) ? 0x3f80 : 0);
        }
        // this tile's q / dy fragments are (re)read from L2 in one batch: kept resident across the loop they are spilled and
        // every use (6 + 1 per tile) starts with its own exposed scratch reload
        bf16x8 qraw_t[KS], dyf_t[KS];
        make_frag(qraw_t, qrow, nullptr);
        make_frag(dyf_t, dyrow, nullptr);
        {
            const f32x16 P = p_tile(t);
            f32x16 dP = g_tile(V0s, dyf_t, t);
            if (drop.thresh) {                 // attn_drop: dP = (dy v0^T) keep / (1 - p); delta = dy . y_base already carries it
#pragma unroll
                for (int g = 0; g < 16; ++g) dP[g] = fa_drop_keep(drop, rowh, 32 * t + tile_row(g, h)) ? dP[g] * drop.inv_keep : 0.f;
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) dS[g] = P[g] * (dP[g] - delta);      // padding keys: the record holds Smix = -inf there, so P = 0 and with it dS
        }
        // pass 1: gates -> gA = G_and - nb G_not, g1 = G_or ; lse tile ; per-view direct score gradients
        //   dS_v(direct) = dSmix * (v == 0 ? 1 - g1 + g1 pi_0 : gA + g1 pi_v),  pi_v = exp(S_v - lse)   -> parked per view
        f32x16 O = zero16(), lse;
        {
            f32x16 gA = gate_tile(t, 0), g1;
            { const f32x16 G2 = gate_tile(t, 2);
#pragma unroll
              for (int g = 0; g < 16; ++g) gA[g] = fmaf(-nb, G2[g], gA[g]); }
            g1 = gate_tile(t, 1);
            bf16x8 qe[KS];
            {   // L = lse - S0, exported by the forward in fp32 (x log2 e)
                const f32x4 *p = (const f32x4 *)(svb + SL.oL + (size_t)w * 2 * Cfg::SLOT) + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool ltr = ctrim && t == NT - 1 && q >= 2;
                    f32x4 l4 = __builtin_nontemporal_load(&p[(4 * t + (ltr ? q - 2 : q)) * 64]);
                    if (ltr) l4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    lse[4 * q] = l4[0] * 0.6931471805599453f; lse[4 * q + 1] = l4[1] * 0.6931471805599453f;
                    lse[4 * q + 2] = l4[2] * 0.6931471805599453f; lse[4 * q + 3] = l4[3] * 0.6931471805599453f;
                }
            }
            for (int v = 0; v < V; ++v) {
                scale_frag(qe, qraw_t, sqk + v * DK);
                const f32x16 Sv = s_tile(qe, t);
                if (v == 0) lse += Sv; else O += Sv;
                f32x16 dir;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float pi = __expf(Sv[g] - lse[g]);
                    const float coef = v == 0 ? (1.f - g1[g]) + g1[g] * pi : fmaf(g1[g], pi, gA[g]);
                    dir[g] = dS[g] * coef;
                }
                bf16x8 bl, bh;
                pack_tile_bf(bl, bh, dir);
                u32x4 *p = slot(X_DIR + v); __builtin_nontemporal_store(as_u4(bl), &p[(2 * t) * 64]); if (!(ctrim && t == NT - 1)) __builtin_nontemporal_store(as_u4(bh), &p[(2 * t + 1) * 64]);
            }
        }
        // pass 2: gate gradients.  terms: and -> O, or -> L = lse - S0, not -> -nb O, chain -> log C->
        f32x16 L, Cr;
        {
            bf16x8 qe[KS];
            scale_frag(qe, qraw_t, sqk);
            const f32x16 S0 = s_tile(qe, t);
            L = lse - S0;
            const u32x4 *pc = slot(S_CF);
            Cr = unpack_tile_bf(as_b8(pc[(2 * t) * 64]), as_b8(trim_val(2 * t + 1, pc[trim_idx(2 * t + 1) * 64])));
#pragma unroll
            for (int g = 0; g < 16; ++g) Cr[g] = __logf(Cr[g] + EPSC);
        }
        f32x16 dbt = zero16();
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x16 G = gate_tile(t, g4);
            f32x16 dZ;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float term = g4 == 0 ? O[g] : (g4 == 1 ? L[g] : (g4 == 2 ? -nb * O[g] : Cr[g]));
                dZ[g] = dS[g] * term * G[g] * (1.f - G[g]);
            }
            if (g4 == 3) {
                bf16x8 bl, bh;
                f32x16 c3;
#pragma unroll
                for (int g = 0; g < 16; ++g) c3[g] = dS[g] * G[g];
                pack_tile_bf(bl, bh, c3);
                u32x4 *p = slot(X_C3); __builtin_nontemporal_store(as_u4(bl), &p[(2 * t) * 64]); if (!(ctrim && t == NT - 1)) __builtin_nontemporal_store(as_u4(bh), &p[(2 * t + 1) * 64]);
            }
            // gate-logit gradients enter two contractions over ~N^2 edges whose result is a small difference of large sums: dZ is
            // split into a bf16 value and its bf16 remainder (as a and b already are), so the products are fp32-accurate
            bf16x8 zl, zh, yl, yh;
            pack_tile_bf(zl, zh, dZ);
            {
                const f32x16 rem = dZ - unpack_tile_bf(zl, zh);
                pack_tile_bf(yl, yh, rem);
            }
            const bool mine = ((r >> 2) & 3) == g4;               // rows 4g4..4g4+3 (hi) and 16+4g4.. (lo)
            // da[rho, i] += sum_j bmat_g[rho][j] dZ^T[j, i]      (rows of gate g4 only)
            {
                bf16x8 a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
                if (mine) {
                    a0 = *(const bf16x8 *)&bmat[r * LDA + 32 * t + 8 * h];
                    a1 = *(const bf16x8 *)&bmat[r * LDA + 32 * t + 16 + 8 * h];
                }
                daacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, zl, daacc, 0, 0, 0);
                daacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, zh, daacc, 0, 0, 0);
                daacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, yl, daacc, 0, 0, 0);
                daacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, yh, daacc, 0, 0, 0);
            }
            // db[rho, j] (keys of tile t) = sum_i amat_g[rho][i] dZ[i, j]: the contraction runs over this wave's queries (= lanes), so
            // the tile is transposed on the matrix core (X . I: lane = key, registers = queries; exact for bf16 values) and its packed
            // halves are the B fragments in the accumulator's k order -- amat's columns are stored in that order
            {
                bf16x8 aa0 = {0, 0, 0, 0, 0, 0, 0, 0}, aa1 = aa0;
                if (mine) {
                    aa0 = *(const bf16x8 *)&amat[r * LDA + 32 * w + 8 * h];
                    aa1 = *(const bf16x8 *)&amat[r * LDA + 32 * w + 16 + 8 * h];
                }
                f32x16 tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zl, idl, zero16(), 0, 0, 0);
                tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh, idh, tr, 0, 0, 0);
                bf16x8 tl, th;
                pack_tile_bf(tl, th, tr);
                dbt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa0, tl, dbt, 0, 0, 0);
                dbt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa1, th, dbt, 0, 0, 0);
                tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl, idl, zero16(), 0, 0, 0);
                tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh, idh, tr, 0, 0, 0);
                pack_tile_bf(tl, th, tr);
                dbt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa0, tl, dbt, 0, 0, 0);
                dbt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa1, th, dbt, 0, 0, 0);
            }
        }
        // dbt rows rho (hi: regs 0-7, lo: regs 8-15), lanes = keys of tile t
#pragma unroll
        for (int g = 0; g < 8; ++g) dbp[((size_t)w * 16 + tile_row(g, h)) * NP + 32 * t + r] = dbt[g] + dbt[8 + g];
    }
    __syncthreads();                                   // bT / bmat / amat / tbuf dead; dbp complete
#pragma unroll
    for (int g = 0; g < 8; ++g) dav[tile_row(g, h) * NP + qi] = daacc[g] + daacc[8 + g];
    for (int c = tid; c < 16 * NP; c += NTH) {
        float pv[NT];                     // the NT partials are requested together (the compiler chained load -> wait -> add otherwise)
#pragma unroll
        for (int ww = 0; ww < NT; ++ww) pv[ww] = dbp[(size_t)ww * 16 * NP + c];
#pragma unroll
        for (int ww = 0; ww < NT; ++ww) asm volatile("" : "+v"(pv[ww]));
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NT; ++ww) s += pv[ww];
        dbv[c] = s;
    }
    __syncthreads();
    STAMP();
    REFRESH();
    // ================= P7: gradients of the gate-head inputs / weights =================
    {
        // dW partials first (they read rS/cS which dmean is about to overwrite)
        const int nO = 4 * RK;
        float *xfeat = dbv + 16 * NP;                 // [2][E][NP] extra feature rows (row side, then col side), zero padded; behind dav | dbv, below Wsm
        if (E) {            // thread = (token n, side): XU channels requested per round trip
            const int n = tid % NP, sd = tid / NP;
            if (sd < 2)
                for (int e0 = 0; e0 < E; e0 += XU) {
                    float fx[XU];
                    load_extra(fx, sd ? colx : rowx, N, n, e0, E, n < N);
                    for (int u = 0; u < XU && e0 + u < E; ++u) xfeat[(sd * E + e0 + u) * NP + n] = fx[u];
                }
            __syncthreads();
        }
        for (int idx = tid; idx < 2 * nO * (C + 1); idx += NTH) {
            const int side = idx / (nO * (C + 1)), rem = idx % (nO * (C + 1));
            const int o = rem / (C + 1), c = rem % (C + 1);
            const int rho = 4 * (o / RK) + (o % RK);
            const float *g = (side ? dbv : dav) + rho * NP;
            // feature row of this (side, channel): row side = [rS | cS | rCr rCl], col side = [cS | rS | cCr cCl]
            const float *f = nullptr;
            if (c < V) f = (side ? cS : rS) + c * NP;
            else if (c < 2 * V) f = (side ? rS : cS) + (c - V) * NP;
            else if (c == 2 * V) f = side ? cCr : rCr;
            else if (c == 2 * V + 1) f = side ? cCl : rCl;
            else if (c < C) f = xfeat + ((side ? E : 0) + c - (2 * V + 2)) * NP;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int n = 0;
            if (f) {         // 16-byte LDS reads (rows are 16-byte aligned): a quarter of the LDS instructions of the scalar loop
                for (; n + 4 <= N; n += 4) {
                    const float4 gv = *(const float4 *)&g[n], fv = *(const float4 *)&f[n];
                    s0 = fmaf(gv.x, fv.x, s0); s1 = fmaf(gv.y, fv.y, s1); s2 = fmaf(gv.z, fv.z, s2); s3 = fmaf(gv.w, fv.w, s3);
                }
                for (; n < N; ++n) s0 = fmaf(g[n], f[n], s0);
            } else {
                for (; n + 4 <= N; n += 4) { const float4 gv = *(const float4 *)&g[n]; s0 += gv.x; s1 += gv.y; s2 += gv.z; s3 += gv.w; }
                for (; n < N; ++n) s0 += g[n];
            }
            const float s = (s0 + s1) + (s2 + s3);
            dwp[idx] = first_pass ? s : dwp[idx] + s;
        }
        __syncthreads();
        // dmean rows: [0,V) drS_v ; [V,2V) dcS_v ; 2V drCr ; 2V+1 dcCr ; 2V+2 drCl ; 2V+3 dcCl   (pre-divided by N)
        // row m gathers the row-side channel cr and/or the col-side channel cc of the head's input gradient
        for (int item = tid; item < (2 * V + 4) * NP; item += NTH) {
            const int m = item / NP, n = item % NP;
            const int cr = m < 2 * V ? m : (m == 2 * V ? 2 * V : (m == 2 * V + 2 ? 2 * V + 1 : -1));
            const int cc = m < V ? V + m : (m < 2 * V ? m - V : (m == 2 * V + 1 ? 2 * V : (m == 2 * V + 3 ? 2 * V + 1 : -1)));
            float sr = 0.f, sc = 0.f;
            if (n < N)
                for (int g4 = 0; g4 < 4; ++g4)          // o = g4 RK + k, rho = 4 g4 + k: nested so that no runtime division is needed
                    for (int k = 0; k < RK; ++k) {
                        const int o = g4 * RK + k, rho = 4 * g4 + k;
                        sr = fmaf(Wsm[o * WST + (cr >= 0 ? cr : 0)], dav[rho * NP + n], sr);
                        sc = fmaf(Wsm[(16 + o) * WST + (cc >= 0 ? cc : 0)], dbv[rho * NP + n], sc);
                    }
            const float dm = ((cr >= 0 ? sr : 0.f) + (cc >= 0 ? sc : 0.f)) * invN;
            xdmean[item] = dm;                // hand-off: PH_B / PH_C stage these vectors in their own LDS
        }
        // gradient with respect to the extra feature rows: straight to the caller (they are inputs of the kernel, not functions of q, k)
        for (int item = tid; item < 2 * E * N; item += NTH) {
            const int sd = item / (E * N), e = (item / N) % E, n = item % N;
            const float *gv = sd ? dbv : dav, *Wx = Wsm + (sd ? 16 * WST : 0) + 2 * V + 2 + e;
            float sx = 0.f;
            for (int g4 = 0; g4 < 4; ++g4)
                for (int k = 0; k < RK; ++k) sx = fmaf(Wx[(g4 * RK + k) * WST], gv[(4 * g4 + k) * NP + n], sx);
            (sd ? dw.dcolx : dw.drowx)[(size_t)bh * (E * N) + e * N + n] = sx;
        }
        __syncthreads();
    }
    } else {
    // ================= dense gate head (HEAD == 1; reference :250-272 minus the 3x3, :312-318) =================
    // Per pair of tile registers (two edges per lane): features f = [S_v, S_v^T, Cr, Cl] recomputed on the matrix core (one chain per
    // operand order for all views), then per edge the MLP forward (gates), the mix backward dG_g = dSmix * term_g and the MLP backward
    // -> dz2, dz1, df.  df joins the direct score gradients (S_v channels: DIR_v; S_v^T channels: their own slabs, added transposed by
    // launch C), C3 (Cr) and a new slab (Cl, seeds the <- chain in launch B).  Weight gradients are sums over edges = sums over lanes,
    // which no MFMA contracts: each half-wave parks its edges' rows [dz1 | f, 1 | h | dz2] (fp32) in a wave-private LDS buffer and
    // v_mfma_f32_16x16x4_f32 (exact fp32) sums dz1 (x) [f, 1] and h (x) dz2 over them, four edges per issue; the two accumulator
    // tiles live in registers for the whole (b,h).
    {
        unsigned short *Qsm = R;                                    // q rows [token][d]: A operand of the S_v^T tiles
        float *wacc = (float *)(R + NP * LDK);                      // [NT][WACC] weight-gradient sums of this (b,h), per wave
        constexpr int WACC = 16 * 16 + 16 * 4 + 4;                  // [c][k]: dW1[k][c] for c < C <= 14, row 15 = db1[k] | dW2^T [k][m] | db2[m]
        constexpr int RS = 52, STG = 32 * RS + 16;                  // staging row (floats): dz1 16 | f .. 1 .. 16 | h 16 | dz2 4; 16 floats of over-read pad
        static_assert(NP * LDK * 2 + NT * (WACC + STG) * 4 <= Cfg::R_BYTES, "dense launch A: q rows + tables + staging rows live in R");
        float *stg = wacc + NT * WACC + w * STG;                    // this wave's 32 staging rows
        {
            const IOT *qp = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh;
            constexpr int CH = DK / 8;
            for (int c = tid; c < NP * CH; c += NTH) {
                const int j = c / CH, dc = c % CH;
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (j < N) v = load8_bf16<IOT>(qp + (int64_t)j * a.q.sn + dc * 8);
                *(bf16x8 *)&Qsm[j * LDK + dc * 8] = v;
            }
            // weights transposed as in the forward: W1T[c][k] at Wsm[c * 16 + k], b1 at Wsm[288 + k], W2T[k][m] at Wsm[320 + 4 k + m], b2 at Wsm[384 + m]
            for (int c = tid; c < 16 * C; c += NTH) Wsm[(c % C) * 16 + c / C] = dw.W1[c];
            for (int c = tid; c < 16; c += NTH) Wsm[18 * 16 + c] = dw.b1[c];
            for (int c = tid; c < 64; c += NTH) Wsm[320 + (c % 16) * 4 + c / 16] = dw.W2[c];
            if (tid < 4) Wsm[384 + tid] = dw.b2[tid];
            for (int c = tid; c < NT * (WACC + STG); c += NTH) wacc[c] = 0.f;
        }
        __syncthreads();
        float *wme = wacc + w * WACC;
        auto st_tile = [&](const bf16x8 (&ke)[KS], int t) -> f32x16 {      // S_v^T tile: element (lane = my token i, register = token j) = S_v(j, i)
            f32x16 acc = zero16();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 af = *(const bf16x8 *)&(Qsm + r * LDK + 8 * h)[(32 * t) * LDK + 16 * s];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ke[s], acc, 0, 0, 0);
            }
            return acc;
        };
        typedef __attribute__((ext_vector_type(2))) float f32x2;   // the two edges of a pass: v_pk_fma_f32 / v_pk_mul_f32 do both per issue
        auto bc = [](float x) -> f32x2 { return f32x2{x, x}; };
        f32x4 aw1 = {0.f, 0.f, 0.f, 0.f}, aw2 = aw1;                // dz1 (x) [f, 1] and h (x) dz2 sums of this wave (16x16 accumulator tiles)
        float db2a[4] = {0.f, 0.f, 0.f, 0.f};                       // per-lane db2 partials
#pragma nounroll
        for (int t = 0; t < NT; ++t) {
            // the tile's dP stays resident across the register groups; the q / k fragments of this lane's token are re-read from the
            // LDS images per group (32 fewer registers live through the MLP part)
            f32x16 dPt;
            {
                bf16x8 dyf_t[KS];
                make_frag(dyf_t, dyrow, nullptr);
                dPt = g_tile(V0s, dyf_t, t);                        // dP = dy v0^T for the whole tile; quarters index into it
            }
            // E edges per lane and pass: registers E qq .. E qq + E - 1 of the tile.  E = 2 is what one (key, view)-row chain covers for
            // V <= 8 (16 accumulator registers per lane half); the MLP then runs per edge.
            constexpr int E = 2, NQ = 16 / E;
#pragma nounroll
            for (int qq = 0; qq < NQ; ++qq) {
                const int g0 = E * qq;                                // first register of this group (even)
                // packed tiles hold elements 2d, 2d + 1 in dword d of the (lo | hi) pair of 16-byte vectors
                auto getw = [&](const u32x4 *p, int g) -> unsigned int { return ((const unsigned int *)&p[(2 * t + (g >> 3)) * 64])[(g >> 1) & 3]; };
                auto putw = [&](u32x4 *p, int g, float x0, float x1) { ((unsigned int *)&p[(2 * t + (g >> 3)) * 64])[(g >> 1) & 3] = pack_bf16(x0, x1); };
                f32x2 dSq, Lq, Crq, Clq;
                {
                    const float *lp = (const float *)((const f32x4 *)(svb + SL.oL + (size_t)w * 2 * Cfg::SLOT) + lane);     // [4 t + q][lane] x 16 B
                    const int g = g0;
                    const unsigned int sm = getw(slot(S_SM), g), cf = getw(slot(S_CF), g), cb = getw(slot(S_CB), g);
                    const float2 l2 = *(const float2 *)&lp[(size_t)(4 * t + (g >> 2)) * 64 * 4 + (g & 3)];
                    const float smx[2] = {h2_lo(sm), h2_hi(sm)}, lv[2] = {l2.x, l2.y};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int j = 32 * t + tile_row(g + e, h);
                        float dp = dPt[g + e];
                        if (drop.thresh) dp = fa_drop_keep(drop, rowh, j) ? dp * drop.inv_keep : 0.f;
                        const float P = __expf(smx[e] - mxrow) * invl;
                        dSq[e] = keep_if(j < N, P * (dp - delta));
                        Lq[e] = lv[e] * 0.6931471805599453f;
                        Crq[e] = __logf(bf2f((unsigned short)(cf >> (16 * e))) + EPSC);
                        Clq[e] = __logf(bf2f((unsigned short)(cb >> (16 * e))) + EPSC);
                    }
                }
                // ---- all views' scores of the group's two edges from ONE matrix-core chain per operand order: the 32 A rows are
                //      (key, view) pairs -- row m feeds accumulator register (m & 3) + 4 (m >> 3) of lane half (m >> 2) & 1, so register v
                //      (8 + v) of a lane is S_v of its first (second) edge -- and each A lane scales its key's row by its view's sqk on the
                //      way in (8 products per k-step, instead of re-scaling this lane's token per view and running V chains)
                f32x16 aS = zero16(), aT = zero16();
                {
                    const int m = lane & 31, hp = (m >> 2) & 1, gq = (m & 3) + 4 * (m >> 3), va = min(gq & 7, V - 1);
                    const int key = 32 * t + tile_row(g0 + (gq >> 3), hp);
                    const float *sca = sqk + va * DK + 8 * h;
                    const unsigned short *kr = Ksm + key * LDK + 8 * h, *qr = Qsm + key * LDK + 8 * h;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const float4 s0 = *(const float4 *)&sca[16 * s], s1 = *(const float4 *)&sca[16 * s + 4];
                        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                        const bf16x8 rk = *(const bf16x8 *)&kr[16 * s], rq = *(const bf16x8 *)&qr[16 * s];
                        bf16x8 ak, aq;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            ak[j] = (short)f2bf(bf2f((unsigned short)rk[j]) * sc[j]);
                            aq[j] = (short)f2bf(bf2f((unsigned short)rq[j]) * sc[j]);
                        }
                        const bf16x8 qmine = *(const bf16x8 *)&Qsm[(32 * w + r) * LDK + 16 * s + 8 * h];
                        const bf16x8 kmine = *(const bf16x8 *)&Ksm[(32 * w + r) * LDK + 16 * s + 8 * h];
                        aS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qmine, aS, 0, 0, 0);      // S_v(i, j) = (k_j * sqk_v) . q_i
                        aT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kmine, aT, 0, 0, 0);      // S_v(j, i) = (q_j * sqk_v) . k_i
                    }
                }
                // ---- the MLP, one edge at a time (rolled: the per-edge state -- 16 pre-activations, 16 hidden values, the feature vector --
                //      stays inside the register budget; two edges at once spilled ~90 scratch accesses per group into the hot loop)
                f32x16 O0 = zero16();                               // first edge's channel outputs, kept until the second edge's are known (one packed store)
#pragma nounroll
                for (int e = 0; e < E; ++e) {
                    const float dSe = e ? dSq[1] : dSq[0], Le = e ? Lq[1] : Lq[0], Cre = e ? Crq[1] : Crq[0], Cle = e ? Clq[1] : Clq[0];
                    float z1[16];                                   // pre-activation, then gelu', then dz1 of hidden unit k
                    f32x16 F = zero16();                            // F[c]: feature channel c (C <= 14), slot C = 1 (the bias column); c may be a runtime index
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const float4 bv = *(const float4 *)&Wsm[18 * 16 + 4 * k4];
                        z1[4 * k4] = bv.x; z1[4 * k4 + 1] = bv.y; z1[4 * k4 + 2] = bv.z; z1[4 * k4 + 3] = bv.w;
                    }
                    auto accum = [&](int c, float f) {
                        F[c] = f;
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) {
                            const float4 wv4 = *(const float4 *)&Wsm[c * 16 + 4 * k4];
                            z1[4 * k4] = fmaf(wv4.x, f, z1[4 * k4]); z1[4 * k4 + 1] = fmaf(wv4.y, f, z1[4 * k4 + 1]);
                            z1[4 * k4 + 2] = fmaf(wv4.z, f, z1[4 * k4 + 2]); z1[4 * k4 + 3] = fmaf(wv4.w, f, z1[4 * k4 + 3]);
                        }
                    };
                    float Oe = 0.f, S0e = 0.f;
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        if (v < V) {
                            const float fs = e ? aS[8 + v] : aS[v];
                            accum(v, fs);
                            if (v == 0) S0e = fs; else Oe += fs;
                            accum(V + v, e ? aT[8 + v] : aT[v]);
                        }
                    }
                    accum(2 * V, Cre);
                    accum(2 * V + 1, Cle);
                    F[C] = 1.f;
                    // gelu, its derivative (kept in place of z1: no transcendental in the second sweep), second layer and gates
                    float zz[4], hq[16];
                    {
                        const float4 bv = *(const float4 *)&Wsm[384];
                        zz[0] = bv.x; zz[1] = bv.y; zz[2] = bv.z; zz[3] = bv.w;
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float4 wv4 = *(const float4 *)&Wsm[320 + 4 * k];
                        const float u = z1[k], u2 = u * u;
                        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-1.5957691216057308f * (u + 0.044715f * u * u2)));
                        const float hv = u * sg;
                        z1[k] = sg + hv * (1.f - sg) * (1.5957691216057308f + 0.21406444881780076f * u2);     // d gelu_tanh / du
                        hq[k] = hv;
                        zz[0] = fmaf(wv4.x, hv, zz[0]); zz[1] = fmaf(wv4.y, hv, zz[1]); zz[2] = fmaf(wv4.z, hv, zz[2]); zz[3] = fmaf(wv4.w, hv, zz[3]);
                    }
                    float G[4], dzz[4];
                    {
                        const float term[4] = {Oe, Le, -nb * Oe, Cre};  // d Smix / d G_g: and -> O, or -> L = lse - S0, not -> -nb O, chain -> log C->
#pragma unroll
                        for (int mg = 0; mg < 4; ++mg) {
                            G[mg] = __builtin_amdgcn_rcpf(1.f + __expf(-zz[mg]));
                            dzz[mg] = dSe * term[mg] * G[mg] * (1.f - G[mg]);
                            db2a[mg] += dzz[mg];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) {                    // dz1 = (W2^T dz2) * gelu'
                        const float4 wv4 = *(const float4 *)&Wsm[320 + 4 * k];
                        z1[k] *= fmaf(wv4.w, dzz[3], fmaf(wv4.z, dzz[2], fmaf(wv4.y, dzz[1], wv4.x * dzz[0])));
                    }
                    // weight-gradient sums on the f32 matrix core: per lane half, 32 rows parked, 8 k-steps of 4 edges
#pragma unroll
                    for (int hs = 0; hs < 2; ++hs) {
                        if (h == hs) {
                            f32x4 *row = (f32x4 *)(stg + r * RS);
#pragma unroll
                            for (int k4 = 0; k4 < 4; ++k4) {
                                row[k4] = f32x4{z1[4 * k4], z1[4 * k4 + 1], z1[4 * k4 + 2], z1[4 * k4 + 3]};
                                row[4 + k4] = f32x4{F[4 * k4], F[4 * k4 + 1], F[4 * k4 + 2], F[4 * k4 + 3]};
                                row[8 + k4] = f32x4{hq[4 * k4], hq[4 * k4 + 1], hq[4 * k4 + 2], hq[4 * k4 + 3]};
                            }
                            row[12] = f32x4{dzz[0], dzz[1], dzz[2], dzz[3]};
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const float *rd = stg + (lane >> 4) * RS + (lane & 15);
#pragma unroll
                        for (int s4 = 0; s4 < 8; ++s4) {
                            const float *q4 = rd + 4 * s4 * RS;
                            aw1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[0], q4[16], aw1, 0, 0, 0);
                            aw2 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[32], q4[48], aw2, 0, 0, 0);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    // channels: df_c = sum_k W1[k][c] dz1[k] -> hand-off slabs (rolled: unrolled over the channels it spills into the hot loop)
                    const float g1e = G[1], gAe = fmaf(-nb, G[2], G[0]), lsee = Le + S0e;
                    for (int c = 0; c < C; ++c) {
                        const float f = F[c];
                        const float4 *wr = (const float4 *)&Wsm[c * 16];
                        const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
                        float d0 = w0.x * z1[0], d1 = w1.x * z1[4], d2 = w2.x * z1[8], d3 = w3.x * z1[12];       // four independent chains
                        d0 = fmaf(w0.y, z1[1], d0); d1 = fmaf(w1.y, z1[5], d1); d2 = fmaf(w2.y, z1[9], d2); d3 = fmaf(w3.y, z1[13], d3);
                        d0 = fmaf(w0.z, z1[2], d0); d1 = fmaf(w1.z, z1[6], d1); d2 = fmaf(w2.z, z1[10], d2); d3 = fmaf(w3.z, z1[14], d3);
                        d0 = fmaf(w0.w, z1[3], d0); d1 = fmaf(w1.w, z1[7], d1); d2 = fmaf(w2.w, z1[11], d2); d3 = fmaf(w3.w, z1[15], d3);
                        // one store site for every channel kind (the slab id and the value are selected arithmetically): S_v channels -> DIR_v
                        // (direct score gradient dSmix * coef_v + df), S_v^T channels -> their own slabs (gradient of S_v(j, i), added transposed
                        // by launch C), Cr -> C3 (joins the chain-gate term G_chain dSmix), Cl -> the slab that seeds the <- D-chain (launch B)
                        const int sid = c < V ? X_DIR + c : (c < 2 * V ? X_DT(V) + (c - V) : (c == 2 * V ? (int)X_C3 : X_CL(V)));
                        const float pi = __expf(fminf(f - lsee, 0.f));              // only meaningful for the S_v channels (pi <= 1 there)
                        const float coef = c == 0 ? (1.f - g1e) + g1e * pi : fmaf(g1e, pi, gAe);
                        const float direct = c < V ? dSe * coef : (c == 2 * V ? dSe * G[3] : 0.f);
                        const float o = direct + ((d0 + d1) + (d2 + d3));
                        if (e == 0) O0[c] = o;
                        else putw(slot(sid), g0, O0[c], o);
                    }
                }
            }
        }
        // the wave's accumulator tiles -> its table: D[row = 4 (lane >> 4) + i][col = lane & 15]
        {
            const int col = lane & 15, row0 = 4 * (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (col <= C) wme[16 * (col < C ? col : 15) + row0 + i] = aw1[i];      // [c][k]; column C = the bias column -> row 15
                if (col < 4) wme[16 * 16 + 4 * (row0 + i) + col] = aw2[i];             // dW2^T [k][m]
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float sb = wave_sum(db2a[m]);
                if (lane == 0) wme[16 * 16 + 16 * 4 + m] = sb;
            }
        }
        __syncthreads();
        // weight-gradient partials of this workgroup: waves summed in a fixed order, accumulated over the (b,h) it processes
        for (int c = tid; c < WACC; c += NTH) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NT; ++ww) s += wacc[ww * WACC + c];
            dwp[c] = first_pass ? s : dwp[c] + s;
        }
        for (int c = tid; c < (2 * V + 4) * NP; c += NTH) xdmean[c] = 0.f;      // the dense head has no mean features
        __syncthreads();
    }
    }
    STAMP();
    REFRESH();
    // ================= P8: dv0 = P^T dy, dvL = w C->^T dy =================
    {
        // a packed tile (lane = query, registers = keys) -> rows 32t.. of the AT-format image in R: transposed on the matrix core
        // (X . I), two 16-byte LDS stores per tile (the 2-byte scatter form costs sixteen)
        bf16x8 idl8, idh8;
        identity_frags(idl8, idh8, r, h);
        auto image_tile = [&](int t, bf16x8 lo, bf16x8 hi) {
            f32x16 tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, idl8, zero16(), 0, 0, 0);
            tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, idh8, tr, 0, 0, 0);
            bf16x8 tl, th;
            pack_tile_bf(tl, th, tr);
            unsigned short *dst = R + (32 * t + r) * LDA + 32 * w + 8 * h;
            *(bf16x8 *)dst = tl;
            *(bf16x8 *)(dst + 16) = th;
        };
        // the Smix slab of this row block is requested in one batch (tile by tile, each p_tile() exposed its own HBM round trip: 14.7 k
        // of this phase's 66 k cycles); register pressure is low here, the tile loop is over
        u32x4 smx[NT][2];
        {
            const u32x4 *ps = slot(S_SM);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                smx[t][0] = __builtin_nontemporal_load(&ps[(2 * t) * 64]);
                smx[t][1] = __builtin_nontemporal_load(&ps[trim_idx(2 * t + 1) * 64]);
                if (ctrim && t == NT - 1) smx[t][1] = u32x4{0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u};       // trimmed chunk: Smix = -inf (padding keys)
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x16 P = unpack_tile_h(smx[t][0], smx[t][1]);
#pragma unroll
            for (int g = 0; g < 16; ++g) P[g] = __expf(P[g] - mxrow) * invl;
            if (drop.thresh) {                 // dv0 sees the dropped probabilities
#pragma unroll
                for (int g = 0; g < 16; ++g) P[g] = fa_drop_keep(drop, rowh, 32 * t + tile_row(g, h)) ? P[g] * drop.inv_keep : 0.f;
            }
            bf16x8 lo, hi;
            pack_tile_bf(lo, hi, P);
            image_tile(t, lo, hi);
        }
        __syncthreads();
        f32x16 g0[DT], gL[DT];
        gemm_rows_glob(g0, R, DYT);                    // dV0e[j in tile w][d]
        {   // C-> image: slab tiles requested together, transposed on the matrix core after the readers of the P image are done
            const u32x4 *p = slot(S_CF);
            u32x4 buf[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) { buf[t][0] = p[(2 * t) * 64]; buf[t][1] = trim_val(2 * t + 1, p[trim_idx(2 * t + 1) * 64]); }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < NT; ++t) image_tile(t, as_b8(buf[t][0]), as_b8(buf[t][1]));
        }
        __syncthreads();
        gemm_rows_glob(gL, R, DYT);                    // (C->^T dy)[j][d]
        __syncthreads();
        const IOT *v0p = (const IOT *)a.v0.ptr + b * a.v0.sb + hh * a.v0.sh;
        const IOT *vLp = (const IOT *)a.vL.ptr + b * a.vL.sb + hh * a.vL.sh;
        IOT *d0p = (IOT *)a.dv0.ptr + b * a.dv0.sb + hh * a.dv0.sh;
        IOT *dLp = (IOT *)a.dvL.ptr + b * a.dvL.sb + hh * a.dvL.sh;
        const bool same = a.dv0.ptr == a.dvL.ptr;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 32 * dt + r;                 // lanes <-> d, registers <-> keys of tile w
            float s0 = 0.f, sL = 0.f;
            if (d < DK) {
                const float f0 = vs0[d], fL = vsL[d] * wv;
                // the v0 / vL values of this lane's 16 keys are requested together from rows that exist (padded keys read row 0 and
                // are masked): guarded, each pair of loads sat in its own exec-masked branch with a full wait (32 per lane)
                float x0[16], xL[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int j = 32 * w + tile_row(g, h), jc = j < N ? j : 0;
                    x0[g] = ld_as_f32(v0p + (int64_t)jc * a.v0.sn + d);
                    xL[g] = ld_as_f32(vLp + (int64_t)jc * a.vL.sn + d);
                }
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int j = 32 * w + tile_row(g, h);
                    const float e0 = g0[dt][g], eL = gL[dt][g] * wv;
                    s0 += keep_if(j < N, e0 * x0[g]);
                    sL += keep_if(j < N, eL * xL[g]);
                    if (j < N) {
                        if (same) st_from_f32(d0p + (int64_t)j * a.dv0.sn + d, e0 * f0 + gL[dt][g] * fL);
                        else { st_from_f32(d0p + (int64_t)j * a.dv0.sn + d, e0 * f0); st_from_f32(dLp + (int64_t)j * a.dvL.sn + d, gL[dt][g] * fL); }
                    }
                }
                s0 += __shfl_xor(s0, 32, 64); sL += __shfl_xor(sL, 32, 64);
                if (h == 0) { redbuf[w * DK + d] = s0; redbuf[(NT + w) * DK + d] = sL; }
            }
        }
        __syncthreads();
        if (tid < DK) {
            float s0 = 0.f, sL = 0.f;
            for (int ww = 0; ww < NT; ++ww) { s0 += redbuf[ww * DK + tid]; sL += redbuf[(NT + ww) * DK + tid]; }
            a.dvs0_part[((int64_t)b * H + hh) * DK + tid] = s0;
            a.dvsL_part[((int64_t)b * H + hh) * DK + tid] = sL;
        }
        // dlogit_part = (1-w) sum dy * (w y_chain)
        float dl = 0.f;
        if (qok) {                                      // 16-byte reads: dy of this lane's 8-channel chunks (full precision for fp32 I/O), w * y_chain as float4 pairs
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                float dv8[8];
                ld8_as_f32(dv8, dyrow + 16 * s + 8 * h);
                const float4 y0 = *(const float4 *)&ych[(size_t)qi * DK + 16 * s + 8 * h], y1 = *(const float4 *)&ych[(size_t)qi * DK + 16 * s + 8 * h + 4];
                const float yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) dl = fmaf(dv8[j], yv[j], dl);
            }
        }
        dl = wave_sum(dl);
        if (lane == 0) misc[4 + w] = dl;
        __syncthreads();
        if (tid == 0) { float s = 0.f; for (int ww = 0; ww < NT; ++ww) s += misc[4 + ww]; a.dlogit_part[(int64_t)b * H + hh] = s * (1.f - wv); }
    }
    STAMP();
    }   // PH_A
    if constexpr (PH == PH_B) {
    // ================= PH_B: the two D-chains, one work item per chain; every D_m is exported as a packed slab =================
    REFRESH();
    if (chain_id == 1) {
        // <- chain: D'_{V-1} = dC<- = (drCl_i + dcCl_j) / (C<- + eps);  D'_{m-1}^T = A_{V-1-m} D'_m^T
        const float drl = dmean[(2 * V + 2) * NP + qi];
        bf16x8 Dp[NT][2];
        {
            bf16x8 Xp[NT][2];
            slot_ld(S_CB, Xp);
            pin_slab(Xp);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x16 cb = unpack_tile_bf(Xp[t][0], Xp[t][1]);
                f32x16 dcl = zero16();                 // dense head: per-edge gradient of the log C<- feature (launch A)
                if constexpr (HEAD == 1) { const u32x4 *pc = slot(X_CL(V)); dcl = unpack_tile_bf(as_b8(pc[(2 * t) * 64]), as_b8(pc[(2 * t + 1) * 64])); }
                f32x16 d;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int j = 32 * t + tile_row(g, h);
                    d[g] = keep_if(j < N && qok, (dcl[g] + drl + dmean[(2 * V + 3) * NP + j]) * __builtin_amdgcn_rcpf(cb[g] + EPSC));
                }
                pack_tile_bf(Dp[t][0], Dp[t][1], d);
            }
        }
        for (int m = V - 1; m >= 1; --m) {
            slot_st(X_DL(V) + m, Dp);
            a_image(R, V - 1 - m, true);               // A_av (rows = queries)
            gemm_packed(Dp, R);                        // D'_{m-1}^T = A_av D'_m^T
        }
        slot_st(X_DL(V), Dp);
    } else {
        // -> chain: D_{V-1} = dC-> = (G_chain dSmix + drCr_i + dcCr_j) / (C-> + eps) + w dy vL^T;  D_{v-1}^T = A_v D_v^T
        bf16x8 Dp[NT][2];
        {
            const float drr = dmean[(2 * V) * NP + qi];
            bf16x8 Cp[NT][2], C3[NT][2];
            bf16x8 dyf2[KS];
            make_frag(dyf2, dyrow, nullptr);
            slot_ld(S_CF, Cp);
            slot_ld(X_C3, C3);
            pin_slab(Cp); pin_slab(C3);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x16 cf = unpack_tile_bf(Cp[t][0], Cp[t][1]);
                const f32x16 c3 = unpack_tile_bf(C3[t][0], C3[t][1]);
                f32x16 dyv = zero16();                           // (dy vL^T)^T tile: vL rows from LDS (staged over R by P0)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 af = *(const bf16x8 *)&R[(32 * t + r) * LDK + 16 * s + 8 * h];
                    dyv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dyf2[s], dyv, 0, 0, 0);
                }
                f32x16 d;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int j = 32 * t + tile_row(g, h);
                    d[g] = keep_if(j < N && qok, (c3[g] + drr + dmean[(2 * V + 1) * NP + j]) * __builtin_amdgcn_rcpf(cf[g] + EPSC) + wv * dyv[g]);
                }
                pack_tile_bf(Dp[t][0], Dp[t][1], d);
            }
        }
        for (int v = V - 1; v >= 1; --v) {
            STAMP2();
            slot_st(X_DR(V) + v, Dp);
            a_image(R, v, true);
            gemm_packed(Dp, R);                        // D_{v-1}^T = A_v D_v^T
        }
        STAMP2();
        slot_st(X_DR(V), Dp);
    }
    }   // PH_B
    if constexpr (PH == PH_C) {
    // ================= PH_C: per view dA_v -> dS_v -> dQe_v, dK_v; dq / dk summed over the views in registers =================
    REFRESH();
    {
        f32x16 dqa[DT], dka[DT];           // dq^T [d, my query] and dk [key of tile w, d], summed over the views
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dqa[dt] = zero16(); dka[dt] = zero16(); }
        for (int v = V - 1; v >= 0; --v) {
            REFRESH();
            // a parked slab (X layout: lane = query, registers = keys) -> LDS image in AT format (rows = keys, k-permuted query columns).
            // The transpose is taken on the matrix core: X . I comes back with lane = key, registers = queries, and its packed halves
            // are 16-byte chunks of the image rows -- two ds_write_b128 per tile instead of sixteen 2-byte stores, and no slab array
            // that a rolled store loop would demote to scratch memory.
            auto stage_image = [&](int sid) {
                const u32x4 *p = slot(sid);
                bf16x8 idl, idh;              // B fragments of the 32 x 32 identity in the accumulator's k order
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    idl[e] = (short)(r == tile_row(e, h) ? 0x3f80 : 0);          // bf16(1.0)
                    idh[e] = (short)(r == 16 + tile_row(e, h) ? 0x3f80 : 0);
                }
                u32x4 buf[NT][2];
#pragma unroll
                for (int t = 0; t < NT; ++t) { buf[t][0] = __builtin_nontemporal_load(&p[(2 * t) * 64]); buf[t][1] = trim_val(2 * t + 1, __builtin_nontemporal_load(&p[trim_idx(2 * t + 1) * 64])); }
                lds_barrier();                     // previous readers of R are done
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x16 tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b8(buf[t][0]), idl, zero16(), 0, 0, 0);
                    tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b8(buf[t][1]), idh, tr, 0, 0, 0);
                    bf16x8 lo, hi;
                    pack_tile_bf(lo, hi, tr);     // exact: every entry is one bf16 value times 1.0
                    // lane (r, h) now holds key 32t + r, queries 32w + {4h + 0..3, 8 + 4h + 0..3} (lo) and 16 + the same (hi):
                    // chunks 4w + h and 4w + 2 + h of image row 32t + r
                    unsigned short *dst = R + (32 * t + r) * LDA + 32 * w + 8 * h;
                    *(bf16x8 *)dst = lo;
                    *(bf16x8 *)(dst + 16) = hi;
                }
            };
            // ---- dA_v^T slab (rows = keys, lanes = my queries as A_v's row index), kept as packed bf16 tiles
            bf16x8 dAp[NT][2];
            const int mp = V - 1 - v;
            if (v >= 1) {
                bf16x8 Bf[NT][2];
                stage_image(X_DR(V) + v);                                // D_v of the -> chain (PH_B)
                load_rows(Bf, Tg + (size_t)(v - 1) * NP * LDA);          // in flight across the barrier
                lds_barrier();
                gemm_stream(dAp, R, Bf, false);
            } else {
                slot_ld(X_DR(V), dAp);
            }
            {
                __builtin_amdgcn_sched_barrier(0);
                if (mp >= 1) {
                    bf16x8 Bf[NT][2];
                    stage_image(X_DL(V) + mp);                           // D'_mp of the <- chain
                    load_rows(Bf, Ug + (size_t)(mp - 1) * NP * LDA);
                    lds_barrier();
                    gemm_stream(dAp, R, Bf, true);
                } else {
                    bf16x8 Dl[NT][2];
                    slot_ld(X_DL(V), Dl);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        f32x16 x = unpack_tile_bf(dAp[t][0], dAp[t][1]);
                        x += unpack_tile_bf(Dl[t][0], Dl[t][1]);
                        pack_tile_bf(dAp[t][0], dAp[t][1], x);
                    }
                }
            }
            if (v == V - 1) STAMP();
            REFRESH();
            // ---- dA_v^T is parked in LDS for the two rolled passes below (indexed by a runtime tile number as a register array it
            //      is demoted to private memory: 3 GB of scratch traffic per launch).  Tile t of wave w goes where the same wave
            //      later writes tile t of the dS^T image -- rows 32t.., columns 32w.. of R, 2 KiB either way -- so the dS pass
            //      replaces each tile in place and no other wave touches the slot before the barrier in front of the dK GEMM.
            const float cv = cstats[v * NP + qi];
            bf16x8 qe[KS];
            make_frag(qe, qrow, sqk2 + v * DK);           // base-2 scores: A = 2^(S' - c)
            unsigned short *park = R + r * LDA + 32 * w + 16 * h;
            lds_barrier();                                 // every wave is done reading the D image of the last GEMM
#pragma unroll
            for (int t = 0; t < NT; ++t) { *(bf16x8 *)(park + 32 * t * LDA) = dAp[t][0]; *(bf16x8 *)(park + 32 * t * LDA + 8) = dAp[t][1]; }
            // ---- softmax-backward row dot  sum_j A_v dA_v  (rolled pass over key tiles)
            float dot = 0.f;
#pragma nounroll
            for (int t = 0; t < NT; ++t) {
                const f32x16 A = a_tile(qe, t, cv);
                const bf16x8 dl_ = *(const bf16x8 *)(park + 32 * t * LDA), dh_ = *(const bf16x8 *)(park + 32 * t * LDA + 8);
                const f32x16 dA = unpack_tile_bf(dl_, dh_);
#pragma unroll
                for (int g = 0; g < 16; ++g) dot = fmaf(A[g], dA[g], dot);
            }
            dot += __shfl_xor(dot, 32, 64);
            // ---- softmax backward + direct + mean terms -> dS_v^T, streamed per key tile: each dS tile goes straight
            //      into the LDS image (A operand of dK) and into dQe_v^T += K^T[:, tile] dS^T[tile, :] (4 MFMAs)
            f32x16 dq[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = zero16();
            {
                const float drs = dmean[v * NP + qi];
                // software prefetch: the parked direct-gradient tile and the K^T fragments of tile t+1 are requested
                // while tile t is computed (a rolled loop exposes one full L2/HBM round trip per iteration otherwise)
                const u32x4 *pd = slot(X_DIR + v);
                const unsigned short *ktb = KT + r * LDA + 8 * h;
                u32x4 nd0 = __builtin_nontemporal_load(&pd[0]), nd1 = trim_val(1, __builtin_nontemporal_load(&pd[trim_idx(1) * 64]));
                bf16x8 nk[DT][2];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) { nk[dt][0] = *(const bf16x8 *)&ktb[(32 * dt) * LDA]; nk[dt][1] = *(const bf16x8 *)&ktb[(32 * dt) * LDA + 16]; }
#pragma nounroll
                for (int t = 0; t < NT; ++t) {
                    const u32x4 d0 = nd0, d1 = nd1;
                    bf16x8 kf[DT][2];
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) { kf[dt][0] = nk[dt][0]; kf[dt][1] = nk[dt][1]; }
                    const bf16x8 dl_ = *(const bf16x8 *)(park + 32 * t * LDA), dh_ = *(const bf16x8 *)(park + 32 * t * LDA + 8);
                    {   // unconditional (the last iteration re-requests its own tile): a conditional prefetch makes the number of
                        // outstanding loads unknown at the join and every later s_waitcnt in the iteration becomes vmcnt(0)
                        const int tn = t + 1 < NT ? t + 1 : t;
                        nd0 = __builtin_nontemporal_load(&pd[(2 * tn) * 64]); nd1 = trim_val(2 * tn + 1, __builtin_nontemporal_load(&pd[trim_idx(2 * tn + 1) * 64]));
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            nk[dt][0] = *(const bf16x8 *)&ktb[(32 * dt) * LDA + 32 * tn];
                            nk[dt][1] = *(const bf16x8 *)&ktb[(32 * dt) * LDA + 32 * tn + 16];
                        }
                    }
                    const f32x16 A = a_tile(qe, t, cv);
                    f32x16 dir = unpack_tile_bf(as_b8(d0), as_b8(d1));
                    if constexpr (HEAD == 1) {
                        // dense head: the gradient of the S_v^T feature channel at edge (j, i) belongs to dS_v(i, j).  Launch A left it in
                        // the slab of the wave that owns query block t, at its key tile w: that tile, transposed on the matrix core
                        // (X . I: lane = my query, registers = keys of tile t), is added here
                        const u32x4 *pt = (const u32x4 *)(xf + W.xSlots + ((size_t)(X_DT(V) + v - X_C3) * NT + t) * Cfg::SLOT) + lane;
                        bf16x8 idl8, idh8;
                        identity_frags(idl8, idh8, r, h);
                        f32x16 tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b8(pt[(2 * w) * 64]), idl8, zero16(), 0, 0, 0);
                        tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b8(pt[(2 * w + 1) * 64]), idh8, tr, 0, 0, 0);
                        dir += tr;
                    }
                    const f32x16 dA = unpack_tile_bf(dl_, dh_);
                    f32x16 dS;
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int j = 32 * t + tile_row(g, h);
                        dS[g] = A[g] * (dA[g] - dot) + dir[g] + drs + dmean[(V + v) * NP + j];
                    }
                    if (32 * t + 32 > N) {            // only the tile that holds padding keys is masked (wave-uniform)
#pragma unroll
                        for (int g = 0; g < 16; ++g) dS[g] = keep_if(32 * t + tile_row(g, h) < N, dS[g]);
                    }
                    bf16x8 lo, hi;
                    pack_tile_bf(lo, hi, dS);
                    store_i_tile(R, t, lo, hi);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[dt][0], lo, dq[dt], 0, 0, 0);
                        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[dt][1], hi, dq[dt], 0, 0, 0);
                    }
                }
            }
            if (v == V - 1) STAMP();
            REFRESH();
            // ---- dQe_v^T = K^T dS^T ; dq += sqk_v * dQe_v ; dsqk_v = sum_i q * dQe_v
            {
                // q values of this lane's (query, d) pairs: requested unconditionally and all at once per d-tile (a row that exists is
                // read for padded queries / d >= dk, the product is masked) -- guarded loads became one branch + full wait per quad
                const IOT *qsafe = qok ? qrow : (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    float c[16];
                    float qv[4][4];
                    f32x16 dqs;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int d0 = 32 * dt + 8 * g4 + 4 * h, d0c = d0 < DK ? d0 : 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) qv[g4][e] = ld_as_f32(qsafe + d0c + e);
                    }
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int d0 = 32 * dt + 8 * g4 + 4 * h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int g = 4 * g4 + e;
                            const float sc = d0 < DK ? sqk[v * DK + d0 + e] : 0.f;
                            dqs[g] = sc * dq[dt][g];
                            c[g] = keep_if(qok && d0 < DK, qv[g4][e] * dq[dt][g]);
                        }
                    }
                    dqa[dt] += dqs;
                    // reduce over the 32 queries of this half (butterfly), lane r even holds register r>>1
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const int n = 8 >> st;
                        const bool up = (r >> (4 - st)) & 1;
#pragma unroll
                        for (int k = 0; k < n; ++k) {
                            const float keep = up ? c[k + n] : c[k], send = up ? c[k] : c[k + n];
                            c[k] = keep + __shfl_xor(send, 16 >> st, 64);
                        }
                    }
                    c[0] += __shfl_xor(c[0], 1, 64);
                    const int d = 32 * dt + tile_row(r >> 1, h);
                    if ((r & 1) == 0 && d < DK) redbuf[w * DK + d] = c[0];
                }
            }
            if (v == V - 1) STAMP();
            REFRESH();
            // ---- dK += sqk_v * (dS^T Q) through the LDS image written above
            lds_barrier();
            if (tid < DK) {
                float s = 0.f;
                for (int ww = 0; ww < NT; ++ww) s += redbuf[ww * DK + tid];
                a.dsqk_part[(((int64_t)b * V + v) * H + hh) * DK + tid] = s;
            }
            {
                f32x16 dk[DT];
                gemm_rows_glob(dk, R, QT);             // (dS^T q)[j in tile w][d]
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int d = 32 * dt + r;
                    const float sc = d < DK ? sqk[v * DK + d] : 0.f;
                    dka[dt] += dk[dt] * sc;
                }
            }
            if (v == V - 1) STAMP();
            REFRESH();
        }
        // ================= P11: write dq, dk =================
        REFRESH();
        {
            IOT *dqp = (IOT *)a.dq.ptr + b * a.dq.sb + hh * a.dq.sh + (int64_t)qi * a.dq.sn;
            if (qok) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int d0 = 32 * dt + 8 * g4 + 4 * h;
                        if (d0 < DK) store4<IOT>(dqp + d0, dqa[dt][4 * g4], dqa[dt][4 * g4 + 1], dqa[dt][4 * g4 + 2], dqa[dt][4 * g4 + 3]);
                    }
            }
            IOT *dkp = (IOT *)a.dk_.ptr + b * a.dk_.sb + hh * a.dk_.sh;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = 32 * dt + r;
                if (d < DK) {
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int j = 32 * w + tile_row(g, h);
                        if (j < N) st_from_f32(dkp + (int64_t)j * a.dk_.sn + d, dka[dt][g]);
                    }
                }
            }
        }
    }
    }   // PH_C
    STAMP();
    __syncthreads();      // LDS / scratch reuse by the next work item
    }   // persistent loop
    STAMP();
}

// ------------------------------------------------------------------ host side (see edgewise_fused.hip)
#define MOPK_CAT_(a, b, c, d) a##b##c##d
#define MOPK_CAT(a, b, c, d) MOPK_CAT_(a, b, c, d)
#if MOPK_INST_NT != 0
// persistent workgroups: as many per CU as its registers (two waves per SIMD at 256 VGPRs) and LDS hold -- one for N > 128, two for
// 64 < N <= 96 (the reference's CIFAR sequence length, N = 65), four / eight below
static int bwd_grid(const MopkEdgewiseArgs *a, int items_per_bh) {
    constexpr int NT = MOPK_INST_NT;
    const int lds = BwdCfg<MOPK_INST_NT, MOPK_INST_DK>::lds_bytes(a->V);
    int per_cu = NT <= 3 ? 8 / NT : 1;            // matches the __launch_bounds__ occupancy hint of the kernels
    if (lds * per_cu > 160 * 1024) per_cu = 160 * 1024 / lds;
    if (per_cu < 1) per_cu = 1;
    const int n = a->B * a->H * items_per_bh, cap = 256 * per_cu;
    return n < cap ? n : cap;
}
static size_t a256h(size_t x) { return (x + 255) & ~(size_t)255; }
// workspace = [per-workgroup scratch x grid | per-(b,h) hand-off x B*H | (save_for_backward == 0 only) full `saved` record + y scratch
// for the forward re-run in export mode]
static bool bwd_dense(const MopkEdgewiseArgs *a) { return a->ext && a->ext->gate_mode == 1; }
static size_t bwd_core_bytes(const MopkEdgewiseArgs *a) {
    return a256h(BwdCfg<MOPK_INST_NT, MOPK_INST_DK>::total_bytes(a->V, bwd_grid(a, 2), a->B * a->H, bwd_dense(a)));   // scratch for the widest grid (PH_B)
}
static size_t bwd_full_saved_bytes(const MopkEdgewiseArgs *a) {
    return a256h(fused_saved_layout<MOPK_INST_NT, MOPK_INST_DK>(a->N, a->V, true).stride * (size_t)a->B * a->H + 256);
}
size_t MOPK_CAT(ew_fused_bwd_ws_nt, MOPK_INST_NT, _dk, MOPK_INST_DK)(const MopkEdgewiseArgs *a) {
    size_t n = bwd_core_bytes(a) + 256;
    if (!a->save_for_backward) n += bwd_full_saved_bytes(a) + a256h((size_t)a->B * a->H * a->N * a->dk * 4);
    return n;
}
void ew_fused_dw_reduce(const MopkEdgewiseArgs *a, const BwdWs &W, int nwg, hipStream_t st);
void ew_fused_dense_dw_reduce(const MopkEdgewiseArgs *a, const BwdWs &W, int nwg, hipStream_t st);
int MOPK_CAT(ew_fused_fwd_nt, MOPK_INST_NT, _dk, MOPK_INST_DK)(const MopkEdgewiseArgs *a, hipStream_t st);
int MOPK_CAT(ew_fused_bwd_nt, MOPK_INST_NT, _dk, MOPK_INST_DK)(const MopkEdgewiseArgs *a_in, hipStream_t st) {
    constexpr int NT = MOPK_INST_NT, DK = MOPK_INST_DK;
    using Cfg = BwdCfg<NT, DK>;
    const int lds = Cfg::lds_bytes(a_in->V);
    const int n_extra = a_in->ext && a_in->ext->gate_mode == 0 ? a_in->ext->n_extra : 0;
    if (lds > 160 * 1024 || 2 * a_in->V + 2 + n_extra > WST - 1) return MOPK_ERR_UNSUPPORTED;
    if (n_extra && (!a_in->ext->row_extra || !a_in->ext->col_extra || !a_in->ext->d_row_extra || !a_in->ext->d_col_extra)) return MOPK_ERR_BAD_ARG;
    MopkEdgewiseArgs args = *a_in;
    if (!a_in->save_for_backward) {
        // small `saved`: rebuild the full record (chain + mix state) by running the forward in export mode into the workspace
        unsigned char *rb = (unsigned char *)a_in->workspace + bwd_core_bytes(a_in);
        MopkEdgewiseArgs f = *a_in;
        f.save_for_backward = 1;
        f.saved = rb;
        f.y.ptr = rb + bwd_full_saved_bytes(a_in);
        f.y.sb = (int64_t)a_in->H * a_in->N * a_in->dk; f.y.sh = (int64_t)a_in->N * a_in->dk; f.y.sn = a_in->dk;
        const int rc = MOPK_CAT(ew_fused_fwd_nt, MOPK_INST_NT, _dk, MOPK_INST_DK)(&f, st);
        if (rc != MOPK_OK) return rc;
        args.saved = rb;
        args.save_for_backward = 1;
    }
    const MopkEdgewiseArgs *a = &args;
    const int nwgA = bwd_grid(a, 1), nwgB = bwd_grid(a, 2);
    const bool dense = bwd_dense(a);
    const BwdWs W = Cfg::carve(a->workspace, a->V, nwgB, a->B * a->H, dense);     // nwgB >= nwgA
    FusedDenseW dw{nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (dense) { dw.W1 = a->ext->W1; dw.b1 = a->ext->b1; dw.W2 = a->ext->W2; dw.b2 = a->ext->b2; }
    else if (n_extra) { dw.E = n_extra; dw.rowx = a->ext->row_extra; dw.colx = a->ext->col_extra; dw.drowx = a->ext->d_row_extra; dw.dcolx = a->ext->d_col_extra; }
    const dim3 block(NT * 64);
#define MOPK_LAUNCH_H(IOT_, PH_, GRID_, HEAD_) do {                                                               \
        auto kfn = ew_fused_bwd_kernel<NT, DK, IOT_, PH_, HEAD_>;                                                 \
        static int lds_set = 0;      /* per instantiation and per process: the attribute is sticky (and may not be set during stream capture) */ \
        if (lds_set < lds) { if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MOPK_ERR_LAUNCH; lds_set = lds; } \
        hipLaunchKernelGGL(kfn, dim3(GRID_), block, lds, st, *a, W, dw);                                          \
    } while (0)
#define MOPK_LAUNCH(IOT_, PH_, GRID_) do { if (dense) MOPK_LAUNCH_H(IOT_, PH_, GRID_, 1); else MOPK_LAUNCH_H(IOT_, PH_, GRID_, 0); } while (0)
    if (a->io_dtype == MOPK_BF16) MOPK_LAUNCH(unsigned short, PH_A, nwgA); else MOPK_LAUNCH(float, PH_A, nwgA);
    MOPK_CHECK_LAUNCH();
    {   // launch B: the two D-chains
        if (a->io_dtype == MOPK_BF16) MOPK_LAUNCH(unsigned short, PH_B, nwgB); else MOPK_LAUNCH(float, PH_B, nwgB);
    }
    if (a->io_dtype == MOPK_BF16) MOPK_LAUNCH(unsigned short, PH_C, nwgA); else MOPK_LAUNCH(float, PH_C, nwgA);
#undef MOPK_LAUNCH
#undef MOPK_LAUNCH_H
    MOPK_CHECK_LAUNCH();
    if (dense) ew_fused_dense_dw_reduce(a, W, nwgA, st); else ew_fused_dw_reduce(a, W, nwgA, st);
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}
#else
// sum the per-workgroup dW partials: out[idx] = sum_bh dwp[bh][idx]   (deterministic order)
__global__ void ew_fused_dw_reduce_kernel(MopkEdgewiseArgs a, BwdWs W, int nwg, int C) {
    // one block per output; 256 threads stride over the workgroups, fixed-shape tree reduction
    __shared__ float red[256];
    const int nO = 4 * a.r;
    const int idx = blockIdx.x;
    float s = 0.f;
    for (int g = threadIdx.x; g < nwg; g += 256) s += ((const float *)(W.base + (size_t)g * W.stride + W.oDW))[idx];
    red[threadIdx.x] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x != 0) return;
    s = red[0];
    const int side = idx / (nO * (C + 1)), rem = idx % (nO * (C + 1));
    const int o = rem / (C + 1), c = rem % (C + 1);
    if (c < C) (side ? a.dWc : a.dWr)[o * C + c] = s; else (side ? a.dbc : a.dbr)[o] = s;
}

void ew_fused_dw_reduce(const MopkEdgewiseArgs *a, const BwdWs &W, int nwg, hipStream_t st) {
    const int C = 2 * a->V + 2 + (a->ext && a->ext->gate_mode == 0 ? a->ext->n_extra : 0);      // input channels of the head
    const int nout = 2 * 4 * a->r * (C + 1);
    hipLaunchKernelGGL(ew_fused_dw_reduce_kernel, dim3(nout), dim3(256), 0, st, *a, W, nwg, C);
}
// dense head: per-workgroup partials [[c][k] table: dW1[k][c] for c < C, row 15 = db1[k] | dW2^T [k][m] | db2[m]] -> conv1 / conv2 gradients
struct DenseGradOut { float *dW1, *db1, *dW2, *db2; };
__global__ void ew_fused_dense_dw_reduce_kernel(BwdWs W, int nwg, int C, DenseGradOut o) {
    __shared__ float red[256];
    const int idx = blockIdx.x;                       // 0 .. 323
    float s = 0.f;
    for (int g = threadIdx.x; g < nwg; g += 256) s += ((const float *)(W.base + (size_t)g * W.stride + W.oDW))[idx];
    red[threadIdx.x] = s; __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x != 0) return;
    s = red[0];
    if (idx < 256) { const int c = idx / 16, k = idx % 16; if (c < C) o.dW1[k * C + c] = s; else if (c == 15) o.db1[k] = s; }
    else if (idx < 320) { const int k = (idx - 256) / 4, m = (idx - 256) % 4; o.dW2[m * 16 + k] = s; }
    else o.db2[idx - 320] = s;
}
void ew_fused_dense_dw_reduce(const MopkEdgewiseArgs *a, const BwdWs &W, int nwg, hipStream_t st) {
    const DenseGradOut o{a->ext->dW1, a->ext->db1, a->ext->dW2, a->ext->db2};
    hipLaunchKernelGGL(ew_fused_dense_dw_reduce_kernel, dim3(324), dim3(256), 0, st, W, nwg, 2 * a->V + 2, o);
}
int ew_fused_fwd_supported(const MopkEdgewiseArgs *a);
#define MOPK_DECL(NT_, DK_) int ew_fused_bwd_nt##NT_##_dk##DK_(const MopkEdgewiseArgs *a, hipStream_t st); \
                            size_t ew_fused_bwd_ws_nt##NT_##_dk##DK_(const MopkEdgewiseArgs *a);
MOPK_DECL(1, 16) MOPK_DECL(1, 32) MOPK_DECL(1, 64) MOPK_DECL(2, 16) MOPK_DECL(2, 32) MOPK_DECL(2, 64)
MOPK_DECL(3, 16) MOPK_DECL(3, 32) MOPK_DECL(3, 64) MOPK_DECL(4, 16) MOPK_DECL(4, 32) MOPK_DECL(4, 64) MOPK_DECL(5, 16) MOPK_DECL(5, 32) MOPK_DECL(5, 64) MOPK_DECL(6, 16) MOPK_DECL(6, 32) MOPK_DECL(6, 64) MOPK_DECL(7, 16) MOPK_DECL(7, 32) MOPK_DECL(7, 64)
#undef MOPK_DECL
static int pick_nt_b(int N) { return N <= 32 ? 1 : N <= 64 ? 2 : N <= 96 ? 3 : N <= 128 ? 4 : N <= 160 ? 5 : N <= 192 ? 6 : N <= 224 ? 7 : 0; }
#define MOPK_BWD_DISPATCH(PFX, ...)                                                   \
    switch (pick_nt_b(a->N)) {                                                       \
        case 1: switch (a->dk) { case 16: return PFX##nt1_dk16(__VA_ARGS__); case 32: return PFX##nt1_dk32(__VA_ARGS__); default: return PFX##nt1_dk64(__VA_ARGS__); } \
        case 2: switch (a->dk) { case 16: return PFX##nt2_dk16(__VA_ARGS__); case 32: return PFX##nt2_dk32(__VA_ARGS__); default: return PFX##nt2_dk64(__VA_ARGS__); } \
        case 3: switch (a->dk) { case 16: return PFX##nt3_dk16(__VA_ARGS__); case 32: return PFX##nt3_dk32(__VA_ARGS__); default: return PFX##nt3_dk64(__VA_ARGS__); } \
        case 4: switch (a->dk) { case 16: return PFX##nt4_dk16(__VA_ARGS__); case 32: return PFX##nt4_dk32(__VA_ARGS__); default: return PFX##nt4_dk64(__VA_ARGS__); } \
        case 5: switch (a->dk) { case 16: return PFX##nt5_dk16(__VA_ARGS__); case 32: return PFX##nt5_dk32(__VA_ARGS__); default: return PFX##nt5_dk64(__VA_ARGS__); } \
        case 6: switch (a->dk) { case 16: return PFX##nt6_dk16(__VA_ARGS__); case 32: return PFX##nt6_dk32(__VA_ARGS__); default: return PFX##nt6_dk64(__VA_ARGS__); } \
        default: switch (a->dk) { case 16: return PFX##nt7_dk16(__VA_ARGS__); case 32: return PFX##nt7_dk32(__VA_ARGS__); default: return PFX##nt7_dk64(__VA_ARGS__); } \
    }
template <int NT, int DK> static int lds_bwd(int V) { return BwdCfg<NT, DK>::lds_bytes(V); }
int ew_fused_bwd_lds_bytes(int nt, int dk, int V) {
#define MOPK_L(NT_) (dk == 16 ? lds_bwd<NT_, 16>(V) : dk == 32 ? lds_bwd<NT_, 32>(V) : lds_bwd<NT_, 64>(V))
    switch (nt) { case 1: return MOPK_L(1); case 2: return MOPK_L(2); case 3: return MOPK_L(3); case 4: return MOPK_L(4); case 5: return MOPK_L(5); case 6: return MOPK_L(6); case 7: return MOPK_L(7); default: return 1 << 30; }
#undef MOPK_L
}
int ew_fused_bwd_supported(const MopkEdgewiseArgs *a) {
    if (!ew_fused_fwd_supported(a)) return 0;
    if (a->ext && a->ext->gate_mode != 0) {                   // dense head (no 3x3): one 16-slot row holds dW1[k][0..C-1] and db1[k]
        if (a->V > 6 || !a->ext->dW1 || !a->ext->db1 || !a->ext->dW2 || !a->ext->db2) return 0;
    }
    if (a->dq.sv != 0 || a->dk_.sv != 0) return 0;
    // the backward reads dy and writes dq / dk / dv with 16-byte vectors, like the forward does q, k, v, y
    const int64_t al = 16 / (a->io_dtype == MOPK_BF16 ? 2 : 4);
    auto ok = [&](const void *p, int64_t sb, int64_t sh, int64_t sn) {
        return p != nullptr && ((uintptr_t)p & 15) == 0 && sb % al == 0 && sh % al == 0 && sn % al == 0;
    };
    if (!ok(a->dy.ptr, a->dy.sb, a->dy.sh, a->dy.sn) || !ok(a->dq.ptr, a->dq.sb, a->dq.sh, a->dq.sn) ||
        !ok(a->dk_.ptr, a->dk_.sb, a->dk_.sh, a->dk_.sn) || !ok(a->dv0.ptr, a->dv0.sb, a->dv0.sh, a->dv0.sn) ||
        !ok(a->dvL.ptr, a->dvL.sb, a->dvL.sh, a->dvL.sn)) return 0;
    return 1;
}
size_t ew_fused_bwd_ws_bytes(const MopkEdgewiseArgs *a) {
    MOPK_BWD_DISPATCH(ew_fused_bwd_ws_, a)
}
int ew_fused_bwd(const MopkEdgewiseArgs *a, hipStream_t st) {
    if (!ew_fused_bwd_supported(a)) return MOPK_ERR_UNSUPPORTED;
    MOPK_BWD_DISPATCH(ew_fused_bwd_, a, st)
}
#endif

}  // namespace mopk
