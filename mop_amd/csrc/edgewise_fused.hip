// EdgewiseMSA low-rank core -- fused gfx950 forward kernel (bf16 MFMA, fp32 accumulate).
//
// One workgroup per (batch, head); NT = ceil(N/32) waves, wave w owns queries
// [32w, 32w+32).  Every N x N map lives on-chip only:
//
//   * "X layout": a wave holds the TRANSPOSED row block M^T[:, I] of an N x N matrix M as
//     NT accumulator tiles of v_mfma_f32_32x32x16_bf16 (lane = query i in I, registers =
//     key j).  Row softmax, row means, log, the score mix and the final softmax are then
//     per-lane register loops (+ one lane-half exchange); no LDS round trip.
//   * chain products (reference attention_variants.py:508-515) run transposed:
//         (T A_m)^T[:, I] = A_m^T . T^T[:, I]
//     the previous product is the B operand straight from its accumulator registers
//     (cdna guide: "accumulator tile as the next MFMA's operand"), A_m^T is staged once per
//     step in LDS (k-permuted columns so one ds_read_b128 matches the accumulator k order).
//   * C<- is reduced to its row/col log-means and dropped; C-> stays in registers for the
//     mix, the transport term y_chain = A_0(A_1(..A_{V-1} v_L)) is evaluated as C-> v_L.
//   * gate logits Z_g = a_g^T b_g (rank r <= 4) are ONE MFMA per (tile, gate) with hi/lo bf16
//     splits of a and b packed into the K=16 slots, i.e. ~fp32-accurate.
//
// HBM traffic per (b,h): q,k,v in + y out (~100 KB) for 0.9 GFLOP -> MFMA/VALU bound.
#include <stdlib.h>

#include "fused_common.h"

#ifdef MOPK_WHATIF_NOBAR       // MOPK_WHATIF_*: timing experiments only (results are wrong); built with tools/build_variant.py, read with tools/stamps_fwd.py
#define __syncthreads() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
#ifndef MOPK_PF
#define MOPK_PF 4
#endif
namespace mopk {

// workgroup barrier for LDS hand-offs: waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// make every barrier wait for the training exports' global stores (nothing in this kernel reads global data another wave wrote)
#ifndef MOPK_WHATIF_NOBAR
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

// HEAD: 0 = low-rank gate head (row / col means, rank-r outer products); 1 = dense gate head without the 3x3 convolution
//       (reference :250-272, :312-318: per edge sigmoid(W2 gelu_tanh(W1 f + b1) + b2) on f = [S_v, S_v^T, Cr, Cl]) evaluated inside
//       the mix tile loop: the score tiles of both orientations are recomputed per register quarter (40 MFMAs -- the matrix pipe is
//       idle here), the 16 hidden units of four edges per lane live in registers, C<- comes back from this wave's own export.
template <int NT, int DK, typename IOT, bool SAVE, int HEAD = 0>
__global__ void __launch_bounds__(NT * 64, NT <= 3 ? 2 : 1) ew_fused_fwd_kernel(MopkEdgewiseArgs a, FusedDenseW dw) {
    using Cfg = FusedCfg<NT, DK>;
    constexpr int NP = Cfg::NP, LDA = Cfg::LDA, LDK = Cfg::LDK, KS = Cfg::KS, DT = Cfg::DT, DP = Cfg::DP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short *AT = (unsigned short *)smem;                   // R region
    unsigned short *ring = AT;                                     // [RING][32][LDA]  parts of A_m^T (rows = keys, k-permuted query columns)
    unsigned short *Qsm = AT + Cfg::RING * Cfg::PART;              // [NP][LDK]   q rows (chains only)
    unsigned short *VT0 = AT;                                      // [DP][LDA]   (aliases the ring / q rows after the chains)
    unsigned short *VTL = AT + DP * LDA;                           // [DP][LDA]
    unsigned short *bT = AT + 2 * DP * LDA;                        // [4][NP][BTS]
    unsigned short *Ksm = (unsigned short *)(smem + Cfg::R_BYTES); // [NP][LDK]
    float *fs = (float *)(smem + Cfg::R_BYTES + Cfg::K_BYTES);
    float *sqk = fs, *sqk2 = sqk + 8 * DK, *qbar = sqk2 + 8 * DK, *kbar = qbar + DK, *vs0 = kbar + DK, *vsL = vs0 + DK;   // sqk2 = sqk * log2(e)
    float *rCr = vsL + DK, *rCl = rCr + NP, *cCr = rCl + NP, *cCl = cCr + NP;
    float *colpart = cCl + NP;                                     // [NT][NP]
    float *rS = colpart + NT * NP, *cS = rS + a.V * NP;            // [V][NP]
    float *cst = rS;                                               // [V][NP] NEGATED softmax constants -c_v[i], c = log2 sum_j 2^(S'_v[i,j]); aliases rS (dead until the gate phase)
    float *wsig = cS + a.V * NP;

    const int tid = threadIdx.x, w = tid >> 6;
    int lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int N = a.N, V = a.V, H = a.H, R = a.r;
    const int b = blockIdx.x / H, hh = blockIdx.x % H;
    int qi = 32 * w + r;                       // this lane's query
    bool qok = qi < N;
    // REFRESH(): opaque lane id per phase so lane-derived addresses are recomputed instead of hoisted + spilled
#define REFRESH() do { asm volatile("" : "+v"(lane)); r = lane & 31; h = lane >> 5; qi = 32 * w + r; qok = qi < N; } while (0)
    const float invN = 1.f / (float)N;
#ifdef MOPK_STAMPS
    unsigned long long *stamps = (unsigned long long *)a.workspace;      // 256-byte forward workspace: up to 32 stamps of workgroup 0
    int stamp_i = 0;
#define FSTAMP() do { if (blockIdx.x == 0 && tid == 0 && stamp_i < 32) stamps[stamp_i] = __builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#else
#define FSTAMP() do { } while (0)
#endif
#ifdef MOPK_STAMPS3
    // timeline of ONE part (<- chain, step 2, tile 3) for wave 0 (slots 0-15) and its SIMD partner wave 4 (slots 16-31)
#define FSTAMP3(i_) do { if (!forward && m == 2 && to == 3) { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && (tid == 0 || tid == 256)) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ((unsigned long long *)cCr)[(tid ? 16 : 0) + (i_)] = t_; } __builtin_amdgcn_sched_barrier(0); \
        if ((i_) == 11 && blockIdx.x == 0 && tid < 32) ((unsigned long long *)a.workspace)[tid] = ((unsigned long long *)cCr)[tid]; } } while (0)
#else
#define FSTAMP3(i_) do { } while (0)
#endif
#ifdef MOPK_STAMPS2
#define FSTAMP2(c_) do { if (c_) FSTAMP(); } while (0)
#else
#define FSTAMP2(c_) do { } while (0)
#endif
    FSTAMP();

    // ---------------- P0: stage K, q fragments, scales ----------------
    {
        const IOT *kp = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh;
        const IOT *qp0 = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh;
        constexpr int CH = DK / 8;
        for (int c = tid; c < NP * CH; c += NT * 64) {
            const int j = c / CH, dc = c % CH;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0}, u = v;
            if (j < N) { v = load8_bf16<IOT>(kp + (int64_t)j * a.k.sn + dc * 8); u = load8_bf16<IOT>(qp0 + (int64_t)j * a.q.sn + dc * 8); }
            *(bf16x8 *)&Ksm[j * LDK + dc * 8] = v;
            *(bf16x8 *)&Qsm[j * LDK + dc * 8] = u;
        }
        for (int c = tid; c < V * DK; c += NT * 64) { const float t = a.sqk[((c / DK) * H + hh) * DK + (c % DK)]; sqk[c] = t; sqk2[c] = t * 1.4426950408889634f; }
        for (int c = tid; c < DK; c += NT * 64) { vs0[c] = a.vs0[hh * DK + c]; vsL[c] = a.vsL[hh * DK + c]; }
        if (tid == 0) wsig[0] = 1.f / (1.f + __expf(-*a.chain_logit));
    }
    {
    // qbar partials: thread (p, d) sums q[j][d] (bf16-rounded like the fragments) over the tokens j = p, p + NT, ... straight from
    // global memory (independent 2-byte loads); colpart doubles as [NT][DK] scratch here.  (32 five-step lane reductions of the
    // fragments serialise on their shuffles.)
    {
        const int d = tid % DK, p = tid / DK;
        if (p < NT) {
            const IOT *qp = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + d;
            float sacc = 0.f;
            for (int j = p; j < N; j += NT) sacc += bf2f(f2bf(ld_as_f32<IOT>(qp + (int64_t)j * a.q.sn)));
            colpart[p * DK + d] = sacc;
        }
    }
    LDS_BARRIER();                                    // Ksm staged, qbar partials written
    key_mean_partials<NT, DK>(rS, Ksm, N, tid);         // rS is free until the chains are done
    LDS_BARRIER();
    if (tid < DK) {
        float sk = 0.f, sq = 0.f;
        for (int p = 0; p < NT * 64 / DK; ++p) sk += rS[p * DK + tid];
        for (int ww = 0; ww < NT; ++ww) sq += colpart[ww * DK + tid];
        kbar[tid] = sk * invN; qbar[tid] = sq * invN;
    }
    LDS_BARRIER();
    }

    const FusedSavedLayout SL = fused_saved_layout<NT, DK>(N, V, SAVE);
    unsigned char *svb = (unsigned char *)a.saved + (size_t)blockIdx.x * SL.stride;    // this (b,h)'s record
    // ---------------- helpers ----------------
    const IOT *qrow = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh + (int64_t)qi * a.q.sn;
    auto make_qe_t = [&](bf16x8 (&qe)[KS], const float *tab, int v) {   // Qe_v fragments; q re-read from L2 (keeps 16 VGPRs free)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 qv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qok) qv = load8_bf16<IOT>(qrow + 16 * s + 8 * h);
            const float4 s0 = *(const float4 *)&tab[v * DK + 16 * s + 8 * h];
            const float4 s1 = *(const float4 *)&tab[v * DK + 16 * s + 8 * h + 4];
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) qe[s][j] = (short)f2bf(bf2f((unsigned short)qv[j]) * sc[j]);
        }
    };
    auto make_qe = [&](bf16x8 (&qe)[KS], int v) { make_qe_t(qe, sqk, v); };      // natural scores (mix)
    // same from raw q fragments already in registers (no global round trip inside the mix tile loop)
    auto scale_qe = [&](bf16x8 (&qe)[KS], const bf16x8 (&qraw)[KS], int v) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 s0 = *(const float4 *)&sqk[v * DK + 16 * s + 8 * h], s1 = *(const float4 *)&sqk[v * DK + 16 * s + 8 * h + 4];
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) qe[s][j] = (short)f2bf(bf2f((unsigned short)qraw[s][j]) * sc[j]);
        }
    };
    auto make_qe2 = [&](bf16x8 (&qe)[KS], int v) { make_qe_t(qe, sqk2, v); };    // scores * log2(e) (softmax slabs: exp2, no multiply)
    auto s_tile = [&](const bf16x8 (&qe)[KS], int t) -> f32x16 {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        bf16x8 af[KS];                    // all K fragments of the tile requested first: one LDS round trip instead of one per MFMA
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = *(const bf16x8 *)&(Ksm + r * LDK + 8 * h)[(32 * t) * LDK + 16 * s];
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(af[s]));
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], qe[s], acc, 0, 0, 0);
        return acc;
    };
    // ---- tile-streamed primitives: no N x 32 fp32 slab is ever held in registers; chain state lives as packed
    //      bf16 B-operand fragments (8 VGPRs per 32x32 tile), accumulators are consumed tile by tile.
    constexpr float NEG = -1e30f;                     // finite "-inf" (keeps the online softmax NaN-free)
    // softmax constant of this lane's query for view v, online over key tiles (scores pre-scaled by log2 e):
    //   c = log2 sum_j 2^(S'[i,j])   ->   A_v[i,j] = 2^(S'[i,j] - c)                                  :500-507
    // only tiles that contain keys >= N pay for the mask
    auto row_const = [&](const bf16x8 (&qe)[KS]) -> float {
        float m = NEG, l = 0.f;
#pragma unroll 2
        for (int t = 0; t < NT; ++t) {
            f32x16 S = s_tile(qe, t);
            if (32 * t + 32 > N) {
#pragma unroll
                for (int g = 0; g < 16; ++g) S[g] = (32 * t + tile_row(g, h) >= N) ? NEG : S[g];     // select (no per-lane branch around a vector element write)
            }
            float tm = NEG;
#pragma unroll
            for (int g = 0; g < 16; ++g) tm = fmaxf(tm, S[g]);
            const float mn = fmaxf(m, tm);
            float sm = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) sm += __builtin_amdgcn_exp2f(S[g] - mn);
            l = fmaf(l, __builtin_amdgcn_exp2f(m - mn), sm);
            m = mn;
        }
        const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64);
        const float mx = fmaxf(m, m2);
        return mx + __builtin_amdgcn_logf(l * __builtin_amdgcn_exp2f(m - mx) + l2 * __builtin_amdgcn_exp2f(m2 - mx));
    };
    auto a_tile = [&](const bf16x8 (&qe)[KS], int t, float c) -> f32x16 {   // A_v^T tile (keys >= N -> 0)
        f32x16 S = s_tile(qe, t);
#pragma unroll
        for (int g = 0; g < 16; ++g) S[g] = __builtin_amdgcn_exp2f(S[g] - c);
        if (32 * t + 32 > N) {
#pragma unroll
            for (int g = 0; g < 16; ++g) S[g] = (32 * t + tile_row(g, h) >= N) ? 0.f : S[g];
        }
        return S;
    };
    auto pack_tile = [&](bf16x8 &lo, bf16x8 &hi, const f32x16 &x) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { lo[j] = (short)f2bf(x[j]); hi[j] = (short)f2bf(x[8 + j]); }
    };
    // epilogue of the LAST chain step for one output tile: v = log(C + eps); row-sum, per-wave column partials
    // (butterfly over the 32 lanes of a half: 16 shuffles per tile; lane r even ends with register r>>1)
    auto log_tile = [&](f32x16 &X, int t, float &rs) {
        float c[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float v = __logf(X[g] + EPSC);
            X[g] = v;
            rs += (32 * t + tile_row(g, h) < N) ? v : 0.f;
            c[g] = qok ? v : 0.f;
        }
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
        if ((r & 1) == 0) colpart[w * NP + 32 * t + tile_row(r >> 1, h)] = c[0];
    };
    // chain product (transposed, row-block local):  X <- A_{o[V-1]}^T .. A_{o[1]}^T A_{o[0]}^T[:, I]
    // the last step hands every fp32 output tile to `epi(to, acc)`.
    //
    // Part pipeline.  The A operand of step m is the image A_m^T (rows = keys, k-permuted query columns).  It is never resident as a
    // whole: it streams through a ring of two 32-key PARTS.  Per part every wave (1) builds its 32 x 32 piece of the NEXT part --
    // keys of that part x its own queries -- and (2) multiplies the CURRENT part into its row block (one output tile, 2 NT MFMAs),
    // then one barrier.  The piece is computed with the operands of the score MFMA swapped (A = this wave's scaled q fragments,
    // B = the part's K rows): the tile comes out with lane = key, registers = queries, which is the image's row order -- two 16-byte
    // LDS stores, no transpose -- and the softmax constant of a query, now a per-REGISTER value, enters as the initial accumulator
    // (-c from LDS), so the tile needs 16 exp2 and no subtraction.  A piece depends on q, k and the row constants only, never on the
    // chain state, so the stream runs across the step boundaries: 4 NT parts per chain, the only serial dependency is each wave's
    // own slab.  SIMD partners (waves w, w + 4) run build / multiply in opposite order, so one's exp2 / pack work sits beside the
    // other's MFMAs.
    const bool klast = N > NP - 16;        // N <= NP - 16: the last 16-wide k-step of every contraction over keys is all padding
    // ... and so is the last 16-B chunk of every packed slab of the record (keys / queries NP - 16 .. NP - 1): not stored (7 % of the export
    // stream at N = 197); the backward does not load it either
    const bool ctrim = SAVE && HEAD == 0 && !klast;
    auto load_qe2 = [&](bf16x8 (&qe)[KS], int v) {      // Qe_v fragments x log2(e) of this lane's query, q from the LDS rows
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 qv = *(const bf16x8 *)&Qsm[qi * LDK + 16 * s + 8 * h];
            const float4 s0 = *(const float4 *)&sqk2[v * DK + 16 * s + 8 * h], s1 = *(const float4 *)&sqk2[v * DK + 16 * s + 8 * h + 4];
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) qe[s][j] = (short)f2bf(bf2f((unsigned short)qv[j]) * sc[j]);
        }
    };
    auto build_piece = [&](const bf16x8 (&qe)[KS], int v, int to, int slot) {
        f32x16 acc;
        const float *nc = cst + v * NP + 32 * w + 4 * h;            // -c of queries 32w + 8 q4 + 4h + {0..3} = rows of registers 4 q4 ..
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 c4 = *(const float4 *)&nc[8 * q4];
            acc[4 * q4] = c4.x; acc[4 * q4 + 1] = c4.y; acc[4 * q4 + 2] = c4.z; acc[4 * q4 + 3] = c4.w;
        }
        bf16x8 kf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[s] = *(const bf16x8 *)&(Ksm + r * LDK + 8 * h)[(32 * to) * LDK + 16 * s];
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qe[s], kf[s], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = __builtin_amdgcn_exp2f(acc[g]);
        if (32 * to + 32 > N) {
            const bool dead = 32 * to + r >= N;                     // this lane's key is padding: its image row is zero
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = dead ? 0.f : acc[g];
        }
        bf16x8 lo, hi;
        pack_tile(lo, hi, acc);
        unsigned short *dst = ring + slot * Cfg::PART + r * LDA + 32 * w + 8 * h;
        *(bf16x8 *)dst = lo;
        *(bf16x8 *)(dst + 16) = hi;
    };
    auto run_chain = [&](bool forward, bf16x8 (&Xp)[NT][2], auto &&epi) {      // Xp: the final product, packed, on return
        bf16x8 Xn[NT][2];
        bf16x8 qe[KS];
        {
            const int v = forward ? 0 : V - 1;
            load_qe2(qe, v);
            const float c = -cst[v * NP + qi];
#pragma unroll
            for (int t = 0; t < NT; ++t) { const f32x16 A = a_tile(qe, t, c); pack_tile(Xp[t][0], Xp[t][1], A); }
        }
        LDS_BARRIER();                       // the ring is free (previous chain / phase)
        bf16x8 idl, idh;                     // identity fragments of the export transposes: kept live (regenerating them costs 48 VALU
        identity_frags(idl, idh, r, h);      // instructions per part, and the vector ALU, not the matrix pipe, is what a part waits for)
        // pieces are built LA parts ahead of the product and the workgroup meets at a barrier every LA parts (ring of 2 LA slots): the
        // fixed cost of a rendezvous -- LDS round trips in front of the first MFMA, the last MFMA's latency, the barrier itself, about
        // 900 cycles -- is paid per LA parts
        constexpr int LA = Cfg::RING / 2;
        load_qe2(qe, forward ? 1 : V - 2);
        static_for<0, LA>([&](auto pc) {
            constexpr int p0 = decltype(pc)::value;
            if constexpr (p0 < NT) build_piece(qe, forward ? 1 : V - 2, p0, p0);
            else { if (V > 2) { load_qe2(qe, forward ? 2 : V - 3); build_piece(qe, forward ? 2 : V - 3, p0 - NT, p0); } }    // NT == 1, LA == 2: the second part is the next step's
        });
        LDS_BARRIER();
        int cur = 0, P = 0;                  // ring slot of the current part, global part index
        for (int m = 1; m < V; ++m) {
            const int v = forward ? m : V - 1 - m;
            const bool last = m == V - 1;
            typedef __attribute__((ext_vector_type(4))) unsigned int u4;
            u4 *out = (u4 *)(svb + (forward ? SL.oT : SL.oU) + (size_t)(m - 1) * NP * LDA * 2) + lane;
            static_for<0, NT>([&](auto tc) {
                // One part = one hand-ordered instruction stream: the three independent jobs of a part -- the next piece (score MFMAs, exp2,
                // pack, LDS stores), this part's product (2 NT MFMAs on LDS fragments) and the export of one prefix tile (two transposing
                // MFMAs, pack, stores) -- are interleaved so that the VALU work sits in the shadow of the product's MFMAs.
                // One barrier per part: the piece of part P + 1 is written while part P is read (two ring slots).
                constexpr int to = decltype(tc)::value;
                constexpr bool wrap = to + LA >= NT;                     // the piece built here belongs to the next step
                constexpr int tb = (to + LA) % NT;                       // its key tile
                constexpr int NK = 2 * NT, PF = NK < MOPK_PF ? NK : MOPK_PF;
                FSTAMP2(!forward && m == 2);
                FSTAMP3(0);
                const int nxt = (cur + LA) & (Cfg::RING - 1);             // slot of the piece built here (part P + LA)
                // the next part: tile to + 1 of this step, or tile 0 of the next step's view (the chain's very last part rebuilds a piece
                // nobody reads: one dummy piece per chain keeps the stream branch-free)
                const int vb = (wrap && !last) ? (forward ? m + 1 : V - 2 - m) : v;
                if (to + LA == NT && !last) load_qe2(qe, vb);        // first piece of the next step's view (NT >= LA)
                // (B) next piece: -c of this wave's queries (a per-REGISTER constant in this orientation) as the initial accumulator, K rows
                f32x16 bacc;
                {
                    const float *nc = cst + vb * NP + 32 * w + 4 * h;   // -c of queries 32w + 8 q4 + 4h + {0..3} = rows of registers 4 q4 ..
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float4 c4 = *(const float4 *)&nc[8 * q4];
                        bacc[4 * q4] = c4.x; bacc[4 * q4 + 1] = c4.y; bacc[4 * q4 + 2] = c4.z; bacc[4 * q4 + 3] = c4.w;
                    }
                }
                bf16x8 kf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) kf[s] = *(const bf16x8 *)&(Ksm + r * LDK + 8 * h)[(32 * tb) * LDK + 16 * s];
                FSTAMP3(1);
                // (C) export of tile `to` of the prefix product for the backward's dA GEMMs in "row slab" order: wave w' of the backward
                // reads, per lane (key a = 32w' + r'), the fragments {T[i, a] : i} -> [wave w'][chunk q][lane], 16 B each.  That is the
                // TRANSPOSE of the tile this wave holds (lane = query, registers = keys), taken on the matrix core: D = X . I (two MFMAs
                // with identity B fragments) comes back with lane = key, registers = queries, and its packed halves are exactly chunks
                // q = 2w, 2w+1 of wave t's slab.  (non-temporal stores: the record is not read again by this kernel)
                f32x16 tr = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef MOPK_WHATIF_NOEXPORT
                constexpr bool EXPORT = false;
#else
                constexpr bool EXPORT = SAVE;
#endif
                if (EXPORT) {
                    tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xp[to][0], idl, tr, 0, 0, 0);
                    tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xp[to][1], idh, tr, 0, 0, 0);
                }
                FSTAMP3(2);
                // (D) scores of the next piece, operands swapped: lane = key, registers = queries
#ifndef MOPK_WHATIF_NOBUILD
#pragma unroll
                for (int s = 0; s < KS; ++s) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qe[s], kf[s], bacc, 0, 0, 0);
#endif
                // (A) the first A fragments of this part
                const unsigned abase = (unsigned)(uintptr_t)(ring + cur * Cfg::PART + r * LDA + 8 * h);
                bf16x8 rg[PF];
                static_for<0, PF>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    bf16x8 tmp;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tmp) : "v"(abase), "i"(32 * f));
                    rg[f] = tmp;
                });
                FSTAMP3(3);
                // (E) this part's product; k-step k carries filler(k)
                bf16x8 plo, phi;
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                static_for<0, NK>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    constexpr int pend = (NK - 1 - k) < (PF - 1) ? (NK - 1 - k) : (PF - 1);
                    bf16x8 af = rg[k % PF];
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af) : "i"(pend));
#ifndef MOPK_WHATIF_NOGEMM
                    if (k < NK - 1 || klast) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, Xp[k >> 1][k & 1], acc, 0, 0, 0);
#else
                    if (k == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, Xp[k >> 1][k & 1], acc, 0, 0, 0);
                    asm volatile("" : "+v"(af));
#endif
                    if constexpr (k + PF < NK) {
                        bf16x8 tmp;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tmp) : "v"(abase), "i"(32 * (k + PF)));
                        rg[(k + PF) % PF] = tmp;
                    }
                    // fillers.  NK >= 8: export pack + stores at k = 0, 1; exp2 of the piece from k = 2 (its MFMAs are two latencies old);
                    // pack at NK - 3, LDS stores at NK - 2.  Short chains (NT < 4) put everything behind the last MFMA.
                    if constexpr (k == 1 || k == 4 || k == 7 || k == 10 || k == 12) FSTAMP3(4 + (k == 1 ? 0 : k == 4 ? 1 : k == 7 ? 2 : k == 10 ? 3 : 4));
                    constexpr bool SHORT = NK < 8;
                    if constexpr (!SHORT) {
                        if constexpr (k == 0) {
                            if (EXPORT) {
                                bf16x8 lo, hi;
                                pack_tile(lo, hi, tr);    // exact: every entry is one bf16 value times 1.0
                                __builtin_nontemporal_store(__builtin_bit_cast(u4, lo), &out[((size_t)to * 2 * NT + 2 * w) * 64]);
                                if (!(ctrim && w == NT - 1)) __builtin_nontemporal_store(__builtin_bit_cast(u4, hi), &out[((size_t)to * 2 * NT + 2 * w + 1) * 64]);
                            }
                        }
                        constexpr int E0 = 2, ESTEPS = NK - 3 - E0;               // exp2 steps: k = E0 .. NK - 4
                        if constexpr (k >= E0 && k < E0 + ESTEPS) {
                            constexpr int g0 = 16 * (k - E0) / ESTEPS, g1 = 16 * (k - E0 + 1) / ESTEPS;
#ifndef MOPK_WHATIF_NOBUILD
#pragma unroll
                            for (int g = g0; g < g1; ++g) bacc[g] = __builtin_amdgcn_exp2f(bacc[g]);
#endif
                        }
                        if constexpr (k == NK - 3) {
                            if (32 * tb + 32 > N) {
                                const bool dead = 32 * tb + r >= N;             // this lane's key is padding: its image row is zero
#pragma unroll
                                for (int g = 0; g < 16; ++g) bacc[g] = dead ? 0.f : bacc[g];
                            }
                            pack_tile(plo, phi, bacc);
                        }
                        if constexpr (k == NK - 2) {
                            unsigned short *dst = ring + nxt * Cfg::PART + r * LDA + 32 * w + 8 * h;
                            *(bf16x8 *)dst = plo;
                            *(bf16x8 *)(dst + 16) = phi;
                        }
                    }
                });
                if constexpr (NK < 8) {
                    if (EXPORT) {
                        bf16x8 lo, hi;
                        pack_tile(lo, hi, tr);
                        __builtin_nontemporal_store(__builtin_bit_cast(u4, lo), &out[((size_t)to * 2 * NT + 2 * w) * 64]);
                        if (!(ctrim && w == NT - 1)) __builtin_nontemporal_store(__builtin_bit_cast(u4, hi), &out[((size_t)to * 2 * NT + 2 * w + 1) * 64]);
                    }
#pragma unroll
                    for (int g = 0; g < 16; ++g) bacc[g] = __builtin_amdgcn_exp2f(bacc[g]);
                    if (32 * tb + 32 > N) {
                        const bool dead = 32 * tb + r >= N;
#pragma unroll
                        for (int g = 0; g < 16; ++g) bacc[g] = dead ? 0.f : bacc[g];
                    }
                    pack_tile(plo, phi, bacc);
                    unsigned short *dst = ring + nxt * Cfg::PART + r * LDA + 32 * w + 8 * h;
                    *(bf16x8 *)dst = plo;
                    *(bf16x8 *)(dst + 16) = phi;
                }
                FSTAMP3(9);
                pack_tile(Xn[to][0], Xn[to][1], acc);
                if (last) epi(to, acc, Xn[to][0], Xn[to][1]);
                FSTAMP3(10);
                if (((P + 1) & (LA - 1)) == 0) LDS_BARRIER();
                FSTAMP3(11);
                cur = (cur + 1) & (Cfg::RING - 1); ++P;
            });
#pragma unroll
            for (int t = 0; t < NT; ++t) { Xp[t][0] = Xn[t][0]; Xp[t][1] = Xn[t][1]; }
        }
    };

    FSTAMP();
    REFRESH();
    // ---------------- softmax constants of every view for this lane's query (wave-private: a wave only ever builds pieces of its own queries)
    for (int v = 0; v < V; ++v) {
        bf16x8 qe[KS];
        load_qe2(qe, v);
        const float c = row_const(qe);
        if (h == 0) cst[v * NP + qi] = -c;
    }
    FSTAMP();
    REFRESH();
    // ---------------- chain <- : only its log-means survive           :513-515, :521
    {
        float rs = 0.f;
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        u4 *cbp = (u4 *)(svb + SL.oCB + (size_t)w * NT * 8 * 64 * 4) + lane;       // packed C<- slab of this wave
        bf16x8 Xb[NT][2];
        run_chain(false, Xb, [&](int to, f32x16 &acc, const bf16x8 &lo, const bf16x8 &hi) {
            if (SAVE) { __builtin_nontemporal_store(__builtin_bit_cast(u4, lo), &cbp[(2 * to) * 64]); if (!(ctrim && to == NT - 1)) __builtin_nontemporal_store(__builtin_bit_cast(u4, hi), &cbp[(2 * to + 1) * 64]); }
            log_tile(acc, to, rs);
        });
        rs += __shfl_xor(rs, 32, 64);
        if (h == 0) rCl[qi] = rs * invN;
    }
    LDS_BARRIER();
    if (tid < NP) { float c = 0.f; for (int ww = 0; ww < NT; ++ww) c += colpart[ww * NP + tid]; cCl[tid] = c * invN; }
    FSTAMP();
    REFRESH();
    // ---------------- chain -> : C-> kept as packed bf16 (for y_chain) and log C-> as packed fp16 (for the mix)
    unsigned int crp[NT][8];              // Cr (later Smix) as packed fp16
    IOT *yp = (IOT *)a.y.ptr + b * a.y.sb + hh * a.y.sh + (int64_t)qi * a.y.sn;
    float *ych = (float *)(svb + SL.oYch);                         // saved: w * y_chain (N,dk) fp32
    {
        bf16x8 Xc[NT][2];
        run_chain(true, Xc, [&](int, f32x16 &, const bf16x8 &, const bf16x8 &) {});
        if (SAVE) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u4;
            u4 *cfp = (u4 *)(svb + SL.oCF + (size_t)w * NT * 8 * 64 * 4) + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) { __builtin_nontemporal_store(__builtin_bit_cast(u4, Xc[t][0]), &cfp[(2 * t) * 64]); if (!(ctrim && t == NT - 1)) __builtin_nontemporal_store(__builtin_bit_cast(u4, Xc[t][1]), &cfp[(2 * t + 1) * 64]); }
        }
        // log C-> from the bf16-rounded product (the same rounding the backward sees): means + packed fp16 copy
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x16 c;
#pragma unroll
            for (int j = 0; j < 8; ++j) { c[j] = bf2f((unsigned short)Xc[t][0][j]); c[8 + j] = bf2f((unsigned short)Xc[t][1][j]); }
            log_tile(c, t, rs);
            if (!SAVE) {                  // with the record, the mix loop re-reads the C-> export instead (56 registers free through the <- chain)
#pragma unroll
                for (int p = 0; p < 8; ++p) crp[t][p] = pack_h2(c[2 * p], c[2 * p + 1]);
            }
        }
        rs += __shfl_xor(rs, 32, 64);
        if (h == 0) rCr[qi] = rs * invN;
        LDS_BARRIER();                      // AT free: build V0^T, VL^T (k-permuted key columns); colpart complete
        {
            const IOT *v0p = (const IOT *)a.v0.ptr + b * a.v0.sb + hh * a.v0.sh;
            const IOT *vLp = (const IOT *)a.vL.ptr + b * a.vL.sb + hh * a.vL.sh;
            constexpr int CH = DK / 8;
            for (int c = tid; c < NP * CH; c += NT * 64) {
                const int j = c % NP, dc = c / NP;     // consecutive lanes = consecutive keys: conflict-free 2-byte transposing stores
                bf16x8 x0 = {0, 0, 0, 0, 0, 0, 0, 0}, xL = x0;
                if (j < N) { x0 = load8_bf16<IOT>(v0p + (int64_t)j * a.v0.sn + dc * 8); xL = load8_bf16<IOT>(vLp + (int64_t)j * a.vL.sn + dc * 8); }
                const int col = (j & ~15) + kperm16(j & 15);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int d = dc * 8 + e;
                    VT0[d * LDA + col] = f2bf(bf2f((unsigned short)x0[e]) * vs0[d]);
                    VTL[d * LDA + col] = f2bf(bf2f((unsigned short)xL[e]) * vsL[d]);
                }
            }
            if (DK < DP) for (int c = tid; c < (DP - DK) * LDA; c += NT * 64) { VT0[DK * LDA + c] = 0; VTL[DK * LDA + c] = 0; }
            if constexpr (HEAD == 1) {      // q rows [token][d] in the region the low-rank head uses for its b vectors: A operand of S_v^T
                const IOT *qp = (const IOT *)a.q.ptr + b * a.q.sb + hh * a.q.sh;
                for (int c = tid; c < NP * CH; c += NT * 64) {
                    const int j = c / CH, dc = c % CH;
                    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (j < N) v = load8_bf16<IOT>(qp + (int64_t)j * a.q.sn + dc * 8);
                    *(bf16x8 *)&bT[j * LDK + dc * 8] = v;
                }
            }
            if (tid < NP) { float c = 0.f; for (int ww = 0; ww < NT; ++ww) c += colpart[ww * NP + tid]; c *= invN; cCr[tid] = c;
                            if (SAVE) ((float *)(svb + SL.oMeans))[2 * NP + tid] = c; }
            if (SAVE) {   // softmax constants of every view (still in `cst`, about to be overwritten by rS) and the log-means
                float *gc = (float *)(svb + SL.oCst), *gm = (float *)(svb + SL.oMeans);
                for (int c = tid; c < V * NP; c += NT * 64) gc[c] = -cst[c];
                if (tid < NP) { gm[tid] = rCr[tid]; gm[NP + tid] = rCl[tid]; gm[3 * NP + tid] = cCl[tid]; }
                LDS_BARRIER();          // cst fully copied before rS/cS overwrite it
            }
            // row / col means of S_v are linear in q, k:  rS_v[i] = Qe_v[i,:].kbar ; cS_v[j] = k[j,:].(sqk_v*qbar)
            // (written over the softmax constants `cst`, which the chains no longer need)
            if constexpr (HEAD == 0) {
                bf16x8 qf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) { bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0}; if (qok) v = load8_bf16<IOT>(qrow + 16 * s + 8 * h); qf[s] = v; }
                for (int v = 0; v < V; ++v) {
                    float p = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int d = 16 * s + 8 * h + j;
                            p = fmaf(bf2f((unsigned short)qf[s][j]) * sqk[v * DK + d], kbar[d], p);
                        }
                    p += __shfl_xor(p, 32, 64);
                    if (h == 0) rS[v * NP + qi] = p;
                }
                col_means_mfma<NT, DK>(cS, Ksm, sqk, qbar, V, w, r, h);
            }
        }
        LDS_BARRIER();                      // VT*, cCr, rS, cS complete
        const float wv = wsig[0];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {     // y_chain^T = VL^T C->^T       :556-560 (as C-> vL); parked in `saved`
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 af = *(const bf16x8 *)&(VTL + r * LDA + 8 * h)[(32 * dt) * LDA + 32 * t + 16 * s];
                    if (2 * t + s < 2 * NT - 1 || klast) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, Xc[t][s], acc, 0, 0, 0);
                }
            if (qok) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = 32 * dt + 8 * g4 + 4 * h;
                    if (d0 < DK) *(float4 *)&ych[(size_t)qi * DK + d0] = make_float4(wv * acc[4 * g4], wv * acc[4 * g4 + 1], wv * acc[4 * g4 + 2], wv * acc[4 * g4 + 3]);
                }
            }
        }
    }
    FSTAMP();
    REFRESH();
    // ---------------- gate vectors                                     :323-326
    const int E = HEAD == 0 ? dw.E : 0;                  // extra feature channels of the low-rank head (given as row / column means)
    const int C = 2 * V + 2 + E;
    const float *rowx = dw.rowx + (size_t)blockIdx.x * (E * N), *colx = dw.colx + (size_t)blockIdx.x * (E * N);
    float *Wsm = Cfg::WSM_EXTRA ? wsig + 8 : colpart;   // low-rank: [2][16][WST] gate-head weights (+bias in slot WST-1); dense: W1^T [C][16], b1, W2^T [16][4], b2.  colpart is dead from here on
    bf16x8 af4[4];                        // low-rank: a[g,k,i] as B fragments: slots [a_hi | a_lo | a_hi | 0]
    if constexpr (HEAD == 1) {
        // dense head weights, transposed so that one 16-byte LDS read yields the four values an inner loop needs:
        //   W1T[c][k] (k = hidden unit) at Wsm[c * 16 + k], b1 at Wsm[18 * 16 + k], W2T[k][m] at Wsm[320 + k * 4 + m], b2 at Wsm[384 + m]
        for (int c = tid; c < 16 * C; c += NT * 64) Wsm[(c % C) * 16 + c / C] = dw.W1[c];             // W1 is (16, C) row-major
        for (int c = tid; c < 16; c += NT * 64) Wsm[18 * 16 + c] = dw.b1[c];
        for (int c = tid; c < 64; c += NT * 64) Wsm[320 + (c % 16) * 4 + c / 16] = dw.W2[c];          // W2 is (4, 16) row-major
        if (tid < 4) Wsm[384 + tid] = dw.b2[tid];
#pragma unroll
        for (int g = 0; g < 4; ++g) af4[g] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    } else {
    for (int c = tid; c < 2 * 4 * R * (C + 1); c += NT * 64) {
        const int side = c / (4 * R * (C + 1)), rem = c % (4 * R * (C + 1)), o = rem / (C + 1), cc = rem % (C + 1);
        const float *Wg = side ? a.Wc : a.Wr, *bg = side ? a.bc : a.br;
        Wsm[(side * 16 + o) * WST + (cc < C ? cc : WST - 1)] = cc < C ? Wg[o * C + cc] : bg[o];
    }
    LDS_BARRIER();
    for (int p = tid; p < 4 * NP; p += NT * 64) {          // b[g,k,j] -> bT[g][j][slots] = [b_hi | b_hi | b_lo | 0], one (j, g) per thread-iteration
        const int j = p % NP, g = p / NP;
        unsigned short hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
        if (j < N) {
            // the (up to) four rank channels of this gate side by side: four independent fma chains, every feature read once
            const float *Wg4 = Wsm + (16 + g * R) * WST;
            float sk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sk[k] = Wg4[(k < R ? k : 0) * WST + WST - 1];
            for (int c = 0; c < V; ++c) {
                const float fc = cS[c * NP + j], fr = rS[c * NP + j];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float *Wo = Wg4 + (k < R ? k : 0) * WST; sk[k] = fmaf(Wo[c], fc, fmaf(Wo[V + c], fr, sk[k])); }
            }
            {
                const float f0 = cCr[j], f1 = cCl[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float *Wo = Wg4 + (k < R ? k : 0) * WST; sk[k] = fmaf(Wo[2 * V], f0, fmaf(Wo[2 * V + 1], f1, sk[k])); }
            }
            for (int e0 = 0; e0 < E; e0 += XU) {
                float fx[XU];
                load_extra(fx, colx, N, j, e0, E, true);
                for (int u = 0; u < XU && e0 + u < E; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) sk[k] = fmaf(Wg4[(k < R ? k : 0) * WST + 2 * V + 2 + e0 + u], fx[u], sk[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < R) { hi[k] = f2bf(sk[k]); lo[k] = f2bf(sk[k] - bf2f(hi[k])); }
        }
        unsigned short *row = bT + (g * NP + j) * BTS;
#pragma unroll
        for (int k = 0; k < 4; ++k) { row[k] = hi[k]; row[4 + k] = hi[k]; row[8 + k] = lo[k]; row[12 + k] = 0; }
    }
    float av16[4][4];                      // a[g,k] for this lane's query: 16 independent fma chains over the feature channels
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) av16[g][k] = Wsm[(g * R + (k < R ? k : 0)) * WST + WST - 1];
    for (int c = 0; c < V; ++c) {
        const float fr = rS[c * NP + qi], fc = cS[c * NP + qi];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float *Wo = Wsm + (g * R + (k < R ? k : 0)) * WST; av16[g][k] = fmaf(Wo[c], fr, fmaf(Wo[V + c], fc, av16[g][k])); }
    }
    {
        const float f0 = rCr[qi], f1 = rCl[qi];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float *Wo = Wsm + (g * R + (k < R ? k : 0)) * WST; av16[g][k] = fmaf(Wo[2 * V], f0, fmaf(Wo[2 * V + 1], f1, av16[g][k])); }
    }
    for (int e0 = 0; e0 < E; e0 += XU) {
        float fx[XU];
        load_extra(fx, rowx, N, qi, e0, E, qok);
        for (int u = 0; u < XU && e0 + u < E; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) av16[g][k] = fmaf(Wsm[(g * R + (k < R ? k : 0)) * WST + 2 * V + 2 + e0 + u], fx[u], av16[g][k]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float av[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) av[k] = k < R ? av16[g][k] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a2 = av[k] * 1.4426950408889634f;               // gate logits come out pre-scaled by log2(e): sigmoid = 1/(1+2^-z')
            const unsigned short hi = f2bf(a2), lo = f2bf(a2 - bf2f(hi));
            af4[g][k] = (short)hi;                          // h=0: slots 0-3 a_hi ; h=1: slots 8-11 a_hi
            af4[g][4 + k] = h == 0 ? (short)lo : (short)0;  // h=0: slots 4-7 a_lo ; h=1: slots 12-15 0
        }
    }
    }
    LDS_BARRIER();                      // bT complete
    FSTAMP();
    REFRESH();
    // ---------------- score-space mix, tile by tile                    :537-547
    const float nb = a.beta_not / (float)(V > 1 ? V - 1 : 1);
    float mxrow = -INFINITY;
    auto gate_tile = [&](int t, int g4) -> f32x16 {        // sigmoid(a_g^T b_g) for one 32x32 tile
        const f32x16 z0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bf16x8 bfrag = *(const bf16x8 *)&bT[(g4 * NP + 32 * t + r) * BTS + 8 * h];
        f32x16 z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfrag, af4[g4], z0, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) z[g] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-z[g]));
        return z;
    };
    bf16x8 qraw[KS];                      // raw q fragments resident through the mix loop
#pragma unroll
    for (int s = 0; s < KS; ++s) { bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0}; if (qok) v = load8_bf16<IOT>(qrow + 16 * s + 8 * h); qraw[s] = v; }
    bf16x8 kraw[HEAD == 1 ? KS : 1];      // dense head: raw k fragments of this lane's token (B operand of the S_v^T tiles)
    if constexpr (HEAD == 1) {
        const IOT *krow = (const IOT *)a.k.ptr + b * a.k.sb + hh * a.k.sh + (int64_t)qi * a.k.sn;
#pragma unroll
        for (int s = 0; s < KS; ++s) { bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0}; if (qok) v = load8_bf16<IOT>(krow + 16 * s + 8 * h); kraw[s] = v; }
    }
    // S_v^T tile: element (lane = my token i, register = token j of tile t) = S_v(j, i) = (q_j * sqk_v) . k_i  -- q rows from LDS
    auto st_tile = [&](const bf16x8 (&ke)[KS], int t) -> f32x16 {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        bf16x8 af[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = *(const bf16x8 *)&(bT + r * LDK + 8 * h)[(32 * t) * LDK + 16 * s];
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(af[s]));
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], ke[s], acc, 0, 0, 0);
        return acc;
    };
    // runtime loop over key tiles (an unrolled one makes hipcc overlap the tiles' live ranges and
    // spill ~1000 VGPRs); the per-tile packed Cr/Smix registers are selected with a uniform switch.
#pragma nounroll
    for (int t = 0; t < NT; ++t) {
        unsigned int cw[8];
        if (SAVE) {                       // log C-> of this tile from the wave's own export: the same bf16 values, the same fp16 packing
            typedef __attribute__((ext_vector_type(4))) unsigned int u4;
            const u4 *cfp = (const u4 *)(svb + SL.oCF + (size_t)w * NT * 8 * 64 * 4) + lane;
            const bool tr1 = ctrim && t == NT - 1;              // trimmed chunk: the neighbour is re-read (its keys are padding, masked below)
            const bf16x8 cl = __builtin_bit_cast(bf16x8, cfp[(2 * t) * 64]), ch = __builtin_bit_cast(bf16x8, cfp[(2 * t + (tr1 ? 0 : 1)) * 64]);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                cw[p] = pack_h2(__logf(bf2f((unsigned short)cl[2 * p]) + EPSC), __logf(bf2f((unsigned short)cl[2 * p + 1]) + EPSC));
                cw[4 + p] = pack_h2(__logf(bf2f((unsigned short)ch[2 * p]) + EPSC), __logf(bf2f((unsigned short)ch[2 * p + 1]) + EPSC));
            }
        } else {
        switch (t) {
#define MOPK_CASE(T_) case T_: if (T_ < NT) { _Pragma("unroll") for (int p = 0; p < 8; ++p) cw[p] = crp[T_ < NT ? T_ : 0][p]; } break;
            MOPK_CASE(0) MOPK_CASE(1) MOPK_CASE(2) MOPK_CASE(3) MOPK_CASE(4) MOPK_CASE(5) MOPK_CASE(6)
#undef MOPK_CASE
            default: break;
        }
        }
        f32x16 S0, O, L;
        {
            bf16x8 qe[KS];
            scale_qe(qe, qraw, 0);
            S0 = s_tile(qe, t);
            f32x16 mx = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, se;
            O = mx;
#pragma unroll
            for (int g = 0; g < 16; ++g) se[g] = 1.f;
            for (int v = 1; v < V; ++v) {     // online logsumexp over views, relative to S_0
                scale_qe(qe, qraw, v);
                const f32x16 Sv = s_tile(qe, t);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    O[g] += Sv[g];
                    const float d = Sv[g] - S0[g];
                    const float e = __expf(-fabsf(d - mx[g]));
                    se[g] = d > mx[g] ? fmaf(se[g], e, 1.f) : se[g] + e;
                    mx[g] = fmaxf(mx[g], d);
                }
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) L[g] = mx[g] + __logf(se[g]);   // lse_v S_v - S_0
        }
        if (SAVE) {                       // L x log2(e) in fp32, accumulator order (the backward's S_L slab: [wave][4 t + q][lane] x 16 B)
            f32x4 *lp = (f32x4 *)(svb + SL.oL + (size_t)w * 2 * NT * 8 * 64 * 4) + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (!(ctrim && t == NT - 1 && q >= 2)) __builtin_nontemporal_store(f32x4{L[4 * q], L[4 * q + 1], L[4 * q + 2], L[4 * q + 3]} * 1.4426950408889634f, &lp[(4 * t + q) * 64]);
        }
        // Smix = S0 + (G_and - nb G_not) O + G_or L + G_chain Cr, one gate at a time
        f32x16 G3d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (HEAD == 1) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u4;
            const u4 *cbp = (const u4 *)(svb + SL.oCB + (size_t)w * NT * 8 * 64 * 4) + lane;      // this wave's own C<- export
            const bf16x8 cbl = __builtin_bit_cast(bf16x8, cbp[(2 * t) * 64]), cbh = __builtin_bit_cast(bf16x8, cbp[(2 * t + 1) * 64]);
            f32x16 Crv, Clv;                      // log C-> (from the packed fp16 registers) and log C<- of this tile
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                Crv[2 * p] = h2_lo(cw[p]); Crv[2 * p + 1] = h2_hi(cw[p]);
                Clv[p] = __logf(bf2f((unsigned short)cbl[p]) + EPSC); Clv[8 + p] = __logf(bf2f((unsigned short)cbh[p]) + EPSC);
            }
            // rolled over the four register quarters (unrolled, hipcc overlaps them and spills 700 registers); the quarter's registers
            // are picked with a wave-uniform index into the accumulator vectors (register-indexed moves)
#pragma nounroll
            for (int q4 = 0; q4 < 4; ++q4) {                          // registers 4 q4 .. 4 q4 + 3 of the tile: four edges per lane
                typedef __attribute__((ext_vector_type(2))) float f32x2;   // edge pairs: v_pk_fma_f32 does two edges per issue
                f32x2 z1[16][2];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 bv = *(const float4 *)&Wsm[18 * 16 + 4 * k4];
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int e = 0; e < 2; ++e) z1[4 * k4 + kk][e] = f32x2{bb[kk], bb[kk]};
                }
                auto accum = [&](int c, int e, f32x2 f) {              // z1[:, pair e] += W1[:, c] (x) f, f = two of the quarter's four edges
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const float4 wv = *(const float4 *)&Wsm[c * 16 + 4 * k4];
                        z1[4 * k4][e] = __builtin_elementwise_fma(f32x2{wv.x, wv.x}, f, z1[4 * k4][e]);
                        z1[4 * k4 + 1][e] = __builtin_elementwise_fma(f32x2{wv.y, wv.y}, f, z1[4 * k4 + 1][e]);
                        z1[4 * k4 + 2][e] = __builtin_elementwise_fma(f32x2{wv.z, wv.z}, f, z1[4 * k4 + 2][e]);
                        z1[4 * k4 + 3][e] = __builtin_elementwise_fma(f32x2{wv.w, wv.w}, f, z1[4 * k4 + 3][e]);
                    }
                };
                {
                    // All views' scores of two edges per lane from ONE matrix-core chain per operand order: the 32 A rows are (key, view)
                    // pairs -- row m feeds accumulator register (m & 3) + 4 (m >> 3) of lane half (m >> 2) & 1, so register v (8 + v) of
                    // a lane is S_v of its first (second) edge -- and each A lane scales its key's row by its view's sqk on the way in.
                    // Two such pairs per quarter: 16 MFMAs and 4 x 8 products per k-step instead of 2V chains and 2V re-scaled fragments.
                    const int m = lane & 31, hp = (m >> 2) & 1, gq = (m & 3) + 4 * (m >> 3), va = min(gq & 7, V - 1);
                    const float *sca = sqk + va * DK + 8 * h;
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const int key = 32 * t + tile_row(4 * q4 + 2 * pr + (gq >> 3), hp);
                        const unsigned short *kr = Ksm + key * LDK + 8 * h, *qr = bT + key * LDK + 8 * h;
                        f32x16 xs = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, xt = xs;
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
                            xs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qraw[s], xs, 0, 0, 0);      // S_v(i, j) = (k_j * sqk_v) . q_i
                            xt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kraw[s], xt, 0, 0, 0);      // S_v(j, i) = (q_j * sqk_v) . k_i
                        }
#pragma unroll
                        for (int v = 0; v < 8; ++v) {
                            if (v < V) {
                                accum(v, pr, f32x2{xs[v], xs[8 + v]});
                                accum(V + v, pr, f32x2{xt[v], xt[8 + v]});
                            }
                        }
                        asm volatile("" ::: "memory");              // the second pair's chain after the first pair's accumulation (live ranges)
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    accum(2 * V, e, f32x2{Crv[4 * q4 + 2 * e], Crv[4 * q4 + 2 * e + 1]});
                    accum(2 * V + 1, e, f32x2{Clv[4 * q4 + 2 * e], Clv[4 * q4 + 2 * e + 1]});
                }
                f32x2 zz[4][2];
                {
                    const float4 bv = *(const float4 *)&Wsm[384];
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int e = 0; e < 2; ++e) zz[m][e] = f32x2{bb[m], bb[m]};
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float4 wv = *(const float4 *)&Wsm[320 + 4 * k];
                    const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x2 u = z1[k][e];                     // gelu_tanh(u) = u * sigmoid(2 sqrt(2/pi) (u + 0.044715 u^3))
                        const f32x2 a = -1.5957691216057308f * (u + 0.044715f * u * u * u);
                        const f32x2 hv = u * f32x2{__builtin_amdgcn_rcpf(1.f + __expf(a[0])), __builtin_amdgcn_rcpf(1.f + __expf(a[1]))};
#pragma unroll
                        for (int m = 0; m < 4; ++m) zz[m][e] = __builtin_elementwise_fma(f32x2{ww[m], ww[m]}, hv, zz[m][e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int g = 4 * q4 + e;
                    const float g0 = __builtin_amdgcn_rcpf(1.f + __expf(-zz[0][e >> 1][e & 1])), g1 = __builtin_amdgcn_rcpf(1.f + __expf(-zz[1][e >> 1][e & 1]));
                    const float g2 = __builtin_amdgcn_rcpf(1.f + __expf(-zz[2][e >> 1][e & 1]));
                    S0[g] = fmaf(g0, O[g], S0[g]);
                    S0[g] = fmaf(g1, L[g], S0[g]);
                    S0[g] = fmaf(-nb * g2, O[g], S0[g]);
                    G3d[g] = __builtin_amdgcn_rcpf(1.f + __expf(-zz[3][e >> 1][e & 1]));
                }
            }
        } else {
        {
            const f32x16 G = gate_tile(t, 0);
#pragma unroll
            for (int g = 0; g < 16; ++g) S0[g] = fmaf(G[g], O[g], S0[g]);
        }
        {
            const f32x16 G = gate_tile(t, 1);
#pragma unroll
            for (int g = 0; g < 16; ++g) S0[g] = fmaf(G[g], L[g], S0[g]);
        }
        {
            const f32x16 G = gate_tile(t, 2);
#pragma unroll
            for (int g = 0; g < 16; ++g) S0[g] = fmaf(-nb * G[g], O[g], S0[g]);
        }
        }
        {
            f32x16 G = G3d;
            if constexpr (HEAD == 0) G = gate_tile(t, 3);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float s0 = fmaf(G[2 * p], h2_lo(cw[p]), S0[2 * p]);
                float s1 = fmaf(G[2 * p + 1], h2_hi(cw[p]), S0[2 * p + 1]);
                if (32 * t + 32 > N) {              // wave-uniform: only the tile that holds padding keys is masked
                    s0 = (32 * t + tile_row(2 * p, h) >= N) ? -INFINITY : s0;
                    s1 = (32 * t + tile_row(2 * p + 1, h) >= N) ? -INFINITY : s1;
                }
                mxrow = fmaxf(mxrow, fmaxf(s0, s1));
                cw[p] = pack_h2(s0, s1);
            }
            if (SAVE) {                   // Smix as packed fp16 (the backward's S_SM slab)
                typedef __attribute__((ext_vector_type(4))) unsigned int u4;
                u4 *sp = (u4 *)(svb + SL.oSm + (size_t)w * NT * 8 * 64 * 4) + lane;
                __builtin_nontemporal_store(u4{cw[0], cw[1], cw[2], cw[3]}, &sp[(2 * t) * 64]);
                if (!(ctrim && t == NT - 1)) __builtin_nontemporal_store(u4{cw[4], cw[5], cw[6], cw[7]}, &sp[(2 * t + 1) * 64]);
            }
        }
        if (!SAVE) {
        switch (t) {
#define MOPK_CASE(T_) case T_: if (T_ < NT) { _Pragma("unroll") for (int p = 0; p < 8; ++p) crp[T_ < NT ? T_ : 0][p] = cw[p]; } break;
            MOPK_CASE(0) MOPK_CASE(1) MOPK_CASE(2) MOPK_CASE(3) MOPK_CASE(4) MOPK_CASE(5) MOPK_CASE(6)
#undef MOPK_CASE
            default: break;
        }
        }
    }
    FSTAMP();
    REFRESH();
    // ---------------- softmax over keys + P V0                         :551-554
    mxrow = fmaxf(mxrow, __shfl_xor(mxrow, 32, 64));
    float l = 0.f;
    const FaDrop drop = fa_drop(a.dropout_p, a.dropout_seed);      // attn_drop (:552): keep / (1 - p) on P; the row sum stays undropped
    const uint32_t rowh = fa_drop_row(drop, b * H + hh, qi);
    if (SAVE) {                           // Smix back from the wave's own S_SM export (L2-hot) rather than 56 registers held through the mix loop
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        const u4 *sp = (const u4 *)(svb + SL.oSm + (size_t)w * NT * 8 * 64 * 4) + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool tr1 = ctrim && t == NT - 1;
            const u4 lo = sp[(2 * t) * 64];
            u4 hi = sp[(2 * t + (tr1 ? 0 : 1)) * 64];
            if (tr1) hi = u4{0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u};       // trimmed chunk: -inf (padding keys)
#pragma unroll
            for (int p = 0; p < 4; ++p) { crp[t][p] = lo[p]; crp[t][4 + p] = hi[p]; }
        }
    }
    // P V0 tile by tile.  With the record (training) the probabilities enter the MFMAs as a bf16 value plus its bf16 remainder: the
    // backward takes delta_i = sum_j P_ij dP_ij from y_base = P v0 (dy . y_base), and a y_base built from once-rounded P disagrees with
    // the P the backward forms itself by 4e-3 per edge -- a row-wise bias in dSmix that the gate-head gradients (differences of N^2
    // terms) magnify.
    f32x16 pv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) pv[dt] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bf16x8 ph[2], pl[2];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float e0 = __expf(h2_lo(crp[t][p]) - mxrow), e1 = __expf(h2_hi(crp[t][p]) - mxrow);
            l += e0 + e1;
            if (drop.thresh) {
                e0 = fa_drop_keep(drop, rowh, 32 * t + tile_row(2 * p, h)) ? e0 * drop.inv_keep : 0.f;
                e1 = fa_drop_keep(drop, rowh, 32 * t + tile_row(2 * p + 1, h)) ? e1 * drop.inv_keep : 0.f;
            }
            const unsigned short h0 = f2bf(e0), h1 = f2bf(e1);
            ph[p >> 2][2 * (p & 3)] = (short)h0;
            ph[p >> 2][2 * (p & 3) + 1] = (short)h1;
            if (SAVE) {
                pl[p >> 2][2 * (p & 3)] = (short)f2bf(e0 - bf2f(h0));
                pl[p >> 2][2 * (p & 3) + 1] = (short)f2bf(e1 - bf2f(h1));
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (2 * t + s < 2 * NT - 1 || klast) {
                    const bf16x8 af = *(const bf16x8 *)&(VT0 + r * LDA + 8 * h)[(32 * dt) * LDA + 32 * t + 16 * s];
                    pv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ph[s], pv[dt], 0, 0, 0);
                    if (SAVE) pv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pl[s], pv[dt], 0, 0, 0);
                }
            }
    }
    l += __shfl_xor(l, 32, 64);
    const float invl = 1.f / l;
    if (SAVE && h == 0) { float *rw = (float *)(svb + SL.oRow); rw[qi] = mxrow; rw[NP + qi] = invl; }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const f32x16 acc = pv[dt];
        if (qok) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = 32 * dt + 8 * g4 + 4 * h;
                if (d0 < DK) {
                    const float4 yc = *(const float4 *)&ych[(size_t)qi * DK + d0];      // w * y_chain (own earlier store)
                    store4<IOT>(yp + d0, fmaf(acc[4 * g4], invl, yc.x), fmaf(acc[4 * g4 + 1], invl, yc.y),
                                fmaf(acc[4 * g4 + 2], invl, yc.z), fmaf(acc[4 * g4 + 3], invl, yc.w));
                    if (SAVE) *(float4 *)&((float *)(svb + SL.oYb))[(size_t)qi * DK + d0] =
                        make_float4(acc[4 * g4] * invl, acc[4 * g4 + 1] * invl, acc[4 * g4 + 2] * invl, acc[4 * g4 + 3] * invl);
                }
            }
        }
    }
    FSTAMP();
}

// ------------------------------------------------------------------ host side
// The file is compiled once per (NT, DK) with -DMOPK_INST_NT / -DMOPK_INST_DK (one translation unit
// per instantiation pair keeps the build parallel) and once with MOPK_INST_NT=0 for the dispatcher.
#define MOPK_CAT_(a, b, c, d) a##b##c##d
#define MOPK_CAT(a, b, c, d) MOPK_CAT_(a, b, c, d)
#if MOPK_INST_NT != 0
size_t MOPK_CAT(ew_fused_saved_nt, MOPK_INST_NT, _dk, MOPK_INST_DK)(const MopkEdgewiseArgs *a) {
    return fused_saved_layout<MOPK_INST_NT, MOPK_INST_DK>(a->N, a->V, a->save_for_backward != 0).stride * (size_t)a->B * a->H + 256;
}
int MOPK_CAT(ew_fused_fwd_nt, MOPK_INST_NT, _dk, MOPK_INST_DK)(const MopkEdgewiseArgs *a, hipStream_t st) {
    constexpr int NT = MOPK_INST_NT, DK = MOPK_INST_DK;
    const int lds = FusedCfg<NT, DK>::lds_bytes(a->V);
    if (lds > 160 * 1024) return MOPK_ERR_UNSUPPORTED;
    const dim3 grid(a->B * a->H), block(NT * 64);
    const bool dense = a->ext && a->ext->gate_mode == 1;          // dense gate head (no 3x3, no lens): HEAD = 1, always with the full record
    FusedDenseW dw{nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (dense) { dw.W1 = a->ext->W1; dw.b1 = a->ext->b1; dw.W2 = a->ext->W2; dw.b2 = a->ext->b2; }
    else if (a->ext && a->ext->n_extra > 0) { dw.E = a->ext->n_extra; dw.rowx = a->ext->row_extra; dw.colx = a->ext->col_extra; }
#define MOPK_LAUNCH(IOT_, SAVE_, HEAD_) do {                                                                      \
        auto kfn = ew_fused_fwd_kernel<NT, DK, IOT_, SAVE_, HEAD_>;                                               \
        static int lds_set = 0;      /* per instantiation and per process: the attribute is sticky (and may not be set during stream capture) */ \
        if (lds_set < lds) { if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MOPK_ERR_LAUNCH; lds_set = lds; } \
        hipLaunchKernelGGL(kfn, grid, block, lds, st, *a, dw);                                                    \
    } while (0)
    if (dense) { if (a->io_dtype == MOPK_BF16) MOPK_LAUNCH(unsigned short, true, 1); else MOPK_LAUNCH(float, true, 1); }
    else if (a->io_dtype == MOPK_BF16) { if (a->save_for_backward) MOPK_LAUNCH(unsigned short, true, 0); else MOPK_LAUNCH(unsigned short, false, 0); }
    else { if (a->save_for_backward) MOPK_LAUNCH(float, true, 0); else MOPK_LAUNCH(float, false, 0); }
#undef MOPK_LAUNCH
    MOPK_CHECK_LAUNCH();
    return MOPK_OK;
}
#else
#define MOPK_DECL(NT_, DK_) int ew_fused_fwd_nt##NT_##_dk##DK_(const MopkEdgewiseArgs *a, hipStream_t st); \
                            size_t ew_fused_saved_nt##NT_##_dk##DK_(const MopkEdgewiseArgs *a);
MOPK_DECL(1, 16) MOPK_DECL(1, 32) MOPK_DECL(1, 64) MOPK_DECL(2, 16) MOPK_DECL(2, 32) MOPK_DECL(2, 64)
MOPK_DECL(3, 16) MOPK_DECL(3, 32) MOPK_DECL(3, 64) MOPK_DECL(4, 16) MOPK_DECL(4, 32) MOPK_DECL(4, 64) MOPK_DECL(5, 16) MOPK_DECL(5, 32) MOPK_DECL(5, 64) MOPK_DECL(6, 16) MOPK_DECL(6, 32) MOPK_DECL(6, 64) MOPK_DECL(7, 16) MOPK_DECL(7, 32) MOPK_DECL(7, 64)
#undef MOPK_DECL
static int pick_nt(int N) { return N <= 32 ? 1 : N <= 64 ? 2 : N <= 96 ? 3 : N <= 128 ? 4 : N <= 160 ? 5 : N <= 192 ? 6 : N <= 224 ? 7 : 0; }
int ew_fused_bwd_lds_bytes(int nt, int dk, int V);
template <int NT, int DK> static int lds_fwd(int V) { return FusedCfg<NT, DK>::lds_bytes(V); }
static int ew_fused_lds_bytes(int nt, int dk, int V) {
    int f = 1 << 30;
#define MOPK_L(NT_) (dk == 16 ? lds_fwd<NT_, 16>(V) : dk == 32 ? lds_fwd<NT_, 32>(V) : lds_fwd<NT_, 64>(V))
    switch (nt) { case 1: f = MOPK_L(1); break; case 2: f = MOPK_L(2); break; case 3: f = MOPK_L(3); break; case 4: f = MOPK_L(4); break; case 5: f = MOPK_L(5); break; case 6: f = MOPK_L(6); break; case 7: f = MOPK_L(7); break; default: break; }
#undef MOPK_L
    const int bw = ew_fused_bwd_lds_bytes(nt, dk, V);
    return f > bw ? f : bw;
}
static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

int ew_fused_fwd_supported(const MopkEdgewiseArgs *a) {
    if (a->precision != MOPK_PREC_BF16) return 0;             // fused kernels are the bf16-MFMA path
    if (a->mask) return 0;                                    // attention mask (extension): generic path
    if (a->ext && a->ext->n_lens > 0) return 0;               // lens banks: generic path
    if (a->ext && a->ext->gate_mode != 0) {                   // dense gate head: forward only, without the 3x3 convolution, full record
        if (a->ext->gate_mode != 1 || a->ext->use_k3 || !a->save_for_backward) return 0;
        if (!a->ext->W1 || !a->ext->b1 || !a->ext->W2 || !a->ext->b2) return 0;
        if (a->ext->n_extra != 0) return 0;                   // extra feature channels: low-rank head only
    }
    if (a->ext && a->ext->n_extra != 0) {
        if (a->ext->n_extra < 0 || 2 * a->V + 2 + a->ext->n_extra > WST - 1 || !a->ext->row_extra || !a->ext->col_extra) return 0;
    }
    if (a->q.sv != 0 || a->k.sv != 0) return 0;               // share_qkv only (per-view K restaging not built)
    if (pick_nt(a->N) == 0) return 0;
    if (a->dk != 16 && a->dk != 32 && a->dk != 64) return 0;
    if (a->V < 2 || a->V > 8 || a->r < 1 || a->r > 4) return 0;
    if (ew_fused_lds_bytes(pick_nt(a->N), a->dk, a->V) > 160 * 1024) return 0;   // forward AND backward must fit
    const int es = a->io_dtype == MOPK_BF16 ? 2 : 4;
    const int64_t al = 16 / es;                               // vector loads: 8 (bf16) / 4 (fp32) element alignment
    const MopkView4 vs[3] = {a->v0, a->vL, a->y};
    for (const auto &v : vs) if (!aligned16(v.ptr) || v.sb % al || v.sh % al || v.sn % al) return 0;
    if (!aligned16(a->q.ptr) || a->q.sb % al || a->q.sh % al || a->q.sn % al) return 0;
    if (!aligned16(a->k.ptr) || a->k.sb % al || a->k.sh % al || a->k.sn % al) return 0;
    return 1;
}

size_t ew_fused_saved_bytes(const MopkEdgewiseArgs *a) {
#define MOPK_DKS(NT_) switch (a->dk) { case 16: return ew_fused_saved_nt##NT_##_dk16(a); case 32: return ew_fused_saved_nt##NT_##_dk32(a); default: return ew_fused_saved_nt##NT_##_dk64(a); }
    switch (pick_nt(a->N)) { case 1: MOPK_DKS(1) case 2: MOPK_DKS(2) case 3: MOPK_DKS(3) case 4: MOPK_DKS(4) case 5: MOPK_DKS(5) case 6: MOPK_DKS(6) case 7: MOPK_DKS(7) default: return 0; }
#undef MOPK_DKS
}
int ew_fused_fwd(const MopkEdgewiseArgs *a, hipStream_t st) {
    if (!ew_fused_fwd_supported(a)) return MOPK_ERR_UNSUPPORTED;
#define MOPK_DK(NT_)                                                         \
    switch (a->dk) {                                                         \
        case 16: return ew_fused_fwd_nt##NT_##_dk16(a, st);                  \
        case 32: return ew_fused_fwd_nt##NT_##_dk32(a, st);                  \
        default: return ew_fused_fwd_nt##NT_##_dk64(a, st);                  \
    }
    switch (pick_nt(a->N)) {
        case 1: MOPK_DK(1)
        case 2: MOPK_DK(2)
        case 3: MOPK_DK(3)
        case 4: MOPK_DK(4)
        case 5: MOPK_DK(5)
        case 6: MOPK_DK(6)
        default: MOPK_DK(7)
    }
#undef MOPK_DK
    return MOPK_ERR_UNSUPPORTED;
}
#endif

}  // namespace mopk
