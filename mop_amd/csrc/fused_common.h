// Shared device helpers of the fused gfx950 Edgewise kernels (forward + backward).
#pragma once
#include <type_traits>
#include <cstdint>

#include "common.h"

namespace mopk {

constexpr float EPSC = 1e-6f;   // attention_variants.py:516

// compile-time loop: f(integral_constant<int, I>) for I in [B, E) -- the index is a constant expression inside the body (asm "i" operands)
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
constexpr int BTS = 24;         // bT row stride (ushorts): 16 k-slots + 8 pad (48 B, 16-B aligned)
constexpr int WST = 27;         // row stride (floats) of the staged low-rank gate-head weights: up to 26 input channels + the bias in slot WST - 1

__host__ __device__ constexpr int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int kperm16(int k) { return (k & 3) | ((k & 4) << 1) | ((k & 8) >> 1); }
__device__ __forceinline__ int tile_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

template <int NT, int DK>
struct FusedCfg {
    static constexpr int NP = NT * 32, LDA = NP + 8, LDK = DK + 8, KS = DK / 16;
    static constexpr int DT = DK >= 32 ? DK / 32 : 1, DP = DT * 32;
    // R region: during the chains a ring of RING image parts (32 keys x LDA each) + the q rows [NP][LDK]; afterwards V0^T | VL^T | bT
    static constexpr int RING = NT >= 2 ? 4 : 2, PART = 32 * LDA;    // ring slots (a power of two), ushorts per slot
    static constexpr int R_BYTES = imax(RING * PART * 2 + NP * LDK * 2, 2 * DP * LDA * 2 + 4 * NP * BTS * 2);
    static constexpr int K_BYTES = NP * LDK * 2;
    // fp32 scratch (floats): sqk sqk2 [8][DK] qbar kbar vs0 vsL [DK] | rCr rCl cCr cCl [NP] | colpart[NT][NP] | rS cS cst [V][NP] | wsig
    // gate-head weights [2][16][WST] are staged over colpart when it is large enough (NT = 7), else in their own slot
    static constexpr int WSM_FLOATS = 2 * 16 * WST, WSM_EXTRA = NT * NP >= WSM_FLOATS ? 0 : WSM_FLOATS;
    static __host__ __device__ constexpr int small_floats(int V) {
        return 16 * DK + 4 * DK + 4 * NP + NT * NP + 2 * V * NP + 8 + WSM_EXTRA;
    }
    static __host__ __device__ constexpr int lds_bytes(int V) { return R_BYTES + K_BYTES + 4 * small_floats(V); }
};

__device__ __forceinline__ float half_sum32(float v) {   // sum over the 32 lanes of this lane's half
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned int pack_h2(float a, float b) {
    _Float16 x = (_Float16)a, y = (_Float16)b;
    return (unsigned int)__builtin_bit_cast(unsigned short, x) | ((unsigned int)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float h2_lo(unsigned int u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffff)); }
__device__ __forceinline__ float h2_hi(unsigned int u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }

template <typename IOT> __device__ __forceinline__ bf16x8 load8_bf16(const IOT *p);
template <> __device__ __forceinline__ bf16x8 load8_bf16<unsigned short>(const unsigned short *p) {
    return *(const bf16x8 *)p;
}
template <> __device__ __forceinline__ bf16x8 load8_bf16<float>(const float *p) {
    const float4 a = *(const float4 *)p, b = *(const float4 *)(p + 4);
    bf16x8 r;
    r[0] = f2bf(a.x); r[1] = f2bf(a.y); r[2] = f2bf(a.z); r[3] = f2bf(a.w);
    r[4] = f2bf(b.x); r[5] = f2bf(b.y); r[6] = f2bf(b.z); r[7] = f2bf(b.w);
    return r;
}
// eight consecutive elements as floats (16-byte aligned): one 16-byte read (bf16) / two (fp32), exact
__device__ __forceinline__ void ld8_as_f32(float (&v)[8], const unsigned short *p) {
    const bf16x8 x = *(const bf16x8 *)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f((unsigned short)x[e]);
}
__device__ __forceinline__ void ld8_as_f32(float (&v)[8], const float *p) {
    const float4 a = *(const float4 *)p, b = *(const float4 *)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
// four consecutive elements as floats (8-byte aligned for bf16, 16-byte for fp32)
__device__ __forceinline__ void ld4_as_f32(float (&v)[4], const unsigned short *p) {
    const uint2 x = *(const uint2 *)p;
    v[0] = __builtin_bit_cast(float, x.x << 16); v[1] = __builtin_bit_cast(float, x.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, x.y << 16); v[3] = __builtin_bit_cast(float, x.y & 0xffff0000u);
}
__device__ __forceinline__ void ld4_as_f32(float (&v)[4], const float *p) {
    const float4 a = *(const float4 *)p;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <typename IOT> __device__ __forceinline__ void store4(IOT *p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float *p, float a, float b, float c, float d) {
    *(float4 *)p = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<unsigned short>(unsigned short *p, float a, float b, float c, float d) {
    *(uint2 *)p = make_uint2(pack_bf16(a, b), pack_bf16(c, d));
}

// B-operand fragments (k order of the accumulator) of an X-layout slab
template <int NT>
__device__ __forceinline__ void pack_slab(bf16x8 (&Xp)[NT][2], const f32x16 (&X)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) Xp[t][s][j] = (short)f2bf(X[t][8 * s + j]);
}

// column means of every view's score map with KS MFMAs per wave:  cS_v[j] = K[j,:] . (sqk_v * qbar)   (mean_i S_v[i,j])
// lanes 0-7 carry the bf16 "hi" part of u_v = sqk_v * qbar for view v = lane, lanes 8-15 the "lo" remainder (fp32-accurate
// product); wave w covers keys [32w, 32w+32).  Requires V <= 8.
template <int NT, int DK>
__device__ __forceinline__ void col_means_mfma(float *cS, const unsigned short *Ksm, const float *sqk, const float *qbar, int V,
                                               int w, int r, int h) {
    constexpr int NP = NT * 32, LDK = DK + 8;
    const int vv = r & 7;
    const bool act = r < 16 && vv < V, lo_part = (r >> 3) == 1;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < DK / 16; ++s) {
        bf16x8 bf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = 16 * s + 8 * h + e;
            const float u = act ? sqk[vv * DK + d] * qbar[d] : 0.f;
            const unsigned short hi = f2bf(u);
            bf[e] = (short)(lo_part ? f2bf(u - bf2f(hi)) : hi);
        }
        const bf16x8 af = *(const bf16x8 *)&Ksm[(32 * w + r) * LDK + 16 * s + 8 * h];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const float tot = acc[g] + __shfl_xor(acc[g], 8, 64);
        if (r < 8 && r < V) cS[r * NP + 32 * w + tile_row(g, h)] = tot;
    }
}
// kbar[d] = mean_j K[j,d]: all threads take a strided share of the rows, partials in `part` ([threads/DK][DK])
template <int NT, int DK>
__device__ __forceinline__ void key_mean_partials(float *part, const unsigned short *Ksm, int N, int tid) {
    constexpr int NPART = NT * 64 / DK, LDK = DK + 8;
    const int d = tid % DK, p = tid / DK;
    if (p < NPART) {
        float s = 0.f;
        for (int j = p; j < N; j += NPART) s += bf2f(Ksm[j * LDK + d]);
        part[p * DK + d] = s;
    }
}

// ---- layout of the fused path's `saved` buffer, one record per (b,h).  Inference forward writes only `ych`
// (w * y_chain, fp32); the training forward additionally exports what the backward would otherwise recompute:
// the prefix products T_m / U_m as AT images, the final products C-> / C<- as packed per-wave slabs, the per-view
// softmax constants, the four log-mean vectors, and the mix phase's Smix / L slabs with the final softmax row statistics.
struct FusedSavedLayout {
    size_t oYch, oT, oU, oCF, oCB, oCst, oMeans, oSm, oL, oRow, oYb, stride;
};
template <int NT, int DK>
__host__ __device__ inline FusedSavedLayout fused_saved_layout(int N, int V, bool full) {
    constexpr size_t NP = NT * 32, LDA = NP + 8, MAT = NP * LDA * 2, WSLOT = (size_t)NT * 8 * 64 * 4;
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    FusedSavedLayout L{};
    size_t o = 0;
    L.oYch = o; o += a256((size_t)N * DK * 4);
    if (full) {
        L.oT = o; o += a256((size_t)(V - 1) * MAT);
        L.oU = o; o += a256((size_t)(V - 1) * MAT);
        L.oCF = o; o += a256(NT * WSLOT);
        L.oCB = o; o += a256(NT * WSLOT);
        L.oCst = o; o += a256((size_t)V * NP * 4);
        L.oMeans = o; o += a256((size_t)4 * NP * 4);
        // mixed logits Smix as a packed fp16 per-wave slab, L = lse_v S_v - S_0 (x log2 e) as an fp32 slab (it multiplies the OR gate's
        // gradient edge by edge: rounded to fp16 it alone put 2-5 % on the gate-head weight gradients), final-softmax row max / 1 / row sum:
        // with these the backward has no mix-recompute pass
        L.oSm = o; o += a256(NT * WSLOT);
        L.oL = o; o += a256(2 * NT * WSLOT);
        L.oRow = o; o += a256((size_t)2 * NP * 4);
        L.oYb = o; o += a256((size_t)N * DK * 4);      // y_base = P v0 (fp32): delta_i = sum_j P_ij dP_ij = dy_i . y_base_i
    }
    L.stride = o;
    return L;
}

// ---- select / update one 32x32 tile (16 regs) or 8 packed dwords of a register array with a UNIFORM runtime
// index: keeps the array in VGPRs while letting the tile loop stay rolled (an unrolled loop makes hipcc
// overlap the tiles' live ranges and spill).
template <int NT> __device__ __forceinline__ f32x16 tile_get(const f32x16 (&X)[NT], int t) {
    f32x16 r = X[0];
    switch (t) {   // t is wave-uniform: scalar branches, no per-lane select chains
#define MOPK_TG(K_) case K_: if (K_ < NT) r = X[K_ < NT ? K_ : 0]; break;
        MOPK_TG(1) MOPK_TG(2) MOPK_TG(3) MOPK_TG(4) MOPK_TG(5) MOPK_TG(6) MOPK_TG(7)
#undef MOPK_TG
        default: break;
    }
    return r;
}
template <int NT> __device__ __forceinline__ void tile_set(f32x16 (&X)[NT], int t, const f32x16 &v) {
    switch (t) {
#define MOPK_TS(K_) case K_: if (K_ < NT) X[K_ < NT ? K_ : 0] = v; break;
        MOPK_TS(0) MOPK_TS(1) MOPK_TS(2) MOPK_TS(3) MOPK_TS(4) MOPK_TS(5) MOPK_TS(6) MOPK_TS(7)
#undef MOPK_TS
        default: break;
    }
}


// Extra feature channels (FusedDenseW::rowx / colx): XU values of one token are requested together, so a pass over E channels exposes
// ceil(E / XU) memory round trips instead of E
constexpr int XU = 8;
__device__ __forceinline__ void load_extra(float (&v)[XU], const float *x, int N, int n, int e0, int E, bool ok) {
#pragma unroll
    for (int u = 0; u < XU; ++u) v[u] = (ok && e0 + u < E) ? x[(e0 + u) * N + n] : 0.f;
#pragma unroll
    for (int u = 0; u < XU; ++u) asm volatile("" : "+v"(v[u]));      // keeps the loads ahead of the first use (one wait for the batch)
}

// device pointers taken from MopkEdgewiseExt (a host struct): the dense gate head's conv1 (16, C) / (16), conv2 (4, 16) / (4); and the
// low-rank head's E extra feature channels as their row / column means (B,H,E,N) with the gradients' destinations
struct FusedDenseW { const float *W1, *b1, *W2, *b2; int E; const float *rowx, *colx; float *drowx, *dcolx; };

// ---- stream GEMM over an LDS operand image -------------------------------------------------------------------------------------
// For every output tile `to`:  acc = init(to);  acc += Am[32 to + r][:] . Bf;  epi(to, acc).   `am_lane` = Am + r * LDA + 8 * h.
// The NT x 2NT A-fragment reads form ONE stream that runs PF fragments ahead of the MFMAs.  Reads and their waits are inline asm:
// hipcc sinks each compiler-visible read next to its MFMA and waits lgkmcnt(0) behind it (read -> wait -> MFMA per k-step); asm
// statements keep their order and LDS returns in order, so before fragment f only the younger reads may be outstanding.  LDS
// traffic of `init` / `epi` only makes a counted wait conservative (it is younger than the reads that wait needs).
// `klast` == false skips the last 16-wide k-step of every tile: with N <= NP - 16 that fragment is all padding (N = 197: the
// contraction runs over 208 instead of 224).
template <int NT, typename Init, typename Epi>
__device__ __forceinline__ void gemm_stream_epi(const unsigned short *am_lane, const bf16x8 (&Bf)[NT][2], bool klast, Init &&init, Epi &&epi) {
    constexpr int NP = NT * 32, LDA = NP + 8;
    constexpr int NK = 2 * NT, NF = NT * NK, PF = 6;
    const unsigned abase = (unsigned)(uintptr_t)am_lane;
    unsigned rowb[NT];                                  // + 32 to rows (byte offsets exceed the 16-bit immediate)
#pragma unroll
    for (int to = 0; to < NT; ++to) rowb[to] = abase + (unsigned)(32 * to * LDA * 2);
    bf16x8 ring[PF];
    static_for<0, (PF < NF ? PF : NF)>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        const unsigned rb = rowb[f / NK];
        bf16x8 tmp;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tmp) : "v"(rb), "i"(32 * (f % NK)));
        ring[f] = tmp;
    });
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    static_for<0, NF>([&](auto fc) {
        constexpr int f = decltype(fc)::value, to = f / NK, k = f % NK;
        constexpr int pend = (NF - 1 - f) < (PF - 1) ? (NF - 1 - f) : (PF - 1);
        if (k == 0) acc = init(to);
        bf16x8 af = ring[f % PF];
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af) : "i"(pend));
        if (k < NK - 1 || klast) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, Bf[k >> 1][k & 1], acc, 0, 0, 0);
        if constexpr (f + PF < NF) {
            constexpr int fn = f + PF;
            const unsigned rb = rowb[fn / NK];
            bf16x8 tmp;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tmp) : "v"(rb), "i"(32 * (fn % NK)));
            ring[fn % PF] = tmp;
        }
        if (k == NK - 1) epi(to, acc);
    });
}
// B fragments of the 32 x 32 identity in the accumulator's k order: D = X . I returns a packed tile X (lane = row, registers = columns)
// transposed (lane = column, registers = rows) -- two MFMAs per tile, exact (every entry is one bf16 value times 1.0)
__device__ __forceinline__ void identity_frags(bf16x8 &idl, bf16x8 &idh, int r, int h) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        idl[e] = (short)(r == tile_row(e, h) ? 0x3f80 : 0);          // bf16(1.0)
        idh[e] = (short)(r == 16 + tile_row(e, h) ? 0x3f80 : 0);
    }
}
}  // namespace mopk
