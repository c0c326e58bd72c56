// Shared definitions of the fused Edgewise backward launches (edgewise_fused_bwd.hip): workspace carve-up, slab ids of the hand-off region, packed-tile helpers.
#pragma once
#include "fused_common.h"

namespace mopk {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

enum { PH_A = 0, PH_B = 1, PH_C = 2 };       // the three launches of the backward (see the header comment)

struct BwdWs {
    unsigned char *base;     // per-workgroup scratch (persistent workgroups: indexed by blockIdx)
    size_t stride;           // bytes per workgroup
    size_t oKT, oQT, oDYT, oV0s, oVLs, oDbp, oDW, oStamp;
    unsigned char *xbase;    // per-(b,h) hand-off region between the three launches
    size_t xstride;
    size_t xSlots, xDmean;
};
// slab ids.  S_CF .. S_L live in the forward's `saved` record; X_* in the hand-off region
enum { S_CF = 0, S_CB, S_SM, S_L, X_C3, X_DIR };   // X_DIR .. X_DIR+V-1 (direct score gradients per view), X_DR(V)+v: D_v of the -> chain, X_DL(V)+m: D'_m of the <- chain
__host__ __device__ constexpr int X_DR(int V) { return X_DIR + V; }
__host__ __device__ constexpr int X_DL(int V) { return X_DIR + 2 * V; }
// dense gate head only: X_DT(V)+v = gradient of the S_v^T feature channel (launch C adds it transposed), X_CL(V) = gradient of log C<-
__host__ __device__ constexpr int X_DT(int V) { return X_DIR + 3 * V; }
__host__ __device__ constexpr int X_CL(int V) { return X_DIR + 4 * V; }
__host__ __device__ constexpr int X_COUNT(int V, bool dense = false) { return dense ? 2 + 4 * V : 1 + 3 * V; }      // slabs in the hand-off region (ids X_C3 ..)

template <int NT, int DK>
struct BwdCfg {
    using F = FusedCfg<NT, DK>;
    static constexpr int NP = F::NP, LDA = F::LDA, DP = F::DP, DT = F::DT;
    static constexpr size_t MAT = (size_t)NP * LDA * 2;
    static constexpr size_t SLOT = (size_t)NT * 8 * 64 * 4;                // one packed slab of one wave
    static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }
    // base: nwg per-workgroup scratch regions, then nbh hand-off regions
    static BwdWs carve(void *base, int V, int nwg, int nbh, bool dense = false) {
        BwdWs w{};
        size_t o = 0;
        w.base = (unsigned char *)base;
        w.oStamp = o; o += 512;                                    // diagnostic s_memtime stamps (MOPK_STAMPS builds): first, so tools find them at the workspace base
        w.oKT = o; o += a256((size_t)DP * LDA * 2);
        w.oQT = o; o += a256((size_t)DP * LDA * 2);
        w.oDYT = o; o += a256((size_t)DP * LDA * 2);
        w.oV0s = o; o += a256((size_t)NP * DK * 2);
        w.oVLs = o; o += a256((size_t)NP * DK * 2);
        w.oDbp = o; o += a256((size_t)NT * 16 * NP * 4);
        w.oDW = o; o += a256((size_t)2 * 16 * WST * 4);
        w.stride = a256(o);
        w.xbase = w.base + w.stride * (size_t)nwg;
        size_t x = 0;
        w.xSlots = x; x += a256((size_t)X_COUNT(V, dense) * NT * SLOT);
        w.xDmean = x; x += a256((size_t)(2 * V + 4) * NP * 4);
        w.xstride = a256(x);
        (void)nbh;
        return w;
    }
    static size_t total_bytes(int V, int nwg, int nbh, bool dense = false) {
        const BwdWs w = carve(nullptr, V, nwg, nbh, dense);
        return w.stride * (size_t)nwg + w.xstride * (size_t)nbh;
    }
    // LDS: R region | Ksm | floats
    static constexpr int GATE_BYTES = 4 * NP * BTS * 2 + 2 * 32 * LDA * 2 + NT * 32 * 40 * 2;   // bT | bmat | amat | tbuf
    static constexpr int WSM_FLOATS = 2 * 16 * WST;                                               // gate-head weights + bias, row | col side
    static constexpr int R_BYTES = imax(imax(NP * LDA * 2, 3 * DP * LDA * 2), GATE_BYTES + WSM_FLOATS * 4);
    static constexpr int K_BYTES = F::K_BYTES;
    static __host__ __device__ constexpr int small_floats(int V) {
        // sqk[8][DK] qbar kbar vs0 vsL | rCr rCl cCr cCl | colpart[NT][NP] | rS cS [V][NP] (later: dmean[2V+4][NP]) | misc
        return 16 * DK + 4 * DK + 4 * NP + NT * NP + imax(2 * V * NP, (2 * V + 4) * NP - NT * NP) + 2 * NT * DK + 16;
    }
    static __host__ __device__ constexpr int lds_bytes(int V) { return R_BYTES + K_BYTES + 4 * small_floats(V); }
};

__device__ __forceinline__ u32x4 as_u4(bf16x8 v) { return __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ bf16x8 as_b8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ void pack_tile_bf(bf16x8 &lo, bf16x8 &hi, const f32x16 &x) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (short)f2bf(x[j]); hi[j] = (short)f2bf(x[8 + j]); }
}
__device__ __forceinline__ f32x16 unpack_tile_bf(bf16x8 lo, bf16x8 hi) {
    f32x16 x;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = bf2f((unsigned short)lo[j]); x[8 + j] = bf2f((unsigned short)hi[j]); }
    return x;
}
__device__ __forceinline__ void pack_tile_h(u32x4 &lo, u32x4 &hi, const f32x16 &x) {
#pragma unroll
    for (int p = 0; p < 4; ++p) { lo[p] = pack_h2(x[2 * p], x[2 * p + 1]); hi[p] = pack_h2(x[8 + 2 * p], x[8 + 2 * p + 1]); }
}
__device__ __forceinline__ f32x16 unpack_tile_h(u32x4 lo, u32x4 hi) {
    f32x16 x;
#pragma unroll
    for (int p = 0; p < 4; ++p) { x[2 * p] = h2_lo(lo[p]); x[2 * p + 1] = h2_hi(lo[p]); x[8 + 2 * p] = h2_lo(hi[p]); x[8 + 2 * p + 1] = h2_hi(hi[p]); }
    return x;
}
// cond ? x : 0 as a bit mask.  Written as a ternary, hipcc sinks the loads and arithmetic of x under the condition and emits
// one exec-masked branch (with a full s_waitcnt) per element of a tile; the mask form stays straight-line code.
__device__ __forceinline__ float keep_if(bool cond, float x) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, x) & (cond ? 0xffffffffu : 0u));
}
__device__ __forceinline__ f32x16 zero16() { return f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }

}  // namespace mopk
