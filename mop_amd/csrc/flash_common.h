// Shared device helpers of the fused (flash-style) attention kernels: sdpa_flash.hip, quartet_flash.hip.
#pragma once
#include "fused_common.h"

namespace mopk {
namespace {   // internal linkage: each translation unit gets its own copies
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
// The same staging split in two so the global loads of tile t+1 fly while tile t is computed: fa_fetch (global -> registers),
// fa_put (registers -> LDS images).  The registers hold MFMA A fragments: wave w owns the 32-token half (w & 1) and the 32-feature
// tile (w >> 1) of the 64-token tile, lane (r, h) the two 16-byte chunks [32 dt + 16 kk + 8 h, +8) of token row 32 s2 + r.  The
// row image takes them as they are; the transposed image is the fragment pair times the identity on the matrix core (exact): the
// product comes back with lane = feature, registers = tokens in accumulator order -- which IS the k-permuted column order the
// consumers read -- so it is two 16-byte LDS stores per lane, not sixteen 2-byte scatter stores with 8-way bank conflicts.
template <int DK> struct FaTile { bf16x8 v[2]; };
template <int DK, typename IOT>
__device__ __forceinline__ void fa_fetch(FaTile<DK> &f, const IOT *base, int64_t sn, int t0, int N, float scale, int tid) {
    const int w = tid >> 6, r = tid & 31, h = (tid >> 5) & 1, s2 = w & 1, dt = w >> 1, row = t0 + 32 * s2 + r;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (dt < DK / 32 && row < N) v = load8_bf16<IOT>(base + (int64_t)row * sn + 32 * dt + 16 * kk + 8 * h);
        if (scale != 1.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (short)f2bf(bf2f((unsigned short)v[e]) * scale);
        }
        f.v[kk] = v;
    }
}
template <int DK, bool ROWS, bool COLS>
__device__ __forceinline__ void fa_put(unsigned short *rows, unsigned short *cols, const FaTile<DK> &f, int tid) {
    constexpr int LDK = DK + 8;
    const int w = tid >> 6, r = tid & 31, h = (tid >> 5) & 1, s2 = w & 1, dt = w >> 1;
    if (dt >= DK / 32) return;            // dk = 32: two waves cover the tile
    if (ROWS) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) *(bf16x8 *)&rows[(32 * s2 + r) * LDK + 32 * dt + 16 * kk + 8 * h] = f.v[kk];
    }
    if (COLS) {
        bf16x8 id0, id1;                  // B fragments of the 32 x 32 identity, k-steps 0 and 1: element e of lane (n, h) is [16 kk + 8 h + e == n]
#pragma unroll
        for (int e = 0; e < 8; ++e) { id0[e] = (short)(r == 8 * h + e ? 0x3f80 : 0); id1[e] = (short)(r == 16 + 8 * h + e ? 0x3f80 : 0); }
        f32x16 tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.v[0], id0, fa_zero(), 0, 0, 0);
        tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.v[1], id1, tr, 0, 0, 0);
        bf16x8 lo, hi;
        fa_pack(lo, hi, tr);              // exact: every entry is one bf16 value times 1.0
        unsigned short *dst = &cols[(32 * dt + r) * FA_LDT + 32 * s2 + 8 * h];
        *(bf16x8 *)dst = lo;
        *(bf16x8 *)(dst + 16) = hi;
    }
}
// stage a tile of 64 tokens in one go (no prefetch): row-major image rows[tok][DK+8] and/or transposed image cols[d][perm(tok)]
template <int DK, typename IOT, bool ROWS, bool COLS>
__device__ __forceinline__ void fa_stage(unsigned short *rows, unsigned short *cols, const IOT *base, int64_t sn, int t0, int N,
                                         float scale, int tid) {
    FaTile<DK> f;
    fa_fetch<DK, IOT>(f, base, sn, t0, N, scale, tid);
    fa_put<DK, ROWS, COLS>(rows, cols, f, tid);
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
// workgroup id -> (block along the token axis, (b,h)).  The grid is 1-D; ids are dealt round-robin over the 8 XCDs (each with
// its own L2), so the bijective swizzle of cdna_hip_programming.md gives every XCD a CONTIGUOUS chunk of the (bh-major) tile
// list: all token blocks of one (b,h) -- which re-read the same K/V (or Q/dY) -- run behind the same L2.  Speed only.
__device__ __forceinline__ void fa_block_id(int nq, int &qb, int &bh) {
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig % 8, q = nwg / 8, rr = nwg % 8;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + orig / 8;
    qb = swz % nq; bh = swz / nq;
}
}  // namespace
}  // namespace mopk
