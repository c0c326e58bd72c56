// Shared device/host helpers for libmopk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mopk.h"

namespace mopk {

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator

constexpr int WAVE = 64;

__device__ __forceinline__ unsigned short f2bf(float f) {
    // round-to-nearest-even; plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 at -O3)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) {
    return __builtin_bit_cast(float, ((unsigned int)u) << 16);
}
// pack two floats into one dword of 2 x bf16 (lo = a, hi = b)
__device__ __forceinline__ unsigned int pack_bf16(float a, float b) {
    return (unsigned int)f2bf(a) | ((unsigned int)f2bf(b) << 16);
}

template <typename T> __device__ __forceinline__ float ld_as_f32(const T *p);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld_as_f32<unsigned short>(const unsigned short *p) {
    return bf2f(*p);
}
template <typename T> __device__ __forceinline__ void st_from_f32(T *p, float v);
template <> __device__ __forceinline__ void st_from_f32<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_from_f32<unsigned short>(unsigned short *p, float v) {
    *p = f2bf(v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- attention dropout: counter-based keep mask, a pure function of (seed, b*H+h, query, key) -- evaluated again in the backward
__host__ __device__ inline uint32_t fa_hash(uint32_t x) {          // "lowbias32" integer finalizer
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct FaDrop { uint32_t thresh, seed_lo, seed_hi; float inv_keep; };      // thresh == 0: dropout off
__host__ __device__ inline FaDrop fa_drop(float p, uint64_t seed) {
    FaDrop d{0u, (uint32_t)seed, (uint32_t)(seed >> 32), 1.f};
    if (p > 0.f) {
        const double t = (double)p * 4294967296.0;
        d.thresh = t >= 4294967295.0 ? 0xffffffffu : (t < 1.0 ? 1u : (uint32_t)t);
        d.inv_keep = 1.f / (1.f - p);
    }
    return d;
}
__host__ __device__ inline uint32_t fa_drop_row(const FaDrop &d, int bh, int i) {
    return fa_hash((uint32_t)i * 0x9E3779B1u + (uint32_t)bh * 0x85EBCA77u + d.seed_hi) ^ d.seed_lo;
}
__host__ __device__ inline bool fa_drop_keep(const FaDrop &d, uint32_t rowh, int j) {
    return fa_hash(rowh ^ ((uint32_t)j * 0xC2B2AE3Du)) >= d.thresh;
}

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// bump allocator over a caller-provided buffer (256-byte aligned carve-outs)
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *p) : base((char *)p) {}
    template <typename T> T *take(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        T *r = base ? (T *)(base + off) : nullptr;
        off += bytes;
        return r;
    }
};

#define MOPK_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return MOPK_ERR_LAUNCH; \
    } while (0)

}  // namespace mopk
