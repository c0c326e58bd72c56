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
