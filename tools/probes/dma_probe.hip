// dev probe (GPU box): global_load_lds_dwordx4 copies 1-KiB pieces global -> LDS in lane order; partial EXEC for a 512-B tail
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;
__device__ __forceinline__ void dma16(const void *g, unsigned lds_off) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(uintptr_t)lds_off, 16, 0, 0);
}
__global__ void k(const unsigned char *src, unsigned char *dst, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned base = (unsigned)(uintptr_t)smem;
    const int npc = nbytes / 1024, rem = nbytes % 1024;
    for (int pc = w; pc < npc; pc += nw) dma16(src + pc * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(base + pc * 1024));
    if (rem && w == npc % nw) { if (lane * 16 < rem) dma16(src + npc * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(base + npc * 1024)); }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = threadIdx.x; i < nbytes / 16; i += blockDim.x) ((uint4 *)dst)[i] = ((const uint4 *)smem)[i];
}
int main() {
    const int nbytes = 224 * 232 * 2;
    std::vector<unsigned char> h(nbytes), o(nbytes);
    for (int i = 0; i < nbytes; ++i) h[i] = (unsigned char)((i * 131 + (i >> 8) * 7) & 255);
    unsigned char *d, *e;
    hipMalloc(&d, nbytes + 1024); hipMalloc(&e, nbytes);
    hipMemcpy(d, h.data(), nbytes, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024);
    hipLaunchKernelGGL(k, dim3(4), dim3(448), 110 * 1024, 0, d, e, nbytes);
    hipDeviceSynchronize();
    hipMemcpy(o.data(), e, nbytes, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < nbytes; ++i) bad += o[i] != h[i];
    printf("dma probe: %d bytes, mismatches %d (%s)\n", nbytes, bad, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
