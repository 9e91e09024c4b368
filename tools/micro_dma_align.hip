// Does an LDS-DMA (global_load_lds_dwordx4, 16 bytes per lane, lane-linear) accept an LDS base that is 8 (not 16) bytes aligned?
// (round 5 probe for the fp32x weight-gradient tiles: rows 8-15 of every 16 shifted by 8 bytes would put their hi halves on the other
// two banks of every 16-byte slot and make the hi-only / lo-only transposed reads conflict-free.)
//   hipcc --offload-arch=gfx950 -O2 tools/micro_dma_align.hip -o gpurun_variants/micro_dma_align && gpurun_variants/micro_dma_align
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint32_t* src, uint32_t* dst, int shift) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds) + shift;
    const void* g = src + threadIdx.x * 4;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) dst[i] = lds[i];
}
int main() {
    uint32_t h[256], *s, *d, o[1024];
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    hipMalloc(&s, sizeof h); hipMalloc(&d, sizeof o);
    hipMemcpy(s, h, sizeof h, hipMemcpyHostToDevice);
    for (int shift : {0, 8, 4, 16}) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, s, d, shift);
        hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int i = 0; i < 256; ++i) ok &= (o[i + shift / 4] == h[i]);
        printf("shift %2d bytes: %s  (words %d..%d: %x %x %x %x | %x %x)\n", shift, ok ? "lane-linear image at base + shift" : "NOT as expected", 0, 5, o[0], o[1], o[2], o[3], o[4], o[5]);
    }
    return 0;
}
