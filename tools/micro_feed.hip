// Micro-benchmark (debug aid): how much L2 -> LDS operand traffic an MFMA-paced block can take before the LDS-DMA stream, not the matrix
// pipe, sets its pace -- the question behind "would a Winograd-domain 3x3 kernel be faster" (DESIGN.md section 9b).
//
// One 512-thread block per CU (the shape of conv_nt4_kernel).  Per step every wave issues D LDS-DMA pieces (1 KB each, from a 4 MB
// array every block re-reads: L2-resident like a layer's weights) two steps ahead into a wave-private 3-slot ring, waits for the
// step's own pieces with a counted vmcnt, meets the block at ONE barrier (the real kernels share their tiles), reads R 16-byte
// fragments of REAL landed data and feeds 32 v_mfma_f32_16x16x32_f16.
//   direct 3x3 conv today (per wave and tap = 32 MFMAs): D = 3, R = 16
//   F(2x2,3x3) at the accumulator-limited 8x16-pixel x 128-channel tile: D = 17 (128 KB of transformed weights + 10 KB of halo per
//   32-channel step over 8 waves), R = 32, plus 48 packed adds for B^T d B (V = 1)
// Prints ns per step and the matrix rate per configuration; MFMA-only (D = 0, R = 0) is the ceiling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(gsrc) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D, int R, int V>
__global__ __launch_bounds__(512, 1) void feed_kernel(const char* __restrict__ src, long src_bytes, float* __restrict__ sink, int steps) {
    // LDS footprint: a slot keeps at most 4 KB per wave -- pieces beyond that land on top of earlier ones (the stream's traffic is what
    // is measured; a real Winograd block would hold one single-buffered step of ~138 KB)
    constexpr int DD = D > 4 ? 4 : (D > 0 ? D : 1);
    extern __shared__ __attribute__((aligned(16))) char lds[];            // [8 waves][3 slots][DD KB]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = lds + wave * 3 * DD * 1024;
    const uint32_t mask = (uint32_t)src_bytes - 1;                         // src_bytes is a power of two
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    h16x8 a0;
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = (h16)(0.01f * ((lane * 7 + e * 3) % 29 - 14));
    uint32_t pos = ((blockIdx.x * 8 + wave) * 65536u) & mask;
    auto issue = [&](int slot) {
#pragma unroll
        for (int d = 0; d < D; ++d) glds16(src + ((pos + d * 1024 + lane * 16) & mask), ring + (slot * DD + d % DD) * 1024);
        pos = (pos + 37 * 1024) & mask;
    };
    if (D > 0) { issue(0); issue(1); }
    h16x8 frag[R > 0 ? R : 1];
    for (int s = 0; s < steps; ++s) {
        const int slot = s % 3;
        if (D > 0) {
            issue((s + 2) % 3);
            wait_vm<2 * D>();                                              // this step's pieces landed, two younger groups in flight
        }
        __builtin_amdgcn_s_barrier();
        const char* cur = ring + slot * DD * 1024;
#pragma unroll
        for (int r = 0; r < R; ++r) frag[r] = *reinterpret_cast<const h16x8*>(cur + ((r * 1024 + lane * 16) % (DD * 1024)));
        if (V) {                                                           // the input transform's packed adds on the raw fragments
#pragma unroll
            for (int v = 0; v < 12; ++v) frag[v % (R > 0 ? R : 1)] = frag[v % (R > 0 ? R : 1)] + frag[(v + 1) % (R > 0 ? R : 1)];     // 12 x 4 v_pk_add_f16
        }
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const h16x8 b = R > 0 ? frag[m % R] : a0;
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a0), "v"(b));
        }
    }
    wait_vm<0>();
    float f = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 512 + threadIdx.x] = f;
}

template <int D, int R, int V> void run(const char* name, const char* src, long bytes, float* sink) {
    const int steps = 4000, ncu = 256;
    const size_t shm = 8 * 3 * (D > 4 ? 4 : (D > 0 ? D : 1)) * 1024;
    hipFuncSetAttribute((const void*)feed_kernel<D, R, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    feed_kernel<D, R, V><<<ncu, 512, shm>>>(src, bytes, sink, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        feed_kernel<D, R, V><<<ncu, 512, shm>>>(src, bytes, sink, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns_step = best * 1e6 / steps;
    const double tf = (double)ncu * 8 * 32 * 16384.0 * steps / (best * 1e-3) / 1e12;
    const double gbs = (double)ncu * 8 * D * 1024.0 * steps / (best * 1e-3) / 1e9;
    printf("| %-52s | D %2d | R %2d | %7.1f ns/step | %7.1f TF/s | L2->LDS %7.0f GB/s |%s\n", name, D, R, ns_step, tf, gbs,
           hipGetLastError() == hipSuccess ? "" : " LAUNCH ERROR");
}

int main() {
    const long bytes = 4l << 20;
    char* src; float* sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 256 * 512 * 4);
    // real data: small normal-ish fp16 values
    h16* h = (h16*)malloc(bytes);
    uint32_t s = 1234567u;
    for (long i = 0; i < bytes / 2; ++i) { s = s * 1664525u + 1013904223u; h[i] = (h16)(((int)(s >> 16) & 0x3ff) * (1.0f / 512.0f) - 1.0f); }
    hipMemcpy(src, h, bytes, hipMemcpyHostToDevice);
    printf("micro_feed: one 8-wave block per CU, 32 MFMAs (16x16x32 fp16) per wave and step; D = 1 KB LDS-DMA pieces, R = ds_read_b128 per wave and step\n");
    run<0, 0, 0>("MFMA only (ceiling)", src, bytes, sink);
    run<0, 16, 0>("+ 16 fragment reads of a resident tile", src, bytes, sink);
    run<3, 16, 0>("direct 3x3 conv today (3 pieces per tap)", src, bytes, sink);
    run<6, 16, 0>("2x the operand stream", src, bytes, sink);
    run<9, 24, 0>("3x", src, bytes, sink);
    run<12, 32, 0>("4x", src, bytes, sink);
    run<17, 32, 0>("F(2x2,3x3) at its accumulator-limited tile, no transform", src, bytes, sink);
    run<17, 32, 1>("F(2x2,3x3) ... + 48 packed adds (B^T d B)", src, bytes, sink);
    run<6, 24, 1>("F(2,3) 1-D at a 16x16-pixel x 64-channel tile + transform", src, bytes, sink);
    return 0;
}
