// Clock / matrix-rate probe for bench.py (measurement aid, not on the model's path).
//
// The MI355X boxes of one pool hold different clocks under a dense MFMA load (MI355X_MICROARCH.md "DVFS give-back" items 5-6:
// one binary, 12 % apart in wall time across devices), so a bench line is only comparable across boxes together with the clock
// the chip held.  This kernel is the guide's check (6): a register-only fp16 MFMA loop on random operands, stamped once around
// the loop with s_memtime (shader cycles) and s_memrealtime (100 MHz), one 4-wave block per CU:
//     clock [MHz] = d(s_memtime) / d(s_memrealtime) * 100
// The stamps go to a buffer of their own; the accumulators are folded into a second buffer only to keep the loop alive.
#include "common.h"

__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ stamps, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    // operands: a fixed pseudo-random pattern per lane (non-zero, both signs, spread exponents)
    h16x8 a, b;
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s = s * 1664525u + 1013904223u;
        a[e] = (h16)(((int)(s >> 16) & 0xfff) * (1.0f / 2048.0f) - 1.0f);
        s = s * 1664525u + 1013904223u;
        b[e] = (h16)(((int)(s >> 16) & 0xfff) * (1.0f / 2048.0f) - 1.0f);
    }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        // in-place accumulators through inline asm: compiled from the builtin, hipcc rotated the eight accumulators through
        // overlapping register tuples across the unrolled body and every MFMA waited for its neighbour (36 instead of 16 cycles each)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float f = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * 256 + threadIdx.x] = f;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
    (void)lane;
}

// stamps: [nblk][2] uint64 (d s_memtime, d s_memrealtime); sink: nblk*256 floats (scratch).  One launch of `nblk` 4-wave blocks,
// each wave issuing iters * 16 v_mfma_f32_16x16x32_f16 (16 384 FLOP each).
extern "C" int mu_clock_probe(void* stamps, void* sink, int nblk, int iters, void* stream) {
    if (!stamps || !sink || nblk <= 0 || iters <= 0) return MU_ERR_ARG;
    clock_probe_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>((unsigned long long*)stamps, (float*)sink, iters);
    MU_CHECK_LAUNCH();
    return MU_OK;
}
