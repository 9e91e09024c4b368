// Micro-benchmark (debug aid, not part of the library): how well do MFMA blocks and VALU blocks of co-resident waves overlap on
// one SIMD?  Each wave alternates a block of NM fp16 16x16x32 MFMAs (register operands, independent accumulators) with a block of
// VALU work shaped like the attention softmax (NE v_exp_f32 + NF v_fma_f32 + NE/2 cvt_pk).
//   mode 0: no synchronisation, W waves per SIMD (blocks of 256 threads, W blocks per CU)
//   mode 1: ping-pong: 512-thread blocks (2 waves per SIMD), group B one phase behind group A, one s_barrier per phase
//   mode 2: mode 0 with 512-thread blocks and a barrier per iteration, no stagger (lockstep partners)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro_pingpong.hip -o /tmp/micro_pingpong ; run: /tmp/micro_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// role-split test with 32x32x16 MFMAs: NM32 MFMAs of 32 cycles each per iteration
template <int NM32, int NE, int NF, int MODE>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
    constexpr int NA = 4;
    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    h16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (h16)(0.01f * (threadIdx.x % 7 + i)); b[i] = (h16)(0.02f * (threadIdx.x % 5 + i)); }
    float e[NE], f = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) e[i] = 0.001f * (threadIdx.x + i);
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    if (MODE == 6) { if (grp == 1) __builtin_amdgcn_s_setprio(3); }
    if (MODE == 7) { if (grp == 0) __builtin_amdgcn_s_setprio(3); }
    if (grp == 0) {
        if (MODE != 5)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int i = 0; i < NM32; ++i) acc[i % NA] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i % NA], 0, 0, 0);
    } else {
        if (MODE != 4)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < NE; ++i) e[i] = __builtin_amdgcn_exp2f(e[i] * 0.5f - 1.0f);
#pragma unroll
                for (int i = 0; i < NF; ++i) e[i % NE] = fmaf(e[i % NE], 0.999f, 0.001f);
            }
    }
    float s = f;
#pragma unroll
    for (int i = 0; i < NE; ++i) s += e[i];
#pragma unroll
    for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM32, int NE, int NF, int MODE>
static void run32(const char* name, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k32<NM32, NE, NF, MODE><<<256, 512>>>(out, 1000);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        k32<NM32, NE, NF, MODE><<<256, 512>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-44s NM32=%d NE=%d NF=%d: %.3f ms (MFMA cycles/SIMD %.2fM, VALU port cycles %.2fM)\n", name, NM32, NE, NF, best,
           iters * NM32 * 32.0 / 1e6, iters * (NE * 8.0 + NF * 4.0) / 1e6);
}

template <int NM, int NE, int NF>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[NM], const h16x8& a, const h16x8& b) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
}
template <int NE, int NF>
__device__ __forceinline__ void valu_block(float (&e)[NE], float& f) {
#pragma unroll
    for (int i = 0; i < NE; ++i) e[i] = __builtin_amdgcn_exp2f(e[i] * 0.5f - 1.0f);
#pragma unroll
    for (int i = 0; i < NF; ++i) e[i % NE] = fmaf(e[i % NE], 0.999f, 0.001f);      // independent across i (NE chains)
    unsigned int pk = 0;
#pragma unroll
    for (int i = 0; i + 1 < NE; i += 2) {
        auto p = __builtin_amdgcn_cvt_pkrtz(e[i], e[i + 1]);
        pk ^= *reinterpret_cast<unsigned int*>(&p);
    }
    f += __uint_as_float(pk & 0x3f800000u);
}

template <int NM, int NE, int NF, int MODE>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512) void k(float* out, int iters) {
    f32x4 acc[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    h16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (h16)(0.01f * (threadIdx.x % 7 + i)); b[i] = (h16)(0.02f * (threadIdx.x % 5 + i)); }
    float e[NE], f = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) e[i] = 0.001f * (threadIdx.x + i);
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    if (MODE == 6) { if (grp == 1) __builtin_amdgcn_s_setprio(3); }
    if (MODE == 7) { if (grp == 0) __builtin_amdgcn_s_setprio(3); }
    if (MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7) {
        // role split: waves 0-3 (one per SIMD) only MFMAs, waves 4-7 only VALU blocks; MODE 4: MFMA waves alone; MODE 5: VALU waves alone
        if (grp == 0) {
            if (MODE != 5)
                for (int it = 0; it < iters; ++it) mfma_block<NM, NE, NF>(acc, a, b);
        } else {
            if (MODE != 4)
                for (int it = 0; it < iters; ++it) valu_block<NE, NF>(e, f);
        }
        float s = f;
#pragma unroll
        for (int i = 0; i < NE; ++i) s += e[i];
#pragma unroll
        for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        return;
    }
    if (MODE == 1 && grp == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        mfma_block<NM, NE, NF>(acc, a, b);
        if (MODE == 1) __builtin_amdgcn_s_barrier();
        // the VALU block consumes the previous MFMA results (as softmax consumes S) -- a real dependency
#pragma unroll
        for (int i = 0; i < NE; ++i) e[i] += acc[i % NM][i & 3] * 1e-6f;
        valu_block<NE, NF>(e, f);
        if (MODE == 1 || MODE == 2) __builtin_amdgcn_s_barrier();
        a[0] = (h16)(f * 1e-9f);        // and the next MFMA block consumes the VALU result (as PV consumes P)
    }
    if (MODE == 1 && grp == 0) __builtin_amdgcn_s_barrier();
    float s = f;
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// intra-wave interleave: every wave issues MFMA, then a few independent VALU ops, MFMA, ... (one wave per SIMD or two)
template <int NM, int EPM, int FPM, int SHAPE>
__global__ __launch_bounds__(256) void kint(float* out, int iters) {
    constexpr int NA = SHAPE == 32 ? 4 : 8;
    f32x16 acc32[SHAPE == 32 ? NA : 1];
    f32x4 acc16[SHAPE == 32 ? 1 : NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        if (SHAPE == 32) { for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f; }
        else acc16[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    h16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (h16)(0.01f * (threadIdx.x % 7 + i)); b[i] = (h16)(0.02f * (threadIdx.x % 5 + i)); }
    constexpr int NE = 16;
    float e[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) e[i] = 0.001f * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (SHAPE == 32) acc32[i % NA] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[i % NA], 0, 0, 0);
            else acc16[i % NA] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc16[i % NA], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
#pragma unroll
            for (int k = 0; k < EPM; ++k) e[(i * EPM + k) % NE] = __builtin_amdgcn_exp2f(e[(i * EPM + k) % NE]);
#pragma unroll
            for (int k = 0; k < FPM; ++k) e[(i * FPM + k + 5) % NE] = fmaf(e[(i * FPM + k + 5) % NE], 0.999f, 0.001f);
            __builtin_amdgcn_sched_group_barrier(0x2, EPM + FPM, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) s += e[i];
#pragma unroll
    for (int i = 0; i < NA; ++i) s += SHAPE == 32 ? acc32[i][0] : acc16[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 7 && threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(out + 256 * 4 * 512 - 2) = t1 - t0;
}
template <int NM, int EPM, int FPM, int SHAPE>
static void runint(const char* name, int waves_per_simd, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kint<NM, EPM, FPM, SHAPE><<<256 * waves_per_simd, 256>>>(out, 1000);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        kint<NM, EPM, FPM, SHAPE><<<256 * waves_per_simd, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, out + 256 * 4 * 512 - 2, 8, hipMemcpyDeviceToHost);
    printf("   [wave cycles per MFMA slot: %.2f, clock %.2f GHz] ", (double)cyc / ((double)iters * NM), (double)cyc / best / 1e6);
    const double mc = (double)waves_per_simd * iters * NM * (SHAPE == 32 ? 32.0 : 16.0), vc = (double)waves_per_simd * iters * NM * (EPM * 8.0 + FPM * 4.0);
    printf("%-40s shape %d NM=%d exp/MFMA=%d fma/MFMA=%d waves/SIMD=%d: %.3f ms  (MFMA cycles %.2fM, VALU port cycles %.2fM; at 2 GHz max %.2f ms, sum %.2f ms)\n",
           name, SHAPE, NM, EPM, FPM, waves_per_simd, best, mc / 1e6, vc / 1e6, (mc > vc ? mc : vc) / 2e6, (mc + vc) / 2e6);
}

template <int NM, int NE, int NF, int MODE>
static void run(const char* name, int waves_per_simd, float* out) {
    const int threads = MODE == 0 ? 256 : 512;
    const int blocks_per_cu = MODE == 0 ? waves_per_simd : waves_per_simd / 2;
    const int grid = 256 * blocks_per_cu;           // exactly one resident round
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NM, NE, NF, MODE><<<grid, threads>>>(out, 1000);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        k<NM, NE, NF, MODE><<<grid, threads>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // per SIMD: waves_per_simd * iters * NM MFMAs of 16 cycles
    const double mfma_cycles = (double)waves_per_simd * iters * NM * 16.0;
    const double flops = (double)grid * (threads / 64) * iters * NM * 16.0 * 16 * 32 * 2;
    printf("%-34s NM=%d NE=%d NF=%d waves/SIMD=%d: %.3f ms  %.0f TF/s  (MFMA-bound time at 2.0 GHz %.3f ms -> util %.0f%%)\n", name, NM, NE, NF,
           waves_per_simd, best, flops / best / 1e9, mfma_cycles / 2.0e6, 100.0 * mfma_cycles / 2.0e6 / best);
}

int main() {
    float* out;
    hipMalloc(&out, (256 * 4 * 512 + 16) * sizeof(float));
    // forward-attention-like: 36 MFMAs vs 32 exp + 48 other VALU per tile;  dK/dV-like: 32 MFMAs vs 16 exp + 48 other
    run<36, 32, 48, 0>("fwd-like unsync", 3, out);
    run<36, 32, 48, 0>("fwd-like unsync", 2, out);
    run<36, 32, 48, 0>("fwd-like unsync", 1, out);
    run<36, 32, 48, 1>("fwd-like ping-pong", 2, out);
    run<36, 32, 48, 2>("fwd-like lockstep barrier", 2, out);
    run<32, 16, 48, 0>("dkv-like unsync", 3, out);
    run<32, 16, 48, 0>("dkv-like unsync", 2, out);
    run<32, 16, 48, 1>("dkv-like ping-pong", 2, out);
    run<32, 16, 48, 2>("dkv-like lockstep barrier", 2, out);
    run<36, 32, 48, 3>("role split: MFMA wave + VALU wave", 2, out);
    run<36, 32, 48, 4>("role split: MFMA waves alone", 2, out);
    run<36, 32, 48, 5>("role split: VALU waves alone", 2, out);
    run<36, 32, 48, 6>("role split both, VALU wave prio 3", 2, out);
    run<36, 32, 48, 7>("role split both, MFMA wave prio 3", 2, out);
    run32<18, 32, 48, 6>("32x32x16 role split both, VALU wave prio 3", out);
    run32<18, 32, 48, 7>("32x32x16 role split both, MFMA wave prio 3", out);
    run32<18, 32, 48, 3>("32x32x16 role split: MFMA wave + VALU wave", out);
    run32<18, 32, 48, 4>("32x32x16 role split: MFMA waves alone", out);
    run32<18, 32, 48, 5>("32x32x16 role split: VALU waves alone", out);
    run32<18, 32, 112, 3>("32x32x16 role split, plain-VALU heavy: both", out);
    run32<18, 32, 112, 5>("32x32x16 role split, plain-VALU heavy: VALU alone", out);
    runint<32, 0, 0, 16>("interleaved in one stream", 1, out);
    runint<16, 0, 0, 32>("interleaved in one stream", 1, out);
    runint<32, 1, 1, 16>("interleaved in one stream", 1, out);
    runint<32, 1, 1, 16>("interleaved in one stream", 2, out);
    runint<32, 1, 0, 16>("interleaved in one stream", 1, out);
    runint<32, 0, 2, 16>("interleaved in one stream", 1, out);
    runint<32, 0, 1, 16>("interleaved in one stream", 1, out);
    runint<16, 2, 2, 32>("interleaved in one stream", 1, out);
    runint<16, 2, 2, 32>("interleaved in one stream", 2, out);
    runint<16, 1, 4, 32>("interleaved in one stream", 1, out);
    runint<16, 0, 5, 32>("interleaved in one stream", 1, out);
    run<32, 0 + 2, 2, 0>("MFMA only", 1, out);
    run<32, 2, 2, 0>("MFMA only", 2, out);
    return 0;
}
