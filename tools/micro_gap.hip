// Micro-benchmark (debug aid): cycles per v_mfma_f32_32x32x16_f16 / 16x16x32 with VALU fillers hand-placed in the gap
// (one asm block per MFMA gap, exact instruction order), one wave per SIMD.  Prints wave cycles per MFMA from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GAP32(FILL) asm volatile("v_mfma_f32_32x32x16_f16 %0, %8, %9, %0\n\t" FILL \
    : "+v"(acc[i & 3]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]) : "v"(a), "v"(b), "v"(c0))
#define GAP16(FILL) asm volatile("v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n\t" FILL \
    : "+v"(acc4[i & 7]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]) : "v"(a), "v"(b), "v"(c0))

template <int V>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[4];
    f32x4 acc4[8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int i = 0; i < 8; ++i) acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (h16)(0.01f * (threadIdx.x % 7 + i)); b[i] = (h16)(0.02f * (threadIdx.x % 5 + i)); }
    float e[7], c0 = 0.999f;
    for (int i = 0; i < 7; ++i) e[i] = 0.001f * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (V == 0) GAP32("");
            if (V == 1) GAP32("v_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_fma_f32 %3, %3, %10, %10\n\tv_fma_f32 %4, %4, %10, %10");
            if (V == 2) GAP32("v_exp_f32 %1, %1\n\tv_fma_f32 %3, %3, %10, %10\n\tv_exp_f32 %2, %2\n\tv_fma_f32 %4, %4, %10, %10");
            if (V == 3) GAP32("v_fma_f32 %3, %3, %10, %10\n\tv_fma_f32 %4, %4, %10, %10\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2");
            if (V == 4) GAP32("s_nop 7\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_fma_f32 %3, %3, %10, %10\n\tv_fma_f32 %4, %4, %10, %10");
            if (V == 5) GAP32("v_exp_f32 %1, %1\n\tv_exp_f32 %2, %2");
            if (V == 6) GAP32("v_fma_f32 %3, %3, %10, %10\n\tv_fma_f32 %4, %4, %10, %10\n\tv_fma_f32 %5, %5, %10, %10\n\tv_fma_f32 %6, %6, %10, %10\n\tv_fma_f32 %7, %7, %10, %10");
            if (V == 7) GAP32("v_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_cvt_pk_f16_f32 %3, %4, %5\n\tv_dot2_f32_f16 %6, %3, %3, %6");
            if (V == 8) GAP32("v_exp_f32 %1, %1\n\tv_fma_f32 %3, %3, %10, %10\n\tv_fma_f32 %4, %4, %10, %10");
            if (V == 10) GAP16("");
            if (V == 11) GAP16("v_exp_f32 %1, %1");
            if (V == 12) GAP16("v_fma_f32 %3, %3, %10, %10");
            if (V == 13) GAP16("v_fma_f32 %3, %3, %10, %10\n\tv_fma_f32 %4, %4, %10, %10");
            if (V == 14) GAP16("v_exp_f32 %1, %1\n\tv_fma_f32 %3, %3, %10, %10");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 7; ++i) s += e[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += acc4[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 3 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int V> void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 20000;
    k<V><<<256, 256>>>(out, cyc, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<V><<<256, 256>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %.2f cycles per MFMA   (%.3f ms, clock %.2f GHz)\n", name, (double)c / (16.0 * iters), ms, (double)c / ms / 1e6);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<0>("32x32x16 bare", out, cyc);
    run<5>("32x32x16 + exp exp", out, cyc);
    run<8>("32x32x16 + exp fma fma", out, cyc);
    run<1>("32x32x16 + exp exp fma fma", out, cyc);
    run<2>("32x32x16 + exp fma exp fma", out, cyc);
    run<3>("32x32x16 + fma fma exp exp", out, cyc);
    run<4>("32x32x16 + s_nop 7, exp exp fma fma", out, cyc);
    run<6>("32x32x16 + 5 fma", out, cyc);
    run<7>("32x32x16 + exp exp cvt_pk dot2", out, cyc);
    run<10>("16x16x32 bare", out, cyc);
    run<11>("16x16x32 + exp", out, cyc);
    run<12>("16x16x32 + fma", out, cyc);
    run<13>("16x16x32 + fma fma", out, cyc);
    run<14>("16x16x32 + exp fma", out, cyc);
    return 0;
}
