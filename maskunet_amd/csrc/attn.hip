// Masked single-head attention, flash-style (no N x N tensor), forward and backward.
// Reference op replaced: Mask2FormerAttention.forward (ade_semantic.py:163-190):
//   scores = QK^T/sqrt(C) + {0,-inf} key mask; softmax over keys; PV + x; LayerNorm([C]).
// The additive key mask drops whole keys for every query, so the kernels iterate only over
// the KEPT keys (compacted index list kidx[b][0..kcnt[b])): exp(-inf) = 0 exactly, hence
// softmax over the kept keys is the reference's softmax; ~half the FLOPs disappear.
//
// Layouts: qkv [B,N,3C] (q | k | v per token), x/out/oattn/dY [B,N,C], all T (fp16 or fp32);
// lse2/delta/ln_mean/ln_rstd fp32 [B,N].  lse2 is in log2 units of the scaled scores.
//
// MFMA: 16x16 tiles.  "row products" contract over C with both operands read as 16-byte row
// pieces; "accumulator-operand products" feed a 16x16 accumulator tile (rows on registers,
// columns on lanes) straight back as the B operand of the next MFMA, the other operand coming
// from a transposing LDS read (fp16: ds_read_b64_tr_b16) or plain column reads (fp32).
#include "common.h"
#include "../../include/maskunet_hip.h"
#include <stdlib.h>
#include <type_traits>
// Tuning knobs (overridable with -D for tools/ab_bench.py A/B runs).  Measured in-process, N=16384 C=64 B=64 fp16:
//   dK/dV  : 3 waves/SIMD (168 VGPRs, 4 spilled) 4.70 ms vs 2 waves/SIMD 5.24 ms; 1 wave/SIMD with 64 keys/wave 9 ms
//   dQ     : 32-key tiles (152 VGPRs, 3 waves/SIMD) 3.07 ms vs 64-key tiles (196 VGPRs) 3.22 ms; forcing 3 waves/SIMD on the
//            64-key version spills into the loop: 10.97 ms
//   forward: already 3 waves/SIMD at 158 VGPRs
#ifndef MU_DKV_OCC
#define MU_DKV_OCC 3
#endif
#ifndef MU_DQ_OCC
#define MU_DQ_OCC 3
#endif
#ifndef MU_FWD_OCC
#define MU_FWD_OCC 2
#endif
// C = 128 blocks (N = 4096): without a bound the compiler takes ~320 registers (VGPR + AGPR) and runs ONE wave per SIMD.
// In-process A/B (B=64, N=4096, C=128): forward 0.624 -> 0.341 ms with a 2-waves/SIMD bound and 32-key tiles (168 VGPRs),
// dQ 0.520 -> 0.374 ms with the bound alone (226 VGPRs, no spills), dK/dV 1.174 -> 0.959 ms with 16 keys per wave (NKT = 1,
// 150 VGPRs; the 32-key version spills under the bound).  3 waves/SIMD: no further gain.
#ifndef MU_FWD_OCC128
#define MU_FWD_OCC128 3       // (round 2, with the conflict-free swizzle of the 256-byte rows: 0.346 -> 0.301 ms; dQ spills under the same bound)
#endif
#ifndef MU_DQ_OCC128
#define MU_DQ_OCC128 2
#endif
#ifndef MU_DKV_OCC128
#define MU_DKV_OCC128 2
#endif
// C = 256 (N = 1024): the forward gains from the 2-waves/SIMD bound even with 26 spilled registers (0.134 -> 0.102 ms), dQ loses
// (115 spills: 0.092 -> 0.197 ms), dK/dV does not fit at all
#ifndef MU_FWD_OCC256
#define MU_FWD_OCC256 2
#endif
#ifndef MU_DQ_NQT128
#define MU_DQ_NQT128 2          // 1 = one 16-query tile per wave in the C = 128 dQ sweep (A/B below)
#endif
#ifndef MU_DQ_OCC128_Q1
#define MU_DQ_OCC128_Q1 3
#endif
#ifndef MU_DQ_OCC256
#define MU_DQ_OCC256 1
#endif
#ifndef MU_DKV_OCC256
#define MU_DKV_OCC256 1
#endif
#ifndef MU_FWD_KT128
#define MU_FWD_KT128 32
#endif
#ifndef MU_FWD_NW128
#define MU_FWD_NW128 4
#endif
#ifndef MU_DKV_NKT128
#define MU_DKV_NKT128 1
#endif
#ifndef MU_DKV_NW128
#define MU_DKV_NW128 12      // 12-wave blocks = 3 waves/SIMD at 168 VGPRs (in-process, N = 4096: 4 waves x 2 blocks 0.68, 8 waves 0.67, 12 waves 0.58 ms)
#endif
#ifndef MU_DKV_NW64
#define MU_DKV_NW64 4
#endif
#ifndef MU_DKV_NW256
#define MU_DKV_NW256 4
#endif
#ifndef MU_DQ_KT
#define MU_DQ_KT 32
#endif
// forward: optimistic sweep without per-tile running-max tracking, verified afterwards (see attn_fwd2_kernel); 0 = always exact
#ifndef MU_FWD_OPTIMISTIC
#define MU_FWD_OPTIMISTIC 1
#endif
// s_setprio(1) around the MFMA clusters.  In-process A/B (B=64, N=16384, C=64, fp16): dQ 2.94 -> 2.87 ms, dK/dV 3.98 -> 3.90 ms,
// forward 2.03 -> 2.02 ms (noise); at C=128 the dK/dV sweep LOSES 1.5 %.  1 = the C <= 64 backward sweeps only, 2 = everywhere, 0 = off
#ifndef MU_ATTN_SETPRIO
#define MU_ATTN_SETPRIO 1
#endif
// LDS operand prefetch ahead of the VALU phase (number of 16-column blocks; 0 = off)
#ifndef MU_DKV_PREFETCH128
#define MU_DKV_PREFETCH128 0   // (8 = all blocks prefetched: -5 % with 8-wave blocks, spills under the 3-waves/SIMD bound of the 12-wave blocks)
#endif
#ifndef MU_DKV_PREFETCH
#define MU_DKV_PREFETCH 0
#endif
#ifndef MU_DQ_PREFETCH
#define MU_DQ_PREFETCH 0
#endif
#ifndef MU_DKV_PKMUL
#define MU_DKV_PKMUL 1
#endif
#ifndef MU_DKV_ROWC_ONE
#define MU_DKV_ROWC_ONE 0
#endif
// fp32x (chunk-encoded fp32 tiles): the 256-/512-byte-row swizzle of the fp16 tiles for the row fragment + transposed reads, and
// the occupancy bound of the C <= 64 sweeps (the fp32 instantiations run one wave per SIMD)
#ifndef MU_XF_SWZ256
#define MU_XF_SWZ256 1
#endif
// fp32x, C = 128 dK/dV: the 4-deep Q / dO ring takes 128 KB, so a 4-wave block is alone on its CU (one wave per SIMD: matrix pipe busy 25 %);
// 8 waves share the ring (two per SIMD, 16 keys each)
#ifndef MU_XF_DKV_NW128
#define MU_XF_DKV_NW128 8
#endif
#ifndef MU_XF_FWD_KT32
#define MU_XF_FWD_KT32 1
#endif
#ifndef MU_XF_OCC
#define MU_XF_OCC 2
#endif
#ifndef MU_XF_DQ_OCC
#define MU_XF_DQ_OCC 2
#endif
#ifndef MU_XF_DKV_SCHED
#define MU_XF_DKV_SCHED 0
#endif
#ifndef MU_XF_DKV_SCHED2
#define MU_XF_DKV_SCHED2 0
#endif
#ifndef MU_XF_OPAQUE
#define MU_XF_OPAQUE 1
#endif
#ifndef MU_H16_OPAQUE
#define MU_H16_OPAQUE 0
#endif
#ifndef MU_XF_PK_MAXD
#define MU_XF_PK_MAXD 512
#endif
#ifndef MU_FWD_PREFETCH
#define MU_FWD_PREFETCH 0
#endif
#define MU_PRIO_ON(bwd) (MU_ATTN_SETPRIO == 2 || (MU_ATTN_SETPRIO == 1 && (bwd) && D <= 64 && sizeof(T) == 2))
#define MU_PRIO(x) do { if (MU_PRIO_ON(MU_PRIO_BWD)) __builtin_amdgcn_s_setprio(x); } while (0)

typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
// LDS-DMA through inline asm (see conv.hip glds16a): hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of every
// ds_read_b64_tr_b16 (and of reads of a second __shared__ object) while an LDS-DMA it knows of is pending, which turned
// every "tile j+1 in flight while tile j is processed" scheme here into synchronous staging.  Hidden DMAs are ordered by
// hand: MU_SYNC_DMA() (vmcnt(0) + workgroup barrier) where the code used to rely on __syncthreads(), counted waits in
// the ring kernels.  M0 carries the wave-uniform LDS destination; nothing else in these kernels uses M0.
__device__ __forceinline__ void glds16a(const void* gsrc, void* lds_wave_base) {
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(gsrc) : "memory");
}
// scalar-base form: source = sbase (wave-uniform, SGPR pair) + voff (per-lane unsigned byte offset): no 64-bit per-lane
// pointer arithmetic and no pointer pairs held in VGPRs across the loop
__device__ __forceinline__ void glds16s(const void* sbase, uint32_t voff, void* lds_wave_base) {
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    const uint64_t b = (uint64_t)(uintptr_t)sbase;
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);      // (the builtin returns a SIGNED int)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(voff), "s"(sb) : "memory");
}
#define MU_SYNC_DMA()                                        \
    do {                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
        __syncthreads();                                     \
    } while (0)

#define LDS_TR16(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(ptr))

// A tile base the compiler cannot fold into its address arithmetic (fp32x tiles: linear images, every read = per-lane offset + constant).
// DS instructions take a VGPR address + a 16-bit immediate; with the base a compile-time constant (ring slot / buffer index) hipcc
// re-associates base + lane offset + constant into one loop-invariant address REGISTER per (slot, column block, ...) once the images
// reach past 64 KB -- ~90 registers in the C = 128 dK/dV sweep, which then spilled its resident operands and ran every LDS read
// synchronously.  An opaque scalar base costs one v_add per tile and kind of read; everything behind it is an immediate.
template <typename T> __device__ __forceinline__ const T* lds_opaque(const T* p) {
    uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const T*)p;
    asm volatile("" : "+s"(a));
    return (const T*)(__attribute__((address_space(3))) const T*)(uintptr_t)a;
}

// Block -> (image, tile) through the XCD-aware remap: the tiles of one image run on ONE XCD, so the K/V (forward, dQ) or Q/dO (dK/dV)
// rows that every block of the image streams are fetched into that XCD's L2 once instead of once per XCD.
#ifndef MU_ATTN_XCD
#define MU_ATTN_XCD 1
#endif
__device__ __forceinline__ void attn_block(int& bx, int& b) {
    if (MU_ATTN_XCD) {
        const int L = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
        b = L / gridDim.x;
        bx = L - b * gridDim.x;
    } else {
        b = blockIdx.y;
        bx = blockIdx.x;
    }
}

template <typename T> struct AT;
template <> struct AT<h16> {
    // VN: elements per 16-byte chunk; KR: contraction width of one row-product fragment; GS: column stride between the fragments of
    // the four lane groups g; PK: P / dS enter the matrix core as one packed fp16 operand
    static constexpr int VN = 8, KR = 32, GS = 8;
    static constexpr bool PK = true;
    using Frag = h16x8;
    struct AccA { h16x8 v; };
    static __device__ __forceinline__ Frag ld(const h16* p) { return *reinterpret_cast<const Frag*>(p); }
    template <typename Z> static __device__ __forceinline__ Frag ldt(const h16* tile, int row, int col) { return ld(tile + Z::off(row, col)); }
    static __device__ __forceinline__ Frag ld_scaled(const h16* p, float sc) {
        Frag f = ld(p);
#pragma unroll
        for (int e = 0; e < VN; ++e) f[e] = (h16)((float)f[e] * sc);
        return f;
    }
    static __device__ __forceinline__ Frag zero() { return (h16x8)(h16)0; }
    static __device__ __forceinline__ void mma_row(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    // first k-step of a product whose accumulator starts from a row constant: C operand = the constant, D = the accumulator
    // (written as "acc = c0; mma_row(a, b, acc)" the compiler materialises one 4-register copy per accumulator)
    static __device__ __forceinline__ f32x4 mma_row_from(const Frag& a, const Frag& b, const f32x4& c0) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    }
    // (fp32x only: the row products with the key / value operand as ONE term -- see AT<xf32>; here they are the plain products)
    static __device__ __forceinline__ void mma_row_sa(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ void mma_row_sb(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ f32x4 mma_row_from_sa(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    static __device__ __forceinline__ f32x4 mma_row_from_sb(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    // backward-only names (fp32x trades precision differently in the backward sweeps -- see AT<xf32>; here they are the plain products)
    static __device__ __forceinline__ void mma_row_bs_sa(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ void mma_row_bs_sb(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ f32x4 mma_row_from_bs_sa(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    static __device__ __forceinline__ f32x4 mma_row_from_bs_sb(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    static __device__ __forceinline__ void mma_row_bp_sa(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ void mma_row_bp_sb(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ f32x4 mma_row_from_bp_sa(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    static __device__ __forceinline__ f32x4 mma_row_from_bp_sb(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    // A operand [m = tile column col0+r16][k-slot j <-> tile row 4g+j, 16+4g+j]
    static __device__ __forceinline__ AccA ld_acc_a(const h16* tile, int stride, int col0, int g, int r16) {
        const int q = r16 >> 2, pc = r16 & 3;
        auto lo = LDS_TR16(tile + (4 * g + q) * stride + col0 + 4 * pc);
        auto hi = LDS_TR16(tile + (16 + 4 * g + q) * stride + col0 + 4 * pc);
        AccA a;
        a.v = (h16x8){(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
        return a;
    }
    static __device__ __forceinline__ void mma_acc(const AccA& a, const f32x4& p0, const f32x4& p1, f32x4& c) {
        h16x8 b = {(h16)p0[0], (h16)p0[1], (h16)p0[2], (h16)p0[3], (h16)p1[0], (h16)p1[1], (h16)p1[2], (h16)p1[3]};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b, c, 0, 0, 0);
    }
    // the row-sum product (A = the constant-ones operand)
    static __device__ __forceinline__ void mma_ones(const AccA& a, const f32x4& p0, const f32x4& p1, f32x4& c) { mma_acc(a, p0, p1, c); }
    // the same with the B operand already packed (MU_DKV_PKMUL: dS = P * dP' as four v_pk_mul_f16 on the packed halves)
    using Packed = h16x8;
    static __device__ __forceinline__ Packed pack(const f32x4& p0, const f32x4& p1) {
        return (h16x8){(h16)p0[0], (h16)p0[1], (h16)p0[2], (h16)p0[3], (h16)p1[0], (h16)p1[1], (h16)p1[2], (h16)p1[3]};
    }
    static __device__ __forceinline__ void mma_acc_pk(const AccA& a, const Packed& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_accb(const AccA& a, const f32x4& p0, const f32x4& p1, f32x4& c) { mma_acc(a, p0, p1, c); }
    static __device__ __forceinline__ void mma_accb_pk(const AccA& a, const Packed& b, f32x4& c) { mma_acc_pk(a, b, c); }
};
template <> struct AT<float> {
    static constexpr int VN = 4, KR = 16, GS = 4;
    static constexpr bool PK = false;
    using Frag = f32x4;
    struct AccA { float v[8]; };
    static __device__ __forceinline__ Frag ld(const float* p) { return *reinterpret_cast<const Frag*>(p); }
    template <typename Z> static __device__ __forceinline__ Frag ldt(const float* tile, int row, int col) { return ld(tile + Z::off(row, col)); }
    static __device__ __forceinline__ Frag ld_scaled(const float* p, float sc) { return ld(p) * sc; }
    static __device__ __forceinline__ Frag zero() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void mma_row(const Frag& a, const Frag& b, f32x4& c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_row_sa(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ void mma_row_sb(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ f32x4 mma_row_from_sa(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    static __device__ __forceinline__ f32x4 mma_row_from_sb(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    // backward-only names (fp32x trades precision differently in the backward sweeps -- see AT<xf32>; here they are the plain products)
    static __device__ __forceinline__ void mma_row_bs_sa(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ void mma_row_bs_sb(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ f32x4 mma_row_from_bs_sa(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    static __device__ __forceinline__ f32x4 mma_row_from_bs_sb(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    static __device__ __forceinline__ void mma_row_bp_sa(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ void mma_row_bp_sb(const Frag& a, const Frag& b, f32x4& c) { mma_row(a, b, c); }
    static __device__ __forceinline__ f32x4 mma_row_from_bp_sa(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    static __device__ __forceinline__ f32x4 mma_row_from_bp_sb(const Frag& a, const Frag& b, const f32x4& c0) { return mma_row_from(a, b, c0); }
    static __device__ __forceinline__ f32x4 mma_row_from(const Frag& a, const Frag& b, const f32x4& c0) {
        f32x4 c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c0, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ AccA ld_acc_a(const float* tile, int stride, int col0, int g, int r16) {
        AccA a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a.v[r] = tile[(4 * g + r) * stride + col0 + r16];
            a.v[4 + r] = tile[(16 + 4 * g + r) * stride + col0 + r16];
        }
        return a;
    }
    static __device__ __forceinline__ void mma_acc(const AccA& a, const f32x4& p0, const f32x4& p1, f32x4& c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[r], p0[r], c, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[4 + r], p1[r], c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_ones(const AccA& a, const f32x4& p0, const f32x4& p1, f32x4& c) { mma_acc(a, p0, p1, c); }
    static __device__ __forceinline__ void mma_accb(const AccA& a, const f32x4& p0, const f32x4& p1, f32x4& c) { mma_acc(a, p0, p1, c); }
};

// fp32x (common.h): fp32 storage and the fp32 kernels' tiling on FP16-PAIR-ENCODED qkv / dY (32 bytes = [8 fp16 hi | 8 fp16 lo] of
// eight fp32 values, two 16-byte chunks): a row fragment is the hi chunk + the lo chunk of one group (lane group g: columns
// 32 ks + 8 g ..), three v_mfma_f32_16x16x32_f16 per row product (lo hi + hi lo + hi hi); the transposed operands of the
// accumulator-operand products come from ds_read_b64_tr_b16 on the hi and on the lo chunks, and P / dS enter as ONE packed fp16
// operand straight from the accumulators (two MFMAs per product, no register split).  The resident, pre-scaled operands are decoded,
// scaled and re-split once outside the sweep.  Range management (the backward's power-of-two scales): see attn_bwd_t.
template <> struct AT<xf32> {
    static constexpr int VN = 4, KR = 32, GS = 8;
    static constexpr bool PK = true;
    using Frag = SplitH8;
    using AccA = SplitH8;
    using Packed = h16x8;
    // col = 32 ks + 8 g: chunk col / 4 holds the group's hi parts, the next one its lo parts
    template <typename Z> static __device__ __forceinline__ Frag ldt(const xf32* tile, int row, int col) {
        Frag r;
        r.hi = *reinterpret_cast<const h16x8*>(tile + Z::off(row, col));
        r.lo = *reinterpret_cast<const h16x8*>(tile + Z::off(row, col + 4));
        return r;
    }
    // resident operands (global rows, contiguous): decode, scale, split again -- once per wave, outside the sweep
    static __device__ __forceinline__ Frag ld_scaled(const xf32* p, float sc) {
        Frag e;
        e.hi = *reinterpret_cast<const h16x8*>(p);
        e.lo = *reinterpret_cast<const h16x8*>(p + 4);
        float v[8];
        mu_hdec8(e, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= sc;
        return mu_hsplit8(v);
    }
    static __device__ __forceinline__ Frag zero() {
        Frag r;
        r.hi = (h16x8)(h16)0;
        r.lo = (h16x8)(h16)0;
        return r;
    }
    static __device__ __forceinline__ void mma_row(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mma_row_from(const Frag& a, const Frag& b, const f32x4& c0) {
        f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, c0, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    // Round 6: the KEY operand of the score product S = Q K^T and the VALUE operand of dP = dO V^T as ONE fp16 term (their hi halves)
    // against the two-term query / dO: two MFMAs instead of three -- in the forward, the dQ sweep (`_sa`: the single operand is A) and the
    // dK/dV sweep (`_sb`: it is B) alike, so every sweep recomputes exactly the S the forward normalised with.  Sized on the CPU oracle
    // first (tests/aids/numerics_attn_single_term.py, KSINGLE / VSINGLE): outputs 2.0e-5 -> 2.3e-5, worst gradient 5.9e-3 -> 8.7e-3
    // (gates 1e-3 / 5e-2), the value operand without any measurable effect.  The lo halves of K / V stay in memory: P V, dS K keep them.
#ifndef MU_XF_KSINGLE
#define MU_XF_KSINGLE 1
#endif
    static __device__ __forceinline__ void mma_row_sa(const Frag& a, const Frag& b, f32x4& c) {
        if (!MU_XF_KSINGLE) { mma_row(a, b, c); return; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_row_sb(const Frag& a, const Frag& b, f32x4& c) {
        if (!MU_XF_KSINGLE) { mma_row(a, b, c); return; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mma_row_from_sa(const Frag& a, const Frag& b, const f32x4& c0) {
        if (!MU_XF_KSINGLE) return mma_row_from(a, b, c0);
        f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, c0, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mma_row_from_sb(const Frag& a, const Frag& b, const f32x4& c0) {
        if (!MU_XF_KSINGLE) return mma_row_from(a, b, c0);
        f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, c0, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static __device__ __forceinline__ Packed pack(const f32x4& p0, const f32x4& p1) {
        return (h16x8){(h16)p0[0], (h16)p0[1], (h16)p0[2], (h16)p0[3], (h16)p1[0], (h16)p1[1], (h16)p1[2], (h16)p1[3]};
    }
    static __device__ __forceinline__ void mma_acc_pk(const AccA& a, const Packed& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_acc(const AccA& a, const f32x4& p0, const f32x4& p1, f32x4& c) { mma_acc_pk(a, pack(p0, p1), c); }
    static __device__ __forceinline__ void mma_ones(const AccA& a, const f32x4& p0, const f32x4& p1, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, pack(p0, p1), c, 0, 0, 0);          // (the ones operand has no lo part)
    }
    // Round 6, BACKWARD sweeps only (the forward keeps the two-term products above: north_star's 1e-3 is a gate on OUTPUTS): dP = dO V^T
    // (MU_XF_BWD_DP1) and the gradient products dV = P^T dO, dK = dS^T Q, dQ = dS K (MU_XF_BWD_G1) as ONE fp16 MFMA on the hi halves, i.e.
    // the fp16 kernels' arithmetic on exactly scaled operands (dY carries its power-of-two scale, P its 2^pshift).  Each of these sums runs
    // over hundreds to thousands of keys / queries, and the 2^-12 operand roundings are random-signed: the sum carries ~2^-12 of the root-sum-square of its terms, below the single-term dS
    // noise these sweeps already have: the CPU sizing
    // (tests/aids/numerics_attn_single_term.py DP1 / DV1 / DK1 / DQ1 on the reference's golden) shows NO measurable change of any gradient
    // metric (worst parameter gradient 8.7e-3 with and without, gate 5e-2), and on the GPU the goldens' worst gradients did not move.
    // The recomputed SCORES keep the forward's two terms (MU_XF_BWD_S1 = 0): with one term the backward's P is no longer the forward's
    // (dQ / dK of the kernel-level check 6.4e-4 -> 1.1e-3 against its 1e-3 gate) for another 4.5 % of the step -- measured, not adopted.
#ifndef MU_XF_BWD_S1
#define MU_XF_BWD_S1 0
#endif
#ifndef MU_XF_BWD_DP1
#define MU_XF_BWD_DP1 1
#endif
#ifndef MU_XF_BWD_G1
#define MU_XF_BWD_G1 1
#endif
    static __device__ __forceinline__ void mma_row_bs_sa(const Frag& a, const Frag& b, f32x4& c) {
        if (!MU_XF_BWD_S1) { mma_row_sa(a, b, c); return; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_row_bs_sb(const Frag& a, const Frag& b, f32x4& c) {
        if (!MU_XF_BWD_S1) { mma_row_sb(a, b, c); return; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mma_row_from_bs_sa(const Frag& a, const Frag& b, const f32x4& c0) {
        if (!MU_XF_BWD_S1) return mma_row_from_sa(a, b, c0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c0, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mma_row_from_bs_sb(const Frag& a, const Frag& b, const f32x4& c0) {
        if (!MU_XF_BWD_S1) return mma_row_from_sb(a, b, c0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c0, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_row_bp_sa(const Frag& a, const Frag& b, f32x4& c) {
        if (!MU_XF_BWD_DP1) { mma_row_sa(a, b, c); return; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_row_bp_sb(const Frag& a, const Frag& b, f32x4& c) {
        if (!MU_XF_BWD_DP1) { mma_row_sb(a, b, c); return; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mma_row_from_bp_sa(const Frag& a, const Frag& b, const f32x4& c0) {
        if (!MU_XF_BWD_DP1) return mma_row_from_sa(a, b, c0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c0, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mma_row_from_bp_sb(const Frag& a, const Frag& b, const f32x4& c0) {
        if (!MU_XF_BWD_DP1) return mma_row_from_sb(a, b, c0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c0, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_accb_pk(const AccA& a, const Packed& b, f32x4& c) {
        if (!MU_XF_BWD_G1) { mma_acc_pk(a, b, c); return; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_accb(const AccA& a, const f32x4& p0, const f32x4& p1, f32x4& c) { mma_accb_pk(a, pack(p0, p1), c); }
};
// reductions over the 4 lanes {r16 + 16g}: v_permlane16_swap / v_permlane32_swap exchange 16-/32-lane rows in
// registers (no LDS round trip, unlike the ds_bpermute behind __shfl_xor).  swap(v, v) returns {own, partner} in
// an order that depends on the row parity, so a symmetric combine needs no select.
__device__ __forceinline__ float grp_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float grp_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <typename T> __device__ __forceinline__ void store4(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4<h16>(h16* p, const float v[4]) {
    h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
    *reinterpret_cast<h16x4*>(p) = o;
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<xf32>(xf32* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
// fp32x: a gradient row piece written as the matrix-operand encoding its consumer reads (common.h: [4 bf16 hi | 4 bf16 lo] per 16 bytes)
template <typename T> __device__ __forceinline__ void store4e(T* p, const float v[4], bool enc) {
    if constexpr (std::is_same<T, xf32>::value) {
        if (enc) { *reinterpret_cast<uint4*>(p) = mu_enc4((f32x4){v[0], v[1], v[2], v[3]}); return; }
    }
    store4<T>(p, v);
}
template <typename T> __device__ __forceinline__ void load4(const T* p, float v[4]);
template <> __device__ __forceinline__ void load4<h16>(const h16* p, float v[4]) {
    h16x4 o = *reinterpret_cast<const h16x4*>(p);
    v[0] = (float)o[0]; v[1] = (float)o[1]; v[2] = (float)o[2]; v[3] = (float)o[3];
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
    float4 o = *reinterpret_cast<const float4*>(p);
    v[0] = o.x; v[1] = o.y; v[2] = o.z; v[3] = o.w;
}
template <> __device__ __forceinline__ void load4<xf32>(const xf32* p, float v[4]) {
    float4 o = *reinterpret_cast<const float4*>(p);
    v[0] = o.x; v[1] = o.y; v[2] = o.z; v[3] = o.w;
}

// ------------------------------------------------------------------------------------------
// forward: flash-style masked attention over the kept keys (the first, register-staged version of this kernel and of the
// backward sweeps was removed once these LDS-DMA versions had replaced it everywhere)
//  * K/V tiles (KT kept keys) arrive by LDS-DMA (global_load_lds_dwordx4 with per-lane gathered
//    source rows; padded rows read a zero page), double-buffered: tile j+1 is in flight while tile j
//    is processed; the XOR chunk swizzle sits on the source chunk and on every read, which makes
//    both the ds_read_b128 row reads of K and the transposed reads of V conflict-free at C=64;
//  * the softmax denominator is accumulated by the matrix core (a constant-ones A operand) instead
//    of VALU adds + a final cross-lane reduce;
//  * the O/l rescale runs only when some row's running max actually moved (wave-uniform branch);
//  * key-range masking only in the final, partial tile.
// ------------------------------------------------------------------------------------------

template <typename T, int D> struct SwzTile {
    static constexpr int VN = AT<T>::VN;
    static constexpr int ROWB = D * (int)sizeof(T);       // bytes per row
    static constexpr int CPR = ROWB / 16;                 // 16-byte chunks per row
    static constexpr int RPW = 1024 / ROWB;               // rows per wave LDS-DMA instruction (one 1 KB block)
    static constexpr int BLK = RPW * D;                   // elements between the LDS destinations of consecutive DMA instructions
    static constexpr int SW = CPR < 8 ? CPR - 1 : 7;
    static __device__ __forceinline__ constexpr int TE(int rows) { return rows * D; }      // elements of a tile of `rows` rows
    // XOR key of a row (applied to the 16-byte chunk index by the DMA source permutation and by every read).
    //  * rows of <= 128 bytes: row & 7 -- both the ds_read_b128 row reads and the transposed ds_read_b64_tr_b16 reads are conflict-free;
    //  * fp16 rows of 256 / 512 bytes (C = 128 / 256): every row starts on bank 0, so the bank slot is the chunk index mod 16.
    //    row & 7 left BOTH read kinds 2-way conflicting (SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE on all three C = 128
    //    kernels): the 16 lanes of a b128 group (rows {0-3, 12-15} at chunk c, rows {4-11} at chunk c+1) fell on 8 slots, and the
    //    8 rows x 2 chunks of a transposed read on 8 slots.  (row & 7) << 1 puts each of 8 consecutive rows on its own even/odd slot
    //    PAIR (transposed reads touch {c, c+1} with c even) and makes row -> slot injective over the 16 rows of a b128 group.
    static __device__ __forceinline__ constexpr int key(int row) {
        return (sizeof(T) == 2 && ROWB >= 256) ? ((row & 7) << 1) : (row & SW);
    }
    // element offset of (row, col) in the swizzled image
    static __device__ __forceinline__ int off(int row, int col) {
        return row * D + ((((col / VN) ^ key(row))) * VN) + (col % VN);
    }
    // LDS-DMA (lane-linear: lane l of instruction i lands at byte i * 1024 + 16 l): the row inside the block, and the source chunk of that row
    static __device__ __forceinline__ int lrow(int lane) { return lane / CPR; }
    static __device__ __forceinline__ int schunk(int row, int lane) { return (lane % CPR) ^ key(row); }
};

// fp32x tiles (fp16-pair encoded rows of 4-byte elements; chunk 2s = the hi parts of column group s, chunk 2s + 1 its lo parts): a LINEAR
// image per 1 KB DMA block instead of an XOR swizzle.  An LDS-DMA instruction lands lane-linear, but which source chunk a lane fetches
// is free, and so is every instruction's LDS base.  A block holds RPW = 1024 / ROWB rows; it is laid out
//     [16-column block dt][hi | lo][row in block][half h of the block's two groups]        (S = 2 RPW chunks per [dt][part])
// and consecutive blocks start 1024 + 16 (S mod 16) bytes apart.  Then
//   * a transposed read (one 16-column block, hi or lo parts, 8 consecutive rows per 32-lane half) takes S consecutive chunks from each
//     of the 8 / RPW blocks involved, which the block stride places on disjoint sixteenths of the bank row: conflict-free;
//   * a ds_read_b128 lane group (8 rows of column group s, 8 rows of group s + 1) lands on the 16 chunks {2 row + h}: conflict-free;
//   * every address is (per-lane constant) + dt * 32 S + part * 16 S bytes: immediates, no XOR arithmetic and no per-block address
//     registers (the XOR image of round 4 -- [4 hi | 4 lo] per chunk, key (row & 7) << 1 -- spent 25-33 % of its LDS cycles in bank
//     conflicts, profiles/r04_fp32x_lds_conflicts.md: a transposed read touched only the hi or only the lo half of every chunk).
template <int D> struct SwzTile<xf32, D> {
    static constexpr int VN = 4;
    static constexpr int ROWB = D * 4, CPR = ROWB / 16, RPW = 1024 / ROWB;
    static constexpr int S = 2 * RPW;
    static constexpr int BLK = (1024 + 16 * (S % 16)) / 4;
    static __device__ __forceinline__ constexpr int TE(int rows) { return (rows / RPW) * BLK; }
    static __device__ __forceinline__ constexpr int key(int) { return 0; }
    static __device__ __forceinline__ int off(int row, int col) {
        const int c = col >> 2, sg = c >> 1;                 // chunk, column group
        return (row / RPW) * BLK + (((sg >> 1) * 2 * S + (c & 1) * S + 2 * (row % RPW) + (sg & 1)) << 2) + (col & 3);
    }
    static __device__ __forceinline__ int lrow(int lane) { return (lane % S) >> 1; }
    static __device__ __forceinline__ int schunk(int, int lane) { return 2 * (2 * (lane / (2 * S)) + (lane & 1)) + ((lane / S) & 1); }
};

// Kept-key row indices of one KT-key tile for this lane's DMA rows (-1 = past the end -> zero page).  Loaded one
// iteration AHEAD of the DMA that consumes them, so the dependent index->address chain never stalls the loop.
template <typename T, int D, int KT, int NW> struct KvStage {
    using Z = SwzTile<T, D>;
    static constexpr int NI = KT / Z::RPW;                 // wave DMA instructions per tile (K and V each)
    static constexpr int NPW = (NI + NW - 1) / NW;         // per wave
    int idx[NPW];

    // Unconditional, clamped loads (rows past the end re-read the last kept key: finite data, masked through the score):
    // a conditional load needs a "-1" default, and overwriting a register with a possibly pending load costs a
    // s_waitcnt vmcnt(0) -- right behind the DMA issue, i.e. it would serialise the staging.
    __device__ __forceinline__ void load_idx(const int* kidx_b, int j0, int Nk, int wave, int lane) {
        const int lrow = Z::lrow(lane);
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int i = wave + NW * n;
            int j = j0 + i * Z::RPW + lrow;
            j = j < Nk ? j : Nk - 1;
#ifdef MU_ATTN_ABL_NOIDX
            idx[n] = j < 0 ? 0 : j;                          // timing-only ablation: contiguous rows instead of the kept-key gather
#else
            idx[n] = kidx_b[j < 0 ? 0 : j];
#endif
        }
    }
    // Rows past the end re-read a valid key row instead of a zero page: every consumer masks those keys through the score
    // (s = -inf -> p = 0 -> ds = 0), so they only need FINITE data, and a common scalar base keeps the DMA in its
    // SGPR-base + 32-bit-offset form.
    template <bool DOK = true, bool DOV = true>
    __device__ __forceinline__ void issue(T* Kt, T* Vt, const T* qkv_b, int wave, int lane) const {
        const int lrow = Z::lrow(lane);
        // all offsets first: each consumes an index load, and the wait for the LAST index load is a vmcnt(0) as far as the
        // compiler can tell -- it must not come after the first (hidden) DMA or that DMA is drained on the spot
        uint32_t off[NPW];
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int row = (wave + NW * n) * Z::RPW + lrow;
            const int sc = Z::schunk(row, lane);
            off[n] = (uint32_t)((idx[n] * 3 * D + sc * Z::VN) * (int)sizeof(T));
            asm volatile("" ::"v"(off[n]));
        }
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int i = wave + NW * n;
            if constexpr (NI % NW != 0) { if (i >= NI) break; }     // wave-uniform; no branch at all when every wave has NPW rows
            if (DOK) glds16s(qkv_b + D, off[n], Kt + i * Z::BLK);
            if (DOV) glds16s(qkv_b + 2 * D, off[n], Vt + i * Z::BLK);
        }
    }
};

// accumulator-operand A fragment from a swizzled tile: rows r0+4g+j (j<4) and r0+16+4g+j, column col0+r16
template <typename T, int D> struct AccLd;
template <int D> struct AccLd<h16, D> {
    static __device__ __forceinline__ AT<h16>::AccA ld(const h16* tile, int r0, int col0, int g, int r16) {
        using Z = SwzTile<h16, D>;
        const int q = r16 >> 2, pc = r16 & 3;
        auto lo = LDS_TR16(tile + Z::off(r0 + 4 * g + q, col0 + 4 * pc));
        auto hi = LDS_TR16(tile + Z::off(r0 + 16 + 4 * g + q, col0 + 4 * pc));
        AT<h16>::AccA a;
        a.v = (h16x8){(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
        return a;
    }
    static __device__ __forceinline__ AT<h16>::AccA ones() {
        AT<h16>::AccA a;
        a.v = (h16x8)(h16)1.0f;
        return a;
    }
};
template <int D> struct AccLd<float, D> {
    static __device__ __forceinline__ AT<float>::AccA ld(const float* tile, int r0, int col0, int g, int r16) {
        using Z = SwzTile<float, D>;
        AT<float>::AccA a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a.v[r] = tile[Z::off(r0 + 4 * g + r, col0 + r16)];
            a.v[4 + r] = tile[Z::off(r0 + 16 + 4 * g + r, col0 + r16)];
        }
        return a;
    }
    static __device__ __forceinline__ AT<float>::AccA ones() {
        AT<float>::AccA a;
#pragma unroll
        for (int r = 0; r < 8; ++r) a.v[r] = 1.0f;
        return a;
    }
};

template <int D> struct AccLd<xf32, D> {
    // fp16-pair tile: lane (g, q = r16 >> 2, pc = r16 & 3) addresses row r0 + 4g + q (and + 16), columns col0 + 4 pc .. + 3: half
    // (pc & 1) of the hi chunk of group (col0 + 4 pc) / 8, and the same half of the lo chunk next to it; the transposing read hands
    // lane r16 column col0 + r16 of rows 4g .. 4g + 3
    static __device__ __forceinline__ SplitH8 ld(const xf32* tile, int r0, int col0, int g, int r16) {
        using Z = SwzTile<xf32, D>;
        const int q = r16 >> 2, pc = r16 & 3;
        const int cg = (col0 + 4 * pc) & ~7, hb = (pc & 1) * 8;
        const int ra = r0 + 4 * g + q, rb = ra + 16;
        const char* a0 = reinterpret_cast<const char*>(tile + Z::off(ra, cg)) + hb;
        const char* a1 = reinterpret_cast<const char*>(tile + Z::off(rb, cg)) + hb;
        const char* l0 = reinterpret_cast<const char*>(tile + Z::off(ra, cg + 4)) + hb;
        const char* l1 = reinterpret_cast<const char*>(tile + Z::off(rb, cg + 4)) + hb;
        const uint2 h0 = __builtin_bit_cast(uint2, LDS_TR16(a0)), h1 = __builtin_bit_cast(uint2, LDS_TR16(a1));
        const uint2 w0 = __builtin_bit_cast(uint2, LDS_TR16(l0)), w1 = __builtin_bit_cast(uint2, LDS_TR16(l1));
        SplitH8 r;
        r.hi = __builtin_bit_cast(h16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
        r.lo = __builtin_bit_cast(h16x8, make_uint4(w0.x, w0.y, w1.x, w1.y));
        return r;
    }
    static __device__ __forceinline__ SplitH8 ones() {
        SplitH8 r;
        r.hi = (h16x8)(h16)1.0f;
        r.lo = (h16x8)(h16)0;
        return r;
    }
};

template <typename T, int D, int KT, int NW, int OCC = 0, int NQ = 2>
__global__ __launch_bounds__(NW * 64, OCC ? OCC : ((NW == 4 && D <= 128 && std::is_same<T, xf32>::value) ? MU_XF_OCC : (NW == 4 && D <= 64 && sizeof(T) == 2) ? MU_FWD_OCC : ((NW == 4 && D == 128 && sizeof(T) == 2) ? MU_FWD_OCC128 : ((NW == 4 && D == 256 && sizeof(T) == 2) ? MU_FWD_OCC256 : 1)))) void attn_fwd2_kernel(const T* __restrict__ qkv, const T* __restrict__ x, const int* __restrict__ kidx,
                                                        const int* __restrict__ kcnt, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, T* __restrict__ oattn,
                                                        float* __restrict__ lse2, float* __restrict__ ln_mean, float* __restrict__ ln_rstd,
                                                        int N, int nkmax, float scale_log2, float eps) {
    using A = AT<T>;
    using Frag = typename A::Frag;
    using Z = SwzTile<T, D>;
    constexpr int VN = A::VN, KR = A::KR, NKS = D / KR, NDT = D / 16, NKT = KT / 16, NH = KT / 32;
    constexpr bool MU_PRIO_BWD = false;
    constexpr int TEK = Z::TE(KT);                                       // elements of one K (or V) tile image
    __shared__ __attribute__((aligned(16))) T lds[2 * 2 * TEK];          // [buf][K|V][tile image]

    int bx_, b;
    attn_block(bx_, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int q0 = bx_ * (NW * NQ * 16) + wave * (NQ * 16);     // NQ 16-query tiles per wave
    const T* qkv_b = qkv + (long)b * N * 3 * D;
    const int Nk = kcnt[b];
    const int* kidx_b = kidx + (long)b * nkmax;

    // LDS: [buf][K|V][KT][D].  The tile loop is unrolled by two with the buffer as a compile-time constant so every
    // LDS address is (loop-invariant per-lane base) + (immediate): no address arithmetic is left in the loop.
    KvStage<T, D, KT, NW> stg;
    stg.load_idx(kidx_b, 0, Nk, wave, lane);
    stg.issue(lds, lds + TEK, qkv_b, wave, lane);
    stg.load_idx(kidx_b, KT, Nk, wave, lane);            // indices of tile 1, consumed inside iteration 0

    // Q fragments pre-multiplied by log2(e)/sqrt(C): the score MFMA then yields exponents directly
    Frag qf[NQ][NKS];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        int qrow = q0 + t * 16 + r16;
        if (qrow > N - 1) qrow = N - 1;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[t][ks] = A::ld_scaled(qkv_b + (long)qrow * 3 * D + ks * KR + g * A::GS, scale_log2);
        }
    }
    f32x4 o[NDT][NQ], lacc[NQ], negm[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        lacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        negm[t] = (f32x4){0.f, 0.f, 0.f, 0.f};           // running max m = 0 until the first tile fixes it
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const typename A::AccA ones = AccLd<T, D>::ones();
    MU_SYNC_DMA();

    // one KT-key tile out of LDS buffer BUF.  S' = K (c Q)^T - m comes straight out of the matrix core (C operand =
    // -m broadcast), so the common case is p = exp2(S') with no per-element subtract; only when some row's maximum
    // moves (or on the very first tile) is the correction path taken.
    auto tile = [&](auto BUFC, auto EXACTC, int j0) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr bool EXACT = decltype(EXACTC)::value;
        const T* Kt = lds + BUF * 2 * TEK;
        if constexpr (std::is_same<T, xf32>::value && MU_XF_OPAQUE) Kt = lds_opaque(Kt);
        const T* Vt = Kt + TEK;
#ifndef MU_FWD_ABL_NODMA
        if (j0 + KT < Nk) {
            T* Kn = lds + (BUF ^ 1) * 2 * TEK;
            stg.issue(Kn, Kn + TEK, qkv_b, wave, lane);
            stg.load_idx(kidx_b, j0 + 2 * KT, Nk, wave, lane);
        }
#endif
        f32x4 s[NKT][NQ];
        MU_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#ifdef MU_FWD_ABL_NOS
                // timing-only ablation: no K reads, no score MFMAs
                if (ks == 0)
                    for (int t = 0; t < NQ; ++t) { s[kt][t] = negm[t]; asm volatile("" : "+v"(s[kt][t])); }
#else
                Frag a = A::template ldt<Z>(Kt, kt * 16 + r16, ks * KR + g * A::GS);
#pragma unroll
                for (int t = 0; t < NQ; ++t) {
                    if (ks == 0) s[kt][t] = A::mma_row_from_sa(a, qf[t][0], negm[t]);     // -m rides in as the C operand
                    else A::mma_row_sa(a, qf[t][ks], s[kt][t]);
                }
#endif
            }
        MU_PRIO(0);
        constexpr int PREV = (MU_FWD_PREFETCH && sizeof(T) == 2) ? (MU_FWD_PREFETCH < NDT ? MU_FWD_PREFETCH : NDT) : 0;
        typename A::AccA vap[PREV ? PREV : 1];
        if constexpr (PREV > 0) {                    // transposed V operands of the first 32 keys: in flight during the softmax VALU
#pragma unroll
            for (int dt = 0; dt < PREV; ++dt) vap[dt] = AccLd<T, D>::ld(Vt, 0, dt * 16, g, r16);
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool partial = j0 + KT > Nk;          // wave-uniform
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            if (partial) {
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (j0 + kt * 16 + 4 * g + r >= Nk) s[kt][t][r] = -INFINITY;
            }
            // max_i32 on the float bit patterns orders all non-negative floats correctly and keeps every negative one
            // below zero -- all the fast path needs ("did any score exceed the running max?"), without the
            // canonicalising v_max hipcc puts in front of every fmaxf of an MFMA result.  The first tile takes the
            // exact float maximum (it may be negative).
            float mx = 0.f;
            if (j0 == 0) {
                mx = fmaxf(fmaxf(s[0][t][0], s[0][t][1]), fmaxf(s[0][t][2], s[0][t][3]));
#pragma unroll
                for (int kt = 1; kt < NKT; ++kt) mx = fmaxf(fmaxf(mx, fmaxf(s[kt][t][0], s[kt][t][1])), fmaxf(s[kt][t][2], s[kt][t][3]));
                mx = grp_max(mx);
            } else if (EXACT) {
                int mi = __float_as_int(s[0][t][0]);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mi = max(mi, __float_as_int(s[kt][t][r]));
                mx = grp_max(__int_as_float(mi < 0 ? (int)0x80000000 : mi));     // -0.0 stands for "nothing above the max"
            }
            // !EXACT (the optimistic sweep): after the first tile the reference maximum m stays where the first tile put it.  The
            // softmax is invariant to m; what m must prevent is overflow of p = 2^(s-m) in the fp16 B operand (s - m > 16), and that
            // shows up as a non-finite row sum, which the caller checks once after the sweep (then redoes it with EXACT tracking).
            if (j0 == 0 || (EXACT && !__all(mx <= 0.f))) {      // first tile, or some row's max moved (rare afterwards)
                const float d = (j0 == 0) ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-d);
                negm[t] -= d;
                lacc[t] *= alpha;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt][t] *= alpha;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) s[kt][t] -= d;
            }
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
#ifdef MU_FWD_ABL_NOEXP
                for (int r = 0; r < 4; ++r) s[kt][t][r] = s[kt][t][r] * 0.001f;          // timing-only ablation: no exponentials
#else
                for (int r = 0; r < 4; ++r) s[kt][t][r] = __builtin_amdgcn_exp2f(s[kt][t][r]);
#endif
        }
#ifdef MU_FWD_ABL_NOPV
        // timing-only ablation: no V reads, no P.V / row-sum MFMAs (P kept live)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int t = 0; t < NQ; ++t) asm volatile("" ::"v"(s[kt][t]));
#else
        MU_PRIO(1);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int t = 0; t < NQ; ++t) A::mma_ones(ones, s[2 * h][t], s[2 * h + 1][t], lacc[t]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                typename A::AccA va;
                if (h == 0 && dt < PREV) va = vap[dt];
                else va = AccLd<T, D>::ld(Vt, 32 * h, dt * 16, g, r16);
#pragma unroll
                for (int t = 0; t < NQ; ++t) A::mma_acc(va, s[2 * h][t], s[2 * h + 1][t], o[dt][t]);
            }
        }
        MU_PRIO(0);
#endif
#ifdef MU_FWD_ABL_NOBAR
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        MU_SYNC_DMA();        // tile j+1 landed (vmcnt(0)) and everyone is done reading tile j
#endif
    };
    // Optimistic sweep first (wherever P is an fp16 operand: fp16 storage and fp32x): no per-tile running-max scan / cross-lane reduce / branch (13 % of the kernel at
    // N = 16384, C = 64: a serial dependent chain between the score MFMAs and the exponentials).  A row whose later scores exceed the
    // first tile's maximum by more than 16 (log2 units) overflows fp16 and leaves an infinite row sum: the whole block then repeats
    // the sweep with exact tracking (wave-uniform decision through the barrier; never taken on the model's data, forced in the tests).
    constexpr bool OPTIMISTIC = MU_FWD_OPTIMISTIC && A::PK;
    bool redo = false;
    if (OPTIMISTIC) {
        for (int j0 = 0; j0 < Nk; j0 += 2 * KT) {
            tile(std::integral_constant<int, 0>{}, std::false_type{}, j0);
            if (j0 + KT < Nk) tile(std::integral_constant<int, 1>{}, std::false_type{}, j0 + KT);
        }
        bool bad = false;
#pragma unroll
        for (int t = 0; t < NQ; ++t) bad = bad || !(lacc[t][0] < 3.0e38f);
        redo = __syncthreads_or(bad);
        if (redo) {
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                lacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                negm[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            stg.load_idx(kidx_b, 0, Nk, wave, lane);
            stg.issue(lds, lds + TEK, qkv_b, wave, lane);
            stg.load_idx(kidx_b, KT, Nk, wave, lane);
            MU_SYNC_DMA();
        }
    }
    if (!OPTIMISTIC || redo) {
        for (int j0 = 0; j0 < Nk; j0 += 2 * KT) {
            tile(std::integral_constant<int, 0>{}, std::true_type{}, j0);
            if (j0 + KT < Nk) tile(std::integral_constant<int, 1>{}, std::true_type{}, j0 + KT);
        }
    }
    float m[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) m[t] = -negm[t][0];

#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const float ltot = lacc[t][0];           // every row of the ones-product holds the same column sum
        const float inv = 1.0f / ltot;
        const int qrow = q0 + t * 16 + r16;
        const bool valid = qrow < N;
        const long tok = (long)b * N + (valid ? qrow : 0);
        float yv[NDT][4];
        float sum = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            float xr[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid) load4<T>(x + tok * D + dt * 16 + 4 * g, xr);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[dt][t][r] *= inv;
                yv[dt][r] = o[dt][t][r] + xr[r];
                sum += yv[dt][r];
            }
        }
        const float mean = grp_sum(sum) * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = yv[dt][r] - mean; sq += d * d; }
        const float rstd = rsqrtf(grp_sum(sq) * (1.0f / D) + eps);
        if (valid) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const int c = dt * 16 + 4 * g;
                float ov[4], av[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ov[r] = (yv[dt][r] - mean) * rstd * gamma[c + r] + beta[c + r];
                    av[r] = o[dt][t][r];
                }
                store4<T>(out + tok * D + c, ov);
                store4<T>(oattn + tok * D + c, av);
            }
            if (g == 0) {
                lse2[tok] = m[t] + log2f(ltot);
                ln_mean[tok] = mean;
                ln_rstd[tok] = rstd;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward prepass: LayerNorm backward per token + delta = rowsum(dY * O) + dgamma/dbeta partials
//   y = O + x; xhat = (y-mean)*rstd; dY = rstd*(g*gamma - mean_c(g*gamma) - xhat*mean_c(g*gamma*xhat))
// one row is handled by C/VN adjacent lanes.
// ------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_ln_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ oattn, const T* __restrict__ x,
                                                          const float* __restrict__ ln_mean, const float* __restrict__ ln_rstd,
                                                          const float* __restrict__ gamma, T* __restrict__ dY, float* __restrict__ delta,
                                                          double* __restrict__ part, long rows, const float* __restrict__ lse2,
                                                          float* __restrict__ rowc, int N, float scale, int cv, float pshift,
                                                          float* __restrict__ amax_part) {
    // pshift / amax_part (fp32x only, 0 / NULL otherwise): the row constant -lse2 is stored as pshift - lse2 (the sweeps then compute
    // 2^pshift P), and every block leaves max|dY| over its rows in amax_part[block] (-> the power-of-two scale of the encoded dY)
    // cv <= D: channels the LayerNorm really spans (channel counts that are not one of the kernels' widths run zero-padded to D)
    constexpr int VN = AT<T>::VN, LPR = D / VN, RPI = 256 / LPR;     // lanes per row, rows per block-iteration
    const int tid = threadIdx.x;
    const int lc = tid % LPR, lr = tid / LPR;
    const int c = lc * VN;
    float ga[VN], dga[VN], dbe[VN], am = 0.f;
#pragma unroll
    for (int i = 0; i < VN; ++i) { ga[i] = gamma[c + i]; dga[i] = 0.f; dbe[i] = 0.f; }
    const long rows_per_blk = (rows + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_blk, r1 = r0 + rows_per_blk < rows ? r0 + rows_per_blk : rows;
    for (long rb = r0; rb < r1; rb += RPI) {
        const long r = rb + lr;
        const bool ok = r < r1;
        const long rr = ok ? r : r0;
        Vec16<T> gv, ov, xv, dv;
        gv.load(gout + rr * D + c); ov.load(oattn + rr * D + c); xv.load(x + rr * D + c);
        const float mu = ln_mean[rr], rs = ln_rstd[rr];
        float xh[VN], gg[VN], a = 0.f, bsum = 0.f;
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            xh[i] = (ov.get(i) + xv.get(i) - mu) * rs;
            gg[i] = gv.get(i) * ga[i];
            a += gg[i];
            bsum += gg[i] * xh[i];
        }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) { a += __shfl_xor(a, o); bsum += __shfl_xor(bsum, o); }
        const float icv = 1.0f / (float)cv;
        a *= icv; bsum *= icv;
        float dl = 0.f;
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const float d = (c + i < cv) ? rs * (gg[i] - a - xh[i] * bsum) : 0.f;
            dv.set(i, d);
            dl += dv.get(i) * ov.get(i);
            if (ok) am = (d == d) ? fmaxf(am, fabsf(d)) : d;           // a NaN sticks (fmaxf would drop it)
            if (ok && c + i < cv) { dga[i] += gv.get(i) * xh[i]; dbe[i] += gv.get(i); }
        }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) dl += __shfl_xor(dl, o);
        if (ok) {
            dv.store(dY + r * D + c);
            if (lc == 0) {
                delta[r] = dl;
                // row constants for the dK/dV sweep, grouped per 32-query tile: [b][tile][{-lse2, -delta/sqrt(C)}][32]
                const long bb = r / N;
                const int qn = (int)(r - bb * N);
                const int ntile = (N + 31) >> 5;
                float* rc = rowc + ((bb * ntile + (qn >> 5)) * 2) * 32 + (qn & 31);
                rc[0] = pshift - lse2[r];
                rc[32] = -dl * scale;
            }
        }
    }
    // column partials: reduce the RPI row-lanes of this block through LDS
    __shared__ float sh[256 * 2 * 8];
#pragma unroll
    for (int i = 0; i < VN; ++i) { sh[(tid * VN + i) * 2] = dga[i]; sh[(tid * VN + i) * 2 + 1] = dbe[i]; }
    MU_SYNC_DMA();
    for (int cc = tid; cc < D; cc += 256) {
        double sa = 0.0, sb = 0.0;
        const int lcc = cc / VN, ii = cc % VN;
        for (int k = 0; k < RPI; ++k) {
            const int t = k * LPR + lcc;
            sa += (double)sh[(t * VN + ii) * 2];
            sb += (double)sh[(t * VN + ii) * 2 + 1];
        }
        part[((long)blockIdx.x * D + cc) * 2] = sa;
        part[((long)blockIdx.x * D + cc) * 2 + 1] = sb;
    }
    if (amax_part) {
        __shared__ float wam[4];
        const bool nan = __any(am != am);
        am = nan ? __builtin_nanf("") : wave_max(am);
        if ((tid & 63) == 0) wam[tid >> 6] = am;
        __syncthreads();
        if (tid == 0) {
            float m = wam[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) m = (m == m && wam[w] == wam[w]) ? fmaxf(m, wam[w]) : __builtin_nanf("");
            amax_part[blockIdx.x] = m;
        }
    }
}

// out = LayerNorm_{first cv channels}(oattn + x) * gamma + beta, pad channels 0: redoes the forward kernel's fused epilogue (which
// normalises over all D channels) for channel counts that run zero-padded -- standalone Mask2FormerAttention(channels, ...) with a
// width the UNet does not use (ade_semantic.py:153-161 accepts any); not on the model's path.
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_ln_fwd_kernel(const T* __restrict__ oattn, const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ out, float* __restrict__ ln_mean,
                                                          float* __restrict__ ln_rstd, long rows, int cv, float eps) {
    constexpr int VN = AT<T>::VN, LPR = D / VN, RPI = 256 / LPR;
    const int lc = threadIdx.x % LPR, lr = threadIdx.x / LPR, c = lc * VN;
    for (long rb = (long)blockIdx.x * RPI; rb < rows; rb += (long)gridDim.x * RPI) {
        const long r = rb + lr;
        const bool ok = r < rows;
        const long rr = ok ? r : 0;
        Vec16<T> ov, xv, yo;
        ov.load(oattn + rr * D + c); xv.load(x + rr * D + c);
        float y[VN], sum = 0.f;
#pragma unroll
        for (int i = 0; i < VN; ++i) { y[i] = (c + i < cv) ? ov.get(i) + xv.get(i) : 0.f; sum += y[i]; }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)cv;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < VN; ++i) { const float d = (c + i < cv) ? y[i] - mean : 0.f; sq += d * d; }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) sq += __shfl_xor(sq, o);
        const float rstd = rsqrtf(sq / (float)cv + eps);
#pragma unroll
        for (int i = 0; i < VN; ++i) yo.set(i, (c + i < cv) ? (y[i] - mean) * rstd * gamma[c + i] + beta[c + i] : 0.f);
        if (ok) {
            yo.store(out + r * D + c);
            if (lc == 0) { ln_mean[r] = mean; ln_rstd[r] = rstd; }
        }
    }
}

// Power-of-two scale of the fp16-pair-encoded dY: gs * max|dY| in [2^-3, 2^-2).  Headroom: |dP'| = |gs (dO . V - delta) / sqrt(C)| <=
// 2^-2 (sqrt(C) max|V| + ...), times 2^pshift P <= 2^12 -- dS overflows fp16 only where P ~ 1 meets |V| >~ 16 in every channel (and
// then surfaces as inf / NaN gradients, never silently); typical values sit 10-20 binades above fp16's subnormal floor.
__device__ __forceinline__ float attn_gscale_from_amax(float amax) {
    if (!(amax == amax)) return amax;                        // NaN gradient: the scale is NaN and so is everything downstream
    int e = (int)(__float_as_uint(amax) >> 23) - 127;        // floor(log2(amax)) for normal values; -127 for 0 / subnormals
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return __uint_as_float((uint32_t)(127 - 3 - e) << 23);
}
__global__ void attn_ln_bwd_final_kernel(const double* __restrict__ part, int nblk, int D, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                         const float* __restrict__ amax_part, float* __restrict__ gscale) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (amax_part && c == 0) {                               // fp32x: max|dY| over the prepass blocks -> the scale of the encoded dY
        float m = 0.f;
        bool nan = false;
        for (int k = lane; k < nblk; k += 64) { const float v = amax_part[k]; nan = nan || v != v; m = fmaxf(m, v); }
        nan = __any(nan);
        m = wave_max(m);
        if (lane == 0) *gscale = attn_gscale_from_amax(nan ? __builtin_nanf("") : m);
    }
    if (c >= D) return;
    double av[16], bv[16];           // nblk <= ATT_LN_MAXBLK = 1024: every load in flight before the first add (one L2 round trip, not 16)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int k = lane + 64 * u;
        const double2 v = k < nblk ? *reinterpret_cast<const double2*>(part + ((long)k * D + c) * 2) : make_double2(0.0, 0.0);
        av[u] = v.x; bv[u] = v.y;
    }
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) { a += av[u]; b += bv[u]; }
    a = wave_sum_d(a); b = wave_sum_d(b);
    if (lane) return;
    dgamma[c] = (float)a;
    dbeta[c] = (float)b;
}



// fp32x: dY -> the fp16-pair encoding of gs dY the two sweeps read (common.h), gs from attn_ln_bwd_final_kernel; the thread that owns a
// row's first group also scales the row's -delta / sqrt(C) constant of the dK/dV sweep by gs (the dQ sweep scales delta itself).
__global__ __launch_bounds__(256) void attn_dy_encode_kernel(const f32x4* __restrict__ dY, uint4* __restrict__ dYs, float* __restrict__ rowc,
                                                             const float* __restrict__ gscale, long ngroups, int gpr, int N) {
    const float gs = *gscale;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ngroups; i += (long)gridDim.x * 256) {
        const f32x4 a = __builtin_nontemporal_load(dY + 2 * i) * gs, b = __builtin_nontemporal_load(dY + 2 * i + 1) * gs;
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        const SplitH8 e = mu_hsplit8(v);
        dYs[2 * i] = __builtin_bit_cast(uint4, e.hi);
        dYs[2 * i + 1] = __builtin_bit_cast(uint4, e.lo);
        if (i % gpr == 0) {
            const long r = i / gpr, bb = r / N;
            const int qn = (int)(r - bb * N), ntile = (N + 31) >> 5;
            rowc[((bb * ntile + (qn >> 5)) * 2) * 32 + (qn & 31) + 32] *= gs;
        }
    }
}

// fp32x operand encoding of qkv (C ABI: mu_split_encode_h): n_elems fp32 values (a multiple of 8) -> [8 fp16 hi | 8 fp16 lo] per
// aligned 32-byte group, in place or into dst
__global__ __launch_bounds__(256) void split_encode_h_kernel(const f32x4* src, uint4* dst, long ngroups) {      // (may alias: no __restrict__)
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ngroups; i += 2 * stride) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (i + u * stride < ngroups) { a[u] = __builtin_nontemporal_load(src + 2 * (i + u * stride)); b[u] = __builtin_nontemporal_load(src + 2 * (i + u * stride) + 1); }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (i + u * stride < ngroups) {
                const float v[8] = {a[u][0], a[u][1], a[u][2], a[u][3], b[u][0], b[u][1], b[u][2], b[u][3]};
                const SplitH8 e = mu_hsplit8(v);
                dst[2 * (i + u * stride)] = __builtin_bit_cast(uint4, e.hi);
                dst[2 * (i + u * stride) + 1] = __builtin_bit_cast(uint4, e.lo);
            }
    }
}
extern "C" int mu_split_encode_h(const void* src, void* dst, long n_elems, void* stream) {
    if (!src || !dst || n_elems <= 0 || n_elems % 8) return MU_ERR_ARG;
    const long ng = n_elems / 8;
    long g = (ng + 511) / 512;
    g = g < 1 ? 1 : (g > 8192 ? 8192 : g);
    split_encode_h_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>((const f32x4*)src, (uint4*)dst, ng);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// backward v2 kernels: LDS-DMA double-buffered tiles, swizzled images (see attn_fwd2_kernel)
// ------------------------------------------------------------------------------------------
// NQT = 16-query tiles per wave (2 everywhere but C = 128: there one tile per wave halves the resident Q / dO fragments and the dQ
// accumulators -- 222 -> ~130 registers in fp16, two -> three waves per SIMD)
template <typename T, int D, int KT, int NW, int NQT = 2, int OCCQ = 0>
__global__ __launch_bounds__(NW * 64, OCCQ ? OCCQ : (NW == 4 && D <= 64 && std::is_same<T, xf32>::value) ? MU_XF_DQ_OCC : (NW == 4 && D <= 64 && sizeof(T) == 2) ? MU_DQ_OCC : ((NW == 4 && D == 128 && sizeof(T) == 2) ? MU_DQ_OCC128 : ((NW == 4 && D == 256 && sizeof(T) == 2) ? MU_DQ_OCC256 : 1))) void attn_bwd_dq2_kernel(const T* __restrict__ qkv, const T* __restrict__ dY, const int* __restrict__ kidx,
                                                           const int* __restrict__ kcnt, const float* __restrict__ lse2,
                                                           const float* __restrict__ delta, T* __restrict__ dqkv, int N, int nkmax,
                                                           float scale, float scale_log2, const float* __restrict__ gsp, float pshift, int enc_out) {
    using A = AT<T>;
    using Frag = typename A::Frag;
    using Z = SwzTile<T, D>;
    constexpr int VN = A::VN, KR = A::KR, NKS = D / KR, NDT = D / 16, NKT = KT / 16, NH = KT / 32;
    constexpr bool MU_PRIO_BWD = true;
    constexpr bool XF = std::is_same<T, xf32>::value;
    constexpr int TEK = Z::TE(KT);
    __shared__ __attribute__((aligned(16))) T lds[2 * 2 * TEK];
    // fp32x: dY arrives scaled by the power of two gs (attn_bwd_t), the probabilities are computed as 2^pshift P: dS -- ONE fp16 operand --
    // then sits inside fp16's exponent range; the accumulators are un-scaled once, in the epilogue (exact: powers of two)
    float gs = 1.0f, un = 1.0f;
    if constexpr (XF) {
        gs = *gsp;
        un = 1.0f / (gs * exp2f(pshift));
    }

    int bx_, b;
    attn_block(bx_, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int q0 = bx_ * (NW * NQT * 16) + wave * (NQT * 16);
    const T* qkv_b = qkv + (long)b * N * 3 * D;
    const int Nk = kcnt[b];
    const int* kidx_b = kidx + (long)b * nkmax;

    KvStage<T, D, KT, NW> stg;
    stg.load_idx(kidx_b, 0, Nk, wave, lane);
    stg.issue(lds, lds + TEK, qkv_b, wave, lane);
    stg.load_idx(kidx_b, KT, Nk, wave, lane);            // indices of tile 1, consumed inside iteration 0

    // Q pre-scaled by log2(e)/sqrt(C) and dO by 1/sqrt(C): with the row constants -lse2 and -delta/sqrt(C) as the
    // MFMA C operands, the matrix core hands back the exponent of P and the scaled (dP - delta) directly.
    Frag qf[NQT][NKS], dof[NQT][NKS];
    f32x4 nlse[NQT], ndel[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        int qrow = q0 + t * 16 + r16;
        if (qrow > N - 1) qrow = N - 1;
        const long tok = (long)b * N + qrow;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[t][ks] = A::ld_scaled(qkv_b + (long)qrow * 3 * D + ks * KR + g * A::GS, scale_log2);
            dof[t][ks] = A::ld_scaled(dY + tok * D + ks * KR + g * A::GS, scale);
        }
        const float l = (XF ? pshift : 0.f) - lse2[tok], d = -delta[tok] * scale * gs;
        nlse[t] = (f32x4){l, l, l, l};
        ndel[t] = (f32x4){d, d, d, d};
    }
    f32x4 dq[NDT][NQT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int t = 0; t < NQT; ++t) dq[dt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    MU_SYNC_DMA();

    auto tile = [&](auto BUFC, int j0) {
        constexpr int BUF = decltype(BUFC)::value;
        const T* Kt = lds + BUF * 2 * TEK;
        if constexpr (XF && MU_XF_OPAQUE) Kt = lds_opaque(Kt);
        const T* Vt = Kt + TEK;
        if (j0 + KT < Nk) {
            T* Kn = lds + (BUF ^ 1) * 2 * TEK;
            stg.issue(Kn, Kn + TEK, qkv_b, wave, lane);
            stg.load_idx(kidx_b, j0 + 2 * KT, Nk, wave, lane);
        }
        f32x4 s[NKT][NQT], dp[NKT][NQT];
        MU_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                Frag ka = A::template ldt<Z>(Kt, kt * 16 + r16, ks * KR + g * A::GS);
                Frag va = A::template ldt<Z>(Vt, kt * 16 + r16, ks * KR + g * A::GS);
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    if (ks == 0) {                            // row constants as the C operand of the first k-step (no copies)
                        s[kt][t] = A::mma_row_from_bs_sa(ka, qf[t][0], nlse[t]);
                        dp[kt][t] = A::mma_row_from_bp_sa(va, dof[t][0], ndel[t]);
                    } else {
                        A::mma_row_bs_sa(ka, qf[t][ks], s[kt][t]);
                        A::mma_row_bp_sa(va, dof[t][ks], dp[kt][t]);
                    }
                }
            }
        MU_PRIO(0);
        if (j0 + KT > Nk) {                          // wave-uniform: only the last, partial tile pays for key-range masking
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (j0 + kt * 16 + 4 * g + r >= Nk) { s[kt][t][r] = -INFINITY; dp[kt][t][r] = 0.f; }
        }
        constexpr int PREQ = (MU_DQ_PREFETCH && sizeof(T) == 2 && NH == 1) ? (MU_DQ_PREFETCH < NDT ? MU_DQ_PREFETCH : NDT) : 0;
        typename A::AccA kap[PREQ ? PREQ : 1];
        if constexpr (PREQ > 0) {                    // transposed K operands of the dQ product: in flight during the exponentials
#pragma unroll
            for (int dt = 0; dt < PREQ; ++dt) kap[dt] = AccLd<T, D>::ld(Kt, 0, dt * 16, g, r16);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][t][r] = __builtin_amdgcn_exp2f(s[kt][t][r]) * dp[kt][t][r];
        if constexpr (PREQ > 0) __builtin_amdgcn_sched_barrier(0);
        MU_PRIO(1);
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                typename A::AccA ka;
                if (h == 0 && dt < PREQ) ka = kap[dt];
                else ka = AccLd<T, D>::ld(Kt, 32 * h, dt * 16, g, r16);
#pragma unroll
                for (int t = 0; t < NQT; ++t) A::mma_accb(ka, s[2 * h][t], s[2 * h + 1][t], dq[dt][t]);
            }
        MU_PRIO(0);
#ifdef MU_DQ_ABL_NOBAR
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        MU_SYNC_DMA();
#endif
    };
    for (int j0 = 0; j0 < Nk; j0 += 2 * KT) {
        tile(std::integral_constant<int, 0>{}, j0);
        if (j0 + KT < Nk) tile(std::integral_constant<int, 1>{}, j0 + KT);
    }
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const int qrow = q0 + t * 16 + r16;
        if (qrow >= N) continue;
        T* dst = dqkv + ((long)b * N + qrow) * 3 * D;
        if (Nk == 0) {
            // an image with NO visible key: the reference's softmax over -inf only is NaN for every query (ade_semantic.py:183-185), and so
            // are all three gradients; this sweep visits every token row once, so it writes the NaN dQ, dK and dV of the row
            const float nanv[4] = {__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
#pragma unroll
            for (int c = 0; c < 3 * NDT; ++c) store4e<T>(dst + c * 16 + 4 * g, nanv, enc_out);
            continue;
        }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            float v[4] = {dq[dt][t][0], dq[dt][t][1], dq[dt][t][2], dq[dt][t][3]};
            if constexpr (XF) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= un;
            }
            store4e<T>(dst + dt * 16 + 4 * g, v, enc_out);
        }
    }
}



// ------------------------------------------------------------------------------------------
// dK/dV v3: the query sweep is a 4-deep LDS ring fed by LDS-DMA three tiles ahead with COUNTED vmcnt waits and one
// raw s_barrier per tile.  With one-tile-ahead double buffering (v2) every barrier waited for a DMA issued only
// ~0.7 us earlier -- shorter than the L2/HBM round trip -- so the matrix cores idled at each tile.  Every
// vector-memory op in the loop is an LDS-DMA (the row constants -lse2 and -delta/sqrt(C) come in through the same
// ring), so the counted waits are exact: each wave issues exactly 3 DMA instructions per tile.
// ------------------------------------------------------------------------------------------
// NW = waves per block (16 * NKT keys each).  The Q / dO stream a block pulls through L2 -> LDS is shared by NW * NKT * 16 keys: at
// C = 128 with 4 waves of 16 keys every launch moved 4.2 GB in 0.7 ms (6 TB/s, the L2 -> LDS ceiling) -- 8 waves halve that.
template <typename T, int D, int NKT, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 1 : ((D <= 64 && std::is_same<T, xf32>::value) ? MU_XF_OCC : (D <= 64 && sizeof(T) == 2 && NKT <= 2) ? MU_DKV_OCC : ((D == 128 && sizeof(T) == 2 && NKT <= 2) ? MU_DKV_OCC128 : ((D == 256 && sizeof(T) == 2 && NKT == 1) ? MU_DKV_OCC256 : 1)))) void attn_bwd_dkv3_kernel(
    const T* __restrict__ qkv, const T* __restrict__ dY, const int* __restrict__ kidx, const int* __restrict__ kcnt,
    const float* __restrict__ rowc, T* __restrict__ dqkv, int N, int nkmax, float scale, float scale_log2, int zero_masked,
    const float* __restrict__ gsp, float pshift) {
    using A = AT<T>;
    using Frag = typename A::Frag;
    using Z = SwzTile<T, D>;
    constexpr int VN = A::VN, KR = A::KR, NKS = D / KR, NDT = D / 16, QT = 32;
    constexpr bool MU_PRIO_BWD = true;
    constexpr int NI = QT / Z::RPW;                          // DMA wave-instructions per tensor per tile
    static_assert(NI == 4 || NI == 8 || NI == 16 || NI == 32 || NI == 2, "unexpected tile geometry");
    constexpr int NPW = (NI + NW - 1) / NW;                  // per wave (Q and dO each)
    constexpr int TEQ = Z::TE(QT);                           // elements of one Q (or dO) tile image
    constexpr int STG = 2 * TEQ;                             // elements per ring slot (Q | dO)
    constexpr int DKV_RING = (STG * (int)sizeof(T) <= 36864) ? 4 : 2;      // ring depth; prefetch distance = depth - 1 (36864: the padded fp32x image at C = 128)
    __shared__ __attribute__((aligned(16))) T lds[DKV_RING * STG];
    __shared__ __attribute__((aligned(16))) float rcs[DKV_RING * 256 + 256];   // 1 KB per slot: [{-lse2},{-delta*scale}][32] in its first 256 B; + 1 KB dump

    int bx_, b;
    attn_block(bx_, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int Nk = kcnt[b];
    const int kb0 = bx_ * (NW * NKT * 16);
    const int* kidx_b = kidx + (long)b * nkmax;
    if (Nk == 0) return;                                      // no visible key at all: NaN gradients, written by the dQ sweep (see there)
    if (kb0 >= Nk) {
        // zero_masked (kidx rows are whole permutations, the masked keys listed after the kept ones): this block's keys are all
        // masked -- their dK / dV rows are exact zeros, written here instead of by a memset of the whole dqkv buffer
        if (zero_masked & 1) {
            constexpr int LPK = 2 * D * (int)sizeof(T) / 16;             // 16-byte lanes per key row (K and V parts are adjacent)
            for (int i = threadIdx.x; i < NW * NKT * 16 * LPK; i += NW * 64) {
                const int j = kb0 + i / LPK;
                if (j < nkmax) {
                    T* dst = dqkv + ((long)b * N + kidx_b[j]) * 3 * D + D;
                    reinterpret_cast<float4*>(dst)[i % LPK] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        return;
    }
    const T* qkv_b = qkv + (long)b * N * 3 * D;
    const T* dY_b = dY + (long)b * N * D;
    const int ntile = (N + QT - 1) / QT;
    const float* rowc_b = rowc + (long)b * ntile * 64;

    // exactly 2*NPW + 1 DMA instructions per wave per tile.  Per-lane source offsets are loop-invariant (precomputed); a tile
    // only moves the scalar base.  (The generic index arithmetic made the three DMAs cost ~250 issue cycles per tile.)
    int qlane[NPW], olane[NPW], rowl[NPW];
    {
        const int lrow = Z::lrow(lane);
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int i = wave + NW * n;
            const int ii = i < NI ? i : 0;                   // (NI >= 4 for every instantiation: never clamps)
            const int row = ii * Z::RPW + lrow;
            const int sc = Z::schunk(row, lane);
            rowl[n] = row;
            qlane[n] = row * 3 * D + sc * Z::VN;
            olane[n] = row * D + sc * Z::VN;
        }
    }
    auto issue = [&](int tile) {
        const int slot = tile % DKV_RING;
        T* Qt = lds + slot * STG;
        T* Ot = Qt + TEQ;
        const T* qb = qkv_b + (long)tile * QT * 3 * D;       // wave-uniform bases
        const T* ob = dY_b + (long)tile * QT * D;
        const bool full = tile * QT + QT <= N;               // wave-uniform: only the last tile can be partial
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            if (NW > NI && wave + NW * n >= NI) break;       // blocks with more waves than pieces: the extra waves issue nothing (and
            const int ii = (wave + NW * n) < NI ? (wave + NW * n) : 0;      // their vmcnt waits pass at once; the barrier orders them)
            int qo = qlane[n], oo = olane[n];
            if (!full && tile * QT + rowl[n] >= N) {          // rows past N: re-read row N-1 (finite); their row constants
                const int back = tile * QT + rowl[n] - (N - 1);      // (-inf, 0) zero the probabilities
                qo -= back * 3 * D;
                oo -= back * D;
            }
            glds16s(qb, (uint32_t)(qo * (int)sizeof(T)), Qt + ii * Z::BLK);
            glds16s(ob, (uint32_t)(oo * (int)sizeof(T)), Ot + ii * Z::BLK);
        }
        // row constants: 16 lanes x 16 B = the tile's 64 floats.  Every wave issues one DMA so that all waves count the same
        // number of vector-memory ops: wave 0's lanes >= 16 repeat the constants into the unused rest of the slot, waves 1-3
        // write theirs to a dump area
#if MU_DKV_ROWC_ONE
        // only wave 0 moves the row constants; the other waves count one DMA less per tile (wave-uniform branch on the waits below)
        if (wave == 0) glds16s(rowc_b + (long)tile * 64, (uint32_t)((lane & 15) * 16), rcs + slot * 256);
#else
        if (NW <= NI || wave < NI)
            glds16s(rowc_b + (long)tile * 64, (uint32_t)((lane & 15) * 16), rcs + (wave == 0 ? slot * 256 : DKV_RING * 256));
#endif
    };
    constexpr int OPS = 2 * NPW + 1;

    issue(0);
    if (DKV_RING == 4) {
        if (ntile > 1) issue(1);
        if (ntile > 2) issue(2);
#ifdef MU_DKV_ABL_NODMA
        if (ntile > 3) issue(3);                             // timing-only ablation: all four slots hold real data, no DMA in the loop
#endif
    }

    Frag kf[NKT][NKS], vf[NKT][NKS];
    int keyrow[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int j = kb0 + (wave * NKT + kt) * 16 + r16;
        keyrow[kt] = j < Nk ? kidx_b[j] : -1;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            Frag fk = A::zero(), fv = A::zero();
            if (keyrow[kt] >= 0) {
                fk = A::ld_scaled(qkv_b + (long)keyrow[kt] * 3 * D + D + ks * KR + g * A::GS, scale_log2);
                fv = A::ld_scaled(qkv_b + (long)keyrow[kt] * 3 * D + 2 * D + ks * KR + g * A::GS, scale);
            }
            kf[kt][ks] = fk;
            vf[kt][ks] = fv;
        }
    }
    f32x4 dk[NDT][NKT], dv[NDT][NKT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) { dk[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float un = 1.0f;
    if constexpr (std::is_same<T, xf32>::value) {
        un = 1.0f / (*gsp * exp2f(pshift));
        asm volatile("" : "+v"(un));                         // consumed HERE: no ordinary load may be pending under the counted waits below
    }
    // the prologue's ordinary loads (kidx, K, V) are complete here: the frags were consumed by the scaling above

    auto tile = [&](auto SLOTC, int tl) {
        constexpr int SLOT = decltype(SLOTC)::value;
        // tile tl's DMAs were issued RING-1 issue-groups ago; newer groups still in flight: min(RING-2, tiles left after tl)
        const int newer = ntile - 1 - tl;
#ifdef MU_DKV_ABL_NODMA
        if (tl == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
#endif
#if MU_DKV_ROWC_ONE
        if (DKV_RING == 4 && newer >= 2) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (OPS - 1)) : "memory");
        } else if (DKV_RING == 4 && newer == 1) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS - 1) : "memory");
        }
#else
        if (DKV_RING == 4 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
        else if (DKV_RING == 4 && newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
#endif
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef MU_DKV_ABL_NOBAR
        __builtin_amdgcn_s_barrier();                        // everyone's share of tile tl landed; tile tl-1 fully consumed
#endif
#ifdef MU_DKV_ISSUE_FIRST
        if (tl + DKV_RING - 1 < ntile) issue(tl + DKV_RING - 1);
#endif
        const T* Qt = lds + SLOT * STG;
        if constexpr ((std::is_same<T, xf32>::value && MU_XF_OPAQUE) || (sizeof(T) == 2 && D >= 128 && MU_H16_OPAQUE)) Qt = lds_opaque(Qt);
        const T* Ot = Qt + TEQ;
        const float* rc = rcs + SLOT * 256;
        f32x4 s[2][NKT], dp[2][NKT];
        f32x4 nl[2], nd[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            nl[qt] = *reinterpret_cast<const f32x4*>(rc + qt * 16 + 4 * g);
            nd[qt] = *reinterpret_cast<const f32x4*>(rc + 32 + qt * 16 + 4 * g);
        }
        MU_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                Frag qa = A::template ldt<Z>(Qt, qt * 16 + r16, ks * KR + g * A::GS);
                Frag oa = A::template ldt<Z>(Ot, qt * 16 + r16, ks * KR + g * A::GS);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    if (ks == 0) {                           // the row constants enter as the C operand of the first k-step
                        s[qt][kt] = A::mma_row_from_bs_sb(qa, kf[kt][0], nl[qt]);
                        dp[qt][kt] = A::mma_row_from_bp_sb(oa, vf[kt][0], nd[qt]);
                    } else {
                        A::mma_row_bs_sb(qa, kf[kt][ks], s[qt][kt]);
                        A::mma_row_bp_sb(oa, vf[kt][ks], dp[qt][kt]);
                    }
                }
                if constexpr (std::is_same<T, xf32>::value && D >= 128 && MU_XF_DKV_SCHED2) { if (qt == 1) __builtin_amdgcn_sched_barrier(0); }
            }
        MU_PRIO(0);
        if (tl * QT + QT > N) {                              // last, partial tile: padded queries contribute nothing
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (tl * QT + qt * 16 + 4 * g + r >= N) {
#pragma unroll
                        for (int kt = 0; kt < NKT; ++kt) { s[qt][kt][r] = -INFINITY; dp[qt][kt][r] = 0.f; }
                    }
        }
        // The transposed dO / Q operands of the dV / dK products are read from LDS BEFORE the exponentials (PRE of the NDT column
        // blocks): their latency then hides behind the VALU phase instead of standing, once per column block, between the MFMAs
        // (the compiler otherwise issues each block's reads right in front of its MFMAs: ~60 idle cycles per block per wave).
        constexpr int PFN = D == 128 ? MU_DKV_PREFETCH128 : MU_DKV_PREFETCH;
        constexpr int PRE = (PFN && sizeof(T) == 2) ? (PFN < NDT ? PFN : NDT) : 0;
        typename A::AccA oap[PRE ? PRE : 1], qap[PRE ? PRE : 1];
        if constexpr (PRE > 0) {
#pragma unroll
            for (int dt = 0; dt < PRE; ++dt) {
                oap[dt] = AccLd<T, D>::ld(Ot, 0, dt * 16, g, r16);
                qap[dt] = AccLd<T, D>::ld(Qt, 0, dt * 16, g, r16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // MU_DKV_PKMUL (fp16 storage): P and dP' are rounded to fp16 first -- both are fp16 MFMA operands' worth of precision anyway --
        // and dS = P * dP' is FOUR packed fp16 multiplies per 8 scores instead of 8 fp32 ones: 24 instead of 32 VALU instructions per
        // 32-query x 32-key tile behind the 16 exponentials (the sweep runs at MFMA + VALU issue time)
        // fp32x takes the same path (the row constants carry the power-of-two scales that keep P and dP' inside fp16's range: attn_bwd_t)
        constexpr bool PKMUL = (MU_DKV_PKMUL && sizeof(T) == 2) || (std::is_same<T, xf32>::value && D < MU_XF_PK_MAXD);
        if constexpr (PKMUL) {
            typename A::Packed pb[NKT], db[NKT];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qt][kt][r] = __builtin_amdgcn_exp2f(s[qt][kt][r]);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                pb[kt] = A::pack(s[0][kt], s[1][kt]);
                db[kt] = A::pack(dp[0][kt], dp[1][kt]) * pb[kt];
            }
            if constexpr (PRE > 0) __builtin_amdgcn_sched_barrier(0);
            MU_PRIO(1);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                typename A::AccA oa, qa;
                if (dt < PRE) { oa = oap[dt]; qa = qap[dt]; }
                else { oa = AccLd<T, D>::ld(Ot, 0, dt * 16, g, r16); qa = AccLd<T, D>::ld(Qt, 0, dt * 16, g, r16); }
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    A::mma_accb_pk(oa, pb[kt], dv[dt][kt]);
                    A::mma_accb_pk(qa, db[kt], dk[dt][kt]);
                }
                // fp32x at C >= 128: the resident K / V pairs and the accumulators alone are 128 registers; left alone the scheduler
                // hoists the transposed reads of ALL column blocks (16 registers each) above the first MFMA and spills ~130 registers
                if constexpr (std::is_same<T, xf32>::value && D >= 128 && MU_XF_DKV_SCHED) { if (dt % MU_XF_DKV_SCHED == MU_XF_DKV_SCHED - 1) __builtin_amdgcn_sched_barrier(0); }
            }
            MU_PRIO(0);
        } else {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[qt][kt][r]);
                    s[qt][kt][r] = p;
                    dp[qt][kt][r] *= p;
                }
        if constexpr (PRE > 0) __builtin_amdgcn_sched_barrier(0);
        MU_PRIO(1);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            typename A::AccA oa, qa;
            if (dt < PRE) { oa = oap[dt]; qa = qap[dt]; }
            else { oa = AccLd<T, D>::ld(Ot, 0, dt * 16, g, r16); qa = AccLd<T, D>::ld(Qt, 0, dt * 16, g, r16); }
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                A::mma_accb(oa, s[0][kt], s[1][kt], dv[dt][kt]);
                A::mma_accb(qa, dp[0][kt], dp[1][kt], dk[dt][kt]);
            }
        }
        MU_PRIO(0);
        }
        // Refill the slot tile tl-1 vacated -- issued LAST in the tile: LDS reads queue behind an in-flight LDS-DMA issue
        // (in-kernel s_memtime stamps: the row-constant reads right after the DMA cost ~980 cycles/tile, ~80 without it)
#if !defined(MU_DKV_ISSUE_FIRST) && !defined(MU_DKV_ABL_NODMA)
        if (tl + DKV_RING - 1 < ntile) issue(tl + DKV_RING - 1);
#endif
    };
    for (int tl = 0; tl < ntile; tl += DKV_RING) {
        tile(std::integral_constant<int, 0>{}, tl);
        if (tl + 1 < ntile) tile(std::integral_constant<int, 1>{}, tl + 1);
        if constexpr (DKV_RING == 4) {
            if (tl + 2 < ntile) tile(std::integral_constant<int, 2>{}, tl + 2);
            if (tl + 3 < ntile) tile(std::integral_constant<int, 3>{}, tl + 3);
        }
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        int row = keyrow[kt];
        const bool live = row >= 0;
        if (!live) {                                         // masked key inside the last kept block: exact zeros (see above)
            const int j = kb0 + (wave * NKT + kt) * 16 + r16;
            if (!(zero_masked & 1) || j >= nkmax) continue;
            row = kidx_b[j];
        }
        T* dst = dqkv + ((long)b * N + row) * 3 * D;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            float kv[4] = {dk[dt][kt][0], dk[dt][kt][1], dk[dt][kt][2], dk[dt][kt][3]};
            float vv[4] = {dv[dt][kt][0], dv[dt][kt][1], dv[dt][kt][2], dv[dt][kt][3]};
            if constexpr (std::is_same<T, xf32>::value) {            // the sweep ran on gs dY and 2^pshift P (see attn_bwd_dq2_kernel)
#pragma unroll
                for (int r = 0; r < 4; ++r) { kv[r] *= un; vv[r] *= un; }
            }
            if (!live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { kv[r] = 0.f; vv[r] = 0.f; }
            }
            store4e<T>(dst + D + dt * 16 + 4 * g, kv, zero_masked & 2);      // (bit 1 of zero_masked: dqkv written chunk-encoded, fp32x)
            store4e<T>(dst + 2 * D + dt * 16 + 4 * g, vv, zero_masked & 2);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------
template <typename T>
static int attn_fwd_t(const T* qkv, const T* x, const int* kidx, const int* kcnt, const float* gamma, const float* beta, T* out,
                      T* oattn, float* lse2, float* mean, float* rstd, int B, int N, int C, int cv, int nkmax, float eps, hipStream_t st) {
    const float sl2 = (float)(1.4426950408889634 / sqrt((double)cv));      // 1/sqrt(channels) of the TRUE channel count (ade_semantic.py:174)
    // (measured and removed: 8-wave blocks sharing each K/V tile among 256 queries, -8 %; 64 queries per wave against 32-key tiles,
    //  -4 %; 128- / 32-key tiles and other occupancy bounds, +-0: the sweep is bound by MFMA + VALU issue time)
#define LAUNCH_FWD(DD, KT) \
    attn_fwd2_kernel<T, DD, KT, 4><<<dim3(mu_cdiv(N, 128), B), 256, 0, st>>>(qkv, x, kidx, kcnt, gamma, beta, out, oattn, lse2, mean, rstd, N, nkmax, sl2, eps)
    switch (C) {
        case 32: LAUNCH_FWD(32, 64); break;
        case 64:
            if constexpr (std::is_same<T, xf32>::value && MU_XF_FWD_KT32)       // 32-key tiles: 32 KB of LDS and <= 168 registers -> three waves per SIMD
                attn_fwd2_kernel<T, 64, 32, 4, 3><<<dim3(mu_cdiv(N, 128), B), 256, 0, st>>>(qkv, x, kidx, kcnt, gamma, beta, out, oattn, lse2, mean, rstd, N, nkmax, sl2, eps);
            else LAUNCH_FWD(64, 64);
            break;
        case 128:
#if MU_FWD_NW128 == 8
            attn_fwd2_kernel<T, 128, MU_FWD_KT128, 8, 1><<<dim3(mu_cdiv(N, 256), B), 512, 0, st>>>(qkv, x, kidx, kcnt, gamma, beta, out, oattn, lse2, mean, rstd, N, nkmax, sl2, eps);
#else
            LAUNCH_FWD(128, MU_FWD_KT128);
#endif
            break;
        case 256: LAUNCH_FWD(256, 32); break;
        default: return MU_ERR_SHAPE;
    }
#undef LAUNCH_FWD
    if (cv != C) {           // zero-padded channels: LayerNorm over the first cv channels only (re-does the fused epilogue's out / mean / rstd)
        using TS = typename std::conditional<std::is_same<T, xf32>::value, float, T>::type;      // storage type (no matrix product here)
        const TS *oa = (const TS*)oattn, *xs = (const TS*)x;
        TS* os = (TS*)out;
        const long rows = (long)B * N;
        const int nb = (int)(rows / 64 < 1 ? 1 : (rows / 64 > 4096 ? 4096 : rows / 64));
        switch (C) {
            case 32: attn_ln_fwd_kernel<TS, 32><<<nb, 256, 0, st>>>(oa, xs, gamma, beta, os, mean, rstd, rows, cv, eps); break;
            case 64: attn_ln_fwd_kernel<TS, 64><<<nb, 256, 0, st>>>(oa, xs, gamma, beta, os, mean, rstd, rows, cv, eps); break;
            case 128: attn_ln_fwd_kernel<TS, 128><<<nb, 256, 0, st>>>(oa, xs, gamma, beta, os, mean, rstd, rows, cv, eps); break;
            default: attn_ln_fwd_kernel<TS, 256><<<nb, 256, 0, st>>>(oa, xs, gamma, beta, os, mean, rstd, rows, cv, eps); break;
        }
    }
    return MU_OK;
}

extern "C" int mu_attn_fwd_padded(const void* qkv, const void* x, const int* kidx, const int* kcnt, const float* gamma, const float* beta,
                                  void* out, void* oattn, float* lse2, float* ln_mean, float* ln_rstd, int B, int N, int C, int c_valid,
                                  int nkmax, float eps, int dtype, void* stream) {
    if (!qkv || !x || !kidx || !kcnt || !gamma || !beta || !out || !oattn || !lse2 || !ln_mean || !ln_rstd) return MU_ERR_ARG;
    if (B <= 0 || N <= 0 || nkmax <= 0 || c_valid <= 0 || c_valid > C) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == MU_F16) rc = attn_fwd_t<h16>((const h16*)qkv, (const h16*)x, kidx, kcnt, gamma, beta, (h16*)out, (h16*)oattn, lse2, ln_mean, ln_rstd, B, N, C, c_valid, nkmax, eps, st);
    else if (dtype == MU_F32) rc = attn_fwd_t<float>((const float*)qkv, (const float*)x, kidx, kcnt, gamma, beta, (float*)out, (float*)oattn, lse2, ln_mean, ln_rstd, B, N, C, c_valid, nkmax, eps, st);
    else if (dtype == MU_F32X) rc = attn_fwd_t<xf32>((const xf32*)qkv, (const xf32*)x, kidx, kcnt, gamma, beta, (xf32*)out, (xf32*)oattn, lse2, ln_mean, ln_rstd, B, N, C, c_valid, nkmax, eps, st);
    else return MU_ERR_ARG;
    if (rc) return rc;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_attn_fwd(const void* qkv, const void* x, const int* kidx, const int* kcnt, const float* gamma, const float* beta,
                           void* out, void* oattn, float* lse2, float* ln_mean, float* ln_rstd, int B, int N, int C, int nkmax,
                           float eps, int dtype, void* stream) {
    return mu_attn_fwd_padded(qkv, x, kidx, kcnt, gamma, beta, out, oattn, lse2, ln_mean, ln_rstd, B, N, C, C, nkmax, eps, dtype, stream);
}

#define ATT_LN_MAXBLK 1024
static inline long attn_ln_part_bytes(int C) { return (long)ATT_LN_MAXBLK * C * 2 * sizeof(double); }
static inline long attn_rowc_bytes(int B, int N) { return (long)B * ((N + 31) / 32) * 64 * sizeof(float); }
static inline long attn_amax_bytes() { return (long)(ATT_LN_MAXBLK + 64) * sizeof(float); }      // per-block max|dY| + the scale (fp32x)
// the sweeps compute 2^pshift P (fp32x): typical probabilities ~1/N land near 1, P <= 1 stays below 2^12
static inline float attn_pshift(int N) {
    int j = 0;
    while ((2 << j) <= N) ++j;                               // floor(log2(N))
    j -= 2;
    return (float)(j < 0 ? 0 : (j > 12 ? 12 : j));
}
// LayerNorm partials | max|dY| partials + scale | row constants of the dK/dV sweep | (MU_F32X) the fp16-pair-encoded copy of dY the two sweeps read
extern "C" long mu_attn_bwd_workspace_bytes(int B, int N, int C) {
    return attn_ln_part_bytes(C) + attn_amax_bytes() + attn_rowc_bytes(B, N) + (long)B * N * C * (long)sizeof(float);
}

template <typename T>
static int attn_bwd_t(const T* qkv, const T* x, const T* oattn, const T* gout, const int* kidx, const int* kcnt, const float* lse2,
                      const float* mean, const float* rstd, const float* gamma, T* dY, float* delta, T* dqkv, float* dgamma,
                      float* dbeta, int B, int N, int C, int cv, int nkmax, void* ws, hipStream_t st, int phases) {
    using TS = typename std::conditional<std::is_same<T, xf32>::value, float, T>::type;      // storage type for the LayerNorm prepass
    const long rows = (long)B * N;
    float* amaxp = (float*)((char*)ws + attn_ln_part_bytes(C));
    float* gsc = amaxp + ATT_LN_MAXBLK;
    float* rowc = (float*)((char*)ws + attn_ln_part_bytes(C) + attn_amax_bytes());
    constexpr bool XF = std::is_same<T, xf32>::value;
    const float pshift = XF ? attn_pshift(N) : 0.f;
    int nblk = (int)(rows / 64 < 1 ? 1 : (rows / 64 > ATT_LN_MAXBLK ? ATT_LN_MAXBLK : rows / 64));
    const float scale = (float)(1.0 / sqrt((double)cv));
    const float sl2 = (float)(1.4426950408889634 / sqrt((double)cv));
    dim3 gq(mu_cdiv(N, 128), B);
    // phases & 8 (MU_ATTN_KIDX_PERMUTATION): every kidx row is a whole permutation of 0..N-1 with the masked keys after the kept
    // ones, so the dK/dV sweep writes the masked keys' zero rows itself (dQ parts are written for every row by the dQ sweep)
    // phases & 16 (MU_ATTN_DQKV_ENCODED, MU_F32X only): dqkv is written in the chunk encoding the projection's data- and weight-gradient
    // kernels read (mu_split_encode form) instead of plain fp32 -- saves the separate encoding pass over the largest tensor of the block
    const int enc_out = (XF && (phases & 16)) ? 1 : 0;
    const int zero_masked = (((phases & 8) && nkmax == N) ? 1 : 0) | (enc_out << 1);
    if ((phases & 1) && !(zero_masked & 1) && hipMemsetAsync(dqkv, 0, (size_t)rows * 3 * C * sizeof(T), st) != hipSuccess) return MU_ERR_LAUNCH;
    // fp32x: the sweeps take dY chunk-encoded like qkv (the caller encodes qkv; dY is produced here, by phase 1): its encoded copy
    // lives in the workspace behind the row constants, and `dYs` is what the sweeps read
    const T* dYs = dY;
    if constexpr (XF) dYs = (const T*)((char*)ws + attn_ln_part_bytes(C) + attn_amax_bytes() + attn_rowc_bytes(B, N));
#define LAUNCH_BWD(DD, NKT)                                                                                                     \
    if (phases & 1) {                                                                                                           \
        attn_ln_bwd_kernel<TS, DD><<<nblk, 256, 0, st>>>((const TS*)gout, (const TS*)oattn, (const TS*)x, mean, rstd, gamma, (TS*)dY, delta, (double*)ws, rows, lse2, rowc, N, scale, cv, pshift, XF ? amaxp : nullptr); \
        attn_ln_bwd_final_kernel<<<mu_cdiv(DD, 4), 256, 0, st>>>((const double*)ws, nblk, DD, dgamma, dbeta, XF ? amaxp : nullptr, gsc); \
        if constexpr (XF) {                                                                                                     \
            const long ng = rows * DD / 8;                                                                                      \
            const long gr = (ng + 511) / 512;                                                                                   \
            attn_dy_encode_kernel<<<(int)(gr < 1 ? 1 : (gr > 8192 ? 8192 : gr)), 256, 0, st>>>((const f32x4*)dY, (uint4*)dYs, rowc, gsc, ng, DD / 8, N); \
        }                                                                                                                       \
    }                                                                                                                           \
    if (phases & 2) {                                                                                                           \
        if constexpr (DD == 128 && MU_DQ_NQT128 == 1)                                                                           \
            attn_bwd_dq2_kernel<T, DD, KTQ, 4, 1, MU_DQ_OCC128_Q1><<<dim3(mu_cdiv(N, 64), B), 256, 0, st>>>(qkv, dYs, kidx, kcnt, lse2, delta, dqkv, N, nkmax, scale, sl2, gsc, pshift, enc_out); \
        else                                                                                                                    \
            attn_bwd_dq2_kernel<T, DD, KTQ, 4><<<gq, 256, 0, st>>>(qkv, dYs, kidx, kcnt, lse2, delta, dqkv, N, nkmax, scale, sl2, gsc, pshift, enc_out); \
    }                                                                                                                           \
    if (phases & 4) {                                                                                                           \
        if constexpr (sizeof(T) == 2 && DD == 64 && MU_DKV_NW64 != 4)                                                           \
            attn_bwd_dkv3_kernel<T, DD, NKT, MU_DKV_NW64><<<dim3(mu_cdiv(nkmax, 16 * MU_DKV_NW64 * NKT), B), 64 * MU_DKV_NW64, 0, st>>>(qkv, dYs, kidx, kcnt, rowc, dqkv, N, nkmax, scale, sl2, zero_masked, gsc, pshift); \
        else if constexpr (sizeof(T) == 2 && DD == 128 && MU_DKV_NW128 != 4)                                                         \
            attn_bwd_dkv3_kernel<T, DD, NKT, MU_DKV_NW128><<<dim3(mu_cdiv(nkmax, 16 * MU_DKV_NW128 * NKT), B), 64 * MU_DKV_NW128, 0, st>>>(qkv, dYs, kidx, kcnt, rowc, dqkv, N, nkmax, scale, sl2, zero_masked, gsc, pshift); \
        else if constexpr (std::is_same<T, xf32>::value && DD == 128 && MU_XF_DKV_NW128 != 4)                                    \
            attn_bwd_dkv3_kernel<T, DD, NKT, MU_XF_DKV_NW128><<<dim3(mu_cdiv(nkmax, 16 * MU_XF_DKV_NW128 * NKT), B), 64 * MU_XF_DKV_NW128, 0, st>>>(qkv, dYs, kidx, kcnt, rowc, dqkv, N, nkmax, scale, sl2, zero_masked, gsc, pshift); \
        else if constexpr (sizeof(T) == 2 && DD == 256 && MU_DKV_NW256 == 8)                                                    \
            attn_bwd_dkv3_kernel<T, DD, NKT, 8><<<dim3(mu_cdiv(nkmax, 128 * NKT), B), 512, 0, st>>>(qkv, dYs, kidx, kcnt, rowc, dqkv, N, nkmax, scale, sl2, zero_masked, gsc, pshift); \
        else                                                                                                                    \
            attn_bwd_dkv3_kernel<T, DD, NKT><<<dim3(mu_cdiv(nkmax, 64 * NKT), B), 256, 0, st>>>(qkv, dYs, kidx, kcnt, rowc, dqkv, N, nkmax, scale, sl2, zero_masked, gsc, pshift); \
    }
    if (N % 4) return MU_ERR_SHAPE;
    switch (C) {
        case 32: { constexpr int KTQ = 64; LAUNCH_BWD(32, 2); } break;
        case 64: { constexpr int KTQ = MU_DQ_KT; LAUNCH_BWD(64, 2); } break;
        case 128: { constexpr int KTQ = 32; LAUNCH_BWD(128, MU_DKV_NKT128); } break;
        case 256: { constexpr int KTQ = 32; LAUNCH_BWD(256, 1); } break;
        default: return MU_ERR_SHAPE;
    }
#undef LAUNCH_BWD
    return MU_OK;
}

extern "C" int mu_attn_bwd_phases_padded(const void* qkv, const void* x, const void* oattn, const void* grad_out, const int* kidx, const int* kcnt,
                           const float* lse2, const float* ln_mean, const float* ln_rstd, const float* gamma, void* dY, float* delta,
                           void* dqkv, float* dgamma, float* dbeta, int B, int N, int C, int c_valid, int nkmax, void* workspace, long ws_bytes,
                           int dtype, int phases, void* stream) {
    if (!qkv || !x || !oattn || !grad_out || !kidx || !kcnt || !lse2 || !ln_mean || !ln_rstd || !gamma || !dY || !delta || !dqkv ||
        !dgamma || !dbeta || !workspace)
        return MU_ERR_ARG;
    if (B <= 0 || N <= 0 || nkmax <= 0 || c_valid <= 0 || c_valid > C) return MU_ERR_ARG;
    if (ws_bytes < mu_attn_bwd_workspace_bytes(B, N, C)) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == MU_F16)
        rc = attn_bwd_t<h16>((const h16*)qkv, (const h16*)x, (const h16*)oattn, (const h16*)grad_out, kidx, kcnt, lse2, ln_mean, ln_rstd, gamma, (h16*)dY, delta, (h16*)dqkv, dgamma, dbeta, B, N, C, c_valid, nkmax, workspace, st, phases);
    else if (dtype == MU_F32)
        rc = attn_bwd_t<float>((const float*)qkv, (const float*)x, (const float*)oattn, (const float*)grad_out, kidx, kcnt, lse2, ln_mean, ln_rstd, gamma, (float*)dY, delta, (float*)dqkv, dgamma, dbeta, B, N, C, c_valid, nkmax, workspace, st, phases);
    else if (dtype == MU_F32X)
        rc = attn_bwd_t<xf32>((const xf32*)qkv, (const xf32*)x, (const xf32*)oattn, (const xf32*)grad_out, kidx, kcnt, lse2, ln_mean, ln_rstd, gamma, (xf32*)dY, delta, (xf32*)dqkv, dgamma, dbeta, B, N, C, c_valid, nkmax, workspace, st, phases);
    else return MU_ERR_ARG;
    if (rc) return rc;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_attn_bwd_phases(const void* qkv, const void* x, const void* oattn, const void* grad_out, const int* kidx, const int* kcnt,
                           const float* lse2, const float* ln_mean, const float* ln_rstd, const float* gamma, void* dY, float* delta,
                           void* dqkv, float* dgamma, float* dbeta, int B, int N, int C, int nkmax, void* workspace, long ws_bytes,
                           int dtype, int phases, void* stream) {
    return mu_attn_bwd_phases_padded(qkv, x, oattn, grad_out, kidx, kcnt, lse2, ln_mean, ln_rstd, gamma, dY, delta, dqkv, dgamma, dbeta, B, N, C, C,
                                     nkmax, workspace, ws_bytes, dtype, phases, stream);
}

extern "C" int mu_attn_bwd(const void* qkv, const void* x, const void* oattn, const void* grad_out, const int* kidx, const int* kcnt,
                           const float* lse2, const float* ln_mean, const float* ln_rstd, const float* gamma, void* dY, float* delta,
                           void* dqkv, float* dgamma, float* dbeta, int B, int N, int C, int nkmax, void* workspace, long ws_bytes,
                           int dtype, void* stream) {
    return mu_attn_bwd_phases(qkv, x, oattn, grad_out, kidx, kcnt, lse2, ln_mean, ln_rstd, gamma, dY, delta, dqkv, dgamma, dbeta, B, N, C,
                              nkmax, workspace, ws_bytes, dtype, 7, stream);
}
