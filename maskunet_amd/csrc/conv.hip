// MFMA implicit-GEMM kernels for the 3x3 / 1x1 convolutions and the Linear layers (NHWC).
// Reference ops replaced: nn.Conv2d(k=3,pad=1,bias=False) in ConvBlock (ade_semantic.py:199,202),
// nn.Conv2d 1x1 heads (:284; city_instance.py:243-249), nn.Linear q/k/v (:157-159,170-172) and
// their autograd backward (data-grad = same kernel on flipped/transposed weights; weight-grad =
// pixel-reduction "TN" kernel below).
//
// MFMA shapes: fp16 storage -> v_mfma_f32_16x16x32_f16, fp32 storage -> v_mfma_f32_16x16x4_f32
// (exact fp32 FMA chain).  Both consume the same LDS image: rows of 64 bytes along K.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/maskunet_hip.h"

template <typename T> struct Mma;
template <> struct Mma<h16> {
    static constexpr int VN = 8;    // elements per 16-byte fragment
    using Frag = h16x8;
    static __device__ __forceinline__ Frag ld(const void* p) { return *reinterpret_cast<const Frag*>(p); }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static constexpr bool PAIR = false;      // (see Mma<xf32>)
    using Frag2 = Frag;
    static __device__ __forceinline__ Frag2 ld2(const void* p0, const void*) { return ld(p0); }
    static __device__ __forceinline__ void mma2(const Frag2& a, const Frag2& b, f32x4& c) { mma(a, b, c); }
};
// fp32x (common.h): the fp32 kernels on CHUNK-ENCODED operands -- a 16-byte fragment is [4 bf16 hi | 4 bf16 lo] of four fp32 values,
// i.e. the (hi, lo) operand pair itself -- and three v_mfma_f32_16x16x16_bf16 per fragment pair
template <> struct Mma<xf32> {
    static constexpr int VN = 4;
    using Frag = SplitF4;
    static __device__ __forceinline__ Frag ld(const void* p) { return mu_frag_enc(*reinterpret_cast<const uint4*>(p)); }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) { mu_mma_split(a, b, c); }
    // kernels whose K stage is 128 bytes (two chunks per lane group, kk = 0 / 1): the two chunks' hi halves side by side are the
    // eight-value operand of v_mfma_f32_16x16x32_bf16 -- half the MFMA instructions of the K = 16 form
    static constexpr bool PAIR = true;
    using Frag2 = SplitF8;
    static __device__ __forceinline__ Frag2 ld2(const void* p0, const void* p1) {
        const uint4 e0 = *reinterpret_cast<const uint4*>(p0), e1 = *reinterpret_cast<const uint4*>(p1);
        Frag2 r;
        r.hi = __builtin_bit_cast(bf16x8, make_uint4(e0.x, e0.y, e1.x, e1.y));
        r.lo = __builtin_bit_cast(bf16x8, make_uint4(e0.z, e0.w, e1.z, e1.w));
        return r;
    }
    static __device__ __forceinline__ void mma2(const Frag2& a, const Frag2& b, f32x4& c) { mu_mma_split(a, b, c); }
};
// fp32x 3x3 layers (round 6, common.h xh32): the same chunk layout with fp16 halves -- three v_mfma_f32_16x16x16/32_f16 per fragment pair
template <> struct Mma<xh32> {
    static constexpr int VN = 4;
    using Frag = SplitH4;
    static __device__ __forceinline__ Frag ld(const void* p) {
        const uint4 e = *reinterpret_cast<const uint4*>(p);
        Frag r;
        r.hi = __builtin_bit_cast(h16x4, make_uint2(e.x, e.y));
        r.lo = __builtin_bit_cast(h16x4, make_uint2(e.z, e.w));
        return r;
    }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, b.hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.hi, c, 0, 0, 0);
    }
    static constexpr bool PAIR = true;
    using Frag2 = SplitH8;
    static __device__ __forceinline__ Frag2 ld2(const void* p0, const void* p1) {
        const uint4 e0 = *reinterpret_cast<const uint4*>(p0), e1 = *reinterpret_cast<const uint4*>(p1);
        Frag2 r;
        r.hi = __builtin_bit_cast(h16x8, make_uint4(e0.x, e0.y, e1.x, e1.y));
        r.lo = __builtin_bit_cast(h16x8, make_uint4(e0.z, e0.w, e1.z, e1.w));
        return r;
    }
    static __device__ __forceinline__ void mma2(const Frag2& a, const Frag2& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
    }
};
// the xh32 weights carry a static shift of 2^MU_XH_WSHIFT (common.h): every kernel un-shifts its accumulators in front of its epilogue
template <typename T, int TM, int TN>
__device__ __forceinline__ void mma_unshift(f32x4 (&acc)[TM][TN]) {
    if constexpr (std::is_same<T, xh32>::value) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= 1.0f / (float)(1 << MU_XH_WSHIFT);
    }
}
template <> struct Mma<float> {
    static constexpr int VN = 4;
    using Frag = f32x4;
    static __device__ __forceinline__ Frag ld(const void* p) { return *reinterpret_cast<const Frag*>(p); }
    static constexpr bool PAIR = false;
    using Frag2 = Frag;
    static __device__ __forceinline__ Frag2 ld2(const void* p0, const void*) { return ld(p0); }
    static __device__ __forceinline__ void mma2(const Frag2& a, const Frag2& b, f32x4& c) { mma(a, b, c); }
    // lane group g holds k = 4g..4g+3; step s multiplies k = 4g+s of A with the same k of B
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    }
};

// LDS image of a [rows][64 B] tile: 16-byte chunk c of row r lives at chunk c ^ f[(r>>2)&3],
// f = {0,2,3,1}; makes the ds_read_b128 fragment reads (row = lane&15, chunk = lane>>4)
// conflict-free within each of the instruction's four 16-lane service groups.
__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3); }


// ------------------------------------------------------------------------------------------
// y[p][co] = bias[co] + sum_{tap,ci} x[p + shift(tap)][ci] * w[tap][co][ci]
// block tile: (WR*TM*16) output channels x (WC*TN*16) pixels, 4 waves, D[co][pixel].
// ------------------------------------------------------------------------------------------
// FEPI (all forward kernels below): the inference epilogue y = act(conv * scale[co] + bias[co] + res) -- eval-mode BatchNorm folded
// into per-channel (scale, shift), the residual add and GELU / ReLU of ConvBlock (ade_semantic.py:199-208, validation loop :443-471)
// inside the conv epilogue instead of a separate read + write pass per BatchNorm.  Compile-time: the training instantiations
// (FEPI = false) are unchanged.  res has y's row stride.
template <typename T>
__device__ __forceinline__ float epi_act(float v, int act) { return mu_act_t<sizeof(T) == 2>(v, act); }

// HL (round 6, T = h16 only: the two-term data gradient of the fp32x 3x3 layers, mu_conv_dgrad_h): x is the ONE-term fp16 dy, w the "HL"
// weight rows [Cin lo | Cin hi] of 2 * Cin halves (elementwise.hip mu_prep_weight), the K loop walks every input chunk twice -- against
// the lo halves, then against the hi halves -- and the result leaves as fp32 rows yf, multiplied by oscale[1] / 2^MU_XH_WSHIFT.
// HL = 2: both halves (two MFMAs per product); HL = 1: the hi halves only -- the weights of the data gradient as ONE fp16 term (one MFMA
// per product: sized on the oracle like everything else, NOTES_r06 §2b); HL = 0: the ordinary kernel.
template <typename T, int TM, int TN, int WR, int TAPS, bool FEPI = false, int HL = 0>
__global__ __launch_bounds__(256) void conv_nt_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                      T* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                                      const float* __restrict__ scale = nullptr, const T* __restrict__ res = nullptr, int act = 0,
                                                      float* __restrict__ yf = nullptr, const float* __restrict__ oscale = nullptr) {
    using M_ = Mma<T>;
    using Frag = typename M_::Frag;
    constexpr int VN = M_::VN, KC = 4 * VN;
    constexpr int WC = 4 / WR;
    constexpr int BCO = WR * TM * 16, BPX = WC * TN * 16;
    constexpr int NA = (BCO * 4 + 255) / 256, NB = (BPX * 4) / 256;
    static_assert(BPX * 4 % 256 == 0, "pixel tile must be a multiple of 64");

    constexpr int OPITCH = BCO * 2 + 16;                    // fp16 epilogue: output tile staged as [pixel][channel] rows
    constexpr int LDS0 = 2 * (BCO + BPX) * 64;
    constexpr int LDSB = (sizeof(T) == 2 && BPX * OPITCH > LDS0) ? BPX * OPITCH : LDS0;
    __shared__ __attribute__((aligned(16))) char lds[LDSB];
    char* As = lds;
    char* Bs = lds + 2 * BCO * 64;

    const long Mtot = (long)B * H * W;
    const int npb = (int)((Mtot + BPX - 1) / BPX), ncb = (Cout + BCO - 1) / BCO;
    const int L = xcd_remap(blockIdx.x, npb * ncb);
    const int cb = L % ncb, pb = L / ncb;
    const int co0 = cb * BCO;
    const long px0 = (long)pb * BPX;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int r16 = lane & 15, g = lane >> 4;
    const int ch = tid & 3;                      // this thread's 16-byte chunk along K
    const int ldrow = tid >> 2;                  // 0..63

    // pixel bookkeeping for the rows this thread stages
    long prow[NB]; int ph[NB], pw[NB]; bool pok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        long p = px0 + ldrow + i * 64;
        pok[i] = p < Mtot;
        long pp = pok[i] ? p : 0;
        pw[i] = (int)(pp % W);
        ph[i] = (int)((pp / W) % H);
        prow[i] = pp;
    }

    const int kreal = Cin / KC;
    const int kchunks = HL == 2 ? 2 * kreal : kreal;
    const int Cw = HL ? 2 * Cin : Cin;                       // weight row length
    const int nsteps = TAPS * kchunks;
    uint4 ra[NA], rb[NB];

    auto gload = [&](int s) {
        const int tap = s / kchunks, ciw = (s % kchunks) * KC + (HL == 1 ? Cin : 0);
        const int ci0 = HL ? (ciw >= Cin ? ciw - Cin : ciw) : ciw;
        const int dh = TAPS == 9 ? tap / 3 - 1 : 0, dw = TAPS == 9 ? tap % 3 - 1 : 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = ldrow + i * 64, co = co0 + row;
            if (row < BCO && co < Cout)
                ra[i] = *reinterpret_cast<const uint4*>(w + ((long)tap * Cout + co) * Cw + ciw + ch * VN);
            else
                ra[i] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int hh = ph[i] + dh, ww = pw[i] + dw;
            if (pok[i] && hh >= 0 && hh < H && ww >= 0 && ww < W)
                rb[i] = *reinterpret_cast<const uint4*>(x + (prow[i] + dh * W + dw) * x_ld + ci0 + ch * VN);
            else
                rb[i] = make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = ldrow + i * 64;
            if (row < BCO) *reinterpret_cast<uint4*>(As + buf * BCO * 64 + row * 64 + swz64(row, ch) * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = ldrow + i * 64;
            *reinterpret_cast<uint4*>(Bs + buf * BPX * 64 + row * 64 + swz64(row, ch) * 16) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        Frag a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (wr * TM + i) * 16 + r16;
            a[i] = M_::ld(As + buf * BCO * 64 + row * 64 + swz64(row, g) * 16);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = (wc * TN + j) * 16 + r16;
            b[j] = M_::ld(Bs + buf * BPX * 64 + row * 64 + swz64(row, g) * 16);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) M_::mma(a[i], b[j], acc[i][j]);
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    mma_unshift<T>(acc);

    if constexpr (HL) {
        const float os = oscale[1] * (1.0f / (float)(1 << MU_XH_WSHIFT));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long p = px0 + (wc * TN + j) * 16 + r16;
            if (p >= Mtot) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int co = co0 + (wr * TM + i) * 16 + 4 * g;
                if (co >= Cout) continue;
                *reinterpret_cast<f32x4*>(yf + p * y_ld + co) = acc[i][j] * os;
            }
        }
        return;
    }
    if constexpr (sizeof(T) == 2) {
        // fp16: the block's tile goes through LDS (all fragment reads are behind the loop's last barrier) and leaves as 16-byte
        // chunks of contiguous output rows (as conv_nt2_kernel) instead of 8-byte pieces of sixteen rows per store
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int px = (wc * TN + j) * 16 + r16;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int co = (wr * TM + i) * 16 + 4 * g;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool in = co0 + co + r < Cout;
                    v[r] = acc[i][j][r];
                    if constexpr (FEPI) { if (scale && in) v[r] *= scale[co0 + co + r]; }
                    v[r] += (bias && in ? bias[co0 + co + r] : 0.f);
                }
                h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                *reinterpret_cast<h16x4*>(lds + px * OPITCH + co * 2) = o;
            }
        }
        __syncthreads();
        constexpr int CH = BCO / 8;
        for (int idx = tid; idx < BPX * CH; idx += 256) {
            const int px = idx / CH, ch8 = idx - px * CH;
            const long p = px0 + px;
            if (p < Mtot && co0 + ch8 * 8 < Cout) {
                uint4 o4 = *reinterpret_cast<const uint4*>(lds + px * OPITCH + ch8 * 16);
                if constexpr (FEPI) {
                    h16x8 ov = *reinterpret_cast<const h16x8*>(&o4), rv = (h16x8)(h16)0;
                    if (res) rv = *reinterpret_cast<const h16x8*>(res + p * y_ld + co0 + ch8 * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (h16)epi_act<T>((float)ov[e] + (float)rv[e], act);
                    o4 = *reinterpret_cast<const uint4*>(&ov);
                }
                *reinterpret_cast<uint4*>(y + p * y_ld + co0 + ch8 * 8) = o4;
            }
        }
        return;
    }
    // epilogue: lane holds 4 consecutive output channels of one pixel per (i,j) tile
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long p = px0 + (wc * TN + j) * 16 + r16;
        if (p >= Mtot) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int co = co0 + (wr * TM + i) * 16 + 4 * g;
            if (co >= Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
            if constexpr (FEPI) {
                if (scale) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= scale[co + r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (bias ? bias[co + r] : 0.f);
            if constexpr (FEPI) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = epi_act<T>(v[r] + (res ? (float)res[p * y_ld + co + r] : 0.f), act);
            }
            if constexpr (sizeof(T) == 2) {
                h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                *reinterpret_cast<h16x4*>(y + p * y_ld + co) = o;
            } else {
                *reinterpret_cast<float4*>(y + p * y_ld + co) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// v2 of the same implicit GEMM for K rows of >= 128 bytes (Cin % 64 == 0 fp16, % 32 fp32):
//  * BK = 128 bytes per stage (half the barriers of the 64-byte version);
//  * both tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR staging and no
//    ds_write pass, which was the LDS-pipe bottleneck of v1.  The LDS image is lane-linear
//    (8 rows x 128 B per wave instruction); the XOR swizzle is applied to the per-lane SOURCE
//    chunk and again on the fragment read (cdna guide rule 21).  Padding pixels of the 3x3
//    halo and out-of-range rows read a 16-byte zero page instead of branching around the load.
//  * double-buffered: stage s+1 is in flight while stage s is multiplied.
// ------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) char mu_zero_page[16];

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same LDS-DMA issued through inline asm, for the ring-buffered kernels that order their DMAs with explicit counted
// s_waitcnt vmcnt(N) + s_barrier.  hipcc's waitcnt pass treats ds_read_b64_tr_b16 (no memory operand) as aliasing every
// pending LDS-DMA it knows of and puts s_waitcnt vmcnt(0) in front of the first transposed read of each stage, which
// drains the whole ring (measured: ring depth 2..8 made no difference until the DMA was hidden from the pass).  Hidden
// DMAs are NOT covered by __syncthreads(): every consumer must sit behind an explicit vmcnt wait and a barrier.
// M0 carries the wave-uniform LDS destination; nothing else in these kernels uses M0.
__device__ __forceinline__ void glds16a(const void* gsrc, void* lds_wave_base) {
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(gsrc) : "memory");
}

#ifndef MU_NT2_SB_OCC
#define MU_NT2_SB_OCC 3
#endif
#ifndef MU_NT2_OCC
#define MU_NT2_OCC 2
#endif
#ifndef MU_NT2_EPI
#define MU_NT2_EPI 1        // fp16: stage the output tile in LDS and store whole rows
#endif
// SB: the whole reduction is ONE stage (1x1, Cin == one 128-byte row): a single LDS buffer, so more blocks fit a CU -- the block is
// load -> multiply -> store with nothing to pipeline inside it, only other resident blocks hide its latencies.
#ifndef MU_NT2_SB
#define MU_NT2_SB 1
#endif
#define MU_NT2_ENC_H (-1)       // `act` of the fp32x training instantiation: write y in the attention operand encoding (mu_conv1x1_fwd_enc_h)
template <typename T, int TM, int TN, int WR, int TAPS, bool SB = false, bool FEPI = false>
__global__ __launch_bounds__(256, SB ? MU_NT2_SB_OCC : MU_NT2_OCC) void conv_nt2_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                       T* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                                       const T* __restrict__ addend = nullptr, const float* __restrict__ scale = nullptr,
                                                       int act = 0) {
    // FEPI: y = act(conv * scale[co] + bias[co] + addend); the training instantiations (FEPI = false) ignore the three arguments
    using M_ = Mma<T>;
    using Frag = typename M_::Frag;
    constexpr int VN = M_::VN, KC = 8 * VN;                 // elements per 128-byte stage row
    constexpr int WC = 4 / WR;
    constexpr int BCO = WR * TM * 16, BPX = WC * TN * 16;
    constexpr int PA = BCO / 32, PB = BPX / 32;            // staging passes (4 waves x 8 rows each)
    static_assert(BCO % 32 == 0 && BPX % 32 == 0, "tiles must be multiples of 32 rows");
    constexpr int STAGE = (BCO + BPX) * 128;

    constexpr int OPITCH = BCO * 2 + 16;                    // fp16 epilogue: the block's output tile staged as [pixel][channel] rows
    constexpr bool EPI = sizeof(T) == 2 && (SB || MU_NT2_EPI);
    constexpr int LDS0 = (SB ? 1 : 2) * STAGE;
    constexpr int LDSB = EPI && BPX * OPITCH > LDS0 ? BPX * OPITCH : LDS0;

    __shared__ __attribute__((aligned(16))) char lds[LDSB];

    const long Mtot = (long)B * H * W;
    const int npb = (int)((Mtot + BPX - 1) / BPX), ncb = (Cout + BCO - 1) / BCO;
    const int L = xcd_remap(blockIdx.x, npb * ncb);
    const int cb = L % ncb, pb = L / ncb;
    const int co0 = cb * BCO;
    const long px0 = (long)pb * BPX;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int r16 = lane & 15, g = lane >> 4;
    const int srow = lane >> 3, sch = lane & 7;             // staging: row within the wave's 8-row group, dest chunk

    long prow[PB]; int ph[PB], pw[PB]; bool pok[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const long p = px0 + (i * 4 + wave) * 8 + srow;
        pok[i] = p < Mtot;
        const long pp = pok[i] ? p : 0;
        pw[i] = (int)(pp % W);
        ph[i] = (int)((pp / W) % H);
        prow[i] = pp;
    }
    const int kchunks = SB ? 1 : Cin / KC;
    const int nsteps = SB ? 1 : TAPS * kchunks;

    auto stage = [&](int s, int buf) {
        const int tap = s / kchunks, ci0 = (s % kchunks) * KC;
        const int dh = TAPS == 9 ? tap / 3 - 1 : 0, dw = TAPS == 9 ? tap % 3 - 1 : 0;
        char* Ab = lds + buf * STAGE;
        char* Bb = Ab + BCO * 128;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (i * 4 + wave) * 8 + srow, co = co0 + row;
            const int sc = sch ^ (row & 7);
            const void* src = co < Cout ? (const void*)(w + ((long)tap * Cout + co) * Cin + ci0 + sc * VN) : (const void*)mu_zero_page;
            glds16(src, Ab + (i * 4 + wave) * 1024);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = (i * 4 + wave) * 8 + srow;
            const int sc = sch ^ (row & 7);
            const int hh = ph[i] + dh, ww = pw[i] + dw;
            const bool ok = pok[i] && hh >= 0 && hh < H && ww >= 0 && ww < W;
            const void* src = ok ? (const void*)(x + (prow[i] + dh * W + dw) * x_ld + ci0 + sc * VN) : (const void*)mu_zero_page;
            glds16(src, Bb + (i * 4 + wave) * 1024);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) stage(s + 1, buf ^ 1);
        const char* Ab = lds + buf * STAGE;
        const char* Bb = Ab + BCO * 128;
        if constexpr (M_::PAIR) {
            typename M_::Frag2 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wr * TM + i) * 16 + r16;
                a[i] = M_::ld2(Ab + row * 128 + ((g ^ (row & 7)) << 4), Ab + row * 128 + (((4 + g) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = (wc * TN + j) * 16 + r16;
                b[j] = M_::ld2(Bb + row * 128 + ((g ^ (row & 7)) << 4), Bb + row * 128 + (((4 + g) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) M_::mma2(a[i], b[j], acc[i][j]);
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            Frag a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wr * TM + i) * 16 + r16;
                a[i] = M_::ld(Ab + row * 128 + (((kk * 4 + g) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = (wc * TN + j) * 16 + r16;
                b[j] = M_::ld(Bb + row * 128 + (((kk * 4 + g) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) M_::mma(a[i], b[j], acc[i][j]);
        }
        }
        __syncthreads();          // drains the LDS-DMA of stage s+1 (vmcnt(0)) and fences the reads of stage s
    }
    mma_unshift<T>(acc);

    if constexpr (EPI) {
        // the block's BPX x BCO tile goes through LDS (the loop's last barrier fenced the fragment reads) and leaves as whole
        // 16-byte chunks of contiguous output rows: the streams these layers are need full-line writes
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int px = (wc * TN + j) * 16 + r16;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int co = (wr * TM + i) * 16 + 4 * g;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool in = co0 + co + r < Cout;
                    v[r] = acc[i][j][r];
                    if constexpr (FEPI) { if (scale && in) v[r] *= scale[co0 + co + r]; }
                    v[r] += (bias && in ? bias[co0 + co + r] : 0.f);
                }
                h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                *reinterpret_cast<h16x4*>(lds + px * OPITCH + co * 2) = o;
            }
        }
        __syncthreads();
        constexpr int CH = BCO / 8;
        for (int idx = tid; idx < BPX * CH; idx += 256) {
            const int px = idx / CH, ch = idx - px * CH;
            const long p = px0 + px;
            if (p < Mtot && co0 + ch * 8 < Cout) {
                uint4 o = *reinterpret_cast<const uint4*>(lds + px * OPITCH + ch * 16);
                if constexpr (FEPI) {           // y = act(conv(x) + addend) (addend has y's row layout): the residual-branch gradient of
                    uint4 a4 = make_uint4(0, 0, 0, 0);      // the attention block (act none), or an inference epilogue
                    if (addend) a4 = *reinterpret_cast<const uint4*>(addend + p * y_ld + co0 + ch * 8);
                    const h16x8 ov = *reinterpret_cast<const h16x8*>(&o), av = *reinterpret_cast<const h16x8*>(&a4);
                    h16x8 r;
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = (h16)epi_act<T>((float)ov[e] + (float)av[e], act);
                    o = *reinterpret_cast<const uint4*>(&r);
                }
                *reinterpret_cast<uint4*>(y + p * y_ld + co0 + ch * 8) = o;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long p = px0 + (wc * TN + j) * 16 + r16;
        if (p >= Mtot) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int co = co0 + (wr * TM + i) * 16 + 4 * g;
            if (co >= Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r];
                if constexpr (FEPI) { if (scale) v[r] *= scale[co + r]; }
                v[r] += (bias ? bias[co + r] : 0.f);
                if constexpr (FEPI) v[r] = epi_act<T>(v[r] + (addend ? (float)addend[p * y_ld + co + r] : 0.f), act);
            }
            if constexpr (sizeof(T) == 2) {
                h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                *reinterpret_cast<h16x4*>(y + p * y_ld + co) = o;
            } else {
                if constexpr (std::is_same<T, xf32>::value && !FEPI) {
                    if (act == MU_NT2_ENC_H) {
                        // the attention operand encoding written by the producer (mu_conv1x1_fwd_enc_h; common.h): a group of eight channels is
                        // [8 fp16 hi | 8 fp16 lo]; this lane holds four of them (the lower four for even g): its hi parts go to bytes
                        // 8 (g & 1) of the group, its lo parts 16 bytes further -- two 8-byte stores instead of one 16-byte store
                        uint32_t h0, h1, l0, l1;
                        mu_hsplit2(v[0], v[1], h0, l0);
                        mu_hsplit2(v[2], v[3], h1, l1);
                        char* gp = reinterpret_cast<char*>(y + p * y_ld + (co & ~7)) + (g & 1) * 8;
                        *reinterpret_cast<uint2*>(gp) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(gp + 16) = make_uint2(l0, l1);
                        continue;
                    }
                }
                *reinterpret_cast<float4*>(y + p * y_ld + co) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// v3 for 3x3: LDS-staged halo tile.  A block owns a TH x 16 spatial tile of one image; for every
// 64-channel chunk the (TH+2) x 18 input halo is brought into LDS ONCE (LDS-DMA, zero page for the
// padding ring) and re-used by all nine taps -- the tap shift is just a different row of the halo image
// in the fragment read.  Only the 16 KB weight tile changes per tap.  Versus v2 this cuts the
// activation L2->LDS traffic 9x (v2 is pinned at the ~10 TB/s L2->LDS ceiling on the Cin<=256 layers).
// The next chunk's halo is streamed in 1/6 pieces behind the first six tap steps.
// ------------------------------------------------------------------------------------------
#ifndef MU_NT3_RING
#define MU_NT3_RING 1
#endif
#ifndef MU_NT3_ABL_WONCE
#define MU_NT3_ABL_WONCE 0
#endif
#ifndef MU_NT3_STATS
#define MU_NT3_STATS 1          // BatchNorm-statistics epilogue of the halo-tile kernel (fp16 and fp32x training launches)
#endif
#ifndef MU_XF_NT3_RING8
#define MU_XF_NT3_RING8 0       // measured: 14.45 vs 14.5 ms/step over the 51 launches (session r04s) -- neutral, the 4-wave two-block form stays
#endif
// Per-channel (sum, sum of squares) of a wave's staged 64-pixel x 64-channel output tile, taken from the SAME fp16-rounded
// values that were just stored (what BatchNorm will read): lane = (pixel sub-index lane>>3, 8-channel group q), 8 pixels per lane,
// then the eight lanes of a channel group are folded with xor-shuffles and lanes 0-7 write 8 channels x {sum, sumsq} each.
// part row layout [Cout][2] floats; rows = one per (tile, 4-image-row group).  Saves BatchNorm's separate statistics sweep.
__device__ __forceinline__ void tile_stats_accum(const h16x8& o, float (&ssum)[8], float (&ssq)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float v = (float)o[c];
        ssum[c] += v;
        ssq[c] = fmaf(v, v, ssq[c]);
    }
}
__device__ __forceinline__ void tile_stats_store(float (&ssum)[8], float (&ssq)[8], float* __restrict__ row, int co, int lane) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            ssum[c] += __shfl_xor(ssum[c], m);
            ssq[c] += __shfl_xor(ssq[c], m);
        }
    }
    if (lane < 8) {
#pragma unroll
        for (int c = 0; c < 8; c += 2)
            *reinterpret_cast<float4*>(row + (co + c) * 2) = make_float4(ssum[c], ssq[c], ssum[c + 1], ssq[c + 1]);
    }
}

// HL: see conv_nt_kernel -- here the lo / hi passes of one input chunk follow each other (chunk instance c = 2 * chunk + {lo, hi}), so the
// halo of a chunk is staged ONCE for both (the odd instance stages the next chunk's halo, the even one stages nothing).
template <typename T, int TM, int TN, int WR, int NWV, bool RINGP, bool FEPI, int HL = 0>
__device__ __forceinline__ void conv_nt3_body(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                              T* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                              const float* __restrict__ scale, const T* __restrict__ res, int act,
                                              float* __restrict__ stat_part = nullptr, float* __restrict__ yf = nullptr,
                                              const float* __restrict__ oscale = nullptr) {
    // stat_part (training kernels, may be NULL): per-channel (sum, sum of squares) of the values this kernel stores, one row per
    // (tile, wave row wc) -- [ntile * WC][Cout][2] floats, the layout mu_bn_train_stats_rows folds (round 5: the statistics epilogue the
    // ping-pong kernels have had since round 1, for the layers the halo-tile kernel serves: Cout = 64 and small grids)
    using M_ = Mma<T>;
    using Frag = typename M_::Frag;
    constexpr int VN = M_::VN, KC = 8 * VN;
    constexpr int WC = NWV / WR;
    constexpr int BCO = WR * TM * 16;
    constexpr int TH = WC * TN, TW = 16, HW_ = TW + 2;      // spatial tile TH x 16, halo row width 18
    constexpr int HROWS = (TH + 2) * HW_;
    constexpr int HINST = (HROWS + 7) / 8;                  // 8 halo rows (128 B each) per wave DMA instruction
    static_assert((HINST + NWV - 1) / NWV <= 9, "halo does not fit the 9 tap steps");
    constexpr int PA = BCO / (8 * NWV);
    static_assert(BCO % (8 * NWV) == 0, "weight tile rows must split over the waves");
    constexpr int HBYTES = HINST * 1024, WBYTES = BCO * 128;
    // RING (fp16, 64-channel blocks = the Cout = 64 layers): THREE weight slots, W(s+2) issued at tap s, counted s_waitcnt vmcnt +
    // raw s_barrier per tap instead of __syncthreads() (= vmcnt(0): every tap waited for a DMA issued ~0.3 us earlier; PMC: waves
    // parked 48 %, MFMA pipe 26 % busy).  Exactly PA + 1 DMAs per wave per tap (dummies to a dump page keep the counts uniform).
    // In-process A/B: 128 -> 64 @128^2 0.226 -> 0.203 ms; with a single 64-channel chunk (9 taps per tile) the longer prologue
    // costs more than the waits it removes (64 -> 64 @128^2 0.110 -> 0.116 ms), so the launcher picks RINGP for Cin >= 128 only.
    // (fp32x, MU_XF_NT3_RING8: the same ring on 8-wave blocks with 128 output channels x 16 x 16 pixels -- one block per CU, two waves per SIMD)
    constexpr bool RING = RINGP && MU_NT3_RING && ((sizeof(T) == 2 && BCO == 64 && NWV == 4) || (mu_is_split<T>::value && BCO == 128 && NWV == 8));
    constexpr int NWS = RING ? 3 : 2;

    __shared__ __attribute__((aligned(16))) char lds[2 * HBYTES + NWS * WBYTES + (RING ? 1024 : 0)];
    char* Hs = lds;
    char* Ws = lds + 2 * HBYTES;
    char* dump = lds + 2 * HBYTES + NWS * WBYTES;

    const int tiles_w = W / TW, tiles_h = H / TH;
    const int ntile = B * tiles_h * tiles_w, ncb = Cout / BCO;
    const int L = xcd_remap(blockIdx.x, ntile * ncb);
    const int cb = L % ncb, tl = L / ncb;
    const int co0 = cb * BCO;
    const int tw_ = tl % tiles_w, th_ = (tl / tiles_w) % tiles_h, bimg = tl / (tiles_w * tiles_h);
    const int h0 = th_ * TH, w0 = tw_ * TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int r16 = lane & 15, g = lane >> 4;
    const int srow = lane >> 3, sch = lane & 7;

    constexpr bool HL2 = HL == 2;
    const int kchunks = HL2 ? 2 * (Cin / KC) : Cin / KC;    // chunk INSTANCES (HL = 2: lo and hi pass of every chunk)
    const int Cw = HL ? 2 * Cin : Cin;                       // weight row length
    const int nsteps = 9 * kchunks;

    // DMA source offsets are loop-invariant per lane too: precompute them once (element offsets, int), so a stage costs one
    // scalar base update + one add per instruction instead of 64-bit multiply-adds.
    int wl[PA];                                             // weights: (co0 + row) * Cin + swizzled chunk
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (i * NWV + wave) * 8 + srow;
        wl[i] = (co0 + row) * Cw + (sch ^ (row & 7)) * VN;
    }
    constexpr int HPW = (HINST + NWV - 1) / NWV;            // halo instructions owned by this wave: inst = k * NWV + wave
    int hl[HPW];                                            // (hh * W + ww) * x_ld + swizzled chunk, or -1 for the zero ring
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int hr = (k * NWV + wave) * 8 + srow;
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int hh = h0 - 1 + hy, ww = w0 - 1 + hx;
        const bool ok = (k * NWV + wave) < HINST && hr < HROWS && hh >= 0 && hh < H && ww >= 0 && ww < W;
        hl[k] = ok ? (int)(((long)hh * W + ww) * x_ld) + (sch ^ (hx & 7)) * VN : -1;
    }
    const T* xb = x + (long)bimg * H * W * x_ld;

    auto stage_w = [&](int s, int buf) {
        const int tap = s % 9, c_ = s / 9;
        const int ci0 = HL2 ? ((c_ & 1) ? Cin : 0) + (c_ >> 1) * KC : (HL == 1 ? Cin : 0) + c_ * KC;
        const T* wb = w + (long)tap * Cout * Cw + ci0;       // wave-uniform
        char* Wb = Ws + buf * WBYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(wb + wl[i], Wb + (i * NWV + wave) * 1024);
    };
    auto stage_h = [&](int k, int ci0, int buf) {           // this wave's k-th halo instruction (rows 8*(k*NWV+wave) ..)
        const void* src = hl[k] >= 0 ? (const void*)(xb + hl[k] + ci0) : (const void*)mu_zero_page;
        glds16(src, Hs + buf * HBYTES + (k * NWV + wave) * 1024);
    };
    // halo of chunk instance c: data chunk and buffer (HL: two instances share one halo); next_h: does instance c + 1 need a new halo?
    auto hchunk = [&](int c) { return HL2 ? (c >> 1) : c; };
    auto next_h = [&](int c) { return c + 1 < kchunks && (!HL2 || ((c + 1) & 1) == 0); };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Loop-invariant per-lane LDS byte offsets: the inner loop then needs one scalar+vector add per fragment group and
    // immediates for everything else (the generic index arithmetic cost more VALU time than the MFMAs).
    //   weights: row = (wr*TM+i)*16 + r16 -> i*2048 is an immediate; chunk swizzle key r16 & 7
    //   halo   : row = (wc*TN + j + dh)*18 + r16 + dw -> (j + dh)*2304 immediate/scalar; key (r16 + dw) & 7 depends on dw only
    int aoff[2], boff[3][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        aoff[kk] = (wr * TM * 16 + r16) * 128 + (((kk * 4 + g) ^ (r16 & 7)) << 4);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) boff[dw][kk] = (wc * TN * HW_ + r16 + dw) * 128 + (((kk * 4 + g) ^ ((r16 + dw) & 7)) << 4);
    }

    stage_w(0, 0);
    if constexpr (RING) { if (nsteps > 1) stage_w(1, 1); }
#pragma unroll
    for (int k = 0; k < HPW; ++k)
        if (k * NWV + wave < HINST) stage_h(k, 0, 0);
    __syncthreads();

    int s = 0, wslot = 0;                                    // RING: slot of W(s) = s % 3, carried
    for (int c = 0; c < kchunks; ++c) {
        const int hbuf = (hchunk(c) & 1) * HBYTES;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
#pragma unroll
            for (int dw = 0; dw < 3; ++dw, ++s) {
                const int t = dh * 3 + dw;
                if constexpr (RING) {
                    const int nslot = wslot == 0 ? 2 : wslot - 1;              // (s + 2) % 3
                    if (s + 2 < nsteps) stage_w(s + 2, nslot);
                    else {
#pragma unroll
                        for (int i = 0; i < PA; ++i) glds16(mu_zero_page, dump);
                    }
                    if (t < HPW && next_h(c) && t * NWV + wave < HINST) stage_h(t, hchunk(c + 1) * KC, hchunk(c + 1) & 1);
                    else glds16(mu_zero_page, dump);
                } else {
#if MU_NT3_ABL_WONCE            // timing-only ablation (wrong results): weights staged once, no per-tap wait -- upper bound of a weights-resident kernel
                if (false) stage_w(s + 1, (s + 1) & 1);
#else
                if (s + 1 < nsteps) stage_w(s + 1, (s + 1) & 1);
#endif
                if (t < HPW && next_h(c) && t * NWV + wave < HINST)        // one halo piece of the next chunk per tap step
                    stage_h(t, hchunk(c + 1) * KC, hchunk(c + 1) & 1);
                }
                const char* Wb = Ws + (RING ? wslot : (MU_NT3_ABL_WONCE ? 0 : (s & 1))) * WBYTES;
                const char* Hb = Hs + hbuf + dh * (HW_ * 128);
                if constexpr (M_::PAIR) {
                    typename M_::Frag2 a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = M_::ld2(Wb + aoff[0] + i * 2048, Wb + aoff[1] + i * 2048);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = M_::ld2(Hb + boff[dw][0] + j * (HW_ * 128), Hb + boff[dw][1] + j * (HW_ * 128));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) M_::mma2(a[i], b[j], acc[i][j]);
                } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    Frag a[TM], b[TN];
                    const char* wa = Wb + aoff[kk];
                    const char* hb = Hb + boff[dw][kk];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = M_::ld(wa + i * 2048);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = M_::ld(hb + j * (HW_ * 128));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) M_::mma(a[i], b[j], acc[i][j]);
                }
                }
                if constexpr (RING) {
                    // all but this tap's PA + 1 DMAs have landed: W(s+1) (issued at tap s-1) and every older halo piece.  The MFMAs
                    // above consumed this tap's fragment reads, so behind the barrier slot s % 3 is free for W(s+3)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + 1) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    wslot = wslot == 2 ? 0 : wslot + 1;
                } else {
#if !MU_NT3_ABL_WONCE
                    __syncthreads();
#endif
                }
            }
        }
    }
    if constexpr (RING) {                                    // trailing dummies: drained before the epilogue re-uses the LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

#ifndef MU_NT3_EPI
#define MU_NT3_EPI 1
#endif
    mma_unshift<T>(acc);
    if constexpr (HL) {
        // two-term data gradient: fp32 rows, un-scaled (the dy scale and the weight shift are powers of two); a lane's four channels of
        // one pixel are 16 contiguous bytes, the 16 lanes of a quad group cover 16 consecutive pixels of an image row
        const float os = oscale[1] * (1.0f / (float)(1 << MU_XH_WSHIFT));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long p = ((long)bimg * H + h0 + wc * TN + j) * W + w0 + r16;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int co = co0 + (wr * TM + i) * 16 + 4 * g;
                *reinterpret_cast<f32x4*>(yf + p * y_ld + co) = acc[i][j] * os;
            }
        }
        return;
    }
    if constexpr (sizeof(T) == 2 && TM == 4 && MU_NT3_EPI) {
        // wave-private staged epilogue (as conv_nt4_kernel): the wave's (TN x 16) px x 64 co tile goes through its own LDS slice
        // (every fragment read and DMA of the loop is behind its last barrier) and leaves as whole 128-byte rows
        char* Os = lds + wave * (TN * 16 * 128);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int p = j * 16 + r16;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int co = i * 16 + 4 * g;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][j][r];
                    if constexpr (FEPI) { if (scale) v[r] *= scale[co0 + wr * 64 + co + r]; }
                    v[r] += (bias ? bias[co0 + wr * 64 + co + r] : 0.f);
                }
                h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                *reinterpret_cast<h16x4*>(Os + p * 128 + (((co >> 2) ^ (((p >> 1) & 7) << 1)) << 3)) = o;
            }
        }
        const int q = lane & 7;
        float ssum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ssq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < TN * 2; ++it) {
            const int p = it * 8 + (lane >> 3);
            h16x8 o = *reinterpret_cast<const h16x8*>(Os + p * 128 + ((q ^ ((p >> 1) & 7)) << 4));
            const long gp = ((long)bimg * H + h0 + wc * TN + (p >> 4)) * W + w0 + (p & 15);
            if constexpr (FEPI) {
                h16x8 rv = (h16x8)(h16)0;
                if (res) rv = *reinterpret_cast<const h16x8*>(res + gp * y_ld + co0 + wr * 64 + q * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (h16)epi_act<T>((float)o[e] + (float)rv[e], act);
            }
            *reinterpret_cast<h16x8*>(y + gp * y_ld + co0 + wr * 64 + q * 8) = o;
            if constexpr (!FEPI) { if (stat_part) tile_stats_accum(o, ssum, ssq); }
        }
        if constexpr (!FEPI) {
            if (stat_part) tile_stats_store(ssum, ssq, stat_part + ((long)tl * WC + wc) * Cout * 2, co0 + wr * 64 + q * 8, lane);
        }
        return;
    }
    float tsum[TM][4], tsq[TM][4];                          // fp32 / fp32x: statistics of the directly stored fragments
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { tsum[i][r] = 0.f; tsq[i][r] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long p = ((long)bimg * H + h0 + wc * TN + j) * W + w0 + r16;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int co = co0 + (wr * TM + i) * 16 + 4 * g;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r];
                if constexpr (FEPI) { if (scale) v[r] *= scale[co + r]; }
                v[r] += (bias ? bias[co + r] : 0.f);
                if constexpr (FEPI) v[r] = epi_act<T>(v[r] + (res ? (float)res[p * y_ld + co + r] : 0.f), act);
            }
            if constexpr (sizeof(T) == 2) {
                h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                *reinterpret_cast<h16x4*>(y + p * y_ld + co) = o;
            } else {
                *reinterpret_cast<float4*>(y + p * y_ld + co) = make_float4(v[0], v[1], v[2], v[3]);
                if constexpr (!FEPI) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { tsum[i][r] += v[r]; tsq[i][r] = fmaf(v[r], v[r], tsq[i][r]); }
                }
            }
        }
    }
    if constexpr (!FEPI && sizeof(T) == 4) {
        if (stat_part) {
            // transposing fold over the 16 pixel lanes (r16): every step halves the values a lane still owns and adds its partner's half,
            // 16 + 8 + 4 + 2 shuffles for the 32 (sum, sumsq) values instead of 4 per value; lane r16 ends with channel
            // i = r16 >> 2, r = r16 & 3 of its quad group g
            static_assert(TM == 4, "statistics epilogue: four channel fragments per wave");
            float v8[2][4][2], v4[4][2], v2[2][2], v1[2];
            const bool b3 = r16 & 8, b2 = r16 & 4, b1 = r16 & 2, b0 = r16 & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ks = b3 ? tsum[i + 2][r] : tsum[i][r], ss = b3 ? tsum[i][r] : tsum[i + 2][r];
                    const float kq = b3 ? tsq[i + 2][r] : tsq[i][r], sq = b3 ? tsq[i][r] : tsq[i + 2][r];
                    v8[i][r][0] = ks + __shfl_xor(ss, 8);
                    v8[i][r][1] = kq + __shfl_xor(sq, 8);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 2; ++k) v4[r][k] = (b2 ? v8[1][r][k] : v8[0][r][k]) + __shfl_xor(b2 ? v8[0][r][k] : v8[1][r][k], 4);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 2; ++k) v2[r][k] = (b1 ? v4[r + 2][k] : v4[r][k]) + __shfl_xor(b1 ? v4[r][k] : v4[r + 2][k], 2);
#pragma unroll
            for (int k = 0; k < 2; ++k) v1[k] = (b0 ? v2[1][k] : v2[0][k]) + __shfl_xor(b0 ? v2[0][k] : v2[1][k], 1);
            float* row = stat_part + ((long)tl * WC + wc) * Cout * 2;
            *reinterpret_cast<float2*>(row + (co0 + (wr * TM + (r16 >> 2)) * 16 + 4 * g + (r16 & 3)) * 2) = make_float2(v1[0], v1[1]);
        }
    }
}


template <typename T, int TM, int TN, int WR, int NWV = 4, bool RINGP = false>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void conv_nt3_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                       T* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                                       float* __restrict__ stat_part = nullptr) {
    conv_nt3_body<T, TM, TN, WR, NWV, RINGP, false>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, nullptr, nullptr, 0, stat_part);
}
template <int TM, int TN, int WR, bool RINGP = false, int HL = 2>
__global__ __launch_bounds__(256, 2) void conv_nt3hl_kernel(const h16* __restrict__ dy, const h16* __restrict__ w, float* __restrict__ dx, int B, int H,
                                                            int W, int Cin, int Cout, long x_ld, long y_ld, const float* __restrict__ oscale) {
    conv_nt3_body<h16, TM, TN, WR, 4, RINGP, false, HL>(dy, w, nullptr, nullptr, B, H, W, Cin, Cout, x_ld, y_ld, nullptr, nullptr, 0, nullptr, dx, oscale);
}
template <typename T, int TM, int TN, int WR>
__global__ __launch_bounds__(256, 2) void conv_nt3f_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                           T* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                                           const float* __restrict__ scale, const T* __restrict__ res, int act) {
    conv_nt3_body<T, TM, TN, WR, 4, false, true>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, scale, res, act);
}

// ------------------------------------------------------------------------------------------
// v3p: persistent form of the halo-tile kernel for layers with short K loops (Cin <= 256: 18-36 tap steps per tile).
// A block keeps its output-channel slice and walks spatial tiles (stride gridDim.x); the weight/halo DMA pipeline
// runs ACROSS tile boundaries (the next tile's first weight tile and halo are in flight during the current tile's
// last tap steps and epilogue), which removes the per-tile fill/drain bubble that cost v3 ~30 % on the 128-channel
// layers (760 TF/s at 18 steps vs 1050 TF/s at 72 steps).
// ------------------------------------------------------------------------------------------
template <typename T, int TM, int TN, int WR>
__global__ __launch_bounds__(256, 2) void conv_nt3p_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld) {
    using M_ = Mma<T>;
    using Frag = typename M_::Frag;
    constexpr int NWV = 4;
    constexpr int VN = M_::VN, KC = 8 * VN;
    constexpr int WC = NWV / WR;
    constexpr int BCO = WR * TM * 16;
    constexpr int TH = WC * TN, TW = 16, HW_ = TW + 2;
    constexpr int HROWS = (TH + 2) * HW_;
    constexpr int HINST = (HROWS + 7) / 8;
    constexpr int HPW = (HINST + NWV - 1) / NWV;
    static_assert(HPW <= 9, "halo does not fit the 9 tap steps");
    constexpr int PA = BCO / (8 * NWV);
    constexpr int HBYTES = HINST * 1024, WBYTES = BCO * 128;

    __shared__ __attribute__((aligned(16))) char lds[2 * HBYTES + 2 * WBYTES];
    char* Hs = lds;
    char* Ws = lds + 2 * HBYTES;

    const int tiles_w = W / TW, tiles_h = H / TH;
    const int ntile = B * tiles_h * tiles_w;
    const int co0 = blockIdx.y * BCO;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int r16 = lane & 15, g = lane >> 4;
    const int srow = lane >> 3, sch = lane & 7;
    const int kchunks = Cin / KC;

    int wl[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (i * NWV + wave) * 8 + srow;
        wl[i] = (co0 + row) * Cin + (sch ^ (row & 7)) * VN;
    }
    int aoff[2], boff[3][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        aoff[kk] = (wr * TM * 16 + r16) * 128 + (((kk * 4 + g) ^ (r16 & 7)) << 4);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) boff[dw][kk] = (wc * TN * HW_ + r16 + dw) * 128 + (((kk * 4 + g) ^ ((r16 + dw) & 7)) << 4);
    }

    // halo source offsets of one tile (relative to x), -1 for the zero ring
    auto halo_offsets = [&](int tl, long (&hl)[HPW]) {
        const int tw_ = tl % tiles_w, th_ = (tl / tiles_w) % tiles_h, bimg = tl / (tiles_w * tiles_h);
        const int h0 = th_ * TH, w0 = tw_ * TW;
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            const int hr = (k * NWV + wave) * 8 + srow;
            const int hy = hr / HW_, hx = hr - hy * HW_;
            const int hh = h0 - 1 + hy, ww = w0 - 1 + hx;
            const bool ok = (k * NWV + wave) < HINST && hr < HROWS && hh >= 0 && hh < H && ww >= 0 && ww < W;
            hl[k] = ok ? (((long)bimg * H + hh) * W + ww) * x_ld + (sch ^ (hx & 7)) * VN : -1;
        }
    };
    auto stage_w = [&](int tap, int ci0, int buf) {
        const T* wb = w + (long)tap * Cout * Cin + ci0;
        char* Wb = Ws + buf * WBYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(wb + wl[i], Wb + (i * NWV + wave) * 1024);
    };
    auto stage_h = [&](long off, int k, int ci0, int buf) {
        const void* src = off >= 0 ? (const void*)(x + off + ci0) : (const void*)mu_zero_page;
        glds16(src, Hs + buf * HBYTES + (k * NWV + wave) * 1024);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int tl = blockIdx.x;
    if (tl >= ntile) return;
    long hl[HPW], hn[HPW];
    halo_offsets(tl, hl);
    stage_w(0, 0, 0);
#pragma unroll
    for (int k = 0; k < HPW; ++k)
        if (k * NWV + wave < HINST) stage_h(hl[k], k, 0, 0);
    __syncthreads();

    int s = 0, hpar = 0;                                     // running weight-step counter and halo-buffer parity
    while (true) {
        const int tn = tl + gridDim.x;
        const bool has_next = tn < ntile;                    // block-uniform
        if (has_next) halo_offsets(tn, hn);
        for (int c = 0; c < kchunks; ++c, hpar ^= 1) {
            const bool last_c = c + 1 == kchunks;
            const int hbuf = hpar * HBYTES;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
#pragma unroll
                for (int dw = 0; dw < 3; ++dw, ++s) {
                    const int t = dh * 3 + dw;
                    // next weight tile: next tap of this chunk, first tap of the next chunk, or of the next tile
                    if (t < 8) stage_w(t + 1, c * KC, (s + 1) & 1);
                    else if (!last_c) stage_w(0, (c + 1) * KC, (s + 1) & 1);
                    else if (has_next) stage_w(0, 0, (s + 1) & 1);
                    // one halo piece of the next chunk instance per tap step
                    if (t < HPW && t * NWV + wave < HINST) {
                        if (!last_c) stage_h(hl[t], t, (c + 1) * KC, hpar ^ 1);
                        else if (has_next) stage_h(hn[t], t, 0, hpar ^ 1);
                    }
                    const char* Wb = Ws + (s & 1) * WBYTES;
                    const char* Hb = Hs + hbuf + dh * (HW_ * 128);
                    if constexpr (M_::PAIR) {
                        typename M_::Frag2 a[TM], b[TN];
#pragma unroll
                        for (int i = 0; i < TM; ++i) a[i] = M_::ld2(Wb + aoff[0] + i * 2048, Wb + aoff[1] + i * 2048);
#pragma unroll
                        for (int j = 0; j < TN; ++j) b[j] = M_::ld2(Hb + boff[dw][0] + j * (HW_ * 128), Hb + boff[dw][1] + j * (HW_ * 128));
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) M_::mma2(a[i], b[j], acc[i][j]);
                    } else {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        Frag a[TM], b[TN];
                        const char* wa = Wb + aoff[kk];
                        const char* hb = Hb + boff[dw][kk];
#pragma unroll
                        for (int i = 0; i < TM; ++i) a[i] = M_::ld(wa + i * 2048);
#pragma unroll
                        for (int j = 0; j < TN; ++j) b[j] = M_::ld(hb + j * (HW_ * 128));
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) M_::mma(a[i], b[j], acc[i][j]);
                    }
                    }
                    __syncthreads();
                }
            }
        }
        // epilogue of this tile (the next tile's first DMAs are already in flight)
        {
            const int tw_ = tl % tiles_w, th_ = (tl / tiles_w) % tiles_h, bimg = tl / (tiles_w * tiles_h);
            const int h0 = th_ * TH, w0 = tw_ * TW;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const long p = ((long)bimg * H + h0 + wc * TN + j) * W + w0 + r16;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int co = co0 + (wr * TM + i) * 16 + 4 * g;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[i][j][r];
                        if constexpr (std::is_same<T, xh32>::value) v[r] *= 1.0f / (float)(1 << MU_XH_WSHIFT);
                        v[r] += (bias ? bias[co + r] : 0.f);
                    }
                    if constexpr (sizeof(T) == 2) {
                        h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                        *reinterpret_cast<h16x4*>(y + p * y_ld + co) = o;
                    } else {
                        *reinterpret_cast<float4*>(y + p * y_ld + co) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (!has_next) break;
        tl = tn;
#pragma unroll
        for (int k = 0; k < HPW; ++k) hl[k] = hn[k];
    }
}

// ------------------------------------------------------------------------------------------
// v4 for 3x3, fp16, Cout % 128 == 0: the halo-tile kernel re-scheduled as a ping-pong pipeline (cdna guide, "256^2 8-phase
// template": one block per CU, LDS-DMA prefetch that stays in flight across raw s_barriers, counted vmcnt, two wave groups
// staggered by one barrier so that one group's MFMA section overlaps the other group's LDS-read/DMA-issue section).
//   block : 512 threads = 8 waves, tile 128 output channels x (16 x 16) pixels of one image; wave = 64 co x (4 rows x 16) px
//   LDS   : halo 18 x 18 x 128 B double-buffered (2 x 41 KB) + ring of FOUR 16 KB weight tiles (tap x 64 ci) + 1 KB dump
//   phase : half a tap (32 of the 64 ci): 8 ds_read_b128 -> s_barrier -> 16 MFMA -> s_barrier; 4 barriers per tap
//   DMA   : first phase of tap s: one halo piece of the next 64-channel chunk (taps 1..6; a dummy afterwards, so that every
//           wave issues exactly 3 DMAs per tap and vmcnt counts stay uniform) -- its target, the other halo buffer, has no
//           readers during this chunk (tap 0's piece waits for the second phase: that buffer was read until the previous
//           chunk's last phase); second phase: the 2 instructions of this wave's share of W(s+3) into ring slot
//           (s+3)&3 == (s-1)&3, whose last readers passed a barrier a full phase ago.  (All three in the second phase made
//           that read section twice as long as the MFMA section it hides behind: 1 + 2 is 3 % faster on the 32^2 / 16^2 layers.)
//   wait  : s_waitcnt vmcnt(4) (tap 0: 3) in the first phase's MFMA section of tap s retires W(s+1) (issued in tap s-2) and
//           leaves H(s-1), W(s+2) and H(s) in flight; two barriers separate that wait from the first read of W(s+1) -- one
//           more than the unstaggered rule needs, because group B trails group A by one barrier.  The last halo piece H(6)
//           is retired by tap 8's wait, two phases before the next chunk reads it.
//   NOTE  : timing ablations that drop the DMAs (or feed the zero page) leave ZERO operands in LDS; the chip then holds a
//           higher clock and the kernel looks 25-30 % faster (cdna guide rule 25) -- an L2-resident halo source, with real
//           data, changes nothing: the DMA stream does not bound this kernel.
// Versus v3 the weight tile is shared by 256 instead of 128 pixels (half the L2->LDS bytes per flop) and the DMA
// lookahead grows from one tap (~0.25 us, less than an L2 hit under load) to three phases.
// ------------------------------------------------------------------------------------------
#ifndef MU_CONV_NT4
#define MU_CONV_NT4 1
#endif

#ifndef MU_CONV_NT4P
#define MU_CONV_NT4P 1
#endif
#ifndef MU_NT4_MINBLK
#define MU_NT4_MINBLK 0         // (round 5 probe: grids below this many blocks fall back to the 8 x 16-pixel-tile kernel)
#endif
#ifndef MU_CONV_WIDE1X1
#define MU_CONV_WIDE1X1 1
#endif
// (body shared by the training kernel, whose signature and code are exactly what they were without the inference epilogue, and the
//  FEPI kernel below: three more kernel arguments on the hot kernel shifted its code enough to cost 0.1 ms per training step)
// HL (round 6): the two-term data gradient of the fp32x 3x3 layers on this pipeline (see conv_nt_kernel / conv_nt3_body): x = the one-term
// fp16 dy, w = HL rows of 2 * Cin halves, chunk instance c = 2 * chunk + {lo, hi} -- the DMA slot that would fetch the next chunk's halo
// carries a dummy in the even instances (the counted waits stay as they are) --, fp32 output rows through conv_nt4x_kernel's epilogue.
template <bool FEPI, int HL = 0>
__device__ __forceinline__ void conv_nt4_body(const h16* __restrict__ x, const h16* __restrict__ w, const float* __restrict__ bias,
                                              h16* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                              float* __restrict__ stat_part, const float* __restrict__ scale, const h16* __restrict__ res, int act,
                                              float* __restrict__ yf = nullptr, const float* __restrict__ oscale = nullptr) {
    using M_ = Mma<h16>;
    using Frag = M_::Frag;
    constexpr int VN = 8, KC = 64, TM = 4, TN = 4, WC = 4, NWV = 8, BCO = 128;
    constexpr int TH = 16, TW = 16, HW_ = TW + 2, HROWS = (TH + 2) * HW_;       // 324 halo rows of 128 B
    constexpr int HINST = (HROWS + 7) / 8;                                       // 41 wave-DMA instructions (8 rows each)
    constexpr int HPW = 7;                                                       // halo pieces per wave: taps 0..6
    static_assert(HPW * NWV >= HINST, "halo does not fit the 7 tap steps");
    constexpr int HBYTES = HINST * 1024, WBYTES = BCO * 128, NWB = 4;

    __shared__ __attribute__((aligned(16))) char lds[2 * HBYTES + NWB * WBYTES + 1024];
    char* Hs = lds;
    char* Ws = lds + 2 * HBYTES;
    char* dump = lds + 2 * HBYTES + NWB * WBYTES;

    const int tiles_w = W / TW, tiles_h = H / TH;
    const int ntile = B * tiles_h * tiles_w, ncb = Cout / BCO;
    const int L = xcd_remap(blockIdx.x, ntile * ncb);
    const int cb = L % ncb, tl = L / ncb;
    const int co0 = cb * BCO;
    const int tw_ = tl % tiles_w, th_ = (tl / tiles_w) % tiles_h, bimg = tl / (tiles_w * tiles_h);
    const int h0 = th_ * TH, w0 = tw_ * TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar branches
    const int wr = wave >> 2, wc = wave & 3;                 // wr doubles as the ping-pong group (waves 0-3 / 4-7: one of each per SIMD)
    const int r16 = lane & 15, g = lane >> 4;
    const int srow = lane >> 3, sch = lane & 7;

    constexpr bool HL2 = HL == 2;
    const int kchunks = HL2 ? 2 * (Cin / KC) : Cin / KC;     // chunk instances
    const int Cw = HL ? 2 * Cin : Cin;                       // weight row length
    const int nsteps = 9 * kchunks;

    int wl[2];                                               // this wave's two weight-DMA instructions: rows (i*8+wave)*8 + srow
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * NWV + wave) * 8 + srow;
        wl[i] = (co0 + row) * Cw + (sch ^ (row & 7)) * VN;
    }
    int hl[HPW];                                             // halo pieces k*8+wave: element offset, -1 = zero ring
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int hr = (k * NWV + wave) * 8 + srow;
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int hh = h0 - 1 + hy, ww = w0 - 1 + hx;
        const bool ok = hr < HROWS && hh >= 0 && hh < H && ww >= 0 && ww < W;
        hl[k] = ok ? (int)(((long)hh * W + ww) * x_ld) + (sch ^ (hx & 7)) * VN : -1;
    }
    const h16* xb = x + (long)bimg * H * W * x_ld;

    auto stage_w = [&](int s) {                              // W(s) -> ring slot s & 3 (dummy beyond the last step)
        if (s < nsteps) {
            const int tap = s % 9, c_ = s / 9;
            const int ci0 = HL2 ? ((c_ & 1) ? Cin : 0) + (c_ >> 1) * KC : (HL == 1 ? Cin : 0) + c_ * KC;
            const h16* wb = w + (long)tap * Cout * Cw + ci0;
            char* Wb = Ws + (s & 3) * WBYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(wb + wl[i], Wb + (i * NWV + wave) * 1024);
        } else {
            glds16(mu_zero_page, dump);
            glds16(mu_zero_page, dump);
        }
    };
    auto stage_h = [&](int k, int c) {                       // piece k of chunk c's halo -> buffer c & 1 (dummy if none)
        const int off = hl[k];
        const int hc = HL2 ? (c >> 1) : c;                   // HL = 2: instances 2 hc and 2 hc + 1 share a halo, staged for the even one
        if (c < kchunks && (!HL2 || (c & 1) == 0) && k * NWV + wave < HINST) {       // wave-uniform
#ifdef MU_NT4_ABL_NOHALO
            const void* src = (const void*)mu_zero_page;
#else
            const void* src = off >= 0 ? (const void*)(xb + off + hc * KC) : (const void*)mu_zero_page;
#endif
            glds16(src, Hs + (hc & 1) * HBYTES + (k * NWV + wave) * 1024);
        } else {
            glds16(mu_zero_page, dump);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int aoff[2], boff[3][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        aoff[kk] = (wr * TM * 16 + r16) * 128 + (((kk * 4 + g) ^ (r16 & 7)) << 4);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) boff[dw][kk] = (wc * TN * HW_ + r16 + dw) * 128 + (((kk * 4 + g) ^ ((r16 + dw) & 7)) << 4);
    }

    // prologue: halo of chunk 0 and W(0..2), drained; then group B falls one barrier behind
#pragma unroll
    for (int k = 0; k < HPW; ++k) stage_h(k, 0);
    stage_w(0);
    stage_w(1);
    stage_w(2);
#ifdef MU_NT4_ABL_NODMA_REAL
    // timing-only ablation (wrong results): every ring slot and both halo buffers hold REAL data from here on and the tap loop issues no
    // DMA and waits for none -- the upper bound of what loader waves that own all DMA issue could give the consumer waves
#pragma unroll
    for (int k = 0; k < HPW; ++k) stage_h(k, 1);
    stage_w(3);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    int s = 0;
    for (int c = 0; c < kchunks; ++c) {
        const int hbuf = ((HL2 ? (c >> 1) : c) & 1) * HBYTES;
#pragma unroll
        for (int t = 0; t < 9; ++t, ++s) {
            const int dh = t / 3, dw = t % 3;
            const char* Wb = Ws + (s & 3) * WBYTES;
            const char* Hb = Hs + hbuf + dh * (HW_ * 128);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                Frag a[TM], b[TN];
                const char* wa = Wb + aoff[kk];
                const char* hb = Hb + boff[dw][kk];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = M_::ld(wa + i * 2048);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = M_::ld(hb + j * (HW_ * 128));
                // exactly three DMAs per wave per tap: the halo piece in the first phase (tap 0: in the second -- the buffer it
                // refills was read until the previous chunk's last phase), the two weight pieces in the second
#ifndef MU_NT4_ABL_NODMA_REAL
                if ((kk == 0) == (t != 0)) {
                    if (t < HPW) stage_h(t, c + 1); else glds16(mu_zero_page, dump);
                }
                if (kk == 1) stage_w(s + 3);
#endif
#ifndef MU_NT4_ABL_NOBAR1
                __builtin_amdgcn_s_barrier();
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#ifndef MU_NT4_ABL_NOPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) M_::mma(a[i], b[j], acc[i][j]);
#ifndef MU_NT4_ABL_NOPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
#if !defined(MU_NT4_ABL_NOWAIT) && !defined(MU_NT4_ABL_NODMA_REAL)
                if (kk == 0) {
                    if (t == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");      // younger: H(8), W(s+2) x2
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             // younger: H(t-1), W(s+2) x2, H(t)
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
#ifndef MU_NT4_ABL_NOBAR2
                __builtin_amdgcn_s_barrier();
#endif
            }
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();

    if constexpr (HL) {
        // fp32 rows, un-scaled: conv_nt4x_kernel's epilogue (the wave's 64 px x 64 co tile as [pixel][16 slots of 16 B], slot ^ (p & 15),
        // read back as whole 256-byte pixel rows)
        const float os = oscale[1] * (1.0f / (float)(1 << MU_XH_WSHIFT));
        char* Of = lds + wave * 16384;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int p = j * 16 + r16;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int co = i * 16 + 4 * g;
                *reinterpret_cast<f32x4*>(Of + p * 256 + ((((co >> 2)) ^ (p & 15)) << 4)) = acc[i][j] * os;
            }
        }
        const int q4 = lane & 15, pl = lane >> 4;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int p = it * 4 + pl;
            const f32x4 o = *reinterpret_cast<const f32x4*>(Of + p * 256 + ((q4 ^ (p & 15)) << 4));
            const long gp = ((long)bimg * H + h0 + wc * TN + (p >> 4)) * W + w0 + (p & 15);
            *reinterpret_cast<f32x4*>(yf + gp * y_ld + co0 + wr * 64 + q4 * 4) = o;
        }
        return;
    }
    // Epilogue, wave-private and barrier-free (all LDS is free now): every wave stages its own 64 co x 64 px tile (8 KB as
    // [pixel][64 co], 128-byte rows) and reads it back as 16-byte pieces, so each wave store writes eight whole 128-byte rows
    // instead of 32-byte fragments.  8-byte slot XORed with ((p >> 1) & 7) << 1 on the write == 16-byte slot ^ ((p >> 1) & 7)
    // on the read: both sides bank-conflict free.
    char* Os = lds + wave * 8192;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = j * 16 + r16;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int co = i * 16 + 4 * g;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r];
                if constexpr (FEPI) { if (scale) v[r] *= scale[co0 + wr * 64 + co + r]; }
                v[r] += (bias ? bias[co0 + wr * 64 + co + r] : 0.f);
            }
            h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
            *reinterpret_cast<h16x4*>(Os + p * 128 + (((co >> 2) ^ (((p >> 1) & 7) << 1)) << 3)) = o;
        }
    }
    const int q = lane & 7;
    float ssum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ssq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int p = it * 8 + (lane >> 3);                 // pixel inside the wave tile: image row wc*4 + (p >> 4), column p & 15
        h16x8 o = *reinterpret_cast<const h16x8*>(Os + p * 128 + ((q ^ ((p >> 1) & 7)) << 4));
        const long gp = ((long)bimg * H + h0 + wc * TN + (p >> 4)) * W + w0 + (p & 15);
        if constexpr (FEPI) {
            h16x8 rv = (h16x8)(h16)0;
            if (res) rv = *reinterpret_cast<const h16x8*>(res + gp * y_ld + co0 + wr * 64 + q * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (h16)epi_act<h16>((float)o[e] + (float)rv[e], act);
        }
#ifdef MU_NT4_ABL_NOSTORE
        if (B < 0)
#endif
        *reinterpret_cast<h16x8*>(y + gp * y_ld + co0 + wr * 64 + q * 8) = o;
        if (stat_part) tile_stats_accum(o, ssum, ssq);
    }
    if (stat_part) tile_stats_store(ssum, ssq, stat_part + ((long)tl * 4 + wc) * Cout * 2, co0 + wr * 64 + q * 8, lane);
}

__global__ __launch_bounds__(512, 1) void conv_nt4_kernel(const h16* __restrict__ x, const h16* __restrict__ w, const float* __restrict__ bias,
                                                          h16* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                                          float* __restrict__ stat_part) {
    conv_nt4_body<false>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, stat_part, nullptr, nullptr, 0);
}
template <int HL>
__global__ __launch_bounds__(512, 1) void conv_nt4hl_kernel(const h16* __restrict__ dy, const h16* __restrict__ w, float* __restrict__ dx, int B, int H,
                                                            int W, int Cin, int Cout, long x_ld, long y_ld, const float* __restrict__ oscale) {
    conv_nt4_body<false, HL>(dy, w, nullptr, nullptr, B, H, W, Cin, Cout, x_ld, y_ld, nullptr, nullptr, nullptr, 0, dx, oscale);
}
__global__ __launch_bounds__(512, 1) void conv_nt4f_kernel(const h16* __restrict__ x, const h16* __restrict__ w, const float* __restrict__ bias,
                                                           h16* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                                           const float* __restrict__ scale, const h16* __restrict__ res, int act) {
    conv_nt4_body<true>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, nullptr, scale, res, act);
}

// ------------------------------------------------------------------------------------------
// v4x: the ping-pong pipeline for fp32x (round 5).  Byte for byte the LDS image, DMA stream, counted waits and barrier schedule of
// conv_nt4_body -- a K chunk is 32 fp32 channels = the same 128-byte rows (chunk-encoded: eight [4 bf16 hi | 4 bf16 lo] chunks), so halo
// (2 x 41 KB), weight ring (4 x 16 KB) and the three-DMAs-per-wave-per-tap accounting carry over; what changes is the arithmetic per
// (tap, chunk) step: ONE K = 32 step of three bf16 MFMAs per tile pair (lo hi + hi lo + hi hi) instead of two K = 32 steps of one, i.e.
// 48 MFMAs per wave and step for the same 16 fragment reads (fp16: 32).  The step is split into two phases by PIXEL rows instead of by K:
// phase 0 reads the four weight fragments (both chunks of a lane group: chunk g and g + 4 side by side are the eight-value operand) and
// the first two pixel-row fragments -> 24 MFMAs; phase 1 reads the other two -> 24 MFMAs.  Until round 5 the fp32x mode ran the generic
// two-stage kernel (conv_nt3: one __syncthreads per tap, 8 x 16 pixel tiles, 55 % MFMA-busy).  The epilogue writes fp32 rows (a wave's
// 64 px x 64 co tile staged in 16 KB of LDS, XOR-swizzled 16-byte slots, whole 256-byte rows per store) and, like the fp16 kernel, the
// per-tile BatchNorm statistics of the values it stores.
// ------------------------------------------------------------------------------------------
#ifndef MU_CONV_NT4X
#define MU_CONV_NT4X 1
#endif
template <typename XT>           // xf32 (bf16 pairs) or xh32 (fp16 pairs, round 6: what the 3x3 layers of the fp32x mode run on)
__global__ __launch_bounds__(512, 1) void conv_nt4x_kernel(const XT* __restrict__ x, const XT* __restrict__ w, const float* __restrict__ bias,
                                                           XT* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                                           float* __restrict__ stat_part) {
    using M_ = Mma<XT>;
    using Frag2 = typename M_::Frag2;
    constexpr int VN = 4, KC = 32, TM = 4, TN = 4, NWV = 8, BCO = 128;
    constexpr int TH = 16, TW = 16, HW_ = TW + 2, HROWS = (TH + 2) * HW_;       // 324 halo rows of 128 B
    constexpr int HINST = (HROWS + 7) / 8;                                       // 41 wave-DMA instructions (8 rows each)
    constexpr int HPW = 7;                                                       // halo pieces per wave: taps 0..6
    constexpr int HBYTES = HINST * 1024, WBYTES = BCO * 128, NWB = 4;

    __shared__ __attribute__((aligned(16))) char lds[2 * HBYTES + NWB * WBYTES + 1024];
    char* Hs = lds;
    char* Ws = lds + 2 * HBYTES;
    char* dump = lds + 2 * HBYTES + NWB * WBYTES;

    const int tiles_w = W / TW, tiles_h = H / TH;
    const int ntile = B * tiles_h * tiles_w, ncb = Cout / BCO;
    const int L = xcd_remap(blockIdx.x, ntile * ncb);
    const int cb = L % ncb, tl = L / ncb;
    const int co0 = cb * BCO;
    const int tw_ = tl % tiles_w, th_ = (tl / tiles_w) % tiles_h, bimg = tl / (tiles_w * tiles_h);
    const int h0 = th_ * TH, w0 = tw_ * TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // wr doubles as the ping-pong group (waves 0-3 / 4-7: one of each per SIMD)
    const int r16 = lane & 15, g = lane >> 4;
    const int srow = lane >> 3, sch = lane & 7;

    const int kchunks = Cin / KC;
    const int nsteps = 9 * kchunks;

    int wl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * NWV + wave) * 8 + srow;
        wl[i] = (co0 + row) * Cin + (sch ^ (row & 7)) * VN;
    }
    int hl[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int hr = (k * NWV + wave) * 8 + srow;
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int hh = h0 - 1 + hy, ww = w0 - 1 + hx;
        const bool ok = hr < HROWS && hh >= 0 && hh < H && ww >= 0 && ww < W;
        hl[k] = ok ? (int)(((long)hh * W + ww) * x_ld) + (sch ^ (hx & 7)) * VN : -1;
    }
    const XT* xb = x + (long)bimg * H * W * x_ld;

    auto stage_w = [&](int s) {
        if (s < nsteps) {
            const int tap = s % 9, ci0 = (s / 9) * KC;
            const XT* wb = w + (long)tap * Cout * Cin + ci0;
            char* Wb = Ws + (s & 3) * WBYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(wb + wl[i], Wb + (i * NWV + wave) * 1024);
        } else {
            glds16(mu_zero_page, dump);
            glds16(mu_zero_page, dump);
        }
    };
    auto stage_h = [&](int k, int c) {
        const int off = hl[k];
        if (c < kchunks && k * NWV + wave < HINST) {        // wave-uniform
            const void* src = off >= 0 ? (const void*)(xb + off + c * KC) : (const void*)mu_zero_page;
            glds16(src, Hs + (c & 1) * HBYTES + (k * NWV + wave) * 1024);
        } else {
            glds16(mu_zero_page, dump);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int aoff[2], boff[3][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        aoff[kk] = (wr * TM * 16 + r16) * 128 + (((kk * 4 + g) ^ (r16 & 7)) << 4);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) boff[dw][kk] = (wc * TN * HW_ + r16 + dw) * 128 + (((kk * 4 + g) ^ ((r16 + dw) & 7)) << 4);
    }

#pragma unroll
    for (int k = 0; k < HPW; ++k) stage_h(k, 0);
    stage_w(0);
    stage_w(1);
    stage_w(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    int s = 0;
    for (int c = 0; c < kchunks; ++c) {
        const int hbuf = (c & 1) * HBYTES;
#pragma unroll
        for (int t = 0; t < 9; ++t, ++s) {
            const int dh = t / 3, dw = t % 3;
            const char* Wb = Ws + (s & 3) * WBYTES;
            const char* Hb = Hs + hbuf + dh * (HW_ * 128);
            const char* wa0 = Wb + aoff[0];
            const char* wa1 = Wb + aoff[1];
            const char* hb0 = Hb + boff[dw][0];
            const char* hb1 = Hb + boff[dw][1];
            Frag2 a[TM], b[2];
            // ---- phase 0: the four weight fragments + pixel rows 0, 1 of the wave's four
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = M_::ld2(wa0 + i * 2048, wa1 + i * 2048);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = M_::ld2(hb0 + j * (HW_ * 128), hb1 + j * (HW_ * 128));
            if (t != 0) {
                if (t < HPW) stage_h(t, c + 1); else glds16(mu_zero_page, dump);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) M_::mma2(a[i], b[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            if (t == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");      // younger: H(8), W(s+2) x2
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             // younger: H(t-1), W(s+2) x2, H(t)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 1: pixel rows 2, 3 (the weight fragments stay in registers)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = M_::ld2(hb0 + (j + 2) * (HW_ * 128), hb1 + (j + 2) * (HW_ * 128));
            if (t == 0) stage_h(0, c + 1);
            stage_w(s + 3);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) M_::mma2(a[i], b[j], acc[i][j + 2]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();

    // Epilogue, wave-private and barrier-free: the wave's 64 px x 64 co fp32 tile as [pixel][16 slots of 16 B], slot XORed with (p & 15)
    // (the 16 lanes that write one (i, g) slot hold 16 different pixels), read back as whole 256-byte pixel rows: 4 rows per wave store.
    mma_unshift<XT>(acc);
    char* Os = lds + wave * 16384;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = j * 16 + r16;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int co = i * 16 + 4 * g;
            f32x4 v = acc[i][j];
            if (bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bias[co0 + wr * 64 + co + r];
            }
            *reinterpret_cast<f32x4*>(Os + p * 256 + ((((co >> 2)) ^ (p & 15)) << 4)) = v;
        }
    }
    const int q = lane & 15, pl = lane >> 4;
    f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int p = it * 4 + pl;                            // pixel inside the wave tile: image row wc*4 + (p >> 4), column p & 15
        const f32x4 o = *reinterpret_cast<const f32x4*>(Os + p * 256 + ((q ^ (p & 15)) << 4));
        const long gp = ((long)bimg * H + h0 + wc * TN + (p >> 4)) * W + w0 + (p & 15);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + gp * y_ld + co0 + wr * 64 + q * 4) = o;
        if (stat_part) {
            ssum += o;
            ssq += o * o;
        }
    }
    if (stat_part) {                                          // fold the four pixel groups (lanes q, q+16, q+32, q+48), lanes 0-15 write
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ssum[c] += __shfl_xor(ssum[c], 16); ssq[c] += __shfl_xor(ssq[c], 16);
            ssum[c] += __shfl_xor(ssum[c], 32); ssq[c] += __shfl_xor(ssq[c], 32);
        }
        if (lane < 16) {
            float* row = stat_part + ((long)tl * 4 + wc) * Cout * 2 + (co0 + wr * 64 + q * 4) * 2;
            *reinterpret_cast<float4*>(row) = make_float4(ssum[0], ssq[0], ssum[1], ssq[1]);
            *reinterpret_cast<float4*>(row + 4) = make_float4(ssum[2], ssq[2], ssum[3], ssq[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// v4p: the ping-pong kernel as a PERSISTENT tile loop -- one block per CU walks the 16x16 tiles of its output-channel block.
// Measured on v4 (in-process, 128->128 @128^2): every tile pays ~4.8 us of launch / index / prologue (halo + three weight
// tiles behind a vmcnt(0)) / epilogue around 18 taps x 0.87 us.  Here the DMA stream simply runs on across the tile
// boundary: the last chunk of a tile prefetches the NEXT tile's first halo (pieces at taps 0..6) and its first three weight
// tiles, so only the first tile of a block has a prologue.  The epilogue stages the output through the just-freed halo
// buffer in two 32 KB halves (the other buffer already holds the next tile's halo).  Its 8 global stores per thread count
// in vmcnt like the DMAs: the first two taps of a tile wait with vmcnt(11) / vmcnt(12) = 3 / 4 DMAs + 8 younger stores, afterwards vmcnt(4).
// LDS-DMA through inline asm (glds16a): the epilogue's LDS writes would otherwise be fenced with vmcnt(0) by the compiler.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void conv_nt4p_kernel(const h16* __restrict__ x, const h16* __restrict__ w, const float* __restrict__ bias,
                                                           h16* __restrict__ y, int B, int H, int W, int Cin, int Cout, long x_ld, long y_ld,
                                                           float* __restrict__ stat_part) {
    using M_ = Mma<h16>;
    using Frag = M_::Frag;
    constexpr int VN = 8, KC = 64, TM = 4, TN = 4, NWV = 8, BCO = 128;
    constexpr int TH = 16, TW = 16, HW_ = TW + 2, HROWS = (TH + 2) * HW_;
    constexpr int HINST = (HROWS + 7) / 8;
    constexpr int HPW = 7;
    constexpr int HBYTES = HINST * 1024, WBYTES = BCO * 128, NWB = 4;

    __shared__ __attribute__((aligned(16))) char lds[2 * HBYTES + NWB * WBYTES + 1024 + 8192];
    char* Hs = lds;
    char* Ws = lds + 2 * HBYTES;
    char* dump = lds + 2 * HBYTES + NWB * WBYTES;
    char* spare = dump + 1024;                               // 8 KB: wave 7's output stage

    const int tiles_w = W / TW, tiles_h = H / TH;
    const int ntile = B * tiles_h * tiles_w, ncb = Cout / BCO;
    const int nblk = gridDim.x / ncb;                        // blocks per output-channel block
    const int L = xcd_remap(blockIdx.x, gridDim.x);          // the ncb blocks sweeping the same tiles sit on one XCD
    const int cb = L % ncb, bx = L / ncb;
    const int co0 = cb * BCO;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, g = lane >> 4;
    const int srow = lane >> 3, sch = lane & 7;
    const int kchunks = Cin / KC;

    int wl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * NWV + wave) * 8 + srow;
        wl[i] = (co0 + row) * Cin + (sch ^ (row & 7)) * VN;
    }
    // per-lane halo source offsets of tile tl (element offsets inside its image, -1 = zero ring) and that image's base.
    // Recomputed at every chunk start for the chunk's prefetch target (~80 VALU against 9 taps of MFMA work) rather than
    // kept for two tiles: the kernel sits at the register limit.
    int hl[HPW];
    const h16* xb;
    auto halo_offsets = [&](int tl) {
        const int tw_ = tl % tiles_w, th_ = (tl / tiles_w) % tiles_h, bi = tl / (tiles_w * tiles_h);
        const int hh0 = th_ * TH - 1, ww0 = tw_ * TW - 1;
        xb = x + (long)bi * H * W * x_ld;
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            const int hr = (k * NWV + wave) * 8 + srow;
            const int hy = hr / HW_, hx = hr - hy * HW_;
            const int hh = hh0 + hy, ww = ww0 + hx;
            const bool ok = hr < HROWS && hh >= 0 && hh < H && ww >= 0 && ww < W;
            hl[k] = ok ? (int)(((long)hh * W + ww) * x_ld) + (sch ^ (hx & 7)) * VN : -1;
        }
    };
    auto stage_w = [&](int slot, int tap, int chunk, bool real) {     // one (tap, 64-channel chunk) weight tile -> ring slot
        if (real) {
            const h16* wb = w + (long)tap * Cout * Cin + chunk * KC;
            char* Wb = Ws + slot * WBYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16a(wb + wl[i], Wb + (i * NWV + wave) * 1024);
        } else {
            glds16a(mu_zero_page, dump);
            glds16a(mu_zero_page, dump);
        }
    };
    auto stage_h = [&](int k, int chunk, int buf, bool real) {
        if (real && k * NWV + wave < HINST) {
            const int off = hl[k];
            const void* src = off >= 0 ? (const void*)(xb + off + chunk * KC) : (const void*)mu_zero_page;
            glds16a(src, Hs + buf * HBYTES + (k * NWV + wave) * 1024);
        } else {
            glds16a(mu_zero_page, dump);
        }
    };

    int aoff[2], boff[3][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        aoff[kk] = (wr * TM * 16 + r16) * 128 + (((kk * 4 + g) ^ (r16 & 7)) << 4);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) boff[dw][kk] = (wc * TN * HW_ + r16 + dw) * 128 + (((kk * 4 + g) ^ ((r16 + dw) & 7)) << 4);
    }

    int tl = bx;
    if (tl >= ntile) return;                                 // (whole block: no barrier has been executed yet)
    halo_offsets(tl);

    // prologue of the block's first tile: halo of chunk 0 and W(0..2), drained; then group B falls one barrier behind
#pragma unroll
    for (int k = 0; k < HPW; ++k) stage_h(k, 0, 0, true);
#pragma unroll
    for (int q = 0; q < 3; ++q) stage_w(q, q % 9, q / 9, q < 9 * kchunks);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    int sg = 0, G = 0;                                       // running tap step (weight ring slot) and chunk (halo buffer) counters
    for (; tl < ntile; tl += nblk) {
        const bool has_next = tl + nblk < ntile;
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < kchunks; ++c, ++G) {
            const int hbuf = (G & 1) * HBYTES;
            const bool last_chunk = c + 1 == kchunks;
            // this chunk prefetches the halo of the tile's next chunk, or of the next tile's first chunk
            if (last_chunk) { if (has_next) halo_offsets(tl + nblk); }
            else if (c == 0) halo_offsets(tl);
            const bool pf_real = !last_chunk || has_next;
            const int pf_chunk = last_chunk ? 0 : c + 1;
#pragma unroll
            for (int t = 0; t < 9; ++t, ++sg) {
                const int dh = t / 3, dw = t % 3;
                const char* Wb = Ws + (sg & 3) * WBYTES;
                const char* Hb = Hs + hbuf + dh * (HW_ * 128);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    Frag a[TM], b[TN];
                    const char* wa = Wb + aoff[kk];
                    const char* hb = Hb + boff[dw][kk];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = M_::ld(wa + i * 2048);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = M_::ld(hb + j * (HW_ * 128));
                    // exactly three DMAs per wave per tap: the halo piece in the first phase (tap 0: in the second, behind two more
                    // barriers -- its buffer was read until the previous chunk's last phase and staged the previous tile's output),
                    // the two weight pieces in the second
                    if ((kk == 0) == (t != 0)) {
                        if (t < HPW) stage_h(t, pf_chunk, (G + 1) & 1, pf_real);
                        else glds16a(mu_zero_page, dump);
                    }
                    if (kk == 1) {
                        const int t3 = (t + 3) % 9, c3 = c + (t + 3) / 9;       // the tap three steps ahead (may be the next tile's)
                        if (c3 < kchunks) stage_w((sg + 3) & 3, t3, c3, true);
                        else stage_w((sg + 3) & 3, t3, 0, has_next);
                    }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) M_::mma(a[i], b[j], acc[i][j]);
                    __builtin_amdgcn_s_setprio(0);
                    if (kk == 0) {
                        // taps 0 and 1 of a tile: the previous tile's 8 output stores are younger than the awaited weight tile
                        // younger than the awaited weight tile: tap 0: H(8), W(s+2) x2 [+ 8 stores]; tap 1: [8 stores +] H(0), W(s+2) x2,
                        // H(1); later taps: H(t-1), W(s+2) x2, H(t)
                        if (t == 0) { if (c == 0) asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
                        else if (t == 1) { if (c == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();          // group A waits for B: both are done with this tile's LDS

        // Epilogue, wave-private and barrier-free: every wave stages its own 64 co x 64 px tile (8 KB as [pixel][64 co], 128-byte
        // rows) in LDS that is free at this point -- waves 0-4 in the halo buffer of the chunk just finished, waves 5-6 in the
        // one weight slot that holds no prefetched tile, wave 7 in a spare 8 KB -- and reads it back as 16-byte pieces, so each
        // wave store writes eight whole 128-byte rows.  8-byte slot XORed with ((p >> 1) & 7) << 1 on the write, i.e. 16-byte
        // slot ^ ((p >> 1) & 7) on the read.  The next DMA into these regions is issued behind two more block barriers.
        char* Os = wave < 5 ? Hs + ((G - 1) & 1) * HBYTES + wave * 8192
                            : (wave < 7 ? Ws + ((sg + 3) & 3) * WBYTES + (wave - 5) * 8192 : spare);
        const int tw_ = tl % tiles_w, th_ = (tl / tiles_w) % tiles_h, bimg = tl / (tiles_w * tiles_h);
        const int h0 = th_ * TH, w0 = tw_ * TW;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int p = j * 16 + r16;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int co = i * 16 + 4 * g;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + (bias ? bias[co0 + wr * 64 + co + r] : 0.f);
                h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                *reinterpret_cast<h16x4*>(Os + p * 128 + (((co >> 2) ^ (((p >> 1) & 7) << 1)) << 3)) = o;
            }
        }
        {
            const int q = lane & 7;
            float ssum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ssq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int p = it * 8 + (lane >> 3);             // pixel inside the wave tile: image row wc*4 + (p >> 4), column p & 15
                const h16x8 o = *reinterpret_cast<const h16x8*>(Os + p * 128 + ((q ^ ((p >> 1) & 7)) << 4));
                const long gp = ((long)bimg * H + h0 + wc * TN + (p >> 4)) * W + w0 + (p & 15);
#ifdef MU_NT4_ABL_NOSTORE
                if (B < 0)
#endif
                *reinterpret_cast<h16x8*>(y + gp * y_ld + co0 + wr * 64 + q * 8) = o;
                if (stat_part) tile_stats_accum(o, ssum, ssq);
            }
            if (stat_part) tile_stats_store(ssum, ssq, stat_part + ((long)tl * 4 + wc) * Cout * 2, co0 + wr * 64 + q * 8, lane);
        }
        if (has_next && wr == 1) __builtin_amdgcn_s_barrier();      // re-stagger
    }
}

// ------------------------------------------------------------------------------------------
// v5, fp16 3x3 with Cin == Cout == 64 (the 64 -> 64 layers at 128^2 / 64^2 and their data gradients): WEIGHTS-RESIDENT persistent kernel.
// The halo-tile kernel spends these layers waiting: with one 64-channel chunk a tile is nine taps of 16 MFMAs per wave, each behind a
// block barrier that waits for an 8 KB weight tile requested one tap earlier (PMC: matrix pipe 25 % busy, 120 us against ~65 us of HBM
// time for the 268 MB of x + y at 128^2).  Here the whole weight tensor (9 x 64 x 64 fp16 = 72 KB) is loaded into LDS ONCE per block and
// stays; a block (one per CU) walks 16 x 16 pixel tiles and the only stream is the 18 x 18 halo (41 KB per tile).
// Two 4-wave groups (one wave per SIMD each) work on alternate tiles, HALF A PERIOD APART: in a phase one group runs its tile's 288 MFMAs
// per wave (64 co x 64 px per wave, no barrier inside) while the other one requests its next halo (LDS-DMA into its own, single, buffer:
// every wave of the group finished reading it before the phase barrier), writes its previous tile's outputs straight from the
// accumulators (8-byte stores, 4 per 128-byte row; BatchNorm statistics of the stored values by a transposing cross-lane fold) and waits
// for the halo -- counted vmcnt: only the output stores are younger.  One block barrier per phase.  154 KB of LDS.
// First form (all 8 waves in lockstep on one tile, double-buffered halo, staged epilogue): 120 -> 88 us at 128^2, B = 64; ablations:
// without MFMAs 40 us, without stores 80, without halo DMAs 71 -- data movement and arithmetic were not overlapping.
// ------------------------------------------------------------------------------------------
#ifndef MU_CONV_NT5
#define MU_CONV_NT5 1
#endif
#ifndef MU_NT5_MINTILES
#define MU_NT5_MINTILES 512          // two tiles per block at least: below, one of the two wave groups would idle
#endif
// Cout_total = 128 (grid.y = 2; launches without a statistics epilogue, i.e. the data-gradient of a 128 -> 64 layer): blockIdx.y picks a
// 64-channel half of the output -- two independent 64 -> 64 problems on the same input, each with its half of the weights resident.
__global__ __launch_bounds__(512, 1) void conv_nt5_kernel(const h16* __restrict__ x, const h16* __restrict__ w, const float* __restrict__ bias,
                                                          h16* __restrict__ y, int B, int H, int W, long x_ld, long y_ld,
                                                          float* __restrict__ stat_part, int Cout_total) {
    using M_ = Mma<h16>;
    using Frag = M_::Frag;
    constexpr int VN = 8, CI = 64, CO = 64, TM = 4, TN = 4, NWV = 8, GW = 4;
    constexpr int TH = 16, TW = 16, HW_ = TW + 2, HROWS = (TH + 2) * HW_;
    constexpr int HINST = (HROWS + 7) / 8, HPW = (HINST + GW - 1) / GW;
    constexpr int HBYTES = HINST * 1024, TAPB = CO * 128, WBYTES = 9 * TAPB;
    constexpr int NST = TM * TN;                              // output stores per wave and tile (one 8-byte store per fragment)

    __shared__ __attribute__((aligned(16))) char lds[2 * HBYTES + WBYTES + 1024];
    char* Ws = lds + 2 * HBYTES;
    char* dump = Ws + WBYTES;

    const int tiles_w = W / TW, tiles_h = H / TH;
    const int ntile = B * tiles_h * tiles_w, nblk = gridDim.x;
    const int bx = xcd_remap(blockIdx.x, gridDim.x);         // neighbouring tiles (shared halo rows) on one XCD's L2
    if (bx >= ntile) return;                                 // (whole block: no barrier has been executed yet)
    const int nt = (ntile - bx + nblk - 1) / nblk;           // tiles of this block: bx, bx + nblk, ...

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3;                // group (tile parity), wave inside the group = 4-row strip of the tile
    const int r16 = lane & 15, g = lane >> 4;
    const int srow = lane >> 3, sch = lane & 7;
    char* Hg = lds + grp * HBYTES;
    const int co_base = blockIdx.y * CO;

    auto stage_halo = [&](int tl) {                          // this group's waves request the halo of tile tl: HPW DMAs per wave
        const int tw_ = tl % tiles_w, th_ = (tl / tiles_w) % tiles_h, bi = tl / (tiles_w * tiles_h);
        const int hh0 = th_ * TH - 1, ww0 = tw_ * TW - 1;
        const h16* xb = x + (long)bi * H * W * x_ld;
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            const int inst = k * GW + wg;
            const int hr = inst * 8 + srow;
            const int hy = hr / HW_, hx = hr - hy * HW_;
            const int hh = hh0 + hy, ww = ww0 + hx;
            const bool ok = hr < HROWS && hh >= 0 && hh < H && ww >= 0 && ww < W;
            const void* src = ok ? (const void*)(xb + ((long)hh * W + ww) * x_ld + (sch ^ (hx & 7)) * VN) : (const void*)mu_zero_page;
            if (inst < HINST) glds16a(src, Hg + inst * 1024);
            else glds16a(mu_zero_page, dump);
        }
    };

    int aoff[2], boff[3][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        aoff[kk] = r16 * 128 + (((kk * 4 + g) ^ (r16 & 7)) << 4);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) boff[dw][kk] = (wg * TN * HW_ + r16 + dw) * 128 + (((kk * 4 + g) ^ ((r16 + dw) & 7)) << 4);
    }
    float bv[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[i][r] = bias ? bias[co_base + i * 16 + 4 * g + r] : 0.f;

    {                                                        // the nine weight tiles, once: [tap][64 co rows of 128 B], 16-byte chunk ^ (row & 7)
        const int row = wave * 8 + srow;
#pragma unroll
        for (int t = 0; t < 9; ++t) glds16a(w + ((long)t * Cout_total + co_base + row) * CI + (sch ^ (row & 7)) * VN, Ws + t * TAPB + wave * 1024);
    }
    if (grp == 0) stage_halo(bx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // phase p: group (p & 1) multiplies local tile p; the other group requests local tile p + 1 and writes out local tile p - 1
    for (int p = 0; p <= nt; ++p) {
        if (grp == (p & 1)) {
            if (p < nt) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dh = t / 3, dw = t % 3;
                    const char* Wb = Ws + t * TAPB;
                    const char* Hb = Hg + dh * (HW_ * 128);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        Frag a[TM], b[TN];
#pragma unroll
                        for (int i = 0; i < TM; ++i) a[i] = M_::ld(Wb + aoff[kk] + i * 2048);
#pragma unroll
                        for (int j = 0; j < TN; ++j) b[j] = M_::ld(Hb + boff[dw][kk] + j * (HW_ * 128));
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) M_::mma(a[i], b[j], acc[i][j]);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the halo buffer is free once every wave of the group is at the barrier
        } else {
            const bool pf = p + 1 < nt, ep = p >= 1;
            if (pf) stage_halo(bx + (p + 1) * nblk);
            if (ep) {
                const int tl = bx + (p - 1) * nblk;
                const int tw_ = tl % tiles_w, th_ = (tl / tiles_w) % tiles_h, bimg = tl / (tiles_w * tiles_h);
                h16* yb = y + (((long)bimg * H + th_ * TH + wg * TN) * W + tw_ * TW + r16) * y_ld + co_base + 4 * g;
                float tsum[TM][4], tsq[TM][4];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { tsum[i][r] = 0.f; tsq[i][r] = 0.f; }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const h16x4 o = {(h16)(acc[i][j][0] + bv[i][0]), (h16)(acc[i][j][1] + bv[i][1]), (h16)(acc[i][j][2] + bv[i][2]),
                                         (h16)(acc[i][j][3] + bv[i][3])};
                        *reinterpret_cast<h16x4*>(yb + (long)j * W * y_ld + i * 16) = o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float v = (float)o[r]; tsum[i][r] += v; tsq[i][r] = fmaf(v, v, tsq[i][r]); }
                    }
                if (stat_part) {
                    // transposing fold over the 16 pixel lanes (see conv_nt3_body): lane r16 ends with channel (r16 >> 2) * 16 + 4 g + (r16 & 3)
                    float v8[2][4][2], v4[4][2], v2[2][2], v1[2];
                    const bool b3 = r16 & 8, b2 = r16 & 4, b1 = r16 & 2, b0 = r16 & 1;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v8[i][r][0] = (b3 ? tsum[i + 2][r] : tsum[i][r]) + __shfl_xor(b3 ? tsum[i][r] : tsum[i + 2][r], 8);
                            v8[i][r][1] = (b3 ? tsq[i + 2][r] : tsq[i][r]) + __shfl_xor(b3 ? tsq[i][r] : tsq[i + 2][r], 8);
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < 2; ++k) v4[r][k] = (b2 ? v8[1][r][k] : v8[0][r][k]) + __shfl_xor(b2 ? v8[0][r][k] : v8[1][r][k], 4);
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int k = 0; k < 2; ++k) v2[r][k] = (b1 ? v4[r + 2][k] : v4[r][k]) + __shfl_xor(b1 ? v4[r][k] : v4[r + 2][k], 2);
#pragma unroll
                    for (int k = 0; k < 2; ++k) v1[k] = (b0 ? v2[1][k] : v2[0][k]) + __shfl_xor(b0 ? v2[0][k] : v2[1][k], 1);
                    *reinterpret_cast<float2*>(stat_part + (((long)tl * GW + wg) * CO + (r16 >> 2) * 16 + 4 * g + (r16 & 3)) * 2) = make_float2(v1[0], v1[1]);
                }
            }
            // the requested halo has landed in this wave's share: only this phase's output stores are younger than its DMAs
            if (!pf || !ep) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (stat_part) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
}
#ifndef MU_NT5_SPLIT128
#define MU_NT5_SPLIT128 1
#endif
#ifndef MU_NT5_SPLIT_MINTILES
#define MU_NT5_SPLIT_MINTILES 1024
#endif
static inline bool nt5_serves(int B, int H, int W, int Cin, int Cout) {
    if (!MU_CONV_NT5 || getenv("MU_CONV_NO_NT5")) return false;
    return Cin == 64 && Cout == 64 && H % 16 == 0 && W % 16 == 0 && (long)B * (H / 16) * (W / 16) >= MU_NT5_MINTILES;
}

// The same dispatch for the inference epilogue y = act(conv * scale + bias + res): FEPI instantiations of the non-persistent kernels
// (the persistent ping-pong kernel counts its epilogue's memory operations in hand-placed vmcnt waits and takes no epilogue loads).
// fp32x: which 3x3 shapes the ping-pong kernel serves.  One 512-thread block per CU: a grid of fewer than ~200 blocks (16^2 256 -> 256 at
// B = 64: 128) leaves half of the chip idle, where the generic kernel's 8 x 16 pixel tiles still fill it (61 vs 72 us in-process)
static inline bool nt4x_serves(int B, int H, int W, int Cin, int Cout) {
    if (!MU_CONV_NT4X || getenv("MU_CONV_NO_NT4")) return false;
    if (Cout % 128 || H % 16 || W % 16 || Cin % 32) return false;
    return (long)B * (H / 16) * (W / 16) * (Cout / 128) >= 192;
}

template <typename T, int TAPS>
static int conv_fwd_fused_launch(const T* x, const T* w, const float* scale, const float* bias, const T* res, int act, T* y, int B, int H, int W,
                                 int Cin, int Cout, long x_ld, long y_ld, hipStream_t st) {
    const long M = (long)B * H * W;
    const int npb = (int)((M + 127) / 128);
    if (TAPS == 9 && (Cin * (int)sizeof(T)) % 128 == 0 && W % 16 == 0) {
        if constexpr (sizeof(T) == 2 && MU_CONV_NT4) {
            if (Cout % 128 == 0 && H % 16 == 0 && Cin % 64 == 0) {
                conv_nt4f_kernel<<<B * (H / 16) * (W / 16) * (Cout / 128), 512, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, scale, res, act);
                return MU_OK;
            }
        }
        if (Cout % 128 == 0 && H % 8 == 0) {
            conv_nt3f_kernel<T, 4, 4, 2><<<B * (H / 8) * (W / 16) * (Cout / 128), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, scale, res, act);
            return MU_OK;
        }
        if (Cout % 64 == 0 && H % 8 == 0) {
            conv_nt3f_kernel<T, 4, 2, 1><<<B * (H / 8) * (W / 16) * (Cout / 64), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, scale, res, act);
            return MU_OK;
        }
    }
    if constexpr (TAPS == 1 && sizeof(T) == 2) {
        if ((Cin * 2) % 128 == 0 && Cout == 160) {                      // the 150-class head: one tile spans all output channels
            conv_nt2_kernel<T, 5, 4, 2, 1, false, true><<<npb, 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, res, scale, act);
            return MU_OK;
        }
    }
    if ((Cin * (int)sizeof(T)) % 128 == 0 && Cout % 64 == 0) {         // LDS-DMA version (run-time epilogue arguments)
        if (Cout % 128 == 0)
            conv_nt2_kernel<T, 4, 4, 2, TAPS, false, true><<<npb * (Cout / 128), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, res, scale, act);
        else
            conv_nt2_kernel<T, 4, 4, 1, TAPS, false, true><<<(int)((M + 255) / 256) * (Cout / 64), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, res, scale, act);
        return MU_OK;
    }
    if (Cout % 128 == 0) conv_nt_kernel<T, 4, 4, 2, TAPS, true><<<npb * (Cout / 128), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, scale, res, act);
    else if (Cout % 64 == 0) conv_nt_kernel<T, 4, 2, 1, TAPS, true><<<npb * (Cout / 64), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, scale, res, act);
    else conv_nt_kernel<T, 2, 2, 1, TAPS, true><<<npb * (Cout / 32), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, scale, res, act);
    return MU_OK;
}

template <typename T, int TAPS>
static int conv_fwd_launch(const T* x, const T* w, const float* bias, T* y, int B, int H, int W, int Cin, int Cout, long x_ld,
                           long y_ld, hipStream_t st, float* stat_part = nullptr) {
    const long M = (long)B * H * W;
    const int npb = (int)((M + 127) / 128);
    if (TAPS == 9 && (Cin * (int)sizeof(T)) % 128 == 0 && W % 16 == 0) {     // halo-tile version
        if (Cout % 128 == 0 && H % 16 == 0 && (long)B * H * W >= 262144 && getenv("MU_CONV_NW8")) {
            // experiment kept for reference: 16x16 spatial tile, 8 waves -- the weight tile is shared by 256 pixels (half the
            // L2->LDS bytes per flop) yet the step time is unchanged (45.39 vs 45.29 ms): v3 is no longer L2->LDS bound
            conv_nt3_kernel<T, 4, 4, 2, 8><<<B * (H / 16) * (W / 16) * (Cout / 128), 512, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
            return MU_OK;
        }
        if constexpr (mu_is_split<T>::value && MU_XF_NT3_RING8) {
            if (Cout % 128 == 0 && H % 16 == 0) {
                conv_nt3_kernel<T, 4, 4, 2, 8, true><<<B * (H / 16) * (W / 16) * (Cout / 128), 512, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
                return MU_OK;
            }
        }
        if constexpr (sizeof(T) == 2) {
            if (nt5_serves(B, H, W, Cin, Cout)) {
                const int ntile5 = B * (H / 16) * (W / 16);
                conv_nt5_kernel<<<ntile5 < 256 ? ntile5 : 256, 512, 0, st>>>((const h16*)x, (const h16*)w, bias, (h16*)y, B, H, W, x_ld, y_ld, stat_part, 64);
                return MU_OK;
            }
            // 64 -> 128 without a statistics epilogue (the data-gradient of the 128 -> 64 layers) as two resident 64 -> 64 halves
            if (MU_NT5_SPLIT128 && !stat_part && Cout == 128 && nt5_serves(B, H, W, Cin, 64) && (long)B * (H / 16) * (W / 16) >= MU_NT5_SPLIT_MINTILES) {
                conv_nt5_kernel<<<dim3(256, 2), 512, 0, st>>>((const h16*)x, (const h16*)w, bias, (h16*)y, B, H, W, x_ld, y_ld, nullptr, 128);
                return MU_OK;
            }
        }
        if constexpr (sizeof(T) == 2 && MU_CONV_NT4) {
            if (Cout % 128 == 0 && H % 16 == 0 && Cin % 64 == 0 && !getenv("MU_CONV_NO_NT4") &&
                (long)B * (H / 16) * (W / 16) * (Cout / 128) >= MU_NT4_MINBLK) {
#if MU_CONV_NT4P
                const int ntile4 = B * (H / 16) * (W / 16), ncb4 = Cout / 128;
                // measured (in-process A/B): 128->128 @128^2 322 -> 307 us, 64->128 @128^2 205 -> 176 us, 256->256 @64^2 equal,
                // 512->512 @32^2 and two-tile blocks 1-3 % slower -> persistent only for >= 4 tiles per block and short K loops
                if (ntile4 * ncb4 >= 1024 && Cin <= 256 && 256 % ncb4 == 0) {
                    conv_nt4p_kernel<<<256, 512, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, stat_part);
                    return MU_OK;
                }
#endif
                conv_nt4_kernel<<<B * (H / 16) * (W / 16) * (Cout / 128), 512, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, stat_part);
                return MU_OK;
            }
        }
        if constexpr (mu_is_split<T>::value && MU_CONV_NT4X) {
            if (nt4x_serves(B, H, W, Cin, Cout)) {
                conv_nt4x_kernel<T><<<B * (H / 16) * (W / 16) * (Cout / 128), 512, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, stat_part);
                return MU_OK;
            }
        }
        if (Cout % 128 == 0 && H % 8 == 0) {
            const int ntile = B * (H / 8) * (W / 16), ncb = Cout / 128;
            // ~2 resident blocks per CU in total (MU_CONV_PERSIST_BLOCKS overrides the block count: parity tests use tiny grids)
            const int per_cb = getenv("MU_CONV_PERSIST_BLOCKS") ? atoi(getenv("MU_CONV_PERSIST_BLOCKS")) : 512 / ncb;
            // measured: 407 vs 418 us on 128->128 @128^2 but 332 vs 318 us on 256->256 @64^2 -- the fill/drain bubble is already
            // hidden by the second resident block, so the persistent walk stays opt-in (MU_CONV_PERSIST / MU_CONV_PERSIST_BLOCKS)
            if (ntile >= 4 * per_cb && (getenv("MU_CONV_PERSIST") || getenv("MU_CONV_PERSIST_BLOCKS"))) {
                conv_nt3p_kernel<T, 4, 4, 2><<<dim3(per_cb, ncb), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
                return MU_OK;
            }
            conv_nt3_kernel<T, 4, 4, 2><<<ntile * ncb, 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, stat_part);
            return MU_OK;
        }
        if (Cout % 64 == 0 && H % 8 == 0) {
            const int grid3 = B * (H / 8) * (W / 16) * (Cout / 64);
            if (Cin * (int)sizeof(T) >= 256) conv_nt3_kernel<T, 4, 2, 1, 4, true><<<grid3, 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, stat_part);
            else conv_nt3_kernel<T, 4, 2, 1><<<grid3, 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld, stat_part);
            return MU_OK;
        }
    }
    if constexpr (TAPS == 1 && sizeof(T) == 2 && MU_CONV_WIDE1X1) {
        // 1x1 layers are HBM-bound streams (q/k/v projection of the N = 16384 block: 134 MB in, 402 MB out): a tile that spans all
        // output channels reads the activations once instead of once per 64-channel tile
        if ((Cin * 2) % 128 == 0 && Cout % 192 == 0) {
            if (Cin == 64 && MU_NT2_SB) conv_nt2_kernel<T, 6, 4, 2, 1, true><<<npb * (Cout / 192), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
            else conv_nt2_kernel<T, 6, 4, 2, 1><<<npb * (Cout / 192), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
            return MU_OK;
        }
        if ((Cin * 2) % 128 == 0 && Cout == 160) {
            if (Cin == 64 && MU_NT2_SB) conv_nt2_kernel<T, 5, 4, 2, 1, true><<<npb, 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
            else conv_nt2_kernel<T, 5, 4, 2, 1><<<npb, 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
            return MU_OK;
        }
    }
    if constexpr (TAPS == 1 && mu_is_split<T>::value && MU_CONV_WIDE1X1) {
        // fp32x: the 150-class head (64 -> 160) on one tile that spans all output channels instead of five 32-wide tiles of the generic
        // register-staged kernel (320 -> ~130 us at B = 64)
        if ((Cin * 4) % 128 == 0 && Cout == 160) {
            conv_nt2_kernel<T, 5, 4, 2, 1><<<npb, 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
            return MU_OK;
        }
    }
    if ((Cin * (int)sizeof(T)) % 128 == 0 && Cout % 64 == 0) {         // LDS-DMA version
        if (Cout % 128 == 0)
            conv_nt2_kernel<T, 4, 4, 2, TAPS><<<npb * (Cout / 128), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
        else
            conv_nt2_kernel<T, 4, 4, 1, TAPS><<<(int)((M + 255) / 256) * (Cout / 64), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
        return MU_OK;
    }
    if (Cout % 128 == 0) {
        conv_nt_kernel<T, 4, 4, 2, TAPS><<<npb * (Cout / 128), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
    } else if (Cout % 64 == 0) {
        conv_nt_kernel<T, 4, 2, 1, TAPS><<<npb * (Cout / 64), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
    } else {
        conv_nt_kernel<T, 2, 2, 1, TAPS><<<npb * (Cout / 32), 256, 0, st>>>(x, w, bias, y, B, H, W, Cin, Cout, x_ld, y_ld);
    }
    return MU_OK;
}

// Rows of per-tile BatchNorm statistics mu_conv_fwd_stats writes for this layer shape (0: the kernel that serves it has no
// statistics epilogue -- run mu_bn_train_stats on the output instead).  One row per 16x16 output tile and 4-image-row group.
extern "C" int mu_conv_stats_rows(int B, int H, int W, int Cin, int Cout, int taps, int dtype) {
    // mirrors conv_fwd_launch<T, 9>: which kernel serves the shape, and how many statistics rows its epilogue writes
    if (taps != 9 || B <= 0 || (dtype != MU_F16 && dtype != MU_F32X)) return 0;
    const int es = dtype == MU_F16 ? 2 : 4;
    if ((Cin * es) % 128 || W % 16 || Cout % 32) return 0;
    if (getenv("MU_CONV_NW8") || getenv("MU_CONV_PERSIST") || getenv("MU_CONV_PERSIST_BLOCKS") || (dtype == MU_F32X && MU_XF_NT3_RING8)) return 0;
    if (dtype == MU_F16 && MU_CONV_NT4 && !getenv("MU_CONV_NO_NT4") && Cout % 128 == 0 && H % 16 == 0 && Cin % 64 == 0 &&
        (long)B * (H / 16) * (W / 16) * (Cout / 128) >= MU_NT4_MINBLK)
        return B * (H / 16) * (W / 16) * 4;
    if (dtype == MU_F32X && nt4x_serves(B, H, W, Cin, Cout)) return B * (H / 16) * (W / 16) * 4;
    if (dtype == MU_F16 && nt5_serves(B, H, W, Cin, Cout)) return B * (H / 16) * (W / 16) * 4;     // conv_nt5_kernel: one row per wave of a group
    if (!MU_NT3_STATS || H % 8) return 0;
    if (Cout % 128 == 0) return B * (H / 8) * (W / 16) * 2;           // conv_nt3_kernel<T, 4, 4, 2>: two wave rows per 8 x 16 tile
    if (Cout % 64 == 0) return B * (H / 8) * (W / 16) * 4;            // conv_nt3_kernel<T, 4, 2, 1>: four
    return 0;
}

extern "C" int mu_conv_fwd_stats(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                                 int taps, long x_ld, long y_ld, int dtype, float* stat_part, void* stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return MU_ERR_ARG;
    if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32 || x_ld < Cin || y_ld < Cout || x_ld % 8 || y_ld % 8) return MU_ERR_SHAPE;
    if (stat_part && mu_conv_stats_rows(B, H, W, Cin, Cout, taps, dtype) == 0) return MU_ERR_SHAPE;
    if (!stat_part) return mu_conv_fwd(x, w, bias, y, B, H, W, Cin, Cout, taps, x_ld, y_ld, dtype, stream);
    if (dtype == MU_F32X) conv_fwd_launch<xh32, 9>((const xh32*)x, (const xh32*)w, bias, (xh32*)y, B, H, W, Cin, Cout, x_ld, y_ld, (hipStream_t)stream, stat_part);
    else conv_fwd_launch<h16, 9>((const h16*)x, (const h16*)w, bias, (h16*)y, B, H, W, Cin, Cout, x_ld, y_ld, (hipStream_t)stream, stat_part);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_conv_fwd_fused(const void* x, const void* w, const float* scale, const float* bias, const void* res, int act, void* y, int B,
                                 int H, int W, int Cin, int Cout, int taps, long x_ld, long y_ld, int dtype, void* stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return MU_ERR_ARG;
    if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32 || x_ld < Cin || y_ld < Cout || x_ld % 8 || y_ld % 8) return MU_ERR_SHAPE;
    if ((taps != 1 && taps != 9) || act < MU_ACT_NONE || act > MU_ACT_RELU) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F16) {
        if (taps == 9) conv_fwd_fused_launch<h16, 9>((const h16*)x, (const h16*)w, scale, bias, (const h16*)res, act, (h16*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
        else conv_fwd_fused_launch<h16, 1>((const h16*)x, (const h16*)w, scale, bias, (const h16*)res, act, (h16*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
    } else if (dtype == MU_F32) {
        if (taps == 9) conv_fwd_fused_launch<float, 9>((const float*)x, (const float*)w, scale, bias, (const float*)res, act, (float*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
        else conv_fwd_fused_launch<float, 1>((const float*)x, (const float*)w, scale, bias, (const float*)res, act, (float*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
    } else if (dtype == MU_F32X) {
        if (taps == 9) conv_fwd_fused_launch<xh32, 9>((const xh32*)x, (const xh32*)w, scale, bias, (const xh32*)res, act, (xh32*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
        else conv_fwd_fused_launch<xf32, 1>((const xf32*)x, (const xf32*)w, scale, bias, (const xf32*)res, act, (xf32*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
    } else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_conv_fwd(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int taps,
                           long x_ld, long y_ld, int dtype, void* stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return MU_ERR_ARG;
    if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32 || x_ld < Cin || y_ld < Cout || x_ld % 8 || y_ld % 8) return MU_ERR_SHAPE;
    if (taps != 1 && taps != 9) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F16) {
        if (taps == 9) conv_fwd_launch<h16, 9>((const h16*)x, (const h16*)w, bias, (h16*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
        else conv_fwd_launch<h16, 1>((const h16*)x, (const h16*)w, bias, (h16*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
    } else if (dtype == MU_F32) {
        if (taps == 9) conv_fwd_launch<float, 9>((const float*)x, (const float*)w, bias, (float*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
        else conv_fwd_launch<float, 1>((const float*)x, (const float*)w, bias, (float*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
    } else if (dtype == MU_F32X) {
        if (taps == 9) conv_fwd_launch<xh32, 9>((const xh32*)x, (const xh32*)w, bias, (xh32*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
        else conv_fwd_launch<xf32, 1>((const xf32*)x, (const xf32*)w, bias, (xf32*)y, B, H, W, Cin, Cout, x_ld, y_ld, st);
    } else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// fp32x, 3x3 layers: the TWO-TERM data gradient (round 6).  dx = conv3x3(dy, flipped / transposed weights) with dy as ONE fp16 operand
// S * dy (mu_bn_act_bwd_h / mu_bn_pair_bwd_h / mu_dy_encode_h: rows of Cin halves, row stride dy_ld halves) against the fp16 (hi, lo) pair
// of the weights (the HL data-gradient block of mu_prep_weight / mu_prep_weights_multi with MU_F32X: [9][Cout][2 * Cin] halves): two fp16
// MFMAs per product instead of the forward's three.  dx: plain fp32 rows (row stride dx_ld floats), multiplied by dy_scale[1] = 1 / S and by
// 2^-MU_XH_WSHIFT in the epilogue (exact: powers of two).  Cin = channels of dy (the layer's output), Cout = channels of dx (its input).
// (1 = the weights as one fp16 term, one MFMA per product: step -1.5 ms same-box, but the small-module goldens' parameter gradients land
//  at 1.06e-3 .. 1.17e-3 against their 1e-3 gate -- 9 * Cout = 72 .. 288 terms average too little --, so the default keeps both halves)
#ifndef MU_DGRAD_H_TERMS
#define MU_DGRAD_H_TERMS 2
#endif
extern "C" int mu_conv_dgrad_h(const void* dy_h, const void* w_hl, const float* dy_scale, void* dx, int B, int H, int W, int Cin, int Cout,
                               long dy_ld, long dx_ld, void* stream) {
    if (!dy_h || !w_hl || !dy_scale || !dx || B <= 0 || H <= 0 || W <= 0) return MU_ERR_ARG;
    if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32 || dy_ld < Cin || dx_ld < Cout || dy_ld % 8 || dx_ld % 4) return MU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const h16* x = (const h16*)dy_h;
    const h16* w = (const h16*)w_hl;
    float* y = (float*)dx;
    const long M = (long)B * H * W;
    const int npb = (int)((M + 127) / 128);
    // weight terms: 2 = lo + hi halves (default, MU_DGRAD_H_TERMS), 1 = the hi halves only; MU_DGRAD_H_TERMS in the environment overrides (A/B aid)
    const int terms = getenv("MU_DGRAD_H_TERMS") ? atoi(getenv("MU_DGRAD_H_TERMS")) : MU_DGRAD_H_TERMS;
#define MU_DGH(KERNEL2, KERNEL1, GRID, BLK) do { if (terms == 2) KERNEL2<<<GRID, BLK, 0, st>>>(x, w, y, B, H, W, Cin, Cout, dy_ld, dx_ld, dy_scale); \
                                                 else KERNEL1<<<GRID, BLK, 0, st>>>(x, w, y, B, H, W, Cin, Cout, dy_ld, dx_ld, dy_scale); } while (0)
#define MU_DGG(TM_, TN_, WR_, GRID) do { if (terms == 2) conv_nt_kernel<h16, TM_, TN_, WR_, 9, false, 2><<<GRID, 256, 0, st>>>(x, w, nullptr, nullptr, B, H, W, Cin, Cout, dy_ld, dx_ld, nullptr, nullptr, 0, y, dy_scale); \
                                         else conv_nt_kernel<h16, TM_, TN_, WR_, 9, false, 1><<<GRID, 256, 0, st>>>(x, w, nullptr, nullptr, B, H, W, Cin, Cout, dy_ld, dx_ld, nullptr, nullptr, 0, y, dy_scale); } while (0)
    if (Cin % 64 == 0 && W % 16 == 0 && H % 8 == 0 && Cout % 64 == 0 && !getenv("MU_DGRAD_H_GENERIC")) {
        if (Cout % 128 == 0 && H % 16 == 0 && MU_CONV_NT4 && !getenv("MU_CONV_NO_NT4"))
            MU_DGH(conv_nt4hl_kernel<2>, conv_nt4hl_kernel<1>, B * (H / 16) * (W / 16) * (Cout / 128), 512);
        else if (Cout % 128 == 0)
            MU_DGH((conv_nt3hl_kernel<4, 4, 2, false, 2>), (conv_nt3hl_kernel<4, 4, 2, false, 1>), B * (H / 8) * (W / 16) * (Cout / 128), 256);
        else
            MU_DGH((conv_nt3hl_kernel<4, 2, 1, true, 2>), (conv_nt3hl_kernel<4, 2, 1, true, 1>), B * (H / 8) * (W / 16) * (Cout / 64), 256);
    } else if (Cout % 128 == 0) {
        MU_DGG(4, 4, 2, npb * (Cout / 128));
    } else if (Cout % 64 == 0) {
        MU_DGG(4, 2, 1, npb * (Cout / 64));
    } else {
        MU_DGG(2, 2, 1, npb * (Cout / 32));
    }
#undef MU_DGH
#undef MU_DGG
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// y = conv1x1(x) + addend (addend and y share the row stride y_ld; they may be the same buffer only if y == addend exactly).
// Joins the two gradients of Mask2FormerAttention's input -- the data-gradient of the q/k/v projection and the residual branch dY
// (`attention_output += x`, ade_semantic.py:187) -- inside the projection's epilogue instead of a separate elementwise pass.
// Only shapes served by the LDS-DMA 1x1 kernel (Cin * elem_size % 128 == 0, Cout % 64 == 0); MU_ERR_SHAPE otherwise.
extern "C" int mu_conv1x1_add_supported(int Cin, int Cout, int dtype) {
    const int es = dtype == MU_F16 ? 2 : 4;
    return (dtype == MU_F16 || dtype == MU_F32 || dtype == MU_F32X) && Cin > 0 && Cout > 0 && (Cin * es) % 128 == 0 && Cout % 64 == 0 ? 1 : 0;
}

template <typename T>
static void conv1x1_add_launch(const T* x, const T* w, const T* addend, T* y, long M, int Cin, int Cout, long x_ld, long y_ld, hipStream_t st) {
    const int npb = (int)((M + 127) / 128);
    if (Cout % 128 == 0)
        conv_nt2_kernel<T, 4, 4, 2, 1, false, true><<<npb * (Cout / 128), 256, 0, st>>>(x, w, nullptr, y, 1, 1, (int)M, Cin, Cout, x_ld, y_ld, addend);
    else
        conv_nt2_kernel<T, 4, 4, 1, 1, false, true><<<(int)((M + 255) / 256) * (Cout / 64), 256, 0, st>>>(x, w, nullptr, y, 1, 1, (int)M, Cin, Cout, x_ld, y_ld, addend);
}

// The q/k/v projection of the fp32x attention block with its output written DIRECTLY in the attention operand encoding ([8 fp16 hi |
// 8 fp16 lo] per eight channels, mu_split_encode_h form): saves the read + write pass over qkv [B, N, 3C] behind the projection.
extern "C" int mu_conv1x1_fwd_enc_h(const void* x, const void* w, const float* bias, void* y, long M, int Cin, int Cout, long x_ld, long y_ld,
                                    void* stream) {
    if (!x || !w || !y || M <= 0 || M > 0x7fffffffL) return MU_ERR_ARG;
    if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 64 || x_ld < Cin || y_ld < Cout || x_ld % 8 || y_ld % 8) return MU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const xf32* xs = (const xf32*)x;
    const xf32* ws = (const xf32*)w;
    const int npb = (int)((M + 127) / 128);
    if (Cout % 128 == 0)
        conv_nt2_kernel<xf32, 4, 4, 2, 1><<<npb * (Cout / 128), 256, 0, st>>>(xs, ws, bias, (xf32*)y, 1, 1, (int)M, Cin, Cout, x_ld, y_ld, nullptr, nullptr, MU_NT2_ENC_H);
    else
        conv_nt2_kernel<xf32, 4, 4, 1, 1><<<(int)((M + 255) / 256) * (Cout / 64), 256, 0, st>>>(xs, ws, bias, (xf32*)y, 1, 1, (int)M, Cin, Cout, x_ld, y_ld, nullptr, nullptr, MU_NT2_ENC_H);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_conv1x1_fwd_add(const void* x, const void* w, const void* addend, void* y, long M, int Cin, int Cout, long x_ld, long y_ld,
                                  int dtype, void* stream) {
    if (!x || !w || !addend || !y || M <= 0 || M > 0x7fffffffL) return MU_ERR_ARG;
    if (!mu_conv1x1_add_supported(Cin, Cout, dtype) || x_ld < Cin || y_ld < Cout || x_ld % 8 || y_ld % 8) return MU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F16) conv1x1_add_launch<h16>((const h16*)x, (const h16*)w, (const h16*)addend, (h16*)y, M, Cin, Cout, x_ld, y_ld, st);
    else if (dtype == MU_F32X) conv1x1_add_launch<xf32>((const xf32*)x, (const xf32*)w, (const xf32*)addend, (xf32*)y, M, Cin, Cout, x_ld, y_ld, st);
    else conv1x1_add_launch<float>((const float*)x, (const float*)w, (const float*)addend, (float*)y, M, Cin, Cout, x_ld, y_ld, st);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// weight gradient: dW[tap][co][ci] = sum_p dy[p][co] * x[p + shift(tap)][ci]
// "TN" GEMM: both operands have the reduction index (pixel) as their row index in memory, so
// fragments are fetched with transposing LDS reads (fp16: ds_read_b64_tr_b16; fp32: b32 columns).
// Split over pixel ranges; fp32 partial slabs + deterministic reduce (no float atomics).
// ------------------------------------------------------------------------------------------
template <typename T> struct WgTile;
template <> struct WgTile<h16> { static constexpr int PAD = 0; static constexpr int KP = 32; };
// fp32: row pad (elements) puts the g=0/1 pixel rows of a ds_read_b32 on different banks
template <> struct WgTile<float> { static constexpr int PAD = 16; static constexpr int KP = 16; };
template <> struct WgTile<xf32> { static constexpr int PAD = 16; static constexpr int KP = 32; };     // K = 32 per stage: v_mfma_f32_16x16x32_bf16 triples (1x1 layers)
#ifndef MU_WG_XSWAP
#define MU_WG_XSWAP 1
#endif

typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));

// BIAS (fp16): the bias gradient db[co] = sum_p dy[p][co] rides along as one more column tile of ones -- the dy fragments are in
// registers anyway, so the separate column-sum sweep over dy (402 MB for the N = 16384 q/k/v projection) disappears.
template <typename T, int TM, int TN, int WR, int TAPS, bool BIAS = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                                         int B, int H, int W, int Cin, int Cout, long x_ld, long dy_ld, int nsplit,
                                                         long pix_per_split, float* __restrict__ bias_part = nullptr) {
    constexpr int VN = Mma<T>::VN;
    constexpr int WC = 4 / WR;
    constexpr int BCO = WR * TM * 16, BCI = WC * TN * 16;
    constexpr int KP = WgTile<T>::KP;                        // pixels per LDS stage
    constexpr int SA = BCO + WgTile<T>::PAD, SB = BCI + WgTile<T>::PAD;   // LDS row strides (elements)
    constexpr int CA = BCO / VN, CB = BCI / VN;              // 16-byte chunks per row
    constexpr int NA = (KP * CA + 255) / 256, NB = (KP * CB + 255) / 256;

    __shared__ __attribute__((aligned(16))) T lds[2 * KP * (SA + SB)];
    T* As = lds;
    T* Bs = lds + 2 * KP * SA;

    const long Mtot = (long)B * H * W;
    const int nco = (Cout + BCO - 1) / BCO, nci = (Cin + BCI - 1) / BCI;
    int bid = blockIdx.x;
    const int split = bid % nsplit; bid /= nsplit;
    const int cib = bid % nci; bid /= nci;
    const int cob = bid % nco; bid /= nco;
    const int tap = bid;
    const int dh = TAPS == 9 ? tap / 3 - 1 : 0, dw = TAPS == 9 ? tap % 3 - 1 : 0;
    const int co0 = cob * BCO, ci0 = cib * BCI;
    const long p_begin = (long)split * pix_per_split;
    const long p_end = p_begin + pix_per_split < Mtot ? p_begin + pix_per_split : Mtot;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int r16 = lane & 15, g = lane >> 4;

    uint4 ra[NA], rb[NB];
    auto gload = [&](long pbase) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * 256, row = idx / CA, c = (idx % CA) * VN;
            const long p = pbase + row;
            if (idx < KP * CA && p < p_end && co0 + c < Cout) ra[i] = *reinterpret_cast<const uint4*>(dy + p * dy_ld + co0 + c);
            else ra[i] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * 256, row = idx / CB, c = (idx % CB) * VN;
            const long p = pbase + row;
            bool ok = idx < KP * CB && p < p_end && ci0 + c < Cin;
            if (ok && TAPS == 9) {
                const int ww = (int)(p % W) + dw, hh = (int)((p / W) % H) + dh;
                ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
            }
            if (ok) rb[i] = *reinterpret_cast<const uint4*>(x + (p + dh * W + dw) * x_ld + ci0 + c);
            else rb[i] = make_uint4(0, 0, 0, 0);
        }
    };
    // fp32x (MU_WG_XSWAP): the transposed reads below fetch ONLY the hi (or only the lo) halves of the 16-byte chunks, i.e. banks
    // = 0,1 (mod 4) -- a 32-lane group (pixel rows 8g + q, g in {0,1}) can then reach 32 of the 64 banks: 2-way conflicts whatever the
    // row padding.  Rows with bit 3 set are therefore stored with their halves swapped ([lo | hi]): the two row quartets of a lane group
    // use complementary banks.  Two 8-byte stores per chunk (as fast as one 16-byte store, MI355X_MICROARCH LDS table).
    constexpr bool XSWAP = std::is_same<T, xf32>::value && MU_WG_XSWAP;
    auto put = [&](T* dst, int row, const uint4& v) {
        if constexpr (XSWAP) {
            const int sw = ((row >> 3) & 1) * 8;
            char* d = reinterpret_cast<char*>(dst);
            *reinterpret_cast<uint2*>(d + sw) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2*>(d + 8 - sw) = make_uint2(v.z, v.w);
        } else {
            *reinterpret_cast<uint4*>(dst) = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * 256, row = idx / CA, c = (idx % CA) * VN;
            if (idx < KP * CA) put(As + (buf * KP + row) * SA + c, row, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * 256, row = idx / CB, c = (idx % CB) * VN;
            if (idx < KP * CB) put(Bs + (buf * KP + row) * SB + c, row, rb[i]);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accb[BIAS ? TM : 1];
#pragma unroll
    for (int i = 0; i < (BIAS ? TM : 1); ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = (int)((p_end - p_begin + KP - 1) / KP);
    if (nsteps > 0) {
        gload(p_begin);
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(p_begin + (long)(s + 1) * KP);
        const T* At = As + buf * KP * SA;
        const T* Bt = Bs + buf * KP * SB;
        if constexpr (sizeof(T) == 2) {
            // lane (16-lane group g, index li): q = li>>2 selects the pixel row of a 4x16 block,
            // pc = li&3 its 4-column piece; the transposed read returns, for column li, the 4 rows.
            const int q = r16 >> 2, pc = r16 & 3;
            h16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int col = (wr * TM + i) * 16 + 4 * pc;
                auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(At + (8 * g + q) * SA + col));
                auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(At + (8 * g + 4 + q) * SA + col));
                a[i] = (h16x8){(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = (wc * TN + j) * 16 + 4 * pc;
                auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(Bt + (8 * g + q) * SB + col));
                auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(Bt + (8 * g + 4 + q) * SB + col));
                b[j] = (h16x8){(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            if constexpr (BIAS) {
                if (wc == 0) {                               // the waves of one row group share their dy rows: one of them sums
                    const h16x8 ones = {(h16)1.f, (h16)1.f, (h16)1.f, (h16)1.f, (h16)1.f, (h16)1.f, (h16)1.f, (h16)1.f};
#pragma unroll
                    for (int i = 0; i < TM; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], ones, accb[i], 0, 0, 0);
                }
            }
        } else if constexpr (std::is_same<T, xf32>::value) {
            // fp32x, chunk-encoded tiles: the hi (bytes 0-7) and lo (bytes 8-15) halves of a chunk are four bf16 of four adjacent
            // channels, so ds_read_b64_tr_b16 transposes them exactly as in the fp16 branch: lane (g, q = r16 >> 2, pc = r16 & 3) points
            // at pixel rows 8g + q and 8g + 4 + q, channel chunk pc of its 16-channel tile and receives channel r16 of pixels
            // 8g..8g+7 -- the same k set on both operands.  Three v_mfma_f32_16x16x32_bf16 per tile pair, no VALU.
            static_assert(KP == 32, "one K = 32 MFMA triple per stage");
            const int q = r16 >> 2, pc = r16 & 3;
            auto frag = [&](const T* tile, int stride, int col) {
                const char* r0 = reinterpret_cast<const char*>(tile + (8 * g + q) * stride + col + 4 * pc);
                const char* r1 = reinterpret_cast<const char*>(tile + (8 * g + 4 + q) * stride + col + 4 * pc);
                const int sw = XSWAP ? (g & 1) * 8 : 0;      // rows 8g + q and 8g + 4 + q: bit 3 = g & 1 (see `put`)
                const uint2 h0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(r0 + sw)));
                const uint2 h1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(r1 + sw)));
                const uint2 l0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(r0 + 8 - sw)));
                const uint2 l1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(r1 + 8 - sw)));
                SplitF8 f;
                f.hi = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
                f.lo = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
                return f;
            };
            SplitF8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = frag(At, SA, (wr * TM + i) * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = frag(Bt, SB, (wc * TN + j) * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mu_mma_split(a[i], b[j], acc[i][j]);
        } else {
#pragma unroll
            for (int ks = 0; ks < KP / 4; ++ks) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = At[(4 * ks + g) * SA + (wr * TM + i) * 16 + r16];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bt[(4 * ks + g) * SB + (wc * TN + j) * 16 + r16];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }

    // partial slab [split][tap][Cout][Cin]; lane holds rows co = 4g+r, column ci = r16
    float* out = part + ((long)split * TAPS + tap) * Cout * Cin;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = ci0 + (wc * TN + j) * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wr * TM + i) * 16 + 4 * g + r;
                if (co < Cout && ci < Cin) out[(long)co * Cin + ci] = acc[i][j][r];
            }
        }
    if constexpr (BIAS) {
        if (wc == 0 && cib == 0 && r16 == 0) {               // every column of the ones tile holds the same sums: lane column 0 writes
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + (wr * TM + i) * 16 + 4 * g + r;
                    if (co < Cout) bias_part[(long)split * Cout + co] = accb[i][r];
                }
        }
    }
}

// db[co] = sum over the pixel ranges of the bias partials, in a fixed order: 16 channels per 256-thread block, 16 k-lanes per channel
// (lane kl sums ranges kl, kl+16, .. with four independent accumulators; the lanes are then combined through LDS in lane order).
// One thread per channel walking all <= 512 ranges serially cost 23 us per launch.
__global__ __launch_bounds__(256) void wgrad_bias_reduce_kernel(const float* __restrict__ bias_part, int nsplit, int Cout, int cout_valid,
                                                                float* __restrict__ db) {
    __shared__ float red[16][16];
    const int c = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int co = blockIdx.x * 16 + c;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (co < cout_valid) {
        int k = kl;
        for (; k + 48 < nsplit; k += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += bias_part[(long)(k + 16 * u) * Cout + co];
        }
        for (; k < nsplit; k += 16) a[0] += bias_part[(long)k * Cout + co];
    }
    red[kl][c] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (kl == 0 && co < cout_valid) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][c];
        db[co] = t;
    }
}


// ------------------------------------------------------------------------------------------
// weight gradient v2 (fp16, 3x3, W % 32 == 0): one block accumulates the THREE taps of a kernel row
// (dh fixed, dw = -1,0,+1).  A 32-pixel stage never crosses an image row, so the three shifted input
// tiles are 32-row windows (offset 0,1,2) of ONE 34-row LDS window: the dy tile and the x window are
// loaded once for 3x the MFMAs of v1 -> 3x fewer L2->LDS bytes per flop (v1 sits on the ~10 TB/s
// L2->LDS ceiling).  Boundary columns (w = 0 / W-1) only touch window rows 0 / 33, each used by a single
// tap, so they are zeroed at load time (zero page).  Tiles arrive by LDS-DMA, double-buffered; the LDS
// image XORs the 32-byte chunk index with a row hash so the transposed reads are conflict-free.
// ------------------------------------------------------------------------------------------
// 64-channel tiles (128-byte rows, four 32-byte chunks): a transposed read touches rows {a..a+3, a+8..a+11}; two rows share a
// 256-byte bank line, so the rows of equal parity -- a, a+2, a+8, a+10 -- must land on four different chunks: row bits 1 and 3.
// (row & 3 put rows a and a+8 on the same chunk: SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles of the 64 x 64 kernel.)
template <int BCH> __device__ __forceinline__ int wg_hash(int row) {
    return (BCH >= 128) ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
}
template <int N> __device__ __forceinline__ void wait_vmcnt_c() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The pixel stream comes straight from HBM (every block sweeps its own pixel range once), so the stage ring is NS deep with
// NS-1 stages in flight: counted s_waitcnt vmcnt + ONE raw s_barrier per 32-pixel stage (the former double buffer prefetched
// a single stage, ~0.2 us of MFMA work, far less than an HBM round trip).
#ifndef MU_WG_NS
#define MU_WG_NS 6
#endif
#ifndef MU_WG_SPS2
#define MU_WG_SPS2 1
#endif
#ifndef MU_WG_SPS2_64
#define MU_WG_SPS2_64 1
#endif
// ping-pong schedule of the weight-grad kernel (W % 64 == 0 layers, 128 x 128 tiles).  With the whole stage's DMAs in the first
// k-step's load section it was neutral to slower (that section ran 1.6x as long as the 24-MFMA section it hides behind); with the dy
// pieces in the first and the x pieces in the second load section: 128->128 @128^2 0.330 -> 0.307 ms, 256->256 @64^2 0.297 -> 0.277 ms.
// (Timing builds with real data: no DMA in the loop -23 %, the same DMA instructions on L1-resident data -4 % -- it is the
// ISSUE of the LDS-DMA instructions, not the bytes, that the lockstep schedule exposed.)
#ifndef MU_WG_PP
#define MU_WG_PP 1
#endif
// Lockstep schedule (non-ping-pong 8-wave kernels: W = 32 and W = 16 layers): the two waves of a SIMD issue their LDS-DMAs at different
// points of the stage -- waves 0-3 in front of the first k-step's MFMAs, waves 4-7 one k-step later (W = 16, one k-step per stage: behind
// their MFMAs) -- so that one wave's DMA issue overlaps the other's matrix section: 512->512 @32^2 0.307 -> 0.295 ms, @16^2 0.107 -> 0.105.
#ifndef MU_WG_STAGGER
#define MU_WG_STAGGER 2
#endif
#ifndef MU_WG_SPLIT_DMA
#define MU_WG_SPLIT_DMA 1
#endif
// SPS = 32-pixel k-steps per DMA stage.  SPS = 2 (W % 64 == 0): one barrier / DMA batch / ring step per 64 pixels -- the two
// waves of a SIMD run in lockstep behind the per-stage barrier, so the ~500 cycles of scalar + address work per ring step sit
// in front of both waves' MFMA bursts (PMC: SQ_ACTIVE_INST_SCA 18 % of wave cycles, MFMA pipe 44 % busy at SPS = 1).
// W16 = two-image-rows-per-stage mode: W == 16 with SPS = 1 (two 18-row windows) or W == 32 with SPS = 2 (two 34-row windows).
// PP = ping-pong schedule (8 waves, SPS = 2): the two wave groups (wr = 0 / 1, one wave of each per SIMD) run one section apart,
// a section being either a k-step's 24 MFMAs or its DMA issue + 20 transposed reads, so one group's matrix burst covers the
// other group's scalar / address / LDS work instead of both doing each in lockstep.
// (fp32x, round 6: the 3x3 weight gradient of that mode runs THIS fp16 kernel on the 16-bit view of its chunk-encoded input -- mu_conv_wgrad_h;
//  the bf16-pair instantiation of rounds 4-5 is gone.)
template <typename T, int TM, int TN, int WR, int NWV = 4, bool W16 = false, int SPS = 1, bool PP = false>
__global__ __launch_bounds__(NWV * 64, 1) void conv_wgrad3_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                                             int B, int H, int W, int Cin, int Cout, long x_ld, long dy_ld, int nsplit,
                                                             long pix_per_split) {
    constexpr int WC = NWV / WR;
    constexpr int BCO = WR * TM * 16, BCI = WC * TN * 16;
    // the dy tile (BCO wide) and the x window (BCI wide) may differ in width (128 x 64 tiles for the 64-channel layers): each has
    // its own row size, DMA lane mapping and swizzle hash
    static_assert(sizeof(T) == 2, "fp16 operands (the fp32x mode hands over 16-bit views)");
    constexpr int VN = 16 / (int)sizeof(T);                 // elements per 16-byte chunk
    constexpr int GS = 1;                                   // log2(chunks per 16-channel swizzle granule)
    constexpr int CPRA = BCO / VN, RPWA = 64 / CPRA, CPRB = BCI / VN, RPWB = 64 / CPRB;
    constexpr int KP = 32, SP = SPS * KP;                   // pixels per k-step (one MFMA K) and per DMA stage
    constexpr int RW = SP / 2;                              // two-row mode: image width
    constexpr int XR = ((W16 ? SP + 4 : SP + 2) + RPWB - 1) / RPWB * RPWB;   // x-window rows allocated (SP + 2, or 2 x (RW + 2))
    constexpr int NIA = SP / RPWA, NIB = XR / RPWB;        // DMA wave-instructions per tile
    constexpr int PADA = 0, PADB = 0;
    constexpr int STAGE = SP * BCO + PADA + XR * BCI + PADB;   // elements per stage

    static_assert(!PP || (SPS == 2 && NWV == 8 && WR == 2), "ping-pong needs two 4-wave groups and two k-steps per stage");
    constexpr int NS = SPS == 1 ? MU_WG_NS : 4;
    __shared__ __attribute__((aligned(16))) T lds[NS * STAGE];

    const long Mtot = (long)B * H * W;
    const int nco = Cout / BCO, nci = Cin / BCI;
    // The 3*nco*nci blocks that sweep the SAME pixel range (one per kernel row and channel-tile pair) get consecutive
    // logical ids inside one XCD, so they run side by side on that XCD and all but the first find the x / dy rows in its
    // L2: without this every block streams its operands from HBM (3.4-3.8 TB/s measured on every layer shape = the bound).
    int bid = xcd_remap(blockIdx.x, gridDim.x);
#ifdef MU_WG_NO_XCD
    bid = blockIdx.x;
#endif
    const int cib = bid % nci; bid /= nci;
    const int cob = bid % nco; bid /= nco;
    const int dh = bid % 3 - 1; bid /= 3;                   // kernel row: taps 3*(dh+1) + {0,1,2}
    const int split = bid;
    const int co0 = cob * BCO, ci0 = cib * BCI;
    const long p_begin = (long)split * pix_per_split;
    const long p_end = p_begin + pix_per_split < Mtot ? p_begin + pix_per_split : Mtot;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int r16 = lane & 15, g = lane >> 4;
    const int lrowA = lane / CPRA, c16A = lane % CPRA, lrowB = lane / CPRB, c16B = lane % CPRB;
    // DMA instructions this wave issues per stage (dy tile: i = wave, wave+NWV, ..; x window likewise)
    constexpr int NAW = (NIA + NWV - 1) / NWV, NBW = (NIB + NWV - 1) / NWV;
    const int n_w = (NIA - wave + NWV - 1) / NWV + (NIB - wave + NWV - 1) / NWV;

    // Per-lane source offsets are loop-invariant; the stage position (pixel, column, image row) is carried as scalars and
    // advanced by 32 pixels per stage -- the 64-bit div/mod of the flat pixel index used to cost more SALU time per stage
    // than the stage's MFMAs.
    int aoff[NAW], boff[NBW], bkind[NBW];      // bkind: 0 plain row, 1 window row 0 (needs w0 > 0), 2 row KP+1 (needs w0+KP < W), 3 unused
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
        const int row = (wave + k * NWV) * RPWA + lrowA;
        const int sc = (((c16A >> GS) ^ wg_hash<BCO>(row)) << GS) | (c16A & ((1 << GS) - 1));
        aoff[k] = row * (int)dy_ld + co0 + sc * VN;
    }
    // W == 16: a 32-pixel stage is two whole image rows; the window is two 18-row halves (columns -1 .. 16 of each image
    // row, the outer two always zero) instead of one 34-row run of the flat pixel index
    constexpr bool w16 = W16;
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int row = (wave + k * NWV) * RPWB + lrowB;    // window row: flat pixel pbase + dh*W - 1 + row
        const int sc = (((c16B >> GS) ^ wg_hash<BCI>(row)) << GS) | (c16B & ((1 << GS) - 1));
        if (w16) {
            const int half = row >= RW + 2, kk = row - half * (RW + 2);
            boff[k] = (half * RW + kk - 1) * (int)x_ld + ci0 + sc * VN;
            bkind[k] = (kk == 0 || kk == RW + 1 || row >= 2 * (RW + 2)) ? 3 : (half ? 5 : 4);     // 4 / 5: plain row of the first / second image row
        } else {
            boff[k] = (row - 1) * (int)x_ld + ci0 + sc * VN;
            bkind[k] = row == 0 ? 1 : (row == SP + 1 ? 2 : (row > SP + 1 ? 3 : 0));
        }
    }
    long pis = p_begin;                                     // next stage to issue
    int wi = (int)(p_begin % W), hi = (int)((p_begin / W) % H);
    const T* dyp = dy + p_begin * dy_ld;
    const T* xp = x + (p_begin + (long)dh * W) * x_ld;

    // (stage = stage_a + stage_b: the two-k-step kernels issue the dy pieces in front of the first k-step's MFMAs and the x pieces
    //  in front of the second's instead of all of them in one burst)
    auto stage_a = [&](int buf) {                // past p_end: an all-zero stage (keeps the per-wave DMA count uniform)
        T* At = lds + buf * STAGE;
        const bool live = pis < p_end;
#pragma unroll
        for (int k = 0; k < NAW; ++k) {
            const int i = wave + k * NWV;
            if (i < NIA) {
                const void* src = live ? (const void*)(dyp + aoff[k]) : (const void*)mu_zero_page;
                glds16a(src, At + i * RPWA * BCO);
            }
        }
    };
    auto stage_b = [&](int buf) {
        T* At = lds + buf * STAGE;
        T* Bt = At + SP * BCO + PADA;
        const bool live = pis < p_end;
        const int hh = hi + dh;
        const bool rowok = live && hh >= 0 && hh < H, rowok1 = live && hh + 1 >= 0 && hh + 1 < H;
        const bool lok = wi > 0, rok = wi + SP < W;
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            const int i = wave + k * NWV;
            if (i < NIB) {
                const bool ok = W16 ? ((rowok && bkind[k] == 4) || (rowok1 && bkind[k] == 5))
                                    : (rowok && (bkind[k] == 0 || (bkind[k] == 1 && lok) || (bkind[k] == 2 && rok)));
                const void* src = ok ? (const void*)(xp + boff[k]) : (const void*)mu_zero_page;
                glds16a(src, Bt + i * RPWB * BCI);
            }
        }
        pis += SP;
        dyp += SP * dy_ld;
        xp += SP * x_ld;
        if (w16) {
            hi += 2;
            if (hi >= H) hi -= H;
        } else {
            wi += SP;
            if (wi >= W) { wi = 0; hi = hi + 1 == H ? 0 : hi + 1; }
        }
    };
    auto stage = [&](int buf) { stage_a(buf); stage_b(buf); };

    f32x4 acc[3][TM][TN];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = (int)((p_end - p_begin + KP - 1) / KP);      // k-steps (32 pixels each)
    const int q = r16 >> 2, pc = r16 & 3;

    // Register double-buffered fragments: the transposed LDS reads of stage s+1 are issued before the MFMAs of stage s, so
    // the LDS latency (8 + 12 dependent-free ds_read_tr per 24 MFMAs) no longer sits between the MFMA groups.
    typedef h16x8 Frag;
    struct Frags { Frag a[TM]; Frag b[3][TN]; };
    const int wsh = (W16 && SPS == 1 && g >= 2) ? 2 : 0;     // W = 16: the second image row's window starts 18 rows in
    auto rd_tr = [&](const T* tile, int stride, int r0, int col, int hash0, int hash1) -> Frag {
        const T* p0 = tile + r0 * stride + ((((col >> 4) ^ hash0) << 4) | (col & 15));
        const T* p1 = tile + (r0 + 4) * stride + ((((col >> 4) ^ hash1) << 4) | (col & 15));
        auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(p0));
        auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(p1));
        return (h16x8){(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
    };
    auto load_frags = [&](int buf, int half, Frags& f) {     // k-step `half` of the stage in slot `buf`
        const T* At = lds + buf * STAGE;
        const T* Bt = At + SP * BCO + PADA;
        // absolute tile rows (the swizzle hash is a function of the row the DMA wrote): k-step `half` starts at dy row half*32
        // and at window row half*32 (flat) or half*34 (W = 32: one window per image row)
        const int ra = half * KP, rb = half * (W16 ? KP + 2 : KP);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int col = (wr * TM + i) * 16 + 4 * pc;    // 4 channels inside the 16-channel granule (col >> 4)
            const int r0 = ra + 8 * g + q;
            f.a[i] = rd_tr(At, BCO, r0, col, wg_hash<BCO>(r0), wg_hash<BCO>(r0 + 4));
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = (wc * TN + j) * 16 + 4 * pc;
                const int r0 = rb + 8 * g + q + t + wsh;
                f.b[t][j] = rd_tr(Bt, BCI, r0, col, wg_hash<BCI>(r0), wg_hash<BCI>(r0 + 4));
            }
        }
    };
    auto compute = [&](const Frags& f) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[i], f.b[t][j], acc[t][i][j], 0, 0, 0);
                }
    };

    // Ring protocol (one raw barrier per stage):
    //   top of step s : this wave's DMAs of stage s+1 have landed (vmcnt <= (NS-3) n_w); barrier -> everybody's have, and
    //                   everybody's reads of stage s-1 (issued in step s-2, consumed by the MFMAs of step s-1) are complete
    //   then          : DMA of stage s+NS-1 into the slot of stage s-1; fragment reads of stage s+1; MFMAs of stage s
    if constexpr (PP) {
        // Epochs: group A runs X(k) [DMA issue of stage S+2 when k opens stage S, then the k-step's fragment reads] at epoch 2k and
        // M(k) [its MFMAs, and when k opens a stage the wait for stage S+1] at 2k+1; group B one epoch later; one raw barrier per
        // epoch.  RAW: stage S+1 is first read in X(2S+2) (A: epoch 4S+4); both groups' waits sit in M(2S) (B: epoch 4S+2) and a
        // barrier follows.  WAR: stage S+2 overwrites the slot of stage S-2, last read in X(2S-3) (B: epoch 4S-5) -- four slots.
        stage(0);
        stage(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();
        Frags f;
        int buf = 0;
        auto kstep = [&](auto HALFC) {
            constexpr int HALF = decltype(HALFC)::value;
            // stage S+2 -> slot (S+2) & 3: the dy pieces in the first k-step's load section, the x pieces in the second's, so that
            // both sections are about as long as a 24-MFMA section (all DMAs in the first made it 1.6x as long)
            if (HALF == 0) stage_a(buf >= 2 ? buf - 2 : buf + 2);
            else stage_b(buf >= 2 ? buf - 2 : buf + 2);
            load_frags(buf, HALF, f);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            compute(f);
            __builtin_amdgcn_s_setprio(0);
            if (HALF == 0) {                                             // stage S+1 landed; the dy pieces of stage S+2 may stay in flight
                static_assert(NIA % NWV == 0, "every wave issues the same number of dy pieces");
                wait_vmcnt_c<NAW>();
            } else {
                buf = (buf + 1) & 3;
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        const int nstage = (nsteps + 1) / 2;
        for (int S = 0; S < nstage; ++S) {
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});            // (an odd tail k-step multiplies an all-zero half stage)
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
    } else {
#pragma unroll 1
    for (int k = 0; k < NS - 1; ++k) stage(k);
    auto wait_ring = [&]() {                                 // all but the newest NS-3 stages of this wave have landed
        constexpr int FULL = NAW + NBW;
        if (n_w == FULL) wait_vmcnt_c<(NS - 3) * FULL>();
        else if (n_w == FULL - 1) wait_vmcnt_c<(NS - 3) * (FULL - 1)>();
        else wait_vmcnt_c<(NS - 3) * (FULL > 2 ? FULL - 2 : 0)>();
    };
    wait_ring();
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    load_frags(0, 0, f0);
    int buf = 0;                                            // slot of the stage being multiplied
    // one k-step: (ring step if it opens a stage) -> MFMAs of `cur` -> fragment reads of the next k-step into `nxt`.
    // MFMAs first (in-process A/B: 2-3 % faster than reads first; hiding the reads from the compiler with inline asm and a
    // manual lgkmcnt(0) per step -- no waits between the MFMAs at all -- was 6 % SLOWER)
    auto kstep = [&](int s, auto HALFC, Frags& cur, Frags& nxt) {
        constexpr int HALF = decltype(HALFC)::value;
        if (HALF == 0) {
            wait_ring();
            __builtin_amdgcn_s_barrier();
#if MU_WG_STAGGER
            // half the waves (one of each SIMD's two) issue their whole share now, the other half one k-step later
            if (SPS == 2 && NWV == 8) { if (wr == 0) stage(buf == 0 ? NS - 1 : buf - 1); }
            else if (SPS == 1 && NWV == 8 && MU_WG_STAGGER == 2) { if (wr == 0) stage(buf == 0 ? NS - 1 : buf - 1); }
            else
#endif
            if (SPS == 2 && MU_WG_SPLIT_DMA) stage_a(buf == 0 ? NS - 1 : buf - 1);
            else stage(buf == 0 ? NS - 1 : buf - 1);
        } else if (SPS == 2 && MU_WG_STAGGER && NWV == 8) {
            if (wr == 1) stage(buf == 0 ? NS - 1 : buf - 1);
        } else if (SPS == 2 && MU_WG_SPLIT_DMA) {
            stage_b(buf == 0 ? NS - 1 : buf - 1);
        }
        compute(cur);
#if MU_WG_STAGGER == 2
        if (SPS == 1 && NWV == 8 && wr == 1) stage(buf == 0 ? NS - 1 : buf - 1);      // one-k-step stages: the second group issues behind its MFMAs
#endif
        if (s + 1 < nsteps) {
            if (HALF + 1 < SPS) {
                load_frags(buf, HALF + 1, nxt);
            } else {
                buf = buf + 1 == NS ? 0 : buf + 1;
                load_frags(buf, 0, nxt);
            }
        } else if (HALF + 1 == SPS) {
            buf = buf + 1 == NS ? 0 : buf + 1;
        }
    };
    for (int s = 0; s < nsteps; s += 2) {
        kstep(s, std::integral_constant<int, 0>{}, f0, f1);
        if (s + 1 < nsteps) kstep(s + 1, std::integral_constant<int, (SPS == 2 ? 1 : 0)>{}, f1, f0);
    }

    }

#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float* out = part + ((long)split * 9 + (dh + 1) * 3 + t) * Cout * Cin;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ci = ci0 + (wc * TN + j) * 16 + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + (wr * TM + i) * 16 + 4 * g + r;
                    out[(long)co * Cin + ci] = acc[t][i][j][r];
                }
            }
    }
}

// dst_oihw[o][i][t] = sum_split part[split][t][o][i]   (valid region only)
// KL k-lanes per output element: each lane sums every KL-th slab with 4 independent accumulators (16 loads in flight per
// element instead of one dependent chain); lanes are combined through LDS in a fixed order, so the result is deterministic.
// oscale (may be NULL): the result is multiplied by oscale[1] (mu_conv_wgrad_h1: the 1 / S of a scaled fp16 dy)
template <int KL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dst, int nsplit, int taps,
                                                           int Cout, int Cin, int O, int I, const float* __restrict__ oscale = nullptr) {
    constexpr int EPB = 256 / KL;                     // elements per block
    __shared__ float red[KL][EPB];
    const long n = (long)O * I * taps;
    const int e = threadIdx.x % EPB, kl = threadIdx.x / EPB;
    const long slab = (long)taps * Cout * Cin;
    for (long base = (long)blockIdx.x * EPB; base < n; base += (long)gridDim.x * EPB) {
        const long idx = base + e;
        // iterate in slab order (t, o, i) for coalesced reads
        const int i = idx % I;
        const int o = (idx / I) % O;
        const int t = idx / ((long)I * O);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (idx < n) {
            const float* p = part + ((long)t * Cout + o) * Cin + i;
            int k = kl;
            for (; k + 3 * KL < nsplit; k += 4 * KL) {
                s0 += p[(long)k * slab];
                s1 += p[(long)(k + KL) * slab];
                s2 += p[(long)(k + 2 * KL) * slab];
                s3 += p[(long)(k + 3 * KL) * slab];
            }
            for (; k < nsplit; k += KL) s0 += p[(long)k * slab];
        }
        float s = (s0 + s1) + (s2 + s3);
        if (KL > 1) {
            red[kl][e] = s;
            __syncthreads();
            if (kl == 0) {
#pragma unroll
                for (int j = 1; j < KL; ++j) s += red[j][e];
            }
        }
        if (kl == 0 && idx < n) dst[((long)o * I + i) * taps + t] = oscale ? s * oscale[1] : s;
        if (KL > 1) __syncthreads();
    }
}

// The same reduce for the two-term weight gradient of the fp32x 3x3 layers (mu_conv_wgrad_h): the slabs hold dW against the 16-bit VIEW of
// the chunk-encoded input -- column 8 (i / 4) + i % 4 is the hi half of input channel i, 4 columns further its lo half (common.h
// mu_ench4) -- so an output element is the sum of two slab columns, times the 1 / S of the scaled fp16 dy (oscale[1]).
template <int KL>
__global__ __launch_bounds__(256) void wgrad_reduce_pair_kernel(const float* __restrict__ part, float* __restrict__ dst, int nsplit, int taps,
                                                                int Cout, int Cin2, int O, int I, const float* __restrict__ oscale) {
    constexpr int EPB = 256 / KL;
    __shared__ float red[KL][EPB];
    const long n = (long)O * I * taps;
    const int e = threadIdx.x % EPB, kl = threadIdx.x / EPB;
    const long slab = (long)taps * Cout * Cin2;
    const float os = oscale[1];
    for (long base = (long)blockIdx.x * EPB; base < n; base += (long)gridDim.x * EPB) {
        const long idx = base + e;
        const int i = idx % I;
        const int o = (idx / I) % O;
        const int t = idx / ((long)I * O);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (idx < n) {
            const float* p = part + ((long)t * Cout + o) * Cin2 + 8 * (i >> 2) + (i & 3);
            int k = kl;
            for (; k + KL < nsplit; k += 2 * KL) {
                s0 += p[(long)k * slab];
                s1 += p[(long)k * slab + 4];
                s2 += p[(long)(k + KL) * slab];
                s3 += p[(long)(k + KL) * slab + 4];
            }
            for (; k < nsplit; k += KL) { s0 += p[(long)k * slab]; s1 += p[(long)k * slab + 4]; }
        }
        float s = (s1 + s3) + (s0 + s2);                  // lo columns, then hi columns
        if (KL > 1) {
            red[kl][e] = s;
            __syncthreads();
            if (kl == 0) {
#pragma unroll
                for (int j = 1; j < KL; ++j) s += red[j][e];
            }
        }
        if (kl == 0 && idx < n) dst[((long)o * I + i) * taps + t] = s * os;
        if (KL > 1) __syncthreads();
    }
}

#ifndef MU_WG1_WIDE
#define MU_WG1_WIDE 1           // fp16 1x1 layers: tiles that span all (or 192) output channels, both operands read once
#endif
#ifndef MU_WG1_SPLITS
#define MU_WG1_SPLITS 512       // pixel ranges per wide tile: these launches are HBM streams and want ~2 blocks per CU in flight
#endif
static inline bool wgrad_is_wide(int bco) { return bco == 192 || bco == 160; }

static inline void wgrad_plan(long M, int Cin, int Cout, int taps, int bco, int bci, int* nsplit, long* pps) {
    long tiles = (long)taps * ((Cout + bco - 1) / bco) * ((Cin + bci - 1) / bci);
    long want = 2048 / tiles;
    if (want < 1) want = 1;
    long max_split = (M + 255) / 256;           // at least 256 pixels per split
    if (want > max_split) want = max_split;
    const long cap = wgrad_is_wide(bco) ? MU_WG1_SPLITS / tiles > 64 ? MU_WG1_SPLITS / tiles : 64 : 256;
    if (want > cap) want = cap;
    long p = (M + want - 1) / want;
    p = (p + 31) / 32 * 32;                    // multiple of both stage depths (32 / 16)
    *pps = p;
    *nsplit = (int)((M + p - 1) / p);
}

static inline void wgrad_tile(int Cin, int Cout, int* bco, int* bci, int taps = 9, bool fp16 = false) {
    *bco = (Cout % 128 == 0) ? 128 : (Cout % 64 == 0 ? 64 : 32);
    *bci = (Cin % 128 == 0) ? 128 : (Cin % 64 == 0 ? 64 : 32);
    if (MU_WG1_WIDE && fp16 && taps == 1 && *bci >= 64 && (Cout % 192 == 0 || Cout == 160)) {
        // q/k/v projection (Cout = 3C) and the 150 -> 160 class head: with square tiles the narrow operand was re-read per
        // output-channel tile (64 -> 192: 804 MB instead of 537; 64 -> 160 with 32x32 tiles: 1.34 GB instead of 470 MB)
        *bco = Cout == 160 ? 160 : 192;
        return;
    }
    // supported tile pairs: 128x128, 64x64, 32x32 (+ mixed via the smaller square)
    int m = *bco < *bci ? *bco : *bci;
    *bco = m; *bci = m;
}

// v2 (3 taps per block) applies to fp16 3x3 layers with W % 32 == 0 (or W == 16) and 64/128-wide channel tiles.
// Tile choice: 128x128 (8 waves) or 64x64 (4 waves).  The kernel also runs 128x64 / 64x128 tiles (MU_WG_MIXED=1: layers with one
// 64-channel side, dy or x tile twice as wide), parity-clean but measured SLOWER in-process (64->128 @128^2: 333 vs 252 us,
// 128->64: 356 vs 250 us): at 4 waves they need ~256 VGPRs and only the 32-pixel stages fit twice into the LDS.
#ifndef MU_WG_MIXED
#define MU_WG_MIXED 0
#endif
// Blocks per launch = exactly what is resident at once (8-wave tiles: one per CU; 4-wave tiles: two per CU): every block does the
// same work, so one whole round has no tail, and the fp32 slab traffic (write + reduce: 2.4 GB/step at 512 / 1536 blocks, the reduce
// kernel at HBM rate) shrinks with the split count.  In-process A/B: 128->128 @128^2 341 -> 324 us, 512->512 @16^2 123 -> 105 us,
// 64->64 @128^2 147 -> 125 us; 768 blocks (1.5 rounds) for the 4-wave tiles is 10 % WORSE than 512.
#ifndef MU_WG_BLOCKS128
#define MU_WG_BLOCKS128 256
#endif
#ifndef MU_WG_BLOCKS64
#define MU_WG_BLOCKS64 512
#endif
static inline bool wgrad3_choose(int H, int W, int Cin, int Cout, int taps, int dtype, int* tco, int* tci) {
    if (dtype != MU_F16 || taps != 9 || !(W % 32 == 0 || (W == 16 && H % 2 == 0))) return false;
    const int a = Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 0), b = Cin % 128 == 0 ? 128 : (Cin % 64 == 0 ? 64 : 0);
    if (!a || !b) return false;
    if (a == b || MU_WG_MIXED) { *tco = a; *tci = b; }
    else { *tco = 64; *tci = 64; }
    return true;
}
static inline void wgrad3_plan(long M, int Cin, int Cout, int tco, int tci, int* nsplit, long* pps) {
    long tiles = 3L * (Cout / tco) * (Cin / tci);
    long want = ((tco == 128 && tci == 128) ? MU_WG_BLOCKS128 : MU_WG_BLOCKS64) / tiles;
    if (want < 1) want = 1;
    long max_split = (M + 511) / 512;
    if (want > max_split) want = max_split;
    if (want > 256) want = 256;
    long p = (M + want - 1) / want;
    p = (p + 63) / 64 * 64;                     // whole 64-pixel stages (the two-k-step kernel); M is a multiple of 32
    *pps = p;
    *nsplit = (int)((M + p - 1) / p);
}

// ------------------------------------------------------------------------------------------
// weight gradient of the first layer (<= 4 valid input channels, 3x3): 64 x 3 x 9 outputs, each a dot product over every
// pixel -- 3.6 GFLOP but 200 MB of dy/x.  The MFMA tile kernels spend nine 32-channel-padded tap passes on it (0.5 ms);
// this one streams dy once with plain FMAs: a block walks image rows, keeps the three input rows around the current one
// in LDS as float4 (zero ring), lane = (pixel sub-index, 4 output channels), 27 x 4 accumulators per lane.
// Partial slab layout [block][tap][Cout][4] feeds the same deterministic reduce as the other kernels.
// ------------------------------------------------------------------------------------------
#define MU_RGB_MAXW 256
#ifndef MU_RGB2
#define MU_RGB2 1              // fp16, W % 32 == 0: the matrix-core form below (0 = the plain-FMA kernel everywhere)
#endif
// Round 3: the sweep was latency-bound (157 us against ~40 us of HBM time for the 200 MB of dy / x): every 4-pixel step loaded its dy
// values and then ran 108 dependent FMAs, with two block barriers per image row around a scalar-load staging of the three input
// rows.  Now a lane fetches the dy values of a whole 128-pixel chunk (8 steps) one chunk AHEAD of the FMAs that consume them, the
// next image row's input pixels (one 8-byte load each) are in flight during the current row's arithmetic and land in the other
// half of a double-buffered LDS image -- one barrier per image row.  Two blocks per CU (251 registers; at one wave per SIMD a lone wave
// issues an FMA every 4 cycles instead of 2).  In-process A/B at B = 64: 141.8 -> 111.2 us, bit-identical sums.
template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_rgb_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                                        int B, int H, int W, int Cout, int cin_valid, long x_ld, long dy_ld) {
    constexpr int KU = 8;                                       // pixel steps per chunk: 8 x 16 pixels
    constexpr int NX = (3 * (MU_RGB_MAXW + 2) + 255) / 256;     // input pixels a thread stages per image row
    using DV = typename std::conditional<sizeof(T) == 2, h16x4, float4>::type;
    __shared__ float4 xs[2][3][MU_RGB_MAXW + 2];
    __shared__ float red[4][27][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pi = lane >> 4, cg = lane & 15;                 // pixel sub-index (4 per wave step), 4-channel group
    const int co0 = blockIdx.y * 64;
    float acc[27][4];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;

    const int rows = B * H;
    const int nchunk = (W + 16 * KU - 1) / (16 * KU);
    const int nstage = 3 * (W + 2);

    auto load_dy = [&](int r, int chunk, DV (&d)[KU]) {
        const T* dyr = dy + ((long)r * W) * dy_ld + co0 + cg * 4;
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            const int p = chunk * (16 * KU) + 16 * k + wave * 4 + pi;
            if (p < W) d[k] = *reinterpret_cast<const DV*>(dyr + (long)p * dy_ld);
            else d[k] = DV{};
        }
    };
    auto load_x = [&](int r, DV (&xr)[NX]) {                    // 4 stored channels of each staged input pixel (channels >= cin_valid are zero padding)
        const int b = r / H, h = r - b * H;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + j * 256;
            xr[j] = DV{};
            if (i < nstage) {
                const int dh = i / (W + 2), wc = i - dh * (W + 2);
                const int hh = h + dh - 1, ww = wc - 1;
                if (hh >= 0 && hh < H && ww >= 0 && ww < W) xr[j] = *reinterpret_cast<const DV*>(x + (((long)b * H + hh) * W + ww) * x_ld);
            }
        }
    };
    auto store_x = [&](int buf, const DV (&xr)[NX]) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + j * 256;
            if (i < nstage) {
                const int dh = i / (W + 2), wc = i - dh * (W + 2);
                float4 v;
                if constexpr (sizeof(T) == 2) v = make_float4((float)xr[j][0], (float)xr[j][1], (float)xr[j][2], (float)xr[j][3]);
                else v = make_float4(xr[j].x, xr[j].y, xr[j].z, xr[j].w);
                if (cin_valid < 4) v.w = 0.f;
                if (cin_valid < 3) v.z = 0.f;
                if (cin_valid < 2) v.y = 0.f;
                xs[buf][dh][wc] = v;
            }
        }
    };

    int r = blockIdx.x, buf = 0;
    DV xr[NX], dcur[KU], dnxt[KU];
    if (r < rows) {
        load_x(r, xr);
        load_dy(r, 0, dcur);
    }
    while (r < rows) {
        store_x(buf, xr);
        __syncthreads();                                       // this row's input image is complete; the other buffer's readers are done
        const int rn = r + gridDim.x;
        if (rn < rows) load_x(rn, xr);
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            if (chunk + 1 < nchunk) load_dy(r, chunk + 1, dnxt);
            else if (rn < rows) load_dy(rn, 0, dnxt);
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int p = chunk * (16 * KU) + 16 * k + wave * 4 + pi;
                const int pc = p < W ? p : 0;                    // (d is zero there)
                float d[4];
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        d[c] = (float)dcur[k][c];
                        asm volatile("" : "+v"(d[c]));          // one conversion per value: keeps the 27 FMAs on it plain v_fma_f32 (not v_fma_mix)
                    }
                } else {
                    d[0] = dcur[k].x; d[1] = dcur[k].y; d[2] = dcur[k].z; d[3] = dcur[k].w;
                }
#pragma unroll
                for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const float4 xv = xs[buf][dh][pc + dw];
                        const int t = (dh * 3 + dw) * 3;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            acc[t + 0][c] = fmaf(d[c], xv.x, acc[t + 0][c]);
                            acc[t + 1][c] = fmaf(d[c], xv.y, acc[t + 1][c]);
                            acc[t + 2][c] = fmaf(d[c], xv.z, acc[t + 2][c]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);               // keep the nine LDS reads of a step next to its FMAs (the scheduler otherwise
            }                                                    // hoists all 72 of a chunk: 400 registers, spills)
#pragma unroll
            for (int k = 0; k < KU; ++k) dcur[k] = dnxt[k];
        }
        r = rn;
        buf ^= 1;
    }
    // (tap, ci) x 4 co per lane: fold the four pixel sub-indices (lanes 16 apart), then the four waves, in a fixed order
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = acc[t][c];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            acc[t][c] = v;
        }
    if (pi == 0) {
#pragma unroll
        for (int t = 0; t < 27; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[wave][t][cg * 4 + c] = acc[t][c];
    }
    __syncthreads();
    // part[blk][tap][Cout][4]: ci in the last index (entries >= cin_valid are never read by the reduce)
    for (int i = tid; i < 27 * 64; i += 256) {
        const int t = i / 64, co = i - t * 64;
        const float v = (red[0][t][co] + red[1][t][co]) + (red[2][t][co] + red[3][t][co]);
        const int tap = t / 3, ci = t - tap * 3;
        part[(((long)blockIdx.x * 9 + tap) * Cout + co0 + co) * 4 + ci] = v;
    }
}
#define MU_RGB_MAXBLK 1024

// ------------------------------------------------------------------------------------------
// The same weight gradient on the matrix cores (fp16 storage, W % 32 == 0; round 5).  dW[co][(tap, ci)] = sum_p dy[p][co] x[p + tap][ci] is a
// GEMM with K = pixels, M = 64 output channels and N = 9 taps x 4 stored input channels = 36 columns (three 16-column tiles, the last
// one three quarters empty).  Both operands are K-major in memory, so both fragments come from transposing LDS reads
// (ds_read_b64_tr_b16: each lane hands in the address of ONE 8-byte piece, a 16-lane group gets the 4 x 16 block of pieces
// transposed) -- and since the piece addresses are free, the im2col matrix is never built: piece (pixel, tap) IS the 8-byte pixel
// x[pixel + tap] of the staged input rows.  A block owns a band of consecutive image rows: the dy row (16 KB at W = 128) and ONE new
// input row per step arrive through registers (loaded a row ahead) into a double-buffered dy image / a 4-slot ring of input rows, so
// dy and x are read from HBM once (the FMA kernel above re-reads every input row three times and issues 1728 FMAs per pixel: 111-132 us
// at B = 64; this one: ~12 MFMAs per 32 pixels).  Wave w takes the k-steps w, w + 4, ..; its 12 accumulator tiles are folded over the
// four waves in a fixed order at the end.  Slab layout [block][tap][Cout][4] as above (same deterministic reduce).
// ------------------------------------------------------------------------------------------
#define MU_RGB2_DYS 80         // halves per staged dy pixel row: 64 channels + 16 pad (160 bytes: the four pixel rows of a transposed read on disjoint banks)
template <int NPT>             // NPT = W / 32: 16-byte dy pieces per thread and row (and 32-pixel k-steps per row)
__global__ __launch_bounds__(256, 3) void wgrad_rgb2_kernel(const h16* __restrict__ x, const h16* __restrict__ dy, float* __restrict__ part,
                                                         int B, int H, int W, int Cout, long x_ld, long dy_ld, int rows_per_blk) {
    extern __shared__ __attribute__((aligned(16))) h16 rgb2_lds[];
    // [2][W][80] dy images | [4][W + 2][4] input-row ring | [W + 2][4] zeros
    h16* dys = rgb2_lds;
    h16* xs = dys + 2 * W * MU_RGB2_DYS;
    h16* xz = xs + 4 * (W + 2) * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, q = r16 >> 2, pc = r16 & 3;
    const int co0 = blockIdx.y * 64;
    const int rows = B * H;
    const int rbeg = blockIdx.x * rows_per_blk, rend = rbeg + rows_per_blk < rows ? rbeg + rows_per_blk : rows;
    constexpr int nks = NPT;                                  // 32-pixel k-steps per image row (W = 32 NPT: W * 8 = 256 NPT pieces per dy row)
    const int xw = W + 2;

    f32x4 acc[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[a][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < xw; i += 256) *reinterpret_cast<uint2*>(xz + i * 4) = make_uint2(0u, 0u);

    // staging through registers: NPT dy pieces and <= 2 input pixels per thread and row.  (Named scalars, not arrays: hipcc left a
    // `uint4 dreg[NPT]` touched from unrolled loops in scratch memory -- 16 NPT + 16 bytes per lane, every piece through a scratch
    // store and load.)
    uint4 d0, d1, d2, d3, d4, d5, d6, d7;
    uint2 x0, x1;
#define RGB2_EACH(OP) do { OP(0, d0); if constexpr (NPT > 1) OP(1, d1); if constexpr (NPT > 2) OP(2, d2); if constexpr (NPT > 3) OP(3, d3); \
                           if constexpr (NPT > 4) OP(4, d4); if constexpr (NPT > 5) OP(5, d5); if constexpr (NPT > 6) OP(6, d6); if constexpr (NPT > 7) OP(7, d7); } while (0)
#define RGB2_LD(K, D) D = *reinterpret_cast<const uint4*>(dsrc + (long)((tid + K * 256) >> 3) * dy_ld + ((tid + K * 256) & 7) * 8)
#define RGB2_ST(K, D) *reinterpret_cast<uint4*>(ddst + ((tid + K * 256) >> 3) * MU_RGB2_DYS + ((tid + K * 256) & 7) * 8) = D
#define RGB2_LOAD_DY(r_) do { const h16* dsrc = dy + (long)(r_) * W * dy_ld + co0; RGB2_EACH(RGB2_LD); } while (0)
#define RGB2_STORE_DY(buf_) do { h16* ddst = dys + (buf_) * W * MU_RGB2_DYS; RGB2_EACH(RGB2_ST); } while (0)
    // input row `xr` (a global row index b * H + h; the rows a tap must not see -- outside the image -- are replaced by the zero row below)
#define RGB2_LOAD_X(xr_) do {                                                                                                   \
        const int xr = (xr_);                                                                                                    \
        const bool ok = xr >= 0 && xr < rows;                                                                                    \
        const h16* xsrc = x + (long)(ok ? xr : 0) * W * x_ld;                                                                    \
        x0 = make_uint2(0u, 0u); x1 = make_uint2(0u, 0u);                                                                        \
        if (ok && tid >= 1 && tid <= W) x0 = *reinterpret_cast<const uint2*>(xsrc + (long)(tid - 1) * x_ld);                     \
        if (ok && tid + 256 <= W) x1 = *reinterpret_cast<const uint2*>(xsrc + (long)(tid + 255) * x_ld);                         \
    } while (0)
#define RGB2_STORE_X(xr_) do {                                                                                                  \
        h16* xdst = xs + ((xr_) & 3) * xw * 4;                                                                                   \
        if (tid < xw) *reinterpret_cast<uint2*>(xdst + tid * 4) = x0;                                                            \
        if (tid + 256 < xw) *reinterpret_cast<uint2*>(xdst + (tid + 256) * 4) = x1;                                              \
    } while (0)

    // this lane's three pieces (column tiles n = 0 .. 2): tap 4 n + pc = (dh, dw); taps 9 .. 11 do not exist (zero row)
    int pdh[3], pdw[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const int tap = n * 4 + pc;
        pdh[n] = tap < 9 ? tap / 3 : 3;
        pdw[n] = tap < 9 ? tap - (tap / 3) * 3 : 0;
    }

    if (rbeg < rend) {
        // prologue: input rows rbeg - 1 and rbeg into the ring, dy row rbeg and input row rbeg + 1 into registers
        RGB2_LOAD_X(rbeg - 1); RGB2_STORE_X(rbeg - 1);
        RGB2_LOAD_X(rbeg); RGB2_STORE_X(rbeg);
        RGB2_LOAD_X(rbeg + 1);
        RGB2_LOAD_DY(rbeg);
    }
    int buf = 0;
    for (int r = rbeg; r < rend; ++r) {
        RGB2_STORE_DY(buf);
        RGB2_STORE_X(r + 1);
        __syncthreads();                                       // row r's dy image and input rows r - 1 .. r + 1 are complete
        if (r + 1 < rend) { RGB2_LOAD_DY(r + 1); RGB2_LOAD_X(r + 2); }    // in flight during this row's arithmetic
        const int h = r % H;
        // the three input rows of this image row (ring slots by global row index; rows outside the image: the zero row)
        const h16* rowp0 = h > 0 ? xs + ((r - 1) & 3) * xw * 4 : xz;
        const h16* rowp1 = xs + (r & 3) * xw * 4;
        const h16* rowp2 = h + 1 < H ? xs + ((r + 1) & 3) * xw * 4 : xz;
        const h16* pbase[3];
#pragma unroll
        for (int n = 0; n < 3; ++n)
            pbase[n] = (pdh[n] == 0 ? rowp0 : (pdh[n] == 1 ? rowp1 : (pdh[n] == 2 ? rowp2 : xz))) + pdw[n] * 4 + (4 * g + q) * 4;
        const h16* dyt = dys + buf * W * MU_RGB2_DYS;
        for (int ks = wave; ks < nks; ks += 4) {
            const int p0 = ks * 32;
            // k-slot j of lane group g: pixel p0 + 4 g + j (j < 4), p0 + 16 + 4 g + (j - 4) -- the same for both operands
            h16x8 bf[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const h16* a0 = pbase[n] + p0 * 4;
                auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(a0));
                auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(a0 + 16 * 4));
                bf[n] = (h16x8){(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const h16* a0 = dyt + (p0 + 4 * g + q) * MU_RGB2_DYS + a * 16 + 4 * pc;
                auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(a0));
                auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(a0 + 16 * MU_RGB2_DYS));
                const h16x8 af = {(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
#pragma unroll
                for (int n = 0; n < 3; ++n) acc[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[n], acc[a][n], 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    // fold the four waves in a fixed order, one column tile at a time through a [4 waves][64 co][16] float buffer (16 KB over the dy
    // images; the launch sizes the LDS for both): accumulator tile (a, n) holds D[co = 16 a + 4 g + i][column 16 n + r16]
    float* red = reinterpret_cast<float*>(rgb2_lds);
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        __syncthreads();                                       // the last row's readers / the previous tile's readers are done
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(wave * 64 + a * 16 + 4 * g + i) * 16 + r16] = acc[a][n][i];
        __syncthreads();
        for (int e = tid; e < 64 * 16; e += 256) {
            const int co = e >> 4, col = e & 15;
            const float v = (red[(0 * 64 + co) * 16 + col] + red[(1 * 64 + co) * 16 + col]) + (red[(2 * 64 + co) * 16 + col] + red[(3 * 64 + co) * 16 + col]);
            const int tap = n * 4 + (col >> 2), ci = col & 3;
            if (tap < 9) part[(((long)blockIdx.x * 9 + tap) * Cout + co0 + co) * 4 + ci] = v;
        }
    }
}
#undef RGB2_EACH
#undef RGB2_LD
#undef RGB2_ST
#undef RGB2_LOAD_DY
#undef RGB2_STORE_DY
#undef RGB2_LOAD_X
#undef RGB2_STORE_X

// ------------------------------------------------------------------------------------------
// Nine taps per block for the 64-channel-wide 3x3 layers (fp16, W % 32 == 0, W <= 128; round 5).  The three-taps-per-block ring kernel
// above gives a 64 x 64 channel tile 48 MFMAs per 32-pixel stage behind a full set of LDS-DMAs and a block barrier (MFMA pipe 24 % busy
// on these layers).  Here a block owns a band of consecutive image rows of ONE 64 x 64 channel tile and all nine taps: the dy row and a
// 4-slot ring of input rows sit in LDS as plain padded pixel rows (160 bytes: the eight pixel rows a 32-lane half of a transposing read
// touches on disjoint banks -- 144 bytes measured 44 % of the LDS cycles as conflicts, 160: 5 %, at equal time), both MFMA operands come
// from ds_read_b64_tr_b16, and a tap is nothing but a pixel offset into the ring (rows outside the
// image: a zero row).  Wave w owns the 16 input channels 16 w .. 16 w + 15 for all taps and all 64 output channels: 8 + 18 fragment
// reads and 36 MFMAs per 32 pixels, 144 accumulators, no cross-wave fold; 145 KB of LDS at W = 128.  dy and x are read from HBM once per channel tile.
// Slab layout [band][tap][Cout][Cin] = the ring kernel's, same deterministic reduce.
// ------------------------------------------------------------------------------------------
#ifndef MU_WG9
#define MU_WG9 1
#endif
#ifndef MU_WG9_S
#define MU_WG9_S 80            // halves per staged pixel row: 64 channels + 16 pad = 160 bytes = 32 x 5 -- the 8 pixel rows a 32-lane half of a transposing
#endif                         // read touches (4 pieces of 8 bytes each) then start on 8 distinct multiples of 32 bytes modulo the 256-byte bank row

#ifndef MU_WG9_BLOCKS
#define MU_WG9_BLOCKS 256
#endif
#ifndef MU_WG9_BLOCKS_W64
#define MU_WG9_BLOCKS_W64 256  // (W <= 64: 66 KB of LDS, two blocks fit a CU)
#endif
#ifndef MU_WG9_MAXC
#define MU_WG9_MAXC 128
#endif
#ifndef MU_WG9_MAXC32
#define MU_WG9_MAXC32 128
#endif
#ifndef MU_WG9_NEED64
#define MU_WG9_NEED64 1
#endif
template <int NPT>             // NPT = W / 32
__global__ __launch_bounds__(256, 1) void conv_wgrad9_kernel(const h16* __restrict__ x, const h16* __restrict__ dy, float* __restrict__ part,
                                                          int B, int H, int W, int Cin, int Cout, long x_ld, long dy_ld, int rows_per_blk) {
    constexpr int WC = 32 * NPT;                               // == W (the launch picks NPT = W / 32)
    __shared__ __attribute__((aligned(16))) h16 wg9_lds[(2 * WC + 5 * (WC + 2)) * MU_WG9_S];      // 145 KB at W = 128
    const int xw = W + 2;
    h16* dys = wg9_lds;                                        // [2][W][72]
    h16* xs = dys + 2 * W * MU_WG9_S;                          // [4][W + 2][72] ring (columns 0 and W + 1 stay zero) | [W + 2][72] zeros
    h16* xz = xs + 4 * xw * MU_WG9_S;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, q = r16 >> 2, pc = r16 & 3;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const int rows = B * H;
    const int rbeg = blockIdx.x * rows_per_blk, rend = rbeg + rows_per_blk < rows ? rbeg + rows_per_blk : rows;

    f32x4 acc[4][9];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[a][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // zero the ring (its border columns are never written again) and the zero row
    for (int i = tid; i < 5 * xw * MU_WG9_S / 8; i += 256) *reinterpret_cast<uint4*>(xs + i * 8) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    // staging through registers (named scalars: see wgrad_rgb2_kernel): NPT 16-byte pieces of the dy row and of the input row per thread
    uint4 d0, d1, d2, d3, e0, e1, e2, e3;
#define WG9_EACH(OP, P) do { OP(0, P##0); if constexpr (NPT > 1) OP(1, P##1); if constexpr (NPT > 2) OP(2, P##2); if constexpr (NPT > 3) OP(3, P##3); } while (0)
#define WG9_LD(K, D) D = *reinterpret_cast<const uint4*>(gsrc + (long)((tid + K * 256) >> 3) * gld + ((tid + K * 256) & 7) * 8)
#define WG9_ST(K, D) *reinterpret_cast<uint4*>(ldst + ((tid + K * 256) >> 3) * MU_WG9_S + ((tid + K * 256) & 7) * 8) = D
#define WG9_LOAD_DY(r_) do { const h16* gsrc = dy + (long)(r_) * W * dy_ld + co0; const long gld = dy_ld; WG9_EACH(WG9_LD, d); } while (0)
#define WG9_STORE_DY(buf_) do { h16* ldst = dys + (buf_) * W * MU_WG9_S; WG9_EACH(WG9_ST, d); } while (0)
    // input row xr_ (global row index; rows < 0 / >= rows are never used: the taps that would see them read the zero row)
#define WG9_LOAD_X(xr_) do { const int xr = (xr_); const h16* gsrc = x + (long)(xr >= 0 && xr < rows ? xr : 0) * W * x_ld + ci0; const long gld = x_ld; WG9_EACH(WG9_LD, e); } while (0)
#define WG9_STORE_X(xr_) do { h16* ldst = xs + (((xr_) & 3) * xw + 1) * MU_WG9_S; WG9_EACH(WG9_ST, e); } while (0)

    if (rbeg < rend) {
        WG9_LOAD_X(rbeg - 1); WG9_STORE_X(rbeg - 1);
        WG9_LOAD_X(rbeg); WG9_STORE_X(rbeg);
        WG9_LOAD_X(rbeg + 1);
        WG9_LOAD_DY(rbeg);
    }
    int buf = 0;
    for (int r = rbeg; r < rend; ++r) {
        WG9_STORE_DY(buf);
        WG9_STORE_X(r + 1);
        __syncthreads();                                       // row r's dy image and input rows r - 1 .. r + 1 are complete
        if (r + 1 < rend) { WG9_LOAD_DY(r + 1); WG9_LOAD_X(r + 2); }      // in flight during this row's arithmetic
        const int h = r % H;
        const h16* rowp[3];
        rowp[0] = (h > 0 ? xs + ((r - 1) & 3) * xw * MU_WG9_S : xz) + (4 * g + q) * MU_WG9_S + wave * 16 + 4 * pc;
        rowp[1] = xs + (r & 3) * xw * MU_WG9_S + (4 * g + q) * MU_WG9_S + wave * 16 + 4 * pc;
        rowp[2] = (h + 1 < H ? xs + ((r + 1) & 3) * xw * MU_WG9_S : xz) + (4 * g + q) * MU_WG9_S + wave * 16 + 4 * pc;
        const h16* dyt = dys + buf * W * MU_WG9_S + (4 * g + q) * MU_WG9_S + 4 * pc;
#pragma unroll
        for (int ks = 0; ks < NPT; ++ks) {
            const int p0 = ks * 32;
            // k-slot j of lane group g: pixel p0 + 4 g + j (j < 4), p0 + 16 + 4 g + (j - 4) -- the same for both operands
            h16x8 af[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const h16* a0 = dyt + p0 * MU_WG9_S + a * 16;
                auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(a0));
                auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(a0 + 16 * MU_WG9_S));
                af[a] = (h16x8){(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const h16* b0 = rowp[t / 3] + (p0 + t % 3) * MU_WG9_S;          // window column = image column + 1 + (kw - 1)
                auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(b0));
                auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(b0 + 16 * MU_WG9_S));
                const h16x8 bf = {(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a], bf, acc[a][t], 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    // accumulator tile (a, t) holds D[co = 16 a + 4 g + i][ci = 16 wave + r16] of tap t
    float* slab = part + (long)blockIdx.x * 9 * Cout * Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                slab[((long)t * Cout + co0 + 16 * a + 4 * g + i) * Cin + ci0 + 16 * wave + r16] = acc[a][t][i];
}
#undef WG9_EACH
#undef WG9_LD
#undef WG9_ST
#undef WG9_LOAD_DY
#undef WG9_STORE_DY
#undef WG9_LOAD_X
#undef WG9_STORE_X
// The same for the 16-pixel-wide layers (W == 16, H even): a 32-pixel k-step is a PAIR of image rows -- k-slots 0-15 the pixels of row r,
// 16-31 those of row r + 1 (the transposing reads take any piece address, so the second half of every fragment simply points into the
// other row's image).  8-slot input-row ring (rows r - 1 .. r + 2 live, r + 3 and r + 4 arriving), one barrier per row pair.
__global__ __launch_bounds__(256, 1) void conv_wgrad9_w16_kernel(const h16* __restrict__ x, const h16* __restrict__ dy, float* __restrict__ part,
                                                              int B, int H, int Cin, int Cout, long x_ld, long dy_ld, int rows_per_blk) {
    constexpr int W = 16, xw = W + 2;
    __shared__ __attribute__((aligned(16))) h16 lds[(2 * 32 + 9 * xw) * MU_WG9_S];
    h16* dys = lds;                                            // [2][32][72]: the dy rows r, r + 1
    h16* xs = dys + 2 * 32 * MU_WG9_S;                         // [8][18][72] ring | [18][72] zeros
    h16* xz = xs + 8 * xw * MU_WG9_S;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, q = r16 >> 2, pc = r16 & 3;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const int rows = B * H;
    const int rbeg = blockIdx.x * rows_per_blk, rend = rbeg + rows_per_blk < rows ? rbeg + rows_per_blk : rows;      // both even

    f32x4 acc[4][9];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[a][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 9 * xw * MU_WG9_S / 8; i += 256) *reinterpret_cast<uint4*>(xs + i * 8) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    // one 16-byte piece of a 32-pixel run (two image rows) per thread: pixel tid >> 3, channels 8 (tid & 7) ..
    const int pp = tid >> 3, pch = (tid & 7) * 8;
    uint4 dreg, xreg;
    auto load_x2 = [&](int xr) {                               // rows xr, xr + 1 (clamped: rows past the end are never read by a valid tap)
        const int r0 = xr + (pp >> 4);
        const int rc = r0 < 0 ? 0 : (r0 >= rows ? rows - 1 : r0);
        xreg = *reinterpret_cast<const uint4*>(x + ((long)rc * W + (pp & 15)) * x_ld + ci0 + pch);
    };
    auto store_x2 = [&](int xr) {
        const int r0 = xr + (pp >> 4);
        *reinterpret_cast<uint4*>(xs + (((r0 & 7) * xw) + 1 + (pp & 15)) * MU_WG9_S + pch) = xreg;
    };
    auto load_dy2 = [&](int r) { dreg = *reinterpret_cast<const uint4*>(dy + ((long)r * W + pp) * dy_ld + co0 + pch); };
    auto store_dy2 = [&](int buf) { *reinterpret_cast<uint4*>(dys + (buf * 32 + pp) * MU_WG9_S + pch) = dreg; };

    if (rbeg < rend) {
        load_x2(rbeg - 1); store_x2(rbeg - 1);                 // rows rbeg - 1, rbeg
        load_x2(rbeg + 1); store_x2(rbeg + 1);                 // rows rbeg + 1, rbeg + 2
        load_x2(rbeg + 3);                                     // rows rbeg + 3, rbeg + 4: stored by the first iteration
        load_dy2(rbeg);
    }
    int buf = 0;
    const int lane_off = (4 * g + q) * MU_WG9_S + 4 * pc;
    for (int r = rbeg; r < rend; r += 2) {
        store_dy2(buf);
        store_x2(r + 3);
        __syncthreads();
        if (r + 2 < rend) { load_dy2(r + 2); load_x2(r + 5); }
        const int h = r % H;                                   // even; h + 1 < H
        const h16* ring[4];                                    // rows r - 1 .. r + 2 (outside the image: zeros)
        ring[0] = (h > 0 ? xs + ((r - 1) & 7) * xw * MU_WG9_S : xz) + lane_off + wave * 16;
        ring[1] = xs + (r & 7) * xw * MU_WG9_S + lane_off + wave * 16;
        ring[2] = xs + ((r + 1) & 7) * xw * MU_WG9_S + lane_off + wave * 16;
        ring[3] = (h + 2 < H ? xs + ((r + 2) & 7) * xw * MU_WG9_S : xz) + lane_off + wave * 16;
        const h16* dyt = dys + buf * 32 * MU_WG9_S + lane_off;
        h16x8 af[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(dyt + a * 16));
            auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(dyt + a * 16 + 16 * MU_WG9_S));
            af[a] = (h16x8){(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // k-slots 0-15: row r, its tap row r + kh - 1 = ring[kh]; k-slots 16-31: row r + 1, tap row ring[kh + 1]
            auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(ring[t / 3] + (t % 3) * MU_WG9_S));
            auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)(ring[t / 3 + 1] + (t % 3) * MU_WG9_S));
            const h16x8 bf = {(h16)lo[0], (h16)lo[1], (h16)lo[2], (h16)lo[3], (h16)hi[0], (h16)hi[1], (h16)hi[2], (h16)hi[3]};
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a], bf, acc[a][t], 0, 0, 0);
        }
        buf ^= 1;
    }
    float* slab = part + (long)blockIdx.x * 9 * Cout * Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                slab[((long)t * Cout + co0 + 16 * a + 4 * g + i) * Cin + ci0 + 16 * wave + r16] = acc[a][t][i];
}
#ifndef MU_WG9_W16
#define MU_WG9_W16 1
#endif
#ifndef MU_WG9_MAXC16
#define MU_WG9_MAXC16 512
#endif
// `pair` (mu_conv_wgrad_h, round 6): Cin counts the 16-bit COLUMNS of a chunk-encoded fp32x input -- two per channel -- and the layer-size
// limits below, which were measured on real channel counts, apply to Cin / 2 (MU_WG9_PAIR_WIDE=0 in the environment: the limits as they are)
static inline int wg9_pair_div(bool pair) { return (pair && !(getenv("MU_WG9_PAIR_WIDE") && atoi(getenv("MU_WG9_PAIR_WIDE")) == 0)) ? 2 : 1; }
static inline bool wgrad9_w16_choose(int B, int H, int W, int Cin, int Cout, int taps, int dtype, int* nb, int* rpb, bool pair = false) {
    if (!MU_WG9_W16 || dtype != MU_F16 || taps != 9 || W != 16 || H % 2 || Cin % 64 || Cout % 64 || Cin / wg9_pair_div(pair) > MU_WG9_MAXC16 || Cout > MU_WG9_MAXC16) return false;
    const long rows = (long)B * H;
    long want = MU_WG9_BLOCKS / ((Cin / 64) * (Cout / 64));
    if (want < 1) want = 1;
    long r = (rows + want - 1) / want;
    r = (r + 1) / 2 * 2;                                       // whole row pairs
    *rpb = (int)r;
    *nb = (int)((rows + r - 1) / r);
    return true;
}

static inline bool wgrad9_choose(int B, int H, int W, int Cin, int Cout, int taps, int dtype, int* nb, int* rpb, bool pair = false) {
    // which layers: one of the two channel counts 64 (the ring kernel's 64-wide tiles), or both <= MU_WG9_MAXC32 at W = 32 -- in-process A/B, B = 64:
    // 32^2 128->128 40.9 -> 36.1 us; the 128-wide layers at 64^2 / 128^2 are 5-7 % SLOWER here than on the ring kernel's 128 x 128 tiles
    const int maxc = W <= 32 ? MU_WG9_MAXC32 : MU_WG9_MAXC;
    const bool need64 = MU_WG9_NEED64 && W > 32;
    const int creal = Cin / wg9_pair_div(pair);
    if (!MU_WG9 || dtype != MU_F16 || taps != 9 || W % 32 || W > 128 || Cin % 64 || Cout % 64 || (need64 && !(creal == 64 || Cout == 64)) ||
        creal > maxc || Cout > maxc) return false;
    const long rows = (long)B * H;
    long want = (W <= 64 ? MU_WG9_BLOCKS_W64 : MU_WG9_BLOCKS) / ((Cin / 64) * (Cout / 64));
    if (want > rows) want = rows;
    if (want < 1) want = 1;
    *rpb = (int)((rows + want - 1) / want);
    *nb = (int)((rows + *rpb - 1) / *rpb);
    return true;
}

static long wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps, bool pair) {
    int bco, bci, nsplit; long pps;
    wgrad_tile(Cin, Cout, &bco, &bci);
    wgrad_plan((long)B * H * W, Cin, Cout, taps, bco, bci, &nsplit, &pps);
    long a = (long)nsplit * taps * Cout * Cin * sizeof(float);
    wgrad_tile(Cin, Cout, &bco, &bci, taps, true);          // the fp16 plan of a 1x1 layer may use more pixel ranges
    wgrad_plan((long)B * H * W, Cin, Cout, taps, bco, bci, &nsplit, &pps);
    if ((long)nsplit * taps * Cout * Cin * (long)sizeof(float) > a) a = (long)nsplit * taps * Cout * Cin * sizeof(float);
    if (taps == 1) a += (long)nsplit * Cout * (long)sizeof(float);          // bias partials of mu_conv_wgrad_bias
    int tco, tci;
    if (wgrad3_choose(H, W, Cin, Cout, taps, MU_F16, &tco, &tci)) {
        wgrad3_plan((long)B * H * W, Cin, Cout, tco, tci, &nsplit, &pps);
        long b = (long)nsplit * taps * Cout * Cin * sizeof(float);
        if (b > a) a = b;
    }
    int nb9, rpb9;
    if (wgrad9_choose(B, H, W, Cin, Cout, taps, MU_F16, &nb9, &rpb9, pair) || wgrad9_w16_choose(B, H, W, Cin, Cout, taps, MU_F16, &nb9, &rpb9, pair)) {
        long b = (long)nb9 * taps * Cout * Cin * sizeof(float);
        if (b > a) a = b;
    }
    if (taps == 9 && Cin == 32) {                       // first-layer kernel (<= 4 valid input channels): [blocks][9][Cout][4]
        long c = (long)MU_RGB_MAXBLK * 9 * Cout * 4 * sizeof(float);
        if (c > a) a = c;
    }
    return a;
}

extern "C" long mu_conv_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps) {
    return wgrad_workspace_bytes(B, H, W, Cin, Cout, taps, false);
}

template <typename T, int TAPS>
static int wgrad_launch(const T* x, const T* dy, float* part, int B, int H, int W, int Cin, int Cout, long x_ld, long dy_ld, int bt,
                        int nsplit, long pps, hipStream_t st, int bci = 0, float* bias_part = nullptr) {
    if constexpr (TAPS == 1 && (sizeof(T) == 2 || std::is_same<T, xf32>::value)) {
        if (wgrad_is_wide(bt)) {                            // bt x bci tiles (bci = 64 or 128)
            const int grid = ((Cout + bt - 1) / bt) * ((Cin + bci - 1) / bci) * nsplit;
#define WG1(TM_, TN_)                                                                                                                  \
    do {                                                                                                                               \
        if (bias_part) conv_wgrad_kernel<T, TM_, TN_, 2, 1, true><<<grid, 256, 0, st>>>(x, dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, nsplit, pps, bias_part); \
        else conv_wgrad_kernel<T, TM_, TN_, 2, 1><<<grid, 256, 0, st>>>(x, dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, nsplit, pps);    \
    } while (0)
            if (bt == 192 && bci == 64) WG1(6, 2);
            else if (bt == 192) WG1(6, 4);
            else if (bci == 64) WG1(5, 2);
            else WG1(5, 4);
#undef WG1
            return MU_OK;
        }
    }
    if (bias_part) return MU_ERR_SHAPE;
    const int nt = ((Cout + bt - 1) / bt) * ((Cin + bt - 1) / bt);
    const int grid = TAPS * nt * nsplit;
    if (bt == 128) conv_wgrad_kernel<T, 4, 4, 2, TAPS><<<grid, 256, 0, st>>>(x, dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, nsplit, pps);
    else if (bt == 64) conv_wgrad_kernel<T, 2, 2, 2, TAPS><<<grid, 256, 0, st>>>(x, dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, nsplit, pps);
    else conv_wgrad_kernel<T, 1, 1, 2, TAPS><<<grid, 256, 0, st>>>(x, dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, nsplit, pps);
    return MU_OK;
}

extern "C" int mu_conv_wgrad_bias_supported(int Cin, int Cout, int taps, int dtype) {
    int bco, bci;
    wgrad_tile(Cin, Cout, &bco, &bci, taps, dtype == MU_F16);
    return dtype == MU_F16 && taps == 1 && Cin % 32 == 0 && Cout % 32 == 0 && wgrad_is_wide(bco) ? 1 : 0;
}

// pair_I > 0 (mu_conv_wgrad_h): x is the 16-bit view of a chunk-encoded fp32x input (Cin = 2 x its channels), dy the scaled fp16 gradient;
// the slabs are reduced pairwise into dw_oihw[cout_valid][pair_I][taps] and multiplied by oscale[1]
static int conv_wgrad_impl(const void* x, const void* dy, float* dw_oihw, float* db, int B, int H, int W, int Cin, int Cout, int taps,
                           int cin_valid, int cout_valid, long x_ld, long dy_ld, void* workspace, long ws_bytes, int dtype,
                           void* stream, int pair_I = 0, const float* oscale = nullptr) {
    if (!x || !dy || !dw_oihw || !workspace || B <= 0 || H <= 0 || W <= 0) return MU_ERR_ARG;
    if (db && !mu_conv_wgrad_bias_supported(Cin, Cout, taps, dtype)) return MU_ERR_SHAPE;
    if (Cin % 32 || Cout % 32 || x_ld < Cin || dy_ld < Cout || x_ld % 8 || dy_ld % 8) return MU_ERR_SHAPE;
    if (cin_valid <= 0 || cin_valid > Cin || cout_valid <= 0 || cout_valid > Cout) return MU_ERR_ARG;
    if (taps != 1 && taps != 9) return MU_ERR_ARG;
    int bco, bci, nsplit; long pps;
    // (fp32x: the wide 160-channel tile of the class head as well -- 32 x 32 tiles otherwise; the 192-wide q/k/v tiles stay fp16-only:
    //  64 x 64 tiles serve those layers)
    wgrad_tile(Cin, Cout, &bco, &bci, taps, dtype == MU_F16 || (dtype == MU_F32X && Cout == 160));
    wgrad_plan((long)B * H * W, Cin, Cout, taps, bco, bci, &nsplit, &pps);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    if (taps == 9 && cin_valid <= 3 && Cout % 64 == 0 && W <= MU_RGB_MAXW && (dtype == MU_F16 || dtype == MU_F32 || dtype == MU_F32X)) {
        // (plain-FMA kernel: the fp32x mode runs it as fp32)
        long nb = ws_bytes / ((long)9 * Cout * 4 * (long)sizeof(float));
        if (nb > MU_RGB_MAXBLK) nb = MU_RGB_MAXBLK;
        if (nb > (long)B * H) nb = (long)B * H;
        if (nb < 1) return MU_ERR_WORKSPACE;
        dim3 grid((int)nb, Cout / 64);
        if (dtype == MU_F16 && MU_RGB2 && W % 32 == 0 && x_ld % 4 == 0) {
            // matrix-core form: a band of consecutive image rows per block (the grid covers the rows exactly: an idle block writes a zero slab)
            if (nb > 768) nb = 768;                            // three 45 KB blocks per CU: one round
            const int rpb = (int)(((long)B * H + nb - 1) / nb);
            nb = ((long)B * H + rpb - 1) / rpb;
            grid.x = (int)nb;
            const size_t lds = ((size_t)2 * W * MU_RGB2_DYS + 5 * (size_t)(W + 2) * 4) * sizeof(h16);
            const size_t lds_red = (size_t)4 * 64 * 16 * sizeof(float);
#define RGB2(NPT) wgrad_rgb2_kernel<NPT><<<grid, 256, lds > lds_red ? lds : lds_red, st>>>((const h16*)x, (const h16*)dy, part, B, H, W, Cout, x_ld, dy_ld, rpb)
            switch (W / 32) {
                case 1: RGB2(1); break; case 2: RGB2(2); break; case 3: RGB2(3); break; case 4: RGB2(4); break;
                case 5: RGB2(5); break; case 6: RGB2(6); break; case 7: RGB2(7); break; default: RGB2(8); break;
            }
#undef RGB2
        }
        else if (dtype == MU_F16) wgrad_rgb_kernel<h16><<<grid, 256, 0, st>>>((const h16*)x, (const h16*)dy, part, B, H, W, Cout, cin_valid, x_ld, dy_ld);
        else wgrad_rgb_kernel<float><<<grid, 256, 0, st>>>((const float*)x, (const float*)dy, part, B, H, W, Cout, cin_valid, x_ld, dy_ld);
        const long n = (long)cout_valid * cin_valid * 9;
        const long nblk = (n + 63) / 64;
        wgrad_reduce_kernel<4><<<(int)(nblk > 4096 ? 4096 : nblk), 256, 0, st>>>(part, dw_oihw, (int)nb, 9, Cout, 4, cout_valid, cin_valid);
        MU_CHECK_LAUNCH();
        return MU_OK;
    }
    int tco, tci, nb9, rpb9;
    if (wgrad9_choose(B, H, W, Cin, Cout, taps, dtype, &nb9, &rpb9, pair_I > 0) && x_ld % 8 == 0 && dy_ld % 8 == 0) {
        if (ws_bytes < (long)nb9 * taps * Cout * Cin * (long)sizeof(float)) return MU_ERR_WORKSPACE;
        const dim3 grid9(nb9, Cout / 64, Cin / 64);
#define WG9(NPT) conv_wgrad9_kernel<NPT><<<grid9, 256, 0, st>>>((const h16*)x, (const h16*)dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, rpb9)
        switch (W / 32) { case 1: WG9(1); break; case 2: WG9(2); break; case 3: WG9(3); break; default: WG9(4); break; }
#undef WG9
        nsplit = nb9;
    } else if (wgrad9_w16_choose(B, H, W, Cin, Cout, taps, dtype, &nb9, &rpb9, pair_I > 0) && x_ld % 8 == 0 && dy_ld % 8 == 0) {
        if (ws_bytes < (long)nb9 * taps * Cout * Cin * (long)sizeof(float)) return MU_ERR_WORKSPACE;
        conv_wgrad9_w16_kernel<<<dim3(nb9, Cout / 64, Cin / 64), 256, 0, st>>>((const h16*)x, (const h16*)dy, part, B, H, Cin, Cout, x_ld, dy_ld, rpb9);
        nsplit = nb9;
    } else
    if (wgrad3_choose(H, W, Cin, Cout, taps, dtype, &tco, &tci)) {
        wgrad3_plan((long)B * H * W, Cin, Cout, tco, tci, &nsplit, &pps);
        if (ws_bytes < (long)nsplit * taps * Cout * Cin * (long)sizeof(float)) return MU_ERR_WORKSPACE;
        const int grid = 3 * (Cout / tco) * (Cin / tci) * nsplit;
        const h16 *xh = (const h16*)x, *dyh = (const h16*)dy;
#define WG3(...) conv_wgrad3_kernel<h16, __VA_ARGS__><<<grid, (tco == 128 && tci == 128) ? 512 : 256, 0, st>>>(xh, dyh, part, B, H, W, Cin, Cout, x_ld, dy_ld, nsplit, pps)
        const bool two_row32 = W == 32 && H % 2 == 0, flat64 = W % 64 == 0;
        if (tco == 128 && tci == 128) {   // 8 waves, 64x32 tile x 3 taps per wave (96 accumulators): 2 waves/SIMD
            if (W == 16) WG3(4, 2, 2, 8, true);
            else if (two_row32 && MU_WG_SPS2 && MU_WG_PP == 2) WG3(4, 2, 2, 8, true, 2, true);      // (W = 32: +2.6 % slower, opt-in)
            else if (two_row32 && MU_WG_SPS2) WG3(4, 2, 2, 8, true, 2);
            else if (flat64 && MU_WG_SPS2 && MU_WG_PP) WG3(4, 2, 2, 8, false, 2, true);
            else if (flat64 && MU_WG_SPS2) WG3(4, 2, 2, 8, false, 2);
            else WG3(4, 2, 2, 8);
        } else if (tco == 128) {          // 128 x 64: 4 waves, 64x32 per wave
            if (W == 16) WG3(4, 2, 2, 4, true);
            else WG3(4, 2, 2, 4);
        } else if (tci == 128) {          // 64 x 128: 4 waves, 32x64 per wave
            if (W == 16) WG3(2, 4, 2, 4, true);
            else WG3(2, 4, 2, 4);
        } else {
            if (W == 16) WG3(2, 2, 2, 4, true);
            else if (two_row32 && MU_WG_SPS2_64) WG3(2, 2, 2, 4, true, 2);
            else if (flat64 && MU_WG_SPS2_64) WG3(2, 2, 2, 4, false, 2);
            else WG3(2, 2, 2);
        }
#undef WG3
    } else if (ws_bytes < (long)nsplit * taps * Cout * Cin * (long)sizeof(float)) {
        return MU_ERR_WORKSPACE;
    } else if (dtype == MU_F16) {
        if (taps == 9) wgrad_launch<h16, 9>((const h16*)x, (const h16*)dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, bco, nsplit, pps, st);
        else {
            float* bias_part = db ? part + (long)nsplit * Cout * Cin : nullptr;
            if (db && ws_bytes < ((long)nsplit * Cout * Cin + (long)nsplit * Cout) * (long)sizeof(float)) return MU_ERR_WORKSPACE;
            int rc = wgrad_launch<h16, 1>((const h16*)x, (const h16*)dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, bco, nsplit, pps, st, bci, bias_part);
            if (rc) return rc;
            if (db) wgrad_bias_reduce_kernel<<<mu_cdiv(cout_valid, 16), 256, 0, st>>>(bias_part, nsplit, Cout, cout_valid, db);
        }
    } else if (dtype == MU_F32) {
        if (taps == 9) wgrad_launch<float, 9>((const float*)x, (const float*)dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, bco, nsplit, pps, st);
        else wgrad_launch<float, 1>((const float*)x, (const float*)dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, bco, nsplit, pps, st);
    } else if (dtype == MU_F32X) {
        // (3x3 layers of this mode: mu_conv_wgrad_h -- fp16-pair input, one scaled fp16 dy; only the <= 3-channel first layer, served above on
        //  plain operands, comes through here with taps = 9)
        if (taps == 9) return MU_ERR_ARG;
        else wgrad_launch<xf32, 1>((const xf32*)x, (const xf32*)dy, part, B, H, W, Cin, Cout, x_ld, dy_ld, bco, nsplit, pps, st, bci);
    } else return MU_ERR_ARG;
    if (pair_I > 0) {
        const long n2 = (long)cout_valid * pair_I * taps;
        if (nsplit >= 16) {
            const long nb = (n2 + 63) / 64;
            wgrad_reduce_pair_kernel<4><<<(int)(nb > 4096 ? 4096 : nb), 256, 0, st>>>(part, dw_oihw, nsplit, taps, Cout, Cin, cout_valid, pair_I, oscale);
        } else {
            const long nb = (n2 + 255) / 256;
            wgrad_reduce_pair_kernel<1><<<(int)(nb > 2048 ? 2048 : nb), 256, 0, st>>>(part, dw_oihw, nsplit, taps, Cout, Cin, cout_valid, pair_I, oscale);
        }
        MU_CHECK_LAUNCH();
        return MU_OK;
    }
    const long n = (long)cout_valid * cin_valid * taps;
    if (nsplit >= 16) {
        const long nb = (n + 63) / 64;
        wgrad_reduce_kernel<4><<<(int)(nb > 4096 ? 4096 : nb), 256, 0, st>>>(part, dw_oihw, nsplit, taps, Cout, Cin, cout_valid, cin_valid, oscale);
    } else {
        const long nb = (n + 255) / 256;
        wgrad_reduce_kernel<1><<<(int)(nb > 2048 ? 2048 : nb), 256, 0, st>>>(part, dw_oihw, nsplit, taps, Cout, Cin, cout_valid, cin_valid, oscale);
    }
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_conv_wgrad(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int taps,
                             int cin_valid, int cout_valid, long x_ld, long dy_ld, void* workspace, long ws_bytes, int dtype,
                             void* stream) {
    return conv_wgrad_impl(x, dy, dw_oihw, nullptr, B, H, W, Cin, Cout, taps, cin_valid, cout_valid, x_ld, dy_ld, workspace, ws_bytes, dtype,
                           stream);
}

// fp32x, 3x3 layers: the TWO-TERM weight gradient (round 6).  x: the layer's saved input, chunk-encoded fp16 pairs (mu_split_encode_h4 /
// mu_bn_act_fwd with MU_F32X; Cin fp32 channels per pixel, row stride x_ld floats); dy_h: ONE scaled fp16 operand (rows of Cout halves,
// stride dy_ld halves) with its scale pair dy_scale = {S, 1 / S}.  The fp16 kernels of mu_conv_wgrad run on the 16-bit view of x
// ([M, 2 Cin] halves: dy x hi and dy x lo columns -- two MFMAs per product), the slab reduce adds the column pairs and applies 1 / S.
extern "C" long mu_conv_wgrad_h_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    return wgrad_workspace_bytes(B, H, W, 2 * Cin, Cout, 9, true);
}
extern "C" int mu_conv_wgrad_h(const void* x, const void* dy_h, const float* dy_scale, float* dw_oihw, int B, int H, int W, int Cin, int Cout,
                               int cin_valid, int cout_valid, long x_ld, long dy_ld, void* workspace, long ws_bytes, void* stream) {
    if (!dy_scale || cin_valid <= 3) return MU_ERR_ARG;      // (the <= 3-channel first layer is a plain-FMA kernel on plain operands: mu_conv_wgrad)
    if (Cin % 32 || cin_valid > Cin) return MU_ERR_SHAPE;
    return conv_wgrad_impl(x, dy_h, dw_oihw, nullptr, B, H, W, 2 * Cin, Cout, 9, 2 * Cin, cout_valid, 2 * x_ld, dy_ld, workspace, ws_bytes, MU_F16,
                           stream, cin_valid, dy_scale);
}

// The ONE-term form: x_h = the fp16 ROUNDING of the layer's input (rows of Cin halves, stride x_ld halves: the second output of
// mu_bn_act_fwd_enc / mu_split_encode_h4x), dy_h / dy_scale as above -- the fp16 weight-gradient kernels as they are, one MFMA per product,
// 1 / S applied in the slab reduce.  dW sums over every pixel of the batch and the 2^-12 roundings of x are random-signed (the sum carries ~2^-12 of the root-sum-square
// of its terms): the CPU sizing shows no
// change of any gradient metric against the two-term form (tests/aids/numerics_conv_bwd_two_term.py h1x).  Workspace: mu_conv_wgrad_workspace_bytes.
extern "C" int mu_conv_wgrad_h1(const void* x_h, const void* dy_h, const float* dy_scale, float* dw_oihw, int B, int H, int W, int Cin, int Cout,
                                int cin_valid, int cout_valid, long x_ld, long dy_ld, void* workspace, long ws_bytes, void* stream) {
    if (!dy_scale || cin_valid <= 3) return MU_ERR_ARG;
    return conv_wgrad_impl(x_h, dy_h, dw_oihw, nullptr, B, H, W, Cin, Cout, 9, cin_valid, cout_valid, x_ld, dy_ld, workspace, ws_bytes, MU_F16,
                           stream, 0, dy_scale);
}

extern "C" int mu_conv_wgrad_bias(const void* x, const void* dy, float* dw_oihw, float* db, int B, int H, int W, int Cin, int Cout,
                                  int taps, int cin_valid, int cout_valid, long x_ld, long dy_ld, void* workspace, long ws_bytes,
                                  int dtype, void* stream) {
    if (!db) return MU_ERR_ARG;
    return conv_wgrad_impl(x, dy, dw_oihw, db, B, H, W, Cin, Cout, taps, cin_valid, cout_valid, x_ld, dy_ld, workspace, ws_bytes, dtype, stream);
}
