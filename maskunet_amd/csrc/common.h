// Shared device helpers for the MaskAttn-UNet HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/maskunet_hip.h"

#define MU_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return MU_ERR_LAUNCH;        \
    } while (0)

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int mu_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// MU_F32X ("fp32x"): fp32 STORAGE, matrix products on the bf16 matrix cores with every fp32 operand split into two bf16 parts,
//     x = hi + lo,  hi = bf16_rne(x),  lo = bf16_rne(x - hi)          (|x - hi - lo| <= 2^-17 |x|)
//     a * b ~= hi_a hi_b + hi_a lo_b + lo_a hi_b                       (the dropped lo_a lo_b is <= 2^-16 |a b|)
// accumulated in the fp32 MFMA accumulator: ~1e-5 relative per product with random sign (the "bf16x3" scheme behind
// torch.set_float32_matmul_precision("high")), against 2^-11 for one fp16 rounding -- and three 16-cycle bf16 MFMAs per K = 16
// where the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate) needs four 32-cycle ones.  The kernels are the fp32
// instantiations: `xf32` is float storage under another name, so that the fragment traits (conv.hip Mma<T>, attn.hip AT<T>) can
// pick the split arithmetic; everything that is not a matrix product treats MU_F32X as MU_F32.
// ------------------------------------------------------------------------------------------
struct xf32 {
    float v;
    __host__ __device__ xf32() = default;
    __host__ __device__ xf32(float f) : v(f) {}
    __host__ __device__ operator float() const { return v; }
};
static_assert(sizeof(xf32) == 4, "xf32 is float storage");
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct SplitF4 { s16x4 hi, lo; };          // four fp32 values as bf16 hi | bf16 lo: the A / B operand pair of v_mfma_f32_16x16x16_bf16
struct SplitF8 { bf16x8 hi, lo; };         // eight: v_mfma_f32_16x16x32_bf16

// (hi, lo) of two floats, packed: v_cvt_pk_bf16_f32, two bit ops, two subtracts, v_cvt_pk_bf16_f32
__device__ __forceinline__ void mu_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const bf16x2 h = {(__bf16)x0, (__bf16)x1};
    hi = __builtin_bit_cast(uint32_t, h);
    const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
    const bf16x2 l = {(__bf16)(x0 - h0), (__bf16)(x1 - h1)};
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ SplitF4 mu_split4(const f32x4& x) {
    uint2 h, l;
    mu_split2(x[0], x[1], h.x, l.x);
    mu_split2(x[2], x[3], h.y, l.y);
    SplitF4 r;
    r.hi = __builtin_bit_cast(s16x4, h);
    r.lo = __builtin_bit_cast(s16x4, l);
    return r;
}
__device__ __forceinline__ SplitF8 mu_split8(const float (&x)[8]) {
    uint4 h, l;
    mu_split2(x[0], x[1], h.x, l.x);
    mu_split2(x[2], x[3], h.y, l.y);
    mu_split2(x[4], x[5], h.z, l.z);
    mu_split2(x[6], x[7], h.w, l.w);
    SplitF8 r;
    r.hi = __builtin_bit_cast(bf16x8, h);
    r.lo = __builtin_bit_cast(bf16x8, l);
    return r;
}
// Chunk encoding of an fp32x MATRIX OPERAND in memory: every aligned 16-byte chunk of four fp32 values is stored as
//     [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3]      (bf16 each: bytes 0-7 the hi parts, bytes 8-15 the lo parts)
// -- the same 16 bytes, so strides, LDS-DMA pieces, swizzles and tile shapes of the fp32 kernels are untouched, but a 16-byte
// fragment load IS the (hi, lo) operand pair (no VALU in the sweep) and ds_read_b64_tr_b16 transposes the hi / lo halves like an
// fp16 tile.  Operands are encoded once per tensor by mu_split_encode (one read + one write), not once per fragment per wave.
__device__ __forceinline__ uint4 mu_enc4(const f32x4& x) {
    uint4 e;
    mu_split2(x[0], x[1], e.x, e.z);
    mu_split2(x[2], x[3], e.y, e.w);
    return e;
}
__device__ __forceinline__ f32x4 mu_dec4(const uint4& e) {
    return (f32x4){__uint_as_float(e.x << 16) + __uint_as_float(e.z << 16), __uint_as_float(e.x & 0xffff0000u) + __uint_as_float(e.z & 0xffff0000u),
                   __uint_as_float(e.y << 16) + __uint_as_float(e.w << 16), __uint_as_float(e.y & 0xffff0000u) + __uint_as_float(e.w & 0xffff0000u)};
}
__device__ __forceinline__ SplitF4 mu_frag_enc(const uint4& e) {
    SplitF4 r;
    r.hi = __builtin_bit_cast(s16x4, make_uint2(e.x, e.y));
    r.lo = __builtin_bit_cast(s16x4, make_uint2(e.z, e.w));
    return r;
}

// c += a * b over K = 16 / 32 with both operands split: smallest terms first
__device__ __forceinline__ void mu_mma_split(const SplitF4& a, const SplitF4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.hi, c, 0, 0, 0);
}
__device__ __forceinline__ void mu_mma_split(const SplitF8& a, const SplitF8& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// fp16-pair encoding of the ATTENTION operands in the fp32x mode (round 5): every aligned 32-byte group of eight fp32 values is
//     [hi0 .. hi7 | lo0 .. lo7]      (fp16 each: bytes 0-15 the hi parts, bytes 16-31 the lo parts)
// with hi = fp16_rne(s x), lo = fp16_rne(s x - hi) and s a power of two (1 for qkv; chosen from max|dY| for the gradient so that
// the tiny values of a backward pass sit inside fp16's exponent range).  hi + lo carries 22 mantissa bits (bf16 pairs: 16) down to
// an absolute floor of 2^-25, and -- the point of it -- the softmax probabilities P and dS = P o (dP - delta) can then enter the
// matrix core as ONE fp16 operand straight from the accumulators, as in the fp16 kernels: P V, dV = P^T dO, dK = dS^T Q and
// dQ = dS K are two fp16 MFMAs per product (P x hi, P x lo) instead of three bf16 ones, and the ~50 VALU instructions per 32 x 32 tile
// that split P / dS in registers are gone.  Q K^T and dO V^T keep three terms (lo x hi + hi x lo + hi x hi).  Sizing on the CPU
// oracle (tests/aids/numerics_attn_single_term.py, the reference's own golden unet1_c150_b2_train): outputs 2.0e-5, worst parameter
// gradient 5.9e-3 against gates of 1e-3 / 5e-2.  |s x| must stay below 65504 (a larger value encodes as inf and surfaces as NaN).
// A 32-byte group = two 16-byte LDS-DMA pieces, so strides, pieces and tile shapes of the fp32 kernels are untouched.
// ------------------------------------------------------------------------------------------
struct SplitH8 { h16x8 hi, lo; };
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mu_hsplit2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const h16x2 h = {(h16)x0, (h16)x1};
    hi = __builtin_bit_cast(uint32_t, h);
    const h16x2 l = {(h16)(x0 - (float)h[0]), (h16)(x1 - (float)h[1])};
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ SplitH8 mu_hsplit8(const float (&x)[8]) {
    uint4 h, l;
    mu_hsplit2(x[0], x[1], h.x, l.x);
    mu_hsplit2(x[2], x[3], h.y, l.y);
    mu_hsplit2(x[4], x[5], h.z, l.z);
    mu_hsplit2(x[6], x[7], h.w, l.w);
    SplitH8 r;
    r.hi = __builtin_bit_cast(h16x8, h);
    r.lo = __builtin_bit_cast(h16x8, l);
    return r;
}
__device__ __forceinline__ void mu_hdec8(const SplitH8& e, float (&x)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)e.hi[i] + (float)e.lo[i];
}

// ------------------------------------------------------------------------------------------
// fp16-pair CHUNK encoding of the 3x3-convolution operands in the fp32x mode (round 6; `xh32` = float storage like xf32): every aligned
// 16-byte chunk of four fp32 values is
//     [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3]      (fp16 each)
// -- the layout of the bf16 chunk encoding above (same strides, LDS-DMA pieces, swizzles, tile shapes) with fp16 halves: 22 mantissa
// bits instead of 16 down to an absolute floor of 2^-25 (v_mfma_f32_16x16x32_f16 keeps subnormal inputs: tools/probe_mfma_denorm.py).
// Forward: three fp16 MFMAs per product (lo hi + hi lo + hi hi).  Backward: dy travels as ONE fp16 operand under a per-tensor
// power-of-two scale (written so by the BatchNorm backward that produces it, norm.hip) against the two-term pair of its partner -- the
// weights for the data gradient, the saved input for the weight gradient: two MFMAs per product.  Weights are encoded under a static
// shift of 2^MU_XH_WSHIFT so that their lo halves stay fp16-normal (|w| < 2^(16 - MU_XH_WSHIFT)); the conv epilogues undo it.
// Sizing on the CPU oracle before building: tests/aids/numerics_conv_bwd_two_term.py (outputs 4.8e-6, worst gradient 1.5e-3 on the
// reference golden; the bf16 scheme it replaces: 5.1e-5 / 2.1e-2).  1x1 convolutions / Linear layers keep the bf16 encoding (their
// backward operands are unscaled gradients, far below fp16's range).
// ------------------------------------------------------------------------------------------
#define MU_XH_WSHIFT 6
struct xh32 {
    float v;
    __host__ __device__ xh32() = default;
    __host__ __device__ xh32(float f) : v(f) {}
    __host__ __device__ operator float() const { return v; }
};
static_assert(sizeof(xh32) == 4, "xh32 is float storage");
struct SplitH4 { h16x4 hi, lo; };
__device__ __forceinline__ uint4 mu_ench4(const f32x4& x) {
    uint4 e;
    mu_hsplit2(x[0], x[1], e.x, e.z);
    mu_hsplit2(x[2], x[3], e.y, e.w);
    return e;
}
__device__ __forceinline__ f32x4 mu_dech4(const uint4& e) {
    const h16x4 h = __builtin_bit_cast(h16x4, make_uint2(e.x, e.y)), l = __builtin_bit_cast(h16x4, make_uint2(e.z, e.w));
    return (f32x4){(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
}
template <typename T> struct mu_is_split { static constexpr bool value = false; };
template <> struct mu_is_split<xf32> { static constexpr bool value = true; };
template <> struct mu_is_split<xh32> { static constexpr bool value = true; };

// 16-byte vector of T: 8 halves or 4 floats.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ __forceinline__ void load_nt(const float* p) { load(p); }
    __device__ __forceinline__ void store_nt(float* p) const { store(p); }
    __device__ __forceinline__ float get(int i) const { return reinterpret_cast<const float*>(&v)[i]; }
    __device__ __forceinline__ void set(int i, float f) { reinterpret_cast<float*>(&v)[i] = f; }
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <> struct Vec16<h16> {
    static constexpr int N = 8;
    h16x8 v;
    __device__ __forceinline__ void load(const h16* p) { v = *reinterpret_cast<const h16x8*>(p); }
    __device__ __forceinline__ void store(h16* p) const { *reinterpret_cast<h16x8*>(p) = v; }
    __device__ __forceinline__ void load_nt(const h16* p) { v = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(p)); }
    __device__ __forceinline__ void store_nt(h16* p) const { __builtin_nontemporal_store(v, reinterpret_cast<h16x8*>(p)); }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = (h16)f; }
    __device__ __forceinline__ void zero() { v = (h16x8)(h16)0; }
};

__device__ __forceinline__ float mu_gelu(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float mu_gelu_grad(float x) {
    // d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi)
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float mu_act(float x, int act) {
    return act == MU_ACT_GELU ? mu_gelu(x) : (act == MU_ACT_RELU ? fmaxf(x, 0.f) : x);
}
__device__ __forceinline__ float mu_act_grad(float x, int act) {
    return act == MU_ACT_GELU ? mu_gelu_grad(x) : (act == MU_ACT_RELU ? (x > 0.f ? 1.f : 0.f) : 1.f);
}

// Exact-GELU pieces for fp16 storage: Phi(x) = 0.5 erfc(-x/sqrt2) from the Abramowitz-Stegun 7.1.26 rational form
// erfc(z) = t (a1 + t (a2 + ...)) exp(-z^2), t = 1/(1 + p z), |abs error| <= 1.5e-7 * exp(-z^2) -- three decimal orders
// below the fp16 rounding of the result -- with one v_rcp and one v_exp instead of the ~40-instruction erff.  The
// negative side is evaluated as erfc directly (no 1 - erf cancellation).  *pdf = exp(-x^2/2), shared with the gradient.
__device__ __forceinline__ float mu_phi_fast(float x, float* e_out) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);      // exp(-x^2/2)
    const float q = 0.5f * t * p * e;
    *e_out = e;
    return x < 0.f ? q : 1.0f - q;
}
// The BatchNorm passes are VALU-bound on GELU, not HBM-bound (30 instructions per element against 4-6 bytes: the backward statistics
// sweep ran at 3.8 TB/s), so the fp16-storage path evaluates Phi and GELU' as odd minimax polynomials in fp32 -- no v_exp / v_rcp, 10-12
// instructions instead of ~20 (two of them quarter-rate):
//   Phi(x)   ~ 0.5 + xc P(xc^2),  xc = clamp(x, +-4.25), deg P = 8:  |x Phi~(x) - gelu(x)| <= 5.9e-5 for |x| <= 1e2 (fp32 Horner included)
//   GELU'(x) ~ 0.5 + xc R(xc^2),  xc = clamp(x, +-4.5),  deg R = 9:  |error| <= 2.05e-4 (fp32 Horner included)
// i.e. below half an fp16 ulp of the stored results around |y| >= 0.25 and far inside the fp16 path's 3e-2 parity gate; fp32 storage
// keeps erff (exact).  Beyond the clamp the polynomials are constants, not the exact limits: Phi~(-4.25) = 3e-7, so x Phi~(x) is
// -3e-7 |x| instead of 0 for large negative x (-0.018 at x = -6e4, the edge of fp16), and GELU' saturates at 1.00007 / -7e-5.  Fitted against erf in fp64 (weighted least squares iterated to equi-ripple); MU_GELU_POLY=0 restores the
// erfc rational form above (|error| < 2e-7).
#ifndef MU_GELU_POLY
#define MU_GELU_POLY 1
#endif
__device__ __forceinline__ float mu_phi_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.25f, 4.25f);
    const float t = xc * xc;
    float p = 5.564560793e-11f;
    p = fmaf(p, t, -5.327550104e-09f);
    p = fmaf(p, t, 2.255368745e-07f);
    p = fmaf(p, t, -5.626339241e-06f);
    p = fmaf(p, t, 9.341795188e-05f);
    p = fmaf(p, t, -1.108557347e-03f);
    p = fmaf(p, t, 9.815962033e-03f);
    p = fmaf(p, t, -6.634448100e-02f);
    p = fmaf(p, t, 3.989023355e-01f);
    return fmaf(xc, p, 0.5f);
}
__device__ __forceinline__ float mu_gelu_grad_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
    const float t = xc * xc;
    float p = -2.210659350e-11f;
    p = fmaf(p, t, 2.521191438e-09f);
    p = fmaf(p, t, -1.268006067e-07f);
    p = fmaf(p, t, 3.721837969e-06f);
    p = fmaf(p, t, -7.122077918e-05f);
    p = fmaf(p, t, 9.405331742e-04f);
    p = fmaf(p, t, -8.815820455e-03f);
    p = fmaf(p, t, 5.860921086e-02f);
    p = fmaf(p, t, -2.649255782e-01f);
    p = fmaf(p, t, 7.976261104e-01f);
    return fmaf(xc, p, 0.5f);
}
// Two elements per instruction: the Horner chains as packed fp32 FMAs (v_pk_fma_f32, twice the scalar rate in a VALU-bound kernel).
// Same coefficients and operation order per element as the scalar forms above -> bit-identical results.
#ifndef MU_GELU_PK
#define MU_GELU_PK 1
#endif
typedef float mu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mu_f32x2 mu_phi_poly2(mu_f32x2 x) {
    const mu_f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -4.25f, 4.25f), __builtin_amdgcn_fmed3f(x[1], -4.25f, 4.25f)};
    const mu_f32x2 t = xc * xc;
    mu_f32x2 p = (mu_f32x2)(5.564560793e-11f);
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(-5.327550104e-09f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(2.255368745e-07f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(-5.626339241e-06f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(9.341795188e-05f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(-1.108557347e-03f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(9.815962033e-03f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(-6.634448100e-02f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(3.989023355e-01f));
    return __builtin_elementwise_fma(xc, p, (mu_f32x2)(0.5f));
}
__device__ __forceinline__ mu_f32x2 mu_gelu_grad_poly2(mu_f32x2 x) {
    const mu_f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x[1], -4.5f, 4.5f)};
    const mu_f32x2 t = xc * xc;
    mu_f32x2 p = (mu_f32x2)(-2.210659350e-11f);
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(2.521191438e-09f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(-1.268006067e-07f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(3.721837969e-06f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(-7.122077918e-05f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(9.405331742e-04f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(-8.815820455e-03f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(5.860921086e-02f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(-2.649255782e-01f));
    p = __builtin_elementwise_fma(p, t, (mu_f32x2)(7.976261104e-01f));
    return __builtin_elementwise_fma(xc, p, (mu_f32x2)(0.5f));
}
template <bool FAST>
__device__ __forceinline__ float mu_act_t(float x, int act) {
    if (act == MU_ACT_GELU) {
        if (FAST) {
            if (MU_GELU_POLY) return x * mu_phi_poly(x);
            float e;
            return x * mu_phi_fast(x, &e);
        }
        return mu_gelu(x);
    }
    return act == MU_ACT_RELU ? fmaxf(x, 0.f) : x;
}
template <bool FAST>
__device__ __forceinline__ float mu_act_grad_t(float x, int act) {
    if (act == MU_ACT_GELU) {
        if (FAST) {
            if (MU_GELU_POLY) return mu_gelu_grad_poly(x);
            float e;
            const float phi = mu_phi_fast(x, &e);
            return fmaf(x * 0.39894228040143267794f, e, phi);
        }
        return mu_gelu_grad(x);
    }
    return act == MU_ACT_RELU ? (x > 0.f ? 1.f : 0.f) : 1.f;
}

// act / act' of two elements: the packed polynomials on the fp16-storage GELU path, the scalar forms otherwise
template <bool FAST>
__device__ __forceinline__ void mu_act2_t(float x0, float x1, int act, float& o0, float& o1) {
    if (FAST && MU_GELU_POLY && MU_GELU_PK && act == MU_ACT_GELU) {
        const mu_f32x2 x = {x0, x1};
        const mu_f32x2 r = x * mu_phi_poly2(x);
        o0 = r[0];
        o1 = r[1];
    } else {
        o0 = mu_act_t<FAST>(x0, act);
        o1 = mu_act_t<FAST>(x1, act);
    }
}
template <bool FAST>
__device__ __forceinline__ void mu_act_grad2_t(float x0, float x1, int act, float& o0, float& o1) {
    if (FAST && MU_GELU_POLY && MU_GELU_PK && act == MU_ACT_GELU) {
        const mu_f32x2 r = mu_gelu_grad_poly2(mu_f32x2{x0, x1});
        o0 = r[0];
        o1 = r[1];
    } else {
        o0 = mu_act_grad_t<FAST>(x0, act);
        o1 = mu_act_grad_t<FAST>(x1, act);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware block id: blocks b and b+8 share an XCD (private L2); hand each XCD a contiguous
// range of logical tiles so that tiles sharing an activation panel hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
