// Masked attention for channel counts the flash-style sweeps of attn.hip have no register tiling for (C > 256): the GENERIC path.
// Reference op: Mask2FormerAttention.forward (ade_semantic.py:163-190) accepts any `channels`; the model itself only uses 64 / 128 / 256.
//
// Per image the products run as plain GEMMs on the existing 1x1-conv entry points (mu_conv_fwd / mu_conv_wgrad: S = Q Kg^T, O = P Vg,
// dP = dO Vg^T, dQ = dS Kg, dKg = dS^T Q, dVg = P^T dO with Kg / Vg the KEPT key rows), so an N x Nk score tile per image does exist
// here -- this path trades the flash kernels' memory footprint and speed for generality.  This file holds the row kernels in between:
// gather / scatter of the kept rows, the masked row softmax, dS = P o (dP - rowsum(P o dP)) / sqrt(C), and LayerNorm([C]) forward /
// backward over any C.  T = fp16 or fp32 storage, fp32 arithmetic; no atomics (results are run-to-run identical).
#include "common.h"
#include "../../include/maskunet_hip.h"

__device__ __forceinline__ float blk_sum(float v, float* sh) {      // 256 threads
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ float blk_max(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// dst[j][0..C) = j < cnt ? src[idx[j]][0..C) : 0      (rows_out rows; src rows src_ld elements apart)
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, long src_ld, const int* __restrict__ idx,
                                                          const int* __restrict__ cnt, T* __restrict__ dst, int rows_out, int C) {
    const int n = *cnt;
    for (int j = blockIdx.x; j < rows_out; j += gridDim.x) {
        const bool live = j < n;
        const T* s = src + (long)(live ? idx[j] : 0) * src_ld;
        for (int c = threadIdx.x; c < C; c += 256) dst[(long)j * C + c] = live ? s[c] : (T)0.f;
    }
}

// dst[idx[j]][0..C) = (T) src[j][0..C) for j < cnt (dst rows dst_ld elements apart; the other rows of dst are left alone).
// cnt == 0 (an image without a visible key): every one of the dst_rows rows becomes NaN -- the reference's dK / dV of such an image are
// NaN throughout (its softmax rows are), and so are the parameter gradients summed over the batch.
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, const int* __restrict__ cnt,
                                                           T* __restrict__ dst, long dst_ld, int dst_rows, int C) {
    const int n = *cnt;
    if (n == 0) {
        for (int r = blockIdx.x; r < dst_rows; r += gridDim.x)
            for (int c = threadIdx.x; c < C; c += 256) dst[(long)r * dst_ld + c] = (T)NAN;
        return;
    }
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        T* d = dst + (long)idx[j] * dst_ld;
        for (int c = threadIdx.x; c < C; c += 256) d[c] = (T)src[(long)j * C + c];
    }
}

// S[r][j] <- softmax_j(scale * S[r][j]) over j < cnt, 0 for cnt <= j < ld; no visible key: NaN like the reference's softmax of an all -inf row
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ S, int N, int ld, const int* __restrict__ cnt, float scale) {
    __shared__ float sh[4];
    const int n = *cnt;
    for (int r = blockIdx.x; r < N; r += gridDim.x) {
        T* row = S + (long)r * ld;
        float m = -INFINITY;
        for (int j = threadIdx.x; j < n; j += 256) m = fmaxf(m, (float)row[j]);
        m = blk_max(m, sh);
        float s = 0.f;
        for (int j = threadIdx.x; j < n; j += 256) s += __expf(scale * ((float)row[j] - m));
        s = blk_sum(s, sh);
        const float inv = 1.0f / s;                      // n == 0: m = -inf, s = 0 -> NaN rows (0 * inf), as in the reference
        for (int j = threadIdx.x; j < ld; j += 256) row[j] = j < n ? (T)(__expf(scale * ((float)row[j] - m)) * inv) : (n ? (T)0.f : (T)NAN);
    }
}

// dP[r][j] <- P[r][j] * (dP[r][j] - sum_k P[r][k] dP[r][k]) * scale  (j < cnt), 0 behind (NaN rows when cnt == 0)
template <typename T>
__global__ __launch_bounds__(256) void attn_wide_ds_kernel(const T* __restrict__ P, T* __restrict__ dP, int N, int ld, const int* __restrict__ cnt,
                                                           float scale) {
    __shared__ float sh[4];
    const int n = *cnt;
    for (int r = blockIdx.x; r < N; r += gridDim.x) {
        const T* p = P + (long)r * ld;
        T* d = dP + (long)r * ld;
        float a = 0.f;
        for (int j = threadIdx.x; j < n; j += 256) a = fmaf((float)p[j], (float)d[j], a);
        a = blk_sum(a, sh);
        // (no visible key: P is NaN throughout and so is dS -- the reference's dQ of such an image is NaN)
        for (int j = threadIdx.x; j < ld; j += 256) d[j] = j < n ? (T)((float)p[j] * ((float)d[j] - a) * scale) : (n ? (T)0.f : (T)NAN);
    }
}

// out = LayerNorm_{first cv channels}(o + x) * gamma + beta (pad channels 0); one wave per row
template <typename T>
__global__ __launch_bounds__(256) void ln_rows_fwd_kernel(const T* __restrict__ o, const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ out, float* __restrict__ mean,
                                                          float* __restrict__ rstd, long rows, int C, int cv, float eps) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const T* po = o + r * C;
        const T* px = x + r * C;
        float s = 0.f;
        for (int c = lane; c < cv; c += 64) s += (float)po[c] + (float)px[c];
        const float mu = wave_sum(s) / (float)cv;
        float q = 0.f;
        for (int c = lane; c < cv; c += 64) { const float d = (float)po[c] + (float)px[c] - mu; q = fmaf(d, d, q); }
        const float rs = rsqrtf(wave_sum(q) / (float)cv + eps);
        for (int c = lane; c < C; c += 64)
            out[r * C + c] = c < cv ? (T)(((float)po[c] + (float)px[c] - mu) * rs * gamma[c] + beta[c]) : (T)0.f;
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    }
}

// dY = rstd * (g gamma - mean_c(g gamma) - xhat mean_c(g gamma xhat)),  gxh = g * xhat (its column sums are dgamma; those of g are dbeta)
template <typename T>
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(const T* __restrict__ g, const T* __restrict__ o, const T* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, T* __restrict__ dY, T* __restrict__ gxh, long rows,
                                                          int C, int cv) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const float mu = mean[r], rs = rstd[r];
        float a = 0.f, b = 0.f;
        for (int c = lane; c < cv; c += 64) {
            const float xh = ((float)o[r * C + c] + (float)x[r * C + c] - mu) * rs, gg = (float)g[r * C + c] * gamma[c];
            a += gg;
            b = fmaf(gg, xh, b);
        }
        a = wave_sum(a) / (float)cv;
        b = wave_sum(b) / (float)cv;
        for (int c = lane; c < C; c += 64) {
            if (c < cv) {
                const float xh = ((float)o[r * C + c] + (float)x[r * C + c] - mu) * rs, gv = (float)g[r * C + c];
                dY[r * C + c] = (T)(rs * (gv * gamma[c] - a - xh * b));
                gxh[r * C + c] = (T)(gv * xh);
            } else {
                dY[r * C + c] = (T)0.f;
                gxh[r * C + c] = (T)0.f;
            }
        }
    }
}

static inline int wide_grid(long n) { return (int)(n < 1 ? 1 : (n > 4096 ? 4096 : n)); }
#define WIDE_DISPATCH(dtype, CALL16, CALL32)                 \
    if ((dtype) == MU_F16) { CALL16; }                       \
    else if ((dtype) == MU_F32 || (dtype) == MU_F32X) { CALL32; } \
    else return MU_ERR_ARG;

extern "C" int mu_gather_rows(const void* src, long src_ld, const int* idx, const int* cnt, void* dst, int rows_out, int C, int dtype,
                              void* stream) {
    if (!src || !idx || !cnt || !dst || rows_out <= 0 || C <= 0 || src_ld < C) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    WIDE_DISPATCH(dtype, (gather_rows_kernel<h16><<<wide_grid(rows_out), 256, 0, st>>>((const h16*)src, src_ld, idx, cnt, (h16*)dst, rows_out, C)),
                  (gather_rows_kernel<float><<<wide_grid(rows_out), 256, 0, st>>>((const float*)src, src_ld, idx, cnt, (float*)dst, rows_out, C)))
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_scatter_rows(const float* src, const int* idx, const int* cnt, void* dst, long dst_ld, int dst_rows, int C, int dtype,
                               void* stream) {
    if (!src || !idx || !cnt || !dst || dst_rows <= 0 || C <= 0 || dst_ld < C) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    WIDE_DISPATCH(dtype, (scatter_rows_kernel<h16><<<wide_grid(dst_rows), 256, 0, st>>>(src, idx, cnt, (h16*)dst, dst_ld, dst_rows, C)),
                  (scatter_rows_kernel<float><<<wide_grid(dst_rows), 256, 0, st>>>(src, idx, cnt, (float*)dst, dst_ld, dst_rows, C)))
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_softmax_rows(void* S, int N, int ld, const int* cnt, float scale, int dtype, void* stream) {
    if (!S || !cnt || N <= 0 || ld <= 0) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    WIDE_DISPATCH(dtype, (softmax_rows_kernel<h16><<<wide_grid(N), 256, 0, st>>>((h16*)S, N, ld, cnt, scale)),
                  (softmax_rows_kernel<float><<<wide_grid(N), 256, 0, st>>>((float*)S, N, ld, cnt, scale)))
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_attn_wide_ds(const void* P, void* dP, int N, int ld, const int* cnt, float scale, int dtype, void* stream) {
    if (!P || !dP || !cnt || N <= 0 || ld <= 0) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    WIDE_DISPATCH(dtype, (attn_wide_ds_kernel<h16><<<wide_grid(N), 256, 0, st>>>((const h16*)P, (h16*)dP, N, ld, cnt, scale)),
                  (attn_wide_ds_kernel<float><<<wide_grid(N), 256, 0, st>>>((const float*)P, (float*)dP, N, ld, cnt, scale)))
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_ln_rows_fwd(const void* o, const void* x, const float* gamma, const float* beta, void* out, float* mean, float* rstd,
                              long rows, int C, int c_valid, float eps, int dtype, void* stream) {
    if (!o || !x || !gamma || !beta || !out || !mean || !rstd || rows <= 0 || C <= 0 || c_valid <= 0 || c_valid > C) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = wide_grid((rows + 3) / 4);
    WIDE_DISPATCH(dtype, (ln_rows_fwd_kernel<h16><<<grid, 256, 0, st>>>((const h16*)o, (const h16*)x, gamma, beta, (h16*)out, mean, rstd, rows, C, c_valid, eps)),
                  (ln_rows_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)o, (const float*)x, gamma, beta, (float*)out, mean, rstd, rows, C, c_valid, eps)))
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_ln_rows_bwd(const void* grad_out, const void* o, const void* x, const float* mean, const float* rstd, const float* gamma,
                              void* dY, void* g_xhat, long rows, int C, int c_valid, int dtype, void* stream) {
    if (!grad_out || !o || !x || !mean || !rstd || !gamma || !dY || !g_xhat || rows <= 0 || C <= 0 || c_valid <= 0 || c_valid > C) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = wide_grid((rows + 3) / 4);
    WIDE_DISPATCH(dtype, (ln_rows_bwd_kernel<h16><<<grid, 256, 0, st>>>((const h16*)grad_out, (const h16*)o, (const h16*)x, mean, rstd, gamma, (h16*)dY, (h16*)g_xhat, rows, C, c_valid)),
                  (ln_rows_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)grad_out, (const float*)o, (const float*)x, mean, rstd, gamma, (float*)dY, (float*)g_xhat, rows, C, c_valid)))
    MU_CHECK_LAUNCH();
    return MU_OK;
}
