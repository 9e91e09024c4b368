// Bandwidth-bound layout / pooling / resampling kernels (NHWC, 16-byte vector accesses).
// Reference ops replaced: x.view().permute (ade_semantic.py:168), the .view scramble (:190),
// nn.MaxPool2d(2) (:216), nn.Upsample(bilinear, align_corners=True) + torch.cat (:235,250-253),
// nn.Dropout(0.3) (:273,304,307), and the OIHW<->tap-major weight re-layouts of this build.
#include "common.h"
// Nontemporal loads for operands a kernel reads exactly once (`nt`: past the CU's L1, L2-served): see norm.hip MU_BN_NT.
#ifndef MU_EW_NT
#define MU_EW_NT 1
#endif
#define EW_LD(vec, ptr_) do { if (MU_EW_NT) (vec).load_nt(ptr_); else (vec).load(ptr_); } while (0)
#include "../../include/maskunet_hip.h"

// ------------------------------------------------------------------------------------------
// batched 2-D transpose with dtype conversion:  dst[b][c][r] = src[b][r][c]
// ------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_kernel(const TS* __restrict__ src, long src_ld, TD* __restrict__ dst,
                                                        long dst_ld, int R, int C, int R_pad) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const TS* s = src + (long)b * R * src_ld;
    TD* d = dst + (long)b * C * dst_ld;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? (float)s[(long)r * src_ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        int c = c0 + i, r = r0 + tx;
        if (c < C && r < R_pad) d[(long)c * dst_ld + r] = (TD)tile[tx][i];      // rows >= R hold zeros
    }
}

// Vector variant: 4-element (8/16-byte) global accesses on both sides, rows r in [R, R_pad) of the destination are written
// as zeros (channel padding of the NHWC layout, so the caller does not need a separate memset).
// Requires 4-element alignment of both base pointers and leading dimensions; edges fall back to guarded scalars.
template <typename T> struct V4 { T v[4]; };
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_vec_kernel(const TS* __restrict__ src, long src_ld, TD* __restrict__ dst,
                                                            long dst_ld, int R, int C, int R_pad) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const TS* s = src + (long)b * R * src_ld;
    TD* d = dst + (long)b * C * dst_ld;
    const int tv = (threadIdx.x & 15) * 4, ty = threadIdx.x >> 4;
#pragma unroll
    for (int i = ty; i < 64; i += 16) {
        const int r = r0 + i, c = c0 + tv;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
            if (c + 3 < C) {
                V4<TS> q = *reinterpret_cast<const V4<TS>*>(s + (long)r * src_ld + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (float)q.v[k];
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (c + k < C) v[k] = (float)s[(long)r * src_ld + c + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[i][tv + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = ty; i < 64; i += 16) {
        const int c = c0 + i, r = r0 + tv;
        if (c >= C || r >= R_pad) continue;
        V4<TD> q;
#pragma unroll
        for (int k = 0; k < 4; ++k) q.v[k] = (TD)tile[tv + k][i];     // rows >= R hold zeros
        if (r + 3 < R_pad) {
            *reinterpret_cast<V4<TD>*>(d + (long)c * dst_ld + r) = q;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (r + k < R_pad) d[(long)c * dst_ld + r + k] = q.v[k];
        }
    }
}

extern "C" int mu_transpose_pad(const void* src, int src_dtype, long src_ld, void* dst, int dst_dtype, long dst_ld,
                                int batch, int R, int C, int R_pad, void* stream) {
    if (!src || !dst || batch <= 0 || R <= 0 || C <= 0 || src_ld < C || R_pad < R || dst_ld < R_pad) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t es = src_dtype == MU_F32 ? 4 : 2, ed = dst_dtype == MU_F32 ? 4 : 2;
    const bool vec = ((size_t)src % (4 * es) == 0) && ((size_t)dst % (4 * ed) == 0) && src_ld % 4 == 0 && dst_ld % 4 == 0 &&
                     ((long)R * src_ld) % 4 == 0 && ((long)C * dst_ld) % 4 == 0;
    if (vec) {
        dim3 grid(mu_cdiv(C, 64), mu_cdiv(R_pad, 64), batch), block(256);
        if (src_dtype == MU_F32 && dst_dtype == MU_F32)
            transpose_vec_kernel<float, float><<<grid, block, 0, st>>>((const float*)src, src_ld, (float*)dst, dst_ld, R, C, R_pad);
        else if (src_dtype == MU_F32 && dst_dtype == MU_F16)
            transpose_vec_kernel<float, h16><<<grid, block, 0, st>>>((const float*)src, src_ld, (h16*)dst, dst_ld, R, C, R_pad);
        else if (src_dtype == MU_F16 && dst_dtype == MU_F32)
            transpose_vec_kernel<h16, float><<<grid, block, 0, st>>>((const h16*)src, src_ld, (float*)dst, dst_ld, R, C, R_pad);
        else if (src_dtype == MU_F16 && dst_dtype == MU_F16)
            transpose_vec_kernel<h16, h16><<<grid, block, 0, st>>>((const h16*)src, src_ld, (h16*)dst, dst_ld, R, C, R_pad);
        else
            return MU_ERR_ARG;
        MU_CHECK_LAUNCH();
        return MU_OK;
    }
    dim3 grid(mu_cdiv(C, 64), mu_cdiv(R_pad, 64), batch), block(256);
    if (src_dtype == MU_F32 && dst_dtype == MU_F32)
        transpose_kernel<float, float><<<grid, block, 0, st>>>((const float*)src, src_ld, (float*)dst, dst_ld, R, C, R_pad);
    else if (src_dtype == MU_F32 && dst_dtype == MU_F16)
        transpose_kernel<float, h16><<<grid, block, 0, st>>>((const float*)src, src_ld, (h16*)dst, dst_ld, R, C, R_pad);
    else if (src_dtype == MU_F16 && dst_dtype == MU_F32)
        transpose_kernel<h16, float><<<grid, block, 0, st>>>((const h16*)src, src_ld, (float*)dst, dst_ld, R, C, R_pad);
    else if (src_dtype == MU_F16 && dst_dtype == MU_F16)
        transpose_kernel<h16, h16><<<grid, block, 0, st>>>((const h16*)src, src_ld, (h16*)dst, dst_ld, R, C, R_pad);
    else
        return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_transpose(const void* src, int src_dtype, long src_ld, void* dst, int dst_dtype, long dst_ld,
                            int batch, int R, int C, void* stream) {
    return mu_transpose_pad(src, src_dtype, src_ld, dst, dst_dtype, dst_ld, batch, R, C, R, stream);
}

// ------------------------------------------------------------------------------------------
// weight re-layout: OIHW fp32 -> [tap][rows_pad][cols_pad] T
//   mode 0 (forward):   dst[t][o][i]      = w[o][i][t]
//   mode 1 (data-grad): dst[T-1-t][i][o]  = w[o][i][t]   (taps flipped, in/out swapped)
//   mode 2 (both, one launch): the mode-0 block [taps][rows_pad][cols_pad] followed by the mode-1 block [taps][cols_pad][rows_pad]
// rows/cols beyond the valid extent are zero.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void prep_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int O, int I, int taps,
                                   int rows_pad, int cols_pad, int mode) {
    const long n = (long)taps * rows_pad * cols_pad;
    const long total = mode == 2 ? 2 * n : n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const bool second = idx >= n;
        const int m = mode == 2 ? (second ? 1 : 0) : mode;
        const long j = second ? idx - n : idx;
        const int rp = second ? cols_pad : rows_pad, cp = second ? rows_pad : cols_pad;
        int c = j % cp;
        int r = (j / cp) % rp;
        int t = j / ((long)cp * rp);
        float v = 0.f;
        if (m == 0) {
            if (r < O && c < I) v = w[((long)r * I + c) * taps + t];
        } else {
            if (r < I && c < O) v = w[((long)c * I + r) * taps + (taps - 1 - t)];
        }
        dst[idx] = (T)v;
    }
}

// fp32x operand encodings of the prepared layouts (common.h; round 6):
//   1x1 layers / Linear: both blocks chunk-encoded bf16 pairs (mu_split_encode form);
//   3x3 layers: forward block chunk-encoded fp16 pairs of 2^MU_XH_WSHIFT * w (mu_split_encode_h4 form); data-gradient block as the
//   "HL" rows the two-term data gradient reads: row (tap, in) of rows_pad fp32 values -> [rows_pad fp16 lo | rows_pad fp16 hi] of
//   2^MU_XH_WSHIFT * w (the same bytes; the kernel walks the lo halves first, then the hi halves: smallest terms first).
__global__ __launch_bounds__(256) void split_encode_kernel(const f32x4* src, uint4* dst, long n16);
__global__ __launch_bounds__(256) void split_encode_h4_kernel(const f32x4* src, uint4* dst, long n16, float scale);
__global__ __launch_bounds__(256) void encode_hl_rows_kernel(float* rows, long nrows, int len, float scale) {
    // in place: a block owns a row, every 16-byte piece is in registers before the first store
    constexpr int MAXV = 8;                                  // up to 8 x 256 x 4 = 8192 values per row
    const int nv = len / 4;
    for (long r = blockIdx.x; r < nrows; r += gridDim.x) {
        float* row = rows + r * len;
        f32x4 v[MAXV];
#pragma unroll
        for (int u = 0; u < MAXV; ++u)
            if (threadIdx.x + u * 256 < nv) v[u] = *reinterpret_cast<const f32x4*>(row + (threadIdx.x + u * 256) * 4) * scale;
        __syncthreads();
        h16* hrow = reinterpret_cast<h16*>(row);
#pragma unroll
        for (int u = 0; u < MAXV; ++u)
            if (threadIdx.x + u * 256 < nv) {
                const uint4 e = mu_ench4(v[u]);              // (x, y) = four hi halves, (z, w) = four lo halves
                *reinterpret_cast<uint2*>(hrow + (threadIdx.x + u * 256) * 4) = make_uint2(e.z, e.w);
                *reinterpret_cast<uint2*>(hrow + len + (threadIdx.x + u * 256) * 4) = make_uint2(e.x, e.y);
            }
        __syncthreads();
    }
}
static inline int enc_grid(long n16) { long g = (n16 + 1023) / 1024; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

extern "C" int mu_prep_weight(const float* w_oihw, void* dst, int dtype, int O, int I, int taps, int rows_pad,
                              int cols_pad, int mode, void* stream) {
    if (!w_oihw || !dst || O <= 0 || I <= 0 || (taps != 1 && taps != 9)) return MU_ERR_ARG;
    if (mode < 0 || mode > 2) return MU_ERR_ARG;
    if (mode != 1 ? (rows_pad < O || cols_pad < I) : (rows_pad < I || cols_pad < O)) return MU_ERR_ARG;
    long n = (long)taps * rows_pad * cols_pad * (mode == 2 ? 2 : 1);
    int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32 || dtype == MU_F32X)
        prep_weight_kernel<float><<<grid, 256, 0, st>>>(w_oihw, (float*)dst, O, I, taps, rows_pad, cols_pad, mode);
    else if (dtype == MU_F16)
        prep_weight_kernel<h16><<<grid, 256, 0, st>>>(w_oihw, (h16*)dst, O, I, taps, rows_pad, cols_pad, mode);
    else
        return MU_ERR_ARG;
    if (dtype == MU_F32X) {
        if (rows_pad % 4 || cols_pad % 4) return MU_ERR_SHAPE;
        const long n1 = (long)taps * rows_pad * cols_pad;    // one block
        const float sh = (float)(1 << MU_XH_WSHIFT);
        float* d = (float*)dst;
        if (taps == 1) {
            split_encode_kernel<<<enc_grid(n / 4), 256, 0, st>>>((const f32x4*)d, (uint4*)d, n / 4);
        } else {
            if (mode != 1) split_encode_h4_kernel<<<enc_grid(n1 / 4), 256, 0, st>>>((const f32x4*)d, (uint4*)d, n1 / 4, sh);
            if (mode != 0) {
                // the data-gradient block's rows run over the layer's OUTPUT channels: rows_pad long in mode 2, cols_pad (as the caller
                // named the padded output count) in mode 1
                float* d1 = mode == 2 ? d + n1 : d;
                const int len = mode == 2 ? rows_pad : cols_pad;
                const long nrows = n1 / len;
                if (len > 8192) return MU_ERR_SHAPE;
                encode_hl_rows_kernel<<<(int)(nrows < 4096 ? nrows : 4096), 256, 0, st>>>(d1, nrows, len, sh);
            }
        }
    }
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// The same re-layout for MANY layers in one launch (a training forward re-lays all of a model's conv weights: 33 launches of 5-25 us
// each otherwise).  `jobs` is a device array of MU_PREP_JOB_FIELDS int64 per layer:
//   { w (address of the OIHW fp32 weight), dst_off (elements from dst_base), first_tile, O, I, taps, rows_pad, cols_pad, mode, 0 }
// A block moves one 32 (out) x 32 (in) x taps tile through LDS: the OIHW rows are read as contiguous 32*taps-float runs, both layouts
// are written as 16-byte vectors of 32-element runs (the per-layer kernel above gathers with a stride of `taps` floats and stores single
// elements: 156 us for the UNet's 36 layers when simply batched into one launch, 120 us tiled with 2-byte stores, 77 us with 16-byte
// stores).  Tiles [first_tile, first_tile + rows_pad/32 * cols_pad/32) belong to the layer.
#define MU_PREP_JOB_FIELDS 10
#define MU_PREP_MAX_JOBS 128
// ENC (T = float, MU_F32X): the blocks are written in their operand encodings directly (see mu_prep_weight): no separate encoding pass.
template <typename T, bool ENC = false>
__global__ __launch_bounds__(256) void prep_weights_multi_kernel(const long* __restrict__ jobs, int njobs, long ntiles, T* __restrict__ base) {
    constexpr int TS = 32, MAXT = 9, LD = TS * MAXT + 1;      // LDS row of one output channel: [in][tap], padded to an odd length
    __shared__ long sj[MU_PREP_MAX_JOBS * MU_PREP_JOB_FIELDS];
    __shared__ float tile[TS * LD];
    for (int i = threadIdx.x; i < njobs * MU_PREP_JOB_FIELDS; i += blockDim.x) sj[i] = jobs[i];
    __syncthreads();
    for (long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {                                      // the last job whose first tile is <= tl
            const int mid = (lo + hi + 1) >> 1;
            if (sj[mid * MU_PREP_JOB_FIELDS + 2] <= tl) lo = mid; else hi = mid - 1;
        }
        const long* J = sj + lo * MU_PREP_JOB_FIELDS;
        const float* w = reinterpret_cast<const float*>(J[0]);
        T* dst = base + J[1];
        const int O = (int)J[3], I = (int)J[4], taps = (int)J[5], rows_pad = (int)J[6], cols_pad = (int)J[7], mode = (int)J[8];
        const int tcols = cols_pad / TS;
        const int rel = (int)(tl - J[2]);
        const int o0 = (rel / tcols) * TS, i0 = (rel % tcols) * TS;
        const int run = TS * taps;                             // floats of one output channel's 32 input channels: contiguous in OIHW
#pragma unroll 4
        for (int k = threadIdx.x; k < TS * run; k += 256) {
            const int ol = k / run, rem = k - ol * run;
            const int o = o0 + ol, i = i0 + rem / taps;
            tile[ol * LD + rem] = (o < O && i < I) ? w[((long)o * I + i0) * taps + rem] : 0.f;
        }
        __syncthreads();
        const long n = (long)taps * rows_pad * cols_pad;
        constexpr int VN = 16 / (int)sizeof(T), VPR = TS / VN;  // 16-byte stores: VN elements, VPR vectors per 32-element run
        if (mode != 1)                                         // forward block [tap][out][in]
            for (int k = threadIdx.x; k < taps * TS * VPR; k += 256) {
                const int iv = k % VPR, ol = (k / VPR) & 31, t = k / (VPR * TS);
                Vec16<T> v;
#pragma unroll
                for (int e = 0; e < VN; ++e) v.set(e, tile[ol * LD + (iv * VN + e) * taps + t]);
                T* q = dst + ((long)t * rows_pad + o0 + ol) * cols_pad + i0 + iv * VN;
                if constexpr (ENC) {
                    const f32x4 f = {v.get(0), v.get(1), v.get(2), v.get(3)};
                    *reinterpret_cast<uint4*>(q) = taps == 1 ? mu_enc4(f) : mu_ench4(f * (float)(1 << MU_XH_WSHIFT));
                } else v.store(q);
            }
        if (mode != 0) {                                       // data-gradient block [taps-1-tap][in][out]
            T* d1 = mode == 2 ? dst + n : dst;
            for (int k = threadIdx.x; k < taps * TS * VPR; k += 256) {
                const int ov = k % VPR, il = (k / VPR) & 31, t = k / (VPR * TS);
                Vec16<T> v;
#pragma unroll
                for (int e = 0; e < VN; ++e) v.set(e, tile[(ov * VN + e) * LD + il * taps + t]);
                T* q = d1 + ((long)(taps - 1 - t) * cols_pad + i0 + il) * rows_pad + o0 + ov * VN;
                if constexpr (ENC) {
                    const f32x4 f = {v.get(0), v.get(1), v.get(2), v.get(3)};
                    if (taps == 1) *reinterpret_cast<uint4*>(q) = mu_enc4(f);
                    else {                                     // HL row: [rows_pad fp16 lo | rows_pad fp16 hi]
                        const uint4 e4 = mu_ench4(f * (float)(1 << MU_XH_WSHIFT));
                        h16* hrow = reinterpret_cast<h16*>(d1 + ((long)(taps - 1 - t) * cols_pad + i0 + il) * rows_pad);
                        *reinterpret_cast<uint2*>(hrow + o0 + ov * VN) = make_uint2(e4.z, e4.w);
                        *reinterpret_cast<uint2*>(hrow + rows_pad + o0 + ov * VN) = make_uint2(e4.x, e4.y);
                    }
                } else v.store(q);
            }
        }
        __syncthreads();
    }
}

extern "C" int mu_prep_weights_multi(const void* jobs, int njobs, long ntiles, void* dst_base, int dtype, void* stream) {
    if (!jobs || !dst_base || njobs <= 0 || njobs > MU_PREP_MAX_JOBS || ntiles <= 0) return MU_ERR_ARG;
    const int grid = (int)(ntiles < 4096 ? ntiles : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32) prep_weights_multi_kernel<float><<<grid, 256, 0, st>>>((const long*)jobs, njobs, ntiles, (float*)dst_base);
    else if (dtype == MU_F32X) prep_weights_multi_kernel<float, true><<<grid, 256, 0, st>>>((const long*)jobs, njobs, ntiles, (float*)dst_base);
    else if (dtype == MU_F16) prep_weights_multi_kernel<h16><<<grid, 256, 0, st>>>((const long*)jobs, njobs, ntiles, (h16*)dst_base);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// generic elementwise cast (fp32 parameter vectors -> compute dtype)
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = (TD)(float)s[i];
}
extern "C" int mu_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream) {
    if (!src || !dst || n <= 0) return MU_ERR_ARG;
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == MU_F32 && dst_dtype == MU_F16) cast_kernel<float, h16><<<grid, 256, 0, st>>>((const float*)src, (h16*)dst, n);
    else if (src_dtype == MU_F16 && dst_dtype == MU_F32) cast_kernel<h16, float><<<grid, 256, 0, st>>>((const h16*)src, (float*)dst, n);
    else if (src_dtype == MU_F32 && dst_dtype == MU_F32) cast_kernel<float, float><<<grid, 256, 0, st>>>((const float*)src, (float*)dst, n);
    else if (src_dtype == MU_F16 && dst_dtype == MU_F16) cast_kernel<h16, h16><<<grid, 256, 0, st>>>((const h16*)src, (h16*)dst, n);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// fp32x operand encoding (common.h): n16 aligned 16-byte chunks of four fp32 values -> [4 x bf16 hi | 4 x bf16 lo].  In place or
// out of place; contiguous rows only (a chunk never straddles a row: channel counts are multiples of 32).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_encode_kernel(const f32x4* src, uint4* dst, long n16) {      // (may alias: no __restrict__)
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) dst[i + u * stride] = mu_enc4(v[u]);
    }
}
// the fp16-pair chunk encoding of the 3x3-convolution operands (common.h mu_ench4); scale: a power of two (1 for activations)
__global__ __launch_bounds__(256) void split_encode_h4_kernel(const f32x4* src, uint4* dst, long n16, float scale) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) dst[i + u * stride] = mu_ench4(v[u] * scale);
    }
}
// the same with the fp16 rounding of the values (the hi halves) as a second, plain output: dst16 = n_elems halves
__global__ __launch_bounds__(256) void split_encode_h4x_kernel(const f32x4* __restrict__ src, uint4* __restrict__ dst, uint2* __restrict__ dst16, long n16) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) {
                const uint4 e = mu_ench4(v[u]);
                dst[i + u * stride] = e;
                dst16[i + u * stride] = make_uint2(e.x, e.y);
            }
    }
}
extern "C" int mu_split_encode_h4x(const void* src, void* dst, void* dst16, long n_elems, void* stream) {
    if (!src || !dst || !dst16 || src == dst || n_elems <= 0 || n_elems % 4) return MU_ERR_ARG;
    const long n16 = n_elems / 4;
    split_encode_h4x_kernel<<<enc_grid(n16), 256, 0, (hipStream_t)stream>>>((const f32x4*)src, (uint4*)dst, (uint2*)dst16, n16);
    MU_CHECK_LAUNCH();
    return MU_OK;
}
extern "C" int mu_split_encode_h4(const void* src, void* dst, long n_elems, void* stream) {
    if (!src || !dst || n_elems <= 0 || n_elems % 4) return MU_ERR_ARG;
    const long n16 = n_elems / 4;
    split_encode_h4_kernel<<<enc_grid(n16), 256, 0, (hipStream_t)stream>>>((const f32x4*)src, (uint4*)dst, n16, 1.0f);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// A plain fp32 gradient -> ONE power-of-two-scaled fp16 operand + its scale {S, 1 / S} (the form mu_bn_act_bwd_h writes; for a 3x3
// convolution whose dy does not come from a BatchNorm backward).  Two launches, no atomics, no host sync: per-block maxima, then every
// block reduces all of them (<= 1024 values) and converts its share.  S max|dy| in [2^13, 2^14).
#define MU_DYH_MAXBLK 1024
__global__ __launch_bounds__(256) void dyh_max_kernel(const f32x4* __restrict__ src, long n16, float* __restrict__ pmax) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const f32x4 v = src[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    m = wave_max(m);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) pmax[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__global__ __launch_bounds__(256) void dyh_convert_kernel(const f32x4* __restrict__ src, long n16, const float* __restrict__ pmax, int nmax,
                                                          h16x4* __restrict__ dst, float* __restrict__ dy_scale) {
    float m = 0.f;
    for (int k = threadIdx.x; k < nmax; k += 256) m = fmaxf(m, pmax[k]);
    m = wave_max(m);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    const int e = (int)((__float_as_uint(m) >> 23) & 0xff) - 127;
    float S = 1.f;
    if (m > 0.f && e > -127 && e < 128) {
        int k = 13 - e;
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
        S = __uint_as_float((uint32_t)(127 + k) << 23);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { dy_scale[0] = S; dy_scale[1] = 1.0f / S; }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const f32x4 v = src[i] * S;
        dst[i] = (h16x4){(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
    }
}
extern "C" long mu_dy_encode_h_workspace_bytes(void) { return (long)MU_DYH_MAXBLK * sizeof(float); }
extern "C" int mu_dy_encode_h(const void* dy, void* dy_h, float* dy_scale, long n_elems, void* workspace, long ws_bytes, void* stream) {
    if (!dy || !dy_h || !dy_scale || !workspace || n_elems <= 0 || n_elems % 4 || dy == dy_h) return MU_ERR_ARG;
    if (ws_bytes < mu_dy_encode_h_workspace_bytes()) return MU_ERR_WORKSPACE;
    const long n16 = n_elems / 4;
    long g = (n16 + 1023) / 1024;
    const int nb = (int)(g < 1 ? 1 : (g > MU_DYH_MAXBLK ? MU_DYH_MAXBLK : g));
    hipStream_t st = (hipStream_t)stream;
    dyh_max_kernel<<<nb, 256, 0, st>>>((const f32x4*)dy, n16, (float*)workspace);
    dyh_convert_kernel<<<enc_grid(n16), 256, 0, st>>>((const f32x4*)dy, n16, (const float*)workspace, nb, (h16x4*)dy_h, dy_scale);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_split_encode(const void* src, void* dst, long n_elems, void* stream) {
    if (!src || !dst || n_elems <= 0 || n_elems % 4) return MU_ERR_ARG;
    const long n16 = n_elems / 4;
    long g = (n16 + 1023) / 1024;
    g = g < 1 ? 1 : (g > 8192 ? 8192 : g);
    split_encode_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>((const f32x4*)src, (uint4*)dst, n16);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// MaxPool2d(2), NHWC.  Backward recomputes the arg-max from x (first maximum in (kh,kw) scan
// order, the aten tie rule) so no index tensor is stored; windows are disjoint -> no atomics.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    constexpr int N = Vec16<T>::N;
    const int cv = C / N, Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo * cv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int c = (idx % cv) * N;
        long p = idx / cv;
        int wo = p % Wo, ho = (p / Wo) % Ho, b = p / ((long)Wo * Ho);
        const T* base = x + (((long)b * H + 2 * ho) * W + 2 * wo) * C + c;
        Vec16<T> v00, v01, v10, v11, o;
        EW_LD(v00, base); EW_LD(v01, base + C); EW_LD(v10, base + (long)W * C); EW_LD(v11, base + (long)W * C + C);
#pragma unroll
        for (int i = 0; i < N; ++i) o.set(i, fmaxf(fmaxf(v00.get(i), v01.get(i)), fmaxf(v10.get(i), v11.get(i))));
        o.store(y + p * C + c);
    }
}

// dy2 (optional, pooled resolution): a second gradient of the pooled output, summed on the fly -- the residual branch of the ConvBlock
// behind the pool (ade_semantic.py:208: gelu(x + block(x)) makes x a consumer twice).  dx_add (optional, input resolution): a gradient
// the pooled tensor's INPUT received from elsewhere (the skip connection into UpSample, :253), added to the scattered result.
// Both replace an elementwise autograd accumulation kernel (read 2 + write 1 tensors) by one extra read here.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ dy2,
                                                          const T* __restrict__ dx_add, T* __restrict__ dx, int B, int H, int W, int C) {
    constexpr int N = Vec16<T>::N;
    const int cv = C / N, Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo * cv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int c = (idx % cv) * N;
        long p = idx / cv;
        int wo = p % Wo, ho = (p / Wo) % Ho, b = p / ((long)Wo * Ho);
        long off = (((long)b * H + 2 * ho) * W + 2 * wo) * C + c;
        const long offs[4] = {off, off + C, off + (long)W * C, off + (long)W * C + C};
        Vec16<T> v[4], g, g2, a[4], o[4];
        EW_LD(v[0], x + offs[0]); EW_LD(v[1], x + offs[1]); EW_LD(v[2], x + offs[2]); EW_LD(v[3], x + offs[3]);
        EW_LD(g, dy + p * C + c);
        if (dy2) EW_LD(g2, dy2 + p * C + c);
        if (dx_add) { EW_LD(a[0], dx_add + offs[0]); EW_LD(a[1], dx_add + offs[1]); EW_LD(a[2], dx_add + offs[2]); EW_LD(a[3], dx_add + offs[3]); }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            int best = 0;
            float m = v[0].get(i);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                float t = v[k].get(i);
                if (t > m) { m = t; best = k; }
            }
            float gi = g.get(i);
            if (dy2) gi += g2.get(i);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k].set(i, (k == best ? gi : 0.f) + (dx_add ? a[k].get(i) : 0.f));
        }
        o[0].store(dx + offs[0]); o[1].store(dx + offs[1]); o[2].store(dx + offs[2]); o[3].store(dx + offs[3]);
    }
}

static inline int ew_grid(long total) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// nn.MaxPool2d(2) floors odd sizes (ade_semantic.py:216): the last row / column of an odd H / W belongs to no window, so its
// gradient is zero (plus dx_add where the tensor has a second consumer).  One thread per 16-byte vector of the uncovered pixels.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_tail_kernel(const T* __restrict__ dx_add, T* __restrict__ dx, int B, int H, int W, int C) {
    constexpr int N = Vec16<T>::N;
    const int cv = C / N, He = H & ~1, We = W & ~1;
    const int ntail = (H - He) * W + He * (W - We);            // uncovered pixels per image
    const long total = (long)B * ntail * cv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (idx % cv) * N;
        const long t = idx / cv;
        const int k = (int)(t % ntail), b = (int)(t / ntail);
        int h, w;
        if (k < (H - He) * W) { h = He; w = k; }               // the odd last row
        else { h = k - (H - He) * W; w = We; }                 // the odd last column of the covered rows
        const long off = (((long)b * H + h) * W + w) * C + c;
        Vec16<T> o;
        if (dx_add) o.load(dx_add + off); else o.zero();
        o.store(dx + off);
    }
}

extern "C" int mu_maxpool2_fwd(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    if (!x || !y || B <= 0 || H < 2 || W < 2 || C % 8) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32) {
        long total = (long)B * (H / 2) * (W / 2) * (C / 4);
        maxpool_fwd_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)x, (float*)y, B, H, W, C);
    } else if (dtype == MU_F16) {
        long total = (long)B * (H / 2) * (W / 2) * (C / 8);
        maxpool_fwd_kernel<h16><<<ew_grid(total), 256, 0, st>>>((const h16*)x, (h16*)y, B, H, W, C);
    } else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_maxpool2_bwd_acc(const void* x, const void* dy, const void* dy2, const void* dx_add, void* dx, int B, int H, int W, int C,
                                   int dtype, void* stream) {
    if (!x || !dy || !dx || B <= 0 || H < 2 || W < 2 || C % 8) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long tail = (long)B * ((H & 1) * W + (H & ~1) * (W & 1));
    if (dtype == MU_F32) {
        long total = (long)B * (H / 2) * (W / 2) * (C / 4);
        maxpool_bwd_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)x, (const float*)dy, (const float*)dy2, (const float*)dx_add,
                                                                  (float*)dx, B, H, W, C);
        if (tail) maxpool_tail_kernel<float><<<ew_grid(tail * (C / 4)), 256, 0, st>>>((const float*)dx_add, (float*)dx, B, H, W, C);
    } else if (dtype == MU_F16) {
        long total = (long)B * (H / 2) * (W / 2) * (C / 8);
        maxpool_bwd_kernel<h16><<<ew_grid(total), 256, 0, st>>>((const h16*)x, (const h16*)dy, (const h16*)dy2, (const h16*)dx_add, (h16*)dx,
                                                                B, H, W, C);
        if (tail) maxpool_tail_kernel<h16><<<ew_grid(tail * (C / 8)), 256, 0, st>>>((const h16*)dx_add, (h16*)dx, B, H, W, C);
    } else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_maxpool2_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
    return mu_maxpool2_bwd_acc(x, dy, nullptr, nullptr, dx, B, H, W, C, dtype, stream);
}

// ------------------------------------------------------------------------------------------
// bilinear x2 upsample (align_corners=True) fused with the channel concat [skip, up]
//   y[b, ho, wo, 0:Cs]      = skip[b, ho, wo, :]
//   y[b, ho, wo, Cs:Cs+Cx]  = lerp of x[b, h0/h1, w0/w1, :]
// src = dst * (in-1)/(out-1)  (aten area_pixel_compute_source_index, align_corners branch)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lerp_axis(int d, float scale, int n_in, int& i0, int& i1, float& f) {
#pragma clang fp contract(off)                                // rounded product, rounded difference: no FMA contraction, so that every
    float s = scale * (float)d;                               // kernel of this op (and the CPU reference) sees the same weight
    i0 = (int)s;                                              // (__fmul_rn / __fsub_rn are plain operators in HIP: they contract)
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    f = s - (float)i0;
}

__device__ __forceinline__ float lerp2(float a00, float a01, float a10, float a11, float fh, float fw) {
#pragma clang fp contract(off)
    const float r0 = a00 * (1.f - fh) + a10 * fh;
    const float r1 = a01 * (1.f - fh) + a11 * fh;
    return r0 * (1.f - fw) + r1 * fw;
}

template <typename T>
__global__ __launch_bounds__(256) void upcat_fwd_kernel(const T* __restrict__ x, const T* __restrict__ skip, T* __restrict__ y,
                                                        int B, int h, int w, int Cx, int Cs) {
    constexpr int N = Vec16<T>::N;
    const int Ho = 2 * h, Wo = 2 * w, Ct = Cx + Cs, cv = Ct / N;
    const float sh = h > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sw = w > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
    const long total = (long)B * Ho * Wo * cv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int c = (idx % cv) * N;
        long p = idx / cv;
        Vec16<T> o;
        if (c < Cs) {
            o.load(skip + p * Cs + c);
        } else {
            int wo = p % Wo, ho = (p / Wo) % Ho, b = p / ((long)Wo * Ho);
            int h0, h1, w0, w1; float fh, fw;
            lerp_axis(ho, sh, h, h0, h1, fh);
            lerp_axis(wo, sw, w, w0, w1, fw);
            const T* xb = x + (long)b * h * w * Cx + (c - Cs);
            Vec16<T> a00, a01, a10, a11;
            a00.load(xb + ((long)h0 * w + w0) * Cx); a01.load(xb + ((long)h0 * w + w1) * Cx);
            a10.load(xb + ((long)h1 * w + w0) * Cx); a11.load(xb + ((long)h1 * w + w1) * Cx);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                // same association as the oracle: rows first (over h), then columns; products and sums rounded separately (no FMA
                // contraction), as the reference's CPU kernel does -- and so that every kernel of this op gives the same bits
                o.set(i, lerp2(a00.get(i), a01.get(i), a10.get(i), a11.get(i), fh, fw));
            }
        }
        o.store(y + p * Ct + c);
    }
}

// backward: dskip = dy[..., :Cs];  dx = bilinear^T(dy[..., Cs:]) as a deterministic gather
// dy2 (optional): a second gradient of the concatenated tensor, summed on the fly (the residual branch of the ConvBlock behind the
// concat, ade_semantic.py:208,237-238) instead of an autograd accumulation kernel in front of this one.
template <typename T>
__global__ __launch_bounds__(256) void upcat_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, T* __restrict__ dx,
                                                        T* __restrict__ dskip, int B, int h, int w, int Cx, int Cs) {
    constexpr int N = Vec16<T>::N;
    const int Ho = 2 * h, Wo = 2 * w, Ct = Cx + Cs;
    const float sh = h > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sw = w > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
    const int cvs = Cs / N, cvx = Cx / N;
    const long n_skip = (long)B * Ho * Wo * cvs, n_x = (long)B * h * w * cvx;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n_skip + n_x; idx += (long)gridDim.x * 256) {
        if (idx < n_skip) {
            int c = (idx % cvs) * N;
            long p = idx / cvs;
            Vec16<T> v;
            v.load(dy + p * Ct + c);
            if (dy2) {
                Vec16<T> v2;
                v2.load(dy2 + p * Ct + c);
#pragma unroll
                for (int i = 0; i < N; ++i) v.set(i, v.get(i) + v2.get(i));
            }
            v.store(dskip + p * Cs + c);
            continue;
        }
        long j = idx - n_skip;
        int c = (j % cvx) * N;
        long p = j / cvx;
        int wi = p % w, hi = (p / w) % h, b = p / ((long)w * h);
        float acc[N];
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = 0.f;
        // destination rows whose source interval touches hi lie within [2hi-2, 2hi+3]
        for (int ho = max(0, 2 * hi - 2); ho <= min(Ho - 1, 2 * hi + 3); ++ho) {
            int h0, h1; float fh;
            lerp_axis(ho, sh, h, h0, h1, fh);
            float wh = (h0 == hi ? 1.f - fh : 0.f) + (h1 == hi ? fh : 0.f);
            if (h0 == hi && h1 == hi) wh = 1.f;
            if (wh == 0.f) continue;
            for (int wo = max(0, 2 * wi - 2); wo <= min(Wo - 1, 2 * wi + 3); ++wo) {
                int w0, w1; float fw;
                lerp_axis(wo, sw, w, w0, w1, fw);
                float ww = (w0 == wi ? 1.f - fw : 0.f) + (w1 == wi ? fw : 0.f);
                if (w0 == wi && w1 == wi) ww = 1.f;
                if (ww == 0.f) continue;
                Vec16<T> g;
                const long goff = (((long)b * Ho + ho) * Wo + wo) * Ct + Cs + c;
                g.load(dy + goff);
                float wt = wh * ww;
                if (dy2) {
                    Vec16<T> g2;
                    g2.load(dy2 + goff);
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[i] += wt * (g.get(i) + g2.get(i));
                } else {
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[i] += wt * g.get(i);
                }
            }
        }
        Vec16<T> o;
#pragma unroll
        for (int i = 0; i < N; ++i) o.set(i, acc[i]);
        o.store(dx + p * Cx + c);
    }
}

// Row-per-block forms of the two kernels above (the UNet's shapes: 256 % (channels / vector) == 0).  The element-per-thread forms pay
// three 64-bit divisions per 16-byte vector and keep ONE load per lane in flight (2.4 / 1.9 TB/s on the 128 x 128 decoder level, the
// gradient gather walking its 6 x 6 candidate window with a divergent `continue` per tap).  Here a block owns one output row (forward, skip
// gradient) or one input row (upsample gradient): the row interpolation is block-uniform, a lane's channel vector is fixed, pixels advance
// by a constant, and every load of an iteration is issued before the first use.  Same arithmetic per element -> bit-identical results.
#ifndef MU_UPCAT_ROWS
#define MU_UPCAT_ROWS 1
#endif
template <typename T, int U>
__global__ __launch_bounds__(256) void upcat_fwd_rows_kernel(const T* __restrict__ x, const T* __restrict__ skip, T* __restrict__ y,
                                                             int B, int h, int w, int Cx, int Cs) {
    constexpr int N = Vec16<T>::N;
    const int Ho = 2 * h, Wo = 2 * w, Ct = Cx + Cs, cv = Ct / N;
    const float sh = h > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sw = w > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
    const int ppi = 256 / cv;                                  // pixels per block-iteration
    const int vi = threadIdx.x % cv, pl = threadIdx.x / cv, c = vi * N;
    for (int row = blockIdx.x; row < B * Ho; row += gridDim.x) {
        const int ho = row % Ho, b = row / Ho;
        int h0, h1; float fh;
        lerp_axis(ho, sh, h, h0, h1, fh);
        const T* srow = skip + (long)row * Wo * Cs + c;
        const T* x0 = x + ((long)b * h + h0) * w * Cx + (c - Cs);
        const T* x1 = x + ((long)b * h + h1) * w * Cx + (c - Cs);
        T* yrow = y + (long)row * Wo * Ct + c;
        for (int wb = pl; wb < Wo; wb += U * ppi) {
            if (c < Cs) {
                Vec16<T> v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) if (wb + u * ppi < Wo) EW_LD(v[u], srow + (long)(wb + u * ppi) * Cs);
#pragma unroll
                for (int u = 0; u < U; ++u) if (wb + u * ppi < Wo) v[u].store(yrow + (long)(wb + u * ppi) * Ct);
            } else {
                Vec16<T> a00[U], a01[U], a10[U], a11[U];
                float fw[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int wo = wb + u * ppi;
                    if (wo < Wo) {
                        int w0, w1;
                        lerp_axis(wo, sw, w, w0, w1, fw[u]);
                        a00[u].load(x0 + (long)w0 * Cx); a01[u].load(x0 + (long)w1 * Cx);
                        a10[u].load(x1 + (long)w0 * Cx); a11[u].load(x1 + (long)w1 * Cx);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int wo = wb + u * ppi;
                    if (wo < Wo) {
                        Vec16<T> o;
#pragma unroll
                        for (int i = 0; i < N; ++i) {
                            o.set(i, lerp2(a00[u].get(i), a01[u].get(i), a10[u].get(i), a11[u].get(i), fh, fw[u]));
                        }
                        o.store(yrow + (long)wo * Ct);
                    }
                }
            }
        }
    }
}

// blocks [0, B*Ho): one output row of dskip each; blocks [B*Ho, B*Ho + B*h): one input row of dx each
template <typename T>
__global__ __launch_bounds__(256) void upcat_bwd_rows_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, T* __restrict__ dx,
                                                             T* __restrict__ dskip, int B, int h, int w, int Cx, int Cs) {
    constexpr int N = Vec16<T>::N;
    const int Ho = 2 * h, Wo = 2 * w, Ct = Cx + Cs;
    const float sh = h > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sw = w > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
    if ((int)blockIdx.x < B * Ho) {
        constexpr int U = 4;
        const int cvs = Cs / N, ppi = 256 / cvs;
        const int c = (threadIdx.x % cvs) * N, pl = threadIdx.x / cvs;
        const long row = blockIdx.x;
        const T* g1 = dy + row * Wo * Ct + c;
        const T* g2 = dy2 ? dy2 + row * Wo * Ct + c : nullptr;
        T* o = dskip + row * Wo * Cs + c;
        for (int wb = pl; wb < Wo; wb += U * ppi) {
            Vec16<T> v[U], v2[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (wb + u * ppi < Wo) {
                    EW_LD(v[u], g1 + (long)(wb + u * ppi) * Ct);
                    if (g2) EW_LD(v2[u], g2 + (long)(wb + u * ppi) * Ct);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (wb + u * ppi < Wo) {
                    if (g2) {
#pragma unroll
                        for (int i = 0; i < N; ++i) v[u].set(i, v[u].get(i) + v2[u].get(i));
                    }
                    v[u].store(o + (long)(wb + u * ppi) * Cs);
                }
        }
        return;
    }
    const int rowx = blockIdx.x - B * Ho;
    const int hi = rowx % h, b = rowx / h;
    const int cvx = Cx / N, ppi = 256 / cvx;
    const int c = (threadIdx.x % cvx) * N, pl = threadIdx.x / cvx;
    // destination rows whose source interval touches hi lie within [2hi-2, 2hi+3]: their weights are block-uniform
    float wh[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int ho = 2 * hi - 2 + k;
        wh[k] = 0.f;
        if (ho >= 0 && ho < Ho) {
            int h0, h1; float fh;
            lerp_axis(ho, sh, h, h0, h1, fh);
            wh[k] = (h0 == hi ? 1.f - fh : 0.f) + (h1 == hi ? fh : 0.f);
            if (h0 == hi && h1 == hi) wh[k] = 1.f;
        }
    }
    for (int wi = pl; wi < w; wi += ppi) {
        float ww[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int wo = 2 * wi - 2 + k;
            ww[k] = 0.f;
            if (wo >= 0 && wo < Wo) {
                int w0, w1; float fw;
                lerp_axis(wo, sw, w, w0, w1, fw);
                ww[k] = (w0 == wi ? 1.f - fw : 0.f) + (w1 == wi ? fw : 0.f);
                if (w0 == wi && w1 == wi) ww[k] = 1.f;
            }
        }
        float acc[N];
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = 0.f;
        for (int kr = 0; kr < 6; ++kr) {                      // (uniform) rows in ascending order, columns ascending: the order of the loops above
            if (wh[kr] == 0.f) continue;
            const int ho = 2 * hi - 2 + kr;
            const long rbase = (((long)b * Ho + ho) * Wo) * Ct + Cs + c;
            Vec16<T> g[6], g2v[6];
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (ww[k] != 0.f) {
                    g[k].load(dy + rbase + (long)(2 * wi - 2 + k) * Ct);
                    if (dy2) g2v[k].load(dy2 + rbase + (long)(2 * wi - 2 + k) * Ct);
                }
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (ww[k] != 0.f) {
                    const float wt = wh[kr] * ww[k];
                    if (dy2) {
#pragma unroll
                        for (int i = 0; i < N; ++i) acc[i] += wt * (g[k].get(i) + g2v[k].get(i));
                    } else {
#pragma unroll
                        for (int i = 0; i < N; ++i) acc[i] += wt * g[k].get(i);
                    }
                }
        }
        Vec16<T> o;
#pragma unroll
        for (int i = 0; i < N; ++i) o.set(i, acc[i]);
        o.store(dx + (((long)b * h + hi) * w + wi) * Cx + c);
    }
}

// Compacting variants for channel counts that are not multiples of the 32-channel padding (stand-alone UpSample with any
// channel counts, ade_semantic.py:231-256): x has Cxv valid of Cxl stored channels, skip Csv of Csl; the output row is
// [skip valid | up valid | zeros] with Ctl stored channels.  One element per thread (not a hot path: inside the UNet every
// count is a multiple of 32 and the vector kernels above run).
template <typename T>
__global__ __launch_bounds__(256) void upcat_compact_fwd_kernel(const T* __restrict__ x, const T* __restrict__ skip, T* __restrict__ y,
                                                                int B, int h, int w, int Cxl, int Cxv, int Csl, int Csv, int Ctl) {
    const int Ho = 2 * h, Wo = 2 * w;
    const float sh = h > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sw = w > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
    const long total = (long)B * Ho * Wo * Ctl;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int c = idx % Ctl;
        long p = idx / Ctl;
        float o = 0.f;
        if (c < Csv) {
            o = (float)skip[p * Csl + c];
        } else if (c < Csv + Cxv) {
            int wo = p % Wo, ho = (p / Wo) % Ho, b = p / ((long)Wo * Ho);
            int h0, h1, w0, w1; float fh, fw;
            lerp_axis(ho, sh, h, h0, h1, fh);
            lerp_axis(wo, sw, w, w0, w1, fw);
            const T* xb = x + (long)b * h * w * Cxl + (c - Csv);
            float a00 = (float)xb[((long)h0 * w + w0) * Cxl], a01 = (float)xb[((long)h0 * w + w1) * Cxl];
            float a10 = (float)xb[((long)h1 * w + w0) * Cxl], a11 = (float)xb[((long)h1 * w + w1) * Cxl];
            o = lerp2(a00, a01, a10, a11, fh, fw);
        }
        y[idx] = (T)o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upcat_compact_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, T* __restrict__ dskip,
                                                                int B, int h, int w, int Cxl, int Cxv, int Csl, int Csv, int Ctl) {
    const int Ho = 2 * h, Wo = 2 * w;
    const float sh = h > 1 ? (float)(h - 1) / (float)(Ho - 1) : 0.f, sw = w > 1 ? (float)(w - 1) / (float)(Wo - 1) : 0.f;
    const long n_skip = (long)B * Ho * Wo * Csl, n_x = (long)B * h * w * Cxl;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n_skip + n_x; idx += (long)gridDim.x * 256) {
        if (idx < n_skip) {
            int c = idx % Csl;
            long p = idx / Csl;
            dskip[idx] = c < Csv ? dy[p * Ctl + c] : (T)0.f;
            continue;
        }
        long j = idx - n_skip;
        int c = j % Cxl;
        long p = j / Cxl;
        float acc = 0.f;
        if (c < Cxv) {
            int wi = p % w, hi = (p / w) % h, b = p / ((long)w * h);
            for (int ho = max(0, 2 * hi - 2); ho <= min(Ho - 1, 2 * hi + 3); ++ho) {
                int h0, h1; float fh;
                lerp_axis(ho, sh, h, h0, h1, fh);
                float wh = (h0 == hi ? 1.f - fh : 0.f) + (h1 == hi ? fh : 0.f);
                if (h0 == hi && h1 == hi) wh = 1.f;
                if (wh == 0.f) continue;
                for (int wo = max(0, 2 * wi - 2); wo <= min(Wo - 1, 2 * wi + 3); ++wo) {
                    int w0, w1; float fw;
                    lerp_axis(wo, sw, w, w0, w1, fw);
                    float ww = (w0 == wi ? 1.f - fw : 0.f) + (w1 == wi ? fw : 0.f);
                    if (w0 == wi && w1 == wi) ww = 1.f;
                    if (ww == 0.f) continue;
                    acc += wh * ww * (float)dy[(((long)b * Ho + ho) * Wo + wo) * Ctl + Csv + c];
                }
            }
        }
        dx[j] = (T)acc;
    }
}

extern "C" int mu_upcat_compact_fwd(const void* x, const void* skip, void* y, int B, int h, int w, int Cx_ld, int Cx_valid, int Cs_ld,
                                    int Cs_valid, int Ct_ld, int dtype, void* stream) {
    if (!x || !skip || !y || B <= 0 || h <= 0 || w <= 0) return MU_ERR_ARG;
    if (Cx_valid <= 0 || Cs_valid <= 0 || Cx_valid > Cx_ld || Cs_valid > Cs_ld || Cx_valid + Cs_valid > Ct_ld) return MU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    long total = (long)B * 4 * h * w * Ct_ld;
    if (dtype == MU_F32)
        upcat_compact_fwd_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)x, (const float*)skip, (float*)y, B, h, w, Cx_ld,
                                                                        Cx_valid, Cs_ld, Cs_valid, Ct_ld);
    else if (dtype == MU_F16)
        upcat_compact_fwd_kernel<h16><<<ew_grid(total), 256, 0, st>>>((const h16*)x, (const h16*)skip, (h16*)y, B, h, w, Cx_ld, Cx_valid,
                                                                      Cs_ld, Cs_valid, Ct_ld);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_upcat_compact_bwd(const void* dy, void* dx, void* dskip, int B, int h, int w, int Cx_ld, int Cx_valid, int Cs_ld,
                                    int Cs_valid, int Ct_ld, int dtype, void* stream) {
    if (!dy || !dx || !dskip || B <= 0 || h <= 0 || w <= 0) return MU_ERR_ARG;
    if (Cx_valid <= 0 || Cs_valid <= 0 || Cx_valid > Cx_ld || Cs_valid > Cs_ld || Cx_valid + Cs_valid > Ct_ld) return MU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    long total = (long)B * 4 * h * w * Cs_ld + (long)B * h * w * Cx_ld;
    if (dtype == MU_F32)
        upcat_compact_bwd_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)dy, (float*)dx, (float*)dskip, B, h, w, Cx_ld,
                                                                        Cx_valid, Cs_ld, Cs_valid, Ct_ld);
    else if (dtype == MU_F16)
        upcat_compact_bwd_kernel<h16><<<ew_grid(total), 256, 0, st>>>((const h16*)dy, (h16*)dx, (h16*)dskip, B, h, w, Cx_ld, Cx_valid,
                                                                      Cs_ld, Cs_valid, Ct_ld);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_upcat_fwd(const void* x, const void* skip, void* y, int B, int h, int w, int Cx, int Cs, int dtype, void* stream) {
    if (!x || !skip || !y || B <= 0 || h <= 0 || w <= 0 || Cx % 8 || Cs % 8) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32) {
        long total = (long)B * 4 * h * w * ((Cx + Cs) / 4);
        upcat_fwd_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)x, (const float*)skip, (float*)y, B, h, w, Cx, Cs);
    } else if (dtype == MU_F16) {
        long total = (long)B * 4 * h * w * ((Cx + Cs) / 8);
        const int cv = (Cx + Cs) / 8;
        if (MU_UPCAT_ROWS && cv <= 256 && 256 % cv == 0 && (long)B * 2 * h < (1 << 30))
            upcat_fwd_rows_kernel<h16, 4><<<B * 2 * h < 16384 ? B * 2 * h : 16384, 256, 0, st>>>((const h16*)x, (const h16*)skip, (h16*)y, B, h, w, Cx, Cs);
        else
            upcat_fwd_kernel<h16><<<ew_grid(total), 256, 0, st>>>((const h16*)x, (const h16*)skip, (h16*)y, B, h, w, Cx, Cs);
    } else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_upcat_bwd_acc(const void* dy, const void* dy2, void* dx, void* dskip, int B, int h, int w, int Cx, int Cs, int dtype,
                                void* stream) {
    if (!dy || !dx || !dskip || B <= 0 || h <= 0 || w <= 0 || Cx % 8 || Cs % 8) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32) {
        long total = (long)B * 4 * h * w * (Cs / 4) + (long)B * h * w * (Cx / 4);
        upcat_bwd_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)dy, (const float*)dy2, (float*)dx, (float*)dskip, B, h, w, Cx, Cs);
    } else if (dtype == MU_F16) {
        long total = (long)B * 4 * h * w * (Cs / 8) + (long)B * h * w * (Cx / 8);
        const int cvs = Cs / 8, cvx = Cx / 8;
        if (MU_UPCAT_ROWS && cvs <= 256 && 256 % cvs == 0 && cvx <= 256 && 256 % cvx == 0 && (long)B * 3 * h < (1 << 30))
            upcat_bwd_rows_kernel<h16><<<B * 3 * h, 256, 0, st>>>((const h16*)dy, (const h16*)dy2, (h16*)dx, (h16*)dskip, B, h, w, Cx, Cs);
        else
            upcat_bwd_kernel<h16><<<ew_grid(total), 256, 0, st>>>((const h16*)dy, (const h16*)dy2, (h16*)dx, (h16*)dskip, B, h, w, Cx, Cs);
    } else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_upcat_bwd(const void* dy, void* dx, void* dskip, int B, int h, int w, int Cx, int Cs, int dtype, void* stream) {
    return mu_upcat_bwd_acc(dy, nullptr, dx, dskip, B, h, w, Cx, Cs, dtype, stream);
}

// ------------------------------------------------------------------------------------------
// Dropout: y = x * keep / (1-p).  keep is either an explicit uint8 mask (parity tests inject
// the reference's captured mask) or drawn from a counter-based generator keyed on
// (seed, element index) so the backward regenerates it instead of storing it.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long nvec, float p, float scale,
                                                      uint64_t seed, const uint8_t* __restrict__ mask, uint8_t* __restrict__ mask_out,
                                                      const unsigned long long* __restrict__ seed_step) {
    constexpr int N = Vec16<T>::N;
    const uint32_t thr = (uint32_t)(p * 65536.0f);
    if (seed_step) seed ^= splitmix64(0x9E3779B97F4A7C15ull * (uint64_t)seed_step[0]);     // per-replay stream of a captured step
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        Vec16<T> a, o;
        EW_LD(a, x + v * N);
        uint64_t r0 = 0, r1 = 0;
        if (!mask) {
            r0 = splitmix64(seed ^ (uint64_t)(2 * v) * 0xD6E8FEB86659FD93ull);
            r1 = splitmix64(seed ^ (uint64_t)(2 * v + 1) * 0xD6E8FEB86659FD93ull);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            bool keep;
            if (mask) keep = mask[v * N + i] != 0;
            else {
                uint32_t r = (uint32_t)(((i < 4 ? r0 : r1) >> (16 * (i & 3))) & 0xFFFFu);
                keep = r >= thr;
            }
            if (mask_out) mask_out[v * N + i] = keep ? 1 : 0;
            o.set(i, keep ? a.get(i) * scale : 0.f);
        }
        o.store(y + v * N);
    }
}

extern "C" int mu_dropout_step(const void* x, void* y, long n, float p, unsigned long long seed, const unsigned long long* seed_step,
                               const unsigned char* mask, unsigned char* mask_out, int dtype, void* stream);

extern "C" int mu_dropout(const void* x, void* y, long n, float p, unsigned long long seed, const unsigned char* mask,
                          unsigned char* mask_out, int dtype, void* stream) {
    return mu_dropout_step(x, y, n, p, seed, nullptr, mask, mask_out, dtype, stream);
}

extern "C" int mu_dropout_step(const void* x, void* y, long n, float p, unsigned long long seed, const unsigned long long* seed_step,
                               const unsigned char* mask, unsigned char* mask_out, int dtype, void* stream) {
    if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float scale = 1.0f / (1.0f - p);
    if (dtype == MU_F32) {
        if (n % 4) return MU_ERR_SHAPE;
        dropout_kernel<float><<<ew_grid(n / 4), 256, 0, st>>>((const float*)x, (float*)y, n / 4, p, scale, seed, mask, mask_out, seed_step);
    } else if (dtype == MU_F16) {
        if (n % 8) return MU_ERR_SHAPE;
        dropout_kernel<h16><<<ew_grid(n / 8), 256, 0, st>>>((const h16*)x, (h16*)y, n / 8, p, scale, seed, mask, mask_out, seed_step);
    } else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// out = a + b (residual-branch gradient join of the attention block)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long nvec) {
    constexpr int N = Vec16<T>::N;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        Vec16<T> x, y, z;
        EW_LD(x, a + v * N); EW_LD(y, b + v * N);
#pragma unroll
        for (int i = 0; i < N; ++i) z.set(i, x.get(i) + y.get(i));
        z.store(o + v * N);
    }
}
extern "C" int mu_add(const void* a, const void* b, void* out, long n, int dtype, void* stream) {
    if (!a || !b || !out || n <= 0) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32) { if (n % 4) return MU_ERR_SHAPE; add_kernel<float><<<ew_grid(n / 4), 256, 0, st>>>((const float*)a, (const float*)b, (float*)out, n / 4); }
    else if (dtype == MU_F16) { if (n % 8) return MU_ERR_SHAPE; add_kernel<h16><<<ew_grid(n / 8), 256, 0, st>>>((const h16*)a, (const h16*)b, (h16*)out, n / 8); }
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// decoded image bytes -> network input.  uint8 HWC (what cv2.imread/cvtColor/resize leave in memory,
// ade_semantic.py:72-76) -> [0,1] float (ToTensor, :85) in the NHWC compute layout, channel-padded with zeros.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void u8_to_nhwc_kernel(const uint8_t* __restrict__ src, T* __restrict__ dst, long npix, int C, int Cp) {
    constexpr int N = Vec16<T>::N;
    const int cv = Cp / N;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < npix * cv; idx += (long)gridDim.x * 256) {
        const long p = idx / cv;
        const int c0 = (int)(idx % cv) * N;
        Vec16<T> o;
#pragma unroll
        for (int i = 0; i < N; ++i) o.set(i, c0 + i < C ? (float)src[p * C + c0 + i] / 255.0f : 0.f);      // IEEE division, as ToTensor's .div(255) on the host
        o.store(dst + p * Cp + c0);
    }
}
extern "C" int mu_u8_to_nhwc(const unsigned char* src, void* dst, long npix, int C, int Cp, int dtype, void* stream) {
    if (!src || !dst || npix <= 0 || C <= 0 || Cp < C || Cp % 8) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32) u8_to_nhwc_kernel<float><<<ew_grid(npix * (Cp / 4)), 256, 0, st>>>(src, (float*)dst, npix, C, Cp);
    else if (dtype == MU_F16) u8_to_nhwc_kernel<h16><<<ew_grid(npix * (Cp / 8)), 256, 0, st>>>(src, (h16*)dst, npix, C, Cp);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// Key-mask compaction.  Mask2FormerAttention draws keep = randint(0,2,(B,H,W)) and adds {0,-inf} per KEY (ade_semantic.py:177-183);
// the attention kernels iterate over the kept keys only, through an index list.  kidx[b] = the kept keys in ascending order followed
// by the masked keys in ascending order -- a whole permutation of 0..N-1, exactly torch.argsort(keep, descending=True, stable=True)
// (the MU_ATTN_KIDX_PERMUTATION promise) -- and kcnt[b] = the number kept.  One 1024-thread block per image: every thread owns a
// contiguous run of keys, counts its kept ones, a block-wide exclusive scan places both partitions.  `keep` is uint8 or int64 (what
// torch.randint returns), non-zero = visible; keep8 (optional) receives the {0,1} bytes.  Under mask_mode="resample" (the reference's
// behaviour under multi-GPU nn.DataParallel) this runs six times per step in place of six torch.argsort calls.
// ------------------------------------------------------------------------------------------
template <typename KT>
__global__ __launch_bounds__(1024) void compact_keys_kernel(const KT* __restrict__ keep, int N, int* __restrict__ kidx, int* __restrict__ kcnt,
                                                            uint8_t* __restrict__ keep8) {
    __shared__ int wsum[16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const KT* kp = keep + (long)b * N;
    const int per = (N + 1023) / 1024;
    const long s0 = (long)t * per;
    const int i0 = (int)(s0 < N ? s0 : N), i1 = i0 + per < N ? i0 + per : N;
    int c = 0;
    for (int i = i0; i < i1; ++i) c += kp[i] != 0 ? 1 : 0;
    int s = c;                                               // inclusive scan over the wave, then over the 16 waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(s, o);
        if (lane >= o) s += v;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int v = wsum[w];
        if (w < wave) base += v;
        total += v;
    }
    int kpos = base + s - c;                                 // kept keys in front of this thread's run
    int mpos = total + (i0 - kpos);                          // masked keys in front of it, behind all the kept ones
    int* row = kidx + (long)b * N;
    for (int i = i0; i < i1; ++i) {
        const bool f = kp[i] != 0;
        if (f) row[kpos++] = i; else row[mpos++] = i;
        if (keep8) keep8[(long)b * N + i] = f ? 1 : 0;
    }
    if (t == 0) kcnt[b] = total;
}

extern "C" int mu_compact_keys(const void* keep, int keep_elem_bytes, int B, int N, int* kidx, int* kcnt, unsigned char* keep8,
                               void* stream) {
    if (!keep || !kidx || !kcnt || B <= 0 || N <= 0) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (keep_elem_bytes == 1) compact_keys_kernel<uint8_t><<<B, 1024, 0, st>>>((const uint8_t*)keep, N, kidx, kcnt, keep8);
    else if (keep_elem_bytes == 8) compact_keys_kernel<long><<<B, 1024, 0, st>>>((const long*)keep, N, kidx, kcnt, keep8);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// q/k/v projection weights of one attention block (three nn.Linear(C,C) with bias, ade_semantic.py:157-159) -> the compute layouts
// of ONE [3C, C] 1x1 layer in one launch: the forward block [3C][C], the data-gradient block [C][3C] (mu_prep_weight mode 2 of the
// concatenated weight) and the concatenated fp32 bias [3C].  Replaces two torch.cat + two .float() launches per block and step.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void prep_qkv_kernel(const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
                                const float* __restrict__ bq, const float* __restrict__ bk, const float* __restrict__ bv,
                                T* __restrict__ dst, float* __restrict__ bias, int C) {
    const long n = 3L * C * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * n + 3 * C; idx += (long)gridDim.x * blockDim.x) {
        if (idx >= 2 * n) {
            const int j = (int)(idx - 2 * n), which = j / C, c = j - which * C;
            bias[j] = (which == 0 ? bq : (which == 1 ? bk : bv))[c];
            continue;
        }
        int o, i;                                             // output row (0..3C) and input column (0..C) of the concatenated weight
        if (idx < n) { o = (int)(idx / C); i = (int)(idx % C); }
        else { const long j = idx - n; i = (int)(j / (3 * C)); o = (int)(j % (3 * C)); }
        const int which = o / C, r = o - which * C;
        dst[idx] = (T)(which == 0 ? wq : (which == 1 ? wk : wv))[(long)r * C + i];
    }
}

extern "C" int mu_prep_qkv(const float* wq, const float* wk, const float* wv, const float* bq, const float* bk, const float* bv, void* dst,
                           float* bias, int dtype, int C, void* stream) {
    if (!wq || !wk || !wv || !bq || !bk || !bv || !dst || !bias || C <= 0 || C % 32) return MU_ERR_ARG;
    const long n = 6L * C * C + 3 * C;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32) prep_qkv_kernel<float><<<grid, 256, 0, st>>>(wq, wk, wv, bq, bk, bv, (float*)dst, bias, C);
    else if (dtype == MU_F16) prep_qkv_kernel<h16><<<grid, 256, 0, st>>>(wq, wk, wv, bq, bk, bv, (h16*)dst, bias, C);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// SURVEY 8-f4, resize half: decoded image bytes of ANY size -> the network input, and the label map.
// The reference datasets run cv2.resize(image, (128,128), INTER_LINEAR) / cv2.resize(mask, (128,128), INTER_NEAREST) on the host
// (ade_semantic.py:72-73; opencv-python-headless 4.10, requirement.txt:168) after cv2.COLOR_BGR2RGB (:65), then ToTensor (:85).
// These kernels restate OpenCV's 8-bit algorithms (modules/imgproc/src/resize.cpp; cv2 itself is not vendored by the reference):
//   INTER_LINEAR, CV_8U: 11-bit fixed-point coefficients, horizontal pass in int, vertical pass
//       uchar((((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2);
//       columns whose taps leave the image collapse to the edge pixel with weight 2048, rows are clamped with their weights kept;
//       exactly-2x downscale in both directions takes cv::resize's INTER_AREA shortcut (a + b + c + d + 2) >> 2;
//   INTER_NEAREST: source index = min(floor(d * scale), n - 1), no half-pixel offset.
// The resized bytes are produced bit-exactly (u8_out, optional) and leave as [0,1] activations in the NHWC compute layout
// (x / 255, channel-padded with zeros), optionally with channels 0 and 2 swapped (BGR -> RGB).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void cv_lin_coef(int d, double scale, int sn, bool clamp_taps, int& s, int& c0, int& c1) {
    // two separately rounded double operations, as the host code this restates runs them: no fused multiply-add contraction (HIP's
    // __dmul_rn / __dadd_rn are plain operators and would contract)
#pragma clang fp contract(off)
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (clamp_taps) {
        if (si < 0) { f = 0.f; si = 0; }
        if (si + 1 >= sn) { f = 0.f; si = sn - 1; }
    }
    s = si;
    c0 = __float2int_rn((1.0f - f) * 2048.0f);               // cvRound: round half to even
    c1 = __float2int_rn(f * 2048.0f);
}

template <typename T>
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int B, int Hs, int Ws, int C, int swap_rb,
                                                               double scale_x, double scale_y, T* __restrict__ dst, uint8_t* __restrict__ u8_out,
                                                               int Hd, int Wd, int Cp) {
    constexpr int N = Vec16<T>::N;
    const long total = (long)B * Hd * Wd;
    const bool area2 = Ws == 2 * Wd && Hs == 2 * Hd;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int dx = (int)(idx % Wd), dy = (int)((idx / Wd) % Hd), b = (int)(idx / ((long)Wd * Hd));
        const uint8_t* img = src + (long)b * Hs * Ws * C;
        int v[4] = {0, 0, 0, 0};
        if (area2) {
            const uint8_t* p0 = img + ((long)(2 * dy) * Ws + 2 * dx) * C;
            const uint8_t* p1 = p0 + (long)Ws * C;
            for (int c = 0; c < C && c < 4; ++c) v[c] = ((int)p0[c] + (int)p0[C + c] + (int)p1[c] + (int)p1[C + c] + 2) >> 2;
        } else {
            int sx, a0, a1, sy, b0, b1;
            cv_lin_coef(dx, scale_x, Ws, true, sx, a0, a1);
            cv_lin_coef(dy, scale_y, Hs, false, sy, b0, b1);
            const int sx1 = sx + 1 < Ws ? sx + 1 : Ws - 1;       // weight 0 there at the right edge
            const int r0 = sy < 0 ? 0 : (sy < Hs ? sy : Hs - 1), r1 = sy + 1 < 0 ? 0 : (sy + 1 < Hs ? sy + 1 : Hs - 1);
            const uint8_t *q00 = img + ((long)r0 * Ws + sx) * C, *q01 = img + ((long)r0 * Ws + sx1) * C;
            const uint8_t *q10 = img + ((long)r1 * Ws + sx) * C, *q11 = img + ((long)r1 * Ws + sx1) * C;
            for (int c = 0; c < C && c < 4; ++c) {
                const int d0 = (int)q00[c] * a0 + (int)q01[c] * a1, d1 = (int)q10[c] * a0 + (int)q11[c] * a1;
                v[c] = ((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2) & 0xFF;
            }
        }
        if (swap_rb && C >= 3) { const int t = v[0]; v[0] = v[2]; v[2] = t; }
        if (u8_out)
            for (int c = 0; c < C && c < 4; ++c) u8_out[idx * C + c] = (uint8_t)v[c];
        for (int c0 = 0; c0 < Cp; c0 += N) {
            Vec16<T> o;
#pragma unroll
            for (int i = 0; i < N; ++i) o.set(i, (c0 + i < C && c0 + i < 4) ? (float)v[c0 + i] / 255.0f : 0.f);       // IEEE division (ToTensor on the host)
            o.store(dst + idx * Cp + c0);
        }
    }
}

extern "C" int mu_resize_u8_nhwc(const unsigned char* src, int B, int Hs, int Ws, int C, int swap_rb, void* dst, unsigned char* u8_out, int Hd,
                                 int Wd, int Cp, int dtype, void* stream) {
    if (!src || !dst || B <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return MU_ERR_ARG;
    if (C <= 0 || C > 4 || Cp < C || Cp % 8) return MU_ERR_SHAPE;
    // exactly what cv::resize derives from dsize: inv_scale = dsize / ssize, scale = 1 / inv_scale (doubles)
    const double scale_x = 1.0 / ((double)Wd / (double)Ws), scale_y = 1.0 / ((double)Hd / (double)Hs);
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)B * Hd * Wd;
    if (dtype == MU_F32) resize_linear_u8_kernel<float><<<ew_grid(total), 256, 0, st>>>(src, B, Hs, Ws, C, swap_rb, scale_x, scale_y, (float*)dst, u8_out, Hd, Wd, Cp);
    else if (dtype == MU_F16) resize_linear_u8_kernel<h16><<<ew_grid(total), 256, 0, st>>>(src, B, Hs, Ws, C, swap_rb, scale_x, scale_y, (h16*)dst, u8_out, Hd, Wd, Cp);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// label map: uint8 [B][Hs][Ws] -> int64 [B][Hd][Wd], cv2.INTER_NEAREST + torch.from_numpy(mask).long() (ade_semantic.py:73,78)
__global__ __launch_bounds__(256) void resize_nearest_u8_kernel(const uint8_t* __restrict__ src, int B, int Hs, int Ws, double ifx, double ify,
                                                                long* __restrict__ dst, int Hd, int Wd) {
    const long total = (long)B * Hd * Wd;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int dx = (int)(idx % Wd), dy = (int)((idx / Wd) % Hd), b = (int)(idx / ((long)Wd * Hd));
        int sx = (int)floor((double)dx * ifx), sy = (int)floor((double)dy * ify);
        sx = sx < Ws - 1 ? sx : Ws - 1;
        sy = sy < Hs - 1 ? sy : Hs - 1;
        dst[idx] = (long)src[((long)b * Hs + sy) * Ws + sx];
    }
}

extern "C" int mu_resize_nearest_u8(const unsigned char* src, int B, int Hs, int Ws, long* dst, int Hd, int Wd, void* stream) {
    if (!src || !dst || B <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return MU_ERR_ARG;
    const double ifx = 1.0 / ((double)Wd / (double)Ws), ify = 1.0 / ((double)Hd / (double)Hs);
    resize_nearest_u8_kernel<<<ew_grid((long)B * Hd * Wd), 256, 0, (hipStream_t)stream>>>(src, B, Hs, Ws, ifx, ify, dst, Hd, Wd);
    MU_CHECK_LAUNCH();
    return MU_OK;
}
