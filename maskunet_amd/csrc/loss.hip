// "Next" rows of the scope table: the steps either side of the forward/backward path.
//   f1  fused pixel-wise softmax cross-entropy on the NHWC (channel-padded) logits, forward + dlogits
//       (nn.CrossEntropyLoss, ade_semantic.py:377,399; ignore_index=255 in city_semantic.py:341)
//   f3  on-device mean IoU (mean_iou, ade_semantic.py:128-146): arg-max + per-class intersection / union counts,
//       no host round trips (the reference syncs once per class, :139)
#include "common.h"

// ------------------------------------------------------------------------------------------
// cross-entropy.  16 lanes cooperate on one pixel row (channels strided by 16*VN), 16 rows per block-iteration.
// forward: lse[row], per-block partial (sum of losses, number of counted rows) in fp64 -> finalize -> mean loss.
// backward: dlogits = (softmax - onehot) * gscale / count  (0 for ignored rows and for the channel padding).
// ------------------------------------------------------------------------------------------
#define CE_MAXBLK 1024
#ifndef MU_CE_U2
#define MU_CE_U2 4
#endif

// Rows of up to 3 * 128 channels are held in registers (one read of the logits; the two-pass form re-reads every row from L2 with a
// single load in flight per lane and ran at 1.5 TB/s); several rows per lane group per iteration keep more loads in flight.  Branch-free:
// padding channels enter as -inf (exp -> 0) and the target logit is fetched by address (the conditional per-element form compiled to an
// exec-mask branch per element and ~60 VALU cycles per logit: 162 us for the 64 x 128 x 128 x 150 logits of the bench, 2 TB/s).
template <typename T, int NV, int U>
__device__ __forceinline__ void ce_rows_in_regs(const T* __restrict__ logits, const long* __restrict__ labels, long M, int Cp, int C,
                                                long ignore_index, float* __restrict__ lse_out, double& loss, double& cnt, int l16, int rowl) {
    constexpr int VN = Vec16<T>::N;
    for (long r0 = (long)blockIdx.x * (16 * U); r0 < M; r0 += (long)gridDim.x * (16 * U)) {
        Vec16<T> v[U][NV];
        long lab[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = r0 + u * 16 + rowl;
            ok[u] = r < M;
            const long rr = ok[u] ? r : M - 1;
            lab[u] = labels[rr];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = (k * 16 + l16) * VN;
                if (k + 1 < NV || c < Cp) v[u][k].load(logits + rr * Cp + c);      // only the last vector of a row can lie past Cp
                else v[u][k].zero();
            }
        }
        // the target logit: one more (L2-resident, lane-group-uniform) load per row instead of a compare + select + add per element
        float tgt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = r0 + u * 16 + rowl;
            const long rr = r < M ? r : M - 1;
            tgt[u] = (lab[u] >= 0 && lab[u] < C) ? (float)logits[rr * Cp + lab[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float f[NV * VN];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = (k * 16 + l16) * VN;
                const bool whole = (k + 1) * 16 * VN <= C;                  // uniform: every lane's channels of this vector are real
#pragma unroll
                for (int i = 0; i < VN; ++i) {
                    const float x = v[u][k].get(i);
                    f[k * VN + i] = (whole || c + i < C) ? x : -INFINITY;    // channel padding: exp -> 0
                    mx = fmaxf(mx, f[k * VN + i]);
                }
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            const float nmx = -mx * 1.4426950408889634f;
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < NV * VN; ++j) se += __builtin_amdgcn_exp2f(fmaf(f[j], 1.4426950408889634f, nmx));
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) se += __shfl_xor(se, o);
            const float lse = mx + __logf(se);
            if (ok[u] && l16 == 0) {
                lse_out[r0 + u * 16 + rowl] = lse;
                if (lab[u] != ignore_index) { loss += (double)(lse - tgt[u]); cnt += 1.0; }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const long* __restrict__ labels, long M, int Cp,
                                                     int C, long ignore_index, float* __restrict__ lse_out, double* __restrict__ part) {
    constexpr int VN = Vec16<T>::N;
    const int tid = threadIdx.x, l16 = tid & 15, rowl = tid >> 4;
    double loss = 0.0, cnt = 0.0;
    const int nv = (Cp + 16 * VN - 1) / (16 * VN);
    // U rows per 16-lane group and iteration: the bytes in flight per wave (one exposed memory round trip per iteration)
    if (nv == 1) ce_rows_in_regs<T, 1, 4>(logits, labels, M, Cp, C, ignore_index, lse_out, loss, cnt, l16, rowl);
    else if (nv == 2) ce_rows_in_regs<T, 2, MU_CE_U2>(logits, labels, M, Cp, C, ignore_index, lse_out, loss, cnt, l16, rowl);
    else if (nv == 3) ce_rows_in_regs<T, 3, 2>(logits, labels, M, Cp, C, ignore_index, lse_out, loss, cnt, l16, rowl);
    else
    for (long r0 = (long)blockIdx.x * 16; r0 < M; r0 += (long)gridDim.x * 16) {
        const long r = r0 + rowl;
        const bool ok = r < M;
        const long rr = ok ? r : M - 1;
        const long lab = labels[rr];
        float mx = -INFINITY;
        for (int c = l16 * VN; c < Cp; c += 16 * VN) {
            Vec16<T> v;
            v.load(logits + rr * Cp + c);
#pragma unroll
            for (int i = 0; i < VN; ++i) if (c + i < C) mx = fmaxf(mx, v.get(i));
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float se = 0.f, tgt = 0.f;
        for (int c = l16 * VN; c < Cp; c += 16 * VN) {
            Vec16<T> v;
            v.load(logits + rr * Cp + c);
#pragma unroll
            for (int i = 0; i < VN; ++i) {
                if (c + i < C) {
                    se += __expf(v.get(i) - mx);
                    if (c + i == lab) tgt = v.get(i);
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { se += __shfl_xor(se, o); tgt += __shfl_xor(tgt, o); }
        const float lse = mx + __logf(se);
        if (ok && l16 == 0) {
            lse_out[r] = lse;
            if (lab != ignore_index) { loss += (double)(lse - tgt); cnt += 1.0; }
        }
    }
    __shared__ double sh[32];
    loss = wave_sum_d(loss);
    cnt = wave_sum_d(cnt);
    if ((tid & 63) == 0) { sh[(tid >> 6) * 2] = loss; sh[(tid >> 6) * 2 + 1] = cnt; }
    __syncthreads();
    if (tid == 0) {
        part[blockIdx.x * 2] = sh[0] + sh[2] + sh[4] + sh[6];
        part[blockIdx.x * 2 + 1] = sh[1] + sh[3] + sh[5] + sh[7];
    }
}

__global__ void ce_final_kernel(const double* __restrict__ part, int nblk, float* __restrict__ loss, float* __restrict__ count) {
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) { a += part[i * 2]; c += part[i * 2 + 1]; }
    a = wave_sum_d(a);
    c = wave_sum_d(c);
    if (threadIdx.x == 0) {
        loss[0] = (float)(a / c);          // 0/0 = NaN when every pixel is ignored, as torch
        count[0] = (float)c;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const long* __restrict__ labels, const float* __restrict__ lse,
                                                     const float* __restrict__ count, const float* __restrict__ gout, float gscale,
                                                     long M, int Cp, int C, long ignore_index, T* __restrict__ dlogits) {
    constexpr int VN = Vec16<T>::N;
    const int cv = Cp / VN;
    const long total = M * cv;
    const float k = gout[0] * gscale / count[0];
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / cv;
        const int c = (int)(idx % cv) * VN;
        const long lab = labels[r];
        Vec16<T> v, o;
        v.load_nt(logits + r * Cp + c);       // read once (norm.hip MU_BN_NT)
        const float l = lse[r];
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            float g = 0.f;
            if (lab != ignore_index && c + i < C) g = (__expf(v.get(i) - l) - (c + i == lab ? 1.f : 0.f)) * k;
            o.set(i, g);
        }
        o.store(dlogits + r * Cp + c);
    }
}

extern "C" long mu_ce_workspace_bytes(void) { return (long)CE_MAXBLK * 2 * sizeof(double); }

extern "C" int mu_ce_fwd(const void* logits, const long* labels, long M, int Cp, int C, long ignore_index, float* lse, float* loss,
                         float* count, void* workspace, long ws_bytes, int dtype, void* stream) {
    if (!logits || !labels || !lse || !loss || !count || !workspace || M <= 0 || C <= 0 || Cp < C || Cp % 8) return MU_ERR_ARG;
    if (ws_bytes < mu_ce_workspace_bytes()) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int nblk = (int)((M + 31) / 32 < CE_MAXBLK ? (M + 31) / 32 : CE_MAXBLK);
    if (dtype == MU_F16) ce_fwd_kernel<h16><<<nblk, 256, 0, st>>>((const h16*)logits, labels, M, Cp, C, ignore_index, lse, (double*)workspace);
    else if (dtype == MU_F32) ce_fwd_kernel<float><<<nblk, 256, 0, st>>>((const float*)logits, labels, M, Cp, C, ignore_index, lse, (double*)workspace);
    else return MU_ERR_ARG;
    ce_final_kernel<<<1, 64, 0, st>>>((const double*)workspace, nblk, loss, count);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_ce_bwd(const void* logits, const long* labels, const float* lse, const float* count, const float* grad_out,
                         float grad_scale, long M, int Cp, int C, long ignore_index, void* dlogits, int dtype, void* stream) {
    if (!logits || !labels || !lse || !count || !grad_out || !dlogits || M <= 0 || C <= 0 || Cp < C || Cp % 8) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F16) {
        long total = M * (Cp / 8);
        int g = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        ce_bwd_kernel<h16><<<g, 256, 0, st>>>((const h16*)logits, labels, lse, count, grad_out, grad_scale, M, Cp, C, ignore_index, (h16*)dlogits);
    } else if (dtype == MU_F32) {
        long total = M * (Cp / 4);
        int g = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        ce_bwd_kernel<float><<<g, 256, 0, st>>>((const float*)logits, labels, lse, count, grad_out, grad_scale, M, Cp, C, ignore_index, (float*)dlogits);
    } else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// cross-entropy on the module's own output layout: NCHW logits [B, C, HW] (fp32 or fp16), labels [B, HW].
// One lane owns VEC consecutive pixels and walks the channel planes (each wave reads whole 128-byte rows of a plane);
// single pass with an online max / rescaled sum, eight planes in flight.  Same reductions and outputs as the NHWC kernels.
// ------------------------------------------------------------------------------------------
template <typename T, int VEC> struct PixVec;
template <> struct PixVec<float, 4> {
    static __device__ inline void load(const float* p, float* o) { const float4 v = *(const float4*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    static __device__ inline void store(float* p, const float* o) { *(float4*)p = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct PixVec<float, 1> {
    static __device__ inline void load(const float* p, float* o) { o[0] = p[0]; }
    static __device__ inline void store(float* p, const float* o) { p[0] = o[0]; }
};
template <> struct PixVec<h16, 4> {
    static __device__ inline void load(const h16* p, float* o) {
        const uint2 v = *(const uint2*)p;
        const h16* h = (const h16*)&v;
        for (int i = 0; i < 4; ++i) o[i] = (float)h[i];
    }
    static __device__ inline void store(h16* p, const float* o) {
        uint2 v;
        h16* h = (h16*)&v;
        for (int i = 0; i < 4; ++i) h[i] = (h16)o[i];
        *(uint2*)p = v;
    }
};
template <> struct PixVec<h16, 1> {
    static __device__ inline void load(const h16* p, float* o) { o[0] = (float)p[0]; }
    static __device__ inline void store(h16* p, const float* o) { p[0] = (h16)o[0]; }
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void ce_nchw_fwd_kernel(const T* __restrict__ logits, const long* __restrict__ labels, int B, int C, long HW,
                                                          long ignore_index, float* __restrict__ lse_out, double* __restrict__ part) {
    constexpr int U = 8;
    const int tid = threadIdx.x;
    const long groups = HW / VEC, total = (long)B * groups;
    double loss = 0.0, cnt = 0.0;
    for (long g = (long)blockIdx.x * 256 + tid; g < total; g += (long)gridDim.x * 256) {
        const long b = g / groups, p0 = (g - b * groups) * VEC;
        const T* base = logits + (b * C) * HW + p0;
        long lab[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) lab[i] = labels[b * HW + p0 + i];
        float mx[VEC], se[VEC], tgt[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { mx[i] = -INFINITY; se[i] = 0.f; tgt[i] = 0.f; }
        for (int c0 = 0; c0 < C; c0 += U) {
            float v[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u < C ? c0 + u : C - 1;
                PixVec<T, VEC>::load(base + (long)c * HW, v[u]);
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float m = mx[i];
#pragma unroll
                for (int u = 0; u < U; ++u) if (c0 + u < C) m = fmaxf(m, v[u][i]);
                const float mm = m == -INFINITY ? 0.f : m;      // a chunk of -inf logits contributes exp(-inf) = 0, not NaN
                float s = se[i] * __expf(mx[i] - mm);            // exp(-inf - mm) = 0 on the first chunk (se = 0 there)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (c0 + u < C) {
                        s += __expf(v[u][i] - mm);
                        if ((long)(c0 + u) == lab[i]) tgt[i] = v[u][i];
                    }
                }
                mx[i] = m;
                se[i] = s;
            }
        }
        float l[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            l[i] = mx[i] + __logf(se[i]);
            if (lab[i] != ignore_index) { loss += (double)(l[i] - tgt[i]); cnt += 1.0; }
        }
        PixVec<float, VEC>::store(lse_out + b * HW + p0, l);
    }
    __shared__ double sh[8];
    loss = wave_sum_d(loss);
    cnt = wave_sum_d(cnt);
    if ((tid & 63) == 0) { sh[(tid >> 6) * 2] = loss; sh[(tid >> 6) * 2 + 1] = cnt; }
    __syncthreads();
    if (tid == 0) {
        part[blockIdx.x * 2] = sh[0] + sh[2] + sh[4] + sh[6];
        part[blockIdx.x * 2 + 1] = sh[1] + sh[3] + sh[5] + sh[7];
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void ce_nchw_bwd_kernel(const T* __restrict__ logits, const long* __restrict__ labels, const float* __restrict__ lse,
                                                          const float* __restrict__ count, const float* __restrict__ gout, float gscale,
                                                          int B, int C, long HW, long ignore_index, T* __restrict__ dlogits) {
    constexpr int U = 8;
    const long groups = HW / VEC, total = (long)B * groups;
    const float k = gout[0] * gscale / count[0];
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
        const long b = g / groups, p0 = (g - b * groups) * VEC;
        const long off = (b * C) * HW + p0;
        long lab[VEC];
        float l[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) lab[i] = labels[b * HW + p0 + i];
        PixVec<float, VEC>::load(lse + b * HW + p0, l);
        for (int c0 = 0; c0 < C; c0 += U) {
            float v[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u < C ? c0 + u : C - 1;
                PixVec<T, VEC>::load(logits + off + (long)c * HW, v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c0 + u < C) {
                    float o[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        o[i] = lab[i] != ignore_index ? (__expf(v[u][i] - l[i]) - ((long)(c0 + u) == lab[i] ? 1.f : 0.f)) * k : 0.f;
                    PixVec<T, VEC>::store(dlogits + off + (long)(c0 + u) * HW, o);
                }
            }
        }
    }
}

template <typename T, int VEC>
static void ce_nchw_fwd_t(const void* logits, const long* labels, int B, int C, long HW, long ignore_index, float* lse, double* part, int nblk,
                          hipStream_t st) {
    ce_nchw_fwd_kernel<T, VEC><<<nblk, 256, 0, st>>>((const T*)logits, labels, B, C, HW, ignore_index, lse, part);
}

extern "C" int mu_ce_nchw_fwd(const void* logits, const long* labels, int B, int C, long HW, long ignore_index, float* lse, float* loss,
                              float* count, void* workspace, long ws_bytes, int dtype, void* stream) {
    if (!logits || !labels || !lse || !loss || !count || !workspace || B <= 0 || C <= 0 || HW <= 0) return MU_ERR_ARG;
    if (ws_bytes < mu_ce_workspace_bytes()) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = HW % 4 == 0;
    const long groups = (long)B * (v4 ? HW / 4 : HW);
    const int nblk = (int)((groups + 255) / 256 < CE_MAXBLK ? (groups + 255) / 256 : CE_MAXBLK);
    if (dtype == MU_F32) {
        if (v4) ce_nchw_fwd_t<float, 4>(logits, labels, B, C, HW, ignore_index, lse, (double*)workspace, nblk, st);
        else ce_nchw_fwd_t<float, 1>(logits, labels, B, C, HW, ignore_index, lse, (double*)workspace, nblk, st);
    } else if (dtype == MU_F16) {
        if (v4) ce_nchw_fwd_t<h16, 4>(logits, labels, B, C, HW, ignore_index, lse, (double*)workspace, nblk, st);
        else ce_nchw_fwd_t<h16, 1>(logits, labels, B, C, HW, ignore_index, lse, (double*)workspace, nblk, st);
    } else return MU_ERR_ARG;
    ce_final_kernel<<<1, 64, 0, st>>>((const double*)workspace, nblk, loss, count);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

template <typename T, int VEC>
static void ce_nchw_bwd_t(const void* logits, const long* labels, const float* lse, const float* count, const float* gout, float gscale, int B,
                          int C, long HW, long ignore_index, void* dlogits, hipStream_t st) {
    const long groups = (long)B * (HW / VEC);
    const int nblk = (int)((groups + 255) / 256 < 4096 ? (groups + 255) / 256 : 4096);
    ce_nchw_bwd_kernel<T, VEC><<<nblk, 256, 0, st>>>((const T*)logits, labels, lse, count, gout, gscale, B, C, HW, ignore_index, (T*)dlogits);
}

extern "C" int mu_ce_nchw_bwd(const void* logits, const long* labels, const float* lse, const float* count, const float* grad_out,
                              float grad_scale, int B, int C, long HW, long ignore_index, void* dlogits, int dtype, void* stream) {
    if (!logits || !labels || !lse || !count || !grad_out || !dlogits || B <= 0 || C <= 0 || HW <= 0) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = HW % 4 == 0;
    if (dtype == MU_F32) {
        if (v4) ce_nchw_bwd_t<float, 4>(logits, labels, lse, count, grad_out, grad_scale, B, C, HW, ignore_index, dlogits, st);
        else ce_nchw_bwd_t<float, 1>(logits, labels, lse, count, grad_out, grad_scale, B, C, HW, ignore_index, dlogits, st);
    } else if (dtype == MU_F16) {
        if (v4) ce_nchw_bwd_t<h16, 4>(logits, labels, lse, count, grad_out, grad_scale, B, C, HW, ignore_index, dlogits, st);
        else ce_nchw_bwd_t<h16, 1>(logits, labels, lse, count, grad_out, grad_scale, B, C, HW, ignore_index, dlogits, st);
    } else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// mean IoU.  counts[0][c] = #(pred==c && label==c), counts[1][c] = #(pred==c), counts[2][c] = #(label==c);
// iou_c = (I + smooth) / (U + smooth) with U = P + L - I, averaged over classes with U > 0 (ade_semantic.py:135-146).
// logits: row r = pixel, element (r, c) at logits[(r / inner) * outer_stride + c * c_stride + (r % inner) * p_stride]
//   NHWC padded: inner = M, outer_stride = 0, c_stride = 1, p_stride = Cp;  NCHW: inner = H*W, outer = C*H*W, c_stride = H*W, p_stride = 1
// softmax(y/0.5) is monotone, so the arg-max is taken on the logits (first maximum, as torch.argmax).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void iou_count_kernel(const T* __restrict__ logits, const long* __restrict__ labels, long M, int C,
                                                        long inner, long outer_stride, long c_stride, long p_stride,
                                                        unsigned int* __restrict__ counts) {
    extern __shared__ unsigned int sh[];          // [3][C]
    for (int i = threadIdx.x; i < 3 * C; i += 256) sh[i] = 0;
    __syncthreads();
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
        const T* base = logits + (r / inner) * outer_stride + (r % inner) * p_stride;
        float best = (float)base[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = (float)base[(long)c * c_stride];
            if (v > best) { best = v; arg = c; }
        }
        const long lab = labels[r];
        atomicAdd(&sh[C + arg], 1u);
        if (lab >= 0 && lab < C) {
            atomicAdd(&sh[2 * C + (int)lab], 1u);
            if (lab == arg) atomicAdd(&sh[arg], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += 256) if (sh[i]) atomicAdd(&counts[i], sh[i]);
}

__global__ void iou_final_kernel(const unsigned int* __restrict__ counts, int C, float smooth, float* __restrict__ out) {
    float s = 0.f, n = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) {
        const float I = (float)counts[c], U = (float)counts[C + c] + (float)counts[2 * C + c] - (float)counts[c];
        if (U > 0.f) { s += (I + smooth) / (U + smooth); n += 1.f; }
    }
    s = wave_sum(s);
    n = wave_sum(n);
    if (threadIdx.x == 0) out[0] = s / n;
}

extern "C" int mu_mean_iou(const void* logits, const long* labels, long M, int C, long inner, long outer_stride, long c_stride,
                           long p_stride, float smooth, unsigned int* counts, float* out, int dtype, void* stream) {
    if (!logits || !labels || !counts || !out || M <= 0 || C <= 0 || C > 4096 || inner <= 0) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, (size_t)3 * C * sizeof(unsigned int), st) != hipSuccess) return MU_ERR_LAUNCH;
    int g = (int)((M + 255) / 256 > 2048 ? 2048 : (M + 255) / 256);
    size_t lds = (size_t)3 * C * sizeof(unsigned int);
    if (dtype == MU_F16) iou_count_kernel<h16><<<g, 256, lds, st>>>((const h16*)logits, labels, M, C, inner, outer_stride, c_stride, p_stride, counts);
    else if (dtype == MU_F32) iou_count_kernel<float><<<g, 256, lds, st>>>((const float*)logits, labels, M, C, inner, outer_stride, c_stride, p_stride, counts);
    else return MU_ERR_ARG;
    iou_final_kernel<<<1, 64, 0, st>>>(counts, C, smooth, out);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// f2: fused multi-tensor AdamW (optim.AdamW(lr, weight_decay), ade_semantic.py:379,401; torch semantics: decoupled decay,
// bias-corrected moments).  One launch updates every parameter: a device table lists {p, g, m, v, n} per tensor and a
// block map assigns (tensor, chunk) to each block.  grad_scale_inv un-scales fp16-loss-scaled gradients on the fly.
// ------------------------------------------------------------------------------------------
struct MuAdamEntry { float* p; const float* g; float* m; float* v; long n; long step; };   // 48 bytes
// step = the tensor's own count of ATTEMPTED updates including this one (torch keeps one counter per parameter); the number of
// those that were skipped for a non-finite gradient lives in skipped[tensor] on the device, so the bias corrections
// bc1 = 1 - beta1^t, bc2 = 1 - beta2^t use t = step - skipped without the host ever reading the overflow flag.
#define ADAM_CHUNK 4096

__global__ __launch_bounds__(256) void adamw_kernel(const MuAdamEntry* __restrict__ table, const int* __restrict__ block_tensor,
                                                    const int* __restrict__ block_chunk, float lr, float beta1, float beta2, float eps,
                                                    float wd, float ginv, const float* __restrict__ grad_scale,
                                                    const float* __restrict__ found_inf, const int* __restrict__ skipped) {
    if (found_inf && *found_inf != 0.f) return;     // an overflowed step updates nothing (GradScaler semantics), decided on the device
    const int ti = block_tensor[blockIdx.x];
    const MuAdamEntry e = table[ti];
    if (!e.g) return;                               // parameter without a gradient this step
    if (grad_scale) ginv = 1.0f / *grad_scale;      // GradScaler's device-side scale
    __shared__ float sbc[2];
    if (threadIdx.x == 0) {
        long ti_ = e.step - (skipped ? skipped[ti] : 0);
        const double t = (double)(ti_ < 1 ? 1 : ti_);    // never 0 or negative (bc1 = 0 -> lr / bc1 = inf), whatever the host counters say
        sbc[0] = (float)(1.0 - pow((double)beta1, t));
        sbc[1] = (float)sqrt(1.0 - pow((double)beta2, t));
    }
    __syncthreads();
    const float bc1 = sbc[0], bc2_sqrt = sbc[1];
    const long base = (long)block_chunk[blockIdx.x] * ADAM_CHUNK;
    for (int i = threadIdx.x; i < ADAM_CHUNK; i += 256) {
        const long k = base + i;
        if (k >= e.n) break;
        const float g = e.g[k] * ginv;
        float p = e.p[k];
        p *= 1.f - lr * wd;
        const float m = beta1 * e.m[k] + (1.f - beta1) * g;
        const float v = beta2 * e.v[k] + (1.f - beta2) * g * g;
        e.m[k] = m;
        e.v[k] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        e.p[k] = p - (lr / bc1) * (m / denom);
    }
}

// found_inf[0] = 1 if any gradient element of the table is inf / NaN (the caller zeroes it first): one read of every gradient
__global__ __launch_bounds__(256) void grads_nonfinite_kernel(const MuAdamEntry* __restrict__ table, const int* __restrict__ block_tensor,
                                                              const int* __restrict__ block_chunk, float* __restrict__ found_inf) {
    const MuAdamEntry e = table[block_tensor[blockIdx.x]];
    if (!e.g) return;
    const long base = (long)block_chunk[blockIdx.x] * ADAM_CHUNK;
    bool bad = false;
    for (int i = threadIdx.x; i < ADAM_CHUNK; i += 256) {
        const long k = base + i;
        if (k >= e.n) break;
        const float g = e.g[k];
        bad = bad || !(fabsf(g) <= 3.4028234e38f);     // false for inf and for NaN
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) *found_inf = 1.0f;
}
// after the (possibly skipped) update: count the skip for every tensor that had a gradient
__global__ void adamw_skip_count_kernel(const MuAdamEntry* __restrict__ table, int ntensors, const float* __restrict__ found_inf,
                                        int* __restrict__ skipped) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntensors && *found_inf != 0.f && table[i].g) skipped[i] += 1;
}

extern "C" int mu_adamw_chunk(void) { return ADAM_CHUNK; }

extern "C" int mu_adamw_multi(const void* table, const int* block_tensor, const int* block_chunk, int nblocks, int ntensors, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float grad_scale_inv, const float* grad_scale,
                              float* found_inf, int check_finite, int* skipped, void* stream) {
    if (!table || !block_tensor || !block_chunk || nblocks <= 0 || ntensors <= 0) return MU_ERR_ARG;
    if (check_finite && !found_inf) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (check_finite == 1 && hipMemsetAsync(found_inf, 0, sizeof(float), st) != hipSuccess) return MU_ERR_LAUNCH;
    if (check_finite) grads_nonfinite_kernel<<<nblocks, 256, 0, st>>>((const MuAdamEntry*)table, block_tensor, block_chunk, found_inf);
    if (check_finite != 2) {
        adamw_kernel<<<nblocks, 256, 0, st>>>((const MuAdamEntry*)table, block_tensor, block_chunk, lr, beta1, beta2, eps, weight_decay,
                                              grad_scale_inv, grad_scale, found_inf, skipped);
        if (found_inf && skipped)
            adamw_skip_count_kernel<<<(ntensors + 255) / 256, 256, 0, st>>>((const MuAdamEntry*)table, ntensors, found_inf, skipped);
    }
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// f1 (second half): InstanceContrastiveLoss (ade_panoptic.py:390-418 and coco_panoptic.py:482-521, no ignore label; city_instance.py:279-307,
// ignore label 255 -- ignore_label < 0 here means "none") without
// torch.unique / nonzero host round trips.  Per instance id != 0 (and != ignore) with >= 2 pixels: triplet margin loss between the
// feature columns addressed by the first two pixels of the instance and by the floor(u[k] n_neg)-th pixel outside it; mean over the
// instances.  "Feature column of pixel (b,h,w)" is features[:, :, b, h] ([B*C] values) -- the reference indexes dims 2,3 with the
// batch and row index vectors of nonzero(as_tuple=True); restated as is (needs B <= H and H <= W).
// Workspace (ints unless noted): count[id_cap] first[id_cap] second[id_cap] ids[max_inst] K neg[max_inst] | floats li[max_inst]
// dap[max_inst] dan[max_inst].  Instance ids must lie in [0, id_cap); others are ignored (flag in ws).
// ------------------------------------------------------------------------------------------
struct InstWs {
    int *count, *first, *second, *ids, *K, *neg, *flag;
    float *li, *dap, *dan;
};
static inline long inst_ws_bytes(int id_cap, int max_inst) { return (3L * id_cap + 2L * max_inst + 8) * sizeof(int) + 3L * max_inst * sizeof(float); }
__host__ __device__ static inline InstWs inst_ws(void* ws, int id_cap, int max_inst) {
    InstWs w;
    int* p = (int*)ws;
    w.count = p; p += id_cap;
    w.first = p; p += id_cap;
    w.second = p; p += id_cap;
    w.ids = p; p += max_inst;
    w.neg = p; p += max_inst;
    w.K = p; w.flag = p + 1; p += 8;
    w.li = (float*)p; w.dap = w.li + max_inst; w.dan = w.dap + max_inst;
    return w;
}
__global__ void inst_init_kernel(InstWs w, int id_cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < id_cap) { w.count[i] = 0; w.first[i] = 0x7fffffff; w.second[i] = 0x7fffffff; }
    if (i == 0) { *w.K = 0; *w.flag = 0; }
}
// Block-aggregated: every block owns a contiguous pixel range and folds it into a 128-slot open-addressing table in LDS
// (LDS atomics), then issues ONE global atomic per distinct id it saw.  (A per-pixel global atomicAdd on ~20 hot counters
// serialises: 10 ms for 1M pixels; one per wave and id still 0.5 ms.)  Ids that do not find a slot fall back to global atomics.
#define INST_SLOTS 128
template <int PASS>
__global__ __launch_bounds__(256) void inst_scan_kernel(const long* __restrict__ mask, int npix, int id_cap, InstWs w) {
    __shared__ int key[INST_SLOTS], cnt[INST_SLOTS], mn[INST_SLOTS];
    for (int i = threadIdx.x; i < INST_SLOTS; i += 256) { key[i] = -1; cnt[i] = 0; mn[i] = 0x7fffffff; }
    __syncthreads();
    const int per = (npix + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = lo + per < npix ? lo + per : npix;
    for (int p = lo + threadIdx.x; p < hi; p += 256) {
        const long idl = mask[p];
        if (idl < 0 || idl >= id_cap) { if (PASS == 0) *w.flag = 1; continue; }
        const int id = (int)idl;
        if (PASS == 1 && p <= w.first[id]) continue;
        int slot = (id * 40503u) & (INST_SLOTS - 1), tries = 0;
        for (; tries < 16; ++tries) {
            const int prev = atomicCAS(&key[slot], -1, id);
            if (prev == -1 || prev == id) break;
            slot = (slot + 1) & (INST_SLOTS - 1);
        }
        if (tries < 16) {
            if (PASS == 0) atomicAdd(&cnt[slot], 1);
            atomicMin(&mn[slot], p);
        } else if (PASS == 0) {
            atomicAdd(&w.count[id], 1);
            atomicMin(&w.first[id], p);
        } else {
            atomicMin(&w.second[id], p);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < INST_SLOTS; i += 256) {
        if (key[i] < 0) continue;
        if (PASS == 0) {
            atomicAdd(&w.count[key[i]], cnt[i]);
            atomicMin(&w.first[key[i]], mn[i]);
        } else if (mn[i] != 0x7fffffff) {
            atomicMin(&w.second[key[i]], mn[i]);
        }
    }
}
// sorted list of the instance ids that reach the draw (one block; ids in increasing order = torch.unique's order)
__global__ __launch_bounds__(1024) void inst_list_kernel(InstWs w, int id_cap, int npix, int ignore, int max_inst) {
    __shared__ int psum[1024];
    const int per = (id_cap + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < id_cap ? lo + per : id_cap;
    int n = 0;
    for (int id = lo; id < hi; ++id) n += (id != 0 && id != ignore && w.count[id] >= 2 && npix - w.count[id] > 0);
    psum[threadIdx.x] = n;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                         // inclusive scan
        const int v = threadIdx.x >= o ? psum[threadIdx.x - o] : 0;
        __syncthreads();
        psum[threadIdx.x] += v;
        __syncthreads();
    }
    int k = psum[threadIdx.x] - n;
    for (int id = lo; id < hi; ++id)
        if (id != 0 && id != ignore && w.count[id] >= 2 && npix - w.count[id] > 0) {
            if (k < max_inst) w.ids[k] = id;
            ++k;
        }
    if (threadIdx.x == 1023) *w.K = psum[1023] < max_inst ? psum[1023] : max_inst;
}
// block k: the floor(u[k] * n_neg)-th pixel (row-major) whose label differs from instance k's id.  8192 pixels per iteration
// (8 consecutive per thread), exclusive prefix of the per-thread negative counts by wave shuffles + a 16-entry LDS hand-over.
__global__ __launch_bounds__(1024) void inst_neg_kernel(const long* __restrict__ mask, int npix, const float* __restrict__ u, InstWs w) {
    const int k = blockIdx.x;
    if (k >= *w.K) return;
    const long id = w.ids[k];
    const int nneg = npix - w.count[id];
    int r = (int)(u[k] * (float)nneg);
    r = r < nneg - 1 ? r : nneg - 1;
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int base = 0;                                              // negatives before this iteration (same in every thread)
    for (int p0 = 0; p0 < npix; p0 += 8192) {
        const int pt = p0 + threadIdx.x * 8;
        int neg[8], c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { neg[i] = (pt + i < npix && mask[pt + i] != id) ? 1 : 0; c += neg[i]; }
        int incl = c;                                          // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int before = base + incl - c;
        for (int i = 0; i < wv; ++i) before += wsum[i];
        int tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += wsum[i];
        if (r >= before && r < before + c) {
            int run = before;
#pragma unroll
            for (int i = 0; i < 8; ++i) { if (neg[i] && run == r) w.neg[k] = pt + i; run += neg[i]; }
        }
        base += tot;
        __syncthreads();
        if (base > r) break;
    }
}
__device__ __forceinline__ long inst_col(int p, int H, int W) {       // pixel p = (b, h, w) -> offset of features[0, 0, b, h]
    const int b = p / (H * W), h = (p / W) % H;
    return (long)b * W + h;
}
// block k: d(a,p), d(a,n) over the B*C feature column entries, li = max(d_ap - d_an + margin, 0)
__global__ __launch_bounds__(256) void inst_dist_kernel(const float* __restrict__ feat, int BC, int H, int W, float margin, InstWs w) {
    const int k = blockIdx.x;
    if (k >= *w.K) return;
    const int id = w.ids[k];
    const long ca = inst_col(w.first[id], H, W), cp = inst_col(w.second[id], H, W), cn = inst_col(w.neg[k], H, W);
    const long hw = (long)H * W;
    float sp = 0.f, sn = 0.f;
    for (int e = threadIdx.x; e < BC; e += 256) {
        const float a = feat[e * hw + ca];
        const float dp = a - feat[e * hw + cp] + 1e-6f, dn = a - feat[e * hw + cn] + 1e-6f;
        sp = fmaf(dp, dp, sp);
        sn = fmaf(dn, dn, sn);
    }
    __shared__ float rp[256], rn[256];
    rp[threadIdx.x] = sp; rn[threadIdx.x] = sn;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { rp[threadIdx.x] += rp[threadIdx.x + o]; rn[threadIdx.x] += rn[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float dap = sqrtf(rp[0]), dan = sqrtf(rn[0]);
        w.dap[k] = dap; w.dan[k] = dan;
        w.li[k] = fmaxf(dap - dan + margin, 0.f);
    }
}
__global__ void inst_final_kernel(InstWs w, float* __restrict__ loss) {
    const int K = *w.K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += w.li[k];
    loss[0] = K > 0 ? s / (float)K : 0.f;
}
// one block walks the instances in order (their feature columns may coincide), threads over the B*C entries: deterministic
__global__ __launch_bounds__(1024) void inst_bwd_kernel(const float* __restrict__ feat, int BC, int H, int W, InstWs w,
                                                        const float* __restrict__ grad_out, float* __restrict__ dfeat) {
    const int K = *w.K;
    const long hw = (long)H * W;
    const float g = K > 0 ? grad_out[0] / (float)K : 0.f;
    for (int k = 0; k < K; ++k) {
        if (w.li[k] > 0.f) {
            const int id = w.ids[k];
            const long ca = inst_col(w.first[id], H, W), cp = inst_col(w.second[id], H, W), cn = inst_col(w.neg[k], H, W);
            const float ip = g / w.dap[k], in_ = g / w.dan[k];
            for (int e = threadIdx.x; e < BC; e += 1024) {
                const float a = feat[e * hw + ca];
                const float up = (a - feat[e * hw + cp] + 1e-6f) * ip, un = (a - feat[e * hw + cn] + 1e-6f) * in_;
                dfeat[e * hw + ca] += up - un;          // same thread, in order: columns may coincide (a == p when the first two
                dfeat[e * hw + cp] -= up;               // pixels share an image row)
                dfeat[e * hw + cn] += un;
            }
        }
        __syncthreads();
    }
}

extern "C" long mu_inst_triplet_workspace_bytes(int id_cap, int max_inst) { return inst_ws_bytes(id_cap, max_inst); }

extern "C" int mu_inst_triplet_fwd(const float* feat, const long* mask, int B, int C, int H, int W, int ignore_label, float margin,
                                   const float* u, int id_cap, int max_inst, void* workspace, long ws_bytes, float* loss, void* stream) {
    if (!feat || !mask || !u || !workspace || !loss || B <= 0 || C <= 0 || H <= 0 || W <= 0 || id_cap <= 1 || max_inst <= 0) return MU_ERR_ARG;
    if (B > H || H > W) return MU_ERR_SHAPE;             // the (batch, row) -> (h, w) indexing of the reference would run off the tensor
    if ((long)B * H * W > 0x7fffffffL) return MU_ERR_SHAPE;
    if (ws_bytes < inst_ws_bytes(id_cap, max_inst)) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const InstWs w = inst_ws(workspace, id_cap, max_inst);
    const int npix = B * H * W;
    const int nb = (npix + 4095) / 4096 < 1024 ? (npix + 4095) / 4096 : 1024;      // >= 4096 pixels per block: few table flushes
    inst_init_kernel<<<(id_cap + 255) / 256, 256, 0, st>>>(w, id_cap);
    inst_scan_kernel<0><<<nb, 256, 0, st>>>(mask, npix, id_cap, w);
    inst_scan_kernel<1><<<nb, 256, 0, st>>>(mask, npix, id_cap, w);
    inst_list_kernel<<<1, 1024, 0, st>>>(w, id_cap, npix, ignore_label, max_inst);
    inst_neg_kernel<<<max_inst, 1024, 0, st>>>(mask, npix, u, w);
    inst_dist_kernel<<<max_inst, 256, 0, st>>>(feat, B * C, H, W, margin, w);
    inst_final_kernel<<<1, 1, 0, st>>>(w, loss);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_inst_triplet_bwd(const float* feat, int B, int C, int H, int W, const void* workspace, int id_cap, int max_inst,
                                   const float* grad_out, float* dfeat, void* stream) {
    if (!feat || !workspace || !grad_out || !dfeat || B <= 0 || C <= 0 || H <= 0 || W <= 0) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dfeat, 0, (size_t)B * C * H * W * sizeof(float), st) != hipSuccess) return MU_ERR_LAUNCH;
    inst_bwd_kernel<<<1, 1024, 0, st>>>(feat, B * C, H, W, inst_ws(const_cast<void*>(workspace), id_cap, max_inst), grad_out, dfeat);
    MU_CHECK_LAUNCH();
    return MU_OK;
}
