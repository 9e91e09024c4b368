// BatchNorm2d (+GELU/ReLU, +residual) and per-sample LayerNorm kernels, NHWC rows [M, C].
// Reference ops replaced: nn.BatchNorm2d / nn.GELU / F.gelu(x + block(x)) in ConvBlock
// (ade_semantic.py:198-210), the trailing BatchNorm2d of DownSample/UpSample (:219,240),
// final_layer BN+ReLU (:285-286) and nn.LayerNorm([64,128,128]) (:281,311).
// Statistics are accumulated in fp64 (sum, sum of squares) so the biased variance is exact to
// fp32 rounding regardless of mean/variance ratio; everything else is fp32 math on T storage.
#include "common.h"
// Nontemporal loads of the streamed operands (bits: 1 bn_act_fwd x, 4 statistics sweeps, 8 bn_bwd_apply, 16 residual operand, 32 per-sample LayerNorm apply passes): the passes
// touch every byte once per launch; `nt` loads bypass the CU's L1 (MI355X_MICROARCH: L2-served).
#ifndef MU_BN_NT
#define MU_BN_NT 57              // whole step 29.88 -> 29.70 ms; with the statistics sweeps too (29): 29.77
#endif
#define MU_LD(bit, vec, ptr_) do { if (MU_BN_NT & (bit)) (vec).load_nt(ptr_); else (vec).load(ptr_); } while (0)
#include "../../include/maskunet_hip.h"

#define MU_STAT_MAXBLK 1024
#ifndef MU_BN_BLK1
#define MU_BN_BLK1 768
#endif
#ifndef MU_BN_U1
#define MU_BN_U1 2
#endif
#ifndef MU_BN_UF
#define MU_BN_UF 2
#endif
#ifndef MU_BN_UA
#define MU_BN_UA 2
#endif
#ifndef MU_BN_STRIDED
#define MU_BN_STRIDED 1            // whole step 28.80 -> 28.71 ms, the backward sweep 0.275 -> 0.268 ms on a 256 MiB tensor
#endif
#ifndef MU_BN_FROWS
#define MU_BN_FROWS 1
#endif
#ifndef MU_BN_OCC1
#define MU_BN_OCC1 1
#endif

// ------------------------------------------------------------------------------------------
// per-channel partial sums over a block of rows.
// MODE 0: (sum x, sum x^2)                                  -> BN forward statistics
// MODE 1: dz = g * act'(pre), writes dz to dzbuf, sums (dz, dz*xhat) -> BN backward
// layout of thread work: tc = channel vector chunk, tr = row lane; LDS [tr][C][2] reduce.
// Each thread keeps U independent 16-byte loads per operand in flight (the sweep is latency-bound otherwise: one load
// per wave covers 1 KB and a CU needs ~64 KB outstanding to saturate HBM), sums the U rows in fp32 and folds that
// short sum into the fp64 accumulators -- U-fold fewer fp64 instructions, same final precision (fp32 sum of <= 8 terms).
// FAST = the fp16-storage GELU (common.h mu_phi_fast); fp32 storage keeps erff.
// ------------------------------------------------------------------------------------------
// ACT / RES (round 5): the activation and the presence of a residual operand as COMPILE-TIME constants (-1 = read the runtime argument).
// With `act` a kernel argument every element pair sat in its own basic block behind two scalar branches (GELU / ReLU / none), the
// residual select cost a v_cndmask + a conversion per element even without a residual, and -- the expensive part in these VALU-bound
// sweeps -- the dependent Horner chains of different pairs could not be interleaved (s_nop between every two packed FMAs).
// MAXT (round 6, MODE 1 with fp32 storage): the block also leaves PER CHANNEL max|dz| and max|xhat| over its rows in
// pmax[(block * C + c) * 2 + {0, 1}] -- the finalize kernel turns them into a per-channel bound of |dx|, the apply pass into the power-of-two
// scale under which it writes dx as ONE fp16 operand (see bn_bwd_final_kernel / bn_bwd_apply_kernel).
template <typename T, int MODE, int ACT = -1, int RES = -1, bool MAXT = false>
__global__ __launch_bounds__(256, MODE == 1 ? MU_BN_OCC1 : 1) void bn_partial_kernel(const T* __restrict__ x, const T* __restrict__ g, const T* __restrict__ res,
                                                         T* __restrict__ dzbuf, long M, int C, long ld,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int act_arg,
                                                         double* __restrict__ part, float* __restrict__ pmax = nullptr) {
    constexpr int N = Vec16<T>::N;
    constexpr int U = MODE == 0 ? 8 : MU_BN_U1;
    constexpr bool FAST = sizeof(T) == 2;
    const int act = ACT >= 0 ? ACT : act_arg;
    const bool has_res = RES >= 0 ? (RES != 0) : (res != nullptr);
    extern __shared__ __attribute__((aligned(16))) double sh[];   // [rpi][C][2]
    const int cv = C / N;
    const int rpi = 256 / cv;
    const int tc = threadIdx.x % cv, tr = threadIdx.x / cv;
#if MU_BN_STRIDED
    // chunks of U * rpi rows dealt round-robin: at any moment the resident blocks read ONE front of the tensor (as the grid-stride apply
    // passes do) instead of gridDim.x separate sequential streams
    const long r0 = (long)blockIdx.x * (U * rpi), r1 = M;
    const long rstride = (long)gridDim.x * (U * rpi);
#else
    const long rows_per_blk = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_blk, r1 = (r0 + rows_per_blk < M ? r0 + rows_per_blk : M);
    const long rstride = (long)U * rpi;
#endif
    double s0[N], s1[N];
    float mxg[N], mxx[N];                                       // MAXT: running max|dz|, max|xhat| of this thread's channels
#pragma unroll
    for (int i = 0; i < N; ++i) { mxg[i] = 0.f; mxx[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < N; ++i) { s0[i] = 0.0; s1[i] = 0.0; }
    if (tr < rpi) {
        const int c = tc * N;
        float mu[N], rs[N], ga[N], be[N];
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) { mu[i] = mean[c + i]; rs[i] = rstd[c + i]; ga[i] = gamma[c + i]; be[i] = beta[c + i]; }
        }
        for (long r = r0 + tr; r < r1; r += rstride) {
            Vec16<T> xv[U], gv[U], rv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long rr = r + (long)u * rpi;
                if (rr < r1) {
                    MU_LD(4, xv[u], x + rr * ld + c);
                    if (MODE == 1) {
                        MU_LD(4, gv[u], g + rr * ld + c);
                        if (has_res) MU_LD(16, rv[u], res + rr * ld + c);
                    }
                } else {
                    xv[u].zero();
                    if (MODE == 1) { gv[u].zero(); rv[u].zero(); }      // g = 0 -> dz = 0: no contribution
                }
            }
            float f0[N], f1[N];
#pragma unroll
            for (int i = 0; i < N; ++i) { f0[i] = 0.f; f1[i] = 0.f; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const float v = xv[u].get(i);
                        f0[i] += v;
                        f1[i] = fmaf(v, v, f1[i]);
                    }
                } else {
                    const long rr = r + (long)u * rpi;
                    Vec16<T> dz;
#pragma unroll
                    for (int i = 0; i < N; i += 2) {
                        // element pairs as 2-vectors: v_pk_add / v_pk_mul / v_pk_fma_f32 (the same IEEE operations per element as the
                        // scalar forms the apply pass mirrors)
                        const mu_f32x2 xh = (mu_f32x2{xv[u].get(i), xv[u].get(i + 1)} - mu_f32x2{mu[i], mu[i + 1]}) * mu_f32x2{rs[i], rs[i + 1]};
                        mu_f32x2 pre = __builtin_elementwise_fma(xh, mu_f32x2{ga[i], ga[i + 1]}, mu_f32x2{be[i], be[i + 1]});
                        if (has_res) pre += mu_f32x2{rv[u].get(i), rv[u].get(i + 1)};
                        float ag[2];
                        mu_act_grad2_t<FAST>(pre[0], pre[1], act, ag[0], ag[1]);
                        const mu_f32x2 dzf = mu_f32x2{gv[u].get(i), gv[u].get(i + 1)} * mu_f32x2{ag[0], ag[1]};
                        dz.set(i, dzf[0]);
                        dz.set(i + 1, dzf[1]);
                        const mu_f32x2 d = {dz.get(i), dz.get(i + 1)};       // the values the apply pass will re-read (rounded to T)
                        if constexpr (MAXT) {
                            if (rr < r1) {                                   // (rows beyond the tensor: xhat of a zero is not data)
                                mxg[i] = fmaxf(mxg[i], fabsf(d[0])); mxg[i + 1] = fmaxf(mxg[i + 1], fabsf(d[1]));
                                mxx[i] = fmaxf(mxx[i], fabsf(xh[0])); mxx[i + 1] = fmaxf(mxx[i + 1], fabsf(xh[1]));
                            }
                        }
                        mu_f32x2 a0 = {f0[i], f0[i + 1]}, a1 = {f1[i], f1[i + 1]};
                        a0 += d;
                        a1 = __builtin_elementwise_fma(d, xh, a1);
                        f0[i] = a0[0]; f0[i + 1] = a0[1];
                        f1[i] = a1[0]; f1[i + 1] = a1[1];
                    }
                    if (dzbuf && rr < r1) dz.store(dzbuf + rr * ld + c);
                }
            }
#pragma unroll
            for (int i = 0; i < N; ++i) { s0[i] += (double)f0[i]; s1[i] += (double)f1[i]; }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            sh[((long)tr * C + c + i) * 2 + 0] = s0[i];
            sh[((long)tr * C + c + i) * 2 + 1] = s1[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < rpi; ++k) { a += sh[((long)k * C + c) * 2]; b += sh[((long)k * C + c) * 2 + 1]; }
        part[((long)blockIdx.x * C + c) * 2] = a;
        part[((long)blockIdx.x * C + c) * 2 + 1] = b;
    }
    if constexpr (MAXT) {                                       // max is exact and order-free: any reduction order gives the same bits
        __syncthreads();                                        // the sums above have been read out of `sh`
        float* shm = reinterpret_cast<float*>(sh);              // [rpi][C][2] floats
        if (tr < rpi) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                shm[((long)tr * C + tc * N + i) * 2 + 0] = mxg[i];
                shm[((long)tr * C + tc * N + i) * 2 + 1] = mxx[i];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < rpi; ++k) { a = fmaxf(a, shm[((long)k * C + c) * 2]); b = fmaxf(b, shm[((long)k * C + c) * 2 + 1]); }
            *reinterpret_cast<float2*>(pmax + ((long)blockIdx.x * C + c) * 2) = make_float2(a, b);
        }
    }
}

// BN forward finalize: mean, rstd (biased var) + running-stat update (unbiased var, momentum)
// FROWS: `part` holds the conv epilogue's FLOAT statistics rows themselves ([row][C][2], nblk = rows <= MU_STAT_MAXBLK) instead of the
// fold kernel's double partials -- the small layers (16^2 / 32^2: 256-1024 rows) skip the fold launch (round 5: 20 launches per step)
template <bool FROWS = false>
__global__ void bn_fwd_final_kernel(const void* __restrict__ part_, int nblk, int C, long M, float eps, float momentum,
                                    float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                    float* __restrict__ running_var, int c_valid, long* __restrict__ num_batches_tracked) {
    const double* part = reinterpret_cast<const double*>(part_);
    const float* fpart = reinterpret_cast<const float*>(part_);
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;   // nn.BatchNorm2d's step counter
    // one wave per channel: lanes stride over the partial blocks, then a wave reduction
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    // all of a lane's (<= MU_STAT_MAXBLK / 64 = 16) partial pairs are requested before the first is summed: the rolled loop paid one
    // dependent L2 round trip per 64 blocks (12 in a row at 768 blocks = most of the launch's 5.8 us); same summation order as before
    double sv[MU_STAT_MAXBLK / 64], qv[MU_STAT_MAXBLK / 64];
#pragma unroll
    for (int u = 0; u < MU_STAT_MAXBLK / 64; ++u) {
        const int b = lane + 64 * u;
        if constexpr (FROWS) {
            const float2 v = b < nblk ? *reinterpret_cast<const float2*>(fpart + ((long)b * C + c) * 2) : make_float2(0.f, 0.f);
            sv[u] = (double)v.x; qv[u] = (double)v.y;
        } else {
            const double2 v = b < nblk ? *reinterpret_cast<const double2*>(part + ((long)b * C + c) * 2) : make_double2(0.0, 0.0);
            sv[u] = v.x; qv[u] = v.y;
        }
    }
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int u = 0; u < MU_STAT_MAXBLK / 64; ++u) { s += sv[u]; q += qv[u]; }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if (lane) return;
    double m = s / (double)M;
    double var = q / (double)M - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean && c < c_valid) {
        double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
}

// eval mode: (mean, rstd) from running statistics; padded channels get (0, 1)
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd, int C, int c_valid) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c < c_valid) { mean[c] = running_mean[c]; rstd[c] = 1.0f / sqrtf(running_var[c] + eps); }
    else { mean[c] = 0.f; rstd[c] = 1.f; }
}

// BN backward finalize: dgamma = sum dz*xhat, dbeta = sum dz; s1 = dbeta/M, s2 = dgamma/M (0 in eval)
__global__ void bn_bwd_final_kernel(const double* __restrict__ part, int nblk, int C, long M, int training,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ s1, float* __restrict__ s2,
                                    const float* __restrict__ xscale = nullptr, const float* __restrict__ pc2 = nullptr,
                                    const float* __restrict__ pc1 = nullptr, float* __restrict__ pair_grads = nullptr,
                                    const float* __restrict__ pmax = nullptr, const float* __restrict__ gamma = nullptr,
                                    const float* __restrict__ rstd = nullptr, float* __restrict__ bound = nullptr) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double av[MU_STAT_MAXBLK / 64], bv[MU_STAT_MAXBLK / 64];       // every load in flight before the first add (see bn_fwd_final_kernel)
#pragma unroll
    for (int u = 0; u < MU_STAT_MAXBLK / 64; ++u) {
        const int k = lane + 64 * u;
        const double2 v = k < nblk ? *reinterpret_cast<const double2*>(part + ((long)k * C + c) * 2) : make_double2(0.0, 0.0);
        av[u] = v.x; bv[u] = v.y;
    }
    float2 pv[MU_STAT_MAXBLK / 64];                                // mu_bn_act_bwd_h: this channel's (max|dz|, max|xhat|) per block, requested with the sums
    if (bound) {
#pragma unroll
        for (int u = 0; u < MU_STAT_MAXBLK / 64; ++u) {
            const int k = lane + 64 * u;
            pv[u] = k < nblk ? *reinterpret_cast<const float2*>(pmax + ((long)k * C + c) * 2) : make_float2(0.f, 0.f);
        }
    }
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int u = 0; u < MU_STAT_MAXBLK / 64; ++u) { a += av[u]; b += bv[u]; }
    a = wave_sum_d(a); b = wave_sum_d(b);
    float mg = 0.f, mx = 0.f;
    if (bound) {
#pragma unroll
        for (int u = 0; u < MU_STAT_MAXBLK / 64; ++u) { mg = fmaxf(mg, pv[u].x); mx = fmaxf(mx, pv[u].y); }
        mg = wave_max(mg); mx = wave_max(mx);
    }
    if (lane) return;
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
    if (pair_grads) {                                              // BatchNorm pair (mu_bn_pair_bwd): dgamma2, dgamma1, dbeta1 = 0
        pair_grads[c] = pc2[c] * (float)b;
        pair_grads[C + c] = pc1[c] * (float)b;
        pair_grads[2 * C + c] = 0.f;
    }
    const float s1c = training ? (float)(a / (double)M) : 0.f;
    const float s2c = training ? (float)((xscale ? (double)xscale[c] : 1.0) * b / (double)M) : 0.f;      // xscale: BatchNorm pair (below)
    s1[c] = s1c;
    s2[c] = s2c;
    // dx = gamma rstd (dz - s1 - xhat s2)  =>  |dx| <= |gamma rstd| (max|dz| + |s1| + max|xhat| |s2|) in this channel: the apply pass takes
    // the maximum over the channels and writes dx as fp16(S dx) with the power of two S that puts that maximum into [2^13, 2^14)
    if (bound) bound[c] = fabsf(gamma[c] * rstd[c]) * (mg + fabsf(s1c) + mx * fabsf(s2c));
}

// Elementwise passes: grid-stride over 16-byte vectors with a stride that is a multiple of the vectors per row (ew_grid),
// so a thread's channel chunk -- and its per-channel constants -- never change; U vectors per operand in flight.
// ENC (fp32 storage only, MU_F32X at the entry point): y feeds nothing but a 3x3 convolution, so it is written as that convolution's chunk-encoded
// matrix operand (common.h mu_ench4, the fp16-pair form: a thread's 16-byte vector IS one chunk) -- the separate mu_split_encode_h4 pass
// (read + write) disappears.
// y16 (ENC only, may be NULL): the fp16 ROUNDING of y (= the hi halves of the encoding) as plain rows of C halves -- what the one-term
// weight gradient of the convolution behind reads (mu_conv_wgrad_h1) and the only form of y the backward keeps.
template <typename T, bool ENC = false, int ACT = -1, int RES = -1>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, long M,
                                                         int C, long ld, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int act_arg,
                                                         h16* __restrict__ y16 = nullptr) {
    constexpr int N = Vec16<T>::N;
    constexpr int U = MU_BN_UF;
    constexpr bool FAST = sizeof(T) == 2;
    const int act = ACT >= 0 ? ACT : act_arg;
    const bool has_res = RES >= 0 ? (RES != 0) : (res != nullptr);
    const int cv = C / N;
    const long total = M * cv;
    const long stride = (long)gridDim.x * 256;
    const long first = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(first % cv) * N;
    const long rstep = stride / cv;
    float a[N], b[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        a[i] = rstd[c + i] * gamma[c + i];
        b[i] = beta[c + i] - mean[c + i] * a[i];
    }
    for (long idx = first; idx < total; idx += U * stride) {
        const long r = idx / cv;
        Vec16<T> xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (idx + u * stride < total) {
                MU_LD(1, xv[u], x + (r + u * rstep) * ld + c);
                if (has_res) MU_LD(16, rv[u], res + (r + u * rstep) * ld + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (idx + u * stride < total) {
                Vec16<T> o;
#pragma unroll
                for (int i = 0; i < N; i += 2) {
                    mu_f32x2 pp = __builtin_elementwise_fma(mu_f32x2{xv[u].get(i), xv[u].get(i + 1)}, mu_f32x2{a[i], a[i + 1]}, mu_f32x2{b[i], b[i + 1]});
                    if (has_res) pp += mu_f32x2{rv[u].get(i), rv[u].get(i + 1)};
                    float o0, o1;
                    mu_act2_t<FAST>(pp[0], pp[1], act, o0, o1);
                    o.set(i, o0);
                    o.set(i + 1, o1);
                }
                if constexpr (ENC) {
                    const uint4 e4 = mu_ench4((f32x4){o.get(0), o.get(1), o.get(2), o.get(3)});
                    *reinterpret_cast<uint4*>(y + (r + u * rstep) * ld + c) = e4;
                    if (y16) *reinterpret_cast<uint2*>(y16 + (r + u * rstep) * (long)C + c) = make_uint2(e4.x, e4.y);
                } else {
#if MU_BN_NT & 2
                o.store_nt(y + (r + u * rstep) * ld + c);
#else
                o.store(y + (r + u * rstep) * ld + c);
#endif
                }
            }
        }
    }
}

// dx = gamma*rstd*(dz - s1 - xhat*s2).  RECOMP = false: dz is read from dzbuf (may alias dx) -- the residual form, whose
// d(residual) IS dz and has to be written anyway.  RECOMP = true (no residual): dz = g * act'(pre) is recomputed from x and
// the incoming gradient, so the statistics sweep writes nothing: 5 tensor passes per BatchNorm backward instead of 6.
// ENC (fp32 storage, mu_bn_act_bwd_h): dx -- the dy of the 3x3 convolution in front of this BatchNorm and of nothing else -- is written
// as ONE fp16 operand S * dx (plain fp16 rows of C halves at the start of the dx buffer, row stride C) with S = dy_scale[0] the
// power-of-two the finalize kernel derived: half the bytes, and the convolution's data- and weight-gradient take two MFMAs per product.
template <typename T, bool RECOMP, bool ENC = false, int ACT = -1>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* dzbuf, T* dx, long M, int C, long ld,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int act_arg,
                                                           const float* __restrict__ s1, const float* __restrict__ s2,
                                                           const float* __restrict__ bound = nullptr, float* __restrict__ dy_scale = nullptr) {
    constexpr int N = Vec16<T>::N;
    constexpr int U = MU_BN_UA;
    constexpr bool FAST = sizeof(T) == 2;
    const int act = ACT >= 0 ? ACT : act_arg;
    const int cv = C / N;
    const long total = M * cv;
    const long stride = (long)gridDim.x * 256;
    const long first = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(first % cv) * N;
    const long rstep = stride / cv;
    // dx = k0 + k1 * x + gr * dz with xhat = (x - mu) rs:  k1 = -gr rs s2,  k0 = -gr s1 - k1 mu
    // RECOMP: pre = xhat * gamma + beta, evaluated exactly as in the statistics sweep so both see the same rounded dz
    float gr[N], k0[N], k1[N], mu[N], rsv[N], ga[N], be[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float rs = rstd[c + i];
        gr[i] = gamma[c + i] * rs;
        k1[i] = -gr[i] * rs * s2[c + i];
        k0[i] = -gr[i] * s1[c + i] - k1[i] * mean[c + i];
        mu[i] = mean[c + i]; rsv[i] = rs; ga[i] = gamma[c + i]; be[i] = RECOMP ? beta[c + i] : 0.f;
    }
    float S = 1.f;
    if constexpr (ENC) {
        // every wave derives the same S from the per-channel bounds (<= 512 floats out of L2; maxima only: deterministic); a zero or
        // non-finite bound gives S = 1 (a NaN / inf gradient then surfaces as NaN / inf, never silently)
        float bm = 0.f;
        for (int k = threadIdx.x & 63; k < C; k += 64) bm = fmaxf(bm, bound[k]);
        bm = wave_max(bm);
        const int e = (int)((__float_as_uint(bm) >> 23) & 0xff) - 127;           // floor(log2 bm) of a normal value
        if (bm > 0.f && e > -127 && e < 128) {
            int k = 13 - e;
            k = k < -100 ? -100 : (k > 100 ? 100 : k);
            S = __uint_as_float((uint32_t)(127 + k) << 23);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) { dy_scale[0] = S; dy_scale[1] = 1.0f / S; }
    }
    for (long idx = first; idx < total; idx += U * stride) {
        const long r = idx / cv;
        Vec16<T> xv[U], dz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (idx + u * stride < total) {
                MU_LD(8, xv[u], x + (r + u * rstep) * ld + c);
                MU_LD(8, dz[u], dzbuf + (r + u * rstep) * ld + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (idx + u * stride < total) {
                Vec16<T> o;
#pragma unroll
                for (int i = 0; i < N; i += 2) {
                    const mu_f32x2 xf = {xv[u].get(i), xv[u].get(i + 1)};
                    mu_f32x2 d = {dz[u].get(i), dz[u].get(i + 1)};
                    if (RECOMP) {
                        Vec16<T> t;                         // round dz to T exactly like the statistics sweep did
                        const mu_f32x2 xh = (xf - mu_f32x2{mu[i], mu[i + 1]}) * mu_f32x2{rsv[i], rsv[i + 1]};
                        const mu_f32x2 pre = __builtin_elementwise_fma(xh, mu_f32x2{ga[i], ga[i + 1]}, mu_f32x2{be[i], be[i + 1]});
                        float ag[2];
                        mu_act_grad2_t<FAST>(pre[0], pre[1], act, ag[0], ag[1]);
                        const mu_f32x2 dzf = d * mu_f32x2{ag[0], ag[1]};
                        t.set(i, dzf[0]);
                        t.set(i + 1, dzf[1]);
                        d = mu_f32x2{t.get(i), t.get(i + 1)};
                    }
                    const mu_f32x2 ov = __builtin_elementwise_fma(mu_f32x2{gr[i], gr[i + 1]}, d,
                                                                  __builtin_elementwise_fma(mu_f32x2{k1[i], k1[i + 1]}, xf, mu_f32x2{k0[i], k0[i + 1]}));
                    o.set(i, ov[0]);
                    o.set(i + 1, ov[1]);
                }
                if constexpr (ENC) {
                    static_assert(N == 4, "fp16 dx rows come from fp32 storage");
                    const h16x4 oh = {(h16)(o.get(0) * S), (h16)(o.get(1) * S), (h16)(o.get(2) * S), (h16)(o.get(3) * S)};
                    *reinterpret_cast<h16x4*>(reinterpret_cast<h16*>(dx) + (r + u * rstep) * (long)C + c) = oh;
                } else o.store(dx + (r + u * rstep) * ld + c);
            }
        }
    }
}

static inline int stat_blocks(long M) {
    long b = M / 32;
    return (int)(b < 1 ? 1 : (b > MU_STAT_MAXBLK ? MU_STAT_MAXBLK : b));
}
static inline long mu_gcd(long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a; }
// blocks of 256 threads, ~4 vectors per thread, and 256*grid a multiple of cv (vectors per row): the kernels rely on it
static inline int ew_grid(long total, int cv) {
    const long m = cv / mu_gcd(cv, 256);
    long g = (total + 1023) / 1024;
    g = g < 1 ? 1 : (g > 8192 ? 8192 : g);
    g = (g + m - 1) / m * m;
    return (int)g;
}

__global__ void colsum_final_kernel(const double* __restrict__ part, int nblk, int C, float* __restrict__ out) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += part[((long)b * C + c) * 2];
    s = wave_sum_d(s);
    if (lane == 0) out[c] = (float)s;
}

extern "C" long mu_colsum_workspace_bytes(int C) { return (long)MU_STAT_MAXBLK * C * 2 * sizeof(double); }

// column sums of a CHUNK-ENCODED fp32x tensor (common.h: 16 bytes = [4 bf16 hi | 4 bf16 lo]): thread = (row in the block's iteration,
// chunk column); fp32 sums of up to 64 decoded rows folded into doubles, the block's row-lanes reduced through LDS into the
// [block][C][2] partial layout colsum_final_kernel reads.
__global__ __launch_bounds__(256) void colsum_enc_partial_kernel(const uint4* __restrict__ x, long M, int C, long ld, double* __restrict__ part) {
    const int cv = C / 4, rpi = 256 / cv;
    const int tid = threadIdx.x, cc = tid % cv, lr = tid / cv;
    const long rows_per_blk = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_blk, r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (lr < rpi) {
        f32x4 sh = {0.f, 0.f, 0.f, 0.f};
        int n = 0;
        for (long r = r0 + lr; r < r1; r += 4L * rpi) {      // four rows in flight per thread (a plain HBM stream)
            uint4 e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + (long)u * rpi < r1)
                    e[u] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + ((r + (long)u * rpi) * ld) / 4 + cc));
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + (long)u * rpi < r1) sh += mu_dec4(e[u]);
            if (++n == 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += (double)sh[i];
                sh = (f32x4){0.f, 0.f, 0.f, 0.f};
                n = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += (double)sh[i];
    }
    extern __shared__ double shd[];                          // [rpi][C]
    if (lr < rpi) {
#pragma unroll
        for (int i = 0; i < 4; ++i) shd[lr * C + cc * 4 + i] = acc[i];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double t = 0.0;
        for (int k = 0; k < rpi; ++k) t += shd[k * C + c];
        part[((long)blockIdx.x * C + c) * 2] = t;
    }
}

// out[c] = sum_r x[r][c]: bias gradients.  Same vectorised row sweep as the BN statistics.
extern "C" int mu_colsum(const void* x, long M, int C, long ld, float* out, void* workspace, long ws_bytes, int dtype, void* stream) {
    if (!x || !out || !workspace || M <= 0 || C <= 0 || C % 8 || ld < C) return MU_ERR_ARG;
    if (ws_bytes < mu_colsum_workspace_bytes(C)) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = stat_blocks(M);
    if (dtype == MU_F32) {
        const int cv = C / 4;
        if (cv > 256) return MU_ERR_SHAPE;
        bn_partial_kernel<float, 0><<<nblk, 256, (size_t)(256 / cv) * C * 2 * sizeof(double), st>>>((const float*)x, nullptr, nullptr, nullptr, M, C, ld, nullptr, nullptr, nullptr, nullptr, 0, (double*)workspace);
    } else if (dtype == MU_F16) {
        const int cv = C / 8;
        if (cv > 256) return MU_ERR_SHAPE;
        bn_partial_kernel<h16, 0><<<nblk, 256, (size_t)(256 / cv) * C * 2 * sizeof(double), st>>>((const h16*)x, nullptr, nullptr, nullptr, M, C, ld, nullptr, nullptr, nullptr, nullptr, 0, (double*)workspace);
    } else if (dtype == MU_F32X) {                           // x chunk-encoded (mu_split_encode form, or written so by mu_attn_bwd_phases)
        const int cv = C / 4;
        if (cv > 256 || ld % 4) return MU_ERR_SHAPE;
        colsum_enc_partial_kernel<<<nblk, 256, (size_t)(256 / cv) * C * sizeof(double), st>>>((const uint4*)x, M, C, ld, (double*)workspace);
    } else return MU_ERR_ARG;
    colsum_final_kernel<<<mu_cdiv(C, 4), 256, 0, st>>>((const double*)workspace, nblk, C, out);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// fp64 partial slab | s1, s2 (the apply pass's per-channel means) | A (mu_bn_pair_bwd: the single-layer dgamma nobody outside reads)
// ... | bound[C] | per-block, per-channel (max|dz|, max|xhat|) of the backward statistics sweep (mu_bn_act_bwd_h)
extern "C" long mu_bn_workspace_bytes(int C) {
    return (long)MU_STAT_MAXBLK * C * 2 * sizeof(double) + 4L * C * sizeof(float) + (long)MU_STAT_MAXBLK * C * 2 * sizeof(float);
}

template <typename T>
static int bn_train_stats_t(const T* x, long M, int C, long ld, float* mean, float* rstd, float* rmean, float* rvar, long* nbt,
                            int c_valid, float momentum, float eps, void* ws, hipStream_t st) {
    constexpr int N = Vec16<T>::N;
    int cv = C / N;
    if (cv > 256) return MU_ERR_SHAPE;
    int rpi = 256 / cv;
    int nblk = stat_blocks(M);
    size_t lds = (size_t)rpi * C * 2 * sizeof(double);
    bn_partial_kernel<T, 0><<<nblk, 256, lds, st>>>(x, nullptr, nullptr, nullptr, M, C, ld, nullptr, nullptr, nullptr, nullptr, 0, (double*)ws);
    bn_fwd_final_kernel<false><<<mu_cdiv(C, 4), 256, 0, st>>>((const double*)ws, nblk, C, M, eps, momentum, mean, rstd, rmean, rvar, c_valid, nbt);
    return MU_OK;
}

extern "C" int mu_bn_train_stats(const void* x, long M, int C, long ld, float* mean, float* rstd, float* running_mean,
                                 float* running_var, long* num_batches_tracked, int c_valid, float momentum, float eps,
                                 void* workspace, long ws_bytes, int dtype, void* stream) {
    if (!x || !mean || !rstd || !workspace || M <= 0 || C <= 0 || C % 8 || ld < C) return MU_ERR_ARG;
    if (ws_bytes < mu_bn_workspace_bytes(C)) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == MU_F32) rc = bn_train_stats_t<float>((const float*)x, M, C, ld, mean, rstd, running_mean, running_var, num_batches_tracked, c_valid, momentum, eps, workspace, st);
    else if (dtype == MU_F16) rc = bn_train_stats_t<h16>((const h16*)x, M, C, ld, mean, rstd, running_mean, running_var, num_batches_tracked, c_valid, momentum, eps, workspace, st);
    else return MU_ERR_ARG;
    if (rc) return rc;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// Statistics from the conv epilogue's per-tile rows: part[rows][C][2] floats (sum, sum of squares of the fp16 outputs).
// Stage 1 folds 16+ rows per block into the fp64 partial layout of the in-kernel path (coalesced: consecutive threads read
// consecutive floats of a row), stage 2 is the regular finalize.
__global__ __launch_bounds__(256) void bn_fold_rows_kernel(const float* __restrict__ part, int rows, int rpb, int C2, double* __restrict__ out) {
    const int r0 = blockIdx.x * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
    for (int c = threadIdx.x; c < C2; c += 256) {
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int r = r0;
        for (; r + 7 < r1; r += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += (double)part[(long)(r + u) * C2 + c];
        }
        for (; r < r1; ++r) a[0] += (double)part[(long)r * C2 + c];
        out[(long)blockIdx.x * C2 + c] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
}

extern "C" int mu_bn_train_stats_rows(const float* stat_part, int rows, long M, int C, float* mean, float* rstd, float* running_mean,
                                      float* running_var, long* num_batches_tracked, int c_valid, float momentum, float eps,
                                      void* workspace, long ws_bytes, void* stream) {
    if (!stat_part || !mean || !rstd || !workspace || rows <= 0 || M <= 0 || C <= 0) return MU_ERR_ARG;
    // many short blocks (the fold is latency-bound): 16 rows each, more only when that would exceed the partial-slab capacity
    int rpb = 16;
    if ((rows + rpb - 1) / rpb > MU_STAT_MAXBLK) rpb = ((rows + MU_STAT_MAXBLK - 1) / MU_STAT_MAXBLK + 7) / 8 * 8;
    const int nblk = (rows + rpb - 1) / rpb;
    if (ws_bytes < mu_bn_workspace_bytes(C)) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (MU_BN_FROWS && rows <= MU_STAT_MAXBLK) {        // few rows: the finalize kernel sums them itself (one launch less)
        bn_fwd_final_kernel<true><<<mu_cdiv(C, 4), 256, 0, st>>>(stat_part, rows, C, M, eps, momentum, mean, rstd, running_mean, running_var,
                                                                 c_valid, num_batches_tracked);
        MU_CHECK_LAUNCH();
        return MU_OK;
    }
    bn_fold_rows_kernel<<<nblk, 256, 0, st>>>(stat_part, rows, rpb, 2 * C, (double*)workspace);
    bn_fwd_final_kernel<false><<<mu_cdiv(C, 4), 256, 0, st>>>((const double*)workspace, nblk, C, M, eps, momentum, mean, rstd, running_mean,
                                                        running_var, c_valid, num_batches_tracked);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean, float* rstd, int C,
                                int c_valid, void* stream) {
    if (!running_mean || !running_var || !mean || !rstd || C <= 0) return MU_ERR_ARG;
    bn_eval_stats_kernel<<<mu_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(running_mean, running_var, eps, mean, rstd, C, c_valid);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// Inference: BatchNorm2d with running statistics is a per-channel affine map, and two of them back to back (DownSample / UpSample tails,
// ade_semantic.py:218-219,239-240) compose into one.  (scale, shift) for the conv epilogue y = act(conv * scale + shift + res):
//   a1 = gamma1 / sqrt(var1 + eps1), s1 = beta1 + (conv_bias - mean1) * a1;  second layer: a2 likewise, scale = a1 a2, shift = (s1 - mean2) a2 + beta2.
// Padded channels get (0, 0) so they stay exact zeros.
__global__ void bn_eval_fold_kernel(const float* __restrict__ rm1, const float* __restrict__ rv1, const float* __restrict__ g1,
                                    const float* __restrict__ b1, float eps1, const float* __restrict__ rm2, const float* __restrict__ rv2,
                                    const float* __restrict__ g2, const float* __restrict__ b2, float eps2, const float* __restrict__ conv_bias,
                                    float* __restrict__ scale, float* __restrict__ shift, int C, int c_valid) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c >= c_valid) { scale[c] = 0.f; shift[c] = 0.f; return; }
    float a = (g1 ? g1[c] : 1.f) / sqrtf(rv1[c] + eps1);
    float sh = (b1 ? b1[c] : 0.f) + ((conv_bias ? conv_bias[c] : 0.f) - rm1[c]) * a;
    if (rm2) {
        const float a2 = (g2 ? g2[c] : 1.f) / sqrtf(rv2[c] + eps2);
        sh = (sh - rm2[c]) * a2 + (b2 ? b2[c] : 0.f);
        a *= a2;
    }
    scale[c] = a;
    shift[c] = sh;
}

extern "C" int mu_bn_eval_fold(const float* running_mean1, const float* running_var1, const float* gamma1, const float* beta1, float eps1,
                               const float* running_mean2, const float* running_var2, const float* gamma2, const float* beta2, float eps2,
                               const float* conv_bias, float* scale, float* shift, int C, int c_valid, void* stream) {
    if (!running_mean1 || !running_var1 || !scale || !shift || C <= 0 || c_valid <= 0 || c_valid > C) return MU_ERR_ARG;
    if ((running_mean2 != nullptr) != (running_var2 != nullptr)) return MU_ERR_ARG;
    bn_eval_fold_kernel<<<mu_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(running_mean1, running_var1, gamma1, beta1, eps1, running_mean2, running_var2,
                                                                       gamma2, beta2, eps2, conv_bias, scale, shift, C, c_valid);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// one instantiation per (activation, residual or not): see bn_partial_kernel
#define MU_BN_ACT_SWITCH(act, CALL)                              \
    switch (act) {                                               \
        case MU_ACT_GELU: { constexpr int A_ = MU_ACT_GELU; CALL; } break; \
        case MU_ACT_RELU: { constexpr int A_ = MU_ACT_RELU; CALL; } break; \
        default:          { constexpr int A_ = MU_ACT_NONE; CALL; } break; \
    }
template <typename T, bool ENC>
static void bn_act_fwd_launch(const T* x, const T* res, T* y, long M, int C, long ld, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, hipStream_t st, h16* y16 = nullptr) {
    constexpr int N = Vec16<T>::N;
    const int grid = ew_grid(M * (C / N), C / N);
    if (res) { MU_BN_ACT_SWITCH(act, (bn_act_fwd_kernel<T, ENC, A_, 1><<<grid, 256, 0, st>>>(x, res, y, M, C, ld, mean, rstd, gamma, beta, act, y16))) }
    else     { MU_BN_ACT_SWITCH(act, (bn_act_fwd_kernel<T, ENC, A_, 0><<<grid, 256, 0, st>>>(x, res, y, M, C, ld, mean, rstd, gamma, beta, act, y16))) }
}

// fp32x: mu_bn_act_fwd(MU_F32X) with the second output y16 (M rows of C halves, row stride C; contiguous input rows: ld == C)
extern "C" int mu_bn_act_fwd_enc(const void* x, const void* res, void* y_enc, void* y16, long M, int C, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, int act, void* stream) {
    if (!x || !y_enc || !y16 || !mean || !rstd || !gamma || !beta || M <= 0 || C <= 0 || C % 8) return MU_ERR_ARG;
    if (act != MU_ACT_NONE && act != MU_ACT_GELU && act != MU_ACT_RELU) return MU_ERR_ARG;
    bn_act_fwd_launch<float, true>((const float*)x, (const float*)res, (float*)y_enc, M, C, C, mean, rstd, gamma, beta, act, (hipStream_t)stream, (h16*)y16);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_bn_act_fwd(const void* x, const void* res, void* y, long M, int C, long ld, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, int act, int dtype, void* stream) {
    if (!x || !y || !mean || !rstd || !gamma || !beta || M <= 0 || C <= 0 || C % 8 || ld < C) return MU_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (act != MU_ACT_NONE && act != MU_ACT_GELU && act != MU_ACT_RELU) return MU_ERR_ARG;
    if (dtype == MU_F32)
        bn_act_fwd_launch<float, false>((const float*)x, (const float*)res, (float*)y, M, C, ld, mean, rstd, gamma, beta, act, st);
    else if (dtype == MU_F32X)      // fp32 storage, y written chunk-encoded (it only feeds a convolution in the fp32x mode)
        bn_act_fwd_launch<float, true>((const float*)x, (const float*)res, (float*)y, M, C, ld, mean, rstd, gamma, beta, act, st);
    else if (dtype == MU_F16)
        bn_act_fwd_launch<h16, false>((const h16*)x, (const h16*)res, (h16*)y, M, C, ld, mean, rstd, gamma, beta, act, st);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

template <typename T>
static int bn_act_bwd_t(const T* x, const T* res, const T* g, T* dx, T* dres, long M, int C, long ld, const float* mean,
                        const float* rstd, const float* gamma, const float* beta, int act, int training, float* dgamma,
                        float* dbeta, void* ws, hipStream_t st, const float* xscale = nullptr, bool enc = false, const float* pc2 = nullptr,
                        const float* pc1 = nullptr, float* pair_grads = nullptr, float* dy_scale = nullptr) {
    constexpr int N = Vec16<T>::N;
    int cv = C / N;
    if (cv > 256) return MU_ERR_SHAPE;
    int rpi = 256 / cv;
    // the backward statistics kernel holds 3 blocks per CU (136 VGPRs): 768 resident blocks = one whole round (1024 would leave a
    // second round one third full)
    int nblk = stat_blocks(M);
    if (nblk > MU_BN_BLK1) nblk = MU_BN_BLK1;
    size_t lds = (size_t)rpi * C * 2 * sizeof(double);
    double* part = (double*)ws;
    float* s1 = (float*)((char*)ws + (size_t)MU_STAT_MAXBLK * C * 2 * sizeof(double));
    float* s2 = s1 + C;
    float* bound = s1 + 3 * (size_t)C;
    float* pmax = bound + C;
    if (act != MU_ACT_NONE && act != MU_ACT_GELU && act != MU_ACT_RELU) return MU_ERR_ARG;
    if (enc && (sizeof(T) != 4 || !dy_scale || ld != C)) return MU_ERR_ARG;      // fp16 dx rows: fp32 storage, contiguous rows
    if (res) {                      // d(residual) == dz exactly, so it doubles as the dz buffer
        if constexpr (sizeof(T) == 4) {
            if (enc) {
                MU_BN_ACT_SWITCH(act, (bn_partial_kernel<T, 1, A_, 1, true><<<nblk, 256, lds, st>>>(x, g, res, dres, M, C, ld, mean, rstd, gamma, beta, act, part, pmax)))
                bn_bwd_final_kernel<<<mu_cdiv(C, 4), 256, 0, st>>>(part, nblk, C, M, training, dgamma, dbeta, s1, s2, xscale, nullptr, nullptr, nullptr, pmax, gamma, rstd, bound);
                bn_bwd_apply_kernel<T, false, true><<<ew_grid(M * cv, cv), 256, 0, st>>>(x, dres, dx, M, C, ld, mean, rstd, gamma, beta, act, s1, s2, bound, dy_scale);
                return MU_OK;
            }
        }
        MU_BN_ACT_SWITCH(act, (bn_partial_kernel<T, 1, A_, 1><<<nblk, 256, lds, st>>>(x, g, res, dres, M, C, ld, mean, rstd, gamma, beta, act, part)))
        bn_bwd_final_kernel<<<mu_cdiv(C, 4), 256, 0, st>>>(part, nblk, C, M, training, dgamma, dbeta, s1, s2, xscale);
        bn_bwd_apply_kernel<T, false><<<ew_grid(M * cv, cv), 256, 0, st>>>(x, dres, dx, M, C, ld, mean, rstd, gamma, beta, act, s1, s2);
    } else {                        // no residual: nothing is written by the statistics sweep, dz is recomputed in the apply pass
        if constexpr (sizeof(T) == 4) {
            if (enc) {
                MU_BN_ACT_SWITCH(act, (bn_partial_kernel<T, 1, A_, 0, true><<<nblk, 256, lds, st>>>(x, g, nullptr, nullptr, M, C, ld, mean, rstd, gamma, beta, act, part, pmax)))
                bn_bwd_final_kernel<<<mu_cdiv(C, 4), 256, 0, st>>>(part, nblk, C, M, training, dgamma, dbeta, s1, s2, xscale, pc2, pc1, pair_grads, pmax, gamma, rstd, bound);
                MU_BN_ACT_SWITCH(act, (bn_bwd_apply_kernel<T, true, true, A_><<<ew_grid(M * cv, cv), 256, 0, st>>>(x, g, dx, M, C, ld, mean, rstd, gamma, beta, act, s1, s2, bound, dy_scale)))
                return MU_OK;
            }
        }
        MU_BN_ACT_SWITCH(act, (bn_partial_kernel<T, 1, A_, 0><<<nblk, 256, lds, st>>>(x, g, nullptr, nullptr, M, C, ld, mean, rstd, gamma, beta, act, part)))
        bn_bwd_final_kernel<<<mu_cdiv(C, 4), 256, 0, st>>>(part, nblk, C, M, training, dgamma, dbeta, s1, s2, xscale, pc2, pc1, pair_grads);
        MU_BN_ACT_SWITCH(act, (bn_bwd_apply_kernel<T, true, false, A_><<<ew_grid(M * cv, cv), 256, 0, st>>>(x, g, dx, M, C, ld, mean, rstd, gamma, beta, act, s1, s2)))
    }
    return MU_OK;
}

extern "C" int mu_bn_act_bwd_scaled(const void* x, const void* res, const void* grad_out, void* dx, void* dres, long M, int C, long ld,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta, int act, int training,
                                    float* dgamma, float* dbeta, const float* xhat_scale, void* workspace, long ws_bytes, int dtype,
                                    void* stream);

extern "C" int mu_bn_act_bwd(const void* x, const void* res, const void* grad_out, void* dx, void* dres, long M, int C, long ld,
                             const float* mean, const float* rstd, const float* gamma, const float* beta, int act, int training,
                             float* dgamma, float* dbeta, void* workspace, long ws_bytes, int dtype, void* stream) {
    return mu_bn_act_bwd_scaled(x, res, grad_out, dx, dres, M, C, ld, mean, rstd, gamma, beta, act, training, dgamma, dbeta, nullptr,
                                workspace, ws_bytes, dtype, stream);
}

// ------------------------------------------------------------------------------------------
// BatchNorm pair.  DownSample / UpSample end with nn.BatchNorm2d directly behind the last BatchNorm2d of their ConvBlock
// (ade_semantic.py:216-219, 237-240).  In training mode the second layer's batch statistics follow from the first's:
//   y1 = gamma1*u + beta1 (u = xhat of the conv output z, sum u = 0, mean u^2 = var1/(var1+eps1) =: q)
//   => mean(y1) = beta1, biased var(y1) = gamma1^2 * q, r2 = 1/sqrt(gamma1^2*q + eps2), y2 = gamma2*gamma1*r2*u + beta2.
// So the pair is ONE normalisation of z with gamma_eff = gamma1*gamma2*r2 and beta2 (mu_bn_act_fwd), and its backward is the single-layer
// formula with the xhat term scaled by k = r2^2*(gamma1^2 + eps2) (mu_bn_act_bwd_scaled):
//   dz = gamma_eff*r1*(g - mean g - u*k*mean(g*u));  dbeta2 = sum g;  dgamma2 = gamma1*r2*A;  dgamma1 = gamma2*r2^3*eps2*A;  dbeta1 = 0
// with A = sum g*u (what mu_bn_act_bwd_scaled returns in dgamma).  One statistics sweep, one apply pass and their backward less.
// ------------------------------------------------------------------------------------------
__global__ void bn_pair_compose_kernel(const float* __restrict__ rstd1, const float* __restrict__ gamma1, const float* __restrict__ beta1,
                                       const float* __restrict__ gamma2, int C, int c_valid, long M, float eps1, float eps2, float momentum2,
                                       float* __restrict__ running_mean2, float* __restrict__ running_var2, long* __restrict__ nbt2,
                                       float* __restrict__ gamma_eff, float* __restrict__ xhat_scale, float* __restrict__ dgamma2_coef,
                                       float* __restrict__ dgamma1_coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (nbt2 && c == 0) *nbt2 += 1;
    if (c >= C) return;
    const double r1 = (double)rstd1[c], g1 = (double)gamma1[c], g2 = (double)gamma2[c];
    double q = 1.0 - (double)eps1 * r1 * r1;          // var1 / (var1 + eps1)
    if (q < 0.0) q = 0.0;
    const double var2 = g1 * g1 * q;
    const double r2 = 1.0 / sqrt(var2 + (double)eps2);
    gamma_eff[c] = (float)(g1 * g2 * r2);
    xhat_scale[c] = (float)(r2 * r2 * (g1 * g1 + (double)eps2));
    dgamma2_coef[c] = (float)(g1 * r2);
    dgamma1_coef[c] = (float)(g2 * r2 * r2 * r2 * (double)eps2);
    if (running_mean2 && c < c_valid) {
        const double unb = M > 1 ? var2 * ((double)M / (double)(M - 1)) : var2;
        running_mean2[c] = (float)((1.0 - (double)momentum2) * (double)running_mean2[c] + (double)momentum2 * (double)beta1[c]);
        running_var2[c] = (float)((1.0 - (double)momentum2) * (double)running_var2[c] + (double)momentum2 * unb);
    }
}

extern "C" int mu_bn_pair_compose(const float* rstd1, const float* gamma1, const float* beta1, const float* gamma2, int C, int c_valid,
                                  long M, float eps1, float eps2, float momentum2, float* running_mean2, float* running_var2,
                                  long* num_batches_tracked2, float* gamma_eff, float* xhat_scale, float* dgamma2_coef,
                                  float* dgamma1_coef, void* stream) {
    if (!rstd1 || !gamma1 || !beta1 || !gamma2 || !gamma_eff || !xhat_scale || !dgamma2_coef || !dgamma1_coef || C <= 0 || M <= 0)
        return MU_ERR_ARG;
    if ((running_mean2 != nullptr) != (running_var2 != nullptr)) return MU_ERR_ARG;
    bn_pair_compose_kernel<<<mu_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(rstd1, gamma1, beta1, gamma2, C, c_valid, M, eps1, eps2, momentum2,
                                                                          running_mean2, running_var2, num_batches_tracked2, gamma_eff,
                                                                          xhat_scale, dgamma2_coef, dgamma1_coef);
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_bn_act_bwd_scaled(const void* x, const void* res, const void* grad_out, void* dx, void* dres, long M, int C, long ld,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta, int act, int training,
                                    float* dgamma, float* dbeta, const float* xhat_scale, void* workspace, long ws_bytes, int dtype,
                                    void* stream) {
    if (!x || !grad_out || !dx || !mean || !rstd || !gamma || !beta || !dgamma || !dbeta || !workspace) return MU_ERR_ARG;
    if ((res != nullptr) != (dres != nullptr)) return MU_ERR_ARG;
    if (M <= 0 || C <= 0 || C % 8 || ld < C) return MU_ERR_ARG;
    if (ws_bytes < mu_bn_workspace_bytes(C)) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == MU_F32)
        rc = bn_act_bwd_t<float>((const float*)x, (const float*)res, (const float*)grad_out, (float*)dx, (float*)dres, M, C, ld, mean, rstd, gamma, beta, act, training, dgamma, dbeta, workspace, st, xhat_scale);
    else if (dtype == MU_F16)
        rc = bn_act_bwd_t<h16>((const h16*)x, (const h16*)res, (const h16*)grad_out, (h16*)dx, (h16*)dres, M, C, ld, mean, rstd, gamma, beta, act, training, dgamma, dbeta, workspace, st, xhat_scale);
    else return MU_ERR_ARG;
    if (rc) return rc;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_bn_pair_bwd(const void* x, const void* grad_out, void* dx, long M, int C, long ld, const float* mean, const float* rstd,
                              const float* gamma_eff, const float* beta2, const float* xhat_scale, const float* dgamma2_coef,
                              const float* dgamma1_coef, float* pair_grads, float* dbeta2, void* workspace, long ws_bytes, int dtype,
                              void* stream) {
    if (!x || !grad_out || !dx || !mean || !rstd || !gamma_eff || !beta2 || !xhat_scale || !dgamma2_coef || !dgamma1_coef || !pair_grads ||
        !dbeta2 || !workspace)
        return MU_ERR_ARG;
    if (M <= 0 || C <= 0 || C % 8 || ld < C) return MU_ERR_ARG;
    if (ws_bytes < mu_bn_workspace_bytes(C)) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* A = (float*)((char*)workspace + (size_t)MU_STAT_MAXBLK * C * 2 * sizeof(double)) + 2 * (size_t)C;
    int rc;
    if (dtype == MU_F32)
        rc = bn_act_bwd_t<float>((const float*)x, nullptr, (const float*)grad_out, (float*)dx, nullptr, M, C, ld, mean, rstd, gamma_eff, beta2, MU_ACT_NONE, 1, A, dbeta2, workspace, st, xhat_scale, false, dgamma2_coef, dgamma1_coef, pair_grads);
    else if (dtype == MU_F16)
        rc = bn_act_bwd_t<h16>((const h16*)x, nullptr, (const h16*)grad_out, (h16*)dx, nullptr, M, C, ld, mean, rstd, gamma_eff, beta2, MU_ACT_NONE, 1, A, dbeta2, workspace, st, xhat_scale, false, dgamma2_coef, dgamma1_coef, pair_grads);
    else return MU_ERR_ARG;
    if (rc) return rc;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// fp32x (round 6): the same backward passes with dx written as ONE power-of-two-scaled fp16 operand -- dx is the dy of the 3x3
// convolution in front of the BatchNorm (ConvBlock wiring, ade_semantic.py:199-204) and of nothing else.  dx_h: M rows of C halves
// (row stride C) = the first half of a buffer sized like x; dy_scale: two floats {S, 1 / S} written on the device (no host sync) and
// handed to mu_conv_dgrad_h / mu_conv_wgrad_h.  fp32 storage, contiguous rows (ld == C).  dres (residual form) stays plain fp32.
// The scale: the statistics sweep also collects max|dz| and max|xhat| per channel, the finalize kernel bounds |dx| per channel
// (|gamma rstd| (max|dz| + |s1| + max|xhat| |s2|): tight to ~2 bits, and a padded or degenerate channel -- rstd = 1 / sqrt(eps), dz = 0 --
// bounds itself with 0 instead of inflating a global product), the apply pass takes the maximum.  Maxima only: bit-reproducible.
extern "C" int mu_bn_act_bwd_h(const void* x, const void* res, const void* grad_out, void* dx_h, void* dres, long M, int C,
                               const float* mean, const float* rstd, const float* gamma, const float* beta, int act, int training,
                               float* dgamma, float* dbeta, float* dy_scale, void* workspace, long ws_bytes, void* stream) {
    if (!x || !grad_out || !dx_h || !mean || !rstd || !gamma || !beta || !dgamma || !dbeta || !dy_scale || !workspace) return MU_ERR_ARG;
    if ((res != nullptr) != (dres != nullptr)) return MU_ERR_ARG;
    if (M <= 0 || C <= 0 || C % 8) return MU_ERR_ARG;
    if (ws_bytes < mu_bn_workspace_bytes(C)) return MU_ERR_WORKSPACE;
    int rc = bn_act_bwd_t<float>((const float*)x, (const float*)res, (const float*)grad_out, (float*)dx_h, (float*)dres, M, C, C, mean, rstd, gamma,
                                 beta, act, training, dgamma, dbeta, workspace, (hipStream_t)stream, nullptr, true, nullptr, nullptr, nullptr, dy_scale);
    if (rc) return rc;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

extern "C" int mu_bn_pair_bwd_h(const void* x, const void* grad_out, void* dx_h, long M, int C, const float* mean, const float* rstd,
                                const float* gamma_eff, const float* beta2, const float* xhat_scale, const float* dgamma2_coef,
                                const float* dgamma1_coef, float* pair_grads, float* dbeta2, float* dy_scale, void* workspace, long ws_bytes,
                                void* stream) {
    if (!x || !grad_out || !dx_h || !mean || !rstd || !gamma_eff || !beta2 || !xhat_scale || !dgamma2_coef || !dgamma1_coef || !pair_grads ||
        !dbeta2 || !dy_scale || !workspace)
        return MU_ERR_ARG;
    if (M <= 0 || C <= 0 || C % 8) return MU_ERR_ARG;
    if (ws_bytes < mu_bn_workspace_bytes(C)) return MU_ERR_WORKSPACE;
    float* A = (float*)((char*)workspace + (size_t)MU_STAT_MAXBLK * C * 2 * sizeof(double)) + 2 * (size_t)C;
    int rc = bn_act_bwd_t<float>((const float*)x, nullptr, (const float*)grad_out, (float*)dx_h, nullptr, M, C, C, mean, rstd, gamma_eff, beta2,
                                 MU_ACT_NONE, 1, A, dbeta2, workspace, (hipStream_t)stream, xhat_scale, true, dgamma2_coef, dgamma1_coef, pair_grads,
                                 dy_scale);
    if (rc) return rc;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

// ------------------------------------------------------------------------------------------
// per-sample LayerNorm over L = C*H*W elements with a full-shape affine (nn.LayerNorm([64,H,W])).
// x, y, dy, dx: [B, L] in T;  w, b, dw, db: fp32 [L].
// ------------------------------------------------------------------------------------------
#define LN_CHUNKS 128     // partial-sum blocks per sample

// MODE 0: (sum x, sum x^2);  MODE 1: (sum dy*w, sum dy*w*xhat)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void lns_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ w,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd, long L,
                                                          double* __restrict__ part) {
    constexpr int N = Vec16<T>::N;
    const int b = blockIdx.y;
    const long nvec = L / N;
    const long per = (nvec + gridDim.x - 1) / gridDim.x;
    const long v0 = (long)blockIdx.x * per, v1 = (v0 + per < nvec ? v0 + per : nvec);
    const T* xb = x + (long)b * L;
    double s0 = 0.0, s1 = 0.0;
    float mu = 0.f, rs = 0.f;
    if (MODE == 1) { mu = mean[b]; rs = rstd[b]; }
    for (long v = v0 + threadIdx.x; v < v1; v += 256) {
        Vec16<T> xv;
        xv.load(xb + v * N);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) { double t = (double)xv.get(i); s0 += t; s1 += t * t; }
        } else {
            Vec16<T> gv;
            gv.load(dy + (long)b * L + v * N);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float gw = gv.get(i) * w[v * N + i];
                float xh = (xv.get(i) - mu) * rs;
                s0 += (double)gw;
                s1 += (double)gw * (double)xh;
            }
        }
    }
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    __shared__ double sh[8];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[wid * 2] = s0; sh[wid * 2 + 1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = sh[0] + sh[2] + sh[4] + sh[6], c = sh[1] + sh[3] + sh[5] + sh[7];
        part[((long)b * gridDim.x + blockIdx.x) * 2] = a;
        part[((long)b * gridDim.x + blockIdx.x) * 2 + 1] = c;
    }
}

__global__ void lns_final_kernel(const double* __restrict__ part, int nchunk, long L, float eps, int mode, float* __restrict__ o0,
                                 float* __restrict__ o1, int B) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    double a = 0.0, c = 0.0;
    for (int k = lane; k < nchunk; k += 64) { a += part[((long)b * nchunk + k) * 2]; c += part[((long)b * nchunk + k) * 2 + 1]; }
    a = wave_sum_d(a); c = wave_sum_d(c);
    if (lane) return;
    if (mode == 0) {
        double m = a / (double)L, var = c / (double)L - m * m;
        if (var < 0.0) var = 0.0;
        o0[b] = (float)m;
        o1[b] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
        o0[b] = (float)(a / (double)L);
        o1[b] = (float)(c / (double)L);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void lns_fwd_apply_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ y,
                                                            long L, int B) {
    constexpr int N = Vec16<T>::N;
    const long nvec = L / N;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        float wv[N], bv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { wv[i] = w[v * N + i]; bv[i] = bias[v * N + i]; }
        for (int b = 0; b < B; ++b) {
            const float mu = mean[b], rs = rstd[b];
            Vec16<T> xv, o;
            MU_LD(32, xv, x + (long)b * L + v * N);
#pragma unroll
            for (int i = 0; i < N; ++i) o.set(i, (xv.get(i) - mu) * rs * wv[i] + bv[i]);
            o.store(y + (long)b * L + v * N);
        }
    }
}

// dx = rstd*(dy*w - m1 - xhat*m2);  dw[f] = sum_b dy*xhat;  db[f] = sum_b dy
template <typename T>
__global__ __launch_bounds__(256) void lns_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ w,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ m1, const float* __restrict__ m2, T* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db, long L, int B) {
    constexpr int N = Vec16<T>::N;
    const long nvec = L / N;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        float wv[N], aw[N], ab[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { wv[i] = w[v * N + i]; aw[i] = 0.f; ab[i] = 0.f; }
        for (int b = 0; b < B; ++b) {
            const float mu = mean[b], rs = rstd[b], a1 = m1[b], a2 = m2[b];
            Vec16<T> xv, gv, o;
            MU_LD(32, xv, x + (long)b * L + v * N);
            MU_LD(32, gv, dy + (long)b * L + v * N);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float xh = (xv.get(i) - mu) * rs, g = gv.get(i);
                aw[i] += g * xh;
                ab[i] += g;
                o.set(i, rs * (g * wv[i] - a1 - xh * a2));
            }
            o.store(dx + (long)b * L + v * N);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) { dw[v * N + i] = aw[i]; db[v * N + i] = ab[i]; }
    }
}

extern "C" long mu_ln_sample_workspace_bytes(int B) { return (long)B * LN_CHUNKS * 2 * sizeof(double) + 2L * B * sizeof(float); }

template <typename T>
static int lns_fwd_t(const T* x, const float* w, const float* b, T* y, float* mean, float* rstd, int B, long L, float eps, void* ws,
                     hipStream_t st) {
    constexpr int N = Vec16<T>::N;
    dim3 grid(LN_CHUNKS, B);
    lns_partial_kernel<T, 0><<<grid, 256, 0, st>>>(x, nullptr, nullptr, nullptr, nullptr, L, (double*)ws);
    lns_final_kernel<<<mu_cdiv(B, 4), 256, 0, st>>>((const double*)ws, LN_CHUNKS, L, eps, 0, mean, rstd, B);
    long nvec = L / N;
    int g = (int)((nvec + 255) / 256 > 8192 ? 8192 : (nvec + 255) / 256);
    lns_fwd_apply_kernel<T><<<g, 256, 0, st>>>(x, w, b, mean, rstd, y, L, B);
    return MU_OK;
}

extern "C" int mu_ln_sample_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int B, long L,
                                float eps, void* workspace, long ws_bytes, int dtype, void* stream) {
    if (!x || !w || !b || !y || !mean || !rstd || !workspace || B <= 0 || L <= 0 || L % 8) return MU_ERR_ARG;
    if (ws_bytes < mu_ln_sample_workspace_bytes(B)) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32) lns_fwd_t<float>((const float*)x, w, b, (float*)y, mean, rstd, B, L, eps, workspace, st);
    else if (dtype == MU_F16) lns_fwd_t<h16>((const h16*)x, w, b, (h16*)y, mean, rstd, B, L, eps, workspace, st);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}

template <typename T>
static int lns_bwd_t(const T* x, const T* dy, const float* w, const float* mean, const float* rstd, T* dx, float* dw, float* db,
                     int B, long L, void* ws, hipStream_t st) {
    constexpr int N = Vec16<T>::N;
    dim3 grid(LN_CHUNKS, B);
    float* m1 = (float*)((char*)ws + (size_t)B * LN_CHUNKS * 2 * sizeof(double));
    float* m2 = m1 + B;
    lns_partial_kernel<T, 1><<<grid, 256, 0, st>>>(x, dy, w, mean, rstd, L, (double*)ws);
    lns_final_kernel<<<mu_cdiv(B, 4), 256, 0, st>>>((const double*)ws, LN_CHUNKS, L, 0.f, 1, m1, m2, B);
    long nvec = L / N;
    int g = (int)((nvec + 255) / 256 > 8192 ? 8192 : (nvec + 255) / 256);
    lns_bwd_apply_kernel<T><<<g, 256, 0, st>>>(x, dy, w, mean, rstd, m1, m2, dx, dw, db, L, B);
    return MU_OK;
}

extern "C" int mu_ln_sample_bwd(const void* x, const void* dy, const float* w, const float* mean, const float* rstd, void* dx,
                                float* dw, float* db, int B, long L, void* workspace, long ws_bytes, int dtype, void* stream) {
    if (!x || !dy || !w || !mean || !rstd || !dx || !dw || !db || !workspace || B <= 0 || L <= 0 || L % 8) return MU_ERR_ARG;
    if (ws_bytes < mu_ln_sample_workspace_bytes(B)) return MU_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MU_F32) lns_bwd_t<float>((const float*)x, (const float*)dy, w, mean, rstd, (float*)dx, dw, db, B, L, workspace, st);
    else if (dtype == MU_F16) lns_bwd_t<h16>((const h16*)x, (const h16*)dy, w, mean, rstd, (h16*)dx, dw, db, B, L, workspace, st);
    else return MU_ERR_ARG;
    MU_CHECK_LAUNCH();
    return MU_OK;
}
