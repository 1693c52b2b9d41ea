// Training building blocks for gfx950: train-mode BatchNorm (+ fused activation) forward / backward, the fused
// Adam step, and the task-weight half of GradNorm.  SURVEY section 8 rows A13, A15, A16 (main.py:623-777).
//
// All of these are HBM-bound streaming kernels over NHWC [pixels][channels] tensors:
//   * a thread owns ONE 8-channel chunk column for its whole life (per-channel constants live in registers)
//     and walks pixel rows; a wave therefore reads whole contiguous rows -> fully coalesced 16/32-byte loads;
//   * statistics are accumulated in f64 per thread, combined through LDS, written as per-workgroup partials and
//     folded by a tiny finalize launch in a fixed order -> bitwise reproducible, no atomics;
//   * algorithmic bytes: BN forward = 2 reads + 1 write of the tensor (statistics pass + apply pass),
//     BN backward = 2 reads (x, dy) for the reduction + 2..3 reads + 1 write for the apply pass.
#include "common.h"
#include "conv_common.h"

namespace {

using ppnconv::load8;
using ppnconv::store8;

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;

__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == PPN_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == PPN_ACT_LRELU) return z > 0.f ? z : 0.1f * z;
    return z;
}
__device__ __forceinline__ float act_slope(float z, int act) {
    if (act == PPN_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == PPN_ACT_LRELU) return z > 0.f ? 1.f : 0.1f;
    return 1.f;
}

struct Slab {
    int cc;          // chunk columns = C/8
    int rpi;         // pixel rows handled per iteration by one workgroup = 256/cc
    long long slab;  // pixel rows per workgroup (multiple of rpi)
    int nblocks;
};

// A workgroup's per-thread sums a[8], b[8] (thread = one 8-channel column x its rows) -> partial[(block * C + c) * 2 + {0, 1}]:
// thread (column, j) folds the rpi row-threads of its column in a fixed order.  Shared by the reduction kernel and by the apply
// kernels that fold the NEXT BatchNorm's sums as they write its input (same slab, same order: the same bits).
__device__ __forceinline__ void fold_block_partials(const double (&a)[8], const double (&b)[8], const Slab& s, int C,
                                                    double* __restrict__ partial, double (*red)[17]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[t][j] = a[j];
        red[t][8 + j] = b[j];
    }
    __syncthreads();
    for (int o = t; o < s.cc * 16; o += kThreads) {
        const int c_ = o / 16, j = o % 16;
        double acc = 0.0;
        for (int r = 0; r < s.rpi; ++r) acc += red[r * s.cc + c_][j];
        const int c = c_ * 8 + (j & 7);
        partial[((size_t)blockIdx.x * C + c) * 2 + (j >> 3)] = acc;
    }
}

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<__bf16>(float v) { return (float)(__bf16)v; }

// ---- per-channel reductions: out[b][c] = {sum f(x), sum h(x)} over the workgroup's pixel slab ------------
// MODE 0: {x, x*x}            (forward statistics)
// MODE 1: {g, g*x_hat}        (backward: dbeta, dgamma), g = dy*act'(x*scale+shift)
template <typename T, int MODE>
__global__ void __launch_bounds__(kThreads) bn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int act, long long P,
                                                             int C, Slab s, double* __restrict__ partial,
                                                             long long sstride = 0, long long pstride = 0) {
    // blockIdx.y: one of several gradient streams over the SAME x (the second-order tail's stacked tangents / adjoints):
    // dy and the partials move by a stream stride, everything per-channel is shared
    dy += (size_t)blockIdx.y * sstride;
    partial += (size_t)blockIdx.y * pstride;
    __shared__ double red[kThreads][17];
    const int t = threadIdx.x;
    const int col = t % s.cc, roff = t / s.cc;
    const long long p0 = (long long)blockIdx.x * s.slab;
    const long long p1 = p0 + s.slab < P ? p0 + s.slab : P;
    double a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = b[j] = 0.0;
    float sc[8], sh[8], mu[8], rs[8];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = col * 8 + j;
            mu[j] = mean[c];
            rs[j] = rstd[c];
            sc[j] = gamma[c] * rs[j];
            sh[j] = beta[c] - mu[j] * sc[j];
        }
    }
    const size_t row_bytes = (size_t)C * sizeof(T);
    const char* xb = reinterpret_cast<const char*>(x) + (size_t)col * 8 * sizeof(T);
    const char* db = reinterpret_cast<const char*>(dy) + (size_t)col * 8 * sizeof(T);
    for (long long p = p0 + roff; p < p1; p += s.rpi) {
        float v[8];
        load8<T>(xb + p * row_bytes, v);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double d = (double)v[j];
                a[j] += d;
                b[j] += d * d;
            }
        } else {
            float g[8];
            load8<T>(db + p * row_bytes, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float z = v[j] * sc[j] + sh[j];
                const float gg = g[j] * act_slope(z, act);
                const float xh = (v[j] - mu[j]) * rs[j];
                a[j] += (double)gg;
                b[j] += (double)gg * (double)xh;
            }
        }
    }
    fold_block_partials(a, b, s, C, partial, red);
}

// forward finalize: mean / rstd / folded affine / running statistics
__global__ void bn_fwd_finalize_kernel(const double* __restrict__ partial, int nblocks, int C, long long P,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                       float momentum, float* __restrict__ running_mean,
                                       float* __restrict__ running_var, float* __restrict__ save_mean,
                                       float* __restrict__ save_rstd, float* __restrict__ scale,
                                       float* __restrict__ shift) {
    // one wave per channel: lanes stride over the workgroup partials, then a fixed-order butterfly
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double S = 0.0, Q = 0.0;
    for (int b = lane; b < nblocks; b += 64) {
        S += partial[((size_t)b * C + c) * 2];
        Q += partial[((size_t)b * C + c) * 2 + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        S += __shfl_xor(S, off, 64);
        Q += __shfl_xor(Q, off, 64);
    }
    if (lane != 0) return;
    const double n = (double)P;
    const double m = S / n;
    double var = Q / n - m * m;
    var = var > 0.0 ? var : 0.0;
    const float mf = (float)m, vf = (float)var;
    const float r = 1.0f / sqrtf(vf + eps);
    save_mean[c] = mf;
    save_rstd[c] = r;
    if (scale) {
        const float sc = gamma[c] * r;
        scale[c] = sc;
        shift[c] = beta[c] - mf * sc;
    }
    if (running_mean) {
        const float unbiased = (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// EMIT: also fold {sum y, sum y^2} of the values it STORES (rounded to T) into `partial` -- the reduction pass of a BatchNorm
// that takes y as its input (a pre-activation block's bn1 behind a conv-BN-ReLU unit, drn.py:47-63), bit for bit
template <typename T, bool EMIT>
__global__ void __launch_bounds__(kThreads) bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int act, long long P,
                                                            int C, Slab s, T* __restrict__ y, double* __restrict__ partial) {
    __shared__ double red[EMIT ? kThreads : 1][17];
    const int t = threadIdx.x;
    const int col = t % s.cc, roff = t / s.cc;
    const long long p0 = (long long)blockIdx.x * s.slab;
    const long long p1 = p0 + s.slab < P ? p0 + s.slab : P;
    float sc[8], sh[8];
    double a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = col * 8 + j;
        sc[j] = gamma[c] * rstd[c];
        sh[j] = beta[c] - mean[c] * sc[j];
        a[j] = b[j] = 0.0;
    }
    const size_t row_bytes = (size_t)C * sizeof(T);
    const char* xb = reinterpret_cast<const char*>(x) + (size_t)col * 8 * sizeof(T);
    char* yb = reinterpret_cast<char*>(y) + (size_t)col * 8 * sizeof(T);
    for (long long p = p0 + roff; p < p1; p += s.rpi) {
        float v[8];
        load8<T>(xb + p * row_bytes, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j] * sc[j] + sh[j], act);
        store8<T>(yb + p * row_bytes, v);
        if constexpr (EMIT) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double d = (double)round_to<T>(v[j]);
                a[j] += d;
                b[j] += d * d;
            }
        }
    }
    if constexpr (EMIT) fold_block_partials(a, b, s, C, partial, red);
}

// backward finalize: dgamma, dbeta and the three per-channel coefficients of  dx = ca*g + cb*x + cc
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblocks, int C, long long P,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ coef, long long pstride = 0) {
    partial += (size_t)blockIdx.y * pstride;           // stream blockIdx.y: its own partials, dgamma / dbeta rows, coefficients
    dgamma += (size_t)blockIdx.y * C;
    dbeta += (size_t)blockIdx.y * C;
    coef += (size_t)blockIdx.y * 3 * C;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double db = 0.0, dg = 0.0;
    for (int b = lane; b < nblocks; b += 64) {
        db += partial[((size_t)b * C + c) * 2];
        dg += partial[((size_t)b * C + c) * 2 + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        db += __shfl_xor(db, off, 64);
        dg += __shfl_xor(dg, off, 64);
    }
    if (lane != 0) return;
    dgamma[c] = (float)dg;
    dbeta[c] = (float)db;
    const double n = (double)P, g = gamma[c], r = rstd[c], m = mean[c];
    coef[c] = (float)(g * r);
    coef[C + c] = (float)(-g * r * r * dg / n);
    coef[2 * C + c] = (float)(g * r * (m * r * dg - db) / n);
}

// NEXT: the dx this kernel writes is the dy of ANOTHER BatchNorm (+ activation) over the tensor nx.x of the same shape -- the
// projection shortcut's BatchNorm of the block below, or the conv-BN-ReLU unit below (trainer.py backward()): fold that
// BatchNorm's {sum g, sum g * xhat} as bn_reduce_kernel<T, 1> would from the stored dx, bit for bit, into nx.partial.
template <typename T>
struct NextBn {
    const T* x;
    const float *gamma, *beta, *mean, *rstd;
    int act;
    double* partial;
};

template <typename T, bool NEXT>
__global__ void __launch_bounds__(kThreads) bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                const T* __restrict__ add,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ coef, int act,
                                                                long long P, int C, Slab s, T* __restrict__ dx,
                                                                long long sstride, NextBn<T> nx) {
    __shared__ double red[NEXT ? kThreads : 1][17];
    dy += (size_t)blockIdx.y * sstride;                // stream blockIdx.y (see bn_reduce_kernel)
    dx += (size_t)blockIdx.y * sstride;
    if (add) add += (size_t)blockIdx.y * sstride;
    coef += (size_t)blockIdx.y * 3 * C;
    const int t = threadIdx.x;
    const int col = t % s.cc, roff = t / s.cc;
    const long long p0 = (long long)blockIdx.x * s.slab;
    const long long p1 = p0 + s.slab < P ? p0 + s.slab : P;
    float sc[8], sh[8], ca[8], cb[8], cc[8];
    float sc2[8], sh2[8], mu2[8], rs2[8];
    double a2[8], b2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = col * 8 + j;
        sc[j] = gamma[c] * rstd[c];
        sh[j] = beta[c] - mean[c] * sc[j];
        ca[j] = coef[c];
        cb[j] = coef[C + c];
        cc[j] = coef[2 * C + c];
        if constexpr (NEXT) {
            mu2[j] = nx.mean[c];
            rs2[j] = nx.rstd[c];
            sc2[j] = nx.gamma[c] * rs2[j];
            sh2[j] = nx.beta[c] - mu2[j] * sc2[j];
            a2[j] = b2[j] = 0.0;
        }
    }
    const size_t row_bytes = (size_t)C * sizeof(T);
    const size_t cofs = (size_t)col * 8 * sizeof(T);
    const char* xb = reinterpret_cast<const char*>(x) + cofs;
    const char* db = reinterpret_cast<const char*>(dy) + cofs;
    const char* ab = add ? reinterpret_cast<const char*>(add) + cofs : nullptr;
    const char* nb = NEXT ? reinterpret_cast<const char*>(nx.x) + cofs : nullptr;
    char* ob = reinterpret_cast<char*>(dx) + cofs;
    for (long long p = p0 + roff; p < p1; p += s.rpi) {
        float v[8], g[8], o[8];
        load8<T>(xb + p * row_bytes, v);
        load8<T>(db + p * row_bytes, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = v[j] * sc[j] + sh[j];
            o[j] = ca[j] * (g[j] * act_slope(z, act)) + cb[j] * v[j] + cc[j];
        }
        if (ab) {
            float r[8];
            load8<T>(ab + p * row_bytes, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += r[j];
        }
        store8<T>(ob + p * row_bytes, o);
        if constexpr (NEXT) {
            float w[8];
            load8<T>(nb + p * row_bytes, w);
#pragma unroll
            for (int j = 0; j < 8; ++j) {                            // bn_reduce_kernel MODE 1 on (nx.x, the stored dx)
                const float z = w[j] * sc2[j] + sh2[j];
                const float gg = round_to<T>(o[j]) * act_slope(z, nx.act);
                const float xh = (w[j] - mu2[j]) * rs2[j];
                a2[j] += (double)gg;
                b2[j] += (double)gg * (double)xh;
            }
        }
    }
    if constexpr (NEXT) fold_block_partials(a2, b2, s, C, nx.partial, red);
}

// ---- Adam ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n,
                                                        float omb1, float b2, float omb2, float eps, float wd,
                                                        float step_size, float inv_bc2_sqrt, float gscale,
                                                        __bf16* __restrict__ lp) {
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = m[i] + (gi - m[i]) * omb1;                  // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + omb2 * gi * gi;                 // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        const float po = pi - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
        p[i] = po;
        if (lp) lp[i] = (__bf16)po;
    }
}

// ---- sum of squares -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) sumsq_partial_kernel(const float* __restrict__ x, long long n,
                                                                 double* __restrict__ partial) {
    __shared__ double red[kThreads];
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const double d = (double)x[i];
        acc += d * d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void sumsq_final_kernel(const double* __restrict__ partial, int nb, float* __restrict__ out) {
    double acc = 0.0;
    for (int b = 0; b < nb; ++b) acc += partial[b];
    out[0] = (float)acc;
}

// ---- GradNorm probe statistics (main.py:704-717 for the limb loss by linearity; trainer.py _second_order_tail) ----
// gw4 = (total - sum_{i<4} c_i g_i) / c_4 and seven sums of squares in one pass: ||g_0..3||^2, ||gw4||^2, the unscaled
// remainder's and the total's (the trust test of the bf16 mode).  Same element -> thread mapping and the same
// reduction tree as sumsq_partial_kernel, so every sum equals what ppn_sumsq returns for that tensor.
struct ProbeStatsArgs {
    const float* g[4];
    const float* total;
    float c[5];
    long long n;
    float* gw4;
    double* partial;      // [gridDim.x][7]
};
__global__ void __launch_bounds__(kThreads) probe_stats_partial_kernel(ProbeStatsArgs a) {
    __shared__ double red[kThreads];
    double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < a.n; i += stride) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = a.g[k][i];
            acc[k] += (double)g * (double)g;
            s = s + a.c[k] * g;
        }
        const float t = a.total[i];
        const float rest = t - s;
        const float g4 = rest / a.c[4];
        a.gw4[i] = g4;
        acc[4] += (double)g4 * (double)g4;
        acc[5] += (double)rest * (double)rest;
        acc[6] += (double)t * (double)t;
    }
    for (int k = 0; k < 7; ++k) {
        __syncthreads();
        red[threadIdx.x] = acc[k];
        __syncthreads();
        for (int o = kThreads / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.partial[(size_t)blockIdx.x * 7 + k] = red[0];
    }
}
__global__ void probe_stats_final_kernel(const double* __restrict__ partial, int nb, float* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= 7) return;
    double acc = 0.0;
    for (int b = 0; b < nb; ++b) acc += partial[(size_t)b * 7 + k];
    out[k] = (float)acc;
}

// ---- GradNorm task weights (5 scalars, one lane) ------------------------------------------------------------
__global__ void gradnorm_kernel(float* __restrict__ w, const float* __restrict__ L, const float* __restrict__ gn,
                                const float* __restrict__ base, float alpha, float* __restrict__ m,
                                float* __restrict__ v, float omb1, float b2, float omb2, float eps,
                                float step_size, float inv_bc2_sqrt, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float l[5], G[5], lhat[5], Cc[5], dw[5];
    float gsum = 0.f, lsum = 0.f;
    for (int i = 0; i < 5; ++i) {
        l[i] = w[i] * L[i];                          // main.py:668-672
        G[i] = fabsf(w[i]) * gn[i];                  // ||d(w_i L_i)/dW||_2, main.py:704-721
    }
    gsum = (((G[0] + G[1]) + G[2]) + G[3]) + G[4];
    const float gavg = gsum / 5.f;                   // main.py:723
    for (int i = 0; i < 5; ++i) lhat[i] = l[i] / base[i];
    lsum = (((lhat[0] + lhat[1]) + lhat[2]) + lhat[3]) + lhat[4];
    const float lavg = lsum / 5.f;                   // main.py:732
    float lgrad = 0.f;
    for (int i = 0; i < 5; ++i) {
        Cc[i] = gavg * powf(lhat[i] / lavg, alpha);  // main.py:735-746, detached
        const float d = G[i] - Cc[i];
        lgrad += fabsf(d);                           // nn.L1Loss on scalars
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float sw = w[i] > 0.f ? 1.f : (w[i] < 0.f ? -1.f : 0.f);
        dw[i] = sgn * sw * gn[i];
    }
    for (int i = 0; i < 5; ++i) {                    // optimizerR.step(): Adam on the 5 weights
        const float mi = m[i] + (dw[i] - m[i]) * omb1;
        const float vi = v[i] * b2 + omb2 * dw[i] * dw[i];
        m[i] = mi;
        v[i] = vi;
        w[i] = w[i] - step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
    }
    if (out) {
        for (int i = 0; i < 5; ++i) {
            out[i] = G[i];
            out[5 + i] = Cc[i];
            out[10 + i] = dw[i];
            out[15 + i] = 0.f;
        }
        out[15] = lgrad;
        out[16] = gavg;
    }
}

__global__ void gradnorm_renorm_kernel(float* __restrict__ w, float world) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float t[5], s = 0.f;
    for (int i = 0; i < 5; ++i) {
        t[i] = w[i] / world;
        t[i] = t[i] < 0.f ? 0.f : t[i];              // clamp_(min=0.0)
    }
    s = (((t[0] + t[1]) + t[2]) + t[3]) + t[4];
    const float mean = s / 5.f;
    for (int i = 0; i < 5; ++i) w[i] = t[i] / mean;
}


// ---- column sums (bias gradient of a convolution that feeds NHWC) ----------------------------------------------
__global__ void colsum_finalize_kernel(const double* __restrict__ partial, int nblocks, int C, float* __restrict__ out) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double S = 0.0;
    for (int b = lane; b < nblocks; b += 64) S += partial[((size_t)b * C + c) * 2];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) S += __shfl_xor(S, off, 64);
    if (lane == 0) out[c] = (float)S;
}

// ---- head: gradient through the sigmoid + NCHW f32 -> NHWC relayout ---------------------------------------------
// dz[b][hw][c] = g[b][c][hw] * s*(1-s),  s = head[b][c][hw]   (channels c >= C of the padded output are zero)
template <typename T, bool SIG>
__global__ void __launch_bounds__(256) head_grad_kernel(const float* __restrict__ head, const float* __restrict__ grad,
                                                        int Ctot, int C, int HW, int Cpad, T* __restrict__ dz) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, p0 = blockIdx.y * 64, b = blockIdx.z;
    const int t = threadIdx.x;
    const int px = t & 63;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int cl = (t >> 6) + 4 * i;
        const int c = c0 + cl, p = p0 + px;
        float v = 0.f;
        if (c < C && p < HW) {
            const size_t o = ((size_t)b * Ctot + c) * HW + p;
            if constexpr (SIG) {
                const float sg = head[o];
                v = grad[o] * (sg * (1.f - sg));
            } else {
                v = grad[o];                               // plain relayout (the gradient is already w.r.t. the logits)
            }
        }
        tile[px][cl] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = t + 256 * i;
        const int pr = item >> 3, ch = item & 7;
        const int p = p0 + pr;
        if (p < HW && c0 + ch * 8 < Cpad) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[pr][ch * 8 + j];
            store8<T>(reinterpret_cast<char*>(dz) + (((size_t)b * HW + p) * Cpad + c0 + ch * 8) * sizeof(T), v);
        }
    }
}

// dbias[c] = sum_{b,hw} g*s*(1-s): one wave per channel, fixed order
__global__ void __launch_bounds__(256) head_bias_grad_kernel(const float* __restrict__ head,
                                                             const float* __restrict__ grad, int B, int C, int HW,
                                                             float* __restrict__ dbias) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double acc = 0.0;
    for (int b = 0; b < B; ++b) {
        const size_t o = ((size_t)b * C + c) * HW;
        for (int p = lane; p < HW; p += 64) {
            const float sg = head[o + p];
            acc += (double)(grad[o + p] * (sg * (1.f - sg)));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) dbias[c] = (float)acc;
}


// ---- second-order pieces (GradNorm's Lgrad.backward(), main.py:759) ------------------------------------------------
// out = in * act'(x*scale + shift): the activation mask of a forward-mode tangent that went through BN first
template <typename T>
__global__ void __launch_bounds__(kThreads) bn_act_mask_kernel(const T* __restrict__ x, const T* __restrict__ in,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, int act, long long P,
                                                               int C, Slab s, T* __restrict__ out, long long sstride = 0) {
    in += (size_t)blockIdx.y * sstride;                // stream blockIdx.y (see bn_reduce_kernel)
    out += (size_t)blockIdx.y * sstride;
    const int t = threadIdx.x;
    const int col = t % s.cc, roff = t / s.cc;
    const long long p0 = (long long)blockIdx.x * s.slab;
    const long long p1 = p0 + s.slab < P ? p0 + s.slab : P;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = col * 8 + j;
        sc[j] = gamma[c] * rstd[c];
        sh[j] = beta[c] - mean[c] * sc[j];
    }
    const size_t row_bytes = (size_t)C * sizeof(T), cofs = (size_t)col * 8 * sizeof(T);
    for (long long p = p0 + roff; p < p1; p += s.rpi) {
        float v[8], g[8];
        load8<T>(reinterpret_cast<const char*>(x) + cofs + p * row_bytes, v);
        load8<T>(reinterpret_cast<const char*>(in) + cofs + p * row_bytes, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] *= act_slope(v[j] * sc[j] + sh[j], act);
        store8<T>(reinterpret_cast<char*>(out) + cofs + p * row_bytes, g);
    }
}

// Adjoint of the train-mode BN forward-mode tangent  ydot = (gamma*rstd) * P(xdot),  P(v) = v - mean(v) - xhat*mean(xhat*v),
// with respect to x (through xhat and rstd) and gamma.  q = dyt * act'(z) is the adjoint arriving at ydot.
// Five per-channel sums: Sq, Sqx = sum q*xhat, Sx = sum xdot, Sxx = sum xdot*xhat, Sqd = sum q*xdot.
template <typename T>
__global__ void __launch_bounds__(kThreads) bn_dual_reduce_kernel(const T* __restrict__ x, const T* __restrict__ xdot,
                                                                  const T* __restrict__ dyt,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, int act,
                                                                  long long P, int C, Slab s,
                                                                  double* __restrict__ partial, long long sstride = 0,
                                                                  long long pstride = 0) {
    xdot += (size_t)blockIdx.y * sstride;              // stream blockIdx.y (see bn_reduce_kernel)
    dyt += (size_t)blockIdx.y * sstride;
    partial += (size_t)blockIdx.y * pstride;
    __shared__ double red[kThreads][9];
    const int t = threadIdx.x;
    const int col = t % s.cc, roff = t / s.cc;
    const long long p0 = (long long)blockIdx.x * s.slab;
    const long long p1 = p0 + s.slab < P ? p0 + s.slab : P;
    const size_t row_bytes = (size_t)C * sizeof(T), cofs = (size_t)col * 8 * sizeof(T);
    float mu[8], rs[8], sc[8], sh[8];
    double acc[5][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = col * 8 + j;
        mu[j] = mean[c]; rs[j] = rstd[c];
        sc[j] = gamma[c] * rs[j]; sh[j] = beta[c] - mu[j] * sc[j];
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k][j] = 0.0;
    }
    for (long long p = p0 + roff; p < p1; p += s.rpi) {
        float xv[8], xd[8], dv[8];
        load8<T>(reinterpret_cast<const char*>(x) + cofs + p * row_bytes, xv);
        load8<T>(reinterpret_cast<const char*>(xdot) + cofs + p * row_bytes, xd);
        load8<T>(reinterpret_cast<const char*>(dyt) + cofs + p * row_bytes, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float q = dv[j] * act_slope(xv[j] * sc[j] + sh[j], act);
            const float xh = (xv[j] - mu[j]) * rs[j];
            acc[0][j] += q; acc[1][j] += (double)q * xh; acc[2][j] += xd[j]; acc[3][j] += (double)xd[j] * xh;
            acc[4][j] += (double)q * xd[j];
        }
    }
    // fold the row-threads of every chunk column, one of the five sums at a time (18 KB of LDS)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[t][j] = acc[k][j];
        __syncthreads();
        for (int o = t; o < s.cc * 8; o += kThreads) {
            const int c_ = o / 8, j = o % 8;
            double a_ = 0.0;
            for (int r = 0; r < s.rpi; ++r) a_ += red[r * s.cc + c_][j];
            partial[((size_t)blockIdx.x * C + c_ * 8 + j) * 5 + k] = a_;
        }
        __syncthreads();
    }
}

// coef[c][0..3]: dx_tan = c0*xhat + c1*q + c2*xdot + c3   (xhat = (x-mean)*rstd), dgamma_tan[c] = A*rstd
__global__ void bn_dual_finalize_kernel(const double* __restrict__ partial, int nblocks, int C, long long P,
                                        const float* __restrict__ gamma, const float* __restrict__ rstd,
                                        float* __restrict__ coef, float* __restrict__ dgamma_tan, long long pstride = 0) {
    partial += (size_t)blockIdx.y * pstride;
    coef += (size_t)blockIdx.y * 4 * C;
    dgamma_tan += (size_t)blockIdx.y * C;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double S[5] = {0, 0, 0, 0, 0};
    for (int b = lane; b < nblocks; b += 64)
        for (int k = 0; k < 5; ++k) S[k] += partial[((size_t)b * C + c) * 5 + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 5; ++k) S[k] += __shfl_xor(S[k], off, 64);
    if (lane != 0) return;
    const double n = (double)P, Sq = S[0], Sqx = S[1], Sx = S[2], Sxx = S[3], Sqd = S[4];
    const double m1 = Sx / n, m2 = Sxx / n, qb = Sq / n;
    const double A = Sqd - Sq * Sx / n - Sqx * Sxx / n;
    const double g = gamma[c], r = rstd[c];
    const double k0 = -g * r * r;
    // dx_tan = k0 * [ A*xhat/n + m2*(q - qb - xhat*Sqx/n) + (Sqx/n)*(xdot - m1 - xhat*m2) ]
    coef[c * 4 + 0] = (float)(k0 * (A / n - 2.0 * m2 * Sqx / n));
    coef[c * 4 + 1] = (float)(k0 * m2);
    coef[c * 4 + 2] = (float)(k0 * Sqx / n);
    coef[c * 4 + 3] = (float)(k0 * (-m2 * qb - Sqx / n * m1));
    dgamma_tan[c] = (float)(A * r);
}

template <typename T>
__global__ void __launch_bounds__(kThreads) bn_dual_apply_kernel(const T* __restrict__ x, const T* __restrict__ xdot,
                                                                 const T* __restrict__ dyt,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd,
                                                                 const float* __restrict__ coef, int act, long long P,
                                                                 int C, Slab s, T* __restrict__ dx, long long sstride = 0) {
    xdot += (size_t)blockIdx.y * sstride;              // stream blockIdx.y (see bn_reduce_kernel)
    dyt += (size_t)blockIdx.y * sstride;
    dx += (size_t)blockIdx.y * sstride;
    coef += (size_t)blockIdx.y * 4 * C;
    const int t = threadIdx.x;
    const int col = t % s.cc, roff = t / s.cc;
    const long long p0 = (long long)blockIdx.x * s.slab;
    const long long p1 = p0 + s.slab < P ? p0 + s.slab : P;
    float sc[8], sh[8], mu[8], rs[8], k0[8], k1[8], k2[8], k3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = col * 8 + j;
        mu[j] = mean[c]; rs[j] = rstd[c];
        sc[j] = gamma[c] * rs[j]; sh[j] = beta[c] - mu[j] * sc[j];
        k0[j] = coef[c * 4]; k1[j] = coef[c * 4 + 1]; k2[j] = coef[c * 4 + 2]; k3[j] = coef[c * 4 + 3];
    }
    const size_t row_bytes = (size_t)C * sizeof(T), cofs = (size_t)col * 8 * sizeof(T);
    for (long long p = p0 + roff; p < p1; p += s.rpi) {
        float v[8], xd[8], g[8], o[8];
        load8<T>(reinterpret_cast<const char*>(x) + cofs + p * row_bytes, v);
        load8<T>(reinterpret_cast<const char*>(xdot) + cofs + p * row_bytes, xd);
        load8<T>(reinterpret_cast<const char*>(dyt) + cofs + p * row_bytes, g);
        load8<T>(reinterpret_cast<const char*>(dx) + cofs + p * row_bytes, o);          // accumulate into dx
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float q = g[j] * act_slope(v[j] * sc[j] + sh[j], act);
            const float xh = (v[j] - mu[j]) * rs[j];
            o[j] += k0[j] * xh + k1[j] * q + k2[j] * xd[j] + k3[j];
        }
        store8<T>(reinterpret_cast<char*>(dx) + cofs + p * row_bytes, o);
    }
}


// ---- Bottleneck tail: out = relu(z + r) (drn.py:92-95) and its backward mask -------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads) add_relu_kernel(const T* __restrict__ z, const T* __restrict__ r, long long n8,
                                                            T* __restrict__ out) {
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n8; i += stride) {
        float a[8], b[8];
        load8<T>(reinterpret_cast<const char*>(z) + i * 8 * sizeof(T), a);
        load8<T>(reinterpret_cast<const char*>(r) + i * 8 * sizeof(T), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float v = a[j] + b[j]; a[j] = v > 0.f ? v : 0.f; }
        store8<T>(reinterpret_cast<char*>(out) + i * 8 * sizeof(T), a);
    }
}
// dz = dout * (out > 0)   [+ add]
template <typename T>
__global__ void __launch_bounds__(kThreads) relu_mask_kernel(const T* __restrict__ out, const T* __restrict__ dout,
                                                             const T* __restrict__ add, long long n8,
                                                             T* __restrict__ dz) {
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n8; i += stride) {
        float o[8], g[8];
        load8<T>(reinterpret_cast<const char*>(out) + i * 8 * sizeof(T), o);
        load8<T>(reinterpret_cast<const char*>(dout) + i * 8 * sizeof(T), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
        if (add) {
            float r[8];
            load8<T>(reinterpret_cast<const char*>(add) + i * 8 * sizeof(T), r);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += r[j];
        }
        store8<T>(reinterpret_cast<char*>(dz) + i * 8 * sizeof(T), g);
    }
}

// ---- stride-2 input gradients without torch's strided copies ----------------------------------------------------------
// dst[b][y][x][:] = src[b][y/s][x/s][:] where y, x are multiples of s inside src, 0 elsewhere: the zero-upsampled dy of the
// transposed-convolution identity (train.conv_dgrad) and the even-pixel scatter of a 1x1 stride-2 projection's input
// gradient, written in ONE pass (torch: a fill + a strided copy_, 16 + 26 us for a 150 MB tensor).  16-byte items.
__global__ void __launch_bounds__(kThreads) upsample_zero_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int B,
                                                                 int Hs, int Ws, int Hd, int Wd, int c16, int s) {
    const long long n = (long long)B * Hd * Wd * c16;
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % c16);
        long long p = i / c16;
        const int x = (int)(p % Wd); p /= Wd;
        const int y = (int)(p % Hd);
        const int b = (int)(p / Hd);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (y % s == 0 && x % s == 0 && y / s < Hs && x / s < Ws)
            v = src[(((size_t)b * Hs + y / s) * Ws + x / s) * c16 + c];
        dst[i] = v;
    }
}
// dx[b][y][x][:] = o_{y&1, x&1}[b][(y + (y&1)) >> 1][(x + (x&1)) >> 1][:]: the four parity sub-convolutions of a stride-2
// 3x3 input gradient (train._dgrad_stride2) interleaved in one pass instead of four strided copies.
__global__ void __launch_bounds__(kThreads) interleave_parity_kernel(const uint4* __restrict__ o00, const uint4* __restrict__ o01,
                                                                     const uint4* __restrict__ o10, const uint4* __restrict__ o11,
                                                                     uint4* __restrict__ dx, int B, int H, int W, int Ho1,
                                                                     int Wo1, int c16) {
    const long long n = (long long)B * H * W * c16;
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % c16);
        long long p = i / c16;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        const int py = y & 1, px = x & 1;
        const uint4* o = py ? (px ? o11 : o10) : (px ? o01 : o00);
        dx[i] = o[(((size_t)b * Ho1 + ((y + py) >> 1)) * Wo1 + ((x + px) >> 1)) * c16 + c];
    }
}
// ... with the four sub-convolutions' outputs stacked along the channels of ONE tensor o [B][Ho1][Wo1][4 * c16] (parity
// (py, px) in channel block 2 py + px): they then come from a single launch that reads dy once.
__global__ void __launch_bounds__(kThreads) interleave_parity_stacked_kernel(const uint4* __restrict__ o, uint4* __restrict__ dx,
                                                                             int B, int H, int W, int Ho1, int Wo1, int c16) {
    const long long n = (long long)B * H * W * c16;
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % c16);
        long long p = i / c16;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        const int py = y & 1, px = x & 1;
        dx[i] = o[((((size_t)b * Ho1 + ((y + py) >> 1)) * Wo1 + ((x + px) >> 1)) * 4 + (2 * py + px)) * c16 + c];
    }
}
// f32 NCHW image [B][3][H][W] -> NHWC [B][H][W][cpad] (cpad 4 or 8, channels 3.. zero): the copy of the input the 7x7
// layer's weight gradient reads (torch: zeros + a strided permute-copy).
template <typename T, int CP>
__global__ void __launch_bounds__(kThreads) image_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B,
                                                                 long long HW) {
    const long long n = (long long)B * HW;
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const long long b = i / HW, p = i % HW;
        const float* s0 = src + (size_t)b * 3 * HW + p;
        T v[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) v[c] = (T)0.f;
        v[0] = (T)s0[0]; v[1] = (T)s0[HW]; v[2] = (T)s0[2 * HW];
        T* d = dst + (size_t)i * CP;
#pragma unroll
        for (int c = 0; c < CP; ++c) d[c] = v[c];
    }
}

int make_slab(int C, long long P, Slab* s) {
    if (C < 8 || C > 2048 || (C & (C - 1)) != 0)
        return ppn::fail(PPN_E_UNSUPPORTED, "BatchNorm channels must be a power of two in [8, 2048], got %d", C);
    if (P <= 0) return ppn::fail(PPN_E_INVALID, "BatchNorm needs pixels > 0");
    s->cc = C / 8;
    s->rpi = kThreads / s->cc;
    const long long rows_iter = s->rpi;
    long long nb = (P + rows_iter * 8 - 1) / (rows_iter * 8);        // >= 8 rows per thread when possible
    if (nb > kMaxBlocks) nb = kMaxBlocks;
    if (nb < 1) nb = 1;
    long long slab = (P + nb - 1) / nb;
    slab = (slab + rows_iter - 1) / rows_iter * rows_iter;
    s->slab = slab;
    s->nblocks = (int)((P + slab - 1) / slab);
    return PPN_OK;
}

void adam_consts(double lr, double b1, double b2, int step, float* step_size, float* inv_bc2_sqrt) {
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    *step_size = (float)(lr / bc1);
    *inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
}

}  // namespace

extern "C" {

size_t ppn_bn_workspace_bytes(int32_t channels) {
    if (channels <= 0) return 0;
    return (size_t)kMaxBlocks * channels * 2 * sizeof(double) + (size_t)3 * channels * sizeof(float);
}

int ppn_bn_train_fwd(const ppn_bn_desc* d, void* stream) {
    if (!d || !d->x || !d->gamma || !d->beta || !d->save_mean || !d->save_rstd || !d->workspace)
        return ppn::fail(PPN_E_INVALID, "ppn_bn_train_fwd: x, gamma, beta, save_mean, save_rstd, workspace required");
    if ((d->scale == nullptr) != (d->shift == nullptr) || (d->running_mean == nullptr) != (d->running_var == nullptr))
        return ppn::fail(PPN_E_INVALID, "ppn_bn_train_fwd: scale/shift and running_mean/var come in pairs");
    if (d->dtype != PPN_F32 && d->dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", d->dtype);
    if (d->act < PPN_ACT_NONE || d->act > PPN_ACT_LRELU) return ppn::fail(PPN_E_UNSUPPORTED, "bad act %d", d->act);
    Slab s;
    if (int rc = make_slab(d->channels, d->pixels, &s)) return rc;
    hipStream_t st = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(d->workspace);
    const int C = d->channels;
    if (d->stats_blocks < 0 || d->stats_blocks > kMaxBlocks)
        return ppn::fail(PPN_E_INVALID, "ppn_bn_train_fwd: stats_blocks must be in [0, %d]", kMaxBlocks);
    const int nfold = d->stats_blocks > 0 ? d->stats_blocks : s.nblocks;     // partials from the producing convolution's epilogue
    if (d->stats_blocks > 0) {
    } else if (d->dtype == PPN_F32)
        bn_reduce_kernel<float, 0><<<s.nblocks, kThreads, 0, st>>>((const float*)d->x, nullptr, nullptr, nullptr,
                                                                   nullptr, nullptr, 0, d->pixels, C, s, partial);
    else
        bn_reduce_kernel<__bf16, 0><<<s.nblocks, kThreads, 0, st>>>((const __bf16*)d->x, nullptr, nullptr, nullptr,
                                                                    nullptr, nullptr, 0, d->pixels, C, s, partial);
    PPN_LAUNCH_CHECK();
    bn_fwd_finalize_kernel<<<(C + 3) / 4, 256, 0, st>>>(partial, nfold, C, d->pixels, d->gamma, d->beta, d->eps,
                                                         d->momentum, d->running_mean, d->running_var, d->save_mean,
                                                         d->save_rstd, d->scale, d->shift);
    PPN_LAUNCH_CHECK();
    if (d->emit_blocks) *d->emit_blocks = 0;
    if (d->y) {
        // emit_blocks: the apply pass also folds {sum y, sum y^2} of what it stores into the workspace (the finalize launch has
        // consumed this BatchNorm's own partials by then) -- the reduction pass of a BatchNorm whose input is y
        const bool emit = d->emit_blocks != nullptr;
        if (d->dtype == PPN_F32) {
            auto k = emit ? bn_apply_kernel<float, true> : bn_apply_kernel<float, false>;
            k<<<s.nblocks, kThreads, 0, st>>>((const float*)d->x, d->gamma, d->beta, d->save_mean, d->save_rstd, d->act, d->pixels, C,
                                              s, (float*)d->y, partial);
        } else {
            auto k = emit ? bn_apply_kernel<__bf16, true> : bn_apply_kernel<__bf16, false>;
            k<<<s.nblocks, kThreads, 0, st>>>((const __bf16*)d->x, d->gamma, d->beta, d->save_mean, d->save_rstd, d->act, d->pixels,
                                              C, s, (__bf16*)d->y, partial);
        }
        PPN_LAUNCH_CHECK();
        if (emit) *d->emit_blocks = s.nblocks;
    }
    return PPN_OK;
}

static int bn_train_bwd_impl(const ppn_bn_bwd_desc* d, int nstreams, void* stream) {
    if (!d || !d->x || !d->dy || !d->gamma || !d->beta || !d->save_mean || !d->save_rstd || !d->dgamma ||
        !d->dbeta || !d->dx || !d->workspace)
        return ppn::fail(PPN_E_INVALID, "ppn_bn_train_bwd: NULL argument");
    if (d->dtype != PPN_F32 && d->dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", d->dtype);
    if (d->act < PPN_ACT_NONE || d->act > PPN_ACT_LRELU) return ppn::fail(PPN_E_UNSUPPORTED, "bad act %d", d->act);
    if (nstreams < 1 || nstreams > 64) return ppn::fail(PPN_E_INVALID, "ppn_bn_train_bwd_streams: 1 <= nstreams <= 64");
    Slab s;
    if (int rc = make_slab(d->channels, d->pixels, &s)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int C = d->channels;
    const long long pstride = (long long)kMaxBlocks * C * 2, sstride = d->pixels * C;
    double* partial = reinterpret_cast<double*>(d->workspace);
    float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(d->workspace) +
                                           (size_t)nstreams * pstride * sizeof(double));
    const dim3 grid(s.nblocks, nstreams), fgrid((C + 3) / 4, nstreams);
    if (d->stats_blocks < 0 || d->stats_blocks > kMaxBlocks || (d->stats_blocks > 0 && nstreams != 1))
        return ppn::fail(PPN_E_INVALID, "ppn_bn_train_bwd: stats_blocks must be in [0, %d] and needs a single stream", kMaxBlocks);
    const int nfold = d->stats_blocks > 0 ? d->stats_blocks : s.nblocks;     // partials from the input-gradient convolution's epilogue
    if (d->stats_blocks > 0) {
    } else if (d->dtype == PPN_F32)
        bn_reduce_kernel<float, 1><<<grid, kThreads, 0, st>>>((const float*)d->x, (const float*)d->dy, d->gamma,
                                                              d->beta, d->save_mean, d->save_rstd, d->act,
                                                              d->pixels, C, s, partial, sstride, pstride);
    else
        bn_reduce_kernel<__bf16, 1><<<grid, kThreads, 0, st>>>((const __bf16*)d->x, (const __bf16*)d->dy,
                                                               d->gamma, d->beta, d->save_mean, d->save_rstd,
                                                               d->act, d->pixels, C, s, partial, sstride, pstride);
    PPN_LAUNCH_CHECK();
    bn_bwd_finalize_kernel<<<fgrid, 256, 0, st>>>(partial, nfold, C, d->pixels, d->gamma, d->save_mean,
                                                   d->save_rstd, d->dgamma, d->dbeta, coef, pstride);
    PPN_LAUNCH_CHECK();
    // next_x: the dx of this call is the dy of the BatchNorm over next_x (same shape): its sums are folded by the apply pass
    // into the workspace (this BatchNorm's own partials are consumed: the apply pass reads `coef` only)
    const bool next = d->next_x != nullptr;
    if (d->next_blocks) *d->next_blocks = 0;
    if (next && (nstreams != 1 || !d->next_gamma || !d->next_beta || !d->next_mean || !d->next_rstd || !d->next_blocks ||
                 d->next_act < PPN_ACT_NONE || d->next_act > PPN_ACT_LRELU))
        return ppn::fail(PPN_E_INVALID, "ppn_bn_train_bwd: next_x needs next_gamma / beta / mean / rstd / blocks, an activation "
                                        "none / relu / lrelu and a single stream");
    if (d->dtype == PPN_F32) {
        const NextBn<float> nx{(const float*)d->next_x, d->next_gamma, d->next_beta, d->next_mean, d->next_rstd, d->next_act, partial};
        auto k = next ? bn_bwd_apply_kernel<float, true> : bn_bwd_apply_kernel<float, false>;
        k<<<grid, kThreads, 0, st>>>((const float*)d->x, (const float*)d->dy, (const float*)d->dx_add, d->gamma, d->beta,
                                     d->save_mean, d->save_rstd, coef, d->act, d->pixels, C, s, (float*)d->dx, sstride, nx);
    } else {
        const NextBn<__bf16> nx{(const __bf16*)d->next_x, d->next_gamma, d->next_beta, d->next_mean, d->next_rstd, d->next_act,
                                partial};
        auto k = next ? bn_bwd_apply_kernel<__bf16, true> : bn_bwd_apply_kernel<__bf16, false>;
        k<<<grid, kThreads, 0, st>>>((const __bf16*)d->x, (const __bf16*)d->dy, (const __bf16*)d->dx_add, d->gamma, d->beta,
                                     d->save_mean, d->save_rstd, coef, d->act, d->pixels, C, s, (__bf16*)d->dx, sstride, nx);
    }
    PPN_LAUNCH_CHECK();
    if (next) *d->next_blocks = s.nblocks;
    return PPN_OK;
}

int ppn_bn_train_bwd(const ppn_bn_bwd_desc* d, void* stream) { return bn_train_bwd_impl(d, 1, stream); }

int ppn_bn_train_bwd_streams(const ppn_bn_bwd_desc* d, int32_t nstreams, void* stream) {
    return bn_train_bwd_impl(d, nstreams, stream);
}

int ppn_colsum(int32_t dtype, const void* x, int64_t pixels, int32_t channels, float* out, void* workspace,
               void* stream) {
    if (!x || !out || !workspace) return ppn::fail(PPN_E_INVALID, "ppn_colsum: NULL argument");
    if (dtype != PPN_F32 && dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    Slab s;
    if (int rc = make_slab(channels, pixels, &s)) return rc;
    hipStream_t st = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(workspace);
    if (dtype == PPN_F32)
        bn_reduce_kernel<float, 0><<<s.nblocks, kThreads, 0, st>>>((const float*)x, nullptr, nullptr, nullptr, nullptr,
                                                                   nullptr, 0, pixels, channels, s, partial);
    else
        bn_reduce_kernel<__bf16, 0><<<s.nblocks, kThreads, 0, st>>>((const __bf16*)x, nullptr, nullptr, nullptr,
                                                                    nullptr, nullptr, 0, pixels, channels, s, partial);
    PPN_LAUNCH_CHECK();
    colsum_finalize_kernel<<<(channels + 3) / 4, 256, 0, st>>>(partial, s.nblocks, channels, out);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_head_grad(int32_t dtype, const float* head, const float* grad_head, int32_t batch, int32_t channels,
                  int32_t hw, int32_t channels_used, int32_t channels_pad, void* dz, float* dbias, void* stream) {
    if (!head || !grad_head || !dz) return ppn::fail(PPN_E_INVALID, "ppn_head_grad: NULL argument");
    if (dtype != PPN_F32 && dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    if (batch < 1 || channels < 1 || hw < 1 || channels_used < 1 || channels_used > channels ||
        channels_pad < channels_used || channels_pad % 64)
        return ppn::fail(PPN_E_INVALID, "ppn_head_grad: need channels_used <= channels and channels_pad a multiple "
                                        "of 64 >= channels_used");
    if (dbias && channels_used != channels)
        return ppn::fail(PPN_E_INVALID, "ppn_head_grad: dbias needs all channels");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(channels_pad / 64, (hw + 63) / 64, batch);
    if (dtype == PPN_F32)
        head_grad_kernel<float, true><<<grid, 256, 0, st>>>(head, grad_head, channels, channels_used, hw, channels_pad,
                                                            (float*)dz);
    else
        head_grad_kernel<__bf16, true><<<grid, 256, 0, st>>>(head, grad_head, channels, channels_used, hw, channels_pad,
                                                             (__bf16*)dz);
    PPN_LAUNCH_CHECK();
    if (dbias) {
        head_bias_grad_kernel<<<(channels + 3) / 4, 256, 0, st>>>(head, grad_head, batch, channels, hw, dbias);
        PPN_LAUNCH_CHECK();
    }
    return PPN_OK;
}

int ppn_nchw_to_nhwc(int32_t dtype, const float* src, int32_t batch, int32_t channels, int32_t hw, int32_t channels_used,
                     int32_t channels_pad, void* dst, void* stream) {
    if (!src || !dst) return ppn::fail(PPN_E_INVALID, "ppn_nchw_to_nhwc: NULL argument");
    if (dtype != PPN_F32 && dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    if (batch < 1 || channels < 1 || hw < 1 || channels_used < 1 || channels_used > channels ||
        channels_pad < channels_used || channels_pad % 64)
        return ppn::fail(PPN_E_INVALID, "ppn_nchw_to_nhwc: bad channel counts");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(channels_pad / 64, (hw + 63) / 64, batch);
    if (dtype == PPN_F32)
        head_grad_kernel<float, false><<<grid, 256, 0, st>>>(src, src, channels, channels_used, hw, channels_pad, (float*)dst);
    else
        head_grad_kernel<__bf16, false><<<grid, 256, 0, st>>>(src, src, channels, channels_used, hw, channels_pad, (__bf16*)dst);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

static int bn_act_mask_impl(const ppn_bn_bwd_desc* d, int nstreams, void* stream) {
    // uses x, dy (= the tensor to mask), gamma, beta, save_mean, save_rstd, act, dx (= output)
    if (!d || !d->x || !d->dy || !d->gamma || !d->beta || !d->save_mean || !d->save_rstd || !d->dx)
        return ppn::fail(PPN_E_INVALID, "ppn_bn_act_mask: NULL argument");
    if (nstreams < 1 || nstreams > 64) return ppn::fail(PPN_E_INVALID, "ppn_bn_act_mask_streams: 1 <= nstreams <= 64");
    Slab s;
    if (int rc = make_slab(d->channels, d->pixels, &s)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(s.nblocks, nstreams);
    const long long sstride = d->pixels * d->channels;
    if (d->dtype == PPN_F32)
        bn_act_mask_kernel<float><<<grid, kThreads, 0, st>>>((const float*)d->x, (const float*)d->dy, d->gamma, d->beta,
                                                             d->save_mean, d->save_rstd, d->act, d->pixels,
                                                             d->channels, s, (float*)d->dx, sstride);
    else if (d->dtype == PPN_BF16)
        bn_act_mask_kernel<__bf16><<<grid, kThreads, 0, st>>>((const __bf16*)d->x, (const __bf16*)d->dy, d->gamma,
                                                              d->beta, d->save_mean, d->save_rstd, d->act,
                                                              d->pixels, d->channels, s, (__bf16*)d->dx, sstride);
    else return ppn::fail(PPN_E_INVALID, "bad dtype %d", d->dtype);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_bn_act_mask(const ppn_bn_bwd_desc* d, void* stream) { return bn_act_mask_impl(d, 1, stream); }
int ppn_bn_act_mask_streams(const ppn_bn_bwd_desc* d, int32_t nstreams, void* stream) {
    return bn_act_mask_impl(d, nstreams, stream);
}

size_t ppn_bn_dual_workspace_bytes(int32_t channels) {
    if (channels <= 0) return 0;
    return (size_t)kMaxBlocks * channels * 5 * sizeof(double) + (size_t)4 * channels * sizeof(float);
}

static int bn_dual_bwd_impl(const ppn_bn_bwd_desc* d, const void* xdot, float* dgamma_tan, int nstreams, void* stream,
                            bool shared_dx = false) {
    // d->dy = adjoint arriving at the (activated) tangent output, d->dx is ACCUMULATED into, d->workspace >=
    // nstreams * ppn_bn_dual_workspace_bytes(channels); d->dgamma / dbeta / dx_add are not used
    if (!d || !d->x || !d->dy || !xdot || !d->gamma || !d->beta || !d->save_mean || !d->save_rstd || !d->dx ||
        !d->workspace || !dgamma_tan)
        return ppn::fail(PPN_E_INVALID, "ppn_bn_dual_bwd: NULL argument");
    if (d->dtype != PPN_F32 && d->dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", d->dtype);
    if (nstreams < 1 || nstreams > 64) return ppn::fail(PPN_E_INVALID, "ppn_bn_dual_bwd_streams: 1 <= nstreams <= 64");
    Slab s;
    if (int rc = make_slab(d->channels, d->pixels, &s)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int C = d->channels;
    const long long pstride = (long long)kMaxBlocks * C * 5, sstride = d->pixels * C;
    double* partial = reinterpret_cast<double*>(d->workspace);
    float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(d->workspace) + (size_t)nstreams * pstride * sizeof(double));
    const dim3 grid(s.nblocks, nstreams), fgrid((C + 3) / 4, nstreams);
    if (d->dtype == PPN_F32)
        bn_dual_reduce_kernel<float><<<grid, kThreads, 0, st>>>((const float*)d->x, (const float*)xdot,
                                                                (const float*)d->dy, d->gamma, d->beta, d->save_mean,
                                                                d->save_rstd, d->act, d->pixels, C, s, partial, sstride, pstride);
    else
        bn_dual_reduce_kernel<__bf16><<<grid, kThreads, 0, st>>>((const __bf16*)d->x, (const __bf16*)xdot,
                                                                 (const __bf16*)d->dy, d->gamma, d->beta,
                                                                 d->save_mean, d->save_rstd, d->act, d->pixels, C, s,
                                                                 partial, sstride, pstride);
    PPN_LAUNCH_CHECK();
    bn_dual_finalize_kernel<<<fgrid, 256, 0, st>>>(partial, s.nblocks, C, d->pixels, d->gamma, d->save_rstd, coef,
                                                    dgamma_tan, pstride);
    PPN_LAUNCH_CHECK();
    if (shared_dx) {
        // every stream's term accumulates into the SAME dx (the caller only needs the sum over the streams): one launch per
        // stream, in stream order (stream order = summation order: reproducible)
        const size_t es = d->dtype == PPN_F32 ? 4 : 2;
        for (int g = 0; g < nstreams; ++g) {
            const char* xd = static_cast<const char*>(xdot) + (size_t)g * sstride * es;
            const char* dyt = static_cast<const char*>(d->dy) + (size_t)g * sstride * es;
            const float* cf = coef + (size_t)g * 4 * C;
            if (d->dtype == PPN_F32)
                bn_dual_apply_kernel<float><<<s.nblocks, kThreads, 0, st>>>((const float*)d->x, (const float*)xd, (const float*)dyt,
                                                                            d->gamma, d->beta, d->save_mean, d->save_rstd, cf,
                                                                            d->act, d->pixels, C, s, (float*)d->dx, 0);
            else
                bn_dual_apply_kernel<__bf16><<<s.nblocks, kThreads, 0, st>>>((const __bf16*)d->x, (const __bf16*)xd,
                                                                             (const __bf16*)dyt, d->gamma, d->beta, d->save_mean,
                                                                             d->save_rstd, cf, d->act, d->pixels, C, s,
                                                                             (__bf16*)d->dx, 0);
            PPN_LAUNCH_CHECK();
        }
        return PPN_OK;
    }
    if (d->dtype == PPN_F32)
        bn_dual_apply_kernel<float><<<grid, kThreads, 0, st>>>((const float*)d->x, (const float*)xdot,
                                                               (const float*)d->dy, d->gamma, d->beta, d->save_mean,
                                                               d->save_rstd, coef, d->act, d->pixels, C, s,
                                                               (float*)d->dx, sstride);
    else
        bn_dual_apply_kernel<__bf16><<<grid, kThreads, 0, st>>>((const __bf16*)d->x, (const __bf16*)xdot,
                                                                (const __bf16*)d->dy, d->gamma, d->beta,
                                                                d->save_mean, d->save_rstd, coef, d->act, d->pixels,
                                                                C, s, (__bf16*)d->dx, sstride);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_bn_dual_bwd(const ppn_bn_bwd_desc* d, const void* xdot, float* dgamma_tan, void* stream) {
    return bn_dual_bwd_impl(d, xdot, dgamma_tan, 1, stream);
}
int ppn_bn_dual_bwd_streams(const ppn_bn_bwd_desc* d, const void* xdot, float* dgamma_tan, int32_t nstreams, void* stream) {
    return bn_dual_bwd_impl(d, xdot, dgamma_tan, nstreams, stream);
}
int ppn_bn_dual_bwd_streams_sum(const ppn_bn_bwd_desc* d, const void* xdot, float* dgamma_tan, int32_t nstreams,
                                void* stream) {
    return bn_dual_bwd_impl(d, xdot, dgamma_tan, nstreams, stream, true);
}

int ppn_add_relu(int32_t dtype, const void* z, const void* r, int64_t n, void* out, void* stream) {
    if (!z || !r || !out || n < 0 || n % 8) return ppn::fail(PPN_E_INVALID, "ppn_add_relu: NULL pointer or n %% 8 != 0");
    long long blocks = (n / 8 + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) return PPN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PPN_F32) add_relu_kernel<float><<<(int)blocks, kThreads, 0, st>>>((const float*)z, (const float*)r, n / 8, (float*)out);
    else if (dtype == PPN_BF16) add_relu_kernel<__bf16><<<(int)blocks, kThreads, 0, st>>>((const __bf16*)z, (const __bf16*)r, n / 8, (__bf16*)out);
    else return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_relu_mask(int32_t dtype, const void* out, const void* dout, const void* add, int64_t n, void* dz, void* stream) {
    if (!out || !dout || !dz || n < 0 || n % 8) return ppn::fail(PPN_E_INVALID, "ppn_relu_mask: NULL pointer or n %% 8 != 0");
    long long blocks = (n / 8 + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) return PPN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PPN_F32)
        relu_mask_kernel<float><<<(int)blocks, kThreads, 0, st>>>((const float*)out, (const float*)dout, (const float*)add, n / 8, (float*)dz);
    else if (dtype == PPN_BF16)
        relu_mask_kernel<__bf16><<<(int)blocks, kThreads, 0, st>>>((const __bf16*)out, (const __bf16*)dout, (const __bf16*)add, n / 8, (__bf16*)dz);
    else return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_upsample_zero(int32_t dtype, const void* src, int32_t batch, int32_t src_h, int32_t src_w, int32_t channels,
                      int32_t stride, int32_t dst_h, int32_t dst_w, void* dst, void* stream) {
    if (!src || !dst || batch < 1 || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1 || stride < 1 || channels < 1)
        return ppn::fail(PPN_E_INVALID, "ppn_upsample_zero: bad arguments");
    if (dtype != PPN_F32 && dtype != PPN_BF16 && dtype != PPN_F16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    const int es = dtype == PPN_F32 ? 4 : 2;
    if ((channels * es) % 16) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_upsample_zero: a pixel must be a multiple of 16 bytes");
    if ((long long)(src_h - 1) * stride >= dst_h || (long long)(src_w - 1) * stride >= dst_w)
        return ppn::fail(PPN_E_INVALID, "ppn_upsample_zero: the source does not fit the destination");
    const int c16 = channels * es / 16;
    const long long n = (long long)batch * dst_h * dst_w * c16;
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    upsample_zero_kernel<<<(int)blocks, kThreads, 0, (hipStream_t)stream>>>((const uint4*)src, (uint4*)dst, batch, src_h, src_w,
                                                                            dst_h, dst_w, c16, stride);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_interleave_parity(int32_t dtype, const void* o00, const void* o01, const void* o10, const void* o11, int32_t batch,
                          int32_t h, int32_t w, int32_t channels, void* dx, void* stream) {
    if (!o00 || !o01 || !o10 || !o11 || !dx || batch < 1 || h < 1 || w < 1 || channels < 1)
        return ppn::fail(PPN_E_INVALID, "ppn_interleave_parity: bad arguments");
    if (dtype != PPN_F32 && dtype != PPN_BF16 && dtype != PPN_F16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    const int es = dtype == PPN_F32 ? 4 : 2;
    if ((channels * es) % 16) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_interleave_parity: a pixel must be a multiple of 16 bytes");
    const int c16 = channels * es / 16;
    const int Ho1 = (h + 2 - 3) / 2 + 2, Wo1 = (w + 2 - 3) / 2 + 2;       // the sub-convolutions' output: (Ho + 1) x (Wo + 1)
    const long long n = (long long)batch * h * w * c16;
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    interleave_parity_kernel<<<(int)blocks, kThreads, 0, (hipStream_t)stream>>>((const uint4*)o00, (const uint4*)o01,
                                                                                (const uint4*)o10, (const uint4*)o11, (uint4*)dx,
                                                                                batch, h, w, Ho1, Wo1, c16);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_interleave_parity_stacked(int32_t dtype, const void* o, int32_t batch, int32_t h, int32_t w, int32_t channels, void* dx,
                                  void* stream) {
    if (!o || !dx || batch < 1 || h < 1 || w < 1 || channels < 1)
        return ppn::fail(PPN_E_INVALID, "ppn_interleave_parity_stacked: bad arguments");
    if (dtype != PPN_F32 && dtype != PPN_BF16 && dtype != PPN_F16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    const int es = dtype == PPN_F32 ? 4 : 2;
    if ((channels * es) % 16) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_interleave_parity_stacked: a pixel must be a multiple of 16 bytes");
    const int c16 = channels * es / 16;
    const int Ho1 = (h + 2 - 3) / 2 + 2, Wo1 = (w + 2 - 3) / 2 + 2;
    const long long n = (long long)batch * h * w * c16;
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    interleave_parity_stacked_kernel<<<(int)blocks, kThreads, 0, (hipStream_t)stream>>>((const uint4*)o, (uint4*)dx, batch, h, w,
                                                                                       Ho1, Wo1, c16);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_image_to_nhwc(int32_t dtype, const float* src, int32_t batch, int32_t h, int32_t w, int32_t channels_pad, void* dst,
                      void* stream) {
    if (!src || !dst || batch < 1 || h < 1 || w < 1 || (channels_pad != 4 && channels_pad != 8))
        return ppn::fail(PPN_E_INVALID, "ppn_image_to_nhwc: bad arguments (channels_pad 4 or 8)");
    const long long HW = (long long)h * w, n = (long long)batch * HW;
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PPN_F32 && channels_pad == 8) image_to_nhwc_kernel<float, 8><<<(int)blocks, kThreads, 0, st>>>(src, (float*)dst, batch, HW);
    else if (dtype == PPN_F32) image_to_nhwc_kernel<float, 4><<<(int)blocks, kThreads, 0, st>>>(src, (float*)dst, batch, HW);
    else if (dtype == PPN_BF16 && channels_pad == 8) image_to_nhwc_kernel<__bf16, 8><<<(int)blocks, kThreads, 0, st>>>(src, (__bf16*)dst, batch, HW);
    else if (dtype == PPN_BF16) image_to_nhwc_kernel<__bf16, 4><<<(int)blocks, kThreads, 0, st>>>(src, (__bf16*)dst, batch, HW);
    else return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                  double beta1, double beta2, double eps, double weight_decay, int32_t step, float grad_scale,
                  void* param_lp, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1)
        return ppn::fail(PPN_E_INVALID, "ppn_adam_step: NULL buffer, n < 0 or step < 1");
    if (n == 0) return PPN_OK;
    float step_size, inv_bc2_sqrt;
    adam_consts(lr, beta1, beta2, step, &step_size, &inv_bc2_sqrt);
    long long blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    // 1 - beta is formed in double from the caller's value, as torch does, then rounded once
    adam_kernel<<<(int)blocks, kThreads, 0, (hipStream_t)stream>>>(
        param, grad, exp_avg, exp_avg_sq, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
        (float)weight_decay, step_size, inv_bc2_sqrt, grad_scale, (__bf16*)param_lp);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_sumsq(const float* x, int64_t n, float* out, void* workspace, void* stream) {
    if (!x || !out || !workspace || n < 0) return ppn::fail(PPN_E_INVALID, "ppn_sumsq: NULL argument");
    long long blocks = (n + kThreads * 8 - 1) / (kThreads * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    sumsq_partial_kernel<<<(int)blocks, kThreads, 0, st>>>(x, n, (double*)workspace);
    PPN_LAUNCH_CHECK();
    sumsq_final_kernel<<<1, 1, 0, st>>>((const double*)workspace, (int)blocks, out);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_gradnorm_probe_stats(const float* g0, const float* g1, const float* g2, const float* g3, const float* total,
                             const float* coeff, int64_t n, float* gw4, float* stats, void* workspace, void* stream) {
    if (!g0 || !g1 || !g2 || !g3 || !total || !coeff || !gw4 || !stats || !workspace || n < 0)
        return ppn::fail(PPN_E_INVALID, "ppn_gradnorm_probe_stats: NULL argument");
    if (!(coeff[4] != 0.f)) return ppn::fail(PPN_E_INVALID, "ppn_gradnorm_probe_stats: coeff[4] must not be zero");
    long long blocks = (n + kThreads * 8 - 1) / (kThreads * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    ProbeStatsArgs a;
    a.g[0] = g0; a.g[1] = g1; a.g[2] = g2; a.g[3] = g3; a.total = total;
    for (int i = 0; i < 5; ++i) a.c[i] = coeff[i];
    a.n = n; a.gw4 = gw4; a.partial = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    probe_stats_partial_kernel<<<(int)blocks, kThreads, 0, st>>>(a);
    PPN_LAUNCH_CHECK();
    probe_stats_final_kernel<<<1, 64, 0, st>>>((const double*)workspace, (int)blocks, stats);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_gradnorm_weight_step(float* w, const float* losses, const float* gnorm, const float* base, float alpha,
                             float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                             int32_t step, float* out5x4, void* stream) {
    if (!w || !losses || !gnorm || !base || !exp_avg || !exp_avg_sq || step < 1)
        return ppn::fail(PPN_E_INVALID, "ppn_gradnorm_weight_step: NULL argument or step < 1");
    float step_size, inv_bc2_sqrt;
    adam_consts(lr, beta1, beta2, step, &step_size, &inv_bc2_sqrt);
    gradnorm_kernel<<<1, 64, 0, (hipStream_t)stream>>>(w, losses, gnorm, base, alpha, exp_avg, exp_avg_sq,
                                                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                                                       (float)eps, step_size, inv_bc2_sqrt, out5x4);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int ppn_gradnorm_renorm(float* w, int32_t world_size, void* stream) {
    if (!w || world_size < 1) return ppn::fail(PPN_E_INVALID, "ppn_gradnorm_renorm: NULL w or world_size < 1");
    gradnorm_renorm_kernel<<<1, 64, 0, (hipStream_t)stream>>>(w, (float)world_size);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

}  // extern "C"
