// norm.hip -- BatchNorm1d + tanh over [N,d] for gfx950 (replaces self.layer_act(self.bn(.)),
// src/jmac_model.py:52; nn.BatchNorm1d semantics: biased batch variance for normalisation, unbiased
// for the running estimate, eps inside the sqrt).
//
// HBM-bound streaming passes with 16 B per lane.  Column statistics use sums shifted by row 0 of the
// column (single pass, no catastrophic cancellation), reduced deterministically: per-block partial
// rows -> one finalising block.  No atomics.
#include "common.h"

using namespace jmac;

namespace {

constexpr int kBlock = 256;
constexpr int kStatBlocks = 512;

// partial[b][0][c] = sum_r (x[r][c]-K[c]),  partial[b][1][c] = sum_r (x[r][c]-K[c])^2, K = x[0][:]
// rows are dealt to blocks round-robin in groups of RPB = kBlock / D4 rows.
__global__ __launch_bounds__(kBlock) void col_stats_partial_kernel(const float* __restrict__ x, int64_t ldx, int64_t N,
                                                                   int D4, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* red = reinterpret_cast<float4*>(smem);   // [2][rpb][D4]
    const int rpb = kBlock / D4 > 0 ? kBlock / D4 : 1;
    const int tid = threadIdx.x;
    // when D4 > kBlock a thread covers several column chunks
    for (int cbase = 0; cbase < D4; cbase += kBlock) {
        const int rsub = D4 >= kBlock ? 0 : tid / D4;
        const int c4 = D4 >= kBlock ? cbase + tid : tid % D4;
        const bool active = (D4 >= kBlock ? c4 < D4 : tid < rpb * D4);
        float4 s1 = f4zero(), s2 = f4zero();
        if (active) {
            const float4 K = ld4(x + c4 * 4);
            for (int64_t r = (int64_t)blockIdx.x * rpb + rsub; r < N; r += (int64_t)gridDim.x * rpb) {
                float4 v = ld4(x + r * ldx + c4 * 4);
                v.x -= K.x; v.y -= K.y; v.z -= K.z; v.w -= K.w;
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
            }
        }
        if (D4 >= kBlock) {
            if (active) {
                st4(partial + ((int64_t)blockIdx.x * 2 + 0) * D4 * 4 + c4 * 4, s1);
                st4(partial + ((int64_t)blockIdx.x * 2 + 1) * D4 * 4 + c4 * 4, s2);
            }
        } else {
            if (active) {
                red[(0 * rpb + rsub) * D4 + c4] = s1;
                red[(1 * rpb + rsub) * D4 + c4] = s2;
            }
            __syncthreads();
            if (tid < D4) {
                float4 a1 = red[tid], a2 = red[rpb * D4 + tid];
                for (int q = 1; q < rpb; ++q) {
                    float4 b1 = red[q * D4 + tid], b2 = red[(rpb + q) * D4 + tid];
                    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
                    a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
                }
                st4(partial + ((int64_t)blockIdx.x * 2 + 0) * D4 * 4 + tid * 4, a1);
                st4(partial + ((int64_t)blockIdx.x * 2 + 1) * D4 * 4 + tid * 4, a2);
            }
            break;
        }
    }
}

// sums[0][c], sums[1][c] = reduced partials
__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ x, int64_t N,
                                   int d, float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const float s1 = sums[c], s2 = sums[d + c];
    const float n = (float)N;
    const float K = x[c];
    const float mshift = s1 / n;
    const float mean = K + mshift;
    float var = s2 / n - mshift * mshift;
    var = var > 0.f ? var : 0.f;
    save_mean[c] = mean;
    save_invstd[c] = rsqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) {
        const float unbiased = N > 1 ? var * n / (n - 1.f) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// reduce_rows over the [nparts, 2d] partial sums and bn_finalize in one launch: a block owns 16 features and sums both
// their s1 and s2 columns (64 row lanes each), then finalises them
constexpr int RF_COLS = 16, RF_LANES = 1024 / RF_COLS;
__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const float* __restrict__ partial, int nparts,
                                                                  const float* __restrict__ x, int64_t N, int d, float eps,
                                                                  float momentum, float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, float* __restrict__ save_mean,
                                                                  float* __restrict__ save_invstd) {
    __shared__ float red1[RF_LANES][RF_COLS], red2[RF_LANES][RF_COLS];
    const int cl = threadIdx.x % RF_COLS, rl = threadIdx.x / RF_COLS;
    const int c = blockIdx.x * RF_COLS + cl;
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    if (c < d) {
        int p = rl;
        for (; p + RF_LANES < nparts; p += 2 * RF_LANES) {
            a1 += partial[(int64_t)p * 2 * d + c];
            a2 += partial[(int64_t)p * 2 * d + d + c];
            b1 += partial[(int64_t)(p + RF_LANES) * 2 * d + c];
            b2 += partial[(int64_t)(p + RF_LANES) * 2 * d + d + c];
        }
        if (p < nparts) {
            a1 += partial[(int64_t)p * 2 * d + c];
            a2 += partial[(int64_t)p * 2 * d + d + c];
        }
    }
    red1[rl][cl] = a1 + b1;
    red2[rl][cl] = a2 + b2;
    __syncthreads();
    if (rl == 0 && c < d) {
        float s1 = red1[0][cl], s2 = red2[0][cl];
#pragma unroll 8
        for (int r = 1; r < RF_LANES; ++r) {
            s1 += red1[r][cl];
            s2 += red2[r][cl];
        }
        const float n = (float)N;
        const float mshift = s1 / n;
        const float mean = x[c] + mshift;
        float var = s2 / n - mshift * mshift;
        var = var > 0.f ? var : 0.f;
        save_mean[c] = mean;
        save_invstd[c] = rsqrtf(var + eps);
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) {
            const float unbiased = N > 1 ? var * n / (n - 1.f) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
}

// local column moments from the shifted sums: mean = K + s1/n, M2 = sum (x - mean)^2 = s2 - s1^2/n
__global__ void moments_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ x, int64_t N, int d,
                                        float* __restrict__ mean, float* __restrict__ m2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const float s1 = sums[c], s2 = sums[d + c], n = (float)N;
    mean[c] = x[c] + s1 / n;
    const float v = s2 - s1 * s1 / n;
    m2[c] = v > 0.f ? v : 0.f;
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var, int d,
                                     float eps, float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < d) {
        save_mean[c] = running_mean[c];
        save_invstd[c] = rsqrtf(running_var[c] + eps);
    }
}

__global__ __launch_bounds__(kBlock) void bn_tanh_apply_kernel(const float* __restrict__ x, int64_t ldx, int64_t N, int D4,
                                                               const float* __restrict__ weight, const float* __restrict__ bias,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               float* __restrict__ y, int64_t ldy, float* __restrict__ y2,
                                                               int64_t ldy2) {
    const int64_t total = N * D4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / D4;
        const int c4 = (int)(i % D4);
        const float4 v = ld4(x + r * ldx + c4 * 4);
        const float4 mu = ld4(mean + c4 * 4), is = ld4(invstd + c4 * 4), w = ld4(weight + c4 * 4), b = ld4(bias + c4 * 4);
        float4 o;
        o.x = tanhf(fmaf((v.x - mu.x) * is.x, w.x, b.x));
        o.y = tanhf(fmaf((v.y - mu.y) * is.y, w.y, b.y));
        o.z = tanhf(fmaf((v.z - mu.z) * is.z, w.z, b.z));
        o.w = tanhf(fmaf((v.w - mu.w) * is.w, w.w, b.w));
        st4(y + r * ldy + c4 * 4, o);
        if (y2) st4(y2 + r * ldy2 + c4 * 4, o);         // the same rows into a second (strided) destination: a cat operand
    }
}

// backward column sums: partial[b][0][c] = sum gz, partial[b][1][c] = sum gz * xhat, gz = gy*(1-y^2)
__global__ __launch_bounds__(kBlock) void bn_bwd_partial_kernel(const float* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ y, int64_t ldy,
                                                                const float* __restrict__ gy, int64_t ldgy,
                                                                const float* __restrict__ gy2, int64_t ldgy2, int64_t N, int D4,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* red = reinterpret_cast<float4*>(smem);
    const int rpb = kBlock / D4 > 0 ? kBlock / D4 : 1;
    const int tid = threadIdx.x;
    for (int cbase = 0; cbase < D4; cbase += kBlock) {
        const int rsub = D4 >= kBlock ? 0 : tid / D4;
        const int c4 = D4 >= kBlock ? cbase + tid : tid % D4;
        const bool active = (D4 >= kBlock ? c4 < D4 : tid < rpb * D4);
        float4 s1 = f4zero(), s2 = f4zero();
        if (active) {
            const float4 mu = ld4(mean + c4 * 4), is = ld4(invstd + c4 * 4);
            for (int64_t r = (int64_t)blockIdx.x * rpb + rsub; r < N; r += (int64_t)gridDim.x * rpb) {
                const float4 xv = ld4(x + r * ldx + c4 * 4), yv = ld4(y + r * ldy + c4 * 4);
                float4 g = ld4(gy + r * ldgy + c4 * 4);
                if (gy2) {                              // the output fed two consumers: their gradients are summed here
                    const float4 g2 = ld4(gy2 + r * ldgy2 + c4 * 4);
                    g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
                }
                const float gz0 = g.x * (1.f - yv.x * yv.x), gz1 = g.y * (1.f - yv.y * yv.y);
                const float gz2 = g.z * (1.f - yv.z * yv.z), gz3 = g.w * (1.f - yv.w * yv.w);
                s1.x += gz0; s1.y += gz1; s1.z += gz2; s1.w += gz3;
                s2.x = fmaf(gz0, (xv.x - mu.x) * is.x, s2.x); s2.y = fmaf(gz1, (xv.y - mu.y) * is.y, s2.y);
                s2.z = fmaf(gz2, (xv.z - mu.z) * is.z, s2.z); s2.w = fmaf(gz3, (xv.w - mu.w) * is.w, s2.w);
            }
        }
        if (D4 >= kBlock) {
            if (active) {
                st4(partial + ((int64_t)blockIdx.x * 2 + 0) * D4 * 4 + c4 * 4, s1);
                st4(partial + ((int64_t)blockIdx.x * 2 + 1) * D4 * 4 + c4 * 4, s2);
            }
        } else {
            if (active) {
                red[(0 * rpb + rsub) * D4 + c4] = s1;
                red[(1 * rpb + rsub) * D4 + c4] = s2;
            }
            __syncthreads();
            if (tid < D4) {
                float4 a1 = red[tid], a2 = red[rpb * D4 + tid];
                for (int q = 1; q < rpb; ++q) {
                    float4 b1 = red[q * D4 + tid], b2 = red[(rpb + q) * D4 + tid];
                    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
                    a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
                }
                st4(partial + ((int64_t)blockIdx.x * 2 + 0) * D4 * 4 + tid * 4, a1);
                st4(partial + ((int64_t)blockIdx.x * 2 + 1) * D4 * 4 + tid * 4, a2);
            }
            break;
        }
    }
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ sums, int d, float* __restrict__ gweight,
                                       float* __restrict__ gbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    gbias[c] = sums[c];
    gweight[c] = sums[d + c];
}

__global__ __launch_bounds__(kBlock) void bn_bwd_apply_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ gy, int64_t ldgy,
                                                              const float* __restrict__ gy2, int64_t ldgy2, int64_t N, int D4,
                                                              const float* __restrict__ weight, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gweight,
                                                              const float* __restrict__ gbias, int training, float invn,
                                                              float* __restrict__ gx, int64_t ldgx) {
    const int64_t total = N * D4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / D4;
        const int c4 = (int)(i % D4);
        const float4 xv = ld4(x + r * ldx + c4 * 4), yv = ld4(y + r * ldy + c4 * 4);
        float4 g = ld4(gy + r * ldgy + c4 * 4);
        if (gy2) {
            const float4 g2 = ld4(gy2 + r * ldgy2 + c4 * 4);
            g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
        }
        const float4 mu = ld4(mean + c4 * 4), is = ld4(invstd + c4 * 4), w = ld4(weight + c4 * 4);
        const float4 gw = ld4(gweight + c4 * 4), gb = ld4(gbias + c4 * 4);
        float4 o;
#define JMAC_BN_BWD(comp)                                                                 \
        {                                                                                 \
            const float gz = g.comp * (1.f - yv.comp * yv.comp);                          \
            const float xh = (xv.comp - mu.comp) * is.comp;                               \
            o.comp = training ? w.comp * is.comp * (gz - invn * (gb.comp + xh * gw.comp)) \
                              : w.comp * is.comp * gz;                                    \
        }
        JMAC_BN_BWD(x) JMAC_BN_BWD(y) JMAC_BN_BWD(z) JMAC_BN_BWD(w)
#undef JMAC_BN_BWD
        st4(gx + r * ldgx + c4 * 4, o);
    }
}


// ---- segmented forms: the rows are a stack of `nb` blocks (KGs of one launch set: the reference's training step encodes
// ---- two KGs per batch, src/jmac_model.py:325-326,263-264, each forward_base call with batch statistics of ITS rows) ----
// Every block has its own batch statistics (mean / invstd [nb, d]); weight, bias and the running estimates are the layer's
// (shared), and the running estimates are updated once per block in the callers' CALL order (seg.order), exactly what nb
// separate forward_base calls leave behind:  r <- (1-m)((1-m) r + m b_1) + m b_2 ...
constexpr int kMaxSeg = JMAC_BN_MAX_BLOCKS;
struct SegTab {
    int nb;
    int order[kMaxSeg];        // block processed k-th by the running-statistics update
    int gofs[kMaxSeg + 1];     // first statistics workgroup (= partial row) of each block
    int64_t ptr[kMaxSeg + 1];  // first row of each block; ptr[nb] = rows in all
};
__device__ __forceinline__ int seg_of_row(const SegTab& s, int64_t r) {
    int b = 0;
    for (int k = 1; k < s.nb; ++k) b += r >= s.ptr[k] ? 1 : 0;
    return b;
}
__device__ __forceinline__ int seg_of_group(const SegTab& s, int wg) {
    int b = 0;
    for (int k = 1; k < s.nb; ++k) b += wg >= s.gofs[k] ? 1 : 0;
    return b;
}

// col_stats_partial_kernel per block: workgroup wg works on block b = seg_of_group(wg); partial row index = wg
__global__ __launch_bounds__(kBlock) void col_stats_partial_seg_kernel(const float* __restrict__ x, int64_t ldx, SegTab seg, int D4,
                                                                       float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* red = reinterpret_cast<float4*>(smem);   // [2][rpb][D4]
    const int b = seg_of_group(seg, (int)blockIdx.x);
    const int lb = (int)blockIdx.x - seg.gofs[b], gb = seg.gofs[b + 1] - seg.gofs[b];
    const int64_t r0 = seg.ptr[b], r1 = seg.ptr[b + 1];
    const int rpb = kBlock / D4 > 0 ? kBlock / D4 : 1;
    const int tid = threadIdx.x;
    for (int cbase = 0; cbase < D4; cbase += kBlock) {
        const int rsub = D4 >= kBlock ? 0 : tid / D4;
        const int c4 = D4 >= kBlock ? cbase + tid : tid % D4;
        const bool active = (D4 >= kBlock ? c4 < D4 : tid < rpb * D4);
        float4 s1 = f4zero(), s2 = f4zero();
        if (active) {
            const float4 K = ld4(x + r0 * ldx + c4 * 4);
            for (int64_t r = r0 + (int64_t)lb * rpb + rsub; r < r1; r += (int64_t)gb * rpb) {
                float4 v = ld4(x + r * ldx + c4 * 4);
                v.x -= K.x; v.y -= K.y; v.z -= K.z; v.w -= K.w;
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
            }
        }
        if (D4 >= kBlock) {
            if (active) {
                st4(partial + ((int64_t)blockIdx.x * 2 + 0) * D4 * 4 + c4 * 4, s1);
                st4(partial + ((int64_t)blockIdx.x * 2 + 1) * D4 * 4 + c4 * 4, s2);
            }
        } else {
            if (active) {
                red[(0 * rpb + rsub) * D4 + c4] = s1;
                red[(1 * rpb + rsub) * D4 + c4] = s2;
            }
            __syncthreads();
            if (tid < D4) {
                float4 a1 = red[tid], a2 = red[rpb * D4 + tid];
                for (int q = 1; q < rpb; ++q) {
                    float4 b1 = red[q * D4 + tid], b2 = red[(rpb + q) * D4 + tid];
                    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
                    a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
                }
                st4(partial + ((int64_t)blockIdx.x * 2 + 0) * D4 * 4 + tid * 4, a1);
                st4(partial + ((int64_t)blockIdx.x * 2 + 1) * D4 * 4 + tid * 4, a2);
            }
            break;
        }
    }
}

// bn_reduce_finalize_kernel over the blocks in CALL order: a workgroup owns 16 features, reduces block after block and
// carries the running estimates of its features through the blocks in registers
__global__ __launch_bounds__(1024) void bn_reduce_finalize_seg_kernel(const float* __restrict__ partial, SegTab seg,
                                                                      const float* __restrict__ x, int64_t ldx, int d, float eps,
                                                                      float momentum, float* __restrict__ running_mean,
                                                                      float* __restrict__ running_var, float* __restrict__ save_mean,
                                                                      float* __restrict__ save_invstd) {
    __shared__ float red1[RF_LANES][RF_COLS], red2[RF_LANES][RF_COLS];
    const int cl = threadIdx.x % RF_COLS, rl = threadIdx.x / RF_COLS;
    const int c = blockIdx.x * RF_COLS + cl;
    float rm = 0.f, rv = 0.f;
    if (rl == 0 && c < d) {
        if (running_mean) rm = running_mean[c];
        if (running_var) rv = running_var[c];
    }
    for (int k = 0; k < seg.nb; ++k) {
        const int b = seg.order[k];
        const int p0 = seg.gofs[b], p1 = seg.gofs[b + 1];
        float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
        if (c < d) {
            int p = p0 + rl;
            for (; p + RF_LANES < p1; p += 2 * RF_LANES) {
                a1 += partial[(int64_t)p * 2 * d + c];
                a2 += partial[(int64_t)p * 2 * d + d + c];
                b1 += partial[(int64_t)(p + RF_LANES) * 2 * d + c];
                b2 += partial[(int64_t)(p + RF_LANES) * 2 * d + d + c];
            }
            if (p < p1) {
                a1 += partial[(int64_t)p * 2 * d + c];
                a2 += partial[(int64_t)p * 2 * d + d + c];
            }
        }
        red1[rl][cl] = a1 + b1;
        red2[rl][cl] = a2 + b2;
        __syncthreads();
        if (rl == 0 && c < d) {
            float s1 = red1[0][cl], s2 = red2[0][cl];
#pragma unroll 8
            for (int r = 1; r < RF_LANES; ++r) {
                s1 += red1[r][cl];
                s2 += red2[r][cl];
            }
            const int64_t nrow = seg.ptr[b + 1] - seg.ptr[b];
            const float n = (float)nrow;
            const float mshift = s1 / n;
            const float mean = x[seg.ptr[b] * ldx + c] + mshift;
            float var = s2 / n - mshift * mshift;
            var = var > 0.f ? var : 0.f;
            save_mean[(int64_t)b * d + c] = mean;
            save_invstd[(int64_t)b * d + c] = rsqrtf(var + eps);
            rm = (1.f - momentum) * rm + momentum * mean;
            const float unbiased = nrow > 1 ? var * n / (n - 1.f) : var;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        __syncthreads();
    }
    if (rl == 0 && c < d) {
        if (running_mean) running_mean[c] = rm;
        if (running_var) running_var[c] = rv;
    }
}

__global__ __launch_bounds__(kBlock) void bn_tanh_apply_seg_kernel(const float* __restrict__ x, int64_t ldx, SegTab seg, int D4,
                                                                   const float* __restrict__ weight, const float* __restrict__ bias,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   float* __restrict__ y, int64_t ldy, float* __restrict__ y2,
                                                                   int64_t ldy2) {
    const int64_t total = seg.ptr[seg.nb] * D4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / D4;
        const int c4 = (int)(i % D4);
        const int64_t so = (int64_t)seg_of_row(seg, r) * D4 * 4;
        const float4 v = ld4(x + r * ldx + c4 * 4);
        const float4 mu = ld4(mean + so + c4 * 4), is = ld4(invstd + so + c4 * 4), w = ld4(weight + c4 * 4), b = ld4(bias + c4 * 4);
        float4 o;
        o.x = tanhf(fmaf((v.x - mu.x) * is.x, w.x, b.x));
        o.y = tanhf(fmaf((v.y - mu.y) * is.y, w.y, b.y));
        o.z = tanhf(fmaf((v.z - mu.z) * is.z, w.z, b.z));
        o.w = tanhf(fmaf((v.w - mu.w) * is.w, w.w, b.w));
        st4(y + r * ldy + c4 * 4, o);
        if (y2) st4(y2 + r * ldy2 + c4 * 4, o);
    }
}

// bn_bwd_partial_kernel per block (mean / invstd of the workgroup's block)
__global__ __launch_bounds__(kBlock) void bn_bwd_partial_seg_kernel(const float* __restrict__ x, int64_t ldx,
                                                                    const float* __restrict__ y, int64_t ldy,
                                                                    const float* __restrict__ gy, int64_t ldgy,
                                                                    const float* __restrict__ gy2, int64_t ldgy2, SegTab seg, int D4,
                                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* red = reinterpret_cast<float4*>(smem);
    const int b = seg_of_group(seg, (int)blockIdx.x);
    const int lb = (int)blockIdx.x - seg.gofs[b], gb = seg.gofs[b + 1] - seg.gofs[b];
    const int64_t r0 = seg.ptr[b], r1 = seg.ptr[b + 1];
    const int64_t so = (int64_t)b * D4 * 4;
    const int rpb = kBlock / D4 > 0 ? kBlock / D4 : 1;
    const int tid = threadIdx.x;
    for (int cbase = 0; cbase < D4; cbase += kBlock) {
        const int rsub = D4 >= kBlock ? 0 : tid / D4;
        const int c4 = D4 >= kBlock ? cbase + tid : tid % D4;
        const bool active = (D4 >= kBlock ? c4 < D4 : tid < rpb * D4);
        float4 s1 = f4zero(), s2 = f4zero();
        if (active) {
            const float4 mu = ld4(mean + so + c4 * 4), is = ld4(invstd + so + c4 * 4);
            for (int64_t r = r0 + (int64_t)lb * rpb + rsub; r < r1; r += (int64_t)gb * rpb) {
                const float4 xv = ld4(x + r * ldx + c4 * 4), yv = ld4(y + r * ldy + c4 * 4);
                float4 g = ld4(gy + r * ldgy + c4 * 4);
                if (gy2) {
                    const float4 g2 = ld4(gy2 + r * ldgy2 + c4 * 4);
                    g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
                }
                const float gz0 = g.x * (1.f - yv.x * yv.x), gz1 = g.y * (1.f - yv.y * yv.y);
                const float gz2 = g.z * (1.f - yv.z * yv.z), gz3 = g.w * (1.f - yv.w * yv.w);
                s1.x += gz0; s1.y += gz1; s1.z += gz2; s1.w += gz3;
                s2.x = fmaf(gz0, (xv.x - mu.x) * is.x, s2.x); s2.y = fmaf(gz1, (xv.y - mu.y) * is.y, s2.y);
                s2.z = fmaf(gz2, (xv.z - mu.z) * is.z, s2.z); s2.w = fmaf(gz3, (xv.w - mu.w) * is.w, s2.w);
            }
        }
        if (D4 >= kBlock) {
            if (active) {
                st4(partial + ((int64_t)blockIdx.x * 2 + 0) * D4 * 4 + c4 * 4, s1);
                st4(partial + ((int64_t)blockIdx.x * 2 + 1) * D4 * 4 + c4 * 4, s2);
            }
        } else {
            if (active) {
                red[(0 * rpb + rsub) * D4 + c4] = s1;
                red[(1 * rpb + rsub) * D4 + c4] = s2;
            }
            __syncthreads();
            if (tid < D4) {
                float4 a1 = red[tid], a2 = red[rpb * D4 + tid];
                for (int q = 1; q < rpb; ++q) {
                    float4 b1 = red[q * D4 + tid], b2 = red[(rpb + q) * D4 + tid];
                    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
                    a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
                }
                st4(partial + ((int64_t)blockIdx.x * 2 + 0) * D4 * 4 + tid * 4, a1);
                st4(partial + ((int64_t)blockIdx.x * 2 + 1) * D4 * 4 + tid * 4, a2);
            }
            break;
        }
    }
}

// per-block sums of the backward partial rows -> bsums [nb, 2d] (sum gz | sum gz*xhat of each block) and their totals over
// the blocks in block order -> gbw [2d] = [grad bias | grad weight] (the parameters are shared by the blocks)
__global__ __launch_bounds__(1024) void bn_bwd_reduce_seg_kernel(const float* __restrict__ partial, SegTab seg, int W,
                                                                 float* __restrict__ bsums, float* __restrict__ gbias,
                                                                 float* __restrict__ gweight, int d) {
    __shared__ float red[RF_LANES][RF_COLS];
    const int cl = threadIdx.x % RF_COLS, rl = threadIdx.x / RF_COLS;
    const int c = blockIdx.x * RF_COLS + cl;
    float total = 0.f;
    for (int b = 0; b < seg.nb; ++b) {
        const int p0 = seg.gofs[b], p1 = seg.gofs[b + 1];
        float a0 = 0.f, a1 = 0.f;
        if (c < W) {
            int p = p0 + rl;
            for (; p + RF_LANES < p1; p += 2 * RF_LANES) {
                a0 += partial[(int64_t)p * W + c];
                a1 += partial[(int64_t)(p + RF_LANES) * W + c];
            }
            if (p < p1) a0 += partial[(int64_t)p * W + c];
        }
        red[rl][cl] = a0 + a1;
        __syncthreads();
        if (rl == 0 && c < W) {
            float s = red[0][cl];
#pragma unroll 8
            for (int r = 1; r < RF_LANES; ++r) s += red[r][cl];
            bsums[(int64_t)b * W + c] = s;
            total += s;
        }
        __syncthreads();
    }
    if (rl == 0 && c < W) {
        if (c < d) gbias[c] = total;
        else gweight[c - d] = total;
    }
}

__global__ __launch_bounds__(kBlock) void bn_bwd_apply_seg_kernel(const float* __restrict__ x, int64_t ldx,
                                                                  const float* __restrict__ y, int64_t ldy,
                                                                  const float* __restrict__ gy, int64_t ldgy,
                                                                  const float* __restrict__ gy2, int64_t ldgy2, SegTab seg, int D4,
                                                                  const float* __restrict__ weight, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ bsums,
                                                                  float* __restrict__ gx, int64_t ldgx) {
    const int64_t total = seg.ptr[seg.nb] * D4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / D4;
        const int c4 = (int)(i % D4);
        const int b = seg_of_row(seg, r);
        const int64_t so = (int64_t)b * D4 * 4;
        const float invn = 1.f / (float)(seg.ptr[b + 1] - seg.ptr[b]);
        const float4 xv = ld4(x + r * ldx + c4 * 4), yv = ld4(y + r * ldy + c4 * 4);
        float4 g = ld4(gy + r * ldgy + c4 * 4);
        if (gy2) {
            const float4 g2 = ld4(gy2 + r * ldgy2 + c4 * 4);
            g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
        }
        const float4 mu = ld4(mean + so + c4 * 4), is = ld4(invstd + so + c4 * 4), w = ld4(weight + c4 * 4);
        const float4 gb = ld4(bsums + 2 * so + c4 * 4), gw = ld4(bsums + 2 * so + D4 * 4 + c4 * 4);
        float4 o;
#define JMAC_BN_BWD(comp)                                                                 \
        {                                                                                 \
            const float gz = g.comp * (1.f - yv.comp * yv.comp);                          \
            const float xh = (xv.comp - mu.comp) * is.comp;                               \
            o.comp = w.comp * is.comp * (gz - invn * (gb.comp + xh * gw.comp));           \
        }
        JMAC_BN_BWD(x) JMAC_BN_BWD(y) JMAC_BN_BWD(z) JMAC_BN_BWD(w)
#undef JMAC_BN_BWD
        st4(gx + r * ldgx + c4 * 4, o);
    }
}

inline unsigned stat_grid(int64_t N, int D4) {
    const int rpb = kBlock / D4 > 0 ? kBlock / D4 : 1;
    int64_t need = (N + rpb - 1) / rpb;
    if (need < 1) need = 1;
    return (unsigned)(need < kStatBlocks ? need : kStatBlocks);
}
inline size_t stat_smem(int D4) {
    const int rpb = kBlock / D4 > 0 ? kBlock / D4 : 1;
    return D4 >= kBlock ? 16 : (size_t)2 * rpb * D4 * 16;
}
inline unsigned stream_grid(int64_t total) {
    int64_t need = (total + kBlock - 1) / kBlock;
    if (need < 1) need = 1;
    return (unsigned)(need < 8192 ? need : 8192);
}

// ---- row L2 normalisation: y = x / max(||x||_2, eps)  (F.normalize(x, 2, -1), src/jmac_model.py:179,191,227-228) ------
// one wave per row, scalar lane layout (any d, any 4-byte-aligned leading dimension); inv[r] = 1 / max(||x_r||, eps).
__global__ __launch_bounds__(kBlock) void row_normalize_fwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t N, int d,
                                                                   float eps, float* __restrict__ y, int64_t ldy,
                                                                   float* __restrict__ inv) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t r = w0; r < N; r += nw) {
        const float* xr = x + r * ldx;
        float ss = 0.f;
        for (int c = lane; c < d; c += 64) ss = fmaf(xr[c], xr[c], ss);
        ss = wave_sum(ss);
        const float iv = 1.f / fmaxf(sqrtf(ss), eps);
        for (int c = lane; c < d; c += 64) y[r * ldy + c] = xr[c] * iv;
        if (lane == 0) inv[r] = iv;
    }
}

// gx = inv * (g - y (g.y))   where the norm exceeds eps; gx = inv * g where it was clamped (constant denominator)
__global__ __launch_bounds__(kBlock) void row_normalize_bwd_kernel(const float* __restrict__ y, int64_t ldy,
                                                                   const float* __restrict__ g, int64_t ldg,
                                                                   const float* __restrict__ inv, int64_t N, int d, float eps,
                                                                   float* __restrict__ gx, int64_t ldgx) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t r = w0; r < N; r += nw) {
        const float* yr = y + r * ldy;
        const float* gr = g + r * ldg;
        float dot = 0.f;
        for (int c = lane; c < d; c += 64) dot = fmaf(gr[c], yr[c], dot);
        dot = wave_sum(dot);
        const float iv = inv[r];
        if (iv * eps >= 1.f) dot = 0.f;                // ||x|| <= eps: y = x / eps, no radial term
        for (int c = lane; c < d; c += 64) gx[r * ldgx + c] = iv * (gr[c] - yr[c] * dot);
    }
}

// ---- F.normalize followed by dropout (src/jmac_model.py:179,191: completion_dropout(F.normalize(.))), one pass each way --------
// y = x * inv * m  with m = mask * scale (mask: the caller's {0,1} draws; NULL = no dropout).  The backward recomputes the
// normalised row from x and inv (nothing but inv [N] is kept from the forward):
//   gn = g * m;  gx (+)= inv * (gn - xn (gn . xn)),  xn = x * inv   (no radial term on rows whose norm was clamped)
// one wave per row, 16 bytes per lane (d % 4 == 0, leading dimensions % 4 == 0).
// Dropout draws made in the kernel (SEED form): Philox4x32-10 keyed by the caller's device-resident seed, counter = index of
// the float4 (row * d/4 + column quad): its four words decide the four elements.  The backward regenerates the same draws, so
// no mask tensor is written or read (28 MB each way at DBP-5L size, plus the launch that drew it).
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
struct DropSrc {
    const float* mask;       // {0,1} draws of the caller, or NULL
    int64_t ldm;
    const int64_t* seed;     // SEED form: device int64
    uint32_t keep_thr;       // keep iff word < keep_thr  (keep probability = keep_thr / 2^32)
    float scale;
};
template <bool SEED>
__device__ __forceinline__ float4 drop_factors(const DropSrc& s, uint2 key, int64_t r, int c, int D4) {
    if (SEED) {
        const uint64_t idx = (uint64_t)r * (uint64_t)D4 + (uint64_t)c;
        const uint4 w = philox4x32_10(make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), 0u, 0u), key);
        return make_float4(w.x < s.keep_thr ? s.scale : 0.f, w.y < s.keep_thr ? s.scale : 0.f, w.z < s.keep_thr ? s.scale : 0.f,
                           w.w < s.keep_thr ? s.scale : 0.f);
    }
    const float4 m = ld4(s.mask + r * s.ldm + c * 4);
    return make_float4(m.x * s.scale, m.y * s.scale, m.z * s.scale, m.w * s.scale);
}

template <bool SEED>
__global__ __launch_bounds__(kBlock) void row_normalize_drop_fwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t N, int D4,
                                                                        float eps, DropSrc ds, float* __restrict__ y, int64_t ldy,
                                                                        float* __restrict__ inv) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
    uint2 key = make_uint2(0u, 0u);
    if (SEED) {
        const uint64_t sd = (uint64_t)ds.seed[0];
        key = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
    }
    const bool drop = SEED || ds.mask != nullptr;
    for (int64_t r = w0; r < N; r += nw) {
        const float* xr = x + r * ldx;
        float ss = 0.f;
        for (int c = lane; c < D4; c += 64) {
            const float4 v = ld4(xr + c * 4);
            ss = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, ss))));
        }
        ss = wave_sum(ss);
        const float iv = 1.f / fmaxf(sqrtf(ss), eps);
        for (int c = lane; c < D4; c += 64) {
            float4 v = ld4(xr + c * 4);
            v.x *= iv; v.y *= iv; v.z *= iv; v.w *= iv;
            if (drop) {
                const float4 m = drop_factors<SEED>(ds, key, r, c, D4);
                v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            }
            st4(y + r * ldy + c * 4, v);
        }
        if (lane == 0) inv[r] = iv;
    }
}

template <bool SEED>
__global__ __launch_bounds__(kBlock) void row_normalize_drop_bwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                                        const float* __restrict__ inv, DropSrc ds,
                                                                        const float* __restrict__ g, int64_t ldg, int64_t N, int D4,
                                                                        float eps, float* __restrict__ gx, int64_t ldgx,
                                                                        int accumulate) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
    uint2 key = make_uint2(0u, 0u);
    if (SEED) {
        const uint64_t sd = (uint64_t)ds.seed[0];
        key = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
    }
    const bool drop = SEED || ds.mask != nullptr;
    for (int64_t r = w0; r < N; r += nw) {
        const float* xr = x + r * ldx;
        const float* gr = g + r * ldg;
        const float iv = inv[r];
        float dot = 0.f;
        for (int c = lane; c < D4; c += 64) {
            const float4 v = ld4(xr + c * 4);
            float4 q = ld4(gr + c * 4);
            if (drop) {
                const float4 m = drop_factors<SEED>(ds, key, r, c, D4);
                q.x *= m.x; q.y *= m.y; q.z *= m.z; q.w *= m.w;
            }
            dot = fmaf(q.x, v.x * iv, fmaf(q.y, v.y * iv, fmaf(q.z, v.z * iv, fmaf(q.w, v.w * iv, dot))));
        }
        dot = wave_sum(dot);
        if (iv * eps >= 1.f) dot = 0.f;                // ||x|| <= eps: y = x / eps, no radial term
        for (int c = lane; c < D4; c += 64) {
            const float4 v = ld4(xr + c * 4);
            float4 q = ld4(gr + c * 4);
            if (drop) {
                const float4 m = drop_factors<SEED>(ds, key, r, c, D4);
                q.x *= m.x; q.y *= m.y; q.z *= m.z; q.w *= m.w;
            }
            float4 o;
            o.x = iv * (q.x - v.x * iv * dot); o.y = iv * (q.y - v.y * iv * dot);
            o.z = iv * (q.z - v.z * iv * dot); o.w = iv * (q.w - v.w * iv * dot);
            if (accumulate) {
                const float4 p = ld4(gx + r * ldgx + c * 4);
                o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
            }
            st4(gx + r * ldgx + c * 4, o);
        }
    }
}

// ---- the same two kernels with R rows per wave in flight (rows of up to 128 float4: d <= 512) ---------------------------------
// The forms above give every row a wave that reads it, reduces, reads it AGAIN and writes: two dependent round trips per row and
// (backward) two Philox evaluations per element; at DBP-5L size (11 805 rows x 300) they move 28-56 MB at 2.2 TB/s.  Here a wave
// keeps R rows in registers (two float4 per lane and row), issues all their loads before the first reduction (interleaved DPP
// chains: wave_sum_n) and writes from registers.  Per-lane summation order as above: the same bits.
template <bool SEED, int R>
__global__ __launch_bounds__(kBlock) void row_normalize_drop_fwd_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t N, int D4,
                                                                             float eps, DropSrc ds, float* __restrict__ y, int64_t ldy,
                                                                             float* __restrict__ inv) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
    uint2 key = make_uint2(0u, 0u);
    if (SEED) {
        const uint64_t sd = (uint64_t)ds.seed[0];
        key = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
    }
    const bool drop = SEED || ds.mask != nullptr;
    const int cc[2] = {lane, lane + 64};
    const bool ok[2] = {cc[0] < D4, cc[1] < D4};
    for (int64_t r0 = w0 * R; r0 < N; r0 += nw * R) {
        float4 v[R][2];
        float ss[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const float* xr = x + (r0 + u < N ? r0 + u : N - 1) * ldx;
#pragma unroll
            for (int k = 0; k < 2; ++k) v[u][k] = ld4(xr + (ok[k] ? cc[k] : 0) * 4);
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            ss[u] = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!ok[k]) v[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 q = v[u][k];
                ss[u] = fmaf(q.x, q.x, fmaf(q.y, q.y, fmaf(q.z, q.z, fmaf(q.w, q.w, ss[u]))));
            }
        }
        wave_sum_n<R>(ss);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t r = r0 + u;
            if (r >= N) break;                                     // wave-uniform
            const float iv = 1.f / fmaxf(sqrtf(ss[u]), eps);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!ok[k]) continue;
                float4 q = v[u][k];
                q.x *= iv; q.y *= iv; q.z *= iv; q.w *= iv;
                if (drop) {
                    const float4 m = drop_factors<SEED>(ds, key, r, cc[k], D4);
                    q.x *= m.x; q.y *= m.y; q.z *= m.z; q.w *= m.w;
                }
                st4(y + r * ldy + cc[k] * 4, q);
            }
            if (lane == 0) inv[r] = iv;
        }
    }
}

template <bool SEED, int R>
__global__ __launch_bounds__(kBlock) void row_normalize_drop_bwd_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                                             const float* __restrict__ inv, DropSrc ds,
                                                                             const float* __restrict__ g, int64_t ldg, int64_t N, int D4,
                                                                             float eps, float* __restrict__ gx, int64_t ldgx,
                                                                             int accumulate) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
    uint2 key = make_uint2(0u, 0u);
    if (SEED) {
        const uint64_t sd = (uint64_t)ds.seed[0];
        key = make_uint2((uint32_t)sd, (uint32_t)(sd >> 32));
    }
    const bool drop = SEED || ds.mask != nullptr;
    const int cc[2] = {lane, lane + 64};
    const bool ok[2] = {cc[0] < D4, cc[1] < D4};
    for (int64_t r0 = w0 * R; r0 < N; r0 += nw * R) {
        float4 v[R][2], q[R][2], p[R][2];
        float iv[R], dot[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t r = r0 + u < N ? r0 + u : N - 1;
            iv[u] = inv[r];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = ok[k] ? cc[k] : 0;
                v[u][k] = ld4(x + r * ldx + c * 4);
                q[u][k] = ld4(g + r * ldg + c * 4);
                if (accumulate) p[u][k] = ld4(gx + r * ldgx + c * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t r = r0 + u < N ? r0 + u : N - 1;
            dot[u] = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!ok[k]) {
                    q[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    v[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if (drop) {
                    const float4 m = drop_factors<SEED>(ds, key, r, cc[k], D4);
                    q[u][k].x *= m.x; q[u][k].y *= m.y; q[u][k].z *= m.z; q[u][k].w *= m.w;
                }
                const float4 a = v[u][k], b = q[u][k];
                dot[u] = fmaf(b.x, a.x * iv[u], fmaf(b.y, a.y * iv[u], fmaf(b.z, a.z * iv[u], fmaf(b.w, a.w * iv[u], dot[u]))));
            }
        }
        wave_sum_n<R>(dot);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t r = r0 + u;
            if (r >= N) break;                                     // wave-uniform
            const float d_ = iv[u] * eps >= 1.f ? 0.f : dot[u];   // ||x|| <= eps: y = x / eps, no radial term
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!ok[k]) continue;
                const float4 a = v[u][k], b = q[u][k];
                float4 o;
                o.x = iv[u] * (b.x - a.x * iv[u] * d_); o.y = iv[u] * (b.y - a.y * iv[u] * d_);
                o.z = iv[u] * (b.z - a.z * iv[u] * d_); o.w = iv[u] * (b.w - a.w * iv[u] * d_);
                if (accumulate) {
                    o.x += p[u][k].x; o.y += p[u][k].y; o.z += p[u][k].z; o.w += p[u][k].w;
                }
                st4(gx + r * ldgx + cc[k] * 4, o);
            }
        }
    }
}

constexpr int kDropRows = 4;             // rows per wave of the *_rows forms
constexpr int64_t kDropRowsMinN = 2048;  // below this the one-row-per-wave forms have more waves to hide latency with

}  // namespace

extern "C" {

size_t jmac_bn_tanh_workspace_bytes(int64_t N, int64_t d) {
    (void)N;
    return align_up((size_t)(kStatBlocks + 1) * 2 * (d > 0 ? d : 0) * 4) + 256;
}

int jmac_bn_tanh_fwd2_f32(const float* x, int64_t ldx, int64_t N, int64_t d, const float* weight, const float* bias,
                          float* running_mean, float* running_var, int32_t training, float momentum, float eps, float* y,
                          int64_t ldy, float* y2, int64_t ldy2, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                          jmac_stream_t stream) {
    if (N < 0 || !weight || !bias || !save_mean || !save_invstd) return JMAC_EINVAL;
    if (d <= 0 || d % 4 || ldx % 4 || ldy % 4 || (y2 && ldy2 % 4)) return JMAC_EDIM;
    if (N == 0) return JMAC_OK;
    if (!x || !y) return JMAC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int D4 = (int)(d / 4);
    if (training) {
        if (!ws || ws_bytes < jmac_bn_tanh_workspace_bytes(N, d)) return JMAC_EWORKSPACE;
        float* partial = (float*)ws;
        const unsigned g = stat_grid(N, D4);
        hipLaunchKernelGGL(col_stats_partial_kernel, dim3(g), dim3(kBlock), stat_smem(D4), st, x, ldx, N, D4, partial);
        hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3((unsigned)((d + RF_COLS - 1) / RF_COLS)), dim3(1024), 0, st, partial,
                           (int)g, x, N, (int)d, eps, momentum, running_mean, running_var, save_mean, save_invstd);
    } else {
        if (!running_mean || !running_var) return JMAC_EINVAL;
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, running_mean, running_var,
                           (int)d, eps, save_mean, save_invstd);
    }
    hipLaunchKernelGGL(bn_tanh_apply_kernel, dim3(stream_grid(N * D4)), dim3(kBlock), 0, st, x, ldx, N, D4, weight, bias,
                       save_mean, save_invstd, y, ldy, y2, ldy2);
    return (int)hipGetLastError();
}

int jmac_bn_tanh_fwd_f32(const float* x, int64_t ldx, int64_t N, int64_t d, const float* weight, const float* bias,
                         float* running_mean, float* running_var, int32_t training, float momentum, float eps, float* y,
                         int64_t ldy, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    return jmac_bn_tanh_fwd2_f32(x, ldx, N, d, weight, bias, running_mean, running_var, training, momentum, eps, y, ldy, nullptr, 0,
                                 save_mean, save_invstd, ws, ws_bytes, stream);
}

int jmac_bn_tanh_bwd_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy, int64_t N,
                         int64_t d, const float* weight, const float* save_mean, const float* save_invstd, int32_t training,
                         float* gx, int64_t ldgx, float* gweight, float* gbias, void* ws, size_t ws_bytes,
                         jmac_stream_t stream) {
    return jmac_bn_tanh_bwd2_f32(x, ldx, y, ldy, gy, ldgy, nullptr, 0, N, d, weight, save_mean, save_invstd, training, gx, ldgx,
                                 gweight, gbias, ws, ws_bytes, stream);
}

int jmac_bn_tanh_bwd2_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                          const float* gy2, int64_t ldgy2, int64_t N, int64_t d, const float* weight, const float* save_mean,
                          const float* save_invstd, int32_t training, float* gx, int64_t ldgx, float* gweight, float* gbias,
                          void* ws, size_t ws_bytes, jmac_stream_t stream) {
    if (N < 0 || !weight || !save_mean || !save_invstd || !gweight || !gbias) return JMAC_EINVAL;
    if (d <= 0 || d % 4 || ldx % 4 || ldy % 4 || ldgy % 4 || ldgx % 4 || (gy2 && ldgy2 % 4)) return JMAC_EDIM;
    if (!ws || ws_bytes < jmac_bn_tanh_workspace_bytes(N, d)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int D4 = (int)(d / 4);
    float* partial = (float*)ws;
    unsigned g = 0;
    if (N > 0) {
        if (!x || !y || !gy || !gx) return JMAC_EINVAL;
        g = stat_grid(N, D4);
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(g), dim3(kBlock), stat_smem(D4), st, x, ldx, y, ldy, gy, ldgy, gy2, ldgy2, N,
                           D4, save_mean, save_invstd, partial);
    }
    if (gweight == gbias + d) {                  // [gbias | gweight] contiguous: the reduction writes them directly
        launch_reduce_rows(partial, (int)g, (int)(2 * d), 1.f, gbias, st);
    } else {
        float* sums = partial + (size_t)kStatBlocks * 2 * d;
        launch_reduce_rows(partial, (int)g, (int)(2 * d), 1.f, sums, st);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, sums, (int)d, gweight, gbias);
    }
    if (N > 0)
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(N * D4)), dim3(kBlock), 0, st, x, ldx, y, ldy, gy, ldgy, gy2, ldgy2, N,
                           D4, weight, save_mean, save_invstd, gweight, gbias, training, 1.f / (float)N, gx, ldgx);
    return (int)hipGetLastError();
}

int jmac_row_normalize_fwd_f32(const float* x, int64_t ldx, int64_t N, int64_t d, float eps, float* y, int64_t ldy, float* inv,
                               jmac_stream_t stream) {
    if (N < 0 || d <= 0 || d >= INT32_MAX) return JMAC_EINVAL;
    if (N == 0) return JMAC_OK;
    if (!x || !y || !inv) return JMAC_EINVAL;
    int64_t blocks = (N + kBlock / 64 - 1) / (kBlock / 64);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(row_normalize_fwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, ldx, N, (int)d, eps,
                       y, ldy, inv);
    return (int)hipGetLastError();
}

int jmac_row_normalize_bwd_f32(const float* y, int64_t ldy, const float* g, int64_t ldg, const float* inv, int64_t N, int64_t d,
                               float eps, float* gx, int64_t ldgx, jmac_stream_t stream) {
    if (N < 0 || d <= 0 || d >= INT32_MAX) return JMAC_EINVAL;
    if (N == 0) return JMAC_OK;
    if (!y || !g || !inv || !gx) return JMAC_EINVAL;
    int64_t blocks = (N + kBlock / 64 - 1) / (kBlock / 64);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(row_normalize_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, y, ldy, g, ldg, inv, N,
                       (int)d, eps, gx, ldgx);
    return (int)hipGetLastError();
}

static int drop_fwd(const float* x, int64_t ldx, int64_t N, int64_t d, float eps, const DropSrc& ds, bool seeded, float* y, int64_t ldy,
                    float* inv, jmac_stream_t stream) {
    if (N < 0 || d <= 0 || d >= INT32_MAX) return JMAC_EINVAL;
    if (d % 4 || ldx % 4 || ldy % 4 || (ds.mask && ds.ldm % 4)) return JMAC_EDIM;
    if (N == 0) return JMAC_OK;
    if (!x || !y || !inv) return JMAC_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)ds.mask) & 15) != 0) return JMAC_EDIM;
    if (d <= 512 && N >= kDropRowsMinN) {                          // R rows per wave in flight
        int64_t rb = ((N + kDropRows - 1) / kDropRows + kBlock / 64 - 1) / (kBlock / 64);
        if (rb > 8192) rb = 8192;
        if (seeded)
            hipLaunchKernelGGL((row_normalize_drop_fwd_rows_kernel<true, kDropRows>), dim3((unsigned)rb), dim3(kBlock), 0,
                               (hipStream_t)stream, x, ldx, N, (int)(d / 4), eps, ds, y, ldy, inv);
        else
            hipLaunchKernelGGL((row_normalize_drop_fwd_rows_kernel<false, kDropRows>), dim3((unsigned)rb), dim3(kBlock), 0,
                               (hipStream_t)stream, x, ldx, N, (int)(d / 4), eps, ds, y, ldy, inv);
        return (int)hipGetLastError();
    }
    int64_t blocks = (N + kBlock / 64 - 1) / (kBlock / 64);
    if (blocks > 8192) blocks = 8192;
    if (seeded)
        hipLaunchKernelGGL(row_normalize_drop_fwd_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, ldx, N,
                           (int)(d / 4), eps, ds, y, ldy, inv);
    else
        hipLaunchKernelGGL(row_normalize_drop_fwd_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, ldx, N,
                           (int)(d / 4), eps, ds, y, ldy, inv);
    return (int)hipGetLastError();
}

static int drop_bwd(const float* x, int64_t ldx, const float* inv, const DropSrc& ds, bool seeded, const float* g, int64_t ldg, int64_t N,
                    int64_t d, float eps, float* gx, int64_t ldgx, int32_t accumulate, jmac_stream_t stream) {
    if (N < 0 || d <= 0 || d >= INT32_MAX) return JMAC_EINVAL;
    if (d % 4 || ldx % 4 || ldg % 4 || ldgx % 4 || (ds.mask && ds.ldm % 4)) return JMAC_EDIM;
    if (N == 0) return JMAC_OK;
    if (!x || !inv || !g || !gx) return JMAC_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)g | (uintptr_t)gx | (uintptr_t)ds.mask) & 15) != 0) return JMAC_EDIM;
    if (d <= 512 && N >= kDropRowsMinN) {
        int64_t rb = ((N + kDropRows - 1) / kDropRows + kBlock / 64 - 1) / (kBlock / 64);
        if (rb > 8192) rb = 8192;
        if (seeded)
            hipLaunchKernelGGL((row_normalize_drop_bwd_rows_kernel<true, kDropRows>), dim3((unsigned)rb), dim3(kBlock), 0,
                               (hipStream_t)stream, x, ldx, inv, ds, g, ldg, N, (int)(d / 4), eps, gx, ldgx, accumulate ? 1 : 0);
        else
            hipLaunchKernelGGL((row_normalize_drop_bwd_rows_kernel<false, kDropRows>), dim3((unsigned)rb), dim3(kBlock), 0,
                               (hipStream_t)stream, x, ldx, inv, ds, g, ldg, N, (int)(d / 4), eps, gx, ldgx, accumulate ? 1 : 0);
        return (int)hipGetLastError();
    }
    int64_t blocks = (N + kBlock / 64 - 1) / (kBlock / 64);
    if (blocks > 8192) blocks = 8192;
    if (seeded)
        hipLaunchKernelGGL(row_normalize_drop_bwd_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, ldx, inv,
                           ds, g, ldg, N, (int)(d / 4), eps, gx, ldgx, accumulate ? 1 : 0);
    else
        hipLaunchKernelGGL(row_normalize_drop_bwd_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, ldx, inv,
                           ds, g, ldg, N, (int)(d / 4), eps, gx, ldgx, accumulate ? 1 : 0);
    return (int)hipGetLastError();
}

// keep probability 1 - p as a 32-bit threshold; scale = 1 / (1 - p)
static int seeded_src(const int64_t* seed, float p_drop, DropSrc& ds) {
    if (!seed || !(p_drop >= 0.f) || !(p_drop < 1.f)) return JMAC_EINVAL;
    const double keep = 1.0 - (double)p_drop, t = keep * 4294967296.0;
    ds = DropSrc{nullptr, 0, seed, t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t, (float)(1.0 / keep)};
    return JMAC_OK;
}

int jmac_row_normalize_drop_fwd_f32(const float* x, int64_t ldx, int64_t N, int64_t d, float eps, const float* mask, int64_t ldm,
                                    float scale, float* y, int64_t ldy, float* inv, jmac_stream_t stream) {
    return drop_fwd(x, ldx, N, d, eps, DropSrc{mask, ldm, nullptr, 0u, scale}, false, y, ldy, inv, stream);
}

int jmac_row_normalize_drop_bwd_f32(const float* x, int64_t ldx, const float* inv, const float* mask, int64_t ldm, float scale,
                                    const float* g, int64_t ldg, int64_t N, int64_t d, float eps, float* gx, int64_t ldgx,
                                    int32_t accumulate, jmac_stream_t stream) {
    return drop_bwd(x, ldx, inv, DropSrc{mask, ldm, nullptr, 0u, scale}, false, g, ldg, N, d, eps, gx, ldgx, accumulate, stream);
}

int jmac_row_normalize_dropseed_fwd_f32(const float* x, int64_t ldx, int64_t N, int64_t d, float eps, const int64_t* seed,
                                        float p_drop, float* y, int64_t ldy, float* inv, jmac_stream_t stream) {
    DropSrc ds;
    if (int rc = seeded_src(seed, p_drop, ds)) return rc;
    return drop_fwd(x, ldx, N, d, eps, ds, true, y, ldy, inv, stream);
}

int jmac_row_normalize_dropseed_bwd_f32(const float* x, int64_t ldx, const float* inv, const int64_t* seed, float p_drop,
                                        const float* g, int64_t ldg, int64_t N, int64_t d, float eps, float* gx, int64_t ldgx,
                                        int32_t accumulate, jmac_stream_t stream) {
    DropSrc ds;
    if (int rc = seeded_src(seed, p_drop, ds)) return rc;
    return drop_bwd(x, ldx, inv, ds, true, g, ldg, N, d, eps, gx, ldgx, accumulate, stream);
}

// ---- segmented BatchNorm + tanh (block-batched encoder: several KGs in one launch set) ------------------------------
static int make_seg(int32_t nblocks, const int64_t* blk_ptr, const int32_t* order, int D4, SegTab& s) {
    if (nblocks < 1 || nblocks > kMaxSeg || !blk_ptr || blk_ptr[0] != 0) return JMAC_EINVAL;
    s.nb = nblocks;
    unsigned seen = 0;
    s.gofs[0] = 0;
    for (int b = 0; b < kMaxSeg; ++b) s.order[b] = 0;
    for (int b = 0; b < nblocks; ++b) {
        const int64_t n = blk_ptr[b + 1] - blk_ptr[b];
        if (n <= 0) return JMAC_EINVAL;
        s.ptr[b] = blk_ptr[b];
        s.gofs[b + 1] = s.gofs[b] + (int)stat_grid(n, D4);
        const int o = order ? order[b] : b;
        if (o < 0 || o >= nblocks || (seen >> o & 1u)) return JMAC_EINVAL;    // a permutation of the blocks
        seen |= 1u << o;
        s.order[b] = o;
    }
    for (int b = nblocks; b <= kMaxSeg; ++b) s.ptr[b] = blk_ptr[nblocks];
    for (int b = nblocks + 1; b <= kMaxSeg; ++b) s.gofs[b] = s.gofs[nblocks];
    return JMAC_OK;
}

size_t jmac_bn_tanh_seg_workspace_bytes(int32_t nblocks, int64_t d) {
    if (nblocks < 1) nblocks = 1;
    if (d < 0) d = 0;
    // partial rows of every block (<= kStatBlocks each) + the per-block backward sums [nb, 2d]
    return align_up((size_t)nblocks * kStatBlocks * 2 * d * 4) + align_up((size_t)nblocks * 2 * d * 4) + 256;
}

int jmac_bn_tanh_seg_fwd2_f32(const float* x, int64_t ldx, int64_t d, int32_t nblocks, const int64_t* blk_ptr,
                              const int32_t* order, const float* weight, const float* bias, float* running_mean,
                              float* running_var, float momentum, float eps, float* y, int64_t ldy, float* y2, int64_t ldy2,
                              float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    if (!x || !y || !weight || !bias || !save_mean || !save_invstd) return JMAC_EINVAL;
    if (d <= 0 || d % 4 || ldx % 4 || ldy % 4 || (y2 && ldy2 % 4)) return JMAC_EDIM;
    const int D4 = (int)(d / 4);
    SegTab seg;
    if (int rc = make_seg(nblocks, blk_ptr, order, D4, seg)) return rc;
    if (!ws || ws_bytes < jmac_bn_tanh_seg_workspace_bytes(nblocks, d)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)ws;
    hipLaunchKernelGGL(col_stats_partial_seg_kernel, dim3((unsigned)seg.gofs[nblocks]), dim3(kBlock), stat_smem(D4), st, x, ldx, seg,
                       D4, partial);
    hipLaunchKernelGGL(bn_reduce_finalize_seg_kernel, dim3((unsigned)((d + RF_COLS - 1) / RF_COLS)), dim3(1024), 0, st, partial, seg,
                       x, ldx, (int)d, eps, momentum, running_mean, running_var, save_mean, save_invstd);
    hipLaunchKernelGGL(bn_tanh_apply_seg_kernel, dim3(stream_grid(seg.ptr[nblocks] * D4)), dim3(kBlock), 0, st, x, ldx, seg, D4,
                       weight, bias, save_mean, save_invstd, y, ldy, y2, ldy2);
    return (int)hipGetLastError();
}

int jmac_bn_tanh_seg_bwd2_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                              const float* gy2, int64_t ldgy2, int64_t d, int32_t nblocks, const int64_t* blk_ptr,
                              const float* weight, const float* save_mean, const float* save_invstd, float* gx, int64_t ldgx,
                              float* gweight, float* gbias, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    if (!x || !y || !gy || !gx || !weight || !save_mean || !save_invstd || !gweight || !gbias) return JMAC_EINVAL;
    if (d <= 0 || d % 4 || ldx % 4 || ldy % 4 || ldgy % 4 || ldgx % 4 || (gy2 && ldgy2 % 4)) return JMAC_EDIM;
    const int D4 = (int)(d / 4);
    SegTab seg;
    if (int rc = make_seg(nblocks, blk_ptr, nullptr, D4, seg)) return rc;
    if (!ws || ws_bytes < jmac_bn_tanh_seg_workspace_bytes(nblocks, d)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)ws;
    float* bsums = (float*)((char*)ws + align_up((size_t)nblocks * kStatBlocks * 2 * d * 4));
    hipLaunchKernelGGL(bn_bwd_partial_seg_kernel, dim3((unsigned)seg.gofs[nblocks]), dim3(kBlock), stat_smem(D4), st, x, ldx, y, ldy,
                       gy, ldgy, gy2, ldgy2, seg, D4, save_mean, save_invstd, partial);
    hipLaunchKernelGGL(bn_bwd_reduce_seg_kernel, dim3((unsigned)((2 * d + RF_COLS - 1) / RF_COLS)), dim3(1024), 0, st, partial, seg,
                       (int)(2 * d), bsums, gbias, gweight, (int)d);
    hipLaunchKernelGGL(bn_bwd_apply_seg_kernel, dim3(stream_grid(seg.ptr[nblocks] * D4)), dim3(kBlock), 0, st, x, ldx, y, ldy, gy, ldgy,
                       gy2, ldgy2, seg, D4, weight, save_mean, save_invstd, bsums, gx, ldgx);
    return (int)hipGetLastError();
}

// ---- phased forms for batch statistics that span several ranks (destination-sharded layer, jmac_amd/dist.py) ---------
// moments -> [combine across ranks on the host side of the ABI] -> apply;  backward: sums -> [all-reduce] -> apply.

int jmac_col_moments_f32(const float* x, int64_t ldx, int64_t N, int64_t d, float* mean, float* m2, void* ws, size_t ws_bytes,
                         jmac_stream_t stream) {
    if (N <= 0 || !x || !mean || !m2) return JMAC_EINVAL;
    if (d <= 0 || d % 4 || ldx % 4) return JMAC_EDIM;
    if (!ws || ws_bytes < jmac_bn_tanh_workspace_bytes(N, d)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int D4 = (int)(d / 4);
    float* partial = (float*)ws;
    const unsigned g = stat_grid(N, D4);
    hipLaunchKernelGGL(col_stats_partial_kernel, dim3(g), dim3(kBlock), stat_smem(D4), st, x, ldx, N, D4, partial);
    float* sums = partial + (size_t)kStatBlocks * 2 * d;
    launch_reduce_rows(partial, (int)g, (int)(2 * d), 1.f, sums, st);
    hipLaunchKernelGGL(moments_finalize_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, sums, x, N, (int)d, mean, m2);
    return (int)hipGetLastError();
}

int jmac_bn_tanh_apply_f32(const float* x, int64_t ldx, int64_t N, int64_t d, const float* weight, const float* bias,
                           const float* mean, const float* invstd, float* y, int64_t ldy, jmac_stream_t stream) {
    if (N < 0 || !weight || !bias || !mean || !invstd) return JMAC_EINVAL;
    if (d <= 0 || d % 4 || ldx % 4 || ldy % 4) return JMAC_EDIM;
    if (N == 0) return JMAC_OK;
    if (!x || !y) return JMAC_EINVAL;
    const int D4 = (int)(d / 4);
    hipLaunchKernelGGL(bn_tanh_apply_kernel, dim3(stream_grid(N * D4)), dim3(kBlock), 0, (hipStream_t)stream, x, ldx, N, D4, weight,
                       bias, mean, invstd, y, ldy, (float*)nullptr, (int64_t)0);
    return (int)hipGetLastError();
}

int jmac_bn_tanh_bwd_sums_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy, int64_t N,
                              int64_t d, const float* mean, const float* invstd, float* sums, void* ws, size_t ws_bytes,
                              jmac_stream_t stream) {
    if (N < 0 || !mean || !invstd || !sums) return JMAC_EINVAL;
    if (d <= 0 || d % 4 || ldx % 4 || ldy % 4 || ldgy % 4) return JMAC_EDIM;
    if (!ws || ws_bytes < jmac_bn_tanh_workspace_bytes(N, d)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int D4 = (int)(d / 4);
    float* partial = (float*)ws;
    unsigned g = 0;
    if (N > 0) {
        if (!x || !y || !gy) return JMAC_EINVAL;
        g = stat_grid(N, D4);
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(g), dim3(kBlock), stat_smem(D4), st, x, ldx, y, ldy, gy, ldgy,
                           (const float*)nullptr, (int64_t)0, N, D4, mean, invstd, partial);
    }
    launch_reduce_rows(partial, (int)g, (int)(2 * d), 1.f, sums, st);     // sums[0:d] = sum gz, sums[d:2d] = sum gz * xhat
    return (int)hipGetLastError();
}

int jmac_bn_tanh_bwd_apply_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy, int64_t N,
                               int64_t d, const float* weight, const float* mean, const float* invstd, const float* sums,
                               int64_t n_total, float* gx, int64_t ldgx, jmac_stream_t stream) {
    if (N < 0 || n_total <= 0 || !weight || !mean || !invstd || !sums) return JMAC_EINVAL;
    if (d <= 0 || d % 4 || ldx % 4 || ldy % 4 || ldgy % 4 || ldgx % 4) return JMAC_EDIM;
    if (N == 0) return JMAC_OK;
    if (!x || !y || !gy || !gx) return JMAC_EINVAL;
    const int D4 = (int)(d / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(N * D4)), dim3(kBlock), 0, (hipStream_t)stream, x, ldx, y, ldy, gy, ldgy,
                       (const float*)nullptr, (int64_t)0, N, D4, weight, mean, invstd, sums + d, sums, 1, 1.f / (float)n_total, gx,
                       ldgx);
    return (int)hipGetLastError();
}

}  // extern "C"
