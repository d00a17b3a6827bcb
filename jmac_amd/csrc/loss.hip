// loss.hip -- fused gathers of the completion / alignment losses for gfx950.
//
//  * jmac_triple_l1_{fwd,bwd}_f32    score_t = || E[h_t] + R[r_t] - E[t_t] ||_1      src/jmac_model.py:345-350
//  * jmac_pair_cosine_{fwd,bwd}_f32  dist_p  = 1 - cos(E1[i_p], E2[j_p])             src/jmac_model.py:245-247,271-291
//
// The reference materialises three (two) gathered [T,d] copies, the sum, the norm and -- in the backward --
// three (two) index_add passes per call.  Here one wavefront owns a triple (pair): the rows are read once with
// 16 B per lane, reduced across the wave, and the backward re-reads them, rebuilds the sign (the normalised
// rows) in registers and adds straight into the gradient tables with hardware float atomics
// (global_atomic_add_f32; same accumulation-order freedom as torch's index_add_ that it replaces).
// Indices are the reference's int64 tensors, used as they are.  HBM/L2-bound: 3 (2) row reads per unit.
#include "common.h"

using namespace jmac;

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// ---- triple L1 ---------------------------------------------------------------------------------------
// Lane layout: element c = lane + 64k (a wave instruction covers 256 contiguous bytes: the shape float atomics
// run at full rate with; 16 B per lane would spread one instruction over sixteen 64-B atomic requests).
//
// `period` is a HINT about the batch layout of the reference (train.py:347-352: sub.repeat(K+1), rel.repeat(K+1),
// cat(obj, negatives)): triples x, x+period, x+2*period ... are expected to share (h, r).  One wave takes such a
// run: the E[h] + R[r] row is read once, and in the backward the run's contributions to dE[h] / dR[r] are summed
// in registers and added once (28 atomic rows per positive instead of 78, and K+1 times fewer adds into the few
// hot relation rows).  A triple of the run whose (h, r) differs is handled on its own: any input is correct.
template <int NK>
__global__ __launch_bounds__(kBlock) void triple_l1_fwd_kernel(const float* __restrict__ ent, int64_t lde,
                                                               const float* __restrict__ rel, int64_t ldr,
                                                               const int64_t* __restrict__ h, const int64_t* __restrict__ r,
                                                               const int64_t* __restrict__ t, int64_t T, int64_t period, int parts,
                                                               int d, float* __restrict__ score) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t nrun = (period < T ? period : T) * parts;
    const int64_t stride = period * parts;
    for (int64_t u = w0; u < nrun; u += nw) {     // unit = (run b, part p): triples b + (p + parts*j) * period
        const int64_t b = u / parts, p = u % parts;
        int64_t ch = -1, cr = -1;
        float hr[NK];
        // two triples of the run per iteration: the index loads and the tail rows of both are in flight together
        // (a wave's run is only a few triples long: without this every triple is its own index -> row -> reduce chain)
        int64_t x = b + p * period;
        while (x < T) {
            const int64_t x2 = x + stride;
            const bool in2 = x2 < T;
            const int64_t ih = h[x], ir = r[x], it = t[x];
            const int64_t ih2 = h[in2 ? x2 : x], ir2 = r[in2 ? x2 : x], it2 = t[in2 ? x2 : x];
            const bool pair = in2 && ih2 == ih && ir2 == ir;          // wave-uniform
            if (ih != ch || ir != cr) {
                ch = ih;
                cr = ir;
                const float* ph = ent + ih * lde;
                const float* pr = rel + ir * ldr;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int c = lane + 64 * k;
                    hr[k] = c < d ? ph[c] + pr[c] : 0.f;
                }
            }
            const float* pt = ent + it * lde;
            const float* pt2 = ent + (pair ? it2 : it) * lde;
            float tv[NK], tv2[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int c = lane + 64 * k;
                tv[k] = c < d ? pt[c] : 0.f;
                tv2[k] = c < d ? pt2[c] : 0.f;
            }
            float acc[2] = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int c = lane + 64 * k;
                if (c < d) {
                    acc[0] += fabsf(hr[k] - tv[k]);
                    acc[1] += fabsf(hr[k] - tv2[k]);
                }
            }
            wave_sum_n<2>(acc);
            if (lane == 0) {
                score[x] = acc[0];
                if (pair) score[x2] = acc[1];
            }
            x += pair ? 2 * stride : stride;
        }
    }
}

// MARGIN: the score gradient is not read but derived on the spot from the margin ranking loss that consumed the scores
// (completion_loss, src/jmac_model.py:351-378: dscore[b] = c sum_k w_{b,k}, dscore[B + kB + b] = -c w_{b,k}, c = gloss/(BK),
// w = 1 (pos - neg > -gamma), 1/2 (tie), 0) -- `gscore` then holds the SCORES, and no dscore vector / launch exists.
struct MarginArgs {
    const float* gamma;
    const float* gloss;
    int64_t B, K;
};
// EXACT (MARGIN only): every contribution of the margin ranking loss is an integer multiple of u = gloss / (2 B K) (the weights w
// are 0, 1/2 or 1), so the kernel adds the INTEGERS G_x * sgn(.) with G_x = dscore_x / u -- sums of integers below 2^24 are exact
// in fp32 whatever order the memory-side atomic units see them in -- and a scaling pass multiplies the finished tables by u:
// bitwise reproducible gradients with the same atomics (launcher: jmac_triple_l1_margin_bwd_f32 whenever 4 B K < 2^24).
template <int NK, bool MARGIN, bool EXACT = false>
__global__ __launch_bounds__(kBlock) void triple_l1_bwd_kernel(const float* __restrict__ ent, int64_t lde,
                                                               const float* __restrict__ rel, int64_t ldr,
                                                               const int64_t* __restrict__ h, const int64_t* __restrict__ r,
                                                               const int64_t* __restrict__ t, int64_t T, int64_t period, int parts,
                                                               int d, const float* __restrict__ gscore, float* __restrict__ dent,
                                                               int64_t ldde, float* __restrict__ drel, int64_t lddr, MarginArgs ma) {
    float mg_c = 0.f, mg_gamma = 0.f;
    if (MARGIN) {
        mg_gamma = ma.gamma[0];
        mg_c = EXACT ? 2.f : ma.gloss[0] / ((float)ma.B * (float)ma.K);      // EXACT: units of u (2 w is 0, 1 or 2)
    }
    auto gof = [&](int64_t x) -> float {                       // x is wave-uniform
        if (!MARGIN) return gscore[x];
        auto w = [&](float diff) { return diff > -mg_gamma ? 1.f : (diff == -mg_gamma ? 0.5f : 0.f); };
        if (x >= ma.B) return -mg_c * w(gscore[(x - ma.B) % ma.B] - gscore[x]);
        const float pos = gscore[x];
        float sum = 0.f;
        int64_t k = 0;
        for (; k + 8 <= ma.K; k += 8) {                        // eight negatives in flight per trip (the loop is a latency chain
            float nv[8];                                       // otherwise); same summation order as margin_loss_bwd_kernel
#pragma unroll
            for (int u = 0; u < 8; ++u) nv[u] = gscore[ma.B + (k + u) * ma.B + x];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += w(pos - nv[u]);
        }
        for (; k < ma.K; ++k) sum += w(pos - gscore[ma.B + k * ma.B + x]);
        return mg_c * sum;
    };
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t nrun = (period < T ? period : T) * parts;
    const int64_t stride = period * parts;
    for (int64_t u = w0; u < nrun; u += nw) {
        const int64_t b = u / parts, p = u % parts;
        int64_t ch = -1, cr = -1;
        float hr[NK], acc[NK];
        auto flush = [&]() {
            if (ch < 0) return;
            float* qh = dent + ch * ldde;
            float* qr = drel + cr * lddr;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int c = lane + 64 * k;
                if (c < d && acc[k] != 0.f) {
                    atomicAdd(qh + c, acc[k]);
                    atomicAdd(qr + c, acc[k]);
                }
            }
        };
        int64_t x = b + p * period;
        while (x < T) {
            const int64_t x2 = x + stride;
            const bool in2 = x2 < T;
            const int64_t ih = h[x], ir = r[x], it = t[x];
            const int64_t ih2 = h[in2 ? x2 : x], ir2 = r[in2 ? x2 : x], it2 = t[in2 ? x2 : x];
            const bool pair = in2 && ih2 == ih && ir2 == ir;          // wave-uniform
            if (ih != ch || ir != cr) {
                flush();
                ch = ih;
                cr = ir;
                const float* ph = ent + ih * lde;
                const float* pr = rel + ir * ldr;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int c = lane + 64 * k;
                    hr[k] = c < d ? ph[c] + pr[c] : 0.f;
                    acc[k] = 0.f;
                }
            }
            const float g = gof(x), g2 = pair ? gof(x2) : 0.f;
            const float* pt = ent + it * lde;
            const float* pt2 = ent + (pair ? it2 : it) * lde;
            float* qt = dent + it * ldde;
            float* qt2 = dent + (pair ? it2 : it) * ldde;
            float tv[NK], tv2[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int c = lane + 64 * k;
                tv[k] = c < d ? pt[c] : 0.f;
                tv2[k] = c < d ? pt2[c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int c = lane + 64 * k;
                if (c < d) {
                    const float v = g * sgn(hr[k] - tv[k]);
                    const float v2 = g2 * sgn(hr[k] - tv2[k]);
                    acc[k] += v + v2;
                    if (v != 0.f) atomicAdd(qt + c, -v);
                    if (v2 != 0.f) atomicAdd(qt2 + c, -v2);
                }
            }
            x += pair ? 2 * stride : stride;
        }
        flush();
    }
}

// ---- pair cosine distance ----------------------------------------------------------------------------
// F.normalize(x, 2, -1) = x / max(||x||, eps), eps = 1e-12 (torch default, src/jmac_model.py:245-246)
constexpr float kNormEps = 1e-12f;

template <bool VEC>
__device__ __forceinline__ void pair_dots(const float* __restrict__ pa, const float* __restrict__ pb, int d, int lane,
                                          float& ab, float& aa, float& bb) {
    float s_ab = 0.f, s_aa = 0.f, s_bb = 0.f;
    if (VEC) {
        const int D4 = d >> 2;
#pragma unroll 2
        for (int c = lane; c < D4; c += 64) {
            const float4 a = ld4(pa + 4 * c), b = ld4(pb + 4 * c);
            s_ab += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            s_aa += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
            s_bb += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        }
    } else {
        for (int c = lane; c < d; c += 64) {
            const float a = pa[c], b = pb[c];
            s_ab += a * b;
            s_aa += a * a;
            s_bb += b * b;
        }
    }
    float v[3] = {s_ab, s_aa, s_bb};
    wave_sum_n<3>(v);
    ab = v[0];
    aa = v[1];
    bb = v[2];
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void pair_cosine_fwd_kernel(const float* __restrict__ e1, int64_t ld1,
                                                                 const float* __restrict__ e2, int64_t ld2,
                                                                 const int64_t* __restrict__ i1, const int64_t* __restrict__ i2,
                                                                 int64_t L, int d, float* __restrict__ dist, float* __restrict__ stats) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t x = w0; x < L; x += nw) {
        float ab, aa, bb;
        pair_dots<VEC>(e1 + i1[x] * ld1, e2 + i2[x] * ld2, d, lane, ab, aa, bb);
        const float na = fmaxf(sqrtf(aa), kNormEps), nb = fmaxf(sqrtf(bb), kNormEps);
        if (lane == 0) {
            dist[x] = 1.f - ab / (na * nb);
            if (stats) *reinterpret_cast<float4*>(stats + 4 * x) = make_float4(ab, aa, bb, 0.f);   // for the sorted backward
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void pair_cosine_bwd_kernel(const float* __restrict__ e1, int64_t ld1,
                                                                 const float* __restrict__ e2, int64_t ld2,
                                                                 const int64_t* __restrict__ i1, const int64_t* __restrict__ i2,
                                                                 int64_t L, int d, const float* __restrict__ gdist,
                                                                 float* __restrict__ de1, int64_t ldd1, float* __restrict__ de2,
                                                                 int64_t ldd2) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t x = w0; x < L; x += nw) {
        const int64_t ia = i1[x], ib = i2[x];
        const float* pa = e1 + ia * ld1;
        const float* pb = e2 + ib * ld2;
        float ab, aa, bb;
        pair_dots<VEC>(pa, pb, d, lane, ab, aa, bb);
        const float ra = sqrtf(aa), rb = sqrtf(bb);
        const float na = fmaxf(ra, kNormEps), nb = fmaxf(rb, kNormEps);
        // dist = 1 - (a.b)/(na nb);  with u = a/na, v = b/nb:  d dist/d a = -(v - (u.v) u)/na  (0 below eps)
        const float g = -gdist[x];
        const float inv = 1.f / (na * nb);
        const float c = ab * inv;
        const float ka = ra > kNormEps ? c / (na * na) : 0.f;   // clamped norm is constant: no radial term
        const float kb = rb > kNormEps ? c / (nb * nb) : 0.f;
        float* qa = de1 + ia * ldd1;
        float* qb = de2 + ib * ldd2;
        // scalar lane layout for the atomics (256 contiguous bytes per wave instruction)
        for (int cidx = lane; cidx < d; cidx += 64) {
            const float a = pa[cidx], b = pb[cidx];
            atomicAdd(qa + cidx, g * (b * inv - ka * a));
            atomicAdd(qb + cidx, g * (a * inv - kb * b));
        }
    }
}

// EXACT margin backward, second pass: the integer tables times u = gloss / (2 B K) (one launch for both tables)
__global__ __launch_bounds__(kBlock) void scale_rows2_kernel(float* __restrict__ a, int64_t lda, int64_t rows_a, float* __restrict__ b,
                                                             int64_t ldb, int64_t rows_b, int d, const float* __restrict__ gloss,
                                                             float inv_2bk) {
    const float u = gloss[0] * inv_2bk;
    const int64_t na = rows_a * d, total = na + rows_b * d;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        float* p = i < na ? a + (i / d) * lda + (i % d) : b + ((i - na) / d) * ldb + ((i - na) % d);
        *p *= u;
    }
}
// the same for dense tables (ld == d, 16-byte aligned, d % 4 == 0): one flat float4 stream per table
__global__ __launch_bounds__(kBlock) void scale_flat2_kernel(float4* __restrict__ a, int64_t na4, float4* __restrict__ b, int64_t nb4,
                                                             const float* __restrict__ gloss, float inv_2bk) {
    const float u = gloss[0] * inv_2bk;
    const int64_t total = na4 + nb4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        float4* p = i < na4 ? a + i : b + (i - na4);
        float4 v = *p;
        v.x *= u; v.y *= u; v.z *= u; v.w *= u;
        *p = v;
    }
}

// ---- pair cosine, deterministic backward -------------------------------------------------------------------------------------
// The index vectors of an alignment loss are constant over many steps (the seed links of a KG pair: every batch of an epoch,
// train.py:347-352; the mined negatives: until the next refresh), so the host sorts the 2 L (pair, side) incidences by the
// gradient row they touch once per index tensor (like the graph's CSR).  One wave per sorted position; the wave at the HEAD of a
// run of equal rows sums the run's contributions in sorted order (ties in pair order: the sort is stable) and writes the row
// with a plain store -- no atomics, bitwise reproducible; rows no pair touches stay at the caller's zero fill.  The forward
// leaves (a.b, a.a, b.b) per pair, so an incidence re-reads its partner row only.
// rec [2L] int4 per SORTED position = {pair x, own table row, partner table row, flags | gradient row}: flags bit 31 = head of its
// run, bit 30 = the incidence is side 1 (own row in e2, partner in e1), bit 29 = the gradient row lives in de2.  One 16-byte
// record per position instead of key -> order -> index -> row (three dependent round trips before the first row read).
constexpr unsigned kRecHead = 1u << 31, kRecSide = 1u << 30, kRecDe2 = 1u << 29, kRecRow = (1u << 29) - 1u;
__global__ __launch_bounds__(kBlock) void pair_cosine_bwd_sorted_kernel(const float* __restrict__ e1, int64_t ld1,
                                                                        const float* __restrict__ e2, int64_t ld2, int64_t n_inc, int d,
                                                                        const float* __restrict__ gdist, const float* __restrict__ stats,
                                                                        const int4* __restrict__ rec, float* __restrict__ de1,
                                                                        int64_t ldd1, float* __restrict__ de2, int64_t ldd2) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t p = w0; p < n_inc; p += nw) {
        int4 rc = rec[p];
        if (!((unsigned)rc.w & kRecHead)) continue;                    // not the head of its run (wave-uniform)
        const float* own = ((unsigned)rc.w & kRecSide) ? e2 + (int64_t)rc.y * ld2 : e1 + (int64_t)rc.y * ld1;
        float* dst = ((unsigned)rc.w & kRecDe2) ? de2 + (int64_t)((unsigned)rc.w & kRecRow) * ldd2
                                                : de1 + (int64_t)((unsigned)rc.w & kRecRow) * ldd1;
        constexpr int NKM = 8;                                         // d <= 512
        float acc[NKM], ow[NKM];
#pragma unroll
        for (int k = 0; k < NKM; ++k) {
            const int c = lane + 64 * k;
            acc[k] = 0.f;
            ow[k] = c < d ? own[c] : 0.f;
        }
        for (int64_t q = p;;) {
            const int64_t qn = q + 1;
            const int4 nx = rec[qn < n_inc ? qn : q];                  // the next record is in flight while this one is summed
            const bool side = ((unsigned)rc.w & kRecSide) != 0;
            const float* partner = side ? e1 + (int64_t)rc.z * ld1 : e2 + (int64_t)rc.z * ld2;
            const float4 st = *reinterpret_cast<const float4*>(stats + 4 * (int64_t)rc.x);
            const float g = -gdist[rc.x];
            float pr[NKM];
#pragma unroll
            for (int k = 0; k < NKM; ++k) {
                const int c = lane + 64 * k;
                pr[k] = c < d ? partner[c] : 0.f;
            }
            const float ra = sqrtf(st.y), rb = sqrtf(st.z);
            const float na = fmaxf(ra, kNormEps), nb = fmaxf(rb, kNormEps);
            const float inv = 1.f / (na * nb);
            const float c_ = st.x * inv;
            // side 0 (row a of pair x): g (b inv - ka a);  side 1 (row b): g (a inv - kb b)
            const float kself = side ? (rb > kNormEps ? c_ / (nb * nb) : 0.f) : (ra > kNormEps ? c_ / (na * na) : 0.f);
#pragma unroll
            for (int k = 0; k < NKM; ++k) acc[k] += g * (pr[k] * inv - kself * ow[k]);
            if (qn >= n_inc || ((unsigned)nx.w & kRecHead)) break;
            q = qn;
            rc = nx;
        }
#pragma unroll
        for (int k = 0; k < NKM; ++k) {
            const int c = lane + 64 * k;
            if (c < d) dst[c] = acc[k];
        }
    }
}

// The same reduction ROW by ROW (first writer of the whole gradient table: no zero fill by the caller): one wave per gradient row
// over [0, n1 + n2) -- rows [0, n1) live in de1, rows [n1, n1 + n2) in de2 (n2 = 0 when both sides share one table) --, rowptr
// [n1 + n2 + 1] = first sorted position of every row's run.  A row no pair touches is written as zeros.  The incoming gradient is a
// vector gdist [L] or (gscalar != nullptr) the scalar gscalar[0] / gscale for every pair: the mean over the pairs, no expand / div
// launch (alignment_loss_simple's .mean(), src/jmac_model.py:249).
__global__ __launch_bounds__(kBlock) void pair_cosine_bwd_rows_kernel(const float* __restrict__ e1, int64_t ld1,
                                                                      const float* __restrict__ e2, int64_t ld2, int d,
                                                                      const float* __restrict__ gdist, const float* __restrict__ gscalar,
                                                                      float gscale, const float* __restrict__ stats,
                                                                      const int4* __restrict__ rec, const int32_t* __restrict__ rowptr,
                                                                      int64_t n1, int64_t n2, float* __restrict__ de1, int64_t ldd1,
                                                                      float* __restrict__ de2, int64_t ldd2) {
    const int lane = lane_id();
    const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * kWavesPerBlock;
    const float gs = gscalar ? gscalar[0] / gscale : 0.f;          // (a division, like the mean's own backward: the same bits)
    constexpr int NKM = 8;                                             // d <= 512
    for (int64_t row = w0; row < n1 + n2; row += nw) {
        const int p0 = rowptr[row], p1 = rowptr[row + 1];
        float* dst = row < n1 ? de1 + row * ldd1 : de2 + (row - n1) * ldd2;
        float acc[NKM];
#pragma unroll
        for (int k = 0; k < NKM; ++k) acc[k] = 0.f;
        if (p1 > p0) {                                                 // wave-uniform
            int4 rc = rec[p0];
            const float* own = ((unsigned)rc.w & kRecSide) ? e2 + (int64_t)rc.y * ld2 : e1 + (int64_t)rc.y * ld1;
            float ow[NKM];
#pragma unroll
            for (int k = 0; k < NKM; ++k) {
                const int c = lane + 64 * k;
                ow[k] = c < d ? own[c] : 0.f;
            }
            for (int q = p0; q < p1; ++q) {                            // the run in sorted order: the sums of the sorted form, bit for bit
                const int4 nx = rec[q + 1 < p1 ? q + 1 : q];
                const bool side = ((unsigned)rc.w & kRecSide) != 0;
                const float* partner = side ? e1 + (int64_t)rc.z * ld1 : e2 + (int64_t)rc.z * ld2;
                const float4 st = *reinterpret_cast<const float4*>(stats + 4 * (int64_t)rc.x);
                const float g = -(gscalar ? gs : gdist[rc.x]);
                float pr[NKM];
#pragma unroll
                for (int k = 0; k < NKM; ++k) {
                    const int c = lane + 64 * k;
                    pr[k] = c < d ? partner[c] : 0.f;
                }
                const float ra = sqrtf(st.y), rb = sqrtf(st.z);
                const float na = fmaxf(ra, kNormEps), nb = fmaxf(rb, kNormEps);
                const float inv = 1.f / (na * nb);
                const float c_ = st.x * inv;
                const float kself = side ? (rb > kNormEps ? c_ / (nb * nb) : 0.f) : (ra > kNormEps ? c_ / (na * na) : 0.f);
#pragma unroll
                for (int k = 0; k < NKM; ++k) acc[k] += g * (pr[k] * inv - kself * ow[k]);
                rc = nx;
            }
        }
#pragma unroll
        for (int k = 0; k < NKM; ++k) {
            const int c = lane + 64 * k;
            if (c < d) dst[c] = acc[k];
        }
    }
}

// EXACT margin backward on PERSISTENT count tables (jmac_triple_l1_margin_bwd_exact2_f32): out (+)= cnt * u, cnt = 0.  The count
// tables start at zero and every call leaves them at zero again, so no caller ever zero-fills a gradient table for the atomics:
// the pass that scales the integers anyway also writes the output (first writer, or on top of what `accumulate` says is there)
__global__ __launch_bounds__(kBlock) void scale_clear_flat2_kernel(float4* __restrict__ cnt_a, float4* __restrict__ out_a, int64_t na4,
                                                                   float4* __restrict__ cnt_b, float4* __restrict__ out_b, int64_t nb4,
                                                                   const float* __restrict__ gloss, float inv_2bk, int acc_a, int acc_b) {
    const float u = gloss[0] * inv_2bk;
    const int64_t total = na4 + nb4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const bool in_a = i < na4;
        float4* pc = in_a ? cnt_a + i : cnt_b + (i - na4);
        float4* po = in_a ? out_a + i : out_b + (i - na4);
        const float4 c = *pc;
        float4 v = make_float4(c.x * u, c.y * u, c.z * u, c.w * u);
        if (in_a ? acc_a : acc_b) {                 // two roundings (no fma contraction): the bits of "scale, then add the tables"
            const float4 o = *po;
            v.x = __fadd_rn(__fmul_rn(c.x, u), o.x); v.y = __fadd_rn(__fmul_rn(c.y, u), o.y);
            v.z = __fadd_rn(__fmul_rn(c.z, u), o.z); v.w = __fadd_rn(__fmul_rn(c.w, u), o.w);
        }
        *po = v;
        if (c.x != 0.f || c.y != 0.f || c.z != 0.f || c.w != 0.f) *pc = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

inline unsigned wave_grid(int64_t units) {
    int64_t b = (units + kWavesPerBlock - 1) / kWavesPerBlock;
    if (b < 1) b = 1;
    return (unsigned)(b < 8192 ? b : 8192);
}

// a run of T/period triples is cut into `parts` interleaved sub-runs so that at least ~8k wavefronts exist
// (1000 runs of 26 left three quarters of the chip idle and each wave on a 26-deep dependent chain)
inline int run_parts(int64_t T, int64_t period) {
    const int64_t len = (T + period - 1) / period;
    int64_t p = (8192 + period - 1) / period;
    if (p > len) p = len;
    if (p < 1) p = 1;
    return (int)p;
}

inline bool vec_ok(int64_t d, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
    if (d % 4) return false;
    for (int64_t l : lds)
        if (l % 4) return false;
    for (const void* p : ptrs)
        if ((uintptr_t)p % 16) return false;
    return true;
}

// ---- margin ranking loss of completion_loss (src/jmac_model.py:351-378) --------------------------------------------------
//   pos = score[:B].view(-1, B).permute(1, 0)  [B,1];  neg = score[B:].view(-1, B).permute(1, 0)  [B,K]  (n-major, as consumed)
//   loss = mean_{b,k} max(pos_b - neg_{b,k}, -gamma) + gamma          with neg_{b,k} = score[B + k*B + b]
// The reference spends ~10 element-wise launches each way on 26 000 scores; here one block each way.  The sum is reduced
// in a fixed order (bitwise reproducible).  torch.max(a, b) hands a tie half of the gradient: kept.
constexpr int kMarginBlock = 1024;
__global__ __launch_bounds__(kMarginBlock) void margin_loss_fwd_kernel(const float* __restrict__ score, int B, int K,
                                                                        const float* __restrict__ gamma_p, const float* __restrict__ add_to,
                                                                        float* __restrict__ loss) {
    __shared__ float red[kMarginBlock];
    const float gamma = gamma_p[0];
    float acc = 0.f;
    const int total = B * K;                  // (launcher: B K < 2^31) 32-bit index arithmetic: a 64-bit modulo is ~40 instructions
    // eight independent (positive, negative) pairs in flight per trip: a rolled loop pays the load latency once per element
    // (25 dependent round trips per thread at the reference's batch: 11 us for 26 000 scores)
    int i = threadIdx.x;
    for (; i + 7 * kMarginBlock < total; i += 8 * kMarginBlock) {
        float p[8], n[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = i + u * kMarginBlock;
            p[u] = score[(unsigned)j % (unsigned)B];
            n[u] = score[B + j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += fmaxf(p[u] - n[u], -gamma);      // fixed order per thread: reproducible
    }
    for (; i < total; i += kMarginBlock) acc += fmaxf(score[(unsigned)i % (unsigned)B] - score[B + i], -gamma);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kMarginBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (red[0] / (float)total + gamma) + (add_to ? add_to[0] : 0.f);   // add_to: the running loss of the step
}

// out[0] = mean(x[0 .. n)) (+ add_to[0]): the mean over the pairs of alignment_loss_simple (src/jmac_model.py:249) and the sum
// with the step's running loss in one launch, fixed summation order (one block; n ~ 10^3)
__global__ __launch_bounds__(kMarginBlock) void vec_mean_acc_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ add_to,
                                                                     float* __restrict__ out) {
    __shared__ float red[kMarginBlock];
    float acc = 0.f;
    int64_t i = threadIdx.x;
    for (; i + 3 * kMarginBlock < n; i += 4 * kMarginBlock) {
        const float a = x[i], b = x[i + kMarginBlock], c = x[i + 2 * kMarginBlock], e = x[i + 3 * kMarginBlock];
        acc += a; acc += b; acc += c; acc += e;
    }
    for (; i < n; i += kMarginBlock) acc += x[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kMarginBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (float)n + (add_to ? add_to[0] : 0.f);
}

// dscore[b] = g/(BK) sum_k w_{b,k};  dscore[B + k*B + b] = -g/(BK) w_{b,k};  w = 1 (diff > -gamma), 1/2 (tie), 0
__global__ __launch_bounds__(kBlock) void margin_loss_bwd_kernel(const float* __restrict__ score, int B, int K,
                                                                 const float* __restrict__ gamma_p, const float* __restrict__ gloss,
                                                                 float* __restrict__ dscore) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const float gamma = gamma_p[0], c = gloss[0] / ((float)B * (float)K), pos = score[b];
    float sum = 0.f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {                               // eight negatives in flight per trip
        float nv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) nv[u] = score[(int64_t)B + (int64_t)(k + u) * B + b];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float diff = pos - nv[u];
            const float w = diff > -gamma ? 1.f : (diff == -gamma ? 0.5f : 0.f);
            dscore[(int64_t)B + (int64_t)(k + u) * B + b] = -c * w;
            sum += w;
        }
    }
    for (; k < K; ++k) {
        const int64_t j = (int64_t)B + (int64_t)k * B + b;
        const float diff = pos - score[j];
        const float w = diff > -gamma ? 1.f : (diff == -gamma ? 0.5f : 0.f);
        dscore[j] = -c * w;
        sum += w;
    }
    dscore[b] = c * sum;
}

}  // namespace

extern "C" {

#define JMAC_DISPATCH_NK(nk, ...)                          \
    switch (nk) {                                          \
        case 1: { constexpr int NK = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int NK = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int NK = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int NK = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int NK = 5; __VA_ARGS__; } break; \
        case 6: { constexpr int NK = 6; __VA_ARGS__; } break; \
        case 7: { constexpr int NK = 7; __VA_ARGS__; } break; \
        default: { constexpr int NK = 8; __VA_ARGS__; } break; \
    }

int jmac_triple_l1_fwd_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr, const int64_t* h, const int64_t* r,
                           const int64_t* t, int64_t T, int64_t period, int64_t d, float* score, jmac_stream_t stream) {
    if (T < 0 || d <= 0) return JMAC_EINVAL;
    if (d > 512) return JMAC_EDIM;
    if (T == 0) return JMAC_OK;
    if (!ent || !rel || !h || !r || !t || !score) return JMAC_EINVAL;
    if (period <= 0 || period > T) period = T;
    hipStream_t st = (hipStream_t)stream;
    const int nk = (int)((d + 63) / 64);
    const int parts = run_parts(T, period);
    JMAC_DISPATCH_NK(nk, hipLaunchKernelGGL((triple_l1_fwd_kernel<NK>), dim3(wave_grid(period * parts)), dim3(kBlock), 0, st, ent, lde,
                                            rel, ldr, h, r, t, T, period, parts, (int)d, score));
    return (int)hipGetLastError();
}

int jmac_triple_l1_bwd_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr, const int64_t* h, const int64_t* r,
                           const int64_t* t, int64_t T, int64_t period, int64_t d, const float* gscore, float* dent,
                           int64_t ldde, float* drel, int64_t lddr, jmac_stream_t stream) {
    if (T < 0 || d <= 0) return JMAC_EINVAL;
    if (d > 512) return JMAC_EDIM;
    if (T == 0) return JMAC_OK;
    if (!ent || !rel || !h || !r || !t || !gscore || !dent || !drel) return JMAC_EINVAL;
    if (period <= 0 || period > T) period = T;
    hipStream_t st = (hipStream_t)stream;
    const int nk = (int)((d + 63) / 64);
    const int parts = run_parts(T, period);
    JMAC_DISPATCH_NK(nk, hipLaunchKernelGGL((triple_l1_bwd_kernel<NK, false>), dim3(wave_grid(period * parts)), dim3(kBlock), 0, st,
                                            ent, lde, rel, ldr, h, r, t, T, period, parts, (int)d, gscore, dent, ldde, drel, lddr,
                                            MarginArgs{}));
    return (int)hipGetLastError();
}

int jmac_triple_l1_margin_bwd_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr, const int64_t* h, const int64_t* r,
                                  const int64_t* t, int64_t B, int64_t K, int64_t d, const float* score, const float* gamma,
                                  const float* gloss, float* dent, int64_t ldde, float* drel, int64_t lddr, jmac_stream_t stream) {
    if (B <= 0 || K <= 0 || d <= 0 || B >= INT32_MAX || K >= INT32_MAX) return JMAC_EINVAL;
    if (d > 512) return JMAC_EDIM;
    if (!ent || !rel || !h || !r || !t || !score || !gamma || !gloss || !dent || !drel) return JMAC_EINVAL;
    const int64_t T = B * (K + 1), period = B;
    hipStream_t st = (hipStream_t)stream;
    const int nk = (int)((d + 63) / 64);
    const int parts = run_parts(T, period);
    JMAC_DISPATCH_NK(nk, hipLaunchKernelGGL((triple_l1_bwd_kernel<NK, true>), dim3(wave_grid(period * parts)), dim3(kBlock), 0, st,
                                            ent, lde, rel, ldr, h, r, t, T, period, parts, (int)d, score, dent, ldde, drel, lddr,
                                            MarginArgs{gamma, gloss, B, K}));
    return (int)hipGetLastError();
}

int jmac_triple_l1_margin_bwd_exact_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr, const int64_t* h, const int64_t* r,
                                        const int64_t* t, int64_t B, int64_t K, int64_t d, const float* score, const float* gamma,
                                        const float* gloss, float* dent, int64_t ldde, int64_t n_ent, float* drel, int64_t lddr,
                                        int64_t n_rel, jmac_stream_t stream) {
    if (B <= 0 || K <= 0 || d <= 0 || B >= INT32_MAX || K >= INT32_MAX || n_ent < 0 || n_rel < 0) return JMAC_EINVAL;
    if (d > 512) return JMAC_EDIM;
    if (!ent || !rel || !h || !r || !t || !score || !gamma || !gloss || !dent || !drel) return JMAC_EINVAL;
    if (4 * B * K >= (1LL << 24))          // a row's integer sum could leave fp32's exact range: the plain form
        return jmac_triple_l1_margin_bwd_f32(ent, lde, rel, ldr, h, r, t, B, K, d, score, gamma, gloss, dent, ldde, drel, lddr, stream);
    const int64_t T = B * (K + 1), period = B;
    hipStream_t st = (hipStream_t)stream;
    const int nk = (int)((d + 63) / 64);
    const int parts = run_parts(T, period);
    JMAC_DISPATCH_NK(nk, hipLaunchKernelGGL((triple_l1_bwd_kernel<NK, true, true>), dim3(wave_grid(period * parts)), dim3(kBlock), 0, st,
                                            ent, lde, rel, ldr, h, r, t, T, period, parts, (int)d, score, dent, ldde, drel, lddr,
                                            MarginArgs{gamma, gloss, B, K}));
    const int64_t total = (n_ent + n_rel) * d;
    if (total > 0) {
        const float inv_2bk = 1.f / (2.f * (float)B * (float)K);
        if (ldde == d && lddr == d && d % 4 == 0 && ((((uintptr_t)dent | (uintptr_t)drel) & 15) == 0)) {
            int64_t blocks = (total / 4 + kBlock - 1) / kBlock;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(scale_flat2_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, reinterpret_cast<float4*>(dent),
                               n_ent * d / 4, reinterpret_cast<float4*>(drel), n_rel * d / 4, gloss, inv_2bk);
        } else {
            int64_t blocks = (total + kBlock - 1) / kBlock;
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(scale_rows2_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, dent, ldde, n_ent, drel, lddr, n_rel,
                               (int)d, gloss, inv_2bk);
        }
    }
    return (int)hipGetLastError();
}

int jmac_pair_cosine_fwd_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2, const int64_t* i1, const int64_t* i2,
                             int64_t L, int64_t d, float* dist, jmac_stream_t stream) {
    if (L < 0 || d <= 0 || d >= INT32_MAX) return JMAC_EINVAL;
    if (L == 0) return JMAC_OK;
    if (!e1 || !e2 || !i1 || !i2 || !dist) return JMAC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (vec_ok(d, {ld1, ld2}, {e1, e2}))
        hipLaunchKernelGGL(pair_cosine_fwd_kernel<true>, dim3(wave_grid(L)), dim3(kBlock), 0, st, e1, ld1, e2, ld2, i1, i2, L, (int)d, dist,
                           (float*)nullptr);
    else
        hipLaunchKernelGGL(pair_cosine_fwd_kernel<false>, dim3(wave_grid(L)), dim3(kBlock), 0, st, e1, ld1, e2, ld2, i1, i2, L, (int)d, dist,
                           (float*)nullptr);
    return (int)hipGetLastError();
}

int jmac_pair_cosine_fwd_stats_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2, const int64_t* i1, const int64_t* i2,
                                   int64_t L, int64_t d, float* dist, float* stats, jmac_stream_t stream) {
    if (L < 0 || d <= 0 || d >= INT32_MAX) return JMAC_EINVAL;
    if (L == 0) return JMAC_OK;
    if (!e1 || !e2 || !i1 || !i2 || !dist || !stats || ((uintptr_t)stats & 15)) return JMAC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (vec_ok(d, {ld1, ld2}, {e1, e2}))
        hipLaunchKernelGGL(pair_cosine_fwd_kernel<true>, dim3(wave_grid(L)), dim3(kBlock), 0, st, e1, ld1, e2, ld2, i1, i2, L, (int)d, dist, stats);
    else
        hipLaunchKernelGGL(pair_cosine_fwd_kernel<false>, dim3(wave_grid(L)), dim3(kBlock), 0, st, e1, ld1, e2, ld2, i1, i2, L, (int)d, dist, stats);
    return (int)hipGetLastError();
}

int jmac_pair_cosine_bwd_sorted_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2, int64_t L, int64_t d,
                                    const float* gdist, const float* stats, const int32_t* rec, float* de1, int64_t ldd1,
                                    float* de2, int64_t ldd2, jmac_stream_t stream) {
    if (L < 0 || d <= 0) return JMAC_EINVAL;
    if (d > 512) return JMAC_EDIM;
    if (L == 0) return JMAC_OK;
    if (2 * L >= INT32_MAX) return JMAC_ERANGE;
    if (!e1 || !e2 || !gdist || !stats || !rec || !de1 || !de2 || ((uintptr_t)rec & 15)) return JMAC_EINVAL;
    hipLaunchKernelGGL(pair_cosine_bwd_sorted_kernel, dim3(wave_grid(2 * L)), dim3(kBlock), 0, (hipStream_t)stream, e1, ld1, e2, ld2,
                       2 * L, (int)d, gdist, stats, reinterpret_cast<const int4*>(rec), de1, ldd1, de2, ldd2);
    return (int)hipGetLastError();
}

int jmac_pair_cosine_bwd_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2, const int64_t* i1, const int64_t* i2,
                             int64_t L, int64_t d, const float* gdist, float* de1, int64_t ldd1, float* de2, int64_t ldd2,
                             jmac_stream_t stream) {
    if (L < 0 || d <= 0 || d >= INT32_MAX) return JMAC_EINVAL;
    if (L == 0) return JMAC_OK;
    if (!e1 || !e2 || !i1 || !i2 || !gdist || !de1 || !de2) return JMAC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (vec_ok(d, {ld1, ld2}, {e1, e2}))
        hipLaunchKernelGGL(pair_cosine_bwd_kernel<true>, dim3(wave_grid(L)), dim3(kBlock), 0, st, e1, ld1, e2, ld2, i1, i2, L, (int)d,
                           gdist, de1, ldd1, de2, ldd2);
    else
        hipLaunchKernelGGL(pair_cosine_bwd_kernel<false>, dim3(wave_grid(L)), dim3(kBlock), 0, st, e1, ld1, e2, ld2, i1, i2, L, (int)d,
                           gdist, de1, ldd1, de2, ldd2);
    return (int)hipGetLastError();
}

int jmac_margin_loss_fwd_acc_f32(const float* score, int64_t B, int64_t K, const float* gamma, const float* add_to, float* loss,
                                 jmac_stream_t stream) {
    if (B <= 0 || K <= 0 || B >= INT32_MAX || K >= INT32_MAX || B * K + B >= INT32_MAX) return JMAC_EINVAL;
    if (!score || !gamma || !loss) return JMAC_EINVAL;
    hipLaunchKernelGGL(margin_loss_fwd_kernel, dim3(1), dim3(kMarginBlock), 0, (hipStream_t)stream, score, (int)B, (int)K, gamma, add_to,
                       loss);
    return (int)hipGetLastError();
}

int jmac_margin_loss_fwd_f32(const float* score, int64_t B, int64_t K, const float* gamma, float* loss, jmac_stream_t stream) {
    return jmac_margin_loss_fwd_acc_f32(score, B, K, gamma, nullptr, loss, stream);
}

int jmac_vec_mean_acc_f32(const float* x, int64_t n, const float* add_to, float* out, jmac_stream_t stream) {
    if (n <= 0 || !x || !out) return JMAC_EINVAL;
    hipLaunchKernelGGL(vec_mean_acc_kernel, dim3(1), dim3(kMarginBlock), 0, (hipStream_t)stream, x, n, add_to, out);
    return (int)hipGetLastError();
}

int jmac_pair_cosine_bwd_rows_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2, int64_t L, int64_t d, const float* gdist,
                                  const float* gscalar, float gscale, const float* stats, const int32_t* rec, const int32_t* rowptr,
                                  int64_t n1, int64_t n2, float* de1, int64_t ldd1, float* de2, int64_t ldd2, jmac_stream_t stream) {
    if (L < 0 || d <= 0 || n1 < 0 || n2 < 0) return JMAC_EINVAL;
    if (d > 512) return JMAC_EDIM;
    if (n1 + n2 == 0) return JMAC_OK;
    if (2 * L >= INT32_MAX) return JMAC_ERANGE;
    if (!e1 || !e2 || (!gdist && !gscalar) || !rowptr || !de1 || (n2 > 0 && !de2)) return JMAC_EINVAL;
    if (L > 0 && (!stats || !rec || ((uintptr_t)rec & 15) || ((uintptr_t)stats & 15))) return JMAC_EINVAL;
    hipLaunchKernelGGL(pair_cosine_bwd_rows_kernel, dim3(wave_grid(n1 + n2)), dim3(kBlock), 0, (hipStream_t)stream, e1, ld1, e2, ld2, (int)d,
                       gdist, gscalar, gscale, stats, reinterpret_cast<const int4*>(rec), rowptr, n1, n2, de1, ldd1, de2 ? de2 : de1, ldd2);
    return (int)hipGetLastError();
}

int jmac_triple_l1_margin_bwd_exact2_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr, const int64_t* h, const int64_t* r,
                                         const int64_t* t, int64_t B, int64_t K, int64_t d, const float* score, const float* gamma,
                                         const float* gloss, int64_t ent_off, int64_t rel_off, float* cnt_ent, float* cnt_rel,
                                         float* dent, int64_t rows_ent, int32_t acc_ent, float* drel, int64_t rows_rel, int32_t acc_rel,
                                         jmac_stream_t stream) {
    if (B <= 0 || K <= 0 || d <= 0 || B >= INT32_MAX || K >= INT32_MAX || rows_ent <= 0 || rows_rel <= 0 || ent_off < 0 || rel_off < 0 ||
        ent_off >= rows_ent || rel_off >= rows_rel)
        return JMAC_EINVAL;
    if (d > 512 || d % 4) return JMAC_EDIM;
    if (!ent || !rel || !h || !r || !t || !score || !gamma || !gloss || !cnt_ent || !cnt_rel || !dent || !drel) return JMAC_EINVAL;
    if ((((uintptr_t)cnt_ent | (uintptr_t)cnt_rel | (uintptr_t)dent | (uintptr_t)drel) & 15) != 0) return JMAC_EINVAL;
    if (4 * B * K >= (1LL << 24)) return JMAC_ERANGE;       // a row's integer sum could leave fp32's exact range (callers: the plain form)
    const int64_t T = B * (K + 1), period = B;
    hipStream_t st = (hipStream_t)stream;
    const int nk = (int)((d + 63) / 64);
    const int parts = run_parts(T, period);
    // the integer contributions go into the count tables (dense, pitch d) at the window the ids are local to
    JMAC_DISPATCH_NK(nk, hipLaunchKernelGGL((triple_l1_bwd_kernel<NK, true, true>), dim3(wave_grid(period * parts)), dim3(kBlock), 0, st,
                                            ent, lde, rel, ldr, h, r, t, T, period, parts, (int)d, score, cnt_ent + ent_off * d, d,
                                            cnt_rel + rel_off * d, d, MarginArgs{gamma, gloss, B, K}));
    const int64_t na4 = rows_ent * d / 4, nb4 = rows_rel * d / 4;
    int64_t blocks = (na4 + nb4 + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_clear_flat2_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, reinterpret_cast<float4*>(cnt_ent),
                       reinterpret_cast<float4*>(dent), na4, reinterpret_cast<float4*>(cnt_rel), reinterpret_cast<float4*>(drel), nb4, gloss,
                       1.f / (2.f * (float)B * (float)K), (int)acc_ent, (int)acc_rel);
    return (int)hipGetLastError();
}

int jmac_margin_loss_bwd_f32(const float* score, int64_t B, int64_t K, const float* gamma, const float* gloss, float* dscore,
                             jmac_stream_t stream) {
    if (B <= 0 || K <= 0 || B >= INT32_MAX || K >= INT32_MAX) return JMAC_EINVAL;
    if (!score || !gamma || !gloss || !dscore) return JMAC_EINVAL;
    hipLaunchKernelGGL(margin_loss_bwd_kernel, dim3((unsigned)((B + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, score,
                       (int)B, (int)K, gamma, gloss, dscore);
    return (int)hipGetLastError();
}

}  // extern "C"
