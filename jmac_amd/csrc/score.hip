// score.hip -- triple and entity-pair scoring for gfx950.
//
//  * jmac_l1_score_f32      torch.cdist(er, all_kg_emb, p=1)            src/jmac_model.py:312
//  * jmac_filtered_rank_f32 filter + sort + np.where ranking loop        src/validate.py:50-64
//  * jmac_sim_matrix_f32    torch.mm(ILL_vec, KG_vec.t())                modules/utils/util.py:52, train.py:239
//  * jmac_row_topk_f32 / jmac_sim_topk_f32   sim.topk(k, dim=1)          modules/utils/util.py:53
//  * jmac_softmax_entropy_f32, jmac_masked_row_softmax_f32               train.py:241-257
//
// L1 distance is |a-b| accumulation: not a contraction, so it runs on the VALU (register-tiled through
// LDS); the similarity matrices are contractions and run on the matrix cores with the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate), so top-k indices are decided on
// full-precision scores.
#include "common.h"
#include <type_traits>

using namespace jmac;

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// L1 score: 64x64 output tile per block, 4x4 per thread, K staged 16 at a time (transposed in LDS)
// ------------------------------------------------------------------------------------------------
constexpr int L1_T = 64, L1_K = 16, L1_LD = L1_T + 4;

// acc + |a - b| in two full-rate VALU ops (v_sub_f32, v_add_f32 with the |.| source modifier).  Left to the
// compiler, fabsf() becomes v_and_b32 and the adds are SLP-packed into half-rate v_pk_add_f32: 3 issue slots.
// (Round 6, tools/r6_l1_probe.py: with -fno-slp-vectorize the compiler's own form keeps the |.| modifier and has none of the
// hazard s_nops hipcc puts between dependent asm statements -- one per sub/add pair here -- but it hoists a slab's 32 fragment
// reads: 165 VGPRs against 56, three waves per SIMD, and the tile kernels run 8-60 % SLOWER; at eight waves per SIMD the nops
// cost nothing, another wave issues in their slot.)
__device__ __forceinline__ float add_absdiff(float acc, float a, float b) {
    float d, r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    asm("v_add_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(acc), "v"(d));
    return r;
}

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((uint32_t)(*p) << 16); }

// TT = float or bf16_t (raw bits): bf16 operands are widened when they are staged into LDS, the |a-b| accumulation
// is fp32 either way (the kernel is VALU-bound, so the narrower tables change the bytes, not the time)
template <typename TT, bool VEC>
__global__ __launch_bounds__(kBlock) void l1_score_kernel(const TT* __restrict__ er, int64_t lder,
                                                          const TT* __restrict__ tab, int64_t ldt, int B, int N, int d,
                                                          float* __restrict__ out, int64_t ldout, int accumulate) {
    __shared__ __attribute__((aligned(16))) float As[2][L1_K][L1_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][L1_K][L1_LD];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int b0 = blockIdx.y * L1_T, n0 = blockIdx.x * L1_T;
    // loader mapping: one float4 (4 consecutive k) of one row per thread and per operand
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    const int64_t arow = b0 + lrow, brow = n0 + lrow;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    // Every global load is UNCONDITIONAL at a clamped address: a load behind an exec-mask branch gets its s_waitcnt
    // inside the branch (hipcc), i.e. the next slab's prefetch would be waited for before the current slab's arithmetic
    // instead of after it.  Rows past the end re-read the last row (they only feed outputs that are never stored); k
    // quads past d re-read the last quad and are zeroed when they are written to LDS.  (d % 4 != 0: scalar tail form.)
    constexpr bool vec = VEC;                    // d % 4 == 0 (the launcher picks the instantiation)
    auto gload = [&](const TT* base, int64_t ld, int64_t row, int64_t nrows, int k0) -> float4 {
        const int k = k0 + lk;
        if constexpr (vec) {
            const int64_t r = row < nrows ? row : nrows - 1;
            return cvt4(ldraw(base + r * ld + (k < d ? k : d - 4)));
        }
        float4 v = f4zero();
        if (row < nrows) {
            if (k + 3 < d) v = cvt4(ldraw(base + row * ld + k));
            else {
                const TT* p = base + row * ld;
                if (k + 0 < d) v.x = ld1(p + k + 0);
                if (k + 1 < d) v.y = ld1(p + k + 1);
                if (k + 2 < d) v.z = ld1(p + k + 2);
            }
        }
        return v;
    };
    auto sstore = [&](float (*S)[L1_LD], float4 v, int k0, bool row_ok) {
        if (vec && !(row_ok && k0 + lk < d)) v = f4zero();          // the clamped loads' padding: |0 - 0| adds nothing
        S[lk + 0][lrow] = v.x;
        S[lk + 1][lrow] = v.y;
        S[lk + 2][lrow] = v.z;
        S[lk + 3][lrow] = v.w;
    };
    const int nk = (d + L1_K - 1) / L1_K;
    float4 ra = gload(er, lder, arow, B, 0), rb = gload(tab, ldt, brow, N, 0);
    sstore(As[0], ra, 0, arow < B);
    sstore(Bs[0], rb, 0, brow < N);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            ra = gload(er, lder, arow, B, (kt + 1) * L1_K);
            rb = gload(tab, ldt, brow, N, (kt + 1) * L1_K);
        }
#pragma unroll
        for (int k = 0; k < L1_K; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(&As[cur][k][ty * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&Bs[cur][k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = add_absdiff(acc[i][j], av[i], bv[j]);
        }
        if (kt + 1 < nk) {
            sstore(As[cur ^ 1], ra, (kt + 1) * L1_K, arow < B);
            sstore(Bs[cur ^ 1], rb, (kt + 1) * L1_K, brow < N);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t b = b0 + ty * 4 + i;
        if (b >= B) continue;
        const int64_t n = n0 + tx * 4;
        float* o = out + b * ldout + n;
        if (n + 3 < N && (ldout % 4 == 0)) {
            float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            if (accumulate) {
                const float4 p = ld4(o);
                v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
            }
            st4(o, v);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < N) o[j] = accumulate ? o[j] + acc[i][j] : acc[i][j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// filtered rank: one block per query
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void filtered_rank_kernel(const float* __restrict__ score, int64_t lds,
                                                               const int32_t* __restrict__ gold,
                                                               const int32_t* __restrict__ filt_ptr,
                                                               const int32_t* __restrict__ filt_idx, int N, int descending,
                                                               int32_t* __restrict__ rank) {
    __shared__ int red[kBlock / 64];
    const int b = blockIdx.x;
    const float* row = score + (int64_t)b * lds;
    const int g = gold[b];
    const float gs = row[g];
    // ascending: a DISTANCE (smaller ranks first); descending: a SIMILARITY (larger first) -- no negated copy needed
    auto before = [&](float s, int n) { return (descending ? s > gs : s < gs) || (s == gs && n < g); };
    int cnt = 0;
    for (int n = threadIdx.x; n < N; n += kBlock) cnt += before(row[n], n) ? 1 : 0;
    if (filt_ptr) {
        for (int f = filt_ptr[b] + threadIdx.x; f < filt_ptr[b + 1]; f += kBlock) {
            const int n = filt_idx[f];
            if (n == g || n < 0 || n >= N) continue;
            cnt -= before(row[n], n) ? 1 : 0;
        }
    }
    cnt = wave_sum_i(cnt);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += red[w];
        rank[b] = t + 1;
    }
}

// ------------------------------------------------------------------------------------------------
// fused link prediction: ranks of the gold tails WITHOUT the [B, N] distance matrix
//   dist[b, n] = sum over layers l, then k, of |(E_l[h_b] +/- R_l[r_b])[k] - T_l[n, k]|   (ONE running fp32 sum, in that order)
//   rank[b]    = 1 + #{n : dist[b, n] before dist[b, gold_b]} - #{filtered n != gold : ... before ...}
// "before" = smaller, or equal with the lower index (jmac_filtered_rank_f32).  Two launches: the prep kernel builds the
// query rows, the gold distances and the filter correction; the tile kernel is the L1 score kernel with a layer loop around
// its slab loop and a compare-and-count epilogue (integer atomics: order-independent) instead of the [B, N] store.
// ------------------------------------------------------------------------------------------------
constexpr int LR_MAX_LAYERS = 4;
struct LinkRankArgs {
    const float* ent[LR_MAX_LAYERS];      // [*, d] fp32 entity tables the query rows are gathered from
    const float* rel[LR_MAX_LAYERS];      // [*, d] fp32 relation tables
    const void* tab[LR_MAX_LAYERS];       // [N, d] candidate tables of element type TT (== ent for fp32)
    int64_t ld_ent[LR_MAX_LAYERS], ld_rel[LR_MAX_LAYERS], ld_tab[LR_MAX_LAYERS];
    int32_t nl, B, N, d, dq;              // dq = d rounded up to 4: row stride of the query workspace
    int32_t rows;                         // candidate rows the prep kernel stages per pass (LDS budget)
    float sign;                           // +1: E[h] + R[r] (tail prediction), -1: E[h] - R[r]
    const int32_t *h, *r, *gold, *filt_ptr, *filt_idx;
    void* er;                             // workspace [nl][B][dq] of TT
    float* gs;                            // workspace [B]
    int32_t* rank;                        // [B]: the counters themselves
};

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = f32_to_bf16_rne(v); }

// one block per query: (1) its rows er_l = E_l[h] +/- R_l[r] (rounded to TT) -> workspace and LDS; (2) the listed candidates
// (the gold first, then the filter entries), LR_ROWS at a time: their table rows are staged into LDS with coalesced loads, then
// one lane per candidate runs the SAME sequential sum the tile kernel runs (bit-identical distances: exact ties resolve by
// index as in the materialised path); (3) rank[b] starts at 1 - #{filtered entries before the gold}
constexpr int LR_ROWS = 16;                        // at most; fewer when nl * d is large (60 KB of LDS)
template <typename TT>
__global__ __launch_bounds__(kBlock) void link_rank_prep_kernel(LinkRankArgs a) {
    extern __shared__ float lr_sh[];                // [nl*dq] query rows (widened) + [LR_ROWS][nl*dq + 1] candidate rows
    __shared__ float gs_sh;
    __shared__ int cand[LR_ROWS];
    __shared__ int red[kBlock / 64];
    const int b = blockIdx.x, d = a.d, dq = a.dq, W = a.nl * dq, WS = W + 1;
    float* const erow = lr_sh;
    float* const crow = lr_sh + W;
    const int hb = a.h[b], rb = a.r[b], g = a.gold[b];
    TT* const er = static_cast<TT*>(a.er);
    for (int l = 0; l < a.nl; ++l)
        for (int k = threadIdx.x; k < dq; k += kBlock) {
            float v = 0.f;
            if (k < d) v = a.ent[l][(int64_t)hb * a.ld_ent[l] + k] + a.sign * a.rel[l][(int64_t)rb * a.ld_rel[l] + k];
            TT* dst = er + ((int64_t)l * a.B + b) * dq + k;
            st1(dst, v);
            erow[l * dq + k] = ld1(dst);            // what the tile kernel will read back (bf16: rounded)
        }
    const int f0 = a.filt_ptr ? a.filt_ptr[b] : 0, f1 = a.filt_ptr ? a.filt_ptr[b + 1] : 0;
    const int n_list = 1 + (f1 - f0);               // entry 0 = the gold
    float gs = 0.f;
    int cnt = 0;
    for (int c0 = 0; c0 < n_list; c0 += a.rows) {
        const int nc = min(a.rows, n_list - c0);
        __syncthreads();                            // previous chunk consumed; erow complete
        if (threadIdx.x < nc) {
            const int c = c0 + threadIdx.x;
            int n = c == 0 ? g : a.filt_idx[f0 + c - 1];
            if (c > 0 && (n == g || n < 0 || n >= a.N)) n = -1;      // skipped entries (as jmac_filtered_rank_f32)
            cand[threadIdx.x] = n;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nc * W; i += kBlock) {
            const int rI = i / W, q = i - rI * W, l = q / dq, k = q - l * dq;
            const int n = cand[rI];
            float v = 0.f;
            if (n >= 0 && k < d) v = ld1(static_cast<const TT*>(a.tab[l]) + (int64_t)n * a.ld_tab[l] + k);
            crow[rI * WS + q] = v;
        }
        __syncthreads();
        if (threadIdx.x < nc) {
            const float* row = crow + threadIdx.x * WS;
            float acc = 0.f;
            for (int l = 0; l < a.nl; ++l) {
                int k = 0;
                for (; k + 4 <= d; k += 4) {
                    const float e0 = erow[l * dq + k], e1 = erow[l * dq + k + 1], e2 = erow[l * dq + k + 2], e3 = erow[l * dq + k + 3];
                    const float t0 = row[l * dq + k], t1 = row[l * dq + k + 1], t2 = row[l * dq + k + 2], t3 = row[l * dq + k + 3];
                    acc = add_absdiff(acc, e0, t0);
                    acc = add_absdiff(acc, e1, t1);
                    acc = add_absdiff(acc, e2, t2);
                    acc = add_absdiff(acc, e3, t3);
                }
                for (; k < d; ++k) acc = add_absdiff(acc, erow[l * dq + k], row[l * dq + k]);
            }
            if (c0 == 0 && threadIdx.x == 0) gs_sh = acc;
            crow[threadIdx.x * WS] = acc;           // parked for the compare below (row data no longer needed)
        }
        __syncthreads();
        gs = gs_sh;
        if (threadIdx.x < nc && !(c0 == 0 && threadIdx.x == 0)) {
            const int n = cand[threadIdx.x];
            const float sc = crow[threadIdx.x * WS];
            if (n >= 0) cnt += (sc < gs || (sc == gs && n < g)) ? 1 : 0;
        }
    }
    cnt = wave_sum_i(cnt);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += red[w];
        a.gs[b] = gs;
        a.rank[b] = 1 - t;
    }
}

// the L1 score tile kernel over the concatenated (layer, k) axis; epilogue: count the entries that rank before the gold
template <typename TT, bool VEC>
__global__ __launch_bounds__(kBlock) void link_rank_tile_kernel(LinkRankArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][L1_K][L1_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][L1_K][L1_LD];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int B = a.B, N = a.N, d = a.d;
    const int b0 = blockIdx.y * L1_T, n0 = blockIdx.x * L1_T;
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    const int64_t arow = b0 + lrow, brow = n0 + lrow;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    constexpr bool vec = VEC;
    // unconditional clamped loads, zeroed when staged (see l1_score_kernel)
    auto gload = [&](const TT* base, int64_t ld, int64_t row, int64_t nrows, int k0) -> float4 {
        const int k = k0 + lk;
        if constexpr (vec) {
            const int64_t r = row < nrows ? row : nrows - 1;
            return cvt4(ldraw(base + r * ld + (k < d ? k : d - 4)));
        }
        float4 v = f4zero();
        if (row < nrows) {
            const TT* p = base + row * ld;
            if (k + 0 < d) v.x = ld1(p + k + 0);
            if (k + 1 < d) v.y = ld1(p + k + 1);
            if (k + 2 < d) v.z = ld1(p + k + 2);
            if (k + 3 < d) v.w = ld1(p + k + 3);
        }
        return v;
    };
    auto sstore = [&](float (*S)[L1_LD], float4 v, int k0, bool row_ok) {
        if (vec && !(row_ok && k0 + lk < d)) v = f4zero();
        S[lk + 0][lrow] = v.x;
        S[lk + 1][lrow] = v.y;
        S[lk + 2][lrow] = v.z;
        S[lk + 3][lrow] = v.w;
    };
    const int nk = (d + L1_K - 1) / L1_K, ns = a.nl * nk;      // slabs: layer-major, k inside
    const TT* const er = static_cast<const TT*>(a.er);
    auto slab = [&](int s, float4& ra, float4& rb) {
        const int l = s / nk, k0 = (s - l * nk) * L1_K;
        ra = gload(er + (int64_t)l * B * a.dq, a.dq, arow, B, k0);
        rb = gload(static_cast<const TT*>(a.tab[l]), a.ld_tab[l], brow, N, k0);
    };
    float4 ra, rb;
    slab(0, ra, rb);
    sstore(As[0], ra, 0, arow < B);
    sstore(Bs[0], rb, 0, brow < N);
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        const int cur = s & 1;
        if (s + 1 < ns) slab(s + 1, ra, rb);
#pragma unroll
        for (int k = 0; k < L1_K; ++k) {
            const float4 av4 = *reinterpret_cast<const float4*>(&As[cur][k][ty * 4]);
            const float4 bv4 = *reinterpret_cast<const float4*>(&Bs[cur][k][tx * 4]);
            const float av[4] = {av4.x, av4.y, av4.z, av4.w};
            const float bv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = add_absdiff(acc[i][j], av[i], bv[j]);
        }
        if (s + 1 < ns) {
            const int l1 = (s + 1) / nk, k1 = (s + 1 - l1 * nk) * L1_K;
            sstore(As[cur ^ 1], ra, k1, arow < B);
            sstore(Bs[cur ^ 1], rb, k1, brow < N);
        }
        __syncthreads();
    }
    // count: 4 rows x 4 columns per thread; the 16 threads of a row (tx = 0..15: consecutive lanes) are summed with DPP
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = b0 + ty * 4 + i;
        const bool row_ok = b < B;
        const float gs = row_ok ? a.gs[b] : 0.f;
        const int g = row_ok ? a.gold[b] : 0;
        int c = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            const float sc = acc[i][j];
            c += (row_ok && n < N && (sc < gs || (sc == gs && n < g))) ? 1 : 0;
        }
        c += __shfl_xor(c, 1);
        c += __shfl_xor(c, 2);
        c += __shfl_xor(c, 4);
        c += __shfl_xor(c, 8);
        if (tx == 0 && row_ok && c) atomicAdd(a.rank + b, c);
    }
}

// ------------------------------------------------------------------------------------------------
// similarity GEMM  C = A * B^T  on the fp32-input MFMA (32x32x2), 128x128 or 128x256 tile per 4-wave block, K staged 16 deep
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Tile: 128 rows x (64 WJ) columns per 4-wave block, wave grid 2 x 2, a wave owns 64 x (32 WJ).  WJ = 2: the 128 x 128 tile of
// rounds 1-5 (64 accumulator registers, 3 blocks per CU); WJ = 4: 128 x 256 (128 accumulator registers, 2 blocks per CU): per flop
// a quarter fewer operand bytes through L2 and LDS and half as many barriers (round 6: +3.5 % where its rounds of resident blocks are
// no coarser, profiles/r6_simgemm_ablation.txt).  launch_sim picks per shape.  The contraction order per output
// element is the same for both (planes k = 8q + 4h + s), so the results are bit-identical whichever runs.
constexpr int SG_T = 128, SG_K = 16;
constexpr int SG_PLANE = SG_T * 4 + 8;      // floats per (sub-slab q, k-half h) plane; +8: the 4 planes a wave store
                                            // instruction touches start on different banks
constexpr int SG_IMG = 4 * SG_PLANE;        // one operand slab: planes (q, h) = (0,0) (0,1) (1,0) (1,1)
template <int WJ>
struct SgTile {
    static constexpr int TN = 64 * WJ;              // tile columns
    static constexpr int PLANE_B = TN * 4 + 8;      // B operand's plane
    static constexpr int IMG_B = 4 * PLANE_B;
    static constexpr int LB = TN / 64;              // float4s per thread of one B slab (A: 2)
    static constexpr int SUPER_N = 8 * 2 / WJ;      // super-tile of 1024 x 1024 outputs: 8 x 8 tiles of 128 x 128, or 8 x 4 of 128 x 256
    static constexpr int OCC = WJ == 2 ? 3 : 2;     // resident blocks per CU the register budget is set for
};
constexpr int SG_SUPER = 8;                 // super-tile of 1024 x 1024 outputs per XCD visit (L2: 2 x 1.2 MB of operand rows at d=300)
constexpr int SG_FL_CAP = 256;              // FILTER epilogue: passing elements a wave collects before it claims their slots
// Ablation builds (tools/r6_simgemm_ablation.sh; the product is built with 0): where does the matrix pipe's idle time go?
//   bit 0  the C tile is not stored (a store behind a never-true data-dependent test keeps the accumulators live)
//   bit 1  no global loads after a block's first slab (the staging registers are re-used: LDS traffic and barriers stay)
//   bit 2  no LDS stores, fragment reads or barriers inside the K loop: the MFMA chain alone (implies nothing about results)
#ifndef JMAC_SG_ABLATE
#define JMAC_SG_ABLATE 0
#endif
constexpr bool SG_NO_STORE = (JMAC_SG_ABLATE & 1) != 0, SG_NO_LOAD = (JMAC_SG_ABLATE & 2) != 0, SG_MFMA_ONLY = (JMAC_SG_ABLATE & 4) != 0;

// FILTER = true: the scores are not stored.  An element that reaches its row's threshold tau (a lower bound of the row's k-th
// largest score, taken from a column sample: jmac_sim_topk_f32) is appended to the row's candidate list instead -- the running
// top-k of SURVEY K8: the L x N matrix never exists.
struct SimFilter {
    const float* tau;        // tau[m * tau_stride]
    int64_t tau_stride;
    int* cnt;                // [M] candidates seen per row (may exceed cap: the row then recomputes its scores, see below)
    float* cval;             // [M, cap]
    int* cidx;               // [M, cap]
    int cap;
    int n_off;               // the product covers columns [n_off, n_off + N) of the full matrix: candidate index = n_off + n
};

template <bool FILTER, int WJ>
__global__ __launch_bounds__(kBlock, SgTile<WJ>::OCC) void sim_gemm_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm,
                                                          int64_t ldb, int M, int N, int d, float* __restrict__ C, int64_t ldc,
                                                          int tiles_m, int tiles_n, int super_order, int n_ids, SimFilter flt) {
    // LDS image of one operand slab (16 k): plane (q, h) holds, for every tile row, the 4 floats k = 8q + 4h .. +3.
    // Lane (r = l&31, h = l>>5) of a wave reads its float4 of row r from plane (q, h): 32 lanes x 16 B contiguous,
    // conflict free for ds_read_b128's lane groups.  The MFMA step s of a sub-slab contracts k = {8q+s, 8q+4+s}.
    __shared__ __attribute__((aligned(16))) float As[2][SG_IMG];
    constexpr int SG_WJ = WJ, SG_TN = SgTile<WJ>::TN, SG_PLANE_B = SgTile<WJ>::PLANE_B, SG_LB = SgTile<WJ>::LB, SG_SUPER_N = SgTile<WJ>::SUPER_N;
    __shared__ __attribute__((aligned(16))) float Bs[2][SgTile<WJ>::IMG_B];
    __shared__ int fl_m[FILTER ? kBlock / 64 : 1][FILTER ? SG_FL_CAP : 1];      // FILTER: per-wave lists of passing elements
    __shared__ int fl_n[FILTER ? kBlock / 64 : 1][FILTER ? SG_FL_CAP : 1];
    __shared__ float fl_v[FILTER ? kBlock / 64 : 1][FILTER ? SG_FL_CAP : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;           // wave grid 2 x 2, each wave 64 x 64

    // Tile ids are walked persistently: block b takes ids b, b + gridDim, ... (gridDim is a multiple of 8, so a block's
    // ids stay on its XCD).  XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs, so ids {x, x+8, ...}
    // share one L2; those ids walk ONE super-tile of SG_SUPER x SG_SUPER tiles before moving on: its A and B panels
    // stay L2-resident.  Ids that fall outside the matrix (ragged super-tiles) are skipped.
    auto tile_of = [&](int id, int& tm, int& tn) -> bool {
        if (super_order) {
            const int sup_n = (tiles_n + SG_SUPER_N - 1) / SG_SUPER_N, sup_m = (tiles_m + SG_SUPER - 1) / SG_SUPER;
            const int per = SG_SUPER * SG_SUPER_N;
            const int xcd = id & 7, local = id >> 3;
            const int sup = (local / per) * 8 + xcd, within = local % per;
            if (sup >= sup_m * sup_n) return false;
            tm = (sup / sup_n) * SG_SUPER + within / SG_SUPER_N;
            tn = (sup % sup_n) * SG_SUPER_N + within % SG_SUPER_N;
            return tm < tiles_m && tn < tiles_n;
        }
        tm = id / tiles_n;                           // few tiles: plain row-major order, every XCD busy
        tn = id % tiles_n;
        return true;
    };
    auto next_tile = [&](int id, int& tm, int& tn) -> int {     // first valid id >= id, or n_ids
        while (id < n_ids && !tile_of(id, tm, tn)) id += gridDim.x;
        return id < n_ids ? id : n_ids;
    };

    // loader: float4 f = tid + 256 i (i < 2): row = f / 4, k quad kq = f % 4 (4 consecutive lanes read 64 contiguous
    // bytes of a row).  Every load is unconditional at a clamped address (a load behind an exec-mask branch is not
    // overlapped with the MFMAs): rows past the end re-read the last row -- they only feed outputs that are never
    // stored -- and float4s past d (d % 4 == 0, enforced by the launcher) are zeroed at store time.
    auto gload = [&](const float* base, int64_t ld, int row0, int nrows, int k0, auto& v) {
        constexpr int CNT = sizeof(v) / sizeof(float4);
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int f = tid + 256 * i;
            const int row = min(row0 + (f >> 2), nrows - 1), k = k0 + 4 * (f & 3);
            v[i] = ld4(base + (int64_t)row * ld + min(k, d - 4));     // zeroing waits for the data: done at store time
        }
    };
    auto sstore = [&](float* S, const auto& v, int k0) {
        constexpr int CNT = sizeof(v) / sizeof(float4);
        constexpr int PLANE = CNT == 2 ? SG_PLANE : SG_PLANE_B;       // (WJ = 2: the two planes are the same size)
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int f = tid + 256 * i;
            // zeroing as a bit mask (exact, and no select the compiler could turn into a branch: the slab body stays ONE basic block)
            const unsigned keep = k0 + 4 * (f & 3) < d ? 0xffffffffu : 0u;
            const float4 x = make_float4(__uint_as_float(__float_as_uint(v[i].x) & keep), __uint_as_float(__float_as_uint(v[i].y) & keep),
                                         __uint_as_float(__float_as_uint(v[i].z) & keep), __uint_as_float(__float_as_uint(v[i].w) & keep));
            *reinterpret_cast<float4*>(S + (f & 3) * PLANE + (f >> 2) * 4) = x;      // plane index = kq = 2q + h
        }
    };
    const int r = lane & 31, h = lane >> 5;
    const int aoff = h * SG_PLANE + (wm * 64 + r) * 4, boff = h * SG_PLANE_B + (wn * 32 * SG_WJ + r) * 4;
    auto frags = [&](int buf, int q, float4 (&af)[2], float4 (&bf)[SG_WJ]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(As[buf] + q * 2 * SG_PLANE + aoff + i * 32 * 4);
#pragma unroll
        for (int j = 0; j < SG_WJ; ++j) bf[j] = *reinterpret_cast<const float4*>(Bs[buf] + q * 2 * SG_PLANE_B + boff + j * 32 * 4);
    };
    f32x16 acc[2][SG_WJ];
    // the 2 WJ accumulators rotate (dependency distance 2 WJ >= 4)
    auto mfma16 = [&](const float4 (&af)[2], const float4 (&bf)[SG_WJ]) {
#define JMAC_SG_STEP(c)                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
            _Pragma("unroll") for (int j = 0; j < SG_WJ; ++j)                                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].c, bf[j].c, acc[i][j], 0, 0, 0);
        JMAC_SG_STEP(x)
        JMAC_SG_STEP(y)
        JMAC_SG_STEP(z)
        JMAC_SG_STEP(w)
#undef JMAC_SG_STEP
    };

    const int nk = (d + SG_K - 1) / SG_K;
    float4 ra[2], rb[SG_LB], af0[2], bf0[SG_WJ], af1[2], bf1[SG_WJ];
    int tm, tn;
    int id = next_tile(blockIdx.x, tm, tn);
    if (id >= n_ids) return;
    // A tile's first slab: staged rows -> LDS buffer 0, slab 1 requested, the first fragments read.  Called for the block's first
    // tile here and for every further tile at the END of the loop body, behind the previous tile's epilogue: a separate instance
    // of the code, so that its wait for the slab's loads is counted on that path alone -- the loads were issued BEFORE the epilogue's
    // 64 stores and vmcnt retires in order, so the stores stay in flight (vmcnt(63)).  With the prologue at the loop's top the
    // kernel-entry path (nothing but four loads pending) and the back edge merged into vmcnt(3): every tile began by waiting for
    // the previous tile's stores to be acknowledged (round 6: the "nostore" ablation's 6 %).
    auto tile_prologue = [&](const int m0, const int n0) {
        sstore(As[0], ra, 0);
        sstore(Bs[0], rb, 0);
        if (nk > 1 && !SG_NO_LOAD) {
            gload(A, lda, m0, M, SG_K, ra);
            gload(Bm, ldb, n0, N, SG_K, rb);
        }
        __syncthreads();
        frags(0, 0, af0, bf0);
        if (SG_MFMA_ONLY) frags(0, 1, af1, bf1);
    };
    gload(A, lda, tm * SG_T, M, 0, ra);
    gload(Bm, ldb, tn * SG_TN, N, 0, rb);
    tile_prologue(tm * SG_T, tn * SG_TN);
    while (id < n_ids) {
        const int m0 = tm * SG_T, n0 = tn * SG_TN;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < SG_WJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        // Software pipeline, one barrier per slab.  At the top of slab kt: fragment set 0 holds (kt, q=0), the staging registers
        // hold slab kt+1 (loads in flight).  One slab is ONE basic block (round 6): no zeroing select, no loop-carried branch --
        // HAS1 = slab kt+1 exists (its staged rows go to LDS, its first fragments are read behind the barrier), HAS2 = slab kt+2
        // exists (requested); the steady state (both) runs in the loop, the last two slabs are peeled -- and inside it the LDS and
        // memory operations are spread between the MFMAs (sched_group_barrier): one operation behind each MFMA instead of a burst
        // behind sixteen of them (ablation and A/B: profiles/r6_simgemm_ablation.txt; +3 % over the burst form).
        auto slab_body = [&](const int kt, auto has1, auto has2) {
            constexpr bool HAS1 = decltype(has1)::value && !SG_MFMA_ONLY, HAS2 = decltype(has2)::value && !SG_MFMA_ONLY && !SG_NO_LOAD;
            const int cur = kt & 1;
            if (!SG_MFMA_ONLY) frags(cur, 1, af1, bf1);
            mfma16(af0, bf0);
            if constexpr (HAS1) {
                sstore(As[cur ^ 1], ra, (kt + 1) * SG_K);
                sstore(Bs[cur ^ 1], rb, (kt + 1) * SG_K);
                if constexpr (HAS2) {
                    gload(A, lda, m0, M, (kt + 2) * SG_K, ra);
                    gload(Bm, ldb, n0, N, (kt + 2) * SG_K, rb);
                }
            }
            if constexpr (!SG_MFMA_ONLY) {
                // 2 + WJ fragment reads, 8 WJ MFMAs, then per staged float4 one LDS store and (HAS2) one global load
#pragma unroll
                for (int i = 0; i < 2 + SG_WJ; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8 * SG_WJ - (2 + SG_WJ) - (HAS1 ? 2 + SG_LB : 0) - (HAS2 ? 2 + SG_LB : 0), 0);
                if constexpr (HAS1) {
#pragma unroll
                    for (int i = 0; i < 2 + SG_LB; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                }
                if constexpr (HAS2) {
#pragma unroll
                    for (int i = 0; i < 2 + SG_LB; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                }
                __syncthreads();
            }
            if constexpr (HAS1) frags(cur ^ 1, 0, af0, bf0);
            mfma16(af1, bf1);
            if constexpr (HAS1) {
#pragma unroll
                for (int i = 0; i < 2 + SG_WJ; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
        };
        {
            int kt = 0;
            for (; kt + 2 < nk; ++kt) slab_body(kt, std::true_type{}, std::true_type{});
            if (kt + 1 < nk) {
                slab_body(kt, std::true_type{}, std::false_type{});
                ++kt;
            }
            slab_body(kt, std::false_type{}, std::false_type{});
        }
        if (SG_MFMA_ONLY) __syncthreads();               // (the next tile's first sstore must not race this tile's first fragment reads)
        // the next tile's first slab is requested BEFORE this tile's 64 KB of results are stored: the stores and the
        // loads overlap, and no wave reads LDS any more (the last fragment reads precede the last barrier)
        int ntm = 0, ntn = 0;
        const int nid = next_tile(id + gridDim.x, ntm, ntn);
        if (nid < n_ids && !SG_NO_LOAD) {
            gload(A, lda, ntm * SG_T, M, 0, ra);
            gload(Bm, ldb, ntn * SG_TN, N, 0, rb);
        }
        // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const bool full = m0 + SG_T <= M && n0 + SG_TN <= N;      // block-uniform: interior tiles store without bounds tests
        if constexpr (FILTER) {
            // Passing elements are rare (~k * N / Ns per row: under 1 % of the tile).  A returned global atomic per passing
            // element inside the scan would cost one memory round trip each (measured: the product ran 1.6x longer); the wave
            // first compacts its passing elements into a small LDS list (ballot + prefix popcount, no memory traffic) and
            // then claims their slots with ONE batch of atomics per flush.
            int fcount = 0;                                    // wave-uniform
            auto flush = [&]() {
                __threadfence_block();                         // the list's LDS writes are visible to the whole wave
                for (int e = lane; e < fcount; e += 64) {
                    const int m = fl_m[wave][e];
                    const int pos = atomicAdd(flt.cnt + m, 1);
                    if (pos < flt.cap) {
                        flt.cval[(int64_t)m * flt.cap + pos] = fl_v[wave][e];
                        flt.cidx[(int64_t)m * flt.cap + pos] = fl_n[wave][e];
                    }
                }
                __threadfence_block();
                fcount = 0;
            };
            // thresholds of the wave's 64 rows: ONE coalesced load (lane l <-> row l of the wave's row block); the value for
            // accumulator register (i, reg) comes out of it with two v_readlane (rows (i, reg, h = 0 / 1) are 4 apart)
            const float trow = flt.tau[(int64_t)min(m0 + wm * 64 + lane, M - 1) * flt.tau_stride];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rbase = i * 32 + (reg & 3) + 8 * (reg >> 2);           // compile-time
                    const int m = m0 + wm * 64 + rbase + 4 * h;
                    const float tau = h ? bcast_f(trow, rbase + 4) : bcast_f(trow, rbase);
#pragma unroll
                    for (int j = 0; j < SG_WJ; ++j) {
                        const int n = n0 + wn * 32 * SG_WJ + j * 32 + r;
                        const float v = acc[i][j][reg];
                        const bool pass = m < M && n < N && v >= tau;
                        const unsigned long long mask = __ballot(pass);
                        if (mask != 0ull) {                    // wave-uniform
                            const int np = __popcll(mask);
                            if (fcount + np > SG_FL_CAP) flush();
                            if (pass) {
                                const int slot = fcount + __popcll(mask & ((1ull << lane) - 1ull));
                                fl_m[wave][slot] = m;
                                fl_n[wave][slot] = n + flt.n_off;
                                fl_v[wave][slot] = v;
                            }
                            fcount += np;
                        }
                    }
                }
            flush();
        } else if (SG_NO_STORE || full) {
            // interior tile: 64 unconditional stores per wave, then (its own instance: see tile_prologue) the next tile's first slab
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < SG_WJ; ++j) {
                    float* cbase = C + (int64_t)(m0 + wm * 64 + i * 32 + 4 * h) * ldc + (n0 + wn * 32 * SG_WJ + j * 32 + r);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        if (SG_NO_STORE) {                 // ablation: nothing leaves the CU, the accumulators stay live
                            if (acc[i][j][reg] == 123456.789f) cbase[(int64_t)((reg & 3) + 8 * (reg >> 2)) * ldc] = acc[i][j][reg];
                        } else {
                            cbase[(int64_t)((reg & 3) + 8 * (reg >> 2)) * ldc] = acc[i][j][reg];
                        }
                    }
                }
            if (nid < n_ids) tile_prologue(ntm * SG_T, ntn * SG_TN);
            id = nid;
            tm = ntm;
            tn = ntn;
            continue;
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < SG_WJ; ++j)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int64_t m = m0 + wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                        const int64_t n = n0 + wn * 32 * SG_WJ + j * 32 + r;
                        if (m < M && n < N) C[m * ldc + n] = acc[i][j][reg];
                    }
        }
        id = nid;
        tm = ntm;
        tn = ntn;
        if (id < n_ids) tile_prologue(tm * SG_T, tn * SG_TN);     // (no wave reads LDS any more: see above)
    }
}

// ------------------------------------------------------------------------------------------------
// The same product with the operand slabs DMA'd straight into LDS (global_load_lds_dwordx4), 32 k per slab (round 6 experiment,
// JMAC_SG_GLDS).  What the ablation of the register-staged kernel above left on the table (profiles/r6_simgemm_ablation.txt):
// its loads touch 16 half cache lines per wave instruction, every staged float4 costs a 13-cycle ds_write_b128 on the LDS data
// path, and there is one barrier per 16 k.  Here one wave instruction fetches 8 FULL 128-byte lines (8 rows x 32 k) into 1 KB of
// LDS, nothing is staged in registers, and there is one barrier per 32 k.
//   LDS image of an operand slab: [128 rows][8 slots of 16 B]; slot = kq ^ ((row >> 1) & 7), kq = the float4 index inside the
//   row's 32 k.  The LDS destination of a DMA is lane-linear (wave base + 16 lane), so the swizzle is applied to the SOURCE address
//   (lane l of a DMA fetches row 8c + l/8, float4 kq = (l & 7) ^ ((row >> 1) & 7)) and again on the fragment read: a ds_read_b128
//   lane group (16 rows of one parity pattern, MI355X_MICROARCH.md "LDS") then touches 64 distinct banks.
//   Contraction order per output element: group g of 8 k, step s: k = {8g + s, 8g + 4 + s} -- the order of sim_gemm_kernel, so
//   the results are bit-identical to it (and to cand_select_kernel's recomputation).
//   Pipeline: the (tile, slab) sequence of a block is flat: while slab t is multiplied, slab t+1 -- of this tile or slab 0 of the
//   block's next tile -- is in flight into the other buffer; one raw s_barrier per slab, behind a counted vmcnt wait (the C stores
//   of a finished tile stay in flight across it: vmcnt completes in order, the DMAs were issued first).
// ------------------------------------------------------------------------------------------------
#ifndef JMAC_SG_GLDS
#define JMAC_SG_GLDS 0              // 1: tools/r6_simgemm_ablation.sh variant "glds" (closed: 104 vs 103 TF/s of the kernel above)
#endif
#if JMAC_SG_GLDS
constexpr int GL_K = 32;                       // k per slab
constexpr int GL_OP = SG_T * GL_K;             // floats per operand slab (16 KB)
__device__ __attribute__((aligned(16))) float sg_zero16[4] = {0.f, 0.f, 0.f, 0.f};      // source of the float4s past d

typedef void __attribute__((address_space(3))) * gl_dst_t;
// One LDS-DMA: lane l's 16 bytes at gsrc -> LDS byte address lds_dst + 16 l (lds_dst wave-uniform, in an SGPR).  Inline asm, so
// the compiler neither counts it in its own s_waitcnt bookkeeping nor drains it (vmcnt(0)) before the next ds_read: the kernel
// counts vmcnt by hand (cdna_hip_programming.md section 5, "Pipelining across barriers").  M0 is written and restored in the
// same statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(gl_dst_t)(const_cast<void*>(p));
}
// s_waitcnt immediates (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8])
constexpr int gl_waitcnt(int vm, int lgkm) { return (vm & 0xf) | ((vm >> 4) << 14) | (0x7 << 4) | ((lgkm & 0xf) << 8); }

__global__ __launch_bounds__(kBlock, 2) void sim_gemm_glds_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm,
                                                                int64_t ldb, int M, int N, int d, float* __restrict__ C, int64_t ldc,
                                                                int tiles_m, int tiles_n, int super_order, int n_ids,
                                                                const float* __restrict__ zero16) {
    __shared__ __attribute__((aligned(1024))) float Ls[2][2][GL_OP];          // [buffer][A | B][row * 32 + slot * 4]: 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    auto tile_of = [&](int id, int& tm, int& tn) -> bool {
        if (super_order) {
            const int sup_n = (tiles_n + SG_SUPER - 1) / SG_SUPER, sup_m = (tiles_m + SG_SUPER - 1) / SG_SUPER;
            const int per = SG_SUPER * SG_SUPER;
            const int xcd = id & 7, local = id >> 3;
            const int sup = (local / per) * 8 + xcd, within = local % per;
            if (sup >= sup_m * sup_n) return false;
            tm = (sup / sup_n) * SG_SUPER + within / SG_SUPER;
            tn = (sup % sup_n) * SG_SUPER + within % SG_SUPER;
            return tm < tiles_m && tn < tiles_n;
        }
        tm = id / tiles_n;
        tn = id % tiles_n;
        return true;
    };
    auto next_tile = [&](int id, int& tm, int& tn) -> int {
        while (id < n_ids && !tile_of(id, tm, tn)) id += gridDim.x;
        return id < n_ids ? id : n_ids;
    };
    // DMA side: instruction i of a wave fills rows 8 (4 i + wave) .. + 7 of the slab; the lane's row inside them and its float4
    const int rr = lane >> 3;
    const int kq_src = (lane & 7) ^ ((((wave & 1) << 2) | (rr >> 1)) & 7);   // ((row >> 1) & 7 with row = 32 i + 8 wave + rr)
    const float* pa[4];
    const float* pb[4];
    auto set_rows = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (4 * i + wave) * 8 + rr;
            pa[i] = A + (int64_t)min(m0 + row, M - 1) * lda + 4 * kq_src;
            pb[i] = Bm + (int64_t)min(n0 + row, N - 1) * ldb + 4 * kq_src;
        }
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(&Ls[0][0][0]) + wave * 8 * GL_K * 4);     // wave's first DMA target
    auto issue = [&](int buf, int k0) {
        // float4s past d (only in a row's last slab, d % 32 != 0) come from 16 zero bytes: an integer select on the ADDRESS
        const uint64_t keep = k0 + 4 * kq_src < d ? ~0ull : 0ull, zero = (uint64_t)(uintptr_t)zero16 & ~keep;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t sa = ((uint64_t)(uintptr_t)(pa[i] + k0) & keep) | zero;
            const uint64_t sb = ((uint64_t)(uintptr_t)(pb[i] + k0) & keep) | zero;
            const unsigned dst = lds0 + (unsigned)((buf * 2 * GL_OP + 4 * i * 8 * GL_K) * 4);
            glds16(reinterpret_cast<const void*>(sa), dst);
            glds16(reinterpret_cast<const void*>(sb), dst + GL_OP * 4);
        }
    };
    // fragment side
    const int r = lane & 31, h = lane >> 5;
    const int sw = (r >> 1) & 7;
    const int arow = (wm * 64 + r) * GL_K, brow = (wn * 64 + r) * GL_K;      // + 32 GL_K for the second 32-row sub-tile
    auto frags = [&](int buf, int g, float4 (&af)[2], float4 (&bf)[2]) {
        const int slot = (((2 * g + h) ^ sw) & 7) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = *reinterpret_cast<const float4*>(&Ls[buf][0][arow + i * 32 * GL_K + slot]);
            bf[i] = *reinterpret_cast<const float4*>(&Ls[buf][1][brow + i * 32 * GL_K + slot]);
        }
    };
    f32x16 acc[2][2];
    auto mfma16 = [&](const float4 (&af)[2], const float4 (&bf)[2]) {
#define JMAC_GL_STEP(c)                                                                                   \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0].c, bf[0].c, acc[0][0], 0, 0, 0);          \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0].c, bf[1].c, acc[0][1], 0, 0, 0);          \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1].c, bf[0].c, acc[1][0], 0, 0, 0);          \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1].c, bf[1].c, acc[1][1], 0, 0, 0);
        JMAC_GL_STEP(x)
        JMAC_GL_STEP(y)
        JMAC_GL_STEP(z)
        JMAC_GL_STEP(w)
#undef JMAC_GL_STEP
    };
    const int nk = (d + GL_K - 1) / GL_K;
    int tm, tn;
    int id = next_tile(blockIdx.x, tm, tn);
    if (id >= n_ids) return;
    set_rows(tm * SG_T, tn * SG_T);
    issue(0, 0);
    int buf = 0;
    __builtin_amdgcn_s_waitcnt(gl_waitcnt(0, 0));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float4 af[2][2], bf[2][2];
    while (id < n_ids) {
        const int m0 = tm * SG_T, n0 = tn * SG_T;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        int ntm = 0, ntn = 0;
        const int nid = next_tile(id + gridDim.x, ntm, ntn);
        // every slab but the tile's last: the tile's next slab goes into the other buffer (every wave has passed the barrier behind
        // its last fragment read of that buffer), four groups of 8 k are multiplied with alternating fragment sets
        for (int kt = 0; kt + 1 < nk; ++kt) {
            issue(buf ^ 1, (kt + 1) * GL_K);
            frags(buf, 0, af[0], bf[0]);
            __builtin_amdgcn_sched_barrier(0);
            // group g's 16 MFMAs carry group g+1's four fragment reads between them (one read behind each of the first four)
#define JMAC_GL_OVERLAP()                                                                  \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                             \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                         \
            }                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);                            \
            __builtin_amdgcn_sched_barrier(0);
            frags(buf, 1, af[1], bf[1]);
            mfma16(af[0], bf[0]);
            JMAC_GL_OVERLAP()
            frags(buf, 2, af[0], bf[0]);
            mfma16(af[1], bf[1]);
            JMAC_GL_OVERLAP()
            frags(buf, 3, af[1], bf[1]);
            mfma16(af[0], bf[0]);
            JMAC_GL_OVERLAP()
#undef JMAC_GL_OVERLAP
            mfma16(af[1], bf[1]);
            __builtin_amdgcn_s_waitcnt(gl_waitcnt(0, 0));      // the wave's own DMAs of the next slab have landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");                     // no fragment read of the next slab moves above the barrier
            buf ^= 1;
        }
        // the last slab (d % 32 != 0: fewer groups): slab 0 of the block's NEXT tile is requested first
        if (nid < n_ids) {
            set_rows(ntm * SG_T, ntn * SG_T);
            issue(buf ^ 1, 0);
        }
        {
            const int ng = (d - (nk - 1) * GL_K + 7) >> 3;
            frags(buf, 0, af[0], bf[0]);
            for (int g = 0; g < ng; ++g) {
                mfma16(af[0], bf[0]);
                if (g + 1 < ng) frags(buf, g + 1, af[0], bf[0]);
            }
        }
        {
            // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
            const bool full = m0 + SG_T <= M && n0 + SG_T <= N;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float* cbase = C + (int64_t)(m0 + wm * 64 + i * 32 + 4 * h) * ldc + (n0 + wn * 64 + j * 32 + r);
                    if (full) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) cbase[(int64_t)((reg & 3) + 8 * (reg >> 2)) * ldc] = acc[i][j][reg];
                    } else {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int64_t m = m0 + wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                            const int64_t n = n0 + wn * 64 + j * 32 + r;
                            if (m < M && n < N) C[m * ldc + n] = acc[i][j][reg];
                        }
                    }
                }
        }
        // the 8 DMAs of the next tile's first slab were issued before the (up to 64) stores: vmcnt retires in order
        __builtin_amdgcn_s_waitcnt(gl_waitcnt(63, 0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        buf ^= 1;
        id = nid;
        tm = ntm;
        tn = ntn;
    }
}

#endif  // JMAC_SG_GLDS

// ------------------------------------------------------------------------------------------------
// row top-k: one block per row.  Order: value descending, ties -> lower index first.
//   pass A  histogram of the 12 leading bits of an order-preserving key -> the bin b* that holds the k-th largest value
//   pass B  every entry of bins >= b* (a few dozen for similarity rows) is collected in LDS
//   select  each candidate counts the candidates that beat it: that count is its output slot
// Two streaming passes over the row instead of k; rows whose candidate set exceeds the LDS list (e.g. constant rows)
// take k rounds of (value, index) arg-max over the row in place.
// ------------------------------------------------------------------------------------------------
constexpr int TK_BINS = 4096, TK_CAP = 1024;

__device__ __forceinline__ unsigned tk_key(float v) {          // ascending float order == ascending unsigned order
    const unsigned u = v == 0.f ? 0u : __float_as_uint(v);    // -0 and +0 compare equal: one key
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// a beats b: larger value, or equal value and lower index
__device__ __forceinline__ bool tk_beats(unsigned ka, int ia, unsigned kb, int ib) { return ka > kb || (ka == kb && ia < ib); }

__global__ __launch_bounds__(kBlock) void row_topk_kernel(const float* __restrict__ S, int64_t lds, int N, int k,
                                                          float* __restrict__ val, int32_t* __restrict__ idx) {
    __shared__ int hist[TK_BINS];
    __shared__ unsigned ckey[TK_CAP];
    __shared__ int cidx[TK_CAP];
    __shared__ int chunk_sum[kBlock];
    __shared__ int sh_bin, sh_above, sh_cnt;
    __shared__ float wv[kBlock / 64];
    __shared__ int wi[kBlock / 64];
    __shared__ float pick_v;
    __shared__ int pick_i;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = S + (int64_t)b * lds;
    const bool vec = (((uintptr_t)row) & 15) == 0;
    for (int i = tid; i < TK_BINS; i += kBlock) hist[i] = 0;
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    // ---- pass A
    const int N4 = vec ? (N >> 2) : 0;
    for (int q = tid; q < N4; q += kBlock) {
        const float4 v = ld4(row + 4 * q);
        atomicAdd(&hist[tk_key(v.x) >> 20], 1);
        atomicAdd(&hist[tk_key(v.y) >> 20], 1);
        atomicAdd(&hist[tk_key(v.z) >> 20], 1);
        atomicAdd(&hist[tk_key(v.w) >> 20], 1);
    }
    for (int n = 4 * N4 + tid; n < N; n += kBlock) atomicAdd(&hist[tk_key(row[n]) >> 20], 1);
    __syncthreads();
    // ---- b*: chunk c = bins [BINS - 16(c+1), BINS - 16c), scanned from the top
    {
        int s16 = 0;
        const int hi = TK_BINS - 16 * tid;
#pragma unroll
        for (int j = 1; j <= 16; ++j) s16 += hist[hi - j];
        chunk_sum[tid] = s16;
        __syncthreads();
        if (tid < 64) {          // one wave: inclusive scan of the 256 chunk sums, 4 per lane
            int a0 = chunk_sum[4 * tid], a1 = chunk_sum[4 * tid + 1], a2 = chunk_sum[4 * tid + 2], a3 = chunk_sum[4 * tid + 3];
            const int tot = a0 + a1 + a2 + a3;
            int inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, 64);
                if (tid >= o) inc += t;
            }
            int before = inc - tot;                       // entries in chunks above this lane's four
            const int sums[4] = {a0, a1, a2, a3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (before < k && before + sums[j] >= k) {   // the k-th largest lies in chunk 4*tid + j
                    const int top = TK_BINS - 16 * (4 * tid + j);
                    int above = before;
                    for (int t = 1; t <= 16; ++t) {
                        const int h = hist[top - t];
                        if (above + h >= k) {
                            sh_bin = top - t;
                            sh_above = above;
                            break;
                        }
                        above += h;
                    }
                }
                before += sums[j];
            }
        }
        __syncthreads();
    }
    const int bstar = sh_bin;
    const int C = sh_above + hist[bstar];               // candidates = all entries of bins >= b*  (C >= k)
    if (C <= TK_CAP) {
        // ---- pass B
        for (int q = tid; q < N4; q += kBlock) {
            const float4 v = ld4(row + 4 * q);
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned key = tk_key(e[j]);
                if ((int)(key >> 20) >= bstar) {
                    const int slot = atomicAdd(&sh_cnt, 1);
                    ckey[slot] = key;
                    cidx[slot] = 4 * q + j;
                }
            }
        }
        for (int n = 4 * N4 + tid; n < N; n += kBlock) {
            const unsigned key = tk_key(row[n]);
            if ((int)(key >> 20) >= bstar) {
                const int slot = atomicAdd(&sh_cnt, 1);
                ckey[slot] = key;
                cidx[slot] = n;
            }
        }
        __syncthreads();
        // ---- select: slot = number of candidates that beat this one (keys are distinct as (key, index) pairs)
        for (int c = tid; c < C; c += kBlock) {
            const unsigned kc = ckey[c];
            const int ic = cidx[c];
            int better = 0;
            int o = 0;
            for (; o + 8 <= C; o += 8) {                     // eight independent LDS reads in flight per trip (a rolled loop
                unsigned k8[8];                                // waits out the LDS latency once per candidate: 80 cycles each)
                int i8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    k8[u] = ckey[o + u];
                    i8[u] = cidx[o + u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) better += tk_beats(k8[u], i8[u], kc, ic) ? 1 : 0;
            }
            for (; o < C; ++o) better += tk_beats(ckey[o], cidx[o], kc, ic) ? 1 : 0;
            if (better < k) {
                idx[(int64_t)b * k + better] = ic;
                if (val) val[(int64_t)b * k + better] = row[ic];
            }
        }
        return;
    }
    // ---- fallback: k rounds of arg-max after the previous pick
    float pv = INFINITY;
    int pi = -1;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = INT32_MAX;
        for (int n = tid; n < N; n += kBlock) {
            const float v = row[n];
            const bool after = (v < pv) || (v == pv && n > pi);       // not yet picked
            const bool better = (v > bv) || (v == bv && n < bi);
            if (after && better) {
                bv = v;
                bi = n;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((tid & 63) == 0) {
            wv[tid >> 6] = bv;
            wi[tid >> 6] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kBlock / 64; ++w)
                if (wv[w] > bv || (wv[w] == bv && wi[w] < bi)) {
                    bv = wv[w];
                    bi = wi[w];
                }
            pick_v = bv;
            pick_i = bi;
            if (val) val[(int64_t)b * k + r] = bv;
            idx[(int64_t)b * k + r] = bi == INT32_MAX ? -1 : bi;
        }
        __syncthreads();
        pv = pick_v;
        pi = pick_i;
    }
}

// ------------------------------------------------------------------------------------------------
// top-k of a row from its candidate list (fused similarity + top-k, jmac_sim_topk_f32): every element >= tau was listed, and
// at least k elements are (tau is the k-th largest of a column sample), so the list's k best ARE the row's k best; order and
// ties as row_topk_kernel (value descending, lower index first).
// A row whose list overflowed (more than cap elements reach tau: e.g. a constant row) recomputes its N scores with the
// SAME instruction sequence per element as sim_gemm_kernel -- v_mfma_f32_32x32x2_f32 over k in the same order, so the bits
// are the GEMM's -- once per pass of the two-pass selection (and per arg-max round if even that overflows).  Slow by design:
// it exists so that degenerate inputs still get the exact answer.
// ------------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void for_each_row_score(const float* __restrict__ arow, const float* __restrict__ Bm, int64_t ldb, int N,
                                                   int d, F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nk = (d + SG_K - 1) / SG_K;
    for (int n0 = wave * 32; n0 < N; n0 += 32 * (kBlock / 64)) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        const float* brow = Bm + (int64_t)min(n0 + r, N - 1) * ldb;
        for (int kt = 0; kt < nk; ++kt)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = kt * SG_K + 8 * q + 4 * h;
                float4 a4 = ld4(arow + min(k, d - 4)), b4 = ld4(brow + min(k, d - 4));
                if (k >= d) a4 = b4 = f4zero();
                if (r != 0) a4 = f4zero();                     // row 0 of the 32x32 tile is the row; the rest is padding
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
        if (h == 0 && n0 + r < N) f(n0 + r, acc[0]);           // C/D map: row 0 = register 0 of the lanes with h == 0
    }
}

__global__ __launch_bounds__(kBlock) void cand_select_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm,
                                                             int64_t ldb, int N, int d, int k, const int* __restrict__ cnt,
                                                             const float* __restrict__ cval, const int* __restrict__ cidx_g, int cap,
                                                             const float* __restrict__ sval, const int32_t* __restrict__ sidx,
                                                             float* __restrict__ val, int32_t* __restrict__ idx) {
    __shared__ int hist[TK_BINS];
    __shared__ unsigned ckey[TK_CAP];
    __shared__ int cidx[TK_CAP];
    __shared__ unsigned long long skey[TK_CAP];
    __shared__ int sh_bin, sh_above, sh_cnt;
    __shared__ float wv[kBlock / 64];
    __shared__ int wi[kBlock / 64];
    __shared__ float pick_v;
    __shared__ int pick_i;
    const int b = blockIdx.x, tid = threadIdx.x;
    int C = cnt[b];
    // the Cn candidates in LDS -> their k best: a bitonic sort of (key, ~index) pairs, descending (value descending, lower
    // index first).  Ranking every candidate against every other one costs Cn^2 compares per row: 70 us for 3 000 rows of
    // ~300 candidates; the sort is Cn log^2 Cn.
    auto select = [&](int Cn) {
        int P = 64;
        while (P < Cn) P <<= 1;                                // <= TK_CAP (a power of two)
        for (int c = tid; c < P; c += kBlock)
            skey[c] = c < Cn ? (((unsigned long long)ckey[c] << 32) | (unsigned)(~cidx[c])) : 0ull;
        __syncthreads();
        for (int size = 2; size <= P; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < (P >> 1); t += kBlock) {
                    const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                    const bool desc = (lo & size) == 0;        // descending runs first: the result is descending overall
                    const unsigned long long x = skey[lo], y = skey[hi];
                    if ((x < y) == desc) {
                        skey[lo] = y;
                        skey[hi] = x;
                    }
                }
                __syncthreads();
            }
        for (int c = tid; c < k; c += kBlock) {
            const unsigned long long e = skey[c];
            const int ic = (int)~(unsigned)(e & 0xffffffffull);
            idx[(int64_t)b * k + c] = ic;
            if (val) {
                const unsigned key = (unsigned)(e >> 32);      // invert tk_key
                const unsigned u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
                val[(int64_t)b * k + c] = __uint_as_float(u);
            }
        }
    };
    if (C + k <= cap) {                                        // the normal case (cap <= TK_CAP)
        // candidates: the row's list (columns past the sample) + the sample's own k best (the threshold came from them; a
        // sample element tied with tau but outside that list has a higher index than every listed tie: it cannot win)
        for (int c = tid; c < C; c += kBlock) {
            const float v = cval[(int64_t)b * cap + c];
            ckey[c] = tk_key(v);
            cidx[c] = cidx_g[(int64_t)b * cap + c];
        }
        for (int c = tid; c < k; c += kBlock) {
            const float v = sval[(int64_t)b * k + c];
            ckey[C + c] = tk_key(v);
            cidx[C + c] = sidx[(int64_t)b * k + c];
        }
        __syncthreads();
        select(C + k);
        return;
    }
    // ---- overflow: two-pass selection over recomputed scores
    const float* arow = A + (int64_t)b * lda;
    for (int i = tid; i < TK_BINS; i += kBlock) hist[i] = 0;
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    for_each_row_score(arow, Bm, ldb, N, d, [&](int n, float v) { atomicAdd(&hist[tk_key(v) >> 20], 1); });
    __syncthreads();
    if (tid == 0) {                                            // (rare path: a serial scan from the top bin is fine)
        int above = 0, bin = TK_BINS - 1;
        for (; bin > 0; --bin) {
            if (above + hist[bin] >= k) break;
            above += hist[bin];
        }
        sh_bin = bin;
        sh_above = above;
    }
    __syncthreads();
    const int bstar = sh_bin;
    C = sh_above + hist[bstar];
    if (C <= TK_CAP) {
        for_each_row_score(arow, Bm, ldb, N, d, [&](int n, float v) {
            const unsigned key = tk_key(v);
            if ((int)(key >> 20) >= bstar) {
                const int slot = atomicAdd(&sh_cnt, 1);
                ckey[slot] = key;
                cidx[slot] = n;
            }
        });
        __syncthreads();
        select(C);
        return;
    }
    // ---- even the top bin overflows (e.g. a constant row): k rounds of arg-max after the previous pick
    float pv = INFINITY;
    int pi = -1;
    for (int rr = 0; rr < k; ++rr) {
        float bv = -INFINITY;
        int bi = INT32_MAX;
        for_each_row_score(arow, Bm, ldb, N, d, [&](int n, float v) {
            const bool after = (v < pv) || (v == pv && n > pi);
            const bool better = (v > bv) || (v == bv && n < bi);
            if (after && better) {
                bv = v;
                bi = n;
            }
        });
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((tid & 63) == 0) {
            wv[tid >> 6] = bv;
            wi[tid >> 6] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kBlock / 64; ++w)
                if (wv[w] > bv || (wv[w] == bv && wi[w] < bi)) {
                    bv = wv[w];
                    bi = wi[w];
                }
            pick_v = bv;
            pick_i = bi;
            if (val) val[(int64_t)b * k + rr] = bv;
            idx[(int64_t)b * k + rr] = bi == INT32_MAX ? -1 : bi;
        }
        __syncthreads();
        pv = pick_v;
        pi = pick_i;
    }
}

// ------------------------------------------------------------------------------------------------
// column top-k values (CSLS column term, similarity.py:58-78 on the transposed matrix) without transposing:
// one lane per column (a wave reads 256 contiguous bytes of a row), rows split over the 4 waves of a block and over
// gridDim.y row ranges; every (block, wave) keeps its column's k largest values in registers (branch-free bubble
// insertion, skipped by the whole wave when no lane's value beats its current k-th) and writes them as a candidate
// list; a second kernel merges the lists of a column.  Output [n2, k] sorted descending.
// ------------------------------------------------------------------------------------------------
constexpr int CT_KMAX = 16;

// inserts v into the descending list top[0..k) and returns its new k-th (smallest kept) value; register-only:
// the list is never indexed with a run-time subscript
__device__ __forceinline__ float ct_insert(float (&top)[CT_KMAX], int k, float v) {
    float x = v, kth = -INFINITY;
#pragma unroll
    for (int j = 0; j < CT_KMAX; ++j)
        if (j < k) {
            const float hi = fmaxf(top[j], x);
            x = fminf(top[j], x);
            top[j] = hi;
            kth = j == k - 1 ? hi : kth;
        }
    return kth;
}

__global__ __launch_bounds__(kBlock) void col_topk_partial_kernel(const float* __restrict__ S, int64_t lds, int n1, int n2, int k,
                                                                  float* __restrict__ cand) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int rows_per = (n1 + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(n1, r0 + rows_per);
    float top[CT_KMAX];
#pragma unroll
    for (int j = 0; j < CT_KMAX; ++j) top[j] = -INFINITY;
    const int cc = min(col, n2 - 1);
    float kth = -INFINITY;
    for (int r = r0 + wave; r < r1; r += kBlock / 64) {
        const float v = col < n2 ? S[(int64_t)r * lds + cc] : -INFINITY;
        if (__any(v > kth)) kth = ct_insert(top, k, v);
    }
    const int part = blockIdx.y * (kBlock / 64) + wave;
    if (col < n2) {
        float* o = cand + ((int64_t)part * n2 + col) * k;
#pragma unroll
        for (int j = 0; j < CT_KMAX; ++j)
            if (j < k) o[j] = top[j];
    }
}

__global__ __launch_bounds__(kBlock) void col_topk_merge_kernel(const float* __restrict__ cand, int parts, int n2, int k,
                                                                float* __restrict__ out) {
    const int col = blockIdx.x * kBlock + threadIdx.x;
    if (col >= n2) return;
    float top[CT_KMAX];
#pragma unroll
    for (int j = 0; j < CT_KMAX; ++j) top[j] = -INFINITY;
    float kth = -INFINITY;
    for (int p = 0; p < parts; ++p) {
        const float* c = cand + ((int64_t)p * n2 + col) * k;
        for (int j = 0; j < k; ++j) {
            const float v = c[j];
            if (!(v > kth)) break;                        // lists are sorted: nothing further of this one can enter
            kth = ct_insert(top, k, v);
        }
    }
#pragma unroll
    for (int j = 0; j < CT_KMAX; ++j)
        if (j < k) out[(int64_t)col * k + j] = top[j];
}

// ------------------------------------------------------------------------------------------------
// softmax entropy of the rows of scale*S;  masked row softmax
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < kBlock / 64; ++w) r = fmaxf(r, sh[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < kBlock / 64; ++w) r += sh[w];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kBlock) void row_entropy_kernel(const float* __restrict__ S, int64_t lds, int N, float scale,
                                                             float* __restrict__ ent) {
    __shared__ float sh[kBlock / 64];
    const float* row = S + (int64_t)blockIdx.x * lds;
    float m = -INFINITY;
    for (int n = threadIdx.x; n < N; n += kBlock) m = fmaxf(m, row[n] * scale);
    m = block_max(m, sh);
    float z = 0.f, y = 0.f;
    for (int n = threadIdx.x; n < N; n += kBlock) {
        const float x = row[n] * scale - m;
        const float e = expf(x);
        z += e;
        y = fmaf(e, x, y);
    }
    z = block_sum(z, sh);
    y = block_sum(y, sh);
    if (threadIdx.x == 0) ent[blockIdx.x] = logf(z) - y / z;
}

// softmax over a ROW of the masked, scaled matrix; out (the probabilities) and ent (the row's softmax entropy) are optional
__global__ __launch_bounds__(kBlock) void masked_row_softmax_kernel(const float* __restrict__ S, int64_t lds, int N,
                                                                    const uint8_t* __restrict__ row_mask,
                                                                    const uint8_t* __restrict__ col_mask, float fill, float scale,
                                                                    float* __restrict__ out, int64_t ldo, float* __restrict__ ent) {
    __shared__ float sh[kBlock / 64];
    const int i = blockIdx.x;
    const float* row = S + (int64_t)i * lds;
    const bool rk = row_mask ? row_mask[i] != 0 : true;
    auto val = [&](int n) -> float {
        const bool keep = rk && (col_mask ? col_mask[n] != 0 : true);
        return (keep ? row[n] : fill) * scale;
    };
    float m = -INFINITY;
    for (int n = threadIdx.x; n < N; n += kBlock) m = fmaxf(m, val(n));
    m = block_max(m, sh);
    float z = 0.f, y = 0.f;
    for (int n = threadIdx.x; n < N; n += kBlock) {
        const float x = val(n) - m;
        const float e = expf(x);
        z += e;
        y = fmaf(e, x, y);
    }
    z = block_sum(z, sh);
    if (ent) {
        y = block_sum(y, sh);
        if (threadIdx.x == 0) ent[i] = logf(z) - y / z;
    }
    if (out) {
        float* orow = out + (int64_t)i * ldo;
        const float iz = 1.f / z;
        for (int n = threadIdx.x; n < N; n += kBlock) orow[n] = expf(val(n) - m) * iz;
    }
}

// ------------------------------------------------------------------------------------------------
// softmax over the COLUMNS of the same masked, scaled matrix, from the one row-major S (no second, transposed GEMM):
//   stats   one lane per column (a wave reads 256 contiguous bytes of a row), rows split over the 4 waves of a block and
//           over gridDim.y row ranges; online (max, sum e^x, sum e^x x) per lane, one partial triple per (part, column)
//   merge   combines the partials of a column in part order -> (max, 1/sum) per column and the column's entropy
//   apply   64x64 tile through LDS: p = exp(x - max_j) / sum_j, written TRANSPOSED ([n2, n1], coalesced both ways)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cs_push(float& m, float& z, float& y, float x) {
    // running (m, z = sum e^(x-m), y = sum e^(x-m) (x-m)) with a new element x
    const float mn = fmaxf(m, x);
    const float f = expf(m - mn);                  // m = -inf: f = 0 (z = y = 0 there)
    const float dm = m - mn;                       // <= 0; -inf - finite = -inf only while z = y = 0
    y = f * (y + (z > 0.f ? dm * z : 0.f));
    z = f * z;
    const float xe = x - mn;
    const float e = expf(xe);
    z += e;
    y = fmaf(e, xe, y);
    m = mn;
}

__global__ __launch_bounds__(kBlock) void col_softmax_stats_kernel(const float* __restrict__ S, int64_t lds, int n1, int n2,
                                                                   const uint8_t* __restrict__ row_mask,
                                                                   const uint8_t* __restrict__ col_mask, float fill, float scale,
                                                                   float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int rows_per = (n1 + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(n1, r0 + rows_per);
    const int cc = min(col, n2 - 1);
    const bool ck = col_mask ? col_mask[cc] != 0 : true;
    float m = -INFINITY, z = 0.f, y = 0.f;
    for (int r = r0 + wave; r < r1; r += kBlock / 64) {
        const bool keep = ck && (row_mask ? row_mask[r] != 0 : true);
        const float v = S[(int64_t)r * lds + cc];
        cs_push(m, z, y, (keep ? v : fill) * scale);
    }
    const int p = blockIdx.y * (kBlock / 64) + wave;
    if (col < n2) {
        float* o = part + ((int64_t)p * n2 + col) * 3;
        o[0] = m;
        o[1] = z;
        o[2] = y;
    }
}

__global__ __launch_bounds__(kBlock) void col_softmax_merge_kernel(const float* __restrict__ part, int parts, int n2,
                                                                   float* __restrict__ cmax, float* __restrict__ cinv,
                                                                   float* __restrict__ ent) {
    const int col = blockIdx.x * kBlock + threadIdx.x;
    if (col >= n2) return;
    float m = -INFINITY, z = 0.f, y = 0.f;
    for (int p = 0; p < parts; ++p) {
        const float* q = part + ((int64_t)p * n2 + col) * 3;
        const float pm = q[0], pz = q[1], py = q[2];
        if (!(pz > 0.f)) continue;                                  // a part without rows
        const float mn = fmaxf(m, pm);
        const float fa = expf(m - mn), fb = expf(pm - mn);
        const float da = m - mn, db = pm - mn;
        const float ya = z > 0.f ? fa * (y + da * z) : 0.f;
        const float yb = fb * (py + db * pz);
        y = ya + yb;
        z = fa * z + fb * pz;
        m = mn;
    }
    cmax[col] = m;
    cinv[col] = 1.f / z;
    if (ent) ent[col] = logf(z) - y / z;
}

__global__ __launch_bounds__(kBlock) void col_softmax_apply_kernel(const float* __restrict__ S, int64_t lds, int n1, int n2,
                                                                   const uint8_t* __restrict__ row_mask,
                                                                   const uint8_t* __restrict__ col_mask, float fill, float scale,
                                                                   const float* __restrict__ cmax, const float* __restrict__ cinv,
                                                                   float* __restrict__ out_t, int64_t ldo) {
    __shared__ float tile[64][65];
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = j0 + tx;
    const bool jin = j < n2;
    const int jc = min(j, n2 - 1);
    const bool ck = col_mask ? col_mask[jc] != 0 : true;
    const float mj = cmax[jc], zj = cinv[jc];
    for (int r = ty; r < 64; r += kBlock / 64) {
        const int i = i0 + r;
        float p = 0.f;
        if (i < n1 && jin) {
            const bool keep = ck && (row_mask ? row_mask[i] != 0 : true);
            const float x = (keep ? S[(int64_t)i * lds + j] : fill) * scale;
            p = expf(x - mj) * zj;
        }
        tile[r][tx] = p;
    }
    __syncthreads();
    const int i = i0 + tx;
    for (int c = ty; c < 64; c += kBlock / 64) {
        const int jj = j0 + c;
        if (jj < n2 && i < n1) out_t[(int64_t)jj * ldo + i] = tile[tx][c];
    }
}

// out[i][j] = 2*S[i][j] - r1[i] - r2[j]   (CSLS, modules/finding/similarity.py:58-78)
__global__ __launch_bounds__(kBlock) void csls_apply_kernel(const float* __restrict__ S, int64_t lds, int64_t n1, int64_t n2,
                                                            const float* __restrict__ r1, const float* __restrict__ r2,
                                                            float* __restrict__ out, int64_t ldo) {
    const int64_t total = n1 * n2;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / n2, c = i % n2;
        out[r * ldo + c] = 2.f * S[r * lds + c] - r1[r] - r2[c];
    }
}

// rank of column gold[i] in row i of the CSLS-rescored matrix 2 S - r1[i] - r2[j], descending, ties -> lower index first:
// csls_apply + filtered_rank(descending) without writing or re-reading the rescored matrix.
__global__ __launch_bounds__(kBlock) void csls_rank_kernel(const float* __restrict__ S, int64_t lds, int n2,
                                                           const float* __restrict__ r1, const float* __restrict__ r2,
                                                           const int32_t* __restrict__ gold, int32_t* __restrict__ rank) {
    __shared__ int red[kBlock / 64];
    const int i = blockIdx.x;
    const float* row = S + (int64_t)i * lds;
    const float a = r1[i];
    const int g = gold[i];
    const float gs = 2.f * row[g] - a - r2[g];
    int cnt = 0;
    for (int n = threadIdx.x; n < n2; n += kBlock) {
        const float v = 2.f * row[n] - a - r2[n];
        cnt += (v > gs || (v == gs && n < g)) ? 1 : 0;
    }
    cnt = wave_sum_i(cnt);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += red[w];
        rank[i] = t + 1;
    }
}

// One shape's launch geometry with WJ column sub-tiles per wave: tiles, the XCD-aware id space and the blocks resident at once.
template <int WJ>
struct SimGeom {
    int tiles_m, tiles_n, super_order;
    int64_t n_ids, tiles;
    SimGeom(int64_t M, int64_t N) {
        tiles_m = (int)((M + SG_T - 1) / SG_T);
        tiles_n = (int)((N + SgTile<WJ>::TN - 1) / SgTile<WJ>::TN);
        const int64_t sup = (int64_t)((tiles_m + SG_SUPER - 1) / SG_SUPER) * ((tiles_n + SgTile<WJ>::SUPER_N - 1) / SgTile<WJ>::SUPER_N);
        super_order = sup >= 64 ? 1 : 0;                                  // >= 8 super-tiles per XCD: the tail imbalance is small
        tiles = (int64_t)tiles_m * tiles_n;
        n_ids = super_order ? (sup + 7) / 8 * 8 * SG_SUPER * SgTile<WJ>::SUPER_N : tiles;
    }
};

// persistent grids: as many blocks as are resident at once (occupancy query: 3 per CU for the 128 x 128 tile -- 64 accumulation
// VGPRs, 33 KB of LDS -- and 2 for 128 x 256), rounded down to a multiple of 8 so that a block's tile ids stay on its XCD
template <bool FILTER, int WJ>
int sim_resident(int dev) {
    static int resident_of[64] = {0};                                     // per device: CU counts may differ between devices
    int& resident = resident_of[dev >= 0 && dev < 64 ? dev : 0];
    if (resident == 0 || dev >= 64) {
        int cus = 256, occ = SgTile<WJ>::OCC;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(sim_gemm_kernel<FILTER, WJ>), kBlock, 0) != hipSuccess || occ < 1)
            occ = SgTile<WJ>::OCC;
        resident = (cus * occ) / 8 * 8;
        if (resident < 8) resident = 8;
    }
    return resident;
}

template <bool FILTER, int WJ>
int launch_sim_wj(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t d, float* C, int64_t ldc,
                  hipStream_t st, const SimFilter* flt, int dev) {
    const SimGeom<WJ> g(M, N);
    if (g.n_ids >= INT32_MAX) return JMAC_ERANGE;
    const int resident = sim_resident<FILTER, WJ>(dev);
    const unsigned grid = (unsigned)(g.n_ids < resident ? g.n_ids : resident);
    hipLaunchKernelGGL((sim_gemm_kernel<FILTER, WJ>), dim3(grid), dim3(kBlock), 0, st, A, lda, B, ldb, (int)M, (int)N, (int)d, C, ldc,
                       g.tiles_m, g.tiles_n, g.super_order, (int)g.n_ids, flt ? *flt : SimFilter{});
    return (int)hipGetLastError();
}

int launch_sim(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t d, float* C, int64_t ldc,
               hipStream_t st, const SimFilter* flt = nullptr) {
    if (M == 0 || N == 0) return 0;
    if (d % 4) return JMAC_EDIM;
    int dev = 0;
    (void)hipGetDevice(&dev);
#if JMAC_SG_GLDS
    if (!flt) {                                                           // closed experiment: the LDS-DMA form of the plain product
        const SimGeom<2> g(M, N);
        if (g.n_ids >= INT32_MAX) return JMAC_ERANGE;
        static const float* zero_of[64] = {nullptr};
        const float*& zero16 = zero_of[dev >= 0 && dev < 64 ? dev : 0];
        if (zero16 == nullptr || dev >= 64) {
            void* p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(sg_zero16)) != hipSuccess) return (int)hipGetLastError();
            zero16 = static_cast<const float*>(p);
        }
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int res2 = (cus * 2) / 8 * 8;
        hipLaunchKernelGGL(sim_gemm_glds_kernel, dim3((unsigned)(g.n_ids < res2 ? g.n_ids : res2)), dim3(kBlock), 0, st, A, lda, B, ldb, (int)M,
                           (int)N, (int)d, C, ldc, g.tiles_m, g.tiles_n, g.super_order, (int)g.n_ids, zero16);
        return (int)hipGetLastError();
    }
#endif
    // Which tile?  Both give the same bits.  A round of resident blocks is (resident blocks) x (tile area) of matrix-pipe work, and
    // the wide tile does a given amount of it ~3.5 % faster (fewer barriers and operand bytes per flop) -- but its rounds are coarser.
    // Take it when its quantised makespan, rounds x resident x area, is no larger than the narrow tile's (measured, config-5 shapes:
    // 12 000^2 and 3 000 x 30 000 tie on that count and gain 3.4 / 3.5 %; 10 500^2 needs 7 168 against 6 912 units and loses 4 %).
#if defined(JMAC_SG_FORCE_WJ)
    const bool wide = JMAC_SG_FORCE_WJ == 4;
#else
    const SimGeom<2> g2(M, N);
    const SimGeom<4> g4(M, N);
    const int64_t r2 = flt ? sim_resident<true, 2>(dev) : sim_resident<false, 2>(dev);
    const int64_t r4 = flt ? sim_resident<true, 4>(dev) : sim_resident<false, 4>(dev);
    const int64_t cost2 = (g2.tiles + r2 - 1) / r2 * r2, cost4 = (g4.tiles + r4 - 1) / r4 * r4 * 2;
    const bool wide = g4.tiles >= r4 && cost4 <= cost2;
#endif
    if (wide)
        return flt ? launch_sim_wj<true, 4>(A, lda, B, ldb, M, N, d, C, ldc, st, flt, dev)
                   : launch_sim_wj<false, 4>(A, lda, B, ldb, M, N, d, C, ldc, st, flt, dev);
    return flt ? launch_sim_wj<true, 2>(A, lda, B, ldb, M, N, d, C, ldc, st, flt, dev)
               : launch_sim_wj<false, 2>(A, lda, B, ldb, M, N, d, C, ldc, st, flt, dev);
}

int launch_topk(const float* S, int64_t lds, int64_t L, int64_t N, int32_t k, float* val, int32_t* idx, hipStream_t st) {
    if (L == 0) return 0;
    hipLaunchKernelGGL(row_topk_kernel, dim3((unsigned)L), dim3(kBlock), 0, st, S, lds, (int)N, (int)k, val, idx);
    return (int)hipGetLastError();
}

}  // namespace

template <typename TT>
static int launch_link_rank(const jmac_link_layer_t* layers, int32_t n_layers, const int32_t* h, const int32_t* r, int32_t pred_head,
                            const int32_t* gold, const int32_t* filt_ptr, const int32_t* filt_idx, int64_t B, int64_t N, int64_t d,
                            int32_t* rank, void* ws, size_t ws_bytes, hipStream_t st) {
    if (B < 0 || N <= 0 || d <= 0 || n_layers <= 0 || n_layers > LR_MAX_LAYERS) return JMAC_EINVAL;
    if (B == 0) return JMAC_OK;
    if (!layers || !h || !r || !gold || !rank || (filt_ptr && !filt_idx)) return JMAC_EINVAL;
    if (B >= INT32_MAX || N >= INT32_MAX || d > 512) return JMAC_ERANGE;
    if (!ws || ws_bytes < jmac_linkpred_rank_workspace_bytes(B, d, n_layers)) return JMAC_EWORKSPACE;
    LinkRankArgs a{};
    a.nl = n_layers; a.B = (int32_t)B; a.N = (int32_t)N; a.d = (int32_t)d; a.dq = (int32_t)((d + 3) / 4 * 4);
    bool vec = d % 4 == 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!layers[l].ent || !layers[l].rel || !layers[l].table) return JMAC_EINVAL;
        a.ent[l] = layers[l].ent; a.rel[l] = layers[l].rel; a.tab[l] = layers[l].table;
        a.ld_ent[l] = layers[l].ld_ent; a.ld_rel[l] = layers[l].ld_rel; a.ld_tab[l] = layers[l].ld_table;
        // the vector form issues 4-element loads (16 B fp32, 8 B bf16): leading dimension AND base pointer must allow it
        if (a.ld_tab[l] % 4 || ((uintptr_t)a.tab[l] % (4 * sizeof(TT)))) vec = false;
    }
    a.sign = pred_head ? -1.f : 1.f;
    a.h = h; a.r = r; a.gold = gold; a.filt_ptr = filt_ptr; a.filt_idx = filt_idx;
    a.gs = (float*)ws;
    a.er = (char*)ws + align_up((size_t)B * 4);
    a.rank = rank;
    const int64_t W = (int64_t)n_layers * a.dq, words = 60 * 1024 / 4;
    int64_t rows = (words - W) / (W + 1);
    if (rows < 1) return JMAC_ERANGE;
    a.rows = (int32_t)(rows < LR_ROWS ? rows : LR_ROWS);
    const size_t shm = (size_t)(W + a.rows * (W + 1)) * sizeof(float);
    hipLaunchKernelGGL((link_rank_prep_kernel<TT>), dim3((unsigned)B), dim3(kBlock), shm, st, a);
    dim3 grid((unsigned)((N + L1_T - 1) / L1_T), (unsigned)((B + L1_T - 1) / L1_T));
    if (vec) hipLaunchKernelGGL((link_rank_tile_kernel<TT, true>), grid, dim3(kBlock), 0, st, a);
    else hipLaunchKernelGGL((link_rank_tile_kernel<TT, false>), grid, dim3(kBlock), 0, st, a);
    return (int)hipGetLastError();
}

extern "C" {

int jmac_l1_score_f32(const float* er, int64_t lder, const float* table, int64_t ldt, int64_t B, int64_t N, int64_t d,
                      float* out, int64_t ldout, int32_t accumulate, jmac_stream_t stream) {
    if (B < 0 || N < 0 || d <= 0) return JMAC_EINVAL;
    if (B == 0 || N == 0) return JMAC_OK;
    if (!er || !table || !out) return JMAC_EINVAL;
    if (lder % 4 || ldt % 4) return JMAC_EDIM;
    if (B >= INT32_MAX || N >= INT32_MAX) return JMAC_ERANGE;
    dim3 grid((unsigned)((N + L1_T - 1) / L1_T), (unsigned)((B + L1_T - 1) / L1_T));
    // 4-element vector loads need 4-element-aligned base pointers too (a column-offset view is not): scalar form otherwise
    const bool aligned = (((uintptr_t)er | (uintptr_t)table) % (4 * sizeof(float))) == 0;
    if (d % 4 == 0 && aligned)
        hipLaunchKernelGGL((l1_score_kernel<float, true>), grid, dim3(kBlock), 0, (hipStream_t)stream, er, lder, table, ldt, (int)B,
                           (int)N, (int)d, out, ldout, accumulate);
    else
        hipLaunchKernelGGL((l1_score_kernel<float, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, er, lder, table, ldt, (int)B,
                           (int)N, (int)d, out, ldout, accumulate);
    return (int)hipGetLastError();
}

int jmac_l1_score_bf16(const uint16_t* er, int64_t lder, const uint16_t* table, int64_t ldt, int64_t B, int64_t N, int64_t d,
                       float* out, int64_t ldout, int32_t accumulate, jmac_stream_t stream) {
    if (B < 0 || N < 0 || d <= 0) return JMAC_EINVAL;
    if (B == 0 || N == 0) return JMAC_OK;
    if (!er || !table || !out) return JMAC_EINVAL;
    if (lder % 4 || ldt % 4) return JMAC_EDIM;
    if (B >= INT32_MAX || N >= INT32_MAX) return JMAC_ERANGE;
    dim3 grid((unsigned)((N + L1_T - 1) / L1_T), (unsigned)((B + L1_T - 1) / L1_T));
    // 4-element vector loads need 4-element-aligned base pointers too (a column-offset view is not): scalar form otherwise
    const bool aligned = (((uintptr_t)er | (uintptr_t)table) % (4 * sizeof(uint16_t))) == 0;
    if (d % 4 == 0 && aligned)
        hipLaunchKernelGGL((l1_score_kernel<bf16_t, true>), grid, dim3(kBlock), 0, (hipStream_t)stream, er, lder, table, ldt, (int)B,
                           (int)N, (int)d, out, ldout, accumulate);
    else
        hipLaunchKernelGGL((l1_score_kernel<bf16_t, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, er, lder, table, ldt,
                           (int)B, (int)N, (int)d, out, ldout, accumulate);
    return (int)hipGetLastError();
}

int jmac_filtered_rank_f32(const float* score, int64_t lds, const int32_t* gold, const int32_t* filt_ptr,
                           const int32_t* filt_idx, int64_t B, int64_t N, int32_t descending, int32_t* rank,
                           jmac_stream_t stream) {
    if (B < 0 || N <= 0) return JMAC_EINVAL;
    if (B == 0) return JMAC_OK;
    if (!score || !gold || !rank || (filt_ptr && !filt_idx && false)) return JMAC_EINVAL;
    if (N >= INT32_MAX) return JMAC_ERANGE;
    hipLaunchKernelGGL(filtered_rank_kernel, dim3((unsigned)B), dim3(kBlock), 0, (hipStream_t)stream, score, lds, gold, filt_ptr,
                       filt_idx, (int)N, descending ? 1 : 0, rank);
    return (int)hipGetLastError();
}

size_t jmac_linkpred_rank_workspace_bytes(int64_t B, int64_t d, int32_t n_layers) {
    if (B < 0 || d <= 0 || n_layers <= 0) return 0;
    return align_up((size_t)B * 4) + align_up((size_t)n_layers * (size_t)B * (size_t)((d + 3) / 4 * 4) * 4) + 256;
}

int jmac_linkpred_rank_f32(const jmac_link_layer_t* layers, int32_t n_layers, const int32_t* h, const int32_t* r,
                           int32_t pred_head, const int32_t* gold, const int32_t* filt_ptr, const int32_t* filt_idx, int64_t B,
                           int64_t N, int64_t d, int32_t* rank, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    return launch_link_rank<float>(layers, n_layers, h, r, pred_head, gold, filt_ptr, filt_idx, B, N, d, rank, ws, ws_bytes,
                                   (hipStream_t)stream);
}

int jmac_linkpred_rank_bf16(const jmac_link_layer_t* layers, int32_t n_layers, const int32_t* h, const int32_t* r,
                            int32_t pred_head, const int32_t* gold, const int32_t* filt_ptr, const int32_t* filt_idx, int64_t B,
                            int64_t N, int64_t d, int32_t* rank, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    return launch_link_rank<bf16_t>(layers, n_layers, h, r, pred_head, gold, filt_ptr, filt_idx, B, N, d, rank, ws, ws_bytes,
                                    (hipStream_t)stream);
}

int jmac_sim_matrix_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t d, float* C,
                        int64_t ldc, jmac_stream_t stream) {
    if (M < 0 || N < 0 || d <= 0) return JMAC_EINVAL;
    if (M == 0 || N == 0) return JMAC_OK;
    if (!A || !B || !C) return JMAC_EINVAL;
    if (lda % 4 || ldb % 4) return JMAC_EDIM;
    if (M >= INT32_MAX || N >= INT32_MAX) return JMAC_ERANGE;
    return launch_sim(A, lda, B, ldb, M, N, d, C, ldc, (hipStream_t)stream);
}

// fused similarity + running top-k: used when the matrix is wide enough for the column sample to pay
constexpr int ST_CAP = TK_CAP;            // candidates kept per row
constexpr int ST_KMAX = 64;
constexpr int64_t ST_MIN_N = 8192;
static inline bool st_fused(int64_t N, int64_t k) { return N >= ST_MIN_N && k <= ST_KMAX; }
static inline int64_t st_sample(int64_t N) {
    int64_t ns = N / 12 > 2048 ? N / 12 : 2048;            // expected candidates per row ~ k * N / Ns <= 12k (cap: 1024)
    return (ns + 127) / 128 * 128;
}
struct StWs { size_t s0, val0, idx0, cnt, cval, cidx, total; };
static StWs st_layout(int64_t L, int64_t N, int64_t k) {
    StWs w{};
    size_t off = 0;
    w.s0 = off;   off += align_up((size_t)L * (size_t)st_sample(N) * 4);
    w.val0 = off; off += align_up((size_t)L * (size_t)k * 4);
    w.idx0 = off; off += align_up((size_t)L * (size_t)k * 4);
    w.cnt = off;  off += align_up((size_t)L * 4);
    w.cval = off; off += align_up((size_t)L * ST_CAP * 4);
    w.cidx = off; off += align_up((size_t)L * ST_CAP * 4);
    w.total = off + 256;
    return w;
}

size_t jmac_sim_topk_workspace_bytes(int64_t L, int64_t N, int32_t k) {
    if (L < 0 || N < 0 || k <= 0) return 0;
    if (st_fused(N, k)) return st_layout(L, N, k).total;      // sample scores + candidate lists: no L x N matrix
    return align_up((size_t)L * (size_t)N * 4) + 256;
}

int jmac_row_topk_f32(const float* S, int64_t lds, int64_t L, int64_t N, int32_t k, float* val, int32_t* idx,
                      jmac_stream_t stream) {
    if (L < 0 || N <= 0 || k <= 0 || k > N) return JMAC_EINVAL;
    if (L == 0) return JMAC_OK;
    if (!S || !idx) return JMAC_EINVAL;
    return launch_topk(S, lds, L, N, k, val, idx, (hipStream_t)stream);
}

int jmac_sim_topk_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t L, int64_t N, int64_t d, int32_t k,
                      float* val, int32_t* idx, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    if (L < 0 || N <= 0 || d <= 0 || k <= 0 || k > N) return JMAC_EINVAL;
    if (L == 0) return JMAC_OK;
    if (!A || !B || !idx) return JMAC_EINVAL;
    if (lda % 4 || ldb % 4) return JMAC_EDIM;
    if (!ws || ws_bytes < jmac_sim_topk_workspace_bytes(L, N, k)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!st_fused(N, k)) {                                   // narrow matrices: scores to the workspace, then the row pass
        float* S = (float*)ws;
        if (int rc = launch_sim(A, lda, B, ldb, L, N, d, S, N, st)) return rc;
        return launch_topk(S, N, L, N, k, val, idx, st);
    }
    // 1. tau[m] = k-th largest score of row m among the first Ns columns: a lower bound of the row's final k-th score
    const StWs w = st_layout(L, N, k);
    char* wb = (char*)ws;
    const int64_t Ns = st_sample(N);
    float* S0 = (float*)(wb + w.s0);
    float* val0 = (float*)(wb + w.val0);
    if (int rc = launch_sim(A, lda, B, ldb, L, Ns, d, S0, Ns, st)) return rc;
    if (int rc = launch_topk(S0, Ns, L, Ns, k, val0, (int32_t*)(wb + w.idx0), st)) return rc;
    // 2. the product over the REMAINING columns with the filtering epilogue: candidates instead of the matrix
    SimFilter f{};
    f.tau = val0 + (k - 1); f.tau_stride = k;
    f.cnt = (int*)(wb + w.cnt); f.cval = (float*)(wb + w.cval); f.cidx = (int*)(wb + w.cidx); f.cap = ST_CAP;
    f.n_off = (int)Ns;
    if (hipMemsetAsync(f.cnt, 0, (size_t)L * 4, st) != hipSuccess) return (int)hipGetLastError();
    if (int rc = launch_sim(A, lda, B + Ns * ldb, ldb, L, N - Ns, d, nullptr, 0, st, &f)) return rc;
    // 3. the k best of every candidate list (+ the sample's k best)
    hipLaunchKernelGGL(cand_select_kernel, dim3((unsigned)L), dim3(kBlock), 0, st, A, lda, B, ldb, (int)N, (int)d, (int)k, f.cnt, f.cval,
                       f.cidx, ST_CAP, val0, (const int32_t*)(wb + w.idx0), val, idx);
    return (int)hipGetLastError();
}

size_t jmac_softmax_entropy_workspace_bytes(int64_t n1, int64_t n2) {
    if (n1 < 0 || n2 < 0) return 0;
    return align_up((size_t)n1 * (size_t)n2 * 4) + jmac_col_softmax_workspace_bytes(n1, n2) + 256;
}

int jmac_softmax_entropy_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t n1, int64_t n2, int64_t d,
                             float scale, float* ent_rows, float* ent_cols, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    if (n1 < 0 || n2 < 0 || d <= 0) return JMAC_EINVAL;
    if (n1 == 0 || n2 == 0) return JMAC_OK;
    if (!A || !B || !ent_rows || !ent_cols) return JMAC_EINVAL;
    if (lda % 4 || ldb % 4) return JMAC_EDIM;
    if (!ws || ws_bytes < jmac_softmax_entropy_workspace_bytes(n1, n2)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // ONE GEMM: the column entropies come from a column-wise pass over the same row-major S (the reference's
    // softmax(simi.t()) of train.py:245 reads the transpose of the same matrix, not a second product)
    float* S = (float*)ws;
    const size_t soff = align_up((size_t)n1 * (size_t)n2 * 4);
    if (int rc = launch_sim(A, lda, B, ldb, n1, n2, d, S, n2, st)) return rc;
    hipLaunchKernelGGL(row_entropy_kernel, dim3((unsigned)n1), dim3(kBlock), 0, st, S, n2, (int)n2, scale, ent_rows);
    return jmac_col_softmax_f32(S, n2, n1, n2, nullptr, nullptr, 0.f, scale, nullptr, 0, ent_cols, (char*)ws + soff,
                                ws_bytes - soff, stream);
}

int jmac_csls_apply_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, const float* r1, const float* r2, float* out,
                        int64_t ldo, jmac_stream_t stream) {
    if (n1 < 0 || n2 < 0) return JMAC_EINVAL;
    if (n1 == 0 || n2 == 0) return JMAC_OK;
    if (!S || !r1 || !r2 || !out) return JMAC_EINVAL;
    int64_t blocks = (n1 * n2 + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(csls_apply_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, S, lds, n1, n2, r1, r2, out,
                       ldo);
    return (int)hipGetLastError();
}

int jmac_row_softmax_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, const uint8_t* row_mask,
                         const uint8_t* col_mask, float fill, float scale, float* out, int64_t ldo, float* ent,
                         jmac_stream_t stream) {
    if (n1 < 0 || n2 <= 0) return JMAC_EINVAL;
    if (n1 == 0) return JMAC_OK;
    if (!S || (!out && !ent)) return JMAC_EINVAL;
    if (n2 >= INT32_MAX) return JMAC_ERANGE;
    hipLaunchKernelGGL(masked_row_softmax_kernel, dim3((unsigned)n1), dim3(kBlock), 0, (hipStream_t)stream, S, lds, (int)n2,
                       row_mask, col_mask, fill, scale, out, ldo, ent);
    return (int)hipGetLastError();
}

int jmac_masked_row_softmax_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, const uint8_t* row_mask,
                                const uint8_t* col_mask, float fill, float scale, float* out, int64_t ldo,
                                jmac_stream_t stream) {
    if (!out) return JMAC_EINVAL;
    return jmac_row_softmax_f32(S, lds, n1, n2, row_mask, col_mask, fill, scale, out, ldo, nullptr, stream);
}

static int col_softmax_row_splits(int64_t n1, int64_t n2) {
    const int64_t strips = (n2 + 63) / 64;
    int64_t rs = (2048 + strips - 1) / strips;            // ~2k blocks
    const int64_t max_rs = (n1 + 63) / 64;                // at least 64 rows per block
    if (rs > max_rs) rs = max_rs;
    if (rs < 1) rs = 1;
    return (int)rs;
}

size_t jmac_col_softmax_workspace_bytes(int64_t n1, int64_t n2) {
    if (n1 < 0 || n2 < 0) return 0;
    return align_up((size_t)col_softmax_row_splits(n1, n2) * (kBlock / 64) * (size_t)n2 * 12) + 2 * align_up((size_t)n2 * 4) + 256;
}

int jmac_col_softmax_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, const uint8_t* row_mask,
                         const uint8_t* col_mask, float fill, float scale, float* out_t, int64_t ldo, float* ent,
                         void* ws, size_t ws_bytes, jmac_stream_t stream) {
    if (n1 <= 0 || n2 < 0) return JMAC_EINVAL;
    if (n2 == 0) return JMAC_OK;
    if (!S || (!out_t && !ent)) return JMAC_EINVAL;
    if (n1 >= INT32_MAX || n2 >= INT32_MAX) return JMAC_ERANGE;
    if (!ws || ws_bytes < jmac_col_softmax_workspace_bytes(n1, n2)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int rs = col_softmax_row_splits(n1, n2);
    const int parts = rs * (kBlock / 64);
    float* part = (float*)ws;
    float* cmax = (float*)((char*)ws + align_up((size_t)parts * (size_t)n2 * 12));
    float* cinv = (float*)((char*)cmax + align_up((size_t)n2 * 4));
    hipLaunchKernelGGL(col_softmax_stats_kernel, dim3((unsigned)((n2 + 63) / 64), (unsigned)rs), dim3(kBlock), 0, st, S, lds,
                       (int)n1, (int)n2, row_mask, col_mask, fill, scale, part);
    hipLaunchKernelGGL(col_softmax_merge_kernel, dim3((unsigned)((n2 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, part, parts,
                       (int)n2, cmax, cinv, ent);
    if (out_t)
        hipLaunchKernelGGL(col_softmax_apply_kernel, dim3((unsigned)((n2 + 63) / 64), (unsigned)((n1 + 63) / 64)), dim3(kBlock), 0, st,
                           S, lds, (int)n1, (int)n2, row_mask, col_mask, fill, scale, cmax, cinv, out_t, ldo);
    return (int)hipGetLastError();
}

static int col_topk_row_splits(int64_t n1, int64_t n2) {
    const int64_t strips = (n2 + 63) / 64;
    int64_t rs = (2048 + strips - 1) / strips;            // ~2k blocks
    const int64_t max_rs = (n1 + 63) / 64;                // at least 64 rows per block
    if (rs > max_rs) rs = max_rs;
    if (rs < 1) rs = 1;
    return (int)rs;
}

size_t jmac_col_topk_workspace_bytes(int64_t n1, int64_t n2, int32_t k) {
    if (n1 < 0 || n2 < 0 || k <= 0) return 0;
    return align_up((size_t)col_topk_row_splits(n1, n2) * (kBlock / 64) * (size_t)n2 * (size_t)k * 4) + 256;
}

int jmac_col_topk_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, int32_t k, float* val, void* ws, size_t ws_bytes,
                      jmac_stream_t stream) {
    if (n1 <= 0 || n2 <= 0 || k <= 0 || k > n1) return JMAC_EINVAL;
    if (k > CT_KMAX) return JMAC_EDIM;
    if (!S || !val) return JMAC_EINVAL;
    if (n1 >= INT32_MAX || n2 >= INT32_MAX) return JMAC_ERANGE;
    if (!ws || ws_bytes < jmac_col_topk_workspace_bytes(n1, n2, k)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int rs = col_topk_row_splits(n1, n2);
    float* cand = (float*)ws;
    hipLaunchKernelGGL(col_topk_partial_kernel, dim3((unsigned)((n2 + 63) / 64), (unsigned)rs), dim3(kBlock), 0, st, S, lds, (int)n1,
                       (int)n2, (int)k, cand);
    hipLaunchKernelGGL(col_topk_merge_kernel, dim3((unsigned)((n2 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, cand,
                       rs * (kBlock / 64), (int)n2, (int)k, val);
    return (int)hipGetLastError();
}

int jmac_csls_rank_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, const float* r1, const float* r2, const int32_t* gold,
                       int32_t* rank, jmac_stream_t stream) {
    if (n1 < 0 || n2 <= 0) return JMAC_EINVAL;
    if (n1 == 0) return JMAC_OK;
    if (!S || !r1 || !r2 || !gold || !rank) return JMAC_EINVAL;
    if (n2 >= INT32_MAX) return JMAC_ERANGE;
    hipLaunchKernelGGL(csls_rank_kernel, dim3((unsigned)n1), dim3(kBlock), 0, (hipStream_t)stream, S, lds, (int)n2, r1, r2, gold, rank);
    return (int)hipGetLastError();
}

}  // extern "C"
