// gemm.hip -- small fp32 GEMM for the relation-side projections of the layer (gfx950).
//
// The reference transforms the relation table with dense products on ~10^3 rows: rel @ W1, act(.) @ W2
// (src/jmac_model.py:40-42), the hoisted rel'' @ [Wb|Wg] of the factorised layer, and JMAC's relation MLPs
// (src/jmac_model.py:195-196) -- [962,300] x [300,300..600] at DBP-5L size, 36 of them per training step with their
// backward forms.  A library GEMM spends ~17 us on each (24 workgroups of a 256-CU chip); here one 8-wave block owns
// a 32x32 output tile: both operand panels are staged through LDS in one coalesced round trip, the waves split K and the
// partial tiles are summed through LDS in wave order (deterministic).  v_mfma_f32_32x32x2_f32: exact fp32 products,
// fp32 accumulation.  Products that do not depend on each other share one launch (jmac_gemm_grouped_f32).
//
// C[M,N] = op(A) op(B), row-major, op = identity or transpose: the three forms autograd needs
//   forward  C = A B           (NN)        dA = G B^T  (NT)        dB = A^T G  (TN)
#include "common.h"

using namespace jmac;

namespace {

#ifndef JMAC_GG_WAVES
#define JMAC_GG_WAVES 8
#endif
#ifndef JMAC_GG_KC
#define JMAC_GG_KC 304
#endif
constexpr int kWaves = JMAC_GG_WAVES;  // waves per 32x32 tile = K split factor
constexpr int kBlock = 64 * kWaves;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Fragment of one 8-deep K step for the 32x32x2 MFMA: lane (r = lane & 31, h = lane >> 5) holds the operand's
// values for row/column r and k = k0 + 4h + s, s = 0..3; MFMA step s then contracts k = {k0 + s, k0 + 4 + s}.
// KC = true: the operand is stored with k contiguous (row r is a memory row): one 16-byte load.
// KC = false: stored with r contiguous (k selects the memory row): four 4-byte loads, each coalesced over r.

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ---- grouped form: many independent small products in ONE launch ------------------------------------------------------
// The relation side of a training step is ~40 products of ~10^3 rows (src/jmac_model.py:39-42 per layer, :195-196, the
// hoisted R''[Wb|Wg], and their backward forms).  Each is far too small to fill the chip, so as separate launches they
// cost a launch + a latency chain each (~9 us, whatever their size).  Products with no dependence on each other are
// issued together: the task table travels in the kernel arguments (by value: safe under stream capture), a block finds its
// task by a scan over the table's tile prefix, and each task carries its own operand forms and epilogue:
//   act on the output (LeakyReLU / ReLU between the two relation transforms), multiplication by act'(saved output) (the
//   backward of that activation, fused into the product that produces its input gradient), accumulation into C, an
//   operand whose last rows live in a second buffer (cat(rel_emb, loop_rel), src/jmac_model.py:39, without the cat) and
//   an output whose last rows go to a second buffer (its adjoint).
struct GOperand {
    const float* p;
    const float* p2;       // memory rows >= split come from p2 (row index - split)
    int64_t ld;
    int32_t split;
    __device__ __forceinline__ const float* row(int i) const {
        return i < split ? p + (int64_t)i * ld : p2 + (int64_t)(i - split) * ld;
    }
};
struct GTask {
    GOperand A, B;
    float* C;
    float* C2;             // output rows >= c_split go to C2
    const float* mask;     // act' source (same shape as C)
    int64_t ldc, ldmask;
    int32_t M, N, K, c_split;
    int32_t ta, tb, vec, act, accumulate, big;     // big: 64x64 tiles (grouped_tile64)
    float slope;
    int32_t tiles_n, tile_begin;       // tile_begin: the task's tile count (grid.x bound)
};
constexpr int kMaxTasks = JMAC_GEMM_MAX_TASKS;
struct GTable {
    GTask t[kMaxTasks];
};

// ---- one 32x32 output tile per 8-wave block ------------------------------------------------------------------------------
// Both operand panels of the tile (32 x Kc each, Kc <= kKc = 304) are staged through LDS with fully coalesced 16-byte
// loads, all of them in flight together: ONE global round trip per K chunk (K <= 304 -- every forward product at d <= 304 --
// is a single chunk).  The eight waves then split the chunk's 8-deep K steps, read their MFMA fragments from LDS and the
// partial tiles are summed through LDS in wave order.  (The first form of this kernel loaded fragments straight from L2
// into registers: 32 rows x 32 bytes per wave instruction, four dependent-latency rounds per wave: 12 us for one
// 962x300x300 product, 41 us for a level of four.)
//
// Panel forms in LDS, chosen by how the operand is stored:
//   KC ("k contiguous": A not transposed, B transposed)  lds[r * kLdk + k], r < 32: fragment = one ds_read_b128 of k0+4h..+3
//       (kLdk = 308 = 4 * 77: rows 4*odd banks apart -> the 16-lane groups of a b128 read hit 64 distinct banks)
//   RC ("r contiguous": A transposed, B not transposed)  lds[k * 32 + r]: fragment = four ds_read_b32, lanes r consecutive
// Fragment of one 8-deep K step for the 32x32x2 MFMA: lane (r = lane & 31, h = lane >> 5) holds the operand's values for
// row/column r and k = k0 + 4h + s, s = 0..3; MFMA step s contracts k = {k0 + s, k0 + 4 + s}.
constexpr int kKc = JMAC_GG_KC;                // K chunk (38 steps of 8)
constexpr int kLdk = kKc + 4;                  // row pitch of a KC panel (floats): 4 * odd
constexpr int kPanel = 32 * kLdk;              // floats per panel (KC form; the RC form needs kKc * 32 <= this)
constexpr int kStage = (32 * (kKc / 4) + kBlock - 1) / kBlock;      // float4s per thread per panel = 5
constexpr int kSteps = (kKc / 8 + kWaves - 1) / kWaves;             // K steps per wave per chunk = 5
constexpr size_t kGroupedLds = (size_t)2 * kPanel * sizeof(float);  // 78 848 B: two blocks per CU

// The task's pointers come out of the kernarg copy as GENERIC pointers: without the casts below every access is a flat_*
// instruction (aperture check, both vmcnt and lgkmcnt) instead of global_*.
typedef const float __attribute__((address_space(1))) * gcf_t;
typedef float __attribute__((address_space(1))) * gf_t;
typedef const jmac_f32x4 __attribute__((address_space(1))) * gcf4_t;
__device__ __forceinline__ float gld(const float* p) { return *(gcf_t)p; }
__device__ __forceinline__ float4 gld4(const float* p) {
    const jmac_f32x4 v = *(gcf4_t)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gst(float* p, float v) { *(gf_t)p = v; }

// global -> registers: kStage float4s of one panel (unconditional loads at clamped addresses; zeros past K / past the chunk)
template <bool KC>
__device__ __forceinline__ void panel_load(const GOperand& o, int r0, int R, int k0, int K, float4 (&v)[kStage]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
        const int f = tid + i * kBlock;
        if (KC) {                                              // f -> (row r, quad q): 76 quads per row
            const int r = min(f / (kKc / 4), 31), q = f % (kKc / 4);
            const int k = k0 + 4 * q;
            v[i] = gld4(o.row(min(r0 + r, R - 1)) + max(min(k, K - 4), 0));
            if (k >= K) v[i] = f4zero();
        } else {                                               // f -> (k row kk, quad q of the 32 columns)
            const int kk = min(f >> 3, kKc - 1), q = f & 7;
            const int k = k0 + kk;
            v[i] = gld4(o.row(min(k, K - 1)) + min(r0 + 4 * q, R - 4));
            if (k >= K) v[i] = f4zero();
        }
    }
}

// operands that do not allow 16-byte loads (odd K, unaligned rows, a ragged last column tile of an RC panel): element by
// element, straight into LDS, in a rolled loop (this path is about correctness for any shape, not speed)
template <bool KC>
__device__ __forceinline__ void panel_stage_scalar(const GOperand& o, int r0, int R, int k0, int K, float* __restrict__ lds) {
#pragma unroll 1
    for (int f = threadIdx.x; f < 32 * kKc; f += kBlock) {
        const int r = KC ? f / kKc : f % 32, kk = KC ? f % kKc : f / 32;
        const int k = k0 + kk, rr = min(r0 + r, R - 1), kc = min(k, K - 1);
        const float x = KC ? gld(o.row(rr) + kc) : gld(o.row(kc) + rr);
        lds[KC ? r * kLdk + kk : kk * 32 + r] = k < K ? x : 0.f;
    }
}

template <bool KC>
__device__ __forceinline__ void panel_store(float* __restrict__ lds, const float4 (&v)[kStage]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
        const int f = tid + i * kBlock;
        if (KC) {
            if (f < 32 * (kKc / 4)) st4(lds + (f / (kKc / 4)) * kLdk + 4 * (f % (kKc / 4)), v[i]);
        } else {
            if (f < kKc * 8) st4(lds + (f >> 3) * 32 + 4 * (f & 7), v[i]);
        }
    }
}

template <bool KC>
__device__ __forceinline__ float4 panel_frag(const float* __restrict__ lds, int r, int kk) {
    if (KC) return ld4(lds + r * kLdk + kk);
    return make_float4(lds[(kk + 0) * 32 + r], lds[(kk + 1) * 32 + r], lds[(kk + 2) * 32 + r], lds[(kk + 3) * 32 + r]);
}

// VA / VB: 16-byte global loads are legal for that operand in THIS tile (alignment, leading dimension, K % 4 for the KC form,
// a full 32-column tile for the RC form)
// MULTI: K spans several chunks -> the next chunk's global loads are prefetched into registers during the MFMAs;
// single-chunk products (every forward product at d <= 304) prefetch all of a wave's LDS fragments instead.
#ifdef JMAC_GG_TRACE
// debug builds (tools/gg_trace.py): per-block timestamps of the phases, written behind the output of task 0's C2 pointer
#define GG_STAMP(i) do { if (threadIdx.x == 0) { gg_trace_buf[((int64_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 8 + (i)] = __builtin_readcyclecounter(); if ((i) == 0 || (i) == 4) gg_trace_buf[((int64_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 8 + ((i) == 0 ? 5 : 6)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
__device__ unsigned long long* gg_trace_buf;
#else
#define GG_STAMP(i)
#endif

template <bool TA, bool TB, bool VA, bool VB, bool MULTI>
__device__ __forceinline__ void grouped_tile(const GTask& t, int tile, float* __restrict__ lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = (tile / t.tiles_n) * 32, n0 = (tile % t.tiles_n) * 32;
    const int M = t.M, N = t.N, K = t.K;
    float* la = lds;
    float* lb = lds + kPanel;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // chunk loop: the NEXT chunk's global loads are issued before this chunk's MFMAs (registers), so a long-K product
    // (the weight gradients: K = the ~10^3 relation rows, four chunks) pays one exposed round trip, not four
    GG_STAMP(1);
    float4 va[kStage], vb[kStage];
    if constexpr (VA) panel_load<!TA>(t.A, m0, M, 0, K, va);            // 10 loads per thread in flight
    if constexpr (VB) panel_load<TB>(t.B, n0, N, 0, K, vb);
    for (int k0 = 0; k0 < K; k0 += kKc) {
        if (k0 > 0) __syncthreads();                           // the previous chunk's fragments have been read
        if constexpr (VA) panel_store<!TA>(la, va);
        else panel_stage_scalar<!TA>(t.A, m0, M, k0, K, la);
        if constexpr (VB) panel_store<TB>(lb, vb);
        else panel_stage_scalar<TB>(t.B, n0, N, k0, K, lb);
        // UNCONDITIONAL (a load issued inside a branch is waited for before the branch is left): past the last chunk the
        // clamped addresses hit lines this block has just read and the values are never stored
        if constexpr (MULTI) {
            if constexpr (VA) panel_load<!TA>(t.A, m0, M, k0 + kKc, K, va);
            if constexpr (VB) panel_load<TB>(t.B, n0, N, k0 + kKc, K, vb);
        }
        __syncthreads();
        GG_STAMP(2);
        // wave w takes steps w, w + 8, ... (<= kSteps of them): all their fragments are read first (one LDS latency), the
        // dependent MFMA chain follows; steps past the chunk read a clamped address and are skipped (wave-uniform test)
        const int steps = (min(K - k0, kKc) + 7) / 8;          // <= 38
        if constexpr (!MULTI) {
            float4 fa[kSteps], fb[kSteps];
#pragma unroll
            for (int j = 0; j < kSteps; ++j) {
                const int kk = min(wave + j * kWaves, kKc / 8 - 1) * 8 + 4 * h;
                fa[j] = panel_frag<!TA>(la, r, kk);
                fb[j] = panel_frag<TB>(lb, r, kk);
            }
#pragma unroll
            for (int j = 0; j < kSteps; ++j) {
                if (wave + j * kWaves < steps) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].x, fb[j].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].y, fb[j].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].z, fb[j].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j].w, fb[j].w, acc, 0, 0, 0);
                }
            }
        } else {
            float4 a = panel_frag<!TA>(la, r, min(wave, kKc / 8 - 1) * 8 + 4 * h), b = panel_frag<TB>(lb, r, min(wave, kKc / 8 - 1) * 8 + 4 * h);
#pragma unroll
            for (int j = 0; j < kSteps; ++j) {
                const int kn = min(wave + (j + 1) * kWaves, kKc / 8 - 1) * 8 + 4 * h;
                const float4 an = panel_frag<!TA>(la, r, kn), bn = panel_frag<TB>(lb, r, kn);      // one step ahead
                if (wave + j * kWaves < steps) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                }
                a = an;
                b = bn;
            }
        }
    }
    GG_STAMP(3);
    __syncthreads();                                           // panels are dead: their memory carries the partial tiles
    // every wave leaves its 16 accumulator registers in LDS; wave w then finishes registers 2w and 2w + 1 of the tile: the
    // eight partials are summed in wave order (bitwise reproducible), the epilogue's own loads (act' source, accumulate)
    // are issued together and unconditionally at clamped addresses, the two stores come last with nothing waiting on them
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(lds);
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    const int n = n0 + r, nc = min(n, N - 1);
    constexpr int kRegs = 16 / kWaves;
    float v[kRegs], msk[kRegs], old[kRegs];
    float* cp[kRegs];
    const float* mp[kRegs];
    bool ok[kRegs], acc_ok[kRegs];
#pragma unroll
    for (int j = 0; j < kRegs; ++j) {
        msk[j] = old[j] = 0.f;
        const int i = kRegs * wave + j;
        float sum = red[0][i][lane];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) sum += red[w][i][lane];
        v[j] = sum;
        const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h, mc = min(m, M - 1);
        ok[j] = m < M && n < N;
        cp[j] = mc < t.c_split ? t.C + (int64_t)mc * t.ldc + nc : t.C2 + (int64_t)(mc - t.c_split) * t.ldc + nc;
        acc_ok[j] = mc < t.c_split;                            // accumulate is a property of C; C2 rows are always stored
        mp[j] = t.mask + (int64_t)mc * t.ldmask + nc;
    }
    // block-uniform branches: a task without these epilogue inputs pays no round trip for them
    if (t.act == JMAC_GEMM_DACT_LEAKY || t.act == JMAC_GEMM_DACT_RELU) {
#pragma unroll
        for (int j = 0; j < kRegs; ++j) msk[j] = gld(mp[j]);
    }
    if (t.accumulate) {
#pragma unroll
        for (int j = 0; j < kRegs; ++j) old[j] = gld(cp[j]);
    }
#pragma unroll
    for (int j = 0; j < kRegs; ++j) {
        float x = v[j];
        if (t.act == JMAC_GEMM_ACT_LEAKY) x = x > 0.f ? x : x * t.slope;
        else if (t.act == JMAC_GEMM_ACT_RELU) x = x > 0.f ? x : 0.f;
        else if (t.act == JMAC_GEMM_DACT_LEAKY) x = msk[j] > 0.f ? x : x * t.slope;
        else if (t.act == JMAC_GEMM_DACT_RELU) x = msk[j] > 0.f ? x : 0.f;
        if (t.accumulate && acc_ok[j]) x += old[j];
        v[j] = x;
    }
#pragma unroll
    for (int j = 0; j < kRegs; ++j)
        if (ok[j]) gst(cp[j], v[j]);
    GG_STAMP(4);
}

// ---- one 64x64 output tile per 8-wave block: the products on LONG relation tables (the 5-KG union: ~4 800 rows) -------------
// A launch's time is rounds of resident blocks x a block's life.  At ~10^3 rows a level's 32x32 tiles are a few rounds and the
// form below changes nothing (measured: forward levels 26.6 / 24.4 / 25.2 -> 23.4 / 22.8 / 23.4 us, single products 10.4 ->
// 13.6 us); at ~5 000 rows they are 17 rounds, and a 64x64 tile feeds four times the MFMAs from the same panel bytes per K
// chunk.  Waves: q = wave & 3 picks the 32x32 quadrant, kh = wave >> 2 the K steps of the chunk (even / odd); K chunks of 80
// (2 x 64 x 84 floats = 43 KB), the next chunk's global loads in flight across the MFMAs; the two K halves are summed through
// LDS, half 0 first.  A is not transposed here (KC panel); B either way; vector loads only.
constexpr int kKc2 = 80;
constexpr int kLdk2 = kKc2 + 4;                 // 84 = 4 * 21
constexpr int kPanel2 = 64 * kLdk2;             // floats per panel (the RC form needs kKc2 * 64 <= this)
constexpr int kStage2 = (64 * (kKc2 / 4) + kBlock - 1) / kBlock;    // float4s per thread per panel = 3
constexpr size_t kGroupedLds2 = (size_t)2 * kPanel2 * sizeof(float);
static_assert(kWaves == 8, "the 64x64 form is laid out for eight waves");
static_assert(kKc2 * 64 <= kPanel2 && kGroupedLds2 <= kGroupedLds, "64x64 panels fit the launch's LDS");

template <bool KC>
__device__ __forceinline__ void panel2_load(const GOperand& o, int r0, int R, int k0, int K, float4 (&v)[kStage2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < kStage2; ++i) {
        const int f = tid + i * kBlock;
        if (KC) {                                              // f -> (row r, quad q): 20 quads per row
            const int r = min(f / (kKc2 / 4), 63), q = f % (kKc2 / 4);
            const int k = k0 + 4 * q;
            v[i] = gld4(o.row(min(r0 + r, R - 1)) + max(min(k, K - 4), 0));
            if (k >= K) v[i] = f4zero();
        } else {                                               // f -> (k row kk, quad q of the 64 columns); R % 4 == 0
            const int kk = min(f >> 4, kKc2 - 1), q = f & 15;
            const int k = k0 + kk;
            v[i] = gld4(o.row(min(k, K - 1)) + min(r0 + 4 * q, R - 4));
            if (k >= K) v[i] = f4zero();
        }
    }
}
template <bool KC>
__device__ __forceinline__ void panel2_store(float* __restrict__ lds, const float4 (&v)[kStage2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < kStage2; ++i) {
        const int f = tid + i * kBlock;
        if (KC) {
            if (f < 64 * (kKc2 / 4)) st4(lds + (f / (kKc2 / 4)) * kLdk2 + 4 * (f % (kKc2 / 4)), v[i]);
        } else {
            if (f < kKc2 * 16) st4(lds + (f >> 4) * 64 + 4 * (f & 15), v[i]);
        }
    }
}
template <bool KC>
__device__ __forceinline__ float4 panel2_frag(const float* __restrict__ lds, int r, int kk) {
    if (KC) return ld4(lds + r * kLdk2 + kk);
    return make_float4(lds[(kk + 0) * 64 + r], lds[(kk + 1) * 64 + r], lds[(kk + 2) * 64 + r], lds[(kk + 3) * 64 + r]);
}

template <bool TB>
__device__ __forceinline__ void grouped_tile64(const GTask& t, int tile, float* __restrict__ lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q = wave & 3, kh = wave >> 2;
    const int m0 = (tile / t.tiles_n) * 64, n0 = (tile % t.tiles_n) * 64;
    const int ra = (q >> 1) * 32 + r, rb = (q & 1) * 32 + r;       // this lane's panel row (A) / column (B)
    const int M = t.M, N = t.N, K = t.K;
    float* la = lds;
    float* lb = lds + kPanel2;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    GG_STAMP(1);
    float4 va[kStage2], vb[kStage2];
    panel2_load<true>(t.A, m0, M, 0, K, va);
    panel2_load<TB>(t.B, n0, N, 0, K, vb);
    constexpr int kHalf = (kKc2 / 8 + 1) / 2;                      // K steps per wave per chunk = 5
    for (int k0 = 0; k0 < K; k0 += kKc2) {
        if (k0 > 0) __syncthreads();
        panel2_store<true>(la, va);
        panel2_store<TB>(lb, vb);
        panel2_load<true>(t.A, m0, M, k0 + kKc2, K, va);           // unconditional: see grouped_tile
        panel2_load<TB>(t.B, n0, N, k0 + kKc2, K, vb);
        __syncthreads();
        GG_STAMP(2);
        const int steps = (min(K - k0, kKc2) + 7) / 8;             // <= 10
        float4 a = panel2_frag<true>(la, ra, kh * 8 + 4 * h), b = panel2_frag<TB>(lb, rb, kh * 8 + 4 * h);
#pragma unroll
        for (int j = 0; j < kHalf; ++j) {
            const int kn = min(kh + 2 * (j + 1), kKc2 / 8 - 1) * 8 + 4 * h;
            const float4 an = panel2_frag<true>(la, ra, kn), bn = panel2_frag<TB>(lb, rb, kn);     // one step ahead
            if (kh + 2 * j < steps) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
            a = an;
            b = bn;
        }
    }
    GG_STAMP(3);
    __syncthreads();
    // wave (q, kh) finishes accumulator registers 8 kh .. 8 kh + 7 of its quadrant: it leaves the other eight in LDS for its
    // partner and reads the partner's; the sum is always (half 0) + (half 1)
    float (*red)[2][8][64] = reinterpret_cast<float (*)[2][8][64]>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) red[q][kh][i][lane] = kh ? acc[i] : acc[8 + i];
    __syncthreads();
    const int n = n0 + rb, nc = min(n, N - 1);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float mine = kh ? acc[8 + j] : acc[j], other = red[q][1 - kh][j][lane];
        v[j] = kh ? other + mine : mine + other;
    }
    // register j of this wave's eight: output row mb + (j & 3) + 8 (j >> 2); two groups of four, one after the other
    const int mb = m0 + (q >> 1) * 32 + 16 * kh + 4 * h;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float msk[4], old[4];
        float* cp[4];
        bool ok[4], acc_ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            msk[j] = old[j] = 0.f;
            const int m = mb + j + 8 * g, mc = min(m, M - 1);
            ok[j] = m < M && n < N;
            cp[j] = mc < t.c_split ? t.C + (int64_t)mc * t.ldc + nc : t.C2 + (int64_t)(mc - t.c_split) * t.ldc + nc;
            acc_ok[j] = mc < t.c_split;
            if (t.act == JMAC_GEMM_DACT_LEAKY || t.act == JMAC_GEMM_DACT_RELU) msk[j] = gld(t.mask + (int64_t)mc * t.ldmask + nc);
        }
        if (t.accumulate) {
#pragma unroll
            for (int j = 0; j < 4; ++j) old[j] = gld(cp[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = v[4 * g + j];
            if (t.act == JMAC_GEMM_ACT_LEAKY) x = x > 0.f ? x : x * t.slope;
            else if (t.act == JMAC_GEMM_ACT_RELU) x = x > 0.f ? x : 0.f;
            else if (t.act == JMAC_GEMM_DACT_LEAKY) x = msk[j] > 0.f ? x : x * t.slope;
            else if (t.act == JMAC_GEMM_DACT_RELU) x = msk[j] > 0.f ? x : 0.f;
            if (t.accumulate && acc_ok[j]) x += old[j];
            if (ok[j]) gst(cp[j], x);
        }
    }
    GG_STAMP(4);
}

// ---- TN products (the weight gradients: C = A^T B, A [K,M], B [K,N], K = the ~10^3 relation rows) without LDS staging ------
// In this form BOTH operands are stored with the tile's 32 rows / columns contiguous, which is exactly the fragment layout of
// v_mfma_f32_32x32x2_f32 (lane (r, h) holds the element of row / column r at k0 + h): a wave's fragment load is two fully
// used 128-byte segments straight from L2 into the register the MFMA reads.  No panel, no barrier, no chunk passes: wave w
// walks the k pairs w, w + 8, ... with TN_U pairs of loads in flight, and the eight partial tiles meet in the usual LDS
// reduction.  (Staged through LDS, a K = 962 tile was four dependent chunk passes, ~15 us per block: the products that held
// every backward level of the step.)  No alignment requirement: the loads are 4 bytes per lane.
constexpr int TN_U = 8;
__device__ __forceinline__ void grouped_tile_tn(const GTask& t, int tile, float* __restrict__ lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = (tile / t.tiles_n) * 32, n0 = (tile % t.tiles_n) * 32;
    const int M = t.M, N = t.N, K = t.K;
    const int mc = min(m0 + r, M - 1), nc = min(n0 + r, N - 1);     // clamped: rows / columns past the edge are never stored
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    GG_STAMP(1);
    // Full batches below the A operand's split run on two incremented row pointers (two VALU adds per load: with the
    // general row() form -- clamp, split select, 64-bit multiply -- per load the loop was bound by its ADDRESS arithmetic,
    // ~45 VALU instructions per k pair, and two blocks on a CU took twice as long as one).  The last, partial batch (and the
    // rows of a second buffer) takes the general form once.
    const int pairs = (K + 1) >> 1;
    const int pairs_fast = min(K, t.A.split) >> 1;             // pairs whose two rows both lie in A.p (and below K)
    const int64_t sa = (int64_t)2 * kWaves * t.A.ld, sb = (int64_t)2 * kWaves * t.B.ld;
    const float* qa = t.A.p + (int64_t)(2 * wave + h) * t.A.ld + mc;
    const float* qb = t.B.p + (int64_t)(2 * wave + h) * t.B.ld + nc;
    int p = wave;
    // software pipeline over the full batches: batch i + 1's loads are issued before batch i's (dependent) MFMA chain
    float a[TN_U], b[TN_U];
    bool have = p + (TN_U - 1) * kWaves < pairs_fast;
    if (have) {
#pragma unroll
        for (int u = 0; u < TN_U; ++u) {
            a[u] = gld(qa + u * sa);
            b[u] = gld(qb + u * sb);
        }
        qa += TN_U * sa;
        qb += TN_U * sb;
        p += TN_U * kWaves;
    }
    while (have) {
        float an[TN_U], bn[TN_U];
        const bool more = p + (TN_U - 1) * kWaves < pairs_fast;            // wave-uniform
        if (more) {
#pragma unroll
            for (int u = 0; u < TN_U; ++u) {
                an[u] = gld(qa + u * sa);
                bn[u] = gld(qb + u * sb);
            }
            qa += TN_U * sa;
            qb += TN_U * sb;
            p += TN_U * kWaves;
        }
#pragma unroll
        for (int u = 0; u < TN_U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        if (more) {
#pragma unroll
            for (int u = 0; u < TN_U; ++u) {
                a[u] = an[u];
                b[u] = bn[u];
            }
        }
        have = more;
    }
    for (; p < pairs; p += TN_U * kWaves) {                    // wave-uniform; one pass unless the split leaves a long tail
        float a[TN_U], b[TN_U];                                // (the pipeline's registers are dead here)
#pragma unroll
        for (int u = 0; u < TN_U; ++u) {
            const int pu = p + u * kWaves, k = 2 * pu + h, kc = min(k, K - 1);
            a[u] = gld(t.A.row(kc) + mc);
            b[u] = gld(t.B.row(kc) + nc);
            if (pu >= pairs || k >= K) a[u] = 0.f;             // past the end / odd K: the pair contributes nothing
        }
#pragma unroll
        for (int u = 0; u < TN_U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    GG_STAMP(2);
    GG_STAMP(3);
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(lds);
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    const int n = n0 + r;
    constexpr int kRegs = 16 / kWaves;
#pragma unroll
    for (int j = 0; j < kRegs; ++j) {
        const int i = kRegs * wave + j;
        float sum = red[0][i][lane];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) sum += red[w][i][lane];
        const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h, mr = min(m, M - 1);
        const bool in_c = mr < t.c_split;
        float* cp = in_c ? t.C + (int64_t)mr * t.ldc + nc : t.C2 + (int64_t)(mr - t.c_split) * t.ldc + nc;
        float x = sum;
        if (t.act == JMAC_GEMM_ACT_LEAKY) x = x > 0.f ? x : x * t.slope;
        else if (t.act == JMAC_GEMM_ACT_RELU) x = x > 0.f ? x : 0.f;
        else if (t.act == JMAC_GEMM_DACT_LEAKY || t.act == JMAC_GEMM_DACT_RELU) {
            const float mk = gld(t.mask + (int64_t)mr * t.ldmask + nc);
            x = mk > 0.f ? x : (t.act == JMAC_GEMM_DACT_LEAKY ? x * t.slope : 0.f);
        }
        if (t.accumulate && in_c) x += gld(cp);
        if (m < M && n < N) gst(cp, x);
    }
    GG_STAMP(4);
}

__global__ __launch_bounds__(kBlock, 4) void grouped_gemm_kernel(const GTable tab) {   // 4 waves per SIMD (two 8-wave blocks per CU)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    GG_STAMP(0);
    // The table is a by-value kernel argument indexed by a run-time task id.  Indexing `tab` itself makes the compiler copy
    // all 3.6 KB of it into per-lane scratch; the kernarg segment is ordinary constant memory, so the scan and the one task
    // this block runs are read from there with scalar loads instead.
    typedef const GTable __attribute__((address_space(4))) * KTab;
    KTab kt = (KTab)__builtin_amdgcn_kernarg_segment_ptr();
    (void)tab;
    // grid = (largest tile count of the launch, tasks): the task index is blockIdx.y, so the block's only dependent scalar
    // round trip is the read of its own task (a prefix scan over the table cost two more: ~1 us each from a kernarg buffer
    // the host has just written); blocks past their task's tile count leave at once
    // The task is fetched with ONE vector load per wave (lane i <- dword i of the task) and spread to scalars with
    // v_readlane: a single memory round trip.  Read field by field with scalar loads, the compiler sinks the loads next to
    // their uses -- three dependent trips to the kernarg buffer (task pointers, dimensions, epilogue parameters) at ~1.7 us
    // each, half of a block's life (phase timestamps: tools/gg_trace.py).
    GTask t;                                                                 // 38 dwords, block-uniform: SGPRs
    {
        static_assert(sizeof(GTask) / 4 <= 64, "one lane per dword of the task");
        typedef const uint32_t __attribute__((address_space(1))) * GWords;
        GWords src = (GWords)(uintptr_t)(&kt->t[blockIdx.y]);
        const int l = threadIdx.x & 63;
        const uint32_t w = src[l < (int)(sizeof(GTask) / 4) ? l : 0];
        uint32_t* dst = reinterpret_cast<uint32_t*>(&t);
#pragma unroll
        for (unsigned i = 0; i < sizeof(GTask) / 4; ++i) dst[i] = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)i);
    }
    // XCD-aware tile order.  Workgroups go to the eight XCDs round-robin by linear id, so block x of a task runs on XCD
    // (x + const) % 8; each XCD has its own L2.  Tiles that share an operand panel (consecutive tile ids share the A panel of
    // their row block) are therefore given to ONE XCD: x -> tile (x % 8) * chunk + x / 8, chunk = ceil(tiles / 8).  With the
    // identity order a panel is fetched by up to eight L2s (a backward level requests ~210 MB of panels for ~15 MB of
    // operands).  Measured on warm operands it is worth 1 us of a 49 us level: the launches are bound by a block's latency
    // chain, not by the fabric; kept for the cold operands of the step.
    const int chunk = (t.tile_begin + 7) >> 3;                               // tile_begin holds the task's tile COUNT
    const int tile = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
    if ((int)blockIdx.x >= 8 * chunk || tile >= t.tile_begin) return;
    if (t.ta && !t.tb) {
        grouped_tile_tn(t, tile, lds);
        return;
    }
    if (t.big) {
        if (t.tb) grouped_tile64<true>(t, tile, lds);
        else grouped_tile64<false>(t, tile, lds);
        return;
    }
    // an RC panel (32 tile columns contiguous) takes vector loads only where the whole 32-column tile lies inside the operand
    const int m0t = (tile / t.tiles_n) * 32, n0t = (tile % t.tiles_n) * 32;
    const bool va = (t.vec & 1) != 0 && (!t.ta || m0t + 32 <= t.M), vb = (t.vec & 2) != 0 && (t.tb || n0t + 32 <= t.N);
    const bool multi = t.K > kKc;
#define JMAC_GG(TA, TB)                                                                   \
    do {                                                                                  \
        if (va && vb) {                                                                   \
            if (multi) grouped_tile<TA, TB, true, true, true>(t, tile, lds);              \
            else grouped_tile<TA, TB, true, true, false>(t, tile, lds);                   \
        } else if (va) grouped_tile<TA, TB, true, false, true>(t, tile, lds);             \
        else if (vb) grouped_tile<TA, TB, false, true, true>(t, tile, lds);               \
        else grouped_tile<TA, TB, false, false, true>(t, tile, lds);                      \
    } while (0)
    if (t.ta) JMAC_GG(true, true);                       // (TN left above)
    else if (t.tb) JMAC_GG(false, true);
    else JMAC_GG(false, false);
#undef JMAC_GG
}

// ---- [Wt | Wb | Wg] of up to four layers in one launch, and its adjoint ---------------------------------------------------
// w_att = [Wt; Wb] is stacked by rows (src/jmac_model.py:24,75-76), the projection GEMM wants [Wt|Wb|Wg] [d, 3d] side by
// side.  torch.cat builds it with one launch per layer (and the backward needs a cat + a strided clone per layer to cut
// d w_att / d gcn_weight out of d[Wt|Wb|Wg]); here all layers of an encoder call go in one launch each way.
struct WcatArgs {
    const float* w_att[JMAC_WCAT_MAX];     // [2d, d]
    const float* gcn[JMAC_WCAT_MAX];       // [d, d]
    float* wcat[JMAC_WCAT_MAX];            // [d, 3d]
    float* d_watt[JMAC_WCAT_MAX];          // adjoint outputs
    float* d_gcn[JMAC_WCAT_MAX];
    const float* dwcat[JMAC_WCAT_MAX];
    const float* extra_src;                // one more contiguous copy riding along (blockIdx.y == n), or NULL
    float* extra_dst;
    int64_t extra_n4;                      // float4s
    int64_t* counters[JMAC_WCAT_MAX];      // incremented by one (BatchNorm's num_batches_tracked of the layers), pack only
    int n_counters;
    int64_t* seed_state;                   // pack only: [2] persistent dropout seed words, advanced by one per launch ...
    int64_t* seed_out;                     // ... and copied here: the seeds THIS step's normalise + dropout kernels draw from
    int n, d;
};
template <bool ADJOINT>
__global__ __launch_bounds__(256) void wcat_kernel(const WcatArgs a) {
    const int l = blockIdx.y, d = a.d, D4 = d / 4;
    if (!ADJOINT && blockIdx.x == 0 && l == 0 && threadIdx.x < (unsigned)a.n_counters) a.counters[threadIdx.x][0] += 1;
    if (!ADJOINT && blockIdx.x == 0 && l == 0 && a.seed_state && threadIdx.x >= 64 && threadIdx.x < 66) {
        const int64_t v = a.seed_state[threadIdx.x - 64] + 1;
        a.seed_state[threadIdx.x - 64] = v;
        a.seed_out[threadIdx.x - 64] = v;
    }
    if (l == a.n) {                                            // the extra copy
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.extra_n4; i += (int64_t)gridDim.x * 256)
            st4(a.extra_dst + 4 * i, ld4(a.extra_src + 4 * i));
        return;
    }
    const int64_t total = (int64_t)d * 3 * D4;                 // float4s of one [d, 3d] matrix
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / (3 * D4)), c4 = (int)(i % (3 * D4));
        const int part = c4 / D4, c = (c4 % D4) * 4;           // 0: Wt, 1: Wb, 2: Wg
        if (!ADJOINT) {
            const float* src = part == 0 ? a.w_att[l] + (int64_t)r * d + c
                             : part == 1 ? a.w_att[l] + (int64_t)(d + r) * d + c : a.gcn[l] + (int64_t)r * d + c;
            st4(a.wcat[l] + (int64_t)r * 3 * d + part * d + c, ld4(src));
        } else {
            float* dst = part == 0 ? a.d_watt[l] + (int64_t)r * d + c
                       : part == 1 ? a.d_watt[l] + (int64_t)(d + r) * d + c : a.d_gcn[l] + (int64_t)r * d + c;
            st4(dst, ld4(a.dwcat[l] + (int64_t)r * 3 * d + part * d + c));
        }
    }
}

// ---- used-relation compaction (jmac_rows_compact_f32 / jmac_rows_expand_f32) ------------------------------------------------
// A DBP-5L KG touches 153-833 of its 961 relation rows (ja: 158); the layer's relation transform R'' = act(cat(R, loop) W1) W2 and
// its projection [Rq|Rz] are needed for the rows some edge names and for nothing else, so the encoder runs the relation-side
// products on the compact table of used rows (six times fewer rows on ja) and these two kernels move rows in and gradients out.
constexpr int kRowsMax = 4;
struct RowsArgs {
    const float* src[kRowsMax];
    float* dst[kRowsMax];
    int64_t ld[kRowsMax];          // compact: leading dimension of src;  expand: of dst
    int accumulate[kRowsMax];
    const int64_t* idx;            // compact: row of src for each compact row
    const int32_t* pos;            // expand: compact row of each full row, -1 = none
    int64_t rows;                  // compact: n_used;  expand: rows of the full tables
    int D4, n;
};
template <bool EXPAND>
__global__ __launch_bounds__(256) void rows_move_kernel(const RowsArgs a) {
    const int t = blockIdx.y;
    const int64_t total = a.rows * a.D4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.D4;
        const int c = (int)(i % a.D4) * 4;
        if (!EXPAND) {
            st4(a.dst[t] + r * (4 * a.D4) + c, ld4(a.src[t] + a.idx[r] * a.ld[t] + c));
        } else {
            const int p = a.pos[r];
            float* q = a.dst[t] + r * a.ld[t] + c;
            if (a.accumulate[t]) {
                if (p >= 0) {
                    const float4 v = ld4(a.src[t] + (int64_t)p * (4 * a.D4) + c), o = ld4(q);
                    st4(q, make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w));
                }
            } else {
                st4(q, p >= 0 ? ld4(a.src[t] + (int64_t)p * (4 * a.D4) + c) : make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
    }
}

}  // namespace

extern "C" {

int jmac_gemm_f32(const float* A, int64_t lda, int32_t transA, const float* B, int64_t ldb, int32_t transB, int64_t M, int64_t N,
                  int64_t K, float* C, int64_t ldc, jmac_stream_t stream) {
    jmac_gemm_task_t t{};                              // one product = a grouped launch of one task
    t.A = A; t.lda = lda; t.transA = transA; t.B = B; t.ldb = ldb; t.transB = transB;
    t.C = C; t.ldc = ldc; t.M = M; t.N = N; t.K = K;
    return jmac_gemm_grouped_f32(&t, 1, stream);
}

#ifdef JMAC_GG_TRACE
int jmac_gemm_trace_buffer(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(gg_trace_buf), &buf, sizeof(buf));
}
#endif

static int64_t big_rows() {                          // rows from which a product takes 64x64 tiles (JMAC_GG_TILE64_ROWS: A/B switch)
    static const int64_t rows = [] { const char* e = getenv("JMAC_GG_TILE64_ROWS"); return e ? (int64_t)atoll(e) : (int64_t)2048; }();
    return rows;
}

int jmac_gemm_grouped_f32(const jmac_gemm_task_t* tasks, int32_t n_tasks, jmac_stream_t stream) {
    if (n_tasks < 0 || n_tasks > kMaxTasks || (n_tasks > 0 && !tasks)) return JMAC_EINVAL;
    GTable tab{};
    int64_t tiles = 0;
    int n = 0;
    for (int i = 0; i < n_tasks; ++i) {
        const jmac_gemm_task_t& u = tasks[i];
        if (u.M < 0 || u.N < 0 || u.K < 0) return JMAC_EINVAL;
        if (u.M == 0 || u.N == 0) continue;
        if (u.K == 0) return JMAC_EINVAL;                  // an empty contraction: nothing to launch for
        if (!u.C || (u.K > 0 && (!u.A || !u.B))) return JMAC_EINVAL;
        if (u.M >= INT32_MAX || u.N >= INT32_MAX || u.K >= INT32_MAX) return JMAC_ERANGE;
        if (u.act < 0 || u.act > JMAC_GEMM_DACT_RELU) return JMAC_EINVAL;
        if ((u.act == JMAC_GEMM_DACT_LEAKY || u.act == JMAC_GEMM_DACT_RELU) && !u.act_src) return JMAC_EINVAL;
        if ((u.a_split > 0 && !u.A2) || (u.c_split > 0 && !u.C2)) return JMAC_EINVAL;
        GTask& t = tab.t[n++];
        // the memory rows of A are its M rows (no transpose) or its K rows (transposed); a_split counts memory rows
        t.A = GOperand{u.A, u.A2, u.lda, u.a_split > 0 ? (int32_t)u.a_split : INT32_MAX};
        t.B = GOperand{u.B, nullptr, u.ldb, INT32_MAX};
        t.C = u.C; t.C2 = u.C2; t.ldc = u.ldc; t.c_split = u.c_split > 0 ? (int32_t)u.c_split : INT32_MAX;
        t.mask = u.act_src; t.ldmask = u.ld_act_src;
        t.M = (int32_t)u.M; t.N = (int32_t)u.N; t.K = (int32_t)u.K;
        t.ta = u.transA ? 1 : 0; t.tb = u.transB ? 1 : 0;
        t.act = u.act; t.accumulate = u.accumulate ? 1 : 0; t.slope = u.slope;
        // 16-byte staging loads per operand.  KC form (k contiguous): K % 4 == 0, 16-byte aligned rows.  RC form (the 32
        // tile columns contiguous): 16-byte aligned rows (tile column offsets are multiples of 32 floats).
        const bool a_kc = !u.transA, b_kc = u.transB != 0;
        const bool a_al = u.lda % 4 == 0 && aligned16(u.A) && (u.a_split <= 0 || aligned16(u.A2));
        const bool b_al = u.ldb % 4 == 0 && aligned16(u.B);
        const bool k4 = u.K >= 4 && u.K % 4 == 0;
        t.vec = ((a_al && (!a_kc || k4)) ? 1 : 0) | ((b_al && (!b_kc || k4)) ? 2 : 0);
        // 64x64 tiles for long tables: A row-major with vector loads, B either K-contiguous (vector loads) or N-contiguous with
        // N % 4 == 0 (ragged column tiles then end on a quad boundary)
        t.big = (!u.transA && (t.vec & 1) && (t.vec & 2) && (u.transB || u.N % 4 == 0) && u.M >= big_rows() && u.N >= 64) ? 1 : 0;
        const int ts = t.big ? 64 : 32;
        t.tiles_n = (int32_t)((u.N + ts - 1) / ts);
        const int64_t nt = (int64_t)((u.M + ts - 1) / ts) * t.tiles_n;
        if (nt >= 65536 * 16) return JMAC_ERANGE;
        t.tile_begin = (int32_t)nt;                    // this launch form: the task's tile count
        const int64_t nx = 8 * ((nt + 7) / 8);         // block ids of the XCD-aware order (8 * ceil(tiles / 8))
        tiles = nx > tiles ? nx : tiles;
    }
    if (tiles == 0) return JMAC_OK;
    // 77 KB of dynamic LDS per block: above the 64 KB default.  The attribute is PER DEVICE (a process that launches on a second
    // GPU needs it there too): remembered per device ordinal, set unconditionally past the table
    static bool lds_ok[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    if (dev < 0 || dev >= 64 || !lds_ok[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(grouped_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kGroupedLds) != hipSuccess)
            return (int)hipGetLastError();
        if (dev >= 0 && dev < 64) lds_ok[dev] = true;
    }
    hipLaunchKernelGGL(grouped_gemm_kernel, dim3((unsigned)tiles, (unsigned)n), dim3(kBlock), kGroupedLds, (hipStream_t)stream, tab);
    return (int)hipGetLastError();
}

static int wcat_extra(WcatArgs& a, const float* src, float* dst, int64_t n) {
    if (n < 0 || (n > 0 && (!src || !dst)) || n % 4 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return JMAC_EINVAL;
    a.extra_src = n ? src : nullptr; a.extra_dst = dst; a.extra_n4 = n / 4;
    return JMAC_OK;
}

int jmac_wcat_pack_f32(const float* const* w_att, const float* const* gcn, float* const* wcat, int32_t n_layers, int64_t d,
                       const float* extra_src, float* extra_dst, int64_t extra_floats, int64_t* const* counters,
                       int32_t n_counters, jmac_stream_t stream) {
    return jmac_wcat_pack_seed_f32(w_att, gcn, wcat, n_layers, d, extra_src, extra_dst, extra_floats, counters, n_counters, nullptr,
                                   nullptr, stream);
}

int jmac_wcat_pack_seed_f32(const float* const* w_att, const float* const* gcn, float* const* wcat, int32_t n_layers, int64_t d,
                            const float* extra_src, float* extra_dst, int64_t extra_floats, int64_t* const* counters,
                            int32_t n_counters, int64_t* seed_state, int64_t* seed_out, jmac_stream_t stream) {
    if (n_layers <= 0 || n_layers > JMAC_WCAT_MAX || d <= 0 || !w_att || !gcn || !wcat) return JMAC_EINVAL;
    if (d % 4) return JMAC_EDIM;
    if ((seed_state == nullptr) != (seed_out == nullptr)) return JMAC_EINVAL;
    WcatArgs a{};
    a.seed_state = seed_state; a.seed_out = seed_out;
    a.n = n_layers; a.d = (int)d;
    for (int i = 0; i < n_layers; ++i) {
        if (!w_att[i] || !gcn[i] || !wcat[i]) return JMAC_EINVAL;
        a.w_att[i] = w_att[i]; a.gcn[i] = gcn[i]; a.wcat[i] = wcat[i];
    }
    const int64_t total = d * 3 * (d / 4);
    if (int rc = wcat_extra(a, extra_src, extra_dst, extra_floats)) return rc;
    if (n_counters < 0 || n_counters > JMAC_WCAT_MAX || (n_counters > 0 && !counters)) return JMAC_EINVAL;
    a.n_counters = n_counters;
    for (int i = 0; i < n_counters; ++i) {
        if (!counters[i]) return JMAC_EINVAL;
        a.counters[i] = counters[i];
    }
    hipLaunchKernelGGL(wcat_kernel<false>, dim3((unsigned)((total + 255) / 256), (unsigned)(n_layers + (a.extra_src ? 1 : 0))),
                       dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int jmac_wcat_unpack_f32(const float* const* dwcat, float* const* d_watt, float* const* d_gcn, int32_t n_layers, int64_t d,
                         const float* extra_src, float* extra_dst, int64_t extra_floats, jmac_stream_t stream) {
    if (n_layers <= 0 || n_layers > JMAC_WCAT_MAX || d <= 0 || !dwcat || !d_watt || !d_gcn) return JMAC_EINVAL;
    if (d % 4) return JMAC_EDIM;
    WcatArgs a{};
    a.n = n_layers; a.d = (int)d;
    for (int i = 0; i < n_layers; ++i) {
        if (!dwcat[i] || !d_watt[i] || !d_gcn[i]) return JMAC_EINVAL;
        a.dwcat[i] = dwcat[i]; a.d_watt[i] = d_watt[i]; a.d_gcn[i] = d_gcn[i];
    }
    const int64_t total = d * 3 * (d / 4);
    if (int rc = wcat_extra(a, extra_src, extra_dst, extra_floats)) return rc;
    hipLaunchKernelGGL(wcat_kernel<true>, dim3((unsigned)((total + 255) / 256), (unsigned)(n_layers + (a.extra_src ? 1 : 0))),
                       dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int jmac_rows_compact_f32(const float* const* h_src, const int64_t* h_ld_src, float* const* h_dst, int32_t n_tables,
                          const int64_t* idx, int64_t n_used, int64_t d, jmac_stream_t stream) {
    if (n_tables <= 0 || n_tables > kRowsMax || n_used < 0 || d <= 0 || !h_src || !h_ld_src || !h_dst) return JMAC_EINVAL;
    if (d % 4) return JMAC_EDIM;
    if (n_used == 0) return JMAC_OK;
    if (!idx) return JMAC_EINVAL;
    RowsArgs a{};
    a.idx = idx; a.rows = n_used; a.D4 = (int)(d / 4); a.n = n_tables;
    for (int t = 0; t < n_tables; ++t) {
        if (!h_src[t] || !h_dst[t] || h_ld_src[t] % 4 || (((uintptr_t)h_src[t] | (uintptr_t)h_dst[t]) & 15)) return JMAC_EINVAL;
        a.src[t] = h_src[t]; a.dst[t] = h_dst[t]; a.ld[t] = h_ld_src[t];
    }
    const int64_t total = n_used * (d / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rows_move_kernel<false>, dim3((unsigned)blocks, (unsigned)n_tables), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int jmac_rows_expand_f32(const float* const* h_src, float* const* h_dst, const int64_t* h_ld_dst, const int32_t* h_accumulate,
                         int32_t n_tables, const int32_t* pos, int64_t rows, int64_t d, jmac_stream_t stream) {
    if (n_tables <= 0 || n_tables > kRowsMax || rows < 0 || d <= 0 || !h_src || !h_ld_dst || !h_dst) return JMAC_EINVAL;
    if (d % 4) return JMAC_EDIM;
    if (rows == 0) return JMAC_OK;
    if (!pos) return JMAC_EINVAL;
    RowsArgs a{};
    a.pos = pos; a.rows = rows; a.D4 = (int)(d / 4); a.n = n_tables;
    for (int t = 0; t < n_tables; ++t) {
        if (!h_dst[t] || h_ld_dst[t] % 4 || (((uintptr_t)h_src[t] | (uintptr_t)h_dst[t]) & 15)) return JMAC_EINVAL;
        a.src[t] = h_src[t]; a.dst[t] = h_dst[t]; a.ld[t] = h_ld_dst[t];
        a.accumulate[t] = h_accumulate ? h_accumulate[t] : 0;
    }
    const int64_t total = rows * (d / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rows_move_kernel<true>, dim3((unsigned)blocks, (unsigned)n_tables), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // extern "C"
