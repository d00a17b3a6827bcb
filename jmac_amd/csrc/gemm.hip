// gemm.hip -- small fp32 GEMM for the relation-side projections of the layer (gfx950).
//
// The reference transforms the relation table with dense products on ~10^3 rows: rel @ W1, act(.) @ W2
// (src/jmac_model.py:40-42), the hoisted rel'' @ [Wb|Wg] of the factorised layer, and JMAC's relation MLPs
// (src/jmac_model.py:195-196) -- [962,300] x [300,300..600] at DBP-5L size, 36 of them per training step with their
// backward forms.  A library GEMM spends ~17 us on each (24 workgroups of a 256-CU chip); here one 8-wave block owns
// a 32x32 output tile, the waves split K, every operand fragment goes from L2 straight to registers (the matrices
// are ~1 MB: no LDS staging, no barrier inside the K loop) and the partial tiles are summed through LDS in wave
// order (deterministic).  v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation.
//
// C[M,N] = op(A) op(B), row-major, op = identity or transpose: the three forms autograd needs
//   forward  C = A B           (NN)        dA = G B^T  (NT)        dB = A^T G  (TN)
#include "common.h"

using namespace jmac;

namespace {

constexpr int kWaves = 8;             // waves per 32x32 tile = K split factor
constexpr int kBlock = 64 * kWaves;
constexpr int kAhead = 4;             // K steps in flight per wave ahead of the MFMAs (8 measured no faster)
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Fragment of one 8-deep K step for the 32x32x2 MFMA: lane (r = lane & 31, h = lane >> 5) holds the operand's
// values for row/column r and k = k0 + 4h + s, s = 0..3; MFMA step s then contracts k = {k0 + s, k0 + 4 + s}.
// KC = true: the operand is stored with k contiguous (row r is a memory row): one 16-byte load.
// KC = false: stored with r contiguous (k selects the memory row): four 4-byte loads, each coalesced over r.
template <bool KC, bool VEC>
__device__ __forceinline__ float4 load_frag(const float* __restrict__ p, int64_t ld, int r, int R, int k, int K) {
    const int rc = min(r, R - 1);                      // rows past the end feed outputs that are never stored
    if (KC) {
        const float* q = p + (int64_t)rc * ld;
        if (VEC) {                                     // K % 4 == 0, rows 16-byte aligned
            const float4 v = ld4(q + min(k, K - 4));
            return k < K ? v : f4zero();
        }
        float4 v;
        v.x = k + 0 < K ? q[k + 0] : 0.f;
        v.y = k + 1 < K ? q[k + 1] : 0.f;
        v.z = k + 2 < K ? q[k + 2] : 0.f;
        v.w = k + 3 < K ? q[k + 3] : 0.f;
        return v;
    }
    float4 v;
    v.x = k + 0 < K ? p[(int64_t)(k + 0) * ld + rc] : 0.f;
    v.y = k + 1 < K ? p[(int64_t)(k + 1) * ld + rc] : 0.f;
    v.z = k + 2 < K ? p[(int64_t)(k + 2) * ld + rc] : 0.f;
    v.w = k + 3 < K ? p[(int64_t)(k + 3) * ld + rc] : 0.f;
    return v;
}

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(kBlock) void small_gemm_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                            int64_t ldb, int M, int N, int K, float* __restrict__ C, int64_t ldc,
                                                            int tiles_n) {
    __shared__ float red[kWaves - 1][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
    // K split over the waves in 8-deep steps
    const int nsteps = (K + 7) / 8;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // op(A)(m, k): TA ? A[k][m] : A[m][k]  -> k contiguous unless TA;   op(B)(k, n): TB ? B[n][k] : B[k][n] -> k contiguous if TB
    auto fa = [&](int step) { return load_frag<!TA, VEC>(A, lda, m0 + r, M, step * 8 + 4 * h, K); };
    auto fb = [&](int step) { return load_frag<TB, VEC>(B, ldb, n0 + r, N, step * 8 + 4 * h, K); };
    // a ring of kAhead fragment pairs: the loads of kAhead steps are in flight while one step's MFMAs run.
    // Each wave owns a CONTIGUOUS run of steps, so the 128-byte lines of a k-contiguous operand (4 steps each) are
    // fetched by one wave instead of four.
    const int per = (nsteps + kWaves - 1) / kWaves;
    const int sbeg = wave * per, send = min(nsteps, sbeg + per);
    float4 ra[kAhead], rb[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
        ra[j] = f4zero();
        rb[j] = f4zero();
        if (sbeg + j < send) {
            ra[j] = fa(sbeg + j);
            rb[j] = fb(sbeg + j);
        }
    }
    for (int base = sbeg; base < send; base += kAhead) {
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {
            const int s = base + j;
            if (s < send) {                                    // wave-uniform
                const float4 a = ra[j], b = rb[j];
                if (s + kAhead < send) {
                    ra[j] = fa(s + kAhead);
                    rb[j] = fb(s + kAhead);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
    }
    // sum the partial tiles in wave order (0 + 1 + ... ): fixed order, bitwise reproducible
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = acc[i];
#pragma unroll
            for (int w = 0; w < kWaves - 1; ++w) v += red[w][i][lane];
            // C/D map of the 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
            const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h, n = n0 + r;
            if (m < M && n < N) C[(int64_t)m * ldc + n] = v;
        }
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

int jmac_gemm_f32(const float* A, int64_t lda, int32_t transA, const float* B, int64_t ldb, int32_t transB, int64_t M, int64_t N,
                  int64_t K, float* C, int64_t ldc, jmac_stream_t stream) {
    if (M < 0 || N < 0 || K < 0) return JMAC_EINVAL;
    if (M == 0 || N == 0) return JMAC_OK;
    if (!C || (K > 0 && (!A || !B))) return JMAC_EINVAL;
    if (M >= INT32_MAX || N >= INT32_MAX || K >= INT32_MAX) return JMAC_ERANGE;
    const int tiles_m = (int)((M + 31) / 32), tiles_n = (int)((N + 31) / 32);
    if ((int64_t)tiles_m * tiles_n >= INT32_MAX) return JMAC_ERANGE;
    const dim3 grid((unsigned)(tiles_m * tiles_n)), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    // the 16-byte fragment loads need K % 4 == 0 and 16-byte aligned rows on whichever operands are k-contiguous
    const bool vec = K >= 4 && K % 4 == 0 && (transA || (lda % 4 == 0 && aligned16(A))) && (!transB || (ldb % 4 == 0 && aligned16(B)));
#define JMAC_GEMM_LAUNCH(TA, TB)                                                                                            \
    do {                                                                                                                    \
        if (vec) hipLaunchKernelGGL((small_gemm_kernel<TA, TB, true>), grid, block, 0, st, A, lda, B, ldb, (int)M, (int)N,  \
                                    (int)K, C, ldc, tiles_n);                                                               \
        else hipLaunchKernelGGL((small_gemm_kernel<TA, TB, false>), grid, block, 0, st, A, lda, B, ldb, (int)M, (int)N,     \
                                (int)K, C, ldc, tiles_n);                                                                   \
    } while (0)
    if (transA && transB) JMAC_GEMM_LAUNCH(true, true);
    else if (transA) JMAC_GEMM_LAUNCH(true, false);
    else if (transB) JMAC_GEMM_LAUNCH(false, true);
    else JMAC_GEMM_LAUNCH(false, false);
#undef JMAC_GEMM_LAUNCH
    return (int)hipGetLastError();
}

}  // extern "C"
