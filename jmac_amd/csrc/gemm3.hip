// gemm3.hip -- fp32 GEMM on the bf16 matrix cores (gfx950): C[M,N] = A[M,K] * B[N,K]^T with fp32-level accuracy.
//
// The N-row dense products of the encoder (src/jmac_model.py:177-203 uni_linear / all_linear mixes) and of the
// factorised layer ([P|Q|Z] = X [Wt|Wb|Wg], and the adjoint d X = d[P|Q|Z] [Wt|Wb|Wg]^T) are 60 % of the DBP-5L
// training step when they run as library fp32 GEMMs: the fp32-input MFMA (v_mfma_f32_32x32x2_f32) issues at the VECTOR
// rate, 1/16 of the bf16 matrix rate.  Here every fp32 operand element is split into three bf16 terms
//      x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (24 mantissa bits kept)
// while its tile is staged into LDS, and the product is accumulated from the six term pairs whose magnitude exceeds
// 2^-24 |a b|:   hi*hi + (hi*mid + mid*hi) + (hi*lo + mid*mid + lo*hi)     (bf16 x bf16 is exact in fp32; fp32 accumulate).
// Six v_mfma_f32_32x32x16_bf16 per fp32 k-step pair of 16 = 6/16 of the fp32 MFMA's issue time.
// The dropped pairs (mid*lo, lo*mid, lo*lo) are below 2^-32 |a b|.  No scaling is needed: bf16 has fp32's exponent range.
//
// Shape: 128 x BN tile per 256-thread block (BN = 128: waves 2 x 2, 64 x 64 each; BN = 64: 64 x 32 each), K staged 32
// deep: global float4 loads (unconditional, clamped) -> split in registers -> three bf16 planes [row][k] in LDS (rows
// padded to 80 B: conflict-free ds_read_b128 fragment reads) -> fragments -> MFMA.  One LDS buffer, two barriers per
// slab, the next slab's global loads in flight across the MFMAs; two blocks per CU so that one block's split (VALU) runs
// beside the other block's MFMAs (separate pipes).
#include <cstdlib>

#include "common.h"
#include "jmac_hip_testing.h"

using namespace jmac;

namespace {

constexpr int kBlock = 256;
constexpr int G3_BK = 32;
constexpr int G3_ROWB = 64;                      // bytes per LDS row: 32 bf16, four 16-byte chunks, XOR-swizzled by row
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {        // v_cvt_pk_bf16_f32: round to nearest even
    bf16x2 v;
    v.x = (__bf16)a;
    v.y = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo_f(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// x (4 consecutive k of one row) -> three bf16 quadruples
__device__ __forceinline__ void split4(float4 x, uint2& h, uint2& m, uint2& l) {
    h.x = pk_bf16(x.x, x.y);
    h.y = pk_bf16(x.z, x.w);
    const float r0 = x.x - lo_f(h.x), r1 = x.y - hi_f(h.x), r2 = x.z - lo_f(h.y), r3 = x.w - hi_f(h.y);
    m.x = pk_bf16(r0, r1);
    m.y = pk_bf16(r2, r3);
    l.x = pk_bf16(r0 - lo_f(m.x), r1 - hi_f(m.x));
    l.y = pk_bf16(r2 - lo_f(m.y), r3 - hi_f(m.y));
}

// LDS image of one bf16 plane of a tile: [row][32 k], 64-byte rows; the 16-byte chunk q (k = 8q..8q+7) of row r sits at
// chunk q ^ ((r >> 2) & 3): the 16 rows a ds_read_b128 lane group touches then cover all sixteen 4-dword bank slots
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * G3_ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int BM, int BN, int WPE>
__global__ __launch_bounds__(kBlock, WPE) void gemm_bf16x3_nt_kernel(const float* __restrict__ A, int64_t lda,
                                                                    const float* __restrict__ Bm, int64_t ldb, int M, int N, int K,
                                                                    float* __restrict__ C, int64_t ldc, int tiles_n) {
    constexpr int WM = BM / 2, WN = BN / 2;      // rows / columns per wave (waves 2 x 2)
    constexpr int NI = WM / 32, NJ = WN / 32;    // 32 x 32 MFMA tiles per wave along m / n (2 or 1 each)
    constexpr int A_PLANE = BM * G3_ROWB, B_PLANE = BN * G3_ROWB;
    constexpr int NLA = BM * (G3_BK / 4) / kBlock;       // fp32 float4 loads per thread and slab: A (4 or 2)
    constexpr int NLB = BN * (G3_BK / 4) / kBlock;       // B (4 or 2)
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * A_PLANE + 3 * B_PLANE];
    unsigned char* const As = lds;
    unsigned char* const Bs = lds + 3 * A_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;      // n fastest: neighbouring blocks share the A panel
    const int m0 = tm * BM, n0 = tn * BN;

    // loader (both operands are k-contiguous fp32): float4 f = tid + 256 i: row = f >> 3, k quad = f & 7 (8 consecutive
    // lanes read 128 contiguous bytes of a row); unconditional at clamped addresses, zeroed at store time when k >= K
    auto gload = [&](const float* base, int64_t ld, int row0, int nrows, int k0, float4* v, int nl) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nl) break;
            const int f = tid + kBlock * i;
            const int row = min(row0 + (f >> 3), nrows - 1), k = k0 + 4 * (f & 7);
            v[i] = ld4(base + (int64_t)row * ld + min(k, K - 4));
        }
    };
    // split in registers, three bf16 planes into LDS
    auto sstore = [&](unsigned char* S, int plane_bytes, const float4* v, int k0, int nl) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nl) break;
            const int f = tid + kBlock * i;
            const float4 x = k0 + 4 * (f & 7) < K ? v[i] : f4zero();
            uint2 h, m, l;
            split4(x, h, m, l);
            unsigned char* p = S + lds_off(f >> 3, (f & 7) >> 1) + ((f & 1) << 3);
            *reinterpret_cast<uint2*>(p) = h;
            *reinterpret_cast<uint2*>(p + plane_bytes) = m;
            *reinterpret_cast<uint2*>(p + 2 * plane_bytes) = l;
        }
    };
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    // 32-wide column blocks of this wave that lie inside the matrix (wave-uniform): a ragged last tile skips the MFMAs and
    // fragment reads of the blocks it does not have (N = 900: the eighth 128-wide tile holds 4 columns = one block of four)
    bool jin[NJ], iin[NI];
#pragma unroll
    for (int j = 0; j < NJ; ++j) jin[j] = n0 + wn * WN + j * 32 < N;
#pragma unroll
    for (int i = 0; i < NI; ++i) iin[i] = m0 + wm * WM + i * 32 < M;

    const int nk = (K + G3_BK - 1) / G3_BK;
    float4 ra[4], rb[4];
    gload(A, lda, m0, M, 0, ra, NLA);
    gload(Bm, ldb, n0, N, 0, rb, NLB);
    for (int kt = 0; kt < nk; ++kt) {
        sstore(As, A_PLANE, ra, kt * G3_BK, NLA);
        sstore(Bs, B_PLANE, rb, kt * G3_BK, NLB);
        if (kt + 1 < nk) {                                   // next slab's loads fly across the barrier and the MFMAs
            gload(A, lda, m0, M, (kt + 1) * G3_BK, ra, NLA);
            gload(Bm, ldb, n0, N, (kt + 1) * G3_BK, rb, NLB);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < G3_BK / 16; ++s) {
            // fragments: lane (r, h) holds k = 16 s + 8 h .. + 7 of row r = chunk 2 s + h: one 16-byte read per plane and tile
            bf16x8 af[NI][3], bf[NJ][3];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[i][p] = *reinterpret_cast<const bf16x8*>(As + p * A_PLANE + lds_off(wm * WM + i * 32 + r, 2 * s + h));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bf[j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * B_PLANE + lds_off(wn * WN + j * 32 + r, 2 * s + h));
            // six term pairs, smallest first
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (!(iin[i] && jin[j])) continue;
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], c, 0, 0, 0);    // lo  * hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], c, 0, 0, 0);    // hi  * lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);    // mid * mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);    // mid * hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);    // hi  * mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);    // hi  * hi
                    acc[i][j] = c;
                }
        }
        __syncthreads();
    }
    // Epilogue through LDS: the MFMA C/D map puts ONE column per lane (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) +
    // 4 (lane >> 5)), i.e. 4-byte stores of 128-byte row pieces; each wave parks a 32 x WN strip of its tile in LDS
    // (the operand planes are dead after the last barrier) and writes it out as 16 bytes per lane, whole 4 WN-byte rows.
    constexpr int CLD = WN + 4;                                   // floats per parked row (+4: the 8 rows a store instruction
    float* const cw = reinterpret_cast<float*>(lds) + wave * 32 * CLD;   //  touches start on different banks)
    static_assert(4 * 32 * CLD * 4 <= 3 * A_PLANE + 3 * B_PLANE, "C strip does not fit the operand LDS");
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (!iin[i]) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                cw[((reg & 3) + 8 * (reg >> 2) + 4 * h) * CLD + j * 32 + r] = jin[j] ? acc[i][j][reg] : 0.f;
        // (same wave wrote and reads: no barrier, only the LDS counter)
        constexpr int C4 = WN / 4;                                 // float4 per strip row
#pragma unroll
        for (int q = 0; q < 32 * C4 / 64; ++q) {
            const int e = lane + 64 * q;
            const int row = e / C4, c4 = e % C4;
            const int64_t m = m0 + wm * WM + i * 32 + row;
            const int n = n0 + wn * WN + c4 * 4;
            const float4 v = *reinterpret_cast<const float4*>(cw + row * CLD + c4 * 4);
            if (m < M) {
                float* o = C + m * ldc + n;
                if (n + 3 < N && (ldc & 3) == 0) st4(o, v);
                else {
                    if (n + 0 < N) o[0] = v.x;
                    if (n + 1 < N) o[1] = v.y;
                    if (n + 2 < N) o[2] = v.z;
                    if (n + 3 < N) o[3] = v.w;
                }
            }
        }
    }
}

}  // namespace

extern "C" {

int jmac_gemm_nt_x3_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t K, float* C,
                        int64_t ldc, jmac_stream_t stream) {
    if (M < 0 || N < 0 || K <= 0) return JMAC_EINVAL;
    if (M == 0 || N == 0) return JMAC_OK;
    if (!A || !B || !C) return JMAC_EINVAL;
    if (K % 4 || lda % 4 || ldb % 4) return JMAC_EDIM;
    if (M >= INT32_MAX || N >= INT32_MAX || K >= INT32_MAX) return JMAC_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    // tile shape: the kernel is whole blocks of MFMA work on k x 256 resident block slots, so its cost is
    // rounds x work per tile; ragged column blocks are skipped at 32-column granularity
    static const int force = getenv("JMAC_G3_TILE") ? atoi(getenv("JMAC_G3_TILE")) : 0;      // tuning knob (debug): BM*1000+BN
    int bm = 128, bn = 128;
    if (force) {
        bm = force / 1000;
        bn = force % 1000;
    } else {
        double best = 1e300;
        const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
        for (auto& c : cand) {
            const int64_t tm = (M + c[0] - 1) / c[0], tn = (N + c[1] - 1) / c[1];
            const int64_t slots = 256 * (c[0] * c[1] >= 128 * 128 ? 3 : (c[0] * c[1] >= 128 * 64 ? 4 : 6));   // blocks per CU (VGPRs / LDS)
            const int64_t nblk32 = (N + 31) / 32;                                  // column blocks actually computed
            const double work = (double)c[0] * (double)nblk32 * 32.0 / (double)tn;   // average columns x rows per tile
            // smaller tiles re-read operands more often and issue fewer MFMAs per fragment: penalise them mildly
            const double eff = c[0] * c[1] >= 128 * 128 ? 1.0 : (c[0] * c[1] >= 128 * 64 ? 0.9 : 0.8);
            const double cost = (double)((tm * tn + slots - 1) / slots) * work / eff;
            if (cost < best) { best = cost; bm = c[0]; bn = c[1]; }
        }
    }
    const int64_t tiles_m = (M + bm - 1) / bm, tiles_n = (N + bn - 1) / bn;
    if (tiles_m * tiles_n >= INT32_MAX) return JMAC_ERANGE;
    const dim3 grid((unsigned)(tiles_m * tiles_n));
#define JMAC_G3_LAUNCH(BMv, BNv, WPEv)                                                                                       \
    hipLaunchKernelGGL((gemm_bf16x3_nt_kernel<BMv, BNv, WPEv>), grid, dim3(kBlock), 0, st, A, lda, B, ldb, (int)M, (int)N, \
                       (int)K, C, ldc, (int)tiles_n)
    if (bm == 128 && bn == 128) JMAC_G3_LAUNCH(128, 128, 3);
    else if (bm == 128 && bn == 64) JMAC_G3_LAUNCH(128, 64, 4);
    else if (bm == 64 && bn == 128) JMAC_G3_LAUNCH(64, 128, 4);
    else JMAC_G3_LAUNCH(64, 64, 6);
#undef JMAC_G3_LAUNCH
    return (int)hipGetLastError();
}

}  // extern "C"
