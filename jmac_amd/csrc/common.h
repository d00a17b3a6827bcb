// common.h -- shared device helpers for libjmac_hip (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jmac_hip.h"

namespace jmac {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- cross-lane reductions -----------------------------------------------------------------------
// xor-butterfly inside each 16-lane row with 4 fused v_*_dpp, then the gfx9 row_bcast:15 / row_bcast:31
// steps fold the four rows into lane 63, which v_readlane broadcasts through an SGPR.
// (v_permlane{16,32}_swap would also do the cross-row steps, but hipcc 7.2 folds r[0] op r[1] of
//  permlane*_swap(x, x) into x op x -- wrong results -- so it is not used.)
#define JMAC_DPP(v, old, ctrl, rmask)                                                                   \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)),       \
                                                          __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))

// every lane receives the sum over the 64 lanes (wave-uniform result)
__device__ __forceinline__ float wave_sum(float v) {
    v += JMAC_DPP(v, 0.f, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    v += JMAC_DPP(v, 0.f, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    v += JMAC_DPP(v, 0.f, 0x141, 0xF);   // row_half_mirror
    v += JMAC_DPP(v, 0.f, 0x140, 0xF);   // row_mirror      -> every lane holds its row's sum
    v += JMAC_DPP(v, 0.f, 0x142, 0xA);   // row_bcast:15 into rows 1,3
    v += JMAC_DPP(v, 0.f, 0x143, 0xC);   // row_bcast:31 into rows 2,3 -> lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, JMAC_DPP(v, -INFINITY, 0xB1, 0xF));
    v = fmaxf(v, JMAC_DPP(v, -INFINITY, 0x4E, 0xF));
    v = fmaxf(v, JMAC_DPP(v, -INFINITY, 0x141, 0xF));
    v = fmaxf(v, JMAC_DPP(v, -INFINITY, 0x140, 0xF));
    v = fmaxf(v, JMAC_DPP(v, -INFINITY, 0x142, 0xA));
    v = fmaxf(v, JMAC_DPP(v, -INFINITY, 0x143, 0xC));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// wave-uniform broadcast of lane `src` (src must be wave-uniform)
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float bcast_f(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : x * slope; }
// identical to leaky() whenever 0 <= slope <= 1 (x>0: x >= slope*x; x<0: slope*x >= x); 2 VALU ops instead of 3
__device__ __forceinline__ float leaky01(float x, float slope) { return fmaxf(x, x * slope); }
// exp through v_exp_f32 (2^x): relative error ~1e-6 for the |x| <~ 30 softmax arguments used here
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// U independent wave sums with the DPP steps interleaved, so the VALU->DPP hazard slots of one chain are
// filled by the others instead of s_nop
template <int U>
__device__ __forceinline__ void wave_sum_n(float (&v)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0xB1, 0xF);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0x4E, 0xF);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0x141, 0xF);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0x140, 0xF);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0x142, 0xA);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0x143, 0xC);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[u]), 63));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// non-temporal (streaming) 16-byte load: for rows that are read once, so they do not evict the relation table
typedef float jmac_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
    const jmac_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const jmac_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
// ---- table element types: float, or bf16 stored as uint16_t (inference tables; arithmetic stays fp32) -------
// A lane's chunk is 4 consecutive elements either way: 16 B of fp32 or 8 B of bf16, so both table types share one
// lane -> element map.  Gathered chunks stay in their raw form (4 or 2 VGPRs) until the arithmetic needs them.
typedef uint16_t bf16_t;
__device__ __forceinline__ float4 ldraw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 ldraw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 cvt4(float4 r) { return r; }
__device__ __forceinline__ float4 cvt4(uint2 r) {   // bf16 -> fp32 is a 16-bit shift: exact
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}
template <typename TT> struct RawOf { typedef float4 type; };
template <> struct RawOf<bf16_t> { typedef uint2 type; };

__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// out[c] = scale * sum_p partial[p*W + c], c in [0,W): deterministic, parallel over columns and row lanes
// (defined in aggregate.hip; used for the a_att gradient, the self-loop column sum and the BN statistics)
void launch_reduce_rows(const float* partial, int nparts, int W, float scale, float* out, hipStream_t st);

}  // namespace jmac
