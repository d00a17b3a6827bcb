// common.h -- shared device helpers for libjmac_hip (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jmac_hip.h"

namespace jmac {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- cross-lane reductions: 4 fused v_add_f32_dpp + v_permlane16_swap + v_permlane32_swap --------
#define JMAC_DPP_ADD(v, ctrl)                                                                            \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
#define JMAC_DPP_MAX(v, ctrl)                                                                            \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true)))

// all 64 lanes receive the sum
__device__ __forceinline__ float wave_sum(float v) {
    JMAC_DPP_ADD(v, 0xB1);    // quad_perm [1,0,3,2]
    JMAC_DPP_ADD(v, 0x4E);    // quad_perm [2,3,0,1]
    JMAC_DPP_ADD(v, 0x141);   // row_half_mirror
    JMAC_DPP_ADD(v, 0x140);   // row_mirror
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    v = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    v = __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, q[1]);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    JMAC_DPP_MAX(v, 0xB1);
    JMAC_DPP_MAX(v, 0x4E);
    JMAC_DPP_MAX(v, 0x141);
    JMAC_DPP_MAX(v, 0x140);
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    v = fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
    auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    v = fmaxf(__builtin_bit_cast(float, q[0]), __builtin_bit_cast(float, q[1]));
    return v;
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

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace jmac
