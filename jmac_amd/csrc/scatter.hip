// scatter.hip -- torch_scatter-compatible primitives for gfx950 (replace the third-party calls at
// src/jmac_model.py:105 and modules/helper/message_passing.py:24,28), so that the UNMODIFIED reference
// layer can run on this library.  Unsorted indices, float atomics: this is the compatibility path, not
// the fused fast path of aggregate.hip.
#include "common.h"

using namespace jmac;

namespace {

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    // monotone int mapping of IEEE floats: non-negative -> signed max, negative -> unsigned min
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void scatter_sum_kernel(const float* __restrict__ src, const int64_t* __restrict__ index, int64_t E, int64_t d,
                                   int64_t N, float* __restrict__ out) {
    const int64_t total = E * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / d, c = i % d;
        const int64_t r = index[e];
        if (r >= 0 && r < N) atomicAdd(out + r * d + c, src[i]);
    }
}

__global__ void fill_f32_kernel(float* __restrict__ p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void seg_max_kernel(const float* __restrict__ src, const int64_t* __restrict__ index, int64_t E, int64_t d,
                               int64_t N, float* __restrict__ mx) {
    const int64_t total = E * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / d, c = i % d;
        const int64_t r = index[e];
        if (r >= 0 && r < N) atomic_max_f32(mx + r * d + c, src[i]);
    }
}

__global__ void seg_expsum_kernel(const float* __restrict__ src, const int64_t* __restrict__ index, int64_t E, int64_t d,
                                  int64_t N, const float* __restrict__ mx, float* __restrict__ sum, float* __restrict__ out) {
    const int64_t total = E * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / d, c = i % d;
        const int64_t r = index[e];
        if (r >= 0 && r < N) {
            const float ex = expf(src[i] - mx[r * d + c]);
            out[i] = ex;
            atomicAdd(sum + r * d + c, ex);
        }
    }
}

__global__ void seg_div_kernel(const int64_t* __restrict__ index, int64_t E, int64_t d, int64_t N,
                               const float* __restrict__ sum, float* __restrict__ out) {
    const int64_t total = E * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / d, c = i % d;
        const int64_t r = index[e];
        if (r >= 0 && r < N) out[i] = out[i] / sum[r * d + c];
    }
}

inline unsigned grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g < 1) g = 1;
    return (unsigned)(g < 16384 ? g : 16384);
}

}  // namespace

extern "C" {

int jmac_scatter_sum_f32(const float* src, const int64_t* index, int64_t E, int64_t d, int64_t N, float* out,
                         jmac_stream_t stream) {
    if (E < 0 || d <= 0 || N < 0) return JMAC_EINVAL;
    if (E == 0) return JMAC_OK;
    if (!src || !index || !out) return JMAC_EINVAL;
    hipLaunchKernelGGL(scatter_sum_kernel, dim3(grid_for(E * d)), dim3(256), 0, (hipStream_t)stream, src, index, E, d, N, out);
    return (int)hipGetLastError();
}

size_t jmac_scatter_softmax_workspace_bytes(int64_t N, int64_t d) {
    if (N < 0 || d < 0) return 0;
    return 2 * align_up((size_t)N * (size_t)d * 4) + 256;
}

int jmac_scatter_softmax_f32(const float* src, const int64_t* index, int64_t E, int64_t d, int64_t N, float* out, void* ws,
                             size_t ws_bytes, jmac_stream_t stream) {
    if (E < 0 || d <= 0 || N < 0) return JMAC_EINVAL;
    if (E == 0) return JMAC_OK;
    if (!src || !index || !out) return JMAC_EINVAL;
    if (!ws || ws_bytes < jmac_scatter_softmax_workspace_bytes(N, d)) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* mx = (float*)ws;
    float* sum = (float*)((char*)ws + align_up((size_t)N * (size_t)d * 4));
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(N * d)), dim3(256), 0, st, mx, N * d, -INFINITY);
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(N * d)), dim3(256), 0, st, sum, N * d, 0.f);
    hipLaunchKernelGGL(seg_max_kernel, dim3(grid_for(E * d)), dim3(256), 0, st, src, index, E, d, N, mx);
    hipLaunchKernelGGL(seg_expsum_kernel, dim3(grid_for(E * d)), dim3(256), 0, st, src, index, E, d, N, mx, sum, out);
    hipLaunchKernelGGL(seg_div_kernel, dim3(grid_for(E * d)), dim3(256), 0, st, index, E, d, N, sum, out);
    return (int)hipGetLastError();
}

const char* jmac_strerror(int rc) {
    switch (rc) {
        case JMAC_OK: return "ok";
        case JMAC_EINVAL: return "jmac: invalid argument (null pointer or negative size)";
        case JMAC_EDIM: return "jmac: unsupported dimension (need d % 4 == 0, d <= 512, ld % 4 == 0)";
        case JMAC_EWORKSPACE: return "jmac: workspace missing or too small";
        case JMAC_ERANGE: return "jmac: size exceeds int32 indexing";
        default: break;
    }
    if (rc > 0) return hipGetErrorString((hipError_t)rc);
    return "jmac: unknown error";
}

// 100: rounds 1-2;  110: round 3 changed jmac_sim_topk_workspace_bytes to (L, N, k);  120: round 4 additions (segmented BatchNorm,
// padded bf16 aggregation, exact margin / sorted cosine adjoints; jmac_gemm_nt_x3_f32 moved to the testing library);
// 121: jmac_softmax_parts_merge_f32 added
int jmac_version(void) { return 124; }

}  // extern "C"
