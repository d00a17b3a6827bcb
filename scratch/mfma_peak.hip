// fp32-input MFMA issue-rate probe: back-to-back MFMAs on register operands, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    if (KIND == 0) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        f32x4 c[8] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
        float s = 0;
        for (int j = 0; j < 8; ++j) s += c[j][0];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind)
        for (int blocks : {256, 512, 1024, 2048}) {
            const int iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 4 * iters * (kind == 0 ? 4 * 4096.0 : 8 * 2048.0);
            printf("%s blocks=%d (%d waves/SIMD): %.3f ms  %.1f TFLOP/s\n", kind == 0 ? "32x32x2 " : "16x16x4 ", blocks, blocks / 256, ms, flop / ms / 1e9);
        }
    return 0;
}
