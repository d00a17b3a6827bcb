#include "common.h"
#include <cstdio>
#include <vector>
using namespace jmac;
__global__ void k(const float* in, float* o1, float* o2, float* o3) {
    float v = in[threadIdx.x];
    o1[threadIdx.x] = wave_sum(v);
    float s = v;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    o2[threadIdx.x] = s;
    o3[threadIdx.x] = wave_max(v);
}
int main() {
    float *in, *o1, *o2, *o3;
    hipMalloc(&in, 256); hipMalloc(&o1, 256); hipMalloc(&o2, 256); hipMalloc(&o3, 256);
    for (int trial = 0; trial < 3; ++trial) {
        std::vector<float> h(64);
        for (int i = 0; i < 64; ++i) h[i] = trial == 0 ? (i < 2 ? i + 1.f : 0.f) : (trial == 1 ? (float)(1 << (i % 20)) * (i + 1) : (float)((i * 37) % 11) - 5.f);
        hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, o1, o2, o3);
        std::vector<float> a(64), b(64), c(64);
        hipMemcpy(a.data(), o1, 256, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), o2, 256, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), o3, 256, hipMemcpyDeviceToHost);
        double ref = 0; float mx = -1e30; for (float x : h) { ref += x; mx = x > mx ? x : mx; }
        printf("trial %d ref %.1f max %.1f\n wave_sum:", trial, ref, mx);
        for (int i = 0; i < 64; i += 7) printf(" %.1f", a[i]);
        printf("\n shfl_sum:");
        for (int i = 0; i < 64; i += 7) printf(" %.1f", b[i]);
        printf("\n wave_max:");
        for (int i = 0; i < 64; i += 7) printf(" %.1f", c[i]);
        printf("\n");
    }
}
