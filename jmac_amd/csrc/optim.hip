// optim.hip -- the Adam update of the training step (train.py:406-407 build torch.optim.Adam(model.parameters(), lr); :358-359 step it
// once per batch) as ONE launch over every parameter tensor of the model, sized to fill the chip.
//
// Why it exists: torch's fused multi-tensor Adam cuts the tensors into 65 536-element chunks, one 512-thread workgroup each -- 125
// workgroups for the 6.3 M parameters of the DBP-5L model, i.e. fewer than half of the 256 CUs carry any work (50 us = 3.5 TB/s for
// 176 MB of reads + writes).  Here a chunk is 4 096 elements (four float4 per thread and array, 256 threads): ~1 700 workgroups.
//
// The step count lives on the device (a captured step replays with the right bias corrections), and beside it the running powers
// beta1^step, beta2^step in double (`aux`): a workgroup's prologue is two loads and a few double operations instead of two pow()
// (measured: with pow() per workgroup the 1 024-element-chunk form took 86 us, the 4 096-element one 36).  Every workgroup reads
// them before it touches anything else; the LAST workgroup to finish (a relaxed ticket -- no data visibility is needed, only the
// count) stores step + 1 and the next powers and resets the ticket; the next launch sees them through the launch boundary.
#include "common.h"

namespace {

constexpr int kAdamBlock = 256;

struct AdamArgs {
    jmac_adam_task_t t[JMAC_ADAM_MAX_TASKS];
    int chunk_end[JMAC_ADAM_MAX_TASKS];              // exclusive prefix end of task i's chunks
    int n_tasks, n_chunks;
    double lr, beta1, beta2, eps, weight_decay;
    int decoupled, maximize, advance;
    float* step;                                     // [1] device: number of completed steps (float, as torch keeps it)
    double* aux;                                     // [3] device: beta1^step, beta2^step, ticket word (zero at rest)
};

struct AdamConst {
    float one_minus_b1, b2, one_minus_b2, step_size, inv_bc2_sqrt, eps, wd, lr_wd;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamConst& c, bool decoupled, bool maximize) {
    if (maximize) g = -g;
    if (c.wd != 0.f) {
        if (decoupled) p = p - c.lr_wd * p;          // AdamW: p *= 1 - lr * wd
        else g = g + c.wd * p;                       // Adam: L2 term joins the gradient
    }
    m = m + (g - m) * c.one_minus_b1;                // lerp(m, g, 1 - beta1)
    v = v * c.b2 + c.one_minus_b2 * g * g;
    const float denom = __fsqrt_rn(v) * c.inv_bc2_sqrt + c.eps;
    p = p - c.step_size * (m / denom);
}

// one workgroup = one chunk of V float4 per thread and array (V * 1024 elements); the grid is the chunk list, so the hardware
// balances the tail.
template <int V>
__global__ __launch_bounds__(kAdamBlock) void adam_step_kernel(const AdamArgs a) {
    constexpr int kChunk = kAdamBlock * 4 * V;
    const int ch = blockIdx.x;
    int task = 0;
    if (a.n_chunks > 0) {
        int lo = 0, hi = a.n_tasks - 1;                                 // first task whose chunk range ends past ch
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ch >= a.chunk_end[mid]) lo = mid + 1; else hi = mid;
        }
        task = lo;
    }
    const jmac_adam_task_t T = a.t[task];
    const int first = task ? a.chunk_end[task - 1] : 0;
    const int64_t base = (int64_t)(ch - first) * kChunk;
    const int64_t left = a.n_chunks > 0 ? T.n - base : 0;
    const bool full = left >= kChunk && T.vec4;
    float4 p[V], g[V], m[V], v[V];
    if (full) {
        const float4* p4 = reinterpret_cast<const float4*>(T.p + base);
        const float4* g4 = reinterpret_cast<const float4*>(T.g + base);
        const float4* m4 = reinterpret_cast<const float4*>(T.m + base);
        const float4* v4 = reinterpret_cast<const float4*>(T.v + base);
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const int i = threadIdx.x + u * kAdamBlock;
            p[u] = p4[i]; g[u] = g4[i]; m[u] = m4[i]; v[u] = v4[i];
        }
    }
    // bias corrections in double, like torch (1 - beta^t loses its digits in float: 1 - 0.999 is 1e-3 +- 6e-8); t = step + 1
    AdamConst c;
    {
        const double bc1 = 1.0 - a.aux[0] * a.beta1, bc2 = 1.0 - a.aux[1] * a.beta2;
        c.one_minus_b1 = (float)(1.0 - a.beta1);
        c.b2 = (float)a.beta2;
        c.one_minus_b2 = (float)(1.0 - a.beta2);
        c.step_size = (float)(a.lr / bc1);
        c.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
        c.eps = (float)a.eps;
        c.wd = (float)a.weight_decay;
        c.lr_wd = (float)(a.lr * a.weight_decay);
    }
    const bool dec = a.decoupled != 0, mx = a.maximize != 0;
    if (full) {
        float4* p4 = reinterpret_cast<float4*>(T.p + base);
        float4* m4 = reinterpret_cast<float4*>(T.m + base);
        float4* v4 = reinterpret_cast<float4*>(T.v + base);
#pragma unroll
        for (int u = 0; u < V; ++u) {
            adam_elem(p[u].x, g[u].x, m[u].x, v[u].x, c, dec, mx);
            adam_elem(p[u].y, g[u].y, m[u].y, v[u].y, c, dec, mx);
            adam_elem(p[u].z, g[u].z, m[u].z, v[u].z, c, dec, mx);
            adam_elem(p[u].w, g[u].w, m[u].w, v[u].w, c, dec, mx);
        }
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const int i = threadIdx.x + u * kAdamBlock;
            p4[i] = p[u]; m4[i] = m[u]; v4[i] = v[u];
        }
    } else {                                                            // a tensor's tail chunk / unaligned tensors
        const int64_t n = left < kChunk ? left : kChunk;
        for (int64_t i = threadIdx.x; i < n; i += kAdamBlock) {
            float pp = T.p[base + i], mm = T.m[base + i], vv = T.v[base + i];
            adam_elem(pp, T.g[base + i], mm, vv, c, dec, mx);
            T.p[base + i] = pp; T.m[base + i] = mm; T.v[base + i] = vv;
        }
    }
    if (a.advance) {
        __syncthreads();                                                // every wave of this workgroup has read aux
        if (threadIdx.x == 0) {
            unsigned int* ticket = reinterpret_cast<unsigned int*>(a.aux + 2);
            const unsigned int done = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done == gridDim.x - 1) {                                // all workgroups hold their constants
                a.step[0] = a.step[0] + 1.f;
                a.aux[0] = a.aux[0] * a.beta1;
                a.aux[1] = a.aux[1] * a.beta2;
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int V>
int launch_adam(const jmac_adam_task_t* tasks, int32_t n_tasks, float* step, double* aux, double lr, double beta1, double beta2,
                double eps, double weight_decay, int32_t decoupled, int32_t maximize, hipStream_t st) {
    constexpr int kChunk = kAdamBlock * 4 * V;
    // launches of up to JMAC_ADAM_MAX_TASKS tensors; the step count advances in the last one (no tensors at all: it still advances)
    int32_t at = 0;
    do {
        AdamArgs a;
        const int32_t n = (n_tasks - at) < JMAC_ADAM_MAX_TASKS ? (n_tasks - at) : JMAC_ADAM_MAX_TASKS;
        int64_t chunks = 0;
        for (int32_t i = 0; i < n; ++i) {
            const jmac_adam_task_t& t = tasks[at + i];
            if (t.n < 0 || (t.n > 0 && (!t.p || !t.g || !t.m || !t.v))) return JMAC_EINVAL;
            a.t[i] = t;
            a.t[i].vec4 = !(((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15);
            chunks += (t.n + kChunk - 1) / kChunk;
            if (chunks >= INT32_MAX) return JMAC_ERANGE;
            a.chunk_end[i] = (int)chunks;
        }
        for (int32_t i = n; i < JMAC_ADAM_MAX_TASKS; ++i) {
            a.t[i] = jmac_adam_task_t{nullptr, nullptr, nullptr, nullptr, 0, 0};
            a.chunk_end[i] = INT32_MAX;
        }
        a.n_tasks = n > 0 ? n : 1;
        a.n_chunks = (int)chunks;
        a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
        a.decoupled = decoupled; a.maximize = maximize;
        at += n;
        a.advance = at >= n_tasks;
        a.step = step;
        a.aux = aux;
        hipLaunchKernelGGL(adam_step_kernel<V>, dim3((unsigned)(chunks < 1 ? 1 : chunks)), dim3(kAdamBlock), 0, st, a);
        const int rc = (int)hipGetLastError();
        if (rc) return rc;
    } while (at < n_tasks);
    return JMAC_OK;
}

}  // namespace

extern "C" int jmac_adam_step_f32(const jmac_adam_task_t* tasks, int32_t n_tasks, float* step, double* aux, double lr, double beta1,
                                  double beta2, double eps, double weight_decay, int32_t decoupled, int32_t maximize,
                                  jmac_stream_t stream) {
    if (n_tasks < 0 || !step || !aux || ((uintptr_t)aux & 7) || (n_tasks > 0 && !tasks)) return JMAC_EINVAL;
    if (!(lr >= 0.0) || !(eps >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(weight_decay >= 0.0))
        return JMAC_EINVAL;
    for (int32_t i = 0; i < n_tasks; ++i)
        if (tasks[i].n < 0 || (tasks[i].n > 0 && (!tasks[i].p || !tasks[i].g || !tasks[i].m || !tasks[i].v))) return JMAC_EINVAL;
    // V = 4 (4 096-element chunks): measured on the headline model's 39 tensors / 6.9 M parameters (tools/r5_adam_probe.py,
    // profiles/r5_adam.txt) 35.7 us against 85.6 / 49.0 / 38.4 us at V = 1 / 2 / 8 -- the ticket is one same-address atomic per
    // workgroup (~8 ns each, serialised): 1 690 of them hide behind the stream, 6 760 do not (without the ticket: 31 / 31 / 33 us)
    return launch_adam<4>(tasks, n_tasks, step, aux, lr, beta1, beta2, eps, weight_decay, decoupled, maximize, (hipStream_t)stream);
}
