// graph.hip -- graph ingest for the JMAC hot path on gfx950.
//
// COO edge lists (the reference's edge_index [2,E] / edge_type [E] int64, train.py:116-135 and
// src/utils.py:112-149) -> CSR by aggregation destination, plus the by-source / by-relation groupings
// and the wave-sized work schedules the aggregation kernels run over.  The degree the reference gets
// from scatter_add (src/jmac_model.py:103-108) is rowptr[i+1]-rowptr[i] here.
//
// Sorting is rocPRIM's device radix sort (stable), so the edge order inside a row -- and therefore
// the floating-point summation order of every kernel downstream -- is fixed by the input order.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {

__global__ void coo_keys_kernel(const int64_t* __restrict__ edge_index, int64_t E, int32_t* __restrict__ keys,
                                int32_t* __restrict__ iota) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) {
        keys[e] = (int32_t)edge_index[e];   // row 0 = destination
        iota[e] = (int32_t)e;
    }
}

// keys[s] = (int32) src[perm[s]]   (second sort pass: the destination of every type-ordered edge)
__global__ void gather_keys_kernel(const int64_t* __restrict__ src, const int32_t* __restrict__ perm, int64_t E,
                                   int32_t* __restrict__ keys) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < E) keys[s] = (int32_t)src[perm[s]];
}
__global__ void type_keys_kernel(const int64_t* __restrict__ edge_type, int64_t E, int32_t* __restrict__ keys,
                                 int32_t* __restrict__ iota) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) {
        keys[e] = (int32_t)edge_type[e];
        iota[e] = (int32_t)e;
    }
}

// bad[0] += number of entries outside [lo, hi)  (the reference raises IndexError on such an id; the kernels trust them)
template <typename IT>
__global__ void index_check_kernel(const IT* __restrict__ idx, int64_t n, int64_t lo, int64_t hi, int32_t* __restrict__ bad) {
    int cnt = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = (int64_t)idx[e];
        cnt += (v < lo || v >= hi) ? 1 : 0;
    }
    cnt = jmac::wave_sum_i(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(bad, cnt);
}

__global__ void iota_kernel(int64_t E, int32_t* __restrict__ iota) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) iota[e] = (int32_t)e;
}

// ptr[r] = first position whose key is >= r, for r in [0, S]; keys sorted ascending.
__global__ void ptr_from_sorted_kernel(const int32_t* __restrict__ keys, int64_t E, int64_t S,
                                       int32_t* __restrict__ ptr) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (E == 0) {
        for (int64_t r = e; r <= S; r += (int64_t)gridDim.x * blockDim.x) ptr[r] = 0;
        return;
    }
    if (e >= E) return;
    int64_t k0 = (e == 0) ? -1 : keys[e - 1];
    int64_t k1 = keys[e];
    for (int64_t r = k0 + 1; r <= k1; ++r) ptr[r] = (int32_t)e;
    if (e == E - 1)
        for (int64_t r = k1 + 1; r <= S; ++r) ptr[r] = (int32_t)E;
}

__global__ void csr_gather_kernel(const int64_t* __restrict__ edge_index, const int64_t* __restrict__ edge_type,
                                  const int32_t* __restrict__ perm, int64_t E, int32_t* __restrict__ col,
                                  int32_t* __restrict__ etype) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < E) {
        int64_t e = perm[s];
        col[s] = (int32_t)edge_index[E + e];   // row 1 = message source
        etype[s] = (int32_t)edge_type[e];
    }
}

// Segment classes of a schedule:
//   empty   no entries
//   plain   1 .. chunk entries, one item, finalised by its wave
//   coop    coop_min < len <= coop_max (small graphs only; coop_max = 0 disables): exactly kCoop = 4 items of
//           ceil(len / 4) entries, the four waves of ONE workgroup; the forward kernel merges their partial softmax
//           states through LDS (no combine launch)
//   split   len > chunk: ceil(len / chunk) items of <= chunk entries, merged by the combine pass
// Item order:  [coop: 4 per segment][split items][plain segments, in segment order][empty segments, in order].
// coop first: block b of a one-wave-per-item launch owns coop segment b (its four item slots are block aligned), and
// the longest rows of the graph start first.  Split items next (longest-first balancing under the round-robin
// item->wave map).  Empty segments (DBP-5L train graphs: 54 % of the destinations) last: the forward kernel hands
// several of them to one wave -- they need no gather, only the self term.
constexpr int kCoop = 4;

__device__ __forceinline__ int seg_class(int32_t len, int32_t chunk, int32_t coop_min, int32_t coop_max) {
    if (len == 0) return 0;                                        // empty
    if (coop_max > 0 && len > coop_min && len <= coop_max) return 2;   // coop
    return len <= chunk ? 1 : 3;                                   // plain : split
}

__global__ void seg_counts_kernel(const int32_t* __restrict__ ptr, int64_t S, int32_t chunk, int32_t coop_min, int32_t coop_max,
                                  int32_t* __restrict__ nempty, int32_t* __restrict__ ncoop, int32_t* __restrict__ npart,
                                  int32_t* __restrict__ nsplit) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) {
        const int32_t len = ptr[s + 1] - ptr[s];
        const int c = seg_class(len, chunk, coop_min, coop_max);
        nempty[s] = c == 0;
        ncoop[s] = c == 2;
        npart[s] = c == 3 ? (len + chunk - 1) / chunk : 0;          // partial slots of the (non-coop) split segments
        nsplit[s] = c == 3;
    }
}

__global__ void items_fill_kernel(const int32_t* __restrict__ ptr, int64_t S, int32_t chunk, int32_t coop_min, int32_t coop_max,
                                  const int32_t* __restrict__ empty_off, const int32_t* __restrict__ coop_off,
                                  const int32_t* __restrict__ part_off, const int32_t* __restrict__ split_off,
                                  jmac_item_t* __restrict__ items, jmac_split_t* __restrict__ splits,
                                  int32_t* __restrict__ counts) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    // totals from the last segment (every thread reads the same words)
    const int32_t last_len = ptr[S] - ptr[S - 1];
    const int lc = seg_class(last_len, chunk, coop_min, coop_max);
    const int32_t n_empty = empty_off[S - 1] + (lc == 0);
    const int32_t n_coop = coop_off[S - 1] + (lc == 2);
    const int32_t n_parts_nc = part_off[S - 1] + (lc == 3 ? (last_len + chunk - 1) / chunk : 0);
    const int32_t n_splits_nc = split_off[S - 1] + (lc == 3);
    const int32_t n_parts = kCoop * n_coop + n_parts_nc;
    const int32_t n_plain = (int32_t)S - n_coop - n_splits_nc - n_empty;
    const int32_t beg = ptr[s], end = ptr[s + 1];
    const int32_t len = end - beg;
    const int c = seg_class(len, chunk, coop_min, coop_max);
    if (c == 0) {
        items[n_parts + n_plain + empty_off[s]] = jmac_item_t{(int32_t)s, beg, end, -1};
    } else if (c == 1) {
        items[n_parts + (int32_t)s - coop_off[s] - split_off[s] - empty_off[s]] = jmac_item_t{(int32_t)s, beg, end, -1};
    } else if (c == 2) {
        const int32_t po = kCoop * coop_off[s];
        const int32_t q = (len + kCoop - 1) / kCoop;
        for (int32_t w = 0; w < kCoop; ++w) {
            const int32_t b = beg + w * q < end ? beg + w * q : end;
            const int32_t e = b + q < end ? b + q : end;
            items[po + w] = jmac_item_t{(int32_t)s, b, e, po + w};
        }
        splits[coop_off[s]] = jmac_split_t{(int32_t)s, po, kCoop, 1};
    } else {
        const int32_t nch = (len + chunk - 1) / chunk;
        const int32_t po = kCoop * n_coop + part_off[s];
        for (int32_t k = 0; k < nch; ++k) {
            const int32_t b = beg + k * chunk;
            const int32_t e = b + chunk < end ? b + chunk : end;
            items[po + k] = jmac_item_t{(int32_t)s, b, e, po + k};
        }
        splits[n_coop + split_off[s]] = jmac_split_t{(int32_t)s, po, nch, 0};
    }
    if (s == S - 1) {
        counts[0] = n_parts + n_plain + n_empty;
        counts[1] = n_coop + n_splits_nc;
        counts[2] = n_parts;
        counts[3] = n_empty;
        counts[4] = n_coop;
        counts[5] = counts[6] = counts[7] = 0;
    }
}

// edges[it] = {col[beg], etype[beg], col[beg+1], etype[beg+1]} of item it (-1 where the item has fewer entries): the first
// two entries of an item inline with its header, so that a wave that owns ONE short item (small graphs: one wave per item)
// starts its row gathers one dependent memory round trip earlier (header -> gathers instead of header -> col/type -> gathers)
__global__ void item_edges_kernel(const jmac_item_t* __restrict__ items, const int32_t* __restrict__ counts,
                                  const int32_t* __restrict__ col, const int32_t* __restrict__ etype,
                                  int4* __restrict__ edges) {
    const int n_items = counts[0];
    for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += gridDim.x * blockDim.x) {
        const jmac_item_t x = items[it];
        int4 e = make_int4(-1, -1, -1, -1);
        if (x.end > x.beg) {
            e.x = col[x.beg];
            e.y = etype[x.beg];
        }
        if (x.end > x.beg + 1) {
            e.z = col[x.beg + 1];
            e.w = etype[x.beg + 1];
        }
        edges[it] = e;
    }
}

__global__ void zero_counts_kernel(int32_t* counts) {
    if (threadIdx.x < 8) counts[threadIdx.x] = 0;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

size_t sort_temp_bytes(int64_t E) {
    size_t bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs<rocprim::default_config, int32_t*, int32_t*, int32_t*, int32_t*>(
        nullptr, bytes, nullptr, nullptr, nullptr, nullptr, (size_t)E, 0, 32, 0);
    if (e != hipSuccess || bytes == 0) bytes = (size_t)E * 16 + (8u << 20);   // no device: generous bound
    return align_up(bytes);
}

size_t scan_temp_bytes(int64_t S) {
    size_t bytes = 0;
    hipError_t e = rocprim::exclusive_scan<rocprim::default_config, int32_t*, int32_t*, int32_t, rocprim::plus<int32_t>>(
        nullptr, bytes, nullptr, nullptr, 0, (size_t)S, rocprim::plus<int32_t>(), 0);
    if (e != hipSuccess || bytes == 0) bytes = (size_t)S * 4 + (1u << 20);
    return align_up(bytes);
}

// stable sort of (keys, iota) -> (keys_sorted, order)
int sort_by_key(int32_t* keys_in, int32_t* keys_out, int32_t* vals_in, int32_t* vals_out, int64_t E, int64_t S,
                void* tmp, size_t tmp_bytes, hipStream_t st) {
    if (E == 0) return 0;
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)E, 0, 32, st);
    if (e != hipSuccess) return (int)e;
    if (need > tmp_bytes) return JMAC_EWORKSPACE;
    unsigned end_bit = 1;
    while (end_bit < 32 && ((int64_t)1 << end_bit) < S) ++end_bit;
    e = rocprim::radix_sort_pairs(tmp, need, keys_in, keys_out, vals_in, vals_out, (size_t)E, 0, end_bit, st);
    return (int)e;
}

}  // namespace

extern "C" {

size_t jmac_graph_workspace_bytes(int64_t E, int64_t S) {
    if (E < 0) E = 0;
    if (S < 0) S = 0;
    size_t a = 3 * align_up((size_t)E * 4) + sort_temp_bytes(E);
    size_t b = 8 * align_up((size_t)(S + 1) * 4) + scan_temp_bytes(S + 1);   // 4 flag arrays + 4 scans
    return (a > b ? a : b) + 1024;
}

int jmac_index_check(const void* idx, int32_t elem_bytes, int64_t n, int64_t lo, int64_t hi, int32_t* bad,
                     jmac_stream_t stream) {
    if (n < 0 || !bad || (n > 0 && !idx) || (elem_bytes != 4 && elem_bytes != 8)) return JMAC_EINVAL;
    if (n == 0) return JMAC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int T = 256;
    int64_t nb = (n + T - 1) / T;
    if (nb > 2048) nb = 2048;
    if (elem_bytes == 8)
        hipLaunchKernelGGL(index_check_kernel<int64_t>, dim3((unsigned)nb), dim3(T), 0, st, (const int64_t*)idx, n, lo, hi, bad);
    else
        hipLaunchKernelGGL(index_check_kernel<int32_t>, dim3((unsigned)nb), dim3(T), 0, st, (const int32_t*)idx, n, lo, hi, bad);
    return (int)hipGetLastError();
}

int jmac_csr_build(const int64_t* edge_index, const int64_t* edge_type, int64_t E, int64_t N, int64_t nrel, int32_t* rowptr,
                   int32_t* col, int32_t* etype, int32_t* perm, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    if (E < 0 || N < 0 || !rowptr || (E > 0 && (!edge_index || !edge_type || !col || !etype || !perm))) return JMAC_EINVAL;
    if (E >= INT32_MAX || N >= INT32_MAX) return JMAC_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    const int T = 256;
    if (E == 0) {
        hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3((unsigned)((N + 1 + T - 1) / T)), dim3(T), 0, st, nullptr, 0, N, rowptr);
        return (int)hipGetLastError();
    }
    size_t arr = align_up((size_t)E * 4);
    if (!ws || ws_bytes < 3 * arr) return JMAC_EWORKSPACE;
    char* w = (char*)ws;
    int32_t* keys = (int32_t*)w;
    int32_t* keys_sorted = (int32_t*)(w + arr);
    int32_t* iota = (int32_t*)(w + 2 * arr);
    void* tmp = w + 3 * arr;
    unsigned nb = (unsigned)((E + T - 1) / T);
    int rc;
    if (nrel > 0) {
        // order inside a row: by relation type, then input order -- two stable passes (type, then destination).  Edges of
        // one destination that share a relation become neighbours, so their [Rq|Rz] row is fetched once per group of
        // gathers instead of once per edge (the repeats hit L1), and a hub's tail relations are touched in one burst.
        hipLaunchKernelGGL(type_keys_kernel, dim3(nb), dim3(T), 0, st, edge_type, E, keys, iota);
        rc = sort_by_key(keys, keys_sorted, iota, perm, E, nrel, tmp, ws_bytes - 3 * arr, st);        // perm = order by type
        if (rc) return rc;
        hipLaunchKernelGGL(gather_keys_kernel, dim3(nb), dim3(T), 0, st, edge_index, perm, E, keys);
        rc = sort_by_key(keys, keys_sorted, perm, iota, E, N, tmp, ws_bytes - 3 * arr, st);           // iota = final order
        if (rc) return rc;
        if (hipMemcpyAsync(perm, iota, (size_t)E * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return (int)hipGetLastError();
    } else {
        hipLaunchKernelGGL(coo_keys_kernel, dim3(nb), dim3(T), 0, st, edge_index, E, keys, iota);
        rc = sort_by_key(keys, keys_sorted, iota, perm, E, N, tmp, ws_bytes - 3 * arr, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3(nb), dim3(T), 0, st, keys_sorted, E, N, rowptr);
    hipLaunchKernelGGL(csr_gather_kernel, dim3(nb), dim3(T), 0, st, edge_index, edge_type, perm, E, col, etype);
    return (int)hipGetLastError();
}

int jmac_group_build(const int32_t* keys, int64_t E, int64_t S, int32_t* ptr, int32_t* order, void* ws,
                     size_t ws_bytes, jmac_stream_t stream) {
    if (E < 0 || S < 0 || !ptr || (E > 0 && (!keys || !order))) return JMAC_EINVAL;
    if (E >= INT32_MAX || S >= INT32_MAX) return JMAC_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    const int T = 256;
    if (E == 0) {
        hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3((unsigned)((S + 1 + T - 1) / T)), dim3(T), 0, st, nullptr, 0, S, ptr);
        return (int)hipGetLastError();
    }
    size_t arr = align_up((size_t)E * 4);
    if (!ws || ws_bytes < 2 * arr) return JMAC_EWORKSPACE;
    char* w = (char*)ws;
    int32_t* keys_sorted = (int32_t*)w;
    int32_t* iota = (int32_t*)(w + arr);
    void* tmp = w + 2 * arr;
    unsigned nb = (unsigned)((E + T - 1) / T);
    hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(T), 0, st, E, iota);
    int rc = sort_by_key(const_cast<int32_t*>(keys), keys_sorted, iota, order, E, S, tmp, ws_bytes - 2 * arr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ptr_from_sorted_kernel, dim3(nb), dim3(T), 0, st, keys_sorted, E, S, ptr);
    return (int)hipGetLastError();
}

// coop_min > 0: cooperative splits are enabled for segments longer than coop_min (each adds kCoop - 1 items, one split
// entry and kCoop partial slots; there are at most E / (coop_min + 1) of them)
int64_t jmac_items_max(int64_t S, int64_t E, int32_t chunk, int32_t coop_min) {
    if (chunk < 1) chunk = 1;
    return S + E / chunk + 1 + (coop_min > 0 ? (kCoop - 1) * (E / (coop_min + 1)) : 0);
}
int64_t jmac_splits_max(int64_t E, int32_t chunk, int32_t coop_min) {
    if (chunk < 1) chunk = 1;
    return E / chunk + 1 + (coop_min > 0 ? E / (coop_min + 1) : 0);
}
int64_t jmac_parts_max(int64_t E, int32_t chunk, int32_t coop_min) {
    if (chunk < 1) chunk = 1;
    return 2 * (E / chunk) + 2 + (coop_min > 0 ? kCoop * (E / (coop_min + 1)) : 0);
}

int jmac_items_build(const int32_t* ptr, int64_t S, int32_t chunk, int32_t coop_min, int32_t coop_max, jmac_item_t* items,
                     jmac_split_t* splits, int32_t* counts, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    if (S < 0 || chunk < 1 || coop_min < 0 || coop_max < 0 || !counts || (S > 0 && (!ptr || !items || !splits))) return JMAC_EINVAL;
    if (coop_max > 0 && (coop_min < 1 || coop_max <= coop_min)) return JMAC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (S == 0) {
        hipLaunchKernelGGL(zero_counts_kernel, dim3(1), dim3(64), 0, st, counts);
        return (int)hipGetLastError();
    }
    size_t arr = align_up((size_t)S * 4);
    if (!ws || ws_bytes < 8 * arr) return JMAC_EWORKSPACE;
    char* w = (char*)ws;
    int32_t* flag[4];
    int32_t* off[4];
    for (int i = 0; i < 4; ++i) {
        flag[i] = (int32_t*)(w + i * arr);           // empty, coop, parts (non-coop), split (non-coop)
        off[i] = (int32_t*)(w + (4 + i) * arr);
    }
    void* tmp = w + 8 * arr;
    size_t tmp_bytes = ws_bytes - 8 * arr;
    const int T = 256;
    unsigned nb = (unsigned)((S + T - 1) / T);
    hipLaunchKernelGGL(seg_counts_kernel, dim3(nb), dim3(T), 0, st, ptr, S, chunk, coop_min, coop_max, flag[0], flag[1], flag[2],
                       flag[3]);
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, flag[0], off[0], 0, (size_t)S, rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) return (int)e;
    if (need > tmp_bytes) return JMAC_EWORKSPACE;
    for (int i = 0; i < 4; ++i) {
        e = rocprim::exclusive_scan(tmp, need, flag[i], off[i], 0, (size_t)S, rocprim::plus<int32_t>(), st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(items_fill_kernel, dim3(nb), dim3(T), 0, st, ptr, S, chunk, coop_min, coop_max, off[0], off[1], off[2], off[3],
                       items, splits, counts);
    return (int)hipGetLastError();
}

int jmac_item_edges_build(const jmac_item_t* items, const int32_t* counts, int64_t n_items_max, const int32_t* col,
                          const int32_t* etype, int32_t* item_edges, jmac_stream_t stream) {
    if (n_items_max < 0 || !counts || (n_items_max > 0 && (!items || !col || !etype || !item_edges))) return JMAC_EINVAL;
    if (n_items_max == 0) return JMAC_OK;
    const int T = 256;
    int64_t nb = (n_items_max + T - 1) / T;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(item_edges_kernel, dim3((unsigned)nb), dim3(T), 0, (hipStream_t)stream, items, counts, col, etype,
                       reinterpret_cast<int4*>(item_edges));
    return (int)hipGetLastError();
}

}  // extern "C"
