// aggregate.hip -- relation-aware attention aggregation for gfx950 (forward + backward).
//
// Replaces, in the factorised form of DESIGN.md, the reference's per-edge message / attention /
// scatter_softmax / scatter-sum chain (modules/helper/message_passing.py:4-29,55-90;
// src/jmac_model.py:56-89), the self-loop propagate (src/jmac_model.py:44-45,50) and the (nb+self)/2
// of src/jmac_model.py:52.
//
// Execution model (CDNA4): one 64-lane wavefront per schedule item (a run of <= chunk edges of one
// destination).  A node's [Q|Z] row (2d floats) is covered by NCH float4 "chunks" per lane, chunk
// c = lane + 64k: chunks c < d/4 are the attention half (h-role), the rest the message half (v-role).
// Every gather is a contiguous 8d-byte row read with 16 B per lane; U edges are kept in flight per
// wave; logits are reduced across the wave with DPP + permlane swaps; the softmax is online (running
// max / denominator), so each edge row is read exactly once.  The relation table [Rq|Rz] stays
// L2-resident.  HBM-bound by design: see DESIGN.md for the byte model.
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace jmac;

namespace {

constexpr int kBlock = 256;            // 4 waves
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kPersistBlocks = 2048;   // 256 CUs x 8
#ifndef JMAC_EMPTY_PACK
#define JMAC_EMPTY_PACK 4
#endif
constexpr int kEmptyPack = JMAC_EMPTY_PACK;          // empty segments handled by one wave of the forward kernel
// (non-temporal gathers were measured and rejected: 13.4 vs 12.3 ms on config 4, 36 vs 26 us on ja)

// IGroupLP pipeline for the machine scheduler: first the group's n vector-memory reads, then the vector ALU work.
// (A plain sched_barrier does not do it: instruction selection already linearises the arithmetic ahead of it.)
#define JMAC_LOADS_FIRST(n)                                         \
    do {                                                            \
        __builtin_amdgcn_sched_group_barrier(0x020, (n), 0);        \
        __builtin_amdgcn_sched_group_barrier(0x002, 4096, 0);       \
    } while (0)

struct FwdArgs {
    const void *P, *QZ, *RR;     // tables of element type TT (float or bf16_t)
    const float* a_att;
    int64_t ldp, ldqz, ldrr, ldo;
    const int32_t *rowptr, *col, *etype;
    const jmac_item_t* items;
    const jmac_split_t* splits;
    const int32_t* counts;
    const int4* item_edges;   // optional: {col, type} of the first two entries of every item (small graphs)
    int32_t N, D4, loop_rel, n_items_max;
    int32_t n_coop;       // cooperative segments (host copy of counts[4], exact): the launch geometry depends on it
    int64_t self_off;     // row of QZ that holds destination 0 (fused self term; 0 unless destinations are a slice of the sources)
    int32_t zoff;         // element offset of the Z / Rz half inside a [Q|Z] / [Rq|Rz] row: d, or the padded half pitch (bf16)
    float slope, out_scale;
    float *out, *seg_max, *seg_den;
    float *part_acc, *part_ml;
};

// Lane -> chunk map of one [Q|Z] row.  D4T != 0 fixes d/4 at compile time, so that "does chunk k hold any
// attention-half (h) lane / any message-half (v) lane / any lane past the row end" fold to constants and
// the unused half of the per-chunk arithmetic disappears; D4T == 0 is the generic run-time form.
// Every gather is issued at a CLAMPED offset (lanes past the row end re-read chunk 0) so that no load sits
// behind an exec-mask branch: hipcc puts an s_waitcnt inside such branches, which serialises the gathers.
template <int NCH, int D4T>
struct Lanes {
    int D4r;
    int coff[NCH];     // float offset of this lane's chunk inside a [Q|Z] row (unclamped)
    int coffc[NCH];    // clamped: 0 for lanes past the row end
    bool valid[NCH];
    bool is_h[NCH];
    __device__ __forceinline__ int D4() const { return D4T ? D4T : D4r; }
    __device__ __forceinline__ bool any_h(int k) const { return 64 * k < D4(); }
    __device__ __forceinline__ bool any_v(int k) const { return 64 * (k + 1) > D4(); }
    __device__ __forceinline__ bool all_valid(int k) const { return 64 * (k + 1) <= 2 * D4(); }
    __device__ __forceinline__ bool is_v(int k) const { return valid[k] && !is_h[k]; }
    // zpad: elements between the end of the Q half and the start of the Z half of a gathered row (padded bf16 tables: the
    // halves sit at a 16-byte aligned pitch dh >= d; 0 for the plain [Q|Z] layout)
    __device__ __forceinline__ void init(int lane, int d4_runtime, int zpad = 0) {
        D4r = d4_runtime;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = lane + 64 * k;
            valid[k] = c < 2 * D4();
            is_h[k] = c < D4();
            coff[k] = c * 4;
            coffc[k] = valid[k] ? c * 4 + (is_h[k] ? 0 : zpad) : 0;
        }
    }
};

__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 c) {
    return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
// MAXFORM (specialised kernels; the launcher guarantees 0 <= slope <= 1): max(x, slope*x)
template <bool MAXFORM>
__device__ __forceinline__ float4 leaky4(float4 h, float slope) {
    if (MAXFORM) return make_float4(leaky01(h.x, slope), leaky01(h.y, slope), leaky01(h.z, slope), leaky01(h.w, slope));
    return make_float4(leaky(h.x, slope), leaky(h.y, slope), leaky(h.z, slope), leaky(h.w, slope));
}
__device__ __forceinline__ float4 sel4(bool c, float4 a) { return c ? a : f4zero(); }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
#ifndef JMAC_FWD_WPE
#define JMAC_FWD_WPE 0
#endif
#if JMAC_FWD_WPE
#define JMAC_FWD_ATTR __attribute__((amdgpu_waves_per_eu(JMAC_FWD_WPE, JMAC_FWD_WPE)))
#else
#define JMAC_FWD_ATTR
#endif
// bid / nblk: the block's index and the block count of ITS job (one launch may carry two independent jobs: below)
template <int NCH, int U, int D4T, typename TT>
__device__ __forceinline__ void rel_attn_fwd_body(const FwdArgs& a, const int bid, const int nblk) {
    typedef typename RawOf<TT>::type raw_t;
    // cooperative segments: the partial softmax states of the block's four waves meet here
    __shared__ float4 coop_acc[kWavesPerBlock][NCH][64];
    __shared__ float coop_ml[kWavesPerBlock][2];
    const TT* const tP = static_cast<const TT*>(a.P);
    const TT* const tQZ = static_cast<const TT*>(a.QZ);
    const TT* const tRR = static_cast<const TT*>(a.RR);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = nblk * kWavesPerBlock;
    const int n_items = a.counts[0];
    const int n_empty = a.counts[3];
    const int n_coop = a.n_coop;                  // == counts[4]; from the host so that no address below waits for it
    const int n_reg = n_items - n_empty;          // items with entries come first, the empty segments are the tail of the list
    Lanes<NCH, D4T> L;
    L.init(lane, a.D4, a.zoff - 4 * a.D4);
    const int voff = 4 * L.D4();
    float4 av[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) av[k] = L.any_h(k) ? sel4(L.is_h[k], ld4(a.a_att + (L.is_h[k] ? L.coff[k] : 0))) : f4zero();
    const bool has_loop = a.loop_rel >= 0;
    const TT* rloop = tRR + (int64_t)(has_loop ? a.loop_rel : 0) * a.ldrr;

    // Work units of a launch, in this order for every block / wave:
    //   1. cooperative segments (block cb = blockIdx, blockIdx + grid, ...): four waves, one quarter of the entries each
    //   2. items (wave-strided over the item list behind the cooperative items)
    //   3. packs of kEmptyPack empty segments
    // Items are software-pipelined: while item k computes, the header of item k+2 and the first col/type
    // batch of item k+1 are already in flight, so a wave's critical path per item is one gather round trip
    // instead of header -> col/type -> gather (measured: 65k rows of degree 1 took 114 us serialised).
    // The first header is fetched at a clamped index so that it does not wait for the device-side counts.
    const int it0 = bid * kWavesPerBlock + wave;
    // the loop relation's Rz row is the same for every destination: one read per wave
    // (kept RAW like the Z[i] chunks below: half the registers for bf16 tables, converted where they are used)
    raw_t rl[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
        if (L.any_v(k)) rl[k] = ldraw(rloop + L.coffc[k]);

    // ---- the edges [item.beg, item.end) of destination item.seg: online softmax over them into (m, l, acc) ----------
    // my_col / my_typ: source and type of entry item.beg + lane (first batch of up to 64), supplied by the caller
    auto edge_loop = [&](const jmac_item_t& item, int my_col, int my_typ, const float4 (&pv)[NCH], float& m, float& l,
                         float4 (&acc)[NCH]) {
        for (int e0 = item.beg; e0 < item.end; e0 += 64) {
            const int nb = min(64, item.end - e0);
            if (e0 != item.beg) {
                const int le = min(lane, nb - 1);
                my_col = a.col[e0 + le];
                my_typ = a.etype[e0 + le];
            }
            // group body for UU edges that are all valid; dispatched on the edges left (4 / 2 / 1) so that a
            // degree-1 destination does not pay the instruction count of a full 4-edge group
            auto group = [&](auto uu_c, const int u0) {
                constexpr int UU = decltype(uu_c)::value;
                raw_t qraw[UU][NCH], rraw[UU][NCH];
                float4 q[UU][NCH];
                // 1) issue every gather of the group before any arithmetic
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const int j = bcast_i(my_col, u0 + u);
                    const int t = bcast_i(my_typ, u0 + u);
                    const TT* qrow = tQZ + (int64_t)j * a.ldqz;
                    const TT* rrow = tRR + (int64_t)t * a.ldrr;
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        qraw[u][k] = ldraw(qrow + L.coffc[k]);
                        rraw[u][k] = ldraw(rrow + L.coffc[k]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // measured: 11.6 ms vs 13.3 ms with the IGroupLP form (config 4)
                // 2) logits
                float s[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    float part = 0.f;
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        q[u][k] = sub4(cvt4(qraw[u][k]), cvt4(rraw[u][k]));
                        if (!L.all_valid(k)) q[u][k] = sel4(L.valid[k], q[u][k]);
                        if (L.any_h(k)) part += dot4(av[k], leaky4<D4T != 0>(add4(pv[k], q[u][k]), a.slope));
                    }
                    s[u] = part;
                }
                wave_sum_n<UU>(s);
                // 3) online softmax update
                float gmax = s[0];
#pragma unroll
                for (int u = 1; u < UU; ++u) gmax = fmaxf(gmax, s[u]);
                const float mn = fmaxf(m, gmax);
                const float sc = fast_exp(m - mn);
                float w[UU], wsum = 0.f;
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    w[u] = fast_exp(s[u] - mn);
                    wsum += w[u];
                }
                l = l * sc + wsum;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    if (!L.any_v(k)) continue;
                    float4 x = mul4(acc[k], sc);
#pragma unroll
                    for (int u = 0; u < UU; ++u) x = fma4(q[u][k], w[u], x);
                    acc[k] = x;
                }
                m = mn;
            };
            if constexpr (U >= 4) {
                // Large graphs (HBM-bound): groups of two edges, SOFTWARE-PIPELINED -- the gathers of group g+1 are in
                // flight while group g is reduced, so a wave always has rows outstanding (with "load 4, wait, compute 4"
                // a wave's memory queue is empty for the whole compute phase).  Two register buffers, roles fixed per
                // unrolled half; an odd tail edge is issued twice (clamped) and weighted zero.
                // Group size: two fp32 edges or FOUR bf16 edges -- a bf16 row is half the bytes, so pairs would leave half as
                // many bytes in flight per wave (measured with pairs: 0.45 of the HBM peak on bf16 tables against 0.59 on fp32).
                constexpr int GS = sizeof(TT) == 2 ? 4 : 2;
                raw_t qA[GS][NCH], rA[GS][NCH], qB[GS][NCH], rB[GS][NCH];
                auto issue = [&](const int g, raw_t (&qr)[GS][NCH], raw_t (&rr)[GS][NCH]) {
#pragma unroll
                    for (int u = 0; u < GS; ++u) {
                        const int le = min(GS * g + u, nb - 1);
                        const int j = bcast_i(my_col, le);
                        const int t = bcast_i(my_typ, le);
                        const TT* qrow = tQZ + (int64_t)j * a.ldqz;
                        const TT* rrow = tRR + (int64_t)t * a.ldrr;
#pragma unroll
                        for (int k = 0; k < NCH; ++k) {
                            qr[u][k] = ldraw(qrow + L.coffc[k]);
                            rr[u][k] = ldraw(rrow + L.coffc[k]);
                        }
                    }
                };
                auto consume = [&](const int g, const raw_t (&qr)[GS][NCH], const raw_t (&rr)[GS][NCH]) {
                    float4 q[GS][NCH];
                    float sv[GS];
#pragma unroll
                    for (int u = 0; u < GS; ++u) {
                        float part = 0.f;
#pragma unroll
                        for (int k = 0; k < NCH; ++k) {
                            q[u][k] = sub4(cvt4(qr[u][k]), cvt4(rr[u][k]));
                            if (!L.all_valid(k)) q[u][k] = sel4(L.valid[k], q[u][k]);
                            if (L.any_h(k)) part += dot4(av[k], leaky4<D4T != 0>(add4(pv[k], q[u][k]), a.slope));
                        }
                        sv[u] = part;
                    }
                    wave_sum_n<GS>(sv);
                    float mn = m;
#pragma unroll
                    for (int u = 0; u < GS; ++u) {
                        if (u > 0 && GS * g + u >= nb) sv[u] = -INFINITY;   // tail: the clamped duplicates get weight exp(-inf) = 0
                        mn = fmaxf(mn, sv[u]);
                    }
                    const float sc = fast_exp(m - mn);
                    float w[GS], wsum = 0.f;
#pragma unroll
                    for (int u = 0; u < GS; ++u) {
                        w[u] = fast_exp(sv[u] - mn);
                        wsum += w[u];
                    }
                    l = l * sc + wsum;
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        if (!L.any_v(k)) continue;
                        float4 x = mul4(acc[k], sc);
#pragma unroll
                        for (int u = 0; u < GS; ++u) x = fma4(q[u][k], w[u], x);
                        acc[k] = x;
                    }
                    m = mn;
                };
                const int ngr = (nb + GS - 1) / GS;
                issue(0, qA, rA);
                for (int g = 0; g < ngr; g += 2) {
                    if (g + 1 < ngr) issue(g + 1, qB, rB);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(g, qA, rA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 1 < ngr) {
                        if (g + 2 < ngr) issue(g + 2, qA, rA);
                        __builtin_amdgcn_sched_barrier(0);
                        consume(g + 1, qB, rB);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                int u0 = 0;
                for (; u0 + 2 <= nb; u0 += 2) group(std::integral_constant<int, 2>{}, u0);
                if (u0 < nb) group(std::integral_constant<int, 1>{}, u0);
            }
        }
    };
    // ---- out[i] = out_scale * ( sqrt(deg) / l * acc  +  Z[i] - Rz[loop] ),  softmax statistics for the backward ---------
    auto finish = [&](int i, int deg, float m, float l, const float4 (&acc)[NCH], const raw_t (&zs)[NCH]) {
        const float scale = l > 0.f ? sqrtf((float)deg) / l : 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (L.any_v(k) && L.is_v(k)) {
                const float4 self = has_loop ? sub4(cvt4(zs[k]), cvt4(rl[k])) : f4zero();
                const float4 o = add4(mul4(acc[k], scale), self);                 // + Z[i] - Rz[loop]
                st4(a.out + (int64_t)i * a.ldo + (L.coff[k] - voff), mul4(o, a.out_scale));
            }
        }
        if (lane == 0) {
            a.seg_max[i] = m;
            a.seg_den[i] = l;
        }
    };
    auto first_batch = [&](const jmac_item_t& item, int& c, int& t) {
        const int cnb = min(64, item.end - item.beg);
        const int idx = cnb > 0 ? item.beg + min(lane, cnb - 1) : 0;
        c = a.col[idx];
        t = a.etype[idx];
    };

    // ---- 1. cooperative segments: one workgroup per segment -----------------------------------------------------------
    for (int cb = bid; cb < n_coop; cb += nblk) {
        const jmac_item_t item = a.items[cb * kWavesPerBlock + wave];
        const int i = item.seg;
        int ccol, ctyp;
        first_batch(item, ccol, ctyp);
        float4 pv[NCH], acc[NCH];
        raw_t zs[NCH];
        const TT* prow = tP + (int64_t)i * a.ldp;
        const TT* zrow = tQZ + ((int64_t)(has_loop ? i : 0) + a.self_off) * a.ldqz;   // no loop: row 0 (valid, unused)
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            pv[k] = L.any_h(k) ? cvt4(ldraw(prow + (L.is_h[k] ? L.coff[k] : 0))) : f4zero();
            if (L.any_v(k)) zs[k] = ldraw(zrow + L.coffc[k]);
            acc[k] = f4zero();
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (L.any_h(k) && L.any_v(k)) pv[k] = sel4(L.is_h[k], pv[k]);
        float m = -INFINITY, l = 0.f;
        edge_loop(item, ccol, ctyp, pv, m, l, acc);
        if (wave != 0) {
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (L.any_v(k)) coop_acc[wave][k][lane] = acc[k];
            if (lane == 0) {
                coop_ml[wave][0] = m;
                coop_ml[wave][1] = l;
            }
        }
        __syncthreads();
        if (wave == 0) {
            // merge in wave order (fixed summation order): M = max m_w, state_w scaled by e^(m_w - M)
            float M = m;
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) M = fmaxf(M, coop_ml[w][0]);
            float f = fast_exp(m - M);               // a wave without entries: m = -inf, l = 0 -> factor 0
            float lsum = l * f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) acc[k] = mul4(acc[k], f);
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) {
                f = fast_exp(coop_ml[w][0] - M);
                lsum = fmaf(coop_ml[w][1], f, lsum);
#pragma unroll
                for (int k = 0; k < NCH; ++k)
                    if (L.any_v(k)) acc[k] = fma4(coop_acc[w][k][lane], f, acc[k]);
            }
            finish(i, a.rowptr[i + 1] - a.rowptr[i], M, lsum, acc, zs);
        }
        __syncthreads();
    }

    // ---- 3. (defined here, used below) empty segments, kEmptyPack to a wave: out = out_scale * (Z[i] - Rz[loop]) -------
    auto empties = [&](int p0) {
        const int n_packs = (n_empty + kEmptyPack - 1) / kEmptyPack;
        for (int p = p0; p < n_packs; p += nwaves) {
            int seg[kEmptyPack];
            raw_t zs[kEmptyPack][NCH];
#pragma unroll
            for (int u = 0; u < kEmptyPack; ++u) seg[u] = a.items[min(n_reg + p * kEmptyPack + u, n_items - 1)].seg;
#pragma unroll
            for (int u = 0; u < kEmptyPack; ++u) {
                const TT* zrow = tQZ + ((int64_t)(has_loop ? seg[u] : 0) + a.self_off) * a.ldqz;
#pragma unroll
                for (int k = 0; k < NCH; ++k)
                    if (L.any_v(k)) zs[u][k] = ldraw(zrow + L.coffc[k]);
            }
#pragma unroll
            for (int u = 0; u < kEmptyPack; ++u) {
                if (p * kEmptyPack + u >= n_empty) break;
                const int i = seg[u];
#pragma unroll
                for (int k = 0; k < NCH; ++k)
                    if (L.any_v(k) && L.is_v(k))
                        st4(a.out + (int64_t)i * a.ldo + (L.coff[k] - voff),
                            mul4(has_loop ? sub4(cvt4(zs[u][k]), cvt4(rl[k])) : f4zero(), a.out_scale));
                if (lane == 0) {
                    a.seg_max[i] = -INFINITY;
                    a.seg_den[i] = 0.f;
                }
            }
        }
    };

    // ---- 2. items ---------------------------------------------------------------------------------------------------------
    const int it_base = n_coop * kWavesPerBlock;           // the cooperative items occupy the head of the item list
    int it = it_base + it0;
    jmac_item_t item = a.items[min(it, a.n_items_max - 1)];     // clamped: does not wait for the device-side counts
    // the first two entries of the item arrive WITH its header (item_edges): a short item starts its row gathers one
    // dependent round trip earlier
    int4 e4 = make_int4(0, 0, 0, 0);
    if (a.item_edges) e4 = a.item_edges[min(it, a.n_items_max - 1)];
    if (it >= n_reg) {                                     // this wave owns no item with entries
        empties(it - n_reg);
        return;
    }
    jmac_item_t nitem = a.items[min(it + nwaves, n_items - 1)];
    int ccol, ctyp;
    {
        const int cnb = min(64, item.end - item.beg);
        if (a.item_edges && cnb <= 2) {
            ccol = lane == 0 ? e4.x : e4.z;
            ctyp = lane == 0 ? e4.y : e4.w;
            ccol = cnb > lane ? ccol : e4.x;             // lanes past the item re-use entry 0 (clamped, like the loads)
            ctyp = cnb > lane ? ctyp : e4.y;
        } else {
            first_batch(item, ccol, ctyp);
        }
    }
    for (;;) {
        const jmac_item_t nnitem = a.items[min(it + 2 * nwaves, n_items - 1)];
        int ncol, ntyp;
        first_batch(nitem, ncol, ntyp);
        const int i = item.seg;
        float4 pv[NCH], acc[NCH];
        raw_t zs[NCH];
        const TT* prow = tP + (int64_t)i * a.ldp;
        const TT* zrow = tQZ + ((int64_t)(has_loop ? i : 0) + a.self_off) * a.ldqz;   // no loop: row 0 (valid, unused)
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            pv[k] = L.any_h(k) ? cvt4(ldraw(prow + (L.is_h[k] ? L.coff[k] : 0))) : f4zero();
            // self-loop term Z[i] (v-role lanes), fetched with the header-dependent loads rather than after the edges
            if (L.any_v(k)) zs[k] = ldraw(zrow + L.coffc[k]);
            acc[k] = f4zero();
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (L.any_h(k) && L.any_v(k)) pv[k] = sel4(L.is_h[k], pv[k]);
        float m = -INFINITY, l = 0.f;
        edge_loop(item, ccol, ctyp, pv, m, l, acc);
        if (item.pslot < 0) {
            finish(i, item.end - item.beg, m, l, acc, zs);
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (L.any_v(k) && L.is_v(k)) st4(a.part_acc + (int64_t)item.pslot * voff + (L.coff[k] - voff), acc[k]);
            if (lane == 0) {
                a.part_ml[2 * item.pslot] = m;
                a.part_ml[2 * item.pslot + 1] = l;
            }
        }
        it += nwaves;
        if (it >= n_reg) break;
        item = nitem;
        nitem = nnitem;
        ccol = ncol;
        ctyp = ntyp;
    }
    empties(it - n_reg);                    // persistent grids: the waves share the empty segments after their items
}

template <int NCH, int U, int D4T, typename TT>
__global__ __launch_bounds__(kBlock) JMAC_FWD_ATTR void rel_attn_fwd_kernel(FwdArgs a) {
    rel_attn_fwd_body<NCH, U, D4T, TT>(a, (int)blockIdx.x, (int)gridDim.x);
}

// TWO independent aggregation jobs in one launch (blocks [0, g0): job 0, the rest: job 1): conv1_alignment and conv1_completion of
// JMAC.forward_name (src/jmac_model.py:183,190) read different tables, weights and loop rows but share the graph and do not depend
// on each other -- at DBP-5L size each launch is a chain of dependent round trips with most of the chip idle, so two jobs side by
// side cost little more than one.  Same body, same per-row arithmetic and order: results are those of two launches, bit for bit.
// The body is instantiated once per job, each on ITS by-value argument struct: a struct picked at run time (a reference, or a
// copy fetched from the kernarg segment at a run-time offset) loses the "kernel arguments point to global memory" inference --
// 90 flat_loads instead of global_loads in the ISA, each counted on two wait counters.
template <int NCH, int U, int D4T, typename TT>
__global__ __launch_bounds__(kBlock) JMAC_FWD_ATTR void rel_attn_fwd_jobs_kernel(FwdArgs a0, FwdArgs a1, int g0) {
    if ((int)blockIdx.x < g0)                                         // block-uniform
        rel_attn_fwd_body<NCH, U, D4T, TT>(a0, (int)blockIdx.x, g0);
    else
        rel_attn_fwd_body<NCH, U, D4T, TT>(a1, (int)blockIdx.x - g0, (int)gridDim.x - g0);
}

// ------------------------------------------------------------------------------------------------
// forward, HALF-WAVE lane map (d = 256 / 300): 32 lanes per edge, two edges per wave instruction
// ------------------------------------------------------------------------------------------------
// A [Q|Z] row is cut into 16-byte chunks (4 fp32 / 8 bf16 elements); chunk c = (lane & 31) + 32 k of the row belongs to
// lane (lane & 31) of BOTH halves of the wave, and the two halves gather two DIFFERENT edges of the same destination.
//   fp32, d = 300: 150 chunks on 5 x 32 slots (94 % of the lane slots carry data; the 64-lane map: 150 of 192 = 78 %)
//   bf16, d = 300 (halves padded to 304 elements = 608 B, 16-byte aligned): 76 chunks on 3 x 32 slots with 16-byte lane loads
//         (the 64-lane map loads 8 B per lane: 0.54-0.70 x the 16-byte rate, MI355X_MICROARCH.md)
// Per edge PAIR: one set of gather instructions, ONE cross-lane reduction of the logits (five DPP steps inside the 32-lane
// halves, lanes 31 / 63 read out), one online-softmax update with a per-half weight; the two halves' accumulators meet once
// per item (v_permlane32_swap of two chunk registers: the lower half ends with one complete chunk, the upper half with
// another -- no LDS, no shuffle per element).  Schedule, cooperative segments, combine pass and outputs are those of
// rel_attn_fwd_kernel.  DC = chunks per half row (compile time: 75 / 64 fp32, 38 / 32 bf16).
template <typename TT> struct HwElem;
template <> struct HwElem<float> { static constexpr int CH = 4; };
template <> struct HwElem<bf16_t> { static constexpr int CH = 8; };
template <int CH> struct VF { float v[CH]; };
__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
typedef unsigned jmac_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16_nt(const void* p) {   // streaming: the line is first in line for eviction from L2
    const jmac_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const jmac_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <typename TT> __device__ __forceinline__ VF<HwElem<TT>::CH> hw_cvt(uint4 r);
template <> __device__ __forceinline__ VF<4> hw_cvt<float>(uint4 r) {
    VF<4> o;
    o.v[0] = __uint_as_float(r.x); o.v[1] = __uint_as_float(r.y); o.v[2] = __uint_as_float(r.z); o.v[3] = __uint_as_float(r.w);
    return o;
}
template <> __device__ __forceinline__ VF<8> hw_cvt<bf16_t>(uint4 r) {   // bf16 -> fp32 is a 16-bit shift: exact
    VF<8> o;
    o.v[0] = __uint_as_float(r.x << 16); o.v[1] = __uint_as_float(r.x & 0xffff0000u);
    o.v[2] = __uint_as_float(r.y << 16); o.v[3] = __uint_as_float(r.y & 0xffff0000u);
    o.v[4] = __uint_as_float(r.z << 16); o.v[5] = __uint_as_float(r.z & 0xffff0000u);
    o.v[6] = __uint_as_float(r.w << 16); o.v[7] = __uint_as_float(r.w & 0xffff0000u);
    return o;
}
// sums inside the two 32-lane halves: sa = lanes 0..31, sb = lanes 32..63 (wave-uniform results), U chains interleaved
template <int U>
__device__ __forceinline__ void half_sum_n(float (&v)[U], float (&sa)[U], float (&sb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0xB1, 0xF);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0x4E, 0xF);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0x141, 0xF);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0x140, 0xF);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] += JMAC_DPP(v[u], 0.f, 0x142, 0xA);   // row_bcast:15: lanes 31 / 63 hold their half's sum
#pragma unroll
    for (int u = 0; u < U; ++u) {
        sa[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[u]), 31));
        sb[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[u]), 63));
    }
}
// x: lower half keeps its own lanes and receives the UPPER half of ... see v_permlane32_swap: after the swap
//   a' = { a[0..31], b[0..31] },  b' = { a[32..63], b[32..63] }   =>   a' + b' = { sum of a's halves | sum of b's halves }
__device__ __forceinline__ float halves_meet(float a, float b) {
    // inline asm, not __builtin_amdgcn_permlane32_swap: hipcc 7.2 folds r[0] + r[1] of the builtin's result pair into
    // r[0] + r[0] (seen in the ISA: v_permlane32_swap v0, v52 ; v_pk_add_f32 v[4:5], v[0:1], v[0:1]) -- the same family
    // of mis-folds common.h notes for permlane*_swap(x, x).  s_nop 1 = the two wait states between a VALU write of an
    // operand and the swap (LLVM gfx950 hazard rule; nothing pads the inside of an asm statement).
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
    return a + b;
}

// HOT > 0 (experiment, JMAC_FWD_HOT=1: north_star's "LDS-staged relation tiles"): rows 0 .. HOT-1 of [Rq|Rz] -- the hottest
// relations when the ids are ordered by frequency -- are staged in LDS once per workgroup; an edge of such a relation reads its
// relation chunks from LDS (its global load is pointed at row 0, an L1 hit), the Zipf tail keeps its L2 / Infinity-Cache path.
// PIPE: 0 = load / reduce in turn (one wave per item); 12 = two groups' gathers in flight, uniform issue (the persistent form);
// 2 = two groups, conditional issue (A/B knob).  One pair of edges per group (GP = 1: two pairs measured 8.94 against 6.66 ms)
template <int DC, int PIPE, typename TT, int HOT = 0, int NT = 0>
__device__ __forceinline__ void rel_attn_fwd_hw_body(const FwdArgs& a) {
    constexpr int GP = 1;
    constexpr int CH = HwElem<TT>::CH;
    constexpr int NCH = (2 * DC + 31) / 32;            // chunk slots per lane
    constexpr int KH = (DC + 31) / 32;                 // slots k < KH hold attention-half (h) chunks on some lane
    constexpr int KV0 = DC / 32;                       // slots k >= KV0 hold message-half (v) chunks on some lane
    constexpr int NV = NCH - KV0;
    constexpr int NP = (NV + 1) / 2;                   // v slots in pairs: one complete chunk per half after halves_meet
    typedef VF<CH> vf;
    __shared__ float coop_acc[kWavesPerBlock][NP][CH][64];
    __shared__ float coop_ml[kWavesPerBlock][2];
    __shared__ uint4 hot_rows[HOT > 0 ? HOT * 2 * DC : 1];
    const TT* const tP = static_cast<const TT*>(a.P);
    const TT* const tQZ = static_cast<const TT*>(a.QZ);
    const TT* const tRR = static_cast<const TT*>(a.RR);
    const int lane = lane_id();
    const int hl = lane & 31;
    const bool upper = lane >= 32;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = gridDim.x * kWavesPerBlock;
    const int n_items = a.counts[0];
    const int n_empty = a.counts[3];
    const int n_coop = a.n_coop;
    const int n_reg = n_items - n_empty;
    const int dtrue = 4 * a.D4;                        // output width (<= DC * CH: the tables' pad columns are zero)
    int coff[NCH];                                     // element offset of the lane's chunk in a [Q|Z] row, clamped to a valid one
    bool valid[NCH], is_h[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = hl + 32 * k;
        valid[k] = c < 2 * DC;
        is_h[k] = c < DC;
        coff[k] = valid[k] ? c * CH : 0;
    }
    auto all_valid = [](int k) { return 32 * (k + 1) <= 2 * DC; };
    vf av[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k)
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int col = (hl + 32 * k) * CH + e;
            av[k].v[e] = (is_h[k] && col < dtrue) ? a.a_att[col] : 0.f;
        }
    const bool has_loop = a.loop_rel >= 0;
    const TT* rloop = tRR + (int64_t)(has_loop ? a.loop_rel : 0) * a.ldrr;
    // chunk of the [Q|Z] row that OUTPUT slot p holds on this lane once the halves have met (the lower half keeps v slot
    // KV0 + 2p, the upper half v slot KV0 + 2p + 1); -1: none (an h chunk, past the row, or an unpaired last slot)
    int oc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int k = KV0 + 2 * p + (upper ? 1 : 0);
        const int c = hl + 32 * k;
        oc[p] = (k < NCH && c >= DC && c < 2 * DC) ? c : -1;
    }
    // Every load stays RAW (uint4) until its first use: a conversion or a select right behind a load is waited for where it
    // stands, i.e. in front of the gathers that should be in flight together with it (one more round trip per item).
    // relation table as a buffer resource (HOT only): 32-bit lane offsets, out-of-range lanes are dropped by the range check
    const auto rr_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<TT*>(tRR), 0, (int)0xFFFFFFF0u, 0x00020000);
    if constexpr (HOT > 0) {                           // stage the hot relation rows (the table has more than HOT rows: launcher)
        for (int i = threadIdx.x; i < HOT * 2 * DC; i += kBlock)
            hot_rows[i] = ld16(tRR + (int64_t)(i / (2 * DC)) * a.ldrr + (i % (2 * DC)) * CH);
        __syncthreads();
    }
    uint4 rlraw[NP];                                   // the loop relation's Rz chunks in the OUTPUT layout: one read per wave
#pragma unroll
    for (int p = 0; p < NP; ++p) rlraw[p] = ld16(rloop + (oc[p] >= 0 ? oc[p] : 0) * CH);

    // ---- online softmax over the entries [item.beg, item.end) of one destination; both halves carry PARTIAL accumulators
    auto edge_loop = [&](const jmac_item_t& item, int my_col, int my_typ, const uint4 (&praw)[KH], float& m, float& l, vf (&acc)[NCH]) {
        auto issue = [&](const int nb, const int mc, const int mt, const int p0, uint4 (&qr)[GP][NCH], uint4 (&rr)[GP][NCH], int (&th)[GP]) {
#pragma unroll
            for (int u = 0; u < GP; ++u) {
                const int ea = max(min(2 * (p0 + u), nb - 1), 0), eb = max(min(2 * (p0 + u) + 1, nb - 1), 0);
                const int ja = bcast_i(mc, ea), jb = bcast_i(mc, eb);
                const int ta = bcast_i(mt, ea), tb = bcast_i(mt, eb);
                const TT* qrow = tQZ + (int64_t)(upper ? jb : ja) * a.ldqz;
                const int tsel = upper ? tb : ta;
                th[u] = -1;
                const TT* rrow = tRR + (int64_t)tsel * a.ldrr;
                if constexpr (HOT > 0) {
                    // hot: the chunks come from LDS and the lane's relation load must cost NOTHING -- pointed at a dummy row (row 0,
                    // or the lane's own [Q|Z] chunk) it still went to L2 and the kernel ran 2.4 x longer.  A buffer load whose
                    // offset lies past the descriptor's range is dropped by the range check: no request leaves the CU.
                    th[u] = tsel < HOT ? tsel : -1;
                    const unsigned rb = tsel < HOT ? 0xFFFFFFFFu : (unsigned)tsel * (unsigned)(a.ldrr * sizeof(TT));
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        qr[u][k] = ld16(qrow + coff[k]);
                        const unsigned vo = tsel < HOT ? 0xFFFFFFFFu : rb + (unsigned)(coff[k] * sizeof(TT));
                        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rr_rsrc, (int)vo, 0, 0);
                        rr[u][k] = make_uint4(v[0], v[1], v[2], v[3]);
                    }
                    continue;
                }
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    qr[u][k] = NT ? ld16_nt(qrow + coff[k]) : ld16(qrow + coff[k]);
                    rr[u][k] = ld16(rrow + coff[k]);
                }
            }
        };
        // the item's FIRST group of gathers goes out here, in one basic block with the loads of P[i] / Z[i] and ahead of the
        // conversion of P[i]: that conversion waits for P[i] only (the gathers are younger), not the other way round
        uint4 qA[GP][NCH], rA[GP][NCH];
        int tA[GP];
        issue(min(64, item.end - item.beg), my_col, my_typ, 0, qA, rA, tA);
        vf pv[KH];
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            pv[k] = hw_cvt<TT>(praw[k]);
            if (32 * (k + 1) > DC)
#pragma unroll
                for (int e = 0; e < CH; ++e) pv[k].v[e] = is_h[k] ? pv[k].v[e] : 0.f;
        }
        for (int e0 = item.beg; e0 < item.end; e0 += 64) {
            const int nb = min(64, item.end - e0);
            if (e0 != item.beg) {
                const int le = min(lane, nb - 1);
                my_col = a.col[e0 + le];
                my_typ = a.etype[e0 + le];
                issue(nb, my_col, my_typ, 0, qA, rA, tA);
            }
            const int npairs = (nb + 1) >> 1;
            auto consume = [&](const int p0, const uint4 (&qr)[GP][NCH], const uint4 (&rr)[GP][NCH], const int (&th)[GP]) {
                vf x[GP][NCH];
                float part[GP];
#pragma unroll
                for (int u = 0; u < GP; ++u) {
                    part[u] = 0.f;
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        uint4 rraw = rr[u][k];
                        if constexpr (HOT > 0) {
                            const uint4 lv = hot_rows[max(th[u], 0) * (2 * DC) + min(hl + 32 * k, 2 * DC - 1)];
                            // component-wise: hipcc lowers a ternary over uint4 STRUCTS to two scratch stores and an indexed
                            // scratch load (48 B of private segment, 2.4 x the kernel's time); on dwords it is v_cndmask
                            const bool hsel = th[u] >= 0;
                            rraw.x = hsel ? lv.x : rraw.x; rraw.y = hsel ? lv.y : rraw.y;
                            rraw.z = hsel ? lv.z : rraw.z; rraw.w = hsel ? lv.w : rraw.w;
                        }
                        const vf q = hw_cvt<TT>(qr[u][k]), r = hw_cvt<TT>(rraw);
#pragma unroll
                        for (int e = 0; e < CH; ++e) {
                            float d = q.v[e] - r.v[e];
                            if (!all_valid(k)) d = valid[k] ? d : 0.f;
                            x[u][k].v[e] = d;
                            if (k < KH) part[u] = fmaf(av[k].v[e], leaky01(pv[k].v[e] + d, a.slope), part[u]);   // av = 0 off the h lanes
                        }
                    }
                }
                float sa[GP], sb[GP];
                half_sum_n<GP>(part, sa, sb);
                float mn = m;
#pragma unroll
                for (int u = 0; u < GP; ++u) {
                    if (2 * (p0 + u) >= nb) sa[u] = -INFINITY;          // pairs / edges past the batch: weight exp(-inf) = 0
                    if (2 * (p0 + u) + 1 >= nb) sb[u] = -INFINITY;
                    mn = fmaxf(mn, fmaxf(sa[u], sb[u]));
                }
                const float sc = fast_exp(m - mn);
                float w[GP], wsum = 0.f;
#pragma unroll
                for (int u = 0; u < GP; ++u) {
                    const float wa = fast_exp(sa[u] - mn), wb = fast_exp(sb[u] - mn);
                    wsum += wa + wb;
                    w[u] = upper ? wb : wa;
                }
                l = l * sc + wsum;
#pragma unroll
                for (int k = KV0; k < NCH; ++k)
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        float t = acc[k].v[e] * sc;
#pragma unroll
                        for (int u = 0; u < GP; ++u) t = fmaf(x[u][k].v[e], w[u], t);
                        acc[k].v[e] = t;
                    }
                m = mn;
            };
            if constexpr (PIPE == 12) {
                // two groups in flight, EVERY path issuing the same loads: a group past the batch's end re-reads the last pair's rows
                // (an L1 hit; it is never reduced).  The compiler's vmcnt bookkeeping is exact only when all paths into a wait have
                // issued the same number of loads: behind a conditional issue (the PIPE 2 form below, and the three-deep form this
                // replaced) it falls back to vmcnt(0) -- the whole pipeline drained once per trip, seen in the ISA; here the waits
                // count down vmcnt(11) .. vmcnt(6) chunk by chunk while the younger group stays in flight
                uint4 qB[GP][NCH], rB[GP][NCH];
                int tB[GP];
                int p = 0;
                for (;;) {
                    issue(nb, my_col, my_typ, p + GP, qB, rB, tB);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(p, qA, rA, tA);
                    __builtin_amdgcn_sched_barrier(0);
                    p += GP;
                    if (p >= npairs) break;
                    issue(nb, my_col, my_typ, p + GP, qA, rA, tA);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(p, qB, rB, tB);
                    __builtin_amdgcn_sched_barrier(0);
                    p += GP;
                    if (p >= npairs) break;
                }
            } else if constexpr (PIPE == 2) {
                uint4 qB[GP][NCH], rB[GP][NCH];
                int tB[GP];
                for (int p = 0; p < npairs; p += 2 * GP) {
                    if (p + GP < npairs) issue(nb, my_col, my_typ, p + GP, qB, rB, tB);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(p, qA, rA, tA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + GP < npairs) {
                        if (p + 2 * GP < npairs) issue(nb, my_col, my_typ, p + 2 * GP, qA, rA, tA);
                        __builtin_amdgcn_sched_barrier(0);
                        consume(p + GP, qB, rB, tB);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                for (int p = 0; p < npairs; p += GP) {
                    if (p > 0) issue(nb, my_col, my_typ, p, qA, rA, tA);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(p, qA, rA, tA);
                }
            }
        }
    };
    // ---- the two halves' partial accumulators -> NP complete chunks per lane (oc[p]): an odd last slot is paired with slot
    // ---- KV0 once more (the upper half's copy of that pair is not used)
    auto meet = [&](const vf (&acc)[NCH], vf (&outv)[NP]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            constexpr int dummy = KV0;
            const int k1 = KV0 + 2 * p, k2 = (KV0 + 2 * p + 1 < NCH) ? KV0 + 2 * p + 1 : dummy;
#pragma unroll
            for (int e = 0; e < CH; ++e) outv[p].v[e] = halves_meet(acc[k1].v[e], acc[k2].v[e]);
        }
    };
    auto store_chunks = [&](float* rowp, const vf (&o)[NP]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (oc[p] < 0) continue;
            const int col = (oc[p] - DC) * CH;
#pragma unroll
            for (int e = 0; e < CH; e += 4)
                if (col + e < dtrue) st4(rowp + col + e, make_float4(o[p].v[e], o[p].v[e + 1], o[p].v[e + 2], o[p].v[e + 3]));
        }
    };
    // raw Z[i] chunks in the OUTPUT layout (the fused self loop); converted in finish
    auto load_z = [&](int i, uint4 (&zraw)[NP]) {
        const TT* zrow = tQZ + ((int64_t)(has_loop ? i : 0) + a.self_off) * a.ldqz;   // no loop: row 0 (valid, unused)
#pragma unroll
        for (int p = 0; p < NP; ++p) zraw[p] = ld16(zrow + (oc[p] >= 0 ? oc[p] : 0) * CH);
    };
    // out[i] = out_scale * (scale * o + Z[i] - Rz[loop])
    auto finish = [&](int i, float scale, const vf (&o)[NP], const uint4 (&zraw)[NP]) {
        vf r[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const vf z = hw_cvt<TT>(zraw[p]), rz = hw_cvt<TT>(rlraw[p]);
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const float self = has_loop ? z.v[e] - rz.v[e] : 0.f;
                r[p].v[e] = fmaf(o[p].v[e], scale, self) * a.out_scale;
            }
        }
        store_chunks(a.out + (int64_t)i * a.ldo, r);
    };
    auto first_batch = [&](const jmac_item_t& item, int& c, int& t) {
        const int cnb = min(64, item.end - item.beg);
        const int idx = cnb > 0 ? item.beg + min(lane, cnb - 1) : 0;
        c = a.col[idx];
        t = a.etype[idx];
    };
    auto load_p = [&](int i, uint4 (&praw)[KH]) {
        const TT* prow = tP + (int64_t)i * a.ldp;
#pragma unroll
        for (int k = 0; k < KH; ++k) praw[k] = ld16(prow + (is_h[k] ? (hl + 32 * k) * CH : 0));
    };
    auto zero_acc = [&](vf (&acc)[NCH]) {
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[k].v[e] = 0.f;
    };

    // ---- 1. cooperative segments: one workgroup per segment ------------------------------------------------------------
    for (int cb = blockIdx.x; cb < n_coop; cb += gridDim.x) {
        const jmac_item_t item = a.items[cb * kWavesPerBlock + wave];
        const int i = item.seg;
        int ccol, ctyp;
        first_batch(item, ccol, ctyp);
        uint4 praw[KH], zraw[NP];
        vf acc[NCH], o[NP];
        load_p(i, praw);
        load_z(i, zraw);                                  // (wave 0 uses it)
        zero_acc(acc);
        float m = -INFINITY, l = 0.f;
        edge_loop(item, ccol, ctyp, praw, m, l, acc);
        meet(acc, o);
        if (wave != 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int e = 0; e < CH; ++e) coop_acc[wave][p][e][lane] = o[p].v[e];
            if (lane == 0) {
                coop_ml[wave][0] = m;
                coop_ml[wave][1] = l;
            }
        }
        __syncthreads();
        if (wave == 0) {
            float M = m;
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) M = fmaxf(M, coop_ml[w][0]);
            float f = fast_exp(m - M);
            float lsum = l * f;
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int e = 0; e < CH; ++e) o[p].v[e] *= f;
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) {
                f = fast_exp(coop_ml[w][0] - M);
                lsum = fmaf(coop_ml[w][1], f, lsum);
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int e = 0; e < CH; ++e) o[p].v[e] = fmaf(coop_acc[w][p][e][lane], f, o[p].v[e]);
            }
            const int deg = a.rowptr[i + 1] - a.rowptr[i];
            finish(i, lsum > 0.f ? sqrtf((float)deg) / lsum : 0.f, o, zraw);
            if (lane == 0) {
                a.seg_max[i] = M;
                a.seg_den[i] = lsum;
            }
        }
        __syncthreads();
    }

    // ---- 3. empty segments, kEmptyPack to a wave: out = out_scale * (Z[i] - Rz[loop]) ----------------------------------------
    auto empties = [&](int p0) {
        const int n_packs = (n_empty + kEmptyPack - 1) / kEmptyPack;
        for (int pk = p0; pk < n_packs; pk += nwaves) {
            int seg[kEmptyPack];
            uint4 zraw[kEmptyPack][NP];
#pragma unroll
            for (int u = 0; u < kEmptyPack; ++u) seg[u] = a.items[min(n_reg + pk * kEmptyPack + u, n_items - 1)].seg;
#pragma unroll
            for (int u = 0; u < kEmptyPack; ++u) load_z(seg[u], zraw[u]);
#pragma unroll
            for (int u = 0; u < kEmptyPack; ++u) {
                if (pk * kEmptyPack + u >= n_empty) break;
                vf zero[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int e = 0; e < CH; ++e) zero[p].v[e] = 0.f;
                finish(seg[u], 0.f, zero, zraw[u]);
                if (lane == 0) {
                    a.seg_max[seg[u]] = -INFINITY;
                    a.seg_den[seg[u]] = 0.f;
                }
            }
        }
    };

    // ---- 2. items ---------------------------------------------------------------------------------------------------------
    const int it_base = n_coop * kWavesPerBlock;
    const int it0 = blockIdx.x * kWavesPerBlock + wave;
    int it = it_base + it0;
    jmac_item_t item = a.items[min(it, a.n_items_max - 1)];
    int4 e4 = make_int4(0, 0, 0, 0);
    if (a.item_edges) e4 = a.item_edges[min(it, a.n_items_max - 1)];
    if (it >= n_reg) {
        empties(it - n_reg);
        return;
    }
    jmac_item_t nitem = a.items[min(it + nwaves, n_items - 1)];
    int ccol, ctyp;
    {
        const int cnb = min(64, item.end - item.beg);
        if (a.item_edges && cnb <= 2) {
            ccol = lane == 0 ? e4.x : e4.z;
            ctyp = lane == 0 ? e4.y : e4.w;
            ccol = cnb > lane ? ccol : e4.x;
            ctyp = cnb > lane ? ctyp : e4.y;
        } else {
            first_batch(item, ccol, ctyp);
        }
    }
    for (;;) {
        const jmac_item_t nnitem = a.items[min(it + 2 * nwaves, n_items - 1)];
        int ncol, ntyp;
        first_batch(nitem, ncol, ntyp);
        const int i = item.seg;
        uint4 praw[KH], zraw[NP];
        vf acc[NCH], o[NP];
        load_p(i, praw);
        load_z(i, zraw);
        zero_acc(acc);
        float m = -INFINITY, l = 0.f;
        edge_loop(item, ccol, ctyp, praw, m, l, acc);
        meet(acc, o);
        if (item.pslot < 0) {
            finish(i, l > 0.f ? sqrtf((float)(item.end - item.beg)) / l : 0.f, o, zraw);
            if (lane == 0) {
                a.seg_max[i] = m;
                a.seg_den[i] = l;
            }
        } else {
            // partial state of a split destination: the combine pass reads d floats per slot in row order
            store_chunks(a.part_acc + (int64_t)item.pslot * dtrue, o);
            if (lane == 0) {
                a.part_ml[2 * item.pslot] = m;
                a.part_ml[2 * item.pslot + 1] = l;
            }
        }
        it += nwaves;
        if (it >= n_reg) break;
        item = nitem;
        nitem = nnitem;
        ccol = ncol;
        ctyp = ntyp;
    }
    empties(it - n_reg);
}

template <int DC, int PIPE, typename TT, int HOT = 0, int NT = 0>
__global__ __launch_bounds__(kBlock) void rel_attn_fwd_hw_kernel(FwdArgs a) {
    rel_attn_fwd_hw_body<DC, PIPE, TT, HOT, NT>(a);
}

// merges the partial (max, denominator, accumulator) triples of destinations that were split: one BLOCK per split
// destination, wave w takes chunks w, w+4, ..., the four wave results are combined through LDS in wave order
template <int NCH, typename TT>
__global__ __launch_bounds__(kBlock) void rel_attn_fwd_combine_kernel(FwdArgs a) {
    __shared__ float4 red[kWavesPerBlock][NCH][64];
    __shared__ float redl[kWavesPerBlock];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_splits = a.counts[1];
    const int n_coop = a.counts[4];          // cooperative segments (head of the splits array) were merged in LDS already
    Lanes<NCH, 0> L;
    L.init(lane, a.D4);
    const int voff = 4 * a.D4;
    for (int sp = n_coop + blockIdx.x; sp < n_splits; sp += gridDim.x) {
        const jmac_split_t s = a.splits[sp];
        const int i = s.seg;
        float M = -INFINITY;
        for (int c = lane; c < s.nchunks; c += 64) M = fmaxf(M, a.part_ml[2 * (s.pslot0 + c)]);
        M = wave_max(M);
        float lsum = 0.f;
        float4 acc[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) acc[k] = f4zero();
        // eight partial states in flight per wave (a hub of config 4 has ~1 000 of them: a chain of round trips otherwise);
        // the tail re-reads the last chunk with factor 0, so that every load is unconditional
        constexpr int CI = 8;
        for (int c = wave; c < s.nchunks; c += CI * kWavesPerBlock) {
            float2 ml[CI];
            float4 row[CI][NCH];
#pragma unroll
            for (int q = 0; q < CI; ++q) {
                const int ps = s.pslot0 + min(c + q * kWavesPerBlock, s.nchunks - 1);
                ml[q] = *reinterpret_cast<const float2*>(a.part_ml + 2 * ps);
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int o = L.is_v(k) ? L.coff[k] - voff : 0;
                    row[q][k] = ld4(a.part_acc + (int64_t)ps * voff + o);
                }
            }
#pragma unroll
            for (int q = 0; q < CI; ++q) {
                const float f = c + q * kWavesPerBlock < s.nchunks ? fast_exp(ml[q].x - M) : 0.f;
                lsum += ml[q].y * f;
#pragma unroll
                for (int k = 0; k < NCH; ++k) acc[k] = fma4(row[q][k], f, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) red[wave][k][lane] = acc[k];
        if (lane == 0) redl[wave] = lsum;
        __syncthreads();
        if (wave == 0) {
            lsum = (redl[0] + redl[1]) + (redl[2] + redl[3]);
            const int deg = a.rowptr[i + 1] - a.rowptr[i];
            const float scale = lsum > 0.f ? sqrtf((float)deg) / lsum : 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (L.is_v(k)) {
                    float4 o = add4(add4(red[0][k][lane], red[1][k][lane]), add4(red[2][k][lane], red[3][k][lane]));
                    o = mul4(o, scale);
                    if (a.loop_rel >= 0) {
                        float4 z = cvt4(ldraw(static_cast<const TT*>(a.QZ) + ((int64_t)i + a.self_off) * a.ldqz + (L.coff[k] - voff + a.zoff)));
                        float4 rz = cvt4(ldraw(static_cast<const TT*>(a.RR) + (int64_t)a.loop_rel * a.ldrr + (L.coff[k] - voff + a.zoff)));
                        o = add4(o, sub4(z, rz));
                    }
                    st4(a.out + (int64_t)i * a.ldo + (L.coff[k] - voff), mul4(o, a.out_scale));
                }
            }
            if (lane == 0) {
                a.seg_max[i] = M;
                a.seg_den[i] = lsum;
            }
        }
        __syncthreads();
    }
}

// ---- merge / reduction bodies: device functions over a (block id, block count) pair so that several of them can share
// ---- one launch (the DBP-5L-scale backward is bound by launches, not by bytes) ------------------------------------------
struct SumPartsTask {           // plain sum of the partial rows of split segments (width = 4*W4 floats)
    const jmac_split_t* splits;
    const int32_t* counts;
    const float* part;
    int W4;
    float sign;
    float* outp;
    int64_t ldout;
    const float* G;             // optional self term (pass B): out[seg][d:] += kappa * G[seg - self_off]
    int64_t ldg;
    int D4;
    float kappa;
    int add_self;
    int64_t self_off, n_self;
    int nblocks;                // blocks of the launch that work on this task (0 = task absent)
};
struct ReduceTask {             // out[c] = scale * sum_p partial[p*W + c]
    const float* partial;
    int nparts, W;
    float scale;
    float* outp;
    int nblocks;
};
struct ColsumTask {             // per-block partial column sums of X [R, 4*W4] -> partial [nblocks, 4*W4]
    const float* X;
    int64_t ldx, R;
    int W4;
    float* partial;
    int nblocks;
};

// one BLOCK per split segment: wave w sums partial rows w, w+4, w+8, ... (two rows in flight per lane), the four
// wave sums are combined through LDS in wave order -> fixed summation order, and the hottest relation's thousands
// of partial rows are no longer one wave's serial chain
__device__ __forceinline__ void sum_parts_body(const SumPartsTask& t, int bid, int nb, float4 (*red)[64]) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_splits = t.counts[1];
    const int W4 = t.W4;
    for (int sp = bid; sp < n_splits; sp += nb) {
        const jmac_split_t s = t.splits[sp];
        for (int c0 = 0; c0 < W4; c0 += 64) {
            const int c4 = c0 + lane;
            // SP_INFLIGHT partial rows in flight per lane: a long list (the hottest relation of a DBP-5L graph: ~170 partial
            // rows) is a chain of dependent round trips, not bandwidth
            constexpr int SP_INFLIGHT = 4;
            float4 acc[SP_INFLIGHT];
#pragma unroll
            for (int q = 0; q < SP_INFLIGHT; ++q) acc[q] = f4zero();
            if (c4 < W4) {
                const float* base = t.part + ((int64_t)s.pslot0 * W4 + c4) * 4;
                const int64_t rs = (int64_t)W4 * 4;
                int c = wave;
                for (; c + (SP_INFLIGHT - 1) * kWavesPerBlock < s.nchunks; c += SP_INFLIGHT * kWavesPerBlock) {
#pragma unroll
                    for (int q = 0; q < SP_INFLIGHT; ++q) acc[q] = add4(acc[q], ld4(base + (int64_t)(c + q * kWavesPerBlock) * rs));
                }
                for (; c < s.nchunks; c += kWavesPerBlock) acc[0] = add4(acc[0], ld4(base + (int64_t)c * rs));
            }
#pragma unroll
            for (int q = 1; q < SP_INFLIGHT; ++q) acc[0] = add4(acc[0], acc[q]);
            red[wave][lane] = acc[0];
            __syncthreads();
            if (wave == 0 && c4 < W4) {
                float4 acc = add4(add4(red[0][lane], red[1][lane]), add4(red[2][lane], red[3][lane]));
                acc = mul4(acc, t.sign);
                if (t.add_self && c4 >= t.D4 && s.seg >= t.self_off && s.seg - t.self_off < t.n_self)
                    acc = fma4(ld4(t.G + ((int64_t)s.seg - t.self_off) * t.ldg + (c4 - t.D4) * 4), t.kappa, acc);
                st4(t.outp + (int64_t)s.seg * t.ldout + c4 * 4, acc);
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kBlock) void sum_parts_kernel(SumPartsTask t) {
    __shared__ float4 red[kWavesPerBlock][64];
    sum_parts_body(t, blockIdx.x, gridDim.x, red);
}

// column sums of a [R, 4*W4] matrix in two deterministic steps: per-block partials (here), then a reduce task.
// rpb = kBlock / W4 rows are in flight per block (thread = (row lane, float4 column)), 2 loads in flight per thread;
// the row lanes are combined through LDS in a fixed order
__device__ __forceinline__ void colsum_partial_body(const ColsumTask& t, int bid, int nb, float4* red) {
    const int tid = threadIdx.x;
    const int W4 = t.W4;
    if (W4 > kBlock) {   // wide rows: one thread per float4 column, looped
        for (int c4 = tid; c4 < W4; c4 += kBlock) {
            float4 acc = f4zero();
            for (int64_t r = bid; r < t.R; r += nb) acc = add4(acc, ld4(t.X + r * t.ldx + c4 * 4));
            st4(t.partial + ((int64_t)bid * W4 + c4) * 4, acc);
        }
        return;
    }
    const int rpb = kBlock / W4;
    const int rl = tid / W4, c4 = tid % W4;
    float4 a0 = f4zero(), a1 = f4zero();
    if (rl < rpb) {
        const int64_t stride = (int64_t)nb * rpb;
        int64_t r = (int64_t)bid * rpb + rl;
        for (; r + stride < t.R; r += 2 * stride) {
            a0 = add4(a0, ld4(t.X + r * t.ldx + c4 * 4));
            a1 = add4(a1, ld4(t.X + (r + stride) * t.ldx + c4 * 4));
        }
        if (r < t.R) a0 = add4(a0, ld4(t.X + r * t.ldx + c4 * 4));
    }
    red[tid] = add4(a0, a1);
    __syncthreads();
    if (tid < W4) {
        float4 acc = red[tid];
        for (int q = 1; q < rpb; ++q) acc = add4(acc, red[q * W4 + tid]);
        st4(t.partial + ((int64_t)bid * W4 + tid) * 4, acc);
    }
}

__global__ __launch_bounds__(kBlock) void colsum_partial_kernel(ColsumTask t) {
    __shared__ float4 red[kBlock];
    colsum_partial_body(t, blockIdx.x, gridDim.x, red);
}

// out[c] = scale * sum_p partial[p][c]: 16 columns x 64 row lanes per 1024-thread block (a block's rows are read as 64-byte
// segments; 16 columns per block instead of 64 puts 19 blocks instead of 5 on a 300-wide reduction: 6.6 -> ~4 us)
constexpr int RR_COLS = 16, RR_LANES = 1024 / RR_COLS;
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ partial, int nparts, int W, float scale,
                                                           float* __restrict__ outp) {
    __shared__ float red[RR_LANES][RR_COLS];
    const int cl = threadIdx.x % RR_COLS, rl = threadIdx.x / RR_COLS;
    const int c = blockIdx.x * RR_COLS + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < W) {
        int p = rl;
        for (; p + 3 * RR_LANES < nparts; p += 4 * RR_LANES) {
            a0 += partial[(int64_t)p * W + c];
            a1 += partial[(int64_t)(p + RR_LANES) * W + c];
            a2 += partial[(int64_t)(p + 2 * RR_LANES) * W + c];
            a3 += partial[(int64_t)(p + 3 * RR_LANES) * W + c];
        }
        for (; p < nparts; p += RR_LANES) a0 += partial[(int64_t)p * W + c];
    }
    red[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && c < W) {
        float s = red[0][cl];
#pragma unroll 8
        for (int r = 1; r < RR_LANES; ++r) s += red[r][cl];
        outp[c] = s * scale;
    }
}

// the same reduction for a 256-thread block that shares a launch with other tasks: 8 columns x 32 row lanes
constexpr int RT_COLS = 8, RT_LANES = kBlock / RT_COLS;
__device__ __forceinline__ void reduce_rows_body(const ReduceTask& t, int bid, float (*red)[RT_COLS]) {
    const int cl = threadIdx.x % RT_COLS, rl = threadIdx.x / RT_COLS;
    const int c = bid * RT_COLS + cl;
    // RT_INFLIGHT partial rows in flight per thread: the a_att reduction of a DBP-5L graph sums ~3 000 partial rows, i.e. a
    // chain of dependent round trips per thread, not bandwidth
    constexpr int RT_INFLIGHT = 12;
    float acc[RT_INFLIGHT];
#pragma unroll
    for (int q = 0; q < RT_INFLIGHT; ++q) acc[q] = 0.f;
    if (c < t.W) {
        int p = rl;
        for (; p + (RT_INFLIGHT - 1) * RT_LANES < t.nparts; p += RT_INFLIGHT * RT_LANES) {
#pragma unroll
            for (int q = 0; q < RT_INFLIGHT; ++q) acc[q] += t.partial[(int64_t)(p + q * RT_LANES) * t.W + c];
        }
        for (; p < t.nparts; p += RT_LANES) acc[0] += t.partial[(int64_t)p * t.W + c];
    }
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < RT_INFLIGHT; ++q) tot += acc[q];
    red[rl][cl] = tot;
    __syncthreads();
    if (rl == 0 && c < t.W) {
        float s = red[0][cl];
#pragma unroll 8
        for (int r = 1; r < RT_LANES; ++r) s += red[r][cl];
        t.outp[c] = s * t.scale;
    }
}

// Backward, last launch: every merge of the deterministic backward in ONE grid -- the partial rows of the split
// destinations / sources / relations (dP, d[Q|Z], d[Rq|Rz]), the a_att gradient (per-block partial rows of pass A) and
// the column sum of G (dRz[loop]).  Blocks are dealt to the tasks in order; every task loops over its own block range.
struct FinalizeArgs {
    SumPartsTask sp[3];
    ReduceTask rd[2];
};
__global__ __launch_bounds__(kBlock) void bwd_finalize_kernel(FinalizeArgs f) {
    __shared__ float4 red4[kWavesPerBlock][64];
    __shared__ float redr[RT_LANES][RT_COLS];
    int b = blockIdx.x;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (b < f.sp[i].nblocks) {
            sum_parts_body(f.sp[i], b, f.sp[i].nblocks, red4);
            return;
        }
        b -= f.sp[i].nblocks;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (b < f.rd[i].nblocks) {
            reduce_rows_body(f.rd[i], b, redr);
            return;
        }
        b -= f.rd[i].nblocks;
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct BwdArgs {
    const float *P, *QZ, *RR, *a_att, *out, *seg_max, *seg_den, *G;
    int64_t ldp, ldqz, ldrr, ldo, ldg, lddp, lddqz, lddrr;
    const int32_t *rowptr, *col, *etype, *dst_of_slot;
    const jmac_item_t* items;
    const jmac_split_t* splits;
    const int32_t* counts;
    const int32_t* order;     // pass B / C: CSR slot of each entry
    int32_t N, D4, loop_rel, nrel;
    int64_t self_off;
    float slope, out_scale;
    float *dP, *dQZ, *dRR;
    float* da_part;           // [gridDim.x, d]
    float* part;              // partial rows of split segments
    float2* wds;              // [E] (w_e, ds_e)
    unsigned char* bits;      // [E, 64] sign bits of h_e: lane's byte = nibble per h-role chunk
    float sign;               // pass B: +1, pass C: -1
    int32_t add_self;         // pass B: add g_j to the Z half
    // small graphs (one wave per item, bound by dependent round trips): the first two entries of every item inline with its
    // header -- pass A: {col, type} pairs, passes B / C: {CSR slot, destination} pairs -- or NULL
    const int4* item_edges;
    const int32_t* entry_dst; // passes B / C: destination of each entry in GROUPED order (= dst_of_slot[order[x]]), or NULL
    int32_t n_items_max;      // bound of items[] (host value): headers are fetched before the device-side count arrives
};

// Pass A: by destination.  MODE 0: float atomics into dQZ / dRR.  MODE 1: per-edge records.
template <int NCH, int U, int MODE, int D4T>
__global__ __launch_bounds__(kBlock) void rel_attn_bwd_dst_kernel(BwdArgs a, int gridA, ColsumTask cs) {
    constexpr int NCH_H = (NCH + 1) / 2;
    __shared__ float4 red[kWavesPerBlock][NCH_H][64];
    // the blocks behind the first gridA share the launch: column sum of G for the fused self loop (dRz[loop])
    if ((int)blockIdx.x >= gridA) {
        colsum_partial_body(cs, (int)blockIdx.x - gridA, cs.nblocks, &red[0][0][0]);
        return;
    }
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = gridA * kWavesPerBlock;
    const int n_items = a.counts[0];
    Lanes<NCH, D4T> L;
    L.init(lane, a.D4);
    const int voff = 4 * L.D4();
    const float kappa = a.out_scale;
    const float inv_kappa = 1.f / kappa;
    float4 av[NCH_H];
    float4 da[NCH_H];
#pragma unroll
    for (int k = 0; k < NCH_H; ++k) {
        av[k] = sel4(L.is_h[k], ld4(a.a_att + (L.is_h[k] ? L.coff[k] : 0)));
        da[k] = f4zero();
    }
    const bool has_loop = a.loop_rel >= 0;
    const float* rloop = a.RR + (int64_t)(has_loop ? a.loop_rel : 0) * a.ldrr;
    float4 rl[NCH];                                    // the loop relation's [Rq|Rz] chunks: the same for every destination
#pragma unroll
    for (int k = 0; k < NCH; ++k) rl[k] = L.any_v(k) ? ld4(rloop + L.coffc[k]) : f4zero();

    // the first header (and the item's first two entries) at a clamped index: they do not wait for the device-side count
    const int it0 = blockIdx.x * kWavesPerBlock + wave;
    jmac_item_t item = a.items[min(it0, a.n_items_max - 1)];
    int4 e4 = make_int4(0, 0, 0, 0);
    if (a.item_edges) e4 = a.item_edges[min(it0, a.n_items_max - 1)];
    for (int it = it0; it < n_items; it += nwaves) {
        if (it != it0) {
            item = a.items[it];
            if (a.item_edges) e4 = a.item_edges[it];
        }
        const int i = item.seg;
        // first batch of entries: requested BEFORE the destination's own rows, so that the first gathers follow the rows'
        // round trip directly (items of up to two entries carry them inline: no request at all)
        int fcol, ftyp;
        {
            const int nb0 = min(64, item.end - item.beg);
            if (a.item_edges && nb0 <= 2) {
                fcol = (lane == 0 || nb0 < 2) ? e4.x : e4.z;
                ftyp = (lane == 0 || nb0 < 2) ? e4.y : e4.w;
            } else {
                const int idx = nb0 > 0 ? item.beg + min(lane, nb0 - 1) : 0;
                fcol = a.col[idx];
                ftyp = a.etype[idx];
            }
        }
        if (item.beg == item.end && item.pslot < 0) {
            // a destination without in-edges (more than half of a DBP-5L graph) contributes dP[i] = 0 and nothing else:
            // none of its rows is read.  (An empty QUARTER of a cooperative segment owns a partial slot: general path.)
#pragma unroll
            for (int k = 0; k < NCH_H; ++k)
                if (L.valid[k] && L.is_h[k]) st4(a.dP + (int64_t)i * a.lddp + L.coff[k], f4zero());
            continue;
        }
        float4 pv[NCH_H], gv[NCH], accP[NCH_H];
        float tpart = 0.f;
        const float* prow = a.P + (int64_t)i * a.ldp;
        const float* zrow = has_loop ? a.QZ + ((int64_t)i + a.self_off) * a.ldqz : prow;      // (no loop: any valid row, unused)
        const float* grow = a.G + (int64_t)i * a.ldg;
        const float* orow = a.out + (int64_t)i * a.ldo;
        // EVERY load of the destination's own data goes out before the first use: written chunk by chunk (load G, out; then,
        // behind `if (has_loop)`, Z and Rz[loop]; then the scalars behind the wave sum) the compiler kept that order and the
        // item paid five dependent round trips here instead of one (seen in the ISA: vmcnt(1) / vmcnt(0) pairs per chunk)
        float4 graw[NCH], oraw[NCH], zraw[NCH];
#pragma unroll
        for (int k = 0; k < NCH_H; ++k) pv[k] = ld4(prow + (L.is_h[k] ? L.coff[k] : 0));
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (!L.any_v(k)) continue;
            const int vo = L.is_v(k) ? L.coff[k] - voff : 0;
            graw[k] = ld4(grow + vo);
            oraw[k] = ld4(orow + vo);
            zraw[k] = ld4(zrow + (has_loop ? L.coffc[k] : 0));
        }
        const float m_i = a.seg_max[i];
        const float l_i = a.seg_den[i];
        const int rp0 = a.rowptr[i], rp1 = a.rowptr[i + 1];
#pragma unroll
        for (int k = 0; k < NCH_H; ++k) {
            pv[k] = sel4(L.is_h[k], pv[k]);
            accP[k] = f4zero();
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            gv[k] = f4zero();
            if (!L.any_v(k)) continue;
            const float4 g = mul4(graw[k], kappa);
            float4 nbv = mul4(oraw[k], inv_kappa);                 // nb_i = out/kappa - (Z[i] - Rz[loop])
            if (has_loop) nbv = sub4(nbv, sub4(zraw[k], rl[k]));
            gv[k] = sel4(L.is_v(k), g);
            // (off the v lanes nbv holds whatever the clamped loads fetched -- e.g. the Q half of row i, which callers that
            // project Q for SOURCE rows only never write: select it away, 0 x garbage is not 0)
            tpart += dot4(gv[k], sel4(L.is_v(k), nbv));
        }
        const float t_i = wave_sum(tpart);
        const float inv_l = l_i > 0.f ? 1.f / l_i : 0.f;
        const float c_i = sqrtf((float)(rp1 - rp0));

        for (int e0 = item.beg; e0 < item.end; e0 += 64) {
            const int nb = min(64, item.end - e0);
            const int le = min(lane, nb - 1);
            int my_col = fcol, my_typ = ftyp;
            if (e0 != item.beg) {
                my_col = a.col[e0 + le];
                my_typ = a.etype[e0 + le];
            }
            float my_w = 0.f, my_ds = 0.f;
            auto group = [&](auto uu_c, const int u0) {
                constexpr int UU = decltype(uu_c)::value;
                float4 q[UU][NCH], r[UU][NCH];
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const int j = bcast_i(my_col, u0 + u);
                    const int t = bcast_i(my_typ, u0 + u);
                    const float* qrow = a.QZ + (int64_t)j * a.ldqz;
                    const float* rrow = a.RR + (int64_t)t * a.ldrr;
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        q[u][k] = ld4(qrow + L.coffc[k]);
                        r[u][k] = ld4(rrow + L.coffc[k]);
                    }
                }
                JMAC_LOADS_FIRST(UU * NCH * 2);
                float4 hv[UU][NCH_H];
                float sp[UU], up[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    float spart = 0.f, upart = 0.f;
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        q[u][k] = sub4(q[u][k], r[u][k]);
                        if (k < NCH_H) {
                            hv[u][k] = add4(pv[k], q[u][k]);
                            if (L.any_h(k)) spart += dot4(av[k], leaky4<D4T != 0>(hv[u][k], a.slope));
                        }
                        if (L.any_v(k)) upart += dot4(gv[k], q[u][k]);   // gv is zero off the v-role lanes
                    }
                    sp[u] = spart;
                    up[u] = upart;
                }
                wave_sum_n<UU>(sp);
                wave_sum_n<UU>(up);
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const float alpha = fast_exp(sp[u] - m_i) * inv_l;
                    const float w = c_i * alpha;
                    const float ds = w * up[u] - alpha * t_i;
                    if (lane == u0 + u) {
                        my_w = w;
                        my_ds = ds;
                    }
                    unsigned mybits = 0u;
                    float4 dh[NCH_H];
#pragma unroll
                    for (int k = 0; k < NCH_H; ++k) {
                        const float4 h = hv[u][k];
                        const float4 g = make_float4(h.x > 0.f ? 1.f : a.slope, h.y > 0.f ? 1.f : a.slope,
                                                     h.z > 0.f ? 1.f : a.slope, h.w > 0.f ? 1.f : a.slope);
                        dh[k] = make_float4(ds * av[k].x * g.x, ds * av[k].y * g.y, ds * av[k].z * g.z, ds * av[k].w * g.w);
                        accP[k] = add4(accP[k], dh[k]);
                        da[k] = fma4(leaky4<D4T != 0>(h, a.slope), ds, da[k]);
                        if (MODE == 1)
                            mybits |= ((h.x > 0.f ? 1u : 0u) | (h.y > 0.f ? 2u : 0u) | (h.z > 0.f ? 4u : 0u) |
                                       (h.w > 0.f ? 8u : 0u)) << (4 * k);
                    }
                    if (MODE == 1) {
                        a.bits[(int64_t)(e0 + u0 + u) * 64 + lane] = (unsigned char)mybits;
                    } else {
                        const int j = bcast_i(my_col, u0 + u);
                        const int t = bcast_i(my_typ, u0 + u);
                        float* qd = a.dQZ + (int64_t)j * a.lddqz;
                        float* rd = a.dRR + (int64_t)t * a.lddrr;
#pragma unroll
                        for (int k = 0; k < NCH; ++k) {
                            if (!L.valid[k]) continue;
                            float4 v = mul4(gv[k], w);
                            if (k < NCH_H) v = L.is_h[k] ? dh[k] : v;
                            atomicAdd(qd + L.coff[k] + 0, v.x);
                            atomicAdd(qd + L.coff[k] + 1, v.y);
                            atomicAdd(qd + L.coff[k] + 2, v.z);
                            atomicAdd(qd + L.coff[k] + 3, v.w);
                            atomicAdd(rd + L.coff[k] + 0, -v.x);
                            atomicAdd(rd + L.coff[k] + 1, -v.y);
                            atomicAdd(rd + L.coff[k] + 2, -v.z);
                            atomicAdd(rd + L.coff[k] + 3, -v.w);
                        }
                    }
                }
            };
            int u0 = 0;
            for (; u0 + 2 <= nb; u0 += 2) group(std::integral_constant<int, 2>{}, u0);
            if (u0 < nb) group(std::integral_constant<int, 1>{}, u0);
            if (MODE == 1 && lane < nb) a.wds[e0 + lane] = make_float2(my_w, my_ds);
        }
        // dP row (h-role chunks)
        float* dprow = item.pslot < 0 ? a.dP + (int64_t)i * a.lddp : a.part + (int64_t)item.pslot * voff;
#pragma unroll
        for (int k = 0; k < NCH_H; ++k)
            if (L.valid[k] && L.is_h[k]) st4(dprow + L.coff[k], accP[k]);
    }
    // block-level reduction of the a_att gradient, one partial row per block
#pragma unroll
    for (int k = 0; k < NCH_H; ++k) red[wave][k][lane] = da[k];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < NCH_H; ++k) {
            if (L.valid[k] && L.is_h[k]) {
                float4 s = red[0][k][lane];
#pragma unroll
                for (int w = 1; w < kWavesPerBlock; ++w) s = add4(s, red[w][k][lane]);
                st4(a.da_part + (int64_t)blockIdx.x * voff + L.coff[k], s);
            }
        }
    }
}

// Pass B (by source, sign=+1) and pass C (by relation, sign=-1): sum the per-edge records.
//   row[h-half] = sign * sum_e ds_e * a (.) lrelu'(h_e)         (sign bits from pass A)
//   row[v-half] = sign * sum_e w_e * g_{dst(e)}  (+ g_j for the fused self loop in pass B)
// Both passes share ONE launch: blocks [0, grid_b) run pass B (arguments ab), the rest pass C (arguments ac).
// The body is instantiated once per pass, each reading ITS argument struct straight from the kernel arguments: a reference
// picked at run time (`is_b ? ab : ac`) made the compiler park the structs' pointers in scratch and reach everything through
// flat loads (40 B of private segment, a scratch load + vmcnt(0) per 64-edge batch).
template <int NCH, int U, int D4T>
__device__ __forceinline__ void rel_attn_bwd_gather_body(const BwdArgs& a, float* __restrict__ outp, const int64_t ldout, const int bid,
                                                         const int nblk) {
    constexpr int NCH_H = (NCH + 1) / 2;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = nblk * kWavesPerBlock;
    const int n_items = min(a.counts[0], a.n_items_max);
    Lanes<NCH, D4T> L;
    L.init(lane, a.D4);
    const int voff = 4 * L.D4();
    const float kappa = a.out_scale;
    float4 av[NCH_H];
    int goff[NCH];
#pragma unroll
    for (int k = 0; k < NCH_H; ++k) av[k] = sel4(L.is_h[k], ld4(a.a_att + (L.is_h[k] ? L.coff[k] : 0)));
#pragma unroll
    for (int k = 0; k < NCH; ++k) goff[k] = L.is_v(k) ? L.coff[k] - voff : 0;

    const int it0 = bid * kWavesPerBlock + wave;
    jmac_item_t item = a.items[max(min(it0, a.n_items_max - 1), 0)];     // clamped: does not wait for the device-side count
    int4 e4 = make_int4(0, 0, 0, 0);
    if (a.item_edges) e4 = a.item_edges[max(min(it0, a.n_items_max - 1), 0)];
    for (int it = it0; it < n_items; it += nwaves) {
        if (it != it0) {
            item = a.items[it];
            if (a.item_edges) e4 = a.item_edges[it];
        }
        const int seg = item.seg;
        float4 acc[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) acc[k] = f4zero();
        for (int x0 = item.beg; x0 < item.end; x0 += 64) {
            const int nb = min(64, item.end - x0);
            const int le = min(lane, nb - 1);
            int my_slot, my_dst;
            if (a.item_edges && x0 == item.beg && nb <= 2) {        // both entries arrived with the header
                my_slot = (lane == 0 || nb < 2) ? e4.x : e4.z;
                my_dst = (lane == 0 || nb < 2) ? e4.y : e4.w;
            } else {
                my_slot = a.order[x0 + le];
                my_dst = a.entry_dst ? a.entry_dst[x0 + le] : a.dst_of_slot[my_slot];
            }
            const float2 my_wd = a.wds[my_slot];
            auto group = [&](auto uu_c, const int u0) {
                constexpr int UU = decltype(uu_c)::value;
                float4 gq[UU][NCH];
                unsigned bw[UU];
                float w[UU], ds[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const int slot = bcast_i(my_slot, u0 + u);
                    const int i = bcast_i(my_dst, u0 + u);
                    w[u] = bcast_f(my_wd.x, u0 + u) * kappa;
                    ds[u] = bcast_f(my_wd.y, u0 + u);
                    const float* grow = a.G + (int64_t)i * a.ldg;
#pragma unroll
                    for (int k = 0; k < NCH; ++k)
                        if (L.any_v(k)) gq[u][k] = ld4(grow + goff[k]);
                    bw[u] = a.bits[(int64_t)slot * 64 + lane];
                }
                JMAC_LOADS_FIRST(UU * (NCH + 1));
#pragma unroll
                for (int u = 0; u < UU; ++u) {
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        float4 hpart = f4zero();
                        if (k < NCH_H && L.any_h(k)) {
                            const unsigned nib = bw[u] >> (4 * k);
                            hpart.x = ds[u] * av[k].x * ((nib & 1u) ? 1.f : a.slope);
                            hpart.y = ds[u] * av[k].y * ((nib & 2u) ? 1.f : a.slope);
                            hpart.z = ds[u] * av[k].z * ((nib & 4u) ? 1.f : a.slope);
                            hpart.w = ds[u] * av[k].w * ((nib & 8u) ? 1.f : a.slope);
                        }
                        if (L.any_h(k) && L.any_v(k)) {
                            const float4 vpart = mul4(gq[u][k], w[u]);
                            acc[k] = add4(acc[k], L.is_h[k] ? hpart : vpart);
                        } else if (L.any_h(k)) {
                            acc[k] = add4(acc[k], hpart);
                        } else {
                            acc[k] = fma4(gq[u][k], w[u], acc[k]);
                        }
                    }
                }
            };
            int u0 = 0;
            for (; u0 + 4 <= nb; u0 += 4) group(std::integral_constant<int, 4>{}, u0);
            if (u0 + 2 <= nb) {
                group(std::integral_constant<int, 2>{}, u0);
                u0 += 2;
            }
            if (u0 < nb) group(std::integral_constant<int, 1>{}, u0);
        }
        const bool direct = item.pslot < 0;
        // the fused self loop contributes g_j to dZ[j]; it is added once, by the finalising item or
        // (for split segments) by the combine pass.
        float* row = direct ? outp + (int64_t)seg * ldout : a.part + (int64_t)item.pslot * (2 * voff);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (!L.valid[k]) continue;
            float4 v = mul4(acc[k], a.sign);
            if (direct && a.add_self && !L.is_h[k] && seg >= a.self_off && seg - a.self_off < a.N)
                v = fma4(ld4(a.G + ((int64_t)seg - a.self_off) * a.ldg + (L.coff[k] - voff)), kappa, v);
            st4(row + L.coff[k], v);
        }
    }
}

template <int NCH, int U, int D4T>
__global__ __launch_bounds__(kBlock) void rel_attn_bwd_gather_kernel(BwdArgs ab, float* __restrict__ outb, int64_t ldoutb, int grid_b,
                                                                     BwdArgs ac, float* __restrict__ outc, int64_t ldoutc) {
    if ((int)blockIdx.x < grid_b)                                     // block-uniform
        rel_attn_bwd_gather_body<NCH, U, D4T>(ab, outb, ldoutb, (int)blockIdx.x, grid_b);
    else
        rel_attn_bwd_gather_body<NCH, U, D4T>(ac, outc, ldoutc, (int)blockIdx.x - grid_b, (int)gridDim.x - grid_b);
}

// atomic mode: dQZ starts at [0 | kappa*G[i]] (the fused self loop's dZ term) or at zero
__global__ void init_dqz_kernel(float* __restrict__ dQZ, int64_t lddqz, int64_t Nsrc, int64_t d, const float* __restrict__ G,
                                int64_t ldg, float kappa, int add_self, int64_t self_off, int64_t n_self) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Nsrc * 2 * d) {
        int64_t r = i / (2 * d), c = i % (2 * d);
        const bool self = add_self && c >= d && r >= self_off && r - self_off < n_self;
        dQZ[r * lddqz + c] = self ? kappa * G[(r - self_off) * ldg + (c - d)] : 0.f;
    }
}

__global__ void fill_rows_kernel(float* __restrict__ p, int64_t rows, int64_t width, int64_t ld, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * width) p[(i / width) * ld + (i % width)] = v;
}

inline int check_dims(int64_t d, int64_t ld0, int64_t ld1, int64_t ld2) {
    if (d <= 0 || d % 4 != 0 || d > 512) return JMAC_EDIM;
    if (ld0 % 4 || ld1 % 4 || ld2 % 4) return JMAC_EDIM;
    return 0;
}

inline int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

inline unsigned persist_grid(int64_t n_items_max) {
    static const int cap = env_int("JMAC_GRID", kPersistBlocks);   // tuning knob (debug)
    int64_t need = (n_items_max + kWavesPerBlock - 1) / kWavesPerBlock;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

// forward: graphs of up to 16k items get one wave per item (measured on the DBP-5L ja shape: 21 us against 24 us
// with the persistent 2048-block grid -- at that size the kernel is a chain of dependent round trips, not a stream)
inline unsigned fwd_grid(int64_t n_items_max) {
    static const int cap = env_int("JMAC_GRID", 0);
    if (cap > 0) return persist_grid(n_items_max);
    const int64_t need = (n_items_max + kWavesPerBlock - 1) / kWavesPerBlock;
    if (need <= 2 * kPersistBlocks) return (unsigned)(need < 1 ? 1 : need);
    return kPersistBlocks;
}

inline unsigned split_grid(int64_t n_splits_max) {
    if (n_splits_max < 1) n_splits_max = 1;
    return (unsigned)(n_splits_max < 4 * kPersistBlocks ? n_splits_max : 4 * kPersistBlocks);
}

#define JMAC_DISPATCH_NCH(nch, ...)                       \
    switch (nch) {                                        \
        case 1: { constexpr int NCH = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int NCH = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int NCH = 3; __VA_ARGS__; } break; \
        default: { constexpr int NCH = 4; __VA_ARGS__; } break; \
    }

// d = 256 and d = 300 (the reference's default and BASELINE's dim) get compile-time chunk roles
#define JMAC_DISPATCH_D(D4v, nch, ...)                                        \
    if ((D4v) == 64 && slope01) { constexpr int NCH = 2; constexpr int D4T = 64; __VA_ARGS__; } \
    else if ((D4v) == 75 && slope01) { constexpr int NCH = 3; constexpr int D4T = 75; __VA_ARGS__; } \
    else { constexpr int D4T = 0; JMAC_DISPATCH_NCH(nch, __VA_ARGS__); }

}  // namespace

namespace jmac {
void launch_reduce_rows(const float* partial, int nparts, int W, float scale, float* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((W + RR_COLS - 1) / RR_COLS)), dim3(1024), 0, st, partial, nparts, W, scale, out);
}
}  // namespace jmac

// What launch_rel_attn_fwd decided for one call, for jmac_rel_attn_aggregate_fwd_jobs_f32: a call whose form is the plain
// small-graph kernel (one wave per item) is NOT launched when a plan is asked for -- the caller pairs two of them in one launch
struct FwdPlan {
    FwdArgs a;
    unsigned grid;
    int nch;
    bool plain_small, slope01, launched;
    int64_t n_comb;                      // split (non-cooperative) destinations: the combine kernel follows the main launch
};

// dh: pitch of the halves of a [Q|Z] / [Rq|Rz] row in elements (Q at 0, Z at dh): d, or the padded pitch of the bf16 tables
template <typename TT>
static int launch_rel_attn_fwd(const TT* P, int64_t ldp, const TT* QZ, int64_t ldqz, const TT* RR, int64_t ldrr, int64_t dh,
                               const float* a_att,
                               const int32_t* col, const int32_t* etype, const jmac_view_t* v, int64_t N, int64_t d, float slope,
                               int32_t loop_rel, int64_t self_off, float out_scale, float* out, int64_t ldo, float* seg_max,
                               float* seg_den, void* ws, size_t ws_bytes, jmac_stream_t stream, FwdPlan* plan = nullptr) {
    if (plan) *plan = FwdPlan{};
    if (N < 0 || !v || v->n_items_max < 0 || v->n_splits_max < 0) return JMAC_EINVAL;
    if (N == 0) {
        if (plan) plan->launched = true;
        return JMAC_OK;
    }
    if (!P || !QZ || !RR || !a_att || !v->ptr || !v->items || !v->counts || !out || !seg_max || !seg_den) return JMAC_EINVAL;
    if (int rc = check_dims(d, ldp, ldqz, ldrr)) return rc;
    if (ldo % 4) return JMAC_EDIM;
    if (N >= INT32_MAX) return JMAC_ERANGE;
    const int64_t n_items_max = v->n_items_max, n_splits_max = v->n_splits_max;
    int64_t n_parts_max = v->n_parts_max, n_empty = v->n_empty, n_coop = v->n_coop;
    if (n_empty < 0 || n_coop < 0 || n_coop * kWavesPerBlock + n_empty > n_items_max || n_coop > n_splits_max) return JMAC_EINVAL;
    FwdArgs a;
    a.P = P; a.QZ = QZ; a.RR = RR; a.a_att = a_att;
    a.ldp = ldp; a.ldqz = ldqz; a.ldrr = ldrr; a.ldo = ldo;
    a.rowptr = v->ptr; a.col = col; a.etype = etype;
    a.items = v->items; a.splits = v->splits; a.counts = v->counts;
    a.item_edges = reinterpret_cast<const int4*>(v->item_edges);
    if (loop_rel >= 0 && self_off < 0) return JMAC_EINVAL;
    a.N = (int32_t)N; a.D4 = (int32_t)(d / 4); a.loop_rel = loop_rel; a.self_off = loop_rel >= 0 ? self_off : 0;
    if (dh < d || dh % 4) return JMAC_EDIM;
    a.zoff = (int32_t)dh;
    a.n_items_max = (int32_t)(n_items_max > 0 ? n_items_max : 1);
    a.n_coop = (int32_t)n_coop;
    a.slope = slope; a.out_scale = out_scale;
    a.out = out; a.seg_max = seg_max; a.seg_den = seg_den;
    // ws layout: [part_ml: 2 floats per slot][part_acc: d floats per slot]
    if (n_parts_max < 0) n_parts_max = 0;
    if (n_parts_max > 0 && (!ws || ws_bytes < jmac_rel_attn_fwd_workspace_bytes(n_parts_max, d))) return JMAC_EWORKSPACE;
    a.part_ml = (float*)ws;
    a.part_acc = ws ? (float*)((char*)ws + align_up((size_t)n_parts_max * 8)) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)((2 * (d / 4) + 63) / 64);
    // work units: cooperative segments (a block each) + items (a wave each) + packs of kEmptyPack empty segments
    // (a wave each).  One-wave-per-unit grids cover max(blocks for the cooperative segments, blocks for the rest):
    // block b first does cooperative segment b, then its items.
    const int64_t n_wave_units = n_items_max - n_coop * kWavesPerBlock - n_empty + (n_empty + kEmptyPack - 1) / kEmptyPack;
    int64_t need = (n_wave_units + kWavesPerBlock - 1) / kWavesPerBlock;
    if (need < n_coop) need = n_coop;
    if (need < 1) need = 1;
    static const int grid_env = env_int("JMAC_GRID", 0);               // tuning knob (debug)
    unsigned grid;
    if (grid_env > 0) grid = (unsigned)(need < grid_env ? need : grid_env);
    else grid = (unsigned)(need <= 2 * kPersistBlocks ? need : kPersistBlocks);
    // small graphs (one wave per item) are bound by round trips, not by gathers in flight: the 2-edge group
    // is the faster form there (DBP-5L ja: 20.4 us against 22.6 us).  "Small" is the host's call (jmac_amd/graph.py:
    // schedules that carry their items' first entries inline, the real el + ja pair step: 1.571 -> 1.527 ms)
    static const int fwd_u_env = env_int("JMAC_FWD_U", 0);         // tuning knob (debug)
    const int fwd_u = fwd_u_env ? fwd_u_env : ((v->item_edges || n_items_max <= 8 * kPersistBlocks) ? 2 : 4);
    const bool slope01 = slope >= 0.f && slope <= 1.f;
    // half-wave lane map (rel_attn_fwd_hw_kernel): d = 256 / 300 with 16-byte aligned rows and halves
    static const int hw_env = env_int("JMAC_FWD_HW", 1);             // tuning knobs (debug)
    constexpr int CHE = 16 / (int)sizeof(TT);                          // elements per 16-byte chunk
    const bool aligned = ((((uintptr_t)P | (uintptr_t)QZ | (uintptr_t)RR) & 15) == 0) && ldp % CHE == 0 && ldqz % CHE == 0 &&
                         ldrr % CHE == 0 && dh % CHE == 0;
    const int dc = (int)(dh / CHE);
    const bool hw_shape = sizeof(TT) == 4 ? ((d == 300 || d == 256) && dh == d) : ((d == 300 && dh == 304) || (d == 256 && dh == 256));
    static const int hw_small_env = env_int("JMAC_FWD_HW_SMALL", 0);
    // measured (rocprofv3 / HIP events, MI355X): config 4 (persistent grid) bf16 7.53 -> 6.48 ms (0.43 -> 0.50 of the HBM peak),
    // fp32 10.93 -> 10.45 / 9.67 ms; on the 56 589-entity union (one wave per item) the 64-lane kernel stays ahead
    // (fp32 116 against 120 us, bf16 87 against 105 us): the small-graph form keeps it
    const bool use_hw = hw_env && slope01 && aligned && hw_shape && (fwd_u != 2 || hw_small_env);
    if (plan) {
        plan->a = a; plan->grid = grid; plan->nch = nch; plan->slope01 = slope01; plan->n_comb = n_splits_max - n_coop;
        plan->plain_small = !use_hw && fwd_u == 2 && sizeof(TT) == 4;
        if (plan->plain_small) return JMAC_OK;          // the caller launches it (paired with another job where it can)
        plan->launched = true;
    }
    if (use_hw) {
        static const int hw_depth_env = env_int("JMAC_FWD_HW_DEPTH", 0);
        // gathers in flight per wave (persistent form): two groups, EVERY path issuing the same loads (PIPE 12: exact vmcnt waits).
        // Config 4, MI355X: bf16 6.9 ms with the three-deep conditional form this replaced (one vmcnt(0) per trip in its ISA)
        // -> 6.1-6.2 ms two- or three-deep uniform (0.47 -> 0.53 of the HBM peak); four-deep conditional 7.2 ms; held to three waves
        // per SIMD (168 VGPRs) 6.3 ms.  fp32: 10.3 ms either way (the launch moves 62 GB at 6 TB/s: the fabric's rate); the
        // conditional two-deep form stays selectable (JMAC_FWD_HW_DEPTH=2) for A/B runs.
        const int depth = fwd_u == 2 ? 0 : (hw_depth_env ? hw_depth_env : 12);
        // experiments (default off, profiles/r4_pmc_config4_experiments.json): the hottest relation rows (ids 0 .. HOT-1: tables
        // whose relation ids are ordered by frequency) in LDS; streaming [Q|Z] gathers
        static const int hot_env = env_int("JMAC_FWD_HOT", 0);
        static const int nt_env = env_int("JMAC_FWD_NT", 0);
        constexpr int HOTN = sizeof(TT) == 2 ? 40 : 20;                 // 48 KB of rows either way
        const bool hot = hot_env && depth == 12 && v->n_items_max > 0 && loop_rel >= HOTN &&
                         (uint64_t)(loop_rel + 1) * (uint64_t)ldrr * sizeof(TT) < 0xFFFFFFF0ull;
#define JMAC_HW_LAUNCH(DCv)                                                                                                   \
        do {                                                                                                                  \
            if (hot) hipLaunchKernelGGL((rel_attn_fwd_hw_kernel<DCv, 12, TT, HOTN>), dim3(grid), dim3(kBlock), 0, st, a);     \
            else if (nt_env && depth == 12) hipLaunchKernelGGL((rel_attn_fwd_hw_kernel<DCv, 12, TT, 0, 1>), dim3(grid), dim3(kBlock), 0, st, a); \
            else if (depth == 12) hipLaunchKernelGGL((rel_attn_fwd_hw_kernel<DCv, 12, TT>), dim3(grid), dim3(kBlock), 0, st, a); \
            else if (depth == 2) hipLaunchKernelGGL((rel_attn_fwd_hw_kernel<DCv, 2, TT>), dim3(grid), dim3(kBlock), 0, st, a);  \
            else hipLaunchKernelGGL((rel_attn_fwd_hw_kernel<DCv, 0, TT>), dim3(grid), dim3(kBlock), 0, st, a);                  \
        } while (0)
        if constexpr (sizeof(TT) == 4) {
            if (dc == 75) JMAC_HW_LAUNCH(75); else JMAC_HW_LAUNCH(64);
        } else {
            if (dc == 38) JMAC_HW_LAUNCH(38); else JMAC_HW_LAUNCH(32);
        }
#undef JMAC_HW_LAUNCH
    } else if (fwd_u == 2) {
        JMAC_DISPATCH_D(a.D4, nch, hipLaunchKernelGGL((rel_attn_fwd_kernel<NCH, 2, D4T, TT>), dim3(grid), dim3(kBlock), 0, st, a));
    } else {
        JMAC_DISPATCH_D(a.D4, nch, hipLaunchKernelGGL((rel_attn_fwd_kernel<NCH, 4, D4T, TT>), dim3(grid), dim3(kBlock), 0, st, a));
    }
    if (n_splits_max - n_coop > 0) {
        const unsigned g2 = split_grid(n_splits_max - n_coop);
        JMAC_DISPATCH_NCH(nch, hipLaunchKernelGGL((rel_attn_fwd_combine_kernel<NCH, TT>), dim3(g2), dim3(kBlock), 0, st, a));
    }
    return (int)hipGetLastError();
}

extern "C" {

size_t jmac_rel_attn_fwd_workspace_bytes(int64_t n_parts_max, int64_t d) {
    if (n_parts_max < 0) n_parts_max = 0;
    return align_up((size_t)n_parts_max * 8) + align_up((size_t)n_parts_max * d * 4) + 256;
}

int jmac_rel_attn_aggregate_fwd_f32(const float* P, int64_t ldp, const float* QZ, int64_t ldqz, const float* RR,
                                    int64_t ldrr, const float* a_att, const int32_t* col, const int32_t* etype,
                                    const jmac_view_t* by_dst, int64_t N, int64_t d, float slope, int32_t loop_rel,
                                    int64_t self_off, float out_scale, float* out, int64_t ldo, float* seg_max, float* seg_den,
                                    void* ws, size_t ws_bytes, jmac_stream_t stream) {
    return launch_rel_attn_fwd<float>(P, ldp, QZ, ldqz, RR, ldrr, d, a_att, col, etype, by_dst, N, d, slope, loop_rel, self_off,
                                      out_scale, out, ldo, seg_max, seg_den, ws, ws_bytes, stream);
}

int jmac_rel_attn_aggregate_fwd_jobs_f32(const jmac_agg_fwd_job_t* jobs, int32_t n_jobs, jmac_stream_t stream) {
    if (!jobs || n_jobs < 1 || n_jobs > 2) return JMAC_EINVAL;
    FwdPlan pl[2];
    for (int k = 0; k < n_jobs; ++k) {
        const jmac_agg_fwd_job_t& q = jobs[k];
        if (int rc = launch_rel_attn_fwd<float>(q.P, q.ldp, q.QZ, q.ldqz, q.RR, q.ldrr, q.d, q.a_att, q.col, q.etype, q.by_dst, q.N, q.d,
                                                q.slope, q.loop_rel, q.self_off, q.out_scale, q.out, q.ldo, q.seg_max, q.seg_den, q.ws,
                                                q.ws_bytes, stream, &pl[k]))
            return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    auto single = [&](const FwdPlan& p) {
        const FwdArgs& a = p.a;
        const bool slope01 = p.slope01;
        const int nch = p.nch;
        JMAC_DISPATCH_D(a.D4, nch, hipLaunchKernelGGL((rel_attn_fwd_kernel<NCH, 2, D4T, float>), dim3(p.grid), dim3(kBlock), 0, st, a));
    };
    const bool pair = n_jobs == 2 && !pl[0].launched && !pl[1].launched && pl[0].plain_small && pl[1].plain_small &&
                      pl[0].nch == pl[1].nch && pl[0].a.D4 == pl[1].a.D4 && pl[0].slope01 == pl[1].slope01;
    if (pair) {
        const bool slope01 = pl[0].slope01;
        const int nch = pl[0].nch;
        const unsigned grid = pl[0].grid + pl[1].grid;
        JMAC_DISPATCH_D(pl[0].a.D4, nch, hipLaunchKernelGGL((rel_attn_fwd_jobs_kernel<NCH, 2, D4T, float>), dim3(grid), dim3(kBlock), 0,
                                                            st, pl[0].a, pl[1].a, (int)pl[0].grid));
    } else {
        for (int k = 0; k < n_jobs; ++k)
            if (!pl[k].launched) single(pl[k]);
    }
    for (int k = 0; k < n_jobs; ++k) {
        if (pl[k].launched || pl[k].n_comb <= 0) continue;          // (a job launched by launch_rel_attn_fwd ran its combine there)
        const FwdArgs& a = pl[k].a;
        const int nch = pl[k].nch;
        const unsigned g2 = split_grid(pl[k].n_comb);
        JMAC_DISPATCH_NCH(nch, hipLaunchKernelGGL((rel_attn_fwd_combine_kernel<NCH, float>), dim3(g2), dim3(kBlock), 0, st, a));
    }
    return (int)hipGetLastError();
}

int jmac_rel_attn_aggregate_fwd_bf16(const uint16_t* P, int64_t ldp, const uint16_t* QZ, int64_t ldqz, const uint16_t* RR,
                                     int64_t ldrr, const float* a_att, const int32_t* col, const int32_t* etype,
                                     const jmac_view_t* by_dst, int64_t N, int64_t d, float slope, int32_t loop_rel,
                                     int64_t self_off, float out_scale, float* out, int64_t ldo, float* seg_max, float* seg_den,
                                     void* ws, size_t ws_bytes, jmac_stream_t stream) {
    return launch_rel_attn_fwd<bf16_t>(P, ldp, QZ, ldqz, RR, ldrr, d, a_att, col, etype, by_dst, N, d, slope, loop_rel, self_off,
                                       out_scale, out, ldo, seg_max, seg_den, ws, ws_bytes, stream);
}

int jmac_rel_attn_aggregate_fwd_bf16_padded(const uint16_t* P, int64_t ldp, const uint16_t* QZ, int64_t ldqz, const uint16_t* RR,
                                            int64_t ldrr, int64_t dh, const float* a_att, const int32_t* col, const int32_t* etype,
                                            const jmac_view_t* by_dst, int64_t N, int64_t d, float slope, int32_t loop_rel,
                                            int64_t self_off, float out_scale, float* out, int64_t ldo, float* seg_max,
                                            float* seg_den, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    return launch_rel_attn_fwd<bf16_t>(P, ldp, QZ, ldqz, RR, ldrr, dh, a_att, col, etype, by_dst, N, d, slope, loop_rel, self_off,
                                       out_scale, out, ldo, seg_max, seg_den, ws, ws_bytes, stream);
}

// ---- merge of per-source-chunk partial aggregations (slab-pipelined exchange, jmac_amd/dist.py) --------------------------------
// Part c is the forward above run on the edges whose SOURCE lies in chunk c (no self loop, out_scale 1): out_c[i] =
// sqrt(deg_c) * sum_e softmax_c(e) x_e with the chunk's own (max m_c, denominator l_c).  Over all edges of destination i:
//   M = max_c m_c,  L = sum_c exp(m_c - M) l_c,  nb[i] = sqrt(deg) * sum_c (exp(m_c - M) l_c / L) * out_c[i] / sqrt(deg_c)
// (M, L) are what the backward of the whole graph expects as seg_max / seg_den.  With a self table the layer's fused epilogue
// is applied here: out[i] = out_scale * (nb[i] + Zself[i] - rz_loop)  (src/jmac_model.py:49-50), else out_scale * nb[i].
struct MergeArgs {
    const float* out[JMAC_MERGE_MAX_PARTS];
    const float* smax[JMAC_MERGE_MAX_PARTS];
    const float* sden[JMAC_MERGE_MAX_PARTS];
    const int32_t* rowptr[JMAC_MERGE_MAX_PARTS];
    int n_parts;
    int64_t N, ldo, ldn;
    int d4;
    float* nb;
    float* seg_max;
    float* seg_den;
    const float* zself;
    const float* rz_loop;
    int64_t ldz;
    float out_scale;
};

__global__ __launch_bounds__(kBlock) void softmax_parts_merge_kernel(MergeArgs a) {
    const int lane = lane_id();
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= a.N) return;
    float m[JMAC_MERGE_MAX_PARTS], l[JMAC_MERGE_MAX_PARTS];
    int deg[JMAC_MERGE_MAX_PARTS];
    float M = -INFINITY;
    int dtot = 0;
#pragma unroll
    for (int c = 0; c < JMAC_MERGE_MAX_PARTS; ++c) {
        if (c < a.n_parts) {
            m[c] = a.smax[c][i];
            l[c] = a.sden[c][i];
            deg[c] = a.rowptr[c][i + 1] - a.rowptr[c][i];
        } else {
            m[c] = -INFINITY;
            l[c] = 0.f;
            deg[c] = 0;
        }
        if (deg[c] > 0) M = fmaxf(M, m[c]);
        dtot += deg[c];
    }
    float L = 0.f, f[JMAC_MERGE_MAX_PARTS];
#pragma unroll
    for (int c = 0; c < JMAC_MERGE_MAX_PARTS; ++c) {
        f[c] = deg[c] > 0 ? __expf(m[c] - M) * l[c] : 0.f;      // parts are visited in index order: a fixed summation order
        L += f[c];
    }
    const float sq = sqrtf((float)dtot);
#pragma unroll
    for (int c = 0; c < JMAC_MERGE_MAX_PARTS; ++c) f[c] = (deg[c] > 0 && L > 0.f) ? f[c] / L * sq * rsqrtf((float)deg[c]) : 0.f;
    for (int q = lane; q < a.d4; q += 64) {
        float4 acc = f4zero();
#pragma unroll
        for (int c = 0; c < JMAC_MERGE_MAX_PARTS; ++c)
            if (c < a.n_parts && deg[c] > 0) acc = fma4(ld4(a.out[c] + i * a.ldo + 4 * q), f[c], acc);
        if (a.zself) acc = add4(acc, sub4(ld4(a.zself + i * a.ldz + 4 * q), ld4(a.rz_loop + 4 * q)));
        st4(a.nb + i * a.ldn + 4 * q, mul4(acc, a.out_scale));
    }
    if (lane == 0) {
        a.seg_max[i] = dtot > 0 ? M : -INFINITY;
        a.seg_den[i] = L;
    }
}

int jmac_softmax_parts_merge_f32(const float* const* h_out, int64_t ldo, const float* const* h_seg_max, const float* const* h_seg_den,
                                 const int32_t* const* h_rowptr, int32_t n_parts, int64_t N, int64_t d, const float* Zself,
                                 int64_t ldz, const float* rz_loop, float out_scale, float* nb, int64_t ldn, float* seg_max,
                                 float* seg_den, jmac_stream_t stream) {
    if (n_parts < 0 || n_parts > JMAC_MERGE_MAX_PARTS) return JMAC_EINVAL;
    if (d <= 0 || d % 4 || ldo % 4 || ldn % 4 || d > ldo || d > ldn) return JMAC_EDIM;
    if (Zself && (!rz_loop || ldz % 4 || d > ldz)) return JMAC_EINVAL;
    if (N < 0 || (N > 0 && (!nb || !seg_max || !seg_den)) || (n_parts > 0 && (!h_out || !h_seg_max || !h_seg_den || !h_rowptr)))
        return JMAC_EINVAL;
    if (N == 0) return JMAC_OK;
    MergeArgs a{};
    for (int c = 0; c < n_parts; ++c) {
        if (!h_out[c] || !h_seg_max[c] || !h_seg_den[c] || !h_rowptr[c]) return JMAC_EINVAL;
        a.out[c] = h_out[c];
        a.smax[c] = h_seg_max[c];
        a.sden[c] = h_seg_den[c];
        a.rowptr[c] = h_rowptr[c];
    }
    a.n_parts = n_parts;
    a.N = N;
    a.ldo = ldo;
    a.ldn = ldn;
    a.d4 = (int)(d / 4);
    a.nb = nb;
    a.seg_max = seg_max;
    a.seg_den = seg_den;
    a.zself = Zself;
    a.rz_loop = rz_loop;
    a.ldz = ldz;
    a.out_scale = out_scale;
    hipLaunchKernelGGL(softmax_parts_merge_kernel, dim3((unsigned)((N + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// ---- backward workspace carving (shared by the size query and the launcher) ----
struct BwdWs {
    size_t da_part, colsum_part, part_dst, part_src, part_rel, wds, bits, total;
};

static BwdWs bwd_ws_layout(int64_t E, int64_t d, int64_t pd, int64_t ps, int64_t pr, int32_t mode) {
    BwdWs w;
    size_t off = 0;
    // the per-edge records come BEFORE the partial rows and the by-source partials LAST: a caller that runs the passes in
    // separate calls (jmac_rel_attn_aggregate_bwd_phases_f32: pass B slab by slab, each slab with its own by-source view)
    // finds the records of pass A at the same offsets whatever that view's partial count is
    w.da_part = off;     off += align_up((size_t)2 * kPersistBlocks * d * 4);   // one partial row per pass-A block (<= 2 x 2048)
    w.colsum_part = off; off += align_up((size_t)kPersistBlocks * d * 4);
    w.wds = off;         off += mode ? align_up((size_t)E * 8) : 0;
    w.bits = off;        off += mode ? align_up((size_t)E * 64) : 0;
    w.part_dst = off;    off += align_up((size_t)(pd > 0 ? pd : 0) * d * 4);
    w.part_rel = off;    off += mode ? align_up((size_t)(pr > 0 ? pr : 0) * 2 * d * 4) : 0;
    w.part_src = off;    off += mode ? align_up((size_t)(ps > 0 ? ps : 0) * 2 * d * 4) : 0;
    w.total = off + 256;
    return w;
}

size_t jmac_rel_attn_bwd_workspace_bytes(int64_t N, int64_t E, int64_t nrel, int64_t d, int64_t n_parts_max_dst,
                                         int64_t n_parts_max_src, int64_t n_parts_max_rel, int32_t mode) {
    (void)N; (void)nrel;
    if (d <= 0) return 0;
    return bwd_ws_layout(E < 0 ? 0 : E, d, n_parts_max_dst, n_parts_max_src, n_parts_max_rel, mode).total;
}

}  // extern "C"

// phases (deterministic mode): 1 = pass A (+ the column-sum partials of G), 2 = pass B by source + the merge of its split
// sources, 4 = pass C by relation + its merge, 8 = the merges that hang on pass A alone (split destinations, da, dRz[loop]).
// 15 = the whole backward.  A call with phase 2 alone may address a SLAB of the source rows (see the header).
enum { kPhaseA = 1, kPhaseB = 2, kPhaseC = 4, kPhaseM = 8, kPhaseAll = 15 };

static int rel_attn_bwd_impl(const float* P, int64_t ldp, const float* QZ, int64_t ldqz, const float* RR,
                             int64_t ldrr, const float* a_att, const int32_t* col, const int32_t* etype,
                             const int32_t* dst_of_slot, const jmac_view_t* by_dst, const jmac_view_t* by_src,
                             const jmac_view_t* by_rel, int64_t N, int64_t Nsrc, int64_t E, int64_t nrel, int64_t d,
                             float slope, int32_t loop_rel, int64_t self_off, float out_scale, const float* out, int64_t ldo,
                             const float* seg_max, const float* seg_den, const float* G, int64_t ldg, float* dP,
                             int64_t lddp, float* dQZ, int64_t lddqz, float* dRR, int64_t lddrr, float* da,
                             int32_t mode, void* ws, size_t ws_bytes, jmac_stream_t stream, int32_t phases) {
    if (N < 0 || Nsrc < 0 || E < 0 || nrel <= 0) return JMAC_EINVAL;
    if (loop_rel < 0) self_off = 0;
    // the fused self term reads QZ[self_off + i] (pass A) and adds g_i to d[Q|Z][self_off + i] (pass B).  A pass-B-only call on a
    // slab of the source rows passes self_off RELATIVE to the slab (possibly negative / past it: rows outside get no self term)
    if ((phases & kPhaseA) && loop_rel >= 0 && (self_off < 0 || self_off + N > Nsrc)) return JMAC_EINVAL;
    // a rank of the destination-sharded layer may own NO row (N == 0): its row-indexed buffers are empty (null), and the
    // call still has to produce d[Q|Z] (zeros: no edge reads the table from here), dRR and da
    if (!QZ || !RR || !a_att || !by_dst || !seg_max || !seg_den || !dQZ || !dRR || !da) return JMAC_EINVAL;
    if (N > 0 && (!P || !out || !G || !dP)) return JMAC_EINVAL;
    if (mode != 0 && (!by_src || !by_rel || !dst_of_slot)) return JMAC_EINVAL;
    if (int rc = check_dims(d, ldp, ldqz, ldrr)) return rc;
    if (ldo % 4 || ldg % 4 || lddp % 4 || lddqz % 4 || lddrr % 4) return JMAC_EDIM;
    if (out_scale == 0.f) return JMAC_EINVAL;
    const BwdWs w = bwd_ws_layout(E, d, by_dst->n_parts_max, mode ? by_src->n_parts_max : 0,
                                  mode ? by_rel->n_parts_max : 0, mode);
    if (!ws || ws_bytes < w.total) return JMAC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* wsb = (char*)ws;
    const int D4 = (int)(d / 4);
    const int nch = (2 * D4 + 63) / 64;

    BwdArgs a;
    a.P = P; a.QZ = QZ; a.RR = RR; a.a_att = a_att; a.out = out; a.seg_max = seg_max; a.seg_den = seg_den; a.G = G;
    a.ldp = ldp; a.ldqz = ldqz; a.ldrr = ldrr; a.ldo = ldo; a.ldg = ldg; a.lddp = lddp; a.lddqz = lddqz; a.lddrr = lddrr;
    a.rowptr = by_dst->ptr; a.col = col; a.etype = etype; a.dst_of_slot = dst_of_slot;
    a.items = by_dst->items; a.splits = by_dst->splits; a.counts = by_dst->counts; a.order = nullptr;
    a.N = (int32_t)N; a.D4 = D4; a.loop_rel = loop_rel; a.nrel = (int32_t)nrel; a.self_off = self_off;
    a.slope = slope; a.out_scale = out_scale;
    a.dP = dP; a.dQZ = dQZ; a.dRR = dRR;
    a.da_part = (float*)(wsb + w.da_part);
    a.part = (float*)(wsb + w.part_dst);
    a.wds = (float2*)(wsb + w.wds);
    a.bits = (unsigned char*)(wsb + w.bits);
    a.sign = 1.f; a.add_self = 0;
    a.item_edges = reinterpret_cast<const int4*>(by_dst->item_edges);
    a.entry_dst = nullptr;
    a.n_items_max = (int32_t)(by_dst->n_items_max > 0 ? by_dst->n_items_max : 1);

    [[maybe_unused]] const int T = 256;                            // (atomic test mode only)
    const unsigned gridA = fwd_grid(by_dst->n_items_max);          // small graphs: one wave per item, like the forward
    const bool slope01 = slope >= 0.f && slope <= 1.f;
    // column sum of G for the fused self loop (dRz[loop] -= kappa * sum_i G[i]): per-block partials by extra blocks of
    // the pass-A launch, reduced in the last launch
    float* colsum_part = (float*)(wsb + w.colsum_part);
    ColsumTask cs{};
    cs.X = G; cs.ldx = ldg; cs.R = N; cs.W4 = D4; cs.partial = colsum_part;
    cs.nblocks = (loop_rel >= 0 && N > 0) ? (int)(N < 512 ? N : 512) : 0;
    const unsigned gcs = (unsigned)cs.nblocks;
    auto sum_task = [&](const jmac_view_t* v, const float* part, int W4, float* outp, int64_t ldout, int add_self) {
        SumPartsTask t{};
        t.splits = v->splits; t.counts = v->counts; t.part = part; t.W4 = W4; t.sign = 1.f; t.outp = outp; t.ldout = ldout;
        t.G = G; t.ldg = ldg; t.D4 = D4; t.kappa = out_scale; t.add_self = add_self; t.self_off = self_off; t.n_self = N;
        t.nblocks = v->n_splits_max > 0 ? (int)split_grid(v->n_splits_max) : 0;
        return t;
    };
    auto reduce_task = [&](const float* partial, int nparts, float scale, float* outp) {
        ReduceTask t{};
        t.partial = partial; t.nparts = nparts; t.W = (int)d; t.scale = scale; t.outp = outp;
        t.nblocks = nparts > 0 ? (int)((d + RT_COLS - 1) / RT_COLS) : 0;
        return t;
    };
#ifndef JMAC_TEST_ATOMIC_BWD
    // the product library carries the deterministic backward only; the atomic variant (an independent second implementation
    // for the parity tests) is compiled into libjmac_hip_testing.so (csrc/Makefile)
    if (mode == 0) return JMAC_EINVAL;
#else
    if (mode == 0) {
        // dQZ / dRR are accumulated with atomics: initialise them (dZ half of dQZ starts at the self term)
        hipLaunchKernelGGL(init_dqz_kernel, dim3((unsigned)((Nsrc * 2 * d + T - 1) / T)), dim3(T), 0, st, dQZ, lddqz, Nsrc, d, G, ldg,
                           out_scale, loop_rel >= 0 ? 1 : 0, self_off, N);
        hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((nrel * 2 * d + T - 1) / T)), dim3(T), 0, st, dRR, nrel, 2 * d, lddrr, 0.f);
        JMAC_DISPATCH_D(D4, nch, hipLaunchKernelGGL((rel_attn_bwd_dst_kernel<NCH, 2, 0, D4T>), dim3(gridA + gcs), dim3(kBlock), 0, st, a,
                                                  (int)gridA, cs));
        FinalizeArgs f{};
        f.sp[0] = sum_task(by_dst, a.part, D4, dP, lddp, 0);
        f.rd[0] = reduce_task(a.da_part, (int)gridA, 1.f, da);
        // the fill wrote zeros into dRz[loop]: overwrite it
        f.rd[1] = reduce_task(colsum_part, (int)gcs, -out_scale, dRR + (int64_t)(loop_rel >= 0 ? loop_rel : 0) * lddrr + d);
        const int nb = f.sp[0].nblocks + f.rd[0].nblocks + f.rd[1].nblocks;
        if (nb > 0) hipLaunchKernelGGL(bwd_finalize_kernel, dim3((unsigned)nb), dim3(kBlock), 0, st, f);
        return (int)hipGetLastError();
    }
#endif
    // ---- deterministic mode: three launches ----------------------------------------------------------------------------
    // 1. pass A by destination (+ the column-sum partials of G)
    if (phases & kPhaseA) {
        JMAC_DISPATCH_D(D4, nch, hipLaunchKernelGGL((rel_attn_bwd_dst_kernel<NCH, 2, 1, D4T>), dim3(gridA + gcs), dim3(kBlock), 0, st, a,
                                                  (int)gridA, cs));
    }
    // 2. pass B by source and pass C by relation, side by side
    BwdArgs b = a;
    b.items = by_src->items; b.splits = by_src->splits; b.counts = by_src->counts; b.order = by_src->order;
    b.part = (float*)(wsb + w.part_src);
    b.sign = 1.f; b.add_self = loop_rel >= 0 ? 1 : 0;
    b.item_edges = reinterpret_cast<const int4*>(by_src->item_edges); b.entry_dst = by_src->entry_dst;
    b.n_items_max = (int32_t)(by_src->n_items_max > 0 ? by_src->n_items_max : 1);
    BwdArgs c = a;
    c.items = by_rel->items; c.splits = by_rel->splits; c.counts = by_rel->counts; c.order = by_rel->order;
    c.part = (float*)(wsb + w.part_rel);
    c.sign = -1.f; c.add_self = 0;
    c.item_edges = reinterpret_cast<const int4*>(by_rel->item_edges); c.entry_dst = by_rel->entry_dst;
    c.n_items_max = (int32_t)(by_rel->n_items_max > 0 ? by_rel->n_items_max : 1);
    const unsigned gB = (phases & kPhaseB) ? persist_grid(by_src->n_items_max) : 0u;
    const unsigned gC = (phases & kPhaseC) ? persist_grid(by_rel->n_items_max) : 0u;
    if (gB + gC > 0) {
        JMAC_DISPATCH_D(D4, nch, hipLaunchKernelGGL((rel_attn_bwd_gather_kernel<NCH, 4, D4T>), dim3(gB + gC), dim3(kBlock), 0, st, b, dQZ,
                                                  lddqz, (int)gB, c, dRR, lddrr));
    }
    // 3. every merge: partial rows of the split destinations / sources / relations, da, dRz[loop]
    FinalizeArgs f{};
    if (phases & kPhaseM) f.sp[0] = sum_task(by_dst, a.part, D4, dP, lddp, 0);
    if (phases & kPhaseB) f.sp[1] = sum_task(by_src, b.part, 2 * D4, dQZ, lddqz, b.add_self);
    if (phases & kPhaseC) f.sp[2] = sum_task(by_rel, c.part, 2 * D4, dRR, lddrr, 0);
    if (phases & kPhaseM) {
        f.rd[0] = reduce_task(a.da_part, (int)gridA, 1.f, da);
        // pass C wrote zeros into dRz[loop] (the loop relation has no edges): overwrite it (phase 8 runs with or after phase 4)
        f.rd[1] = reduce_task(colsum_part, (int)gcs, -out_scale, dRR + (int64_t)(loop_rel >= 0 ? loop_rel : 0) * lddrr + d);
    }
    const int nb = f.sp[0].nblocks + f.sp[1].nblocks + f.sp[2].nblocks + f.rd[0].nblocks + f.rd[1].nblocks;
    if (nb > 0) hipLaunchKernelGGL(bwd_finalize_kernel, dim3((unsigned)nb), dim3(kBlock), 0, st, f);
    return (int)hipGetLastError();
}

extern "C" {

int jmac_rel_attn_aggregate_bwd_f32(const float* P, int64_t ldp, const float* QZ, int64_t ldqz, const float* RR,
                                    int64_t ldrr, const float* a_att, const int32_t* col, const int32_t* etype,
                                    const int32_t* dst_of_slot, const jmac_view_t* by_dst, const jmac_view_t* by_src,
                                    const jmac_view_t* by_rel, int64_t N, int64_t Nsrc, int64_t E, int64_t nrel, int64_t d,
                                    float slope, int32_t loop_rel, int64_t self_off, float out_scale, const float* out, int64_t ldo,
                                    const float* seg_max, const float* seg_den, const float* G, int64_t ldg, float* dP,
                                    int64_t lddp, float* dQZ, int64_t lddqz, float* dRR, int64_t lddrr, float* da,
                                    int32_t mode, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    return rel_attn_bwd_impl(P, ldp, QZ, ldqz, RR, ldrr, a_att, col, etype, dst_of_slot, by_dst, by_src, by_rel, N, Nsrc, E, nrel, d, slope,
                             loop_rel, self_off, out_scale, out, ldo, seg_max, seg_den, G, ldg, dP, lddp, dQZ, lddqz, dRR, lddrr, da, mode,
                             ws, ws_bytes, stream, kPhaseAll);
}

int jmac_rel_attn_aggregate_bwd_phases_f32(const float* P, int64_t ldp, const float* QZ, int64_t ldqz, const float* RR,
                                           int64_t ldrr, const float* a_att, const int32_t* col, const int32_t* etype,
                                           const int32_t* dst_of_slot, const jmac_view_t* by_dst, const jmac_view_t* by_src,
                                           const jmac_view_t* by_rel, int64_t N, int64_t Nsrc, int64_t E, int64_t nrel, int64_t d,
                                           float slope, int32_t loop_rel, int64_t self_off, float out_scale, const float* out,
                                           int64_t ldo, const float* seg_max, const float* seg_den, const float* G, int64_t ldg,
                                           float* dP, int64_t lddp, float* dQZ, int64_t lddqz, float* dRR, int64_t lddrr, float* da,
                                           int32_t phases, void* ws, size_t ws_bytes, jmac_stream_t stream) {
    if (phases <= 0 || phases > kPhaseAll) return JMAC_EINVAL;
    return rel_attn_bwd_impl(P, ldp, QZ, ldqz, RR, ldrr, a_att, col, etype, dst_of_slot, by_dst, by_src, by_rel, N, Nsrc, E, nrel, d, slope,
                             loop_rel, self_off, out_scale, out, ldo, seg_max, seg_den, G, ldg, dP, lddp, dQZ, lddqz, dRR, lddrr, da, 1,
                             ws, ws_bytes, stream, phases);
}

}  // extern "C"
