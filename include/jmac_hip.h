/*
 * jmac_hip.h -- C ABI of libjmac_hip.so: the MI355X (gfx950) implementation of JMAC's relation-aware
 * GNN layer and triple / entity-pair scoring hot path.
 *
 * The reference (vinhsuhi/JMAC) is pure Python/PyTorch and has no FFI of its own; the seams this
 * library replaces are the Python call sites cited on every entry point below (paths relative to the
 * reference tree).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous row-major data unless the name starts with h_;
 *  - the caller owns every buffer; the library never allocates device memory.  Scratch space is
 *    passed in (`ws`, `ws_bytes`); its size comes from the matching *_workspace_bytes() function;
 *  - row tables carry a leading dimension in ELEMENTS (ld*): rows must be 16-byte aligned
 *    (ld % 4 == 0, base 16-B aligned) and d % 4 == 0, d <= 512 (the Python host pads d);
 *  - indices on the device side are int32 (edge / node / relation ids), sizes are int64;
 *    the reference's int64 edge lists are converted by jmac_csr_build;
 *  - every function takes the stream to launch on (pass torch.cuda.current_stream().cuda_stream),
 *    is asynchronous, never synchronises, and is hipGraph-capturable;
 *  - return value: 0 = ok, negative = JMAC_E* argument error, positive = hipError_t;
 *  - no global state; re-entrant; one host thread per device.
 */
#ifndef JMAC_HIP_H_
#define JMAC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* jmac_stream_t; /* hipStream_t */

#define JMAC_OK 0
#define JMAC_EINVAL (-1)    /* bad argument (null pointer, negative size)            */
#define JMAC_EDIM (-2)      /* unsupported d (d % 4 != 0 or d > 512) or ld           */
#define JMAC_EWORKSPACE (-3) /* workspace too small                                   */
#define JMAC_ERANGE (-4)    /* size exceeds int32 indexing                           */

/* A schedule item: one wavefront's unit of work = a run of at most `chunk` consecutive entries of one
 * segment (segment = destination node, source node or relation type).  pslot < 0: the item covers its
 * whole segment and finalises it; pslot >= 0: partial result slot, merged by the combine pass. */
typedef struct { int32_t seg, beg, end, pslot; } jmac_item_t;
/* A segment that was split over several items. */
typedef struct { int32_t seg, pslot0, nchunks, pad; } jmac_split_t;

const char* jmac_strerror(int rc);
int jmac_version(void);

/* ---------------------------------------------------------------------------------------------
 * Graph ingest (replaces: compute_norm's scatter_add degree, src/jmac_model.py:99-109; the implicit
 * COO edge lists of train.py:116-135 and src/utils.py:112-149).
 * --------------------------------------------------------------------------------------------- */

/* Bytes of scratch needed by jmac_csr_build / jmac_group_build for E entries and S segments. */
size_t jmac_graph_workspace_bytes(int64_t E, int64_t S);

/* Range check of an index array (elem_bytes = 4: int32, 8: int64): *bad (device int32, zeroed by the caller) +=
 * number of entries outside [lo, hi).  The reference raises IndexError on such an id (torch indexing); the kernels
 * of this library TRUST their indices, so a host that cannot vouch for them runs this first and reads *bad back
 * (jmac_amd does: once per graph at build time, once per index tensor for the loss / ranking entry points). */
int jmac_index_check(const void* idx, int32_t elem_bytes, int64_t n, int64_t lo, int64_t hi, int32_t* bad,
                     jmac_stream_t stream);

/* COO (edge_index [2,E] int64: row 0 = aggregation destination, row 1 = message source;
 * edge_type [E] int64) -> CSR by destination with a DETERMINISTIC order inside each row:
 *   nrel > 0 (types in [0, nrel)): by relation type, then input order (edges of a row that share a relation are
 *            neighbours: their [Rq|Rz] row is read once per group of gathers);
 *   nrel = 0: input order.
 *   rowptr [N+1], col [E] (source of each CSR slot), etype [E], perm [E] (original edge id).
 * Precondition: destinations in [0,N) (check with jmac_index_check; sources / types are range-checked against the
 * tables by the caller the same way). */
int jmac_csr_build(const int64_t* edge_index, const int64_t* edge_type, int64_t E, int64_t N, int64_t nrel,
                   int32_t* rowptr, int32_t* col, int32_t* etype, int32_t* perm,
                   void* ws, size_t ws_bytes, jmac_stream_t stream);

/* Generic stable grouping of E int32 keys in [0,S): ptr [S+1], order [E] (positions sorted by key).
 * Used for the by-source (CSC) and by-relation views of the CSR slots needed by the backward. */
int jmac_group_build(const int32_t* keys, int64_t E, int64_t S, int32_t* ptr, int32_t* order,
                     void* ws, size_t ws_bytes, jmac_stream_t stream);

/* Upper bounds for the arrays jmac_items_build fills (coop_min as passed to it; 0 = no cooperative splits). */
int64_t jmac_items_max(int64_t S, int64_t E, int32_t chunk, int32_t coop_min);   /* items          */
int64_t jmac_splits_max(int64_t E, int32_t chunk, int32_t coop_min);             /* split segments */
int64_t jmac_parts_max(int64_t E, int32_t chunk, int32_t coop_min);              /* partial slots  */

/* Cut segments (ptr [S+1]) into items (one wavefront's unit of work each).
 *   items [jmac_items_max], splits [jmac_splits_max],
 *   counts [8] = {n_items, n_splits, n_parts, n_empty, n_coop, 0, 0, 0}  (device ints: kernels read the counts, the
 *   host never has to).
 * Segment classes: empty (no entries); plain (<= chunk entries: one item, finalised by its wave); split (> chunk
 * entries: items of <= chunk entries each, merged by the combine pass); and, when coop_max > 0, cooperative
 * (coop_min < len <= coop_max: exactly 4 items of ceil(len/4) entries -- the four wavefronts of one workgroup, whose
 * partial results the forward kernel merges through LDS; for every other consumer they are ordinary split segments
 * with 4 partial slots).  coop_max = 0 disables the class (schedules of the backward's by-source / by-relation views,
 * large graphs).
 * Item order: [cooperative items, 4 per segment][split items][plain segments in segment order][the n_empty empty
 * segments]; the splits array lists the n_coop cooperative segments first (their .pad = 1). */
int jmac_items_build(const int32_t* ptr, int64_t S, int32_t chunk, int32_t coop_min, int32_t coop_max,
                     jmac_item_t* items, jmac_split_t* splits, int32_t* counts, void* ws, size_t ws_bytes,
                     jmac_stream_t stream);

/* item_edges [n_items_max][4] int32 = {col[beg], etype[beg], col[beg+1], etype[beg+1]} per item (-1 where the item is
 * shorter): the first two entries of an item inline with its header, optional input of the kernels on small graphs (one
 * dependent round trip fewer per wavefront).  By-destination schedule: col / etype of the CSR (forward, backward pass A).
 * By-source / by-relation schedules: pass `order` as col and the view's `entry_dst` as etype (backward passes B / C). */
int jmac_item_edges_build(const jmac_item_t* items, const int32_t* counts, int64_t n_items_max,
                          const int32_t* col, const int32_t* etype, int32_t* item_edges,
                          jmac_stream_t stream);

/* One schedule over the CSR slots: the CSR by destination itself (order = NULL) or a regrouping of its slots by source /
 * by relation (jmac_group_build), cut into items by jmac_items_build.  n_*_max: array bounds (the exact counts after a
 * host read of `counts`, or the jmac_*_max upper bounds).  n_empty / n_coop: HOST copies of counts[3] / counts[4] -- exact
 * values (read `counts` back once after jmac_items_build; 0 / 0 for a schedule built with coop_max = 0 whose empty
 * segments need not be packed): they fix the launch geometry of the forward kernel without a device-side wait. */
typedef struct {
    const int32_t* ptr;          /* [S+1]                         */
    const int32_t* order;        /* [E] CSR slot per entry, or NULL for the CSR itself */
    const jmac_item_t* items;
    const jmac_split_t* splits;
    const int32_t* counts;       /* device {n_items,n_splits,n_parts,n_empty,n_coop,0,0,0} */
    int64_t n_items_max, n_splits_max, n_parts_max;
    const int32_t* item_edges;   /* jmac_item_edges_build's array for `items`, or NULL */
    int64_t n_empty, n_coop;
    const int32_t* entry_dst;    /* by-source / by-relation views: destination node of every entry IN GROUPED ORDER
                                  * (= dst_of_slot[order[x]]; saves the backward a dependent load), or NULL */
} jmac_view_t;

/* ---------------------------------------------------------------------------------------------
 * Relation-aware attention aggregation (replaces: MessagePassing.propagate + message + scatter_,
 * modules/helper/message_passing.py:4-29,55-90 and src/jmac_model.py:56-89; the self-loop
 * propagate of src/jmac_model.py:44-45,50; the (nb+self)/2 of :52), in the factorised form
 *     h_e = P[i] + Q[j] - Rq[t]        s_e = a . LeakyReLU(h_e)      alpha = softmax over in(i)
 *     out[i] = out_scale * ( sqrt(deg_i) * sum_e alpha_e (Z[j] - Rz[t])  +  [Z[i] - Rz[loop]] )
 * for edges e = (i <- j, type t).  Tables: P [N,d] (ldp); QZ [N,2d] = Q|Z per node (ldqz);
 * RR [nr+1,2d] = Rq|Rz per relation (ldrr); a_att [d].  loop_rel < 0 drops the bracketed self term.
 * seg_max/seg_den [N] receive the per-destination softmax max / denominator for the backward.
 * Destinations (rows of P / out, N of them) and sources (rows of QZ, Nsrc >= max(col)+1 of them) may be
 * different index spaces (destination-sharded multi-GPU: P holds the rank's rows, QZ the all-gathered
 * table); the fused self term (loop_rel >= 0) reads QZ[self_off + i]: self_off = 0 when the two spaces coincide,
 * = the row of the gathered table that holds the rank's destination 0 when the destinations are a slice of it.
 * by_dst: the schedule over the CSR rows (by_dst->ptr = rowptr [N+1], order = NULL).
 * --------------------------------------------------------------------------------------------- */
size_t jmac_rel_attn_fwd_workspace_bytes(int64_t n_parts_max, int64_t d);

int jmac_rel_attn_aggregate_fwd_f32(
    const float* P, int64_t ldp, const float* QZ, int64_t ldqz, const float* RR, int64_t ldrr,
    const float* a_att, const int32_t* col, const int32_t* etype, const jmac_view_t* by_dst,
    int64_t N, int64_t d, float slope, int32_t loop_rel, int64_t self_off, float out_scale,
    float* out, int64_t ldo, float* seg_max, float* seg_den,
    void* ws, size_t ws_bytes, jmac_stream_t stream);

/* Same op with bf16 TABLES (P, QZ, RR: raw bf16 bits, rows 8-byte aligned: ld % 4 == 0, d % 4 == 0);
 * a_att, the logits, the softmax, the accumulators and every output stay fp32 ("keep
 * P/Q/Z tables bf16 but logits, softmax, accumulators ... in fp32").  Halves the gathered bytes of
 * the HBM-bound kernel; forward only (BASELINE config 3: bf16 union-graph scoring is inference). */
int jmac_rel_attn_aggregate_fwd_bf16(
    const uint16_t* P, int64_t ldp, const uint16_t* QZ, int64_t ldqz, const uint16_t* RR, int64_t ldrr,
    const float* a_att, const int32_t* col, const int32_t* etype, const jmac_view_t* by_dst,
    int64_t N, int64_t d, float slope, int32_t loop_rel, int64_t self_off, float out_scale,
    float* out, int64_t ldo, float* seg_max, float* seg_den,
    void* ws, size_t ws_bytes, jmac_stream_t stream);

/* The same on bf16 tables whose row halves are PADDED to `dh` elements (dh >= d, dh % 8 == 0: 16-byte aligned halves, e.g.
 * d = 300 -> dh = 304): P rows hold dh elements, a [Q|Z] / [Rq|Rz] row is Q at 0 and Z at dh (2 dh elements), pad columns
 * are ZERO (the host pads the projection weights, so the GEMM writes them).  With (d, dh) = (300, 304) or (256, 256) and
 * 16-byte aligned rows the persistent-grid form of the kernel (large graphs) takes 16 bytes per lane and two edges per wave
 * instruction (half a wave per edge) -- the 8-byte lane loads of the unpadded d = 300 layout run at 0.54-0.70 x the 16-byte
 * rate; every other case reads the same layout with the 64-lane map (dh % 4 == 0).  out stays [N, d] fp32. */
int jmac_rel_attn_aggregate_fwd_bf16_padded(const uint16_t* P, int64_t ldp, const uint16_t* QZ, int64_t ldqz,
                                            const uint16_t* RR, int64_t ldrr, int64_t dh, const float* a_att,
                                            const int32_t* col, const int32_t* etype, const jmac_view_t* by_dst,
                                            int64_t N, int64_t d, float slope, int32_t loop_rel, int64_t self_off,
                                            float out_scale, float* out, int64_t ldo, float* seg_max, float* seg_den,
                                            void* ws, size_t ws_bytes, jmac_stream_t stream);

/* Up to TWO independent calls of jmac_rel_attn_aggregate_fwd_f32 as ONE launch (round 5): the first two layers of
 * JMAC.forward_name (conv1_alignment, conv1_completion: src/jmac_model.py:183,190) read different tables / weights / loop rows but
 * the same graph and do not depend on each other; at DBP-5L size each launch is a chain of dependent round trips with most of the
 * chip idle.  A job = the arguments of the single call.  Jobs whose form is the one-wave-per-item kernel (small graphs, fp32)
 * share a grid (blocks [0, g0) job 0, the rest job 1); any other pair runs as two launches.  Results: those of the single calls,
 * bit for bit. */
typedef struct {
    const float *P; int64_t ldp; const float *QZ; int64_t ldqz; const float *RR; int64_t ldrr; const float *a_att;
    const int32_t *col, *etype; const jmac_view_t *by_dst; int64_t N, d; float slope; int32_t loop_rel; int64_t self_off;
    float out_scale; float *out; int64_t ldo; float *seg_max, *seg_den; void *ws; size_t ws_bytes;
} jmac_agg_fwd_job_t;
int jmac_rel_attn_aggregate_fwd_jobs_f32(const jmac_agg_fwd_job_t* jobs, int32_t n_jobs, jmac_stream_t stream);

/* Merge of per-source-chunk partial aggregations: the slab-pipelined exchange of the destination-sharded layer
 * (jmac_amd/dist.py; the reference has one process and no exchange: modules/helper/message_passing.py:24,28 are the
 * per-destination softmax / sum that make the partials mergeable).  Part c = the forward above on the edges whose SOURCE is
 * in chunk c (loop_rel < 0, out_scale 1): out_c [N,d], its seg_max_c / seg_den_c [N], and the chunk graph's rowptr_c [N+1]
 * (in-degree within the chunk).  Writes out [N,d] = out_scale * (nb + self), nb = sqrt(deg) * sum_e softmax(e) x_e over ALL
 * edges, self = Zself[i] - rz_loop (the fused self loop, src/jmac_model.py:49-50; Zself NULL: none), and the whole graph's
 * seg_max / seg_den (what jmac_rel_attn_aggregate_bwd_f32 on the whole graph expects).  h_* are HOST arrays of n_parts
 * device pointers (n_parts <= JMAC_MERGE_MAX_PARTS); parts are combined in index order (bitwise reproducible). */
#define JMAC_MERGE_MAX_PARTS 16
int jmac_softmax_parts_merge_f32(const float* const* h_out, int64_t ldo, const float* const* h_seg_max,
                                 const float* const* h_seg_den, const int32_t* const* h_rowptr, int32_t n_parts,
                                 int64_t N, int64_t d, const float* Zself, int64_t ldz, const float* rz_loop,
                                 float out_scale, float* out, int64_t ldo2, float* seg_max, float* seg_den,
                                 jmac_stream_t stream);

/* Backward of the op above (replaces autograd through the same reference lines).
 *   G [N,d] (ldg) = dL/dout;  outputs: dP [N,d] (lddp), dQZ [N,2d] (lddqz), dRR [nr+1,2d] (lddrr),
 *   da [d].  All outputs are fully written (no pre-zeroing needed).
 * mode 0 ("atomic"): one pass by destination, float atomics into dQZ / dRR.
 * mode 1 ("deterministic", default): pass A by destination writes per-edge records, pass B by source
 *   and pass C by relation sum them with plain stores: bitwise reproducible, no atomics.
 *   Needs the by-source view (sptr [N+1], sorder [E], sitems ...) and the by-relation view
 *   (tptr [nrel+1], torder [E], titems ...) of the CSR slots, built with jmac_group_build +
 *   jmac_items_build. dst_of_slot [E] = destination node of each CSR slot. */
size_t jmac_rel_attn_bwd_workspace_bytes(int64_t N, int64_t E, int64_t nrel, int64_t d,
                                         int64_t n_parts_max_dst, int64_t n_parts_max_src,
                                         int64_t n_parts_max_rel, int32_t mode);

int jmac_rel_attn_aggregate_bwd_f32(
    const float* P, int64_t ldp, const float* QZ, int64_t ldqz, const float* RR, int64_t ldrr,
    const float* a_att, const int32_t* col, const int32_t* etype, const int32_t* dst_of_slot,
    const jmac_view_t* by_dst, const jmac_view_t* by_src, const jmac_view_t* by_rel,
    int64_t N, int64_t Nsrc, int64_t E, int64_t nrel, int64_t d, float slope, int32_t loop_rel,
    int64_t self_off, float out_scale, const float* out, int64_t ldo, const float* seg_max, const float* seg_den,
    const float* G, int64_t ldg,
    float* dP, int64_t lddp, float* dQZ, int64_t lddqz, float* dRR, int64_t lddrr, float* da,
    int32_t mode, void* ws, size_t ws_bytes, jmac_stream_t stream);

/* The deterministic backward above in separately callable PHASES (bit mask; 15 = the whole backward = the call above, mode 1):
 *   1 = pass A by destination (dP of plain destinations, the per-edge records in ws, da / column-sum partials)
 *   2 = pass B by source (d[Q|Z], + the merge of its split sources)       4 = pass C by relation (d[Rq|Rz], + its merge)
 *   8 = the merges that hang on pass A alone: dP of split destinations, da, dRz[loop] (run with or after phase 4)
 * The same ws (sized for the whole graph's views by jmac_rel_attn_bwd_workspace_bytes) must be passed to every call of one
 * backward: the records pass A leaves in it are read by passes B and C.
 * A phase-2 call may cover a SLAB of the source rows, so that a destination-sharded layer can reduce-scatter slab c of d[Q|Z]
 * over xGMI while pass B works on slab c+1 (jmac_amd.dist; adjoint of the all-gather in front of
 * modules/helper/message_passing.py:24,28): by_src = a view built by jmac_items_build on the slab's slice of the by-source
 * segment pointer (segments numbered from the slab's first row; `order` / `entry_dst` the whole graph's), dQZ = the slab's first
 * row, Nsrc = its rows, self_off = (whole-table self_off) - (slab's first row): negative or past the slab where the rank's own
 * rows lie elsewhere. */
int jmac_rel_attn_aggregate_bwd_phases_f32(
    const float* P, int64_t ldp, const float* QZ, int64_t ldqz, const float* RR, int64_t ldrr,
    const float* a_att, const int32_t* col, const int32_t* etype, const int32_t* dst_of_slot,
    const jmac_view_t* by_dst, const jmac_view_t* by_src, const jmac_view_t* by_rel,
    int64_t N, int64_t Nsrc, int64_t E, int64_t nrel, int64_t d, float slope, int32_t loop_rel,
    int64_t self_off, float out_scale, const float* out, int64_t ldo, const float* seg_max, const float* seg_den,
    const float* G, int64_t ldg,
    float* dP, int64_t lddp, float* dQZ, int64_t lddqz, float* dRR, int64_t lddrr, float* da,
    int32_t phases, void* ws, size_t ws_bytes, jmac_stream_t stream);

/* BatchNorm1d (batch statistics or running statistics) + tanh on [N,d]
 * (replaces: self.layer_act(self.bn(.)), src/jmac_model.py:52).
 * training != 0: mean/var are computed over the N rows (biased var), written to save_mean /
 * save_invstd [d], and running_mean/var (may be NULL) are updated with `momentum` (unbiased var). */
size_t jmac_bn_tanh_workspace_bytes(int64_t N, int64_t d);
int jmac_bn_tanh_fwd_f32(const float* x, int64_t ldx, int64_t N, int64_t d, const float* weight,
                         const float* bias, float* running_mean, float* running_var,
                         int32_t training, float momentum, float eps, float* y, int64_t ldy,
                         float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                         jmac_stream_t stream);
int jmac_bn_tanh_bwd_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy,
                         int64_t ldgy, int64_t N, int64_t d, const float* weight,
                         const float* save_mean, const float* save_invstd, int32_t training,
                         float* gx, int64_t ldgx, float* gweight, float* gbias, void* ws,
                         size_t ws_bytes, jmac_stream_t stream);

/* The same with a second destination / a second gradient source: the layer's output is an operand of two concatenations in
 * JMAC.forward_name (src/jmac_model.py:192 cat(comp_l1, align_l1) and :203 cat(align_layers)): the forward writes the rows
 * into both cat buffers (y2 / ldy2, may be NULL), the backward sums the two incoming gradients (gy2 / ldgy2, may be NULL)
 * while it reads them -- no cat copy, no gradient add. */
int jmac_bn_tanh_fwd2_f32(const float* x, int64_t ldx, int64_t N, int64_t d, const float* weight,
                          const float* bias, float* running_mean, float* running_var,
                          int32_t training, float momentum, float eps, float* y, int64_t ldy,
                          float* y2, int64_t ldy2, float* save_mean, float* save_invstd, void* ws,
                          size_t ws_bytes, jmac_stream_t stream);
int jmac_bn_tanh_bwd2_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy,
                          int64_t ldgy, const float* gy2, int64_t ldgy2, int64_t N, int64_t d,
                          const float* weight, const float* save_mean, const float* save_invstd,
                          int32_t training, float* gx, int64_t ldgx, float* gweight, float* gbias,
                          void* ws, size_t ws_bytes, jmac_stream_t stream);

/* Segmented form (train mode): the N rows are a STACK of `nblocks` row blocks, h_blk_ptr [nblocks+1] (HOST array, h_blk_ptr[0] = 0)
 * -- the KGs of one launch set.  The reference's training step encodes two KGs per batch with one set of layer weights
 * (src/jmac_model.py:325-326, 263-264: forward_base(graph1), forward_base(graph2)); each of those calls normalises with the
 * batch statistics of ITS rows and moves the layer's running estimates once.  Here every block gets its own statistics
 * (save_mean / save_invstd [nblocks, d]); weight / bias / running_mean / running_var are the layer's; the running estimates
 * are updated block after block in the order h_order [nblocks] (HOST; NULL = 0,1,2,...; a permutation: the reference's
 * call order), i.e. r <- (1-m)((1-m) r + m b_first) + m b_second ...; the caller bumps num_batches_tracked by nblocks.
 * The backward sums grad weight / grad bias over the blocks (shared parameters) and applies every block's own sums and
 * row count.  nblocks <= JMAC_BN_MAX_BLOCKS; blocks must be non-empty. */
#define JMAC_BN_MAX_BLOCKS 16
size_t jmac_bn_tanh_seg_workspace_bytes(int32_t nblocks, int64_t d);
int jmac_bn_tanh_seg_fwd2_f32(const float* x, int64_t ldx, int64_t d, int32_t nblocks, const int64_t* h_blk_ptr,
                              const int32_t* h_order, const float* weight, const float* bias,
                              float* running_mean, float* running_var, float momentum, float eps, float* y,
                              int64_t ldy, float* y2, int64_t ldy2, float* save_mean, float* save_invstd,
                              void* ws, size_t ws_bytes, jmac_stream_t stream);
int jmac_bn_tanh_seg_bwd2_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy,
                              int64_t ldgy, const float* gy2, int64_t ldgy2, int64_t d, int32_t nblocks,
                              const int64_t* h_blk_ptr, const float* weight, const float* save_mean,
                              const float* save_invstd, float* gx, int64_t ldgx, float* gweight, float* gbias,
                              void* ws, size_t ws_bytes, jmac_stream_t stream);

/* Phased forms of the same BatchNorm + tanh for batch statistics that span several ranks (destination-sharded
 * layer): the caller combines the per-rank moments / sums between the phases (torch.distributed over RCCL).
 *   forward : jmac_col_moments_f32 -> mean[d], m2[d] = sum (x - mean)^2 of THIS rank's N rows
 *             [all-gather, Chan's parallel combination -> global mean, invstd]
 *             jmac_bn_tanh_apply_f32
 *   backward: jmac_bn_tanh_bwd_sums_f32 -> sums[0:d] = sum gz, sums[d:2d] = sum gz*xhat over this rank's rows
 *             (gz = gy (1 - y^2); these are also the rank's contributions to grad bias / grad weight)
 *             [all-reduce]   jmac_bn_tanh_bwd_apply_f32 with the global sums and the global row count n_total. */
int jmac_col_moments_f32(const float* x, int64_t ldx, int64_t N, int64_t d, float* mean, float* m2,
                         void* ws, size_t ws_bytes, jmac_stream_t stream);
int jmac_bn_tanh_apply_f32(const float* x, int64_t ldx, int64_t N, int64_t d, const float* weight,
                           const float* bias, const float* mean, const float* invstd, float* y,
                           int64_t ldy, jmac_stream_t stream);
int jmac_bn_tanh_bwd_sums_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy,
                              int64_t ldgy, int64_t N, int64_t d, const float* mean,
                              const float* invstd, float* sums, void* ws, size_t ws_bytes,
                              jmac_stream_t stream);
int jmac_bn_tanh_bwd_apply_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy,
                               int64_t ldgy, int64_t N, int64_t d, const float* weight,
                               const float* mean, const float* invstd, const float* sums,
                               int64_t n_total, float* gx, int64_t ldgx, jmac_stream_t stream);

/* Row L2 normalisation  y = x / max(||x||_2, eps)  (replaces F.normalize(x, 2, -1) of JMAC.forward_name and get_emb,
 * src/jmac_model.py:179,191,227-228; eps = 1e-12 is torch's default).  inv [N] receives 1 / max(||x_r||, eps) for the
 * backward:  gx = inv (g - y (g.y)),  or inv g on rows whose norm was clamped.  Any d and leading dimensions. */
int jmac_row_normalize_fwd_f32(const float* x, int64_t ldx, int64_t N, int64_t d, float eps, float* y,
                               int64_t ldy, float* inv, jmac_stream_t stream);
int jmac_row_normalize_bwd_f32(const float* y, int64_t ldy, const float* g, int64_t ldg, const float* inv,
                               int64_t N, int64_t d, float eps, float* gx, int64_t ldgx,
                               jmac_stream_t stream);

/* completion_dropout(F.normalize(x)) in one pass each way (src/jmac_model.py:179,191).  mask [N,d] holds the caller's
 * {0,1} Bernoulli draws (NULL: no dropout), scale = 1/(1-p); y may be a strided slice of a cat buffer.  The backward keeps
 * only inv [N] from the forward and recomputes the normalised row from x:  gx (+)= d/dx of the forward applied to g
 * (accumulate != 0 adds into gx).  d % 4 == 0, leading dimensions % 4 == 0, 16-byte aligned bases. */
int jmac_row_normalize_drop_fwd_f32(const float* x, int64_t ldx, int64_t N, int64_t d, float eps,
                                    const float* mask, int64_t ldm, float scale, float* y, int64_t ldy,
                                    float* inv, jmac_stream_t stream);
int jmac_row_normalize_drop_bwd_f32(const float* x, int64_t ldx, const float* inv, const float* mask,
                                    int64_t ldm, float scale, const float* g, int64_t ldg, int64_t N,
                                    int64_t d, float eps, float* gx, int64_t ldgx, int32_t accumulate,
                                    jmac_stream_t stream);

/* The same with the Bernoulli draws made INSIDE the kernels (torch's F.dropout draws them with its own Philox stream: the
 * draws are equally i.i.d. Bernoulli(1 - p_drop) but not the same bits).  seed: ONE device-resident int64 the caller fills from
 * its generator (a device value, so that a captured step draws a fresh mask on every replay); element (r, c) is kept iff word
 * c % 4 of Philox4x32-10(key = seed, counter = r * d/4 + c/4) < (1 - p_drop) * 2^32, kept elements are scaled by 1/(1 - p_drop).
 * The backward regenerates the draws from the same seed: no mask tensor exists.  0 <= p_drop < 1. */
int jmac_row_normalize_dropseed_fwd_f32(const float* x, int64_t ldx, int64_t N, int64_t d, float eps,
                                        const int64_t* seed, float p_drop, float* y, int64_t ldy,
                                        float* inv, jmac_stream_t stream);
int jmac_row_normalize_dropseed_bwd_f32(const float* x, int64_t ldx, const float* inv, const int64_t* seed,
                                        float p_drop, const float* g, int64_t ldg, int64_t N, int64_t d,
                                        float eps, float* gx, int64_t ldgx, int32_t accumulate,
                                        jmac_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Completion scoring (replaces: torch.cdist(er, all_kg_emb, p=1), src/jmac_model.py:312; the
 * filter/sort/np.where ranking loop of src/validate.py:50-64).
 * --------------------------------------------------------------------------------------------- */
/* out[b,n] (+)= sum_k |er[b,k] - table[n,k]|;  accumulate != 0 adds to out (layer loop of :302). */
int jmac_l1_score_f32(const float* er, int64_t lder, const float* table, int64_t ldt, int64_t B,
                      int64_t N, int64_t d, float* out, int64_t ldout, int32_t accumulate,
                      jmac_stream_t stream);
/* Same with bf16 operands (raw bits; rows 8-byte aligned), fp32 accumulation and output
 * (BASELINE config 3: bf16 entity tables scored against all entities of the union graph). */
int jmac_l1_score_bf16(const uint16_t* er, int64_t lder, const uint16_t* table, int64_t ldt, int64_t B,
                       int64_t N, int64_t d, float* out, int64_t ldout, int32_t accumulate,
                       jmac_stream_t stream);

/* Fused link prediction (replaces: forward_linkpred's layer loop src/jmac_model.py:302-313 TOGETHER WITH the ranking loop of
 * src/validate.py:50-64, for callers that want the ranks and not the [B,N] matrix -- CompletionEvaluator.test):
 *     dist[b,n] = sum over layers l of || (ent_l[h[b]] +/- rel_l[r[b]]) - table_l[n] ||_1      (pred_head != 0: minus)
 *     rank[b]   = 1 + #{n in [0,N): dist[b,n] before dist[b,gold[b]]} - #{filtered n != gold[b]: dist[b,n] before ...}
 * with "before" = smaller, or equal and lower index, exactly as jmac_filtered_rank_f32.  The distance is ONE running fp32 sum
 * over (layer, k) -- the materialised path rounds once more per layer (out += layer), so the two agree to fp32 rounding and
 * give the same rank wherever the gold is not tied with a neighbour at that level.  ent / rel: fp32 tables the query rows are
 * gathered from; table: the N candidate rows, fp32 (== ent) for _f32, raw bf16 for _bf16 (the query rows are then rounded to
 * bf16 as well: BASELINE config 3).  n_layers <= 4, n_layers * d <= ~15 000 (the prep kernel stages candidate rows through
 * 60 KB of LDS; JMAC_ERANGE beyond), d <= 512.  The [B,N] matrix is never written. */
typedef struct {
    const float* ent;  int64_t ld_ent;      /* [*, d] */
    const float* rel;  int64_t ld_rel;      /* [*, d] */
    const void* table; int64_t ld_table;    /* [N, d] fp32 or bf16 */
} jmac_link_layer_t;
size_t jmac_linkpred_rank_workspace_bytes(int64_t B, int64_t d, int32_t n_layers);
int jmac_linkpred_rank_f32(const jmac_link_layer_t* layers, int32_t n_layers, const int32_t* h, const int32_t* r,
                           int32_t pred_head, const int32_t* gold, const int32_t* filt_ptr, const int32_t* filt_idx,
                           int64_t B, int64_t N, int64_t d, int32_t* rank, void* ws, size_t ws_bytes, jmac_stream_t stream);
int jmac_linkpred_rank_bf16(const jmac_link_layer_t* layers, int32_t n_layers, const int32_t* h, const int32_t* r,
                            int32_t pred_head, const int32_t* gold, const int32_t* filt_ptr, const int32_t* filt_idx,
                            int64_t B, int64_t N, int64_t d, int32_t* rank, void* ws, size_t ws_bytes, jmac_stream_t stream);

/* rank[b] = 1 + #{n : score[b,n] < score[b,gold] or (== and n < gold[b])}, where entries listed in
 * the filter CSR (filt_ptr [B+1], filt_idx) other than the gold are skipped.  descending == 0: `score` is a
 * DISTANCE (the reference's predictions = -dist, sorted descending); descending != 0: `score` is a SIMILARITY
 * and `<` reads `>` (alignment ranks, modules/finding/alignment.py:87-112).  filt_ptr may be NULL (raw ranking). */
int jmac_filtered_rank_f32(const float* score, int64_t lds, const int32_t* gold,
                           const int32_t* filt_ptr, const int32_t* filt_idx, int64_t B, int64_t N,
                           int32_t descending, int32_t* rank, jmac_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Alignment scoring (replaces: get_neg's mm + topk, modules/utils/util.py:52-53; the mm / softmax /
 * entropy of compute_alignment_quality, train.py:239-248).
 * --------------------------------------------------------------------------------------------- */
/* C[m,n] = sum_k A[m,k] * B[n,k]  (fp32-in / fp32-accumulate MFMA: exact fp32 products). */
int jmac_sim_matrix_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M,
                        int64_t N, int64_t d, float* C, int64_t ldc, jmac_stream_t stream);

size_t jmac_sim_topk_workspace_bytes(int64_t L, int64_t N, int32_t k);
/* Per row of A: the k largest similarities against all rows of B, descending, ties -> lower index
 * first.  val [L,k] (may be NULL), idx [L,k] int32.
 * N >= 8192 and k <= 64: the L x N score matrix is never written (running top-k in the product's epilogue): a column sample
 * gives every row a threshold (the k-th largest of its first ~N/8 scores, a lower bound of its final k-th score), the full
 * product appends the scores that reach it to per-row candidate lists, a last pass takes the k best of each list; a row whose
 * list overflows (mass ties) recomputes its scores with the product's own instruction sequence.  Same scores bit for bit,
 * hence the same indices as the two-step form; workspace O(L * N / 8) instead of L * N * 4. */
int jmac_sim_topk_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t L, int64_t N,
                      int64_t d, int32_t k, float* val, int32_t* idx, void* ws, size_t ws_bytes,
                      jmac_stream_t stream);

/* Row top-k of an existing score matrix (same ordering rule). */
int jmac_row_topk_f32(const float* S, int64_t lds, int64_t L, int64_t N, int32_t k, float* val,
                      int32_t* idx, jmac_stream_t stream);

size_t jmac_softmax_entropy_workspace_bytes(int64_t n1, int64_t n2);
/* S = A B^T (n1 x n2); ent_rows[i] = H(softmax_j(scale*S[i,:])), ent_cols[j] = H(softmax_i(scale*S[:,j])). */
int jmac_softmax_entropy_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t n1,
                             int64_t n2, int64_t d, float scale, float* ent_rows, float* ent_cols,
                             void* ws, size_t ws_bytes, jmac_stream_t stream);

/* Row softmax of scale*S with rows in `row_mask` and columns in `col_mask` (uint8, 1 = keep) left as
 * is and every other entry replaced by `fill` first (train.py:252-257: fill = -1). In place allowed. */
int jmac_masked_row_softmax_f32(const float* S, int64_t lds, int64_t n1, int64_t n2,
                                const uint8_t* row_mask, const uint8_t* col_mask, float fill,
                                float scale, float* out, int64_t ldo, jmac_stream_t stream);

/* Generalisations of the call above on an existing score matrix S [n1,n2] (row-major): the entries kept by the masks
 * are scaled, every other entry is `fill` * scale (NULL mask = keep all), then
 *   jmac_row_softmax_f32: softmax over each ROW    -> out [n1,n2] (may be NULL), ent [n1] = entropy of the row (may be NULL)
 *   jmac_col_softmax_f32: softmax over each COLUMN -> out_t [n2,n1] = the TRANSPOSED probabilities, i.e.
 *                         softmax(scale * S^T, dim=1) (may be NULL), ent [n2] (may be NULL)
 * so that both orientations of compute_alignment_quality (train.py:241-257; DBPv1 trainer/jmac_trainer.py:281-300:
 * softmax(simi) and softmax(simi.t())) come from ONE similarity GEMM. */
int jmac_row_softmax_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, const uint8_t* row_mask,
                         const uint8_t* col_mask, float fill, float scale, float* out, int64_t ldo, float* ent,
                         jmac_stream_t stream);
size_t jmac_col_softmax_workspace_bytes(int64_t n1, int64_t n2);
int jmac_col_softmax_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, const uint8_t* row_mask,
                         const uint8_t* col_mask, float fill, float scale, float* out_t, int64_t ldo, float* ent,
                         void* ws, size_t ws_bytes, jmac_stream_t stream);

/* The k largest values of every COLUMN of S [n1,n2], descending (val [n2,k]; k <= 16) -- the column term of CSLS
 * (calculate_nearest_k on sim_mat.T, modules/finding/similarity.py:66-67,81-84) without materialising the transpose. */
size_t jmac_col_topk_workspace_bytes(int64_t n1, int64_t n2, int32_t k);
int jmac_col_topk_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, int32_t k, float* val, void* ws,
                      size_t ws_bytes, jmac_stream_t stream);

/* CSLS rescoring (replaces csls_sim, modules/finding/similarity.py:58-78):
 * out[i,j] = 2*S[i,j] - r1[i] - r2[j], r1/r2 = mean of the k largest entries of row i / column j. */
int jmac_csls_apply_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, const float* r1,
                        const float* r2, float* out, int64_t ldo, jmac_stream_t stream);

/* rank[i] = 1 + #{j : c(i,j) > c(i,gold[i]) or (== and j < gold[i])} with c(i,j) = 2*S[i,j] - r1[i] - r2[j]:
 * jmac_csls_apply_f32 followed by the descending rank count, fused (the rescored matrix is neither written nor
 * re-read; same arithmetic, identical ranks).  Alignment evaluation: alignment.py:87-112 on similarity.py:58-78. */
int jmac_csls_rank_f32(const float* S, int64_t lds, int64_t n1, int64_t n2, const float* r1, const float* r2,
                       const int32_t* gold, int32_t* rank, jmac_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Small fp32 GEMM for the relation-side projections (replaces the torch.mm calls on the ~10^3-row
 * relation tables: src/jmac_model.py:40-42 rel_transform_weight1/2, :195-196 relation MLPs, and the
 * hoisted rel'' @ [Wb|Wg] of the factorised layer) and their backward forms.
 *   C[M,N] = op(A) op(B), row-major; transX != 0: op(X) = X^T, i.e. A is stored [K,M] / B is stored [N,K].
 * Exact fp32 products on the fp32-input MFMA, fixed summation order (bitwise reproducible).  Any M, N, K, ld.
 * --------------------------------------------------------------------------------------------- */
int jmac_gemm_f32(const float* A, int64_t lda, int32_t transA, const float* B, int64_t ldb,
                  int32_t transB, int64_t M, int64_t N, int64_t K, float* C, int64_t ldc,
                  jmac_stream_t stream);

/* Grouped form: up to JMAC_GEMM_MAX_TASKS independent small products in ONE launch (the relation side of a training step is
 * ~40 such products, each launch-bound on its own: products of one dependency level go out together).  `tasks` is a HOST
 * array; it is copied into the kernel arguments, so the call is safe under stream capture.  Per task:
 *   C[M,N] (+)= epilogue( op(A) op(B) ),  op as in jmac_gemm_f32;
 *   a_split > 0: the MEMORY rows >= a_split of A are read from A2 (row index - a_split) -- cat(rel_emb, loop_rel) of
 *                src/jmac_model.py:39 without the copy; c_split > 0: output rows >= c_split are written to C2 (its adjoint);
 *   act: JMAC_GEMM_ACT_LEAKY / _RELU apply the activation to the product (src/jmac_model.py:41; DBPv1 :51: ReLU);
 *        JMAC_GEMM_DACT_LEAKY / _RELU multiply the product by act'(z) where act_src holds act(z) (same shape as C; the
 *        sign of act(z) is the sign of z for slope > 0) -- the activation's backward fused into the product that
 *        produces its incoming gradient;
 *   accumulate != 0: C += (applied after the epilogue); rows that go to C2 are always stored, never accumulated.
 * Exact fp32 products (fp32-input MFMA), fixed summation order per element (bitwise reproducible). */
#define JMAC_GEMM_MAX_TASKS 24
#define JMAC_GEMM_ACT_NONE 0
#define JMAC_GEMM_ACT_LEAKY 1
#define JMAC_GEMM_ACT_RELU 2
#define JMAC_GEMM_DACT_LEAKY 3
#define JMAC_GEMM_DACT_RELU 4
typedef struct {
    const float* A;  const float* A2;  int64_t lda;  int64_t a_split;  int32_t transA;  int32_t transB;
    const float* B;  int64_t ldb;
    float* C;  float* C2;  int64_t ldc;  int64_t c_split;
    int64_t M, N, K;
    const float* act_src;  int64_t ld_act_src;
    int32_t act;  int32_t accumulate;  float slope;  int32_t pad_;
} jmac_gemm_task_t;
int jmac_gemm_grouped_f32(const jmac_gemm_task_t* tasks, int32_t n_tasks, jmac_stream_t stream);

/* [Wt | Wb | Wg] [d, 3d] of up to JMAC_WCAT_MAX layers in one launch (w_att = [Wt; Wb] [2d, d] stacked by rows,
 * src/jmac_model.py:24,75-76; gcn_weight [d, d]) -- the operand of the hoisted node projection X [Wt|Wb|Wg] -- and its adjoint:
 * d w_att [2d, d] and d gcn_weight [d, d] cut out of d[Wt|Wb|Wg].  The pointer arrays are HOST arrays of device pointers
 * (copied into the kernel arguments: capture-safe); all matrices contiguous; d % 4 == 0.
 * extra_src / extra_dst / extra_floats: one more contiguous copy in the same launch (the encoder's U11 block of the folded
 * name projection, src/jmac_model.py:177,180, and its adjoint), 16-byte aligned, a multiple of 4 floats; 0 floats: none.
 * counters (pack only): up to JMAC_WCAT_MAX device int64 incremented by one in the same launch -- nn.BatchNorm1d's
 * num_batches_tracked of the layers a training-mode encoder call is about to run (src/jmac_model.py:52). */
#define JMAC_WCAT_MAX 4
int jmac_wcat_pack_f32(const float* const* w_att, const float* const* gcn, float* const* wcat,
                       int32_t n_layers, int64_t d, const float* extra_src, float* extra_dst,
                       int64_t extra_floats, int64_t* const* counters, int32_t n_counters,
                       jmac_stream_t stream);
/* The same with the step's DROPOUT SEEDS riding along (round 5): seed_state [2] = persistent device int64 words, advanced by one
 * per launch; seed_out [2] receives the advanced values -- the seeds jmac_row_normalize_dropseed_{fwd,bwd}_f32 draw from in this
 * step (completion_dropout, src/jmac_model.py:179,191).  No torch RNG op per step: a captured step that uses none is replayed
 * without the generator-state fills torch puts in front of every replay of a graph that does.  Both NULL: as above. */
int jmac_wcat_pack_seed_f32(const float* const* w_att, const float* const* gcn, float* const* wcat, int32_t n_layers,
                            int64_t d, const float* extra_src, float* extra_dst, int64_t extra_floats,
                            int64_t* const* counters, int32_t n_counters, int64_t* seed_state, int64_t* seed_out,
                            jmac_stream_t stream);
int jmac_wcat_unpack_f32(const float* const* dwcat, float* const* d_watt, float* const* d_gcn,
                         int32_t n_layers, int64_t d, const float* extra_src, float* extra_dst,
                         int64_t extra_floats, jmac_stream_t stream);

/* Adam over every parameter tensor of the model in ONE launch (round 5).  train.py:406-407 build torch.optim.Adam(model.parameters(),
 * lr) and :358-359 step it once per batch; torch's fused multi-tensor form gives each 65 536-element chunk to one workgroup -- 125
 * workgroups for the 6.3 M parameters of the DBP-5L model, under half of the 256 CUs.  Same arithmetic as torch.optim.Adam (defaults:
 * betas 0.9 / 0.999, eps 1e-8, weight_decay 0; decoupled != 0 = AdamW's decay; amsgrad is not offered), bias corrections in double:
 *   g' = (maximize ? -g : g) (+ wd p);  m = m + (g' - m)(1 - b1);  v = b2 v + (1 - b2) g'^2
 *   p = p - lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),      t = step[0] + 1
 * tasks: HOST array (copied into the kernel arguments: capture-safe; more than JMAC_ADAM_MAX_TASKS: several launches); p / g / m / v
 * contiguous fp32 of n elements (`vec4` is set by the library).  step [1] device float = completed steps, advanced by one by the
 * call.  aux [3] device double, 8-byte aligned = { beta1^step, beta2^step, 0 }: the running powers the bias corrections are made
 * from (advanced by the call; the caller initialises them for the count it starts from -- {1, 1, 0} for a fresh optimizer -- and
 * again if it changes the betas) and a word that is zero at rest (the last workgroup to finish advances step and powers: every
 * workgroup has read them by then). */
#define JMAC_ADAM_MAX_TASKS 64
typedef struct {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    int32_t vec4;
} jmac_adam_task_t;
int jmac_adam_step_f32(const jmac_adam_task_t* tasks, int32_t n_tasks, float* step, double* aux, double lr, double beta1,
                       double beta2, double eps, double weight_decay, int32_t decoupled, int32_t maximize, jmac_stream_t stream);

/* Used-relation compaction.  A DBP-5L KG names 153-833 of its 961 relation rows in its edges (ja: 158), and the layer's
 * relation transform + projection (src/jmac_model.py:39-42 and the hoisted R''[Wb|Wg]) matter for named rows only: the encoder
 * runs those products on the COMPACT table of used rows (+ the loop row), with edge types renumbered accordingly.
 *   compact: dst[t][p, :] = src[t][idx[p], :]            p < n_used            (idx: device int64 [n_used], ascending)
 *   expand : dst[t][r, :] = src[t][pos[r], :] or 0       r < rows              (pos: device int32 [rows], -1 = unused row;
 *            h_accumulate[t] != 0: dst[t][r, :] += src[t][pos[r], :] on used rows, other rows untouched)
 * Up to 4 tables per launch (host arrays of device pointers); compact tables are dense [., d]; d % 4 == 0, 16-byte rows. */
int jmac_rows_compact_f32(const float* const* h_src, const int64_t* h_ld_src, float* const* h_dst, int32_t n_tables,
                          const int64_t* idx, int64_t n_used, int64_t d, jmac_stream_t stream);
int jmac_rows_expand_f32(const float* const* h_src, float* const* h_dst, const int64_t* h_ld_dst,
                         const int32_t* h_accumulate, int32_t n_tables, const int32_t* pos, int64_t rows, int64_t d,
                         jmac_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Loss gathers.  Indices are the reference's int64 tensors (batch_h /
 * batch_r / batch_t, links, neg_left ...), values in range; rows may have any d (16-byte aligned rows
 * with d % 4 == 0 take the vector path).
 * --------------------------------------------------------------------------------------------- */

/* score[x] = || ent[h[x]] + rel[r[x]] - ent[t[x]] ||_1   (replaces the three gathers, the sum and
 * torch.norm(score, 1, -1) of completion_loss, src/jmac_model.py:345-350).  d <= 512.
 * period: a HINT that triples x, x+period, x+2*period ... share (h, r) -- the reference's batch is
 * sub.repeat(K+1), rel.repeat(K+1), cat(obj, negatives) with period = batch size (train.py:347-352) --
 * so that one wavefront reads E[h]+R[r] once per run.  Any input is correct (a triple of the run with
 * another (h, r) is handled on its own); period <= 0 or >= T: no grouping. */
int jmac_triple_l1_fwd_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr,
                           const int64_t* h, const int64_t* r, const int64_t* t, int64_t T,
                           int64_t period, int64_t d, float* score, jmac_stream_t stream);
/* Adjoint: dent[h[x]] += g s, drel[r[x]] += g s, dent[t[x]] -= g s with s = sign(ent[h]+rel[r]-ent[t])
 * (sign(0) = 0, as torch's norm backward).  dent [N,d] / drel [nrel,d] must be ZEROED by the caller (or
 * hold a gradient to accumulate into); float atomics; with `period` the run's dent[h] / drel[r] terms are
 * summed in registers and added once. */
int jmac_triple_l1_bwd_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr,
                           const int64_t* h, const int64_t* r, const int64_t* t, int64_t T,
                           int64_t period, int64_t d, const float* gscore, float* dent, int64_t ldde, float* drel, int64_t lddr,
                           jmac_stream_t stream);
/* The same adjoint for a score vector that fed the margin ranking loss of completion_loss directly (T = B (K + 1) triples, the
 * batch layout of train.py:347-352, src/jmac_model.py:351-378): the score gradient is derived inside the kernel from the
 * scores, gamma (device scalar) and gloss (device scalar, the loss' incoming gradient) by jmac_margin_loss_bwd_f32's rule --
 * no dscore vector, no launch for it.  dent / drel zeroed by the caller as above. */
int jmac_triple_l1_margin_bwd_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr,
                                  const int64_t* h, const int64_t* r, const int64_t* t, int64_t B, int64_t K,
                                  int64_t d, const float* score, const float* gamma, const float* gloss,
                                  float* dent, int64_t ldde, float* drel, int64_t lddr, jmac_stream_t stream);

/* Bitwise REPRODUCIBLE form of jmac_triple_l1_margin_bwd_f32 (same arguments + the row counts of the two gradient tables,
 * which it scales in a second pass): the gradient of the margin ranking loss is (gloss / (2 B K)) times an INTEGER matrix (the
 * weights of torch.max are 0, 1/2 or 1), so the float atomics add exact small integers -- order-independent below 2^24 -- and a
 * scaling pass finishes the tables.  dent [n_ent rows] / drel [n_rel rows] must be ZERO on entry.  4 B K >= 2^24: the plain form. */
int jmac_triple_l1_margin_bwd_exact_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr, const int64_t* h,
                                        const int64_t* r, const int64_t* t, int64_t B, int64_t K, int64_t d,
                                        const float* score, const float* gamma, const float* gloss, float* dent,
                                        int64_t ldde, int64_t n_ent, float* drel, int64_t lddr, int64_t n_rel,
                                        jmac_stream_t stream);

/* Pair cosine distance with a bitwise reproducible backward: the forward also leaves stats [L,4] = (a.b, a.a, b.b, 0) per pair;
 * the backward walks the 2 L (pair, side) incidences in the order the HOST sorted them by the gradient row they touch (stable:
 * ties in pair order) and writes every touched row ONCE with a plain store (sum in sorted order); untouched rows keep the
 * caller's zero fill.  rec [2L][4] int32, one record per SORTED position: {pair x, own table row, partner table row,
 * flags | gradient row} -- own / partner rows relative to the e1 / e2 pointers passed here; flags: bit 31 = first incidence of
 * its gradient row, bit 30 = side 1 (own row in e2, partner in e1), bit 29 = the gradient row is a row of de2 (else de1);
 * bits 0-28 = the gradient row.  The index vectors of an alignment loss are constant over many steps (seed links:
 * train.py:347-352), so the host sorts once per index tensor (jmac_amd.losses._pair_index). */
int jmac_pair_cosine_fwd_stats_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2, const int64_t* i1,
                                   const int64_t* i2, int64_t L, int64_t d, float* dist, float* stats,
                                   jmac_stream_t stream);
int jmac_pair_cosine_bwd_sorted_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2, int64_t L, int64_t d,
                                    const float* gdist, const float* stats, const int32_t* rec, float* de1, int64_t ldd1,
                                    float* de2, int64_t ldd2, jmac_stream_t stream);

/* Round 5 -- the same two adjoints as FIRST WRITERS of their gradient tables (no zero fill by the caller, no add of the step's
 * gradient contributions afterwards; src/jmac_model.py:237-249, 345-378):
 *  - jmac_pair_cosine_bwd_rows_f32: one wave per gradient ROW over [0, n1 + n2) (rows [0,n1) of de1, then n2 rows of de2; n2 = 0
 *    when both sides share one table), rowptr [n1+n2+1] = first sorted position (into rec) of every row's run; rows without
 *    incidences are written as zeros; sums in sorted order (the bits of the sorted form).  The incoming gradient is gdist [L], or
 *    the scalar gscalar[0] / gscale for every pair (the .mean() over the pairs of alignment_loss_simple, :249).
 *  - jmac_triple_l1_margin_bwd_exact2_f32: the exact-integer atomics go into PERSISTENT count tables cnt_ent [rows_ent, d] /
 *    cnt_rel [rows_rel, d] (dense, 16-byte aligned, ZERO on entry and left at zero again) at the windows [ent_off ..), [rel_off ..)
 *    the ids are local to; the scaling pass writes dent / drel (dense, all rows) = cnt * gloss / (2 B K), on top of their
 *    contents where acc_ent / acc_rel != 0 (e.g. the rows jmac_pair_cosine_bwd_rows_f32 wrote).  d % 4 == 0; 4 B K >= 2^24:
 *    JMAC_ERANGE (use jmac_triple_l1_margin_bwd_f32).
 *  - jmac_vec_mean_acc_f32: out[0] = mean(x[0..n)) + (add_to ? add_to[0] : 0), fixed summation order;
 *    jmac_margin_loss_fwd_acc_f32: jmac_margin_loss_fwd_f32 + add_to[0] -- a step's loss terms chain through add_to instead of
 *    through element-wise adds. */
int jmac_pair_cosine_bwd_rows_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2, int64_t L, int64_t d,
                                  const float* gdist, const float* gscalar, float gscale, const float* stats,
                                  const int32_t* rec, const int32_t* rowptr, int64_t n1, int64_t n2, float* de1,
                                  int64_t ldd1, float* de2, int64_t ldd2, jmac_stream_t stream);
int jmac_triple_l1_margin_bwd_exact2_f32(const float* ent, int64_t lde, const float* rel, int64_t ldr, const int64_t* h,
                                         const int64_t* r, const int64_t* t, int64_t B, int64_t K, int64_t d,
                                         const float* score, const float* gamma, const float* gloss, int64_t ent_off,
                                         int64_t rel_off, float* cnt_ent, float* cnt_rel, float* dent, int64_t rows_ent,
                                         int32_t acc_ent, float* drel, int64_t rows_rel, int32_t acc_rel,
                                         jmac_stream_t stream);
int jmac_vec_mean_acc_f32(const float* x, int64_t n, const float* add_to, float* out, jmac_stream_t stream);
int jmac_margin_loss_fwd_acc_f32(const float* score, int64_t B, int64_t K, const float* gamma, const float* add_to,
                                 float* loss, jmac_stream_t stream);

/* dist[x] = 1 - <u, v>, u = e1[i1[x]] / max(||.||, 1e-12), v = e2[i2[x]] / max(||.||, 1e-12)
 * (replaces F.normalize(E[idx]) x2 + sum of alignment_loss / alignment_loss_simple,
 * src/jmac_model.py:245-247, 271-291). */
int jmac_pair_cosine_fwd_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2,
                             const int64_t* i1, const int64_t* i2, int64_t L, int64_t d, float* dist,
                             jmac_stream_t stream);
/* Adjoint into de1 [N1,d] / de2 [N2,d] (zeroed by the caller); float atomics. */
int jmac_pair_cosine_bwd_f32(const float* e1, int64_t ld1, const float* e2, int64_t ld2,
                             const int64_t* i1, const int64_t* i2, int64_t L, int64_t d,
                             const float* gdist, float* de1, int64_t ldd1, float* de2, int64_t ldd2,
                             jmac_stream_t stream);

/* Margin ranking loss of completion_loss (src/jmac_model.py:351-378) on the [B + B*K] score vector of one batch, with the
 * reference's n-major consumption of the negative block kept (neg_{b,k} = score[B + k*B + b]):
 *   loss[0] = mean_{b<B,k<K} max(score[b] - neg_{b,k}, -gamma[0]) + gamma[0]
 * gamma: DEVICE pointer (the model's margin_completion parameter).  bwd: dscore [B + B*K] = gloss[0] * d loss / d score
 * (a tie diff == -gamma receives half the gradient, as torch.max(a, b) gives it).  One launch each way, fixed summation order. */
int jmac_margin_loss_fwd_f32(const float* score, int64_t B, int64_t K, const float* gamma, float* loss,
                             jmac_stream_t stream);
int jmac_margin_loss_bwd_f32(const float* score, int64_t B, int64_t K, const float* gamma,
                             const float* gloss, float* dscore, jmac_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch_scatter-compatible primitives (replace the third-party calls at src/jmac_model.py:105 and
 * modules/helper/message_passing.py:24,28) so the UNMODIFIED reference layer can run on this library.
 * index need not be sorted.  out must be pre-zeroed by the caller for scatter_sum.
 * --------------------------------------------------------------------------------------------- */
int jmac_scatter_sum_f32(const float* src, const int64_t* index, int64_t E, int64_t d, int64_t N,
                         float* out, jmac_stream_t stream);
size_t jmac_scatter_softmax_workspace_bytes(int64_t N, int64_t d);
int jmac_scatter_softmax_f32(const float* src, const int64_t* index, int64_t E, int64_t d, int64_t N,
                             float* out, void* ws, size_t ws_bytes, jmac_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* JMAC_HIP_H_ */
