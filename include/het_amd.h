/*
 * het_amd.h -- C ABI of libhet_amd.so: MI355X (gfx950) kernels for the
 * relational-GNN hot path of K-Wu/HET (segment GEMM, hetero edge-softmax,
 * gather-scatter aggregation; forward and backward).
 *
 * The reference exposes this path as torch custom ops registered with
 * TORCH_LIBRARY_FRAGMENT(torch_hrt, m) (a dispatcher boundary, not a C ABI:
 * the .inc.h files under hrt/include/DGLHackKernel/OpExport/).  Every entry point below is the
 * plain-pointer form of one of those ops: same name (prefixed het_), same
 * argument meaning, with each at::Tensor replaced by its device pointer and
 * sizes, and each torch::Dict<string, Tensor> flattened into named pointers.
 * het_amd/kernels.py re-registers the ops under torch.ops.torch_hrt.* on top of
 * this ABI; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions (as the reference, RGNNOps.inc.h:191-199, RGATOps.inc.h:35-48):
 *   - all pointers are DEVICE pointers to contiguous buffers on the current HIP
 *     device; floats are fp32, indices int64;
 *   - outputs are caller-allocated; "+=" in a comment means accumulated into,
 *     "=" means overwritten (no pre-zeroing needed);
 *   - launches go to `stream` (a hipStream_t; NULL = default stream), nothing
 *     synchronises the host, nothing allocates -- except het_grouping_create;
 *   - return value 0 = ok; otherwise an HET_ERR_* code, message in
 *     het_last_error() (thread-local).  Arguments are validated on the host
 *     before any launch (the reference only has compiled-out asserts).
 *
 * Paths in comments are relative to /root/reference/hrt/.
 */
#ifndef HET_AMD_H
#define HET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HET_OK 0
#define HET_ERR_INVALID_ARG 1
#define HET_ERR_HIP 2
#define HET_ERR_UNSUPPORTED 3

typedef void* het_stream; /* hipStream_t */

/* CompactAsOfNodeKind, include/kernel_enums.h:6-14 (the int the ops take) */
#define HET_KIND_DISABLED 0
#define HET_KIND_ENABLED 1
#define HET_KIND_DIRECT_INDEX 2
#define HET_KIND_DUAL_LIST 3
#define HET_KIND_DUAL_LIST_DIRECT_INDEX 4

/* replaces torch.ops.torch_hrt.build_debug_info (buildutils/genutils/gen_headers.py:17-41) */
const char* het_build_info(void);
const char* het_last_error(void);

/* Per-kernel device timing for bench.py's roofline figures (no reference counterpart: the reference times whole
 * forward / backward passes with CUDA events, RGNNUtils.py:291-311).  While enabled, the library records a HIP event
 * pair on the launch stream around every launch of its dominant kernels (the kernel alone, not the entry point's
 * fills or neighbouring launches).  het_kernel_timing_read synchronises the recorded events and returns the summed
 * duration and the number of launches of the kernels whose name starts with `name_prefix` (the __global__ name as
 * rocprofv3 prints it, without template arguments, e.g. "HET_gat_aggregate").  Enabling clears earlier records. */
int het_kernel_timing_enable(int on);
int het_kernel_timing_read(const char* name_prefix, double* total_ms, int64_t* launches);

/* ------------------------------------------------------------------------
 * Groupings: a one-time, device-side preprocessing of an index list that the
 * fast paths use to replace float atomics by segmented reductions.  A grouping
 * sorts the E positions of a relation-bucketed list by (relation, key[i]) --
 * or by key[i] alone when rel_ptrs is NULL -- and records the segments.  It has
 * no counterpart in the reference (which uses atomicAdd everywhere, e.g.
 * RGAT/RGATKernelsSeparateCOO.cu.h:77,195); every op accepts NULL instead and
 * then runs its atomics-based kernel.  Allocates device memory (hipMalloc, or
 * the caller's allocator: het_set_allocator) and synchronises `stream` once;
 * destroy with het_grouping_destroy.
 * ------------------------------------------------------------------------ */
/* Device memory the library keeps beyond a call (groupings, their lazily built lists) and its construction scratch comes
 * from hipMalloc / hipFree unless the host side installs its own allocator -- a framework binding passes its caching
 * allocator here (csrc/torch_export.cpp: c10::hip::HIPCachingAllocator; het_amd/_lib.py: torch.cuda.caching_allocator_alloc),
 * so that the groupings show up in that allocator's statistics, are returned to its pool when a cache evicts them, and no
 * hipFree (a device-wide synchronisation) happens on an op's path.  `stream`: the stream the memory is first used on.
 * alloc returns NULL on failure.  A pointer is always released through the allocator it came from, also after the
 * allocator was changed or reset (NULL, NULL: back to hipMalloc).  No counterpart in the reference (its launchers allocate
 * thrust::device_vector scratch per call, RGNNOps.inc.h:163-169). */
typedef void* (*het_alloc_fn)(size_t bytes, het_stream stream, void* user);
typedef void (*het_free_fn)(void* ptr, void* user);
int het_set_allocator(het_alloc_fn alloc, het_free_fn free_, void* user);
int het_allocator_is_external(void); /* 1 while an allocator is installed */
typedef struct het_grouping het_grouping;
/* payload0/payload1 (optional, [E]): per-position int64 arrays that are carried along in sorted
 * order (stored as int32), so that a grouped kernel reads them coalesced instead of chasing
 * perm -> array.  Each op documents what it expects there. */
int het_grouping_create(const int64_t* rel_ptrs /* [R+1] or NULL */, int64_t num_rels,
                        const int64_t* keys /* [E] */, int64_t num_positions, int64_t key_bound,
                        const int64_t* payload0, const int64_t* payload1,
                        het_stream stream, het_grouping** out);
void het_grouping_destroy(het_grouping* g);
/* Lifetime across streams.  A grouping's arrays belong to the stream it was created on.  With the default allocator
 * het_grouping_destroy ends in hipFree, which waits for the device.  With a stream-ordered allocator installed
 * (het_set_allocator: torch's caching allocator in both torch registrations) a freed block is handed out again to that stream
 * at once, so a binding that runs ops with the grouping on ANOTHER stream reports that stream here (cheap; idempotent): destroy
 * then makes the creation stream wait for an event on every reported stream before it releases anything -- no host
 * synchronisation.  Both registrations call it on every cache lookup (het_amd/plan.py, csrc/torch_export.cpp). */
void het_grouping_note_stream(const het_grouping* g, het_stream stream);
/* number of segments (distinct (relation, key) pairs) */
int64_t het_grouping_num_segments(const het_grouping* g);
/* device bytes the grouping currently holds (with the default hipMalloc they are outside the caller's allocator statistics) */
int64_t het_grouping_bytes(const het_grouping* g);
/* out[i] = sorted rank of position i (the inverse of the grouping's permutation), [E] */
int het_grouping_rank_of_position(const het_grouping* g, int64_t* out, het_stream stream);
/* map[r, k] = segment of (relation r, key k) of a grouping by (relation, key), -1 where that pair has no position; map
 * [R, num_keys] int32, num_keys >= the grouping's key bound.  (Segments are in ascending (relation, key) order: segment s is
 * row s of the sorted unique (relation, key) list of the positions -- het_node_row_map on that list gives the same map.) */
int het_grouping_segment_map(const het_grouping* g, int64_t num_keys, int32_t* map, het_stream stream);
/* out[j, :] = values[payload1 of sorted rank j, :] (H floats per entry, [E, H]): per-edge-id values (an edge norm) brought into the
 * grouping's order once, for callers that pass the same values every step -- the passes then read them as a stream */
int het_grouping_gather_payload1(const het_grouping* g, const float* values, int64_t H, float* out, het_stream stream);

/* ------------------------------------------------------------------------
 * a1  rgnn_relational_matmul            OpExport/RGNNOps.inc.h:238-295
 *   kind 0: ret[scatter_idx[i], h, :] = x[gather_idx[i], (h), :] . W[r(i), h]    i in [0, num_rows)
 *           (dict keys separate_coo_rel_ptrs / separate_coo_node_indices / separate_coo_eids)
 *   kind 1: ret[i, h, :] = x[gather_idx[i], (h), :] . W[r(i), h]   scatter_idx = NULL
 *           (dict keys unique_srcs_and_dests_rel_ptrs / unique_srcs_and_dests_node_indices)
 *   W [R,H,K,D]; x [*, K] if in1head else [*, H, K]; ret [*, H, D].
 *   by_rel_gather (optional, kind 0: the grouping of a2 -- positions by (relation, gather_idx), payload0 =
 *   scatter_idx) + workspace of S*H floats: with in1head and D == 1 (RGAT's --multiply_among_weights_first_flag:
 *   er = x[dst] . (W.attn_r)) the S distinct (relation, x row) products are formed once and duplicated to their
 *   positions -- same values.  NULL / 0: every position is computed on its own.
 * ------------------------------------------------------------------------ */
int het_rgnn_relational_matmul(int64_t kind, const int64_t* rel_ptrs, int64_t num_rels,
                               const int64_t* gather_idx, const int64_t* scatter_idx, int64_t num_rows,
                               const float* weights, const float* x, float* ret,
                               int64_t H, int64_t K, int64_t D, int in1head,
                               const het_grouping* by_rel_gather, void* workspace, int64_t workspace_bytes,
                               het_stream stream);

/* a1 + the attention-vector product of RGAT in the GEMM epilogue (extension; RGAT/models.py:288-296 computes
 * el = <feat, attn_l[r]> with a second, D_out = 1 segment GEMM that re-reads the [rows,H,D] tensor just written):
 *   ret as het_rgnn_relational_matmul with in1head = 1, and
 *   dot_out[scatter_idx[i], h] = < ret[scatter_idx[i], h, :], dot_w[r, h, :] >       dot_w [R,H,D], dot_out [*,H]
 * MFMA shapes only (K and H*D in {32, 64, 128}, D a power of two >= 4): HET_ERR_UNSUPPORTED otherwise.
 * by_rel_gather (optional, kind 0; the grouping of a2) + workspace of S*(H*D + H) floats: rows that share
 * (relation, gather_idx) are identical, so the GEMM runs on the S distinct rows and a broadcast kernel duplicates
 * them to their positions (same values; on ogbn-mag S = 3.7 M of E = 21.1 M rows by source).
 * comp_rows (optional, [S, H*D]): receives the distinct rows (the workspace then needs S*H floats only).
 * ret == NULL (grouping path only, H a power of two >= 4): the caller wants dot_out alone -- RGAT's er, whose
 * per-edge projection no other op reads -- and the [E,H,D] tensor is never written.
 * dot_grouping (optional, grouping path only): a grouping of the same (relation, gather_idx) keys whose payload0 says
 * where the dots of every position go (dot_out[payload0[i], h]) when that is not scatter_idx -- e.g. the rank of the
 * edge in the destination-grouped order of a4, so that a4 reads el / er as coalesced streams. */
int het_rgnn_relational_matmul_attn_dot(int64_t kind, const int64_t* rel_ptrs, int64_t num_rels,
                                        const int64_t* gather_idx, const int64_t* scatter_idx, int64_t num_rows,
                                        const float* weights, const float* x, float* ret, const float* dot_w,
                                        float* dot_out, int64_t H, int64_t K, int64_t D,
                                        const het_grouping* by_rel_gather, void* workspace, int64_t workspace_bytes,
                                        float* comp_rows, const het_grouping* dot_grouping, het_stream stream);

/* backward of the above for a caller that used only dot_out (RGAT's er = <x[dst].W, attn_r>): the gradient of ret is
 * grad_dot (x) dot_w[r], rank one per head, so the (relation, gather_idx) segment sums are taken over the [rows,H]
 * gradient: gs = SUM grad_dot;  G = gs * dot_w[r];  grad_x[v] (+)= G . Wt[r];  grad_w[r] (+)= x[v]^T (x) G.
 * Needs by_rel_gather (as in a2) and a workspace of (S*H rounded up to 4) + S*H*D floats; HET_ERR_UNSUPPORTED otherwise.
 * The gradient of dot_w: with comp_rows (the distinct rows kept from the forward) and grad_dot_w [R,H,D] it is formed
 * here as SUM_s gs[s,h] * comp_rows[s,h,:]; otherwise it is the plain D_out = 1 weight gradient over the [E,H,D]
 * tensor (het_backward_rgnn_relational_matmul with grad_x = NULL). */
int het_backward_rgnn_relational_matmul_attn_dot_only(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* gather_idx,
                                                      const int64_t* scatter_idx, int64_t num_rows, int64_t num_x_rows,
                                                      const float* weights_t, const float* x, const float* dot_w,
                                                      const float* grad_dot, float* grad_x, float* grad_w, int64_t H,
                                                      int64_t K, int64_t D, int accumulate,
                                                      const het_grouping* by_rel_gather, void* workspace,
                                                      int64_t workspace_bytes, const float* comp_rows,
                                                      float* grad_dot_w, het_stream stream);

/* a2  backward_rgnn_relational_matmul   OpExport/RGNNOps.inc.h:946-1010
 *   grad_x[gather_idx[i], (h), :] += gradout[scatter_idx[i], h, :] . Wt[r, h]   (heads summed iff in1head)
 *   grad_w[r, h]                  += x[gather_idx[i], (h), :]^T (x) gradout[scatter_idx[i], h, :]
 *   weights_t [R,H,D,K].  by_rel_gather: optional grouping of the rows by (relation, gather_idx):
 *   het_grouping_create(rel_ptrs, R, gather_idx, num_rows, <rows of x>, payload0 = scatter_idx, NULL, ...);
 *   with it, `workspace` must hold het_grouping_num_segments(g) * H * D floats (16-byte aligned).
 *   accumulate != 0: "+=" into the caller's buffers as the reference does (its wrapper zero-fills them,
 *   rgnn_layers_and_funcs.py:52-55); accumulate == 0: grad_x and grad_w are overwritten, no pre-zeroing
 *   needed (saves a fill + a read-modify-write pass when every grad_x row has a single writer).
 *   grad_x may be NULL for the D == 1 shapes (attention vectors, per-head or shared input): weight gradient only.
 *   `accumulate` is a bit set: HET_ACC_ADD (1) as above; HET_ACC_DISTINCT_ROWS (2), kind 1 only: the caller GUARANTEES
 *   that gather_idx holds a node at most once inside every relation (a unique (relation, node) list, what the reference's
 *   graph builders produce).  Only with that bit are the input gradients of a relation added with plain
 *   read-modify-write, relation by relation (deterministic, faster); without it -- the reference-named torch op -- any
 *   list is valid, duplicates included, and the adds are float atomics as in the reference's compact backward
 *   (RGNN/my_shmem_sgemm_func.cu.h:711-776).  A list with duplicates passed WITH the bit loses contributions. */
#define HET_ACC_ADD 1
#define HET_ACC_DISTINCT_ROWS 2
int het_backward_rgnn_relational_matmul(int64_t kind, const int64_t* rel_ptrs, int64_t num_rels,
                                        const int64_t* gather_idx, const int64_t* scatter_idx, int64_t num_rows,
                                        int64_t num_x_rows /* rows of x and grad_x */,
                                        const float* weights_t, const float* x, const float* gradout,
                                        float* grad_x, float* grad_w,
                                        int64_t H, int64_t K, int64_t D, int in1head, int accumulate,
                                        const het_grouping* by_rel_gather, void* workspace,
                                        int64_t workspace_bytes, het_stream stream);

/* a3  rgnn_relational_matmul_no_scatter_gather_list / backward_...   RGNNOps.inc.h:21-88, 660-753
 *   rows offsets[t]:offsets[t+1] use W[t]; x is [rows, K] (x_per_head = 0) or [rows, H, K]. */
int het_rgnn_relational_matmul_no_scatter_gather_list(const int64_t* offsets, int64_t num_types, int64_t num_rows,
                                                      const float* weights, const float* x, float* ret,
                                                      int64_t H, int64_t K, int64_t D, int x_per_head,
                                                      het_stream stream);
int het_backward_rgnn_relational_matmul_no_scatter_gather_list(const int64_t* offsets, int64_t num_types,
                                                               int64_t num_rows, const float* weights_t,
                                                               const float* x, const float* gradout,
                                                               float* grad_x, float* grad_w,
                                                               int64_t H, int64_t K, int64_t D, int x_per_head,
                                                               int accumulate, het_stream stream);

/* ------------------------------------------------------------------------
 * a4  relational_fused_gat_separate_coo    OpExport/RGATOps.inc.h:170-245
 *   exp[eids[i], h]  = leaky_exp(el[srow(i), h] + er[drow(i), h])
 *   sum[col[i], h]   = SUM_i exp            (all in-edges of the destination, every relation)
 *   ret[col[i],h,:]  = SUM_i exp/sum * feat[srow(i), h, :]
 *   sum, exp, ret are overwritten (zeroed internally; the reference accumulates into
 *   uninitialised buffers).
 * Row maps by kind (dict of the reference op flattened into four pointers):
 *   kind 0  srow = drow = eids[i]                               all four NULL
 *   kind 1  srow/drow = row of (r(i), row[i]) / (r(i), col[i]) in the unique list
 *           map_row_a = map_col_a = unique_srcs_and_dests_rel_ptrs [R+1]
 *           map_row_b = map_col_b = unique_srcs_and_dests_node_indices
 *   kind 3  map_row_a/b = unique_srcs_and_dests_rel_ptrs / ..._node_indices_row
 *           map_col_a/b = unique_srcs_and_dests_rel_ptrs_col / ..._node_indices_col
 *   kind 4  srow = map_row_a[eids[i]], drow = map_col_a[eids[i]]
 *           (edata_idx_to_inverse_idx_row / _col); map_*_b NULL
 *   feat [n_src_rows, H, D]; el [n_src_rows, H]; er [n_dst_rows, H]; sum [N,H]; exp [E,H]; ret [N,H,D].
 *   by_dst: optional grouping of the positions by col: het_grouping_create(NULL, 0, col, E, N,
 *   payload0 = eids, payload1 = NULL or the feat row of every position (srow), ...).
 *   exp_sorted (optional, [E,H], needs by_dst): a second copy of exp in the grouping's order, which the
 *   backward can stream instead of gathering exp / el / er by edge id.
 *   el_sorted, er_sorted (extension, kind 0 with by_dst and exp_sorted; both or neither): [E,H], the attention terms
 *   already in the grouping's order (row j = the edge at sorted rank j, het_grouping_rank_of_position).  The
 *   aggregation pass then forms exp itself from two coalesced streams -- no separate exp pass, no per-edge 16-byte
 *   gathers -- and writes exp_sorted; el, er may be NULL, exp (edge order) is written only if not NULL.
 * ------------------------------------------------------------------------ */
int het_relational_fused_gat_separate_coo(const int64_t* eids, const int64_t* rel_ptrs, const int64_t* row,
                                          const int64_t* col, int64_t num_rels, int64_t num_edges,
                                          int64_t num_nodes, int64_t kind,
                                          const int64_t* map_row_a, const int64_t* map_row_b,
                                          const int64_t* map_col_a, const int64_t* map_col_b,
                                          const float* feat, const float* el, const float* er,
                                          float* sum, float* exp, float* ret, float* exp_sorted,
                                          int64_t H, int64_t D, double slope,
                                          const het_grouping* by_dst, const float* el_sorted, const float* er_sorted,
                                          het_stream stream);

/* a5  backward_relational_fused_gat_separate_coo   OpExport/RGATOps.inc.h:465-551
 *   a = exp[eids[i],h] / sum[col[i],h]
 *   grad_feat[srow,h,:] += a * gradout[col[i],h,:]
 *   t = SUM_d gradout[col[i],h,d] * (feat[srow,h,d] - ret[col[i],h,d]) * a * (z > 0 ? 1 : slope)
 *   grad_el[srow,h] += t ; grad_er[drow,h] += t
 *   kind 0: every feat/el/er row belongs to one edge, so the three gradients are OVERWRITTEN (stores, no
 *   atomics, no pre-zeroing needed) and grad_er may alias grad_el (same values).  Other kinds: "+=".
 *   exp_sorted: optional copy of exp in by_dst order written by the forward (slope >= 0 required: the
 *   leaky-ReLU branch is then recovered from exp > 1); with it (kind 0, by_dst) el, er and exp are not read and
 *   may be NULL.
 *   Compact kinds, optional fast path (slope >= 0): by_src_row = het_grouping_create(NULL, 0, srow, E,
 *   n_src_rows, payload0 = eids, payload1 = col) over the feat row of every position, by_dst_row likewise
 *   over the er row (payload0 = eids), workspace of (N*2H + E*H) floats; grad_feat / grad_el / grad_er are
 *   then overwritten.  n_src_rows / n_dst_rows: rows of feat/el and of er.
 *   fold_attn_l (extension, NULL = the reference op): [R,H,D]; for a caller that formed
 *   el[e,h] = <feat[e,h,:], fold_attn_l[r,h,:]> (RGAT/models.py:288-296) the gradient through that product,
 *   grad_el[e,h] * fold_attn_l[r,h,:], is added into grad_feat by the same store -- kind 0 with by_dst only
 *   (error otherwise; by_dst must then carry payload1 = relation of every position); replaces a
 *   read-modify-write pass over the [E,H,D] gradient.  grad_fold_attn_l (optional with fold_attn_l, R <= 8): [R,H,D],
 *   receives += SUM_e grad_el[e,h] * feat[e,h,:] per relation -- the weight gradient of that product, from the feat
 *   rows the kernel reads anyway (saves a pass over feat); the caller zero-fills it.  With a workspace of
 *   64*R*H*D floats (kind 0) the workgroups spread their final atomic adds over 64 copies that a last tiny kernel
 *   sums -- on small graphs all workgroups finish together and would serialise on the R*H*D words.
 *   Compact kinds with the by_src_row / by_dst_row groupings: fold_attn_l works on the compact rows
 *   (el[u,h] = <feat[u,h,:], fold_attn_l[r(u),h,:]>), fold_row_rel_ptrs [R+1] = the rows' relation pointers.
 *   grad_el_sorted (extension, kind 0 with by_dst only): [E,H], receives grad_el in by_dst order (row j = the edge at
 *   sorted rank j, see het_grouping_rank_of_position) -- sequential 16-byte stores instead of scattered ones, and a
 *   consumer that sums it by (relation, destination) reads contiguous runs.  grad_el / grad_er may then be NULL. */
int het_backward_relational_fused_gat_separate_coo(const int64_t* eids, const int64_t* rel_ptrs,
                                                   const int64_t* row, const int64_t* col, int64_t num_rels,
                                                   int64_t num_edges, int64_t num_nodes, int64_t kind,
                                                   const int64_t* map_row_a, const int64_t* map_row_b,
                                                   const int64_t* map_col_a, const int64_t* map_col_b,
                                                   const float* feat, const float* el, const float* er,
                                                   const float* sum, const float* exp, const float* ret,
                                                   const float* exp_sorted,
                                                   const float* gradout, float* grad_feat, float* grad_el,
                                                   float* grad_er, int64_t H, int64_t D, double slope,
                                                   const het_grouping* by_dst, const het_grouping* by_src_row,
                                                   const het_grouping* by_dst_row, int64_t n_src_rows,
                                                   int64_t n_dst_rows, void* workspace, int64_t workspace_bytes,
                                                   const float* fold_attn_l, float* grad_fold_attn_l,
                                                   const int64_t* fold_row_rel_ptrs, float* grad_el_sorted,
                                                   het_stream stream);

/* a6  relational_fused_gat_csr / backward_relational_fused_gat_csr   RGATOps.inc.h:251-277, 430-460
 *   forward over the in-CSR (rows = dst, col_indices = src); backward over the out-CSR
 *   (rows = src, col_indices = dst).  compact != 0: feat/el/er rows are (relation, node) rows of
 *   the unique list (uniq_rel_ptrs, uniq_node_idx), binary-searched. */
int het_relational_fused_gat_csr(const int64_t* in_row_ptrs, const int64_t* in_col, const int64_t* in_eids,
                                 const int64_t* in_reltypes, int64_t num_nodes, int64_t num_edges,
                                 const int64_t* uniq_rel_ptrs, const int64_t* uniq_node_idx, int64_t num_rels,
                                 const float* feat, const float* el, const float* er,
                                 float* sum, float* exp, float* ret, int64_t H, int64_t D, double slope,
                                 int compact, het_stream stream);
int het_backward_relational_fused_gat_csr(const int64_t* out_row_ptrs, const int64_t* out_col,
                                          const int64_t* out_eids, const int64_t* out_reltypes,
                                          int64_t num_nodes, int64_t num_edges,
                                          const int64_t* uniq_rel_ptrs, const int64_t* uniq_node_idx,
                                          int64_t num_rels, const float* feat, const float* el, const float* er,
                                          const float* sum, const float* exp, const float* ret,
                                          const float* gradout, float* grad_feat, float* grad_el, float* grad_er,
                                          int64_t H, int64_t D, double slope, int compact, het_stream stream);

/* ------------------------------------------------------------------------
 * a7  rgcn_layer1_separate_coo            OpExport/RGCNOps.inc.h:84-138
 *   ret[col[i], :] += (x[row[i], :] * norm[eids[i]]) . W[r(i)]        W [R,K,D]
 *   by_rel_dst: optional grouping of the positions by (relation, col):
 *   het_grouping_create(rel_ptrs, R, col, E, N, payload0 = row, payload1 = eids, ...) with a workspace of
 *   het_grouping_num_segments(g) * K floats; the backward takes the grouping by (relation, row) with
 *   payload0 = col, payload1 = eids and num_segments * D floats.
 * a8  backward_rgcn_layer1_separate_coo   OpExport/RGCNOps.inc.h:368-467
 *   grad_x[row[i], :] += (gradout[col[i], :] * norm[eids[i]]) . Wt[r]   (intended direction)
 *   grad_w[r]         += (x[row[i], :] * norm[eids[i]])^T (x) gradout[col[i], :]
 *   grad_norm is not written (disabled in the reference, my_shmem_sgemm_func_rgcn_hgt.cu.h:680-684).
 * ------------------------------------------------------------------------ */
int het_rgcn_layer1_separate_coo(const int64_t* rel_ptrs, const int64_t* eids, const int64_t* row,
                                 const int64_t* col, int64_t num_rels, int64_t num_edges, int64_t num_nodes,
                                 const float* x, const float* weights, const float* norm, float* ret,
                                 int64_t K, int64_t D, const het_grouping* by_rel_dst, void* workspace,
                                 int64_t workspace_bytes, het_stream stream);
int het_backward_rgcn_layer1_separate_coo(const int64_t* rel_ptrs, const int64_t* eids, const int64_t* row,
                                          const int64_t* col, int64_t num_rels, int64_t num_edges,
                                          int64_t num_nodes, const float* x, const float* weights_t,
                                          const float* norm, float* grad_norm, float* grad_x,
                                          const float* gradout, float* grad_w, int64_t K, int64_t D,
                                          const het_grouping* by_rel_src, void* workspace, int64_t workspace_bytes,
                                          het_stream stream);

/* a9  rgcn_node_mean_aggregation_compact_as_of_node_separate_coo (+ backward)  RGCNOps.inc.h:24-82, 303-366
 *   ret[col[i], :]        = SUM_i enorm[eids[i]] * feat[crow(i), :]            (ret overwritten)
 *   grad_feat[crow(i), :] += enorm[eids[i]] * gradout[col[i], :]
 *   crow(i) = map_a[eids[i]] if direct (inverse_indices_row), else the row of (r(i), row[i]) in the
 *   unique list map_a = rel_ptrs_row [R+1], map_b = node_indices_row  (intended mapping).
 *   Optional fast paths (X a power of two in 4..256): by_dst = het_grouping_create(NULL, 0, col, E, N, payload0 = crow
 *   per position, payload1 = eids); by_src_row = het_grouping_create(NULL, 0, crow, E, n_src_rows, payload0 = col,
 *   payload1 = eids): segmented sums instead of E*X float atomics. */
int het_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(
    const int64_t* eids, const int64_t* rel_ptrs, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const int64_t* map_a, const int64_t* map_b, const float* feat,
    const float* enorm, float* ret, int64_t X, int direct, const het_grouping* by_dst, het_stream stream);
int het_backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(
    const int64_t* eids, const int64_t* rel_ptrs, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const int64_t* map_a, const int64_t* map_b, const float* feat,
    const float* enorm, const float* ret, const float* gradout, float* grad_feat, int64_t X, int direct,
    const het_grouping* by_src_row, int64_t n_src_rows, het_stream stream);

/* ------------------------------------------------------------------------
 * a10  hgt_full_graph_edge_softmax_ops_separate_coo     OpExport/HGTOpsEdgeParallel.inc.h:18-31 -> HGTOps.inc.h:23-106
 *   m[eids[i],h] = exp(score[eids[i],h] * mu[r(i),h]);  sum[col[i],h] = SUM_i m;  a = m / sum[col[i],h]
 *   (denominator keyed by the DESTINATION over all E edges: intended semantics).
 *   sum, m, a are overwritten.
 *      backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo   HGTOps.inc.h:597-648
 *   tmp[col[i],h] = SUM_i a*grad_a;  c = (grad_a - tmp[col[i],h]) * a
 *   grad_score[eids[i],h] = c * mu[r,h];  grad_mu[r,h] += c * score[eids[i],h]      
 *   by_dst (optional, H % 4 == 0): het_grouping_create(NULL, 0, col, E, N, payload0 = eids, payload1 = relation of
 *   every position) -- the denominators / tmp become segmented sums instead of E*H float atomics.
 * ------------------------------------------------------------------------ */
int het_hgt_full_graph_edge_softmax_ops_separate_coo(const int64_t* row, const int64_t* col, const int64_t* eids,
                                                     const int64_t* rel_ptrs, int64_t num_rels, int64_t num_edges,
                                                     int64_t num_nodes, const float* score, const float* mu,
                                                     float* sum, float* m, float* a, int64_t H,
                                                     const het_grouping* by_dst, het_stream stream);
int het_backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(
    const int64_t* row, const int64_t* col, const int64_t* eids, const int64_t* rel_ptrs, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const float* score, const float* a, const float* grad_a, const float* mu,
    float* grad_score, float* grad_mu, float* tmp, int64_t H, const het_grouping* by_dst, het_stream stream);

/* a11  hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo (+ backward)
 *      OpExport/HGTOpsEdgeParallel.inc.h:33-88, 295-369
 *   new_h[col[i],h,:] += (v[row[i],h,:] * a[eids[i],h]) . W[r,h]                      W [R,H,dk,dout]
 *   grad_v[row[i],h,:] += (gradout[col[i],h,:] * a) . Wt[r,h]     (intended direction)
 *   grad_w[r,h]        += (v[row[i],h,:] * a)^T (x) gradout[col[i],h,:]
 *   grad_a[eids[i],h]   = < gradout[col[i],h,:] . Wt[r,h], v[row[i],h,:] >            Wt [R,H,dout,dk]
 *   Optional fast paths: by_rel_dst / by_rel_src are the groupings of rgcn_layer1 (by (relation, col) with
 *   payload0 = row, payload1 = eids; by (relation, row) with payload0 = col, payload1 = eids) and a workspace of
 *   num_segments * H * dk (forward) / num_segments * H * dout (backward) floats; heads wider than 16 floats
 *   (dk = dout = 32, 64, 128: the reference sweep's --num_heads 1) need 2 * num_segments * H * dout + R*H*dk*dout
 *   in the backward (grad_a is then taken from the per-(relation, source) message rows). */
int het_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(
    const int64_t* rel_ptrs, const int64_t* eids, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const float* v, const float* weights, const float* a, float* new_h,
    int64_t H, int64_t dk, int64_t dout, const het_grouping* by_rel_dst, void* workspace, int64_t workspace_bytes,
    het_stream stream);
int het_backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(
    const int64_t* rel_ptrs, const int64_t* eids, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const float* v, const float* weights_t, const float* a, const float* new_h,
    float* grad_v, float* grad_w, float* grad_a, const float* gradout, int64_t H, int64_t dk, int64_t dout,
    const het_grouping* by_rel_src, void* workspace, int64_t workspace_bytes, het_stream stream);

/* a12  rgnn_inner_product_right_node_separatecoo (+ backward)    OpExport/RGNNOps.inc.h:609-658, 1131-1181
 *   out[eids[i],h] = < left[lrow(i),h,:], right[row[i],h,:] >                               (out overwritten)
 *   lrow(i) = eids[i] (kind 0) | row of (r(i), col[i]) in the unique list map_a = rel_ptrs, map_b = node ids
 *             (kind 1) | map_a[eids[i]] (kind 2, edata_idx_to_inverse_idx)
 *   grad_left[lrow(i),h,:] += gradout[eids[i],h] * right[row[i],h,:];   grad_right[row[i],h,:] += ... * left[lrow(i),h,:]
 *   (accumulating; the reference kernel stores without atomics).  accumulate == 0: both
 *   gradients are overwritten instead (n_left_rows / n_right_rows = their row counts).  by_right: optional
 *   grouping of the positions by row (het_grouping_create(NULL, 0, row, E, n_right_rows, payload0 = lrow per
 *   position, payload1 = eids)), used for kinds 0 and 2.  by_left (kind 2): grouping of the positions by lrow
 *   (payload0 = row, payload1 = eids): the gradient of a shared compact row becomes a segmented sum instead of
 *   float atomics.  (A caller holding kind-1 lists can precompute lrow per edge id once and call with kind 2.) */
int het_rgnn_inner_product_right_node_separatecoo(int64_t kind, const int64_t* map_a, const int64_t* map_b,
                                                  const int64_t* rel_ptrs, const int64_t* eids, const int64_t* row,
                                                  const int64_t* col, int64_t num_rels, int64_t num_edges,
                                                  const float* left, const float* right, float* out, int64_t H,
                                                  int64_t D, het_stream stream);
int het_backward_inner_product_right_node_separatecoo(int64_t kind, const int64_t* map_a, const int64_t* map_b,
                                                      const int64_t* rel_ptrs, const int64_t* eids,
                                                      const int64_t* row, const int64_t* col, int64_t num_rels,
                                                      int64_t num_edges, const float* left, const float* right,
                                                      const float* gradout, float* grad_left, float* grad_right,
                                                      int64_t H, int64_t D, int accumulate,
                                                      const het_grouping* by_right, const het_grouping* by_left,
                                                      int64_t n_left_rows, int64_t n_right_rows, het_stream stream);

/*      hgt_full_graph_hetero_attention_ops_coo (+ backward)     OpExport/HGTOpsEdgeParallel.inc.h:95-158, 166-293
 *   inner[eids[i],h,:] = k[row[i],h,:] . W[r,h];   score[eids[i],h] = < inner[eids[i],h,:], q[col[i],h,:] >
 *   grad_q[col[i],h,:] += gs * inner[eids[i],h,:];  grad_k[row[i],h,:] += (gs * q[col[i],h,:]) . Wt[r,h];
 *   grad_w[r,h] += k[row[i],h,:]^T (x) (gs * q[col[i],h,:]),   gs = grad_score[eids[i],h]
 *   Optional fast path: by_dst (by col, payload0 = eids), by_rel_src (by (relation, row), payload0 = col,
 *   payload1 = eids), n_q_rows rows of q, workspace of by_rel_src segments * H * dout floats. */
int het_hgt_full_graph_hetero_attention_ops_coo(const int64_t* row, const int64_t* col, const int64_t* eids,
                                                const int64_t* rel_ptrs, int64_t num_rels, int64_t num_edges,
                                                const float* k, const float* q, const float* weights, float* inner,
                                                float* score, int64_t H, int64_t dk, int64_t dout, het_stream stream);
int het_backward_hgt_full_graph_hetero_attention_ops_coo(const int64_t* row, const int64_t* col, const int64_t* eids,
                                                         const int64_t* rel_ptrs, int64_t num_rels, int64_t num_edges,
                                                         float* grad_w, const float* weights_t, const float* k,
                                                         const float* q, const float* inner, const float* grad_score,
                                                         float* grad_k, float* grad_q, int64_t H, int64_t dk,
                                                         int64_t dout, const het_grouping* by_dst,
                                                         const het_grouping* by_rel_src, int64_t n_q_rows,
                                                         void* workspace, int64_t workspace_bytes, het_stream stream);

/* ------------------------------------------------------------------------
 * HGT attention + message aggregation on the distinct (relation, source) rows, without a per-edge float tensor
 * (layer-level fusion; no reference op of its own).  Inside the one-node HGT layer it replaces the chain
 * rgnn_relational_matmul (relation_att) -> rgnn_inner_product_right_node -> hgt_full_graph_edge_softmax_ops ->
 * hgt_full_graph_fused_message_calc_and_mean_aggregation and their backward ops (HGT/models.py:172-262;
 * OpExport/HGTOps.inc.h, OpExport/HGTOpsEdgeParallel.inc.h; kernels hrt/include/DGLHackKernel/HGT/ *.cu.h): same values,
 *   a_e[h] = exp(s_e[h]) / SUM_{e' into dst_e} exp(s_e'[h]),  s_e[h] = <k'[srow_e,h,:], q[dst_e,h,:]>,
 *   out[v,h,:] = SUM_{e into v} a_e[h] * m[srow_e,h,:]
 * with k' = k . relation_att . (relation_pri / sqrt(dk)) and m = v . relation_msg formed per distinct (relation, source)
 * row by the caller (one segment GEMM from the layer input when the typed projections are folded into the weights).
 *   kv_c [S_row, 2, H, D]: k' then m of every (relation, source) row;  q [N,H,D];  out [N,H,D];
 *   lsum [N,H] = log SUM exp(s) (the softmax subtracts a running maximum: finite for any score, unlike the reference's raw exp)
 *   by_dst:  het_grouping_create(NULL, 0, col, E, N, payload0 = (relation, source) row of every position, NULL)
 *   by_srow: het_grouping_create(NULL, 0, that row of every position, E, S_row, payload0 = col, NULL)
 * forward overwrites lsum and out (zero rows for destinations without in-edges); backward overwrites grad_kv_c [S_row,2,H,D]
 * and grad_q [N,H,D].  workspace (backward): het_hgt_backward_compact_workspace(N, H) bytes, 16-byte aligned; (forward):
 * het_hgt_aggregate_compact_workspace(by_dst, H, D) bytes (0 unless a destination has more than 256 in-edges).
 * Shapes: H*D in {8, 16, 32, 64, 128}, D a power of two >= 8 (het_hgt_compact_shape_ok); else HET_ERR_UNSUPPORTED. */
int het_hgt_compact_shape_ok(int64_t H, int64_t D);
int het_hgt_aggregate_compact(const het_grouping* by_dst, const float* kv_c, const float* q, float* lsum, float* out,
                              int64_t num_nodes, int64_t num_src_rows, int64_t H, int64_t D, void* workspace,
                              int64_t workspace_bytes, het_stream stream);
int64_t het_hgt_aggregate_compact_workspace(const het_grouping* by_dst, int64_t H, int64_t D);
int64_t het_hgt_backward_compact_workspace(int64_t num_nodes, int64_t H);
int het_hgt_backward_compact(const het_grouping* by_dst, const het_grouping* by_srow, const float* kv_c, const float* q,
                             const float* lsum, const float* out, const float* gradout, float* grad_kv_c, float* grad_q,
                             int64_t num_nodes, int64_t num_src_rows, int64_t H, int64_t D, void* workspace,
                             int64_t workspace_bytes, het_stream stream);

/* The per-step folding of the HGT layer's parameters into one source-side weight per relation (round 5; extension, reached from
 * het_amd/backend/hgt_fused_layer.py only), forward and backward as one / two launches instead of ~14 + ~24 torch kernels:
 *   w_kv[r, i, h*dk+e]         = pri[r,h] / sqrt(dk) * SUM_d k_lin[st(r), i, h*dk+d] * A[r,h,d,e]
 *   w_kv[r, i, H*dk + h*dk+e]  =                      SUM_d v_lin[st(r), i, h*dk+d] * msg[r,h,d,e]
 * A = rel_att (transpose_att = 0: the fused score <k . att, q>) or its transpose over (d, e) (1: <q . att, k>), st = src_type [R] int64
 * (device): the node type of every relation's sources.  The factors are the reference's, composed per edge there
 * (HGT/models.py:159-262: K_linear / V_linear per node type, relation_att, relation_pri / sqrt_dk, relation_msg).
 * k_lin, v_lin [T, K_in, H*dk]; rel_att, rel_msg [R,H,dk,dk]; rel_pri [R,H]; w_kv [R, K_in, 2*H*dk].  The backward overwrites all five
 * gradients (same shapes as the parameters). */
int het_hgt_fold_source_weights(const float* k_lin, const float* v_lin, const float* rel_att, const float* rel_msg,
                                const float* rel_pri, const int64_t* src_type, int64_t num_types, int64_t num_rels, int64_t H,
                                int64_t dk, int64_t K_in, int transpose_att, float* w_kv, het_stream stream);
int het_hgt_fold_source_weights_backward(const float* grad_w_kv, const float* k_lin, const float* v_lin, const float* rel_att,
                                         const float* rel_msg, const float* rel_pri, const int64_t* src_type, int64_t num_types,
                                         int64_t num_rels, int64_t H, int64_t dk, int64_t K_in, int transpose_att, float* grad_k_lin,
                                         float* grad_v_lin, float* grad_att, float* grad_msg, float* grad_pri, het_stream stream);

/* ------------------------------------------------------------------------
 * RGAT on the distinct (relation, node) rows without a per-edge float tensor (layer-level fusion; no reference op of
 * its own).  Replaces the pair relational_fused_gat_separate_coo / backward_... with CompactAsOfNodeKind 4
 * (OpExport/RGATOps.inc.h:170-245, 465-551; kernels RGAT/RGATKernelsSeparateCOO.cu.h:17-204,
 * RGATBackwardKernelsSeparateCOO.cu.h:9-117) inside the one-node RGAT layer: both passes form
 * exp(leaky_relu(el_c[srow] + er_c[drow])) from the compact tables instead of writing / re-reading exp [E,H].
 *   feat_c [S_row,H,D], el_c [S_row,H]: per distinct (relation, source) row;  er_c [S_col,H]: per (relation, destination)
 *   by_dst:  het_grouping_create(NULL, 0, col, E, N, payload0 = feat row of every position, payload1 = its er row)
 *   by_srow: het_grouping_create(NULL, 0, feat row of every position, E, S_row, payload0 = col, payload1 = er row)
 *   by_drow: het_grouping_create(NULL, 0, er row of every position, E, S_col, payload0 = rank of the position in by_srow
 *            (het_grouping_rank_of_position), NULL)
 * Softmax without overflow: the reference exponentiates the raw pre-activation (gatLeakyReluExp, GAT/FusedGAT.cu.h:23-26; the
 *   reference-named a4 / a5 keep that because exp and sum are API tensors there).  These two entry points subtract a running
 *   maximum per (destination, head): `sum` receives lse[v,h] = log(SUM_e exp(leaky_relu(el + er))) instead of the sum itself and
 *   the backward forms the attention weight as exp(leaky_relu(el + er) - lse[v,h]) -- the reference's value wherever its formula
 *   is finite, finite for any pre-activation.  The stored lse of a destination with in-edges is never exactly 0 (an exact 0 is
 *   stored as FLT_MIN), so `sum[v,h] == 0` -- the forward's zero fill -- marks a node without in-edges: the backward skips the
 *   `ret` rows of those nodes, so `sum` has to be the forward's output.  workspace (forward): het_rgat_aggregate_compact_workspace(by_dst, H, D) bytes,
 *   16-byte aligned (0 unless a destination has more than 256 in-edges: its work items park their partial sums there and a
 *   finishing pass brings them to one maximum -- no float atomics).
 * forward:  sum [N,H] (= lse), ret [N,H,D] are overwritten (a4's ret; exp is not produced).  h_inout [h_rows, H*D] (optional):
 *   the layer output so far (self-loop + bias, het_rows_linear_bias); ret's row is added to it in place for every
 *   destination < h_rows, and ret is then defined only for destinations WITH in-edges (no 0.5 GB zero fill).
 * backward: grad_feat_c, grad_el_c, grad_er_c are overwritten (a5's outputs on the compact rows).  fold_attn_l [R,H,D]
 *   (optional, with row_rel_ptrs [R+1] = relation pointers of the feat rows): adds grad_el_c[u,h] * fold_attn_l[r(u),h,:]
 *   into grad_feat_c -- the gradient through el_c = <feat_c, attn_l[r]>.  grad_bias [H*D] (optional): column sums of
 *   gradout's first bias_rows rows (the layer's bias gradient) from the pass that reads gradout anyway.
 *   workspace: het_rgat_backward_compact_workspace(N, E, H, D, grad_bias != NULL) bytes, 16-byte aligned. */
int het_rgat_aggregate_compact(const het_grouping* by_dst, const float* feat_c, const float* el_c, const float* er_c,
                               float* sum, float* ret, int64_t num_nodes, int64_t H, int64_t D, double slope,
                               float* h_inout, int64_t h_rows, void* workspace, int64_t workspace_bytes, het_stream stream);
int64_t het_rgat_aggregate_compact_workspace(const het_grouping* by_dst, int64_t H, int64_t D);
int64_t het_rgat_backward_compact_workspace(int64_t num_nodes, int64_t num_edges, int64_t H, int64_t D, int with_bias);
int het_rgat_backward_compact(const het_grouping* by_srow, const het_grouping* by_drow, const float* feat_c,
                              const float* el_c, const float* er_c, const float* sum, const float* ret,
                              const float* gradout, float* grad_feat_c, float* grad_el_c, float* grad_er_c,
                              const float* fold_attn_l, const int64_t* row_rel_ptrs, int64_t num_rels,
                              float* grad_bias, int64_t bias_rows, int64_t num_nodes, int64_t num_src_rows,
                              int64_t num_dst_rows, int64_t H, int64_t D, double slope, void* workspace,
                              int64_t workspace_bytes, het_stream stream);

/* The same pair with grad_er taken from per-RUN sums instead of a per-edge term (a run = the edges of one (relation,
 * destination) pair = one er row).  The reference's backward adds grad_er edge by edge with float atomics
 * (RGATBackwardKernelsSeparateCOO.cu.h:60-117); het_rgat_backward_compact writes a per-edge term [E,H] in (relation, source)
 * order and sums it in (relation, destination) order.  gradout[v] is common to a run, so the forward -- which visits the feat
 * rows of the run anyway -- can leave
 *   q_rows[w,h,:] = SUM_e w_e dl_e feat_c[srow_e,h,:],  q_sum[w,h] = SUM_e w_e dl_e,   w_e = exp(leaky(el + er) - q_ref[w,h]),
 *   dl_e = 1 or slope (the leaky-ReLU branch of edge e)
 * per er row w, and grad_er_c[w,h] = exp(q_ref[w,h] - lse[v,h]) (<gradout[v,h,:], q_rows[w,h,:]> - <gradout, ret>[v,h] q_sum[w,h]).
 *   by_dst_rel: het_grouping_create(NULL, 0, col * num_rels + relation of the position, E, N * num_rels, NULL, NULL) -- the same
 *     sorted order as by_dst when the positions are relation-major (a separate-COO list); its work items never cross a run.
 *   q_rows [S_col,H,D], q_sum / q_ref [S_col,H]: written by the forward, read by the backward.  drow_nodes [S_col] int64:
 *     destination node of every er row.  Shapes: rows of 32 / 64 / 128 floats with heads of >= 16 (else HET_ERR_UNSUPPORTED).
 *   attn_l [R,H,D] + feat_rel_ptrs_host (HOST array [R+1]: first feat row of every relation; feat rows are relation-major) --
 *     optional, both or neither: when el_c IS <feat_c[row,h,:], attn_l[relation of the row,h,:]> (the layer's case) the pass forms it
 *     from the row it gathers anyway instead of gathering el_c per edge (heads of 16 floats, up to 8 relations; el_c is still read
 *     by the other shapes and by the backward).
 *   workspaces: het_rgat_aggregate_compact_runs_workspace(by_dst, by_dst_rel, num_rels, H, D, stream) -- one record per work item of
 *     a destination with more than 128 in-edges (HET_RGAT_HUB_MIN); the first call lists those items on `stream` (kept with by_dst_rel and
 *     rebuilt if it is later used with another by_dst object); -1 on error;  het_rgat_backward_compact_runs_workspace (below) */
int64_t het_rgat_aggregate_compact_runs_workspace(const het_grouping* by_dst, const het_grouping* by_dst_rel, int64_t num_rels,
                                                  int64_t H, int64_t D, het_stream stream);
int het_rgat_aggregate_compact_runs(const het_grouping* by_dst, const het_grouping* by_dst_rel, int64_t num_rels,
                                    const float* feat_c, const float* el_c, const float* er_c, float* sum, float* ret,
                                    int64_t num_nodes, int64_t H, int64_t D, double slope, float* h_inout, int64_t h_rows,
                                    float* q_rows, float* q_sum, float* q_ref, int64_t num_dst_rows, const float* attn_l,
                                    const int64_t* feat_rel_ptrs_host, void* workspace, int64_t workspace_bytes, het_stream stream);
int het_rgat_backward_compact_runs(const het_grouping* by_srow, const float* q_rows, const float* q_sum, const float* q_ref,
                                   const int64_t* drow_nodes, const float* feat_c, const float* el_c, const float* er_c,
                                   const float* sum, const float* ret, const float* gradout, float* grad_feat_c,
                                   float* grad_el_c, float* grad_er_c, const float* fold_attn_l, const int64_t* row_rel_ptrs,
                                   int64_t num_rels, float* grad_bias, int64_t bias_rows, int64_t num_nodes,
                                   int64_t num_src_rows, int64_t num_dst_rows, int64_t H, int64_t D, double slope,
                                   float* grad_attn_l, void* workspace, int64_t workspace_bytes, het_stream stream);
/* grad_attn_l [R,H,D] (optional; needs fold_attn_l, <= 8 relations): the weight gradient of el_c = <feat_c, attn_l[r]>,
 *   SUM_u grad_el_c[u,h] feat_c[u,h,:] per relation, overwritten -- formed from the rows the source-row kernels hold where a
 *   segment ends (per-workgroup partial rows + a finishing pass) instead of a row-dot pass that reads feat_c again.
 * workspace: het_rgat_backward_compact_runs_workspace(by_srow, N, num_dst_rows, H, D, grad_bias != NULL, grad_attn_l != NULL, stream)
 *   bytes (the first call with the attention gradient builds by_srow's packs on `stream`); -1 on error.  It holds, beside the
 *   per-destination {lse, <gradout, ret>} pairs, one 16-byte record {er, lse, <gradout, ret>, 0} per (er row, head): the source-row
 *   kernels fetch everything they need from the destination side of an edge with one load (round 4; HET_RGAT_DROW_REC=0: two). */
int64_t het_rgat_backward_compact_runs_workspace(const het_grouping* by_srow, int64_t num_nodes, int64_t num_dst_rows, int64_t H,
                                                 int64_t D, int with_bias, int with_attn_grad, het_stream stream);

/* The two halves of a2 (backward_rgnn_relational_matmul, one input head, matrix-core shapes) as separate calls, so that a
 * caller can order them around a collective (het_amd/dist.py).  Rows i in [0, num_rows) of relation-bucketed lists:
 *   dx: grad_x[gather_idx[i], :] (+)= gradout[g_rows[i], :] . Wt[r(i)]      atomic 0: "=";  1: "+=" with float atomics (rows
 *       may repeat);  2: "+=" for lists whose gather_idx are distinct inside every relation (a unique (relation, node)
 *       list): relation by relation with plain read-modify-write (up to 8 relations, atomics beyond)
 *   dw: grad_w[r(i)]             (+)= x[gather_idx[i], :]^T (x) gradout[g_rows[i], :]
 * gather_idx / g_rows NULL = row i.  weights_t [R,H,D,K], grad_w [R,H,K,D]. */
int het_rows_matmul_backward_dx(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* gather_idx, const int64_t* g_rows,
                                int64_t num_rows, const float* weights_t, const float* gradout, float* grad_x, int64_t H,
                                int64_t K, int64_t D, int atomic, het_stream stream);
int het_rows_matmul_backward_dw(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* gather_idx, const int64_t* g_rows,
                                int64_t num_rows, const float* x, const float* gradout, float* grad_w, int64_t H, int64_t K,
                                int64_t D, int accumulate, het_stream stream);
/* The same with the column sums of the gradout rows of the launch from the same pass: colsum[H*D] = SUM_i gradout[g_rows[i], :]
 * (always "="; NULL: none).  For a list that names every output row once (the self-loop product of the RGAT layer,
 * RGAT/models.py:378-381) that is the bias gradient, which otherwise costs its own pass over gradout. */
int het_rows_matmul_backward_dw_colsum(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* gather_idx,
                                       const int64_t* g_rows, int64_t num_rows, const float* x, const float* gradout,
                                       float* grad_w, float* colsum, int64_t H, int64_t K, int64_t D, int accumulate,
                                       het_stream stream);

/* ------------------------------------------------------------------------
 * Node-major input gradient of the one-node RGAT layer (layer-level fusion; no reference op of its own).  It replaces,
 * inside het_amd/backend/rgat_fused_layer.py, the per-relation input-gradient passes of a2 (backward_rgnn_relational_matmul
 * with CompactAsOfNodeKind 1, OpExport/RGNNOps.inc.h:946-1010 -> kernels RGNN/my_shmem_sgemm_func.cu.h:711-776), the self-loop's
 * a3 backward (RGNNOps.inc.h:660-753) and the D_out = 1 product of the attention-vector side: ONE pass over the nodes forms
 *   grad_x[n,:]  = grad_h[n,:] . loop_wt  +  SUM_r g_rows[row_map[r,n],:] . weights_t[r]  +  SUM_r,h g_er[dst_map[r,n],h] * wa_t[r,h,:]
 * for the nodes n in [n_begin, n_end) and stores the row once; terms whose map entry is -1 (and the grad_h term for
 * n >= n_loop) are absent.
 *   row_map / dst_map [R, num_nodes] int32: row of (relation, node) in the unique (relation, source) / (relation,
 *   destination) list, -1 if the node has none (het_node_row_map);  grad_h [n_loop, H*D];  g_rows [S_row, H*D];
 *   g_er [S_col, H] (NULL: no such term);  loop_wt [H*D, K] (= loop_weight^T);  weights_t [R,H,D,K];  wa_t [R,H,K].
 *   node_order [num_nodes] int32 (optional): positions [n_begin, n_end) of this list are the nodes of the call (NULL: node p at
 *   position p).  The kernel multiplies 32-node tiles per relation that has a row in the tile; a list sorted by WHICH relations a
 *   node has rows in makes the tiles homogeneous (no zero rows: 26 % fewer matrix-core instructions on ogbn-mag).
 *   Shapes: K and H*D in {32, 64}, H in {1,2,4,8}, R*H <= 32, all 1 + R weights resident in LDS (het_rgat_node_gemm_ok);
 *   HET_ERR_INVALID_ARG otherwise -- callers fall back to the per-relation entry points. */
int het_rgat_node_gemm_ok(int64_t num_rels, int64_t H, int64_t K, int64_t D);
int het_node_row_map(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* nodes, int64_t num_rows, int64_t num_nodes,
                     int32_t* map, het_stream stream);
int het_rgat_node_backward_dx(int64_t n_begin, int64_t n_end, int64_t n_loop, int64_t num_nodes, int64_t num_rels,
                              const float* grad_h, const float* loop_wt, const float* g_rows, const float* weights_t,
                              const int32_t* row_map, const float* g_er, const float* wa_t, const int32_t* dst_map,
                              float* grad_x, int64_t H, int64_t K, int64_t D, const int32_t* node_order, het_stream stream);

/* Layer-level extension (no reference op of its own): the node-major sum of row x weight products
 *     out[n, :] = SUM_s rows_s[map_s[n], :] . weights_t[s]          n = node_order[p] (or p) for p in [n_begin, n_end)
 * -- the input gradient of a layer input that feeds several projections, formed in ONE pass over the nodes instead of one
 * read-modify-write (the reference: float-atomic) pass per projection: replaces the a2 / a3 input-gradient launches behind the
 * HGT layer (backward_rgnn_relational_matmul / ..._no_scatter_gather_list, OpExport/RGNNOps.inc.h:946-1010, 660-753, as
 * HGT/models.py:159-262 composes them).  Source s: rows[s] = its first row (a column offset into a wider row is part of the
 * pointer: the k' and the m half of HGT's [S_row, 2X] gradient are two sources), row_strides[s] floats between rows,
 * maps[s] [num_nodes] int32 = row of node n or -1 (het_node_row_map gives [R, N] of them), or NULL: row = n for
 * n < ident_rows[s]; weights_t[s] [KS, XO] row-major (the projection's weight transposed).  Every node of the range is
 * written (zeros where it has no row).  KS, XO in {32, 64}; all weights of a call stay in LDS: at most 9 sources and
 * het_node_rows_matmul_sum_ok(num_sources, KS, XO).  node_order as in het_rgat_node_backward_dx. */
int het_node_rows_matmul_sum_ok(int64_t num_sources, int64_t KS, int64_t XO);
int het_node_rows_matmul_sum(int64_t n_begin, int64_t n_end, int64_t num_nodes, int64_t num_sources,
                             const float* const* rows, const int64_t* row_strides, const int32_t* const* maps,
                             const int64_t* ident_rows, const float* const* weights_t, float* out, int64_t KS, int64_t XO,
                             const int32_t* node_order, het_stream stream);

/* the same with a bias row: out[n, :] = bias[:] + SUM_s ...   (bias [XO] or NULL; a node without any row gets the bias) */
int het_node_rows_matmul_sum_bias(int64_t n_begin, int64_t n_end, int64_t num_nodes, int64_t num_sources,
                                  const float* const* rows, const int64_t* row_strides, const int32_t* const* maps,
                                  const int64_t* ident_rows, const float* const* weights_t, const float* bias, float* out,
                                  int64_t KS, int64_t XO, const int32_t* node_order, het_stream stream);

/* ------------------------------------------------------------------------
 * The RGCN layer as two calls (layer-level fusion; no reference op of its own).  Inside het_amd/backend/rgcn_layers_and_funcs.py
 * it replaces the pair a7 / a8 (rgcn_layer1_separate_coo / backward_rgcn_layer1_separate_coo, OpExport/RGCNOps.inc.h:84-138,
 * 368-467) together with the layer's "node_repr + h_bias" (RGCN/RGCN.py:338-340) and the fills the
 * reference wrappers make around the ops (rgcn_layers_and_funcs.py:481-601).  Same values:
 *   forward   ssum[(r,v), :] = SUM over the in-edges e of v in relation r of norm[eid_e] * x[src_e, :]       (kept for the backward)
 *             ret[v, :]      = bias[:] + SUM_r ssum[(r,v), :] . W[r]          one pass over the nodes, every row stored once
 *   backward  gsum[(r,u), :] = SUM over the out-edges e of u in relation r of norm[eid_e] * gradout[dst_e, :]
 *             grad_x[u, :]   = SUM_r gsum[(r,u), :] . Wt[r]                   one pass over the nodes, every row stored once
 *             grad_w[r]      = SUM over the (r,v) rows of ssum[(r,v), :]^T (x) gradout[v, :];   grad_bias = column sums of gradout
 *             (grad_w and grad_bias on the library's side stream beside the gather pass; joined before the call returns)
 * by_rel_dst = het_grouping_create(rel_ptrs, R, col, E, N_dst, payload0 = row, payload1 = eids), by_rel_src = the same with row
 * and col exchanged (the groupings of a7 / a8).  dst_map / src_map [R, N] int32: segment of (relation, node) in that grouping =
 * its row in the sorted unique (relation, node) list, -1 = none (het_grouping_segment_map).  node_order: optional, as in
 * het_rgat_node_backward_dx.  norm [E] by edge id, or norm_sorted [E] in the order of the call's gather grouping (by_rel_dst forward,
 * by_rel_src backward: het_grouping_gather_payload1; ogbn-mag: the forward's gather pass 0.85 -> see DESIGN.md 4.5) -- one of the two
 * may be NULL.  ssum [by_rel_dst segments, K];  weights [R,K,D];  weights_t [R,D,K];  bias / grad_bias [D] or NULL.
 * Shapes: K, D in {32, 64}, all R weights resident in LDS (het_rgcn_layer_ok); HET_ERR_INVALID_ARG otherwise -- callers use a7 / a8.
 * workspace: het_rgcn_layer_backward_workspace(segments of by_rel_src, D) bytes, 16-byte aligned.  grad_x NULL: the layer input
 * needs no gradient (fixed features): only grad_w / grad_bias are formed -- no gather pass. */
int het_rgcn_layer_ok(int64_t num_rels, int64_t K, int64_t D);
int64_t het_rgcn_layer_backward_workspace(int64_t n_src_rows, int64_t D);
int het_rgcn_layer_forward(const het_grouping* by_rel_dst, int64_t num_rels, int64_t num_nodes, const float* x,
                           const float* weights, const float* norm, const float* norm_sorted, const float* bias,
                           const int32_t* dst_map, const int32_t* node_order, float* ssum, float* ret, int64_t K, int64_t D,
                           het_stream stream);
int het_rgcn_layer_backward(const het_grouping* by_rel_src, const het_grouping* by_rel_dst, int64_t num_rels,
                            int64_t num_src_nodes, int64_t num_dst_nodes, const float* ssum, const float* weights_t,
                            const float* norm, const float* norm_sorted, const float* gradout, const int32_t* src_map,
                            const int32_t* node_order, float* grad_x, float* grad_w, float* grad_bias, int64_t K, int64_t D, void* workspace,
                            int64_t workspace_bytes, het_stream stream);

/* self-loop + bias of a layer as one pass (RGAT/models.py:378-381: h + th.matmul(inputs_dst, loop_weight) + h_bias):
 * out[i,:] = x[i,:] . w + bias for rows [offsets[0], offsets[1]) (offsets: device array), w [K,X], bias [X] or NULL.
 * Matrix-core shapes only (HET_ERR_UNSUPPORTED otherwise). */
int het_rows_linear_bias(const int64_t* offsets, const float* x, const float* w, const float* bias, float* out,
                         int64_t num_rows, int64_t K, int64_t X, het_stream stream);

/* layer epilogue (RGAT/models.py:377-383: h + loop_message + h_bias): out[i,:] = a[i,:] (+ b[i,:]) (+ bias[:]) in one
 * pass; b and bias optional, X % 4 == 0 */
int het_rows_add_bias(const float* a, const float* b, const float* bias, float* out, int64_t num_rows, int64_t X,
                      het_stream stream);

/* Halo pack / unpack of the multi-GPU path (no reference counterpart: het_amd/dist.py pushes the features of remote source
 * nodes with an all-to-all before the forward and returns their gradients after the backward).
 *   het_rows_gather:       out[i, :] = x[idx[i], :]        i in [0, num_rows)   (X % 4 == 0, 16-byte aligned rows)
 *   het_rows_scatter_add:  out[idx[i], :] += src[i, :]     (atomic: idx may repeat) */
int het_rows_gather(const float* x, const int64_t* idx, int64_t num_rows, int64_t X, float* out, het_stream stream);
int het_rows_scatter_add(const float* src, const int64_t* idx, int64_t num_rows, int64_t X, float* out, het_stream stream);
/*   het_rows_scatter_add_grouped: the same sum without float atomics (and deterministic) for an idx that is used every step:
 *   by_idx = het_grouping_create(NULL, 0, idx, n, out_rows, payload0 = 0 .. n-1, NULL); X a power of two in 4 .. 256. */
int het_rows_scatter_add_grouped(const het_grouping* by_idx, const float* src, int64_t X, float* out, int64_t out_rows,
                                 het_stream stream);

/* ------------------------------------------------------------------------
 * Layout builders (the step before the path).  Device-side replacements of the reference's CPU
 * converters: torch.ops.torch_hrt.convert_integrated_{coo,csr}_to_separate_{coo,csr} / transpose_csr
 * (OpExport/DataConverters.inc.h:10-344 over MyHyb/MyHyb.h:1047-1150) and the Python unique-list builders
 * (hrt/python/utils_lite/mydgl_graph_methods.py:10-157).  int64 device arrays in and out, outputs caller-allocated;
 * one stable radix sort per call; the call returns after the stream has drained (temporaries are freed).
 * ------------------------------------------------------------------------ */
/* edges bucketed by relation (out_rel_ptrs [R+1]), inside a bucket sorted by eid (< eid_bound), ties in input order */
int het_layout_separate_coo(const int64_t* row, const int64_t* col, const int64_t* rel, const int64_t* eids,
                            int64_t num_edges, int64_t num_rels, int64_t eid_bound, int64_t* out_rel_ptrs,
                            int64_t* out_row, int64_t* out_col, int64_t* out_eids, het_stream stream);
/* integrated COO -> CSR over `row` (stable); out_row_ptrs [num_rows+1] */
int het_layout_coo_to_csr(const int64_t* row, const int64_t* col, const int64_t* rel, const int64_t* eids,
                          int64_t num_edges, int64_t num_rows, int64_t* out_row_ptrs, int64_t* out_col, int64_t* out_rel,
                          int64_t* out_eids, het_stream stream);
/* transpose_csr: rows of the result are the columns of the input; out_row_ptrs [num_cols+1] */
int het_layout_transpose_csr(const int64_t* row_ptrs, const int64_t* col, const int64_t* eids, const int64_t* rel,
                             int64_t num_rows, int64_t num_edges, int64_t num_cols, int64_t* out_row_ptrs, int64_t* out_col,
                             int64_t* out_eids, int64_t* out_rel, het_stream stream);
/* sorted unique (relation, node) pairs of a relation-bucketed list: out_nodes (capacity E, or 2E with nodes_b),
 * out_rel_ptrs [R+1], out_inverse (optional; [E], or [2E] for the dual list in the reference's order: per relation the
 * entries of nodes_a then those of nodes_b), *out_count (host) = number of pairs */
int het_layout_unique_rel_nodes(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* nodes_a, const int64_t* nodes_b,
                                int64_t num_edges, int64_t num_nodes, int64_t* out_nodes, int64_t* out_rel_ptrs,
                                int64_t* out_inverse, int64_t* out_count, het_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* HET_AMD_H */
