// C ABI of the typed linear projections (segment GEMM ops) and the fused RGCN layer.
#include "seg_gemm.hip.h"
#include "seg_gemm_mfma.hip.h"
#include "seg_gemm_any.hip.h"
#include "seg_rowdot.hip.h"
#include "seg_reduce.hip.h"

namespace {

int check_matmul(const char* op, int64_t kind, const int64_t* rel_ptrs, int64_t num_rels, const int64_t* gather,
                 const int64_t* scatter, int64_t num_rows, int64_t H, int64_t K, int64_t D) {
  if (kind != HET_KIND_DISABLED && kind != HET_KIND_ENABLED) {
    het_set_error("%s: CompactAsOfNodeKind %lld not supported (the reference asserts, RGNNOps.inc.h:292-294)", op,
                  (long long)kind);
    return HET_ERR_UNSUPPORTED;
  }
  HET_REQUIRE(num_rels > 0 && num_rows >= 0 && H > 0 && K > 0 && D > 0, "%s: bad sizes", op);
  HET_REQUIRE(H * K < (1ll << 31) && H * D < (1ll << 31) && num_rels < (1ll << 31), "%s: dims too large", op);
  HET_REQUIRE(rel_ptrs && (num_rows == 0 || gather), "%s: null index pointer", op);
  HET_REQUIRE(kind == HET_KIND_ENABLED || num_rows == 0 || scatter, "%s: kind 0 needs separate_coo_eids", op);
  return HET_OK;
}

}  // namespace

extern "C" int het_rgnn_relational_matmul(int64_t kind, const int64_t* rel_ptrs, int64_t num_rels,
                                          const int64_t* gather_idx, const int64_t* scatter_idx, int64_t num_rows,
                                          const float* weights, const float* x, float* ret, int64_t H, int64_t K,
                                          int64_t D, int in1head, const het_grouping* by_rel_gather, void* workspace,
                                          int64_t workspace_bytes, het_stream stream) {
  const char* op = "rgnn_relational_matmul";
  if (H == 1) in1head = 1;  // one head: x [*, 1, K] and x [*, K] are the same rows -- take the paths of the shared-input form
  if (int rc = check_matmul(op, kind, rel_ptrs, num_rels, gather_idx, scatter_idx, num_rows, H, K, D)) return rc;
  HET_REQUIRE(num_rows == 0 || (weights && x && ret), "%s: null data pointer", op);
  hipStream_t s = (hipStream_t)stream;
  const idx_t* scatter = kind == HET_KIND_ENABLED ? nullptr : scatter_idx;
  if (in1head && by_rel_gather && kind == HET_KIND_DISABLED && by_rel_gather->R == (int)num_rels &&
      by_rel_gather->E == num_rows && by_rel_gather->p0 && by_rel_gather->S > 0 && D > 1 && segment_rows_supported((int)(H * D)) &&
      workspace && workspace_bytes >= (int64_t)sizeof(float) * by_rel_gather->S * H * D &&
      ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(ret)) & 15) == 0) {
    // the caller passed the (relation, x row) grouping (het_amd.kernels does for shapes outside the 32..128 MFMA tiles: an
    // 8- or 16-wide output layer, 256-wide features): project only the S distinct rows and duplicate them, instead of
    // one GEMM row per position
    const het_grouping* g = by_rel_gather;
    float* comp = static_cast<float*>(workspace);
    MfmaGemmArgs m;
    m.A = x; m.a_ld = K; m.gather = g->seg_key64; m.B = weights; m.b_rel_stride = H * K * D; m.b_headcat = 1;
    m.headcat_d = (int)D; m.C = comp; m.c_ld = H * D; m.seg_ptrs = g->seg_rel_ptr64; m.num_segs = (int)num_rels;
    m.num_rows = g->S; m.K = (int)K; m.X = (int)(H * D);
    if (int rc = launch_rows_gemm(m, s)) return rc;
    return launch_segment_broadcast(g, comp, ret, (int)(H * D), nullptr, nullptr, 0, s);
  }
  if (in1head && mfma_fwd_supported((int)K, (int)(H * D)) && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    return launch_seg_gemm_mfma_fwd(x, K, gather_idx, weights, H * K * D, (int)H, (int)D, ret, H * D, scatter, rel_ptrs,
                                    (int)num_rels, num_rows, (int)K, s);
  if (!in1head && D == 1 && rowdot_supported((int)H, (int)K) && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    RowDotArgs q;
    q.A = x; q.gather = gather_idx; q.W = weights; q.out = ret; q.scatter = scatter; q.seg_ptrs = rel_ptrs;
    q.num_segs = (int)num_rels; q.num_rows = num_rows; q.H = (int)H; q.K = (int)K;
    return launch_rowdot_fwd(q, s);
  }
  if (in1head && D == 1 && rowdot1h_supported((int)H, (int)K) && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    RowDotArgs q;  // one x row against the H folded attention vectors of its relation
    q.A = x; q.gather = gather_idx; q.W = weights; q.out = ret; q.scatter = scatter; q.seg_ptrs = rel_ptrs;
    q.num_segs = (int)num_rels; q.num_rows = num_rows; q.H = (int)H; q.K = (int)K;
    const het_grouping* g = by_rel_gather;
    if (g && kind == HET_KIND_DISABLED && g->R == (int)num_rels && g->E == num_rows && g->p0 && g->S > 0 &&
        segment_rows_supported((int)H) && workspace && workspace_bytes >= (int64_t)sizeof(float) * g->S * H &&
        (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(ret) & 15) == 0) {
      // positions sharing (relation, x row) hold the same [H] product: form the S distinct ones, then duplicate
      float* dots = static_cast<float*>(workspace);
      q.gather = g->seg_key64; q.scatter = nullptr; q.out = dots; q.seg_ptrs = g->seg_rel_ptr64; q.num_rows = g->S;
      if (int rc = launch_rowdot1h_fwd(q, s)) return rc;
      return launch_segment_broadcast(g, dots, ret, (int)H, nullptr, nullptr, 0, s);
    }
    return launch_rowdot1h_fwd(q, s);
  }
  if (!in1head && H > 1 && mfma_shape_supported((int)(H * K), (int)(H * D)) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(ret) & 15) == 0) {
    // per-head K x D products (HGT: 8 heads of 8 x 8) as ONE row GEMM with a block-diagonal weight: the
    // kernel is bound by moving the [rows, H*K] / [rows, H*D] tensors, not by the (mostly zero) MFMA work
    MfmaGemmArgs m;
    m.A = x; m.a_ld = H * K; m.gather = gather_idx; m.B = weights; m.b_rel_stride = H * K * D; m.b_headcat = 2;
    m.headcat_d = (int)D; m.blockdiag_k = (int)K; m.C = ret; m.c_ld = H * D; m.scatter = scatter;
    m.seg_ptrs = rel_ptrs; m.num_segs = (int)num_rels; m.num_rows = num_rows; m.K = (int)(H * K); m.X = (int)(H * D);
    return launch_seg_gemm_mfma(m, s);
  }
  SegGemmArgs a;
  a.A = x; a.gather = gather_idx; a.B = weights; a.C = ret; a.scatter = scatter;
  a.seg_ptrs = rel_ptrs; a.num_segs = (int)num_rels; a.num_rows = num_rows; a.KA = (int)K;
  a.b_rel_stride = H * K * D; a.c_ld = H * D;
  if (in1head) {
    a.a_ld = K; a.b_headcat = 1; a.headcat_d = (int)D; a.NB = (int)(H * D); a.heads_z = 1;
  } else {
    a.a_ld = H * K; a.a_head_stride = K; a.b_head_stride = K * D; a.c_head_stride = D; a.NB = (int)D; a.heads_z = (int)H;
  }
  return launch_seg_gemm(a, s);
}

extern "C" int het_rgnn_relational_matmul_attn_dot(int64_t kind, const int64_t* rel_ptrs, int64_t num_rels,
                                                   const int64_t* gather_idx, const int64_t* scatter_idx,
                                                   int64_t num_rows, const float* weights, const float* x, float* ret,
                                                   const float* dot_w, float* dot_out, int64_t H, int64_t K, int64_t D,
                                                   const het_grouping* by_rel_gather, void* workspace,
                                                   int64_t workspace_bytes, float* comp_rows,
                                                   const het_grouping* dot_grouping, het_stream stream) {
  const char* op = "rgnn_relational_matmul_attn_dot";
  if (int rc = check_matmul(op, kind, rel_ptrs, num_rels, gather_idx, scatter_idx, num_rows, H, K, D)) return rc;
  HET_REQUIRE(num_rows == 0 || (weights && x && dot_w && dot_out), "%s: null data pointer", op);
  if (num_rows == 0) return HET_OK;
  if (!(mfma_fwd_supported((int)K, (int)(H * D)) && D >= 4 && (D & (D - 1)) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(dot_w) & 15) == 0)) {
    het_set_error("%s: only the MFMA shapes (K, H*D in {32, 64, 128}, D a power of two >= 4, 16-byte aligned rows)", op);
    return HET_ERR_UNSUPPORTED;
  }
  MfmaGemmArgs a;
  a.A = x; a.a_ld = K; a.B = weights; a.b_rel_stride = H * K * D; a.b_headcat = 1; a.headcat_d = (int)D;
  a.K = (int)K; a.X = (int)(H * D); a.num_segs = (int)num_rels; a.dot_w = dot_w;
  const het_grouping* g = by_rel_gather;
  const int64_t X = H * D;
  const int64_t ws_need = g ? (int64_t)sizeof(float) * g->S * ((comp_rows ? 0 : X) + H) : 0;
  if (g && kind == HET_KIND_DISABLED && g->R == (int)num_rels && g->E == num_rows && g->p0 && segment_sum_supported((int)X) &&
      H <= X / 4 && workspace && workspace_bytes >= ws_need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(ret) & 15) == 0 && (reinterpret_cast<uintptr_t>(comp_rows) & 15) == 0 && num_rows > 0 &&
      segment_rows_supported((int)H)) {
    // Rows that share (relation, gather_idx) are identical: project the S distinct rows once (dense, into comp_rows
    // or the workspace), then duplicate every row to the positions of its segment -- same values as the per-position
    // GEMM.  ret == NULL: the caller wants the attention term only; just the [S,H] dots are duplicated.
    float* ws = static_cast<float*>(workspace);
    float* comp = comp_rows ? comp_rows : ws;
    float* comp_dot = comp_rows ? ws : ws + g->S * X;
    a.gather = g->seg_key64; a.C = comp; a.c_ld = X; a.scatter = nullptr; a.seg_ptrs = g->seg_rel_ptr64; a.num_rows = g->S;
    a.dot_out = comp_dot;
    if (int rc = launch_seg_gemm_mfma(a, (hipStream_t)stream)) return rc;
    // dot_grouping: the same (relation, gather_idx) segments with another payload0 -- where the dots go
    const het_grouping* gd = dot_grouping ? dot_grouping : g;
    HET_REQUIRE(gd->S == g->S && gd->E == g->E && gd->p0, "%s: dot_grouping must group the same keys", op);
    if (!ret) return launch_segment_broadcast(gd, comp_dot, dot_out, (int)H, nullptr, nullptr, 0, (hipStream_t)stream);
    // rows and dots as two launches: the 16-byte dot stores interleaved with the row stores of one launch cost more than
    // a second pass over the index streams (same-box A/B: 2.64 -> 2.43 ms for the two projections of a step)
    if (int rc = launch_segment_broadcast(g, comp, ret, (int)X, nullptr, nullptr, 0, (hipStream_t)stream)) return rc;
    return launch_segment_broadcast(gd, comp_dot, dot_out, (int)H, nullptr, nullptr, 0, (hipStream_t)stream);
  }
  HET_REQUIRE(ret && !comp_rows && !dot_grouping, "%s: ret == NULL / comp_rows / dot_grouping need the (relation, gather_idx) grouping path", op);
  a.gather = gather_idx; a.C = ret; a.c_ld = X; a.scatter = kind == HET_KIND_ENABLED ? nullptr : scatter_idx;
  a.seg_ptrs = rel_ptrs; a.num_rows = num_rows; a.dot_out = dot_out;
  return launch_seg_gemm_mfma(a, (hipStream_t)stream);
}

namespace {
// G[s, h, :] = gs[s, h] * attn[r(s), h, :] for the S (relation, node) segments of a grouping (r from seg_rel_ptr)
__global__ __launch_bounds__(256) void HET_expand_rank1(const float* __restrict__ gs, const float* __restrict__ attn,
                                                        const idx_t* __restrict__ seg_rel_ptr, int R, int64_t S, int H, int D,
                                                        float* __restrict__ G) {
  const int64_t X4 = (int64_t)H * D / 4, total = S * X4;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t sidx = t / X4;
    const int x = (int)(t - sidx * X4) * 4, h = x / D;
    const int r = find_segment(seg_rel_ptr, R, sidx);
    const float g = gs[sidx * H + h];
    const float4 a = *reinterpret_cast<const float4*>(attn + (int64_t)r * H * D + x);
    *reinterpret_cast<float4*>(G + sidx * (int64_t)H * D + x) = make_float4(g * a.x, g * a.y, g * a.z, g * a.w);
  }
}
}  // namespace

// Backward of het_rgnn_relational_matmul_attn_dot for a caller that used ONLY dot_out: the gradient of the projection
// output is then grad_dot (x) dot_w[r] -- rank one per head -- so rows sharing (relation, gather_idx) are summed over the
// [rows, H] gradient (not over an [rows, H, D] tensor that would first have to be written and read back):
//   gs[(r,v), h] = SUM grad_dot[scatter_idx[i], h];  G[(r,v), h, :] = gs * dot_w[r, h, :]
//   grad_x[v] (+)= G . Wt[r];   grad_w[r] (+)= x[v]^T (x) G
extern "C" int het_backward_rgnn_relational_matmul_attn_dot_only(
    const int64_t* rel_ptrs, int64_t num_rels, const int64_t* gather_idx, const int64_t* scatter_idx, int64_t num_rows,
    int64_t num_x_rows, const float* weights_t, const float* x, const float* dot_w, const float* grad_dot, float* grad_x,
    float* grad_w, int64_t H, int64_t K, int64_t D, int accumulate, const het_grouping* by_rel_gather, void* workspace,
    int64_t workspace_bytes, const float* comp_rows, float* grad_dot_w, het_stream stream) {
  const char* op = "backward_rgnn_relational_matmul_attn_dot_only";
  if (int rc = check_matmul(op, HET_KIND_DISABLED, rel_ptrs, num_rels, gather_idx, scatter_idx, num_rows, H, K, D)) return rc;
  HET_REQUIRE(num_rows == 0 || (weights_t && x && dot_w && grad_dot && grad_x && grad_w), "%s: null data pointer", op);
  const het_grouping* g = by_rel_gather;
  const int64_t X = H * D;
  if (!(g && g->R == (int)num_rels && g->E == num_rows && g->p0 && segment_rows_supported((int)H) && (D % 4) == 0 &&
        mfma_shape_supported((int)X, (int)K) && mfma_dw_supported((int)K, (int)X) && workspace &&
        workspace_bytes >= (int64_t)sizeof(float) * g->S * (H + X) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(grad_dot) & 15) == 0 && (reinterpret_cast<uintptr_t>(dot_w) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0)) {
    het_set_error("%s: needs the (relation, gather_idx) grouping, a workspace of S*(H + H*D) floats and MFMA shapes", op);
    return HET_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    HET_HIP(hipMemsetAsync(grad_w, 0, sizeof(float) * num_rels * H * K * D, s));
    HET_HIP(hipMemsetAsync(grad_x, 0, sizeof(float) * num_x_rows * K, s));
  }
  if (num_rows == 0) return HET_OK;
  float* gs = static_cast<float*>(workspace);
  float* G = gs + ((g->S * H + 3) / 4) * 4;
  if (int rc = launch_segment_sum(g, grad_dot, gs, (int)H, nullptr, s)) return rc;
  if (comp_rows && grad_dot_w) {
    // grad_dot_w[r, h, :] (+)= SUM over the segments of r of gs[s, h] * comp_rows[s, h, :]  -- the weight gradient of
    // the attention product from the S distinct projected rows of the forward instead of the E duplicated ones
    HET_REQUIRE(rowdot_supported((int)H, (int)D) && (reinterpret_cast<uintptr_t>(comp_rows) & 15) == 0,
                "%s: comp_rows path needs the row-dot shapes", op);
    if (!accumulate) HET_HIP(hipMemsetAsync(grad_dot_w, 0, sizeof(float) * num_rels * X, s));
    RowDotArgs q;
    q.A = comp_rows; q.go = gs; q.out = grad_dot_w; q.seg_ptrs = g->seg_rel_ptr64; q.num_segs = (int)num_rels;
    q.num_rows = g->S; q.H = (int)H; q.K = (int)D;
    if (int rc = launch_rowdot_bwd_dw(q, s)) return rc;
  }
  const int64_t total = g->S * (X / 4);
  int64_t nb = ceil_div64(total, 256);
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(HET_expand_rank1, dim3((unsigned)nb), dim3(256), 0, s, gs, dot_w, g->seg_rel_ptr64, (int)num_rels, g->S,
                     (int)H, (int)D, G);
  HET_LAUNCH_CHECK("HET_expand_rank1");
  MfmaGemmArgs m;
  m.A = G; m.a_ld = X; m.B = weights_t; m.b_rel_stride = X * K; m.C = grad_x; m.c_ld = K; m.scatter = g->seg_key64; m.atomic = 1;
  m.seg_ptrs = g->seg_rel_ptr64; m.num_segs = (int)num_rels; m.num_rows = g->S; m.K = (int)X; m.X = (int)K;
  if (int rc = launch_seg_gemm_mfma(m, s)) return rc;
  MfmaDwArgs w;
  w.A = x; w.a_ld = K; w.gather = g->seg_key64; w.G = G; w.g_ld = X; w.dW = grad_w; w.dw_rel_stride = H * K * D;
  w.headcat = 1; w.headcat_d = (int)D; w.seg_ptrs = g->seg_rel_ptr64; w.num_segs = (int)num_rels; w.num_rows = g->S;
  w.K = (int)K; w.X = (int)X;
  return launch_seg_dw_mfma(w, s);
}

extern "C" int het_backward_rgnn_relational_matmul(int64_t kind, const int64_t* rel_ptrs, int64_t num_rels,
                                                   const int64_t* gather_idx, const int64_t* scatter_idx,
                                                   int64_t num_rows, int64_t num_x_rows, const float* weights_t,
                                                   const float* x, const float* gradout, float* grad_x,
                                                   float* grad_w, int64_t H,
                                                   int64_t K, int64_t D, int in1head, int accumulate,
                                                   const het_grouping* by_rel_gather, void* workspace,
                                                   int64_t workspace_bytes, het_stream stream) {
  const char* op = "backward_rgnn_relational_matmul";
  // HET_ACC_DISTINCT_ROWS: the caller states that gather_idx holds every node at most once per relation (a unique
  // (relation, node) list); only then may kind 1 add its input gradients with plain read-modify-write, relation by
  // relation.  Without the bit -- the reference-named op -- any list is valid: float atomics, as the reference's
  // compact backward (RGNN/my_shmem_sgemm_func.cu.h:711-776).
  const bool distinct_rows = (accumulate & HET_ACC_DISTINCT_ROWS) != 0;
  accumulate &= HET_ACC_ADD;
  if (H == 1 && D > 1) in1head = 1;  // as in the forward (the D == 1 row-dot shapes keep their own per-head kernels)
  if (int rc = check_matmul(op, kind, rel_ptrs, num_rels, gather_idx, scatter_idx, num_rows, H, K, D)) return rc;
  HET_REQUIRE(num_rows == 0 || (weights_t && x && gradout && grad_w), "%s: null data pointer", op);
  hipStream_t s = (hipStream_t)stream;
  const idx_t* scatter = kind == HET_KIND_ENABLED ? nullptr : scatter_idx;
  HET_REQUIRE(accumulate || num_x_rows >= 0, "%s: num_x_rows needed to overwrite grad_x", op);
  // same condition as the reference's ACGatherScatterListIdentical dispatch (RGNNOps.inc.h:253): the
  // gather list IS the edge-id list, so every input row belongs to exactly one position
  const bool unique = kind == HET_KIND_DISABLED && gather_idx == scatter_idx;
  const bool rowdot = !in1head && D == 1 && rowdot_supported((int)H, (int)K) &&
                      (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0;
  // grad_x == NULL: weight gradient only (a caller that folded the input gradient elsewhere); row-dot shape only
  const bool rowdot1h = in1head && D == 1 && rowdot1h_supported((int)H, (int)K) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0;
  HET_REQUIRE(num_rows == 0 || grad_x || rowdot || rowdot1h, "%s: grad_x may be NULL only for the D == 1 shapes", op);
  if (!accumulate) {  // "=" semantics: zero what the kernels below accumulate into
    HET_HIP(hipMemsetAsync(grad_w, 0, sizeof(float) * num_rels * H * K * D, s));
    // rows of x that no position gathers must read zero; with a unique row-dot list of all rows every
    // row is stored exactly once instead
    if (grad_x && !(rowdot && unique && num_rows == num_x_rows))
      HET_HIP(hipMemsetAsync(grad_x, 0, sizeof(float) * num_x_rows * (in1head ? K : H * K), s));
  }
  if (rowdot) {
    RowDotArgs q;
    q.A = x; q.gather = gather_idx; q.W = weights_t; q.scatter = scatter; q.go = gradout; q.seg_ptrs = rel_ptrs;
    q.num_segs = (int)num_rels; q.num_rows = num_rows; q.H = (int)H; q.K = (int)K;
    q.unique_rows = unique;
    q.overwrite = !accumulate && unique && num_rows == num_x_rows;
    q.out = grad_x;
    if (grad_x)
      if (int rc = launch_rowdot_bwd_dx(q, s)) return rc;
    q.out = grad_w;
    return launch_rowdot_bwd_dw(q, s);
  }
  const het_grouping* g = by_rel_gather;
  if (rowdot1h) {
    RowDotArgs q;
    q.A = x; q.gather = gather_idx; q.W = weights_t; q.scatter = scatter; q.go = gradout; q.seg_ptrs = rel_ptrs;
    q.num_segs = (int)num_rels; q.num_rows = num_rows; q.H = (int)H; q.K = (int)K;
    if (g && kind == HET_KIND_DISABLED && g->R == (int)num_rels && g->E == num_rows && g->p0 &&
        segment_rows_supported((int)H) && workspace && workspace_bytes >= (int64_t)sizeof(float) * g->S * H &&
        (reinterpret_cast<uintptr_t>(gradout) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
      // positions sharing (relation, x row) share both factors: sum their [H] gradients first
      float* gsum = static_cast<float*>(workspace);
      if (int rc = launch_segment_sum(g, gradout, gsum, (int)H, nullptr, s)) return rc;
      q.gather = g->seg_key64; q.scatter = nullptr; q.go = gsum; q.seg_ptrs = g->seg_rel_ptr64; q.num_rows = g->S;
    }
    q.out = grad_x;
    if (!grad_x) {
      // weight gradient only (the caller formed the input gradient elsewhere: csrc/node_gemm.hip)
    } else if (kind == HET_KIND_ENABLED && distinct_rows && num_rels <= kRmwMaxSegments) {
      // rows of one relation are distinct nodes: relation by relation, plain read-modify-write instead of atomics
      for (int r = 0; r < (int)num_rels; ++r) {
        RowDotArgs qr = q;
        qr.seg_ptrs = q.seg_ptrs + r; qr.num_segs = 1; qr.W = q.W + (int64_t)r * H * K; qr.rmw = 1;
        if (int rc = launch_rowdot1h_bwd_dx(qr, s)) return rc;
      }
    } else {
      if (int rc = launch_rowdot1h_bwd_dx(q, s)) return rc;
    }
    q.out = grad_w;
    return launch_rowdot1h_bwd_dw(q, s);
  }
  if (!in1head && H > 1 && mfma_shape_supported((int)(H * D), (int)(H * K)) && mfma_dw_supported((int)(H * K), (int)(H * D)) &&
      (reinterpret_cast<uintptr_t>(gradout) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    // per-head products as block-diagonal row GEMMs (see the forward)
    const bool grouped = g && kind == HET_KIND_DISABLED && g->R == (int)num_rels && g->E == num_rows && g->p0 &&
                         segment_sum_supported((int)(H * D)) && workspace &&
                         workspace_bytes >= (int64_t)sizeof(float) * g->S * H * D &&
                         (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
    const float* G = gradout;
    const idx_t *g_rows = scatter, *x_rows = gather_idx, *segs = rel_ptrs;
    int64_t rows = num_rows;
    if (grouped) {
      float* gsum = static_cast<float*>(workspace);
      if (int rc = launch_segment_sum(g, gradout, gsum, (int)(H * D), nullptr, s, 0, -1, 0, 0, /*nt_in=*/0)) return rc;
      G = gsum; g_rows = nullptr; x_rows = g->seg_key64; segs = g->seg_rel_ptr64; rows = g->S;
    }
    MfmaGemmArgs m;
    m.A = G; m.a_ld = H * D; m.gather = g_rows; m.B = weights_t; m.b_rel_stride = H * D * K; m.b_headcat = 2;
    m.headcat_d = (int)K; m.blockdiag_k = (int)D; m.C = grad_x; m.c_ld = H * K; m.scatter = x_rows; m.atomic = 1;
    m.seg_ptrs = segs; m.num_segs = (int)num_rels; m.num_rows = rows; m.K = (int)(H * D); m.X = (int)(H * K);
    if (int rc = launch_seg_gemm_mfma(m, s)) return rc;
    MfmaDwArgs w;
    w.A = x; w.a_ld = H * K; w.gather = x_rows; w.G = G; w.g_ld = H * D; w.g_gather = g_rows;
    w.dW = grad_w; w.dw_rel_stride = H * K * D; w.headcat = 2; w.headcat_d = (int)D; w.blockdiag_k = (int)K;
    w.seg_ptrs = segs; w.num_segs = (int)num_rels; w.num_rows = rows; w.K = (int)(H * K); w.X = (int)(H * D);
    return launch_seg_dw_mfma(w, s);
  }
  if (in1head && g && kind == HET_KIND_DISABLED && g->R == (int)num_rels && g->E == num_rows && g->p0 &&
      segment_sum_supported((int)(H * D)) && workspace && workspace_bytes >= (int64_t)sizeof(float) * g->S * H * D &&
      (reinterpret_cast<uintptr_t>(gradout) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
    // Rows that share (relation, gather_idx) share x row and weight: by linearity sum their gradout rows
    // first (one pass over gradout), then run both GEMMs on the S distinct (relation, node) rows only.
    float* gsum = static_cast<float*>(workspace);
    if (int rc = launch_segment_sum(g, gradout, gsum, (int)(H * D), nullptr, s, 0, -1, 0, 0, /*nt_in=*/0)) return rc;
    MfmaGemmArgs m;
    m.A = gsum; m.a_ld = H * D; m.B = weights_t; m.b_rel_stride = H * D * K;
    m.C = grad_x; m.c_ld = K; m.scatter = g->seg_key64; m.atomic = 1;
    m.seg_ptrs = g->seg_rel_ptr64; m.num_segs = (int)num_rels; m.num_rows = g->S; m.K = (int)(H * D); m.X = (int)K;
    if (int rc = launch_rows_gemm(m, s)) return rc;  // matrix cores for their shapes, LDS-tiled FMA otherwise
    MfmaDwArgs w;
    w.A = x; w.a_ld = K; w.gather = g->seg_key64; w.G = gsum; w.g_ld = H * D;
    w.dW = grad_w; w.dw_rel_stride = H * K * D; w.headcat = 1; w.headcat_d = (int)D;
    w.seg_ptrs = g->seg_rel_ptr64; w.num_segs = (int)num_rels; w.num_rows = g->S; w.K = (int)K; w.X = (int)(H * D);
    return launch_rows_dw(w, s);
  }
  if (in1head && mfma_shape_supported((int)(H * D), (int)K) && mfma_dw_supported((int)K, (int)(H * D)) &&
      (reinterpret_cast<uintptr_t>(gradout) & 15) == 0) {
    // grad_x[gather] += gradout[scatter] . Wt[r]: Wt[r] read as one [H*D, K] matrix (heads summed by the GEMM)
    MfmaGemmArgs m;
    m.A = gradout; m.a_ld = H * D; m.gather = scatter; m.B = weights_t; m.b_rel_stride = H * D * K;
    m.C = grad_x; m.c_ld = K; m.scatter = gather_idx; m.atomic = 1;
    m.seg_ptrs = rel_ptrs; m.num_segs = (int)num_rels; m.num_rows = num_rows; m.K = (int)(H * D); m.X = (int)K;
    // kind 1: a relation's rows are distinct nodes (its unique list) -- relation by relation the gradient rows are added
    // with plain read-modify-write instead of float atomics (0.52 -> 0.3 ms for the 2.4 M rows of ogbn-mag)
    if (kind == HET_KIND_ENABLED && distinct_rows && num_rels <= kRmwMaxSegments && H * D <= 128 && K <= 128) {
      if (int rc = launch_seg_gemm_mfma_rmw_per_segment(m, s)) return rc;
    } else {
      if (int rc = launch_seg_gemm_mfma(m, s)) return rc;
    }
    MfmaDwArgs w;
    w.A = x; w.a_ld = K; w.gather = gather_idx; w.G = gradout; w.g_ld = H * D; w.g_gather = scatter;
    w.dW = grad_w; w.dw_rel_stride = H * K * D; w.headcat = 1; w.headcat_d = (int)D;
    w.seg_ptrs = rel_ptrs; w.num_segs = (int)num_rels; w.num_rows = num_rows; w.K = (int)K; w.X = (int)(H * D);
    return launch_seg_dw_mfma(w, s);
  }
  // grad_x[gather] += gradout[scatter] . Wt[r]
  SegGemmArgs a;
  a.A = gradout; a.gather = scatter; a.B = weights_t; a.C = grad_x; a.scatter = gather_idx; a.atomic = 1;
  a.seg_ptrs = rel_ptrs; a.num_segs = (int)num_rels; a.num_rows = num_rows;
  a.b_rel_stride = H * D * K; a.a_ld = H * D; a.NB = (int)K;
  if (in1head) {
    a.KA = (int)(H * D); a.c_ld = K; a.heads_z = 1;  // Wt[r] read as one [H*D, K] matrix: heads summed
  } else {
    a.KA = (int)D; a.a_head_stride = D; a.b_head_stride = D * K; a.c_ld = H * K; a.c_head_stride = K; a.heads_z = (int)H;
  }
  // kind 1: a relation's rows are distinct nodes (its unique list) -- relation by relation with plain read-modify-write
  if (kind == HET_KIND_ENABLED && distinct_rows && num_rels <= kRmwMaxSegments) {
    if (int rc = launch_seg_gemm_rmw_per_segment(a, s)) return rc;
  } else {
    if (int rc = launch_seg_gemm(a, s)) return rc;
  }
  // grad_w[r,h] += x[gather]^T (x) gradout[scatter]
  SegDwArgs w;
  w.A = x; w.gather = gather_idx; w.G = gradout; w.g_gather = scatter; w.dW = grad_w;
  w.seg_ptrs = rel_ptrs; w.num_segs = (int)num_rels; w.num_rows = num_rows;
  w.dw_rel_stride = H * K * D; w.g_ld = H * D; w.KA = (int)K;
  if (in1head) {
    w.a_ld = K; w.headcat = 1; w.headcat_d = (int)D; w.NB = (int)(H * D); w.heads_z = 1;
  } else {
    w.a_ld = H * K; w.a_head_stride = K; w.g_head_stride = D; w.dw_head_stride = K * D; w.NB = (int)D; w.heads_z = (int)H;
  }
  return launch_seg_dw(w, s);
}

// The two halves of a2 (one input head, matrix-core shapes) as separate entry points, so that a caller can order them
// around a collective (het_amd/dist.py: the input gradient of the halo rows leaves first, the weight gradient is formed
// while the reverse all-to-all is in flight).  gather_idx NULL: row i of x / grad_x; g_rows NULL: row i of gradout.
extern "C" int het_rows_matmul_backward_dx(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* gather_idx,
                                           const int64_t* g_rows, int64_t num_rows, const float* weights_t,
                                           const float* gradout, float* grad_x, int64_t H, int64_t K, int64_t D,
                                           int atomic, het_stream stream) {
  const char* op = "het_rows_matmul_backward_dx";
  HET_REQUIRE(rel_ptrs && num_rels > 0 && num_rows >= 0 && H > 0 && K > 0 && D > 0, "%s: bad arguments", op);
  if (num_rows == 0) return HET_OK;
  HET_REQUIRE(weights_t && gradout && grad_x, "%s: null data pointer", op);
  if (!(mfma_shape_supported((int)(H * D), (int)K) && (reinterpret_cast<uintptr_t>(gradout) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0)) {
    het_set_error("%s: only the matrix-core shapes", op);
    return HET_ERR_UNSUPPORTED;
  }
  MfmaGemmArgs m;  // grad_x[gather] (+)= gradout[g_rows] . Wt[r]: Wt[r] read as one [H*D, K] matrix (heads summed by the GEMM)
  m.A = gradout; m.a_ld = H * D; m.gather = g_rows; m.B = weights_t; m.b_rel_stride = H * D * K;
  m.C = grad_x; m.c_ld = K; m.scatter = gather_idx; m.atomic = atomic ? 1 : 0;
  m.seg_ptrs = rel_ptrs; m.num_segs = (int)num_rels; m.num_rows = num_rows; m.K = (int)(H * D); m.X = (int)K;
  if (atomic == 2 && num_rels <= kRmwMaxSegments && H * D <= 128 && K <= 128)  // rows distinct inside every relation
    return launch_seg_gemm_mfma_rmw_per_segment(m, (hipStream_t)stream);
  return launch_seg_gemm_mfma(m, (hipStream_t)stream);
}

extern "C" int het_rows_matmul_backward_dw_colsum(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* gather_idx,
                                           const int64_t* g_rows, int64_t num_rows, const float* x, const float* gradout,
                                           float* grad_w, float* colsum, int64_t H, int64_t K, int64_t D,
                                                  int accumulate, het_stream stream) {
  const char* op = "het_rows_matmul_backward_dw_colsum";
  HET_REQUIRE(rel_ptrs && num_rels > 0 && num_rows >= 0 && H > 0 && K > 0 && D > 0 && grad_w, "%s: bad arguments", op);
  hipStream_t s = (hipStream_t)stream;
  if (!(mfma_dw_supported((int)K, (int)(H * D)) && (reinterpret_cast<uintptr_t>(gradout) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
    het_set_error("%s: only the matrix-core shapes", op);
    return HET_ERR_UNSUPPORTED;
  }
  if (!accumulate) HET_HIP(hipMemsetAsync(grad_w, 0, sizeof(float) * num_rels * H * K * D, s));
  if (colsum) HET_HIP(hipMemsetAsync(colsum, 0, sizeof(float) * H * D, s));  // always "=": the sums of this launch's rows
  if (num_rows == 0) return HET_OK;
  HET_REQUIRE(x && gradout, "%s: null data pointer", op);
  MfmaDwArgs w;
  w.A = x; w.a_ld = K; w.gather = gather_idx; w.G = gradout; w.g_ld = H * D; w.g_gather = g_rows;
  w.dW = grad_w; w.colsum = colsum; w.dw_rel_stride = H * K * D; w.headcat = 1; w.headcat_d = (int)D;
  w.seg_ptrs = rel_ptrs; w.num_segs = (int)num_rels; w.num_rows = num_rows; w.K = (int)K; w.X = (int)(H * D);
  return launch_seg_dw_mfma(w, s);
}

extern "C" int het_rows_matmul_backward_dw(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* gather_idx,
                                           const int64_t* g_rows, int64_t num_rows, const float* x, const float* gradout,
                                           float* grad_w, int64_t H, int64_t K, int64_t D, int accumulate,
                                           het_stream stream) {
  return het_rows_matmul_backward_dw_colsum(rel_ptrs, num_rels, gather_idx, g_rows, num_rows, x, gradout, grad_w, nullptr, H, K, D,
                                            accumulate, stream);
}

extern "C" int het_rgnn_relational_matmul_no_scatter_gather_list(const int64_t* offsets, int64_t num_types,
                                                                 int64_t num_rows, const float* weights,
                                                                 const float* x, float* ret, int64_t H, int64_t K,
                                                                 int64_t D, int x_per_head, het_stream stream) {
  const char* op = "rgnn_relational_matmul_no_scatter_gather_list";
  HET_REQUIRE(num_types > 0 && num_rows >= 0 && H > 0 && K > 0 && D > 0 && offsets, "%s: bad arguments", op);
  HET_REQUIRE(num_rows == 0 || (weights && x && ret), "%s: null data pointer", op);
  hipStream_t s = (hipStream_t)stream;
  if (!x_per_head && mfma_fwd_supported((int)K, (int)(H * D)) && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    return launch_seg_gemm_mfma_fwd(x, K, nullptr, weights, H * K * D, (int)H, (int)D, ret, H * D, nullptr, offsets,
                                    (int)num_types, num_rows, (int)K, s);
  if ((x_per_head || H == 1) && D == 1 && rowdot_supported((int)H, (int)K) && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    RowDotArgs q;
    q.A = x; q.W = weights; q.out = ret; q.seg_ptrs = offsets; q.num_segs = (int)num_types; q.num_rows = num_rows;
    q.H = (int)H; q.K = (int)K;
    return launch_rowdot_fwd(q, s);
  }
  SegGemmArgs a;
  a.A = x; a.B = weights; a.C = ret; a.seg_ptrs = offsets; a.num_segs = (int)num_types; a.num_rows = num_rows;
  a.KA = (int)K; a.b_rel_stride = H * K * D; a.c_ld = H * D;
  if (!x_per_head) {
    a.a_ld = K; a.b_headcat = 1; a.headcat_d = (int)D; a.NB = (int)(H * D); a.heads_z = 1;
  } else {
    a.a_ld = H * K; a.a_head_stride = K; a.b_head_stride = K * D; a.c_head_stride = D; a.NB = (int)D; a.heads_z = (int)H;
  }
  return launch_seg_gemm(a, s);
}

namespace {
// rows outside [offsets[0], offsets[T]) of a [num_rows, X] tensor (normally none)
__global__ __launch_bounds__(256) void HET_zero_rows_outside(const idx_t* __restrict__ offsets, int T, int64_t num_rows,
                                                             float* __restrict__ out, int X) {
  const idx_t lo = offsets[0], hi = offsets[T];
  const int64_t head = lo < num_rows ? lo : num_rows, tail = hi < num_rows ? num_rows - (hi > 0 ? hi : 0) : 0;
  const int64_t total = (head + tail) * X;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t rowi = t / X, c = t - rowi * X;
    const int64_t rr = rowi < head ? rowi : hi + (rowi - head);
    out[rr * X + c] = 0.f;
  }
}
}  // namespace

// out[i, :] = x[i, :] . W + bias   for the rows [offsets[0], offsets[1])  -- the self-loop + bias term of a layer as ONE
// pass (bias added in the GEMM epilogue).  HET_ERR_UNSUPPORTED outside the matrix-core shapes (the caller then uses
// rgnn_relational_matmul_no_scatter_gather_list + het_rows_add_bias).
extern "C" int het_rows_linear_bias(const int64_t* offsets, const float* x, const float* w, const float* bias, float* out,
                                    int64_t num_rows, int64_t K, int64_t X, het_stream stream) {
  const char* op = "het_rows_linear_bias";
  HET_REQUIRE(offsets && num_rows >= 0 && K > 0 && X > 0, "%s: bad arguments", op);
  if (num_rows == 0) return HET_OK;
  HET_REQUIRE(x && w && out, "%s: null data pointer", op);
  if (!(mfma_shape_supported((int)K, (int)X) && K <= 128 && X <= 128 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) {
    het_set_error("%s: only the matrix-core shapes (K, X in {32, 64, 128}, 16-byte aligned rows)", op);
    return HET_ERR_UNSUPPORTED;
  }
  MfmaGemmArgs a;
  a.A = x; a.a_ld = K; a.B = w; a.b_rel_stride = K * X; a.C = out; a.c_ld = X; a.seg_ptrs = offsets; a.num_segs = 1;
  a.num_rows = num_rows; a.K = (int)K; a.X = (int)X; a.bias = bias;
  return launch_seg_gemm_mfma(a, (hipStream_t)stream);
}

extern "C" int het_backward_rgnn_relational_matmul_no_scatter_gather_list(
    const int64_t* offsets, int64_t num_types, int64_t num_rows, const float* weights_t, const float* x,
    const float* gradout, float* grad_x, float* grad_w, int64_t H, int64_t K, int64_t D, int x_per_head,
    int accumulate, het_stream stream) {
  const char* op = "backward_rgnn_relational_matmul_no_scatter_gather_list";
  HET_REQUIRE(num_types > 0 && num_rows >= 0 && H > 0 && K > 0 && D > 0 && offsets, "%s: bad arguments", op);
  HET_REQUIRE(num_rows == 0 || (weights_t && x && gradout && grad_w), "%s: null data pointer", op);
  hipStream_t s = (hipStream_t)stream;
  // rows are their own gather list here: every grad_x row has exactly one writer (rows outside
  // [offsets[0], offsets[T]) have none: they are zeroed when overwriting)
  const bool rowdot = (x_per_head || H == 1) && D == 1 && rowdot_supported((int)H, (int)K) &&
                      (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0;
  const bool mfma = grad_x && !x_per_head && mfma_shape_supported((int)(H * D), (int)K) && mfma_dw_supported((int)K, (int)(H * D)) &&
                    (reinterpret_cast<uintptr_t>(gradout) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0;
  // grad_x == NULL: weight gradient only (per-head D == 1 shape, as in het_backward_rgnn_relational_matmul)
  HET_REQUIRE(num_rows == 0 || grad_x || rowdot, "%s: grad_x may be NULL only for the per-head D == 1 shape", op);
  if (!accumulate) {
    HET_HIP(hipMemsetAsync(grad_w, 0, sizeof(float) * num_types * H * K * D, s));
    // (rows outside [offsets[0], offsets[T]) have no writer and must read zero: on the matrix-core path, whose stores
    // overwrite every row of a segment, only those rows are cleared -- a full memset of grad_x cost 60 us per call on ogbn-mag)
    if (grad_x && !mfma) HET_HIP(hipMemsetAsync(grad_x, 0, sizeof(float) * num_rows * (x_per_head ? H * K : K), s));
    if (grad_x && mfma && num_rows > 0) {
      hipLaunchKernelGGL(HET_zero_rows_outside, dim3(256), dim3(256), 0, s, offsets, (int)num_types, num_rows, grad_x, (int)K);
      HET_LAUNCH_CHECK("HET_zero_rows_outside");
    }
  }
  if (mfma) {
    MfmaGemmArgs m;  // grad_x[i] (+)= gradout[i] . Wt[t]; plain stores would do, atomics give "+=" on both paths
    m.A = gradout; m.a_ld = H * D; m.B = weights_t; m.b_rel_stride = H * D * K; m.C = grad_x; m.c_ld = K;
    m.atomic = accumulate ? 1 : 0;  // every row has one writer: plain stores when overwriting
    m.seg_ptrs = offsets; m.num_segs = (int)num_types; m.num_rows = num_rows; m.K = (int)(H * D); m.X = (int)K;
    if (int rc = launch_seg_gemm_mfma(m, s)) return rc;
    MfmaDwArgs w;
    w.A = x; w.a_ld = K; w.G = gradout; w.g_ld = H * D; w.dW = grad_w; w.dw_rel_stride = H * K * D;
    w.headcat = 1; w.headcat_d = (int)D; w.seg_ptrs = offsets; w.num_segs = (int)num_types; w.num_rows = num_rows;
    w.K = (int)K; w.X = (int)(H * D);
    return launch_seg_dw_mfma(w, s);
  }
  if (rowdot) {
    RowDotArgs q;
    q.A = x; q.W = weights_t; q.go = gradout; q.seg_ptrs = offsets; q.num_segs = (int)num_types; q.num_rows = num_rows;
    q.H = (int)H; q.K = (int)K; q.unique_rows = 1;
    q.out = grad_x;
    if (grad_x)
      if (int rc = launch_rowdot_bwd_dx(q, s)) return rc;
    q.out = grad_w;
    return launch_rowdot_bwd_dw(q, s);
  }
  SegGemmArgs a;  // rows are disjoint (their own gather list): plain stores when overwriting, plain "+=" when accumulating
  a.A = gradout; a.B = weights_t; a.C = grad_x; a.atomic = accumulate ? 2 : 0;
  a.seg_ptrs = offsets; a.num_segs = (int)num_types; a.num_rows = num_rows;
  a.b_rel_stride = H * D * K; a.a_ld = H * D; a.NB = (int)K;
  if (!x_per_head) {
    a.KA = (int)(H * D); a.c_ld = K; a.heads_z = 1;
  } else {
    a.KA = (int)D; a.a_head_stride = D; a.b_head_stride = D * K; a.c_ld = H * K; a.c_head_stride = K; a.heads_z = (int)H;
  }
  if (int rc = launch_seg_gemm(a, s)) return rc;
  SegDwArgs w;
  w.A = x; w.G = gradout; w.dW = grad_w; w.seg_ptrs = offsets; w.num_segs = (int)num_types; w.num_rows = num_rows;
  w.dw_rel_stride = H * K * D; w.g_ld = H * D; w.KA = (int)K;
  if (!x_per_head) {
    w.a_ld = K; w.headcat = 1; w.headcat_d = (int)D; w.NB = (int)(H * D); w.heads_z = 1;
  } else {
    w.a_ld = H * K; w.a_head_stride = K; w.g_head_stride = D; w.dw_head_stride = K * D; w.NB = (int)D; w.heads_z = (int)H;
  }
  return launch_seg_dw(w, s);
}

// ---- fused RGCN layer --------------------------------------------------------------------
extern "C" int het_rgcn_layer1_separate_coo(const int64_t* rel_ptrs, const int64_t* eids, const int64_t* row,
                                            const int64_t* col, int64_t num_rels, int64_t num_edges,
                                            int64_t num_nodes, const float* x, const float* weights,
                                            const float* norm, float* ret, int64_t K, int64_t D,
                                            const het_grouping* by_rel_dst, void* workspace, int64_t workspace_bytes,
                                            het_stream stream) {
  const char* op = "rgcn_layer1_separate_coo";
  HET_REQUIRE(num_rels > 0 && num_edges >= 0 && num_nodes >= 0 && K > 0 && D > 0, "%s: bad sizes", op);
  HET_REQUIRE(rel_ptrs && (num_edges == 0 || (eids && row && col && x && weights && norm && ret)), "%s: null pointer", op);
  const het_grouping* g = by_rel_dst;
  if (g && g->R == (int)num_rels && g->E == num_edges && g->p0 && g->p1 && segment_sum_supported((int)K) &&
      workspace && workspace_bytes >= (int64_t)sizeof(float) * g->S * K &&
      (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(ret) & 15) == 0) {
    // Edges that share (relation, destination) share weight and output row: sum their scaled source rows first
    // (one gather pass over x), then one GEMM row per distinct (relation, destination) pair, added into ret.
    hipStream_t s = (hipStream_t)stream;
    float* ssum = static_cast<float*>(workspace);
    if (int rc = launch_segment_sum(g, x, ssum, (int)K, norm, s)) return rc;
    MfmaGemmArgs m;
    m.A = ssum; m.a_ld = K; m.B = weights; m.b_rel_stride = K * D; m.C = ret; m.c_ld = D; m.scatter = g->seg_key64;
    m.atomic = 1; m.seg_ptrs = g->seg_rel_ptr64; m.num_segs = (int)num_rels; m.num_rows = g->S; m.K = (int)K; m.X = (int)D;
    return launch_rows_gemm_add_unique(m, s);  // the destinations of a relation's segments are distinct
  }
  SegGemmArgs a;
  a.A = x; a.a_ld = K; a.gather = row; a.row_scale = norm; a.scale_idx = eids;
  a.B = weights; a.b_rel_stride = K * D; a.C = ret; a.c_ld = D; a.scatter = col; a.atomic = 1;
  a.seg_ptrs = rel_ptrs; a.num_segs = (int)num_rels; a.num_rows = num_edges; a.KA = (int)K; a.NB = (int)D;
  return launch_seg_gemm(a, (hipStream_t)stream);
}

extern "C" int het_backward_rgcn_layer1_separate_coo(const int64_t* rel_ptrs, const int64_t* eids, const int64_t* row,
                                                     const int64_t* col, int64_t num_rels, int64_t num_edges,
                                                     int64_t num_nodes, const float* x, const float* weights_t,
                                                     const float* norm, float* grad_norm, float* grad_x,
                                                     const float* gradout, float* grad_w, int64_t K, int64_t D,
                                                     const het_grouping* by_rel_src, void* workspace,
                                                     int64_t workspace_bytes, het_stream stream) {
  const char* op = "backward_rgcn_layer1_separate_coo";
  HET_REQUIRE(num_rels > 0 && num_edges >= 0 && num_nodes >= 0 && K > 0 && D > 0, "%s: bad sizes", op);
  HET_REQUIRE(rel_ptrs && (num_edges == 0 || (eids && row && col && x && weights_t && norm && grad_x && gradout && grad_w)),
              "%s: null pointer", op);
  (void)grad_norm;
  hipStream_t s = (hipStream_t)stream;
  const het_grouping* g = by_rel_src;
  if (g && g->R == (int)num_rels && g->E == num_edges && g->p0 && g->p1 && segment_sum_supported((int)D) &&
      workspace && workspace_bytes >= (int64_t)sizeof(float) * g->S * D && (reinterpret_cast<uintptr_t>(gradout) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0) {
    // gsum[(r,u)] = SUM over the out-edges of u in relation r of norm * gradout[dst]; then
    //   grad_x[u] += gsum[(r,u)] . Wt[r]      and      grad_w[r] += x[u]^T (x) gsum[(r,u)]
    float* gsum = static_cast<float*>(workspace);
    if (int rc = launch_segment_sum(g, gradout, gsum, (int)D, norm, s)) return rc;
    MfmaGemmArgs m;
    m.A = gsum; m.a_ld = D; m.B = weights_t; m.b_rel_stride = D * K; m.C = grad_x; m.c_ld = K; m.scatter = g->seg_key64;
    m.atomic = 1; m.seg_ptrs = g->seg_rel_ptr64; m.num_segs = (int)num_rels; m.num_rows = g->S; m.K = (int)D; m.X = (int)K;
    if (int rc = launch_rows_gemm_add_unique(m, s)) return rc;
    MfmaDwArgs w;
    w.A = x; w.a_ld = K; w.gather = g->seg_key64; w.G = gsum; w.g_ld = D; w.dW = grad_w; w.dw_rel_stride = K * D;
    w.seg_ptrs = g->seg_rel_ptr64; w.num_segs = (int)num_rels; w.num_rows = g->S; w.K = (int)K; w.X = (int)D;
    return launch_rows_dw(w, s);
  }
  SegGemmArgs a;  // grad_x[row] += (gradout[col] * norm) . Wt[r]
  a.A = gradout; a.a_ld = D; a.gather = col; a.row_scale = norm; a.scale_idx = eids;
  a.B = weights_t; a.b_rel_stride = D * K; a.C = grad_x; a.c_ld = K; a.scatter = row; a.atomic = 1;
  a.seg_ptrs = rel_ptrs; a.num_segs = (int)num_rels; a.num_rows = num_edges; a.KA = (int)D; a.NB = (int)K;
  if (int rc = launch_seg_gemm(a, s)) return rc;
  SegDwArgs w;  // grad_w[r] += (x[row] * norm)^T (x) gradout[col]
  w.A = x; w.a_ld = K; w.gather = row; w.row_scale = norm; w.scale_idx = eids;
  w.G = gradout; w.g_ld = D; w.g_gather = col; w.dW = grad_w; w.dw_rel_stride = K * D;
  w.seg_ptrs = rel_ptrs; w.num_segs = (int)num_rels; w.num_rows = num_edges; w.KA = (int)K; w.NB = (int)D;
  return launch_seg_dw(w, s);
}

// ---- the RGCN layer as two calls (layer-level fusion of a7 / a8 with the layer's bias; include/het_amd.h) -------------------
namespace {
// column sums of a [rows, 4 * LPR] matrix: per-workgroup partial rows (grid-stride, 16 bytes per lane), finished by HET_colsum_finish
template <int LPR>
__global__ __launch_bounds__(256) void HET_colsum_partial(const float* __restrict__ in, int64_t rows, float* __restrict__ part) {
  constexpr int RPI = 256 / LPR;
  __shared__ float4 red[256];
  const int c = threadIdx.x % LPR, r = threadIdx.x / LPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * RPI + r; i < rows; i += (int64_t)gridDim.x * RPI) {
    const float4 v = *reinterpret_cast<const float4*>(in + i * (4 * LPR) + 4 * c);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < RPI; ++k) {
      const float4 v = red[k * LPR + c];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * (4 * LPR) + 4 * c) = acc;
  }
}
__global__ __launch_bounds__(256) void HET_colsum_finish(const float* __restrict__ part, int P, int X, float* __restrict__ out) {
  __shared__ float red[256];
  const int x = blockIdx.x, t = threadIdx.x;
  float a = 0.f;
  for (int p = t; p < P; p += 256) a += part[(int64_t)p * X + x];
  red[t] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) out[x] = red[0];
}
constexpr int kColsumBlocks = 1024;

int launch_colsum(const float* in, int64_t rows, int X, float* part, float* out, hipStream_t s) {
  HET_KTIME("HET_colsum", s);
  const int lpr = X / 4;
  const int64_t rpi = 256 / lpr;
  int blocks = (int)(ceil_div64(rows, rpi) < kColsumBlocks ? ceil_div64(rows, rpi) : kColsumBlocks);
  if (blocks < 1) blocks = 1;
  switch (lpr) {
    case 8: hipLaunchKernelGGL(HET_colsum_partial<8>, dim3(blocks), dim3(256), 0, s, in, rows, part); break;
    case 16: hipLaunchKernelGGL(HET_colsum_partial<16>, dim3(blocks), dim3(256), 0, s, in, rows, part); break;
    default: het_set_error("colsum: %d columns unsupported", X); return HET_ERR_UNSUPPORTED;
  }
  HET_LAUNCH_CHECK("HET_colsum_partial");
  hipLaunchKernelGGL(HET_colsum_finish, dim3(X), dim3(256), 0, s, part, blocks, X, out);
  HET_LAUNCH_CHECK("HET_colsum_finish");
  return HET_OK;
}

bool rgcn_layer_shape_ok(int64_t R, int64_t K, int64_t D) {
  return (K == 32 || K == 64) && (D == 32 || D == 64) && R >= 1 && het_node_rows_matmul_sum_ok(R, K, D) &&
         het_node_rows_matmul_sum_ok(R, D, K) && segment_sum_supported((int)K) && segment_sum_supported((int)D) &&
         mfma_dw_supported((int)K, (int)D);
}
}  // namespace

extern "C" int het_grouping_gather_payload1(const het_grouping* g, const float* values, int64_t H, float* out, het_stream stream) {
  HET_REQUIRE(g && (g->E == 0 || (values && out)) && H >= 1 && H <= 1024, "het_grouping_gather_payload1: bad arguments");
  return launch_gather_by_p1(g, values, (int)H, out, (hipStream_t)stream);
}

extern "C" int het_rgcn_layer_ok(int64_t num_rels, int64_t K, int64_t D) { return rgcn_layer_shape_ok(num_rels, K, D) ? 1 : 0; }

extern "C" int64_t het_rgcn_layer_backward_workspace(int64_t n_src_rows, int64_t D) {
  return (int64_t)sizeof(float) * ((n_src_rows > 0 ? n_src_rows : 1) * D + (int64_t)kColsumBlocks * D);
}

extern "C" int het_rgcn_layer_forward(const het_grouping* by_rel_dst, int64_t num_rels, int64_t num_nodes, const float* x,
                                      const float* weights, const float* norm, const float* norm_sorted, const float* bias,
                                      const int32_t* dst_map, const int32_t* node_order, float* ssum, float* ret, int64_t K,
                                      int64_t D, het_stream stream) {
  const char* op = "het_rgcn_layer_forward";
  const het_grouping* g = by_rel_dst;
  HET_REQUIRE(g && g->R == (int)num_rels && g->p0 && g->p1, "%s: needs the grouping by (relation, destination) with payloads (source row, edge id)", op);
  HET_REQUIRE(rgcn_layer_shape_ok(num_rels, K, D), "%s: unsupported shape (het_rgcn_layer_ok)", op);
  HET_REQUIRE(num_nodes >= 0 && num_nodes < (1ll << 31), "%s: bad node count", op);
  if (num_nodes == 0) return HET_OK;
  HET_REQUIRE(x && weights && (norm || norm_sorted) && dst_map && ssum && ret, "%s: null pointer", op);
  HET_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ssum) | reinterpret_cast<uintptr_t>(ret) |
                reinterpret_cast<uintptr_t>(bias)) & 15) == 0, "%s: 16-byte aligned pointers expected", op);
  hipStream_t s = (hipStream_t)stream;
  // ssum[(r,v), :] = SUM over the in-edges of v in relation r of norm * x[src]   (one gather pass over x)
  // (norm_sorted: the norm in the grouping's order, het_grouping_gather_payload1 -- a coalesced stream instead of a random 4-byte
  //  gather per edge: the forward visits the edges by destination, their ids are scattered)
  if (g->E > 0)
    if (int rc = launch_segment_sum(g, x, ssum, (int)K, norm_sorted ? norm_sorted : norm, s, 0, -1, 0, 0, 0, norm_sorted ? 1 : 0)) return rc;
  // ret[v, :] = bias + SUM_r ssum[(r,v), :] . W[r]: one pass over the nodes, every output row stored once
  const float* rows[16]; int64_t strides[16]; const int32_t* maps[16]; int64_t ident[16]; const float* wts[16];
  for (int r = 0; r < (int)num_rels; ++r) {
    rows[r] = ssum; strides[r] = K; maps[r] = dst_map + (int64_t)r * num_nodes; ident[r] = 0; wts[r] = weights + (int64_t)r * K * D;
  }
  return het_node_rows_matmul_sum_bias(0, num_nodes, num_nodes, num_rels, rows, strides, maps, ident, wts, bias, ret, K, D, node_order, stream);
}

extern "C" int het_rgcn_layer_backward(const het_grouping* by_rel_src, const het_grouping* by_rel_dst, int64_t num_rels,
                                       int64_t num_src_nodes, int64_t num_dst_nodes, const float* ssum, const float* weights_t,
                                       const float* norm, const float* norm_sorted, const float* gradout,
                                       const int32_t* src_map, const int32_t* node_order, float* grad_x, float* grad_w,
                                       float* grad_bias, int64_t K, int64_t D, void* workspace, int64_t workspace_bytes,
                                       het_stream stream) {
  const char* op = "het_rgcn_layer_backward";
  const het_grouping *gs = by_rel_src, *gd = by_rel_dst;
  HET_REQUIRE(gs && gd && gs->R == (int)num_rels && gd->R == (int)num_rels && gs->E == gd->E && gs->p0 && gs->p1,
              "%s: needs the groupings by (relation, source) [payloads: destination row, edge id] and by (relation, destination)", op);
  HET_REQUIRE(rgcn_layer_shape_ok(num_rels, K, D), "%s: unsupported shape (het_rgcn_layer_ok)", op);
  HET_REQUIRE(num_src_nodes >= 0 && num_src_nodes < (1ll << 31) && num_dst_nodes >= gd->key_bound, "%s: bad node count", op);
  HET_REQUIRE(grad_w && (num_src_nodes == 0 || (ssum && gradout && (!grad_x || (weights_t && (norm || norm_sorted) && src_map)))), "%s: null pointer", op);
  HET_REQUIRE(workspace && workspace_bytes >= het_rgcn_layer_backward_workspace(gs->S, D) &&
              ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(gradout) | reinterpret_cast<uintptr_t>(grad_x) |
                reinterpret_cast<uintptr_t>(ssum)) & 15) == 0, "%s: workspace too small (het_rgcn_layer_backward_workspace) or pointers not 16-byte aligned", op);
  hipStream_t s = (hipStream_t)stream;
  float* gsum = static_cast<float*>(workspace);
  float* cpart = gsum + (gs->S > 0 ? gs->S : 1) * D;
  // grad_w[r] = SUM over the (r, v) rows of ssum[(r,v)]^T (x) gradout[v] (half as many rows as the (relation, source) form, no second
  // read of x) and the bias gradient: streams of rows, on the library's side stream beside the node pass (HET_RGCN_BWD_FORK: 0 = on
  // the caller's stream, 1 = beside the gather pass, 2 = beside the node pass)
  static const int fork_mode = [] { const char* v = getenv("HET_RGCN_BWD_FORK"); return v ? atoi(v) : 2; }();
  auto weight_gradients = [&](hipStream_t st) -> int {
    HET_HIP(hipMemsetAsync(grad_w, 0, sizeof(float) * num_rels * K * D, st));
    if (gd->S > 0) {
      MfmaDwArgs w;
      w.A = ssum; w.a_ld = K; w.G = gradout; w.g_ld = D; w.g_gather = gd->seg_key64; w.dW = grad_w; w.dw_rel_stride = K * D;
      w.seg_ptrs = gd->seg_rel_ptr64; w.num_segs = (int)num_rels; w.num_rows = gd->S; w.K = (int)K; w.X = (int)D;
      if (int rc = launch_seg_dw_mfma(w, st)) return rc;
    }
    if (grad_bias)
      if (int rc = launch_colsum(gradout, num_dst_nodes, (int)D, cpart, grad_bias, st)) return rc;
    return HET_OK;
  };
  // gsum[(r,u), :] = SUM over the out-edges of u in relation r of norm * gradout[dst]
  // (grad_x NULL: the layer input needs no gradient -- fixed features -- and the gather pass + node pass are skipped)
  auto gather_pass = [&]() -> int {
    if (gs->E > 0 && grad_x)
      return launch_segment_sum(gs, gradout, gsum, (int)D, norm_sorted ? norm_sorted : norm, s, 0, -1, 0, 0, 0, norm_sorted ? 1 : 0);
    return HET_OK;
  };
  // grad_x[u] = SUM_r gsum[(r,u)] . Wt[r]
  auto node_pass = [&]() -> int {
    if (!(num_src_nodes > 0 && grad_x)) return HET_OK;
    const float* rows[16]; int64_t strides[16]; const int32_t* maps[16]; int64_t ident[16]; const float* wts[16];
    for (int r = 0; r < (int)num_rels; ++r) {
      rows[r] = gsum; strides[r] = D; maps[r] = src_map + (int64_t)r * num_src_nodes; ident[r] = 0; wts[r] = weights_t + (int64_t)r * D * K;
    }
    return het_node_rows_matmul_sum_bias(0, num_src_nodes, num_src_nodes, num_rels, rows, strides, maps, ident, wts, nullptr, grad_x, D, K,
                                         node_order, stream);
  };
  if (fork_mode == 0 || !grad_x) {
    if (int rc = weight_gradients(s)) return rc;
    if (int rc = gather_pass()) return rc;
    return node_pass();
  }
  if (fork_mode == 1) {
    HetFork fk(s);
    if (int rc = weight_gradients(fk.side)) return rc;
    if (int rc = gather_pass()) return rc;
    if (int rc = node_pass()) return rc;
    HET_HIP(fk.join());
    return HET_OK;
  }
  if (int rc = gather_pass()) return rc;
  HetFork fk(s);
  if (int rc = weight_gradients(fk.side)) return rc;
  if (int rc = node_pass()) return rc;
  HET_HIP(fk.join());
  return HET_OK;
}

// ---- layer epilogue: out[i, :] = a[i, :] (+ b[i, :]) (+ bias[:]) -- the "h + loop_message + h_bias" of the layers
// (RGAT/models.py:377-383) as one pass instead of two elementwise adds
namespace {
__global__ __launch_bounds__(256) void HET_rows_add_bias(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int64_t total4, int X4) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (int64_t)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(a)[t];
    if (b) {
      const float4 w = reinterpret_cast<const float4*>(b)[t];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    if (bias) {
      const float4 w = reinterpret_cast<const float4*>(bias)[t % X4];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    reinterpret_cast<float4*>(out)[t] = v;
  }
}
}  // namespace

extern "C" int het_rows_add_bias(const float* a, const float* b, const float* bias, float* out, int64_t num_rows, int64_t X,
                                 het_stream stream) {
  HET_REQUIRE(num_rows >= 0 && X > 0 && X % 4 == 0 && (num_rows == 0 || (a && out)), "rows_add_bias: bad arguments");
  HET_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(bias) |
                reinterpret_cast<uintptr_t>(out)) & 15) == 0, "rows_add_bias: 16-byte aligned pointers expected");
  if (num_rows == 0) return HET_OK;
  const int64_t total4 = num_rows * (X / 4);
  int64_t nb = ceil_div64(total4, 256);
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(HET_rows_add_bias, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a, b, bias, out, total4, (int)(X / 4));
  HET_LAUNCH_CHECK("HET_rows_add_bias");
  return HET_OK;
}

// ---- halo pack / unpack of the multi-GPU path (het_amd/dist.py): rows of a [*, X] tensor gathered into a send buffer,
// received gradient rows added back (several ranks may return a gradient for the same row: atomics, one lane per float
// so that an instruction covers whole lines)
namespace {
__global__ __launch_bounds__(256) void HET_rows_gather(const float* __restrict__ x, const idx_t* __restrict__ idx, int64_t total4,
                                                       int X4, float* __restrict__ out) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (int64_t)gridDim.x * 256) {
    const int64_t i = t / X4;
    const int c = (int)(t - i * X4);
    reinterpret_cast<float4*>(out)[t] = reinterpret_cast<const float4*>(x)[idx[i] * X4 + c];
  }
}
__global__ __launch_bounds__(256) void HET_rows_scatter_add(const float* __restrict__ src, const idx_t* __restrict__ idx,
                                                            int64_t total, int X, float* __restrict__ out) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t i = t / X;
    atomicAdd(out + idx[i] * X + (t - i * X), src[t]);
  }
}
}  // namespace

extern "C" int het_rows_gather(const float* x, const int64_t* idx, int64_t num_rows, int64_t X, float* out, het_stream stream) {
  HET_REQUIRE(num_rows >= 0 && X > 0 && X % 4 == 0 && (num_rows == 0 || (x && idx && out)), "rows_gather: bad arguments");
  HET_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "rows_gather: 16-byte aligned pointers expected");
  if (num_rows == 0) return HET_OK;
  const int64_t total4 = num_rows * (X / 4);
  int64_t nb = ceil_div64(total4, 256);
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(HET_rows_gather, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, idx, total4, (int)(X / 4), out);
  HET_LAUNCH_CHECK("HET_rows_gather");
  return HET_OK;
}

// The same sum without float atomics: by_idx groups the positions of idx by their value (payload0 = the position), so a
// destination row's contributions are summed in registers and added to it once (a row of a partition goes to up to 7 peers and
// comes back as many times: 0.14 -> 0.05 ms for 670 K returned rows of 64 floats).  Deterministic.
extern "C" int het_rows_scatter_add_grouped(const het_grouping* by_idx, const float* src, int64_t X, float* out, int64_t out_rows,
                                            het_stream stream) {
  HET_REQUIRE(by_idx && out_rows >= 0 && X > 0, "rows_scatter_add_grouped: bad arguments");
  if (by_idx->E == 0) return HET_OK;
  HET_REQUIRE(src && out && by_idx->R == 0 && by_idx->p0 && by_idx->key_bound <= out_rows,
              "rows_scatter_add_grouped: by_idx = het_grouping_create(NULL, 0, idx, n, out_rows, positions 0 .. n-1, NULL)");
  if (!segment_sum_supported((int)X)) { het_set_error("rows_scatter_add_grouped: rows of 4 .. 256 floats, a power of two"); return HET_ERR_UNSUPPORTED; }
  return launch_segment_sum(by_idx, src, out, (int)X, nullptr, (hipStream_t)stream, 0, out_rows, /*accumulate=*/1);
}

extern "C" int het_rows_scatter_add(const float* src, const int64_t* idx, int64_t num_rows, int64_t X, float* out, het_stream stream) {
  HET_REQUIRE(num_rows >= 0 && X > 0 && (num_rows == 0 || (src && idx && out)), "rows_scatter_add: bad arguments");
  if (num_rows == 0) return HET_OK;
  const int64_t total = num_rows * X;
  int64_t nb = ceil_div64(total, 256);
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(HET_rows_scatter_add, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, idx, total, (int)X, out);
  HET_LAUNCH_CHECK("HET_rows_scatter_add");
  return HET_OK;
}
