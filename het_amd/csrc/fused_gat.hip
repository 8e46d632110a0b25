// Hetero edge-softmax + weighted aggregation (RGAT), forward and backward.
//
// Two schedules:
//   * edge-parallel with float atomics: needs nothing but the op's own arguments
//     (used when no grouping is supplied, for the CSR twins and for odd shapes);
//   * destination-grouped (fused_gat_grouped.hip): wave-per-destination segmented
//     reduction over a het_grouping by col -- no atomics on ret/sum.
#include "edge_view.hip.h"
#include "fused_gat.hip.h"

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(int64_t total) {
  int64_t b = ceil_div64(total, kBlock);
  const int64_t cap = 256 * 64;  // 64 workgroups per CU, grid-stride beyond
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// exp[eid,h] = leaky_exp(el[srow,h] + er[drow,h]);  sum[dst,h] += exp
__global__ __launch_bounds__(kBlock) void HET_gat_exp_sum_edge(EdgeView v, RowMaps m, const float* __restrict__ el,
                                                                const float* __restrict__ er, float* __restrict__ sum,
                                                                float* __restrict__ exp, int H, float slope) {
  const int64_t total = (int64_t)v.E * H, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / H;
    const int h = (int)(t - i * H);
    const idx_t eid = v.eids[i], s = (m.kind == 1 || m.kind == 3) ? ev_src(v, i) : 0, d = ev_dst(v, i);
    idx_t srow, drow;
    ev_rows(v, m, i, eid, s, d, srow, drow);
    const float e = leaky_exp(el[srow * H + h] + er[drow * H + h], slope);
    exp[eid * H + h] = e;
    atomicAdd(&sum[d * H + h], e);
  }
}

// ret[dst,h,:] += exp[eid,h] / sum[dst,h] * feat[srow,h,:]
__global__ __launch_bounds__(kBlock) void HET_gat_aggregate_edge(EdgeView v, RowMaps m,
                                                                  const float* __restrict__ feat,
                                                                  const float* __restrict__ sum,
                                                                  const float* __restrict__ exp,
                                                                  float* __restrict__ ret, int H, int D) {
  const int X = H * D;
  const int64_t total = (int64_t)v.E * X, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / X;
    const int x = (int)(t - i * X), h = x / D;
    const idx_t eid = v.eids[i], s = (m.kind == 1 || m.kind == 3) ? ev_src(v, i) : 0, d = ev_dst(v, i);
    idx_t srow, drow;
    ev_rows(v, m, i, eid, s, d, srow, drow);
    const float a = exp[eid * H + h] / sum[d * H + h];
    atomicAdd(&ret[d * X + x], a * feat[srow * X + x]);
  }
}

// Backward, one thread per (edge, head, feature).  UNIQUE_ROWS (kind 0): every
// feat/el/er row belongs to exactly one edge, so grad_feat / grad_el / grad_er are
// stored, not accumulated.  WAVE_REDUCE: D is a power of two <= 64, the D partial
// products of a head sit in adjacent lanes and are summed with xor-shuffles.
template <bool UNIQUE_ROWS, bool WAVE_REDUCE>
__global__ __launch_bounds__(kBlock) void HET_gat_backward_edge(
    EdgeView v, RowMaps m, const float* __restrict__ feat, const float* __restrict__ el, const float* __restrict__ er,
    const float* __restrict__ sum, const float* __restrict__ exp, const float* __restrict__ ret,
    const float* __restrict__ gradout, float* __restrict__ grad_feat, float* __restrict__ grad_el,
    float* __restrict__ grad_er, int H, int D, float slope) {
  const int X = H * D;
  const int64_t total = (int64_t)v.E * X, stride = (int64_t)gridDim.x * kBlock;
  const int64_t total_up = (total + 63) / 64 * 64;  // whole waves iterate together (shuffles below)
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total_up; t += stride) {
    const bool valid = t < total;
    float tt = 0.f;
    idx_t srow = 0, drow = 0;
    int h = 0, x = 0;
    if (valid) {
      const idx_t i = t / X;
      x = (int)(t - i * X);
      h = x / D;
      const idx_t eid = v.eids[i], s = (m.kind == 1 || m.kind == 3) ? ev_src(v, i) : 0, d = ev_dst(v, i);
      ev_rows(v, m, i, eid, s, d, srow, drow);
      const float a = exp[eid * H + h] / sum[d * H + h];
      const float g = gradout[d * X + x];
      const float gf = a * g;
      if (UNIQUE_ROWS) grad_feat[srow * X + x] = gf; else atomicAdd(&grad_feat[srow * X + x], gf);
      const float z = el[srow * H + h] + er[drow * H + h];
      tt = g * (feat[srow * X + x] - ret[d * X + x]) * a * (z > 0.f ? 1.f : slope);
    }
    if (WAVE_REDUCE) {
      for (int off = D >> 1; off > 0; off >>= 1) tt += __shfl_xor(tt, off);
      if (valid && (x % D) == 0) {
        if (UNIQUE_ROWS) {
          grad_el[srow * H + h] = tt;
          grad_er[drow * H + h] = tt;
        } else {
          atomicAdd(&grad_el[srow * H + h], tt);
          atomicAdd(&grad_er[drow * H + h], tt);
        }
      }
    } else if (valid) {
      atomicAdd(&grad_el[srow * H + h], tt);
      atomicAdd(&grad_er[drow * H + h], tt);
    }
  }
}

inline bool is_pow2(int64_t x) { return x > 0 && (x & (x - 1)) == 0; }

// CompactAsOfNodeKind 2 (EnabledWithDirectIndexing) in the reference's fused GAT maps BOTH edge ends through the one
// inverse index of the two-sided unique list (RGATKernelsSeparateCOO.cu.h:163-170: the non-dual branch hands the same
// mapper to the source and the destination lookup, and direct indexing ignores the node id): it is kind 4 with
// map_col_a == map_row_a.
void normalize_kind(int64_t& kind, const idx_t*& ra, const idx_t*& ca) {
  if (kind == HET_KIND_DIRECT_INDEX && ra) {
    kind = HET_KIND_DUAL_LIST_DIRECT_INDEX;
    if (!ca) ca = ra;
  }
}

int check_maps(const char* op, int64_t kind, const idx_t* ra, const idx_t* rb, const idx_t* ca, const idx_t* cb) {
  switch (kind) {
    case HET_KIND_DISABLED: return HET_OK;
    case HET_KIND_ENABLED:
    case HET_KIND_DUAL_LIST:
      HET_REQUIRE(ra && rb && ca && cb, "%s: kind %lld needs the unique (relation, node) lists", op, (long long)kind);
      return HET_OK;
    case HET_KIND_DUAL_LIST_DIRECT_INDEX:
      HET_REQUIRE(ra && ca, "%s: kind 4 needs edata_idx_to_inverse_idx_row/_col", op);
      return HET_OK;
    default:
      het_set_error("%s: CompactAsOfNodeKind %lld is not supported (kind 2 needs edata_idx_to_inverse_idx as map_row_a)", op,
                    (long long)kind);
      return HET_ERR_UNSUPPORTED;
  }
}

}  // namespace

int gat_forward_edge(const EdgeView& v, const RowMaps& m, const float* feat, const float* el, const float* er,
                     float* sum, float* exp, float* ret, int H, int D, float slope, hipStream_t s) {
  const int64_t X = (int64_t)H * D;
  HET_HIP(hipMemsetAsync(sum, 0, sizeof(float) * v.N * H, s));
  HET_HIP(hipMemsetAsync(ret, 0, sizeof(float) * v.N * X, s));
  if (v.E == 0) return HET_OK;
  hipLaunchKernelGGL(HET_gat_exp_sum_edge, dim3(grid_for(v.E * H)), dim3(kBlock), 0, s, v, m, el, er, sum, exp, H, slope);
  HET_LAUNCH_CHECK("HET_gat_exp_sum_edge");
  hipLaunchKernelGGL(HET_gat_aggregate_edge, dim3(grid_for(v.E * X)), dim3(kBlock), 0, s, v, m, feat, sum, exp, ret, H, D);
  HET_LAUNCH_CHECK("HET_gat_aggregate_edge");
  return HET_OK;
}

int gat_backward_edge(const EdgeView& v, const RowMaps& m, const float* feat, const float* el, const float* er,
                      const float* sum, const float* exp, const float* ret, const float* gradout, float* grad_feat,
                      float* grad_el, float* grad_er, int H, int D, float slope, hipStream_t s) {
  if (v.E == 0) return HET_OK;
  const int64_t X = (int64_t)H * D;
  const bool unique = m.kind == HET_KIND_DISABLED, wred = is_pow2(D) && D <= 64;
  HET_REQUIRE(grad_el != grad_er || (unique && wred),
              "backward_relational_fused_gat: grad_er may alias grad_el only for kind 0 with a power-of-two D <= 64");
  dim3 grid(grid_for(v.E * X)), block(kBlock);
#define HET_GAT_BWD(U, W)                                                                                        \
  hipLaunchKernelGGL((HET_gat_backward_edge<U, W>), grid, block, 0, s, v, m, feat, el, er, sum, exp, ret, gradout, \
                     grad_feat, grad_el, grad_er, H, D, slope)
  if (unique && wred) {
    HET_GAT_BWD(true, true);
  } else if (unique) {
    // stores for grad_feat, atomics for grad_el/er: those two must start from zero
    HET_HIP(hipMemsetAsync(grad_el, 0, sizeof(float) * v.E * H, s));
    HET_HIP(hipMemsetAsync(grad_er, 0, sizeof(float) * v.E * H, s));
    HET_GAT_BWD(true, false);
  } else if (wred) {
    HET_GAT_BWD(false, true);
  } else {
    HET_GAT_BWD(false, false);
  }
#undef HET_GAT_BWD
  HET_LAUNCH_CHECK("HET_gat_backward_edge");
  return HET_OK;
}

// ------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------
extern "C" int het_relational_fused_gat_separate_coo(
    const int64_t* eids, const int64_t* rel_ptrs, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, int64_t kind, const int64_t* map_row_a, const int64_t* map_row_b,
    const int64_t* map_col_a, const int64_t* map_col_b, const float* feat, const float* el, const float* er,
    float* sum, float* exp, float* ret, float* exp_sorted, int64_t H, int64_t D, double slope,
    const het_grouping* by_dst, const float* el_sorted, const float* er_sorted, het_stream stream) {
  const char* op = "relational_fused_gat_separate_coo";
  HET_REQUIRE(num_edges >= 0 && num_nodes >= 0 && num_rels >= 0 && H > 0 && D > 0, "%s: bad sizes", op);
  HET_REQUIRE(!el_sorted == !er_sorted, "%s: el_sorted and er_sorted come together", op);
  HET_REQUIRE(!el_sorted || (by_dst && exp_sorted && kind == HET_KIND_DISABLED),
              "%s: el_sorted / er_sorted need kind 0, the by_dst grouping and exp_sorted", op);
  HET_REQUIRE(sum && ret && (num_edges == 0 || (eids && rel_ptrs && row && col && feat && (el_sorted || (el && er && exp)))),
              "%s: null pointer", op);
  HET_REQUIRE(num_edges < (1ll << 31) && num_nodes < (1ll << 31), "%s: more than 2^31 edges or nodes", op);
  normalize_kind(kind, map_row_a, map_col_a);
  if (num_edges > 0)  // empty index lists of an edgeless graph arrive as NULL
    if (int rc = check_maps(op, kind, map_row_a, map_row_b, map_col_a, map_col_b)) return rc;
  EdgeView v;
  v.E = num_edges; v.N = num_nodes; v.eids = eids; v.src = row; v.dst = col; v.rel_ptrs = rel_ptrs; v.R = (int)num_rels;
  RowMaps m;
  m.kind = (int)kind; m.ra = map_row_a; m.rb = map_row_b; m.ca = map_col_a; m.cb = map_col_b;
  hipStream_t s = (hipStream_t)stream;
  HET_REQUIRE(!exp_sorted || by_dst, "%s: exp_sorted needs the by_dst grouping", op);
  if (by_dst)
    return gat_forward_grouped(by_dst, v, m, feat, el, er, sum, exp, ret, exp_sorted, (int)H, (int)D, (float)slope,
                               el_sorted, er_sorted, s);
  return gat_forward_edge(v, m, feat, el, er, sum, exp, ret, (int)H, (int)D, (float)slope, s);
}

extern "C" int het_backward_relational_fused_gat_separate_coo(
    const int64_t* eids, const int64_t* rel_ptrs, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, int64_t kind, const int64_t* map_row_a, const int64_t* map_row_b,
    const int64_t* map_col_a, const int64_t* map_col_b, const float* feat, const float* el, const float* er,
    const float* sum, const float* exp, const float* ret, const float* exp_sorted, const float* gradout,
    float* grad_feat, float* grad_el, float* grad_er, int64_t H, int64_t D, double slope, const het_grouping* by_dst,
    const het_grouping* by_src_row, const het_grouping* by_dst_row, int64_t n_src_rows, int64_t n_dst_rows,
    void* workspace, int64_t workspace_bytes, const float* fold_attn_l, float* grad_fold_attn_l,
    const int64_t* fold_row_rel_ptrs, float* grad_el_sorted, het_stream stream) {
  const char* op = "backward_relational_fused_gat_separate_coo";
  HET_REQUIRE(num_edges >= 0 && num_nodes >= 0 && num_rels >= 0 && H > 0 && D > 0, "%s: bad sizes", op);
  // with exp_sorted on the destination-grouped kind-0 path el / er / exp are not read
  const bool streams_only = exp_sorted && by_dst && kind == HET_KIND_DISABLED && slope >= 0;
  HET_REQUIRE(num_edges == 0 || (eids && rel_ptrs && row && col && feat && ((el && er && exp) || streams_only) && sum && ret &&
                                 gradout && grad_feat && ((grad_el && grad_er) || (grad_el_sorted && !grad_el && !grad_er))),
              "%s: null pointer", op);
  HET_REQUIRE(!grad_el_sorted || (by_dst && kind == HET_KIND_DISABLED), "%s: grad_el_sorted needs kind 0 and the by_dst grouping", op);
  HET_REQUIRE(num_edges < (1ll << 31) && num_nodes < (1ll << 31), "%s: more than 2^31 edges or nodes", op);
  normalize_kind(kind, map_row_a, map_col_a);
  if (num_edges > 0)  // empty index lists of an edgeless graph arrive as NULL
    if (int rc = check_maps(op, kind, map_row_a, map_row_b, map_col_a, map_col_b)) return rc;
  EdgeView v;
  v.E = num_edges; v.N = num_nodes; v.eids = eids; v.src = row; v.dst = col; v.rel_ptrs = rel_ptrs; v.R = (int)num_rels;
  RowMaps m;
  m.kind = (int)kind; m.ra = map_row_a; m.rb = map_row_b; m.ca = map_col_a; m.cb = map_col_b;
  hipStream_t s = (hipStream_t)stream;
  if (by_dst && kind == HET_KIND_DISABLED)
    return gat_backward_grouped(by_dst, v, m, feat, el, er, sum, exp, ret, exp_sorted, gradout, grad_feat, grad_el,
                                grad_er, (int)H, (int)D, (float)slope, fold_attn_l, grad_fold_attn_l, grad_el_sorted,
                                static_cast<float*>(workspace), workspace_bytes, s);
  HET_REQUIRE(!grad_fold_attn_l && (kind != HET_KIND_DISABLED || !fold_attn_l),
              "%s: fold_attn_l needs the by_dst grouping (kind 0); grad_fold_attn_l is kind 0 only", op);
  if (kind != HET_KIND_DISABLED && workspace &&
      workspace_bytes >= (int64_t)sizeof(float) * (num_nodes * 2 * H + num_edges * H) &&
      gat_backward_compact_supported(by_src_row, by_dst_row, num_edges, n_dst_rows, (int)H, (int)D, (float)slope))
    return gat_backward_compact_grouped(by_src_row, by_dst_row, v, n_src_rows, n_dst_rows, feat, sum, exp, ret, gradout,
                                        grad_feat, grad_el, grad_er, static_cast<float*>(workspace), (int)H, (int)D,
                                        (float)slope, fold_attn_l, fold_row_rel_ptrs, s);
  HET_REQUIRE(!fold_attn_l, "%s: fold_attn_l on compact rows needs the by_src_row / by_dst_row groupings", op);
  return gat_backward_edge(v, m, feat, el, er, sum, exp, ret, gradout, grad_feat, grad_el, grad_er, (int)H, (int)D,
                           (float)slope, s);
}

extern "C" int het_relational_fused_gat_csr(const int64_t* in_row_ptrs, const int64_t* in_col, const int64_t* in_eids,
                                            const int64_t* in_reltypes, int64_t num_nodes, int64_t num_edges,
                                            const int64_t* uniq_rel_ptrs, const int64_t* uniq_node_idx,
                                            int64_t num_rels, const float* feat, const float* el, const float* er,
                                            float* sum, float* exp, float* ret, int64_t H, int64_t D, double slope,
                                            int compact, het_stream stream) {
  const char* op = "relational_fused_gat_csr";
  HET_REQUIRE(num_edges >= 0 && num_nodes >= 0 && H > 0 && D > 0, "%s: bad sizes", op);
  HET_REQUIRE(in_row_ptrs && sum && ret && (num_edges == 0 || (in_col && in_eids && feat && el && er && exp)),
              "%s: null pointer", op);
  HET_REQUIRE(!compact || (in_reltypes && uniq_rel_ptrs && uniq_node_idx), "%s: compact needs rel types + unique list", op);
  HET_REQUIRE(num_edges < (1ll << 31) && num_nodes < (1ll << 31), "%s: more than 2^31 edges or nodes", op);
  EdgeView v;
  v.E = num_edges; v.N = num_nodes; v.eids = in_eids; v.src = in_col; v.dst_ptrs = in_row_ptrs;
  v.rel_types = in_reltypes; v.R = (int)num_rels;
  RowMaps m;
  if (compact) { m.kind = HET_KIND_ENABLED; m.ra = m.ca = uniq_rel_ptrs; m.rb = m.cb = uniq_node_idx; }
  return gat_forward_edge(v, m, feat, el, er, sum, exp, ret, (int)H, (int)D, (float)slope, (hipStream_t)stream);
}

extern "C" int het_backward_relational_fused_gat_csr(
    const int64_t* out_row_ptrs, const int64_t* out_col, const int64_t* out_eids, const int64_t* out_reltypes,
    int64_t num_nodes, int64_t num_edges, const int64_t* uniq_rel_ptrs, const int64_t* uniq_node_idx, int64_t num_rels,
    const float* feat, const float* el, const float* er, const float* sum, const float* exp, const float* ret,
    const float* gradout, float* grad_feat, float* grad_el, float* grad_er, int64_t H, int64_t D, double slope,
    int compact, het_stream stream) {
  const char* op = "backward_relational_fused_gat_csr";
  HET_REQUIRE(num_edges >= 0 && num_nodes >= 0 && H > 0 && D > 0, "%s: bad sizes", op);
  HET_REQUIRE(out_row_ptrs && (num_edges == 0 || (out_col && out_eids && feat && el && er && sum && exp && ret &&
                                                  gradout && grad_feat && grad_el && grad_er)),
              "%s: null pointer", op);
  HET_REQUIRE(!compact || (out_reltypes && uniq_rel_ptrs && uniq_node_idx), "%s: compact needs rel types + unique list", op);
  HET_REQUIRE(num_edges < (1ll << 31) && num_nodes < (1ll << 31), "%s: more than 2^31 edges or nodes", op);
  EdgeView v;
  v.E = num_edges; v.N = num_nodes; v.eids = out_eids; v.dst = out_col; v.src_ptrs = out_row_ptrs;
  v.rel_types = out_reltypes; v.R = (int)num_rels;
  RowMaps m;
  if (compact) { m.kind = HET_KIND_ENABLED; m.ra = m.ca = uniq_rel_ptrs; m.rb = m.cb = uniq_node_idx; }
  return gat_backward_edge(v, m, feat, el, er, sum, exp, ret, gradout, grad_feat, grad_el, grad_er, (int)H, (int)D,
                           (float)slope, (hipStream_t)stream);
}
