// RGCN aggregation of a compact (relation, source) feature tensor, forward and backward:
//   ret[col[i], :]        = SUM_i enorm[eids[i]] * feat[crow(i), :]
//   grad_feat[crow(i), :] += enorm[eids[i]] * gradout[col[i], :]
// Edge-parallel, one thread per (edge, feature), float atomics on the output rows
// (reference schedule: RGCN/RGCNKernelsEdgeParallel.cu.h:20-92).
#include "edge_view.hip.h"
#include "seg_reduce.hip.h"

namespace {

constexpr int kBlock = 256;

// BACKWARD = false: out = ret (row = dst), in = feat (row = crow)
// BACKWARD = true : out = grad_feat (row = crow), in = gradout (row = dst)
template <bool BACKWARD>
__global__ __launch_bounds__(kBlock) void HET_rgcn_compact_aggregate(EdgeView v, int direct, const idx_t* __restrict__ map_a,
                                                                      const idx_t* __restrict__ map_b,
                                                                      const float* __restrict__ in,
                                                                      const float* __restrict__ enorm,
                                                                      float* __restrict__ out, int X) {
  const int64_t total = (int64_t)v.E * X, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / X;
    const int x = (int)(t - i * X);
    const idx_t eid = v.eids[i], d = v.dst[i];
    const idx_t cr = direct ? map_a[eid] : compact_row(HET_KIND_ENABLED, map_a, map_b, ev_rel(v, i), v.src[i], eid);
    const float w = enorm[eid];
    if (BACKWARD) atomicAdd(&out[cr * X + x], w * in[d * X + x]);
    else atomicAdd(&out[d * X + x], w * in[cr * X + x]);
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = ceil_div64(total, kBlock);
  const int64_t cap = 256 * 64;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int het_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(
    const int64_t* eids, const int64_t* rel_ptrs, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const int64_t* map_a, const int64_t* map_b, const float* feat,
    const float* enorm, float* ret, int64_t X, int direct, const het_grouping* by_dst, het_stream stream) {
  const char* op = "rgcn_node_mean_aggregation_compact_as_of_node_separate_coo";
  HET_REQUIRE(num_rels > 0 && num_edges >= 0 && num_nodes >= 0 && X > 0, "%s: bad sizes", op);
  HET_REQUIRE(ret && (num_edges == 0 || (eids && rel_ptrs && row && col && feat && enorm && map_a && (direct || map_b))),
              "%s: null pointer", op);
  hipStream_t s = (hipStream_t)stream;
  if (by_dst && by_dst->R == 0 && by_dst->E == num_edges && by_dst->p0 && by_dst->p1 && segment_sum_supported((int)X) &&
      num_edges > 0 && ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(ret)) & 15) == 0)
    // ret[dst, :] = SUM over the in-edges of enorm[eid] * feat[srow, :]: a segmented sum over the destination grouping
    // (payload0 = compact row of the edge's source, payload1 = edge id) instead of E*X float atomics
    return launch_segment_sum(by_dst, feat, ret, (int)X, enorm, s, 0, num_nodes, 0);
  HET_HIP(hipMemsetAsync(ret, 0, sizeof(float) * num_nodes * X, s));
  if (num_edges == 0) return HET_OK;
  EdgeView v;
  v.E = num_edges; v.N = num_nodes; v.eids = eids; v.src = row; v.dst = col; v.rel_ptrs = rel_ptrs; v.R = (int)num_rels;
  hipLaunchKernelGGL(HET_rgcn_compact_aggregate<false>, dim3(grid_for(num_edges * X)), dim3(kBlock), 0, s, v, direct,
                     map_a, map_b, feat, enorm, ret, (int)X);
  HET_LAUNCH_CHECK("HET_rgcn_compact_aggregate");
  return HET_OK;
}

extern "C" int het_backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(
    const int64_t* eids, const int64_t* rel_ptrs, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const int64_t* map_a, const int64_t* map_b, const float* feat,
    const float* enorm, const float* ret, const float* gradout, float* grad_feat, int64_t X, int direct,
    const het_grouping* by_src_row, int64_t n_src_rows, het_stream stream) {
  const char* op = "backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo";
  HET_REQUIRE(num_rels > 0 && num_edges >= 0 && num_nodes >= 0 && X > 0, "%s: bad sizes", op);
  HET_REQUIRE(num_edges == 0 || (eids && rel_ptrs && row && col && enorm && gradout && grad_feat && map_a && (direct || map_b)),
              "%s: null pointer", op);
  (void)feat; (void)ret;
  if (num_edges == 0) return HET_OK;
  hipStream_t s = (hipStream_t)stream;
  if (by_src_row && by_src_row->R == 0 && by_src_row->E == num_edges && by_src_row->p0 && by_src_row->p1 &&
      segment_sum_supported((int)X) && n_src_rows >= 0 &&
      ((reinterpret_cast<uintptr_t>(gradout) | reinterpret_cast<uintptr_t>(grad_feat)) & 15) == 0)
    // grad_feat[srow, :] += SUM over the edges of that compact row of enorm[eid] * gradout[dst, :]
    // (grouping by compact row, payload0 = destination, payload1 = edge id)
    return launch_segment_sum(by_src_row, gradout, grad_feat, (int)X, enorm, s, 0, n_src_rows, 1);
  EdgeView v;
  v.E = num_edges; v.N = num_nodes; v.eids = eids; v.src = row; v.dst = col; v.rel_ptrs = rel_ptrs; v.R = (int)num_rels;
  hipLaunchKernelGGL(HET_rgcn_compact_aggregate<true>, dim3(grid_for(num_edges * X)), dim3(kBlock), 0, s, v, direct,
                     map_a, map_b, gradout, enorm, grad_feat, (int)X);
  HET_LAUNCH_CHECK("HET_rgcn_compact_aggregate");
  return HET_OK;
}
