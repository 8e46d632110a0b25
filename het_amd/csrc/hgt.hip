// HGT ops: edge softmax with per-relation temperature, fused message generation + weighted
// aggregation, edge-wise inner products and the fused attention score; forward and backward.
// The dk x dk per-head products run through the generic segment GEMM kernels (seg_gemm.hip) with
// per-(edge, head) row scales; the edge-wise parts are the small kernels below.
#include "edge_view.hip.h"
#include "seg_gemm.hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxLdsHeads = 64 * 64;  // (relation-local) head slots reduced in LDS

inline unsigned grid_for(int64_t total) {
  int64_t b = ceil_div64(total, kBlock);
  const int64_t cap = 256 * 64;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// m[eid,h] = exp(score[eid,h] * mu[r,h]);  sum[dst,h] += m
__global__ __launch_bounds__(kBlock) void HET_hgt_softmax_exp_sum(const idx_t* __restrict__ col,
                                                                   const idx_t* __restrict__ eids,
                                                                   const idx_t* __restrict__ rel_ptrs, int R, int64_t E,
                                                                   const float* __restrict__ score,
                                                                   const float* __restrict__ mu, float* __restrict__ sum,
                                                                   float* __restrict__ m, int H) {
  const int64_t total = E * H, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / H;
    const int h = (int)(t - i * H);
    const int r = find_segment(rel_ptrs, R, i);
    const idx_t eid = eids[i];
    const float v = expf(score[eid * H + h] * mu[r * H + h]);
    m[eid * H + h] = v;
    atomicAdd(&sum[col[i] * H + h], v);
  }
}

__global__ __launch_bounds__(kBlock) void HET_hgt_softmax_normalize(const idx_t* __restrict__ col,
                                                                     const idx_t* __restrict__ eids, int64_t E,
                                                                     const float* __restrict__ sum,
                                                                     const float* __restrict__ m, float* __restrict__ a,
                                                                     int H) {
  const int64_t total = E * H, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / H;
    const int h = (int)(t - i * H);
    const idx_t eid = eids[i];
    a[eid * H + h] = m[eid * H + h] / sum[col[i] * H + h];
  }
}

// tmp[dst,h] += a * grad_a
__global__ __launch_bounds__(kBlock) void HET_hgt_softmax_bwd_stage0(const idx_t* __restrict__ col,
                                                                      const idx_t* __restrict__ eids, int64_t E,
                                                                      const float* __restrict__ a,
                                                                      const float* __restrict__ grad_a,
                                                                      float* __restrict__ tmp, int H) {
  const int64_t total = E * H, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / H;
    const int h = (int)(t - i * H);
    const idx_t eid = eids[i];
    atomicAdd(&tmp[col[i] * H + h], a[eid * H + h] * grad_a[eid * H + h]);
  }
}

// c = (grad_a - tmp[dst]) * a;  grad_score = c * mu[r];  grad_mu[r,h] += c * score
// One workgroup owns a chunk of edges of ONE relation; the per-head partial sums of grad_mu are
// reduced in LDS and flushed with one atomic per head and workgroup.
__global__ __launch_bounds__(kBlock) void HET_hgt_softmax_bwd_stage1(const idx_t* __restrict__ col,
                                                                      const idx_t* __restrict__ eids,
                                                                      const idx_t* __restrict__ rel_ptrs, int R,
                                                                      int chunk, const float* __restrict__ score,
                                                                      const float* __restrict__ a,
                                                                      const float* __restrict__ grad_a,
                                                                      const float* __restrict__ mu,
                                                                      const float* __restrict__ tmp,
                                                                      float* __restrict__ grad_score,
                                                                      float* __restrict__ grad_mu, int H) {
  extern __shared__ float part[];  // [H]
  int r;
  idx_t rb, re;
  if (!tile_to_relation(rel_ptrs, R, chunk, blockIdx.x, r, rb, re)) return;
  for (int h = threadIdx.x; h < H; h += kBlock) part[h] = 0.f;
  __syncthreads();
  const int64_t total = (re - rb) * H;
  for (int64_t t = threadIdx.x; t < total; t += kBlock) {
    const idx_t i = rb + t / H;
    const int h = (int)(t % H);
    const idx_t eid = eids[i];
    const float av = a[eid * H + h];
    const float c = (grad_a[eid * H + h] - tmp[col[i] * H + h]) * av;
    grad_score[eid * H + h] = c * mu[r * H + h];
    atomicAdd(&part[h], c * score[eid * H + h]);
  }
  __syncthreads();
  for (int h = threadIdx.x; h < H; h += kBlock) atomicAdd(&grad_mu[(int64_t)r * H + h], part[h]);
}

// out[eids[i], h] = < left[lrow(i), h, :], right[ridx[i], h, :] >
__global__ __launch_bounds__(kBlock) void HET_edge_inner_product(EdgeView v, int kind, const idx_t* __restrict__ map_a,
                                                                  const idx_t* __restrict__ map_b,
                                                                  const idx_t* __restrict__ lnode,
                                                                  const idx_t* __restrict__ ridx,
                                                                  const float* __restrict__ left,
                                                                  const float* __restrict__ right,
                                                                  float* __restrict__ out, int H, int D) {
  const int64_t total = (int64_t)v.E * H, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / H;
    const int h = (int)(t - i * H);
    const idx_t eid = v.eids[i];
    idx_t lr = eid;
    if (kind == HET_KIND_DIRECT_INDEX) lr = map_a[eid];
    else if (kind == HET_KIND_ENABLED) lr = compact_row(HET_KIND_ENABLED, map_a, map_b, ev_rel(v, i), lnode[i], eid);
    const float* l = left + (lr * H + h) * D;
    const float* rr = right + (ridx[i] * H + h) * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(l[d], rr[d], s);
    out[eid * H + h] = s;
  }
}

// grad_left[lrow, h, :] += g * right[ridx, h, :];  grad_right[ridx, h, :] += g * left[lrow, h, :]
// (either output may be NULL)
__global__ __launch_bounds__(kBlock) void HET_edge_inner_product_bwd(
    EdgeView v, int kind, const idx_t* __restrict__ map_a, const idx_t* __restrict__ map_b,
    const idx_t* __restrict__ lnode, const idx_t* __restrict__ ridx, const float* __restrict__ left,
    const float* __restrict__ right, const float* __restrict__ gout, float* __restrict__ grad_left,
    float* __restrict__ grad_right, int H, int D) {
  const int X = H * D;
  const int64_t total = (int64_t)v.E * X, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / X;
    const int x = (int)(t - i * X), h = x / D;
    const idx_t eid = v.eids[i];
    idx_t lr = eid;
    if (kind == HET_KIND_DIRECT_INDEX) lr = map_a[eid];
    else if (kind == HET_KIND_ENABLED) lr = compact_row(HET_KIND_ENABLED, map_a, map_b, ev_rel(v, i), lnode[i], eid);
    const float g = gout[eid * H + h];
    const idx_t rr = ridx[i];
    if (grad_left) atomicAdd(&grad_left[lr * X + x], g * right[rr * X + x]);
    if (grad_right) atomicAdd(&grad_right[rr * X + x], g * left[lr * X + x]);
  }
}

// grad_a[eids[i], h] = < gradout[col[i],h,:] . Wt[r,h], v[row[i],h,:] >   (Wt [R,H,do,dk])
__global__ __launch_bounds__(kBlock) void HET_hgt_grad_attn(const idx_t* __restrict__ row, const idx_t* __restrict__ col,
                                                             const idx_t* __restrict__ eids,
                                                             const idx_t* __restrict__ rel_ptrs, int R, int64_t E,
                                                             const float* __restrict__ vfeat,
                                                             const float* __restrict__ Wt,
                                                             const float* __restrict__ gradout,
                                                             float* __restrict__ grad_a, int H, int dk, int dout) {
  const int64_t total = E * H, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / H;
    const int h = (int)(t - i * H);
    const int r = find_segment(rel_ptrs, R, i);
    const float* g = gradout + (col[i] * H + h) * dout;
    const float* vv = vfeat + (row[i] * H + h) * dk;
    const float* w = Wt + ((int64_t)r * H + h) * dout * dk;
    float s = 0.f;
    for (int d = 0; d < dout; ++d) {
      float b = 0.f;
      for (int k = 0; k < dk; ++k) b = fmaf(w[d * dk + k], vv[k], b);
      s = fmaf(g[d], b, s);
    }
    grad_a[eids[i] * H + h] = s;
  }
}

int check_edges(const char* op, const idx_t* row, const idx_t* col, const idx_t* eids, const idx_t* rel_ptrs,
                int64_t R, int64_t E, int64_t N) {
  HET_REQUIRE(R > 0 && E >= 0 && N >= 0 && E < (1ll << 31) && N < (1ll << 31), "%s: bad sizes", op);
  HET_REQUIRE(rel_ptrs && (E == 0 || (row && col && eids)), "%s: null index pointer", op);
  return HET_OK;
}

}  // namespace

extern "C" int het_hgt_full_graph_edge_softmax_ops_separate_coo(
    const int64_t* row, const int64_t* col, const int64_t* eids, const int64_t* rel_ptrs, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const float* score, const float* mu, float* sum, float* m, float* a, int64_t H,
    het_stream stream) {
  const char* op = "hgt_full_graph_edge_softmax_ops_separate_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, num_nodes)) return rc;
  HET_REQUIRE(H > 0 && sum && (num_edges == 0 || (score && mu && m && a)), "%s: null data pointer", op);
  hipStream_t s = (hipStream_t)stream;
  HET_HIP(hipMemsetAsync(sum, 0, sizeof(float) * num_nodes * H, s));
  if (num_edges == 0) return HET_OK;
  hipLaunchKernelGGL(HET_hgt_softmax_exp_sum, dim3(grid_for(num_edges * H)), dim3(kBlock), 0, s, col, eids, rel_ptrs,
                     (int)num_rels, num_edges, score, mu, sum, m, (int)H);
  HET_LAUNCH_CHECK("HET_hgt_softmax_exp_sum");
  hipLaunchKernelGGL(HET_hgt_softmax_normalize, dim3(grid_for(num_edges * H)), dim3(kBlock), 0, s, col, eids, num_edges,
                     sum, m, a, (int)H);
  HET_LAUNCH_CHECK("HET_hgt_softmax_normalize");
  return HET_OK;
}

extern "C" int het_backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(
    const int64_t* row, const int64_t* col, const int64_t* eids, const int64_t* rel_ptrs, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const float* score, const float* a, const float* grad_a, const float* mu,
    float* grad_score, float* grad_mu, float* tmp, int64_t H, het_stream stream) {
  const char* op = "backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, num_nodes)) return rc;
  HET_REQUIRE(H > 0 && H <= kMaxLdsHeads && tmp && (num_edges == 0 || (score && a && grad_a && mu && grad_score && grad_mu)),
              "%s: null data pointer or too many heads", op);
  hipStream_t s = (hipStream_t)stream;
  HET_HIP(hipMemsetAsync(tmp, 0, sizeof(float) * num_nodes * H, s));
  if (num_edges == 0) return HET_OK;
  hipLaunchKernelGGL(HET_hgt_softmax_bwd_stage0, dim3(grid_for(num_edges * H)), dim3(kBlock), 0, s, col, eids, num_edges,
                     a, grad_a, tmp, (int)H);
  HET_LAUNCH_CHECK("HET_hgt_softmax_bwd_stage0");
  int64_t chunk = ceil_div64(num_edges, 4096);
  if (chunk < 256) chunk = 256;
  hipLaunchKernelGGL(HET_hgt_softmax_bwd_stage1, dim3((unsigned)(ceil_div64(num_edges, chunk) + num_rels)), dim3(kBlock),
                     sizeof(float) * H, s, col, eids, rel_ptrs, (int)num_rels, (int)chunk, score, a, grad_a, mu, tmp,
                     grad_score, grad_mu, (int)H);
  HET_LAUNCH_CHECK("HET_hgt_softmax_bwd_stage1");
  return HET_OK;
}

extern "C" int het_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(
    const int64_t* rel_ptrs, const int64_t* eids, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const float* v, const float* weights, const float* a, float* new_h, int64_t H,
    int64_t dk, int64_t dout, het_stream stream) {
  const char* op = "hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, num_nodes)) return rc;
  HET_REQUIRE(H > 0 && dk > 0 && dout > 0 && (num_edges == 0 || (v && weights && a && new_h)), "%s: null data pointer", op);
  SegGemmArgs g;  // new_h[col, h, :] += (v[row, h, :] * a[eid, h]) . W[r, h]
  g.A = v; g.a_ld = H * dk; g.a_head_stride = dk; g.gather = row;
  g.row_scale = a; g.scale_idx = eids; g.scale_ld = H; g.scale_zs = 1;
  g.B = weights; g.b_rel_stride = H * dk * dout; g.b_head_stride = dk * dout;
  g.C = new_h; g.c_ld = H * dout; g.c_head_stride = dout; g.scatter = col; g.atomic = 1;
  g.seg_ptrs = rel_ptrs; g.num_segs = (int)num_rels; g.num_rows = num_edges; g.KA = (int)dk; g.NB = (int)dout;
  g.heads_z = (int)H;
  return launch_seg_gemm(g, (hipStream_t)stream);
}

extern "C" int het_backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(
    const int64_t* rel_ptrs, const int64_t* eids, const int64_t* row, const int64_t* col, int64_t num_rels,
    int64_t num_edges, int64_t num_nodes, const float* v, const float* weights_t, const float* a, const float* new_h,
    float* grad_v, float* grad_w, float* grad_a, const float* gradout, int64_t H, int64_t dk, int64_t dout,
    het_stream stream) {
  const char* op = "backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, num_nodes)) return rc;
  HET_REQUIRE(H > 0 && dk > 0 && dout > 0 &&
                  (num_edges == 0 || (v && weights_t && a && grad_v && grad_w && grad_a && gradout)),
              "%s: null data pointer", op);
  (void)new_h;
  if (num_edges == 0) return HET_OK;
  hipStream_t s = (hipStream_t)stream;
  SegGemmArgs g;  // grad_v[row, h, :] += (gradout[col, h, :] * a[eid, h]) . Wt[r, h]
  g.A = gradout; g.a_ld = H * dout; g.a_head_stride = dout; g.gather = col;
  g.row_scale = a; g.scale_idx = eids; g.scale_ld = H; g.scale_zs = 1;
  g.B = weights_t; g.b_rel_stride = H * dout * dk; g.b_head_stride = dout * dk;
  g.C = grad_v; g.c_ld = H * dk; g.c_head_stride = dk; g.scatter = row; g.atomic = 1;
  g.seg_ptrs = rel_ptrs; g.num_segs = (int)num_rels; g.num_rows = num_edges; g.KA = (int)dout; g.NB = (int)dk;
  g.heads_z = (int)H;
  if (int rc = launch_seg_gemm(g, s)) return rc;
  SegDwArgs w;  // grad_w[r, h] += (v[row, h, :] * a)^T (x) gradout[col, h, :]
  w.A = v; w.a_ld = H * dk; w.a_head_stride = dk; w.gather = row;
  w.row_scale = a; w.scale_idx = eids; w.scale_ld = H; w.scale_zs = 1;
  w.G = gradout; w.g_ld = H * dout; w.g_head_stride = dout; w.g_gather = col;
  w.dW = grad_w; w.dw_rel_stride = H * dk * dout; w.dw_head_stride = dk * dout;
  w.seg_ptrs = rel_ptrs; w.num_segs = (int)num_rels; w.num_rows = num_edges; w.KA = (int)dk; w.NB = (int)dout;
  w.heads_z = (int)H;
  if (int rc = launch_seg_dw(w, s)) return rc;
  hipLaunchKernelGGL(HET_hgt_grad_attn, dim3(grid_for(num_edges * H)), dim3(kBlock), 0, s, row, col, eids, rel_ptrs,
                     (int)num_rels, num_edges, v, weights_t, gradout, grad_a, (int)H, (int)dk, (int)dout);
  HET_LAUNCH_CHECK("HET_hgt_grad_attn");
  return HET_OK;
}

extern "C" int het_rgnn_inner_product_right_node_separatecoo(
    int64_t kind, const int64_t* map_a, const int64_t* map_b, const int64_t* rel_ptrs, const int64_t* eids,
    const int64_t* row, const int64_t* col, int64_t num_rels, int64_t num_edges, const float* left, const float* right,
    float* out, int64_t H, int64_t D, het_stream stream) {
  const char* op = "rgnn_inner_product_right_node_separatecoo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, 0)) return rc;
  HET_REQUIRE(kind == 0 || kind == 1 || kind == 2, "%s: unsupported CompactAsOfNodeKind %lld", op, (long long)kind);
  HET_REQUIRE((kind == 0) || (map_a && (kind == 2 || map_b)), "%s: compact kinds need their index lists", op);
  HET_REQUIRE(H > 0 && D > 0 && (num_edges == 0 || (left && right && out)), "%s: null data pointer", op);
  if (num_edges == 0) return HET_OK;
  EdgeView v;
  v.E = num_edges; v.eids = eids; v.rel_ptrs = rel_ptrs; v.R = (int)num_rels;
  hipLaunchKernelGGL(HET_edge_inner_product, dim3(grid_for(num_edges * H)), dim3(kBlock), 0, (hipStream_t)stream, v,
                     (int)kind, map_a, map_b, col, row, left, right, out, (int)H, (int)D);
  HET_LAUNCH_CHECK("HET_edge_inner_product");
  return HET_OK;
}

extern "C" int het_backward_inner_product_right_node_separatecoo(
    int64_t kind, const int64_t* map_a, const int64_t* map_b, const int64_t* rel_ptrs, const int64_t* eids,
    const int64_t* row, const int64_t* col, int64_t num_rels, int64_t num_edges, const float* left, const float* right,
    const float* gradout, float* grad_left, float* grad_right, int64_t H, int64_t D, het_stream stream) {
  const char* op = "backward_inner_product_right_node_separatecoo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, 0)) return rc;
  HET_REQUIRE(kind == 0 || kind == 1 || kind == 2, "%s: unsupported CompactAsOfNodeKind %lld", op, (long long)kind);
  HET_REQUIRE((kind == 0) || (map_a && (kind == 2 || map_b)), "%s: compact kinds need their index lists", op);
  HET_REQUIRE(H > 0 && D > 0 && (num_edges == 0 || (left && right && gradout && grad_left && grad_right)),
              "%s: null data pointer", op);
  if (num_edges == 0) return HET_OK;
  EdgeView v;
  v.E = num_edges; v.eids = eids; v.rel_ptrs = rel_ptrs; v.R = (int)num_rels;
  hipLaunchKernelGGL(HET_edge_inner_product_bwd, dim3(grid_for(num_edges * H * D)), dim3(kBlock), 0, (hipStream_t)stream,
                     v, (int)kind, map_a, map_b, col, row, left, right, gradout, grad_left, grad_right, (int)H, (int)D);
  HET_LAUNCH_CHECK("HET_edge_inner_product_bwd");
  return HET_OK;
}

extern "C" int het_hgt_full_graph_hetero_attention_ops_coo(
    const int64_t* row, const int64_t* col, const int64_t* eids, const int64_t* rel_ptrs, int64_t num_rels,
    int64_t num_edges, const float* k, const float* q, const float* weights, float* inner, float* score, int64_t H,
    int64_t dk, int64_t dout, het_stream stream) {
  const char* op = "hgt_full_graph_hetero_attention_ops_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, 0)) return rc;
  HET_REQUIRE(H > 0 && dk > 0 && dout > 0 && (num_edges == 0 || (k && q && weights && inner && score)),
              "%s: null data pointer", op);
  if (num_edges == 0) return HET_OK;
  hipStream_t s = (hipStream_t)stream;
  SegGemmArgs g;  // inner[eid, h, :] = k[row, h, :] . W[r, h]
  g.A = k; g.a_ld = H * dk; g.a_head_stride = dk; g.gather = row;
  g.B = weights; g.b_rel_stride = H * dk * dout; g.b_head_stride = dk * dout;
  g.C = inner; g.c_ld = H * dout; g.c_head_stride = dout; g.scatter = eids;
  g.seg_ptrs = rel_ptrs; g.num_segs = (int)num_rels; g.num_rows = num_edges; g.KA = (int)dk; g.NB = (int)dout;
  g.heads_z = (int)H;
  if (int rc = launch_seg_gemm(g, s)) return rc;
  EdgeView v;
  v.E = num_edges; v.eids = eids; v.rel_ptrs = rel_ptrs; v.R = (int)num_rels;
  hipLaunchKernelGGL(HET_edge_inner_product, dim3(grid_for(num_edges * H)), dim3(kBlock), 0, s, v, 0, nullptr, nullptr,
                     col, col, inner, q, score, (int)H, (int)dout);
  HET_LAUNCH_CHECK("HET_edge_inner_product");
  return HET_OK;
}

extern "C" int het_backward_hgt_full_graph_hetero_attention_ops_coo(
    const int64_t* row, const int64_t* col, const int64_t* eids, const int64_t* rel_ptrs, int64_t num_rels,
    int64_t num_edges, float* grad_w, const float* weights_t, const float* k, const float* q, const float* inner,
    const float* grad_score, float* grad_k, float* grad_q, int64_t H, int64_t dk, int64_t dout, het_stream stream) {
  const char* op = "backward_hgt_full_graph_hetero_attention_ops_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, 0)) return rc;
  HET_REQUIRE(H > 0 && dk > 0 && dout > 0 &&
                  (num_edges == 0 || (grad_w && weights_t && k && q && inner && grad_score && grad_k && grad_q)),
              "%s: null data pointer", op);
  if (num_edges == 0) return HET_OK;
  hipStream_t s = (hipStream_t)stream;
  EdgeView v;
  v.E = num_edges; v.eids = eids; v.rel_ptrs = rel_ptrs; v.R = (int)num_rels;
  // grad_q[col, h, :] += gs * inner[eid, h, :]
  hipLaunchKernelGGL(HET_edge_inner_product_bwd, dim3(grid_for(num_edges * H * dout)), dim3(kBlock), 0, s, v, 0, nullptr,
                     nullptr, col, col, inner, q, grad_score, (float*)nullptr, grad_q, (int)H, (int)dout);
  HET_LAUNCH_CHECK("HET_edge_inner_product_bwd");
  SegGemmArgs g;  // grad_k[row, h, :] += (gs * q[col, h, :]) . Wt[r, h]
  g.A = q; g.a_ld = H * dout; g.a_head_stride = dout; g.gather = col;
  g.row_scale = grad_score; g.scale_idx = eids; g.scale_ld = H; g.scale_zs = 1;
  g.B = weights_t; g.b_rel_stride = H * dout * dk; g.b_head_stride = dout * dk;
  g.C = grad_k; g.c_ld = H * dk; g.c_head_stride = dk; g.scatter = row; g.atomic = 1;
  g.seg_ptrs = rel_ptrs; g.num_segs = (int)num_rels; g.num_rows = num_edges; g.KA = (int)dout; g.NB = (int)dk;
  g.heads_z = (int)H;
  if (int rc = launch_seg_gemm(g, s)) return rc;
  SegDwArgs w;  // grad_w[r, h] += (gs * k[row, h, :])^T (x) q[col, h, :]
  w.A = k; w.a_ld = H * dk; w.a_head_stride = dk; w.gather = row;
  w.row_scale = grad_score; w.scale_idx = eids; w.scale_ld = H; w.scale_zs = 1;
  w.G = q; w.g_ld = H * dout; w.g_head_stride = dout; w.g_gather = col;
  w.dW = grad_w; w.dw_rel_stride = H * dk * dout; w.dw_head_stride = dk * dout;
  w.seg_ptrs = rel_ptrs; w.num_segs = (int)num_rels; w.num_rows = num_edges; w.KA = (int)dk; w.NB = (int)dout;
  w.heads_z = (int)H;
  return launch_seg_dw(w, s);
}
