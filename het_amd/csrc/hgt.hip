// HGT ops: edge softmax with per-relation temperature, fused message generation + weighted
// aggregation, edge-wise inner products and the fused attention score; forward and backward.
// The dk x dk per-head products run through the generic segment GEMM kernels (seg_gemm.hip) with
// per-(edge, head) row scales; the edge-wise parts are the small kernels below.
#include "coop.hip.h"
#include "edge_view.hip.h"
#include "seg_gemm.hip.h"
#include "seg_gemm_mfma.hip.h"
#include "seg_gemm_any.hip.h"
#include "seg_reduce.hip.h"

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

// Softmax on the destination grouping (by_dst: payload0 = edge id, payload1 = relation of the position), two kernels
// and no E*H float atomics:
//   HET_hgt_softmax_sum_grouped   sum[dst, :] = SUM over the in-edges of exp(score[eid, :] * mu[r, :])
//                                 (wave per work item, H/4 lanes x float4 per edge; atomics only for split hub segments)
//   HET_hgt_softmax_finish        edge order, streaming: m = exp(score * mu[r]), a = m / sum[dst]
__device__ __forceinline__ float4 hgt_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void hgt_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_hgt_softmax_sum_grouped(
    const int32_t* __restrict__ item_seg, const int32_t* __restrict__ item_begin, const int32_t* __restrict__ item_end,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_key, int64_t num_items,
    const int32_t* __restrict__ p_eid, const int32_t* __restrict__ p_rel, const float* __restrict__ score,
    const float* __restrict__ mu, float* __restrict__ sum) {
  constexpr int EPW = 64 / LPR, H = LPR * 4, U = 4;
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= num_items) return;
  const int seg = item_seg[item], b = item_begin[item], e = item_end[item];
  const int slot = lane / LPR, x = (lane % LPR) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    int64_t eid[U];
    int rl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * EPW, jc = j < e ? j : e - 1;
      eid[u] = p_eid[jc];
      rl[u] = p_rel[jc];
    }
    float4 sc[U], mv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) sc[u] = hgt_ld4(score + eid[u] * H + x);
#pragma unroll
    for (int u = 0; u < U; ++u) mv[u] = hgt_ld4(mu + (int64_t)rl[u] * H + x);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float ok = j0 + u * EPW < e ? 1.f : 0.f;
      acc.x += ok * expf(sc[u].x * mv[u].x); acc.y += ok * expf(sc[u].y * mv[u].y);
      acc.z += ok * expf(sc[u].z * mv[u].z); acc.w += ok * expf(sc[u].w * mv[u].w);
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
  }
  if (slot != 0) return;
  float* p = sum + (int64_t)seg_key[seg] * H + x;
  if (b == seg_ptr[seg] && e == seg_ptr[seg + 1]) {
    hgt_st4(p, acc);
  } else {  // hub destination split over several items (sum is zero-filled by the caller)
    atomicAdd(p + 0, acc.x); atomicAdd(p + 1, acc.y); atomicAdd(p + 2, acc.z); atomicAdd(p + 3, acc.w);
  }
}

// one workgroup = a chunk of edges of ONE relation; a lane owns 4 consecutive heads of an edge
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_hgt_softmax_finish(const idx_t* __restrict__ col, const idx_t* __restrict__ eids,
                                                                  const idx_t* __restrict__ rel_ptrs, int R, int chunk,
                                                                  const float* __restrict__ score,
                                                                  const float* __restrict__ mu,
                                                                  const float* __restrict__ sum, float* __restrict__ m,
                                                                  float* __restrict__ a) {
  constexpr int H = LPR * 4, EPB = kBlock / LPR;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(rel_ptrs, R, chunk, blockIdx.x, r, rb, re)) return;
  const int slot = threadIdx.x / LPR, x = (threadIdx.x % LPR) * 4;
  const float4 mv = hgt_ld4(mu + (int64_t)r * H + x);
  for (idx_t i = rb + slot; i < re; i += EPB) {
    const idx_t eid = eids[i], dst = col[i];
    const float4 sc = hgt_ld4(score + eid * H + x), sv = hgt_ld4(sum + dst * H + x);
    const float4 v = make_float4(expf(sc.x * mv.x), expf(sc.y * mv.y), expf(sc.z * mv.z), expf(sc.w * mv.w));
    hgt_st4(m + eid * H + x, v);
    hgt_st4(a + eid * H + x, make_float4(v.x / sv.x, v.y / sv.y, v.z / sv.z, v.w / sv.w));
  }
}

inline bool softmax_grouped_ok(const het_grouping* g, int64_t E, int64_t H) {
  return g && g->R == 0 && g->E == E && g->p0 && g->p1 && H % 4 == 0 && H <= 256 && ((H / 4) & (H / 4 - 1)) == 0;
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
  if (kBlock % H == 0) {
    // a thread keeps its head for the whole chunk (t % H == threadIdx.x % H): accumulate in a register,
    // one LDS atomic per thread at the end
    const int h = threadIdx.x % H;
    const float muv = mu[r * H + h];
    float acc = 0.f;
    for (int64_t t = threadIdx.x; t < total; t += kBlock) {
      const idx_t i = rb + t / H;
      const idx_t eid = eids[i];
      const float av = a[eid * H + h];
      const float c = (grad_a[eid * H + h] - tmp[col[i] * H + h]) * av;
      grad_score[eid * H + h] = c * muv;
      acc = fmaf(c, score[eid * H + h], acc);
    }
    atomicAdd(&part[h], acc);
  } else {
    for (int64_t t = threadIdx.x; t < total; t += kBlock) {
      const idx_t i = rb + t / H;
      const int h = (int)(t % H);
      const idx_t eid = eids[i];
      const float av = a[eid * H + h];
      const float c = (grad_a[eid * H + h] - tmp[col[i] * H + h]) * av;
      grad_score[eid * H + h] = c * mu[r * H + h];
      atomicAdd(&part[h], c * score[eid * H + h]);
    }
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

// ---- row kernels: a feature row of X = H*D floats is covered by LPR = X/4 lanes (float4 each); the DL = D/4
// lanes of a head combine with xor-shuffles; U rows per lane group are in flight per step -----------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
constexpr int UR = 4;
__device__ __forceinline__ int quad_bcast_sw(int v, int q) {  // value of lane q of the caller's quad (DPP quad_perm)
  switch (q) {
    case 0: return quad_bcast_i<0>(v);
    case 1: return quad_bcast_i<1>(v);
    case 2: return quad_bcast_i<2>(v);
    default: return quad_bcast_i<3>(v);
  }
}

// Cooperative form of HET_rows_inner_product for LPR >= 4: lane (sub % 4) of every quad fetches the ids of edge (sub % 4) of
// the step and the quad shares them with DPP broadcasts -- 3 id instructions per step of 4 edges instead of 12 (the per-edge
// kernels were bound by the number of vector-memory instructions).
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rows_inner_product_coop(const idx_t* __restrict__ eids,
                                                                       const idx_t* __restrict__ map_a,
                                                                       const idx_t* __restrict__ ridx, int64_t E,
                                                                       const float* __restrict__ left,
                                                                       const float* __restrict__ right,
                                                                       float* __restrict__ out, int H, int D) {
  constexpr int EPW = 64 / LPR, X = LPR * 4;
  static_assert(LPR >= 4 && UR == 4, "whole quads per lane group, 4 edges per step");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2, q4 = sub & 3;
  const int64_t step = (int64_t)gridDim.x * 4 * EPW * UR;
  for (int64_t base = (int64_t)blockIdx.x * 4 * EPW * UR; base < E; base += step) {
    // this lane's edge of the step (u = q4), ids as 32-bit values (E, rows < 2^31 are checked by the callers)
    const int64_t iq = base + (wave * UR + q4) * EPW + slot;
    const int64_t ic = iq < E ? iq : E - 1;
    const int eidv = (int)eids[ic], rrv = (int)ridx[ic];
    const int lrv = map_a ? (int)map_a[eidv] : eidv;
    float4 l[UR], r[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) l[u] = ld4(left + (int64_t)quad_bcast_sw(lrv, u) * X + x);
#pragma unroll
    for (int u = 0; u < UR; ++u) r[u] = ld4(right + (int64_t)quad_bcast_sw(rrv, u) * X + x);
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      float p = l[u].x * r[u].x + l[u].y * r[u].y + l[u].z * r[u].z + l[u].w * r[u].w;
      for (int off = DL >> 1; off > 0; off >>= 1) p += __shfl_xor(p, off);
      const bool ok = base + (wave * UR + u) * EPW + slot < E;
      const int eu = quad_bcast_sw(eidv, u);  // outside the branch: a DPP read of a lane the branch switched off returns 0
      if (ok && (sub & (DL - 1)) == 0) out[(int64_t)eu * H + h] = p;
    }
  }
}

// out[eids[i], h] = < left[lrow(i), h, :], right[ridx[i], h, :] >,  lrow = eid or map_a[eid]
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rows_inner_product(const idx_t* __restrict__ eids,
                                                                  const idx_t* __restrict__ map_a,
                                                                  const idx_t* __restrict__ ridx, int64_t E,
                                                                  const float* __restrict__ left,
                                                                  const float* __restrict__ right,
                                                                  float* __restrict__ out, int H, int D) {
  constexpr int EPW = 64 / LPR, X = LPR * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const int64_t step = (int64_t)gridDim.x * 4 * EPW * UR;
  for (int64_t base = (int64_t)blockIdx.x * 4 * EPW * UR; base < E; base += step) {
    idx_t eid[UR], lr[UR], rr[UR];
    bool ok[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t i = base + (wave * UR + u) * EPW + slot;
      ok[u] = i < E;
      const int64_t ic = ok[u] ? i : E - 1;
      eid[u] = eids[ic];
      rr[u] = ridx[ic];
    }
    if (map_a) {
#pragma unroll
      for (int u = 0; u < UR; ++u) lr[u] = map_a[eid[u]];
    } else {
#pragma unroll
      for (int u = 0; u < UR; ++u) lr[u] = eid[u];
    }
    float4 l[UR], r[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) l[u] = ld4(left + lr[u] * X + x);
#pragma unroll
    for (int u = 0; u < UR; ++u) r[u] = ld4(right + rr[u] * X + x);
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      float p = l[u].x * r[u].x + l[u].y * r[u].y + l[u].z * r[u].z + l[u].w * r[u].w;
      for (int off = DL >> 1; off > 0; off >>= 1) p += __shfl_xor(p, off);
      if (ok[u] && (sub & (DL - 1)) == 0) out[eid[u] * H + h] = p;
    }
  }
}

// grad_left[lrow(i), h, :] (+)= gout[eids[i], h] * right[ridx[i], h, :]
// MODE 0: atomics (shared rows); 1: read-modify-write (one writer per row); 2: store
template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void HET_rows_inner_product_bwd_left(const idx_t* __restrict__ eids,
                                                                           const idx_t* __restrict__ map_a,
                                                                           const idx_t* __restrict__ ridx, int64_t E,
                                                                           const float* __restrict__ right,
                                                                           const float* __restrict__ gout,
                                                                           float* __restrict__ grad_left, int H, int D) {
  constexpr int EPW = 64 / LPR, X = LPR * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, x = (lane % LPR) * 4, h = x / D;
  const int64_t step = (int64_t)gridDim.x * 4 * EPW * UR;
  for (int64_t base = (int64_t)blockIdx.x * 4 * EPW * UR; base < E; base += step) {
    idx_t eid[UR], lr[UR], rr[UR];
    bool ok[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t i = base + (wave * UR + u) * EPW + slot;
      ok[u] = i < E;
      const int64_t ic = ok[u] ? i : E - 1;
      eid[u] = eids[ic];
      rr[u] = ridx[ic];
    }
    if (map_a) {
#pragma unroll
      for (int u = 0; u < UR; ++u) lr[u] = map_a[eid[u]];
    } else {
#pragma unroll
      for (int u = 0; u < UR; ++u) lr[u] = eid[u];
    }
    float g[UR];
    float4 r[UR], c[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) g[u] = gout[eid[u] * H + h];
#pragma unroll
    for (int u = 0; u < UR; ++u) r[u] = ld4(right + rr[u] * X + x);
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < UR; ++u) c[u] = ld4(grad_left + lr[u] * X + x);
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (!ok[u]) continue;
      float* p = grad_left + lr[u] * X + x;
      if (MODE == 2) st4(p, make_float4(g[u] * r[u].x, g[u] * r[u].y, g[u] * r[u].z, g[u] * r[u].w));
      else if (MODE == 1) st4(p, make_float4(fmaf(g[u], r[u].x, c[u].x), fmaf(g[u], r[u].y, c[u].y), fmaf(g[u], r[u].z, c[u].z), fmaf(g[u], r[u].w, c[u].w)));
      else { atomicAdd(p, g[u] * r[u].x); atomicAdd(p + 1, g[u] * r[u].y); atomicAdd(p + 2, g[u] * r[u].z); atomicAdd(p + 3, g[u] * r[u].w); }
    }
  }
}

// grad_a[eids[i], h] = < gradout[col[i],h,:] . Wt[r,h], v[row[i],h,:] >  with dk == dout == DK:
// every lane turns its 4 gradout values into partial "back" values for all DK inputs (Wt slice in registers),
// the DK/4 lanes of the head combine them, then dot with the lane's 4 v values.
template <int LPR, int DK>
__global__ __launch_bounds__(kBlock) void HET_hgt_grad_attn_rows(const idx_t* __restrict__ row, const idx_t* __restrict__ col,
                                                                  const idx_t* __restrict__ eids,
                                                                  const idx_t* __restrict__ rel_ptrs, int R, int chunk,
                                                                  const float* __restrict__ vfeat,
                                                                  const float* __restrict__ Wt,
                                                                  const float* __restrict__ gradout,
                                                                  float* __restrict__ grad_a, int H) {
  constexpr int EPW = 64 / LPR, X = LPR * 4, DL = DK / 4, U2 = 2;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(rel_ptrs, R, chunk, blockIdx.x, r, rb, re)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / DK, d0 = x - h * DK;
  float w[4][DK];  // Wt[r, h, d0 + j, :]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < DK; ++k) w[j][k] = Wt[(((int64_t)r * H + h) * DK + d0 + j) * DK + k];
  for (idx_t base = rb; base < re; base += 4 * EPW * U2) {
    idx_t sr[U2], ds[U2], eid[U2];
    bool ok[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      const idx_t i = base + (wave * U2 + u) * EPW + slot;
      ok[u] = i < re;
      const idx_t ic = ok[u] ? i : re - 1;
      sr[u] = row[ic]; ds[u] = col[ic]; eid[u] = eids[ic];
    }
    float4 g[U2], vv[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) g[u] = ld4(gradout + ds[u] * X + x);
#pragma unroll
    for (int u = 0; u < U2; ++u) vv[u] = ld4(vfeat + sr[u] * X + x);
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      float back[DK];
#pragma unroll
      for (int k = 0; k < DK; ++k) back[k] = g[u].x * w[0][k] + g[u].y * w[1][k] + g[u].z * w[2][k] + g[u].w * w[3][k];
#pragma unroll
      for (int off = DL >> 1; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < DK; ++k) back[k] += __shfl_xor(back[k], off);
      // this lane's v slice covers inputs d0 .. d0+3 of the head
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        const float vk = (k == d0) ? vv[u].x : (k == d0 + 1) ? vv[u].y : (k == d0 + 2) ? vv[u].z : (k == d0 + 3) ? vv[u].w : 0.f;
        t = fmaf(back[k], vk, t);
      }
      for (int off = DL >> 1; off > 0; off >>= 1) t += __shfl_xor(t, off);
      if (ok[u] && (sub & (DL - 1)) == 0) grad_a[eid[u] * H + h] = t;
    }
  }
}

// grad_a for any head width: msg[(r,u), h, :] = v[u, h, :] . W[r, h] is formed once per distinct (relation, source) row by
// the block-diagonal MFMA GEMM, then   grad_a[eid, h] = < gradout[dst, h, :], msg[seg, h, :] >   rank-parallel over the
// (relation, source) grouping (payload0 = destination, payload1 = edge id): a lane group per edge, UR edges in flight.
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_hgt_grad_attn_msg(const int32_t* __restrict__ seg_of_rank,
                                                                 const int32_t* __restrict__ p_dst,
                                                                 const int32_t* __restrict__ p_eid, int64_t E,
                                                                 const float* __restrict__ msg,
                                                                 const float* __restrict__ gradout,
                                                                 float* __restrict__ grad_a, int H, int D) {
  constexpr int EPW = 64 / LPR, X = LPR * 4, UG = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const int64_t step = (int64_t)gridDim.x * 4 * EPW * UG;
  for (int64_t base = (int64_t)blockIdx.x * 4 * EPW * UG; base < E; base += step) {
    int64_t sg[UG], ds[UG], ed[UG];
    bool ok[UG];
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      const int64_t j = base + (wave * UG + u) * EPW + slot;
      ok[u] = j < E;
      const int64_t jc = ok[u] ? j : E - 1;
      sg[u] = seg_of_rank[jc]; ds[u] = p_dst[jc]; ed[u] = p_eid[jc];
    }
    float4 m[UG], g[UG];
#pragma unroll
    for (int u = 0; u < UG; ++u) m[u] = ld4(msg + sg[u] * X + x);
#pragma unroll
    for (int u = 0; u < UG; ++u) g[u] = ld4(gradout + ds[u] * X + x);
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      float t = m[u].x * g[u].x + m[u].y * g[u].y + m[u].z * g[u].z + m[u].w * g[u].w;
      for (int off = DL >> 1; off > 0; off >>= 1) t += __shfl_xor(t, off);
      if (ok[u] && (sub & (DL - 1)) == 0) grad_a[ed[u] * H + h] = t;
    }
  }
}

// W[r, h, k, d] = Wt[r, h, d, k]  (the op receives the transposed weight only)
__global__ __launch_bounds__(kBlock) void HET_hgt_untranspose_w(const float* __restrict__ Wt, float* __restrict__ W, int64_t RH,
                                                                 int dk, int dout) {
  const int64_t total = RH * dk * dout;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t rh = t / ((int64_t)dk * dout);
    const int rem = (int)(t - rh * dk * dout), k = rem / dout, d = rem - k * dout;
    W[t] = Wt[(rh * dout + d) * dk + k];
  }
}

inline bool is_pow2(int64_t x) { return x > 0 && (x & (x - 1)) == 0; }
inline bool rows_shape_ok(int64_t H, int64_t D) {  // X/4 a power of two <= 64, a head = whole float4 pieces
  const int64_t X = H * D;
  return D >= 4 && is_pow2(D) && is_pow2(X) && X / 4 <= 64;
}

#define HET_HGT_LPR(LPRV, CALL)                         \
  switch (LPRV) {                                       \
    case 1: { constexpr int LPR = 1; CALL; break; }     \
    case 2: { constexpr int LPR = 2; CALL; break; }     \
    case 4: { constexpr int LPR = 4; CALL; break; }     \
    case 8: { constexpr int LPR = 8; CALL; break; }     \
    case 16: { constexpr int LPR = 16; CALL; break; }   \
    case 32: { constexpr int LPR = 32; CALL; break; }   \
    default: { constexpr int LPR = 64; CALL; break; }   \
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
    const het_grouping* by_dst, het_stream stream) {
  const char* op = "hgt_full_graph_edge_softmax_ops_separate_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, num_nodes)) return rc;
  HET_REQUIRE(H > 0 && sum && (num_edges == 0 || (score && mu && m && a)), "%s: null data pointer", op);
  hipStream_t s = (hipStream_t)stream;
  HET_HIP(hipMemsetAsync(sum, 0, sizeof(float) * num_nodes * H, s));
  if (num_edges == 0) return HET_OK;
  if (softmax_grouped_ok(by_dst, num_edges, H) && ((reinterpret_cast<uintptr_t>(score) | reinterpret_cast<uintptr_t>(mu) |
                                                     reinterpret_cast<uintptr_t>(sum) | reinterpret_cast<uintptr_t>(m) |
                                                     reinterpret_cast<uintptr_t>(a)) & 15) == 0) {
    const het_grouping* g = by_dst;
    const unsigned nb = (unsigned)ceil_div64(g->num_items, kBlock / 64);
    HET_HGT_LPR((int)(H / 4), hipLaunchKernelGGL(HET_hgt_softmax_sum_grouped<LPR>, dim3(nb), dim3(kBlock), 0, s, g->item_seg,
                                                 g->item_begin, g->item_end, g->seg_ptr, g->seg_key, g->num_items, g->p0,
                                                 g->p1, score, mu, sum));
    HET_LAUNCH_CHECK("HET_hgt_softmax_sum_grouped");
    const int64_t chunk = 2048;
    const unsigned nf = (unsigned)(ceil_div64(num_edges, chunk) + num_rels);
    HET_HGT_LPR((int)(H / 4), hipLaunchKernelGGL(HET_hgt_softmax_finish<LPR>, dim3(nf), dim3(kBlock), 0, s, col, eids, rel_ptrs,
                                                 (int)num_rels, (int)chunk, score, mu, sum, m, a));
    HET_LAUNCH_CHECK("HET_hgt_softmax_finish");
    return HET_OK;
  }
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
    float* grad_score, float* grad_mu, float* tmp, int64_t H, const het_grouping* by_dst, het_stream stream) {
  const char* op = "backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, num_nodes)) return rc;
  HET_REQUIRE(H > 0 && H <= kMaxLdsHeads && tmp && (num_edges == 0 || (score && a && grad_a && mu && grad_score && grad_mu)),
              "%s: null data pointer or too many heads", op);
  hipStream_t s = (hipStream_t)stream;
  if (num_edges > 0 && softmax_grouped_ok(by_dst, num_edges, H) && segment_sum_supported((int)H) &&
      ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(grad_a) | reinterpret_cast<uintptr_t>(tmp)) & 15) == 0) {
    // tmp[dst, :] = SUM over the in-edges of a[eid, :] * grad_a[eid, :]: a segmented sum instead of E*H float atomics
    if (int rc = launch_segment_sum(by_dst, grad_a, tmp, (int)H, a, s, (int)H, num_nodes, 0, 1)) return rc;
  } else {
    HET_HIP(hipMemsetAsync(tmp, 0, sizeof(float) * num_nodes * H, s));
    if (num_edges == 0) return HET_OK;
    hipLaunchKernelGGL(HET_hgt_softmax_bwd_stage0, dim3(grid_for(num_edges * H)), dim3(kBlock), 0, s, col, eids, num_edges,
                       a, grad_a, tmp, (int)H);
    HET_LAUNCH_CHECK("HET_hgt_softmax_bwd_stage0");
  }
  if (num_edges == 0) return HET_OK;
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
    int64_t dk, int64_t dout, const het_grouping* by_rel_dst, void* workspace, int64_t workspace_bytes,
    het_stream stream) {
  const char* op = "hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, num_nodes)) return rc;
  HET_REQUIRE(H > 0 && dk > 0 && dout > 0 && (num_edges == 0 || (v && weights && a && new_h)), "%s: null data pointer", op);
  const het_grouping* gr = by_rel_dst;
  if (gr && gr->R == (int)num_rels && gr->E == num_edges && gr->p0 && gr->p1 && rows_shape_ok(H, dk) &&
      segment_sum_supported((int)(H * dk)) && workspace &&
      workspace_bytes >= (int64_t)sizeof(float) * gr->S * H * dk && (reinterpret_cast<uintptr_t>(v) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(new_h) & 15) == 0) {
    // new_h[dst] += SUM_r ( SUM_{e in (r,dst)} a[e,h] * v[src_e,h,:] ) . W[r,h]: attention-weighted segment sum of
    // the source rows per (relation, destination), then one block-diagonal row GEMM per distinct pair
    hipStream_t s = (hipStream_t)stream;
    float* ssum = static_cast<float*>(workspace);
    if (int rc = launch_segment_sum(gr, v, ssum, (int)(H * dk), a, s, (int)H)) return rc;
    MfmaGemmArgs m;
    m.A = ssum; m.a_ld = H * dk; m.B = weights; m.b_rel_stride = H * dk * dout; m.b_headcat = 2; m.headcat_d = (int)dout;
    m.blockdiag_k = (int)dk; m.C = new_h; m.c_ld = H * dout; m.scatter = gr->seg_key64; m.atomic = 1;
    m.seg_ptrs = gr->seg_rel_ptr64; m.num_segs = (int)num_rels; m.num_rows = gr->S; m.K = (int)(H * dk); m.X = (int)(H * dout);
    return launch_rows_gemm(m, s);
  }
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
    const het_grouping* by_rel_src, void* workspace, int64_t workspace_bytes, het_stream stream) {
  const char* op = "backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, num_nodes)) return rc;
  HET_REQUIRE(H > 0 && dk > 0 && dout > 0 &&
                  (num_edges == 0 || (v && weights_t && a && grad_v && grad_w && grad_a && gradout)),
              "%s: null data pointer", op);
  (void)new_h;
  if (num_edges == 0) return HET_OK;
  hipStream_t s = (hipStream_t)stream;
  const het_grouping* gr = by_rel_src;
  // grad_a: from the per-(relation, source) message rows when the workspace has room for them (3.2 ms for the op on
  // ogbn-mag at any head count), else -- narrow heads only -- with the Wt slice of a head in a lane's registers (4.0-4.6 ms)
  const int64_t ws_rows = (int64_t)sizeof(float) * (gr ? gr->S * H * dout + num_rels * H * dk * dout : 0);
  const bool msg_rows = gr && workspace_bytes >= (int64_t)sizeof(float) * gr->S * H * dout + ws_rows;
  const bool reg_w = !msg_rows && dk == dout && (dk == 4 || dk == 8 || dk == 16);
  const int64_t ws_msg = msg_rows ? ws_rows : 0;
  if (!msg_rows && !reg_w) gr = nullptr;  // neither fits: the generic kernels below
  if (gr && gr->R == (int)num_rels && gr->E == num_edges && gr->p0 && gr->p1 && dk == dout &&
      rows_shape_ok(H, dk) && segment_sum_supported((int)(H * dout)) &&
      workspace && workspace_bytes >= (int64_t)sizeof(float) * gr->S * H * dout + ws_msg && (reinterpret_cast<uintptr_t>(gradout) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(v) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(grad_v) & 15) == 0) {
    // gsum[(r,u)] = SUM over the out-edges of u in relation r of a[e,h] * gradout[dst_e,h,:]; then
    //   grad_v[u] += gsum . Wt[r] (block diagonal),  grad_w[r,h] += v[u,h,:]^T (x) gsum[(r,u),h,:]
    float* gsum = static_cast<float*>(workspace);
    if (int rc = launch_segment_sum(gr, gradout, gsum, (int)(H * dout), a, s, (int)H)) return rc;
    MfmaGemmArgs m;
    m.A = gsum; m.a_ld = H * dout; m.B = weights_t; m.b_rel_stride = H * dout * dk; m.b_headcat = 2; m.headcat_d = (int)dk;
    m.blockdiag_k = (int)dout; m.C = grad_v; m.c_ld = H * dk; m.scatter = gr->seg_key64; m.atomic = 1;
    m.seg_ptrs = gr->seg_rel_ptr64; m.num_segs = (int)num_rels; m.num_rows = gr->S; m.K = (int)(H * dout); m.X = (int)(H * dk);
    if (int rc = launch_rows_gemm(m, s)) return rc;
    MfmaDwArgs w;
    w.A = v; w.a_ld = H * dk; w.gather = gr->seg_key64; w.G = gsum; w.g_ld = H * dout; w.dW = grad_w;
    w.dw_rel_stride = H * dk * dout; w.headcat = 2; w.headcat_d = (int)dout; w.blockdiag_k = (int)dk;
    w.seg_ptrs = gr->seg_rel_ptr64; w.num_segs = (int)num_rels; w.num_rows = gr->S; w.K = (int)(H * dk); w.X = (int)(H * dout);
    if (int rc = launch_rows_dw(w, s)) return rc;
    if (!reg_w) {  // wide heads (dk = 32, 64, 128: 1 or 2 heads): grad_a from the per-(relation, source) message rows
      float* msg = gsum + gr->S * H * dout;
      float* W = msg + gr->S * H * dout;
      const int64_t wn = num_rels * H * dk * dout;
      hipLaunchKernelGGL(HET_hgt_untranspose_w, dim3((unsigned)ceil_div64(wn, kBlock)), dim3(kBlock), 0, s, weights_t, W,
                         num_rels * H, (int)dk, (int)dout);
      HET_LAUNCH_CHECK("HET_hgt_untranspose_w");
      MfmaGemmArgs f;
      f.A = v; f.a_ld = H * dk; f.gather = gr->seg_key64; f.B = W; f.b_rel_stride = H * dk * dout; f.b_headcat = 2;
      f.headcat_d = (int)dout; f.blockdiag_k = (int)dk; f.C = msg; f.c_ld = H * dout; f.scatter = nullptr;
      f.seg_ptrs = gr->seg_rel_ptr64; f.num_segs = (int)num_rels; f.num_rows = gr->S; f.K = (int)(H * dk); f.X = (int)(H * dout);
      if (int rc = launch_rows_gemm(f, s)) return rc;
      if (int rc = grouping_seg_of_rank(gr, s)) return rc;
      const int epw = 64 / (int)(H * dout / 4);
      int64_t nbm = ceil_div64(num_edges, (int64_t)4 * epw * 4);
      if (nbm > 256 * 64) nbm = 256 * 64;
      HET_HGT_LPR((int)(H * dout / 4), hipLaunchKernelGGL(HET_hgt_grad_attn_msg<LPR>, dim3((unsigned)nbm), dim3(kBlock), 0, s,
                                                          gr->seg_of_rank, gr->p0, gr->p1, num_edges, msg, gradout, grad_a,
                                                          (int)H, (int)dout));
      HET_LAUNCH_CHECK("HET_hgt_grad_attn_msg");
      return HET_OK;
    }
    int64_t chunk = 4096;
    dim3 grid((unsigned)(ceil_div64(num_edges, chunk) + num_rels)), block(kBlock);
#define HET_GA(DKV) HET_HGT_LPR((int)(H * dk / 4), hipLaunchKernelGGL((HET_hgt_grad_attn_rows<LPR, DKV>), grid, block, 0, s, \
                                row, col, eids, rel_ptrs, (int)num_rels, (int)chunk, v, weights_t, gradout, grad_a, (int)H))
    if (dk == 4) { HET_GA(4); } else if (dk == 8) { HET_GA(8); } else { HET_GA(16); }
#undef HET_GA
    HET_LAUNCH_CHECK("HET_hgt_grad_attn_rows");
    return HET_OK;
  }
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
  if (kind != HET_KIND_ENABLED && rows_shape_ok(H, D) && (reinterpret_cast<uintptr_t>(left) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(right) & 15) == 0) {
    const unsigned nb = grid_for(num_edges * (H * D / 4));
    if (H * D / 4 >= 4 && num_edges < (1ll << 31)) {
      switch ((int)(H * D / 4)) {
#define HET_IP_COOP(L) case L: hipLaunchKernelGGL(HET_rows_inner_product_coop<L>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, \
                                                  eids, kind == 2 ? map_a : nullptr, row, num_edges, left, right, out, (int)H, (int)D); break
        HET_IP_COOP(4); HET_IP_COOP(8); HET_IP_COOP(16); HET_IP_COOP(32);
        default: hipLaunchKernelGGL(HET_rows_inner_product_coop<64>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, eids,
                                    kind == 2 ? map_a : nullptr, row, num_edges, left, right, out, (int)H, (int)D); break;
#undef HET_IP_COOP
      }
      HET_LAUNCH_CHECK("HET_rows_inner_product_coop");
      return HET_OK;
    }
    HET_HGT_LPR((int)(H * D / 4), hipLaunchKernelGGL(HET_rows_inner_product<LPR>, dim3(nb), dim3(kBlock), 0,
                                                      (hipStream_t)stream, eids, kind == 2 ? map_a : nullptr, row,
                                                      num_edges, left, right, out, (int)H, (int)D));
    HET_LAUNCH_CHECK("HET_rows_inner_product");
    return HET_OK;
  }
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
    const float* gradout, float* grad_left, float* grad_right, int64_t H, int64_t D, int accumulate,
    const het_grouping* by_right, const het_grouping* by_left, int64_t n_left_rows, int64_t n_right_rows,
    het_stream stream) {
  const char* op = "backward_inner_product_right_node_separatecoo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, 0)) return rc;
  HET_REQUIRE(kind == 0 || kind == 1 || kind == 2, "%s: unsupported CompactAsOfNodeKind %lld", op, (long long)kind);
  HET_REQUIRE((kind == 0) || (map_a && (kind == 2 || map_b)), "%s: compact kinds need their index lists", op);
  HET_REQUIRE(H > 0 && D > 0 && (num_edges == 0 || (left && right && gradout && grad_left && grad_right)),
              "%s: null data pointer", op);
  hipStream_t s = (hipStream_t)stream;
  const int64_t X = H * D;
  const het_grouping* gr = by_right;
  const bool fast = kind != HET_KIND_ENABLED && rows_shape_ok(H, D) && gr && gr->R == 0 && gr->E == num_edges && gr->p0 &&
                    gr->p1 && segment_sum_supported((int)X) && n_right_rows >= 0 && n_left_rows >= 0 &&
                    (reinterpret_cast<uintptr_t>(left) & 15) == 0 && (reinterpret_cast<uintptr_t>(right) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(grad_left) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_right) & 15) == 0;
  if (fast) {
    // grad_left: one writer per row for kind 0 (left rows are edge rows) -> store / read-modify-write;
    // shared compact rows (kind 2) -> atomics on zeroed rows
    const idx_t* lmap = kind == 2 ? map_a : nullptr;
    const int mode = kind == 0 ? (accumulate ? 1 : 2) : 0;
    const bool left_grouped = mode == 0 && by_left && by_left->R == 0 && by_left->E == num_edges && by_left->p0 && by_left->p1;
    if (left_grouped) {
      // shared (compact) left rows: grad_left[row] (+)= SUM over its edges of gout[e,h] * right[ridx[e],h,:]
      // (payload0 = right row, payload1 = edge id) -- no float atomics
      if (int rc = launch_segment_sum(by_left, right, grad_left, (int)X, gradout, s, (int)H, n_left_rows, accumulate)) return rc;
    } else if (!accumulate && mode == 0) {
      HET_HIP(hipMemsetAsync(grad_left, 0, sizeof(float) * n_left_rows * X, s));
    }
    if (num_edges > 0 && !left_grouped) {
      const unsigned nb = grid_for(num_edges * (X / 4));
#define HET_IPL(MODEV) HET_HGT_LPR((int)(X / 4), hipLaunchKernelGGL((HET_rows_inner_product_bwd_left<LPR, MODEV>), dim3(nb), \
                                   dim3(kBlock), 0, s, eids, lmap, row, num_edges, right, gradout, grad_left, (int)H, (int)D))
      if (mode == 2) { HET_IPL(2); } else if (mode == 1) { HET_IPL(1); } else { HET_IPL(0); }
#undef HET_IPL
      HET_LAUNCH_CHECK("HET_rows_inner_product_bwd_left");
    }
    // grad_right[node] (+)= SUM over the edges whose right operand is node of gout[e,h] * left[lrow(e),h,:]
    return launch_segment_sum(gr, left, grad_right, (int)X, gradout, s, (int)H, n_right_rows, accumulate, 0,
                              /*nt_in*/ 0);
  }
  if (!accumulate) {
    HET_REQUIRE(n_left_rows >= 0 && n_right_rows >= 0, "%s: row counts needed to overwrite the gradients", op);
    HET_HIP(hipMemsetAsync(grad_left, 0, sizeof(float) * n_left_rows * X, s));
    HET_HIP(hipMemsetAsync(grad_right, 0, sizeof(float) * n_right_rows * X, s));
  }
  if (num_edges == 0) return HET_OK;
  EdgeView v;
  v.E = num_edges; v.eids = eids; v.rel_ptrs = rel_ptrs; v.R = (int)num_rels;
  hipLaunchKernelGGL(HET_edge_inner_product_bwd, dim3(grid_for(num_edges * H * D)), dim3(kBlock), 0, s, v, (int)kind,
                     map_a, map_b, col, row, left, right, gradout, grad_left, grad_right, (int)H, (int)D);
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
  if (mfma_shape_supported((int)(H * dk), (int)(H * dout)) && rows_shape_ok(H, dout) &&
      (reinterpret_cast<uintptr_t>(k) & 15) == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(inner) & 15) == 0) {
    MfmaGemmArgs m;  // inner[eid, :] = k[row, :] . blockdiag(W[r])
    m.A = k; m.a_ld = H * dk; m.gather = row; m.B = weights; m.b_rel_stride = H * dk * dout; m.b_headcat = 2;
    m.headcat_d = (int)dout; m.blockdiag_k = (int)dk; m.C = inner; m.c_ld = H * dout; m.scatter = eids;
    m.seg_ptrs = rel_ptrs; m.num_segs = (int)num_rels; m.num_rows = num_edges; m.K = (int)(H * dk); m.X = (int)(H * dout);
    if (int rc = launch_seg_gemm_mfma(m, s)) return rc;
    const unsigned nb = grid_for(num_edges * (H * dout / 4));
    HET_HGT_LPR((int)(H * dout / 4), hipLaunchKernelGGL(HET_rows_inner_product<LPR>, dim3(nb), dim3(kBlock), 0, s, eids,
                                                         (const idx_t*)nullptr, col, num_edges, inner, q, score, (int)H,
                                                         (int)dout));
    HET_LAUNCH_CHECK("HET_rows_inner_product");
    return HET_OK;
  }
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
    const float* grad_score, float* grad_k, float* grad_q, int64_t H, int64_t dk, int64_t dout,
    const het_grouping* by_dst, const het_grouping* by_rel_src, int64_t n_q_rows, void* workspace,
    int64_t workspace_bytes, het_stream stream) {
  const char* op = "backward_hgt_full_graph_hetero_attention_ops_coo";
  if (int rc = check_edges(op, row, col, eids, rel_ptrs, num_rels, num_edges, 0)) return rc;
  HET_REQUIRE(H > 0 && dk > 0 && dout > 0 &&
                  (num_edges == 0 || (grad_w && weights_t && k && q && inner && grad_score && grad_k && grad_q)),
              "%s: null data pointer", op);
  if (num_edges == 0) return HET_OK;
  hipStream_t s = (hipStream_t)stream;
  const het_grouping *gd = by_dst, *gs = by_rel_src;
  if (gd && gs && gd->R == 0 && gd->E == num_edges && gd->p0 && gs->R == (int)num_rels && gs->E == num_edges && gs->p0 &&
      gs->p1 && rows_shape_ok(H, dout) && segment_sum_supported((int)(H * dout)) &&
      mfma_shape_supported((int)(H * dout), (int)(H * dk)) && mfma_dw_supported((int)(H * dk), (int)(H * dout)) &&
      workspace && workspace_bytes >= (int64_t)sizeof(float) * gs->S * H * dout && n_q_rows >= 0 &&
      (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (reinterpret_cast<uintptr_t>(inner) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_q) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(grad_k) & 15) == 0 && (reinterpret_cast<uintptr_t>(k) & 15) == 0) {
    // grad_q[dst] += SUM_{e into dst} gs[e,h] * inner[e,h,:]     (by_dst: payload0 = eids)
    if (int rc = launch_segment_sum(gd, inner, grad_q, (int)(H * dout), grad_score, s, (int)H, n_q_rows, 1, 0, /*nt_in=*/0)) return rc;
    // qs[(r,u)] = SUM over the out-edges of u in relation r of gs[e,h] * q[dst_e,h,:]   (payload0 = col, payload1 = eids)
    float* qs = static_cast<float*>(workspace);
    if (int rc = launch_segment_sum(gs, q, qs, (int)(H * dout), grad_score, s, (int)H)) return rc;
    MfmaGemmArgs m;  // grad_k[u] += qs[(r,u)] . blockdiag(Wt[r])
    m.A = qs; m.a_ld = H * dout; m.B = weights_t; m.b_rel_stride = H * dout * dk; m.b_headcat = 2; m.headcat_d = (int)dk;
    m.blockdiag_k = (int)dout; m.C = grad_k; m.c_ld = H * dk; m.scatter = gs->seg_key64; m.atomic = 1;
    m.seg_ptrs = gs->seg_rel_ptr64; m.num_segs = (int)num_rels; m.num_rows = gs->S; m.K = (int)(H * dout); m.X = (int)(H * dk);
    if (int rc = launch_seg_gemm_mfma(m, s)) return rc;
    MfmaDwArgs w;  // grad_w[r,h] += k[u,h,:]^T (x) qs[(r,u),h,:]
    w.A = k; w.a_ld = H * dk; w.gather = gs->seg_key64; w.G = qs; w.g_ld = H * dout; w.dW = grad_w;
    w.dw_rel_stride = H * dk * dout; w.headcat = 2; w.headcat_d = (int)dout; w.blockdiag_k = (int)dk;
    w.seg_ptrs = gs->seg_rel_ptr64; w.num_segs = (int)num_rels; w.num_rows = gs->S; w.K = (int)(H * dk); w.X = (int)(H * dout);
    return launch_seg_dw_mfma(w, s);
  }
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
