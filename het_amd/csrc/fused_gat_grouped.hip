// Destination-grouped RGAT kernels: one wave (64 lanes) per work item of a
// het_grouping by destination.  A feature row of X = H*D floats is covered by
// LPR = X/4 lanes holding a float4 each, so a wave streams 64/LPR edges per
// step; partial sums are combined with xor-shuffles and every destination row
// is written once -- no float atomics except for hub destinations whose
// segment was split over several items (> HET_ITEM_MAX in-edges).
#include <stdlib.h>

#include "coop.hip.h"
#include "fused_gat.hip.h"
#include "seg_reduce.hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kFoldRelMax = 8;  // relations the fused weight gradient of the folded GAT backward keeps in registers
constexpr int kFoldReplicas = 64;  // copies of that [R,X] gradient the workgroups spread their final atomics over

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// exp[eid,h] = leaky_exp(el[srow,h] + er[drow,h]) -- pure streaming, no sum.
__global__ __launch_bounds__(kBlock) void HET_gat_exp_edge(EdgeView v, RowMaps m, const float* __restrict__ el,
                                                            const float* __restrict__ er, float* __restrict__ exp,
                                                            int H, float slope) {
  const int64_t total = (int64_t)v.E * H, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
    const idx_t i = t / H;
    const int h = (int)(t - i * H);
    const idx_t eid = v.eids[i];
    idx_t srow = eid, drow = eid;
    if (m.kind != HET_KIND_DISABLED) {
      const bool need_nodes = m.kind == HET_KIND_ENABLED || m.kind == HET_KIND_DUAL_LIST;
      ev_rows(v, m, i, eid, need_nodes ? ev_src(v, i) : 0, need_nodes ? ev_dst(v, i) : 0, srow, drow);
    }
    exp[eid * H + h] = leaky_exp(el[srow * H + h] + er[drow * H + h], slope);
  }
}

struct Items {
  const int32_t *seg, *begin, *end, *seg_ptr, *seg_key;
  int64_t n;
};

// Wave per work item: the 64/LPR lane groups take the item's edges round-robin, U edges per group and step; the
// ids of the next step are fetched while the current rows are in flight (one dependent round trip per step).
// (A lane group per item was measured slower here: in-edge lists are skewed and a 256-edge item then runs serially.)
// EXPF: exp is formed here from el / er given in the grouping's order (two coalesced streams) instead of gathered by
// edge id from a separate pass' output; exp_edge (edge order) is written only when the caller asks for it.
// Occupancy (same-box A/B, ogbn-mag): 92 VGPRs = 5 waves/SIMD 1.73 ms; forced to 6 / 8 waves (the compiler then keeps
// fewer rows in flight per wave) 2.21 / 2.92 ms; capped below 5 waves with dynamic LDS 2.0-3.3 ms: the pass is bound by
// the bytes it keeps in flight and sits at the optimum of waves x rows per wave the register file allows.
// SLOT: a lane group (not a wave) per work item, 64/LPR items side by side -- for graphs whose destinations have few
// in-edges (a rank's share of a partition, a sampled block, AIFB: < 16 on average), where a wave per item idles.
template <int LPR, bool EXPF = false, bool SLOT = false>
__global__ __launch_bounds__(kBlock) void HET_gat_aggregate_grouped(Items it, const int32_t* __restrict__ p_eid,
                                                                     const int32_t* __restrict__ p_srow,
                                                                     const float* __restrict__ feat,
                                                                     const float* __restrict__ exp,
                                                                     float* __restrict__ sum, float* __restrict__ ret,
                                                                     float* __restrict__ exp_sorted, int H, int D,
                                                                     const float* __restrict__ el_sorted = nullptr,
                                                                     const float* __restrict__ er_sorted = nullptr,
                                                                     float* __restrict__ exp_edge = nullptr,
                                                                     float slope = 0.f) {
  constexpr int EPW = 64 / LPR, U = 4;
  constexpr int ST = SLOT ? 1 : EPW;  // rank stride between the edges of one lane group
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, x = (lane % LPR) * 4, h = x / D;
  const int j_off = SLOT ? 0 : slot;
  const int64_t wave_id = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t item = SLOT ? wave_id * EPW + slot : wave_id;
  if (item >= it.n) return;
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  const int64_t X = (int64_t)H * D;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float ssum = 0.f;
  int jn[U];
  int64_t eidn[U], srown[U];
#pragma unroll
  for (int u = 0; u < U; ++u) jn[u] = b + j_off + u * ST < e ? b + j_off + u * ST : e - 1;
#pragma unroll
  for (int u = 0; u < U; ++u) eidn[u] = p_eid[jn[u]];
  if (p_srow) {
#pragma unroll
    for (int u = 0; u < U; ++u) srown[u] = p_srow[jn[u]];
  }
  const int64_t v = it.seg_key[seg];
  const bool whole = b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1];
  for (int j0 = b + j_off; j0 < e; j0 += ST * U) {
    int jc[U];
    float w[U];
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) jc[u] = jn[u];
    float zl[U], zr[U];
    int64_t eidc[U];
    if (EXPF) {
#pragma unroll
      for (int u = 0; u < U; ++u) zl[u] = el_sorted[(int64_t)jc[u] * H + h];
#pragma unroll
      for (int u = 0; u < U; ++u) zr[u] = er_sorted[(int64_t)jc[u] * H + h];
#pragma unroll
      for (int u = 0; u < U; ++u) eidc[u] = eidn[u];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) w[u] = exp[eidn[u] * H + h];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = ld4(feat + (p_srow ? srown[u] : eidn[u]) * X + x);
#pragma unroll
    for (int u = 0; u < U; ++u) jn[u] = j0 + (U + u) * ST < e ? j0 + (U + u) * ST : e - 1;
#pragma unroll
    for (int u = 0; u < U; ++u) eidn[u] = p_eid[jn[u]];
    if (p_srow) {
#pragma unroll
      for (int u = 0; u < U; ++u) srown[u] = p_srow[jn[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = j0 + u * ST < e;
      if (EXPF) {
        w[u] = leaky_exp(zl[u] + zr[u], slope);
        if (exp_edge && ok && x % D == 0) exp_edge[eidc[u] * H + h] = w[u];
      }
      const float wu = ok ? w[u] : 0.f;
      if (exp_sorted && ok && x % D == 0) exp_sorted[(int64_t)jc[u] * H + h] = wu;
      acc.x = fmaf(wu, f[u].x, acc.x);
      acc.y = fmaf(wu, f[u].y, acc.y);
      acc.z = fmaf(wu, f[u].z, acc.z);
      acc.w = fmaf(wu, f[u].w, acc.w);
      ssum += wu;
    }
  }
  if (!SLOT) {
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      acc.x += __shfl_xor(acc.x, off);
      acc.y += __shfl_xor(acc.y, off);
      acc.z += __shfl_xor(acc.z, off);
      acc.w += __shfl_xor(acc.w, off);
      ssum += __shfl_xor(ssum, off);
    }
    if (slot != 0) return;
  }
  float* rp = ret + v * X + x;
  if (whole) {
    const float inv = 1.f / ssum;
    st4(rp, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
    if (x % D == 0) sum[v * H + h] = ssum;
  } else {  // hub destination: unnormalised partials, normalised by HET_gat_normalize_split
    atomicAdd(rp + 0, acc.x);
    atomicAdd(rp + 1, acc.y);
    atomicAdd(rp + 2, acc.z);
    atomicAdd(rp + 3, acc.w);
    if (x % D == 0) atomicAdd(&sum[v * H + h], ssum);
  }
}


// Cooperative form of HET_gat_aggregate_grouped<LPR, false> (exp gathered by edge id; the reference-named op for every
// kind): the edge id, the feat row and exp[eid, h] of the 4 edges of a step are fetched by the lanes (head h, d = edge) --
// 3 scalar instructions per step instead of 12 -- and spread with quad broadcasts.
template <int LPR, int DL>
__global__ __launch_bounds__(kBlock) void HET_gat_aggregate_grouped_coop(Items it, const int32_t* __restrict__ p_eid,
                                                                          const int32_t* __restrict__ p_srow,
                                                                          const float* __restrict__ feat,
                                                                          const float* __restrict__ exp,
                                                                          float* __restrict__ sum, float* __restrict__ ret,
                                                                          float* __restrict__ exp_sorted, int H) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4;
  static_assert(DL >= U, "a head needs at least U lanes");
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, d = sub % DL;
  const int dq = d < U ? d : U - 1;
  const int64_t item = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= it.n) return;
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float ssum = 0.f;
  int jn = b + slot + dq * EPW < e ? b + slot + dq * EPW : e - 1;
  int eidn = p_eid[jn], srown = p_srow ? p_srow[jn] : eidn;
  const int64_t v = it.seg_key[seg];
  const bool whole = b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1];
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    const int jc = jn, srowv = srown;
    const float wraw = exp[(int64_t)eidn * H + h];
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = ld4(feat + (int64_t)head_bcast_i<DL>(srowv, u, lane) * X + x);
    const bool okq = j0 + dq * EPW < e;
    jn = j0 + (U + dq) * EPW < e ? j0 + (U + dq) * EPW : e - 1;
    eidn = p_eid[jn];
    srown = p_srow ? p_srow[jn] : eidn;
    const float wv = okq ? wraw : 0.f;
    if (exp_sorted && okq && d < U) exp_sorted[(int64_t)jc * H + h] = wv;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = head_bcast<DL>(wv, u, lane);
      acc.x = fmaf(w, f[u].x, acc.x);
      acc.y = fmaf(w, f[u].y, acc.y);
      acc.z = fmaf(w, f[u].z, acc.z);
      acc.w = fmaf(w, f[u].w, acc.w);
      ssum += w;
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off);
    acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off);
    acc.w += __shfl_xor(acc.w, off);
    ssum += __shfl_xor(ssum, off);
  }
  if (slot != 0) return;
  float* rp = ret + v * X + x;
  if (whole) {
    const float inv = 1.f / ssum;
    st4(rp, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
    if (d == 0) sum[v * H + h] = ssum;
  } else {  // hub destination: unnormalised partials, normalised by HET_gat_normalize_split
    atomicAdd(rp + 0, acc.x);
    atomicAdd(rp + 1, acc.y);
    atomicAdd(rp + 2, acc.z);
    atomicAdd(rp + 3, acc.w);
    if (d == 0) atomicAdd(&sum[v * H + h], ssum);
  }
}

// exp[eid,h] for kind 0 with 4 heads: a thread per edge, 16-byte loads / stores (the generic kernel runs a thread per
// (edge, head) with scalar accesses: 0.33 ms for ogbn-mag's 21 M edges, this one streams them)
__global__ __launch_bounds__(kBlock) void HET_gat_exp_edge_h4(const idx_t* __restrict__ eids, int64_t E,
                                                               const float* __restrict__ el, const float* __restrict__ er,
                                                               float* __restrict__ exp, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < E; i += (int64_t)gridDim.x * kBlock) {
    const idx_t eid = eids[i];
    const float4 a = ld4(el + eid * 4), c = ld4(er + eid * 4);
    st4(exp + eid * 4, make_float4(leaky_exp(a.x + c.x, slope), leaky_exp(a.y + c.y, slope), leaky_exp(a.z + c.z, slope),
                                   leaky_exp(a.w + c.w, slope)));
  }
}

// out[i] += SUM over the replicas of part[rep][i]   (i < n: the R*X weight-gradient words of the folded backward)
__global__ __launch_bounds__(kBlock) void HET_gat_fold_w_reduce(const float* __restrict__ part, int replicas, int n,
                                                                 float* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float t = 0.f;
  for (int r = 0; r < replicas; ++r) t += part[(int64_t)r * n + i];
  out[i] += t;
}

__global__ __launch_bounds__(kBlock) void HET_gat_normalize_split(const int32_t* __restrict__ split_seg,
                                                                   const int32_t* __restrict__ seg_key,
                                                                   int64_t num_split, const float* __restrict__ sum,
                                                                   float* __restrict__ ret, int H, int D) {
  const int64_t X = (int64_t)H * D, total = num_split * X;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t k = t / X;
    const int x = (int)(t - k * X);
    const int64_t v = seg_key[split_seg[k]];
    ret[v * X + x] /= sum[v * H + x / D];
  }
}

// Backward for kind 0 (feat/el/er rows are edge rows): gradout[dst], ret[dst], sum[dst]
// are loaded once per destination; grad_feat / grad_el / grad_er rows are written
// exactly once with plain stores.  DL = D/4 lanes share a head.
// SORTED: exp is read from the by_dst-ordered copy the forward wrote (a coalesced stream) and the
// leaky-ReLU branch is recovered from it (slope >= 0: z > 0 <=> exp(leaky(z)) > 1), so el / er are
// not touched at all.
// FOLD: the caller formed el[e,h] = <feat[e,h,:], fold_w[r,h,:]> (r = relation of the edge's position, carried as
// payload1 of the grouping) and wants the gradient through that product added here:
// grad_feat[e,h,:] += grad_el[e,h] * fold_w[r,h,:].  Saves the separate read-modify-write pass over the [E,H,D]
// gradient.  Wave per item with id prefetch, like the forward.
// DW (with FOLD): also accumulates grad_fold_w[r,h,:] += SUM grad_el[e,h] * feat[e,h,:] -- the weight gradient of the
// folded product, from the feat rows this kernel reads anyway -- in per-relation registers (R <= kFoldRelMax); the grid
// is then a fixed number of workgroups striding over the items, each flushing once through LDS with R*X atomics.
template <int LPR, bool SORTED, bool FOLD, int RMAX = 0, bool SLOT = false>
__global__ __launch_bounds__(kBlock) void HET_gat_backward_grouped(
    Items it, const int32_t* __restrict__ p_eid, const float* __restrict__ feat, const float* __restrict__ el,
    const float* __restrict__ er, const float* __restrict__ sum, const float* __restrict__ exp,
    const float* __restrict__ ret, const float* __restrict__ gradout, float* __restrict__ grad_feat,
    float* __restrict__ grad_el, float* __restrict__ grad_er, int H, int D, float slope,
    const int32_t* __restrict__ p_rel, const float* __restrict__ fold_w, float* __restrict__ grad_fold_w, int R,
    float* __restrict__ grad_el_sorted, int replicas = 1) {
  constexpr int EPW = 64 / LPR, U = 2;  // same-box A/B: U = 1 3.89 ms, 2 3.76 ms, 4 3.82 ms
  constexpr int ST = SLOT ? 1 : EPW;     // SLOT: a lane group per item (short in-edge lists), see the forward
  constexpr int IPW = SLOT ? EPW : 1;    // items per wave and pass
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const int j_off = SLOT ? 0 : slot;
  const int64_t X = (int64_t)H * D;
  constexpr bool DW = RMAX > 0;
  float4 accw[DW ? RMAX : 1];
#pragma unroll
  for (int q = 0; q < (DW ? RMAX : 1); ++q) accw[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t item = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * IPW + (SLOT ? slot : 0); item < it.n;
       item += (int64_t)gridDim.x * (kBlock / 64) * IPW) {
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  int jn[U], reln[U];
  int64_t eidn[U];
#pragma unroll
  for (int u = 0; u < U; ++u) jn[u] = b + j_off + u * ST < e ? b + j_off + u * ST : e - 1;
#pragma unroll
  for (int u = 0; u < U; ++u) eidn[u] = p_eid[jn[u]];
  if (FOLD) {
#pragma unroll
    for (int u = 0; u < U; ++u) reln[u] = p_rel[jn[u]];
  }
  const int64_t v = it.seg_key[seg];
  const float4 g = ld4(gradout + v * X + x), r = ld4(ret + v * X + x);
  const float sinv = 1.f / sum[v * H + h];
  for (int j0 = b + j_off; j0 < e; j0 += ST * U) {
    int64_t eid[U];
    float ex[U], dl[U];
    float4 f[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) eid[u] = eidn[u];
    if (SORTED) {
#pragma unroll
      for (int u = 0; u < U; ++u) ex[u] = exp[(int64_t)jn[u] * H + h];
#pragma unroll
      for (int u = 0; u < U; ++u) f[u] = ld4_nt(feat + eid[u] * X + x);
    } else {
      float zl[U], zr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ex[u] = exp[eid[u] * H + h];
#pragma unroll
      for (int u = 0; u < U; ++u) zl[u] = el[eid[u] * H + h];
#pragma unroll
      for (int u = 0; u < U; ++u) zr[u] = er[eid[u] * H + h];
#pragma unroll
      for (int u = 0; u < U; ++u) f[u] = ld4_nt(feat + eid[u] * X + x);
#pragma unroll
      for (int u = 0; u < U; ++u) dl[u] = (zl[u] + zr[u]) > 0.f ? 1.f : slope;
    }
    int rlc[U];
    if (FOLD) {
#pragma unroll
      for (int u = 0; u < U; ++u) rlc[u] = reln[u];
#pragma unroll
      for (int u = 0; u < U; ++u) w[u] = ld4(fold_w + rlc[u] * X + x);
    }
    // ids of the next step (clamped: the last step re-reads its own)
#pragma unroll
    for (int u = 0; u < U; ++u) jn[u] = j0 + (U + u) * ST < e ? j0 + (U + u) * ST : e - 1;
#pragma unroll
    for (int u = 0; u < U; ++u) eidn[u] = p_eid[jn[u]];
    if (FOLD) {
#pragma unroll
      for (int u = 0; u < U; ++u) reln[u] = p_rel[jn[u]];
    }
    if (SORTED) {
#pragma unroll
      for (int u = 0; u < U; ++u) dl[u] = ex[u] > 1.f ? 1.f : slope;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = j0 + u * ST < e;  // uniform within the LPR lanes of a slot (shuffles below stay inside it)
      const float a = ex[u] * sinv;
      float tt = g.x * (f[u].x - r.x) + g.y * (f[u].y - r.y) + g.z * (f[u].z - r.z) + g.w * (f[u].w - r.w);
      for (int off = DL >> 1; off > 0; off >>= 1) tt += __shfl_xor(tt, off);
      tt *= a * dl[u];  // every lane of the head holds the edge's grad_el
      if (ok) {
        float4 o = make_float4(a * g.x, a * g.y, a * g.z, a * g.w);
        if (FOLD) {
          o.x = fmaf(tt, w[u].x, o.x); o.y = fmaf(tt, w[u].y, o.y);
          o.z = fmaf(tt, w[u].z, o.z); o.w = fmaf(tt, w[u].w, o.w);
        }
        st4_nt(grad_feat + eid[u] * X + x, o);
      }
      if (ok && (sub & (DL - 1)) == 0) {
        if (grad_el_sorted) grad_el_sorted[(int64_t)(j0 + u * ST) * H + h] = tt;  // rank order: sequential
        if (grad_el) grad_el[eid[u] * H + h] = tt;
        if (grad_er && grad_er != grad_el) grad_er[eid[u] * H + h] = tt;
      }
      if (DW) {
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
          const float sel = (ok && rlc[u] == q) ? tt : 0.f;
          accw[q].x = fmaf(sel, f[u].x, accw[q].x); accw[q].y = fmaf(sel, f[u].y, accw[q].y);
          accw[q].z = fmaf(sel, f[u].z, accw[q].z); accw[q].w = fmaf(sel, f[u].w, accw[q].w);
        }
      }
    }
  }
  }  // items
  if (DW) {
    __shared__ float4 part[kBlock / 64][DW ? RMAX : 1][64];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
      float4 t = accw[q];
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) {
        t.x += __shfl_xor(t.x, off); t.y += __shfl_xor(t.y, off);
        t.z += __shfl_xor(t.z, off); t.w += __shfl_xor(t.w, off);
      }
      part[wave][q][lane] = t;
    }
    __syncthreads();
    if (wave == 0 && slot == 0) {
      for (int q = 0; q < R && q < RMAX; ++q) {
        float4 t = part[0][q][lane];
        for (int wv = 1; wv < kBlock / 64; ++wv) {
          const float4 o = part[wv][q][lane];
          t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        // replicas > 1: the workgroups spread their flushes over that many copies (summed by HET_gat_fold_w_reduce)
        float* p = grad_fold_w + ((int64_t)(blockIdx.x % (unsigned)replicas) * R + q) * X + x;
        atomicAdd(p + 0, t.x); atomicAdd(p + 1, t.y); atomicAdd(p + 2, t.z); atomicAdd(p + 3, t.w);
      }
    }
  }
}


// Cooperative form of HET_gat_backward_grouped<LPR, false, false> (kind 0, exp / el / er gathered by edge id: the
// reference-named backward op called on its own): per step of 4 edges per lane group 1 id + 3 scalar loads, 4 feat
// rows in, 4 gradient rows out and 1-2 scalar stores instead of 32 instructions.
template <int LPR, int DL>
__global__ __launch_bounds__(kBlock) void HET_gat_backward_grouped_coop(
    Items it, const int32_t* __restrict__ p_eid, const float* __restrict__ feat, const float* __restrict__ el,
    const float* __restrict__ er, const float* __restrict__ sum, const float* __restrict__ exp,
    const float* __restrict__ ret, const float* __restrict__ gradout, float* __restrict__ grad_feat,
    float* __restrict__ grad_el, float* __restrict__ grad_er, int H, float slope) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4;
  static_assert(DL >= U, "a head needs at least U lanes");
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, d = sub % DL;
  const int dq = d < U ? d : U - 1;
  const int64_t item = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= it.n) return;
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  int jn = b + slot + dq * EPW < e ? b + slot + dq * EPW : e - 1;
  int eidn = p_eid[jn];
  const int64_t v = it.seg_key[seg];
  const float4 g = ld4(gradout + v * X + x), r = ld4(ret + v * X + x);
  const float sinv = 1.f / sum[v * H + h];
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    const int eidv = eidn;
    const float exv = exp[(int64_t)eidv * H + h];
    const float zlv = el[(int64_t)eidv * H + h];
    const float zrv = er[(int64_t)eidv * H + h];
    int eid[U];
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) eid[u] = head_bcast_i<DL>(eidv, u, lane);
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = ld4_nt(feat + (int64_t)eid[u] * X + x);
    const bool okq = j0 + dq * EPW < e;
    jn = j0 + (U + dq) * EPW < e ? j0 + (U + dq) * EPW : e - 1;
    eidn = p_eid[jn];
    const float av = exv * sinv;
    const float adv = av * ((zlv + zrv) > 0.f ? 1.f : slope);
    float tq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = j0 + u * EPW < e;  // uniform within the lane group
      const float a = head_bcast<DL>(av, u, lane), ad = head_bcast<DL>(adv, u, lane);
      float tt = g.x * (f[u].x - r.x) + g.y * (f[u].y - r.y) + g.z * (f[u].z - r.z) + g.w * (f[u].w - r.w);
#pragma unroll
      for (int off = DL >> 1; off > 0; off >>= 1) tt += __shfl_xor(tt, off);
      tq[u] = tt * ad;
      if (ok) st4_nt(grad_feat + (int64_t)eid[u] * X + x, make_float4(a * g.x, a * g.y, a * g.z, a * g.w));
    }
    float ts = tq[0];
#pragma unroll
    for (int u = 1; u < U; ++u) ts = d == u ? tq[u] : ts;
    if (okq && d < U) {  // lane (h, q) owns the (edge q, head h) scalars
      grad_el[(int64_t)eidv * H + h] = ts;
      if (grad_er != grad_el) grad_er[(int64_t)eidv * H + h] = ts;
    }
  }
}

// ---- backward for the compact kinds (feat / el rows shared by all out-edges of a (relation, source)) ----
// pack[v] = { 1/sum[v,h] (H floats), <gradout[v,h,:], ret[v,h,:]> (H floats) }: what an edge needs from its
// destination besides the gradout row, in one 2H-float record.
// Row-per-lane-group: X/4 lanes x float4 read the gradout and ret rows of a node coalesced, D/4-lane shuffle per head.
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_gat_dst_pack(const float* __restrict__ sum, const float* __restrict__ ret,
                                                            const float* __restrict__ gradout, float* __restrict__ pack,
                                                            int64_t N, int H, int D) {
  constexpr int EPW = 64 / LPR, X = LPR * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const int64_t step = (int64_t)gridDim.x * (kBlock / 64) * EPW;
  for (int64_t v0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * EPW; v0 < N; v0 += step) {
    const int64_t v = v0 + slot < N ? v0 + slot : N - 1;  // past the end: the last node again (same bytes rewritten)
    const float4 g = ld4(gradout + v * X + x), r = ld4(ret + v * X + x);
    float dot = g.x * r.x + g.y * r.y + g.z * r.z + g.w * r.w;
    for (int off = DL >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    if ((sub & (DL - 1)) == 0) {
      pack[v * 2 * H + h] = 1.f / sum[v * H + h];
      pack[v * 2 * H + H + h] = dot;
    }
  }
}

// Wave per work item of the grouping by feat row u (payload0 = edge id, payload1 = destination):
//   grad_feat[u,h,:] = SUM_e a_e * gradout[dst_e,h,:]
//   t_e = a_e * dl_e * (<gradout[dst_e,h,:], feat[u,h,:]> - <gradout, ret>[dst_e,h]);  grad_el[u,h] = SUM_e t_e
//   tbuf[eid_e,h] = t_e   (summed per er row by a second segmented pass)
// a_e = exp[eid,h] / sum[dst,h];  dl_e = exp > 1 ? 1 : slope  (slope >= 0: exp(leaky(z)) > 1 <=> z > 0)
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_gat_backward_src_grouped(
    Items it, const int32_t* __restrict__ p_eid, const int32_t* __restrict__ p_dst, const float* __restrict__ feat,
    const float* __restrict__ exp, const float* __restrict__ pack, const float* __restrict__ gradout,
    float* __restrict__ grad_feat, float* __restrict__ grad_el, float* __restrict__ tbuf, int H, int D, float slope,
    const float* __restrict__ fold_w, const idx_t* __restrict__ fold_row_rel_ptrs, int R) {
  constexpr int EPW = 64 / LPR, U = 1;  // same-box A/B on ogbn-mag: U = 1 2.22 ms, 2 2.29 ms, 4 3.00 ms
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= it.n) return;
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  const int64_t u = it.seg_key[seg];
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const int64_t X = (int64_t)H * D;
  int jn[U];
  int64_t eidn[U], dstn[U];
#pragma unroll
  for (int q = 0; q < U; ++q) jn[q] = b + slot + q * EPW < e ? b + slot + q * EPW : e - 1;
#pragma unroll
  for (int q = 0; q < U; ++q) eidn[q] = p_eid[jn[q]];
#pragma unroll
  for (int q = 0; q < U; ++q) dstn[q] = p_dst[jn[q]];
  const float4 f = ld4(feat + u * X + x);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float acc_el = 0.f;
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    int64_t eid[U], dst[U];
#pragma unroll
    for (int q = 0; q < U; ++q) { eid[q] = eidn[q]; dst[q] = dstn[q]; }
    float ex[U], sinv[U], gr[U];
    float4 g[U];
#pragma unroll
    for (int q = 0; q < U; ++q) ex[q] = exp[eid[q] * H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) sinv[q] = pack[dst[q] * 2 * H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) gr[q] = pack[dst[q] * 2 * H + H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) g[q] = ld4(gradout + dst[q] * X + x);
    // ids of the next step, fetched while this step's rows are in flight (clamped at the end)
#pragma unroll
    for (int q = 0; q < U; ++q) jn[q] = j0 + (U + q) * EPW < e ? j0 + (U + q) * EPW : e - 1;
#pragma unroll
    for (int q = 0; q < U; ++q) eidn[q] = p_eid[jn[q]];
#pragma unroll
    for (int q = 0; q < U; ++q) dstn[q] = p_dst[jn[q]];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const bool ok = j0 + q * EPW < e;  // uniform within a lane group
      const float a = ok ? ex[q] * sinv[q] : 0.f;
      acc.x = fmaf(a, g[q].x, acc.x); acc.y = fmaf(a, g[q].y, acc.y);
      acc.z = fmaf(a, g[q].z, acc.z); acc.w = fmaf(a, g[q].w, acc.w);
      float dot = g[q].x * f.x + g[q].y * f.y + g[q].z * f.z + g[q].w * f.w;
      for (int off = DL >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
      const float t = a * (ex[q] > 1.f ? 1.f : slope) * (dot - gr[q]);
      if (ok && (sub & (DL - 1)) == 0) tbuf[eid[q] * H + h] = t;
      acc_el += t;  // identical in the DL lanes of a head; one of them is kept below
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    acc_el += __shfl_xor(acc_el, off);
  }
  if (slot != 0) return;
  if (fold_w) {  // el[u,h] = <feat[u,h,:], fold_w[r(u),h,:]>: its gradient joins grad_feat here (linear: also per split item)
    const float4 w = ld4(fold_w + (int64_t)find_segment(fold_row_rel_ptrs, R, (idx_t)u) * X + x);
    acc.x = fmaf(acc_el, w.x, acc.x); acc.y = fmaf(acc_el, w.y, acc.y);
    acc.z = fmaf(acc_el, w.z, acc.z); acc.w = fmaf(acc_el, w.w, acc.w);
  }
  float* gp = grad_feat + u * X + x;
  if (b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1]) {
    st4(gp, acc);
    if ((sub & (DL - 1)) == 0) grad_el[u * H + h] = acc_el;
  } else {
    atomicAdd(gp + 0, acc.x); atomicAdd(gp + 1, acc.y); atomicAdd(gp + 2, acc.z); atomicAdd(gp + 3, acc.w);
    if ((sub & (DL - 1)) == 0) atomicAdd(&grad_el[u * H + h], acc_el);
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = ceil_div64(total, kBlock);
  const int64_t cap = 256 * 64;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline bool is_pow2(int64_t x) { return x > 0 && (x & (x - 1)) == 0; }

// shapes the grouped kernels cover: D a power of two >= 4, X/4 a power of two <= 64
inline bool grouped_shape_ok(int H, int D) {
  const int64_t X = (int64_t)H * D;
  return is_pow2(D) && D >= 4 && is_pow2(X) && X / 4 <= 64;
}

// Lane group per work item (64/LPR items side by side in a wave): feat rows have few out-edges each (5.7 on average on
// ogbn-mag by (relation, source)), so a wave per item leaves most lane groups idle.  Same math as the kernel above, no
// cross-group reduction.
template <int LPR, int U>
__global__ __launch_bounds__(kBlock) void HET_gat_backward_src_slot(
    Items it, const int32_t* __restrict__ p_eid, const int32_t* __restrict__ p_dst, const float* __restrict__ feat,
    const float* __restrict__ exp, const float* __restrict__ pack, const float* __restrict__ gradout,
    float* __restrict__ grad_feat, float* __restrict__ grad_el, float* __restrict__ tbuf, int H, int D, float slope,
    const float* __restrict__ fold_w, const idx_t* __restrict__ fold_row_rel_ptrs, int R) {
  constexpr int EPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const int64_t item = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * EPW + slot;
  if (item >= it.n) return;
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  const int64_t u = it.seg_key[seg];
  const int64_t X = (int64_t)H * D;
  int jn[U];
  int64_t eidn[U], dstn[U];
#pragma unroll
  for (int q = 0; q < U; ++q) jn[q] = b + q < e ? b + q : e - 1;
#pragma unroll
  for (int q = 0; q < U; ++q) eidn[q] = p_eid[jn[q]];
#pragma unroll
  for (int q = 0; q < U; ++q) dstn[q] = p_dst[jn[q]];
  const float4 f = ld4(feat + u * X + x);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float acc_el = 0.f;
  for (int j0 = b; j0 < e; j0 += U) {
    int64_t eid[U], dst[U];
#pragma unroll
    for (int q = 0; q < U; ++q) { eid[q] = eidn[q]; dst[q] = dstn[q]; }
    float ex[U], sinv[U], gr[U];
    float4 g[U];
#pragma unroll
    for (int q = 0; q < U; ++q) ex[q] = exp[eid[q] * H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) sinv[q] = pack[dst[q] * 2 * H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) gr[q] = pack[dst[q] * 2 * H + H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) g[q] = ld4(gradout + dst[q] * X + x);
#pragma unroll
    for (int q = 0; q < U; ++q) jn[q] = j0 + U + q < e ? j0 + U + q : e - 1;
#pragma unroll
    for (int q = 0; q < U; ++q) eidn[q] = p_eid[jn[q]];
#pragma unroll
    for (int q = 0; q < U; ++q) dstn[q] = p_dst[jn[q]];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const bool ok = j0 + q < e;  // uniform within the lane group
      const float a = ok ? ex[q] * sinv[q] : 0.f;
      acc.x = fmaf(a, g[q].x, acc.x); acc.y = fmaf(a, g[q].y, acc.y);
      acc.z = fmaf(a, g[q].z, acc.z); acc.w = fmaf(a, g[q].w, acc.w);
      float dot = g[q].x * f.x + g[q].y * f.y + g[q].z * f.z + g[q].w * f.w;
      for (int off = DL >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
      const float t = a * (ex[q] > 1.f ? 1.f : slope) * (dot - gr[q]);
      if (ok && (sub & (DL - 1)) == 0) tbuf[eid[q] * H + h] = t;
      acc_el += t;
    }
  }
  if (fold_w) {
    const float4 w = ld4(fold_w + (int64_t)find_segment(fold_row_rel_ptrs, R, (idx_t)u) * X + x);
    acc.x = fmaf(acc_el, w.x, acc.x); acc.y = fmaf(acc_el, w.y, acc.y);
    acc.z = fmaf(acc_el, w.z, acc.z); acc.w = fmaf(acc_el, w.w, acc.w);
  }
  float* gp = grad_feat + u * X + x;
  if (b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1]) {
    st4(gp, acc);
    if ((sub & (DL - 1)) == 0) grad_el[u * H + h] = acc_el;
  } else {
    atomicAdd(gp + 0, acc.x); atomicAdd(gp + 1, acc.y); atomicAdd(gp + 2, acc.z); atomicAdd(gp + 3, acc.w);
    if ((sub & (DL - 1)) == 0) atomicAdd(&grad_el[u * H + h], acc_el);
  }
}

}  // namespace

#define HET_DISPATCH_LPR(LPRV, CALL)           \
  switch (LPRV) {                              \
    case 1: { constexpr int LPR = 1; CALL; break; }   \
    case 2: { constexpr int LPR = 2; CALL; break; }   \
    case 4: { constexpr int LPR = 4; CALL; break; }   \
    case 8: { constexpr int LPR = 8; CALL; break; }   \
    case 16: { constexpr int LPR = 16; CALL; break; } \
    case 32: { constexpr int LPR = 32; CALL; break; } \
    default: { constexpr int LPR = 64; CALL; break; } \
  }

// (lanes per row, lanes per head) pairs the cooperative kernels are built for: rows of 32 / 64 / 128 floats, heads of >= 16
#define HET_DISPATCH_COOP_G(LPRV, DLV, CALL)                                \
  switch ((LPRV) * 64 + (DLV)) {                                            \
    case 8 * 64 + 4: { constexpr int LPR = 8, DL = 4; CALL; break; }        \
    case 8 * 64 + 8: { constexpr int LPR = 8, DL = 8; CALL; break; }        \
    case 16 * 64 + 4: { constexpr int LPR = 16, DL = 4; CALL; break; }      \
    case 16 * 64 + 8: { constexpr int LPR = 16, DL = 8; CALL; break; }      \
    case 16 * 64 + 16: { constexpr int LPR = 16, DL = 16; CALL; break; }    \
    case 32 * 64 + 4: { constexpr int LPR = 32, DL = 4; CALL; break; }      \
    case 32 * 64 + 8: { constexpr int LPR = 32, DL = 8; CALL; break; }      \
    case 32 * 64 + 16: { constexpr int LPR = 32, DL = 16; CALL; break; }    \
    default: { constexpr int LPR = 32, DL = 32; CALL; break; }              \
  }
static bool gat_coop_shape(int H, int D) {
  static const bool off = [] { const char* v = getenv("HET_GAT_COOP"); return v && v[0] == '0'; }();  // A/B switch
  const int lpr = H * D / 4, dl = D / 4;
  return !off && (lpr == 8 || lpr == 16 || lpr == 32) && dl >= 4 && dl <= lpr && D % 4 == 0;
}

// Destinations with few in-edges on average: a lane group per item instead of a wave per item (layer path kernels).
// Same-box A/B on one rank's share of an 8-way ogbn-mag partition (9 in-edges per destination): backward 0.528 -> 0.506 ms,
// forward unchanged.
static bool short_items(const het_grouping* g) { return g->E < 16 * g->num_items; }

int gat_forward_grouped(const het_grouping* g, const EdgeView& v, const RowMaps& m, const float* feat,
                        const float* el, const float* er, float* sum, float* exp, float* ret, float* exp_sorted,
                        int H, int D, float slope, const float* el_sorted, const float* er_sorted, hipStream_t s) {
  const bool have_rows = m.kind == HET_KIND_DISABLED || g->p1 != nullptr;
  if (!grouped_shape_ok(H, D) || !g->p0 || !have_rows || g->E != v.E || g->R != 0) {
    HET_REQUIRE(!exp_sorted && !el_sorted, "relational_fused_gat_separate_coo: exp_sorted / el_sorted need a shape the grouped kernels cover");
    return gat_forward_edge(v, m, feat, el, er, sum, exp, ret, H, D, slope, s);
  }
  const int64_t X = (int64_t)H * D;
  // destinations without in-edges keep zero rows; split (hub) destinations accumulate atomically
  HET_HIP(hipMemsetAsync(sum, 0, sizeof(float) * v.N * H, s));
  HET_HIP(hipMemsetAsync(ret, 0, sizeof(float) * v.N * X, s));
  if (v.E == 0) return HET_OK;
  Items it{g->item_seg, g->item_begin, g->item_end, g->seg_ptr, g->seg_key, g->num_items};
  const unsigned nb = (unsigned)ceil_div64(g->num_items, kBlock / 64);
  const int32_t* srow = m.kind == HET_KIND_DISABLED ? nullptr : g->p1;
  // (a persistent variant -- a fixed grid of waves striding over the items, the next item's record and first ids
  // prefetched during the current item's rows -- was 25-40 % slower at 2048..16384 workgroups, same box: the hardware's
  // own workgroup turnover already overlaps the per-item prologues; the pass runs at the rate random 256-byte rows
  // come out of HBM)
  if (!el_sorted) {
    if (m.kind == HET_KIND_DISABLED && H == 4 && ((reinterpret_cast<uintptr_t>(el) | reinterpret_cast<uintptr_t>(er) |
                                                  reinterpret_cast<uintptr_t>(exp)) & 15) == 0) {
      hipLaunchKernelGGL(HET_gat_exp_edge_h4, dim3(grid_for(v.E)), dim3(kBlock), 0, s, v.eids, v.E, el, er, exp, slope);
    } else {
      hipLaunchKernelGGL(HET_gat_exp_edge, dim3(grid_for(v.E * H)), dim3(kBlock), 0, s, v, m, el, er, exp, H, slope);
    }
    HET_LAUNCH_CHECK("HET_gat_exp_edge");
  }
  {
  HET_KTIME("HET_gat_aggregate_grouped", s);
  if (el_sorted && short_items(g)) {
    const unsigned nbs = (unsigned)ceil_div64(g->num_items, (int64_t)(kBlock / 64) * (64 / (X / 4)));
    HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL((HET_gat_aggregate_grouped<LPR, true, true>), dim3(nbs), dim3(kBlock), 0,
                                                      s, it, g->p0, srow, feat, (const float*)nullptr, sum, ret, exp_sorted,
                                                      H, D, el_sorted, er_sorted, exp, slope));
  } else if (el_sorted) {
    HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL((HET_gat_aggregate_grouped<LPR, true>), dim3(nb), dim3(kBlock), 0, s,
                                                      it, g->p0, srow, feat, (const float*)nullptr, sum, ret, exp_sorted,
                                                      H, D, el_sorted, er_sorted, exp, slope));
  } else if (gat_coop_shape(H, D)) {
    HET_DISPATCH_COOP_G((int)(X / 4), D / 4,
                        hipLaunchKernelGGL((HET_gat_aggregate_grouped_coop<LPR, DL>), dim3(nb), dim3(kBlock), 0, s, it, g->p0,
                                           srow, feat, exp, sum, ret, exp_sorted, H));
  } else {
    HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL((HET_gat_aggregate_grouped<LPR, false>), dim3(nb), dim3(kBlock), 0, s,
                                                      it, g->p0, srow, feat, exp, sum, ret, exp_sorted, H, D));
  }
  }
  HET_LAUNCH_CHECK("HET_gat_aggregate_grouped");
  if (g->num_split > 0) {
    hipLaunchKernelGGL(HET_gat_normalize_split, dim3(grid_for(g->num_split * X)), dim3(kBlock), 0, s, g->split_seg,
                       g->seg_key, g->num_split, sum, ret, H, D);
    HET_LAUNCH_CHECK("HET_gat_normalize_split");
  }
  return HET_OK;
}

bool gat_backward_fold_supported(const het_grouping* g, const EdgeView& v, const RowMaps& m, int H, int D) {
  return g && m.kind == HET_KIND_DISABLED && grouped_shape_ok(H, D) && g->p0 && g->p1 && g->E == v.E && g->R == 0;
}

int gat_backward_grouped(const het_grouping* g, const EdgeView& v, const RowMaps& m, const float* feat,
                         const float* el, const float* er, const float* sum, const float* exp, const float* ret,
                         const float* exp_sorted, const float* gradout, float* grad_feat, float* grad_el,
                         float* grad_er, int H, int D, float slope, const float* fold_w, float* grad_fold_w,
                         float* grad_el_sorted, float* workspace, int64_t workspace_bytes, hipStream_t s) {
  HET_REQUIRE(!grad_fold_w || (fold_w && v.R <= kFoldRelMax), "backward_relational_fused_gat_separate_coo: grad_fold_attn_l needs fold_attn_l and at most %d relations", kFoldRelMax);
  if (m.kind != HET_KIND_DISABLED || !grouped_shape_ok(H, D) || !g->p0 || g->E != v.E || g->R != 0) {
    HET_REQUIRE(!fold_w && !grad_el_sorted && (v.E == 0 || (el && er && exp)), "backward_relational_fused_gat_separate_coo: fold_attn_l / grad_el_sorted / NULL el, er, exp need the destination-grouped path");
    return gat_backward_edge(v, m, feat, el, er, sum, exp, ret, gradout, grad_feat, grad_el, grad_er, H, D, slope, s);
  }
  if (v.E == 0) return HET_OK;
  const int64_t X = (int64_t)H * D;
  Items it{g->item_seg, g->item_begin, g->item_end, g->seg_ptr, g->seg_key, g->num_items};
  HET_REQUIRE(!fold_w || g->p1, "backward_relational_fused_gat_separate_coo: fold_attn_l needs payload1 = relation");
  const unsigned nb = (unsigned)ceil_div64(g->num_items, kBlock / 64);
  const bool sorted = exp_sorted && slope >= 0.f;
  const float* ex = sorted ? exp_sorted : exp;
  HetKTimer* kt = nullptr;  // closed right after the kernel launch (before the tiny replica reduction)
#define HET_GAT_BWD(SORTED, FOLD)                                                                                    \
  HET_DISPATCH_LPR((int)(X / 4),                                                                                     \
                   hipLaunchKernelGGL((HET_gat_backward_grouped<LPR, SORTED, FOLD>), dim3(nb), dim3(kBlock), 0, s, it, \
                                      g->p0, feat, el, er, sum, ex, ret, gradout, grad_feat, grad_el, grad_er, H, D,  \
                                      slope, g->p1, fold_w, (float*)nullptr, v.R, grad_el_sorted))
  if (grad_fold_w) {
    // fixed grid striding over the items: every workgroup flushes R*X atomics once (same-box A/B, exp/ab_bwd.sh:
    // 2048 .. 16384 workgroups and U = 1 / 2 all within +-3 %; 1280 workgroups 25 % slower; one workgroup per 4 items
    // 5x slower -- the atomics on the R*X words serialise).  With the replicas below 1024 .. 16384 workgroups are within
    // 5 % at 1/8 of the graph too (0.43 ms; 0.51 ms at 4096 and 1.6 ms at 16384 workgroups without them)
    const unsigned nbw = nb < 4096u ? nb : 4096u;
    // every workgroup ends with R*X atomic adds: onto kFoldReplicas copies in the workspace when the caller gave one
    // (all workgroups of a small graph finish together and serialise on the R*X words otherwise)
    const int n_w = v.R * (int)X;
    int replicas = 1;
    float* dw_out = grad_fold_w;
    if (workspace && workspace_bytes >= (int64_t)sizeof(float) * kFoldReplicas * n_w) {
      replicas = kFoldReplicas;
      dw_out = workspace;
      HET_HIP(hipMemsetAsync(dw_out, 0, sizeof(float) * replicas * n_w, s));
    }
    if (het_ktime_on()) kt = new HetKTimer("HET_gat_backward_grouped", s);
#define HET_GAT_BWD_DW(SORTED, RM, SLOT)                                                                                 \
  HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL((HET_gat_backward_grouped<LPR, SORTED, true, RM, SLOT>), dim3(nbw),    \
                                                    dim3(kBlock), 0, s, it, g->p0, feat, el, er, sum, ex, ret, gradout, \
                                                    grad_feat, grad_el, grad_er, H, D, slope, g->p1, fold_w,            \
                                                    dw_out, v.R, grad_el_sorted, replicas))
    if (sorted && short_items(g)) {  // the layer path on a low-degree graph: lane group per item
      if (v.R <= 4) { HET_GAT_BWD_DW(true, 4, true); } else { HET_GAT_BWD_DW(true, 8, true); }
    } else if (v.R <= 4) {
      if (sorted) { HET_GAT_BWD_DW(true, 4, false); } else { HET_GAT_BWD_DW(false, 4, false); }
    } else {
      if (sorted) { HET_GAT_BWD_DW(true, 8, false); } else { HET_GAT_BWD_DW(false, 8, false); }
    }
#undef HET_GAT_BWD_DW
    delete kt;
    kt = nullptr;
    if (replicas > 1) {
      HET_LAUNCH_CHECK("HET_gat_backward_grouped");
      hipLaunchKernelGGL(HET_gat_fold_w_reduce, dim3((n_w + kBlock - 1) / kBlock), dim3(kBlock), 0, s, dw_out, replicas, n_w,
                         grad_fold_w);
    }
  } else if (fold_w) {
    HET_KTIME("HET_gat_backward_grouped", s);
    if (sorted) { HET_GAT_BWD(true, true); } else { HET_GAT_BWD(false, true); }
  } else if (!sorted && !grad_el_sorted && grad_el && grad_er && gat_coop_shape(H, D)) {
    HET_KTIME("HET_gat_backward_grouped", s);
    HET_DISPATCH_COOP_G((int)(X / 4), D / 4,
                        hipLaunchKernelGGL((HET_gat_backward_grouped_coop<LPR, DL>), dim3(nb), dim3(kBlock), 0, s, it, g->p0,
                                           feat, el, er, sum, ex, ret, gradout, grad_feat, grad_el, grad_er, H, slope));
  } else {
    HET_KTIME("HET_gat_backward_grouped", s);
    if (sorted) { HET_GAT_BWD(true, false); } else { HET_GAT_BWD(false, false); }
  }
#undef HET_GAT_BWD
  HET_LAUNCH_CHECK("HET_gat_backward_grouped");
  return HET_OK;
}

int gat_backward_compact_grouped(const het_grouping* by_srow, const het_grouping* by_drow, const EdgeView& v,
                                 int64_t n_src_rows, int64_t n_dst_rows, const float* feat, const float* sum,
                                 const float* exp, const float* ret, const float* gradout, float* grad_feat,
                                 float* grad_el, float* grad_er, float* workspace, int H, int D, float slope,
                                 const float* fold_w, const idx_t* fold_row_rel_ptrs, hipStream_t s) {
  HET_REQUIRE(!fold_w || fold_row_rel_ptrs, "backward_relational_fused_gat_separate_coo: fold_attn_l on compact rows needs their relation pointers");
  if (v.E == 0) return HET_OK;
  const int64_t X = (int64_t)H * D;
  float* pack = workspace;               // [N, 2H]
  float* tbuf = workspace + v.N * 2 * H; // [E, H]
  HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL(HET_gat_dst_pack<LPR>, dim3(grid_for(v.N * (X / 4))), dim3(kBlock), 0, s,
                                                    sum, ret, gradout, pack, v.N, H, D));
  HET_LAUNCH_CHECK("HET_gat_dst_pack");
  // rows whose segment was split accumulate atomically: start them (all) from zero
  HET_HIP(hipMemsetAsync(grad_el, 0, sizeof(float) * n_src_rows * H, s));
  if (by_srow->num_split > 0 || by_srow->S != n_src_rows)
    HET_HIP(hipMemsetAsync(grad_feat, 0, sizeof(float) * n_src_rows * X, s));
  Items it{by_srow->item_seg, by_srow->item_begin, by_srow->item_end, by_srow->seg_ptr, by_srow->seg_key,
           by_srow->num_items};
  if (by_srow->E < 16 * by_srow->num_items) {
    // short segments (5.7 edges per (relation, source) row on ogbn-mag): a lane group per item, two edges in flight
    // (same box: wave per item 3.4-3.5 ms for the whole op, lane group per item with U = 1 / 2 / 4: 3.7 / 3.1 / 3.65 ms)
    const unsigned nbs = (unsigned)ceil_div64(by_srow->num_items, (int64_t)(kBlock / 64) * (64 / (X / 4)));
    HET_KTIME("HET_gat_backward_src", s);
    HET_DISPATCH_LPR((int)(X / 4),
                     hipLaunchKernelGGL((HET_gat_backward_src_slot<LPR, 2>), dim3(nbs), dim3(kBlock), 0, s, it, by_srow->p0,
                                        by_srow->p1, feat, exp, pack, gradout, grad_feat, grad_el, tbuf, H, D, slope, fold_w,
                                        fold_row_rel_ptrs, v.R));
    HET_LAUNCH_CHECK("HET_gat_backward_src_slot");
  } else {
    const unsigned nb = (unsigned)ceil_div64(by_srow->num_items, kBlock / 64);
    HET_KTIME("HET_gat_backward_src", s);
    HET_DISPATCH_LPR((int)(X / 4),
                     hipLaunchKernelGGL(HET_gat_backward_src_grouped<LPR>, dim3(nb), dim3(kBlock), 0, s, it, by_srow->p0,
                                        by_srow->p1, feat, exp, pack, gradout, grad_feat, grad_el, tbuf, H, D, slope, fold_w,
                                        fold_row_rel_ptrs, v.R));
    HET_LAUNCH_CHECK("HET_gat_backward_src_grouped");
  }
  // grad_er[w, :] = SUM over the edges of er row w of tbuf[eid, :]   (segments of by_drow are the er rows in order; when some
  // er rows have no edge -- a two-sided unique list: rows that are sources only -- every segment lands in the row of its key
  // and the others read zero)
  return launch_segment_sum(by_drow, tbuf, grad_er, H, nullptr, s, 0, by_drow->S == n_dst_rows ? -1 : n_dst_rows);
}

bool gat_backward_compact_supported(const het_grouping* by_srow, const het_grouping* by_drow, int64_t E,
                                    int64_t n_dst_rows, int H, int D, float slope) {
  return by_srow && by_drow && grouped_shape_ok(H, D) && segment_rows_supported(H) && slope >= 0.f && by_srow->E == E &&
         by_drow->E == E && by_srow->R == 0 && by_drow->R == 0 && by_srow->p0 && by_srow->p1 && by_drow->p0 &&
         by_drow->S <= n_dst_rows && by_drow->key_bound <= n_dst_rows;
}
