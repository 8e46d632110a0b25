// HGT attention + message aggregation on the distinct (relation, source) rows: three gather passes, no per-edge tensor.
//
// The reference's HGT layer (hrt/python/HGT/models.py:120-286) runs, per step, typed K / Q / V projections of the nodes,
// a per-relation product + inner product for the score [E,H], an edge softmax (HGT/EdgeSoftmaxKernelsSeparateCOO /
// hrt/include/DGLHackKernel/HGT/*.cu.h) and a per-relation message product fused with the weighted aggregation -- each
// reading and writing [E,H] or [E,H,dk] tensors.  Every per-edge quantity of that chain is a function of TWO rows only:
//
//   s_e[h]   = < k'[srow_e, h, :], q[dst_e, h, :] >          k'[(r,u)] = k[u] . att[r] . pri[r] / sqrt(dk)   (row of the pair)
//   msg_e    = m[srow_e]                                      m[(r,u)]  = v[u] . msg[r]
//
// where srow_e is the row of (relation_e, source_e) in the graph's unique (relation, source) list -- the reference's
// compact dataflow (--compact_as_of_node_flag) applied to the source side of both products.  kv_c [S_row, 2, H, D] holds
// k' and m of a pair next to each other (one 2*H*D*4-byte gather per edge), q [N,H,D] the destination side.  Then
//
//   forward   (by destination)   w_e = exp(s_e);  lsum[v,h] = SUM w_e;  out[v,h,:] = SUM w_e m[srow_e,h,:] / lsum[v,h]
//   backward  (by destination)   a_e = w_e / lsum;  ga_e = <gradout[v,h,:], m[srow_e,h,:]>;  gs_e = a_e (ga_e - <gradout, out>[v,h])
//                                grad_q[v,h,:] = SUM gs_e k'[srow_e,h,:];   pack2[v,h] = {1/lsum, <gradout, out>}
//             (by source row u)  s_e, a_e, ga_e, gs_e again from k'[u], m[u] (loaded once per row) and q[dst_e], gradout[dst_e],
//                                pack2[dst_e]:  grad_k'[u,h,:] = SUM gs_e q[dst_e,h,:];   grad_m[u,h,:] = SUM a_e gradout[dst_e,h,:]
//
// exp without a running maximum, as the reference's edge softmax computes it (HGT/models.py:243-262 -> oracle/layers.py
// hgt_layer: m = exp(s * mu), a = m / SUM m).  Work distribution as in gat_compact.hip: wave per destination work item;
// lane group per pack of short source-row segments, wave per work item of the long ones; edge ids fetched by one lane of
// every quad for a 4-edge step and shared with DPP broadcasts (coop.hip.h).
#include <stdlib.h>

#include "coop.hip.h"
#include "grouping.hip.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float4& acc, float w, float4 v) {
  acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}
__device__ __forceinline__ void atomic_add4(float* p, float4 v) {
  atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}

__device__ __forceinline__ int quad_bcast(int v, int q) {  // value of lane q of the caller's quad; q is a constant after unrolling
  switch (q) {
    case 0: return quad_bcast_i<0>(v);
    case 1: return quad_bcast_i<1>(v);
    case 2: return quad_bcast_i<2>(v);
    default: return quad_bcast_i<3>(v);
  }
}

struct Items {
  const int32_t *seg, *begin, *end, *seg_ptr, *seg_key;
  int64_t n;
};
struct Packs {
  const int32_t *ptr, *key;
  int64_t n;
};

// Forward: wave per destination work item, the 64/LPR lane groups take its edges round-robin, U = 4 rows per group in
// flight, the ids of the next step prefetched.
template <int LPR, int DL>
__global__ __launch_bounds__(kBlock) void HET_hgt_aggregate_rows(Items it, const int32_t* __restrict__ p_srow,
                                                                  const float* __restrict__ kv, const float* __restrict__ q,
                                                                  float* __restrict__ lsum, float* __restrict__ out, int H,
                                                                  float* __restrict__ part) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4;
  static_assert(DL <= LPR, "heads inside the lane group");  // (LPR >= 4: ids shared by quads; LPR == 2: every lane loads its ids)
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, d = sub % DL, qd = sub & 3;
  const int64_t item = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= it.n) return;
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  int sidn[U];  // ids of the next step: one per lane and quad-shared (LPR >= 4), or all U in every lane
  if constexpr (LPR >= 4) {
    const int jn = b + slot + qd * EPW < e ? b + slot + qd * EPW : e - 1;
    sidn[0] = p_srow[jn];
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) sidn[u] = p_srow[b + slot + u * EPW < e ? b + slot + u * EPW : e - 1];
  }
  const int64_t v = it.seg_key[seg];
  const bool whole = b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1];
  const float4 q4 = ld4(q + v * X + x);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // softmax relative to a running maximum of the scores (the reference exponentiates the raw score, HGT/models.py via
  // hgt_full_graph_edge_softmax_ops: exp(score * mu) -- finite only while |score| < 88): rescale when it grows, keep
  // lse = max + log(sum) where the reference-named op keeps the sum
  float ssum = 0.f, m = -INFINITY;
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    float4 kk[U], mm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int sid;
      if constexpr (LPR >= 4) sid = quad_bcast(sidn[0], u); else sid = sidn[u];
      const float* rowp = kv + (int64_t)sid * (2 * X) + x;
      kk[u] = ld4(rowp);
      mm[u] = ld4(rowp + X);
    }
    if constexpr (LPR >= 4) {
      const int jn = j0 + (U + qd) * EPW < e ? j0 + (U + qd) * EPW : e - 1;
      sidn[0] = p_srow[jn];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) sidn[u] = p_srow[j0 + (U + u) * EPW < e ? j0 + (U + u) * EPW : e - 1];
    }
    float sc[U], mn = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sc[u] = j0 + u * EPW < e ? head_sum<DL>(dot4(kk[u], q4)) : -INFINITY;  // (uniform within the lanes of a head)
      mn = fmaxf(mn, sc[u]);
    }
    const float c = __expf(m - mn);  // (the first edge of a step exists: mn is finite; exp(-inf) = 0 the first time)
    acc.x *= c; acc.y *= c; acc.z *= c; acc.w *= c; ssum *= c;
    m = mn;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = __expf(sc[u] - m);  // 0 for the padding edges
      fma4(acc, w, mm[u]);
      ssum += w;
    }
  }
  {  // the lane groups of the wave saw different edges: bring their sums to the common maximum
    float M = m;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) M = fmaxf(M, __shfl_xor(M, off));
    const float c = m == -INFINITY ? 0.f : __expf(m - M);
    acc.x *= c; acc.y *= c; acc.z *= c; acc.w *= c; ssum *= c;
    m = M;
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    ssum += __shfl_xor(ssum, off);
  }
  if (slot != 0) return;
  if (whole) {
    const float inv = 1.f / ssum;
    st4(out + v * X + x, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
    if (d == 0) lsum[v * H + h] = m + __logf(ssum);
  } else {  // a piece of a hub destination: parked {acc[X], max[H], sum[H]} for HET_hgt_finish_split
    float* pp = part + item * (X + 2 * H);
    st4(pp + x, acc);
    if (d == 0) { pp[X + h] = m; pp[X + H + h] = ssum; }
  }
}

// One wave per split (hub) destination: its work items' parked partial sums brought to the common maximum, divided, stored.
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_hgt_finish_split(const int32_t* __restrict__ split_seg, int64_t num_split, Items it,
                                                                const float* __restrict__ part, float* __restrict__ lse,
                                                                float* __restrict__ out, int H, int D) {
  constexpr int EPW = 64 / LPR;
  const int lane = threadIdx.x & 63, slot = lane / LPR, x = (lane % LPR) * 4, h = x / D;
  const int64_t k = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (k >= num_split) return;
  const int seg = split_seg[k];
  const int64_t X = (int64_t)H * D, v = it.seg_key[seg];
  int64_t lo = 0, hi = it.n;  // first work item of the segment (items are in segment order)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (it.seg[mid] < seg) lo = mid + 1; else hi = mid;
  }
  const int64_t n_items = (it.seg_ptr[seg + 1] - it.seg_ptr[seg] + HET_ITEM_MAX - 1) / HET_ITEM_MAX;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float ssum = 0.f, m = -INFINITY;
  for (int64_t i = slot; i < n_items; i += EPW) {
    const float* pp = part + (lo + i) * (X + 2 * H);
    const float mi = pp[X + h], si = pp[X + H + h];
    const float4 a = ld4(pp + x);
    const float mn = fmaxf(m, mi), c = __expf(m - mn), ci = __expf(mi - mn);
    acc.x = acc.x * c + a.x * ci; acc.y = acc.y * c + a.y * ci; acc.z = acc.z * c + a.z * ci; acc.w = acc.w * c + a.w * ci;
    ssum = ssum * c + si * ci;
    m = mn;
  }
  float M = m;
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) M = fmaxf(M, __shfl_xor(M, off));
  const float c = m == -INFINITY ? 0.f : __expf(m - M);
  acc.x *= c; acc.y *= c; acc.z *= c; acc.w *= c; ssum *= c;
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    ssum += __shfl_xor(ssum, off);
  }
  if (slot != 0) return;
  const float inv = 1.f / ssum;
  st4(out + v * X + x, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
  if (x % D == 0) lse[v * H + h] = M + __logf(ssum);
}

// rows[key of segment list[k]] (row_floats wide) = 0: the rows several work items add to
__global__ __launch_bounds__(kBlock) void HET_hgt_zero_rows(const int32_t* __restrict__ list, const int32_t* __restrict__ seg_key,
                                                             int64_t n, float* __restrict__ rows, int row_floats) {
  const int per = row_floats / 4;
  const int64_t total = n * per;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t k = t / per;
    st4(rows + (int64_t)seg_key[list[k]] * row_floats + (t - k * per) * 4, make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

__global__ __launch_bounds__(kBlock) void HET_hgt_normalize_rows(const int32_t* __restrict__ list, const int32_t* __restrict__ seg_key,
                                                                  int64_t n, const float* __restrict__ lsum, float* __restrict__ out,
                                                                  int H, int D) {
  const int X = H * D, per = X / 4;
  const int64_t total = n * per;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t k = t / per;
    const int x = (int)(t - k * per) * 4;
    const int64_t v = seg_key[list[k]];
    const float inv = 1.f / lsum[v * H + x / D];
    float4 r = ld4(out + v * X + x);
    st4(out + v * X + x, make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv));
  }
}

// Backward, destination side: same schedule as the forward.
template <int LPR, int DL>
__global__ __launch_bounds__(kBlock) void HET_hgt_backward_dst_rows(Items it, const int32_t* __restrict__ p_srow,
                                                                     const float* __restrict__ kv, const float* __restrict__ q,
                                                                     const float* __restrict__ lsum, const float* __restrict__ out,
                                                                     const float* __restrict__ gradout, float* __restrict__ grad_q,
                                                                     float* __restrict__ pack2, int H) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4;
  static_assert(DL <= LPR, "heads inside the lane group");  // (LPR >= 4: ids shared by quads; LPR == 2: every lane loads its ids)
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, d = sub % DL, qd = sub & 3;
  const int64_t item = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= it.n) return;
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  int sidn[U];  // ids of the next step: one per lane and quad-shared (LPR >= 4), or all U in every lane
  if constexpr (LPR >= 4) {
    const int jn = b + slot + qd * EPW < e ? b + slot + qd * EPW : e - 1;
    sidn[0] = p_srow[jn];
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) sidn[u] = p_srow[b + slot + u * EPW < e ? b + slot + u * EPW : e - 1];
  }
  const int64_t v = it.seg_key[seg];
  const bool whole = b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1];
  const float4 q4 = ld4(q + v * X + x), go = ld4(gradout + v * X + x), o4 = ld4(out + v * X + x);
  const float lse_v = lsum[v * H + h];  // log-sum-exp of the destination (forward)
  const float dotn = head_sum<DL>(dot4(go, o4));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    float4 kk[U], mm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int sid;
      if constexpr (LPR >= 4) sid = quad_bcast(sidn[0], u); else sid = sidn[u];
      const float* rowp = kv + (int64_t)sid * (2 * X) + x;
      kk[u] = ld4(rowp);
      mm[u] = ld4(rowp + X);
    }
    if constexpr (LPR >= 4) {
      const int jn = j0 + (U + qd) * EPW < e ? j0 + (U + qd) * EPW : e - 1;
      sidn[0] = p_srow[jn];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) sidn[u] = p_srow[j0 + (U + u) * EPW < e ? j0 + (U + u) * EPW : e - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float s = head_sum<DL>(dot4(kk[u], q4));
      const float a = j0 + u * EPW < e ? __expf(s - lse_v) : 0.f;
      const float ga = head_sum<DL>(dot4(go, mm[u]));
      fma4(acc, a * (ga - dotn), kk[u]);
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
  }
  if (slot != 0) return;
  if (whole) {
    st4_nt(grad_q + v * X + x, acc);
  } else {
    atomic_add4(grad_q + v * X + x, acc);
  }
  if (d == 0) *reinterpret_cast<float2*>(pack2 + (v * H + h) * 2) = make_float2(lse_v, dotn);  // (the same from every item of v)
}

// Backward, source side, SHORT segments: lane group per pack of whole (relation, source) segments.
template <int LPR, int DL>
__global__ __launch_bounds__(kBlock) void HET_hgt_backward_src_short(Packs pk, const int4* __restrict__ kp01,
                                                                      const float* __restrict__ kv, const float* __restrict__ q,
                                                                      const float* __restrict__ pack2,
                                                                      const float* __restrict__ gradout,
                                                                      float* __restrict__ grad_kv, int H) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4;
  static_assert(DL <= LPR, "heads inside the lane group");  // (LPR >= 4: ids shared by quads; LPR == 2: every lane loads its ids)
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, qd = sub & 3;
  const int64_t pid = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * EPW + slot;
  if (pid >= pk.n) return;
  const uint32_t pb = (uint32_t)pk.ptr[pid];
  const int b = (int)(pb & 0x7fffffffu), e = (int)((uint32_t)pk.ptr[pid + 1] & 0x7fffffffu);
  if (pb >> 31) return;  // a long segment: HET_hgt_backward_src_long takes its work items
  int4 idn[U];  // {source row, destination, -, -} of the edges of the next step (grouping_packed_ids): quad-shared or per lane
  if constexpr (LPR >= 4) {
    idn[0] = kp01[b + qd < e ? b + qd : e - 1];
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) idn[u] = kp01[b + u < e ? b + u : e - 1];
  }
  int prev_key = -1;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 kcur = zero, mcur = zero, acck = zero, accm = zero;
  for (int j0 = b; j0 < e; j0 += U) {
    int key[U], dsti[U];
    float4 qr[U], gr[U], kq[U], mq[U];
    float2 p2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (LPR >= 4) { key[u] = quad_bcast(idn[0].x, u); dsti[u] = quad_bcast(idn[0].y, u); }
      else { key[u] = idn[u].x; dsti[u] = idn[u].y; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t dv = dsti[u];
      qr[u] = ld4(q + dv * X + x);
      gr[u] = ld4(gradout + dv * X + x);
      p2[u] = *reinterpret_cast<const float2*>(pack2 + (dv * H + h) * 2);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {  // rows of a segment that starts inside this step (uniform per lane group)
      kq[u] = zero;
      mq[u] = zero;
      if (j0 + u < e && key[u] != (u == 0 ? prev_key : key[u - 1])) {
        const float* rowp = kv + (int64_t)key[u] * (2 * X) + x;
        kq[u] = ld4(rowp);
        mq[u] = ld4(rowp + X);
      }
    }
    int key_after;
    if constexpr (LPR >= 4) {
      idn[0] = kp01[j0 + U + qd < e ? j0 + U + qd : e - 1];
      key_after = quad_bcast_i<0>(idn[0].x);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) idn[u] = kp01[j0 + U + u < e ? j0 + U + u : e - 1];
      key_after = idn[0].x;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = j0 + u < e;  // uniform within the lane group
      if (ok && key[u] != (u == 0 ? prev_key : key[u - 1])) {
        kcur = kq[u];
        mcur = mq[u];
      }
      const float s = head_sum<DL>(dot4(kcur, qr[u]));
      const float a = ok ? __expf(s - p2[u].x) : 0.f;  // p2.x: lse of the destination
      const float ga = head_sum<DL>(dot4(gr[u], mcur));
      fma4(acck, a * (ga - p2[u].y), qr[u]);
      fma4(accm, a, gr[u]);
      const int key_next = u + 1 < U ? key[u + 1] : key_after;
      if (ok && (j0 + u == e - 1 || key_next != key[u])) {  // the segment ends: one store of its two rows
        float* gp = grad_kv + (int64_t)key[u] * (2 * X) + x;
        st4_nt(gp, acck);  // (non-temporal: read next by the node-major pass and the weight gradient -- 6.29 -> 6.27 ms per step, three pairs)
        st4_nt(gp + X, accm);
        acck = zero;
        accm = zero;
      }
    }
    prev_key = key[U - 1];
  }
}

// Backward, source side, LONG segments: wave per work item (<= HET_ITEM_MAX edges of one source row).
template <int LPR, int DL>
__global__ __launch_bounds__(kBlock) void HET_hgt_backward_src_long(Items it, const int32_t* __restrict__ long_items,
                                                                     int64_t num_long_items, const int32_t* __restrict__ p_dst,
                                                                     const float* __restrict__ kv, const float* __restrict__ q,
                                                                     const float* __restrict__ pack2,
                                                                     const float* __restrict__ gradout,
                                                                     float* __restrict__ grad_kv, int H) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4;
  static_assert(DL <= LPR, "heads inside the lane group");  // (LPR >= 4: ids shared by quads; LPR == 2: every lane loads its ids)
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, qd = sub & 3;
  const int64_t wid = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (wid >= num_long_items) return;
  const int item = long_items[wid];
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  int dstn[U];
  if constexpr (LPR >= 4) {
    dstn[0] = p_dst[b + slot + qd * EPW < e ? b + slot + qd * EPW : e - 1];
  } else {
#pragma unroll
    for (int t = 0; t < U; ++t) dstn[t] = p_dst[b + slot + t * EPW < e ? b + slot + t * EPW : e - 1];
  }
  const int64_t u = it.seg_key[seg];
  const bool whole = b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1];
  const float4 kcur = ld4(kv + u * (2 * X) + x), mcur = ld4(kv + u * (2 * X) + X + x);
  float4 acck = make_float4(0.f, 0.f, 0.f, 0.f), accm = acck;
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    float4 qr[U], gr[U];
    float2 p2[U];
#pragma unroll
    for (int t = 0; t < U; ++t) {
      int64_t dv;
      if constexpr (LPR >= 4) dv = quad_bcast(dstn[0], t); else dv = dstn[t];
      qr[t] = ld4(q + dv * X + x);
      gr[t] = ld4(gradout + dv * X + x);
      p2[t] = *reinterpret_cast<const float2*>(pack2 + (dv * H + h) * 2);
    }
    if constexpr (LPR >= 4) {
      dstn[0] = p_dst[j0 + (U + qd) * EPW < e ? j0 + (U + qd) * EPW : e - 1];
    } else {
#pragma unroll
      for (int t = 0; t < U; ++t) dstn[t] = p_dst[j0 + (U + t) * EPW < e ? j0 + (U + t) * EPW : e - 1];
    }
#pragma unroll
    for (int t = 0; t < U; ++t) {
      const float s = head_sum<DL>(dot4(kcur, qr[t]));
      const float a = j0 + t * EPW < e ? __expf(s - p2[t].x) : 0.f;  // p2.x: lse of the destination
      const float ga = head_sum<DL>(dot4(gr[t], mcur));
      fma4(acck, a * (ga - p2[t].y), qr[t]);
      fma4(accm, a, gr[t]);
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acck.x += __shfl_xor(acck.x, off); acck.y += __shfl_xor(acck.y, off);
    acck.z += __shfl_xor(acck.z, off); acck.w += __shfl_xor(acck.w, off);
    accm.x += __shfl_xor(accm.x, off); accm.y += __shfl_xor(accm.y, off);
    accm.z += __shfl_xor(accm.z, off); accm.w += __shfl_xor(accm.w, off);
  }
  if (slot != 0) return;
  float* gp = grad_kv + u * (2 * X) + x;
  if (whole) {
    st4_nt(gp, acck);
    st4_nt(gp + X, accm);
  } else {  // (rows cleared by HET_hgt_zero_rows)
    atomic_add4(gp, acck);
    atomic_add4(gp + X, accm);
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = ceil_div64(total, kBlock);
  const int64_t cap = 256 * 64;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

// (lanes per row, lanes per head): rows of 16 .. 128 floats, heads of 8 .. 128
#define HET_DISPATCH_HGT_ROWS(LPRV, DLV, CALL)                                                                       \
  switch ((LPRV) * 64 + (DLV)) {                                                                                     \
    case 2 * 64 + 2: { constexpr int LPR = 2, DL = 2; CALL; break; }                                                 \
    case 4 * 64 + 2: { constexpr int LPR = 4, DL = 2; CALL; break; }                                                 \
    case 4 * 64 + 4: { constexpr int LPR = 4, DL = 4; CALL; break; }                                                 \
    case 8 * 64 + 2: { constexpr int LPR = 8, DL = 2; CALL; break; }                                                 \
    case 8 * 64 + 4: { constexpr int LPR = 8, DL = 4; CALL; break; }                                                 \
    case 8 * 64 + 8: { constexpr int LPR = 8, DL = 8; CALL; break; }                                                 \
    case 16 * 64 + 2: { constexpr int LPR = 16, DL = 2; CALL; break; }                                               \
    case 16 * 64 + 4: { constexpr int LPR = 16, DL = 4; CALL; break; }                                               \
    case 16 * 64 + 8: { constexpr int LPR = 16, DL = 8; CALL; break; }                                               \
    case 16 * 64 + 16: { constexpr int LPR = 16, DL = 16; CALL; break; }                                             \
    case 32 * 64 + 2: { constexpr int LPR = 32, DL = 2; CALL; break; }                                               \
    case 32 * 64 + 4: { constexpr int LPR = 32, DL = 4; CALL; break; }                                               \
    case 32 * 64 + 8: { constexpr int LPR = 32, DL = 8; CALL; break; }                                               \
    case 32 * 64 + 16: { constexpr int LPR = 32, DL = 16; CALL; break; }                                             \
    default: { constexpr int LPR = 32, DL = 32; CALL; break; }                                                       \
  }

static bool hgt_rows_shape_ok(int64_t H, int64_t D) {
  const int64_t X = H * D, lpr = X / 4, dl = D / 4;
  const bool p2 = D > 0 && (D & (D - 1)) == 0 && X > 0 && (X & (X - 1)) == 0;
  return p2 && (lpr == 2 || lpr == 4 || lpr == 8 || lpr == 16 || lpr == 32) && dl >= 2 && dl <= lpr;
}

extern "C" int het_hgt_compact_shape_ok(int64_t H, int64_t D) { return hgt_rows_shape_ok(H, D) ? 1 : 0; }

// bytes of het_hgt_aggregate_compact's workspace: one {acc[H*D], max[H], sum[H]} record per work item when destinations are split
// over several work items (more than HET_ITEM_MAX in-edges); only the records of those items are touched
extern "C" int64_t het_hgt_aggregate_compact_workspace(const het_grouping* by_dst, int64_t H, int64_t D) {
  if (!by_dst || by_dst->num_split == 0) return 0;
  return (int64_t)sizeof(float) * by_dst->num_items * (H * D + 2 * H);
}

extern "C" int het_hgt_aggregate_compact(const het_grouping* by_dst, const float* kv_c, const float* q, float* lsum, float* out,
                                         int64_t num_nodes, int64_t num_src_rows, int64_t H, int64_t D, void* workspace,
                                         int64_t workspace_bytes, het_stream stream) {
  const char* op = "het_hgt_aggregate_compact";
  hipStream_t s = (hipStream_t)stream;
  HET_REQUIRE(by_dst && lsum && out && num_nodes >= 0, "%s: null argument", op);
  if (!hgt_rows_shape_ok(H, D)) { het_set_error("%s: unsupported shape H=%lld D=%lld", op, (long long)H, (long long)D); return HET_ERR_UNSUPPORTED; }
  HET_REQUIRE(by_dst->R == 0 && by_dst->key_bound <= num_nodes && num_src_rows >= 0 && (by_dst->E == 0 || (by_dst->p0 && kv_c && q)),
              "%s: by_dst must group the positions by destination with payload0 = the (relation, source) row", op);
  const int64_t need = het_hgt_aggregate_compact_workspace(by_dst, H, D);
  HET_REQUIRE(need == 0 || (workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0),
              "%s: a 16-byte aligned workspace of %lld bytes is needed (het_hgt_aggregate_compact_workspace)", op, (long long)need);
  const int64_t X = H * D;
  HET_HIP(hipMemsetAsync(lsum, 0, sizeof(float) * num_nodes * H, s));
  if (by_dst->S != num_nodes) HET_HIP(hipMemsetAsync(out, 0, sizeof(float) * num_nodes * X, s));  // destinations without in-edges: zero rows
  if (by_dst->E == 0) return HET_OK;
  Items it{by_dst->item_seg, by_dst->item_begin, by_dst->item_end, by_dst->seg_ptr, by_dst->seg_key, by_dst->num_items};
  const unsigned nb = (unsigned)ceil_div64(by_dst->num_items, kBlock / 64);
  float* part = static_cast<float*>(workspace);
  {
    HET_KTIME("HET_hgt_aggregate_rows", s);
    HET_DISPATCH_HGT_ROWS((int)(X / 4), (int)(D / 4),
                          hipLaunchKernelGGL((HET_hgt_aggregate_rows<LPR, DL>), dim3(nb), dim3(kBlock), 0, s, it, by_dst->p0, kv_c,
                                             q, lsum, out, (int)H, part));
  }
  HET_LAUNCH_CHECK("HET_hgt_aggregate_rows");
  if (by_dst->num_split > 0) {
    const unsigned nbs = (unsigned)ceil_div64(by_dst->num_split, kBlock / 64);
    switch ((int)(X / 4)) {
      case 2: hipLaunchKernelGGL(HET_hgt_finish_split<2>, dim3(nbs), dim3(kBlock), 0, s, by_dst->split_seg, by_dst->num_split, it, part, lsum, out, (int)H, (int)D); break;
      case 4: hipLaunchKernelGGL(HET_hgt_finish_split<4>, dim3(nbs), dim3(kBlock), 0, s, by_dst->split_seg, by_dst->num_split, it, part, lsum, out, (int)H, (int)D); break;
      case 8: hipLaunchKernelGGL(HET_hgt_finish_split<8>, dim3(nbs), dim3(kBlock), 0, s, by_dst->split_seg, by_dst->num_split, it, part, lsum, out, (int)H, (int)D); break;
      case 16: hipLaunchKernelGGL(HET_hgt_finish_split<16>, dim3(nbs), dim3(kBlock), 0, s, by_dst->split_seg, by_dst->num_split, it, part, lsum, out, (int)H, (int)D); break;
      default: hipLaunchKernelGGL(HET_hgt_finish_split<32>, dim3(nbs), dim3(kBlock), 0, s, by_dst->split_seg, by_dst->num_split, it, part, lsum, out, (int)H, (int)D); break;
    }
    HET_LAUNCH_CHECK("HET_hgt_finish_split");
  }
  return HET_OK;
}

extern "C" int64_t het_hgt_backward_compact_workspace(int64_t num_nodes, int64_t H) {
  return (int64_t)sizeof(float) * ((num_nodes * 2 * H + 3) / 4 * 4);
}

extern "C" int het_hgt_backward_compact(const het_grouping* by_dst, const het_grouping* by_srow, const float* kv_c, const float* q,
                                        const float* lsum, const float* out, const float* gradout, float* grad_kv_c, float* grad_q,
                                        int64_t num_nodes, int64_t num_src_rows, int64_t H, int64_t D, void* workspace,
                                        int64_t workspace_bytes, het_stream stream) {
  const char* op = "het_hgt_backward_compact";
  hipStream_t s = (hipStream_t)stream;
  HET_REQUIRE(by_dst && by_srow && lsum && out && gradout && grad_kv_c && grad_q, "%s: null argument", op);
  if (!hgt_rows_shape_ok(H, D)) { het_set_error("%s: unsupported shape H=%lld D=%lld", op, (long long)H, (long long)D); return HET_ERR_UNSUPPORTED; }
  const int64_t E = by_dst->E, X = H * D;
  HET_REQUIRE(by_dst->R == 0 && by_srow->R == 0 && by_srow->E == E && by_dst->key_bound <= num_nodes &&
                  by_srow->key_bound <= num_src_rows && (E == 0 || (by_dst->p0 && by_srow->p0 && kv_c && q)),
              "%s: by_dst groups the positions by destination (payload0 = (relation, source) row), by_srow by that row "
              "(payload0 = destination)", op);
  const int64_t need = het_hgt_backward_compact_workspace(num_nodes, H);
  HET_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "%s: a 16-byte aligned workspace of %lld bytes is needed (het_hgt_backward_compact_workspace)", op, (long long)need);
  float* pack2 = (float*)workspace;  // [N, H, 2]
  if (by_dst->S != num_nodes) {
    HET_HIP(hipMemsetAsync(grad_q, 0, sizeof(float) * num_nodes * X, s));
  } else if (by_dst->num_split > 0) {
    hipLaunchKernelGGL(HET_hgt_zero_rows, dim3(grid_for(by_dst->num_split * (X / 4))), dim3(kBlock), 0, s, by_dst->split_seg,
                       by_dst->seg_key, by_dst->num_split, grad_q, (int)X);
    HET_LAUNCH_CHECK("HET_hgt_zero_rows");
  }
  if (by_srow->S != num_src_rows) {
    HET_HIP(hipMemsetAsync(grad_kv_c, 0, sizeof(float) * num_src_rows * 2 * X, s));
  } else if (by_srow->num_split > 0) {
    hipLaunchKernelGGL(HET_hgt_zero_rows, dim3(grid_for(by_srow->num_split * (X / 2))), dim3(kBlock), 0, s, by_srow->split_seg,
                       by_srow->seg_key, by_srow->num_split, grad_kv_c, (int)(2 * X));
    HET_LAUNCH_CHECK("HET_hgt_zero_rows");
  }
  if (E == 0) return HET_OK;
  {
    Items it{by_dst->item_seg, by_dst->item_begin, by_dst->item_end, by_dst->seg_ptr, by_dst->seg_key, by_dst->num_items};
    const unsigned nb = (unsigned)ceil_div64(by_dst->num_items, kBlock / 64);
    HET_KTIME("HET_hgt_backward_dst_rows", s);
    HET_DISPATCH_HGT_ROWS((int)(X / 4), (int)(D / 4),
                          hipLaunchKernelGGL((HET_hgt_backward_dst_rows<LPR, DL>), dim3(nb), dim3(kBlock), 0, s, it, by_dst->p0,
                                             kv_c, q, lsum, out, gradout, grad_q, pack2, (int)H));
  }
  HET_LAUNCH_CHECK("HET_hgt_backward_dst_rows");
  if (int rc = grouping_packed_ids(by_srow, true, s)) return rc;  // (builds the packs as well)
  {
    Packs pk{by_srow->pack_ptr, by_srow->key_of_rank, by_srow->num_packs};
    const unsigned nb = (unsigned)ceil_div64(by_srow->num_packs, (int64_t)(kBlock / 64) * (64 / (X / 4)));
    HET_KTIME("HET_hgt_backward_src_short", s);
    HET_DISPATCH_HGT_ROWS((int)(X / 4), (int)(D / 4),
                          hipLaunchKernelGGL((HET_hgt_backward_src_short<LPR, DL>), dim3(nb), dim3(kBlock), 0, s, pk, by_srow->kp01,
                                             kv_c, q, pack2, gradout, grad_kv_c, (int)H));
  }
  HET_LAUNCH_CHECK("HET_hgt_backward_src_short");
  if (by_srow->num_long_items > 0) {
    Items it{by_srow->item_seg, by_srow->item_begin, by_srow->item_end, by_srow->seg_ptr, by_srow->seg_key, by_srow->num_items};
    const unsigned nbl = (unsigned)ceil_div64(by_srow->num_long_items, kBlock / 64);
    HET_KTIME("HET_hgt_backward_src_long", s);
    HET_DISPATCH_HGT_ROWS((int)(X / 4), (int)(D / 4),
                          hipLaunchKernelGGL((HET_hgt_backward_src_long<LPR, DL>), dim3(nbl), dim3(kBlock), 0, s, it,
                                             by_srow->long_items, by_srow->num_long_items, by_srow->p0, kv_c, q, pack2, gradout,
                                             grad_kv_c, (int)H));
    HET_LAUNCH_CHECK("HET_hgt_backward_src_long");
  }
  return HET_OK;
}
