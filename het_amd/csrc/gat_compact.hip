// RGAT on the distinct (relation, node) rows, without any per-edge float tensor between forward and backward.
//
// The layer's projections live on the S_row distinct (relation, source) rows (feat_c [S_row,H,D], el_c [S_row,H]) and
// the S_col distinct (relation, destination) rows (er_c [S_col,H]).  The reference's ops pass exp [E,H] from the forward
// to the backward (RGAT/RGATKernelsSeparateCOO.cu.h:190-196 writes it, RGATBackwardKernelsSeparateCOO.cu.h:60-80 reads
// it); here both passes form exp(leaky_relu(el_c[srow] + er_c[drow])) from the two small tables (38 MB + 20 MB on
// ogbn-mag: they live in L2 / Infinity Cache), so the [E,H] tensor, its pass (HET_gat_exp_edge) and its 16-byte-per-128-
// byte-line gathers are gone.  Same values: exp is a pure function of el + er.
//
//   forward   wave per destination work item (het_grouping by destination, payload0 = feat row, payload1 = er row); round 3's
//             default form also leaves per-RUN sums for grad_er (one run = the edges of one er row): lane group per pack of
//             whole destinations walking the edges in order, hubs through the work items of a (destination, relation)
//             grouping + a finish pass -- "grad_er without a per-edge term" below
//   backward  SHORT (relation, source) segments (<= HET_PACK_T edges; the median is 2): lane group per PACK of whole
//             segments (grouping_packs) -- a work unit per segment would spend its time in dependent prologues (item
//             record -> ids -> rows); a pack of ~32 consecutive ranks streams its ids a step ahead and keeps 4 gradient
//             rows in flight per lane group whatever the segment lengths are; rows of a segment are summed in registers
//             and stored once.  LONG segments (61 % of the edges of the skewed ogbn-mag-like graph): wave per work item
//             of <= HET_ITEM_MAX edges, lane groups round-robin, one cross-group reduction and store per item.
//             grad_er: from the forward's run sums (HET_rgat_grad_er_runs), or -- het_rgat_backward_compact -- a per-edge
//             term summed per er row.
#include <stdlib.h>

#include "coop.hip.h"
#include "fused_gat.hip.h"
#include "seg_reduce.hip.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

struct Items {
  const int32_t *seg, *begin, *end, *seg_ptr, *seg_key;
  int64_t n;
};

// Softmax WITHOUT overflow inside the one-node layer.  The reference's kernels exponentiate the raw pre-activation
// (gatLeakyReluExp, GAT/FusedGAT.cu.h:23-26: exp(leaky_relu(el + er)), no maximum subtracted -- SURVEY.md Q2), which the
// reference-named ops keep because exp / sum are API tensors there.  Here neither is visible to the caller, so every pass
// works with s = leaky_relu(el + er) relative to a maximum: the forward keeps a running maximum per (destination, head) and
// rescales its partial sums when it grows (online softmax), stores lse[v,h] = max + log(SUM exp(s - max)) where the
// reference stores the sum, and the backward forms the attention weight as exp(s - lse[v,h]).  Same value as
// exp(s) / SUM exp(s) wherever the reference's formula is finite; finite everywhere else too (|el + er| of 100 and more).
__device__ __forceinline__ float lrelu(float z, float slope) { return z > 0.f ? z : slope * z; }
// The stored log-sum-exp of a destination WITH in-edges is never exactly 0 (an exact 0 becomes the smallest normal float: a
// relative change of 1e-38 in the weights), so that "lse == 0" means "no in-edges" -- the zero fill of the forward -- and the
// backward's per-destination pass can leave the rows of those nodes (60 % of ogbn-mag's) unread.
__device__ __forceinline__ float lse_of(float m, float ssum) {
  const float L = m + __logf(ssum);
  return L == 0.f ? 1.17549435e-38f : L;
}

// ret[v,h,:] = SUM_e w_e * feat[srow_e,h,:] / SUM_e w_e,  w_e = exp(leaky(el[srow_e,h] + er[drow_e,h])); sum[v,h] = SUM_e w_e
// Same schedule as HET_gat_aggregate_grouped (fused_gat_grouped.hip): 64/LPR lane groups take the item's edges
// round-robin, U rows per group in flight, ids of the next step prefetched.
// A whole segment is finished here (running maximum, one store).  An item of a split (hub) destination parks its partial
// result {acc[X], max[H], sum[H]} in part[item]; HET_rgat_finish_split brings the items of a destination to their common maximum.
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rgat_aggregate_compact(Items it, const int32_t* __restrict__ p_srow,
                                                                      const int32_t* __restrict__ p_drow,
                                                                      const float* __restrict__ feat,
                                                                      const float* __restrict__ el,
                                                                      const float* __restrict__ er,
                                                                      float* __restrict__ lse, float* __restrict__ ret,
                                                                      int H, int D, float slope, float* __restrict__ hio,
                                                                      int64_t hio_rows, float* __restrict__ part) {
  constexpr int EPW = 64 / LPR, U = 4;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, x = (lane % LPR) * 4, h = x / D;
  const int64_t item = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= it.n) return;
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  const bool whole = b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1];
  const int64_t X = (int64_t)H * D;
  const int64_t v = it.seg_key[seg];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float ssum = 0.f, m = -INFINITY;
  int jn[U];
  int64_t srown[U], drown[U];
#pragma unroll
  for (int u = 0; u < U; ++u) jn[u] = b + slot + u * EPW < e ? b + slot + u * EPW : e - 1;
#pragma unroll
  for (int u = 0; u < U; ++u) srown[u] = p_srow[jn[u]];
#pragma unroll
  for (int u = 0; u < U; ++u) drown[u] = p_drow[jn[u]];
  // hio (optional): the layer output so far (self-loop + bias rows); the aggregated row is added to it in place.  Its row
  // is requested here so that the read-modify-write at the end of the item does not wait for it
  const bool add_h = hio && whole && slot == 0 && v < hio_rows;
  float4 h0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (add_h) h0 = ld4(hio + v * X + x);
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    float zl[U], zr[U], sv[U];
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) zl[u] = el[srown[u] * H + h];
#pragma unroll
    for (int u = 0; u < U; ++u) zr[u] = er[drown[u] * H + h];
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = ld4(feat + srown[u] * X + x);
#pragma unroll
    for (int u = 0; u < U; ++u) jn[u] = j0 + (U + u) * EPW < e ? j0 + (U + u) * EPW : e - 1;
#pragma unroll
    for (int u = 0; u < U; ++u) srown[u] = p_srow[jn[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) drown[u] = p_drow[jn[u]];
    float mn = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sv[u] = j0 + u * EPW < e ? lrelu(zl[u] + zr[u], slope) : -INFINITY;
      mn = fmaxf(mn, sv[u]);
    }
    {  // (the first edge of a step exists: mn is finite)
      const float c = __expf(m - mn);
      acc.x *= c; acc.y *= c; acc.z *= c; acc.w *= c; ssum *= c;
      m = mn;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = __expf(sv[u] - m);  // exp(-inf) = 0 for the padding edges
      acc.x = fmaf(w, f[u].x, acc.x);
      acc.y = fmaf(w, f[u].y, acc.y);
      acc.z = fmaf(w, f[u].z, acc.z);
      acc.w = fmaf(w, f[u].w, acc.w);
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
    acc.x += __shfl_xor(acc.x, off);
    acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off);
    acc.w += __shfl_xor(acc.w, off);
    ssum += __shfl_xor(ssum, off);
  }
  if (slot != 0) return;
  if (whole) {
    const float inv = 1.f / ssum;
    const float4 r4 = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    st4(ret + v * X + x, r4);
    if (add_h) st4(hio + v * X + x, make_float4(h0.x + r4.x, h0.y + r4.y, h0.z + r4.z, h0.w + r4.w));
    if (x % D == 0) lse[v * H + h] = lse_of(m, ssum);
  } else {  // a piece of a hub destination: parked for HET_rgat_finish_split
    float* pp = part + item * (X + 2 * H);
    st4(pp + x, acc);
    if (x % D == 0) { pp[X + h] = m; pp[X + H + h] = ssum; }
  }
}

// One wave per split (hub) destination: its work items parked {acc[X], max[H], sum[H]} each (part[item]); the lane groups of
// the wave take them round-robin (the largest hub of ogbn-mag has ~400), are brought to the common maximum, divided, and the
// row is stored (ret, lse) and added to the layer output.
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rgat_finish_split(const int32_t* __restrict__ split_seg, int64_t num_split, Items it,
                                                                 const float* __restrict__ part, float* __restrict__ lse,
                                                                 float* __restrict__ ret, int H, int D, float* __restrict__ hio,
                                                                 int64_t hio_rows) {
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
  const float4 r4 = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  st4(ret + v * X + x, r4);
  if (hio && v < hio_rows) {
    const float4 h0 = ld4(hio + v * X + x);
    st4(hio + v * X + x, make_float4(h0.x + r4.x, h0.y + r4.y, h0.z + r4.z, h0.w + r4.w));
  }
  if (x % D == 0) lse[v * H + h] = lse_of(M, ssum);
}

// pack[v] = { lse[v,h] (H floats), <gradout[v,h,:], ret[v,h,:]> (H floats) }, or interleaved per head ([N,H,2]); bias_part (optional, [gridDim.x * waves, X]):
// per-wave column sums of gradout over the nodes the wave visited (the bias gradient, reduced by HET_rgat_colsum_finish).
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rgat_dst_pack(const float* __restrict__ sum, const float* __restrict__ ret,
                                                             const float* __restrict__ gradout, float* __restrict__ pack,
                                                             int64_t N, int H, int D, float* __restrict__ bias_part,
                                                             int64_t bias_rows, int interleaved) {
  constexpr int EPW = 64 / LPR, X = LPR * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const int64_t step = (int64_t)gridDim.x * (kBlock / 64) * EPW;
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t v0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * EPW; v0 < N; v0 += step) {
    const bool ok = v0 + slot < N;
    const int64_t v = ok ? v0 + slot : N - 1;  // past the end: the last node again (same bytes rewritten)
    // a node without in-edges (lse == 0: lse_of) has no row of ret worth reading and no edge reads its pack record: its load goes
    // to row 0 (cached) and its dot is 0
    const float ls = sum[v * H + h];
    const bool has = ls != 0.f;
    const float4 g = ld4(gradout + v * X + x), r = ld4(ret + (has ? v : 0) * X + x);
    float dot = has ? g.x * r.x + g.y * r.y + g.z * r.z + g.w * r.w : 0.f;
    for (int off = DL >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    if ((sub & (DL - 1)) == 0) {
      if (interleaved) {
        *reinterpret_cast<float2*>(pack + (v * H + h) * 2) = make_float2(ls, dot);
      } else {
        pack[v * 2 * H + h] = ls;
        pack[v * 2 * H + H + h] = dot;
      }
    }
    if (bias_part && ok && v < bias_rows) { bs.x += g.x; bs.y += g.y; bs.z += g.z; bs.w += g.w; }
  }
  if (bias_part) {
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      bs.x += __shfl_xor(bs.x, off); bs.y += __shfl_xor(bs.y, off);
      bs.z += __shfl_xor(bs.z, off); bs.w += __shfl_xor(bs.w, off);
    }
    if (slot == 0) st4(bias_part + ((int64_t)blockIdx.x * (kBlock / 64) + wave) * X + x, bs);
  }
}

// out[x] = SUM_p part[p, x]: one workgroup per 64 columns... X <= 256 columns, P partial rows: thread per (column, slice)
__global__ __launch_bounds__(kBlock) void HET_rgat_colsum_finish(const float* __restrict__ part, int64_t P, int X,
                                                                  float* __restrict__ out) {
  __shared__ float red[kBlock];
  const int x = blockIdx.x, t = threadIdx.x;
  float a = 0.f;
  for (int64_t p = t; p < P; p += kBlock) a += part[p * X + x];
  red[t] = a;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) out[x] = red[0];
}

struct Packs {
  const int32_t *ptr, *key;
  int64_t n;
};

// Lane group per pack of the grouping by feat row u (payload0 = destination, payload1 = er row):
//   a_e = exp(leaky(el[u,h] + er[drow_e,h]) - lse[dst_e,h]);  dl_e = (el + er > 0) ? 1 : slope
//   grad_feat[u,h,:] = SUM_e a_e * gradout[dst_e,h,:]  (+ grad_el[u,h] * fold_w[r(u),h,:])
//   t_e = a_e * dl_e * (<gradout[dst_e,h,:], feat[u,h,:]> - <gradout, ret>[dst_e,h]);   grad_el[u,h] = SUM_e t_e
//   tbuf[j,h] = t_e for the edge at sorted rank j   (summed per er row by a segmented pass over the grouping by er row)
template <int LPR, int U>
__global__ __launch_bounds__(kBlock) void HET_rgat_backward_src_packed(
    Packs pk, const int32_t* __restrict__ p_dst, const int32_t* __restrict__ p_drow, const float* __restrict__ feat,
    const float* __restrict__ el, const float* __restrict__ er, const float* __restrict__ pack,
    const float* __restrict__ gradout, float* __restrict__ grad_feat, float* __restrict__ grad_el,
    float* __restrict__ tbuf, int H, int D, float slope, const float* __restrict__ fold_w,
    const idx_t* __restrict__ fold_row_rel_ptrs, int R, int skip_long) {
  constexpr int EPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const int64_t pid = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * EPW + slot;
  if (pid >= pk.n) return;
  const uint32_t pb = (uint32_t)pk.ptr[pid];
  const int b = (int)(pb & 0x7fffffffu), e = (int)((uint32_t)pk.ptr[pid + 1] & 0x7fffffffu);
  if (skip_long && (pb >> 31)) return;  // a long segment: HET_rgat_backward_src_long_any takes its work items
  constexpr bool partial = false;  // (without skip_long a long segment is one pack: summed by this lane group alone)
  const int64_t X = (int64_t)H * D;
  const bool head_lane = (sub & (DL - 1)) == 0;
  // ids of the first batch (clamped to the pack); keyn[U] = key of the rank after the batch
  int keyn[U + 1];
  int64_t dstn[U], drown[U];
#pragma unroll
  for (int q = 0; q <= U; ++q) keyn[q] = pk.key[b + q < e ? b + q : e];  // key[e] belongs to the next pack (or -1)
#pragma unroll
  for (int q = 0; q < U; ++q) dstn[q] = p_dst[b + q < e ? b + q : e - 1];
#pragma unroll
  for (int q = 0; q < U; ++q) drown[q] = p_drow[b + q < e ? b + q : e - 1];
  int prev_key = -1;  // no segment open
  float4 fcur = make_float4(0.f, 0.f, 0.f, 0.f), acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float zlcur = 0.f, acc_el = 0.f;
  for (int j0 = b; j0 < e; j0 += U) {
    int key[U + 1];
    int64_t dst[U];
    float zr[U], sinv[U], gr[U], zlq[U];
    float4 g[U], fq[U];
#pragma unroll
    for (int q = 0; q <= U; ++q) key[q] = keyn[q];
#pragma unroll
    for (int q = 0; q < U; ++q) dst[q] = dstn[q];
#pragma unroll
    for (int q = 0; q < U; ++q) zr[q] = er[drown[q] * H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) sinv[q] = pack[dst[q] * 2 * H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) gr[q] = pack[dst[q] * 2 * H + H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) g[q] = ld4(gradout + dst[q] * X + x);
    // the feat row / el of a segment are loaded where the segment starts inside this batch (predicated: uniform per lane group)
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const bool start = j0 + q < e && key[q] != (q == 0 ? prev_key : key[q - 1]);
      fq[q] = fcur;
      zlq[q] = zlcur;
      if (start) {
        fq[q] = ld4(feat + (int64_t)key[q] * X + x);
        zlq[q] = el[(int64_t)key[q] * H + h];
      }
    }
    // ids of the next batch, in flight while this batch's rows arrive
#pragma unroll
    for (int q = 0; q <= U; ++q) keyn[q] = pk.key[j0 + U + q < e ? j0 + U + q : e];
#pragma unroll
    for (int q = 0; q < U; ++q) dstn[q] = p_dst[j0 + U + q < e ? j0 + U + q : e - 1];
#pragma unroll
    for (int q = 0; q < U; ++q) drown[q] = p_drow[j0 + U + q < e ? j0 + U + q : e - 1];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const bool ok = j0 + q < e;  // uniform within the lane group
      const bool start = ok && key[q] != (q == 0 ? prev_key : key[q - 1]);
      if (start) { fcur = fq[q]; zlcur = zlq[q]; }
      const float z = zlcur + zr[q];
      const float a = ok ? __expf(lrelu(z, slope) - sinv[q]) : 0.f;  // (sinv: lse of the destination)
      acc.x = fmaf(a, g[q].x, acc.x); acc.y = fmaf(a, g[q].y, acc.y);
      acc.z = fmaf(a, g[q].z, acc.z); acc.w = fmaf(a, g[q].w, acc.w);
      float dot = g[q].x * fcur.x + g[q].y * fcur.y + g[q].z * fcur.z + g[q].w * fcur.w;
      for (int off = DL >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
      const float t = a * (z > 0.f ? 1.f : slope) * (dot - gr[q]);
      if (ok && head_lane) tbuf[(int64_t)(j0 + q) * H + h] = t;
      acc_el += t;  // identical in the DL lanes of a head
      const bool last = ok && (j0 + q == e - 1 || key[q + 1] != key[q]);
      if (last) {  // the segment (or this pack's piece of it) ends here: one store of its gradient row
        const int64_t u = key[q];
        float4 o = acc;
        if (fold_w) {  // el[u,h] = <feat[u,h,:], fold_w[r(u),h,:]>: its gradient joins grad_feat here (linear: also per piece)
          const float4 w = ld4(fold_w + (int64_t)find_segment(fold_row_rel_ptrs, R, (idx_t)u) * X + x);
          o.x = fmaf(acc_el, w.x, o.x); o.y = fmaf(acc_el, w.y, o.y);
          o.z = fmaf(acc_el, w.z, o.z); o.w = fmaf(acc_el, w.w, o.w);
        }
        float* gp = grad_feat + u * X + x;
        if (!partial) {
          st4(gp, o);
          if (head_lane) grad_el[u * H + h] = acc_el;
        } else {
          atomicAdd(gp + 0, o.x); atomicAdd(gp + 1, o.y); atomicAdd(gp + 2, o.z); atomicAdd(gp + 3, o.w);
          if (head_lane) atomicAdd(&grad_el[u * H + h], acc_el);
        }
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        acc_el = 0.f;
      }
    }
    prev_key = key[U - 1];
  }
}


// The LONG (relation, source) segments for the shapes the cooperative kernels are not built for (heads of 4 or 8 floats, rows
// of other widths): wave per work item, lane groups round-robin, every lane fetches its own scalars.  Without it a row with
// 12 000 edges was summed by ONE lane group of the pack kernel above (10.6 ms on ogbn-mag with 8 heads of 8).
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rgat_backward_src_long_any(
    Items it, const int32_t* __restrict__ long_items, int64_t num_long_items, const int32_t* __restrict__ p_dst,
    const int32_t* __restrict__ p_drow, const float* __restrict__ feat, const float* __restrict__ el,
    const float* __restrict__ er, const float* __restrict__ pack, const float* __restrict__ gradout,
    float* __restrict__ grad_feat, float* __restrict__ grad_el, float* __restrict__ tbuf, int H, int D, float slope,
    const float* __restrict__ fold_w, const idx_t* __restrict__ fold_row_rel_ptrs, int R) {
  constexpr int EPW = 64 / LPR, U = 4;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const bool head_lane = (sub & (DL - 1)) == 0;
  const int64_t wid = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (wid >= num_long_items) return;
  const int item = long_items[wid];
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  const int64_t u = it.seg_key[seg], X = (int64_t)H * D;
  const bool whole = b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1];
  const float4 f = ld4(feat + u * X + x);
  const float zl = el[u * H + h];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float acc_el = 0.f;
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    int64_t dst[U], drow[U];
    bool ok[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int j = j0 + q * EPW;
      ok[q] = j < e;
      dst[q] = p_dst[ok[q] ? j : e - 1];
      drow[q] = p_drow[ok[q] ? j : e - 1];
    }
    float zr[U], sinv[U], gr[U];
    float4 g[U];
#pragma unroll
    for (int q = 0; q < U; ++q) zr[q] = er[drow[q] * H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) sinv[q] = pack[dst[q] * 2 * H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) gr[q] = pack[dst[q] * 2 * H + H + h];
#pragma unroll
    for (int q = 0; q < U; ++q) g[q] = ld4(gradout + dst[q] * X + x);
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const float z = zl + zr[q];
      const float a = ok[q] ? __expf(lrelu(z, slope) - sinv[q]) : 0.f;  // (sinv: lse of the destination)
      acc.x = fmaf(a, g[q].x, acc.x); acc.y = fmaf(a, g[q].y, acc.y);
      acc.z = fmaf(a, g[q].z, acc.z); acc.w = fmaf(a, g[q].w, acc.w);
      float dot = g[q].x * f.x + g[q].y * f.y + g[q].z * f.z + g[q].w * f.w;
      for (int off = DL >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
      const float t = a * (z > 0.f ? 1.f : slope) * (dot - gr[q]);
      if (ok[q] && head_lane) tbuf[(int64_t)(j0 + q * EPW) * H + h] = t;
      acc_el += t;  // identical in the DL lanes of a head
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    acc_el += __shfl_xor(acc_el, off);
  }
  if (slot != 0) return;
  if (fold_w) {
    const float4 w = ld4(fold_w + (int64_t)find_segment(fold_row_rel_ptrs, R, (idx_t)u) * X + x);
    acc.x = fmaf(acc_el, w.x, acc.x); acc.y = fmaf(acc_el, w.y, acc.y);
    acc.z = fmaf(acc_el, w.z, acc.z); acc.w = fmaf(acc_el, w.w, acc.w);
  }
  float* gp = grad_feat + u * X + x;
  if (whole) {
    st4(gp, acc);
    if (head_lane) grad_el[u * H + h] = acc_el;
  } else {  // (rows cleared by HET_rgat_zero_long_rows)
    atomicAdd(gp + 0, acc.x); atomicAdd(gp + 1, acc.y); atomicAdd(gp + 2, acc.z); atomicAdd(gp + 3, acc.w);
    if (head_lane) atomicAdd(&grad_el[u * H + h], acc_el);
  }
}

// ---- cooperative scalar loads -------------------------------------------------------------------------------------
// Both passes were bound by the NUMBER of vector-memory instructions, not by bytes: every per-edge scalar (an index, an
// attention term) cost a wave instruction that fetched 4 distinct values for 64 lanes, and a step of 4 edges per lane
// group issued 20 (forward) / 41 (backward) of them against 4 row loads (measured: 1.15 / 2.4 ms, i.e. the same ~60 us
// per instruction-per-step in both).  Here lane (head h, d) of a lane group fetches the scalar of edge d of the step --
// one instruction per step and stream instead of one per edge -- and the DL = D/4 lanes of a head exchange the values
// with DPP quad broadcasts (DL == 4: no LDS, no extra instruction slot) or a bpermute.

// Forward, cooperative form of HET_rgat_aggregate_compact (DL = D/4 >= 4 lanes per head, 4 edges per lane group and step).
// Measured and dropped (same box, ogbn-mag):
//  * splitting it like the backward (packs of short destination segments per lane group, long ones per wave): 0.94-0.96 ms
//    against 0.93 ms for this kernel alone -- the pass runs at the memory system's rate either way;
//  * the edge ids of a step through the scalar cache (they are wave-uniform addresses: s_load instead of two of the eight
//    vector-memory instructions per step): 0.93 -> 2.45 ms -- scalar loads return out of order, so every step waits for all
//    of them, and 32 ids per step do not stream through the scalar cache;
//  * persistent waves that walk the items with a stride and fetch the record and first ids of their next item while the
//    current one is in flight (an item is 26 edges = 2 steps on average, so a wave per item is one dependent chain item
//    record -> ids -> rows -> store): 0.92 -> 1.02-1.11 ms with 1024 .. 8192 workgroups;
//  * also accumulating P[(r,v),h,:] = SUM_e w_e dl_e feat_c[srow_e] per (relation, destination) here, so that the backward
//    gets grad_er from S_col rows instead of a per-edge term [E,H] + a segmented sum of 16-byte gathers (0.59 ms, 3 GB):
//    the per-relation accumulators take the kernel from 53 to 104 VGPRs = 8 -> 4 waves per SIMD and 0.93 -> 1.44-1.81 ms,
//    more than the 0.7 ms it saves in the backward.  (Round 3 does it with ONE accumulator by walking the edges in run
//    order: HET_rgat_aggregate_runs_packed / _hub_items below.)
template <int LPR, int DL>
__global__ __launch_bounds__(kBlock) void HET_rgat_aggregate_coop(Items it, const int2* __restrict__ p01,
                                                                   const float* __restrict__ feat,
                                                                   const float* __restrict__ el,
                                                                   const float* __restrict__ er,
                                                                   float* __restrict__ lse, float* __restrict__ ret,
                                                                   int H, float slope, float* __restrict__ hio,
                                                                   int64_t hio_rows, float* __restrict__ part) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4;
  static_assert(DL >= U, "a head needs at least U lanes");
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, d = sub % DL;
  const int dq = d < U ? d : U - 1;  // the edge of the step whose scalars this lane fetches
  const int64_t item = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= it.n) return;
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  const bool whole = b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1];
  const int64_t v = it.seg_key[seg];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float ssum = 0.f, m = -INFINITY;
  int jn = b + slot + dq * EPW < e ? b + slot + dq * EPW : e - 1;
  int2 idn = p01[jn];  // {feat row, er row} of the edge: one load (grouping_packed_ids)
  const bool add_h = hio && whole && slot == 0 && v < hio_rows;  // see HET_rgat_aggregate_compact
  float4 h0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (add_h) h0 = ld4(hio + v * X + x);
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    const int srowv = idn.x, drowv = idn.y;
    const float zlv = el[(int64_t)srowv * H + h];
    const float zrv = er[(int64_t)drowv * H + h];
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = ld4(feat + (int64_t)head_bcast_i<DL>(srowv, u, lane) * X + x);
    jn = j0 + (U + dq) * EPW < e ? j0 + (U + dq) * EPW : e - 1;
    idn = p01[jn];
    // s of this lane's edge; the running maximum of the head over the U edges of the step (lanes d >= U repeat edge U - 1)
    const float sv = j0 + dq * EPW < e ? lrelu(zlv + zrv, slope) : -INFINITY;
    {
      float mn = m;
#pragma unroll
      for (int u = 0; u < U; ++u) mn = fmaxf(mn, head_bcast<DL>(sv, u, lane));
      const float c = __expf(m - mn);  // (the first edge of a step exists: mn is finite; exp(-inf) = 0 the first time)
      acc.x *= c; acc.y *= c; acc.z *= c; acc.w *= c; ssum *= c;
      m = mn;
    }
    const float wv = __expf(sv - m);  // one exp per (edge, head); 0 for the padding edges
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
    acc.x += __shfl_xor(acc.x, off);
    acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off);
    acc.w += __shfl_xor(acc.w, off);
    ssum += __shfl_xor(ssum, off);
  }
  if (slot != 0) return;
  if (whole) {
    const float inv = 1.f / ssum;
    const float4 r4 = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    st4(ret + v * X + x, r4);
    if (add_h) st4(hio + v * X + x, make_float4(h0.x + r4.x, h0.y + r4.y, h0.z + r4.z, h0.w + r4.w));
    if (d == 0) lse[v * H + h] = lse_of(m, ssum);
  } else {  // a piece of a hub destination: parked for HET_rgat_finish_split
    float* pp = part + item * (X + 2 * H);
    st4(pp + x, acc);
    if (d == 0) { pp[X + h] = m; pp[X + H + h] = ssum; }
  }
}

// ---- grad_er without a per-edge term ------------------------------------------------------------------------------
// grad_er[(r,v),h] = SUM over the edges e of the RUN (relation r, destination v) of a_e dl_e (<gradout[v,h,:], feat[srow_e,h,:]> -
// <gradout, ret>[v,h]).  gradout[v] is the same for the whole run, so with
//     Q[(r,v),h,:] = SUM_e w_e dl_e feat[srow_e,h,:],   q[(r,v),h] = SUM_e w_e dl_e,   w_e = exp(s_e - ref[(r,v),h])
// it is exp(ref - lse[v,h]) (<gradout[v,h,:], Q> - <gradout, ret>[v,h] q): S_col rows instead of a per-edge tensor [E,H] written
// in (relation, source) order and summed in (relation, destination) order (16-byte gathers: 0.58 ms + 0.34 GB of stores on
// ogbn-mag).  The forward visits the feat rows of a run anyway; what made this lose in round 2 (note above: one accumulator
// per relation, 104 VGPRs) is avoided by walking a destination's edges IN ORDER -- the grouping by destination is a stable
// sort of relation-major positions, so the runs of a destination are contiguous -- with ONE run accumulator:
//   destinations of <= hub_min in-edges: lane group per pack of whole destinations (or per destination), edges in order,
//     running maximum per destination, run sums stored where the run ends with the maximum of that moment as `ref`;
//   hubs (54 % of the edges of ogbn-mag at 256): wave per work item of the grouping by (destination, relation) -- same sorted
//     order, items never cross a run -- partial {O[X], Q[X], max[H], sum[H], q[H]} parked; one wave per hub then finishes
//     ret / lse and every run of the hub (ref = lse).

// el recomputed from the gathered row instead of gathered itself: el[(r,u),h] = <feat_c[(r,u),h,:], attn_l[r,h,:]> is a function of the
// row the edge loads anyway, so the forward passes can form it in registers (4 FMAs + two quad shuffles per edge and lane) and drop the
// per-edge 16-byte gather of el_c -- a 128-byte line of a 38 MB table per edge, a quarter of the lines these passes request (the L2
// window experiment of DESIGN.md 4.0 left exactly those misses: one per edge).  Needs heads of 16 floats (4 lanes per head: quad DPP)
// and the relation of the row: rows are relation-major, so r = number of relation boundaries at or below the row id -- counted once
// per edge when the packed ids are tagged (grouping_tag_kp01), not per edge and launch (round 5).
constexpr int kElMaxRels = 8;
struct ElFold {
  const float* attn;  // [R, X] attn_l, or NULL: el is gathered
  int R;
  int thr[kElMaxRels - 1];  // first feat row of relations 1 .. R-1 (INT_MAX beyond)
};
__device__ __forceinline__ int el_relation(const ElFold& f, int srow) {
  int r = 0;
#pragma unroll
  for (int k = 0; k < kElMaxRels - 1; ++k) r += srow >= f.thr[k] ? 1 : 0;
  return r;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

// byte-offset type of the row kernels (common.hip.h: "byte offsets off a wave-uniform base")
template <bool W64> struct OffSel { typedef uint32_t type; };
template <> struct OffSel<true> { typedef uint64_t type; };
// lse = m + log(ssum) with ssum >= 1 (the edge that holds the maximum contributes exp(0)): v_log_f32 without the denormal path
__device__ __forceinline__ float lse_fast(float m, float ssum) {
  const float L = fmaf(__builtin_amdgcn_logf(ssum), 0.69314718056f, m);
  return L == 0.f ? 1.17549435e-38f : L;
}

// one edge joins the running sums of its destination (acc, ssum) and of its run (accq, sq); all relative to the running maximum m
__device__ __forceinline__ void online_edge(float s, float dl, const float4& f, float& m, float4& acc, float& ssum, float4& accq,
                                            float& sq) {
  const float t = __expf(-fabsf(s - m));  // (m = -inf before the first edge: t = 0, the old sums are dropped)
  const bool grow = s > m;
  const float c = grow ? t : 1.f, w = grow ? 1.f : t, wd = w * dl;
  m = grow ? s : m;
  acc.x = fmaf(acc.x, c, w * f.x); acc.y = fmaf(acc.y, c, w * f.y); acc.z = fmaf(acc.z, c, w * f.z); acc.w = fmaf(acc.w, c, w * f.w);
  accq.x = fmaf(accq.x, c, wd * f.x); accq.y = fmaf(accq.y, c, wd * f.y); accq.z = fmaf(accq.z, c, wd * f.z); accq.w = fmaf(accq.w, c, wd * f.w);
  ssum = fmaf(ssum, c, w);
  sq = fmaf(sq, c, wd);
}

// Round 5: the instruction stream of this kernel, not its bytes, set its time (VALU 61 % of the issue slots).  Per edge
// it no longer (a) counts relation boundaries (7 compares + 7 selects + the hazard nops between them) and reads attn_l[r] from LDS
// behind a full lgkmcnt wait -- the relation rides in the tag of the packed id record and the attention vector of the current RUN
// stays in registers; (b) compares the destination / er row against the previous edge's to find where a run or a destination ends --
// the tag says so, and says it of the LAST edge, so the sums are stored with the ids of the edge in hand (no carried ids, no flush
// after the loop); (c) forms 64-bit addresses -- one v_lshl_or_b32 per row off a scalar base (W64 = false: every table below 4 GiB,
// checked by the launcher); (d) divides with the IEEE sequence where a destination ends (v_rcp_f32: 1 ulp).  H = LPR / DL.
// Measured and dropped (round 5): the feature rows of step k + 1 requested before the sums of step k are formed (two row buffers
// that swap roles, ids two steps ahead) -- 90 VGPRs instead of 60 = 5 waves per SIMD instead of 8: 0.534 -> 0.590 ms alone.  More
// waves, not more loads per wave, is what this pass wants.
template <int LPR, int DL, bool ELR, bool W64>
__global__ __launch_bounds__(kBlock) void HET_rgat_aggregate_runs_packed(
    Packs pk, const int4* __restrict__ kp01, const float* __restrict__ feat, const float* __restrict__ el,
    const float* __restrict__ er, float* __restrict__ lse, float* __restrict__ ret, float slope, float* __restrict__ hio,
    int hio_rows, float* __restrict__ qrow, float* __restrict__ qsum, float* __restrict__ qref, int hub_min, ElFold ef) {
  constexpr int EPW = 64 / LPR, U = 4, H = LPR / DL;
  constexpr int RS = het_log2_ce(LPR * 16), HS = het_log2_ce(H * 4);  // log2 of the bytes of a feature row / of an [.,H] row
  typedef typename OffSel<W64>::type O;
  static_assert(DL >= U, "a head needs at least U lanes");
  static_assert(!ELR || DL == 4, "el from the row: heads of 4 lanes");
  __shared__ float4 al_s[ELR ? kElMaxRels * LPR : 1];  // attn_l[r] as the lanes of a row hold it
  if (ELR) {
    for (int i = threadIdx.x; i < ef.R * LPR; i += kBlock) al_s[i] = ld4(ef.attn + (int64_t)i * 4);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, h = sub / DL, d = sub % DL;
  const O xb = (O)(sub * 16), hb = (O)(h * 4);
  const int dq = d < U ? d : U - 1;
  const int64_t pid = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * EPW + slot;
  if (pid >= pk.n) return;
  const uint32_t pb = (uint32_t)pk.ptr[pid];
  const int b = (int)(pb & 0x7fffffffu), e = (int)((uint32_t)pk.ptr[pid + 1] & 0x7fffffffu);
  if ((pb >> 31) && e - b > hub_min) return;  // a hub (a pack of its own): HET_rgat_aggregate_hub_items + HET_rgat_finish_hubs
  int rel_cur = -1;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accq = acc, h0 = acc, a4 = acc;
  float ssum = 0.f, sq = 0.f, m = -INFINITY;
  // {destination, feat row, er row, tag} of the edge this lane fetches the ids of, for the step that starts at rank j
  auto ids_of = [&](int j) { return kp01[j + dq < e ? j + dq : e - 1]; };
  // the loads of a step: its U feature rows, the er (and el) terms of the lane's edge
  auto rows_of = [&](const int4& id, float4 (&f)[U], float& zl, float& zr) {
    zl = ELR ? 0.f : ld1_at<O>(el, ((O)id.y << HS) | hb);
    zr = ld1_at<O>(er, ((O)id.z << HS) | hb);
#pragma unroll
    for (int q = 0; q < U; ++q) f[q] = ld4_at<O>(feat, ((O)head_bcast_i<DL>(id.y, q, lane) << RS) | xb);
  };
  // the arithmetic (and the stores) of the step that starts at rank j0
  auto step = [&](int j0, const int4& id, const float4 (&f)[U], float zlv, float zrv) {
    const int dstv = id.x, drowv = id.z, tagv = id.w;
    const float zv = zlv + zrv, sv = lrelu(zv, slope), dlv = zv > 0.f ? 1.f : slope;
#pragma unroll
    for (int q = 0; q < U; ++q) {
      if (j0 + q < e) {  // (uniform within the lane group, like everything below)
        const int tagq = head_bcast_i<DL>(tagv, q, lane);
#ifndef HET_ABL_NO_HIO
        if (hio && (tagq & HET_TAG_FIRST_KEY)) {  // the destination's row of the layer output so far: needed when the destination ends
          const int dstq = head_bcast_i<DL>(dstv, q, lane);
          if (dstq < hio_rows) h0 = ld4_at<O>(hio, ((O)dstq << RS) | xb);
        }
#endif
        if (ELR) {  // s of the edge from its row: every lane of the head forms it (no broadcast of a gathered term)
          const int rel = tagq >> HET_TAG_REL_SHIFT;
          if (rel != rel_cur) {  // (a run lies in one relation: at most once per run)
            a4 = al_s[rel * LPR + sub];
            rel_cur = rel;
          }
          const float zq = quad_sum(dot4(f[q], a4)) + head_bcast<DL>(zrv, q, lane);
          online_edge(lrelu(zq, slope), zq > 0.f ? 1.f : slope, f[q], m, acc, ssum, accq, sq);
        } else {
          online_edge(head_bcast<DL>(sv, q, lane), head_bcast<DL>(dlv, q, lane), f[q], m, acc, ssum, accq, sq);
        }
        if (tagq & HET_TAG_LAST_RUN) {  // the run ends with this edge: its sums, relative to the maximum of this moment
          const int drowq = head_bcast_i<DL>(drowv, q, lane);
#ifndef HET_ABL_NO_Q
          st4_at<O>(qrow, ((O)drowq << RS) | xb, accq);
#endif
#ifndef HET_ABL_NO_QS
          if (d == 0) {
            st1_at<O>(qsum, ((O)drowq << HS) | hb, sq);
            st1_at<O>(qref, ((O)drowq << HS) | hb, m);
          }
#endif
          accq = make_float4(0.f, 0.f, 0.f, 0.f);
          sq = 0.f;
          if (tagq & HET_TAG_LAST_KEY) {  // ... and so does the destination
            const int dstq = head_bcast_i<DL>(dstv, q, lane);
            const float inv = __builtin_amdgcn_rcpf(ssum);
            const float4 r4 = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
#ifndef HET_ABL_NO_RET
            st4_at<O>(ret, ((O)dstq << RS) | xb, r4);
#endif
#ifndef HET_ABL_NO_HIO
            if (hio && dstq < hio_rows)
              st4_at<O>(hio, ((O)dstq << RS) | xb, make_float4(h0.x + r4.x, h0.y + r4.y, h0.z + r4.z, h0.w + r4.w));
#endif
#ifndef HET_ABL_NO_LSE
            if (d == 0) st1_at<O>(lse, ((O)dstq << HS) | hb, lse_fast(m, ssum));
#endif
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            ssum = 0.f;
            m = -INFINITY;
          }
        }
      }
    }
  };
  int4 idn = ids_of(b);
  asm volatile("" ::"v"(idn.x), "v"(idn.y), "v"(idn.z), "v"(idn.w));  // (in hand at the loop's entry too: see below)
  for (int j0 = b; j0 < e; j0 += U) {
    const int4 id = idn;
    float4 f[U];
    float zl, zr;
    rows_of(id, f, zl, zr);
    idn = ids_of(j0 + U);
    // The next ids are waited for HERE, together with the rows they were requested behind: left pending across the step's
    // conditional stores (which count in vmcnt on gfx9 and whose number the compiler cannot know) they made the top of every
    // trip an s_waitcnt vmcnt(0), i.e. a wait for the acknowledgement of the stores the step had just issued.
    asm volatile("" ::"v"(idn.x), "v"(idn.y), "v"(idn.z), "v"(idn.w));
    step(j0, id, f, zl, zr);
  }
}

__device__ __forceinline__ int64_t lower_bound_i32(const int32_t* __restrict__ a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Hubs: wave per hub work item of the grouping by (destination, relation) `it` (keys destination * R + relation; same sorted
// order as the grouping by destination whose packed ids p01 it reads; hub_items: grouping_hub_items -- a wave per item of
// that grouping with a test for "hub" spent 1.7 ms on 1.3 M early exits).  part[k] = {O[X], Q[X], max[H], sum[H], q[H]}.
template <int LPR, int DL, bool ELR, bool W64>
__global__ __launch_bounds__(kBlock) void HET_rgat_aggregate_hub_items(
    Items it, const int32_t* __restrict__ hub_items, const int32_t* __restrict__ hub_order, int64_t num_hub_items,
    const int2* __restrict__ p01, const float* __restrict__ feat, const float* __restrict__ el, const float* __restrict__ er,
    float slope, float* __restrict__ part, ElFold ef) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4, H = LPR / DL;
  constexpr int RS = het_log2_ce(LPR * 16), HS = het_log2_ce(H * 4);
  typedef typename OffSel<W64>::type O;
  static_assert(DL >= U, "a head needs at least U lanes");
  static_assert(!ELR || DL == 4, "el from the row: heads of 4 lanes");
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, d = sub % DL;
  const O xb = (O)(sub * 16), hb = (O)(h * 4);
  const int dq = d < U ? d : U - 1;
  const int64_t kk = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (kk >= num_hub_items) return;
  const int64_t k = hub_order ? hub_order[kk] : kk;  // launch order: by the first feat row of the item (grouping_hub_items)
  const int item = hub_items[k];
  const int b = it.begin[item], e = it.end[item];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accq = acc;
  float ssum = 0.f, sq = 0.f, m = -INFINITY;
  int jn = b + slot + dq * EPW < e ? b + slot + dq * EPW : e - 1;
  int2 idn = p01[jn];
  // an item of this grouping lies inside ONE run (one relation, one destination = one er row): its er term is loaded once
  const int2 first = p01[b];
  const float zrv = ld1_at<O>(er, ((O)first.y << HS) | hb);
  // (the item lies in one relation: its attention vector is loaded once)
  const float4 a4 = ELR ? ld4(ef.attn + (int64_t)el_relation(ef, first.x) * X + x) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    const int srowv = idn.x;
    float zlv = ELR ? 0.f : ld1_at<O>(el, ((O)srowv << HS) | hb);
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = ld4_at<O>(feat, ((O)head_bcast_i<DL>(srowv, u, lane) << RS) | xb);
    jn = j0 + (U + dq) * EPW < e ? j0 + (U + dq) * EPW : e - 1;
    idn = p01[jn];
    if (ELR) {  // el of the lane's own edge (edge dq of the step) from the rows: every lane forms all four, keeps its own
      float zl[U];
#pragma unroll
      for (int u = 0; u < U; ++u) zl[u] = quad_sum(dot4(f[u], a4));
      zlv = dq == 0 ? zl[0] : dq == 1 ? zl[1] : dq == 2 ? zl[2] : zl[3];
    }
    const float zv = zlv + zrv;
    const float sv = j0 + dq * EPW < e ? lrelu(zv, slope) : -INFINITY;
    {
      float mn = m;
#pragma unroll
      for (int u = 0; u < U; ++u) mn = fmaxf(mn, head_bcast<DL>(sv, u, lane));
      const float c = __expf(m - mn);
      acc.x *= c; acc.y *= c; acc.z *= c; acc.w *= c; ssum *= c;
      accq.x *= c; accq.y *= c; accq.z *= c; accq.w *= c; sq *= c;
      m = mn;
    }
    const float wv = __expf(sv - m), wdv = wv * (zv > 0.f ? 1.f : slope);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = head_bcast<DL>(wv, u, lane), wd = head_bcast<DL>(wdv, u, lane);
      acc.x = fmaf(w, f[u].x, acc.x); acc.y = fmaf(w, f[u].y, acc.y); acc.z = fmaf(w, f[u].z, acc.z); acc.w = fmaf(w, f[u].w, acc.w);
      accq.x = fmaf(wd, f[u].x, accq.x); accq.y = fmaf(wd, f[u].y, accq.y); accq.z = fmaf(wd, f[u].z, accq.z); accq.w = fmaf(wd, f[u].w, accq.w);
      ssum += w;
      sq += wd;
    }
  }
  {
    float M = m;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) M = fmaxf(M, __shfl_xor(M, off));
    const float c = m == -INFINITY ? 0.f : __expf(m - M);
    acc.x *= c; acc.y *= c; acc.z *= c; acc.w *= c; ssum *= c;
    accq.x *= c; accq.y *= c; accq.z *= c; accq.w *= c; sq *= c;
    m = M;
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    accq.x += __shfl_xor(accq.x, off); accq.y += __shfl_xor(accq.y, off); accq.z += __shfl_xor(accq.z, off); accq.w += __shfl_xor(accq.w, off);
    ssum += __shfl_xor(ssum, off);
    sq += __shfl_xor(sq, off);
  }
  if (slot != 0) return;
  float* pp = part + k * (2 * X + 3 * H);
  st4(pp + x, acc);
  st4(pp + X + x, accq);
  if (d == 0) { pp[2 * X + h] = m; pp[2 * X + H + h] = ssum; pp[2 * X + 2 * H + h] = sq; }
}

// One WORKGROUP per hub (hub_segs: segments of the grouping by destination): its runs are the segments of `it` with keys in
// [v * R, (v + 1) * R), its work items are consecutive.  First ret / lse over all items, then the sums of every run relative to lse.
// (Round 5: a wave per hub walked the records of the largest hub -- ~3 600 items on the ogbn-mag-like graph -- four at a time, and
// that one wave WAS the launch: 0.135 ms at the end of the forward's critical path.  The 16 lane groups of a workgroup take the
// records round-robin and meet in LDS; hubs of a few items leave three waves idle, which costs nothing at 17 K hubs.)
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rgat_finish_hubs(
    const int4* __restrict__ hub_rec, int64_t num_hubs, Items it, const int32_t* __restrict__ p_drow,
    const float* __restrict__ part, float* __restrict__ lse, float* __restrict__ ret, int H,
    int D, float* __restrict__ hio, int64_t hio_rows, float* __restrict__ qrow, float* __restrict__ qsum,
    float* __restrict__ qref) {
  constexpr int EPW = 64 / LPR, NW = kBlock / 64;
  __shared__ float4 s_acc[NW][LPR];
  __shared__ float s_m[NW][LPR], s_sum[NW][LPR], s_L[LPR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D;
  const int64_t k = blockIdx.x;
  if (k >= num_hubs) return;
  // {first run, one past the last run, first record, destination} of the hub: searched once per grouping (grouping_hub_items) -- as
  // five dependent binary searches per hub HERE they were ~100 dependent loads in front of every wave, i.e. the launch's 0.13 ms
  const int4 hr = hub_rec[k];
  const int64_t X = (int64_t)H * D, v = hr.w, rec = 2 * X + 3 * H;
  const int64_t s_lo = hr.x, s_hi = hr.y, i_lo = hr.z;
  int64_t n_all = 0;
  for (int64_t s2 = s_lo; s2 < s_hi; ++s2) n_all += (it.seg_ptr[s2 + 1] - it.seg_ptr[s2] + HET_ITEM_MAX - 1) / HET_ITEM_MAX;
  const int64_t i_hi = i_lo + n_all;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float ssum = 0.f, m = -INFINITY;
  for (int64_t i = i_lo + wave * EPW + slot; i < i_hi; i += NW * EPW) {
    const float* pp = part + i * rec;
    const float mi = pp[2 * X + h], si = pp[2 * X + H + h];
    const float4 a = ld4(pp + x);
    const float mn = fmaxf(m, mi), c = __expf(m - mn), ci = __expf(mi - mn);
    acc.x = acc.x * c + a.x * ci; acc.y = acc.y * c + a.y * ci; acc.z = acc.z * c + a.z * ci; acc.w = acc.w * c + a.w * ci;
    ssum = ssum * c + si * ci;
    m = mn;
  }
  float M = m;
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) M = fmaxf(M, __shfl_xor(M, off));
  {
    const float c = m == -INFINITY ? 0.f : __expf(m - M);
    acc.x *= c; acc.y *= c; acc.z *= c; acc.w *= c; ssum *= c;
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    ssum += __shfl_xor(ssum, off);
  }
  if (slot == 0) { s_acc[wave][sub] = acc; s_m[wave][sub] = M; s_sum[wave][sub] = ssum; }  // (M = -inf, sums 0: a wave without items)
  __syncthreads();
  if (wave == 0 && slot == 0) {  // the waves' partial results to their common maximum
    float Mx = s_m[0][sub];
#pragma unroll
    for (int w = 1; w < NW; ++w) Mx = fmaxf(Mx, s_m[w][sub]);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    float ts = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float c = s_m[w][sub] == -INFINITY ? 0.f : __expf(s_m[w][sub] - Mx);
      const float4 a = s_acc[w][sub];
      t.x = fmaf(a.x, c, t.x); t.y = fmaf(a.y, c, t.y); t.z = fmaf(a.z, c, t.z); t.w = fmaf(a.w, c, t.w);
      ts = fmaf(s_sum[w][sub], c, ts);
    }
    const float L = lse_of(Mx, ts);
    s_L[sub] = L;
    const float inv = 1.f / ts;
    const float4 r4 = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    st4(ret + v * X + x, r4);
    if (hio && v < hio_rows) {
      const float4 h0 = ld4(hio + v * X + x);
      st4(hio + v * X + x, make_float4(h0.x + r4.x, h0.y + r4.y, h0.z + r4.z, h0.w + r4.w));
    }
    if (x % D == 0) lse[v * H + h] = L;
  }
  __syncthreads();
  const float L = s_L[sub];
  int64_t i0 = i_lo;
  for (int64_t s2 = s_lo; s2 < s_hi; ++s2) {
    const int64_t n_items = (it.seg_ptr[s2 + 1] - it.seg_ptr[s2] + HET_ITEM_MAX - 1) / HET_ITEM_MAX;
    float4 aq = make_float4(0.f, 0.f, 0.f, 0.f);
    float sq = 0.f;
    for (int64_t i = i0 + wave * EPW + slot; i < i0 + n_items; i += NW * EPW) {
      const float* pp = part + i * rec;
      const float ci = __expf(pp[2 * X + h] - L);
      const float4 a = ld4(pp + X + x);
      aq.x = fmaf(a.x, ci, aq.x); aq.y = fmaf(a.y, ci, aq.y); aq.z = fmaf(a.z, ci, aq.z); aq.w = fmaf(a.w, ci, aq.w);
      sq = fmaf(pp[2 * X + 2 * H + h], ci, sq);
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      aq.x += __shfl_xor(aq.x, off); aq.y += __shfl_xor(aq.y, off);
      aq.z += __shfl_xor(aq.z, off); aq.w += __shfl_xor(aq.w, off);
      sq += __shfl_xor(sq, off);
    }
    __syncthreads();  // (the previous round's LDS slots have been read)
    if (slot == 0) { s_acc[wave][sub] = aq; s_sum[wave][sub] = sq; }
    __syncthreads();
    if (wave == 0 && slot == 0) {
      float4 t = s_acc[0][sub];
      float ts = s_sum[0][sub];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float4 a = s_acc[w][sub];
        t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
        ts += s_sum[w][sub];
      }
      const int64_t wrow = p_drow[it.seg_ptr[s2]];  // er row of the run
      st4(qrow + wrow * X + x, t);
      if (x % D == 0) { qsum[wrow * H + h] = ts; qref[wrow * H + h] = L; }
    }
    i0 += n_items;
  }
}

// grad_er[w,h] = exp(ref[w,h] - lse[v,h]) (<gradout[v,h,:], Q[w,h,:]> - <gradout, ret>[v,h] q[w,h]),  v = drow_nodes[w];
// pack2 [N,H,2] = {lse, <gradout, ret>}.  An er row without edges (q == 0, rows never written) gets 0.
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rgat_grad_er_runs(const float* __restrict__ qrow, const float* __restrict__ qsum,
                                                                 const float* __restrict__ qref, const int64_t* __restrict__ drow_nodes,
                                                                 const float* __restrict__ pack2, const float* __restrict__ gradout,
                                                                 float* __restrict__ grad_er, int64_t n_rows, int H, int D) {
  constexpr int EPW = 64 / LPR, X = LPR * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / D, DL = D >> 2;
  const int64_t step = (int64_t)gridDim.x * (kBlock / 64) * EPW;
  for (int64_t w0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * EPW; w0 < n_rows; w0 += step) {
    const bool ok = w0 + slot < n_rows;
    const int64_t w = ok ? w0 + slot : n_rows - 1;
    const int64_t v = drow_nodes[w];
    const float sq = qsum[w * H + h];
    const float4 g = ld4(gradout + v * X + x), q = ld4(qrow + w * X + x);
    const float2 pkv = *reinterpret_cast<const float2*>(pack2 + (v * H + h) * 2);
    const float ref = qref[w * H + h];
    float dot = g.x * q.x + g.y * q.y + g.z * q.z + g.w * q.w;
    for (int off = DL >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    if (ok && (sub & (DL - 1)) == 0) grad_er[w * H + h] = sq == 0.f ? 0.f : __expf(ref - pkv.x) * (dot - pkv.y * sq);
  }
}

// ---- weight gradient of the attention vector from the rows the source-row kernels hold ---------------------------------
// grad_attn_l[r,h,:] = SUM_u grad_el[u,h] feat_c[u,h,:] over the rows u of relation r: a separate row-dot pass read feat_c again
// (0.6 GB, 0.15 ms alone and 0.4 ms beside the matrix-core passes).  The source-row kernels have both factors in registers where
// a segment (or a piece of one: the sum is linear) ends, so they keep ga += grad_el * feat per lane group; the workgroup's
// partial row (rows are relation-major: a workgroup sees one relation, two at a boundary) goes to part[workgroup] with its
// relation, and HET_rgat_attn_grad_finish adds the partial rows up.  Pieces that do not fit that pattern (a second relation inside
// a lane group, a wave or a workgroup) are added with float atomics -- a handful per launch.
template <int LPR>
__device__ __forceinline__ void ga_flush_atomic(float* __restrict__ out, int rel, int x, const float4& v) {
  float* p = out + (int64_t)rel * (LPR * 4) + x;
  atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}
// every thread of the workgroup calls this once; (ga, rel) per lane group, rel < 0: none
template <int LPR>
__device__ __forceinline__ void ga_block_reduce(float4 ga, int rel, float* __restrict__ part, int* __restrict__ part_rel,
                                                float* __restrict__ out) {
  constexpr int X = LPR * 4, NW = kBlock / 64;
  __shared__ float sg[NW][X];
  __shared__ int sr[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane / LPR, x = (lane % LPR) * 4;
  int r0 = rel;
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) r0 = max(r0, __shfl_xor(r0, off));
  if (rel >= 0 && rel != r0) {  // this lane group belongs to another relation than the wave's: added directly
    ga_flush_atomic<LPR>(out, rel, x, ga);
    ga = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (rel < 0) ga = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    ga.x += __shfl_xor(ga.x, off); ga.y += __shfl_xor(ga.y, off);
    ga.z += __shfl_xor(ga.z, off); ga.w += __shfl_xor(ga.w, off);
  }
  if (slot == 0) st4(&sg[wave][x], ga);
  if (lane == 0) sr[wave] = r0;
  __syncthreads();
  if (wave == 0 && slot == 0) {
    int R0 = -1;
#pragma unroll
    for (int w = 0; w < NW; ++w) R0 = max(R0, sr[w]);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float4 v = ld4(&sg[w][x]);
      if (sr[w] == R0) { t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      else if (sr[w] >= 0) ga_flush_atomic<LPR>(out, sr[w], x, v);
    }
    st4(part + (int64_t)blockIdx.x * X + x, t);
    if (lane == 0) part_rel[blockIdx.x] = R0;
  }
}
constexpr int kAttnFinishRows = 128;  // partial rows per workgroup of the finishing pass (52 workgroups of 1024 rows took 0.2 ms)
// out[r, :] += SUM of the partial rows tagged r (out zeroed before the producers ran: they add boundary pieces atomically)
__global__ __launch_bounds__(kBlock) void HET_rgat_attn_grad_finish(const float* __restrict__ part, const int* __restrict__ part_rel,
                                                                     int64_t P, int X, float* __restrict__ out) {
  const int x = threadIdx.x % X, sub = threadIdx.x / X, nsub = kBlock / X;
  constexpr int kRowsPerBlock = kAttnFinishRows;
  const int64_t p0 = (int64_t)blockIdx.x * kRowsPerBlock, p1 = p0 + kRowsPerBlock < P ? p0 + kRowsPerBlock : P;
  int cur = -1;
  float a = 0.f;
  for (int64_t p = p0 + sub; p < p1; p += nsub) {
    const int r = part_rel[p];
    if (r != cur) {
      if (cur >= 0) atomicAdd(out + (int64_t)cur * X + x, a);
      cur = r;
      a = 0.f;
    }
    if (r >= 0) a += part[p * X + x];
  }
  if (cur >= 0) atomicAdd(out + (int64_t)cur * X + x, a);
}

// Backward, cooperative form of HET_rgat_backward_src_packed.  pack2 [N,H,2] = {lse, <gradout, ret>} interleaved.
// GA: also the partial rows of grad_attn_l (ga_block_reduce above; needs fold_w for the relation of a row)
// REC: er / lse / <gradout, ret> of an edge come from ONE 16-byte record per (er row, head) -- rec4 [S_col, H] {er, lse, dot, 0},
// HET_rgat_drow_rec -- instead of a 4-byte gather from er and an 8-byte one from pack2: a vector-memory instruction and a
// 128-byte line less per edge (the gathers of these kernels miss L2 once per table and edge: profiles/r04/locality_counters.txt)
// Round 5 (see HET_rgat_aggregate_runs_packed): where a segment starts / ends and the relation of its row come from the tag of the
// packed id record (no carried key, no compares against the neighbours, no boundary search), addresses are 32-bit byte offsets off
// scalar bases (W64 = false), the dot over the lanes of a head is two DPP adds instead of two ds_bpermute.  H = LPR / DL.
template <int LPR, int DL, bool GA, bool REC, bool W64>
__global__ __launch_bounds__(kBlock, GA ? 5 : 1) void HET_rgat_backward_src_coop(
    Packs pk, const int4* __restrict__ kp01, const float* __restrict__ feat,
    const float* __restrict__ el, const float* __restrict__ er, const float* __restrict__ pack2,
    const float* __restrict__ gradout, float* __restrict__ grad_feat, float* __restrict__ grad_el,
    float* __restrict__ tbuf, float slope, const float* __restrict__ fold_w,
    const idx_t* __restrict__ fold_row_rel_ptrs, int R, float* __restrict__ ga_part, int* __restrict__ ga_rel,
    float* __restrict__ ga_out) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4, H = LPR / DL;
  constexpr int RS = het_log2_ce(LPR * 16), HS = het_log2_ce(H * 4);
  typedef typename OffSel<W64>::type O;
  static_assert(DL >= U, "a head needs at least U lanes");
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, d = sub % DL;
  const O xb = (O)(sub * 16), hb = (O)(h * 4);
  const int dq = d < U ? d : U - 1;
  const int64_t pid = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * EPW + slot;
  if (!GA && pid >= pk.n) return;
  const int64_t pidc = pid < pk.n ? pid : pk.n - 1;  // (GA: every lane group stays for the workgroup reduction, with an empty range)
  const uint32_t pb = (uint32_t)pk.ptr[pidc];
  int b = (int)(pb & 0x7fffffffu), e = (int)((uint32_t)pk.ptr[pidc + 1] & 0x7fffffffu);
  if (!GA && (pb >> 31)) return;  // a long segment: HET_rgat_backward_src_long takes its work items
  if (GA && (pid >= pk.n || (pb >> 31))) e = b;
  float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
  int jn = b + dq < e ? b + dq : e - 1;
  if (GA && jn < 0) jn = 0;
  int4 idn = kp01[jn];  // {feat row, destination, er row, tag} of the edge: one load (grouping_packed_ids, grouping_tag_kp01)
  asm volatile("" ::"v"(idn.x), "v"(idn.y), "v"(idn.z), "v"(idn.w));  // (in hand at the loop's entry: see the forward)
  int rel_cur = -1;
  float4 fcur = make_float4(0.f, 0.f, 0.f, 0.f), wcur = fcur, acc = fcur;
  float acc_el = 0.f;
  for (int j0 = b; j0 < e; j0 += U) {
    const int keyv = idn.x, dstv = idn.y, drowv = idn.z, tagv = idn.w;
    // scalars of the step: lane (h, q) fetches those of edge q, head h
    float zrv;
    float2 pkv;
    if (REC) {
      const float4 rv = ld4_at<O>(er, ((O)drowv << (HS + 2)) | (O)(h * 16));  // (er: the record table)
      zrv = rv.x; pkv = make_float2(rv.y, rv.z);
    } else {
      zrv = ld1_at<O>(er, ((O)drowv << HS) | hb);
      pkv = ld2_at<O>(pack2, ((O)dstv << (HS + 1)) | (O)(h * 8));
    }
    const float zlv = ld1_at<O>(el, ((O)keyv << HS) | hb);
    int tag[U];
    float4 g[U], fl[U];
#pragma unroll
    for (int q = 0; q < U; ++q) tag[q] = head_bcast_i<DL>(tagv, q, lane);
#pragma unroll
    for (int q = 0; q < U; ++q) g[q] = ld4_at<O>(gradout, ((O)head_bcast_i<DL>(dstv, q, lane) << RS) | xb);
    // feat row of a segment that starts inside this step (uniform per lane group).  fl[q] has NO other definition: a value merged
    // with the row in hand (fq[q] = fcur; if (start) fq[q] = load) made the compiler copy the loaded registers right behind the load,
    // i.e. wait for every load in flight at each segment start
#pragma unroll
    for (int q = 0; q < U; ++q)
      if (j0 + q < e && (tag[q] & HET_TAG_FIRST_KEY)) fl[q] = ld4_at<O>(feat, ((O)head_bcast_i<DL>(keyv, q, lane) << RS) | xb);
    // ids of the next step, in flight while this step's rows arrive
    jn = j0 + U + dq < e ? j0 + U + dq : e - 1;
    idn = kp01[jn];
    // per (edge, head) once: attention weight, its leaky-ReLU branch, the destination's <gradout, ret>
    const float zv = zlv + zrv;
    const float av = j0 + dq < e ? __expf(lrelu(zv, slope) - pkv.x) : 0.f;  // pkv.x: lse of the destination
    const float adv = av * (zv > 0.f ? 1.f : slope);
    asm volatile("" ::"v"(idn.x), "v"(idn.y), "v"(idn.z), "v"(idn.w));  // (waited for before the step's stores: see the forward)
    float tq[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const bool ok = j0 + q < e;  // uniform within the lane group
      if (ok && (tag[q] & HET_TAG_FIRST_KEY)) {
        fcur = fl[q];
        if (fold_w) {
          const int rel = tag[q] >> HET_TAG_REL_SHIFT;
          if (rel != rel_cur) {  // rows are relation-major: a handful of times per launch
            if (GA) {  // (the segments summed so far belong to the old relation)
              if (rel_cur >= 0) ga_flush_atomic<LPR>(ga_out, rel_cur, x, ga);
              ga = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            wcur = ld4(fold_w + (int64_t)rel * X + x);
            rel_cur = rel;
          }
        }
      }
      const float a = head_bcast<DL>(av, q, lane), ad = head_bcast<DL>(adv, q, lane), gr = head_bcast<DL>(pkv.y, q, lane);
      acc.x = fmaf(a, g[q].x, acc.x); acc.y = fmaf(a, g[q].y, acc.y);
      acc.z = fmaf(a, g[q].z, acc.z); acc.w = fmaf(a, g[q].w, acc.w);
      const float dot = head_sum<DL>(dot4(g[q], fcur));
      const float t = ad * (dot - gr);  // 0 for the padding edges of the last step (a == 0)
      tq[q] = t;
      acc_el += t;  // identical in the DL lanes of a head
      if (ok && (tag[q] & HET_TAG_LAST_KEY)) {  // the segment ends with this edge: one store
        const int u = head_bcast_i<DL>(keyv, q, lane);
        float4 o = acc;
        if (fold_w) {  // el[u,h] = <feat[u,h,:], fold_w[r(u),h,:]>: its gradient joins grad_feat here
          o.x = fmaf(acc_el, wcur.x, o.x); o.y = fmaf(acc_el, wcur.y, o.y);
          o.z = fmaf(acc_el, wcur.z, o.z); o.w = fmaf(acc_el, wcur.w, o.w);
        }
        st4_at<O>(grad_feat, ((O)u << RS) | xb, o);
        if (grad_el && d == 0) st1_at<O>(grad_el, ((O)u << HS) | hb, acc_el);  // (NULL: nobody reads it -- the fold and the attention gradient are formed here)
        if (GA) {  // grad_attn_l[rel_cur] += grad_el[u] * feat[u]  (flushed where rel_cur changes, above)
          ga.x = fmaf(acc_el, fcur.x, ga.x); ga.y = fmaf(acc_el, fcur.y, ga.y);
          ga.z = fmaf(acc_el, fcur.z, ga.z); ga.w = fmaf(acc_el, fcur.w, ga.w);
        }
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        acc_el = 0.f;
      }
    }
    // t of edge q, head h leaves through lane (h, q): one 16-byte-per-edge store instruction per step
    if (tbuf) {  // (NULL: grad_er comes from the run sums)
      float ts = tq[0];
#pragma unroll
      for (int q = 1; q < U; ++q) ts = d == q ? tq[q] : ts;
      // (writing tbuf in the order of the grouping by er row instead -- scattered 16-byte stores, a streaming segmented
      //  sum afterwards -- was measured: +0.38 ms here and in the long-segment kernel, -0.40 ms there)
      if (d < U && j0 + d < e) tbuf[(int64_t)(j0 + d) * H + h] = ts;
    }
  }
  if (GA) ga_block_reduce<LPR>(ga, rel_cur, ga_part, ga_rel, ga_out);
}


// Backward for the LONG (relation, source) segments (> HET_PACK_T edges; 61 % of the edges of the skewed ogbn-mag-like
// graph): wave per work item (<= HET_ITEM_MAX edges of ONE feat row), the 64/LPR lane groups take its edges round-robin
// as the forward does, scalars fetched cooperatively; feat row, el and the fold row are per item.  One store per item
// (atomic adds only for the items of a segment longer than HET_ITEM_MAX, whose rows HET_rgat_zero_long_rows cleared).
template <int LPR, int DL, bool GA, bool REC, bool W64>
__global__ __launch_bounds__(kBlock) void HET_rgat_backward_src_long(
    Items it, const int32_t* __restrict__ long_items, int64_t num_long_items, const int2* __restrict__ p01,
    const float* __restrict__ feat, const float* __restrict__ el,
    const float* __restrict__ er, const float* __restrict__ pack2, const float* __restrict__ gradout,
    float* __restrict__ grad_feat, float* __restrict__ grad_el, float* __restrict__ tbuf, float slope,
    const float* __restrict__ fold_w, const idx_t* __restrict__ fold_row_rel_ptrs, int R, float* __restrict__ ga_part,
    int* __restrict__ ga_rel, float* __restrict__ ga_out) {
  constexpr int EPW = 64 / LPR, U = 4, X = LPR * 4, H = LPR / DL;
  constexpr int RS = het_log2_ce(LPR * 16), HS = het_log2_ce(H * 4);
  typedef typename OffSel<W64>::type O;
  static_assert(DL >= U, "a head needs at least U lanes");
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, d = sub % DL;
  const O xb = (O)(sub * 16), hb = (O)(h * 4);
  const int dq = d < U ? d : U - 1;
  const int64_t wid = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (wid >= num_long_items) {  // (GA: the wave stays for the workgroup reduction)
    if (GA) ga_block_reduce<LPR>(make_float4(0.f, 0.f, 0.f, 0.f), -1, ga_part, ga_rel, ga_out);
    return;
  }
  const int item = long_items[wid];
  const int seg = it.seg[item], b = it.begin[item], e = it.end[item];
  int jn = b + slot + dq * EPW < e ? b + slot + dq * EPW : e - 1;
  int2 idn = p01[jn];
  const int64_t u = it.seg_key[seg];
  const bool whole = b == it.seg_ptr[seg] && e == it.seg_ptr[seg + 1];
  const float4 f = ld4(feat + u * X + x);
  const float zl = el[u * H + h];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float acc_el = 0.f;
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    const int dstv = idn.x, drowv = idn.y;
    float zrv;
    float2 pkv;
    if (REC) {
      const float4 rv = ld4_at<O>(er, ((O)drowv << (HS + 2)) | (O)(h * 16));  // (er: the record table, see HET_rgat_backward_src_coop)
      zrv = rv.x; pkv = make_float2(rv.y, rv.z);
    } else {
      zrv = ld1_at<O>(er, ((O)drowv << HS) | hb);
      pkv = ld2_at<O>(pack2, ((O)dstv << (HS + 1)) | (O)(h * 8));
    }
    float4 g[U];
#pragma unroll
    for (int q = 0; q < U; ++q) g[q] = ld4_at<O>(gradout, ((O)head_bcast_i<DL>(dstv, q, lane) << RS) | xb);
    jn = j0 + (U + dq) * EPW < e ? j0 + (U + dq) * EPW : e - 1;
    idn = p01[jn];
    const float zv = zl + zrv;
    const float av = j0 + dq * EPW < e ? __expf(lrelu(zv, slope) - pkv.x) : 0.f;  // pkv.x: lse of the destination
    const float adv = av * (zv > 0.f ? 1.f : slope);
    float tq[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const float a = head_bcast<DL>(av, q, lane), ad = head_bcast<DL>(adv, q, lane), gr = head_bcast<DL>(pkv.y, q, lane);
      acc.x = fmaf(a, g[q].x, acc.x); acc.y = fmaf(a, g[q].y, acc.y);
      acc.z = fmaf(a, g[q].z, acc.z); acc.w = fmaf(a, g[q].w, acc.w);
      const float dot = head_sum<DL>(dot4(g[q], f));
      tq[q] = ad * (dot - gr);  // 0 for the padding edges of the last step
      acc_el += tq[q];
    }
    if (tbuf) {
      float ts = tq[0];
#pragma unroll
      for (int q = 1; q < U; ++q) ts = d == q ? tq[q] : ts;
      if (d < U && j0 + d * EPW < e) tbuf[(int64_t)(j0 + d * EPW) * H + h] = ts;
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    acc_el += __shfl_xor(acc_el, off);
  }
  if (GA) {  // this item's piece of grad_attn_l[rel] = grad_el piece * feat row (lane group 0 carries it)
    const int rel = find_segment(fold_row_rel_ptrs, R, (idx_t)u);
    const float s0 = slot == 0 ? acc_el : 0.f;
    ga_block_reduce<LPR>(make_float4(s0 * f.x, s0 * f.y, s0 * f.z, s0 * f.w), rel, ga_part, ga_rel, ga_out);
  }
  if (slot != 0) return;
  if (fold_w) {
    const float4 w = ld4(fold_w + (int64_t)find_segment(fold_row_rel_ptrs, R, (idx_t)u) * X + x);
    acc.x = fmaf(acc_el, w.x, acc.x); acc.y = fmaf(acc_el, w.y, acc.y);
    acc.z = fmaf(acc_el, w.z, acc.z); acc.w = fmaf(acc_el, w.w, acc.w);
  }
  float* gp = grad_feat + u * X + x;
  if (whole) {
    st4(gp, acc);
    if (grad_el && d == 0) grad_el[u * H + h] = acc_el;
  } else {
    atomicAdd(gp + 0, acc.x); atomicAdd(gp + 1, acc.y); atomicAdd(gp + 2, acc.z); atomicAdd(gp + 3, acc.w);
    if (grad_el && d == 0) atomicAdd(&grad_el[u * H + h], acc_el);
  }
}

// rec4[w, h] = {er[w, h], lse[v, h], <gradout, ret>[v, h], 0}, v = drow_nodes[w]: everything the source-row kernels need per edge
// from the destination side, one 16-byte record per (er row, head).  pack2 [N,H,2] interleaved (HET_rgat_dst_pack).
__global__ __launch_bounds__(kBlock) void HET_rgat_drow_rec(const float* __restrict__ er, const float* __restrict__ pack2,
                                                             const int64_t* __restrict__ drow_nodes, int64_t n_rows, int H,
                                                             float* __restrict__ rec4) {
  const int64_t total = n_rows * H;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t w = t / H;
    const int h = (int)(t - w * H);
    const int64_t v = drow_nodes[w];
    const float2 p = *reinterpret_cast<const float2*>(pack2 + (v * H + h) * 2);
    st4(rec4 + t * 4, make_float4(er[t], p.x, p.y, 0.f));
  }
}

// Round 5: everything the backward needs per er row w = (relation, destination v) in ONE pass, instead of a pass over all N nodes
// (HET_rgat_dst_pack: {lse, <gradout, ret>} per node), a pass over the er rows that re-packs it (HET_rgat_drow_rec) and, after the
// source-row kernels, a third one (HET_rgat_grad_er_runs) that reads gradout[v] once more:
//   rec4[w,h]    = {er[w,h], lse[v,h], <gradout[v,h,:], ret[v,h,:]>, 0}                    (read per edge by the source-row kernels)
//   grad_er[w,h] = exp(ref[w,h] - lse[v,h]) (<gradout[v,h,:], Q[w,h,:]> - <gradout, ret>[v,h] q[w,h]),  0 for a row without edges
// gradout / ret rows of a destination with several relations are read once per relation (adjacent work: cache hits); nodes
// without in-edges are not visited at all.  Lane group per er row, two rows per lane group in flight.
template <int LPR, int DL>
__global__ __launch_bounds__(kBlock) void HET_rgat_drow_pass(const float* __restrict__ er, const float* __restrict__ lse,
                                                              const float* __restrict__ ret, const float* __restrict__ gradout,
                                                              const float* __restrict__ qrow, const float* __restrict__ qsum,
                                                              const float* __restrict__ qref, const int64_t* __restrict__ drow_nodes,
                                                              const int32_t* __restrict__ order, int64_t n_rows,
                                                              float* __restrict__ rec4, float* __restrict__ grad_er) {
  constexpr int EPW = 64 / LPR, X = LPR * 4, H = LPR / DL, U = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = sub / DL, d = sub % DL;
  const int64_t w0 = (((int64_t)blockIdx.x * (kBlock / 64) + wave) * U) * EPW + slot;
  int64_t w[U], v[U];
  bool ok[U];
  // order (optional): the er rows by their destination node, so that the rows of one node -- one per relation that reaches it -- sit
  // in neighbouring lane groups and its gradout / ret rows cross the memory interface once (the list itself is relation-major: the
  // two rows of a paper of ogbn-mag are 0.7 M rows apart)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    ok[u] = w0 + u * EPW < n_rows;
    const int64_t j = ok[u] ? w0 + u * EPW : n_rows - 1;
    w[u] = order ? (int64_t)order[j] : j;
    v[u] = drow_nodes[w[u]];
  }
  float4 g[U], r[U], q[U];
  float erv[U], ls[U], qs[U], rf[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    g[u] = ld4(gradout + v[u] * X + x);
    r[u] = ld4(ret + v[u] * X + x);
    q[u] = ld4(qrow + w[u] * X + x);
    erv[u] = er[w[u] * H + h];
    ls[u] = lse[v[u] * H + h];
    qs[u] = qsum[w[u] * H + h];
    rf[u] = qref[w[u] * H + h];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float dr = head_sum<DL>(dot4(g[u], r[u])), dq = head_sum<DL>(dot4(g[u], q[u]));
    if (ok[u] && d == 0) {
      // (a row without edges: q == 0, Q / ref never written, and its destination's ret row may be unwritten too -- nothing of it is used)
      const bool has = qs[u] != 0.f;
      st4(rec4 + (w[u] * H + h) * 4, make_float4(erv[u], ls[u], has ? dr : 0.f, 0.f));
      grad_er[w[u] * H + h] = has ? __expf(rf[u] - ls[u]) * (dq - dr * qs[u]) : 0.f;
    }
  }
}

// bias_part [gridDim.x * waves, X]: per-wave column sums of rows [0, n) of g [., X] (grid-stride; HET_rgat_colsum_finish adds them up)
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rgat_colsum_rows(const float* __restrict__ g, int64_t n, float* __restrict__ bias_part) {
  constexpr int EPW = 64 / LPR, X = LPR * 4, U = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, x = (lane % LPR) * 4;
  const int64_t step = (int64_t)gridDim.x * (kBlock / 64) * EPW * U;
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t v0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * EPW * U + slot; v0 < n; v0 += step) {
    float4 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = v0 + u * EPW < n ? ld4(g + (v0 + u * EPW) * X + x) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < U; ++u) { bs.x += a[u].x; bs.y += a[u].y; bs.z += a[u].z; bs.w += a[u].w; }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    bs.x += __shfl_xor(bs.x, off); bs.y += __shfl_xor(bs.y, off);
    bs.z += __shfl_xor(bs.z, off); bs.w += __shfl_xor(bs.w, off);
  }
  if (slot == 0) st4(bias_part + ((int64_t)blockIdx.x * (kBlock / 64) + wave) * X + x, bs);
}

// rows of the long segments start from zero (their pieces add atomically)
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rgat_zero_long_rows(const int32_t* __restrict__ long_seg,
                                                                   const int32_t* __restrict__ seg_key, int64_t n,
                                                                   float* __restrict__ grad_feat, float* __restrict__ grad_el,
                                                                   int H) {
  constexpr int X = LPR * 4;
  const int64_t total = n * LPR;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t k = t / LPR;
    const int sub = (int)(t - k * LPR);
    const int64_t u = seg_key[long_seg[k]];
    st4(grad_feat + u * X + sub * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    if (grad_el && sub < H) grad_el[u * H + sub] = 0.f;
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = ceil_div64(total, kBlock);
  const int64_t cap = 256 * 64;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline bool is_pow2(int64_t x) { return x > 0 && (x & (x - 1)) == 0; }

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
#define HET_DISPATCH_COOP(LPRV, DLV, CALL)                                  \
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
static bool coop_shape_ok(int64_t H, int64_t D) {
  static const bool off = [] { const char* v = getenv("HET_RGAT_COOP"); return v && v[0] == '0'; }();  // A/B switch
  const int64_t lpr = H * D / 4, dl = D / 4;
  return !off && (lpr == 8 || lpr == 16 || lpr == 32) && dl >= 4 && dl <= lpr;
}

static bool compact_shape_ok(int64_t H, int64_t D) {
  const int64_t X = H * D;
  return is_pow2(D) && D >= 4 && is_pow2(X) && X / 4 <= 64 && H <= X / 4;
}

// bytes of het_rgat_aggregate_compact's workspace: one {acc[H*D], max[H], sum[H]} record per work item when the grouping has
// destinations that are split over several work items (more than HET_ITEM_MAX in-edges); only the records of those items are touched
extern "C" int64_t het_rgat_aggregate_compact_workspace(const het_grouping* by_dst, int64_t H, int64_t D) {
  if (!by_dst || by_dst->num_split == 0) return 0;
  return (int64_t)sizeof(float) * by_dst->num_items * (H * D + 2 * H);
}

extern "C" int het_rgat_aggregate_compact(const het_grouping* by_dst, const float* feat_c, const float* el_c,
                                          const float* er_c, float* sum, float* ret, int64_t num_nodes, int64_t H,
                                          int64_t D, double slope, float* h_inout, int64_t h_rows, void* workspace,
                                          int64_t workspace_bytes, het_stream stream) {
  const char* op = "het_rgat_aggregate_compact";
  hipStream_t s = (hipStream_t)stream;
  HET_REQUIRE(by_dst && sum && ret && num_nodes >= 0, "%s: null argument", op);
  if (!compact_shape_ok(H, D)) { het_set_error("%s: unsupported shape H=%lld D=%lld", op, (long long)H, (long long)D); return HET_ERR_UNSUPPORTED; }
  HET_REQUIRE(by_dst->R == 0 && by_dst->key_bound <= num_nodes && (by_dst->E == 0 || (by_dst->p0 && by_dst->p1 && feat_c && el_c && er_c)),
              "%s: by_dst must group the positions by destination with payload0 = feat row and payload1 = er row", op);
  const int64_t need = het_rgat_aggregate_compact_workspace(by_dst, H, D);
  HET_REQUIRE(need == 0 || (workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0),
              "%s: a 16-byte aligned workspace of %lld bytes is needed (het_rgat_aggregate_compact_workspace)", op, (long long)need);
  const int64_t X = H * D;
  HET_HIP(hipMemsetAsync(sum, 0, sizeof(float) * num_nodes * H, s));  // destinations without in-edges (never read by an edge)
  // ret: zero rows for destinations without in-edges -- unless the caller takes the layer output through h_inout and
  // reads ret only where edges point (the backward): then nothing is filled (0.5 GB less)
  if (!h_inout) HET_HIP(hipMemsetAsync(ret, 0, sizeof(float) * num_nodes * X, s));
  if (by_dst->E == 0) return HET_OK;
  Items it{by_dst->item_seg, by_dst->item_begin, by_dst->item_end, by_dst->seg_ptr, by_dst->seg_key, by_dst->num_items};
  const unsigned nb = (unsigned)ceil_div64(by_dst->num_items, kBlock / 64);
  float* part = static_cast<float*>(workspace);
  if (coop_shape_ok(H, D))
    if (int rc = grouping_packed_ids(by_dst, false, s)) return rc;
  {
    HET_KTIME("HET_rgat_aggregate", s);
    if (coop_shape_ok(H, D)) {
      HET_DISPATCH_COOP((int)(X / 4), (int)(D / 4),
                        hipLaunchKernelGGL((HET_rgat_aggregate_coop<LPR, DL>), dim3(nb), dim3(kBlock), 0, s, it, by_dst->p01,
                                           feat_c, el_c, er_c, sum, ret, (int)H, (float)slope, h_inout, h_rows, part));
    } else {
      HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL(HET_rgat_aggregate_compact<LPR>, dim3(nb), dim3(kBlock), 0, s, it,
                                                        by_dst->p0, by_dst->p1, feat_c, el_c, er_c, sum, ret, (int)H, (int)D,
                                                        (float)slope, h_inout, h_rows, part));
    }
  }
  HET_LAUNCH_CHECK("HET_rgat_aggregate_compact");
  if (by_dst->num_split > 0) {  // the pieces of the hub destinations: common maximum, division, lse
    const unsigned nbs = (unsigned)ceil_div64(by_dst->num_split, kBlock / 64);
    HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL(HET_rgat_finish_split<LPR>, dim3(nbs), dim3(kBlock), 0, s, by_dst->split_seg,
                                                      by_dst->num_split, it, part, sum, ret, (int)H, (int)D, h_inout, h_rows));
    HET_LAUNCH_CHECK("HET_rgat_finish_split");
  }
  return HET_OK;
}

// ---- forward that also leaves the run sums (see "grad_er without a per-edge term" above) ----
// Destinations with more in-edges than this go to the wave-per-item hub path (parked partial sums), the others are walked in
// order by one lane group.  At least HET_PACK_T (shorter destinations share a pack); measured on ogbn-mag: see DESIGN.md 4.1.
static int rgat_hub_min() {
  static const int v = [] {
    const char* e = getenv("HET_RGAT_HUB_MIN");
    // (256 in round 3, 128 once el came from the row -- the hub launch gained more from that than the pack-form one --, 256 again in
    //  round 5: the pack-form launch got faster (tags, 32-bit offsets) and the hub finish is a workgroup per hub;
    //  profiles/r05/ab_round5_misc.txt: 64 / 96 / 128 / 192 / 256 -> 4.03 / 3.86 / 3.78 / 3.74 / 3.72 ms per step)
    const int t = e ? atoi(e) : 256;
    return t < HET_PACK_T ? HET_PACK_T : t;
  }();
  return v;
}

// Short / long threshold of the backward's (relation, source) segments: up to this many edges a segment is walked by one lane group
// inside a pack, longer ones are wave-per-item work.  64 instead of the library-wide HET_PACK_T = 32 (exp: the whole library built
// with 16 / 32 / 64: RGAT 3.98 / 3.97 / 3.79 ms per step -- the backward op 1.57 -> 1.44 -- while RGCN's and HGT's segment sums
// lose 1-2 % at 64, so only this grouping asks for it).
static int rgat_bwd_pack_t() {
  static const int v = [] {
    const char* e = getenv("HET_RGAT_BWD_PACK_T");
    const int t = e ? atoi(e) : 64;
    return t < 8 ? 8 : (t > 256 ? 256 : t);
  }();
  return v;
}

static bool hub_in_row_order() {
  static const bool on = [] { const char* e = getenv("HET_RGAT_HUB_ORDER"); return !(e && e[0] == '0'); }();  // A/B switch
  return on;
}

extern "C" int64_t het_rgat_aggregate_compact_runs_workspace(const het_grouping* by_dst, const het_grouping* by_dst_rel,
                                                             int64_t num_rels, int64_t H, int64_t D, het_stream stream) {
  if (!by_dst || !by_dst_rel || num_rels <= 0) return -1;
  if (grouping_hub_items(by_dst_rel, by_dst, (int)num_rels, rgat_hub_min(), (hipStream_t)stream) != HET_OK) return -1;  // (built on first use)
  return (int64_t)sizeof(float) * by_dst_rel->num_hub_items * (2 * H * D + 3 * H);
}

extern "C" int het_rgat_aggregate_compact_runs(const het_grouping* by_dst, const het_grouping* by_dst_rel, int64_t num_rels,
                                               const float* feat_c, const float* el_c, const float* er_c, float* sum, float* ret,
                                               int64_t num_nodes, int64_t H, int64_t D, double slope, float* h_inout,
                                               int64_t h_rows, float* q_rows, float* q_sum, float* q_ref, int64_t num_dst_rows,
                                               const float* attn_l, const int64_t* feat_rel_ptrs_host, void* workspace,
                                               int64_t workspace_bytes, het_stream stream) {
  const char* op = "het_rgat_aggregate_compact_runs";
  hipStream_t s = (hipStream_t)stream;
  HET_REQUIRE(by_dst && by_dst_rel && sum && ret && q_rows && q_sum && q_ref && num_nodes >= 0 && num_rels > 0, "%s: null argument", op);
  if (!compact_shape_ok(H, D) || !coop_shape_ok(H, D)) {
    het_set_error("%s: unsupported shape H=%lld D=%lld", op, (long long)H, (long long)D);
    return HET_ERR_UNSUPPORTED;
  }
  HET_REQUIRE(by_dst->R == 0 && by_dst->key_bound <= num_nodes && (by_dst->E == 0 || (by_dst->p0 && by_dst->p1 && feat_c && el_c && er_c)),
              "%s: by_dst must group the positions by destination with payload0 = feat row and payload1 = er row", op);
  HET_REQUIRE(by_dst_rel->R == 0 && by_dst_rel->E == by_dst->E && by_dst_rel->key_bound <= num_nodes * num_rels,
              "%s: by_dst_rel must group the same positions by destination * num_rels + relation", op);
  const int64_t need = het_rgat_aggregate_compact_runs_workspace(by_dst, by_dst_rel, num_rels, H, D, stream);
  if (need < 0) return HET_ERR_INVALID_ARG;
  HET_REQUIRE(need == 0 || (workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0),
              "%s: a 16-byte aligned workspace of %lld bytes is needed (het_rgat_aggregate_compact_runs_workspace)", op, (long long)need);
  const int64_t X = H * D;
  HET_HIP(hipMemsetAsync(sum, 0, sizeof(float) * num_nodes * H, s));
  HET_HIP(hipMemsetAsync(q_sum, 0, sizeof(float) * num_dst_rows * H, s));  // er rows without edges: q == 0 marks them
  if (!h_inout) HET_HIP(hipMemsetAsync(ret, 0, sizeof(float) * num_nodes * X, s));
  if (by_dst->E == 0) return HET_OK;
  if (int rc = grouping_packed_ids(by_dst, true, s)) return rc;  // (builds the packs too)
  if (by_dst_rel->num_hub_items > 0)
    if (int rc = grouping_packed_ids(by_dst, false, s)) return rc;
  float* part = static_cast<float*>(workspace);
  // el from the gathered row (ElFold above): heads of 16 floats, up to 8 relations, the caller names the relation boundaries of the
  // feat rows (host array [R+1]) and attn_l [R, H*D]; otherwise (or HET_RGAT_EL_FROM_ROW=0) el_c is gathered per edge
  static const bool el_from_row = [] { const char* v = getenv("HET_RGAT_EL_FROM_ROW"); return !(v && v[0] == '0'); }();
  ElFold ef{};
  const bool elr = el_from_row && attn_l && feat_rel_ptrs_host && D == 16 && num_rels <= kElMaxRels &&
                   (reinterpret_cast<uintptr_t>(attn_l) & 15) == 0;
  if (elr) {
    ef.attn = attn_l; ef.R = (int)num_rels;
    for (int k = 0; k < kElMaxRels - 1; ++k)
      ef.thr[k] = k + 1 < num_rels && feat_rel_ptrs_host[k + 1] < 0x7fffffffll ? (int)feat_rel_ptrs_host[k + 1] : 0x7fffffff;
  }
  // packed id records tagged with the relation of the feat row and the run / destination ends (grouping_tag_kp01)
  if (int rc = grouping_tag_kp01(by_dst, 1, elr ? ef.thr : nullptr, s)) return rc;
  // 32-bit byte offsets when every table the row kernels index stays below 4 GiB (HET_RGAT_WIDE_OFFSETS=1: always 64-bit, A/B)
  static const bool wide_env = [] { const char* v = getenv("HET_RGAT_WIDE_OFFSETS"); return v && v[0] == '1'; }();
  // (destinations: ret / h_inout / lse rows by key; feat rows by payload0; er / q rows by payload1)
  const bool w64 = wide_env || !(het_fits_u32(by_dst->key_bound, X * 4) && het_fits_u32(by_dst->p0_max + 1, X * 4) &&
                                 het_fits_u32(by_dst->p1_max + 1, X * 4));
  const int hio_rows32 = (int)(h_rows < 0 ? 0 : (h_rows > 0x7fffffffll ? 0x7fffffffll : h_rows));
  HetFork fk(s);  // the hub launches beside the pack-form one: disjoint destinations, both bound by gather latency
  {
    HET_KTIME("HET_rgat_aggregate_packs", s);
    Packs pk{by_dst->pack_ptr, by_dst->key_of_rank, by_dst->num_packs};
    const unsigned nb = (unsigned)ceil_div64(by_dst->num_packs, (int64_t)(kBlock / 64) * (64 / (X / 4)));
#define HET_PACKS_LAUNCH2(ELRV, WV)                                                                                                \
  hipLaunchKernelGGL((HET_rgat_aggregate_runs_packed<LPR, DL, ELRV, WV>), dim3(nb), dim3(kBlock), 0, s, pk, by_dst->kp01, feat_c,   \
                     el_c, er_c, sum, ret, (float)slope, h_inout, hio_rows32, q_rows, q_sum, q_ref, rgat_hub_min(), ef)
#define HET_PACKS_LAUNCH(ELRV) do { if (w64) { HET_PACKS_LAUNCH2(ELRV, true); } else { HET_PACKS_LAUNCH2(ELRV, false); } } while (0)
    if (elr) {
      switch (X / 4) {
        case 8: { constexpr int LPR = 8, DL = 4; HET_PACKS_LAUNCH(true); break; }
        case 16: { constexpr int LPR = 16, DL = 4; HET_PACKS_LAUNCH(true); break; }
        default: { constexpr int LPR = 32, DL = 4; HET_PACKS_LAUNCH(true); break; }
      }
    } else {
      HET_DISPATCH_COOP((int)(X / 4), (int)(D / 4), HET_PACKS_LAUNCH(false));
    }
#undef HET_PACKS_LAUNCH
#undef HET_PACKS_LAUNCH2
  }
  HET_LAUNCH_CHECK("HET_rgat_aggregate_runs_packed");
  if (by_dst_rel->num_hub_items > 0) {
    hipStream_t s2 = fk.side;
    Items it{by_dst_rel->item_seg, by_dst_rel->item_begin, by_dst_rel->item_end, by_dst_rel->seg_ptr, by_dst_rel->seg_key,
             by_dst_rel->num_items};
    const int64_t n_hub = by_dst_rel->num_hub_items;
    {
      HET_KTIME("HET_rgat_aggregate_hubs", s2);
      const unsigned nbh = (unsigned)ceil_div64(n_hub, kBlock / 64);
#define HET_HUBS_LAUNCH2(ELRV, WV)                                                                                              \
  hipLaunchKernelGGL((HET_rgat_aggregate_hub_items<LPR, DL, ELRV, WV>), dim3(nbh), dim3(kBlock), 0, s2, it, by_dst_rel->hub_items,  \
                     hub_in_row_order() ? by_dst_rel->hub_order : nullptr, n_hub, by_dst->p01, feat_c, el_c, er_c,                 \
                     (float)slope, part, ef)
#define HET_HUBS_LAUNCH(ELRV) do { if (w64) { HET_HUBS_LAUNCH2(ELRV, true); } else { HET_HUBS_LAUNCH2(ELRV, false); } } while (0)
      if (elr) {
        switch (X / 4) {
          case 8: { constexpr int LPR = 8, DL = 4; HET_HUBS_LAUNCH(true); break; }
          case 16: { constexpr int LPR = 16, DL = 4; HET_HUBS_LAUNCH(true); break; }
          default: { constexpr int LPR = 32, DL = 4; HET_HUBS_LAUNCH(true); break; }
        }
      } else {
        HET_DISPATCH_COOP((int)(X / 4), (int)(D / 4), HET_HUBS_LAUNCH(false));
      }
#undef HET_HUBS_LAUNCH
#undef HET_HUBS_LAUNCH2
    }
    HET_LAUNCH_CHECK("HET_rgat_aggregate_hub_items");
    const unsigned nbs = (unsigned)by_dst_rel->num_hub_segs;  // (a workgroup per hub)
    {
    HET_KTIME("HET_rgat_aggregate_finish", s2);
    HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL(HET_rgat_finish_hubs<LPR>, dim3(nbs), dim3(kBlock), 0, s2, by_dst_rel->hub_rec,
                                                      by_dst_rel->num_hub_segs, it, by_dst->p1, part, sum, ret, (int)H, (int)D,
                                                      h_inout, h_rows, q_rows, q_sum, q_ref));
    }
    HET_LAUNCH_CHECK("HET_rgat_finish_hubs");
  }
  HET_HIP(fk.join());
  return HET_OK;
}

extern "C" int64_t het_rgat_backward_compact_workspace(int64_t num_nodes, int64_t num_edges, int64_t H, int64_t D, int with_bias) {
  const int64_t n_pack = (num_nodes * 2 * H + 3) / 4 * 4, n_tbuf = (num_edges * H + 3) / 4 * 4;
  return (int64_t)sizeof(float) * (n_pack + n_tbuf + (with_bias ? (int64_t)2048 * (kBlock / 64) * H * D : 0));
}

// runs (q_rows != NULL): grad_er from the run sums the forward left (no per-edge term, no by_drow); else from tbuf + by_drow
struct RunSums {
  const float *q_rows, *q_sum, *q_ref;
  const int64_t* drow_nodes;
};
// rows of partial grad_attn_l sums the two source-row launches write (one per workgroup): from the backward's packs of by_srow
static int64_t attn_grad_partial_rows(const PackView& pv, int64_t X) {
  const int64_t nb = ceil_div64(pv.num_packs, (int64_t)(kBlock / 64) * (64 / (X / 4)));
  const int64_t nbl = ceil_div64(pv.num_long_items, kBlock / 64);
  return nb + nbl;
}

static int rgat_backward_compact_impl(const char* op, const het_grouping* by_srow, const het_grouping* by_drow, const RunSums* runs,
                                      float* grad_attn_l, const float* feat_c, const float* el_c, const float* er_c, const float* sum, const float* ret,
                                      const float* gradout, float* grad_feat_c, float* grad_el_c, float* grad_er_c,
                                      const float* fold_attn_l, const int64_t* row_rel_ptrs, int64_t num_rels,
                                      float* grad_bias, int64_t bias_rows, int64_t num_nodes, int64_t num_src_rows,
                                      int64_t num_dst_rows, int64_t H, int64_t D, double slope, void* workspace,
                                      int64_t workspace_bytes, het_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  HET_REQUIRE(by_srow && (by_drow || runs) && sum && ret && gradout && grad_feat_c && grad_er_c, "%s: null argument", op);
  // grad_el_c may be NULL when nobody reads it: its two consumers -- the gradient through el = <feat, attn_l> (fold_attn_l) and the
  // weight gradient of attn_l (grad_attn_l) -- are both formed inside the pass (cooperative shapes only)
  HET_REQUIRE(grad_el_c || (runs && fold_attn_l && grad_attn_l && coop_shape_ok(H, D)),
              "%s: grad_el_c may only be NULL in the run-sum form with fold_attn_l and grad_attn_l (cooperative shapes)", op);
  if (!compact_shape_ok(H, D) || !segment_rows_supported((int)H) || slope < 0 || (runs && !coop_shape_ok(H, D))) {
    het_set_error("%s: unsupported shape H=%lld D=%lld (or slope < 0)", op, (long long)H, (long long)D);
    return HET_ERR_UNSUPPORTED;
  }
  const int64_t E = by_srow->E, X = H * D;
  HET_REQUIRE(by_srow->R == 0 && by_srow->key_bound <= num_src_rows && (E == 0 || (by_srow->p0 && by_srow->p1)),
              "%s: by_srow groups the positions by feat row (payload0 = destination, payload1 = er row)", op);
  HET_REQUIRE(runs || (by_drow->R == 0 && by_drow->E == E && by_drow->S == num_dst_rows && (E == 0 || by_drow->p0)),
              "%s: by_drow groups the positions by er row with payload0 = their rank in by_srow and has one segment per er row", op);
  HET_REQUIRE(!fold_attn_l || (row_rel_ptrs && num_rels > 0), "%s: fold_attn_l needs the relation pointers of the feat rows", op);
  const bool coop = coop_shape_ok(H, D);
  HET_REQUIRE(!grad_attn_l || (coop && fold_attn_l && num_rels <= 8), "%s: grad_attn_l needs the cooperative shapes, fold_attn_l and <= 8 relations", op);
  // the backward's own packs of by_srow (threshold rgat_bwd_pack_t(): a second set beside the library-wide one, grouping.hip.h)
  PackView pv;
  if (int rc = grouping_pack_view(by_srow, s, rgat_bwd_pack_t(), &pv)) return rc;
  constexpr int kBiasBlocks = 2048;
  const int64_t bias_part_rows = grad_bias ? (int64_t)kBiasBlocks * (kBlock / 64) : 0;
  const int64_t ga_rows = (grad_attn_l && E > 0) ? attn_grad_partial_rows(pv, X) : 0, n_ga = (ga_rows * (X + 1) + 3) / 4 * 4;
  const int64_t n_pack = (num_nodes * 2 * H + 3) / 4 * 4, n_tbuf = runs ? 0 : (E * H + 3) / 4 * 4;  // 16-byte aligned pieces
  static const bool rec_on = [] { const char* v = getenv("HET_RGAT_DROW_REC"); return !(v && v[0] == '0'); }();  // A/B switch
  const bool use_rec = runs && coop && rec_on && E > 0 && num_dst_rows > 0;
  const int64_t n_rec = (runs && coop) ? num_dst_rows * H * 4 : 0;
  const int64_t need = (int64_t)sizeof(float) * (n_pack + n_tbuf + bias_part_rows * X + n_ga + n_rec);
  HET_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "%s: a 16-byte aligned workspace of %lld bytes is needed (het_rgat_backward_compact_workspace)", op, (long long)need);
  float* pack = (float*)workspace;  // [N, 2H]
  float* tbuf = runs ? nullptr : pack + n_pack;  // [E, H], rank order of by_srow
  float* bias_part = grad_bias ? pack + n_pack + n_tbuf : nullptr;
  float* ga_part = ga_rows ? pack + n_pack + n_tbuf + bias_part_rows * X : nullptr;  // [ga_rows, X] then [ga_rows] relation tags
  int* ga_rel = ga_rows ? reinterpret_cast<int*>(ga_part + ga_rows * X) : nullptr;
  float* rec4 = use_rec ? pack + n_pack + n_tbuf + bias_part_rows * X + n_ga : nullptr;  // [S_col, H, 4]
  if (grad_attn_l) HET_HIP(hipMemsetAsync(grad_attn_l, 0, sizeof(float) * num_rels * X, s));  // (boundary pieces add atomically)
  if (by_srow->S != num_src_rows) {  // feat rows without an edge (none when the lists come from the graph): zero gradient
    HET_HIP(hipMemsetAsync(grad_feat_c, 0, sizeof(float) * num_src_rows * X, s));
    if (grad_el_c) HET_HIP(hipMemsetAsync(grad_el_c, 0, sizeof(float) * num_src_rows * H, s));
  }
  // One pass over the er rows (HET_rgat_drow_pass: the records of the source-row kernels AND grad_er) instead of dst pack + record
  // pack + grad_er pass; the bias gradient's column sums then are a pass of their own on the side stream (HET_RGAT_DROW_PASS=0: A/B)
  static const bool drow_pass_on = [] { const char* v = getenv("HET_RGAT_DROW_PASS"); return !(v && v[0] == '0'); }();
  const bool fused_drow = use_rec && drow_pass_on;
  if (num_nodes > 0 && !fused_drow) {
    const unsigned nbp = grad_bias ? kBiasBlocks : grid_for(num_nodes * (X / 4));
    {
      HET_KTIME("HET_rgat_backward_dst_pack", s);
      HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL(HET_rgat_dst_pack<LPR>, dim3(nbp), dim3(kBlock), 0, s, sum, ret, gradout,
                                                        pack, num_nodes, (int)H, (int)D, bias_part, bias_rows, (int)coop));
    }
    HET_LAUNCH_CHECK("HET_rgat_dst_pack");
  } else if (grad_bias && num_nodes <= 0) {
    HET_HIP(hipMemsetAsync(grad_bias, 0, sizeof(float) * X, s));
  }
  if (E == 0) {
    if (grad_bias && num_nodes > 0) {
      if (fused_drow) {
        HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL(HET_rgat_colsum_rows<LPR>, dim3(kBiasBlocks), dim3(kBlock), 0, s, gradout,
                                                          bias_rows < num_nodes ? bias_rows : num_nodes, bias_part));
        HET_LAUNCH_CHECK("HET_rgat_colsum_rows");
      }
      hipLaunchKernelGGL(HET_rgat_colsum_finish, dim3((unsigned)X), dim3(kBlock), 0, s, bias_part, bias_part_rows, (int)X, grad_bias);
      HET_LAUNCH_CHECK("HET_rgat_colsum_finish");
    }
    HET_HIP(hipMemsetAsync(grad_er_c, 0, sizeof(float) * num_dst_rows * H, s));
    return HET_OK;
  }
  Packs pk{pv.pack_ptr, by_srow->key_of_rank, pv.num_packs};
  const unsigned nb = (unsigned)ceil_div64(pv.num_packs, (int64_t)(kBlock / 64) * (64 / (X / 4)));
  if (fused_drow) {  // before the fork: both source-row launches read the records
    // (the rows in the order of their destination nodes; the order is kept with by_srow, the one grouping this path always has.
    //  HET_RGAT_DROW_ORDER=0: A/B)
    static const bool drow_order_on = [] { const char* v = getenv("HET_RGAT_DROW_ORDER"); return !(v && v[0] == '0'); }();
    const int32_t* drow_order = nullptr;
    if (drow_order_on && num_dst_rows > 0) {
      if (int rc = grouping_value_order(by_srow, runs->drow_nodes, num_dst_rows, s)) return rc;
      drow_order = by_srow->val_order;
    }
    HET_KTIME("HET_rgat_backward_drow_pass", s);
    const unsigned nbd = (unsigned)ceil_div64(num_dst_rows, (int64_t)(kBlock / 64) * (64 / (X / 4)) * 2);
    HET_DISPATCH_COOP((int)(X / 4), (int)(D / 4),
                      hipLaunchKernelGGL((HET_rgat_drow_pass<LPR, DL>), dim3(nbd), dim3(kBlock), 0, s, er_c, sum, ret, gradout,
                                         runs->q_rows, runs->q_sum, runs->q_ref, runs->drow_nodes, drow_order, num_dst_rows, rec4,
                                         grad_er_c));
    HET_LAUNCH_CHECK("HET_rgat_drow_pass");
  } else if (use_rec) {
    hipLaunchKernelGGL(HET_rgat_drow_rec, dim3(grid_for(num_dst_rows * H)), dim3(kBlock), 0, s, er_c, pack, runs->drow_nodes, num_dst_rows,
                       (int)H, rec4);
    HET_LAUNCH_CHECK("HET_rgat_drow_rec");
  }
  const float* er_arg = use_rec ? rec4 : er_c;
  bool w64 = true;
  if (coop) {
    if (int rc = grouping_packed_ids(by_srow, true, s)) return rc;
    if (pv.num_long_items > 0)
      if (int rc = grouping_packed_ids(by_srow, false, s)) return rc;
    // tags of the packed id records: segment ends + the relation of the feat row (read from the caller's device array; a grouping
    // keeps the tags of the array it saw first -- the relation boundaries of a row list belong to the list)
    if (fold_attn_l) {
      if (int rc = grouping_tag_kp01_dev(by_srow, 0, row_rel_ptrs, (int)num_rels, s)) return rc;
    } else {
      if (int rc = grouping_tag_kp01(by_srow, 0, nullptr, s)) return rc;
    }
    static const bool wide_env = [] { const char* v = getenv("HET_RGAT_WIDE_OFFSETS"); return v && v[0] == '1'; }();
    // (feat / grad rows by key; gradout and pack2 rows by payload0 = destination; er / record rows by payload1)
    w64 = wide_env || !(het_fits_u32(num_src_rows, X * 4) && het_fits_u32(by_srow->p0_max + 1, X * 4) &&
                        het_fits_u32(by_srow->p1_max + 1, H * 16) && het_fits_u32(by_srow->p1_max + 1, X * 4));
  }
  // Two chains after the per-destination pack, joined at the end (kernel trace on ogbn-mag: the short-segment launch 1.10 ms beside
  // the long-segment one 1.09 ms; with the small launches in front of the short one the op was 0.1 ms longer):
  //   caller's stream: short segments, grad_er from the run sums
  //   side stream:     bias column sums (needed by nobody here), zeroed rows of the split segments, long segments
  HetFork fk(s);
  hipStream_t s2 = fk.side;
  if (grad_bias && num_nodes > 0) {
    if (fused_drow) {
      HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL(HET_rgat_colsum_rows<LPR>, dim3(kBiasBlocks), dim3(kBlock), 0, s2, gradout,
                                                        bias_rows < num_nodes ? bias_rows : num_nodes, bias_part));
      HET_LAUNCH_CHECK("HET_rgat_colsum_rows");
    }
    hipLaunchKernelGGL(HET_rgat_colsum_finish, dim3((unsigned)X), dim3(kBlock), 0, s2, bias_part, bias_part_rows, (int)X, grad_bias);
    HET_LAUNCH_CHECK("HET_rgat_colsum_finish");
  }
  if (by_srow->num_split > 0) {  // segments of several work items (> HET_ITEM_MAX edges): their items add atomically
    HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL(HET_rgat_zero_long_rows<LPR>, dim3(grid_for(by_srow->num_split * (X / 4))),
                                                      dim3(kBlock), 0, s2, by_srow->split_seg, by_srow->seg_key, by_srow->num_split,
                                                      grad_feat_c, grad_el_c, (int)H));
    HET_LAUNCH_CHECK("HET_rgat_zero_long_rows");
  }
  if (coop) {
    {
      HET_KTIME("HET_rgat_backward_src_short", s);
#define HET_SRC_COOP2(GA_, REC_, W_, gp_, gr_, go_)                                                                              \
  HET_DISPATCH_COOP((int)(X / 4), (int)(D / 4),                                                                                 \
                    hipLaunchKernelGGL((HET_rgat_backward_src_coop<LPR, DL, GA_, REC_, W_>), dim3(nb), dim3(kBlock), 0, s, pk,   \
                                       by_srow->kp01, feat_c, el_c, er_arg, pack, gradout, grad_feat_c, grad_el_c, tbuf,         \
                                       (float)slope, fold_attn_l, row_rel_ptrs, (int)num_rels, gp_, gr_, go_))
#define HET_SRC_COOP(GA_, REC_, gp_, gr_, go_) \
  do { if (w64) { HET_SRC_COOP2(GA_, REC_, true, gp_, gr_, go_); } else { HET_SRC_COOP2(GA_, REC_, false, gp_, gr_, go_); } } while (0)
      if (ga_rows) {
        if (use_rec) { HET_SRC_COOP(true, true, ga_part, ga_rel, grad_attn_l); } else { HET_SRC_COOP(true, false, ga_part, ga_rel, grad_attn_l); }
      } else {
        if (use_rec) { HET_SRC_COOP(false, true, nullptr, nullptr, nullptr); } else { HET_SRC_COOP(false, false, nullptr, nullptr, nullptr); }
      }
#undef HET_SRC_COOP
#undef HET_SRC_COOP2
    }
    HET_LAUNCH_CHECK("HET_rgat_backward_src_coop");
    if (pv.num_long_items > 0) {
      Items it{by_srow->item_seg, by_srow->item_begin, by_srow->item_end, by_srow->seg_ptr, by_srow->seg_key, by_srow->num_items};
      const unsigned nbl = (unsigned)ceil_div64(pv.num_long_items, kBlock / 64);
      int* ga_rel_long = ga_rel ? ga_rel + nb : nullptr;  // (the long launch's workgroups follow the short launch's in the partial rows)
      HET_KTIME("HET_rgat_backward_src_long", s2);
#define HET_SRC_LONG2(GA_, REC_, W_, gp_, gr_, go_)                                                                              \
  HET_DISPATCH_COOP((int)(X / 4), (int)(D / 4),                                                                                 \
                    hipLaunchKernelGGL((HET_rgat_backward_src_long<LPR, DL, GA_, REC_, W_>), dim3(nbl), dim3(kBlock), 0, s2, it, \
                                       pv.long_items, pv.num_long_items, by_srow->p01, feat_c, el_c, er_arg, pack,   \
                                       gradout, grad_feat_c, grad_el_c, tbuf, (float)slope, fold_attn_l, row_rel_ptrs,           \
                                       (int)num_rels, gp_, gr_, go_))
#define HET_SRC_LONG(GA_, REC_, gp_, gr_, go_) \
  do { if (w64) { HET_SRC_LONG2(GA_, REC_, true, gp_, gr_, go_); } else { HET_SRC_LONG2(GA_, REC_, false, gp_, gr_, go_); } } while (0)
      if (ga_rows) {
        if (use_rec) { HET_SRC_LONG(true, true, ga_part + (int64_t)nb * X, ga_rel_long, grad_attn_l); }
        else { HET_SRC_LONG(true, false, ga_part + (int64_t)nb * X, ga_rel_long, grad_attn_l); }
      } else {
        if (use_rec) { HET_SRC_LONG(false, true, nullptr, nullptr, nullptr); } else { HET_SRC_LONG(false, false, nullptr, nullptr, nullptr); }
      }
#undef HET_SRC_LONG
#undef HET_SRC_LONG2
    }
  } else {
    static const int u_rows = [] { const char* v = getenv("HET_RGAT_BWD_U"); return v ? atoi(v) : 4; }();  // A/B switch
    const int skip_long = pv.num_long_items > 0 ? 1 : 0;
    {
      HET_KTIME("HET_rgat_backward_src_short", s);
#define HET_BWD_PACKED(UU)                                                                                                 \
  HET_DISPATCH_LPR((int)(X / 4),                                                                                           \
                   hipLaunchKernelGGL((HET_rgat_backward_src_packed<LPR, UU>), dim3(nb), dim3(kBlock), 0, s, pk, by_srow->p0, \
                                      by_srow->p1, feat_c, el_c, er_c, pack, gradout, grad_feat_c, grad_el_c, tbuf, (int)H, \
                                      (int)D, (float)slope, fold_attn_l, row_rel_ptrs, (int)num_rels, skip_long))
      if (u_rows == 2) { HET_BWD_PACKED(2); } else if (u_rows == 8) { HET_BWD_PACKED(8); } else { HET_BWD_PACKED(4); }
#undef HET_BWD_PACKED
    }
    if (skip_long) {
      HET_LAUNCH_CHECK("HET_rgat_backward_src_packed");
      Items it{by_srow->item_seg, by_srow->item_begin, by_srow->item_end, by_srow->seg_ptr, by_srow->seg_key, by_srow->num_items};
      const unsigned nbl = (unsigned)ceil_div64(pv.num_long_items, kBlock / 64);
      HET_KTIME("HET_rgat_backward_src_long", s2);
      HET_DISPATCH_LPR((int)(X / 4),
                       hipLaunchKernelGGL(HET_rgat_backward_src_long_any<LPR>, dim3(nbl), dim3(kBlock), 0, s2, it, pv.long_items,
                                          pv.num_long_items, by_srow->p0, by_srow->p1, feat_c, el_c, er_c, pack, gradout,
                                          grad_feat_c, grad_el_c, tbuf, (int)H, (int)D, (float)slope, fold_attn_l, row_rel_ptrs,
                                          (int)num_rels));
    }
  }
  HET_LAUNCH_CHECK("HET_rgat_backward_src_packed");
  if (runs) {
    if (num_dst_rows > 0 && !fused_drow) {
      HET_KTIME("HET_rgat_backward_er_runs", s);
      HET_DISPATCH_LPR((int)(X / 4), hipLaunchKernelGGL(HET_rgat_grad_er_runs<LPR>, dim3(grid_for(num_dst_rows * (X / 4))), dim3(kBlock),
                                                        0, s, runs->q_rows, runs->q_sum, runs->q_ref, runs->drow_nodes, pack, gradout,
                                                        grad_er_c, num_dst_rows, (int)H, (int)D));
    }
    HET_LAUNCH_CHECK("HET_rgat_grad_er_runs");
    HET_HIP(fk.join());
    if (ga_rows) {
      hipLaunchKernelGGL(HET_rgat_attn_grad_finish, dim3((unsigned)ceil_div64(ga_rows, kAttnFinishRows)), dim3(kBlock), 0, s, ga_part, ga_rel, ga_rows,
                         (int)X, grad_attn_l);
      HET_LAUNCH_CHECK("HET_rgat_attn_grad_finish");
    }
    return HET_OK;
  }
  HET_HIP(fk.join());
  if (ga_rows) {
    hipLaunchKernelGGL(HET_rgat_attn_grad_finish, dim3((unsigned)ceil_div64(ga_rows, kAttnFinishRows)), dim3(kBlock), 0, s, ga_part, ga_rel, ga_rows,
                       (int)X, grad_attn_l);
    HET_LAUNCH_CHECK("HET_rgat_attn_grad_finish");
  }
  // grad_er[w, :] = SUM over the edges of er row w of tbuf[rank, :]   (segments of by_drow are the er rows in order)
  return launch_segment_sum(by_drow, tbuf, grad_er_c, (int)H, nullptr, s);
}

extern "C" int het_rgat_backward_compact(const het_grouping* by_srow, const het_grouping* by_drow, const float* feat_c,
                                         const float* el_c, const float* er_c, const float* sum, const float* ret,
                                         const float* gradout, float* grad_feat_c, float* grad_el_c, float* grad_er_c,
                                         const float* fold_attn_l, const int64_t* row_rel_ptrs, int64_t num_rels,
                                         float* grad_bias, int64_t bias_rows, int64_t num_nodes, int64_t num_src_rows,
                                         int64_t num_dst_rows, int64_t H, int64_t D, double slope, void* workspace,
                                         int64_t workspace_bytes, het_stream stream) {
  HET_REQUIRE(by_drow, "het_rgat_backward_compact: null argument");
  return rgat_backward_compact_impl("het_rgat_backward_compact", by_srow, by_drow, nullptr, nullptr, feat_c, el_c, er_c, sum, ret, gradout,
                                    grad_feat_c, grad_el_c, grad_er_c, fold_attn_l, row_rel_ptrs, num_rels, grad_bias, bias_rows,
                                    num_nodes, num_src_rows, num_dst_rows, H, D, slope, workspace, workspace_bytes, stream);
}

// The backward after het_rgat_aggregate_compact_runs: q_rows / q_sum / q_ref as that call left them, drow_nodes [num_dst_rows] the
// destination node of every er row.  Workspace: het_rgat_backward_compact_workspace with num_edges = 0.
// bytes of het_rgat_backward_compact_runs' workspace (builds the packs of by_srow on `stream` the first time when the attention
// gradient is asked for: its partial rows are counted in workgroups); -1 on error
extern "C" int64_t het_rgat_backward_compact_runs_workspace(const het_grouping* by_srow, int64_t num_nodes, int64_t num_dst_rows,
                                                            int64_t H, int64_t D, int with_bias, int with_attn_grad,
                                                            het_stream stream) {
  if (!by_srow || num_dst_rows < 0) return -1;
  int64_t bytes = het_rgat_backward_compact_workspace(num_nodes, 0, H, D, with_bias);
  if (coop_shape_ok(H, D)) bytes += (int64_t)sizeof(float) * num_dst_rows * H * 4;  // the per-(er row, head) records
  if (with_attn_grad && by_srow->E > 0) {
    PackView pv;
    if (grouping_pack_view(by_srow, (hipStream_t)stream, rgat_bwd_pack_t(), &pv) != HET_OK) return -1;
    bytes += (int64_t)sizeof(float) * ((attn_grad_partial_rows(pv, H * D) * (H * D + 1) + 3) / 4 * 4);
  }
  return bytes;
}

extern "C" int het_rgat_backward_compact_runs(const het_grouping* by_srow, const float* q_rows, const float* q_sum, const float* q_ref,
                                              const int64_t* drow_nodes, const float* feat_c, const float* el_c, const float* er_c,
                                              const float* sum, const float* ret, const float* gradout, float* grad_feat_c,
                                              float* grad_el_c, float* grad_er_c, const float* fold_attn_l,
                                              const int64_t* row_rel_ptrs, int64_t num_rels, float* grad_bias, int64_t bias_rows,
                                              int64_t num_nodes, int64_t num_src_rows, int64_t num_dst_rows, int64_t H, int64_t D,
                                              double slope, float* grad_attn_l, void* workspace, int64_t workspace_bytes,
                                              het_stream stream) {
  HET_REQUIRE(q_rows && q_sum && q_ref && (drow_nodes || num_dst_rows == 0), "het_rgat_backward_compact_runs: null argument");
  const RunSums runs{q_rows, q_sum, q_ref, drow_nodes};
  return rgat_backward_compact_impl("het_rgat_backward_compact_runs", by_srow, nullptr, &runs, grad_attn_l, feat_c, el_c, er_c, sum, ret, gradout,
                                    grad_feat_c, grad_el_c, grad_er_c, fold_attn_l, row_rel_ptrs, num_rels, grad_bias, bias_rows,
                                    num_nodes, num_src_rows, num_dst_rows, H, D, slope, workspace, workspace_bytes, stream);
}
