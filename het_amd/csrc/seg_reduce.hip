// Segmented sum of gathered rows, one lane group per work item (no atomics except for split hub segments).
#include <stdlib.h>

#include <mutex>

#include "seg_reduce.hip.h"

namespace {

constexpr int kBlock = 256;
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// Slot-per-item: the LPR lanes of a slot own one work item each, so a wave runs 64/LPR items side by side.
// Short segments (a few rows) then keep every lane group busy, the per-item prologue (item record -> ids -> rows) is
// shared by 64/LPR items, and the row ids of the next step are fetched while the current rows are in flight, which
// leaves one dependent round trip per step.  No cross-slot reduction is needed.
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_segment_sum(const int32_t* __restrict__ item_seg,
                                                                 const int32_t* __restrict__ item_begin,
                                                                 const int32_t* __restrict__ item_end,
                                                                 const int32_t* __restrict__ seg_ptr, int64_t num_items,
                                                                 const int32_t* __restrict__ p_row,
                                                                 const int32_t* __restrict__ p_scale,
                                                                 const float* __restrict__ scale, int scale_heads,
                                                                 const float* __restrict__ in, float* __restrict__ out,
                                                                 const int32_t* __restrict__ out_row, int accumulate,
                                                                 int nt_in, int contig) {
  constexpr int EPW = 64 / LPR, X = LPR * 4, U = 4;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, x = (lane % LPR) * 4;
  const int64_t item = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * EPW + slot;
  if (item >= num_items) return;
  const int seg = item_seg[item], b = item_begin[item], e = item_end[item];
  // scale_heads == X: one scale per element (a float4 per lane); else one per (row, head) or per row
  const bool ew = scale_heads == X;
  const int sld = scale_heads ? scale_heads : 1, sh = ew ? x : (scale_heads ? x / (X / scale_heads) : 0);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int jn[U];
  int64_t rown[U];
  int sin[U];
#pragma unroll
  for (int u = 0; u < U; ++u) jn[u] = b + u < e ? b + u : e - 1;
  // contig: the rows of a segment are consecutive rows of `in` (the list was already sorted): no id loads
#pragma unroll
  for (int u = 0; u < U; ++u) rown[u] = contig ? jn[u] : p_row[jn[u]];
  if (scale) {
#pragma unroll
    for (int u = 0; u < U; ++u) sin[u] = p_scale ? p_scale[jn[u]] : jn[u];
  }
  for (int j0 = b; j0 < e; j0 += U) {
    float4 w[U];
    float4 f[U];
    if (scale && ew) {
#pragma unroll
      for (int u = 0; u < U; ++u) w[u] = ld4(scale + (int64_t)sin[u] * sld + sh);
    } else if (scale) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float t = scale[(int64_t)sin[u] * sld + sh];
        w[u] = make_float4(t, t, t, t);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) w[u] = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    if (nt_in) {  // `in` is a once-read [E, X] stream
#pragma unroll
      for (int u = 0; u < U; ++u) f[u] = ld4_nt(in + rown[u] * X + x);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) f[u] = ld4(in + rown[u] * X + x);
    }
    // ids of the next step (clamped: the last step re-reads its own ids)
#pragma unroll
    for (int u = 0; u < U; ++u) jn[u] = j0 + U + u < e ? j0 + U + u : e - 1;
#pragma unroll
    for (int u = 0; u < U; ++u) rown[u] = contig ? jn[u] : p_row[jn[u]];
    if (scale) {
#pragma unroll
      for (int u = 0; u < U; ++u) sin[u] = p_scale ? p_scale[jn[u]] : jn[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float ok = (j0 + u < e) ? 1.f : 0.f;
      acc.x = fmaf(ok * w[u].x, f[u].x, acc.x); acc.y = fmaf(ok * w[u].y, f[u].y, acc.y);
      acc.z = fmaf(ok * w[u].z, f[u].z, acc.z); acc.w = fmaf(ok * w[u].w, f[u].w, acc.w);
    }
  }
  float* p = out + (int64_t)(out_row ? out_row[seg] : seg) * X + x;
  if (b == seg_ptr[seg] && e == seg_ptr[seg + 1]) {
    if (accumulate) {
      const float4 c = ld4(p);
      acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
    }
    st4(p, acc);
  } else {
    atomicAdd(p + 0, acc.x); atomicAdd(p + 1, acc.y); atomicAdd(p + 2, acc.z); atomicAdd(p + 3, acc.w);
  }
}


// ---- cooperative forms (the gather kernels are bound by the NUMBER of vector-memory instructions) ------
// Per-edge scalars (row id, scale index, scale) are fetched by lane (sub % 4) for edge (sub % 4) of the step -- one
// instruction per step and stream -- and spread inside each quad with DPP broadcasts; rows: 4 per lane group and step.
// Valid when a scale (if any) is the same for the 4 lanes of a quad: per row, or per head with >= 16 floats per head.
template <int Q>
__device__ __forceinline__ int ss_quad_i(int v) { return __builtin_amdgcn_update_dpp(0, v, Q * 0x55, 0xf, 0xf, false); }
__device__ __forceinline__ int ss_bcast_i(int v, int q) {
  switch (q) {
    case 0: return ss_quad_i<0>(v);
    case 1: return ss_quad_i<1>(v);
    case 2: return ss_quad_i<2>(v);
    default: return ss_quad_i<3>(v);
  }
}
__device__ __forceinline__ float ss_bcast(float v, int q) { return __int_as_float(ss_bcast_i(__float_as_int(v), q)); }

// SHORT segments: a lane group per pack of whole segments (grouping_packs), rows of a segment summed in registers and
// stored once; ids a step ahead.
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_segment_sum_packed(const int32_t* __restrict__ pack_ptr, int64_t num_packs,
                                                                  const int32_t* __restrict__ seg_of_rank,
                                                                  const int32_t* __restrict__ p_row,
                                                                  const int32_t* __restrict__ p_scale,
                                                                  const float* __restrict__ scale, int scale_heads,
                                                                  const float* __restrict__ in, float* __restrict__ out,
                                                                  const int32_t* __restrict__ out_row, int accumulate,
                                                                  int contig, int scale_quad) {
  constexpr int EPW = 64 / LPR, X = LPR * 4, U = 4;
  static_assert(LPR >= 4, "needs whole quads per lane group");
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, q4 = sub & 3;
  const int64_t pid = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * EPW + slot;
  if (pid >= num_packs) return;
  const uint32_t pb = (uint32_t)pack_ptr[pid];
  if (pb >> 31) return;  // a long segment: HET_segment_sum_long takes its work items
  const int b = (int)pb, e = (int)((uint32_t)pack_ptr[pid + 1] & 0x7fffffffu);
  const int sld = scale_heads ? scale_heads : 1, sh = scale_heads ? x / (X / scale_heads) : 0;
  int jn = b + q4 < e ? b + q4 : e - 1;
  int rown = contig ? jn : p_row[jn], segn = seg_of_rank[jn], sin = scale ? (p_scale ? p_scale[jn] : jn) : 0;
  int cur = -1;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto flush = [&](int seg) {
    float* p = out + (int64_t)(out_row ? out_row[seg] : seg) * X + x;
    if (accumulate) {
      const float4 c = ld4(p);
      acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
    }
    st4_nt(p, acc);  // (non-temporal: the sums are read next by the node-major pass; RGCN step 2.096 -> 2.077 ms, three pairs: profiles/r05/ab_dense.txt call 16)
  };
  for (int j0 = b; j0 < e; j0 += U) {
    const int rowv = rown, segv = segn;
    // scale_quad: the scale is the same for the 4 lanes of a quad (per row, or heads of >= 16 floats): fetched
    // cooperatively too; else (narrow heads, e.g. HGT's 8 floats per head) each lane loads its own head's scale per edge
    const float wv = (scale && scale_quad) ? scale[(int64_t)sin * sld + sh] : 1.f;
    float wq[U];
    if (scale && !scale_quad) {
#pragma unroll
      for (int u = 0; u < U; ++u) wq[u] = scale[(int64_t)ss_bcast_i(sin, u) * sld + sh];
    }
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = ld4(in + (int64_t)ss_bcast_i(rowv, u) * X + x);
    jn = j0 + U + q4 < e ? j0 + U + q4 : e - 1;
    rown = contig ? jn : p_row[jn];
    segn = seg_of_rank[jn];
    if (scale) sin = p_scale ? p_scale[jn] : jn;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j0 + u < e) {  // uniform within the lane group
        const int sg = ss_bcast_i(segv, u);
        if (sg != cur) {
          if (cur >= 0) flush(cur);
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
          cur = sg;
        }
        const float w = (scale && !scale_quad) ? wq[u] : ss_bcast(wv, u);
        acc.x = fmaf(w, f[u].x, acc.x); acc.y = fmaf(w, f[u].y, acc.y);
        acc.z = fmaf(w, f[u].z, acc.z); acc.w = fmaf(w, f[u].w, acc.w);
      }
    }
  }
  if (cur >= 0) flush(cur);
}

// LONG segments: a wave per work item (<= HET_ITEM_MAX rows of one segment), lane groups round-robin, one cross-group
// reduction; an item that is not its whole segment adds atomically (the launcher cleared those rows).
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_segment_sum_long(const int32_t* __restrict__ long_items, int64_t num_long,
                                                                const int32_t* __restrict__ item_seg,
                                                                const int32_t* __restrict__ item_begin,
                                                                const int32_t* __restrict__ item_end,
                                                                const int32_t* __restrict__ seg_ptr,
                                                                const int32_t* __restrict__ p_row,
                                                                const int32_t* __restrict__ p_scale,
                                                                const float* __restrict__ scale, int scale_heads,
                                                                const float* __restrict__ in, float* __restrict__ out,
                                                                const int32_t* __restrict__ out_row, int accumulate,
                                                                int contig, int scale_quad) {
  constexpr int EPW = 64 / LPR, X = LPR * 4, U = 4;
  static_assert(LPR >= 4, "needs whole quads per lane group");
  const int lane = threadIdx.x & 63;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, q4 = sub & 3;
  const int64_t wid = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (wid >= num_long) return;
  const int item = long_items[wid];
  const int seg = item_seg[item], b = item_begin[item], e = item_end[item];
  const int sld = scale_heads ? scale_heads : 1, sh = scale_heads ? x / (X / scale_heads) : 0;
  int jn = b + slot + q4 * EPW < e ? b + slot + q4 * EPW : e - 1;
  int rown = contig ? jn : p_row[jn], sin = scale ? (p_scale ? p_scale[jn] : jn) : 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = b + slot; j0 < e; j0 += EPW * U) {
    const int rowv = rown;
    const float wv = (j0 + q4 * EPW < e) ? ((scale && scale_quad) ? scale[(int64_t)sin * sld + sh] : 1.f) : 0.f;
    float wq[U];
    if (scale && !scale_quad) {
#pragma unroll
      for (int u = 0; u < U; ++u) wq[u] = scale[(int64_t)ss_bcast_i(sin, u) * sld + sh];
    }
    float4 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = ld4(in + (int64_t)ss_bcast_i(rowv, u) * X + x);
    jn = j0 + (U + q4) * EPW < e ? j0 + (U + q4) * EPW : e - 1;
    rown = contig ? jn : p_row[jn];
    if (scale) sin = p_scale ? p_scale[jn] : jn;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // (padding edges of the last step: wv is 0 there; the per-lane scales are masked the same way)
      const float w = (scale && !scale_quad) ? (j0 + u * EPW < e ? wq[u] : 0.f) : ss_bcast(wv, u);
      acc.x = fmaf(w, f[u].x, acc.x); acc.y = fmaf(w, f[u].y, acc.y);
      acc.z = fmaf(w, f[u].z, acc.z); acc.w = fmaf(w, f[u].w, acc.w);
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
  }
  if (slot != 0) return;
  float* p = out + (int64_t)(out_row ? out_row[seg] : seg) * X + x;
  if (b == seg_ptr[seg] && e == seg_ptr[seg + 1]) {
    if (accumulate) {
      const float4 c = ld4(p);
      acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
    }
    st4_nt(p, acc);
  } else {
    atomicAdd(p + 0, acc.x); atomicAdd(p + 1, acc.y); atomicAdd(p + 2, acc.z); atomicAdd(p + 3, acc.w);
  }
}

// Rows of 1 or 2 floats ([E,H] attention terms with fewer than 4 heads): a thread per (item, float), plain loop.
__global__ __launch_bounds__(kBlock) void HET_segment_sum_narrow(const int32_t* __restrict__ item_seg,
                                                                 const int32_t* __restrict__ item_begin,
                                                                 const int32_t* __restrict__ item_end,
                                                                 const int32_t* __restrict__ seg_ptr, int64_t num_items,
                                                                 const int32_t* __restrict__ p_row, int X,
                                                                 const float* __restrict__ in, float* __restrict__ out,
                                                                 int accumulate, const int32_t* __restrict__ out_row) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= num_items * X) return;
  const int64_t item = t / X;
  const int c = (int)(t - item * X);
  const int seg = item_seg[item], b = item_begin[item], e = item_end[item];
  float acc = 0.f;
  for (int j = b; j < e; ++j) acc += in[(int64_t)p_row[j] * X + c];
  float* p = out + (int64_t)(out_row ? out_row[seg] : seg) * X + c;
  if (b == seg_ptr[seg] && e == seg_ptr[seg + 1]) *p = accumulate ? *p + acc : acc;
  else atomicAdd(p, acc);
}
// Rows of 4 floats (the per-edge term of RGAT's grad_er: 21 M rows of 16 bytes on ogbn-mag, segments of 17 on average):
// rank-parallel instead of a lane per segment.  A wave takes 64 * U consecutive ranks: ids and segment numbers are coalesced
// 4-byte streams, every lane has U independent 16-byte gathers in flight, and the rows of a segment are summed by a
// segmented scan across the lanes (ranks are sorted by segment); the last lane of a segment inside the wave stores the sum --
// plain store when the whole segment lies inside the wave's 64 ranks, float atomics for the pieces of longer / straddling
// ones (out is zero-filled by the launcher).  The lane-per-segment kernel ran at 0.59 ms on ogbn-mag (divergent segment
// lengths, strided id reads); random 16-byte gathers alone take 0.34 ms (exp/gather16.hip).
template <int U>
__global__ __launch_bounds__(kBlock) void HET_segment_sum_flat4(const int32_t* __restrict__ seg_of_rank,
                                                                const int32_t* __restrict__ seg_ptr,
                                                                const int32_t* __restrict__ p_row, int64_t E,
                                                                const float* __restrict__ in, float* __restrict__ out,
                                                                int contig) {
  const int lane = threadIdx.x & 63;
  const int64_t base = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 64 * U;
  if (base >= E) return;
  int key[U], row[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t j = base + u * 64 + lane, jc = j < E ? j : E - 1;
    key[u] = seg_of_rank[jc];
    row[u] = contig ? (int)jc : p_row[jc];
  }
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(in + (int64_t)row[u] * 4);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t first = base + u * 64;
    if (first >= E) break;
    float4 a = first + lane < E ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);  // (ranks past the end repeat the last key with zeros)
    const int k = key[u];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int nk = __shfl_up(k, off);
      const float nx = __shfl_up(a.x, off), ny = __shfl_up(a.y, off), nz = __shfl_up(a.z, off), nw = __shfl_up(a.w, off);
      if (lane >= off && nk == k) { a.x += nx; a.y += ny; a.z += nz; a.w += nw; }
    }
    const int knext = __shfl_down(k, 1);
    if (lane == 63 || knext != k) {  // last rank of segment k inside this wave's 64 ranks: a holds the sum of its ranks here
      float* p = out + (int64_t)k * 4;
      if (seg_ptr[k] >= first && seg_ptr[k + 1] <= first + 64) {
        *reinterpret_cast<float4*>(p) = a;
      } else {
        atomicAdd(p + 0, a.x); atomicAdd(p + 1, a.y); atomicAdd(p + 2, a.z); atomicAdd(p + 3, a.w);
      }
    }
  }
}
}  // namespace

bool segment_sum_supported(int X) { return X >= 4 && X <= 256 && (X & (X - 1)) == 0; }
bool segment_rows_supported(int X) { return X == 1 || X == 2 || segment_sum_supported(X); }

namespace {
// out[split_seg[k], :] = 0 (X4 float4 pieces per row)
__global__ __launch_bounds__(kBlock) void HET_segsum_zero_split(const int32_t* __restrict__ split_seg, int64_t n4, int X4,
                                                                float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n4) return;
  const int64_t k = t / X4;
  reinterpret_cast<float4*>(out)[(int64_t)split_seg[k] * X4 + (t - k * X4)] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(256) void HET_gather_by_index(const int32_t* __restrict__ idx, const float* __restrict__ in, int64_t n,
                                                           int H, float* __restrict__ out) {
  const int64_t total = n * H;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t j = t / H;
    out[t] = in[(int64_t)idx[j] * H + (t - j * H)];
  }
}
}  // namespace

int launch_gather_by_p1(const het_grouping* g, const float* values, int H, float* out, hipStream_t s) {
  HET_REQUIRE(g && g->p1 && H >= 1, "gather by payload1: the grouping carries no second payload");
  if (g->E == 0) return HET_OK;
  int64_t nb = ceil_div64(g->E * H, 256);
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(HET_gather_by_index, dim3((unsigned)nb), dim3(256), 0, s, g->p1, values, g->E, H, out);
  HET_LAUNCH_CHECK("HET_gather_by_index");
  return HET_OK;
}

int launch_segment_sum(const het_grouping* g, const float* in, float* out, int X, const float* scale, hipStream_t s,
                       int scale_heads, int64_t scatter_rows, int accumulate, int scale_by_p0, int nt_in, int scale_sorted) {
  HET_REQUIRE(segment_rows_supported(X) && g->p0, "segment sum: unsupported shape or grouping");
  if (X < 4) {
    HET_REQUIRE(!scale, "segment sum: rows of fewer than 4 floats take no scale");
    if (!accumulate) {
      if (scatter_rows >= 0) HET_HIP(hipMemsetAsync(out, 0, sizeof(float) * scatter_rows * X, s));
      else if (g->num_split > 0) HET_HIP(hipMemsetAsync(out, 0, sizeof(float) * g->S * X, s));
    }
    if (g->S == 0) return HET_OK;
    hipLaunchKernelGGL(HET_segment_sum_narrow, dim3((unsigned)ceil_div64(g->num_items * X, kBlock)), dim3(kBlock), 0, s,
                       g->item_seg, g->item_begin, g->item_end, g->seg_ptr, g->num_items, g->p0, X, in, out, accumulate,
                       scatter_rows >= 0 ? g->seg_key : nullptr);
    HET_LAUNCH_CHECK("HET_segment_sum_narrow");
    return HET_OK;
  }
  HET_REQUIRE(scale_heads == 0 || scale_heads == X || (X % scale_heads == 0 && (X / scale_heads) % 4 == 0),
              "segment sum: a head must cover whole float4 pieces (or scale_heads == X: one scale per element)");
  // scatter_rows >= 0: out has that many rows and segment s lands in row seg_key[s] (rows without a
  // segment read zero unless accumulating); otherwise out is dense [S, X]
  const int32_t* out_row = scatter_rows >= 0 ? g->seg_key : nullptr;
  bool split_rows_zeroed = true;
  if (!accumulate) {
    if (scatter_rows >= 0) HET_HIP(hipMemsetAsync(out, 0, sizeof(float) * scatter_rows * X, s));
    else if (g->num_split > 0) split_rows_zeroed = false;  // (below: only the rows the atomics add to)
  }
  if (g->S == 0) return HET_OK;
  // Dense [S, X] output: every segment that is one work item is STORED; only the segments split over several items are summed
  // with atomics and need zeros first -- a few thousand rows, not the whole output (ogbn-mag, [2.4 M, 64]: a 0.6 GB fill, 0.1 ms)
  auto zero_split_rows = [&]() -> int {
    if (split_rows_zeroed) return HET_OK;
    split_rows_zeroed = true;
    if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) {
      HET_HIP(hipMemsetAsync(out, 0, sizeof(float) * g->S * X, s));
      return HET_OK;
    }
    const int64_t n4 = g->num_split * (X / 4);
    hipLaunchKernelGGL(HET_segsum_zero_split, dim3((unsigned)ceil_div64(n4, kBlock)), dim3(kBlock), 0, s, g->split_seg, n4, X / 4, out);
    HET_LAUNCH_CHECK("HET_segsum_zero_split");
    return HET_OK;
  };
  static const bool flat_off = [] { const char* v = getenv("HET_SEGSUM_FLAT"); return v && v[0] == '0'; }();  // A/B switch
  if (X == 4 && !scale && scatter_rows < 0 && !accumulate && !nt_in && !flat_off && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    if (int rc = grouping_seg_of_rank(g, s)) return rc;
    HET_HIP(hipMemsetAsync(out, 0, sizeof(float) * g->S * X, s));  // (pieces of straddling segments are added atomically)
    constexpr int U = 4;
    HET_KTIME("HET_segment_sum", s);
    hipLaunchKernelGGL(HET_segment_sum_flat4<U>, dim3((unsigned)ceil_div64(g->E, (int64_t)(kBlock / 64) * 64 * U)), dim3(kBlock), 0, s,
                       g->seg_of_rank, g->seg_ptr, g->p0, g->E, in, out, (int)g->p0_contiguous);
    HET_LAUNCH_CHECK("HET_segment_sum_flat4");
    return HET_OK;
  }
  // scale_sorted: `scale` is already in the grouping's order (scale[j] belongs to sorted rank j: launch_gather_by_p1) -- a
  // coalesced stream instead of an index load + a random 4-byte gather per edge (NULL index list = the rank itself)
  if (int rc = zero_split_rows()) return rc;
  const int32_t* p_scale = scale_sorted ? nullptr : ((g->p1 && !scale_by_p0) ? g->p1 : g->p0);
  const unsigned nb = (unsigned)ceil_div64(g->num_items, (int64_t)(kBlock / 64) * (64 / (X / 4)));
  const int contig = g->p0_contiguous;  // same box: 2.35 -> 2.25 ms for the a2 backward of C3
  HET_KTIME("HET_segment_sum", s);
  // cooperative kernels: rows of >= 16 floats, a scale shared by the 4 lanes of a quad (per row, or heads of >= 16 floats),
  // cached rows (nt_in streams keep the item kernel)
  static const bool coop_off = [] { const char* v = getenv("HET_SEGSUM_COOP"); return v && v[0] == '0'; }();  // A/B switch
  const int LPRv = X / 4;
  const bool coop = !coop_off && LPRv >= 4 && LPRv <= 64 && !nt_in && !(scale && scale_heads == X);
  const int scale_quad = !scale || scale_heads == 0 || (X / scale_heads) % 16 == 0;
  if (coop) {
    if (int rc = grouping_packs(g, s)) return rc;
    const unsigned nbp = (unsigned)ceil_div64(g->num_packs, (int64_t)(kBlock / 64) * (64 / LPRv));
#define HET_SSP(L)                                                                                                         \
  hipLaunchKernelGGL(HET_segment_sum_packed<L>, dim3(nbp), dim3(kBlock), 0, s, g->pack_ptr, g->num_packs, g->seg_of_rank,   \
                     g->p0, p_scale, scale, scale_heads, in, out, out_row, accumulate, contig, scale_quad)
    switch (LPRv) {
      case 4: HET_SSP(4); break;
      case 8: HET_SSP(8); break;
      case 16: HET_SSP(16); break;
      case 32: HET_SSP(32); break;
      default: HET_SSP(64); break;
    }
#undef HET_SSP
    HET_LAUNCH_CHECK("HET_segment_sum_packed");
    if (g->num_long_items > 0) {
      const unsigned nbl = (unsigned)ceil_div64(g->num_long_items, kBlock / 64);
#define HET_SSL(L)                                                                                                         \
  hipLaunchKernelGGL(HET_segment_sum_long<L>, dim3(nbl), dim3(kBlock), 0, s, g->long_items, g->num_long_items, g->item_seg, \
                     g->item_begin, g->item_end, g->seg_ptr, g->p0, p_scale, scale, scale_heads, in, out, out_row,          \
                     accumulate, contig, scale_quad)
      switch (LPRv) {
        case 4: HET_SSL(4); break;
        case 8: HET_SSL(8); break;
        case 16: HET_SSL(16); break;
        case 32: HET_SSL(32); break;
        default: HET_SSL(64); break;
      }
#undef HET_SSL
      HET_LAUNCH_CHECK("HET_segment_sum_long");
    }
    return HET_OK;
  }
#define HET_SS(L)                                                                                                   \
  hipLaunchKernelGGL(HET_segment_sum<L>, dim3(nb), dim3(kBlock), 0, s, g->item_seg, g->item_begin, g->item_end,     \
                     g->seg_ptr, g->num_items, g->p0, p_scale, scale, scale_heads, in, out, out_row, accumulate, nt_in,  \
                     contig)
  switch (X / 4) {
    case 1: HET_SS(1); break;
    case 2: HET_SS(2); break;
    case 4: HET_SS(4); break;
    case 8: HET_SS(8); break;
    case 16: HET_SS(16); break;
    case 32: HET_SS(32); break;
    default: HET_SS(64); break;
  }
#undef HET_SS
  HET_LAUNCH_CHECK("HET_segment_sum");
  return HET_OK;
}

// ---- segment broadcast: the inverse of the segment sum -----------------------------------------------------------
// out[p0[j], :] = in[seg(j), :] for every sorted rank j (and optionally out2[p0[j], :] = in2[seg(j), :] with X2 floats
// per row).  Rank-parallel and streaming: a lane group (X/4 lanes) per rank, U ranks in flight; consecutive ranks
// share their segment's row (cache hits), the row ids p0[j] and the segment ids are read as coalesced 4-byte streams.
// Used by the per-edge projection: rows that share (relation, node) are identical, so the GEMM runs on the S distinct
// rows and this kernel duplicates them.  (An item-parallel version -- one lane group per work item -- ran at 2.7 TB/s:
// the average segment has 6 rows.)
namespace {
__global__ __launch_bounds__(kBlock) void HET_grouping_seg_of_rank(const int32_t* __restrict__ seg_ptr, int64_t S, int64_t E,
                                                                    int32_t* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < E; j += (int64_t)gridDim.x * kBlock) {
    int64_t lo = 0, hi = S;  // last segment s with seg_ptr[s] <= j
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (seg_ptr[mid] <= j) lo = mid; else hi = mid;
    }
    out[j] = (int32_t)lo;
  }
}

__global__ __launch_bounds__(kBlock) void HET_segment_broadcast_narrow(const int32_t* __restrict__ seg_of_rank,
                                                                        const int32_t* __restrict__ p_row, int64_t total, int X,
                                                                        const float* __restrict__ in, float* __restrict__ out) {
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t j = t / X;
    const int c = (int)(t - j * X);
    out[(int64_t)p_row[j] * X + c] = in[(int64_t)seg_of_rank[j] * X + c];
  }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_segment_broadcast(const int32_t* __restrict__ seg_of_rank,
                                                                 const int32_t* __restrict__ p_row, int64_t E,
                                                                 const float* __restrict__ in, float* __restrict__ out,
                                                                 const float* __restrict__ in2, float* __restrict__ out2,
                                                                 int X2) {
  constexpr int EPW = 64 / LPR, X = LPR * 4, U = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4;
  const int64_t step = (int64_t)gridDim.x * (kBlock / 64) * EPW * U;
  for (int64_t base = (int64_t)blockIdx.x * (kBlock / 64) * EPW * U; base < E; base += step) {
    int64_t row[U], seg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = base + (wave * U + u) * EPW + slot, jc = j < E ? j : E - 1;  // past the end: last rank again
      row[u] = p_row[jc];
      seg[u] = seg_of_rank[jc];
    }
    float4 v[U];
    float v2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4(in + seg[u] * X + x);
    if (in2) {
#pragma unroll
      for (int u = 0; u < U; ++u) v2[u] = in2[seg[u] * X2 + (sub < X2 ? sub : 0)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      st4(out + row[u] * X + x, v[u]);
      if (in2 && sub < X2) out2[row[u] * X2 + sub] = v2[u];
    }
  }
}
}  // namespace

static std::mutex g_seg_of_rank_mu;

int grouping_seg_of_rank(const het_grouping* g, hipStream_t s) {
  // one-time, cached in the grouping.  Built under a lock and published only after the fill kernel has finished, so a
  // second stream or thread that finds the pointer set never reads a half-written buffer
  std::lock_guard<std::mutex> lk(g_seg_of_rank_mu);
  if (g->seg_of_rank || g->E == 0) return HET_OK;
  int32_t* p = nullptr;
  HET_HIP(het_malloc_e((void**)&p, sizeof(int32_t) * g->E, s));
  int64_t nb0 = ceil_div64(g->E, kBlock);
  hipLaunchKernelGGL(HET_grouping_seg_of_rank, dim3((unsigned)(nb0 > 65536 ? 65536 : nb0)), dim3(kBlock), 0, s, g->seg_ptr,
                     g->S, g->E, p);
  if (hipGetLastError() != hipSuccess) {
    (void)het_free_e(p);
    HET_REQUIRE(false, "HET_grouping_seg_of_rank: launch failed");
  }
  if (hipStreamSynchronize(s) != hipSuccess) {
    (void)het_free_e(p);
    HET_REQUIRE(false, "HET_grouping_seg_of_rank: kernel failed");
  }
  g->seg_of_rank = p;
  return HET_OK;
}

int launch_segment_broadcast(const het_grouping* g, const float* in, float* out, int X, const float* in2, float* out2,
                             int X2, hipStream_t s) {
  HET_REQUIRE(segment_rows_supported(X) && g->p0 && (!in2 || (X >= 4 && out2 && X2 >= 1 && X2 <= X / 4)),
              "segment broadcast: unsupported shape or grouping");
  if (g->S == 0 || g->E == 0) return HET_OK;
  if (int rc = grouping_seg_of_rank(g, s)) return rc;
  if (X < 4) {
    int64_t nbn = ceil_div64(g->E * X, kBlock);
    if (nbn > 65536) nbn = 65536;
    hipLaunchKernelGGL(HET_segment_broadcast_narrow, dim3((unsigned)nbn), dim3(kBlock), 0, s, g->seg_of_rank, g->p0, g->E * X, X,
                       in, out);
    HET_LAUNCH_CHECK("HET_segment_broadcast_narrow");
    return HET_OK;
  }
  const int epw = 64 / (X / 4);
  int64_t nb = ceil_div64(g->E, (int64_t)(kBlock / 64) * epw * 4);
  if (nb > 256 * 64) nb = 256 * 64;
#define HET_SB(L) hipLaunchKernelGGL(HET_segment_broadcast<L>, dim3((unsigned)nb), dim3(kBlock), 0, s, g->seg_of_rank, g->p0, \
                                     g->E, in, out, in2, out2, X2)
  switch (X / 4) {
    case 1: HET_SB(1); break;
    case 2: HET_SB(2); break;
    case 4: HET_SB(4); break;
    case 8: HET_SB(8); break;
    case 16: HET_SB(16); break;
    case 32: HET_SB(32); break;
    default: HET_SB(64); break;
  }
#undef HET_SB
  HET_LAUNCH_CHECK("HET_segment_broadcast");
  return HET_OK;
}
