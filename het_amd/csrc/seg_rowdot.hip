// Row-dot kernels (see seg_rowdot.hip.h).  A row of H*K floats is covered by LPR = H*K/4 lanes
// (one float4 each, fully coalesced); KL = K/4 adjacent lanes share a head and combine their
// partial dot products with xor-shuffles.  A wave handles 64/LPR rows per step.
#include <stdlib.h>

#include "seg_rowdot.hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 2048;  // rows of one relation per workgroup (upper bound)
// rows per workgroup so that a short list (the S = 1-4 M distinct (relation, node) rows, not E = 21 M edges) still
// gives every CU many waves: 1.5 M rows at 2048 per workgroup were 730 workgroups, 3 per CU
int chunk_for(int64_t num_rows) {
  int c = kChunk;
  while (c > 128 && num_rows / c < 8192) c >>= 1;
  return c;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

constexpr int U = 4;  // rows per lane group and step; loads are issued in independent phases

// Row of (step base, wave, u, slot) -- every (wave, u) pair covers EPW consecutive rows (1 KiB).
#define HET_ROWDOT_ROWS(EPW)                                                      \
  idx_t ic[U];                                                                    \
  bool ok[U];                                                                     \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                 \
    const idx_t i = base + (wave * U + u) * (EPW) + slot;                         \
    ok[u] = i < re;                                                               \
    ic[u] = ok[u] ? i : re - 1;                                                   \
  }                                                                               \
  idx_t gi[U], si[U];                                                             \
  if (a.gather) {                                                                 \
    _Pragma("unroll") for (int u = 0; u < U; ++u) gi[u] = a.gather[ic[u]];        \
  } else {                                                                        \
    _Pragma("unroll") for (int u = 0; u < U; ++u) gi[u] = ic[u];                  \
  }                                                                               \
  if (a.scatter == a.gather) {                                                    \
    _Pragma("unroll") for (int u = 0; u < U; ++u) si[u] = gi[u];                  \
  } else if (a.scatter) {                                                         \
    _Pragma("unroll") for (int u = 0; u < U; ++u) si[u] = a.scatter[ic[u]];       \
  } else {                                                                        \
    _Pragma("unroll") for (int u = 0; u < U; ++u) si[u] = ic[u];                  \
  }

// (fetching the ids of step k+1 before touching the rows of step k was measured on these kernels: no gain -- the lists are
//  short and the launches small; what they wait for is not the id -> row round trip)
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rowdot_fwd(RowDotArgs a, int chunk) {
  constexpr int EPW = 64 / LPR;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, chunk, blockIdx.x, r, rb, re)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, KL = a.K >> 2, h = x / a.K;
  const int HK = a.H * a.K;
  const float4 w = ld4(a.W + (int64_t)r * HK + x);
  for (idx_t base = rb; base < re; base += 4 * EPW * U) {
    HET_ROWDOT_ROWS(EPW)
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4(a.A + gi[u] * HK + x);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float p = v[u].x * w.x + v[u].y * w.y + v[u].z * w.z + v[u].w * w.w;
      for (int off = KL >> 1; off > 0; off >>= 1) p += __shfl_xor(p, off);
      if (ok[u] && (sub & (KL - 1)) == 0) a.out[si[u] * a.H + h] = p;
    }
  }
}

// grad_A[g_i, h, :] += go[s_i, h] * Wt[r, h, :]
// MODE 0: atomics; 1: unique rows, read-modify-write; 2: unique rows, plain store
template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void HET_rowdot_bwd_dx(RowDotArgs a, int chunk) {
  constexpr int EPW = 64 / LPR;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, chunk, blockIdx.x, r, rb, re)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / a.K;
  const int HK = a.H * a.K;
  const float4 w = ld4(a.W + (int64_t)r * HK + x);
  for (idx_t base = rb; base < re; base += 4 * EPW * U) {
    HET_ROWDOT_ROWS(EPW)
    float g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) g[u] = a.go[si[u] * a.H + h];
    float4 c[U];
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) c[u] = ld4(a.out + gi[u] * HK + x);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      float* p = a.out + gi[u] * HK + x;
      if (MODE == 2) {
        st4(p, make_float4(g[u] * w.x, g[u] * w.y, g[u] * w.z, g[u] * w.w));
      } else if (MODE == 1) {
        st4(p, make_float4(fmaf(g[u], w.x, c[u].x), fmaf(g[u], w.y, c[u].y), fmaf(g[u], w.z, c[u].z),
                           fmaf(g[u], w.w, c[u].w)));
      } else {
        atomicAdd(p + 0, g[u] * w.x); atomicAdd(p + 1, g[u] * w.y);
        atomicAdd(p + 2, g[u] * w.z); atomicAdd(p + 3, g[u] * w.w);
      }
    }
  }
}

// dW[r, h, :] += sum_i go[s_i, h] * A[g_i, h, :]
template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rowdot_bwd_dw(RowDotArgs a, int chunk) {
  constexpr int EPW = 64 / LPR;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, chunk, blockIdx.x, r, rb, re)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / a.K;
  const int HK = a.H * a.K;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (idx_t base = rb; base < re; base += 4 * EPW * U) {
    HET_ROWDOT_ROWS(EPW)
    float g[U];
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) g[u] = a.go[si[u] * a.H + h];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4(a.A + gi[u] * HK + x);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float gu = ok[u] ? g[u] : 0.f;
      acc.x = fmaf(gu, v[u].x, acc.x); acc.y = fmaf(gu, v[u].y, acc.y);
      acc.z = fmaf(gu, v[u].z, acc.z); acc.w = fmaf(gu, v[u].w, acc.w);
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
  }
  __shared__ float4 part[4][64];
  if (slot == 0) part[wave][sub] = acc;
  __syncthreads();
  if (wave == 0 && slot == 0) {
    float4 t = part[0][sub];
    for (int wv = 1; wv < 4; ++wv) {
      const float4 o = part[wv][sub];
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    float* p = a.out + (int64_t)r * HK + x;
    atomicAdd(p + 0, t.x); atomicAdd(p + 1, t.y); atomicAdd(p + 2, t.z); atomicAdd(p + 3, t.w);
  }
}

// ---- one shared input head, H weight heads, D_out = 1 ----------------------------------------------------
// A row of K floats is covered by LPR = K/4 lanes; every lane forms its 4-element partial dot product with each
// of the H weight vectors and the LPR lanes combine them with xor-shuffles.
template <int LPR, int H>
__global__ __launch_bounds__(kBlock) void HET_rowdot1h_fwd(RowDotArgs a, int chunk) {
  constexpr int EPW = 64 / LPR, K = LPR * 4;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, chunk, blockIdx.x, r, rb, re)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4;
  float4 w[H];
#pragma unroll
  for (int h = 0; h < H; ++h) w[h] = ld4(a.W + ((int64_t)r * H + h) * K + x);
  for (idx_t base = rb; base < re; base += 4 * EPW * U) {
    HET_ROWDOT_ROWS(EPW)
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4(a.A + gi[u] * K + x);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float p[H];
#pragma unroll
      for (int h = 0; h < H; ++h) p[h] = v[u].x * w[h].x + v[u].y * w[h].y + v[u].z * w[h].z + v[u].w * w[h].w;
#pragma unroll
      for (int off = LPR >> 1; off > 0; off >>= 1)
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] += __shfl_xor(p[h], off);
      if (ok[u] && sub < H) {
        float o = p[0];
#pragma unroll
        for (int h = 1; h < H; ++h) o = (sub == h) ? p[h] : o;
        a.out[si[u] * H + sub] = o;
      }
    }
  }
}

// One lane per COLUMN of the input row: the K adds of a row are K consecutive floats, so an atomic instruction covers
// whole 128-byte lines (a float4-per-lane mapping spreads it over four times as many lines and runs at a quarter of the
// rate: 1.02 ms for 1.5 M rows of 64 floats on ogbn-mag).
template <int LPR, int H>
__global__ __launch_bounds__(kBlock) void HET_rowdot1h_bwd_dx(RowDotArgs a, int chunk) {
  constexpr int K = LPR * 4, RPB = kBlock / K > 0 ? kBlock / K : 1;  // rows per block and pass
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, chunk, blockIdx.x, r, rb, re)) return;
  float w[H];
  for (int col = threadIdx.x % K; col < K; col += kBlock) {  // one trip unless K > kBlock
#pragma unroll
    for (int h = 0; h < H; ++h) w[h] = a.W[((int64_t)r * H + h) * K + col];
    for (idx_t base = rb + threadIdx.x / K; base < re; base += (idx_t)RPB * U) {
      idx_t gi[U], si[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const idx_t i = base + (idx_t)u * RPB;
        ok[u] = i < re;
        const idx_t ic = ok[u] ? i : re - 1;
        gi[u] = a.gather ? a.gather[ic] : ic;
        si[u] = a.scatter == a.gather ? gi[u] : (a.scatter ? a.scatter[ic] : ic);
      }
      float o[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        o[u] = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) o[u] = fmaf(a.go[si[u] * H + h], w[h], o[u]);
      }
      if (a.rmw) {  // the launch's rows hit distinct output rows: plain read-modify-write
        float c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = a.out[gi[u] * K + col];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (ok[u]) a.out[gi[u] * K + col] = c[u] + o[u];
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (ok[u]) atomicAdd(a.out + gi[u] * K + col, o[u]);
      }
    }
  }
}

// The read-modify-write form (a.rmw: the launch's rows hit distinct output rows) with a lane group per row: one 16-byte
// load and store per lane, 64/LPR rows per instruction, the H gradients of a row as one vector load when H == 4.
template <int LPR, int H>
__global__ __launch_bounds__(kBlock) void HET_rowdot1h_bwd_dx_rmw(RowDotArgs a, int chunk) {
  constexpr int EPW = 64 / LPR, K = LPR * 4;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, chunk, blockIdx.x, r, rb, re)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4;
  float4 w[H];
#pragma unroll
  for (int h = 0; h < H; ++h) w[h] = ld4(a.W + ((int64_t)r * H + h) * K + x);
  for (idx_t base = rb; base < re; base += 4 * EPW * U) {
    HET_ROWDOT_ROWS(EPW)
    float g[U][H];
    float4 c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (H == 4) {
        const float4 t = ld4(a.go + si[u] * H);
        g[u][0] = t.x; g[u][1] = t.y; g[u][2] = t.z; g[u][3] = t.w;
      } else {
#pragma unroll
        for (int h = 0; h < H; ++h) g[u][h] = a.go[si[u] * H + h];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = ld4(a.out + gi[u] * K + x);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        c[u].x = fmaf(g[u][h], w[h].x, c[u].x); c[u].y = fmaf(g[u][h], w[h].y, c[u].y);
        c[u].z = fmaf(g[u][h], w[h].z, c[u].z); c[u].w = fmaf(g[u][h], w[h].w, c[u].w);
      }
      st4(a.out + gi[u] * K + x, c[u]);
    }
  }
}

template <int LPR, int H>
__global__ __launch_bounds__(kBlock) void HET_rowdot1h_bwd_dw(RowDotArgs a, int chunk) {
  constexpr int EPW = 64 / LPR, K = LPR * 4;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, chunk, blockIdx.x, r, rb, re)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4;
  float4 acc[H];
#pragma unroll
  for (int h = 0; h < H; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (idx_t base = rb; base < re; base += 4 * EPW * U) {
    HET_ROWDOT_ROWS(EPW)
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4(a.A + gi[u] * K + x);
    float g[U][H];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (H == 4 && (reinterpret_cast<uintptr_t>(a.go) & 15) == 0) {  // the H gradients of a row as one 16-byte load
        const float4 t = ld4(a.go + si[u] * H);
        g[u][0] = t.x; g[u][1 % H] = t.y; g[u][2 % H] = t.z; g[u][3 % H] = t.w;
      } else {
#pragma unroll
        for (int h = 0; h < H; ++h) g[u][h] = a.go[si[u] * H + h];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float gu = ok[u] ? g[u][h] : 0.f;
        acc[h].x = fmaf(gu, v[u].x, acc[h].x); acc[h].y = fmaf(gu, v[u].y, acc[h].y);
        acc[h].z = fmaf(gu, v[u].z, acc[h].z); acc[h].w = fmaf(gu, v[u].w, acc[h].w);
      }
    }
  }
  __shared__ float4 part[4][64];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float4 t = acc[h];
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      t.x += __shfl_xor(t.x, off); t.y += __shfl_xor(t.y, off);
      t.z += __shfl_xor(t.z, off); t.w += __shfl_xor(t.w, off);
    }
    __syncthreads();
    if (slot == 0) part[wave][sub] = t;
    __syncthreads();
    if (wave == 0 && slot == 0) {
      float4 o = part[0][sub];
      for (int wv = 1; wv < 4; ++wv) {
        const float4 q = part[wv][sub];
        o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
      }
      float* p = a.out + ((int64_t)r * H + h) * K + x;
      atomicAdd(p + 0, o.x); atomicAdd(p + 1, o.y); atomicAdd(p + 2, o.z); atomicAdd(p + 3, o.w);
    }
  }
}

inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

#define HET_ROWDOT_DISPATCH(LPRV, CALL)                 \
  switch (LPRV) {                                       \
    case 1: { constexpr int LPR = 1; CALL; break; }     \
    case 2: { constexpr int LPR = 2; CALL; break; }     \
    case 4: { constexpr int LPR = 4; CALL; break; }     \
    case 8: { constexpr int LPR = 8; CALL; break; }     \
    case 16: { constexpr int LPR = 16; CALL; break; }   \
    case 32: { constexpr int LPR = 32; CALL; break; }   \
    default: { constexpr int LPR = 64; CALL; break; }   \
  }

}  // namespace

bool rowdot_supported(int H, int K) {
  return K >= 4 && is_pow2(K) && is_pow2(H) && (int64_t)H * K / 4 <= 64;
}

int launch_rowdot_fwd(const RowDotArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  const int chunk = chunk_for(a.num_rows);
  dim3 grid((unsigned)(ceil_div64(a.num_rows, chunk) + a.num_segs)), block(kBlock);
  HET_ROWDOT_DISPATCH(a.H * a.K / 4, hipLaunchKernelGGL(HET_rowdot_fwd<LPR>, grid, block, 0, s, a, chunk));
  HET_LAUNCH_CHECK("HET_rowdot_fwd");
  return HET_OK;
}

int launch_rowdot_bwd_dx(const RowDotArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  const int chunk = chunk_for(a.num_rows);
  dim3 grid((unsigned)(ceil_div64(a.num_rows, chunk) + a.num_segs)), block(kBlock);
  if (a.unique_rows && a.overwrite) {
    HET_ROWDOT_DISPATCH(a.H * a.K / 4, hipLaunchKernelGGL((HET_rowdot_bwd_dx<LPR, 2>), grid, block, 0, s, a, chunk));
  } else if (a.unique_rows) {
    HET_ROWDOT_DISPATCH(a.H * a.K / 4, hipLaunchKernelGGL((HET_rowdot_bwd_dx<LPR, 1>), grid, block, 0, s, a, chunk));
  } else {
    HET_ROWDOT_DISPATCH(a.H * a.K / 4, hipLaunchKernelGGL((HET_rowdot_bwd_dx<LPR, 0>), grid, block, 0, s, a, chunk));
  }
  HET_LAUNCH_CHECK("HET_rowdot_bwd_dx");
  return HET_OK;
}

int launch_rowdot_bwd_dw(const RowDotArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  // Every workgroup ends with one atomic flush into the SAME H*K floats of its relation, and those serialise (~50 ns per
  // workgroup): on ogbn-mag (2.4 M rows) 8192 / 4096 / 2048 / 1024 / 512 workgroups take 0.48 / 0.27 / 0.17 / 0.136 / 0.131 ms.
  static const int64_t min_chunk = [] { const char* v = getenv("HET_ROWDOT_DW_MIN"); return v ? (int64_t)atoi(v) : 512; }();  // A/B switches (the minimum only matters for short lists: a rank's share of a partition)
  static const int64_t n_wg = [] { const char* v = getenv("HET_ROWDOT_DW_WGS"); return v ? (int64_t)atoi(v) : 512; }();
  int64_t chunk = ceil_div64(a.num_rows, n_wg);
  if (chunk < min_chunk) chunk = min_chunk;
  dim3 grid((unsigned)(ceil_div64(a.num_rows, chunk) + a.num_segs)), block(kBlock);
  HET_ROWDOT_DISPATCH(a.H * a.K / 4, hipLaunchKernelGGL(HET_rowdot_bwd_dw<LPR>, grid, block, 0, s, a, (int)chunk));
  HET_LAUNCH_CHECK("HET_rowdot_bwd_dw");
  return HET_OK;
}

bool rowdot1h_supported(int H, int K) {
  return (H == 1 || H == 2 || H == 4 || H == 8) && K >= 4 * H && is_pow2(K) && K / 4 <= 64;
}

#define HET_ROWDOT1H_DISPATCH(KERNEL, ...)                                                           \
  switch (a.H) {                                                                                     \
    case 1: HET_ROWDOT_DISPATCH(a.K / 4, hipLaunchKernelGGL((KERNEL<LPR, 1>), grid, block, 0, s, __VA_ARGS__)); break; \
    case 2: HET_ROWDOT_DISPATCH(a.K / 4, hipLaunchKernelGGL((KERNEL<LPR, 2>), grid, block, 0, s, __VA_ARGS__)); break; \
    case 4: HET_ROWDOT_DISPATCH(a.K / 4, hipLaunchKernelGGL((KERNEL<LPR, 4>), grid, block, 0, s, __VA_ARGS__)); break; \
    default: HET_ROWDOT_DISPATCH(a.K / 4, hipLaunchKernelGGL((KERNEL<LPR, 8>), grid, block, 0, s, __VA_ARGS__)); break; \
  }

int launch_rowdot1h_fwd(const RowDotArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  const int chunk = chunk_for(a.num_rows);
  dim3 grid((unsigned)(ceil_div64(a.num_rows, chunk) + a.num_segs)), block(kBlock);
  HET_ROWDOT1H_DISPATCH(HET_rowdot1h_fwd, a, chunk)
  HET_LAUNCH_CHECK("HET_rowdot1h_fwd");
  return HET_OK;
}

int launch_rowdot1h_bwd_dx(const RowDotArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  const int chunk = chunk_for(a.num_rows);
  dim3 grid((unsigned)(ceil_div64(a.num_rows, chunk) + a.num_segs)), block(kBlock);
  const bool vec = ((reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.go) | reinterpret_cast<uintptr_t>(a.W)) & 15) == 0;
  if (a.rmw && vec) {
    HET_ROWDOT1H_DISPATCH(HET_rowdot1h_bwd_dx_rmw, a, chunk)
  } else {
    HET_ROWDOT1H_DISPATCH(HET_rowdot1h_bwd_dx, a, chunk)
  }
  HET_LAUNCH_CHECK("HET_rowdot1h_bwd_dx");
  return HET_OK;
}

int launch_rowdot1h_bwd_dw(const RowDotArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  static const int64_t min_chunk = [] { const char* v = getenv("HET_ROWDOT_DW_MIN"); return v ? (int64_t)atoi(v) : 512; }();  // A/B switches (the minimum only matters for short lists: a rank's share of a partition)
  static const int64_t n_wg = [] { const char* v = getenv("HET_ROWDOT_DW_WGS"); return v ? (int64_t)atoi(v) : 512; }();
  int64_t chunk = ceil_div64(a.num_rows, n_wg);  // see launch_rowdot_bwd_dw
  if (chunk < min_chunk) chunk = min_chunk;
  dim3 grid((unsigned)(ceil_div64(a.num_rows, chunk) + a.num_segs)), block(kBlock);
  HET_ROWDOT1H_DISPATCH(HET_rowdot1h_bwd_dw, a, (int)chunk)
  HET_LAUNCH_CHECK("HET_rowdot1h_bwd_dw");
  return HET_OK;
}
