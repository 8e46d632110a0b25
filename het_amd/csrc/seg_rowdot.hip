// Row-dot kernels (see seg_rowdot.hip.h).  A row of H*K floats is covered by LPR = H*K/4 lanes
// (one float4 each, fully coalesced); KL = K/4 adjacent lanes share a head and combine their
// partial dot products with xor-shuffles.  A wave handles 64/LPR rows per step.
#include "seg_rowdot.hip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 2048;  // rows of one relation per workgroup

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <int LPR>
__global__ __launch_bounds__(kBlock) void HET_rowdot_fwd(RowDotArgs a) {
  constexpr int EPW = 64 / LPR;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, kChunk, blockIdx.x, r, rb, re)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, KL = a.K >> 2, h = x / a.K;
  const int HK = a.H * a.K;
  const float4 w = ld4(a.W + (int64_t)r * HK + x);
  for (idx_t i = rb + wave * EPW + slot; i < re; i += 4 * EPW) {
    const idx_t gi = a.gather ? a.gather[i] : i;
    const float4 v = ld4(a.A + gi * HK + x);
    float p = v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
    for (int off = KL >> 1; off > 0; off >>= 1) p += __shfl_xor(p, off);
    if ((sub & (KL - 1)) == 0) a.out[(a.scatter ? a.scatter[i] : i) * a.H + h] = p;
  }
}

// grad_A[g_i, h, :] += go[s_i, h] * Wt[r, h, :]
template <int LPR, bool UNIQUE>
__global__ __launch_bounds__(kBlock) void HET_rowdot_bwd_dx(RowDotArgs a) {
  constexpr int EPW = 64 / LPR;
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, kChunk, blockIdx.x, r, rb, re)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR, x = sub * 4, h = x / a.K;
  const int HK = a.H * a.K;
  const float4 w = ld4(a.W + (int64_t)r * HK + x);
  for (idx_t i = rb + wave * EPW + slot; i < re; i += 4 * EPW) {
    const idx_t gi = a.gather ? a.gather[i] : i;
    const float g = a.go[(a.scatter ? a.scatter[i] : i) * a.H + h];
    float* p = a.out + gi * HK + x;
    if (UNIQUE) {
      float4 c = ld4(p);
      c.x = fmaf(g, w.x, c.x); c.y = fmaf(g, w.y, c.y); c.z = fmaf(g, w.z, c.z); c.w = fmaf(g, w.w, c.w);
      st4(p, c);
    } else {
      atomicAdd(p + 0, g * w.x); atomicAdd(p + 1, g * w.y); atomicAdd(p + 2, g * w.z); atomicAdd(p + 3, g * w.w);
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
  for (idx_t i = rb + wave * EPW + slot; i < re; i += 4 * EPW) {
    const idx_t gi = a.gather ? a.gather[i] : i;
    const float g = a.go[(a.scatter ? a.scatter[i] : i) * a.H + h];
    const float4 v = ld4(a.A + gi * HK + x);
    acc.x = fmaf(g, v.x, acc.x); acc.y = fmaf(g, v.y, acc.y); acc.z = fmaf(g, v.z, acc.z); acc.w = fmaf(g, v.w, acc.w);
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
  dim3 grid((unsigned)(ceil_div64(a.num_rows, kChunk) + a.num_segs)), block(kBlock);
  HET_ROWDOT_DISPATCH(a.H * a.K / 4, hipLaunchKernelGGL(HET_rowdot_fwd<LPR>, grid, block, 0, s, a));
  HET_LAUNCH_CHECK("HET_rowdot_fwd");
  return HET_OK;
}

int launch_rowdot_bwd_dx(const RowDotArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  dim3 grid((unsigned)(ceil_div64(a.num_rows, kChunk) + a.num_segs)), block(kBlock);
  if (a.unique_rows) {
    HET_ROWDOT_DISPATCH(a.H * a.K / 4, hipLaunchKernelGGL((HET_rowdot_bwd_dx<LPR, true>), grid, block, 0, s, a));
  } else {
    HET_ROWDOT_DISPATCH(a.H * a.K / 4, hipLaunchKernelGGL((HET_rowdot_bwd_dx<LPR, false>), grid, block, 0, s, a));
  }
  HET_LAUNCH_CHECK("HET_rowdot_bwd_dx");
  return HET_OK;
}

int launch_rowdot_bwd_dw(const RowDotArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  int64_t chunk = ceil_div64(a.num_rows, 2048);  // about 2048 workgroups, one atomic flush each
  if (chunk < 1024) chunk = 1024;
  dim3 grid((unsigned)(ceil_div64(a.num_rows, chunk) + a.num_segs)), block(kBlock);
  HET_ROWDOT_DISPATCH(a.H * a.K / 4, hipLaunchKernelGGL(HET_rowdot_bwd_dw<LPR>, grid, block, 0, s, a, (int)chunk));
  HET_LAUNCH_CHECK("HET_rowdot_bwd_dw");
  return HET_OK;
}
