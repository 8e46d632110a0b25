// Generic (any-shape) segment GEMM kernels: LDS-tiled fp32 FMA, 256-thread
// workgroups (4 waves of 64), 64x64 output tile, 4x4 outputs per thread.
// These cover every shape the ops accept (K = 16 on AIFB, D = 1 attention
// vectors, dk = 8 HGT heads ...).  The MFMA fast path for the 64-wide
// feature GEMMs lives in seg_gemm_mfma.hip and is chosen by the launchers
// in capi.hip when the shape allows.
#include "seg_gemm.hip.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16, PAD = 4;

// MODE 0: C row = product; 1: atomic add; 2: plain read-modify-write (rows of the launch are distinct C rows)
template <int MODE>
__global__ __launch_bounds__(256) void HET_seg_gemm_generic(SegGemmArgs a) {
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, TM, blockIdx.x, r, rb, re)) return;
  const int z = blockIdx.z, n0 = blockIdx.y * TN, tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float As[TK][TM + PAD];
  __shared__ __attribute__((aligned(16))) float Bs[TK][TN + PAD];
  __shared__ idx_t a_rows[TM], c_rows[TM];
  __shared__ float scl[TM];
  if (tid < TM) {
    idx_t i = rb + tid;
    bool v = i < re;
    a_rows[tid] = v ? (a.gather ? a.gather[i] : i) : -1;
    c_rows[tid] = v ? (a.scatter ? a.scatter[i] : i) : -1;
    scl[tid] = (v && a.row_scale) ? a.row_scale[(a.scale_idx ? a.scale_idx[i] : i) * a.scale_ld + blockIdx.z * a.scale_zs] : 1.f;
  }
  __syncthreads();
  const float* __restrict__ Bm = a.B + (int64_t)r * a.b_rel_stride + (a.b_headcat ? 0 : (int64_t)z * a.b_head_stride);
  const float* __restrict__ Ab = a.A + (int64_t)z * a.a_head_stride;
  const int KA = a.KA, NB = a.NB, Dh = a.headcat_d;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int tx = tid & 15, ty = tid >> 4;
  for (int k0 = 0; k0 < KA; k0 += TK) {
    {
      const int row = tid >> 2;
      const idx_t ar = a_rows[row];
      const float sc = scl[row];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = (tid & 3) * 4 + j;
        float v = 0.f;
        if (ar >= 0 && k0 + kk < KA) v = Ab[ar * a.a_ld + k0 + kk] * sc;
        As[kk][row] = v;
      }
    }
    {
      const int kk = tid >> 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = (tid & 15) * 4 + j, gn = n0 + n;
        float v = 0.f;
        if (k0 + kk < KA && gn < NB) {
          if (a.b_headcat) {
            const int h = gn / Dh, d = gn - h * Dh;
            v = Bm[(int64_t)h * KA * Dh + (int64_t)(k0 + kk) * Dh + d];
          } else {
            v = Bm[(int64_t)(k0 + kk) * NB + gn];
          }
        }
        Bs[kk][n] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      const float4 av = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
      const float aa[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* __restrict__ Cb = a.C + (int64_t)z * a.c_head_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const idx_t cr = c_rows[ty * 4 + i];
    if (cr < 0) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn < NB) {
        float* p = Cb + cr * a.c_ld + gn;
        if (MODE == 1) atomicAdd(p, acc[i][j]); else if (MODE == 2) *p += acc[i][j]; else *p = acc[i][j];
      }
    }
  }
}

constexpr int DWC = 16;  // rows per staged chunk

__global__ __launch_bounds__(256) void HET_seg_dw_generic(SegDwArgs a, int rows_per_block, int nkt) {
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, rows_per_block, blockIdx.x, r, rb, re)) return;
  const int z = blockIdx.z, tid = threadIdx.x;
  const int k0 = (blockIdx.y % nkt) * 64, n0 = (blockIdx.y / nkt) * 64;
  __shared__ __attribute__((aligned(16))) float As[DWC][64 + PAD];
  __shared__ __attribute__((aligned(16))) float Gs[DWC][64 + PAD];
  const int KA = a.KA, NB = a.NB;
  const float* __restrict__ Ab = a.A + (int64_t)z * a.a_head_stride;
  const float* __restrict__ Gb = a.G + (int64_t)z * a.g_head_stride;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int tx = tid & 15, ty = tid >> 4;
  const int lrow = tid >> 4, lc = (tid & 15) * 4;
  for (idx_t c = rb; c < re; c += DWC) {
    const idx_t i = c + lrow;
    const bool valid = i < re;
    idx_t ar = 0, gr = 0;
    float sc = 1.f;
    if (valid) {
      ar = a.gather ? a.gather[i] : i;
      gr = a.g_gather ? a.g_gather[i] : i;
      if (a.row_scale) sc = a.row_scale[(a.scale_idx ? a.scale_idx[i] : i) * a.scale_ld + z * a.scale_zs];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + lc + j, nn = n0 + lc + j;
      As[lrow][lc + j] = (valid && kk < KA) ? Ab[ar * a.a_ld + kk] * sc : 0.f;
      Gs[lrow][lc + j] = (valid && nn < NB) ? Gb[gr * a.g_ld + nn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < DWC; ++rr) {
      const float4 av = *reinterpret_cast<const float4*>(&As[rr][ty * 4]);
      const float4 gv = *reinterpret_cast<const float4*>(&Gs[rr][tx * 4]);
      const float aa[4] = {av.x, av.y, av.z, av.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i2][j] = fmaf(aa[i2], gg[j], acc[i2][j]);
    }
    __syncthreads();
  }
  float* __restrict__ out = a.dW + (int64_t)r * a.dw_rel_stride;
  const int Dh = a.headcat_d;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty * 4 + i;
    if (k >= KA) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= NB) continue;
      int64_t off;
      if (a.headcat) {
        const int h = n / Dh, d = n - h * Dh;
        off = (int64_t)h * KA * Dh + (int64_t)k * Dh + d;
      } else {
        off = (int64_t)z * a.dw_head_stride + (int64_t)k * NB + n;
      }
      atomicAdd(out + off, acc[i][j]);
    }
  }
}

}  // namespace

int launch_seg_gemm(const SegGemmArgs& a, hipStream_t s) {
  if (a.num_rows == 0 || a.NB == 0) return HET_OK;
  const int64_t gx = ceil_div64(a.num_rows, TM) + a.num_segs;
  HET_REQUIRE(gx < (1ll << 31), "segment GEMM: too many row tiles (%lld)", (long long)gx);
  dim3 grid((unsigned)gx, (unsigned)ceil_div64(a.NB, TN), (unsigned)a.heads_z), block(256);
  if (a.atomic == 2)
    hipLaunchKernelGGL(HET_seg_gemm_generic<2>, grid, block, 0, s, a);
  else if (a.atomic)
    hipLaunchKernelGGL(HET_seg_gemm_generic<1>, grid, block, 0, s, a);
  else
    hipLaunchKernelGGL(HET_seg_gemm_generic<0>, grid, block, 0, s, a);
  HET_LAUNCH_CHECK("HET_seg_gemm_generic");
  return HET_OK;
}

// C[scatter[i]] += A[i] . B[r] for lists whose rows are distinct inside every segment: segment by segment (launches are
// ordered on the stream) with plain read-modify-write instead of one float atomic per element.
int launch_seg_gemm_rmw_per_segment(const SegGemmArgs& a, hipStream_t s) {
  for (int r = 0; r < a.num_segs; ++r) {
    SegGemmArgs m = a;
    m.seg_ptrs = a.seg_ptrs + r;  // the one segment [seg_ptrs[r], seg_ptrs[r+1]); row indices stay absolute
    m.num_segs = 1;
    m.B = a.B + (int64_t)r * a.b_rel_stride;
    m.atomic = 2;
    if (int rc = launch_seg_gemm(m, s)) return rc;
  }
  return HET_OK;
}

int launch_seg_dw(const SegDwArgs& a, hipStream_t s) {
  if (a.num_rows == 0 || a.NB == 0 || a.KA == 0) return HET_OK;
  // about 2048 row-chunks in flight; each ends with one atomic 64x64 tile flush
  int64_t rpb = ceil_div64(ceil_div64(a.num_rows, 2048), DWC) * DWC;
  if (rpb < 256) rpb = 256;
  const int64_t gx = ceil_div64(a.num_rows, rpb) + a.num_segs;
  const int nkt = (int)ceil_div64(a.KA, 64), nnt = (int)ceil_div64(a.NB, 64);
  dim3 grid((unsigned)gx, (unsigned)(nkt * nnt), (unsigned)a.heads_z), block(256);
  hipLaunchKernelGGL(HET_seg_dw_generic, grid, block, 0, s, a, (int)rpb, nkt);
  HET_LAUNCH_CHECK("HET_seg_dw_generic");
  return HET_OK;
}
