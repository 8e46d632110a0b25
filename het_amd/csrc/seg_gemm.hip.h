// Internal interface of the segment (per-relation) GEMM kernels.
#pragma once
#include "common.hip.h"

// C[cs(i), zc + n] (+)= sum_kk scale(i) * A[ga(i), za + kk] * B_r(kk, n)      i in segment r
struct SegGemmArgs {
  const float* A = nullptr;
  int64_t a_ld = 0;           // floats between consecutive A rows
  int64_t a_head_stride = 0;  // added per blockIdx.z
  const idx_t* gather = nullptr;  // A row of position i (NULL: i)
  const float* B = nullptr;
  int64_t b_rel_stride = 0;   // floats between the matrices of consecutive segments
  int64_t b_head_stride = 0;  // added per blockIdx.z (plain layout)
  int b_headcat = 0;          // 1: B_r is [Hc][KA][Dh] and column n = (h, d) -> B[h][kk][d]
  int headcat_d = 1;          // Dh
  float* C = nullptr;
  int64_t c_ld = 0;
  int64_t c_head_stride = 0;
  const idx_t* scatter = nullptr;  // C row of position i (NULL: i)
  int atomic = 0;                  // 0 "=", 1 atomic "+=", 2 plain "+=" (distinct C rows in the launch)
  const idx_t* seg_ptrs = nullptr;  // [num_segs + 1]
  int num_segs = 0;
  int64_t num_rows = 0;
  const float* row_scale = nullptr;  // optional per-row scale: row_scale[sidx(i) * scale_ld + z * scale_zs]
  const idx_t* scale_idx = nullptr;  // sidx(i) = scale_idx[i] (NULL: i)
  int64_t scale_ld = 1, scale_zs = 0;
  int KA = 0, NB = 0, heads_z = 1;
};
int launch_seg_gemm(const SegGemmArgs& a, hipStream_t s);
int launch_seg_gemm_rmw_per_segment(const SegGemmArgs& a, hipStream_t s);  // atomic-free "+=" for segment-wise distinct rows

// dW_r(k, n) += sum_{i in segment r} scale(i) * A[ga(i), za + k] * G[gg(i), zg + n]
struct SegDwArgs {
  const float* A = nullptr;
  int64_t a_ld = 0, a_head_stride = 0;
  const idx_t* gather = nullptr;
  const float* row_scale = nullptr;
  const idx_t* scale_idx = nullptr;
  int64_t scale_ld = 1, scale_zs = 0;
  const float* G = nullptr;
  int64_t g_ld = 0, g_head_stride = 0;
  const idx_t* g_gather = nullptr;
  float* dW = nullptr;
  int64_t dw_rel_stride = 0, dw_head_stride = 0;
  int headcat = 0;  // 1: output column n = (h, d) -> dW[h][k][d]
  int headcat_d = 1;
  const idx_t* seg_ptrs = nullptr;
  int num_segs = 0;
  int64_t num_rows = 0;
  int KA = 0, NB = 0, heads_z = 1;
};
int launch_seg_dw(const SegDwArgs& a, hipStream_t s);
