// fp32 MFMA (v_mfma_f32_32x32x2_f32) segment GEMM for the feature-width shapes
// (K, X in {32, 64, 128}; 256 as 128-wide slabs of the weight); everything else goes through seg_gemm.hip.
#pragma once
#include "common.hip.h"

struct MfmaGemmArgs {
  const float* A = nullptr;       // [*, K] rows, gathered
  int64_t a_ld = 0;
  const idx_t* gather = nullptr;  // NULL: identity
  const float* row_scale = nullptr;
  const idx_t* scale_idx = nullptr;
  const float* B = nullptr;       // per segment: plain [K][X], or head-concatenated [Hc][K][Dh]
  int64_t b_rel_stride = 0;
  int b_headcat = 0, headcat_d = 1;  // 0 plain [K][X]; 1 head-concatenated [Hc][K][Dh]; 2 block diagonal [H][Kh][Dh]
  int blockdiag_k = 1;               // Kh for layout 2
  // a launch may cover a K x X window of a larger weight (K or X = 256 are run as 128-wide slabs by the launcher):
  // window origin and the full sizes the layouts above are indexed with (0 = the launch's own K / X)
  int b_k0 = 0, b_n0 = 0, b_kfull = 0, b_xfull = 0;
  float* C = nullptr;             // [*, X]
  int64_t c_ld = 0;
  const idx_t* scatter = nullptr; // NULL: identity
  int atomic = 0;                 // 0: C rows are stored; 1: float atomic adds; 2: non-atomic read-modify-write (the rows of
                                  // the LAUNCH hit distinct C rows -- launch_seg_gemm_mfma_rmw_per_segment)
  const idx_t* seg_ptrs = nullptr;
  int num_segs = 0;
  int64_t num_rows = 0;
  int K = 0, X = 0;
  // optional epilogue (plain stores, b_headcat == 1): dot_out[cs(i), h] = < C[cs(i), h, :], dot_w[r, h, :] >
  const float* dot_w = nullptr;  // [num_segs, X]
  float* dot_out = nullptr;      // [*, X / headcat_d]
  // optional epilogue (plain stores, K, X <= 128): C row += bias[:]   (one [X] vector for all segments)
  const float* bias = nullptr;
};

bool mfma_shape_supported(int K, int X);
int launch_seg_gemm_mfma(const MfmaGemmArgs& a, hipStream_t s);
// C[scatter(i)] += A . B_r with the scatter rows distinct INSIDE every segment (a unique (relation, node) list) but
// shared between segments: one launch per segment (stream order serialises them), each adding with plain
// read-modify-write.  Few segments only (every launch is sized for all rows): callers fall back to atomics above 8.
constexpr int kRmwMaxSegments = 8;
int launch_seg_gemm_mfma_rmw_per_segment(const MfmaGemmArgs& a, hipStream_t s);

// forward projection with one input head: C[scatter(i), (h,d)] = A[gather(i), :] . W[r, h, :, d]
inline bool mfma_fwd_supported(int K, int X) { return mfma_shape_supported(K, X); }
inline int launch_seg_gemm_mfma_fwd(const float* x, int64_t x_ld, const idx_t* gather, const float* W,
                                    int64_t w_rel_stride, int H, int D, float* ret, int64_t ret_ld,
                                    const idx_t* scatter, const idx_t* seg_ptrs, int num_segs, int64_t num_rows,
                                    int K, hipStream_t s) {
  MfmaGemmArgs a;
  a.A = x; a.a_ld = x_ld; a.gather = gather; a.B = W; a.b_rel_stride = w_rel_stride; a.b_headcat = 1; a.headcat_d = D;
  a.C = ret; a.c_ld = ret_ld; a.scatter = scatter; a.seg_ptrs = seg_ptrs; a.num_segs = num_segs; a.num_rows = num_rows;
  a.K = K; a.X = H * D;
  return launch_seg_gemm_mfma(a, s);
}

// dW_r(k, n) += sum_{i in segment r} scale(i) * A[ga(i), k] * G[gg(i), n]   on the matrix cores
// (the MFMA k dimension runs over rows).  Output layout: plain [K][X] per segment, or
// head-concatenated [Hc][K][Dh] with n = (h, d).
struct MfmaDwArgs {
  const float* A = nullptr;
  int64_t a_ld = 0;
  const idx_t* gather = nullptr;
  const float* row_scale = nullptr;
  const idx_t* scale_idx = nullptr;
  const float* G = nullptr;
  int64_t g_ld = 0;
  const idx_t* g_gather = nullptr;
  float* dW = nullptr;
  int64_t dw_rel_stride = 0;
  float* colsum = nullptr;           // [X] += column sums of the G rows of the launch (zeroed by the caller), or NULL
  int headcat = 0, headcat_d = 1;    // 0 plain; 1 head-concatenated; 2 block diagonal (per-head Kh x Dh blocks)
  int blockdiag_k = 1;
  const idx_t* seg_ptrs = nullptr;
  int num_segs = 0;
  int64_t num_rows = 0;
  int K = 0, X = 0;
};
bool mfma_dw_supported(int K, int X);
int launch_seg_dw_mfma(const MfmaDwArgs& a, hipStream_t s);
