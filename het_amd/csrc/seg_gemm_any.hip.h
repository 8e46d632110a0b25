// Row GEMM / weight-gradient launchers that take the MFMA argument structs and pick the kernel: the matrix-core kernels
// for their shapes (K, X in {32, 64, 128, 256}), else the LDS-tiled FMA kernels of seg_gemm.hip with the same semantics.
// The grouped dataflows (segment sum -> GEMMs on the S distinct rows, distinct rows -> broadcast) are worth far more
// than the choice of GEMM kernel, so small or odd feature widths (an 8- or 16-wide output layer: the reference CLI's
// default --num_classes 8) keep them.
#pragma once
#include "seg_gemm.hip.h"
#include "seg_gemm_mfma.hip.h"

inline int launch_rows_gemm(const MfmaGemmArgs& m, hipStream_t s) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(m.A) | reinterpret_cast<uintptr_t>(m.C)) & 15) == 0;
  if (mfma_shape_supported(m.K, m.X) && aligned) return launch_seg_gemm_mfma(m, s);
  HET_REQUIRE(!m.dot_out, "row GEMM: the dot epilogue needs an MFMA shape");
  SegGemmArgs a;
  a.A = m.A; a.a_ld = m.a_ld; a.gather = m.gather; a.row_scale = m.row_scale; a.scale_idx = m.scale_idx;
  a.B = m.B; a.b_rel_stride = m.b_rel_stride; a.C = m.C; a.c_ld = m.c_ld; a.scatter = m.scatter; a.atomic = m.atomic;
  a.seg_ptrs = m.seg_ptrs; a.num_segs = m.num_segs; a.num_rows = m.num_rows;
  if (m.b_headcat == 2) {  // block diagonal: one z slice per head
    const int Dh = m.headcat_d, Kh = m.blockdiag_k;
    a.KA = Kh; a.NB = Dh; a.heads_z = m.X / Dh;
    a.a_head_stride = Kh; a.b_head_stride = (int64_t)Kh * Dh; a.c_head_stride = Dh;
  } else {
    a.KA = m.K; a.NB = m.X; a.heads_z = 1;
    if (m.b_headcat == 1) { a.b_headcat = 1; a.headcat_d = m.headcat_d; }
  }
  if (m.atomic == 2) return launch_seg_gemm_rmw_per_segment(a, s);  // (segment-wise distinct C rows: the caller's contract)
  return launch_seg_gemm(a, s);
}

// C[scatter[i]] += A[i] . B[r] for lists whose rows are DISTINCT inside every segment (the (relation, key) segments of a
// grouping, a unique (relation, node) list): segment by segment with plain read-modify-write instead of float atomics
// when the segments are few (launches are ordered on the stream); else atomics.
inline int launch_rows_gemm_add_unique(const MfmaGemmArgs& m, hipStream_t s) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(m.A) | reinterpret_cast<uintptr_t>(m.C)) & 15) == 0;
  if (mfma_shape_supported(m.K, m.X) && aligned && m.num_segs <= kRmwMaxSegments && m.K <= 128 && m.X <= 128 && !m.dot_w && !m.bias)
    return launch_seg_gemm_mfma_rmw_per_segment(m, s);
  MfmaGemmArgs a = m;
  a.atomic = m.num_segs <= kRmwMaxSegments ? 2 : 1;  // the any-shape kernel has the same read-modify-write form
  if (a.atomic == 2 && mfma_shape_supported(m.K, m.X) && aligned) a.atomic = 1;  // (an MFMA shape that failed the other conditions)
  return launch_rows_gemm(a, s);
}

inline int launch_rows_dw(const MfmaDwArgs& m, hipStream_t s) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(m.A) | reinterpret_cast<uintptr_t>(m.G)) & 15) == 0;
  if (mfma_dw_supported(m.K, m.X) && aligned) return launch_seg_dw_mfma(m, s);
  SegDwArgs w;
  w.A = m.A; w.a_ld = m.a_ld; w.gather = m.gather; w.row_scale = m.row_scale; w.scale_idx = m.scale_idx;
  w.G = m.G; w.g_ld = m.g_ld; w.g_gather = m.g_gather; w.dW = m.dW; w.dw_rel_stride = m.dw_rel_stride;
  w.seg_ptrs = m.seg_ptrs; w.num_segs = m.num_segs; w.num_rows = m.num_rows;
  if (m.headcat == 2) {
    const int Dh = m.headcat_d, Kh = m.blockdiag_k;
    w.KA = Kh; w.NB = Dh; w.heads_z = m.X / Dh;
    w.a_head_stride = Kh; w.g_head_stride = Dh; w.dw_head_stride = (int64_t)Kh * Dh;
  } else {
    w.KA = m.K; w.NB = m.X; w.heads_z = 1;
    if (m.headcat == 1) { w.headcat = 1; w.headcat_d = m.headcat_d; }
  }
  return launch_seg_dw(w, s);
}
