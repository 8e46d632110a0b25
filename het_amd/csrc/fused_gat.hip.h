// Internal interface between the edge-parallel and the grouped GAT kernels.
#pragma once
#include "edge_view.hip.h"
#include "grouping.hip.h"

int gat_forward_edge(const EdgeView& v, const RowMaps& m, const float* feat, const float* el, const float* er,
                     float* sum, float* exp, float* ret, int H, int D, float slope, hipStream_t s);
int gat_backward_edge(const EdgeView& v, const RowMaps& m, const float* feat, const float* el, const float* er,
                      const float* sum, const float* exp, const float* ret, const float* gradout, float* grad_feat,
                      float* grad_el, float* grad_er, int H, int D, float slope, hipStream_t s);
// by_dst: grouping of the positions by destination with payload0 = eids and
// optionally payload1 = feat row.  Fall back to the edge kernels when the shape
// or the row maps do not fit.
int gat_forward_grouped(const het_grouping* by_dst, const EdgeView& v, const RowMaps& m, const float* feat,
                        const float* el, const float* er, float* sum, float* exp, float* ret, float* exp_sorted,
                        int H, int D, float slope, const float* el_sorted, const float* er_sorted, hipStream_t s);
int gat_backward_grouped(const het_grouping* by_dst, const EdgeView& v, const RowMaps& m, const float* feat,
                         const float* el, const float* er, const float* sum, const float* exp, const float* ret,
                         const float* exp_sorted, const float* gradout, float* grad_feat, float* grad_el,
                         float* grad_er, int H, int D, float slope, const float* fold_w, float* grad_fold_w,
                         float* grad_el_sorted, float* workspace, int64_t workspace_bytes, hipStream_t s);

// Compact kinds (1/3/4): by_srow groups the positions by feat row (payload0 = eids, payload1 = col), by_drow by
// er row (payload0 = eids).  workspace: N*2H + E*H floats.
bool gat_backward_compact_supported(const het_grouping* by_srow, const het_grouping* by_drow, int64_t E,
                                    int64_t n_dst_rows, int H, int D, float slope);
int gat_backward_compact_grouped(const het_grouping* by_srow, const het_grouping* by_drow, const EdgeView& v,
                                 int64_t n_src_rows, int64_t n_dst_rows, const float* feat, const float* sum,
                                 const float* exp, const float* ret, const float* gradout, float* grad_feat,
                                 float* grad_el, float* grad_er, float* workspace, int H, int D, float slope,
                                 const float* fold_w, const idx_t* fold_row_rel_ptrs, hipStream_t s);
