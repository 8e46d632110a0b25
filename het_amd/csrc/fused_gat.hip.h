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
                        int H, int D, float slope, hipStream_t s);
int gat_backward_grouped(const het_grouping* by_dst, const EdgeView& v, const RowMaps& m, const float* feat,
                         const float* el, const float* er, const float* sum, const float* exp, const float* ret,
                         const float* exp_sorted, const float* gradout, float* grad_feat, float* grad_el,
                         float* grad_er, int H, int D, float slope, hipStream_t s);
