// Segmented row sums over a het_grouping: out[s, :] = sum_{j in segment s} w(j) * in[row(j), :]
#pragma once
#include "grouping.hip.h"

// row(j) = g->p0[j] (payload0 of the grouping); w(j) = scale[g->p1[j]] when scale != NULL, else 1; with
// scale_heads = H > 0 the scale is per (row, head): scale[g->p1[j] * H + h] for the X/H floats of head h.
// X floats per row, X/4 a power of two <= 64.  `out` has g->S rows and is fully overwritten.
bool segment_sum_supported(int X);
// also rows of 1 or 2 floats (no scale, no scatter, no second pair): the [E,H] attention terms of RGAT with 1 or 2 heads
bool segment_rows_supported(int X);
// scatter_rows >= 0: `out` has scatter_rows rows and segment s is written to row seg_key[s] instead of row s.
// accumulate: add to `out` instead of overwriting it.  The scale index is payload1, or payload0 without one
// (scale_by_p0: payload0 even when the grouping carries a payload1).  scale_heads == X: one scale per element,
// i.e. out[s, :] = SUM scale[idx(j), :] * in[row(j), :].
int launch_segment_sum(const het_grouping* g, const float* in, float* out, int X, const float* scale, hipStream_t s,
                       int scale_heads = 0, int64_t scatter_rows = -1, int accumulate = 0, int scale_by_p0 = 0,
                       int nt_in = 0,          // nt_in: `in` is read once (an [E, X] stream): non-temporal loads
                       int scale_sorted = 0);  // `scale` is in the grouping's order: scale[j (* H + h)] belongs to sorted rank j
// out[j, :] = values[payload1 of rank j, :] (H floats per entry): a per-edge-id scale brought into the grouping's order once, for
// callers that pass the same scale every step (an edge norm)
int launch_gather_by_p1(const het_grouping* g, const float* values, int H, float* out, hipStream_t s);

// out[p0[j], :] = in[s, :] for every sorted rank j of segment s; optionally a second, narrower pair (X2 <= X/4 floats)
int launch_segment_broadcast(const het_grouping* g, const float* in, float* out, int X, const float* in2, float* out2,
                             int X2, hipStream_t s);

// g->seg_of_rank ([E]: segment of every sorted rank), built on first use and cached in the grouping
int grouping_seg_of_rank(const het_grouping* g, hipStream_t s);
