// Segmented row sums over a het_grouping: out[s, :] = sum_{j in segment s} w(j) * in[row(j), :]
#pragma once
#include "grouping.hip.h"

// row(j) = g->p0[j] (payload0 of the grouping); w(j) = scale[g->p1[j]] when scale != NULL, else 1.
// X floats per row, X/4 a power of two <= 64.  `out` has g->S rows and is fully overwritten.
bool segment_sum_supported(int X);
int launch_segment_sum(const het_grouping* g, const float* in, float* out, int X, const float* scale, hipStream_t s);
