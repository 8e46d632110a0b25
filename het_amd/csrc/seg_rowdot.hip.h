// Per-head "row dot" kernels: the D_out = 1 case of the segment GEMM (attention vectors
// el = <feat[e,h,:], attn_l[r,h,:]>), which is pure streaming -- no tile GEMM needed.
#pragma once
#include "common.hip.h"

struct RowDotArgs {
  const float* A = nullptr;        // [*, H, K] rows
  const idx_t* gather = nullptr;   // A row of position i (NULL: i)
  const float* W = nullptr;        // [R, H, K] (the [R,H,K,1] weight, or its [R,H,1,K] transpose)
  float* out = nullptr;            // fwd: [*, H];  bwd dX: grad A rows [*, H, K];  bwd dW: [R, H, K]
  const idx_t* scatter = nullptr;  // row of the [*, H] tensor for position i (NULL: i)
  const float* go = nullptr;       // bwd: gradient of the [*, H] output
  const idx_t* seg_ptrs = nullptr;
  int num_segs = 0;
  int64_t num_rows = 0;
  int H = 0, K = 0;
  int unique_rows = 0;             // bwd dX: gather rows are pairwise distinct -> no atomics
  int overwrite = 0;               // bwd dX with unique rows: store instead of read-modify-write
};

bool rowdot_supported(int H, int K);
int launch_rowdot_fwd(const RowDotArgs& a, hipStream_t s);
int launch_rowdot_bwd_dx(const RowDotArgs& a, hipStream_t s);
int launch_rowdot_bwd_dw(const RowDotArgs& a, hipStream_t s);
