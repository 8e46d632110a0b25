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
  int rmw = 0;                     // one-head bwd dX: the launch's rows hit distinct output rows -> plain read-modify-write
};

bool rowdot_supported(int H, int K);
int launch_rowdot_fwd(const RowDotArgs& a, hipStream_t s);
int launch_rowdot_bwd_dx(const RowDotArgs& a, hipStream_t s);
int launch_rowdot_bwd_dw(const RowDotArgs& a, hipStream_t s);

// One input head shared by H weight heads, D_out = 1 (the reference's --multiply_among_weights_first_flag:
// er = x[dst] . (W . attn_r), weight [R,H,K,1]):  out[s_i, h] = sum_k A[g_i, k] * W[r, h, k]
//   bwd dX: grad_A[g_i, :] += sum_h go[s_i, h] * W[r, h, :]      (atomics: rows are shared between positions)
//   bwd dW: dW[r, h, :]    += sum_i go[s_i, h] * A[g_i, :]
// A rows are [*, K]; same RowDotArgs, H <= 8, K/4 a power of two <= 64.
bool rowdot1h_supported(int H, int K);
int launch_rowdot1h_fwd(const RowDotArgs& a, hipStream_t s);
int launch_rowdot1h_bwd_dx(const RowDotArgs& a, hipStream_t s);
int launch_rowdot1h_bwd_dw(const RowDotArgs& a, hipStream_t s);
