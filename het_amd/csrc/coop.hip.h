// Cooperative scalar loads for the gather kernels: lane (head h, d) of a lane group fetches the scalar of edge d of a
// 4-edge step and the DL = D/4 lanes of a head exchange the values with DPP quad broadcasts (DL == 4) or a bpermute.
// See het_amd/csrc/gat_compact.hip and DESIGN.md section 4.1 (the passes were bound by the number of vector-memory
// instructions, not by bytes).
// Hazard: a DPP read of a lane that the current branch has switched off returns the `old` operand (0), not the lane's
// value -- call the broadcasts where all lanes of the quad / head group are active, never inside a per-lane branch.
#pragma once
#include "common.hip.h"

namespace {
template <int Q>
__device__ __forceinline__ int quad_bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, Q * 0x55, 0xf, 0xf, false);  // quad_perm:[Q,Q,Q,Q]
}
template <int DL>
__device__ __forceinline__ int head_bcast_i(int v, int q, int lane) {
  if constexpr (DL == 4) {
    switch (q) {
      case 0: return quad_bcast_i<0>(v);
      case 1: return quad_bcast_i<1>(v);
      case 2: return quad_bcast_i<2>(v);
      default: return quad_bcast_i<3>(v);
    }
  } else {
    return __shfl(v, (lane & ~(DL - 1)) | q, 64);
  }
}
template <int DL>
__device__ __forceinline__ float head_bcast(float v, int q, int lane) {
  return __int_as_float(head_bcast_i<DL>(__float_as_int(v), q, lane));
}
__device__ __forceinline__ float fast_leaky_exp(float z, float slope) { return __expf(z > 0.f ? z : slope * z); }
}  // namespace
