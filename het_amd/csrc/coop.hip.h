// Cooperative scalar loads for the gather kernels: lane (head h, d) of a lane group fetches the scalar of edge d of a
// 4-edge step and the DL = D/4 lanes of a head exchange the values with DPP quad broadcasts (DL == 4) or a bpermute.
// See het_amd/csrc/gat_compact.hip and DESIGN.md section 4.1 (the passes were bound by the number of vector-memory
// instructions, not by bytes).
// Hazard: a DPP read of a lane that the current branch has switched off returns the `old` operand (0), not the lane's
// value -- call the broadcasts where all lanes of the quad / head group are active, never inside a per-lane branch.
#pragma once
#include "common.hip.h"

namespace {
// bound_ctrl:1 = a switched-off source lane reads as 0 (what `old = 0` gave) WITHOUT the `v_mov_b32 old, 0` in front of every
// broadcast that the update_dpp form cost; the mov then also folds into its consumer (v_add_f32_dpp, v_fmac_f32_dpp ...).
template <int Q>
__device__ __forceinline__ int quad_bcast_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, Q * 0x55, 0xf, 0xf, true);  // quad_perm:[Q,Q,Q,Q]
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
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 4 lanes of a quad (every lane gets it)
__device__ __forceinline__ float quad_sum(float p) {
  p += dpp_f<0xB1>(p);  // quad_perm:[1,0,3,2]
  p += dpp_f<0x4E>(p);  // quad_perm:[2,3,0,1]
  return p;
}
// Sum over the DL lanes of a head (DL a power of two, heads aligned to DL lanes); every lane of the head gets the sum.
// xor 1 / xor 2 as quad permutes; then, all lanes of a quad (of 8, of 16) being equal, the mirrors within 8 and 16 lanes
// reach the other half.  VALU only: __shfl_xor compiles to ds_bpermute_b32 (the LDS crossbar).
template <int DL>
__device__ __forceinline__ float head_sum(float p) {
  if constexpr (DL >= 2) p += dpp_f<0xB1>(p);   // quad_perm [1,0,3,2]
  if constexpr (DL >= 4) p += dpp_f<0x4E>(p);   // quad_perm [2,3,0,1]
  if constexpr (DL >= 8) p += dpp_f<0x141>(p);  // row_half_mirror
  if constexpr (DL >= 16) p += dpp_f<0x140>(p); // row_mirror
  if constexpr (DL >= 32) p += __shfl_xor(p, 16);
  return p;
}
__device__ __forceinline__ float fast_leaky_exp(float z, float slope) { return __expf(z > 0.f ? z : slope * z); }
}  // namespace
