// Node-major sum of row x weight products:   out[n, :] = SUM_s rows_s[map_s[n], :] . W_s     for the nodes of a list.
//
// The input gradient of a layer whose input feeds several projections is such a sum: every projection s contributes the
// gradient of ITS output rows (rows_s: one row per (relation, node) pair, per destination, per node ...) times its transposed
// weight to the node the row belongs to.  The reference forms it projection by projection with float atomics into the shared
// [N,K] gradient (backward_rgnn_relational_matmul, OpExport/RGNNOps.inc.h:946-1010 -> RGNN/my_shmem_sgemm_func.cu.h:711-776;
// for HGT four times per layer: HGT/models.py:159-262 through hrt/python/backend/rgnn_layers_and_funcs.py:52-70); round 3 of
// this library ran HGT's as one read-modify-write launch per relation on 128-wide rows (1.41 ms per step on ogbn-mag) while
// the RGAT layer already had the node-major form (node_gemm.hip: HET_node_dx, which this file generalises -- sources are a list
// of {row pointer, row stride, node -> row map, weight}, no RGAT-specific terms).  One pass over the NODES: a 32-node tile
// collects the rows of every source that has one for any of its nodes in one MFMA accumulator tile and stores the output row
// once; the weights of the launch stay in LDS.  A wide row ([k' | m] of HGT's folded source projection: 2X floats) enters as
// two sources -- two halves of the same rows, each with its half of the weight.
#include <stdlib.h>

#include "common.hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMaxSrc = 9;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

struct SumArgs {
  int64_t n_begin, n_end;        // positions of `order` (or node ids) of this launch
  int64_t N;                     // nodes = rows of out = length of every map
  int S;                         // sources
  const float* rows[kMaxSrc];    // first float of source s's row 0 (a column offset into a wider row is part of the pointer)
  int64_t stride[kMaxSrc];       // floats between consecutive rows of source s
  const int32_t* map[kMaxSrc];   // [N] row of node n in source s, -1 = none; NULL: row = n for n < ident_rows[s]
  int64_t ident_rows[kMaxSrc];
  const float* wt[kMaxSrc];      // [KS][XO] row-major
  const int32_t* order;          // [>= n_end] node at position p, or NULL (node p)
  float* out;                    // [N, XO]
};

// Workgroup = WAVES independent waves sharing the S weights in LDS; a wave walks 32-node tiles (grid-stride), loads the rows of
// every PRESENT source coalesced (KS/4 lanes x float4 per row) into its LDS tile, reads them back as MFMA A fragments and
// multiplies them into the same XO/32 accumulators; the rows of the next present source are in flight during the MFMAs.
template <int KS, int NO, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void HET_node_rows_sum(SumArgs a) {
  constexpr int XO = NO * 32, KH = KS / 2;
  constexpr int LD = (KS > XO ? KS : XO) + 4;
  constexpr int LPRA = KS / 4, RPIA = 64 / LPRA, NITA = 32 / RPIA;
  constexpr int LPRC = XO / 4, RPIC = 64 / LPRC, NITC = 32 / RPIC;
  constexpr bool PAIRED = NO == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = a.S;
  float* Bs = smem;                                                   // [S][KS][XO]
  float* Ws = Bs + S * KS * XO + wave * (32 * LD + (S + 1) * 32);     // wave-private tile
  int* idsL = reinterpret_cast<int*>(Ws + 32 * LD);                   // [S][32] row of every node of the tile in source s, -1 = none
  int* idsN = idsL + S * 32;                                          // [32] node of every row of the tile
  for (int s = 0; s < S; ++s)
    for (int e = tid; e < KS * XO; e += WAVES * 64) Bs[s * KS * XO + e] = a.wt[s][e];
  __syncthreads();

  const int row = lane & 31, half = lane >> 5;
  const int ra = lane / LPRA, ca = (lane % LPRA) * 4;
  const int rc = lane / LPRC, cc = (lane % LPRC) * 4;
  const int64_t tiles = (a.n_end - a.n_begin + 31) / 32, stride = (int64_t)gridDim.x * WAVES;
  int64_t t = (int64_t)blockIdx.x * WAVES + wave;
  if (t >= tiles) return;
  int mcur[kMaxSrc];
  int ncur = 0;
  auto load_maps = [&](int64_t tt) {
    const int64_t pos = a.n_begin + tt * 32 + row;
    const int64_t pc = pos < a.n_end ? pos : a.n_end - 1;
    const int64_t nc = a.order ? a.order[pc] : pc;
    ncur = (int)nc;
#pragma unroll
    for (int s = 0; s < kMaxSrc; ++s) {
      mcur[s] = -1;
      if (s < S) mcur[s] = a.map[s] ? a.map[s][nc] : (nc < a.ident_rows[s] ? (int)nc : -1);
    }
  };
  load_maps(t);
  for (; t < tiles; t += stride) {
    const int64_t nb = a.n_begin + t * 32;
    unsigned mask = 0;
    {
      const bool nv = nb + row < a.n_end;
      idsN[row] = nv ? ncur : -1;
#pragma unroll
      for (int s = 0; s < kMaxSrc; ++s) {
        if (s < S) {
          const int id = nv ? mcur[s] : -1;
          idsL[s * 32 + row] = id;
          if (__ballot(id >= 0)) mask |= 1u << s;
        }
      }
    }
    if (t + stride < tiles) load_maps(t + stride);  // consumed one tile later

    f32x16 acc[NO];
#pragma unroll
    for (int nt = 0; nt < NO; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    float4 areg[NITA];
    auto issue = [&](int s) {
      const float* base = a.rows[s];
      const int64_t rs = a.stride[s];
#pragma unroll
      for (int it = 0; it < NITA; ++it) {
        const int id = idsL[s * 32 + it * RPIA + ra];
        areg[it] = ld4(base + (int64_t)(id < 0 ? 0 : id) * rs + ca);
      }
    };
    int s = mask ? __ffs(mask) - 1 : -1;
    if (s >= 0) issue(s);
    while (s >= 0) {
#pragma unroll
      for (int it = 0; it < NITA; ++it) {
        const int id = idsL[s * 32 + it * RPIA + ra];
        st4(&Ws[(it * RPIA + ra) * LD + ca], id >= 0 ? areg[it] : make_float4(0.f, 0.f, 0.f, 0.f));
      }
      const unsigned rest = mask & ~((2u << s) - 1u);
      const int sn = rest ? __ffs(rest) - 1 : -1;
      if (sn >= 0) issue(sn);
      float af[KH];
#pragma unroll
      for (int q = 0; q < KH / 4; ++q) {
        const float4 v = ld4(&Ws[row * LD + half * KH + q * 4]);
        af[4 * q + 0] = v.x; af[4 * q + 1] = v.y; af[4 * q + 2] = v.z; af[4 * q + 3] = v.w;
      }
      const float* B = Bs + s * KS * XO;
#pragma unroll
      for (int q = 0; q < KH; ++q) {
        if (PAIRED) {
          const float2 b2 = *reinterpret_cast<const float2*>(&B[(half * KH + q) * XO + 2 * row]);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], b2.x, acc[0], 0, 0, 0);
          acc[NO - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], b2.y, acc[NO - 1], 0, 0, 0);
        } else {
#pragma unroll
          for (int nt = 0; nt < NO; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], B[(half * KH + q) * XO + nt * 32 + row], acc[nt], 0, 0, 0);
        }
      }
      s = sn;
    }
    // epilogue: transpose through the wave's LDS tile, whole 16-byte pieces per output row (a node without any row gets zeros)
#pragma unroll
    for (int nt = 0; nt < NO; ++nt)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        Ws[((reg & 3) + 8 * (reg >> 2) + 4 * half) * LD + (PAIRED ? 2 * row + nt : nt * 32 + row)] = acc[nt][reg];
#pragma unroll
    for (int it = 0; it < NITC; ++it) {
      const int64_t node = idsN[it * RPIC + rc];
      const float4 v = ld4(&Ws[(it * RPIC + rc) * LD + cc]);
      if (node >= 0) st4(a.out + node * XO + cc, v);
    }
  }
}

template <int KS, int NO>
size_t lds_for(int S, int waves) {
  constexpr int XO = NO * 32, LD = (KS > XO ? KS : XO) + 4;
  return sizeof(float) * ((size_t)S * KS * XO + (size_t)waves * (32 * LD + (S + 1) * 32));
}

template <int KS, int NO>
int launch_sum(const SumArgs& a, hipStream_t s) {
  const int64_t tiles = (a.n_end - a.n_begin + 31) / 32;
  const size_t limit = 160 * 1024;
  HET_KTIME("HET_node_rows_sum", s);
  if (lds_for<KS, NO>(a.S, 8) <= limit) {
    const size_t lds = lds_for<KS, NO>(a.S, 8);
    int64_t gx = (tiles + 8 * 4 - 1) / (8 * 4);  // ~4 tiles per wave: the weights are staged once per workgroup
    if (gx < 1) gx = 1;
    HET_HIP(hipFuncSetAttribute((const void*)HET_node_rows_sum<KS, NO, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_node_rows_sum<KS, NO, 8>), dim3((unsigned)gx), dim3(512), lds, s, a);
  } else {
    const size_t lds = lds_for<KS, NO>(a.S, 4);
    HET_REQUIRE(lds <= limit, "het_node_rows_matmul_sum: the weights of %d sources do not fit the LDS (het_node_rows_matmul_sum_ok)", a.S);
    int64_t gx = (tiles + 4 * 4 - 1) / (4 * 4);
    if (gx < 1) gx = 1;
    HET_HIP(hipFuncSetAttribute((const void*)HET_node_rows_sum<KS, NO, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_node_rows_sum<KS, NO, 4>), dim3((unsigned)gx), dim3(256), lds, s, a);
  }
  HET_LAUNCH_CHECK("HET_node_rows_sum");
  return HET_OK;
}

size_t lds_any(int S, int64_t KS, int64_t XO, int waves) {
  if (KS == 64) return XO == 64 ? lds_for<64, 2>(S, waves) : lds_for<64, 1>(S, waves);
  return XO == 64 ? lds_for<32, 2>(S, waves) : lds_for<32, 1>(S, waves);
}

}  // namespace

extern "C" int het_node_rows_matmul_sum_ok(int64_t num_sources, int64_t KS, int64_t XO) {
  if (!(num_sources >= 1 && num_sources <= kMaxSrc && (KS == 32 || KS == 64) && (XO == 32 || XO == 64))) return 0;
  return lds_any((int)num_sources, KS, XO, 4) <= 160 * 1024 ? 1 : 0;
}

extern "C" int het_node_rows_matmul_sum(int64_t n_begin, int64_t n_end, int64_t num_nodes, int64_t num_sources,
                                        const float* const* rows, const int64_t* row_strides, const int32_t* const* maps,
                                        const int64_t* ident_rows, const float* const* weights_t, float* out, int64_t KS,
                                        int64_t XO, const int32_t* node_order, het_stream stream) {
  const char* op = "het_node_rows_matmul_sum";
  HET_REQUIRE(0 <= n_begin && n_begin <= n_end && n_end <= num_nodes && num_nodes < (1ll << 31), "%s: bad node range", op);
  HET_REQUIRE(het_node_rows_matmul_sum_ok(num_sources, KS, XO), "%s: unsupported shape: %lld sources of %lld -> %lld floats", op,
              (long long)num_sources, (long long)KS, (long long)XO);
  if (n_begin == n_end) return HET_OK;
  HET_REQUIRE(rows && row_strides && maps && ident_rows && weights_t && out, "%s: null argument", op);
  SumArgs a{};
  a.n_begin = n_begin; a.n_end = n_end; a.N = num_nodes; a.S = (int)num_sources; a.order = node_order; a.out = out;
  for (int s = 0; s < a.S; ++s) {
    HET_REQUIRE(rows[s] && weights_t[s] && row_strides[s] >= KS && (row_strides[s] & 3) == 0 && ((uintptr_t)rows[s] & 15) == 0,
                "%s: source %d: null pointer, row stride below the row width, or rows not 16-byte aligned", op, s);
    HET_REQUIRE(maps[s] || (ident_rows[s] >= 0 && ident_rows[s] <= num_nodes), "%s: source %d: neither a map nor a valid identity range", op, s);
    a.rows[s] = rows[s]; a.stride[s] = row_strides[s]; a.map[s] = maps[s]; a.ident_rows[s] = ident_rows[s]; a.wt[s] = weights_t[s];
  }
  hipStream_t st = (hipStream_t)stream;
  if (KS == 64) return XO == 64 ? launch_sum<64, 2>(a, st) : launch_sum<64, 1>(a, st);
  return XO == 64 ? launch_sum<32, 2>(a, st) : launch_sum<32, 1>(a, st);
}
