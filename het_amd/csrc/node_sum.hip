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
  const float* bias;             // [XO] added to every output row, or NULL
  int64_t mix;                   // tile walked at step L of the grid-stride loop: (L * mix) % tiles (1: in list order).  A list sorted by
                                 // presence puts the nodes without any row first: in list order every wave stores its empty tiles first
                                 // (matrix cores idle) and multiplies afterwards (stores idle); a stride near 0.618 * tiles, coprime with
                                 // tiles, hands every wave a uniform sample of the classes (RGCN forward pass 0.239 -> 0.225 ms,
                                 // the backward one unchanged; HET_NODE_SUM_MIX=0: A/B)
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
  const float4 bias4 = a.bias ? ld4(a.bias + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
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
      float4 v = ld4(&Ws[(it * RPIC + rc) * LD + cc]);
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (node >= 0) st4(a.out + node * XO + cc, v);
    }
  }
}

// ---- round-4 form: twice the waves per CU -------------------------------------------------------------------------------
// The kernel above keeps a [32][max(KS,XO)+4] tile + the row ids per wave in LDS (9.3 KB): beside 64-144 KB of weights that is 8
// waves per CU, two per SIMD, and a wave's tile is a serial chain (row gather -> LDS -> 64 MFMAs -> transposed store), so the
// matrix cores idle through every gather: 0.29 ms for a pass whose rows stream in 0.16 ms and whose MFMAs take 0.07 (RGCN, ogbn-mag).
// Here a wave stages HALF a row width at a time (32 columns: [32][36] floats = 4.5 KB), keeps the row ids in registers (lane
// l: the row of tile node l & 31; the loading lane fetches its row's id with a shuffle) and stores the accumulators straight
// from registers (PAIRED layout: lane (i, kk) holds columns 2i, 2i+1 of a tile row per accumulator register -- the 32 lanes of
// a half write one whole 256-byte output row), so 16 waves fit beside 64-88 KB of weights: four per SIMD hide each other's gathers.
// Measured (exp/node_sum_probe.py, ogbn-mag, 4 relations of 64 x 64): forward pass 0.259 -> 0.227 ms, backward 0.355 -> 0.326; the
// counters of the 16-wave form: MFMA busy 29 % of the launch, waves waiting on memory 35 % of their cycles.  What is left is the rate
// of scattered 256-byte rows: the nodes WITHOUT rows alone (bias stores to 1.17 M scattered rows) take 0.076 ms = 3.9 TB/s where a
// dense fill of as many bytes runs at 6.9 TB/s; the pass as a whole moves its 0.81 GB at 3.6 TB/s.  Looking the maps up by position
// of the sorted list instead of by node id (coalesced instead of one random 4-byte gather per node and map) changed nothing
// measurable -- the [R, N] int32 maps sit in the L2 / Infinity Cache -- and was removed again.
// k mapping of a 32-column phase p: MFMA step q of lane (i, kk) multiplies A[i][32p + 16kk + q] with B[32p + 16kk + q][col].
template <int KS, int NO>
__global__ __launch_bounds__(1024) void HET_node_rows_sum_w16(SumArgs a) {
  constexpr int XO = NO * 32;
  constexpr int LDA = 36;                                   // staged row: 32 floats + 4 (conflict-free 16-byte reads down a column of rows)
  constexpr int LPRA = KS / 4, RPIA = 64 / LPRA, NITA = 32 / RPIA;
  constexpr int PH = KS / 32;                               // phases per source
  constexpr bool PAIRED = NO == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6;
  const int S = a.S;
  float* Bs = smem;                                         // [S][KS][XO]
  float* As = Bs + S * KS * XO + wave * (32 * LDA);         // wave-private [32][LDA]
  for (int s = 0; s < S; ++s) {
    const float4* src = reinterpret_cast<const float4*>(a.wt[s]);
    float4* dst = reinterpret_cast<float4*>(Bs + s * KS * XO);
    for (int e = tid; e < KS * XO / 4; e += blockDim.x) dst[e] = src[e];
  }
  __syncthreads();

  const int i = lane & 31, kk = lane >> 5;
  const int ra = lane / LPRA, ca = lane % LPRA;             // loading lane: row ra (+ RPIA per iteration), 16-byte chunk ca
  const int64_t tiles = (a.n_end - a.n_begin + 31) / 32, stride = (int64_t)gridDim.x * waves;
  int64_t t = (int64_t)blockIdx.x * waves + wave;
  if (t >= tiles) return;
  float2 bias2 = make_float2(0.f, 0.f);
  if (a.bias) {
    if (PAIRED) bias2 = *reinterpret_cast<const float2*>(a.bias + 2 * i);
    else bias2.x = a.bias[i];
  }
  int mnext[kMaxSrc];
  int nnext = -1;
  auto load_maps = [&](int64_t tt) {
    const int64_t pos = a.n_begin + tt * 32 + i;
    const bool nv = pos < a.n_end;
    const int64_t pc = nv ? pos : a.n_end - 1;
    const int64_t nc = a.order ? a.order[pc] : pc;
    nnext = nv ? (int)nc : -1;
#pragma unroll
    for (int s = 0; s < kMaxSrc; ++s) {
      mnext[s] = -1;
      if (s < S && nv) mnext[s] = a.map[s] ? a.map[s][nc] : (nc < a.ident_rows[s] ? (int)nc : -1);
    }
  };
  const int64_t mix = a.mix;
  load_maps(t * mix % tiles);
  for (; t < tiles; t += stride) {
    int ids[kMaxSrc];
    const int node = nnext;
    unsigned mask = 0;
#pragma unroll
    for (int s = 0; s < kMaxSrc; ++s) {
      ids[s] = mnext[s];
      if (s < S && __ballot(ids[s] >= 0)) mask |= 1u << s;
    }
    if (t + stride < tiles) load_maps((t + stride) * mix % tiles);  // consumed one tile later

    f32x16 acc[NO];
#pragma unroll
    for (int nt = 0; nt < NO; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    float4 areg[NITA];
    auto issue = [&](int s) {
      // (ids[] is indexed with a wave-uniform s: a switch keeps it in registers)
      int idl = -1;
#pragma unroll
      for (int q = 0; q < kMaxSrc; ++q)
        if (q == s) idl = ids[q];
      const float* base = a.rows[s];
      const int64_t rs = a.stride[s];
#pragma unroll
      for (int it = 0; it < NITA; ++it) {
        const int id = __shfl(idl, it * RPIA + ra);
        const float4 v = ld4(base + (int64_t)(id < 0 ? 0 : id) * rs + ca * 4);
        areg[it] = id >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    int s = mask ? __ffs(mask) - 1 : -1;
    if (s >= 0) issue(s);
    while (s >= 0) {
      const unsigned rest = mask & ~((2u << s) - 1u);
      const int sn = rest ? __ffs(rest) - 1 : -1;
      const float* B = Bs + s * KS * XO;
#pragma unroll
      for (int p = 0; p < PH; ++p) {
        // stage the 32 columns of phase p: the loading lanes whose chunk lies in them
        if (PH == 1 || (ca >> 3) == p) {
#pragma unroll
          for (int it = 0; it < NITA; ++it) st4(&As[(it * RPIA + ra) * LDA + (ca & 7) * 4], areg[it]);
        }
        wave_lds_fence();  // (half the lanes stage, all lanes read: common.hip.h)
        if (p == PH - 1 && sn >= 0) issue(sn);  // the row registers are free: the next source's rows fly during the MFMAs below
        float af[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = ld4(&As[i * LDA + kk * 16 + q * 4]);
          af[4 * q + 0] = v.x; af[4 * q + 1] = v.y; af[4 * q + 2] = v.z; af[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int k = p * 32 + kk * 16 + q;
          if (PAIRED) {
            const float2 b2 = *reinterpret_cast<const float2*>(&B[k * XO + 2 * i]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], b2.x, acc[0], 0, 0, 0);
            acc[NO - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], b2.y, acc[NO - 1], 0, 0, 0);
          } else {
#pragma unroll
            for (int nt = 0; nt < NO; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], B[k * XO + nt * 32 + i], acc[nt], 0, 0, 0);
          }
        }
      }
      s = sn;
    }
    // epilogue: accumulator register `reg` of lane (i, kk) is tile row (reg & 3) + 8 * (reg >> 2) + 4 * kk
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int64_t n = __shfl(node, (reg & 3) + 8 * (reg >> 2) + 4 * kk);
      if (n >= 0) {
        if (PAIRED) {
          *reinterpret_cast<float2*>(a.out + n * XO + 2 * i) = make_float2(acc[0][reg] + bias2.x, acc[NO - 1][reg] + bias2.y);
        } else {
#pragma unroll
          for (int nt = 0; nt < NO; ++nt) a.out[n * XO + nt * 32 + i] = acc[nt][reg] + (a.bias ? a.bias[nt * 32 + i] : 0.f);
        }
      }
    }
  }
}

template <int KS, int NO>
int launch_sum_w16(const SumArgs& a, hipStream_t s, bool* done) {
  constexpr int XO = NO * 32;
  const size_t limit = het_lds_budget(), wbytes = sizeof(float) * (size_t)a.S * KS * XO, per_wave = sizeof(float) * 32 * 36;
  *done = false;
  if (wbytes + 8 * per_wave > limit) return HET_OK;  // fewer than 8 waves: the tile form above does as well
  // (HET_NODE_SUM_WAVES: A/B -- 16 waves of 128 VGPRs are a CU's whole register file, nothing runs beside such a workgroup)
  static const int max_waves = [] { const char* v = getenv("HET_NODE_SUM_WAVES"); const int w = v ? atoi(v) : 16; return w < 4 ? 4 : (w > 16 ? 16 : w); }();
  int waves = (int)((limit - wbytes) / per_wave);
  if (waves > max_waves) waves = max_waves;
  const int64_t tiles = (a.n_end - a.n_begin + 31) / 32;
  int64_t gx = (tiles + waves - 1) / waves;
  const int64_t cus = het_num_cus();
  if (gx > cus) gx = cus;  // one workgroup per CU (the weights are staged once), its waves walk the tiles grid-stride
  if (gx < 1) gx = 1;
  const size_t lds = wbytes + (size_t)waves * per_wave;
  SumArgs b = a;
  static const bool no_mix = getenv("HET_NODE_SUM_MIX") && atoi(getenv("HET_NODE_SUM_MIX")) == 0;  // A/B: tiles in list order
  b.mix = 1;
  if (!no_mix && a.order && tiles > 4 * gx * waves) {
    auto gcd = [](int64_t x, int64_t y) { while (y) { const int64_t r = x % y; x = y; y = r; } return x; };
    int64_t m = (int64_t)(0.6180339887 * (double)tiles) | 1;
    while (m > 1 && gcd(m, tiles) != 1) m -= 2;
    b.mix = m < 1 ? 1 : m;
  }
  HET_KTIME("HET_node_rows_sum", s);
  HET_HIP(hipFuncSetAttribute((const void*)HET_node_rows_sum_w16<KS, NO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((HET_node_rows_sum_w16<KS, NO>), dim3((unsigned)gx), dim3(waves * 64), lds, s, b);
  HET_LAUNCH_CHECK("HET_node_rows_sum_w16");
  *done = true;
  return HET_OK;
}

template <int KS, int NO>
size_t lds_for(int S, int waves) {
  constexpr int XO = NO * 32, LD = (KS > XO ? KS : XO) + 4;
  return sizeof(float) * ((size_t)S * KS * XO + (size_t)waves * (32 * LD + (S + 1) * 32));
}

template <int KS, int NO>
int launch_sum(const SumArgs& a, hipStream_t s) {
  static const bool tile_form = getenv("HET_NODE_SUM_LDS_TILE") && atoi(getenv("HET_NODE_SUM_LDS_TILE")) != 0;  // A/B: the round-4a kernel
  if (!tile_form) {
    bool done = false;
    if (int rc = launch_sum_w16<KS, NO>(a, s, &done)) return rc;
    if (done) return HET_OK;
  }
  const int64_t tiles = (a.n_end - a.n_begin + 31) / 32;
  const size_t limit = het_lds_budget();
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
  return lds_any((int)num_sources, KS, XO, 4) <= het_lds_budget() ? 1 : 0;
}

extern "C" int het_node_rows_matmul_sum_bias(int64_t n_begin, int64_t n_end, int64_t num_nodes, int64_t num_sources,
                                             const float* const* rows, const int64_t* row_strides, const int32_t* const* maps,
                                             const int64_t* ident_rows, const float* const* weights_t, const float* bias,
                                             float* out, int64_t KS, int64_t XO, const int32_t* node_order, het_stream stream) {
  const char* op = "het_node_rows_matmul_sum";
  HET_REQUIRE(0 <= n_begin && n_begin <= n_end && n_end <= num_nodes && num_nodes < (1ll << 31), "%s: bad node range", op);
  HET_REQUIRE(het_node_rows_matmul_sum_ok(num_sources, KS, XO), "%s: unsupported shape: %lld sources of %lld -> %lld floats", op,
              (long long)num_sources, (long long)KS, (long long)XO);
  if (n_begin == n_end) return HET_OK;
  HET_REQUIRE(rows && row_strides && maps && ident_rows && weights_t && out, "%s: null argument", op);
  HET_REQUIRE(((uintptr_t)bias & 15) == 0 && ((uintptr_t)out & 15) == 0, "%s: bias / out not 16-byte aligned", op);
  SumArgs a{};
  a.n_begin = n_begin; a.n_end = n_end; a.N = num_nodes; a.S = (int)num_sources; a.order = node_order; a.out = out; a.bias = bias;
  for (int s = 0; s < a.S; ++s) {
    HET_REQUIRE(rows[s] && weights_t[s] && row_strides[s] >= KS && (row_strides[s] & 3) == 0 &&
                    (((uintptr_t)rows[s] | (uintptr_t)weights_t[s]) & 15) == 0,
                "%s: source %d: null pointer, row stride below the row width, or rows / weight not 16-byte aligned", op, s);
    HET_REQUIRE(maps[s] || (ident_rows[s] >= 0 && ident_rows[s] <= num_nodes), "%s: source %d: neither a map nor a valid identity range", op, s);
    a.rows[s] = rows[s]; a.stride[s] = row_strides[s]; a.map[s] = maps[s]; a.ident_rows[s] = ident_rows[s]; a.wt[s] = weights_t[s];
  }
  hipStream_t st = (hipStream_t)stream;
  if (KS == 64) return XO == 64 ? launch_sum<64, 2>(a, st) : launch_sum<64, 1>(a, st);
  return XO == 64 ? launch_sum<32, 2>(a, st) : launch_sum<32, 1>(a, st);
}

extern "C" int het_node_rows_matmul_sum(int64_t n_begin, int64_t n_end, int64_t num_nodes, int64_t num_sources,
                                        const float* const* rows, const int64_t* row_strides, const int32_t* const* maps,
                                        const int64_t* ident_rows, const float* const* weights_t, float* out, int64_t KS,
                                        int64_t XO, const int32_t* node_order, het_stream stream) {
  return het_node_rows_matmul_sum_bias(n_begin, n_end, num_nodes, num_sources, rows, row_strides, maps, ident_rows, weights_t, nullptr,
                                       out, KS, XO, node_order, stream);
}
