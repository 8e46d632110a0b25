// Node-major GEMM passes of the RGAT layer on the distinct (relation, node) rows.
//
// The a2 backward of the layer (backward_rgnn_relational_matmul, OpExport/RGNNOps.inc.h:946-1010, kernels
// RGNN/my_shmem_sgemm_func.cu.h:711-776) runs per relation in the reference: every relation adds its rows into the shared
// [N,K] input gradient (float atomics there; one read-modify-write launch per relation in round 2 here), and the self-loop
// (RGAT/models.py:378-381) and the attention-vector side add theirs with more passes over the same rows.  Counted in bytes
// the input gradient of ogbn-mag was written once and then re-read and re-written 1.5 times, and the layer input x was read
// three times for the three weight gradients.
//
// Here ONE pass walks the NODES: a tile of 32 nodes collects every term of its rows
//     grad_x[n] = grad_h[n] . W_loop^T  +  SUM_r grad_feat_c[row_r(n)] . W_r^T  +  SUM_r grad_er_c[drow_r(n)] . wa_r^T
// in the accumulators of one MFMA tile and stores the row once (HET_node_dx), and the weight gradients
//     dW_loop += x[n]^T grad_h[n];   dW_r += x[n]^T grad_feat_c[row_r(n)];   dwa_r += x[n]^T grad_er_c[drow_r(n)]
// share one read of x[n] (HET_node_dw).  row_r(n) / drow_r(n) come from [R,N] int32 maps (-1 = the node has no row in
// relation r) built once per graph (het_node_row_map).  Values are those of the per-relation passes up to the order of
// the floating-point sums.
#include <stdlib.h>

#include "common.hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMaxRels = 8;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__global__ __launch_bounds__(256) void HET_node_row_map(const idx_t* __restrict__ rel_ptrs, int R,
                                                         const idx_t* __restrict__ nodes, int64_t N,
                                                         int32_t* __restrict__ map) {
  const idx_t total = rel_ptrs[R];
  for (idx_t i = (idx_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (idx_t)gridDim.x * 256) {
    const int r = find_segment(rel_ptrs, R, i);
    map[(int64_t)r * N + nodes[i]] = (int32_t)i;
  }
}

struct NodeArgs {
  int64_t n_begin, n_end;  // nodes of this launch
  int64_t n_loop;          // nodes < n_loop carry the self-loop term (rows of grad_h)
  int64_t N;               // rows of x / grad_x = stride of the maps
  int R, H, D, rhp;        // rhp: R*H rounded up to a multiple of 8 (<= 32)
  const float* gh;         // [n_loop, X]   gradient of the layer output (NULL: no self-loop term)
  const float* g_rows;     // [S_row, X]    gradient of the (relation, source) rows
  const int32_t* row_map;  // [R, N]
  const float* g_er;       // [S_col, H]    gradient of er (NULL: none)
  const int32_t* dst_map;  // [R, N]
  const float* g_el;       // [S_row, H]    dw: gradient of el (NULL: none) -> grad_wl
  // dx
  const float* loop_wt;    // [X, K]  W_loop^T
  const float* wt;         // [R, X, K]  (= weights_transposed [R,H,D,K])
  const float* wa_t;       // [R, H, K]
  float* grad_x;           // [N, K]
  // dw
  const float* x;          // [N, K]
  float* grad_loop;        // [K, X]
  float* grad_w;           // [R, H, K, D]
  float* grad_wa;          // [R, H, K]
  float* grad_wl;          // [R, H, K]   SUM_n g_el[row_r(n), h] * x[n, :]
  int chunk;               // dw: nodes per workgroup
};

// ---- input gradient -------------------------------------------------------------------------------------------------------
// Workgroup = WAVES independent waves sharing the weights in LDS: [1 + R] matrices [KS][XO] and wa [rhp][XO].  A wave walks
// 32-node tiles; per tile and PRESENT source (self-loop; relation r if any of the 32 nodes has a row in it) the source rows
// are loaded coalesced (KS/4 lanes x float4 per row) into the wave's LDS tile, read back as MFMA A fragments and multiplied
// into the same XO/32 accumulators; the rows of the next present source are in flight during the MFMAs.  The er term is a
// rank-(R*H) extension of the contraction whose fragments the lanes fetch directly.
template <int KS, int NO, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void HET_node_dx(NodeArgs a) {
  constexpr int XO = NO * 32, KH = KS / 2;
  constexpr int LD = (KS > XO ? KS : XO) + 4;
  constexpr int LPRA = KS / 4, RPIA = 64 / LPRA, NITA = 32 / RPIA;
  constexpr int LPRC = XO / 4, RPIC = 64 / LPRC, NITC = 32 / RPIC;
  constexpr bool PAIRED = NO == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = a.R, nmat = 1 + R;
  float* Bs = smem;                                  // [1 + R][KS][XO]
  float* WAs = Bs + nmat * KS * XO;                  // [rhp][XO]
  float* Ws = WAs + a.rhp * XO + wave * (32 * LD + nmat * 32);  // wave-private tile
  int* idsL = reinterpret_cast<int*>(Ws + 32 * LD);  // [1 + R][32] row ids of the tile's sources, -1 = none
  int* idsD = reinterpret_cast<int*>(Ws);            // [R][32] er rows: only live before the first source enters the tile
  for (int e = tid; e < KS * XO; e += WAVES * 64) Bs[e] = a.loop_wt ? a.loop_wt[e] : 0.f;
  for (int e = tid; e < R * KS * XO; e += WAVES * 64) Bs[KS * XO + e] = a.wt[e];
  for (int e = tid; e < a.rhp * XO; e += WAVES * 64) WAs[e] = (a.wa_t && e < R * a.H * XO) ? a.wa_t[e] : 0.f;
  __syncthreads();

  const int row = lane & 31, half = lane >> 5;
  const int ra = lane / LPRA, ca = (lane % LPRA) * 4;
  const int rc = lane / LPRC, cc = (lane % LPRC) * 4;
  const int hshift = a.H == 1 ? 0 : (a.H == 2 ? 1 : (a.H == 4 ? 2 : 3));
  const int erh = a.rhp >> 1;  // er fragments per lane half
  const int64_t tiles = (a.n_end - a.n_begin + 31) / 32, stride = (int64_t)gridDim.x * WAVES;
  int64_t t = (int64_t)blockIdx.x * WAVES + wave;
  if (t >= tiles) return;
  int mcur[kMaxRels], dcur[kMaxRels];
  auto load_maps = [&](int64_t tt) {
    const int64_t node = a.n_begin + tt * 32 + row;
    const int64_t nc = node < a.n_end ? node : a.n_end - 1;
#pragma unroll
    for (int r = 0; r < kMaxRels; ++r) {
      mcur[r] = -1; dcur[r] = -1;
      if (r < R) {
        mcur[r] = a.row_map[(int64_t)r * a.N + nc];
        if (a.g_er) dcur[r] = a.dst_map[(int64_t)r * a.N + nc];
      }
    }
  };
  load_maps(t);
  for (; t < tiles; t += stride) {
    const int64_t nb = a.n_begin + t * 32;
    unsigned mask = 0, dmask = 0;
    {
      const int64_t node = nb + row;
      const bool nv = node < a.n_end;
      const int id0 = (a.gh && nv && node < a.n_loop) ? (int)node : -1;
      idsL[row] = id0;
      if (__ballot(id0 >= 0)) mask |= 1u;
#pragma unroll
      for (int r = 0; r < kMaxRels; ++r) {
        if (r < R) {
          const int id = nv ? mcur[r] : -1, idd = nv ? dcur[r] : -1;
          idsL[(1 + r) * 32 + row] = id;
          idsD[r * 32 + row] = idd;
          if (__ballot(id >= 0)) mask |= 2u << r;
          if (__ballot(idd >= 0)) dmask |= 1u << r;
        }
      }
    }
    if (t + stride < tiles) load_maps(t + stride);  // consumed one tile later

    f32x16 acc[NO];
#pragma unroll
    for (int nt = 0; nt < NO; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    // er fragments of this lane: element k = half * erh + s of the [32][rhp] extension, k = (relation, head)
    float af2[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) af2[s] = 0.f;
    if (dmask) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        af2[s] = 0.f;
        if (s < erh) {
          const int k = half * erh + s, r = k >> hshift, h = k & (a.H - 1);
          if (r < R) {
            const int id = idsD[r * 32 + row];
            if (id >= 0) af2[s] = a.g_er[(int64_t)id * a.H + h];
          }
        }
      }
    }

    float4 areg[NITA];
    auto issue = [&](int s) {
      const float* base = s == 0 ? a.gh : a.g_rows;
#pragma unroll
      for (int it = 0; it < NITA; ++it) {
        const int id = idsL[s * 32 + it * RPIA + ra];
        areg[it] = ld4(base + (int64_t)(id < 0 ? 0 : id) * KS + ca);
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
    if (dmask) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (q < erh) {
          if (PAIRED) {
            const float2 b2 = *reinterpret_cast<const float2*>(&WAs[(half * erh + q) * XO + 2 * row]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af2[q], b2.x, acc[0], 0, 0, 0);
            acc[NO - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af2[q], b2.y, acc[NO - 1], 0, 0, 0);
          } else {
#pragma unroll
            for (int nt = 0; nt < NO; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af2[q], WAs[(half * erh + q) * XO + nt * 32 + row], acc[nt], 0, 0, 0);
          }
        }
      }
    }
    // epilogue: transpose through the wave's LDS tile, whole 16-byte pieces per output row
#pragma unroll
    for (int nt = 0; nt < NO; ++nt)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        Ws[((reg & 3) + 8 * (reg >> 2) + 4 * half) * LD + (PAIRED ? 2 * row + nt : nt * 32 + row)] = acc[nt][reg];
#pragma unroll
    for (int it = 0; it < NITC; ++it) {
      const int64_t node = nb + it * RPIC + rc;
      const float4 v = ld4(&Ws[(it * RPIC + rc) * LD + cc]);
      if (node < a.n_end) st4(a.grad_x + node * XO + cc, v);
    }
  }
}

// ---- weight gradients -----------------------------------------------------------------------------------------------------
// Workgroup = 4 independent waves over a chunk of nodes, a quarter each.  A wave keeps the FULL K x X accumulators of up to
// kDwMaxP products (self-loop, relations: KT*NT 32x32 tiles = 64 registers each for 64 x 64) plus the K x 32 accumulators of
// the narrow product [g_er | g_el] (attention vectors) in the accumulation registers, and walks its nodes two rows per MFMA
// step (the MFMA k dimension runs over the rows; A = x^T: feature index on the M axis).  One 8-byte load of x[n] and one per
// product of g[row_p(n)] feed KT*NT MFMAs per product -- x is read ONCE for all weight gradients of the layer, and the rows go
// from global memory straight into the MFMA operands (no LDS, no barrier in the loop).
//   * Which products the chunk has at all (typed graphs: a node type is the source of a few relations only) is found by a
//     scan of the maps at the start; the row loop is instantiated for P = 1 .. kDwMaxP products (more: further passes).
//   * Row ids: lane l holds the map entry of node blk + l of a 64-node block (one load per product and block, a block
//     ahead); the id of a step's rows is read from that register with v_readlane.
//   * The operands of the next batch of SB steps are in flight during the MFMAs of the current one.  The accumulators
//     leave one wave per SIMD: all latency hiding is this software pipeline.
//   * Narrow product: the [64][32] tile {g_er[drow_r(n), h] | g_el[row_r(n), h]} of a block is staged in a wave-private LDS
//     tile (values fetched a block ahead into registers).
//   * The four waves' accumulators are summed through LDS, product by product; one atomic flush per workgroup and product.
#ifndef HET_DW_SB
#define HET_DW_SB 4
#endif
constexpr int kDwMaxP = 3, kErK = 16;

struct DwSlot {
  const float* g;        // rows of this product's gradient
  const int32_t* map;    // node -> row (NULL: the self-loop, row = node if node < loop_rows)
  int64_t loop_rows;     // map == NULL: rows of the self-loop gradient (0: an unused slot, every row absent)
};

template <int KT, int NT, int P>
__device__ __forceinline__ void dw_pass(const NodeArgs& a, int64_t w0, int64_t w1, const DwSlot (&sl)[P], bool do_narrow, int lane,
                                        float* __restrict__ es, f32x16 (&acc)[kDwMaxP][KT][NT], f32x16 (&acc_n)[KT]) {
  constexpr int K = KT * 32, X = NT * 32, SB = HET_DW_SB;
  const int col = lane & 31, half = lane >> 5;
  const int RH = a.R * a.H;
  const int hshift = a.H == 1 ? 0 : (a.H == 2 ? 1 : (a.H == 4 ? 2 : 3));
  if (w0 >= w1) return;
  // Every load below is UNCONDITIONAL (clamped index, value selected afterwards): a load under a branch makes the compiler
  // drain the whole queue (s_waitcnt vmcnt(0)) at the join, and with it the operand prefetch of the row loop.
  int idv[P], idn[P];  // row ids of the current / next block: lane l <-> node blk + l
  auto load_ids = [&](int64_t blk, int (&ids)[P]) {
    const int64_t node = blk + lane, nc = node < w1 ? node : w1 - 1;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int32_t* mp = sl[p].map ? sl[p].map : a.row_map;  // (the self-loop slot reads a valid dummy entry)
      const int v = mp[nc];
      ids[p] = sl[p].map ? v : (nc < sl[p].loop_rows ? (int)nc : -1);
    }
  };
  // narrow product: ids / values of the NEXT block in registers, the current block's tile in LDS
  const int32_t* dmap = a.g_er ? a.dst_map : a.row_map;  // (valid dummies when a side is absent: its values are masked)
  const float* ger = a.g_er ? a.g_er : a.x;
  const float* gel = a.g_el ? a.g_el : a.x;
  int did[kMaxRels], rid[kMaxRels];
  float nv[2 * kErK];
  unsigned nmask = 0, nmask_next = 0;  // bit k: nv[k] is a real value (else zero)
  auto load_nids = [&](int64_t blk) {
    const int64_t node = blk + lane, nc = node < w1 ? node : w1 - 1;
#pragma unroll
    for (int r = 0; r < kMaxRels; ++r) {
      const int rr = r < a.R ? r : a.R - 1;
      did[r] = dmap[(int64_t)rr * a.N + nc];
      rid[r] = a.row_map[(int64_t)rr * a.N + nc];
    }
    nmask_next = node < w1 ? 0xffffffffu : 0u;
  };
  auto load_nv = [&]() {  // values of the block whose ids are in did[] / rid[]
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < kErK; ++k) {
      const int kk = k < RH ? k : RH - 1, r = kk >> hshift, h = kk & (a.H - 1);
      int de = -1, ro = -1;
#pragma unroll
      for (int rr = 0; rr < kMaxRels; ++rr) { de = rr == r ? did[rr] : de; ro = rr == r ? rid[rr] : ro; }
      nv[k] = ger[(int64_t)(de < 0 ? 0 : de) * a.H + h];
      nv[kErK + k] = gel[(int64_t)(ro < 0 ? 0 : ro) * a.H + h];
      if (k < RH && de >= 0 && a.g_er) m |= 1u << k;
      if (k < RH && ro >= 0 && a.g_el) m |= 1u << (kErK + k);
    }
    nmask = m & nmask_next;
  };
  auto store_es = [&]() {
#pragma unroll
    for (int k = 0; k < 2 * kErK; ++k) es[lane * 33 + k] = ((nmask >> k) & 1u) ? nv[k] : 0.f;
  };
  float av[2][SB][KT], gv[2][P][SB][NT], bv[2][SB];
  auto load_batch = [&](int64_t blk, int b, const int (&ids)[P], float (&A)[SB][KT], float (&G)[P][SB][NT]) {
#pragma unroll
    for (int j = 0; j < SB; ++j) {
      const int64_t n = blk + 2 * SB * b + 2 * j + half, nc = n < w1 ? n : w1 - 1;
      const uint32_t off = (uint32_t)nc * K;
      if (KT == 2) {
        const float2 t = *reinterpret_cast<const float2*>(a.x + off + 2 * col);
        A[j][0] = t.x; A[j][KT - 1] = t.y;
      } else {
        A[j][0] = a.x[off + col];
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const int i0 = __builtin_amdgcn_readlane(ids[p], 2 * SB * b + 2 * j), i1 = __builtin_amdgcn_readlane(ids[p], 2 * SB * b + 2 * j + 1);
        const int id = half ? i1 : i0;
        const uint32_t off = (uint32_t)(id < 0 ? 0 : id) * X;
        if (NT == 2) {
          const float2 t = *reinterpret_cast<const float2*>(sl[p].g + off + 2 * col);
          G[p][j][0] = t.x; G[p][j][NT - 1] = t.y;
        } else {
          G[p][j][0] = sl[p].g[off + col];
        }
      }
  };
  auto mma_batch = [&](int64_t blk, int b, const int (&ids)[P], const float (&A)[SB][KT], const float (&G)[P][SB][NT], const float (&Bv)[SB]) {
#pragma unroll
    for (int j = 0; j < SB; ++j) {
      const int64_t n = blk + 2 * SB * b + 2 * j + half;
      const bool nin = n < w1;  // rows past the wave's range enter every product as zeros
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i0 = __builtin_amdgcn_readlane(ids[p], 2 * SB * b + 2 * j), i1 = __builtin_amdgcn_readlane(ids[p], 2 * SB * b + 2 * j + 1);
        const bool ok = (half ? i1 : i0) >= 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float g = ok ? G[p][j][nt] : 0.f;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
            acc[p][kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(nin ? A[j][kt] : 0.f, g, acc[p][kt][nt], 0, 0, 0);
        }
      }
      if (do_narrow) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
          acc_n[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(nin ? A[j][kt] : 0.f, Bv[j], acc_n[kt], 0, 0, 0);
      }
    }
  };
  auto load_bv = [&](int b, float (&Bv)[SB]) {
#pragma unroll
    for (int j = 0; j < SB; ++j) Bv[j] = es[(2 * SB * b + 2 * j + half) * 33 + col];
  };

  load_ids(w0, idv);
  if (do_narrow) {
    load_nids(w0);
    load_nv();
    store_es();
    load_nids(w0 + 64);
  }
  for (int64_t blk = w0; blk < w1; blk += 64) {
    const bool more = blk + 64 < w1;
    if (more) load_ids(blk + 64, idn);
    if (do_narrow && more) load_nv();  // values of block blk + 64 (their ids arrived during the previous block)
    load_batch(blk, 0, idv, av[0], gv[0]);
    if (do_narrow) load_bv(0, bv[0]);
#pragma unroll 1
    for (int b = 0; b < 32 / SB; b += 2) {  // two batches per iteration: the operand buffers keep static names
      load_batch(blk, b + 1, idv, av[1], gv[1]);
      if (do_narrow) load_bv(b + 1, bv[1]);
      __builtin_amdgcn_sched_barrier(0);
      mma_batch(blk, b, idv, av[0], gv[0], bv[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (b + 2 < 32 / SB) {
        load_batch(blk, b + 2, idv, av[0], gv[0]);
        if (do_narrow) load_bv(b + 2, bv[0]);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_batch(blk, b + 1, idv, av[1], gv[1], bv[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
#pragma unroll
      for (int p = 0; p < P; ++p) idv[p] = idn[p];
      if (do_narrow) {
        store_es();               // the tile of block blk + 64 (all reads of the current one were issued above)
        load_nids(blk + 128);
      }
    }
  }
}

template <int KT, int NT>
__global__ __launch_bounds__(256) void HET_node_dw(NodeArgs a) {
  constexpr int K = KT * 32, X = NT * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4][64 * 33] narrow tiles, [4][16][64] one parked tile, presence words
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: scalar registers)
  const int R = a.R;
  float* es = smem + wave * 64 * 33;
  float* red = smem + 4 * 64 * 33;
  unsigned* pres = reinterpret_cast<unsigned*>(red + 4 * 16 * 64);  // [2]: products, narrow
  const int64_t c0 = a.n_begin + (int64_t)blockIdx.x * a.chunk;
  const int64_t c1 = c0 + a.chunk < a.n_end ? c0 + a.chunk : a.n_end;
  if (c0 >= c1) return;
  const int64_t q = ((c1 - c0 + 3) / 4 + 63) / 64 * 64;  // nodes per wave (whole 64-node blocks)
  const int64_t w0 = c0 + wave * q < c1 ? c0 + wave * q : c1, w1 = w0 + q < c1 ? w0 + q : c1;
  // which products does the chunk have?
  if (tid < 2) pres[tid] = 0;
  __syncthreads();
  {
    unsigned m = 0, mn = 0;
    for (int64_t blk = w0; blk < w1; blk += 64) {
      const int64_t node = blk + lane;
      if (node < w1) {
        if (a.gh && node < a.n_loop) m |= 1u;
#pragma unroll
        for (int r = 0; r < kMaxRels; ++r) {
          if (r < R) {
            if (a.row_map[(int64_t)r * a.N + node] >= 0) { m |= 2u << r; if (a.g_el) mn = 1u; }
            if (a.g_er && a.dst_map[(int64_t)r * a.N + node] >= 0) mn = 1u;
          }
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { m |= __shfl_xor((int)m, off); mn |= __shfl_xor((int)mn, off); }
    if (lane == 0) { atomicOr(&pres[0], m); atomicOr(&pres[1], mn); }
  }
  __syncthreads();
  const unsigned mask = pres[0];
  const bool has_narrow = pres[1] != 0;

  f32x16 acc_n[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_n[kt][e] = 0.f;
  // Flush of one 32x32 accumulator tile: the four waves park it in LDS ([wave][register][lane]), then all 256 threads sum the
  // four copies of 4 elements each and add them to the output with one float atomic per element (addresses are computed for
  // 4 elements per thread, not for the 64 accumulator registers of a product at once).
  // element of register e, lane l: row m = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5) of the tile, column l & 31
  auto flush_tile = [&](const f32x16& t, int kt, int nt, int s) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = t[e];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * j, e = i >> 6, l = i & 63;
      const float v = red[i] + red[1024 + i] + red[2048 + i] + red[3072 + i];
      const int m = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), c = l & 31;
      const int k = KT == 2 ? 2 * m + kt : m, n = NT == 2 ? 2 * c + nt : c;
      if (s == 0) {
        atomicAdd(a.grad_loop + (int64_t)k * X + n, v);
      } else if (s > 0) {
        const int h = n / a.D, d = n - h * a.D;
        atomicAdd(a.grad_w + ((int64_t)(s - 1) * a.H + h) * K * a.D + (int64_t)k * a.D + d, v);
      } else {  // narrow product: columns [0, R*H): er side -> grad_wa;  [kErK, kErK + R*H): el side -> grad_wl
        const int cc = c < kErK ? c : c - kErK;
        float* out = c < kErK ? a.grad_wa : a.grad_wl;
        if (cc < R * a.H && out) atomicAdd(out + (int64_t)cc * K + k, v);
      }
    }
  };
  unsigned rest = mask;
  bool first = true;
  while (rest || (first && has_narrow)) {
    // the next (up to) kDwMaxP present products
    DwSlot sl[kDwMaxP];
    int sid[kDwMaxP];
    int np = 0;
#pragma unroll
    for (int p = 0; p < kDwMaxP; ++p) {
      sid[p] = -1; sl[p].g = a.x; sl[p].map = nullptr; sl[p].loop_rows = 0;  // unused slot: every row absent (zeros)
      if (rest) {
        const int s = __ffs(rest) - 1;
        rest &= rest - 1;
        sid[p] = s;
        sl[p].g = s == 0 ? a.gh : a.g_rows;
        sl[p].map = s == 0 ? nullptr : a.row_map + (int64_t)(s - 1) * a.N;
        sl[p].loop_rows = s == 0 ? a.n_loop : 0;
        np = p + 1;
      }
    }
    f32x16 acc[kDwMaxP][KT][NT];
#pragma unroll
    for (int p = 0; p < kDwMaxP; ++p)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[p][kt][nt][e] = 0.f;
    const bool do_narrow = first && has_narrow;
    // (one instantiation of the row loop: a pass with fewer than kDwMaxP products multiplies zeros in the unused slots)
    dw_pass<KT, NT, kDwMaxP>(a, w0, w1, sl, do_narrow, lane, es, acc, acc_n);
    first = false;
#pragma unroll
    for (int p = 0; p < kDwMaxP; ++p) {
      if (p < np) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) flush_tile(acc[p][kt][nt], kt, nt, sid[p]);
      }
    }
  }
  if (has_narrow) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) flush_tile(acc_n[kt], kt, 0, -1);
  }
}

inline int pow2_heads(int64_t H) { return H == 1 || H == 2 || H == 4 || H == 8; }

template <int KS, int NO>
int launch_dx(const NodeArgs& a, hipStream_t s) {
  constexpr int XO = NO * 32, LD = (KS > XO ? KS : XO) + 4;
  const int64_t tiles = (a.n_end - a.n_begin + 31) / 32;
  auto lds_for = [&](int waves) {
    return sizeof(float) * ((size_t)(1 + a.R) * KS * XO + (size_t)a.rhp * XO + (size_t)waves * (32 * LD + (1 + a.R) * 32));
  };
  const size_t limit = 160 * 1024;
  HET_KTIME("HET_node_dx", s);
  if (lds_for(8) <= limit) {
    const size_t lds = lds_for(8);
    int64_t gx = (tiles + 8 * 4 - 1) / (8 * 4);  // ~4 tiles per wave: the weights are staged once per workgroup
    if (gx < 1) gx = 1;
    HET_HIP(hipFuncSetAttribute((const void*)HET_node_dx<KS, NO, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_node_dx<KS, NO, 8>), dim3((unsigned)gx), dim3(512), lds, s, a);
  } else {
    const size_t lds = lds_for(4);
    HET_REQUIRE(lds <= limit, "het_rgat_node_backward_dx: the weights of %d relations do not fit the LDS", a.R);
    int64_t gx = (tiles + 4 * 4 - 1) / (4 * 4);
    if (gx < 1) gx = 1;
    HET_HIP(hipFuncSetAttribute((const void*)HET_node_dx<KS, NO, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_node_dx<KS, NO, 4>), dim3((unsigned)gx), dim3(256), lds, s, a);
  }
  HET_LAUNCH_CHECK("HET_node_dx");
  return HET_OK;
}

template <int KT, int NT>
int launch_dw(NodeArgs a, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)4 * 64 * 33 + (size_t)4 * 16 * 64) + 2 * sizeof(unsigned);
  // one workgroup per CU is resident (the accumulators take the register file): a few rounds of them
  static const int64_t n_chunks = [] { const char* v = getenv("HET_NODE_DW_CHUNKS"); return v ? (int64_t)atoi(v) : 1024; }();
  const int64_t n = a.n_end - a.n_begin;
  int64_t chunk = (ceil_div64(n, n_chunks) + 255) / 256 * 256;
  if (chunk < 256) chunk = 256;
  a.chunk = (int)chunk;
  HET_REQUIRE((uint64_t)a.N * (KT * 32) < (1ull << 32), "het_rgat_node_backward_dw: x has more than 2^32 elements");
  HET_HIP(hipFuncSetAttribute((const void*)HET_node_dw<KT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HET_KTIME("HET_node_dw", s);
  hipLaunchKernelGGL((HET_node_dw<KT, NT>), dim3((unsigned)ceil_div64(n, chunk)), dim3(256), lds, s, a);
  HET_LAUNCH_CHECK("HET_node_dw");
  return HET_OK;
}

int check_node_args(const char* op, int64_t n_begin, int64_t n_end, int64_t n_loop, int64_t N, int64_t R, int64_t H, int64_t K,
                    int64_t D) {
  HET_REQUIRE(0 <= n_begin && n_begin <= n_end && n_end <= N && n_loop >= 0 && N < (1ll << 31), "%s: bad node range", op);
  HET_REQUIRE(R >= 0 && R <= kMaxRels && pow2_heads(H) && R * H <= 32 && D > 0 && (K == 32 || K == 64) && (H * D == 32 || H * D == 64),
              "%s: unsupported shape R=%lld H=%lld K=%lld D=%lld (het_rgat_node_gemm_ok)", op, (long long)R, (long long)H,
              (long long)K, (long long)D);
  return HET_OK;
}

}  // namespace

extern "C" int het_rgat_node_gemm_ok(int64_t R, int64_t H, int64_t K, int64_t D) {
  if (!(R >= 1 && R <= kMaxRels && pow2_heads(H) && R * H <= kErK && D > 0 && (K == 32 || K == 64) && (H * D == 32 || H * D == 64)))
    return 0;
  // the input-gradient pass keeps all 1 + R transposed weights in LDS (4 waves at least)
  const int64_t KS = H * D, LD = (KS > K ? KS : K) + 4, rhp = (R * H + 7) / 8 * 8;
  const int64_t lds = 4 * ((1 + R) * KS * K + rhp * K + 4 * (32 * LD + (1 + R) * 32));
  return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int het_node_row_map(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* nodes, int64_t num_rows,
                                int64_t num_nodes, int32_t* map, het_stream stream) {
  const char* op = "het_node_row_map";
  hipStream_t s = (hipStream_t)stream;
  HET_REQUIRE(rel_ptrs && num_rels > 0 && num_rows >= 0 && num_nodes >= 0 && (map || num_nodes == 0) && num_rows < (1ll << 31),
              "%s: bad arguments", op);
  if (num_nodes == 0) return HET_OK;
  HET_HIP(hipMemsetAsync(map, 0xff, sizeof(int32_t) * num_rels * num_nodes, s));
  if (num_rows == 0) return HET_OK;
  HET_REQUIRE(nodes, "%s: null node list", op);
  int64_t nb = ceil_div64(num_rows, 256);
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(HET_node_row_map, dim3((unsigned)nb), dim3(256), 0, s, rel_ptrs, (int)num_rels, nodes, num_nodes, map);
  HET_LAUNCH_CHECK("HET_node_row_map");
  return HET_OK;
}

extern "C" int het_rgat_node_backward_dx(int64_t n_begin, int64_t n_end, int64_t n_loop, int64_t num_nodes, int64_t num_rels,
                                         const float* grad_h, const float* loop_wt, const float* g_rows,
                                         const float* weights_t, const int32_t* row_map, const float* g_er, const float* wa_t,
                                         const int32_t* dst_map, float* grad_x, int64_t H, int64_t K, int64_t D,
                                         het_stream stream) {
  const char* op = "het_rgat_node_backward_dx";
  if (int rc = check_node_args(op, n_begin, n_end, n_loop, num_nodes, num_rels, H, K, D)) return rc;
  if (n_begin == n_end) return HET_OK;
  HET_REQUIRE(grad_x && (num_rels == 0 || (g_rows && weights_t && row_map)) && (!grad_h || loop_wt) &&
                  (!g_er || (wa_t && dst_map)),
              "%s: null data pointer", op);
  HET_REQUIRE(!grad_h || n_loop <= num_nodes, "%s: n_loop exceeds the node count", op);
  NodeArgs a{};
  a.n_begin = n_begin; a.n_end = n_end; a.n_loop = grad_h ? n_loop : 0; a.N = num_nodes;
  a.R = (int)num_rels; a.H = (int)H; a.D = (int)D; a.rhp = g_er ? (int)((num_rels * H + 7) / 8 * 8) : 0;
  a.gh = grad_h; a.g_rows = g_rows; a.row_map = row_map; a.g_er = g_er; a.dst_map = dst_map;
  a.loop_wt = loop_wt; a.wt = weights_t; a.wa_t = wa_t; a.grad_x = grad_x;
  hipStream_t s = (hipStream_t)stream;
  const int KS = (int)(H * D);
  if (KS == 64) return K == 64 ? launch_dx<64, 2>(a, s) : launch_dx<64, 1>(a, s);
  return K == 64 ? launch_dx<32, 2>(a, s) : launch_dx<32, 1>(a, s);
}

extern "C" int het_rgat_node_backward_dw(int64_t n_begin, int64_t n_end, int64_t n_loop, int64_t num_nodes, int64_t num_rels,
                                         int64_t num_src_rows, const float* x, const float* grad_h, const float* g_rows,
                                         const int32_t* row_map, const float* g_er, const int32_t* dst_map, const float* g_el,
                                         float* grad_loop, float* grad_w, float* grad_wa, float* grad_wl, int64_t H, int64_t K,
                                         int64_t D, int accumulate, het_stream stream) {
  const char* op = "het_rgat_node_backward_dw";
  if (int rc = check_node_args(op, n_begin, n_end, n_loop, num_nodes, num_rels, H, K, D)) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int64_t X = H * D;
  HET_REQUIRE((!grad_h || grad_loop) && (num_rels == 0 || (grad_w && row_map)) && (!g_er || (grad_wa && dst_map)) &&
                  (!g_el || (grad_wl && row_map)),
              "%s: null output pointer", op);
  HET_REQUIRE((!g_er && !g_el) || num_rels * H <= kErK, "%s: the attention-vector side takes R*H <= %d", op, kErK);
  HET_REQUIRE(num_src_rows >= 0 && (uint64_t)num_src_rows * X < (1ull << 32) && (uint64_t)num_nodes * X < (1ull << 32),
              "%s: more than 2^32 elements in a gradient tensor", op);
  if (!accumulate) {
    if (grad_h) HET_HIP(hipMemsetAsync(grad_loop, 0, sizeof(float) * K * X, s));
    if (num_rels) HET_HIP(hipMemsetAsync(grad_w, 0, sizeof(float) * num_rels * K * X, s));
    if (g_er) HET_HIP(hipMemsetAsync(grad_wa, 0, sizeof(float) * num_rels * H * K, s));
    if (g_el) HET_HIP(hipMemsetAsync(grad_wl, 0, sizeof(float) * num_rels * H * K, s));
  }
  if (n_begin == n_end) return HET_OK;
  HET_REQUIRE(x && (num_rels == 0 || g_rows), "%s: null data pointer", op);
  NodeArgs a{};
  a.n_begin = n_begin; a.n_end = n_end; a.n_loop = grad_h ? n_loop : 0; a.N = num_nodes;
  a.R = (int)num_rels; a.H = (int)H; a.D = (int)D; a.rhp = 0;
  a.gh = grad_h; a.g_rows = g_rows; a.row_map = row_map; a.g_er = g_er; a.dst_map = dst_map; a.g_el = g_el;
  a.x = x; a.grad_loop = grad_loop; a.grad_w = grad_w; a.grad_wa = grad_wa; a.grad_wl = grad_wl;
  if (K == 64) return X == 64 ? launch_dw<2, 2>(a, s) : launch_dw<2, 1>(a, s);
  return X == 64 ? launch_dw<1, 2>(a, s) : launch_dw<1, 1>(a, s);
}
