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
// in the accumulators of one MFMA tile and stores the row once (HET_node_dx).  row_r(n) / drow_r(n) come from [R,N] int32
// maps (-1 = the node has no row in relation r) built once per graph (het_node_row_map).  Values are those of the
// per-relation passes up to the order of the floating-point sums.  (The weight gradients stay per relation: a node-major pass
// multiplies zero rows wherever a node has no row in a relation -- 37 % of the slots on ogbn-mag -- and the fp32 MFMA rate is
// what bounds those kernels once x is read once; five forms were measured, exp/node_dw.hip.txt.)
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
  const int32_t* order;    // [N] or NULL: the node at position p of [n_begin, n_end) (NULL: node p).  A list sorted by which relations
                           // a node has rows in makes the 32-node tiles homogeneous: no zero rows in the MFMA tiles
  // dx
  const float* loop_wt;    // [X, K]  W_loop^T
  const float* wt;         // [R, X, K]  (= weights_transposed [R,H,D,K])
  const float* wa_t;       // [R, H, K]
  float* grad_x;           // [N, K]
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
  float* Ws = WAs + a.rhp * XO + wave * (32 * LD + (nmat + 1) * 32);  // wave-private tile
  int* idsL = reinterpret_cast<int*>(Ws + 32 * LD);  // [1 + R][32] row ids of the tile's sources, -1 = none
  int* idsN = idsL + nmat * 32;                      // [32] node of every row of the tile
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
  int ncur = 0;
  auto load_maps = [&](int64_t tt) {
    const int64_t pos = a.n_begin + tt * 32 + row;
    const int64_t pc = pos < a.n_end ? pos : a.n_end - 1;
    const int64_t nc = a.order ? a.order[pc] : pc;
    ncur = (int)nc;
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
      const int64_t node = ncur;
      const bool nv = nb + row < a.n_end;
      const int id0 = (a.gh && nv && node < a.n_loop) ? (int)node : -1;
      idsL[row] = id0;
      idsN[row] = nv ? (int)node : -1;
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
      const int64_t node = idsN[it * RPIC + rc];
      const float4 v = ld4(&Ws[(it * RPIC + rc) * LD + cc]);
      // non-temporal: the rows leave in the presence-sorted order of the tiles, i.e. scattered over grad_x, and nothing reads them
      // before the step ends -- same-box A/B on the RGAT step 3.691 -> 3.642 / 3.632 ms (profiles/r05/ab_dense.txt, call 8)
      if (node >= 0) st4_nt(a.grad_x + node * XO + cc, v);
    }
  }
}

inline int pow2_heads(int64_t H) { return H == 1 || H == 2 || H == 4 || H == 8; }

template <int KS, int NO>
int launch_dx(const NodeArgs& a, hipStream_t s) {
  constexpr int XO = NO * 32, LD = (KS > XO ? KS : XO) + 4;
  const int64_t tiles = (a.n_end - a.n_begin + 31) / 32;
  auto lds_for = [&](int waves) {
    return sizeof(float) * ((size_t)(1 + a.R) * KS * XO + (size_t)a.rhp * XO + (size_t)waves * (32 * LD + (2 + a.R) * 32));
  };
  const size_t limit = het_lds_budget();
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
  if (!(R >= 1 && R <= kMaxRels && pow2_heads(H) && R * H <= 32 && D > 0 && (K == 32 || K == 64) && (H * D == 32 || H * D == 64)))
    return 0;
  // the input-gradient pass keeps all 1 + R transposed weights in LDS (4 waves at least)
  const int64_t KS = H * D, LD = (KS > K ? KS : K) + 4, rhp = (R * H + 7) / 8 * 8;
  const int64_t lds = 4 * ((1 + R) * KS * K + rhp * K + 4 * (32 * LD + (2 + R) * 32));
  return lds <= het_lds_budget() ? 1 : 0;
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
                                         const int32_t* node_order, het_stream stream) {
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
  a.loop_wt = loop_wt; a.wt = weights_t; a.wa_t = wa_t; a.grad_x = grad_x; a.order = node_order;
  hipStream_t s = (hipStream_t)stream;
  const int KS = (int)(H * D);
  if (KS == 64) return K == 64 ? launch_dx<64, 2>(a, s) : launch_dx<64, 1>(a, s);
  return K == 64 ? launch_dx<32, 2>(a, s) : launch_dx<32, 1>(a, s);
}

