// EXPERIMENT (measured and dropped, round 3; not built into the library).  Two-pass transposition of the per-edge grad_er term:
// pass A (chunk permute through LDS) 0.176 ms, pass B (block sum into an LDS table) 0.58 ms on ogbn-mag -- against 0.58 ms for the
// direct 16-byte gathers of HET_segment_sum_flat4.  Pass B without its LDS float atomics: 0.106 ms; the atomics alone: 0.50 ms
// (ds_add_f32 retires ~1 lane per 3 cycles per CU, whatever the bank spread).  Kept for the record; see DESIGN.md section 4.1.
// Segmented sum of 16-byte rows whose producer writes them in ANOTHER sorted order -- as two local passes.
//
// The RGAT backward on the distinct (relation, node) rows (gat_compact.hip) produces the per-edge term t_e [E,4] of grad_er
// in the rank order of the (relation, source) grouping and needs its sums per (relation, destination) row.  Read directly
// that is 21 M random 16-byte gathers on ogbn-mag -- one 128-byte line each, 0.58 ms whatever the kernel shape
// (exp/gather16.hip: 16-byte records of a 338 MB buffer in random order move at 36 G records/s; inside windows of 1 K
// records at 130 G/s, sequentially at 190 G/s).  The permutation is static (it is the graph), so it is applied in two
// passes that are local on both sides:
//   A  chunk-permute: a workgroup loads a chunk of 8192 consecutive records (128 KiB) into LDS and writes them out sorted by
//      their destination BLOCK (then chunk, then destination position): sequential reads, runs of ~25 records (400 B) on the
//      write side.  The buffer it produces is ordered (block, chunk, position).
//   B  block-sum: a workgroup streams the records of one block of destination rows (<= 64 K records, <= 4096 rows) --
//      contiguous in that buffer -- and adds each to its row of an LDS table; the table is stored once.  A row with more
//      records than a block is spread over single-row blocks that reduce in registers and add with one float atomic each.
// Plan (per grouping, built once on first use, cached with it): for pass A {destination, LDS slot} of every output record,
// for pass B the table row of every record and the block list.
#include <hipcub/hipcub.hpp>

#include <mutex>
#include <vector>

#include "grouping.hip.h"
#include "seg_reduce.hip.h"

namespace {

constexpr int kChunkLog = 13, kChunk = 1 << kChunkLog;  // records per chunk of pass A (128 KiB of LDS)
constexpr int kBlockPos = 32768, kBlockRows = 4096;     // records / table rows per block of pass B

struct Scratch {
  std::vector<void*> p;
  ~Scratch() { for (void* q : p) (void)hipFree(q); }
  hipError_t alloc(void** out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes ? bytes : 4);
    if (e == hipSuccess) p.push_back(*out);
    return e;
  }
};

__device__ __forceinline__ int block_of(const int4* __restrict__ blk, int nb, int q) {  // last block with pos_begin <= q
  int lo = 0, hi = nb;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk[mid].x <= q) lo = mid; else hi = mid;
  }
  return lo;
}

// key of destination position q: (block, chunk of its record); value q
__global__ __launch_bounds__(256) void HET_tp_keys1(const int4* __restrict__ blk, int nb, const int32_t* __restrict__ p0, int64_t E,
                                                    uint32_t num_chunks, uint32_t* __restrict__ key, int32_t* __restrict__ val) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < E; q += (int64_t)gridDim.x * 256) {
    key[q] = (uint32_t)block_of(blk, nb, (int)q) * num_chunks + ((uint32_t)p0[q] >> kChunkLog);
    val[q] = (int32_t)q;
  }
}
// for the record at position pi of the permuted buffer: its table row, and the key (chunk) / value (pi) of the second sort
__global__ __launch_bounds__(256) void HET_tp_keys2(const int4* __restrict__ blk, int nb, const int32_t* __restrict__ order,
                                                    const int32_t* __restrict__ p0, const int32_t* __restrict__ seg_of_rank,
                                                    int64_t E, uint16_t* __restrict__ lid, uint32_t* __restrict__ key,
                                                    int32_t* __restrict__ val) {
  for (int64_t pi = (int64_t)blockIdx.x * 256 + threadIdx.x; pi < E; pi += (int64_t)gridDim.x * 256) {
    const int q = order[pi];
    const int4 b = blk[block_of(blk, nb, (int)pi)];  // (a block covers the same range of positions in both orders)
    lid[pi] = (uint16_t)(seg_of_rank[q] - b.z);
    key[pi] = (uint32_t)p0[q] >> kChunkLog;
    val[pi] = (int32_t)pi;
  }
}
__global__ __launch_bounds__(256) void HET_tp_slots(const int32_t* __restrict__ a_dst, const int32_t* __restrict__ order,
                                                    const int32_t* __restrict__ p0, int64_t E, uint16_t* __restrict__ a_slot,
                                                    int32_t* __restrict__ bad) {
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < E; s += (int64_t)gridDim.x * 256) {
    const int r = p0[order[a_dst[s]]];
    a_slot[s] = (uint16_t)(r & (kChunk - 1));
    if ((r >> kChunkLog) != (int)(s >> kChunkLog)) atomicAdd(bad, 1);  // p0 is not a permutation of the ranks
  }
}

// pass A
__global__ __launch_bounds__(1024) void HET_rows4_chunk_permute(const float4* __restrict__ in, float4* __restrict__ out,
                                                                const int32_t* __restrict__ a_dst,
                                                                const uint16_t* __restrict__ a_slot, int64_t E) {
  extern __shared__ __attribute__((aligned(16))) float4 tile[];
  const int64_t base = (int64_t)blockIdx.x * kChunk;
  for (int i = threadIdx.x; i < kChunk; i += 1024)
    if (base + i < E) tile[i] = in[base + i];
  __syncthreads();
  for (int i = threadIdx.x; i < kChunk; i += 1024) {
    const int64_t s = base + i;
    if (s < E) out[a_dst[s]] = tile[a_slot[s]];
  }
}

// pass B
__global__ __launch_bounds__(1024) void HET_rows4_block_sum(const float4* __restrict__ in, const uint16_t* __restrict__ lid,
                                                            const int4* __restrict__ blk, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float tab[];  // [rows][4]
  const int4 b = blk[blockIdx.x];
  const int pb = b.x, pe = b.y, seg0 = b.z, nseg = b.w & 0x7fffffff;
  const bool atomic = (b.w >> 31) & 1;
  const int tid = threadIdx.x;
  constexpr int U = 8;
  if (nseg == 1) {  // one row: sum in registers, one LDS add per wave
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = pb; base < pe; base += 1024 * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pi = base + u * 1024 + tid;
        v[u] = pi < pe ? in[pi] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
    }
    if (tid < 4) tab[tid] = 0.f;
    __syncthreads();
    if ((tid & 63) == 0) { atomicAdd(&tab[0], a.x); atomicAdd(&tab[1], a.y); atomicAdd(&tab[2], a.z); atomicAdd(&tab[3], a.w); }
    __syncthreads();
    if (tid < 4) {
      if (atomic) atomicAdd(out + (int64_t)seg0 * 4 + tid, tab[tid]);
      else out[(int64_t)seg0 * 4 + tid] = tab[tid];
    }
    return;
  }
  // table as [4][nseg]: the four adds of a record then spread over the banks by ROW (row-major [nseg][4] put every lane of one
  // ds_add instruction on the 8 banks of its component: 8-way conflicts)
  for (int i = tid; i < nseg * 4; i += 1024) tab[i] = 0.f;
  __syncthreads();
  for (int base = pb; base < pe; base += 1024 * U) {  // U independent 16-byte loads per thread in flight
    float4 v[U];
    int l[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pi = base + u * 1024 + tid, pc = pi < pe ? pi : pe - 1;
#ifdef HET_EXP_TP_NOLOAD
      v[u] = make_float4(1.f, 2.f, 3.f, (float)pc);
      l[u] = (pc * 7) % nseg;
#else
      v[u] = in[pc];
      l[u] = lid[pc];
#endif
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#ifdef HET_EXP_TP_NOATOM
      if (v[u].x == 12345.f && base + u * 1024 + tid < pe) {
#else
      if (base + u * 1024 + tid < pe) {
#endif
        float* t = tab + l[u];
        atomicAdd(t, v[u].x); atomicAdd(t + nseg, v[u].y); atomicAdd(t + 2 * nseg, v[u].z); atomicAdd(t + 3 * nseg, v[u].w);
      }
    }
  }
  __syncthreads();
  float4* o4 = reinterpret_cast<float4*>(out + (int64_t)seg0 * 4);
  for (int i = tid; i < nseg; i += 1024) o4[i] = make_float4(tab[i], tab[nseg + i], tab[2 * nseg + i], tab[3 * nseg + i]);
}

inline unsigned blocks_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

std::mutex g_tp_mu;

}  // namespace

struct het_transpose_plan {
  int32_t* a_dst = nullptr;    // [E] pass A: output record s (chunk s / kChunk) goes to this position of the permuted buffer ...
  uint16_t* a_slot = nullptr;  // [E] ... from this slot of the chunk's LDS tile
  uint16_t* lid = nullptr;     // [E] pass B: table row of the record at each position of the permuted buffer
  int4* blk = nullptr;         // [num_blocks] {pos_begin, pos_end, first segment, segments | atomic << 31}
  int num_blocks = 0;
  int max_rows = 0;
};

void transpose_plan_free(het_transpose_plan* p) {
  if (!p) return;
  (void)hipFree(p->a_dst); (void)hipFree(p->a_slot); (void)hipFree(p->lid); (void)hipFree(p->blk);
  delete p;
}

int64_t transpose_plan_bytes(const het_transpose_plan* p, int64_t E) { return p ? 8 * E + 16 * (int64_t)p->num_blocks : 0; }

bool rows4_transposed_sum_supported(const het_grouping* g) {
  return g && g->p0 && g->R == 0 && g->E > 0 && g->E < (1ll << 31) && g->S > 0;
}

// Builds g->tp once (thread-safe; synchronises `s` before publishing).  g->p0 must be a permutation of [0, E): the rank of
// every position in the producer's order.
int grouping_transpose_plan(const het_grouping* g, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_tp_mu);
  if (g->tp) return HET_OK;
  HET_REQUIRE(rows4_transposed_sum_supported(g), "transposed segment sum: the grouping needs payload0 = the producer's rank");
  const int64_t E = g->E, S = g->S;
  if (int rc = grouping_seg_of_rank(g, s)) return rc;
  // blocks of whole segments (host: one pass over the segment pointers)
  std::vector<int32_t> sp((size_t)S + 1);
  HET_HIP(hipMemcpyAsync(sp.data(), g->seg_ptr, sizeof(int32_t) * (S + 1), hipMemcpyDeviceToHost, s));
  HET_HIP(hipStreamSynchronize(s));
  std::vector<int4> hb;
  int max_rows = 1;
  for (int64_t i = 0; i < S;) {
    const int len = sp[i + 1] - sp[i];
    if (len > kBlockPos) {  // a hub row: single-row blocks that add atomically (out is zero-filled)
      for (int b = sp[i]; b < sp[i + 1]; b += kBlockPos)
        hb.push_back(make_int4(b, b + kBlockPos < sp[i + 1] ? b + kBlockPos : sp[i + 1], (int)i, 1 | (int)0x80000000));
      ++i;
      continue;
    }
    int64_t j = i;
    while (j < S && j - i < kBlockRows && sp[j + 1] - sp[i] <= kBlockPos) ++j;
    hb.push_back(make_int4(sp[i], sp[j], (int)i, (int)(j - i)));
    if (j - i > max_rows) max_rows = (int)(j - i);
    i = j;
  }
  const int nb = (int)hb.size();
  const uint32_t num_chunks = (uint32_t)((E + kChunk - 1) >> kChunkLog);
  HET_REQUIRE((uint64_t)nb * num_chunks < (1ull << 32), "transposed segment sum: too many (block, chunk) pairs");
  het_transpose_plan* p = new het_transpose_plan();
  struct Guard { het_transpose_plan* p; ~Guard() { transpose_plan_free(p); } } guard{p};
  p->num_blocks = nb; p->max_rows = max_rows;
  HET_HIP(hipMalloc((void**)&p->blk, sizeof(int4) * (size_t)nb));
  HET_HIP(hipMalloc((void**)&p->a_dst, sizeof(int32_t) * (size_t)E));
  HET_HIP(hipMalloc((void**)&p->a_slot, sizeof(uint16_t) * (size_t)E));
  HET_HIP(hipMalloc((void**)&p->lid, sizeof(uint16_t) * (size_t)E));
  HET_HIP(hipMemcpyAsync(p->blk, hb.data(), sizeof(int4) * (size_t)nb, hipMemcpyHostToDevice, s));
  Scratch tmp;
  uint32_t *key = nullptr, *key_out = nullptr;
  int32_t *val = nullptr, *order = nullptr, *bad = nullptr;
  HET_HIP(tmp.alloc((void**)&key, sizeof(uint32_t) * (size_t)E));
  HET_HIP(tmp.alloc((void**)&key_out, sizeof(uint32_t) * (size_t)E));
  HET_HIP(tmp.alloc((void**)&val, sizeof(int32_t) * (size_t)E));
  HET_HIP(tmp.alloc((void**)&order, sizeof(int32_t) * (size_t)E));
  HET_HIP(tmp.alloc((void**)&bad, sizeof(int32_t)));
  HET_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), s));
  int bits1 = 1, bits2 = 1;
  while (bits1 < 32 && ((uint64_t)nb * num_chunks >> bits1)) ++bits1;
  while (bits2 < 32 && (num_chunks >> bits2)) ++bits2;
  size_t tb1 = 0, tb2 = 0;
  HET_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, key, key_out, val, order, (int)E, 0, bits1, s));
  HET_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, key, key_out, val, p->a_dst, (int)E, 0, bits2, s));
  void* t0 = nullptr;
  HET_HIP(tmp.alloc(&t0, tb1 > tb2 ? tb1 : tb2));
  hipLaunchKernelGGL(HET_tp_keys1, dim3(blocks_for(E)), dim3(256), 0, s, p->blk, nb, g->p0, E, num_chunks, key, val);
  HET_LAUNCH_CHECK("HET_tp_keys1");
  HET_HIP(hipcub::DeviceRadixSort::SortPairs(t0, tb1, key, key_out, val, order, (int)E, 0, bits1, s));  // stable: q ascending inside a key
  hipLaunchKernelGGL(HET_tp_keys2, dim3(blocks_for(E)), dim3(256), 0, s, p->blk, nb, order, g->p0, g->seg_of_rank, E, p->lid, key, val);
  HET_LAUNCH_CHECK("HET_tp_keys2");
  HET_HIP(hipcub::DeviceRadixSort::SortPairs(t0, tb2, key, key_out, val, p->a_dst, (int)E, 0, bits2, s));
  hipLaunchKernelGGL(HET_tp_slots, dim3(blocks_for(E)), dim3(256), 0, s, p->a_dst, order, g->p0, E, p->a_slot, bad);
  HET_LAUNCH_CHECK("HET_tp_slots");
  int32_t h_bad = 0;
  HET_HIP(hipMemcpyAsync(&h_bad, bad, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HET_HIP(hipStreamSynchronize(s));  // published only once complete
  HET_REQUIRE(h_bad == 0, "transposed segment sum: payload0 of the grouping is not a permutation of the producer's ranks");
  guard.p = nullptr;
  g->tp = p;
  return HET_OK;
}

// out[seg, :] = SUM over the positions j of segment seg of in[p0[j], :]   (rows of 4 floats; tmp: [E,4] scratch)
int launch_rows4_transposed_sum(const het_grouping* g, const float* in, float* tmp, float* out, hipStream_t s) {
  if (int rc = grouping_transpose_plan(g, s)) return rc;
  const het_transpose_plan* p = g->tp;
  const int64_t E = g->E;
  HET_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(tmp) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(out) & 15) == 0, "transposed segment sum: rows must be 16-byte aligned");
  HET_HIP(hipMemsetAsync(out, 0, sizeof(float) * 4 * g->S, s));  // rows of hub segments are added to atomically
  const size_t ldsA = sizeof(float4) * kChunk, ldsB = sizeof(float) * 4 * (size_t)p->max_rows;
  HET_HIP(hipFuncSetAttribute((const void*)HET_rows4_chunk_permute, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
  HET_HIP(hipFuncSetAttribute((const void*)HET_rows4_block_sum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
  {
    HET_KTIME("HET_rows4_chunk_permute", s);
    hipLaunchKernelGGL(HET_rows4_chunk_permute, dim3((unsigned)((E + kChunk - 1) >> kChunkLog)), dim3(1024), ldsA, s,
                       reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(tmp), p->a_dst, p->a_slot, E);
  }
  HET_LAUNCH_CHECK("HET_rows4_chunk_permute");
  {
    HET_KTIME("HET_rows4_block_sum", s);
    hipLaunchKernelGGL(HET_rows4_block_sum, dim3((unsigned)p->num_blocks), dim3(1024), ldsB, s,
                       reinterpret_cast<const float4*>(tmp), p->lid, p->blk, out);
  }
  HET_LAUNCH_CHECK("HET_rows4_block_sum");
  return HET_OK;
}

// C entry point (tests, other callers): by_key = het_grouping_create(NULL, 0, key of every position, E, S, payload0 = the rank
// of the position in the order `in` is stored in, NULL) with one segment per output row.
extern "C" int het_segment_sum_rows4_transposed(const het_grouping* by_key, const float* in, float* tmp, float* out,
                                                het_stream stream) {
  const char* op = "het_segment_sum_rows4_transposed";
  HET_REQUIRE(by_key, "%s: null grouping", op);
  if (by_key->E == 0) return HET_OK;
  HET_REQUIRE(in && tmp && out && rows4_transposed_sum_supported(by_key), "%s: null pointer or a grouping without payload0", op);
  return launch_rows4_transposed_sum(by_key, in, tmp, out, (hipStream_t)stream);
}
