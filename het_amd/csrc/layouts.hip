// Device-side layout builders: the step immediately before the hot path.
// The reference converts layouts on the CPU -- C++ vector-of-vector bucketing behind
// convert_integrated_coo_to_separate_coo / transpose_csr (hrt/include/DGLHackKernel/OpExport/DataConverters.inc.h:10-344,
// hrt/include/MyHyb/MyHyb.h:1047-1150) and Python loops for the unique (relation, node) lists
// (hrt/python/utils_lite/mydgl_graph_methods.py:10-157).  Here every conversion is one stable radix sort of the
// edge positions by a composite key (hipCUB) followed by coalesced gather / boundary-search / run-length kernels; all
// int64 in, int64 out, caller-allocated outputs, results identical to the reference builders (tests/golden).
#include <hipcub/hipcub.hpp>

#include "common.hip.h"

namespace {

constexpr int kBlock = 256;

inline unsigned blocks_for(int64_t n) {
  int64_t b = ceil_div64(n, kBlock);
  return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}
inline int bits_for(int64_t n) {  // bits to represent values in [0, n)
  int b = 1;
  while (b < 62 && (1ll << b) < n) ++b;
  return b;
}

struct Scratch {  // frees device temporaries on every exit path
  void* p[12] = {};
  int n = 0;
  hipStream_t s = nullptr;  // the stream the temporaries are used on (a caller's allocator orders their reuse on it)
  explicit Scratch(hipStream_t st) : s(st) {}
  ~Scratch() { for (int i = 0; i < n; ++i) (void)het_free_e(p[i]); }
  hipError_t alloc(void** out, size_t bytes) {
    hipError_t e = het_malloc_e(out, bytes ? bytes : 8, s);
    if (e == hipSuccess) p[n++] = *out;
    return e;
  }
};

// key = (hi << lo_bits) | lo for position i of part q (parts are concatenated: position n_per_part * q + i)
__global__ void HET_layout_make_keys(const idx_t* __restrict__ hi, const idx_t* __restrict__ hi_ptrs, int n_hi,
                                     const idx_t* __restrict__ lo_a, const idx_t* __restrict__ lo_b, int64_t n,
                                     int lo_bits, uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t total = lo_b ? 2 * n : n;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t i = t < n ? t : t - n;
    // the high key is given per position (hi) or as bucket pointers over the positions (hi_ptrs)
    const uint64_t h = hi ? (uint64_t)hi[i] : (hi_ptrs ? (uint64_t)find_segment(hi_ptrs, n_hi, i) : 0ull);
    const uint64_t l = lo_a ? (uint64_t)(t < n ? lo_a[i] : lo_b[i]) : 0ull;
    keys[t] = (h << lo_bits) | l;
    vals[t] = (int32_t)t;
  }
}

// out_k[j] = src_k[perm[j]] for up to four arrays (NULL pairs are skipped)
__global__ void HET_layout_gather4(const int32_t* __restrict__ perm, int64_t n, const idx_t* s0, idx_t* d0,
                                   const idx_t* s1, idx_t* d1, const idx_t* s2, idx_t* d2, const idx_t* s3, idx_t* d3) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) {
    const int64_t p = perm[j];
    if (s0) d0[j] = s0[p];
    if (s1) d1[j] = s1[p];
    if (s2) d2[j] = s2[p];
    if (s3) d3[j] = s3[p];
  }
}

// ptrs[k] = number of sorted keys whose high part is < k, k in [0, n_hi]
__global__ void HET_layout_bucket_ptrs(const uint64_t* __restrict__ sorted, int64_t n, int lo_bits, int64_t n_hi,
                                       idx_t* __restrict__ ptrs) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k <= n_hi; k += (int64_t)gridDim.x * kBlock) {
    const uint64_t target = (uint64_t)k << lo_bits;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sorted[mid] < target) lo = mid + 1; else hi = mid;
    }
    ptrs[k] = lo;
  }
}

__global__ void HET_layout_expand_rows(const idx_t* __restrict__ row_ptrs, int64_t num_rows, int64_t n,
                                       idx_t* __restrict__ rows) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) {
    int64_t lo = 0, hi = num_rows;  // last row r with row_ptrs[r] <= j
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (row_ptrs[mid] <= j) lo = mid; else hi = mid;
    }
    rows[j] = lo;
  }
}

__global__ void HET_layout_run_heads(const uint64_t* __restrict__ sorted, int64_t n, int32_t* __restrict__ head) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock)
    head[j] = (j == 0 || sorted[j] != sorted[j - 1]) ? 1 : 0;
}

// run id of sorted rank j = (inclusive sum of heads)[j] - 1.  Writes the unique low keys, and the inverse index of
// every original position: for the single list inverse[pos] = run; for the dual list (two node arrays of the same
// bucketed positions) in the reference's order, per relation [run ids of its rows..., run ids of its cols...].
__global__ void HET_layout_unique_write(const uint64_t* __restrict__ sorted, const int32_t* __restrict__ perm,
                                        const int32_t* __restrict__ run_incl, int64_t total, int64_t n, int lo_bits,
                                        const idx_t* __restrict__ rel_ptrs, idx_t* __restrict__ out_nodes,
                                        idx_t* __restrict__ out_inverse) {
  const uint64_t lo_mask = (1ull << lo_bits) - 1;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < total; j += (int64_t)gridDim.x * kBlock) {
    const int64_t run = run_incl[j] - 1;
    const uint64_t key = sorted[j];
    if (j == 0 || key != sorted[j - 1]) out_nodes[run] = (idx_t)(key & lo_mask);
    if (out_inverse) {
      const int64_t t = perm[j];
      if (total == n) {
        out_inverse[t] = run;
      } else {
        const int64_t i = t < n ? t : t - n;
        const int64_t r = (int64_t)(key >> lo_bits);
        const int64_t b = rel_ptrs[r], cnt = rel_ptrs[r + 1] - b;
        out_inverse[2 * b + (t < n ? 0 : cnt) + (i - b)] = run;
      }
    }
  }
}

__global__ void HET_layout_rank_to_run(const idx_t* __restrict__ pos, const int32_t* __restrict__ run_incl, int64_t n,
                                       idx_t* __restrict__ out) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock)
    out[k] = pos[k] == 0 ? 0 : run_incl[pos[k] - 1];
}

// Stable sort of `total` positions by the 64-bit keys; perm = positions in sorted order.
int sort_positions(Scratch& tmp, uint64_t* keys_in, int32_t* vals_in, int64_t total, int bits, uint64_t** keys_out,
                   int32_t** perm, hipStream_t s) {
  HET_HIP(tmp.alloc((void**)keys_out, sizeof(uint64_t) * total));
  HET_HIP(tmp.alloc((void**)perm, sizeof(int32_t) * total));
  if (total == 0) return HET_OK;
  size_t tb = 0;
  HET_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_in, *keys_out, vals_in, *perm, (int)total, 0, bits, s));
  void* t0 = nullptr;
  HET_HIP(tmp.alloc(&t0, tb));
  HET_HIP(hipcub::DeviceRadixSort::SortPairs(t0, tb, keys_in, *keys_out, vals_in, *perm, (int)total, 0, bits, s));
  return HET_OK;
}

int keyed_sort(Scratch& tmp, const idx_t* hi, const idx_t* hi_ptrs, int64_t n_hi, const idx_t* lo_a, const idx_t* lo_b,
               int64_t lo_bound, int64_t n, int* lo_bits_out, uint64_t** sorted, int32_t** perm, hipStream_t s) {
  const int64_t total = lo_b ? 2 * n : n;
  const int lo_bits = lo_a ? bits_for(lo_bound) : 0, hi_bits = (hi || hi_ptrs) ? bits_for(n_hi) : 0;
  HET_REQUIRE(lo_bits + hi_bits <= 62, "layout: composite key needs %d bits", lo_bits + hi_bits);
  uint64_t* keys_in = nullptr;
  int32_t* vals_in = nullptr;
  HET_HIP(tmp.alloc((void**)&keys_in, sizeof(uint64_t) * total));
  HET_HIP(tmp.alloc((void**)&vals_in, sizeof(int32_t) * total));
  if (total > 0) {
    hipLaunchKernelGGL(HET_layout_make_keys, dim3(blocks_for(total)), dim3(kBlock), 0, s, hi, hi_ptrs, (int)n_hi, lo_a, lo_b,
                       n, lo_bits, keys_in, vals_in);
    HET_LAUNCH_CHECK("HET_layout_make_keys");
  }
  *lo_bits_out = lo_bits;
  return sort_positions(tmp, keys_in, vals_in, total, lo_bits + hi_bits > 0 ? lo_bits + hi_bits : 1, sorted, perm, s);
}

int check_sizes(const char* op, int64_t n, int64_t a, int64_t b) {
  HET_REQUIRE(n >= 0 && n < (1ll << 30), "%s: number of edges out of range", op);
  HET_REQUIRE(a >= 0 && b >= 0 && a < (1ll << 40) && b < (1ll << 40), "%s: bad bounds", op);
  return HET_OK;
}

}  // namespace

// Bucket the edges by relation, inside a bucket sorted by eid (ties keep input order).
extern "C" int het_layout_separate_coo(const int64_t* row, const int64_t* col, const int64_t* rel, const int64_t* eids,
                                       int64_t num_edges, int64_t num_rels, int64_t eid_bound, int64_t* out_rel_ptrs,
                                       int64_t* out_row, int64_t* out_col, int64_t* out_eids, het_stream stream) {
  const char* op = "het_layout_separate_coo";
  if (int rc = check_sizes(op, num_edges, num_rels, eid_bound)) return rc;
  HET_REQUIRE(num_rels > 0 && out_rel_ptrs && (num_edges == 0 || (row && col && rel && eids && out_row && out_col && out_eids)),
              "%s: null pointer", op);
  hipStream_t s = (hipStream_t)stream;
  Scratch tmp(s);
  uint64_t* sorted = nullptr;
  int32_t* perm = nullptr;
  int lo_bits = 0;
  if (int rc = keyed_sort(tmp, rel, nullptr, num_rels, eids, nullptr, eid_bound, num_edges, &lo_bits, &sorted, &perm, s)) return rc;
  hipLaunchKernelGGL(HET_layout_bucket_ptrs, dim3(blocks_for(num_rels + 1)), dim3(kBlock), 0, s, sorted, num_edges, lo_bits,
                     num_rels, out_rel_ptrs);
  HET_LAUNCH_CHECK("HET_layout_bucket_ptrs");
  if (num_edges > 0) {
    hipLaunchKernelGGL(HET_layout_gather4, dim3(blocks_for(num_edges)), dim3(kBlock), 0, s, perm, num_edges, row, out_row, col,
                       out_col, eids, out_eids, (const idx_t*)nullptr, (idx_t*)nullptr);
    HET_LAUNCH_CHECK("HET_layout_gather4");
  }
  HET_HIP(hipStreamSynchronize(s));  // temporaries are freed on return
  return HET_OK;
}

// Integrated COO -> CSR over `row` (stable: ties keep COO order).
extern "C" int het_layout_coo_to_csr(const int64_t* row, const int64_t* col, const int64_t* rel, const int64_t* eids,
                                     int64_t num_edges, int64_t num_rows, int64_t* out_row_ptrs, int64_t* out_col,
                                     int64_t* out_rel, int64_t* out_eids, het_stream stream) {
  const char* op = "het_layout_coo_to_csr";
  if (int rc = check_sizes(op, num_edges, num_rows, 0)) return rc;
  HET_REQUIRE(out_row_ptrs && (num_edges == 0 || (row && col && rel && eids && out_col && out_rel && out_eids)),
              "%s: null pointer", op);
  hipStream_t s = (hipStream_t)stream;
  Scratch tmp(s);
  uint64_t* sorted = nullptr;
  int32_t* perm = nullptr;
  int lo_bits = 0;
  if (int rc = keyed_sort(tmp, row, nullptr, num_rows > 0 ? num_rows : 1, nullptr, nullptr, 0, num_edges, &lo_bits, &sorted, &perm, s))
    return rc;
  hipLaunchKernelGGL(HET_layout_bucket_ptrs, dim3(blocks_for(num_rows + 1)), dim3(kBlock), 0, s, sorted, num_edges, lo_bits,
                     num_rows, out_row_ptrs);
  HET_LAUNCH_CHECK("HET_layout_bucket_ptrs");
  if (num_edges > 0) {
    hipLaunchKernelGGL(HET_layout_gather4, dim3(blocks_for(num_edges)), dim3(kBlock), 0, s, perm, num_edges, col, out_col, rel,
                       out_rel, eids, out_eids, (const idx_t*)nullptr, (idx_t*)nullptr);
    HET_LAUNCH_CHECK("HET_layout_gather4");
  }
  HET_HIP(hipStreamSynchronize(s));
  return HET_OK;
}

// CSR -> CSR of the transposed adjacency (DataConverters.inc.h:283-344): rows of the result are the columns.
extern "C" int het_layout_transpose_csr(const int64_t* row_ptrs, const int64_t* col, const int64_t* eids, const int64_t* rel,
                                        int64_t num_rows, int64_t num_edges, int64_t num_cols, int64_t* out_row_ptrs,
                                        int64_t* out_col, int64_t* out_eids, int64_t* out_rel, het_stream stream) {
  const char* op = "het_layout_transpose_csr";
  if (int rc = check_sizes(op, num_edges, num_rows, num_cols)) return rc;
  HET_REQUIRE(row_ptrs && out_row_ptrs && (num_edges == 0 || (col && eids && rel && out_col && out_eids && out_rel)),
              "%s: null pointer", op);
  hipStream_t s = (hipStream_t)stream;
  Scratch tmp(s);
  idx_t* rows = nullptr;
  HET_HIP(tmp.alloc((void**)&rows, sizeof(idx_t) * num_edges));
  if (num_edges > 0) {
    hipLaunchKernelGGL(HET_layout_expand_rows, dim3(blocks_for(num_edges)), dim3(kBlock), 0, s, row_ptrs, num_rows, num_edges, rows);
    HET_LAUNCH_CHECK("HET_layout_expand_rows");
  }
  return het_layout_coo_to_csr(col, rows, rel, eids, num_edges, num_cols, out_row_ptrs, out_col, out_rel, out_eids, stream);
}

// Sorted unique (relation, node) pairs of positions bucketed by relation (rel_ptrs), their per-relation pointers and
// the inverse index of every position.  nodes_b != NULL: the dual list over both node arrays
// (mydgl_graph_methods.py:104-157), inverse then has 2 * num_edges entries in the reference's order: per relation
// the entries of its nodes_a positions followed by those of its nodes_b positions.
extern "C" int het_layout_unique_rel_nodes(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* nodes_a,
                                           const int64_t* nodes_b, int64_t num_edges, int64_t num_nodes,
                                           int64_t* out_nodes, int64_t* out_rel_ptrs, int64_t* out_inverse,
                                           int64_t* out_count, het_stream stream) {
  const char* op = "het_layout_unique_rel_nodes";
  if (int rc = check_sizes(op, 2 * num_edges, num_rels, num_nodes)) return rc;
  HET_REQUIRE(rel_ptrs && num_rels > 0 && out_rel_ptrs && out_count && (num_edges == 0 || (nodes_a && out_nodes)),
              "%s: null pointer", op);
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = nodes_b ? 2 * num_edges : num_edges;
  Scratch tmp(s);
  uint64_t* sorted = nullptr;
  int32_t *perm = nullptr, *head = nullptr, *run = nullptr;
  int lo_bits = 0;
  if (int rc = keyed_sort(tmp, nullptr, rel_ptrs, num_rels, nodes_a, nodes_b, num_nodes > 0 ? num_nodes : 1, num_edges, &lo_bits,
                          &sorted, &perm, s))
    return rc;
  *out_count = 0;
  if (total > 0) {
    HET_HIP(tmp.alloc((void**)&head, sizeof(int32_t) * total));
    HET_HIP(tmp.alloc((void**)&run, sizeof(int32_t) * total));
    hipLaunchKernelGGL(HET_layout_run_heads, dim3(blocks_for(total)), dim3(kBlock), 0, s, sorted, total, head);
    HET_LAUNCH_CHECK("HET_layout_run_heads");
    size_t tb = 0;
    HET_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, head, run, (int)total, s));
    void* t0 = nullptr;
    HET_HIP(tmp.alloc(&t0, tb));
    HET_HIP(hipcub::DeviceScan::InclusiveSum(t0, tb, head, run, (int)total, s));
    hipLaunchKernelGGL(HET_layout_unique_write, dim3(blocks_for(total)), dim3(kBlock), 0, s, sorted, perm, run, total, num_edges,
                       lo_bits, rel_ptrs, out_nodes, out_inverse);
    HET_LAUNCH_CHECK("HET_layout_unique_write");
    int32_t h_count = 0;
    HET_HIP(hipMemcpyAsync(&h_count, run + (total - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HET_HIP(hipStreamSynchronize(s));
    *out_count = h_count;
  }
  // out_rel_ptrs[k] = number of unique keys with relation < k = runs before the first sorted rank of relation k
  {
    idx_t* pos = nullptr;
    HET_HIP(tmp.alloc((void**)&pos, sizeof(idx_t) * (num_rels + 1)));
    hipLaunchKernelGGL(HET_layout_bucket_ptrs, dim3(blocks_for(num_rels + 1)), dim3(kBlock), 0, s, sorted, total, lo_bits, num_rels, pos);
    HET_LAUNCH_CHECK("HET_layout_bucket_ptrs");
    hipLaunchKernelGGL(HET_layout_rank_to_run, dim3(blocks_for(num_rels + 1)), dim3(kBlock), 0, s, pos, run, num_rels + 1, out_rel_ptrs);
    HET_LAUNCH_CHECK("HET_layout_rank_to_run");
    HET_HIP(hipStreamSynchronize(s));
  }
  return HET_OK;
}
