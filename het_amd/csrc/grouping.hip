// Device-side construction of het_grouping (radix sort + run-length encode via hipCUB).
#include <hipcub/hipcub.hpp>

#include <atomic>
#include <mutex>

#include "grouping.hip.h"
#include "seg_reduce.hip.h"

namespace {

__global__ void HET_grouping_make_keys(const idx_t* __restrict__ rel_ptrs, int R, const idx_t* __restrict__ keys,
                                       int64_t E, int kb, uint64_t* __restrict__ out, int32_t* __restrict__ vals) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t r = R > 0 ? (uint64_t)find_segment(rel_ptrs, R, i) : 0;
    out[i] = (r << kb) | (uint64_t)keys[i];
    vals[i] = (int32_t)i;
  }
}

__global__ void HET_grouping_segments(const uint64_t* __restrict__ uniq, const int32_t* __restrict__ counts,
                                      int64_t S, int64_t E, int kb, int R, int32_t* __restrict__ seg_ptr,
                                      int32_t* __restrict__ seg_key, int32_t* __restrict__ seg_rel_ptr,
                                      idx_t* __restrict__ seg_key64, idx_t* __restrict__ seg_rel_ptr64,
                                      int32_t* __restrict__ nitems) {
  const uint64_t mask = (kb >= 64) ? ~0ull : ((1ull << kb) - 1);
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (int64_t)gridDim.x * blockDim.x) {
    seg_key[s] = (int32_t)(uniq[s] & mask);
    seg_key64[s] = (idx_t)(uniq[s] & mask);
    nitems[s] = (counts[s] + HET_ITEM_MAX - 1) / HET_ITEM_MAX;
    if (s == 0) seg_ptr[S] = (int32_t)E;
  }
  // relation -> first segment (tiny: one thread per relation boundary)
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (R > 0 && gid <= R) {
    const uint64_t want = (uint64_t)gid << kb;
    int64_t lo = 0, hi = S;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (uniq[mid] < want) lo = mid + 1; else hi = mid;
    }
    seg_rel_ptr[gid] = (int32_t)lo;
    seg_rel_ptr64[gid] = (idx_t)lo;
  }
}

__global__ void HET_grouping_items(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ item_off,
                                   const int32_t* __restrict__ nitems, int64_t S, int32_t* __restrict__ item_seg,
                                   int32_t* __restrict__ item_begin, int32_t* __restrict__ item_end,
                                   int32_t* __restrict__ split_seg, int32_t* __restrict__ split_count) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t b = seg_ptr[s], e = seg_ptr[s + 1], n = nitems[s], o = item_off[s];
    for (int32_t t = 0; t < n; ++t) {
      item_seg[o + t] = (int32_t)s;
      item_begin[o + t] = b + t * HET_ITEM_MAX;
      item_end[o + t] = (b + (t + 1) * HET_ITEM_MAX < e) ? b + (t + 1) * HET_ITEM_MAX : e;
    }
    if (n > 1) split_seg[atomicAdd(split_count, 1)] = (int32_t)s;
  }
}

// d_max (optional): the largest payload value (a bound for the row tables a kernel indexes with it: 32-bit byte offsets)
__global__ void HET_grouping_payload(const int32_t* __restrict__ perm, const idx_t* __restrict__ src, int64_t E,
                                     int32_t* __restrict__ dst, int32_t* __restrict__ d_max) {
  int32_t mx = 0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = (int32_t)src[perm[j]];
    dst[j] = v;
    mx = v > mx ? v : mx;
  }
  if (d_max) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int32_t o = __shfl_xor(mx, off);
      mx = o > mx ? o : mx;
    }
    // (an atomic per wave on ONE word serialises: 3 ms for 21 M values.  The maximum only ever grows, so a wave that cannot raise
    //  it -- all but a handful, once a few large values have been seen -- skips the atomic after a plain read)
    if ((threadIdx.x & 63) == 0 && mx > *reinterpret_cast<volatile int32_t*>(d_max)) atomicMax(d_max, mx);
  }
}

// d_flag[0] |= 1 when payload0 in sorted order is not 0, 1, 2, ... (rows of a segment then are not contiguous)
__global__ void HET_grouping_p0_not_identity(const int32_t* __restrict__ p0, int64_t E, int32_t* __restrict__ d_flag) {
  bool bad = false;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (int64_t)gridDim.x * blockDim.x)
    bad |= p0[j] != (int32_t)j;
  if (bad) *d_flag = 1;  // every writer stores the same value (an atomic per wave on one word took 2 ms)
}

__global__ void HET_grouping_rank_of_position(const int32_t* __restrict__ perm, int64_t E, idx_t* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (int64_t)gridDim.x * blockDim.x)
    out[perm[j]] = j;
}

// flag[j] = 1: a pack starts at rank j; 2: a piece of a long segment starts there.  One thread per segment: a short segment
// opens a pack when it is the first to start inside its HET_PACK_T-block of ranks (or follows a long segment); a long
// one is a pack of its own, flagged (its work goes to the wave-per-item kernels through long_items).  Every flag lies
// inside the writing segment's own range: no races.
__global__ void HET_grouping_long_items(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ item_seg,
                                        int64_t num_items, uint8_t* __restrict__ is_long, int pack_t) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < num_items; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = item_seg[t];
    is_long[t] = seg_ptr[s + 1] - seg_ptr[s] > pack_t;
  }
}

// is_hub[t] = the key of item t's segment (seg_key / R) owns more than hub_min positions in the twin grouping
__global__ void HET_grouping_hub_flags(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_key,
                                       const int32_t* __restrict__ item_seg, int64_t num_items, int R,
                                       const int32_t* __restrict__ twin_key, const int32_t* __restrict__ twin_ptr, int64_t twin_S,
                                       int hub_min, uint8_t* __restrict__ is_hub) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < num_items; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = item_seg[t];
    bool hub = seg_ptr[s + 1] - seg_ptr[s] > hub_min;
    if (!hub) {
      const int32_t v = seg_key[s] / R;
      int64_t lo = 0, hi = twin_S;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (twin_key[mid] < v) lo = mid + 1; else hi = mid;
      }
      hub = lo < twin_S && twin_key[lo] == v && twin_ptr[lo + 1] - twin_ptr[lo] > hub_min;
    }
    is_hub[t] = hub;
  }
}
// key[k] = payload0 of the twin at the first rank of hub item k (the twin shares the sorted order); val[k] = k
__global__ void HET_grouping_hub_first_p0(const int32_t* __restrict__ hub_items, int64_t n, const int32_t* __restrict__ item_begin,
                                          const int32_t* __restrict__ twin_p0, uint32_t* __restrict__ key, int32_t* __restrict__ val) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    key[k] = (uint32_t)twin_p0[item_begin[hub_items[k]]];
    val[k] = (int32_t)k;
  }
}
__device__ __forceinline__ int64_t hub_lower_bound(const int32_t* __restrict__ a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// rec[k] = {s_lo, s_hi, i_lo, v}: the runs (segments of the grouping by key * R + relation) of hub k = twin segment hub_segs[k] with key
// v are [s_lo, s_hi); its work items are consecutive and so are their positions in hub_items, the first one at i_lo
__global__ void HET_grouping_hub_records(const int32_t* __restrict__ hub_segs, int64_t num_hubs, const int32_t* __restrict__ twin_seg_key,
                                         const int32_t* __restrict__ seg_key, int64_t S, const int32_t* __restrict__ item_seg, int64_t NI,
                                         const int32_t* __restrict__ hub_items, int64_t num_hub_items, int R, int4* __restrict__ rec) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < num_hubs; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = twin_seg_key[hub_segs[k]];
    const int64_t s_lo = hub_lower_bound(seg_key, S, v * R), s_hi = hub_lower_bound(seg_key, S, (v + 1) * R);
    const int64_t item_lo = hub_lower_bound(item_seg, NI, s_lo);
    const int64_t i_lo = hub_lower_bound(hub_items, num_hub_items, item_lo);
    rec[k] = make_int4((int)s_lo, (int)s_hi, (int)i_lo, (int)v);
  }
}
__global__ void HET_grouping_long_seg_flags(const int32_t* __restrict__ seg_ptr, int64_t S, int min_len, uint8_t* __restrict__ flag) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (int64_t)gridDim.x * blockDim.x)
    flag[s] = seg_ptr[s + 1] - seg_ptr[s] > min_len;
}

__global__ void HET_grouping_pack_flags(const int32_t* __restrict__ seg_ptr, int64_t S, uint8_t* __restrict__ flag, int pack_t) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t b = seg_ptr[s], e = seg_ptr[s + 1];
    const bool lng = e - b > pack_t;
    if (lng) {
      flag[b] = 2;
    } else if (e > b) {
      const int32_t pb = s > 0 ? seg_ptr[s - 1] : -1;
      const bool prev_long = s > 0 && b - pb > pack_t;
      if (s == 0 || prev_long || b / pack_t != pb / pack_t) flag[b] = 1;
    }
  }
}

__global__ void HET_grouping_pack_finish(int32_t* __restrict__ pack_ptr, const int32_t* __restrict__ d_num,
                                         const uint8_t* __restrict__ flag, int64_t E) {
  const int64_t n = *d_num;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += (int64_t)gridDim.x * blockDim.x) {
    if (k == n) { pack_ptr[k] = (int32_t)E; continue; }
    const int32_t j = pack_ptr[k];
    if (flag[j] == 2) pack_ptr[k] = (int32_t)((uint32_t)j | 0x80000000u);
  }
}

__global__ void HET_grouping_key_of_rank(const int32_t* __restrict__ seg_of_rank, const int32_t* __restrict__ seg_key,
                                         int64_t E, int32_t* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= E; j += (int64_t)gridDim.x * blockDim.x)
    out[j] = j < E ? seg_key[seg_of_rank[j]] : -1;
}

int bits_for(int64_t n) {  // bits to represent values in [0, n)
  int b = 1;
  while (b < 63 && (1ll << b) < n) ++b;
  return b;
}

unsigned blocks_for(int64_t n) {
  int64_t b = ceil_div64(n, 256);
  return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

struct Scratch {  // frees device temporaries on every exit path
  static constexpr int kSlots = 16;
  void* p[kSlots] = {};
  int n = 0;
  hipStream_t s = nullptr;  // the stream the temporaries are used on (a caller's allocator orders their reuse on it)
  explicit Scratch(hipStream_t st) : s(st) {}
  ~Scratch() { for (int i = 0; i < n; ++i) (void)het_free_e(p[i]); }
  hipError_t alloc(void** out, size_t bytes) {
    if (n >= kSlots) return hipErrorOutOfMemory;  // (more temporaries than slots: a bug in the caller, not a crash)
    hipError_t e = het_malloc_e(out, bytes ? bytes : 4, s);
    if (e == hipSuccess) p[n++] = *out;
    return e;
  }
};

}  // namespace

static std::mutex g_used_mu;

extern "C" void het_grouping_note_stream(const het_grouping* g, het_stream stream) {
  if (!g) return;
  hipStream_t s = (hipStream_t)stream;
  if (s == g->home) return;
  std::lock_guard<std::mutex> lk(g_used_mu);
  for (hipStream_t u : g->used)
    if (u == s) return;
  g->used.push_back(s);
}

extern "C" void het_grouping_destroy(het_grouping* g) {
  if (!g) return;
  // Order the release after the last use on every stream that read the arrays (see het_grouping::home): `home` waits for an event
  // recorded now on each of them -- no host synchronisation.  With hipMalloc / hipFree (the default allocator) hipFree itself
  // waits for the device; the waits below are then redundant and cheap.  The library's own side stream is joined into the
  // caller's stream before an entry point returns, so the caller's streams cover it.
  {
    std::vector<hipStream_t> used;
    {
      std::lock_guard<std::mutex> lk(g_used_mu);
      used.swap(g->used);
    }
    for (hipStream_t u : used) {
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); break; }
      if (hipEventRecord(ev, u) != hipSuccess || hipStreamWaitEvent(g->home, ev, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();  // (a stream that no longer exists, a capture in progress ...: the blunt form)
      }
      (void)hipEventDestroy(ev);  // (released by the runtime once it has completed)
    }
  }
  void* ptrs[] = {g->seg_key64, g->seg_rel_ptr64, g->perm, g->seg_ptr, g->seg_key, g->seg_rel_ptr, g->item_seg, g->item_begin, g->item_end,
                  g->split_seg, g->p0, g->p1, g->seg_of_rank, g->pack_ptr, g->key_of_rank, g->long_items, g->p01, g->kp01, g->hub_items, g->hub_segs, g->hub_order, g->hub_rec, g->val_order, g->alt_pack_ptr, g->alt_long_items};
  for (void* p : ptrs)
    if (p) (void)het_free_e(p);
  for (void* p : g->retired) (void)het_free_e(p);
  delete g;
}

extern "C" int64_t het_grouping_num_segments(const het_grouping* g) { return g ? g->S : -1; }

// Device bytes the grouping holds right now: what a memory report has to add when they came from hipMalloc (the default);
// with a caller's allocator (het_set_allocator) they are inside that allocator's own statistics.
extern "C" int64_t het_grouping_bytes(const het_grouping* g) {
  if (!g) return 0;
  const int64_t E = g->E, S = g->S, I = g->num_items, R = g->R;
  int64_t b = 4 * (E > 0 ? E : 1);                                   // perm
  b += 4 * (S + 1) + 4 * (S > 0 ? S : 1) + 8 * (S > 0 ? S : 1);      // seg_ptr, seg_key, seg_key64
  if (R > 0) b += 4 * (R + 1) + 8 * (R + 1);                         // seg_rel_ptr, seg_rel_ptr64
  b += 3 * 4 * (I > 0 ? I : 1) + 4 * (E / HET_ITEM_MAX + 1);         // item_seg / begin / end, split_seg
  if (g->p0) b += 4 * E;
  if (g->p1) b += 4 * E;
  if (g->seg_of_rank) b += 4 * E;
  if (g->pack_ptr) b += 4 * (g->num_packs + 1) + 4 * (g->num_long_items + 1);
  if (g->alt_pack_ptr) b += 4 * (g->alt_num_packs + 1) + 4 * (g->alt_num_long_items + 1);
  if (g->key_of_rank) b += 4 * (E + 1);
  if (g->p01) b += 8 * (E > 0 ? E : 1);
  if (g->kp01) b += 16 * (E + 1);
  if (g->val_order) b += 4 * g->val_order_n;
  if (g->hub_items) b += 4 * (g->num_hub_items + 1) + 4 * (g->num_hub_segs + 1) + (g->hub_order ? 4 * g->num_hub_items : 0) + (g->hub_rec ? 16 * g->num_hub_segs : 0);
  return b;
}

extern "C" int het_grouping_rank_of_position(const het_grouping* g, int64_t* out, het_stream stream) {
  HET_REQUIRE(g && (g->E == 0 || (out && g->perm)), "het_grouping_rank_of_position: null argument");
  if (g->E == 0) return HET_OK;
  hipLaunchKernelGGL(HET_grouping_rank_of_position, dim3(blocks_for(g->E)), dim3(256), 0, (hipStream_t)stream, g->perm, g->E, out);
  HET_LAUNCH_CHECK("HET_grouping_rank_of_position");
  return HET_OK;
}

extern "C" int het_grouping_segment_map(const het_grouping* g, int64_t num_keys, int32_t* map, het_stream stream) {
  HET_REQUIRE(g && g->R > 0 && num_keys >= g->key_bound, "het_grouping_segment_map: needs a grouping by (relation, key) and num_keys >= its key bound");
  return het_node_row_map(g->seg_rel_ptr64, g->R, g->seg_key64, g->S, num_keys, map, stream);
}

extern "C" int het_grouping_create(const int64_t* rel_ptrs, int64_t num_rels, const int64_t* keys,
                                   int64_t num_positions, int64_t key_bound, const int64_t* payload0,
                                   const int64_t* payload1, het_stream stream, het_grouping** out) {
  const char* op = "het_grouping_create";
  HET_REQUIRE(out, "%s: out is NULL", op);
  *out = nullptr;
  HET_REQUIRE(num_positions >= 0 && num_positions < (1ll << 31), "%s: num_positions out of range", op);
  HET_REQUIRE(key_bound >= 1 && key_bound < (1ll << 31), "%s: key_bound out of range", op);
  HET_REQUIRE(num_positions == 0 || keys, "%s: keys is NULL", op);
  HET_REQUIRE(rel_ptrs ? (num_rels > 0 && num_rels < (1 << 20)) : true, "%s: bad num_rels", op);
  hipStream_t s = (hipStream_t)stream;
  const int64_t E = num_positions;
  const int R = rel_ptrs ? (int)num_rels : 0;
  const int kb = bits_for(key_bound), rb = R > 0 ? bits_for(R) : 0;

  het_grouping* g = new het_grouping;
  static std::atomic<uint64_t> next_serial{1};
  g->serial = next_serial.fetch_add(1);
  g->E = E; g->R = R; g->key_bound = key_bound;
  g->home = s;
  struct Guard { het_grouping* g; ~Guard() { if (g) het_grouping_destroy(g); } } guard{g};
#define GALLOC(field, count) HET_HIP(het_malloc_e((void**)&g->field, sizeof(int32_t) * ((count) > 0 ? (count) : 1), s))
  GALLOC(perm, E);
  if (R > 0) GALLOC(seg_rel_ptr, R + 1);
  if (R > 0) HET_HIP(het_malloc_e((void**)&g->seg_rel_ptr64, sizeof(idx_t) * (R + 1), s));

  Scratch tmp(s);
  uint64_t *keys_in = nullptr, *keys_out = nullptr, *uniq = nullptr;
  int32_t *vals_in = nullptr, *counts = nullptr, *d_scalars = nullptr, *nitems = nullptr, *item_off = nullptr;
  HET_HIP(tmp.alloc((void**)&keys_in, sizeof(uint64_t) * E));
  HET_HIP(tmp.alloc((void**)&keys_out, sizeof(uint64_t) * E));
  HET_HIP(tmp.alloc((void**)&vals_in, sizeof(int32_t) * E));
  HET_HIP(tmp.alloc((void**)&d_scalars, sizeof(int32_t) * 8));
  HET_HIP(hipMemsetAsync(d_scalars, 0, sizeof(int32_t) * 8, s));

  int32_t h_runs = 0;
  if (E > 0) {
    hipLaunchKernelGGL(HET_grouping_make_keys, dim3(blocks_for(E)), dim3(256), 0, s, rel_ptrs, R, keys, E, kb, keys_in, vals_in);
    HET_LAUNCH_CHECK("HET_grouping_make_keys");
    size_t tb = 0;
    HET_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_in, keys_out, vals_in, g->perm, (int)E, 0, kb + rb, s));
    void* t0 = nullptr;
    HET_HIP(het_malloc_e(&t0, tb ? tb : 4, s));
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(t0, tb, keys_in, keys_out, vals_in, g->perm, (int)E, 0, kb + rb, s);
    // keys_in is dead after the sort: reuse it for the unique keys, vals_in for the run lengths
    uniq = keys_in;
    counts = vals_in;
    size_t tb2 = 0;
    if (e == hipSuccess) e = hipcub::DeviceRunLengthEncode::Encode(nullptr, tb2, keys_out, uniq, counts, d_scalars, (int)E, s);
    if (e == hipSuccess && tb2 > tb) {
      (void)hipStreamSynchronize(s);
      (void)het_free_e(t0);
      t0 = nullptr;
      e = het_malloc_e(&t0, tb2, s);
    }
    if (e == hipSuccess) e = hipcub::DeviceRunLengthEncode::Encode(t0, tb2, keys_out, uniq, counts, d_scalars, (int)E, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_runs, d_scalars, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)het_free_e(t0);
    HET_HIP(e);
  }
  const int64_t S = h_runs;
  g->S = S;
  GALLOC(seg_ptr, S + 1);
  GALLOC(seg_key, S);
  HET_HIP(het_malloc_e((void**)&g->seg_key64, sizeof(idx_t) * (S > 0 ? S : 1), s));
  HET_HIP(tmp.alloc((void**)&nitems, sizeof(int32_t) * S));
  HET_HIP(tmp.alloc((void**)&item_off, sizeof(int32_t) * S));
  if (S == 0) {
    HET_HIP(hipMemsetAsync(g->seg_ptr, 0, sizeof(int32_t), s));
    if (R > 0) HET_HIP(hipMemsetAsync(g->seg_rel_ptr, 0, sizeof(int32_t) * (R + 1), s));
    if (R > 0) HET_HIP(hipMemsetAsync(g->seg_rel_ptr64, 0, sizeof(idx_t) * (R + 1), s));
    GALLOC(item_seg, 0); GALLOC(item_begin, 0); GALLOC(item_end, 0); GALLOC(split_seg, 0);
  } else {
    size_t tb = 0;
    HET_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, counts, g->seg_ptr, (int)S, s));
    void* t0 = nullptr;
    HET_HIP(het_malloc_e(&t0, tb ? tb : 4, s));
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(t0, tb, counts, g->seg_ptr, (int)S, s);
    if (e == hipSuccess) {
      const int64_t n = S > R + 1 ? S : R + 1;
      hipLaunchKernelGGL(HET_grouping_segments, dim3(blocks_for(n)), dim3(256), 0, s, uniq, counts, S, E, kb, R,
                         g->seg_ptr, g->seg_key, g->seg_rel_ptr, g->seg_key64, g->seg_rel_ptr64, nitems);
      e = hipGetLastError();
    }
    size_t tb2 = 0;
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, nitems, item_off, (int)S, s);
    if (e == hipSuccess && tb2 > tb) {
      (void)hipStreamSynchronize(s);
      (void)het_free_e(t0);
      t0 = nullptr;
      e = het_malloc_e(&t0, tb2, s);
    }
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(t0, tb2, nitems, item_off, (int)S, s);
    int32_t last_off = 0, last_n = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&last_off, item_off + (S - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&last_n, nitems + (S - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)het_free_e(t0);
    HET_HIP(e);
    g->num_items = (int64_t)last_off + last_n;
    GALLOC(item_seg, g->num_items);
    GALLOC(item_begin, g->num_items);
    GALLOC(item_end, g->num_items);
    GALLOC(split_seg, E / HET_ITEM_MAX + 1);  // a split segment has > HET_ITEM_MAX positions
    hipLaunchKernelGGL(HET_grouping_items, dim3(blocks_for(S)), dim3(256), 0, s, g->seg_ptr, item_off, nitems, S,
                       g->item_seg, g->item_begin, g->item_end, g->split_seg, d_scalars + 1);
    HET_LAUNCH_CHECK("HET_grouping_items");
    int32_t h_split = 0;
    HET_HIP(hipMemcpyAsync(&h_split, d_scalars + 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HET_HIP(hipStreamSynchronize(s));
    g->num_split = h_split;
  }
  if (payload0 && E > 0) {
    GALLOC(p0, E);
    hipLaunchKernelGGL(HET_grouping_payload, dim3(blocks_for(E)), dim3(256), 0, s, g->perm, payload0, E, g->p0, d_scalars + 4);
    HET_LAUNCH_CHECK("HET_grouping_payload");
    hipLaunchKernelGGL(HET_grouping_p0_not_identity, dim3(blocks_for(E)), dim3(256), 0, s, g->p0, E, d_scalars + 2);
    HET_LAUNCH_CHECK("HET_grouping_p0_not_identity");
    int32_t h_bad = 1, h_max = 0;
    HET_HIP(hipMemcpyAsync(&h_bad, d_scalars + 2, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HET_HIP(hipMemcpyAsync(&h_max, d_scalars + 4, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HET_HIP(hipStreamSynchronize(s));
    g->p0_contiguous = h_bad == 0;
    g->p0_max = h_max;
  }
  if (payload1 && E > 0) {
    GALLOC(p1, E);
    hipLaunchKernelGGL(HET_grouping_payload, dim3(blocks_for(E)), dim3(256), 0, s, g->perm, payload1, E, g->p1, d_scalars + 5);
    HET_LAUNCH_CHECK("HET_grouping_payload");
    int32_t h_max = 0;
    HET_HIP(hipMemcpyAsync(&h_max, d_scalars + 5, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HET_HIP(hipStreamSynchronize(s));
    g->p1_max = h_max;
  }
  HET_HIP(hipStreamSynchronize(s));  // temporaries are freed on return
#undef GALLOC
  guard.g = nullptr;
  *out = g;
  return HET_OK;
}

static std::mutex g_pack_mu;

// One set of packs for a threshold (caller holds g_pack_mu).  key_of_rank, which does not depend on the threshold, is built with the
// first set and shared.
static int build_pack_set(const het_grouping* g, hipStream_t s, int pack_t, int32_t** pack_ptr_out, int32_t** long_items_out,
                          int64_t* num_packs_out, int64_t* num_long_out) {
  const int64_t E = g->E, S = g->S;
  if (int rc = grouping_seg_of_rank(g, s)) return rc;
  Scratch tmp(s);
  const int64_t NI = g->num_items;
  uint8_t *flag = nullptr, *is_long = nullptr;
  int32_t* d_num = nullptr;
  HET_HIP(tmp.alloc((void**)&flag, (size_t)E));
  HET_HIP(tmp.alloc((void**)&is_long, (size_t)NI));
  HET_HIP(tmp.alloc((void**)&d_num, sizeof(int32_t) * 2));
  HET_HIP(hipMemsetAsync(flag, 0, (size_t)E, s));
  hipLaunchKernelGGL(HET_grouping_pack_flags, dim3(blocks_for(S)), dim3(256), 0, s, g->seg_ptr, S, flag, pack_t);
  HET_LAUNCH_CHECK("HET_grouping_pack_flags");
  hipLaunchKernelGGL(HET_grouping_long_items, dim3(blocks_for(NI)), dim3(256), 0, s, g->seg_ptr, g->item_seg, NI, is_long, pack_t);
  HET_LAUNCH_CHECK("HET_grouping_long_items");
  int32_t *pack_tmp = nullptr, *long_tmp = nullptr;
  HET_HIP(tmp.alloc((void**)&pack_tmp, sizeof(int32_t) * (size_t)(E + 1)));
  HET_HIP(tmp.alloc((void**)&long_tmp, sizeof(int32_t) * (size_t)NI));
  hipcub::CountingInputIterator<int32_t> ranks(0);
  size_t tb = 0, tb2 = 0;
  HET_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb, ranks, flag, pack_tmp, d_num, (int)E, s));
  HET_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb2, ranks, is_long, long_tmp, d_num + 1, (int)NI, s));
  void* t0 = nullptr;
  HET_HIP(tmp.alloc(&t0, tb > tb2 ? tb : tb2));
  HET_HIP(hipcub::DeviceSelect::Flagged(t0, tb, ranks, flag, pack_tmp, d_num, (int)E, s));
  HET_HIP(hipcub::DeviceSelect::Flagged(t0, tb2, ranks, is_long, long_tmp, d_num + 1, (int)NI, s));
  int32_t h_num[2] = {0, 0};
  HET_HIP(hipMemcpyAsync(h_num, d_num, sizeof(h_num), hipMemcpyDeviceToHost, s));
  HET_HIP(hipStreamSynchronize(s));
  int32_t *pack_ptr = nullptr, *key_of_rank = nullptr, *long_items = nullptr;
  const bool need_keys = g->key_of_rank == nullptr;
  HET_HIP(het_malloc_e((void**)&pack_ptr, sizeof(int32_t) * ((size_t)h_num[0] + 1), s));
  hipError_t e = need_keys ? het_malloc_e((void**)&key_of_rank, sizeof(int32_t) * ((size_t)E + 1), s) : hipSuccess;
  if (e == hipSuccess) e = het_malloc_e((void**)&long_items, sizeof(int32_t) * ((size_t)h_num[1] + 1), s);
  if (e == hipSuccess) e = hipMemcpyAsync(pack_ptr, pack_tmp, sizeof(int32_t) * (size_t)h_num[0], hipMemcpyDeviceToDevice, s);
  // (issuing the long work items in the order of their first gathered row -- as the hub items of the RGAT forward are -- was
  //  measured with (source, destination)-ordered edge lists and lost: RGAT 4.17 -> 4.28 ms, RGCN 3.02 -> 3.12; segment order stays)
  if (e == hipSuccess) e = hipMemcpyAsync(long_items, long_tmp, sizeof(int32_t) * (size_t)h_num[1], hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(HET_grouping_pack_finish, dim3(blocks_for(h_num[0] + 1)), dim3(256), 0, s, pack_ptr, d_num, flag, E);
    if (need_keys)
      hipLaunchKernelGGL(HET_grouping_key_of_rank, dim3(blocks_for(E + 1)), dim3(256), 0, s, g->seg_of_rank, g->seg_key, E,
                         key_of_rank);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // published only once complete: later users may be on other streams
  if (e != hipSuccess) {
    (void)het_free_e(pack_ptr); (void)het_free_e(key_of_rank); (void)het_free_e(long_items);
    HET_HIP(e);
  }
  if (need_keys) g->key_of_rank = key_of_rank;
  *pack_ptr_out = pack_ptr;
  *long_items_out = long_items;
  *num_packs_out = h_num[0];
  *num_long_out = h_num[1];
  return HET_OK;
}

int grouping_packs(const het_grouping* g, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_pack_mu);
  if (g->pack_ptr || g->E == 0 || g->S == 0) return HET_OK;
  int32_t *pp = nullptr, *li = nullptr;
  int64_t np = 0, nl = 0;
  if (int rc = build_pack_set(g, s, HET_PACK_T, &pp, &li, &np, &nl)) return rc;
  g->long_items = li;
  g->num_long_items = nl;
  g->num_packs = np;
  g->pack_ptr = pp;
  return HET_OK;
}

int grouping_pack_view(const het_grouping* g, hipStream_t s, int pack_t, PackView* out) {
  *out = PackView{};
  if (g->E == 0 || g->S == 0) return HET_OK;
  if (pack_t <= 0 || pack_t == HET_PACK_T) {
    if (int rc = grouping_packs(g, s)) return rc;
    *out = PackView{g->pack_ptr, g->long_items, g->num_packs, g->num_long_items};
    return HET_OK;
  }
  std::lock_guard<std::mutex> lk(g_pack_mu);
  if (g->alt_pack_t != pack_t) {
    if (g->alt_pack_t != 0) {  // another threshold before: retired, not freed (a launch of another thread may still read it)
      g->retired.push_back(g->alt_pack_ptr);
      g->retired.push_back(g->alt_long_items);
      g->alt_pack_ptr = g->alt_long_items = nullptr;
      g->alt_pack_t = 0;
    }
    int32_t *pp = nullptr, *li = nullptr;
    int64_t np = 0, nl = 0;
    if (int rc = build_pack_set(g, s, pack_t, &pp, &li, &np, &nl)) return rc;
    g->alt_pack_ptr = pp;
    g->alt_long_items = li;
    g->alt_num_packs = np;
    g->alt_num_long_items = nl;
    g->alt_pack_t = pack_t;
  }
  *out = PackView{g->alt_pack_ptr, g->alt_long_items, g->alt_num_packs, g->alt_num_long_items};
  return HET_OK;
}

int grouping_hub_items(const het_grouping* g, const het_grouping* twin, int R, int hub_min, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_pack_mu);
  if (g->num_hub_items >= 0) {
    if (g->hub_twin_serial == twin->serial && g->hub_min == hub_min) return HET_OK;
    // paired with another twin object before (a cache that evicted and rebuilt the grouping by key alone) or another threshold:
    // the lists are rebuilt.  The old ones are RETIRED, not freed: another thread may be between its workspace query and its
    // launch with the pointers it read (and a caller's allocator would hand the memory out again without waiting for the
    // device); they go with the grouping (het_grouping_destroy).
    for (int32_t* old : {g->hub_items, g->hub_segs, g->hub_order, reinterpret_cast<int32_t*>(g->hub_rec)})
      if (old) g->retired.push_back(old);
    g->hub_items = g->hub_segs = g->hub_order = nullptr;
    g->hub_rec = nullptr;
    g->num_hub_items = -1;
    g->num_hub_segs = 0;
  }
  const int64_t NI = g->num_items, TS = twin->S;
  int32_t *items = nullptr, *segs = nullptr, *order = nullptr;
  int4* rec = nullptr;
  int32_t h_num[2] = {0, 0};
  if (NI > 0 && TS > 0) {
    Scratch tmp(s);
    uint8_t *is_hub = nullptr, *is_long = nullptr;
    int32_t *d_num = nullptr, *sel = nullptr, *sel2 = nullptr;
    HET_HIP(tmp.alloc((void**)&is_hub, (size_t)NI));
    HET_HIP(tmp.alloc((void**)&is_long, (size_t)TS));
    HET_HIP(tmp.alloc((void**)&d_num, sizeof(int32_t) * 2));
    HET_HIP(tmp.alloc((void**)&sel, sizeof(int32_t) * (size_t)NI));
    HET_HIP(tmp.alloc((void**)&sel2, sizeof(int32_t) * (size_t)TS));
    hipLaunchKernelGGL(HET_grouping_hub_flags, dim3(blocks_for(NI)), dim3(256), 0, s, g->seg_ptr, g->seg_key, g->item_seg, NI, R,
                       twin->seg_key, twin->seg_ptr, TS, hub_min, is_hub);
    HET_LAUNCH_CHECK("HET_grouping_hub_flags");
    hipLaunchKernelGGL(HET_grouping_long_seg_flags, dim3(blocks_for(TS)), dim3(256), 0, s, twin->seg_ptr, TS, hub_min, is_long);
    HET_LAUNCH_CHECK("HET_grouping_long_seg_flags");
    hipcub::CountingInputIterator<int32_t> ranks(0);
    size_t tb = 0, tb2 = 0;
    HET_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb, ranks, is_hub, sel, d_num, (int)NI, s));
    HET_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb2, ranks, is_long, sel2, d_num + 1, (int)TS, s));
    void* t0 = nullptr;
    HET_HIP(tmp.alloc(&t0, tb > tb2 ? tb : tb2));
    HET_HIP(hipcub::DeviceSelect::Flagged(t0, tb, ranks, is_hub, sel, d_num, (int)NI, s));
    HET_HIP(hipcub::DeviceSelect::Flagged(t0, tb2, ranks, is_long, sel2, d_num + 1, (int)TS, s));
    HET_HIP(hipMemcpyAsync(h_num, d_num, sizeof(h_num), hipMemcpyDeviceToHost, s));
    HET_HIP(hipStreamSynchronize(s));
    HET_HIP(het_malloc_e((void**)&items, sizeof(int32_t) * ((size_t)h_num[0] + 1), s));
    hipError_t e = het_malloc_e((void**)&segs, sizeof(int32_t) * ((size_t)h_num[1] + 1), s);
    if (e == hipSuccess) e = hipMemcpyAsync(items, sel, sizeof(int32_t) * (size_t)h_num[0], hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(segs, sel2, sizeof(int32_t) * (size_t)h_num[1], hipMemcpyDeviceToDevice, s);
    // launch order of the hub items: by the first payload0 (feat row) of the item, so that items in flight together read the
    // same window of the table (within a run the rows ascend when the positions are in (relation, source) order)
    if (e == hipSuccess && h_num[0] > 0 && twin->p0) {
      e = het_malloc_e((void**)&order, sizeof(int32_t) * (size_t)h_num[0], s);
      uint32_t *k_in = nullptr, *k_out = nullptr;
      int32_t* v_in = nullptr;
      if (e == hipSuccess) e = tmp.alloc((void**)&k_in, sizeof(uint32_t) * (size_t)h_num[0]);
      if (e == hipSuccess) e = tmp.alloc((void**)&k_out, sizeof(uint32_t) * (size_t)h_num[0]);
      if (e == hipSuccess) e = tmp.alloc((void**)&v_in, sizeof(int32_t) * (size_t)h_num[0]);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(HET_grouping_hub_first_p0, dim3(blocks_for(h_num[0])), dim3(256), 0, s, items, (int64_t)h_num[0], g->item_begin,
                           twin->p0, k_in, v_in);
        e = hipGetLastError();
      }
      size_t sb = 0;
      void* st = nullptr;
      if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, sb, k_in, k_out, v_in, order, h_num[0], 0, 32, s);
      if (e == hipSuccess) e = tmp.alloc(&st, sb);
      if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(st, sb, k_in, k_out, v_in, order, h_num[0], 0, 32, s);
    }
    if (e == hipSuccess && h_num[1] > 0) {
      e = het_malloc_e((void**)&rec, sizeof(int4) * (size_t)h_num[1], s);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(HET_grouping_hub_records, dim3(blocks_for(h_num[1])), dim3(256), 0, s, segs, (int64_t)h_num[1], twin->seg_key,
                           g->seg_key, g->S, g->item_seg, NI, items, (int64_t)h_num[0], R, rec);
        e = hipGetLastError();
      }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)het_free_e(items); (void)het_free_e(segs); (void)het_free_e(order); (void)het_free_e(rec); HET_HIP(e); }
  }
  g->hub_rec = rec;
  g->hub_order = order;
  g->hub_items = items;
  g->hub_segs = segs;
  g->num_hub_segs = h_num[1];
  g->hub_twin_serial = twin->serial;
  g->hub_min = hub_min;
  g->num_hub_items = h_num[0];
  return HET_OK;
}

namespace {
__global__ __launch_bounds__(256) void HET_grouping_pack_ids(const int32_t* __restrict__ key, const int32_t* __restrict__ p0,
                                                             const int32_t* __restrict__ p1, int64_t E, int2* __restrict__ o2,
                                                             int4* __restrict__ o4) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j <= E; j += (int64_t)gridDim.x * 256) {
    const int64_t jc = j < E ? j : (E > 0 ? E - 1 : 0);
    const int a = p0 ? p0[jc] : 0, b = p1 ? p1[jc] : 0;
    if (o2 && j < E) o2[j] = make_int2(a, b);
    if (o4) o4[j] = make_int4(j < E ? key[j] : -1, a, b, 0);
  }
}
}  // namespace

namespace {
struct TagThr { int v[7]; };
// dev_ptrs (optional, [R+1] int64 on the device): the relation boundaries (thr unused); R > 8: searched per rank
__global__ __launch_bounds__(256) void HET_grouping_tag_kp01(int4* __restrict__ kp01, int64_t E, int which, TagThr thr,
                                                             const idx_t* __restrict__ dev_ptrs, int R) {
  if (dev_ptrs && R <= 8) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const idx_t v = k + 1 < R ? dev_ptrs[k + 1] : 0x7fffffffll;
      thr.v[k] = v < 0x7fffffffll ? (int)v : 0x7fffffff;
    }
  }
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < E; j += (int64_t)gridDim.x * 256) {
    const int4 c = kp01[j], n = kp01[j + 1];  // (record E is the sentinel: key -1)
    const int prev_key = j > 0 ? kp01[j - 1].x : -2;
    const int val = which == 0 ? c.x : c.y;
    int rel = 0;
    if (dev_ptrs && R > 8) {
      rel = find_segment(dev_ptrs, R, (idx_t)val);
    } else {
#pragma unroll
      for (int k = 0; k < 7; ++k) rel += val >= thr.v[k] ? 1 : 0;
    }
    int tag = rel << HET_TAG_REL_SHIFT;
    if (c.x != prev_key) tag |= HET_TAG_FIRST_KEY;
    if (n.x != c.x) tag |= HET_TAG_LAST_KEY | HET_TAG_LAST_RUN;
    else if (n.z != c.z) tag |= HET_TAG_LAST_RUN;
    reinterpret_cast<int*>(kp01 + j)[3] = tag;  // (only this word is written: the neighbours' reads of x / y / z do not race)
  }
}
}  // namespace

int grouping_tag_kp01(const het_grouping* g, int which, const int* thr, hipStream_t s) {
  if (int rc = grouping_packed_ids(g, true, s)) return rc;
  std::lock_guard<std::mutex> lk(g_pack_mu);
  if (g->E == 0 || !g->kp01) return HET_OK;
  if (!thr && g->tag_which >= 0) return HET_OK;  // (a user of the segment / run flags alone: they do not depend on the thresholds)
  TagThr t;
  for (int k = 0; k < 7; ++k) t.v[k] = thr ? thr[k] : 0x7fffffff;
  thr = t.v;
  bool same = g->tag_which == which;
  for (int k = 0; k < 7 && same; ++k) same = g->tag_thr[k] == thr[k];
  if (same) return HET_OK;
  hipLaunchKernelGGL(HET_grouping_tag_kp01, dim3(blocks_for(g->E)), dim3(256), 0, s, g->kp01, g->E, which, t, nullptr, 0);
  HET_LAUNCH_CHECK("HET_grouping_tag_kp01");
  HET_HIP(hipStreamSynchronize(s));  // published only once complete
  for (int k = 0; k < 7; ++k) g->tag_thr[k] = thr[k];
  g->tag_which = which;
  g->tag_dev_src = nullptr;
  return HET_OK;
}

int grouping_tag_kp01_dev(const het_grouping* g, int which, const idx_t* rel_ptrs_dev, int R, hipStream_t s) {
  if (int rc = grouping_packed_ids(g, true, s)) return rc;
  std::lock_guard<std::mutex> lk(g_pack_mu);
  if (g->E == 0 || !g->kp01) return HET_OK;
  if (g->tag_which == which && g->tag_dev_src == rel_ptrs_dev && g->tag_dev_R == R) return HET_OK;
  TagThr t{};
  hipLaunchKernelGGL(HET_grouping_tag_kp01, dim3(blocks_for(g->E)), dim3(256), 0, s, g->kp01, g->E, which, t, rel_ptrs_dev, R);
  HET_LAUNCH_CHECK("HET_grouping_tag_kp01");
  HET_HIP(hipStreamSynchronize(s));  // published only once complete
  for (int k = 0; k < 7; ++k) g->tag_thr[k] = -1;  // (the values stay on the device)
  g->tag_which = which;
  g->tag_dev_src = rel_ptrs_dev;
  g->tag_dev_R = R;
  return HET_OK;
}

namespace {
__global__ void HET_grouping_value_keys(const idx_t* __restrict__ values, int64_t n, uint32_t* __restrict__ keys, int32_t* __restrict__ idx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    keys[i] = (uint32_t)values[i];
    idx[i] = (int32_t)i;
  }
}
}  // namespace

int grouping_value_order(const het_grouping* g, const idx_t* values, int64_t n, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_pack_mu);
  if (g->val_order && g->val_order_src == values && g->val_order_n == n) return HET_OK;
  HET_REQUIRE(values && n > 0 && n < (1ll << 31), "grouping_value_order: bad arguments");
  if (g->val_order) {  // built from another list before: retired, not freed (see grouping_hub_items)
    g->retired.push_back(g->val_order);
    g->val_order = nullptr;
  }
  int32_t* order = nullptr;
  HET_HIP(het_malloc_e((void**)&order, sizeof(int32_t) * (size_t)n, s));
  hipError_t e = hipSuccess;
  {
    Scratch tmp(s);
    uint32_t *k_in = nullptr, *k_out = nullptr;
    int32_t* v_in = nullptr;
    e = tmp.alloc((void**)&k_in, sizeof(uint32_t) * (size_t)n);
    if (e == hipSuccess) e = tmp.alloc((void**)&k_out, sizeof(uint32_t) * (size_t)n);
    if (e == hipSuccess) e = tmp.alloc((void**)&v_in, sizeof(int32_t) * (size_t)n);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(HET_grouping_value_keys, dim3(blocks_for(n)), dim3(256), 0, s, values, n, k_in, v_in);
      e = hipGetLastError();
    }
    size_t sb = 0;
    void* st = nullptr;
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, sb, k_in, k_out, v_in, order, (int)n, 0, 32, s);  // (stable)
    if (e == hipSuccess) e = tmp.alloc(&st, sb);
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(st, sb, k_in, k_out, v_in, order, (int)n, 0, 32, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // published only once complete
  }
  if (e != hipSuccess) { (void)het_free_e(order); HET_HIP(e); }
  g->val_order = order;
  g->val_order_src = values;
  g->val_order_n = n;
  return HET_OK;
}

int grouping_packed_ids(const het_grouping* g, bool with_keys, hipStream_t s) {
  if (with_keys)
    if (int rc = grouping_packs(g, s)) return rc;
  std::lock_guard<std::mutex> lk(g_pack_mu);
  if ((with_keys ? (void*)g->kp01 : (void*)g->p01) || g->E == 0) return HET_OK;
  const int64_t E = g->E;
  int2* o2 = nullptr;
  int4* o4 = nullptr;
  if (with_keys) {
    HET_REQUIRE(g->key_of_rank, "grouping_packed_ids: the grouping has no packs (no segments?)");
    HET_HIP(het_malloc_e((void**)&o4, sizeof(int4) * (size_t)(E + 1), s));
  } else {
    HET_HIP(het_malloc_e((void**)&o2, sizeof(int2) * (size_t)E, s));
  }
  hipLaunchKernelGGL(HET_grouping_pack_ids, dim3(blocks_for(E + 1)), dim3(256), 0, s, g->key_of_rank, g->p0, g->p1, E, o2, o4);
  hipError_t e = hipGetLastError();

  if (e == hipSuccess) e = hipStreamSynchronize(s);  // published only once complete
  if (e != hipSuccess) {
    (void)het_free_e(o2); (void)het_free_e(o4);
    HET_HIP(e);
  }
  if (with_keys) g->kp01 = o4; else g->p01 = o2;
  return HET_OK;
}
