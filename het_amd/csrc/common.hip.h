// Shared device/host helpers for libhet_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/het_amd.h"

typedef int64_t idx_t;

#define HET_WAVE 64

// ---- host-side error plumbing ------------------------------------------------
void het_set_error(const char* fmt, ...);

#define HET_REQUIRE(cond, ...)             \
  do {                                     \
    if (!(cond)) {                         \
      het_set_error(__VA_ARGS__);          \
      return HET_ERR_INVALID_ARG;          \
    }                                      \
  } while (0)

#define HET_LAUNCH_CHECK(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      het_set_error("%s: kernel launch failed: %s", name, hipGetErrorString(e__));    \
      return HET_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)

#define HET_HIP(call)                                                                 \
  do {                                                                                \
    hipError_t e__ = (call);                                                          \
    if (e__ != hipSuccess) {                                                          \
      het_set_error("%s failed: %s", #call, hipGetErrorString(e__));                  \
      return HET_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)

// ---- device memory the library keeps (groupings, layout scratch): hipMalloc / hipFree, or the caller's allocator ---------
// (het_set_allocator, include/het_amd.h; capi.hip).  A pointer is released through the allocator it came from.
int het_dev_alloc(void** out, size_t bytes, hipStream_t s);  // HET_OK, or HET_ERR_HIP with het_last_error() set
void het_dev_free(void* p);                                  // NULL is fine
// hipMalloc / hipFree-shaped forms for code written against hipError_t (the failing call's message is in het_last_error())
static inline hipError_t het_malloc_e(void** out, size_t bytes, hipStream_t s) {
  return het_dev_alloc(out, bytes, s) == HET_OK ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t het_free_e(void* p) { het_dev_free(p); return hipSuccess; }
#define HET_ALLOC(ptr, bytes, s)                                          \
  do {                                                                    \
    int rc__ = het_dev_alloc((void**)&(ptr), (size_t)(bytes), (s));       \
    if (rc__ != HET_OK) return rc__;                                      \
  } while (0)

// ---- optional per-kernel timing (het_kernel_timing_*; capi.hip) ---------------------
bool het_ktime_on();
void het_ktime_begin(const char* name, hipStream_t s);
void het_ktime_end(hipStream_t s);
struct HetKTimer {  // scope guard: { HET_KTIME("HET_kernel", s); hipLaunchKernelGGL(...); }
  hipStream_t s;
  bool on;
  HetKTimer(const char* name, hipStream_t st) : s(st), on(het_ktime_on()) { if (on) het_ktime_begin(name, s); }
  ~HetKTimer() { if (on) het_ktime_end(s); }
};
#define HET_KTIME(name, s) HetKTimer het_ktimer__(name, s)

// ---- fork / join onto the library's side stream -------------------------------------------------------------------
// Independent launches of ONE entry point that are bound by latency rather than by a saturated unit (the short- and the
// long-segment gather passes, the pack-form and the hub passes of the RGAT forward) run side by side: the side stream (one per
// device, made on first use) starts after everything enqueued on the caller's stream so far and the caller's stream waits for
// it before the entry point returns -- to the caller the call is still ordered on `stream` alone.  HET_SIDE_STREAM=0 runs
// everything on the caller's stream.  (Works inside a stream capture: event record / wait become graph dependencies.)
hipStream_t het_side_stream();  // NULL when switched off or when it could not be made
hipEvent_t het_fork_event();    // this thread's event for the current device (made on first use, never destroyed), or NULL
struct HetFork {
  hipStream_t main, side;
  hipEvent_t ev = nullptr;
  // The event is the calling thread's own and lives for the life of the process: destroying an event that a stream capture has
  // seen leaves the runtime with a dangling pointer until the capture ends (hipStreamEndCapture crashed now and then).
  explicit HetFork(hipStream_t m) : main(m), side(het_side_stream()) {
    if (!side) { side = main; return; }
    ev = het_fork_event();
    if (!ev || hipEventRecord(ev, main) != hipSuccess || hipStreamWaitEvent(side, ev, 0) != hipSuccess) {
      (void)hipGetLastError();
      ev = nullptr;
      side = main;  // no fork: the side launches simply follow on the caller's stream
    }
  }
  // the caller's stream waits for the side stream's work so far (idempotent: the destructor joins whatever an early
  // return -- a failed launch check between fork and join -- left on the side stream, so the caller never frees a workspace
  // the side stream still uses, and a stream capture never ends with an unjoined fork)
  bool joined = false;
  hipError_t join() {
    if (joined || side == main || !ev) { joined = true; return hipSuccess; }
    joined = true;
    hipError_t e = hipEventRecord(ev, side);
    if (e == hipSuccess) e = hipStreamWaitEvent(main, ev, 0);
    return e;
  }
  ~HetFork() { if (!joined && join() != hipSuccess) (void)hipGetLastError(); }
  HetFork(const HetFork&) = delete;
  HetFork& operator=(const HetFork&) = delete;
};

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// LDS a workgroup of the current device may ask for (160 KB on gfx950, 64 KB on the other gfx9 parts, the runtime's
// hipDeviceAttributeMaxSharedMemoryPerBlock elsewhere; 64 KB when nothing can be queried) -- the `_ok` predicates and the launchers of the weight-resident kernels both use it,
// so a build for another target (the Makefile's ARCH override) falls back instead of failing at the launch.  node_sum.hip.
// Wave-private LDS tiles are written by some lanes and read by others without a workgroup barrier (LDS instructions of one wave
// execute in order).  The compiler, however, reasons per lane: a lane that did NOT store may have its next load of the same address
// replaced by the value it loaded before (seen in round 5: HET_node_fwd's second staging phase multiplied the first phase's
// fragments in the lanes that had not staged).  A wavefront-scope fence costs no instruction and forbids that.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

size_t het_lds_budget();
int64_t het_num_cus();  // compute units of the current device (capi.hip)

// ---- device helpers -----------------------------------------------------------
// Segment s with ptrs[s] <= i < ptrs[s+1]; ptrs non-decreasing, empty segments allowed.
__device__ __forceinline__ int find_segment(const idx_t* __restrict__ ptrs, int n, idx_t i) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ptrs[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// Index of `key` in the sorted list a[0..n) (the key is present by construction).
__device__ __forceinline__ idx_t lower_bound_idx(const idx_t* __restrict__ a, idx_t n, idx_t key) {
  idx_t lo = 0, hi = n;
  while (lo < hi) {
    idx_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Row of a compact (relation, node) tensor for one edge end
// (reference: find_relational_compact_as_of_node_index, include/kernel_enums.h:101-119).
//   kind 0: eid;  kind 1/3: a = rel_ptrs of the unique list, b = its node ids (binary search);
//   kind 4: a = inverse index by edata idx.
__device__ __forceinline__ idx_t compact_row(int kind, const idx_t* __restrict__ a, const idx_t* __restrict__ b,
                                             int rel, idx_t node, idx_t eid) {
  if (kind == HET_KIND_DISABLED) return eid;
  if (kind == HET_KIND_DUAL_LIST_DIRECT_INDEX) return a[eid];
  idx_t base = a[rel];
  return base + lower_bound_idx(b + base, a[rel + 1] - base, node);
}

// Map a global tile id onto (relation, row range): relation r owns ceil(n_r / tile_rows) tiles.
// All lanes execute the same uniform loop (scalar loads).
__device__ __forceinline__ bool tile_to_relation(const idx_t* __restrict__ rel_ptrs, int R, int tile_rows, long t,
                                                 int& r, idx_t& row_begin, idx_t& row_end) {
  long acc = 0;
  for (int i = 0; i < R; ++i) {
    idx_t a = rel_ptrs[i], b = rel_ptrs[i + 1];
    long nt = (b - a + tile_rows - 1) / tile_rows;
    if (t < acc + nt) {
      r = i;
      row_begin = a + (t - acc) * tile_rows;
      row_end = row_begin + tile_rows < b ? row_begin + tile_rows : b;
      return true;
    }
    acc += nt;
  }
  return false;
}

// Non-temporal 16-byte accesses.  Same-box A/B (exp/README.md): they pay where a kernel SCATTERS whole rows of a
// once-written [E, X] tensor in random order -- the RGAT backward's grad_feat stores: 4.12 ms plain, 3.36 ms with nt
// stores, 3.21 ms with nt stores + nt loads of the once-read feat rows -- and they cost where rows are written in
// (nearly) sequential order (segment broadcast: +0.4 ms per launch) or gathered by a pure read kernel (aggregation:
// +0.4 ms; segment sum: +0.05 ms) or where the scattered stores are 4..16 bytes (grad_el / tbuf / broadcast dots:
// +0.1..0.25 ms per step), so only the backward's row accesses use them.
typedef float het_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  const het_f4v t = __builtin_nontemporal_load(reinterpret_cast<const het_f4v*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  het_f4v t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<het_f4v*>(p));
}

// ---- byte offsets off a wave-uniform base --------------------------------------------------------------------------------
// A gather written as base[row * X + x] with a 64-bit row costs three VALU instructions per address (sign extension, 64-bit shift,
// 64-bit add: v_ashrrev_i32 + v_lshlrev_b64 + v_lshl_add_u64) and a VGPR pair; a table below 4 GiB addressed by an unsigned 32-bit
// BYTE offset costs one (v_lshl_or_b32) and the load takes the base from scalar registers (global_load_dwordx4 v, v_off, s[base:+1]).
// The gather kernels are bound by their instruction streams, so their row kernels take the offset type as a
// template argument: uint32_t when the host has checked every table they index, uint64_t otherwise.
template <typename O>
__device__ __forceinline__ float4 ld4_at(const float* base, O byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename O>
__device__ __forceinline__ float2 ld2_at(const float* base, O byte_off) {
  return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename O>
__device__ __forceinline__ float ld1_at(const float* base, O byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename O>
__device__ __forceinline__ void st4_at(float* base, O byte_off, float4 v) {
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
template <typename O>
__device__ __forceinline__ void st1_at(float* base, O byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
constexpr int het_log2_ce(int v) { return v <= 1 ? 0 : 1 + het_log2_ce(v >> 1); }
// rows * row_bytes fits an unsigned 32-bit byte offset (with room for the lane's piece of the row)
static inline bool het_fits_u32(int64_t rows, int64_t row_bytes) { return rows >= 0 && rows * row_bytes <= 0xffffffffll - 4096; }

__device__ __forceinline__ float leaky_exp(float z, float slope) {
  // gatLeakyReluExp, DGLHackKernel/GAT/FusedGAT.cu.h:23-26.
  // The grouped backward recovers the leaky-ReLU branch from the stored value (slope >= 0: z > 0 <=> exp(..) > 1).  For
  // 0 < z < 6e-8 expf(z) rounds to exactly 1.0f; it is stored as the next float up (exp(z) = 1 + z lies between the two,
  // so either is a correctly rounded neighbour) which keeps that equivalence exact.
  if (z > 0.f) {
    const float v = expf(z);
    return v > 1.f ? v : 1.00000012f;
  }
  return expf(slope * z);
}
