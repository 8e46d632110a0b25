// Library-level entry points: error string, build info.
#include <stdarg.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hip.h"

static thread_local char g_err[512] = "";

void het_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* het_last_error(void) { return g_err; }

#ifndef HET_GIT_SHA
#define HET_GIT_SHA "unknown"
#endif

extern "C" const char* het_build_info(void) {
  return "het_amd (libhet_amd.so) git " HET_GIT_SHA " | target gfx950 (MI355X, CDNA4) | hipcc " __VERSION__
         " | built " __DATE__ " " __TIME__;
}

// ---- device memory ---------------------------------------------------------------------
namespace {
std::mutex g_amu;
het_alloc_fn g_alloc = nullptr;
het_free_fn g_free = nullptr;
void* g_alloc_user = nullptr;
struct Owner { het_free_fn free; void* user; };
std::unordered_map<void*, Owner> g_external;  // pointers that came from a caller's allocator (a few dozen per grouping)
}  // namespace

extern "C" int het_set_allocator(het_alloc_fn alloc, het_free_fn free_, void* user) {
  HET_REQUIRE((alloc == nullptr) == (free_ == nullptr), "het_set_allocator: pass both functions or neither");
  std::lock_guard<std::mutex> lk(g_amu);
  g_alloc = alloc;
  g_free = free_;
  g_alloc_user = user;
  return HET_OK;
}

extern "C" int het_allocator_is_external(void) {
  std::lock_guard<std::mutex> lk(g_amu);
  return g_alloc != nullptr;
}

int het_dev_alloc(void** out, size_t bytes, hipStream_t s) {
  if (!bytes) bytes = 8;
  het_alloc_fn a;
  het_free_fn f;
  void* u;
  {
    std::lock_guard<std::mutex> lk(g_amu);
    a = g_alloc; f = g_free; u = g_alloc_user;
  }
  if (a) {
    void* p = a(bytes, (het_stream)s, u);
    if (!p) {
      het_set_error("het_dev_alloc: the caller's allocator (het_set_allocator) returned NULL for %zu bytes", bytes);
      return HET_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(g_amu);
    g_external[p] = Owner{f, u};
    *out = p;
    return HET_OK;
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    het_set_error("hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    *out = nullptr;
    return HET_ERR_HIP;
  }
  return HET_OK;
}

void het_dev_free(void* p) {
  if (!p) return;
  Owner o{nullptr, nullptr};
  {
    std::lock_guard<std::mutex> lk(g_amu);
    auto it = g_external.find(p);
    if (it != g_external.end()) { o = it->second; g_external.erase(it); }
  }
  if (o.free) o.free(p, o.user); else (void)hipFree(p);
}

// ---- per-kernel timing ---------------------------------------------------------------
namespace {
struct KRec { const char* name; hipEvent_t a, b; };
std::mutex g_kmu;
std::vector<KRec> g_krecs;
std::atomic<bool> g_kon{false};
thread_local int g_kcur = -1;  // record opened by this thread's current launch

void krecs_clear() {
  for (auto& r : g_krecs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_krecs.clear();
}
}  // namespace

bool het_ktime_on() { return g_kon.load(std::memory_order_relaxed); }

void het_ktime_begin(const char* name, hipStream_t s) {
  KRec r{name, nullptr, nullptr};
  g_kcur = -1;
  if (hipEventCreate(&r.a) != hipSuccess) return;
  if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return; }
  (void)hipEventRecord(r.a, s);
  std::lock_guard<std::mutex> lk(g_kmu);
  g_krecs.push_back(r);
  g_kcur = (int)g_krecs.size() - 1;
}

void het_ktime_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_kmu);
  if (g_kcur >= 0 && g_kcur < (int)g_krecs.size()) (void)hipEventRecord(g_krecs[g_kcur].b, s);
  g_kcur = -1;
}

extern "C" int het_kernel_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_kmu);
  if (on) krecs_clear();
  g_kon.store(on != 0);
  return HET_OK;
}

hipStream_t het_side_stream() {
  static const bool off = [] { const char* v = getenv("HET_SIDE_STREAM"); return v && v[0] == '0'; }();
  if (off) return nullptr;
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!streams[dev] && hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    streams[dev] = nullptr;
  }
  return streams[dev];
}

hipEvent_t het_fork_event() {
  thread_local hipEvent_t events[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!events[dev] && hipEventCreateWithFlags(&events[dev], hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    events[dev] = nullptr;
  }
  return events[dev];
}

extern "C" int het_kernel_timing_read(const char* name_prefix, double* total_ms, int64_t* launches) {
  HET_REQUIRE(name_prefix && total_ms && launches, "het_kernel_timing_read: null argument");
  std::lock_guard<std::mutex> lk(g_kmu);
  const size_t n = strlen(name_prefix);
  double t = 0.0;
  int64_t c = 0;
  for (auto& r : g_krecs) {
    if (strncmp(r.name, name_prefix, n) != 0) continue;
    HET_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    HET_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    t += ms;
    ++c;
  }
  *total_ms = t;
  *launches = c;
  return HET_OK;
}

// ---- LDS budget of the current device (common.hip.h) ---------------------------------------------------------------------
// compute units of the current device (cached per device; a benign race writes the same value twice)
int64_t het_num_cus() {
  static int cache[64];
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  if (dev >= 0 && dev < 64 && cache[dev] > 0) return cache[dev];
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
  if (dev >= 0 && dev < 64) cache[dev] = n;
  return n;
}

size_t het_lds_budget() {
  static thread_local int cached_dev = -1;
  static thread_local size_t cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 64 * 1024; }
  if (dev != cached_dev) {
    // by architecture name first: the attribute reports the 64 KB a launch gets WITHOUT hipFuncAttributeMaxDynamicSharedMemorySize on
    // some runtimes, and the kernels here opt in to the whole LDS of a gfx950 CU (160 KB)
    hipDeviceProp_t prop;
    size_t v = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) v = 160 * 1024;
      else if (strncmp(prop.gcnArchName, "gfx9", 4) == 0) v = 64 * 1024;
    } else {
      (void)hipGetLastError();
    }
    if (v == 0) {
      int a = 0;
      if (hipDeviceGetAttribute(&a, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || a <= 0) { (void)hipGetLastError(); a = 64 * 1024; }
      v = (size_t)a;
    }
    cached = v;
    cached_dev = dev;
  }
  return cached;
}

