// libtorch_hrt.so for ROCm: the reference's op registration, compiled, over the C ABI of include/het_amd.h.
//
// The reference binds its kernels with one shared object that registers every op through TORCH_LIBRARY_FRAGMENT(torch_hrt, m)
// (hrt/include/DGLHackKernel/OpExport/*.inc.h: RGNNOps.inc.h:1188-1206, RGATOps.inc.h:560-571, RGCNOps.inc.h, HGTOps*.inc.h,
// DataConverters.inc.h; generated export list hrt/buildutils/genutils/gen_torch_export.py) and loads it with
//     torch.ops.load_library(".../libtorch_hrt.so")          (hrt/python/kernels/__init__.py:4-16)
// This file is that object for MI355X: same op names, same positional arguments, same Dict(str, Tensor) keys.  Every op unwraps
// its tensors to device pointers and calls ONE entry point of libhet_amd.so on the current HIP stream; the only state kept here
// is a small cache of the device-side groupings (het_grouping_create) the entry points take as an optional fast path, keyed by
// the identity of the index tensors they were built from -- what het_amd/plan.py does for the Python registration.
// Build: make -C het_amd/csrc torch_hrt   (hipcc + the torch headers of the running interpreter; links libhet_amd.so).
#include <ATen/ATen.h>
#include <ATen/DeviceGuard.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/het_amd.h"

namespace {

using at::Tensor;
using Dict = c10::Dict<std::string, Tensor>;

inline het_stream stream_of(const Tensor& t) { return (het_stream)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
// Every compute op starts with HET_ON_DEVICE_OF(<one of its GPU tensors>): the launches, the library's side stream and its fork /
// join events key on the CURRENT device, so tensors on another GPU of the process make that GPU current for the op (what the
// Python registration does with torch.cuda.device(tensor.device)).
#define HET_ON_DEVICE_OF(t) const c10::OptionalDeviceGuard het_device_guard__(at::device_of(t))

// The library's own device memory (groupings, construction scratch) from torch's caching allocator (include/het_amd.h:
// het_set_allocator): inside torch.cuda.memory_allocated, back in torch's pool when a grouping is dropped, no hipFree on an op's path.
void* torch_alloc(size_t bytes, het_stream stream, void*) {
  try {
    return c10::hip::HIPCachingAllocator::raw_alloc_with_stream(bytes, (hipStream_t)stream);
  } catch (...) {
    return nullptr;  // (out of memory: the library reports it through its error code)
  }
}
void torch_free(void* p, void*) {
  try { c10::hip::HIPCachingAllocator::raw_delete(p); } catch (...) {}
}
const bool g_allocator_installed = [] {
  const char* v = getenv("HET_TORCH_ALLOCATOR");
  if (v && v[0] == '0') return false;
  return het_set_allocator(torch_alloc, torch_free, nullptr) == HET_OK;
}();
inline void check(int rc, const char* op) { TORCH_CHECK(rc == HET_OK, op, ": ", het_last_error()); }

inline const float* fp(const Tensor& t) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(), "torch_hrt: expected a contiguous float32 GPU tensor");
  return t.data_ptr<float>();
}
inline float* fpw(Tensor& t) { return const_cast<float*>(fp(t)); }
inline const int64_t* ip(const Tensor& t) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kLong && t.is_contiguous(), "torch_hrt: expected a contiguous int64 GPU tensor");
  return t.data_ptr<int64_t>();
}
inline const Tensor& key(const Dict& d, const char* k) {
  auto it = d.find(k);
  TORCH_CHECK(it != d.end(), "torch_hrt: missing dict key ", k);
  return it->value();
}

// ---- grouping cache -------------------------------------------------------------------------------------------------
struct Ident {
  const void* p; int64_t n; uint32_t v;
  bool operator==(const Ident& o) const { return p == o.p && n == o.n && v == o.v; }
};
inline Ident ident(const Tensor* t) {
  if (!t || !t->defined()) return {nullptr, 0, 0};
  return {t->data_ptr(), t->numel(), (uint32_t)t->_version()};
}
// A grouping is shared between the cache and every op call that is using it: eviction only drops the cache's reference, the
// object goes when the last op that holds it has enqueued its launches (a concurrent call on another thread -- the model thread
// and an autograd worker -- can no longer destroy a grouping under a launch that is about to read its index arrays).
using GroupingRef = std::shared_ptr<het_grouping>;
struct Entry {
  Ident a, b, c, d; int64_t bound; int dev;
  GroupingRef g;
  std::vector<Tensor> keep;  // the source tensors stay alive with the grouping (a data_ptr cannot be recycled meanwhile)
};
std::mutex g_mu;
// (never destructed: at process exit the groupings would be released into a caching allocator that may already be gone)
std::list<Entry>& g_cache = *new std::list<Entry>();
constexpr size_t kMaxEntries = 24;
bool groupings_enabled() {
  static const bool on = [] { const char* v = getenv("HET_SHIM_GROUPINGS"); return !(v && v[0] == '0'); }();
  return on;
}

// Grouping of the positions of `keys` by (relation, key); by key alone without rel_ptrs.  NULL when disabled.  Hold the
// returned reference until the op's launches are enqueued.  HET_SHIM_GROUPING_CACHE=<n>: entries kept (24; one-shot graphs --
// sampled blocks -- want few).
GroupingRef grouping(const Tensor* rel_ptrs, const Tensor& keys, int64_t key_bound, const Tensor* p0, const Tensor* p1) {
  if (!groupings_enabled()) return nullptr;
  static const size_t max_entries = [] { const char* v = getenv("HET_SHIM_GROUPING_CACHE"); const long n = v ? atol(v) : 0; return n > 0 ? (size_t)n : kMaxEntries; }();
  const Ident a = ident(rel_ptrs), b = ident(&keys), c = ident(p0), d = ident(p1);
  const int dev = keys.device().index();
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
    if (it->a == a && it->b == b && it->c == c && it->d == d && it->bound == key_bound && it->dev == dev) {
      g_cache.splice(g_cache.begin(), g_cache, it);
      het_grouping_note_stream(g_cache.front().g.get(), stream_of(keys));  // (destroy orders the release after this stream's use)
      return g_cache.front().g;
    }
  het_grouping* raw = nullptr;
  check(het_grouping_create(rel_ptrs ? ip(*rel_ptrs) : nullptr, rel_ptrs ? rel_ptrs->numel() - 1 : 0, ip(keys), keys.numel(), key_bound,
                            p0 ? ip(*p0) : nullptr, p1 ? ip(*p1) : nullptr, stream_of(keys), &raw),
        "het_grouping_create");
  GroupingRef g(raw, [](het_grouping* q) { het_grouping_destroy(q); });
  Entry e{a, b, c, d, key_bound, dev, g, {}};
  if (rel_ptrs) e.keep.push_back(*rel_ptrs);
  e.keep.push_back(keys);
  if (p0) e.keep.push_back(*p0);
  if (p1) e.keep.push_back(*p1);
  g_cache.push_front(std::move(e));
  while (g_cache.size() > max_entries) g_cache.pop_back();  // (drops the cache's reference only)
  return g;
}
Tensor workspace(int64_t floats, const Tensor& like) {
  return at::empty({floats > 0 ? floats : 1}, like.options().dtype(at::kFloat));
}

// relation of every position of a relation-bucketed list (derived index tensor; cached through the grouping that uses it)
Tensor rel_by_position(const Tensor& rel_ptrs, int64_t n) {
  const int64_t R = rel_ptrs.numel() - 1;
  return at::repeat_interleave(at::arange(R, rel_ptrs.options()), rel_ptrs.slice(0, 1) - rel_ptrs.slice(0, 0, R), 0, n).contiguous();
}
struct Derived { Ident a; Tensor t; std::vector<Tensor> keep; };
std::list<Derived> g_derived;
Tensor cached_rel_by_position(const Tensor& rel_ptrs, int64_t n) {
  const Ident a = ident(&rel_ptrs);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& d : g_derived)
      if (d.a == a && d.t.numel() == n) return d.t;
  }
  Tensor t = rel_by_position(rel_ptrs, n);
  std::lock_guard<std::mutex> lk(g_mu);
  g_derived.push_front(Derived{a, t, {rel_ptrs}});
  while (g_derived.size() > 8) g_derived.pop_back();
  return t;
}

// Derived index tensors of a graph (row of every edge position in a unique (relation, node) list, edge-id -> row maps ...):
// built once with ATen ops, looked up by the identity of ALL the tensors they were derived from (as het_amd/kernels.py does).
struct DerivedSet { std::string tag; std::vector<Ident> ids; std::vector<Tensor> out, keep; };
std::list<DerivedSet> g_sets;
template <class F>
std::vector<Tensor> derived(const char* tag, std::vector<const Tensor*> src, F build) {
  std::vector<Ident> ids;
  for (const Tensor* t : src) ids.push_back(ident(t));
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto it = g_sets.begin(); it != g_sets.end(); ++it)
      if (it->tag == tag && it->ids == ids) {
        g_sets.splice(g_sets.begin(), g_sets, it);
        return g_sets.front().out;
      }
  }
  std::vector<Tensor> out = build();
  DerivedSet e{tag, ids, out, {}};
  for (const Tensor* t : src)
    if (t && t->defined()) e.keep.push_back(*t);
  std::lock_guard<std::mutex> lk(g_mu);
  g_sets.push_front(std::move(e));
  while (g_sets.size() > 24) g_sets.pop_back();
  return out;
}

// row of (relation of the position, node) in a unique (relation, node) list: ua = its relation pointers, ub = its node ids
Tensor rows_by_search(const Tensor& rel_of_pos, const Tensor& nodes, const Tensor& ua, const Tensor& ub) {
  const int64_t Ru = ua.numel() - 1;
  const int64_t bound = std::max<int64_t>(nodes.numel() ? nodes.max().item<int64_t>() : 0, ub.numel() ? ub.max().item<int64_t>() : 0) + 1;
  Tensor rel_u = at::repeat_interleave(at::arange(Ru, ua.options()), ua.slice(0, 1) - ua.slice(0, 0, Ru));
  return at::searchsorted(rel_u * bound + ub, rel_of_pos * bound + nodes).contiguous();
}

// ---- info + layout converters (DataConverters.inc.h) ------------------------------------------------------------------
// The dispatcher does not bump the version counter of a custom op's `(a!)` arguments, and the library writes through raw pointers:
// a cache keyed by (data_ptr, numel, version) -- the sorted exp stream of a4 / a5 below, scale_in_rank_order in the Python
// registration -- would keep serving a copy of a buffer that another op of this library has refilled in place.  Every
// op bumps what it is about to write FIRST, so that what it then records about its own outputs carries the new version.
void will_write(std::initializer_list<Tensor> ts) {
  for (const Tensor& t : ts)
    if (t.defined() && !t.is_inference()) t.unsafeGetTensorImpl()->bump_version();
}

void build_debug_info() { printf("%s\n", het_build_info()); }

Tensor dev64(const Tensor& t) { return t.to(at::kLong).to(at::Device(at::kCUDA)).contiguous(); }
std::vector<Tensor> back(std::vector<Tensor> v, const Tensor& like) {
  if (!like.is_cuda())
    for (auto& t : v) t = t.cpu();
  return v;
}

std::vector<Tensor> separate_coo(const Tensor& row_in, const Tensor& col_in, const Tensor& rel_in, const Tensor& eids_in, int64_t num_rels) {
  Tensor row = dev64(row_in), col = dev64(col_in), rel = dev64(rel_in), eids = dev64(eids_in);
  const int64_t E = row.numel();
  Tensor rp = at::empty({num_rels + 1}, row.options()), r = at::empty_like(row), c = at::empty_like(row), e = at::empty_like(row);
  const int64_t bound = E ? eids.max().item<int64_t>() + 1 : 1;
  check(het_layout_separate_coo(ip(row), ip(col), ip(rel), ip(eids), E, num_rels, bound, rp.data_ptr<int64_t>(), r.data_ptr<int64_t>(),
                                c.data_ptr<int64_t>(), e.data_ptr<int64_t>(), stream_of(row)), "convert_integrated_coo_to_separate_coo");
  return back({rp, r, c, e}, row_in);
}
std::vector<Tensor> coo_to_csr(const Tensor& row, const Tensor& col, const Tensor& rel, const Tensor& eids, int64_t num_rows) {
  const int64_t E = row.numel();
  Tensor ptrs = at::empty({num_rows + 1}, row.options()), c = at::empty_like(col), r = at::empty_like(rel), e = at::empty_like(eids);
  check(het_layout_coo_to_csr(ip(row), ip(col), ip(rel), ip(eids), E, num_rows, ptrs.data_ptr<int64_t>(), c.data_ptr<int64_t>(),
                              r.data_ptr<int64_t>(), e.data_ptr<int64_t>(), stream_of(row)), "het_layout_coo_to_csr");
  return {ptrs, c, r, e};
}
Tensor csr_rows(const Tensor& row_ptrs) {
  const int64_t n = row_ptrs.numel() - 1;
  return at::repeat_interleave(at::arange(n, row_ptrs.options()), row_ptrs.slice(0, 1) - row_ptrs.slice(0, 0, n)).contiguous();
}
int64_t rel_count(const Tensor& rel) { return rel.numel() ? rel.max().item<int64_t>() + 1 : 1; }

std::vector<Tensor> transpose_csr(const Tensor& row_ptrs_in, const Tensor& col_in, const Tensor& eids_in, const Tensor& rel_in) {
  // DataConverters.inc.h:283-344: (row_ptrs, col_indices, eids, rel_types) of the transpose
  Tensor row_ptrs = dev64(row_ptrs_in), col = dev64(col_in), eids = dev64(eids_in), rel = dev64(rel_in);
  const int64_t n = row_ptrs.numel() - 1, E = col.numel();
  Tensor ptrs = at::empty({n + 1}, col.options()), c = at::empty_like(col), e = at::empty_like(col), r = at::empty_like(col);
  check(het_layout_transpose_csr(ip(row_ptrs), ip(col), ip(eids), ip(rel), n, E, n, ptrs.data_ptr<int64_t>(), c.data_ptr<int64_t>(),
                                 e.data_ptr<int64_t>(), r.data_ptr<int64_t>(), stream_of(col)), "transpose_csr");
  return back({ptrs, c, e, r}, row_ptrs_in);
}
std::vector<Tensor> convert_integrated_coo_to_separate_coo(const Tensor& row, const Tensor& col, const Tensor& rel, const Tensor& eids,
                                                           int64_t /*num_nodes*/, int64_t num_rels) {
  return separate_coo(row, col, rel, eids, num_rels);  // DataConverters.inc.h:216-281 -> MyHyb.h:1047-1096
}
std::vector<Tensor> convert_integrated_csr_to_separate_coo(const Tensor& row_ptrs, const Tensor& col, const Tensor& rel, const Tensor& eids) {
  Tensor rp = dev64(row_ptrs), r = dev64(rel);  // DataConverters.inc.h:10-77 -> MyHyb.h:1099-1150
  return back(separate_coo(csr_rows(rp), dev64(col), r, dev64(eids), rel_count(r)), row_ptrs);
}
std::vector<Tensor> separate_csr(const Tensor& rows, const Tensor& cols, const Tensor& rels, const Tensor& eids, int64_t num_rows,
                                 int64_t num_rels, const Tensor& like) {
  // a CSR over the composite row (relation, row): stable, so a row keeps its edges in input order
  auto v = coo_to_csr((rels * num_rows + rows).contiguous(), cols, rels, eids, num_rows * num_rels);
  Tensor rel_ptrs = v[0].slice(0, 0, v[0].numel(), num_rows).contiguous();
  return back({rel_ptrs, v[0], v[1], v[3]}, like);
}
std::vector<Tensor> convert_integrated_csr_to_separate_csr(const Tensor& row_ptrs, const Tensor& col, const Tensor& rel, const Tensor& eids) {
  Tensor rp = dev64(row_ptrs), r = dev64(rel);  // DataConverters.inc.h:79-145
  return separate_csr(csr_rows(rp), dev64(col), r, dev64(eids), rp.numel() - 1, rel_count(r), row_ptrs);
}
std::vector<Tensor> convert_integrated_coo_to_separate_csr(const Tensor& row, const Tensor& col, const Tensor& rel, const Tensor& eids,
                                                           int64_t num_nodes, int64_t num_rels) {
  return separate_csr(dev64(row), dev64(col), dev64(rel), dev64(eids), num_nodes, num_rels, row);  // DataConverters.inc.h:147-214
}

// ---- a1 / a2 / a3: segment GEMMs (RGNNOps.inc.h) ------------------------------------------------------------------------
struct Lists { const Tensor* rp; const Tensor* g; const Tensor* s; };
Lists matmul_lists(const Dict& d, int64_t kind) {
  if (kind == 0) return {&key(d, "separate_coo_rel_ptrs"), &key(d, "separate_coo_node_indices"), &key(d, "separate_coo_eids")};
  TORCH_CHECK(kind == 1, "rgnn_relational_matmul: CompactAsOfNodeKind ", kind, " not supported (the reference asserts, RGNNOps.inc.h:292-294)");
  return {&key(d, "unique_srcs_and_dests_rel_ptrs"), &key(d, "unique_srcs_and_dests_node_indices"), nullptr};
}

void rgnn_relational_matmul(Dict d, int64_t kind, Tensor W, Tensor x, Tensor ret, bool in1head) {
  will_write({ret});
  HET_ON_DEVICE_OF(ret);
  const Lists l = matmul_lists(d, kind);
  const int64_t R = W.size(0), H = W.size(1), K = W.size(2), D = W.size(3), X = H * D;
  const bool mfma = (K == 32 || K == 64 || K == 128) && (X == 32 || X == 64 || X == 128);
  GroupingRef g;
  Tensor ws;
  if (kind == 0 && in1head && ((D == 1 && (H & (H - 1)) == 0) || (D > 1 && !mfma && (X & (X - 1)) == 0 && X <= 256)) &&
      l.g->numel() > 0 && l.g->data_ptr() != l.s->data_ptr()) {
    g = grouping(l.rp, *l.g, x.size(0), l.s, nullptr);
    if (g) ws = workspace(std::max<int64_t>(1, het_grouping_num_segments(g.get())) * X, ret);
  }
  check(het_rgnn_relational_matmul(kind, ip(*l.rp), R, ip(*l.g), l.s ? ip(*l.s) : nullptr, l.g->numel(), fp(W), fp(x), fpw(ret), H, K, D,
                                   in1head, g.get(), ws.defined() ? ws.data_ptr() : nullptr, ws.defined() ? ws.numel() * 4 : 0, stream_of(ret)),
        "rgnn_relational_matmul");
}

void backward_rgnn_relational_matmul(Dict d, int64_t kind, Tensor Wt, Tensor x, Tensor gradout, Tensor grad_x, Tensor grad_w, bool in1head) {
  will_write({grad_x, grad_w});
  HET_ON_DEVICE_OF(gradout);
  const Lists l = matmul_lists(d, kind);
  const int64_t R = Wt.size(0), H = Wt.size(1), D = Wt.size(2), K = Wt.size(3);
  GroupingRef g;
  Tensor ws;
  if (kind == 0 && l.g->data_ptr() != l.s->data_ptr() && (in1head || D > 1)) {
    g = grouping(l.rp, *l.g, x.size(0), l.s, nullptr);
    if (g) ws = workspace(std::max<int64_t>(1, het_grouping_num_segments(g.get())) * H * D, gradout);
  }
  check(het_backward_rgnn_relational_matmul(kind, ip(*l.rp), R, ip(*l.g), l.s ? ip(*l.s) : nullptr, l.g->numel(), x.size(0), fp(Wt), fp(x),
                                            fp(gradout), fpw(grad_x), fpw(grad_w), H, K, D, in1head, HET_ACC_ADD, g.get(),
                                            ws.defined() ? ws.data_ptr() : nullptr, ws.defined() ? ws.numel() * 4 : 0, stream_of(gradout)),
        "backward_rgnn_relational_matmul");
}

void rgnn_relational_matmul_no_scatter_gather_list(Tensor offsets, Tensor W, Tensor x, Tensor ret) {
  will_write({ret});
  HET_ON_DEVICE_OF(ret);
  const int64_t T = W.size(0), H = W.size(1), K = W.size(2), D = W.size(3), n = x.size(0);
  const int per_head = H > 1 && x.numel() == n * H * K;
  check(het_rgnn_relational_matmul_no_scatter_gather_list(ip(offsets), T, n, fp(W), fp(x), fpw(ret), H, K, D, per_head, stream_of(ret)),
        "rgnn_relational_matmul_no_scatter_gather_list");
}
void backward_rgnn_relational_matmul_no_scatter_gather_list(Tensor offsets, Tensor Wt, Tensor x, Tensor gradout, Tensor grad_x, Tensor grad_w) {
  will_write({grad_x, grad_w});
  HET_ON_DEVICE_OF(gradout);
  const int64_t T = Wt.size(0), H = Wt.size(1), D = Wt.size(2), K = Wt.size(3), n = x.size(0);
  const int per_head = H > 1 && x.numel() == n * H * K;
  check(het_backward_rgnn_relational_matmul_no_scatter_gather_list(ip(offsets), T, n, fp(Wt), fp(x), fp(gradout), fpw(grad_x), fpw(grad_w), H,
                                                                   K, D, per_head, 1, stream_of(gradout)),
        "backward_rgnn_relational_matmul_no_scatter_gather_list");
}

// ---- a4 / a5 / a6: fused GAT (RGATOps.inc.h) ------------------------------------------------------------------------------
struct Maps { const Tensor* m[4]; };
Maps gat_maps(int64_t kind, const Dict& d) {  // key names as the launchers read them (RGATOps.inc.h:180-236 forward, :476-540 backward)
  if (kind == 0) return {{nullptr, nullptr, nullptr, nullptr}};
  if (kind == 1) {
    const Tensor *rp = &key(d, "unique_srcs_and_dests_rel_ptrs"), *n = &key(d, "unique_srcs_and_dests_node_indices");
    return {{rp, n, rp, n}};
  }
  if (kind == 3) {
    const Tensor* rpc = d.contains("unique_srcs_and_dests_rel_ptrs_col") ? &key(d, "unique_srcs_and_dests_rel_ptrs_col")
                                                                         : &key(d, "unique_srcs_and_dests_rel_col");
    return {{&key(d, "unique_srcs_and_dests_rel_ptrs"), &key(d, "unique_srcs_and_dests_node_indices_row"), rpc,
             &key(d, "unique_srcs_and_dests_node_indices_col")}};
  }
  if (kind == 4) return {{&key(d, "edata_idx_to_inverse_idx_row"), nullptr, &key(d, "edata_idx_to_inverse_idx_col"), nullptr}};
  if (kind == 2) {  // one inverse index for both edge ends, as the reference's non-dual direct-indexing branch reads it
    const Tensor* m = &key(d, "edata_idx_to_inverse_idx");
    return {{m, nullptr, m, nullptr}};
  }
  TORCH_CHECK(false, "relational_fused_gat: CompactAsOfNodeKind ", kind, " is not supported");
}
inline const int64_t* mp(const Tensor* t) { return t ? ip(*t) : nullptr; }
inline bool gat_grouped_shape_ok(int64_t H, int64_t D) {
  const int64_t X = H * D;
  return D >= 4 && (D & (D - 1)) == 0 && (X & (X - 1)) == 0 && X / 4 <= 64;
}

// The compact kinds on the grouped kernels (as het_amd/kernels.py::_gat_direct / _by_dst): every kind becomes the direct-index
// kind 4 once per graph -- {feat row, er row} of every edge POSITION and of every edge ID -- so the kernels read rows without
// searching.  out: {srow_by_position, drow_by_position, row_of_eid, col_row_of_eid}
std::vector<Tensor> direct_rows(int64_t kind, const Maps& m, const Tensor& rel_ptrs, const Tensor& row, const Tensor& col, const Tensor& eids) {
  return derived(kind == 4 ? "gat4" : (kind == 2 ? "gat2" : (kind == 1 ? "gat1" : "gat3")),
                 {m.m[0], m.m[1], m.m[2], m.m[3], &rel_ptrs, &row, &col, &eids}, [&]() -> std::vector<Tensor> {
    Tensor srow, drow;
    if (kind == 4 || kind == 2) {
      srow = m.m[0]->index({eids}).contiguous();
      drow = m.m[2]->index({eids}).contiguous();
      return {srow, drow, *m.m[0], *m.m[2]};
    }
    Tensor relp = rel_by_position(rel_ptrs, eids.numel());
    srow = rows_by_search(relp, row, *m.m[0], *m.m[1]);
    drow = rows_by_search(relp, col, *m.m[2], *m.m[3]);
    const int64_t n = eids.numel() ? eids.max().item<int64_t>() + 1 : 0;
    Tensor mr = at::empty({n}, eids.options()), mc = at::empty({n}, eids.options());
    mr.index_put_({eids}, srow);
    mc.index_put_({eids}, drow);
    return {srow, drow, mr, mc};
  });
}

// ---- the op-level a5 streams the sorted copy of exp its a4 left (as het_amd/kernels.py: _sorted_stream_*) -------------------
// Between the two reference-named calls the destination-sorted copy of exp that the grouped forward can write is kept, keyed by
// the identity of the tensors the backward would otherwise gather by edge id (exp, el, er) and of the edge lists; weak
// references make a recycled address a miss, _version an in-place edit.  A miss = the gathers.  HET_A5_SORTED_STREAM=0: off.
struct SortedStream {
  Ident exp, el, er, eids, col; double slope; int dev;
  Tensor exs;
  c10::weak_intrusive_ptr<c10::TensorImpl> w_exp, w_el, w_er;
};
std::list<SortedStream>& g_streams = *new std::list<SortedStream>();
bool sorted_stream_on() {
  static const bool on = [] { const char* v = getenv("HET_A5_SORTED_STREAM"); return !(v && v[0] == '0'); }();
  return on;
}
void sorted_stream_put(const Tensor& exp, const Tensor& el, const Tensor& er, const Tensor& eids, const Tensor& col, double slope, Tensor exs) {
  const Ident a = ident(&exp), b = ident(&el), c = ident(&er), d = ident(&eids), e = ident(&col);
  const int dev = exp.device().index();
  std::lock_guard<std::mutex> lk(g_mu);
  g_streams.remove_if([&](const SortedStream& q) { return q.exp.p == a.p && q.exp.n == a.n && q.dev == dev; });
  if (!exs.defined()) return;
  g_streams.push_front(SortedStream{a, b, c, d, e, slope, dev, std::move(exs), c10::weak_intrusive_ptr<c10::TensorImpl>(exp.getIntrusivePtr()),
                                    c10::weak_intrusive_ptr<c10::TensorImpl>(el.getIntrusivePtr()),
                                    c10::weak_intrusive_ptr<c10::TensorImpl>(er.getIntrusivePtr())});
  while (g_streams.size() > 2) g_streams.pop_back();
}
Tensor sorted_stream_get(const Tensor& exp, const Tensor& el, const Tensor& er, const Tensor& eids, const Tensor& col, double slope) {
  if (!sorted_stream_on() || !groupings_enabled()) return Tensor();
  const Ident a = ident(&exp), b = ident(&el), c = ident(&er), d = ident(&eids), e = ident(&col);
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& q : g_streams)
    if (q.exp == a && q.el == b && q.er == c && q.eids == d && q.col == e && q.slope == slope && q.dev == exp.device().index())
      return (q.w_exp.expired() || q.w_el.expired() || q.w_er.expired()) ? Tensor() : q.exs;
  return Tensor();
}

void gat_forward(const Tensor& eids, const Tensor& rel_ptrs, const Tensor& row, const Tensor& col, int64_t kind, const Maps& m_in,
                 const Tensor& feat, const Tensor& el, const Tensor& er, Tensor& sum, Tensor& exp, Tensor& ret, double slope) {
  const int64_t E = eids.numel(), N = ret.size(0), H = sum.size(1);
  const int64_t D = feat.numel() ? feat.numel() / (feat.size(0) * H) : ret.numel() / std::max<int64_t>(1, N * H);
  GroupingRef g;
  Maps m = m_in;
  std::vector<Tensor> dr;
  Tensor exs;
  const bool kind0 = kind == 0;
  if (kind == 0 && E > 0) {  // positions by destination; payload0 = edge id, payload1 = relation of the position
    Tensor relp = cached_rel_by_position(rel_ptrs, E);
    g = grouping(nullptr, col, N, &eids, &relp);
    if (g && sorted_stream_on() && slope >= 0 && gat_grouped_shape_ok(H, D)) exs = at::empty({E, H}, exp.options());
  } else if (kind != 0 && E > 0 && groupings_enabled() && gat_grouped_shape_ok(H, D)) {  // payload1 = feat row of the position
    dr = direct_rows(kind, m_in, rel_ptrs, row, col, eids);
    g = grouping(nullptr, col, N, &eids, &dr[0]);
    m = Maps{{&dr[2], nullptr, &dr[3], nullptr}};
    kind = 4;
  }
  check(het_relational_fused_gat_separate_coo(ip(eids), ip(rel_ptrs), ip(row), ip(col), rel_ptrs.numel() - 1, E, N, kind, mp(m.m[0]), mp(m.m[1]),
                                              mp(m.m[2]), mp(m.m[3]), fp(feat), fp(el), fp(er), fpw(sum), fpw(exp), fpw(ret),
                                              exs.defined() ? exs.data_ptr<float>() : nullptr, H, D, slope, g.get(), nullptr, nullptr,
                                              stream_of(ret)),
        "relational_fused_gat_separate_coo");
  if (kind0) sorted_stream_put(exp, el, er, eids, col, slope, exs);
}
void gat_backward(const Tensor& eids, const Tensor& rel_ptrs, const Tensor& row, const Tensor& col, int64_t kind, const Maps& m_in,
                  const Tensor& feat, const Tensor& el, const Tensor& er, const Tensor& sum, const Tensor& exp, const Tensor& ret,
                  const Tensor& gradout, Tensor& gfeat, Tensor& gel, Tensor& ger, double slope) {
  const int64_t E = eids.numel(), N = ret.size(0), H = sum.size(1), D = ret.numel() / std::max<int64_t>(1, N * H);
  GroupingRef g, gs, gd;
  Maps m = m_in;
  std::vector<Tensor> dr;
  Tensor ws;
  Tensor exs;
  if (kind == 0 && E > 0) {
    Tensor relp = cached_rel_by_position(rel_ptrs, E);
    g = grouping(nullptr, col, N, &eids, &relp);
    if (g) exs = sorted_stream_get(exp, el, er, eids, col, slope);
  } else if (kind != 0 && E > 0 && groupings_enabled() && gat_grouped_shape_ok(H, D) && slope >= 0) {
    // by feat row (payloads: edge id, destination) and by er row (payload: edge id): the compact backward sums a row's gradient in
    // registers and stores it once (csrc/fused_gat_grouped.hip) instead of E*H*D float atomics
    dr = direct_rows(kind, m_in, rel_ptrs, row, col, eids);
    gs = grouping(nullptr, dr[0], feat.size(0), &eids, &col);
    gd = grouping(nullptr, dr[1], er.size(0), &eids, nullptr);
    if (gs && gd) {
      ws = workspace(N * 2 * H + E * H, ret);
      m = Maps{{&dr[2], nullptr, &dr[3], nullptr}};
      kind = 4;
    } else {
      gs.reset(); gd.reset();
    }
  }
  check(het_backward_relational_fused_gat_separate_coo(ip(eids), ip(rel_ptrs), ip(row), ip(col), rel_ptrs.numel() - 1, E, N, kind, mp(m.m[0]),
                                                       mp(m.m[1]), mp(m.m[2]), mp(m.m[3]), fp(feat), fp(el), fp(er), fp(sum), fp(exp), fp(ret),
                                                       exs.defined() ? exs.data_ptr<float>() : nullptr, fp(gradout), fpw(gfeat), fpw(gel),
                                                       fpw(ger), H, D, slope, g.get(), gs.get(), gd.get(),
                                                       feat.size(0), er.size(0), ws.defined() ? ws.data_ptr() : nullptr,
                                                       ws.defined() ? ws.numel() * 4 : 0, nullptr, nullptr, nullptr, nullptr, stream_of(ret)),
        "backward_relational_fused_gat_separate_coo");
}

// CompactAsOfNodeFlag of the CSR pair (RGATOps.inc.h:251-277, 430-460): feat / el / er live on the rows of ONE unique (relation, node)
// list and every edge end is looked up by (relation of the edge, node).  Once per graph: the direct-index maps of kind 4.
// out: {rows of the CSR expanded per position, feat row of every edge id, er row of every edge id}
std::vector<Tensor> csr_compact_maps(const Tensor& row_ptr, const Tensor& col, const Tensor& eids, const Tensor& reltypes, const Tensor& urp,
                                     const Tensor& unodes, bool rows_are_dst) {
  return derived(rows_are_dst ? "csrc_in" : "csrc_out", {&row_ptr, &col, &eids, &reltypes, &urp, &unodes}, [&]() -> std::vector<Tensor> {
    Tensor rows = csr_rows(row_ptr);
    const Tensor& src = rows_are_dst ? col : rows;
    const Tensor& dst = rows_are_dst ? rows : col;
    Tensor srow = rows_by_search(reltypes, src, urp, unodes), drow = rows_by_search(reltypes, dst, urp, unodes);
    const int64_t n = eids.numel() ? eids.max().item<int64_t>() + 1 : 0;
    Tensor mr = at::empty({n}, eids.options()), mc = at::empty({n}, eids.options());
    mr.index_put_({eids}, srow);
    mc.index_put_({eids}, drow);
    return {rows, mr, mc};
  });
}

void relational_fused_gat_separate_coo(Tensor eids, Tensor rel_ptrs, Tensor row, Tensor col, int64_t kind, Dict d, Tensor feat, Tensor el,
                                       Tensor er, Tensor sum, Tensor exp, Tensor ret, double slope) {
  will_write({sum, exp, ret});
  HET_ON_DEVICE_OF(ret);
  gat_forward(eids, rel_ptrs, row, col, kind, gat_maps(kind, d), feat, el, er, sum, exp, ret, slope);
}
void backward_relational_fused_gat_separate_coo(Tensor eids, Tensor rel_ptrs, Tensor row, Tensor col, int64_t kind, Dict d, Tensor feat, Tensor el,
                                                Tensor er, Tensor sum, Tensor exp, Tensor ret, Tensor gradout, Tensor gfeat, Tensor gel,
                                                Tensor ger, double slope) {
  will_write({gfeat, gel, ger});
  HET_ON_DEVICE_OF(ret);
  gat_backward(eids, rel_ptrs, row, col, kind, gat_maps(kind, d), feat, el, er, sum, exp, ret, gradout, gfeat, gel, ger, slope);
}

void relational_fused_gat_csr(Tensor row_ptr, Tensor col, Tensor eids, Tensor reltypes, Tensor urp, Tensor unodes, Tensor feat, Tensor el, Tensor er,
                              Tensor sum, Tensor exp, Tensor ret, double slope, bool compact) {
  will_write({sum, exp, ret});
  HET_ON_DEVICE_OF(ret);
  const int64_t N = row_ptr.numel() - 1, E = eids.numel(), H = el.size(1), D = ret.numel() / std::max<int64_t>(1, N * H);
  if (!compact && groupings_enabled() && E > 0 && gat_grouped_shape_ok(H, D)) {
    // the in-CSR IS the edge list grouped by destination: (eids, src, dst) in CSR order through the destination-grouped kernels
    Tensor dst = csr_rows(row_ptr), rp1 = at::tensor({(int64_t)0, E}, row_ptr.options());
    gat_forward(eids, rp1, col, dst, 0, gat_maps(0, Dict()), feat, el, er, sum, exp, ret, slope);
    return;
  }
  if (compact && groupings_enabled() && E > 0 && gat_grouped_shape_ok(H, D)) {
    // compact rows: the separate-COO pair with CompactAsOfNodeKind 4 once every edge id knows its feat row and er row
    std::vector<Tensor> cm = csr_compact_maps(row_ptr, col, eids, reltypes, urp, unodes, true);
    Tensor rp1 = at::tensor({(int64_t)0, E}, row_ptr.options());
    gat_forward(eids, rp1, col, cm[0], 4, Maps{{&cm[1], nullptr, &cm[2], nullptr}}, feat, el, er, sum, exp, ret, slope);
    return;
  }
  check(het_relational_fused_gat_csr(ip(row_ptr), ip(col), ip(eids), ip(reltypes), N, E, ip(urp), ip(unodes), std::max<int64_t>(0, urp.numel() - 1),
                                     fp(feat), fp(el), fp(er), fpw(sum), fpw(exp), fpw(ret), H, D, slope, compact, stream_of(ret)),
        "relational_fused_gat_csr");
}
void backward_relational_fused_gat_csr(Tensor row_ptr, Tensor col, Tensor eids, Tensor reltypes, Tensor urp, Tensor unodes, Tensor feat, Tensor el,
                                       Tensor er, Tensor sum, Tensor exp, Tensor ret, Tensor gradout, Tensor gfeat, Tensor gel, Tensor ger,
                                       double slope, bool compact) {
  will_write({gfeat, gel, ger});
  HET_ON_DEVICE_OF(ret);
  const int64_t N = row_ptr.numel() - 1, E = eids.numel(), H = el.size(1), D = ret.numel() / std::max<int64_t>(1, N * H);
  if (!compact && groupings_enabled() && E > 0 && gat_grouped_shape_ok(H, D) && slope >= 0) {
    Tensor src = csr_rows(row_ptr), rp1 = at::tensor({(int64_t)0, E}, row_ptr.options());  // out-CSR rows are the sources
    gat_backward(eids, rp1, src, col, 0, gat_maps(0, Dict()), feat, el, er, sum, exp, ret, gradout, gfeat, gel, ger, slope);
    return;
  }
  if (compact && groupings_enabled() && E > 0 && gat_grouped_shape_ok(H, D) && slope >= 0) {
    std::vector<Tensor> cm = csr_compact_maps(row_ptr, col, eids, reltypes, urp, unodes, false);  // out-CSR rows are the sources
    Tensor rp1 = at::tensor({(int64_t)0, E}, row_ptr.options());
    // "+=" contract of the reference-named op: the grouped kind-4 kernels overwrite, so they run into temporaries that are added
    Tensor gf = at::zeros_like(gfeat), gl = at::zeros_like(gel), gr = at::zeros_like(ger);
    gat_backward(eids, rp1, cm[0], col, 4, Maps{{&cm[1], nullptr, &cm[2], nullptr}}, feat, el, er, sum, exp, ret, gradout, gf, gl, gr, slope);
    gfeat.add_(gf);
    gel.add_(gl);
    ger.add_(gr);
    return;
  }
  check(het_backward_relational_fused_gat_csr(ip(row_ptr), ip(col), ip(eids), ip(reltypes), N, E, ip(urp), ip(unodes),
                                              std::max<int64_t>(0, urp.numel() - 1), fp(feat), fp(el), fp(er), fp(sum), fp(exp), fp(ret),
                                              fp(gradout), fpw(gfeat), fpw(gel), fpw(ger), H, D, slope, compact, stream_of(ret)),
        "backward_relational_fused_gat_csr");
}

// ---- a7 / a8 / a9: RGCN (RGCNOps.inc.h) ----------------------------------------------------------------------------------
void rgcn_layer1_separate_coo(Tensor rel_ptrs, Tensor eids, Tensor row, Tensor col, Tensor x, Tensor W, Tensor norm, Tensor out) {
  will_write({out});
  HET_ON_DEVICE_OF(out);
  const int64_t R = W.size(0), K = W.size(1), D = W.size(2), N = out.size(0);
  GroupingRef g = grouping(&rel_ptrs, col, N, &row, &eids);
  Tensor ws = g ? workspace(std::max<int64_t>(1, het_grouping_num_segments(g.get())) * K, W) : Tensor();
  check(het_rgcn_layer1_separate_coo(ip(rel_ptrs), ip(eids), ip(row), ip(col), R, eids.numel(), N, fp(x), fp(W), fp(norm), fpw(out), K, D, g.get(),
                                     g ? ws.data_ptr() : nullptr, g ? ws.numel() * 4 : 0, stream_of(out)),
        "rgcn_layer1_separate_coo");
}
void backward_rgcn_layer1_separate_coo(Tensor rel_ptrs, Tensor eids, Tensor row, Tensor col, Tensor x, Tensor Wt, Tensor norm, Tensor grad_norm,
                                       Tensor grad_x, Tensor gradout, Tensor grad_w) {
  will_write({grad_norm, grad_x, grad_w});
  HET_ON_DEVICE_OF(gradout);
  const int64_t R = Wt.size(0), D = Wt.size(1), K = Wt.size(2), N = gradout.size(0);
  GroupingRef g = grouping(&rel_ptrs, row, x.size(0), &col, &eids);
  Tensor ws = g ? workspace(std::max<int64_t>(1, het_grouping_num_segments(g.get())) * D, grad_w) : Tensor();
  check(het_backward_rgcn_layer1_separate_coo(ip(rel_ptrs), ip(eids), ip(row), ip(col), R, eids.numel(), N, fp(x), fp(Wt), fp(norm), fpw(grad_norm),
                                              fpw(grad_x), fp(gradout), fpw(grad_w), K, D, g.get(), g ? ws.data_ptr() : nullptr,
                                              g ? ws.numel() * 4 : 0, stream_of(grad_w)),
        "backward_rgcn_layer1_separate_coo");
}
void rgcn_node_mean_aggregation(Tensor eids, Tensor rel_ptrs, Tensor row, Tensor col, Dict d, Tensor feat, Tensor enorm, Tensor ret, bool direct) {
  will_write({ret});
  HET_ON_DEVICE_OF(ret);
  const Tensor* a = direct ? &key(d, "inverse_indices_row") : &key(d, "rel_ptrs_row");
  const Tensor* b = direct ? nullptr : &key(d, "node_indices_row");
  const int64_t N = ret.size(0);
  check(het_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(ip(eids), ip(rel_ptrs), ip(row), ip(col), rel_ptrs.numel() - 1, eids.numel(),
                                                                       N, ip(*a), mp(b), fp(feat), fp(enorm), fpw(ret),
                                                                       ret.numel() / std::max<int64_t>(1, N), direct, nullptr, stream_of(ret)),
        "rgcn_node_mean_aggregation_compact_as_of_node_separate_coo");
}
void backward_rgcn_node_mean_aggregation(Tensor eids, Tensor rel_ptrs, Tensor row, Tensor col, Dict d, Tensor feat, Tensor enorm, Tensor ret,
                                         Tensor gradout, Tensor gfeat, bool direct) {
  will_write({gfeat});
  HET_ON_DEVICE_OF(ret);
  const Tensor* a = direct ? &key(d, "inverse_indices_row") : &key(d, "rel_ptrs_row");
  const Tensor* b = direct ? nullptr : &key(d, "node_indices_row");
  const int64_t N = ret.size(0);
  check(het_backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(
            ip(eids), ip(rel_ptrs), ip(row), ip(col), rel_ptrs.numel() - 1, eids.numel(), N, ip(*a), mp(b), fp(feat), fp(enorm), fp(ret),
            fp(gradout), fpw(gfeat), ret.numel() / std::max<int64_t>(1, N), direct, nullptr, gfeat.size(0), stream_of(ret)),
        "backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo");
}

// ---- a10 / a11 / a12: HGT (HGTOps.inc.h, HGTOpsEdgeParallel.inc.h, RGNNOps.inc.h:609-658, 1131-1181) -------------------------
struct IpMaps { int64_t kind; const Tensor* a; const Tensor* b; };
IpMaps ip_maps(const Dict& d, int64_t kind) {
  if (kind == 0) return {0, nullptr, nullptr};
  if (kind == 1) return {1, &key(d, "unique_srcs_and_dests_rel_ptrs"), &key(d, "unique_srcs_and_dests_node_indices")};
  TORCH_CHECK(kind == 2, "rgnn_inner_product_right_node: CompactAsOfNodeKind ", kind, " not supported");
  return {2, &key(d, "edata_idx_to_inverse_idx"), nullptr};
}
void rgnn_inner_product_right_node_separatecoo(Dict d, int64_t kind, Tensor rel_ptrs, Tensor eids, Tensor row, Tensor col, Tensor left, Tensor right,
                                               Tensor out) {
  will_write({out});
  HET_ON_DEVICE_OF(out);
  const IpMaps m = ip_maps(d, kind);
  const int64_t H = out.size(1), D = right.numel() / std::max<int64_t>(1, right.size(0) * H);
  check(het_rgnn_inner_product_right_node_separatecoo(m.kind, mp(m.a), mp(m.b), ip(rel_ptrs), ip(eids), ip(row), ip(col), rel_ptrs.numel() - 1,
                                                      eids.numel(), fp(left), fp(right), fpw(out), H, D, stream_of(out)),
        "rgnn_inner_product_right_node_separatecoo");
}
void backward_inner_product_right_node_separatecoo(Dict d, int64_t kind, Tensor rel_ptrs, Tensor eids, Tensor row, Tensor col, Tensor left,
                                                   Tensor right, Tensor gradout, Tensor gleft, Tensor gright) {
  will_write({gleft, gright});
  HET_ON_DEVICE_OF(gradout);
  const IpMaps m = ip_maps(d, kind);
  const int64_t H = gradout.size(1), D = right.numel() / std::max<int64_t>(1, right.size(0) * H);
  GroupingRef g = m.kind == 0 ? grouping(nullptr, row, right.size(0), &eids, &eids) : nullptr;
  check(het_backward_inner_product_right_node_separatecoo(m.kind, mp(m.a), mp(m.b), ip(rel_ptrs), ip(eids), ip(row), ip(col),
                                                          rel_ptrs.numel() - 1, eids.numel(), fp(left), fp(right), fp(gradout), fpw(gleft),
                                                          fpw(gright), H, D, 1, g.get(), nullptr, left.size(0), right.size(0), stream_of(gradout)),
        "backward_inner_product_right_node_separatecoo");
}
GroupingRef by_dst_rel(const Tensor& rel_ptrs, const Tensor& col, const Tensor& eids, int64_t N) {
  if (eids.numel() == 0) return nullptr;
  Tensor relp = cached_rel_by_position(rel_ptrs, eids.numel());
  return grouping(nullptr, col, N, &eids, &relp);
}
void hgt_edge_softmax(Tensor row, Tensor col, Tensor eids, Tensor rel_ptrs, Tensor score, Tensor mu, Tensor sum, Tensor m, Tensor a) {
  will_write({sum, m, a});
  HET_ON_DEVICE_OF(sum);
  const int64_t H = mu.size(1), N = sum.size(0);
  GroupingRef g = H % 4 == 0 ? by_dst_rel(rel_ptrs, col, eids, N) : nullptr;
  check(het_hgt_full_graph_edge_softmax_ops_separate_coo(ip(row), ip(col), ip(eids), ip(rel_ptrs), rel_ptrs.numel() - 1, eids.numel(), N, fp(score),
                                                         fp(mu), fpw(sum), fpw(m), fpw(a), H, g.get(), stream_of(mu)),
        "hgt_full_graph_edge_softmax_ops_separate_coo");
}
void backward_hgt_edge_softmax(Tensor row, Tensor col, Tensor eids, Tensor rel_ptrs, Tensor score, Tensor a, Tensor grad_a, Tensor mu, Tensor gscore,
                               Tensor gmu, Tensor tmp) {
  will_write({gscore, gmu, tmp});
  HET_ON_DEVICE_OF(tmp);
  const int64_t H = mu.size(1), N = tmp.size(0);
  GroupingRef g = H % 4 == 0 ? by_dst_rel(rel_ptrs, col, eids, N) : nullptr;
  check(het_backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(ip(row), ip(col), ip(eids), ip(rel_ptrs), rel_ptrs.numel() - 1,
                                                                                  eids.numel(), N, fp(score), fp(a), fp(grad_a), fp(mu),
                                                                                  fpw(gscore), fpw(gmu), fpw(tmp), H, g.get(), stream_of(mu)),
        "backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo");
}
void hgt_message_mean_aggregation(Tensor rel_ptrs, Tensor eids, Tensor row, Tensor col, Tensor x, Tensor W, Tensor norm, Tensor new_h) {
  will_write({new_h});
  HET_ON_DEVICE_OF(new_h);
  const int64_t R = W.size(0), H = W.size(1), dk = W.size(2), dout = W.size(3);
  GroupingRef g = grouping(&rel_ptrs, col, new_h.size(0), &row, &eids);
  Tensor ws = g ? workspace(std::max<int64_t>(1, het_grouping_num_segments(g.get())) * H * dk, new_h) : Tensor();
  check(het_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(ip(rel_ptrs), ip(eids), ip(row), ip(col), R, eids.numel(),
                                                                                new_h.size(0), fp(x), fp(W), fp(norm), fpw(new_h), H, dk, dout,
                                                                                g.get(), g ? ws.data_ptr() : nullptr, g ? ws.numel() * 4 : 0,
                                                                                stream_of(new_h)),
        "hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo");
}
void backward_hgt_message_mean_aggregation(Tensor rel_ptrs, Tensor eids, Tensor row, Tensor col, Tensor x, Tensor Wt, Tensor norm, Tensor new_h,
                                           Tensor gx, Tensor gw, Tensor gnorm, Tensor gradout) {
  will_write({gx, gw, gnorm});
  HET_ON_DEVICE_OF(gradout);
  const int64_t R = Wt.size(0), H = Wt.size(1), dout = Wt.size(2), dk = Wt.size(3);
  GroupingRef g = grouping(&rel_ptrs, row, x.size(0), &col, &eids);
  Tensor ws = g ? workspace(2 * std::max<int64_t>(1, het_grouping_num_segments(g.get())) * H * dout + Wt.numel(), gradout) : Tensor();
  check(het_backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(
            ip(rel_ptrs), ip(eids), ip(row), ip(col), R, eids.numel(), new_h.size(0), fp(x), fp(Wt), fp(norm), fp(new_h), fpw(gx), fpw(gw),
            fpw(gnorm), fp(gradout), H, dk, dout, g.get(), g ? ws.data_ptr() : nullptr, g ? ws.numel() * 4 : 0, stream_of(gradout)),
        "backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo");
}
void hgt_hetero_attention(Tensor row, Tensor col, Tensor eids, Tensor rel_ptrs, Tensor k, Tensor q, Tensor W, Tensor inner, Tensor score) {
  will_write({inner, score});
  HET_ON_DEVICE_OF(score);
  const int64_t R = W.size(0), H = W.size(1), dk = W.size(2), dout = W.size(3);
  check(het_hgt_full_graph_hetero_attention_ops_coo(ip(row), ip(col), ip(eids), ip(rel_ptrs), R, eids.numel(), fp(k), fp(q), fp(W), fpw(inner),
                                                    fpw(score), H, dk, dout, stream_of(score)),
        "hgt_full_graph_hetero_attention_ops_coo");
}
void backward_hgt_hetero_attention(Tensor /*incsr_row_ptrs*/, Tensor /*incsr_col*/, Tensor /*incsr_eids*/, Tensor /*incsr_rel*/, Tensor row,
                                   Tensor col, Tensor eids, Tensor rel_ptrs, Tensor gW, Tensor Wt, Tensor k, Tensor q, Tensor inner, Tensor gscore,
                                   Tensor gk, Tensor gq) {
  will_write({gW, gk, gq});
  HET_ON_DEVICE_OF(gk);
  const int64_t R = Wt.size(0), H = Wt.size(1), dout = Wt.size(2), dk = Wt.size(3), nq = q.size(0);
  GroupingRef gd = grouping(nullptr, col, nq, &eids, nullptr);
  GroupingRef gs = grouping(&rel_ptrs, row, k.size(0), &col, &eids);
  Tensor ws = gs ? workspace(std::max<int64_t>(1, het_grouping_num_segments(gs.get())) * H * dout, gk) : Tensor();
  check(het_backward_hgt_full_graph_hetero_attention_ops_coo(ip(row), ip(col), ip(eids), ip(rel_ptrs), R, eids.numel(), fpw(gW), fp(Wt), fp(k), fp(q),
                                                             fp(inner), fp(gscore), fpw(gk), fpw(gq), H, dk, dout, gd.get(), gs.get(), nq,
                                                             gs ? ws.data_ptr() : nullptr, gs ? ws.numel() * 4 : 0, stream_of(gk)),
        "backward_hgt_full_graph_hetero_attention_ops_coo");
}

}  // namespace

// One m.def per op of the reference's export list (same names, same argument order).  The reference lets torch INFER the schemas
// from its C++ signatures (OpExport/*.inc.h: `m.def("name", fn)`), and inference ignores the `at::Tensor&` of an output: a build of
// the reference registers `name(Tensor _0, Tensor _1, ...) -> ()` with no alias information at all (checked with a ten-line
// extension against this torch: tests/test_abi.py::test_inferred_schema_of_a_tensor_ref_has_no_alias_info).  Ours are written out,
// with the reference's parameter names and `Tensor(a!)` on every tensor an op writes -- the strings of the Python registration
// (het_amd/kernels.py), character for character (tests/test_abi.py::test_compiled_and_python_schemas_are_identical): positional
// calls, the only kind the reference's Python makes, see no difference, and functionalization / torch.compile see the mutation.
TORCH_LIBRARY_FRAGMENT(torch_hrt, m) {
  m.def("build_debug_info() -> ()",
        build_debug_info);
  m.def("transpose_csr(Tensor row_ptrs, Tensor col_indices, Tensor eids, Tensor rel_types) -> Tensor[]",
        transpose_csr);
  m.def("convert_integrated_coo_to_separate_coo(Tensor row_indices, Tensor col_indices, Tensor rel_types, Tensor eids, "
        "int num_nodes, int num_rels) -> Tensor[]",
        convert_integrated_coo_to_separate_coo);
  m.def("convert_integrated_csr_to_separate_coo(Tensor row_ptrs, Tensor col_indices, Tensor rel_types, Tensor eids) -> Tensor[]",
        convert_integrated_csr_to_separate_coo);
  m.def("convert_integrated_csr_to_separate_csr(Tensor row_ptrs, Tensor col_indices, Tensor rel_types, Tensor eids) -> Tensor[]",
        convert_integrated_csr_to_separate_csr);
  m.def("convert_integrated_coo_to_separate_csr(Tensor row_indices, Tensor col_indices, Tensor rel_types, Tensor eids, "
        "int num_nodes, int num_rels) -> Tensor[]",
        convert_integrated_coo_to_separate_csr);
  m.def("rgnn_relational_matmul(Dict(str, Tensor) args_tensor_dict, int IntKind, Tensor weights, Tensor node_feat, "
        "Tensor(a!) ret, bool InputNumHeadOneFlag) -> ()",
        rgnn_relational_matmul);
  m.def("backward_rgnn_relational_matmul(Dict(str, Tensor) args_tensor_dict, int IntKind, Tensor weights_transposed, "
        "Tensor node_feat, Tensor gradout, Tensor(a!) grad_node_feat, Tensor(b!) grad_weights, bool InputNumHeadOneFlag) -> ()",
        backward_rgnn_relational_matmul);
  m.def("rgnn_relational_matmul_no_scatter_gather_list(Tensor ntype_offset_ptrs, Tensor weights, Tensor inputs, "
        "Tensor(a!) ret) -> ()",
        rgnn_relational_matmul_no_scatter_gather_list);
  m.def("backward_rgnn_relational_matmul_no_scatter_gather_list(Tensor ntype_offset_ptrs, Tensor weights_transposed, "
        "Tensor inputs, Tensor gradout, Tensor(a!) grad_input, Tensor(b!) grad_weights) -> ()",
        backward_rgnn_relational_matmul_no_scatter_gather_list);
  m.def("relational_fused_gat_separate_coo(Tensor separate_coo_eids, Tensor separate_coo_rel_ptrs, "
        "Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, int IntKind, Dict(str, Tensor) args_tensor_dict, "
        "Tensor feat_src, Tensor el, Tensor er, Tensor(a!) sum, Tensor(b!) exp, Tensor(c!) ret, float slope) -> ()",
        relational_fused_gat_separate_coo);
  m.def("backward_relational_fused_gat_separate_coo(Tensor separate_coo_eids, Tensor separate_coo_rel_ptrs, "
        "Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, int IntKind, Dict(str, Tensor) args_tensor_dict, "
        "Tensor feat_src, Tensor el, Tensor er, Tensor sum, Tensor exp, Tensor ret, Tensor gradout, Tensor(a!) grad_feat_src, "
        "Tensor(b!) grad_el, Tensor(c!) grad_er, float slope) -> ()",
        backward_relational_fused_gat_separate_coo);
  m.def("relational_fused_gat_csr(Tensor incsr_row_ptr, Tensor incsr_col_indices, Tensor incsr_eids, Tensor incsr_reltypes, "
        "Tensor unique_srcs_and_dests_rel_ptrs, Tensor unique_srcs_and_dests_node_indices, Tensor feat_src, Tensor el, "
        "Tensor er, Tensor(a!) sum, Tensor(b!) exp, Tensor(c!) ret, float slope, bool CompactAsOfNodeFlag=False) -> ()",
        relational_fused_gat_csr);
  m.def("backward_relational_fused_gat_csr(Tensor outcsr_row_ptr, Tensor outcsr_col_indices, Tensor outcsr_eids, "
        "Tensor outcsr_reltypes, Tensor unique_srcs_and_dests_rel_ptrs, Tensor unique_srcs_and_dests_node_indices, "
        "Tensor feat_src, Tensor el, Tensor er, Tensor sum, Tensor exp, Tensor ret, Tensor gradout, Tensor(a!) grad_feat_src, "
        "Tensor(b!) grad_el, Tensor(c!) grad_er, float slope, bool CompactAsOfNodeFlag=False) -> ()",
        backward_relational_fused_gat_csr);
  m.def("rgcn_layer1_separate_coo(Tensor separate_coo_relptrs, Tensor separate_coo_eids, Tensor separate_coo_row_indices, "
        "Tensor separate_coo_col_indices, Tensor node_feat_input, Tensor weights, Tensor edge_norm, "
        "Tensor(a!) node_feat_output) -> ()",
        rgcn_layer1_separate_coo);
  m.def("backward_rgcn_layer1_separate_coo(Tensor separate_coo_relptrs, Tensor separate_coo_eids, "
        "Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Tensor node_feat_input, Tensor weights_transposed, "
        "Tensor edge_norm, Tensor(a!) grad_edge_norm, Tensor(b!) delta_node_feat_input, Tensor delta_node_feat_output, "
        "Tensor(c!) delta_weights) -> ()",
        backward_rgcn_layer1_separate_coo);
  m.def("rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(Tensor separate_coo_eids, Tensor separate_coo_rel_ptrs, "
        "Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Dict(str, Tensor) args_tensor_dict, "
        "Tensor feat_src, Tensor enorm, Tensor(a!) ret, bool DirectIndexFlag) -> ()",
        rgcn_node_mean_aggregation);
  m.def("backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(Tensor separate_coo_eids, "
        "Tensor separate_coo_rel_ptrs, Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Dict(str, "
        "Tensor) args_tensor_dict, Tensor feat_src, Tensor enorm, Tensor ret, Tensor gradout, Tensor(a!) grad_feat_src, "
        "bool DirectIndexFlag) -> ()",
        backward_rgcn_node_mean_aggregation);
  m.def("rgnn_inner_product_right_node_separatecoo(Dict(str, Tensor) arg_tensor_dict, int IntKind, "
        "Tensor separate_coo_rel_ptrs, Tensor separate_coo_eids, Tensor separate_coo_row_indices, "
        "Tensor separate_coo_col_indices, Tensor left_side_data, Tensor right_node_vectors, "
        "Tensor(a!) edge_inner_product) -> ()",
        rgnn_inner_product_right_node_separatecoo);
  m.def("backward_inner_product_right_node_separatecoo(Dict(str, Tensor) arg_tensor_dict, int IntKind, "
        "Tensor separate_coo_rel_ptrs, Tensor separate_coo_eids, Tensor separate_coo_row_indices, "
        "Tensor separate_coo_col_indices, Tensor left_side_data, Tensor right_node_vectors, Tensor gradout, "
        "Tensor(a!) grad_left_side_data, Tensor(b!) grad_right_node_vectors) -> ()",
        backward_inner_product_right_node_separatecoo);
  m.def("hgt_full_graph_edge_softmax_ops_separate_coo(Tensor row_indices, Tensor col_indices, Tensor eids, Tensor rel_ptrs, "
        "Tensor unnormalized_attn_score, Tensor mu, Tensor(a!) edgesoftmax_sum_per_node, "
        "Tensor(b!) mu_softmax_applied_unnormalized_attn_score, Tensor(c!) normalized_attn_score) -> ()",
        hgt_edge_softmax);
  m.def("backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(Tensor row_indices, Tensor col_indices, "
        "Tensor eids, Tensor rel_ptrs, Tensor unnormalized_attn_score, Tensor normalized_attn_score, "
        "Tensor grad_normalized_attn_score, Tensor mu, Tensor(a!) grad_unnormalized_attn_score, Tensor(b!) grad_mu, "
        "Tensor(c!) sum_incoming_edges_product_softmax_score) -> ()",
        backward_hgt_edge_softmax);
  m.def("hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(Tensor separate_coo_relptrs, "
        "Tensor separate_coo_eids, Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Tensor inputs, "
        "Tensor weights, Tensor edge_norm, Tensor(a!) new_h) -> ()",
        hgt_message_mean_aggregation);
  m.def("backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(Tensor separate_coo_relptrs, "
        "Tensor separate_coo_eids, Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Tensor inputs, "
        "Tensor weights_transposed, Tensor edge_norm, Tensor new_h, Tensor(a!) grad_input, Tensor(b!) grad_weights, "
        "Tensor(c!) grad_edge_norm, Tensor gradout) -> ()",
        backward_hgt_message_mean_aggregation);
  m.def("hgt_full_graph_hetero_attention_ops_coo(Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, "
        "Tensor separate_coo_eids, Tensor separate_coo_relptrs, Tensor applied_klinear_node_features, "
        "Tensor applied_qlinear_node_features, Tensor attn_score_weight, Tensor(a!) attn_score_inner_product, "
        "Tensor(b!) unnormalized_attn_score) -> ()",
        hgt_hetero_attention);
  m.def("backward_hgt_full_graph_hetero_attention_ops_coo(Tensor incsr_row_ptrs, Tensor incsr_col_indices, Tensor incsr_eids, "
        "Tensor incsr_reltypes, Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Tensor separate_coo_eids, "
        "Tensor separate_coo_relptrs, Tensor(a!) grad_attn_weight, Tensor attn_score_weight_transposed, "
        "Tensor applied_klinear_node_features, Tensor applied_qlinear_node_features, Tensor attn_score_inner_product, "
        "Tensor grad_unnorm_attn_score, Tensor(b!) grad_k, Tensor(c!) grad_q) -> ()",
        backward_hgt_hetero_attention);
}
