// The C ABI without PyTorch: raw hipMalloc buffers, the segment GEMM (a1) and its backward (a2) through
// include/het_amd.h, checked against plain host loops.  Built and run by tests/test_gpu_capi.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "het_amd.h"

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); return 2; } } while (0)
#define HK(x) do { int rc__ = (x); if (rc__ != HET_OK) { printf("het error %d: %s\n", rc__, het_last_error()); return 3; } } while (0)

template <class T>
T* to_dev(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(T) * (h.empty() ? 1 : h.size())) != hipSuccess) return nullptr;
  if (!h.empty()) (void)hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice);
  return d;
}

int main() {
  printf("%s\n", het_build_info());
  const int64_t R = 3, H = 4, K = 64, D = 16, N = 500, E = 7000, X = H * D;
  srand(7);
  std::vector<int64_t> rel_ptrs = {0, 2500, 2500 + 3100, E}, row(E), eids(E);
  for (int64_t i = 0; i < E; ++i) { row[i] = rand() % N; eids[i] = i; }
  for (int64_t i = E - 1; i > 0; --i) { int64_t j = rand() % (i + 1); std::swap(eids[i], eids[j]); }  // a real permutation
  std::vector<float> W(R * H * K * D), x(N * K), go(E * X);
  for (auto& v : W) v = (rand() % 2001 - 1000) / 4000.f;
  for (auto& v : x) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : go) v = (rand() % 2001 - 1000) / 1000.f;
  // host reference: ret[eids[i], h, :] = x[row[i], :] . W[r, h];  grad_x[row[i]] += go[eids[i], h, :] . W[r, h]^T
  std::vector<double> ret_ref(E * X, 0.0), gx_ref(N * K, 0.0), gw_ref(R * H * K * D, 0.0);
  for (int64_t r = 0; r < R; ++r)
    for (int64_t i = rel_ptrs[r]; i < rel_ptrs[r + 1]; ++i)
      for (int64_t h = 0; h < H; ++h)
        for (int64_t k = 0; k < K; ++k)
          for (int64_t d = 0; d < D; ++d) {
            const double w = W[((r * H + h) * K + k) * D + d], xv = x[row[i] * K + k], g = go[eids[i] * X + h * D + d];
            ret_ref[eids[i] * X + h * D + d] += xv * w;
            gx_ref[row[i] * K + k] += g * w;
            gw_ref[((r * H + h) * K + k) * D + d] += xv * g;
          }
  std::vector<float> Wt(R * H * D * K);
  for (int64_t rh = 0; rh < R * H; ++rh)
    for (int64_t k = 0; k < K; ++k)
      for (int64_t d = 0; d < D; ++d) Wt[(rh * D + d) * K + k] = W[(rh * K + k) * D + d];
  int64_t *d_rp = to_dev(rel_ptrs), *d_row = to_dev(row), *d_eids = to_dev(eids);
  float *d_W = to_dev(W), *d_Wt = to_dev(Wt), *d_x = to_dev(x), *d_go = to_dev(go), *d_ret, *d_gx, *d_gw, *d_ws;
  CK(hipMalloc((void**)&d_ret, sizeof(float) * E * X));
  CK(hipMalloc((void**)&d_gx, sizeof(float) * N * K));
  CK(hipMalloc((void**)&d_gw, sizeof(float) * R * H * K * D));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  HK(het_rgnn_relational_matmul(HET_KIND_DISABLED, d_rp, R, d_row, d_eids, E, d_W, d_x, d_ret, H, K, D, 1, nullptr, nullptr, 0, s));
  // backward with the optional (relation, row) grouping: payload0 = the scatter list
  het_grouping* g = nullptr;
  HK(het_grouping_create(d_rp, R, d_row, E, N, d_eids, nullptr, s, &g));
  const int64_t S = het_grouping_num_segments(g);
  CK(hipMalloc((void**)&d_ws, sizeof(float) * S * X));
  HK(het_backward_rgnn_relational_matmul(HET_KIND_DISABLED, d_rp, R, d_row, d_eids, E, N, d_Wt, d_x, d_go, d_gx, d_gw, H, K, D, 1,
                                         /*accumulate=*/0, g, d_ws, (int64_t)sizeof(float) * S * X, s));
  CK(hipStreamSynchronize(s));
  std::vector<float> ret(E * X), gx(N * K), gw(R * H * K * D);
  CK(hipMemcpy(ret.data(), d_ret, sizeof(float) * ret.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(gx.data(), d_gx, sizeof(float) * gx.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(gw.data(), d_gw, sizeof(float) * gw.size(), hipMemcpyDeviceToHost));
  auto worst = [](const std::vector<float>& a, const std::vector<double>& b) {
    double m = 0, sc = 1;
    for (size_t i = 0; i < a.size(); ++i) { m = std::fmax(m, std::fabs(a[i] - b[i])); sc = std::fmax(sc, std::fabs(b[i])); }
    return m / sc;
  };
  const double e1 = worst(ret, ret_ref), e2 = worst(gx, gx_ref), e3 = worst(gw, gw_ref);
  printf("segments %lld  rel err: ret %.2e grad_x %.2e grad_w %.2e\n", (long long)S, e1, e2, e3);
  // argument validation comes back as an error code + message, not a fault
  const int rc = het_rgnn_relational_matmul(7, d_rp, R, d_row, d_eids, E, d_W, d_x, d_ret, H, K, D, 1, nullptr, nullptr, 0, s);
  printf("bad kind -> rc %d (%s)\n", rc, het_last_error());
  het_grouping_destroy(g);
  const bool ok = e1 < 2e-4 && e2 < 2e-4 && e3 < 2e-4 && rc == HET_ERR_UNSUPPORTED;
  printf(ok ? "CAPI STANDALONE OK\n" : "CAPI STANDALONE FAILED\n");
  return ok ? 0 : 1;
}
