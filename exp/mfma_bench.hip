// Scratch micro-benchmark (not part of the product): times the MFMA segment GEMM forward on a
// mag-sized synthetic gather pattern, with optional ablations selected at compile time.
#include <vector>
#include <random>
#include <algorithm>
#include <cstdio>
#include "../het_amd/csrc/capi.hip"
#include "../het_amd/csrc/seg_gemm_mfma.hip"

int main(int argc, char** argv) {
  const int64_t E = 21111007, N = 1939743;
  const int K = 64, X = 64, R = 4;
  std::vector<int64_t> gather(E), scatter(E), relp = {0, 1043998, 1043998 + 5416271, 1043998 + 5416271 + 7505078, E};
  std::mt19937_64 rng(1);
  for (int r = 0; r < R; ++r) {
    for (int64_t i = relp[r]; i < relp[r + 1]; ++i) gather[i] = rng() % N;
    if (argc < 2) std::sort(gather.begin() + relp[r], gather.begin() + relp[r + 1]);
  }
  for (int64_t i = 0; i < E; ++i) scatter[i] = i;
  int64_t *dg, *ds, *dr; float *dx, *dw, *dc;
  hipMalloc(&dg, E * 8); hipMalloc(&ds, E * 8); hipMalloc(&dr, (R + 1) * 8);
  hipMalloc(&dx, N * K * 4); hipMalloc(&dw, R * K * X * 4); hipMalloc(&dc, E * X * 4);
  hipMemcpy(dg, gather.data(), E * 8, hipMemcpyHostToDevice);
  hipMemcpy(ds, scatter.data(), E * 8, hipMemcpyHostToDevice);
  hipMemcpy(dr, relp.data(), (R + 1) * 8, hipMemcpyHostToDevice);
  std::vector<float> hx(N * K), hw(R * K * X);
  std::uniform_real_distribution<float> u(-1, 1);
  for (auto& v : hx) v = u(rng);
  for (auto& v : hw) v = u(rng);
  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  MfmaGemmArgs a;
  a.A = dx; a.a_ld = K; a.gather = dg; a.B = dw; a.b_rel_stride = K * X; a.b_headcat = 1; a.headcat_d = 16;
  a.C = dc; a.c_ld = X; a.scatter = ds; a.seg_ptrs = dr; a.num_segs = R; a.num_rows = E; a.K = K; a.X = X;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) launch_seg_gemm_mfma(a, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int iters = 10;
  for (int it = 0; it < iters; ++it) launch_seg_gemm_mfma(a, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.3f ms per launch  (%.1f TF, %.1f GB/s written)\n", argc < 2 ? "sorted gather" : "random gather", ms / iters,
         2.0 * E * K * X / (ms / iters * 1e-3) / 1e12, (double)E * X * 4 / (ms / iters * 1e-3) / 1e9);
  return 0;
}
