// Scratch: what do 16-byte records cost when one side of a permutation is random?  (the per-edge term
// of grad_er is written in (relation, source) order and summed in (relation, destination) order.)
//   gather : out[i] = in[p(i)]   sequential 16-B stores, random 16-B loads (one 128-B line each?)
//   scatter: out[p(i)] = in[i]   sequential loads, random stores
// p permutes inside windows of W records (W = 2^k): W = all -> fully random, small W -> the random side stays inside a window
// that one workgroup (or a few neighbours) completes.  Load flavours: plain, nt, sc1, sc0 sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t wmask) {
  const uint32_t base = i & ~wmask, j = ((i & wmask) * 2654435761u + 12345u) & wmask;  // odd multiplier: a bijection mod 2^k
  return base | j;
}

template <int FLAVOUR>
__device__ __forceinline__ float4 ld16(const float4* p) {
  float4 v;
  if (FLAVOUR == 0) v = *p;
  if (FLAVOUR == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (FLAVOUR == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (FLAVOUR == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int FLAVOUR>
__global__ void k_gather(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n, uint32_t wmask) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t j = perm(i, wmask);
    if (j >= n) j = i;
    out[i] = ld16<FLAVOUR>(in + j);
  }
}
// 4 independent gathers in flight per thread (the inline-asm flavours above wait after every load)
__global__ void k_gather4(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n, uint32_t wmask) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      uint32_t ii = i + u * stride, j = perm(ii < n ? ii : i, wmask);
      if (j >= n) j = i;
      v[u] = in[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < n) out[i + u * stride] = v[u];
  }
}
__global__ void k_scatter(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n, uint32_t wmask) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t j = perm(i, wmask);
    if (j >= n) j = i;
    out[j] = in[i];
  }
}
// a workgroup owns a contiguous run of records (instead of the grid-stride interleave): the window is written by ONE workgroup
__global__ void k_scatter_wg(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n, uint32_t wmask, uint32_t per_wg) {
  const uint32_t b = blockIdx.x * per_wg, e = b + per_wg < n ? b + per_wg : n;
  for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
    uint32_t j = perm(i, wmask);
    if (j >= n) j = i;
    out[j] = in[i];
  }
}
__global__ void k_gather_wg(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n, uint32_t wmask, uint32_t per_wg) {
  const uint32_t b = blockIdx.x * per_wg, e = b + per_wg < n ? b + per_wg : n;
  for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
    uint32_t j = perm(i, wmask);
    if (j >= n) j = i;
    out[i] = in[j];
  }
}

int main() {
  const uint32_t n = 21111007;
  float4 *a, *b;
  hipMalloc(&a, (size_t)n * 16); hipMalloc(&b, (size_t)n * 16);
  hipMemset(a, 1, (size_t)n * 16); hipMemset(b, 0, (size_t)n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, uint32_t w, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      for (int it = 0; it < 5; ++it) launch();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      if (ms < best) best = ms;
    }
    printf("%-22s window %9u records: %.3f ms  (%.2f TB/s of useful bytes)\n", name, w, best, 2.0 * n * 16 / (best * 1e-3) / 1e12);
  };
  const int grid = 16384;
  for (uint32_t w : {1u << 25, 1u << 20, 1u << 16, 1u << 13, 1u << 10, 1u << 7, 1u << 3}) {
    const uint32_t m = w - 1;
    time("gather plain", w, [&] { hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, a, b, n, m); });
    time("gather x4 in flight", w, [&] { hipLaunchKernelGGL(k_gather4, dim3(grid / 4), dim3(256), 0, 0, a, b, n, m); });
    time("scatter", w, [&] { hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(256), 0, 0, a, b, n, m); });
    const uint32_t per_wg = w < 4096 ? 4096 : (w > 65536 ? 65536 : w);
    time("scatter, run per wg", w, [&] { hipLaunchKernelGGL(k_scatter_wg, dim3((n + per_wg - 1) / per_wg), dim3(256), 0, 0, a, b, n, m, per_wg); });
    time("gather, run per wg", w, [&] { hipLaunchKernelGGL(k_gather_wg, dim3((n + per_wg - 1) / per_wg), dim3(256), 0, 0, a, b, n, m, per_wg); });
  }
  const uint32_t m = (1u << 25) - 1;
  time("gather nt", 1u << 25, [&] { hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, a, b, n, m); });
  time("gather sc1", 1u << 25, [&] { hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), 0, 0, a, b, n, m); });
  time("gather sc0 sc1", 1u << 25, [&] { hipLaunchKernelGGL(k_gather<3>, dim3(grid), dim3(256), 0, 0, a, b, n, m); });
  time("gather plain (again)", 1u << 25, [&] { hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, a, b, n, m); });
  return 0;
}
