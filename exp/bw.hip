// Scratch: calibrate streaming read / write / copy bandwidth on this device.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read(const float4* __restrict__ a, float4* out, size_t n) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x == 12345.f) out[0] = acc;
}
__global__ void k_write(float4* __restrict__ a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
  const size_t bytes = 5400000000ull / 16 * 16, n = bytes / 16;
  float4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {2048, 8192, 65536}) {
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < 5; ++it) {
          if (mode == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, n);
          if (mode == 1) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n);
          if (mode == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
          if (mode == 3) hipMemsetAsync(b, 0, bytes, 0);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const char* nm[] = {"read", "write", "copy", "memset"};
      printf("grid %6d %-6s %.3f ms  %.2f TB/s (moved bytes)\n", grid, nm[mode], ms, (mode == 2 ? 2 : 1) * bytes / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
