// Achievable HBM read / copy bandwidth on the box the tests run on (context for the roofline fractions in DESIGN.md):
// streaming sum of a large buffer with 16-byte loads per lane, and a copy, both over several sizes.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/hbm_stream.hip -o tools/microbench/hbm_stream && ./hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void read_kernel(const double2* __restrict__ a, size_t n, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = a[i];
    s += v.x + v.y;
  }
  if (s == 123.456) out[0] = s;  // never true: keeps the loads alive
}
__global__ __launch_bounds__(256) void copy_kernel(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// 8-byte loads per lane, 32-lane segments of 256 B from rows scattered with a stride (the SpMV's access shape)
__global__ __launch_bounds__(256) void read8_kernel(const double* __restrict__ a, size_t n, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 123.456) out[0] = s;
}

int main() {
  const size_t bytes_list[] = {64ull << 20, 1ull << 30, 4ull << 30};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double* out;
  hipMalloc(&out, 8);
  for (size_t bytes : bytes_list) {
    double2 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    const size_t n = bytes / sizeof(double2);
    for (int grid : {2048, 8192, 65536}) {
      float ms_r = 0, ms_c = 0, ms_8 = 0;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int k = 0; k < 5; k++) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_r, e0, e1);
        hipEventRecord(e0);
        for (int k = 0; k < 5; k++) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_c, e0, e1);
        hipEventRecord(e0);
        for (int k = 0; k < 5; k++) hipLaunchKernelGGL(read8_kernel, dim3(grid), dim3(256), 0, 0, (const double*)a, 2 * n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_8, e0, e1);
      }
      std::printf("bytes %6.2f GiB grid %6d: read16 %.2f TB/s  read8 %.2f TB/s  copy %.2f TB/s (read+write bytes)\n",
                  bytes / 1073741824.0, grid, 5.0 * bytes / (ms_r * 1e-3) / 1e12, 5.0 * bytes / (ms_8 * 1e-3) / 1e12,
                  5.0 * 2 * bytes / (ms_c * 1e-3) / 1e12);
    }
    hipFree(a);
    hipFree(b);
  }
  return 0;
}
