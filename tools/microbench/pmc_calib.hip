// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes this engine uses (the guide calibrates
// only 16-byte-per-lane streams: FETCH_SIZE reads half the bytes there).  Every kernel moves a KNOWN number of bytes of
// a 2 GiB buffer (8x the Infinity Cache) exactly once:
//   read16 / read8 : streaming reads, 16 / 8 bytes per lane
//   rows8          : 1200-byte rows (one element's grad N) in a scattered order, 8 bytes per lane, 10-lane segments of
//                    80 bytes (the fused assembly's h_j loads)
//   write16 / write8 : streaming stores
// Build + run under the profiler (separate passes):
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/pmc_calib.hip -o tools/microbench/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE -d out -o run --output-format csv -- tools/microbench/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void read16(const double2* __restrict__ a, size_t n, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = a[i];
    s += v.x + v.y;
  }
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void read8(const double* __restrict__ a, size_t n, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 123.456) out[0] = s;
}
// rows of 150 doubles; a wave reads 6 rows per step (lane = 10 k + j reads row[k][q*30 + d*10 + j] for the 15 (q,d))
__global__ __launch_bounds__(64) void rows8(const double* __restrict__ a, size_t n_rows, double* __restrict__ out) {
  const int lane = threadIdx.x, k = lane / 10, j = lane - 10 * k;
  double s = 0.0;
  for (size_t r0 = (size_t)blockIdx.x * 6; r0 + 6 <= n_rows; r0 += (size_t)gridDim.x * 6) {
    if (k < 6) {
      const size_t r = ((r0 + k) * 7919) % n_rows;  // scattered but a bijection (7919 prime, n_rows not a multiple)
      const double* p = a + r * 150 + j;
#pragma unroll
      for (int t = 0; t < 15; t++) s += p[t * 10];
    }
  }
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void write16(double2* __restrict__ a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a[i] = make_double2(1.0, 2.0);
}
__global__ __launch_bounds__(256) void write8(double* __restrict__ a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = 3.0;
}

int main() {
  const size_t bytes = 2ull << 30;
  double *a, *out;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
  hipMemset(a, 0, bytes);
  const size_t n_rows = bytes / 1200 - 1;  // 1 789 568 rows; 7919 does not divide it
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto timed = [&](const char* name, double gb, auto launch) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-8s %.3f GB in %.3f ms = %.2f TB/s\n", name, gb, ms, gb / ms);
  };
  for (int rep = 0; rep < 2; rep++) {
    timed("read16", bytes / 1e9, [&] { hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const double2*)a, bytes / 16, out); });
    timed("read8", bytes / 1e9, [&] { hipLaunchKernelGGL(read8, dim3(8192), dim3(256), 0, 0, a, bytes / 8, out); });
    timed("rows8", (n_rows / 6) * 6 * 1200 / 1e9, [&] { hipLaunchKernelGGL(rows8, dim3(32768), dim3(64), 0, 0, a, n_rows, out); });
    timed("write16", bytes / 1e9, [&] { hipLaunchKernelGGL(write16, dim3(8192), dim3(256), 0, 0, (double2*)a, bytes / 16); });
    timed("write8", bytes / 1e9, [&] { hipLaunchKernelGGL(write8, dim3(8192), dim3(256), 0, 0, a, bytes / 8); });
  }
  hipDeviceSynchronize();
  return 0;
}
