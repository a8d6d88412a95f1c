// Cost of a hand-rolled barrier among a FEW co-resident workgroups, with a small vector exchanged through global
// memory every step: the shape of a fused coarse-level Chebyshev solve of the p-multigrid cycle,
// to be compared with the ~3.8 us spacing of dependent kernel launches inside a hipGraph.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/grid_barrier.hip -o tools/microbench/grid_barrier && ./grid_barrier
// Grid = n_active * stride workgroups of which only blockIdx % stride == 0 take part (stride 8 puts them all on one XCD
// under the round-robin workgroup -> XCD dispatch).  Every spin is bounded: a barrier that does not complete sets an
// error flag and all waves leave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kSpinLimit = 1 << 22;

template <bool FENCES>
__device__ __forceinline__ bool barrier(unsigned* counter, unsigned target, int* err) {
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = 1;
    // release at agent scope: this workgroup's earlier stores are written back before the arrival is counted
    if (FENCES) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {  // the data went out as agent-scope atomic stores: wait for their acknowledgement, then count the arrival
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int spin = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spin > kSpinLimit || *(volatile int*)err) {
        *err = 1;
        ok = 0;
        break;
      }
    }
    if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // one invalidate after the wait, not one per poll
  }
  __syncthreads();
  return ok != 0;
}

template <bool FENCES>
__global__ __launch_bounds__(1024) void exchange_kernel(unsigned* counter, int n_active, int stride, int steps,
                                                        float* buf0, float* buf1, int n_vec, int* err, float* out) {
  if (blockIdx.x % stride) return;
  const int wg = blockIdx.x / stride;
  if (wg >= n_active) return;
  const int per = (n_vec + n_active - 1) / n_active;
  const int lo = wg * per, hi = min(n_vec, lo + per);
  float acc = 0.f;
  for (int s = 0; s < steps; s++) {
    float* w = (s & 1) ? buf1 : buf0;
    const float* r = (s & 1) ? buf1 : buf0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
      const float v = (float)(s + 1) + acc * 0.f;
      if (FENCES) w[i] = v;
      else __hip_atomic_store(w + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!FENCES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every lane's stores acknowledged before __syncthreads
    if (!barrier<FENCES>(counter, (unsigned)(s + 1) * n_active, err)) return;
    // gather 16 entries owned by other workgroups: must all carry this step's value
    for (int k = 0; k < 16; k++) {
      const int j = (int)(((unsigned)(threadIdx.x * 16 + k) * 2654435761u + (unsigned)wg * 40503u) % (unsigned)n_vec);
      const float v = FENCES ? r[j] : __hip_atomic_load(r + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (float)(s + 1)) *err = 2;
      acc += v;
    }
  }
  if (acc == 123.f) out[0] = acc;
}

int main() {
  unsigned* counter;
  int* err;
  float *b0, *b1, *out;
  const int n_vec = 6591;  // config B's coarse level: 2197 nodes x 3
  (void)hipMalloc(&counter, 4);
  (void)hipMalloc(&err, 4);
  (void)hipMalloc(&b0, n_vec * 4);
  (void)hipMalloc(&b1, n_vec * 4);
  (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int steps = 2000;
  for (int fences : {1, 0})
  for (int stride : {1, 8})
    for (int n_active : {4, 16, 64}) {
      if (n_active * stride > 1024) continue;
      float best = 1e30f;
      int h_err = 0;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipMemset(counter, 0, 4);
        (void)hipMemset(err, 0, 4);
        (void)hipEventRecord(e0);
        if (fences)
          hipLaunchKernelGGL(exchange_kernel<true>, dim3(n_active * stride), dim3(1024), 0, 0, counter, n_active, stride,
                             steps, b0, b1, n_vec, err, out);
        else
          hipLaunchKernelGGL(exchange_kernel<false>, dim3(n_active * stride), dim3(1024), 0, 0, counter, n_active, stride,
                             steps, b0, b1, n_vec, err, out);
        (void)hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) {
          printf("launch failed\n");
          return 1;
        }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        (void)hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost);
        if (h_err) break;
      }
      printf("%s  stride %d  workgroups %3d : %.3f us per exchange step  err=%d\n", fences ? "bulk fences  " : "atomic access", stride, n_active, 1e3f * best / steps, h_err);
      fflush(stdout);
      if (h_err == 1) return 2;  // a barrier timed out: do not keep launching
    }
  return 0;
}
