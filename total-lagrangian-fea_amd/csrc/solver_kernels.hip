// solver_kernels.hip -- vector kernels of the ALM/Newton step and the device linear solver.
//
// The reference solves H dv = -g with cuDSS (SyncedNewton.cu:995-1029,1103-1114).  Here H (SPD:
// M/h + h K + C_vis + h^2 rho J^T J) is solved by a 3x3-block-Jacobi preconditioned CG that streams
// the CSR values once per iteration.  All reductions use a fixed number of partial sums that every
// consumer re-adds in the same order, so results are bitwise reproducible run to run (the
// reference's atomics are not).
#include "tlfea_internal.h"

#include <algorithm>
#include <cstdlib>

namespace tlfea {

// ---- deterministic reductions -------------------------------------------------------------------
// wave64 butterfly + one LDS hop: 1-2 barriers per workgroup sum instead of a 9-barrier tree (these
// kernels are latency-, not bandwidth-bound on small meshes).  Orders are fixed -> bitwise reproducible.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sum over the workgroup (blockDim.x/64 waves <= 16); sh needs 16 doubles
__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  double r = 0.0;
  for (int k = 0; k < nw; k++) r += sh[k];
  __syncthreads();
  return r;
}

__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) {
    sh[wv] = a;
    sh[16 + wv] = b;
  }
  __syncthreads();
  double ra = 0.0, rb = 0.0;
  for (int k = 0; k < nw; k++) {
    ra += sh[k];
    rb += sh[16 + k];
  }
  __syncthreads();
  a = ra;
  b = rb;
}

// totals of two kNPart-slot partial arrays, identical in every workgroup of every kernel
__device__ __forceinline__ void sum_slots2(const double* __restrict__ pa, const double* __restrict__ pb, double& a,
                                           double& b, double* sh) {
  double va = 0.0, vb = 0.0;
  for (int k = threadIdx.x; k < kNPart; k += blockDim.x) {
    va += pa[k];
    vb += pb[k];
  }
  block_sum2(va, vb, sh);
  a = va;
  b = vb;
}

__global__ __launch_bounds__(256) void norm2_part_kernel(const double* __restrict__ a, const double* __restrict__ w,
                                                        int n, double* __restrict__ part) {
  __shared__ double sh[32];
  double s = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += (w ? w[i] : 1.0) * a[i] * a[i];
  const double r = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double sh[32];
  double v = 0.0;
  for (int k = threadIdx.x; k < kNPart; k += 256) v += part[k];
  const double r = block_sum(v, sh);
  if (threadIdx.x == 0) out[0] = r;
}

void launch_norm2(hipStream_t s, const double* a, const double* w, int n, double* part, double* out) {
  hipLaunchKernelGGL(norm2_part_kernel, dim3(kNPart), dim3(256), 0, s, a, w, n, part);
  hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(256), 0, s, part, out);
}

void launch_sum_parts(hipStream_t s, const double* part, double* out) {
  hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(256), 0, s, part, out);
}

// ---- block-Jacobi preconditioner ---------------------------------------------------------------
__global__ void extract_dinv_kernel(int N, Incidence inc, const double* __restrict__ Hval,
                                    double* __restrict__ Dinv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int off0 = inc.off[i], deg = inc.off[i + 1] - off0, dp = inc.diagpos[i];
  const double* Hi = Hval + (size_t)9 * off0;
  double D[3][3];
#pragma unroll
  for (int d = 0; d < 3; d++)
#pragma unroll
    for (int e = 0; e < 3; e++) D[d][e] = Hi[d * 3 * deg + 3 * dp + e];
  const double det = D[0][0] * (D[1][1] * D[2][2] - D[1][2] * D[2][1]) -
                     D[0][1] * (D[1][0] * D[2][2] - D[1][2] * D[2][0]) +
                     D[0][2] * (D[1][0] * D[2][1] - D[1][1] * D[2][0]);
  const double id = 1.0 / det;
  double* o = Dinv + (size_t)9 * i;
  o[0] = (D[1][1] * D[2][2] - D[1][2] * D[2][1]) * id;
  o[1] = (D[0][2] * D[2][1] - D[0][1] * D[2][2]) * id;
  o[2] = (D[0][1] * D[1][2] - D[0][2] * D[1][1]) * id;
  o[3] = (D[1][2] * D[2][0] - D[1][0] * D[2][2]) * id;
  o[4] = (D[0][0] * D[2][2] - D[0][2] * D[2][0]) * id;
  o[5] = (D[0][2] * D[1][0] - D[0][0] * D[1][2]) * id;
  o[6] = (D[1][0] * D[2][1] - D[1][1] * D[2][0]) * id;
  o[7] = (D[0][1] * D[2][0] - D[0][0] * D[2][1]) * id;
  o[8] = (D[0][0] * D[1][1] - D[0][1] * D[1][0]) * id;
}

void launch_extract_dinv(hipStream_t s, int N, const Incidence& inc, const double* Hval, double* Dinv) {
  hipLaunchKernelGGL(extract_dinv_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, inc, Hval, Dinv);
}

// ---- PCG --------------------------------------------------------------------------------------
// `w` (optional) weights each DOF in the dot products: 1/multiplicity of partition-interface DOFs,
// so that the sum over ranks of the local dots is the global dot.
// Every reduction lands in a kNPart-slot partial array (workgroups that do not exist leave their slot 0);
// every consumer re-adds the slots in the same order.
__global__ __launch_bounds__(256) void pcg_init_kernel(int N, const double* __restrict__ b,
                                                      const double* __restrict__ Dinv, const double* __restrict__ w,
                                                      double* __restrict__ x, double* __restrict__ r,
                                                      double* __restrict__ z, double* __restrict__ rz_part,
                                                      double* __restrict__ bb_part) {
  __shared__ double sh[32];
  double rz = 0.0, bb = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const double r0 = b[3 * i], r1 = b[3 * i + 1], r2 = b[3 * i + 2];
    const double* D = Dinv + (size_t)9 * i;
    const double z0 = D[0] * r0 + D[1] * r1 + D[2] * r2;
    const double z1 = D[3] * r0 + D[4] * r1 + D[5] * r2;
    const double z2 = D[6] * r0 + D[7] * r1 + D[8] * r2;
    x[3 * i] = x[3 * i + 1] = x[3 * i + 2] = 0.0;
    r[3 * i] = r0; r[3 * i + 1] = r1; r[3 * i + 2] = r2;
    z[3 * i] = z0; z[3 * i + 1] = z1; z[3 * i + 2] = z2;
    const double w0 = w ? w[3 * i] : 1.0, w1 = w ? w[3 * i + 1] : 1.0, w2 = w ? w[3 * i + 2] : 1.0;
    rz += w0 * r0 * z0 + w1 * r1 * z1 + w2 * r2 * z2;
    bb += w0 * r0 * r0 + w1 * r1 * r1 + w2 * r2 * r2;
  }
  block_sum2(rz, bb, sh);
  if (threadIdx.x == 0) {
    rz_part[blockIdx.x] = rz;
    bb_part[blockIdx.x] = bb;
  }
}

void launch_pcg_init(hipStream_t s, int N, const double* b, const double* Dinv, const double* w, double* x,
                     double* r, double* z, double* rz_part, double* bb_part) {
  hipLaunchKernelGGL(pcg_init_kernel, dim3(kNPart), dim3(256), 0, s, N, b, Dinv, w, x, r, z, rz_part, bb_part);
}

// q = H p_new with p_new = z + beta p_old formed on the fly (beta = rz_new/rz_old from the partial slots;
// first iteration: p_new = z), so CG needs no separate direction kernel: 2 launches per iteration.
// Two node rows (6 CSR rows) per wavefront at a time: a 32-lane half-wave walks the 3*deg (k,e) columns of
// one node row, so the three value rows and the column-node list are read coalesced; the value loads are
// issued ahead of the dependent cols -> z/p gather chain.  Workgroups of 1024 threads own contiguous row
// chunks (neighbouring rows share most column nodes -> the gathers hit L1/L2) and the grid never exceeds
// kNPart workgroups, so the p.q partials fit the common slot array.
template <bool NT>
__device__ __forceinline__ double ld_stream(const double* p) {
  // H is streamed once per CG iteration: a non-temporal load keeps it from evicting the gathered vectors
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// FUSED: p_new = z + beta p_old is formed on the fly (small meshes: one launch fewer per iteration).
// !FUSED: p was written by pcg_direction_kernel; only p is gathered (large meshes: half the gather traffic).
template <bool FUSED, bool NT, int LANES>
__global__ __launch_bounds__(1024) void spmv_dir_dot_kernel(int N, Incidence inc, const double* __restrict__ Hval,
                                                           const double* __restrict__ z,
                                                           const double* __restrict__ p_old, int first,
                                                           const double* __restrict__ rz_part_old,
                                                           const double* __restrict__ rz_part_new,
                                                           double* __restrict__ p_new, double* __restrict__ q,
                                                           double* __restrict__ pq_part) {
  __shared__ double sh[32];
  double beta = 0.0;
  if (FUSED && !first) {
    double rz_old, rz_new;
    sum_slots2(rz_part_old, rz_part_new, rz_old, rz_new, sh);
    beta = rz_new / rz_old;
  }
  // LANES (32 or 16) lanes walk one node row: 2 or 4 rows per wavefront at a time
  const int l32 = threadIdx.x & (LANES - 1), hw = threadIdx.x / LANES;
  constexpr int kGroups = 1024 / LANES;
  const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  double pq = 0.0;
  for (int i = r0 + hw; i < r1; i += kGroups) {
    const int off0 = inc.off[i], deg = inc.off[i + 1] - off0, row = 3 * deg;
    const double* Hi = Hval + (size_t)9 * off0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    // two column rounds in flight per pass (the loads of both rounds are issued before either gather returns)
    for (int t = l32; t < row; t += 2 * LANES) {
      const int t2 = t + LANES;
      const bool has2 = t2 < row;
      const double h0 = ld_stream<NT>(Hi + t), h1 = ld_stream<NT>(Hi + row + t), h2 = ld_stream<NT>(Hi + 2 * row + t);
      const int k = t / 3, e = t - 3 * k;
      const int ca = 3 * inc.cols[off0 + k] + e;
      double g0 = 0.0, g1 = 0.0, g2 = 0.0;
      int cb = ca;
      if (has2) {
        g0 = ld_stream<NT>(Hi + t2);
        g1 = ld_stream<NT>(Hi + row + t2);
        g2 = ld_stream<NT>(Hi + 2 * row + t2);
        const int kb = t2 / 3, eb = t2 - 3 * kb;
        cb = 3 * inc.cols[off0 + kb] + eb;
      }
      double pa, pb;
      if (FUSED) {
        pa = first ? z[ca] : (z[ca] + beta * p_old[ca]);
        pb = first ? z[cb] : (z[cb] + beta * p_old[cb]);
      } else {
        pa = p_old[ca];
        pb = p_old[cb];
      }
      s0 += h0 * pa + g0 * pb;
      s1 += h1 * pa + g1 * pb;
      s2 += h2 * pa + g2 * pb;
    }
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) {
      s0 += __shfl_xor(s0, o);
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    if (l32 < 3) {
      const int c = 3 * i + l32;
      double pv;
      if (FUSED) {
        pv = first ? z[c] : (z[c] + beta * p_old[c]);
        p_new[c] = pv;
      } else {
        pv = p_old[c];
      }
      const double sv = (l32 == 0) ? s0 : ((l32 == 1) ? s1 : s2);
      q[c] = sv;
      pq += pv * sv;  // sv is this rank's partial (H_r p)_c: sum_r p.(H_r p) = p.Hp, so no interface weight here
    }
  }
  const double r = block_sum(pq, sh);
  if (threadIdx.x == 0) pq_part[blockIdx.x] = r;
  // slots of workgroups that do not exist on THIS rank must read 0: after a cross-rank sum they hold other
  // ranks' partials of the previous iteration (ranks differ in grid size)
  if (blockIdx.x == 0)
    for (int k = gridDim.x + threadIdx.x; k < kNPart; k += blockDim.x) pq_part[k] = 0.0;
}

int spmv_grid(int N) { return std::max(1, std::min(kNPart, (N + 31) / 32)); }  // small meshes: one row per half-wave

void launch_spmv_dir_dot(hipStream_t s, int N, const Incidence& inc, const double* Hval, const double* z,
                         const double* p_old, int first, const double* rz_part_old, const double* rz_part_new,
                         double* p_new, double* q, double* pq_part, bool fused, bool nt) {
  const dim3 g(spmv_grid(N)), b(1024);
  static const int lanes = std::getenv("TLFEA_SPMV_LANES") ? std::atoi(std::getenv("TLFEA_SPMV_LANES")) : 32;
#define TLFEA_SPMV(F, T, L)                                                                                       \
  hipLaunchKernelGGL((spmv_dir_dot_kernel<F, T, L>), g, b, 0, s, N, inc, Hval, z, p_old, first, rz_part_old,          \
                     rz_part_new, p_new, q, pq_part)
  if (lanes == 16) {
    if (fused) TLFEA_SPMV(true, false, 16);
    else TLFEA_SPMV(false, false, 16);
  } else if (fused && nt) TLFEA_SPMV(true, true, 32);
  else if (fused) TLFEA_SPMV(true, false, 32);
  else if (nt) TLFEA_SPMV(false, true, 32);
  else TLFEA_SPMV(false, false, 32);
#undef TLFEA_SPMV
}

// p = z + beta p (beta from the partial slots; first iteration p = z) -- large meshes only
__global__ __launch_bounds__(256) void pcg_direction_kernel(int n, const double* __restrict__ z, int first,
                                                           const double* __restrict__ rz_part_old,
                                                           const double* __restrict__ rz_part_new,
                                                           double* __restrict__ p) {
  __shared__ double sh[32];
  double beta = 0.0;
  if (!first) {
    double rz_old, rz_new;
    sum_slots2(rz_part_old, rz_part_new, rz_old, rz_new, sh);
    beta = rz_new / rz_old;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = first ? z[i] : (z[i] + beta * p[i]);
}

void launch_pcg_direction(hipStream_t s, int n, const double* z, int first, const double* rz_part_old,
                          const double* rz_part_new, double* p) {
  hipLaunchKernelGGL(pcg_direction_kernel, dim3(std::min(2048, (n + 255) / 256)), dim3(256), 0, s, n, z, first,
                     rz_part_old, rz_part_new, p);
}

// x += alpha p ; r -= alpha q ; z = Dinv r ; partials of r.z and r.r
__global__ __launch_bounds__(256) void pcg_update_kernel(int N, const double* __restrict__ Dinv,
                                                        const double* __restrict__ w, const double* __restrict__ p,
                                                        const double* __restrict__ q,
                                                        const double* __restrict__ rz_part_old,
                                                        const double* __restrict__ pq_part, double* __restrict__ x,
                                                        double* __restrict__ r, double* __restrict__ z,
                                                        double* __restrict__ rz_part_new,
                                                        double* __restrict__ rr_part) {
  __shared__ double sh[32];
  double rz_old, pq;
  sum_slots2(rz_part_old, pq_part, rz_old, pq, sh);
  const double alpha = rz_old / pq;
  double rz = 0.0, rr = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    double rv[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      x[3 * i + d] += alpha * p[3 * i + d];
      rv[d] = r[3 * i + d] - alpha * q[3 * i + d];
      r[3 * i + d] = rv[d];
    }
    const double* D = Dinv + (size_t)9 * i;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const double zd = D[3 * d] * rv[0] + D[3 * d + 1] * rv[1] + D[3 * d + 2] * rv[2];
      z[3 * i + d] = zd;
      const double wd = w ? w[3 * i + d] : 1.0;
      rz += wd * rv[d] * zd;
      rr += wd * rv[d] * rv[d];
    }
  }
  block_sum2(rz, rr, sh);
  if (threadIdx.x == 0) {
    rz_part_new[blockIdx.x] = rz;
    rr_part[blockIdx.x] = rr;
  }
  if (blockIdx.x == 0)
    for (int k = gridDim.x + threadIdx.x; k < kNPart; k += blockDim.x) {
      rz_part_new[k] = 0.0;
      rr_part[k] = 0.0;
    }
}

void launch_pcg_update(hipStream_t s, int N, const double* Dinv, const double* w, const double* p, const double* q,
                       const double* rz_part_old, const double* pq_part, double* x, double* r, double* z,
                       double* rz_part_new, double* rr_part) {
  const int n_blocks = std::max(1, std::min(kNPart, (N + 255) / 256));
  hipLaunchKernelGGL(pcg_update_kernel, dim3(n_blocks), dim3(256), 0, s, N, Dinv, w, p, q, rz_part_old, pq_part, x, r,
                     z, rz_part_new, rr_part);
}

// ---- Chebyshev polynomial preconditioner ----------------------------------------------------------
// z = p_d(D^-1 H) D^-1 r : d steps of the Chebyshev iteration for H z = r on [lmax/kappa, lmax] of D^-1 H, started
// from z = 0.  A fixed polynomial, so it is a valid (SPD) CG preconditioner; unlike CG it needs NO dot products:
// every step is one SpMV fused with its vector updates -- on small meshes the two reductions per CG iteration are
// what the iteration costs, on large ones the fused step streams H once with no extra vector passes.
// step 0 (no SpMV):  d = D^-1 r / theta ; z = d ; res = r
__global__ __launch_bounds__(256) void cheb_init_kernel(int N, const double* __restrict__ Dinv,
                                                       const double* __restrict__ r, double inv_theta,
                                                       double* __restrict__ d, double* __restrict__ z,
                                                       double* __restrict__ res) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const double r0 = r[3 * i], r1 = r[3 * i + 1], r2 = r[3 * i + 2];
  const double* D = Dinv + (size_t)9 * i;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const double v = (D[3 * c] * r0 + D[3 * c + 1] * r1 + D[3 * c + 2] * r2) * inv_theta;
    d[3 * i + c] = v;
    z[3 * i + c] = v;
  }
  res[3 * i] = r0;
  res[3 * i + 1] = r1;
  res[3 * i + 2] = r2;
}

void launch_cheb_init(hipStream_t s, int N, const double* Dinv, const double* r, double inv_theta, double* d,
                      double* z, double* res) {
  hipLaunchKernelGGL(cheb_init_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, Dinv, r, inv_theta, d, z, res);
}

// step k >= 1:  res -= H d_old ; d_new = c1 d_old + c2 D^-1 res ; z += d_new          (one launch, no reduction)
// LAST adds the partial slots of r.z for the enclosing CG (w: 1/multiplicity weights, multi-GPU).
template <bool LAST>
__global__ __launch_bounds__(1024) void cheb_step_kernel(int N, Incidence inc, const double* __restrict__ Hval,
                                                        const double* __restrict__ Dinv,
                                                        const double* __restrict__ d_old, double c1, double c2,
                                                        double* __restrict__ d_new, double* __restrict__ z,
                                                        double* __restrict__ res, const double* __restrict__ r,
                                                        const double* __restrict__ w, double* __restrict__ rz_part) {
  __shared__ double sh[32];
  const int l32 = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  double rz = 0.0;
  for (int i = r0 + hw; i < r1; i += 32) {
    const int off0 = inc.off[i], deg = inc.off[i + 1] - off0, row = 3 * deg;
    const double* Hi = Hval + (size_t)9 * off0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int t = l32; t < row; t += 64) {
      const int t2 = t + 32;
      const bool has2 = t2 < row;
      const double h0 = Hi[t], h1 = Hi[row + t], h2 = Hi[2 * row + t];
      const int k = t / 3, e = t - 3 * k;
      const int ca = 3 * inc.cols[off0 + k] + e;
      double g0 = 0.0, g1 = 0.0, g2 = 0.0;
      int cb = ca;
      if (has2) {
        g0 = Hi[t2];
        g1 = Hi[row + t2];
        g2 = Hi[2 * row + t2];
        const int kb = t2 / 3, eb = t2 - 3 * kb;
        cb = 3 * inc.cols[off0 + kb] + eb;
      }
      const double pa = d_old[ca], pb = d_old[cb];
      s0 += h0 * pa + g0 * pb;
      s1 += h1 * pa + g1 * pb;
      s2 += h2 * pa + g2 * pb;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      s0 += __shfl_xor(s0, o);
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    // all lanes hold the row sums: lanes 0..2 finish component c
    if (l32 < 3) {
      const int c = l32;
      const double e0 = res[3 * i] - s0, e1 = res[3 * i + 1] - s1, e2 = res[3 * i + 2] - s2;
      const double* D = Dinv + (size_t)9 * i;
      const double zc = D[3 * c] * e0 + D[3 * c + 1] * e1 + D[3 * c + 2] * e2;
      const double dn = c1 * d_old[3 * i + c] + c2 * zc;
      const double zn = z[3 * i + c] + dn;
      d_new[3 * i + c] = dn;
      z[3 * i + c] = zn;
      if (LAST) rz += (w ? w[3 * i + c] : 1.0) * r[3 * i + c] * zn;
    }
    // res is read by lanes 0..2 of THIS half-wave only, so it can be overwritten once they are done
    if (!LAST) {
      const double ec = (l32 == 0) ? s0 : ((l32 == 1) ? s1 : s2);
      if (l32 < 3) res[3 * i + l32] = res[3 * i + l32] - ec;
    }
  }
  if (LAST) {
    const double t = block_sum(rz, sh);
    if (threadIdx.x == 0) rz_part[blockIdx.x] = t;
    if (blockIdx.x == 0)
      for (int k = gridDim.x + threadIdx.x; k < kNPart; k += blockDim.x) rz_part[k] = 0.0;
  }
}

void launch_cheb_step(hipStream_t s, int N, const Incidence& inc, const double* Hval, const double* Dinv,
                      const double* d_old, double c1, double c2, double* d_new, double* z, double* res,
                      const double* r, const double* w, double* rz_part, bool last) {
  const dim3 g(spmv_grid(N)), b(1024);
  if (last)
    hipLaunchKernelGGL((cheb_step_kernel<true>), g, b, 0, s, N, inc, Hval, Dinv, d_old, c1, c2, d_new, z, res, r, w,
                       rz_part);
  else
    hipLaunchKernelGGL((cheb_step_kernel<false>), g, b, 0, s, N, inc, Hval, Dinv, d_old, c1, c2, d_new, z, res, r, w,
                       rz_part);
}

// the same step with the SpMV result q = H d_old already summed over ranks (multi-GPU path)
template <bool LAST>
__global__ __launch_bounds__(256) void cheb_update_kernel(int N, const double* __restrict__ Dinv,
                                                         const double* __restrict__ q,
                                                         const double* __restrict__ d_old, double c1, double c2,
                                                         double* __restrict__ d_new, double* __restrict__ z,
                                                         double* __restrict__ res, const double* __restrict__ r,
                                                         const double* __restrict__ w, double* __restrict__ rz_part) {
  __shared__ double sh[32];
  double rz = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const double e0 = res[3 * i] - q[3 * i], e1 = res[3 * i + 1] - q[3 * i + 1], e2 = res[3 * i + 2] - q[3 * i + 2];
    res[3 * i] = e0;
    res[3 * i + 1] = e1;
    res[3 * i + 2] = e2;
    const double* D = Dinv + (size_t)9 * i;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double zc = D[3 * c] * e0 + D[3 * c + 1] * e1 + D[3 * c + 2] * e2;
      const double dn = c1 * d_old[3 * i + c] + c2 * zc;
      const double zn = z[3 * i + c] + dn;
      d_new[3 * i + c] = dn;
      z[3 * i + c] = zn;
      if (LAST) rz += (w ? w[3 * i + c] : 1.0) * r[3 * i + c] * zn;
    }
  }
  if (LAST) {
    const double t = block_sum(rz, sh);
    if (threadIdx.x == 0) rz_part[blockIdx.x] = t;
    if (blockIdx.x == 0)
      for (int k = gridDim.x + threadIdx.x; k < kNPart; k += blockDim.x) rz_part[k] = 0.0;
  }
}

void launch_cheb_update(hipStream_t s, int N, const double* Dinv, const double* q, const double* d_old, double c1,
                        double c2, double* d_new, double* z, double* res, const double* r, const double* w,
                        double* rz_part, bool last) {
  const dim3 g(std::max(1, std::min(kNPart, (N + 255) / 256))), b(256);
  if (last)
    hipLaunchKernelGGL((cheb_update_kernel<true>), g, b, 0, s, N, Dinv, q, d_old, c1, c2, d_new, z, res, r, w, rz_part);
  else
    hipLaunchKernelGGL((cheb_update_kernel<false>), g, b, 0, s, N, Dinv, q, d_old, c1, c2, d_new, z, res, r, w, rz_part);
}

// x += alpha p ; r -= alpha q ; partial r.r  (z comes from the polynomial preconditioner afterwards)
__global__ __launch_bounds__(256) void pcg_update_noz_kernel(int N, const double* __restrict__ w,
                                                            const double* __restrict__ p, const double* __restrict__ q,
                                                            const double* __restrict__ rz_part_old,
                                                            const double* __restrict__ pq_part, double* __restrict__ x,
                                                            double* __restrict__ r, double* __restrict__ rr_part) {
  __shared__ double sh[32];
  double rz_old, pq;
  sum_slots2(rz_part_old, pq_part, rz_old, pq, sh);
  const double alpha = rz_old / pq;
  double rr = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * N; i += gridDim.x * 256) {
    x[i] += alpha * p[i];
    const double rv = r[i] - alpha * q[i];
    r[i] = rv;
    rr += (w ? w[i] : 1.0) * rv * rv;
  }
  const double t = block_sum(rr, sh);
  if (threadIdx.x == 0) rr_part[blockIdx.x] = t;
  if (blockIdx.x == 0)
    for (int k = gridDim.x + threadIdx.x; k < kNPart; k += blockDim.x) rr_part[k] = 0.0;
}

void launch_pcg_update_noz(hipStream_t s, int N, const double* w, const double* p, const double* q,
                           const double* rz_part_old, const double* pq_part, double* x, double* r, double* rr_part) {
  const int n_blocks = std::max(1, std::min(kNPart, (3 * N + 255) / 256));
  hipLaunchKernelGGL(pcg_update_noz_kernel, dim3(n_blocks), dim3(256), 0, s, N, w, p, q, rz_part_old, pq_part, x, r,
                     rr_part);
}

// v <- D^-1 q (power iteration for lambda_max of D^-1 H)
__global__ void apply_dinv_kernel(int N, const double* __restrict__ Dinv, const double* __restrict__ q,
                                  double* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double* D = Dinv + (size_t)9 * i;
  const double a = q[3 * i], b = q[3 * i + 1], c = q[3 * i + 2];
  v[3 * i] = D[0] * a + D[1] * b + D[2] * c;
  v[3 * i + 1] = D[3] * a + D[4] * b + D[5] * c;
  v[3 * i + 2] = D[6] * a + D[7] * b + D[8] * c;
}
void launch_apply_dinv(hipStream_t s, int N, const double* Dinv, const double* q, double* v) {
  hipLaunchKernelGGL(apply_dinv_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, Dinv, q, v);
}
__global__ void scale_kernel(int n, double a, double* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] *= a;
}
void launch_scale(hipStream_t s, int n, double a, double* v) {
  hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, a, v);
}

// ---- Newton vector updates (SyncedNewton.cu:413-534) ---------------------------------------------
__global__ void newton_update_kernel(int N, const double* __restrict__ dv, double* __restrict__ v,
                                     const double* __restrict__ xp, const double* __restrict__ yp,
                                     const double* __restrict__ zp, double h, double* __restrict__ x,
                                     double* __restrict__ y, double* __restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double v0 = v[3 * i] + dv[3 * i], v1 = v[3 * i + 1] + dv[3 * i + 1], v2 = v[3 * i + 2] + dv[3 * i + 2];
  v[3 * i] = v0;
  v[3 * i + 1] = v1;
  v[3 * i + 2] = v2;
  x[i] = xp[i] + v0 * h;
  y[i] = yp[i] + v1 * h;
  z[i] = zp[i] + v2 * h;
}

void launch_newton_update(hipStream_t s, int N, const double* dv, double* v, const double* xp, const double* yp,
                          const double* zp, double h, double* x, double* y, double* z) {
  hipLaunchKernelGGL(newton_update_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, dv, v, xp, yp, zp, h, x, y, z);
}

__global__ void neg_kernel(int n, const double* __restrict__ g, double* __restrict__ r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) r[i] = -g[i];
}
void launch_axpy_neg(hipStream_t s, int n, const double* g, double* r) {
  hipLaunchKernelGGL(neg_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, g, r);
}

__global__ void dual_update_kernel(int nc, const double* __restrict__ c, double rho, double* __restrict__ lam) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nc) lam[i] += rho * c[i];
}
void launch_dual_update(hipStream_t s, int nc, const double* cons, double rho, double* lam) {
  if (nc <= 0) return;
  hipLaunchKernelGGL(dual_update_kernel, dim3((nc + 255) / 256), dim3(256), 0, s, nc, cons, rho, lam);
}

// ---- partition-interface pack/unpack (multi-GPU exchange buffers) --------------------------------
// buf[dim*slot + c] <-> field[dim*node + c]; slots index the GLOBAL interface list, identical on all ranks
__global__ void pack_kernel(int n, int dim, const int* __restrict__ node, const int* __restrict__ slot,
                            const double* __restrict__ src, double* __restrict__ buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * dim) return;
  const int k = t / dim, c = t - k * dim;
  buf[(size_t)dim * slot[k] + c] = src[(size_t)dim * node[k] + c];
}
__global__ void unpack_kernel(int n, int dim, const int* __restrict__ node, const int* __restrict__ slot,
                              const double* __restrict__ buf, double* __restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * dim) return;
  const int k = t / dim, c = t - k * dim;
  dst[(size_t)dim * node[k] + c] = buf[(size_t)dim * slot[k] + c];
}
void launch_pack(hipStream_t s, int n, int dim, const int* node, const int* slot, const double* src, double* buf) {
  if (n <= 0) return;
  hipLaunchKernelGGL(pack_kernel, dim3((n * dim + 255) / 256), dim3(256), 0, s, n, dim, node, slot, src, buf);
}
void launch_unpack(hipStream_t s, int n, int dim, const int* node, const int* slot, const double* buf, double* dst) {
  if (n <= 0) return;
  hipLaunchKernelGGL(unpack_kernel, dim3((n * dim + 255) / 256), dim3(256), 0, s, n, dim, node, slot, buf, dst);
}

__global__ void extract_diag_kernel(int N, Incidence inc, const double* __restrict__ Hval, double* __restrict__ D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int off0 = inc.off[i], deg = inc.off[i + 1] - off0, dp = inc.diagpos[i];
  const double* Hi = Hval + (size_t)9 * off0;
#pragma unroll
  for (int d = 0; d < 3; d++)
#pragma unroll
    for (int e = 0; e < 3; e++) D[(size_t)9 * i + 3 * d + e] = Hi[d * 3 * deg + 3 * dp + e];
}
__global__ void invert_diag_kernel(int N, const double* __restrict__ Dm, double* __restrict__ Dinv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double D[3][3];
#pragma unroll
  for (int d = 0; d < 3; d++)
#pragma unroll
    for (int e = 0; e < 3; e++) D[d][e] = Dm[(size_t)9 * i + 3 * d + e];
  const double det = D[0][0] * (D[1][1] * D[2][2] - D[1][2] * D[2][1]) -
                     D[0][1] * (D[1][0] * D[2][2] - D[1][2] * D[2][0]) +
                     D[0][2] * (D[1][0] * D[2][1] - D[1][1] * D[2][0]);
  const double id = 1.0 / det;
  double* o = Dinv + (size_t)9 * i;
  o[0] = (D[1][1] * D[2][2] - D[1][2] * D[2][1]) * id;
  o[1] = (D[0][2] * D[2][1] - D[0][1] * D[2][2]) * id;
  o[2] = (D[0][1] * D[1][2] - D[0][2] * D[1][1]) * id;
  o[3] = (D[1][2] * D[2][0] - D[1][0] * D[2][2]) * id;
  o[4] = (D[0][0] * D[2][2] - D[0][2] * D[2][0]) * id;
  o[5] = (D[0][2] * D[1][0] - D[0][0] * D[1][2]) * id;
  o[6] = (D[1][0] * D[2][1] - D[1][1] * D[2][0]) * id;
  o[7] = (D[0][1] * D[2][0] - D[0][0] * D[2][1]) * id;
  o[8] = (D[0][0] * D[1][1] - D[0][1] * D[1][0]) * id;
}
void launch_extract_diag(hipStream_t s, int N, const Incidence& inc, const double* Hval, double* D) {
  hipLaunchKernelGGL(extract_diag_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, inc, Hval, D);
}
void launch_invert_diag(hipStream_t s, int N, const double* D, double* Dinv) {
  hipLaunchKernelGGL(invert_diag_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, D, Dinv);
}

}  // namespace tlfea
