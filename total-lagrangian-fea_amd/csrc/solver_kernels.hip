// solver_kernels.hip -- vector kernels of the ALM/Newton step and the device linear solver.
//
// The reference solves H dv = -g with cuDSS (SyncedNewton.cu:995-1029,1103-1114).  Here H (SPD:
// M/h + h K + C_vis + h^2 rho J^T J) is solved by a 3x3-block-Jacobi preconditioned CG that streams
// the CSR values once per iteration.  All reductions use a fixed number of partial sums that every
// consumer re-adds in the same order, so results are bitwise reproducible run to run (the
// reference's atomics are not).
#include "tlfea_internal.h"

namespace tlfea {

// ---- deterministic reductions -------------------------------------------------------------------
// sum of kNPart partials by one workgroup of 256 threads, identical order everywhere
__device__ __forceinline__ double block_sum_parts(const double* __restrict__ part, double* sh, int n = kNPart) {
  const int t = threadIdx.x;
  double s = 0.0;
  for (int k = t; k < n; k += 256) s += part[k];
  sh[t] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) sh[t] += sh[t + w];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__device__ __forceinline__ double block_reduce(double v, double* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) sh[t] += sh[t + w];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void norm2_part_kernel(const double* __restrict__ a, const double* __restrict__ w,
                                                        int n, double* __restrict__ part) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += (w ? w[i] : 1.0) * a[i] * a[i];
  const double r = block_reduce(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double sh[256];
  const double r = block_sum_parts(part, sh);
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
__global__ __launch_bounds__(256) void pcg_init_kernel(int N, const double* __restrict__ b,
                                                      const double* __restrict__ Dinv, const double* __restrict__ w,
                                                      double* __restrict__ x, double* __restrict__ r,
                                                      double* __restrict__ z, double* __restrict__ p,
                                                      double* __restrict__ rz_part, double* __restrict__ bb_part) {
  __shared__ double sh[256];
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
    p[3 * i] = z0; p[3 * i + 1] = z1; p[3 * i + 2] = z2;
    const double w0 = w ? w[3 * i] : 1.0, w1 = w ? w[3 * i + 1] : 1.0, w2 = w ? w[3 * i + 2] : 1.0;
    rz += w0 * r0 * z0 + w1 * r1 * z1 + w2 * r2 * z2;
    bb += w0 * r0 * r0 + w1 * r1 * r1 + w2 * r2 * r2;
  }
  const double a = block_reduce(rz, sh);
  const double c = block_reduce(bb, sh);
  if (threadIdx.x == 0) {
    rz_part[blockIdx.x] = a;
    bb_part[blockIdx.x] = c;
  }
}

void launch_pcg_init(hipStream_t s, int N, const double* b, const double* Dinv, const double* w, double* x,
                     double* r, double* z, double* p, double* rz_part, double* bb_part) {
  hipLaunchKernelGGL(pcg_init_kernel, dim3(kNPart), dim3(256), 0, s, N, b, Dinv, w, x, r, z, p, rz_part, bb_part);
}

// q = H p_new with p_new = z + beta p_old formed on the fly (beta = rz_new/rz_old re-summed from the
// partials; first iteration: p_new = z), so CG needs no separate direction kernel: 2 launches/iteration.
// One node row (3 CSR rows) per wavefront pass; lane t walks the 3*deg (k,e) columns so the three value
// rows and the column-node list are read coalesced.  Each workgroup owns a contiguous chunk of rows
// (neighbouring rows share most of their column nodes -> the p/z gathers hit L1/L2).
__global__ __launch_bounds__(256) void spmv_dir_dot_kernel(int N, Incidence inc, const double* __restrict__ Hval,
                                                          const double* __restrict__ z,
                                                          const double* __restrict__ p_old, int first,
                                                          const double* __restrict__ rz_part_old,
                                                          const double* __restrict__ rz_part_new,
                                                          const double* __restrict__ w, double* __restrict__ p_new,
                                                          double* __restrict__ q, double* __restrict__ pq_part) {
  __shared__ double sh[256];
  double beta = 0.0;
  if (!first) {
    const double rz_old = block_sum_parts(rz_part_old, sh);
    const double rz_new = block_sum_parts(rz_part_new, sh);
    beta = rz_new / rz_old;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  double pq = 0.0;
  for (int i = r0 + wv; i < r1; i += 4) {
    const int off0 = inc.off[i], deg = inc.off[i + 1] - off0, row = 3 * deg;
    const double* Hi = Hval + (size_t)9 * off0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < row; t += 64) {
      const int k = t / 3, e = t - 3 * k;
      const int c = 3 * inc.cols[off0 + k] + e;
      const double pv = first ? z[c] : (z[c] + beta * p_old[c]);
      s0 += Hi[t] * pv;
      s1 += Hi[row + t] * pv;
      s2 += Hi[2 * row + t] * pv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s0 += __shfl_xor(s0, o);
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    if (lane < 3) {
      const int c = 3 * i + lane;
      const double pv = first ? z[c] : (z[c] + beta * p_old[c]);
      const double sv = (lane == 0) ? s0 : ((lane == 1) ? s1 : s2);
      p_new[c] = pv;
      q[c] = sv;
      pq += (w ? w[c] : 1.0) * pv * sv;
    }
  }
  const double r = block_reduce(pq, sh);
  if (threadIdx.x == 0) pq_part[blockIdx.x] = r;
}

void launch_spmv_dir_dot(hipStream_t s, int N, int n_blocks, const Incidence& inc, const double* Hval,
                         const double* z, const double* p_old, int first, const double* rz_part_old,
                         const double* rz_part_new, const double* w, double* p_new, double* q, double* pq_part) {
  hipLaunchKernelGGL(spmv_dir_dot_kernel, dim3(n_blocks), dim3(256), 0, s, N, inc, Hval, z, p_old, first, rz_part_old,
                     rz_part_new, w, p_new, q, pq_part);
}

// x += alpha p ; r -= alpha q ; z = Dinv r ; partials of r.z and r.r
__global__ __launch_bounds__(256) void pcg_update_kernel(int N, const double* __restrict__ Dinv,
                                                        const double* __restrict__ w, const double* __restrict__ p,
                                                        const double* __restrict__ q,
                                                        const double* __restrict__ rz_part_old,
                                                        const double* __restrict__ pq_part, int n_pq,
                                                        double* __restrict__ x, double* __restrict__ r,
                                                        double* __restrict__ z, double* __restrict__ rz_part_new,
                                                        double* __restrict__ rr_part) {
  __shared__ double sh[256];
  const double rz_old = block_sum_parts(rz_part_old, sh);
  const double pq = block_sum_parts(pq_part, sh, n_pq);
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
  const double a = block_reduce(rz, sh);
  const double c = block_reduce(rr, sh);
  if (threadIdx.x == 0) {
    rz_part_new[blockIdx.x] = a;
    rr_part[blockIdx.x] = c;
  }
}

void launch_pcg_update(hipStream_t s, int N, const double* Dinv, const double* w, const double* p, const double* q,
                       const double* rz_part_old, const double* pq_part, int n_pq, double* x, double* r, double* z,
                       double* rz_part_new, double* rr_part) {
  hipLaunchKernelGGL(pcg_update_kernel, dim3(kNPart), dim3(256), 0, s, N, Dinv, w, p, q, rz_part_old, pq_part, n_pq,
                     x, r, z, rz_part_new, rr_part);
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
__global__ void pack_kernel(int n, const int* __restrict__ idx, const double* __restrict__ src,
                            double* __restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = src[idx[i]];
}
__global__ void unpack_kernel(int n, const int* __restrict__ idx, const double* __restrict__ buf,
                              double* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = buf[i];
}
void launch_pack(hipStream_t s, int n, const int* idx, const double* src, double* buf) {
  if (n <= 0) return;
  hipLaunchKernelGGL(pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, idx, src, buf);
}
void launch_unpack(hipStream_t s, int n, const int* idx, const double* buf, double* dst) {
  if (n <= 0) return;
  hipLaunchKernelGGL(unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, idx, buf, dst);
}

}  // namespace tlfea
