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
bool row_map_tiled();
// XCD-aware row sweep of the two streaming kernels (polynomial step, SpMV).  Workgroups are dealt round-robin over the 8
// XCDs, so the workgroups b with the same b & 7 share one L2.  They take ONE contiguous eighth of the rows and sweep it in
// tiles together, so that each XCD's gathers stay inside its own eighth of the vector (plus the neighbouring planes).
// Measured at config C (round 3, A/B on one box): the cache-resident vertex level gains 8 % (15.6 -> 14.3 us per step); the
// fine level and the fp64 SpMV do NOT (262 -> 266 us, 752 -> 771 us: their 1.4x traffic is not cross-XCD duplication of the
// vector), so the eighths are used between 64 k and 600 k rows only.  TLFEA_XCD_ROWS=0 | 2: never | on every size.
static int xcd_rows_mode() {
  static const int m = std::getenv("TLFEA_XCD_ROWS") ? std::atoi(std::getenv("TLFEA_XCD_ROWS")) : 1;
  return m;
}
struct RowSweep {  // rows of this lane group: i = first, first + stride, ... < end
  int first, end, stride;
};
__device__ __forceinline__ RowSweep row_sweep(int N, int G, int grp, int xcd) {
  RowSweep r;
  if (xcd && (gridDim.x & 7) == 0 && N >= 65536 && (xcd == 2 || N <= 600000)) {
    const int nb = gridDim.x >> 3, lb = blockIdx.x >> 3, x = blockIdx.x & 7;
    const int per = (((N + 7) >> 3) + G - 1) / G * G;  // an eighth, rounded to whole tiles
    const int r0 = x * per;
    r.end = min(N, r0 + per);
    r.first = r0 + lb * G + grp;
    r.stride = nb * G;
  } else {
    r.first = blockIdx.x * G + grp;
    r.end = N;
    r.stride = gridDim.x * G;
  }
  return r;
}

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
template <bool NT, typename HT>
__device__ __forceinline__ double ld_stream(const HT* p) {
  // H is streamed once per CG iteration: a non-temporal load keeps it from evicting the gathered vectors
  return (double)(NT ? __builtin_nontemporal_load(p) : *p);
}

// FUSED: p_new = z + beta p_old is formed on the fly (small meshes: one launch fewer per iteration).
// !FUSED: p was written by pcg_direction_kernel; only p is gathered (large meshes: half the gather traffic).
// HT: double = H itself; float = its single-precision copy (same layout), the CG's working operator between two
// fp64 residual replacements (tlfea_api.hip, pcg()): products and sums stay fp64.
template <typename HT, bool FUSED, bool NT, int LANES>
__global__ __launch_bounds__(1024) void spmv_dir_dot_kernel(int N, Incidence inc, const HT* __restrict__ Hval,
                                                           const double* __restrict__ z,
                                                           const double* __restrict__ p_old, int first,
                                                           const double* __restrict__ rz_part_old,
                                                           const double* __restrict__ rz_part_new,
                                                           double* __restrict__ p_new, double* __restrict__ q,
                                                           double* __restrict__ pq_part, int tiled,
                                                           const double* __restrict__ wown) {
  __shared__ double sh[32];
  double beta = 0.0;
  if (FUSED && !first) {
    double rz_old, rz_new;
    sum_slots2(rz_part_old, rz_part_new, rz_old, rz_new, sh);
    beta = rz_old != 0.0 ? rz_new / rz_old : 0.0;  // exact convergence (r = 0): keep iterating harmlessly
  }
  // LANES (32 or 16) lanes walk one node row: 2 or 4 rows per wavefront at a time
  const int l32 = threadIdx.x & (LANES - 1), hw = threadIdx.x / LANES;
  constexpr int kGroups = 1024 / LANES;
  // tiled: the workgroups sweep the rows together (kGroups rows each, round robin) so that the chip gathers from a
  // narrow window of the vector that stays in L2; otherwise one contiguous chunk per workgroup (see cheb_lp_kernel)
  const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const RowSweep rs = row_sweep(N, kGroups, hw, tiled >> 1);
  const int r0 = (tiled & 1) ? rs.first - hw : blockIdx.x * rows_per_block;
  const int r1 = (tiled & 1) ? rs.end : min(N, r0 + rows_per_block);
  const int stride = (tiled & 1) ? rs.stride : kGroups;
  double pq = 0.0;
  for (int i = r0 + hw; i < r1; i += stride) {
    const int off0 = inc.off[i], deg = inc.off[i + 1] - off0, row = 3 * deg;
    const HT* Hi = Hval + (size_t)9 * off0;
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
      // boundary-sum partition: sv is this rank's partial (H_r p)_c and sum_r p.(H_r p) = p.Hp, so no weight (wown null);
      // overlapping partition: the row is complete on every rank that holds it, only its owner counts it (wown 1 | 0)
      pq += (wown ? wown[c] : 1.0) * pv * sv;
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
                         double* p_new, double* q, double* pq_part, bool fused, bool nt, const double* wown) {
  const dim3 g(spmv_grid(N)), b(1024);
  const int tiled = (row_map_tiled() ? 1 : 0) | (xcd_rows_mode() << 1);
  static const int lanes = std::getenv("TLFEA_SPMV_LANES") ? std::atoi(std::getenv("TLFEA_SPMV_LANES")) : 32;
#define TLFEA_SPMV(F, T, L)                                                                                       \
  hipLaunchKernelGGL((spmv_dir_dot_kernel<double, F, T, L>), g, b, 0, s, N, inc, Hval, z, p_old, first, rz_part_old,  \
                     rz_part_new, p_new, q, pq_part, tiled, wown)
  if (lanes == 16) {
    if (fused) TLFEA_SPMV(true, false, 16);
    else TLFEA_SPMV(false, false, 16);
  } else if (fused && nt) TLFEA_SPMV(true, true, 32);
  else if (fused) TLFEA_SPMV(true, false, 32);
  else if (nt) TLFEA_SPMV(false, true, 32);
  else TLFEA_SPMV(false, false, 32);
#undef TLFEA_SPMV
}

// the same launch on the single-precision copy of H
void launch_spmv_dir_dot_f32(hipStream_t s, int N, const Incidence& inc, const float* Hval32, const double* z,
                             const double* p_old, int first, const double* rz_part_old, const double* rz_part_new,
                             double* p_new, double* q, double* pq_part, bool fused, const double* wown) {
  const dim3 g(spmv_grid(N)), b(1024);
  const int tiled = (row_map_tiled() ? 1 : 0) | (xcd_rows_mode() << 1);
  if (fused)
    hipLaunchKernelGGL((spmv_dir_dot_kernel<float, true, false, 32>), g, b, 0, s, N, inc, Hval32, z, p_old, first,
                       rz_part_old, rz_part_new, p_new, q, pq_part, tiled, wown);
  else
    hipLaunchKernelGGL((spmv_dir_dot_kernel<float, false, false, 32>), g, b, 0, s, N, inc, Hval32, z, p_old, first,
                       rz_part_old, rz_part_new, p_new, q, pq_part, tiled, wown);
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
    beta = rz_old != 0.0 ? rz_new / rz_old : 0.0;
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
  const double alpha = pq != 0.0 ? rz_old / pq : 0.0;
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
// sc != null: scaled space of the low-precision path (Dinv is then (S D S)^-1 and res^ = S r)
__global__ __launch_bounds__(256) void cheb_init_kernel(int N, const double* __restrict__ Dinv,
                                                       const double* __restrict__ r,
                                                       const double* __restrict__ sc, const double* __restrict__ coef,
                                                       double* __restrict__ d, double* __restrict__ z,
                                                       double* __restrict__ res) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const double inv_theta = coef[0];
  double r0 = r[3 * i], r1 = r[3 * i + 1], r2 = r[3 * i + 2];
  if (sc) {
    r0 *= sc[3 * i];
    r1 *= sc[3 * i + 1];
    r2 *= sc[3 * i + 2];
  }
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

void launch_cheb_init(hipStream_t s, int N, const double* Dinv, const double* r, const double* sc,
                      const double* coef, double* d, double* z, double* res) {
  hipLaunchKernelGGL(cheb_init_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, Dinv, r, sc, coef, d, z, res);
}

// step k >= 1:  res -= H d_old ; d_new = c1 d_old + c2 D^-1 res ; z += d_new          (one launch, no reduction)
// LAST adds the partial slots of r.z for the enclosing CG (w: 1/multiplicity weights, multi-GPU).
template <bool LAST>
__global__ __launch_bounds__(1024) void cheb_step_kernel(int N, Incidence inc, const double* __restrict__ Hval,
                                                        const double* __restrict__ Dinv,
                                                        const double* __restrict__ d_old, const double* __restrict__ coef,
                                                        double* __restrict__ d_new, double* __restrict__ z,
                                                        double* __restrict__ res, const double* __restrict__ r,
                                                        const double* __restrict__ w, double* __restrict__ rz_part,
                                                        int tiled) {
  const double c1 = coef[0], c2 = coef[1];  // device-resident: the launch sequence is replayed as a hipGraph
  __shared__ double sh[32];
  const int l32 = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int r0 = tiled ? blockIdx.x * 32 : blockIdx.x * rows_per_block;
  const int r1 = tiled ? N : min(N, r0 + rows_per_block);
  const int stride = tiled ? gridDim.x * 32 : 32;
  double rz = 0.0;
  for (int i = r0 + hw; i < r1; i += stride) {
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
                      const double* d_old, const double* coef, double* d_new, double* z, double* res,
                      const double* r, const double* w, double* rz_part, bool last) {
  const dim3 g(spmv_grid(N)), b(1024);
  if (last)
    hipLaunchKernelGGL((cheb_step_kernel<true>), g, b, 0, s, N, inc, Hval, Dinv, d_old, coef, d_new, z, res, r, w,
                       rz_part, row_map_tiled() ? 1 : 0);
  else
    hipLaunchKernelGGL((cheb_step_kernel<false>), g, b, 0, s, N, inc, Hval, Dinv, d_old, coef, d_new, z, res, r, w,
                       rz_part, row_map_tiled() ? 1 : 0);
}

// ---- low-precision copy of H for the polynomial preconditioner -----------------------------------------
// The preconditioner only has to be a FIXED SPD operator close to H^-1, so its matrix need not be fp64: the
// Chebyshev steps stream a symmetrically scaled copy  Hs = S H S,  S = diag(H)^-1/2  (|entries| <= 1, unit
// diagonal) stored in fp16 or fp32 -- 1/4 or 1/2 of the bytes of the dominant kernel; rounding is symmetric, so
// the polynomial stays symmetric, and p(lambda) > 0 on the whole real axis keeps it positive definite.  CG on the
// fp64 H around it still converges to the fp64 answer (measured: same iteration counts as the fp64 polynomial).
// The copy is block-CSR: per (row node, column node) the 3x3 block row-major, first 8 entries in one aligned
// 16/32-byte record and the 9th in a side array, so ONE lane streams one block with two loads and gathers its
// three vector entries; the steps work in the scaled space (d^ = S^-1 d, res^ = S res) and the last one returns
// z = S z^.
// 8-bit storage of the scaled matrix (experiment, TLFEA_FINE_BITS=8): OCP e4m3 as gfx950 converts it in hardware
// (3 mantissa bits: 6 % relative error per entry, entries of S H S lie in [-1, 1])
struct Fp8 {
  unsigned char x;
  Fp8() = default;
  __device__ __forceinline__ Fp8(int) : x(0) {}
  __device__ __forceinline__ explicit Fp8(double h) {
    // stored times 256: e4m3 is normal down to 2^-6, so entries keep 3 mantissa bits down to 2^-14 of the diagonal
    x = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32((float)h * 256.f, 0.f, 0, false) & 0xff);
  }
  __device__ __forceinline__ explicit operator float() const {
    return __builtin_amdgcn_cvt_f32_fp8((int)x, 0) * 0.00390625f;
  }
};
template <typename HT>
struct alignas(8 * sizeof(HT)) Blk8 {
  HT v[8];
};

// sc = diag(D)^-1/2 per DOF;  Dinv_s = (S D S)^-1 = Dinv / (sc sc^T)
__global__ void lp_scale_kernel(int N, const double* __restrict__ D, const double* __restrict__ Dinv,
                                double* __restrict__ sc, double* __restrict__ Dinv_s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double s[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const double dd = D[(size_t)9 * i + 4 * c];
    s[c] = dd > 0.0 ? 1.0 / sqrt(dd) : 1.0;
    sc[3 * i + c] = s[c];
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int e = 0; e < 3; e++) Dinv_s[(size_t)9 * i + 3 * c + e] = Dinv[(size_t)9 * i + 3 * c + e] / (s[c] * s[e]);
}

void launch_lp_scale(hipStream_t s, int N, const double* D, const double* Dinv, double* sc, double* Dinv_s) {
  hipLaunchKernelGGL(lp_scale_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, D, Dinv, sc, Dinv_s);
}

// own / Dglob (multi-GPU, rank-local preconditioner): the copy is the block of the ASSEMBLED matrix on the nodes this
// rank owns -- rows and columns of replicated nodes owned by another rank are dropped (identity on their diagonal),
// and the diagonal blocks of owned partition-boundary nodes come from Dglob, the block diagonal summed over ranks
// (off-diagonal boundary-boundary blocks keep this rank's part only: still symmetric positive definite).
template <typename HT>
__global__ __launch_bounds__(256) void lp_convert_kernel(int N, Incidence inc, const double* __restrict__ Hval,
                                                        const double* __restrict__ sc, const int* __restrict__ own,
                                                        const double* __restrict__ Dglob, Blk8<HT>* __restrict__ B8,
                                                        HT* __restrict__ B1) {
  const int l32 = threadIdx.x & 31;
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5);  // 8 node rows per workgroup, 32 lanes per row
  if (i >= N) return;
  const int off0 = inc.off[i], deg = inc.off[i + 1] - off0, row = 3 * deg;
  const double* Hi = Hval + (size_t)9 * off0;
  const double si[3] = {sc[3 * i], sc[3 * i + 1], sc[3 * i + 2]};
  const bool own_i = !own || own[i];
  for (int k = l32; k < deg; k += 32) {
    const int c = inc.cols[off0 + k];
    const double sj[3] = {sc[3 * c], sc[3 * c + 1], sc[3 * c + 2]};
    const bool keep = own_i && (!own || own[c]);
    const bool diag = own && c == i;
    Blk8<HT> b;
    HT last = (HT)0;
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
      for (int e = 0; e < 3; e++) {
        double h = diag ? (own_i ? Dglob[(size_t)9 * i + 3 * d + e] * si[d] * sj[e] : (d == e ? 1.0 : 0.0))
                        : (keep ? Hi[(size_t)d * row + 3 * k + e] * si[d] * sj[e] : 0.0);
        const HT v = (HT)h;
        if (3 * d + e < 8) b.v[3 * d + e] = v;
        else last = v;
      }
    B8[off0 + k] = b;
    B1[off0 + k] = last;
  }
}

void launch_lp_convert(hipStream_t s, int N, const Incidence& inc, const double* Hval, const double* sc, const int* own,
                       const double* Dglob, void* B8, void* B1, int bits) {
  const dim3 g((N + 7) / 8), b(256);
  if (bits == 8)
    hipLaunchKernelGGL((lp_convert_kernel<Fp8>), g, b, 0, s, N, inc, Hval, sc, own, Dglob, (Blk8<Fp8>*)B8, (Fp8*)B1);
  else if (bits == 16)
    hipLaunchKernelGGL((lp_convert_kernel<_Float16>), g, b, 0, s, N, inc, Hval, sc, own, Dglob, (Blk8<_Float16>*)B8,
                       (_Float16*)B1);
  else
    hipLaunchKernelGGL((lp_convert_kernel<float>), g, b, 0, s, N, inc, Hval, sc, own, Dglob, (Blk8<float>*)B8,
                       (float*)B1);
}

// ---- ANCF: node-block (12 x 12) scaling of the polynomial's operator -------------------------------------------------
// The four coefficient vectors of an ANCF node (r, r_x, r_y, r_z) are strongly coupled (thickness / gradient stiffness
// against the position's): block-Jacobi on the 12 x 12 node blocks D12 = L L^T instead of the 3 x 3 ones cuts kappa by ~2
// on the plate (CPU prototype: 416 -> 315 polynomial steps per solve).  Done as a change of variables, so that every
// polynomial kernel stays as it is: the streamed copy holds  H^ = L^-1 H L^-T  (its 12 x 12 diagonal blocks are the
// identity: no diagonal scaling, (S D S)^-1 = I), the preconditioner is  z = L^-T p(H^) L^-1 r.
//
// Linv[p]: row-major 12 x 12, lower triangle = inverse of the Cholesky factor of node p's diagonal block, zeros above
__global__ void blk12_factor_kernel(int Np, Incidence inc, const double* __restrict__ Hval, double* __restrict__ Linv,
                                    float* __restrict__ Linv_f, double* __restrict__ sc, double* __restrict__ Dinv_s,
                                    int* __restrict__ err) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Np) return;
  double A[12][12];
  for (int a = 0; a < 4; a++) {
    const int i = 4 * p + a, off0 = inc.off[i], deg = inc.off[i + 1] - off0, row = 3 * deg;
    const int k0 = inc.diagpos[i] - a;  // the node's four columns are adjacent in the sorted row
    const double* Hi = Hval + (size_t)9 * off0;
    for (int b = 0; b < 4; b++)
      for (int d = 0; d < 3; d++)
        for (int e = 0; e < 3; e++) A[3 * a + d][3 * b + e] = Hi[(size_t)d * row + 3 * (k0 + b) + e];
  }
  bool bad = false;
  for (int j = 0; j < 12; j++) {  // Cholesky, lower, in place
    double dj = A[j][j];
    for (int q = 0; q < j; q++) dj -= A[j][q] * A[j][q];
    if (!(dj > 0.0)) {
      bad = true;
      dj = 1.0;
    }
    const double l = sqrt(dj), il = 1.0 / l;
    A[j][j] = l;
    for (int i = j + 1; i < 12; i++) {
      double v = A[i][j];
      for (int q = 0; q < j; q++) v -= A[i][q] * A[j][q];
      A[i][j] = v * il;
    }
  }
  if (bad) *err = 1;
  double* out = Linv + (size_t)144 * p;
  for (int c = 0; c < 12; c++) {  // column c of L^-1 by forward substitution
    double x[12];
    for (int i = 0; i < 12; i++) {
      double v = i == c ? 1.0 : 0.0;
      for (int q = c; q < i; q++) v -= A[i][q] * x[q];
      x[i] = i < c ? 0.0 : v / A[i][i];
    }
    for (int i = 0; i < 12; i++) {
      out[12 * i + c] = x[i];
      Linv_f[(size_t)144 * p + 12 * i + c] = (float)x[i];  // the copy the two applications per CG iteration stream
    }
  }
  for (int a = 0; a < 4; a++) {
    const int i = 4 * p + a;
    for (int d = 0; d < 3; d++) {
      sc[3 * i + d] = 1.0;
      for (int e = 0; e < 3; e++) Dinv_s[(size_t)9 * i + 3 * d + e] = d == e ? 1.0 : 0.0;
    }
  }
}

// out = L^-1 in (T = false) or L^-T in (T = true), node by node; 12 lanes per node
template <bool T>
__global__ void blk12_apply_kernel(int Np, const float* __restrict__ Linv, const double* __restrict__ in,
                                   double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = t / 12, r = t - 12 * p;
  if (p >= Np) return;
  const float* M = Linv + (size_t)144 * p;
  const double* v = in + (size_t)12 * p;
  double s = 0.0;
  if (T) {
    for (int q = r; q < 12; q++) s += M[12 * q + r] * v[q];
  } else {
    for (int q = 0; q <= r; q++) s += M[12 * r + q] * v[q];
  }
  out[(size_t)12 * p + r] = s;
}

// streamed copy of H^ = L^-1 H L^-T.  One wavefront per node p; a 12 x 12 block (p, q) is a tile of 16 lanes (a, b), four
// tiles at a time: every lane loads ITS 3 x 3 block of H once, then two passes through LDS --
//   T(a, b)  = sum_{a2 <= a} Linv_p(a, a2) H(a2, b)         (left factor; the tile's column b)
//   H^(a, b) = sum_{b2 <= b} T(a, b2) Linv_q(b, b2)^T       (right factor; the tile's row a)
// 8 x 27 multiply-adds per block at most (a single pass over both sums is 16 x 54 and re-reads every block 16 times:
// 8.3 ms at config D against 1.9 for this form).
__device__ __forceinline__ void wave_sync_s() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <typename HT>
__global__ __launch_bounds__(64) void lp_convert12_kernel(int Np, Incidence inc, const double* __restrict__ Hval,
                                                         const double* __restrict__ Linv, Blk8<HT>* __restrict__ B8,
                                                         HT* __restrict__ B1) {
  __shared__ double Hs[64][9], Ts[64][9];
  const int p = blockIdx.x, lane = threadIdx.x, tl = lane >> 4, a = (lane >> 2) & 3, b = lane & 3;
  const int i = 4 * p + a, off0 = inc.off[i], deg = inc.off[i + 1] - off0, row = 3 * deg, tiles = deg >> 2;
  const double* Lp = Linv + (size_t)144 * p + 36 * a;  // rows 3a .. 3a + 2 of Linv_p
  const double* Hi = Hval + (size_t)9 * off0;
  for (int t0 = 0; t0 < tiles; t0 += 4) {
    const int t = t0 + tl, k = 4 * t + b;
    const bool act = t < tiles;
    int q = 0;
    if (act) {
      q = inc.cols[off0 + k] >> 2;
#pragma unroll
      for (int d = 0; d < 3; d++)
#pragma unroll
        for (int e = 0; e < 3; e++) Hs[lane][3 * d + e] = Hi[(size_t)d * row + 3 * k + e];
    }
    wave_sync_s();
    if (act) {
      double T[9] = {};
      for (int a2 = 0; a2 <= a; a2++) {
        const double* Hb = Hs[(lane & ~12) | (a2 << 2)];
#pragma unroll
        for (int d = 0; d < 3; d++)
#pragma unroll
          for (int e = 0; e < 3; e++)
            T[3 * d + e] += Lp[12 * d + 3 * a2] * Hb[e] + Lp[12 * d + 3 * a2 + 1] * Hb[3 + e] + Lp[12 * d + 3 * a2 + 2] * Hb[6 + e];
      }
#pragma unroll
      for (int v = 0; v < 9; v++) Ts[lane][v] = T[v];
    }
    wave_sync_s();
    if (act) {
      const double* Lq = Linv + (size_t)144 * q + 36 * b;
      double R[9] = {};
      for (int b2 = 0; b2 <= b; b2++) {
        const double* Tb = Ts[(lane & ~3) | b2];
#pragma unroll
        for (int d = 0; d < 3; d++)
#pragma unroll
          for (int e = 0; e < 3; e++)
            R[3 * d + e] += Tb[3 * d] * Lq[12 * e + 3 * b2] + Tb[3 * d + 1] * Lq[12 * e + 3 * b2 + 1] + Tb[3 * d + 2] * Lq[12 * e + 3 * b2 + 2];
      }
      Blk8<HT> blk;
#pragma unroll
      for (int v = 0; v < 8; v++) blk.v[v] = (HT)R[v];
      B8[off0 + k] = blk;
      B1[off0 + k] = (HT)R[8];
    }
    wave_sync_s();
  }
}

void launch_blk12_factor(hipStream_t s, int Np, const Incidence& inc, const double* Hval, double* Linv, float* Linv_f,
                         double* sc, double* Dinv_s, int* err) {
  hipLaunchKernelGGL(blk12_factor_kernel, dim3((Np + 63) / 64), dim3(64), 0, s, Np, inc, Hval, Linv, Linv_f, sc, Dinv_s, err);
}
void launch_blk12_apply(hipStream_t s, int Np, const float* Linv, bool transpose, const double* in, double* out) {
  const dim3 g((unsigned)(((size_t)12 * Np + 255) / 256)), b(256);
  if (transpose) hipLaunchKernelGGL(blk12_apply_kernel<true>, g, b, 0, s, Np, Linv, in, out);
  else hipLaunchKernelGGL(blk12_apply_kernel<false>, g, b, 0, s, Np, Linv, in, out);
}
void launch_lp_convert12(hipStream_t s, int N, const Incidence& inc, const double* Hval, const double* Linv, void* B8,
                         void* B1, int bits) {
  const dim3 g(N / 4), b(64);
  if (bits == 16)
    hipLaunchKernelGGL(lp_convert12_kernel<_Float16>, g, b, 0, s, N / 4, inc, Hval, Linv, (Blk8<_Float16>*)B8, (_Float16*)B1);
  else
    hipLaunchKernelGGL(lp_convert12_kernel<float>, g, b, 0, s, N / 4, inc, Hval, Linv, (Blk8<float>*)B8, (float*)B1);
}

// sc_mask = own ? sc : 0 : scaling into / out of the polynomial that also drops the nodes another rank owns
__global__ void mask_scale_kernel(int N, const double* __restrict__ sc, const int* __restrict__ own,
                                  double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double m = own[i] ? 1.0 : 0.0;
  out[3 * i] = m * sc[3 * i];
  out[3 * i + 1] = m * sc[3 * i + 1];
  out[3 * i + 2] = m * sc[3 * i + 2];
}
void launch_mask_scale(hipStream_t s, int N, const double* sc, const int* own, double* out) {
  hipLaunchKernelGGL(mask_scale_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, sc, own, out);
}

// MODE 0: Chebyshev step (as cheb_step_kernel, scaled space); MODE 1: last step (returns z = S z^ and the r.z
// slots); MODE 2: q = Hs d only (multi-GPU path: q is summed over ranks before the vector update).
// L lanes walk the column nodes of one node row, two blocks per lane per round.  The kernel is bound by memory
// latency, not issue: a row is a chain  offsets -> columns/blocks -> gathered vector entries -> reduction ->
// epilogue operands -> stores.  So everything of the NEXT row that does not depend on data of this one (its
// offsets, the first round of columns and blocks, the epilogue operands res / D^-1 / d / z) is requested while this
// row's gathers are in flight, and the only exposed latency per row is the gather itself.
template <typename HT>
struct LpRound {  // one lane's share of a round: two blocks (8+1 entries each) and their column nodes
  Blk8<HT> a, b;
  HT a8, b8;
  int ca, cb;
};
struct LpEpilogue {  // what lanes 0..2 of a group need to finish component c of their row
  double e0, e1, e2, D0, D1, D2, dc, zc, sc, rc, wc;
};

template <typename HT, int L>
__device__ __forceinline__ void lp_load_round(LpRound<HT>& R, const Incidence& inc, const Blk8<HT>* __restrict__ B8,
                                              const HT* __restrict__ B1, int base, int deg, int k) {
  // lanes past the end of the row read its first block (always there: the diagonal) and are masked by the caller
  const int ga = base + (k < deg ? k : 0), gb = base + (k + L < deg ? k + L : 0);
  R.ca = inc.cols[ga];
  R.cb = inc.cols[gb];
  R.a = B8[ga];
  R.b = B8[gb];
  R.a8 = B1[ga];
  R.b8 = B1[gb];
}

template <int MODE>
__device__ __forceinline__ void lp_load_epilogue(LpEpilogue& E, int i, int c, const double* __restrict__ Dinv_s,
                                                 const double* __restrict__ sc, const double* __restrict__ d_old,
                                                 const double* __restrict__ z, const double* __restrict__ res,
                                                 const double* __restrict__ r, const double* __restrict__ w) {
  if (MODE == 2) return;
  E.e0 = res[3 * i];
  E.e1 = res[3 * i + 1];
  E.e2 = res[3 * i + 2];
  const double* D = Dinv_s + (size_t)9 * i + 3 * c;
  E.D0 = D[0];
  E.D1 = D[1];
  E.D2 = D[2];
  E.dc = d_old[3 * i + c];
  E.zc = z[3 * i + c];
  if (MODE == 1) {
    E.sc = sc[3 * i + c];
    E.rc = r[3 * i + c];
    E.wc = w ? w[3 * i + c] : 1.0;
  }
}

template <typename HT>
__device__ __forceinline__ void lp_fma_round(const LpRound<HT>& R, const double (&x)[3], const double (&y)[3],
                                             double& s0, double& s1, double& s2) {
  s0 += (double)R.a.v[0] * x[0] + (double)R.a.v[1] * x[1] + (double)R.a.v[2] * x[2];
  s1 += (double)R.a.v[3] * x[0] + (double)R.a.v[4] * x[1] + (double)R.a.v[5] * x[2];
  s2 += (double)R.a.v[6] * x[0] + (double)R.a.v[7] * x[1] + (double)R.a8 * x[2];
  s0 += (double)R.b.v[0] * y[0] + (double)R.b.v[1] * y[1] + (double)R.b.v[2] * y[2];
  s1 += (double)R.b.v[3] * y[0] + (double)R.b.v[4] * y[1] + (double)R.b.v[5] * y[2];
  s2 += (double)R.b.v[6] * y[0] + (double)R.b.v[7] * y[1] + (double)R.b8 * y[2];
}

// TILED: the workgroups sweep the rows together, G rows (one per lane group) at a time in round-robin order, instead
// of each owning one contiguous chunk: at any moment the whole chip then gathers from a narrow window of the vector
// (a few mesh planes) that stays in every XCD's L2, whereas 64 resident chunks per XCD spread over the whole vector
// do not fit (4 MiB) once the mesh is large.
template <typename HT, int L, int MODE, bool TILED>
__global__ __launch_bounds__(1024) void cheb_lp_kernel(int N, Incidence inc, const Blk8<HT>* __restrict__ B8,
                                                      const HT* __restrict__ B1, const double* __restrict__ Dinv_s,
                                                      const double* __restrict__ sc,
                                                      const double* __restrict__ d_old, const double* __restrict__ coef,
                                                      double* __restrict__ d_new, const double* __restrict__ z,
                                                      double* __restrict__ z_new, const double* __restrict__ res,
                                                      double* __restrict__ res_new, const double* __restrict__ r,
                                                      const double* __restrict__ w, double* __restrict__ out) {
  const double c1 = coef[0], c2 = coef[1];  // device-resident: the launch sequence is replayed as a hipGraph
  __shared__ double sh[32];
  constexpr int G = 1024 / L;
  const int lane = threadIdx.x & (L - 1), grp = threadIdx.x / L;
  const int c = lane < 3 ? lane : 0;
  int i, r1, stride;
  if (TILED) {
    i = blockIdx.x * G + grp;
    r1 = N;
    stride = gridDim.x * G;
  } else {
    const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block;
    i = r0 + grp;
    r1 = min(N, r0 + rows_per_block);
    stride = G;
  }
  double rz = 0.0;
  for (; i < r1; i += stride) {
    const int base = inc.off[i], deg = inc.off[i + 1] - base;
    // operands lanes 0..2 need to finish the row: requested first, back before the reduction is done
    LpEpilogue E;
    if (lane < 3) lp_load_epilogue<MODE>(E, i, c, Dinv_s, sc, d_old, z, res, r, w);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int k = lane; k < deg; k += 2 * L) {
      LpRound<HT> T;
      lp_load_round<HT, L>(T, inc, B8, B1, base, deg, k);
      const double mb = k + L < deg ? 1.0 : 0.0;
      const double* xa = d_old + 3 * (size_t)T.ca;
      const double* xb = d_old + 3 * (size_t)T.cb;
      const double xx[3] = {xa[0], xa[1], xa[2]};
      const double yy[3] = {mb * xb[0], mb * xb[1], mb * xb[2]};
      lp_fma_round<HT>(T, xx, yy, s0, s1, s2);
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
      s0 += __shfl_xor(s0, o);
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    if (lane < 3) {
      if (MODE == 2) {
        out[3 * i + c] = (c == 0) ? s0 : ((c == 1) ? s1 : s2);
      } else {
        const double e0 = E.e0 - s0, e1 = E.e1 - s1, e2 = E.e2 - s2;
        const double zc = E.D0 * e0 + E.D1 * e1 + E.D2 * e2;
        const double dn = c1 * E.dc + c2 * zc;
        const double zn = E.zc + dn;
        d_new[3 * i + c] = dn;
        if (MODE == 1) {
          const double zt = E.sc * zn;
          z_new[3 * i + c] = zt;
          rz += E.wc * E.rc * zt;
        } else {
          z_new[3 * i + c] = zn;
          res_new[3 * i + c] = (c == 0) ? e0 : ((c == 1) ? e1 : e2);
        }
      }
    }
  }
  if (MODE == 1) {
    const double t = block_sum(rz, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
    if (blockIdx.x == 0)
      for (int k = gridDim.x + threadIdx.x; k < kNPart; k += blockDim.x) out[k] = 0.0;
  }
}

bool row_map_tiled() {
  static const bool t = std::getenv("TLFEA_ROW_TILED") ? std::atoi(std::getenv("TLFEA_ROW_TILED")) != 0 : true;
  return t;
}

int lp_lanes(int N, int nnz_coef) {
  static const int forced = std::getenv("TLFEA_LP_LANES") ? std::atoi(std::getenv("TLFEA_LP_LANES")) : 0;
  if (forced == 8 || forced == 16 || forced == 32) return forced;
  const double avg = (double)nnz_coef / std::max(1, N);
  return avg > 48.0 ? 32 : 16;
}

void launch_cheb_lp(hipStream_t s, int N, int nnz_coef, const Incidence& inc, const void* B8, const void* B1, int bits,
                    const double* Dinv_s, const double* sc, const double* d_old, const double* coef, double* d_new,
                    const double* z, double* z_new, const double* res, double* res_new, const double* r,
                    const double* w, double* out, int mode) {
  const int L = lp_lanes(N, nnz_coef);
  const dim3 g(std::max(1, std::min(kNPart, (N + 1024 / L - 1) / (1024 / L)))), b(1024);
  const bool tiled = row_map_tiled();
#define TLFEA_LP(T, LL, M)                                                                                           \
  do {                                                                                                               \
    if (tiled)                                                                                                       \
      hipLaunchKernelGGL((cheb_lp_kernel<T, LL, M, true>), g, b, 0, s, N, inc, (const Blk8<T>*)B8, (const T*)B1,      \
                         Dinv_s, sc, d_old, coef, d_new, z, z_new, res, res_new, r, w, out);                       \
    else                                                                                                             \
      hipLaunchKernelGGL((cheb_lp_kernel<T, LL, M, false>), g, b, 0, s, N, inc, (const Blk8<T>*)B8, (const T*)B1,     \
                         Dinv_s, sc, d_old, coef, d_new, z, z_new, res, res_new, r, w, out);                       \
  } while (0)
#define TLFEA_LP_M(T, LL)                     \
  do {                                        \
    if (mode == 0) TLFEA_LP(T, LL, 0);        \
    else if (mode == 1) TLFEA_LP(T, LL, 1);   \
    else TLFEA_LP(T, LL, 2);                  \
  } while (0)
#define TLFEA_LP_L(T)                         \
  do {                                        \
    if (L == 8) TLFEA_LP_M(T, 8);             \
    else if (L == 16) TLFEA_LP_M(T, 16);      \
    else TLFEA_LP_M(T, 32);                   \
  } while (0)
  if (bits == 16) TLFEA_LP_L(_Float16);
  else TLFEA_LP_L(float);
#undef TLFEA_LP_L
#undef TLFEA_LP_M
#undef TLFEA_LP
}

// ---- the polynomial in single precision (single-GPU path) -------------------------------------------------
// Inside the preconditioner nothing needs fp64: the Chebyshev recurrence only reduces its residual by ~10x, so fp32
// vectors (d, res, z), an fp32 block-Jacobi factor and fp32 accumulation of the fp16/fp32 matrix entries change
// z by ~1e-6 relative -- CG treats that as a slightly inexact preconditioner (measured: same iteration counts, same
// attained true residual; the CG recurrences on H stay fp64).  It halves the gather bytes, the vector traffic and the
// registers per block in flight (more wavefronts per SIMD on a kernel that is bound by memory latency x occupancy).
//   init:   res^ = S r ; d = (SDS)^-1 res^ / theta ; z^ = d                         (fp64 r in, fp32 out)
//   step k: res^ -= Hs d ; d' = c1 d + c2 (SDS)^-1 res^ ; z^ += d'
//   last:   as a step, then z = S z^ (fp64 out) and the r.z slots
__global__ __launch_bounds__(256) void cheb32_init_kernel(int N, const float* __restrict__ Dinv_f,
                                                         const double* __restrict__ r, const double* __restrict__ sc,
                                                         const double* __restrict__ coef, float* __restrict__ d,
                                                         float* __restrict__ z, float* __restrict__ res, int row0) {
  const int i = row0 + blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float inv_theta = (float)coef[0];
  const float zw0 = coef[1] != 0.0 ? (float)coef[1] : 1.f;  // fourth-kind smoothers: z^0 = beta_1 d0
  const float r0 = (float)(r[3 * i] * sc[3 * i]), r1 = (float)(r[3 * i + 1] * sc[3 * i + 1]),
              r2 = (float)(r[3 * i + 2] * sc[3 * i + 2]);
  const float* D = Dinv_f + (size_t)9 * i;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float v = (D[3 * c] * r0 + D[3 * c + 1] * r1 + D[3 * c + 2] * r2) * inv_theta;
    d[3 * i + c] = v;
    z[3 * i + c] = zw0 * v;
  }
  res[3 * i] = r0;
  res[3 * i + 1] = r1;
  res[3 * i + 2] = r2;
}

void launch_cheb32_init(hipStream_t s, int N, const float* Dinv_f, const double* r, const double* sc,
                        const double* coef, float* d, float* z, float* res, int row0) {  // rows row0 .. N - 1
  if (N <= row0) return;
  hipLaunchKernelGGL(cheb32_init_kernel, dim3((N - row0 + 255) / 256), dim3(256), 0, s, N, Dinv_f, r, sc, coef, d, z, res,
                     row0);
}

// Dinv_f = float((S D S)^-1)
__global__ void to_float_kernel(size_t n, const double* __restrict__ a, float* __restrict__ b) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    b[i] = (float)a[i];
}
void launch_to_float(hipStream_t s, size_t n, const double* a, float* b) {
  hipLaunchKernelGGL(to_float_kernel, dim3((unsigned)std::min<size_t>(4096, (n + 255) / 256)), dim3(256), 0, s, n, a, b);
}

// One lane's two blocks of a round of a row (branch-free: lanes past the end of the row re-read its first block, which
// is always there, and are masked) ...
template <typename HT, int L>
struct C32Round {
  Blk8<HT> a, b;
  HT a8, b8;
  int ca, cb;
  float ma, mb;
  __device__ __forceinline__ void load(const Incidence& inc, const Blk8<HT>* __restrict__ B8,
                                       const HT* __restrict__ B1, int base, int deg, int k) {
    const int ga = base + (k < deg ? k : 0), gb = base + (k + L < deg ? k + L : 0);
    ma = k < deg ? 1.f : 0.f;
    mb = k + L < deg ? 1.f : 0.f;
    ca = inc.cols[ga];
    cb = inc.cols[gb];
    a = B8[ga];
    b = B8[gb];
    a8 = B1[ga];
    b8 = B1[gb];
  }
};
// ... and their contribution to the lane's three row sums once the gathered vector entries x (block a), y (block b)
// are back
template <typename HT, int L>
__device__ __forceinline__ void c32_fma(const C32Round<HT, L>& R, const float (&x)[3], const float (&y)[3], float& s0,
                                        float& s1, float& s2) {
  const float x0 = R.ma * x[0], x1 = R.ma * x[1], x2 = R.ma * x[2];
  const float y0 = R.mb * y[0], y1 = R.mb * y[1], y2 = R.mb * y[2];
  s0 += (float)R.a.v[0] * x0 + (float)R.a.v[1] * x1 + (float)R.a.v[2] * x2;
  s1 += (float)R.a.v[3] * x0 + (float)R.a.v[4] * x1 + (float)R.a.v[5] * x2;
  s2 += (float)R.a.v[6] * x0 + (float)R.a.v[7] * x1 + (float)R.a8 * x2;
  s0 += (float)R.b.v[0] * y0 + (float)R.b.v[1] * y1 + (float)R.b.v[2] * y2;
  s1 += (float)R.b.v[3] * y0 + (float)R.b.v[4] * y1 + (float)R.b.v[5] * y2;
  s2 += (float)R.b.v[6] * y0 + (float)R.b.v[7] * y1 + (float)R.b8 * y2;
}

// TWO: a lane group works on two rows at a time (i and i + stride).  A row is three dependent memory phases (offsets +
// epilogue operands -> columns + blocks -> gathered vector entries); two independent rows per group double the bytes
// in flight per wavefront at each phase.  Pays on large meshes (config C/D: -2..-4 %), costs parallelism on small
// ones (config B: +4 %), so the launcher picks by size.  All loads of the common path (first round) are branch-free,
// so they issue back to back.
// Rows are swept in tiles (see cheb_lp_kernel).  TPB threads per workgroup: 256 for the ordinary steps -- the grid is
// sized to what is resident at once (occupancy x CUs) so that the sweep stays one narrow window; 1024 for the last
// step, whose r.z partials must fit the kNPart reduction slots.
template <typename HT, int L, bool LAST, int TPB, bool TWO>
__global__ __launch_bounds__(TPB) void cheb32_kernel(int N, Incidence inc, const Blk8<HT>* __restrict__ B8,
                                                     const HT* __restrict__ B1, const float* __restrict__ Dinv_f,
                                                     const double* __restrict__ sc, const float* __restrict__ d_old,
                                                     const double* __restrict__ coef, float* __restrict__ d_new,
                                                     const float* __restrict__ z, float* __restrict__ z_new,
                                                     const float* __restrict__ res, float* __restrict__ res_new,
                                                     const double* __restrict__ r, double* __restrict__ z_out,
                                                     double* __restrict__ rz_part, C32Bnd bnd) {
  __shared__ double sh[32];
  const float c1 = (float)coef[0], c2 = (float)coef[1];
  const float zwf = bnd.zw ? (float)bnd.zw[0] : 1.f;
  constexpr int G = TPB / L;
  const int lane = threadIdx.x & (L - 1), grp = threadIdx.x / L;
  const int c = lane < 3 ? lane : 0;
  const RowSweep rsw = row_sweep(N, G, grp, bnd.xcd);
  const int r1 = rsw.end, stride = rsw.stride;
  int i = rsw.first;
  double rz = 0.0;
  for (; i < r1; i += (TWO ? 2 : 1) * stride) {
    const bool two = TWO && i + stride < r1;
    const int iA = i, iB = two ? i + stride : i;  // without a second row the first is shadowed (stores masked)
    // phase 1: offsets and the operands lanes 0..2 need to finish component c of each row
    const int baseA = inc.off[iA], degA = inc.off[iA + 1] - baseA;
    const int baseB = inc.off[iB], degB = inc.off[iB + 1] - baseB;
    float eA0 = 0.f, eA1 = 0.f, eA2 = 0.f, eB0 = 0.f, eB1 = 0.f, eB2 = 0.f, DA0 = 0.f, DA1 = 0.f, DA2 = 0.f, DB0 = 0.f,
          DB1 = 0.f, DB2 = 0.f, dcA = 0.f, zcA = 0.f, dcB = 0.f, zcB = 0.f;
    if (TWO || lane < 3) {  // TWO: branch-free (every lane loads, the group's addresses coincide); else lanes 0..2 only
      eA0 = res[3 * iA];
      eA1 = res[3 * iA + 1];
      eA2 = res[3 * iA + 2];
      const float* DA = Dinv_f + (size_t)9 * iA + 3 * c;
      DA0 = DA[0];
      DA1 = DA[1];
      DA2 = DA[2];
      dcA = d_old[3 * iA + c];
      zcA = z[3 * iA + c];
      if (TWO) {
        eB0 = res[3 * iB];
        eB1 = res[3 * iB + 1];
        eB2 = res[3 * iB + 2];
        const float* DB = Dinv_f + (size_t)9 * iB + 3 * c;
        DB0 = DB[0];
        DB1 = DB[1];
        DB2 = DB[2];
        dcB = d_old[3 * iB + c];
        zcB = z[3 * iB + c];
      }
    }
    // phase 2: first round of both rows
    C32Round<HT, L> RA, RB;
    RA.load(inc, B8, B1, baseA, degA, lane);
    if (TWO) RB.load(inc, B8, B1, baseB, degB, lane);
    else RB = RA;
    // phase 3: gathers
    float xA[3], yA[3], xB[3], yB[3];
    {
      const float* pa = d_old + 3 * (size_t)RA.ca;
      const float* pb = d_old + 3 * (size_t)RA.cb;
      const float* qa = d_old + 3 * (size_t)RB.ca;
      const float* qb = d_old + 3 * (size_t)RB.cb;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        xA[j] = pa[j];
        yA[j] = pb[j];
        xB[j] = TWO ? qa[j] : 0.f;
        yB[j] = TWO ? qb[j] : 0.f;
      }
    }
    float sA0 = 0.f, sA1 = 0.f, sA2 = 0.f, sB0 = 0.f, sB1 = 0.f, sB2 = 0.f;
    c32_fma<HT, L>(RA, xA, yA, sA0, sA1, sA2);
    if (TWO) c32_fma<HT, L>(RB, xB, yB, sB0, sB1, sB2);
    // further rounds of long rows (more than 2 L blocks)
    for (int k = lane + 2 * L; k < degA; k += 2 * L) {
      C32Round<HT, L> T;
      T.load(inc, B8, B1, baseA, degA, k);
      const float* pa = d_old + 3 * (size_t)T.ca;
      const float* pb = d_old + 3 * (size_t)T.cb;
      const float x[3] = {pa[0], pa[1], pa[2]}, y[3] = {pb[0], pb[1], pb[2]};
      c32_fma<HT, L>(T, x, y, sA0, sA1, sA2);
    }
    for (int k = lane + 2 * L; TWO && k < degB; k += 2 * L) {
      C32Round<HT, L> T;
      T.load(inc, B8, B1, baseB, degB, k);
      const float* pa = d_old + 3 * (size_t)T.ca;
      const float* pb = d_old + 3 * (size_t)T.cb;
      const float x[3] = {pa[0], pa[1], pa[2]}, y[3] = {pb[0], pb[1], pb[2]};
      c32_fma<HT, L>(T, x, y, sB0, sB1, sB2);
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
      sA0 += __shfl_xor(sA0, o);
      sA1 += __shfl_xor(sA1, o);
      sA2 += __shfl_xor(sA2, o);
      if (TWO) {
        sB0 += __shfl_xor(sB0, o);
        sB1 += __shfl_xor(sB1, o);
        sB2 += __shfl_xor(sB2, o);
      }
    }
    if (lane < 3) {
#pragma unroll
      for (int w = 0; w < 2; w++) {
        if (w == 1 && !two) break;
        const int iw = w ? iB : iA;
        float t0 = w ? sB0 : sA0, t1 = w ? sB1 : sA1, t2 = w ? sB2 : sA2;
        if (bnd.bslot) {  // multi-GPU: a partition-boundary row takes the sum over ranks of its partial rows
          const int sl = bnd.bslot[iw];
          if (sl >= 0) {
            t0 = (float)bnd.bsum[3 * sl];
            t1 = (float)bnd.bsum[3 * sl + 1];
            t2 = (float)bnd.bsum[3 * sl + 2];
          }
        }
        const float e0 = (w ? eB0 : eA0) - t0, e1 = (w ? eB1 : eA1) - t1, e2 = (w ? eB2 : eA2) - t2;
        const float dn = c1 * (w ? dcB : dcA) +
                         c2 * ((w ? DB0 : DA0) * e0 + (w ? DB1 : DA1) * e1 + (w ? DB2 : DA2) * e2);
        const float zn = (w ? zcB : zcA) + zwf * dn;
        d_new[3 * iw + c] = dn;
        if (LAST) {
          const double zt = sc[3 * iw + c] * (double)zn;
          z_out[3 * iw + c] = zt;
          rz += (bnd.w ? bnd.w[3 * iw + c] : 1.0) * r[3 * iw + c] * zt;
        } else {
          z_new[3 * iw + c] = zn;
          res_new[3 * iw + c] = (c == 0) ? e0 : ((c == 1) ? e1 : e2);
        }
      }
    }
  }
  if (LAST) {
    const double t = block_sum(rz, sh);
    if (threadIdx.x == 0) rz_part[blockIdx.x] = t;
    if (blockIdx.x == 0)
      for (int k = gridDim.x + threadIdx.x; k < kNPart; k += blockDim.x) rz_part[k] = 0.0;
  }
}

template <typename K>
static int resident_blocks(K kernel, int tpb) {  // workgroups of this kernel resident on the device at once
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, tpb, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  return per_cu * cus;
}

static int c32_bandwidth_rows() {
  static const int n = std::getenv("TLFEA_C32_BW_N") ? std::atoi(std::getenv("TLFEA_C32_BW_N")) : 100000;
  return n;
}

template <typename T, int LL>
static void launch_cheb32_t(hipStream_t s, int N, const Incidence& inc, const void* B8, const void* B1,
                            const float* Dinv_f, const double* sc, const float* d_old, const double* coef,
                            float* d_new, const float* z, float* z_new, const float* res, float* res_new,
                            const double* r, double* z_out, double* rz_part, bool last, C32Bnd bnd) {
  bnd.xcd = xcd_rows_mode();
#define TLFEA_C32_ARGS N, inc, (const Blk8<T>*)B8, (const T*)B1, Dinv_f, sc, d_old, coef, d_new, z, z_new, res, res_new, r, z_out, rz_part, bnd
  if (last) {
    constexpr int TPB = 1024, G = TPB / LL;
    const dim3 g(std::max(1, std::min(kNPart, (N + G - 1) / G))), b(TPB);
    hipLaunchKernelGGL((cheb32_kernel<T, LL, true, TPB, false>), g, b, 0, s, TLFEA_C32_ARGS);
  } else if (N > c32_bandwidth_rows()) {  // bandwidth regime: two rows per lane group, 256-thread workgroups, resident grid
    constexpr int TPB = 256, G = TPB / LL;
    static const int resident = resident_blocks(cheb32_kernel<T, LL, false, TPB, true>, TPB);
    const dim3 g(std::max(1, std::min(resident, (N + 2 * G - 1) / (2 * G)))), b(TPB);
    hipLaunchKernelGGL((cheb32_kernel<T, LL, false, TPB, true>), g, b, 0, s, TLFEA_C32_ARGS);
  } else {  // latency regime: as many independent groups as there are rows
    constexpr int TPB = 1024, G = TPB / LL;
    const dim3 g(std::max(1, std::min(kNPart, (N + G - 1) / G))), b(TPB);
    hipLaunchKernelGGL((cheb32_kernel<T, LL, false, TPB, false>), g, b, 0, s, TLFEA_C32_ARGS);
  }
#undef TLFEA_C32_ARGS
}

void launch_cheb32(hipStream_t s, int N, int nnz_coef, const Incidence& inc, const void* B8, const void* B1, int bits,
                   const float* Dinv_f, const double* sc, const float* d_old, const double* coef, float* d_new,
                   const float* z, float* z_new, const float* res, float* res_new, const double* r, double* z_out,
                   double* rz_part, bool last, C32Bnd bnd) {
  // lanes per row: two blocks per lane and round; the vertex level of the p-multigrid cycle has ~15 blocks per row
  // (8 lanes hold them in one round with every lane busy), quadratic tets ~30 (16 lanes), the ANCF shells 100+ (32)
  static const int forced = std::getenv("TLFEA_C32_LANES") ? std::atoi(std::getenv("TLFEA_C32_LANES")) : 0;
  const double avg = (double)nnz_coef / std::max(1, N);
  int L = lp_lanes(N, nnz_coef) >= 32 ? 32 : (avg <= 18.0 ? 8 : 16);
  if (forced == 8 || forced == 16 || forced == 32) L = forced;
#define TLFEA_C32(T, LL) \
  launch_cheb32_t<T, LL>(s, N, inc, B8, B1, Dinv_f, sc, d_old, coef, d_new, z, z_new, res, res_new, r, z_out, rz_part, last, bnd)
  if (bits == 8) {
    TLFEA_C32(Fp8, 16);   // fine level of T10 meshes only (the experiment's scope)
  } else if (bits == 16) {
    if (L == 32) TLFEA_C32(_Float16, 32);
    else if (L == 8) TLFEA_C32(_Float16, 8);
    else TLFEA_C32(_Float16, 16);
  } else {
    if (L == 32) TLFEA_C32(float, 32);
    else if (L == 8) TLFEA_C32(float, 8);
    else TLFEA_C32(float, 16);
  }
#undef TLFEA_C32
}

// Multi-GPU: this rank's part of (Hs d) on a list of rows (the partition-boundary rows), 16 lanes per row, written
// as doubles straight into the exchange buffer (slot order agreed by all ranks); the fused step then reads the sum.
template <typename HT>
__global__ __launch_bounds__(256) void spmv32_rows_kernel(int n_rows, const int* __restrict__ rows,
                                                         const int* __restrict__ slots, Incidence inc,
                                                         const Blk8<HT>* __restrict__ B8, const HT* __restrict__ B1,
                                                         const float* __restrict__ d, double* __restrict__ out) {
  const int lane = threadIdx.x & 15;
  const int k = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (k >= n_rows) return;
  const int i = rows[k], base = inc.off[i], deg = inc.off[i + 1] - base;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int t = lane; t < deg; t += 16) {
    const int g = base + t;
    const Blk8<HT> a = B8[g];
    const HT a8 = B1[g];
    const float* x = d + 3 * (size_t)inc.cols[g];
    const float x0 = x[0], x1 = x[1], x2 = x[2];
    s0 += (float)a.v[0] * x0 + (float)a.v[1] * x1 + (float)a.v[2] * x2;
    s1 += (float)a.v[3] * x0 + (float)a.v[4] * x1 + (float)a.v[5] * x2;
    s2 += (float)a.v[6] * x0 + (float)a.v[7] * x1 + (float)a8 * x2;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o);
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if (lane < 3) out[3 * (size_t)slots[k] + lane] = (double)(lane == 0 ? s0 : (lane == 1 ? s1 : s2));
}
void launch_spmv32_rows(hipStream_t s, int n_rows, const int* rows, const int* slots, const Incidence& inc, const void* B8,
                        const void* B1, int bits, const float* d, double* out) {
  if (n_rows <= 0) return;
  const dim3 g((n_rows + 15) / 16), b(256);
  if (bits == 16)
    hipLaunchKernelGGL(spmv32_rows_kernel<_Float16>, g, b, 0, s, n_rows, rows, slots, inc, (const Blk8<_Float16>*)B8,
                       (const _Float16*)B1, d, out);
  else
    hipLaunchKernelGGL(spmv32_rows_kernel<float>, g, b, 0, s, n_rows, rows, slots, inc, (const Blk8<float>*)B8,
                       (const float*)B1, d, out);
}

// ---- two-level p-multigrid for T10 (pmg_host.h): Galerkin coarse operator and grid transfers ----------------
// Hc = P^T H P by gather: one thread owns one coarse 3x3 block and adds its contributing fine blocks in ascending
// fine-block order with weights 1, 1/2, 1/4 (no atomics: bitwise reproducible).  H and Hc in the reference's
// DOF-level layout (node row -> [d][k][e]).
__global__ __launch_bounds__(256) void pmg_galerkin_kernel(int nnz_c, const int* __restrict__ c_off,
                                                          const int* __restrict__ cblk_row,
                                                          const int* __restrict__ con_off,
                                                          const int* __restrict__ con_base,
                                                          const int* __restrict__ con_deg,
                                                          const float* __restrict__ con_w,
                                                          const double* __restrict__ Hf, double* __restrict__ Hc) {
  // 16 lanes per coarse block: each lane adds every 16th contribution (about 40 per block), then a fixed-order
  // butterfly; con_base = 9 off[i] + 3 k and con_deg = deg(i) of the contributing fine block are precomputed so that
  // a contribution is one independent round trip instead of a chain of four
  const int lane = threadIdx.x & 15;
  const int cb = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (cb >= nnz_c) return;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int u = con_off[cb] + lane; u < con_off[cb + 1]; u += 16) {
    const double w = (double)con_w[u];
    const int deg3 = 3 * con_deg[u];
    const double* Hi = Hf + con_base[u];
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
      for (int e = 0; e < 3; e++) acc[3 * d + e] += w * Hi[(size_t)d * deg3 + e];
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1)
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] += __shfl_xor(acc[k], o);
  if (lane >= 9) return;
  const int dd = lane / 3, ee = lane - 3 * dd;
  double v = acc[0];
#pragma unroll
  for (int k = 1; k < 9; k++) v = (lane == k) ? acc[k] : v;
  const int I = cblk_row[cb], oc = c_off[I], degc = c_off[I + 1] - oc, pc = cb - oc;
  Hc[(size_t)9 * oc + (size_t)dd * 3 * degc + 3 * pc + ee] = v;
}
void launch_pmg_galerkin(hipStream_t s, int nnz_c, const int* c_off, const int* cblk_row, const int* con_off,
                         const int* con_base, const int* con_deg, const float* con_w, const double* Hf, double* Hc) {
  hipLaunchKernelGGL(pmg_galerkin_kernel, dim3((nnz_c + 15) / 16), dim3(256), 0, s, nnz_c, c_off, cblk_row, con_off,
                     con_base, con_deg, con_w, Hf, Hc);
}

// Restriction fused with the coarse polynomial's start vectors.  The fine residual lives in the fine scaled space
// (res^ = S_f res), the coarse system in its own:  r^_c = S_c P^T S_f^-1 res^ ;  d = (S_c D_c S_c)^-1 r^_c / theta_c
__global__ __launch_bounds__(256) void pmg_restrict_init_kernel(
    int Nc, const int* __restrict__ child_off, const int* __restrict__ child, const float* __restrict__ child_w,
    const float* __restrict__ res_f, const double* __restrict__ sc_f, const double* __restrict__ sc_c,
    const float* __restrict__ Dinv_c, const double* __restrict__ coef_c, float* __restrict__ d_c,
    float* __restrict__ z_c, float* __restrict__ res_c, const int* __restrict__ bslot, const double* __restrict__ bsum) {
  // 16 lanes per coarse node: the ~15 children are loaded in parallel and summed by a fixed-order butterfly
  const int lane = threadIdx.x & 15;
  const int I = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (I >= Nc) return;
  double r[3] = {0.0, 0.0, 0.0};
  for (int t = child_off[I] + lane; t < child_off[I + 1]; t += 16) {
    const int n = child[t];
    const double w = (double)child_w[t];
#pragma unroll
    for (int c = 0; c < 3; c++) r[c] += w * (double)res_f[3 * n + c] / sc_f[3 * n + c];
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    r[0] += __shfl_xor(r[0], o);
    r[1] += __shfl_xor(r[1], o);
    r[2] += __shfl_xor(r[2], o);
  }
  if (lane >= 3) return;
  if (bslot && bslot[I] >= 0) {  // multi-GPU: a partition-boundary coarse node takes the sum over ranks (restrict_rows)
    const double* b = bsum + 3 * (size_t)bslot[I];
    r[0] = b[0];
    r[1] = b[1];
    r[2] = b[2];
  }
  const int c = lane;
  const float inv_theta = (float)coef_c[0];
  const float rs0 = (float)(r[0] * sc_c[3 * I]), rs1 = (float)(r[1] * sc_c[3 * I + 1]), rs2 = (float)(r[2] * sc_c[3 * I + 2]);
  const float* D = Dinv_c + (size_t)9 * I + 3 * c;
  const float v = (D[0] * rs0 + D[1] * rs1 + D[2] * rs2) * inv_theta;
  d_c[3 * I + c] = v;
  z_c[3 * I + c] = v;
  res_c[3 * I + c] = (c == 0) ? rs0 : ((c == 1) ? rs1 : rs2);
}
void launch_pmg_restrict_init(hipStream_t s, int Nc, const int* child_off, const int* child, const float* child_w,
                              const float* res_f, const double* sc_f, const double* sc_c, const float* Dinv_c,
                              const double* coef_c, float* d_c, float* z_c, float* res_c, const int* bslot,
                              const double* bsum) {
  hipLaunchKernelGGL(pmg_restrict_init_kernel, dim3((Nc + 15) / 16), dim3(256), 0, s, Nc, child_off, child, child_w,
                     res_f, sc_f, sc_c, Dinv_c, coef_c, d_c, z_c, res_c, bslot, bsum);
}
__global__ __launch_bounds__(256) void pmg_restrict_rows_kernel(int n_rows, const int* __restrict__ rows,
                                                               const int* __restrict__ slots,
                                                               const int* __restrict__ child_off,
                                                               const int* __restrict__ child,
                                                               const float* __restrict__ child_w,
                                                               const float* __restrict__ res_f,
                                                               const double* __restrict__ sc_f, double* __restrict__ out) {
  const int lane = threadIdx.x & 15;
  const int k = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (k >= n_rows) return;
  const int I = rows[k];
  double r[3] = {0.0, 0.0, 0.0};
  for (int t = child_off[I] + lane; t < child_off[I + 1]; t += 16) {
    const int n = child[t];
    const double w = (double)child_w[t];
#pragma unroll
    for (int c = 0; c < 3; c++) r[c] += w * (double)res_f[3 * n + c] / sc_f[3 * n + c];
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    r[0] += __shfl_xor(r[0], o);
    r[1] += __shfl_xor(r[1], o);
    r[2] += __shfl_xor(r[2], o);
  }
  if (lane < 3) out[3 * (size_t)slots[k] + lane] = lane == 0 ? r[0] : (lane == 1 ? r[1] : r[2]);
}
void launch_pmg_restrict_rows(hipStream_t s, int n_rows, const int* rows, const int* slots, const int* child_off,
                              const int* child, const float* child_w, const float* res_f, const double* sc_f, double* out) {
  if (n_rows <= 0) return;
  hipLaunchKernelGGL(pmg_restrict_rows_kernel, dim3((n_rows + 15) / 16), dim3(256), 0, s, n_rows, rows, slots, child_off,
                     child, child_w, res_f, sc_f, out);
}

// Prolongation of the coarse correction:  corr^ = S_f^-1 P S_c z^_c ;  z^ += corr^ ;  d := corr^ (the next fine step
// subtracts Hs corr^ from the residual and starts the post-smoothing polynomial)
__global__ __launch_bounds__(256) void pmg_prolong_kernel(int N, const int* __restrict__ par0,
                                                         const int* __restrict__ par1,
                                                         const float* __restrict__ z_c,
                                                         const double* __restrict__ sc_c,
                                                         const double* __restrict__ sc_f, float* __restrict__ z_f,
                                                         float* __restrict__ d_f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int a = par0[i], b = par1[i];
  if (a < 0) {  // no coarse parent (ANCF gradient coefficients): the correction is zero here
    d_f[3 * i] = d_f[3 * i + 1] = d_f[3 * i + 2] = 0.f;
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const double ea = sc_c[3 * a + c] * (double)z_c[3 * a + c];
    const double e = (a == b) ? ea : 0.5 * (ea + sc_c[3 * b + c] * (double)z_c[3 * b + c]);
    const float corr = (float)(e / sc_f[3 * i + c]);
    d_f[3 * i + c] = corr;
    z_f[3 * i + c] += corr;
  }
}
void launch_pmg_prolong(hipStream_t s, int N, const int* par0, const int* par1, const float* z_c, const double* sc_c,
                        const double* sc_f, float* z_f, float* d_f) {
  hipLaunchKernelGGL(pmg_prolong_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, par0, par1, z_c, sc_c, sc_f, z_f, d_f);
}

// ---- third level (pmg_host.h agg_build): rigid-body-mode aggregation of the vertex level ---------------------------
// Level-3 node 2A carries the translation t_A, node 2A+1 the rotation w_A of aggregate A;  u_i = t_A + w_A x r_i with
// r_i = x_i - c_A, i.e. the prolongation blocks are W_i0 = I, W_i1 = -[r_i]x (r_i is stored as zero where the
// aggregate's rotations are unusable, which switches them off everywhere below).
__device__ __forceinline__ void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// H3 = P2^T Hc P2 by gather: one thread owns one aggregate pair (A, B) = four 3x3 blocks and adds the vertex-level
// blocks (i in A, j in B) in ascending order (no atomics).  Both matrices in the DOF-level layout [row][d][k][e].
__global__ __launch_bounds__(128) void agg_galerkin_kernel(int n_pairs, const int* __restrict__ pair_A,
                                                          const int* __restrict__ pair_pos, const int* __restrict__ pair_B,
                                                          const int* __restrict__ pcon_off, const int* __restrict__ pcon_base,
                                                          const int* __restrict__ pcon_deg, const int* __restrict__ pcon_i,
                                                          const int* __restrict__ pcon_j, const double* __restrict__ rvec,
                                                          const int* __restrict__ active, const int* __restrict__ off3,
                                                          const double* __restrict__ Hc, double* __restrict__ H3) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  double tt[3][3] = {{0}}, tr[3][3] = {{0}}, rt[3][3] = {{0}}, rr[3][3] = {{0}};
  for (int u = pcon_off[p]; u < pcon_off[p + 1]; u++) {
    const int deg3 = 3 * pcon_deg[u];
    const double* Hi = Hc + pcon_base[u];
    const double* ri = rvec + 3 * (size_t)pcon_i[u];
    const double* rj = rvec + 3 * (size_t)pcon_j[u];
    double K[3][3], KR[3][3];
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
      for (int e = 0; e < 3; e++) K[d][e] = Hi[(size_t)d * deg3 + e];
    // KR = K (-[rj]x): row d of KR = -(K_d x rj)... (K [r]x)_de = sum_m K_dm [r]x_me, [r]x = [[0,-r2,r1],[r2,0,-r0],[-r1,r0,0]]
#pragma unroll
    for (int d = 0; d < 3; d++) {
      KR[d][0] = -(K[d][1] * rj[2] - K[d][2] * rj[1]);
      KR[d][1] = -(K[d][2] * rj[0] - K[d][0] * rj[2]);
      KR[d][2] = -(K[d][0] * rj[1] - K[d][1] * rj[0]);
    }
    // W_i1^T M = [ri]x M: column e of the result = ri x (column e of M)
#pragma unroll
    for (int e = 0; e < 3; e++) {
      const double ck[3] = {K[0][e], K[1][e], K[2][e]}, ckr[3] = {KR[0][e], KR[1][e], KR[2][e]};
      double a[3], b[3];
      cross3(ri, ck, a);
      cross3(ri, ckr, b);
#pragma unroll
      for (int d = 0; d < 3; d++) {
        tt[d][e] += K[d][e];
        tr[d][e] += KR[d][e];
        rt[d][e] += a[d];
        rr[d][e] += b[d];
      }
    }
  }
  const int A = pair_A[p], B = pair_B[p], pos = pair_pos[p];
  if (A == B && active[A] <= 0) {  // unusable rotations (0) / empty grid cell (-1): identity keeps the level-3 matrix
#pragma unroll                   // definite, the modes stay at zero (on several ranks the identities add up: harmless)
    for (int d = 0; d < 3; d++)
#pragma unroll
      for (int e = 0; e < 3; e++) {
        rr[d][e] = (d == e) ? 1.0 : 0.0;
        if (active[A] < 0) tt[d][e] = (d == e) ? 1.0 : 0.0;
      }
  }
#pragma unroll
  for (int s2 = 0; s2 < 2; s2++) {
    const int row = 2 * A + s2, o3 = off3[row], deg3 = 3 * (off3[row + 1] - o3);
#pragma unroll
    for (int t2 = 0; t2 < 2; t2++) {
      double* out = H3 + (size_t)9 * o3 + 3 * (2 * pos + t2);
#pragma unroll
      for (int d = 0; d < 3; d++)
#pragma unroll
        for (int e = 0; e < 3; e++)
          out[(size_t)d * deg3 + e] = s2 == 0 ? (t2 == 0 ? tt[d][e] : tr[d][e]) : (t2 == 0 ? rt[d][e] : rr[d][e]);
    }
  }
}
void launch_agg_galerkin(hipStream_t s, int n_pairs, const int* pair_A, const int* pair_pos, const int* pair_B,
                         const int* pcon_off, const int* pcon_base, const int* pcon_deg, const int* pcon_i,
                         const int* pcon_j, const double* rvec, const int* active, const int* off3, const double* Hc,
                         double* H3) {
  hipLaunchKernelGGL(agg_galerkin_kernel, dim3((n_pairs + 127) / 128), dim3(128), 0, s, n_pairs, pair_A, pair_pos, pair_B,
                     pcon_off, pcon_base, pcon_deg, pcon_i, pcon_j, rvec, active, off3, Hc, H3);
}

// r^_3 = S_3 P2^T S_2^-1 res^_2 ;  d = (S_3 D_3 S_3)^-1 r^_3 / theta_3 ; z = d ; res = r^_3   (16 lanes per level-3 node)
__global__ __launch_bounds__(256) void agg_restrict_init_kernel(
    int N3, const int* __restrict__ mem_off, const int* __restrict__ mem, const double* __restrict__ rvec,
    const float* __restrict__ res2, const double* __restrict__ sc2, const double* __restrict__ sc3,
    const float* __restrict__ Dinv3, const double* __restrict__ coef3, float* __restrict__ d3, float* __restrict__ z3,
    float* __restrict__ res3) {
  const int lane = threadIdx.x & 15;
  const int I = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (I >= N3) return;
  const int A = I >> 1, rot = I & 1;
  double r[3] = {0.0, 0.0, 0.0};
  for (int t = mem_off[A] + lane; t < mem_off[A + 1]; t += 16) {
    const int n = mem[t];
    const double v[3] = {(double)res2[3 * n] / sc2[3 * n], (double)res2[3 * n + 1] / sc2[3 * n + 1],
                         (double)res2[3 * n + 2] / sc2[3 * n + 2]};
    if (rot) {
      double c[3];
      cross3(rvec + 3 * (size_t)n, v, c);  // W_i1^T v = r_i x v
      r[0] += c[0]; r[1] += c[1]; r[2] += c[2];
    } else {
      r[0] += v[0]; r[1] += v[1]; r[2] += v[2];
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    r[0] += __shfl_xor(r[0], o);
    r[1] += __shfl_xor(r[1], o);
    r[2] += __shfl_xor(r[2], o);
  }
  if (lane >= 3) return;
  const int c = lane;
  const float inv_theta = (float)coef3[0];
  const float rs0 = (float)(r[0] * sc3[3 * I]), rs1 = (float)(r[1] * sc3[3 * I + 1]), rs2 = (float)(r[2] * sc3[3 * I + 2]);
  const float* D = Dinv3 + (size_t)9 * I + 3 * c;
  const float v = (D[0] * rs0 + D[1] * rs1 + D[2] * rs2) * inv_theta;
  d3[3 * I + c] = v;
  z3[3 * I + c] = v;
  res3[3 * I + c] = (c == 0) ? rs0 : ((c == 1) ? rs1 : rs2);
}
void launch_agg_restrict_init(hipStream_t s, int N3, const int* mem_off, const int* mem, const double* rvec,
                              const float* res2, const double* sc2, const double* sc3, const float* Dinv3,
                              const double* coef3, float* d3, float* z3, float* res3) {
  hipLaunchKernelGGL(agg_restrict_init_kernel, dim3((N3 + 15) / 16), dim3(256), 0, s, N3, mem_off, mem, rvec, res2, sc2,
                     sc3, Dinv3, coef3, d3, z3, res3);
}

// corr^_2 = S_2^-1 P2 S_3 z^_3 ;  z^_2 += corr ; d_2 := corr
// the same restriction in two launches (overlapping partition): this rank's share of the level-3 residual over its OWNED
// member vertices as doubles -- summed over ranks by the caller -- then the level-3 start vectors from the sum
__global__ __launch_bounds__(256) void agg_restrict_kernel(int N3, const int* __restrict__ mem_off,
                                                          const int* __restrict__ mem, const double* __restrict__ rvec,
                                                          const float* __restrict__ res2, const double* __restrict__ sc2,
                                                          double* __restrict__ r3) {
  const int lane = threadIdx.x & 15;
  const int I = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (I >= N3) return;
  const int A = I >> 1, rot = I & 1;
  double r[3] = {0.0, 0.0, 0.0};
  for (int t = mem_off[A] + lane; t < mem_off[A + 1]; t += 16) {
    const int n = mem[t];
    const double v[3] = {(double)res2[3 * n] / sc2[3 * n], (double)res2[3 * n + 1] / sc2[3 * n + 1],
                         (double)res2[3 * n + 2] / sc2[3 * n + 2]};
    if (rot) {
      double c[3];
      cross3(rvec + 3 * (size_t)n, v, c);
      r[0] += c[0]; r[1] += c[1]; r[2] += c[2];
    } else {
      r[0] += v[0]; r[1] += v[1]; r[2] += v[2];
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    r[0] += __shfl_xor(r[0], o);
    r[1] += __shfl_xor(r[1], o);
    r[2] += __shfl_xor(r[2], o);
  }
  if (lane < 3) r3[3 * I + lane] = lane == 0 ? r[0] : (lane == 1 ? r[1] : r[2]);
}
__global__ __launch_bounds__(256) void agg_init3_kernel(int N3, const double* __restrict__ r3, const double* __restrict__ sc3,
                                                       const float* __restrict__ Dinv3, const double* __restrict__ coef3,
                                                       float* __restrict__ d3, float* __restrict__ z3,
                                                       float* __restrict__ res3) {
  const int I = blockIdx.x * 256 + threadIdx.x;
  if (I >= N3) return;
  const float inv_theta = (float)coef3[0];
  const float rs0 = (float)(r3[3 * I] * sc3[3 * I]), rs1 = (float)(r3[3 * I + 1] * sc3[3 * I + 1]),
              rs2 = (float)(r3[3 * I + 2] * sc3[3 * I + 2]);
  const float* D = Dinv3 + (size_t)9 * I;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float v = (D[3 * c] * rs0 + D[3 * c + 1] * rs1 + D[3 * c + 2] * rs2) * inv_theta;
    d3[3 * I + c] = v;
    z3[3 * I + c] = v;
  }
  res3[3 * I] = rs0;
  res3[3 * I + 1] = rs1;
  res3[3 * I + 2] = rs2;
}
void launch_agg_restrict(hipStream_t s, int N3, const int* mem_off, const int* mem, const double* rvec, const float* res2,
                         const double* sc2, double* r3) {
  hipLaunchKernelGGL(agg_restrict_kernel, dim3((N3 + 15) / 16), dim3(256), 0, s, N3, mem_off, mem, rvec, res2, sc2, r3);
}
void launch_agg_init3(hipStream_t s, int N3, const double* r3, const double* sc3, const float* Dinv3, const double* coef3,
                      float* d3, float* z3, float* res3) {
  hipLaunchKernelGGL(agg_init3_kernel, dim3((N3 + 255) / 256), dim3(256), 0, s, N3, r3, sc3, Dinv3, coef3, d3, z3, res3);
}
__global__ __launch_bounds__(256) void agg_prolong_kernel(int Nc, const int* __restrict__ agg,
                                                         const double* __restrict__ rvec, const float* __restrict__ z3,
                                                         const double* __restrict__ sc3, const double* __restrict__ sc2,
                                                         float* __restrict__ z2, float* __restrict__ d2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Nc) return;
  const int A = agg[i];
  double t[3], w[3], wr[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    t[c] = sc3[6 * A + c] * (double)z3[6 * A + c];
    w[c] = sc3[6 * A + 3 + c] * (double)z3[6 * A + 3 + c];
  }
  cross3(w, rvec + 3 * (size_t)i, wr);  // u_i = t + w x r_i
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float corr = (float)((t[c] + wr[c]) / sc2[3 * i + c]);
    d2[3 * i + c] = corr;
    z2[3 * i + c] += corr;
  }
}
void launch_agg_prolong(hipStream_t s, int Nc, const int* agg, const double* rvec, const float* z3, const double* sc3,
                        const double* sc2, float* z2, float* d2) {
  hipLaunchKernelGGL(agg_prolong_kernel, dim3((Nc + 255) / 256), dim3(256), 0, s, Nc, agg, rvec, z3, sc3, sc2, z2, d2);
}

// the same step with the SpMV result q = H d_old already summed over ranks (multi-GPU path)
template <bool LAST>
__global__ __launch_bounds__(256) void cheb_update_kernel(int N, const double* __restrict__ Dinv,
                                                         const double* __restrict__ q,
                                                         const double* __restrict__ d_old, const double* __restrict__ coef,
                                                         double* __restrict__ d_new, double* __restrict__ z,
                                                         double* __restrict__ res, const double* __restrict__ r,
                                                         const double* __restrict__ w, const double* __restrict__ sc,
                                                         double* __restrict__ rz_part) {
  const double c1 = coef[0], c2 = coef[1];  // device-resident: the launch sequence is replayed as a hipGraph
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
      double zn = z[3 * i + c] + dn;
      d_new[3 * i + c] = dn;
      if (LAST && sc) zn *= sc[3 * i + c];  // low-precision path: back from the scaled space
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

void launch_cheb_update(hipStream_t s, int N, const double* Dinv, const double* q, const double* d_old,
                        const double* coef, double* d_new, double* z, double* res, const double* r, const double* w,
                        const double* sc, double* rz_part, bool last) {
  const dim3 g(std::max(1, std::min(kNPart, (N + 255) / 256))), b(256);
  if (last)
    hipLaunchKernelGGL((cheb_update_kernel<true>), g, b, 0, s, N, Dinv, q, d_old, coef, d_new, z, res, r, w, sc,
                       rz_part);
  else
    hipLaunchKernelGGL((cheb_update_kernel<false>), g, b, 0, s, N, Dinv, q, d_old, coef, d_new, z, res, r, w, sc,
                       rz_part);
}

// x += alpha p ; r -= alpha q ; partial r.r  (z comes from the polynomial preconditioner afterwards)
__global__ __launch_bounds__(256) void pcg_update_noz_kernel(int N, const double* __restrict__ w,
                                                            const double* __restrict__ p, const double* __restrict__ q,
                                                            const double* __restrict__ rz_part_old,
                                                            const double* __restrict__ pq_part, double* __restrict__ x,
                                                            double* __restrict__ r, double* __restrict__ rr_part,
                                                            double* __restrict__ indefinite) {
  __shared__ double sh[32];
  double rz_old, pq;
  sum_slots2(rz_part_old, pq_part, rz_old, pq, sh);
  const double alpha = pq != 0.0 ? rz_old / pq : 0.0;
  // r.z < 0 means the polynomial preconditioner is not positive definite (its interval ends below lambda_max):
  // sticky flag for the host's next convergence test
  if (blockIdx.x == 0 && threadIdx.x == 0 && rz_old < 0.0) indefinite[0] = 1.0;
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

// pcg_update_noz + cheb32_init of the NEXT iteration's polynomial in one launch (single-GPU low-precision path): the
// updated residual is scaled and block-Jacobi-preconditioned into the fp32 start vectors d, z^, res^ while it is
// still in registers.  One launch less per CG iteration (about 4 % of an iteration on small meshes).
__global__ __launch_bounds__(256) void pcg_update_init32_kernel(
    int N, const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ rz_part_old,
    const double* __restrict__ pq_part, double* __restrict__ x, double* __restrict__ r, double* __restrict__ rr_part,
    double* __restrict__ indefinite, const float* __restrict__ Dinv_f, const double* __restrict__ sc,
    const double* __restrict__ coef, float* __restrict__ d, float* __restrict__ z, float* __restrict__ res,
    const double* __restrict__ wown) {
  __shared__ double sh[32];
  double rz_old, pq;
  sum_slots2(rz_part_old, pq_part, rz_old, pq, sh);
  const double alpha = pq != 0.0 ? rz_old / pq : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && rz_old < 0.0) indefinite[0] = 1.0;
  const float inv_theta = (float)coef[0];
  const float zw0 = coef[1] != 0.0 ? (float)coef[1] : 1.f;  // fourth-kind smoothers: z^0 = beta_1 d0
  double rr = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    float rs[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      x[3 * i + c] += alpha * p[3 * i + c];
      const double rv = r[3 * i + c] - alpha * q[3 * i + c];
      r[3 * i + c] = rv;
      rr += (wown ? wown[3 * i + c] : 1.0) * rv * rv;
      rs[c] = (float)(rv * sc[3 * i + c]);
    }
    const float* D = Dinv_f + (size_t)9 * i;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float v = (D[3 * c] * rs[0] + D[3 * c + 1] * rs[1] + D[3 * c + 2] * rs[2]) * inv_theta;
      d[3 * i + c] = v;
      z[3 * i + c] = zw0 * v;
      res[3 * i + c] = rs[c];
    }
  }
  const double t = block_sum(rr, sh);
  if (threadIdx.x == 0) rr_part[blockIdx.x] = t;
  if (blockIdx.x == 0)
    for (int k = gridDim.x + threadIdx.x; k < kNPart; k += blockDim.x) rr_part[k] = 0.0;
}

void launch_pcg_update_init32(hipStream_t s, int N, const double* p, const double* q, const double* rz_part_old,
                              const double* pq_part, double* x, double* r, double* rr_part, double* indefinite,
                              const float* Dinv_f, const double* sc, const double* coef, float* d, float* z,
                              float* res, const double* wown) {
  const int n_blocks = std::max(1, std::min(kNPart, (N + 255) / 256));
  hipLaunchKernelGGL(pcg_update_init32_kernel, dim3(n_blocks), dim3(256), 0, s, N, p, q, rz_part_old, pq_part, x, r,
                     rr_part, indefinite, Dinv_f, sc, coef, d, z, res, wown);
}

// Residual replacement of the mixed-precision CG (q = H x in fp64 on H): r = b - q, its r.r slots, and the fp32 start
// vectors of the next polynomial -- what pcg_update_init32_kernel leaves, from the TRUE residual.
__global__ __launch_bounds__(256) void residual_replace_init32_kernel(
    int N, const double* __restrict__ b, const double* __restrict__ q, double* __restrict__ r,
    double* __restrict__ rr_part, const float* __restrict__ Dinv_f, const double* __restrict__ sc,
    const double* __restrict__ coef, float* __restrict__ d, float* __restrict__ z, float* __restrict__ res,
    const double* __restrict__ wown) {
  __shared__ double sh[32];
  const float inv_theta = (float)coef[0];
  const float zw0 = coef[1] != 0.0 ? (float)coef[1] : 1.f;  // fourth-kind smoothers: z^0 = beta_1 d0
  double rr = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    float rs[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double rv = b[3 * i + c] - q[3 * i + c];
      r[3 * i + c] = rv;
      rr += (wown ? wown[3 * i + c] : 1.0) * rv * rv;
      rs[c] = (float)(rv * sc[3 * i + c]);
    }
    const float* D = Dinv_f + (size_t)9 * i;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float v = (D[3 * c] * rs[0] + D[3 * c + 1] * rs[1] + D[3 * c + 2] * rs[2]) * inv_theta;
      d[3 * i + c] = v;
      z[3 * i + c] = zw0 * v;
      res[3 * i + c] = rs[c];
    }
  }
  const double t = block_sum(rr, sh);
  if (threadIdx.x == 0) rr_part[blockIdx.x] = t;
  if (blockIdx.x == 0)
    for (int k = gridDim.x + threadIdx.x; k < kNPart; k += blockDim.x) rr_part[k] = 0.0;
}

void launch_residual_replace_init32(hipStream_t s, int N, const double* b, const double* q, double* r, double* rr_part,
                                    const float* Dinv_f, const double* sc, const double* coef, float* d, float* z,
                                    float* res, const double* wown) {
  const int n_blocks = std::max(1, std::min(kNPart, (N + 255) / 256));
  hipLaunchKernelGGL(residual_replace_init32_kernel, dim3(n_blocks), dim3(256), 0, s, N, b, q, r, rr_part, Dinv_f, sc, coef,
                     d, z, res, wown);
}

void launch_pcg_update_noz(hipStream_t s, int N, const double* w, const double* p, const double* q,
                           const double* rz_part_old, const double* pq_part, double* x, double* r, double* rr_part,
                           double* indefinite) {
  const int n_blocks = std::max(1, std::min(kNPart, (3 * N + 255) / 256));
  hipLaunchKernelGGL(pcg_update_noz_kernel, dim3(n_blocks), dim3(256), 0, s, N, w, p, q, rz_part_old, pq_part, x, r,
                     rr_part, indefinite);
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
// v *= 1/sqrt(*sumsq) with the scalar on the device (power iteration without host round trips)
__global__ void scale_inv_sqrt_kernel(int n, const double* __restrict__ sumsq, double* __restrict__ v) {
  const double ss = sumsq[0];
  const double a = ss > 0.0 ? 1.0 / sqrt(ss) : 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] *= a;
}
void launch_scale_inv_sqrt(hipStream_t s, int n, const double* sumsq, double* v) {
  hipLaunchKernelGGL(scale_inv_sqrt_kernel, dim3(std::min(2048, (n + 255) / 256)), dim3(256), 0, s, n, sumsq, v);
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
__global__ void diff_kernel(int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) r[i] = a[i] - b[i];
}
void launch_diff(hipStream_t s, int n, const double* a, const double* b, double* r) {  // r = a - b
  hipLaunchKernelGGL(diff_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, a, b, r);
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
// ---- SyncedAdamWNocoop (SyncedAdamWNocoop.cu:140-189) --------------------------------------------------------
// m, v moments and the velocity update of one inner iteration; lr, 1/(1-beta1^t), 1/(1-beta2^t) from the host loop
__global__ void adamw_update_velocity_kernel(int n, const double* __restrict__ g, double beta1, double beta2,
                                             double eps, double weight_decay, double lr, double inv_1mb1t,
                                             double inv_1mb2t, double* __restrict__ m, double* __restrict__ va,
                                             double* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double gi = g[i], vi = v[i];
  const double mt = beta1 * m[i] + (1.0 - beta1) * gi;
  const double vt = beta2 * va[i] + (1.0 - beta2) * gi * gi;
  const double m_hat = mt * inv_1mb1t, v_hat = vt * inv_1mb2t;
  m[i] = mt;
  va[i] = vt;
  v[i] = vi - lr * (m_hat / (sqrt(v_hat) + eps) + weight_decay * vi);
}
void launch_adamw_update_velocity(hipStream_t s, int n, const double* g, double beta1, double beta2, double eps,
                                  double weight_decay, double lr, double inv_1mb1t, double inv_1mb2t, double* m,
                                  double* va, double* v) {
  hipLaunchKernelGGL(adamw_update_velocity_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, g, beta1, beta2, eps,
                     weight_decay, lr, inv_1mb1t, inv_1mb2t, m, va, v);
}
// ---- SyncedNesterov (SyncedNesterov.cu:95-372) --------------------------------------------------------------
// look-ahead  y = v_k + beta (v_k - v_km1)  -> v_guess
__global__ void nesterov_lookahead_kernel(int n, double beta, const double* __restrict__ vk,
                                          const double* __restrict__ vkm1, double* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = vk[i] + beta * (vk[i] - vkm1[i]);
}
void launch_nesterov_lookahead(hipStream_t s, int n, double beta, const double* vk, const double* vkm1, double* v) {
  hipLaunchKernelGGL(nesterov_lookahead_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, beta, vk, vkm1, v);
}
// v_next = y - alpha g  (y is in v_guess)
__global__ void nesterov_step_kernel(int n, double alpha, const double* __restrict__ y, const double* __restrict__ g,
                                     double* __restrict__ vnext) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) vnext[i] = y[i] - alpha * g[i];
}
void launch_nesterov_step(hipStream_t s, int n, double alpha, const double* y, const double* g, double* vnext) {
  hipLaunchKernelGGL(nesterov_step_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, alpha, y, g, vnext);
}

// x = x_prev + dt v   (adamw_update_positions_from_prev_kernel)
__global__ void positions_from_prev_kernel(int N, const double* __restrict__ v, const double* __restrict__ xp,
                                           const double* __restrict__ yp, const double* __restrict__ zp, double dt,
                                           double* __restrict__ x, double* __restrict__ y, double* __restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  x[i] = xp[i] + dt * v[3 * i];
  y[i] = yp[i] + dt * v[3 * i + 1];
  z[i] = zp[i] + dt * v[3 * i + 2];
}
void launch_positions_from_prev(hipStream_t s, int N, const double* v, const double* xp, const double* yp,
                                const double* zp, double dt, double* x, double* y, double* z) {
  hipLaunchKernelGGL(positions_from_prev_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, v, xp, yp, zp, dt, x, y, z);
}

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

// ---- overlapping partition: ghost refresh (pack owned values for a peer / scatter a peer's values into my ghosts) -------
// nvec nodal fields of `dim` values each, interleaved per node in the message: msg[(t * nvec + v) * dim + c]
template <typename T>
__global__ __launch_bounds__(256) void halo_pack_kernel(int n, const int* __restrict__ idx, int dim, int nvec,
                                                       const T* __restrict__ a, const T* __restrict__ b,
                                                       const T* __restrict__ c, T* __restrict__ msg) {
  const int per = dim * nvec;
  for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < (size_t)n * per; t += (size_t)gridDim.x * 256) {
    const int node = (int)(t / per), r = (int)(t - (size_t)node * per), v = r / dim, cc = r - v * dim;
    const T* src = v == 0 ? a : (v == 1 ? b : c);
    msg[t] = src[(size_t)dim * idx[node] + cc];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void halo_unpack_kernel(int n, const int* __restrict__ idx, int dim, int nvec,
                                                         const T* __restrict__ msg, T* __restrict__ a,
                                                         T* __restrict__ b, T* __restrict__ c) {
  const int per = dim * nvec;
  for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < (size_t)n * per; t += (size_t)gridDim.x * 256) {
    const int node = (int)(t / per), r = (int)(t - (size_t)node * per), v = r / dim, cc = r - v * dim;
    T* dst = v == 0 ? a : (v == 1 ? b : c);
    dst[(size_t)dim * idx[node] + cc] = msg[t];
  }
}
static int halo_grid(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(2048, (n + 255) / 256)); }
void launch_halo_pack_f64(hipStream_t s, int n, const int* idx, int dim, int nvec, const double* a, const double* b,
                          const double* c, double* msg) {
  if (n > 0) hipLaunchKernelGGL(halo_pack_kernel<double>, dim3(halo_grid((size_t)n * dim * nvec)), dim3(256), 0, s, n, idx, dim, nvec, a, b, c, msg);
}
void launch_halo_unpack_f64(hipStream_t s, int n, const int* idx, int dim, int nvec, const double* msg, double* a, double* b,
                            double* c) {
  if (n > 0) hipLaunchKernelGGL(halo_unpack_kernel<double>, dim3(halo_grid((size_t)n * dim * nvec)), dim3(256), 0, s, n, idx, dim, nvec, msg, a, b, c);
}
void launch_halo_pack_f32(hipStream_t s, int n, const int* idx, int dim, int nvec, const float* a, const float* b,
                          const float* c, float* msg) {
  if (n > 0) hipLaunchKernelGGL(halo_pack_kernel<float>, dim3(halo_grid((size_t)n * dim * nvec)), dim3(256), 0, s, n, idx, dim, nvec, a, b, c, msg);
}
void launch_halo_unpack_f32(hipStream_t s, int n, const int* idx, int dim, int nvec, const float* msg, float* a, float* b,
                            float* c) {
  if (n > 0) hipLaunchKernelGGL(halo_unpack_kernel<float>, dim3(halo_grid((size_t)n * dim * nvec)), dim3(256), 0, s, n, idx, dim, nvec, msg, a, b, c);
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
