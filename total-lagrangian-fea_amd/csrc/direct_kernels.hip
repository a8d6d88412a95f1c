// direct_kernels.hip -- the engine's own sparse direct solve on gfx950: multifrontal Cholesky of H on the plan of
// mf_host.h.  What it replaces: cuDSS REFACTORIZATION + SOLVE in the reference's Newton loop (SyncedNewton.cu:1103-1114).
//
// A level of the dissection tree is a batch of independent dense fronts (column-major, lower triangle used) that live in
// HBM / L2 (a T10 leaf front is ~400 DOF = 1.3 MB: no front fits the LDS).  Per level:
//   zero the level's workspace -> scatter H's lower node blocks -> extend-add child 0, then child 1 (fixed order) ->
//   panel steps of kMfNB = 48 columns over the fronts that still have columns, grouped in SUPER-PANELS of 8:
//     panel       : every workgroup re-factors the 48x48 diagonal block (one wavefront, a row per lane in registers,
//                   the other rows by v_readlane: redundant per workgroup, cheaper than a launch that would do it once) and
//                   solves its 256 rows of the panel against it, a row per lane in 48 registers (L21 = F21 L11^-T); the
//                   panel goes to the factor's storage
//     update      : inside the super-panel only its remaining columns are updated (64x64 tiles, 4x4 per lane)
//     wide update : a completed super-panel (K up to 384) out of everything to its right, 128x128 tiles, 8x8 per lane, K
//                   staged through LDS in chunks of 16 -- the trailing matrix is read and written once per 384 columns
//                   (64-tiles where 128-tiles would not fill the chip).  fp64 vector FMA: the MI355X's fp64 matrix rate
//                   equals its vector rate, so MFMA would buy nothing here
//   what remains below/right of the own columns is the update matrix the parent adds into its front.
// Solve: forward by levels (front vector = right-hand side of the own DOFs + the children's remainders, through the same
// maps), backward from the root; levels of small fronts one workgroup per front, levels with a large front panel step by
// panel step over many workgroups (the 48x48 triangle from LDS; backward in outer-product form: no cross-workgroup sums).
// No atomics anywhere: same H, same bits.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "tlfea_internal.h"

namespace tlfea {

namespace {
constexpr int NB = kMfNB;

__device__ __forceinline__ void wave_sync_d() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double readlane_d(double v, int lane) {  // lane: compile-time constant after unrolling
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 3x3 node blocks of H (row-major inside a node row, stride sld) -> the front (column-major, stride dld)
__global__ void mf_scatter_h_kernel(int n, const long long* __restrict__ src, const int* __restrict__ sld,
                                    const long long* __restrict__ dst, const int* __restrict__ dld,
                                    const double* __restrict__ H, double* __restrict__ W) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 9LL * n) return;
  const int ent = (int)(t / 9), de = (int)(t % 9), d = de / 3, e = de % 3;
  W[dst[ent] + d + (long long)e * dld[ent]] = H[src[ent] + (long long)d * sld[ent] + e];
}

// parent front += the child's update matrix (rows/columns below its own), node block by node block
__global__ __launch_bounds__(256) void mf_extend_add_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl,
                                                           int slot, double* __restrict__ Wp,
                                                           const double* __restrict__ Wc, const int* __restrict__ map) {
  const MfFrontDev f = fr[lvl[blockIdx.z]];
  const int c = slot ? f.child1 : f.child0;
  if (c < 0) return;
  const MfFrontDev ch = fr[c];
  const int mu = (ch.m - ch.k) / 3;
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= mu || j > i) return;
  const int* mp = map + ch.map_off;
  const double* U = Wc + ch.cF_off + (ch.c_k0 + 3 * i) + (long long)(ch.c_k0 + 3 * j) * ch.c_ld;
  double* A = Wp + f.F_off + 3LL * mp[i] + 3LL * mp[j] * f.m;
#pragma unroll
  for (int e = 0; e < 3; e++)
#pragma unroll
    for (int d = 0; d < 3; d++)
      if (i > j || d >= e) A[d + (long long)e * f.m] += U[d + (long long)e * ch.c_ld];
}

// a front factored in the WORK buffer: its update matrix (lower triangle) compacted onto the stack for the parent
__global__ __launch_bounds__(256) void mf_push_kernel(const MfFrontDev* __restrict__ fr, int front,
                                                     const double* __restrict__ Wk, double* __restrict__ St) {
  const MfFrontDev f = fr[front];
  const int mu = f.m - f.k;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), j0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 16;
  if (i >= mu) return;
  for (int j = j0; j < min(j0 + 16, mu); j++)
    if (j <= i) St[f.cF_off + i + (long long)j * mu] = Wk[f.F_off + (f.k + i) + (long long)(f.k + j) * f.m];
}

__global__ __launch_bounds__(256) void mf_panel_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl, int j0,
                                                      const double* __restrict__ W, double* __restrict__ Ls,
                                                      int* __restrict__ err) {
  const MfFrontDev f = fr[lvl[blockIdx.y]];
  const int m = f.m, k = f.k;
  if (k <= j0) return;
  const int jb = min(NB, k - j0);
  const int r0 = j0 + jb + blockIdx.x * 256;
  if (blockIdx.x > 0 && r0 >= m) return;
  __shared__ double A[NB][NB + 1];
  __shared__ double dinv[NB];
  const double* F = W + f.F_off;
  double* L = Ls + f.L_off;
  const int tid = threadIdx.x;
  // the diagonal block, padded with the identity to NB columns
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx % NB, c = idx / NB;
    A[r][c] = (r < jb && c < jb && r >= c) ? F[(long long)(j0 + r) + (long long)(j0 + c) * m] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  if (tid < 64) {
    // one wavefront, lane r holds row r in registers: right-looking Cholesky, the column's entries of the other rows come
    // by v_readlane (compile-time lanes) -- no LDS round trip inside the 48-step dependent chain
    const int r = tid, rl = min(r, NB - 1);
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) a[c] = A[rl][c];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < NB; c++) {
      double d = readlane_d(a[c], c);
      if (!(d > 0.0)) {
        bad = true;  // not positive definite: the caller reports it; keep the arithmetic finite
        d = 1.0;
      }
      const double inv = 1.0 / sqrt(d);
      const double l = a[c] * inv;  // (the pivot lane gets d / sqrt(d): the diagonal entry, to the last bit or one off)
      a[c] = l;
#pragma unroll
      for (int q = c + 1; q < NB; q++) a[q] -= l * readlane_d(l, q);  // (entries right of the diagonal: unused)
      __builtin_amdgcn_sched_barrier(0);  // keeps one column's broadcasts (2 SGPRs each) in flight, not all 1128
    }
    if (bad && r == 0) *err = 1;
    if (r < NB) {
#pragma unroll
      for (int c = 0; c < NB; c++) A[r][c] = a[c];  // (right of the diagonal: never read)
      dinv[r] = 1.0 / A[r][r];
    }
  }
  __syncthreads();
  if (blockIdx.x == 0)
    for (int idx = tid; idx < NB * NB; idx += 256) {
      const int r = idx % NB, c = idx / NB;
      if (r < jb && c <= r) L[(long long)(j0 + r) + (long long)(j0 + c) * m] = A[r][c];
    }
  const int rr = r0 + tid;
  if (rr >= m) return;
  double X[NB];
  {
    const double* Fp = F + (long long)rr + (long long)j0 * m;
#pragma unroll
    for (int c = 0; c < NB; c++) {
      X[c] = c < jb ? *Fp : 0.0;
      if (c + 1 < jb) Fp += m;
    }
  }
#pragma unroll
  for (int c = 0; c < NB; c++) {
    double s = X[c];
    const double* Ac = A[c];
#pragma unroll
    for (int p = 0; p < c; p++) s -= X[p] * Ac[p];
    X[c] = s * dinv[c];
    asm volatile("" ::: "memory");  // row c's broadcast LDS reads are issued together, but not hoisted above earlier rows
  }
  // unconditional stores, last column first: the padding columns (X = 0) land on column jb - 1 BEFORE its own value does
  // (a store per column under `if (c < jb)` makes the compiler spill ~1900 VGPRs)
  double* Lp = L + (long long)rr + (long long)j0 * m;
#pragma unroll
  for (int c = NB - 1; c >= 0; c--) Lp[(long long)min(c, jb - 1) * m] = X[c];
}

// F[r, c] -= sum over panel columns p in [cb, ce) of L[r, p] L[c, p], 64x64 tiles of the rows/columns from `ce` on:
//   wide = 0, inside a super-panel [S0, s1): the panel at j0 is taken out of the super-panel's remaining columns only
//   wide = 1, a completed super-panel [j0, min(s1, k)) out of everything to its right -- the 64-tile form of
//             mf_update_wide_kernel, for trailing matrices of too few 128-tiles to fill the chip
__global__ __launch_bounds__(256) void mf_update_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl, int j0,
                                                       int s1, int wide, double* __restrict__ W,
                                                       const double* __restrict__ Ls) {
  const MfFrontDev f = fr[lvl[blockIdx.z]];
  const int m = f.m, k = f.k;
  if (k <= j0) return;
  const int cb = j0, ce = wide ? min(s1, k) : j0 + min(NB, k - j0), base = ce, ccap = wide ? m : min(s1, k);
  const int I = blockIdx.x, J = blockIdx.y;
  if (J > I || base + I * 64 >= m || base + J * 64 >= ccap) return;
  __shared__ double Pi[NB][64], Pj[NB][64];
  const double* L = Ls + f.L_off;
  double* F = W + f.F_off;
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  double acc[4][4] = {};
  for (int c0 = cb; c0 < ce; c0 += NB) {
    const int jb = min(NB, ce - c0);
    if (c0 > cb) __syncthreads();
    for (int idx = tid; idx < NB * 64; idx += 256) {
      const int c = idx >> 6, r = idx & 63;
      const int gi = base + I * 64 + r, gj = base + J * 64 + r;
      Pi[c][r] = (c < jb && gi < m) ? L[(long long)gi + (long long)(c0 + c) * m] : 0.0;
      Pj[c][r] = (c < jb && gj < m) ? L[(long long)gj + (long long)(c0 + c) * m] : 0.0;
    }
    __syncthreads();
    for (int c = 0; c < jb; c++) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        a[i] = Pi[c][4 * tx + i];
        b[i] = Pj[c][4 * ty + i];
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] += a[i] * b[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int gj = base + J * 64 + 4 * ty + j;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int gi = base + I * 64 + 4 * tx + i;
      if (gi < m && gj <= gi && gj < ccap) F[(long long)gi + (long long)gj * m] -= acc[i][j];
    }
  }
}

// after a super-panel [S0, S1 = min(S0 + KS, k)): everything to the right of it, F[r, c] -= L[r, S0:S1] L[c, S0:S1]^T for
// r >= c >= S1, in 128x128 tiles, 8x8 per lane, the K range staged through LDS in chunks of 16 columns -- the trailing
// matrix is read and written once per super-panel instead of once per panel
constexpr int WT = 128, KC = 16;
__global__ __launch_bounds__(256) void mf_update_wide_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl,
                                                            int S0, int KS, double* __restrict__ W,
                                                            const double* __restrict__ Ls) {
  const MfFrontDev f = fr[lvl[blockIdx.z]];
  const int m = f.m, k = f.k;
  if (k <= S0) return;
  const int S1 = min(S0 + KS, k), base = S1;
  const int I = blockIdx.x, J = blockIdx.y;
  if (J > I || base + I * WT >= m) return;
  __shared__ double Pi[KC][WT], Pj[KC][WT];
  const double* L = Ls + f.L_off;
  double* F = W + f.F_off;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int lr = tid & 127, lc = tid >> 7;  // loader: row of the tile, first of its 8 chunk columns (stride 2)
  const int gi = base + I * WT + lr, gj = base + J * WT + lr;
  double acc[8][8] = {};
  for (int c0 = S0; c0 < S1; c0 += KC) {
#pragma unroll
    for (int q = 0; q < KC / 2; q++) {
      const int c = lc + 2 * q;
      const bool in = c0 + c < S1;
      Pi[c][lr] = (in && gi < m) ? L[(long long)gi + (long long)(c0 + c) * m] : 0.0;
      Pj[c][lr] = (in && gj < m) ? L[(long long)gj + (long long)(c0 + c) * m] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int c = 0; c < KC; c++) {
      double a[8], b[8];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        a[i] = Pi[c][4 * tx + i];
        a[4 + i] = Pi[c][64 + 4 * tx + i];
        b[i] = Pj[c][4 * ty + i];
        b[4 + i] = Pj[c][64 + 4 * ty + i];
      }
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int cj = base + J * WT + (j < 4 ? 4 * ty + j : 64 + 4 * ty + j - 4);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int ri = base + I * WT + (i < 4 ? 4 * tx + i : 64 + 4 * tx + i - 4);
      if (ri < m && cj <= ri) F[(long long)ri + (long long)cj * m] -= acc[i][j];
    }
  }
}

__global__ __launch_bounds__(512) void mf_forward_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl,
                                                        const double* __restrict__ Ls, const int* __restrict__ map,
                                                        const int* __restrict__ order, const double* __restrict__ b,
                                                        double* __restrict__ v, double* __restrict__ y) {
  const MfFrontDev f = fr[lvl[blockIdx.x]];
  const int m = f.m, k = f.k, tid = threadIdx.x, nt = blockDim.x;
  double* w = v + f.v_off;
  const double* L = Ls + f.L_off;
  for (int i = tid; i < m; i += nt) w[i] = i < k ? b[3 * order[f.c0 + i / 3] + i % 3] : 0.0;
  __syncthreads();
  for (int s = 0; s < 2; s++) {
    const int c = s ? f.child1 : f.child0;
    if (c < 0) continue;
    const MfFrontDev ch = fr[c];
    const double* vc = v + ch.v_off + ch.k;
    const int* mp = map + ch.map_off;
    for (int i = tid; i < ch.m - ch.k; i += nt) w[3 * mp[i / 3] + i % 3] += vc[i];
    __syncthreads();
  }
  __shared__ double ys[NB];
  __shared__ double T[NB][NB + 1];  // the block's triangle, staged by all lanes: read entry by entry from HBM / L2 inside
                                    // the substitution's dependent chain it cost ~100 us per block
  for (int j0 = 0; j0 < k; j0 += NB) {
    const int jb = min(NB, k - j0);
    for (int idx = tid; idx < NB * NB; idx += nt) {
      const int r = idx % NB, c = idx / NB;
      if (r < jb && c <= r) T[r][c] = L[(long long)(j0 + r) + (long long)(j0 + c) * m];
    }
    __syncthreads();
    if (tid < 64) {  // the block's triangle: one wavefront, lane r owns row r
      const int r = tid;
      double val = r < jb ? w[j0 + r] : 0.0;
      for (int c = 0; c < jb; c++) {
        const double yc = __shfl(val, c) / T[c][c];
        if (r == c) val = yc;
        else if (r > c && r < jb) val -= T[r][c] * yc;
      }
      if (r < jb) {
        ys[r] = val;
        w[j0 + r] = val;
      }
    }
    __syncthreads();
    for (int r = j0 + jb + tid; r < m; r += nt) {
      double s = 0.0;
      for (int c = 0; c < jb; c++) s += L[(long long)r + (long long)(j0 + c) * m] * ys[c];
      w[r] -= s;
    }
    __syncthreads();
  }
  for (int i = tid; i < k; i += nt) y[3LL * f.c0 + i] = w[i];
}

__global__ __launch_bounds__(512) void mf_backward_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl,
                                                         const double* __restrict__ Ls, const int* __restrict__ rows,
                                                         const double* __restrict__ y, double* __restrict__ xp,
                                                         double* __restrict__ v) {
  const MfFrontDev f = fr[lvl[blockIdx.x]];
  const int m = f.m, k = f.k, tid = threadIdx.x, nt = blockDim.x;
  double* w = v + f.v_off;
  const double* L = Ls + f.L_off;
  for (int i = tid; i < m; i += nt) w[i] = i < k ? y[3LL * f.c0 + i] : xp[3LL * rows[f.rows_off + i / 3] + i % 3];
  __syncthreads();
  __shared__ double ys[NB];
  __shared__ double T[NB][NB + 1];
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  for (int j0 = ((k - 1) / NB) * NB; j0 >= 0 && k > 0; j0 -= NB) {
    const int jb = min(NB, k - j0);
    for (int c = wave; c < jb; c += nw) {  // column c against everything already solved below the block
      const double* Lc = L + (long long)(j0 + c) * m;
      double s = 0.0;
      for (int r = j0 + jb + lane; r < m; r += 64) s += Lc[r] * w[r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
      if (lane == 0) ys[c] = w[j0 + c] - s;
    }
    for (int idx = tid; idx < NB * NB; idx += nt) {
      const int r = idx % NB, c = idx / NB;
      if (r < jb && c <= r) T[r][c] = L[(long long)(j0 + r) + (long long)(j0 + c) * m];
    }
    __syncthreads();
    if (tid < 64) {
      const int r = tid;
      double val = r < jb ? ys[r] : 0.0;
      for (int c = jb - 1; c >= 0; c--) {
        const double xc = __shfl(val, c) / T[c][c];
        if (r == c) val = xc;
        else if (r < c) val -= T[c][r] * xc;
      }
      if (r < jb) w[j0 + r] = val;
    }
    __syncthreads();
  }
  for (int i = tid; i < k; i += nt) xp[3LL * f.c0 + i] = w[i];
}

// ---- the same solves, panel step by panel step, for levels whose fronts are too large for one workgroup each ----------
// w = [right-hand side of the own DOFs; 0], then += the children's remainders (one launch per child slot: fixed order)
__global__ __launch_bounds__(256) void mf_fwd_init_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl,
                                                         const int* __restrict__ order, const double* __restrict__ b,
                                                         double* __restrict__ v) {
  const MfFrontDev f = fr[lvl[blockIdx.y]];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= f.m) return;
  v[f.v_off + i] = i < f.k ? b[3 * order[f.c0 + i / 3] + i % 3] : 0.0;
}
__global__ __launch_bounds__(256) void mf_fwd_child_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl,
                                                          int slot, const int* __restrict__ map, double* __restrict__ v) {
  const MfFrontDev f = fr[lvl[blockIdx.y]];
  const int c = slot ? f.child1 : f.child0;
  if (c < 0) return;
  const MfFrontDev ch = fr[c];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ch.m - ch.k) return;
  v[f.v_off + 3 * map[ch.map_off + i / 3] + i % 3] += v[ch.v_off + ch.k + i];
}
// block j0: every workgroup solves the 48x48 triangle (one wavefront), then takes the block out of its 256 rows below
__global__ __launch_bounds__(256) void mf_fwd_step_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl, int j0,
                                                         const double* __restrict__ Ls, double* __restrict__ v,
                                                         double* __restrict__ y) {
  const MfFrontDev f = fr[lvl[blockIdx.y]];
  const int m = f.m, k = f.k, tid = threadIdx.x;
  if (k <= j0) return;
  const int jb = min(NB, k - j0);
  const int r0 = j0 + jb + blockIdx.x * 256;
  if (blockIdx.x > 0 && r0 >= m) return;
  double* w = v + f.v_off;
  const double* L = Ls + f.L_off;
  __shared__ double ys[NB];
  __shared__ double T[NB][NB + 1];  // the block's triangle, staged by all lanes (read one entry at a time from HBM/L2 inside
                                    // the substitution's dependent chain it cost ~40 us per step)
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx % NB, c = idx / NB;
    if (r < jb && c <= r) T[r][c] = L[(long long)(j0 + r) + (long long)(j0 + c) * m];
  }
  __syncthreads();
  if (tid < 64) {
    const int r = tid;
    double val = r < jb ? w[j0 + r] : 0.0;
    for (int c = 0; c < jb; c++) {
      const double yc = __shfl(val, c) / T[c][c];
      if (r == c) val = yc;
      else if (r > c && r < jb) val -= T[r][c] * yc;
    }
    if (r < jb) {
      ys[r] = val;
      if (blockIdx.x == 0) y[3LL * f.c0 + j0 + r] = val;
    }
  }
  __syncthreads();
  const int rr = r0 + tid;
  if (rr >= m) return;
  const double* Lr = L + (long long)rr + (long long)j0 * m;
  double s = 0.0;
  for (int c = 0; c < jb; c++) s += Lr[(long long)c * m] * ys[c];
  w[rr] -= s;
}
// backward: w = [y of the own DOFs; x of the ancestors' rows], then w1 -= L21^T w2 (a wavefront per column)
__global__ __launch_bounds__(256) void mf_bwd_gather_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl,
                                                           const int* __restrict__ rows, const double* __restrict__ y,
                                                           const double* __restrict__ xp, double* __restrict__ v) {
  const MfFrontDev f = fr[lvl[blockIdx.y]];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= f.m) return;
  v[f.v_off + i] = i < f.k ? y[3LL * f.c0 + i] : xp[3LL * rows[f.rows_off + i / 3] + i % 3];
}
__global__ __launch_bounds__(256) void mf_bwd_below_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl,
                                                          const double* __restrict__ Ls, double* __restrict__ v) {
  const MfFrontDev f = fr[lvl[blockIdx.y]];
  const int m = f.m, k = f.k;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= k || m == k) return;
  double* w = v + f.v_off;
  const double* Lc = Ls + f.L_off + (long long)c * m;
  double s = 0.0;
  for (int r = k + lane; r < m; r += 64) s += Lc[r] * w[r];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (lane == 0) w[c] -= s;
}
// block j0 (descending): every workgroup solves the transposed triangle, then takes the block out of its 256 columns
// to the left (a lane per column: the block's 48 entries of a column are contiguous)
__global__ __launch_bounds__(256) void mf_bwd_step_kernel(const MfFrontDev* __restrict__ fr, const int* __restrict__ lvl, int j0,
                                                         const double* __restrict__ Ls, double* __restrict__ v,
                                                         double* __restrict__ xp) {
  const MfFrontDev f = fr[lvl[blockIdx.y]];
  const int m = f.m, k = f.k, tid = threadIdx.x;
  if (k <= j0) return;
  const int jb = min(NB, k - j0);
  const int cb = blockIdx.x * 256;
  if (blockIdx.x > 0 && cb >= j0) return;
  double* w = v + f.v_off;
  const double* L = Ls + f.L_off;
  __shared__ double xs[NB];
  __shared__ double T[NB][NB + 1];
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx % NB, c = idx / NB;
    if (r < jb && c <= r) T[r][c] = L[(long long)(j0 + r) + (long long)(j0 + c) * m];
  }
  __syncthreads();
  if (tid < 64) {
    const int r = tid;
    double val = r < jb ? w[j0 + r] : 0.0;
    for (int c = jb - 1; c >= 0; c--) {
      const double xc = __shfl(val, c) / T[c][c];
      if (r == c) val = xc;
      else if (r < c) val -= T[c][r] * xc;
    }
    if (r < jb) {
      xs[r] = val;
      if (blockIdx.x == 0) xp[3LL * f.c0 + j0 + r] = val;
    }
  }
  __syncthreads();
  const int j = cb + tid;
  if (j >= j0) return;
  const double* Lc = L + (long long)j * m + j0;
  double s = 0.0;
  for (int r = 0; r < jb; r++) s += Lc[r] * xs[r];
  w[j] -= s;
}

__global__ void mf_unpermute_kernel(int N, const int* __restrict__ order, const double* __restrict__ xp,
                                    double* __restrict__ x) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const int o = order[p];
  x[3 * o] = xp[3 * p];
  x[3 * o + 1] = xp[3 * p + 1];
  x[3 * o + 2] = xp[3 * p + 2];
}
}  // namespace

// numeric factorisation of the H whose values are at `H` (the engine's layout); err is set when a pivot is not positive
void launch_mf_factor(hipStream_t s, const MfPlan& P, const MfDev& D, const double* H) {
  // super-panel: the trailing matrix is updated once per KS columns (a multiple of the panel width)
  static const int KS = NB * std::max(1, std::getenv("TLFEA_DIRECT_SUPER") ? std::atoi(std::getenv("TLFEA_DIRECT_SUPER")) : 8);
  (void)hipMemsetAsync(D.err, 0, sizeof(int), s);
  for (const MfBatch& B : P.batches) {  // level batches, then (large problems) the top fronts one by one (mf_host.h)
    double* W = D.F[B.wbuf];
    const double* Wc = D.F[B.cbuf];
    const int* lvl = D.blvl + B.first;
    const int nfl = B.count;
    (void)hipMemsetAsync(W + B.F_base, 0, (size_t)B.F_doubles * sizeof(double), s);
    if (B.hent_count > 0) {
      const long long th = 9LL * B.hent_count;
      hipLaunchKernelGGL(mf_scatter_h_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, B.hent_count,
                         D.hsrc + B.hent_off, D.hsld + B.hent_off, D.hdst + B.hent_off, D.hdld + B.hent_off, H, W);
    }
    for (int slot = 0; slot < 2; slot++) {
      int mu = 0;
      for (int t = 0; t < nfl; t++) {
        const MfFront& F = P.fronts[P.batch_fronts[(size_t)B.first + t]];
        if (F.child[slot] >= 0) {
          const MfFront& C = P.fronts[F.child[slot]];
          mu = std::max(mu, C.nrows - (C.c1 - C.c0));
        }
      }
      if (mu <= 0) continue;
      const unsigned g = (unsigned)((mu + 15) / 16);
      for (int z0 = 0; z0 < nfl; z0 += 32768)
        hipLaunchKernelGGL(mf_extend_add_kernel, dim3(g, g, (unsigned)std::min(32768, nfl - z0)), dim3(256), 0, s, D.fr,
                           lvl + z0, slot, W, Wc, D.map);
    }
    for (int t = B.step_off; t < B.step_off + B.step_count; t++) {
      const MfLevelStep& st = P.bsteps[t];
      const int S0 = (st.j0 / KS) * KS, s1 = S0 + KS;
      const unsigned rt = (unsigned)std::max(1, (st.max_below + 255) / 256), ut = (unsigned)((st.max_below + 63) / 64);
      const unsigned ct = (unsigned)((s1 - st.j0 - NB + 63) / 64 + 1);  // column tiles left in the super-panel
      for (int z0 = 0; z0 < st.n_active; z0 += 32768) {
        const unsigned nz = (unsigned)std::min(32768, st.n_active - z0);
        hipLaunchKernelGGL(mf_panel_kernel, dim3(rt, nz), dim3(256), 0, s, D.fr, lvl + z0, st.j0, W, D.L, D.err);
        if (ut > 0 && st.j0 + NB < s1)
          hipLaunchKernelGGL(mf_update_kernel, dim3(ut, std::min(ut, ct), nz), dim3(256), 0, s, D.fr, lvl + z0, st.j0, s1, 0,
                             W, D.L);
      }
      if (st.j0 + NB >= s1 || t + 1 == B.step_off + B.step_count) {  // the super-panel is complete: take it out of the rest
        const MfLevelStep& s0 = P.bsteps[B.step_off + S0 / NB];
        int below = 0;
        for (int q = 0; q < s0.n_active; q++) {
          const MfFront& F = P.fronts[P.batch_fronts[(size_t)B.first + q]];
          below = std::max(below, 3 * F.nrows - std::min(s1, 3 * (F.c1 - F.c0)));
        }
        const unsigned wt = (unsigned)((below + WT - 1) / WT), wt64 = (unsigned)((below + 63) / 64);
        // 128-tiles when they fill the chip (the lower triangle of the largest front x the fronts), 64-tiles below that
        const bool big = (double)wt * (wt + 1) / 2 * s0.n_active >= 2.0 * 256;
        if (wt > 0)
          for (int z0 = 0; z0 < s0.n_active; z0 += 32768) {
            const unsigned nz = (unsigned)std::min(32768, s0.n_active - z0);
            if (big)
              hipLaunchKernelGGL(mf_update_wide_kernel, dim3(wt, wt, nz), dim3(256), 0, s, D.fr, lvl + z0, S0, KS, W, D.L);
            else
              hipLaunchKernelGGL(mf_update_kernel, dim3(wt64, wt64, nz), dim3(256), 0, s, D.fr, lvl + z0, S0, s1, 1, W, D.L);
          }
      }
    }
    if (B.push) {
      const MfFront& F = P.fronts[P.batch_fronts[(size_t)B.first]];
      const int mu = 3 * (F.nrows - (F.c1 - F.c0));
      if (mu > 0)
        hipLaunchKernelGGL(mf_push_kernel, dim3((unsigned)((mu + 63) / 64), (unsigned)((mu + 63) / 64)), dim3(256), 0, s, D.fr,
                           P.batch_fronts[(size_t)B.first], W, D.F[3]);
    }
  }
}

// x = H^-1 b with the factor of the last launch_mf_factor (b, x in the engine's DOF order; x may alias b)
void launch_mf_solve(hipStream_t s, const MfPlan& P, const MfDev& D, const double* b, double* x) {
  const int nl = P.n_levels();
  // a level whose largest front has at most this many DOF rows runs one workgroup per front; above, step by step
  static const int one_wg_rows = std::getenv("TLFEA_DIRECT_ONE_WG_ROWS") ? std::atoi(std::getenv("TLFEA_DIRECT_ONE_WG_ROWS")) : 768;
  std::vector<int> m_max((size_t)nl, 0), mu_max((size_t)nl, 0), k_max((size_t)nl, 0);
  for (int l = 0; l < nl; l++)
    for (int t = P.level_off[l]; t < P.level_off[l + 1]; t++) {
      const MfFront& F = P.fronts[P.level_fronts[t]];
      m_max[l] = std::max(m_max[l], 3 * F.nrows);
      k_max[l] = std::max(k_max[l], 3 * (F.c1 - F.c0));
      for (int c : F.child)
        if (c >= 0) mu_max[l] = std::max(mu_max[l], 3 * (P.fronts[c].nrows - (P.fronts[c].c1 - P.fronts[c].c0)));
    }
  for (int l = 0; l < nl; l++) {
    const unsigned nfl = (unsigned)(P.level_off[l + 1] - P.level_off[l]);
    const int* lvl = D.lvl + P.level_off[l];
    if (m_max[l] <= one_wg_rows) {
      hipLaunchKernelGGL(mf_forward_kernel, dim3(nfl), dim3(512), 0, s, D.fr, lvl, D.L, D.map, D.order, b, D.v, D.y);
      continue;
    }
    hipLaunchKernelGGL(mf_fwd_init_kernel, dim3((unsigned)((m_max[l] + 255) / 256), nfl), dim3(256), 0, s, D.fr, lvl, D.order, b,
                       D.v);
    if (mu_max[l] > 0)
      for (int slot = 0; slot < 2; slot++)
        hipLaunchKernelGGL(mf_fwd_child_kernel, dim3((unsigned)((mu_max[l] + 255) / 256), nfl), dim3(256), 0, s, D.fr, lvl, slot,
                           D.map, D.v);
    for (int t = P.step_off[l]; t < P.step_off[l + 1]; t++) {
      const MfLevelStep& st = P.steps[t];
      hipLaunchKernelGGL(mf_fwd_step_kernel, dim3((unsigned)std::max(1, (st.max_below + 255) / 256), (unsigned)st.n_active),
                         dim3(256), 0, s, D.fr, lvl, st.j0, D.L, D.v, D.y);
    }
  }
  for (int l = nl - 1; l >= 0; l--) {
    const unsigned nfl = (unsigned)(P.level_off[l + 1] - P.level_off[l]);
    const int* lvl = D.lvl + P.level_off[l];
    if (m_max[l] <= one_wg_rows) {
      hipLaunchKernelGGL(mf_backward_kernel, dim3(nfl), dim3(512), 0, s, D.fr, lvl, D.L, D.rows, D.y, D.xp, D.v);
      continue;
    }
    hipLaunchKernelGGL(mf_bwd_gather_kernel, dim3((unsigned)((m_max[l] + 255) / 256), nfl), dim3(256), 0, s, D.fr, lvl, D.rows,
                       D.y, D.xp, D.v);
    hipLaunchKernelGGL(mf_bwd_below_kernel, dim3((unsigned)((k_max[l] + 3) / 4), nfl), dim3(256), 0, s, D.fr, lvl, D.L, D.v);
    for (int t = P.step_off[l + 1] - 1; t >= P.step_off[l]; t--) {
      const MfLevelStep& st = P.steps[t];
      hipLaunchKernelGGL(mf_bwd_step_kernel, dim3((unsigned)std::max(1, (st.j0 + 255) / 256), (unsigned)st.n_active), dim3(256),
                         0, s, D.fr, lvl, st.j0, D.L, D.v, D.xp);
    }
  }
  hipLaunchKernelGGL(mf_unpermute_kernel, dim3((unsigned)((P.N + 255) / 256)), dim3(256), 0, s, P.N, D.order, D.xp, x);
}

}  // namespace tlfea
