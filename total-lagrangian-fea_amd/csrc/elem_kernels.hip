// elem_kernels.hip -- hand-written gfx950 kernels of the element hot path, templated on the element type
// <S shape functions, Q quadrature points>: T10 tetrahedron <10,5>, ANCF-3243 beam <8,12>, ANCF-3443 shell <16,48>
// (FEAT10DataFunc.cuh, ANCF3243DataFunc.cuh, ANCF3443DataFunc.cuh share one formulation: F = sum_a x_a (x) grad N_a).
//
// What the reference does with four kernels, HBM round trips of F/P and ~4500 double atomics +
// binary searches per element (FEAT10DataFunc.cuh:85-293,397-458,513-791) is done here as
//
//   residual : residual_kernel<S,Q>   thread per element (coalesced element-fastest grad-N copy):
//                                     F -> P (SVK | Mooney-Rivlin, + Kelvin-Voigt) -> 3S nodal force
//                                     components; no F/P round trip, no atomics.  T10 on the solver path also
//                                     leaves the per-point F records the fused assembly stages and the element's
//                                     inertia rows M_e (v - v_prev)/h, both through wave-private LDS transposes
//              grad_light_kernel      8 lanes per node (T10): gathers the node's (force | inertia) records, adds
//              / grad_kernel          - f_ext + h J^T(lam + rho c)   (SyncedNewton.cu:344-407); grad_kernel is the
//                                     mass-CSR form of the ANCF kinds
//   tangent + assembly, T10 + St.Venant-Kirchhoff: ONE launch, no element-block buffer
//              assemble_affine_kernel straight-sided elements (the default where it applies): 4 lanes per
//                                     (row, element) instance evaluate the blocks from the four vertex gradients,
//                                     rows accumulate in LDS in H's own layout and stream out once
//              assemble_direct_kernel curved elements: one wavefront per group of node rows, lane per
//                                     (instance, column node), stored grad N
//   tangent + assembly, Mooney-Rivlin and the ANCF kinds: two launches through the element-block buffer
//              tangent_blocks_kernel  one element per wavefront, x / grad-N / F / F*h (or the tangent tensor)
//                                     staged in LDS, a lane owns node-pair block(s) (i<=j) and sums the points
//                                     in registers (h*K + C_vis fused)
//              assemble_rows_kernel   one node row per wavefront: sums the contributions of the row's elements
//                                     in LDS in a fixed order, adds M/h and h^2*rho on pinned rows, streams the
//                                     3 CSR rows out once (no memset, no atomics, bitwise reproducible).
//
// Math follows SVK.cuh:14-55, MooneyRivlin.cuh:17-225, FEAT10Data.cu:97-278 (cited inline).
#include "tlfea_internal.h"
#include "pair_runs.h"

#include <cstdlib>

namespace tlfea {

// ------------------------------------------------------------------------------------------------
// small 3x3 helpers (registers only)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double det3(const double A[3][3]) {
  return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
         A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
}

// inverse-transpose with the reference's determinant clamp (MooneyRivlin.cuh:25-43)
__device__ __forceinline__ void inv_transpose3(const double A[3][3], double detA, double G[3][3]) {
  const double eps = 1e-12;
  double sd = detA;
  if (fabs(sd) < eps) sd = (sd >= 0.0) ? eps : -eps;
  const double id = 1.0 / sd;
  G[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) * id;
  G[0][1] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) * id;
  G[0][2] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) * id;
  G[1][0] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id;
  G[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id;
  G[1][2] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id;
  G[2][0] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id;
  G[2][1] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id;
  G[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
}

// Invariants and helper matrices of compressible Mooney-Rivlin (MooneyRivlin.cuh:48-95).
struct MRState {
  double C[3][3], FC[3][3], FFT[3][3], G[3][3];  // G = F^-T
  double I1, I2, J, t1, t2, t3;
};

__device__ __forceinline__ void mr_state(const double F[3][3], double mu10, double mu01, double kappa, MRState& s) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double c = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        c += F[k][i] * F[k][j];
        b += F[i][k] * F[j][k];
      }
      s.C[i][j] = c;
      s.FFT[i][j] = b;
    }
  s.I1 = s.C[0][0] + s.C[1][1] + s.C[2][2];
  double trC2 = 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int k = 0; k < 3; k++) trC2 += s.C[i][k] * s.C[k][i];
  s.I2 = 0.5 * (s.I1 * s.I1 - trC2);
  s.J = det3(F);
  inv_transpose3(F, s.J, s.G);
  const double J13 = cbrt(s.J);
  const double Jm23 = 1.0 / (J13 * J13);
  const double Jm43 = Jm23 * Jm23;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) v += F[i][k] * s.C[k][j];
      s.FC[i][j] = v;
    }
  s.t1 = 2.0 * mu10 * Jm23;
  s.t2 = 2.0 * mu01 * Jm43;
  s.t3 = kappa * (s.J - 1.0) * s.J;
}

// First Piola-Kirchhoff stress.  SVK: SVK.cuh:14-32;  MR: MooneyRivlin.cuh:45-111.
__device__ __forceinline__ void elastic_P(const double F[3][3], const Material& mat, double P[3][3]) {
  if (mat.model == kMooneyRivlin) {
    MRState s;
    mr_state(F, mat.mu10, mat.mu01, mat.kappa, s);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const double term1 = F[i][j] - (s.I1 / 3.0) * s.G[i][j];
        const double term2 = s.I1 * F[i][j] - s.FC[i][j] - (2.0 * s.I2 / 3.0) * s.G[i][j];
        P[i][j] = s.t1 * term1 + s.t2 * term2 + s.t3 * s.G[i][j];
      }
  } else {
    double FFt[3][3], trFtF = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double b = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) b += F[i][k] * F[j][k];
        FFt[i][j] = b;
        trFtF += F[i][j] * F[i][j];
      }
    const double lf = mat.lambda * (0.5 * trFtF - 1.5);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double fftf = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) fftf += FFt[i][k] * F[k][j];
        P[i][j] = lf * F[i][j] + mat.mu * (fftf - F[i][j]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// reference gradients: dn_du_pre_kernel (FEAT10Data.cu:97-204), thread per (element, qp)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void solve3(const double A[3][3], const double b[3], double x[3]) {
  // partial-pivot Gaussian elimination, zero solution if |pivot| < 1e-14 (FEAT10DataFunc.cuh:30-83)
  double m[3][4];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) m[i][j] = A[i][j];
    m[i][3] = b[i];
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int piv = k;
    double best = fabs(m[k][k]);
#pragma unroll
    for (int i = k + 1; i < 3; i++)
      if (fabs(m[i][k]) > best) {
        best = fabs(m[i][k]);
        piv = i;
      }
#pragma unroll
    for (int i = k + 1; i < 3; i++)
      if (piv == i) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const double t = m[k][j];
          m[k][j] = m[i][j];
          m[i][j] = t;
        }
      }
    if (fabs(m[k][k]) < 1e-14) {
      x[0] = x[1] = x[2] = 0.0;
      return;
    }
#pragma unroll
    for (int i = k + 1; i < 3; i++) {
      const double f = m[i][k] / m[k][k];
#pragma unroll
      for (int j = k; j < 4; j++) m[i][j] -= f * m[k][j];
    }
  }
  x[2] = m[2][3] / m[2][2];
  x[1] = (m[1][3] - m[1][2] * x[2]) / m[1][1];
  x[0] = (m[0][3] - m[0][2] * x[2] - m[0][1] * x[1]) / m[0][0];
}

__global__ void dndu_pre_kernel(int E, int Epad, const int* __restrict__ conn, const double* __restrict__ x,
                                const double* __restrict__ y, const double* __restrict__ z,
                                const double* __restrict__ qx, const double* __restrict__ qy,
                                const double* __restrict__ qz, double* __restrict__ gradN,
                                double* __restrict__ gradN_t, double* __restrict__ detJ) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = tid / kNQ, q = tid % kNQ;
  if (e >= E) return;
  const double L[4] = {1.0 - qx[q] - qy[q] - qz[q], qx[q], qy[q], qz[q]};
  const double dL[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  const int edges[6][2] = {{0, 1}, {1, 2}, {0, 2}, {0, 3}, {1, 3}, {2, 3}};  // FEAT10Data.cu:143
  double dN[kNN][3];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int d = 0; d < 3; d++) dN[i][d] = (4.0 * L[i] - 1.0) * dL[i][d];
#pragma unroll
  for (int k = 0; k < 6; k++)
#pragma unroll
    for (int d = 0; d < 3; d++)
      dN[k + 4][d] = 4.0 * (L[edges[k][0]] * dL[edges[k][1]][d] + L[edges[k][1]] * dL[edges[k][0]][d]);
  double Jm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
  for (int a = 0; a < kNN; a++) {
    const int g = conn[a * E + e];
    const double X[3] = {x[g], y[g], z[g]};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Jm[i][j] += X[i] * dN[a][j];
  }
  detJ[e * kNQ + q] = det3(Jm);
  double JT[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) JT[i][j] = Jm[j][i];
  double* g_ref = gradN + ((size_t)e * kNQ + q) * 30;
#pragma unroll
  for (int a = 0; a < kNN; a++) {
    double ga[3];
    solve3(JT, dN[a], ga);
#pragma unroll
    for (int d = 0; d < 3; d++) {
      g_ref[a + 10 * d] = ga[d];
      gradN_t[((size_t)(q * 3 + d) * kNN + a) * Epad + e] = ga[d];
    }
  }
}

void launch_dndu_pre(hipStream_t s, int E, int Epad, const int* conn, const double* x, const double* y,
                     const double* z, const double* qx, const double* qy, const double* qz, double* gradN,
                     double* gradN_t, double* detJ) {
  const int total = E * kNQ;
  hipLaunchKernelGGL(dndu_pre_kernel, dim3((total + 127) / 128), dim3(128), 0, s, E, Epad, conn, x, y, z, qx, qy,
                     qz, gradN, gradN_t, detJ);
}

// Wave-private LDS hand-offs (the fused assembly kernels' workgroup is ONE wavefront; the residual launch transposes its
// stores inside each wavefront's own slice): a wavefront's LDS instructions execute in program order, so data written by one
// lane is visible to the lanes of every later LDS instruction without a barrier.  What remains of __syncthreads() is the
// compiler-level ordering -- and NOT its s_waitcnt vmcnt(0), which would drain the prefetched index loads and the H row
// stores at every one of the 3-5 synchronisation points of a pass.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// residual: fused compute_p + compute_internal_force (FEAT10DataFunc.cuh:85-293,397-458)
// thread per element; fbuf[e][a][d] = sum_q (P_q grad N_a) detJ_q w_q
// ------------------------------------------------------------------------------------------------
template <int S, int Q, bool STORE, bool MASS>
__global__ __launch_bounds__(128) void residual_kernel(ElemView m, Material mat, const double* __restrict__ v,
                                                      double* __restrict__ fbuf, double* __restrict__ Fo,
                                                      double* __restrict__ Po, double* __restrict__ Fdo,
                                                      double* __restrict__ Pvo, double* __restrict__ Fq, MassTerm mt,
                                                      double fq_h, int fq_slots) {
  // T10 on the solver path (no F/P buffers kept): the point records of the affine assembly and the {force | inertia} rows
  // go through a wave-private LDS transpose, so that every store instruction writes whole 128-byte lines (thread-per-
  // element stores of 128-byte / 48-byte records leave each line to eight / six separate instructions: WRITE_SIZE 1.8x)
  constexpr bool kTr = (S == kNN) && !STORE;
  constexpr int kTrRow = 54;  // doubles per lane: the five 10-double F records of an element + pad (432 B, 16-byte aligned)
  __shared__ __attribute__((aligned(16))) double tr_all[kTr ? 2 * 64 * kTrRow : 2];
  double* tr = tr_all + (kTr ? (threadIdx.x >> 6) * (64 * kTrRow) : 0);  // this wavefront's slice
  const int lane = threadIdx.x & 63;
  const int e0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63);  // first element of this wavefront
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.E) {
    if (!kTr) return;
    e = m.E - 1;  // lanes past the end stay for the transposes: they recompute the last element and store its values again
  }
  const bool damp = (v != nullptr) && (mat.eta != 0.0 || mat.lamd != 0.0);  // FEAT10DataFunc.cuh:137
  int gn[S];
  double xn[S][3];
#pragma unroll
  for (int a = 0; a < S; a++) {
    gn[a] = m.conn[(size_t)a * m.E + e];
    xn[a][0] = m.x[gn[a]];
    xn[a][1] = m.y[gn[a]];
    xn[a][2] = m.z[gn[a]];
  }
  double f[S][3];
#pragma unroll
  for (int a = 0; a < S; a++) f[a][0] = f[a][1] = f[a][2] = 0.0;
  // inertia rows M_e (v - v_prev) / h:  rows_a = rho/h sum_q N_a(q) dV_q sum_b N_b(q) (v - v_prev)_b.  The per-point
  // sums Wq = sum_b N_b(q) (v - v_prev)_b and factors cq = rho/h dV_q wait here for the force rows (written together)
  double Wq[MASS ? kNQ : 1][3], cq[MASS ? kNQ : 1];
  if (MASS) {
    double dv[S][3];
#pragma unroll
    for (int a = 0; a < S; a++)
#pragma unroll
      for (int i = 0; i < 3; i++) dv[a][i] = v[3 * gn[a] + i] - mt.vprev[3 * gn[a] + i];
#pragma unroll
    for (int q = 0; q < (MASS ? kNQ : 0); q++) {
      double w[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int a = 0; a < S; a++)
#pragma unroll
        for (int i = 0; i < 3; i++) w[i] += mt.Nq[q][a < kNN ? a : 0] * dv[a][i];
      cq[q] = mt.rho_inv_h * m.detJ[(size_t)e * Q + q] * m.qw[q];
#pragma unroll
      for (int i = 0; i < 3; i++) Wq[q][i] = w[i];
    }
  }

  // T10 solver path: the launch runs one wavefront per SIMD (its registers hold x, f and a point's 30 gradients), so
  // nothing but the wavefront itself hides the round trip of the gradient loads: the next point's gradients are in
  // flight while this point computes
  constexpr bool kPre = kTr;
  double hn[kPre ? S : 1][3];
  if (kPre) {
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
      for (int a = 0; a < (kPre ? S : 0); a++) hn[a][d] = m.gradN_t[((size_t)d * S + a) * m.Epad + e];
  }
#pragma unroll 1
  for (int q = 0; q < Q; q++) {
    double hq[S][3];
    if (kPre) {
#pragma unroll
      for (int a = 0; a < (kPre ? S : 0); a++)
#pragma unroll
        for (int d = 0; d < 3; d++) hq[a][d] = hn[a][d];
      if (q + 1 < Q) {
#pragma unroll
        for (int d = 0; d < 3; d++)
#pragma unroll
          for (int a = 0; a < (kPre ? S : 0); a++) hn[a][d] = m.gradN_t[((size_t)((q + 1) * 3 + d) * S + a) * m.Epad + e];
      }
    } else {
#pragma unroll
      for (int d = 0; d < 3; d++)
#pragma unroll
        for (int a = 0; a < S; a++) hq[a][d] = m.gradN_t[((size_t)(q * 3 + d) * S + a) * m.Epad + e];
    }
    double F[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int a = 0; a < S; a++)
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) F[i][j] += xn[a][i] * hq[a][j];
    if (Fq && fq_h > 0.0) {
      // affine form of the fused assembly: F (row-major, 10 doubles) per point, [E][5][10] with the centroid point in slot 0
      // and the point where vertex p has L = 1/2 in slot p + 1; everything else of the point (F F^T, tr E) is formed by
      // the assembly's lanes.  Collected in LDS, written after the loop.
      const int slot = (fq_slots >> (4 * q)) & 15;
      double2* fo = kTr ? reinterpret_cast<double2*>(tr + lane * kTrRow + slot * 10)
                        : reinterpret_cast<double2*>(Fq + ((size_t)e * kNQ + slot) * 10);
      fo[0] = make_double2(F[0][0], F[0][1]);
      fo[1] = make_double2(F[0][2], F[1][0]);
      fo[2] = make_double2(F[1][1], F[1][2]);
      fo[3] = make_double2(F[2][0], F[2][1]);
      fo[4] = make_double2(F[2][2], 0.0);
    } else if (Fq) {  // row-major F per (element, point): what the fused assembly stages instead of rebuilding F
      double* fo = Fq + ((size_t)e * Q + q) * 9;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) fo[i * 3 + j] = F[i][j];
    }
    double P[3][3];
    elastic_P(F, mat, P);
    double Fd[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Pv[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    if (damp) {
      // Fdot = sum v_a (x) h_a ; Edot = sym(Fdot^T F) ; S = 2 eta Edot + lamd tr(Edot) I ; P_vis = F S
#pragma unroll
      for (int a = 0; a < S; a++) {
        const double va[3] = {v[3 * gn[a] + 0], v[3 * gn[a] + 1], v[3 * gn[a] + 2]};
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) Fd[i][j] += va[i] * hq[a][j];
      }
      double Ed[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double a1 = 0.0, a2 = 0.0;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            a1 += Fd[k][i] * F[k][j];
            a2 += F[k][i] * Fd[k][j];
          }
          Ed[i][j] = 0.5 * (a1 + a2);
        }
      const double trEd = Ed[0][0] + Ed[1][1] + Ed[2][2];
      double Sv[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Sv[i][j] = 2.0 * mat.eta * Ed[i][j] + (i == j ? mat.lamd * trEd : 0.0);
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < 3; k++) s += F[i][k] * Sv[k][j];
          Pv[i][j] = s;
          P[i][j] += s;
        }
    }
    if (STORE) {  // CalcP keeps the reference's F/P buffers (col-major 3x3 per (e,q)) for Retrieve*
      const size_t o = ((size_t)e * Q + q) * 9;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          Fo[o + i + 3 * j] = F[i][j];
          Po[o + i + 3 * j] = P[i][j];
          Fdo[o + i + 3 * j] = Fd[i][j];
          Pvo[o + i + 3 * j] = Pv[i][j];
        }
    }
    const double dV = m.detJ[(size_t)e * Q + q] * m.qw[q];
#pragma unroll
    for (int a = 0; a < S; a++)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const double c = P[i][0] * hq[a][0] + P[i][1] * hq[a][1] + P[i][2] * hq[a][2];
        f[a][i] += c * dV;
      }
  }
  if (kTr && Fq && fq_h > 0.0) {
    // the wavefront's 64 x 5 F records are 25 600 contiguous bytes: whole-line stores, 16 bytes per lane
    wave_sync();
    double2* out = reinterpret_cast<double2*>(Fq + (size_t)e0 * 50);
#pragma unroll 5
    for (int j = 0; j < 25; j++) {
      const int idx = lane + 64 * j, el = idx / 25, c = idx - 25 * el;
      const double2 val = reinterpret_cast<const double2*>(tr + el * kTrRow)[c];
      if (e0 + el < m.E) out[idx] = val;
    }
    wave_sync();
  }
  if (MASS) {
    // one 48-byte record {force row | inertia row} per (node, element) for the gather of grad_light_kernel, [a][Epad][6]
    // node-major: a wavefront's 64 records of local node a are 3 072 contiguous bytes (Epad is a multiple of 64), stored
    // as three whole-wave 16-byte-per-lane instructions
#pragma unroll
    for (int a = 0; a < S; a++) {
      double mr[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < (MASS ? kNQ : 0); q++) {
        const double ca = cq[q] * mt.Nq[q][a < kNN ? a : 0];
#pragma unroll
        for (int i = 0; i < 3; i++) mr[i] += ca * Wq[q][i];
      }
      double2* my = reinterpret_cast<double2*>(tr + lane * 6);
      my[0] = make_double2(f[a][0], f[a][1]);
      my[1] = make_double2(f[a][2], mr[0]);
      my[2] = make_double2(mr[1], mr[2]);
      wave_sync();
      double2* out = reinterpret_cast<double2*>(mt.mbuf + ((size_t)a * m.Epad + e0) * 6);
#pragma unroll
      for (int j = 0; j < 3; j++)
        if (e0 < m.E) out[lane + 64 * j] = reinterpret_cast<const double2*>(tr)[lane + 64 * j];  // records past E: padding
      wave_sync();
    }
    return;
  }
  double* out = fbuf + (size_t)e * (3 * S);
#pragma unroll
  for (int a = 0; a < S; a++)
#pragma unroll
    for (int i = 0; i < 3; i++) out[a * 3 + i] = f[a][i];
}

template <int S, int Q>
static void launch_residual_t(hipStream_t s, const ElemView& m, const Material& mat, const double* v, double* fbuf,
                              double* F, double* P, double* Fdot, double* Pvis, double* Fq, const MassTerm* mt,
                              double fq_h, int fq_slots) {
  const dim3 grid((m.E + 127) / 128), block(128);
  MassTerm none{};
  none.vprev = nullptr;
  if (F)
    hipLaunchKernelGGL((residual_kernel<S, Q, true, false>), grid, block, 0, s, m, mat, v, fbuf, F, P, Fdot, Pvis, Fq, none,
                       fq_h, fq_slots);
  else if (S == kNN && mt && mt->vprev && v)
    hipLaunchKernelGGL((residual_kernel<S, Q, false, (S == kNN)>), grid, block, 0, s, m, mat, v, fbuf, nullptr, nullptr,
                       nullptr, nullptr, Fq, *mt, fq_h, fq_slots);
  else
    hipLaunchKernelGGL((residual_kernel<S, Q, false, false>), grid, block, 0, s, m, mat, v, fbuf, nullptr, nullptr, nullptr,
                       nullptr, Fq, none, fq_h, fq_slots);
}

void launch_residual(hipStream_t s, const ElemView& m, const Material& mat, const double* v, double* fbuf, double* F,
                     double* P, double* Fdot, double* Pvis, double* Fq, const MassTerm* mt, double fq_h, int fq_slots) {
  if (m.S == 10) launch_residual_t<10, 5>(s, m, mat, v, fbuf, F, P, Fdot, Pvis, Fq, mt, fq_h, fq_slots);
  else if (m.S == 8) launch_residual_t<8, 12>(s, m, mat, v, fbuf, F, P, Fdot, Pvis, Fq, nullptr, 0.0, 0);
  else launch_residual_t<16, 48>(s, m, mat, v, fbuf, F, P, Fdot, Pvis, Fq, nullptr, 0.0, 0);
}

__global__ void fint_gather_kernel(int N, Incidence inc, const double* __restrict__ fbuf,
                                   double* __restrict__ f_int) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double f0 = 0.0, f1 = 0.0, f2 = 0.0;
  for (int k = inc.n2e_off[i]; k < inc.n2e_off[i + 1]; k++) {
    const double* r = fbuf + (size_t)inc.n2e[k] * 3;  // (e*10+il)*3 == e*30 + il*3
    f0 += r[0];
    f1 += r[1];
    f2 += r[2];
  }
  f_int[3 * i + 0] = f0;
  f_int[3 * i + 1] = f1;
  f_int[3 * i + 2] = f2;
}

void launch_fint_gather(hipStream_t s, int N, const Incidence& inc, const double* fbuf, double* f_int) {
  hipLaunchKernelGGL(fint_gather_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, inc, fbuf, f_int);
}

// constraint values c = x[fixed] - target (FEAT10DataFunc.cuh:468-483)
__global__ void constraint_kernel(int n_fixed, const int* __restrict__ fixed, const double* __restrict__ x,
                                  const double* __restrict__ y, const double* __restrict__ z,
                                  const double* __restrict__ xt, const double* __restrict__ yt,
                                  const double* __restrict__ zt, double* __restrict__ c) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_fixed) return;
  const int n = fixed[k];
  c[3 * k + 0] = x[n] - xt[n];
  c[3 * k + 1] = y[n] - yt[n];
  c[3 * k + 2] = z[n] - zt[n];
}

void launch_constraint(hipStream_t s, int n_fixed, const int* fixed_nodes, const double* x, const double* y,
                       const double* z, const double* xt, const double* yt, const double* zt, double* cons) {
  if (n_fixed <= 0) return;
  hipLaunchKernelGGL(constraint_kernel, dim3((n_fixed + 255) / 256), dim3(256), 0, s, n_fixed, fixed_nodes, x, y, z,
                     xt, yt, zt, cons);
}

// ---- general linear constraints  c = J x - rhs  (ANCF3243DataFunc.cuh:477-499) ------------------------------
// J is CSR over constraint rows with columns in the flattened DOF space (3*coef + component).
__global__ void lin_constraint_kernel(int nc, const int* __restrict__ joff, const int* __restrict__ jcol,
                                      const double* __restrict__ jval, const double* __restrict__ rhs,
                                      const double* __restrict__ x, const double* __restrict__ y,
                                      const double* __restrict__ z, double* __restrict__ c) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nc) return;
  double sum = 0.0;
  for (int k = joff[r]; k < joff[r + 1]; k++) {
    const int col = jcol[k], coef = col / 3, comp = col - 3 * coef;
    const double v = comp == 0 ? x[coef] : (comp == 1 ? y[coef] : z[coef]);
    sum += jval[k] * v;
  }
  c[r] = sum - rhs[r];
}
void launch_lin_constraint(hipStream_t s, int nc, const int* joff, const int* jcol, const double* jval,
                           const double* rhs, const double* x, const double* y, const double* z, double* c) {
  if (nc <= 0) return;
  hipLaunchKernelGGL(lin_constraint_kernel, dim3((nc + 255) / 256), dim3(256), 0, s, nc, joff, jcol, jval, rhs, x, y, z, c);
}

// g[dof] += h * sum_k J^T[dof][k] (lambda_k + rho c_k)   (SyncedNewton.cu:377-404); J^T is CSR over DOF rows, its
// entries in ascending constraint id (fixed summation order)
__global__ void lin_constraint_grad_kernel(int ndof, const int* __restrict__ jtoff, const int* __restrict__ jtcol,
                                           const double* __restrict__ jtval, const double* __restrict__ lam,
                                           const double* __restrict__ c, double h, double rho,
                                           double* __restrict__ g) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ndof) return;
  const int a = jtoff[i], b = jtoff[i + 1];
  if (a == b) return;
  double sum = 0.0;
  for (int k = a; k < b; k++) {
    const int r = jtcol[k];
    sum += jtval[k] * (lam[r] + rho * c[r]);
  }
  g[i] += h * sum;
}
void launch_lin_constraint_grad(hipStream_t s, int ndof, const int* jtoff, const int* jtcol, const double* jtval,
                                const double* lam, const double* c, double h, double rho, double* g) {
  hipLaunchKernelGGL(lin_constraint_grad_kernel, dim3((ndof + 255) / 256), dim3(256), 0, s, ndof, jtoff, jtcol, jtval,
                     lam, c, h, rho, g);
}

// H += h^2 rho J^T J  (assemble_sparse_hessian_constraints, SyncedNewton.cu:292-341).  The reference scatters per
// constraint row with atomics; here ONE thread owns one DOF row of H (no atomics, fixed order): for every constraint
// r in the row's J^T list, for every entry (dof_j, J_rj) of J's row r:  H[i][dof_j] += f J_ri J_rj.
// H layout: node row n = i/3, component d = i%3: value (d, k, e) at 9 off[n] + d*3deg + 3k + e, k = position of
// coefficient dof_j/3 in the node's sorted column list (the constraint-aware pattern holds every such pair).
__global__ void lin_constraint_hessian_kernel(int ndof, const int* __restrict__ jtoff, const int* __restrict__ jtcol,
                                              const double* __restrict__ jtval, const int* __restrict__ joff,
                                              const int* __restrict__ jcol, const double* __restrict__ jval,
                                              const int* __restrict__ off, const int* __restrict__ cols, double f,
                                              double* __restrict__ H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ndof) return;
  const int a = jtoff[i], b = jtoff[i + 1];
  if (a == b) return;
  const int n = i / 3, d = i - 3 * n;
  const int o = off[n], deg = off[n + 1] - o;
  double* Hrow = H + (size_t)9 * o + (size_t)d * 3 * deg;
  for (int k = a; k < b; k++) {
    const int r = jtcol[k];
    const double Jri = jtval[k];
    for (int m = joff[r]; m < joff[r + 1]; m++) {
      const int dj = jcol[m], cj = dj / 3, e = dj - 3 * cj;
      int lo = 0, hi = deg;  // lower_bound in the sorted column list
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cols[o + mid] < cj) lo = mid + 1;
        else hi = mid;
      }
      if (lo < deg && cols[o + lo] == cj) Hrow[3 * lo + e] += f * Jri * jval[m];
    }
  }
}
void launch_lin_constraint_hessian(hipStream_t s, int ndof, const int* jtoff, const int* jtcol, const double* jtval,
                                   const int* joff, const int* jcol, const double* jval, const int* off,
                                   const int* cols, double f, double* H) {
  hipLaunchKernelGGL(lin_constraint_hessian_kernel, dim3((ndof + 255) / 256), dim3(256), 0, s, ndof, jtoff, jtcol, jtval,
                     joff, jcol, jval, off, cols, f, H);
}

// Fused: f_int gather + constraints + grad L (SyncedNewton.cu:344-407).  A 32-lane half-wave per coefficient
// row: lanes walk the mass row (coalesced values / columns, gathered velocities) and the row's element force
// rows, then a fixed-order butterfly sums the six partials (deterministic).
__global__ __launch_bounds__(256) void grad_kernel(int N, Incidence inc, const double* __restrict__ fbuf,
                                                  const double* __restrict__ mval, const double* __restrict__ v,
                                                  const double* __restrict__ vprev, const double* __restrict__ f_ext,
                                                  const double* __restrict__ x, const double* __restrict__ y,
                                                  const double* __restrict__ z, const double* __restrict__ xt,
                                                  const double* __restrict__ yt, const double* __restrict__ zt,
                                                  const int* __restrict__ fixed_slot, const double* __restrict__ lam,
                                                  const double* __restrict__ nw, double h, double rho,
                                                  double* __restrict__ f_int, double* __restrict__ cons,
                                                  double* __restrict__ g) {
  const int l32 = threadIdx.x & 31;
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (i >= N) return;
  const double inv_h = 1.0 / h;
  double a[6] = {0, 0, 0, 0, 0, 0};  // f[0..2], mass term[0..2]
  for (int k = inc.n2e_off[i] + l32; k < inc.n2e_off[i + 1]; k += 32) {
    const double* r = fbuf + (size_t)inc.n2e[k] * 3;
    a[0] += r[0];
    a[1] += r[1];
    a[2] += r[2];
  }
  for (int k = inc.off[i] + l32; k < inc.off[i + 1]; k += 32) {
    const int c = inc.cols[k];
    const double mij = mval[k];
#pragma unroll
    for (int d = 0; d < 3; d++) a[3 + d] += mij * (v[3 * c + d] - vprev[3 * c + d]) * inv_h;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1)
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] += __shfl_xor(a[k], o);
  if (l32 < 3) {
    const int d = l32;
    const int slot = fixed_slot ? fixed_slot[i] : -1;
    const double fd = (d == 0) ? a[0] : ((d == 1) ? a[1] : a[2]);
    const double md = (d == 0) ? a[3] : ((d == 1) ? a[4] : a[5]);
    f_int[3 * i + d] = fd;
    double r = md + fd - f_ext[3 * i + d];
    if (slot >= 0) {
      const double cv = (d == 0) ? (x[i] - xt[i]) : ((d == 1) ? (y[i] - yt[i]) : (z[i] - zt[i]));
      cons[3 * slot + d] = cv;
      r += (nw ? nw[i] : 1.0) * h * (lam[3 * slot + d] + rho * cv);  // nw: 1/multiplicity across ranks
    }
    g[3 * i + d] = r;
  }
}

void launch_grad(hipStream_t s, int N, const Incidence& inc, const double* fbuf, const double* mval, const double* v,
                 const double* vprev, const double* f_ext, const double* x, const double* y, const double* z,
                 const double* xt, const double* yt, const double* zt, const int* fixed_slot, const double* lam,
                 const double* nw, double h, double rho, double* f_int, double* cons, double* g) {
  hipLaunchKernelGGL(grad_kernel, dim3((N + 7) / 8), dim3(256), 0, s, N, inc, fbuf, mval, v, vprev, f_ext, x, y, z,
                     xt, yt, zt, fixed_slot, lam, nw, h, rho, f_int, cons, g);
}

// grad L with the inertia rows of the residual launch (T10): 8 lanes per node gather the node's element rows of f_int
// and of M (v - v_prev) / h in ascending element order, fixed-order butterfly, then the same epilogue as grad_kernel.
__global__ __launch_bounds__(256) void grad_light_kernel(int N, int Epad, Incidence inc, const double* __restrict__ fbuf,
                                                        const double* __restrict__ mbuf, const double* __restrict__ f_ext,
                                                        const double* __restrict__ x, const double* __restrict__ y,
                                                        const double* __restrict__ z, const double* __restrict__ xt,
                                                        const double* __restrict__ yt, const double* __restrict__ zt,
                                                        const int* __restrict__ fixed_slot, const double* __restrict__ lam,
                                                        const double* __restrict__ nw, double h, double rho,
                                                        double* __restrict__ f_int, double* __restrict__ cons,
                                                        double* __restrict__ g) {
  const int l8 = threadIdx.x & 7;
  const int i = blockIdx.x * 32 + (threadIdx.x >> 3);
  if (i >= N) return;
  double a[6] = {0, 0, 0, 0, 0, 0};  // f[0..2], inertia[0..2]
  for (int k = inc.n2e_off[i] + l8; k < inc.n2e_off[i + 1]; k += 8) {
    const int code = inc.n2e[k], e = code / kNN, la = code - kNN * e;
    const double2* r = reinterpret_cast<const double2*>(mbuf + ((size_t)la * Epad + e) * 6);  // [a][Epad][6], 16-byte aligned
    const double2 r0 = r[0], r1 = r[1], r2 = r[2];
    a[0] += r0.x;
    a[1] += r0.y;
    a[2] += r1.x;
    a[3] += r1.y;
    a[4] += r2.x;
    a[5] += r2.y;
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1)
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] += __shfl_xor(a[k], o);
  if (l8 < 3) {
    const int d = l8;
    const int slot = fixed_slot ? fixed_slot[i] : -1;
    const double fd = (d == 0) ? a[0] : ((d == 1) ? a[1] : a[2]);
    const double md = (d == 0) ? a[3] : ((d == 1) ? a[4] : a[5]);
    f_int[3 * i + d] = fd;
    double r = md + fd - f_ext[3 * i + d];
    if (slot >= 0) {
      const double cv = (d == 0) ? (x[i] - xt[i]) : ((d == 1) ? (y[i] - yt[i]) : (z[i] - zt[i]));
      cons[3 * slot + d] = cv;
      r += (nw ? nw[i] : 1.0) * h * (lam[3 * slot + d] + rho * cv);
    }
    g[3 * i + d] = r;
  }
}

void launch_grad_light(hipStream_t s, int N, int Epad, const Incidence& inc, const double* fbuf, const double* mbuf,
                       const double* f_ext, const double* x, const double* y, const double* z, const double* xt,
                       const double* yt, const double* zt, const int* fixed_slot, const double* lam, const double* nw,
                       double h, double rho, double* f_int, double* cons, double* g) {
  hipLaunchKernelGGL(grad_light_kernel, dim3((N + 31) / 32), dim3(256), 0, s, N, Epad, inc, fbuf, mbuf, f_ext, x, y, z, xt, yt, zt,
                     fixed_slot, lam, nw, h, rho, f_int, cons, g);
}

// ------------------------------------------------------------------------------------------------
// tangent: one element per wavefront; lane p (+64, +128) owns the node-pair blocks (i <= j)
// ------------------------------------------------------------------------------------------------

// one lane's double as a wave-uniform value (two v_readlane_b32; `l` must be wave-uniform)
__device__ __forceinline__ double read_lane_f64(double v, int l) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffLL), l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// One 64-thread workgroup == one wavefront, so __syncthreads() is a wave-local fence.  The Q points are staged
// through LDS in chunks of QC (all 5 for T10, 6 for the ANCF types), the write-out stage aliases the whole array.
template <int S, int Q, int QC, int MODEL>
struct TangentLds {
  static constexpr int kX = 0;                                   // xs[3][S]
  static constexpr int kH = ((3 * S + 1) / 2) * 2;               // hs[QC][3][S]  (global order)
  static constexpr int kF = kH + ((QC * 3 * S + 1) / 2) * 2;     // Fs[QC][9]
  static constexpr int kU = kF + ((QC * 9 + 1) / 2) * 2;         // SVK: per (point, node) records ; MR: st | At
  // SVK: per (point, node) the packed record {h0,h1,h2,Fh0,Fh1,Fh2} (three 16-byte LDS reads per node in the pair loop)
  static constexpr int kUsize = (MODEL == kSVK) ? (QC * 6 * S) : (QC * 64 + QC * 81);  // SVK: point scalars live in lanes
  static constexpr int kPairs = S * (S + 1) / 2;
  static constexpr int kRaw = (kU + kUsize > kPairs * 9) ? (kU + kUsize) : (kPairs * 9);
  static constexpr int kDV = ((kRaw + 1) / 2) * 2;                // dV[Q] = det J w of every point, loaded once per element
  static constexpr int kTotal = kDV + ((Q + 1) / 2) * 2;          // even: every wave's slice stays 16-byte aligned
};

// WPB wavefronts per workgroup, one element each (own LDS slice); all waves run the same barrier sequence, so
// __syncthreads() stays legal; WPB > 1 cuts the workgroup dispatch count (972k single-wave groups at config C).
template <int S, int Q, int QC, int MODEL, int WPB>
__global__ __launch_bounds__(64 * WPB, (S == 16 && MODEL == kSVK && WPB == 1) ? 3 : 1) void tangent_blocks_kernel(ElemView m, Material mat, double h,
                                                                 double* __restrict__ Kbuf) {
  using LD = TangentLds<S, Q, QC, MODEL>;
  constexpr int P = LD::kPairs;
  constexpr int NPL = (P + 63) / 64;  // pairs per lane
  __shared__ __attribute__((aligned(16))) double lds_all[LD::kTotal * WPB];
  double* lds = lds_all + (threadIdx.x >> 6) * LD::kTotal;
  double* xs = lds + LD::kX;
  double* hs = lds + LD::kH;
  double* Fs = lds + LD::kF;
  double* U = lds + LD::kU;
  const int E = m.E;
  const int e_raw = blockIdx.x * WPB + (threadIdx.x >> 6);
  const bool live = e_raw < E;
  const int e = live ? e_raw : E - 1;  // idle waves shadow the last element (they must still hit every barrier)
  const int lane = threadIdx.x & 63;
  // one wavefront per workgroup: LDS hand-offs need program order only (wave_sync), not s_barrier + s_waitcnt vmcnt(0) --
  // which would also drain the next chunk's gradient loads issued ahead of the pair loop
  auto sync = [] {
    if (WPB == 1) wave_sync();
    else __syncthreads();
  };

  double* dVs = lds + LD::kDV;
  // det J w of all Q points up front (one global round trip per element instead of one per chunk in front of the pair loop)
  for (int t = lane; t < Q; t += 64) dVs[t] = m.detJ[(size_t)e * Q + t] * m.qw[t];
  for (int t = lane; t < 3 * S; t += 64) {
    const int a = t % S, d = t / S;
    const int g = m.conn[(size_t)a * E + e];
    const double* src = (d == 0) ? m.x : ((d == 1) ? m.y : m.z);
    xs[d * S + a] = src[g];
  }
  // The lane's node pairs: (pi, pj0 + n), n < pcnt -- all in ONE block row of the element matrix.  With one pair per lane
  // (T10, ANCF-3243) that is pair `lane` of the row-major upper triangle.  With several (ANCF-3443: 136 pairs, 3 per lane)
  // row i is cut into runs of NPL consecutive columns, one lane per run (51 lanes busy): a point's record of node i, its
  // scalars and F F^T are read from LDS once per lane and point instead of once per pair -- the pair loop of the shell is
  // bound by LDS instruction issue, not by its fp64 operations.
  int pi = 0, pj0 = 0, pcnt = 0;
  static_assert(lane_pair_run_lanes(S, NPL) <= 64, "the pair runs of an element must fit into one wavefront");
  lane_pair_run(S, NPL, lane, pi, pj0, pcnt);  // pair_runs.h
  // SVK accumulates three q-sums per pair instead of the block itself (the ratio of the two rank-1 coefficients does not
  // depend on the point): O = sum dV Fh_i (x) Fh_j (acc) and TT = sum B1 (h_i.h_j) F F^T + the diagonal term (6 unique):
  // 26 fp64 operations per (pair, point) instead of 42
  double acc[NPL][9], accT[MODEL == kSVK ? NPL : 1][6];
#pragma unroll
  for (int n = 0; n < NPL; n++) {
#pragma unroll
    for (int k = 0; k < 9; k++) acc[n][k] = 0.0;
    if (MODEL == kSVK) {
#pragma unroll
      for (int k = 0; k < 6; k++) accT[n][k] = 0.0;
    }
  }

  constexpr int kPre = (QC * 3 * S + 63) / 64;
  double pre[kPre];
  {
    const double* gsrc = m.gradN + (size_t)e * Q * (3 * S);
#pragma unroll
    for (int r = 0; r < kPre; r++) pre[r] = (lane + 64 * r < QC * 3 * S) ? gsrc[lane + 64 * r] : 0.0;
  }
#pragma unroll 1
  for (int q0 = 0; q0 < Q; q0 += QC) {
    sync();  // previous chunk fully consumed (also orders the xs stage)
#pragma unroll
    for (int r = 0; r < kPre; r++)
      if (lane + 64 * r < QC * 3 * S) hs[lane + 64 * r] = pre[r];
    if (q0 + QC < Q) {  // the next chunk's gradients travel while this chunk computes
      const double* gsrc = m.gradN + ((size_t)e * Q + q0 + QC) * (3 * S);
#pragma unroll
      for (int r = 0; r < kPre; r++)
        if (lane + 64 * r < QC * 3 * S) pre[r] = gsrc[lane + 64 * r];
    }
    sync();
    for (int t = lane; t < QC * 9; t += 64) {  // F_q[r][c] = sum_a x_a[r] h_a^q[c]
      const int q = t / 9, r = (t % 9) / 3, c = t % 3;
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < S; a++) s += xs[r * S + a] * hs[q * 3 * S + c * S + a];
      Fs[t] = s;
    }
    sync();

    if (MODEL == kSVK) {
      double* G = U;                  // [QC][S][6]: h_a (3), F h_a (3), 16-byte aligned records
      for (int t = lane; t < QC * S; t += 64) {  // Fh_a^q[r] = sum_c F_q[r][c] h_a^q[c]
        const int q = t / S, a = t - q * S;
        const double* hq = hs + q * 3 * S;
        const double* F = Fs + q * 9;
        const double h0 = hq[a], h1 = hq[S + a], h2 = hq[2 * S + a];
        double* g = G + (size_t)t * 6;
        g[0] = h0;
        g[1] = h1;
        g[2] = h2;
        g[3] = F[0] * h0 + F[1] * h1 + F[2] * h2;
        g[4] = F[3] * h0 + F[4] * h1 + F[5] * h2;
        g[5] = F[6] * h0 + F[7] * h1 + F[8] * h2;
      }
      // The point's scalars and F F^T are the same for every lane: lane q of the chunk keeps them in registers and the
      // pair loop fetches them with v_readlane into scalar registers -- five 16-byte LDS reads per (lane, point) less in a
      // loop that is bound by LDS instruction issue.
      static_assert(10 * QC <= 64, "one lane per (point of the chunk, value)");
      // value k of point q lives in lane 10 q + k: dV, B1, C0, C1, FF^T (00, 01, 02, 11, 12, 22)
      double ptv = 0.0;
      if (lane < 10 * QC) {
        const int q = lane / 10, k = lane - 10 * q;
        const double* F = Fs + q * 9;
        double Fl[9];
#pragma unroll
        for (int t = 0; t < 9; t++) Fl[t] = F[t];
        double trC = 0.0;
#pragma unroll
        for (int t = 0; t < 9; t++) trC += Fl[t] * Fl[t];
        const double trE = 0.5 * (trC - 3.0);
        const double dV = dVs[q0 + q];
        // h*K (SVK.cuh:35-55) + C_vis (FEAT10DataFunc.cuh:695-762) share their rank-1 structure:
        //   dV (h lambda + lamd) Fh_i (x) Fh_j + dV (h mu + eta) Fh_j (x) Fh_i : one sum O = sum_q dV Fh_i (x) Fh_j,
        //   combined as (h lambda + lamd) O + (h mu + eta) O^T when the block is written
        double val[10];
        val[0] = dV;
        val[1] = dV * (h * mat.mu + mat.eta);           // * (h_i.h_j) FF^T
        val[2] = dV * h * (mat.lambda * trE - mat.mu);  // * (h_i.h_j) I
        val[3] = dV * h * mat.mu;                       // * (Fh_i.Fh_j) I
        int kk = 4;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = i; j < 3; j++, kk++)
            val[kk] = Fl[i * 3 + 0] * Fl[j * 3 + 0] + Fl[i * 3 + 1] * Fl[j * 3 + 1] + Fl[i * 3 + 2] * Fl[j * 3 + 2];
#pragma unroll
        for (int t = 0; t < 10; t++) ptv = (k == t) ? val[t] : ptv;
      }
      sync();
      if (pcnt > 0) {
#pragma unroll 1
        for (int q = 0; q < QC; q++) {
          // per (lane, point): the row node's record, the point's scalars and F F^T
          const double2* gi = reinterpret_cast<const double2*>(G + ((size_t)q * S + pi) * 6);
          const double2 i01 = gi[0], i23 = gi[1], i45 = gi[2];
          const double hi0 = i01.x, hi1 = i01.y, hi2 = i23.x, fi0 = i23.y, fi1 = i45.x, fi2 = i45.y;
          const int l0 = 10 * q;
          const double dVq = read_lane_f64(ptv, l0), B1 = read_lane_f64(ptv, l0 + 1), C0 = read_lane_f64(ptv, l0 + 2),
                       C1 = read_lane_f64(ptv, l0 + 3);
          const double u0 = dVq * fi0, u1 = dVq * fi1, u2 = dVq * fi2;
          const double T[6] = {read_lane_f64(ptv, l0 + 4), read_lane_f64(ptv, l0 + 5), read_lane_f64(ptv, l0 + 6),
                               read_lane_f64(ptv, l0 + 7), read_lane_f64(ptv, l0 + 8), read_lane_f64(ptv, l0 + 9)};
#pragma unroll
          for (int n = 0; n < NPL; n++) {
            if (n >= pcnt) continue;
            const double2* gj = reinterpret_cast<const double2*>(G + ((size_t)q * S + pj0 + n) * 6);
            const double2 j01 = gj[0], j23 = gj[1], j45 = gj[2];
            const double hj0 = j01.x, hj1 = j01.y, hj2 = j23.x, fj0 = j23.y, fj1 = j45.x, fj2 = j45.y;
            const double s = hi0 * hj0 + hi1 * hj1 + hi2 * hj2;
            const double t = fi0 * fj0 + fi1 * fj1 + fi2 * fj2;
            const double bs = B1 * s;
            const double cd = C0 * s + C1 * t;  // the diagonal term rides on the diagonal entries of the F F^T sum
            acc[n][0] += u0 * fj0;
            acc[n][1] += u0 * fj1;
            acc[n][2] += u0 * fj2;
            acc[n][3] += u1 * fj0;
            acc[n][4] += u1 * fj1;
            acc[n][5] += u1 * fj2;
            acc[n][6] += u2 * fj0;
            acc[n][7] += u2 * fj1;
            acc[n][8] += u2 * fj2;
            accT[n][0] += bs * T[0] + cd;
            accT[n][1] += bs * T[1];
            accT[n][2] += bs * T[2];
            accT[n][3] += bs * T[3] + cd;
            accT[n][4] += bs * T[4];
            accT[n][5] += bs * T[5] + cd;
          }
        }
      }
    } else {
      // Mooney-Rivlin: per-qp state on lanes 0..QC-1, then the QC x 81 entries of
      //   At = dV * ( h * dP/dF  +  A_vis )         (MooneyRivlin.cuh:113-225, FEAT10DataFunc.cuh:695-762)
      // spread over the wave, then block(i,j)[d][e] = sum_JL At[d][J][e][L] h_i[J] h_j[L].
      double* st = U;            // [QC][64]: C 0, FC 9, FFT 18, G 27, T1 36, T2 45, scalars 54..
      double* At = U + QC * 64;  // [QC][81]
      if (lane < QC) {
        const int q = lane;
        double F[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) F[i][j] = Fs[q * 9 + i * 3 + j];
        MRState ms;
        mr_state(F, mat.mu10, mat.mu01, mat.kappa, ms);
        double* o = st + q * 64;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            o[0 + i * 3 + j] = ms.C[i][j];
            o[9 + i * 3 + j] = ms.FC[i][j];
            o[18 + i * 3 + j] = ms.FFT[i][j];
            o[27 + i * 3 + j] = ms.G[i][j];
            o[36 + i * 3 + j] = F[i][j] - (ms.I1 / 3.0) * ms.G[i][j];
            o[45 + i * 3 + j] = ms.I1 * F[i][j] - ms.FC[i][j] - (2.0 * ms.I2 / 3.0) * ms.G[i][j];
          }
        o[54] = ms.I1;
        o[55] = ms.I2;
        o[56] = ms.t1;
        o[57] = ms.t2;
        o[58] = ms.t3;
        o[59] = mat.kappa * (2.0 * ms.J - 1.0) * ms.J;
        o[60] = dVs[q0 + q];
      }
      sync();
      for (int n = lane; n < QC * 81; n += 64) {
        const int q = n / 81, r = n % 81;
        const int i = r / 27, j = (r / 9) % 3, k = (r / 3) % 3, l = r % 3;
        const double* o = st + q * 64;
        const double* F = Fs + q * 9;
        const double* C = o;
        const double* FC = o + 9;
        const double* FFT = o + 18;
        const double* G = o + 27;
        const double* T1 = o + 36;
        const double* T2 = o + 45;
        const double I1 = o[54], I2 = o[55], t1 = o[56], t2 = o[57], t3 = o[58], k3 = o[59], dV = o[60];
        const double dik = (i == k) ? 1.0 : 0.0, djl = (j == l) ? 1.0 : 0.0;
        const double Gkl = G[k * 3 + l], Gij = G[i * 3 + j], GilGkj = G[i * 3 + l] * G[k * 3 + j];
        const double dt1 = (-2.0 / 3.0) * t1 * Gkl;
        const double dt2 = (-4.0 / 3.0) * t2 * Gkl;
        const double dt3 = k3 * Gkl;
        const double dT1 = dik * djl - (2.0 / 3.0) * F[k * 3 + l] * Gij + (I1 / 3.0) * GilGkj;
        const double dT2 = 2.0 * F[k * 3 + l] * F[i * 3 + j] + I1 * dik * djl -
                           (dik * C[l * 3 + j] + F[i * 3 + l] * F[k * 3 + j] + djl * FFT[i * 3 + k]) -
                           (4.0 / 3.0) * (I1 * F[k * 3 + l] - FC[k * 3 + l]) * Gij + (2.0 * I2 / 3.0) * GilGkj;
        const double Ael = dt1 * T1[i * 3 + j] + t1 * dT1 + dt2 * T2[i * 3 + j] + t2 * dT2 + dt3 * Gij - t3 * GilGkj;
        // viscous: eta F[d][L]F[e][J] + eta FF^T[d][e] d_JL + lamd F[d][J]F[e][L]   (d=i, J=j, e=k, L=l)
        const double Avis = mat.eta * (F[i * 3 + l] * F[k * 3 + j] + FFT[i * 3 + k] * djl) +
                            mat.lamd * F[i * 3 + j] * F[k * 3 + l];
        At[n] = dV * (h * Ael + Avis);
      }
      sync();
      if (pcnt > 0) {
#pragma unroll 1
        for (int q = 0; q < QC; q++) {
          const double* hq = hs + q * 3 * S;
          const double hi[3] = {hq[pi], hq[S + pi], hq[2 * S + pi]};
          const double* A = At + q * 81;
          if (NPL == 1) {
            const int j = pj0;
            const double hj[3] = {hq[j], hq[S + j], hq[2 * S + j]};
#pragma unroll
            for (int d = 0; d < 3; d++)
#pragma unroll
              for (int ee = 0; ee < 3; ee++) {
                double sm = 0.0;
#pragma unroll
                for (int J = 0; J < 3; J++) {
                  const double* a = A + ((d * 3 + J) * 3 + ee) * 3;
                  sm += hi[J] * (a[0] * hj[0] + a[1] * hj[1] + a[2] * hj[2]);
                }
                acc[0][d * 3 + ee] += sm;
              }
          } else {
            // several pairs in one block row: contract the tangent tensor with h_i once per (lane, point)
            double Bh[3][3][3];
#pragma unroll
            for (int d = 0; d < 3; d++)
#pragma unroll
              for (int ee = 0; ee < 3; ee++)
#pragma unroll
                for (int L = 0; L < 3; L++) {
                  double b = 0.0;
#pragma unroll
                  for (int J = 0; J < 3; J++) b += hi[J] * A[((d * 3 + J) * 3 + ee) * 3 + L];
                  Bh[d][ee][L] = b;
                }
#pragma unroll
            for (int n = 0; n < NPL; n++) {
              if (n >= pcnt) continue;
              const int j = pj0 + n;
              const double hj[3] = {hq[j], hq[S + j], hq[2 * S + j]};
#pragma unroll
              for (int d = 0; d < 3; d++)
#pragma unroll
                for (int ee = 0; ee < 3; ee++)
                  acc[n][d * 3 + ee] += Bh[d][ee][0] * hj[0] + Bh[d][ee][1] * hj[1] + Bh[d][ee][2] * hj[2];
            }
          }
        }
      }
    }
  }
  sync();  // everyone is done reading the staged inputs: reuse LDS as the write-out stage
#pragma unroll
  for (int n = 0; n < NPL; n++) {
    if (n < pcnt) {
      const int p = pair_index(S, pi, pj0 + n);
      if (MODEL == kSVK) {
        const double ca = h * mat.lambda + mat.lamd, cb = h * mat.mu + mat.eta;
        const int tix[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++)
            lds[p * 9 + 3 * r + c] = ca * acc[n][3 * r + c] + cb * acc[n][3 * c + r] + accT[n][tix[3 * r + c]];
      } else {
#pragma unroll
        for (int k = 0; k < 9; k++) lds[p * 9 + k] = acc[n][k];
      }
    }
  }
  sync();
  double* out = Kbuf + (size_t)e * (P * 9);
  if (live)
    for (int t = lane; t < P * 9; t += 64) out[t] = lds[t];
}

template <int S, int Q, int QC>
static void launch_tangent_t(hipStream_t s, const ElemView& m, const Material& mat, double h, double* Kbuf) {
  static const int wpb = std::getenv("TLFEA_TANGENT_WPB") ? std::atoi(std::getenv("TLFEA_TANGENT_WPB")) : 1;
  if (wpb == 4) {
    const dim3 g((m.E + 3) / 4), b(256);
    if (mat.model == kMooneyRivlin)
      hipLaunchKernelGGL((tangent_blocks_kernel<S, Q, QC, kMooneyRivlin, 4>), g, b, 0, s, m, mat, h, Kbuf);
    else
      hipLaunchKernelGGL((tangent_blocks_kernel<S, Q, QC, kSVK, 4>), g, b, 0, s, m, mat, h, Kbuf);
    return;
  }
  if (mat.model == kMooneyRivlin)
    hipLaunchKernelGGL((tangent_blocks_kernel<S, Q, QC, kMooneyRivlin, 1>), dim3(m.E), dim3(64), 0, s, m, mat, h, Kbuf);
  else
    hipLaunchKernelGGL((tangent_blocks_kernel<S, Q, QC, kSVK, 1>), dim3(m.E), dim3(64), 0, s, m, mat, h, Kbuf);
}

void launch_tangent_blocks(hipStream_t s, const ElemView& m, const Material& mat, double h, double* Kbuf) {
  if (m.S == 10) launch_tangent_t<10, 5, 5>(s, m, mat, h, Kbuf);
  else if (m.S == 8) launch_tangent_t<8, 12, 6>(s, m, mat, h, Kbuf);
  else launch_tangent_t<16, 48, 6>(s, m, mat, h, Kbuf);
}

// ------------------------------------------------------------------------------------------------
// row-owner assembly: one node row (3 CSR rows of H) per wavefront.
// H layout == the reference's DOF-level CSR (SyncedNewton.cu:163-205): for node i with deg
// neighbours, values[9*off[i] + d*3*deg + 3*k + e] = H(3i+d, 3*cols[off[i]+k]+e).
// ------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(64) void assemble_rows_kernel(int N, Incidence inc, const double* __restrict__ Kbuf,
                                                          const double* __restrict__ mval, double inv_h,
                                                          const int* __restrict__ fixed_slot,
                                                          const double* __restrict__ nw, double penalty,
                                                          double* __restrict__ Hval) {
  extern __shared__ double acc[];
  const int i = blockIdx.x;
  const int lane = threadIdx.x;
  const int off0 = inc.off[i];
  const int deg = inc.off[i + 1] - off0;
  const int n9 = 9 * deg, row = 3 * deg;
  constexpr int npair = S * (S + 1) / 2;
  for (int t = lane; t < n9; t += 64) acc[t] = 0.0;
  __syncthreads();
  // M/h on the xyz-diagonal of every block (SyncedNewton.cu:214-259)
  for (int k = lane; k < deg; k += 64) {
    const double mh = mval[off0 + k] * inv_h;
    acc[0 * row + 3 * k + 0] = mh;
    acc[1 * row + 3 * k + 1] = mh;
    acc[2 * row + 3 * k + 2] = mh;
  }
  __syncthreads();
  // h^2 rho J^T J: one 1.0 per pinned DOF (SyncedNewton.cu:292-341, FEAT10Data.cu:443-459)
  if (lane < 3 && fixed_slot && fixed_slot[i] >= 0)
    acc[lane * row + 3 * inc.diagpos[i] + lane] += (nw ? nw[i] : 1.0) * penalty;
  __syncthreads();
  // The node's incident elements, ascending (fixed summation order).  A row is a chain of dependent memory
  // round trips (incidence code -> block values / column slots -> LDS add): the loads of a whole CHUNK of
  // elements are issued before the first add, branch-free (elements past the end repeat the last one and are
  // skipped at the add), so the row pays two round trips per chunk instead of two per element.
  constexpr int kChunk = 8;
  constexpr int kPass = (9 * S + 63) / 64;
  const int k0 = inc.n2e_off[i], k1 = inc.n2e_off[i + 1];
  // this lane's fixed share of an (element, local node) block row: entries t = lane + 64 u
  int tj[kPass], tdd[kPass], tee[kPass];
#pragma unroll
  for (int u = 0; u < kPass; u++) {
    const int t = min(lane + 64 * u, 9 * S - 1);
    tj[u] = t / 9;
    tdd[u] = (t % 9) / 3;
    tee[u] = t % 3;
  }
  for (int kc = k0; kc < k1; kc += kChunk) {
    const int klast = k1 - 1;
    int code[kChunk];
#pragma unroll
    for (int c = 0; c < kChunk; c++) code[c] = inc.n2e[min(kc + c, klast)];
    double v[kChunk][kPass];
    int pj[kChunk][kPass];
#pragma unroll
    for (int c = 0; c < kChunk; c++) {
      const int e = code[c] / S, il = code[c] - e * S;
      const double* Ke = Kbuf + (size_t)e * (npair * 9);
      const int* pos = inc.n2e_pos + (size_t)min(kc + c, klast) * S;
#pragma unroll
      for (int u = 0; u < kPass; u++) {
        const int j = tj[u];
        v[c][u] = (il <= j) ? Ke[pair_index(S, il, j) * 9 + tdd[u] * 3 + tee[u]]
                            : Ke[pair_index(S, j, il) * 9 + tee[u] * 3 + tdd[u]];
        pj[c][u] = pos[j];
      }
    }
#pragma unroll
    for (int c = 0; c < kChunk; c++) {
      if (kc + c < k1) {
#pragma unroll
        for (int u = 0; u < kPass; u++)
          if (lane + 64 * u < 9 * S) acc[tdd[u] * row + 3 * pj[c][u] + tee[u]] += v[c][u];
      }
      __syncthreads();
    }
  }
  double* out = Hval + (size_t)9 * off0;
  for (int t = lane; t < n9; t += 64) out[t] = acc[t];
}

void launch_assemble_rows(hipStream_t s, int N, int S, int maxdeg, const Incidence& inc, const double* Kbuf,
                          const double* mval, double inv_h, const int* fixed_slot, const double* nw, double penalty,
                          double* Hval) {
  const size_t lds = (size_t)9 * maxdeg * sizeof(double);
  if (S == 10)
    hipLaunchKernelGGL((assemble_rows_kernel<10>), dim3(N), dim3(64), lds, s, N, inc, Kbuf, mval, inv_h, fixed_slot, nw,
                       penalty, Hval);
  else if (S == 8)
    hipLaunchKernelGGL((assemble_rows_kernel<8>), dim3(N), dim3(64), lds, s, N, inc, Kbuf, mval, inv_h, fixed_slot, nw,
                       penalty, Hval);
  else
    hipLaunchKernelGGL((assemble_rows_kernel<16>), dim3(N), dim3(64), lds, s, N, inc, Kbuf, mval, inv_h, fixed_slot, nw,
                       penalty, Hval);
}

// ------------------------------------------------------------------------------------------------
// Fused tangent + assembly, T10 <10,5>, St.Venant-Kirchhoff (+ Kelvin-Voigt): compute_hessian_assemble_csr
// (FEAT10DataFunc.cuh:513-791) + assemble_sparse_hessian_{mass,tangent,constraints} (SyncedNewton.cu:214-341) in ONE
// launch with no element-block buffer in HBM.  Owner computes: a wavefront owns a chunk of row GROUPS of H (RowGroups;
// rows in Morton order, chunks dealt to the XCDs in contiguous ranges so that the grad-N re-reads of neighbouring
// rows hit that XCD's L2) and recomputes, for every (row i, incident element e) INSTANCE, the ten 3x3 blocks
// K_e(i, j), j = 0..9 -- 100 blocks per element over all its rows instead of the 55 of the symmetric element-wise
// form, in exchange for 7.9 kB per element of block-buffer traffic and a second kernel.
//   pass   = up to 6 instances x 10 column nodes = 60 lanes, lane (k, j) sums its block over the 5 points in registers
//   staged = per (instance, point) one 26-double record {h_i, F h_i, F, B1 F F^T, A1, B1, C0, C1} in LDS (F comes from
//            the residual launch that precedes every assembly: Fq); h_j goes from memory straight to registers
//   sum    = ds_add_f64 into the group's row accumulators in LDS (H layout); the first contribution to a block also
//            carries the block's M/h; the rows stream out once when the group's last pass is done (+ h^2 rho on the
//            diagonal of pinned rows)
// The kernel is built around load latency, not bandwidth: the pass table entry is read two passes ahead (scalar), the
// instance codes / packed offsets one pass ahead, so that a pass waits for ONE round trip (grad N, F, mass) before it
// computes; per-row records travel in registers from the group's first pass to its last.
// The LDS adds of one wave execute in program order, lanes of one instruction in the hardware's fixed conflict
// order, so the sum order is fixed: bitwise reproducible like the two-kernel path (tests check it).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int kAdInst = 6;                            // instances per pass
constexpr int kAdRec = 26;                            // doubles per staged record (St.Venant-Kirchhoff)
constexpr int kAdRecTotal = kAdInst * kNQ * kAdRec;   // 780 doubles = 6.1 KiB
constexpr int kAdRecMR = 60;                          // ... Mooney-Rivlin: seven 3-vectors, 2 scalars, c FF^T, F^-T, F, F C
constexpr int kAdHraw = kAdInst * kNQ * 4;            // 120 doubles
}  // namespace

// H row store that does not stay in the XCD's L2 (sc1: written through and dropped, MI355X_MICROARCH.md "stores of each
// flavour"): the 2.7 GB of H would otherwise evict the grad N lines the neighbouring rows are about to re-read
__device__ __forceinline__ void store_through(double* p, double v, int mode) {
  if (mode == 2)
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (mode == 1)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// EXP (tools/tune_assemble.py only): honours the work-skipping experiment bits of store_mode; the shipping instantiation
// (EXP = 0) keeps the store flavour (bits 0-1) and nothing else, so no environment variable can change what H holds
// MR: compressible Mooney-Rivlin (MooneyRivlin.cuh:113-225) instead of St.Venant-Kirchhoff.  The 81-entry tangent tensor
// A = dP/dF is never formed: the block K_ij[d][e] = sum_JL A[d][J][e][L] h_iJ h_jL is a sum of outer products of
//   p = F^-T h_i, fa = F h_i (row side, staged per (instance, point) with their scalar coefficients folded in) and
//   q = F^-T h_j, fb = F h_j, fcb = F C h_j (column side, 27 multiply-adds per lane and point),
// checked against the tensor form to round-off (tests/test_gpu_parity.py::test_hessian[mr|neo|mr_damped]); the
// Kelvin-Voigt block has the same rank-1 structure and rides on the same vectors.
template <int ROLLED, int EXP, int MR = 0>  // ROLLED 0: points unrolled (234 VGPRs, 2 waves per SIMD), 1: rolled (168 VGPRs, 3 waves per SIMD)
__global__ __launch_bounds__(64, (ROLLED && !MR) ? 3 : 2) void assemble_direct_kernel(ElemView m, Material mat, double h, RowGroups rg,
                                                               const double* __restrict__ Fq,
                                                               const double* __restrict__ mval, double inv_h,
                                                               const int* __restrict__ fixed_slot,
                                                               const double* __restrict__ nw, double penalty,
                                                               double* __restrict__ Hval, int store_mode_arg) {
  const int store_mode = EXP ? store_mode_arg : (store_mode_arg & 3);
  extern __shared__ __attribute__((aligned(16))) double lds_ad[];
  constexpr int REC = MR ? kAdRecMR : kAdRec, REC_TOTAL = kAdInst * kNQ * REC;
  double* rec = lds_ad;                        // [kAdInst][kNQ][REC]
  double* hraw = lds_ad + REC_TOTAL;           // [kAdInst][kNQ][4]: h_i of the pass's instances
  double* acc = lds_ad + REC_TOTAL + kAdHraw;  // the group's rows, each in H's layout [d][3 deg]
  // Blocks b, b + 8, ... share an XCD (round-robin dispatch).  XCD x owns groups [x Gper, (x+1) Gper) and its W resident
  // waves walk that range side by side: wave w takes groups w, w + W, w + 2W, ... -- at any time the XCD works on a
  // window of ~W consecutive groups, i.e. spatial neighbours whose elements overlap, so that the re-reads of an
  // element by the owners of its other rows hit the XCD's 4 MiB L2.
  const int W = gridDim.x >> 3;
  const int Gper = (rg.G + 7) >> 3;
  const int gbeg = (blockIdx.x & 7) * Gper, gend = min(rg.G, gbeg + Gper);
  if (gbeg + (int)(blockIdx.x >> 3) >= gend) return;
  const int lane = threadIdx.x;
  const int k = lane / kNN, j = lane - kNN * k;  // item (instance k, column node j); lanes 60..63: k == 6, idle
  const int ts = lane & 31;                      // staging: record ts of 30, lanes 0..29 kinematic half, 32..61 material half
  const bool stager = ts < kAdInst * kNQ;
  const int ks = stager ? ts / kNQ : 0, qs = stager ? ts - kNQ * (ts / kNQ) : 0;
  const int E = m.E, last_inst = rg.n_inst - 1;
  // lead cursor: runs two passes ahead of the pass being computed (all scalar state)
  int lg = gbeg + (blockIdx.x >> 3);                     // its group
  int lp = rg.g_pass_off[lg], lpe = rg.g_pass_off[lg + 1];  // its pass, end of its group's passes
  int np = 0, npe = 0;                                   // pass range of the wave's NEXT group (fetched one group ahead)
  if (lg + W < gend) {
    np = rg.g_pass_off[lg + W];
    npe = rg.g_pass_off[lg + W + 1];
  }
  bool lvalid = true;
  auto advance = [&]() {
    if (lp + 1 < lpe) {
      lp++;
      return;
    }
    lg += W;
    lvalid = lg < gend;
    lp = np;
    lpe = npe;
    if (lg + W < gend) {
      np = rg.g_pass_off[lg + W];
      npe = rg.g_pass_off[lg + W + 1];
    }
  };
  int4 ecur = rg.pt[lp];
  advance();
  bool vnxt = lvalid;
  int4 enxt = rg.pt[vnxt ? lp : 0];
  advance();
  // instance indices of a pass for this lane's two roles (clamped: idle lanes repeat a valid instance)
  auto inst_of = [&](const int4& en, int slot) { return min(en.x + min(slot, max((en.y & 7) - 1, 0)), last_inst); };
  int code_i, pk, mb, code_s;
  {
    const int ii = inst_of(ecur, k), is = inst_of(ecur, ks);
    code_i = rg.gi_code[ii];
    pk = rg.gi_pack[(size_t)ii * kNN + j];
    mb = rg.gi_mb[ii];
    code_s = rg.gi_code[is];
  }
  int4 ri = make_int4(0, 0, 0, 0);  // this lane's row of the current group (lane < rows)
  double pen = 0.0;                 // its h^2 rho (x 1/multiplicity) if the row is pinned

  bool vcur = true;
#pragma unroll 1
  while (vcur) {
    const bool vn2 = lvalid;
    const int4 en2 = rg.pt[vn2 ? lp : 0];
    advance();
    const int cnt = ecur.y & 7, nrows = ecur.y >> 8;
    const bool first = (ecur.y & 8) != 0, last = (ecur.y & 16) != 0;
    // ---- (1) this pass's data: one round trip -------------------------------------------------------------
    // (timing experiments, tools/tune_assemble.py: store_mode bit 8 = every load from element 0, bit 9 = no arithmetic)
    const int e = (store_mode & 256) ? 0 : code_i / kNN;
    const double* gN = m.gradN + (size_t)e * (kNQ * 3 * kNN) + j;
    double hj[kNQ][3];
#pragma unroll
    for (int q = 0; q < kNQ; q++)
#pragma unroll
      for (int d = 0; d < 3; d++) hj[q][d] = gN[q * 3 * kNN + d * kNN];
    const bool act = k < cnt;
    double mh = 0.0;
    if (act && pk < 0) mh = mval[mb + (pk & 0xffff) / 3];
    const int es = (store_mode & 256) ? 0 : code_s / kNN;
    double F[9], s0 = 0.0;  // s0: det J (material half)
    if (stager) {
      // the 72 bytes of F as four 16-byte loads + one: the five points of an element are contiguous, so the 30 lanes of
      // one instruction touch ~20 cache lines instead of 30 per 8-byte load (the vector L1 handles a line per clock)
      const double* Fp = Fq + ((size_t)es * kNQ + qs) * 9;
      const double2 f01 = *reinterpret_cast<const double2*>(Fp), f23 = *reinterpret_cast<const double2*>(Fp + 2),
                    f45 = *reinterpret_cast<const double2*>(Fp + 4), f67 = *reinterpret_cast<const double2*>(Fp + 6);
      F[0] = f01.x; F[1] = f01.y; F[2] = f23.x; F[3] = f23.y; F[4] = f45.x; F[5] = f45.y; F[6] = f67.x; F[7] = f67.y;
      F[8] = Fp[8];
      if (MR ? lane < 32 : lane >= 32) s0 = m.detJ[(size_t)es * kNQ + qs];
    }
    // ---- (2) the next pass's indices ------------------------------------------------------------------------
    int code_i_n, pk_n, mb_n, code_s_n;
    {
      const int ii = inst_of(enxt, k), is = inst_of(enxt, ks);
      code_i_n = rg.gi_code[ii];
      pk_n = rg.gi_pack[(size_t)ii * kNN + j];
      mb_n = rg.gi_mb[ii];
      code_s_n = rg.gi_code[is];
    }
    // ---- (3) first pass of a group: clear its accumulators, fetch its row records ---------------------------
    if (first) {
      for (int t = lane; t < ecur.w; t += 64) acc[t] = 0.0;
      pen = 0.0;
      if (lane < nrows) {
        ri = rg.gr_info[ecur.z + lane];
        if (fixed_slot && fixed_slot[ri.w] >= 0) pen = (nw ? nw[ri.w] : 1.0) * penalty;
      }
    }
    // h_i of instance k is the h_j of its lane j == il: published through LDS instead of 3 more scattered loads per record
    if (k < kAdInst && j == code_i - kNN * (code_i / kNN)) {
      double* hp = hraw + k * (kNQ * 4);
#pragma unroll
      for (int q = 0; q < kNQ; q++) {
        *reinterpret_cast<double2*>(hp + 4 * q) = make_double2(hj[q][0], hj[q][1]);
        hp[4 * q + 2] = hj[q][2];
      }
    }
    wave_sync();  // the previous pass has consumed its records; the accumulators are clear; h_i is published
    // ---- (4) stage the 30 (instance, point) records ---------------------------------------------------------
    if (MR) {
      if (stager && lane < 32) {
        // one lane per (instance, point): invariants of the point, then the row-side vectors with coefficients folded in
        double* R = rec + (ks * kNQ + qs) * REC;
        const double* hp = hraw + ks * (kNQ * 4) + 4 * qs;
        const double a0 = hp[0], a1 = hp[1], a2 = hp[2];
        const double Fm[3][3] = {{F[0], F[1], F[2]}, {F[3], F[4], F[5]}, {F[6], F[7], F[8]}};
        MRState ms;
        mr_state(Fm, mat.mu10, mat.mu01, mat.kappa, ms);
        const double dV = s0 * m.qw[qs], w = h * dV;
        double pv[3], fa[3], fca[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          pv[i] = ms.G[i][0] * a0 + ms.G[i][1] * a1 + ms.G[i][2] * a2;
          fa[i] = Fm[i][0] * a0 + Fm[i][1] * a1 + Fm[i][2] * a2;
          fca[i] = ms.FC[i][0] * a0 + ms.FC[i][1] * a1 + ms.FC[i][2] * a2;
        }
        const double cX1 = -2.0 / 3.0 * ms.t1 * w, cX2 = -4.0 / 3.0 * ms.t2 * w, cXp = mat.kappa * (2.0 * ms.J - 1.0) * ms.J * w;
        const double cY = (ms.t1 * ms.I1 / 3.0 + ms.t2 * 2.0 * ms.I2 / 3.0 - ms.t3) * w;
        const double cZp = (-2.0 / 3.0 * ms.t1 - 4.0 / 3.0 * ms.t2 * ms.I1) * w;
        const double cU = 2.0 * ms.t2 * w + mat.lamd * dV;   // fa (x) fb: elastic + lambda_d (FEAT10DataFunc.cuh:695-762)
        const double cV = -ms.t2 * w + mat.eta * dV;         // fb (x) fa and (h_i.h_j) F F^T: elastic + eta
        R[0] = a0; R[1] = a1; R[2] = a2;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const double t1a = fa[i] - (ms.I1 / 3.0) * pv[i];
          const double t2a = ms.I1 * fa[i] - fca[i] - (2.0 * ms.I2 / 3.0) * pv[i];
          R[3 + i] = cX1 * t1a + cX2 * t2a + cXp * pv[i];
          R[6 + i] = cY * pv[i];
          R[9 + i] = cZp * pv[i] + cU * fa[i];
          R[12 + i] = (4.0 / 3.0 * ms.t2 * w) * pv[i];
          R[15 + i] = cV * fa[i];
          R[18 + i] = fa[i];
        }
        R[21] = w * (ms.t1 + ms.t2 * ms.I1);
        R[22] = -ms.t2 * w;
        R[23] = 0.0;
        R[24] = cV * ms.FFT[0][0]; R[25] = cV * ms.FFT[0][1]; R[26] = cV * ms.FFT[0][2];
        R[27] = cV * ms.FFT[1][1]; R[28] = cV * ms.FFT[1][2]; R[29] = cV * ms.FFT[2][2];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int jj = 0; jj < 3; jj++) {
            R[30 + 3 * i + jj] = ms.G[i][jj];
            R[39 + 3 * i + jj] = Fm[i][jj];
            R[48 + 3 * i + jj] = ms.FC[i][jj];
          }
      }
    } else if (stager) {
      double* R = rec + (ks * kNQ + qs) * kAdRec;
      if (lane < 32) {
        const double* hp = hraw + ks * (kNQ * 4) + 4 * qs;
        const double2 h01 = *reinterpret_cast<const double2*>(hp);
        const double s0 = h01.x, s1 = h01.y, s2 = hp[2];
        double2* R2 = reinterpret_cast<double2*>(R);
        R2[0] = make_double2(s0, s1);
        R2[1] = make_double2(s2, F[0] * s0 + F[1] * s1 + F[2] * s2);
        R2[2] = make_double2(F[3] * s0 + F[4] * s1 + F[5] * s2, F[6] * s0 + F[7] * s1 + F[8] * s2);
        R2[3] = make_double2(F[0], F[1]);
        R2[4] = make_double2(F[2], F[3]);
        R2[5] = make_double2(F[4], F[5]);
        R2[6] = make_double2(F[6], F[7]);
        R2[7] = make_double2(F[8], 0.0);
      } else {
        const double T00 = F[0] * F[0] + F[1] * F[1] + F[2] * F[2], T01 = F[0] * F[3] + F[1] * F[4] + F[2] * F[5],
                     T02 = F[0] * F[6] + F[1] * F[7] + F[2] * F[8], T11 = F[3] * F[3] + F[4] * F[4] + F[5] * F[5],
                     T12 = F[3] * F[6] + F[4] * F[7] + F[5] * F[8], T22 = F[6] * F[6] + F[7] * F[7] + F[8] * F[8];
        const double trE = 0.5 * (T00 + T11 + T22 - 3.0);
        const double dV = s0 * m.qw[qs];
        // h*K (SVK.cuh:35-55) + C_vis (FEAT10DataFunc.cuh:695-762) share their rank-1 structure:
        const double A1 = dV * (h * mat.lambda + mat.lamd);      // * Fh_i (x) Fh_j
        const double B1 = dV * (h * mat.mu + mat.eta);           // * Fh_j (x) Fh_i  and  * (h_i.h_j) FF^T
        const double C0 = dV * h * (mat.lambda * trE - mat.mu);  // * (h_i.h_j) I
        const double C1 = dV * h * mat.mu;                       // * (Fh_i.Fh_j) I
        double2* R2 = reinterpret_cast<double2*>(R + 16);
        R2[0] = make_double2(B1 * T00, B1 * T01);
        R2[1] = make_double2(B1 * T02, B1 * T11);
        R2[2] = make_double2(B1 * T12, B1 * T22);
        R2[3] = make_double2(A1, B1);
        R2[4] = make_double2(C0, C1);
      }
    }
    wave_sync();
    // ---- (5) this lane's block: sum over the 5 points, add into the row accumulator --------------------------
    double a00 = 0, a01 = 0, a02 = 0, a10 = 0, a11 = 0, a12 = 0, a20 = 0, a21 = 0, a22 = 0;
    double cds = mh * inv_h;  // M/h on the xyz-diagonal (SyncedNewton.cu:214-259), carried by the block's first item
    const double2* Rk = reinterpret_cast<const double2*>(rec + (size_t)min(k, kAdInst - 1) * kNQ * REC);
    // ROLLED: one point per trip, h_j[q] read through the register index (166 VGPRs, 3 waves per SIMD); unrolled the
    // scheduler overlaps the points' LDS reads with arithmetic at 230 VGPRs, 2 waves per SIMD
    const int nq = (store_mode & 512) ? 0 : kNQ;
    if (MR) {
#pragma unroll 1
      for (int q = 0; q < nq; q++) {
        const double* R = rec + ((size_t)min(k, kAdInst - 1) * kNQ + q) * REC;
        const double b0 = hj[q][0], b1 = hj[q][1], b2 = hj[q][2];
        double qv[3], fb[3], fcb[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          qv[i] = R[30 + 3 * i] * b0 + R[31 + 3 * i] * b1 + R[32 + 3 * i] * b2;
          fb[i] = R[39 + 3 * i] * b0 + R[40 + 3 * i] * b1 + R[41 + 3 * i] * b2;
          fcb[i] = R[48 + 3 * i] * b0 + R[49 + 3 * i] * b1 + R[50 + 3 * i] * b2;
        }
        const double ab = R[0] * b0 + R[1] * b1 + R[2] * b2;
        const double fafb = R[18] * fb[0] + R[19] * fb[1] + R[20] * fb[2];
        cds += R[21] * ab + R[22] * fafb;
        const double X0 = R[3], X1 = R[4], X2 = R[5], Y0 = R[6], Y1 = R[7], Y2 = R[8], Z0 = R[9], Z1 = R[10], Z2 = R[11];
        const double W0 = R[12], W1 = R[13], W2 = R[14], V0 = R[15], V1 = R[16], V2 = R[17];
        a00 += X0 * qv[0] + qv[0] * Y0 + Z0 * fb[0] + W0 * fcb[0] + fb[0] * V0 + ab * R[24];
        a01 += X0 * qv[1] + qv[0] * Y1 + Z0 * fb[1] + W0 * fcb[1] + fb[0] * V1 + ab * R[25];
        a02 += X0 * qv[2] + qv[0] * Y2 + Z0 * fb[2] + W0 * fcb[2] + fb[0] * V2 + ab * R[26];
        a10 += X1 * qv[0] + qv[1] * Y0 + Z1 * fb[0] + W1 * fcb[0] + fb[1] * V0 + ab * R[25];
        a11 += X1 * qv[1] + qv[1] * Y1 + Z1 * fb[1] + W1 * fcb[1] + fb[1] * V1 + ab * R[27];
        a12 += X1 * qv[2] + qv[1] * Y2 + Z1 * fb[2] + W1 * fcb[2] + fb[1] * V2 + ab * R[28];
        a20 += X2 * qv[0] + qv[2] * Y0 + Z2 * fb[0] + W2 * fcb[0] + fb[2] * V0 + ab * R[26];
        a21 += X2 * qv[1] + qv[2] * Y1 + Z2 * fb[1] + W2 * fcb[1] + fb[2] * V1 + ab * R[28];
        a22 += X2 * qv[2] + qv[2] * Y2 + Z2 * fb[2] + W2 * fcb[2] + fb[2] * V2 + ab * R[29];
      }
    } else
#pragma unroll(ROLLED ? 1 : kNQ)
    for (int q = 0; q < nq; q++) {
      const double2* R2 = Rk + q * (kAdRec / 2);
      const double2 r0 = R2[0], r1 = R2[1], r2 = R2[2], r3 = R2[3], r4 = R2[4], r5 = R2[5], r6 = R2[6], r7 = R2[7],
                    r8 = R2[8], r9 = R2[9], r10 = R2[10], r11 = R2[11], r12 = R2[12];
      const double hj0 = hj[q][0], hj1 = hj[q][1], hj2 = hj[q][2];
      const double fi0 = r1.y, fi1 = r2.x, fi2 = r2.y;
      const double fj0 = r3.x * hj0 + r3.y * hj1 + r4.x * hj2;
      const double fj1 = r4.y * hj0 + r5.x * hj1 + r5.y * hj2;
      const double fj2 = r6.x * hj0 + r6.y * hj1 + r7.x * hj2;
      const double sv = r0.x * hj0 + r0.y * hj1 + r1.x * hj2;  // h_i . h_j
      const double tv = fi0 * fj0 + fi1 * fj1 + fi2 * fj2;     // F h_i . F h_j
      const double A1 = r11.x, B1 = r11.y;
      cds += r12.x * sv + r12.y * tv;
      const double u0 = A1 * fi0, u1 = A1 * fi1, u2 = A1 * fi2;
      const double w0 = B1 * fj0, w1 = B1 * fj1, w2 = B1 * fj2;
      a00 += u0 * fj0 + w0 * fi0 + sv * r8.x;
      a01 += u0 * fj1 + w0 * fi1 + sv * r8.y;
      a02 += u0 * fj2 + w0 * fi2 + sv * r9.x;
      a10 += u1 * fj0 + w1 * fi0 + sv * r8.y;
      a11 += u1 * fj1 + w1 * fi1 + sv * r9.y;
      a12 += u1 * fj2 + w1 * fi2 + sv * r10.x;
      a20 += u2 * fj0 + w2 * fi0 + sv * r9.x;
      a21 += u2 * fj1 + w2 * fi1 + sv * r10.x;
      a22 += u2 * fj2 + w2 * fi2 + sv * r10.y;
    }
    if (act) {
      double* ap = acc + (pk & 0xffff);
      const int st = (pk >> 16) & 0x7fff;
#define TLFEA_LDS_ADD(p, v) (void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
      TLFEA_LDS_ADD(ap + 0, a00 + cds);
      TLFEA_LDS_ADD(ap + 1, a01);
      TLFEA_LDS_ADD(ap + 2, a02);
      TLFEA_LDS_ADD(ap + st + 0, a10);
      TLFEA_LDS_ADD(ap + st + 1, a11 + cds);
      TLFEA_LDS_ADD(ap + st + 2, a12);
      TLFEA_LDS_ADD(ap + 2 * st + 0, a20);
      TLFEA_LDS_ADD(ap + 2 * st + 1, a21);
      TLFEA_LDS_ADD(ap + 2 * st + 2, a22 + cds);
#undef TLFEA_LDS_ADD
    }
    // ---- (6) last pass of a group: h^2 rho J^T J on pinned rows (SyncedNewton.cu:292-341), rows stream out once ----
    if (last) {
      wave_sync();
      if (lane < nrows && pen != 0.0) {
        const int a0 = ri.x & 0xffff, dpos = ri.x >> 16, row = 3 * ri.z;
#pragma unroll
        for (int d = 0; d < 3; d++) acc[a0 + d * row + 3 * dpos + d] += pen;
      }
      wave_sync();
      for (int r = 0; r < nrows; r++) {
        const int a0 = __shfl(ri.x, r) & 0xffff, off0 = __shfl(ri.y, r), n9 = 9 * __shfl(ri.z, r);
        double* out = Hval + (size_t)9 * off0;
        for (int t = lane; t < n9; t += 64) store_through(out + t, acc[a0 + t], store_mode & 3);
      }
    }
    ecur = enxt;
    enxt = en2;
    vcur = vnxt;
    vnxt = vn2;
    code_i = code_i_n;
    pk = pk_n;
    mb = mb_n;
    code_s = code_s_n;
  }
}

void launch_assemble_direct(hipStream_t s, const ElemView& m, const Material& mat, double h, const RowGroups& rg,
                            const double* Fq, const double* mval, const int* fixed_slot, const double* nw,
                            double penalty, double* Hval) {
  if (mat.model == kMooneyRivlin) {
    // Mooney-Rivlin: the same walk with 60-double records (assemble_direct_kernel<1, 0, 1>); one form, no tuning knobs
    const size_t lds = (size_t)(kAdInst * kNQ * kAdRecMR + kAdHraw + rg.acc_max) * sizeof(double);
    const void* fn = (const void*)assemble_direct_kernel<1, 0, 1>;
    static size_t lds_attr_mr = 0;
    if (lds > 64 * 1024 && lds > lds_attr_mr) {
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      lds_attr_mr = lds;
    }
    static int n_cu_mr = 0, occ_mr = 0;
    static size_t occ_lds_mr = ~(size_t)0;
    if (!n_cu_mr) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&n_cu_mr, hipDeviceAttributeMultiprocessorCount, dev);
      if (n_cu_mr <= 0) n_cu_mr = 256;
    }
    if (occ_lds_mr != lds) {
      int o = 0;
      occ_mr = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, fn, 64, lds) == hipSuccess && o > 0) ? o : 4;
      occ_lds_mr = lds;
    }
    const int per_xcd = std::max(1, std::min((n_cu_mr / 8) * std::min(occ_mr, 8), (rg.G + 7) / 8));
    hipLaunchKernelGGL((assemble_direct_kernel<1, 0, 1>), dim3(8 * per_xcd), dim3(64), lds, s, m, mat, h, rg, Fq, mval, 1.0 / h,
                       fixed_slot, nw, penalty, Hval, 0);
    return;
  }
  const size_t lds = (size_t)(kAdRecTotal + kAdHraw + rg.acc_max) * sizeof(double);
  static size_t lds_attr = 0;
  if (lds > 64 * 1024 && lds > lds_attr) {
    for (const void* f : {(const void*)assemble_direct_kernel<1, 0>, (const void*)assemble_direct_kernel<0, 0>,
                          (const void*)assemble_direct_kernel<1, 1>, (const void*)assemble_direct_kernel<0, 1>})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_attr = lds;
  }
  // persistent grid: as many single-wave workgroups as the chip holds at once (8 XCDs x 32 CUs x resident waves per CU),
  // never more than there are groups; a wave that starts late (fewer resident than assumed) still does its share
  static int n_cu = 0;
  static size_t occ_lds = ~(size_t)0;
  static int occ = 8;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
  }
  if (occ_lds != lds) {
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)assemble_direct_kernel<1, 0>, 64, lds) == hipSuccess && o > 0)
      occ = o;
    occ_lds = lds;
  }
  // experiment switches: TLFEA_AD_ROLLED=0|1 kernel form, TLFEA_AD_WAVES resident waves per CU assumed
  // by the persistent grid; TLFEA_AD_TUNE=1 re-reads both at every launch (tools/tune_assemble.py)
  static const bool tune = std::getenv("TLFEA_AD_TUNE") != nullptr;
  static int rolled = std::getenv("TLFEA_AD_ROLLED") ? std::atoi(std::getenv("TLFEA_AD_ROLLED")) : 1;
  static int occ_env = std::getenv("TLFEA_AD_WAVES") ? std::atoi(std::getenv("TLFEA_AD_WAVES")) : 0;
  static int store_mode = std::getenv("TLFEA_AD_STORE") ? std::atoi(std::getenv("TLFEA_AD_STORE")) : 0;
  if (tune) {
    store_mode = std::getenv("TLFEA_AD_STORE") ? std::atoi(std::getenv("TLFEA_AD_STORE")) : 0;
    rolled = std::getenv("TLFEA_AD_ROLLED") ? std::atoi(std::getenv("TLFEA_AD_ROLLED")) : 1;
    occ_env = std::getenv("TLFEA_AD_WAVES") ? std::atoi(std::getenv("TLFEA_AD_WAVES")) : 0;
  }
  // measured at config C (tools/tune_assemble.py): every resident slot of the rolled form (12 per CU: 168 VGPRs, 12 KiB of
  // LDS), 8 of the unrolled one; more workgroups than fit leave a tail
  const int occ_eff = occ_env > 0 ? occ_env : std::min(occ, rolled ? 12 : 8);
  const int per_xcd = std::max(1, std::min((n_cu / 8) * occ_eff, (rg.G + 7) / 8));
  const bool expm = tune && (store_mode & ~3);  // work-skipping experiments: tools only (TLFEA_AD_TUNE), own instantiation
#define TLFEA_AD_LAUNCH(R, X)                                                                                              \
  hipLaunchKernelGGL((assemble_direct_kernel<R, X>), dim3(8 * per_xcd), dim3(64), lds, s, m, mat, h, rg, Fq, mval, 1.0 / h, \
                     fixed_slot, nw, penalty, Hval, store_mode)
  if (expm) {
    if (rolled) TLFEA_AD_LAUNCH(1, 1); else TLFEA_AD_LAUNCH(0, 1);
  } else {
    if (rolled) TLFEA_AD_LAUNCH(1, 0); else TLFEA_AD_LAUNCH(0, 0);
  }
#undef TLFEA_AD_LAUNCH
}

// ------------------------------------------------------------------------------------------------
// Affine-element form of the fused tangent + assembly (T10, SVK + Kelvin-Voigt).  On a straight-sided T10 element
// grad N_j(q) = sum_n c_jn(q) g_n with the four CONSTANT vertex gradients g_n = grad L_n and coefficients that depend on
// the rule only (corner j = n: 4 L_n(q) - 1; mid-edge j = (a, b): 4 L_b(q) on g_a, 4 L_a(q) on g_b).  The block of
// (row node i, column node j) is bilinear in (grad N_i, grad N_j) at every point, so
//     K_ij = sum_q sum_n c_jn(q) R_n(q),   R_n(q) = block(grad N_i(q), g_n; F(q))
// and with the 5-point Keast rule (L = 1/4 at q0; L_m = 1/2 at point q_m, 1/6 at the other three) the sums over q
// collapse:  with D_p = R_n(q_p), S = D_0 + .. + D_3,
//     column "vertex n"      : 4/3 D_n - 1/3 S
//     column "mid-edge (n,p)": 4/3 D_p + 2/3 S + R_n(q0)        (the share of vertex n; vertex p adds the mirror image)
// One lane per (instance, vertex n): 5 block evaluations and four 3x3 results instead of ten lanes x 5 evaluations --
// 2.2x fewer fp64 operations and 2.5x less LDS traffic per element than assemble_direct_kernel, and grad N (1 200 B per
// element) is never read: the kernel stages per instance the element's 128-byte g record and the 400 bytes the residual
// launch leaves behind -- F at the five points of the rule; F F^T and tr E are formed per lane.
//   pass  = up to 16 instances x 4 lanes; records of the pass in LDS (66 doubles per instance: stride = 4 banks)
//   sum   = ds_add_f64 into the group's row accumulators (H layout) as in assemble_direct_kernel; M/h and the pinned
//           rows' penalty are added while a finished group streams out
// Used when launch_affine_pre finds every element affine and the rule of the expected form (tlfea_api.hip); curved
// elements keep assemble_direct_kernel.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int kAfInst = 16;                 // instances per pass
constexpr int kAfRec = 66;                  // doubles per staged instance: 16 (g, det J) + 5 x 10 (F at the points)
constexpr int kAfStage = kAfInst * kAfRec;  // 1 056 doubles = 8.25 KiB
}  // namespace

__global__ void affine_pre_kernel(ElemView m, AffineView av, double* __restrict__ gvec, double* __restrict__ dev_max) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.E) return;
  const double* gN = m.gradN + (size_t)e * (kNQ * 3 * kNN);
  double g[4][3], gmax = 0.0;
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int d = 0; d < 3; d++) {
      g[p][d] = gN[(av.qv[p] * 3 + d) * kNN + p];  // 4 L_p - 1 = 1 there
      gmax = fmax(gmax, fabs(g[p][d]));
    }
  const int ea[6] = {0, 1, 0, 0, 1, 2}, eb[6] = {1, 2, 2, 3, 3, 3};  // FEAT10Data.cu:143
  double dev = 0.0;
  for (int q = 0; q < kNQ; q++) {
    double L[4];
#pragma unroll
    for (int p = 0; p < 4; p++) L[p] = q == av.q0 ? 0.25 : (q == av.qv[p] ? 0.5 : 1.0 / 6.0);
#pragma unroll
    for (int d = 0; d < 3; d++) {
#pragma unroll
      for (int p = 0; p < 4; p++) dev = fmax(dev, fabs(gN[(q * 3 + d) * kNN + p] - (4.0 * L[p] - 1.0) * g[p][d]));
#pragma unroll
      for (int k = 0; k < 6; k++)
        dev = fmax(dev, fabs(gN[(q * 3 + d) * kNN + 4 + k] - 4.0 * (L[eb[k]] * g[ea[k]][d] + L[ea[k]] * g[eb[k]][d])));
    }
  }
  const double dj = m.detJ[(size_t)e * kNQ];
  double rel = gmax > 0.0 ? dev / gmax : 1.0;
  for (int q = 1; q < kNQ; q++) rel = fmax(rel, fabs(m.detJ[(size_t)e * kNQ + q] - dj) / fabs(dj));
  if (!(rel == rel)) rel = 1.0;
  double* o = gvec + (size_t)e * 16;
#pragma unroll
  for (int p = 0; p < 4; p++) {
    o[4 * p + 0] = g[p][0];
    o[4 * p + 1] = g[p][1];
    o[4 * p + 2] = g[p][2];
    o[4 * p + 3] = p == 0 ? dj : 0.0;
  }
  // positive doubles order like their bit patterns
  atomicMax(reinterpret_cast<unsigned long long*>(dev_max), (unsigned long long)__double_as_longlong(rel));
}

void launch_affine_pre(hipStream_t s, const ElemView& m, const AffineView& av, double* gvec, double* dev_max) {
  hipLaunchKernelGGL(affine_pre_kernel, dim3((m.E + 127) / 128), dim3(128), 0, s, m, av, gvec, dev_max);
}

struct AffineCoef {  // per rule point in the kernel's order (q0, then q_0 .. q_3 = the record slots the residual launch writes)
  double cA[5], cB[5], cC[5], cL[5];  // w_q (h lambda + lamd), w_q (h mu + eta), w_q h mu, w_q h lambda   (x det J)
};

template <int TIMING, int EXP>  // EXP: as assemble_direct_kernel (experiment bits honoured by the tools-only instantiation)
__global__ __launch_bounds__(64, 2) void assemble_affine_kernel(RowGroups4 rg, AffineCoef ac,
                                                                const double* __restrict__ gvec,
                                                                const double* __restrict__ Fq16,
                                                                const double* __restrict__ cmass, double mscale,
                                                                const int* __restrict__ fixed_slot,
                                                                const double* __restrict__ nw, double penalty,
                                                                double* __restrict__ Hval, int store_mode_arg,
                                                                unsigned long long* __restrict__ tdbg) {
  const int store_mode = EXP ? store_mode_arg : (store_mode_arg & 3);
  extern __shared__ __attribute__((aligned(16))) double lds_af[];
  double* stage = lds_af;                  // [kAfInst][kAfRec]
  double* cml = lds_af + kAfStage;         // [10][16] mass coefficients of (row node, vertex n, p) x rho / h
  double* acc = lds_af + kAfStage + 160;   // the group's rows, each in H's layout [d][3 deg]
  // same walk as assemble_direct_kernel: XCD x owns a contiguous range of groups, its resident waves take them side by side
  const int W = gridDim.x >> 3;
  const int Gper = (rg.G + 7) >> 3;
  const int gbeg = (blockIdx.x & 7) * Gper, gend = min(rg.G, gbeg + Gper);
  if (gbeg + (int)(blockIdx.x >> 3) >= gend) return;
  const int lane = threadIdx.x;
  const int k = lane >> 2, n = lane & 3;  // instance k of the pass, vertex n
  for (int t = lane; t < 160; t += 64) cml[t] = cmass[t] * mscale;
  const int last_inst = rg.n_inst - 1;
  // The walk over the pass table is wave-uniform, but its loads inside the loop are VECTOR loads (every lane the same
  // address, made opaque to the compiler by zl) read back with readfirstlane one pass later: a scalar load shares its
  // counter with LDS, so that every LDS wait of a pass would also wait for the table entry fetched at its top.
  const int zl = __builtin_amdgcn_mbcnt_lo(0u, 0u);  // 0 in every lane
  auto uni4 = [](const int4& v) {
    return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y),
                     __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w));
  };
  int lg = gbeg + (blockIdx.x >> 3);
  int lp = rg.g_pass_off[lg], lpe = rg.g_pass_off[lg + 1];
  int npx = 0, npy = 0;  // pass range of the wave's next group (vector registers, fetched one group ahead)
  if (lg + W < gend) {
    npx = rg.g_pass_off[lg + W + zl];
    npy = rg.g_pass_off[lg + W + 1 + zl];
  }
  bool lvalid = true;
  auto advance = [&]() {
    if (lp + 1 < lpe) {
      lp++;
      return;
    }
    lg += W;
    lvalid = lg < gend;
    lp = __builtin_amdgcn_readfirstlane(npx);
    lpe = __builtin_amdgcn_readfirstlane(npy);
    if (lg + W < gend) {
      npx = rg.g_pass_off[lg + W + zl];
      npy = rg.g_pass_off[lg + W + 1 + zl];
    }
  };
  // pass entries: current, +1, +2 (the lead cursor runs three passes ahead)
  int4 ecur = rg.pt[lp];
  advance();
  bool vnxt = lvalid;
  int4 enxt = rg.pt[vnxt ? lp : 0];
  advance();
  bool vn2 = lvalid;
  int4 en2 = rg.pt[vn2 ? lp : 0];
  advance();
  auto inst_of = [&](const int4& en) { return min(en.x + min(k, max((en.y & 31) - 1, 0)), last_inst); };
  int2 head, ent, head_n, ent_n;
  {
    const int ii = inst_of(ecur), in = inst_of(enxt);
    head = rg.gi_head[ii];
    ent = rg.gi_ent[(size_t)ii * 4 + n];
    head_n = rg.gi_head[in];
    ent_n = rg.gi_ent[(size_t)in * 4 + n];
  }
  // row records travel one pass ahead of the group they belong to; the pinned flag and the weight of a row are fetched
  // at the group's first pass (independent loads) and used at its last
  int4 ri = make_int4(0, 0, 0, 0), ri_n = make_int4(0, 0, 0, 0);
  if (lane < (ecur.y >> 8)) ri_n = rg.gr_info[ecur.z + lane];
  int fs = -1;
  double wv = 1.0;
  double* Sk = stage + k * kAfRec;
  // records in flight: the four lanes of an instance fetch its 528 bytes (g record + F at the five points) 16 at a
  // time, one pass ahead of the pass that computes, so that a pass never waits for a round trip to memory
  double2 sa, sb, s0, s1, s2, s3, s4, s5, s6 = make_double2(0.0, 0.0);
  auto fetch = [&](const int2& hd) __attribute__((always_inline)) {
    const int e = (store_mode & 256) ? 0 : hd.x / kNN;
    const double2* gp = reinterpret_cast<const double2*>(gvec + (size_t)e * 16) + n;
    const double2* fp = reinterpret_cast<const double2*>(Fq16 + (size_t)e * 50) + n;
    sa = gp[0]; sb = gp[4];
    s0 = fp[0]; s1 = fp[4]; s2 = fp[8]; s3 = fp[12]; s4 = fp[16]; s5 = fp[20];
    if (n == 0) s6 = fp[24];  // 25 16-byte pieces: the last one has one taker
  };
  fetch(head);

  // TIMING: shader-clock cycles per phase, summed over the wave's passes (tools only; perturbs the schedule a little)
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = 0, npass = 0;
#define TLFEA_TICK(i)                                                     \
  if (TIMING) {                                                           \
    const unsigned long long tn = __builtin_readcyclecounter();           \
    tph[i] += tn - tlast;                                                 \
    tlast = tn;                                                           \
  }
  if (TIMING) tlast = __builtin_readcyclecounter();
  bool vcur = true;
#pragma unroll 1
  while (vcur) {
    const bool vn3 = lvalid;
    const int4 en3v = rg.pt[(vn3 ? lp : 0) + zl];
    advance();
    const int cnt = ecur.y & 31, nrows = ecur.y >> 8;
    const bool first = (ecur.y & 32) != 0, last = (ecur.y & 64) != 0;
    const int il = head.x - kNN * (head.x / kNN);
    // ---- (1) indices of the pass after the next ---------------------------------------------------------------------
    int2 head_nn, ent_nn;
    {
      const int ii = inst_of(en2);
      head_nn = rg.gi_head[ii];
      ent_nn = rg.gi_ent[(size_t)ii * 4 + n];
    }
    // ---- (2) first pass of a group: clear its accumulators, fetch its row records ---------------------------------
    if (first) {
      for (int t = lane; t < ecur.w; t += 64) acc[t] = 0.0;
      ri = ri_n;
      fs = -1;
      wv = 1.0;
      if (lane < nrows && fixed_slot) {
        fs = fixed_slot[ri.w];
        if (nw) wv = nw[ri.w];
      }
    }
    if (vnxt && (enxt.y & 32) && lane < (enxt.y >> 8)) ri_n = rg.gr_info[enxt.z + lane];
    TLFEA_TICK(0)  // pass table, index loads, accumulator clear
    wave_sync();  // the previous pass has read its records
    // ---- (3) this pass's records into LDS, the next pass's records into flight ------------------------------------
    {
      double2* S2 = reinterpret_cast<double2*>(Sk) + n;
      S2[0] = sa; S2[4] = sb;
      S2[8] = s0; S2[12] = s1; S2[16] = s2; S2[20] = s3; S2[24] = s4; S2[28] = s5;
      if (n == 0) S2[32] = s6;
    }
    wave_sync();
    TLFEA_TICK(1)  // records into LDS (waits for the loads of the previous pass)
    // ---- (4) the lane's five block evaluations --------------------------------------------------------------------
    // row node: vertex A (corner il) or edge (A, B) (mid-edge il) -- FEAT10Data.cu:143 packed 2 bits per entry
    const int A = (0x904e4 >> (2 * il)) & 3;   // {0,1,2,3,0,1,0,0,1,2}
    const int B = (0xfe9e4 >> (2 * il)) & 3;   // {0,1,2,3,1,2,2,3,3,3}
    const bool mid = il >= 4;
    double gA[3], gB[3], gn[3], detJ;
    {
      const double2 a01 = *reinterpret_cast<const double2*>(Sk + 4 * A), b01 = *reinterpret_cast<const double2*>(Sk + 4 * B),
                    n01 = *reinterpret_cast<const double2*>(Sk + 4 * n);
      gA[0] = a01.x; gA[1] = a01.y; gA[2] = Sk[4 * A + 2];
      gB[0] = b01.x; gB[1] = b01.y; gB[2] = Sk[4 * B + 2];
      gn[0] = n01.x; gn[1] = n01.y; gn[2] = Sk[4 * n + 2];
      detJ = Sk[3];
    }
    TLFEA_TICK(2)  // next pass's loads issued, g / mass coefficients read
    double R0[9], D[4][9];
    // F of the point comes from the staged record; F F^T, tr E and with them B1 F F^T and C0 are formed here: 24 operations
    // per point instead of three more 16-byte reads and 48 more staged bytes per point and instance.
    // D[r] is evaluated at the point where vertex (n + r) & 3 has L = 1/2 -- a lane-RELATIVE order (the four outer points
    // of the Keast rule carry one weight, checked by the launcher), so that the share of the mid-edge column (n, n + r)
    // sits in the same register of every lane and the two lanes of such a column can be paired by a quad permute below.
    auto block = [&](const int sdx, double (&out)[9]) __attribute__((always_inline)) {
      const double2* R2 = reinterpret_cast<const double2*>(Sk + 16 + 10 * sdx);
      const double2 r0 = R2[0], r1 = R2[1], r2 = R2[2], r3 = R2[3], r4 = R2[4];
      const double F0 = r0.x, F1 = r0.y, F2 = r1.x, F3 = r1.y, F4 = r2.x, F5 = r2.y, F6 = r3.x, F7 = r3.y, F8 = r4.x;
      if (store_mode & 512) {  // timing experiment: no block arithmetic
#pragma unroll
        for (int t = 0; t < 9; t++) out[t] = 0.0;
        return;
      }
      const double T00 = F0 * F0 + F1 * F1 + F2 * F2, T01 = F0 * F3 + F1 * F4 + F2 * F5, T02 = F0 * F6 + F1 * F7 + F2 * F8,
                   T11 = F3 * F3 + F4 * F4 + F5 * F5, T12 = F3 * F6 + F4 * F7 + F5 * F8, T22 = F6 * F6 + F7 * F7 + F8 * F8;
      const double trE = 0.5 * (T00 + T11 + T22 - 3.0);
      // grad N_i at this point = al g_A + be g_B
      double al, be;
      if (sdx == 0) {
        al = mid ? 1.0 : 0.0;
        be = al;
      } else {
        const int p = sdx - 1;
        al = mid ? (B == p ? 2.0 : 2.0 / 3.0) : (A == p ? 1.0 : -1.0 / 3.0);
        be = mid ? (A == p ? 2.0 : 2.0 / 3.0) : 0.0;
      }
      const int ci = sdx == 0 ? 0 : 1;  // the centroid point | the four outer points (one weight)
      const double h0 = al * gA[0] + be * gB[0], h1 = al * gA[1] + be * gB[1], h2 = al * gA[2] + be * gB[2];
      const double fi0 = F0 * h0 + F1 * h1 + F2 * h2, fi1 = F3 * h0 + F4 * h1 + F5 * h2, fi2 = F6 * h0 + F7 * h1 + F8 * h2;
      const double b0 = F0 * gn[0] + F1 * gn[1] + F2 * gn[2], b1 = F3 * gn[0] + F4 * gn[1] + F5 * gn[2],
                   b2 = F6 * gn[0] + F7 * gn[1] + F8 * gn[2];
      const double sv = h0 * gn[0] + h1 * gn[1] + h2 * gn[2];  // grad N_i . g_n
      const double tv = fi0 * b0 + fi1 * b1 + fi2 * b2;        // F grad N_i . F g_n
      const double A1 = detJ * (ci ? ac.cA[1] : ac.cA[0]), B1 = detJ * (ci ? ac.cB[1] : ac.cB[0]),
                   C1 = detJ * (ci ? ac.cC[1] : ac.cC[0]);
      const double C0 = detJ * (ci ? ac.cL[1] : ac.cL[0]) * trE - C1;  // dV h (lambda tr E - mu)   (SVK.cuh:35-55)
      const double cd = C0 * sv + C1 * tv;
      const double sb1 = sv * B1;                              // (grad N_i . g_n) B1 F F^T
      const double u0 = A1 * fi0, u1 = A1 * fi1, u2 = A1 * fi2;
      const double w0 = B1 * b0, w1 = B1 * b1, w2 = B1 * b2;
      out[0] = u0 * b0 + w0 * fi0 + sb1 * T00 + cd;
      out[1] = u0 * b1 + w0 * fi1 + sb1 * T01;
      out[2] = u0 * b2 + w0 * fi2 + sb1 * T02;
      out[3] = u1 * b0 + w1 * fi0 + sb1 * T01;
      out[4] = u1 * b1 + w1 * fi1 + sb1 * T11 + cd;
      out[5] = u1 * b2 + w1 * fi2 + sb1 * T12;
      out[6] = u2 * b0 + w2 * fi0 + sb1 * T02;
      out[7] = u2 * b1 + w2 * fi1 + sb1 * T12;
      out[8] = u2 * b2 + w2 * fi2 + sb1 * T22 + cd;
    };
    block(0, R0);
    block(1 + n, D[0]);
    block(1 + ((n + 1) & 3), D[1]);
    block(1 + ((n + 2) & 3), D[2]);
    block(1 + ((n + 3) & 3), D[3]);
    // the next pass's records: issued after the block evaluations (36 registers less while they run; measured no slower
    // than before them): the adds, the rows streaming out and the next pass's top cover the round trip
    fetch(head_n);
    // M/h (SyncedNewton.cu:214-259): on an affine element M_e(i, j) = rho det J sum_q w_q N_i N_j (FEAT10Data.cu:206-278)
    // is det J times a constant of the rule; the two lanes of a mid-edge column carry half of it each
    double cmv[4];  // relative order: column vertex n, mid-edge (n, n + 1), (n, n + 2), (n, n + 3)
    {
      const double* cq = cml + il * 16 + 4 * n;
#pragma unroll
      for (int r = 0; r < 4; r++) cmv[r] = detJ * cq[(n + r) & 3];
    }
    TLFEA_TICK(3)  // five block evaluations
    // ---- (5) the lane's blocks into the row accumulator ------------------------------------------------------------
    // column "vertex n": 4/3 D_n - 1/3 S; column "mid-edge (n, p)": (4/3 D_p + 2/3 S + R0) of lane n PLUS the mirror image
    // of lane p.  The two shares are paired in registers -- lane n takes lane n + 1's share of edge (n, n + 1) and, for
    // n < 2, lane n + 2's share of edge (n, n + 2), by quad permutes -- so an instance costs 10 accumulator blocks
    // (27 ds_add_f64 per lane at most) instead of 16 (36).
    {
      double S1[9];
#pragma unroll
      for (int t = 0; t < 9; t++) S1[t] = (D[0][t] + D[1][t]) + (D[2][t] + D[3][t]);
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const double dg = (t == 0 || t == 4 || t == 8) ? 1.0 : 0.0;
        const double m2 = (2.0 / 3.0) * S1[t] + R0[t];
        D[0][t] = (4.0 / 3.0) * D[0][t] - (1.0 / 3.0) * S1[t] + dg * cmv[0];
        D[1][t] = (4.0 / 3.0) * D[1][t] + m2 + dg * cmv[1];
        D[2][t] = (4.0 / 3.0) * D[2][t] + m2 + dg * cmv[2];
        D[3][t] = (4.0 / 3.0) * D[3][t] + m2 + dg * cmv[3];
      }
      // quad permutes: lane n reads lane (n + 1) & 3 (ctrl 0x39 = [1,2,3,0]) / lane (n + 2) & 3 (0x4E = [2,3,0,1])
      auto from_next = [](double v) __attribute__((always_inline)) {
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x39, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x39, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
      };
      auto from_opposite = [](double v) __attribute__((always_inline)) {
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x4E, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x4E, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
      };
#pragma unroll
      for (int t = 0; t < 9; t++) {
        D[1][t] += from_next(D[3][t]);      // lane n + 1's share towards n sits in ITS D[3]
        D[2][t] += from_opposite(D[2][t]);  // lane n + 2's share towards n sits in ITS D[2]
      }
    }
    if (k < cnt && !(store_mode & 1024)) {
      const int stride = head.y;
      const unsigned long long e64 = ((unsigned long long)(unsigned)ent.y << 32) | (unsigned)ent.x;
#define TLFEA_LDS_ADD(p, v) (void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#pragma unroll
      for (int r = 0; r < 3; r++) {
        if (r == 2 && n >= 2) break;        // edges (0, 2) and (1, 3) are added by lanes 0 and 1
        const int pcol = (n + r) & 3;
        const int word = (int)((e64 >> (16 * pcol)) & 0xffffull);
        double* ap = acc + 3 * word;
        if (store_mode & 4096) ap = acc + 3 * (lane + 64 * r);  // timing experiment: no two lanes add to one address
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
          for (int c = 0; c < 3; c++) TLFEA_LDS_ADD(ap + rr * stride + c, D[r][3 * rr + c]);
      }
#undef TLFEA_LDS_ADD
    }
    TLFEA_TICK(4)  // LDS adds
    // ---- (6) last pass of a group: h^2 rho J^T J on pinned rows (SyncedNewton.cu:292-341), rows stream out once ----
    if (last) {
      wave_sync();
      if (lane < nrows && fs >= 0) {
        const double pen = wv * penalty;
        const int a0 = ri.x & 0xffff, dpos = ri.x >> 16, row = 3 * ri.z;
#pragma unroll
        for (int d = 0; d < 3; d++) acc[a0 + d * row + 3 * dpos + d] += pen;
      }
      wave_sync();
      for (int r = 0; r < ((store_mode & 2048) ? 0 : nrows); r++) {
        const int a0 = __builtin_amdgcn_readlane(ri.x, r) & 0xffff, off0 = __builtin_amdgcn_readlane(ri.y, r),
                  n9 = 9 * __builtin_amdgcn_readlane(ri.z, r);
        double* outp = Hval + (size_t)9 * off0;
        for (int t = lane; t < n9; t += 64) store_through(outp + t, acc[a0 + t], store_mode & 3);
      }
    }
    TLFEA_TICK(5)  // rows out
    npass++;
    ecur = enxt;
    enxt = en2;
    en2 = uni4(en3v);
    vcur = vnxt;
    vnxt = vn2;
    vn2 = vn3;
    head = head_n;
    ent = ent_n;
    head_n = head_nn;
    ent_n = ent_nn;
  }
  if (TIMING && lane == 0) {
    for (int i = 0; i < 6; i++) atomicAdd(tdbg + i, tph[i]);
    atomicAdd(tdbg + 6, npass);
    atomicAdd(tdbg + 7, 1ULL);
  }
#undef TLFEA_TICK
}

void launch_assemble_affine(hipStream_t s, const ElemView& m, const Material& mat, double h, const RowGroups4& rg,
                            const AffineView& av, const double* Fq16, const double* cmass, double rho0,
                            const int* fixed_slot, const double* nw, double penalty, double* Hval) {
  const void* fn = (const void*)assemble_affine_kernel<0, 0>;
  const size_t lds = (size_t)(kAfStage + 160 + rg.acc_max) * sizeof(double);
  static size_t lds_attr = 0;
  if (lds > 64 * 1024 && lds > lds_attr) {
    for (const void* f : {fn, (const void*)assemble_affine_kernel<0, 1>, (const void*)assemble_affine_kernel<1, 1>})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_attr = lds;
  }
  static int n_cu = 0;
  static size_t occ_lds = ~(size_t)0;
  static int occ = 8;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
  }
  if (occ_lds != lds) {
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, fn, 64, lds) == hipSuccess && o > 0) occ = o;
    occ_lds = lds;
  }
  static const bool tune = std::getenv("TLFEA_AD_TUNE") != nullptr;
  static int occ_env = std::getenv("TLFEA_AD_WAVES") ? std::atoi(std::getenv("TLFEA_AD_WAVES")) : 0;
  static int store_mode = std::getenv("TLFEA_AD_STORE") ? std::atoi(std::getenv("TLFEA_AD_STORE")) : 0;
  if (tune) {
    store_mode = std::getenv("TLFEA_AD_STORE") ? std::atoi(std::getenv("TLFEA_AD_STORE")) : 0;
    occ_env = std::getenv("TLFEA_AD_WAVES") ? std::atoi(std::getenv("TLFEA_AD_WAVES")) : 0;
  }
  const int occ_eff = occ_env > 0 ? occ_env : occ;
  const int per_xcd = std::max(1, std::min((n_cu / 8) * occ_eff, (rg.G + 7) / 8));
  AffineCoef ac;
  for (int sdx = 0; sdx < 5; sdx++) {
    const int q = sdx == 0 ? av.q0 : av.qv[sdx - 1];
    ac.cA[sdx] = m.qw[q] * (h * mat.lambda + mat.lamd);
    ac.cB[sdx] = m.qw[q] * (h * mat.mu + mat.eta);
    ac.cC[sdx] = m.qw[q] * h * mat.mu;
    ac.cL[sdx] = m.qw[q] * h * mat.lambda;
  }
  static const bool timing = std::getenv("TLFEA_AF_TIMING") != nullptr;  // tools only: per-phase shader clocks on stderr
  if (timing) {
    static unsigned long long* d_t = nullptr;
    if (!d_t && hipMalloc(&d_t, 8 * sizeof(unsigned long long)) != hipSuccess) return;
    (void)hipMemsetAsync(d_t, 0, 8 * sizeof(unsigned long long), s);
    hipLaunchKernelGGL((assemble_affine_kernel<1, 1>), dim3(8 * per_xcd), dim3(64), lds, s, rg, ac, av.gvec, Fq16, cmass, rho0 / h,
                       fixed_slot, nw, penalty, Hval, store_mode, d_t);
    unsigned long long t[8];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(t, d_t, sizeof(t), hipMemcpyDeviceToHost);
    const double np = (double)std::max(1ULL, t[6]);
    std::fprintf(stderr, "assemble_affine timing: %llu waves, %llu passes; cycles per pass: top %.0f | stage %.0f | g+issue %.0f | "
                 "blocks %.0f | adds %.0f | rows out %.0f\n", t[7], t[6], t[0] / np, t[1] / np, t[2] / np, t[3] / np, t[4] / np,
                 t[5] / np);
    return;
  }
  if (tune && (store_mode & ~3))  // work-skipping experiments: tools only (TLFEA_AD_TUNE), own instantiation
    hipLaunchKernelGGL((assemble_affine_kernel<0, 1>), dim3(8 * per_xcd), dim3(64), lds, s, rg, ac, av.gvec, Fq16, cmass,
                       rho0 / h, fixed_slot, nw, penalty, Hval, store_mode, nullptr);
  else
    hipLaunchKernelGGL((assemble_affine_kernel<0, 0>), dim3(8 * per_xcd), dim3(64), lds, s, rg, ac, av.gvec, Fq16, cmass,
                       rho0 / h, fixed_slot, nw, penalty, Hval, store_mode, nullptr);
}

// ------------------------------------------------------------------------------------------------
// consistent mass (FEAT10Data.cu:206-278), row-owner form: thread per node row, fixed order
// ------------------------------------------------------------------------------------------------
__global__ void mass_values_kernel(ElemView m, Incidence inc, const double* __restrict__ qx,
                                   const double* __restrict__ qy, const double* __restrict__ qz, double rho0,
                                   double* __restrict__ mval) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.N) return;
  const int edges[6][2] = {{0, 1}, {1, 2}, {0, 2}, {0, 3}, {1, 3}, {2, 3}};
  double Nq[kNQ][kNN];
#pragma unroll
  for (int q = 0; q < kNQ; q++) {
    const double L[4] = {1.0 - qx[q] - qy[q] - qz[q], qx[q], qy[q], qz[q]};
#pragma unroll
    for (int k = 0; k < 4; k++) Nq[q][k] = L[k] * (2.0 * L[k] - 1.0);
#pragma unroll
    for (int k = 0; k < 6; k++) Nq[q][k + 4] = 4.0 * L[edges[k][0]] * L[edges[k][1]];
  }
  const int off0 = inc.off[i];
  for (int k = off0; k < inc.off[i + 1]; k++) mval[k] = 0.0;
  for (int k = inc.n2e_off[i]; k < inc.n2e_off[i + 1]; k++) {
    const int code = inc.n2e[k];
    const int e = code / 10, il = code - e * 10;
    const int* pos = inc.n2e_pos + (size_t)k * 10;
    for (int j = 0; j < kNN; j++) {
      double s = 0.0;
      for (int q = 0; q < kNQ; q++) s += rho0 * Nq[q][il] * Nq[q][j] * m.detJ[e * kNQ + q] * m.qw[q];
      mval[off0 + pos[j]] += s;
    }
  }
}

void launch_mass_values(hipStream_t s, const ElemView& m, const Incidence& inc, const double* qx, const double* qy,
                        const double* qz, double rho0, double* mval) {
  hipLaunchKernelGGL(mass_values_kernel, dim3((m.N + 127) / 128), dim3(128), 0, s, m, inc, qx, qy, qz, rho0, mval);
}

// ------------------------------------------------------------------------------------------------
// SyncedVBDSolver: one coloured Gauss-Seidel update (vbd_update_color_block_kernel + vbd_update_pos_from_vel_color +
// the compute_p refresh, SyncedVBD.cu:163-400, 708-722, 1310-1330).  One 64-lane wave per node of the colour; the
// lanes split the node's (incident element, quadrature point) items.  The reference keeps F and P of every
// (element, point) in memory and recomputes ALL of them after every colour group; here each item rebuilds F (and Fdot)
// from the current coordinates in registers.  That is the same arithmetic at the same state: two nodes of one colour
// (or of one colour group) never share an element, so between the reference's refresh and its use no coordinate or
// velocity of that element changes -- and nothing has to be refreshed, stored or re-read.
//   residual  R = M_row (v - v_prev)/h + sum_q P h_a dV - f_ext  (+ h (lam + rho c) on a pinned node)
//   Hessian   H = m_ii/h I + h sum_q K_aa(elastic)               (+ h^2 rho I), symmetrised, + eps max(1, tr H) I
//   v_i += omega * (-H^-1 R) (cofactor inverse, zero update when |det| < 1e-30);  x_i = x_prev_i + h v_i
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void vbd_diag_block(const double F[3][3], const double ha[3], const Material& mat, double w,
                                               double K[3][3]) {
  if (mat.model == kMooneyRivlin) {
    // K[i][k] = sum_jl A[i][j][k][l] h_j h_l with A = dP/dF of MooneyRivlin.cuh:113-225 (FEAT10DataFunc.cuh:353-372),
    // contracted analytically: every term of A is a product of two 3x3 factors, so with a = h_a, Fa = F a, Ga = G a
    // (G = F^-T), FCa = (F C) a, s = |a|^2, q = a.C a:
    //   K = t1 [ s I - 2/3 (T1a (x) Ga + Ga (x) Fa) + I1/3 Ga (x) Ga ]
    //     + t2 [ Fa (x) Fa + (I1 s - q) I - s F F^T - 4/3 (T2a (x) Ga + Ga (x) (I1 Fa - FCa)) + 2 I2/3 Ga (x) Ga ]
    //     + (kappa (2J - 1) J - t3) Ga (x) Ga,        T1a = Fa - I1/3 Ga,  T2a = I1 Fa - FCa - 2 I2/3 Ga
    MRState s;
    mr_state(F, mat.mu10, mat.mu01, mat.kappa, s);
    const double k3 = mat.kappa * (2.0 * s.J - 1.0) * s.J;
    double Fa[3], Ga[3], FCa[3], Ca[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      Fa[i] = F[i][0] * ha[0] + F[i][1] * ha[1] + F[i][2] * ha[2];
      Ga[i] = s.G[i][0] * ha[0] + s.G[i][1] * ha[1] + s.G[i][2] * ha[2];
      FCa[i] = s.FC[i][0] * ha[0] + s.FC[i][1] * ha[1] + s.FC[i][2] * ha[2];
      Ca[i] = s.C[i][0] * ha[0] + s.C[i][1] * ha[1] + s.C[i][2] * ha[2];
    }
    const double ss = ha[0] * ha[0] + ha[1] * ha[1] + ha[2] * ha[2], qq = ha[0] * Ca[0] + ha[1] * Ca[1] + ha[2] * Ca[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double T1a = Fa[i] - (s.I1 / 3.0) * Ga[i], T2a = s.I1 * Fa[i] - FCa[i] - (2.0 * s.I2 / 3.0) * Ga[i];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double dik = (i == k) ? 1.0 : 0.0;
        const double GG = Ga[i] * Ga[k];
        const double e1 = ss * dik - (2.0 / 3.0) * (T1a * Ga[k] + Ga[i] * Fa[k]) + (s.I1 / 3.0) * GG;
        const double e2 = Fa[i] * Fa[k] + (s.I1 * ss - qq) * dik - ss * s.FFT[i][k] -
                          (4.0 / 3.0) * (T2a * Ga[k] + Ga[i] * (s.I1 * Fa[k] - FCa[k])) + (2.0 * s.I2 / 3.0) * GG;
        K[i][k] = (s.t1 * e1 + s.t2 * e2 + (k3 - s.t3) * GG) * w;
      }
    }
    return;
  }
  double FFT[3][3], Fh[3], trC = 0.0;  // SVK.cuh:35-55 with i == j
#pragma unroll
  for (int i = 0; i < 3; i++) {
    Fh[i] = F[i][0] * ha[0] + F[i][1] * ha[1] + F[i][2] * ha[2];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      trC += F[i][j] * F[i][j];
      FFT[i][j] = F[i][0] * F[j][0] + F[i][1] * F[j][1] + F[i][2] * F[j][2];
    }
  }
  const double trE = 0.5 * (trC - 3.0), hij = ha[0] * ha[0] + ha[1] * ha[1] + ha[2] * ha[2];
  const double ff = Fh[0] * Fh[0] + Fh[1] * Fh[1] + Fh[2] * Fh[2];
#pragma unroll
  for (int d = 0; d < 3; d++)
#pragma unroll
    for (int e = 0; e < 3; e++) {
      const double dl = (d == e) ? 1.0 : 0.0;
      K[d][e] = (mat.lambda * Fh[d] * Fh[e] + mat.lambda * trE * hij * dl + mat.mu * ff * dl + mat.mu * Fh[d] * Fh[e] +
                 mat.mu * hij * FFT[d][e] - mat.mu * hij * dl) * w;
    }
}

template <int S, int Q, int LPN, int MODEL, bool DAMP>
__global__ __launch_bounds__(256) void vbd_color_kernel(ElemView m, Material mat_in, Incidence inc,
                                                        const int* __restrict__ nodes, int count,
                                                        const double* __restrict__ mval, const double* __restrict__ f_ext,
                                                        const int* __restrict__ fixed_slot, const double* __restrict__ xt,
                                                        const double* __restrict__ yt, const double* __restrict__ zt,
                                                        const double* __restrict__ lam, double h, double rho, double omega,
                                                        double hess_eps, const double* __restrict__ v_prev,
                                                        const double* __restrict__ xp, const double* __restrict__ yp,
                                                        const double* __restrict__ zp, double* v, double* x, double* y,
                                                        double* z, const int* __restrict__ conn_rm, double* xyz) {
  // LPN lanes per node (16 | 32 | 64, chosen per colour from its nodes' item counts: mid-edge nodes of a tet mesh
  // have 20-40 items, corner nodes 100+); a group's lanes stay together through the shuffles below
  const int lane = threadIdx.x & (LPN - 1);
  const int slot = blockIdx.x * (256 / LPN) + threadIdx.x / LPN;
  if (slot >= count) return;  // whole group
  const int i = nodes[slot];
  const double inv_h = 1.0 / h;
  Material mat = mat_in;
  mat.model = MODEL;  // compile-time material and damping: the other branch's registers (42 doubles of Mooney-Rivlin
  constexpr bool damp = DAMP;  // state, the 9 of Fdot) are not reserved -- occupancy is what hides the gathers here
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.0;
  // mass row (consistent mass, all neighbours)
  for (int k = inc.off[i] + lane; k < inc.off[i + 1]; k += LPN) {
    const int j = inc.cols[k];
    const double mij = mval[k];
#pragma unroll
    for (int d = 0; d < 3; d++) acc[d] += mij * (v[3 * j + d] - v_prev[3 * j + d]) * inv_h;
  }
  // element items
  const int i0 = inc.n2e_off[i], n_items = (inc.n2e_off[i + 1] - i0) * Q;
  for (int w = lane; w < n_items; w += LPN) {
    const int k = w / Q, q = w - k * Q;
    const int code = inc.n2e[i0 + k];
    const int e = code / S, a = code - e * S;
    const double* g = m.gradN + ((size_t)e * Q + q) * 3 * S;
    double F[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Fd[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll 2
    for (int b = 0; b < S; b++) {
      // element-major connectivity and interleaved coordinates (kept by the solver next to the engine's column-major /
      // SoA arrays): one cache line per element and per node instead of S and 3
      const int c = conn_rm[(size_t)e * S + b];
      const double X[3] = {xyz[3 * c], xyz[3 * c + 1], xyz[3 * c + 2]};
      const double hb[3] = {g[b], g[S + b], g[2 * S + b]};
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) F[r][cc] += X[r] * hb[cc];
      if (damp) {
        const double V[3] = {v[3 * c], v[3 * c + 1], v[3 * c + 2]};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int cc = 0; cc < 3; cc++) Fd[r][cc] += V[r] * hb[cc];
      }
    }
    double P[3][3];
    elastic_P(F, mat, P);
    if (damp) {  // Kelvin-Voigt part of compute_p (FEAT10DataFunc.cuh:137-232)
      double Ed[3][3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
          double a1 = 0.0, a2 = 0.0;
#pragma unroll
          for (int t = 0; t < 3; t++) {
            a1 += Fd[t][r] * F[t][cc];
            a2 += F[t][r] * Fd[t][cc];
          }
          Ed[r][cc] = 0.5 * (a1 + a2);
        }
      const double trEd = Ed[0][0] + Ed[1][1] + Ed[2][2];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
          double sv = 0.0;
#pragma unroll
          for (int t = 0; t < 3; t++) sv += F[r][t] * (2.0 * mat.eta * Ed[t][cc] + (t == cc ? mat.lamd * trEd : 0.0));
          P[r][cc] += sv;
        }
    }
    const double ha[3] = {g[a], g[S + a], g[2 * S + a]};
    const double dV = m.detJ[(size_t)e * Q + q] * m.qw[q];
#pragma unroll
    for (int r = 0; r < 3; r++) acc[r] += (P[r][0] * ha[0] + P[r][1] * ha[1] + P[r][2] * ha[2]) * dV;
    double K[3][3];
    vbd_diag_block(F, ha, mat, h * dV, K);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) acc[3 + 3 * r + cc] += K[r][cc];
  }
#pragma unroll
  for (int o = LPN / 2; o > 0; o >>= 1)
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] += __shfl_xor(acc[k], o);
  if (lane != 0) return;
  double R[3], H[3][3];
  const double mii = mval[inc.off[i] + inc.diagpos[i]];  // InitializeMassDiagBlocks: m_ii I (SyncedVBD.cu:1030-1085)
#pragma unroll
  for (int d = 0; d < 3; d++) {
    R[d] = acc[d] - f_ext[3 * i + d];
#pragma unroll
    for (int e = 0; e < 3; e++) H[d][e] = (d == e ? mii * inv_h : 0.0) + acc[3 + 3 * d + e];
  }
  const double vi[3] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
  const double xpi[3] = {xp[i], yp[i], zp[i]};
  const int k = fixed_slot ? fixed_slot[i] : -1;
  if (k >= 0) {
    const double Xi[3] = {xt[i], yt[i], zt[i]};
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const double c = (xpi[d] + h * vi[d]) - Xi[d];
      R[d] += h * (lam[3 * k + d] + rho * c);
      H[d][d] += h * h * rho;
    }
  }
  const double a01 = 0.5 * (H[0][1] + H[1][0]), a02 = 0.5 * (H[0][2] + H[2][0]), a12 = 0.5 * (H[1][2] + H[2][1]);
  H[0][1] = H[1][0] = a01;
  H[0][2] = H[2][0] = a02;
  H[1][2] = H[2][1] = a12;
  const double eps = hess_eps * fmax(1.0, H[0][0] + H[1][1] + H[2][2]);
  H[0][0] += eps;
  H[1][1] += eps;
  H[2][2] += eps;
  const double det = H[0][0] * (H[1][1] * H[2][2] - H[1][2] * H[2][1]) - H[0][1] * (H[1][0] * H[2][2] - H[1][2] * H[2][0]) +
                     H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0]);
  double dv[3] = {0.0, 0.0, 0.0};
  if (fabs(det) >= 1e-30) {  // solve_3x3_vbd (SyncedVBD.cu:47-82)
    const double id = 1.0 / det;
    const double i00 = (H[1][1] * H[2][2] - H[1][2] * H[2][1]) * id, i01 = (H[0][2] * H[2][1] - H[0][1] * H[2][2]) * id,
                 i02 = (H[0][1] * H[1][2] - H[0][2] * H[1][1]) * id, i10 = (H[1][2] * H[2][0] - H[1][0] * H[2][2]) * id,
                 i11 = (H[0][0] * H[2][2] - H[0][2] * H[2][0]) * id, i12 = (H[0][2] * H[1][0] - H[0][0] * H[1][2]) * id,
                 i20 = (H[1][0] * H[2][1] - H[1][1] * H[2][0]) * id, i21 = (H[0][1] * H[2][0] - H[0][0] * H[2][1]) * id,
                 i22 = (H[0][0] * H[1][1] - H[0][1] * H[1][0]) * id;
    dv[0] = -(i00 * R[0] + i01 * R[1] + i02 * R[2]);
    dv[1] = -(i10 * R[0] + i11 * R[1] + i12 * R[2]);
    dv[2] = -(i20 * R[0] + i21 * R[1] + i22 * R[2]);
  }
  const double vn[3] = {vi[0] + omega * dv[0], vi[1] + omega * dv[1], vi[2] + omega * dv[2]};
  v[3 * i] = vn[0];
  v[3 * i + 1] = vn[1];
  v[3 * i + 2] = vn[2];
  const double xn[3] = {xpi[0] + vn[0] * h, xpi[1] + vn[1] * h, xpi[2] + vn[2] * h};  // vbd_update_pos_from_vel_color
  x[i] = xn[0];
  y[i] = xn[1];
  z[i] = xn[2];
  xyz[3 * i] = xn[0];
  xyz[3 * i + 1] = xn[1];
  xyz[3 * i + 2] = xn[2];
}

__global__ void interleave_xyz_kernel(int N, const double* __restrict__ x, const double* __restrict__ y,
                                      const double* __restrict__ z, double* __restrict__ xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  xyz[3 * i] = x[i];
  xyz[3 * i + 1] = y[i];
  xyz[3 * i + 2] = z[i];
}
void launch_interleave_xyz(hipStream_t s, int N, const double* x, const double* y, const double* z, double* xyz) {
  hipLaunchKernelGGL(interleave_xyz_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, x, y, z, xyz);
}

template <int S, int Q, int MODEL, bool DAMP>
static void launch_vbd_color_t(hipStream_t s, int lanes, const ElemView& m, const Material& mat, const Incidence& inc,
                               const int* nodes, int count, const double* mval, const double* f_ext, const int* fixed_slot,
                               const double* xt, const double* yt, const double* zt, const double* lam, double h, double rho,
                               double omega, double hess_eps, const double* v_prev, const double* xp, const double* yp,
                               const double* zp, double* v, double* x, double* y, double* z, const int* conn_rm,
                               double* xyz) {
  const dim3 block(256), grid((count + 256 / lanes - 1) / (256 / lanes));
#define TLFEA_VBD_ARGS m, mat, inc, nodes, count, mval, f_ext, fixed_slot, xt, yt, zt, lam, h, rho, omega, hess_eps, v_prev, xp, yp, zp, v, x, y, z, conn_rm, xyz
  if (lanes == 16) hipLaunchKernelGGL((vbd_color_kernel<S, Q, 16, MODEL, DAMP>), grid, block, 0, s, TLFEA_VBD_ARGS);
  else if (lanes == 32) hipLaunchKernelGGL((vbd_color_kernel<S, Q, 32, MODEL, DAMP>), grid, block, 0, s, TLFEA_VBD_ARGS);
  else hipLaunchKernelGGL((vbd_color_kernel<S, Q, 64, MODEL, DAMP>), grid, block, 0, s, TLFEA_VBD_ARGS);
#undef TLFEA_VBD_ARGS
}

template <int S, int Q>
static void launch_vbd_color_m(hipStream_t s, int lanes, const ElemView& m, const Material& mat, const Incidence& inc,
                               const int* nodes, int count, const double* mval, const double* f_ext, const int* fixed_slot,
                               const double* xt, const double* yt, const double* zt, const double* lam, double h, double rho,
                               double omega, double hess_eps, const double* v_prev, const double* xp, const double* yp,
                               const double* zp, double* v, double* x, double* y, double* z, const int* conn_rm,
                               double* xyz) {
  const bool damp = mat.eta != 0.0 || mat.lamd != 0.0;
#define TLFEA_VBD_FWD s, lanes, m, mat, inc, nodes, count, mval, f_ext, fixed_slot, xt, yt, zt, lam, h, rho, omega, hess_eps, v_prev, xp, yp, zp, v, x, y, z, conn_rm, xyz
  if (mat.model == kMooneyRivlin) {
    if (damp) launch_vbd_color_t<S, Q, kMooneyRivlin, true>(TLFEA_VBD_FWD);
    else launch_vbd_color_t<S, Q, kMooneyRivlin, false>(TLFEA_VBD_FWD);
  } else {
    if (damp) launch_vbd_color_t<S, Q, kSVK, true>(TLFEA_VBD_FWD);
    else launch_vbd_color_t<S, Q, kSVK, false>(TLFEA_VBD_FWD);
  }
#undef TLFEA_VBD_FWD
}

void launch_vbd_color(hipStream_t s, int lanes, const ElemView& m, const Material& mat, const Incidence& inc,
                      const int* nodes, int count, const double* mval, const double* f_ext, const int* fixed_slot,
                      const double* xt, const double* yt, const double* zt, const double* lam, double h, double rho,
                      double omega, double hess_eps, const double* v_prev, const double* xp, const double* yp,
                      const double* zp, double* v, double* x, double* y, double* z, const int* conn_rm, double* xyz) {
  if (count <= 0) return;
#define TLFEA_VBD_FWD s, lanes, m, mat, inc, nodes, count, mval, f_ext, fixed_slot, xt, yt, zt, lam, h, rho, omega, hess_eps, v_prev, xp, yp, zp, v, x, y, z, conn_rm, xyz
  if (m.S == 10) launch_vbd_color_m<10, 5>(TLFEA_VBD_FWD);
  else if (m.S == 8) launch_vbd_color_m<8, 12>(TLFEA_VBD_FWD);
  else launch_vbd_color_m<16, 48>(TLFEA_VBD_FWD);
#undef TLFEA_VBD_FWD
}

}  // namespace tlfea
