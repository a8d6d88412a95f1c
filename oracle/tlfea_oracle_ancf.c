/* tlfea_oracle_ancf.c -- CPU ORACLE (test infrastructure only), part 2: ANCF-3243 beam and ANCF-3443 shell,
 * plus the element-type-generic restatement of the path (S shape functions, Q quadrature points per element)
 * that both share with T10.  Reference: lib_src/elements/ANCF3243Data{.cu,.cuh,Func.cuh},
 * ANCF3443Data{.cu,.cuh,Func.cuh}, lib_utils/cpu_utils.cc:125-420, lib_src/solvers/SyncedNewton.cu.
 *
 * Generic layouts (reference device layouts; "coef" = one 3-vector of generalized coordinates):
 *   conn   int  [S][E]      coefficient ids per element (ANCF: 4*node + slot, ANCF3243DataFunc.cuh:212-215)
 *   gradN  dbl  [E][Q][3][S]  S x 3 column-major per (e,q)   (ANCF3243Data.cuh:40-47)
 *   detJ   dbl  [E][Q] ,  qw dbl [Q] = product of the three 1-D weights (ANCF3243DataFunc.cuh:432-434)
 * Pinned by tests/golden/ancf3243_*.npz / ancf3443_*.npz (generated from the reference's NumPy prototypes),
 * data/utest/mass_matrix_{2,3}_beam.csv, and by agreement with the pinned T10 oracle at S=10, Q=5.
 * The prototypes hold no tangent: K_e of the ANCF types is pinned by finite differences of f_int + symmetry. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tlfea_oracle.h"

#define SMAX 16

/* ---- monomial bases: b_k = u^a v^b w^c ---------------------------------------------------------- */
/* 3243: [1,u,v,w,uv,uw,u^2,u^3] (ANCF3243DataFunc.cuh:115-125); 3443: 16 terms (ANCF3443DataFunc.cuh:114-132) */
static const int kExp3243[8][3] = {{0,0,0},{1,0,0},{0,1,0},{0,0,1},{1,1,0},{1,0,1},{2,0,0},{3,0,0}};
static const int kExp3443[16][3] = {{0,0,0},{1,0,0},{0,1,0},{0,0,1},{1,1,0},{1,0,1},{0,1,1},{1,1,1},
                                    {2,0,0},{0,2,0},{2,1,0},{1,2,0},{3,0,0},{0,3,0},{3,1,0},{1,3,0}};

static double ipow(double x, int n) {
  double r = 1.0;
  for (int i = 0; i < n; i++) r *= x;
  return r;
}
/* b and its derivatives wrt (u,v,w) at a point; which: 0 value, 1 d/du, 2 d/dv, 3 d/dw */
static void basis_row(int S, const int (*ex)[3], double u, double v, double w, int which, double *out) {
  for (int k = 0; k < S; k++) {
    int a = ex[k][0], b = ex[k][1], c = ex[k][2];
    double val;
    if (which == 0) val = ipow(u, a) * ipow(v, b) * ipow(w, c);
    else if (which == 1) val = a ? a * ipow(u, a - 1) * ipow(v, b) * ipow(w, c) : 0.0;
    else if (which == 2) val = b ? b * ipow(u, a) * ipow(v, b - 1) * ipow(w, c) : 0.0;
    else val = c ? c * ipow(u, a) * ipow(v, b) * ipow(w, c - 1) : 0.0;
    out[k] = val;
  }
}

static int invert_dense(int n, double *A /*row-major, overwritten*/, double *Ainv) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ainv[i * n + j] = (i == j);
  for (int k = 0; k < n; k++) {
    int p = k;
    for (int i = k + 1; i < n; i++)
      if (fabs(A[i * n + k]) > fabs(A[p * n + k])) p = i;
    if (fabs(A[p * n + k]) < 1e-300) return 1;
    if (p != k)
      for (int j = 0; j < n; j++) {
        double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t;
        t = Ainv[k * n + j]; Ainv[k * n + j] = Ainv[p * n + j]; Ainv[p * n + j] = t;
      }
    double d = 1.0 / A[k * n + k];
    for (int j = 0; j < n; j++) { A[k * n + j] *= d; Ainv[k * n + j] *= d; }
    for (int i = 0; i < n; i++) {
      if (i == k) continue;
      double f = A[i * n + k];
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++) { A[i * n + j] -= f * A[k * n + j]; Ainv[i * n + j] -= f * Ainv[k * n + j]; }
    }
  }
  return 0;
}

/* B_inv = (B^T)^-1, COLUMN-major S x S as the reference stores it (cpu_utils.cc:125-188, 211-420).
 * kind: 3243 or 3443.  Rows of B: (b, b_u, b_v, b_w) at each node's reference point. */
int orc_ancf_B_inv(int kind, double L, double W, double H, double *B_inv_colmajor) {
  (void)H;
  const int S = kind == 3243 ? 8 : 16;
  const int (*ex)[3] = kind == 3243 ? kExp3243 : kExp3443;
  const int nn = S / 4;
  double pts[4][3];
  if (kind == 3243) {
    pts[0][0] = -L / 2; pts[0][1] = 0; pts[0][2] = 0;
    pts[1][0] = L / 2; pts[1][1] = 0; pts[1][2] = 0;
  } else { /* P1(-,-) P2(+,-) P3(+,+) P4(-,+)  (cpu_utils.cc:213-217) */
    const double sx[4] = {-1, 1, 1, -1}, sy[4] = {-1, -1, 1, 1};
    for (int n = 0; n < 4; n++) { pts[n][0] = sx[n] * L / 2; pts[n][1] = sy[n] * W / 2; pts[n][2] = 0; }
  }
  double B[SMAX * SMAX], BT[SMAX * SMAX], inv[SMAX * SMAX];
  for (int n = 0; n < nn; n++)
    for (int wch = 0; wch < 4; wch++) basis_row(S, ex, pts[n][0], pts[n][1], pts[n][2], wch, B + (4 * n + wch) * S);
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) BT[i * S + j] = B[j * S + i];
  if (invert_dense(S, BT, inv)) return 1;
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) B_inv_colmajor[j * S + i] = inv[i * S + j];
  return 0;
}

static void solve3(const double A[3][3], const double b[3], double x[3]) { /* ANCF3243DataFunc.cuh:30-88 */
  double m[3][4];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m[i][j] = A[i][j]; m[i][3] = b[i]; }
  for (int k = 0; k < 3; k++) {
    int p = k; double best = fabs(m[k][k]);
    for (int i = k + 1; i < 3; i++) if (fabs(m[i][k]) > best) { best = fabs(m[i][k]); p = i; }
    if (p != k) for (int j = 0; j < 4; j++) { double t = m[k][j]; m[k][j] = m[p][j]; m[p][j] = t; }
    if (fabs(m[k][k]) < 1e-14) { x[0] = x[1] = x[2] = 0.0; return; }
    for (int i = k + 1; i < 3; i++) { double f = m[i][k] / m[k][k]; for (int j = k; j < 4; j++) m[i][j] -= f * m[k][j]; }
  }
  x[2] = m[2][3] / m[2][2];
  x[1] = (m[1][3] - m[1][2] * x[2]) / m[1][1];
  x[0] = (m[0][3] - m[0][2] * x[2] - m[0][1] * x[1]) / m[0][0];
}

/* ds/d(xi,eta,zeta) = B_inv * db/d(xi,...) with u = L xi/2 etc. (ANCF3243Data.cu:122-148) */
static void ds_dxi(int S, const int (*ex)[3], const double *Binv_cm, double L, double W, double H, double xi,
                   double eta, double zeta, double ds[3][SMAX]) {
  double db[3][SMAX];
  const double u = L * xi / 2, v = W * eta / 2, w = H * zeta / 2;
  const double sc[3] = {L / 2, W / 2, H / 2};
  for (int d = 0; d < 3; d++) {
    basis_row(S, ex, u, v, w, d + 1, db[d]);
    for (int k = 0; k < S; k++) db[d][k] *= sc[d];
    for (int i = 0; i < S; i++) {
      double s = 0.0;
      for (int j = 0; j < S; j++) s += Binv_cm[j * S + i] * db[d][j];
      ds[d][i] = s;
    }
  }
}

/* precompute_reference_kernel (ANCF3243Data.cu:102-198, ANCF3443Data.cu:96-182): reference geometry from the
 * x12_jac coefficients, force quadrature rule (nq[0] x nq[1] x nq[2], qp = (ixi*nq1 + ieta)*nq2 + izeta). */
void orc_ancf_precompute(int kind, int E, const int *conn /*[S][E]*/, const double *xj, const double *yj,
                         const double *zj, const double *Lv, const double *Wv, const double *Hv,
                         const double *Binv /*[E][S*S] col-major*/, const int *nq, const double *gx,
                         const double *gy, const double *gz, double *gradN, double *detJ_out) {
  const int S = kind == 3243 ? 8 : 16, Q = nq[0] * nq[1] * nq[2];
  const int (*ex)[3] = kind == 3243 ? kExp3243 : kExp3443;
  for (int e = 0; e < E; e++)
    for (int q = 0; q < Q; q++) {
      const int ix = q / (nq[1] * nq[2]), ie = (q / nq[2]) % nq[1], iz = q % nq[2];
      double ds[3][SMAX];
      ds_dxi(S, ex, Binv + (size_t)e * S * S, Lv[e], Wv[e], Hv[e], gx[ix], gy[ie], gz[iz], ds);
      double J[3][3] = {{0}};
      for (int a = 0; a < S; a++) {
        const int c = conn[(size_t)a * E + e];
        const double X[3] = {xj[c], yj[c], zj[c]};
        for (int i = 0; i < 3; i++)
          for (int d = 0; d < 3; d++) J[i][d] += X[i] * ds[d][a];
      }
      detJ_out[(size_t)e * Q + q] = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) -
                                    J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                                    J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      double JT[3][3];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) JT[i][j] = J[j][i];
      double *g = gradN + ((size_t)e * Q + q) * 3 * S;
      for (int a = 0; a < S; a++) {
        double rhs[3] = {ds[0][a], ds[1][a], ds[2][a]}, gr[3];
        solve3(JT, rhs, gr);
        for (int d = 0; d < 3; d++) g[a + S * d] = gr[d];
      }
    }
}

/* mass_matrix_qp_kernel (ANCF3243Data.cu:200-288, ANCF3443Data.cu:184-254): mass rule nqm, s = B_inv b,
 * detJ of the reference map at the mass points; scalar mass between coefficients, summed into the CSR. */
static int bsearch_i(const int *cols, int n, int t) {
  int l = 0, r = n - 1;
  while (l <= r) { int m = l + ((r - l) >> 1); if (cols[m] == t) return m; if (cols[m] < t) l = m + 1; else r = m - 1; }
  return -1;
}
void orc_ancf_mass_values(int kind, int E, const int *conn, const double *xj, const double *yj, const double *zj,
                          const double *Lv, const double *Wv, const double *Hv, const double *Binv, const int *nqm,
                          const double *gx, const double *gy, const double *gz, const double *wx, const double *wy,
                          const double *wz, double rho0, const int *off, const int *cols, double *vals) {
  const int S = kind == 3243 ? 8 : 16, Qm = nqm[0] * nqm[1] * nqm[2];
  const int (*ex)[3] = kind == 3243 ? kExp3243 : kExp3443;
  for (int e = 0; e < E; e++) {
    const double *Bi = Binv + (size_t)e * S * S;
    for (int q = 0; q < Qm; q++) {
      const int ix = q / (nqm[1] * nqm[2]), ie = (q / nqm[2]) % nqm[1], iz = q % nqm[2];
      const double wgt = wx[ix] * wy[ie] * wz[iz];
      double b[SMAX], s[SMAX], ds[3][SMAX];
      basis_row(S, ex, Lv[e] * gx[ix] / 2, Wv[e] * gy[ie] / 2, Hv[e] * gz[iz] / 2, 0, b);
      for (int i = 0; i < S; i++) { double a = 0; for (int j = 0; j < S; j++) a += Bi[j * S + i] * b[j]; s[i] = a; }
      ds_dxi(S, ex, Bi, Lv[e], Wv[e], Hv[e], gx[ix], gy[ie], gz[iz], ds);
      double J[3][3] = {{0}};
      for (int a = 0; a < S; a++) {
        const int c = conn[(size_t)a * E + e];
        const double X[3] = {xj[c], yj[c], zj[c]};
        for (int i = 0; i < 3; i++) for (int d = 0; d < 3; d++) J[i][d] += X[i] * ds[d][a];
      }
      const double detJ = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                          J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++) {
          const int gi = conn[(size_t)i * E + e], gj = conn[(size_t)j * E + e];
          const int k = bsearch_i(cols + off[gi], off[gi + 1] - off[gi], gj);
          if (k >= 0) vals[off[gi] + k] += rho0 * s[i] * s[j] * wgt * detJ;
        }
    }
  }
}

/* ================================ generic (S,Q) element path ==================================== */
extern void orc_free(void *);

/* materials: same arithmetic as the T10 file (SVK.cuh, MooneyRivlin.cuh) */
static double det3(const double A[3][3]) {
  return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
         A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
}
static void invT3(const double A[3][3], double detA, double o[3][3]) {
  const double eps = 1e-12; double sd = detA;
  if (fabs(sd) < eps) sd = (sd >= 0.0) ? eps : -eps;
  double id = 1.0 / sd;
  o[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) * id; o[0][1] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) * id;
  o[0][2] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) * id; o[1][0] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id;
  o[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id; o[1][2] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id;
  o[2][0] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id; o[2][1] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id;
  o[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
}
typedef struct { double C[3][3], FC[3][3], FFT[3][3], G[3][3], I1, I2, J, t1, t2, t3; } mrs;
static void mr_pro(const double F[3][3], const orc_material *m, mrs *s) {
  memset(s, 0, sizeof(*s));
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) { s->C[i][j] += F[k][i] * F[k][j]; s->FFT[i][j] += F[i][k] * F[j][k]; }
  s->I1 = s->C[0][0] + s->C[1][1] + s->C[2][2];
  double tr = 0; for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) tr += s->C[i][k] * s->C[k][i];
  s->I2 = 0.5 * (s->I1 * s->I1 - tr);
  s->J = det3(F); invT3(F, s->J, s->G);
  double J13 = cbrt(s->J), Jm23 = 1.0 / (J13 * J13), Jm43 = Jm23 * Jm23;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) s->FC[i][j] += F[i][k] * s->C[k][j];
  s->t1 = 2.0 * m->mu10 * Jm23; s->t2 = 2.0 * m->mu01 * Jm43; s->t3 = m->kappa * (s->J - 1.0) * s->J;
}
static void elastic_P(const double F[3][3], const orc_material *m, double P[3][3]) {
  if (m->model == ORC_MAT_MOONEY_RIVLIN) {
    mrs s; mr_pro(F, m, &s);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      double t1 = F[i][j] - (s.I1 / 3.0) * s.G[i][j], t2 = s.I1 * F[i][j] - s.FC[i][j] - (2.0 * s.I2 / 3.0) * s.G[i][j];
      P[i][j] = s.t1 * t1 + s.t2 * t2 + s.t3 * s.G[i][j];
    }
  } else {
    double FtF[3][3] = {{0}}, FFt[3][3] = {{0}}, FFtF[3][3] = {{0}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) { FtF[i][j] += F[k][i] * F[k][j]; FFt[i][j] += F[i][k] * F[j][k]; }
    double tr = FtF[0][0] + FtF[1][1] + FtF[2][2];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) FFtF[i][j] += FFt[i][k] * F[k][j];
    double lf = m->lambda * (0.5 * tr - 1.5);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) P[i][j] = lf * F[i][j] + m->mu * (FFtF[i][j] - F[i][j]);
  }
}

/* compute_p for any element type (ANCF3243DataFunc.cuh:189-395, ANCF3443DataFunc.cuh:281-497) */
void orc_gen_compute_p(int S, int Q, int E, const int *conn, const double *x, const double *y, const double *z,
                       const double *v, const double *gradN, const orc_material *mat, double *Fo, double *Po) {
  const int damp = v && (mat->eta_damp != 0.0 || mat->lambda_damp != 0.0);
  for (int e = 0; e < E; e++)
    for (int q = 0; q < Q; q++) {
      const double *g = gradN + ((size_t)e * Q + q) * 3 * S;
      double F[3][3] = {{0}}, Fd[3][3] = {{0}};
      for (int a = 0; a < S; a++) {
        const int c = conn[(size_t)a * E + e];
        const double X[3] = {x[c], y[c], z[c]};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) F[i][j] += X[i] * g[a + S * j];
        if (damp) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Fd[i][j] += v[3 * c + i] * g[a + S * j];
      }
      double P[3][3];
      elastic_P(F, mat, P);
      if (damp) {
        double Ed[3][3], S2[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double a1 = 0, a2 = 0; for (int k = 0; k < 3; k++) { a1 += Fd[k][i] * F[k][j]; a2 += F[k][i] * Fd[k][j]; } Ed[i][j] = 0.5 * (a1 + a2); }
        double tr = Ed[0][0] + Ed[1][1] + Ed[2][2];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S2[i][j] = 2.0 * mat->eta_damp * Ed[i][j] + (i == j ? mat->lambda_damp * tr : 0.0);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += F[i][k] * S2[k][j]; P[i][j] += s; }
      }
      size_t o = ((size_t)e * Q + q) * 9;
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { if (Fo) Fo[o + i + 3 * j] = F[i][j]; Po[o + i + 3 * j] = P[i][j]; }
    }
}

void orc_gen_internal_force(int S, int Q, int E, int N, const int *conn, const double *P, const double *gradN,
                            const double *detJ, const double *qw, double *f_int) {
  memset(f_int, 0, sizeof(double) * 3 * (size_t)N);
  for (int e = 0; e < E; e++)
    for (int a = 0; a < S; a++) {
      double f[3] = {0, 0, 0};
      for (int q = 0; q < Q; q++) {
        const double *Pq = P + ((size_t)e * Q + q) * 9, *g = gradN + ((size_t)e * Q + q) * 3 * S;
        const double dV = detJ[(size_t)e * Q + q] * qw[q];
        for (int i = 0; i < 3; i++) f[i] += (Pq[i] * g[a] + Pq[i + 3] * g[a + S] + Pq[i + 6] * g[a + 2 * S]) * dV;
      }
      const int c = conn[(size_t)a * E + e];
      for (int i = 0; i < 3; i++) f_int[3 * c + i] += f[i];
    }
}

/* A = dP/dF of compressible Mooney-Rivlin (MooneyRivlin.cuh:113-225) */
static void mr_tangent_tensor(const double F[3][3], const orc_material *mat, double A[3][3][3][3]) {
  mrs s; mr_pro(F, mat, &s);
  double T1[3][3], T2[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T1[i][j] = F[i][j] - (s.I1 / 3.0) * s.G[i][j]; T2[i][j] = s.I1 * F[i][j] - s.FC[i][j] - (2.0 * s.I2 / 3.0) * s.G[i][j]; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) {
    double dik = i == k, djl = j == l, dG = -s.G[i][l] * s.G[k][j];
    double dt1 = (-2.0 / 3.0) * s.t1 * s.G[k][l], dt2 = (-4.0 / 3.0) * s.t2 * s.G[k][l], dt3 = mat->kappa * (2.0 * s.J - 1.0) * s.J * s.G[k][l];
    double dT1 = dik * djl - (2.0 / 3.0) * F[k][l] * s.G[i][j] + (s.I1 / 3.0) * s.G[i][l] * s.G[k][j];
    double dT2 = 2.0 * F[k][l] * F[i][j] + s.I1 * dik * djl - (dik * s.C[l][j] + F[i][l] * F[k][j] + djl * s.FFT[i][k]) -
                 (4.0 / 3.0) * (s.I1 * F[k][l] - s.FC[k][l]) * s.G[i][j] + (2.0 * s.I2 / 3.0) * s.G[i][l] * s.G[k][j];
    A[i][j][k][l] = dt1 * T1[i][j] + s.t1 * dT1 + dt2 * T2[i][j] + s.t2 * dT2 + dt3 * s.G[i][j] + s.t3 * dG;
  }
}

/* K_e and C_e (row-major 3S x 3S) summed over the Q points: SVK.cuh:35-55 / MooneyRivlin.cuh:113-225 and the
 * Kelvin-Voigt block of ANCF3243DataFunc.cuh:842-915 (same expression as FEAT10DataFunc.cuh:695-762). */
static void qp_tangent(int S, const double *xn /*[S][3]*/, const double *g, double dV, const orc_material *mat,
                       double *K, double *C, int want_vis) {
  const int n = 3 * S;
  double F[3][3] = {{0}}, FFT[3][3] = {{0}}, Fh[SMAX][3];
  for (int a = 0; a < S; a++) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) F[i][j] += xn[3 * a + i] * g[a + S * j];
  double trC = 0; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) trC += F[i][j] * F[i][j];
  const double trE = 0.5 * (trC - 3.0);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) FFT[i][j] += F[i][k] * F[j][k];
  for (int a = 0; a < S; a++) for (int r = 0; r < 3; r++) { Fh[a][r] = 0; for (int c = 0; c < 3; c++) Fh[a][r] += F[r][c] * g[a + S * c]; }
  double A[3][3][3][3];
  const int use_mr = mat->model == ORC_MAT_MOONEY_RIVLIN;
  if (use_mr) mr_tangent_tensor(F, mat, A);
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) {
      double hij = 0, ff = 0;
      for (int c = 0; c < 3; c++) { hij += g[j + S * c] * g[i + S * c]; ff += Fh[j][c] * Fh[i][c]; }
      for (int d = 0; d < 3; d++)
        for (int e = 0; e < 3; e++) {
          double val;
          if (use_mr) {
            double sum = 0;
            for (int J = 0; J < 3; J++) for (int L = 0; L < 3; L++) sum += A[d][J][e][L] * g[i + S * J] * g[j + S * L];
            val = sum * dV;
          } else {
            double dl = d == e;
            val = (mat->lambda * Fh[i][d] * Fh[j][e] + mat->lambda * trE * hij * dl + mat->mu * ff * dl + mat->mu * Fh[j][d] * Fh[i][e] +
                   mat->mu * hij * FFT[d][e] - mat->mu * hij * dl) * dV;
          }
          K[(size_t)(3 * i + d) * n + 3 * j + e] = val;
          if (want_vis)
            C[(size_t)(3 * i + d) * n + 3 * j + e] =
                (mat->eta_damp * Fh[j][d] * Fh[i][e] + mat->eta_damp * FFT[d][e] * hij + mat->lambda_damp * Fh[i][d] * Fh[j][e]) * dV;
        }
    }
}

void orc_gen_element_tangent(int S, int Q, int e, int E, const int *conn, const double *x, const double *y,
                             const double *z, const double *gradN, const double *detJ, const double *qw,
                             const orc_material *mat, double *Ke, double *Ce) {
  const int n = 3 * S;
  double xn[3 * SMAX];
  for (int a = 0; a < S; a++) { const int c = conn[(size_t)a * E + e]; xn[3 * a] = x[c]; xn[3 * a + 1] = y[c]; xn[3 * a + 2] = z[c]; }
  double *K = (double *)malloc(sizeof(double) * n * n), *C = (double *)malloc(sizeof(double) * n * n);
  memset(Ke, 0, sizeof(double) * n * n);
  if (Ce) memset(Ce, 0, sizeof(double) * n * n);
  for (int q = 0; q < Q; q++) {
    qp_tangent(S, xn, gradN + ((size_t)e * Q + q) * 3 * S, detJ[(size_t)e * Q + q] * qw[q], mat, K, C, Ce != NULL);
    for (int i = 0; i < n * n; i++) { Ke[i] += K[i]; if (Ce) Ce[i] += C[i]; }
  }
  free(K); free(C);
}

static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return (x > y) - (x < y); }
/* BuildMassCSRPattern for any element type (ANCF3243Data.cu:45-98,419-452): sorted unique coefficient pairs */
int orc_gen_mass_pattern(int S, int E, int N, const int *conn, int *offsets, int **columns) {
  size_t total = (size_t)E * S * S, nnz = 0;
  uint64_t *keys = (uint64_t *)malloc(total * sizeof(uint64_t));
  for (int e = 0; e < E; e++) for (int i = 0; i < S; i++) for (int j = 0; j < S; j++)
    keys[((size_t)e * S + i) * S + j] = ((uint64_t)(uint32_t)conn[(size_t)i * E + e] << 32) | (uint32_t)conn[(size_t)j * E + e];
  qsort(keys, total, sizeof(uint64_t), cmp_u64);
  for (size_t k = 0; k < total; k++) if (k == 0 || keys[k] != keys[k - 1]) keys[nnz++] = keys[k];
  int *cols = (int *)malloc(nnz * sizeof(int));
  memset(offsets, 0, sizeof(int) * ((size_t)N + 1));
  for (size_t k = 0; k < nnz; k++) { cols[k] = (int)(keys[k] & 0xffffffffULL); offsets[(keys[k] >> 32) + 1]++; }
  for (int i = 0; i < N; i++) offsets[i + 1] += offsets[i];
  free(keys);
  *columns = cols;
  return (int)nnz;
}

/* M/h + h K + C_vis on a DOF-level pattern (ro, ci) expanded from the coefficient adjacency (ao, ac); the mass
 * CSR (mo, mc, mv) may be a subset of that adjacency (constraint-aware pattern)  (SyncedNewton.cu:214-290) */
static void gen_assemble_core(int S, int Q, int E, int N, const int *conn, const double *x, const double *y,
                              const double *z, const double *gradN, const double *detJ, const double *qw,
                              const orc_material *mat, const int *mo, const int *mc, const double *mv,
                              const int *ao, const int *ac, double h, const int *ro, const int *ci, double *val) {
  const int n = 3 * S;
  memset(val, 0, sizeof(double) * (size_t)ro[3 * N]);
  for (int ni = 0; ni < N; ni++)
    for (int k = mo[ni]; k < mo[ni + 1]; k++)
      for (int d = 0; d < 3; d++) {
        const int row = 3 * ni + d, col = 3 * mc[k] + d;
        const int p = bsearch_i(ci + ro[row], ro[row + 1] - ro[row], col);
        if (p >= 0) val[ro[row] + p] += mv[k] / h;
      }
  const int want_vis = mat->eta_damp != 0.0 || mat->lambda_damp != 0.0;
  double *K = (double *)malloc(sizeof(double) * n * n), *C = (double *)malloc(sizeof(double) * n * n);
  for (int e = 0; e < E; e++) {
    int gn[SMAX], pos[SMAX][SMAX];
    double xn[3 * SMAX];
    for (int a = 0; a < S; a++) { gn[a] = conn[(size_t)a * E + e]; xn[3 * a] = x[gn[a]]; xn[3 * a + 1] = y[gn[a]]; xn[3 * a + 2] = z[gn[a]]; }
    for (int a = 0; a < S; a++) for (int b = 0; b < S; b++) pos[a][b] = bsearch_i(ac + ao[gn[a]], ao[gn[a] + 1] - ao[gn[a]], gn[b]);
    for (int q = 0; q < Q; q++) {
      qp_tangent(S, xn, gradN + ((size_t)e * Q + q) * 3 * S, detJ[(size_t)e * Q + q] * qw[q], mat, K, C, want_vis);
      for (int a = 0; a < S; a++) for (int d = 0; d < 3; d++) {
        double *rv = val + ro[3 * gn[a] + d];
        for (int b = 0; b < S; b++) for (int c = 0; c < 3; c++) {
          if (pos[a][b] < 0) continue;
          rv[3 * pos[a][b] + c] += h * K[(size_t)(3 * a + d) * n + 3 * b + c];
          if (want_vis) rv[3 * pos[a][b] + c] += C[(size_t)(3 * a + d) * n + 3 * b + c];
        }
      }
    }
  }
  free(K); free(C);
}

/* H = M/h + h K + C_vis + h^2 rho J^T J for fixed-coefficient constraints (SyncedNewton.cu:214-341) */
void orc_gen_assemble_hessian(int S, int Q, int E, int N, const int *conn, const double *x, const double *y,
                              const double *z, const double *gradN, const double *detJ, const double *qw,
                              const orc_material *mat, const int *mo, const int *mc, const double *mv,
                              const int *fixed, int n_fixed, double h, double rho, const int *ro, const int *ci,
                              double *val) {
  gen_assemble_core(S, Q, E, N, conn, x, y, z, gradN, detJ, qw, mat, mo, mc, mv, mo, mc, h, ro, ci, val);
  for (int k = 0; k < 3 * n_fixed; k++) {
    const int dof = fixed[k / 3] * 3 + k % 3;
    const int p = bsearch_i(ci + ro[dof], ro[dof + 1] - ro[dof], dof);
    if (p >= 0) val[ro[dof] + p] += h * h * rho;
  }
}

/* ---- general linear constraints  c = J x - rhs  (kConstraintLinearCSR) ------------------------------------
 * J: CSR over constraint rows, columns = 3*coef + component (ANCF3243Data.cuh:803-940). */

/* Constraint-aware coefficient adjacency (SyncedNewton.cu:571-630): {i} + mass row i + every pair of coefficients
 * that share a constraint row; sorted unique.  Returns nnz, *columns malloc'ed. */
int orc_lin_adjacency(int N, const int *mo, const int *mc, int nc, const int *joff, const int *jcol, int *offsets,
                      int **columns) {
  size_t total = (size_t)N + (size_t)mo[N];
  for (int r = 0; r < nc; r++) { size_t m = (size_t)(joff[r + 1] - joff[r]); total += m * m; }
  uint64_t *keys = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
  size_t n = 0, nnz = 0;
  for (int i = 0; i < N; i++) {
    keys[n++] = ((uint64_t)(uint32_t)i << 32) | (uint32_t)i;
    for (int k = mo[i]; k < mo[i + 1]; k++) keys[n++] = ((uint64_t)(uint32_t)i << 32) | (uint32_t)mc[k];
  }
  for (int r = 0; r < nc; r++)
    for (int a = joff[r]; a < joff[r + 1]; a++)
      for (int b = joff[r]; b < joff[r + 1]; b++)
        keys[n++] = ((uint64_t)(uint32_t)(jcol[a] / 3) << 32) | (uint32_t)(jcol[b] / 3);
  qsort(keys, n, sizeof(uint64_t), cmp_u64);
  for (size_t k = 0; k < n; k++) if (k == 0 || keys[k] != keys[k - 1]) keys[nnz++] = keys[k];
  int *cols = (int *)malloc((nnz ? nnz : 1) * sizeof(int));
  memset(offsets, 0, sizeof(int) * ((size_t)N + 1));
  for (size_t k = 0; k < nnz; k++) { cols[k] = (int)(keys[k] & 0xffffffffULL); offsets[(keys[k] >> 32) + 1]++; }
  for (int i = 0; i < N; i++) offsets[i + 1] += offsets[i];
  free(keys);
  *columns = cols;
  return (int)nnz;
}

/* c = J x - rhs  (ANCF3243DataFunc.cuh:477-499) */
void orc_lin_constraint(int nc, const int *joff, const int *jcol, const double *jval, const double *rhs,
                        const double *x, const double *y, const double *z, double *c) {
  for (int r = 0; r < nc; r++) {
    double sum = 0.0;
    for (int k = joff[r]; k < joff[r + 1]; k++) {
      const int coef = jcol[k] / 3, comp = jcol[k] % 3;
      sum += jval[k] * (comp == 0 ? x[coef] : (comp == 1 ? y[coef] : z[coef]));
    }
    c[r] = sum - rhs[r];
  }
}

/* g += h J^T (lam + rho c)  (SyncedNewton.cu:377-404; rows visited in ascending constraint id per DOF) */
void orc_lin_grad_add(int nc, const int *joff, const int *jcol, const double *jval, const double *c,
                      const double *lam, double h, double rho, double *g) {
  for (int r = 0; r < nc; r++)
    for (int k = joff[r]; k < joff[r + 1]; k++) g[jcol[k]] += h * jval[k] * (lam[r] + rho * c[r]);
}

/* H = M/h + h K + C_vis + h^2 rho J^T J on the constraint-aware pattern (SyncedNewton.cu:292-341) */
void orc_gen_assemble_hessian_lin(int S, int Q, int E, int N, const int *conn, const double *x, const double *y,
                                  const double *z, const double *gradN, const double *detJ, const double *qw,
                                  const orc_material *mat, const int *mo, const int *mc, const double *mv,
                                  const int *ao, const int *ac, int nc, const int *joff, const int *jcol,
                                  const double *jval, double h, double rho, const int *ro, const int *ci,
                                  double *val) {
  gen_assemble_core(S, Q, E, N, conn, x, y, z, gradN, detJ, qw, mat, mo, mc, mv, ao, ac, h, ro, ci, val);
  const double f = h * h * rho;
  for (int r = 0; r < nc; r++)
    for (int a = joff[r]; a < joff[r + 1]; a++)
      for (int b = joff[r]; b < joff[r + 1]; b++) {
        const int di = jcol[a], dj = jcol[b];
        const int p = bsearch_i(ci + ro[di], ro[di + 1] - ro[di], dj);
        if (p >= 0) val[ro[di] + p] += f * jval[a] * jval[b];
      }
}

/* One implicit step with general linear constraints (the same ALM/Newton loop, SyncedNewton.cu:1147-1376, with
 * compute_constraint_data in linear-CSR mode). lam has nc entries. */
int orc_gen_newton_step_lin(int S, int Q, int E, int N, const int *conn, double *x, double *y, double *z,
                            const double *gradN, const double *detJ, const double *qw, const orc_material *mat,
                            const int *mo, const int *mc, const double *mv, int nc, const int *joff, const int *jcol,
                            const double *jval, const double *rhs, const double *f_ext,
                            const orc_newton_params *prm, double *v, double *v_prev, double *lam, double *stats) {
  const int n = 3 * N;
  const double h = prm->time_step, rho = prm->rho;
  int *ao = (int *)malloc(sizeof(int) * ((size_t)N + 1)), *ac = NULL;
  const int annz = orc_lin_adjacency(N, mo, mc, nc, joff, jcol, ao, &ac);
  int *ro = (int *)malloc(sizeof(int) * (n + 1)), *ci = (int *)malloc(sizeof(int) * 9 * (size_t)annz);
  orc_hessian_pattern(N, ao, ac, ro, ci);
  double *H = (double *)malloc(sizeof(double) * (size_t)ro[n]), *xp = (double *)malloc(sizeof(double) * n);
  double *P = (double *)malloc(sizeof(double) * 9 * (size_t)E * Q), *f_int = (double *)malloc(sizeof(double) * n);
  double *g = (double *)malloc(sizeof(double) * n), *r = (double *)malloc(sizeof(double) * n), *dv = (double *)malloc(sizeof(double) * n);
  double *c = (double *)calloc(nc > 0 ? nc : 1, sizeof(double));
  memcpy(xp, x, sizeof(double) * N); memcpy(xp + N, y, sizeof(double) * N); memcpy(xp + 2 * N, z, sizeof(double) * N);
  int status = 0, n_outer = 0, n_newton = 0;
  double ng = 0, ncn = 0;
  for (int outer = 0; outer < prm->max_outer && !status; outer++) {
    n_outer++;
    double ng0 = -1.0;
    for (int it = 0; it < prm->max_inner; it++) {
      orc_gen_compute_p(S, Q, E, conn, x, y, z, v, gradN, mat, NULL, P);
      orc_gen_internal_force(S, Q, E, N, conn, P, gradN, detJ, qw, f_int);
      orc_lin_constraint(nc, joff, jcol, jval, rhs, x, y, z, c);
      orc_grad_L(N, mo, mc, mv, v, v_prev, f_int, f_ext, NULL, 0, NULL, NULL, h, rho, g);
      orc_lin_grad_add(nc, joff, jcol, jval, c, lam, h, rho, g);
      ng = 0; for (int i = 0; i < n; i++) ng += g[i] * g[i]; ng = sqrt(ng);
      if (ng0 < 0) ng0 = ng;
      if (ng < prm->inner_atol || (prm->inner_rtol > 0 && ng0 > 0 && ng <= prm->inner_rtol * ng0)) break;
      for (int i = 0; i < n; i++) r[i] = -g[i];
      orc_gen_assemble_hessian_lin(S, Q, E, N, conn, x, y, z, gradN, detJ, qw, mat, mo, mc, mv, ao, ac, nc, joff, jcol,
                                   jval, h, rho, ro, ci, H);
      status = orc_solve_spd_upper(n, ro, ci, H, r, dv);
      if (status) break;
      n_newton++;
      for (int i = 0; i < n; i++) v[i] += dv[i];
      for (int i = 0; i < N; i++) { x[i] = xp[i] + v[3 * i] * h; y[i] = xp[N + i] + v[3 * i + 1] * h; z[i] = xp[2 * N + i] + v[3 * i + 2] * h; }
    }
    memcpy(v_prev, v, sizeof(double) * n);
    orc_lin_constraint(nc, joff, jcol, jval, rhs, x, y, z, c);
    for (int k = 0; k < nc; k++) lam[k] += rho * c[k];
    if (nc > 0) { ncn = 0; for (int k = 0; k < nc; k++) ncn += c[k] * c[k]; ncn = sqrt(ncn); if (ncn < prm->outer_tol) break; }
  }
  if (stats) { stats[0] = n_outer; stats[1] = n_newton; stats[2] = ng; stats[3] = ncn; }
  free(ao); free(ac); free(ro); free(ci); free(H); free(xp); free(P); free(f_int); free(g); free(r); free(dv); free(c);
  return status;
}

/* One implicit step for any element type with fixed-coefficient constraints (SyncedNewton.cu:1147-1261 is the
 * 3243 branch, :1262-1376 the 3443 branch: textually the T10 branch with another cast). */
int orc_gen_newton_step(int S, int Q, int E, int N, const int *conn, double *x, double *y, double *z,
                        const double *xt, const double *yt, const double *zt, const double *gradN,
                        const double *detJ, const double *qw, const orc_material *mat, const int *mo,
                        const int *mc, const double *mv, const int *fixed, int n_fixed, const double *f_ext,
                        const orc_newton_params *prm, double *v, double *v_prev, double *lam, double *stats) {
  const int n = 3 * N, nc = 3 * n_fixed;
  const double h = prm->time_step, rho = prm->rho;
  int *ro = (int *)malloc(sizeof(int) * (n + 1)), *ci = (int *)malloc(sizeof(int) * 9 * (size_t)mo[N]);
  orc_hessian_pattern(N, mo, mc, ro, ci);
  double *H = (double *)malloc(sizeof(double) * (size_t)ro[n]), *xp = (double *)malloc(sizeof(double) * n);
  double *P = (double *)malloc(sizeof(double) * 9 * (size_t)E * Q), *f_int = (double *)malloc(sizeof(double) * n);
  double *g = (double *)malloc(sizeof(double) * n), *r = (double *)malloc(sizeof(double) * n), *dv = (double *)malloc(sizeof(double) * n);
  double *c = (double *)calloc(nc > 0 ? nc : 1, sizeof(double));
  memcpy(xp, x, sizeof(double) * N); memcpy(xp + N, y, sizeof(double) * N); memcpy(xp + 2 * N, z, sizeof(double) * N);
  int status = 0, n_outer = 0, n_newton = 0;
  double ng = 0, ncn = 0;
  for (int outer = 0; outer < prm->max_outer && !status; outer++) {
    n_outer++;
    double ng0 = -1.0;
    for (int it = 0; it < prm->max_inner; it++) {
      orc_gen_compute_p(S, Q, E, conn, x, y, z, v, gradN, mat, NULL, P);
      orc_gen_internal_force(S, Q, E, N, conn, P, gradN, detJ, qw, f_int);
      for (int k = 0; k < n_fixed; k++) { c[3 * k] = x[fixed[k]] - xt[fixed[k]]; c[3 * k + 1] = y[fixed[k]] - yt[fixed[k]]; c[3 * k + 2] = z[fixed[k]] - zt[fixed[k]]; }
      orc_grad_L(N, mo, mc, mv, v, v_prev, f_int, f_ext, fixed, n_fixed, c, lam, h, rho, g);
      ng = 0; for (int i = 0; i < n; i++) ng += g[i] * g[i]; ng = sqrt(ng);
      if (ng0 < 0) ng0 = ng;
      if (ng < prm->inner_atol || (prm->inner_rtol > 0 && ng0 > 0 && ng <= prm->inner_rtol * ng0)) break;
      for (int i = 0; i < n; i++) r[i] = -g[i];
      orc_gen_assemble_hessian(S, Q, E, N, conn, x, y, z, gradN, detJ, qw, mat, mo, mc, mv, fixed, n_fixed, h, rho, ro, ci, H);
      status = orc_solve_spd_upper(n, ro, ci, H, r, dv);
      if (status) break;
      n_newton++;
      for (int i = 0; i < n; i++) v[i] += dv[i];
      for (int i = 0; i < N; i++) { x[i] = xp[i] + v[3 * i] * h; y[i] = xp[N + i] + v[3 * i + 1] * h; z[i] = xp[2 * N + i] + v[3 * i + 2] * h; }
    }
    memcpy(v_prev, v, sizeof(double) * n);
    for (int k = 0; k < n_fixed; k++) { c[3 * k] = x[fixed[k]] - xt[fixed[k]]; c[3 * k + 1] = y[fixed[k]] - yt[fixed[k]]; c[3 * k + 2] = z[fixed[k]] - zt[fixed[k]]; }
    for (int k = 0; k < nc; k++) lam[k] += rho * c[k];
    if (nc > 0) { ncn = 0; for (int k = 0; k < nc; k++) ncn += c[k] * c[k]; ncn = sqrt(ncn); if (ncn < prm->outer_tol) break; }
  }
  if (stats) { stats[0] = n_outer; stats[1] = n_newton; stats[2] = ng; stats[3] = ncn; }
  free(ro); free(ci); free(H); free(xp); free(P); free(f_int); free(g); free(r); free(dv); free(c);
  return status;
}


/* ---- SyncedAdamWNocoopSolver::OneStepAdamWNocoop (SyncedAdamWNocoop.cu:262-500) for any element type ---------
 * First-order ALM loop on the velocities with fixed-coefficient constraints.  Follows the reference line by line:
 * m, v moments and g are zeroed every OUTER iteration (:344-346); lr = lr0 lr_decay^(inner+1), t = inner+2 (:358-361);
 * the velocity update reads the gradient of the PREVIOUS inner iteration (zero in the first, :363-364, :140-174);
 * convergence is tested every check interval with tol_abs = inner_tol (1 + ||v||) or inner_rtol ||g0|| (:387-424);
 * after the inner loop v_prev = v (:438) and lambda += rho dt c TWICE (adamw_dual_update_kernel, :260-264); the outer
 * loop stops when ||c|| < outer_tol AND the inner loop converged (:460-472); without constraints one outer pass.
 * State in/out: x,y,z, v, v_prev, lam.  stats: outer iterations, inner iterations (total), last ||g||, last ||c||,
 * last inner flag. */
typedef struct {
  double lr, beta1, beta2, eps, weight_decay, lr_decay;
  double inner_tol, outer_tol, rho;
  int max_outer, max_inner;
  double time_step;
  int convergence_check_interval;
  double inner_rtol;
} orc_adamw_params;

/* coop = 0: SyncedAdamWNocoopSolver::OneStepAdamWNocoop (SyncedAdamWNocoop.cu:262-500);
 * coop = 1: SyncedAdamWSolver, the cooperative-kernel sibling (SyncedAdamW.cu:96-345) -- the same moments and step, but
 * as written there: the inner-converged flag is cleared once per Solve() (a converged inner loop is not re-entered by
 * the later outer iterations), the multiplier update lam += rho dt c is applied once (Nocoop applies it twice), and
 * the outer loop stops on ||c|| < outer_tol alone. */
static int gen_adamw_step(int S, int Q, int E, int N, const int *conn, double *x, double *y, double *z,
                          const double *xt, const double *yt, const double *zt, const double *gradN,
                          const double *detJ, const double *qw, const orc_material *mat, const int *mo, const int *mc,
                          const double *mv, const int *fixed, int n_fixed, const double *f_ext,
                          const orc_adamw_params *prm, double *v, double *v_prev, double *lam, double *stats, int coop) {
  const int n = 3 * N, nc = 3 * n_fixed;
  const double dt = prm->time_step, rho = prm->rho;
  const int check_every = prm->convergence_check_interval > 0 ? prm->convergence_check_interval : 1;
  const int max_outer = nc > 0 ? prm->max_outer : 1;
  double *xp = (double *)malloc(sizeof(double) * n), *P = (double *)malloc(sizeof(double) * 9 * (size_t)E * Q);
  double *f_int = (double *)malloc(sizeof(double) * n), *g = (double *)malloc(sizeof(double) * n);
  double *m = (double *)malloc(sizeof(double) * n), *va = (double *)malloc(sizeof(double) * n);
  double *c = (double *)calloc(nc > 0 ? nc : 1, sizeof(double));
  memcpy(xp, x, sizeof(double) * N); memcpy(xp + N, y, sizeof(double) * N); memcpy(xp + 2 * N, z, sizeof(double) * N);
  int outer_flag = 0, inner_flag = 0, n_outer = 0, n_inner = 0;
  double ng = 0.0, ncn = 0.0;
  for (int outer = 0; outer < max_outer; outer++) {
    if (outer_flag) break;
    n_outer++;
    memset(g, 0, sizeof(double) * n); memset(m, 0, sizeof(double) * n); memset(va, 0, sizeof(double) * n);
    if (!coop) inner_flag = 0;
    double ng0 = -1.0;
    for (int inner = 0; inner < prm->max_inner; inner++) {
      if (inner_flag) break;
      n_inner++;
      const double lr = prm->lr * pow(prm->lr_decay, inner + 1), t = (double)(inner + 2);
      const double inv1 = 1.0 / (1.0 - pow(prm->beta1, t)), inv2 = 1.0 / (1.0 - pow(prm->beta2, t));
      for (int i = 0; i < n; i++) {
        const double mt = prm->beta1 * m[i] + (1.0 - prm->beta1) * g[i];
        const double vt = prm->beta2 * va[i] + (1.0 - prm->beta2) * g[i] * g[i];
        m[i] = mt; va[i] = vt;
        v[i] = v[i] - lr * ((mt * inv1) / (sqrt(vt * inv2) + prm->eps) + prm->weight_decay * v[i]);
      }
      for (int i = 0; i < N; i++) { x[i] = xp[i] + dt * v[3 * i]; y[i] = xp[N + i] + dt * v[3 * i + 1]; z[i] = xp[2 * N + i] + dt * v[3 * i + 2]; }
      orc_gen_compute_p(S, Q, E, conn, x, y, z, v, gradN, mat, NULL, P);
      orc_gen_internal_force(S, Q, E, N, conn, P, gradN, detJ, qw, f_int);
      for (int k = 0; k < n_fixed; k++) { c[3 * k] = x[fixed[k]] - xt[fixed[k]]; c[3 * k + 1] = y[fixed[k]] - yt[fixed[k]]; c[3 * k + 2] = z[fixed[k]] - zt[fixed[k]]; }
      orc_grad_L(N, mo, mc, mv, v, v_prev, f_int, f_ext, fixed, n_fixed, c, lam, dt, rho, g);
      if (inner % check_every == 0) {
        double nv = 0.0;
        ng = 0.0;
        for (int i = 0; i < n; i++) { ng += g[i] * g[i]; nv += v[i] * v[i]; }
        ng = sqrt(ng); nv = sqrt(nv);
        if (ng0 < 0.0) ng0 = ng;
        const double tol_abs = prm->inner_tol * (1.0 + nv);
        const double tol_rel = (prm->inner_rtol > 0.0 && ng0 > 0.0) ? prm->inner_rtol * ng0 : 0.0;
        if (ng <= tol_abs || (tol_rel > 0.0 && ng <= tol_rel)) inner_flag = 1;
      }
    }
    memcpy(v_prev, v, sizeof(double) * n);
    for (int i = 0; i < N; i++) { x[i] = xp[i] + dt * v[3 * i]; y[i] = xp[N + i] + dt * v[3 * i + 1]; z[i] = xp[2 * N + i] + dt * v[3 * i + 2]; }
    if (nc > 0) {
      for (int k = 0; k < n_fixed; k++) { c[3 * k] = x[fixed[k]] - xt[fixed[k]]; c[3 * k + 1] = y[fixed[k]] - yt[fixed[k]]; c[3 * k + 2] = z[fixed[k]] - zt[fixed[k]]; }
      for (int k = 0; k < nc; k++) { lam[k] += rho * dt * c[k]; if (!coop) lam[k] += rho * dt * c[k]; }
      ncn = 0.0; for (int k = 0; k < nc; k++) ncn += c[k] * c[k]; ncn = sqrt(ncn);
      if (ncn < prm->outer_tol && (coop || inner_flag)) outer_flag = 1;
    }
  }
  for (int i = 0; i < N; i++) { x[i] = xp[i] + dt * v[3 * i]; y[i] = xp[N + i] + dt * v[3 * i + 1]; z[i] = xp[2 * N + i] + dt * v[3 * i + 2]; }
  if (stats) { stats[0] = n_outer; stats[1] = n_inner; stats[2] = ng; stats[3] = ncn; stats[4] = inner_flag; }
  free(xp); free(P); free(f_int); free(g); free(m); free(va); free(c);
  return 0;
}

int orc_gen_adamw_step(int S, int Q, int E, int N, const int *conn, double *x, double *y, double *z,
                       const double *xt, const double *yt, const double *zt, const double *gradN,
                       const double *detJ, const double *qw, const orc_material *mat, const int *mo, const int *mc,
                       const double *mv, const int *fixed, int n_fixed, const double *f_ext,
                       const orc_adamw_params *prm, double *v, double *v_prev, double *lam, double *stats) {
  return gen_adamw_step(S, Q, E, N, conn, x, y, z, xt, yt, zt, gradN, detJ, qw, mat, mo, mc, mv, fixed, n_fixed, f_ext, prm,
                        v, v_prev, lam, stats, 0);
}
int orc_gen_adamw_coop_step(int S, int Q, int E, int N, const int *conn, double *x, double *y, double *z,
                            const double *xt, const double *yt, const double *zt, const double *gradN,
                            const double *detJ, const double *qw, const orc_material *mat, const int *mo, const int *mc,
                            const double *mv, const int *fixed, int n_fixed, const double *f_ext,
                            const orc_adamw_params *prm, double *v, double *v_prev, double *lam, double *stats) {
  return gen_adamw_step(S, Q, E, N, conn, x, y, z, xt, yt, zt, gradN, detJ, qw, mat, mo, mc, mv, fixed, n_fixed, f_ext, prm,
                        v, v_prev, lam, stats, 1);
}


/* ---- SyncedNesterovSolver: one_step_nesterov_kernel (SyncedNesterov.cu:95-372) for any element type -----------
 * Restated phase by phase: x_prev saved and both flags cleared ONCE per call (:99-113); per outer iteration (only
 * while the outer flag is clear) v_k = v_km1 = v_guess, t = 1 (:118-139); per inner iteration (only while the inner
 * flag is clear -- it is never cleared again inside the call) look-ahead y (:151-158), x = x_prev + dt y, compute_p,
 * f_int, constraints, g (:162-228), flag on | ||g|| - ||g_prev|| | < inner_tol for inner > 0 (:230-247), v_next = y -
 * alpha g (:252-258), flag on | ||v_next|| - ||v_k|| | < inner_tol (:262-283), rotate (:287-293); then v_prev = v
 * (:305), x update, constraints, lambda += rho dt c (:333-338), outer flag on ||c|| < outer_tol (:341-354). */
typedef struct {
  double alpha, rho, inner_tol, outer_tol;
  int max_outer, max_inner;
  double time_step;
} orc_nesterov_params;

int orc_gen_nesterov_step(int S, int Q, int E, int N, const int *conn, double *x, double *y, double *z,
                          const double *xt, const double *yt, const double *zt, const double *gradN,
                          const double *detJ, const double *qw, const orc_material *mat, const int *mo,
                          const int *mc, const double *mv, const int *fixed, int n_fixed, const double *f_ext,
                          const orc_nesterov_params *prm, double *v, double *v_prev, double *lam, double *stats) {
  const int n = 3 * N, nc = 3 * n_fixed;
  const double dt = prm->time_step, rho = prm->rho;
  double *xp = (double *)malloc(sizeof(double) * n), *P = (double *)malloc(sizeof(double) * 9 * (size_t)E * Q);
  double *f_int = (double *)malloc(sizeof(double) * n), *g = (double *)malloc(sizeof(double) * n);
  double *vk = (double *)malloc(sizeof(double) * n), *vkm1 = (double *)malloc(sizeof(double) * n);
  double *vnext = (double *)malloc(sizeof(double) * n);
  double *c = (double *)calloc(nc > 0 ? nc : 1, sizeof(double));
  memcpy(xp, x, sizeof(double) * N); memcpy(xp + N, y, sizeof(double) * N); memcpy(xp + 2 * N, z, sizeof(double) * N);
  int inner_flag = 0, outer_flag = 0, n_outer = 0, n_inner = 0;
  double ng = 0.0, ncn = 0.0;
  for (int outer = 0; outer < prm->max_outer; outer++) {
    if (outer_flag) continue;
    n_outer++;
    memcpy(vk, v, sizeof(double) * n); memcpy(vkm1, v, sizeof(double) * n);
    double t = 1.0, prev_ng = 0.0;
    for (int inner = 0; inner < prm->max_inner; inner++) {
      if (inner_flag) continue;
      n_inner++;
      const double t_next = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t * t)), beta = (t - 1.0) / t_next;
      for (int i = 0; i < n; i++) v[i] = vk[i] + beta * (vk[i] - vkm1[i]);
      for (int i = 0; i < N; i++) { x[i] = xp[i] + dt * v[3 * i]; y[i] = xp[N + i] + dt * v[3 * i + 1]; z[i] = xp[2 * N + i] + dt * v[3 * i + 2]; }
      orc_gen_compute_p(S, Q, E, conn, x, y, z, v, gradN, mat, NULL, P);
      orc_gen_internal_force(S, Q, E, N, conn, P, gradN, detJ, qw, f_int);
      for (int k = 0; k < n_fixed; k++) { c[3 * k] = x[fixed[k]] - xt[fixed[k]]; c[3 * k + 1] = y[fixed[k]] - yt[fixed[k]]; c[3 * k + 2] = z[fixed[k]] - zt[fixed[k]]; }
      orc_grad_L(N, mo, mc, mv, v, v_prev, f_int, f_ext, fixed, n_fixed, c, lam, dt, rho, g);
      ng = 0.0; for (int i = 0; i < n; i++) ng += g[i] * g[i]; ng = sqrt(ng);
      if (inner > 0 && fabs(ng - prev_ng) < prm->inner_tol) inner_flag = 1;
      double nvn = 0.0, nvk = 0.0;
      for (int i = 0; i < n; i++) { vnext[i] = v[i] - prm->alpha * g[i]; nvn += vnext[i] * vnext[i]; nvk += vk[i] * vk[i]; }
      nvn = sqrt(nvn); nvk = sqrt(nvk);
      if (inner > 0 && fabs(nvn - nvk) < prm->inner_tol) inner_flag = 1;
      memcpy(vkm1, vk, sizeof(double) * n); memcpy(vk, vnext, sizeof(double) * n); memcpy(v, vnext, sizeof(double) * n);
      t = t_next;
      prev_ng = ng;
    }
    memcpy(v_prev, v, sizeof(double) * n);
    for (int i = 0; i < N; i++) { x[i] = xp[i] + dt * v[3 * i]; y[i] = xp[N + i] + dt * v[3 * i + 1]; z[i] = xp[2 * N + i] + dt * v[3 * i + 2]; }
    for (int k = 0; k < n_fixed; k++) { c[3 * k] = x[fixed[k]] - xt[fixed[k]]; c[3 * k + 1] = y[fixed[k]] - yt[fixed[k]]; c[3 * k + 2] = z[fixed[k]] - zt[fixed[k]]; }
    for (int k = 0; k < nc; k++) lam[k] += rho * dt * c[k];
    ncn = 0.0; for (int k = 0; k < nc; k++) ncn += c[k] * c[k]; ncn = sqrt(ncn);
    if (fabs(ncn) < prm->outer_tol) outer_flag = 1;
  }
  for (int i = 0; i < N; i++) { x[i] = xp[i] + dt * v[3 * i]; y[i] = xp[N + i] + dt * v[3 * i + 1]; z[i] = xp[2 * N + i] + dt * v[3 * i + 2]; }
  if (stats) { stats[0] = n_outer; stats[1] = n_inner; stats[2] = ng; stats[3] = ncn; stats[4] = inner_flag; }
  free(xp); free(P); free(f_int); free(g); free(vk); free(vkm1); free(vnext); free(c);
  return 0;
}

/* =====================================================================================================
 * SyncedVBDSolver (SyncedVBD.cu:47-82 solve, :163-377 node update, :1233-1412 sweep/post graphs, :1475-1641 step):
 * ALM outer loop, inner loop of coloured Gauss-Seidel sweeps of per-node 3x3 Newton updates on the velocities.
 * Restated as written: F/P cached per (element, point) and refreshed for ALL elements after every colour group;
 * residual of node i = full mass row . (v - v_prev)/h + sum cached P h_a dV - f_ext (+ pin terms), Hessian = diagonal
 * mass block / h + h * sum K_aa (elastic part only) (+ h^2 rho I on pinned nodes), symmetrised, + eps max(1, tr) I. */
static void vbd_diag_block(const double F[3][3], const double ha[3], const orc_material *mat, double w, double K[3][3]) {
  if (mat->model == ORC_MAT_MOONEY_RIVLIN) { /* FEAT10DataFunc.cuh:353-372 */
    double A[3][3][3][3];
    mr_tangent_tensor(F, mat, A);
    for (int d = 0; d < 3; d++) for (int e = 0; e < 3; e++) {
      double sum = 0.0;
      for (int J = 0; J < 3; J++) for (int L = 0; L < 3; L++) sum += A[d][J][e][L] * ha[J] * ha[L];
      K[d][e] = sum * w;
    }
    return;
  }
  double FFT[3][3] = {{0}}, Fh[3] = {0, 0, 0}, trC = 0.0; /* SVK.cuh:35-55 with i == j */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { trC += F[i][j] * F[i][j]; Fh[i] += F[i][j] * ha[j]; for (int k = 0; k < 3; k++) FFT[i][j] += F[i][k] * F[j][k]; }
  const double trE = 0.5 * (trC - 3.0), hij = ha[0] * ha[0] + ha[1] * ha[1] + ha[2] * ha[2], ff = Fh[0] * Fh[0] + Fh[1] * Fh[1] + Fh[2] * Fh[2];
  for (int d = 0; d < 3; d++) for (int e = 0; e < 3; e++) {
    const double dl = d == e;
    K[d][e] = (mat->lambda * Fh[d] * Fh[e] + mat->lambda * trE * hij * dl + mat->mu * ff * dl + mat->mu * Fh[d] * Fh[e] +
               mat->mu * hij * FFT[d][e] - mat->mu * hij * dl) * w;
  }
}

/* element contributions of one node from the cached F/P (vbd_accumulate_residual_and_hessian_diag,
 * FEAT10DataFunc.cuh:295-395 and the ANCF twins): r[3] += sum P h_a dV, Kd[3][3] += h * sum K_aa */
static void vbd_node_elements(int S, int Q, int E, int node, const int *inc_off, const int *inc, const double *Fc,
                              const double *Pc, const double *gradN, const double *detJ, const double *qw,
                              const orc_material *mat, double h, double r[3], double Kd[3][3]) {
  (void)E;
  for (int k = inc_off[node]; k < inc_off[node + 1]; k++) {
    const int e = inc[k] / S, a = inc[k] % S;
    for (int q = 0; q < Q; q++) {
      const double *g = gradN + ((size_t)e * Q + q) * 3 * S, *Pq = Pc + ((size_t)e * Q + q) * 9, *Fq = Fc + ((size_t)e * Q + q) * 9;
      const double ha[3] = {g[a], g[a + S], g[a + 2 * S]}, dV = detJ[(size_t)e * Q + q] * qw[q];
      double F[3][3], K[3][3];
      for (int i = 0; i < 3; i++) { r[i] += (Pq[i] * ha[0] + Pq[i + 3] * ha[1] + Pq[i + 6] * ha[2]) * dV; for (int j = 0; j < 3; j++) F[i][j] = Fq[i + 3 * j]; }
      vbd_diag_block(F, ha, mat, h * dV, K);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Kd[i][j] += K[i][j];
    }
  }
}

/* node -> (element, local index) incidence, ascending element (cpu_utils.cc BuildNodeIncidence); entries e*S + a */
static void vbd_incidence(int S, int E, int N, const int *conn, int *off, int *inc) {
  memset(off, 0, sizeof(int) * ((size_t)N + 1));
  for (int e = 0; e < E; e++) for (int a = 0; a < S; a++) off[conn[(size_t)a * E + e] + 1]++;
  for (int i = 0; i < N; i++) off[i + 1] += off[i];
  int *cur = (int *)malloc(sizeof(int) * (size_t)N);
  memcpy(cur, off, sizeof(int) * (size_t)N);
  for (int e = 0; e < E; e++) for (int a = 0; a < S; a++) inc[cur[conn[(size_t)a * E + e]]++] = e * S + a;
  free(cur);
}

/* per-node element terms at the current state (for pinning against the reference's NumPy prototype
 * test-scripts/vbd_proto/alm_vbd_t10_svk.py:175-222: f_int_i and sum K_ii, there without the factor h) */
void orc_gen_vbd_node_terms(int S, int Q, int E, int N, const int *conn, const double *x, const double *y, const double *z,
                            const double *v, const double *gradN, const double *detJ, const double *qw,
                            const orc_material *mat, double h, double *r_out /*3N*/, double *K_out /*9N row-major*/) {
  double *Fc = (double *)malloc(sizeof(double) * 9 * (size_t)E * Q), *Pc = (double *)malloc(sizeof(double) * 9 * (size_t)E * Q);
  int *off = (int *)malloc(sizeof(int) * ((size_t)N + 1)), *inc = (int *)malloc(sizeof(int) * (size_t)S * E);
  vbd_incidence(S, E, N, conn, off, inc);
  orc_gen_compute_p(S, Q, E, conn, x, y, z, v, gradN, mat, Fc, Pc);
  for (int i = 0; i < N; i++) {
    double r[3] = {0, 0, 0}, K[3][3] = {{0}};
    vbd_node_elements(S, Q, E, i, off, inc, Fc, Pc, gradN, detJ, qw, mat, h, r, K);
    for (int d = 0; d < 3; d++) { r_out[3 * i + d] = r[d]; for (int e = 0; e < 3; e++) K_out[9 * (size_t)i + 3 * d + e] = K[d][e]; }
  }
  free(Fc); free(Pc); free(off); free(inc);
}

int orc_gen_vbd_step(int S, int Q, int E, int N, const int *conn, double *x, double *y, double *z, const double *xt,
                     const double *yt, const double *zt, const double *gradN, const double *detJ, const double *qw,
                     const orc_material *mat, const int *mo, const int *mc, const double *mv, const int *fixed,
                     int n_fixed, const double *f_ext, const orc_vbd_params *prm, int n_colors, const int *color_offsets,
                     const int *color_nodes, int n_groups, const int *group_offsets, const int *group_colors, double *v,
                     double *v_prev, double *lam, double *stats) {
  (void)n_colors;
  const int n = 3 * N, nc = 3 * n_fixed;
  const double h = prm->time_step, inv_h = 1.0 / h, rho = prm->rho;
  double *xp = (double *)malloc(sizeof(double) * n), *Fc = (double *)malloc(sizeof(double) * 9 * (size_t)E * Q);
  double *Pc = (double *)malloc(sizeof(double) * 9 * (size_t)E * Q), *f_int = (double *)malloc(sizeof(double) * n);
  double *g = (double *)malloc(sizeof(double) * n), *c = (double *)calloc(nc > 0 ? nc : 1, sizeof(double));
  double *mdiag = (double *)calloc((size_t)N, sizeof(double));
  int *off = (int *)malloc(sizeof(int) * ((size_t)N + 1)), *inc = (int *)malloc(sizeof(int) * (size_t)S * E);
  int *fmap = (int *)malloc(sizeof(int) * (size_t)N);
  vbd_incidence(S, E, N, conn, off, inc);
  for (int i = 0; i < N; i++) { fmap[i] = -1; for (int k = mo[i]; k < mo[i + 1]; k++) if (mc[k] == i) { mdiag[i] = mv[k]; break; } } /* :1030-1085 */
  for (int k = 0; k < n_fixed; k++) fmap[fixed[k]] = k;                                                                            /* :146-160 */
  memcpy(xp, x, sizeof(double) * N); memcpy(xp + N, y, sizeof(double) * N); memcpy(xp + 2 * N, z, sizeof(double) * N);
  int n_outer = 0, n_sweeps = 0;
  double ng = 0.0, ncn = 0.0;
#define VBD_POS() for (int i = 0; i < N; i++) { x[i] = xp[i] + h * v[3 * i]; y[i] = xp[N + i] + h * v[3 * i + 1]; z[i] = xp[2 * N + i] + h * v[3 * i + 2]; }
#define VBD_CONS() for (int k = 0; k < n_fixed; k++) { c[3 * k] = x[fixed[k]] - xt[fixed[k]]; c[3 * k + 1] = y[fixed[k]] - yt[fixed[k]]; c[3 * k + 2] = z[fixed[k]] - zt[fixed[k]]; }
#define VBD_GNORM(out) do { orc_gen_internal_force(S, Q, E, N, conn, Pc, gradN, detJ, qw, f_int); VBD_CONS(); \
    orc_grad_L(N, mo, mc, mv, v, v_prev, f_int, f_ext, fixed, n_fixed, c, lam, h, rho, g); \
    double s_ = 0.0; for (int i = 0; i < n; i++) s_ += g[i] * g[i]; (out) = sqrt(s_); } while (0)
  for (int outer = 0; outer < prm->max_outer; outer++) {
    n_outer++;
    VBD_POS();
    orc_gen_compute_p(S, Q, E, conn, x, y, z, v, gradN, mat, Fc, Pc);
    double R0 = -1.0;
    if (prm->convergence_check_interval > 0) VBD_GNORM(R0);
    for (int inner = 0; inner < prm->max_inner; inner++) {
      n_sweeps++;
      for (int gi = 0; gi < n_groups; gi++) {
        for (int t = group_offsets[gi]; t < group_offsets[gi + 1]; t++) {
          const int col = group_colors[t];
          for (int s = color_offsets[col]; s < color_offsets[col + 1]; s++) { /* vbd_update_color_block_kernel */
            const int i = color_nodes[s];
            double r[3] = {0, 0, 0}, K[3][3] = {{0}};
            for (int k = mo[i]; k < mo[i + 1]; k++) for (int d = 0; d < 3; d++) r[d] += mv[k] * (v[3 * mc[k] + d] - v_prev[3 * mc[k] + d]) * inv_h;
            vbd_node_elements(S, Q, E, i, off, inc, Fc, Pc, gradN, detJ, qw, mat, h, r, K);
            double R[3], H[3][3];
            for (int d = 0; d < 3; d++) { R[d] = r[d] - f_ext[3 * i + d]; for (int e = 0; e < 3; e++) H[d][e] = (d == e ? mdiag[i] * inv_h : 0.0) + K[d][e]; }
            if (nc > 0 && fmap[i] >= 0) {
              const int k = fmap[i];
              const double xi[3] = {xp[i] + h * v[3 * i], xp[N + i] + h * v[3 * i + 1], xp[2 * N + i] + h * v[3 * i + 2]};
              const double Xi[3] = {xt[i], yt[i], zt[i]};
              for (int d = 0; d < 3; d++) { R[d] += h * (lam[3 * k + d] + rho * (xi[d] - Xi[d])); H[d][d] += h * h * rho; }
            }
            const double a01 = 0.5 * (H[0][1] + H[1][0]), a02 = 0.5 * (H[0][2] + H[2][0]), a12 = 0.5 * (H[1][2] + H[2][1]);
            H[0][1] = H[1][0] = a01; H[0][2] = H[2][0] = a02; H[1][2] = H[2][1] = a12;
            const double tr = H[0][0] + H[1][1] + H[2][2], eps = prm->hess_eps * (tr > 1.0 ? tr : 1.0);
            H[0][0] += eps; H[1][1] += eps; H[2][2] += eps;
            const double det = H[0][0] * (H[1][1] * H[2][2] - H[1][2] * H[2][1]) - H[0][1] * (H[1][0] * H[2][2] - H[1][2] * H[2][0]) +
                               H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0]);
            double dv[3] = {0, 0, 0};
            if (fabs(det) >= 1e-30) { /* solve_3x3_vbd: cofactor inverse, dv = -H^-1 R */
              const double id = 1.0 / det;
              double Hi[3][3];
              Hi[0][0] = (H[1][1] * H[2][2] - H[1][2] * H[2][1]) * id; Hi[0][1] = (H[0][2] * H[2][1] - H[0][1] * H[2][2]) * id;
              Hi[0][2] = (H[0][1] * H[1][2] - H[0][2] * H[1][1]) * id; Hi[1][0] = (H[1][2] * H[2][0] - H[1][0] * H[2][2]) * id;
              Hi[1][1] = (H[0][0] * H[2][2] - H[0][2] * H[2][0]) * id; Hi[1][2] = (H[0][2] * H[1][0] - H[0][0] * H[1][2]) * id;
              Hi[2][0] = (H[1][0] * H[2][1] - H[1][1] * H[2][0]) * id; Hi[2][1] = (H[0][1] * H[2][0] - H[0][0] * H[2][1]) * id;
              Hi[2][2] = (H[0][0] * H[1][1] - H[0][1] * H[1][0]) * id;
              for (int d = 0; d < 3; d++) dv[d] = -(Hi[d][0] * R[0] + Hi[d][1] * R[1] + Hi[d][2] * R[2]);
            }
            for (int d = 0; d < 3; d++) v[3 * i + d] += prm->omega * dv[d];
          }
          for (int s = color_offsets[col]; s < color_offsets[col + 1]; s++) { /* vbd_update_pos_from_vel_color */
            const int i = color_nodes[s];
            x[i] = xp[i] + h * v[3 * i]; y[i] = xp[N + i] + h * v[3 * i + 1]; z[i] = xp[2 * N + i] + h * v[3 * i + 2];
          }
        }
        orc_gen_compute_p(S, Q, E, conn, x, y, z, v, gradN, mat, Fc, Pc); /* refresh after the group */
      }
      if (prm->convergence_check_interval > 0 && (inner % prm->convergence_check_interval == 0 || inner == prm->max_inner - 1)) {
        VBD_GNORM(ng);
        const double a = prm->inner_tol, b = prm->inner_rtol * (R0 >= 0.0 ? R0 : ng);
        if (ng <= (a > b ? a : b)) break;
      }
    }
    VBD_POS(); /* post-outer graph: positions, constraints */
    if (nc > 0) {
      VBD_CONS();
      ncn = 0.0; for (int k = 0; k < nc; k++) ncn += c[k] * c[k]; ncn = sqrt(ncn);
      if (ncn < prm->outer_tol) break;
      for (int k = 0; k < nc; k++) lam[k] += rho * c[k]; /* vbd_update_dual */
    }
  }
  memcpy(v_prev, v, sizeof(double) * n);
  if (stats) { stats[0] = n_outer; stats[1] = n_sweeps; stats[2] = ng; stats[3] = ncn; }
#undef VBD_POS
#undef VBD_CONS
#undef VBD_GNORM
  free(xp); free(Fc); free(Pc); free(f_int); free(g); free(c); free(mdiag); free(off); free(inc); free(fmap);
  return 0;
}
