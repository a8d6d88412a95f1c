/* tlfea_oracle.c -- CPU ORACLE (test infrastructure only; see tlfea_oracle.h).
 *
 * Plain-C restatement of the reference's T10 Total-Lagrangian hot path.  Operation order follows
 * the cited reference lines so that element-level results agree with the CUDA code to the last
 * few ulps (FMA contraction aside).  Paths are relative to the reference root.
 */
#include "tlfea_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NQ 5
#define NN 10

void orc_free(void *p) { free(p); }

/* quadrature_utils.h:140-158 */
void orc_keast5(double *qx, double *qy, double *qz, double *qw) {
  const double a = 0.5, b = 1.0 / 6.0;
  const double bary[NQ][4] = {
      {0.25, 0.25, 0.25, 0.25}, {a, b, b, b}, {b, a, b, b}, {b, b, a, b}, {b, b, b, a}};
  const double w[NQ] = {-4.0 / 5.0, 9.0 / 20.0, 9.0 / 20.0, 9.0 / 20.0, 9.0 / 20.0};
  for (int q = 0; q < NQ; q++) {
    qx[q] = bary[q][1];
    qy[q] = bary[q][2];
    qz[q] = bary[q][3];
    qw[q] = w[q] * (1.0 / 6.0);
  }
}

/* cpu_utils.cc:607-624 */
void orc_t10_remap_tetgen(const int *t, int *s) {
  static const int map[NN] = {0, 1, 2, 3, 6, 7, 9, 5, 8, 4};
  for (int i = 0; i < NN; i++) s[i] = t[map[i]];
}

/* FEAT10DataFunc.cuh:30-83 */
static void solve_3x3_system(const double A[3][3], const double b[3], double x[3]) {
  double aug[3][4];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) aug[i][j] = A[i][j];
    aug[i][3] = b[i];
  }
  for (int k = 0; k < 3; k++) {
    int pivot_row = k;
    double max_val = fabs(aug[k][k]);
    for (int i = k + 1; i < 3; i++) {
      if (fabs(aug[i][k]) > max_val) {
        max_val = fabs(aug[i][k]);
        pivot_row = i;
      }
    }
    if (pivot_row != k) {
      for (int j = 0; j < 4; j++) {
        double t = aug[k][j];
        aug[k][j] = aug[pivot_row][j];
        aug[pivot_row][j] = t;
      }
    }
    if (fabs(aug[k][k]) < 1e-14) {
      x[0] = x[1] = x[2] = 0.0;
      return;
    }
    for (int i = k + 1; i < 3; i++) {
      double factor = aug[i][k] / aug[k][k];
      for (int j = k; j < 4; j++) aug[i][j] -= factor * aug[k][j];
    }
  }
  x[2] = aug[2][3] / aug[2][2];
  x[1] = (aug[1][3] - aug[1][2] * x[2]) / aug[1][1];
  x[0] = (aug[0][3] - aug[0][2] * x[2] - aug[0][1] * x[1]) / aug[0][0];
}

static const int kEdges[6][2] = {{0, 1}, {1, 2}, {0, 2}, {0, 3}, {1, 3}, {2, 3}}; /* FEAT10Data.cu:143 */

/* FEAT10Data.cu:97-204 */
void orc_t10_dndu_pre(int E, const int *conn, const double *x, const double *y, const double *z,
                      const double *qx, const double *qy, const double *qz, double *gradN,
                      double *detJ_out) {
  static const double dL[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int e = 0; e < E; e++) {
    double X[NN][3];
    for (int a = 0; a < NN; a++) {
      int g = conn[a * E + e];
      X[a][0] = x[g];
      X[a][1] = y[g];
      X[a][2] = z[g];
    }
    for (int q = 0; q < NQ; q++) {
      double L[4] = {1.0 - qx[q] - qy[q] - qz[q], qx[q], qy[q], qz[q]};
      double dN[NN][3];
      for (int i = 0; i < 4; i++) {
        double factor = 4.0 * L[i] - 1.0;
        for (int j = 0; j < 3; j++) dN[i][j] = factor * dL[i][j];
      }
      for (int k = 0; k < 6; k++) {
        int i = kEdges[k][0], j = kEdges[k][1];
        for (int d = 0; d < 3; d++) dN[k + 4][d] = 4.0 * (L[i] * dL[j][d] + L[j] * dL[i][d]);
      }
      double J[3][3] = {{0}};
      for (int a = 0; a < NN; a++)
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) J[i][j] += X[a][i] * dN[a][j];
      double detJ = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) -
                    J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                    J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      detJ_out[e * NQ + q] = detJ;
      double JT[3][3];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) JT[i][j] = J[j][i];
      double *g = gradN + ((size_t)e * NQ + q) * 30;
      for (int a = 0; a < NN; a++) {
        double ga[3];
        solve_3x3_system(JT, dN[a], ga);
        g[a + 10 * 0] = ga[0];
        g[a + 10 * 1] = ga[1];
        g[a + 10 * 2] = ga[2];
      }
    }
  }
}

/* ---------- materials ---------- */

/* SVK.cuh:14-32 */
static void svk_P(const double F[3][3], double trFtF, const double FFtF[3][3], double lambda,
                  double mu, double P[3][3]) {
  double lambda_factor = lambda * (0.5 * trFtF - 1.5);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) P[i][j] = lambda_factor * F[i][j] + mu * (FFtF[i][j] - F[i][j]);
}

/* SVK.cuh:35-55 */
static void svk_tangent_block(const double Fh_i[3], const double Fh_j[3], double hij, double trE,
                              double Fhj_dot_Fhi, const double FFT[3][3], double lambda,
                              double mu, double dV, double K[3][3]) {
  for (int d = 0; d < 3; d++)
    for (int e = 0; e < 3; e++) {
      double delta = (d == e) ? 1.0 : 0.0;
      double A_de = lambda * Fh_i[d] * Fh_j[e];
      double B_de = lambda * trE * hij * delta;
      double C1_de = mu * Fhj_dot_Fhi * delta;
      double D_de = mu * Fh_j[d] * Fh_i[e];
      double E_de = mu * hij * FFT[d][e];
      double F_de = -mu * hij * delta;
      K[d][e] = (A_de + B_de + C1_de + D_de + E_de + F_de) * dV;
    }
}

/* MooneyRivlin.cuh:17-43 */
static double mr_det3(const double A[3][3]) {
  return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) -
         A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
         A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
}
static void mr_invT(const double A[3][3], double detA, double o[3][3]) {
  const double eps = 1e-12;
  double sd = detA;
  if (fabs(sd) < eps) sd = (sd >= 0.0) ? eps : -eps;
  double id = 1.0 / sd;
  o[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) * id;
  o[0][1] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) * id;
  o[0][2] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) * id;
  o[1][0] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id;
  o[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id;
  o[1][2] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id;
  o[2][0] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id;
  o[2][1] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id;
  o[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
}

typedef struct {
  double C[3][3], FC[3][3], FFT[3][3], FinvT[3][3];
  double I1, I2, J, t1, t2, t3;
} mr_state;

/* shared prologue of mr_compute_P / mr_compute_tangent_tensor (MooneyRivlin.cuh:48-95,116-179) */
static void mr_prologue(const double F[3][3], double mu10, double mu01, double kappa, mr_state *s) {
  memset(s, 0, sizeof(*s));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) s->C[i][j] += F[k][i] * F[k][j];
  s->I1 = s->C[0][0] + s->C[1][1] + s->C[2][2];
  double C2[3][3] = {{0}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) C2[i][j] += s->C[i][k] * s->C[k][j];
  double trC2 = C2[0][0] + C2[1][1] + C2[2][2];
  s->I2 = 0.5 * (s->I1 * s->I1 - trC2);
  s->J = mr_det3(F);
  mr_invT(F, s->J, s->FinvT);
  double J13 = cbrt(s->J);
  double Jm23 = 1.0 / (J13 * J13);
  double Jm43 = Jm23 * Jm23;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) s->FC[i][j] += F[i][k] * s->C[k][j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) s->FFT[i][j] += F[i][k] * F[j][k];
  s->t1 = 2.0 * mu10 * Jm23;
  s->t2 = 2.0 * mu01 * Jm43;
  s->t3 = kappa * (s->J - 1.0) * s->J;
}

/* MooneyRivlin.cuh:45-111 */
static void mr_P(const double F[3][3], double mu10, double mu01, double kappa, double P[3][3]) {
  mr_state s;
  mr_prologue(F, mu10, mu01, kappa, &s);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double term1 = F[i][j] - (s.I1 / 3.0) * s.FinvT[i][j];
      double term2 = s.I1 * F[i][j] - s.FC[i][j] - (2.0 * s.I2 / 3.0) * s.FinvT[i][j];
      double term3 = s.FinvT[i][j];
      P[i][j] = s.t1 * term1 + s.t2 * term2 + s.t3 * term3;
    }
}

/* MooneyRivlin.cuh:113-225 */
static void mr_tangent(const double F[3][3], double mu10, double mu01, double kappa,
                       double A[3][3][3][3]) {
  mr_state s;
  mr_prologue(F, mu10, mu01, kappa, &s);
  double term1[3][3], term2[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      term1[i][j] = F[i][j] - (s.I1 / 3.0) * s.FinvT[i][j];
      term2[i][j] = s.I1 * F[i][j] - s.FC[i][j] - (2.0 * s.I2 / 3.0) * s.FinvT[i][j];
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          double dik = (i == k) ? 1.0 : 0.0, djl = (j == l) ? 1.0 : 0.0;
          double dFinvT = -s.FinvT[i][l] * s.FinvT[k][j];
          double dt1 = (-2.0 / 3.0) * s.t1 * s.FinvT[k][l];
          double dt2 = (-4.0 / 3.0) * s.t2 * s.FinvT[k][l];
          double dt3 = (kappa * (2.0 * s.J - 1.0) * s.J) * s.FinvT[k][l];
          double dT1 = dik * djl - (2.0 / 3.0) * F[k][l] * s.FinvT[i][j] +
                       (s.I1 / 3.0) * s.FinvT[i][l] * s.FinvT[k][j];
          double dT2 = 2.0 * F[k][l] * F[i][j] + s.I1 * dik * djl -
                       (dik * s.C[l][j] + F[i][l] * F[k][j] + djl * s.FFT[i][k]) -
                       (4.0 / 3.0) * (s.I1 * F[k][l] - s.FC[k][l]) * s.FinvT[i][j] +
                       (2.0 * s.I2 / 3.0) * s.FinvT[i][l] * s.FinvT[k][j];
          A[i][j][k][l] = dt1 * term1[i][j] + s.t1 * dT1 + dt2 * term2[i][j] + s.t2 * dT2 +
                          dt3 * s.FinvT[i][j] + s.t3 * dFinvT;
        }
}

/* ---------- compute_p (FEAT10DataFunc.cuh:85-293) ---------- */
void orc_t10_compute_p(int E, const int *conn, const double *x, const double *y, const double *z,
                       const double *v, const double *gradN, const orc_material *mat, double *Fo,
                       double *Po, double *Fdoto, double *Pviso) {
  const int do_damp = (v != NULL) && (mat->eta_damp != 0.0 || mat->lambda_damp != 0.0);
  for (int e = 0; e < E; e++) {
    int gn[NN];
    double xn[NN][3];
    for (int a = 0; a < NN; a++) {
      gn[a] = conn[a * E + e];
      xn[a][0] = x[gn[a]];
      xn[a][1] = y[gn[a]];
      xn[a][2] = z[gn[a]];
    }
    for (int q = 0; q < NQ; q++) {
      const double *g = gradN + ((size_t)e * NQ + q) * 30;
      double gN[NN][3];
      for (int a = 0; a < NN; a++)
        for (int d = 0; d < 3; d++) gN[a][d] = g[a + 10 * d];
      double F[3][3] = {{0}};
      for (int a = 0; a < NN; a++)
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) F[i][j] += xn[a][i] * gN[a][j];
      double Pvis[3][3] = {{0}}, Fdot[3][3] = {{0}};
      if (do_damp) {
        for (int a = 0; a < NN; a++) {
          double va[3] = {v[gn[a] * 3 + 0], v[gn[a] * 3 + 1], v[gn[a] * 3 + 2]};
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Fdot[i][j] += va[i] * gN[a][j];
        }
        double FdotT_F[3][3] = {{0}}, Ft_Fdot[3][3] = {{0}}, Edot[3][3];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) FdotT_F[i][j] += Fdot[k][i] * F[k][j];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) Ft_Fdot[i][j] += F[k][i] * Fdot[k][j];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) Edot[i][j] = 0.5 * (FdotT_F[i][j] + Ft_Fdot[i][j]);
        double trEdot = Edot[0][0] + Edot[1][1] + Edot[2][2];
        double S[3][3];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
            S[i][j] = 2.0 * mat->eta_damp * Edot[i][j] +
                      mat->lambda_damp * trEdot * (i == j ? 1.0 : 0.0);
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) Pvis[i][j] += F[i][k] * S[k][j];
      }
      double FtF[3][3] = {{0}}, FFt[3][3] = {{0}}, FFtF[3][3] = {{0}};
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          for (int k = 0; k < 3; k++) FtF[i][j] += F[k][i] * F[k][j];
      double trFtF = FtF[0][0] + FtF[1][1] + FtF[2][2];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          for (int k = 0; k < 3; k++) FFt[i][j] += F[i][k] * F[j][k];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          for (int k = 0; k < 3; k++) FFtF[i][j] += FFt[i][k] * F[k][j];
      double Pel[3][3];
      if (mat->model == ORC_MAT_MOONEY_RIVLIN)
        mr_P(F, mat->mu10, mat->mu01, mat->kappa, Pel);
      else
        svk_P(F, trFtF, FFtF, mat->lambda, mat->mu, Pel);
      size_t o = ((size_t)e * NQ + q) * 9;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          if (Fo) Fo[o + i + 3 * j] = F[i][j];
          if (Fdoto) Fdoto[o + i + 3 * j] = Fdot[i][j];
          if (Pviso) Pviso[o + i + 3 * j] = Pvis[i][j];
          Po[o + i + 3 * j] = Pel[i][j] + Pvis[i][j];
        }
    }
  }
}

/* FEAT10DataFunc.cuh:397-466 (clear + per (e,a) accumulation; element order = deterministic) */
void orc_t10_internal_force(int E, int N, const int *conn, const double *P, const double *gradN,
                            const double *detJ, const double *qw, double *f_int) {
  memset(f_int, 0, sizeof(double) * 3 * (size_t)N);
  for (int e = 0; e < E; e++)
    for (int a = 0; a < NN; a++) {
      int gnode = conn[a * E + e];
      double f[3] = {0, 0, 0};
      for (int q = 0; q < NQ; q++) {
        const double *Pq = P + ((size_t)e * NQ + q) * 9;
        const double *g = gradN + ((size_t)e * NQ + q) * 30;
        double gn[3] = {g[a], g[a + 10], g[a + 20]};
        double dV = detJ[e * NQ + q] * qw[q];
        for (int i = 0; i < 3; i++) {
          double c = 0.0;
          for (int j = 0; j < 3; j++) c += Pq[i + 3 * j] * gn[j];
          f[i] += c * dV;
        }
      }
      for (int i = 0; i < 3; i++) f_int[3 * gnode + i] += f[i];
    }
}

/* per-QP local matrices exactly as compute_hessian_assemble_csr builds them
 * (FEAT10DataFunc.cuh:520-659 elastic, :695-762 viscous). K,C are [30][30] row-major. */
static void t10_qp_tangent(const double xn[NN][3], const double *g, double dV,
                           const orc_material *mat, double K[30][30], double C[30][30],
                           int want_vis) {
  double gN[NN][3];
  for (int a = 0; a < NN; a++)
    for (int d = 0; d < 3; d++) gN[a][d] = g[a + 10 * d];
  double F[3][3] = {{0}};
  for (int a = 0; a < NN; a++)
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) F[i][j] += xn[a][i] * gN[a][j];
  double Cm[3][3] = {{0}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) Cm[i][j] += F[k][i] * F[k][j];
  double trC = Cm[0][0] + Cm[1][1] + Cm[2][2];
  double trE = 0.5 * (trC - 3.0);
  double FFT[3][3] = {{0}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) FFT[i][j] += F[i][k] * F[j][k];
  double Fh[NN][3];
  for (int i = 0; i < NN; i++)
    for (int r = 0; r < 3; r++) {
      Fh[i][r] = 0.0;
      for (int c = 0; c < 3; c++) Fh[i][r] += F[r][c] * gN[i][c];
    }
  const int use_mr = (mat->model == ORC_MAT_MOONEY_RIVLIN);
  double A[3][3][3][3];
  if (use_mr) mr_tangent(F, mat->mu10, mat->mu01, mat->kappa, A);
  for (int i = 0; i < NN; i++)
    for (int j = 0; j < NN; j++) {
      double hij = gN[j][0] * gN[i][0] + gN[j][1] * gN[i][1] + gN[j][2] * gN[i][2];
      double FhjFhi = Fh[j][0] * Fh[i][0] + Fh[j][1] * Fh[i][1] + Fh[j][2] * Fh[i][2];
      double Kb[3][3];
      if (use_mr) {
        for (int d = 0; d < 3; d++)
          for (int e = 0; e < 3; e++) {
            double sum = 0.0;
            for (int J = 0; J < 3; J++)
              for (int L = 0; L < 3; L++) sum += A[d][J][e][L] * gN[i][J] * gN[j][L];
            Kb[d][e] = sum * dV;
          }
      } else {
        svk_tangent_block(Fh[i], Fh[j], hij, trE, FhjFhi, FFT, mat->lambda, mat->mu, dV, Kb);
      }
      for (int d = 0; d < 3; d++)
        for (int e = 0; e < 3; e++) K[3 * i + d][3 * j + e] = Kb[d][e];
    }
  if (!want_vis) return;
  const double eta = mat->eta_damp, lamd = mat->lambda_damp;
  for (int a = 0; a < NN; a++)
    for (int b = 0; b < NN; b++) {
      double hdot = gN[a][0] * gN[b][0] + gN[a][1] * gN[b][1] + gN[a][2] * gN[b][2];
      for (int d = 0; d < 3; d++)
        for (int e = 0; e < 3; e++)
          C[3 * a + d][3 * b + e] =
              (eta * (Fh[b][d] * Fh[a][e]) + eta * FFT[d][e] * hdot + lamd * (Fh[a][d] * Fh[b][e])) *
              dV;
    }
}

void orc_t10_element_tangent(int e, int E, const int *conn, const double *x, const double *y,
                             const double *z, const double *gradN, const double *detJ,
                             const double *qw, const orc_material *mat, double *Ke, double *Ce) {
  double xn[NN][3];
  for (int a = 0; a < NN; a++) {
    int g = conn[a * E + e];
    xn[a][0] = x[g];
    xn[a][1] = y[g];
    xn[a][2] = z[g];
  }
  memset(Ke, 0, sizeof(double) * 900);
  if (Ce) memset(Ce, 0, sizeof(double) * 900);
  double K[30][30], C[30][30];
  for (int q = 0; q < NQ; q++) {
    double dV = detJ[e * NQ + q] * qw[q];
    t10_qp_tangent(xn, gradN + ((size_t)e * NQ + q) * 30, dV, mat, K, C, Ce != NULL);
    for (int i = 0; i < 900; i++) Ke[i] += (&K[0][0])[i];
    if (Ce)
      for (int i = 0; i < 900; i++) Ce[i] += (&C[0][0])[i];
  }
}

/* ---------- sparsity ---------- */
static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return (x > y) - (x < y);
}

/* FEAT10Data.cu:32-77,372-440: keys (row<<32|col) -> sort -> unique -> counts -> scan */
int orc_t10_mass_pattern(int E, int N, const int *conn, int *offsets, int **columns) {
  size_t total = (size_t)E * 100;
  uint64_t *keys = (uint64_t *)malloc(total * sizeof(uint64_t));
  for (int e = 0; e < E; e++)
    for (int i = 0; i < NN; i++)
      for (int j = 0; j < NN; j++)
        keys[(size_t)e * 100 + i * 10 + j] =
            ((uint64_t)(uint32_t)conn[i * E + e] << 32) | (uint64_t)(uint32_t)conn[j * E + e];
  qsort(keys, total, sizeof(uint64_t), cmp_u64);
  size_t nnz = 0;
  for (size_t k = 0; k < total; k++)
    if (k == 0 || keys[k] != keys[k - 1]) keys[nnz++] = keys[k];
  int *cols = (int *)malloc(nnz * sizeof(int));
  memset(offsets, 0, sizeof(int) * ((size_t)N + 1));
  for (size_t k = 0; k < nnz; k++) {
    cols[k] = (int)(keys[k] & 0xffffffffULL);
    offsets[(keys[k] >> 32) + 1]++;
  }
  for (int i = 0; i < N; i++) offsets[i + 1] += offsets[i];
  free(keys);
  *columns = cols;
  return (int)nnz;
}

static int bsearch_col(const int *cols, int n, int target) { /* FEAT10Data.cu:79-95 */
  int left = 0, right = n - 1;
  while (left <= right) {
    int mid = left + ((right - left) >> 1);
    int v = cols[mid];
    if (v == target) return mid;
    if (v < target)
      left = mid + 1;
    else
      right = mid - 1;
  }
  return -1;
}

/* FEAT10Data.cu:206-278 */
void orc_t10_mass_values(int E, const int *conn, const double *detJ, const double *qx,
                         const double *qy, const double *qz, const double *qw, double rho0,
                         const int *offsets, const int *columns, double *values) {
  double Nq[NQ][NN];
  for (int q = 0; q < NQ; q++) {
    double L[4] = {1.0 - qx[q] - qy[q] - qz[q], qx[q], qy[q], qz[q]};
    for (int k = 0; k < 4; k++) Nq[q][k] = L[k] * (2.0 * L[k] - 1.0);
    for (int k = 0; k < 6; k++) Nq[q][k + 4] = 4.0 * L[kEdges[k][0]] * L[kEdges[k][1]];
  }
  for (int e = 0; e < E; e++)
    for (int i = 0; i < NN; i++)
      for (int j = 0; j < NN; j++) {
        int gi = conn[i * E + e], gj = conn[j * E + e];
        double m = 0.0;
        for (int q = 0; q < NQ; q++) m += rho0 * Nq[q][i] * Nq[q][j] * detJ[e * NQ + q] * qw[q];
        int rs = offsets[gi];
        int k = bsearch_col(columns + rs, offsets[gi + 1] - rs, gj);
        if (k >= 0) values[rs + k] += m;
      }
}

/* SyncedNewton.cu:163-205 */
void orc_hessian_pattern(int N, const int *offsets, const int *columns, int *row_offsets,
                         int *col_indices) {
  row_offsets[0] = 0;
  for (int i = 0; i < N; i++) {
    int deg = offsets[i + 1] - offsets[i];
    for (int d = 0; d < 3; d++) row_offsets[3 * i + d + 1] = row_offsets[3 * i + d] + 3 * deg;
  }
  for (int r = 0; r < 3 * N; r++) {
    int ci = r / 3, out = row_offsets[r];
    for (int k = offsets[ci]; k < offsets[ci + 1]; k++) {
      int base = 3 * columns[k];
      col_indices[out++] = base + 0;
      col_indices[out++] = base + 1;
      col_indices[out++] = base + 2;
    }
  }
}

static inline void add_at(double *p, double v, int atomic) {
  if (atomic) {
#pragma omp atomic
    *p += v;
  } else {
    *p += v;
  }
}

/* SyncedNewton.cu:1080-1097: memset, mass/h (:214-259), h*K (+C_vis) (FEAT10DataFunc.cuh:661-790),
 * h^2 rho J^T J (:292-341). */
void orc_t10_assemble_hessian(int E, int N, const int *conn, const double *x, const double *y,
                              const double *z, const double *gradN, const double *detJ,
                              const double *qw, const orc_material *mat, const int *m_offsets,
                              const int *m_columns, const double *m_values, const int *fixed_nodes,
                              int n_fixed, double h, double rho, const int *ro, const int *ci,
                              double *val, int nthreads) {
  memset(val, 0, sizeof(double) * (size_t)ro[3 * N]);
  const double inv_h = 1.0 / h;
  for (int ni = 0; ni < N; ni++)
    for (int k = m_offsets[ni]; k < m_offsets[ni + 1]; k++) {
      int nj = m_columns[k];
      double contrib = m_values[k] * inv_h;
      for (int d = 0; d < 3; d++) {
        int row = 3 * ni + d, col = 3 * nj + d;
        int p = bsearch_col(ci + ro[row], ro[row + 1] - ro[row], col);
        if (p >= 0) val[ro[row] + p] += contrib;
      }
    }
  const int want_vis = (mat->eta_damp != 0.0 || mat->lambda_damp != 0.0);
  const int par = nthreads > 1;
  (void)par;
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1) if (nthreads > 1)
  for (int e = 0; e < E; e++) {
    int gn[NN];
    double xn[NN][3];
    for (int a = 0; a < NN; a++) {
      gn[a] = conn[a * E + e];
      xn[a][0] = x[gn[a]];
      xn[a][1] = y[gn[a]];
      xn[a][2] = z[gn[a]];
    }
    /* column position of node b's block in node a's row (replaces the per-entry binary search;
     * same indices) */
    int pos[NN][NN];
    for (int a = 0; a < NN; a++) {
      int rs = m_offsets[gn[a]];
      for (int b = 0; b < NN; b++) pos[a][b] = bsearch_col(m_columns + rs, m_offsets[gn[a] + 1] - rs, gn[b]);
    }
    double K[30][30], C[30][30];
    for (int q = 0; q < NQ; q++) {
      double dV = detJ[e * NQ + q] * qw[q];
      t10_qp_tangent(xn, gradN + ((size_t)e * NQ + q) * 30, dV, mat, K, C, want_vis);
      for (int a = 0; a < NN; a++)
        for (int d = 0; d < 3; d++) {
          int row = 3 * gn[a] + d;
          double *rv = val + ro[row];
          for (int b = 0; b < NN; b++)
            for (int c = 0; c < 3; c++) {
              if (pos[a][b] < 0) continue;
              add_at(rv + 3 * pos[a][b] + c, h * K[3 * a + d][3 * b + c], nthreads > 1);
            }
        }
      if (want_vis)
        for (int a = 0; a < NN; a++)
          for (int d = 0; d < 3; d++) {
            int row = 3 * gn[a] + d;
            double *rv = val + ro[row];
            for (int b = 0; b < NN; b++)
              for (int c = 0; c < 3; c++) {
                if (pos[a][b] < 0) continue;
                add_at(rv + 3 * pos[a][b] + c, C[3 * a + d][3 * b + c], nthreads > 1);
              }
          }
    }
  }
  const double factor = h * h * rho;
  for (int k = 0; k < 3 * n_fixed; k++) {
    int dof = fixed_nodes[k / 3] * 3 + k % 3; /* J row k has a single 1.0 (FEAT10Data.cu:443-459) */
    int p = bsearch_col(ci + ro[dof], ro[dof + 1] - ro[dof], dof);
    if (p >= 0) val[ro[dof] + p] += factor * 1.0 * 1.0;
  }
}

/* SyncedNewton.cu:344-407 */
void orc_grad_L(int N, const int *mo, const int *mc, const double *mv, const double *v,
                const double *v_prev, const double *f_int, const double *f_ext,
                const int *fixed_nodes, int n_fixed, const double *c, const double *lam, double h,
                double rho, double *g) {
  const double inv_dt = 1.0 / h;
  for (int tid = 0; tid < 3 * N; tid++) {
    int ni = tid / 3, d = tid % 3;
    double res = 0.0;
    for (int k = mo[ni]; k < mo[ni + 1]; k++) {
      int tj = mc[k] * 3 + d;
      double vdiff = v[tj] - v_prev[tj];
      res += mv[k] * vdiff * inv_dt;
    }
    res -= (-f_int[tid]);
    res -= f_ext[tid];
    g[tid] = res;
  }
  /* J^T rows: constraint k touches DOF fixed[k/3]*3+k%3 with value 1 (FEAT10Data.cu:476-496) */
  for (int k = 0; k < 3 * n_fixed; k++) {
    int dof = fixed_nodes[k / 3] * 3 + k % 3;
    g[dof] += h * 1.0 * (lam[k] + rho * c[k]);
  }
}

/* ---------- linear solvers ---------- */

/* reverse Cuthill-McKee on the node graph given by the block structure of the DOF CSR */
static void rcm_nodes(int N, const int *ro, const int *ci, int *perm /* new->old */) {
  int *deg = (int *)malloc(sizeof(int) * N), *visited = (int *)calloc(N, sizeof(int));
  int *queue = (int *)malloc(sizeof(int) * N), *nbr = (int *)malloc(sizeof(int) * N);
  for (int i = 0; i < N; i++) deg[i] = (ro[3 * i + 1] - ro[3 * i]) / 3;
  int count = 0;
  while (count < N) {
    int start = -1;
    for (int i = 0; i < N; i++)
      if (!visited[i] && (start < 0 || deg[i] < deg[start])) start = i;
    int head = count;
    queue[count++] = start;
    visited[start] = 1;
    while (head < count) {
      int u = queue[head++], nn = 0;
      for (int k = ro[3 * u]; k < ro[3 * u + 1]; k += 3) {
        int w = ci[k] / 3;
        if (!visited[w]) {
          visited[w] = 1;
          nbr[nn++] = w;
        }
      }
      for (int a = 1; a < nn; a++) { /* insertion sort by degree */
        int t = nbr[a], b = a - 1;
        while (b >= 0 && deg[nbr[b]] > deg[t]) {
          nbr[b + 1] = nbr[b];
          b--;
        }
        nbr[b + 1] = t;
      }
      for (int a = 0; a < nn; a++) queue[count++] = nbr[a];
    }
  }
  for (int i = 0; i < N; i++) perm[i] = queue[N - 1 - i];
  free(deg);
  free(visited);
  free(queue);
  free(nbr);
}

int orc_solve_spd_upper(int n, const int *ro, const int *ci, const double *val, const double *rhs,
                        double *sol) {
  int N = n / 3;
  int *perm = (int *)malloc(sizeof(int) * N), *inv = (int *)malloc(sizeof(int) * N);
  rcm_nodes(N, ro, ci, perm);
  for (int i = 0; i < N; i++) inv[perm[i]] = i;
  /* skyline (lower profile) of P A P^T built from the UPPER triangle of A (MVIEW_UPPER) */
  int *first = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) first[i] = i;
  for (int r = 0; r < n; r++)
    for (int k = ro[r]; k < ro[r + 1]; k++) {
      int c = ci[k];
      if (c < r) continue;
      int pr = 3 * inv[r / 3] + r % 3, pc = 3 * inv[c / 3] + c % 3;
      int hi = pr > pc ? pr : pc, lo = pr > pc ? pc : pr;
      if (lo < first[hi]) first[hi] = lo;
    }
  size_t *start = (size_t *)malloc(sizeof(size_t) * (n + 1));
  start[0] = 0;
  for (int i = 0; i < n; i++) start[i + 1] = start[i] + (size_t)(i - first[i] + 1);
  double *L = (double *)calloc(start[n], sizeof(double));
#define LL(i, j) L[start[i] + (size_t)((j)-first[i])]
  for (int r = 0; r < n; r++)
    for (int k = ro[r]; k < ro[r + 1]; k++) {
      int c = ci[k];
      if (c < r) continue;
      int pr = 3 * inv[r / 3] + r % 3, pc = 3 * inv[c / 3] + c % 3;
      int hi = pr > pc ? pr : pc, lo = pr > pc ? pc : pr;
      LL(hi, lo) += val[k];
    }
  int status = 0;
  for (int i = 0; i < n && !status; i++) {
    for (int j = first[i]; j <= i; j++) {
      int k0 = first[i] > first[j] ? first[i] : first[j];
      double s = LL(i, j);
      const double *li = &LL(i, k0), *lj = &LL(j, k0);
      for (int k = 0; k < j - k0; k++) s -= li[k] * lj[k];
      if (j < i) {
        LL(i, j) = s / LL(j, j);
      } else {
        if (!(s > 0.0)) {
          status = 1;
          break;
        }
        LL(i, i) = sqrt(s);
      }
    }
  }
  if (!status) {
    double *yv = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) {
      int old = 3 * perm[i / 3] + i % 3;
      double s = rhs[old];
      const double *li = &LL(i, first[i]);
      for (int k = first[i]; k < i; k++) s -= li[k - first[i]] * yv[k];
      yv[i] = s / LL(i, i);
    }
    for (int i = n - 1; i >= 0; i--) {
      yv[i] /= LL(i, i);
      const double *li = &LL(i, first[i]);
      for (int k = first[i]; k < i; k++) yv[k] -= li[k - first[i]] * yv[i];
    }
    for (int i = 0; i < n; i++) sol[3 * perm[i / 3] + i % 3] = yv[i];
    free(yv);
  }
#undef LL
  free(L);
  free(start);
  free(first);
  free(perm);
  free(inv);
  return status;
}

int orc_solve_pcg(int n, const int *ro, const int *ci, const double *val, const double *rhs,
                  double *x, double rel_tol, int max_iter, int nthreads) {
  int N = n / 3;
  if (nthreads < 1) nthreads = 1;
  double *r = (double *)malloc(sizeof(double) * n), *zv = (double *)malloc(sizeof(double) * n);
  double *p = (double *)malloc(sizeof(double) * n), *q = (double *)malloc(sizeof(double) * n);
  double *Dinv = (double *)malloc(sizeof(double) * 9 * (size_t)N);
  for (int i = 0; i < N; i++) {
    double D[3][3] = {{0}};
    for (int d = 0; d < 3; d++) {
      int row = 3 * i + d;
      int pz = bsearch_col(ci + ro[row], ro[row + 1] - ro[row], 3 * i);
      for (int e = 0; e < 3; e++) D[d][e] = val[ro[row] + pz + e];
    }
    double det = mr_det3(D), T[3][3];
    mr_invT(D, det, T); /* T = D^-T; D symmetric up to rounding -> store transpose */
    for (int d = 0; d < 3; d++)
      for (int e = 0; e < 3; e++) Dinv[9 * (size_t)i + 3 * d + e] = T[e][d];
  }
  memset(x, 0, sizeof(double) * n);
  double rz = 0.0, bnorm2 = 0.0;
  for (int i = 0; i < N; i++) {
    for (int d = 0; d < 3; d++) r[3 * i + d] = rhs[3 * i + d];
    for (int d = 0; d < 3; d++) {
      double s = 0;
      for (int e = 0; e < 3; e++) s += Dinv[9 * (size_t)i + 3 * d + e] * r[3 * i + e];
      zv[3 * i + d] = s;
    }
  }
  for (int i = 0; i < n; i++) {
    p[i] = zv[i];
    rz += r[i] * zv[i];
    bnorm2 += rhs[i] * rhs[i];
  }
  if (bnorm2 == 0.0) {
    free(r); free(zv); free(p); free(q); free(Dinv);
    return 0;
  }
  int it = 0;
  for (; it < max_iter; it++) {
    double pq = 0.0;
#pragma omp parallel for reduction(+ : pq) num_threads(nthreads) schedule(static)
    for (int row = 0; row < n; row++) {
      double s = 0.0;
      for (int k = ro[row]; k < ro[row + 1]; k++) s += val[k] * p[ci[k]];
      q[row] = s;
      pq += s * p[row];
    }
    double alpha = rz / pq, rz_new = 0.0, rr = 0.0;
#pragma omp parallel for reduction(+ : rz_new, rr) num_threads(nthreads) schedule(static)
    for (int i = 0; i < N; i++) {
      for (int d = 0; d < 3; d++) {
        x[3 * i + d] += alpha * p[3 * i + d];
        r[3 * i + d] -= alpha * q[3 * i + d];
      }
      for (int d = 0; d < 3; d++) {
        double s = 0;
        for (int e = 0; e < 3; e++) s += Dinv[9 * (size_t)i + 3 * d + e] * r[3 * i + e];
        zv[3 * i + d] = s;
        rz_new += r[3 * i + d] * s;
        rr += r[3 * i + d] * r[3 * i + d];
      }
    }
    if (rr <= rel_tol * rel_tol * bnorm2) {
      it++;
      break;
    }
    double beta = rz_new / rz;
    rz = rz_new;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < n; i++) p[i] = zv[i] + beta * p[i];
  }
  free(r); free(zv); free(p); free(q); free(Dinv);
  return it;
}

/* ---------- one implicit step (SyncedNewton.cu:1032-1146) ---------- */
static double nrm2(const double *a, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s += a[i] * a[i];
  return sqrt(s);
}

int orc_t10_newton_step(int E, int N, const int *conn, double *x, double *y, double *z,
                        const double *xt, const double *yt, const double *zt, const double *gradN,
                        const double *detJ, const double *qw, const orc_material *mat,
                        const int *mo, const int *mc, const double *mv, const int *fixed,
                        int n_fixed, const double *f_ext, const orc_newton_params *prm, double *v,
                        double *v_prev, double *lam, int solver, int nthreads, double *stats) {
  const int n = 3 * N, nc = 3 * n_fixed;
  const double h = prm->time_step, rho = prm->rho;
  int *ro = (int *)malloc(sizeof(int) * (n + 1));
  int *ci = (int *)malloc(sizeof(int) * 9 * (size_t)mo[N]);
  orc_hessian_pattern(N, mo, mc, ro, ci);
  double *H = (double *)malloc(sizeof(double) * (size_t)ro[n]);
  double *xp = (double *)malloc(sizeof(double) * n); /* x_prev, y_prev, z_prev */
  double *P = (double *)malloc(sizeof(double) * 45 * (size_t)E);
  double *f_int = (double *)malloc(sizeof(double) * n), *g = (double *)malloc(sizeof(double) * n);
  double *r = (double *)malloc(sizeof(double) * n), *dv = (double *)malloc(sizeof(double) * n);
  double *c = (double *)calloc(nc > 0 ? nc : 1, sizeof(double));
  memcpy(xp, x, sizeof(double) * N);           /* cudss_solve_update_pos_prev (:413-422) */
  memcpy(xp + N, y, sizeof(double) * N);
  memcpy(xp + 2 * N, z, sizeof(double) * N);
  int status = 0, n_outer = 0, n_newton = 0;
  double norm_g = 0.0, norm_c = 0.0;
  for (int outer = 0; outer < prm->max_outer && !status; outer++) {
    n_outer++;
    double norm_g0 = -1.0;
    for (int it = 0; it < prm->max_inner; it++) {
      orc_t10_compute_p(E, conn, x, y, z, v, gradN, mat, NULL, P, NULL, NULL);
      orc_t10_internal_force(E, N, conn, P, gradN, detJ, qw, f_int);
      for (int k = 0; k < n_fixed; k++) { /* compute_constraint_data (FEAT10DataFunc.cuh:468-483) */
        c[3 * k + 0] = x[fixed[k]] - xt[fixed[k]];
        c[3 * k + 1] = y[fixed[k]] - yt[fixed[k]];
        c[3 * k + 2] = z[fixed[k]] - zt[fixed[k]];
      }
      orc_grad_L(N, mo, mc, mv, v, v_prev, f_int, f_ext, fixed, n_fixed, c, lam, h, rho, g);
      norm_g = nrm2(g, n);
      if (norm_g0 < 0.0) norm_g0 = norm_g;
      if (norm_g < prm->inner_atol ||
          (prm->inner_rtol > 0.0 && norm_g0 > 0.0 && norm_g <= prm->inner_rtol * norm_g0))
        break;
      for (int i = 0; i < n; i++) r[i] = -g[i];
      orc_t10_assemble_hessian(E, N, conn, x, y, z, gradN, detJ, qw, mat, mo, mc, mv, fixed,
                               n_fixed, h, rho, ro, ci, H, nthreads);
      if (solver == 0)
        status = orc_solve_spd_upper(n, ro, ci, H, r, dv);
      else
        orc_solve_pcg(n, ro, ci, H, r, dv, 1e-13, 20000, nthreads);
      if (status) break;
      n_newton++;
      for (int i = 0; i < n; i++) v[i] += dv[i];
      for (int i = 0; i < N; i++) { /* cudss_solve_update_pos (:504-519) */
        x[i] = xp[i] + v[3 * i + 0] * h;
        y[i] = xp[N + i] + v[3 * i + 1] * h;
        z[i] = xp[2 * N + i] + v[3 * i + 2] * h;
      }
    }
    memcpy(v_prev, v, sizeof(double) * n); /* :1122 -- every OUTER iteration */
    for (int k = 0; k < n_fixed; k++) {
      c[3 * k + 0] = x[fixed[k]] - xt[fixed[k]];
      c[3 * k + 1] = y[fixed[k]] - yt[fixed[k]];
      c[3 * k + 2] = z[fixed[k]] - zt[fixed[k]];
    }
    for (int k = 0; k < nc; k++) lam[k] += rho * c[k];
    if (nc > 0) {
      norm_c = nrm2(c, nc);
      if (norm_c < prm->outer_tol) break;
    }
  }
  if (stats) {
    stats[0] = n_outer;
    stats[1] = n_newton;
    stats[2] = norm_g;
    stats[3] = norm_c;
  }
  free(ro); free(ci); free(H); free(xp); free(P); free(f_int); free(g); free(r); free(dv); free(c);
  return status;
}
