// ancf_host.h -- host-side set-up math of the ANCF-3243 beam and ANCF-3443 shell (one-time per mesh; the
// per-iteration path is the templated device kernels).  Reference: lib_utils/cpu_utils.cc:125-420 (B matrices),
// ANCF3243Data.cu:102-288 / ANCF3443Data.cu:96-254 (reference gradients, mass), ANCF3243DataFunc.cuh:115-135 and
// ANCF3443DataFunc.cuh:114-237 (bases).  Both bases are monomials u^a v^b w^c, so one exponent table per type
// generates b, its derivatives and the B matrix.
#pragma once
#include <cmath>
#include <vector>

namespace tlfea {
namespace ancf {

struct Basis {
  int S;
  const int (*ex)[3];
};
inline Basis basis(int S) {
  static const int e8[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {2, 0, 0}, {3, 0, 0}};
  static const int e16[16][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1},
                                 {2, 0, 0}, {0, 2, 0}, {2, 1, 0}, {1, 2, 0}, {3, 0, 0}, {0, 3, 0}, {3, 1, 0}, {1, 3, 0}};
  return S == 8 ? Basis{8, e8} : Basis{16, e16};
}
inline double ipow(double x, int n) {
  double r = 1.0;
  for (int i = 0; i < n; i++) r *= x;
  return r;
}
// which: 0 value, 1..3 derivative wrt u, v, w
inline void eval(const Basis& B, double u, double v, double w, int which, double* out) {
  for (int k = 0; k < B.S; k++) {
    const int a = B.ex[k][0], b = B.ex[k][1], c = B.ex[k][2];
    double val;
    if (which == 0) val = ipow(u, a) * ipow(v, b) * ipow(w, c);
    else if (which == 1) val = a ? a * ipow(u, a - 1) * ipow(v, b) * ipow(w, c) : 0.0;
    else if (which == 2) val = b ? b * ipow(u, a) * ipow(v, b - 1) * ipow(w, c) : 0.0;
    else val = c ? c * ipow(u, a) * ipow(v, b) * ipow(w, c - 1) : 0.0;
    out[k] = val;
  }
}

// Gauss-Jordan with partial pivoting; A row-major n x n (destroyed).  false if singular.
inline bool invert(int n, std::vector<double>& A, std::vector<double>& inv) {
  inv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = 1.0;
  for (int k = 0; k < n; k++) {
    int p = k;
    for (int i = k + 1; i < n; i++)
      if (std::fabs(A[(size_t)i * n + k]) > std::fabs(A[(size_t)p * n + k])) p = i;
    if (std::fabs(A[(size_t)p * n + k]) < 1e-300) return false;
    if (p != k)
      for (int j = 0; j < n; j++) {
        std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]);
        std::swap(inv[(size_t)k * n + j], inv[(size_t)p * n + j]);
      }
    const double d = 1.0 / A[(size_t)k * n + k];
    for (int j = 0; j < n; j++) {
      A[(size_t)k * n + j] *= d;
      inv[(size_t)k * n + j] *= d;
    }
    for (int i = 0; i < n; i++) {
      if (i == k) continue;
      const double f = A[(size_t)i * n + k];
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++) {
        A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
        inv[(size_t)i * n + j] -= f * inv[(size_t)k * n + j];
      }
    }
  }
  return true;
}

// (B^T)^-1, column-major S x S like the reference's d_B_inv.  Rows of B: (b, b_u, b_v, b_w) at every node's
// reference point: beam (-L/2,0,0),(+L/2,0,0); shell P1(-,-) P2(+,-) P3(+,+) P4(-,+) at w = 0.
inline bool B_inv(int S, double L, double W, double* out_colmajor) {
  const Basis B = basis(S);
  const int nn = S / 4;
  double pts[4][3] = {{0}};
  if (S == 8) {
    pts[0][0] = -L / 2;
    pts[1][0] = L / 2;
  } else {
    const double sx[4] = {-1, 1, 1, -1}, sy[4] = {-1, -1, 1, 1};
    for (int n = 0; n < 4; n++) {
      pts[n][0] = sx[n] * L / 2;
      pts[n][1] = sy[n] * W / 2;
    }
  }
  std::vector<double> Bm((size_t)S * S), BT((size_t)S * S), inv;
  for (int n = 0; n < nn; n++)
    for (int wch = 0; wch < 4; wch++) eval(B, pts[n][0], pts[n][1], pts[n][2], wch, &Bm[(size_t)(4 * n + wch) * S]);
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) BT[(size_t)i * S + j] = Bm[(size_t)j * S + i];
  if (!invert(S, BT, inv)) return false;
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) out_colmajor[(size_t)j * S + i] = inv[(size_t)i * S + j];
  return true;
}

// ds/d(xi,eta,zeta) = B_inv * db/d(xi,..) at u = L xi/2, v = W eta/2, w = H zeta/2
inline void ds_dxi(int S, const double* Binv_cm, double L, double W, double H, double xi, double eta, double zeta,
                   double ds[3][16]) {
  const Basis B = basis(S);
  double db[16];
  const double sc[3] = {L / 2, W / 2, H / 2};
  for (int d = 0; d < 3; d++) {
    eval(B, L * xi / 2, W * eta / 2, H * zeta / 2, d + 1, db);
    for (int i = 0; i < S; i++) {
      double s = 0.0;
      for (int j = 0; j < S; j++) s += Binv_cm[(size_t)j * S + i] * db[j] * sc[d];
      ds[d][i] = s;
    }
  }
}

inline double jacobian(int S, const int* coefs, const double* xj, const double* yj, const double* zj,
                       const double ds[3][16], double J[3][3]) {
  for (int i = 0; i < 3; i++)
    for (int d = 0; d < 3; d++) J[i][d] = 0.0;
  for (int a = 0; a < S; a++) {
    const double X[3] = {xj[coefs[a]], yj[coefs[a]], zj[coefs[a]]};
    for (int i = 0; i < 3; i++)
      for (int d = 0; d < 3; d++) J[i][d] += X[i] * ds[d][a];
  }
  return J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
         J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
}

// J^T g = rhs by pivoted elimination, zero if |pivot| < 1e-14 (ANCF3243DataFunc.cuh:30-88)
inline void solve3(const double A[3][3], const double b[3], double x[3]) {
  double m[3][4];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) m[i][j] = A[i][j];
    m[i][3] = b[i];
  }
  for (int k = 0; k < 3; k++) {
    int p = k;
    for (int i = k + 1; i < 3; i++)
      if (std::fabs(m[i][k]) > std::fabs(m[p][k])) p = i;
    if (p != k)
      for (int j = 0; j < 4; j++) std::swap(m[k][j], m[p][j]);
    if (std::fabs(m[k][k]) < 1e-14) {
      x[0] = x[1] = x[2] = 0.0;
      return;
    }
    for (int i = k + 1; i < 3; i++) {
      const double f = m[i][k] / m[k][k];
      for (int j = k; j < 4; j++) m[i][j] -= f * m[k][j];
    }
  }
  x[2] = m[2][3] / m[2][2];
  x[1] = (m[1][3] - m[1][2] * x[2]) / m[1][1];
  x[0] = (m[0][3] - m[0][2] * x[2] - m[0][1] * x[1]) / m[0][0];
}

}  // namespace ancf
}  // namespace tlfea
