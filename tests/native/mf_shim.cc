// Test shim: exposes the host plan of the sparse direct solve (csrc/mf_host.h, integer work only) to ctypes and EXECUTES
// it with plain loops on the CPU, so that the CPU suite can check the plan (ordering, front rows, extend-add maps, the
// H -> front entry lists, level order) end to end -- a plan that factors and solves correctly here is the one the HIP
// kernels of csrc/direct_kernels.hip walk.  Test infrastructure only.  Built by tests/test_direct_plan.py with g++.
#include <cmath>
#include <cstring>

#include "../../total-lagrangian-fea_amd/csrc/mf_host.h"

static tlfea::MfPlan g_p;

extern "C" int mf_build(int N, const int* off, const int* cols, const double* x, const double* y, const double* z, int leaf,
                        long long max_doubles, long long* info) {
  if (!tlfea::mf_plan_build(N, off, cols, x, y, z, leaf, max_doubles, g_p)) return 1;
  info[0] = (long long)g_p.fronts.size();
  info[1] = g_p.n_levels();
  info[2] = g_p.L_total;
  info[3] = g_p.F_cap[0] + g_p.F_cap[1];
  info[4] = g_p.F_cap[2] + g_p.F_cap[3];
  info[5] = g_p.flops;
  info[6] = (long long)g_p.bsteps.size();
  info[7] = (long long)g_p.h_src.size();
  return 0;
}
extern "C" void mf_fetch(int* order, int* front_c0, int* front_c1, int* front_parent, int* front_depth, int* front_nrows) {
  std::copy(g_p.order.begin(), g_p.order.end(), order);
  for (size_t f = 0; f < g_p.fronts.size(); f++) {
    front_c0[f] = g_p.fronts[f].c0;
    front_c1[f] = g_p.fronts[f].c1;
    front_parent[f] = g_p.fronts[f].parent;
    front_depth[f] = g_p.fronts[f].depth;
    front_nrows[f] = g_p.fronts[f].nrows;
  }
}

// H: the engine's value layout (node row i: 9 off[i] + d * 3 deg + 3 k + e), b, x in the original DOF order
extern "C" int mf_cpu_factor_solve(const double* H, const double* b, double* xout) {
  using namespace tlfea;
  const MfPlan& P = g_p;
  std::vector<double> L((size_t)P.L_total, 0.0), Fb[4];
  for (int t = 0; t < 4; t++) Fb[t].assign((size_t)P.F_cap[t], 0.0);
  for (const MfBatch& B : P.batches) {
    std::vector<double>& W = Fb[B.wbuf];
    const std::vector<double>& Wc = Fb[B.cbuf];
    std::fill(W.begin() + B.F_base, W.begin() + B.F_base + B.F_doubles, 0.0);
    for (int t = B.hent_off; t < B.hent_off + B.hent_count; t++)
      for (int d = 0; d < 3; d++)
        for (int e = 0; e < 3; e++)
          W[(size_t)(P.h_dst[t] + d + (long long)e * P.h_dld[t])] = H[P.h_src[t] + (long long)d * P.h_sld[t] + e];
    for (int t = B.first; t < B.first + B.count; t++) {
      const MfFront& F = P.fronts[P.batch_fronts[t]];
      const long long m = 3LL * F.nrows, k = 3LL * (F.c1 - F.c0);
      double* A = W.data() + F.F_off;
      for (int s = 0; s < 2; s++) {
        if (F.child[s] < 0) continue;
        const MfFront& C = P.fronts[F.child[s]];
        const long long mc = 3LL * C.nrows, kc = 3LL * (C.c1 - C.c0);
        const double* U = Wc.data() + C.cF_off;
        const int* mp = P.map.data() + C.map_off;
        for (long long j = 0; j < mc - kc; j++)
          for (long long i = j; i < mc - kc; i++) {
            const long long pi = 3LL * mp[i / 3] + i % 3, pj = 3LL * mp[j / 3] + j % 3;
            if (pi < pj) return 3;
            A[pi + pj * m] += U[(C.c_k0 + i) + (C.c_k0 + j) * (long long)C.c_ld];
          }
      }
      for (long long j = 0; j < k; j++) {
        const double d = A[j + j * m];
        if (!(d > 0.0)) return 2;
        const double piv = std::sqrt(d);
        A[j + j * m] = piv;
        for (long long i = j + 1; i < m; i++) A[i + j * m] /= piv;
        for (long long c = j + 1; c < m; c++) {
          const double w = A[c + j * m];
          if (w == 0.0) continue;
          for (long long i = c; i < m; i++) A[i + c * m] -= A[i + j * m] * w;
        }
      }
      for (long long j = 0; j < k; j++)
        for (long long i = j; i < m; i++) L[(size_t)(F.L_off + i + j * m)] = A[i + j * m];
      if (B.push)  // the update matrix goes onto the stack (over its children's, which are consumed)
        for (long long j = 0; j < m - k; j++)
          for (long long i = j; i < m - k; i++) Fb[3][(size_t)(F.cF_off + i + j * (m - k))] = A[(k + i) + (k + j) * m];
    }
  }
  const int nl = P.n_levels();
  // forward: front vectors, children before parents
  const int n = 3 * P.N;
  std::vector<double> v((size_t)P.v_total, 0.0), y((size_t)n), xp((size_t)n);
  for (int l = 0; l < nl; l++)
    for (int t = P.level_off[l]; t < P.level_off[l + 1]; t++) {
      const MfFront& F = P.fronts[P.level_fronts[t]];
      const long long m = 3LL * F.nrows, k = 3LL * (F.c1 - F.c0);
      double* w = v.data() + F.v_off;
      for (long long i = 0; i < m; i++) w[i] = i < k ? b[3 * P.order[F.c0 + i / 3] + i % 3] : 0.0;
      for (int s = 0; s < 2; s++) {
        if (F.child[s] < 0) continue;
        const MfFront& C = P.fronts[F.child[s]];
        const long long mc = 3LL * C.nrows, kc = 3LL * (C.c1 - C.c0);
        const int* mp = P.map.data() + C.map_off;
        for (long long i = 0; i < mc - kc; i++) w[3LL * mp[i / 3] + i % 3] += v[(size_t)(C.v_off + kc + i)];
      }
      const double* Lf = L.data() + F.L_off;
      for (long long j = 0; j < k; j++) {
        w[j] /= Lf[j + j * m];
        for (long long i = j + 1; i < m; i++) w[i] -= Lf[i + j * m] * w[j];
      }
      for (long long i = 0; i < k; i++) y[(size_t)(3LL * F.c0 + i)] = w[i];
    }
  for (int l = nl - 1; l >= 0; l--)
    for (int t = P.level_off[l]; t < P.level_off[l + 1]; t++) {
      const MfFront& F = P.fronts[P.level_fronts[t]];
      const long long m = 3LL * F.nrows, k = 3LL * (F.c1 - F.c0);
      double* w = v.data() + F.v_off;
      for (long long i = 0; i < m; i++)
        w[i] = i < k ? y[(size_t)(3LL * F.c0 + i)] : xp[(size_t)(3LL * P.rows[(size_t)F.row_off + i / 3] + i % 3)];
      const double* Lf = L.data() + F.L_off;
      for (long long j = k - 1; j >= 0; j--) {
        double s = w[j];
        for (long long i = j + 1; i < m; i++) s -= Lf[i + j * m] * w[i];
        w[j] = s / Lf[j + j * m];
      }
      for (long long i = 0; i < k; i++) xp[(size_t)(3LL * F.c0 + i)] = w[i];
    }
  for (int p = 0; p < P.N; p++)
    for (int d = 0; d < 3; d++) xout[3 * P.order[p] + d] = xp[(size_t)3 * p + d];
  return 0;
}
