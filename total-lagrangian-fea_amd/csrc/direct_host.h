// direct_host.h -- host set-up of the sparse direct solve (tlfea_linsolve_opts.method = 1): what the reference gets from
// cuDSS's ANALYSIS phase (SyncedNewton.cu:995-1029) -- a fill-reducing ordering and the pattern of the Cholesky factor --
// computed once per mesh so that rocSOLVER's re-factorisation path (rocsolver_dcsrrf_refactchol / _solve, the ROCm
// counterpart of cuDSS REFACTORIZATION + SOLVE, SyncedNewton.cu:1103-1114) can run every Newton iteration on the device.
//
//   ordering : nested dissection of the NODE graph (3x3 blocks stay together) by recursive coordinate bisection -- the
//              mesh's reference coordinates give the cut planes, the graph gives the vertex separators
//   symbolic : elimination tree (Liu) + row patterns of L by walking the tree from every lower neighbour (the row
//              subtrees), on the node graph; expanded to DOF level with full 3x3 blocks below the diagonal and the
//              lower triangle of the diagonal blocks
// Integer work only.  Meant for the sizes a direct solve is sensible at on one GPU (BASELINE configs A, B, the
// TetGen meshes: up to ~150 k DOF); the caller bounds the factor's size.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <vector>

namespace tlfea {

struct DirectHost {
  int n = 0;                   // DOFs (3 N)
  std::vector<int> perm;       // [n] new -> old DOF (the columns of Q)
  std::vector<int> ptrT, indT; // lower triangle (with diagonal) of the factor of Q^T H Q, CSR, sorted columns
  long long nnz_node = 0;      // node-level entries of L below the diagonal
};

// off/cols: node adjacency (sorted columns, diagonal included), x/y/z: coordinates of the N nodes.
// Returns false when the factor would exceed max_nnz entries (DOF level).
inline bool direct_symbolic(int N, const int* off, const int* cols, const double* x, const double* y, const double* z,
                            long long max_nnz, DirectHost& out) {
  out = DirectHost();
  if (N <= 0) return false;
  // ---- nested dissection ---------------------------------------------------------------------------------------
  std::vector<int> order;  // new -> old node
  order.reserve((size_t)N);
  std::vector<int> mark((size_t)N, -1);
  int stamp = 0;
  const double* xyz[3] = {x, y, z};
  std::function<void(std::vector<int>&)> nd = [&](std::vector<int>& nodes) {
    if (nodes.size() <= 64) {
      for (int v : nodes) order.push_back(v);
      return;
    }
    int axis = 0;
    double best = -1.0;
    for (int a = 0; a < 3; a++) {
      double lo = xyz[a][nodes[0]], hi = lo;
      for (int v : nodes) {
        lo = std::min(lo, xyz[a][v]);
        hi = std::max(hi, xyz[a][v]);
      }
      if (hi - lo > best) {
        best = hi - lo;
        axis = a;
      }
    }
    const size_t half = nodes.size() / 2;
    const double* c = xyz[axis];
    std::nth_element(nodes.begin(), nodes.begin() + half, nodes.end(),
                     [&](int a, int b) { return c[a] < c[b] || (c[a] == c[b] && a < b); });
    const int sl = stamp++, sr = stamp++;
    for (size_t t = 0; t < nodes.size(); t++) mark[nodes[t]] = t < half ? sl : sr;
    // vertex separator: the nodes of the right part that touch the left part
    std::vector<int> left(nodes.begin(), nodes.begin() + half), right, sep;
    for (size_t t = half; t < nodes.size(); t++) {
      const int v = nodes[t];
      bool touches = false;
      for (int k = off[v]; k < off[v + 1] && !touches; k++) touches = mark[cols[k]] == sl;
      (touches ? sep : right).push_back(v);
    }
    std::vector<int>().swap(nodes);
    nd(left);
    nd(right);
    for (int v : sep) order.push_back(v);
  };
  {
    std::vector<int> all((size_t)N);
    for (int i = 0; i < N; i++) all[i] = i;
    nd(all);
  }
  if ((int)order.size() != N) return false;
  std::vector<int> inv((size_t)N);
  for (int k = 0; k < N; k++) inv[order[k]] = k;
  // ---- elimination tree of the permuted graph (Liu, with path compression) ------------------------------------------
  std::vector<int> parent((size_t)N, -1), anc((size_t)N, -1);
  for (int k = 0; k < N; k++) {
    const int v = order[k];
    for (int t = off[v]; t < off[v + 1]; t++) {
      int j = inv[cols[t]];
      while (j != -1 && j < k) {
        const int next = anc[j];
        anc[j] = k;
        if (next == -1) parent[j] = k;
        j = next;
      }
    }
  }
  // ---- row patterns of L (node level): the row subtrees ----------------------------------------------------------------
  std::vector<long long> rptr((size_t)N + 1, 0);
  std::vector<int> rcol;
  std::fill(mark.begin(), mark.end(), -1);
  for (int k = 0; k < N; k++) {
    const int v = order[k];
    mark[k] = k;
    const size_t start = rcol.size();
    for (int t = off[v]; t < off[v + 1]; t++) {
      int j = inv[cols[t]];
      if (j >= k) continue;
      while (mark[j] != k) {
        rcol.push_back(j);
        mark[j] = k;
        j = parent[j];
      }
    }
    std::sort(rcol.begin() + start, rcol.end());
    rptr[k + 1] = (long long)rcol.size();
    if (9 * rptr[k + 1] + 6LL * (k + 1) > max_nnz) return false;
  }
  out.nnz_node = rptr[N];
  // ---- DOF level -------------------------------------------------------------------------------------------------------
  const long long nnzT = 9 * rptr[N] + 6LL * N;
  if (nnzT > max_nnz || nnzT >= (1LL << 31)) return false;
  out.n = 3 * N;
  out.perm.resize((size_t)3 * N);
  for (int k = 0; k < N; k++)
    for (int c = 0; c < 3; c++) out.perm[3 * (size_t)k + c] = 3 * order[k] + c;
  out.ptrT.resize((size_t)3 * N + 1);
  out.indT.resize((size_t)nnzT);
  long long w = 0;
  out.ptrT[0] = 0;
  for (int k = 0; k < N; k++)
    for (int r = 0; r < 3; r++) {
      for (long long t = rptr[k]; t < rptr[k + 1]; t++) {
        const int j = rcol[(size_t)t];
        out.indT[(size_t)w++] = 3 * j;
        out.indT[(size_t)w++] = 3 * j + 1;
        out.indT[(size_t)w++] = 3 * j + 2;
      }
      for (int c = 0; c <= r; c++) out.indT[(size_t)w++] = 3 * k + c;
      out.ptrT[3 * (size_t)k + r + 1] = (int)w;
    }
  return true;
}

}  // namespace tlfea
