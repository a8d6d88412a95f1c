// Host-side set-up of the two-level p-multigrid preconditioner for T10 meshes (quadratic tets -> their linear
// vertex mesh).  No reference counterpart: the reference hands the Newton system to cuDSS (SyncedNewton.cu:995-1114);
// this is part of the device solve that replaces it ("Linear solve").
//
//   P  (fine N nodes <- coarse Nc vertex nodes):  vertex node <- itself;  mid-edge node <- 1/2 (a + b), its edge's ends
//   Hc = P^T H P  (Galerkin: SPD whenever H is; pattern = vertices sharing an element, the P1 adjacency)
//
// Everything is integer bookkeeping built once per mesh in a fixed order (bitwise-reproducible products): per fine
// node its (up to two) coarse parents, per coarse node its children (restriction), and per coarse 3x3 block the list
// of fine blocks that contribute to it with weights 1, 1/2, 1/4 (triple product by gather, no atomics).
#ifndef TLFEA_PMG_HOST_H_
#define TLFEA_PMG_HOST_H_
#include <algorithm>
#include <cstdint>
#include <vector>

namespace tlfea {

struct PmgHost {
  int N = 0, Nc = 0, nnz_c = 0;
  std::vector<int> par0, par1;               // [N] coarse ids of the parents (par1 == par0 for a vertex node)
  std::vector<int> c_off, c_cols, c_diagpos; // coarse node adjacency (CSR, sorted columns) + diagonal positions
  std::vector<int> cblk_row;                 // [nnz_c] coarse row of each coarse block
  std::vector<int> child_off, child;         // [Nc+1], fine ids: the vertex itself first, then its mid-edge nodes
  std::vector<float> child_w;                // 1 or 1/2
  std::vector<int> con_off;                  // [nnz_c+1] contributions of fine blocks to each coarse block
  std::vector<int> con_blk;                  // fine block index (off[i] + k), ascending within a coarse block
  std::vector<int> con_base, con_deg;        // of that fine block: 9 off[i] + 3 k (its place in H's layout) and deg(i)
  std::vector<float> con_w;                  // w_i * w_j
  std::vector<int> blk_row;                  // [nnz_f] fine row of each fine block
};

// T10 local edges of the mid-edge nodes 4..9 (FEAT10Data.cu:143, cpu_utils.cc:609-619)
static const int kT10Edge[6][2] = {{0, 1}, {1, 2}, {0, 2}, {0, 3}, {1, 3}, {2, 3}};

// conn: column-major [10][E]; off/cols: fine node adjacency (sorted).  Returns false when the mesh is not a conforming
// T10 mesh in the sense needed here (a node that is a vertex in one element and a mid-edge node in another, a mid-edge
// node with two different parent pairs, or a fine block whose parents are not coarse neighbours).
inline bool pmg_build(int N, int E, const int* conn, const int* off, const int* cols, PmgHost& o) {
  o = PmgHost();
  o.N = N;
  std::vector<char> is_vertex(N, 0), is_mid(N, 0);
  for (int e = 0; e < E; e++) {
    for (int a = 0; a < 4; a++) is_vertex[conn[(size_t)a * E + e]] = 1;
    for (int a = 4; a < 10; a++) is_mid[conn[(size_t)a * E + e]] = 1;
  }
  for (int n = 0; n < N; n++) {
    if (is_vertex[n] && is_mid[n]) return false;
    if (!is_vertex[n] && !is_mid[n]) is_vertex[n] = 1;  // a node no element uses: its own coarse node
  }
  std::vector<int> cid(N, -1);
  int Nc = 0;
  for (int n = 0; n < N; n++)
    if (is_vertex[n]) cid[n] = Nc++;
  o.Nc = Nc;
  o.par0.assign(N, -1);
  o.par1.assign(N, -1);
  for (int n = 0; n < N; n++)
    if (is_vertex[n]) o.par0[n] = o.par1[n] = cid[n];
  for (int e = 0; e < E; e++)
    for (int m = 0; m < 6; m++) {
      const int n = conn[(size_t)(4 + m) * E + e];
      int a = cid[conn[(size_t)kT10Edge[m][0] * E + e]], b = cid[conn[(size_t)kT10Edge[m][1] * E + e]];
      if (a > b) std::swap(a, b);
      if (a < 0 || a == b) return false;
      if (o.par0[n] < 0) {
        o.par0[n] = a;
        o.par1[n] = b;
      } else if (o.par0[n] != a || o.par1[n] != b) {
        return false;
      }
    }
  // coarse adjacency: vertices sharing an element
  {
    std::vector<std::vector<int>> rows((size_t)Nc);
    for (int e = 0; e < E; e++) {
      int v[4];
      for (int a = 0; a < 4; a++) v[a] = cid[conn[(size_t)a * E + e]];
      for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) rows[v[a]].push_back(v[b]);
    }
    o.c_off.assign((size_t)Nc + 1, 0);
    for (int i = 0; i < Nc; i++) {
      auto& r = rows[i];
      r.push_back(i);
      std::sort(r.begin(), r.end());
      r.erase(std::unique(r.begin(), r.end()), r.end());
      o.c_off[i + 1] = o.c_off[i] + (int)r.size();
    }
    o.nnz_c = o.c_off[Nc];
    o.c_cols.resize((size_t)o.nnz_c);
    o.c_diagpos.resize((size_t)Nc);
    o.cblk_row.resize((size_t)o.nnz_c);
    for (int i = 0; i < Nc; i++) {
      std::copy(rows[i].begin(), rows[i].end(), o.c_cols.begin() + o.c_off[i]);
      o.c_diagpos[i] = (int)(std::lower_bound(rows[i].begin(), rows[i].end(), i) - rows[i].begin());
      for (int k = o.c_off[i]; k < o.c_off[i + 1]; k++) o.cblk_row[k] = i;
    }
  }
  // children of every coarse node (restriction): itself first, then its mid-edge nodes in ascending fine id
  {
    o.child_off.assign((size_t)Nc + 1, 0);
    for (int n = 0; n < N; n++) {
      o.child_off[o.par0[n] + 1]++;
      if (o.par1[n] != o.par0[n]) o.child_off[o.par1[n] + 1]++;
    }
    for (int i = 0; i < Nc; i++) o.child_off[i + 1] += o.child_off[i];
    o.child.assign((size_t)o.child_off[Nc], 0);
    o.child_w.assign((size_t)o.child_off[Nc], 0.f);
    std::vector<int> cur(o.child_off.begin(), o.child_off.end() - 1);
    for (int n = 0; n < N; n++)  // vertices first so that slot 0 of every row is the vertex itself
      if (o.par1[n] == o.par0[n]) {
        o.child[cur[o.par0[n]]] = n;
        o.child_w[cur[o.par0[n]]++] = 1.f;
      }
    for (int n = 0; n < N; n++)
      if (o.par1[n] != o.par0[n])
        for (int p : {o.par0[n], o.par1[n]}) {
          o.child[cur[p]] = n;
          o.child_w[cur[p]++] = 0.5f;
        }
  }
  // contributions of fine blocks (i, j) to coarse blocks (I, J), I in parents(i), J in parents(j)
  const int nnz_f = off[N];
  o.blk_row.resize((size_t)nnz_f);
  std::vector<int> tgt((size_t)4 * nnz_f, -1);
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 2048) reduction(+ : bad)
  for (int i = 0; i < N; i++) {
    const int pi[2] = {o.par0[i], o.par1[i]};
    const int npi = pi[0] == pi[1] ? 1 : 2;
    for (int g = off[i]; g < off[i + 1]; g++) {
      o.blk_row[g] = i;
      const int j = cols[g];
      const int pj[2] = {o.par0[j], o.par1[j]};
      const int npj = pj[0] == pj[1] ? 1 : 2;
      for (int a = 0; a < npi; a++)
        for (int b = 0; b < npj; b++) {
          const int* r = o.c_cols.data() + o.c_off[pi[a]];
          const int deg = o.c_off[pi[a] + 1] - o.c_off[pi[a]];
          const int* p = std::lower_bound(r, r + deg, pj[b]);
          if (p == r + deg || *p != pj[b]) bad++;
          else tgt[(size_t)4 * g + 2 * a + b] = o.c_off[pi[a]] + (int)(p - r);
        }
    }
  }
  if (bad) return false;
  o.con_off.assign((size_t)o.nnz_c + 1, 0);
  for (size_t t = 0; t < tgt.size(); t++)
    if (tgt[t] >= 0) o.con_off[(size_t)tgt[t] + 1]++;
  for (int k = 0; k < o.nnz_c; k++) o.con_off[k + 1] += o.con_off[k];
  o.con_blk.assign((size_t)o.con_off[o.nnz_c], 0);
  o.con_w.assign((size_t)o.con_off[o.nnz_c], 0.f);
  o.con_base.assign((size_t)o.con_off[o.nnz_c], 0);
  o.con_deg.assign((size_t)o.con_off[o.nnz_c], 0);
  {
    std::vector<int> cur(o.con_off.begin(), o.con_off.end() - 1);
    for (int g = 0; g < nnz_f; g++) {  // ascending fine block index: fixed summation order
      const int i = o.blk_row[g], j = cols[g];
      const float wi = o.par0[i] == o.par1[i] ? 1.f : 0.5f, wj = o.par0[j] == o.par1[j] ? 1.f : 0.5f;
      for (int t = 0; t < 4; t++) {
        const int cb = tgt[(size_t)4 * g + t];
        if (cb < 0) continue;
        o.con_blk[cur[cb]] = g;
        o.con_base[cur[cb]] = 9 * off[i] + 3 * (g - off[i]);
        o.con_deg[cur[cb]] = off[i + 1] - off[i];
        o.con_w[cur[cb]++] = wi * wj;
      }
    }
  }
  return true;
}

// ---- the same two-level set-up for the ANCF kinds (3243 beams, 3443 shells) ---------------------------------------------
// An ANCF node carries four coefficient vectors (position r and the gradients r_u, r_v, r_w: coefficient 4 node + slot,
// ANCF3243DataFunc.cuh:212-215).  Coarse space = the POSITION coefficient of every node: P injects coarse node I into
// coefficient 4 I and leaves the gradient coefficients to the smoother, Hc = P^T H P is the sub-matrix of H on the
// position coefficients (pattern = nodes sharing an element, 1/16 of H's blocks).  The bookkeeping arrays are those of
// pmg_build with one child and one contributing fine block per coarse entity; par = -1 marks a coefficient without a
// coarse parent (the prolongation leaves it untouched).
inline bool pmg_build_ancf(int N, const int* off, const int* cols, PmgHost& o) {
  o = PmgHost();
  if (N <= 0 || N % 4 != 0) return false;
  o.N = N;
  const int Nc = N / 4;
  o.Nc = Nc;
  o.par0.assign(N, -1);
  o.par1.assign(N, -1);
  for (int I = 0; I < Nc; I++) o.par0[4 * I] = o.par1[4 * I] = I;
  o.c_off.assign((size_t)Nc + 1, 0);
  for (int I = 0; I < Nc; I++) {
    int cnt = 0;
    for (int g = off[4 * I]; g < off[4 * I + 1]; g++) cnt += cols[g] % 4 == 0;
    o.c_off[I + 1] = o.c_off[I] + cnt;
  }
  o.nnz_c = o.c_off[Nc];
  o.c_cols.resize((size_t)o.nnz_c);
  o.c_diagpos.assign((size_t)Nc, -1);
  o.cblk_row.resize((size_t)o.nnz_c);
  o.con_off.resize((size_t)o.nnz_c + 1);
  o.con_blk.resize((size_t)o.nnz_c);
  o.con_base.resize((size_t)o.nnz_c);
  o.con_deg.resize((size_t)o.nnz_c);
  o.con_w.assign((size_t)o.nnz_c, 1.f);
  for (int I = 0; I < Nc; I++) {
    int k = o.c_off[I];
    const int i = 4 * I, deg = off[i + 1] - off[i];
    for (int g = off[i]; g < off[i + 1]; g++) {
      if (cols[g] % 4) continue;
      const int J = cols[g] / 4;
      if (J == I) o.c_diagpos[I] = k - o.c_off[I];
      o.c_cols[k] = J;           // fine columns ascend, so do these
      o.cblk_row[k] = I;
      o.con_off[k] = k;
      o.con_blk[k] = g;
      o.con_base[k] = 9 * off[i] + 3 * (g - off[i]);
      o.con_deg[k] = deg;
      k++;
    }
    if (o.c_diagpos[I] < 0) return false;
  }
  o.con_off[o.nnz_c] = o.nnz_c;
  o.child_off.resize((size_t)Nc + 1);
  o.child.resize((size_t)Nc);
  o.child_w.assign((size_t)Nc, 1.f);
  for (int I = 0; I <= Nc; I++) o.child_off[I] = I;
  for (int I = 0; I < Nc; I++) o.child[I] = 4 * I;
  return true;
}

// ---- third level: aggregation of the vertex mesh with rigid-body modes ---------------------------------------------
// The vertex (P1) problem is still elasticity on the full mesh; a polynomial on it needs a degree that grows with the
// mesh (31 at config C).  Below it sits one more Galerkin level: greedy aggregates of a vertex and its neighbours
// (about 15 vertices), each carrying the six rigid-body modes of its members about their centroid,
//     u_i = t_A + w_A x (x_i - c_A)      (i in A),
// stored as TWO level-3 "nodes" per aggregate (2A: translation t_A, 2A+1: rotation w_A), so that the 3x3-block
// kernels of the other levels run unchanged on it.  Integer bookkeeping in a fixed order, like the level above.
struct AggHost {
  int Nc = 0, Na = 0, N3 = 0, nnz3 = 0, n_pairs = 0;
  std::vector<int> agg;                    // [Nc] aggregate of every vertex node
  std::vector<double> rvec;                // [3 Nc] x_i - c_A (zeroed where the aggregate's rotations are unusable)
  std::vector<int> active;                 // [Na] 1: rotation modes usable (members span 3-D)
  std::vector<int> mem_off, mem;           // members of every aggregate, ascending
  std::vector<int> off3, cols3, diag3;     // level-3 node adjacency (N3 = 2 Na rows, sorted columns)
  std::vector<int> pair_A, pair_pos, pair_B;  // aggregate pairs (A, B in adj(A)): A, position of B in adj(A), B
  std::vector<int> pcon_off, pcon_base, pcon_deg, pcon_i, pcon_j;  // vertex-level blocks feeding every pair
};

// c_off / c_cols: vertex adjacency (sorted, self included); X: [3][Nc] reference coordinates of the vertex nodes
inline bool agg_build(int Nc, const int* c_off, const int* c_cols, const double* X, AggHost& o) {
  o = AggHost();
  o.Nc = Nc;
  o.agg.assign((size_t)Nc, -1);
  int Na = 0;
  // pass 1: a vertex whose whole neighbourhood is free founds an aggregate with it
  for (int i = 0; i < Nc; i++) {
    bool free_nb = true;
    for (int k = c_off[i]; k < c_off[i + 1] && free_nb; k++) free_nb = o.agg[c_cols[k]] < 0;
    if (!free_nb) continue;
    for (int k = c_off[i]; k < c_off[i + 1]; k++) o.agg[c_cols[k]] = Na;
    Na++;
  }
  // pass 2: leftovers join the neighbouring aggregate of pass 1 they touch most often (ties: lowest id)
  {
    std::vector<int> joined(o.agg);
    std::vector<int> cnt;
    for (int i = 0; i < Nc; i++) {
      if (o.agg[i] >= 0) continue;
      int best = -1, best_n = 0;
      for (int k = c_off[i]; k < c_off[i + 1]; k++) {
        const int a = o.agg[c_cols[k]];
        if (a < 0) continue;
        int n = 0;
        for (int k2 = c_off[i]; k2 < c_off[i + 1]; k2++) n += o.agg[c_cols[k2]] == a;
        if (n > best_n || (n == best_n && a < best)) {
          best = a;
          best_n = n;
        }
      }
      joined[i] = best;
    }
    o.agg.swap(joined);
  }
  // pass 3: what is still free (a vertex all of whose neighbours were leftovers) founds small aggregates
  for (int i = 0; i < Nc; i++) {
    if (o.agg[i] >= 0) continue;
    o.agg[i] = Na;
    for (int k = c_off[i]; k < c_off[i + 1]; k++)
      if (o.agg[c_cols[k]] < 0) o.agg[c_cols[k]] = Na;
    Na++;
  }
  o.Na = Na;
  o.N3 = 2 * Na;
  if (Na < 1 || Na >= Nc) return false;
  // members, centroids, offsets from the centroid, usable rotations
  o.mem_off.assign((size_t)Na + 1, 0);
  for (int i = 0; i < Nc; i++) o.mem_off[o.agg[i] + 1]++;
  for (int a = 0; a < Na; a++) o.mem_off[a + 1] += o.mem_off[a];
  o.mem.resize((size_t)Nc);
  {
    std::vector<int> cur(o.mem_off.begin(), o.mem_off.end() - 1);
    for (int i = 0; i < Nc; i++) o.mem[cur[o.agg[i]]++] = i;
  }
  o.rvec.assign((size_t)3 * Nc, 0.0);
  o.active.assign((size_t)Na, 0);
  for (int a = 0; a < Na; a++) {
    double c[3] = {0, 0, 0};
    const int n = o.mem_off[a + 1] - o.mem_off[a];
    for (int t = o.mem_off[a]; t < o.mem_off[a + 1]; t++)
      for (int d = 0; d < 3; d++) c[d] += X[(size_t)d * Nc + o.mem[t]];
    for (int d = 0; d < 3; d++) c[d] /= n;
    double M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // rotational inertia sum(|r|^2 I - r r^T): singular for collinear members
    for (int t = o.mem_off[a]; t < o.mem_off[a + 1]; t++) {
      const int i = o.mem[t];
      double r[3];
      for (int d = 0; d < 3; d++) r[d] = o.rvec[(size_t)3 * i + d] = X[(size_t)d * Nc + i] - c[d];
      const double rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      for (int d = 0; d < 3; d++)
        for (int e = 0; e < 3; e++) M[d][e] += (d == e ? rr : 0.0) - r[d] * r[e];
    }
    const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                       M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    const double tr = M[0][0] + M[1][1] + M[2][2];
    o.active[a] = (n >= 3 && tr > 0.0 && det > 1e-6 * tr * tr * tr / 27.0) ? 1 : 0;
    if (!o.active[a])
      for (int t = o.mem_off[a]; t < o.mem_off[a + 1]; t++)
        for (int d = 0; d < 3; d++) o.rvec[(size_t)3 * o.mem[t] + d] = 0.0;
  }
  // aggregate adjacency and the pairs
  std::vector<int> aoff((size_t)Na + 1, 0), aadj;
  {
    std::vector<int> tmp;
    for (int a = 0; a < Na; a++) {
      tmp.clear();
      for (int t = o.mem_off[a]; t < o.mem_off[a + 1]; t++) {
        const int i = o.mem[t];
        for (int k = c_off[i]; k < c_off[i + 1]; k++) tmp.push_back(o.agg[c_cols[k]]);
      }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      aadj.insert(aadj.end(), tmp.begin(), tmp.end());
      aoff[a + 1] = (int)aadj.size();
    }
  }
  o.n_pairs = (int)aadj.size();
  o.pair_A.resize((size_t)o.n_pairs);
  o.pair_pos.resize((size_t)o.n_pairs);
  o.pair_B = aadj;
  for (int a = 0; a < Na; a++)
    for (int p = aoff[a]; p < aoff[a + 1]; p++) {
      o.pair_A[p] = a;
      o.pair_pos[p] = p - aoff[a];
    }
  // level-3 node pattern: row 2A+s holds 2B, 2B+1 for every B in adj(A)
  o.off3.assign((size_t)o.N3 + 1, 0);
  for (int a = 0; a < Na; a++)
    for (int s2 = 0; s2 < 2; s2++) o.off3[2 * a + s2 + 1] = 2 * (aoff[a + 1] - aoff[a]);
  for (int r = 0; r < o.N3; r++) o.off3[r + 1] += o.off3[r];
  o.nnz3 = o.off3[o.N3];
  o.cols3.resize((size_t)o.nnz3);
  o.diag3.resize((size_t)o.N3);
  for (int a = 0; a < Na; a++)
    for (int s2 = 0; s2 < 2; s2++) {
      const int r = 2 * a + s2;
      for (int p = aoff[a]; p < aoff[a + 1]; p++) {
        o.cols3[o.off3[r] + 2 * (p - aoff[a])] = 2 * aadj[p];
        o.cols3[o.off3[r] + 2 * (p - aoff[a]) + 1] = 2 * aadj[p] + 1;
        if (aadj[p] == a) o.diag3[r] = 2 * (p - aoff[a]) + s2;
      }
    }
  // contributions: every vertex-level block (i, j) feeds the pair (agg i, agg j), ascending block order
  o.pcon_off.assign((size_t)o.n_pairs + 1, 0);
  const int nnz2 = c_off[Nc];
  std::vector<int> blk_pair((size_t)nnz2);
  for (int i = 0; i < Nc; i++) {
    const int a = o.agg[i];
    for (int k = c_off[i]; k < c_off[i + 1]; k++) {
      const int b = o.agg[c_cols[k]];
      const int* lo = aadj.data() + aoff[a];
      const int* hi = aadj.data() + aoff[a + 1];
      const int pos = (int)(std::lower_bound(lo, hi, b) - lo);
      blk_pair[k] = aoff[a] + pos;
      o.pcon_off[(size_t)blk_pair[k] + 1]++;
    }
  }
  for (int p = 0; p < o.n_pairs; p++) o.pcon_off[p + 1] += o.pcon_off[p];
  o.pcon_base.resize((size_t)nnz2);
  o.pcon_deg.resize((size_t)nnz2);
  o.pcon_i.resize((size_t)nnz2);
  o.pcon_j.resize((size_t)nnz2);
  {
    std::vector<int> cur(o.pcon_off.begin(), o.pcon_off.end() - 1);
    for (int i = 0; i < Nc; i++)
      for (int k = c_off[i]; k < c_off[i + 1]; k++) {
        const int u = cur[blk_pair[k]]++;
        o.pcon_base[u] = 9 * c_off[i] + 3 * (k - c_off[i]);
        o.pcon_deg[u] = c_off[i + 1] - c_off[i];
        o.pcon_i[u] = i;
        o.pcon_j[u] = c_cols[k];
      }
  }
  return true;
}

// ---- third level on a REGULAR GRID OF BINS (round 3) -------------------------------------------------------------------
// Aggregate = the vertex nodes inside one cell of a regular grid over the body (cell size a few vertex spacings, agreed by
// all ranks).  Unlike the greedy aggregates above, the aggregate of a vertex is a function of its coordinates alone, so
// every rank of a partitioned mesh derives the SAME level-3 space without communicating, the level-3 pattern is the fixed
// 27-cell stencil of the grid, and the level is small enough to be REPLICATED: every rank sums its owned rows' share of
// H3 = P2^T Hc P2 and of the restricted residual with one all-reduce each and runs the level-3 polynomial redundantly,
// with no exchange inside it.  Empty cells stay in the numbering as inert identity rows.
struct BinGrid {
  double origin[3] = {0, 0, 0};
  double size = 1.0;
  int dims[3] = {1, 1, 1};
  int cells() const { return dims[0] * dims[1] * dims[2]; }
};

// grid over a bounding box: cell size = factor x mean vertex spacing, never below the longest vertex edge (so that
// neighbours sit in adjacent cells).  Multi-GPU: the caller reduces (lo, hi, sum_len, n_len, max_len) over ranks first.
inline BinGrid bin_grid(const double lo[3], const double hi[3], double mean_len, double max_len, double factor) {
  BinGrid g;
  g.size = std::max(factor * mean_len, 1.0001 * max_len);
  for (int d = 0; d < 3; d++) {
    g.origin[d] = lo[d] - 1e-9 * g.size;
    g.dims[d] = std::max(1, (int)std::floor((hi[d] - g.origin[d]) / g.size) + 1);
  }
  return g;
}

inline int bin_of(const BinGrid& g, double x, double y, double z) {
  const double p[3] = {x, y, z};
  int c[3];
  for (int d = 0; d < 3; d++) c[d] = std::min(g.dims[d] - 1, std::max(0, (int)std::floor((p[d] - g.origin[d]) / g.size)));
  return (c[2] * g.dims[1] + c[1]) * g.dims[0] + c[0];
}

// per cell over the OWNED vertices (owned == nullptr: all): count, sum x (3), sum x x^T (xx, xy, xz, yy, yz, zz)
inline void bin_moments(int Nc, const double* X, const BinGrid& g, const char* owned, std::vector<int>& agg,
                        std::vector<double>& mom) {
  agg.resize((size_t)Nc);
  mom.assign((size_t)10 * g.cells(), 0.0);
  for (int i = 0; i < Nc; i++) {
    const double x = X[i], y = X[(size_t)Nc + i], z = X[(size_t)2 * Nc + i];
    const int a = agg[i] = bin_of(g, x, y, z);
    if (owned && !owned[i]) continue;
    double* m = mom.data() + (size_t)10 * a;
    m[0] += 1.0; m[1] += x; m[2] += y; m[3] += z;
    m[4] += x * x; m[5] += x * y; m[6] += x * z; m[7] += y * y; m[8] += y * z; m[9] += z * z;
  }
}

// the level from the (globally summed) moments: rows = owned vertices only, pattern = the grid's 27-cell stencil
inline bool agg_build_bins(int Nc, const int* c_off, const int* c_cols, const double* X, const BinGrid& g,
                           const std::vector<int>& agg, const std::vector<double>& mom, const char* owned, AggHost& o) {
  o = AggHost();
  o.Nc = Nc;
  const int Na = g.cells();
  o.Na = Na;
  o.N3 = 2 * Na;
  o.agg = agg;
  // members (restriction): owned vertices of every cell, ascending
  o.mem_off.assign((size_t)Na + 1, 0);
  for (int i = 0; i < Nc; i++)
    if (!owned || owned[i]) o.mem_off[agg[i] + 1]++;
  for (int a = 0; a < Na; a++) o.mem_off[a + 1] += o.mem_off[a];
  o.mem.resize((size_t)o.mem_off[Na]);
  {
    std::vector<int> cur(o.mem_off.begin(), o.mem_off.end() - 1);
    for (int i = 0; i < Nc; i++)
      if (!owned || owned[i]) o.mem[cur[agg[i]]++] = i;
  }
  // centroids and usable rotations from the moments (the same numbers on every rank); active: 1 rotations usable,
  // 0 not (collinear / too few members), -1 empty cell
  std::vector<double> cen((size_t)3 * Na, 0.0);
  o.active.assign((size_t)Na, -1);
  for (int a = 0; a < Na; a++) {
    const double* m = mom.data() + (size_t)10 * a;
    const double n = m[0];
    if (n < 0.5) continue;
    const double c[3] = {m[1] / n, m[2] / n, m[3] / n};
    for (int d = 0; d < 3; d++) cen[(size_t)3 * a + d] = c[d];
    // S = sum (x - c)(x - c)^T; rotational inertia M = tr(S) I - S
    const double S[3][3] = {{m[4] - n * c[0] * c[0], m[5] - n * c[0] * c[1], m[6] - n * c[0] * c[2]},
                            {m[5] - n * c[0] * c[1], m[7] - n * c[1] * c[1], m[8] - n * c[1] * c[2]},
                            {m[6] - n * c[0] * c[2], m[8] - n * c[1] * c[2], m[9] - n * c[2] * c[2]}};
    const double trS = S[0][0] + S[1][1] + S[2][2];
    double M[3][3];
    for (int d = 0; d < 3; d++)
      for (int e = 0; e < 3; e++) M[d][e] = (d == e ? trS : 0.0) - S[d][e];
    const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                       M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    const double tr = M[0][0] + M[1][1] + M[2][2];
    o.active[a] = (n >= 2.5 && tr > 0.0 && det > 1e-6 * tr * tr * tr / 27.0) ? 1 : 0;
  }
  o.rvec.assign((size_t)3 * Nc, 0.0);
  for (int i = 0; i < Nc; i++) {
    const int a = agg[i];
    if (o.active[a] != 1) continue;
    for (int d = 0; d < 3; d++) o.rvec[(size_t)3 * i + d] = X[(size_t)d * Nc + i] - cen[(size_t)3 * a + d];
  }
  // adjacency: the 27-cell stencil, clipped, ascending cell index -- identical on every rank
  std::vector<int> aoff((size_t)Na + 1, 0), aadj;
  aadj.reserve((size_t)27 * Na);
  for (int a = 0; a < Na; a++) {
    const int cx = a % g.dims[0], cy = (a / g.dims[0]) % g.dims[1], cz = a / (g.dims[0] * g.dims[1]);
    for (int dz = -1; dz <= 1; dz++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          const int x = cx + dx, y = cy + dy, z = cz + dz;
          if (x < 0 || y < 0 || z < 0 || x >= g.dims[0] || y >= g.dims[1] || z >= g.dims[2]) continue;
          aadj.push_back((z * g.dims[1] + y) * g.dims[0] + x);
        }
    aoff[a + 1] = (int)aadj.size();
  }
  o.n_pairs = (int)aadj.size();
  o.pair_A.resize((size_t)o.n_pairs);
  o.pair_pos.resize((size_t)o.n_pairs);
  o.pair_B = aadj;
  for (int a = 0; a < Na; a++)
    for (int p = aoff[a]; p < aoff[a + 1]; p++) {
      o.pair_A[p] = a;
      o.pair_pos[p] = p - aoff[a];
    }
  o.off3.assign((size_t)o.N3 + 1, 0);
  for (int a = 0; a < Na; a++)
    for (int s2 = 0; s2 < 2; s2++) o.off3[2 * a + s2 + 1] = 2 * (aoff[a + 1] - aoff[a]);
  for (int r = 0; r < o.N3; r++) o.off3[r + 1] += o.off3[r];
  o.nnz3 = o.off3[o.N3];
  o.cols3.resize((size_t)o.nnz3);
  o.diag3.resize((size_t)o.N3);
  for (int a = 0; a < Na; a++)
    for (int s2 = 0; s2 < 2; s2++) {
      const int r = 2 * a + s2;
      for (int p = aoff[a]; p < aoff[a + 1]; p++) {
        o.cols3[o.off3[r] + 2 * (p - aoff[a])] = 2 * aadj[p];
        o.cols3[o.off3[r] + 2 * (p - aoff[a]) + 1] = 2 * aadj[p] + 1;
        if (aadj[p] == a) o.diag3[r] = 2 * (p - aoff[a]) + s2;
      }
    }
  // contributions: the vertex-level blocks (i, j) of OWNED rows i feed the pair (cell of i, cell of j)
  o.pcon_off.assign((size_t)o.n_pairs + 1, 0);
  const int nnz2 = c_off[Nc];
  std::vector<int> blk_pair((size_t)nnz2, -1);
  int n_con = 0;
  for (int i = 0; i < Nc; i++) {
    if (owned && !owned[i]) continue;
    const int a = agg[i];
    for (int k = c_off[i]; k < c_off[i + 1]; k++) {
      const int b = agg[c_cols[k]];
      const int* lo = aadj.data() + aoff[a];
      const int* hi = aadj.data() + aoff[a + 1];
      const int* pp = std::lower_bound(lo, hi, b);
      if (pp == hi || *pp != b) return false;  // a vertex edge longer than a cell: the grid is too fine
      blk_pair[k] = aoff[a] + (int)(pp - lo);
      o.pcon_off[(size_t)blk_pair[k] + 1]++;
      n_con++;
    }
  }
  for (int p = 0; p < o.n_pairs; p++) o.pcon_off[p + 1] += o.pcon_off[p];
  o.pcon_base.resize((size_t)std::max(1, n_con));
  o.pcon_deg.resize((size_t)std::max(1, n_con));
  o.pcon_i.resize((size_t)std::max(1, n_con));
  o.pcon_j.resize((size_t)std::max(1, n_con));
  {
    std::vector<int> cur(o.pcon_off.begin(), o.pcon_off.end() - 1);
    for (int i = 0; i < Nc; i++)
      for (int k = c_off[i]; k < c_off[i + 1]; k++) {
        if (blk_pair[k] < 0) continue;
        const int u = cur[blk_pair[k]]++;
        o.pcon_base[u] = 9 * c_off[i] + 3 * (k - c_off[i]);
        o.pcon_deg[u] = c_off[i + 1] - c_off[i];
        o.pcon_i[u] = i;
        o.pcon_j[u] = c_cols[k];
      }
  }
  return true;
}

}  // namespace tlfea
#endif
