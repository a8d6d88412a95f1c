// mf_host.h -- host plan of the engine's own sparse direct solve (tlfea_linsolve_opts.method = 1): what the reference
// gets from cuDSS's ANALYSIS phase (SyncedNewton.cu:995-1029), computed once per mesh; the device then runs
// REFACTORIZATION + SOLVE (SyncedNewton.cu:1103-1114) every Newton iteration with the kernels of direct_kernels.hip.
//
// Multifrontal Cholesky on the NODE graph (3x3 blocks stay together):
//   ordering : nested dissection by recursive coordinate bisection -- the mesh's reference coordinates give the cut
//              planes, the graph gives the vertex separators.  The dissection tree IS the assembly tree: every leaf set
//              and every separator is one FRONT whose own nodes are contiguous in the new order (left subtree, right
//              subtree, separator), so children sit exactly one level below their parent and a level's fronts are
//              independent: the device works level by level, deepest first, all fronts of a level in one launch.
//   front    : dense, lower triangle used; rows = own nodes, then the ancestors' nodes the subtree reaches
//              (sorted): F = [F11 F21; . F22].  Partial factorisation F11 = L11 L11^T, L21 = F21 L11^-T leaves the
//              update matrix U = F22 - L21 L21^T in place, which the parent adds into its own front through `map`
//              (extend-add, child 0 then child 1: fixed order, no atomics, bitwise reproducible).
//   storage  : L panels (3 rows x 3 own nodes, column-major) for the whole tree; front workspaces for two adjacent
//              levels (a level reads its children's U from the other buffer); 64-bit offsets throughout.
// Integer work only.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <vector>

namespace tlfea {

constexpr int kMfNB = 48;  // block columns eliminated per panel step (a multiple of the 3 DOFs of a node)

struct MfFront {
  int c0 = 0, c1 = 0;            // own nodes: new positions [c0, c1)
  int child[2] = {-1, -1};
  int parent = -1, depth = 0;
  int nrows = 0;                 // node rows of the front (the c1 - c0 own ones first)
  long long row_off = 0;         // rows[row_off .. + nrows): new positions, ascending
  long long map_off = 0;         // map[map_off .. + nrows - (c1 - c0)): row index inside the parent's front
  long long L_off = 0;           // doubles: panel (3 nrows) x (3 (c1 - c0)), column-major
  long long F_off = 0;           // doubles inside the workspace of depth parity (depth & 1): (3 nrows)^2, column-major
  long long v_off = 0;           // doubles inside the solve's front vectors: 3 nrows
};

struct MfLevelStep {  // one panel step of a level: fronts [0, n_active) of the level's list still have columns at j0
  int j0, n_active, max_below;  // max_below: largest number of DOF rows below the panel among them
};

struct MfPlan {
  int N = 0;
  std::vector<int> order, inv;          // new -> old node, old -> new
  std::vector<MfFront> fronts;          // children before parents
  std::vector<int> rows, map;
  std::vector<int> level_off;           // [n_levels + 1] into level_fronts; level 0 = deepest
  std::vector<int> level_fronts;        // fronts of a level, most own columns first
  std::vector<int> step_off;            // [n_levels + 1] into steps
  std::vector<MfLevelStep> steps;
  std::vector<long long> level_F;       // doubles of front workspace a level needs
  // H -> fronts: one entry per lower node block of the permuted matrix (level order)
  std::vector<int> hent_off;            // [n_levels + 1]
  std::vector<long long> h_src, h_dst;  // first double of the 3x3 block in H's values / in the level's workspace
  std::vector<int> h_sld, h_dld;        // leading dimensions (H: row stride 3 deg, front: 3 nrows)
  long long L_total = 0, F_cap[2] = {0, 0}, v_total = 0, flops = 0;
  int n_levels() const { return (int)level_off.size() - 1; }
};

// off/cols: node adjacency (sorted columns, diagonal included), x/y/z: coordinates of the N nodes; leaf: nodes below
// which a part is not cut further.  Returns false when L or the workspaces would exceed max_doubles.
inline bool mf_plan_build(int N, const int* off, const int* cols, const double* x, const double* y, const double* z,
                          int leaf, long long max_doubles, MfPlan& P) {
  P = MfPlan();
  if (N <= 0) return false;
  P.N = N;
  leaf = std::max(4, leaf);
  // ---- nested dissection, recording the tree -----------------------------------------------------------------------
  std::vector<int>& order = P.order;
  order.reserve((size_t)N);
  std::vector<int> mark((size_t)N, -1);
  int stamp = 0;
  const double* xyz[3] = {x, y, z};
  std::function<int(std::vector<int>&)> nd = [&](std::vector<int>& nodes) -> int {
    MfFront f;
    if ((int)nodes.size() <= leaf) {
      f.c0 = (int)order.size();
      for (int v : nodes) order.push_back(v);
      f.c1 = (int)order.size();
      P.fronts.push_back(f);
      return (int)P.fronts.size() - 1;
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
    size_t half = nodes.size() / 2;
    const double* c = xyz[axis];
    std::nth_element(nodes.begin(), nodes.begin() + half, nodes.end(),
                     [&](int a, int b) { return c[a] < c[b] || (c[a] == c[b] && a < b); });
    {
      // nodes that share the median coordinate (a plane of a structured mesh) go to ONE side -- a cut through the plane
      // makes a staircase separator twice as large -- as long as the parts stay within 30 % / 70 %
      const double cv = c[nodes[half]];
      size_t nlt = 0, nle = 0;
      for (int v : nodes) {
        nlt += c[v] < cv;
        nle += c[v] <= cv;
      }
      const size_t lo = (size_t)(0.3 * nodes.size()), hi = (size_t)(0.7 * nodes.size());
      const size_t da = nlt > half ? nlt - half : half - nlt, db = nle > half ? nle - half : half - nle;
      size_t pick = da <= db ? nlt : nle;
      if (pick < lo || pick > hi) pick = (da <= db ? nle : nlt);
      if (pick >= lo && pick <= hi && pick != half) {
        const bool strict = pick == nlt;
        std::partition(nodes.begin(), nodes.end(), [&](int v) { return strict ? c[v] < cv : c[v] <= cv; });
        half = pick;
      }
    }
    const int sl = stamp++, sr = stamp++;
    for (size_t t = 0; t < nodes.size(); t++) mark[nodes[t]] = t < half ? sl : sr;
    // vertex separator: the nodes of one part that touch the other part -- whichever side gives the smaller set
    std::vector<int> left, right, sep;
    {
      size_t nr = 0, nl_ = 0;
      for (size_t t = 0; t < nodes.size(); t++) {
        const int v = nodes[t], other = t < half ? sr : sl;
        bool touches = false;
        for (int k = off[v]; k < off[v + 1] && !touches; k++) touches = mark[cols[k]] == other;
        if (touches) (t < half ? nl_ : nr)++;
      }
      const bool from_right = nr <= nl_;
      for (size_t t = 0; t < nodes.size(); t++) {
        const int v = nodes[t];
        const bool in_left = t < half;
        bool touches = false;
        if (in_left != from_right) {
          const int other = in_left ? sr : sl;
          for (int k = off[v]; k < off[v + 1] && !touches; k++) touches = mark[cols[k]] == other;
        }
        (touches ? sep : (in_left ? left : right)).push_back(v);
      }
    }
    std::vector<int>().swap(nodes);
    std::sort(sep.begin(), sep.end());
    const int a = left.empty() ? -1 : nd(left);
    const int b = right.empty() ? -1 : nd(right);
    f.c0 = (int)order.size();
    for (int v : sep) order.push_back(v);
    f.c1 = (int)order.size();
    f.child[0] = a;
    f.child[1] = b;
    P.fronts.push_back(f);
    const int id = (int)P.fronts.size() - 1;
    if (a >= 0) P.fronts[a].parent = id;
    if (b >= 0) P.fronts[b].parent = id;
    return id;
  };
  {
    std::vector<int> all((size_t)N);
    for (int i = 0; i < N; i++) all[i] = i;
    nd(all);
  }
  if ((int)order.size() != N) return false;
  P.inv.resize((size_t)N);
  for (int k = 0; k < N; k++) P.inv[order[k]] = k;
  const int nf = (int)P.fronts.size();
  for (int f = nf - 1; f >= 0; f--)  // parents come after their children: depths from the root down
    if (P.fronts[f].parent >= 0) P.fronts[f].depth = P.fronts[P.fronts[f].parent].depth + 1;
  // ---- row structure of every front, children first ------------------------------------------------------------------
  std::vector<int> tmp;
  for (int f = 0; f < nf; f++) {
    MfFront& F = P.fronts[f];
    tmp.clear();
    for (int p = F.c0; p < F.c1; p++) {
      const int v = order[p];
      for (int t = off[v]; t < off[v + 1]; t++) {
        const int q = P.inv[cols[t]];
        if (q >= F.c1) tmp.push_back(q);
      }
    }
    for (int s = 0; s < 2; s++) {
      if (F.child[s] < 0) continue;
      const MfFront& C = P.fronts[F.child[s]];
      for (int t = C.c1 - C.c0; t < C.nrows; t++) {
        const int q = P.rows[(size_t)C.row_off + t];
        if (q >= F.c1) tmp.push_back(q);
        else if (q < F.c0) return false;  // a child reaches outside its ancestors: not a separator tree
      }
    }
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    F.row_off = (long long)P.rows.size();
    F.nrows = F.c1 - F.c0 + (int)tmp.size();
    for (int p = F.c0; p < F.c1; p++) P.rows.push_back(p);
    P.rows.insert(P.rows.end(), tmp.begin(), tmp.end());
    if (F.parent < 0 && !tmp.empty()) return false;
  }
  // ---- a child's update rows inside its parent's front -----------------------------------------------------------------
  for (int f = 0; f < nf; f++) {
    MfFront& C = P.fronts[f];
    C.map_off = (long long)P.map.size();
    if (C.parent < 0) continue;
    const MfFront& F = P.fronts[C.parent];
    const int* pr = P.rows.data() + F.row_off;
    for (int t = C.c1 - C.c0; t < C.nrows; t++) {
      const int q = P.rows[(size_t)C.row_off + t];
      const int* it = std::lower_bound(pr, pr + F.nrows, q);
      if (it == pr + F.nrows || *it != q) return false;
      P.map.push_back((int)(it - pr));
    }
  }
  // ---- levels (deepest first), storage -------------------------------------------------------------------------------
  int maxd = 0;
  for (const MfFront& F : P.fronts) maxd = std::max(maxd, F.depth);
  const int nl = maxd + 1;
  P.level_off.assign((size_t)nl + 1, 0);
  for (const MfFront& F : P.fronts) P.level_off[(size_t)(maxd - F.depth) + 1]++;
  for (int l = 0; l < nl; l++) P.level_off[l + 1] += P.level_off[l];
  P.level_fronts.resize((size_t)nf);
  {
    std::vector<int> w(P.level_off.begin(), P.level_off.end() - 1);
    for (int f = 0; f < nf; f++) P.level_fronts[(size_t)w[maxd - P.fronts[f].depth]++] = f;
  }
  P.level_F.assign((size_t)nl, 0);
  P.step_off.assign((size_t)nl + 1, 0);
  for (int l = 0; l < nl; l++) {
    int* b = P.level_fronts.data() + P.level_off[l];
    int* e = P.level_fronts.data() + P.level_off[l + 1];
    std::sort(b, e, [&](int a, int c) {
      const int ka = P.fronts[a].c1 - P.fronts[a].c0, kc = P.fronts[c].c1 - P.fronts[c].c0;
      return ka > kc || (ka == kc && a < c);
    });
    long long fo = 0;
    for (int* it = b; it != e; ++it) {
      MfFront& F = P.fronts[*it];
      const long long m = 3LL * F.nrows, k = 3LL * (F.c1 - F.c0);
      F.F_off = fo;
      fo += m * m;
      F.L_off = P.L_total;
      P.L_total += m * k;
      F.v_off = P.v_total;
      P.v_total += m;
      P.flops += k * k * k / 3 + (m - k) * k * k + (m - k) * (m - k) * k;
      if (P.L_total > max_doubles) return false;
    }
    P.level_F[l] = fo;
    const int par = (maxd - l) & 1;
    P.F_cap[par] = std::max(P.F_cap[par], fo);
    const int kmax = b == e ? 0 : 3 * (P.fronts[*b].c1 - P.fronts[*b].c0);
    for (int j0 = 0; j0 < kmax; j0 += kMfNB) {
      MfLevelStep st{j0, 0, 0};
      for (int* it = b; it != e; ++it) {
        const MfFront& F = P.fronts[*it];
        const int k = 3 * (F.c1 - F.c0), m = 3 * F.nrows;
        if (k <= j0) break;
        st.n_active++;
        st.max_below = std::max(st.max_below, m - j0 - std::min(kMfNB, k - j0));
      }
      P.steps.push_back(st);
    }
    P.step_off[l + 1] = (int)P.steps.size();
  }
  if (P.L_total + P.F_cap[0] + P.F_cap[1] > max_doubles) return false;
  // ---- H's lower node blocks -> front entries --------------------------------------------------------------------------
  P.hent_off.assign((size_t)nl + 1, 0);
  for (int l = 0; l < nl; l++) {
    for (int t = P.level_off[l]; t < P.level_off[l + 1]; t++) {
      const MfFront& F = P.fronts[P.level_fronts[t]];
      const int* pr = P.rows.data() + F.row_off;
      const int ld = 3 * F.nrows;
      for (int p = F.c0; p < F.c1; p++) {
        const int c = order[p];
        for (int u = off[c]; u < off[c + 1]; u++) {
          const int q = P.inv[cols[u]];
          if (q < p) continue;
          const int* it = std::lower_bound(pr, pr + F.nrows, q);
          if (it == pr + F.nrows || *it != q) return false;
          const int ri = (int)(it - pr), i = cols[u];
          // block (row node i, column node c) of H: row i's values start at 9 off[i], row stride 3 deg(i)
          const int* ci = std::lower_bound(cols + off[i], cols + off[i + 1], c);
          if (ci == cols + off[i + 1] || *ci != c) return false;  // pattern not symmetric
          P.h_src.push_back(9LL * off[i] + 3LL * (ci - (cols + off[i])));
          P.h_sld.push_back(3 * (off[i + 1] - off[i]));
          P.h_dst.push_back(F.F_off + 3LL * ri + 3LL * (p - F.c0) * ld);
          P.h_dld.push_back(ld);
        }
      }
    }
    P.hent_off[l + 1] = (int)P.h_src.size();
  }
  return true;
}

}  // namespace tlfea
