// mf_host.h -- host plan of the engine's own sparse direct solve (tlfea_linsolve_opts.method = 1): what the reference
// gets from cuDSS's ANALYSIS phase (SyncedNewton.cu:995-1029), computed once per mesh; the device then runs
// REFACTORIZATION + SOLVE (SyncedNewton.cu:1103-1114) every Newton iteration with the kernels of direct_kernels.hip.
//
// Multifrontal Cholesky on the NODE graph (3x3 blocks stay together):
//   ordering : nested dissection by recursive coordinate bisection -- the mesh's reference coordinates give the cut
//              planes (snapped to coordinate planes of structured meshes; the axis with the smallest separator among the
//              near-longest ones), the graph gives the vertex separators (from the smaller side of the cut).
//              The dissection tree IS the assembly tree: every leaf set
//              and every separator is one FRONT whose own nodes are contiguous in the new order (left subtree, right
//              subtree, separator), so children sit exactly one level below their parent and a level's fronts are
//              independent: the device works level by level, deepest first, all fronts of a level in one launch.
//   front    : dense, lower triangle used; rows = own nodes, then the ancestors' nodes the subtree reaches
//              (sorted): F = [F11 F21; . F22].  Partial factorisation F11 = L11 L11^T, L21 = F21 L11^-T leaves the
//              update matrix U = F22 - L21 L21^T in place, which the parent adds into its own front through `map`
//              (extend-add, child 0 then child 1: fixed order, no atomics, bitwise reproducible).
//   storage  : L panels (3 rows x 3 own nodes, column-major) for the whole tree; front workspaces for two adjacent
//              levels (a level reads its children's U from the other buffer) -- or, where whole levels do not fit, the top
//              of the tree front by front with a stack of update matrices (MfBatch); 64-bit offsets throughout.
// Integer work only.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
  long long F_off = 0;           // doubles inside the workspace it is factored in: (3 nrows)^2, column-major
  long long v_off = 0;           // doubles inside the solve's front vectors: 3 nrows
  // where the PARENT reads this front's update matrix: the front itself (c_ld = 3 nrows, c_k0 = own DOFs) or its compact
  // copy on the stack (c_ld = rows below the own DOFs, c_k0 = 0)
  long long cF_off = 0;
  int c_ld = 0, c_k0 = 0;
};

struct MfLevelStep {  // one panel step of a batch / level: fronts [0, n_active) of its list still have columns at j0
  int j0, n_active, max_below;  // max_below: largest number of DOF rows below the panel among them
};

// The factorisation walks BATCHES: fronts that are assembled and factored together.  Fronts deeper than `top_depth` are
// taken level by level (all fronts of one depth under one depth-`top_depth` ancestor: buffers 0 / 1 by depth parity, reused
// from subtree to subtree); fronts of depth <= top_depth one by one in postorder in the WORK buffer (2), their update
// matrices compacted onto a STACK (3) until the parent is assembled -- the classical multifrontal stack, so that the
// workspaces are a few of the largest fronts instead of whole levels of them (config C: 156 GB of whole-level workspaces
// next to a 118 GB factor).  top_depth = -1: every batch a whole level (everything that fits is done that way: fewest
// launches).
struct MfBatch {
  int depth, first, count;      // fronts batch_fronts[first .. first + count), most own columns first
  int step_off, step_count;     // bsteps[...]
  int hent_off, hent_count;     // entries of h_src / h_dst / ...
  int wbuf, cbuf;               // workspace the batch is factored in / the one its children's update matrices are read from
  long long F_base, F_doubles;  // the part of workspace wbuf this batch zeroes before assembling
  int push;                     // 1: a single front whose update matrix goes onto the stack afterwards
};
struct MfPlan {
  int N = 0;
  std::vector<int> order, inv;          // new -> old node, old -> new
  std::vector<MfFront> fronts;          // children before parents
  std::vector<int> rows, map;
  std::vector<int> level_off;           // [n_levels + 1] into level_fronts; level 0 = deepest
  std::vector<int> level_fronts;        // fronts of a level, most own columns first
  std::vector<int> step_off;            // [n_levels + 1] into steps (the SOLVE walks whole levels)
  std::vector<MfLevelStep> steps;
  int top_depth = -1;                   // fronts of depth <= top_depth: one by one, postorder (see MfBatch)
  std::vector<MfBatch> batches;         // the factorisation's order
  std::vector<int> batch_fronts;
  std::vector<MfLevelStep> bsteps;
  // H -> fronts: one entry per lower node block of the permuted matrix (batch order)
  std::vector<long long> h_src, h_dst;  // first double of the 3x3 block in H's values / in the front's workspace buffer
  std::vector<int> h_sld, h_dld;        // leading dimensions (H: row stride 3 deg, front: 3 nrows)
  long long L_total = 0, F_cap[4] = {0, 0, 0, 0}, v_total = 0, flops = 0;  // F_cap: level buffers 0 / 1, WORK, STACK
  int n_levels() const { return (int)level_off.size() - 1; }
  long long F_total() const { return F_cap[0] + F_cap[1] + F_cap[2] + F_cap[3]; }
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
    // the cut: the median plane of one coordinate axis, SNAPPED so that all nodes sharing the median coordinate (a plane of a
    // structured mesh) go to one side -- a cut through the plane makes a staircase separator twice as large -- as long as the
    // parts stay within 30 % / 70 %.  Axis: the longest extent; axes within 60 % of it are tried too and the one with the
    // smallest separator wins (irregular bodies: the shortest cut is not always across the longest side).
    double ext[3];
    for (int a = 0; a < 3; a++) {
      double lo = xyz[a][nodes[0]], hi = lo;
      for (int v : nodes) {
        lo = std::min(lo, xyz[a][v]);
        hi = std::max(hi, xyz[a][v]);
      }
      ext[a] = hi - lo;
    }
    const double emax = std::max(ext[0], std::max(ext[1], ext[2]));
    auto cut_axis = [&](int axis) -> size_t {  // partitions `nodes` (left part first), returns the size of the left part
      size_t half = nodes.size() / 2;
      const double* c = xyz[axis];
      std::nth_element(nodes.begin(), nodes.begin() + half, nodes.end(),
                       [&](int a, int b) { return c[a] < c[b] || (c[a] == c[b] && a < b); });
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
      return half;
    };
    auto separator_size = [&](size_t half) {  // the smaller of the two one-sided vertex separators of the current partition
      const int sl = stamp++, sr = stamp++;
      for (size_t t = 0; t < nodes.size(); t++) mark[nodes[t]] = t < half ? sl : sr;
      size_t nr = 0, nl_ = 0;
      for (size_t t = 0; t < nodes.size(); t++) {
        const int v = nodes[t], other = t < half ? sr : sl;
        bool touches = false;
        for (int k = off[v]; k < off[v + 1] && !touches; k++) touches = mark[cols[k]] == other;
        if (touches) (t < half ? nl_ : nr)++;
      }
      return std::min(nr, nl_);
    };
    static const bool try_axes = !(std::getenv("TLFEA_DIRECT_AXES") && std::atoi(std::getenv("TLFEA_DIRECT_AXES")) == 1);
    int axis = ext[0] >= ext[1] && ext[0] >= ext[2] ? 0 : (ext[1] >= ext[2] ? 1 : 2);
    if (try_axes && nodes.size() <= 200000) {  // (larger parts: the planning time of a 4 M-DOF mesh stays a few seconds)
      size_t best_sep = ~(size_t)0;
      int best_axis = axis;
      int cands = 0;
      for (int a = 0; a < 3; a++) cands += ext[a] >= 0.6 * emax;
      if (cands > 1) {
        for (int a = 0; a < 3; a++) {
          if (ext[a] < 0.6 * emax) continue;
          const size_t sz = separator_size(cut_axis(a));
          if (sz < best_sep || (sz == best_sep && a == axis)) {
            best_sep = sz;
            best_axis = a;
          }
        }
        axis = best_axis;
      }
    }
    size_t half = cut_axis(axis);
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
  P.step_off.assign((size_t)nl + 1, 0);
  auto by_columns = [&](int a, int c) {
    const int ka = P.fronts[a].c1 - P.fronts[a].c0, kc = P.fronts[c].c1 - P.fronts[c].c0;
    return ka > kc || (ka == kc && a < c);
  };
  auto push_steps = [&](const int* b, const int* e, std::vector<MfLevelStep>& out) {
    const int kmax = b == e ? 0 : 3 * (P.fronts[*b].c1 - P.fronts[*b].c0);
    for (int j0 = 0; j0 < kmax; j0 += kMfNB) {
      MfLevelStep st{j0, 0, 0};
      for (const int* it = b; it != e; ++it) {
        const MfFront& F = P.fronts[*it];
        const int k = 3 * (F.c1 - F.c0), m = 3 * F.nrows;
        if (k <= j0) break;
        st.n_active++;
        st.max_below = std::max(st.max_below, m - j0 - std::min(kMfNB, k - j0));
      }
      out.push_back(st);
    }
  };
  for (int l = 0; l < nl; l++) {  // levels: factor / vector storage and the solve's step lists
    int* b = P.level_fronts.data() + P.level_off[l];
    int* e = P.level_fronts.data() + P.level_off[l + 1];
    std::sort(b, e, by_columns);
    for (int* it = b; it != e; ++it) {
      MfFront& F = P.fronts[*it];
      const long long m = 3LL * F.nrows, k = 3LL * (F.c1 - F.c0);
      F.L_off = P.L_total;
      P.L_total += m * k;
      F.v_off = P.v_total;
      P.v_total += m;
      P.flops += k * k * k / 3 + (m - k) * k * k + (m - k) * (m - k) * k;
      if (P.L_total > max_doubles) return false;
    }
    push_steps(b, e, P.steps);
    P.step_off[l + 1] = (int)P.steps.size();
  }
  // ---- batches ----------------------------------------------------------------------------------------------------------
  std::vector<long long> msq((size_t)nf), usq((size_t)nf);
  for (int f = 0; f < nf; f++) {
    const MfFront& F = P.fronts[f];
    msq[f] = 9LL * F.nrows * F.nrows;
    usq[f] = 9LL * (F.nrows - (F.c1 - F.c0)) * (F.nrows - (F.c1 - F.c0));
  }
  // workspaces for a given top depth T: level buffers = the largest level of any subtree hanging below depth T (T = -1: of
  // the whole tree), WORK = the largest front of depth <= T, STACK = peak of the postorder walk over those fronts
  std::vector<int> anc((size_t)nf, -1);
  auto workspace = [&](int T, long long cap[4]) {
    cap[0] = cap[1] = cap[2] = cap[3] = 0;
    for (int f = nf - 1; f >= 0; f--) {
      const int d = P.fronts[f].depth;
      anc[f] = d <= T ? -1 : (d == T + 1 ? (T < 0 ? 0 : P.fronts[f].parent) : anc[P.fronts[f].parent]);
    }
    std::vector<int> slot((size_t)nf, -1);
    std::vector<std::vector<long long>> deep;
    for (int f = 0; f < nf; f++) {
      const int d = P.fronts[f].depth;
      if (d <= T) {
        cap[2] = std::max(cap[2], msq[f]);
        continue;
      }
      int& sl = slot[anc[f] < 0 ? 0 : anc[f]];
      if (sl < 0) {
        sl = (int)deep.size();
        deep.emplace_back((size_t)nl, 0);
      }
      deep[sl][d] += msq[f];
    }
    for (const auto& v : deep)
      for (int d = 0; d < nl; d++) cap[d & 1] = std::max(cap[d & 1], v[d]);
    // stack: postorder over the fronts of depth <= T (children before parents in index order IS a postorder)
    long long sp = 0;
    std::vector<long long> at((size_t)nf, 0);
    for (int f = 0; f < nf; f++) {
      const MfFront& F = P.fronts[f];
      if (F.depth > T) continue;
      long long base = sp;
      if (F.depth < T)
        for (int c : F.child)
          if (c >= 0) base = std::min(base, at[c]);
      at[f] = base;
      sp = base + (F.parent >= 0 ? usq[f] : 0);
      cap[3] = std::max(cap[3], std::max(sp, base));
      // while f is factored its children's matrices are still on the stack: the peak is before the pop
      long long before = base;
      if (F.depth < T)
        for (int c : F.child)
          if (c >= 0) before = std::max(before, at[c] + usq[c]);
      cap[3] = std::max(cap[3], before);
    }
    return cap[0] + cap[1] + cap[2] + cap[3];
  };
  {
    long long cap[4];
    int T = -1;
    if (std::getenv("TLFEA_MF_DEBUG"))
      for (int c = -1; c <= maxd; c++) {
        const long long w = workspace(c, cap);
        std::fprintf(stderr, "top depth %d: workspace %.4g (levels %.4g %.4g, work %.4g, stack %.4g)\n", c, (double)w, (double)cap[0],
                     (double)cap[1], (double)cap[2], (double)cap[3]);
      }
    static const int forced = std::getenv("TLFEA_DIRECT_TOP_DEPTH") ? std::atoi(std::getenv("TLFEA_DIRECT_TOP_DEPTH")) : -2;
    if (forced >= -1) T = std::min(forced, maxd);
    else
      while (T < maxd && P.L_total + workspace(T, cap) > max_doubles) T++;
    if (P.L_total + workspace(T, cap) > max_doubles) return false;
    P.top_depth = T;
    for (int t = 0; t < 4; t++) P.F_cap[t] = cap[t];
  }
  bool entries_ok = true;
  auto emit = [&](int depth, std::vector<int>& fl, int wbuf, int cbuf, long long base, int push) {
    if (fl.empty()) return;
    std::sort(fl.begin(), fl.end(), by_columns);
    MfBatch B{depth, (int)P.batch_fronts.size(), (int)fl.size(), (int)P.bsteps.size(), 0, (int)P.h_src.size(), 0, wbuf, cbuf, base, 0,
              push};
    long long fo = base;
    for (int f : fl) {
      MfFront& F = P.fronts[f];
      F.F_off = fo;
      fo += msq[f];
      if (!push) {  // the parent reads the front where it was factored
        F.cF_off = F.F_off;
        F.c_ld = 3 * F.nrows;
        F.c_k0 = 3 * (F.c1 - F.c0);
      }
      P.batch_fronts.push_back(f);
    }
    B.F_doubles = fo - base;
    push_steps(fl.data(), fl.data() + fl.size(), P.bsteps);
    B.step_count = (int)P.bsteps.size() - B.step_off;
    for (int f : fl) {
      const MfFront& F = P.fronts[f];
      const int* pr = P.rows.data() + F.row_off;
      const int ld = 3 * F.nrows;
      for (int p = F.c0; p < F.c1; p++) {
        const int c = order[p];
        for (int u = off[c]; u < off[c + 1]; u++) {
          const int q = P.inv[cols[u]];
          if (q < p) continue;
          const int* it = std::lower_bound(pr, pr + F.nrows, q);
          const int i = cols[u];
          const int* ci = std::lower_bound(cols + off[i], cols + off[i + 1], c);
          if (it == pr + F.nrows || *it != q || ci == cols + off[i + 1] || *ci != c) {
            entries_ok = false;  // pattern not symmetric
            continue;
          }
          // block (row node i, column node c) of H: row i's values start at 9 off[i], row stride 3 deg(i)
          P.h_src.push_back(9LL * off[i] + 3LL * (ci - (cols + off[i])));
          P.h_sld.push_back(3 * (off[i + 1] - off[i]));
          P.h_dst.push_back(F.F_off + 3LL * (int)(it - pr) + 3LL * (p - F.c0) * ld);
          P.h_dld.push_back(ld);
        }
      }
    }
    B.hent_count = (int)P.h_src.size() - B.hent_off;
    P.batches.push_back(B);
  };
  {
    const int T = P.top_depth;
    for (int f = nf - 1; f >= 0; f--) {
      const int d = P.fronts[f].depth;
      anc[f] = d <= T ? -1 : (d == T + 1 ? (T < 0 ? 0 : P.fronts[f].parent) : anc[P.fronts[f].parent]);
    }
    std::vector<std::vector<int>> by_depth((size_t)nl);
    auto emit_levels_under = [&](int a, int upto) {  // the level batches of everything deeper than T under ancestor a
      for (auto& v : by_depth) v.clear();
      for (int f = 0; f < upto; f++)
        if (P.fronts[f].depth > T && (T < 0 || anc[f] == a)) by_depth[P.fronts[f].depth].push_back(f);
      for (int d = maxd; d > T; d--) emit(d, by_depth[d], d & 1, (d + 1) & 1, 0, 0);
    };
    if (T < 0) emit_levels_under(0, nf);
    long long sp = 0;
    std::vector<int> one;
    for (int f = 0; f < nf; f++) {  // index order is a postorder of the top part
      MfFront& F = P.fronts[f];
      if (F.depth > T) continue;
      if (F.depth == T) emit_levels_under(f, f);
      long long base = sp;
      if (F.depth < T)
        for (int c : F.child)
          if (c >= 0) base = std::min(base, P.fronts[c].cF_off);
      one.assign(1, f);
      emit(F.depth, one, 2, F.depth < T ? 3 : (F.depth + 1) & 1, 0, F.parent >= 0 ? 1 : 0);
      F.cF_off = base;
      F.c_ld = 3 * (F.nrows - (F.c1 - F.c0));
      F.c_k0 = 0;
      sp = base + (F.parent >= 0 ? usq[f] : 0);
    }
  }
  if (!entries_ok) return false;
  long long lower = 0;
  for (int i = 0; i < N; i++)
    for (int u = off[i]; u < off[i + 1]; u++) lower += P.inv[cols[u]] >= P.inv[i];
  if ((long long)P.h_src.size() != lower) return false;  // an entry did not find its front: not a symmetric pattern
  return true;
}

}  // namespace tlfea
