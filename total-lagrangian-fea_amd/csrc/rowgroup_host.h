// rowgroup_host.h -- host set-up of the fused tangent + assembly kernel (assemble_direct_kernel, elem_kernels.hip):
// which node rows of H one wavefront owns, in which order, and where every (row, element, column node) block lands
// in the wavefront's LDS accumulator.  Integer work only, once per mesh.
//
//   * rows are ordered along a Morton (Z-order) curve of the reference coordinates: consecutive groups are spatial
//     neighbours, so the elements they re-read (every element is read by the owners of its S rows) are still in the
//     XCD's L2 -- the kernel deals contiguous group ranges to the XCDs;
//   * a group holds rows until their (row, element) instances exceed kInstBudget or their accumulators kAccBudget
//     doubles; a row with many instances (a corner node of a tet mesh: ~24 elements) forms a group of its own;
//   * per instance and column node j the packed word (accumulator offset of block column pos_j) | (3 deg << 16),
//     16 bits each -- a mesh that does not fit (deg > 10 922 or an accumulator beyond 64 Ki doubles) reports failure and
//     the solver keeps the two-kernel path.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace tlfea {

struct RowGroupsHost {
  std::vector<int> g_inst_off, g_row_off, gr_row, gr_acc, gi_code, gi_pack;
  // what the kernel reads: a flat table of PASSES (<= 6 instances of one group each) with the pass range of every
  // group, per-row records, and per instance the base of its row's mass values
  std::vector<int> pt;         // [P][4]: first instance | count + 8 first-of-group + 16 last-of-group + (rows << 8) |
                               //         first row | doubles of accumulator the group uses
  std::vector<int> g_pass_off; // [G+1] passes of group g
  std::vector<int> gr_info;    // [N][4]: acc offset + (position of the diagonal block << 16) | off[row] | deg | node
  std::vector<int> gi_mb;      // [S*E]: off[row] - acc offset / 3 (mass value of a block = mval[gi_mb + acc offset / 3])
  int acc_max = 0;
  int G() const { return (int)g_inst_off.size() - 1; }
};

inline uint64_t morton_spread21(uint64_t v) {  // 21 bits -> every third bit
  v &= 0x1fffffULL;
  v = (v | (v << 32)) & 0x1f00000000ffffULL;
  v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
  v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
  v = (v | (v << 2)) & 0x1249249249249249ULL;
  return v;
}

// rows along a Morton (Z-order) curve of the reference coordinates
inline void morton_order(int N, const double* x, const double* y, const double* z,
                         std::vector<std::pair<uint64_t, int>>& key) {
  double lo[3] = {x[0], y[0], z[0]}, hi[3] = {x[0], y[0], z[0]};
  for (int i = 1; i < N; i++) {
    lo[0] = std::min(lo[0], x[i]); hi[0] = std::max(hi[0], x[i]);
    lo[1] = std::min(lo[1], y[i]); hi[1] = std::max(hi[1], y[i]);
    lo[2] = std::min(lo[2], z[i]); hi[2] = std::max(hi[2], z[i]);
  }
  // one scale for the three axes: cells of the curve stay cubes on an elongated body
  const double ext = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300});
  const double sc = 2097151.0 / ext;
  key.resize((size_t)N);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    const uint64_t a = (uint64_t)((x[i] - lo[0]) * sc), b = (uint64_t)((y[i] - lo[1]) * sc),
                   c = (uint64_t)((z[i] - lo[2]) * sc);
    key[i] = {morton_spread21(a) | (morton_spread21(b) << 1) | (morton_spread21(c) << 2), i};
  }
  std::sort(key.begin(), key.end());
}

// conn: column-major E x S (coefficient ids), off/cols: node adjacency (sorted columns), n2e_off/n2e: incidence
// (e*S + local, ascending e per node), x/y/z: reference coordinates of the N rows.
inline bool build_row_groups(int N, int E, int S, const int* conn, const int* off, const int* cols, const int* n2e_off,
                             const int* n2e, const double* x, const double* y, const double* z, RowGroupsHost& out) {
  constexpr int kInstBudget = 24, kBigRow = 13, kMaxRows = 16, kPassInst = 6;
  static const int kAccBudget = std::getenv("TLFEA_AD_ACC") ? std::atoi(std::getenv("TLFEA_AD_ACC")) : 600;  // doubles
  if (N <= 0 || E <= 0) return false;
  std::vector<std::pair<uint64_t, int>> key;
  morton_order(N, x, y, z, key);

  out = RowGroupsHost();
  out.gr_row.reserve((size_t)N);
  out.gr_acc.reserve((size_t)N);
  out.g_row_off.push_back(0);
  out.g_inst_off.push_back(0);
  std::vector<int> open_rows;  // the open group of small rows
  int open_inst = 0, open_acc = 0, n_inst = 0;
  auto emit = [&](const int* rows, int n) {
    int a = 0;
    for (int r = 0; r < n; r++) {
      const int i = rows[r];
      out.gr_row.push_back(i);
      out.gr_acc.push_back(a);
      a += 9 * (off[i + 1] - off[i]);
      n_inst += n2e_off[i + 1] - n2e_off[i];
    }
    out.acc_max = std::max(out.acc_max, a);
    out.g_row_off.push_back((int)out.gr_row.size());
    out.g_inst_off.push_back(n_inst);
  };
  for (int t = 0; t < N; t++) {
    const int i = key[t].second;
    const int ni = n2e_off[i + 1] - n2e_off[i], ai = 9 * (off[i + 1] - off[i]);
    if (ai >= 65536 || 3 * (off[i + 1] - off[i]) >= 32768) return false;
    if (ni >= kBigRow || ai > kAccBudget / 2) {
      emit(&i, 1);
      continue;
    }
    if (!open_rows.empty() &&
        (open_inst + ni > kInstBudget || open_acc + ai > kAccBudget || (int)open_rows.size() >= kMaxRows)) {
      emit(open_rows.data(), (int)open_rows.size());
      open_rows.clear();
      open_inst = open_acc = 0;
    }
    open_rows.push_back(i);
    open_inst += ni;
    open_acc += ai;
  }
  if (!open_rows.empty()) emit(open_rows.data(), (int)open_rows.size());
  if (out.acc_max >= 65536) return false;

  const int G = out.G();
  out.gi_code.resize((size_t)std::max(1, n_inst));
  out.gi_mb.resize((size_t)std::max(1, n_inst));
  out.gi_pack.resize((size_t)std::max(1, n_inst) * S);
  out.gr_info.resize((size_t)N * 4);
#pragma omp parallel for schedule(dynamic, 256)
  for (int g = 0; g < G; g++) {
    size_t w = (size_t)out.g_inst_off[g];
    std::vector<char> seen;
    for (int r = out.g_row_off[g]; r < out.g_row_off[g + 1]; r++) {
      const int i = out.gr_row[r], a0 = out.gr_acc[r], deg = off[i + 1] - off[i];
      const int* c = cols + off[i];
      const int dpos = (int)(std::lower_bound(c, c + deg, i) - c);
      out.gr_info[4 * (size_t)r + 0] = a0 | (dpos << 16);
      out.gr_info[4 * (size_t)r + 1] = off[i];
      out.gr_info[4 * (size_t)r + 2] = deg;
      out.gr_info[4 * (size_t)r + 3] = i;
      seen.assign((size_t)deg, 0);
      for (int k = n2e_off[i]; k < n2e_off[i + 1]; k++, w++) {
        const int code = n2e[k], e = code / S;
        out.gi_code[w] = code;
        out.gi_mb[w] = off[i] - a0 / 3;
        for (int j = 0; j < S; j++) {
          const int pos = (int)(std::lower_bound(c, c + deg, conn[(size_t)j * E + e]) - c);
          // bit 31: this item is the first (lowest element) contribution to its block and carries the block's M/h
          const unsigned first = seen[pos] ? 0u : 0x80000000u;
          seen[pos] = 1;
          out.gi_pack[w * S + j] = (int)((unsigned)(a0 + 3 * pos) | ((unsigned)(3 * deg) << 16) | first);
        }
      }
    }
  }
  // pass table (serial: a few integers per pass)
  out.g_pass_off.push_back(0);
  for (int g = 0; g < G; g++) {
    const int i0 = out.g_inst_off[g], i1 = out.g_inst_off[g + 1], r0 = out.g_row_off[g], nr = out.g_row_off[g + 1] - r0;
    const int last_row = out.gr_row[r0 + nr - 1];
    const int acc_n = out.gr_acc[r0 + nr - 1] + 9 * (off[last_row + 1] - off[last_row]);
    int p0 = i0;
    do {  // a group without instances (nodes that belong to no element) still has one pass: its rows are written
      const int cnt = std::min(kPassInst, i1 - p0);
      const int flags = (p0 == i0 ? 8 : 0) | (p0 + kPassInst >= i1 ? 16 : 0);
      out.pt.push_back(p0);
      out.pt.push_back(cnt | flags | (nr << 8));
      out.pt.push_back(r0);
      out.pt.push_back(acc_n);
      p0 += kPassInst;
    } while (p0 < i1);
    out.g_pass_off.push_back((int)(out.pt.size() / 4));
  }
  return true;
}

// ---- the affine-element form of the fused kernel (assemble_affine_kernel): 16 instances per pass, 4 lanes per instance
// (lane n = vertex n of the element), every lane adds four blocks: column "vertex n" and the three mid-edge columns
// (n, p).  Per instance a header {element * 10 + local row node, 3 deg} and per (instance, n) four 16-bit words
// (accumulator offset of the block) / 3 for p = 0..3 (p == n: the vertex column).  M/h and the pinned rows' penalty
// are added when a group's rows are written, so there are no per-block flags.
struct RowGroups4Host {
  std::vector<int> g_inst_off, g_row_off, gr_row, gr_acc;
  std::vector<int> pt;          // [P][4]: first instance | count (bits 0-4) + 32 first + 64 last of group + (rows << 8) |
                                //         first row | accumulator doubles of the group
  std::vector<int> g_pass_off;  // [G+1]
  std::vector<int> gr_info;     // [N][4] as RowGroupsHost
  std::vector<int> gi_head;     // [10 E][2]
  std::vector<int> gi_ent;      // [10 E][4][2]: two 16-bit words per int
  int acc_max = 0;
  int G() const { return (int)g_inst_off.size() - 1; }
};

inline int t10_mid_of(int a, int b) {  // local index of the mid-edge node of vertices a != b (FEAT10Data.cu:143)
  static const int tab[4][4] = {{-1, 4, 6, 7}, {4, -1, 5, 8}, {6, 5, -1, 9}, {7, 8, 9, -1}};
  return tab[a][b];
}

inline bool build_row_groups4(int N, int E, const int* conn, const int* off, const int* cols, const int* n2e_off,
                              const int* n2e, const double* x, const double* y, const double* z, RowGroups4Host& out) {
  constexpr int S = 10, kMaxRows = 16, kPassInst = 16;
  static const int kInstBudget = std::getenv("TLFEA_AF_INST") ? std::atoi(std::getenv("TLFEA_AF_INST")) : 32;
  static const int kAccBudget = std::getenv("TLFEA_AF_ACC") ? std::atoi(std::getenv("TLFEA_AF_ACC")) : 1024;  // doubles
  if (N <= 0 || E <= 0) return false;
  std::vector<std::pair<uint64_t, int>> key;
  morton_order(N, x, y, z, key);
  out = RowGroups4Host();
  out.gr_row.reserve((size_t)N);
  out.gr_acc.reserve((size_t)N);
  out.g_row_off.push_back(0);
  out.g_inst_off.push_back(0);
  int open_rows = 0, open_inst = 0, open_acc = 0, n_inst = 0;
  auto close = [&]() {
    out.acc_max = std::max(out.acc_max, open_acc);
    out.g_row_off.push_back((int)out.gr_row.size());
    out.g_inst_off.push_back(n_inst);
    open_rows = open_inst = open_acc = 0;
  };
  // Rows join the open group in curve order; when the next row does not fit, the following kLook rows are searched for
  // one that does -- preferably one that brings the group to exactly 16 or 32 instances, i.e. full passes (tet meshes:
  // mid-edge rows of 4-6 instances, vertex rows of ~24) -- and a group closes as soon as its passes are full.
  constexpr int kLook = 16;
  std::vector<char> used((size_t)N, 0);
  auto add = [&](int i) {
    const int ni = n2e_off[i + 1] - n2e_off[i], ai = 9 * (off[i + 1] - off[i]);
    out.gr_row.push_back(i);
    out.gr_acc.push_back(open_acc);
    open_rows++;
    open_inst += ni;
    open_acc += ai;
    n_inst += ni;
  };
  for (int t = 0; t < N; t++) {
    if (3 * (off[key[t].second + 1] - off[key[t].second]) >= 32768) return false;
  }
  int head = 0;  // first row of the curve not yet in a group
  while (head < N) {
    if (used[head]) {
      head++;
      continue;
    }
    if (!open_rows) {  // a group starts with the next row of the curve, whatever its size
      used[head] = 1;
      add(key[head].second);
      if (open_inst > 0 && open_inst % kPassInst == 0) close();
      continue;
    }
    int best = -1, best_ni = -1;
    bool exact = false;
    for (int u = head, seen = 0; u < N && seen < kLook; u++) {
      if (used[u]) continue;
      seen++;
      const int i = key[u].second, ni = n2e_off[i + 1] - n2e_off[i], ai = 9 * (off[i + 1] - off[i]);
      if (open_inst + ni > kInstBudget || open_acc + ai > kAccBudget || open_rows >= kMaxRows) continue;
      const bool ex = (open_inst + ni) % kPassInst == 0;
      if ((ex && !exact) || (ex == exact && ni > best_ni)) {
        best = u;
        best_ni = ni;
        exact = ex;
      }
    }
    if (best < 0) {
      close();
      continue;
    }
    used[best] = 1;
    add(key[best].second);
    if (open_inst % kPassInst == 0 && open_inst > 0) close();
  }
  if (open_rows) close();
  if (out.acc_max >= 65536) return false;

  const int G = out.G();
  out.gi_head.resize((size_t)std::max(1, n_inst) * 2);
  out.gi_ent.resize((size_t)std::max(1, n_inst) * 8);
  out.gr_info.resize((size_t)N * 4);
#pragma omp parallel for schedule(dynamic, 256)
  for (int g = 0; g < G; g++) {
    size_t w = (size_t)out.g_inst_off[g];
    for (int r = out.g_row_off[g]; r < out.g_row_off[g + 1]; r++) {
      const int i = out.gr_row[r], a0 = out.gr_acc[r], deg = off[i + 1] - off[i];
      const int* c = cols + off[i];
      const int dpos = (int)(std::lower_bound(c, c + deg, i) - c);
      out.gr_info[4 * (size_t)r + 0] = a0 | (dpos << 16);
      out.gr_info[4 * (size_t)r + 1] = off[i];
      out.gr_info[4 * (size_t)r + 2] = deg;
      out.gr_info[4 * (size_t)r + 3] = i;
      for (int k = n2e_off[i]; k < n2e_off[i + 1]; k++, w++) {
        const int code = n2e[k], e = code / S;
        out.gi_head[2 * w + 0] = code;
        out.gi_head[2 * w + 1] = 3 * deg;
        for (int n = 0; n < 4; n++) {
          unsigned word[4];
          for (int p = 0; p < 4; p++) {
            const int j = p == n ? n : t10_mid_of(n, p);
            const int pos = (int)(std::lower_bound(c, c + deg, conn[(size_t)j * E + e]) - c);
            word[p] = (unsigned)(a0 / 3 + pos);
          }
          out.gi_ent[8 * w + 2 * n + 0] = (int)(word[0] | (word[1] << 16));
          out.gi_ent[8 * w + 2 * n + 1] = (int)(word[2] | (word[3] << 16));
        }
      }
    }
  }
  out.g_pass_off.push_back(0);
  for (int g = 0; g < G; g++) {
    const int i0 = out.g_inst_off[g], i1 = out.g_inst_off[g + 1], r0 = out.g_row_off[g], nr = out.g_row_off[g + 1] - r0;
    const int last_row = out.gr_row[r0 + nr - 1];
    const int acc_n = out.gr_acc[r0 + nr - 1] + 9 * (off[last_row + 1] - off[last_row]);
    int p0 = i0;
    do {
      const int cnt = std::min(kPassInst, i1 - p0);
      const int flags = (p0 == i0 ? 32 : 0) | (p0 + kPassInst >= i1 ? 64 : 0);
      out.pt.push_back(p0);
      out.pt.push_back(cnt | flags | (nr << 8));
      out.pt.push_back(r0);
      out.pt.push_back(acc_n);
      p0 += kPassInst;
    } while (p0 < i1);
    out.g_pass_off.push_back((int)(out.pt.size() / 4));
  }
  return true;
}

}  // namespace tlfea
