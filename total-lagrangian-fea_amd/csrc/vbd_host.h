// Host-side set-up of SyncedVBDSolver (SyncedVBD.cu:764-1028 InitializeColoring): greedy vertex colouring of the
// coefficient graph (two coefficients are adjacent when an element holds both), colour -> node lists, and the
// first-fit grouping of colours that never meet in an element.  The colouring is the reference's
// (lib_utils/cpu_utils.cc:18-75): nodes ordered by std::sort on "degree descending" -- an unstable sort, so equal
// degrees come out in the C++ library's order; the same call is made here so that the colours, and with them the
// sequence of Gauss-Seidel updates, are the reference's.
#ifndef TLFEA_VBD_HOST_H_
#define TLFEA_VBD_HOST_H_
#include <algorithm>
#include <numeric>
#include <vector>

namespace tlfea {

struct VbdColoring {
  int n_colors = 0, n_groups = 0;
  std::vector<int> colors;         // [N]
  std::vector<int> color_offsets;  // [n_colors + 1]
  std::vector<int> color_nodes;    // [N] ascending node id inside a colour
  std::vector<int> group_offsets;  // [n_groups + 1]
  std::vector<int> group_colors;   // [n_colors]
  bool valid = false;              // ValidateColoring (cpu_utils.cc:77-96)
};

// conn: [S][E];  off / cols: node adjacency CSR with sorted columns, self included (the mass pattern)
inline void vbd_build_coloring(int S, int E, int N, const int* conn, const int* off, const int* cols, int group_size,
                               VbdColoring& o) {
  o = VbdColoring();
  std::vector<int> degrees((size_t)N), order((size_t)N);
  for (int i = 0; i < N; i++) {
    int d = off[i + 1] - off[i];
    if (std::binary_search(cols + off[i], cols + off[i + 1], i)) d--;  // the reference's sets hold neighbours only
    degrees[i] = d;
  }
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&degrees](int a, int b) { return degrees[a] > degrees[b]; });
  o.colors.assign((size_t)N, -1);
  std::vector<int> used((size_t)N + 1, -1);  // used[c] == v: colour c is taken around v (no table clear per node)
  for (int v : order) {
    for (int k = off[v]; k < off[v + 1]; k++) {
      const int c = o.colors[cols[k]];
      if (cols[k] != v && c >= 0) used[c] = v;
    }
    int c = 0;
    while (used[c] == v) ++c;
    o.colors[v] = c;
    o.n_colors = std::max(o.n_colors, c + 1);
  }
  // every element must see S different colours
  o.valid = true;
  {
    std::vector<int> stamp((size_t)o.n_colors, -1);
    for (int e = 0; e < E && o.valid; e++)
      for (int a = 0; a < S; a++) {
        const int c = o.colors[conn[(size_t)a * E + e]];
        if (stamp[c] == e) {
          o.valid = false;
          break;
        }
        stamp[c] = e;
      }
  }
  o.color_offsets.assign((size_t)o.n_colors + 1, 0);
  for (int i = 0; i < N; i++) o.color_offsets[o.colors[i] + 1]++;
  for (int c = 0; c < o.n_colors; c++) o.color_offsets[c + 1] += o.color_offsets[c];
  o.color_nodes.resize((size_t)N);
  {
    std::vector<int> cur(o.color_offsets.begin(), o.color_offsets.end() - 1);
    for (int i = 0; i < N; i++) o.color_nodes[cur[o.colors[i]]++] = i;
  }
  // colour groups (SyncedVBD.cu:866-990)
  std::vector<char> conflict((size_t)o.n_colors * o.n_colors, 0);
  for (int e = 0; e < E; e++)
    for (int a = 0; a < S; a++)
      for (int b = a + 1; b < S; b++) {
        const int ca = o.colors[conn[(size_t)a * E + e]], cb = o.colors[conn[(size_t)b * E + e]];
        if (ca != cb) conflict[(size_t)ca * o.n_colors + cb] = conflict[(size_t)cb * o.n_colors + ca] = 1;
      }
  std::vector<std::vector<int>> groups;
  const int gs = std::max(1, group_size);
  for (int c = 0; c < o.n_colors; c++) {
    bool placed = false;
    if (gs > 1)
      for (auto& g : groups) {
        if ((int)g.size() >= gs) continue;
        bool ok = true;
        for (int c2 : g)
          if (conflict[(size_t)c2 * o.n_colors + c]) {
            ok = false;
            break;
          }
        if (ok) {
          g.push_back(c);
          placed = true;
          break;
        }
      }
    if (!placed) groups.push_back({c});
  }
  o.n_groups = (int)groups.size();
  o.group_offsets.assign(1, 0);
  for (auto& g : groups) {
    for (int c : g) o.group_colors.push_back(c);
    o.group_offsets.push_back((int)o.group_colors.size());
  }
}

}  // namespace tlfea
#endif
