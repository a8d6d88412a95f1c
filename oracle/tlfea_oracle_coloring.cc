// CPU oracle, part 3 (TEST INFRASTRUCTURE ONLY -- see tlfea_oracle.h): vertex colouring of SyncedVBDSolver.
// Restates lib_utils/cpu_utils.cc:18-123 (BuildVertexAdjacency, GreedyVertexColoring, ValidateColoring) and the colour
// grouping of SyncedVBD.cu:866-990.  C++ because the reference orders the nodes with std::sort on "degree descending":
// that sort is not stable, so the order of equal-degree nodes -- and with it the colours -- is whatever the C++
// library's introsort produces; calling the same library routine with the same comparator reproduces it.
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <set>
#include <vector>

#include "tlfea_oracle.h"

// conn: [S][E] coefficient ids
extern "C" int orc_vbd_coloring(int S, int E, int N, const int* conn, int* colors) {
  std::vector<std::set<int>> adj((size_t)N);
  for (int e = 0; e < E; e++)
    for (int i = 0; i < S; i++)
      for (int j = i + 1; j < S; j++) {
        const int a = conn[(size_t)i * E + e], b = conn[(size_t)j * E + e];
        adj[a].insert(b);
        adj[b].insert(a);
      }
  std::vector<int> degrees((size_t)N), order((size_t)N);
  for (int i = 0; i < N; i++) degrees[i] = (int)adj[i].size();
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&degrees](int a, int b) { return degrees[a] > degrees[b]; });
  std::fill(colors, colors + N, -1);
  std::vector<char> used((size_t)N + 1);
  int n_colors = 0;
  for (int v : order) {
    std::fill(used.begin(), used.end(), 0);
    for (int nb : adj[v])
      if (colors[nb] >= 0) used[colors[nb]] = 1;
    int c = 0;
    while (used[c]) ++c;
    colors[v] = c;
    n_colors = std::max(n_colors, c + 1);
  }
  return n_colors;
}

extern "C" int orc_vbd_validate_coloring(int S, int E, const int* conn, const int* colors) {
  for (int e = 0; e < E; e++) {
    std::set<int> seen;
    for (int i = 0; i < S; i++)
      if (!seen.insert(colors[conn[(size_t)i * E + e]]).second) return 0;
  }
  return 1;
}

// first-fit grouping of colours that never meet in an element, at most group_size per group; -> number of groups
extern "C" int orc_vbd_color_groups(int S, int E, const int* conn, const int* colors, int n_colors, int group_size,
                                    int* group_offsets, int* group_colors) {
  std::vector<std::vector<char>> conflict((size_t)n_colors, std::vector<char>((size_t)n_colors, 0));
  for (int e = 0; e < E; e++)
    for (int i = 0; i < S; i++)
      for (int j = i + 1; j < S; j++) {
        const int a = colors[conn[(size_t)i * E + e]], b = colors[conn[(size_t)j * E + e]];
        if (a != b) conflict[a][b] = conflict[b][a] = 1;
      }
  std::vector<std::vector<int>> groups;
  for (int c = 0; c < n_colors; c++) {
    bool placed = false;
    if (group_size > 1)
      for (auto& g : groups) {
        if ((int)g.size() >= group_size) continue;
        bool ok = true;
        for (int c2 : g)
          if (conflict[c2][c]) {
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
  int cur = 0;
  group_offsets[0] = 0;
  for (size_t g = 0; g < groups.size(); g++) {
    for (int c : groups[g]) group_colors[cur++] = c;
    group_offsets[g + 1] = cur;
  }
  return (int)groups.size();
}
