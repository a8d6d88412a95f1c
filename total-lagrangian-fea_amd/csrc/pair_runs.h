// Lane -> node-pair assignment of tangent_blocks_kernel (one element per wavefront; the element matrix has S(S+1)/2
// symmetric 3x3 blocks (i <= j), "pairs").  Shared by the kernel and the CPU test (tests/test_rowgroups.py).
#pragma once

#if defined(__HIPCC__)
#define TLFEA_HD __host__ __device__ __forceinline__
#else
#define TLFEA_HD inline
#endif

namespace tlfea {

// index of pair (i, j), i <= j, in the row-major upper triangle (the layout of the element-block buffer)
TLFEA_HD int pair_index(int S, int i, int j) { return i * S - (i * (i - 1)) / 2 + (j - i); }

// pair p of the row-major upper triangle -> (i, j)
TLFEA_HD void pair_of(int S, int p, int& i, int& j) {
  int row = 0, rem = p;
  while (rem >= S - row) {
    rem -= S - row;
    row++;
  }
  i = row;
  j = row + rem;
}

// The pairs of lane `lane` of a 64-lane wavefront: (i, j0 + n), n < cnt -- all in ONE block row, so that the row node's
// data is read once per lane.  npl = pairs per lane = ceil(P / 64):
//   npl == 1 (T10: 55 pairs, ANCF-3243: 36): lane p owns pair p;
//   npl  > 1 (ANCF-3443: 136 pairs, npl = 3): row i is cut into runs of npl consecutive columns, one lane per run
//            (16 rows -> 51 lanes busy); the caller asserts that the runs fit into 64 lanes.
TLFEA_HD void lane_pair_run(int S, int npl, int lane, int& i, int& j0, int& cnt) {
  i = j0 = cnt = 0;
  if (npl == 1) {
    if (lane < S * (S + 1) / 2) {
      pair_of(S, lane, i, j0);
      cnt = 1;
    }
    return;
  }
  int base = 0;
  for (int r = 0; r < S; r++) {
    const int nl = (S - r + npl - 1) / npl;
    if (lane >= base && lane < base + nl) {
      i = r;
      j0 = r + npl * (lane - base);
      cnt = (S - j0 < npl) ? (S - j0) : npl;
    }
    base += nl;
  }
}

// lanes the runs of lane_pair_run occupy (must be <= 64)
constexpr TLFEA_HD int lane_pair_run_lanes(int S, int npl) {
  if (npl == 1) return S * (S + 1) / 2;
  int base = 0;
  for (int r = 0; r < S; r++) base += (S - r + npl - 1) / npl;
  return base;
}

}  // namespace tlfea
