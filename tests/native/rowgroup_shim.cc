// Test shim: exposes the host set-up of the fused assembly kernel (csrc/rowgroup_host.h, integer work only) to ctypes,
// so that the CPU suite can check its invariants without a GPU.  Built by tests/test_rowgroups.py with g++.
#include "../../total-lagrangian-fea_amd/csrc/rowgroup_host.h"

static tlfea::RowGroupsHost g_rg;

extern "C" int rg_build(int N, int E, int S, const int* conn, const int* off, const int* cols, const int* n2e_off,
                        const int* n2e, const double* x, const double* y, const double* z, int* sizes) {
  if (!tlfea::build_row_groups(N, E, S, conn, off, cols, n2e_off, n2e, x, y, z, g_rg)) return 1;
  sizes[0] = g_rg.G();
  sizes[1] = (int)g_rg.gi_code.size();
  sizes[2] = g_rg.acc_max;
  sizes[3] = (int)(g_rg.pt.size() / 4);
  sizes[4] = g_rg.G();
  return 0;
}
extern "C" void rg_fetch_passes(int* pt, int* g_pass_off, int* gr_info, int* gi_mb) {
  std::copy(g_rg.pt.begin(), g_rg.pt.end(), pt);
  std::copy(g_rg.g_pass_off.begin(), g_rg.g_pass_off.end(), g_pass_off);
  std::copy(g_rg.gr_info.begin(), g_rg.gr_info.end(), gr_info);
  std::copy(g_rg.gi_mb.begin(), g_rg.gi_mb.end(), gi_mb);
}
extern "C" void rg_fetch(int* g_inst_off, int* g_row_off, int* gr_row, int* gr_acc, int* gi_code, int* gi_pack) {
  std::copy(g_rg.g_inst_off.begin(), g_rg.g_inst_off.end(), g_inst_off);
  std::copy(g_rg.g_row_off.begin(), g_rg.g_row_off.end(), g_row_off);
  std::copy(g_rg.gr_row.begin(), g_rg.gr_row.end(), gr_row);
  std::copy(g_rg.gr_acc.begin(), g_rg.gr_acc.end(), gr_acc);
  std::copy(g_rg.gi_code.begin(), g_rg.gi_code.end(), gi_code);
  std::copy(g_rg.gi_pack.begin(), g_rg.gi_pack.end(), gi_pack);
}

// ---- affine-element form (build_row_groups4) -----------------------------------------------------------------------------
static tlfea::RowGroups4Host g_rg4;
extern "C" int rg4_build(int N, int E, const int* conn, const int* off, const int* cols, const int* n2e_off, const int* n2e,
                         const double* x, const double* y, const double* z, int* sizes) {
  if (!tlfea::build_row_groups4(N, E, conn, off, cols, n2e_off, n2e, x, y, z, g_rg4)) return 1;
  sizes[0] = g_rg4.G();
  sizes[1] = (int)(g_rg4.gi_head.size() / 2);
  sizes[2] = g_rg4.acc_max;
  sizes[3] = (int)(g_rg4.pt.size() / 4);
  return 0;
}
extern "C" void rg4_fetch(int* g_inst_off, int* g_row_off, int* gr_row, int* gr_acc, int* gi_head, int* gi_ent, int* pt,
                          int* g_pass_off, int* gr_info) {
  std::copy(g_rg4.g_inst_off.begin(), g_rg4.g_inst_off.end(), g_inst_off);
  std::copy(g_rg4.g_row_off.begin(), g_rg4.g_row_off.end(), g_row_off);
  std::copy(g_rg4.gr_row.begin(), g_rg4.gr_row.end(), gr_row);
  std::copy(g_rg4.gr_acc.begin(), g_rg4.gr_acc.end(), gr_acc);
  std::copy(g_rg4.gi_head.begin(), g_rg4.gi_head.end(), gi_head);
  std::copy(g_rg4.gi_ent.begin(), g_rg4.gi_ent.end(), gi_ent);
  std::copy(g_rg4.pt.begin(), g_rg4.pt.end(), pt);
  std::copy(g_rg4.g_pass_off.begin(), g_rg4.g_pass_off.end(), g_pass_off);
  std::copy(g_rg4.gr_info.begin(), g_rg4.gr_info.end(), gr_info);
}

// ---- sparse direct solve: ordering + symbolic factorisation (csrc/direct_host.h) ----------------------------------------
#include "../../total-lagrangian-fea_amd/csrc/direct_host.h"
static tlfea::DirectHost g_dh;
extern "C" int dh_build(int N, const int* off, const int* cols, const double* x, const double* y, const double* z,
                        long long max_nnz, int* sizes) {
  if (!tlfea::direct_symbolic(N, off, cols, x, y, z, max_nnz, g_dh)) return 1;
  sizes[0] = g_dh.n;
  sizes[1] = (int)g_dh.indT.size();
  return 0;
}
extern "C" void dh_fetch(int* perm, int* ptrT, int* indT) {
  std::copy(g_dh.perm.begin(), g_dh.perm.end(), perm);
  std::copy(g_dh.ptrT.begin(), g_dh.ptrT.end(), ptrT);
  std::copy(g_dh.indT.begin(), g_dh.indT.end(), indT);
}

// ---- lane -> node-pair runs of tangent_blocks_kernel (csrc/pair_runs.h) ----------------------------------------------------
#include "../../total-lagrangian-fea_amd/csrc/pair_runs.h"
extern "C" int pair_runs(int S, int npl, int* i, int* j0, int* cnt) {  // 64 lanes each; returns the lanes the runs occupy
  for (int lane = 0; lane < 64; lane++) tlfea::lane_pair_run(S, npl, lane, i[lane], j0[lane], cnt[lane]);
  return tlfea::lane_pair_run_lanes(S, npl);
}
extern "C" int pair_index_of(int S, int i, int j) { return tlfea::pair_index(S, i, j); }
