// tlfea_internal.h -- device views and launch wrappers shared by the .hip translation units.
// gfx950 (MI355X) only.  Nothing here is part of the C-ABI (see include/tlfea_c.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "mf_host.h"

namespace tlfea {

constexpr int kNQ = 5;      // T10: Keast 5-point rule (quadrature_utils.h:134)
constexpr int kNN = 10;     // T10: nodes per element
constexpr int kMaxQ = 48;   // ANCF-3443 force rule 4x4x3 (quadrature_utils.h:22)
constexpr int kMaxS = 16;   // ANCF-3443 shape functions
constexpr int kWave = 64;   // CDNA wavefront

// element kinds: <S shape functions, Q force quadrature points>
enum ElemKind : int { kT10 = 0, kANCF3243 = 1, kANCF3443 = 2 };

enum MaterialModel : int { kSVK = 0, kMooneyRivlin = 1 }; // MaterialModel.cuh:14-17

// Material scalars travel as kernel arguments (SGPRs) -- the reference dereferences one device
// pointer per scalar per thread (FEAT10Data.cuh:825-829).
struct Material {
  int model;
  double lambda, mu;          // SVK
  double mu10, mu01, kappa;   // Mooney-Rivlin
  double eta, lamd;           // Kelvin-Voigt damping
  double rho0;
};

// Device view of one mesh + state (all pointers are device pointers).  "N" counts coefficient vectors: nodes
// for T10, 4 per node (r, r_u, r_v, r_w) for the ANCF types.
struct ElemView {
  int E, N, Epad, S, Q;
  const int* conn;        // [S][E]   coefficient ids, column-major E x S (T10: the reference layout)
  const double* x;        // [N] current coordinates, SoA like the reference (d_h_x12/y12/z12)
  const double* y;
  const double* z;
  const double* gradN;    // [E][Q][3][S]  reference layout: wave-per-element kernels read contiguous runs
  const double* gradN_t;  // [Q][3][S][Epad] element-fastest copy: thread-per-element kernels coalesce
  const double* detJ;     // [E][Q]
  double qw[kMaxQ];       // combined quadrature weights
};

// T10 only: the inertia term M (v - v_prev) / h of grad L (SyncedNewton.cu:362-371) evaluated element by element in the
// residual launch (M = sum_e M_e with the rule of FEAT10Data.cu:206-278) instead of a CSR product with gathered
// velocities: Nq = shape functions at the 5 Keast points, rho = the density the mass matrix was assembled with.
struct MassTerm {
  const double* vprev;   // null: no mass term (mbuf is not written)
  double* mbuf;          // [10][Epad][6] per (node, element): force row | inertia row (replaces fbuf on this path)
  double rho_inv_h;      // rho0 / h
  double Nq[kNQ][kNN];
};

// node -> element incidence and the scatter map of the row-owner ("gather") assembly
struct Incidence {
  const int* n2e_off;   // [N+1]
  const int* n2e;       // [S*E]   e*S + local index, ascending e per node
  const int* n2e_pos;   // [S*E][S] position of column node conn(e,j) inside row i
  const int* off;       // [N+1]   node-level adjacency CSR (== mass CSR pattern)
  const int* cols;      // [nnz_coef] sorted per row
  const int* diagpos;   // [N]     position of i in row i
};

// Work lists of the fused tangent + assembly kernel (T10, SVK): a GROUP is a few node rows of H (3 CSR rows each) whose
// (row, incident element) instances are worked through in PASSES of up to 6; the resident wavefronts of an XCD walk
// that XCD's range of groups side by side (wave w takes groups w, w + W, w + 2W, ...).  Built once per mesh on the host
// (rowgroup_host.h): rows in Morton order of the reference coordinates, so that consecutive groups share elements.
struct RowGroups {
  int G;                  // groups
  int n_inst;             // S*E instances
  int acc_max;            // doubles of LDS accumulator the largest group needs
  const int* g_pass_off;  // [G+1] passes of group g
  const int4* pt;         // [P] pass: first instance | count + 8 first + 16 last of group + (rows << 8) | first row |
                          //          accumulator doubles of the group
  const int4* gr_info;    // [N] row: acc offset + (diagonal block position << 16) | off[row] | deg | node
  const int* gi_code;     // [S*E] element * S + local node, group order (ascending element inside a row)
  const int* gi_mb;       // [S*E] off[row] - acc offset / 3: mass value of an item's block = mval[gi_mb + acc / 3]
  const int* gi_pack;     // [S*E][S] per (instance, column node j): (acc offset of the block) | (3 deg << 16) |
                          //          bit 31: first contribution to the block (carries its M/h)
};

// Work lists of the affine-element form (assemble_affine_kernel; rowgroup_host.h build_row_groups4): passes of up to 16
// instances, 4 lanes per instance.
struct RowGroups4 {
  int G, n_inst, acc_max;
  const int* g_pass_off;  // [G+1]
  const int4* pt;         // [P] first instance | count (bits 0-4) + 32 first + 64 last + (rows << 8) | first row | acc doubles
  const int4* gr_info;    // [N] as RowGroups
  const int2* gi_head;    // [10 E] element * 10 + local row node | 3 deg
  const int2* gi_ent;     // [10 E][4] per (instance, vertex n): (acc offset / 3) of the blocks (n, p), p = 0..3, 16 bits each
};
// What the affine form knows about the mesh and the rule: straight-sided T10 elements have constant vertex gradients
// g_m = grad L_m and constant det J; the 5-point Keast rule has L = 1/4 at point q0 and L_m = 1/2 (others 1/6) at
// point qv[m].  gvec: [E][16] = g_0 | detJ | g_1 | - | g_2 | - | g_3 | -.
struct AffineView {
  const double* gvec;
  int q0, qv[4];
};

// ---- launch wrappers (defined in the .hip files) -------------------------------------------
void launch_dndu_pre(hipStream_t s, int E, int Epad, const int* conn, const double* x, const double* y,
                     const double* z, const double* qx, const double* qy, const double* qz,
                     double* gradN, double* gradN_t, double* detJ);
void launch_residual(hipStream_t s, const ElemView& m, const Material& mat, const double* v /*or null*/,
                     double* fbuf /*[E][30]*/, double* F, double* P, double* Fdot, double* Pvis,
                     double* Fq = nullptr /*[E][Q][9] row-major F per point, for the fused assembly*/,
                     const MassTerm* mt = nullptr /*T10: also write the per-element inertia rows*/,
                     double fq_h = 0.0 /*> 0: affine form, Fq holds [E][5][10] = F per point (centroid point first)*/,
                     int fq_slots = 0x43210 /*record slot of point q in bits 4q..4q+3 (the affine assembly's order)*/);
// grad L without the mass CSR product (T10, inertia rows from the residual launch): 8 lanes per node
void launch_grad_light(hipStream_t s, int N, int Epad, const Incidence& inc, const double* fbuf, const double* mbuf,
                       const double* f_ext, const double* x, const double* y, const double* z, const double* xt,
                       const double* yt, const double* zt, const int* fixed_slot, const double* lam, const double* nw,
                       double h, double rho, double* f_int, double* cons, double* g);
// fused tangent + row assembly (T10, SVK): H rows straight from grad N and the F of the last residual launch
void launch_assemble_direct(hipStream_t s, const ElemView& m, const Material& mat, double h, const RowGroups& rg,
                            const double* Fq, const double* mval, const int* fixed_slot, const double* nw,
                            double penalty, double* Hval);
// affine-element set-up: gvec from grad N / det J, and the largest relative deviation of the stored grad N / det J from
// the affine form (dev_max: one double on the device, zeroed by the caller)
void launch_affine_pre(hipStream_t s, const ElemView& m, const AffineView& av, double* gvec, double* dev_max);
// fused tangent + row assembly, affine elements: Fq16 = [E][5][10], F per point, centroid point first (last residual launch)
void launch_assemble_affine(hipStream_t s, const ElemView& m, const Material& mat, double h, const RowGroups4& rg,
                            const AffineView& av, const double* Fq16,
                            const double* cmass /*[10][16] sum_q w_q N_i N_j per (row node, vertex n, p); mid-edge columns halved*/,
                            double rho0 /*density of the assembled mass matrix, 0 = none*/, const int* fixed_slot,
                            const double* nw, double penalty, double* Hval);
void launch_tangent_blocks(hipStream_t s, const ElemView& m, const Material& mat, double h,
                           double* Kbuf /*[E][55][9]*/);
void launch_assemble_rows(hipStream_t s, int N, int S, int maxdeg, const Incidence& inc, const double* Kbuf,
                          const double* mval, double inv_h, const int* fixed_slot, const double* nw,
                          double penalty, double* Hval);
void launch_vbd_color(hipStream_t s, int lanes /*16|32|64 per node*/, const ElemView& m, const Material& mat,
                      const Incidence& inc, const int* nodes, int count, const double* mval, const double* f_ext, const int* fixed_slot, const double* xt,
                      const double* yt, const double* zt, const double* lam, double h, double rho, double omega,
                      double hess_eps, const double* v_prev, const double* xp, const double* yp, const double* zp,
                      double* v, double* x, double* y, double* z, const int* conn_rm /*[E][S]*/, double* xyz /*[N][3]*/);
void launch_interleave_xyz(hipStream_t s, int N, const double* x, const double* y, const double* z, double* xyz);
void launch_mass_values(hipStream_t s, const ElemView& m, const Incidence& inc, const double* qx,
                        const double* qy, const double* qz, double rho0, double* mval);
void launch_grad(hipStream_t s, int N, const Incidence& inc, const double* fbuf, const double* mval,
                 const double* v, const double* vprev, const double* f_ext, const double* x,
                 const double* y, const double* z, const double* xt, const double* yt, const double* zt,
                 const int* fixed_slot, const double* lam, const double* nw, double h, double rho,
                 double* f_int, double* cons, double* g);
void launch_fint_gather(hipStream_t s, int N, const Incidence& inc, const double* fbuf, double* f_int);
// general linear constraints c = J x - rhs (CSR over constraint rows, columns = 3*coef + component)
void launch_lin_constraint(hipStream_t s, int nc, const int* joff, const int* jcol, const double* jval,
                           const double* rhs, const double* x, const double* y, const double* z, double* c);
void launch_lin_constraint_grad(hipStream_t s, int ndof, const int* jtoff, const int* jtcol, const double* jtval,
                                const double* lam, const double* c, double h, double rho, double* g);
void launch_lin_constraint_hessian(hipStream_t s, int ndof, const int* jtoff, const int* jtcol, const double* jtval,
                                   const int* joff, const int* jcol, const double* jval, const int* off,
                                   const int* cols, double f, double* H);
void launch_constraint(hipStream_t s, int n_fixed, const int* fixed_nodes, const double* x, const double* y,
                       const double* z, const double* xt, const double* yt, const double* zt, double* cons);

// vector / PCG kernels
constexpr int kNPart = 512; // partial sums per reduction (fixed -> run-to-run bitwise reproducible)
void launch_norm2(hipStream_t s, const double* a, const double* w /*weights or null*/, int n, double* part,
                  double* out /*device scalar: sum of squares*/);
void launch_extract_dinv(hipStream_t s, int N, const Incidence& inc, const double* Hval, double* Dinv);
void launch_pcg_init(hipStream_t s, int N, const double* b, const double* Dinv, const double* w, double* x,
                     double* r, double* z, double* rz_part, double* bb_part);
void launch_spmv_dir_dot(hipStream_t s, int N, const Incidence& inc, const double* Hval, const double* z,
                         const double* p_old, int first, const double* rz_part_old, const double* rz_part_new,
                         double* p_new, double* q, double* pq_part, bool fused, bool nt, const double* wown = nullptr);
void launch_pcg_direction(hipStream_t s, int n, const double* z, int first, const double* rz_part_old,
                          const double* rz_part_new, double* p);
void launch_pcg_update(hipStream_t s, int N, const double* Dinv, const double* w, const double* p, const double* q,
                       const double* rz_part_old, const double* pq_part, double* x, double* r, double* z,
                       double* rz_part_new, double* rr_part);
void launch_sum_parts(hipStream_t s, const double* part, double* out);
void launch_cheb_init(hipStream_t s, int N, const double* Dinv, const double* r, const double* sc,
                      const double* coef, double* d, double* z, double* res);
void launch_cheb_step(hipStream_t s, int N, const Incidence& inc, const double* Hval, const double* Dinv,
                      const double* d_old, const double* coef, double* d_new, double* z, double* res,
                      const double* r, const double* w, double* rz_part, bool last);
void launch_cheb_update(hipStream_t s, int N, const double* Dinv, const double* q, const double* d_old,
                        const double* coef, double* d_new, double* z, double* res, const double* r, const double* w,
                        const double* sc, double* rz_part, bool last);
// low-precision (fp16/fp32) scaled block-CSR copy of H for the polynomial preconditioner
void launch_lp_scale(hipStream_t s, int N, const double* D, const double* Dinv, double* sc, double* Dinv_s);
void launch_lp_convert(hipStream_t s, int N, const Incidence& inc, const double* Hval, const double* sc, const int* own,
                       const double* Dglob, void* B8, void* B1, int bits);
void launch_mask_scale(hipStream_t s, int N, const double* sc, const int* own, double* out);
// the polynomial in single precision (single-GPU path): fp32 vectors, fp16/fp32 matrix, fp32 accumulation
void launch_spmv_dir_dot_f32(hipStream_t s, int N, const Incidence& inc, const float* Hval32, const double* z,
                             const double* p_old, int first, const double* rz_part_old, const double* rz_part_new,
                             double* p_new, double* q, double* pq_part, bool fused, const double* wown = nullptr);
void launch_residual_replace_init32(hipStream_t s, int N, const double* b, const double* q, double* r, double* rr_part,
                                    const float* Dinv_f, const double* sc, const double* coef, float* d, float* z,
                                    float* res, const double* wown = nullptr);
// overlapping partition: ghost refresh messages (nvec fields of dim values per node, interleaved per node)
void launch_halo_pack_f64(hipStream_t s, int n, const int* idx, int dim, int nvec, const double* a, const double* b,
                          const double* c, double* msg);
void launch_halo_unpack_f64(hipStream_t s, int n, const int* idx, int dim, int nvec, const double* msg, double* a, double* b,
                            double* c);
void launch_halo_pack_f32(hipStream_t s, int n, const int* idx, int dim, int nvec, const float* a, const float* b,
                          const float* c, float* msg);
void launch_halo_unpack_f32(hipStream_t s, int n, const int* idx, int dim, int nvec, const float* msg, float* a, float* b,
                            float* c);
void launch_to_float(hipStream_t s, size_t n, const double* a, float* b);
void launch_cheb32_init(hipStream_t s, int N, const float* Dinv_f, const double* r, const double* sc,
                        const double* coef, float* d, float* z, float* res, int row0 = 0);
// Partition-boundary rows of the single-precision polynomial on the multi-GPU path: their Hs d is the SUM over ranks of
// the ranks' partial rows (launch_spmv32_rows -> all-reduce), handed to the fused step through bsum; w = 1/multiplicity
// of a DOF in the r.z partials of the last step.
struct C32Bnd {
  const int* bslot = nullptr;    // [N] slot of a boundary node in the exchange buffer, -1 for interior nodes
  const double* bsum = nullptr;  // [3 slots] summed Hs d of the boundary rows
  const double* w = nullptr;     // [3N] weights of the r.z partials (last step), null = 1
  const double* zw = nullptr;    // device scalar: z^ += zw d' instead of z^ += d' (fourth-kind Chebyshev smoothers), null = 1
  int xcd = 0;                   // set by the launcher: XCD-aware row sweep (solver_kernels.hip, row_sweep)
};
void launch_cheb32(hipStream_t s, int N, int nnz_coef, const Incidence& inc, const void* B8, const void* B1, int bits,
                   const float* Dinv_f, const double* sc, const float* d_old, const double* coef, float* d_new,
                   const float* z, float* z_new, const float* res, float* res_new, const double* r, double* z_out,
                   double* rz_part, bool last, C32Bnd bnd = C32Bnd());
// out[3 slot[k] + c] = (Hs d)_{rows[k], c} for a list of rows (this rank's part of the partition-boundary rows)
void launch_spmv32_rows(hipStream_t s, int n_rows, const int* rows, const int* slots, const Incidence& inc, const void* B8,
                        const void* B1, int bits, const float* d, double* out);
// restriction of the partition-boundary coarse rows only: out[3 slot[k] + c] = sum_children w res_f / sc_f (this rank's part)
void launch_pmg_restrict_rows(hipStream_t s, int n_rows, const int* rows, const int* slots, const int* child_off,
                              const int* child, const float* child_w, const float* res_f, const double* sc_f, double* out);
// two-level p-multigrid (T10): Galerkin coarse operator and grid transfers (pmg_host.h holds the integer set-up)
void launch_pmg_galerkin(hipStream_t s, int nnz_c, const int* c_off, const int* cblk_row, const int* con_off,
                         const int* con_base, const int* con_deg, const float* con_w, const double* Hf, double* Hc);
void launch_pmg_restrict_init(hipStream_t s, int Nc, const int* child_off, const int* child, const float* child_w,
                              const float* res_f, const double* sc_f, const double* sc_c, const float* Dinv_c,
                              const double* coef_c, float* d_c, float* z_c, float* res_c, const int* bslot = nullptr,
                              const double* bsum = nullptr);
// third level: rigid-body-mode aggregation of the vertex level (pmg_host.h agg_build)
void launch_agg_galerkin(hipStream_t s, int n_pairs, const int* pair_A, const int* pair_pos, const int* pair_B,
                         const int* pcon_off, const int* pcon_base, const int* pcon_deg, const int* pcon_i,
                         const int* pcon_j, const double* rvec, const int* active, const int* off3, const double* Hc,
                         double* H3);
void launch_agg_restrict_init(hipStream_t s, int N3, const int* mem_off, const int* mem, const double* rvec,
                              const float* res2, const double* sc2, const double* sc3, const float* Dinv3,
                              const double* coef3, float* d3, float* z3, float* res3);
void launch_agg_restrict(hipStream_t s, int N3, const int* mem_off, const int* mem, const double* rvec, const float* res2,
                         const double* sc2, double* r3);
void launch_agg_init3(hipStream_t s, int N3, const double* r3, const double* sc3, const float* Dinv3, const double* coef3,
                      float* d3, float* z3, float* res3);
void launch_agg_prolong(hipStream_t s, int Nc, const int* agg, const double* rvec, const float* z3, const double* sc3,
                        const double* sc2, float* z2, float* d2);
void launch_pmg_prolong(hipStream_t s, int N, const int* par0, const int* par1, const float* z_c, const double* sc_c,
                        const double* sc_f, float* z_f, float* d_f);
// mode 0: Chebyshev step, 1: last step (z back in the unscaled space + r.z slots in out), 2: out = Hs d_old
void launch_cheb_lp(hipStream_t s, int N, int nnz_coef, const Incidence& inc, const void* B8, const void* B1, int bits,
                    const double* Dinv_s, const double* sc, const double* d_old, const double* coef, double* d_new,
                    const double* z, double* z_new, const double* res, double* res_new, const double* r,
                    const double* w, double* out, int mode);
void launch_pcg_update_init32(hipStream_t s, int N, const double* p, const double* q, const double* rz_part_old,
                              const double* pq_part, double* x, double* r, double* rr_part, double* indefinite,
                              const float* Dinv_f, const double* sc, const double* coef, float* d, float* z,
                              float* res, const double* wown = nullptr);
void launch_pcg_update_noz(hipStream_t s, int N, const double* w, const double* p, const double* q,
                           const double* rz_part_old, const double* pq_part, double* x, double* r, double* rr_part,
                           double* indefinite);
void launch_apply_dinv(hipStream_t s, int N, const double* Dinv, const double* q, double* v);
void launch_scale(hipStream_t s, int n, double a, double* v);
void launch_scale_inv_sqrt(hipStream_t s, int n, const double* sumsq, double* v);
void launch_newton_update(hipStream_t s, int N, const double* dv, double* v, const double* xp, const double* yp,
                          const double* zp, double h, double* x, double* y, double* z);
void launch_axpy_neg(hipStream_t s, int n, const double* g, double* r);
void launch_diff(hipStream_t s, int n, const double* a, const double* b, double* r);  // r = a - b
void launch_adamw_update_velocity(hipStream_t s, int n, const double* g, double beta1, double beta2, double eps,
                                  double weight_decay, double lr, double inv_1mb1t, double inv_1mb2t, double* m,
                                  double* va, double* v);
void launch_nesterov_lookahead(hipStream_t s, int n, double beta, const double* vk, const double* vkm1, double* v);
void launch_nesterov_step(hipStream_t s, int n, double alpha, const double* y, const double* g, double* vnext);
void launch_positions_from_prev(hipStream_t s, int N, const double* v, const double* xp, const double* yp,
                                const double* zp, double dt, double* x, double* y, double* z);
void launch_dual_update(hipStream_t s, int nc, const double* cons, double rho, double* lam);
void launch_pack(hipStream_t s, int n, int dim, const int* node, const int* slot, const double* src, double* buf);
void launch_unpack(hipStream_t s, int n, int dim, const int* node, const int* slot, const double* buf, double* dst);
void launch_extract_diag(hipStream_t s, int N, const Incidence& inc, const double* Hval, double* D);
void launch_invert_diag(hipStream_t s, int N, const double* D, double* Dinv);

// ANCF node-block (12 x 12) scaling of the polynomial's operator (solver_kernels.hip)
void launch_blk12_factor(hipStream_t s, int Np, const Incidence& inc, const double* Hval, double* Linv, float* Linv_f,
                         double* sc, double* Dinv_s, int* err);
void launch_blk12_apply(hipStream_t s, int Np, const float* Linv_f, bool transpose, const double* in, double* out);
void launch_lp_convert12(hipStream_t s, int N, const Incidence& inc, const double* Hval, const double* Linv, void* B8,
                         void* B1, int bits);

// ---- sparse direct solve (direct_kernels.hip on the plan of mf_host.h) ------------------------------------------------
struct MfFrontDev {  // a front as the kernels see it (DOF units)
  long long F_off, L_off, v_off, map_off, rows_off;
  long long cF_off;  // where its PARENT reads its update matrix (the front itself, or its compact copy on the stack) ...
  int m, k, c0, child0, child1;
  int c_ld, c_k0, pad;  // ... with this leading dimension and first row / column
};
struct MfDev {
  const MfFrontDev* fr;
  const int *lvl, *blvl, *map, *rows, *order;  // lvl: fronts by level (solve), blvl: fronts by batch (factorisation)
  const long long *hsrc, *hdst;
  const int *hsld, *hdld;
  double* L;
  double* F[4];  // front workspaces: level buffers (depth parity), WORK, STACK (mf_host.h, MfBatch)
  double *v, *y, *xp;
  int* err;
};
void launch_mf_factor(hipStream_t s, const MfPlan& P, const MfDev& D, const double* H);
void launch_mf_solve(hipStream_t s, const MfPlan& P, const MfDev& D, const double* b, double* x);

}  // namespace tlfea
