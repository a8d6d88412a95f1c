/* tlfea_oracle.h -- CPU ORACLE for the Total-Lagrangian T10 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's algorithm
 * (uwsbel/Total-Lagrangian-FEA, CUDA), used by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py as the checker / timed CPU baseline.  Nothing under
 * total-lagrangian-fea_amd/ may include, link or call it.
 *
 * Parity pin: checked against tests/golden/t10_*.npz, which tools/gen_golden.py produced by
 * importing the reference's own NumPy prototypes (test-scripts/T10-tets/ f-form-T10-beam-newton{,-damped}.py), and against the
 * reference's data fixtures (TetGen meshes, Keast table, remap table).  The linear-solve boundary
 * (cuDSS, absent third-party) is pinned by residual and by the prototype's dense Cholesky steps.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 * Layouts are the reference's device layouts verbatim:
 *   conn      int32  [10][E]  column-major E x 10          (FEAT10Data.cuh:36-39)
 *   gradN     double [E][5][3][10]  10x3 col-major blocks  (FEAT10Data.cuh:41-45)
 *   detJ      double [E][5]
 *   F,P,...   double [E][5][9]  3x3 col-major (i + 3*j)    (FEAT10Data.cuh:114-158)
 *   f_int     double [3N] xyz interleaved
 */
#ifndef TLFEA_ORACLE_H
#define TLFEA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MAT_SVK = 0, ORC_MAT_MOONEY_RIVLIN = 1 }; /* MaterialModel.cuh:14-17 */

typedef struct {
  int model;             /* ORC_MAT_* */
  double lambda, mu;     /* SVK Lame constants (FEAT10Data.cuh:604-605) */
  double mu10, mu01, kappa;
  double eta_damp, lambda_damp;
  double rho0;
} orc_material;

typedef struct {
  double inner_atol, inner_rtol, outer_tol, rho;
  int max_outer, max_inner;
  double time_step;
} orc_newton_params; /* SyncedNewton.cuh:29-33 */

/* Keast 5-point rule (quadrature_utils.h:140-158): fills qx,qy,qz,qw [5]. */
void orc_keast5(double *qx, double *qy, double *qz, double *qw);

/* TetGen -> standard mid-node order (cpu_utils.cc:607-624). */
void orc_t10_remap_tetgen(const int *tetgen10, int *standard10);

/* dn_du_pre_kernel (FEAT10Data.cu:97-204) + solve_3x3_system (FEAT10DataFunc.cuh:30-83). */
void orc_t10_dndu_pre(int E, const int *conn, const double *x, const double *y, const double *z,
                      const double *qx, const double *qy, const double *qz, double *gradN,
                      double *detJ);

/* compute_p (FEAT10DataFunc.cuh:85-293); v may be NULL (no damping, as CalcP does). */
void orc_t10_compute_p(int E, const int *conn, const double *x, const double *y, const double *z,
                       const double *v, const double *gradN, const orc_material *mat, double *F,
                       double *P, double *Fdot, double *Pvis);

/* clear + compute_internal_force (FEAT10DataFunc.cuh:397-466); deterministic element order. */
void orc_t10_internal_force(int E, int N, const int *conn, const double *P, const double *gradN,
                            const double *detJ, const double *qw, double *f_int);

/* Element tangent summed over the 5 QPs, row-major 30x30: K_e (SVK.cuh:35-55 or
 * MooneyRivlin.cuh:113-225) and the Kelvin-Voigt C_e (FEAT10DataFunc.cuh:695-762). */
void orc_t10_element_tangent(int e, int E, const int *conn, const double *x, const double *y,
                             const double *z, const double *gradN, const double *detJ,
                             const double *qw, const orc_material *mat, double *Ke, double *Ce);

/* BuildMassCSRPattern (FEAT10Data.cu:372-440): sorted unique (row,col) node pairs.
 * offsets[N+1] caller-allocated; *columns malloc'd (free with orc_free). Returns nnz. */
int orc_t10_mass_pattern(int E, int N, const int *conn, int *offsets, int **columns);
void orc_free(void *p);

/* mass_matrix_qp_kernel (FEAT10Data.cu:206-278). */
void orc_t10_mass_values(int E, const int *conn, const double *detJ, const double *qx,
                         const double *qy, const double *qz, const double *qw, double rho0,
                         const int *offsets, const int *columns, double *values);

/* DOF-level CSR from coefficient adjacency (SyncedNewton.cu:163-205): rows 3N,
 * row nnz = 3*deg, columns sorted.  row_offsets[3N+1], col_indices[9*nnz_coef]. */
void orc_hessian_pattern(int N, const int *offsets, const int *columns, int *row_offsets,
                         int *col_indices);

/* H = M/h (x)I3 + h*K_t + C_vis + h^2 rho J^T J   (SyncedNewton.cu:214-341,
 * FEAT10DataFunc.cuh:513-791).  Pinned DOFs: fixed_nodes[n_fixed]. nthreads>1 uses OpenMP
 * atomics (timing only; summation order then differs run to run like the CUDA reference). */
void orc_t10_assemble_hessian(int E, int N, const int *conn, const double *x, const double *y,
                              const double *z, const double *gradN, const double *detJ,
                              const double *qw, const orc_material *mat, const int *m_offsets,
                              const int *m_columns, const double *m_values, const int *fixed_nodes,
                              int n_fixed, double h, double rho, const int *row_offsets,
                              const int *col_indices, double *values, int nthreads);

/* solver_grad_L (SyncedNewton.cu:344-407). c = constraint values [3*n_fixed], lam multipliers. */
void orc_grad_L(int N, const int *m_offsets, const int *m_columns, const double *m_values,
                const double *v, const double *v_prev, const double *f_int, const double *f_ext,
                const int *fixed_nodes, int n_fixed, const double *c, const double *lam, double h,
                double rho, double *g);

/* Direct SPD solve of the upper-triangle view of CSR H (what cuDSS is asked for,
 * SyncedNewton.cu:1011-1014): RCM + skyline Cholesky. Returns 0 on success. */
int orc_solve_spd_upper(int n, const int *row_offsets, const int *col_indices,
                        const double *values, const double *rhs, double *sol);

/* Jacobi(3x3 block)-preconditioned CG on the full CSR (CPU twin of the HIP solver; used for the
 * timed baseline at sizes where the skyline factor is too slow). Returns iterations. */
int orc_solve_pcg(int n, const int *row_offsets, const int *col_indices, const double *values,
                  const double *rhs, double *sol, double rel_tol, int max_iter, int nthreads);

/* One implicit step = SyncedNewtonSolver::OneStepNewtonCuDSS, T10 branch
 * (SyncedNewton.cu:1032-1146).  State in/out: x,y,z (positions), v (v_guess), v_prev, lam.
 * x_tgt/y_tgt/z_tgt = constraint targets (x12_jac).  solver: 0 = direct, 1 = PCG(rel 1e-13).
 * stats[0]=outer iterations, stats[1]=total newton iterations(solves), stats[2]=last ||g||,
 * stats[3]=last ||c||. */
int orc_t10_newton_step(int E, int N, const int *conn, double *x, double *y, double *z,
                        const double *x_tgt, const double *y_tgt, const double *z_tgt,
                        const double *gradN, const double *detJ, const double *qw,
                        const orc_material *mat, const int *m_offsets, const int *m_columns,
                        const double *m_values, const int *fixed_nodes, int n_fixed,
                        const double *f_ext, const orc_newton_params *prm, double *v,
                        double *v_prev, double *lam, int solver, int nthreads, double *stats);

/* ---- general linear constraints c = J x - rhs (kConstraintLinearCSR; ANCF3243Data.cuh:803-940,
 * ANCF3243DataFunc.cuh:477-499, SyncedNewton.cu:292-341,377-404,556-801).  J: CSR over constraint rows, columns =
 * 3*coef + component.  No reference run pins these (the Python prototypes have no such constraints): the restatement
 * follows the cited lines and is cross-checked against the fixed-coefficient path (a pinned DOF written as a row). */
int orc_lin_adjacency(int N, const int *mo, const int *mc, int nc, const int *joff, const int *jcol, int *offsets,
                      int **columns);
void orc_lin_constraint(int nc, const int *joff, const int *jcol, const double *jval, const double *rhs,
                        const double *x, const double *y, const double *z, double *c);
void orc_lin_grad_add(int nc, const int *joff, const int *jcol, const double *jval, const double *c,
                      const double *lam, double h, double rho, double *g);
void orc_gen_assemble_hessian_lin(int S, int Q, int E, int N, const int *conn, const double *x, const double *y,
                                  const double *z, const double *gradN, const double *detJ, const double *qw,
                                  const orc_material *mat, const int *mo, const int *mc, const double *mv,
                                  const int *ao, const int *ac, int nc, const int *joff, const int *jcol,
                                  const double *jval, double h, double rho, const int *ro, const int *ci,
                                  double *val);
int orc_gen_newton_step_lin(int S, int Q, int E, int N, const int *conn, double *x, double *y, double *z,
                            const double *gradN, const double *detJ, const double *qw, const orc_material *mat,
                            const int *mo, const int *mc, const double *mv, int nc, const int *joff, const int *jcol,
                            const double *jval, const double *rhs, const double *f_ext,
                            const orc_newton_params *prm, double *v, double *v_prev, double *lam, double *stats);

/* ---- SyncedVBDSolver (SyncedVBD.cuh:13-21 parameter order) ---- */
typedef struct {
  double inner_tol, inner_rtol, outer_tol, rho;
  int max_outer, max_inner;
  double time_step, omega, hess_eps;
  int convergence_check_interval, color_group_size;
} orc_vbd_params;
void orc_gen_vbd_node_terms(int S, int Q, int E, int N, const int *conn, const double *x, const double *y, const double *z,
                            const double *v, const double *gradN, const double *detJ, const double *qw,
                            const orc_material *mat, double h, double *r_out, double *K_out);
int orc_gen_vbd_step(int S, int Q, int E, int N, const int *conn, double *x, double *y, double *z, const double *xt,
                     const double *yt, const double *zt, const double *gradN, const double *detJ, const double *qw,
                     const orc_material *mat, const int *mo, const int *mc, const double *mv, const int *fixed,
                     int n_fixed, const double *f_ext, const orc_vbd_params *prm, int n_colors, const int *color_offsets,
                     const int *color_nodes, int n_groups, const int *group_offsets, const int *group_colors, double *v,
                     double *v_prev, double *lam, double *stats);
/* tlfea_oracle_coloring.cc (C++: the reference orders nodes with std::sort, whose tie order is the library's) */
int orc_vbd_coloring(int S, int E, int N, const int *conn, int *colors);
int orc_vbd_validate_coloring(int S, int E, const int *conn, const int *colors);
int orc_vbd_color_groups(int S, int E, const int *conn, const int *colors, int n_colors, int group_size,
                         int *group_offsets, int *group_colors);

#ifdef __cplusplus
}
#endif
#endif
