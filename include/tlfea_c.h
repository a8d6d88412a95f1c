/* tlfea_c.h -- C-ABI of the MI355X-native Total-Lagrangian element engine (libtlfea_hip.so).
 *
 * This is the drop-in boundary for the T10 hot path of uwsbel/Total-Lagrangian-FEA.  The reference
 * has no FFI: its boundary is the C++ class surface the drivers in lib_bin/ call
 * (GPU_FEAT10_Data, SyncedNewtonSolver).  Every entry point below replaces one member function of
 * those classes (cited as file:line, paths relative to the reference root) with a handle-based
 * `extern "C"` function taking plain pointers and sizes.  The header-only C++ facade in
 * total-lagrangian-fea_amd/host/ re-creates the reference class names on top of this ABI, and
 * total-lagrangian-fea_amd/__init__.py binds it with ctypes.
 *
 * Conventions
 *   - all floating point is fp64, all indices int32, zero based;
 *   - host pointers unless the name says `_device_ptr`;
 *   - every function returns 0 on success, non-zero on failure (tlfea_last_error() gives text);
 *     API misuse mirrors the reference: message on stderr + early return (non-zero here);
 *   - one handle <-> one GPU (the current HIP device at create time), default stream, blocking,
 *     not thread-safe (as the reference: FEAT10Data.cu:291,311,332);
 *   - ownership: the data handle owns every device buffer until tlfea_t10_destroy(); a solver
 *     handle borrows the data handle, which must outlive it (SyncedNewton.cuh:37-46).
 *   - layouts handed across the ABI are the reference's: connectivity column-major E x 10
 *     (FEAT10Data.cuh:36-39), grad-N blocks 10x3 column-major per (elem,qp) (:41-45), F/P 3x3
 *     column-major per (elem,qp) (:114-158), forces/velocities xyz-interleaved per node.
 */
#ifndef TLFEA_C_H
#define TLFEA_C_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tlfea_t10_s *tlfea_t10_t;       /* GPU_FEAT10_Data   (FEAT10Data.cuh:19)   */
typedef struct tlfea_newton_s *tlfea_newton_t; /* SyncedNewtonSolver (SyncedNewton.cuh:35) */

/* SyncedNewtonParams (SyncedNewton.cuh:29-33) -- identical field order. */
typedef struct {
  double inner_atol, inner_rtol, outer_tol, rho;
  int max_outer, max_inner;
  double time_step;
} tlfea_newton_params;

/* Linear-solve options.  The reference calls cuDSS (sparse Cholesky, SyncedNewton.cu:995-1029,
 * 1103-1114); this engine solves the same SPD system H dv = -g on the device: by default with CG in fp64 on H,
 * preconditioned by a two-level p-multigrid cycle (T10) or a Chebyshev polynomial of the block-Jacobi-scaled
 * operator (all kinds), both streaming a scaled fp16 copy of H with fp32 work vectors; method = 1 selects the sparse
 * direct solve (the engine's multifrontal Cholesky: analysis once per mesh, re-factorisation + solve per call, as the
 * reference drives cuDSS).  rel_tol is on ||r||/||b|| of the fp64 system. */
typedef struct {
  double rel_tol;    /* default 1e-12 */
  int max_iter;      /* default 20000 (outer CG iterations) */
  int check_every;   /* outer iterations between host convergence checks (default 25) */
  int cheb_degree;   /* degree of the Chebyshev polynomial of D^-1 H used as preconditioner; 1 = plain block-Jacobi;
                        0 (default) = auto = 24 */
  double cheb_kappa; /* the polynomial targets [lmax/kappa, lmax] of D^-1 H; <= 1 (default 0) = auto by degree
                        (400 / 800 / 1600 for degree < 14 / < 22 / above) */
  int cheb_bits;     /* storage precision of the matrix the polynomial streams: 64 = H itself; 32 / 16 = a symmetrically
                        scaled fp32 / fp16 block copy, and on one GPU the polynomial's recurrence then runs in fp32
                        (the outer CG, its residual and its convergence test stay fp64 on H);
                        0 (default) = auto = 16 */
  int precond;       /* 0 (default) = auto, 1 = Chebyshev polynomial of block-Jacobi, 2 = two-level p-multigrid (T10 on one
                        GPU: quadratic tets -> their vertex mesh, Galerkin coarse operator, polynomial smoothers and
                        coarse solve; auto picks it wherever it exists) */
  int method;        /* 0 (default) = preconditioned CG; 1 = sparse direct (single GPU) */
  int on_unconverged;/* a CG solve that ends above rel_tol (max_iter reached, or breakdown after the interval widenings):
                        0 (default) = the call fails, nothing is applied to v and x (the reference aborts when cuDSS
                        fails, SyncedNewton.cu:995-1029); 1 = accept the iterate (experiments with a capped iteration
                        count) -- tlfea_newton_get_linsolve_status reports it */
} tlfea_linsolve_opts;

const char *tlfea_last_error(void);
int tlfea_version(void);
/* number of visible HIP devices; <=0 means the product path cannot run (it never falls back to CPU) */
int tlfea_device_count(void);

/* ---- GPU_FEAT10_Data ------------------------------------------------------------------------ */
/* ctor (FEAT10Data.cuh:377-380) + Initialize() (:382-433) */
int tlfea_t10_create(int n_elem, int n_nodes, tlfea_t10_t *out);
/* Destroy() (:714-779) */
int tlfea_t10_destroy(tlfea_t10_t h);
/* Setup(tet5pt_x,y,z,w, x12,y12,z12, element_connectivity) (:435-533); conn column-major E x 10 */
int tlfea_t10_setup(tlfea_t10_t h, const double *qx, const double *qy, const double *qz,
                    const double *qw, const double *x, const double *y, const double *z,
                    const int *conn_colmajor);
int tlfea_t10_set_density(tlfea_t10_t h, double rho0);                       /* :538-546 */
int tlfea_t10_set_damping(tlfea_t10_t h, double eta_damp, double lambda_damp); /* :553-563 */
int tlfea_t10_set_svk_select(tlfea_t10_t h);                                   /* SetSVK() :568-587 */
int tlfea_t10_set_svk(tlfea_t10_t h, double E, double nu);                     /* SetSVK(E,nu) :594-611 */
int tlfea_t10_set_mooney_rivlin(tlfea_t10_t h, double mu10, double mu01, double kappa); /* :618-634 */
int tlfea_t10_set_external_force(tlfea_t10_t h, const double *f_ext, int n);   /* :636-646 (n must be 3N) */
int tlfea_t10_set_nodal_fixed(tlfea_t10_t h, const int *fixed_nodes, int n_fixed);    /* FEAT10Data.cu:728-749 */
int tlfea_t10_update_nodal_fixed(tlfea_t10_t h, const int *fixed_nodes, int n_fixed); /* FEAT10Data.cu:751-832 */
int tlfea_t10_update_positions(tlfea_t10_t h, const double *x, const double *y, const double *z, int n);          /* :671-685 */
int tlfea_t10_update_constraint_targets(tlfea_t10_t h, const double *x, const double *y, const double *z, int n); /* :687-701 */

int tlfea_t10_calc_dndu_pre(tlfea_t10_t h);            /* CalcDnDuPre        FEAT10Data.cu:284-292 */
int tlfea_t10_build_mass_csr_pattern(tlfea_t10_t h);   /* BuildMassCSRPattern FEAT10Data.cu:372-440 */
int tlfea_t10_calc_mass_matrix(tlfea_t10_t h);         /* CalcMassMatrix     FEAT10Data.cu:351-370 */
int tlfea_t10_calc_constraint_data(tlfea_t10_t h);     /* CalcConstraintData FEAT10Data.cu:335-349 */
int tlfea_t10_convert_to_csr_constraint_jac(tlfea_t10_t h);  /* FEAT10Data.cu:501-534 */
int tlfea_t10_convert_to_csr_constraint_jact(tlfea_t10_t h); /* FEAT10Data.cu:538-601 */
int tlfea_t10_calc_p(tlfea_t10_t h);                   /* CalcP              FEAT10Data.cu:307-312 */
int tlfea_t10_calc_internal_force(tlfea_t10_t h);      /* CalcInternalForce  FEAT10Data.cu:328-333 */

int tlfea_t10_get_n_elem(tlfea_t10_t h);
int tlfea_t10_get_n_coef(tlfea_t10_t h);
int tlfea_t10_get_n_constraint(tlfea_t10_t h);
int tlfea_t10_is_constraint_setup(tlfea_t10_t h); /* Get_Is_Constraint_Setup :785-787 */

/* Retrieve*ToCPU (FEAT10Data.cu:603-726,834-839); flat arrays in the reference's device layouts */
int tlfea_t10_mass_csr_nnz(tlfea_t10_t h, int *nnz);
int tlfea_t10_retrieve_mass_csr(tlfea_t10_t h, int *offsets /*N+1*/, int *columns /*nnz*/, double *values /*nnz*/);
int tlfea_t10_retrieve_internal_force(tlfea_t10_t h, double *f /*3N*/);
int tlfea_t10_retrieve_external_force(tlfea_t10_t h, double *f /*3N*/);
int tlfea_t10_retrieve_position(tlfea_t10_t h, double *x, double *y, double *z);
int tlfea_t10_retrieve_p_from_f(tlfea_t10_t h, double *P /*E*5*9*/);
int tlfea_t10_retrieve_deformation_gradient(tlfea_t10_t h, double *F /*E*5*9*/);
int tlfea_t10_retrieve_dndu_pre(tlfea_t10_t h, double *gradN /*E*5*30*/);
int tlfea_t10_retrieve_detj(tlfea_t10_t h, double *detJ /*E*5*/);
int tlfea_t10_retrieve_connectivity(tlfea_t10_t h, int *conn_colmajor /*E*10*/);
int tlfea_t10_retrieve_constraint_data(tlfea_t10_t h, double *c /*n_constraint*/);
int tlfea_t10_retrieve_constraint_jac_csr(tlfea_t10_t h, int *offsets /*nc+1*/, int *columns /*nc*/, double *values /*nc*/);
int tlfea_t10_retrieve_constraint_jact_csr(tlfea_t10_t h, int *offsets /*3N+1*/, int *columns /*nc*/, double *values /*nc*/);
int tlfea_t10_write_output_vtk(tlfea_t10_t h, const char *filename); /* FEAT10Data.cu:841-877 */

/* device pointers (FEAT10Data.cuh:648-666,781-783) */
const double *tlfea_t10_x12_device_ptr(tlfea_t10_t h);
const double *tlfea_t10_y12_device_ptr(tlfea_t10_t h);
const double *tlfea_t10_z12_device_ptr(tlfea_t10_t h);
double *tlfea_t10_external_force_device_ptr(tlfea_t10_t h);
double *tlfea_t10_constraint_device_ptr(tlfea_t10_t h);

/* ---- GPU_ANCF3243_Data / GPU_ANCF3443_Data -------------------------------------------------------------
 * The ANCF element types share the handle type and every `tlfea_t10_*` entry point above that is not
 * T10-specific (material setters, SetExternalForce, SetNodalFixed -- which takes COEFFICIENT indices for these
 * types, ANCF3243Data.cuh:778-808 --, CalcMassMatrix, CalcP, CalcInternalForce, CalcConstraintData, Retrieve*,
 * Destroy, and the whole SyncedNewtonSolver block below).  "Nodes" there means coefficient vectors
 * (n_coef = 4 * n_nodes; coefficient index = 4*node + slot, DOF = 3*coef + xyz).  Array sizes follow
 * tlfea_elem_dims(): gradients [E][Q][3][S], F/P [E][Q][9], connectivity [S][E] coefficient ids. */
int tlfea_ancf_create(int kind /*3243|3443*/, int n_nodes, int n_elements, tlfea_t10_t *out); /* ANCF3243Data.cuh:434-509, ANCF3443Data.cuh:445-520 */
/* Setup(L,W,H, mass rule, force rule, x12,y12,z12, connectivity) (ANCF3243Data.cuh:511-670, ANCF3443Data.cuh:522-670);
 * L,W,H per element; nqm/nq = {n_xi, n_eta, n_zeta}; conn_nodes E x nn (row-major unless conn_is_colmajor) */
int tlfea_ancf_setup(tlfea_t10_t h, const double *L, const double *W, const double *H, const double *gauss_xi_m,
                     const double *gauss_eta_m, const double *gauss_zeta_m, const double *weight_xi_m,
                     const double *weight_eta_m, const double *weight_zeta_m, const int *nqm,
                     const double *gauss_xi, const double *gauss_eta, const double *gauss_zeta,
                     const double *weight_xi, const double *weight_eta, const double *weight_zeta, const int *nq,
                     const double *x12, const double *y12, const double *z12, const int *conn_nodes,
                     int conn_is_colmajor);
int tlfea_ancf_calc_dsdu_pre(tlfea_t10_t h);
/* ANCF3243_B12_matrix / ANCF3443_B12_matrix (cpu_utils.cc:125-188, 211-420): the (B^T)^-1 matrix of one element,
 * column-major S x S (S = 8 | 16) -- the block layout of ANCF3243_B12_matrix_flat_per_element (cpu_utils.cc:190-209).
 * Host-only (no GPU touched); tlfea_ancf_setup computes the same blocks itself from L, W, H. */
int tlfea_ancf_b12_matrix(int kind /*3243|3443*/, double L, double W, double H, double *out_colmajor);
/* SetLinearConstraintsCSR (ANCF3243Data.cuh:810-940, ANCF3443Data.cuh same member): general linear constraints
 * c = J x - rhs, J in CSR over constraint rows, columns in the flattened DOF space (3*coef + component).  Accepted
 * on every element kind.  Must be called before BuildMassCSRPattern / CalcMassMatrix: the Hessian pattern includes
 * the coefficient pairs a row couples (SyncedNewton.cu:556-801).  Single-GPU path only. */
int tlfea_t10_set_linear_constraints_csr(tlfea_t10_t h, int n_rows, const int *offsets, const int *columns,
                                         const double *values, const double *rhs);
/* UpdateLinearConstraintRHS (ANCF3243Data.cuh / ANCF3443Data.cuh:977-997): new right-hand side of the CSR constraints,
 * J / J^T and the sparsity stay (prescribed motion: the airless-tire driver rotates its hub this way every step) */
int tlfea_t10_update_linear_constraint_rhs(tlfea_t10_t h, const double *rhs /*n_constraint*/, int n);
/* GetConstraintMode (ANCF3243Data.cuh:436-441): 0 none, 1 kConstraintFixedCoefficients, 2 kConstraintLinearCSR */
int tlfea_t10_get_constraint_mode(tlfea_t10_t h);
/* nnz of J (sizes the buffers of tlfea_t10_retrieve_constraint_jac_csr / _jact_csr) */
int tlfea_t10_constraint_jac_nnz(tlfea_t10_t h); /* CalcDsDuPre  ANCF3243Data.cu:290-300, ANCF3443Data.cu:256-266 */
int tlfea_elem_dims(tlfea_t10_t h, int *S /*shape functions*/, int *Q /*force quadrature points*/);

/* ---- SyncedNewtonSolver --------------------------------------------------------------------- */
int tlfea_newton_create(tlfea_t10_t data, int n_constraints, tlfea_newton_t *out); /* SyncedNewton.cuh:37-149 */
int tlfea_newton_destroy(tlfea_newton_t s);                                        /* dtor :151-204 */
int tlfea_newton_setup(tlfea_newton_t s);                                          /* Setup :231-245 */
int tlfea_newton_set_parameters(tlfea_newton_t s, const tlfea_newton_params *p);   /* SetParameters :206-229 */
int tlfea_newton_analyze_hessian_sparsity(tlfea_newton_t s);                       /* SyncedNewton.cu:546-907 */
int tlfea_newton_set_fixed_sparsity_pattern(tlfea_newton_t s, int fixed);          /* :351-353 */
int tlfea_newton_solve(tlfea_newton_t s);            /* Solve()/OneStepNewtonCuDSS  SyncedNewton.cu:909-1394 */
double *tlfea_newton_velocity_guess_device_ptr(tlfea_newton_t s);                  /* :341-343 */
/* compute_l2_norm_cublas (SyncedNewton.cu:536-544) on a device vector */
int tlfea_newton_l2_norm(tlfea_newton_t s, const double *d_vec, int n, double *out);

/* ---- engine extras (no reference counterpart; used by tests/bench/INTEGRATION) -------------- */
int tlfea_newton_set_linsolve_opts(tlfea_newton_t s, const tlfea_linsolve_opts *o);
/* what the auto rules resolved to for this mesh: polynomial degree (1 = block-Jacobi), matrix bits of its steps and
 * the precision of its work vectors (32 on the single-GPU low-precision path, else 64) */
int tlfea_newton_get_linsolve_info(tlfea_newton_t s, int *cheb_degree, int *cheb_bits, int *cheb_vector_bits);
/* preconditioner a solve would use now: 0 block-Jacobi, 1 Chebyshev polynomial, 2 p-multigrid */
int tlfea_newton_get_precond(tlfea_newton_t s);
/* how compute_hessian_assemble_csr (FEAT10DataFunc.cuh:513-791) + the CSR scatter (SyncedNewton.cu:214-341) run now:
 * 1 = tangent blocks into an element-major buffer + row-owner gather (two launches; Mooney-Rivlin, ANCF kinds),
 * 2 = fused row-owner tangent + assembly (one launch, no block buffer; T10 with St.Venant-Kirchhoff) */
int tlfea_newton_get_assembly_mode(tlfea_newton_t s);
/* degree of the coarse-level polynomial of the p-multigrid cycle (grows with the coarse mesh); 0 without p-multigrid */
/* third level of the cycle (rigid-body-mode aggregates of the vertex level; opt-in, TLFEA_PMG_LEVELS=3): number of
 * aggregates (0: two levels), 3x3 blocks of H3 (2 nodes per aggregate: translation, rotation), polynomial degree there;
 * retrieve: aggregate of every vertex node, x_i - c_A (zero where rotations are off), usable-rotation flags, the
 * level-3 block CSR and H3 = P2^T Hc P2 in the DOF-level layout of the other levels */
int tlfea_newton_pmg3_sizes(tlfea_newton_t s, int *n_aggregates, int *nnz_blocks, int *degree);
int tlfea_newton_pmg3_retrieve(tlfea_newton_t s, int *agg, double *rvec, int *active, int *off3, int *cols3, double *H3);
int tlfea_newton_pmg_coarse_degree(tlfea_newton_t s);
/* p-multigrid test hooks: coarse sizes; parent map [N] x 2, coarse block-CSR pattern and Hc = P^T H P of the current H
 * (9 nnz values in the DOF-level layout of H: node row -> [d][k][e]) */
int tlfea_newton_pmg_sizes(tlfea_newton_t s, int *n_coarse, int *nnz_coarse_blocks);
int tlfea_newton_pmg_retrieve(tlfea_newton_t s, int *par0, int *par1, int *c_off, int *c_cols, double *Hc);
int tlfea_newton_hessian_nnz(tlfea_newton_t s, int *nnz);
/* H in the reference's DOF-level CSR (SyncedNewton.cu:163-205): rows 3N, sorted columns */
int tlfea_newton_retrieve_hessian_csr(tlfea_newton_t s, int *row_offsets, int *col_indices, double *values);
/* One residual evaluation at the current state: compute_p + internal force + constraints + grad L
 * (SyncedNewton.cu:1046-1065). Returns ||g||. */
int tlfea_newton_eval_gradient(tlfea_newton_t s, double *norm_g);
/* One Hessian assembly at the current state (SyncedNewton.cu:1080-1099). */
int tlfea_newton_assemble_hessian(tlfea_newton_t s);
/* Solve H x = b for host vectors (b,x length 3N) with the current H; iterations returned. */
int tlfea_newton_linear_solve(tlfea_newton_t s, const double *b, double *x, int *iters, double *rel_res);
/* mean duration (ms) of the hot kernels over `reps` back-to-back launches each (hipEvent pair per kernel on the
 * launch stream): [0] residual, [1] tangent blocks (0 in assembly mode 2), [2] row assembly (mode 2: the fused
 * tangent + assembly launch), [3] CG SpMV (fp64), [4] fine-level polynomial /
 * smoother step, [5] coarse-level polynomial step of the p-multigrid cycle, [6] the same kernel averaged over the
 * launch pattern of one V-cycle (3 fine + kc-1 coarse launches; [5], [6] are 0 without p-multigrid).  State of the
 * Newton iteration is unchanged (only linear-solver work vectors are touched). */
int tlfea_newton_time_kernels(tlfea_newton_t s, int reps, double *out_ms7);
/* y = H x with the current H, host vectors of 3N (partition-boundary rows summed over ranks). */
int tlfea_newton_apply_hessian(tlfea_newton_t s, const double *x, double *y);
/* One full Newton iteration without the convergence test (gradient, assembly, solve, update):
 * the unit bench.py times.  iters = PCG iterations used. */
int tlfea_newton_iteration(tlfea_newton_t s, double *norm_g, int *iters);
int tlfea_newton_retrieve_gradient(tlfea_newton_t s, double *g /*3N*/);
int tlfea_newton_retrieve_velocity(tlfea_newton_t s, double *v /*3N*/);
int tlfea_newton_set_velocity(tlfea_newton_t s, const double *v /*3N*/, const double *v_prev /*3N or NULL*/);
int tlfea_newton_set_lambda(tlfea_newton_t s, const double *lam /*n_constraints multipliers*/);
int tlfea_newton_retrieve_lambda(tlfea_newton_t s, double *lam /*n_constraints*/);
/* stats of the last tlfea_newton_solve(): [0] outer iterations, [1] Newton solves, [2] last ||g||,
 * [3] last ||c||, [4] total PCG iterations, [5] device ms of the step (hipEvent) */
int tlfea_newton_get_stats(tlfea_newton_t s, double *stats6);
/* all-reduce calls the solver has issued since it was built (any exchange path); multi-GPU tests divide by CG iterations */
long tlfea_newton_collectives(tlfea_newton_t s);
/* number of multipliers the solver holds now (follows the data object's count at the start of every solve: an
 * UpdateNodalFixed with another size restarts them from zero) */
int tlfea_newton_n_constraints(tlfea_newton_t s);
/* linear solves since the last tlfea_newton_solve() / tlfea_newton_iteration() began: out4 = [0] ||r||/||b|| of the last
 * one, [1] 1 if it met rel_tol, [2] the worst ||r||/||b|| of them, [3] 1 if all of them met rel_tol */
int tlfea_newton_get_linsolve_status(tlfea_newton_t s, double *out4);
/* per-stage device time (ms, hipEvent pairs on the launch stream; profiling mode) and launch counts since
 * the last reset: [0] residual kernel, [1] gather+grad+norm, [2] tangent-block kernel, [3] row-assembly
 * kernel, [4] whole PCG solve, [5] update kernel, [6] SpMV kernel alone, [7] unused */
int tlfea_newton_get_stage_ms(tlfea_newton_t s, double *ms8, double *counts8, int reset);
/* start of an implicit step when driving Newton iterations by hand: x_prev <- x, v_prev <- v */
int tlfea_newton_begin_step(tlfea_newton_t s);
int tlfea_newton_set_verbose(tlfea_newton_t s, int verbose);
/* per-stage hipEvent timing for tlfea_newton_get_stage_ms (one host sync per stage; off by default) */
int tlfea_newton_set_profiling(tlfea_newton_t s, int on);

/* Multi-GPU hook: partition-boundary exchange (one process per GPU; elements are owned by one rank, nodes on
 * partition boundaries are replicated).  `iface_nodes[k]` (local node id) sits at `iface_slots[k]` of a GLOBAL
 * interface list of `n_global` nodes that is identical on every rank; `node_weight[i]` = 1/(number of ranks
 * holding node i).  The solver then calls `fn(user, d_buf, n)` -- in-place SUM over ranks of a device buffer
 * of n doubles -- exactly where the path needs it: once for the gradient (boundary DOFs), once for the
 * boundary diagonal blocks, and twice per CG iteration (boundary rows of H p fused with the p.Hp slots; the
 * r.z / r.r slots).  The Python host layer points fn at torch.distributed.all_reduce (RCCL over xGMI, or
 * gloo through a host staging copy).  f_ext must already be this rank's share (weight-scaled on replicated
 * nodes).  sync_before_callback=1 drains the stream before every call (needed unless fn enqueues on the
 * null stream, as torch's default stream does). */
typedef int (*tlfea_allreduce_fn)(void *user, double *d_buf, int n);
int tlfea_newton_set_interface(tlfea_newton_t s, const int *iface_nodes, const int *iface_slots, int n_local,
                               int n_global, const double *node_weight /*N*/, tlfea_allreduce_fn fn,
                               void *user, int sync_before_callback);
/* Built-in all-reduce for tlfea_newton_set_interface: RCCL called from C++ on the solver's launch stream -- no host
 * language in the loop, no host synchronisation (opt-in; the Python host layer's default goes through torch.distributed).
 * The RCCL library is resolved at run time (the copy already mapped into the process -- e.g. a PyTorch wheel's -- else
 * librccl.so from the loader path), so the engine carries no link-time dependency on it.
 *   rank 0: tlfea_rccl_unique_id(id) -> ship the 128 bytes to every rank -> tlfea_rccl_comm_create(id, rank, world, &comm)
 *   tlfea_newton_set_interface(..., tlfea_rccl_allreduce_fn(), comm, 0) ;  tlfea_rccl_comm_destroy(comm) at the end */
int tlfea_rccl_unique_id(char *id128);
int tlfea_rccl_comm_create(const char *id128, int rank, int world, void **comm_out);
int tlfea_rccl_comm_destroy(void *comm);
tlfea_allreduce_fn tlfea_rccl_allreduce_fn(void);
/* Owners of the replicated partition-boundary nodes: owned[N] = 1 where this rank owns the node (every node has exactly
 * one owner over all ranks; interior nodes: 1).  Makes the polynomial preconditioner rank-local (block-Jacobi over ranks:
 * no exchange inside the polynomial, one packed all-reduce per CG iteration for its result) instead of one exchange per
 * polynomial step.  Call after tlfea_newton_set_interface. */
int tlfea_newton_set_interface_owners(tlfea_newton_t s, const int *owned);

/* ---- Overlapping partition: owner-computes with ghost layers (the scalable multi-GPU path; the reference has no
 * counterpart -- it is a single-GPU code; SURVEY.md section 8e / north_star ask for it) ---------------------------------
 * Every node has ONE owning rank.  A rank's mesh (the data object it was built on) holds its owned nodes, `depth` layers
 * of ghost nodes (layer k = graph distance k from the owned set, nodes adjacent when they share an element) and every
 * element touching a node of layer < depth; local numbering: owned nodes first, then ghosts by increasing layer
 * (`node_layer` must be non-decreasing).  Rows of H, M and grad L of layers < depth are then complete on the rank:
 * nothing is summed over ranks.  The solver instead REFRESHES ghost values from their owners, neighbour to neighbour
 * (payload independent of the number of ranks), only where a step needs them: one nodal vector per CG iteration on the
 * fine level, the coarse polynomial's vectors once per `depth` steps (the overlap is computed redundantly in between),
 * the Newton update once per iteration, the diagonal blocks once per solve; dot products weigh owned DOFs 1 and ghosts 0
 * and are summed with `allreduce` (two fixed-size calls per CG iteration).  f_ext, fixed nodes and velocities are given
 * in full on every node the rank holds (no 1/multiplicity shares).
 * Lists: for peer k (`peers[k]`), send_nodes[send_off[k] .. send_off[k+1]) = my owned nodes that peer holds as ghosts,
 * ordered by (layer ON THE PEER = send_layer, global id); recv_nodes[...] = my ghosts owned by that peer ordered by
 * (own layer, global id) -- the peer's send list in the same order, so 'layers <= D' is a prefix on both sides.
 * exchange(user, d_send, d_recv, n_peers, peers, send_off, recv_off): for every peer k send bytes
 * [send_off[k], send_off[k+1]) of d_send to it and receive bytes [recv_off[k], recv_off[k+1]) of d_recv from it (device
 * buffers; empty ranges are skipped on both sides).  sync_before_callback as in tlfea_newton_set_interface. */
typedef int (*tlfea_halo_exchange_fn)(void *user, const void *d_send, void *d_recv, int n_peers, const int *peers,
                                      const long long *send_off, const long long *recv_off);
typedef struct {
  int n_peers;
  const int *peers;
  const int *send_off, *send_nodes, *send_layer;
  const int *recv_off, *recv_nodes;
  int rank, world; /* this rank and the number of ranks (world <= 0: unknown -- the replicated third multigrid level, which
                      gathers a few numbers per rank through the all-reduce, is then left out) */
} tlfea_halo_lists;
int tlfea_newton_set_halo(tlfea_newton_t s, const int *node_layer /*N*/, int depth, const tlfea_halo_lists *lists,
                          tlfea_allreduce_fn allreduce, tlfea_halo_exchange_fn exchange, void *user,
                          int sync_before_callback);
/* Built-in exchange on a communicator of tlfea_rccl_comm_create: one ncclGroup of ncclSend / ncclRecv pairs per refresh,
 * ncclAllReduce for the dot-product slots, all enqueued from C++ on the solver's launch stream, so the whole CG
 * iteration -- collectives included -- is captured in the solver's hipGraphs.  user = the communicator. */
tlfea_halo_exchange_fn tlfea_rccl_halo_exchange_fn(void);
/* Known-answer check of a fresh communicator (an all-reduce and a ring send/recv with values a rank can verify) under a
 * watchdog: returns 0 when every rank saw the right answers; on a wrong answer or when the collectives have not
 * completed after timeout_s seconds it prints the reason to stderr and terminates the PROCESS with exit code 97 (a
 * collective that never completes cannot be abandoned safely) -- the launcher sees a failed rank instead of a hang. */
int tlfea_rccl_self_check(void *comm, int rank, int world, double timeout_s);
/* tlfea_rccl_comm_create under the same watchdog (ncclCommInitRank blocks until every rank has joined). */
int tlfea_rccl_comm_create_timeout(const char *id128, int rank, int world, double timeout_s, void **comm_out);
/* Communication since the solver was built.  out8: [0] ghost refreshes (neighbour exchanges), [1] all-reduces,
 * [2] bytes sent in refreshes, [3] bytes all-reduced, [4] ms in exchanges + all-reduces measured with hipEvents on the
 * launch stream (only while tlfea_newton_set_profiling is on), [5] CG iterations, [6] / [7] the refreshes / all-reduces
 * issued INSIDE CG iterations (the per-iteration budget; the rest is per-solve set-up: diagonal blocks, lambda_max
 * estimates, the Newton update). */
int tlfea_newton_get_comm_stats(tlfea_newton_t s, double *out8);
/* Shape of the p-multigrid cycle a solve would run now (after the first solve or tlfea_newton_pmg_sizes): out6 = levels
 * (0: polynomial preconditioner), fine smoother terms, vertex-level smoother terms (three levels), vertex-level polynomial
 * degree (two levels), level-3 polynomial degree, level-3 nodes. */
int tlfea_newton_pmg_cycle_info(tlfea_newton_t s, int *out6);
/* Polynomial preconditioner in use (no reference counterpart: the reference calls cuDSS): out3 = degree, interval ratio
 * kappa (rounded), block size of the diagonal scaling of its operator -- 3 (per coefficient vector / node) or 12 (ANCF:
 * the four coefficient vectors of a node together, as a change of variables L^-1 H L^-T). */
int tlfea_newton_polynomial_info(tlfea_newton_t s, int *out3);

/* ---- SyncedAdamWNocoopSolver (SyncedAdamWNocoop.cuh:22-198, SyncedAdamWNocoop.cu:262-500) ------------------------
 * First-order ALM solver on the same velocity unknowns: per inner iteration one AdamW moment update, x = x_prev + dt v,
 * compute_p + internal force + constraints + grad L (the Newton solver's own residual path).  Single GPU. */
typedef struct tlfea_adamw_s *tlfea_adamw_t;
typedef struct { /* SyncedAdamWParams, field order of SyncedAdamW.cuh:27-34 */
  double lr, beta1, beta2, eps, weight_decay, lr_decay;
  double inner_tol, outer_tol, rho;
  int max_outer, max_inner;
  double time_step;
  int convergence_check_interval;
  double inner_rtol;
} tlfea_adamw_params;
int tlfea_adamw_create(tlfea_t10_t data, int n_constraints, tlfea_adamw_t *out); /* ctor SyncedAdamWNocoop.cuh:24-104 */
int tlfea_adamw_destroy(tlfea_adamw_t a);
int tlfea_adamw_setup(tlfea_adamw_t a);                                           /* Setup :180-198 */
int tlfea_adamw_set_parameters(tlfea_adamw_t a, const tlfea_adamw_params *p);     /* SetParameters :147-178 (also zeroes v, v_prev, lambda) */
int tlfea_adamw_solve(tlfea_adamw_t a);                                           /* Solve()/OneStepAdamWNocoop SyncedAdamWNocoop.cu:262-500 */
double *tlfea_adamw_velocity_guess_device_ptr(tlfea_adamw_t a);
int tlfea_adamw_retrieve_velocity(tlfea_adamw_t a, double *v);
int tlfea_adamw_retrieve_lambda(tlfea_adamw_t a, double *lam);
/* out6: outer iterations, inner iterations (total), last ||g||, last ||c||, inner-converged flag, device ms */
int tlfea_adamw_get_stats(tlfea_adamw_t a, double *out6);
/* 1: semantics of SyncedAdamWSolver, the cooperative-kernel sibling of the same solver (SyncedAdamW.cuh, SyncedAdamW.cu:
 * 96-345): inner-converged flag cleared once per Solve(), lam += rho dt c applied once (Nocoop: twice), outer loop stops
 * on ||c|| < outer_tol alone.  0 (default): SyncedAdamWNocoopSolver. */
int tlfea_adamw_set_cooperative_semantics(tlfea_adamw_t a, int on);
int tlfea_adamw_set_verbose(tlfea_adamw_t a, int v);

/* ---- SyncedNesterovSolver (SyncedNesterov.cuh:26-260, SyncedNesterov.cu:95-372) ---------------------------------
 * Accelerated-gradient ALM solver on the velocities (the reference's cooperative kernel as ordinary launches of the
 * same residual path).  Fixed-coefficient constraints, single GPU. */
typedef struct tlfea_nesterov_s *tlfea_nesterov_t;
typedef struct { /* SyncedNesterovParams (SyncedNesterov.cuh:26-30) */
  double alpha, rho, inner_tol, outer_tol;
  int max_outer, max_inner;
  double time_step;
} tlfea_nesterov_params;
int tlfea_nesterov_create(tlfea_t10_t data, int n_constraints, tlfea_nesterov_t *out);
int tlfea_nesterov_destroy(tlfea_nesterov_t a);
int tlfea_nesterov_setup(tlfea_nesterov_t a);                                          /* Setup :140-156 */
int tlfea_nesterov_set_parameters(tlfea_nesterov_t a, const tlfea_nesterov_params *p); /* SetParameters :118-138 */
int tlfea_nesterov_solve(tlfea_nesterov_t a);                                          /* Solve()/OneStepNesterov */
double *tlfea_nesterov_velocity_guess_device_ptr(tlfea_nesterov_t a);
int tlfea_nesterov_retrieve_velocity(tlfea_nesterov_t a, double *v);
int tlfea_nesterov_retrieve_lambda(tlfea_nesterov_t a, double *lam);
/* out6: outer iterations run, inner iterations (total), last ||g||, last ||c||, inner-converged flag, device ms */
int tlfea_nesterov_get_stats(tlfea_nesterov_t a, double *out6);
int tlfea_nesterov_set_verbose(tlfea_nesterov_t a, int v);


/* ---- SyncedVBDSolver (lib_src/solvers/SyncedVBD.cuh:13-21 params, :23-330 class; SyncedVBD.cu:163-400 node update,
 * :764-1135 Initialize*, :1475-1641 OneStepVBD) -- vertex block descent: ALM outer loop, inner loop of coloured
 * Gauss-Seidel sweeps of per-node 3x3 Newton updates on the velocities.  Pinned-node constraints only (the reference's
 * fixed map).  Call order of the reference drivers: create, Setup, SetParameters, InitializeColoring,
 * InitializeMassDiagBlocks, InitializeFixedMap, Solve per step (Solve runs the Initialize* it still needs). */
typedef struct tlfea_vbd_s *tlfea_vbd_t;
typedef struct { /* == SyncedVBDParams, same field order */
  double inner_tol, inner_rtol, outer_tol, rho;
  int max_outer, max_inner;
  double time_step, omega, hess_eps;
  int convergence_check_interval, color_group_size;
} tlfea_vbd_params;
int tlfea_vbd_create(tlfea_t10_t data, int n_constraints, tlfea_vbd_t *out);
int tlfea_vbd_destroy(tlfea_vbd_t a);
int tlfea_vbd_setup(tlfea_vbd_t a);                                         /* Setup  SyncedVBD.cuh:263-274 */
int tlfea_vbd_set_parameters(tlfea_vbd_t a, const tlfea_vbd_params *p);     /* SetParameters :228-261 */
int tlfea_vbd_initialize_coloring(tlfea_vbd_t a);                           /* SyncedVBD.cu:764-1028 */
int tlfea_vbd_initialize_mass_diag_blocks(tlfea_vbd_t a);                   /* :1030-1085 (runs CalcMassMatrix) */
int tlfea_vbd_initialize_fixed_map(tlfea_vbd_t a);                          /* :1087-1135 */
int tlfea_vbd_solve(tlfea_vbd_t a);                                         /* Solve()/OneStepVBD :1475-1641 */
/* colouring: sizes {n_colors, n_groups}; arrays colors[N], color_offsets[n_colors+1], color_nodes[N],
 * group_offsets[n_groups+1], group_colors[n_colors] (any pointer may be null) */
int tlfea_vbd_coloring_sizes(tlfea_vbd_t a, int *n_colors, int *n_groups);
int tlfea_vbd_retrieve_coloring(tlfea_vbd_t a, int *colors, int *color_offsets, int *color_nodes, int *group_offsets,
                                int *group_colors);
double *tlfea_vbd_velocity_guess_device_ptr(tlfea_vbd_t a);
int tlfea_vbd_retrieve_velocity(tlfea_vbd_t a, double *v);
int tlfea_vbd_retrieve_lambda(tlfea_vbd_t a, double *lam);
/* out6: outer iterations run, sweeps (total), last checked ||g||, last ||c||, 0, device ms */
int tlfea_vbd_get_stats(tlfea_vbd_t a, double *out6);
int tlfea_vbd_set_verbose(tlfea_vbd_t a, int v);

#ifdef __cplusplus
}
#endif
#endif /* TLFEA_C_H */
