"""SyncedNewtonSolver -- host mirror of lib_src/solvers/SyncedNewton.cuh:35-405 on the C-ABI."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from .binding import (ALLREDUCE_FN, AdamWParamsC, LinSolveOptsC, NesterovParamsC, NewtonParams, VbdParamsC, check, dp, ip,
                      load_library)


@dataclass
class SyncedNewtonParams:  # SyncedNewton.cuh:29-33 (same field order)
    inner_atol: float = 1e-4
    inner_rtol: float = 1e-4
    outer_tol: float = 1e-4
    rho: float = 1e14
    max_outer: int = 5
    max_inner: int = 10
    time_step: float = 1e-3


@dataclass
class LinSolveOpts:
    rel_tol: float = 1e-12
    max_iter: int = 20000
    check_every: int = 25
    cheb_degree: int = 0      # Chebyshev polynomial preconditioner degree (1 = block-Jacobi, 0 = auto = 24)
    cheb_kappa: float = 0.0   # polynomial interval [lmax/kappa, lmax] of D^-1 H (0 = auto, by degree)
    cheb_bits: int = 0        # matrix precision of the polynomial's steps: 64 | 32 | 16 (0 = auto = 16)
    precond: int = 0          # 0 auto | 1 Chebyshev polynomial | 2 two-level p-multigrid (T10, one GPU)
    method: int = 0           # 0 preconditioned CG | 1 sparse direct (where built)
    on_unconverged: int = 0   # 0: a solve that misses rel_tol fails the call (nothing applied) | 1: accept the iterate


class SyncedNewtonSolver:
    def __init__(self, data, n_constraints):
        self._lib = load_library()
        self._data = data  # the data object must outlive the solver (SyncedNewton.cuh:37-46)
        self.n_coef = data.get_n_coef()
        self.n_constraints = int(n_constraints)
        self._h = C.c_void_p()
        check(self._lib.tlfea_newton_create(data._h, self.n_constraints, C.byref(self._h)))
        self._cb = None

    def __del__(self):
        try:
            if self._h:
                self._lib.tlfea_newton_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def Setup(self):
        check(self._lib.tlfea_newton_setup(self._h))

    def SetParameters(self, params):
        p = NewtonParams(params.inner_atol, params.inner_rtol, params.outer_tol, params.rho, params.max_outer,
                         params.max_inner, params.time_step)
        check(self._lib.tlfea_newton_set_parameters(self._h, C.byref(p)))

    def SetLinSolveOpts(self, o):
        c = LinSolveOptsC(o.rel_tol, o.max_iter, o.check_every, o.cheb_degree, o.cheb_kappa, o.cheb_bits, o.precond,
                          o.method, o.on_unconverged)
        check(self._lib.tlfea_newton_set_linsolve_opts(self._h, C.byref(c)))

    def GetLinSolveInfo(self):
        """(polynomial degree, matrix bits of its steps, bits of its work vectors) the auto rules resolved to"""
        deg, bits, vbits = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.tlfea_newton_get_linsolve_info(self._h, C.byref(deg), C.byref(bits), C.byref(vbits)))
        return deg.value, bits.value, vbits.value

    def GetPreconditioner(self):
        """0 block-Jacobi, 1 Chebyshev polynomial, 2 two-level p-multigrid (what a solve would use now)"""
        return int(self._lib.tlfea_newton_get_precond(self._h))

    def Collectives(self):
        """all-reduce calls issued by the engine since the solver was built (torch.distributed callback or built-in RCCL)"""
        self._lib.tlfea_newton_collectives.restype = C.c_long
        return int(self._lib.tlfea_newton_collectives(self._h))

    def GetAssemblyMode(self):
        """1: tangent blocks + row-owner gather (two launches), 2: fused row-owner tangent + assembly (T10, SVK)"""
        return int(self._lib.tlfea_newton_get_assembly_mode(self._h))

    def RetrievePmgLevel(self):
        """(par0, par1, c_off, c_cols, Hc_values): parent map of the fine nodes, coarse node adjacency and the Galerkin
        operator P^T H P of the current H in the DOF-level layout of H (test hook)."""
        nc, nnz = C.c_int(), C.c_int()
        check(self._lib.tlfea_newton_pmg_sizes(self._h, C.byref(nc), C.byref(nnz)))
        par0, par1 = np.zeros(self.n_coef, dtype=np.int32), np.zeros(self.n_coef, dtype=np.int32)
        c_off, c_cols = np.zeros(nc.value + 1, dtype=np.int32), np.zeros(nnz.value, dtype=np.int32)
        Hc = np.zeros(9 * nnz.value)
        check(self._lib.tlfea_newton_pmg_retrieve(self._h, ip(par0), ip(par1), ip(c_off), ip(c_cols), dp(Hc)))
        return par0, par1, c_off, c_cols, Hc

    def GetPmgCycleInfo(self):
        """dict(levels, fine_terms, vertex_terms, vertex_degree, level3_degree, level3_nodes) of the cycle in use (levels 0:
        polynomial preconditioner)"""
        out = (C.c_int * 6)()
        check(self._lib.tlfea_newton_pmg_cycle_info(self._h, out))
        return dict(zip(("levels", "fine_terms", "vertex_terms", "vertex_degree", "level3_degree", "level3_nodes"), list(out)))

    def GetPolynomialInfo(self):
        """dict(degree, kappa, block) of the polynomial preconditioner in use; block 12 = ANCF node-block scaling"""
        out = (C.c_int * 3)()
        check(self._lib.tlfea_newton_polynomial_info(self._h, out))
        return dict(zip(("degree", "kappa", "block"), list(out)))

    def GetPmgLevel3Info(self):
        """(aggregates, 3x3 blocks, polynomial degree) of the third level, (0, 0, 0) when the cycle has two levels"""
        na, nnz, deg = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.tlfea_newton_pmg3_sizes(self._h, C.byref(na), C.byref(nnz), C.byref(deg)))
        return na.value, nnz.value, deg.value

    def RetrievePmgLevel3(self):
        """(agg[Nc], rvec[Nc,3], active[Na], off3, cols3, H3_values): test hook for H3 = P2^T Hc P2"""
        nc, nnz = C.c_int(), C.c_int()
        check(self._lib.tlfea_newton_pmg_sizes(self._h, C.byref(nc), C.byref(nnz)))
        na, nnz3, _ = self.GetPmgLevel3Info()
        agg, rvec, active = np.zeros(nc.value, dtype=np.int32), np.zeros(3 * nc.value), np.zeros(na, dtype=np.int32)
        off3, cols3, H3 = np.zeros(2 * na + 1, dtype=np.int32), np.zeros(nnz3, dtype=np.int32), np.zeros(9 * nnz3)
        check(self._lib.tlfea_newton_pmg3_retrieve(self._h, ip(agg), dp(rvec), ip(active), ip(off3), ip(cols3), dp(H3)))
        return agg, rvec.reshape(-1, 3), active, off3, cols3, H3

    def AnalyzeHessianSparsity(self):
        check(self._lib.tlfea_newton_analyze_hessian_sparsity(self._h))

    def SetFixedSparsityPattern(self, fixed):
        check(self._lib.tlfea_newton_set_fixed_sparsity_pattern(self._h, int(bool(fixed))))

    def Solve(self):
        check(self._lib.tlfea_newton_solve(self._h))

    OneStepNewtonCuDSS = Solve  # reference name of the same step (SyncedNewton.cuh:345-349)

    def GetVelocityGuessDevicePtr(self):
        return self._lib.tlfea_newton_velocity_guess_device_ptr(self._h)

    def compute_l2_norm_cublas(self, d_vec, n_dofs):
        out = C.c_double()
        check(self._lib.tlfea_newton_l2_norm(self._h, C.c_void_p(d_vec), int(n_dofs), C.byref(out)))
        return out.value

    # ---- engine extras ------------------------------------------------------------------------
    def SetVerbose(self, v):
        check(self._lib.tlfea_newton_set_verbose(self._h, int(v)))

    def SetProfiling(self, on):
        check(self._lib.tlfea_newton_set_profiling(self._h, int(bool(on))))

    def RetrieveHessianCSRToCPU(self):
        nnz = C.c_int()
        check(self._lib.tlfea_newton_hessian_nnz(self._h, C.byref(nnz)))
        ro = np.zeros(3 * self.n_coef + 1, dtype=np.int32)
        ci = np.zeros(nnz.value, dtype=np.int32)
        val = np.zeros(nnz.value)
        check(self._lib.tlfea_newton_retrieve_hessian_csr(self._h, ip(ro), ip(ci), dp(val)))
        return ro, ci, val

    def EvalGradient(self):
        ng = C.c_double()
        check(self._lib.tlfea_newton_eval_gradient(self._h, C.byref(ng)))
        return ng.value

    def AssembleHessian(self):
        check(self._lib.tlfea_newton_assemble_hessian(self._h))

    def LinearSolve(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        it, rel = C.c_int(), C.c_double()
        check(self._lib.tlfea_newton_linear_solve(self._h, dp(b), dp(x), C.byref(it), C.byref(rel)))
        return x, it.value, rel.value

    def ApplyHessian(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        check(self._lib.tlfea_newton_apply_hessian(self._h, dp(x), dp(y)))
        return y

    def NewtonIteration(self):
        ng, it = C.c_double(), C.c_int()
        check(self._lib.tlfea_newton_iteration(self._h, C.byref(ng), C.byref(it)))
        return ng.value, it.value

    def RetrieveGradientToCPU(self):
        g = np.zeros(3 * self.n_coef)
        check(self._lib.tlfea_newton_retrieve_gradient(self._h, dp(g)))
        return g

    def RetrieveVelocityToCPU(self):
        v = np.zeros(3 * self.n_coef)
        check(self._lib.tlfea_newton_retrieve_velocity(self._h, dp(v)))
        return v

    def SetVelocity(self, v, v_prev=None):
        v = np.ascontiguousarray(v, dtype=np.float64)
        vp = np.ascontiguousarray(v_prev, dtype=np.float64) if v_prev is not None else None
        check(self._lib.tlfea_newton_set_velocity(self._h, dp(v), dp(vp)))

    def SetLambda(self, lam):
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        assert lam.size == self.n_constraints
        check(self._lib.tlfea_newton_set_lambda(self._h, dp(lam)))

    def RetrieveLambdaToCPU(self):
        self.n_constraints = int(self._lib.tlfea_newton_n_constraints(self._h))   # follows UpdateNodalFixed
        lam = np.zeros(self.n_constraints)
        check(self._lib.tlfea_newton_retrieve_lambda(self._h, dp(lam)))
        return lam

    def GetStats(self):
        st = np.zeros(6)
        check(self._lib.tlfea_newton_get_stats(self._h, dp(st)))
        return dict(outer=int(st[0]), newton=int(st[1]), norm_g=st[2], norm_c=st[3], pcg_iters=int(st[4]), ms=st[5])

    def GetLinSolveStatus(self):
        """Linear solves since the last Solve() / NewtonIteration() began: relative residual and tolerance flag of the
        last one, worst relative residual, and whether all of them met rel_tol."""
        st = np.zeros(4)
        check(self._lib.tlfea_newton_get_linsolve_status(self._h, dp(st)))
        return dict(rel_res=st[0], converged=bool(st[1]), worst_rel_res=st[2], all_converged=bool(st[3]))

    STAGES = ["residual", "grad", "tangent_blocks", "assemble_rows", "pcg", "update", "spmv", "_"]

    def GetStageMs(self, reset=False):
        """-> {stage: (total_ms, launches)} measured with hipEvents on the launch stream (profiling mode)."""
        ms, cnt = np.zeros(8), np.zeros(8)
        check(self._lib.tlfea_newton_get_stage_ms(self._h, dp(ms), dp(cnt), int(reset)))
        return {k: (ms[i], int(cnt[i])) for i, k in enumerate(self.STAGES) if k != "_"}

    def TimeKernels(self, reps=20):
        """-> {kernel: mean ms} over `reps` back-to-back launches each (hipEvents on the launch stream)."""
        out = np.zeros(7)
        check(self._lib.tlfea_newton_time_kernels(self._h, int(reps), dp(out)))
        return dict(zip(["residual", "tangent_blocks", "assemble_rows", "spmv", "cheb_step", "cheb_step_coarse",
                         "cheb_step_cycle"], out.tolist()))

    def GetPmgInfo(self):
        """(coarse nodes, coarse 3x3 blocks, coarse polynomial degree) of the p-multigrid level, or None"""
        if self.GetPreconditioner() != 2:
            return None
        nc, nnz = C.c_int(), C.c_int()
        check(self._lib.tlfea_newton_pmg_sizes(self._h, C.byref(nc), C.byref(nnz)))
        return nc.value, nnz.value, int(self._lib.tlfea_newton_pmg_coarse_degree(self._h))

    def BeginStep(self):
        check(self._lib.tlfea_newton_begin_step(self._h))

    def SetInterface(self, iface_nodes, iface_slots, n_global_iface, node_weight, allreduce, sync_before_callback=1):
        """allreduce(ptr:int, n:int) sums a device buffer of n doubles over ranks in place (see partition.py)."""
        nodes = np.ascontiguousarray(iface_nodes, dtype=np.int32)
        slots = np.ascontiguousarray(iface_slots, dtype=np.int32)
        w = np.ascontiguousarray(node_weight, dtype=np.float64)
        assert w.size == self.n_coef and nodes.size == slots.size

        def _cb(_user, ptr, n):
            try:
                allreduce(ptr, n)
                return 0
            except Exception as exc:  # pragma: no cover
                print("allreduce callback failed:", repr(exc), flush=True)
                return 1

        self.n_collectives = 0

        def _cb_counted(user, ptr, n):
            self.n_collectives += 1
            return _cb(user, ptr, n)

        self._cb = ALLREDUCE_FN(_cb_counted)
        check(self._lib.tlfea_newton_set_interface(self._h, ip(nodes), ip(slots), int(nodes.size),
                                                   int(n_global_iface), dp(w), self._cb, None,
                                                   int(sync_before_callback)))

    def SetInterfaceRccl(self, iface_nodes, iface_slots, n_global_iface, node_weight, comm):
        """The same interface with the library's built-in RCCL all-reduce (tlfea_rccl_allreduce_fn): the collective is
        enqueued from C++ on the solver's launch stream, no Python in the loop.  comm: handle from
        partition.rccl_communicator()."""
        nodes = np.ascontiguousarray(iface_nodes, dtype=np.int32)
        slots = np.ascontiguousarray(iface_slots, dtype=np.int32)
        w = np.ascontiguousarray(node_weight, dtype=np.float64)
        assert w.size == self.n_coef and nodes.size == slots.size
        self._lib.tlfea_rccl_allreduce_fn.restype = C.c_void_p
        fn = C.cast(self._lib.tlfea_rccl_allreduce_fn(), ALLREDUCE_FN)
        self.n_collectives = -1   # not counted on this path
        self._cb = fn
        check(self._lib.tlfea_newton_set_interface(self._h, ip(nodes), ip(slots), int(nodes.size),
                                                   int(n_global_iface), dp(w), fn, comm, 0))

    def SetHalo(self, part, allreduce=None, exchange=None, sync_before_callback=1, rccl_comm=None):
        """Overlapping partition (tlfea_newton_set_halo): `part` is a partition.HaloPartition built on the mesh this
        solver's data object holds.  Either Python callbacks -- allreduce(ptr, n) sums n device doubles over ranks in
        place, exchange(send_ptr, recv_ptr, peers, send_off, recv_off) moves byte ranges neighbour to neighbour -- or
        rccl_comm: the library's built-in RCCL exchange on a communicator of partition.rccl_communicator()."""
        from .binding import HALO_EXCHANGE_FN, HaloListsC
        assert len(part.layer) == self.n_coef
        P = len(part.peers)
        peers = np.ascontiguousarray(part.peers, dtype=np.int32)
        cat = lambda arrs: (np.ascontiguousarray(np.concatenate(arrs), dtype=np.int32) if arrs else np.zeros(0, np.int32))  # noqa: E731
        off = lambda arrs: np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(a) for a in arrs])]), dtype=np.int32)  # noqa: E731
        so, ro = off(part.send), off(part.recv)
        sn, sl, rn = cat(part.send), cat(part.send_layer), cat(part.recv)
        layer = np.ascontiguousarray(part.layer, dtype=np.int32)
        lists = HaloListsC(P, ip(peers), ip(so), ip(sn), ip(sl), ip(ro), ip(rn), int(part.rank), int(part.world))
        self._halo_keep = (peers, so, ro, sn, sl, rn, layer, lists)
        if rccl_comm is not None:
            self._lib.tlfea_rccl_allreduce_fn.restype = C.c_void_p
            self._lib.tlfea_rccl_halo_exchange_fn.restype = C.c_void_p
            ar = C.cast(self._lib.tlfea_rccl_allreduce_fn(), ALLREDUCE_FN)
            ex = C.cast(self._lib.tlfea_rccl_halo_exchange_fn(), HALO_EXCHANGE_FN)
            self._cb = (ar, ex)
            check(self._lib.tlfea_newton_set_halo(self._h, ip(layer), int(part.depth), C.byref(lists), ar, ex, rccl_comm, 0))
            return

        def _ar(_user, ptr, n):
            try:
                allreduce(ptr, n)
                return 0
            except Exception as exc:  # pragma: no cover
                print("allreduce callback failed:", repr(exc), flush=True)
                return 1

        def _ex(_user, sp, rp, n_peers, peers_p, so_p, ro_p):
            try:
                exchange(sp, rp, [peers_p[k] for k in range(n_peers)], [so_p[k] for k in range(n_peers + 1)],
                         [ro_p[k] for k in range(n_peers + 1)])
                return 0
            except Exception as exc:  # pragma: no cover
                print("halo exchange callback failed:", repr(exc), flush=True)
                return 1

        self._cb = (ALLREDUCE_FN(_ar), HALO_EXCHANGE_FN(_ex))
        check(self._lib.tlfea_newton_set_halo(self._h, ip(layer), int(part.depth), C.byref(lists), self._cb[0], self._cb[1],
                                              None, int(sync_before_callback)))

    def GetCommStats(self):
        out = np.zeros(8)
        check(self._lib.tlfea_newton_get_comm_stats(self._h, dp(out)))
        return dict(exchanges=int(out[0]), allreduces=int(out[1]), bytes_exchanged=float(out[2]), bytes_allreduced=float(out[3]),
                    comm_ms=float(out[4]), cg_iterations=int(out[5]), exchanges_in_cg=int(out[6]), allreduces_in_cg=int(out[7]))

    def SetInterfaceOwners(self, node_owned):
        """node_owned[n_coef] = 1 where this rank owns the node (one owner per replicated node over all ranks):
        switches the polynomial preconditioner to its rank-local form (see tlfea_c.h)."""
        own = np.ascontiguousarray(node_owned, dtype=np.int32)
        assert own.size == self.n_coef
        check(self._lib.tlfea_newton_set_interface_owners(self._h, ip(own)))


@dataclass
class SyncedAdamWNocoopParams:
    """SyncedAdamWParams, field order of SyncedAdamW.cuh:27-34 (drivers: test_ancf3243.cc:374-376)."""
    lr: float = 2e-4
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 1e-4
    lr_decay: float = 0.998
    inner_tol: float = 1e-1
    outer_tol: float = 1e-6
    rho: float = 1e14
    max_outer: int = 5
    max_inner: int = 500
    time_step: float = 1e-3
    convergence_check_interval: int = 10
    inner_rtol: float = 0.0


class SyncedAdamWNocoopSolver:
    """SyncedAdamWNocoopSolver (SyncedAdamWNocoop.cuh:22-198): first-order ALM solver on the element kernels."""

    def __init__(self, data, n_constraints):
        self._lib = load_library()
        self._data = data
        self.n_coef = data.get_n_coef()
        self.n_constraints = int(n_constraints)
        self._h = C.c_void_p()
        check(self._lib.tlfea_adamw_create(data._h, self.n_constraints, C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                self._lib.tlfea_adamw_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def Setup(self):
        check(self._lib.tlfea_adamw_setup(self._h))

    def SetParameters(self, p):
        c = AdamWParamsC(p.lr, p.beta1, p.beta2, p.eps, p.weight_decay, p.lr_decay, p.inner_tol, p.outer_tol, p.rho,
                         p.max_outer, p.max_inner, p.time_step, p.convergence_check_interval, p.inner_rtol)
        check(self._lib.tlfea_adamw_set_parameters(self._h, C.byref(c)))

    def Solve(self):
        check(self._lib.tlfea_adamw_solve(self._h))

    OneStepAdamWNocoop = Solve

    def SetVerbose(self, v):
        check(self._lib.tlfea_adamw_set_verbose(self._h, int(v)))

    def GetVelocityGuessDevicePtr(self):
        self._lib.tlfea_adamw_velocity_guess_device_ptr.restype = C.c_void_p
        return self._lib.tlfea_adamw_velocity_guess_device_ptr(self._h)

    def RetrieveVelocityToCPU(self):
        v = np.zeros(3 * self.n_coef)
        check(self._lib.tlfea_adamw_retrieve_velocity(self._h, dp(v)))
        return v

    def RetrieveLambdaToCPU(self):
        lam = np.zeros(self.n_constraints)
        check(self._lib.tlfea_adamw_retrieve_lambda(self._h, dp(lam)))
        return lam

    def GetStats(self):
        st = np.zeros(6)
        check(self._lib.tlfea_adamw_get_stats(self._h, dp(st)))
        return dict(outer=int(st[0]), inner=int(st[1]), norm_g=st[2], norm_c=st[3], inner_flag=int(st[4]), ms=st[5])


class SyncedAdamWSolver(SyncedAdamWNocoopSolver):
    """SyncedAdamWSolver (SyncedAdamW.cuh, SyncedAdamW.cu:96-445): the cooperative-kernel sibling of the same solver; as
    written there the inner-converged flag is cleared once per Solve(), the multipliers get rho*dt*c once and the outer
    loop stops on ||c|| < outer_tol alone."""

    def __init__(self, data, n_constraints):
        super().__init__(data, n_constraints)
        check(self._lib.tlfea_adamw_set_cooperative_semantics(self._h, 1))

    OneStepAdamW = SyncedAdamWNocoopSolver.Solve


SyncedAdamWParams = SyncedAdamWNocoopParams


@dataclass
class SyncedNesterovParams:
    """SyncedNesterovParams (SyncedNesterov.cuh:26-30; driver values test_ancf3243.cc:351-352)."""
    alpha: float = 1e-8
    rho: float = 1e14
    inner_tol: float = 1e-6
    outer_tol: float = 1e-6
    max_outer: int = 5
    max_inner: int = 200
    time_step: float = 1e-3


class SyncedNesterovSolver:
    """SyncedNesterovSolver (SyncedNesterov.cuh:32-260): accelerated-gradient ALM solver on the element kernels."""

    def __init__(self, data, n_constraints):
        self._lib = load_library()
        self._data = data
        self.n_coef = data.get_n_coef()
        self.n_constraints = int(n_constraints)
        self._h = C.c_void_p()
        check(self._lib.tlfea_nesterov_create(data._h, self.n_constraints, C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                self._lib.tlfea_nesterov_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def Setup(self):
        check(self._lib.tlfea_nesterov_setup(self._h))

    def SetParameters(self, p):
        c = NesterovParamsC(p.alpha, p.rho, p.inner_tol, p.outer_tol, p.max_outer, p.max_inner, p.time_step)
        check(self._lib.tlfea_nesterov_set_parameters(self._h, C.byref(c)))

    def Solve(self):
        check(self._lib.tlfea_nesterov_solve(self._h))

    OneStepNesterov = Solve

    def SetVerbose(self, v):
        check(self._lib.tlfea_nesterov_set_verbose(self._h, int(v)))

    def GetVelocityGuessDevicePtr(self):
        return self._lib.tlfea_nesterov_velocity_guess_device_ptr(self._h)

    def RetrieveVelocityToCPU(self):
        v = np.zeros(3 * self.n_coef)
        check(self._lib.tlfea_nesterov_retrieve_velocity(self._h, dp(v)))
        return v

    def RetrieveLambdaToCPU(self):
        lam = np.zeros(self.n_constraints)
        check(self._lib.tlfea_nesterov_retrieve_lambda(self._h, dp(lam)))
        return lam

    def GetStats(self):
        st = np.zeros(6)
        check(self._lib.tlfea_nesterov_get_stats(self._h, dp(st)))
        return dict(outer=int(st[0]), inner=int(st[1]), norm_g=st[2], norm_c=st[3], inner_flag=int(st[4]), ms=st[5])


@dataclass
class SyncedVBDParams:
    """SyncedVBDParams (SyncedVBD.cuh:13-21; driver values test_feat10_resolution.cc:379-380 with omega 1.8)."""
    inner_tol: float = 1e-4
    inner_rtol: float = 1e-4
    outer_tol: float = 1e-4
    rho: float = 1e14
    max_outer: int = 5
    max_inner: int = 500
    time_step: float = 1e-3
    omega: float = 1.0
    hess_eps: float = 1e-12
    convergence_check_interval: int = 25
    color_group_size: int = 1


class SyncedVBDSolver:
    """SyncedVBDSolver (SyncedVBD.cuh:23-330): vertex block descent -- coloured per-node 3x3 Newton sweeps."""

    def __init__(self, data, n_constraints):
        self._lib = load_library()
        self._data = data
        self.n_coef = data.get_n_coef()
        self.n_constraints = int(n_constraints)
        self._h = C.c_void_p()
        check(self._lib.tlfea_vbd_create(data._h, self.n_constraints, C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                self._lib.tlfea_vbd_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def Setup(self):
        check(self._lib.tlfea_vbd_setup(self._h))

    def SetParameters(self, p):
        c = VbdParamsC(p.inner_tol, p.inner_rtol, p.outer_tol, p.rho, p.max_outer, p.max_inner, p.time_step, p.omega,
                       p.hess_eps, p.convergence_check_interval, p.color_group_size)
        check(self._lib.tlfea_vbd_set_parameters(self._h, C.byref(c)))

    def InitializeColoring(self):
        check(self._lib.tlfea_vbd_initialize_coloring(self._h))

    def InitializeMassDiagBlocks(self):
        check(self._lib.tlfea_vbd_initialize_mass_diag_blocks(self._h))

    def InitializeFixedMap(self):
        check(self._lib.tlfea_vbd_initialize_fixed_map(self._h))

    def Solve(self):
        check(self._lib.tlfea_vbd_solve(self._h))

    OneStepVBD = Solve

    def SetVerbose(self, v):
        check(self._lib.tlfea_vbd_set_verbose(self._h, int(v)))

    def GetColoring(self):
        """colors[N], color_offsets, color_nodes, group_offsets, group_colors of InitializeColoring"""
        nc, ng = C.c_int(), C.c_int()
        check(self._lib.tlfea_vbd_coloring_sizes(self._h, C.byref(nc), C.byref(ng)))
        out = dict(n_colors=nc.value, n_groups=ng.value, colors=np.zeros(self.n_coef, dtype=np.int32),
                   color_offsets=np.zeros(nc.value + 1, dtype=np.int32), color_nodes=np.zeros(self.n_coef, dtype=np.int32),
                   group_offsets=np.zeros(ng.value + 1, dtype=np.int32), group_colors=np.zeros(nc.value, dtype=np.int32))
        check(self._lib.tlfea_vbd_retrieve_coloring(self._h, ip(out["colors"]), ip(out["color_offsets"]),
                                                    ip(out["color_nodes"]), ip(out["group_offsets"]), ip(out["group_colors"])))
        return out

    def GetVelocityGuessDevicePtr(self):
        return self._lib.tlfea_vbd_velocity_guess_device_ptr(self._h)

    def RetrieveVelocityToCPU(self):
        v = np.zeros(3 * self.n_coef)
        check(self._lib.tlfea_vbd_retrieve_velocity(self._h, dp(v)))
        return v

    def RetrieveLambdaToCPU(self):
        lam = np.zeros(self.n_constraints)
        check(self._lib.tlfea_vbd_retrieve_lambda(self._h, dp(lam)))
        return lam

    def GetStats(self):
        st = np.zeros(6)
        check(self._lib.tlfea_vbd_get_stats(self._h, dp(st)))
        return dict(outer=int(st[0]), sweeps=int(st[1]), norm_g=st[2], norm_c=st[3], ms=st[5])
