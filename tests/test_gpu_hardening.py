"""Failure paths and stale-state hazards of the solver layer (round-1 review): a linear solve that misses its tolerance
is an error, not a silent update; cached hipGraphs follow material / fixed-set changes; the solver's constraint count
cannot drift from the data object's."""
import importlib
import os

import numpy as np
import pytest

from tests.helpers import MATERIALS, fixed_x0, load_mesh, make_gpu

tl = importlib.import_module("total-lagrangian-fea_amd")
TlfeaError = importlib.import_module("total-lagrangian-fea_amd.binding").TlfeaError
pytestmark = pytest.mark.gpu


def beam(mesh="res2", mat="svk", fixed=None):
    X, conn = load_mesh(mesh)
    fixed = fixed_x0(X) if fixed is None else fixed
    f_ext = np.zeros(3 * X.shape[0])
    tip = int(np.argmax(X[:, 0] + 1e-3 * X[:, 1] + 1e-6 * X[:, 2]))
    f_ext[3 * tip] = 2000.0
    f_ext[3 * tip + 2] = -1000.0
    return X, fixed, make_gpu(X, conn, MATERIALS[mat], fixed, f_ext)


def newton(d, n_constraints=None, lin=None):
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint() if n_constraints is None else n_constraints)
    s.Setup()
    s.SetParameters(tl.SyncedNewtonParams(1e-8, 0.0, 1e-8, 1e14, 2, 4, 1e-3))
    s.SetLinSolveOpts(lin or tl.LinSolveOpts(1e-13, 20000, 10))
    return s


def test_unconverged_linear_solve_fails_the_step():
    """max_iter = 3 cannot reach 1e-13 on 466 nodes: Solve() must fail (the reference aborts when cuDSS fails) and leave
    the state untouched; with on_unconverged = 1 the iterate is accepted and the status says so."""
    X, _, d = beam()
    s = newton(d, lin=tl.LinSolveOpts(1e-13, 3, 1))
    x_before = np.stack(d.RetrievePositionToCPU(), axis=1)
    with pytest.raises(TlfeaError, match="did not converge"):
        s.Solve()
    st = s.GetLinSolveStatus()
    assert not st["converged"] and not st["all_converged"] and st["rel_res"] > 1e-13
    assert np.array_equal(np.stack(d.RetrievePositionToCPU(), axis=1), x_before)   # dv was not applied
    assert np.all(s.RetrieveVelocityToCPU() == 0.0)
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 3, 1, on_unconverged=1))
    s.Solve()
    st = s.GetLinSolveStatus()
    assert not st["all_converged"] and st["worst_rel_res"] > 1e-13
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 20000, 10))
    s.Solve()
    st = s.GetLinSolveStatus()
    assert st["all_converged"] and st["worst_rel_res"] <= 1e-13
    del s
    d.Destroy()


def test_solver_constraint_count_must_match_data():
    X, fixed, d = beam()
    with pytest.raises(TlfeaError, match="n_constraints"):
        tl.SyncedNewtonSolver(d, d.get_n_constraint() + 3)
    with pytest.raises(TlfeaError, match="n_constraints"):
        tl.SyncedVBDSolver(d, 3)
    # 0 switches the constraint terms off consistently (SyncedNewton.cu gates them on n_constraints_ > 0): the clamp
    # no longer holds the beam, and nothing indexes the (one-entry) multiplier buffer
    s = newton(d, n_constraints=0)
    s.Solve()
    x = np.stack(d.RetrievePositionToCPU(), axis=1)
    assert np.all(np.isfinite(x)) and np.max(np.abs(x[fixed] - X[fixed])) > 0.0
    assert s.GetStats()["norm_c"] == 0.0
    del s
    d.Destroy()


def test_fixed_set_resized_between_solves():
    """UpdateNodalFixed with another count between two Solve() calls: multipliers and norms follow the new size.  The
    second step equals that of an engine built with the new set and handed the same state."""
    X, fixed, d = beam()
    s = newton(d)
    s.Solve()
    x1 = np.stack(d.RetrievePositionToCPU(), axis=1)
    v1 = s.RetrieveVelocityToCPU()
    fixed2 = fixed[: max(3, len(fixed) // 2)]
    d.UpdateNodalFixed(fixed2)
    assert d.get_n_constraint() == 3 * len(fixed2) != 3 * len(fixed)
    s.Solve()
    xa = np.stack(d.RetrievePositionToCPU(), axis=1)
    assert len(s.RetrieveLambdaToCPU()) == 3 * len(fixed2)

    _, _, d2 = beam(fixed=fixed2)
    s2 = newton(d2)
    d2.UpdatePositions(x1[:, 0], x1[:, 1], x1[:, 2])
    s2.SetVelocity(v1, v1)
    s2.Solve()
    xb = np.stack(d2.RetrievePositionToCPU(), axis=1)
    disp = np.max(np.abs(xb - x1))
    assert disp > 0 and np.max(np.abs(xa - xb)) <= 1e-10 * disp + 8 * np.finfo(float).eps * np.max(np.abs(xb))
    del s, s2
    d.Destroy()
    d2.Destroy()


def vbd_run(graph):
    """two VBD steps with a material change and a resized fixed set in between; graph replay on / off"""
    old = os.environ.get("TLFEA_GRAPH")
    os.environ["TLFEA_GRAPH"] = "1" if graph else "0"
    try:
        X, fixed, d = beam("beam_3x2x1")
        s = tl.SyncedVBDSolver(d, d.get_n_constraint())
    finally:
        if old is None:
            os.environ.pop("TLFEA_GRAPH")
        else:
            os.environ["TLFEA_GRAPH"] = old
    s.Setup()
    s.SetParameters(tl.SyncedVBDParams(inner_tol=0.0, inner_rtol=0.0, outer_tol=0.0, rho=1e14, max_outer=2, max_inner=10,
                                       time_step=1e-3, omega=1.5, hess_eps=1e-12, convergence_check_interval=0,
                                       color_group_size=1))
    s.InitializeColoring()
    s.InitializeMassDiagBlocks()
    s.InitializeFixedMap()
    s.Solve()
    xs = [np.stack(d.RetrievePositionToCPU(), axis=1)]
    d.SetSVK(2.5e8, 0.3)                       # by-value material scalars of the captured launches
    d.SetDamping(0.0, 0.0)
    d.UpdateNodalFixed(fixed[: len(fixed) - 2])  # re-allocates the fixed-slot buffer the captured launches point at
    s.Solve()
    xs.append(np.stack(d.RetrievePositionToCPU(), axis=1))
    del s
    d.Destroy()
    return xs


def test_vbd_graph_follows_material_and_fixed_set():
    g, e = vbd_run(True), vbd_run(False)
    assert np.array_equal(g[0], e[0])
    assert np.array_equal(g[1], e[1])          # a stale graph would replay E = 7e8 on a freed fixed-slot buffer
    assert np.max(np.abs(g[1] - g[0])) > 0


def test_rereferenced_geometry_refreshes_the_affine_cache():
    """CalcDnDuPre after the solver analysed the sparsity (re-referencing, which the reference API permits): the affine
    assembly's cached vertex gradients / det J and its 'all elements straight-sided' decision must follow.  First a
    straight re-reference (scaled mesh: the affine form stays, with new gradients), then a curved one (one displaced
    mid-edge node: the general form takes over).  H equals the oracle's built on the same reference each time."""
    from tests.helpers import make_oracle, perturbed_state
    X, conn = load_mesh("res2")
    m = MATERIALS["svk"]
    fixed = fixed_x0(X)
    h, rho = 1e-3, 1e12
    d = make_gpu(X, conn, m, fixed)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-4, 1e-4, rho, 5, 10, h))
    s.AnalyzeHessianSparsity()
    s.AssembleHessian()
    assert s.GetAssemblyMode() == 3

    def check(Xref, mode):
        d.UpdatePositions(Xref[:, 0], Xref[:, 1], Xref[:, 2])
        d.CalcDnDuPre()
        d.CalcMassMatrix()
        o = make_oracle(Xref, conn, m, fixed)
        x, _ = perturbed_state(Xref)
        o.x, o.y, o.z = x[:, 0].copy(), x[:, 1].copy(), x[:, 2].copy()
        d.UpdatePositions(x[:, 0], x[:, 1], x[:, 2])
        s.AssembleHessian()
        assert s.GetAssemblyMode() == mode
        ro, ci, val = s.RetrieveHessianCSRToCPU()
        ro_o, ci_o, val_o = o.assemble_hessian(h, rho)
        assert np.array_equal(ro, ro_o) and np.array_equal(ci, ci_o)
        err = float(np.max(np.abs(val - val_o)) / np.max(np.abs(val_o)))
        assert err < 1e-12, (mode, err)

    check(X * np.array([1.3, 0.8, 1.1]), 3)
    Xc = X.copy()
    mid = int(conn[0, 4])
    Xc[mid] += 0.03 * np.linalg.norm(X[conn[0, 0]] - X[conn[0, 1]]) * np.array([0.3, -0.5, 0.8])
    check(Xc, 2)
    check(X, 3)                       # and back to the straight mesh: the affine form again
    del s
    d.Destroy()


def test_linear_solve_failing_in_a_later_iteration_rolls_the_step_back(monkeypatch):
    """A linear solve that fails in the SECOND Newton iteration of a step (forced through the TLFEA_TEST_FAIL_LINSOLVE test
    hook: v and x have already moved by then): the call fails, v / x / v_prev / lambda are those at the start of the step,
    the stats say what ran, and a retry gives the same step as a solver that never failed."""
    X, _, d = beam()
    s = newton(d)
    s.Solve()                                             # one good step: non-trivial v, lambda
    x0 = np.stack(d.RetrievePositionToCPU(), axis=1)
    v0, lam0 = s.RetrieveVelocityToCPU(), s.RetrieveLambdaToCPU()
    monkeypatch.setenv("TLFEA_TEST_FAIL_LINSOLVE", "1")
    with pytest.raises(TlfeaError, match="test hook"):
        s.Solve()
    st = s.GetStats()
    assert st["newton"] == 1 and st["pcg_iters"] > 0       # one iteration had been applied before the failing one
    assert np.array_equal(np.stack(d.RetrievePositionToCPU(), axis=1), x0)
    assert np.array_equal(s.RetrieveVelocityToCPU(), v0) and np.array_equal(s.RetrieveLambdaToCPU(), lam0)
    monkeypatch.delenv("TLFEA_TEST_FAIL_LINSOLVE")
    s.Solve()
    xa = np.stack(d.RetrievePositionToCPU(), axis=1)
    _, _, d2 = beam()
    s2 = newton(d2)
    s2.Solve()
    s2.Solve()
    xb = np.stack(d2.RetrievePositionToCPU(), axis=1)
    disp = np.max(np.abs(xb - x0))
    assert disp > 0 and np.max(np.abs(xa - xb)) <= 1e-10 * disp + 8 * np.finfo(float).eps * np.max(np.abs(xb))
    del s, s2
    d.Destroy()
    d2.Destroy()
