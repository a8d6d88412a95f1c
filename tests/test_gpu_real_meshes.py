"""SURVEY 8(d)'s real-mesh sanity set beyond what the parity files already use: the reference's TetGen meshes
beam_3x2x1_res8 (4 567 T10), beam_3x2x1_res16 (20 829 T10) and teapot.1 (12 280 T10) -- unstructured, with the node valences
and element shapes the structured generator never produces.  The oracle's Newton step takes minutes at these sizes (its
direct solve), so ONE Newton iteration is taken apart instead: gradient and Hessian against the oracle, the linear solve
against scipy's sparse solve of the ORACLE's H, the two device solvers (p-multigrid CG, multifrontal Cholesky) against
each other, and the update the engine applies against the one these pieces predict."""
import importlib

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from tests.helpers import MATERIALS, fixed_x0, load_mesh, make_gpu, make_oracle, relerr
from tests.test_gpu_parity import TOL_ELEM, perturbed_state, set_state

tl = importlib.import_module("total-lagrangian-fea_amd")
pytestmark = [pytest.mark.gpu]


def fixed_set(tag, X):
    if tag == "teapot":   # the teapot stands on its base: the lowest 5 % of its height is held
        return np.where(X[:, 2] < X[:, 2].min() + 0.05 * (X[:, 2].max() - X[:, 2].min()))[0].astype(np.int32)
    return fixed_x0(X)


@pytest.mark.parametrize("tag,mat", [("res8", "svk"), ("res8", "neo"), ("res16", "svk"), ("teapot", "svk"), ("teapot", "mr")])
def test_one_newton_iteration_taken_apart(tag, mat):
    X, conn = load_mesh(tag)
    m = MATERIALS[mat]
    fixed = fixed_set(tag, X)
    f_ext = np.zeros(3 * X.shape[0])
    tip = int(np.argmax(X[:, 0] + 1e-3 * X[:, 1] + 1e-6 * X[:, 2]))
    f_ext[3 * tip + 2] = -2.0e4
    o, d = make_oracle(X, conn, m, fixed, f_ext), make_gpu(X, conn, m, fixed, f_ext)
    # noise of 0.5 % of the shortest edge (the teapot's is 1.7e-3: an absolute 1e-4 strains its small elements by 10 % and
    # the St-Venant-Kirchhoff tangent -- the oracle's too -- stops being positive definite)
    edges = np.linalg.norm(X[conn[:, [0, 0, 0, 1, 1, 2]]] - X[conn[:, [1, 2, 3, 2, 3, 3]]], axis=2)
    x, v = perturbed_state(X, sigma=5e-3 * float(edges.min()))
    set_state(o, d, x)
    h, rho = 1e-3, 1e14
    prm = tl.SyncedNewtonParams(1e-4, 1e-4, 1e-4, rho, 5, 10, h)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.SetParameters(prm)
    s.SetVelocity(v, 0.5 * v)
    # gradient and Hessian of the Newton iteration against the oracle
    ng = s.EvalGradient()
    g = s.RetrieveGradientToCPU()
    o.v, o.v_prev = v.copy(), 0.5 * v
    g_o = o.grad_L(o.internal_force(v), h, rho)
    assert relerr(g, g_o) < TOL_ELEM and abs(ng - np.linalg.norm(g_o)) <= 1e-12 * np.linalg.norm(g_o)
    s.AnalyzeHessianSparsity()
    s.AssembleHessian()
    ro, ci, val = s.RetrieveHessianCSRToCPU()
    ro_o, ci_o, val_o = o.assemble_hessian(h, rho, nthreads=8)
    assert np.array_equal(ro, ro_o) and np.array_equal(ci, ci_o)      # indexing: bit-exact
    assert relerr(val, val_o) < TOL_ELEM
    s.AssembleHessian()
    assert np.array_equal(val, s.RetrieveHessianCSRToCPU()[2])        # no atomics: same bits again
    # the linear solve H dv = -g: scipy on the ORACLE's matrix is the reference for both device solvers
    n = 3 * X.shape[0]
    H = sp.csr_matrix((val_o, ci_o, ro_o), shape=(n, n)).tocsc()
    dv_ref = spla.spsolve(H, -g_o)
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 20000, 10))
    dv_it, iters, rel_it = s.LinearSolve(-g)
    assert 0 < iters < 120 and rel_it < 1e-12, (iters, rel_it)
    s.SetLinSolveOpts(tl.LinSolveOpts(method=1))
    dv_dir, it_d, rel_d = s.LinearSolve(-g)
    assert it_d == 1 and rel_d < 1e-10
    nrm = np.linalg.norm(dv_ref)
    assert np.linalg.norm(dv_it - dv_ref) <= 1e-7 * nrm and np.linalg.norm(dv_dir - dv_ref) <= 1e-7 * nrm   # cond(H) ~ 1e8
    assert np.linalg.norm(dv_dir - dv_it) <= 1e-8 * nrm
    assert np.linalg.norm(H @ dv_dir + g_o) <= 1e-9 * np.linalg.norm(g_o)
    del s
    d.Destroy()


def test_newton_step_descends_on_the_largest_real_mesh():
    """beam_3x2x1_res16 through the solver itself: a loaded step converges in the reference's iteration budget, every linear
    solve meets its tolerance, the constraint violation stays at round-off and the loaded end moves the way of the load."""
    X, conn = load_mesh("res16")
    fixed = fixed_x0(X)
    f_ext = np.zeros(3 * X.shape[0])
    face = np.where(X[:, 0] > X[:, 0].max() - 1e-9)[0]
    f_ext[3 * face + 2] = -5000.0 / len(face)                      # test_feat10_resolution.cc:298-312: 5000 N over the end face
    d = make_gpu(X, conn, MATERIALS["svk"], fixed, f_ext)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.Setup()
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-4, 1e-4, 1e14, 5, 10, 1e-3))   # test_feat10_resolution.cc:365
    for _ in range(3):
        s.Solve()
        ls = s.GetLinSolveStatus()
        assert ls["all_converged"] and ls["worst_rel_res"] < 1e-11, ls
    xg = np.stack(d.RetrievePositionToCPU(), axis=1)
    assert np.all(np.isfinite(xg)) and np.abs(xg[fixed] - X[fixed]).max() < 1e-9
    assert (xg[face, 2] - X[face, 2]).mean() < -1e-7
    assert s.GetPreconditioner() == 2 and s.GetStats()["newton"] >= 1
    del s
    d.Destroy()
