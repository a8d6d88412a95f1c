"""Sparse direct linear solve (tlfea_linsolve_opts.method = 1): the engine's own multifrontal Cholesky (csrc/mf_host.h plan,
csrc/direct_kernels.hip) -- the counterpart of the reference's cuDSS analysis-once / refactor-per-iteration
(SyncedNewton.cu:995-1029, 1103-1114).  Checked against the oracle's direct solve and against the iterative path.
The rocSOLVER backend (TLFEA_DIRECT_BACKEND=rocsolver) stays available and is tested in its own process only."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import orc
from tests.helpers import MATERIALS, fixed_x0, load_mesh, make_gpu, make_oracle
from tests.test_gpu_parity import disp_err_ok

tl = importlib.import_module("total-lagrangian-fea_amd")
pytestmark = [pytest.mark.gpu]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pair(mesh, mat):
    X, conn = load_mesh(mesh)
    fixed = fixed_x0(X)
    f_ext = np.zeros(3 * X.shape[0])
    tip = int(np.argmax(X[:, 0] + 1e-3 * X[:, 1] + 1e-6 * X[:, 2]))
    f_ext[3 * tip] = 2.0e5          # moves the tip by ~1e-3: the 1e-10 bar is a relative one here, not the ulp floor
    f_ext[3 * tip + 2] = -1.0e5
    m = MATERIALS[mat]
    return X, make_oracle(X, conn, m, fixed, f_ext), make_gpu(X, conn, m, fixed, f_ext)


@pytest.mark.parametrize("mesh,mat", [("beam_3x2x1", "svk"), ("res2", "svk"), ("res2", "neo")])
def test_direct_newton_steps_match_oracle(mesh, mat):
    X, o, d = pair(mesh, mat)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.Setup()
    prm = (1e-6, 0.0, 1e-6, 1e14, 5, 10, 1e-3)
    s.SetParameters(tl.SyncedNewtonParams(*prm))
    s.SetLinSolveOpts(tl.LinSolveOpts(method=1))
    for step in range(2):
        s.Solve()
        st_o = o.newton_step(orc.NewtonParams(*prm), solver=0)
        st = s.GetStats()
        xg = np.stack(d.RetrievePositionToCPU(), axis=1)
        xo = np.stack([o.x, o.y, o.z], axis=1)
        assert (st["outer"], st["newton"]) == (int(st_o[0]), int(st_o[1])), (st, st_o)
        assert st["pcg_iters"] == st["newton"]                  # one factor + solve per Newton iteration
        assert disp_err_ok(xg, xo, X), (mesh, mat, step, np.max(np.abs(xg - xo)), np.max(np.abs(xo - X)))
        ls = s.GetLinSolveStatus()
        assert ls["all_converged"] and ls["worst_rel_res"] < 1e-9, ls
    assert np.max(np.abs(xo - X)) > 1e-4
    del s
    d.Destroy()


@pytest.mark.parametrize("mesh", ["box", "bunny"])
def test_direct_and_iterative_solutions_agree(mesh):
    """One right-hand side, both methods on the same assembled H (config-B-like cube of 2 592 elements; the TetGen bunny)."""
    if mesh == "box":
        X, conn = tl.mesh_utils.structured_t10_box(6, 6, 12)
        fixed = np.where(X[:, 2] < 1e-12)[0].astype(np.int32)
    else:
        X, conn = load_mesh("bunny")
        fixed = np.where(X[:, 2] < X[:, 2].min() + 0.3)[0].astype(np.int32)
    d = make_gpu(X, conn, MATERIALS["neo"], fixed)
    x = X + 1e-2 * np.sin(np.pi * X)
    d.UpdatePositions(x[:, 0], x[:, 1], x[:, 2])
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-6, 1e-4, 1e14, 5, 10, 1e-3))
    s.AssembleHessian()
    b = np.random.default_rng(3).normal(size=3 * X.shape[0])
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 20000, 10))
    xi, it_i, rel_i = s.LinearSolve(b)
    s.SetLinSolveOpts(tl.LinSolveOpts(method=1))
    xd, it_d, rel_d = s.LinearSolve(b)
    Hxd = s.ApplyHessian(xd)
    assert it_d == 1 and rel_d < 1e-9 and np.linalg.norm(Hxd - b) / np.linalg.norm(b) < 1e-9
    assert np.linalg.norm(xd - xi) <= 1e-7 * np.linalg.norm(xi)      # cond(H) ~ 1e14 * h^2 rho scaling: compare loosely
    xd2, _, _ = s.LinearSolve(b)                                      # re-factorisation of the same H: same bits
    assert np.array_equal(xd, xd2)
    del s
    d.Destroy()


def test_direct_large_front_config_b():
    """Config B's cube (46 875 DOF): the top separator is a ~1 900-DOF dense front, i.e. ~40 panel steps on one front and
    update grids of several hundred tiles -- the multi-tile paths the small meshes do not reach."""
    X, conn = tl.mesh_utils.structured_t10_box(12, 12, 12)
    fixed = np.where(X[:, 2] < 1e-12)[0].astype(np.int32)
    d = make_gpu(X, conn, MATERIALS["neo"], fixed)
    x = X + 5e-3 * np.sin(np.pi * X)
    d.UpdatePositions(x[:, 0], x[:, 1], x[:, 2])
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-6, 1e-4, 1e14, 5, 10, 1e-3))
    s.AssembleHessian()
    b = np.random.default_rng(11).normal(size=3 * X.shape[0])
    s.SetLinSolveOpts(tl.LinSolveOpts(method=1))
    xd, it_d, rel_d = s.LinearSolve(b)
    assert it_d == 1 and rel_d < 1e-10, rel_d
    assert np.linalg.norm(s.ApplyHessian(xd) - b) <= 1e-10 * np.linalg.norm(b)
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 20000, 10))
    xi, _, _ = s.LinearSolve(b)
    assert np.linalg.norm(xd - xi) <= 1e-7 * np.linalg.norm(xi)
    del s
    d.Destroy()


def test_direct_two_disconnected_bodies():
    """Two bodies in one mesh (the MeshManager case): the dissection's root separator is EMPTY -- a front without own
    columns, whose update matrix is just its children's -- and the solve must still equal the iterative one."""
    Xa, ca = tl.mesh_utils.structured_t10_box(3, 3, 4)
    Xb, cb = tl.mesh_utils.structured_t10_box(4, 2, 3)
    X = np.vstack([Xa, Xb + np.array([5.0, 0.0, 0.0])])
    conn = np.vstack([ca, cb + Xa.shape[0]]).astype(np.int32)
    fixed = np.where(X[:, 2] < 1e-12)[0].astype(np.int32)
    d = make_gpu(X, conn, MATERIALS["svk"], fixed)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-6, 1e-4, 1e14, 5, 10, 1e-3))
    s.AssembleHessian()
    b = np.random.default_rng(5).normal(size=3 * X.shape[0])
    s.SetLinSolveOpts(tl.LinSolveOpts(method=1))
    xd, it_d, rel_d = s.LinearSolve(b)
    assert it_d == 1 and rel_d < 1e-10
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 20000, 10, precond=1))
    xi, _, _ = s.LinearSolve(b)
    assert np.linalg.norm(xd - xi) <= 1e-7 * np.linalg.norm(xi)
    del s
    d.Destroy()


def test_direct_reports_indefinite_matrix():
    """A pivot that is not positive fails the call with a message (cuDSS reports the same through its info query)."""
    X, conn = load_mesh("res2")
    d = make_gpu(X, conn, MATERIALS["svk"], fixed_x0(X))
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.SetParameters(tl.SyncedNewtonParams(1e-4, 0.0, 1e-4, 1e14, 5, 10, -1e-3))   # negative step: H = M/h + hK is indefinite
    s.AssembleHessian()
    s.SetLinSolveOpts(tl.LinSolveOpts(method=1))
    with pytest.raises(RuntimeError, match="not positive"):
        s.LinearSolve(np.ones(3 * X.shape[0]))
    del s
    d.Destroy()


@pytest.mark.skipif(not os.environ.get("TLFEA_TEST_DIRECT"),
                    reason="opt-in TLFEA_TEST_DIRECT=1: the rocSOLVER backend maps librocsolver/librocsparse/librocblas "
                           "(1.4 GB of code objects, 4-7 minutes cold on a fresh box; profiles/r03_direct_solver_tests.log)")
def test_rocsolver_backend_in_its_own_process():
    """The rocSOLVER backend against the engine's own factorisation, in a child process (a process that has mapped
    /opt/rocm's rocBLAS must not import torch afterwards)."""
    code = (
        "import importlib, numpy as np, os\n"
        "from tests.helpers import MATERIALS, fixed_x0, load_mesh, make_gpu\n"
        "tl = importlib.import_module('total-lagrangian-fea_amd')\n"
        "X, conn = load_mesh('res2')\n"
        "d = make_gpu(X, conn, MATERIALS['svk'], fixed_x0(X))\n"
        "s = tl.SyncedNewtonSolver(d, d.get_n_constraint())\n"
        "s.SetParameters(tl.SyncedNewtonParams(1e-4, 1e-6, 1e-4, 1e14, 5, 10, 1e-3))\n"
        "s.AssembleHessian()\n"
        "b = np.random.default_rng(3).normal(size=3 * X.shape[0])\n"
        "s.SetLinSolveOpts(tl.LinSolveOpts(method=1))\n"
        "x, it, rel = s.LinearSolve(b)\n"
        "np.save(os.environ['OUT'], x)\n"
        "assert rel < 1e-9\n")
    outs = []
    for be in ("native", "rocsolver"):
        out = os.path.join(ROOT, "gpurun_out", f"direct_{be}.npy")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        env = dict(os.environ, TLFEA_DIRECT_BACKEND=be, OUT=out)
        subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, check=True, timeout=840)
        outs.append(np.load(out))
    assert np.linalg.norm(outs[0] - outs[1]) <= 1e-9 * np.linalg.norm(outs[1])
