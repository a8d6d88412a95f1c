"""SyncedVBDSolver on the HIP element kernels vs the oracle's restatement (same colouring, same sweeps): T10 with both
materials and damping, ANCF-3243 beam and ANCF-3443 shell; colouring bit-exact; the reference driver's parameters."""
import importlib

import numpy as np
import pytest

from oracle import orc
from tests.helpers import MATERIALS, fixed_x0, load_mesh, make_gpu, make_oracle
from tests.test_gpu_parity import disp_err_ok

tl = importlib.import_module("total-lagrangian-fea_amd")
pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def t10_pair(mesh, matname, load=1000.0):
    X, conn = load_mesh(mesh)
    fixed = fixed_x0(X)
    f_ext = np.zeros(3 * X.shape[0])
    tip = int(np.argmax(X[:, 0] + 1e-3 * X[:, 1] + 1e-6 * X[:, 2]))
    f_ext[3 * tip] = load
    f_ext[3 * tip + 2] = -0.5 * load
    m = MATERIALS[matname]
    return X, make_oracle(X, conn, m, fixed, f_ext), make_gpu(X, conn, m, fixed, f_ext)


def vbd_pair(d, o, kw):
    s = tl.SyncedVBDSolver(d, d.get_n_constraint())
    s.Setup()
    s.SetParameters(tl.SyncedVBDParams(**kw))
    s.InitializeColoring()
    s.InitializeMassDiagBlocks()
    s.InitializeFixedMap()
    o.vbd_coloring(kw["color_group_size"])
    return s, orc.VbdParams(*[kw[k] for k, _ in orc.VbdParams._fields_])


def params(**kw):
    base = dict(inner_tol=1e-4, inner_rtol=1e-4, outer_tol=1e-4, rho=1e14, max_outer=5, max_inner=500, time_step=1e-3,
                omega=1.8, hess_eps=1e-12, convergence_check_interval=25, color_group_size=1)  # test_feat10_resolution.cc:379
    base.update(kw)
    return base


@pytest.mark.parametrize("mesh", ["beam_3x2x1", "res2", "bunny"])
def test_coloring_matches_oracle_bit_exact(mesh):
    X, o, d = t10_pair(mesh, "svk")
    s, _ = vbd_pair(d, o, params())
    g = s.GetColoring()
    for k in ("colors", "color_offsets", "color_nodes", "group_offsets", "group_colors"):
        assert np.array_equal(g[k], o.vbd[k]), k
    assert (g["n_colors"], g["n_groups"]) == (o.vbd["n_colors"], o.vbd["n_groups"])
    del s
    d.Destroy()


@pytest.mark.parametrize("matname", ["svk", "mr", "svk_damped", "mr_damped"])
def test_t10_fixed_sweeps_match_oracle(matname):
    """Tolerances that cannot trigger and no convergence checks: both sides run exactly max_outer x max_inner sweeps of
    the same coloured recurrence; positions, velocities and multipliers agree to round-off."""
    X, o, d = t10_pair("beam_3x2x1", matname)
    kw = params(inner_tol=0.0, inner_rtol=0.0, outer_tol=0.0, max_outer=2, max_inner=15, convergence_check_interval=0)
    if "damped" in matname:
        # the local Hessian has no Kelvin-Voigt block (FEAT10DataFunc.cuh:295-395 adds the elastic K_aa only) while the
        # residual carries the viscous stress: with eta = 1e5 the sweep is not a contraction and amplifies round-off,
        # in any implementation; few sweeps without over-relaxation keep the comparison at round-off level
        kw.update(max_inner=3, omega=1.0)
    s, oprm = vbd_pair(d, o, kw)
    for step in range(2):
        s.Solve()
        st_o = o.vbd_step(oprm)
        st = s.GetStats()
        assert (st["outer"], st["sweeps"]) == (int(st_o[0]), int(st_o[1])) == (2, 2 * kw["max_inner"])
        xg = np.stack(d.RetrievePositionToCPU(), axis=1)
        xo = np.stack([o.x, o.y, o.z], axis=1)
        assert disp_err_ok(xg, xo, X), (matname, step)
        assert relerr(s.RetrieveVelocityToCPU(), o.v) < 1e-9
        # lambda += rho c with c = x - X a difference of O(1) coordinates: one ulp of x is worth rho * 2e-16 in lambda
        lam_floor = kw["rho"] * 8 * np.finfo(np.float64).eps * np.abs(xo).max() * kw["max_outer"] * (step + 1)
        assert np.abs(s.RetrieveLambdaToCPU() - o.lam).max() <= 1e-9 * np.abs(o.lam).max() + lam_floor
    del s
    d.Destroy()


@pytest.mark.parametrize("mesh", ["beam_3x2x1", "res2"])
def test_t10_driver_parameters_match_oracle(mesh):
    """The reference driver's parameters ({1e-4,1e-4,1e-4,1e14,5,500,dt,1.8,1e-12,25,1}): same number of outer
    iterations and sweeps, same ||g|| at the last check, same ||c||, same positions."""
    X, o, d = t10_pair(mesh, "svk")
    s, oprm = vbd_pair(d, o, params())
    for step in range(2):
        s.Solve()
        st_o = o.vbd_step(oprm)
        st = s.GetStats()
        assert (st["outer"], st["sweeps"]) == (int(st_o[0]), int(st_o[1])), (st, st_o)
        # g holds h rho c on the pinned nodes, c = x - X a difference of O(1) coordinates: an ulp of x is worth
        # h rho 2e-16 there, whatever the implementation
        g_floor = 1e-3 * 1e14 * np.finfo(np.float64).eps * np.abs(X).max()
        assert abs(st["norm_g"] - st_o[2]) <= 1e-9 * st_o[2] + g_floor
        assert abs(st["norm_c"] - st_o[3]) <= 1e-9 * st_o[3] + np.finfo(np.float64).eps * np.abs(X).max()  # same floor
        xg = np.stack(d.RetrievePositionToCPU(), axis=1)
        xo = np.stack([o.x, o.y, o.z], axis=1)
        assert disp_err_ok(xg, xo, X), (mesh, step)
    del s
    d.Destroy()


def test_vbd_reaches_the_newton_solution_on_the_gpu():
    X, o, d = t10_pair("res2", "svk")
    s, _ = vbd_pair(d, o, params(inner_tol=1e-9, inner_rtol=1e-9, outer_tol=1e-6, max_inner=3000, omega=1.0,
                                 convergence_check_interval=20))
    s.Solve()
    xg = np.stack(d.RetrievePositionToCPU(), axis=1)
    o.newton_step(orc.NewtonParams(1e-8, 0.0, 1e-6, 1e14, 5, 20, 1e-3), solver=0)
    xo = np.stack([o.x, o.y, o.z], axis=1)
    disp = np.abs(xo - X).max()
    assert disp > 1e-7 and np.abs(xg - xo).max() < 2e-5 * disp
    del s
    d.Destroy()


@pytest.mark.parametrize("pname", ["beam3243", "shell3443"])
@pytest.mark.parametrize("group_size", [1, 4])
def test_ancf_fixed_sweeps_match_oracle(pname, group_size):
    from tests.test_gpu_ancf import PROBLEMS, SVK_D, make_pair
    o, d = make_pair(PROBLEMS[pname](), SVK_D)
    kw = params(inner_tol=0.0, inner_rtol=0.0, outer_tol=0.0, max_outer=2, max_inner=12, convergence_check_interval=0,
                omega=1.0, color_group_size=group_size)
    s, oprm = vbd_pair(d, o, kw)
    g = s.GetColoring()
    for k in ("colors", "color_nodes", "group_offsets", "group_colors"):
        assert np.array_equal(g[k], o.vbd[k]), k
    X0 = np.stack([o.xt, o.yt, o.zt], axis=1)
    for step in range(2):
        s.Solve()
        o.vbd_step(oprm)
        xg = np.stack(d.RetrievePositionToCPU(), axis=1)
        xo = np.stack([o.x, o.y, o.z], axis=1)
        assert disp_err_ok(xg, xo, X0), (pname, step)
        assert relerr(s.RetrieveVelocityToCPU(), o.v) < 1e-9
    del s
    d.Destroy()


def test_vbd_rejects_general_linear_constraints():
    """The reference's VBD knows pinned nodes only (its fixed map, SyncedVBD.cu:146-160, 1087-1135)."""
    from tests.test_gpu_ancf import PROBLEMS, SVK
    Q = tl.quadrature
    kind, x, y, z, conn, (L, W, H), fixed, f_ext = PROBLEMS["beam3243"]()
    d = tl.GPU_ANCF3243_Data(len(x) // 4, conn.shape[0])
    d.Initialize()
    d.SetExternalForce(f_ext)
    d.Setup(L, W, H, Q.gauss_xi_m_6, Q.gauss_xi_3, Q.gauss_eta_2, Q.gauss_zeta_2, Q.weight_xi_m_6, Q.weight_xi_3,
            Q.weight_eta_2, Q.weight_zeta_2, x, y, z, conn)
    d.SetDensity(SVK["rho0"])
    d.SetDamping(0.0, 0.0)
    d.SetSVK(SVK["E"], SVK["nu"])
    b = tl.mesh_utils.LinearConstraintBuilder(3 * d.get_n_coef())
    b.AddFixedDof(0, 0.0)
    csr = b.ToCSR()
    d.SetLinearConstraintsCSR(csr.offsets, csr.columns, csr.values, csr.rhs)
    d.CalcDsDuPre()
    d.CalcMassMatrix()
    s = tl.SyncedVBDSolver(d, d.get_n_constraint())
    s.Setup()
    with pytest.raises(tl.TlfeaError):
        s.InitializeFixedMap()
    del s
    d.Destroy()
