"""GPU parity of the general-linear-constraint path (SetLinearConstraintsCSR, constraint-aware Hessian pattern;
SURVEY 8f-1) through the C-ABI against the oracle, on the reference's own mesh files: the welded / pinned 20x20
ANCF-3243 nets of lib_bin/mesh_deform/test_ancf3243_net_newton.cc and the ANCF-3443 airless tire of
test_ancf3443_mesh_newton.cc (same materials, clamps and solver parameters as those drivers)."""
import numpy as np
import pytest

from oracle import orc
from tests.helpers import relerr, tl
from tests.test_gpu_parity import disp_err_ok
from tests.test_linear_constraints import NET_P, NET_W, TIRE

pytestmark = pytest.mark.gpu
Q = tl.quadrature
mu = tl.mesh_utils


def net_problem(path):
    """test_ancf3243_net_newton.cc:405-457: L from the grid line, W = H = 0.1, SVK 7e8/0.33, rho 2700, damping 1e5,
    the four corner joints clamped (all four coefficients of both nodes), -1000 N on the centre joint."""
    m = mu.ReadANCF3243MeshFromFile(path)
    b = mu.LinearConstraintBuilder(12 * m.n_nodes, m.constraints)
    px, py = m.x12[0::4], m.y12[0::4]
    for cx, cy in ((px.min(), py.min()), (px.max(), py.min()), (px.min(), py.max()), (px.max(), py.max())):
        for nid in np.where((np.abs(px - cx) < 1e-9) & (np.abs(py - cy) < 1e-9))[0]:
            for slot in range(4):
                mu.AppendANCF3243FixedCoefficient(b, 4 * int(nid) + slot, m.x12, m.y12, m.z12)
    f_ext = np.zeros(12 * m.n_nodes)
    cx, cy = 0.5 * (px.min() + px.max()), 0.5 * (py.min() + py.max())
    loaded = np.where((np.abs(px - cx) < 1e-9) & (np.abs(py - cy) < 1e-9))[0]
    assert len(loaded) == 2
    f_ext[(4 * loaded) * 3 + 2] = -1000.0 / len(loaded)
    mat = dict(E=7e8, nu=0.33, rho0=2700.0, eta=1e5, lamd=1e5)
    prm = (1e-4, 0.0, 1e-6, 1e14, 5, 10, 1e-3)
    return 3243, m, (m.grid_L, 0.1, 0.1), b.ToCSR(), f_ext, mat, prm


def tire_problem():
    """test_ancf3443_mesh_newton.cc:236-331: thickness x 0.25, SVK 1e8/0.33, rho 2000, damping 5e4, hub (inner spoke
    ends) clamped through AppendANCF3243FixedCoefficient rows, rho = 1e12, 10 outer iterations; a fixed downward
    load on the ring nodes stands in for the driver's state-dependent ground contact."""
    m = mu.ReadANCF3443MeshFromFile(TIRE)
    b = mu.LinearConstraintBuilder(12 * m.n_nodes, m.constraints)
    r = np.hypot(m.x12[0::4], m.z12[0::4])
    spoke = np.array([f == "S" for f in m.node_family])
    hub = np.where(spoke & (r < r[spoke].min() + 1e-9))[0]
    assert len(hub) > 0
    for nid in hub:
        for slot in range(4):
            mu.AppendANCF3243FixedCoefficient(b, 4 * int(nid) + slot, m.x12, m.y12, m.z12)
    f_ext = np.zeros(12 * m.n_nodes)
    ring = np.where(np.array([f == "R" for f in m.node_family]) & (m.z12[0::4] < -0.2))[0]
    f_ext[(4 * ring) * 3 + 2] = 5.0   # small enough for the undamped Newton iteration to converge in 4 steps
    mat = dict(E=1e8, nu=0.33, rho0=2000.0, eta=5e4, lamd=5e4)
    prm = (1e-4, 0.0, 1e-6, 1e12, 10, 10, 1e-3)
    return 3443, m, (m.element_L, m.element_W, 0.25 * m.element_H), b.ToCSR(), f_ext, mat, prm


def make_pair(prob):
    kind, m, (L, W, H), csr, f_ext, mk, _ = prob
    mat = orc.svk(mk["E"], mk["nu"], rho0=mk["rho0"], eta=mk["eta"], lamd=mk["lamd"])
    x, y, z, conn = m.x12, m.y12, m.z12, m.element_connectivity
    o = orc.AncfOracle(kind, x, y, z, conn, L, W, H, mat, f_ext=f_ext)
    o.calc_dsdu_pre()
    o.calc_mass()
    o.set_linear_constraints(csr.offsets, csr.columns, csr.values, csr.rhs)
    d = (tl.GPU_ANCF3243_Data if kind == 3243 else tl.GPU_ANCF3443_Data)(m.n_nodes, m.n_elements)
    d.Initialize()
    d.SetExternalForce(f_ext)
    if kind == 3243:
        d.Setup(L, W, H, Q.gauss_xi_m_6, Q.gauss_xi_3, Q.gauss_eta_2, Q.gauss_zeta_2, Q.weight_xi_m_6, Q.weight_xi_3,
                Q.weight_eta_2, Q.weight_zeta_2, x, y, z, conn)
    else:
        d.Setup(L, W, H, Q.gauss_xi_m_7, Q.gauss_eta_m_7, Q.gauss_zeta_m_3, Q.gauss_xi_4, Q.gauss_eta_4, Q.gauss_zeta_3,
                Q.weight_xi_m_7, Q.weight_eta_m_7, Q.weight_zeta_m_3, Q.weight_xi_4, Q.weight_eta_4, Q.weight_zeta_3,
                x, y, z, conn)
    d.SetDensity(mk["rho0"])
    d.SetDamping(mk["eta"], mk["lamd"])
    d.SetSVK(mk["E"], mk["nu"])
    d.SetLinearConstraintsCSR(csr.offsets, csr.columns, csr.values, csr.rhs)   # after Setup, as the drivers do
    d.CalcDsDuPre()
    d.CalcMassMatrix()
    d.CalcConstraintData()
    return o, d


def perturb(o, d, seed=11, sigma=1e-4, vsigma=1e-2):
    rng = np.random.default_rng(seed)
    dx = rng.normal(0.0, sigma, (3, o.N))
    o.x, o.y, o.z = o.x + dx[0], o.y + dx[1], o.z + dx[2]
    d.UpdatePositions(o.x, o.y, o.z)
    return rng.normal(0.0, vsigma, 3 * o.N)


PROBLEMS = {"net_welded": lambda: net_problem(NET_W), "net_pinned": lambda: net_problem(NET_P), "tire": tire_problem}


@pytest.mark.parametrize("pname", sorted(PROBLEMS))
def test_constraint_data_jacobians_and_mass_pattern(pname):
    prob = PROBLEMS[pname]()
    csr = prob[3]
    o, d = make_pair(prob)
    assert d.GetConstraintMode() == 2 and d.get_n_constraint() == csr.NumRows()
    perturb(o, d)
    d.CalcConstraintData()
    c_ref = o.lin_constraint()
    assert np.max(np.abs(d.RetrieveConstraintDataToCPU() - c_ref)) <= 1e-15 * max(1.0, np.abs(c_ref).max())
    off, col, val = d.RetrieveConstraintJacobianCSRToCPU()
    assert np.array_equal(off, csr.offsets) and np.array_equal(col, csr.columns) and np.array_equal(val, csr.values)
    toff, tcol, tval = d.RetrieveConstraintJacobianTransposeCSRToCPU()
    import scipy.sparse as sp
    J = sp.csr_matrix((csr.values, csr.columns, csr.offsets), shape=(csr.NumRows(), 3 * o.N))
    JT = sp.csr_matrix((tval, tcol, toff), shape=(3 * o.N, csr.NumRows()))
    assert (J.T != JT).nnz == 0
    for i in range(3 * o.N):
        assert np.all(np.diff(tcol[toff[i]:toff[i + 1]]) > 0)      # ascending constraint id per DOF row
    # the mass CSR stays the element pattern although the internal adjacency is constraint-aware
    m_off, m_col, m_val = d.RetrieveMassCSRToCPU()
    assert np.array_equal(m_off, o.m_off) and np.array_equal(m_col, o.m_col)
    assert relerr(m_val, o.m_val) < 1e-12
    d.Destroy()


@pytest.mark.parametrize("pname", sorted(PROBLEMS))
def test_gradient_and_hessian(pname):
    prob = PROBLEMS[pname]()
    h, rho = prob[6][6], prob[6][3]
    o, d = make_pair(prob)
    v = perturb(o, d)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.Setup()
    s.SetParameters(tl.SyncedNewtonParams(*prob[6]))
    s.SetVelocity(v)
    o.v[:] = v
    lam = np.linspace(-3.0, 3.0, o.nc)
    o.lam[:] = lam
    s.SetLambda(lam)
    ng = s.EvalGradient()
    g_ref = o.grad_L_lin(o.internal_force(o.v), h, rho)
    assert relerr(s.RetrieveGradientToCPU(), g_ref) < 1e-12 and abs(ng - np.linalg.norm(g_ref)) < 1e-10 * ng
    s.AssembleHessian()
    ro, ci, val = s.RetrieveHessianCSRToCPU()
    ro_o, ci_o, val_o = o.assemble_hessian_lin(h, rho)
    assert np.array_equal(ro, ro_o) and np.array_equal(ci, ci_o)     # constraint-aware pattern, bit-exact
    assert len(ci_o) > 9 * len(o.m_col)                              # ... and it is larger than the element pattern
    assert relerr(val, val_o) < 1e-12
    del s
    d.Destroy()


@pytest.mark.parametrize("pname,steps", [("net_welded", 2), ("net_pinned", 2), ("tire", 1)])
def test_newton_steps(pname, steps):
    prob = PROBLEMS[pname]()
    o, d = make_pair(prob)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.Setup()
    s.SetParameters(tl.SyncedNewtonParams(*prob[6]))
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 100000, 10))
    s.AnalyzeHessianSparsity()
    oprm = orc.NewtonParams(*prob[6])
    X0 = np.stack([o.xt, o.yt, o.zt], axis=1)
    for _ in range(steps):
        s.Solve()
        st_o = o.newton_step_lin(oprm)
        st_g = s.GetStats()
        xg = np.stack(d.RetrievePositionToCPU(), axis=1)
        xo = np.stack([o.x, o.y, o.z], axis=1)
        assert disp_err_ok(xg, xo, X0), (st_g, st_o)
        assert st_g["outer"] == st_o[0] and st_g["newton"] == st_o[1]
        # lambda += rho c amplifies the round-off of c (1e-16 of a coordinate) by rho = 1e12..1e14
        assert relerr(s.RetrieveLambdaToCPU(), o.lam) < 1e-4
    del s
    d.Destroy()


def test_linear_rows_reproduce_the_fixed_coefficient_path():
    """The same clamp written as SetNodalFixed and as AddFixedDof rows: identical Newton step on the device."""
    from tests.test_gpu_ancf import SVK_D, beam_problem
    from tests.test_gpu_ancf import make_pair as make_fixed
    prob = beam_problem()
    kind, x, y, z, conn, (L, W, H), fixed, f_ext = prob
    o, d_fix = make_fixed(prob, SVK_D)
    bld = mu.LinearConstraintBuilder(3 * len(x))
    for c in fixed:
        mu.AppendANCF3243FixedCoefficient(bld, int(c), x, y, z)
    csr = bld.ToCSR()
    d_lin = tl.GPU_ANCF3243_Data(len(x) // 4, conn.shape[0])
    d_lin.Initialize()
    d_lin.SetLinearConstraintsCSR(csr.offsets, csr.columns, csr.values, csr.rhs)   # before Setup works too
    d_lin.SetExternalForce(f_ext)
    d_lin.Setup(L, W, H, Q.gauss_xi_m_6, Q.gauss_xi_3, Q.gauss_eta_2, Q.gauss_zeta_2, Q.weight_xi_m_6, Q.weight_xi_3,
                Q.weight_eta_2, Q.weight_zeta_2, x, y, z, conn)
    d_lin.SetDensity(SVK_D["rho0"])
    d_lin.SetDamping(SVK_D["eta"], SVK_D["lamd"])
    d_lin.SetSVK(SVK_D["E"], SVK_D["nu"])
    d_lin.CalcDsDuPre()
    d_lin.CalcMassMatrix()
    out = []
    for d in (d_fix, d_lin):
        s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
        s.Setup()
        s.SetParameters(tl.SyncedNewtonParams(1e-4, 0.0, 1e-6, 1e14, 5, 10, 1e-3))
        s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 50000, 10))
        s.Solve()
        out.append((np.stack(d.RetrievePositionToCPU(), axis=1), s.GetStats()))
        del s
        d.Destroy()
    assert out[0][1]["outer"] == out[1][1]["outer"] and out[0][1]["newton"] == out[1][1]["newton"]
    X0 = np.stack([x, y, z], axis=1)
    assert disp_err_ok(out[1][0], out[0][0], X0)


def test_set_linear_constraints_errors():
    d = tl.GPU_ANCF3243_Data(3, 2)
    d.Initialize()
    with pytest.raises(ValueError):
        d.SetLinearConstraintsCSR([1, 2], [0], [1.0], [0.0])
    with pytest.raises(RuntimeError, match="column out of range"):
        d.SetLinearConstraintsCSR([0, 1], [36], [1.0], [0.0])
    d.SetLinearConstraintsCSR([0, 1], [35], [1.0], [0.0])
    with pytest.raises(RuntimeError, match="already set up"):
        d.SetLinearConstraintsCSR([0, 1], [0], [1.0], [0.0])
    d.Destroy()


def tire_drive_inputs(m, csr, hub_row0, hub_coefs, z12, step, theta, dt, ground_z=-0.2, k=5e4, fz_max=2e4):
    """One step of the reference tire driver's inputs (test_ancf3443_mesh_newton.cc:86-121, 345-374): ground contact on
    the ring nodes below the plane (penalty k, clamped) and the hub coefficients rotated about y by the ramped angle."""
    f_ext = np.zeros(12 * m.n_nodes)
    for nid in range(m.n_nodes):
        if m.node_family[nid] == "R" and z12[4 * nid] < ground_z:
            f_ext[(4 * nid) * 3 + 2] += min(k * (ground_z - z12[4 * nid]), fz_max)
    s = np.clip(((step + 0.5) * dt) / 0.05, 0.0, 1.0)
    theta = theta + 1.5 * np.pi * (s * s * (3.0 - 2.0 * s)) * dt
    rhs = csr.rhs.copy()
    c, sn = np.cos(theta), np.sin(theta)
    for i, coef in enumerate(hub_coefs):
        x, y, z = m.x12[coef], m.y12[coef], m.z12[coef]
        rhs[hub_row0 + 3 * i:hub_row0 + 3 * i + 3] = (c * x + sn * z, y, -sn * x + c * z)
    return f_ext, rhs, theta


def test_tire_driver_flow_prescribed_hub_rotation_and_ground_contact():
    """The loop of lib_bin/mesh_deform/test_ancf3443_mesh_newton.cc: every step recomputes the ground-contact forces from
    the current ring positions (SetExternalForce) and moves the hub through UpdateLinearConstraintRHS (J and the
    sparsity stay); device vs oracle over three steps.  Contact stiffness scaled down so that the undamped Newton
    iteration of the test converges within the driver's 10 iterations."""
    prob = tire_problem()
    kind, m, dims, csr, _, mk, prm = prob
    o, d = make_pair(prob)
    hub_row0 = m.constraints.NumRows() if hasattr(m.constraints, "NumRows") else len(m.constraints.rhs)
    r = np.hypot(m.x12[0::4], m.z12[0::4])
    spoke = np.array([f == "S" for f in m.node_family])
    hub = np.where(spoke & (r < r[spoke].min() + 1e-9))[0]
    hub_coefs = [4 * int(n) + s_ for n in hub for s_ in range(4)]
    assert hub_row0 + 3 * len(hub_coefs) == len(csr.rhs)
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.Setup()
    s.SetParameters(tl.SyncedNewtonParams(*prm))
    s.SetLinSolveOpts(tl.LinSolveOpts(1e-13, 50000, 10))
    X0 = np.stack([o.xt, o.yt, o.zt], axis=1)
    theta_g = theta_o = 0.0
    with pytest.raises(tl.TlfeaError):
        d.UpdateLinearConstraintRHS(np.zeros(3))          # size mismatch is refused
    for step in range(3):
        zg = d.RetrievePositionToCPU()[2]
        fg, rg, theta_g = tire_drive_inputs(m, csr, hub_row0, hub_coefs, zg, step, theta_g, 1e-3, ground_z=-0.2, k=100.0)
        fo, ro, theta_o = tire_drive_inputs(m, csr, hub_row0, hub_coefs, o.z, step, theta_o, 1e-3, ground_z=-0.2, k=100.0)
        assert np.count_nonzero(fo) > 0 and np.abs(fg - fo).max() <= 1e-6 * np.abs(fo).max()
        d.SetExternalForce(fg)
        d.UpdateLinearConstraintRHS(rg)
        o.f_ext[:] = fo
        o.j_rhs[:] = ro
        s.Solve()
        o.newton_step_lin(orc.NewtonParams(*prm))
        xg = np.stack(d.RetrievePositionToCPU(), axis=1)
        xo = np.stack([o.x, o.y, o.z], axis=1)
        assert disp_err_ok(xg, xo, X0), step
    # the hub followed the prescribed rotation
    hub_pos = [4 * int(n) for n in hub]
    want = np.stack([np.cos(theta_o) * m.x12[hub_pos] + np.sin(theta_o) * m.z12[hub_pos],
                     -np.sin(theta_o) * m.x12[hub_pos] + np.cos(theta_o) * m.z12[hub_pos]], axis=1)
    got = np.stack([o.x[hub_pos], o.z[hub_pos]], axis=1)
    assert np.abs(got - want).max() < 1e-8 and theta_o > 0
    del s
    d.Destroy()
