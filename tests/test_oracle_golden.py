"""Pins the CPU oracle (oracle/tlfea_oracle.c) against golden vectors generated from the reference's
own NumPy prototypes (tools/gen_golden.py) and against the reference's data fixtures."""
import os

import numpy as np
import pytest

from oracle import orc

TAGS = ["cube", "beam_3x2x1", "res2"]
MESH_FILE = {"cube": "cube.1", "beam_3x2x1": "beam_3x2x1.1", "res2": "beam_3x2x1_res2.1"}


def relerr(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module", params=TAGS)
def case(request, golden_dir, mesh_dir):
    tag = request.param
    g = np.load(os.path.join(golden_dir, f"t10_{tag}.npz"))
    X = orc.read_nodes(os.path.join(mesh_dir, MESH_FILE[tag] + ".node"))
    conn = orc.read_elements(os.path.join(mesh_dir, MESH_FILE[tag] + ".ele"))
    return tag, g, X, conn


def test_reader_matches_reference_reader(case):
    """Connectivity / indexing is bit-exact (adaptive id base + TetGen->standard remap)."""
    _, g, X, conn = case
    assert conn.dtype == np.int32
    assert np.array_equal(conn, g["conn"])
    assert np.array_equal(X, g["X"])


def test_zero_based_mesh_reader(mesh_dir):
    """bunny_ascii_26 is the only 0-based mesh in the reference data (cpu_utils.cc:653-678,713-749)."""
    X = orc.read_nodes(os.path.join(mesh_dir, "bunny_ascii_26.1.node"))
    conn = orc.read_elements(os.path.join(mesh_dir, "bunny_ascii_26.1.ele"))
    assert X.shape == (2095, 3) and conn.shape == (1066, 10)
    assert conn.min() == 0 and conn.max() == 2094
    # first element line: 332 327 333 335 | 352 353 354 355 356 357  -> remap [0,1,2,3,6,7,9,5,8,4]
    assert conn[0].tolist() == [332, 327, 333, 335, 354, 355, 357, 353, 356, 352]


def test_keast_table():
    qx, qy, qz, qw = orc.keast5()
    assert qw[0] == (-4.0 / 5.0) * (1.0 / 6.0) and np.all(qw[1:] == (9.0 / 20.0) * (1.0 / 6.0))
    assert abs(qw.sum() - 1.0 / 6.0) < 1e-16
    assert (qx[0], qy[0], qz[0]) == (0.25, 0.25, 0.25) and qx[1] == 1.0 / 6.0 and qx[2] == 0.5


def test_gradN_detJ(case):
    _, g, X, conn = case
    o = orc.T10Oracle(X, conn, orc.svk_lame(float(g["lam"]), float(g["mu"])))
    o.calc_dndu_pre()
    assert relerr(o.detJ, g["detJ"]) < 1e-13
    assert relerr(o.gradN_a_d(), g["gradN"]) < 1e-12
    assert np.allclose(o.qw, g["wq"], rtol=0, atol=0)


def _perturbed(g, X, conn, mat):
    o = orc.T10Oracle(X, conn, mat)
    o.calc_dndu_pre()  # reference geometry from X (CalcDnDuPre before any motion)
    o.x, o.y, o.z = (np.ascontiguousarray(g["x"][:, i]) for i in range(3))
    return o


def test_internal_force_svk(case):
    _, g, X, conn = case
    o = _perturbed(g, X, conn, orc.svk_lame(float(g["lam"]), float(g["mu"])))
    f = o.internal_force()
    assert relerr(f, g["f_int"]) < 1e-12


def test_internal_force_damped(case):
    _, g, X, conn = case
    mat = orc.svk_lame(float(g["lam"]), float(g["mu"]), eta=float(g["eta_damp"]), lamd=float(g["lam_damp"]))
    o = _perturbed(g, X, conn, mat)
    f = o.internal_force(np.ascontiguousarray(g["v"].reshape(-1)))
    assert relerr(f, g["f_int_damped"]) < 1e-12


def test_element_tangents(case):
    _, g, X, conn = case
    mat = orc.svk_lame(float(g["lam"]), float(g["mu"]), eta=float(g["eta_damp"]), lamd=float(g["lam_damp"]))
    o = _perturbed(g, X, conn, mat)
    Ke, Ce = o.element_tangents(want_vis=True)
    assert relerr(Ke, g["Ke"]) < 1e-12
    assert relerr(Ce, g["Cvis"]) < 1e-12
    assert relerr(Ke, Ke.transpose(0, 2, 1)) < 1e-13  # symmetric


def test_mass_matrix(case):
    _, g, X, conn = case
    o = orc.T10Oracle(X, conn, orc.svk_lame(float(g["lam"]), float(g["mu"]), rho0=float(g["rho0"])))
    o.calc_dndu_pre()
    o.calc_mass()
    N = o.N
    M = np.zeros((N, N))
    for i in range(N):
        M[i, o.m_col[o.m_off[i]:o.m_off[i + 1]]] = o.m_val[o.m_off[i]:o.m_off[i + 1]]
        cols = o.m_col[o.m_off[i]:o.m_off[i + 1]]
        assert np.all(np.diff(cols) > 0)  # sorted, unique (binary search depends on it)
    assert relerr(M, g["M_scalar"]) < 1e-12
    # pattern == union of element node pairs
    assert np.array_equal(M != 0, g["M_scalar"] != 0) or np.count_nonzero(M) >= np.count_nonzero(g["M_scalar"])


def test_assembled_hessian_equals_sum_of_golden_Ke(case):
    """H = M/h + h*K + C_vis + h^2 rho J^T J against a dense assembly of the prototype's K_e/C_vis."""
    _, g, X, conn = case
    h, rho = 1e-3, 1e14
    mat = orc.svk_lame(float(g["lam"]), float(g["mu"]), rho0=float(g["rho0"]), eta=float(g["eta_damp"]),
                       lamd=float(g["lam_damp"]))
    fixed = np.where(np.isclose(X[:, 0], 0.0))[0].astype(np.int32)
    o = orc.T10Oracle(X, conn, mat, fixed=fixed)
    o.calc_dndu_pre()
    o.calc_mass()
    o.x, o.y, o.z = (np.ascontiguousarray(g["x"][:, i]) for i in range(3))
    ro, ci, val = o.assemble_hessian(h, rho)
    n = 3 * o.N
    Hd = np.zeros((n, n))
    for r in range(n):
        Hd[r, ci[ro[r]:ro[r + 1]]] = val[ro[r]:ro[r + 1]]
    ref = np.zeros((n, n))
    for e in range(o.E):
        dofs = (3 * conn[e][:, None] + np.arange(3)[None, :]).reshape(-1)
        ref[np.ix_(dofs, dofs)] += h * g["Ke"][e] + g["Cvis"][e]
    ref += np.kron(g["M_scalar"], np.eye(3)) / h
    for nd in fixed:
        for d in range(3):
            ref[3 * nd + d, 3 * nd + d] += h * h * rho
    assert relerr(Hd, ref) < 1e-12


def test_newton_steps_match_prototype(golden_dir):
    """3 ALM/Newton steps on beam_3x2x1 (prototype __main__ setup, f-form-T10-beam-newton.py:373-397).
    Both sides converge the inner Newton to round-off, so the stopping-rule difference
    (SURVEY App. C items 2,4) does not enter; one outer iteration suffices on both (||c||~1e-13)."""
    g = np.load(os.path.join(golden_dir, "t10_beam_3x2x1_newton.npz"))
    X, conn = g["X"], g["conn"]
    mat = orc.svk_lame(float(g["lam"]), float(g["mu"]), rho0=float(g["rho0"]))
    o = orc.T10Oracle(X, conn, mat, fixed=g["fixed"], f_ext=g["f_ext"])
    o.calc_dndu_pre()
    o.calc_mass()
    prm = orc.NewtonParams(1e-9, 0.0, 1e-6, float(g["rho"]), 5, 30, float(g["h"]))
    for step in range(3):
        st = o.newton_step(prm, solver=0)
        x = np.stack([o.x, o.y, o.z], axis=1)
        disp_ref = g["x_steps"][step] - X
        err = np.max(np.abs(x - g["x_steps"][step])) / np.max(np.abs(disp_ref))
        assert err < 1e-8, (step, err, st)
        assert st[0] == g["outer_iters"][step]


def test_direct_and_pcg_solvers_agree(golden_dir):
    g = np.load(os.path.join(golden_dir, "t10_res2.npz"))
    X, conn = g["X"], g["conn"]
    mat = orc.svk_lame(float(g["lam"]), float(g["mu"]), rho0=float(g["rho0"]))
    fixed = np.where(np.isclose(X[:, 0], 0.0))[0].astype(np.int32)
    o = orc.T10Oracle(X, conn, mat, fixed=fixed)
    o.calc_dndu_pre()
    o.calc_mass()
    ro, ci, val = o.assemble_hessian(1e-3, 1e14)
    rng = np.random.default_rng(0)
    b = rng.normal(size=3 * o.N)
    x1 = orc.solve_spd_upper(ro, ci, val, b)
    x2, it = orc.solve_pcg(ro, ci, val, b, rel_tol=1e-13)
    assert it > 0
    assert relerr(x2, x1) < 1e-9
    # residual of the direct solve
    r = b.copy()
    for row in range(3 * o.N):
        r[row] -= val[ro[row]:ro[row + 1]] @ x1[ci[ro[row]:ro[row + 1]]]
    assert np.linalg.norm(r) / np.linalg.norm(b) < 1e-9
