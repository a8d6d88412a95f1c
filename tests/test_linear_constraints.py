"""General linear constraints + ANCF mesh files: LinearConstraintBuilder / Append* helpers / readers
(mesh_utils.h:105-245, mesh_utils.cc:170-1010) and the oracle's kConstraintLinearCSR path (ANCF3243Data.cuh:803-940,
SyncedNewton.cu:292-341,377-404,556-801).  The reference holds no numeric fixture for this path (its drivers only
print residuals), so the oracle is cross-pinned against its own fixed-coefficient path, which IS pinned: a pinned
coefficient written as three `AddFixedDof` rows must give the same gradient, Hessian and Newton step."""
import os

import numpy as np
import pytest

from oracle import orc
from tests.helpers import tl

mu = tl.mesh_utils
MESH = os.path.join(os.path.dirname(__file__), "golden", "meshes")
NET_W = os.path.join(MESH, "ANCF3243", "net_welded_nx20_ny20_L0.5.ancf3243mesh")
NET_P = os.path.join(MESH, "ANCF3243", "net_pinned_nx20_ny20_L0.5.ancf3243mesh")
TIRE = os.path.join(MESH, "ANCF3443", "airless_tire.ancf3443mesh")


def relerr(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


# ---- builder ----------------------------------------------------------------------------------------------
def test_builder_rows_and_zero_dropping():
    b = mu.LinearConstraintBuilder(30)
    assert b.AddRow([(3, 1.0), (7, 0.0), (9, -2.5)], 0.25) == 0     # zero coefficient dropped (mesh_utils.cc:224)
    assert b.AddFixedDof(5, 1.5) == 1
    c = b.ToCSR()
    assert c.offsets.tolist() == [0, 2, 3] and c.columns.tolist() == [3, 9, 5]
    assert c.values.tolist() == [1.0, -2.5, 1.0] and c.rhs.tolist() == [0.25, 1.5]
    assert (c.NumRows(), c.NumNonZeros(), c.Empty()) == (2, 3, False)
    b2 = mu.LinearConstraintBuilder(30, c)                          # continue from an existing CSR (:181-208)
    b2.AddFixedDof(0, 0.0)
    assert b2.num_rows() == 3 and b2.nnz() == 4


def test_builder_errors():
    with pytest.raises(ValueError):
        mu.LinearConstraintBuilder(0)
    b = mu.LinearConstraintBuilder(6)
    with pytest.raises(ValueError):
        b.AddRow([], 0.0)
    with pytest.raises(IndexError):
        b.AddRow([(6, 1.0)], 0.0)
    with pytest.raises(IndexError):
        mu.AppendANCF3243VectorEqualityConstraint(mu.LinearConstraintBuilder(100), 0, 1, 4)


def test_welded_rows_follow_the_reference_layout():
    # welded a=0 b=1, Q = [[0,-1,0],[1,0,0],[0,0,1]]: position equality (slot 0) then r_b - Q r_a for slots 1..3
    b = mu.LinearConstraintBuilder(24)
    Q = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    mu.AppendANCF3243VectorEqualityConstraint(b, 0, 1, 0)
    mu.AppendANCF3243VectorWeldedConstraint(b, 0, 1, 1, Q)
    c = b.ToCSR()
    dof = lambda node, slot, comp: (4 * node + slot) * 3 + comp  # noqa: E731
    rows = [(c.columns[c.offsets[r]:c.offsets[r + 1]].tolist(), c.values[c.offsets[r]:c.offsets[r + 1]].tolist())
            for r in range(c.NumRows())]
    assert rows[0] == ([dof(1, 0, 0), dof(0, 0, 0)], [1.0, -1.0])
    assert rows[3] == ([dof(1, 1, 0), dof(0, 1, 1)], [1.0, 1.0])      # -Q[0,1] = +1 on a's y component
    assert rows[4] == ([dof(1, 1, 1), dof(0, 1, 0)], [1.0, -1.0])
    assert rows[5] == ([dof(1, 1, 2), dof(0, 1, 2)], [1.0, -1.0])


# ---- readers ----------------------------------------------------------------------------------------------
def test_read_ancf3243_net_meshes():
    w, p = mu.ReadANCF3243MeshFromFile(NET_W), mu.ReadANCF3243MeshFromFile(NET_P)
    for m in (w, p):
        assert (m.version, m.n_nodes, m.n_elements, m.grid_nx, m.grid_ny, m.grid_L) == (1, 882, 840, 20, 20, 0.5)
        assert m.element_connectivity.shape == (840, 2) and m.element_connectivity[:2].tolist() == [[0, 2], [2, 4]]
        assert m.node_family[:2] == ["H", "V"] and len(m.x12) == 4 * 882
        assert m.x12[8:12].tolist() == [0.5, 1.0, 0.0, 0.0]           # node 2: x0 x1 x2 x3
    # 441 joints: welded = 3 position + 9 gradient rows, pinned = 3 position rows (mesh_utils.cc:683-725)
    assert (w.constraints.NumRows(), p.constraints.NumRows()) == (441 * 12, 441 * 3)
    assert np.all(w.constraints.rhs == 0.0)
    assert p.constraints.columns[:2].tolist() == [(4 * 1) * 3, 0] and p.constraints.values[:2].tolist() == [1.0, -1.0]


def test_read_ancf3443_tire_mesh():
    m = mu.ReadANCF3443MeshFromFile(TIRE)
    assert (m.n_nodes, m.n_elements, m.constraints.NumRows()) == (1120, 720, 160 * 12)
    assert m.element_connectivity[0].tolist() == [0, 1, 121, 120]
    assert (m.element_L[0], m.element_W[0], m.element_H[0]) == (0.013088474153936575, 0.066666666666666666, 0.02)
    assert m.element_family[0] == "R" and m.z12[0] == 0.25


def test_reader_rejects_malformed_files(tmp_path):
    src = open(NET_P).read().splitlines()
    bad = tmp_path / "bad.ancf3243mesh"
    bad.write_text("\n".join(["ancf3443_mesh 1"] + src[1:]))
    with pytest.raises(ValueError, match="expected header"):
        mu.ReadANCF3243MeshFromFile(str(bad))
    k = next(i for i, l in enumerate(src) if l.startswith("0 H"))
    bad.write_text("\n".join(src[:k] + [src[k] + " 7"] + src[k + 1:]))
    with pytest.raises(ValueError, match="invalid node line"):
        mu.ReadANCF3243MeshFromFile(str(bad))
    bad.write_text("\n".join(src[:-1] + ["glued 0 1"]))
    with pytest.raises(ValueError, match="unknown constraint type"):
        mu.ReadANCF3243MeshFromFile(str(bad))


# ---- oracle: linear rows == fixed coefficients --------------------------------------------------------------
def small_beam(n_el=5):
    g = mu.GridMeshGenerator(n_el * 0.5, 0.0, 0.5)
    g.generate_mesh()
    x12, y12, z12 = g.get_coordinates()
    conn = g.get_element_connectivity()
    mat = orc.svk(7e8, 0.33, rho0=2700.0, eta=1e4, lamd=1e4)
    f_ext = np.zeros(3 * len(x12))
    f_ext[3 * (len(x12) - 4) + 2] = 3100.0
    return x12, y12, z12, conn, mat, f_ext


def make_pair():
    x12, y12, z12, conn, mat, f_ext = small_beam()
    fixed = np.arange(4, dtype=np.int32)
    a = orc.AncfOracle(3243, x12, y12, z12, conn, 0.5, 0.1, 0.1, mat, fixed=fixed, f_ext=f_ext)
    b = orc.AncfOracle(3243, x12, y12, z12, conn, 0.5, 0.1, 0.1, mat, f_ext=f_ext)
    bld = mu.LinearConstraintBuilder(3 * len(x12))
    for c in fixed:
        mu.AppendANCF3243FixedCoefficient(bld, int(c), x12, y12, z12)
    csr = bld.ToCSR()
    for o in (a, b):
        o.calc_dsdu_pre()
        o.calc_mass()
    b.set_linear_constraints(csr.offsets, csr.columns, csr.values, csr.rhs)
    rng = np.random.default_rng(5)
    dx = rng.normal(0, 1e-3, (3, len(x12)))
    v = rng.normal(0, 1e-1, 3 * len(x12))
    for o in (a, b):
        o.x, o.y, o.z = o.x + dx[0], o.y + dx[1], o.z + dx[2]
        o.v[:] = v
    return a, b


def test_oracle_linear_rows_equal_fixed_coefficients():
    a, b = make_pair()
    h, rho = 1e-3, 1e14
    assert np.array_equal(a.constraint(), b.lin_constraint())
    a.lam[:] = b.lam[:] = np.linspace(-1, 1, len(a.lam))
    f = a.internal_force(a.v)
    assert np.array_equal(a.grad_L(f, h, rho), b.grad_L_lin(f, h, rho))
    ra, ca, va = a.assemble_hessian(h, rho)
    rb, cb, vb = b.assemble_hessian_lin(h, rho)
    assert np.array_equal(ra, rb) and np.array_equal(ca, cb)       # single-DOF rows add no fill
    assert np.array_equal(va, vb)


def test_oracle_newton_step_with_linear_rows_equals_fixed_path():
    a, b = make_pair()
    prm = orc.NewtonParams(1e-4, 0.0, 1e-6, 1e14, 5, 10, 1e-3)
    for _ in range(2):
        sa, sb = a.newton_step(prm), b.newton_step_lin(prm)
        assert sa[:2].tolist() == sb[:2].tolist()
        for p, q in ((a.x, b.x), (a.y, b.y), (a.z, b.z), (a.lam, b.lam)):
            assert np.array_equal(p, q)


def test_oracle_constraint_aware_adjacency():
    m = mu.ReadANCF3243MeshFromFile(NET_P)
    mat = orc.svk(7e8, 0.33, rho0=2700.0)
    o = orc.AncfOracle(3243, m.x12, m.y12, m.z12, m.element_connectivity, 0.5, 0.1, 0.1, mat)
    o.mass_pattern()
    c = m.constraints
    o.set_linear_constraints(c.offsets, c.columns, c.values, c.rhs)
    ao, ac = o.lin_adjacency()
    # pinned joint 0: nodes 0 (H) and 1 (V), position coefficients 0 and 4 become adjacent although no element joins
    row0 = ac[ao[0]:ao[1]].tolist()
    assert 4 in row0 and 0 in ac[ao[4]:ao[5]].tolist()
    assert 4 not in o.m_col[o.m_off[0]:o.m_off[1]].tolist()
    import scipy.sparse as sp
    A = sp.csr_matrix((np.ones(len(ac)), ac, ao), shape=(o.N, o.N))
    assert (A != A.T).nnz == 0 and np.all(np.diff(ao) > 0)
    for i in range(o.N):
        assert np.all(np.diff(ac[ao[i]:ao[i + 1]]) > 0)              # sorted unique (SyncedNewton.cu:626-630)
