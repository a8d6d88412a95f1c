"""Pins the ANCF-3243 / ANCF-3443 oracle (oracle/tlfea_oracle_ancf.c) against the reference's NumPy prototypes
(tests/golden/ancf34*_proto.npz, tools/gen_golden_ancf.py), the reference's mass-matrix CSV fixtures
(lib_utest/utest_3243.cc:34-200, tolerance 1e-4 as there) and the strip / grid generator known answers
(lib_utest/utest_utils.cc:32-222); cross-pins the generic path against the pinned T10 oracle at S=10, Q=5.
The prototypes hold no tangent, so K_e of the ANCF types is pinned by finite differences of f_int + symmetry."""
import os

import numpy as np
import pytest

from oracle import orc
from tests.helpers import load_mesh, tl


def relerr(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def proto_oracle(kind, g, mat=None, **kw):
    if kind == 3243:
        n = int(g["n_elem"])
        conn = np.stack([np.arange(n), np.arange(n) + 1], axis=1)
    else:
        conn = g["element_connectivity"]
    mat = mat or orc.svk(float(g["E"]), float(g["nu"]), rho0=float(g["rho0"]))
    o = orc.AncfOracle(kind, g["x12"], g["y12"], g["z12"], conn, float(g["L"]), float(g["W"]), float(g["H"]), mat, **kw)
    o.calc_dsdu_pre()
    o.calc_mass()
    return o


@pytest.fixture(scope="module", params=[3243, 3443])
def proto(request, golden_dir):
    kind = request.param
    return kind, np.load(os.path.join(golden_dir, f"ancf{kind}_proto.npz"))


def test_B_inv_matches_prototype(proto):
    kind, g = proto
    o = proto_oracle(kind, g)
    S = o.S
    Binv = o.B_inv[0].reshape(S, S).T  # column-major storage -> matrix
    # prototype: s = B_inv @ b with B_inv = inv(B12) (f-form-3243-nesterov.py:184-187): same matrix
    assert relerr(Binv, g["B_inv"]) < 1e-12


def test_gradients_and_detJ_match_prototype(proto):
    kind, g = proto
    o = proto_oracle(kind, g)
    vol8 = float(g["L"]) * float(g["W"]) * float(g["H"]) / 8.0
    for e in range(o.E):  # straight reference elements: physical gradient == ds/du, detJ_xi = detJ_uvw * LWH/8
        assert relerr(o.gradN_a_d()[e], g["ds_du"]) < 1e-12
        assert relerr(o.detJ[e], g["detJ_uvw"] * vol8) < 1e-12


def test_mass_matches_prototype(proto):
    kind, g = proto
    o = proto_oracle(kind, g)
    assert relerr(o.mass_dense(), g["mass"]) < 1e-12


def test_internal_force_matches_prototype(proto):
    kind, g = proto
    o = proto_oracle(kind, g)
    o.x, o.y, o.z = g["xp"].copy(), g["yp"].copy(), g["zp"].copy()
    assert relerr(o.internal_force(), g["f_int"]) < 1e-12


@pytest.mark.parametrize("nb", [2, 3])
def test_3243_mass_reference_csv(nb, mesh_dir):
    """MassMatrix2Beams / MassMatrix3Beams (utest_3243.cc:34-200): L=2, W=H=1, rho=2700, tolerance 1e-4."""
    gen = tl.mesh_utils.GridMeshGenerator(nb * 2.0, 0.0, 2.0, True, False)
    gen.generate_mesh()
    x, y, z = gen.get_coordinates()
    o = orc.AncfOracle(3243, x, y, z, gen.get_element_connectivity(), 2.0, 1.0, 1.0, orc.svk(7e8, 0.33, rho0=2700.0))
    o.calc_dsdu_pre()
    o.calc_mass()
    ref = np.loadtxt(os.path.join(mesh_dir, f"mass_matrix_{nb}_beam.csv"), delimiter=",")
    M = o.mass_dense()
    assert M.shape == ref.shape and np.abs(M - ref).max() < 1e-4
    assert np.linalg.det(M) > 0 and np.abs(M - M.T).max() < 1e-4


def test_generator_known_answers():
    """utest_utils.cc:32-108 (offsets), :114-222 (3443 strip coordinates + connectivity)."""
    s, e = tl.mesh_utils.ANCF3243_calculate_offsets(3)
    assert s.tolist() == [0, 4, 8] and e.tolist() == [7, 11, 15]
    x, y, z, conn = tl.mesh_utils.ANCF3443_generate_beam_coordinates(3)
    assert conn.tolist() == [[0, 1, 2, 3], [1, 4, 5, 2], [4, 6, 7, 5]]
    assert len(x) == 32 and x[0::4].tolist() == [0, 2, 2, 0, 4, 4, 6, 6] and y[0::4].tolist() == [0, 0, 1, 1, 0, 1, 0, 1]
    assert np.all(x[1::4] == 1) and np.all(y[2::4] == 1) and np.all(z[3::4] == 1) and np.all(z[0::4] == 0)
    g = tl.mesh_utils.GridMeshGenerator(15.0, 0.0, 0.5, True, False)  # test_ancf3243.cc:242-246: 31 nodes / 30 elements
    g.generate_mesh()
    assert g.get_num_nodes() == 31 and g.get_num_elements() == 30 and len(g.get_coordinates()[0]) == 124
    with pytest.raises(ValueError):
        tl.mesh_utils.GridMeshGenerator(1.0, 0.0, 0.3)


def test_product_B12_matrix_matches_prototype(proto):
    """ANCF3243_B12_matrix / ANCF3443_B12_matrix of the product library (host arithmetic behind the C-ABI, no GPU call)
    against the B_inv of the reference's NumPy prototypes, and the flat per-element packing (cpu_utils.cc:190-209)."""
    kind, g = proto
    L, W, H = float(g["L"]), float(g["W"]), float(g["H"])
    f = tl.mesh_utils.ANCF3243_B12_matrix if kind == 3243 else tl.mesh_utils.ANCF3443_B12_matrix
    B = f(L, W, H)
    S = B.shape[0]
    assert relerr(B, g["B_inv"]) < 1e-12
    flat = (tl.mesh_utils.ANCF3243_B12_matrix_flat_per_element if kind == 3243
            else tl.mesh_utils.ANCF3443_B12_matrix_flat_per_element)([L, 2 * L], [W, W], [H, H])
    assert flat.shape == (2 * S * S,) and np.array_equal(flat[:S * S].reshape(S, S).T, B)
    assert relerr(flat[S * S:].reshape(S, S).T, f(2 * L, W, H)) == 0.0
    with pytest.raises(tl.binding.TlfeaError):
        tl.mesh_utils._b12(1234, L, W, H)


def test_3243_beam_chain_generator():
    """ANCF3243_generate_beam_coordinates (cpu_utils.cc:443-474): node n at x = -1 + 2 n, y = 1, gradients = identity."""
    x, y, z = tl.mesh_utils.ANCF3243_generate_beam_coordinates(3)
    assert x.tolist() == [-1, 1, 0, 0, 1, 1, 0, 0, 3, 1, 0, 0, 5, 1, 0, 0]
    assert y.tolist() == [1, 0, 1, 0] * 4 and z.tolist() == [0, 0, 0, 1] * 4


@pytest.mark.parametrize("kind", [3243, 3443])
@pytest.mark.parametrize("matname", ["svk", "mr"])
def test_tangent_is_derivative_of_force(kind, matname, golden_dir):
    g = np.load(os.path.join(golden_dir, f"ancf{kind}_proto.npz"))
    mat = (orc.svk(7e8, 0.33, rho0=2700.0) if matname == "svk" else orc.mooney_rivlin(4e7, 1e7, 5e8, rho0=920.0))
    o = proto_oracle(kind, g, mat)
    o.x, o.y, o.z = g["xp"].copy(), g["yp"].copy(), g["zp"].copy()
    Ke, _ = o.element_tangents()
    assert relerr(Ke, Ke.transpose(0, 2, 1)) < 1e-12
    # assembled K vs central differences of f_int
    n = 3 * o.N
    K = np.zeros((n, n))
    for e in range(o.E):
        dofs = (3 * o.conn[e][:, None] + np.arange(3)[None, :]).reshape(-1)
        K[np.ix_(dofs, dofs)] += Ke[e]
    arrs = (o.x, o.y, o.z)
    rng = np.random.default_rng(1)
    for dof in rng.choice(n, size=12, replace=False):
        a, c = arrs[dof % 3], dof // 3
        x0 = a[c]
        hstep = 1e-6
        a[c] = x0 + hstep
        fp = o.internal_force()
        a[c] = x0 - hstep
        fm = o.internal_force()
        a[c] = x0
        col = (fp - fm) / (2 * hstep)
        assert np.max(np.abs(col - K[:, dof])) < 2e-6 * np.abs(K).max()


def test_generic_path_equals_pinned_t10_oracle():
    X, conn = load_mesh("beam_3x2x1")
    mat = orc.svk(7e8, 0.33, rho0=2700.0, eta=1e5, lamd=1e5)
    o = orc.T10Oracle(X, conn, mat)
    o.calc_dndu_pre()
    o.calc_mass()
    gen = orc.ElemOracle(10, 5, X[:, 0], X[:, 1], X[:, 2], conn, o.qw, mat)
    gen.gradN, gen.detJ = o.gradN.copy(), o.detJ.copy()
    gen.mass_pattern()
    assert np.array_equal(gen.m_off, o.m_off) and np.array_equal(gen.m_col, o.m_col)
    rng = np.random.default_rng(3)
    x = X + rng.normal(0, 1e-3, X.shape)
    v = rng.normal(0, 0.1, 3 * X.shape[0])
    for obj in (o, gen):
        obj.x, obj.y, obj.z = (np.ascontiguousarray(x[:, i]) for i in range(3))
    assert relerr(gen.internal_force(v), o.internal_force(v)) < 1e-14
    Kg, Cg = gen.element_tangents(True)
    Kt, Ct = o.element_tangents(True)
    assert relerr(Kg, Kt) < 1e-13 and relerr(Cg, Ct) < 1e-13
    gen.m_val = o.m_val.copy()
    fixed = np.array([0, 5], dtype=np.int32)
    o.fixed = gen.fixed = fixed
    _, _, vt = o.assemble_hessian(1e-3, 1e12)
    _, _, vg = gen.assemble_hessian(1e-3, 1e12)
    assert relerr(vg, vt) < 1e-13
