"""Synthetic workloads of BASELINE.json's configs: structured T10 boxes fed through the
same Setup API as a TetGen mesh.  Pure NumPy host code shared by bench.py, the tests and the drivers."""
import numpy as np

from . import mesh_utils

CONFIGS = {
    # name: cells, box size, material, load
    "B": dict(cells=(12, 12, 12), size=(1.0, 1.0, 1.0), material="neo", desc="T10 cube 12^3x6, neo-Hookean"),
    "C": dict(cells=(90, 60, 30), size=(3.0, 2.0, 1.0), material="svk", desc="T10 bar 90x60x30x6, SVK"),
    "S": dict(cells=(4, 3, 2), size=(3.0, 2.0, 1.0), material="svk", desc="small bar (tests)"),
}


def material(name):
    if name == "neo":  # test_feat10_bunny_newton.cc:26-28,121-126: E=3e8 nu=0.4 rho=920, MR(mu/2, 0, K)
        E, nu = 3.0e8, 0.4
        mu, K = E / (2 * (1 + nu)), E / (3 * (1 - 2 * nu))
        return dict(kind="mr", mu10=mu / 2, mu01=0.0, kappa=K, rho0=920.0, eta=0.0, lamd=0.0)
    if name == "svk":  # test_feat10_resolution.cc:40-42
        return dict(kind="svk", E=7e8, nu=0.33, rho0=2700.0, eta=0.0, lamd=0.0)
    raise KeyError(name)


def build(config, cells=None, x_offset_cells=0):
    """-> dict(X, conn, fixed, f_ext, x0, material, params).  `cells`/`x_offset_cells` let a rank build its own
    x-slab of a longer bar (weak scaling): the slab is shifted so that slabs share their interface plane."""
    cfg = CONFIGS[config]
    nx, ny, nz = cells or cfg["cells"]
    lx, ly, lz = cfg["size"]
    full_nx = cfg["cells"][0]
    X, conn = mesh_utils.structured_t10_box(nx, ny, nz, lx * nx / full_nx, ly, lz)
    X[:, 0] += lx * x_offset_cells / full_nx
    mat = material(cfg["material"])
    n = X.shape[0]
    f_ext = np.zeros(3 * n)
    if config == "B":
        fixed = np.where(np.abs(X[:, 2]) < 1e-12)[0].astype(np.int32)        # z=0 face clamped
        top = np.where(np.abs(X[:, 2] - lz) < 1e-12)[0]
        f_ext[3 * top + 2] = -2000.0 / len(top)                              # -z traction on the top face
        params = (1e-4, 1e-6, 1e-4, 1e14, 5, 10, 1e-3)                       # test_feat10_bunny_newton.cc:201
    else:
        fixed = np.where(np.abs(X[:, 0]) < 1e-12)[0].astype(np.int32)        # x=0 face (resolution.cc:283-296)
        face = np.where(np.abs(X[:, 0] - lx) < 1e-9)[0]
        if len(face):
            f_ext[3 * face] = 5000.0 / len(face)                             # 5000 N over x=L (:298-312)
        params = (1e-4, 1e-4, 1e-4, 1e14, 5, 10, 1e-3)                       # :365
    # state for kernel timing: smooth field + seeded noise (sigma = 1e-4 * element size)
    hsz = min(lx / full_nx, ly / cfg["cells"][1], lz / cfg["cells"][2]) / 2
    u = 1e-2 * np.sin(np.pi * X / np.array([lx, ly, lz]))
    x0 = X + u + np.random.default_rng(12345).normal(0.0, 1e-4 * hsz, X.shape)
    x0[fixed] = X[fixed]
    return dict(X=X, conn=conn, fixed=fixed, f_ext=f_ext, x0=x0, material=mat, params=params, desc=cfg["desc"])


def make_engine(tl, w, with_solver=True):
    """Reference call order (test_feat10_resolution.cc:273-375) on the product path."""
    q = tl.quadrature
    X, conn, m = w["X"], w["conn"], w["material"]
    d = tl.GPU_FEAT10_Data(conn.shape[0], X.shape[0])
    d.Initialize()
    d.SetNodalFixed(w["fixed"])
    d.SetExternalForce(w["f_ext"])
    d.Setup(q.tet5pt_x, q.tet5pt_y, q.tet5pt_z, q.tet5pt_weights, X[:, 0], X[:, 1], X[:, 2], conn)
    d.SetDensity(m["rho0"])
    d.SetDamping(m["eta"], m["lamd"])
    if m["kind"] == "svk":
        d.SetSVK(m["E"], m["nu"])
    else:
        d.SetMooneyRivlin(m["mu10"], m["mu01"], m["kappa"])
    d.CalcDnDuPre()
    d.CalcMassMatrix()
    d.CalcConstraintData()
    d.ConvertToCSR_ConstraintJacT()
    d.BuildConstraintJacobianCSR()
    if not with_solver:
        return d, None
    s = tl.SyncedNewtonSolver(d, d.get_n_constraint())
    s.Setup()
    s.SetParameters(tl.SyncedNewtonParams(*w["params"]))
    s.AnalyzeHessianSparsity()
    s.SetFixedSparsityPattern(True)
    return d, s
