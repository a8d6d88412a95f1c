"""Synthetic workloads of BASELINE.json's configs: structured T10 boxes fed through the
same Setup API as a TetGen mesh.  Pure NumPy host code shared by bench.py, the tests and the drivers."""
import numpy as np

from . import mesh_utils

CONFIGS = {
    # name: cells, box size, material, load
    "B": dict(cells=(12, 12, 12), size=(1.0, 1.0, 1.0), material="neo", desc="T10 cube 12^3x6, neo-Hookean"),
    "C": dict(cells=(90, 60, 30), size=(3.0, 2.0, 1.0), material="svk", desc="T10 bar 90x60x30x6, SVK"),
    "S": dict(cells=(4, 3, 2), size=(3.0, 2.0, 1.0), material="svk", desc="small bar (tests)"),
    # two config-C slabs end to end on ONE GPU: the single-GPU reference of the 2-rank rehearsal
    "C2": dict(cells=(180, 60, 30), size=(6.0, 2.0, 1.0), material="svk", desc="T10 bar 180x60x30x6, SVK"),
    "M": dict(cells=(24, 16, 8), size=(3.0, 2.0, 1.0), material="svk", desc="medium bar (rehearsals)"),
    "M2": dict(cells=(45, 30, 15), size=(3.0, 2.0, 1.0), material="svk", desc="T10 bar 45x30x15x6, SVK"),
    "M3": dict(cells=(60, 40, 20), size=(3.0, 2.0, 1.0), material="svk", desc="T10 bar 60x40x20x6, SVK"),
    "C4": dict(cells=(360, 60, 30), size=(12.0, 2.0, 1.0), material="svk", desc="T10 bar 360x60x30x6, SVK"),
    # ANCF configs (element counts instead of cells)
    "A": dict(kind=3243, n=(30,), dims=(0.5, 0.1, 0.1), material="svk_damped", desc="ANCF-3243 cantilever, 30 beams"),
    "D": dict(kind=3443, n=(512, 500), dims=(0.1, 0.1, 0.01), material="svk", desc="ANCF-3443 plate 512x500"),
    "Ds": dict(kind=3443, n=(16, 12), dims=(0.1, 0.1, 0.01), material="svk", desc="small ANCF-3443 plate (tests)"),
}


def material(name):
    if name == "neo":  # test_feat10_bunny_newton.cc:26-28,121-126: E=3e8 nu=0.4 rho=920, MR(mu/2, 0, K)
        E, nu = 3.0e8, 0.4
        mu, K = E / (2 * (1 + nu)), E / (3 * (1 - 2 * nu))
        return dict(kind="mr", mu10=mu / 2, mu01=0.0, kappa=K, rho0=920.0, eta=0.0, lamd=0.0)
    if name == "svk":  # test_feat10_resolution.cc:40-42
        return dict(kind="svk", E=7e8, nu=0.33, rho0=2700.0, eta=0.0, lamd=0.0)
    if name == "svk_damped":  # test_ancf3243.cc:36-38,287-291
        return dict(kind="svk", E=7e8, nu=0.33, rho0=2700.0, eta=1e5, lamd=1e5)
    raise KeyError(name)


def build(config, cells=None, x_offset_cells=0, offset_cells=None):
    """-> dict(X, conn, fixed, f_ext, x0, material, params).  `cells` / `offset_cells` (or `x_offset_cells`) let a rank
    build its own block of a larger body with the config's cell size (weak scaling): the block is shifted so that blocks
    share their interface planes."""
    cfg = CONFIGS[config]
    if "kind" in cfg:
        return build_ancf(config)
    nx, ny, nz = cells or cfg["cells"]
    lx, ly, lz = cfg["size"]
    full_nx, full_ny, full_nz = cfg["cells"]
    ox, oy, oz = offset_cells if offset_cells is not None else (x_offset_cells, 0, 0)
    X, conn = mesh_utils.structured_t10_box(nx, ny, nz, lx * nx / full_nx, ly * ny / full_ny, lz * nz / full_nz)
    X[:, 0] += lx * ox / full_nx
    X[:, 1] += ly * oy / full_ny
    X[:, 2] += lz * oz / full_nz
    import os
    mat = material(os.environ.get("TLFEA_BENCH_MATERIAL", cfg["material"]))  # A/B runs: e.g. neo-Hookean at config C's size
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


def build_ancf(config):
    """ANCF workloads: config A = lib_bin/beam_sag/test_ancf3243.cc cantilever; config D = 3443 plate, one edge
    clamped (all 4 coefficient vectors of the x=0 nodes), uniform -z line load on the opposite edge."""
    cfg = CONFIGS[config]
    mat = material(cfg["material"])
    L, W, H = cfg["dims"]
    if cfg["kind"] == 3243:
        gen = mesh_utils.GridMeshGenerator(cfg["n"][0] * L, 0.0, L, True, False)
        gen.generate_mesh()
        x, y, z = gen.get_coordinates()
        conn = gen.get_element_connectivity()
        fixed = np.arange(4, dtype=np.int32)
        f_ext = np.zeros(3 * len(x))
        f_ext[(conn[-1, 1] * 4) * 3 + 2] = 3100.0
        params = (1e-4, 0.0, 1e-6, 1e14, 5, 10, 1e-3)  # test_ancf3243.cc:329
    else:
        nx, ny = cfg["n"]
        x, y, z, conn = mesh_utils.structured_3443_plate(nx, ny, L, W)
        edge = np.where(np.abs(x[0::4]) < 1e-12)[0]
        fixed = (4 * edge[:, None] + np.arange(4)[None, :]).reshape(-1).astype(np.int32)
        tip = np.where(np.abs(x[0::4] - nx * L) < 1e-9)[0]
        f_ext = np.zeros(3 * len(x))
        f_ext[(4 * tip) * 3 + 2] = -50.0 / len(tip)
        params = (1e-4, 0.0, 1e-6, 1e14, 5, 10, 1e-3)  # test_ancf3443.cc:357
    rng = np.random.default_rng(12345)
    x0 = np.stack([x, y, z], axis=1)
    pert = 1e-4 * min(L, W) * rng.normal(size=x0.shape)
    pert[fixed] = 0.0
    return dict(kind=cfg["kind"], x12=x, y12=y, z12=z, conn=conn, dims=(L, W, H), fixed=fixed, f_ext=f_ext,
                x0=x0 + pert, X=x0, material=mat, params=params, desc=cfg["desc"])


def make_ancf_engine(tl, w, with_solver=True):
    """Reference call order of lib_bin/beam_sag/test_ancf3243.cc:242-349 / test_ancf3443.cc:240-357."""
    q = tl.quadrature
    m, (L, W, H) = w["material"], w["dims"]
    n_nodes = len(w["x12"]) // 4
    if w["kind"] == 3243:
        d = tl.GPU_ANCF3243_Data(n_nodes, w["conn"].shape[0])
    else:
        d = tl.GPU_ANCF3443_Data(n_nodes, w["conn"].shape[0])
    d.Initialize()
    d.SetNodalFixed(w["fixed"])
    d.SetExternalForce(w["f_ext"])
    if w["kind"] == 3243:
        d.Setup(L, W, H, q.gauss_xi_m_6, q.gauss_xi_3, q.gauss_eta_2, q.gauss_zeta_2, q.weight_xi_m_6, q.weight_xi_3,
                q.weight_eta_2, q.weight_zeta_2, w["x12"], w["y12"], w["z12"], w["conn"])
    else:
        d.Setup(L, W, H, q.gauss_xi_m_7, q.gauss_eta_m_7, q.gauss_zeta_m_3, q.gauss_xi_4, q.gauss_eta_4, q.gauss_zeta_3,
                q.weight_xi_m_7, q.weight_eta_m_7, q.weight_zeta_m_3, q.weight_xi_4, q.weight_eta_4, q.weight_zeta_3,
                w["x12"], w["y12"], w["z12"], w["conn"])
    d.SetDensity(m["rho0"])
    d.SetDamping(m["eta"], m["lamd"])
    d.SetSVK(m["E"], m["nu"])
    d.CalcDsDuPre()
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


def make_engine(tl, w, with_solver=True):
    if "kind" in w:
        return make_ancf_engine(tl, w, with_solver)
    return make_engine_t10(tl, w, with_solver)


def make_engine_t10(tl, w, with_solver=True):
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
