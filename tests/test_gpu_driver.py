"""The C++ facade (reference class names over the C-ABI) driven by the reference's own driver flow:
total-lagrangian-fea_amd/host/test_feat10_resolution must reproduce the oracle's node history."""
import os
import subprocess

import numpy as np
import pytest

from oracle import orc
from tests.helpers import MATERIALS, MESHES, fixed_x0, load_mesh, make_oracle, tl
from tests.test_gpu_parity import disp_err_ok

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "total-lagrangian-fea_amd", "host", "test_feat10_resolution")


@pytest.mark.gpu
def test_cpp_driver_matches_oracle(tmp_path):
    if not os.path.exists(DRIVER):
        subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    csv = tmp_path / "hist.csv"
    out = subprocess.run([DRIVER, "--res=2", "--steps=3", "--dt=1e-3", "--solver=newton", f"--mesh_dir={MESHES}", f"--csv_path={csv}"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = np.loadtxt(csv, delimiter=",", skiprows=1)
    assert open(csv).readline().strip() == "step,x_position"  # reference CSV schema
    X, conn = load_mesh("res2")
    fixed = fixed_x0(X)
    f_ext = np.zeros(3 * X.shape[0])
    face = np.where(np.abs(X[:, 0] - 3.0) < 1e-8)[0]
    f_ext[3 * face] = 5000.0 / len(face)
    o = make_oracle(X, conn, MATERIALS["svk"], fixed, f_ext)
    prm = orc.NewtonParams(1e-4, 1e-4, 1e-4, 1e14, 5, 10, 1e-3)
    for step in range(3):
        o.newton_step(prm, solver=0)
        ref = o.x[89]  # historical target node of res2 (test_feat10_resolution.cc:257)
        assert abs(rows[step, 1] - ref) <= 1e-10 * np.max(np.abs(o.x - X[:, 0])) + 8e-16 * abs(ref)


def test_cpp_facade_compiles_without_gpu():
    """Header-only facade + driver build with plain g++ against the C-ABI (no HIP headers needed)."""
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    assert os.path.exists(DRIVER)
    out = subprocess.run([DRIVER, "--bogus"], capture_output=True, text=True)
    assert out.returncode == 1 and "Unknown argument" in out.stderr


@pytest.mark.gpu
def test_cpp_ancf3243_cantilever_config_a(tmp_path):
    """BASELINE config A: lib_bin/beam_sag/test_ancf3243.cc flow (30 elements, L=0.5, W=H=0.1, tip Fz=3100 N,
    damping 1e5/1e5, params {1e-4,0,1e-6,1e14,5,10,1e-3}) through the C++ facade vs the oracle; CSV `step,tip_z`."""
    from tests.test_gpu_ancf import SVK_D, beam_problem
    drv = os.path.join(os.path.dirname(DRIVER), "test_ancf3243")
    if not os.path.exists(drv):
        subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    csv = tmp_path / "tip.csv"
    out = subprocess.run([drv, "--steps=4", "--dt=1e-3", f"--csv_path={csv}", f"--vtu={tmp_path}/vtu"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert open(csv).readline().strip() == "step,tip_z"
    # --vtu: every 20th step -> only step 0 here, one hexahedron per beam (test_ancf3243.cc:46-47,302-318)
    assert os.listdir(tmp_path / "vtu") == ["ancf3243_newton_000000.vtu"]
    assert 'NumberOfPoints="240" NumberOfCells="30"' in open(tmp_path / "vtu" / "ancf3243_newton_000000.vtu").read()
    rows = np.loadtxt(csv, delimiter=",", skiprows=1)
    kind, x, y, z, conn, (L, W, H), fixed, f_ext = beam_problem(30)
    o = orc.AncfOracle(kind, x, y, z, conn, L, W, H, orc.svk(7e8, 0.33, rho0=2700.0, eta=1e5, lamd=1e5), fixed, f_ext)
    o.calc_dsdu_pre()
    o.calc_mass()
    prm = orc.NewtonParams(1e-4, 0.0, 1e-6, 1e14, 5, 10, 1e-3)
    tip = conn[-1, 1] * 4
    for step in range(4):
        o.newton_step(prm)
        assert abs(rows[step, 1] - o.z[tip]) <= 1e-10 * abs(o.z[tip] - z[tip]) + 8e-16 * max(1.0, abs(o.z[tip]))


@pytest.mark.gpu
@pytest.mark.parametrize("joint", ["welded", "pinned"])
def test_cpp_ancf3243_net_driver(joint, tmp_path):
    """lib_bin/mesh_deform/test_ancf3243_net_newton.cc flow through the C++ facade (mesh reader, LinearConstraintBuilder,
    corner clamps as AddFixedDof rows, SetLinearConstraintsCSR) vs the oracle: centre deflection per step."""
    from tests.test_gpu_linear_constraints import make_pair, net_problem
    from tests.test_linear_constraints import NET_P, NET_W
    drv = os.path.join(os.path.dirname(DRIVER), "test_ancf3243_net_newton")
    if not os.path.exists(drv):
        subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    path = NET_W if joint == "welded" else NET_P
    out = subprocess.run([drv, f"--joint={joint}", "--steps=2", f"--mesh={path}", f"--vtu={tmp_path}/vtu"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert os.listdir(tmp_path / "vtu") == ["ancf3243_net_000000.vtu"]   # every 10th step
    lines = out.stdout.strip().splitlines()
    k = lines.index("step,centre_z,constraint_norm")
    rows = np.array([[float(v) for v in ln.split(",")] for ln in lines[k + 1:]])
    prob = net_problem(path)
    kind, m, (L, W, H), csr, f_ext, mk, prm = prob
    o = orc.AncfOracle(kind, m.x12, m.y12, m.z12, m.element_connectivity, L, W, H,
                       orc.svk(mk["E"], mk["nu"], rho0=mk["rho0"], eta=mk["eta"], lamd=mk["lamd"]), f_ext=f_ext)
    o.calc_dsdu_pre()
    o.calc_mass()
    o.set_linear_constraints(csr.offsets, csr.columns, csr.values, csr.rhs)
    centre = int(np.where(f_ext != 0)[0][0] // 3)
    for step in range(2):
        o.newton_step_lin(orc.NewtonParams(*prm))
        ref = o.z[centre]
        assert abs(rows[step, 1] - ref) <= 1e-10 * np.max(np.abs(o.z - m.z12)) + 8e-16 * max(1.0, abs(ref))
        assert rows[step, 2] < 1e-6


@pytest.mark.gpu
def test_cpp_ancf3243_driver_adamw(tmp_path):
    """`--solver=adamw` of the beam_sag driver (test_ancf3243.cc:373-389: zero damping, SyncedAdamWNocoopParams
    {2e-4,.9,.999,1e-8,1e-4,.998,1e-1,1e-6,1e14,5,500,dt,10,0}) through the C++ facade: runs, writes the CSV schema and
    the tip starts to move down-force-wise (tight parity of this solver is in tests/test_gpu_ancf.py)."""
    drv = os.path.join(os.path.dirname(DRIVER), "test_ancf3243")
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    csv = tmp_path / "tip.csv"
    out = subprocess.run([drv, "--solver=adamw", "--steps=2", "--n_elements=6", f"--csv_path={csv}"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert open(csv).readline().strip() == "step,tip_z"
    rows = np.loadtxt(csv, delimiter=",", skiprows=1)
    assert rows.shape == (2, 2) and rows[1, 1] > rows[0, 1] > 0.0      # tip force is +z (3100 N)
    out = subprocess.run([drv, "--solver=nesterov", "--steps=1", "--n_elements=6", f"--csv_path={csv}"],
                         capture_output=True, text=True, timeout=300)       # test_ancf3243.cc:350-366
    assert out.returncode == 0, out.stderr
    assert np.isfinite(np.loadtxt(csv, delimiter=",", skiprows=1)).all()
    bad = subprocess.run([drv, "--solver=vbd"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Invalid --solver" in bad.stderr


UTEST = os.path.join(ROOT, "total-lagrangian-fea_amd", "host", "utest_facade")


def test_cpp_facade_host_utilities_known_answers(tmp_path):
    """Host-only part of host/utest_facade.cc (lib_utest/utest_utils.cc KATs, B12 matrices, MeshManager, VTU exporters through the facade): runs
    without a GPU, after which the program stops with its "No HIP device" status when none is visible."""
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    out = subprocess.run([UTEST, f"--data_dir={MESHES}", f"--tmp_dir={tmp_path}"], capture_output=True, text=True, timeout=300)
    assert out.stdout.count("[ OK ]") >= 22 and "[FAIL]" not in out.stdout, out.stdout + out.stderr
    assert out.returncode in (0, 101)


@pytest.mark.gpu
def test_cpp_facade_reference_unit_tests(tmp_path):
    """lib_utest/utest_3243.cc mass-matrix known answers (2 and 3 beams vs the reference's CSV fixtures, 1e-4), the
    3443 strip flow of utest_sparse_mass.cc and the per-kind sizes of the Retrieve* members, all through the facade."""
    if not os.path.exists(UTEST):
        subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    out = subprocess.run([UTEST, f"--data_dir={MESHES}", f"--tmp_dir={tmp_path}", "--print_dsdu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "[FAIL]" not in out.stdout and out.stdout.count("[ OK ]") >= 44
    assert "=== Elem 0 Quadrature Point 11 detJ_ref=0.25 ===" in out.stdout and "Shape 7: " in out.stdout


@pytest.mark.gpu
def test_cpp_driver_vbd_matches_oracle(tmp_path):
    """--solver=vbd of lib_bin/beam_sag/test_feat10_resolution.cc:377-391 ({1e-4,1e-4,1e-4,1e14,5,500,dt,omega 1.8,
    1e-12,25,1}; Setup, SetParameters, InitializeColoring, InitializeMassDiagBlocks, InitializeFixedMap, Solve) vs the
    oracle's restatement: the node history of the CSV."""
    csv = tmp_path / "hist_vbd.csv"
    out = subprocess.run([DRIVER, "--res=2", "--steps=2", "--dt=1e-3", "--solver=vbd", f"--mesh_dir={MESHES}", f"--csv_path={csv}"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = np.loadtxt(csv, delimiter=",", skiprows=1)
    X, conn = load_mesh("res2")
    fixed = fixed_x0(X)
    f_ext = np.zeros(3 * X.shape[0])
    face = np.where(np.abs(X[:, 0] - 3.0) < 1e-8)[0]
    f_ext[3 * face] = 5000.0 / len(face)
    o = make_oracle(X, conn, MATERIALS["svk"], fixed, f_ext)
    o.vbd_coloring(1)
    prm = orc.VbdParams(1e-4, 1e-4, 1e-4, 1e14, 5, 500, 1e-3, 1.8, 1e-12, 25, 1)
    for step in range(2):
        o.vbd_step(prm)
        ref = o.x[89]
        assert abs(rows[step, 1] - ref) <= 1e-10 * np.max(np.abs(o.x - X[:, 0])) + 8e-16 * abs(ref)


@pytest.mark.gpu
def test_cpp_driver_default_solver_is_adamw(tmp_path):
    """The reference driver's default --solver is adamw (test_feat10_resolution.cc:47) with the res-dependent
    parameters of :394-416; --omega is validated like there."""
    csv = tmp_path / "hist_adamw.csv"
    out = subprocess.run([DRIVER, "--res=2", "--steps=1", "--dt=1e-3", f"--mesh_dir={MESHES}", f"--csv_path={csv}"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = np.loadtxt(csv, delimiter=",", skiprows=1).reshape(-1, 2)
    assert rows.shape == (1, 2) and np.isfinite(rows[0, 1]) and abs(rows[0, 1] - 3.0) < 1e-3 and rows[0, 1] != 3.0
    bad = subprocess.run([DRIVER, "--solver=vbd", "--omega=-1"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Invalid --omega" in bad.stderr


@pytest.mark.gpu
def test_cpp_ancf3443_airless_tire_driver(tmp_path):
    """lib_bin/mesh_deform/test_ancf3443_mesh_newton.cc on the facade (host/test_ancf3443_mesh_newton.cc): mesh file with
    its own weld constraints, hub driven through UpdateLinearConstraintRHS, ground contact on the ring, VTU export.  The
    CSV it records is compared with the same loop run on the oracle (contact stiffness reduced with --load_fz so that
    the test's Newton iterations converge)."""
    from tests.test_gpu_linear_constraints import make_pair, tire_drive_inputs, tire_problem
    from tests.test_linear_constraints import TIRE
    drv = os.path.join(os.path.dirname(DRIVER), "test_ancf3443_mesh_newton")
    if not os.path.exists(drv):
        subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    csv = tmp_path / "tire.csv"
    out = subprocess.run([drv, f"--mesh={TIRE}", "--steps=2", "--dt=1e-3", "--load_fz=100", f"--csv_path={csv}",
                          f"--vtu={tmp_path}/vtu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Hub prescribed rotation: coef_fixed=" in out.stdout and "Constraint residual: rows=" in out.stdout
    assert os.listdir(tmp_path / "vtu") == ["ancf3443_mesh_000000.vtu"]
    rows = np.loadtxt(csv, delimiter=",", skiprows=1).reshape(-1, 6)
    prob = tire_problem()
    kind, m, dims, csrc, _, mk, prm = prob
    o, d = make_pair(prob)
    d.Destroy()
    hub_row0 = len(m.constraints.rhs)
    r = np.hypot(m.x12[0::4], m.z12[0::4])
    spoke = np.array([f == "S" for f in m.node_family])
    hub = np.where(spoke & (r < r[spoke].min() + 1e-9))[0]
    hub_coefs = [4 * int(n) + s_ for n in hub for s_ in range(4)]
    ring = np.array([4 * n for n in range(m.n_nodes) if m.node_family[n] == "R"])
    theta = 0.0
    for step in range(2):
        f, rhs, theta = tire_drive_inputs(m, csrc, hub_row0, hub_coefs, o.z, step, theta, 1e-3, ground_z=-0.2, k=100.0)
        o.f_ext[:] = f
        o.j_rhs[:] = rhs
        o.newton_step_lin(orc.NewtonParams(*prm))
        assert abs(rows[step, 1] - theta) <= 1e-15 + 1e-12 * theta and rows[step, 2] == np.count_nonzero(f)
        scale = np.abs(o.z[ring] - m.z12[ring]).max() + np.abs(o.x[hub_coefs[0]] - m.x12[hub_coefs[0]])
        # north_star's bar: 1e-10 of the displacement, plus the representation floor of the coordinates (8 ulp)
        assert abs(rows[step, 3] - o.x[hub_coefs[0]]) <= 1e-10 * scale + 8 * np.finfo(float).eps * np.abs(o.x).max()
        assert abs(rows[step, 5] - o.z[ring].min()) <= 1e-10 * scale + 8 * np.finfo(float).eps * np.abs(o.z).max()
    bad = subprocess.run([drv, "--steps=1"], capture_output=True, text=True)
    assert bad.returncode == 2 and "--mesh is required" in bad.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("material", ["svk", "mr"])
def test_cpp_feat10_bunny_newton_driver(tmp_path, material):
    """lib_bin/mesh_deform/test_feat10_bunny_newton.cc on the facade (0-based TetGen mesh, z < -4 pinned, -35 kN on the
    nodes with z > 4, Newton {1e-4,1e-6,1e-4,1e14,5,10,1e-3}; the source of BASELINE config B's material): three steps
    with the load, one after its release, against the oracle; VTK frames as in the reference."""
    drv = os.path.join(os.path.dirname(DRIVER), "test_feat10_bunny_newton")
    if not os.path.exists(drv):
        subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    csv = tmp_path / "bunny.csv"
    out = subprocess.run([drv, f"--mesh_dir={MESHES}", "--steps=4", "--release_step=3", f"--material={material}",
                          f"--vtk_dir={tmp_path}/vtk", "--output_interval=2", f"--csv_path={csv}"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "External force reset to zero at step 3" in out.stdout
    assert sorted(os.listdir(tmp_path / "vtk")) == ["bunny_newton_step_0.vtk", "bunny_newton_step_1.vtk"]
    rows = np.loadtxt(csv, delimiter=",", skiprows=1)
    X, conn = load_mesh("bunny")
    fixed = np.where(X[:, 2] < -4.0)[0].astype(np.int32)
    f_ext = np.zeros(3 * X.shape[0])
    f_ext[3 * np.where(X[:, 2] > 4.0)[0] + 2] = -35000.0
    E, nu = 3.0e8, 0.40
    mu, K = E / (2 * (1 + nu)), E / (3 * (1 - 2 * nu))
    m = (dict(kind="svk", E=E, nu=nu, rho0=920.0, eta=0.0, lamd=0.0) if material == "svk" else
         dict(kind="mr", mu10=0.30 * mu, mu01=0.20 * mu, kappa=1.5 * K, rho0=920.0, eta=0.0, lamd=0.0))
    o = make_oracle(X, conn, m, fixed, f_ext)
    top = int(np.argmax(X[:, 2]))
    prm = orc.NewtonParams(1e-4, 1e-6, 1e-4, 1e14, 5, 10, 1e-3)
    for step in range(4):
        if step == 3:
            o.f_ext[:] = 0.0
        o.newton_step(prm, solver=0)
        disp = np.sqrt((o.x - X[:, 0]) ** 2 + (o.y - X[:, 1]) ** 2 + (o.z - X[:, 2]) ** 2).max()
        assert abs(rows[step, 1] - o.z[top]) <= 1e-10 * disp + 8 * np.finfo(float).eps * abs(o.z[top])
        assert abs(rows[step, 2] - disp) <= 1e-10 * disp + 8 * np.finfo(float).eps * np.abs(X).max()


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["newton", "vbd", "nesterov"])
def test_cpp_ancf3443_strip_driver(tmp_path, solver):
    """lib_bin/beam_sag/test_ancf3443.cc on the facade: strip constructor, left edge pinned, tip load split by --lrratio,
    the reference's parameters per solver kind; CSV `step,tip_z` (mean z of the two tip nodes) against the oracle."""
    from tests.test_gpu_ancf import SVK, make_pair
    drv = os.path.join(os.path.dirname(DRIVER), "test_ancf3443")
    if not os.path.exists(drv):
        subprocess.check_call(["make", "-C", os.path.dirname(DRIVER)])
    csv = tmp_path / "tip.csv"
    out = subprocess.run([drv, f"--solver={solver}", "--n_beam=3", "--steps=2", "--dt=1e-3", "--lrratio=0.25",
                          f"--csv={csv}", f"--vtu={tmp_path}/vtu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert open(csv).readline().strip() == "step,tip_z"
    assert os.listdir(tmp_path / "vtu") == [f"ancf3443_{solver}_000000.vtu"]
    rows = np.loadtxt(csv, delimiter=",", skiprows=1)
    x, y, z, conn = tl.mesh_utils.ANCF3443_generate_beam_coordinates(3)
    fixed = np.array([4 * n + d_ for n in (conn[0, 0], conn[0, 3]) for d_ in range(4)], dtype=np.int32)
    ta, tb = int(conn[-1, 1]), int(conn[-1, 2])
    neg, pos = (ta, tb) if y[4 * ta] <= y[4 * tb] else (tb, ta)
    f_ext = np.zeros(3 * len(x))
    f_ext[(4 * neg) * 3 + 2] += 0.25 * -100.0
    f_ext[(4 * pos) * 3 + 2] += 0.75 * -100.0
    o, d = make_pair((3443, x, y, z, conn, (2.0, 1.0, 0.1), fixed, f_ext), SVK)
    d.Destroy()
    if solver == "vbd":
        o.vbd_coloring(1)
    for step in range(2):
        if solver == "newton":
            o.newton_step(orc.NewtonParams(1e-4, 0.0, 1e-6, 1e14, 5, 10, 1e-3))
        elif solver == "vbd":
            o.vbd_step(orc.VbdParams(1e-4, 1e-4, 1e-4, 1e14, 5, 500, 1e-3, 1.8, 1e-12, 25, 1))
        else:
            o.nesterov_step(orc.NesterovParams(1.0e-8, 1e14, 1.0e-6, 1.0e-6, 5, 300, 1e-3))
        ref = 0.5 * (o.z[4 * ta] + o.z[4 * tb])
        disp = np.abs(o.z - z).max()
        assert disp > 0 and abs(rows[step, 1] - ref) <= 1e-10 * disp + 8 * np.finfo(float).eps * np.abs(o.z).max()


@pytest.mark.gpu
def test_cpp_other_solver_kinds_run(tmp_path):
    """--solver=adamw of the bunny driver (lib_bin/mesh_deform/test_feat10_bunny_adamw.cc: -2 N, AdamW lr 1e-8 ...) and
    --solver=nesterov of the resolution driver (parameters of lib_bin/beam_sag/test_feat10_nesterov.cc:181): the steps
    run, the structure moves in the direction of the load, output files carry the solver's name."""
    drv = os.path.join(os.path.dirname(DRIVER), "test_feat10_bunny_newton")
    csv = tmp_path / "b.csv"
    out = subprocess.run([drv, f"--mesh_dir={MESHES}", "--solver=adamw", "--steps=2", f"--vtk_dir={tmp_path}/vtk",
                          "--output_interval=1", f"--csv_path={csv}"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert sorted(os.listdir(tmp_path / "vtk")) == ["bunny_adamw_step_0.vtk", "bunny_adamw_step_1.vtk"]
    rows = np.loadtxt(csv, delimiter=",", skiprows=1)
    assert rows.shape == (2, 3) and np.all(np.isfinite(rows)) and rows[1, 2] > 0
    csv2 = tmp_path / "n.csv"
    out = subprocess.run([DRIVER, "--res=2", "--steps=2", "--dt=1e-3", "--solver=nesterov", f"--mesh_dir={MESHES}",
                          f"--csv_path={csv2}"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    r2 = np.loadtxt(csv2, delimiter=",", skiprows=1)
    X, _ = load_mesh("res2")
    assert r2.shape == (2, 2) and r2[1, 1] > X[89, 0] and r2[1, 1] - X[89, 0] < 1e-2   # pulled in +x, slightly
