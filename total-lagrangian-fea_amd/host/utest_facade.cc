// Known-answer checks of the reference's unit tests, re-expressed on the C++ facade (gtest is not part of the image,
// so this is a plain program: prints one line per check, exit status = number of failed checks).
//   lib_utest/utest_3243.cc:34-200    3243 consistent mass of 2 / 3 beams == data/utest/mass_matrix_{2,3}_beam.csv (1e-4)
//   lib_utest/utest_utils.cc:32-222   ANCF3243_calculate_offsets pattern, 3443 strip coordinates + connectivity
//   lib_utils/mesh_manager.cc         MeshManager unified numbering, transforms, scalar fields (host) + two-body step
//   lib_utest/utest_sparse_mass.cc    3443 strip: Setup -> CalcDsDuPre -> mass -> constraints -> CalcP (smoke + shapes)
// Usage: utest_facade --data_dir=<dir holding mass_matrix_2_beam.csv, mass_matrix_3_beam.csv>
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "tlfea_facade.h"

namespace {
int n_failed = 0;
void check(bool ok, const std::string& what) {
  std::cout << (ok ? "[ OK ] " : "[FAIL] ") << what << std::endl;
  if (!ok) n_failed++;
}

bool load_csv(const std::string& path, std::vector<std::vector<double>>& rows) {
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty()) continue;
    std::vector<double> r;
    std::stringstream ss(line);
    std::string cell;
    while (std::getline(ss, cell, ',')) r.push_back(std::atof(cell.c_str()));
    rows.push_back(r);
  }
  return !rows.empty();
}

void mass_known_answer(int n_beams, const std::string& data_dir) {
  const double L = 2.0, W = 1.0, H = 1.0;
  ANCFCPUUtils::GridMeshGenerator grid_gen(n_beams * L, 0.0, L, true, false);
  grid_gen.generate_mesh();
  GPU_ANCF3243_Data data(grid_gen.get_num_nodes(), grid_gen.get_num_elements());
  data.Initialize();
  tlfea::VectorXd x, y, z;
  grid_gen.get_coordinates(x, y, z);
  tlfea::MatrixXi conn;
  grid_gen.get_element_connectivity(conn);
  data.Setup(L, W, H, Quadrature::gauss_xi_m_6, Quadrature::gauss_xi_3, Quadrature::gauss_eta_2,
             Quadrature::gauss_zeta_2, Quadrature::weight_xi_m_6, Quadrature::weight_xi_3, Quadrature::weight_eta_2,
             Quadrature::weight_zeta_2, x, y, z, conn);
  data.SetDensity(2700.0);
  data.SetDamping(0.0, 0.0);
  data.SetSVK(7e8, 0.33);
  data.CalcDsDuPre();
  data.CalcMassMatrix();
  std::vector<std::vector<double>> ref;
  const std::string tag = "3243 mass matrix, " + std::to_string(n_beams) + " beams";
  if (!load_csv(data_dir + "/mass_matrix_" + std::to_string(n_beams) + "_beam.csv", ref)) {
    check(false, tag + ": fixture not found under " + data_dir);
    data.Destroy();
    return;
  }
  std::vector<int> off, col;
  std::vector<double> val;
  data.RetrieveMassCSRToCPU(off, col, val);
  const int n = data.get_n_coef();
  check((int)ref.size() == n && (int)ref[0].size() == n, tag + ": shape " + std::to_string(n) + " x " + std::to_string(n));
  std::vector<double> M((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++)
    for (int p = off[i]; p < off[i + 1]; p++) M[(size_t)i * n + col[p]] = val[p];
  double worst = 0.0, asym = 0.0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      worst = std::max(worst, std::fabs(M[(size_t)i * n + j] - ref[i][j]));
      asym = std::max(asym, std::fabs(M[(size_t)i * n + j] - M[(size_t)j * n + i]));
    }
  check(worst < 1e-4, tag + ": max |M - fixture| = " + std::to_string(worst) + " < 1e-4");
  check(asym < 1e-4, tag + ": symmetric");
  // positive definite <=> Cholesky succeeds (the reference checks det > 0)
  bool spd = true;
  std::vector<double> Lc(M);
  for (int j = 0; j < n && spd; j++) {
    double d = Lc[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= Lc[(size_t)j * n + k] * Lc[(size_t)j * n + k];
    if (d <= 0) spd = false;
    d = std::sqrt(d);
    Lc[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = Lc[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= Lc[(size_t)i * n + k] * Lc[(size_t)j * n + k];
      Lc[(size_t)i * n + j] = s / d;
    }
  }
  check(spd, tag + ": positive definite");
  // node-level connectivity round trip and quadrature-sized retrievals (12 force points per beam)
  tlfea::MatrixXi back;
  data.RetrieveConnectivityToCPU(back);
  bool same = back.rows() == conn.rows() && back.cols() == 2;
  for (int e = 0; same && e < conn.rows(); e++) same = back(e, 0) == conn(e, 0) && back(e, 1) == conn(e, 1);
  check(same, tag + ": RetrieveConnectivityToCPU returns the node pairs");
  std::vector<std::vector<double>> dj;
  data.RetrieveDetJToCPU(dj);
  bool djok = (int)dj.size() == n_beams && (int)dj[0].size() == Quadrature::N_TOTAL_QP_3_2_2;
  for (auto& r : dj)
    for (double v : r) djok = djok && std::fabs(v - L * W * H / 8.0) < 1e-12;
  check(djok, tag + ": detJ = L W H / 8 at all 12 force points");
  data.Destroy();
}

void utils_known_answers() {
  {  // quadrature_utils.h:134-171: Keast rule (bary rows sum to 1, columns 1..3 = the points, weights sum to 1/6)
    const tlfea::MatrixXd& b = Quadrature::Tet5ptQuadrature::barycentric();
    double wsum = 0.0;
    bool okq = b.rows() == 5 && b.cols() == 4 && Quadrature::Tet5ptQuadrature::n_points == 5;
    for (int q = 0; okq && q < 5; q++) {
      okq = std::fabs(b(q, 0) + b(q, 1) + b(q, 2) + b(q, 3) - 1.0) < 1e-15 && b(q, 1) == Quadrature::tet5pt_x(q) &&
            Quadrature::tet5pt_xyz(q, 2) == Quadrature::tet5pt_z(q);
      wsum += Quadrature::Tet5ptQuadrature::weights()(q);
    }
    check(okq && std::fabs(wsum - 1.0 / 6.0) < 1e-15 && Quadrature::tet5pt_weights(0) < 0.0 && Quadrature::N_TOTAL_QP_7_7_3 == 147,
          "Quadrature: 5-point Keast rule (negative first weight), size constants");
  }
  tlfea::VectorXi s, e;
  ANCFCPUUtils::ANCF3243_calculate_offsets(5, s, e);
  bool ok = s.size() == 5 && e.size() == 5;
  for (int i = 0; ok && i < 5; i++) ok = s(i) == 4 * i && e(i) == 4 * i + 7;
  check(ok, "ANCF3243_calculate_offsets: start = 4 i, end = start + 7");
  tlfea::VectorXd x(16), y(16), z(16);
  ANCFCPUUtils::ANCF3243_generate_beam_coordinates(3, x, y, z);
  check(x(0) == -1 && x(4) == 1 && x(8) == 3 && x(12) == 5 && x(1) == 1 && y(2) == 1 && z(3) == 1 && y(0) == 1,
        "ANCF3243_generate_beam_coordinates: nodes at x = -1, 1, 3, 5");
  tlfea::MatrixXd B;
  ANCFCPUUtils::ANCF3243_B12_matrix(2.0, 1.0, 1.0, B, Quadrature::N_SHAPE_3243);
  // s = B_inv b: shape function 0 (position of node 0) is 1 at u = -L/2 and 0 at u = +L/2
  auto shape0 = [&](double u) {
    const double b[8] = {1, u, 0, 0, 0, 0, u * u, u * u * u};
    double v = 0;
    for (int k = 0; k < 8; k++) v += B(0, k) * b[k];
    return v;
  };
  check(B.rows() == 8 && std::fabs(shape0(-1.0) - 1.0) < 1e-13 && std::fabs(shape0(1.0)) < 1e-13,
        "ANCF3243_B12_matrix: shape function 0 interpolates node 0");
  tlfea::VectorXd Lv(2), Wv(2), Hv(2), flat;
  Lv(0) = 2.0; Lv(1) = 0.5; Wv(0) = Wv(1) = 1.0; Hv(0) = Hv(1) = 0.1;
  ANCFCPUUtils::ANCF3443_B12_matrix_flat_per_element(Lv, Wv, Hv, flat, Quadrature::N_SHAPE_3443);
  tlfea::MatrixXd B2;
  ANCFCPUUtils::ANCF3443_B12_matrix(0.5, 1.0, 0.1, B2, Quadrature::N_SHAPE_3443);
  bool fl = flat.size() == 2 * 256;
  for (int k = 0; fl && k < 256; k++) fl = flat(256 + k) == B2.data()[k];
  check(fl, "ANCF3443_B12_matrix_flat_per_element: column-major block per element");
}

void strip_3443() {
  const int n_beam = 3;
  tlfea::VectorXd x, y, z;
  tlfea::MatrixXi conn;
  ANCFCPUUtils::ANCF3443_generate_beam_coordinates(n_beam, x, y, z, conn);
  const int want[3][4] = {{0, 1, 2, 3}, {1, 4, 5, 2}, {4, 6, 7, 5}};
  bool ok = conn.rows() == 3 && conn.cols() == 4 && x.size() == 32;
  for (int e = 0; ok && e < 3; e++)
    for (int k = 0; k < 4; k++) ok = ok && conn(e, k) == want[e][k];
  check(ok, "ANCF3443_generate_beam_coordinates(3): 8 nodes, connectivity [[0,1,2,3],[1,4,5,2],[4,6,7,5]]");
  GPU_ANCF3443_Data data(n_beam);  // strip constructor
  check(data.get_n_coef() == 32 && data.get_n_beam() == 3, "GPU_ANCF3443_Data(int n_beam): n_coef = 4 (4 + 2 (n_beam - 1))");
  data.Initialize();
  tlfea::VectorXi fixed(8);
  for (int i = 0; i < 4; i++) {
    fixed(i) = i;           // node 0
    fixed(4 + i) = 12 + i;  // node 3
  }
  data.SetNodalFixed(fixed);
  tlfea::VectorXd f(3 * data.get_n_coef());
  data.SetExternalForce(f);
  data.Setup(2.0, 1.0, 0.1, Quadrature::gauss_xi_m_7, Quadrature::gauss_eta_m_7, Quadrature::gauss_zeta_m_3,
             Quadrature::gauss_xi_4, Quadrature::gauss_eta_4, Quadrature::gauss_zeta_3, Quadrature::weight_xi_m_7,
             Quadrature::weight_eta_m_7, Quadrature::weight_zeta_m_3, Quadrature::weight_xi_4, Quadrature::weight_eta_4,
             Quadrature::weight_zeta_3, x, y, z, conn);
  data.SetDensity(2700.0);
  data.SetDamping(0.0, 0.0);
  data.SetSVK(7e8, 0.33);
  data.CalcDsDuPre();
  data.CalcMassMatrix();
  data.CalcConstraintData();
  data.ConvertToCSR_ConstraintJacT();
  data.BuildConstraintJacobianCSR();
  data.CalcP();
  data.CalcInternalForce();
  tlfea::MatrixXi back;
  data.RetrieveConnectivityToCPU(back);
  bool same = back.rows() == 3 && back.cols() == 4;
  for (int e = 0; same && e < 3; e++)
    for (int k = 0; k < 4; k++) same = same && back(e, k) == want[e][k];
  check(same, "3443 strip: RetrieveConnectivityToCPU returns node ids");
  std::vector<std::vector<tlfea::MatrixXd>> F, P, g;
  data.RetrieveDeformationGradientToCPU(F);
  data.RetrievePFromFToCPU(P);
  data.RetrieveDnDuPreToCPU(g);
  bool shapes = F.size() == 3 && (int)F[0].size() == Quadrature::N_TOTAL_QP_4_4_3 && P.size() == 3 &&
                (int)P[2].size() == Quadrature::N_TOTAL_QP_4_4_3 && g[0][0].rows() == 16 && g[0][0].cols() == 3;
  check(shapes, "3443 strip: F, P per (element, 48 force points), gradients 16 x 3");
  double dev = 0.0, pmax = 0.0;
  for (auto& fe : F)
    for (auto& fq : fe)
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) dev = std::max(dev, std::fabs(fq(i, j) - (i == j ? 1.0 : 0.0)));
  for (auto& pe : P)
    for (auto& pq : pe)
      for (int i = 0; i < 9; i++) pmax = std::max(pmax, std::fabs(pq.data()[i]));
  check(dev < 1e-12, "3443 strip: F = I in the reference configuration (max dev " + std::to_string(dev) + ")");
  check(pmax < 1e-3, "3443 strip: P = 0 in the reference configuration");
  tlfea::VectorXd fi;
  data.RetrieveInternalForceToCPU(fi);
  double fmax = 0.0;
  for (int i = 0; i < fi.size(); i++) fmax = std::max(fmax, std::fabs(fi(i)));
  check(fi.size() == 96 && fmax < 1e-3, "3443 strip: internal force vanishes in the reference configuration");
  tlfea::VectorXd c;
  data.RetrieveConstraintDataToCPU(c);
  double cmax = 0.0;
  for (int i = 0; i < c.size(); i++) cmax = std::max(cmax, std::fabs(c(i)));
  check(c.size() == 24 && cmax == 0.0, "3443 strip: 24 satisfied constraints (8 fixed coefficients)");
  tlfea::MatrixXd Jd;
  data.RetrieveConstraintJacobianToCPU(Jd);
  bool jd = Jd.rows() == 24 && Jd.cols() == 96;
  double jsum = 0.0;
  for (int r = 0; jd && r < 24; r++) {
    for (int cc = 0; cc < 96; cc++) jsum += Jd(r, cc);
    jd = Jd(r, 3 * fixed(r / 3) + r % 3) == 1.0;
  }
  check(jd && jsum == 24.0, "3443 strip: dense constraint Jacobian = selector rows of the pinned coefficients");
  data.Destroy();
}
// visualization_utils.h:491-1097: the VTU files the drivers write (host only): point / cell counts and corner geometry
std::vector<double> vtu_points(const std::string& path, int* n_points, int* n_cells) {
  std::ifstream f(path);
  std::string line;
  std::vector<double> pts;
  bool in_points = false;
  *n_points = *n_cells = -1;
  while (std::getline(f, line)) {
    const size_t a = line.find("NumberOfPoints=\"");
    if (a != std::string::npos) {
      *n_points = std::atoi(line.c_str() + a + 16);
      *n_cells = std::atoi(line.c_str() + line.find("NumberOfCells=\"") + 15);
    }
    if (line.find("<Points>") != std::string::npos) in_points = true;
    else if (line.find("</Points>") != std::string::npos) in_points = false;
    else if (in_points && line.find('<') == std::string::npos) {
      std::stringstream ss(line);
      double v;
      while (ss >> v) pts.push_back(v);
    }
  }
  return pts;
}

void vtu_known_answers(const std::string& tmp) {
  // one straight beam from (0,0,0) to (2,0,0), width 0.2 (along t x z = -y), height 0.1 (along t x n = -z... sign below)
  tlfea::VectorXd x(8), y(8), z(8);
  x(4) = 2.0;
  tlfea::MatrixXi conn(1, 2);
  conn(0, 1) = 1;
  const std::string fb = tmp + "/utest_beam.vtu";
  int np = 0, nc = 0;
  bool ok = ANCFCPUUtils::VisualizationUtils::ExportANCF3243ToVTU(x, y, z, conn, 0.2, 0.1, fb);
  std::vector<double> p = vtu_points(fb, &np, &nc);
  ok = ok && np == 8 && nc == 1 && p.size() == 24;
  // t = x, n = t x z = -y, b = t x n = -z: corner 0 = p0 - W/2 n - H/2 b = (0, +0.1, +0.05)
  ok = ok && std::fabs(p[0]) < 1e-15 && std::fabs(p[1] - 0.1) < 1e-15 && std::fabs(p[2] - 0.05) < 1e-15 &&
       std::fabs(p[12] - 2.0) < 1e-15 && std::fabs(p[3 * 6 + 1] + 0.1) < 1e-15 && std::fabs(p[3 * 6 + 2] + 0.05) < 1e-15;
  check(ok, "ExportANCF3243ToVTU: 8 points / 1 hexahedron, cross-section frame of a beam along x");
  // one flat shell in the z = 0 plane, thickness 0.1: bottom face at z = -0.05 first, then top
  tlfea::VectorXd sx, sy, sz;
  tlfea::MatrixXi sc;
  ANCFCPUUtils::ANCF3443_generate_beam_coordinates(2, sx, sy, sz, sc);
  const std::string fs = tmp + "/utest_shell.vtu";
  ok = ANCFCPUUtils::VisualizationUtils::ExportANCF3443ToVTU(sx, sy, sz, sc, 0.1, fs);
  p = vtu_points(fs, &np, &nc);
  ok = ok && np == 16 && nc == 2 && p.size() == 48;
  for (int k = 0; ok && k < 4; k++) ok = std::fabs(p[3 * k + 2] + 0.05) < 1e-15 && std::fabs(p[3 * (4 + k) + 2] - 0.05) < 1e-15;
  ok = ok && p[3 * 1] == 2.0 && p[3 * 2 + 1] == 1.0 && p[3 * 9] == 4.0;  // element 1 = nodes (1,4,5,2): second corner x = 4
  check(ok, "ExportANCF3443ToVTU: 16 points / 2 hexahedra, bottom face then top face along the element normal");
  tlfea::MatrixXd nodes(5, 3);
  nodes(1, 0) = nodes(2, 1) = nodes(3, 2) = 1.0;
  nodes(4, 0) = 0.5;
  tlfea::MatrixXi tet(1, 10);
  for (int k = 0; k < 4; k++) tet(0, k) = k;
  tlfea::VectorXd disp(15), field(5);
  disp(3) = 3.0; disp(4) = 4.0;
  const std::string fd = tmp + "/utest_disp.vtu", fm = tmp + "/utest_mesh.vtu";
  ok = ANCFCPUUtils::VisualizationUtils::ExportMeshWithDisplacement(nodes, tet, disp, fd) &&
       ANCFCPUUtils::VisualizationUtils::ExportMeshToVTU(nodes, tet, field, fm);
  std::ifstream f(fd);
  std::stringstream all;
  all << f.rdbuf();
  const std::string txt = all.str();
  ok = ok && txt.find("Scalars=\"displacement_magnitude\" Vectors=\"displacement\"") != std::string::npos &&
       txt.find("5.000000000000000e+00") != std::string::npos && txt.find("          10\n") != std::string::npos;
  vtu_points(fm, &np, &nc);
  check(ok && np == 5 && nc == 1, "ExportMeshWithDisplacement / ExportMeshToVTU: corner tets, |(3,4,0)| = 5 at node 1");
}

// cpu_utils.cc:18-123 colouring helpers (host only) on the 36-element bar
void coloring_known_answers(const std::string& d) {
  tlfea::MatrixXi elems;
  tlfea::MatrixXd nodes;
  const int n = ANCFCPUUtils::FEAT10_read_nodes(d + "/beam_3x2x1.1.node", nodes);
  ANCFCPUUtils::FEAT10_read_elements(d + "/beam_3x2x1.1.ele", elems);
  auto adj = ANCFCPUUtils::BuildVertexAdjacency(elems, n);
  tlfea::VectorXi colors = ANCFCPUUtils::GreedyVertexColoring(adj);
  int n_colors = 0;
  for (int i = 0; i < n; i++) n_colors = std::max(n_colors, colors(i) + 1);
  bool proper = true;
  for (int i = 0; i < n; i++)
    for (int nb : adj[i]) proper = proper && colors(nb) != colors(i);
  check(n == 105 && ANCFCPUUtils::ValidateColoring(elems, colors) && proper && n_colors == 12,
        "GreedyVertexColoring: proper colouring of beam_3x2x1 with 12 colours, ValidateColoring accepts it");
  auto c2n = ANCFCPUUtils::BuildColorToNodes(colors, n_colors);
  auto inc = ANCFCPUUtils::BuildNodeIncidence(elems, n);
  size_t total = 0, incs = 0;
  for (auto& v : c2n) total += v.size();
  for (auto& v : inc) incs += v.size();
  check(total == 105 && incs == 360 && inc[elems(3, 7)].size() >= 1, "BuildColorToNodes / BuildNodeIncidence cover every node / (element, local) pair");
  tlfea::VectorXi bad = colors;
  bad(elems(0, 1)) = bad(elems(0, 0));
  check(!ANCFCPUUtils::ValidateColoring(elems, bad), "ValidateColoring rejects a repeated colour inside an element");
}

// mesh_manager.cc:180-220, 443-570 semantics (host only)
void mesh_manager_known_answers(const std::string& d) {
  ANCFCPUUtils::MeshManager mm;
  const int a = mm.LoadMesh(d + "/cube.1.node", d + "/cube.1.ele", "cube");
  const int b = mm.LoadMesh(d + "/beam_3x2x1.1.node", d + "/beam_3x2x1.1.ele");
  check(a == 0 && b == 1 && mm.GetNumMeshes() == 2 && mm.GetTotalNodes() == 27 + 105 && mm.GetTotalElements() == 6 + 36,
        "MeshManager: two meshes, 132 nodes, 42 elements");
  const ANCFCPUUtils::MeshInstance& i1 = mm.GetMeshInstance(1);
  check(i1.node_offset == 27 && i1.element_offset == 6 && i1.name == "mesh_1", "MeshManager: instance offsets and default name");
  tlfea::MatrixXi eb;
  ANCFCPUUtils::FEAT10_read_elements(d + "/beam_3x2x1.1.ele", eb);
  bool shifted = mm.GetAllElements().rows() == 42 && mm.GetAllElements().cols() == 10;
  for (int e = 0; shifted && e < 36; e++)
    for (int k = 0; k < 10; k++) shifted = shifted && mm.GetAllElements()(6 + e, k) == eb(e, k) + 27;
  check(shifted, "MeshManager: element ids of the second mesh shifted by its node offset");
  const double x0 = mm.GetAllNodes()(27, 0), y0 = mm.GetAllNodes()(27, 1), z0 = mm.GetAllNodes()(27, 2), c0 = mm.GetAllNodes()(5, 0);
  mm.TranslateMesh(1, 1.0, 2.0, 3.0);
  check(mm.GetAllNodes()(27, 0) == x0 + 1.0 && mm.GetAllNodes()(27, 1) == y0 + 2.0 && mm.GetAllNodes()(27, 2) == z0 + 3.0 &&
            mm.GetAllNodes()(5, 0) == c0,
        "MeshManager: TranslateMesh moves one instance only");
  ANCFCPUUtils::Matrix4d T = ANCFCPUUtils::uniformScale(2.0);
  mm.TransformMesh(0, T);
  double mx = 0;
  for (int i = 0; i < 27; i++)
    for (int c = 0; c < 3; c++) mx = std::max(mx, std::fabs(mm.GetAllNodes()(i, c)));
  check(mx == 2.0, "MeshManager: TransformMesh(uniformScale(2)) on the unit cube");
  ANCFCPUUtils::Matrix4d R = ANCFCPUUtils::rotationY(std::acos(-1.0) / 2);
  check(std::fabs(R(0, 2) - 1.0) < 1e-15 && std::fabs(R(2, 0) + 1.0) < 1e-15 && std::fabs(R(0, 0)) < 1e-15, "rotationY(pi/2)");
  check(mm.GetMeshIdFromElement(5) == 0 && mm.GetMeshIdFromElement(6) == 1 && mm.GetMeshIdFromElement(99) == -1 &&
            mm.GetMeshIdFromNode(26) == 0 && mm.GetMeshIdFromNode(27) == 1 && mm.GetMeshIdFromNode(-1) == -1,
        "MeshManager: GetMeshIdFromElement / GetMeshIdFromNode");
  bool threw = false;
  try {
    mm.GetMeshInstance(7);
  } catch (const std::out_of_range&) {
    threw = true;
  }
  check(threw, "MeshManager: GetMeshInstance out of range throws std::out_of_range");
  check(mm.LoadMesh("/nonexistent.node", "/nonexistent.ele") == -1 && mm.GetNumMeshes() == 2, "MeshManager: missing file -> -1");
  tlfea::VectorXd f(105), bad(3);
  for (int i = 0; i < 105; i++) f(i) = i;
  check(!mm.HasScalarFields() && !mm.SetScalarField(1, bad) && mm.SetScalarField(1, f) && mm.HasScalarFields() &&
            mm.GetAllScalarFields().size() == 132 && mm.GetAllScalarFields()(27 + 7) == 7.0 && mm.GetAllScalarFields()(3) == 0.0,
        "MeshManager: SetScalarField fills the unified field, other meshes stay zero");
  mm.Clear();
  check(mm.GetNumMeshes() == 0 && mm.GetTotalNodes() == 0 && mm.GetAllNodes().rows() == 0 && !mm.HasScalarFields(), "MeshManager: Clear");
}

// Two copies of the 36-element bar in ONE GPU_FEAT10_Data (the multi-body flow of lib_bin/collision_system without
// the contact forces): same load on both -> the two bodies must move identically, independent of each other.
void two_bodies_one_system(const std::string& d) {
  ANCFCPUUtils::MeshManager mm;
  mm.LoadMesh(d + "/beam_3x2x1.1.node", d + "/beam_3x2x1.1.ele", "bar_0");
  const int m1 = mm.LoadMesh(d + "/beam_3x2x1.1.node", d + "/beam_3x2x1.1.ele", "bar_1");
  mm.TranslateMesh(m1, 0.0, 4.0, 0.0);
  const tlfea::MatrixXd& nodes = mm.GetAllNodes();
  const int N = mm.GetTotalNodes(), E = mm.GetTotalElements(), n1 = mm.GetMeshInstance(m1).node_offset;
  GPU_FEAT10_Data data(E, N);
  data.Initialize();
  tlfea::VectorXd x(N), y(N), z(N), f(3 * N);
  std::vector<int> fixed;
  for (int i = 0; i < N; i++) {
    x(i) = nodes(i, 0); y(i) = nodes(i, 1); z(i) = nodes(i, 2);
    if (std::fabs(x(i)) < 1e-8) fixed.push_back(i);
  }
  tlfea::VectorXi fx(static_cast<int>(fixed.size()));
  for (size_t i = 0; i < fixed.size(); i++) fx(static_cast<int>(i)) = fixed[i];
  data.SetNodalFixed(fx);
  f(3 * 19) = 1000.0;         // node 19 of body 0, +x (f-form-T10-beam-newton.py:389-397)
  f(3 * (n1 + 19)) = 1000.0;  // the same node of body 1
  data.SetExternalForce(f);
  data.Setup(Quadrature::tet5pt_x, Quadrature::tet5pt_y, Quadrature::tet5pt_z, Quadrature::tet5pt_weights, x, y, z,
             mm.GetAllElements());
  data.SetDensity(2700.0);
  data.SetDamping(0.0, 0.0);
  data.SetSVK(7e8, 0.33);
  data.CalcDnDuPre();
  data.CalcMassMatrix();
  data.CalcConstraintData();
  data.ConvertToCSR_ConstraintJacT();
  data.BuildConstraintJacobianCSR();
  SyncedNewtonParams params = {1e-6, 0.0, 1e-6, 1e14, 5, 10, 1e-3};
  SyncedNewtonSolver solver(&data, data.get_n_constraint());
  solver.Setup();
  solver.SetParameters(&params);
  for (int step = 0; step < 3; step++) solver.Solve();
  tlfea::VectorXd xx, yy, zz;
  data.RetrievePositionToCPU(xx, yy, zz);
  double moved = 0.0, diff = 0.0;
  for (int i = 0; i < n1; i++) {
    moved = std::max(moved, std::fabs(xx(i) - x(i)));
    diff = std::max(diff, std::fabs((xx(i) - x(i)) - (xx(n1 + i) - x(n1 + i))));
    diff = std::max(diff, std::fabs((yy(i) - y(i)) - (yy(n1 + i) - y(n1 + i))));
    diff = std::max(diff, std::fabs((zz(i) - z(i)) - (zz(n1 + i) - z(n1 + i))));
  }
  check(moved > 1e-9 && diff <= 1e-9 * moved,
        "two bodies in one system: identical displacement histories (max " + std::to_string(moved * 1e6) + " um, mismatch " +
            std::to_string(diff / moved) + " rel)");
  data.Destroy();
}
}  // namespace

int main(int argc, char** argv) {
  std::string data_dir = ".", tmp_dir = "/tmp";
  bool print = false;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a.rfind("--data_dir=", 0) == 0) data_dir = a.substr(11);
    else if (a.rfind("--tmp_dir=", 0) == 0) tmp_dir = a.substr(10);
    else if (a == "--print_dsdu") print = true;
    else {
      std::cerr << "Unknown argument: " << a << std::endl;
      return 100;
    }
  }
  utils_known_answers();  // host-only
  mesh_manager_known_answers(data_dir);
  coloring_known_answers(data_dir);
  vtu_known_answers(tmp_dir);
  if (tlfea_device_count() <= 0) {
    std::cerr << "No HIP device visible" << std::endl;
    return 101;
  }
  mass_known_answer(2, data_dir);
  mass_known_answer(3, data_dir);
  strip_3443();
  two_bodies_one_system(data_dir);
  if (print) {  // PrintDsDuPre text of one beam (ANCF3243Data.cu:326-360)
    ANCFCPUUtils::GridMeshGenerator gg(2.0, 0.0, 2.0, true, false);
    gg.generate_mesh();
    GPU_ANCF3243_Data d(gg.get_num_nodes(), gg.get_num_elements());
    d.Initialize();
    tlfea::VectorXd x, y, z;
    gg.get_coordinates(x, y, z);
    tlfea::MatrixXi conn;
    gg.get_element_connectivity(conn);
    d.Setup(2.0, 1.0, 1.0, Quadrature::gauss_xi_m_6, Quadrature::gauss_xi_3, Quadrature::gauss_eta_2,
            Quadrature::gauss_zeta_2, Quadrature::weight_xi_m_6, Quadrature::weight_xi_3, Quadrature::weight_eta_2,
            Quadrature::weight_zeta_2, x, y, z, conn);
    d.CalcDsDuPre();
    d.PrintDsDuPre();
    d.Destroy();
  }
  std::cout << (n_failed ? "FAILED " : "PASSED ") << n_failed << " failed" << std::endl;
  return n_failed;
}
