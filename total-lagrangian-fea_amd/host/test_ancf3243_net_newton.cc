// ANCF-3243 beam net with welded / pinned joints: the flow of the reference's
// lib_bin/mesh_deform/test_ancf3243_net_newton.cc (:385-505) on the facade over the C-ABI.
// Same options (--joint --steps --dt --W --H --center_force_z --verbose), materials (:32-37), corner clamps written as
// AddFixedDof rows appended to the mesh file's joint constraints (:233-276), centre point load (:278-337) and solver
// parameters (:484).  --mesh=PATH replaces the reference's data/ paths; --vtu[=DIR] writes one hexahedron per beam every
// 10 steps to output/ancf3243_net/ancf3243_net_<step>.vtu (:40-41, 465-496).
// Prints the centre deflection per step as CSV (the reference prints nothing per step).
#include <cmath>
#include <cstdio>
#include <filesystem>
#include <limits>
#include <memory>

#include "tlfea_facade.h"

namespace {
constexpr double kE = 7e8, kNu = 0.33, kRho0 = 2700;

struct Options {
  std::string joint = "welded", mesh, vtu_dir;
  int steps = 50;
  double dt = 1e-3, W = 0.1, H = 0.1, center_force_z = -1000.0;
  bool verbose = false;
};

bool starts_with(const std::string& s, const std::string& p) { return s.rfind(p, 0) == 0; }

bool parse_args(int argc, char** argv, Options& o) {
  for (int i = 1; i < argc; ++i) {
    const std::string a(argv[i]);
    auto val = [&](const char* key) { return a.substr(std::string(key).size()); };
    try {
      if (starts_with(a, "--joint=")) {
        o.joint = val("--joint=");
        if (o.joint != "welded" && o.joint != "pinned") {
          std::cerr << "Invalid --joint (expected welded|pinned): " << o.joint << "\n";
          return false;
        }
      } else if (starts_with(a, "--mesh=")) o.mesh = val("--mesh=");
      else if (starts_with(a, "--steps=")) o.steps = std::stoi(val("--steps="));
      else if (starts_with(a, "--dt=")) o.dt = std::stod(val("--dt="));
      else if (starts_with(a, "--W=")) o.W = std::stod(val("--W="));
      else if (starts_with(a, "--H=")) o.H = std::stod(val("--H="));
      else if (starts_with(a, "--center_force_z=")) o.center_force_z = std::stod(val("--center_force_z="));
      else if (a == "--vtu") o.vtu_dir = "output/ancf3243_net";
      else if (starts_with(a, "--vtu=")) o.vtu_dir = val("--vtu=");
      else if (a == "--verbose") o.verbose = true;
      else {
        std::cerr << "Unknown argument: " << a << "\n";
        return false;
      }
    } catch (...) {
      std::cerr << "Invalid value: " << a << "\n";
      return false;
    }
  }
  return o.steps > 0 && o.dt > 0.0 && o.W > 0.0 && o.H > 0.0;
}

struct Bounds2D {
  double xmin = std::numeric_limits<double>::infinity(), xmax = -xmin, ymin = xmin, ymax = -xmin;
};
Bounds2D bounds_xy(const ANCFCPUUtils::ANCF3243Mesh& m) {
  Bounds2D b;
  for (int n = 0; n < m.n_nodes; ++n) {
    b.xmin = std::min(b.xmin, m.x12(4 * n)); b.xmax = std::max(b.xmax, m.x12(4 * n));
    b.ymin = std::min(b.ymin, m.y12(4 * n)); b.ymax = std::max(b.ymax, m.y12(4 * n));
  }
  return b;
}
std::vector<int> nodes_at_xy(const ANCFCPUUtils::ANCF3243Mesh& m, double x, double y, double tol) {
  std::vector<int> out;
  for (int n = 0; n < m.n_nodes; ++n)
    if (std::abs(m.x12(4 * n) - x) <= tol && std::abs(m.y12(4 * n) - y) <= tol) out.push_back(n);
  return out;
}
}  // namespace

int main(int argc, char** argv) {
  Options opt;
  if (!parse_args(argc, argv, opt)) return 1;
  if (opt.mesh.empty())
    opt.mesh = std::string("tests/golden/meshes/ANCF3243/net_") + opt.joint + "_nx20_ny20_L0.5.ancf3243mesh";
  ANCFCPUUtils::ANCF3243Mesh mesh;
  std::string err;
  if (!ANCFCPUUtils::ReadANCF3243MeshFromFile(opt.mesh, mesh, &err)) {
    std::cerr << err << "\n";
    return 1;
  }
  const int n_nodes = mesh.n_nodes, n_dofs = 12 * n_nodes;
  const double L = mesh.has_grid ? mesh.grid_L : 0.5;
  std::cout << "ANCF3243 net: mesh=" << opt.mesh << " nodes=" << n_nodes << " elements=" << mesh.n_elements
            << " constraints(from mesh)=" << mesh.constraints.NumRows() << " steps=" << opt.steps << " dt=" << opt.dt
            << std::endl;

  GPU_ANCF3243_Data data(n_nodes, mesh.n_elements);
  data.Initialize();

  const Bounds2D b = bounds_xy(mesh);
  const double tol = std::max(1e-12, 1e-9 * std::hypot(b.xmax - b.xmin, b.ymax - b.ymin));
  tlfea::VectorXd f_ext(n_dofs);
  f_ext.setZero();
  const std::vector<int> centre = nodes_at_xy(mesh, 0.5 * (b.xmin + b.xmax), 0.5 * (b.ymin + b.ymax), tol);
  if (centre.empty()) {
    std::cerr << "no node at the net centre\n";
    return 1;
  }
  for (int n : centre) f_ext((4 * n) * 3 + 2) += opt.center_force_z / static_cast<double>(centre.size());
  data.SetExternalForce(f_ext);

  data.Setup(L, opt.W, opt.H, Quadrature::gauss_xi_m_6, Quadrature::gauss_xi_3, Quadrature::gauss_eta_2,
             Quadrature::gauss_zeta_2, Quadrature::weight_xi_m_6, Quadrature::weight_xi_3, Quadrature::weight_eta_2,
             Quadrature::weight_zeta_2, mesh.x12, mesh.y12, mesh.z12, mesh.element_connectivity);
  data.SetDensity(kRho0);
  data.SetDamping(1e5, 1e5);
  data.SetSVK(kE, kNu);

  std::unique_ptr<ANCFCPUUtils::LinearConstraintBuilder> builder;
  if (mesh.constraints.Empty()) builder = std::make_unique<ANCFCPUUtils::LinearConstraintBuilder>(n_dofs);
  else builder = std::make_unique<ANCFCPUUtils::LinearConstraintBuilder>(n_dofs, mesh.constraints);
  const double corners[4][2] = {{b.xmin, b.ymin}, {b.xmax, b.ymin}, {b.xmin, b.ymax}, {b.xmax, b.ymax}};
  for (const auto& c : corners) {
    const std::vector<int> nodes = nodes_at_xy(mesh, c[0], c[1], tol);
    if (opt.verbose) std::cout << "Corner clamp (" << c[0] << ", " << c[1] << "): found nodes=" << nodes.size() << "\n";
    if (nodes.size() != 2)
      std::cerr << "Warning: expected 2 (H/V) nodes at corner (" << c[0] << ", " << c[1] << "), got " << nodes.size() << "\n";
    for (int n : nodes)
      for (int slot = 0; slot < 4; ++slot)
        ANCFCPUUtils::AppendANCF3243FixedCoefficient(*builder, 4 * n + slot, mesh.x12, mesh.y12, mesh.z12);
  }
  const ANCFCPUUtils::LinearConstraintCSR all = builder->ToCSR();
  data.SetLinearConstraintsCSR(all.offsets, all.columns, all.values, all.rhs);

  data.CalcDsDuPre();
  data.CalcMassMatrix();
  data.CalcConstraintData();
  data.CalcP();
  data.CalcInternalForce();

  SyncedNewtonParams params = {1e-4, 0.0, 1e-6, 1e14, 5, 10, opt.dt};
  SyncedNewtonSolver solver(&data, data.get_n_constraint());
  solver.Setup();
  solver.SetParameters(&params);

  std::printf("step,centre_z,constraint_norm\n");
  tlfea::VectorXd x12, y12, z12, c;
  auto write_vtu = [&](int step) {
    if (opt.vtu_dir.empty() || step % 10 != 0) return;
    char name[64];
    std::snprintf(name, sizeof name, "/ancf3243_net_%06d.vtu", step);
    ANCFCPUUtils::VisualizationUtils::ExportANCF3243ToVTU(x12, y12, z12, mesh.element_connectivity, opt.W, opt.H,
                                                          opt.vtu_dir + name);
  };
  if (!opt.vtu_dir.empty()) {
    std::filesystem::create_directories(opt.vtu_dir);
    data.RetrievePositionToCPU(x12, y12, z12);
    write_vtu(0);
  }
  for (int step = 0; step < opt.steps; ++step) {
    solver.Solve();
    data.RetrievePositionToCPU(x12, y12, z12);
    write_vtu(step + 1);
    data.CalcConstraintData();
    data.RetrieveConstraintDataToCPU(c);
    double cn = 0.0;
    for (int k = 0; k < c.size(); ++k) cn += c(k) * c(k);
    std::printf("%d,%.17g,%.6e\n", step + 1, z12(4 * centre[0]), std::sqrt(cn));
  }
  data.Destroy();
  return 0;
}
