// ANCF-3443 airless tire: the flow of the reference's lib_bin/mesh_deform/test_ancf3443_mesh_newton.cc (:227-384) on the
// facade over the C-ABI.  The hub (innermost spoke nodes) is driven through the CSR constraints: its coefficients get
// AppendANCF3243FixedCoefficient rows after the mesh file's own rows, and every step their right-hand side is the
// reference position rotated about y by an angle ramped with a smoothstep (UpdateLinearConstraintRHS); ring nodes below
// the ground plane receive a clamped penalty force recomputed from the current positions (SetExternalForce).
// Same constants (:36-46), options (--mesh --steps --dt --vtu --apply_load_below_z --load_below_z --load_fz
// --contact_fz_max) and solver parameters ({1e-4,0,1e-6,1e12,10,10,dt}, :329); thickness scaled by 0.25 (:243).
// --csv_path=FILE additionally records, per step, the hub angle and the lowest ring position (for the tests).
#include <algorithm>
#include <cmath>
#include <filesystem>
#include <iomanip>
#include <limits>
#include <memory>

#include "tlfea_facade.h"

namespace {
constexpr double kE = 1e8, kNu = 0.33, kRho0 = 2000, kEtaDamp = 5e4, kLambdaDamp = 5e4;
constexpr int kVtuEvery = 5;
constexpr double kThicknessScale = 0.25;
constexpr double kOmegaTarget = 1.5 * 3.14159265358979323846, kOmegaRampTime = 0.05;

struct Options {
  std::string mesh, csv_path, vtu_dir = "output/ancf3443_mesh";
  int steps = 10;
  double dt = 1e-3, ground_z = -0.2, contact_k = 5e4, contact_fz_max = 2e4;
  bool write_vtu = false, contact = true;
};

bool starts_with(const std::string& s, const std::string& p) { return s.rfind(p, 0) == 0; }
bool parse_bool(const std::string& v) { return !(v == "0" || v == "false" || v == "no" || v == "off"); }

bool parse_args(int argc, char** argv, Options& o) {
  for (int i = 1; i < argc; i++) {
    const std::string a(argv[i]);
    auto val = [&](const char* key) { return a.substr(std::string(key).size()); };
    try {
      if (starts_with(a, "--mesh=")) o.mesh = val("--mesh=");
      else if (starts_with(a, "--steps=")) o.steps = std::stoi(val("--steps="));
      else if (starts_with(a, "--dt=")) o.dt = std::stod(val("--dt="));
      else if (a == "--vtu") o.write_vtu = true;
      else if (starts_with(a, "--vtu=")) { o.write_vtu = true; o.vtu_dir = val("--vtu="); }
      else if (starts_with(a, "--apply_load_below_z=")) o.contact = parse_bool(val("--apply_load_below_z="));
      else if (starts_with(a, "--load_below_z=")) o.ground_z = std::stod(val("--load_below_z="));
      else if (starts_with(a, "--load_fz=")) o.contact_k = std::stod(val("--load_fz="));
      else if (starts_with(a, "--contact_fz_max=")) o.contact_fz_max = std::stod(val("--contact_fz_max="));
      else if (starts_with(a, "--csv_path=")) o.csv_path = val("--csv_path=");
      else if (starts_with(a, "--load_per_node=") || starts_with(a, "--load_fx=") || starts_with(a, "--load_fy=")) {
      }  // legacy options of the reference: accepted and ignored there too (:182-186)
      else {
        std::cerr << "Unknown argument: " << a << "\n";
        return false;
      }
    } catch (...) {
      std::cerr << "Invalid value: " << a << "\n";
      return false;
    }
  }
  return true;
}

double smoothstep01(double s) {
  const double x = std::clamp(s, 0.0, 1.0);
  return x * x * (3.0 - 2.0 * x);
}

// innermost "S" (spoke) nodes in the x-z plane (:123-157)
std::vector<int> inner_spoke_nodes(const ANCFCPUUtils::ANCF3443Mesh& m) {
  std::vector<int> out;
  if (static_cast<int>(m.node_family.size()) != m.n_nodes) return out;
  auto radius = [&](int n) { return std::hypot(m.x12(4 * n), m.z12(4 * n)); };
  double r_min = std::numeric_limits<double>::infinity();
  for (int n = 0; n < m.n_nodes; n++)
    if (m.node_family[n] == "S") r_min = std::min(r_min, radius(n));
  if (!std::isfinite(r_min)) return out;
  const double tol = std::max(1e-12, 1e-8 * std::max(1.0, r_min));
  for (int n = 0; n < m.n_nodes; n++)
    if (m.node_family[n] == "S" && radius(n) - r_min <= tol) out.push_back(n);
  std::cout << "Hub nodes (inner spoke nodes): r_min=" << r_min << " tol=" << tol << " nodes=" << out.size() << std::endl;
  return out;
}

// penalty contact of the ring ("R") nodes with the plane z = ground_z (:86-121); -> number of nodes in contact
int ground_contact(tlfea::VectorXd& f_ext, const ANCFCPUUtils::ANCF3443Mesh& m, const tlfea::VectorXd& z12, double ground_z,
                   double k, double fz_max) {
  if (static_cast<int>(m.node_family.size()) != m.n_nodes) return 0;
  int count = 0;
  for (int n = 0; n < m.n_nodes; n++) {
    if (m.node_family[n] != "R" || !(z12(4 * n) < ground_z)) continue;
    double fz = k * (ground_z - z12(4 * n));
    if (fz_max > 0.0) fz = std::min(fz, fz_max);
    f_ext((4 * n) * 3 + 2) += fz;
    count++;
  }
  return count;
}
}  // namespace

int main(int argc, char** argv) {
  Options opt;
  if (!parse_args(argc, argv, opt)) return 1;
  if (opt.mesh.empty()) {
    std::cerr << "--mesh is required\n";
    return 2;
  }
  if (!std::filesystem::exists(opt.mesh)) {
    std::cerr << "Mesh file not found: " << opt.mesh << "\n";
    return 2;
  }
  if (opt.steps <= 0) {
    std::cerr << "Invalid --steps (must be > 0): " << opt.steps << "\n";
    return 2;
  }
  if (!(opt.dt > 0.0)) {
    std::cerr << "Invalid --dt (must be > 0): " << opt.dt << "\n";
    return 2;
  }
  ANCFCPUUtils::ANCF3443Mesh mesh;
  std::string err;
  if (!ANCFCPUUtils::ReadANCF3443MeshFromFile(opt.mesh, mesh, &err)) {
    std::cerr << err << "\n";
    return 2;
  }
  if (tlfea_device_count() <= 0) {
    std::cerr << "No HIP device visible" << std::endl;
    return 1;
  }
  std::cout << "ANCF3443 mesh: mesh=" << opt.mesh << " nodes=" << mesh.n_nodes << " elements=" << mesh.n_elements
            << " coef=" << 4 * mesh.n_nodes << " constraints(from mesh)=" << mesh.constraints.NumRows()
            << " steps=" << opt.steps << " dt=" << opt.dt << std::endl;

  tlfea::VectorXd H_scaled = mesh.element_H;
  double thickness = 0.0;
  for (int e = 0; e < H_scaled.size(); e++) {
    H_scaled(e) *= kThicknessScale;
    thickness = std::max(thickness, H_scaled(e));
  }
  if (!(thickness > 0.0)) thickness = 1e-3;

  GPU_ANCF3443_Data data(mesh.n_nodes, mesh.n_elements);
  data.Initialize();
  data.Setup(mesh.element_L, mesh.element_W, H_scaled, Quadrature::gauss_xi_m_7, Quadrature::gauss_eta_m_7,
             Quadrature::gauss_zeta_m_3, Quadrature::gauss_xi_4, Quadrature::gauss_eta_4, Quadrature::gauss_zeta_3,
             Quadrature::weight_xi_m_7, Quadrature::weight_eta_m_7, Quadrature::weight_zeta_m_3, Quadrature::weight_xi_4,
             Quadrature::weight_eta_4, Quadrature::weight_zeta_3, mesh.x12, mesh.y12, mesh.z12, mesh.element_connectivity);
  data.SetDensity(kRho0);
  data.SetDamping(kEtaDamp, kLambdaDamp);
  data.SetSVK(kE, kNu);

  const int n_dofs = 4 * mesh.n_nodes * 3;
  std::unique_ptr<ANCFCPUUtils::LinearConstraintBuilder> builder =
      mesh.constraints.Empty() ? std::make_unique<ANCFCPUUtils::LinearConstraintBuilder>(n_dofs)
                               : std::make_unique<ANCFCPUUtils::LinearConstraintBuilder>(n_dofs, mesh.constraints);
  const std::vector<int> hub_nodes = inner_spoke_nodes(mesh);
  const int hub_row0 = builder->num_rows();
  std::vector<int> hub_coefs;
  for (int n : hub_nodes)
    for (int slot = 0; slot < 4; slot++) {
      hub_coefs.push_back(4 * n + slot);
      ANCFCPUUtils::AppendANCF3243FixedCoefficient(*builder, 4 * n + slot, mesh.x12, mesh.y12, mesh.z12);
    }
  std::cout << "Hub prescribed rotation: coef_fixed=" << hub_coefs.size() << " rows=" << 3 * hub_coefs.size() << std::endl;
  const ANCFCPUUtils::LinearConstraintCSR all = builder->ToCSR();
  data.SetLinearConstraintsCSR(all.offsets, all.columns, all.values, all.rhs);
  data.CalcDsDuPre();
  data.CalcMassMatrix();
  data.CalcConstraintData();
  {
    tlfea::VectorXd c;
    data.RetrieveConstraintDataToCPU(c);
    double l2 = 0.0, linf = 0.0;
    for (int i = 0; i < c.size(); i++) {
      l2 += c(i) * c(i);
      linf = std::max(linf, std::fabs(c(i)));
    }
    std::cout << "Constraint residual: rows=" << c.size() << " ||c||_2=" << std::sqrt(l2) << " ||c||_inf=" << linf << std::endl;
  }
  data.CalcP();
  data.CalcInternalForce();

  auto write_vtu = [&](int step, const tlfea::VectorXd& x, const tlfea::VectorXd& y, const tlfea::VectorXd& z) {
    if (!opt.write_vtu || step % kVtuEvery != 0) return;
    std::ostringstream name;
    name << opt.vtu_dir << "/ancf3443_mesh_" << std::setw(6) << std::setfill('0') << step << ".vtu";
    ANCFCPUUtils::VisualizationUtils::ExportANCF3443ToVTU(x, y, z, mesh.element_connectivity, thickness, name.str());
  };
  if (opt.write_vtu) std::filesystem::create_directories(opt.vtu_dir);

  SyncedNewtonParams params = {1e-4, 0.0, 1e-6, 1e12, 10, 10, opt.dt};
  SyncedNewtonSolver solver(&data, data.get_n_constraint());
  solver.Setup();
  solver.SetParameters(&params);

  std::ofstream csv;
  if (!opt.csv_path.empty()) {
    csv.open(opt.csv_path);
    csv << std::setprecision(17) << "step,theta,contacts,hub_x,hub_z,ring_min_z\n";
  }
  tlfea::VectorXd x12, y12, z12;
  data.RetrievePositionToCPU(x12, y12, z12);
  write_vtu(0, x12, y12, z12);
  tlfea::VectorXd rhs = all.rhs;
  double theta = 0.0;
  for (int step = 0; step < opt.steps; step++) {
    data.RetrievePositionToCPU(x12, y12, z12);
    tlfea::VectorXd f_ext(n_dofs);
    const int contacts = opt.contact ? ground_contact(f_ext, mesh, z12, opt.ground_z, opt.contact_k, opt.contact_fz_max) : 0;
    data.SetExternalForce(f_ext);
    const double t_mid = (step + 0.5) * opt.dt;
    theta += kOmegaTarget * smoothstep01(t_mid / kOmegaRampTime) * opt.dt;
    if (!hub_coefs.empty()) {
      rhs = all.rhs;
      const double c = std::cos(theta), sn = std::sin(theta);
      for (size_t i = 0; i < hub_coefs.size(); i++) {
        const int coef = hub_coefs[i], row0 = hub_row0 + static_cast<int>(3 * i);
        const double x = mesh.x12(coef), z = mesh.z12(coef);
        rhs(row0 + 0) = c * x + sn * z;  // rotation about y (:71-79)
        rhs(row0 + 1) = mesh.y12(coef);
        rhs(row0 + 2) = -sn * x + c * z;
      }
      data.UpdateLinearConstraintRHS(rhs);
    }
    solver.Solve();
    data.RetrievePositionToCPU(x12, y12, z12);
    write_vtu(step + 1, x12, y12, z12);
    if (csv.is_open()) {
      double ring_min = std::numeric_limits<double>::infinity();
      for (int n = 0; n < mesh.n_nodes; n++)
        if (mesh.node_family[n] == "R") ring_min = std::min(ring_min, z12(4 * n));
      const int hc = hub_coefs.empty() ? 0 : hub_coefs[0];
      csv << step << "," << theta << "," << contacts << "," << x12(hc) << "," << z12(hc) << "," << ring_min << "\n";
    }
  }
  data.Destroy();
  return 0;
}
