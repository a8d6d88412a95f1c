// ANCF-3443 shell strip clamped at one end with a tip load: the flow of the reference's lib_bin/beam_sag/test_ancf3443.cc
// (:237-460) on the facade over the C-ABI.  Strip of --n_beam 2 x 1 shells of thickness 0.1 (strip constructor and
// ANCF3443_generate_beam_coordinates), the two nodes of the left edge pinned (8 coefficients), the tip force split
// between the two tip nodes by --lrratio, SVK 7e8 / 0.33 / 2700, no damping; the four solver kinds with the reference's
// parameters (default vbd, omega 1.8); CSV schema `step,tip_z`; --vtu writes hexahedra every 20 steps.
//   ./test_ancf3443 [--solver=vbd|newton|nesterov|adamw] [--n_beam=2] [--steps=50] [--dt=1e-3] [--tip_force_z=-100]
//                   [--lrratio=0.5] [--omega=1.8] [--csv[=PATH]] [--vtu[=DIR]]
#include <cmath>
#include <filesystem>
#include <iomanip>
#include <limits>
#include <memory>

#include "tlfea_facade.h"

namespace {
constexpr double kE = 7e8, kNu = 0.33, kRho0 = 2700;  // :35-37
constexpr double kL = 2.0, kW = 1.0, kH = 0.1;        // :39-41
constexpr int kVtuEvery = 20;
bool starts_with(const std::string& s, const std::string& p) { return s.rfind(p, 0) == 0; }
}  // namespace

int main(int argc, char** argv) {
  std::string solver_kind = "vbd", csv_path, vtu_dir;
  int n_beam = 2, steps = 50;
  double dt = 1e-3, tip_fz = std::numeric_limits<double>::quiet_NaN(), lrratio = 0.5,
         omega = std::numeric_limits<double>::quiet_NaN();
  for (int i = 1; i < argc; i++) {
    const std::string a(argv[i]);
    if (starts_with(a, "--solver=")) {
      solver_kind = a.substr(9);
      if (solver_kind != "newton" && solver_kind != "nesterov" && solver_kind != "adamw" && solver_kind != "vbd") {
        std::cerr << "Invalid --solver: " << solver_kind << "\n";
        return 1;
      }
    } else if (starts_with(a, "--n_beam=")) {
      n_beam = std::atoi(a.c_str() + 9);
      if (n_beam <= 0) {
        std::cerr << "Invalid --n_beam: " << a.substr(9) << "\n";
        return 1;
      }
    } else if (starts_with(a, "--steps=")) steps = std::atoi(a.c_str() + 8);
    else if (starts_with(a, "--dt=")) dt = std::atof(a.c_str() + 5);
    else if (starts_with(a, "--tip_force_z=")) tip_fz = std::atof(a.c_str() + 14);
    else if (starts_with(a, "--lrratio=")) lrratio = std::atof(a.c_str() + 10);
    else if (starts_with(a, "--omega=")) omega = std::atof(a.c_str() + 8);
    else if (a == "--csv") csv_path = "tip_z_history_ancf3443_" + solver_kind + ".csv";
    else if (starts_with(a, "--csv=")) csv_path = a.substr(6);
    else if (a == "--vtu") vtu_dir = "output/ancf3443";
    else if (starts_with(a, "--vtu=")) vtu_dir = a.substr(6);
    else {
      std::cerr << "Unknown argument: " << a << "\n";
      return 1;
    }
  }
  if (std::isnan(tip_fz)) tip_fz = -1000.0 * kH;  // :244-246
  if (tlfea_device_count() <= 0) {
    std::cerr << "No HIP device visible" << std::endl;
    return 1;
  }
  GPU_ANCF3443_Data data(n_beam);  // strip constructor
  data.Initialize();
  std::cout << "ANCF3443: beams=" << n_beam << " coef=" << data.get_n_coef() << " solver=" << solver_kind
            << " steps=" << steps << " dt=" << dt << " L=" << kL << " W=" << kW << " H=" << kH
            << " tip_force_z=" << tip_fz << std::endl;
  tlfea::VectorXd x, y, z;
  tlfea::MatrixXi conn;
  ANCFCPUUtils::ANCF3443_generate_beam_coordinates(n_beam, x, y, z, conn);
  tlfea::VectorXi fixed(8);
  {
    int k = 0;
    for (int node : {conn(0, 0), conn(0, 3)})
      for (int d = 0; d < 4; d++) fixed(k++) = 4 * node + d;
  }
  data.SetNodalFixed(fixed);
  tlfea::VectorXd f_ext(3 * data.get_n_coef());
  const int ta = conn(n_beam - 1, 1), tb = conn(n_beam - 1, 2);
  const bool a_neg = y(4 * ta) <= y(4 * tb);
  const int tip_neg = a_neg ? ta : tb, tip_pos = a_neg ? tb : ta;
  f_ext((4 * tip_neg) * 3 + 2) += lrratio * tip_fz;
  f_ext((4 * tip_pos) * 3 + 2) += (1.0 - lrratio) * tip_fz;
  data.SetExternalForce(f_ext);
  data.Setup(kL, kW, kH, Quadrature::gauss_xi_m_7, Quadrature::gauss_eta_m_7, Quadrature::gauss_zeta_m_3,
             Quadrature::gauss_xi_4, Quadrature::gauss_eta_4, Quadrature::gauss_zeta_3, Quadrature::weight_xi_m_7,
             Quadrature::weight_eta_m_7, Quadrature::weight_zeta_m_3, Quadrature::weight_xi_4, Quadrature::weight_eta_4,
             Quadrature::weight_zeta_3, x, y, z, conn);
  data.SetDensity(kRho0);
  data.SetDamping(0.0, 0.0);
  data.SetSVK(kE, kNu);
  data.CalcDsDuPre();
  data.CalcMassMatrix();
  data.CalcConstraintData();
  data.ConvertToCSR_ConstraintJacT();
  data.BuildConstraintJacobianCSR();
  data.CalcP();
  data.CalcInternalForce();

  std::unique_ptr<SolverBase> solver;
  if (solver_kind == "newton") {
    SyncedNewtonParams p = {1e-4, 0.0, 1e-6, 1e14, 5, 10, dt};  // :357
    auto* s = new SyncedNewtonSolver(&data, data.get_n_constraint());
    s->Setup();
    s->SetParameters(&p);
    solver.reset(s);
  } else if (solver_kind == "nesterov") {
    SyncedNesterovParams p = {1.0e-8, 1e14, 1.0e-6, 1.0e-6, 5, 300, dt};  // :379-380
    auto* s = new SyncedNesterovSolver(&data, data.get_n_constraint());
    s->Setup();
    s->SetParameters(&p);
    solver.reset(s);
  } else if (solver_kind == "adamw") {
    SyncedAdamWNocoopParams p = {2e-4, 0.9, 0.999, 1e-8, 1e-4, 0.995, 1e-1, 1e-6, 1e14, 5, 500, dt, 10, 0.0};  // :402-404
    auto* s = new SyncedAdamWNocoopSolver(&data, data.get_n_constraint());
    s->Setup();
    s->SetParameters(&p);
    solver.reset(s);
  } else {
    SyncedVBDParams p = {1e-4, 1e-4, 1e-4, 1e14, 5, 500, dt, std::isnan(omega) ? 1.8 : omega, 1e-12, 25, 1};  // :425-428
    auto* s = new SyncedVBDSolver(&data, data.get_n_constraint());
    s->Setup();
    s->SetParameters(&p);
    s->InitializeColoring();
    s->InitializeMassDiagBlocks();
    s->InitializeFixedMap();
    solver.reset(s);
  }
  auto write_vtu = [&](int step, const tlfea::VectorXd& px, const tlfea::VectorXd& py, const tlfea::VectorXd& pz) {
    if (vtu_dir.empty() || step % kVtuEvery != 0) return;
    std::ostringstream name;
    name << vtu_dir << "/ancf3443_" << solver_kind << "_" << std::setw(6) << std::setfill('0') << step << ".vtu";
    ANCFCPUUtils::VisualizationUtils::ExportANCF3443ToVTU(px, py, pz, conn, kH, name.str());
  };
  if (!vtu_dir.empty()) {
    std::filesystem::create_directories(vtu_dir);
    write_vtu(0, x, y, z);
  }
  std::vector<double> tip_z;
  const int tip_coef_a = 4 * ta, tip_coef_b = 4 * tb;  // the history is the mean z of the two tip nodes (:369)
  for (int step = 0; step < steps; step++) {
    solver->Solve();
    tlfea::VectorXd px, py, pz;
    data.RetrievePositionToCPU(px, py, pz);
    tip_z.push_back(0.5 * (pz(tip_coef_a) + pz(tip_coef_b)));
    std::cout << "Step " << step + 1 << ": tip z = " << std::setprecision(17) << tip_z.back() << std::endl;
    write_vtu(step + 1, px, py, pz);
  }
  if (!csv_path.empty()) {
    std::ofstream csv(csv_path);
    csv << std::fixed << std::setprecision(17) << "step,tip_z\n";
    for (size_t i = 0; i < tip_z.size(); i++) csv << i << "," << tip_z[i] << "\n";
  }
  solver.reset();
  data.Destroy();
  return 0;
}
