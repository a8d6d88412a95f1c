// test_ancf3243 -- the reference's ANCF-3243 cantilever driver (lib_bin/beam_sag/test_ancf3243.cc:222-437, Newton
// branch; BASELINE config A) on the MI355X engine: 30 beam elements L=0.5 W=H=0.1, coefficients 0-3 pinned, tip
// force Fz=3100 N, Kelvin-Voigt damping 1e5/1e5, params {1e-4,0,1e-6,1e14,5,10,dt}; CSV schema `step,tip_z`.
//   ./test_ancf3243 --steps=50 --dt=1e-3 [--n_elements=30] [--tip_force_z=3100] [--csv_path=out.csv] [--vtu[=DIR]]
// --vtu writes one hexahedron per beam every 20 steps to output/ancf3243/ancf3243_<solver>_<step>.vtu (:46-47,237-239,302-318)
#include <filesystem>
#include <iomanip>
#include <limits>
#include <memory>

#include "tlfea_facade.h"

namespace {
constexpr double kE = 7e8, kNu = 0.33, kRho0 = 2700;           // :36-38
constexpr double kL = 0.5, kW = 0.1, kH = 0.1, kTipFz = 3100;  // :40-43
bool StartsWith(const std::string& s, const std::string& p) { return s.rfind(p, 0) == 0; }
}  // namespace

int main(int argc, char** argv) {
  int steps = 50, n_elements = 30;
  double dt = 1e-3, tip_fz = kTipFz;
  std::string csv_path, solver_kind = "newton", vtu_dir;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (StartsWith(a, "--steps=")) steps = std::atoi(a.c_str() + 8);
    else if (StartsWith(a, "--dt=")) dt = std::atof(a.c_str() + 5);
    else if (StartsWith(a, "--n_elements=")) n_elements = std::atoi(a.c_str() + 13);
    else if (StartsWith(a, "--tip_force_z=")) tip_fz = std::atof(a.c_str() + 14);
    else if (StartsWith(a, "--csv_path=")) csv_path = a.substr(11);
    else if (a == "--csv") csv_path = "tip_z_history_ancf3243_newton.csv";
    else if (a == "--vtu") vtu_dir = "output/ancf3243";
    else if (StartsWith(a, "--vtu=")) vtu_dir = a.substr(6);
    else if (StartsWith(a, "--solver=")) {
      solver_kind = a.substr(9);
      if (solver_kind != "newton" && solver_kind != "adamw" && solver_kind != "nesterov") {
        std::cerr << "Invalid --solver (built: newton | adamw | nesterov): " << solver_kind << std::endl;
        return 1;
      }
    }
    else { std::cerr << "Unknown argument: " << a << std::endl; return 1; }
  }
  if (tlfea_device_count() <= 0) {
    std::cerr << "No HIP device visible" << std::endl;
    return 1;
  }
  ANCFCPUUtils::GridMeshGenerator grid_gen(n_elements * kL, 0.0, kL, true, false);
  grid_gen.generate_mesh();
  const int n_nodes = grid_gen.get_num_nodes();
  GPU_ANCF3243_Data data(n_nodes, grid_gen.get_num_elements());
  data.Initialize();
  tlfea::VectorXd h_x12, h_y12, h_z12;
  grid_gen.get_coordinates(h_x12, h_y12, h_z12);
  tlfea::MatrixXi conn;
  grid_gen.get_element_connectivity(conn);
  tlfea::VectorXi h_fixed(4);
  for (int i = 0; i < 4; i++) h_fixed(i) = i;
  data.SetNodalFixed(h_fixed);
  tlfea::VectorXd h_f_ext(data.get_n_coef() * 3);
  const int tip_coef = conn(grid_gen.get_num_elements() - 1, 1) * 4;
  h_f_ext(tip_coef * 3 + 2) = tip_fz;
  data.SetExternalForce(h_f_ext);
  data.Setup(kL, kW, kH, Quadrature::gauss_xi_m_6, Quadrature::gauss_xi_3, Quadrature::gauss_eta_2,
             Quadrature::gauss_zeta_2, Quadrature::weight_xi_m_6, Quadrature::weight_xi_3, Quadrature::weight_eta_2,
             Quadrature::weight_zeta_2, h_x12, h_y12, h_z12, conn);
  data.SetDensity(kRho0);
  if (solver_kind == "newton") data.SetDamping(1e5, 1e5);  // test_ancf3243.cc:286-291
  else data.SetDamping(0.0, 0.0);
  data.SetSVK(kE, kNu);
  data.CalcDsDuPre();
  data.CalcMassMatrix();
  data.CalcConstraintData();
  data.ConvertToCSR_ConstraintJacT();
  data.BuildConstraintJacobianCSR();
  data.CalcP();
  data.CalcInternalForce();

  SyncedNewtonParams params = {1e-4, 0.0, 1e-6, 1e14, 5, 10, dt};  // :329
  SyncedAdamWNocoopParams aparams = {2e-4, 0.9, 0.999, 1e-8, 1e-4, 0.998, 1e-1, 1e-6, 1e14, 5, 500, dt, 10, 0.0};  // :374-376
  std::unique_ptr<SolverBase> solver_ptr;
  if (solver_kind == "newton") {
    auto* sv = new SyncedNewtonSolver(&data, data.get_n_constraint());
    sv->Setup();
    sv->SetParameters(&params);
    solver_ptr.reset(sv);
  } else if (solver_kind == "adamw") {
    auto* sv = new SyncedAdamWNocoopSolver(&data, data.get_n_constraint());
    sv->Setup();
    sv->SetParameters(&aparams);
    solver_ptr.reset(sv);
  } else {
    SyncedNesterovParams nparams = {1.0e-8, 1e14, 1.0e-6, 1.0e-6, 5, 200, dt};  // test_ancf3243.cc:351-352
    auto* sv = new SyncedNesterovSolver(&data, data.get_n_constraint());
    sv->Setup();
    sv->SetParameters(&nparams);
    solver_ptr.reset(sv);
  }
  SolverBase& solver = *solver_ptr;
  std::vector<double> tip_z;
  constexpr int kVtuEvery = 20;
  auto write_vtu = [&](int step, const tlfea::VectorXd& x, const tlfea::VectorXd& y, const tlfea::VectorXd& z) {
    if (vtu_dir.empty() || step % kVtuEvery != 0) return;
    std::ostringstream name;
    name << vtu_dir << "/ancf3243_" << solver_kind << "_" << std::setw(6) << std::setfill('0') << step << ".vtu";
    ANCFCPUUtils::VisualizationUtils::ExportANCF3243ToVTU(x, y, z, conn, kW, kH, name.str());
  };
  if (!vtu_dir.empty()) {
    std::filesystem::create_directories(vtu_dir);
    write_vtu(0, h_x12, h_y12, h_z12);
  }
  for (int step = 0; step < steps; ++step) {
    solver.Solve();
    tlfea::VectorXd x, y, z;
    data.RetrievePositionToCPU(x, y, z);
    write_vtu(step + 1, x, y, z);
    tip_z.push_back(z(tip_coef));
    std::cout << "Step " << step + 1 << ": tip z = " << std::setprecision(17) << z(tip_coef) << std::endl;
  }
  if (!csv_path.empty()) {
    std::ofstream csv(csv_path);
    csv << std::fixed << std::setprecision(17) << "step,tip_z\n";
    for (size_t i = 0; i < tip_z.size(); ++i) csv << i << "," << tip_z[i] << "\n";
  }
  data.Destroy();
  return 0;
}
