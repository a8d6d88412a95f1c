// test_feat10_resolution -- the reference's T10 beam driver flow (lib_bin/beam_sag/test_feat10_resolution.cc:
// 209-431) on the MI355X engine, same flags, solver kinds, parameters and CSV schema (`step,x_position`, 17 digits).
//   ./test_feat10_resolution --mesh_dir=tests/golden/meshes --res=2 --steps=5 --dt=1e-3 [--solver=adamw|newton|vbd|nesterov]
//                            [--omega=1.8] [--csv] [--csv_path=...]
#include <cmath>
#include <iomanip>
#include <limits>

#include "tlfea_facade.h"

namespace {
const double kE = 7e8, kNu = 0.33, kRho0 = 2700;  // test_feat10_resolution.cc:40-42
struct Options {
  int res = 0, steps = 50;
  double dt = 1e-3;
  bool write_csv = false;
  std::string csv_path, mesh_dir = "data/meshes/T10/resolution", solver = "adamw";  // default of the reference (:47)
  double omega = std::numeric_limits<double>::quiet_NaN();                          // VBD only (:50)
};
bool StartsWith(const std::string& s, const std::string& p) { return s.rfind(p, 0) == 0; }
bool ParseArgs(int argc, char** argv, Options& o) {
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (StartsWith(a, "--res=")) o.res = std::atoi(a.c_str() + 6);
    else if (StartsWith(a, "--steps=")) o.steps = std::atoi(a.c_str() + 8);
    else if (StartsWith(a, "--dt=")) o.dt = std::atof(a.c_str() + 5);
    else if (a == "--csv") o.write_csv = true;
    else if (StartsWith(a, "--csv_path=")) { o.csv_path = a.substr(11); o.write_csv = true; }
    else if (StartsWith(a, "--mesh_dir=")) o.mesh_dir = a.substr(11);
    else if (StartsWith(a, "--solver=")) {
      o.solver = a.substr(9);
      if (o.solver != "newton" && o.solver != "vbd" && o.solver != "adamw" && o.solver != "nesterov") { std::cerr << "Invalid --solver: " << o.solver << "\n"; return false; }
    }
    else if (StartsWith(a, "--omega=")) {
      o.omega = std::atof(a.c_str() + 8);
      if (!(o.omega > 0.0)) { std::cerr << "Invalid --omega: " << a.substr(8) << "\n"; return false; }
    }
    else { std::cerr << "Unknown argument: " << a << std::endl; return false; }
  }
  return true;
}
}  // namespace

int main(int argc, char** argv) {
  Options opt;
  if (!ParseArgs(argc, argv, opt)) return 1;
  if (tlfea_device_count() <= 0) {
    std::cerr << "No HIP device visible" << std::endl;
    return 1;
  }
  tlfea::MatrixXd nodes;
  tlfea::MatrixXi elements;
  const std::string res_str = std::to_string(opt.res);
  const int n_nodes = ANCFCPUUtils::FEAT10_read_nodes(opt.mesh_dir + "/beam_3x2x1_res" + res_str + ".1.node", nodes);
  const int n_elems = ANCFCPUUtils::FEAT10_read_elements(opt.mesh_dir + "/beam_3x2x1_res" + res_str + ".1.ele", elements);
  if (!n_nodes || !n_elems) return 1;
  int plot_target_node = 0;  // historical targets (:254-266)
  switch (opt.res) { case 0: plot_target_node = 23; break; case 2: plot_target_node = 89; break;
    case 4: plot_target_node = 353; break; case 8: plot_target_node = 1408; break; case 16: plot_target_node = 5630; break; }
  std::cout << "mesh read nodes: " << n_nodes << "\nmesh read elements: " << n_elems << std::endl;

  GPU_FEAT10_Data data(n_elems, n_nodes);
  data.Initialize();
  tlfea::VectorXd h_x12(n_nodes), h_y12(n_nodes), h_z12(n_nodes);
  for (int i = 0; i < n_nodes; i++) { h_x12(i) = nodes(i, 0); h_y12(i) = nodes(i, 1); h_z12(i) = nodes(i, 2); }
  std::vector<int> fixed;
  for (int i = 0; i < n_nodes; i++) if (std::abs(h_x12(i)) < 1e-8) fixed.push_back(i);
  tlfea::VectorXi h_fixed(static_cast<int>(fixed.size()));
  for (size_t i = 0; i < fixed.size(); i++) h_fixed(static_cast<int>(i)) = fixed[i];
  data.SetNodalFixed(h_fixed);
  tlfea::VectorXd h_f_ext(data.get_n_coef() * 3);
  std::vector<int> force_nodes;
  for (int i = 0; i < n_nodes; i++) if (std::abs(h_x12(i) - 3.0) < 1e-8) force_nodes.push_back(i);
  for (int n : force_nodes) h_f_ext(3 * n) = 5000.0 / force_nodes.size();
  data.SetExternalForce(h_f_ext);
  data.Setup(Quadrature::tet5pt_x, Quadrature::tet5pt_y, Quadrature::tet5pt_z, Quadrature::tet5pt_weights, h_x12,
             h_y12, h_z12, elements);
  data.SetDensity(kRho0);
  data.SetDamping(0.0, 0.0);
  data.SetSVK(kE, kNu);
  data.CalcDnDuPre();
  data.CalcMassMatrix();
  data.CalcConstraintData();
  data.ConvertToCSR_ConstraintJacT();
  data.BuildConstraintJacobianCSR();
  if (opt.solver != "adamw") {  // :330-333
    data.CalcP();
    data.CalcInternalForce();
  }

  std::ofstream csv;
  if (opt.write_csv) {
    csv.open(opt.csv_path.empty() ? "node_x_history_feat10_res" + res_str + "_" + opt.solver + ".csv" : opt.csv_path);
    csv << std::fixed << std::setprecision(17) << "step,x_position\n";
  }
  auto record_step = [&](int step) {
    tlfea::VectorXd x, y, z;
    data.RetrievePositionToCPU(x, y, z);
    std::cout << "Step " << step << ": node " << plot_target_node << " x = " << std::setprecision(17)
              << x(plot_target_node) << std::endl;
    if (opt.write_csv) csv << step << "," << x(plot_target_node) << "\n";
  };
  if (opt.solver == "newton") {
    SyncedNewtonParams params = {1e-4, 1e-4, 1e-4, 1e14, 5, 10, opt.dt};  // :365
    SyncedNewtonSolver solver(&data, data.get_n_constraint());
    solver.Setup();
    solver.SetParameters(&params);
    solver.AnalyzeHessianSparsity();
    solver.SetFixedSparsityPattern(true);
    for (int step = 0; step < opt.steps; ++step) {
      solver.Solve();
      record_step(step);
    }
  } else if (opt.solver == "vbd") {
    const double omega = std::isnan(opt.omega) ? 1.8 : opt.omega;  // :378-380
    SyncedVBDParams params = {1e-4, 1e-4, 1e-4, 1e14, 5, 500, opt.dt, omega, 1e-12, 25, 1};
    SyncedVBDSolver solver(&data, data.get_n_constraint());
    solver.Setup();
    solver.SetParameters(&params);
    solver.InitializeColoring();
    solver.InitializeMassDiagBlocks();
    solver.InitializeFixedMap();
    for (int step = 0; step < opt.steps; ++step) {
      solver.Solve();
      record_step(step);
    }
  } else if (opt.solver == "nesterov") {
    // not a kind of the reference's resolution driver: the parameters of its sibling lib_bin/beam_sag/test_feat10_nesterov.cc:181
    SyncedNesterovParams params = {1.0e-8, 1e14, 1.0e-6, 1.0e-6, 5, 300, opt.dt};
    SyncedNesterovSolver solver(&data, data.get_n_constraint());
    solver.Setup();
    solver.SetParameters(&params);
    for (int step = 0; step < opt.steps; ++step) {
      solver.Solve();
      record_step(step);
    }
  } else {
    // :394-416 (lr 2.5e-4 / decay 0.998 from res 8 up)
    const bool fine = opt.res >= 8;
    if (opt.res != 0 && opt.res != 2 && opt.res != 4 && opt.res != 8 && opt.res != 16) {
      std::cerr << "Unsupported resolution" << std::endl;
      return 1;
    }
    SyncedAdamWNocoopParams params = {fine ? 2.5e-4 : 2e-4, 0.9, 0.999, 1e-8, 1e-4, fine ? 0.998 : 0.995, 1e-1,
                                      1e-6, 1e14, 5, 800, opt.dt, 20, 1e-4};
    SyncedAdamWNocoopSolver solver(&data, data.get_n_constraint());
    solver.Setup();
    solver.SetParameters(&params);
    for (int step = 0; step < opt.steps; ++step) {
      solver.Solve();
      record_step(step);
    }
  }
  data.Destroy();
  return 0;
}
