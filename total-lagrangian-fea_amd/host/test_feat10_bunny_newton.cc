// Stanford-bunny T10 mesh under a push on its top: the flow of the reference's
// lib_bin/mesh_deform/test_feat10_bunny_newton.cc (:32-262) on the facade over the C-ABI (the source of BASELINE config
// B's material and solver parameters).  The reference hard-codes everything; the same values are the defaults here:
// 0-based TetGen mesh bunny_ascii_26.1 (1066 elements), E = 3e8, nu = 0.4, rho = 920, nodes with z < -4 pinned, -35 kN
// in z on every node with z > 4, released after 1000 steps, 8000 steps of dt = 1e-3, Newton {1e-4,1e-6,1e-4,1e14,5,10},
// a VTK file every 10 steps.
// --solver=adamw runs the sibling lib_bin/mesh_deform/test_feat10_bunny_adamw.cc instead (-2 N per loaded node, no
// release, 20000 steps, VTK every 100, AdamW {1e-8,0.9,0.999,1e-8,1e-4,0.998,1e-1,1e-6,1e14,5,500,1e-3,20,0}, :182-183).
//   ./test_feat10_bunny_newton --mesh_dir=tests/golden/meshes [--steps=8000] [--release_step=1000] [--material=svk|mr] [--solver=newton|adamw]
//                              [--vtk_dir=output] [--output_interval=10] [--csv_path=FILE] [--dump]
// --dump prints the reference gradients, detJ, P and f_int like the reference does; --csv_path records per step the
// top node's z and the largest displacement (for the tests).
#include <cmath>
#include <filesystem>
#include <iomanip>
#include <memory>

#include "tlfea_facade.h"

namespace {
const double kE = 3.0e8, kNu = 0.40, kRho0 = 920.0;  // :24-27
bool starts_with(const std::string& s, const std::string& p) { return s.rfind(p, 0) == 0; }
}  // namespace

int main(int argc, char** argv) {
  std::string mesh_dir = "data/meshes/T10", material = "svk", vtk_dir = "output", csv_path, solver_kind = "newton";
  int steps = -1, release_step = -2, output_interval = -1;
  bool dump = false;
  for (int i = 1; i < argc; i++) {
    const std::string a(argv[i]);
    if (starts_with(a, "--mesh_dir=")) mesh_dir = a.substr(11);
    else if (starts_with(a, "--steps=")) steps = std::atoi(a.c_str() + 8);
    else if (starts_with(a, "--release_step=")) release_step = std::atoi(a.c_str() + 15);
    else if (starts_with(a, "--material=")) material = a.substr(11);
    else if (starts_with(a, "--solver=")) solver_kind = a.substr(9);
    else if (starts_with(a, "--vtk_dir=")) vtk_dir = a.substr(10);
    else if (starts_with(a, "--output_interval=")) output_interval = std::atoi(a.c_str() + 18);
    else if (starts_with(a, "--csv_path=")) csv_path = a.substr(11);
    else if (a == "--dump") dump = true;
    else {
      std::cerr << "Unknown argument: " << a << std::endl;
      return 1;
    }
  }
  if (solver_kind != "newton" && solver_kind != "adamw") {
    std::cerr << "Invalid --solver (newton|adamw): " << solver_kind << std::endl;
    return 1;
  }
  const bool adamw = solver_kind == "adamw";
  if (steps < 0) steps = adamw ? 20000 : 8000;
  if (release_step == -2) release_step = adamw ? -1 : 1000;
  if (output_interval < 0) output_interval = adamw ? 100 : 10;
  if (material != "svk" && material != "mr") {
    std::cerr << "Invalid --material (svk|mr): " << material << std::endl;
    return 1;
  }
  if (tlfea_device_count() <= 0) {
    std::cerr << "No HIP device visible" << std::endl;
    return 1;
  }
  tlfea::MatrixXd nodes;
  tlfea::MatrixXi elements;
  const int n_nodes = ANCFCPUUtils::FEAT10_read_nodes(mesh_dir + "/bunny_ascii_26.1.node", nodes);
  const int n_elems = ANCFCPUUtils::FEAT10_read_elements(mesh_dir + "/bunny_ascii_26.1.ele", elements);
  if (n_nodes <= 0 || n_elems <= 0) return 1;
  std::cout << "mesh read nodes: " << n_nodes << "\nmesh read elements: " << n_elems << std::endl;

  GPU_FEAT10_Data data(n_elems, n_nodes);
  data.Initialize();
  tlfea::VectorXd x(n_nodes), y(n_nodes), z(n_nodes);
  std::vector<int> fixed;
  int top = 0;
  for (int i = 0; i < n_nodes; i++) {
    x(i) = nodes(i, 0); y(i) = nodes(i, 1); z(i) = nodes(i, 2);
    if (z(i) < -4.0) fixed.push_back(i);  // :58-62
    if (z(i) > z(top)) top = i;
  }
  tlfea::VectorXi h_fixed(static_cast<int>(fixed.size()));
  for (size_t i = 0; i < fixed.size(); i++) h_fixed(static_cast<int>(i)) = fixed[i];
  std::cout << "Fixed nodes (z < -4.0): " << fixed.size() << std::endl;
  data.SetNodalFixed(h_fixed);
  tlfea::VectorXd f_ext(3 * n_nodes);
  int loaded = 0;
  for (int i = 0; i < n_nodes; i++)
    if (z(i) > 4.0) {  // :80-85
      f_ext(3 * i + 2) = adamw ? -2.0 : -35000.0;  // bunny_adamw.cc:95
      loaded++;
    }
  std::cout << "Loaded nodes (z > 4.0): " << loaded << std::endl;
  data.SetExternalForce(f_ext);
  data.Setup(Quadrature::tet5pt_x, Quadrature::tet5pt_y, Quadrature::tet5pt_z, Quadrature::tet5pt_weights, x, y, z, elements);
  data.SetDensity(kRho0);
  data.SetDamping(0.0, 0.0);
  if (material == "svk") {
    data.SetSVK(kE, kNu);
    std::cout << "Material: SVK" << std::endl;
  } else {  // :112-122
    const double mu = kE / (2.0 * (1.0 + kNu)), K = kE / (3.0 * (1.0 - 2.0 * kNu));
    data.SetMooneyRivlin(0.30 * mu, 0.20 * mu, 1.5 * K);
    std::cout << "Material: Mooney-Rivlin" << std::endl;
  }
  data.CalcDnDuPre();
  data.CalcMassMatrix();
  data.CalcConstraintData();
  data.ConvertToCSR_ConstraintJacT();
  data.BuildConstraintJacobianCSR();
  data.CalcP();
  data.CalcInternalForce();
  if (dump) {
    std::vector<std::vector<tlfea::MatrixXd>> g, P;
    std::vector<std::vector<double>> dj;
    data.RetrieveDnDuPreToCPU(g);
    data.RetrieveDetJToCPU(dj);
    data.RetrievePFromFToCPU(P);
    for (int e = 0; e < n_elems; e++)
      for (int q = 0; q < 5; q++) {
        std::cout << "Element " << e << " Quadrature Point " << q << " detJ " << dj[e][q] << "\n";
        for (int a = 0; a < 10; a++) std::cout << g[e][q](a, 0) << " " << g[e][q](a, 1) << " " << g[e][q](a, 2) << "\n";
        for (int r = 0; r < 3; r++) std::cout << P[e][q](r, 0) << " " << P[e][q](r, 1) << " " << P[e][q](r, 2) << "\n";
      }
    tlfea::VectorXd fi;
    data.RetrieveInternalForceToCPU(fi);
    std::cout << "Internal force vector (size: " << fi.size() << "):\n";
    for (int i = 0; i < fi.size(); i++) std::cout << fi(i) << " ";
    std::cout << std::endl;
  }
  std::unique_ptr<SolverBase> solver_ptr;
  if (adamw) {
    SyncedAdamWParams p = {1e-8, 0.9, 0.999, 1e-8, 1e-4, 0.998, 1e-1, 1e-6, 1e14, 5, 500, 1e-3, 20, 0.0};
    auto* sv = new SyncedAdamWSolver(&data, data.get_n_constraint());  // the cooperative class, as in the reference
    sv->Setup();
    sv->SetParameters(&p);
    solver_ptr.reset(sv);
  } else {
    SyncedNewtonParams params = {1e-4, 1e-6, 1e-4, 1e14, 5, 10, 1e-3};  // :201
    auto* sv = new SyncedNewtonSolver(&data, data.get_n_constraint());
    sv->Setup();
    sv->SetParameters(&params);
    sv->AnalyzeHessianSparsity();
    sv->SetFixedSparsityPattern(true);
    solver_ptr.reset(sv);
  }
  SolverBase& solver = *solver_ptr;
  if (output_interval > 0 && !vtk_dir.empty()) std::filesystem::create_directories(vtk_dir);
  std::ofstream csv;
  if (!csv_path.empty()) {
    csv.open(csv_path);
    csv << std::setprecision(17) << "step,top_z,max_disp\n";
  }
  int frame = 0;
  tlfea::VectorXd xx, yy, zz;
  for (int i = 0; i < steps; i++) {
    if (i == release_step) {  // :213-218
      tlfea::VectorXd zero(3 * n_nodes);
      data.SetExternalForce(zero);
      std::cout << "External force reset to zero at step " << i << std::endl;
    }
    solver.Solve();
    if (output_interval > 0 && !vtk_dir.empty() && i % output_interval == 0)
      data.WriteOutputVTK(vtk_dir + "/bunny_" + solver_kind + "_step_" + std::to_string(frame++) + ".vtk");
    if (csv.is_open()) {
      data.RetrievePositionToCPU(xx, yy, zz);
      double md = 0.0;
      for (int n = 0; n < n_nodes; n++)
        md = std::max(md, std::sqrt((xx(n) - x(n)) * (xx(n) - x(n)) + (yy(n) - y(n)) * (yy(n) - y(n)) + (zz(n) - z(n)) * (zz(n) - z(n))));
      csv << i << "," << zz(top) << "," << md << "\n";
    }
  }
  data.RetrievePositionToCPU(xx, yy, zz);
  std::cout << std::fixed << std::setprecision(17) << "top node " << top << " z: " << zz(top) << std::endl;
  solver_ptr.reset();
  data.Destroy();
  return 0;
}
