// tlfea_facade.h -- header-only C++ facade re-creating the reference's class surface on the C-ABI
// (include/tlfea_c.h): ElementBase / GPU_FEAT10_Data (lib_src/elements/ElementBase.h:20-50,
// FEAT10Data.cuh:306-852), SolverBase / SyncedNewtonParams / SyncedNewtonSolver
// (lib_src/solvers/SolverBase.h:16-23, SyncedNewton.cuh:29-405), Quadrature::tet5pt_*
// (lib_utils/quadrature_utils.h:134-158) and ANCFCPUUtils::FEAT10_read_* (lib_utils/cpu_utils.cc:607-754).
// Same method names, argument order and call-order contract; Eigen types become tlfea::VectorXd etc.
// Error convention of the reference (lib_utils/cuda_utils.h:12-18): device/library failure -> message + exit.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <initializer_list>
#include <set>
#include <stdexcept>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/tlfea_c.h"
#include "tlfea_containers.h"

#define TLFEA_HANDLE_ERROR(call)                                                              \
  do {                                                                                        \
    if ((call) != 0) {                                                                        \
      std::fprintf(stderr, "%s in %s at line %d\n", tlfea_last_error(), __FILE__, __LINE__);  \
      std::exit(EXIT_FAILURE);                                                                \
    }                                                                                         \
  } while (0)
// API misuse: the reference prints and returns (e.g. FEAT10Data.cuh:442-445); the C-ABI already printed.
#define TLFEA_SOFT(call) (void)(call)

namespace Quadrature {  // quadrature_utils.h:134-158
constexpr int N_QP_T10_5 = 5;
constexpr int N_NODE_T10_10 = 10;
inline tlfea::VectorXd make5(const double (&a)[5]) {
  tlfea::VectorXd v(5);
  for (int i = 0; i < 5; i++) v(i) = a[i];
  return v;
}
static const double kB = 1.0 / 6.0;
static const tlfea::VectorXd tet5pt_x = make5({0.25, kB, 0.5, kB, kB});
static const tlfea::VectorXd tet5pt_y = make5({0.25, kB, kB, 0.5, kB});
static const tlfea::VectorXd tet5pt_z = make5({0.25, kB, kB, kB, 0.5});
static const tlfea::VectorXd tet5pt_weights =
    make5({-4.0 / 5.0 * kB, 9.0 / 20.0 * kB, 9.0 / 20.0 * kB, 9.0 / 20.0 * kB, 9.0 / 20.0 * kB});
// barycentric rows [L1..L4] and (xi, eta, zeta) = columns 1..3 of the Keast rule, and the struct that wraps them (:140-171)
inline tlfea::MatrixXd make_tet5pt_bary() {
  tlfea::MatrixXd m(5, 4);
  for (int q = 0; q < 5; q++)
    for (int c = 0; c < 4; c++) m(q, c) = q == 0 ? 0.25 : (c == q - 1 ? 0.5 : kB);
  return m;
}
inline tlfea::MatrixXd make_tet5pt_xyz() {
  tlfea::MatrixXd m(5, 3);
  for (int q = 0; q < 5; q++) {
    m(q, 0) = tet5pt_x(q);
    m(q, 1) = tet5pt_y(q);
    m(q, 2) = tet5pt_z(q);
  }
  return m;
}
static const tlfea::MatrixXd tet5pt_bary = make_tet5pt_bary();
static const tlfea::MatrixXd tet5pt_xyz = make_tet5pt_xyz();
struct Tet5ptQuadrature {
  static constexpr int n_points = N_QP_T10_5;
  static const tlfea::MatrixXd& barycentric() { return tet5pt_bary; }
  static const tlfea::MatrixXd& xyz() { return tet5pt_xyz; }
  static const tlfea::VectorXd& weights() { return tet5pt_weights; }
};
// Gauss-Legendre tables of the ANCF elements (quadrature_utils.h:8-128)
constexpr int N_QP_2 = 2, N_QP_3 = 3, N_QP_4 = 4, N_QP_5 = 5, N_QP_6 = 6, N_QP_7 = 7;
constexpr int N_SHAPE_3243 = 8, N_SHAPE_3443 = 16, N_TOTAL_QP_3_2_2 = 12, N_TOTAL_QP_4_4_3 = 48, N_TOTAL_QP_7_7_3 = 147;
inline tlfea::VectorXd makev(std::initializer_list<double> a) {
  tlfea::VectorXd v(static_cast<int>(a.size()));
  int i = 0;
  for (double x : a) v(i++) = x;
  return v;
}
static const tlfea::VectorXd gauss_xi_m_6 = makev({-0.93246951420315202, -0.66120938646626451, -0.23861918608319691,
                                                   0.23861918608319691, 0.66120938646626451, 0.93246951420315202});
static const tlfea::VectorXd weight_xi_m_6 = makev({0.17132449237917034, 0.36076157304813861, 0.46791393457269104,
                                                    0.46791393457269104, 0.36076157304813861, 0.17132449237917034});
static const tlfea::VectorXd gauss_xi_m_7 = makev({-0.949107912342759, -0.741531185599394, -0.405845151377397, 0.0,
                                                   0.405845151377397, 0.741531185599394, 0.949107912342759});
static const tlfea::VectorXd weight_xi_m_7 = makev({0.129484966168870, 0.279705391489277, 0.381830050505119,
                                                    0.417959183673469, 0.381830050505119, 0.279705391489277,
                                                    0.129484966168870});
static const tlfea::VectorXd gauss_eta_m_7 = gauss_xi_m_7, weight_eta_m_7 = weight_xi_m_7;
static const tlfea::VectorXd gauss_zeta_m_3 = makev({-0.7745966692414834, 0.0, 0.7745966692414834});
static const tlfea::VectorXd weight_zeta_m_3 = makev({0.5555555555555556, 0.8888888888888888, 0.5555555555555556});
static const tlfea::VectorXd gauss_xi_3 = makev({-0.77459666924148340, 0.0, 0.77459666924148340});
static const tlfea::VectorXd weight_xi_3 = makev({0.55555555555555556, 0.88888888888888889, 0.55555555555555556});
static const tlfea::VectorXd gauss_xi_4 = makev({-0.8611363115940526, -0.3399810435848563, 0.3399810435848563,
                                                 0.8611363115940526});
static const tlfea::VectorXd weight_xi_4 = makev({0.3478548451374538, 0.6521451548625461, 0.6521451548625461,
                                                  0.3478548451374538});
static const tlfea::VectorXd gauss_eta_2 = makev({-0.57735026918962576, 0.57735026918962576});
static const tlfea::VectorXd weight_eta_2 = makev({1.0, 1.0});
static const tlfea::VectorXd gauss_eta_4 = gauss_xi_4, weight_eta_4 = weight_xi_4;
static const tlfea::VectorXd gauss_zeta_2 = gauss_eta_2, weight_zeta_2 = weight_eta_2;
static const tlfea::VectorXd gauss_zeta_3 = gauss_xi_3, weight_zeta_3 = weight_xi_3;
}  // namespace Quadrature

namespace ANCFCPUUtils {  // cpu_utils.cc:607-754, mesh_utils.cc:13-167
inline void FEAT10_remap_tetgen_indices(const tlfea::VectorXi& tetgen_elem, tlfea::VectorXi& standard_elem) {
  if (tetgen_elem.size() != 10 || standard_elem.size() != 10) {
    std::cerr << "Error: Element arrays must have size 10 for T10 elements" << std::endl;
    return;
  }
  static const int map[10] = {0, 1, 2, 3, 6, 7, 9, 5, 8, 4};
  for (int i = 0; i < 10; i++) standard_elem(i) = tetgen_elem(map[i]);
}

inline int FEAT10_read_nodes(const std::string& filename, tlfea::MatrixXd& nodes) {
  std::ifstream file(filename);
  if (!file.is_open()) {
    std::cerr << "Error: Could not open node file " << filename << std::endl;
    return 0;
  }
  std::string line;
  if (!std::getline(file, line)) return 0;
  std::istringstream header(line);
  int n_nodes = 0, dim = 0;
  header >> n_nodes >> dim;
  if (dim != 3) {
    std::cerr << "Error: Only 3D nodes are supported, found " << dim << "D" << std::endl;
    return 0;
  }
  nodes.resize(n_nodes, 3);
  int min_id = INT_MAX;
  std::vector<std::tuple<int, double, double, double>> rows;
  for (int i = 0; i < n_nodes; i++) {
    if (!std::getline(file, line) || line.empty()) continue;
    std::istringstream iss(line);
    int id;
    double x, y, z;
    if (iss >> id >> x >> y >> z) {
      min_id = std::min(min_id, id);
      rows.emplace_back(id, x, y, z);
    }
  }
  const int off = (min_id == 0) ? 0 : 1;  // adaptive 0/1-based ids
  for (const auto& r : rows) {
    const int idx = std::get<0>(r) - off;
    if (idx >= 0 && idx < n_nodes) {
      nodes(idx, 0) = std::get<1>(r);
      nodes(idx, 1) = std::get<2>(r);
      nodes(idx, 2) = std::get<3>(r);
    }
  }
  return n_nodes;
}

inline int FEAT10_read_elements(const std::string& filename, tlfea::MatrixXi& elements) {
  std::ifstream file(filename);
  if (!file.is_open()) {
    std::cerr << "Error: Could not open element file " << filename << std::endl;
    return 0;
  }
  std::string line;
  if (!std::getline(file, line)) return 0;
  std::istringstream header(line);
  int n_elements = 0, npe = 0;
  header >> n_elements >> npe;
  if (npe != 10) {
    std::cerr << "Error: Only T10 elements (10 nodes) are supported, found " << npe << std::endl;
    return 0;
  }
  elements.resize(n_elements, 10);
  int min_e = INT_MAX, min_n = INT_MAX;
  std::vector<std::vector<int>> rows;
  for (int i = 0; i < n_elements; i++) {
    if (!std::getline(file, line) || line.empty()) continue;
    std::istringstream iss(line);
    std::vector<int> r(11, 0);
    iss >> r[0];
    min_e = std::min(min_e, r[0]);
    for (int j = 0; j < 10; j++)
      if (iss >> r[1 + j]) min_n = std::min(min_n, r[1 + j]);
    rows.push_back(r);
  }
  const int eoff = (min_e == 0) ? 0 : 1, noff = (min_n == 0) ? 0 : 1;
  tlfea::VectorXi t(10), s(10);
  for (const auto& r : rows) {
    for (int j = 0; j < 10; j++) t(j) = r[1 + j] - noff;
    FEAT10_remap_tetgen_indices(t, s);
    const int e = r[0] - eoff;
    if (e >= 0 && e < n_elements)
      for (int j = 0; j < 10; j++) elements(e, j) = s(j);
  }
  return n_elements;
}
// GridMeshGenerator (mesh_utils.cc:13-167): ANCF-3243 beam / net meshes
class GridMeshGenerator {
 public:
  GridMeshGenerator(double X, double Y, double L, bool include_horizontal = true, bool include_vertical = true)
      : L_(L), h_(include_horizontal), v_(include_vertical) {
    if (L <= 0) throw std::invalid_argument("L must be > 0");
    if (std::abs(std::round(X / L) * L - X) > 1e-12 || std::abs(std::round(Y / L) * L - Y) > 1e-12)
      throw std::invalid_argument("X and Y must be exact multiples of L");
    nx_ = static_cast<int>(std::round(X / L));
    ny_ = static_cast<int>(std::round(Y / L));
    if (h_ && nx_ == 0) h_ = false;
    if (v_ && ny_ == 0) v_ = false;
  }
  void generate_mesh() {
    elems_.clear();
    if (h_)
      for (int j = 0; j <= ny_; j++)
        for (int i = 0; i < nx_; i++) elems_.push_back({node_id(i, j), node_id(i + 1, j)});
    if (v_)
      for (int i = 0; i <= nx_; i++)
        for (int j = 0; j < ny_; j++) elems_.push_back({node_id(i, j), node_id(i, j + 1)});
  }
  int node_id(int i, int j) const {
    if (i < 0 || i > nx_ || j < 0 || j > ny_) throw std::out_of_range("(i,j) out of range");
    return j * (nx_ + 1) + i;
  }
  int get_num_nodes() const { return (nx_ + 1) * (ny_ + 1); }
  int get_num_elements() const { return static_cast<int>(elems_.size()); }
  void get_coordinates(tlfea::VectorXd& x, tlfea::VectorXd& y, tlfea::VectorXd& z) {
    const int n = get_num_nodes();
    x.resize(4 * n);
    y.resize(4 * n);
    z.resize(4 * n);
    for (int j = 0; j <= ny_; j++)
      for (int i = 0; i <= nx_; i++) {
        const int b = 4 * node_id(i, j);
        x(b) = i * L_; x(b + 1) = 1.0;
        y(b) = 1.0; y(b + 2) = 1.0;
        z(b + 3) = 1.0;
      }
  }
  void get_element_connectivity(tlfea::MatrixXi& c) {
    c.resize(get_num_elements(), 2);
    for (int e = 0; e < get_num_elements(); e++) {
      c(e, 0) = elems_[e].first;
      c(e, 1) = elems_[e].second;
    }
  }

 private:
  double L_;
  bool h_, v_;
  int nx_ = 0, ny_ = 0;
  std::vector<std::pair<int, int>> elems_;
};

// Vertex colouring helpers of SyncedVBDSolver (cpu_utils.h:13-58, cpu_utils.cc:18-123).  element_connectivity:
// n_elem x nodes_per_elem; greedy colouring in std::sort's "degree descending" order, as the reference.
inline std::vector<std::set<int>> BuildVertexAdjacency(const tlfea::MatrixXi& element_connectivity, int n_nodes) {
  std::vector<std::set<int>> adj(static_cast<size_t>(n_nodes));
  for (int e = 0; e < element_connectivity.rows(); e++)
    for (int i = 0; i < element_connectivity.cols(); i++)
      for (int j = i + 1; j < element_connectivity.cols(); j++) {
        const int a = element_connectivity(e, i), b = element_connectivity(e, j);
        adj[a].insert(b);
        adj[b].insert(a);
      }
  return adj;
}
inline tlfea::VectorXi GreedyVertexColoring(const std::vector<std::set<int>>& adjacency) {
  const int n = static_cast<int>(adjacency.size());
  std::vector<int> degrees(static_cast<size_t>(n)), order(static_cast<size_t>(n));
  for (int i = 0; i < n; i++) {
    degrees[i] = static_cast<int>(adjacency[i].size());
    order[i] = i;
  }
  std::sort(order.begin(), order.end(), [&degrees](int a, int b) { return degrees[a] > degrees[b]; });
  tlfea::VectorXi colors(n);
  for (int i = 0; i < n; i++) colors(i) = -1;
  std::vector<int> taken(static_cast<size_t>(n) + 1, -1);  // taken[c] == v: colour c is used around v
  for (int v : order) {
    for (int nb : adjacency[v])
      if (colors(nb) >= 0) taken[colors(nb)] = v;
    int c = 0;
    while (taken[c] == v) ++c;
    colors(v) = c;
  }
  return colors;
}
inline bool ValidateColoring(const tlfea::MatrixXi& element_connectivity, const tlfea::VectorXi& colors) {
  for (int e = 0; e < element_connectivity.rows(); e++) {
    std::set<int> seen;
    for (int i = 0; i < element_connectivity.cols(); i++) {
      const int c = colors(element_connectivity(e, i));
      if (!seen.insert(c).second) {
        std::cerr << "Invalid coloring: element " << e << " has duplicate color " << c << std::endl;
        return false;
      }
    }
  }
  return true;
}
inline std::vector<std::vector<std::pair<int, int>>> BuildNodeIncidence(const tlfea::MatrixXi& element_connectivity,
                                                                        int n_nodes) {
  std::vector<std::vector<std::pair<int, int>>> inc(static_cast<size_t>(n_nodes));
  for (int e = 0; e < element_connectivity.rows(); e++)
    for (int a = 0; a < element_connectivity.cols(); a++) inc[element_connectivity(e, a)].push_back({e, a});
  return inc;
}
inline std::vector<std::vector<int>> BuildColorToNodes(const tlfea::VectorXi& colors, int n_colors) {
  std::vector<std::vector<int>> out(static_cast<size_t>(n_colors));
  for (int i = 0; i < colors.size(); i++)
    if (colors(i) >= 0 && colors(i) < n_colors) out[colors(i)].push_back(i);
  return out;
}

// ANCF3243_B12_matrix / ANCF3443_B12_matrix and their per-element packers (cpu_utils.cc:125-209, 211-441): (B^T)^-1
inline void ANCF_B12_matrix_impl(int kind, double L, double W, double H, tlfea::MatrixXd& B_inv_out, int n_shape) {
  if (n_shape != (kind == 3243 ? 8 : 16)) throw std::invalid_argument("ANCF B12 matrix: n_shape does not fit the element");
  B_inv_out.resize(n_shape, n_shape);
  TLFEA_HANDLE_ERROR(tlfea_ancf_b12_matrix(kind, L, W, H, B_inv_out.data()));
}
inline void ANCF3243_B12_matrix(double L, double W, double H, tlfea::MatrixXd& B_inv_out, int n_shape) {
  ANCF_B12_matrix_impl(3243, L, W, H, B_inv_out, n_shape);
}
inline void ANCF3443_B12_matrix(double L, double W, double H, tlfea::MatrixXd& B_inv_out, int n_shape) {
  ANCF_B12_matrix_impl(3443, L, W, H, B_inv_out, n_shape);
}
inline void ANCF_B12_flat_impl(int kind, const tlfea::VectorXd& L, const tlfea::VectorXd& W, const tlfea::VectorXd& H,
                               tlfea::VectorXd& B_inv_flat_out, int n_shape) {
  if (W.size() != L.size() || H.size() != L.size()) throw std::invalid_argument("ANCF B12 flat: L, W, H sizes differ");
  if (n_shape != (kind == 3243 ? 8 : 16)) throw std::invalid_argument("ANCF B12 flat: n_shape does not fit the element");
  B_inv_flat_out.resize(L.size() * n_shape * n_shape);
  for (int e = 0; e < L.size(); e++)
    TLFEA_HANDLE_ERROR(tlfea_ancf_b12_matrix(kind, L(e), W(e), H(e), B_inv_flat_out.data() + (size_t)e * n_shape * n_shape));
}
inline void ANCF3243_B12_matrix_flat_per_element(const tlfea::VectorXd& L, const tlfea::VectorXd& W,
                                                 const tlfea::VectorXd& H, tlfea::VectorXd& out, int n_shape) {
  ANCF_B12_flat_impl(3243, L, W, H, out, n_shape);
}
inline void ANCF3443_B12_matrix_flat_per_element(const tlfea::VectorXd& L, const tlfea::VectorXd& W,
                                                 const tlfea::VectorXd& H, tlfea::VectorXd& out, int n_shape) {
  ANCF_B12_flat_impl(3443, L, W, H, out, n_shape);
}

// ANCF3243_generate_beam_coordinates (cpu_utils.cc:443-474): chain of n_beam beams of length 2 along x; x12/y12/z12
// must already hold 4 * (n_beam + 1) entries, as in the reference
inline void ANCF3243_generate_beam_coordinates(int n_beam, tlfea::VectorXd& x12, tlfea::VectorXd& y12,
                                               tlfea::VectorXd& z12) {
  for (int n = 0; n <= n_beam; n++) {
    const int b = 4 * n;
    x12(b) = -1.0 + 2.0 * n; x12(b + 1) = 1.0; x12(b + 2) = 0.0; x12(b + 3) = 0.0;
    y12(b) = 1.0; y12(b + 1) = 0.0; y12(b + 2) = 1.0; y12(b + 3) = 0.0;
    z12(b) = 0.0; z12(b + 1) = 0.0; z12(b + 2) = 0.0; z12(b + 3) = 1.0;
  }
}
// ANCF3243_calculate_offsets (cpu_utils.cc:597-605; pinned by lib_utest/utest_utils.cc:32-108)
inline void ANCF3243_calculate_offsets(int n_beam, tlfea::VectorXi& offset_start, tlfea::VectorXi& offset_end) {
  offset_start.resize(n_beam);
  offset_end.resize(n_beam);
  for (int i = 0; i < n_beam; i++) {
    offset_start(i) = 4 * i;
    offset_end(i) = 4 * i + 7;
  }
}

// ANCF3443_generate_beam_coordinates (cpu_utils.cc:476-595)
inline void ANCF3443_generate_beam_coordinates(int n_beam, tlfea::VectorXd& x12, tlfea::VectorXd& y12,
                                               tlfea::VectorXd& z12, tlfea::MatrixXi& conn) {
  const int n_nodes = 4 + 2 * (n_beam - 1);
  x12.resize(4 * n_nodes);
  y12.resize(4 * n_nodes);
  z12.resize(4 * n_nodes);
  std::vector<std::pair<double, double>> pos = {{0, 0}, {2, 0}, {2, 1}, {0, 1}};
  for (int i = 1; i < n_beam; i++) {
    pos.push_back({2.0 * (i + 1), 0.0});
    pos.push_back({2.0 * (i + 1), 1.0});
  }
  for (int n = 0; n < n_nodes; n++) {
    x12(4 * n) = pos[n].first; x12(4 * n + 1) = 1.0;
    y12(4 * n) = pos[n].second; y12(4 * n + 2) = 1.0;
    z12(4 * n + 3) = 1.0;
  }
  conn.resize(n_beam, 4);
  conn(0, 0) = 0; conn(0, 1) = 1; conn(0, 2) = 2; conn(0, 3) = 3;
  for (int i = 1; i < n_beam; i++) {
    if (i == 1) {
      conn(i, 0) = 1; conn(i, 1) = 4; conn(i, 2) = 5; conn(i, 3) = 2;
    } else {
      conn(i, 0) = 4 + (i - 2) * 2; conn(i, 1) = 4 + (i - 1) * 2; conn(i, 2) = 4 + (i - 1) * 2 + 1; conn(i, 3) = 5 + (i - 2) * 2;
    }
  }
}

// ---- general linear constraints + ANCF mesh files (mesh_utils.h:100-245, mesh_utils.cc:170-1010) ----------------
struct LinearConstraintCSR {  // c[row] = sum_j values[j] * dof(columns[j]) - rhs[row]; columns = 3*coef + component
  std::vector<int> offsets, columns;
  std::vector<double> values;
  tlfea::VectorXd rhs;
  int NumRows() const { return rhs.size(); }
  int NumNonZeros() const { return static_cast<int>(columns.size()); }
  bool Empty() const { return rhs.size() == 0; }
};

class LinearConstraintBuilder {  // mesh_utils.cc:173-246
 public:
  explicit LinearConstraintBuilder(int n_dofs) : n_dofs_(n_dofs) {
    if (n_dofs_ <= 0) throw std::invalid_argument("LinearConstraintBuilder: n_dofs must be > 0");
    offsets_.push_back(0);
  }
  LinearConstraintBuilder(int n_dofs, const LinearConstraintCSR& initial)
      : n_dofs_(n_dofs), offsets_(initial.offsets), columns_(initial.columns), values_(initial.values) {
    if (n_dofs_ <= 0) throw std::invalid_argument("LinearConstraintBuilder: n_dofs must be > 0");
    if (static_cast<int>(offsets_.size()) != initial.NumRows() + 1)
      throw std::invalid_argument("LinearConstraintBuilder: initial offsets size mismatch");
    if (static_cast<int>(values_.size()) != initial.NumNonZeros())
      throw std::invalid_argument("LinearConstraintBuilder: initial nnz mismatch");
    if (offsets_.empty() || offsets_.front() != 0 || offsets_.back() != static_cast<int>(columns_.size()))
      throw std::invalid_argument("LinearConstraintBuilder: initial CSR offsets invalid");
    for (int i = 0; i < initial.rhs.size(); i++) rhs_.push_back(initial.rhs(i));
  }
  int n_dofs() const { return n_dofs_; }
  int num_rows() const { return static_cast<int>(rhs_.size()); }
  int nnz() const { return static_cast<int>(columns_.size()); }
  int AddRow(const std::vector<std::pair<int, double>>& entries, double rhs) {
    if (entries.empty()) throw std::invalid_argument("LinearConstraintBuilder::AddRow: empty row");
    for (const auto& e : entries) {
      if (e.first < 0 || e.first >= n_dofs_) throw std::out_of_range("LinearConstraintBuilder::AddRow: col out of range");
      if (e.second == 0.0) continue;
      columns_.push_back(e.first);
      values_.push_back(e.second);
    }
    rhs_.push_back(rhs);
    offsets_.push_back(static_cast<int>(columns_.size()));
    return static_cast<int>(rhs_.size()) - 1;
  }
  int AddFixedDof(int col, double rhs) { return AddRow({{col, 1.0}}, rhs); }
  LinearConstraintCSR ToCSR() const {
    LinearConstraintCSR out;
    out.offsets = offsets_;
    out.columns = columns_;
    out.values = values_;
    out.rhs.resize(static_cast<int>(rhs_.size()));
    for (size_t i = 0; i < rhs_.size(); i++) out.rhs(static_cast<int>(i)) = rhs_[i];
    return out;
  }

 private:
  int n_dofs_;
  std::vector<int> offsets_, columns_;
  std::vector<double> values_, rhs_;
};

inline int ANCFDofCol(int node_id, int coef_slot, int component) { return (node_id * 4 + coef_slot) * 3 + component; }
// r(b,slot) - r(a,slot) = 0  (mesh_utils.cc:262-275, :330-343)
inline void AppendANCF3243VectorEqualityConstraint(LinearConstraintBuilder& b, int node_a, int node_b, int coef_slot) {
  if (coef_slot < 0 || coef_slot > 3) throw std::out_of_range("AppendANCF3243VectorEqualityConstraint: coef_slot out of range");
  for (int c = 0; c < 3; ++c)
    b.AddRow({{ANCFDofCol(node_b, coef_slot, c), 1.0}, {ANCFDofCol(node_a, coef_slot, c), -1.0}}, 0.0);
}
// r(b,slot) - Q r(a,slot) = 0, Q row-major 3x3  (mesh_utils.cc:277-299, :345-367)
inline void AppendANCF3243VectorWeldedConstraint(LinearConstraintBuilder& b, int node_a, int node_b, int coef_slot,
                                                 const double Q[9]) {
  if (coef_slot < 0 || coef_slot > 3) throw std::out_of_range("AppendANCF3243VectorWeldedConstraint: coef_slot out of range");
  for (int row = 0; row < 3; ++row) {
    std::vector<std::pair<int, double>> entries;
    entries.push_back({ANCFDofCol(node_b, coef_slot, row), 1.0});
    for (int k = 0; k < 3; ++k) {
      const double w = -Q[3 * row + k];
      if (w == 0.0) continue;
      entries.push_back({ANCFDofCol(node_a, coef_slot, k), w});
    }
    b.AddRow(entries, 0.0);
  }
}
inline void AppendANCF3443VectorEqualityConstraint(LinearConstraintBuilder& b, int a, int nb, int slot) {
  AppendANCF3243VectorEqualityConstraint(b, a, nb, slot);
}
inline void AppendANCF3443VectorWeldedConstraint(LinearConstraintBuilder& b, int a, int nb, int slot, const double Q[9]) {
  AppendANCF3243VectorWeldedConstraint(b, a, nb, slot, Q);
}
// component-wise equality of one coefficient to the reference arrays (mesh_utils.cc:301-315)
inline void AppendANCF3243FixedCoefficient(LinearConstraintBuilder& b, int coef_index, const tlfea::VectorXd& x12_ref,
                                           const tlfea::VectorXd& y12_ref, const tlfea::VectorXd& z12_ref) {
  if (coef_index < 0 || coef_index >= x12_ref.size() || coef_index >= y12_ref.size() || coef_index >= z12_ref.size())
    throw std::out_of_range("AppendANCF3243FixedCoefficient: coef_index out of range");
  b.AddFixedDof(coef_index * 3 + 0, x12_ref(coef_index));
  b.AddFixedDof(coef_index * 3 + 1, y12_ref(coef_index));
  b.AddFixedDof(coef_index * 3 + 2, z12_ref(coef_index));
}

struct ANCF3243Mesh {  // mesh_utils.h:165-184 (std::optional members are plain values + a has_grid flag here)
  int version = 0;
  bool has_grid = false;
  int grid_nx = 0, grid_ny = 0;
  double grid_L = 0.0, grid_origin[3] = {0, 0, 0};
  int n_nodes = 0, n_elements = 0;
  std::vector<std::string> node_family;
  tlfea::VectorXd x12, y12, z12;
  tlfea::MatrixXi element_connectivity;  // n_elements x 2
  LinearConstraintCSR constraints;
};
struct ANCF3443Mesh {  // mesh_utils.h:194-214
  int version = 0;
  int n_nodes = 0, n_elements = 0;
  std::vector<std::string> node_family, element_family;
  tlfea::VectorXd x12, y12, z12, element_L, element_W, element_H;
  tlfea::MatrixXi element_connectivity;  // n_elements x 4
  LinearConstraintCSR constraints;
};

namespace detail {
inline bool next_record(std::ifstream& f, std::vector<std::string>& tok) {
  std::string line;
  while (std::getline(f, line)) {
    const size_t p = line.find('#');
    if (p != std::string::npos) line.erase(p);
    std::istringstream iss(line);
    tok.clear();
    std::string w;
    while (iss >> w) tok.push_back(w);
    if (!tok.empty()) return true;
  }
  return false;
}
inline bool to_int(const std::string& s, int& out) {
  try {
    size_t idx = 0;
    const int v = std::stoi(s, &idx);
    if (idx != s.size()) return false;
    out = v;
    return true;
  } catch (...) {
    return false;
  }
}
inline bool to_double(const std::string& s, double& out) {
  try {
    size_t idx = 0;
    const double v = std::stod(s, &idx);
    if (idx != s.size()) return false;
    out = v;
    return true;
  } catch (...) {
    return false;
  }
}
// shared body of the two readers: nn = nodes per element (2 | 4); 3443 element lines carry L W H
template <class Mesh>
bool read_ancf_mesh(const std::string& path, const char* tag, int nn, Mesh& out, std::string* error, bool* has_grid,
                    int* gnx, int* gny, double* gL, double* gorigin, tlfea::VectorXd* eL, tlfea::VectorXd* eW,
                    tlfea::VectorXd* eH, std::vector<std::string>* efam) {
  const std::string fn = std::string("ReadANCF") + tag + "MeshFromFile: ";
  auto fail = [&](const std::string& m) {
    if (error) *error = fn + m;
    return false;
  };
  std::ifstream file(path);
  if (!file.is_open()) return fail("failed to open " + path);
  std::vector<std::string> t;
  if (!next_record(file, t)) return fail("empty file");
  if (t.size() != 2 || t[0] != std::string("ancf") + tag + "_mesh")
    return fail(std::string("expected header 'ancf") + tag + "_mesh <version>'");
  if (!to_int(t[1], out.version) || out.version <= 0) return fail("invalid mesh version");
  if (!next_record(file, t)) return fail("missing nodes section");
  if (has_grid && t[0] == "grid") {
    if (t.size() != 11 || t[1] != "nx" || t[3] != "ny" || t[5] != "L" || t[7] != "origin") return fail("invalid grid line");
    if (!to_int(t[2], *gnx) || !to_int(t[4], *gny) || !to_double(t[6], *gL) || !to_double(t[8], gorigin[0]) ||
        !to_double(t[9], gorigin[1]) || !to_double(t[10], gorigin[2]))
      return fail("failed to parse grid values");
    *has_grid = true;
    if (!next_record(file, t)) return fail("missing nodes section");
  } else if (!has_grid && (t[0] == "tire" || t[0] == "meta")) {
    if (!next_record(file, t)) return fail("missing nodes section");
  }
  int n_nodes = 0;
  if (t.size() != 2 || t[0] != "nodes" || !to_int(t[1], n_nodes) || n_nodes <= 0) return fail("invalid nodes header");
  out.n_nodes = n_nodes;
  out.node_family.assign(static_cast<size_t>(n_nodes), "");
  out.x12.resize(4 * n_nodes);
  out.y12.resize(4 * n_nodes);
  out.z12.resize(4 * n_nodes);
  std::vector<bool> seen(static_cast<size_t>(n_nodes), false);
  for (int i = 0; i < n_nodes; ++i) {
    if (!next_record(file, t)) return fail("unexpected EOF in nodes");
    if (t.size() != 14) return fail("invalid node line (expected 14 tokens)");
    int id = -1;
    if (!to_int(t[0], id) || id < 0 || id >= n_nodes) return fail("node id out of range");
    if (seen[id]) return fail("duplicate node id");
    seen[id] = true;
    out.node_family[id] = t[1];
    double v[12];
    for (int k = 0; k < 12; ++k)
      if (!to_double(t[2 + k], v[k])) return fail("failed to parse node dofs");
    for (int k = 0; k < 4; ++k) {
      out.x12(4 * id + k) = v[k];
      out.y12(4 * id + k) = v[4 + k];
      out.z12(4 * id + k) = v[8 + k];
    }
  }
  if (!next_record(file, t)) return fail("missing elements section");
  int n_el = 0;
  if (t.size() != 2 || t[0] != "elements" || !to_int(t[1], n_el) || n_el <= 0) return fail("invalid elements header");
  out.n_elements = n_el;
  out.element_connectivity.resize(n_el, nn);
  if (eL) {
    eL->resize(n_el);
    eW->resize(n_el);
    eH->resize(n_el);
    efam->assign(static_cast<size_t>(n_el), "");
  }
  std::vector<bool> seen_e(static_cast<size_t>(n_el), false);
  const size_t ntok = eL ? 9 : 4;
  for (int i = 0; i < n_el; ++i) {
    if (!next_record(file, t)) return fail("unexpected EOF in elements");
    if (t.size() != ntok) return fail("invalid element line (expected " + std::to_string(ntok) + " tokens)");
    int id = -1;
    if (!to_int(t[0], id) || id < 0 || id >= n_el) return fail("element id out of range");
    if (seen_e[id]) return fail("duplicate element id");
    seen_e[id] = true;
    if (eL) {
      (*efam)[id] = t[1];
      double l = 0, w = 0, hh = 0;
      if (!to_double(t[2], l) || !to_double(t[3], w) || !to_double(t[4], hh)) return fail("failed to parse element L/W/H");
      (*eL)(id) = l;
      (*eW)(id) = w;
      (*eH)(id) = hh;
    }
    for (int k = 0; k < nn; ++k) {
      int n = -1;
      if (!to_int(t[ntok - nn + k], n) || n < 0 || n >= n_nodes) return fail("element node id out of range");
      out.element_connectivity(id, k) = n;
    }
  }
  if (!next_record(file, t)) {
    out.constraints = LinearConstraintCSR{};
    return true;
  }
  int n_c = 0;
  if (t.size() != 2 || t[0] != "constraints" || !to_int(t[1], n_c) || n_c < 0) return fail("invalid constraints header");
  LinearConstraintBuilder builder(12 * n_nodes);
  for (int i = 0; i < n_c; ++i) {
    if (!next_record(file, t)) return fail("unexpected EOF in constraints");
    int a = -1, b = -1;
    if (t[0] == "pinned") {
      if (t.size() != 3) return fail("pinned expects 'pinned a b'");
      if (!to_int(t[1], a) || !to_int(t[2], b) || a < 0 || b < 0 || a >= n_nodes || b >= n_nodes)
        return fail("pinned node id out of range");
      AppendANCF3243VectorEqualityConstraint(builder, a, b, 0);
    } else if (t[0] == "welded") {
      if (t.size() != 12) return fail("welded expects 'welded a b q00..q22'");
      if (!to_int(t[1], a) || !to_int(t[2], b) || a < 0 || b < 0 || a >= n_nodes || b >= n_nodes)
        return fail("welded node id out of range");
      double Q[9];
      for (int k = 0; k < 9; ++k)
        if (!to_double(t[3 + k], Q[k])) return fail("welded failed to parse Q");
      AppendANCF3243VectorEqualityConstraint(builder, a, b, 0);  // position continuity (no rotation)
      for (int slot = 1; slot <= 3; ++slot) AppendANCF3243VectorWeldedConstraint(builder, a, b, slot, Q);
    } else {
      return fail("unknown constraint type '" + t[0] + "'");
    }
  }
  out.constraints = builder.ToCSR();
  return true;
}
}  // namespace detail

// mesh_utils.cc:444-736
inline bool ReadANCF3243MeshFromFile(const std::string& path, ANCF3243Mesh& out, std::string* error = nullptr) {
  out = ANCF3243Mesh();
  return detail::read_ancf_mesh(path, "3243", 2, out, error, &out.has_grid, &out.grid_nx, &out.grid_ny, &out.grid_L,
                                out.grid_origin, nullptr, nullptr, nullptr, nullptr);
}
// mesh_utils.cc:738-1010
inline bool ReadANCF3443MeshFromFile(const std::string& path, ANCF3443Mesh& out, std::string* error = nullptr) {
  out = ANCF3443Mesh();
  return detail::read_ancf_mesh(path, "3443", 4, out, error, nullptr, nullptr, nullptr, nullptr, nullptr,
                                &out.element_L, &out.element_W, &out.element_H, &out.element_family);
}
}  // namespace ANCFCPUUtils

#include "tlfea_mesh_manager.h"
#include "tlfea_visualization.h"

enum ElementType { TYPE_3243, TYPE_3443, TYPE_T10 };  // ElementBase.h:20

class ElementBase {  // ElementBase.h:22-50 (host-side virtuals only)
 public:
  ElementType type;
  virtual ~ElementBase() {}
  virtual int get_n_beam() const = 0;
  virtual int get_n_coef() const = 0;
  virtual void CalcMassMatrix() = 0;
  virtual void CalcInternalForce() = 0;
  virtual void CalcConstraintData() = 0;
  virtual void CalcP() = 0;
  virtual void RetrieveInternalForceToCPU(tlfea::VectorXd& internal_force) = 0;
  virtual void RetrievePositionToCPU(tlfea::VectorXd& x12, tlfea::VectorXd& y12, tlfea::VectorXd& z12) = 0;
};

struct GPU_FEAT10_Data : public ElementBase {
  GPU_FEAT10_Data(int num_elements, int num_nodes) : n_elem(num_elements), n_coef(num_nodes) { type = TYPE_T10; }

  void Initialize() { TLFEA_HANDLE_ERROR(tlfea_t10_create(n_elem, n_coef, &h)); }
  void Destroy() {
    TLFEA_HANDLE_ERROR(tlfea_t10_destroy(h));
    h = nullptr;
  }
  void Setup(const tlfea::VectorXd& tet5pt_x_host, const tlfea::VectorXd& tet5pt_y_host,
             const tlfea::VectorXd& tet5pt_z_host, const tlfea::VectorXd& tet5pt_weights_host,
             const tlfea::VectorXd& h_x12, const tlfea::VectorXd& h_y12, const tlfea::VectorXd& h_z12,
             const tlfea::MatrixXi& element_connectivity) {
    TLFEA_SOFT(tlfea_t10_setup(h, tet5pt_x_host.data(), tet5pt_y_host.data(), tet5pt_z_host.data(),
                               tet5pt_weights_host.data(), h_x12.data(), h_y12.data(), h_z12.data(),
                               element_connectivity.data()));  // column-major E x 10, like Eigen::MatrixXi
  }
  void SetDensity(double rho0) { TLFEA_SOFT(tlfea_t10_set_density(h, rho0)); }
  void SetDamping(double eta_damp, double lambda_damp) { TLFEA_SOFT(tlfea_t10_set_damping(h, eta_damp, lambda_damp)); }
  void SetSVK() { TLFEA_SOFT(tlfea_t10_set_svk_select(h)); }
  void SetSVK(double E, double nu) { TLFEA_SOFT(tlfea_t10_set_svk(h, E, nu)); }
  void SetMooneyRivlin(double mu10, double mu01, double kappa) {
    TLFEA_SOFT(tlfea_t10_set_mooney_rivlin(h, mu10, mu01, kappa));
  }
  void SetExternalForce(const tlfea::VectorXd& h_f_ext) {
    TLFEA_SOFT(tlfea_t10_set_external_force(h, h_f_ext.data(), h_f_ext.size()));
  }
  void SetNodalFixed(const tlfea::VectorXi& fixed_nodes) {
    TLFEA_SOFT(tlfea_t10_set_nodal_fixed(h, fixed_nodes.data(), fixed_nodes.size()));
    n_constraint = tlfea_t10_get_n_constraint(h);
  }
  void UpdateNodalFixed(const tlfea::VectorXi& fixed_nodes) {
    TLFEA_SOFT(tlfea_t10_update_nodal_fixed(h, fixed_nodes.data(), fixed_nodes.size()));
    n_constraint = tlfea_t10_get_n_constraint(h);
  }
  void UpdatePositions(const tlfea::VectorXd& x, const tlfea::VectorXd& y, const tlfea::VectorXd& z) {
    TLFEA_SOFT(tlfea_t10_update_positions(h, x.data(), y.data(), z.data(), x.size()));
  }
  void UpdateConstraintTargets(const tlfea::VectorXd& x, const tlfea::VectorXd& y, const tlfea::VectorXd& z) {
    TLFEA_SOFT(tlfea_t10_update_constraint_targets(h, x.data(), y.data(), z.data(), x.size()));
  }

  void CalcDnDuPre() { TLFEA_HANDLE_ERROR(tlfea_t10_calc_dndu_pre(h)); }
  void CalcMassMatrix() override { TLFEA_HANDLE_ERROR(tlfea_t10_calc_mass_matrix(h)); }
  void BuildMassCSRPattern() { TLFEA_HANDLE_ERROR(tlfea_t10_build_mass_csr_pattern(h)); }
  void ConvertToCSR_ConstraintJacT() { TLFEA_HANDLE_ERROR(tlfea_t10_convert_to_csr_constraint_jact(h)); }
  void BuildConstraintJacobianTransposeCSR() { ConvertToCSR_ConstraintJacT(); }
  void ConvertToCSR_ConstraintJac() { TLFEA_HANDLE_ERROR(tlfea_t10_convert_to_csr_constraint_jac(h)); }
  void BuildConstraintJacobianCSR() { ConvertToCSR_ConstraintJac(); }
  void CalcInternalForce() override { TLFEA_HANDLE_ERROR(tlfea_t10_calc_internal_force(h)); }
  void CalcConstraintData() override { TLFEA_SOFT(tlfea_t10_calc_constraint_data(h)); }
  void CalcP() override { TLFEA_HANDLE_ERROR(tlfea_t10_calc_p(h)); }

  void RetrieveMassCSRToCPU(std::vector<int>& offsets, std::vector<int>& columns, std::vector<double>& values) {
    int nnz = 0;
    TLFEA_HANDLE_ERROR(tlfea_t10_mass_csr_nnz(h, &nnz));
    offsets.assign(static_cast<size_t>(n_coef) + 1, 0);
    columns.assign(static_cast<size_t>(nnz), 0);
    values.assign(static_cast<size_t>(nnz), 0.0);
    TLFEA_HANDLE_ERROR(tlfea_t10_retrieve_mass_csr(h, offsets.data(), columns.data(), values.data()));
  }
  void RetrieveInternalForceToCPU(tlfea::VectorXd& f) override {
    f.resize(3 * n_coef);
    TLFEA_HANDLE_ERROR(tlfea_t10_retrieve_internal_force(h, f.data()));
  }
  void RetrieveExternalForceToCPU(tlfea::VectorXd& f) {
    f.resize(3 * n_coef);
    TLFEA_HANDLE_ERROR(tlfea_t10_retrieve_external_force(h, f.data()));
  }
  void RetrievePositionToCPU(tlfea::VectorXd& x, tlfea::VectorXd& y, tlfea::VectorXd& z) override {
    x.resize(n_coef);
    y.resize(n_coef);
    z.resize(n_coef);
    TLFEA_HANDLE_ERROR(tlfea_t10_retrieve_position(h, x.data(), y.data(), z.data()));
  }
  void RetrievePFromFToCPU(std::vector<std::vector<tlfea::MatrixXd>>& P) { retrieve33(P, true); }
  void RetrieveDeformationGradientToCPU(std::vector<std::vector<tlfea::MatrixXd>>& F) { retrieve33(F, false); }
  // sizes follow the element kind behind the handle (tlfea_elem_dims): T10 S = 10, Q = 5; 3243 8, 12; 3443 16, 48
  void RetrieveDnDuPreToCPU(std::vector<std::vector<tlfea::MatrixXd>>& g) {
    int S = 0, Q = 0;
    TLFEA_HANDLE_ERROR(tlfea_elem_dims(h, &S, &Q));
    std::vector<double> flat(static_cast<size_t>(n_elem) * Q * 3 * S);
    TLFEA_HANDLE_ERROR(tlfea_t10_retrieve_dndu_pre(h, flat.data()));
    g.assign(n_elem, std::vector<tlfea::MatrixXd>(Q));
    for (int e = 0; e < n_elem; e++)
      for (int q = 0; q < Q; q++) {
        g[e][q].resize(S, 3);
        std::copy_n(flat.data() + (static_cast<size_t>(e) * Q + q) * 3 * S, 3 * S, g[e][q].data());
      }
  }
  void RetrieveDetJToCPU(std::vector<std::vector<double>>& detJ) {
    int S = 0, Q = 0;
    TLFEA_HANDLE_ERROR(tlfea_elem_dims(h, &S, &Q));
    std::vector<double> flat(static_cast<size_t>(n_elem) * Q);
    TLFEA_HANDLE_ERROR(tlfea_t10_retrieve_detj(h, flat.data()));
    detJ.assign(n_elem, std::vector<double>(Q));
    for (int e = 0; e < n_elem; e++) std::copy_n(flat.data() + static_cast<size_t>(Q) * e, Q, detJ[e].data());
  }
  void RetrieveConnectivityToCPU(tlfea::MatrixXi& connectivity) {
    connectivity.resize(n_elem, 10);
    TLFEA_HANDLE_ERROR(tlfea_t10_retrieve_connectivity(h, connectivity.data()));
  }
  void WriteOutputVTK(const std::string& filename) { TLFEA_HANDLE_ERROR(tlfea_t10_write_output_vtk(h, filename.c_str())); }

  const double* GetX12DevicePtr() const { return tlfea_t10_x12_device_ptr(h); }
  const double* GetY12DevicePtr() const { return tlfea_t10_y12_device_ptr(h); }
  const double* GetZ12DevicePtr() const { return tlfea_t10_z12_device_ptr(h); }
  double* GetExternalForceDevicePtr() { return tlfea_t10_external_force_device_ptr(h); }
  double* Get_Constraint_Ptr() { return tlfea_t10_constraint_device_ptr(h); }
  bool Get_Is_Constraint_Setup() { return tlfea_t10_is_constraint_setup(h) != 0; }
  int get_n_elem() const { return n_elem; }
  int get_n_coef() const override { return n_coef; }
  int get_n_constraint() const { return n_constraint; }
  int get_n_beam() const override { return n_elem; }

  tlfea_t10_t h = nullptr;  // stands where the reference keeps its device mirror `d_data`
  int n_elem;
  int n_coef;
  int n_constraint = 0;  // the reference leaves this uninitialised until SetNodalFixed (FEAT10Data.cuh:793)

 private:
  void retrieve33(std::vector<std::vector<tlfea::MatrixXd>>& out, bool P) {
    int S = 0, Q = 0;
    TLFEA_HANDLE_ERROR(tlfea_elem_dims(h, &S, &Q));
    std::vector<double> flat(static_cast<size_t>(n_elem) * Q * 9);
    TLFEA_HANDLE_ERROR(P ? tlfea_t10_retrieve_p_from_f(h, flat.data())
                         : tlfea_t10_retrieve_deformation_gradient(h, flat.data()));
    out.assign(n_elem, std::vector<tlfea::MatrixXd>(Q));
    for (int e = 0; e < n_elem; e++)
      for (int q = 0; q < Q; q++) {
        out[e][q].resize(3, 3);
        std::copy_n(flat.data() + (static_cast<size_t>(e) * Q + q) * 9, 9, out[e][q].data());
      }
  }
};

// GPU_ANCF3243_Data / GPU_ANCF3443_Data (ANCF3243Data.cuh:33-1152, ANCF3443Data.cuh): share the handle type and
// every non-T10-specific entry point with GPU_FEAT10_Data; n_coef = 4 * n_nodes, SetNodalFixed takes
// coefficient indices.
struct GPU_ANCF_DataBase : public GPU_FEAT10_Data {
  GPU_ANCF_DataBase(int kind, ElementType t, int n_nodes, int n_elements)
      : GPU_FEAT10_Data(n_elements, 4 * n_nodes), kind_(kind), n_nodes_(n_nodes) {
    type = t;
    n_beam = n_elements;
  }
  void Initialize() { TLFEA_HANDLE_ERROR(tlfea_ancf_create(kind_, n_nodes_, n_elem, &h)); }
  void CalcDsDuPre() { TLFEA_HANDLE_ERROR(tlfea_ancf_calc_dsdu_pre(h)); }
  // ANCF3243Data.cuh:810-940: same argument checks and messages, std::cerr + return on misuse
  enum ConstraintMode { kConstraintNone = 0, kConstraintFixedCoefficients = 1, kConstraintLinearCSR = 2 };
  void SetLinearConstraintsCSR(const std::vector<int>& j_offsets, const std::vector<int>& j_columns,
                               const std::vector<double>& j_values, const tlfea::VectorXd& rhs) {
    if (j_offsets.empty() || j_offsets.front() != 0) {
      std::cerr << "SetLinearConstraintsCSR: invalid offsets." << std::endl;
      return;
    }
    if (rhs.size() + 1 != static_cast<int>(j_offsets.size())) {
      std::cerr << "SetLinearConstraintsCSR: offsets/rhs size mismatch." << std::endl;
      return;
    }
    if (j_columns.size() != j_values.size()) {
      std::cerr << "SetLinearConstraintsCSR: columns/values size mismatch." << std::endl;
      return;
    }
    if (j_offsets.back() != static_cast<int>(j_columns.size())) {
      std::cerr << "SetLinearConstraintsCSR: offsets.back != nnz." << std::endl;
      return;
    }
    TLFEA_SOFT(tlfea_t10_set_linear_constraints_csr(h, rhs.size(), j_offsets.data(), j_columns.data(), j_values.data(),
                                                    rhs.data()));
    n_constraint = tlfea_t10_get_n_constraint(h);
  }
  // same misuse messages as ANCF3443Data.cuh:979-993, printed by the C-ABI; std::cerr + return
  void UpdateLinearConstraintRHS(const tlfea::VectorXd& rhs) {
    if (tlfea_t10_update_linear_constraint_rhs(h, rhs.data(), rhs.size()) != 0) std::cerr << tlfea_last_error() << std::endl;
  }
  int GetConstraintMode() const { return tlfea_t10_get_constraint_mode(h); }
  // element connectivity in NODE ids, n_beam x (2 | 4) (ANCF3243Data.cu:630-642, ANCF3443Data.cu:597-603); the handle
  // keeps coefficient ids [S][E] (slot 0 of node n of the element = 4 * node)
  void RetrieveConnectivityToCPU(tlfea::MatrixXi& connectivity) {
    int S = 0, Q = 0;
    TLFEA_HANDLE_ERROR(tlfea_elem_dims(h, &S, &Q));
    std::vector<int> coef(static_cast<size_t>(S) * n_elem);
    TLFEA_HANDLE_ERROR(tlfea_t10_retrieve_connectivity(h, coef.data()));
    connectivity.resize(n_elem, S / 4);
    for (int e = 0; e < n_elem; e++)
      for (int n = 0; n < S / 4; n++) connectivity(e, n) = coef[static_cast<size_t>(4 * n) * n_elem + e] / 4;
  }
  // PrintDsDuPre (ANCF3243Data.cu:326-360, ANCF3443Data.cu same member): same text layout
  void PrintDsDuPre() {
    std::vector<std::vector<tlfea::MatrixXd>> g;
    std::vector<std::vector<double>> dj;
    RetrieveDnDuPreToCPU(g);
    RetrieveDetJToCPU(dj);
    char buf[64];
    for (int e = 0; e < n_elem; e++)
      for (size_t q = 0; q < g[e].size(); q++) {
        std::cout << "\n=== Elem " << e << " Quadrature Point " << q << " detJ_ref=" << dj[e][q] << " ===" << std::endl;
        std::cout << "        dN/dx       dN/dy       dN/dz" << std::endl;
        for (int i = 0; i < g[e][q].rows(); i++) {
          std::cout << "Shape " << i << ": ";
          for (int j = 0; j < 3; j++) {
            std::snprintf(buf, sizeof buf, "%10.6f ", g[e][q](i, j));
            std::cout << buf;
          }
          std::cout << std::endl;
        }
      }
  }
  void RetrieveConstraintJacobianCSRToCPU(std::vector<int>& offsets, std::vector<int>& columns,
                                          std::vector<double>& values) {
    const int nnz = tlfea_t10_constraint_jac_nnz(h);
    offsets.assign(static_cast<size_t>(n_constraint) + 1, 0);
    columns.assign(static_cast<size_t>(nnz), 0);
    values.assign(static_cast<size_t>(nnz), 0.0);
    TLFEA_SOFT(tlfea_t10_retrieve_constraint_jac_csr(h, offsets.data(), columns.data(), values.data()));
  }
  void RetrieveConstraintDataToCPU(tlfea::VectorXd& c) {
    c.resize(n_constraint);
    TLFEA_SOFT(tlfea_t10_retrieve_constraint_data(h, c.data()));
  }
  // dense n_constraint x 3 n_coef Jacobian (ANCF3243Data.cu:758-778: the selector rows of the pinned coefficients; the
  // CSR rows when general linear constraints are set); all zeros before the constraints are set up
  void RetrieveConstraintJacobianToCPU(tlfea::MatrixXd& constraint_jac) {
    constraint_jac.resize(n_constraint, 3 * n_coef);
    if (!tlfea_t10_is_constraint_setup(h) || n_constraint == 0) return;
    std::vector<int> off, col;
    std::vector<double> val;
    RetrieveConstraintJacobianCSRToCPU(off, col, val);
    for (int r = 0; r < n_constraint; r++)
      for (int k = off[r]; k < off[r + 1]; k++) constraint_jac(r, col[k]) = val[k];
  }
  int n_beam;

 protected:
  void setup_impl(const tlfea::VectorXd& L, const tlfea::VectorXd& W, const tlfea::VectorXd& H,
                  const tlfea::VectorXd* mr[6], const tlfea::VectorXd* fr[6], const tlfea::VectorXd& x,
                  const tlfea::VectorXd& y, const tlfea::VectorXd& z, const int* conn, int colmajor) {
    const int nqm[3] = {mr[0]->size(), mr[1]->size(), mr[2]->size()};
    const int nq[3] = {fr[0]->size(), fr[1]->size(), fr[2]->size()};
    TLFEA_SOFT(tlfea_ancf_setup(h, L.data(), W.data(), H.data(), mr[0]->data(), mr[1]->data(), mr[2]->data(),
                                mr[3]->data(), mr[4]->data(), mr[5]->data(), nqm, fr[0]->data(), fr[1]->data(),
                                fr[2]->data(), fr[3]->data(), fr[4]->data(), fr[5]->data(), nq, x.data(), y.data(),
                                z.data(), conn, colmajor));
  }
  static tlfea::VectorXd fill(int n, double v) {
    tlfea::VectorXd r(n);
    for (int i = 0; i < n; i++) r(i) = v;
    return r;
  }
  int kind_, n_nodes_;
};

struct GPU_ANCF3243_Data : public GPU_ANCF_DataBase {
  GPU_ANCF3243_Data(int n_nodes, int n_elements) : GPU_ANCF_DataBase(3243, TYPE_3243, n_nodes, n_elements) {}
  // ANCF3243Data.cuh:511-521 (per-element dimensions) and :653-670 (scalar overload)
  void Setup(const tlfea::VectorXd& length, const tlfea::VectorXd& width, const tlfea::VectorXd& height,
             const tlfea::VectorXd& gauss_xi_m, const tlfea::VectorXd& gauss_xi, const tlfea::VectorXd& gauss_eta,
             const tlfea::VectorXd& gauss_zeta, const tlfea::VectorXd& weight_xi_m, const tlfea::VectorXd& weight_xi,
             const tlfea::VectorXd& weight_eta, const tlfea::VectorXd& weight_zeta, const tlfea::VectorXd& h_x12,
             const tlfea::VectorXd& h_y12, const tlfea::VectorXd& h_z12, const tlfea::MatrixXi& conn) {
    const tlfea::VectorXd* mr[6] = {&gauss_xi_m, &gauss_eta, &gauss_zeta, &weight_xi_m, &weight_eta, &weight_zeta};
    const tlfea::VectorXd* fr[6] = {&gauss_xi, &gauss_eta, &gauss_zeta, &weight_xi, &weight_eta, &weight_zeta};
    setup_impl(length, width, height, mr, fr, h_x12, h_y12, h_z12, conn.data(), /*column-major E x 2*/ 1);
  }
  void Setup(double L, double W, double H, const tlfea::VectorXd& gauss_xi_m, const tlfea::VectorXd& gauss_xi,
             const tlfea::VectorXd& gauss_eta, const tlfea::VectorXd& gauss_zeta, const tlfea::VectorXd& weight_xi_m,
             const tlfea::VectorXd& weight_xi, const tlfea::VectorXd& weight_eta, const tlfea::VectorXd& weight_zeta,
             const tlfea::VectorXd& h_x12, const tlfea::VectorXd& h_y12, const tlfea::VectorXd& h_z12,
             const tlfea::MatrixXi& conn) {
    Setup(fill(n_elem, L), fill(n_elem, W), fill(n_elem, H), gauss_xi_m, gauss_xi, gauss_eta, gauss_zeta, weight_xi_m,
          weight_xi, weight_eta, weight_zeta, h_x12, h_y12, h_z12, conn);
  }
};

struct GPU_ANCF3443_Data : public GPU_ANCF_DataBase {
  GPU_ANCF3443_Data(int n_nodes, int n_elements) : GPU_ANCF_DataBase(3443, TYPE_3443, n_nodes, n_elements) {}
  // strip constructor: every new element of the chain brings 2 new nodes (ANCF3443Data.cuh:445-449)
  explicit GPU_ANCF3443_Data(int num_beams) : GPU_ANCF_DataBase(3443, TYPE_3443, 4 + 2 * (num_beams - 1), num_beams) {}
  // ANCF3443Data.cuh:532-542
  void Setup(const tlfea::VectorXd& length, const tlfea::VectorXd& width, const tlfea::VectorXd& height,
             const tlfea::VectorXd& gauss_xi_m, const tlfea::VectorXd& gauss_eta_m, const tlfea::VectorXd& gauss_zeta_m,
             const tlfea::VectorXd& gauss_xi, const tlfea::VectorXd& gauss_eta, const tlfea::VectorXd& gauss_zeta,
             const tlfea::VectorXd& weight_xi_m, const tlfea::VectorXd& weight_eta_m,
             const tlfea::VectorXd& weight_zeta_m, const tlfea::VectorXd& weight_xi, const tlfea::VectorXd& weight_eta,
             const tlfea::VectorXd& weight_zeta, const tlfea::VectorXd& h_x12, const tlfea::VectorXd& h_y12,
             const tlfea::VectorXd& h_z12, const tlfea::MatrixXi& conn) {
    const tlfea::VectorXd* mr[6] = {&gauss_xi_m, &gauss_eta_m, &gauss_zeta_m, &weight_xi_m, &weight_eta_m, &weight_zeta_m};
    const tlfea::VectorXd* fr[6] = {&gauss_xi, &gauss_eta, &gauss_zeta, &weight_xi, &weight_eta, &weight_zeta};
    setup_impl(length, width, height, mr, fr, h_x12, h_y12, h_z12, conn.data(), 1);
  }
  void Setup(double L, double W, double H, const tlfea::VectorXd& gxm, const tlfea::VectorXd& gym,
             const tlfea::VectorXd& gzm, const tlfea::VectorXd& gx, const tlfea::VectorXd& gy, const tlfea::VectorXd& gz,
             const tlfea::VectorXd& wxm, const tlfea::VectorXd& wym, const tlfea::VectorXd& wzm,
             const tlfea::VectorXd& wx, const tlfea::VectorXd& wy, const tlfea::VectorXd& wz, const tlfea::VectorXd& x,
             const tlfea::VectorXd& y, const tlfea::VectorXd& z, const tlfea::MatrixXi& conn) {
    Setup(fill(n_elem, L), fill(n_elem, W), fill(n_elem, H), gxm, gym, gzm, gx, gy, gz, wxm, wym, wzm, wx, wy, wz, x, y,
          z, conn);
  }
};

class SolverBase {  // SolverBase.h:16-23
 public:
  virtual ~SolverBase() = default;
  virtual void Solve() = 0;
  virtual void SetParameters(void* params) = 0;
};

struct SyncedNewtonParams {  // SyncedNewton.cuh:29-33
  double inner_atol, inner_rtol, outer_tol, rho;
  int max_outer, max_inner;
  double time_step;
};

class SyncedNewtonSolver : public SolverBase {
 public:
  SyncedNewtonSolver(ElementBase* data, int n_constraints) {
    // SyncedNewton.cuh:52-85: the three element types share one device path here
    TLFEA_HANDLE_ERROR(tlfea_newton_create(static_cast<GPU_FEAT10_Data*>(data)->h, n_constraints, &s_));
  }
  ~SyncedNewtonSolver() override { tlfea_newton_destroy(s_); }
  void Setup() { TLFEA_HANDLE_ERROR(tlfea_newton_setup(s_)); }
  void SetParameters(void* params) override {
    const SyncedNewtonParams* p = static_cast<SyncedNewtonParams*>(params);
    tlfea_newton_params c{p->inner_atol, p->inner_rtol, p->outer_tol, p->rho, p->max_outer, p->max_inner, p->time_step};
    TLFEA_HANDLE_ERROR(tlfea_newton_set_parameters(s_, &c));
  }
  void AnalyzeHessianSparsity() { TLFEA_HANDLE_ERROR(tlfea_newton_analyze_hessian_sparsity(s_)); }
  void SetFixedSparsityPattern(bool fixed) { TLFEA_HANDLE_ERROR(tlfea_newton_set_fixed_sparsity_pattern(s_, fixed)); }
  void OneStepNewtonCuDSS() { TLFEA_HANDLE_ERROR(tlfea_newton_solve(s_)); }  // name kept for drop-in; solves with PCG
  void Solve() override { OneStepNewtonCuDSS(); }
  double* GetVelocityGuessDevicePtr() const { return tlfea_newton_velocity_guess_device_ptr(s_); }
  double compute_l2_norm_cublas(double* d_vec, int n_dofs) {
    double out = 0.0;
    TLFEA_HANDLE_ERROR(tlfea_newton_l2_norm(s_, d_vec, n_dofs, &out));
    return out;
  }
  void SetVerbose(int v) { tlfea_newton_set_verbose(s_, v); }
  tlfea_newton_t handle() { return s_; }

 private:
  tlfea_newton_t s_ = nullptr;
};

// SyncedNesterovSolver (SyncedNesterov.cuh:26-260)
struct SyncedNesterovParams {
  double alpha, rho, inner_tol, outer_tol;
  int max_outer, max_inner;
  double time_step;
};
class SyncedNesterovSolver : public SolverBase {
 public:
  SyncedNesterovSolver(ElementBase* data, int n_constraints) {
    TLFEA_HANDLE_ERROR(tlfea_nesterov_create(static_cast<GPU_FEAT10_Data*>(data)->h, n_constraints, &a_));
  }
  ~SyncedNesterovSolver() override { tlfea_nesterov_destroy(a_); }
  void Setup() { TLFEA_HANDLE_ERROR(tlfea_nesterov_setup(a_)); }
  void SetParameters(void* params) override {
    const SyncedNesterovParams* p = static_cast<SyncedNesterovParams*>(params);
    tlfea_nesterov_params c{p->alpha, p->rho, p->inner_tol, p->outer_tol, p->max_outer, p->max_inner, p->time_step};
    TLFEA_HANDLE_ERROR(tlfea_nesterov_set_parameters(a_, &c));
  }
  void OneStepNesterov() { TLFEA_HANDLE_ERROR(tlfea_nesterov_solve(a_)); }
  void Solve() override { OneStepNesterov(); }
  double* GetVelocityGuessDevicePtr() const { return tlfea_nesterov_velocity_guess_device_ptr(a_); }
  void SetVerbose(int v) { tlfea_nesterov_set_verbose(a_, v); }

 private:
  tlfea_nesterov_t a_ = nullptr;
};

// SyncedVBDSolver (SyncedVBD.cuh:13-330): vertex block descent
struct SyncedVBDParams {
  double inner_tol, inner_rtol, outer_tol, rho;
  int max_outer, max_inner;
  double time_step;
  double omega;
  double hess_eps;
  int convergence_check_interval;
  int color_group_size;
};
class SyncedVBDSolver : public SolverBase {
 public:
  SyncedVBDSolver(ElementBase* data, int n_constraints) {
    TLFEA_HANDLE_ERROR(tlfea_vbd_create(static_cast<GPU_FEAT10_Data*>(data)->h, n_constraints, &a_));
  }
  ~SyncedVBDSolver() override { tlfea_vbd_destroy(a_); }
  void Setup() { TLFEA_HANDLE_ERROR(tlfea_vbd_setup(a_)); }
  void SetParameters(void* params) override {
    const SyncedVBDParams* p = static_cast<SyncedVBDParams*>(params);
    tlfea_vbd_params c{p->inner_tol, p->inner_rtol, p->outer_tol, p->rho, p->max_outer, p->max_inner, p->time_step,
                       p->omega, p->hess_eps, p->convergence_check_interval, p->color_group_size};
    TLFEA_HANDLE_ERROR(tlfea_vbd_set_parameters(a_, &c));
  }
  void InitializeColoring() { TLFEA_HANDLE_ERROR(tlfea_vbd_initialize_coloring(a_)); }
  void InitializeMassDiagBlocks() { TLFEA_HANDLE_ERROR(tlfea_vbd_initialize_mass_diag_blocks(a_)); }
  void InitializeFixedMap() { TLFEA_HANDLE_ERROR(tlfea_vbd_initialize_fixed_map(a_)); }
  void OneStepVBD() { TLFEA_HANDLE_ERROR(tlfea_vbd_solve(a_)); }
  void Solve() override { OneStepVBD(); }
  int GetNumColors() const {
    int nc = 0;
    TLFEA_SOFT(tlfea_vbd_coloring_sizes(a_, &nc, nullptr));
    return nc;
  }
  double* GetVelocityGuessDevicePtr() const { return tlfea_vbd_velocity_guess_device_ptr(a_); }
  void SetVerbose(int v) { tlfea_vbd_set_verbose(a_, v); }

 private:
  tlfea_vbd_t a_ = nullptr;
};

// SyncedAdamWNocoopSolver (SyncedAdamWNocoop.cuh:20-198); SyncedAdamWParams field order of SyncedAdamW.cuh:27-34
struct SyncedAdamWParams {
  double lr, beta1, beta2, eps, weight_decay, lr_decay;
  double inner_tol, outer_tol, rho;
  int max_outer, max_inner;
  double time_step;
  int convergence_check_interval;
  double inner_rtol;
};
using SyncedAdamWNocoopParams = SyncedAdamWParams;

class SyncedAdamWNocoopSolver : public SolverBase {
 public:
  SyncedAdamWNocoopSolver(ElementBase* data, int n_constraints) {
    TLFEA_HANDLE_ERROR(tlfea_adamw_create(static_cast<GPU_FEAT10_Data*>(data)->h, n_constraints, &a_));
  }
  ~SyncedAdamWNocoopSolver() override { tlfea_adamw_destroy(a_); }
  void Setup() { TLFEA_HANDLE_ERROR(tlfea_adamw_setup(a_)); }
  void SetParameters(void* params) override {
    const SyncedAdamWNocoopParams* p = static_cast<SyncedAdamWNocoopParams*>(params);
    tlfea_adamw_params c{p->lr, p->beta1, p->beta2, p->eps, p->weight_decay, p->lr_decay, p->inner_tol, p->outer_tol,
                         p->rho, p->max_outer, p->max_inner, p->time_step, p->convergence_check_interval, p->inner_rtol};
    TLFEA_HANDLE_ERROR(tlfea_adamw_set_parameters(a_, &c));
  }
  void OneStepAdamWNocoop() { TLFEA_HANDLE_ERROR(tlfea_adamw_solve(a_)); }
  void Solve() override { OneStepAdamWNocoop(); }
  double* GetVelocityGuessDevicePtr() const { return tlfea_adamw_velocity_guess_device_ptr(a_); }
  void SetVerbose(int v) { tlfea_adamw_set_verbose(a_, v); }

 protected:
  tlfea_adamw_t a_ = nullptr;
};

// SyncedAdamWSolver (SyncedAdamW.cuh, SyncedAdamW.cu:96-445): the cooperative-kernel sibling -- the same step as ordinary
// launches, with that file's flag / multiplier semantics (tlfea_adamw_set_cooperative_semantics)
class SyncedAdamWSolver : public SyncedAdamWNocoopSolver {
 public:
  SyncedAdamWSolver(ElementBase* data, int n_constraints) : SyncedAdamWNocoopSolver(data, n_constraints) {
    TLFEA_HANDLE_ERROR(tlfea_adamw_set_cooperative_semantics(a_, 1));
  }
  void OneStepAdamW() { TLFEA_HANDLE_ERROR(tlfea_adamw_solve(a_)); }
};
