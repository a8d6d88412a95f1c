// tlfea_mesh_manager.h -- ANCFCPUUtils::MeshManager of the reference (lib_utils/mesh_manager.h:67-235,
// mesh_manager.cc:180-220, 443-570): several TetGen T10 meshes behind one node / element numbering, the entry point
// the multi-body drivers use before GPU_FEAT10_Data::Setup.  Same member names and return conventions.  Kept as ONE
// unified node / element array plus a table of instances: a transform rewrites the instance's slice in place (the
// reference keeps per-mesh copies and rebuilds the union after every call).  The NPZ pressure-field loader belongs to
// the collision subsystem (out of scope): LoadScalarFieldFromNpz reports failure.
// Included by tlfea_facade.h.
#pragma once
#include <cmath>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace ANCFCPUUtils {

using Matrix4d = tlfea::MatrixXd;  // 4 x 4 homogeneous transform, column-major like Eigen::Matrix4d

inline Matrix4d identity4() {
  Matrix4d T(4, 4);
  for (int i = 0; i < 4; i++) T(i, i) = 1.0;
  return T;
}
inline Matrix4d rotationX(double a) {  // mesh_manager.h:10-18
  Matrix4d R = identity4();
  R(1, 1) = std::cos(a); R(1, 2) = -std::sin(a);
  R(2, 1) = std::sin(a); R(2, 2) = std::cos(a);
  return R;
}
inline Matrix4d rotationY(double a) {  // :20-29
  Matrix4d R = identity4();
  R(0, 0) = std::cos(a); R(0, 2) = std::sin(a);
  R(2, 0) = -std::sin(a); R(2, 2) = std::cos(a);
  return R;
}
inline Matrix4d rotationZ(double a) {
  Matrix4d R = identity4();
  R(0, 0) = std::cos(a); R(0, 1) = -std::sin(a);
  R(1, 0) = std::sin(a); R(1, 1) = std::cos(a);
  return R;
}
inline Matrix4d translation(double dx, double dy, double dz) {  // :31-37
  Matrix4d T = identity4();
  T(0, 3) = dx; T(1, 3) = dy; T(2, 3) = dz;
  return T;
}
inline Matrix4d uniformScale(double s) {  // :39-45
  Matrix4d S = identity4();
  S(0, 0) = S(1, 1) = S(2, 2) = s;
  return S;
}

struct MeshInstance {  // mesh_manager.h:50-56
  int node_offset, element_offset, num_nodes, num_elements;
  std::string name;
};

class MeshManager {
 public:
  MeshManager() { Clear(); }

  // -> instance id, -1 when either file cannot be read (mesh_manager.cc:180-220)
  int LoadMesh(const std::string& node_file, const std::string& elem_file, const std::string& name = "") {
    tlfea::MatrixXd nodes;
    tlfea::MatrixXi elems;
    const int nn = FEAT10_read_nodes(node_file, nodes);
    const int ne = FEAT10_read_elements(elem_file, elems);
    if (nn <= 0 || ne <= 0) {
      std::cerr << "MeshManager: Failed to load mesh from " << node_file << " and " << elem_file << std::endl;
      return -1;
    }
    const int id = GetNumMeshes();
    const int n0 = GetTotalNodes(), e0 = GetTotalElements();
    inst_.push_back({n0, e0, nn, ne, name.empty() ? "mesh_" + std::to_string(id) : name});
    tlfea::MatrixXd all(n0 + nn, 3);
    for (int c = 0; c < 3; c++) {
      for (int i = 0; i < n0; i++) all(i, c) = nodes_(i, c);
      for (int i = 0; i < nn; i++) all(n0 + i, c) = nodes(i, c);
    }
    const int cols = e0 ? elems_.cols() : elems.cols();
    tlfea::MatrixXi alle(e0 + ne, cols);
    for (int c = 0; c < cols; c++) {
      for (int e = 0; e < e0; e++) alle(e, c) = elems_(e, c);
      for (int e = 0; e < ne; e++) alle(e0 + e, c) = elems(e, c) + n0;  // shift into the unified numbering
    }
    nodes_ = all;
    elems_ = alle;
    if (has_fields_) {  // a mesh loaded after fields were set contributes zeros until its own field arrives
      tlfea::VectorXd f(n0 + nn);
      for (int i = 0; i < n0; i++) f(i) = fields_(i);
      fields_ = f;
    }
    field_set_.push_back(false);
    return id;
  }

  void TransformMesh(int mesh_id, const Matrix4d& T) {  // mesh_manager.cc:467-482
    if (mesh_id < 0 || mesh_id >= GetNumMeshes()) {
      std::cerr << "MeshManager: Invalid mesh_id " << mesh_id << std::endl;
      return;
    }
    const MeshInstance& m = inst_[mesh_id];
    for (int i = m.node_offset; i < m.node_offset + m.num_nodes; i++) {
      const double p[3] = {nodes_(i, 0), nodes_(i, 1), nodes_(i, 2)};
      for (int r = 0; r < 3; r++) nodes_(i, r) = T(r, 0) * p[0] + T(r, 1) * p[1] + T(r, 2) * p[2] + T(r, 3);
    }
  }
  void TranslateMesh(int mesh_id, double dx, double dy, double dz) { TransformMesh(mesh_id, translation(dx, dy, dz)); }

  const tlfea::MatrixXd& GetAllNodes() const { return nodes_; }
  const tlfea::MatrixXi& GetAllElements() const { return elems_; }
  const MeshInstance& GetMeshInstance(int mesh_id) const {  // mesh_manager.cc:527-533
    if (mesh_id < 0 || mesh_id >= GetNumMeshes())
      throw std::out_of_range("MeshManager: Invalid mesh_id " + std::to_string(mesh_id));
    return inst_[mesh_id];
  }
  int GetNumMeshes() const { return static_cast<int>(inst_.size()); }
  int GetTotalNodes() const { return inst_.empty() ? 0 : inst_.back().node_offset + inst_.back().num_nodes; }
  int GetTotalElements() const { return inst_.empty() ? 0 : inst_.back().element_offset + inst_.back().num_elements; }

  bool LoadScalarFieldFromNpz(int, const std::string& npz_file, const std::string& = "p_vertex") {
    std::cerr << "MeshManager: NPZ scalar fields (" << npz_file << ") belong to the collision subsystem, not built"
              << std::endl;
    return false;
  }
  bool LoadScalarFieldFromBinary(int mesh_id, const std::string& bin_file, int n_values) {  // raw float64 array
    std::ifstream f(bin_file, std::ios::binary);
    if (!f) {
      std::cerr << "MeshManager: Failed to open binary file " << bin_file << std::endl;
      return false;
    }
    tlfea::VectorXd v(n_values);
    f.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(n_values) * sizeof(double));
    if (f.gcount() != static_cast<std::streamsize>(n_values) * static_cast<std::streamsize>(sizeof(double))) {
      std::cerr << "MeshManager: Failed to read " << n_values << " values from " << bin_file << std::endl;
      return false;
    }
    return SetScalarField(mesh_id, v);
  }
  bool SetScalarField(int mesh_id, const tlfea::VectorXd& field) {  // mesh_manager.cc:422-441
    if (mesh_id < 0 || mesh_id >= GetNumMeshes()) {
      std::cerr << "MeshManager: Invalid mesh_id " << mesh_id << std::endl;
      return false;
    }
    const MeshInstance& m = inst_[mesh_id];
    if (field.size() != m.num_nodes) {
      std::cerr << "MeshManager: Scalar field size (" << field.size() << ") does not match mesh node count ("
                << m.num_nodes << ")" << std::endl;
      return false;
    }
    if (!has_fields_) {
      fields_.resize(GetTotalNodes());
      has_fields_ = true;
    }
    for (int i = 0; i < m.num_nodes; i++) fields_(m.node_offset + i) = field(i);
    field_set_[mesh_id] = true;
    return true;
  }
  const tlfea::VectorXd& GetAllScalarFields() const { return fields_; }
  bool HasScalarFields() const { return has_fields_; }

  int GetMeshIdFromElement(int global_elem_idx) const {
    for (int k = 0; k < GetNumMeshes(); k++)
      if (global_elem_idx >= inst_[k].element_offset && global_elem_idx < inst_[k].element_offset + inst_[k].num_elements)
        return k;
    return -1;
  }
  int GetMeshIdFromNode(int global_node_idx) const {
    for (int k = 0; k < GetNumMeshes(); k++)
      if (global_node_idx >= inst_[k].node_offset && global_node_idx < inst_[k].node_offset + inst_[k].num_nodes) return k;
    return -1;
  }
  void Clear() {
    inst_.clear();
    field_set_.clear();
    nodes_.resize(0, 3);
    elems_.resize(0, 0);
    fields_.resize(0);
    has_fields_ = false;
  }

 private:
  std::vector<MeshInstance> inst_;
  std::vector<bool> field_set_;
  tlfea::MatrixXd nodes_;
  tlfea::MatrixXi elems_;
  tlfea::VectorXd fields_;
  bool has_fields_ = false;
};

}  // namespace ANCFCPUUtils
