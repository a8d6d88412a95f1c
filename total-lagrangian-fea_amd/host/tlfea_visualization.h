// tlfea_visualization.h -- the VTU exporters the reference drivers call for eyeballing results
// (lib_utils/visualization_utils.h:491-589 ExportMeshToVTU, :718-846 ExportMeshWithDisplacement,
// :848-955 ExportANCF3443ToVTU, :974-1097 ExportANCF3243ToVTU).  Same static member names / argument order under
// ANCFCPUUtils::VisualizationUtils, same ASCII UnstructuredGrid layout (precision 15, scientific) so that ParaView
// states written for the reference's files open these.  One emitter serves all of them: a writer hands it the points,
// the cells (fixed arity) and optional point data.  The contact-patch exporters (VTP/CSV/JSON) belong to the collision
// subsystem and are not built.  Included by tlfea_facade.h.
#pragma once
#include <array>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

namespace ANCFCPUUtils {

struct VisualizationUtils {
  using P3 = std::array<double, 3>;
  struct PointField {
    std::string name;
    int components;
    std::vector<double> values;  // n_points * components
  };

  // points, cells of `arity` point ids each (ids relative to `points`), VTK cell type (10 tetra, 12 hexahedron)
  static bool WriteVTU(const std::string& filename, const std::vector<P3>& points, const std::vector<int>& cells,
                       int arity, int vtk_type, const std::vector<PointField>& fields = {}) {
    std::ofstream file(filename);
    if (!file.is_open()) {
      std::cerr << "Error: Cannot open file " << filename << " for writing" << std::endl;
      return false;
    }
    const size_t n_cells = arity ? cells.size() / arity : 0;
    file << std::setprecision(15) << std::scientific;
    file << "<?xml version=\"1.0\"?>\n"
         << "<VTKFile type=\"UnstructuredGrid\" version=\"1.0\" byte_order=\"LittleEndian\">\n"
         << "  <UnstructuredGrid>\n"
         << "    <Piece NumberOfPoints=\"" << points.size() << "\" NumberOfCells=\"" << n_cells << "\">\n"
         << "      <Points>\n"
         << "        <DataArray type=\"Float64\" NumberOfComponents=\"3\" format=\"ascii\">\n";
    for (const P3& p : points) file << "          " << p[0] << " " << p[1] << " " << p[2] << "\n";
    file << "        </DataArray>\n      </Points>\n";
    if (!fields.empty()) {
      file << "      <PointData";
      for (const PointField& f : fields)
        if (f.components == 1) {
          file << " Scalars=\"" << f.name << "\"";
          break;
        }
      for (const PointField& f : fields)
        if (f.components == 3) {
          file << " Vectors=\"" << f.name << "\"";
          break;
        }
      file << ">\n";
      for (const PointField& f : fields) {
        file << "        <DataArray type=\"Float64\" Name=\"" << f.name << "\"";
        if (f.components > 1) file << " NumberOfComponents=\"" << f.components << "\"";
        file << " format=\"ascii\">\n";
        for (size_t i = 0; i < points.size(); i++) {
          file << "         ";
          for (int c = 0; c < f.components; c++) {
            const size_t k = i * f.components + c;
            file << " " << (k < f.values.size() ? f.values[k] : 0.0);
          }
          file << "\n";
        }
        file << "        </DataArray>\n";
      }
      file << "      </PointData>\n";
    }
    file << "      <Cells>\n        <DataArray type=\"Int32\" Name=\"connectivity\" format=\"ascii\">\n";
    for (size_t c = 0; c < n_cells; c++) {
      file << "         ";
      for (int k = 0; k < arity; k++) file << " " << cells[c * arity + k];
      file << "\n";
    }
    file << "        </DataArray>\n        <DataArray type=\"Int32\" Name=\"offsets\" format=\"ascii\">\n";
    for (size_t c = 0; c < n_cells; c++) file << "          " << (c + 1) * arity << "\n";
    file << "        </DataArray>\n        <DataArray type=\"UInt8\" Name=\"types\" format=\"ascii\">\n";
    for (size_t c = 0; c < n_cells; c++) file << "          " << vtk_type << "\n";
    file << "        </DataArray>\n      </Cells>\n    </Piece>\n  </UnstructuredGrid>\n</VTKFile>\n";
    return static_cast<bool>(file);
  }

  // T10 mesh as linear tets on the four corner nodes + nodal scalar "pressure" (visualization_utils.h:491-589)
  static bool ExportMeshToVTU(const tlfea::MatrixXd& nodes, const tlfea::MatrixXi& elements,
                              const tlfea::VectorXd& scalarField, const std::string& filename) {
    std::vector<P3> pts(nodes.rows());
    PointField p{"pressure", 1, std::vector<double>(nodes.rows(), 0.0)};
    for (int i = 0; i < nodes.rows(); i++) {
      pts[i] = {nodes(i, 0), nodes(i, 1), nodes(i, 2)};
      if (i < scalarField.size()) p.values[i] = scalarField(i);
    }
    if (!WriteVTU(filename, pts, corner_tets(elements), 4, 10, {p})) return false;
    std::cout << "Exported mesh with " << nodes.rows() << " nodes and " << elements.rows() << " elements to " << filename
              << std::endl;
    return true;
  }

  // T10 mesh at `nodes` + nodal "displacement" vectors [dx0, dy0, dz0, dx1, ...] and their magnitude; entries past the
  // end of `displacement` read as zero (visualization_utils.h:718-846)
  static bool ExportMeshWithDisplacement(const tlfea::MatrixXd& nodes, const tlfea::MatrixXi& elements,
                                         const tlfea::VectorXd& displacement, const std::string& filename) {
    const int n = nodes.rows();
    std::vector<P3> pts(n);
    PointField d{"displacement", 3, std::vector<double>(3 * (size_t)n, 0.0)}, m{"displacement_magnitude", 1, std::vector<double>(n)};
    for (int i = 0; i < n; i++) {
      pts[i] = {nodes(i, 0), nodes(i, 1), nodes(i, 2)};
      double s = 0.0;
      for (int c = 0; c < 3; c++) {
        const double v = 3 * i + c < displacement.size() ? displacement(3 * i + c) : 0.0;
        d.values[3 * (size_t)i + c] = v;
        s += v * v;
      }
      m.values[i] = std::sqrt(s);
    }
    return WriteVTU(filename, pts, corner_tets(elements), 4, 10, {m, d});
  }

  // One hexahedron per shell: the four corner positions (coefficient slot 0 of each node) pushed -/+ thickness/2 along
  // the element normal (p1-p0) x (p3-p0); bottom face first (visualization_utils.h:848-955)
  static bool ExportANCF3443ToVTU(const tlfea::VectorXd& x12, const tlfea::VectorXd& y12, const tlfea::VectorXd& z12,
                                  const tlfea::MatrixXi& element_connectivity, double thickness,
                                  const std::string& filename) {
    const int E = element_connectivity.rows();
    std::vector<P3> pts;
    pts.reserve(8 * (size_t)E);
    for (int e = 0; e < E; e++) {
      P3 p[4];
      for (int k = 0; k < 4; k++) p[k] = position(x12, y12, z12, element_connectivity(e, k));
      P3 nrm = cross(sub(p[1], p[0]), sub(p[3], p[0]));
      if (!normalize(nrm)) nrm = {0.0, 0.0, 1.0};
      for (double side : {-0.5, 0.5})
        for (int k = 0; k < 4; k++) pts.push_back(axpy(side * thickness, nrm, p[k]));
    }
    return WriteVTU(filename, pts, iota_cells(E, 8), 8, 12);
  }

  // One hexahedron per beam: a width x height rectangle at each end node, in the frame built from the chord tangent t,
  // n = t x ref (ref = z, or y when t is within ~26 degrees of z) and b = t x n (visualization_utils.h:974-1097)
  static bool ExportANCF3243ToVTU(const tlfea::VectorXd& x12, const tlfea::VectorXd& y12, const tlfea::VectorXd& z12,
                                  const tlfea::MatrixXi& element_connectivity, double width, double height,
                                  const std::string& filename) {
    const int E = element_connectivity.rows();
    std::vector<P3> pts;
    pts.reserve(8 * (size_t)E);
    const double su[4] = {-0.5, 0.5, 0.5, -0.5}, sv[4] = {-0.5, -0.5, 0.5, 0.5};
    for (int e = 0; e < E; e++) {
      const P3 end[2] = {position(x12, y12, z12, element_connectivity(e, 0)),
                         position(x12, y12, z12, element_connectivity(e, 1))};
      P3 t = sub(end[1], end[0]);
      if (!normalize(t)) t = {1.0, 0.0, 0.0};
      const P3 ref = std::fabs(t[2]) > 0.9 ? P3{0.0, 1.0, 0.0} : P3{0.0, 0.0, 1.0};
      P3 n = cross(t, ref);
      if (!normalize(n)) n = {0.0, 1.0, 0.0};
      P3 b = cross(t, n);
      if (!normalize(b)) b = {0.0, 0.0, 1.0};
      for (int s = 0; s < 2; s++)
        for (int k = 0; k < 4; k++) pts.push_back(axpy(sv[k] * height, b, axpy(su[k] * width, n, end[s])));
    }
    return WriteVTU(filename, pts, iota_cells(E, 8), 8, 12);
  }

 private:
  static P3 position(const tlfea::VectorXd& x, const tlfea::VectorXd& y, const tlfea::VectorXd& z, int node) {
    return {x(4 * node), y(4 * node), z(4 * node)};
  }
  static P3 sub(const P3& a, const P3& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
  static P3 axpy(double s, const P3& d, const P3& p) { return {p[0] + s * d[0], p[1] + s * d[1], p[2] + s * d[2]}; }
  static P3 cross(const P3& a, const P3& b) {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  }
  static bool normalize(P3& v) {
    const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (n < 1e-12) return false;
    for (double& c : v) c /= n;
    return true;
  }
  static std::vector<int> iota_cells(int n_cells, int arity) {
    std::vector<int> c((size_t)n_cells * arity);
    for (size_t i = 0; i < c.size(); i++) c[i] = static_cast<int>(i);
    return c;
  }
  static std::vector<int> corner_tets(const tlfea::MatrixXi& elements) {
    std::vector<int> c;
    c.reserve(4 * (size_t)elements.rows());
    for (int e = 0; e < elements.rows(); e++)
      for (int k = 0; k < 4; k++) c.push_back(elements(e, k));
    return c;
  }
};

}  // namespace ANCFCPUUtils
