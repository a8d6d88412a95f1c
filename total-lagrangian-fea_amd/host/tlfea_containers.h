// tlfea_containers.h -- minimal column-major containers standing in for the Eigen types in the reference's
// public signatures (Eigen is not part of this engine: SURVEY.md section 7 "hard parts").  Only what the
// element / solver API needs: data(), size(), rows(), cols(), operator(), resize, setZero.
#pragma once
#include <cstddef>
#include <vector>

namespace tlfea {

template <typename T>
class Vec {
 public:
  Vec() = default;
  explicit Vec(int n) : v_(static_cast<size_t>(n)) {}
  void resize(int n) { v_.assign(static_cast<size_t>(n), T()); }
  void setZero() { std::fill(v_.begin(), v_.end(), T()); }
  int size() const { return static_cast<int>(v_.size()); }
  T* data() { return v_.data(); }
  const T* data() const { return v_.data(); }
  T& operator()(int i) { return v_[static_cast<size_t>(i)]; }
  const T& operator()(int i) const { return v_[static_cast<size_t>(i)]; }

 private:
  std::vector<T> v_;
};

// column-major like Eigen::Matrix<T, Dynamic, Dynamic>: (i,j) -> data[j*rows + i]
template <typename T>
class Mat {
 public:
  Mat() = default;
  Mat(int r, int c) { resize(r, c); }
  void resize(int r, int c) {
    r_ = r;
    c_ = c;
    v_.assign(static_cast<size_t>(r) * c, T());
  }
  void setZero() { std::fill(v_.begin(), v_.end(), T()); }
  int rows() const { return r_; }
  int cols() const { return c_; }
  int size() const { return r_ * c_; }
  T* data() { return v_.data(); }
  const T* data() const { return v_.data(); }
  T& operator()(int i, int j) { return v_[static_cast<size_t>(j) * r_ + i]; }
  const T& operator()(int i, int j) const { return v_[static_cast<size_t>(j) * r_ + i]; }

 private:
  int r_ = 0, c_ = 0;
  std::vector<T> v_;
};

using VectorXd = Vec<double>;
using VectorXi = Vec<int>;
using MatrixXd = Mat<double>;
using MatrixXi = Mat<int>;

}  // namespace tlfea
