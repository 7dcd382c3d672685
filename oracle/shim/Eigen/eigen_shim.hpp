// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// A small, value-semantic stand-in for the parts of Eigen 3 that the okvis reference sources use, written
// from scratch for this repository (no Eigen code).  It exists for ONE purpose: to compile the reference's own
// factor / marginalisation sources UNMODIFIED from /root/reference into oracle/_ref/ (see oracle/ref/Makefile)
// so that the restatement in oracle/ and the golden fixtures can be pinned against the reference's own lines.
// Eigen itself is not installed in this image and there is no network.
//
// Design: no expression templates.  Every operation evaluates eagerly into a plain Matrix; only block(),
// segment(), col(), ... on non-const objects and Map<> are views.  Assignment from a view goes through a
// temporary, so there are no aliasing rules to remember.  Arithmetic follows the textbook order (dot products
// left to right); results therefore agree with a real Eigen build to rounding, not bit for bit.
// Algorithms whose *behaviour* the reference relies on are restated deliberately:
//   * LLT: the unblocked right-looking Cholesky that stops at the first non-positive pivot and leaves the
//     remaining columns untouched (what Eigen's llt_inplace does for n < 32) — PoseError's rank-deficient
//     first-pose prior depends on that;
//   * inverse(): closed form up to 3x3, partial-pivot LU above;
//   * SelfAdjointEigenSolver: cyclic Jacobi, eigenvalues ascending;
//   * ColPivHouseholderQR: Householder reflections with the largest remaining column first (norms down-dated, the
//     chosen one recomputed), rank() = pivots above epsilon * size * largest pivot — the rank test of
//     ProbabilisticStereoTriangulator::getUncertainty;
//   * Quaternion product / toRotationMatrix / rotation-matrix -> quaternion: the standard formulas.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <memory>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW_IF(x)
#define EIGEN_DEFINE_STL_VECTOR_SPECIALIZATION(...)
#define EIGEN_STATIC_ASSERT_VECTOR_SPECIFIC_SIZE(TYPE, SIZE)                                        \
  static_assert(TYPE::SizeAtCompileTime == SIZE || TYPE::SizeAtCompileTime == ::Eigen::Dynamic,     \
                "vector of the wrong size")
#define EIGEN_STATIC_ASSERT_MATRIX_SPECIFIC_SIZE(TYPE, ROWS, COLS)                                  \
  static_assert((TYPE::RowsAtCompileTime == ROWS || TYPE::RowsAtCompileTime == ::Eigen::Dynamic) && \
                    (TYPE::ColsAtCompileTime == COLS || TYPE::ColsAtCompileTime == ::Eigen::Dynamic), \
                "matrix of the wrong size")
#define EIGEN_STATIC_ASSERT_VECTOR_ONLY(TYPE) static_assert(TYPE::IsVectorAtCompileTime, "vector expected")
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 2
#define EIGEN_MINOR_VERSION 0

#include <execinfo.h>
namespace Eigen {
inline void eshim_fail(const char* what) {
  void* bt[32];
  int n = backtrace(bt, 32);
  std::cerr << "eigen_shim: " << what << std::endl;
  backtrace_symbols_fd(bt, n, 2);
  std::abort();
}

constexpr int Dynamic = -1;
constexpr int Infinity = -1;
enum StorageOptions { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
enum { Unaligned = 0, Aligned = 1 };
typedef std::ptrdiff_t Index;
typedef std::ptrdiff_t DenseIndex;

template <class T>
using aligned_allocator = std::allocator<T>;

namespace internal {
template <class T>
struct traits;
template <class T>
struct traits<const T> : traits<T> {};
struct SizeTag {};
constexpr int pick(int a, int b) { return a != Dynamic ? a : b; }
constexpr int mul(int a, int b) { return (a == Dynamic || b == Dynamic) ? Dynamic : a * b; }
template <class T>
struct is_scalar : std::integral_constant<bool, std::is_arithmetic<T>::value> {};
}  // namespace internal

template <class Derived>
class MatrixBase;
template <class S, int R, int C, int O = ColMajor, int MR = R, int MC = C>
class Matrix;
template <class Xpr, int BR = Dynamic, int BC = Dynamic>
class Block;
template <class Plain, int MapOptions = Unaligned>
class Map;
template <class S, int R, int C>
class Array;
template <class S>
class Quaternion;
template <class S>
class AngleAxis;
template <class M>
class LLT;
template <class M>
class ColPivHouseholderQR;

// ------------------------------------------------------------------------------------------------------
// comma initialiser:  m << a, b, c;
// ------------------------------------------------------------------------------------------------------
template <class Xpr>
class CommaInitializer {
 public:
  CommaInitializer(Xpr& x) : x_(x), k_(0) {}
  template <class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type>
  CommaInitializer& operator,(const T& v) {
    put((typename Xpr::Scalar)v);
    return *this;
  }
  template <class D>
  CommaInitializer& operator,(const MatrixBase<D>& m) {
    // only vectors stacked into vectors and row-wise fills of equal-height blocks are needed
    for (Index j = 0; j < m.cols(); ++j)
      for (Index i = 0; i < m.rows(); ++i) put(m.coeff(i, j));
    return *this;
  }
  template <class T>
  void put(T v) {
    const Index c = x_.cols();
    x_.coeffRef(k_ / c, k_ % c) = v;
    ++k_;
  }
  Xpr& finished() { return x_; }

 private:
  Xpr& x_;
  Index k_;
};

// ------------------------------------------------------------------------------------------------------
// MatrixBase
// ------------------------------------------------------------------------------------------------------
template <class Derived>
class MatrixBase {
 public:
  typedef typename internal::traits<Derived>::Scalar Scalar;
  typedef Scalar RealScalar;
  enum {
    RowsAtCompileTime = internal::traits<Derived>::RowsAtCompileTime,
    ColsAtCompileTime = internal::traits<Derived>::ColsAtCompileTime,
    SizeAtCompileTime = internal::mul(RowsAtCompileTime, ColsAtCompileTime),
    IsVectorAtCompileTime = (RowsAtCompileTime == 1 || ColsAtCompileTime == 1),
    IsRowMajor = 0
  };
  typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;
  typedef Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> TransposeReturnType;
  typedef Eigen::Index Index;

  Derived& derived() { return *static_cast<Derived*>(this); }
  const Derived& derived() const { return *static_cast<const Derived*>(this); }
  Derived& const_cast_derived() const { return *const_cast<Derived*>(static_cast<const Derived*>(this)); }

  Index rows() const { return derived().rows(); }
  Index cols() const { return derived().cols(); }
  Index size() const { return rows() * cols(); }
  Index innerSize() const { return rows(); }
  Index outerSize() const { return cols(); }

  decltype(auto) coeff(Index i, Index j) const { return derived().coeff(i, j); }
  decltype(auto) coeff(Index i) const { return cols() == 1 ? derived().coeff(i, 0) : derived().coeff(0, i); }
  Scalar& coeffRef(Index i, Index j) { return derived().coeffRef(i, j); }
  Scalar& coeffRef(Index i) { return cols() == 1 ? derived().coeffRef(i, 0) : derived().coeffRef(0, i); }
  decltype(auto) operator()(Index i, Index j) const { return derived().coeff(i, j); }
  Scalar& operator()(Index i, Index j) { return derived().coeffRef(i, j); }
  decltype(auto) operator()(Index i) const { return coeff(i); }
  Scalar& operator()(Index i) { return coeffRef(i); }
  decltype(auto) operator[](Index i) const { return coeff(i); }
  Scalar& operator[](Index i) { return coeffRef(i); }
  decltype(auto) x() const { return coeff(0); }
  decltype(auto) y() const { return coeff(1); }
  decltype(auto) z() const { return coeff(2); }
  decltype(auto) w() const { return coeff(3); }
  Scalar& x() { return coeffRef(0); }
  Scalar& y() { return coeffRef(1); }
  Scalar& z() { return coeffRef(2); }
  Scalar& w() { return coeffRef(3); }

  PlainObject eval() const { return PlainObject(derived()); }
  Derived& noalias() { return derived(); }

  // ---- assignment (alias-safe: views are copied first) ----
  template <class Other>
  Derived& assign(const MatrixBase<Other>& o) {
    typename MatrixBase<Other>::PlainObject tmp(o.rows(), o.cols(), internal::SizeTag());
    for (Index j = 0; j < o.cols(); ++j)
      for (Index i = 0; i < o.rows(); ++i) tmp.coeffRef(i, j) = o.coeff(i, j);
    derived().resizeLike(tmp.rows(), tmp.cols());
    if (rows() == tmp.rows() && cols() == tmp.cols()) {
      for (Index j = 0; j < cols(); ++j)
        for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = tmp.coeff(i, j);
    } else {  // vector <- transposed vector (Eigen allows it for vectors)
      assert(size() == tmp.size() && (tmp.rows() == 1 || tmp.cols() == 1));
      for (Index i = 0; i < size(); ++i) coeffRef(i) = tmp.coeff(i);
    }
    return derived();
  }
  template <class Other>
  Derived& operator=(const MatrixBase<Other>& o) {
    return assign(o);
  }
  Derived& operator=(const MatrixBase& o) { return assign(o); }
  template <class Other>
  Derived& operator+=(const MatrixBase<Other>& o) {
    return assign(*this + o);
  }
  template <class Other>
  Derived& operator-=(const MatrixBase<Other>& o) {
    return assign(*this - o);
  }
  template <class Other>
  Derived& operator*=(const MatrixBase<Other>& o) {
    return assign(*this * o);
  }
  template <class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type>
  Derived& operator*=(const T& s) {
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) coeffRef(i, j) *= (Scalar)s;
    return derived();
  }
  template <class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type>
  Derived& operator/=(const T& s) {
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) coeffRef(i, j) /= (Scalar)s;
    return derived();
  }
  template <class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type>
  CommaInitializer<Derived> operator<<(const T& v) {
    CommaInitializer<Derived> c(derived());
    c.put((Scalar)v);
    return c;
  }
  template <class D>
  CommaInitializer<Derived> operator<<(const MatrixBase<D>& m) {
    CommaInitializer<Derived> c(derived());
    c, m;
    return c;
  }

  // ---- setters ----
  Derived& setConstant(Scalar v) {
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = v;
    return derived();
  }
  Derived& fill(Scalar v) { return setConstant(v); }
  Derived& setZero() { return setConstant(Scalar(0)); }
  Derived& setOnes() { return setConstant(Scalar(1)); }
  Derived& setZero(Index n) {
    derived().resize(n);
    return setZero();
  }
  Derived& setZero(Index r, Index c) {
    derived().resize(r, c);
    return setZero();
  }
  Derived& setIdentity() {
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0);
    return derived();
  }
  Derived& setIdentity(Index r, Index c) {
    derived().resize(r, c);
    return setIdentity();
  }
  Derived& setRandom() {
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = Scalar(2.0 * std::rand() / RAND_MAX - 1.0);
    return derived();
  }
  void normalize() {
    const Scalar n = norm();
    if (n > Scalar(0)) *this /= n;
  }
  void transposeInPlace() { assign(transpose()); }

  // ---- reductions and unary ----
  Scalar squaredNorm() const {
    Scalar s = 0;
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) s += coeff(i, j) * coeff(i, j);
    return s;
  }
  Scalar norm() const { return std::sqrt(squaredNorm()); }
  Scalar stableNorm() const { return norm(); }
  template <int p>
  Scalar lpNorm() const {
    Scalar s = 0;
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) {
        const Scalar a = std::fabs(coeff(i, j));
        if (p == Infinity) s = std::max(s, a);
        else if (p == 1) s += a;
        else s += a * a;
      }
    return p == 2 ? std::sqrt(s) : s;
  }
  PlainObject normalized() const {
    PlainObject r(derived());
    r.normalize();
    return r;
  }
  Scalar sum() const {
    Scalar s = 0;
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) s += coeff(i, j);
    return s;
  }
  Scalar mean() const { return sum() / Scalar(size()); }
  Scalar prod() const {
    Scalar s = 1;
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) s *= coeff(i, j);
    return s;
  }
  Scalar trace() const {
    Scalar s = 0;
    for (Index i = 0; i < std::min(rows(), cols()); ++i) s += coeff(i, i);
    return s;
  }
  Scalar maxCoeff() const {
    Scalar s = coeff(0, 0);
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) s = std::max(s, coeff(i, j));
    return s;
  }
  Scalar minCoeff() const {
    Scalar s = coeff(0, 0);
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) s = std::min(s, coeff(i, j));
    return s;
  }
  template <class I>
  Scalar maxCoeff(I* idx) const {
    Index k = 0;
    for (Index i = 1; i < size(); ++i)
      if (coeff(i) > coeff(k)) k = i;
    *idx = (I)k;
    return coeff(k);
  }
  template <class I>
  Scalar minCoeff(I* idx) const {
    Index k = 0;
    for (Index i = 1; i < size(); ++i)
      if (coeff(i) < coeff(k)) k = i;
    *idx = (I)k;
    return coeff(k);
  }
  bool allFinite() const {
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i)
        if (!std::isfinite(coeff(i, j))) return false;
    return true;
  }
  bool hasNaN() const {
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i)
        if (std::isnan(coeff(i, j))) return true;
    return false;
  }
  template <class Other>
  bool isApprox(const MatrixBase<Other>& o, Scalar prec = Scalar(1e-12)) const {
    return (*this - o).squaredNorm() <= prec * prec * std::min(squaredNorm(), o.squaredNorm());
  }
  template <class F>
  PlainObject unaryExpr(F f) const {
    PlainObject r(rows(), cols(), internal::SizeTag());
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) = f(coeff(i, j));
    return r;
  }
  PlainObject cwiseAbs() const {
    return unaryExpr([](Scalar v) { return std::fabs(v); });
  }
  PlainObject cwiseAbs2() const {
    return unaryExpr([](Scalar v) { return v * v; });
  }
  PlainObject cwiseSqrt() const {
    return unaryExpr([](Scalar v) { return std::sqrt(v); });
  }
  PlainObject cwiseInverse() const {
    return unaryExpr([](Scalar v) { return Scalar(1) / v; });
  }
  template <class Other>
  PlainObject cwiseProduct(const MatrixBase<Other>& o) const {
    PlainObject r(rows(), cols(), internal::SizeTag());
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) = coeff(i, j) * o.coeff(i, j);
    return r;
  }
  template <class Other>
  PlainObject cwiseQuotient(const MatrixBase<Other>& o) const {
    PlainObject r(rows(), cols(), internal::SizeTag());
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) = coeff(i, j) / o.coeff(i, j);
    return r;
  }
  template <class T>
  Matrix<T, RowsAtCompileTime, ColsAtCompileTime> cast() const {
    Matrix<T, RowsAtCompileTime, ColsAtCompileTime> r(rows(), cols(), internal::SizeTag());
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) = (T)coeff(i, j);
    return r;
  }
  TransposeReturnType transpose() const {
    TransposeReturnType r(cols(), rows(), internal::SizeTag());
    for (Index j = 0; j < cols(); ++j)
      for (Index i = 0; i < rows(); ++i) r.coeffRef(j, i) = coeff(i, j);
    return r;
  }
  TransposeReturnType adjoint() const { return transpose(); }
  template <class Other>
  Scalar dot(const MatrixBase<Other>& o) const {
    assert(size() == o.size());
    Scalar s = 0;
    for (Index i = 0; i < size(); ++i) s += coeff(i) * o.coeff(i);
    return s;
  }
  template <class Other>
  Matrix<Scalar, 3, 1> cross(const MatrixBase<Other>& o) const {
    Matrix<Scalar, 3, 1> r;
    r[0] = coeff(1) * o.coeff(2) - coeff(2) * o.coeff(1);
    r[1] = coeff(2) * o.coeff(0) - coeff(0) * o.coeff(2);
    r[2] = coeff(0) * o.coeff(1) - coeff(1) * o.coeff(0);
    return r;
  }
  Matrix<Scalar, (IsVectorAtCompileTime ? SizeAtCompileTime : Dynamic), 1> diagonal() const {
    const Index n = std::min(rows(), cols());
    Matrix<Scalar, (IsVectorAtCompileTime ? SizeAtCompileTime : Dynamic), 1> d(n, 1, internal::SizeTag());
    for (Index i = 0; i < n; ++i) d.coeffRef(i, 0) = coeff(i, i);
    return d;
  }
  // vector -> dense diagonal matrix (eager; products with it are exact apart from added zeros)
  Matrix<Scalar, SizeAtCompileTime, SizeAtCompileTime> asDiagonal() const {
    const Index n = size();
    Matrix<Scalar, SizeAtCompileTime, SizeAtCompileTime> d(n, n, internal::SizeTag());
    d.setZero();
    for (Index i = 0; i < n; ++i) d.coeffRef(i, i) = coeff(i);
    return d;
  }
  Array<Scalar, RowsAtCompileTime, ColsAtCompileTime> array() const;
  PlainObject matrix() const { return eval(); }
  PlainObject inverse() const;
  Scalar determinant() const;
  LLT<PlainObject> llt() const;
  ColPivHouseholderQR<PlainObject> colPivHouseholderQr() const;
  // fixed sizes up to 4x4 in Eigen: invertible = |det| > threshold, the inverse from the cofactors / det
  template <class Result>
  void computeInverseWithCheck(Result& inverse, bool& invertible, const Scalar& absDeterminantThreshold = Scalar(1e-12)) const {
    invertible = std::fabs(determinant()) > absDeterminantThreshold;
    if (invertible) inverse = this->inverse();
  }
  Matrix<Scalar, internal::pick(SizeAtCompileTime, Dynamic) == Dynamic ? Dynamic : SizeAtCompileTime + 1, 1>
  homogeneous() const {
    Matrix<Scalar, internal::pick(SizeAtCompileTime, Dynamic) == Dynamic ? Dynamic : SizeAtCompileTime + 1, 1> r(
        size() + 1, 1, internal::SizeTag());
    for (Index i = 0; i < size(); ++i) r.coeffRef(i, 0) = coeff(i);
    r.coeffRef(size(), 0) = Scalar(1);
    return r;
  }

  // ---- sub-matrices: views on non-const objects, copies on const ones ----
#define ESHIM_BLOCK_FIXED(NAME, I0, J0)                                             \
  template <int BR, int BC>                                                         \
  Block<Derived, BR, BC> NAME() {                                                   \
    return Block<Derived, BR, BC>(derived(), I0, J0, BR, BC);                       \
  }                                                                                 \
  template <int BR, int BC>                                                         \
  Matrix<Scalar, BR, BC> NAME() const {                                             \
    return Matrix<Scalar, BR, BC>(Block<Derived, BR, BC>(const_cast_derived(), I0, J0, BR, BC)); \
  }
#define ESHIM_BLOCK_DYN(NAME, I0, J0)                                               \
  Block<Derived> NAME(Index br, Index bc) { return Block<Derived>(derived(), I0, J0, br, bc); } \
  Matrix<Scalar, Dynamic, Dynamic> NAME(Index br, Index bc) const {                 \
    return Matrix<Scalar, Dynamic, Dynamic>(Block<Derived>(const_cast_derived(), I0, J0, br, bc)); \
  }
  ESHIM_BLOCK_FIXED(topLeftCorner, 0, 0)
  ESHIM_BLOCK_FIXED(topRightCorner, 0, cols() - BC)
  ESHIM_BLOCK_FIXED(bottomLeftCorner, rows() - BR, 0)
  ESHIM_BLOCK_FIXED(bottomRightCorner, rows() - BR, cols() - BC)
  ESHIM_BLOCK_DYN(topLeftCorner, 0, 0)
  ESHIM_BLOCK_DYN(topRightCorner, 0, cols() - bc)
  ESHIM_BLOCK_DYN(bottomLeftCorner, rows() - br, 0)
  ESHIM_BLOCK_DYN(bottomRightCorner, rows() - br, cols() - bc)
#undef ESHIM_BLOCK_FIXED
#undef ESHIM_BLOCK_DYN
  template <int BR, int BC>
  Block<Derived, BR, BC> block(Index i, Index j) {
    return Block<Derived, BR, BC>(derived(), i, j, BR, BC);
  }
  template <int BR, int BC>
  Matrix<Scalar, BR, BC> block(Index i, Index j) const {
    return Matrix<Scalar, BR, BC>(Block<Derived, BR, BC>(const_cast_derived(), i, j, BR, BC));
  }
  Block<Derived> block(Index i, Index j, Index br, Index bc) { return Block<Derived>(derived(), i, j, br, bc); }
  Matrix<Scalar, Dynamic, Dynamic> block(Index i, Index j, Index br, Index bc) const {
    return Matrix<Scalar, Dynamic, Dynamic>(Block<Derived>(const_cast_derived(), i, j, br, bc));
  }
  // rows / columns
  Block<Derived, RowsAtCompileTime, 1> col(Index j) {
    return Block<Derived, RowsAtCompileTime, 1>(derived(), 0, j, rows(), 1);
  }
  Matrix<Scalar, RowsAtCompileTime, 1> col(Index j) const {
    return Matrix<Scalar, RowsAtCompileTime, 1>(
        Block<Derived, RowsAtCompileTime, 1>(const_cast_derived(), 0, j, rows(), 1));
  }
  Block<Derived, 1, ColsAtCompileTime> row(Index i) {
    return Block<Derived, 1, ColsAtCompileTime>(derived(), i, 0, 1, cols());
  }
  Matrix<Scalar, 1, ColsAtCompileTime> row(Index i) const {
    return Matrix<Scalar, 1, ColsAtCompileTime>(
        Block<Derived, 1, ColsAtCompileTime>(const_cast_derived(), i, 0, 1, cols()));
  }
  Block<Derived> topRows(Index n) { return block(0, 0, n, cols()); }
  Block<Derived> bottomRows(Index n) { return block(rows() - n, 0, n, cols()); }
  Block<Derived> leftCols(Index n) { return block(0, 0, rows(), n); }
  Block<Derived> rightCols(Index n) { return block(0, cols() - n, rows(), n); }
  Matrix<Scalar, Dynamic, Dynamic> topRows(Index n) const { return block(0, 0, n, cols()); }
  Matrix<Scalar, Dynamic, Dynamic> bottomRows(Index n) const { return block(rows() - n, 0, n, cols()); }
  Matrix<Scalar, Dynamic, Dynamic> leftCols(Index n) const { return block(0, 0, rows(), n); }
  Matrix<Scalar, Dynamic, Dynamic> rightCols(Index n) const { return block(0, cols() - n, rows(), n); }
  // vector segments (column or row vectors)
  template <int N>
  Block<Derived, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> segment(Index i) {
    typedef Block<Derived, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> B;
    return cols() == 1 ? B(derived(), i, 0, N, 1) : B(derived(), 0, i, 1, N);
  }
  template <int N>
  Matrix<Scalar, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> segment(Index i) const {
    typedef Block<Derived, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> B;
    return Matrix<Scalar, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)>(
        cols() == 1 ? B(const_cast_derived(), i, 0, N, 1) : B(const_cast_derived(), 0, i, 1, N));
  }
  Block<Derived, (ColsAtCompileTime == 1 ? Dynamic : 1), (ColsAtCompileTime == 1 ? 1 : Dynamic)> segment(Index i,
                                                                                                          Index n) {
    typedef Block<Derived, (ColsAtCompileTime == 1 ? Dynamic : 1), (ColsAtCompileTime == 1 ? 1 : Dynamic)> B;
    return cols() == 1 ? B(derived(), i, 0, n, 1) : B(derived(), 0, i, 1, n);
  }
  Matrix<Scalar, (ColsAtCompileTime == 1 ? Dynamic : 1), (ColsAtCompileTime == 1 ? 1 : Dynamic)> segment(
      Index i, Index n) const {
    typedef Block<Derived, (ColsAtCompileTime == 1 ? Dynamic : 1), (ColsAtCompileTime == 1 ? 1 : Dynamic)> B;
    return Matrix<Scalar, (ColsAtCompileTime == 1 ? Dynamic : 1), (ColsAtCompileTime == 1 ? 1 : Dynamic)>(
        cols() == 1 ? B(const_cast_derived(), i, 0, n, 1) : B(const_cast_derived(), 0, i, 1, n));
  }
  template <int N>
  auto head() -> decltype(this->template segment<N>(0)) {
    return this->template segment<N>(0);
  }
  template <int N>
  auto head() const -> decltype(this->template segment<N>(0)) {
    return this->template segment<N>(0);
  }
  template <int N>
  auto tail() -> decltype(this->template segment<N>(0)) {
    return this->template segment<N>(size() - N);
  }
  template <int N>
  auto tail() const -> decltype(this->template segment<N>(0)) {
    return this->template segment<N>(size() - N);
  }
  auto head(Index n) -> decltype(this->segment(0, n)) { return segment(0, n); }
  auto head(Index n) const -> decltype(this->segment(0, n)) { return segment(0, n); }
  auto tail(Index n) -> decltype(this->segment(0, n)) { return segment(size() - n, n); }
  auto tail(Index n) const -> decltype(this->segment(0, n)) { return segment(size() - n, n); }

  // ---- static constructors ----
  static PlainObject Constant(Scalar v) {
    PlainObject r;
    r.setConstant(v);
    return r;
  }
  static PlainObject Constant(Index n, Scalar v) {
    PlainObject r;
    r.resize(n);
    r.setConstant(v);
    return r;
  }
  static PlainObject Constant(Index rr, Index cc, Scalar v) {
    PlainObject r(rr, cc, internal::SizeTag());
    r.setConstant(v);
    return r;
  }
  static PlainObject Zero() { return Constant(Scalar(0)); }
  static PlainObject Zero(Index n) { return Constant(n, Scalar(0)); }
  static PlainObject Zero(Index r, Index c) { return Constant(r, c, Scalar(0)); }
  static PlainObject Ones() { return Constant(Scalar(1)); }
  static PlainObject Ones(Index n) { return Constant(n, Scalar(1)); }
  static PlainObject Ones(Index r, Index c) { return Constant(r, c, Scalar(1)); }
  static PlainObject Identity() {
    PlainObject r;
    r.setIdentity();
    return r;
  }
  static PlainObject Identity(Index rr, Index cc) {
    PlainObject r(rr, cc, internal::SizeTag());
    r.setIdentity();
    return r;
  }
  static PlainObject Random() {
    PlainObject r;
    r.setRandom();
    return r;
  }
  static PlainObject Random(Index n) {
    PlainObject r;
    r.resize(n);
    r.setRandom();
    return r;
  }
  static PlainObject Random(Index rr, Index cc) {
    PlainObject r(rr, cc, internal::SizeTag());
    r.setRandom();
    return r;
  }
  static PlainObject UnitX() {
    PlainObject r = Zero();
    r[0] = 1;
    return r;
  }
  static PlainObject UnitY() {
    PlainObject r = Zero();
    r[1] = 1;
    return r;
  }
  static PlainObject UnitZ() {
    PlainObject r = Zero();
    r[2] = 1;
    return r;
  }

 protected:
  MatrixBase() {}
  MatrixBase(const MatrixBase&) {}
};

// ------------------------------------------------------------------------------------------------------
// Matrix (owning storage, honouring the storage order in data())
// ------------------------------------------------------------------------------------------------------
namespace internal {
template <class S, int R, int C, int O, int MR, int MC>
struct traits<Matrix<S, R, C, O, MR, MC>> {
  typedef S Scalar;
  enum { RowsAtCompileTime = R, ColsAtCompileTime = C, Options = O };
};
template <class S, int R, int C, bool Fixed = (R != Dynamic && C != Dynamic)>
struct Storage {
  S d[R * C > 0 ? R * C : 1];
  Storage() {
    for (int i = 0; i < R * C; ++i) d[i] = S(0);
  }
  S* data() { return d; }
  const S* data() const { return d; }
  Index rows() const { return R; }
  Index cols() const { return C; }
  void resize(Index r, Index c) {
    if (!(r == R && c == C)) eshim_fail("fixed-size matrix resized");
  }
};
template <class S, int R, int C>
struct Storage<S, R, C, false> {
  std::vector<S> d;
  Index r_, c_;
  Storage() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) {}
  S* data() { return d.data(); }
  const S* data() const { return d.data(); }
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  void resize(Index r, Index c) {
    assert((R == Dynamic || r == R) && (C == Dynamic || c == C));
    if (r * c != r_ * c_) d.assign((size_t)(r * c), S(0));
    r_ = r;
    c_ = c;
  }
};
}  // namespace internal

template <class S, int R, int C, int O, int MR, int MC>
class Matrix : public MatrixBase<Matrix<S, R, C, O, MR, MC>> {
 public:
  typedef MatrixBase<Matrix> Base;
  typedef S Scalar;
  enum { IsRowMajor = (O & RowMajor) ? 1 : 0, Options = O };
  using Base::operator=;
  using Base::operator*=;
  using Base::operator();

  Matrix() {}
  Matrix(const Matrix& o) : Base(), st_(o.st_) {}
  Matrix(Matrix&& o) : Base(), st_(std::move(o.st_)) {}
  Matrix& operator=(const Matrix& o) {
    st_ = o.st_;
    return *this;
  }
  Matrix& operator=(Matrix&& o) {
    st_ = std::move(o.st_);
    return *this;
  }
  // internal: sized construction that can never be mistaken for coefficients
  Matrix(Index r, Index c, internal::SizeTag) { st_.resize(r, c); }

  template <class Other>
  Matrix(const MatrixBase<Other>& o) {
    Base::assign(o);
  }
  template <class S2, int R2, int C2>
  Matrix(const Array<S2, R2, C2>& a);

  // one argument: a size (dynamic vectors) or a data pointer
  template <class T, class = typename std::enable_if<std::is_integral<T>::value>::type>
  explicit Matrix(T n) {
    if (R == Dynamic && C == 1) st_.resize((Index)n, 1);
    else if (R == 1 && C == Dynamic) st_.resize(1, (Index)n);
    else if (R == Dynamic && C == Dynamic) st_.resize((Index)n, (Index)n);
    else if (R * C == 1) st_.data()[0] = (S)n;
  }
  explicit Matrix(const S* p) {
    for (Index i = 0; i < rows() * cols(); ++i) st_.data()[i] = p[i];
  }
  // two arguments: sizes (dynamic) or the coefficients of a fixed 2-vector
  template <class T0, class T1,
            class = typename std::enable_if<internal::is_scalar<T0>::value && internal::is_scalar<T1>::value>::type>
  Matrix(const T0& a, const T1& b) {
    if (R != Dynamic && C != Dynamic && R * C == 2) {
      st_.data()[0] = (S)a;
      st_.data()[1] = (S)b;
    } else {
      st_.resize((Index)a, (Index)b);
    }
  }
  Matrix(const S& a, const S& b, const S& c) {
    static_assert(R * C == 3, "3 coefficients");
    S* d = st_.data();
    d[0] = a, d[1] = b, d[2] = c;
  }
  Matrix(const S& a, const S& b, const S& c, const S& e) {
    static_assert(R * C == 4, "4 coefficients");
    S* d = st_.data();
    d[0] = a, d[1] = b, d[2] = c, d[3] = e;
  }

  Index rows() const { return st_.rows(); }
  Index cols() const { return st_.cols(); }
  S* data() { return st_.data(); }
  const S* data() const { return st_.data(); }
  const S& coeff(Index i, Index j) const {
    assert(i >= 0 && i < rows() && j >= 0 && j < cols());
    return st_.data()[IsRowMajor ? i * cols() + j : j * rows() + i];
  }
  S& coeffRef(Index i, Index j) {
    assert(i >= 0 && i < rows() && j >= 0 && j < cols());
    return st_.data()[IsRowMajor ? i * cols() + j : j * rows() + i];
  }
  using Base::coeff;
  using Base::coeffRef;
  void resize(Index r, Index c) { st_.resize(r, c); }
  void resize(Index n) {
    if (C == 1) st_.resize(n, 1);
    else st_.resize(1, n);
  }
  void resizeLike(Index r, Index c) {
    if (R != Dynamic && C != Dynamic) return;
    if (r == rows() && c == cols()) return;
    if ((R == 1 || C == 1) && (r == 1 || c == 1)) resize(r * c);  // vector <- transposed vector
    else st_.resize(r, c);
  }
  void conservativeResize(Index r, Index c) {
    Matrix t(r, c, internal::SizeTag());
    t.setZero();
    for (Index j = 0; j < std::min(c, cols()); ++j)
      for (Index i = 0; i < std::min(r, rows()); ++i) t.coeffRef(i, j) = coeff(i, j);
    *this = t;
  }
  void conservativeResize(Index n) {
    if (C == 1) conservativeResize(n, 1);
    else conservativeResize(1, n);
  }
  void swap(Matrix& o) { std::swap(st_, o.st_); }
  // 1x1 -> scalar (inner products written as a.transpose()*b)
  template <int RR = R, int CC = C, class = typename std::enable_if<RR == 1 && CC == 1>::type>
  operator S() const {
    return st_.data()[0];
  }
  // ... and to an integral type in one step, as Eigen's (non-template) conversion followed by a standard conversion does
  // (a conversion function TEMPLATE must hit the target type exactly; VioKeyframeWindowMatchingAlgorithm.cpp:332:
  // `const int chi2 = err.transpose() * U.inverse() * err;`)
  template <class T, int RR = R, int CC = C,
            class = typename std::enable_if<RR == 1 && CC == 1 && std::is_integral<T>::value && !std::is_same<T, S>::value>::type>
  operator T() const {
    return (T)st_.data()[0];
  }

 private:
  internal::Storage<S, R, C> st_;
};

// ------------------------------------------------------------------------------------------------------
// Block: a window into another (non-const) expression
// ------------------------------------------------------------------------------------------------------
namespace internal {
template <class Xpr, int BR, int BC>
struct traits<Block<Xpr, BR, BC>> {
  typedef typename traits<Xpr>::Scalar Scalar;
  enum { RowsAtCompileTime = BR, ColsAtCompileTime = BC, Options = 0 };
};
}  // namespace internal
template <class Xpr, int BR, int BC>
class Block : public MatrixBase<Block<Xpr, BR, BC>> {
 public:
  typedef MatrixBase<Block> Base;
  typedef typename internal::traits<Xpr>::Scalar Scalar;
  using Base::operator=;
  using Base::operator*=;
  using Base::operator();
  Block(Xpr& x, Index i0, Index j0, Index r, Index c) : x_(x), i0_(i0), j0_(j0), r_(r), c_(c) {
    assert(i0 >= 0 && j0 >= 0 && i0 + r <= x.rows() && j0 + c <= x.cols());
  }
  Block(const Block& o) : Base(), x_(o.x_), i0_(o.i0_), j0_(o.j0_), r_(o.r_), c_(o.c_) {}
  Block& operator=(const Block& o) { return Base::assign(o); }
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  decltype(auto) coeff(Index i, Index j) const { return const_cast<const Xpr&>(x_).coeff(i0_ + i, j0_ + j); }
  Scalar& coeffRef(Index i, Index j) { return x_.coeffRef(i0_ + i, j0_ + j); }
  using Base::coeff;
  using Base::coeffRef;
  void resizeLike(Index r, Index c) {
    assert((r == r_ && c == c_) || (r * c == r_ * c_ && (r == 1 || c == 1)));
    (void)r;
    (void)c;
  }
  void resize(Index, Index) {}
  void resize(Index) {}

 private:
  Xpr& x_;
  Index i0_, j0_, r_, c_;
};

// ------------------------------------------------------------------------------------------------------
// Map: a view on raw memory with the plain type's storage order
// ------------------------------------------------------------------------------------------------------
namespace internal {
template <class Plain, int MO>
struct traits<Map<Plain, MO>> {
  typedef typename traits<Plain>::Scalar Scalar;
  enum {
    RowsAtCompileTime = traits<Plain>::RowsAtCompileTime,
    ColsAtCompileTime = traits<Plain>::ColsAtCompileTime,
    Options = traits<Plain>::Options
  };
};
}  // namespace internal
template <class Plain, int MO>
class Map : public MatrixBase<Map<Plain, MO>> {
 public:
  typedef MatrixBase<Map> Base;
  typedef typename internal::traits<Plain>::Scalar Scalar;
  typedef typename std::conditional<std::is_const<Plain>::value, const Scalar*, Scalar*>::type Pointer;
  enum {
    R = internal::traits<Plain>::RowsAtCompileTime,
    C = internal::traits<Plain>::ColsAtCompileTime,
    IsRowMajor = (internal::traits<Plain>::Options & RowMajor) ? 1 : 0
  };
  using Base::operator=;
  using Base::operator*=;
  using Base::operator();
  explicit Map(Pointer p) : p_(p), r_(R), c_(C) {}
  Map(Pointer p, Index n) : p_(p), r_(C == 1 ? n : 1), c_(C == 1 ? 1 : n) {}
  Map(Pointer p, Index r, Index c) : p_(p), r_(r), c_(c) {}
  Map(const Map& o) : Base(), p_(o.p_), r_(o.r_), c_(o.c_) {}
  Map& operator=(const Map& o) { return Base::assign(o); }  // copies the coefficients, like Eigen
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  Pointer data() const { return p_; }
  const Scalar& coeff(Index i, Index j) const {
    assert(i >= 0 && i < r_ && j >= 0 && j < c_);
    return p_[IsRowMajor ? i * c_ + j : j * r_ + i];
  }
  Scalar& coeffRef(Index i, Index j) {
    assert(i >= 0 && i < r_ && j >= 0 && j < c_);
    return const_cast<Scalar*>(p_)[IsRowMajor ? i * c_ + j : j * r_ + i];
  }
  using Base::coeff;
  using Base::coeffRef;
  void resizeLike(Index r, Index c) {
    assert((r == r_ && c == c_) || (r * c == r_ * c_ && (r == 1 || c == 1)));
    (void)r;
    (void)c;
  }
  void resize(Index, Index) {}
  void resize(Index) {}

 private:
  Pointer p_;
  Index r_, c_;
};

// ------------------------------------------------------------------------------------------------------
// arithmetic
// ------------------------------------------------------------------------------------------------------
template <class A, class B>
Matrix<typename A::Scalar, internal::pick(A::RowsAtCompileTime, B::RowsAtCompileTime),
       internal::pick(A::ColsAtCompileTime, B::ColsAtCompileTime)>
operator+(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  assert(a.rows() == b.rows() && a.cols() == b.cols());
  Matrix<typename A::Scalar, internal::pick(A::RowsAtCompileTime, B::RowsAtCompileTime),
         internal::pick(A::ColsAtCompileTime, B::ColsAtCompileTime)>
      r(a.rows(), a.cols(), internal::SizeTag());
  for (Index j = 0; j < a.cols(); ++j)
    for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) + b.coeff(i, j);
  return r;
}
template <class A, class B>
Matrix<typename A::Scalar, internal::pick(A::RowsAtCompileTime, B::RowsAtCompileTime),
       internal::pick(A::ColsAtCompileTime, B::ColsAtCompileTime)>
operator-(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  assert(a.rows() == b.rows() && a.cols() == b.cols());
  Matrix<typename A::Scalar, internal::pick(A::RowsAtCompileTime, B::RowsAtCompileTime),
         internal::pick(A::ColsAtCompileTime, B::ColsAtCompileTime)>
      r(a.rows(), a.cols(), internal::SizeTag());
  for (Index j = 0; j < a.cols(); ++j)
    for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) - b.coeff(i, j);
  return r;
}
template <class A>
typename MatrixBase<A>::PlainObject operator-(const MatrixBase<A>& a) {
  typename MatrixBase<A>::PlainObject r(a.rows(), a.cols(), internal::SizeTag());
  for (Index j = 0; j < a.cols(); ++j)
    for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = -a.coeff(i, j);
  return r;
}
template <class A, class B>
Matrix<typename A::Scalar, A::RowsAtCompileTime, B::ColsAtCompileTime> operator*(const MatrixBase<A>& a,
                                                                                 const MatrixBase<B>& b) {
  assert(a.cols() == b.rows());
  Matrix<typename A::Scalar, A::RowsAtCompileTime, B::ColsAtCompileTime> r(a.rows(), b.cols(), internal::SizeTag());
  const Index n = a.cols();
  for (Index j = 0; j < b.cols(); ++j)
    for (Index i = 0; i < a.rows(); ++i) {
      typename A::Scalar s = 0;
      for (Index k = 0; k < n; ++k) s += a.coeff(i, k) * b.coeff(k, j);
      r.coeffRef(i, j) = s;
    }
  return r;
}
template <class A, class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type>
typename MatrixBase<A>::PlainObject operator*(const MatrixBase<A>& a, const T& s) {
  typename MatrixBase<A>::PlainObject r(a.rows(), a.cols(), internal::SizeTag());
  for (Index j = 0; j < a.cols(); ++j)
    for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) * (typename A::Scalar)s;
  return r;
}
template <class A, class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type>
typename MatrixBase<A>::PlainObject operator*(const T& s, const MatrixBase<A>& a) {
  typename MatrixBase<A>::PlainObject r(a.rows(), a.cols(), internal::SizeTag());
  for (Index j = 0; j < a.cols(); ++j)
    for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = (typename A::Scalar)s * a.coeff(i, j);
  return r;
}
template <class A, class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type>
typename MatrixBase<A>::PlainObject operator/(const MatrixBase<A>& a, const T& s) {
  typename MatrixBase<A>::PlainObject r(a.rows(), a.cols(), internal::SizeTag());
  for (Index j = 0; j < a.cols(); ++j)
    for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) / (typename A::Scalar)s;
  return r;
}
// a fixed 1x1 result (inner product written as a^T b) used as a scalar in arithmetic with scalars
#define ESHIM_1X1_OP(OP)                                                                                     \
  template <class S, int O, class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type>  \
  S operator OP(const Matrix<S, 1, 1, O, 1, 1>& a, const T& s) {                                             \
    return a.coeff(0, 0) OP(S) s;                                                                            \
  }                                                                                                          \
  template <class S, int O, class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type>  \
  S operator OP(const T& s, const Matrix<S, 1, 1, O, 1, 1>& a) {                                             \
    return (S)s OP a.coeff(0, 0);                                                                            \
  }
ESHIM_1X1_OP(+)
ESHIM_1X1_OP(-)
#undef ESHIM_1X1_OP
template <class A, class B>
bool operator==(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
  for (Index j = 0; j < a.cols(); ++j)
    for (Index i = 0; i < a.rows(); ++i)
      if (a.coeff(i, j) != b.coeff(i, j)) return false;
  return true;
}
template <class A, class B>
bool operator!=(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  return !(a == b);
}
template <class A>
std::ostream& operator<<(std::ostream& os, const MatrixBase<A>& a) {
  for (Index i = 0; i < a.rows(); ++i) {
    for (Index j = 0; j < a.cols(); ++j) os << (j ? " " : "") << a.coeff(i, j);
    if (i + 1 < a.rows()) os << "\n";
  }
  return os;
}

// ------------------------------------------------------------------------------------------------------
// Array (coefficient-wise world): only what the reference touches
// ------------------------------------------------------------------------------------------------------
template <class S, int R, int C>
class Array {
 public:
  Array() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) { d_.assign((size_t)(r_ * c_), Store()); }
  Array(Index r, Index c) : r_(r), c_(c), d_((size_t)(r * c)) {}
  template <class D>
  explicit Array(const MatrixBase<D>& m) : r_(m.rows()), c_(m.cols()), d_((size_t)(m.rows() * m.cols())) {
    for (Index j = 0; j < c_; ++j)
      for (Index i = 0; i < r_; ++i) (*this)(i, j) = (S)m.coeff(i, j);
  }
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  Index size() const { return r_ * c_; }
  typedef typename std::conditional<std::is_same<S, bool>::value, unsigned char, S>::type Store;  // no vector<bool>
  Store& operator()(Index i, Index j) { return d_[(size_t)(j * r_ + i)]; }
  S operator()(Index i, Index j) const { return (S)d_[(size_t)(j * r_ + i)]; }
  Store& operator()(Index i) { return d_[(size_t)i]; }
  S operator()(Index i) const { return (S)d_[(size_t)i]; }
  Store& operator[](Index i) { return d_[(size_t)i]; }
  S operator[](Index i) const { return (S)d_[(size_t)i]; }
  Matrix<S, R, C> matrix() const {
    Matrix<S, R, C> m(r_, c_, internal::SizeTag());
    for (Index j = 0; j < c_; ++j)
      for (Index i = 0; i < r_; ++i) m.coeffRef(i, j) = (*this)(i, j);
    return m;
  }
  template <class F>
  Array map(F f) const {
    Array a(r_, c_);
    for (size_t k = 0; k < d_.size(); ++k) a.d_[k] = f(d_[k]);
    return a;
  }
  template <class F>
  Array<bool, R, C> test(F f) const {
    Array<bool, R, C> a(r_, c_);
    for (Index k = 0; k < size(); ++k) a[k] = f(d_[(size_t)k]);
    return a;
  }
  Array inverse() const {
    return map([](S v) { return S(1) / v; });
  }
  Array sqrt() const {
    return map([](S v) { return std::sqrt(v); });
  }
  Array abs() const {
    return map([](S v) { return std::fabs(v); });
  }
  Array square() const {
    return map([](S v) { return v * v; });
  }
  S sum() const {
    S s = 0;
    for (size_t k = 0; k < d_.size(); ++k) s += d_[k];
    return s;
  }
  S maxCoeff() const { return *std::max_element(d_.begin(), d_.end()); }
  S minCoeff() const { return *std::min_element(d_.begin(), d_.end()); }
  bool all() const {
    for (size_t k = 0; k < d_.size(); ++k)
      if (!d_[k]) return false;
    return true;
  }
  bool any() const {
    for (size_t k = 0; k < d_.size(); ++k)
      if (d_[k]) return true;
    return false;
  }
  // (cond).select(then, else): `then` a matrix/array of the same shape or a scalar, same for `else`
  template <class T>
  static double pick_(const T& t, Index i, Index j, typename std::enable_if<internal::is_scalar<T>::value>::type* = 0) {
    (void)i;
    (void)j;
    return (double)t;
  }
  template <class T>
  static double pick_(const T& t, Index i, Index j, typename std::enable_if<!internal::is_scalar<T>::value>::type* = 0) {
    return (double)t(i, j);
  }
  template <class T, class E>
  Matrix<double, R, C> select(const T& t, const E& e) const {
    Matrix<double, R, C> m(r_, c_, internal::SizeTag());
    for (Index j = 0; j < c_; ++j)
      for (Index i = 0; i < r_; ++i) m.coeffRef(i, j) = (*this)(i, j) ? pick_(t, i, j) : pick_(e, i, j);
    return m;
  }

 private:
  Index r_, c_;
  std::vector<Store> d_;
};
#define ESHIM_ARRAY_CMP(OP)                                                                        \
  template <class S, int R, int C, class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type> \
  Array<bool, R, C> operator OP(const Array<S, R, C>& a, const T& s) {                             \
    return a.test([s](S v) { return v OP(S) s; });                                                 \
  }
ESHIM_ARRAY_CMP(>)
ESHIM_ARRAY_CMP(<)
ESHIM_ARRAY_CMP(>=)
ESHIM_ARRAY_CMP(<=)
#undef ESHIM_ARRAY_CMP
#define ESHIM_ARRAY_BIN(OP)                                                                        \
  template <class S, int R, int C>                                                                 \
  Array<S, R, C> operator OP(const Array<S, R, C>& a, const Array<S, R, C>& b) {                   \
    Array<S, R, C> r(a.rows(), a.cols());                                                          \
    for (Index k = 0; k < a.size(); ++k) r[k] = a[k] OP b[k];                                      \
    return r;                                                                                      \
  }                                                                                                \
  template <class S, int R, int C, class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type> \
  Array<S, R, C> operator OP(const Array<S, R, C>& a, const T& s) {                                \
    Array<S, R, C> r(a.rows(), a.cols());                                                          \
    for (Index k = 0; k < a.size(); ++k) r[k] = a[k] OP(S) s;                                      \
    return r;                                                                                      \
  }                                                                                                \
  template <class S, int R, int C, class T, class = typename std::enable_if<internal::is_scalar<T>::value>::type> \
  Array<S, R, C> operator OP(const T& s, const Array<S, R, C>& a) {                                \
    Array<S, R, C> r(a.rows(), a.cols());                                                          \
    for (Index k = 0; k < a.size(); ++k) r[k] = (S)s OP a[k];                                      \
    return r;                                                                                      \
  }
ESHIM_ARRAY_BIN(+)
ESHIM_ARRAY_BIN(-)
ESHIM_ARRAY_BIN(*)
ESHIM_ARRAY_BIN(/)
#undef ESHIM_ARRAY_BIN

template <class Derived>
Array<typename MatrixBase<Derived>::Scalar, MatrixBase<Derived>::RowsAtCompileTime,
      MatrixBase<Derived>::ColsAtCompileTime>
MatrixBase<Derived>::array() const {
  return Array<Scalar, RowsAtCompileTime, ColsAtCompileTime>(*this);
}
template <class S, int R, int C, int O, int MR, int MC>
template <class S2, int R2, int C2>
Matrix<S, R, C, O, MR, MC>::Matrix(const Array<S2, R2, C2>& a) {
  st_.resize(a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); ++j)
    for (Index i = 0; i < a.rows(); ++i) coeffRef(i, j) = (S)a(i, j);
}

// ------------------------------------------------------------------------------------------------------
// inverse / determinant
// ------------------------------------------------------------------------------------------------------
template <class Derived>
typename MatrixBase<Derived>::Scalar MatrixBase<Derived>::determinant() const {
  const Index n = rows();
  assert(n == cols());
  if (n == 1) return coeff(0, 0);
  if (n == 2) return coeff(0, 0) * coeff(1, 1) - coeff(0, 1) * coeff(1, 0);
  if (n == 3)
    return coeff(0, 0) * (coeff(1, 1) * coeff(2, 2) - coeff(1, 2) * coeff(2, 1)) -
           coeff(0, 1) * (coeff(1, 0) * coeff(2, 2) - coeff(1, 2) * coeff(2, 0)) +
           coeff(0, 2) * (coeff(1, 0) * coeff(2, 1) - coeff(1, 1) * coeff(2, 0));
  Matrix<Scalar, Dynamic, Dynamic> a(derived());
  Scalar det = 1;
  for (Index k = 0; k < n; ++k) {
    Index p = k;
    for (Index i = k + 1; i < n; ++i)
      if (std::fabs(a(i, k)) > std::fabs(a(p, k))) p = i;
    if (a(p, k) == Scalar(0)) return Scalar(0);
    if (p != k) {
      for (Index j = 0; j < n; ++j) std::swap(a(k, j), a(p, j));
      det = -det;
    }
    det *= a(k, k);
    for (Index i = k + 1; i < n; ++i) {
      const Scalar f = a(i, k) / a(k, k);
      for (Index j = k + 1; j < n; ++j) a(i, j) -= f * a(k, j);
    }
  }
  return det;
}
template <class Derived>
typename MatrixBase<Derived>::PlainObject MatrixBase<Derived>::inverse() const {
  const Index n = rows();
  assert(n == cols());
  PlainObject r(n, n, internal::SizeTag());
  if (n == 1) {
    r(0, 0) = Scalar(1) / coeff(0, 0);
    return r;
  }
  if (n == 2) {
    const Scalar idet = Scalar(1) / determinant();
    r(0, 0) = coeff(1, 1) * idet;
    r(1, 0) = -coeff(1, 0) * idet;
    r(0, 1) = -coeff(0, 1) * idet;
    r(1, 1) = coeff(0, 0) * idet;
    return r;
  }
  if (n == 3) {
    const Scalar c00 = coeff(1, 1) * coeff(2, 2) - coeff(1, 2) * coeff(2, 1);
    const Scalar c10 = coeff(1, 2) * coeff(2, 0) - coeff(1, 0) * coeff(2, 2);
    const Scalar c20 = coeff(1, 0) * coeff(2, 1) - coeff(1, 1) * coeff(2, 0);
    const Scalar idet = Scalar(1) / (coeff(0, 0) * c00 + coeff(0, 1) * c10 + coeff(0, 2) * c20);
    r(0, 0) = c00 * idet;
    r(1, 0) = c10 * idet;
    r(2, 0) = c20 * idet;
    r(0, 1) = (coeff(0, 2) * coeff(2, 1) - coeff(0, 1) * coeff(2, 2)) * idet;
    r(1, 1) = (coeff(0, 0) * coeff(2, 2) - coeff(0, 2) * coeff(2, 0)) * idet;
    r(2, 1) = (coeff(0, 1) * coeff(2, 0) - coeff(0, 0) * coeff(2, 1)) * idet;
    r(0, 2) = (coeff(0, 1) * coeff(1, 2) - coeff(0, 2) * coeff(1, 1)) * idet;
    r(1, 2) = (coeff(0, 2) * coeff(1, 0) - coeff(0, 0) * coeff(1, 2)) * idet;
    r(2, 2) = (coeff(0, 0) * coeff(1, 1) - coeff(0, 1) * coeff(1, 0)) * idet;
    return r;
  }
  // partial-pivot LU, then solve for the columns of the identity
  Matrix<Scalar, Dynamic, Dynamic> a(derived());
  std::vector<Index> perm((size_t)n);
  for (Index i = 0; i < n; ++i) perm[(size_t)i] = i;
  for (Index k = 0; k < n; ++k) {
    Index p = k;
    for (Index i = k + 1; i < n; ++i)
      if (std::fabs(a(i, k)) > std::fabs(a(p, k))) p = i;
    if (p != k) {
      for (Index j = 0; j < n; ++j) std::swap(a(k, j), a(p, j));
      std::swap(perm[(size_t)k], perm[(size_t)p]);
    }
    for (Index i = k + 1; i < n; ++i) {
      a(i, k) /= a(k, k);
      for (Index j = k + 1; j < n; ++j) a(i, j) -= a(i, k) * a(k, j);
    }
  }
  for (Index c = 0; c < n; ++c) {
    std::vector<Scalar> y((size_t)n);
    for (Index i = 0; i < n; ++i) {
      Scalar s = (perm[(size_t)i] == c) ? Scalar(1) : Scalar(0);
      for (Index k = 0; k < i; ++k) s -= a(i, k) * y[(size_t)k];
      y[(size_t)i] = s;
    }
    for (Index i = n - 1; i >= 0; --i) {
      Scalar s = y[(size_t)i];
      for (Index k = i + 1; k < n; ++k) s -= a(i, k) * r(k, c);
      r(i, c) = s / a(i, i);
    }
  }
  return r;
}

// ------------------------------------------------------------------------------------------------------
// LLT
// ------------------------------------------------------------------------------------------------------
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };
template <class M>
class LLT {
 public:
  typedef typename M::Scalar Scalar;
  LLT() : info_(InvalidInput) {}
  template <class D>
  explicit LLT(const MatrixBase<D>& a) {
    compute(a);
  }
  template <class D>
  LLT& compute(const MatrixBase<D>& a) {
    m_ = a;
    const Index n = m_.rows();
    info_ = Success;
    for (Index k = 0; k < n; ++k) {
      Scalar x = m_(k, k);
      for (Index j = 0; j < k; ++j) x -= m_(k, j) * m_(k, j);
      if (x <= Scalar(0)) {  // stop here and leave the rest untouched
        info_ = NumericalIssue;
        break;
      }
      x = std::sqrt(x);
      m_(k, k) = x;
      for (Index i = k + 1; i < n; ++i) {
        Scalar s = m_(i, k);
        for (Index j = 0; j < k; ++j) s -= m_(i, j) * m_(k, j);
        m_(i, k) = s / x;
      }
    }
    return *this;
  }
  // the lower triangle of the working matrix (strict upper part read as zero), like TriangularView<Lower>
  M matrixL() const {
    M l(m_);
    for (Index j = 0; j < l.cols(); ++j)
      for (Index i = 0; i < j; ++i) l(i, j) = Scalar(0);
    return l;
  }
  typename M::TransposeReturnType matrixU() const { return matrixL().transpose(); }
  const M& matrixLLT() const { return m_; }
  ComputationInfo info() const { return info_; }
  template <class D>
  typename MatrixBase<D>::PlainObject solve(const MatrixBase<D>& b) const {
    typename MatrixBase<D>::PlainObject x(b);
    const Index n = m_.rows();
    for (Index c = 0; c < x.cols(); ++c) {
      for (Index i = 0; i < n; ++i) {
        Scalar s = x(i, c);
        for (Index k = 0; k < i; ++k) s -= m_(i, k) * x(k, c);
        x(i, c) = s / m_(i, i);
      }
      for (Index i = n - 1; i >= 0; --i) {
        Scalar s = x(i, c);
        for (Index k = i + 1; k < n; ++k) s -= m_(k, i) * x(k, c);
        x(i, c) = s / m_(i, i);
      }
    }
    return x;
  }

 private:
  M m_;
  ComputationInfo info_;
};
template <class Derived>
LLT<typename MatrixBase<Derived>::PlainObject> MatrixBase<Derived>::llt() const {
  return LLT<PlainObject>(*this);
}

// a 1x1 result compared with a scalar (`v.transpose() * v < 4.0`): the conversion operator is a template and is not
// found for the built-in comparison
template <class S, int O, int MR, int MC>
inline bool operator<(const Matrix<S, 1, 1, O, MR, MC>& a, const S& b) { return a.coeff(0, 0) < b; }
template <class S, int O, int MR, int MC>
inline bool operator>(const Matrix<S, 1, 1, O, MR, MC>& a, const S& b) { return a.coeff(0, 0) > b; }
template <class S, int O, int MR, int MC>
inline bool operator<(const S& a, const Matrix<S, 1, 1, O, MR, MC>& b) { return a < b.coeff(0, 0); }
template <class S, int O, int MR, int MC>
inline bool operator>(const S& a, const Matrix<S, 1, 1, O, MR, MC>& b) { return a > b.coeff(0, 0); }

// ------------------------------------------------------------------------------------------------------
// ColPivHouseholderQR: A P = Q R.  Only what the reference asks of it: rank() (and the R factor for inspection).
// The algorithm is the published one (Golub & Van Loan 5.4.2 with Eigen's thresholds): at step k the remaining column
// of largest norm is swapped in (norms are down-dated after every reflection; the winner's norm is recomputed from its
// entries), a column whose squared norm falls below eps^2 * max column norm^2 * (rows - k) / rows ends the list of non-zero
// pivots, and rank() counts the pivots |R_ii| > eps * min(rows, cols) * max_i |R_ii|.
// ------------------------------------------------------------------------------------------------------
template <class M>
class ColPivHouseholderQR {
 public:
  typedef typename M::Scalar Scalar;
  template <class D>
  explicit ColPivHouseholderQR(const MatrixBase<D>& a) : qr_(a) {
    const Index rows = qr_.rows(), cols = qr_.cols(), size = std::min(rows, cols);
    std::vector<Scalar> cn((size_t)cols);
    Scalar maxn = 0;
    for (Index c = 0; c < cols; ++c) {
      Scalar s = 0;
      for (Index r = 0; r < rows; ++r) s += qr_(r, c) * qr_(r, c);
      cn[(size_t)c] = s;
      maxn = std::max(maxn, s);
    }
    const Scalar eps = std::numeric_limits<Scalar>::epsilon();
    const Scalar threshold_helper = maxn * eps * eps / Scalar(rows);
    nonzero_ = size;
    maxpivot_ = 0;
    for (Index k = 0; k < size; ++k) {
      Index big = k;
      for (Index c = k + 1; c < cols; ++c)
        if (cn[(size_t)c] > cn[(size_t)big]) big = c;
      Scalar bn = 0;
      for (Index r = k; r < rows; ++r) bn += qr_(r, big) * qr_(r, big);
      cn[(size_t)big] = bn;
      if (nonzero_ == size && bn < threshold_helper * Scalar(rows - k)) nonzero_ = k;
      if (big != k) {
        for (Index r = 0; r < rows; ++r) std::swap(qr_(r, k), qr_(r, big));
        std::swap(cn[(size_t)k], cn[(size_t)big]);
      }
      const Scalar c0 = qr_(k, k);
      Scalar tail = 0;
      for (Index r = k + 1; r < rows; ++r) tail += qr_(r, k) * qr_(r, k);
      Scalar beta, tau;
      if (tail <= std::numeric_limits<Scalar>::min()) {
        tau = 0, beta = c0;
        for (Index r = k + 1; r < rows; ++r) qr_(r, k) = 0;
      } else {
        beta = std::sqrt(c0 * c0 + tail);
        if (c0 >= 0) beta = -beta;
        for (Index r = k + 1; r < rows; ++r) qr_(r, k) /= (c0 - beta);
        tau = (beta - c0) / beta;
      }
      qr_(k, k) = beta;
      maxpivot_ = std::max(maxpivot_, std::fabs(beta));
      for (Index c = k + 1; c < cols; ++c) {
        Scalar t = qr_(k, c);
        for (Index r = k + 1; r < rows; ++r) t += qr_(r, k) * qr_(r, c);
        t *= tau;
        qr_(k, c) -= t;
        for (Index r = k + 1; r < rows; ++r) qr_(r, c) -= t * qr_(r, k);
      }
      for (Index c = k + 1; c < cols; ++c) cn[(size_t)c] -= qr_(k, c) * qr_(k, c);
    }
  }
  Index rank() const {
    const Index size = std::min(qr_.rows(), qr_.cols());
    const Scalar thr = maxpivot_ * (std::numeric_limits<Scalar>::epsilon() * Scalar(size));
    Index r = 0;
    for (Index i = 0; i < nonzero_; ++i) r += (std::fabs(qr_(i, i)) > thr) ? 1 : 0;
    return r;
  }
  bool isInvertible() const { return rank() == qr_.rows() && qr_.rows() == qr_.cols(); }
  const M& matrixQR() const { return qr_; }

 private:
  M qr_;
  Index nonzero_;
  Scalar maxpivot_;
};
template <class Derived>
ColPivHouseholderQR<typename MatrixBase<Derived>::PlainObject> MatrixBase<Derived>::colPivHouseholderQr() const {
  return ColPivHouseholderQR<PlainObject>(*this);
}

// ------------------------------------------------------------------------------------------------------
// SelfAdjointEigenSolver (cyclic Jacobi; eigenvalues ascending, eigenvectors in the columns)
// ------------------------------------------------------------------------------------------------------
template <class M>
class SelfAdjointEigenSolver {
 public:
  typedef typename M::Scalar Scalar;
  typedef Matrix<Scalar, M::RowsAtCompileTime, 1> RealVectorType;
  SelfAdjointEigenSolver() : info_(InvalidInput) {}
  template <class D>
  explicit SelfAdjointEigenSolver(const MatrixBase<D>& a, int /*options*/ = 0) {
    compute(a);
  }
  template <class D>
  SelfAdjointEigenSolver& compute(const MatrixBase<D>& a_in, int /*options*/ = 0) {
    const Index n = a_in.rows();
    Matrix<Scalar, Dynamic, Dynamic> a(n, n, internal::SizeTag()), v(n, n, internal::SizeTag());
    for (Index j = 0; j < n; ++j)
      for (Index i = 0; i < n; ++i) a(i, j) = (i >= j) ? a_in.coeff(i, j) : a_in.coeff(j, i);  // lower triangle
    v.setIdentity();
    for (int sweep = 0; sweep < 100; ++sweep) {
      Scalar off = 0, diag = 0;
      for (Index i = 0; i < n; ++i) {
        diag += a(i, i) * a(i, i);
        for (Index j = 0; j < i; ++j) off += 2 * a(i, j) * a(i, j);
      }
      if (off <= Scalar(1e-40) * diag || off == Scalar(0)) break;
      for (Index p = 0; p < n - 1; ++p)
        for (Index q = p + 1; q < n; ++q) {
          const Scalar apq = a(p, q);
          if (apq == Scalar(0)) continue;
          const Scalar theta = (a(q, q) - a(p, p)) / (2 * apq);
          const Scalar t = (theta >= 0 ? Scalar(1) : Scalar(-1)) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
          const Scalar c = Scalar(1) / std::sqrt(t * t + 1), s = t * c;
          for (Index k = 0; k < n; ++k) {
            const Scalar akp = a(k, p), akq = a(k, q);
            a(k, p) = c * akp - s * akq;
            a(k, q) = s * akp + c * akq;
          }
          for (Index k = 0; k < n; ++k) {
            const Scalar apk = a(p, k), aqk = a(q, k);
            a(p, k) = c * apk - s * aqk;
            a(q, k) = s * apk + c * aqk;
          }
          for (Index k = 0; k < n; ++k) {
            const Scalar vkp = v(k, p), vkq = v(k, q);
            v(k, p) = c * vkp - s * vkq;
            v(k, q) = s * vkp + c * vkq;
          }
        }
    }
    std::vector<Index> order((size_t)n);
    for (Index i = 0; i < n; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](Index x, Index y) { return a(x, x) < a(y, y); });
    val_.resizeLike(n, 1);
    vec_.resizeLike(n, n);
    for (Index j = 0; j < n; ++j) {
      val_(j) = a(order[(size_t)j], order[(size_t)j]);
      for (Index i = 0; i < n; ++i) vec_(i, j) = v(i, order[(size_t)j]);
    }
    info_ = Success;
    return *this;
  }
  template <class D>
  SelfAdjointEigenSolver& computeDirect(const MatrixBase<D>& a, int o = 0) {
    return compute(a, o);
  }
  const RealVectorType& eigenvalues() const { return val_; }
  const M& eigenvectors() const { return vec_; }
  ComputationInfo info() const { return info_; }

 private:
  RealVectorType val_;
  M vec_;
  ComputationInfo info_;
};
enum { ComputeEigenvectors = 0x80, EigenvaluesOnly = 0x40 };

// ------------------------------------------------------------------------------------------------------
// Quaternion (coefficients stored x, y, z, w), Map<Quaternion>, AngleAxis
// ------------------------------------------------------------------------------------------------------
template <class Derived, class S>
class QuaternionBase {
 public:
  struct QuaternionTag {};
  typedef S Scalar;
  typedef Matrix<S, 3, 1> Vector3;
  typedef Matrix<S, 3, 3> Matrix3;
  Derived& derived() { return *static_cast<Derived*>(this); }
  const Derived& derived() const { return *static_cast<const Derived*>(this); }
  S x() const { return derived().coeffs().coeff(0, 0); }
  S y() const { return derived().coeffs().coeff(1, 0); }
  S z() const { return derived().coeffs().coeff(2, 0); }
  S w() const { return derived().coeffs().coeff(3, 0); }
  S& x() { return derived().coeffs().coeffRef(0, 0); }
  S& y() { return derived().coeffs().coeffRef(1, 0); }
  S& z() { return derived().coeffs().coeffRef(2, 0); }
  S& w() { return derived().coeffs().coeffRef(3, 0); }
  Vector3 vec() const { return Vector3(x(), y(), z()); }
  auto vec() { return derived().coeffs().template head<3>(); }
  S squaredNorm() const { return x() * x() + y() * y() + z() * z() + w() * w(); }
  S norm() const { return std::sqrt(squaredNorm()); }
  void normalize() {
    const S n = norm();
    x() /= n, y() /= n, z() /= n, w() /= n;
  }
  Quaternion<S> normalized() const {
    const S n = norm();
    return Quaternion<S>(w() / n, x() / n, y() / n, z() / n);
  }
  Quaternion<S> conjugate() const { return Quaternion<S>(w(), -x(), -y(), -z()); }
  Quaternion<S> inverse() const {
    const S n2 = squaredNorm();
    if (n2 > S(0)) return Quaternion<S>(w() / n2, -x() / n2, -y() / n2, -z() / n2);
    return Quaternion<S>(S(0), S(0), S(0), S(0));
  }
  Derived& setIdentity() {
    x() = y() = z() = S(0);
    w() = S(1);
    return derived();
  }
  template <class OD>
  S dot(const QuaternionBase<OD, S>& o) const {
    return x() * o.x() + y() * o.y() + z() * o.z() + w() * o.w();
  }
  template <class OD>
  S angularDistance(const QuaternionBase<OD, S>& o) const {
    Quaternion<S> d = (*this) * o.conjugate();
    return S(2) * std::atan2(d.vec().norm(), std::fabs(d.w()));
  }
  template <class OD>
  Quaternion<S> operator*(const QuaternionBase<OD, S>& b) const {
    const QuaternionBase& a = *this;
    return Quaternion<S>(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                         a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                         a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                         a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  template <class OD>
  Derived& operator*=(const QuaternionBase<OD, S>& b) {
    return derived() = (*this) * b;
  }
  Matrix3 toRotationMatrix() const {
    Matrix3 r;
    const S tx = S(2) * x(), ty = S(2) * y(), tz = S(2) * z();
    const S twx = tx * w(), twy = ty * w(), twz = tz * w();
    const S txx = tx * x(), txy = ty * x(), txz = tz * x();
    const S tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
    r(0, 0) = S(1) - (tyy + tzz);
    r(0, 1) = txy - twz;
    r(0, 2) = txz + twy;
    r(1, 0) = txy + twz;
    r(1, 1) = S(1) - (txx + tzz);
    r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy;
    r(2, 1) = tyz + twx;
    r(2, 2) = S(1) - (txx + tyy);
    return r;
  }
  Matrix3 matrix() const { return toRotationMatrix(); }
  template <class D>
  Vector3 _transformVector(const MatrixBase<D>& v) const {
    // v + 2w (u x v) + 2 u x (u x v)
    const Vector3 u = vec();
    Vector3 uv = u.cross(v);
    uv = uv + uv;
    return Vector3(v) + w() * uv + u.cross(uv);
  }
  template <class D>
  Vector3 operator*(const MatrixBase<D>& v) const {
    return _transformVector(v);
  }
  // rotation matrix -> quaternion (trace method)
  template <class D>
  void fromRotationMatrix(const MatrixBase<D>& m) {
    S t = m.coeff(0, 0) + m.coeff(1, 1) + m.coeff(2, 2);
    if (t > S(0)) {
      t = std::sqrt(t + S(1));
      w() = S(0.5) * t;
      t = S(0.5) / t;
      x() = (m.coeff(2, 1) - m.coeff(1, 2)) * t;
      y() = (m.coeff(0, 2) - m.coeff(2, 0)) * t;
      z() = (m.coeff(1, 0) - m.coeff(0, 1)) * t;
    } else {
      Index i = 0;
      if (m.coeff(1, 1) > m.coeff(0, 0)) i = 1;
      if (m.coeff(2, 2) > m.coeff(i, i)) i = 2;
      const Index j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(m.coeff(i, i) - m.coeff(j, j) - m.coeff(k, k) + S(1));
      S q[3];
      q[i] = S(0.5) * t;
      t = S(0.5) / t;
      w() = (m.coeff(k, j) - m.coeff(j, k)) * t;
      q[j] = (m.coeff(j, i) + m.coeff(i, j)) * t;
      q[k] = (m.coeff(k, i) + m.coeff(i, k)) * t;
      x() = q[0], y() = q[1], z() = q[2];
    }
  }
  template <class OD>
  bool isApprox(const QuaternionBase<OD, S>& o, S prec = S(1e-12)) const {
    return derived().coeffs().isApprox(o.derived().coeffs(), prec);
  }
  template <class OD>
  Quaternion<S> slerp(S t, const QuaternionBase<OD, S>& o) const {
    S d = dot(o), ad = std::fabs(d), s0, s1;
    if (ad >= S(1) - std::numeric_limits<S>::epsilon()) {
      s0 = S(1) - t;
      s1 = t;
    } else {
      const S th = std::acos(ad), st = std::sin(th);
      s0 = std::sin((S(1) - t) * th) / st;
      s1 = std::sin(t * th) / st;
    }
    if (d < S(0)) s1 = -s1;
    return Quaternion<S>(s0 * w() + s1 * o.w(), s0 * x() + s1 * o.x(), s0 * y() + s1 * o.y(),
                         s0 * z() + s1 * o.z());
  }
};

template <class S>
class Quaternion : public QuaternionBase<Quaternion<S>, S> {
 public:
  typedef QuaternionBase<Quaternion<S>, S> Base;
  typedef Matrix<S, 4, 1> Coefficients;
  Quaternion() {}
  Quaternion(const S& w, const S& x, const S& y, const S& z) : c_(x, y, z, w) {}
  explicit Quaternion(const S* p) : c_(p) {}
  Quaternion(const Quaternion& o) : Base(), c_(o.c_) {}
  template <class OD>
  Quaternion(const QuaternionBase<OD, S>& o) : c_(o.x(), o.y(), o.z(), o.w()) {}
  Quaternion(const AngleAxis<S>& aa) { *this = aa; }
  // a 4-vector of coefficients (x,y,z,w) or a 3x3 rotation matrix
  template <class D>
  explicit Quaternion(const MatrixBase<D>& m) {
    *this = m;
  }
  Quaternion& operator=(const Quaternion& o) {
    c_ = o.c_;
    return *this;
  }
  template <class OD>
  Quaternion& operator=(const QuaternionBase<OD, S>& o) {
    c_ = Coefficients(o.x(), o.y(), o.z(), o.w());
    return *this;
  }
  Quaternion& operator=(const AngleAxis<S>& aa);
  template <class D>
  Quaternion& operator=(const MatrixBase<D>& m) {
    if (m.rows() == 3 && m.cols() == 3) this->fromRotationMatrix(m);
    else {
      assert(m.size() == 4);
      for (Index i = 0; i < 4; ++i) c_[i] = m.coeff(i);
    }
    return *this;
  }
  Coefficients& coeffs() { return c_; }
  const Coefficients& coeffs() const { return c_; }
  static Quaternion Identity() { return Quaternion(S(1), S(0), S(0), S(0)); }
  template <class D1, class D2>
  static Quaternion FromTwoVectors(const MatrixBase<D1>& a, const MatrixBase<D2>& b) {
    Matrix<S, 3, 1> v0 = a.normalized(), v1 = b.normalized();
    const S c = v1.dot(v0);
    Matrix<S, 3, 1> axis = v0.cross(v1);
    const S s = std::sqrt((S(1) + c) * S(2));
    return Quaternion(s * S(0.5), axis[0] / s, axis[1] / s, axis[2] / s);
  }
  Quaternion& setFromTwoVectors(const Matrix<S, 3, 1>& a, const Matrix<S, 3, 1>& b) {
    return *this = FromTwoVectors(a, b);
  }

 private:
  Coefficients c_;
};

template <class S, int MO>
class Map<Quaternion<S>, MO> : public QuaternionBase<Map<Quaternion<S>, MO>, S> {
 public:
  typedef Map<Matrix<S, 4, 1>> Coefficients;
  explicit Map(S* p) : c_(p) {}
  Map(const Map& o) : QuaternionBase<Map, S>(), c_(o.c_.data()) {}
  Map& operator=(const Map& o) {
    c_ = o.c_;
    return *this;
  }
  template <class OD>
  Map& operator=(const QuaternionBase<OD, S>& o) {
    const S x = o.x(), y = o.y(), z = o.z(), w = o.w();
    c_[0] = x, c_[1] = y, c_[2] = z, c_[3] = w;
    return *this;
  }
  Map& operator=(const AngleAxis<S>& aa) { return *this = Quaternion<S>(aa); }
  template <class D>
  Map& operator=(const MatrixBase<D>& m) {
    return *this = Quaternion<S>(m);
  }
  Coefficients& coeffs() { return c_; }
  const Coefficients& coeffs() const { return c_; }

 private:
  Coefficients c_;
};
template <class S, int MO>
class Map<const Quaternion<S>, MO> : public QuaternionBase<Map<const Quaternion<S>, MO>, S> {
 public:
  typedef Map<const Matrix<S, 4, 1>> Coefficients;
  explicit Map(const S* p) : c_(p) {}
  Map(const Map& o) : QuaternionBase<Map, S>(), c_(o.c_.data()) {}
  const Coefficients& coeffs() const { return c_; }

 private:
  Coefficients c_;
};

template <class S>
class AngleAxis {
 public:
  typedef Matrix<S, 3, 1> Vector3;
  AngleAxis() : angle_(0), axis_(S(1), S(0), S(0)) {}
  template <class D>
  AngleAxis(const S& angle, const MatrixBase<D>& axis) : angle_(angle), axis_(axis) {}
  template <class QD>
  explicit AngleAxis(const QuaternionBase<QD, S>& q) {
    S n = q.vec().norm();
    if (n < std::numeric_limits<S>::epsilon()) n = q.vec().stableNorm();
    if (n != S(0)) {
      angle_ = S(2) * std::atan2(n, std::fabs(q.w()));
      if (q.w() < S(0)) n = -n;
      axis_ = q.vec() / n;
    } else {
      angle_ = S(0);
      axis_ = Vector3(S(1), S(0), S(0));
    }
  }
  S angle() const { return angle_; }
  S& angle() { return angle_; }
  const Vector3& axis() const { return axis_; }
  Vector3& axis() { return axis_; }
  Matrix<S, 3, 3> toRotationMatrix() const { return Quaternion<S>(*this).toRotationMatrix(); }
  Matrix<S, 3, 3> matrix() const { return toRotationMatrix(); }
  template <class D>
  Vector3 operator*(const MatrixBase<D>& v) const {
    return Quaternion<S>(*this) * v;
  }

 private:
  S angle_;
  Vector3 axis_;
};
template <class S>
Quaternion<S>& Quaternion<S>::operator=(const AngleAxis<S>& aa) {
  const S ha = S(0.5) * aa.angle();
  const S s = std::sin(ha);
  c_ = Coefficients(s * aa.axis()[0], s * aa.axis()[1], s * aa.axis()[2], std::cos(ha));
  return *this;
}

// ------------------------------------------------------------------------------------------------------
// typedefs
// ------------------------------------------------------------------------------------------------------
#define ESHIM_TYPEDEFS(T, SUF)                            \
  typedef Matrix<T, 2, 2> Matrix2##SUF;                   \
  typedef Matrix<T, 3, 3> Matrix3##SUF;                   \
  typedef Matrix<T, 4, 4> Matrix4##SUF;                   \
  typedef Matrix<T, Dynamic, Dynamic> MatrixX##SUF;       \
  typedef Matrix<T, 2, Dynamic> Matrix2X##SUF;            \
  typedef Matrix<T, 3, Dynamic> Matrix3X##SUF;            \
  typedef Matrix<T, 4, Dynamic> Matrix4X##SUF;            \
  typedef Matrix<T, Dynamic, 2> MatrixX2##SUF;            \
  typedef Matrix<T, Dynamic, 3> MatrixX3##SUF;            \
  typedef Matrix<T, 2, 1> Vector2##SUF;                   \
  typedef Matrix<T, 3, 1> Vector3##SUF;                   \
  typedef Matrix<T, 4, 1> Vector4##SUF;                   \
  typedef Matrix<T, Dynamic, 1> VectorX##SUF;             \
  typedef Matrix<T, 1, 2> RowVector2##SUF;                \
  typedef Matrix<T, 1, 3> RowVector3##SUF;                \
  typedef Matrix<T, 1, 4> RowVector4##SUF;                \
  typedef Matrix<T, 1, Dynamic> RowVectorX##SUF;
ESHIM_TYPEDEFS(double, d)
ESHIM_TYPEDEFS(float, f)
ESHIM_TYPEDEFS(int, i)
#undef ESHIM_TYPEDEFS
typedef Quaternion<double> Quaterniond;
typedef Quaternion<float> Quaternionf;
typedef AngleAxis<double> AngleAxisd;
typedef AngleAxis<float> AngleAxisf;

}  // namespace Eigen
