// eigen_lite.hpp -- the sliver of Eigen's API that the kept sdf_tools classes need, for builds
// where Eigen3 is not installed (this image, the GPU box).  When <Eigen/Geometry> exists the real
// library is used instead and this file is inert.  Only what VoxelGrid / SignedDistanceField /
// CollisionMapGrid touch is provided: fixed-size double vectors, a 4x4 matrix, a quaternion
// good enough for rotating a vector, and a rigid transform (Isometry3d).
#pragma once
#if __has_include(<Eigen/Geometry>)
#include <Eigen/Geometry>
#define SDF_TOOLS_HAVE_EIGEN 1
#else
#define SDF_TOOLS_HAVE_EIGEN 0
#include <array>
#include <cmath>
#include <cstddef>

namespace Eigen {

template <int N>
class VecNd {
public:
    VecNd() { v_.fill(0.0); }
    VecNd(double a, double b, double c) { static_assert(N == 3, "3 values"); v_ = {a, b, c}; }
    VecNd(double a, double b, double c, double d) { static_assert(N == 4, "4 values"); v_ = {a, b, c, d}; }
    double& operator()(int i) { return v_[(size_t)i]; }
    double operator()(int i) const { return v_[(size_t)i]; }
    double& operator[](int i) { return v_[(size_t)i]; }
    double operator[](int i) const { return v_[(size_t)i]; }
    double x() const { return v_[0]; }
    double y() const { return v_[1]; }
    double z() const { return v_[2]; }
    VecNd operator+(const VecNd& o) const { VecNd r; for (int i = 0; i < N; i++) r.v_[i] = v_[i] + o.v_[i]; return r; }
    VecNd operator-(const VecNd& o) const { VecNd r; for (int i = 0; i < N; i++) r.v_[i] = v_[i] - o.v_[i]; return r; }
    VecNd operator*(double s) const { VecNd r; for (int i = 0; i < N; i++) r.v_[i] = v_[i] * s; return r; }
    double dot(const VecNd& o) const { double s = 0; for (int i = 0; i < N; i++) s += v_[i] * o.v_[i]; return s; }
    double norm() const { return std::sqrt(dot(*this)); }
    const double* data() const { return v_.data(); }
private:
    std::array<double, N> v_;
};
using Vector3d = VecNd<3>;
using Vector4d = VecNd<4>;

// Column-major 4x4, like Eigen's default storage (matters for serialisation).
class Matrix4d {
public:
    Matrix4d() { m_.fill(0.0); }
    static Matrix4d Identity() { Matrix4d r; for (int i = 0; i < 4; i++) r(i, i) = 1.0; return r; }
    double& operator()(int r, int c) { return m_[(size_t)(c * 4 + r)]; }
    double operator()(int r, int c) const { return m_[(size_t)(c * 4 + r)]; }
    const double* data() const { return m_.data(); }
    double* data() { return m_.data(); }
    Matrix4d operator*(const Matrix4d& o) const {
        Matrix4d r;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += (*this)(i, k) * o(k, j); r(i, j) = s; }
        return r;
    }
private:
    std::array<double, 16> m_;
};

class Matrix3d {
public:
    Matrix3d() { m_.fill(0.0); }
    double& operator()(int r, int c) { return m_[(size_t)(c * 3 + r)]; }
    double operator()(int r, int c) const { return m_[(size_t)(c * 3 + r)]; }
    Vector3d operator*(const Vector3d& v) const {
        return Vector3d((*this)(0, 0) * v(0) + (*this)(0, 1) * v(1) + (*this)(0, 2) * v(2),
                        (*this)(1, 0) * v(0) + (*this)(1, 1) * v(1) + (*this)(1, 2) * v(2),
                        (*this)(2, 0) * v(0) + (*this)(2, 1) * v(1) + (*this)(2, 2) * v(2));
    }
private:
    std::array<double, 9> m_;
};

class Quaterniond {
public:
    Quaterniond() : w_(1), x_(0), y_(0), z_(0) {}
    Quaterniond(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}
    explicit Quaterniond(const Matrix3d& R) {      // Shepperd's method
        const double t = R(0, 0) + R(1, 1) + R(2, 2);
        if (t > 0) { double s = std::sqrt(t + 1.0) * 2; w_ = 0.25 * s; x_ = (R(2, 1) - R(1, 2)) / s; y_ = (R(0, 2) - R(2, 0)) / s; z_ = (R(1, 0) - R(0, 1)) / s; }
        else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) { double s = std::sqrt(1.0 + R(0, 0) - R(1, 1) - R(2, 2)) * 2; w_ = (R(2, 1) - R(1, 2)) / s; x_ = 0.25 * s; y_ = (R(0, 1) + R(1, 0)) / s; z_ = (R(0, 2) + R(2, 0)) / s; }
        else if (R(1, 1) > R(2, 2)) { double s = std::sqrt(1.0 + R(1, 1) - R(0, 0) - R(2, 2)) * 2; w_ = (R(0, 2) - R(2, 0)) / s; x_ = (R(0, 1) + R(1, 0)) / s; y_ = 0.25 * s; z_ = (R(1, 2) + R(2, 1)) / s; }
        else { double s = std::sqrt(1.0 + R(2, 2) - R(0, 0) - R(1, 1)) * 2; w_ = (R(1, 0) - R(0, 1)) / s; x_ = (R(0, 2) + R(2, 0)) / s; y_ = (R(1, 2) + R(2, 1)) / s; z_ = 0.25 * s; }
    }
    double w() const { return w_; } double x() const { return x_; } double y() const { return y_; } double z() const { return z_; }
    Quaterniond operator*(const Quaterniond& o) const {
        return Quaterniond(w_ * o.w_ - x_ * o.x_ - y_ * o.y_ - z_ * o.z_, w_ * o.x_ + x_ * o.w_ + y_ * o.z_ - z_ * o.y_,
                           w_ * o.y_ - x_ * o.z_ + y_ * o.w_ + z_ * o.x_, w_ * o.z_ + x_ * o.y_ - y_ * o.x_ + z_ * o.w_);
    }
    Quaterniond inverse() const { const double n = w_ * w_ + x_ * x_ + y_ * y_ + z_ * z_; return Quaterniond(w_ / n, -x_ / n, -y_ / n, -z_ / n); }
private:
    double w_, x_, y_, z_;
};

// Rigid transform; matrix() is the homogeneous 4x4.
class Isometry3d {
public:
    Isometry3d() : m_(Matrix4d::Identity()) {}
    explicit Isometry3d(const Matrix4d& m) : m_(m) {}
    static Isometry3d Identity() { return Isometry3d(); }
    const Matrix4d& matrix() const { return m_; }
    Matrix4d& matrix() { return m_; }
    Vector3d translation() const { return Vector3d(m_(0, 3), m_(1, 3), m_(2, 3)); }
    void setTranslation(double x, double y, double z) { m_(0, 3) = x; m_(1, 3) = y; m_(2, 3) = z; }
    Matrix3d rotation() const { Matrix3d r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = m_(i, j); return r; }
    Isometry3d inverse() const {
        Isometry3d r;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m_(i, j) = m_(j, i);
        for (int i = 0; i < 3; i++) r.m_(i, 3) = -(r.m_(i, 0) * m_(0, 3) + r.m_(i, 1) * m_(1, 3) + r.m_(i, 2) * m_(2, 3));
        return r;
    }
    Isometry3d operator*(const Isometry3d& o) const { return Isometry3d(m_ * o.m_); }
    Vector4d operator*(const Vector4d& v) const {
        Vector4d r;
        for (int i = 0; i < 4; i++) r(i) = m_(i, 0) * v(0) + m_(i, 1) * v(1) + m_(i, 2) * v(2) + m_(i, 3) * v(3);
        return r;
    }
    Vector3d operator*(const Vector3d& v) const {
        return Vector3d(m_(0, 0) * v(0) + m_(0, 1) * v(1) + m_(0, 2) * v(2) + m_(0, 3),
                        m_(1, 0) * v(0) + m_(1, 1) * v(1) + m_(1, 2) * v(2) + m_(1, 3),
                        m_(2, 0) * v(0) + m_(2, 1) * v(1) + m_(2, 2) * v(2) + m_(2, 3));
    }
private:
    Matrix4d m_;
};

}  // namespace Eigen
#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif
#endif
