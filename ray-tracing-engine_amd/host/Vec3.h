// Vec3.h — small fixed-size vector for the host side of the MI355X renderer.
//
// Written from scratch (the reference's source/Vec3.h carries a restrictive
// third-party notice, SURVEY.md §0.7, and is not used).  What IS kept is the
// public surface user code relies on (operator set, free functions dot / cross /
// normalize / dist / length / mix, Vec3f / Vec3d / Vec3i, stream operators —
// reference source/Vec3.h:18-318) and the arithmetic conventions parity depends
// on: dot = x*x' + y*y' + z*z' evaluated left to right, normalize() multiplies
// by 1/length and leaves zero-length vectors untouched, scalar * vector is
// evaluated as vector * scalar.
#pragma once

#include <cmath>
#include <istream>
#include <ostream>

template <typename T>
class Vec3 {
 public:
  Vec3() : c_{T(0), T(0), T(0)} {}
  Vec3(T x, T y, T z) : c_{x, y, z} {}

  T& operator[](int i) { return c_[i]; }
  const T& operator[](int i) const { return c_[i]; }

  Vec3& init(T x, T y, T z) {
    c_[0] = x, c_[1] = y, c_[2] = z;
    return *this;
  }

  Vec3& operator+=(const Vec3& o) {
    c_[0] += o.c_[0], c_[1] += o.c_[1], c_[2] += o.c_[2];
    return *this;
  }
  Vec3& operator-=(const Vec3& o) {
    c_[0] -= o.c_[0], c_[1] -= o.c_[1], c_[2] -= o.c_[2];
    return *this;
  }
  Vec3& operator*=(const Vec3& o) {
    c_[0] *= o.c_[0], c_[1] *= o.c_[1], c_[2] *= o.c_[2];
    return *this;
  }
  Vec3& operator/=(const Vec3& o) {
    c_[0] /= o.c_[0], c_[1] /= o.c_[1], c_[2] /= o.c_[2];
    return *this;
  }
  Vec3& operator*=(T s) {
    for (int i = 0; i < 3; ++i) c_[i] *= s;
    return *this;
  }
  Vec3& operator/=(T s) {
    for (int i = 0; i < 3; ++i) c_[i] /= s;
    return *this;
  }

  Vec3 operator+(const Vec3& o) const { return Vec3(c_[0] + o.c_[0], c_[1] + o.c_[1], c_[2] + o.c_[2]); }
  Vec3 operator-(const Vec3& o) const { return Vec3(c_[0] - o.c_[0], c_[1] - o.c_[1], c_[2] - o.c_[2]); }
  Vec3 operator*(const Vec3& o) const { return Vec3(c_[0] * o.c_[0], c_[1] * o.c_[1], c_[2] * o.c_[2]); }
  Vec3 operator/(const Vec3& o) const { return Vec3(c_[0] / o.c_[0], c_[1] / o.c_[1], c_[2] / o.c_[2]); }
  Vec3 operator-() const { return Vec3(-c_[0], -c_[1], -c_[2]); }
  Vec3 operator*(T s) const { return Vec3(c_[0] * s, c_[1] * s, c_[2] * s); }
  Vec3 operator/(T s) const { return Vec3(c_[0] / s, c_[1] / s, c_[2] / s); }

  bool operator==(const Vec3& o) const { return c_[0] == o.c_[0] && c_[1] == o.c_[1] && c_[2] == o.c_[2]; }
  bool operator!=(const Vec3& o) const { return !(*this == o); }
  // component-wise "all less" / "all greater-or-equal", as in the reference API
  bool operator<(const Vec3& o) const { return c_[0] < o.c_[0] && c_[1] < o.c_[1] && c_[2] < o.c_[2]; }
  bool operator>=(const Vec3& o) const { return c_[0] >= o.c_[0] && c_[1] >= o.c_[1] && c_[2] >= o.c_[2]; }

  T squaredLength() const { return c_[0] * c_[0] + c_[1] * c_[1] + c_[2] * c_[2]; }
  T length() const { return static_cast<T>(std::sqrt(squaredLength())); }

  // Normalises in place, returns the previous length; a null vector stays null.
  T normalize() {
    const T len = length();
    if (len == T(0)) return T(0);
    const T inv = T(1) / len;
    c_[0] *= inv, c_[1] *= inv, c_[2] *= inv;
    return len;
  }

  // Two vectors orthogonal to *this (and to each other): zero the component of
  // smallest magnitude, swap-and-negate the other two (ties resolved like the
  // reference so that hemisphere frames match), then complete by a cross product.
  void getTwoOrthogonals(Vec3& u, Vec3& v) const {
    const T ax = std::fabs(c_[0]), ay = std::fabs(c_[1]), az = std::fabs(c_[2]);
    if (ax < ay) {
      u = (ax < az) ? Vec3(T(0), -c_[2], c_[1]) : Vec3(-c_[1], c_[0], T(0));
    } else {
      u = (ay < az) ? Vec3(c_[2], T(0), -c_[0]) : Vec3(-c_[1], c_[0], T(0));
    }
    v = Vec3(c_[1] * u.c_[2] - c_[2] * u.c_[1], c_[2] * u.c_[0] - c_[0] * u.c_[2],
             c_[0] * u.c_[1] - c_[1] * u.c_[0]);
  }

  Vec3 projectOn(const Vec3& N, const Vec3& P) const {
    const Vec3 rel = *this - P;
    const T w = rel.c_[0] * N.c_[0] + rel.c_[1] * N.c_[1] + rel.c_[2] * N.c_[2];
    return *this - N * w;
  }

 protected:
  T c_[3];
};

template <class T>
inline T dot(const Vec3<T>& a, const Vec3<T>& b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
template <class T>
inline Vec3<T> cross(const Vec3<T>& a, const Vec3<T>& b) {
  return Vec3<T>(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
template <class T>
inline T length(const Vec3<T>& a) {
  return a.length();
}
template <class T>
inline T dist(const Vec3<T>& a, const Vec3<T>& b) {
  return (a - b).length();
}
template <class T>
inline Vec3<T> normalize(const Vec3<T>& a) {
  Vec3<T> r(a);
  r.normalize();
  return r;
}
template <class T>
inline Vec3<T> mix(const Vec3<T>& u, const Vec3<T>& v, float alpha) {
  return u * (T(1) - alpha) + v * alpha;
}
template <class T>
inline Vec3<T> operator*(const T& s, const Vec3<T>& v) {
  return v * s;
}

// spherical <-> cartesian helpers ([0] radius, [1] polar angle from +z, [2] azimuth)
template <class T>
inline Vec3<T> cartesianToPolar(const Vec3<T>& v) {
  const T rho = std::sqrt(v[0] * v[0] + v[1] * v[1]);
  const T pi = T(3.14159265358979323846);
  Vec3<T> p;
  p[0] = v.length();
  p[1] = v[2] > T(0) ? T(std::atan(rho / v[2])) : v[2] < T(0) ? T(std::atan(rho / v[2])) + pi : pi * T(0.5);
  if (v[0] > T(0)) p[2] = T(std::atan(v[1] / v[0]));
  else if (v[0] < T(0)) p[2] = T(std::atan(v[1] / v[0])) + pi;
  else p[2] = v[1] > T(0) ? pi * T(0.5) : -pi * T(0.5);
  return p;
}
template <class T>
inline Vec3<T> polarToCartesian(const Vec3<T>& p) {
  const T s = T(std::sin(p[1]));
  return Vec3<T>(p[0] * s * T(std::cos(p[2])), p[0] * s * T(std::sin(p[2])), p[0] * T(std::cos(p[1])));
}
template <class T>
inline Vec3<T> projectOntoVector(const Vec3<T>& a, const Vec3<T>& onto) {
  return onto * dot(a, onto);
}

template <class T>
std::ostream& operator<<(std::ostream& os, const Vec3<T>& v) {
  return os << v[0] << " " << v[1] << " " << v[2];
}
template <class T>
std::istream& operator>>(std::istream& is, Vec3<T>& v) {
  return is >> v[0] >> v[1] >> v[2];
}

typedef Vec3<float> Vec3f;
typedef Vec3<double> Vec3d;
typedef Vec3<int> Vec3i;
