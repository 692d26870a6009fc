// lml.h -- the few vector / quaternion / matrix types the ocean path touches.
//
// datum gets these from the un-vendored `leap` library (src/math/vec.h:11-17, src/math/transform.h,
// README.md:31).  leap is not part of the reference checkout, so this is a from-scratch minimal
// equivalent with the same names and the call sites' meaning (Vec2/Vec3/Vec4, Plane, Quaternion3,
// Transform as a dual quaternion, Matrix4f, normsqr/dot/lerp/normalise/cross/pi).  Quaternions are
// (w, x, y, z), matching data/transform.inc:13-28.

#pragma once

#include <cmath>

namespace lml
{
  template<typename T> constexpr T pi() { return T(3.14159265358979323846L); }

  struct Vec2
  {
    float x, y;

    Vec2() = default;
    constexpr Vec2(float x, float y) : x(x), y(y) { }
    explicit constexpr Vec2(float k) : x(k), y(k) { }
  };

  struct Vec3
  {
    float x, y, z;

    Vec3() = default;
    constexpr Vec3(float x, float y, float z) : x(x), y(y), z(z) { }
    explicit constexpr Vec3(float k) : x(k), y(k), z(k) { }
  };

  struct Vec4
  {
    float x, y, z, w;

    Vec4() = default;
    constexpr Vec4(float x, float y, float z, float w) : x(x), y(y), z(z), w(w) { }
    constexpr Vec4(Vec3 const &v, float w) : x(v.x), y(v.y), z(v.z), w(w) { }
  };

  struct Plane
  {
    Vec3 normal;
    float distance;
  };

  inline constexpr bool operator==(Vec2 const &a, Vec2 const &b) { return a.x == b.x && a.y == b.y; }
  inline constexpr bool operator!=(Vec2 const &a, Vec2 const &b) { return !(a == b); }

  inline constexpr Vec2 operator+(Vec2 const &a, Vec2 const &b) { return { a.x + b.x, a.y + b.y }; }
  inline constexpr Vec2 operator-(Vec2 const &a, Vec2 const &b) { return { a.x - b.x, a.y - b.y }; }
  inline constexpr Vec2 operator*(float s, Vec2 const &a) { return { s * a.x, s * a.y }; }
  inline constexpr Vec2 operator*(Vec2 const &a, float s) { return { a.x * s, a.y * s }; }
  inline Vec2 &operator+=(Vec2 &a, Vec2 const &b) { a.x += b.x; a.y += b.y; return a; }

  inline constexpr Vec3 operator+(Vec3 const &a, Vec3 const &b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
  inline constexpr Vec3 operator-(Vec3 const &a, Vec3 const &b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
  inline constexpr Vec3 operator-(Vec3 const &a) { return { -a.x, -a.y, -a.z }; }
  inline constexpr Vec3 operator*(float s, Vec3 const &a) { return { s * a.x, s * a.y, s * a.z }; }

  inline constexpr float dot(Vec2 const &a, Vec2 const &b) { return a.x * b.x + a.y * b.y; }
  inline constexpr float dot(Vec3 const &a, Vec3 const &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

  inline constexpr float normsqr(Vec2 const &a) { return dot(a, a); }
  inline constexpr float normsqr(Vec3 const &a) { return dot(a, a); }

  inline float norm(Vec2 const &a) { return std::sqrt(normsqr(a)); }
  inline float norm(Vec3 const &a) { return std::sqrt(normsqr(a)); }

  inline Vec2 normalise(Vec2 const &a) { float l = norm(a); return { a.x / l, a.y / l }; }
  inline Vec3 normalise(Vec3 const &a) { float l = norm(a); return { a.x / l, a.y / l, a.z / l }; }

  inline constexpr Vec3 cross(Vec3 const &a, Vec3 const &b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }

  // a vector orthogonal to both arguments (right-handed)
  inline constexpr Vec3 orthogonal(Vec3 const &a, Vec3 const &b) { return cross(a, b); }

  inline constexpr float lerp(float a, float b, float t) { return (1 - t) * a + t * b; }
  inline constexpr Vec2 lerp(Vec2 const &a, Vec2 const &b, float t) { return { lerp(a.x, b.x, t), lerp(a.y, b.y, t) }; }

  //|---------------------- Quaternion3 ---------------------------------------

  struct Quaternion3
  {
    float w, x, y, z;

    Quaternion3() = default;
    constexpr Quaternion3(float w, float x, float y, float z) : w(w), x(x), y(y), z(z) { }
    constexpr Quaternion3(float w, Vec3 const &v) : w(w), x(v.x), y(v.y), z(v.z) { }

    // rotation by `angle` about the unit `axis`
    Quaternion3(Vec3 const &axis, float angle)
    {
      float s = std::sin(angle / 2);
      w = std::cos(angle / 2);
      x = axis.x * s;
      y = axis.y * s;
      z = axis.z * s;
    }

    // rotation whose columns are the given orthonormal axes
    Quaternion3(Vec3 const &xaxis, Vec3 const &yaxis, Vec3 const &zaxis)
    {
      float m00 = xaxis.x, m01 = yaxis.x, m02 = zaxis.x;
      float m10 = xaxis.y, m11 = yaxis.y, m12 = zaxis.y;
      float m20 = xaxis.z, m21 = yaxis.z, m22 = zaxis.z;

      float tr = m00 + m11 + m22;

      if (tr > 0)
      {
        float s = std::sqrt(tr + 1) * 2;
        w = 0.25f * s; x = (m21 - m12) / s; y = (m02 - m20) / s; z = (m10 - m01) / s;
      }
      else if (m00 > m11 && m00 > m22)
      {
        float s = std::sqrt(1 + m00 - m11 - m22) * 2;
        w = (m21 - m12) / s; x = 0.25f * s; y = (m01 + m10) / s; z = (m02 + m20) / s;
      }
      else if (m11 > m22)
      {
        float s = std::sqrt(1 + m11 - m00 - m22) * 2;
        w = (m02 - m20) / s; x = (m01 + m10) / s; y = 0.25f * s; z = (m12 + m21) / s;
      }
      else
      {
        float s = std::sqrt(1 + m22 - m00 - m11) * 2;
        w = (m10 - m01) / s; x = (m02 + m20) / s; y = (m12 + m21) / s; z = 0.25f * s;
      }
    }

    Vec3 xyz() const { return { x, y, z }; }
  };

  inline constexpr Quaternion3 conjugate(Quaternion3 const &q) { return { q.w, -q.x, -q.y, -q.z }; }

  // Hamilton product, component sums in the order of data/transform.inc:22-25
  inline constexpr Quaternion3 operator*(Quaternion3 const &a, Quaternion3 const &b)
  {
    return { a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
             a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
             a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x };
  }

  inline Vec3 operator*(Quaternion3 const &q, Vec3 const &v)
  {
    Vec3 t = 2.0f * cross(q.xyz(), v);

    return v + q.w * t + cross(q.xyz(), t);
  }

  //|---------------------- Transform -----------------------------------------
  // rigid transform as a dual quaternion (src/math/transform.h:26-100)

  struct Transform
  {
    Quaternion3 real;
    Quaternion3 dual;

    static Transform identity() { return { Quaternion3(1, 0, 0, 0), Quaternion3(0, 0, 0, 0) }; }

    // transform.h:86-89
    static Transform lookat(Vec3 const &position, Quaternion3 const &orientation)
    {
      return { orientation, Quaternion3(0.0f, 0.5f * position) * orientation };
    }

    // transform.h:93-100
    static Transform lookat(Vec3 const &position, Vec3 const &target, Vec3 const &up)
    {
      Vec3 zaxis = normalise(position - target);
      Vec3 xaxis = normalise(orthogonal(up, zaxis));
      Vec3 yaxis = cross(zaxis, xaxis);

      return lookat(position, Quaternion3(xaxis, yaxis, zaxis));
    }

    Vec3 translation() const { return 2.0f * (dual * conjugate(real)).xyz(); }   // transform.h:39
    Quaternion3 rotation() const { return real; }
  };

  //|---------------------- Matrix4f ------------------------------------------

  struct Matrix4f
  {
    float m[4][4];   // row major: (i, j) = row i, column j

    float &operator()(int i, int j) { return m[i][j]; }
    float const &operator()(int i, int j) const { return m[i][j]; }
  };

  // general 4x4 inverse (Gauss-Jordan with partial pivoting, double accumulation)
  inline Matrix4f inverse(Matrix4f const &a)
  {
    double w[4][8];

    for(int i = 0; i < 4; ++i)
      for(int j = 0; j < 4; ++j)
      {
        w[i][j] = a.m[i][j];
        w[i][4+j] = (i == j) ? 1.0 : 0.0;
      }

    for(int c = 0; c < 4; ++c)
    {
      int p = c;
      for(int r = c + 1; r < 4; ++r)
        if (std::fabs(w[r][c]) > std::fabs(w[p][c]))
          p = r;

      for(int j = 0; j < 8; ++j)
      {
        double t = w[c][j]; w[c][j] = w[p][j]; w[p][j] = t;
      }

      double d = w[c][c];
      for(int j = 0; j < 8; ++j)
        w[c][j] /= d;

      for(int r = 0; r < 4; ++r)
      {
        if (r == c)
          continue;

        double f = w[r][c];
        for(int j = 0; j < 8; ++j)
          w[r][j] -= f * w[c][j];
      }
    }

    Matrix4f out;
    for(int i = 0; i < 4; ++i)
      for(int j = 0; j < 4; ++j)
        out.m[i][j] = (float)w[i][4+j];

    return out;
  }
}
