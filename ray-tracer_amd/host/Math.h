// Host-side vector / quaternion / colour types used to load and flatten scenes.
//
// These mirror the arithmetic of the reference's types (include/Vector.h,
// src/Quaternion.cpp:29-119, src/Color.cpp) expression for expression, because the
// world transforms they produce are inputs of the photon pass: a parent/child
// composition rounded differently would move every hit by an ulp.  Compile with
// -ffp-contract=off.
#pragma once

#include <cmath>

struct Vector
{
    double x = 0.0, y = 0.0, z = 0.0;

    Vector() = default;
    Vector(double ix, double iy, double iz) : x(ix), y(iy), z(iz) {}

    double magnitudeSquared() const { return x * x + y * y + z * z; }
    double magnitude() const { return std::sqrt(magnitudeSquared()); }
    Vector normalized() const
    {
        const double root = std::sqrt(magnitudeSquared());
        return {x / root, y / root, z / root};
    }

    static double dot(const Vector& a, const Vector& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
    static Vector cross(const Vector& a, const Vector& b)
    {
        return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    }
};

inline Vector operator+(const Vector& a, const Vector& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector operator-(const Vector& a, const Vector& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vector operator*(const Vector& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vector operator*(double s, const Vector& a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vector operator/(const Vector& a, double s) { return {a.x / s, a.y / s, a.z / s}; }

struct Quaternion
{
    double x = 0.0, y = 0.0, z = 0.0, w = 1.0;

    Quaternion() = default;
    Quaternion(double ix, double iy, double iz, double iw) : x(ix), y(iy), z(iz), w(iw) {}

    Quaternion conjugate() const { return {-x, -y, -z, w}; }

    static Quaternion fromPitchYawRoll(double pitch, double yaw, double roll)
    {
        const double p = pitch / 2.0, y = yaw / 2.0, r = roll / 2.0;
        const double cp = std::cos(p), cy = std::cos(y), cr = std::cos(r);
        const double sp = std::sin(p), sy = std::sin(y), sr = std::sin(r);
        return {sp * cy * cr + cp * sy * sr, cp * sy * cr - sp * cy * sr, cp * cy * sr + sp * sy * cr,
                cp * cy * cr - sp * sy * sr};
    }
};

inline Quaternion operator*(const Quaternion& l, const Quaternion& r)
{
    return {l.w * r.x + l.x * r.w + l.y * r.z - l.z * r.y, l.w * r.y + l.y * r.w + l.z * r.x - l.x * r.z,
            l.w * r.z + l.z * r.w + l.x * r.y - l.y * r.x, l.w * r.w - l.x * r.x - l.y * r.y - l.z * r.z};
}

inline Vector operator*(const Quaternion& q, const Vector& v)
{
    const Quaternion p{v.x, v.y, v.z, 0.0};
    const Quaternion r = (q * p) * q.conjugate();
    return {r.x, r.y, r.z};
}

struct Transform
{
    Vector position;
    Quaternion rotation;
    Vector scale{1.0, 1.0, 1.0};
};

struct Color
{
    float red = 0.0f, green = 0.0f, blue = 0.0f;

    Color() = default;
    Color(float grey) : red(grey), green(grey), blue(grey) {}
    Color(float r, float g, float b) : red(r), green(g), blue(b) {}

    float brightness() const { return red + green + blue; }
    static constexpr float gamma = 2.2f;
};

inline Color operator*(const Color& c, float s) { return {c.red * s, c.green * s, c.blue * s}; }

namespace Utility
{
constexpr double pi = M_PI;
constexpr double pi2 = M_PI * 2.0;
constexpr double pi4 = M_PI * 4.0;
inline double radians(double degrees) { return pi * (degrees / 180.0); }
}
