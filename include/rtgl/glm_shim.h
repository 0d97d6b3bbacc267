// glm_shim.h -- the subset of glm the reference's public API surface uses (src/renderer.h, src/kdtree.h,
// src/main.cpp): vec2/3/4, ivec2, mat4, quat and a handful of free functions.  Written from glm's
// documented semantics (column-major mat4, quat from Euler angles in pitch/yaw/roll = x/y/z order);
// not a copy of glm.  If the real glm is on the include path, define RTGL_USE_REAL_GLM and this header
// simply includes it.
#pragma once
#ifdef RTGL_USE_REAL_GLM
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtc/quaternion.hpp>
#else
#include <cmath>
#include <ostream>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace glm {

struct vec2 { float x = 0, y = 0; vec2() = default; vec2(float a, float b) : x(a), y(b) {} explicit vec2(float s) : x(s), y(s) {} };
struct ivec2 { int x = 0, y = 0; ivec2() = default; ivec2(int a, int b) : x(a), y(b) {} explicit ivec2(int s) : x(s), y(s) {} };

struct vec3 {
    float x = 0, y = 0, z = 0;
    constexpr vec3() = default;
    constexpr vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    constexpr explicit vec3(float s) : x(s), y(s), z(s) {}
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    const float &operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    vec3 &operator+=(const vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3 &operator-=(const vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
};
constexpr vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
constexpr vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
constexpr vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
constexpr vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
constexpr vec3 operator*(float s, vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
constexpr vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
constexpr vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
constexpr vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
constexpr vec3 operator-(vec3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }

struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    constexpr vec4() = default;
    constexpr vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    constexpr explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    constexpr vec4(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    const float &operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
};
constexpr vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
constexpr vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
constexpr vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline float min(float a, float b) { return b < a ? b : a; }
inline float max(float a, float b) { return a < b ? b : a; }
inline vec3 min(vec3 a, vec3 b) { return {min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)}; }
inline vec3 max(vec3 a, vec3 b) { return {max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)}; }
inline vec4 min(vec4 a, vec4 b) { return {min(a.x, b.x), min(a.y, b.y), min(a.z, b.z), min(a.w, b.w)}; }
inline vec4 max(vec4 a, vec4 b) { return {max(a.x, b.x), max(a.y, b.y), max(a.z, b.z), max(a.w, b.w)}; }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// unit quaternion; quat(vec3 euler) takes (pitch, yaw, roll) = rotations about x, y, z
struct quat {
    float w = 1, x = 0, y = 0, z = 0;
    quat() = default;
    quat(float w_, float x_, float y_, float z_) : w(w_), x(x_), y(y_), z(z_) {}
    explicit quat(vec3 e)
    {
        vec3 c(std::cos(e.x * 0.5f), std::cos(e.y * 0.5f), std::cos(e.z * 0.5f));
        vec3 s(std::sin(e.x * 0.5f), std::sin(e.y * 0.5f), std::sin(e.z * 0.5f));
        w = c.x * c.y * c.z + s.x * s.y * s.z;
        x = s.x * c.y * c.z - c.x * s.y * s.z;
        y = c.x * s.y * c.z + s.x * c.y * s.z;
        z = c.x * c.y * s.z - s.x * s.y * c.z;
    }
};

// column-major 4x4: m[c] is column c
struct mat4 {
    vec4 c[4];
    mat4() : mat4(0.0f) {}
    explicit mat4(float d) { c[0] = {d, 0, 0, 0}; c[1] = {0, d, 0, 0}; c[2] = {0, 0, d, 0}; c[3] = {0, 0, 0, d}; }
    explicit mat4(const quat &q)
    {
        float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
        float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
        c[0] = {1 - 2 * (qyy + qzz), 2 * (qxy + qwz), 2 * (qxz - qwy), 0};
        c[1] = {2 * (qxy - qwz), 1 - 2 * (qxx + qzz), 2 * (qyz + qwx), 0};
        c[2] = {2 * (qxz + qwy), 2 * (qyz - qwx), 1 - 2 * (qxx + qyy), 0};
        c[3] = {0, 0, 0, 1};
    }
    vec4 &operator[](int i) { return c[i]; }
    const vec4 &operator[](int i) const { return c[i]; }
};
inline vec4 operator*(const mat4 &m, vec4 v)
{
    return {m[0].x * v.x + m[1].x * v.y + m[2].x * v.z + m[3].x * v.w, m[0].y * v.x + m[1].y * v.y + m[2].y * v.z + m[3].y * v.w,
            m[0].z * v.x + m[1].z * v.y + m[2].z * v.z + m[3].z * v.w, m[0].w * v.x + m[1].w * v.y + m[2].w * v.z + m[3].w * v.w};
}
inline mat4 operator*(const mat4 &a, const mat4 &b)
{
    mat4 r(0.0f);
    for (int j = 0; j < 4; ++j) r[j] = a * b[j];
    return r;
}
inline mat4 translate(const mat4 &m, vec3 t)
{
    mat4 r = m;
    r[3] = m[0] * t.x + m[1] * t.y + m[2] * t.z + m[3];
    return r;
}
inline mat4 scale(const mat4 &m, vec3 s)
{
    mat4 r = m;
    r[0] = m[0] * s.x; r[1] = m[1] * s.y; r[2] = m[2] * s.z;
    return r;
}
inline const float *value_ptr(const vec3 &v) { return &v.x; }
inline const float *value_ptr(const vec4 &v) { return &v.x; }
inline std::ostream &operator<<(std::ostream &os, const vec3 &v) { return os << "vec3(" << v.x << ", " << v.y << ", " << v.z << ")"; }
inline std::ostream &operator<<(std::ostream &os, const vec4 &v) { return os << "vec4(" << v.x << ", " << v.y << ", " << v.z << ", " << v.w << ")"; }

}  // namespace glm
#endif
