// vecmath.hpp — the handful of glm operations the reference's scene layer uses
// (reference src/scene/*.cppm include <glm/glm.hpp>, <glm/gtc/matrix_transform.hpp>), restated so
// the host layer has no third-party dependency.  Storage and indexing follow glm: matrices are
// column-major and m[c][r] is column c, row r — the reference's rotate() quirk depends
// on exactly that indexing.
#pragma once
#include <cmath>
#include <cstdint>

namespace rtr::vm {

struct vec2 { float x = 0.f, y = 0.f; };

struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    bool operator==(const vec3& o) const { return x == o.x && y == o.y && z == o.z; }
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline float degrees(float rad) { return rad * 57.295779513082320876798154814105f; }
inline vec3 radians(vec3 d) { return {radians(d.x), radians(d.y), radians(d.z)}; }

struct mat3 {
    vec3 c[3];   // columns
    vec3& operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};
inline mat3 operator*(const mat3& a, const mat3& b) {
    mat3 r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) r[j][i] = a[0][i] * b[j][0] + a[1][i] * b[j][1] + a[2][i] * b[j][2];
    return r;
}
// glm::mat3(glm::rotate(glm::mat4(1.0f), angle, axis))
inline mat3 rotation(float angle, vec3 v) {
    const float c = std::cos(angle), s = std::sin(angle);
    const vec3 axis = normalize(v);
    const vec3 temp = (1.0f - c) * axis;
    mat3 R;
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = temp[0] * axis[1] + s * axis[2];
    R[0][2] = temp[0] * axis[2] - s * axis[1];
    R[1][0] = temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = temp[1] * axis[2] + s * axis[0];
    R[2][0] = temp[2] * axis[0] + s * axis[1];
    R[2][1] = temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    return R;
}

// vk::TransformMatrixKHR: 3 rows x 4 columns, row-major
struct Transform34 {
    float matrix[3][4] = {{1.f, 0.f, 0.f, 0.f}, {0.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f}};
};

}  // namespace rtr::vm
