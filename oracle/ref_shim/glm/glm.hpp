// Stand-in for the subset of GLM (third-party, not vendored by the reference) that the reference's
// embree_sycl headers and kernel use. TEST INFRASTRUCTURE: only oracle/ref_driver.cpp is compiled
// against it (oracle/Makefile, target _ref). Not GLM, not reference code: plain structs with the
// member names and the handful of operators those files need.
#pragma once
#include <cstdint>

namespace glm {

struct vec2 {
    float x, y;
    vec2() : x(0.f), y(0.f) {}
    explicit vec2(float s) : x(s), y(s) {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
};
struct vec3 {
    float x, y, z;
    vec3() : x(0.f), y(0.f), z(0.f) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct vec4 {
    float x, y, z, w;
    vec4() : x(0.f), y(0.f), z(0.f), w(0.f) {}
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};
struct uvec3 {
    uint32_t x, y, z;
};
struct mat4 {
    float m[16];
};

// component-wise, evaluated left to right like GLM's templates
inline vec2 operator*(float s, const vec2 &v) { return vec2(s * v.x, s * v.y); }
inline vec2 operator*(const vec2 &v, float s) { return vec2(v.x * s, v.y * s); }
inline vec2 operator+(const vec2 &a, const vec2 &b) { return vec2(a.x + b.x, a.y + b.y); }

// glm::clamp(x, lo, hi) = min(max(x, lo), hi)
inline float clamp(float x, float lo, float hi)
{
    const float t = x < lo ? lo : x;
    return t > hi ? hi : t;
}

} // namespace glm
