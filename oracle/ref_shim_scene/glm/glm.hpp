// Stand-in for the part of GLM (third-party: fetched by the reference's cmake/glm.cmake, not vendored,
// absent from this image) that the reference's scene importer uses: util/scene.cpp, mesh.cpp,
// material.cpp, util.cpp, flatten_gltf.cpp, gltf_types.cpp, buffer_view.cpp and their headers.
// TEST INFRASTRUCTURE: only oracle/ref_scene_driver.cpp + those reference files are compiled against it
// (oracle/Makefile, target _ref/libref_scene.so), to produce golden Scene dumps for the importer tests.
// Not GLM, not reference code: templated PODs with GLM's member names and, where the arithmetic shows
// in the result (dot, normalize, cross, matrix product, translate / scale / mat4_cast), GLM's
// published formulas evaluated in GLM's order.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace glm {

template <typename T> struct tvec3;
template <typename T> struct tvec4;

template <typename T> struct tvec2 {
    union { T x, r, s; };
    union { T y, g, t; };
    tvec2() : x(0), y(0) {}
    explicit tvec2(T v) : x(v), y(v) {}
    template <typename A, typename B> tvec2(A x_, B y_) : x((T)x_), y((T)y_) {}
    template <typename U> tvec2(const tvec2<U> &v) : x((T)v.x), y((T)v.y) {}
    T &operator[](int i) { return i == 0 ? x : y; }
    const T &operator[](int i) const { return i == 0 ? x : y; }
};

template <typename T> struct tvec3 {
    union { T x, r, s; };
    union { T y, g, t; };
    union { T z, b, p; };
    tvec3() : x(0), y(0), z(0) {}
    explicit tvec3(T v) : x(v), y(v), z(v) {}
    template <typename A, typename B, typename C> tvec3(A x_, B y_, C z_) : x((T)x_), y((T)y_), z((T)z_) {}
    template <typename U> tvec3(const tvec3<U> &v) : x((T)v.x), y((T)v.y), z((T)v.z) {}
    template <typename U> tvec3(const tvec4<U> &v); // GLM: implicit unless GLM_FORCE_EXPLICIT_CTOR
    T &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    const T &operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

template <typename T> struct tvec4 {
    union { T x, r, s; };
    union { T y, g, t; };
    union { T z, b, p; };
    union { T w, a, q; };
    tvec4() : x(0), y(0), z(0), w(0) {}
    explicit tvec4(T v) : x(v), y(v), z(v), w(v) {}
    template <typename A, typename B, typename C, typename D>
    tvec4(A x_, B y_, C z_, D w_) : x((T)x_), y((T)y_), z((T)z_), w((T)w_) {}
    template <typename U, typename W> tvec4(const tvec3<U> &v, W w_) : x((T)v.x), y((T)v.y), z((T)v.z), w((T)w_) {}
    template <typename U> tvec4(const tvec4<U> &v) : x((T)v.x), y((T)v.y), z((T)v.z), w((T)v.w) {}
    T &operator[](int i) { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    const T &operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
};
template <typename T> template <typename U> tvec3<T>::tvec3(const tvec4<U> &v) : x((T)v.x), y((T)v.y), z((T)v.z) {}

typedef tvec2<float> vec2;
typedef tvec3<float> vec3;
typedef tvec4<float> vec4;
typedef tvec2<uint32_t> uvec2;
typedef tvec3<uint32_t> uvec3;
typedef tvec3<int32_t> ivec3;
typedef tvec2<int32_t> ivec2;
typedef tvec4<uint32_t> uvec4;

// comparison: templates, like GLM's, so that a caller's own non-template overloads win
template <typename T> bool operator==(const tvec2<T> &a, const tvec2<T> &b) { return a.x == b.x && a.y == b.y; }
template <typename T> bool operator==(const tvec3<T> &a, const tvec3<T> &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
template <typename T> bool operator==(const tvec4<T> &a, const tvec4<T> &b)
{
    return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
}

// component-wise arithmetic
template <typename T> tvec2<T> operator+(const tvec2<T> &a, const tvec2<T> &b) { return tvec2<T>(a.x + b.x, a.y + b.y); }
template <typename T> tvec2<T> operator*(const tvec2<T> &a, T s) { return tvec2<T>(a.x * s, a.y * s); }
template <typename T> tvec2<T> operator*(T s, const tvec2<T> &a) { return tvec2<T>(s * a.x, s * a.y); }
template <typename T> tvec3<T> operator+(const tvec3<T> &a, const tvec3<T> &b) { return tvec3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> tvec3<T> operator-(const tvec3<T> &a, const tvec3<T> &b) { return tvec3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> tvec3<T> operator-(const tvec3<T> &a) { return tvec3<T>(-a.x, -a.y, -a.z); }
template <typename T> tvec3<T> operator*(const tvec3<T> &a, T s) { return tvec3<T>(a.x * s, a.y * s, a.z * s); }
template <typename T> tvec3<T> operator*(T s, const tvec3<T> &a) { return tvec3<T>(s * a.x, s * a.y, s * a.z); }
template <typename T> tvec3<T> operator*(const tvec3<T> &a, const tvec3<T> &b) { return tvec3<T>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename T> tvec4<T> operator+(const tvec4<T> &a, const tvec4<T> &b)
{
    return tvec4<T>(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
template <typename T> tvec4<T> operator-(const tvec4<T> &a) { return tvec4<T>(-a.x, -a.y, -a.z, -a.w); }
template <typename T> tvec4<T> operator*(const tvec4<T> &a, T s) { return tvec4<T>(a.x * s, a.y * s, a.z * s, a.w * s); }
template <typename T> tvec4<T> operator*(T s, const tvec4<T> &a) { return tvec4<T>(s * a.x, s * a.y, s * a.z, s * a.w); }
template <typename T> tvec4<T> operator*(const tvec4<T> &a, const tvec4<T> &b)
{
    return tvec4<T>(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}

// geometric functions: GLM's detail::compute_dot sums (x + y) + z, and (x + y) + (z + w) for vec4;
// normalize(v) = v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x)
template <typename T> T dot(const tvec3<T> &a, const tvec3<T> &b)
{
    const tvec3<T> t(a * b);
    return t.x + t.y + t.z;
}
template <typename T> T dot(const tvec4<T> &a, const tvec4<T> &b)
{
    const tvec4<T> t(a * b);
    return (t.x + t.y) + (t.z + t.w);
}
template <typename T> T inversesqrt(T x) { return T(1) / std::sqrt(x); }
template <typename T> tvec3<T> normalize(const tvec3<T> &v) { return v * inversesqrt(dot(v, v)); }
template <typename T> tvec4<T> normalize(const tvec4<T> &v) { return v * inversesqrt(dot(v, v)); }
template <typename T> tvec3<T> cross(const tvec3<T> &x, const tvec3<T> &y)
{
    return tvec3<T>(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
template <typename T> T clamp(T x, T lo, T hi)
{
    const T t = x < lo ? lo : x; // min(max(x, lo), hi)
    return t > hi ? hi : t;
}

// column-major 4x4
template <typename T> struct tmat4 {
    tvec4<T> value[4];
    tmat4() : tmat4(T(1)) {}
    explicit tmat4(T s)
    {
        value[0] = tvec4<T>(s, 0, 0, 0);
        value[1] = tvec4<T>(0, s, 0, 0);
        value[2] = tvec4<T>(0, 0, s, 0);
        value[3] = tvec4<T>(0, 0, 0, s);
    }
    template <typename U> tmat4(const tmat4<U> &m)
    {
        for (int c = 0; c < 4; ++c) {
            value[c] = tvec4<T>(m.value[c]);
        }
    }
    tvec4<T> &operator[](int c) { return value[c]; }
    const tvec4<T> &operator[](int c) const { return value[c]; }
};
typedef tmat4<float> mat4;

// GLM's mat4 * mat4: each result column = ((A0 * b0 + A1 * b1) + A2 * b2) + A3 * b3
template <typename T> tmat4<T> operator*(const tmat4<T> &m1, const tmat4<T> &m2)
{
    tmat4<T> r(T(0));
    for (int c = 0; c < 4; ++c) {
        r[c] = m1[0] * m2[c][0] + m1[1] * m2[c][1] + m1[2] * m2[c][2] + m1[3] * m2[c][3];
    }
    return r;
}
template <typename T> tvec4<T> column(const tmat4<T> &m, int c) { return m[c]; }

template <typename T> struct tquat {
    T x, y, z, w;
    tquat(T w_, T x_, T y_, T z_) : x(x_), y(y_), z(z_), w(w_) {} // GLM's (w, x, y, z) constructor order
    template <typename A, typename B, typename C, typename D> tquat(A w_, B x_, C y_, D z_) : x((T)x_), y((T)y_), z((T)z_), w((T)w_) {}
};
typedef tquat<float> quat;

} // namespace glm
