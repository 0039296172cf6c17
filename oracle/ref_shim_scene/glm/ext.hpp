// Stand-in for <glm/ext.hpp> (see glm.hpp next to this file): the transform helpers of
// gtx/transform + gtc/quaternion the reference's util/flatten_gltf.cpp uses, with GLM's formulas.
#pragma once
#include "glm.hpp"
#include "gtc/type_ptr.hpp"

namespace glm {

// gtx/transform.hpp: translate(v) = translate(mat4(1), v); ext/matrix_transform.inl:
// Result[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3]
template <typename T> tmat4<T> translate(const tvec3<T> &v)
{
    const tmat4<T> m(T(1));
    tmat4<T> r(m);
    r[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3];
    return r;
}
// scale(v) = scale(mat4(1), v): Result[i] = m[i] * v[i], Result[3] = m[3]
template <typename T> tmat4<T> scale(const tvec3<T> &v)
{
    const tmat4<T> m(T(1));
    tmat4<T> r(T(0));
    r[0] = m[0] * v[0];
    r[1] = m[1] * v[1];
    r[2] = m[2] * v[2];
    r[3] = m[3];
    return r;
}
// gtc/quaternion.inl mat3_cast, widened to 4x4
template <typename T> tmat4<T> mat4_cast(const tquat<T> &q)
{
    tmat4<T> r(T(1));
    const T qxx(q.x * q.x), qyy(q.y * q.y), qzz(q.z * q.z), qxz(q.x * q.z), qxy(q.x * q.y), qyz(q.y * q.z), qwx(q.w * q.x),
        qwy(q.w * q.y), qwz(q.w * q.z);
    r[0][0] = T(1) - T(2) * (qyy + qzz);
    r[0][1] = T(2) * (qxy + qwz);
    r[0][2] = T(2) * (qxz - qwy);
    r[1][0] = T(2) * (qxy - qwz);
    r[1][1] = T(1) - T(2) * (qxx + qzz);
    r[1][2] = T(2) * (qyz + qwx);
    r[2][0] = T(2) * (qxz + qwy);
    r[2][1] = T(2) * (qyz - qwx);
    r[2][2] = T(1) - T(2) * (qxx + qyy);
    return r;
}

} // namespace glm
