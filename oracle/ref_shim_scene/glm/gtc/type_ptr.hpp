// Stand-in for <glm/gtc/type_ptr.hpp> (see ../glm.hpp).
#pragma once
#include "../glm.hpp"

namespace glm {

template <typename T> const T *value_ptr(const tmat4<T> &m) { return &m.value[0].x; }
template <typename T> T *value_ptr(tmat4<T> &m) { return &m.value[0].x; }
template <typename T> const T *value_ptr(const tvec3<T> &v) { return &v.x; }
template <typename T> tvec3<T> make_vec3(const T *p) { return tvec3<T>(p[0], p[1], p[2]); }
template <typename T> tmat4<T> make_mat4(const T *p)
{
    tmat4<T> m(T(0));
    for (int c = 0; c < 4; ++c) {
        m[c] = tvec4<T>(p[4 * c], p[4 * c + 1], p[4 * c + 2], p[4 * c + 3]);
    }
    return m;
}

} // namespace glm
