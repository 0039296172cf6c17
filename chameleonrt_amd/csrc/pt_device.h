// pt_device.h — device-side shading math of the wavefront path tracer (gfx950).
//
// HIP restatement of what the reference's Embree backend computes per hit; every function
// cites the reference file:line whose behaviour it reproduces (paths relative to the
// ChameleonRT tree, backends/embree/ unless stated). The whole library is compiled with
// -ffp-contract=off and without fast-math, so + - * / sqrt round exactly as on the CPU and
// the evaluation order written here (left to right, like ISPC) is the evaluation order
// executed. Only the libm transcendentals (pow, log, sin, cos, atan2, acos) differ from a
// CPU build by a few ulp.
#pragma once
// Timing experiments that render a WRONG image (CRT_EXP_SHADE_*: what a part of k_shade costs, by leaving it out) must never reach
// a product build through a stray -D: they compile only together with an explicit -DCRT_EXPERIMENTS=1 (tools/variants.py builds).
#if (defined(CRT_EXP_SHADE_ONE_MAT) || defined(CRT_EXP_SHADE_NO_UV) || defined(CRT_EXP_SHADE_NO_NEE) || defined(CRT_EXP_SHADE_NO_SAMPLE) || \
     defined(CRT_EXP_SHADE_NO_TEX)) && !defined(CRT_EXPERIMENTS)
#error "CRT_EXP_SHADE_* builds render wrong images (timing experiments only): define CRT_EXPERIMENTS=1 as well"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "crt_types.h"

namespace crt {

#define CRT_DEV __device__ __forceinline__

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
};

CRT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
CRT_DEV V3 v3(float c) { return V3{c, c, c}; }
CRT_DEV V2 v2(float x, float y) { return V2{x, y}; }
CRT_DEV V3 ld3(const float *p) { return V3{p[0], p[1], p[2]}; }

// float3.ih:82-202 (component-wise operators)
CRT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
CRT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
CRT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
CRT_DEV V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
CRT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
CRT_DEV V3 operator*(float s, V3 a) { return a * s; }
CRT_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
CRT_DEV V3 operator+(V3 a, float s) { return v3(a.x + s, a.y + s, a.z + s); }
CRT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
CRT_DEV V2 operator*(float s, V2 a) { return v2(a.x * s, a.y * s); }
CRT_DEV V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
CRT_DEV V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
CRT_DEV V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
CRT_DEV V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }

CRT_DEV float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // float3.ih:96-98
CRT_DEV float len3(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); } // float3.ih:60-62
// float3.ih:64-72; the `l < 0` guard there never fires (quirk Q6), so none here.
CRT_DEV V3 unit(V3 v)
{
    const float c = 1.f / len3(v);
    return v3(v.x * c, v.y * c, v.z * c);
}
CRT_DEV V3 cross3(V3 a, V3 b) // float3.ih:74-80
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
CRT_DEV bool is_black(V3 v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f; } // float3.ih:86-88

constexpr float kPi = 3.14159265358979323846f;      // util.ih:6
constexpr float kInvPi = 0.318309886183790671538f;  // util.ih:7

CRT_DEV float sq(float x) { return x * x; }                                  // util.ih:28-30
CRT_DEV float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }         // util.ih:59-61
CRT_DEV float mix1(float x, float y, float s) { return x * (1.f - s) + y * s; } // util.ih:63-65
CRT_DEV V3 mix3(V3 x, V3 y, float s) { return x * (1.f - s) + y * s; }          // util.ih:67-69
CRT_DEV float luma(V3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; } // util.ih:24-26

// util.ih:17-22
CRT_DEV float linear_to_srgb(float x)
{
    if (x <= 0.0031308f) {
        return 12.92f * x;
    }
    return 1.055f * powf(x, 1.f / 2.4f) - 0.055f;
}
// 8-bit sRGB as every spelled-out backend does it (embree_sycl/render_embree_kernel.inl:312-315);
// the ISPC stdlib float_to_srgb8 the Embree backend calls is not part of the reference tree.
CRT_DEV uint32_t srgb8(float x)
{
    const float s = 255.f * linear_to_srgb(x);
    return (uint32_t)fminf(fmaxf(s, 0.f), 255.f);
}

// util.ih:32-46
CRT_DEV void ortho_basis(V3 &v_x, V3 &v_y, V3 n)
{
    v_y = v3(0.f);
    if (n.x < 0.6f && n.x > -0.6f) {
        v_y.x = 1.f;
    } else if (n.y < 0.6f && n.y > -0.6f) {
        v_y.y = 1.f;
    } else if (n.z < 0.6f && n.z > -0.6f) {
        v_y.z = 1.f;
    } else {
        v_y.x = 1.f;
    }
    v_x = unit(cross3(v_y, n));
    v_y = unit(cross3(n, v_x));
}
// util.ih:48-56
CRT_DEV int wrap_mod(int a, int b)
{
    if (b == 0) {
        b = 1;
    }
    const int r = a - (a / b) * b;
    return r < 0 ? r + b : r;
}
CRT_DEV V3 reflect3(V3 i, V3 n) { return i - 2.f * n * dot3(i, n); } // util.ih:71-73
CRT_DEV V3 refract3(V3 i, V3 n, float eta)                           // util.ih:75-82
{
    const float n_dot_i = dot3(n, i);
    const float k = 1.f - eta * eta * (1.f - n_dot_i * n_dot_i);
    if (k < 0.f) {
        return v3(0.f);
    }
    return eta * i - (eta * n_dot_i + sqrtf(k)) * n;
}

// ---- RNG: lcg_rng.ih:4-59 (integer-exact) ------------------------------------------------
CRT_DEV uint32_t murmur_mix(uint32_t hash, uint32_t k)
{
    k *= 0xcc9e2d51u;
    k = (k << 15) | (k >> 17);
    k *= 0x1b873593u;
    hash ^= k;
    return ((hash << 13) | (hash >> 19)) * 5u + 0xe6546b64u;
}
CRT_DEV uint32_t murmur_finalize(uint32_t hash)
{
    hash ^= hash >> 16;
    hash *= 0x85ebca6bu;
    hash ^= hash >> 13;
    hash *= 0xc2b2ae35u;
    hash ^= hash >> 16;
    return hash;
}
CRT_DEV uint32_t rng_seed(uint32_t pixel_id, uint32_t frame_key)
{
    return murmur_finalize(murmur_mix(murmur_mix(0u, pixel_id), frame_key));
}
CRT_DEV uint32_t rng_next(uint32_t &state)
{
    state = state * 1664525u + 1013904223u;
    return state;
}
// ldexp((float)u32, -32): the int->float conversion rounds to nearest, the scale is exact;
// the result can be exactly 1.0f (quirk Q2).
CRT_DEV float rng_nextf(uint32_t &state) { return (float)rng_next(state) * 2.3283064365386963e-10f; }

// ---- textures: texture2d.ih:13-83, util/texture_channel_mask.h:16-23 ----------------------
// An 8-bit texel as the reference reads it: byte / 255.f (texture2d.ih:26-35). An IEEE division is ~10 instructions and a
// bilinear fetch of an RGB texture makes twelve of them, so the 256 possible quotients are computed ONCE per block --
// by that very division -- into LDS (unorm8_init, then a barrier) and looked up: same bits, one ds_read per channel.
// Every kernel that samples textures (k_shade, k_kat) calls unorm8_init() first.
static __shared__ float crt_unorm8_lut[256];
CRT_DEV void unorm8_init()
{
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) {
        crt_unorm8_lut[i] = (float)(int)i / 255.f;
    }
    __syncthreads();
}
#ifndef CRT_UNORM8_LUT
#define CRT_UNORM8_LUT 1
#endif
CRT_DEV float unorm8(uint32_t byte)
{
#if CRT_UNORM8_LUT
    return crt_unorm8_lut[byte & 0xffu];
#else
    return (float)(int)(byte & 0xffu) / 255.f;
#endif
}
CRT_DEV float texel_channel(const SceneView &sc, const TexRec &t, int px, int py, int channel)
{
    return unorm8(sc.texels[(size_t)t.offset16 * 16 + (size_t)tex_slot(t.width, px, py) * t.channels + channel]);
}
CRT_DEV V4 texel_rgba(const SceneView &sc, const TexRec &t, int px, int py)
{
    const uint8_t *p = sc.texels + (size_t)t.offset16 * 16 + (size_t)tex_slot(t.width, px, py) * t.channels;
    V4 c{0.f, 0.f, 0.f, 0.f};
    c.x = unorm8(p[0]);
    if (t.channels >= 2) {
        c.y = unorm8(p[1]);
    }
    if (t.channels >= 3) {
        c.z = unorm8(p[2]);
    }
    if (t.channels == 4) {
        c.w = unorm8(p[3]);
    }
    return c;
}
struct BilinearTaps {
    int x0, x1, y0, y1;
    float tx, ty;
};
// texture2d.ih:40-49: weights from floor(), texel ids from float->int truncation (quirk Q12)
CRT_DEV BilinearTaps bilinear_taps(const TexRec &t, V2 uv)
{
    const float ux = uv.x * t.width - 0.5f;
    const float uy = uv.y * t.height - 0.5f;
    BilinearTaps b;
    b.tx = ux - floorf(ux);
    b.ty = uy - floorf(uy);
    b.x0 = wrap_mod((int)ux, t.width);
    b.x1 = wrap_mod((int)(ux + 1), t.width);
    b.y0 = wrap_mod((int)uy, t.height);
    b.y1 = wrap_mod((int)(uy + 1), t.height);
    return b;
}
CRT_DEV V4 sample_rgba(const SceneView &sc, const TexRec &t, V2 uv)
{
    const BilinearTaps b = bilinear_taps(t, uv);
    const V4 s00 = texel_rgba(sc, t, b.x0, b.y0);
    const V4 s10 = texel_rgba(sc, t, b.x1, b.y0);
    const V4 s01 = texel_rgba(sc, t, b.x0, b.y1);
    const V4 s11 = texel_rgba(sc, t, b.x1, b.y1);
    return s00 * (1.f - b.tx) * (1.f - b.ty) + s10 * b.tx * (1.f - b.ty) + s01 * (1.f - b.tx) * b.ty +
           s11 * b.tx * b.ty;
}
CRT_DEV float sample_channel(const SceneView &sc, const TexRec &t, V2 uv, int channel)
{
    const BilinearTaps b = bilinear_taps(t, uv);
    const float s00 = texel_channel(sc, t, b.x0, b.y0, channel);
    const float s10 = texel_channel(sc, t, b.x1, b.y0, channel);
    const float s01 = texel_channel(sc, t, b.x0, b.y1, channel);
    const float s11 = texel_channel(sc, t, b.x1, b.y1, channel);
    return s00 * (1.f - b.tx) * (1.f - b.ty) + s10 * b.tx * (1.f - b.ty) + s01 * (1.f - b.tx) * b.ty +
           s11 * b.tx * b.ty;
}

// The four bilinear taps of a 4-CHANNEL texture fetched as whole texels: one dword request per tap carries
// every channel, where texel_rgba / texel_channel issue a byte request per channel and tap. unpack_material
// keeps the taps of the texture it fetched last, so a material whose metallic and roughness are two channels
// of one parameter map (the glTF convention, scene.cpp:383-397) pays for that texture once. Same values,
// same expressions as sample_rgba / sample_channel: byte / 255.f, then the four weighted terms left to right.
struct TexTaps {
    uint32_t id; // texture the taps belong to, 0xffffffff = none
    uint32_t t00, t10, t01, t11;
    float tx, ty;
};
CRT_DEV void fetch_taps4(const SceneView &sc, const TexRec &t, uint32_t id, V2 uv, TexTaps &c)
{
    const BilinearTaps b = bilinear_taps(t, uv);
    const uint32_t *texels = reinterpret_cast<const uint32_t *>(sc.texels + (size_t)t.offset16 * 16);
    const uint32_t tiles_x = tex_tiles_x(t.width);
    const uint32_t r0 = tex_row_part(tiles_x, b.y0), r1 = tex_row_part(tiles_x, b.y1);
    const uint32_t c0 = tex_col_part(b.x0), c1 = tex_col_part(b.x1);
    c.t00 = texels[r0 + c0];
    c.t10 = texels[r0 + c1];
    c.t01 = texels[r1 + c0];
    c.t11 = texels[r1 + c1];
    c.tx = b.tx;
    c.ty = b.ty;
    c.id = id;
}
CRT_DEV float taps_channel(const TexTaps &c, int channel)
{
    const int sh = 8 * channel;
    const float s00 = unorm8(c.t00 >> sh), s10 = unorm8(c.t10 >> sh);
    const float s01 = unorm8(c.t01 >> sh), s11 = unorm8(c.t11 >> sh);
    return s00 * (1.f - c.tx) * (1.f - c.ty) + s10 * c.tx * (1.f - c.ty) + s01 * (1.f - c.tx) * c.ty + s11 * c.tx * c.ty;
}

// ---- Disney BSDF: disney_bsdf.ih:19-429 ---------------------------------------------------
struct Surface { // DisneyMaterial after unpack_material
    V3 base_color;
    float metallic, specular, roughness, specular_tint, anisotropy, sheen, sheen_tint, clearcoat,
        clearcoat_gloss, ior, specular_transmission;
};

// render_embree.ispc:66-77
CRT_DEV float scalar_param(const SceneView &sc, float x, V2 uv, TexTaps &taps)
{
    const uint32_t mask = __float_as_uint(x);
#ifdef CRT_EXP_SHADE_NO_TEX // TIMING EXPERIMENT ONLY (wrong image): what the descriptor -> texel gathers cost
    if (mask & 0x80000000u) {
        return 0.5f;
    }
#endif
    if (mask & 0x80000000u) {
        const uint32_t id = mask & 0x1fffffffu;
        const int channel = (int)((mask >> 29) & 0x3u);
        if (id != taps.id) {
            const TexRec &t = sc.textures[id];
            if (t.channels != 4) {
                return sample_channel(sc, t, uv, channel);
            }
            fetch_taps4(sc, t, id, uv, taps);
        }
        return taps_channel(taps, channel);
    }
    return x;
}
// render_embree.ispc:79-103
CRT_DEV void unpack_material(const SceneView &sc, Surface &m, const float *p, V2 uv)
{
    TexTaps taps;
    taps.id = 0xffffffffu;
    taps.t00 = taps.t10 = taps.t01 = taps.t11 = 0u;
    taps.tx = taps.ty = 0.f;
    const uint32_t mask = __float_as_uint(p[0]);
#ifdef CRT_EXP_SHADE_NO_TEX
    if (mask & 0x80000000u) {
        m.base_color = v3(0.5f, 0.5f, 0.5f);
    } else
#endif
    if (mask & 0x80000000u) {
        const uint32_t id = mask & 0x1fffffffu;
        const TexRec &t = sc.textures[id];
        if (t.channels == 4) {
            fetch_taps4(sc, t, id, uv, taps);
            m.base_color = v3(taps_channel(taps, 0), taps_channel(taps, 1), taps_channel(taps, 2));
        } else {
            const V4 c = sample_rgba(sc, t, uv);
            m.base_color = v3(c.x, c.y, c.z);
        }
    } else {
        m.base_color = v3(p[0], p[1], p[2]);
    }
    m.metallic = scalar_param(sc, p[3], uv, taps);
    m.specular = scalar_param(sc, p[4], uv, taps);
    m.roughness = scalar_param(sc, p[5], uv, taps);
    m.specular_tint = scalar_param(sc, p[6], uv, taps);
    m.anisotropy = scalar_param(sc, p[7], uv, taps);
    m.sheen = scalar_param(sc, p[8], uv, taps);
    m.sheen_tint = scalar_param(sc, p[9], uv, taps);
    m.clearcoat = scalar_param(sc, p[10], uv, taps);
    m.clearcoat_gloss = scalar_param(sc, p[11], uv, taps);
    m.ior = scalar_param(sc, p[12], uv, taps);
    m.specular_transmission = scalar_param(sc, p[13], uv, taps);
}

CRT_DEV bool same_side(V3 w_o, V3 w_i, V3 n) { return dot3(w_o, n) * dot3(w_i, n) > 0.f; } // :38-40

// :44-62
CRT_DEV V3 cosine_hemisphere(V2 u)
{
    const V2 s = 2.f * u - v2(1.f, 1.f);
    V2 d;
    float radius = 0;
    float theta = 0;
    if (s.x == 0.f && s.y == 0.f) {
        d = s;
    } else {
        if (fabsf(s.x) > fabsf(s.y)) {
            radius = s.x;
            theta = kPi / 4.f * (s.y / s.x);
        } else {
            radius = s.y;
            theta = kPi / 2.f - kPi / 4.f * (s.x / s.y);
        }
    }
    d = radius * v2(cosf(theta), sinf(theta));
    return v3(d.x, d.y, sqrtf(fmaxf(0.f, 1.f - d.x * d.x - d.y * d.y)));
}
CRT_DEV V3 polar_dir(float sin_theta, float cos_theta, float phi) // :64-66
{
    return v3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
}
CRT_DEV float mis_power(float n_f, float pdf_f, float n_g, float pdf_g) // :68-72
{
    const float f = n_f * pdf_f;
    const float g = n_g * pdf_g;
    return (f * f) / (f * f + g * g);
}
// :74-76 `pow(clamp(1 - cos_theta, 0, 1), 5)`. The fifth power is formed by three multiplications, (x^2)^2 * x: within 1.5 ulp of
// the exact value, which is the error class of any pow() -- the reference's ISPC pow, the oracle's libm powf and the
// device library's powf already differ from each other in the last place (transcendentals are held
// to 2e-5, not to the bit) -- at 3 instructions instead of the ~60 of a general powf, eleven times per hit in k_shade.
#ifndef CRT_SCHLICK_POWF
#define CRT_SCHLICK_POWF 0
#endif
CRT_DEV float schlick(float cos_theta)
{
    const float x = clamp01(1.f - cos_theta);
#if CRT_SCHLICK_POWF
    return powf(x, 5.f);
#else
    const float x2 = x * x;
    return x2 * x2 * x;
#endif
}
CRT_DEV float fresnel_dielectric(float cos_theta_i, float eta_i, float eta_t)          // :82-89
{
    const float g = sq(eta_t) / sq(eta_i) - 1.f + sq(cos_theta_i);
    if (g < 0.f) {
        return 1.f;
    }
    return 0.5f * sq(g - cos_theta_i) / sq(g + cos_theta_i) *
           (1.f + sq(cos_theta_i * (g + cos_theta_i) - 1.f) / sq(cos_theta_i * (g - cos_theta_i) + 1.f));
}
CRT_DEV float gtr1(float cos_theta_h, float alpha) // :93-99
{
    if (alpha >= 1.f) {
        return kInvPi;
    }
    const float a2 = alpha * alpha;
    return kInvPi * (a2 - 1.f) / (logf(a2) * (1.f + (a2 - 1.f) * cos_theta_h * cos_theta_h));
}
CRT_DEV float gtr2(float cos_theta_h, float alpha) // :103-106
{
    const float a2 = alpha * alpha;
    return kInvPi * a2 / sq(1.f + (a2 - 1.f) * cos_theta_h * cos_theta_h);
}
CRT_DEV float gtr2_aniso(float h_dot_n, float h_dot_x, float h_dot_y, V2 alpha) // :110-113
{
    return kInvPi / (alpha.x * alpha.y * sq(sq(h_dot_x / alpha.x) + sq(h_dot_y / alpha.y) + h_dot_n * h_dot_n));
}
CRT_DEV float smith_ggx(float n_dot_o, float alpha_g) // :115-119
{
    const float a = alpha_g * alpha_g;
    const float b = n_dot_o * n_dot_o;
    return 1.f / (n_dot_o + sqrtf(a + b - a * b));
}
CRT_DEV float smith_ggx_aniso(float n_dot_o, float o_dot_x, float o_dot_y, V2 alpha) // :121-123
{
    return 1.f / (n_dot_o + sqrtf(sq(o_dot_x * alpha.x) + sq(o_dot_y * alpha.y) + sq(n_dot_o)));
}
CRT_DEV V3 to_world(V3 local, V3 n, V3 v_x, V3 v_y) { return local.x * v_x + local.y * v_y + local.z * n; }

CRT_DEV V3 sample_lambert_dir(V3 n, V3 v_x, V3 v_y, V2 s) // :126-129
{
    return to_world(unit(cosine_hemisphere(s)), n, v_x, v_y);
}
CRT_DEV V3 sample_gtr1_h(V3 n, V3 v_x, V3 v_y, float alpha, V2 s) // :132-140
{
    const float phi_h = 2.f * kPi * s.x;
    const float a2 = alpha * alpha;
    const float cos2 = (1.f - powf(a2, 1.f - s.y)) / (1.f - a2);
    const float cos_theta_h = sqrtf(cos2);
    const float sin_theta_h = sqrtf(1.f - cos2);
    return to_world(unit(polar_dir(sin_theta_h, cos_theta_h, phi_h)), n, v_x, v_y);
}
CRT_DEV V3 sample_gtr2_h(V3 n, V3 v_x, V3 v_y, float alpha, V2 s) // :142-149
{
    const float phi_h = 2.f * kPi * s.x;
    const float cos2 = (1.f - s.y) / (1.f + (alpha * alpha - 1.f) * s.y);
    const float cos_theta_h = sqrtf(cos2);
    const float sin_theta_h = sqrtf(1.f - cos2);
    return to_world(unit(polar_dir(sin_theta_h, cos_theta_h, phi_h)), n, v_x, v_y);
}
CRT_DEV V3 sample_gtr2_aniso_h(V3 n, V3 v_x, V3 v_y, V2 alpha, V2 s) // :151-155
{
    const float x = 2.f * kPi * s.x;
    const V3 w_h = sqrtf(s.y / (1.f - s.y)) * (alpha.x * cosf(x) * v_x + alpha.y * sinf(x) * v_y) + n;
    return unit(w_h);
}
CRT_DEV float lambert_pdf(V3 w_i, V3 n) // :157-163
{
    const float d = dot3(w_i, n);
    return d > 0.f ? d * kInvPi : 0.f;
}
CRT_DEV float gtr1_pdf(V3 w_o, V3 w_i, V3 n, float alpha) // :165-173
{
    if (!same_side(w_o, w_i, n)) {
        return 0.f;
    }
    const V3 w_h = unit(w_i + w_o);
    const float cos_theta_h = dot3(n, w_h);
    const float d = gtr1(cos_theta_h, alpha);
    return d * cos_theta_h / (4.f * dot3(w_o, w_h));
}
CRT_DEV float gtr2_pdf(V3 w_o, V3 w_i, V3 n, float alpha) // :175-183
{
    if (!same_side(w_o, w_i, n)) {
        return 0.f;
    }
    const V3 w_h = unit(w_i + w_o);
    const float cos_theta_h = dot3(n, w_h);
    const float d = gtr2(cos_theta_h, alpha);
    return d * cos_theta_h / (4.f * dot3(w_o, w_h));
}
CRT_DEV float gtr2_transmission_pdf(V3 w_o, V3 w_i, V3 n, float alpha, float ior) // :185-201
{
    if (same_side(w_o, w_i, n)) {
        return 0.f;
    }
    const bool entering = dot3(w_o, n) > 0.f;
    const float eta_o = entering ? 1.f : ior;
    const float eta_i = entering ? ior : 1.f;
    const V3 w_h = unit(w_o + w_i * eta_i / eta_o);
    const float cos_theta_h = fabsf(dot3(n, w_h));
    const float i_dot_h = dot3(w_i, w_h);
    const float o_dot_h = dot3(w_o, w_h);
    const float d = gtr2(cos_theta_h, alpha);
    const float dwh_dwi = o_dot_h * sq(eta_o) / sq(eta_o * o_dot_h + eta_i * i_dot_h);
    return d * cos_theta_h * fabsf(dwh_dwi);
}
CRT_DEV float gtr2_aniso_pdf(V3 w_o, V3 w_i, V3 n, V3 v_x, V3 v_y, V2 alpha) // :203-213
{
    if (!same_side(w_o, w_i, n)) {
        return 0.f;
    }
    const V3 w_h = unit(w_i + w_o);
    const float cos_theta_h = dot3(n, w_h);
    const float d = gtr2_aniso(cos_theta_h, fabsf(dot3(w_h, v_x)), fabsf(dot3(w_h, v_y)), alpha);
    return d * cos_theta_h / (4.f * dot3(w_o, w_h));
}
CRT_DEV V3 lobe_diffuse(const Surface &m, V3 n, V3 w_o, V3 w_i) // :215-226
{
    const V3 w_h = unit(w_i + w_o);
    const float n_dot_o = fabsf(dot3(w_o, n));
    const float n_dot_i = fabsf(dot3(w_i, n));
    const float i_dot_h = dot3(w_i, w_h);
    const float fd90 = 0.5f + 2.f * m.roughness * i_dot_h * i_dot_h;
    const float fi = schlick(n_dot_i);
    const float fo = schlick(n_dot_o);
    return m.base_color * kInvPi * mix1(1.f, fd90, fi) * mix1(1.f, fd90, fo);
}
CRT_DEV V3 specular_f0(const Surface &m) // :232-234 and :275-277
{
    const float lum = luma(m.base_color);
    const V3 tint = lum > 0.f ? m.base_color / lum : v3(1.f);
    return mix3(m.specular * 0.08f * mix3(v3(1.f), tint, m.specular_tint), m.base_color, m.metallic);
}
CRT_DEV V3 lobe_microfacet_iso(const Surface &m, V3 n, V3 w_o, V3 w_i) // :228-241
{
    const V3 w_h = unit(w_i + w_o);
    const V3 spec = specular_f0(m);
    const float alpha = fmaxf(0.001f, m.roughness * m.roughness);
    const float d = gtr2(dot3(n, w_h), alpha);
    const V3 f = mix3(spec, v3(1.f), schlick(dot3(w_i, w_h)));
    const float g = smith_ggx(dot3(n, w_i), alpha) * smith_ggx(dot3(n, w_o), alpha);
    return d * f * g;
}
CRT_DEV V3 lobe_transmission_iso(const Surface &m, V3 n, V3 w_o, V3 w_i) // :243-269
{
    const float o_dot_n = dot3(w_o, n);
    const float i_dot_n = dot3(w_i, n);
    if (o_dot_n == 0.f || i_dot_n == 0.f) {
        return v3(0.f);
    }
    const bool entering = o_dot_n > 0.f;
    const float eta_o = entering ? 1.f : m.ior;
    const float eta_i = entering ? m.ior : 1.f;
    const V3 w_h = unit(w_o + w_i * eta_i / eta_o);
    const float alpha = fmaxf(0.001f, m.roughness * m.roughness);
    const float d = gtr2(fabsf(dot3(n, w_h)), alpha);
    const float f = fresnel_dielectric(fabsf(dot3(w_i, n)), eta_o, eta_i);
    const float g = smith_ggx(fabsf(dot3(n, w_i)), alpha) * smith_ggx(fabsf(dot3(n, w_o)), alpha);
    const float i_dot_h = dot3(w_i, w_h);
    const float o_dot_h = dot3(w_o, w_h);
    const float c = fabsf(o_dot_h) / fabsf(dot3(w_o, n)) * fabsf(i_dot_h) / fabsf(dot3(w_i, n)) * sq(eta_o) /
                    sq(eta_o * o_dot_h + eta_i * i_dot_h);
    return m.base_color * c * (1.f - f) * g * d;
}
CRT_DEV V3 lobe_microfacet_aniso(const Surface &m, V3 n, V3 w_o, V3 w_i, V3 v_x, V3 v_y) // :271-287
{
    const V3 w_h = unit(w_i + w_o);
    const V3 spec = specular_f0(m);
    const float aspect = sqrtf(1.f - m.anisotropy * 0.9f);
    const float a = m.roughness * m.roughness;
    const V2 alpha = v2(fmaxf(0.001f, a / aspect), fmaxf(0.001f, a * aspect));
    const float d = gtr2_aniso(dot3(n, w_h), fabsf(dot3(w_h, v_x)), fabsf(dot3(w_h, v_y)), alpha);
    const V3 f = mix3(spec, v3(1.f), schlick(dot3(w_i, w_h)));
    const float g = smith_ggx_aniso(dot3(n, w_i), fabsf(dot3(w_i, v_x)), fabsf(dot3(w_i, v_y)), alpha) *
                    smith_ggx_aniso(dot3(n, w_o), fabsf(dot3(w_o, v_x)), fabsf(dot3(w_o, v_y)), alpha);
    return d * f * g;
}
CRT_DEV float lobe_clearcoat(const Surface &m, V3 n, V3 w_o, V3 w_i) // :289-298
{
    const V3 w_h = unit(w_i + w_o);
    const float alpha = mix1(0.1f, 0.001f, m.clearcoat_gloss);
    const float d = gtr1(dot3(n, w_h), alpha);
    const float f = mix1(0.04f, 1.f, schlick(dot3(w_i, n)));
    const float g = smith_ggx(dot3(n, w_i), 0.25f) * smith_ggx(dot3(n, w_o), 0.25f);
    return 0.25f * m.clearcoat * d * f * g;
}
CRT_DEV V3 lobe_sheen(const Surface &m, V3 n, V3 w_i) // :300-309
{
    const float lum = luma(m.base_color);
    const V3 tint = lum > 0.f ? m.base_color / lum : v3(1.f);
    const V3 sheen_color = mix3(v3(1.f), tint, m.sheen_tint);
    const float f = schlick(dot3(w_i, n));
    return f * m.sheen * sheen_color;
}
CRT_DEV V3 disney_eval(const Surface &m, V3 n, V3 w_o, V3 w_i, V3 v_x, V3 v_y) // :311-332
{
    if (!same_side(w_o, w_i, n)) {
        if (m.specular_transmission > 0.f) {
            const V3 spec_trans = lobe_transmission_iso(m, n, w_o, w_i);
            return spec_trans * (1.f - m.metallic) * m.specular_transmission;
        }
        return v3(0.f);
    }
    const float coat = lobe_clearcoat(m, n, w_o, w_i);
    const V3 sheen = lobe_sheen(m, n, w_i);
    const V3 diffuse = lobe_diffuse(m, n, w_o, w_i);
    V3 gloss;
    if (m.anisotropy == 0.f) {
        gloss = lobe_microfacet_iso(m, n, w_o, w_i);
    } else {
        gloss = lobe_microfacet_aniso(m, n, w_o, w_i, v_x, v_y);
    }
    return (diffuse + sheen) * (1.f - m.metallic) * (1.f - m.specular_transmission) + gloss + coat;
}
CRT_DEV float disney_pdf(const Surface &m, V3 n, V3 w_o, V3 w_i, V3 v_x, V3 v_y) // :334-359
{
    const float alpha = fmaxf(0.001f, m.roughness * m.roughness);
    const float aspect = sqrtf(1.f - m.anisotropy * 0.9f);
    const V2 alpha_aniso = v2(fmaxf(0.001f, alpha / aspect), fmaxf(0.001f, alpha * aspect));
    const float clearcoat_alpha = mix1(0.1f, 0.001f, m.clearcoat_gloss);
    const float diffuse = lambert_pdf(w_i, n);
    const float clear_coat = gtr1_pdf(w_o, w_i, n, clearcoat_alpha);
    float n_comp = 3.f;
    float microfacet;
    float microfacet_transmission = 0.f;
    if (m.anisotropy == 0.f) {
        microfacet = gtr2_pdf(w_o, w_i, n, alpha);
    } else {
        microfacet = gtr2_aniso_pdf(w_o, w_i, n, v_x, v_y, alpha_aniso);
    }
    if (m.specular_transmission > 0.f) {
        n_comp = 4.f;
        microfacet_transmission = gtr2_transmission_pdf(w_o, w_i, n, alpha, m.ior);
    }
    return (diffuse + microfacet + microfacet_transmission + clear_coat) / n_comp;
}
// :364-426, the direction-sampling half of sample_disney_brdf. Draw order: lobe pick, sample x,
// sample y (quirk Q3). Returns false where the reference returns (f = 0, pdf = 0): invalid
// reflection / total internal reflection. Splitting it from the evaluation lets the caller skip
// disney_pdf + disney_eval when their result is provably unused (both are pure functions).
CRT_DEV bool disney_sample_dir(const Surface &m, V3 n, V3 w_o, V3 v_x, V3 v_y, uint32_t &rng, V3 &w_i)
{
    int component;
    if (m.specular_transmission == 0.f) {
        component = (int)(rng_nextf(rng) * 3.f);
        component = min(max(component, 0), 2);
    } else {
        component = (int)(rng_nextf(rng) * 4.f);
        component = min(max(component, 0), 3);
    }
    V2 s;
    s.x = rng_nextf(rng);
    s.y = rng_nextf(rng);
    if (component == 0) {
        w_i = sample_lambert_dir(n, v_x, v_y, s);
    } else if (component == 1) {
        V3 w_h;
        const float alpha = fmaxf(0.001f, m.roughness * m.roughness);
        if (m.anisotropy == 0.f) {
            w_h = sample_gtr2_h(n, v_x, v_y, alpha, s);
        } else {
            const float aspect = sqrtf(1.f - m.anisotropy * 0.9f);
            const V2 alpha_aniso = v2(fmaxf(0.001f, alpha / aspect), fmaxf(0.001f, alpha * aspect));
            w_h = sample_gtr2_aniso_h(n, v_x, v_y, alpha_aniso, s);
        }
        w_i = reflect3(-w_o, w_h);
        if (!same_side(w_o, w_i, n)) {
            w_i = v3(0.f);
            return false;
        }
    } else if (component == 2) {
        const float alpha = mix1(0.1f, 0.001f, m.clearcoat_gloss);
        const V3 w_h = sample_gtr1_h(n, v_x, v_y, alpha, s);
        w_i = reflect3(-w_o, w_h);
        if (!same_side(w_o, w_i, n)) {
            w_i = v3(0.f);
            return false;
        }
    } else {
        const float alpha = fmaxf(0.001f, m.roughness * m.roughness);
        V3 w_h = sample_gtr2_h(n, v_x, v_y, alpha, s);
        if (dot3(w_o, w_h) < 0.f) {
            w_h = -w_h;
        }
        const bool entering = dot3(w_o, n) > 0.f;
        w_i = refract3(-w_o, w_h, entering ? 1.f / m.ior : m.ior);
        if (is_black(w_i)) {
            return false;
        }
    }
    return true;
}
// :364-429 sample_disney_brdf = direction + (:427-428) pdf and value at that direction
CRT_DEV V3 disney_sample(const Surface &m, V3 n, V3 w_o, V3 v_x, V3 v_y, uint32_t &rng, V3 &w_i, float &pdf)
{
    if (!disney_sample_dir(m, n, w_o, v_x, v_y, rng, w_i)) {
        pdf = 0.f;
        return v3(0.f);
    }
    pdf = disney_pdf(m, n, w_o, w_i, v_x, v_y);
    return disney_eval(m, n, w_o, w_i, v_x, v_y);
}

// ---- quad lights: lights.ih:7-69, host layout util/lights.h:6-18 (20 floats) --------------
struct QuadLight {
    V3 emission, position, normal, v_x, v_y;
    float width, height;
};
CRT_DEV QuadLight load_light(const float *l)
{
    QuadLight q;
    q.emission = ld3(l);
    q.position = ld3(l + 4);
    q.normal = ld3(l + 8);
    q.v_x = ld3(l + 12);
    q.width = l[15];
    q.v_y = ld3(l + 16);
    q.height = l[19];
    return q;
}
CRT_DEV V3 light_sample_position(const QuadLight &l, V2 s) // lights.ih:26-30
{
    return s.x * l.v_x * l.width + s.y * l.v_y * l.height + l.position;
}
// lights.ih:35-48; `p - dir`, not `p - orig` (quirk Q4)
CRT_DEV float light_pdf(const QuadLight &l, V3 p, V3 dir)
{
    const float surface_area = l.width * l.height;
    const V3 to_pt = p - dir;
    const float dist_sqr = dot3(to_pt, to_pt);
    const float n_dot_w = dot3(l.normal, -dir);
    if (n_dot_w < RAY_EPS) {
        return 0.f;
    }
    return dist_sqr / (n_dot_w * surface_area);
}
// lights.ih:50-69 (quirk Q5)
CRT_DEV bool light_intersect(const QuadLight &l, V3 orig, V3 dir, float &t, V3 &light_pos)
{
    const float denom = dot3(dir, l.normal);
    if (denom != 0.f) {
        t = dot3(l.position - orig, l.normal) / denom;
        if (t < 0.f) {
            return false;
        }
        light_pos = orig + dir * t;
        const V3 hit_v = light_pos - l.position;
        if (fabsf(dot3(hit_v, l.v_x)) < l.width && fabsf(dot3(hit_v, l.v_y)) < l.height) {
            return true;
        }
    }
    return false;
}

// render_embree.ispc:184-196
CRT_DEV V3 miss_color(V3 dir)
{
    const float u = (1.f + atan2f(dir.x, -dir.z) * kInvPi) * 0.5f;
    const float v = acosf(dir.y) * kInvPi;
    const int check_x = (int)(u * 10.f);
    const int check_y = (int)(v * 10.f);
    if (dir.y > -0.1f && wrap_mod(check_x + check_y, 2) == 0) {
        return v3(0.5f);
    }
    return v3(0.1f);
}

// Russian roulette of the path loop, render_embree.ispc:327-335 (applied by the caller once bounce > 3): returns true if the path
// ends; otherwise the throughput is divided by the survival probability. `max` is the reference's own two-operand select --
// sycl::max(x, y) = x < y ? y : x in the kernel the oracle is pinned to (embree_sycl/render_embree_kernel.inl:284-287), std::max in
// the oracle -- and NOT fmaxf: with a NaN in throughput.x and finite y, z the select keeps the NaN (q = 0.05), fmaxf would drop it
// (q = 1 - max(y, z)). Non-finite throughputs do occur (glass at grazing angles: inf * 0); CRT_KAT_ROULETTE holds this to the bit.
CRT_DEV float ref_max(float a, float b) { return a < b ? b : a; }
CRT_DEV bool russian_roulette(V3 &tp, uint32_t &rng, float *q_out = nullptr)
{
    const float q = ref_max(0.05f, 1.f - ref_max(tp.x, ref_max(tp.y, tp.z)));
    if (q_out) {
        *q_out = q;
    }
    if (rng_nextf(rng) < q) {
        return true;
    }
    tp = tp / (1.f - q);
    return false;
}

} // namespace crt
