// ref_driver.cpp — runs the REFERENCE's own path-tracing kernel on the CPU, to pin the oracle.
//
// TEST INFRASTRUCTURE ONLY (built by `make -C oracle _ref` into oracle/_ref/libref.so, which is
// git-ignored; loaded only by tests/ and tests/golden/make_ref_golden.py).
//
// The reference's Embree backend is ISPC + Embree 4 + TBB and cannot be built here. Its SYCL twin,
// backends/embree_sycl, carries the same per-pixel kernel as plain C++:
//   render_embree_kernel.inl  trace_ray (camera ray, path loop, NEE + MIS, Russian roulette,
//                             running mean, sRGB8), sample_direct_light, unpack_material, miss_shader
//   disney_bsdf.h lights.h lcg_rng.h texture2d.h util.h float3.h mat4.h
// (function for function the ISPC files of backends/embree). This file #includes that
// kernel FROM WHERE IT LIES under /root/reference — nothing is copied into the repo — and compiles
// it with g++ against three stand-in headers for the third-party APIs it needs (oracle/ref_shim/:
// SYCL math = libm, GLM PODs, Embree's ray/hit records). The only thing substituted is Embree
// itself: rtcIntersect1 / rtcOccluded1 below are the oracle's documented ray/triangle rule
// (orc_intersect1 / orc_occluded1), so that every difference between oracle and reference output isolates to the
// part the reference DOES define.
//
// What is restated here (host code of backends/embree_sycl/render_embree.cpp, which needs SYCL,
// TBB and GLM to compile): the camera basis (:169-180), the sRGB -> linear pass over 8-bit textures
// (:98-114 with util/util.cpp:102-108) and the frame loop calling trace_ray for every pixel.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include <CL/sycl.hpp> // oracle/ref_shim: the reference's includer (render_embree.cpp) pulls SYCL in first

#define REPORT_RAY_STATS
#include "render_embree_kernel.inl" // -I/root/reference/backends/embree_sycl

#include "crt_oracle.h"

namespace {

struct StandinScene {
    orc_scene *orc = nullptr;
    const crt_scene_desc *desc = nullptr;
};

inline StandinScene *standin(RTCScene s) { return reinterpret_cast<StandinScene *>(s); }

void cross3(const float a[3], const float b[3], float out[3])
{
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

} // namespace

// ---- the Embree stand-in --------------------------------------------------------------------
void rtcIntersect1(RTCScene scene, RTCRayHit *rh, RTCIntersectArguments *)
{
    StandinScene *s = standin(scene);
    const float org[3] = {rh->ray.org_x, rh->ray.org_y, rh->ray.org_z};
    const float dir[3] = {rh->ray.dir_x, rh->ray.dir_y, rh->ray.dir_z};
    float t, u, v;
    int32_t inst, geom, prim;
    if (!orc_intersect1(s->orc, org, dir, rh->ray.tnear, rh->ray.tfar, &t, &u, &v, &inst, &geom, &prim)) {
        return; // miss: ids stay RTC_INVALID_GEOMETRY_ID
    }
    rh->ray.tfar = t;
    rh->hit.u = u;
    rh->hit.v = v;
    rh->hit.instID[0] = (unsigned int)inst;
    rh->hit.geomID = (unsigned int)geom;
    rh->hit.primID = (unsigned int)prim;
    // Ng = cross(v2 - v0, v0 - v1), unnormalised, in the instance's object space (Embree's triangle
    // convention)
    const crt_scene_desc *d = s->desc;
    const crt_parameterized_mesh_desc &pm = d->parameterized_meshes[d->instances[inst].parameterized_mesh_id];
    const crt_geometry_desc &g = d->geometries[d->meshes[pm.mesh_id].first_geometry + (uint32_t)geom];
    const float *v0 = g.vertices + 3 * (size_t)g.indices[3 * (size_t)prim];
    const float *v1 = g.vertices + 3 * (size_t)g.indices[3 * (size_t)prim + 1];
    const float *v2 = g.vertices + 3 * (size_t)g.indices[3 * (size_t)prim + 2];
    const float e1[3] = {v0[0] - v1[0], v0[1] - v1[1], v0[2] - v1[2]};
    const float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    float ng[3];
    cross3(e2, e1, ng);
    rh->hit.Ng_x = ng[0];
    rh->hit.Ng_y = ng[1];
    rh->hit.Ng_z = ng[2];
}

void rtcOccluded1(RTCScene scene, RTCRay *ray, RTCOccludedArguments *)
{
    StandinScene *s = standin(scene);
    const float org[3] = {ray->org_x, ray->org_y, ray->org_z};
    const float dir[3] = {ray->dir_x, ray->dir_y, ray->dir_z};
    if (orc_occluded1(s->orc, org, dir, ray->tnear, ray->tfar)) {
        ray->tfar = -std::numeric_limits<float>::infinity();
    }
}

// ---- C API ---------------------------------------------------------------------------------------
extern "C" {

// Render `n_frames` accumulated frames of `desc` with the reference kernel. accum: w*h*3 floats,
// framebuffer: w*h*4 bytes, ray_stats: w*h uint16 of the LAST frame (like the reference's buffer).
int ref_render(const crt_scene_desc *desc, int fb_w, int fb_h, const float pos[3], const float dir[3], const float up[3],
               float fovy_deg, int n_frames, float *accum, uint8_t *framebuffer, uint16_t *ray_stats)
{
    if (desc == nullptr || fb_w <= 0 || fb_h <= 0 || desc->n_lights == 0) {
        return -1;
    }
    StandinScene st;
    st.desc = desc;
    st.orc = orc_scene_create(desc);
    if (st.orc == nullptr) {
        return -2;
    }
    // geometry / instance tables in the kernel's layout (embree_utils.h ISPCGeometry, ISPCInstance)
    std::vector<embree::ISPCGeometry> geoms(desc->n_geometries);
    for (uint32_t g = 0; g < desc->n_geometries; ++g) {
        geoms[g].vertex_buf = reinterpret_cast<const glm::vec3 *>(desc->geometries[g].vertices);
        geoms[g].index_buf = reinterpret_cast<const glm::uvec3 *>(desc->geometries[g].indices);
        geoms[g].uv_buf = reinterpret_cast<const glm::vec2 *>(desc->geometries[g].uvs);
    }
    std::vector<embree::ISPCInstance> insts(desc->n_instances);
    for (uint32_t i = 0; i < desc->n_instances; ++i) {
        const crt_parameterized_mesh_desc &pm = desc->parameterized_meshes[desc->instances[i].parameterized_mesh_id];
        insts[i].geometries = geoms.data() + desc->meshes[pm.mesh_id].first_geometry;
        std::memcpy(insts[i].object_to_world, desc->instances[i].transform, sizeof(float) * 16);
        // glm::inverse (third-party, absent here) -> the oracle's restatement of GLM 0.9.9.8's compute_inverse<4,4>, so both sides shade with the same matrices
        if (!orc_invert4x4(desc->instances[i].transform, insts[i].world_to_object)) {
            orc_scene_destroy(st.orc);
            return -3;
        }
        insts[i].material_ids = pm.material_ids;
    }
    std::vector<embree::MaterialParams> mats(desc->n_materials);
    for (uint32_t m = 0; m < desc->n_materials; ++m) {
        const float *p = desc->materials + 16 * (size_t)m;
        mats[m].base_color = glm::vec3(p[0], p[1], p[2]);
        mats[m].metallic = p[3];
        mats[m].specular = p[4];
        mats[m].roughness = p[5];
        mats[m].specular_tint = p[6];
        mats[m].anisotropy = p[7];
        mats[m].sheen = p[8];
        mats[m].sheen_tint = p[9];
        mats[m].clearcoat = p[10];
        mats[m].clearcoat_gloss = p[11];
        mats[m].ior = p[12];
        mats[m].specular_transmission = p[13];
    }
    // textures: sRGB ones linearised in 8 bits on the host (render_embree.cpp:98-114, util.cpp:102-108)
    std::vector<std::vector<uint8_t>> texels(desc->n_textures);
    std::vector<embree::ISPCTexture2D> textures(desc->n_textures);
    for (uint32_t t = 0; t < desc->n_textures; ++t) {
        const crt_image_desc &im = desc->textures[t];
        texels[t].assign(im.data, im.data + (size_t)im.width * im.height * im.channels);
        if (im.color_space != CRT_COLORSPACE_LINEAR) {
            const int convert_channels = im.channels < 3 ? im.channels : 3;
            for (size_t px = 0; px < (size_t)im.width * im.height; ++px) {
                for (int c = 0; c < convert_channels; ++c) {
                    float x = texels[t][px * im.channels + c] / 255.f;
                    x = x <= 0.04045f ? x / 12.92f : (float)std::pow((x + 0.055f) / 1.055f, 2.4);
                    texels[t][px * im.channels + c] = (uint8_t)glm::clamp(x * 255.f, 0.f, 255.f);
                }
            }
        }
        textures[t].width = im.width;
        textures[t].height = im.height;
        textures[t].channels = im.channels;
        textures[t].data = texels[t].data();
    }
    static_assert(sizeof(QuadLight) == 20 * sizeof(float), "QuadLight is the 80-byte record of util/lights.h");
    std::vector<QuadLight> lights(desc->n_lights);
    std::memcpy(lights.data(), desc->lights, sizeof(QuadLight) * desc->n_lights);

    std::vector<float> acc((size_t)fb_w * fb_h * 3, 0.f);
    std::vector<uint8_t> fb((size_t)fb_w * fb_h * 4, 0);
    std::vector<uint16_t> stats((size_t)fb_w * fb_h, 0);

    embree::SceneContext sc;
    sc.scene = reinterpret_cast<RTCScene>(&st);
    sc.instances = insts.data();
    sc.materials = mats.data();
    sc.lights = lights.data();
    sc.textures = textures.data();
    sc.num_lights = desc->n_lights;
    sc.num_instances = desc->n_instances;
    sc.fb_width = (uint32_t)fb_w;
    sc.fb_height = (uint32_t)fb_h;
    sc.accum_buffer = acc.data();
    sc.framebuffer = fb.data();
    sc.ray_stats = stats.data();

    // camera basis, render_embree.cpp:169-180 (glm::radians, cross, normalize spelled out)
    auto cross = [](const float a[3], const float b[3], float out[3]) { cross3(a, b, out); };
    auto normalize = [](float v[3]) {
        const float inv = 1.f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        v[0] *= inv;
        v[1] *= inv;
        v[2] *= inv;
    };
    const float plane_y = 2.f * std::tan(0.5f * fovy_deg * 0.01745329251994329576923690768489f);
    const float plane_x = plane_y * static_cast<float>(fb_w) / fb_h;
    float du[3], dv[3];
    cross(dir, up, du);
    normalize(du);
    for (int k = 0; k < 3; ++k) {
        du[k] *= plane_x;
    }
    cross(du, dir, dv);
    normalize(dv);
    for (int k = 0; k < 3; ++k) {
        dv[k] = -dv[k] * plane_y;
    }
    embree::ViewParams vp;
    vp.pos = glm::vec3(pos[0], pos[1], pos[2]);
    vp.dir_du = glm::vec3(du[0], du[1], du[2]);
    vp.dir_dv = glm::vec3(dv[0], dv[1], dv[2]);
    vp.dir_top_left = glm::vec3(dir[0] - 0.5f * du[0] - 0.5f * dv[0], dir[1] - 0.5f * du[1] - 0.5f * dv[1],
                                dir[2] - 0.5f * du[2] - 0.5f * dv[2]);
    vp.samples_per_pixel = desc->samples_per_pixel;

    for (int f = 0; f < n_frames; ++f) {
        vp.frame_id = (uint32_t)f;
        for (uint32_t j = 0; j < (uint32_t)fb_h; ++j) {
            for (uint32_t i = 0; i < (uint32_t)fb_w; ++i) {
                kernel::trace_ray(sc, vp, i, j);
            }
        }
    }
    std::memcpy(accum, acc.data(), acc.size() * sizeof(float));
    std::memcpy(framebuffer, fb.data(), fb.size());
    std::memcpy(ray_stats, stats.data(), stats.size() * sizeof(uint16_t));
    orc_scene_destroy(st.orc);
    return 0;
}

} // extern "C"
