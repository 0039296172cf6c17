// chameleonrt_standin.h — COMPILE-CHECK STAND-IN, not part of the product.
//
// SDL2 and glm are not installed in the build image, so backends/hip/render_hip.cpp cannot be
// compiled against the reference's real headers here. This single header declares just enough
// of their *shape* (member names, types, virtual order) for `backends/hip/check_shim.sh` to
// type-check the shim with -DCRT_HIP_STANDIN. A real build never sees this file: it uses
// util/render_backend.h, util/scene.h, util/mesh.h, util/material.h, util/lights.h and glm.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace glm {
struct vec2 {
    float x, y;
};
struct vec3 {
    float x, y, z;
};
struct vec4 {
    float x, y, z, w;
};
struct uvec3 {
    unsigned x, y, z;
};
struct mat4 {
    vec4 col[4];
};
} // namespace glm

enum ColorSpace { LINEAR, SRGB };

struct Image {
    std::string name;
    int width = -1, height = -1, channels = -1;
    std::vector<uint8_t> img;
    ColorSpace color_space = LINEAR;
};
struct DisneyMaterial {
    glm::vec3 base_color;
    float metallic, specular, roughness, specular_tint, anisotropy, sheen, sheen_tint, clearcoat, clearcoat_gloss, ior,
        specular_transmission;
    glm::vec2 pad;
};
struct QuadLight {
    glm::vec4 emission, position, normal;
    glm::vec3 v_x;
    float width;
    glm::vec3 v_y;
    float height;
};
struct Geometry {
    std::vector<glm::vec3> vertices, normals;
    std::vector<glm::vec2> uvs;
    std::vector<glm::uvec3> indices;
};
struct Mesh {
    std::vector<Geometry> geometries;
};
struct ParameterizedMesh {
    size_t mesh_id;
    std::vector<uint32_t> material_ids;
};
struct Instance {
    glm::mat4 transform;
    size_t parameterized_mesh_id;
};
struct Scene {
    std::vector<Mesh> meshes;
    std::vector<ParameterizedMesh> parameterized_meshes;
    std::vector<Instance> instances;
    std::vector<DisneyMaterial> materials;
    std::vector<Image> textures;
    std::vector<QuadLight> lights;
    uint32_t samples_per_pixel = 1;
};
struct RenderStats {
    float render_time = 0;
    float rays_per_second = 0;
};
struct RenderBackend {
    std::vector<uint32_t> img;
    uint32_t samples_per_pixel = 1;
    virtual ~RenderBackend() {}
    virtual std::string name() = 0;
    virtual void initialize(const int fb_width, const int fb_height) = 0;
    virtual void set_scene(const Scene &scene) = 0;
    virtual RenderStats render(const glm::vec3 &pos, const glm::vec3 &dir, const glm::vec3 &up, const float fovy,
                               const bool camera_changed, const bool readback_framebuffer) = 0;
};
