// ref_scene_driver.cpp -- C access to the REFERENCE's scene importer, for golden Scene dumps.
// TEST INFRASTRUCTURE (oracle/Makefile, target `ref`): compiled together with the reference's own
// util/scene.cpp, mesh.cpp, material.cpp, util.cpp, flatten_gltf.cpp, gltf_types.cpp, buffer_view.cpp and
// file_mapping.cpp FROM WHERE THEY LIE under $(REFERENCE) (with their vendored tinyobjloader, tinygltf,
// stb_image, nlohmann json and parallel_hashmap), against the GLM stand-in of oracle/ref_shim_scene/
// (GLM itself is fetched by the reference's cmake and is absent here). Output: oracle/_ref/libref_scene.so.
// Nothing of this ships; tests/golden/make_scene_golden.py uses it to dump what
// `Scene::Scene(fname, material_mode)` (util/scene.cpp:49-72) produces for small OBJ / glTF / GLB / CRTS
// files, and tests/test_importers_pinned.py compares chameleonrt_amd's importers against it.
#include <cstring>
#include <exception>
#include <string>

#include "scene.h"

static std::string g_err;

extern "C" {

void *refscene_load(const char *path, int white_diffuse)
{
    try {
        return new Scene(path, white_diffuse ? MaterialMode::WHITE_DIFFUSE : MaterialMode::DEFAULT);
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
const char *refscene_error() { return g_err.c_str(); }
void refscene_free(void *s) { delete static_cast<Scene *>(s); }

// counts: meshes, parameterized meshes, instances, materials, textures, lights, cameras
void refscene_counts(const void *s_, uint64_t out[7])
{
    const Scene &s = *static_cast<const Scene *>(s_);
    out[0] = s.meshes.size();
    out[1] = s.parameterized_meshes.size();
    out[2] = s.instances.size();
    out[3] = s.materials.size();
    out[4] = s.textures.size();
    out[5] = s.lights.size();
    out[6] = s.cameras.size();
}
uint64_t refscene_mesh_geometries(const void *s_, uint64_t mesh) { return static_cast<const Scene *>(s_)->meshes[mesh].geometries.size(); }
// sizes: vertices, normals, uvs, triangles
void refscene_geometry_sizes(const void *s_, uint64_t mesh, uint64_t geom, uint64_t out[4])
{
    const Geometry &g = static_cast<const Scene *>(s_)->meshes[mesh].geometries[geom];
    out[0] = g.vertices.size();
    out[1] = g.normals.size();
    out[2] = g.uvs.size();
    out[3] = g.indices.size();
}
void refscene_geometry_copy(const void *s_, uint64_t mesh, uint64_t geom, float *vertices, float *uvs, uint32_t *indices)
{
    const Geometry &g = static_cast<const Scene *>(s_)->meshes[mesh].geometries[geom];
    static_assert(sizeof(glm::vec3) == 12 && sizeof(glm::vec2) == 8 && sizeof(glm::uvec3) == 12, "tight vectors");
    std::memcpy(vertices, g.vertices.data(), g.vertices.size() * sizeof(glm::vec3));
    if (uvs) {
        std::memcpy(uvs, g.uvs.data(), g.uvs.size() * sizeof(glm::vec2));
    }
    std::memcpy(indices, g.indices.data(), g.indices.size() * sizeof(glm::uvec3));
}
uint64_t refscene_pmesh(const void *s_, uint64_t i, uint64_t *mesh_id)
{
    const ParameterizedMesh &p = static_cast<const Scene *>(s_)->parameterized_meshes[i];
    *mesh_id = p.mesh_id;
    return p.material_ids.size();
}
void refscene_pmesh_materials(const void *s_, uint64_t i, uint32_t *ids)
{
    const ParameterizedMesh &p = static_cast<const Scene *>(s_)->parameterized_meshes[i];
    std::memcpy(ids, p.material_ids.data(), p.material_ids.size() * sizeof(uint32_t));
}
uint64_t refscene_instance(const void *s_, uint64_t i, float transform[16])
{
    const Instance &in = static_cast<const Scene *>(s_)->instances[i];
    static_assert(sizeof(glm::mat4) == 64, "column-major 4x4 floats");
    std::memcpy(transform, &in.transform, 64);
    return in.parameterized_mesh_id;
}
void refscene_material(const void *s_, uint64_t i, float out[16])
{
    static_assert(sizeof(DisneyMaterial) == 64, "DisneyMaterial is 16 floats (util/material.h:29-46)");
    std::memcpy(out, &static_cast<const Scene *>(s_)->materials[i], 64);
}
// info: width, height, channels, color space (0 linear, 1 sRGB)
void refscene_texture_info(const void *s_, uint64_t i, int32_t info[4])
{
    const Image &im = static_cast<const Scene *>(s_)->textures[i];
    info[0] = im.width;
    info[1] = im.height;
    info[2] = im.channels;
    info[3] = im.color_space == SRGB ? 1 : 0;
}
void refscene_texture_copy(const void *s_, uint64_t i, uint8_t *out)
{
    const Image &im = static_cast<const Scene *>(s_)->textures[i];
    std::memcpy(out, im.img.data(), im.img.size());
}
void refscene_light(const void *s_, uint64_t i, float out[20])
{
    static_assert(sizeof(QuadLight) == 80, "QuadLight is 20 floats (util/lights.h:6-18)");
    std::memcpy(out, &static_cast<const Scene *>(s_)->lights[i], 80);
}
// position, center, up, fov_y
void refscene_camera(const void *s_, uint64_t i, float out[10])
{
    const Camera &c = static_cast<const Scene *>(s_)->cameras[i];
    const float v[10] = {c.position.x, c.position.y, c.position.z, c.center.x, c.center.y, c.center.z, c.up.x, c.up.y, c.up.z, c.fov_y};
    std::memcpy(out, v, sizeof(v));
}

} // extern "C"
