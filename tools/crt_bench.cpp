// crt_bench.cpp — headless C++ driver of RenderHIP: the reference app's `-benchmark-frames N`
// protocol (main.cpp:113-345) without SDL/ImGui. Builds against the stand-in headers
// (backends/hip/check_shim.sh) so the backend class that the plugin exports is exercised from
// C++ exactly as `chameleonrt` would: make_renderer -> initialize -> set_scene -> render loop.
//
//   crt_bench [-img W H] [-spp N] [-benchmark-frames N] [-ppm out.ppm]
// Scene: the 34-triangle Cornell box generated in code (no asset files exist in this image).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "render_hip.h"

static glm::vec3 v3(float x, float y, float z) { return glm::vec3{x, y, z}; }
static glm::vec3 sub(glm::vec3 a, glm::vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static glm::vec3 cross(glm::vec3 a, glm::vec3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static glm::vec3 norm(glm::vec3 v)
{
    const float c = 1.f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return v3(v.x * c, v.y * c, v.z * c);
}

static void add_quad(Mesh &m, glm::vec3 a, glm::vec3 b, glm::vec3 c, glm::vec3 d)
{
    Geometry g;
    g.vertices = {a, b, c, d};
    g.uvs = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    g.indices = {{0, 1, 2}, {0, 2, 3}};
    m.geometries.push_back(g);
}
static void add_box(Mesh &m, glm::vec3 c, glm::vec3 h)
{
    Geometry g;
    for (int i = 0; i < 8; ++i) {
        g.vertices.push_back(v3(c.x + ((i & 1) ? h.x : -h.x), c.y + ((i & 2) ? h.y : -h.y), c.z + ((i & 4) ? h.z : -h.z)));
    }
    const unsigned f[12][3] = {{0, 2, 1}, {1, 2, 3}, {4, 5, 6}, {5, 7, 6}, {0, 1, 4}, {1, 5, 4},
                               {2, 6, 3}, {3, 6, 7}, {0, 4, 2}, {2, 4, 6}, {1, 3, 5}, {3, 7, 5}};
    for (auto &t : f) {
        g.indices.push_back({t[0], t[1], t[2]});
    }
    m.geometries.push_back(g);
}
static DisneyMaterial diffuse(float r, float g, float b)
{
    DisneyMaterial m;
    std::memset(&m, 0, sizeof(m));
    m.base_color = v3(r, g, b);
    m.roughness = 1.f;
    m.ior = 1.5f;
    return m;
}

int main(int argc, char **argv)
{
    int w = 512, h = 512, frames = 16;
    uint32_t spp = 1;
    std::string ppm;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "-img")) {
            w = std::atoi(argv[++i]);
            h = std::atoi(argv[++i]);
        } else if (!std::strcmp(argv[i], "-spp")) {
            spp = (uint32_t)std::atoi(argv[++i]);
        } else if (!std::strcmp(argv[i], "-benchmark-frames")) {
            frames = std::atoi(argv[++i]);
        } else if (!std::strcmp(argv[i], "-ppm")) {
            ppm = argv[++i];
        }
    }
    Scene scene;
    Mesh mesh;
    add_quad(mesh, v3(-1, 0, 1), v3(1, 0, 1), v3(1, 0, -1), v3(-1, 0, -1));
    add_quad(mesh, v3(-1, 2, -1), v3(1, 2, -1), v3(1, 2, 1), v3(-1, 2, 1));
    add_quad(mesh, v3(-1, 0, -1), v3(1, 0, -1), v3(1, 2, -1), v3(-1, 2, -1));
    add_quad(mesh, v3(-1, 0, 1), v3(-1, 0, -1), v3(-1, 2, -1), v3(-1, 2, 1));
    add_quad(mesh, v3(1, 0, -1), v3(1, 0, 1), v3(1, 2, 1), v3(1, 2, -1));
    add_box(mesh, v3(0.33f, 0.3f, 0.35f), v3(0.3f, 0.3f, 0.3f));
    add_box(mesh, v3(-0.35f, 0.6f, -0.3f), v3(0.3f, 0.6f, 0.3f));
    scene.meshes.push_back(mesh);
    scene.parameterized_meshes.push_back(ParameterizedMesh{0, {0, 0, 0, 1, 2, 0, 0}});
    Instance inst;
    std::memset(&inst, 0, sizeof(inst));
    inst.transform.col[0].x = inst.transform.col[1].y = inst.transform.col[2].z = inst.transform.col[3].w = 1.f;
    inst.parameterized_mesh_id = 0;
    scene.instances.push_back(inst);
    scene.materials = {diffuse(0.73f, 0.73f, 0.73f), diffuse(0.65f, 0.05f, 0.05f), diffuse(0.12f, 0.45f, 0.15f)};
    QuadLight light; // the light the OBJ importer generates (util/scene.cpp:218-227)
    std::memset(&light, 0, sizeof(light));
    const glm::vec3 n = norm(v3(0.5f, -0.8f, -0.5f));
    light.emission = glm::vec4{20, 20, 20, 20};
    light.normal = glm::vec4{n.x, n.y, n.z, 0};
    light.position = glm::vec4{-10 * n.x, -10 * n.y, -10 * n.z, 0};
    light.v_x = norm(cross(v3(1, 0, 0), n));
    light.v_y = norm(cross(n, light.v_x));
    light.width = light.height = 5.f;
    scene.lights.push_back(light);
    scene.samples_per_pixel = spp;

    try {
        std::unique_ptr<RenderBackend> renderer = std::make_unique<RenderHIP>();
        renderer->initialize(w, h);
        renderer->set_scene(scene);
        const glm::vec3 eye = v3(0, 1, 3.4f), center = v3(0, 1, 0);
        const glm::vec3 dir = norm(sub(center, eye));
        const glm::vec3 up = norm(cross(norm(cross(dir, v3(0, 1, 0))), dir));
        float ms = 0.f, rps = 0.f;
        for (int f = 0; f < frames; ++f) {
            const RenderStats st = renderer->render(eye, dir, up, 40.f, f == 0, f + 1 == frames);
            ms += st.render_time;
            rps += st.rays_per_second;
        }
        std::printf("%s\nBenchmarked %d frames\nRender Time: %g ms/frame (%g FPS)\nRays per-second %g Ray/s\n",
                    renderer->name().c_str(), frames, ms / frames, 1000.f / (ms / frames), rps / frames);
        if (!ppm.empty()) {
            FILE *fp = std::fopen(ppm.c_str(), "wb");
            std::fprintf(fp, "P6 %d %d 255\n", w, h);
            for (uint32_t px : renderer->img) {
                std::fputc(px & 255, fp);
                std::fputc((px >> 8) & 255, fp);
                std::fputc((px >> 16) & 255, fp);
            }
            std::fclose(fp);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
