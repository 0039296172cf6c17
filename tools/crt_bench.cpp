// crt_bench.cpp — `./chameleonrt hip <scene>` minus the window: the reference app's main + run_app (main.cpp:49-345)
// without SDL / ImGui. The renderer comes the way the app gets it: the reference's own loader, `RenderPlugin("crt_hip")`
// (util/render_plugin.cpp:14-60, compiled from where it lies), dlopens <exe dir>/libcrt_hip.so -- backends/hip's
// render_hip_plugin.cpp + render_hip.cpp -- dlsyms `populate_plugin_functions` and calls make_renderer (main.cpp:65-66,160).
// Everything is compiled against the reference's OWN headers (util/render_plugin.h, render_backend.h, scene.h, mesh.h,
// material.h, lights.h, camera.h, display/display.h, imgui.h) and, when a scene file is named, the scene is loaded by the
// reference's own importer (util/scene.cpp and the files oracle/Makefile lists as REF_SCENE_SRC): Scene(fname,
// MaterialMode) -> RenderBackend::set_scene(const Scene &) is then exactly the hand-over of main.cpp:185-214. Stand-ins
// only for what the reference fetches from outside its tree: GLM (oracle/ref_shim_scene) and <SDL.h> (oracle/ref_shim_app:
// three type names + SDL_GetBasePath). Built by `make -C oracle ref` into oracle/_ref/ (it contains compiled reference code).
//
//   crt_bench [scene.obj|.gltf|.glb|.crts] [-img W H] [-spp N] [-benchmark-frames N] [-mat-mode white_diffuse]
//             [-eye x y z] [-center x y z] [-up x y z] [-fov deg] [-camera id] [-ppm out.ppm]
//
// Without a scene file: the 34-triangle Cornell box built in code through the same types. The camera handed to render()
// is printed as hex floats ("camera: ...") so that a test can hand another front end of the C-ABI the same bits.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "render_plugin.h"

static glm::vec3 unit(const glm::vec3 &v)
{
    const float c = 1.f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return glm::vec3(v.x * c, v.y * c, v.z * c);
}
static glm::vec3 cross3(const glm::vec3 &a, const glm::vec3 &b)
{
    return glm::vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

static void add_quad(Mesh &m, glm::vec3 a, glm::vec3 b, glm::vec3 c, glm::vec3 d)
{
    Geometry g;
    g.vertices = {a, b, c, d};
    g.uvs = {glm::vec2(0, 0), glm::vec2(1, 0), glm::vec2(1, 1), glm::vec2(0, 1)};
    g.indices = {glm::uvec3(0, 1, 2), glm::uvec3(0, 2, 3)};
    m.geometries.push_back(g);
}
static void add_box(Mesh &m, glm::vec3 c, glm::vec3 h)
{
    Geometry g;
    for (int i = 0; i < 8; ++i) {
        g.vertices.push_back(glm::vec3(c.x + ((i & 1) ? h.x : -h.x), c.y + ((i & 2) ? h.y : -h.y), c.z + ((i & 4) ? h.z : -h.z)));
    }
    const unsigned f[12][3] = {{0, 2, 1}, {1, 2, 3}, {4, 5, 6}, {5, 7, 6}, {0, 1, 4}, {1, 5, 4},
                               {2, 6, 3}, {3, 6, 7}, {0, 4, 2}, {2, 4, 6}, {1, 3, 5}, {3, 7, 5}};
    for (auto &t : f) {
        g.indices.push_back(glm::uvec3(t[0], t[1], t[2]));
    }
    m.geometries.push_back(g);
}
static DisneyMaterial diffuse(float r, float g, float b)
{
    DisneyMaterial m; // util/material.h:29-46 defaults: roughness 1, ior 1.5, everything else 0
    m.base_color = glm::vec3(r, g, b);
    return m;
}

// The Cornell box of SURVEY 8d S1, through the reference's types (no asset files exist in this image).
static void cornell(Scene &scene)
{
    Mesh mesh;
    add_quad(mesh, glm::vec3(-1, 0, 1), glm::vec3(1, 0, 1), glm::vec3(1, 0, -1), glm::vec3(-1, 0, -1));
    add_quad(mesh, glm::vec3(-1, 2, -1), glm::vec3(1, 2, -1), glm::vec3(1, 2, 1), glm::vec3(-1, 2, 1));
    add_quad(mesh, glm::vec3(-1, 0, -1), glm::vec3(1, 0, -1), glm::vec3(1, 2, -1), glm::vec3(-1, 2, -1));
    add_quad(mesh, glm::vec3(-1, 0, 1), glm::vec3(-1, 0, -1), glm::vec3(-1, 2, -1), glm::vec3(-1, 2, 1));
    add_quad(mesh, glm::vec3(1, 0, -1), glm::vec3(1, 0, 1), glm::vec3(1, 2, 1), glm::vec3(1, 2, -1));
    add_box(mesh, glm::vec3(0.33f, 0.3f, 0.35f), glm::vec3(0.3f, 0.3f, 0.3f));
    add_box(mesh, glm::vec3(-0.35f, 0.6f, -0.3f), glm::vec3(0.3f, 0.6f, 0.3f));
    scene.meshes.push_back(mesh);
    ParameterizedMesh pm;
    pm.mesh_id = 0;
    pm.material_ids = {0, 0, 0, 1, 2, 0, 0};
    scene.parameterized_meshes.push_back(pm);
    Instance inst;
    inst.transform = glm::mat4(1.f);
    inst.parameterized_mesh_id = 0;
    scene.instances.push_back(inst);
    scene.materials = {diffuse(0.73f, 0.73f, 0.73f), diffuse(0.65f, 0.05f, 0.05f), diffuse(0.12f, 0.45f, 0.15f)};
    QuadLight light; // the light the OBJ importer generates (util/scene.cpp:218-227)
    const glm::vec3 n = unit(glm::vec3(0.5f, -0.8f, -0.5f));
    light.emission = glm::vec4(20.f);
    light.normal = glm::vec4(n.x, n.y, n.z, 0.f);
    light.position = glm::vec4(-10 * n.x, -10 * n.y, -10 * n.z, 0.f);
    light.v_x = unit(cross3(glm::vec3(1, 0, 0), n));
    light.v_y = unit(cross3(n, light.v_x));
    light.width = light.height = 5.f;
    scene.lights.push_back(light);
}

int main(int argc, char **argv)
{
    int w = 1280, h = 720, frames = 16; // main.cpp:35-36
    uint32_t spp = 1;
    size_t camera_id = 0;
    bool got_camera_args = false;
    glm::vec3 eye(0, 0, 5), center(0.f), up(0, 1, 0); // main.cpp:122-125
    float fov_y = 65.f;
    MaterialMode material_mode = MaterialMode::DEFAULT;
    std::string ppm, scene_file;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto f3 = [&](glm::vec3 &v) {
            v.x = std::stof(argv[++i]);
            v.y = std::stof(argv[++i]);
            v.z = std::stof(argv[++i]);
            got_camera_args = true;
        };
        if (a == "-img") {
            w = std::atoi(argv[++i]);
            h = std::atoi(argv[++i]);
        } else if (a == "-spp") {
            spp = (uint32_t)std::atoi(argv[++i]);
        } else if (a == "-benchmark-frames") {
            frames = std::atoi(argv[++i]);
        } else if (a == "-ppm") {
            ppm = argv[++i];
        } else if (a == "-eye") {
            f3(eye);
        } else if (a == "-center") {
            f3(center);
        } else if (a == "-up") {
            f3(up);
        } else if (a == "-fov") {
            fov_y = std::stof(argv[++i]);
            got_camera_args = true;
        } else if (a == "-camera") {
            camera_id = (size_t)std::atol(argv[++i]);
        } else if (a == "-mat-mode") {
            if (std::string(argv[++i]) == "white_diffuse") {
                material_mode = MaterialMode::WHITE_DIFFUSE;
            }
        } else if (a[0] != '-') {
            scene_file = a;
        }
    }

    try {
        // main.cpp:65-66,99-102: the plugin outlives the renderer; no window, so no display is made and make_renderer gets none
        RenderPlugin plugin("crt_hip");
        if (plugin.get_window_flags() == 0) {
            throw std::runtime_error("the hip plugin asks for no window flags (expected SDL_WINDOW_OPENGL)");
        }
        std::unique_ptr<RenderBackend> renderer = plugin.make_renderer(nullptr);
        if (!renderer) {
            throw std::runtime_error("make_renderer returned no backend");
        }
        renderer->initialize(w, h);
        {
            // main.cpp:185-214: the Scene lives only for the hand-over; set_scene must have copied what it needs
            std::unique_ptr<Scene> scene;
            if (scene_file.empty()) {
                scene = std::make_unique<Scene>();
                cornell(*scene);
                if (!got_camera_args) {
                    eye = glm::vec3(0, 1, 3.4f);
                    center = glm::vec3(0, 1, 0);
                    fov_y = 40.f;
                }
            } else {
                scene = std::make_unique<Scene>(scene_file, material_mode);
            }
            scene->samples_per_pixel = spp;
            std::printf("# Unique Triangles: %zu\n# Total Triangles: %zu\n# Geometries: %zu\n# Meshes: %zu\n"
                        "# Parameterized Meshes: %zu\n# Instances: %zu\n# Materials: %zu\n# Textures: %zu\n# Lights: %zu\n"
                        "# Cameras: %zu\n# Samples per Pixel: %u\n",
                        scene->unique_tris(), scene->total_tris(), scene->num_geometries(), scene->meshes.size(),
                        scene->parameterized_meshes.size(), scene->instances.size(), scene->materials.size(),
                        scene->textures.size(), scene->lights.size(), scene->cameras.size(), scene->samples_per_pixel);
            renderer->set_scene(*scene);
            if (!got_camera_args && !scene->cameras.empty()) {
                const Camera &c = scene->cameras.at(camera_id);
                eye = c.position;
                center = c.center;
                up = c.up;
                fov_y = c.fov_y;
            }
        }
        // what ArcballCamera(eye, center, up) hands to render(): unit dir, up re-orthogonalised (arcball_camera.cpp:10-23,65-72)
        const glm::vec3 dir = unit(glm::vec3(center.x - eye.x, center.y - eye.y, center.z - eye.z));
        const glm::vec3 cam_up = unit(cross3(unit(cross3(dir, unit(up))), dir));
        std::printf("camera: %a %a %a  %a %a %a  %a %a %a  %a\n", eye.x, eye.y, eye.z, dir.x, dir.y, dir.z, cam_up.x, cam_up.y,
                    cam_up.z, fov_y);
        float ms = 0.f, rps = 0.f;
        for (int f = 0; f < frames; ++f) {
            const RenderStats st = renderer->render(eye, dir, cam_up, fov_y, f == 0, f + 1 == frames);
            ms += st.render_time;
            rps += st.rays_per_second;
        }
        std::printf("%s\nBenchmarked %d frames\nRender Time: %g ms/frame (%g FPS)\nRays per-second %g Ray/s\n",
                    renderer->name().c_str(), frames, ms / frames, 1000.f / (ms / frames), rps / frames);
        if (!ppm.empty()) {
            FILE *fp = std::fopen(ppm.c_str(), "wb");
            if (!fp) {
                throw std::runtime_error("cannot write " + ppm);
            }
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
