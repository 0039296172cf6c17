// render_hip.cpp — RenderHIP: adapts the reference's RenderBackend interface
// (util/render_backend.h:12-32) to the C-ABI of include/crt_hip.h.
#include "render_hip.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <future>
#include <stdexcept>

#include "crt_hip.h"

namespace {

int requested_devices()
{
    int want = 1;
    if (const char *s = std::getenv("CRT_HIP_DEVICES")) {
        want = std::atoi(s);
    }
    const int have = crt_hip_device_count();
    if (have < 1) {
        throw std::runtime_error("RenderHIP: no HIP device (this backend has no CPU fallback)");
    }
    return want < 1 ? 1 : (want > have ? have : want);
}

void hip_ok(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        throw std::runtime_error(std::string("RenderHIP: ") + what + ": " + hipGetErrorString(e));
    }
}
void nccl_ok(ncclResult_t r, const char *what)
{
    if (r != ncclSuccess) {
        throw std::runtime_error(std::string("RenderHIP: ") + what + ": " + ncclGetErrorString(r));
    }
}

} // namespace

struct RenderHIP::MultiGpu {
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    void *gathered = nullptr; // on device 0: world consecutive tile slabs
    size_t slab_bytes = 0;

    explicit MultiGpu(int n)
    {
        std::vector<int> devs(n);
        for (int i = 0; i < n; ++i) {
            devs[i] = i;
        }
        comms.resize(n);
        nccl_ok(ncclCommInitAll(comms.data(), n, devs.data()), "ncclCommInitAll");
        streams.resize(n);
        for (int i = 0; i < n; ++i) {
            hip_ok(hipSetDevice(i), "hipSetDevice");
            hip_ok(hipStreamCreate(&streams[i]), "hipStreamCreate");
        }
    }
    ~MultiGpu()
    {
        for (size_t i = 0; i < comms.size(); ++i) {
            (void)hipSetDevice((int)i);
            (void)hipStreamDestroy(streams[i]);
            ncclCommDestroy(comms[i]);
        }
        if (gathered) {
            (void)hipSetDevice(0);
            (void)hipFree(gathered);
        }
    }
};

RenderHIP::RenderHIP()
{
    const int n = requested_devices();
    if (n > 1) {
        multi = std::make_unique<MultiGpu>(n);
    }
    // a plugin compiled against another version of the boundary (crt_render_stats grew with ABI 2) must not run
    if (crt_hip_abi_version() != CRT_HIP_ABI_VERSION) {
        throw std::runtime_error("RenderHIP: libcrt_hip_core.so speaks ABI " + std::to_string(crt_hip_abi_version()) +
                                 ", this plugin was built for ABI " + std::to_string(CRT_HIP_ABI_VERSION));
    }
    // CRT_HIP_ELIDE=1: do not trace the occlusion rays whose answer the reference never looks at (same image, same ray
    // statistics; include/crt_hip.h CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS). Off by default: the backend then traces what Embree traces.
    const char *elide = std::getenv("CRT_HIP_ELIDE");
    uint32_t flags = (elide != nullptr && elide[0] == '1') ? (uint32_t)CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS : (uint32_t)CRT_HIP_FLAG_NONE;
    // One GPU (`./chameleonrt hip <scene>`): set_scene returns with a quickly built tree and the full-quality one is swapped in a few
    // seconds later, between two frames of the accumulation -- same images (they do not depend on the tree), no multi-second wait
    // in set_scene (include/crt_hip.h CRT_HIP_FLAG_REFINE_IN_BACKGROUND; CRT_HIP_REFINE=0 waits for the full tree like rtcCommitScene).
    const char *refine_env = std::getenv("CRT_HIP_REFINE");
    if (n == 1 && !(refine_env != nullptr && refine_env[0] == '0')) {
        flags |= (uint32_t)CRT_HIP_FLAG_REFINE_IN_BACKGROUND;
    }
    for (int d = 0; d < n; ++d) {
        crt_hip_ctx *c = crt_hip_create(d, flags);
        if (!c) {
            throw std::runtime_error(std::string("RenderHIP: ") + crt_hip_last_error(nullptr));
        }
        ctxs.push_back(c);
        if (n > 1) {
            check(c, crt_hip_set_partition(c, d, n), "crt_hip_set_partition");
            check(c, crt_hip_set_stream(c, multi->streams[d]), "crt_hip_set_stream");
        }
    }
}

RenderHIP::~RenderHIP()
{
    for (crt_hip_ctx *c : ctxs) {
        crt_hip_destroy(c);
    }
}

void RenderHIP::check(crt_hip_ctx *ctx, int rc, const char *what) const
{
    if (rc != CRT_HIP_OK) { // the reference reports errors as exceptions
        throw std::runtime_error(std::string("RenderHIP: ") + what + ": " + crt_hip_last_error(ctx));
    }
}

std::string RenderHIP::name()
{
    std::string n = crt_hip_name(ctxs[0]);
    if (ctxs.size() > 1) {
        n += " x" + std::to_string(ctxs.size()) + " (tile split + RCCL gather)";
    }
    return n;
}

void RenderHIP::initialize(const int width, const int height)
{
    fb_width = width;
    fb_height = height;
    img.resize(size_t(width) * height);
    for (crt_hip_ctx *c : ctxs) {
        check(c, crt_hip_initialize(c, width, height), "crt_hip_initialize");
    }
    if (multi) {
        void *ptr = nullptr;
        check(ctxs[0], crt_hip_tile_buffer(ctxs[0], &ptr, &multi->slab_bytes), "crt_hip_tile_buffer");
        hip_ok(hipSetDevice(0), "hipSetDevice");
        if (multi->gathered) {
            hip_ok(hipFree(multi->gathered), "hipFree");
        }
        hip_ok(hipMalloc(&multi->gathered, multi->slab_bytes * ctxs.size()), "hipMalloc");
    }
}

void RenderHIP::set_scene(const Scene &scene)
{
    samples_per_pixel = scene.samples_per_pixel;

    // Scene (util/scene.h:23-32) -> flat crt_scene_desc. glm::vec3 / vec2 / uvec3 arrays are
    // tightly packed floats / uints, so the pointers are passed through without copying.
    std::vector<crt_geometry_desc> geoms;
    std::vector<crt_mesh_desc> meshes;
    for (const auto &mesh : scene.meshes) {
        crt_mesh_desc md;
        md.first_geometry = uint32_t(geoms.size());
        md.n_geometries = uint32_t(mesh.geometries.size());
        meshes.push_back(md);
        for (const auto &g : mesh.geometries) {
            crt_geometry_desc gd;
            gd.vertices = reinterpret_cast<const float *>(g.vertices.data());
            gd.n_vertices = g.vertices.size();
            gd.indices = reinterpret_cast<const uint32_t *>(g.indices.data());
            gd.n_triangles = g.indices.size();
            gd.uvs = g.uvs.empty() ? nullptr : reinterpret_cast<const float *>(g.uvs.data());
            geoms.push_back(gd);
        }
    }
    std::vector<crt_parameterized_mesh_desc> pmeshes;
    for (const auto &pm : scene.parameterized_meshes) {
        crt_parameterized_mesh_desc d;
        d.mesh_id = uint32_t(pm.mesh_id);
        d.n_material_ids = uint32_t(pm.material_ids.size());
        d.material_ids = pm.material_ids.data();
        pmeshes.push_back(d);
    }
    std::vector<crt_instance_desc> instances;
    for (const auto &inst : scene.instances) {
        crt_instance_desc d;
        std::memcpy(d.transform, &inst.transform, sizeof(float) * 16); // glm::mat4 is column-major
        d.parameterized_mesh_id = uint32_t(inst.parameterized_mesh_id);
        instances.push_back(d);
    }
    std::vector<crt_image_desc> textures;
    for (const auto &im : scene.textures) {
        crt_image_desc d;
        d.width = im.width;
        d.height = im.height;
        d.channels = im.channels;
        d.color_space = im.color_space == SRGB ? CRT_COLORSPACE_SRGB : CRT_COLORSPACE_LINEAR;
        d.data = im.img.data();
        textures.push_back(d);
    }
    static_assert(sizeof(DisneyMaterial) == 16 * sizeof(float), "DisneyMaterial is 16 floats (util/material.h)");
    static_assert(sizeof(QuadLight) == 20 * sizeof(float), "QuadLight is 20 floats (util/lights.h)");

    crt_scene_desc desc;
    std::memset(&desc, 0, sizeof(desc));
    desc.geometries = geoms.data();
    desc.n_geometries = uint32_t(geoms.size());
    desc.meshes = meshes.data();
    desc.n_meshes = uint32_t(meshes.size());
    desc.parameterized_meshes = pmeshes.data();
    desc.n_parameterized_meshes = uint32_t(pmeshes.size());
    desc.instances = instances.data();
    desc.n_instances = uint32_t(instances.size());
    desc.materials = reinterpret_cast<const float *>(scene.materials.data());
    desc.n_materials = uint32_t(scene.materials.size());
    desc.textures = textures.data();
    desc.n_textures = uint32_t(textures.size());
    desc.lights = reinterpret_cast<const float *>(scene.lights.data());
    desc.n_lights = uint32_t(scene.lights.size());
    desc.samples_per_pixel = scene.samples_per_pixel;

    // Every GPU keeps a full scene replica (a San-Miguel-class scene is ~2 GB of 288 GB). The host
    // half of set_scene (BVH build = rtcCommitScene's job, texture linearisation) runs ONCE with all
    // host cores; only the uploads are per device.
    if (ctxs.size() == 1) { // one GPU: prepare + upload in one call (which may refine the tree in the background, see the constructor)
        check(ctxs[0], crt_hip_set_scene(ctxs[0], &desc), "crt_hip_set_scene");
        return;
    }
    crt_hip_prepared_scene *prepared = crt_hip_prepare_scene(&desc, 0);
    if (!prepared) {
        throw std::runtime_error(std::string("RenderHIP: crt_hip_prepare_scene: ") + crt_hip_last_error(nullptr));
    }
    std::vector<std::future<int>> jobs;
    for (crt_hip_ctx *c : ctxs) {
        jobs.push_back(std::async(std::launch::async, [c, prepared]() { return crt_hip_set_prepared_scene(c, prepared); }));
    }
    std::vector<int> rcs;
    for (auto &j : jobs) {
        rcs.push_back(j.get());
    }
    crt_hip_free_prepared_scene(prepared);
    for (size_t i = 0; i < rcs.size(); ++i) {
        check(ctxs[i], rcs[i], "crt_hip_set_prepared_scene");
    }
}

RenderStats RenderHIP::render(const glm::vec3 &pos,
                              const glm::vec3 &dir,
                              const glm::vec3 &up,
                              const float fovy,
                              const bool camera_changed,
                              const bool readback_framebuffer)
{
    // GLDisplay uploads `img` every frame (util/display/gldisplay.cpp:113-121), so the image is
    // always brought back, like the Embree backend which renders into host memory.
    (void)readback_framebuffer;
    return render_impl(pos, dir, up, fovy, camera_changed, true);
}

RenderStats RenderHIP::render_to_device(const glm::vec3 &pos, const glm::vec3 &dir, const glm::vec3 &up, const float fovy,
                                        const bool camera_changed, const bool readback_framebuffer)
{
    return render_impl(pos, dir, up, fovy, camera_changed, readback_framebuffer);
}

void RenderHIP::device_framebuffer(void **device_ptr, size_t *pitch_bytes)
{
    check(ctxs[0], crt_hip_device_framebuffer(ctxs[0], device_ptr, pitch_bytes), "crt_hip_device_framebuffer");
}

RenderStats RenderHIP::render_impl(const glm::vec3 &pos, const glm::vec3 &dir, const glm::vec3 &up, const float fovy,
                                   const bool camera_changed, const bool readback)
{
    const float p[3] = {pos.x, pos.y, pos.z}, d[3] = {dir.x, dir.y, dir.z}, u[3] = {up.x, up.y, up.z};
    RenderStats stats;
    if (!multi) {
        crt_render_stats st;
        check(ctxs[0], crt_hip_render(ctxs[0], p, d, u, fovy, camera_changed, readback ? 1 : 0, &st), "crt_hip_render");
        stats.render_time = st.render_time_ms;
        stats.rays_per_second = st.rays_per_second;
        copy_image(readback);
        return stats;
    }
    // one wall clock around trace + gather + assemble + read-back, like the reference's render()
    // (render_embree.cpp:177-211 times its whole body) and like the single-GPU path above
    const auto t_begin = std::chrono::high_resolution_clock::now();
    const size_t n = ctxs.size();
    std::vector<crt_render_stats> st(n);
    std::vector<std::future<int>> jobs;
    for (size_t i = 0; i < n; ++i) {
        crt_hip_ctx *c = ctxs[i];
        crt_render_stats *out = &st[i];
        jobs.push_back(std::async(std::launch::async, [=]() {
            return crt_hip_render(c, p, d, u, fovy, camera_changed, 0, out);
        }));
    }
    double rays = 0.0;
    for (size_t i = 0; i < n; ++i) {
        check(ctxs[i], jobs[i].get(), "crt_hip_render");
        rays += double(st[i].rays);
    }
    // gather: every device sends its compact RGBA8 tile slab to device 0 (<= 4 MiB each at 4K)
    nccl_ok(ncclGroupStart(), "ncclGroupStart");
    for (size_t i = 0; i < n; ++i) {
        void *slab = nullptr;
        size_t bytes = 0;
        check(ctxs[i], crt_hip_tile_buffer(ctxs[i], &slab, &bytes), "crt_hip_tile_buffer");
        nccl_ok(ncclSend(slab, bytes, ncclUint8, 0, multi->comms[i], multi->streams[i]), "ncclSend");
        nccl_ok(ncclRecv(static_cast<char *>(multi->gathered) + i * bytes, bytes, ncclUint8, int(i), multi->comms[0],
                         multi->streams[0]),
                "ncclRecv");
    }
    nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
    check(ctxs[0], crt_hip_assemble_tiles(ctxs[0], multi->gathered, int(n), readback ? 1 : 0), "crt_hip_assemble_tiles");
    if (!readback) {
        // (assemble_tiles waits only for a read-back; whoever takes the image from HBM next -- RenderHIPGL's copy into the GL
        // texture -- must find it complete, and render() reports the time of the whole frame)
        hip_ok(hipSetDevice(0), "hipSetDevice");
        hip_ok(hipStreamSynchronize(multi->streams[0]), "hipStreamSynchronize");
    }
    copy_image(readback);
    const float ms = std::chrono::duration<float, std::milli>(std::chrono::high_resolution_clock::now() - t_begin).count();
    stats.render_time = ms;
    stats.rays_per_second = float(rays / (ms * 1.0e-3));
    return stats;
}

void RenderHIP::copy_image(bool readback)
{
    if (readback) {
        std::memcpy(img.data(), crt_hip_framebuffer(ctxs[0]), img.size() * sizeof(uint32_t));
    }
}
