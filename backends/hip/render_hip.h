// render_hip.h — `struct RenderHIP : RenderBackend`, the ChameleonRT backend class of the
// MI355X wavefront path tracer. Drop-in next to RenderEmbree
// (reference backends/embree/render_embree.h:11-44); built into libcrt_hip.so and discovered by
// `./chameleonrt hip <scene>` through util/render_plugin.cpp:14-60.
//
// All rendering lives behind the C-ABI of include/crt_hip.h (libcrt_hip_core.so). This class
// only adapts types: Scene -> crt_scene_desc, glm::vec3 -> float[3], error codes -> exceptions.
// With CRT_HIP_DEVICES=N (N > 1) it drives N GPUs of the node: the framebuffer's 64x64 tiles are
// dealt round-robin to one context per device and the RGBA8 tiles are gathered on device 0 with
// RCCL send/recv over xGMI before the un-permute kernel writes `img`.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "render_backend.h"
#include <glm/glm.hpp>

struct crt_hip_ctx;

struct RenderHIP : RenderBackend {
    RenderHIP();
    ~RenderHIP() override;

    std::string name() override;
    void initialize(const int fb_width, const int fb_height) override;
    void set_scene(const Scene &scene) override;
    RenderStats render(const glm::vec3 &pos,
                       const glm::vec3 &dir,
                       const glm::vec3 &up,
                       const float fovy,
                       const bool camera_changed,
                       const bool readback_framebuffer) override;

    // Display interop (render_hip_gl.h): render without the per-frame copy into `img` unless asked, and the
    // assembled row-major RGBA8 image as it sits in device 0's HBM (crt_hip_device_framebuffer).
    RenderStats render_to_device(const glm::vec3 &pos, const glm::vec3 &dir, const glm::vec3 &up, const float fovy,
                                 const bool camera_changed, const bool readback_framebuffer);
    void device_framebuffer(void **device_ptr, size_t *pitch_bytes);

private:
    struct MultiGpu; // RCCL communicators + per-device streams (render_hip.cpp)

    std::vector<crt_hip_ctx *> ctxs; // ctxs[0] owns the assembled image
    std::unique_ptr<MultiGpu> multi;
    int fb_width = 0, fb_height = 0;

    void check(crt_hip_ctx *ctx, int rc, const char *what) const;
    void copy_image(bool readback);
    RenderStats render_impl(const glm::vec3 &pos, const glm::vec3 &dir, const glm::vec3 &up, float fovy, bool camera_changed,
                            bool readback);
};
