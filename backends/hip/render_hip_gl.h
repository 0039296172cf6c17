// render_hip_gl.h — display interop for the `hip` backend: `struct RenderHIPGL :
// GLNativeRenderer`, the reference's hook for backends that hand the display a GL texture instead of
// the host-side `img` (util/display/gldisplay.h:35-37; GLDisplay::display picks it with a
// dynamic_cast, gldisplay.cpp:105-124; the OptiX backend is the reference's user of it,
// backends/optix/render_optix.cpp:104-121,410-426).
//
// Type-checked in this repository against the reference's real gldisplay.h + glad and ROCm's hip_gl_interop.h
// (`make -C oracle boundary_check`, tests/test_boundary_headers.py); it cannot RUN here -- no GL context exists in
// the image or on the GPU box -- and `backends/hip/CMakeLists.txt` builds it with -DCRT_HIP_GL_INTEROP=ON inside
// a ChameleonRT tree. What it stands on IS executed: crt_hip_device_framebuffer (the row-major RGBA8 image in
// HBM, tests/test_gpu_edge_cases.py) and render() with readback = false.
#pragma once
#include <hip/hip_runtime_api.h>

#include "display/gldisplay.h"
#include "render_hip.h"

struct RenderHIPGL : GLNativeRenderer {
    RenderHIPGL();
    ~RenderHIPGL() override;

    std::string name() override;
    void initialize(const int fb_width, const int fb_height) override;
    void set_scene(const Scene &scene) override;
    RenderStats render(const glm::vec3 &pos,
                       const glm::vec3 &dir,
                       const glm::vec3 &up,
                       const float fovy,
                       const bool camera_changed,
                       const bool readback_framebuffer) override;

private:
    RenderHIP inner; // all rendering; this class only moves the finished image into the GL texture
    hipGraphicsResource_t hip_display_texture = nullptr;
    int width = 0, height = 0;
};
