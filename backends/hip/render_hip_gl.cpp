// render_hip_gl.cpp — see render_hip_gl.h. HIP's GL interop mirrors CUDA's call for call
// (hip/hip_gl_interop.h): register the texture once per resize, per frame map it, copy the image
// device-to-device into its array, unmap. The host image is only produced when the app asks for it
// (`readback_framebuffer`: screenshots, -validation dumps; main.cpp:306-325).
#include "render_hip_gl.h"

#include <hip/hip_runtime_api.h>
// after hip_runtime_api.h (it uses hipError_t without including it) and after glad (gldisplay.h), which must precede any GL header
#include <hip/hip_gl_interop.h>

#include <stdexcept>

#include "crt_hip.h"

static void hip_ok(hipError_t e, const char *what)
{
    if (e != hipSuccess) {
        throw std::runtime_error(std::string("RenderHIPGL: ") + what + ": " + hipGetErrorString(e));
    }
}

RenderHIPGL::RenderHIPGL() = default;

RenderHIPGL::~RenderHIPGL()
{
    if (gl_display_texture != GLuint(-1)) {
        (void)hipGraphicsUnregisterResource(hip_display_texture);
        glDeleteTextures(1, &gl_display_texture);
    }
}

std::string RenderHIPGL::name()
{
    return inner.name() + " + GL interop";
}

void RenderHIPGL::initialize(const int fb_width, const int fb_height)
{
    inner.initialize(fb_width, fb_height);
    width = fb_width;
    height = fb_height;
    img.resize(size_t(fb_width) * fb_height);
    if (gl_display_texture != GLuint(-1)) {
        hip_ok(hipGraphicsUnregisterResource(hip_display_texture), "hipGraphicsUnregisterResource");
        glDeleteTextures(1, &gl_display_texture);
    }
    glGenTextures(1, &gl_display_texture);
    glBindTexture(GL_TEXTURE_2D, gl_display_texture);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, fb_width, fb_height, 0, GL_RGBA, GL_UNSIGNED_BYTE, nullptr);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    hip_ok(hipGraphicsGLRegisterImage(&hip_display_texture, gl_display_texture, GL_TEXTURE_2D, hipGraphicsRegisterFlagsWriteDiscard),
           "hipGraphicsGLRegisterImage");
}

void RenderHIPGL::set_scene(const Scene &scene)
{
    inner.set_scene(scene);
    samples_per_pixel = inner.samples_per_pixel;
}

RenderStats RenderHIPGL::render(const glm::vec3 &pos, const glm::vec3 &dir, const glm::vec3 &up, const float fovy,
                                const bool camera_changed, const bool readback_framebuffer)
{
    const RenderStats stats = inner.render_to_device(pos, dir, up, fovy, camera_changed, readback_framebuffer);
    void *d_img = nullptr;
    size_t pitch = 0;
    inner.device_framebuffer(&d_img, &pitch);
    hip_ok(hipGraphicsMapResources(1, &hip_display_texture, nullptr), "hipGraphicsMapResources");
    hipArray_t array = nullptr;
    hip_ok(hipGraphicsSubResourceGetMappedArray(&array, hip_display_texture, 0, 0), "hipGraphicsSubResourceGetMappedArray");
    hip_ok(hipMemcpy2DToArray(array, 0, 0, d_img, pitch, size_t(width) * sizeof(uint32_t), size_t(height), hipMemcpyDeviceToDevice),
           "hipMemcpy2DToArray");
    hip_ok(hipGraphicsUnmapResources(1, &hip_display_texture, nullptr), "hipGraphicsUnmapResources");
    if (readback_framebuffer) {
        img = inner.img;
    }
    return stats;
}
