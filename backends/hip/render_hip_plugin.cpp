// render_hip_plugin.cpp — the four plugin callbacks + populate_plugin_functions
// (reference util/render_plugin.h:23-63). Same display path as the Embree plugin
// (backends/embree/render_embree_plugin.cpp:7-25): an OpenGL window and GLDisplay blitting the
// CPU-side RenderBackend::img.
#include <SDL.h>
#include "display/gldisplay.h"
#include "imgui.h"
#include "render_hip.h"
#ifdef CRT_HIP_GL_INTEROP
#include "render_hip_gl.h"
#endif
#include "render_plugin.h"

static uint32_t hip_window_flags()
{
    return SDL_WINDOW_OPENGL;
}

static void hip_set_imgui_context(ImGuiContext *context)
{
    ImGui::SetCurrentContext(context);
}

static std::unique_ptr<Display> hip_make_display(SDL_Window *window)
{
    return std::make_unique<GLDisplay>(window);
}

static std::unique_ptr<RenderBackend> hip_make_renderer(Display *display)
{
#ifdef CRT_HIP_GL_INTEROP
    // like backends/optix/render_optix_plugin.cpp: the native path only when the display is the GL one
    if (dynamic_cast<GLDisplay *>(display) != nullptr) {
        return std::make_unique<RenderHIPGL>();
    }
#else
    (void)display;
#endif
    return std::make_unique<RenderHIP>();
}

POPULATE_PLUGIN_FUNCTIONS(hip_window_flags, hip_set_imgui_context, hip_make_display, hip_make_renderer)
