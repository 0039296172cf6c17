// Stand-in for <SDL.h> (SDL2 is third party and absent from this image): the three names the reference's
// util/display/gldisplay.h, display.h and a plugin's four callbacks (util/render_plugin.h:23-41) mention.
// TEST INFRASTRUCTURE: only `make -C oracle boundary_check` puts it on an include path, to type-check
// backends/hip/render_hip_plugin.cpp and render_hip_gl.cpp against the reference's real display headers.
#pragma once
#include <cstdint>
struct SDL_Window;
typedef void *SDL_GLContext;
enum { SDL_WINDOW_OPENGL = 0x00000002 };
extern "C" char *SDL_GetBasePath(void);
