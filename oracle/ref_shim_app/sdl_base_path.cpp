// Stand-in for the ONE SDL2 function the reference's plugin loader calls (util/render_plugin.cpp:16:
// `SDL_GetBasePath()` -> the directory of the executable, with a trailing separator; SDL2 is third party and
// absent from this image). TEST INFRASTRUCTURE: linked into oracle/_ref/crt_bench only.
#include <string>
#include <unistd.h>

extern "C" char *SDL_GetBasePath(void)
{
    static std::string dir;
    char buf[4096];
    const ssize_t n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
    dir.assign(buf, n > 0 ? size_t(n) : 0);
    dir.erase(dir.find_last_of('/') + 1);
    return &dir[0];
}
