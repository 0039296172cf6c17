// Decodes every file named on the command line with the harness's JPEG decoder (csrc/jpeg_reader.cpp) and prints one line per
// file: "ok W H checksum" or "refused". Built by tests/test_texture_decoders.py with -fsanitize=address,undefined and
// -fno-sanitize-recover: a heap overrun or an undefined shift on a crafted file ends the process with a report instead of
// going unnoticed. TEST INFRASTRUCTURE.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/crt_scene_io.h"

int main(int argc, char **argv)
{
    for (int i = 1; i < argc; ++i) {
        FILE *f = std::fopen(argv[i], "rb");
        if (!f) {
            std::printf("missing\n");
            continue;
        }
        std::vector<uint8_t> bytes;
        uint8_t buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) {
            bytes.insert(bytes.end(), buf, buf + n);
        }
        std::fclose(f);
        int32_t w = 0, h = 0;
        uint8_t *rgba = nullptr;
        if (crt_image_decode_jpeg(bytes.data(), bytes.size(), &w, &h, &rgba) != 0) {
            std::printf("refused\n");
            continue;
        }
        uint64_t sum = 0;
        for (int64_t k = 0; k < (int64_t)w * h * 4; ++k) {
            sum = sum * 1099511628211ull + rgba[k];
        }
        std::printf("ok %d %d %llu\n", w, h, (unsigned long long)sum);
        crt_image_free(rgba);
    }
    return 0;
}
