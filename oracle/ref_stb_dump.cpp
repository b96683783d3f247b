/* ref_stb_dump — TEST INFRASTRUCTURE.  Decodes an image with the REAL stb_image of the reference tree
 * (reference external/stb/stb_image.h, included from where it lies; nothing is copied) exactly the way the
 * reference's core::file::createTextureImage does (src/core/file.cppm:272-291: flip vertically on load,
 * STBI_grey or STBI_rgb_alpha), and writes "w h c\n" + the raw bytes to stdout.  Used by
 * tests/golden/make_fixtures.py to pin realtimeraytracer_amd/csrc/host/image_io.hpp.
 *   usage: stb_dump <file> <1|4> */
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: stb_dump <file> <1|4>\n"); return 2; }
    const int want = std::atoi(argv[2]);
    stbi_set_flip_vertically_on_load(true);
    int w = 0, h = 0, c = 0;
    unsigned char* px = stbi_load(argv[1], &w, &h, &c, want);
    if (!px) { std::fprintf(stderr, "stbi_load failed: %s\n", stbi_failure_reason()); return 1; }
    std::printf("%d %d %d\n", w, h, want);
    std::fwrite(px, 1, (size_t)w * h * want, stdout);
    stbi_image_free(px);
    return 0;
}
