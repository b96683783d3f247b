/* ref_ltc_dump — TEST INFRASTRUCTURE.  Writes the two 64x64 RGBA32F LTC tables of the reference tree
 * (reference external/LUT/ltc_matrix.h, included from where it lies; the data is not copied into this repository) as
 * 2 x 16384 little-endian floats to stdout.  The reference uploads exactly these arrays as texSamplers[0] and [1]
 * (src/app/setup/create_scene.cppm:162-214); tests/test_host_scene.py uses the dump, where the reference exists, to check
 * the Appendix-B known answers and to render the analytic image through the real tables. */
#include "ltc_matrix.h"
#include <cstdio>
int main() {
    static_assert(sizeof(LTC1) == 16384 * sizeof(float) && sizeof(LTC2) == 16384 * sizeof(float), "64 x 64 x 4 floats each");
    std::fwrite(LTC1, sizeof(float), 16384, stdout);
    std::fwrite(LTC2, sizeof(float), 16384, stdout);
    return 0;
}
