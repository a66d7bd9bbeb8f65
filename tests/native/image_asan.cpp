// image_asan.cpp — host-only sanitizer harness for maray_image_read (test infrastructure): png.cpp + image.cpp built with
// -fsanitize=address,undefined.  Every file named on the command line is read; files whose name holds "good" must decode,
// the others must come back as error codes -- no out-of-bounds access, no allocation the size of a header's lie.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "backend.hpp"
#include "maray_hip.h"

namespace maray {
static std::string g_last;
void set_last_error(const std::string &m) { g_last = m; }
}   // namespace maray
extern "C" void maray_free(void *p) { free(p); }

int main(int argc, char **argv)
{
    int bad = 0;
    for (int i = 1; i < argc; i++) {
        uint8_t *rgb = nullptr;
        uint32_t w = 0, h = 0;
        const int rc = maray_image_read(argv[i], &rgb, &w, &h);
        unsigned long sum = 0;
        if (rc == 0) for (size_t k = 0; k < (size_t)w * h * 3; k++) sum += rgb[k];      // touch every byte of the raster
        printf("%s: rc %d %ux%u %lu %s\n", argv[i], rc, w, h, sum, rc ? maray::g_last.c_str() : "");
        if ((strstr(argv[i], "good") != nullptr) != (rc == 0)) bad++;
        free(rgb);
    }
    printf("%s\n", bad ? "FAILED" : "images ok");
    return bad ? 1 : 0;
}
