// png_asan.cpp — host-only sanitizer harness for the product's PNG reader / writer (test infrastructure).
// Built with -fsanitize=address,undefined together with maray_amd/csrc/png.cpp.  Reads every file given on the
// command line with maray_png_read: crafted files (an IHDR that promises more than its IDAT holds, sizes whose
// products overflow, truncated chunks) must come back as error codes -- no out-of-bounds access, no allocation the
// size of the lie, no exception across the C boundary; well-formed files must round-trip through maray_png_write.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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
        const int rc = maray_png_read(argv[i], &rgb, &w, &h);
        printf("%s: rc %d %ux%u %s\n", argv[i], rc, w, h, rc ? maray::g_last.c_str() : "");
        const bool expect_ok = strstr(argv[i], "good") != nullptr;
        if (expect_ok != (rc == 0)) bad++;
        if (rc == 0) {
            const std::string out = std::string(argv[i]) + ".out.png";
            if (maray_png_write(out.c_str(), rgb, w, h)) bad++;
            uint8_t *again = nullptr;
            uint32_t w2 = 0, h2 = 0;
            if (maray_png_read(out.c_str(), &again, &w2, &h2) || w2 != w || h2 != h || memcmp(again, rgb, (size_t)w * h * 3)) bad++;
            free(again);
        }
        free(rgb);
    }
    // the writer refuses sizes whose raster it could not hold
    if (maray_png_write("/dev/null", (const uint8_t *)"", 0x200000u, 0x200000u) == 0) bad++;
    if (maray_png_write(nullptr, nullptr, 1, 1) == 0) bad++;
    printf("%s\n", bad ? "FAILED" : "png ok");
    return bad ? 1 : 0;
}
