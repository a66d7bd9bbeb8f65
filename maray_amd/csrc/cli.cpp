// cli.cpp — `maray` command line (product code).  Mirrors examples/maray.rs:
//   maray -c N -i in.maray -o out.png [-t tex.png ...]        (:9-47)
// plus --gpus N, --backend {tape,tape-smem,jit}.  -c/--cpus is parsed and
// ignored, exactly like the reference (`_cpus`, examples/maray.rs:55).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "maray_hip.h"

static void usage()
{
    fprintf(stderr,
            "Maray 0.3 (MI355X)\nJIT Ray Tracing using basic math\n\n"
            "Usage: maray [OPTIONS] --input <input> --output <output>\n\n"
            "  -c, --cpus <cpus>            Number of CPU cores (accepted, unused)\n"
            "  -i, --input <input>          Input file `*.maray`\n"
            "  -o, --output <output>        Output file `*.png`\n"
            "  -t, --textures <textures>... Texture files (PNG, BMP, PNM, TGA, QOI, farbfeld, GIF, TIFF)\n"
            "      --gpus <n>               Number of MI355X devices (default: all)\n"
            "      --backend <b>            auto | jit | tape | tape-smem (default: auto)\n");
}

int main(int argc, char **argv)
{
    std::string input, output;
    std::vector<std::string> textures;
    maray_gen_opts go;
    memset(&go, 0, sizeof go);
    go.backend = MARAY_BACKEND_AUTO;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
        if (a == "-c" || a == "--cpus") (void)strtoul(val(), nullptr, 10);
        else if (a == "-i" || a == "--input") input = val();
        else if (a == "-o" || a == "--output") output = val();
        else if (a == "-t" || a == "--textures") { while (i + 1 < argc && argv[i + 1][0] != '-') textures.push_back(argv[++i]); }
        else if (a == "--gpus") go.n_devices = (uint32_t)strtoul(val(), nullptr, 10);
        else if (a == "--backend") {
            std::string b = val();
            if (b == "tape") go.backend = MARAY_BACKEND_TAPE;
            else if (b == "tape-smem") go.backend = MARAY_BACKEND_TAPE_SMEM;
            else if (b == "jit") go.backend = MARAY_BACKEND_JIT;
            else if (b == "auto") go.backend = MARAY_BACKEND_AUTO;
            else { usage(); return 2; }
        } else if (a == "-h" || a == "--help") { usage(); return 0; }
        else { usage(); return 2; }
    }
    if (input.empty() || output.empty()) { usage(); return 2; }

    maray_scene *scene = nullptr;
    if (maray_scene_open(input.c_str(), &scene)) { fprintf(stderr, "Error: %s\n", maray_last_error()); return 1; }
    std::vector<maray_texture> tex;
    std::vector<uint8_t *> rasters;
    for (const std::string &t : textures) {
        uint8_t *rgb = nullptr; uint32_t w = 0, h = 0;
        if (maray_image_read(t.c_str(), &rgb, &w, &h)) { fprintf(stderr, "Error: %s: %s\n", t.c_str(), maray_last_error()); return 1; }
        rasters.push_back(rgb);
        tex.push_back(maray_texture{rgb, w, h});
    }
    maray_report rep{MARAY_REPORT_DURATION_MS, 500};   // Report::Duration(500 ms), examples/maray.rs:77-79
    int rc = maray_gen(scene, tex.data(), (uint32_t)tex.size(), &go, rep, output.c_str());
    if (rc) fprintf(stderr, "Error: %s\n", maray_last_error());
    for (auto p : rasters) maray_free(p);
    maray_scene_free(scene);
    return rc ? 1 : 0;
}
