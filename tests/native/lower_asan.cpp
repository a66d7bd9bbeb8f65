// lower_asan.cpp — host-only sanitizer harness for the product's reader, fix_color, simplify, lowering and ROW-tape cuts
// (test infrastructure).  Built with -fsanitize=address,undefined together with scene.cpp and lower.cpp;
// reads .maray files given on the command line, lowers each with several option sets and validates the
// resulting programs structurally.
#include <pthread.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "backend.hpp"
#include "expr.hpp"
#include "lower.hpp"

using namespace maray;

namespace maray { void validate_program(const maray_program &p); }

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> b;
    FILE *f = fopen(path, "rb");
    if (!f) return b;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + n);
    fclose(f);
    return b;
}

static int g_argc; static char **g_argv; static int g_rc;
static int real_main(int argc, char **argv);
static void *big_stack(void *) { g_rc = real_main(g_argc, g_argv); return nullptr; }

// everything runs on a thread with a large stack, like the library's own entry points (the reader, the simplifier and
// the lowering recurse over trees)
int main(int argc, char **argv)
{
    g_argc = argc; g_argv = argv;
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, (size_t)1 << 30);
    pthread_t th;
    if (pthread_create(&th, &at, big_stack, nullptr) != 0) return 2;
    pthread_join(th, nullptr);
    return g_rc;
}

static int real_main(int argc, char **argv)
{
    int lowered = 0, rejected = 0;
    for (int i = 1; i < argc; i++) {
        std::vector<uint8_t> bytes = slurp(argv[i]);
        // also feed every truncation point class: the reader must fail cleanly, never read out of bounds
        for (size_t cut : {bytes.size(), bytes.size() / 2, (size_t)9, (size_t)3}) {
            if (cut > bytes.size()) continue;
            Scene s;
            try { scene_decode(bytes.data(), cut, s); }
            catch (const Error &) { rejected++; continue; }
            std::vector<uint8_t> again;
            scene_encode(s, again);
            if (bytes.size() < (1u << 20)) {            // the authoring-time simplifier on a copy (small scenes: it is recursive)
                Scene c = s;
                try { scene_simplify(c); scene_encode(c, again); }
                catch (const Error &e) { if (e.code != MARAY_E_LIMIT) { fprintf(stderr, "%s: simplify: %s\n", argv[i], e.msg.c_str()); return 1; } }
            }
            for (int variant = 0; variant < 4; variant++) {
                maray_lower_opts o;
                memset(&o, 0, sizeof o);
                o.hoist_rows = variant != 1;
                o.no_skips = variant == 2;
                o.no_row_guards = variant == 3;
                Tape t;
                try { lower_scene(s, o, t); }
                catch (const Error &e) { if (e.code != MARAY_E_ALIASED && e.code != MARAY_E_CYCLE) { fprintf(stderr, "%s: %s\n", argv[i], e.msg.c_str()); return 1; } rejected++; continue; }
                validate_program(t.program());
                // the cuts the evaluators make of the ROW section (row_split.cpp): every job of 8 y values, both slot allocators
                const maray_program P = t.program();
                if (P.n_row_ops) {
                    const RowTapeDeps deps = row_tape_deps(P);
                    for (uint32_t first = 0; first < P.n_yvals; first += 8) {
                        std::vector<uint32_t> outs;
                        for (uint32_t o : deps.outs) { const uint32_t k = MARAY_INS_AUX(P.row_ops[o]); if (k >= first && k < first + 8) outs.push_back(o); }
                        std::vector<uint64_t> a = compact_tape(row_tape_cone(P, deps, outs, nullptr)), b = a;
                        const uint32_t na = renumber_slots(a), nb = reschedule_tape(b);
                        if (nb > na || b.size() > a.size()) { fprintf(stderr, "%s: cone %u: %u > %u slots\n", argv[i], first, nb, na); return 1; }
                    }
                    (void)any_guard_reads_y(P); (void)numeric_yvals(P);
                }
                lowered++;
            }
        }
    }
    printf("lowered %d programs, rejected %d inputs\n", lowered, rejected);
    return 0;
}
