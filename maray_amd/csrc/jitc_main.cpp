// maray_jitc HIPRTC_LIB SOURCE OUT OPTION... — compiles one generated kernel source to a gfx950 code object with the hiprtc
// library the caller names (the one the calling process has loaded: its version is part of the code key), in a process of
// its own: libmaray_hip.so starts one per module so that the PIXEL and the ROW kernels of a program build side by side
// (hiprtc serialises compiles inside a process), and an LLVM abort ends this process, not the caller's.
// Exit status 0: OUT holds the code object; 3: the source does not compile, OUT holds the log; 2, 4, 5: the helper never
// reached the compiler (usage, no hiprtc, no input: the caller then compiles in-process); a signal or anything else: it
// died compiling -- the caller reports MARAY_E_HIP and does NOT repeat the compile in its own process.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" const char maray_embedded_device_math_h[];
extern "C" const char maray_embedded_libm_h[];
extern "C" const char maray_embedded_libm_tables_h[];

namespace {

typedef struct _hiprtcProgram *hiprtcProgram;
typedef int (*create_fn)(hiprtcProgram *, const char *, const char *, int, const char **, const char **);
typedef int (*compile_fn)(hiprtcProgram, int, const char **);
typedef int (*size_fn)(hiprtcProgram, size_t *);
typedef int (*get_fn)(hiprtcProgram, char *);
typedef int (*destroy_fn)(hiprtcProgram *);

bool write_file(const char *path, const char *data, size_t n)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    const bool ok = fwrite(data, 1, n, f) == n;
    return fclose(f) == 0 && ok;
}

}   // namespace

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: maray_jitc HIPRTC_LIB SOURCE OUT OPTION...\n"); return 2; }
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "maray_jitc: %s\n", dlerror()); return 4; }
    const create_fn create = (create_fn)dlsym(lib, "hiprtcCreateProgram");
    const compile_fn compile = (compile_fn)dlsym(lib, "hiprtcCompileProgram");
    const size_fn log_size = (size_fn)dlsym(lib, "hiprtcGetProgramLogSize"), code_size = (size_fn)dlsym(lib, "hiprtcGetCodeSize");
    const get_fn get_log = (get_fn)dlsym(lib, "hiprtcGetProgramLog"), get_code = (get_fn)dlsym(lib, "hiprtcGetCode");
    const destroy_fn destroy = (destroy_fn)dlsym(lib, "hiprtcDestroyProgram");
    if (!create || !compile || !log_size || !code_size || !get_log || !get_code || !destroy) { fprintf(stderr, "maray_jitc: %s is not a hiprtc\n", argv[1]); return 4; }
    std::string src;
    {
        FILE *f = fopen(argv[2], "rb");
        if (!f) { perror(argv[2]); return 5; }
        char buf[1 << 16];
        for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) src.append(buf, n);
        fclose(f);
    }
    const char *headers[] = {maray_embedded_device_math_h, maray_embedded_libm_h, maray_embedded_libm_tables_h};
    const char *names[] = {"device_math.h", "maray_libm.h", "maray_libm_tables.h"};
    // MARAY_JITC_TEST_ABORT=1 (tests): end the way an LLVM abort inside hiprtc ends this process, once the compiler is at hand
    if (const char *e_ = getenv("MARAY_JITC_TEST_ABORT")) if (e_[0] == '1') abort();
    hiprtcProgram prog;
    if (create(&prog, src.c_str(), "maray_jit.hip", 3, headers, names) != 0) return 6;
    // the options are the caller's (jit_option_words, jit_build.cpp: the very words its code key was made from), one per argument
    std::vector<const char *> opts(argv + 4, argv + argc);
    const int rc = compile(prog, (int)opts.size(), opts.data());
    if (rc != 0) {
        size_t ln = 0;
        log_size(prog, &ln);
        std::string log(ln, '\0');
        if (ln) get_log(prog, &log[0]);
        write_file(argv[3], log.data(), log.size());
        return 3;
    }
    size_t n = 0;
    if (code_size(prog, &n) != 0) return 6;
    std::vector<char> code(n);
    if (get_code(prog, code.data()) != 0) return 6;
    destroy(&prog);
    return write_file(argv[3], code.data(), code.size()) ? 0 : 7;
}
