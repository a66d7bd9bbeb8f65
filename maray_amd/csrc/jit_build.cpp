// jit_build.cpp — HIP source -> gfx950 code objects (product code): hiprtc, the helper processes that run it, the code
// key and the two caches (process, disk).  The back half of the reference's wasmer JIT (Module::new, src/wasm.rs:140-158),
// which compiles its three modules again on every thread of every render (src/render.rs:158-165).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <sys/stat.h>
#include <fcntl.h>
#include <csignal>
#include <unistd.h>
#include <sstream>
#include <spawn.h>
#include <dlfcn.h>
#include <sys/wait.h>
#include <cerrno>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "jit_parts.hpp"
#include "maray_hip.h"

extern "C" const char maray_build_id[];          // _obj/build_id.cpp (Makefile): a hash of this library's sources
extern "C" const char maray_embedded_device_math_h[];
extern "C" const char maray_embedded_libm_h[];
extern "C" const char maray_embedded_libm_tables_h[];

namespace maray {

#define RTC_TRY(expr)                                                                              \
    do {                                                                                           \
        hiprtcResult r_ = (expr);                                                                  \
        if (r_ != HIPRTC_SUCCESS)                                                                  \
            throw Error{MARAY_E_HIP, std::string(#expr) + ": " + hiprtcGetErrorString(r_)};         \
    } while (0)

// The compiler's options, in ONE place: JIT_OPTIONS (jit_parts.hpp, the string that goes into the code key) split into
// words, its -O level replaced by MARAY_JIT_OPT ("-O1" builds faster: 1.7 against 2.6 s for chess, kernel 37.4 against
// 35.6 us), MARAY_JIT_EXTRA ("-mllvm -some-flag ...", a measurement knob) appended.  The in-process compile below and the
// helper processes (which get these very words on their command line) build with exactly what the key was made from.
std::vector<std::string> jit_option_words()
{
    std::vector<std::string> out;
    const char *olevel = getenv("MARAY_JIT_OPT");
    std::istringstream in(JIT_OPTIONS);
    for (std::string w; in >> w;) out.push_back((w.size() == 3 && w[0] == '-' && w[1] == 'O' && olevel && olevel[0] == '-') ? std::string(olevel) : w);
    if (const char *e_ = getenv("MARAY_JIT_EXTRA")) { std::istringstream ex(e_); for (std::string w; ex >> w;) out.push_back(w); }
    return out;
}

void jit_compile(const std::string &src, std::vector<char> &code, std::string &log)
{
    hiprtcProgram prog;
    const char *headers[] = {maray_embedded_device_math_h, maray_embedded_libm_h, maray_embedded_libm_tables_h};
    const char *names[] = {"device_math.h", "maray_libm.h", "maray_libm_tables.h"};
    RTC_TRY(hiprtcCreateProgram(&prog, src.c_str(), "maray_jit.hip", 3, headers, names));
    const std::vector<std::string> words = jit_option_words();
    std::vector<const char *> opts;
    for (const std::string &w : words) opts.push_back(w.c_str());
    hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    size_t ln = 0;
    hiprtcGetProgramLogSize(prog, &ln);
    log.assign(ln, '\0');
    if (ln) hiprtcGetProgramLog(prog, &log[0]);
    if (rc != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        throw Error{MARAY_E_HIP, std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(rc) + "\n" + log};
    }
    size_t n = 0;
    RTC_TRY(hiprtcGetCodeSize(prog, &n));
    code.resize(n);
    RTC_TRY(hiprtcGetCode(prog, code.data()));
    hiprtcDestroyProgram(&prog);
}

// ---- code objects: built once per (program, toolchain), kept in the process and on disk ------------------------
//
// The reference's JIT compiles its three modules again on every thread of every render (src/render.rs:158-165).
// Here a program's two code objects (PIXEL and ROW kernels) are a pure function of the generated sources, the
// compiler options and the hiprtc that builds them: they are built once, shared by every context of the process
// (one per device of a multi-GPU render: the second device loads what the first one built), and stored under
// MARAY_CACHE_DIR (default $XDG_CACHE_HOME/maray_amd or ~/.cache/maray_amd; "off" disables) for the next process.
namespace {

uint64_t fnv1a(const void *data, size_t n, uint64_t h)
{
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

// an unsigned field of the kernel's metadata note (msgpack: the value follows its key)
long code_meta_uint(const std::vector<char> &co, const char *key)
{
    const size_t kn = strlen(key);
    const auto it = std::search(co.begin(), co.end(), key, key + kn);
    if (it == co.end() || (size_t)(co.end() - it) < kn + 5) return -1;
    const unsigned char *v = (const unsigned char *)&*it + kn;
    if (v[0] <= 0x7f) return v[0];
    if (v[0] == 0xcc) return v[1];
    if (v[0] == 0xcd) return ((long)v[1] << 8) | v[2];
    if (v[0] == 0xce) return ((long)v[1] << 24) | ((long)v[2] << 16) | ((long)v[3] << 8) | v[4];
    return -1;
}

std::string cache_dir()
{
    const char *e = getenv("MARAY_CACHE_DIR");
    if (e) {
        if (!e[0] || !strcmp(e, "off") || !strcmp(e, "0")) return "";
        return e;
    }
    if (const char *x = getenv("XDG_CACHE_HOME")) if (x[0]) return std::string(x) + "/maray_amd";
    if (const char *h = getenv("HOME")) if (h[0]) return std::string(h) + "/.cache/maray_amd";
    return "";
}

void mkdirs(const std::string &d)
{
    for (size_t i = 1; i <= d.size(); i++)
        if (i == d.size() || d[i] == '/') (void)mkdir(d.substr(0, i).c_str(), 0777);
}

const uint32_t CACHE_MAGIC = 0x3463726du;    // "mrc4": 10 header words (guard geometry and the two-row flag in the header), checksum over header and code
const int CACHE_HDR = 10;

uint64_t cache_sum(const uint32_t hdr[CACHE_HDR], const JitCode &c)
{
    return fnv1a(c.rows.data(), c.rows.size(), fnv1a(c.pix.data(), c.pix.size(), fnv1a(hdr, 4 * CACHE_HDR, 0xcbf29ce484222325ull)));
}

// The launch geometry comes from this header (the sources it was generated for are not at hand when a file is read), so a
// damaged one must read as a miss, not as a division by zero or a table indexed past its end: every field is range-checked
// and the checksum covers the header too.  guard_w: 64 / 128 / 256 pixels; guard_h: a power of two up to 128 rows (1 = guards
// per row); waves: the occupancies __launch_bounds__ is given.
bool cache_header_ok(const uint32_t hdr[CACHE_HDR])
{
    if (hdr[9] > 1 || (hdr[9] == 1 && hdr[8] < 2)) return false;          // two rows per wavefront need groups of >= 2 rows
    const uint32_t gw = hdr[7], gh = hdr[8];
    return hdr[0] == CACHE_MAGIC && hdr[1] >= 1 && hdr[1] <= 65535 && hdr[2] <= 65535 && (hdr[3] == 0 || hdr[3] == 2 || hdr[3] == 4 || hdr[3] == 6 || hdr[3] == 8) &&
           hdr[4] > 0 && hdr[4] < (1u << 30) && hdr[5] < (1u << 30) && hdr[6] <= 1024 && (gw == 64 || gw == 128 || gw == 256) &&
           gh >= 1 && gh <= 128 && (gh & (gh - 1)) == 0;
}

bool cache_read(const std::string &path, JitCode &c)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint32_t hdr[CACHE_HDR];
    bool ok = fread(hdr, 4, CACHE_HDR, f) == CACHE_HDR && cache_header_ok(hdr);
    if (ok) {
        c.n_row_chunks = hdr[1]; c.n_gjobs = hdr[2]; c.waves = (int)hdr[3];
        c.n_gwords = hdr[6]; c.guard_w = hdr[7]; c.guard_h = hdr[8]; c.rows2 = hdr[9] != 0;
        c.pix.resize(hdr[4]); c.rows.resize(hdr[5]);
        ok = fread(c.pix.data(), 1, c.pix.size(), f) == c.pix.size() && fread(c.rows.data(), 1, c.rows.size(), f) == c.rows.size();
        uint64_t sum = 0;
        ok = ok && fread(&sum, 8, 1, f) == 1 && sum == cache_sum(hdr, c);
    }
    fclose(f);
    return ok;
}

void cache_write(const std::string &dir, const std::string &path, const JitCode &c)
{
    mkdirs(dir);
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;                                       // a read-only or missing cache directory is not an error
    const uint32_t hdr[CACHE_HDR] = {CACHE_MAGIC, c.n_row_chunks, c.n_gjobs, (uint32_t)c.waves, (uint32_t)c.pix.size(), (uint32_t)c.rows.size(),
                                     c.n_gwords, c.guard_w, c.guard_h, c.rows2 ? 1u : 0u};
    const uint64_t sum = cache_sum(hdr, c);
    const bool ok = fwrite(hdr, 4, CACHE_HDR, f) == CACHE_HDR && fwrite(c.pix.data(), 1, c.pix.size(), f) == c.pix.size() &&
                    fwrite(c.rows.data(), 1, c.rows.size(), f) == c.rows.size() && fwrite(&sum, 8, 1, f) == 1;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());      // rename: readers never see half a file
}

std::mutex g_code_mutex;
std::map<std::string, std::shared_future<std::shared_ptr<const JitCode>>> g_code;
std::vector<std::string> g_code_order;              // oldest first: the table keeps the 32 most recent programs (the rest live on disk)

// The code key names the code objects of a program: a hash of the two generated sources, the embedded headers and the
// build options -- whatever changes the kernels changes it.  Generating the sources costs ~0.1 s for chess, so the key of
// a program is remembered under a cheaper name: a hash of the program itself (constants, both sections), the generator
// (MARAY_BUILD_ID: a hash of this library's sources, made by the Makefile), every MARAY_JIT_* knob of the environment and
// the compiler's version -- in the process and, next to the code objects, in <cache>/<name>.key.
struct CodeKey { std::string hex, src_pix, src_rows; uint32_t n_row_chunks = 1, n_gjobs = 0; bool have_src = false; };

std::string hex128(uint64_t h1, uint64_t h2)
{
    char buf[40];
    snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
    return buf;
}

std::string hiprtc_path();

std::string key_salt()
{
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);          // a process that imported PyTorch first compiles with PyTorch's own hiprtc
    // ... and two builds of one version are two compilers: the library's path, size and modification time name the binary
    std::string rtc_id = hiprtc_path();
    struct stat st;
    if (!rtc_id.empty() && stat(rtc_id.c_str(), &st) == 0) rtc_id += ":" + std::to_string((long long)st.st_size) + ":" + std::to_string((long long)st.st_mtime);
    // the library's build id: what a launch does with the kernels (tiles per wavefront, grid shape) is library code, and a
    // profile stamped with a code key has to mean "these kernels, launched this way"
    return std::string(maray_version()) + "|" + maray_build_id + "|hiprtc " + std::to_string(major) + "." + std::to_string(minor) + " " + rtc_id +
           "|" + JIT_OPTIONS + "|" + (getenv("MARAY_JIT_OPT") ? getenv("MARAY_JIT_OPT") : "") + (getenv("MARAY_JIT_EXTRA") ? std::string("|") + getenv("MARAY_JIT_EXTRA") : std::string());
}

std::string program_name(const maray_program &prog)
{
    std::string salt = key_salt() + "|" + maray_build_id;
    std::vector<std::string> knobs;
    for (char **e = environ; e && *e; e++) if (!strncmp(*e, "MARAY_JIT_", 10)) knobs.push_back(*e);
    std::sort(knobs.begin(), knobs.end());
    for (const std::string &kn : knobs) salt += "|" + kn;
    const uint32_t counts[8] = {prog.version, prog.n_consts, prog.n_row_ops, prog.n_row_slots, prog.n_yvals, prog.n_pix_ops, prog.n_pix_slots, prog.n_app};
    uint64_t h[2] = {0xcbf29ce484222325ull, 0x84222325cbf29ce4ull};
    for (uint64_t &x : h) {
        x = fnv1a(salt.data(), salt.size(), x);
        x = fnv1a(counts, sizeof counts, x);
        x = fnv1a(prog.consts, (size_t)prog.n_consts * sizeof(double), x);
        x = fnv1a(prog.row_ops, (size_t)prog.n_row_ops * sizeof(uint64_t), x);
        x = fnv1a(prog.pix_ops, (size_t)prog.n_pix_ops * sizeof(uint64_t), x);
    }
    return hex128(h[0], h[1]);
}

void key_sources(const maray_program &prog, CodeKey &k)
{
    if (k.have_src) return;
    k.src_pix = jit_source(prog);
    if (prog.n_row_ops) k.src_rows = jit_source_rows(prog, &k.n_row_chunks, &k.n_gjobs);
    k.have_src = true;
}

std::mutex g_name_mutex;
std::map<std::string, std::string> g_names;           // program name -> code key

CodeKey code_key(const maray_program &prog)
{
    CodeKey k;
    const std::string name = program_name(prog), dir = cache_dir();
    {
        std::lock_guard<std::mutex> lk(g_name_mutex);
        auto it = g_names.find(name);
        if (it != g_names.end()) { k.hex = it->second; return k; }
    }
    const std::string path = dir.empty() ? "" : dir + "/" + name + ".key";
    if (!path.empty())
        if (FILE *f = fopen(path.c_str(), "rb")) {
            char buf[33] = {0};
            const bool ok = fread(buf, 1, 32, f) == 32 && strspn(buf, "0123456789abcdef") == 32;
            fclose(f);
            if (ok) k.hex = buf;
        }
    if (k.hex.empty()) {
        key_sources(prog, k);
        const std::string salt = key_salt();
        uint64_t h1 = fnv1a(salt.data(), salt.size(), 0xcbf29ce484222325ull), h2 = fnv1a(salt.data(), salt.size(), 0x84222325cbf29ce4ull);
        for (const std::string *t : {&k.src_pix, &k.src_rows}) { h1 = fnv1a(t->data(), t->size() + 1, h1); h2 = fnv1a(t->data(), t->size() + 1, h2); }
        for (const char *hd : {maray_embedded_device_math_h, maray_embedded_libm_h, maray_embedded_libm_tables_h}) { h1 = fnv1a(hd, strlen(hd), h1); h2 = fnv1a(hd, strlen(hd), h2); }
        k.hex = hex128(h1, h2);
        if (!path.empty()) {
            mkdirs(dir);
            const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
            if (FILE *f = fopen(tmp.c_str(), "wb")) {
                const bool ok = fwrite(k.hex.data(), 1, 32, f) == 32;
                if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
            }
        }
    }
    std::lock_guard<std::mutex> lk(g_name_mutex);
    if (g_names.size() > 256) g_names.clear();
    g_names[name] = k.hex;
    return k;
}

// ---- out-of-process builds ---------------------------------------------------------------------------------------
// hiprtc serialises compiles inside a process (two threads: 7.9 s either way for chess, measured), and an LLVM abort
// inside it takes the process down.  So a program's two modules are built by two helper processes side by side
// (maray_jitc, next to this library; it dlopens the very hiprtc this process has loaded -- the compiler's version is
// part of the code key): chess cold 2.6 -> 1.6 s.  MARAY_JIT_HELPER=0, or a helper that is missing or cannot reach the
// compiler: the module is compiled in-process.  A source that does not compile is an error either way, with the
// compiler's log; a helper that dies while compiling is an error too (MARAY_E_HIP; BACKEND_AUTO then takes the interpreter).
std::string self_dir()
{
    Dl_info info;
    if (!dladdr((const void *)&maray_build_id, &info) || !info.dli_fname) return "";
    const std::string p = info.dli_fname;
    const size_t at = p.rfind('/');
    return at == std::string::npos ? "." : p.substr(0, at);
}

std::string hiprtc_path()
{
    Dl_info info;
    if (!dladdr((const void *)&hiprtcCompileProgram, &info) || !info.dli_fname) return "";
    return info.dli_fname;
}

struct HelperJob {
    pid_t pid = -1;
    std::string dir, src_path, out_path;
};

bool read_file(const std::string &path, std::vector<char> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    out.clear();
    for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return true;
}

// starts the helper on `src`; pid stays -1 when it cannot be started.  Source and result live in a directory of their own
// (mkdtemp, 0700): nobody else can put a file or a link where the helper writes and this process reads.
HelperJob helper_start(const std::string &helper, const std::string &rtc, const std::string &src, const char *tag)
{
    HelperJob j;
    const char *tmp = getenv("TMPDIR");
    char path[512];
    snprintf(path, sizeof path, "%s/maray_jit_%ld_%s_XXXXXX", (tmp && tmp[0]) ? tmp : "/tmp", (long)getpid(), tag);
    if (!mkdtemp(path)) return j;
    j.dir = path;
    j.src_path = j.dir + "/kernel.hip";
    j.out_path = j.dir + "/kernel.out";
    const int fd = open(j.src_path.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    if (fd < 0) return j;
    const bool ok = write(fd, src.data(), src.size()) == (ssize_t)src.size();
    close(fd);
    if (!ok) return j;
    const std::vector<std::string> words = jit_option_words();          // the helper compiles with the words the key was made from
    std::vector<char *> argv = {(char *)helper.c_str(), (char *)rtc.c_str(), (char *)j.src_path.c_str(), (char *)j.out_path.c_str()};
    for (const std::string &w : words) argv.push_back((char *)w.c_str());
    argv.push_back(nullptr);
    pid_t pid = -1;
    if (posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv.data(), environ) == 0) j.pid = pid;
    return j;
}

// What became of a helper.  OK: `code` holds the code object.  REJECTED: the source does not compile, `log` holds the
// compiler's errors.  ABSENT: the helper never got as far as the compiler (not started, could not load hiprtc or read its
// input): the caller compiles in-process.  DIED: the helper was running the compiler and ended by a signal or an
// unexpected status -- an abort inside LLVM is what the helper exists to keep out of the caller, so this is an error,
// never a reason to run the same compile in-process; `log` says how it ended and where its source was kept.
enum HelperEnd { HELPER_OK, HELPER_REJECTED, HELPER_ABSENT, HELPER_DIED };

HelperEnd helper_finish(HelperJob &j, std::vector<char> &code, std::string &log)
{
    HelperEnd end = HELPER_ABSENT;
    if (j.pid > 0) {
        int status = 0;
        pid_t r;
        do r = waitpid(j.pid, &status, 0); while (r < 0 && errno == EINTR);
        std::vector<char> out;
        if (r != j.pid) { end = HELPER_DIED; log = "waitpid failed"; }
        else if (WIFSIGNALED(status)) { end = HELPER_DIED; log = "signal " + std::to_string(WTERMSIG(status)) + (WTERMSIG(status) == SIGABRT ? " (abort)" : ""); }
        else if (!WIFEXITED(status)) { end = HELPER_DIED; log = "wait status " + std::to_string(status); }
        else switch (WEXITSTATUS(status)) {
        case 0:
            if (read_file(j.out_path, out) && out.size() >= 64 && memcmp(out.data(), "\177ELF", 4) == 0) { code.swap(out); end = HELPER_OK; }
            else { end = HELPER_DIED; log = "exit status 0 without a code object"; }
            break;
        case 3:
            if (read_file(j.out_path, out)) { log.assign(out.begin(), out.end()); end = HELPER_REJECTED; }
            else { end = HELPER_DIED; log = "exit status 3 without a compiler log"; }
            break;
        case 2: case 4: case 5: case 127: end = HELPER_ABSENT; break;      // usage / no hiprtc / no input / not executable: the compiler never ran
        default: end = HELPER_DIED; log = "exit status " + std::to_string(WEXITSTATUS(status));
        }
    }
    if (end == HELPER_DIED && !j.src_path.empty()) log += "; source kept in " + j.src_path;      // for the bug report
    else if (!j.src_path.empty()) (void)unlink(j.src_path.c_str());
    if (!j.out_path.empty()) (void)unlink(j.out_path.c_str());
    if (!j.dir.empty() && end != HELPER_DIED) (void)rmdir(j.dir.c_str());
    return end;
}

std::shared_ptr<const JitCode> build_code(const maray_program &prog, CodeKey &k)
{
    auto c = std::make_shared<JitCode>();
    const std::string dir = cache_dir(), path = dir.empty() ? "" : dir + "/" + k.hex + ".mrco";
    if (!path.empty() && cache_read(path, *c)) { c->from_disk = true; return c; }
    key_sources(prog, k);          // (a key that came from its cheaper name has no sources yet)
    std::string log;
    // both modules in helper processes, side by side
    bool have_pix = false, have_rows = false;
    {
        const char *e = getenv("MARAY_JIT_HELPER");
        const std::string helper = self_dir() + "/maray_jitc", rtc = hiprtc_path();
        if (!(e && e[0] == '0') && !rtc.empty() && access(helper.c_str(), X_OK) == 0) {
            HelperJob jp = helper_start(helper, rtc, k.src_pix, "pix"), jr;
            if (prog.n_row_ops) jr = helper_start(helper, rtc, k.src_rows, "rows");
            std::string lp, lr;
            const HelperEnd rp = helper_finish(jp, c->pix, lp), rr = prog.n_row_ops ? helper_finish(jr, c->rows, lr) : HELPER_ABSENT;
            if (rp == HELPER_REJECTED) throw Error{MARAY_E_HIP, "hiprtcCompileProgram (maray_jitc): the PIXEL kernel does not compile\n" + lp};
            if (rr == HELPER_REJECTED) throw Error{MARAY_E_HIP, "hiprtcCompileProgram (maray_jitc): the ROW kernel does not compile\n" + lr};
            if (rp == HELPER_DIED) throw Error{MARAY_E_HIP, "the compiler aborted on the PIXEL kernel (maray_jitc: " + lp + ")"};
            if (rr == HELPER_DIED) throw Error{MARAY_E_HIP, "the compiler aborted on the ROW kernel (maray_jitc: " + lr + ")"};
            have_pix = rp == HELPER_OK; have_rows = rr == HELPER_OK;
        }
    }
    // Occupancy: the generator's own choice (8 or 6 waves per SIMD, jit_source), then 6 / 4 / 2 (<= 80 / 128 / 256 VGPRs)
    // until a build needs no scratch: spilled VGPRs are HBM traffic.
    const int ladder[] = {0, 6, 4, 2};
    for (int i = 0; i < 4; i++) {
        if (!(i == 0 && have_pix)) jit_compile(i == 0 ? k.src_pix : jit_source(prog, ladder[i]), c->pix, log);
        c->waves = ladder[i];
        if (code_meta_uint(c->pix, ".private_segment_fixed_size") <= 0) break;
    }
    if (prog.n_row_ops && !have_rows) jit_compile(k.src_rows, c->rows, log);
    c->n_row_chunks = k.n_row_chunks; c->n_gjobs = k.n_gjobs;
    if (prog.n_row_ops) {          // (the guard plan is a walk over the ROW tape: once here, not in every context's creation)
        const GuardGeom geom = jit_guard_geom(prog);
        c->n_gwords = jit_guard_words(prog); c->guard_w = geom.gw; c->guard_h = geom.gh; c->rows2 = jit_rows2(prog);
    }
    if (!path.empty()) cache_write(dir, path, *c);
    return c;
}

}   // namespace

std::string jit_code_key(const maray_program &prog) { return code_key(prog).hex; }

bool jit_code_is_cached(const maray_program &prog)
{
    const CodeKey k = code_key(prog);
    {
        std::lock_guard<std::mutex> lk(g_code_mutex);
        if (g_code.count(k.hex)) return true;
    }
    const std::string dir = cache_dir();
    return !dir.empty() && access((dir + "/" + k.hex + ".mrco").c_str(), R_OK) == 0;
}

std::shared_ptr<const JitCode> jit_code_for(const maray_program &prog)
{
    CodeKey k = code_key(prog);
    std::promise<std::shared_ptr<const JitCode>> mine;
    std::shared_future<std::shared_ptr<const JitCode>> fut;
    bool build = false;
    {
        std::lock_guard<std::mutex> lk(g_code_mutex);
        auto it = g_code.find(k.hex);
        if (it != g_code.end()) fut = it->second;
        else {
            fut = mine.get_future().share(); g_code.emplace(k.hex, fut); build = true;
            g_code_order.push_back(k.hex);
            while (g_code_order.size() > 32) {          // contexts hold their code objects themselves (shared_ptr)
                g_code.erase(g_code_order.front());
                g_code_order.erase(g_code_order.begin());
            }
        }
    }
    if (build) {
        try { mine.set_value(build_code(prog, k)); }
        catch (...) {
            mine.set_exception(std::current_exception());
            std::lock_guard<std::mutex> lk(g_code_mutex);
            g_code.erase(k.hex);                       // a failed build is not remembered (the waiters still see its error)
        }
    }
    return fut.get();         // the contexts of a multi-GPU render: the first builds, the others wait here
}

}   // namespace maray
