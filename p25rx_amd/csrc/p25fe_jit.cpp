// p25fe_jit.cpp -- front-end kernels specialised for a caller's numbers: hipRTC + an on-disk cache.
//
// Host-only C++ (no device code here).  The kernel source is the library's own: p25fe_kernels.hip, p25fe_recv.hip,
// p25fe.h and p25fe_spec.h are embedded as strings at build time (p25fe_embed.inc, written by tools/embed_src.py), so a
// deployed libp25fe.so needs no source tree -- only libhiprtc (every ROCm installation carries it; loaded on first use).  The reference
// fixes the same numbers at ITS compile time (type-level FIR tables, src/demod.rs:27-29; FmDemod::new(5000, 48000),
// src/demod.rs:54; the rtlsdr_iq table, src/demod.rs:83).
#include "p25fe_jit.h"

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <hip/hiprtc.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include "p25fe_embed.inc"      // P25FE_SRC_KERNELS, P25FE_SRC_RECV, P25FE_SRC_P25FE_H, P25FE_SRC_SPEC_H

namespace p25jit {

const char* const KERNEL_NAMES[2][3] = {
    {"p25jit_k1_cf32_lin", "p25jit_k1_cf32_pl", "p25jit_chunk_cf32"},
    {"p25jit_k1_u8_lin", "p25jit_k1_u8_pl", "p25jit_chunk_u8"},
};

// the same code-generation options as the library's own build (p25rx_amd/csrc/Makefile): fma only where the source says
// fma, correctly rounded division -- the specialised kernels must produce the bits of the built-in ones
static const char* const OPTIONS[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                                      "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-function"};
static const int N_OPTIONS = (int)(sizeof OPTIONS / sizeof OPTIONS[0]);

// the translation unit: generated numbers, the kernel source, six entry points with C names
static const char WRAPPER[] =
    "#include \"p25fe_jit_spec.h\"\n"
    "#include \"p25fe_kernels.hip\"\n"
    "using namespace p25k;\n"
    "#define P25JIT_K1(NAME, FMT, OM, WPS) extern \"C\" __global__ __launch_bounds__(WV, WPS) void NAME(K1Args a, const Taps* __restrict__ t) "
    "{ frontend_body<FMT, true, 5, OM, P25FE_JIT_TX>(a, t); }\n"
    "P25JIT_K1(p25jit_k1_cf32_lin, P25FE_FMT_CF32, OUT_LINEAR, Geo<5>::WAVES_PER_SIMD)\n"
    "P25JIT_K1(p25jit_k1_u8_lin, P25FE_FMT_U8, OUT_LINEAR, Geo<5>::WAVES_PER_SIMD)\n"
    "P25JIT_K1(p25jit_k1_cf32_pl, P25FE_FMT_CF32, OUT_PLANAR, P25FE_K1_PLANAR_WPS)\n"
    "P25JIT_K1(p25jit_k1_u8_pl, P25FE_FMT_U8, OUT_PLANAR, (P25FE_JIT_AVG_N > AVG_DPP_MAX ? 2 : P25FE_K1_PLANAR_WPS_U8))\n"
    "extern \"C\" __global__ __launch_bounds__(WV, 2) void p25jit_chunk_cf32(K1Args a, const Taps* __restrict__ t, ChunkTail c) "
    "{ chunk_body<P25FE_FMT_CF32, true, P25FE_JIT_TX>(a, t, c); }\n"
    "extern \"C\" __global__ __launch_bounds__(WV, 2) void p25jit_chunk_u8(K1Args a, const Taps* __restrict__ t, ChunkTail c) "
    "{ chunk_body<P25FE_FMT_U8, true, P25FE_JIT_TX>(a, t, c); }\n";

// hipRTC is loaded on first use (dlopen), not linked: a deployment without it still loads libp25fe.so -- its handles with
// non-default numbers then run cached / ahead-of-time code objects or the generic kernels.
namespace rtc {
struct Api {
    void* lib = nullptr;
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    decltype(&hiprtcGetErrorString) error_string = nullptr;
    decltype(&hiprtcVersion) version = nullptr;                    // (optional)
    bool ok = false;
};
static const Api& api()
{
    static const Api a = [] {
        Api x;
        // (a process that has torch loaded already holds torch's own libhiprtc under the same soname: dlopen hands that one back)
        // $P25FE_HIPRTC names the library explicitly (and is then the only one tried)
        const char* env = getenv("P25FE_HIPRTC");
        const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
        if (env && *env) x.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        else
            for (const char* n : names)
                if ((x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (!x.lib) return x;
#define P25_SYM(field, name) x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.lib, name))
        P25_SYM(create, "hiprtcCreateProgram"); P25_SYM(compile, "hiprtcCompileProgram");
        P25_SYM(log_size, "hiprtcGetProgramLogSize"); P25_SYM(log, "hiprtcGetProgramLog"); P25_SYM(code_size, "hiprtcGetCodeSize");
        P25_SYM(code, "hiprtcGetCode"); P25_SYM(destroy, "hiprtcDestroyProgram"); P25_SYM(error_string, "hiprtcGetErrorString");
        P25_SYM(version, "hiprtcVersion");
#undef P25_SYM
        x.ok = x.create && x.compile && x.log_size && x.log && x.code_size && x.code && x.destroy && x.error_string;
        return x;
    }();
    return a;
}
}  // namespace rtc

static uint64_t fnv(uint64_t h, const void* p, size_t n)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

uint64_t spec_hash(const Spec& s)
{
    uint64_t h = 0xcbf29ce484222325ull;
    // everything the code object depends on: the source, the options, the numbers
    h = fnv(h, "p25fe-jit-2", 11);
    h = fnv(h, P25FE_SRC_KERNELS, sizeof P25FE_SRC_KERNELS);
    h = fnv(h, P25FE_SRC_RECV, sizeof P25FE_SRC_RECV);
    h = fnv(h, P25FE_SRC_P25FE_H, sizeof P25FE_SRC_P25FE_H);
    h = fnv(h, P25FE_SRC_SPEC_H, sizeof P25FE_SRC_SPEC_H);
    h = fnv(h, WRAPPER, sizeof WRAPPER);
    for (int i = 0; i < N_OPTIONS; ++i) h = fnv(h, OPTIONS[i], strlen(OPTIONS[i]) + 1);
    // (NOT the compiler's version: a process without hipRTC must find the objects an ahead-of-time run left.  The version is
    // the SECOND key, versioned_file_name(): what p25fe_create stores and looks for first)
    h = fnv(h, &s.tx, sizeof s.tx);
    h = fnv(h, &s.t1, sizeof s.t1);
    h = fnv(h, &s.t2, sizeof s.t2);
    h = fnv(h, s.dec, sizeof(float) * (size_t)s.t1);
    h = fnv(h, s.ch, sizeof(float) * (size_t)s.t2);
    h = fnv(h, &s.fm_gain, sizeof s.fm_gain);
    h = fnv(h, &s.u8_lut, sizeof s.u8_lut);
    if (s.u8_lut) {
        // (the table's VALUES are read from the handle's device copy: the code is the same for every non-affine table)
    } else {
        h = fnv(h, &s.u8_scale, sizeof s.u8_scale);
        h = fnv(h, &s.u8_offset, sizeof s.u8_offset);
    }
    h = fnv(h, &s.n_avg, sizeof s.n_avg);
    h = fnv(h, &s.avg_uniform, sizeof s.avg_uniform);
    h = fnv(h, s.avg, sizeof(float) * (size_t)s.n_avg);
    return h;
}

std::string file_name(uint64_t hash)
{
    char b[64];
    snprintf(b, sizeof b, "p25fe-%016llx.hsaco", (unsigned long long)hash);
    return b;
}

// The second key: the same hash + the version of the hipRTC that compiled the object.  p25fe_create stores what it compiles
// under THIS name and looks for it first, so a code object another toolchain left in the cache is never picked up by a
// process that can compile its own; the plain name stays the key of ahead-of-time objects (p25fe_specialize), which a
// process WITHOUT hipRTC must be able to find.  Empty when hipRTC is not available.
std::string versioned_file_name(uint64_t hash)
{
    const rtc::Api& R = rtc::api();
    int major = 0, minor = 0;
    if (!R.ok || !R.version || R.version(&major, &minor) != HIPRTC_SUCCESS) return std::string();
    char b[96];
    snprintf(b, sizeof b, "p25fe-%016llx-rtc%d_%d.hsaco", (unsigned long long)hash, major, minor);
    return b;
}

std::string default_cache_dir()
{
    const char* e = getenv("P25FE_CACHE_DIR");
    if (e && *e) return e;
    e = getenv("XDG_CACHE_HOME");
    if (e && *e) return std::string(e) + "/p25fe";
    e = getenv("HOME");
    if (e && *e && access(e, W_OK) == 0) return std::string(e) + "/.cache/p25fe";
    char b[64];
    snprintf(b, sizeof b, "/tmp/p25fe-cache-%u", (unsigned)geteuid());      // (used only if it passes dir_trusted: ours, private)
    return b;
}

// Code objects are LOADED INTO THE GPU PROCESS from these directories: a directory (or a file in it) that somebody else
// can write is not a cache, it is an attack surface -- the file name is a public hash, anybody can compute it.  Trusted =
// a real directory (no symlink) owned by the caller or by root, writable by its owner only.  /tmp/p25fe-cache-<uid>
// pre-created by another user fails this; so does any world- or group-writable deployment directory.
static bool owner_ok(const struct stat& st) { return (st.st_uid == geteuid() || st.st_uid == 0) && (st.st_mode & 022) == 0; }
bool dir_trusted(const std::string& dir, std::string* why)
{
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) { if (why) *why = dir + ": does not exist"; return false; }
    if (!S_ISDIR(st.st_mode)) { if (why) *why = dir + ": not a directory (symbolic links are not followed)"; return false; }
    if (!owner_ok(st)) { if (why) *why = dir + ": owned by another user or writable by group / others -- not used for code objects"; return false; }
    return true;
}

static bool mkdir_p(const std::string& dir)
{
    std::string cur;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') {
            if (!cur.empty() && mkdir(cur.c_str(), 0700) != 0 && errno != EEXIST) return false;   // (private: code objects get loaded from here)
        }
        if (i < dir.size()) cur.push_back(dir[i]);
    }
    return dir_trusted(dir, nullptr);                              // (EEXIST is not enough: it must be OURS and private)
}

// A cached file is data from outside: before it is handed to the loader it must at least be a complete ELF64 image (the
// section header table, which sits at the end of a code object, lies inside the file) -- a truncated copy is "not cached".
static bool elf_complete(const char* d, size_t n)
{
    if (n < 64 || memcmp(d, "\x7f" "ELF", 4) != 0 || d[4] != 2 /* ELFCLASS64 */) return false;
    uint64_t shoff = 0;
    uint16_t shentsize = 0, shnum = 0;
    memcpy(&shoff, d + 0x28, 8);
    memcpy(&shentsize, d + 0x3a, 2);
    memcpy(&shnum, d + 0x3c, 2);
    return shoff >= 64 && shnum > 0 && shoff <= n && (uint64_t)shentsize * shnum <= n - shoff;
}

// File = the code object + a 32-byte trailer that ties the content to its name: the spec hash the object was built for and
// a hash of the image.  A well-formed code object for OTHER numbers under this name (a copy, a rename), a flipped bit, or
// a file some other writer produced fails here and is "not cached"; nothing unverified reaches hipModuleLoadData.
struct Trailer { char magic[8]; uint64_t spec, content, length; };
static const char TRAILER_MAGIC[8] = {'P', '2', '5', 'F', 'E', 'H', 'S', '1'};
static uint64_t content_hash(const char* d, size_t n) { return fnv(fnv(0xcbf29ce484222325ull, "p25fe-obj", 9), d, n); }

static bool read_file(const std::string& path, uint64_t spec, std::vector<char>& out, std::string* why)
{
    struct stat st;
    if (lstat(path.c_str(), &st) != 0) return false;                // (absent: the normal miss, no words)
    if (!S_ISREG(st.st_mode) || !owner_ok(st)) { if (why) *why += path + ": not a private regular file -- ignored\n"; return false; }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long n = ftell(f);
        if (n > 64 + (long)sizeof(Trailer) && n < (64L << 20) && fseek(f, 0, SEEK_SET) == 0) {
            out.resize((size_t)n);
            if (fread(out.data(), 1, (size_t)n, f) == (size_t)n) {
                Trailer t;
                const size_t len = (size_t)n - sizeof t;
                memcpy(&t, out.data() + len, sizeof t);
                ok = memcmp(t.magic, TRAILER_MAGIC, 8) == 0 && t.length == len && t.spec == spec && elf_complete(out.data(), len) &&
                     t.content == content_hash(out.data(), len);
                if (ok) out.resize(len);
            }
        }
    }
    fclose(f);
    if (!ok) { out.clear(); if (why) *why += path + ": truncated, damaged or built for other numbers -- ignored\n"; }
    return ok;
}

// write to a temporary name, then rename: concurrent creators (one process per GPU, all with the same numbers) never
// see a half-written file
static bool write_file_atomic(const std::string& dir, const std::string& name, uint64_t spec, const std::vector<char>& data)
{
    if (!mkdir_p(dir)) return false;
    char tmp[64];
    snprintf(tmp, sizeof tmp, "/.tmp-%ld-%p", (long)getpid(), (const void*)&data);
    const std::string t = dir + tmp, final_ = dir + "/" + name;
    Trailer tr;
    memcpy(tr.magic, TRAILER_MAGIC, 8);
    tr.spec = spec; tr.content = content_hash(data.data(), data.size()); tr.length = data.size();
    const int fd = open(t.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    if (fd < 0) return false;
    FILE* f = fdopen(fd, "wb");
    if (!f) { close(fd); unlink(t.c_str()); return false; }
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size() && fwrite(&tr, 1, sizeof tr, f) == sizeof tr;
    if (fclose(f) != 0 || !ok || rename(t.c_str(), final_.c_str()) != 0) { unlink(t.c_str()); return false; }
    return true;
}

static void put_float(std::string& o, float v)
{
    char b[48];
    if (v != v || v - v != 0.0f) snprintf(b, sizeof b, "__builtin_nanf(\"\")");     // (callers reject non-finite numbers; never emitted)
    else snprintf(b, sizeof b, "%af", (double)v);
    o += b;
}

static std::string gen_header(const Spec& s)
{
    std::string o;
    o.reserve(8192);
    // hipRTC brings no system headers: the fixed-width types p25fe.h and the kernels use
    o += "typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
         "typedef int int32_t; typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t;\n"
         "typedef unsigned long size_t; typedef unsigned long uintptr_t;\n"
         "#define P25FE_JIT 1\n";
    char b[256];
    snprintf(b, sizeof b, "#define P25FE_JIT_TX %d\n", s.tx);
    o += b;
    snprintf(b, sizeof b, "static constexpr float P25FE_JIT_DECIM_TAPS[%d] = {", s.t1);
    o += b;
    for (int k = 0; k < s.t1; ++k) { if (k) o += ", "; put_float(o, s.dec[k]); }
    snprintf(b, sizeof b, "};\nstatic constexpr float P25FE_JIT_CHAN_TAPS[%d] = {", s.t2);
    o += b;
    for (int k = 0; k < s.t2; ++k) { if (k) o += ", "; put_float(o, s.ch[k]); }
    o += "};\n#define P25FE_JIT_FM_GAIN ";
    put_float(o, s.fm_gain);
    snprintf(b, sizeof b, "\n#define P25FE_JIT_U8_LUT %d\n#define P25FE_JIT_U8_SCALE ", s.u8_lut ? 1 : 0);
    o += b;
    put_float(o, s.u8_lut ? 0.0f : s.u8_scale);
    o += "\n#define P25FE_JIT_U8_OFFSET ";
    put_float(o, s.u8_lut ? 0.0f : s.u8_offset);
    snprintf(b, sizeof b, "\n#define P25FE_JIT_AVG_N %d\n#define P25FE_JIT_AVG_UNIFORM %d\nstatic constexpr float P25FE_JIT_AVG_TAPS[%d] = {", s.n_avg,
             s.avg_uniform ? 1 : 0, s.n_avg);
    o += b;
    for (int k = 0; k < s.n_avg; ++k) { if (k) o += ", "; put_float(o, s.avg[k]); }
    o += "};\n";
    return o;
}

static bool compile(const Spec& s, std::vector<char>& code, std::string& log)
{
    const std::string spec_h = gen_header(s);
    const char* headers[] = {spec_h.c_str(), P25FE_SRC_KERNELS, P25FE_SRC_RECV, P25FE_SRC_P25FE_H, P25FE_SRC_SPEC_H};
    const char* names[] = {"p25fe_jit_spec.h", "p25fe_kernels.hip", "p25fe_recv.hip", "p25fe.h", "p25fe_spec.h"};
    const rtc::Api& R = rtc::api();
    if (!R.ok) { log += "libhiprtc is not available in this process: cannot compile (ahead-of-time code objects are still looked up)\n"; return false; }
    hiprtcProgram prog = nullptr;
    hiprtcResult r = R.create(&prog, WRAPPER, "p25fe_jit.hip", 5, headers, names);
    if (r != HIPRTC_SUCCESS) { log += std::string("hiprtcCreateProgram: ") + R.error_string(r) + "\n"; return false; }
    r = R.compile(prog, N_OPTIONS, const_cast<const char**>(OPTIONS));
    size_t ls = 0;
    if (R.log_size(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        std::string l(ls, '\0');
        if (R.log(prog, &l[0]) == HIPRTC_SUCCESS) log += l.c_str();
    }
    bool ok = false;
    if (r == HIPRTC_SUCCESS) {
        size_t cs = 0;
        if (R.code_size(prog, &cs) == HIPRTC_SUCCESS && cs > 0) {
            code.resize(cs);
            ok = R.code(prog, code.data()) == HIPRTC_SUCCESS;
        }
    } else {
        log += std::string("hiprtcCompileProgram: ") + R.error_string(r) + "\n";
    }
    R.destroy(&prog);
    if (!ok) code.clear();
    return ok;
}

bool get_code(const Spec& s, const std::vector<std::string>& dirs, bool do_compile, const std::string& store_dir, bool aot,
              std::vector<char>& code, std::string& path, std::string& log, bool* from_file)
{
    if (from_file) *from_file = false;
    const uint64_t hash = spec_hash(s);
    const std::string plain = file_name(hash), versioned = versioned_file_name(hash);
    for (const std::string& d : dirs) {
        if (d.empty()) continue;
        std::string why;
        if (!dir_trusted(d, &why)) {
            struct stat st;
            if (lstat(d.c_str(), &st) == 0) log += why + "\n";      // (a directory that does not exist yet is the normal cold start)
            continue;
        }
        for (const std::string* name : {&versioned, &plain}) {      // this toolchain's own object first, then an ahead-of-time one
            if (name->empty()) continue;
            const std::string p = d + "/" + *name;
            if (read_file(p, hash, code, &log)) { path = p; if (from_file) *from_file = true; return true; }
        }
    }
    if (!do_compile) { log += "no cached code object " + plain + " and compilation is switched off\n"; return false; }
    if (!compile(s, code, log)) return false;
    path.clear();
    if (!store_dir.empty()) {
        // ahead of time (p25fe_specialize): the plain name, which a process without hipRTC can find; at p25fe_create: this
        // toolchain's versioned name
        const std::string& name = (aot || versioned.empty()) ? plain : versioned;
        if (write_file_atomic(store_dir, name, hash, code)) path = store_dir + "/" + name;
        else log += "could not store " + store_dir + "/" + name + " (not writable, or not a private directory of this user: the code object is used from memory)\n";
    }
    return true;
}

}  // namespace p25jit
