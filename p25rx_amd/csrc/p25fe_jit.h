// p25fe_jit.h -- internal: kernels specialised for a caller's numbers (p25fe_jit.cpp), shared with p25fe_api.hip.
//
// The reference fixes its FIR tables at compile time (type-level filters, src/demod.rs:27-29) and its discriminator /
// LUT constants at construction (src/demod.rs:54, 83).  The library's built-in immediate-coefficient kernels carry the
// build's own numbers; for any other set the same kernel source -- embedded in the library -- is compiled by hipRTC with
// the caller's numbers as immediates, and the code object is cached on disk under a hash of the numbers.
#ifndef P25FE_JIT_H
#define P25FE_JIT_H

#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace p25jit {

struct Spec {
    int tx;                 // 0: evaluation lengths 31 / 41, 1: 64 / 64 (docs/SPEC.md 3.3)
    int t1, t2;             // evaluation lengths (tables zero-padded to them)
    float dec[64], ch[64];
    float fm_gain;
    int u8_lut;             // 1: the u8 table is not affine -> LDS lookup; the VALUES come from the handle's device table, so
                            // one code object serves every such table
    float u8_scale, u8_offset;      // the affine table: byte b -> fma(b, scale, offset) (ignored when u8_lut)
    int n_avg, avg_uniform;         // post-discriminator filter (docs/SPEC.md 3.5): taps, all-equal flag
    float avg[64];
};

// names of the extern "C" kernels of a specialised module: [fmt 0 = cf32, 1 = u8][0 = linear, 1 = planar, 2 = chunk]
extern const char* const KERNEL_NAMES[2][3];

uint64_t spec_hash(const Spec& s);
std::string default_cache_dir();                       // $P25FE_CACHE_DIR, $XDG_CACHE_HOME/p25fe, $HOME/.cache/p25fe, /tmp/p25fe-cache-<uid>
std::string file_name(uint64_t hash);                  // "p25fe-<16 hex>.hsaco": the key of ahead-of-time objects
std::string versioned_file_name(uint64_t hash);        // "p25fe-<16 hex>-rtc<major>_<minor>.hsaco": what p25fe_create stores; "" without hipRTC
// a real directory owned by the caller (or root), writable by its owner only -- the only kind code objects are read from or
// written to; *why (nullable) says what is wrong
bool dir_trusted(const std::string& dir, std::string* why);

// Code object for `s`: looked up in `dirs` (first hit wins; per directory this toolchain's versioned name first, then the
// plain one; directories and files that are not private to the caller are skipped, files are verified against the trailer
// that ties their content to `s`); if absent and `compile`, built with hipRTC and stored in `store_dir` (empty: not
// stored) under the plain name (`aot`, p25fe_specialize) or the versioned one.  Returns true on success; `log` receives
// the compiler's words / what was tried and refused; *from_file (nullable): the object was read from `path`, not compiled.
bool get_code(const Spec& s, const std::vector<std::string>& dirs, bool compile, const std::string& store_dir, bool aot,
              std::vector<char>& code, std::string& path, std::string& log, bool* from_file = nullptr);

}  // namespace p25jit
#endif
