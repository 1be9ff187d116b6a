// AddressSanitizer / UBSan driver for the HOST side of libp25fe.so (built with -fsanitize=address,undefined on the host
// pass only: `make -C p25rx_amd/csrc asan`).  Without a GPU it walks every entry point's argument checks and the pure host
// logic (shard resolve, sizes, error strings); with one it also streams chunks through the host-buffer entry points, whose
// staging / state code is host code.  Exit code 0 and no sanitizer report = pass (tests/test_sanitizers.py).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <unistd.h>

#include "p25fe.h"

#define EXPECT(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main()
{
    p25fe_config_t cfg;
    p25fe_default_config(&cfg);
    p25fe_default_config(nullptr);
    EXPECT(cfg.abi_version == P25FE_ABI_VERSION && cfg.n_channels == 1 && cfg.symbol_clock == P25FE_CLOCK_FIXED);
    for (int s = 1; s >= -8; --s) EXPECT(p25fe_strerror(s) != nullptr && std::strlen(p25fe_strerror(s)) > 0);
    EXPECT(p25fe_n_baseband(0, 16384) == 3276 && p25fe_n_baseband(16384, 16384) == 3277 && p25fe_n_baseband(3, 2) == 1);
    EXPECT(p25fe_n_predecim(0, 9) == 0 && p25fe_n_predecim(0, 10) == 1 && p25fe_shard_halo() % 8 == 0);
    // argument checks (no handle needed)
    p25fe_t* h = nullptr;
    EXPECT(p25fe_create(nullptr, &h) == P25FE_ERR_ARG && p25fe_create(&cfg, nullptr) == P25FE_ERR_ARG);
    p25fe_config_t bad = cfg;
    bad.abi_version = 1;
    EXPECT(p25fe_create(&bad, &h) == P25FE_ERR_ARG);
    bad = cfg; bad.n_decim_taps = 65;
    EXPECT(p25fe_create(&bad, &h) == P25FE_ERR_ARG);
    bad = cfg; bad.symbol_clock = 7;
    EXPECT(p25fe_create(&bad, &h) == P25FE_ERR_ARG);
    // ABI 4: the numbers of the reference's constructors are validated before any device is looked for
    bad = cfg; bad.decim_taps[3] = NAN;
    EXPECT(p25fe_create(&bad, &h) == P25FE_ERR_ARG && p25fe_specialize(&bad, nullptr, nullptr, 0) == P25FE_ERR_ARG);
    bad = cfg; bad.fm_deviation_hz = 0;
    EXPECT(p25fe_create(&bad, &h) == P25FE_ERR_ARG);
    bad = cfg; bad.fm_gain = INFINITY;
    EXPECT(p25fe_create(&bad, &h) == P25FE_ERR_ARG);
    bad = cfg; bad.specialize = 9;
    EXPECT(p25fe_create(&bad, &h) == P25FE_ERR_ARG);
    bad = cfg; bad.u8_lut_valid = 1; bad.u8_lut[200] = -INFINITY;
    EXPECT(p25fe_create(&bad, &h) == P25FE_ERR_ARG);
    EXPECT(p25fe_kernel_variant(nullptr) < 0 && p25fe_run_host_windows(nullptr, nullptr, 0, 0, 0, nullptr, 0, nullptr, nullptr) == P25FE_ERR_ARG);
    {
        // kernel specialisation needs no GPU: header generation, hashing, the cache's file handling and hipRTC itself
        char dir[] = "/tmp/p25fe_asan_XXXXXX", path[512], log[256];
        EXPECT(mkdtemp(dir) != nullptr);
        EXPECT(p25fe_specialize(&cfg, dir, path, sizeof path) == P25FE_OK && path[0] == '\0');      // the build's own numbers: nothing to do
        p25fe_config_t cust = cfg;
        cust.n_decim_taps = 40; cust.n_chan_taps = 17;
        for (int k = 0; k < 40; ++k) cust.decim_taps[k] = 0.02f * (float)std::sin(0.3 * k) + (k == 5 ? -0.0f : 0.f);
        for (int k = 0; k < 17; ++k) cust.chan_taps[k] = 1.0f / 17.0f;
        cust.fm_deviation_hz = 4000;
        cust.u8_lut_valid = 1;
        for (int b = 0; b < 256; ++b) cust.u8_lut[b] = std::tanh((b - 127.5f) / 90.0f);
        EXPECT(p25fe_specialize(&cust, dir, path, 8) == P25FE_OK);                                     // a short path buffer is truncated, not overrun
        EXPECT(p25fe_specialize(&cust, dir, path, sizeof path) == P25FE_OK && std::strstr(path, ".hsaco") != nullptr);
        FILE* f = std::fopen(path, "rb");
        EXPECT(f != nullptr);
        std::fclose(f);
        EXPECT(p25fe_specialize(&cust, dir, path, sizeof path) == P25FE_OK);                           // found, not rebuilt
        EXPECT(p25fe_specialize_log(log, sizeof log) < sizeof log && p25fe_specialize_log(nullptr, 0) == 0);
        EXPECT(p25fe_specialize(&cust, "/proc/this/cannot/be/created", path, sizeof path) == P25FE_ERR_JIT);
        std::remove(path);
        rmdir(dir);
    }
    EXPECT(p25fe_reset(nullptr) == P25FE_ERR_ARG && p25fe_resync(nullptr) == P25FE_ERR_ARG && p25fe_last_hip_error(nullptr) == 0);
    size_t nn = 0;
    EXPECT(p25fe_state_size(nullptr, &nn) == P25FE_ERR_ARG && p25fe_run_u8(nullptr, nullptr, 3, nullptr, 0, nullptr) == P25FE_ERR_ARG);
    EXPECT(p25fe_resync_at_dev(nullptr, nullptr, 0, 0) == P25FE_ERR_ARG && p25fe_join_dev(nullptr, nullptr) == P25FE_ERR_ARG);
    p25fe_destroy(nullptr);
    // shard resolve: pure host logic, both clocks
    {
        std::vector<p25fe_result_t> summ(4);
        std::memset(summ.data(), 0, sizeof(p25fe_result_t) * 4);
        uint64_t bb0[4] = {0, 1000, 2000, 3000}, bbn[4] = {1000, 1000, 1000, 1000}, off[5];
        p25fe_anchor_t anc[4];
        for (auto& r : summ) { r.first_event = -1; r.carry_end = -1; r.first_seg_end = -1; }
        summ[0].first_event = 307; summ[0].carry_end = 308; summ[0].n_dibits_after_first = 69;
        summ[0].anchor_out = p25fe_anchor_t{302, 0.2f, 0.f, -0.2f, 1, 10, 1};
        summ[2].first_event = 2508; summ[2].carry_end = 2509; summ[2].n_dibits_after_first = 49; summ[2].first_seg_end = 2998;
        summ[2].flags = P25FE_RES_FIRST_TRACKS_CARRY | P25FE_RES_OUT_PERIOD_FROM_CARRY;
        summ[2].anchor_out = p25fe_anchor_t{2503, 0.3f, 0.1f, -0.1f, 1, 0, 0};
        EXPECT(p25fe_shard_resolve(summ.data(), bb0, bbn, 4, 0, anc, off) == P25FE_OK && off[4] > off[3] && anc[3].s == 2503);
        EXPECT(p25fe_shard_resolve(summ.data(), bb0, bbn, 4, 1, anc, off) == P25FE_OK && anc[3].period_d == 4 * 2201 && anc[3].period_n == 4 * 220);
        EXPECT(p25fe_shard_resolve(nullptr, bb0, bbn, 4, 0, anc, off) == P25FE_ERR_ARG);
        EXPECT(p25fe_shard_resolve(summ.data(), bb0, bbn, 0, 0, anc, off) == P25FE_OK && off[0] == 0);
    }
    const int rc = p25fe_create(&cfg, &h);
    if (rc == P25FE_ERR_NO_DEVICE) {
        std::printf("abi host driver ok (no device: argument checks and host logic only)\n");
        return 0;
    }
    EXPECT(rc == P25FE_OK && h != nullptr);
    // with a GPU: the streaming entry points (their staging, history and state bookkeeping are host code)
    std::vector<float> iq(2 * 50000);
    double ph = 0;
    for (size_t i = 0; i < iq.size() / 2; ++i) { ph += 0.05 * std::sin(i * 0.002); iq[2 * i] = 0.5f * (float)std::cos(ph); iq[2 * i + 1] = 0.5f * (float)std::sin(ph); }
    std::vector<uint8_t> u8(iq.size()), dib(4000);
    for (size_t i = 0; i < iq.size(); ++i) u8[i] = (uint8_t)(127.5f * (iq[i] + 1.f));
    std::vector<float> bb(12000);
    std::vector<int64_t> sp(64);
    std::vector<uint64_t> sd(64);
    size_t n_out = 0, nd = 0, ns = 0;
    float pw = 0.f;
    const size_t cuts[] = {0, 1, 5, 7, 16384, 16391, 40000, 50000};
    for (size_t k = 0; k + 1 < sizeof cuts / sizeof cuts[0]; ++k) {
        const size_t a = cuts[k], n = cuts[k + 1] - cuts[k];
        EXPECT(p25fe_demod_cf32(h, iq.data() + 2 * a, n, bb.data(), bb.size(), &n_out, (k & 1) ? &pw : nullptr) == P25FE_OK);
        EXPECT(p25fe_slice(h, bb.data(), n_out, dib.data(), dib.size(), &nd, sp.data(), sd.data(), sp.size(), &ns) == P25FE_OK);
    }
    EXPECT(p25fe_demod_cf32(h, iq.data(), 40000, bb.data(), 10, &n_out, nullptr) == P25FE_ERR_CAPACITY);
    EXPECT(p25fe_reset(h) == P25FE_OK);
    for (size_t a = 0; a + 32768 <= u8.size(); a += 32768) EXPECT(p25fe_run_u8(h, u8.data() + a, 32768, dib.data(), dib.size(), &nd) == P25FE_OK);
    EXPECT(p25fe_run_cf32(h, iq.data(), 100, dib.data(), dib.size(), &nd) == P25FE_ERR_FORMAT);     // u8 / cf32 mixed within one stream
    EXPECT(p25fe_state_size(h, &nn) == P25FE_OK);
    std::vector<char> blob(nn);
    EXPECT(p25fe_state_export(h, blob.data(), nn - 1, &nn) == P25FE_ERR_CAPACITY && p25fe_state_export(h, blob.data(), nn, &nn) == P25FE_OK);
    EXPECT(p25fe_state_import(h, blob.data(), nn) == P25FE_OK && p25fe_state_import(h, blob.data(), nn - 1) == P25FE_ERR_ARG);
    EXPECT(p25fe_resync(h) == P25FE_OK);
    p25fe_destroy(h);
    {
        // a long host capture as windows (pageable input: staged by this thread), with other tables: kernels specialised at create
        p25fe_config_t cust = cfg;
        cust.n_decim_taps = 20;
        cust.specialize = P25FE_SPECIALIZE_REQUIRE;
        p25fe_t* hw = nullptr;
        EXPECT(p25fe_create(&cust, &hw) == P25FE_OK && p25fe_kernel_variant(hw) == P25FE_VARIANT_SPECIALIZED);
        std::vector<float> big(2 * 3000000);
        for (size_t i = 0; i < big.size(); ++i) big[i] = iq[i % iq.size()];
        std::vector<uint8_t> dw(big.size() / 2 / 30 + 4);
        p25fe_windows_stats_t st;
        size_t ndw = 0;
        EXPECT(p25fe_run_host_windows(hw, big.data(), P25FE_FMT_CF32, big.size() / 2, 1 << 20, dw.data(), dw.size(), &ndw, &st) == P25FE_OK);
        EXPECT(st.n_windows == 3 && st.pinned_input == 0);
        EXPECT(p25fe_run_host_windows(hw, big.data(), P25FE_FMT_CF32, big.size() / 2, 1 << 20, dw.data(), 10, &ndw, &st) == P25FE_ERR_CAPACITY);
        EXPECT(p25fe_run_host_windows(hw, big.data(), P25FE_FMT_U8, 1000, 0, dw.data(), dw.size(), &ndw, nullptr) == P25FE_ERR_FORMAT);
        p25fe_destroy(hw);
    }
    std::printf("abi host driver ok (device present: streaming entry points exercised)\n");
    return 0;
}
