// p25fe_replay -- file-in / file-out driver in the role of the reference's command line for this path
// (src/main.rs:95-102, 162-175, 278-283 and src/replay.rs:26-57): a deterministic harness for the hot path.
//
//   p25fe_replay [-w BB.f32le] [-j EVENTS.jsonl] [-b CHUNKS] u8|cf32|bb <in> <dibits.out>
//   p25fe_replay -W WINDOW_BYTES u8|cf32 <in> <dibits.out>
//
//     u8    RTL-SDR style interleaved u8 I/Q, the reference's live input (src/consts.rs:6: 32768-byte chunks)
//     cf32  Complex32 I/Q at 240 ksps
//     bb    48 kHz f32le baseband: what `p25rx -w FILE` records and `p25rx -r FILE` replays (src/main.rs:95-102)
//
//     -w FILE   write the baseband as f32le / 48 kHz / mono while receiving -- the reference's `-w`
//               (src/main.rs:99-102, 174-175): RecvTask::run hands every chunk to a callback AFTER feeding it
//               (src/recv.rs:152), and the callback appends it to the file (src/main.rs:278-283).
//               `p25fe_replay bb FILE ...` on that file reproduces the dibits of the recording run.
//     -j FILE   JSON lines in the vocabulary of the reference's hub: {"event":"sigPower"} per power report
//               (HubEvent::UpdateSignalPower, src/demod.rs:95-101, src/hub.rs:344), {"event":"nid"} per frame sync
//               (MessageEvent::PacketNID, src/recv.rs:216-222), {"event":"updateStats","periodic":true} after every 16th
//               baseband chunk (Throttler::new(16), src/recv.rs:141, 162-165), one final {"event":"updateStats"} with
//               the "bch" row of serialize_stats (src/hub.rs:401-402, 559, 574-581).
//     -b N      file chunks per library call (default 64; 1 = the reference's cadence of one 32768-byte buffer per
//               DemodTask iteration, which also fixes the every-4th-chunk power report of src/demod.rs:67, 95).
//               The dibits do not depend on N (streaming semantics); only the PCIe transfer size does.
//
//     -W BYTES  bulk mode for long captures (p25fe_run_host_windows): a READER THREAD fills pinned blocks of eight windows
//               from the file while the library pipelines the previous block -- window k + 1 on its way to the GPU,
//               window k in the kernels, window k - 1's dibits on their way back.  The shape of the reference's own loop (a
//               reader feeding a pool of buffers to the demodulator, src/sdr.rs:25-33, src/demod.rs:62-70) at bus speed;
//               same dibits as the chunked modes.  BYTES takes k / M suffixes (64M is the library's default).
//
// Wires DemodTask -> RecvTask exactly like src/main.rs:270-287, single-threaded through in-memory channels.
#include <chrono>
#include <cinttypes>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <mutex>
#include <thread>

#include <hip/hip_runtime_api.h>

#include "p25fe_host.hpp"

using namespace p25rx;

template <class T> struct Chan {
    std::deque<T> q;
    void send(T v) { q.push_back(std::move(v)); }
    bool recv(T& v)
    {
        if (q.empty()) return false;
        v = std::move(q.front());
        q.pop_front();
        return true;
    }
};

// hub side of the harness: power reports become JSON lines as they arrive
struct Hub {
    FILE* js = nullptr;
    size_t n_power = 0;
    size_t n_stats = 0;
    void send(HubEvent e)
    {
        ++n_power;
        if (js) std::fprintf(js, "{\"event\":\"sigPower\",\"dbm\":%.4f}\n", (double)e.signal_power_dbm);
    }
    void send(StatsEvent e)                                           // every 16th receiver event, src/recv.rs:141, 162-165
    {
        ++n_stats;
        if (js) std::fprintf(js, "{\"event\":\"updateStats\",\"periodic\":true,\"dibits\":%" PRIu64 ",\"syncs\":%" PRIu64 "}\n", e.dibits, e.syncs);
    }
};

struct Sink {
    std::ofstream out;
    std::vector<uint8_t> all_dibits;            // kept for the NID pass at the end of the file (-j)
    std::vector<int64_t> sync_pos;
    std::vector<uint64_t> sync_dibit;
    bool keep = false;
    size_t n_sync = 0, n_dibits = 0;
    void send(Symbols s)
    {
        out.write(reinterpret_cast<const char*>(s.dibits.data()), (std::streamsize)s.dibits.size());
        n_dibits += s.dibits.size();
        n_sync += s.sync_pos.size();
        if (keep) {
            all_dibits.insert(all_dibits.end(), s.dibits.begin(), s.dibits.end());
            sync_pos.insert(sync_pos.end(), s.sync_pos.begin(), s.sync_pos.end());
            sync_dibit.insert(sync_dibit.end(), s.sync_dibit.begin(), s.sync_dibit.end());
        }
    }
};

static int usage(const char* argv0)
{
    std::fprintf(stderr, "usage: %s [-w BB.f32le] [-j EVENTS.jsonl] [-b CHUNKS] u8|cf32|bb <in> <dibits.out>\n"
                         "       %s -W WINDOW_BYTES u8|cf32 <in> <dibits.out>\n", argv0, argv0);
    return 2;
}

// bulk mode: reader thread -> two pinned blocks -> p25fe_run_host_windows
static int bulk(Handle& h, const std::string& mode, std::ifstream& in, const char* out_path, size_t window_bytes)
{
    const int fmt = mode == "u8" ? P25FE_FMT_U8 : P25FE_FMT_CF32;
    const size_t eb = fmt == P25FE_FMT_U8 ? 2 : 8;
    const size_t window = window_bytes / eb / 8 * 8;
    if (window < 8192) { std::fprintf(stderr, "window too small\n"); return 2; }
    const size_t block = 8 * window;                                  // samples per pinned block: the reader fills one while the library pipelines the other
                                                                      // (measured on a 1.15 GB file from the page cache: 8 windows 78 ms, 2 windows 93 - 105 ms;
                                                                      // the single reader thread's ~15 GB/s is what bounds this mode, not the bus)
    char* buf[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; ++b)
        if (hipHostMalloc(reinterpret_cast<void**>(&buf[b]), block * eb, hipHostMallocDefault) != hipSuccess) {
            std::fprintf(stderr, "unable to allocate pinned blocks\n");
            return 1;
        }
    std::mutex mu;
    std::condition_variable cv;
    size_t filled[2] = {0, 0};
    bool full[2] = {false, false}, eof = false, stop = false;      // stop: the consumer gave up (an error): the reader must not block
    std::thread reader([&] {
        for (int b = 0;; b ^= 1) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !full[b] || stop; });
                if (stop) return;
            }
            in.read(buf[b], (std::streamsize)(block * eb));
            const size_t got = (size_t)in.gcount() / eb;
            std::lock_guard<std::mutex> lk(mu);
            filled[b] = got; full[b] = true; eof = got < block;
            cv.notify_all();
            if (got < block) return;
        }
    });
    std::ofstream out(out_path, std::ios::binary);
    std::vector<uint8_t> dib(block / 30 + 4);
    size_t total = 0, samples = 0, windows = 0;
    double ms_h2d = 0.0, ms_comp = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    int failed = 0;                                                  // an error ends the loop, never the process: the reader thread is joinable
    if (!out) { std::fprintf(stderr, "unable to open %s\n", out_path); failed = 1; }
    for (int b = 0; !failed; b ^= 1) {
        size_t n;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return full[b]; }); n = filled[b]; }
        size_t nd = 0;
        p25fe_windows_stats_t st;
        const int rc = p25fe_run_host_windows(h.get(), buf[b], fmt, n, window, dib.data(), dib.size(), &nd, &st);
        if (rc != P25FE_OK) {
            std::fprintf(stderr, "p25fe_replay: unable to run the capture: %s (%d)\n", p25fe_strerror(rc), rc);
            failed = 1;
            break;
        }
        out.write(reinterpret_cast<const char*>(dib.data()), (std::streamsize)nd);
        if (!out.good()) { std::fprintf(stderr, "p25fe_replay: write error on %s (disk full?)\n", out_path); failed = 1; break; }
        total += nd; samples += n; windows += st.n_windows; ms_h2d += st.ms_h2d; ms_comp += st.ms_compute;
        bool last;
        { std::lock_guard<std::mutex> lk(mu); full[b] = false; last = eof && n < block; cv.notify_all(); }
        if (last || n < block) break;
    }
    { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); }
    reader.join();
    out.close();
    if (!failed && out.fail()) { std::fprintf(stderr, "p25fe_replay: write error on %s at close\n", out_path); failed = 1; }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int b = 0; b < 2; ++b) (void)hipHostFree(buf[b]);
    if (failed) return 1;
    std::fprintf(stderr, "p25fe_replay: %zu dibits from %zu samples in %zu windows, %.1f ms (%.1f Msamples/s, %.2f GB/s; H2D copies %.1f ms, "
                         "kernels %.1f ms)\n", total, samples, windows, ms, samples / ms / 1e3, samples * eb / ms / 1e6, ms_h2d, ms_comp);
    return 0;
}

int main(int argc, char** argv)
{
    const char *wpath = nullptr, *jpath = nullptr;
    size_t batch = 64, window_bytes = 0;
    int a = 1;
    for (; a < argc && argv[a][0] == '-' && argv[a][1] != '\0'; a += 2) {
        if (a + 1 >= argc) return usage(argv[0]);
        if (!std::strcmp(argv[a], "-w")) wpath = argv[a + 1];
        else if (!std::strcmp(argv[a], "-j")) jpath = argv[a + 1];
        else if (!std::strcmp(argv[a], "-b")) batch = (size_t)std::strtoull(argv[a + 1], nullptr, 10);
        else if (!std::strcmp(argv[a], "-W")) {
            char* end = nullptr;
            window_bytes = (size_t)std::strtoull(argv[a + 1], &end, 10);
            if (end && (*end == 'k' || *end == 'K')) window_bytes <<= 10;
            else if (end && (*end == 'm' || *end == 'M')) window_bytes <<= 20;
        }
        else return usage(argv[0]);
    }
    if (argc - a != 3 || batch == 0) return usage(argv[0]);
    const std::string mode = argv[a];
    std::ifstream in(argv[a + 1], std::ios::binary);
    if (!in) { std::fprintf(stderr, "unable to open %s\n", argv[a + 1]); return 1; }
    Handle h(0, 1);
    if (window_bytes) {
        if (wpath || jpath || (mode != "u8" && mode != "cf32")) return usage(argv[0]);
        return bulk(h, mode, in, argv[a + 2], window_bytes);
    }
    Chan<std::vector<uint8_t>> reader;
    Hub hub;
    Chan<Baseband> chan;
    Sink sink;
    sink.out.open(argv[a + 2], std::ios::binary);
    sink.keep = jpath != nullptr;
    if (jpath && !(hub.js = std::fopen(jpath, "w"))) { std::fprintf(stderr, "unable to open %s\n", jpath); return 1; }
    std::ofstream bbfile;                                             // samples_file, src/main.rs:174-175
    if (wpath) {
        bbfile.open(wpath, std::ios::binary);
        if (!bbfile) { std::fprintf(stderr, "unable to open baseband file\n"); return 1; }
    }
    auto dump = [&](const std::vector<float>& s) {                    // the closure of src/main.rs:278-283
        if (wpath) bbfile.write(reinterpret_cast<const char*>(s.data()), (std::streamsize)(s.size() * sizeof(float)));
    };
    RecvTask<Chan<Baseband>, Sink> recv(h, chan, sink);

    if (mode == "u8") {
        std::vector<uint8_t> buf(BUF_BYTES * batch);
        DemodTask<Chan<std::vector<uint8_t>>, Hub, Chan<Baseband>> demod(h, reader, hub, chan);
        while (in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size()) || in.gcount() > 0) {
            std::vector<uint8_t> chunk(buf.begin(), buf.begin() + (in.gcount() & ~std::streamsize(1)));
            reader.send(std::move(chunk));
            demod.run();
            recv.run(dump, hub);
        }
    } else if (mode == "cf32") {
        std::vector<float> buf(2 * BUF_SAMPLES * batch), bb(BUF_SAMPLES * batch / 5 + 2);
        unsigned notifier = 0;
        while (in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(buf.size() * 4)) || in.gcount() > 0) {
            const size_t n = (size_t)in.gcount() / 8;
            size_t n_out = 0;
            float power = 0.f;
            const bool want = (++notifier % 4) == 0;                  // Throttler::new(4), src/demod.rs:67, 95
            expect(p25fe_demod_cf32(h.get(), buf.data(), n, bb.data(), bb.size(), &n_out, want ? &power : nullptr),
                   "unable to demodulate");
            if (want) hub.send(HubEvent{power});
            chan.send(Baseband{std::vector<float>(bb.begin(), bb.begin() + (long)n_out)});
            recv.run(dump, hub);
        }
    } else if (mode == "bb") {
        std::vector<float> buf(8192 * batch);                         // replay.rs reads 32768-byte blocks (src/replay.rs:27)
        while (in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(buf.size() * 4)) || in.gcount() > 0) {
            const size_t n = (size_t)in.gcount() / 4;                 // only the bytes actually read (replay.rs:36 re-feeds stale tail)
            chan.send(Baseband{std::vector<float>(buf.begin(), buf.begin() + (long)n)});
            recv.run(dump, hub);
        }
    } else {
        return usage(argv[0]);
    }

    if (hub.js) {
        // network identifiers of all frame syncs in one pass over the finished dibit stream, then the stats row
        std::vector<p25fe_nid_t> nid(sink.sync_pos.size());
        expect(p25fe_nid(h.get(), sink.all_dibits.data(), sink.all_dibits.size(), sink.sync_dibit.data(), sink.sync_pos.data(),
                         nid.size(), nid.data()),
               "unable to decode network identifiers");
        uint64_t words = 0, errs = 0, fixed = 0;
        for (const p25fe_nid_t& r : nid) {
            std::fprintf(hub.js, "{\"event\":\"nid\",\"sync_pos\":%" PRId64 ",\"nac\":%u,\"duid\":%u,\"errors\":%u,\"valid\":%d}\n",
                         r.sync_pos, (unsigned)r.nac, (unsigned)r.duid, (unsigned)r.n_errors, (int)r.valid);
            if (r.valid >= 0) { ++words; if (r.valid == 0) ++errs; else fixed += r.n_errors; }
        }
        // serialize_code_stats (src/hub.rs:574-581): totalWords, errWords, totalSymbols = words * size, fixedSymbols
        std::fprintf(hub.js,
                     "{\"event\":\"updateStats\",\"dibits\":%zu,\"syncs\":%zu,\"bch\":{\"totalWords\":%" PRIu64 ",\"errWords\":%" PRIu64
                     ",\"totalSymbols\":%" PRIu64 ",\"fixedSymbols\":%" PRIu64 "}}\n",
                     sink.n_dibits, sink.n_sync, words, errs, words * 63, fixed);
        std::fclose(hub.js);
    }
    std::fprintf(stderr, "p25fe_replay: %zu dibits, %zu frame syncs, %zu power reports\n", sink.n_dibits, sink.n_sync,
                 hub.n_power);
    return 0;
}
