// p25fe_replay -- file-in / file-out driver in the role of the reference's replay.rs
// (src/replay.rs:26-57, src/main.rs:95-98, 162-169): deterministic harness for the hot path.
//
//   p25fe_replay u8   <iq.u8>    <dibits.out>   RTL-SDR style interleaved u8 I/Q, 32768-byte chunks (src/consts.rs:6)
//   p25fe_replay cf32 <iq.cf32>  <dibits.out>   Complex32 I/Q, 16384-sample chunks
//   p25fe_replay bb   <bb.f32le> <dibits.out>   48 kHz f32le baseband, the reference's -w / -r format (src/main.rs:101)
//
// Wires DemodTask -> RecvTask exactly like src/main.rs:270-287, single-threaded through in-memory channels.
#include <cstring>
#include <deque>
#include <fstream>

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

struct Sink {
    std::ofstream out;
    size_t n_sync = 0, n_dibits = 0;
    void send(Symbols s)
    {
        out.write(reinterpret_cast<const char*>(s.dibits.data()), (std::streamsize)s.dibits.size());
        n_dibits += s.dibits.size();
        n_sync += s.sync_pos.size();
    }
};

int main(int argc, char** argv)
{
    if (argc != 4) {
        std::fprintf(stderr, "usage: %s u8|cf32|bb <in> <dibits.out>\n", argv[0]);
        return 2;
    }
    const std::string mode = argv[1];
    std::ifstream in(argv[2], std::ios::binary);
    if (!in) { std::fprintf(stderr, "unable to open %s\n", argv[2]); return 1; }
    Handle h(0, 1);
    Chan<std::vector<uint8_t>> reader;
    Chan<HubEvent> hub;
    Chan<Baseband> chan;
    Sink sink;
    sink.out.open(argv[3], std::ios::binary);
    RecvTask<Chan<Baseband>, Sink> recv(h, chan, sink);

    if (mode == "u8") {
        std::vector<uint8_t> buf(BUF_BYTES);
        DemodTask<Chan<std::vector<uint8_t>>, Chan<HubEvent>, Chan<Baseband>> demod(h, reader, hub, chan);
        while (in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size()) || in.gcount() > 0) {
            std::vector<uint8_t> chunk(buf.begin(), buf.begin() + (in.gcount() & ~std::streamsize(1)));
            reader.send(std::move(chunk));
            demod.run();
            recv.run([](const std::vector<float>&) {});
        }
    } else if (mode == "cf32") {
        std::vector<float> buf(2 * BUF_SAMPLES), bb(BUF_SAMPLES / 5 + 2);
        while (in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(buf.size() * 4)) || in.gcount() > 0) {
            const size_t n = (size_t)in.gcount() / 8;
            size_t n_out = 0;
            expect(p25fe_demod_cf32(h.get(), buf.data(), n, bb.data(), bb.size(), &n_out, nullptr), "unable to demodulate");
            chan.send(Baseband{std::vector<float>(bb.begin(), bb.begin() + (long)n_out)});
            recv.run([](const std::vector<float>&) {});
        }
    } else if (mode == "bb") {
        std::vector<float> buf(8192);                                 // replay.rs reads 32768-byte blocks (src/replay.rs:27)
        while (in.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(buf.size() * 4)) || in.gcount() > 0) {
            const size_t n = (size_t)in.gcount() / 4;                 // only the bytes actually read (replay.rs:36 re-feeds stale tail)
            chan.send(Baseband{std::vector<float>(buf.begin(), buf.begin() + (long)n)});
            recv.run([](const std::vector<float>&) {});
        }
    } else {
        std::fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    }
    std::fprintf(stderr, "p25fe_replay: %zu dibits, %zu frame syncs, %zu power reports\n", sink.n_dibits, sink.n_sync,
                 hub.q.size());
    return 0;
}
