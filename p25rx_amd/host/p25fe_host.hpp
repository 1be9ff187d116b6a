// p25fe_host.hpp -- C++ host-side mirror of the reference's DemodTask / RecvTask over the C ABI.
//
// The reference is compiled Rust and this image has no Rust toolchain, so the host side above
// include/p25fe.h is C++ (INTEGRATION.md shows the Rust binding a maintainer would add instead).
// Same shape as the reference: DemodTask owns the demodulator state and turns u8 I/Q chunks into
// baseband chunks (src/demod.rs:25-120); RecvTask feeds baseband to the symbol receiver
// (src/recv.rs:140-167, 204-210).  Channels are any type with `bool recv(T&)` / `void send(T)`.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "p25fe.h"

namespace p25rx {

constexpr size_t BUF_BYTES = 32768;                 // src/consts.rs:6
constexpr size_t BUF_SAMPLES = BUF_BYTES / 2;       // src/consts.rs:8

// The reference aborts on every failure (`expect`, panic = "abort"); the wrapper restores that
// convention on top of the status codes of the C ABI.
inline void expect(int status, const char* what)
{
    if (status != P25FE_OK) {
        std::fprintf(stderr, "%s: %s (status %d)\n", what, p25fe_strerror(status), status);
        std::abort();
    }
}

class Handle {
public:
    explicit Handle(int device = 0, int n_channels = 1)
    {
        p25fe_config_t cfg;
        p25fe_default_config(&cfg);
        cfg.device = device;
        cfg.n_channels = n_channels;
        expect(p25fe_create(&cfg, &h_), "unable to create p25fe handle");
    }
    ~Handle() { p25fe_destroy(h_); }
    Handle(const Handle&) = delete;
    Handle& operator=(const Handle&) = delete;
    p25fe_t* get() const { return h_; }

private:
    p25fe_t* h_ = nullptr;
};

struct HubEvent { float signal_power_dbm; };                         // HubEvent::UpdateSignalPower, src/hub.rs:455
struct StatsEvent { uint64_t dibits, syncs; };                       // HubEvent::UpdateStats (src/recv.rs:162-165): what exists of Stats here
struct Baseband { std::vector<float> samples; };                     // RecvEvent::Baseband, src/recv.rs:25

// demod::DemodTask (src/demod.rs:25-120)
template <class Reader, class Hub, class Chan> class DemodTask {
public:
    DemodTask(Handle& h, Reader& reader, Hub& hub, Chan& chan) : h_(h), reader_(reader), hub_(hub), chan_(chan) {}

    // DemodTask::run (src/demod.rs:62-119); returns when the reader channel closes.
    void run()
    {
        std::vector<uint8_t> bytes;
        while (reader_.recv(bytes)) {                                 // :70
            Baseband bb;
            bb.samples.resize(bytes.size() / 2 / 5 + 2);
            size_t n_out = 0;
            float power = 0.f;
            const bool want = (++notifier_ % 4) == 0;                 // :95
            expect(p25fe_demod_u8(h_.get(), bytes.data(), bytes.size(), bb.samples.data(), bb.samples.size(), &n_out,
                                  want ? &power : nullptr),
                   "unable to demodulate");                           // :74-93, 97, 109-114
            bb.samples.resize(n_out);
            if (want) hub_.send(HubEvent{power});                     // :99
            chan_.send(std::move(bb));                                // :116
        }
    }

private:
    Handle& h_;
    Reader& reader_;
    Hub& hub_;
    Chan& chan_;
    // Throttler::new(4) (src/demod.rs:67): the reference creates it once, before the loop that never returns.  A
    // driver that calls run() once per chunk must see the same every-4th-chunk cadence, so the counter lives here.
    unsigned notifier_ = 0;
};

struct Symbols {
    std::vector<uint8_t> dibits;
    std::vector<int64_t> sync_pos;
    std::vector<uint64_t> sync_dibit;
};

// Sample path of recv::RecvTask (src/recv.rs:140-167, 204-210)
template <class Events, class Sink> class RecvTask {
public:
    RecvTask(Handle& h, Events& events, Sink& sink) : h_(h), events_(events), sink_(sink) {}

    void resync() { expect(p25fe_resync(h_.get()), "unable to resync"); }     // msg.resync(), src/recv.rs:136, 179

    // Throttler::new(16) of RecvTask::run (src/recv.rs:141, 162-165): a stats report after every 16th event.  The counter
    // lives in the task (the reference's run() never returns; a driver that calls run() per chunk keeps the cadence).
    template <class F, class StatsHub> void run(F cb, StatsHub& stats_hub)
    {
        Baseband bb;
        while (events_.recv(bb)) {
            feed(bb, cb);
            if (++stats_notifier_ % 16 == 0) stats_hub.send(StatsEvent{n_dibits_, n_sync_});
        }
    }

    template <class F> void run(F cb)                                 // RecvTask::run<F: FnMut(&[f32])>, :140
    {
        Baseband bb;
        while (events_.recv(bb)) {                                    // :144
            feed(bb, cb);
            ++stats_notifier_;
        }
    }

private:
    template <class F> void feed(Baseband& bb, F& cb)
    {
        {
            Symbols s;
            const size_t n = bb.samples.size();
            s.dibits.resize(n / 6 + 2);                              // hard ceiling: re-anchors are at least 6 samples apart
            s.sync_pos.resize(n / 6 + 2);
            s.sync_dibit.resize(n / 6 + 2);
            size_t nd = 0, ns = 0;
            expect(p25fe_slice(h_.get(), bb.samples.data(), n, s.dibits.data(), s.dibits.size(), &nd, s.sync_pos.data(),
                               s.sync_dibit.data(), s.sync_pos.size(), &ns),
                   "unable to slice");                                // :148-150
            s.dibits.resize(nd);
            s.sync_pos.resize(ns < s.sync_pos.size() ? ns : s.sync_pos.size());
            s.sync_dibit.resize(s.sync_pos.size());
            n_dibits_ += nd;
            n_sync_ += ns;
            sink_.send(std::move(s));
            cb(bb.samples);                                           // :152
        }
    }

    Handle& h_;
    Events& events_;
    Sink& sink_;
    unsigned stats_notifier_ = 0;
    uint64_t n_dibits_ = 0, n_sync_ = 0;
};

}  // namespace p25rx
