// p25fe_rccl.cpp -- include/p25fe_rccl.h: the time-sharded step (BASELINE.json config 5) over RCCL, behind the C ABI.
// Host code only (the kernels are libp25fe.so's); built into libp25fe_rccl.so, which links libp25fe.so and librccl.so.
#include "p25fe_rccl.h"

#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <new>
#include <vector>

namespace {

constexpr int SLOTS = 4;                  // >= the scratch sets the pipelined step rotates through
constexpr int RING = 64;                  // steps whose exchange events are kept for p25fe_shard_comm_ms
constexpr size_t SHM_HALO_MAX = 4096 * 8; // bytes of one halo slot in the test hook's shared segment

// TEST HOOK: the collectives through a POSIX shared-memory segment (every rank on the same GPU).  Layout: a sense-
// reversing barrier, one halo slot, one summary slot and one dibit row per rank.
struct ShmHdr {
    std::atomic<unsigned> arrived;
    std::atomic<unsigned> phase;
    unsigned world;
    unsigned pad;
    unsigned long long row_bytes;
};
struct Shm {
    ShmHdr* hd = nullptr;
    char* base = nullptr;
    size_t bytes = 0;
    char* halo(int r) const { return base + sizeof(ShmHdr) + (size_t)r * SHM_HALO_MAX; }
    char* summ(int r) const { return base + sizeof(ShmHdr) + (size_t)hd->world * SHM_HALO_MAX + (size_t)r * sizeof(p25fe_result_t); }
    char* row(int r) const
    {
        return base + sizeof(ShmHdr) + (size_t)hd->world * (SHM_HALO_MAX + sizeof(p25fe_result_t)) + (size_t)r * hd->row_bytes;
    }
    void barrier() const
    {
        const unsigned ph = hd->phase.load(std::memory_order_acquire);
        if (hd->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == hd->world) {
            hd->arrived.store(0, std::memory_order_relaxed);
            hd->phase.store(ph + 1, std::memory_order_release);
        } else {
            while (hd->phase.load(std::memory_order_acquire) == ph) usleep(50);
        }
    }
};

}  // namespace

struct p25fe_shard {
    p25fe_t* h = nullptr;
    int rank = 0, world = 1;
    size_t n = 0, halo = 0, cap = 0;
    bool staged = false;
    ncclComm_t comm = nullptr;
    ncclComm_t comm_halo = nullptr;       // pipelined steps: the halo travels on a communicator of its own (ncclCommSplit), on K1's stream
    ncclComm_t comm_summ = nullptr;       // ... and so do the summaries: RCCL runs the operations of ONE communicator in issue order, whatever
                                          // their streams -- step j + 1's all-gather (receive stream) would wait for step j's dibit gather (side
                                          // stream, megabytes at N = 8) and miss the K1 boundary both were meant to share
    hipEvent_t e_stage = nullptr;         // pipelined steps: summaries gathered (receive stream) -> pass 2 and the dibit gather (side stream)
    int pipe_layout = 2;                  // measurement knob P25FE_SHARD_PIPE_LAYOUT: 1 = the step's own order on the receive stream
    Shm shm;
    hipStream_t cs = nullptr;             // the halo exchange and the shard's head segment run beside K1's main launch
    hipEvent_t e_fork = nullptr, e_head = nullptr;
    hipEvent_t ev[RING][6];               // halo begin / end, all-gather begin / end, gather begin / end (timed steps only)
    int timing_every = 16;                // every k-th step carries those events (0: none): four of them sit between the kernels of
                                          // the step's critical path and cost ~4 us each there
    uint64_t steps = 0, timed = 0, read_from = 0;
    std::vector<uint64_t> bb0, bbn;
    // Pass 1's summary and the gathered summaries live in rings of SLOTS entries indexed by the step: in the pipelined step stage 1 of
    // step j + 1 (receive stream) runs beside stage 2 of step j (side stream), which still reads step j's summaries and writes the
    // caller's result record -- pass 1 therefore never writes that record, and never the summaries of the step before.
    p25fe_result_t* d_summ = nullptr;     // [SLOTS][world]
    p25fe_result_t* d_res1 = nullptr;     // [SLOTS]
    uint64_t *d_bb0 = nullptr, *d_bbn = nullptr, *d_off = nullptr, *d_off_x = nullptr;
    p25fe_anchor_t *d_anc = nullptr, *d_anc_x = nullptr;
    uint8_t *d_gathered = nullptr, *d_stream = nullptr;
    uint8_t* d_stream2 = nullptr;         // pipelined steps alternate between the two ordered-stream buffers (d_stream is always the LAST step's)
    char* d_loop = nullptr;               // one-rank RCCL group (tests on a 1-GPU box): where the halo loops back to
    uint64_t* h_off = nullptr;            // pinned: the world + 1 offsets of the current step (P25FE_GATHER_ROOT_EXACT)
    hipEvent_t e_res = nullptr, e_off = nullptr;
    int gather_ran = P25FE_GATHER_NONE;   // how the last step's dibits actually travelled
    bool head_event_wait = false;         // measurement knob (P25FE_SHARD_HEAD_WAIT=event)
    bool broken = false;                  // a collective failed half-way: the communicator's state is unknown, every later step fails
    bool coll_issued = false;             // the current step has put a collective on the wire (a failure behind it retires the object)
    int rccl_ranks = 0, rccl_rank = -1;   // what the communicator itself says (ncclCommCount / ncclCommUserRank): p25fe_shard_info
    char pci[32] = {0};                   // hipDeviceGetPCIBusId of the handle's device
};

#define HCHK(x) do { if ((x) != hipSuccess) return P25FE_ERR_HIP; } while (0)
#define NCHK(x) do { if ((x) != ncclSuccess) return P25FE_ERR_HIP; } while (0)
// inside ncclGroupStart / ncclGroupEnd: close the group before returning (an open group would swallow every later collective
// of this communicator) and retire the shard object
#define NCHK_G(s, x) do { if ((x) != ncclSuccess) { (void)ncclGroupEnd(); (s)->broken = true; return P25FE_ERR_HIP; } } while (0)

// The side stream must not sit on the hardware queue of a stream whose work it is meant to run BESIDE: HIP hands the streams of one
// priority level four queues to share, in creation order, and two streams on one queue run their kernels one after the other -- the halo
// would no longer hide behind K1, the pipelined step's second stage would hold up the next K1 (measured: 0.35 - 0.36 ms per step instead
// of 0.28 / 0.33).  The probe synchronises the streams it compares (< 1 ms each), so it runs where that is allowed: in p25fe_shard_create
// (against the handle's receive stream) and in p25fe_shard_prepare (against a caller's stream and the receive stream) -- never inside a
// step (a step must not synchronise the host, and a capturing stream cannot be probed).  A side stream that shares is replaced
// by a fresh one (the rejected ones stay alive until the search ends: their queue slots stay taken).
static int side_stream_apart(p25fe_shard_t* s, hipStream_t a, hipStream_t b)
{
    const char* e = getenv("P25FE_SHARD_QUEUE_PROBE");
    if (e && atoi(e) == 0) return P25FE_OK;
    std::vector<hipStream_t> rejected;
    try {
        for (int attempt = 0; attempt < 8; ++attempt) {
            int sh_a = 0, sh_b = 0;
            if (p25fe_streams_share_queue(s->h, a, s->cs, &sh_a) != P25FE_OK) break;
            if (b && b != a && p25fe_streams_share_queue(s->h, b, s->cs, &sh_b) != P25FE_OK) break;
            if (!sh_a && !sh_b) break;
            hipStream_t fresh = nullptr;
            if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) break;
            rejected.push_back(s->cs);
            s->cs = fresh;
        }
    } catch (...) {}
    for (hipStream_t r : rejected) { (void)hipStreamSynchronize(r); (void)hipStreamDestroy(r); }
    return P25FE_OK;
}

extern "C" {

int p25fe_rccl_unique_id(void* id128)
{
    static_assert(sizeof(ncclUniqueId) <= P25FE_RCCL_ID_BYTES, "id size");
    if (!id128) return P25FE_ERR_ARG;
    ncclUniqueId id;
    NCHK(ncclGetUniqueId(&id));
    memset(id128, 0, P25FE_RCCL_ID_BYTES);
    memcpy(id128, &id, sizeof id);
    return P25FE_OK;
}

size_t p25fe_shard_dibit_cap(const p25fe_shard_t* s) { return s ? s->cap : 0; }

void p25fe_shard_destroy(p25fe_shard_t* s)
{
    if (!s) return;
    if (s->h) (void)hipSetDevice(p25fe_device(s->h));
    if (s->cs) { (void)hipStreamSynchronize(s->cs); (void)hipStreamDestroy(s->cs); }
    if (s->comm_halo) (void)ncclCommDestroy(s->comm_halo);
    if (s->comm_summ) (void)ncclCommDestroy(s->comm_summ);
    if (s->comm) (void)ncclCommDestroy(s->comm);
    if (s->e_stage) (void)hipEventDestroy(s->e_stage);
    if (s->e_fork) (void)hipEventDestroy(s->e_fork);
    if (s->e_head) (void)hipEventDestroy(s->e_head);
    if (s->e_res) (void)hipEventDestroy(s->e_res);
    if (s->e_off) (void)hipEventDestroy(s->e_off);
    if (s->h_off) (void)hipHostFree(s->h_off);
    for (auto& row : s->ev) for (hipEvent_t e : row) if (e) (void)hipEventDestroy(e);
    void* bufs[] = {s->d_summ, s->d_res1, s->d_bb0, s->d_bbn, s->d_off, s->d_off_x, s->d_anc, s->d_anc_x, s->d_gathered, s->d_stream, s->d_stream2, s->d_loop};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (s->shm.base) munmap(s->shm.base, s->shm.bytes);
    delete s;
}

int p25fe_shard_create(p25fe_t* h, int rank, int world, const void* id128, size_t n_per_rank, p25fe_shard_t** out)
{
    if (!h || !out || world < 1 || rank < 0 || rank >= world || n_per_rank == 0 || (n_per_rank % 8) != 0) return P25FE_ERR_ARG;
    *out = nullptr;
    // streams, events, buffers and the communicator belong to the handle's device, whatever the caller's current one is
    if (hipSetDevice(p25fe_device(h)) != hipSuccess) return P25FE_ERR_HIP;
    p25fe_shard_t* s = new (std::nothrow) p25fe_shard;
    if (!s) return P25FE_ERR_NOMEM;
    for (auto& row : s->ev) for (hipEvent_t& e : row) e = nullptr;
    s->h = h; s->rank = rank; s->world = world; s->n = n_per_rank; s->halo = p25fe_shard_halo();
    size_t bbmax = 0;
    try {                                                           // no exception crosses the C boundary
        for (int r = 0; r < world; ++r) {
            s->bb0.push_back(p25fe_n_baseband_h(h, 0, (size_t)r * n_per_rank));
            s->bbn.push_back(p25fe_n_baseband_h(h, (uint64_t)r * n_per_rank, n_per_rank));
            if (s->bbn.back() > bbmax) bbmax = (size_t)s->bbn.back();
        }
    } catch (...) {
        delete s;
        return P25FE_ERR_NOMEM;
    }
    // a receiver that re-anchors on every sync word follows the TRANSMITTER's symbol clock: proportional slack (200 ppm)
    s->cap = ((bbmax / 10 + bbmax / 50000 + 64) + 15) / 16 * 16;
    auto fail = [&](int code) { p25fe_shard_destroy(s); return code; };
    {
        // measurement knob P25FE_SHARD_CS_PRIO=1: the side stream at the highest priority.  Its halo exchange competes with K1's
        // 33 000 one-wave workgroups for wave slots and gets in when K1 drains either way (same box, interleaved: 0.3280 /
        // 0.3281 ms per step with, 0.3240 / 0.3261 without -- profiles/r05_shard_step_ab.txt)
        const char* pe = getenv("P25FE_SHARD_CS_PRIO");
        int lo = 0, hi = 0;
        if (pe && atoi(pe) == 1 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
            hipStreamCreateWithPriority(&s->cs, hipStreamNonBlocking, hi) == hipSuccess) {
        } else {
            (void)hipGetLastError();
            s->cs = nullptr;
            if (hipStreamCreateWithFlags(&s->cs, hipStreamNonBlocking) != hipSuccess) return fail(P25FE_ERR_HIP);
        }
    }
    if (hipEventCreateWithFlags(&s->e_stage, hipEventDisableTiming) != hipSuccess) return fail(P25FE_ERR_HIP);
    if (hipEventCreateWithFlags(&s->e_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->e_head, hipEventDisableTiming) != hipSuccess) return fail(P25FE_ERR_HIP);
    for (auto& row : s->ev) for (hipEvent_t& e : row) if (hipEventCreate(&e) != hipSuccess) return fail(P25FE_ERR_HIP);
    if (hipEventCreateWithFlags(&s->e_res, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->e_off, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&s->h_off), ((size_t)world + 1) * 8, hipHostMallocDefault) != hipSuccess)
        return fail(P25FE_ERR_HIP);
    const size_t W = (size_t)world;
    if (hipMalloc(&s->d_summ, SLOTS * W * sizeof(p25fe_result_t)) != hipSuccess || hipMalloc(&s->d_res1, SLOTS * sizeof(p25fe_result_t)) != hipSuccess ||
        hipMalloc(&s->d_bb0, W * 8) != hipSuccess ||
        hipMalloc(&s->d_bbn, W * 8) != hipSuccess || hipMalloc(&s->d_off, (W + 1) * 8) != hipSuccess ||
        hipMalloc(&s->d_off_x, (W + 1) * 8) != hipSuccess || hipMalloc(&s->d_anc_x, W * sizeof(p25fe_anchor_t)) != hipSuccess ||
        hipMalloc(&s->d_anc, W * sizeof(p25fe_anchor_t)) != hipSuccess || hipMalloc(&s->d_gathered, W * s->cap) != hipSuccess ||
        hipMalloc(&s->d_stream, W * s->cap) != hipSuccess || hipMalloc(&s->d_stream2, W * s->cap) != hipSuccess)
        return fail(P25FE_ERR_NOMEM);
    {
        const char* te = getenv("P25FE_SHARD_TIMING");                // (measurement knob; p25fe_shard_comm_timing is the API)
        if (te && *te) s->timing_every = atoi(te) < 0 ? 0 : atoi(te);
        const char* hw = getenv("P25FE_SHARD_HEAD_WAIT");
        s->head_event_wait = hw && !strcmp(hw, "event");
        const char* pl = getenv("P25FE_SHARD_PIPE_LAYOUT");
        if (pl && atoi(pl) == 1) s->pipe_layout = 1;
    }
    if (hipMemcpy(s->d_bb0, s->bb0.data(), W * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(s->d_bbn, s->bbn.data(), W * 8, hipMemcpyHostToDevice) != hipSuccess)
        return fail(P25FE_ERR_HIP);
    if (world > 1 || id128) {
        if (id128) {
            ncclUniqueId id;
            memcpy(&id, id128, sizeof id);
            if (ncclCommInitRank(&s->comm, world, id, rank) != ncclSuccess) return fail(P25FE_ERR_HIP);
            if (world == 1 && hipMalloc(&s->d_loop, s->halo * 8) != hipSuccess) return fail(P25FE_ERR_NOMEM);
            if (ncclCommCount(s->comm, &s->rccl_ranks) != ncclSuccess || ncclCommUserRank(s->comm, &s->rccl_rank) != ncclSuccess ||
                s->rccl_ranks != world || s->rccl_rank != rank)
                return fail(P25FE_ERR_HIP);                              // the communicator is not the one the caller described
            // Two more communicators for the pipelined step (halo, summaries).  ncclCommSplit is a collective on the parent: EVERY rank
            // makes BOTH calls whatever the first one returned here (a rank that skipped the second would leave its peers inside it),
            // and what the step may use is then AGREED: an all-reduce (min) of {halo split ok, summary split ok, layout wanted} on the
            // parent communicator -- ranks that disagreed would issue the halo and the summaries on different communicators and
            // streams and hang without a word.  Any rank without a split, or asking for layout 1, puts every rank on the
            // one-communicator layout.  (Test knob P25FE_SHARD_SPLIT_FAIL = "all" or a rank number: that rank discards its splits.)
            const bool ok_h = ncclCommSplit(s->comm, 0, rank, &s->comm_halo, nullptr) == ncclSuccess && s->comm_halo;
            const bool ok_s = ncclCommSplit(s->comm, 0, rank, &s->comm_summ, nullptr) == ncclSuccess && s->comm_summ;
            int agree[3] = {ok_h ? 1 : 0, ok_s ? 1 : 0, s->pipe_layout};
            if (const char* sf = getenv("P25FE_SHARD_SPLIT_FAIL"))
                if (!strcmp(sf, "all") || (*sf >= '0' && *sf <= '9' && atoi(sf) == rank)) agree[0] = agree[1] = 0;
            int* d_agree = nullptr;
            if (hipMalloc(&d_agree, sizeof agree) != hipSuccess) return fail(P25FE_ERR_NOMEM);
            bool agreed = hipMemcpy(d_agree, agree, sizeof agree, hipMemcpyHostToDevice) == hipSuccess &&
                          ncclAllReduce(d_agree, d_agree, 3, ncclInt32, ncclMin, s->comm, s->cs) == ncclSuccess &&
                          hipStreamSynchronize(s->cs) == hipSuccess &&
                          hipMemcpy(agree, d_agree, sizeof agree, hipMemcpyDeviceToHost) == hipSuccess;
            (void)hipFree(d_agree);
            if (!agreed) return fail(P25FE_ERR_HIP);
            if (!(agree[0] && agree[1]) || agree[2] != 2) {
                if (s->comm_halo) { (void)ncclCommDestroy(s->comm_halo); s->comm_halo = nullptr; }
                if (s->comm_summ) { (void)ncclCommDestroy(s->comm_summ); s->comm_summ = nullptr; }
                s->pipe_layout = 1;
            }
        } else {
            const char* name = getenv("P25FE_SHARD_SHM");
            if (!name || s->halo * 8 > SHM_HALO_MAX) return fail(P25FE_ERR_ARG);
            s->staged = true;
            s->shm.bytes = sizeof(ShmHdr) + W * (SHM_HALO_MAX + sizeof(p25fe_result_t) + s->cap);
            const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)s->shm.bytes) != 0) { if (fd >= 0) close(fd); return fail(P25FE_ERR_ARG); }
            void* p = mmap(nullptr, s->shm.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (p == MAP_FAILED) return fail(P25FE_ERR_NOMEM);
            s->shm.base = static_cast<char*>(p);
            s->shm.hd = reinterpret_cast<ShmHdr*>(p);          // a fresh segment is zero-filled: barrier state starts at 0
            s->shm.hd->world = (unsigned)world;
            s->shm.hd->row_bytes = s->cap;
        }
    }
    if (hipDeviceGetPCIBusId(s->pci, (int)sizeof s->pci, p25fe_device(h)) != hipSuccess) { (void)hipGetLastError(); s->pci[0] = 0; }
    // the handle's own streams must not share a hardware queue with the side stream (p25fe_shard_prepare does the same for a caller's)
    if (!s->staged) {
        void* rx = nullptr;
        if (p25fe_rx_stream(h, &rx) == P25FE_OK) (void)side_stream_apart(s, (hipStream_t)rx, nullptr);
    }
    *out = s;
    return P25FE_OK;
}

int p25fe_shard_prepare(p25fe_shard_t* s, void* stream)
{
    if (!s) return P25FE_ERR_ARG;
    if (s->staged) return P25FE_OK;
    HCHK(hipSetDevice(p25fe_device(s->h)));
    void* rx = nullptr;
    const int rc = p25fe_rx_stream(s->h, &rx);
    if (rc) return rc;
    return side_stream_apart(s, (hipStream_t)stream, (hipStream_t)rx);
}

int p25fe_shard_info(const p25fe_shard_t* s, p25fe_shard_info_t* out)
{
    if (!s || !out) return P25FE_ERR_ARG;
    memset(out, 0, sizeof *out);
    out->rank = s->rank; out->world = s->world;
    out->rccl_ranks = s->comm ? s->rccl_ranks : 0; out->rccl_rank = s->comm ? s->rccl_rank : -1;
    out->device = p25fe_device(s->h);
    out->comms = !s->comm ? 0 : ((s->comm_halo && s->comm_summ) ? 3 : 1);
    out->pipe_layout = (s->pipe_layout == 2 && (s->comm_halo || !s->comm)) ? 2 : 1;
    out->gather_ran = s->gather_ran;
    out->staged = s->staged ? 1 : 0;
    out->head_wait = s->head_event_wait ? 1 : 0;
    out->broken = s->broken ? 1 : 0;
    out->steps = s->steps;
    memcpy(out->pci_bus_id, s->pci, sizeof out->pci_bus_id);
    return P25FE_OK;
}

int p25fe_shard_comm_timing(p25fe_shard_t* s, int every)
{
    if (!s || every < 0) return P25FE_ERR_ARG;
    s->timing_every = every;
    return P25FE_OK;
}

int p25fe_shard_gather_ran(const p25fe_shard_t* s) { return s ? s->gather_ran : P25FE_ERR_ARG; }

// The world + 1 offsets of this step on the HOST (P25FE_GATHER_ROOT_EXACT: the counts are host arguments of the sends and
// receives): a one-thread resolve beside pass 2, into buffers of its own, then one small copy to pinned memory.
static int exact_offsets_begin(p25fe_shard_t* s, const p25fe_result_t* summ, hipStream_t st)
{
    HCHK(hipEventRecord(s->e_res, st));                              // the summaries are in d_summ
    HCHK(hipStreamWaitEvent(s->cs, s->e_res, 0));
    const int rc = p25fe_shard_resolve_dev(s->h, summ, s->d_bb0, s->d_bbn, (size_t)s->world, s->d_anc_x, s->d_off_x, s->cs);
    if (rc) return rc;
    HCHK(hipMemcpyAsync(s->h_off, s->d_off_x, ((size_t)s->world + 1) * 8, hipMemcpyDeviceToHost, s->cs));
    HCHK(hipEventRecord(s->e_off, s->cs));
    return P25FE_OK;
}
static int exact_offsets_wait(p25fe_shard_t* s)
{
    HCHK(hipEventSynchronize(s->e_off));                             // the step's ONE host wait
    const uint64_t* off = s->h_off;
    const size_t room = (size_t)s->world * s->cap;
    for (int r = 0; r < s->world; ++r)
        if (off[r + 1] < off[r] || off[r + 1] - off[r] > s->cap || off[r + 1] > room) return P25FE_ERR_CAPACITY;   // (also reported by p25fe_shard_offsets)
    return P25FE_OK;
}

} // extern "C"

// Where the launches of one step go.
//   plain step:       everything on the caller's stream; halo + head on the side stream beside K1's main launch.
//   pipelined, 1:     K1's main launch on the caller's stream, everything behind it on the handle's receive stream (rx = rx2).
//   pipelined, 2:     the halo (its own communicator) and then the WHOLE front end on the caller's stream -- an RCCL kernel is one
//                     256-thread workgroup with 132 VGPRs and 20 KB of LDS, which finds no room on a chip K1's one-wave workgroups keep
//                     full (measured: it starts when K1 drains), so the exchange is put where the chip IS drained: between two K1s;
//                     detection, scan and the summary all-gather on the receive stream (rx), pass 2, the dibit gather and the
//                     compaction on the side stream (rx2): two stages, each one step long, so that the exchange that ends a stage
//                     finds its K1 boundary without holding up the next step's first stage.
struct StepStreams { hipStream_t st, rx, rx2; bool halo_on_st; };
static bool step_args_ok(const void* d_buf, int fmt, const void* d_dibits, const void* d_result, int gather)
{
    return d_buf && d_dibits && d_result && (fmt == P25FE_FMT_CF32 || fmt == P25FE_FMT_U8) && gather >= P25FE_GATHER_NONE &&
           gather <= P25FE_GATHER_ROOT_EXACT;
}
static int shard_step_impl(p25fe_shard_t* s, void* d_buf, int fmt, uint8_t* d_dibits, p25fe_result_t* d_result, int gather,
                           const StepStreams& ss)
{
    const hipStream_t st = ss.st, rx = ss.rx;
    if (!s || !step_args_ok(d_buf, fmt, d_dibits, d_result, gather)) return P25FE_ERR_ARG;
    const size_t eb = fmt == P25FE_FMT_CF32 ? 8 : 2;
    char* buf = static_cast<char*>(d_buf);
    char* owned = buf + s->halo * eb;
    const size_t n_hist = s->rank > 0 ? s->halo : 0;
    const uint64_t abs0 = (uint64_t)s->rank * s->n;
    const bool multi = s->world > 1 || s->comm != nullptr;
    const bool rccl = multi && !s->staged;
    const bool loopback = rccl && s->world == 1;                     // a one-rank communicator (tests on a 1-GPU box): the root plays its own peer
    // HIP events around the three exchanges: only on every timing_every-th step -- four of them are packets BETWEEN the
    // kernels of the step's critical path
    const bool timed = rccl && s->timing_every > 0 && (s->steps % (uint64_t)s->timing_every) == 0;
    hipEvent_t* ev = s->ev[s->timed % RING];
    p25fe_result_t* const res1 = s->d_res1 + (s->steps % SLOTS);                       // pass 1's summary of THIS step
    p25fe_result_t* const summ = s->d_summ + (s->steps % SLOTS) * (size_t)s->world;    // every rank's, gathered
    int rc;
    if (multi) {
        // ---- 1. halo: my last `halo` samples -> rank + 1, rank - 1's -> the front of my buffer, beside K1's main launch
        if (s->staged) {
            HCHK(hipStreamSynchronize(st));
            if (s->rank + 1 < s->world) HCHK(hipMemcpy(s->shm.halo(s->rank), buf + s->n * eb, s->halo * eb, hipMemcpyDeviceToHost));
            s->shm.barrier();
            if (s->rank > 0) HCHK(hipMemcpy(buf, s->shm.halo(s->rank - 1), s->halo * eb, hipMemcpyHostToDevice));
            s->shm.barrier();
            rc = p25fe_shard_pass1(s->h, owned, fmt, s->n, n_hist, s->n, abs0, res1, st);
            if (rc) return rc;
        } else {
            const hipStream_t hs = ss.halo_on_st ? st : s->cs;
            const ncclComm_t hc = ss.halo_on_st ? s->comm_halo : s->comm;
            if (!ss.halo_on_st) {
                HCHK(hipEventRecord(s->e_fork, st));             // the exchange may not overtake earlier users of the buffers
                HCHK(hipStreamWaitEvent(s->cs, s->e_fork, 0));
            }
            if (timed) HCHK(hipEventRecord(ev[0], hs));
            s->coll_issued = true;
            NCHK(ncclGroupStart());
            if (s->rank + 1 < s->world) NCHK_G(s, ncclSend(buf + s->n * eb, s->halo * eb, ncclUint8, s->rank + 1, hc, hs));
            if (s->rank > 0) NCHK_G(s, ncclRecv(buf, s->halo * eb, ncclUint8, s->rank - 1, hc, hs));
            if (loopback) {                                      // loop the halo back
                NCHK_G(s, ncclSend(buf + s->n * eb, s->halo * eb, ncclUint8, 0, hc, hs));
                NCHK_G(s, ncclRecv(s->d_loop, s->halo * eb, ncclUint8, 0, hc, hs));
            }
            if (ncclGroupEnd() != ncclSuccess) { s->broken = true; return P25FE_ERR_HIP; }
            if (timed) HCHK(hipEventRecord(ev[1], hs));
            if (ss.halo_on_st) {
                rc = p25fe_shard_pass1_k1(s->h, owned, fmt, s->n, n_hist, s->n, abs0, st);       // the halo is in: one launch, head included
                if (rc) return rc;
                rc = p25fe_shard_pass1_finish(s->h, owned, fmt, s->n, n_hist, s->n, abs0, res1, rx);
                if (rc) return rc;
            } else {
            rc = p25fe_shard_pass1_main(s->h, owned, fmt, s->n, n_hist, s->n, abs0, st);
            if (rc) return rc;
            // the head segment (the one workgroup whose input reaches into the halo) follows the halo on ITS stream, beside
            // the main launch; the compute stream only waits for it before the sync detection
            rc = p25fe_shard_pass1_head(s->h, owned, fmt, s->n, n_hist, s->n, abs0, s->cs);
            if (rc) return rc;
            // No event wait between the streams here: the detection's first tile waits for a word a one-thread kernel behind
            // the head writes (p25fe_shard_pass1_head); P25FE_SHARD_HEAD_WAIT=event restores the stream-level wait (A / B:
            // it costs ~10 us of idle GPU between K1 and the detection).
            if (s->head_event_wait) {
                HCHK(hipEventRecord(s->e_head, s->cs));
                HCHK(hipStreamWaitEvent(rx, s->e_head, 0));
            }
            rc = p25fe_shard_pass1_finish(s->h, owned, fmt, s->n, n_hist, s->n, abs0, res1, rx);
            if (rc) return rc;
            }
        }
        // ---- 2. one summary per rank to every rank
        if (s->staged) {
            HCHK(hipStreamSynchronize(rx));
            HCHK(hipMemcpy(s->shm.summ(s->rank), res1, sizeof(p25fe_result_t), hipMemcpyDeviceToHost));
            s->shm.barrier();
            HCHK(hipMemcpy(summ, s->shm.summ(0), (size_t)s->world * sizeof(p25fe_result_t), hipMemcpyHostToDevice));
            s->shm.barrier();
        } else {
            if (timed) HCHK(hipEventRecord(ev[2], rx));
            NCHK(ncclAllGather(res1, summ, sizeof(p25fe_result_t), ncclUint8, (ss.halo_on_st && s->comm_summ) ? s->comm_summ : s->comm, rx));
            if (timed) HCHK(hipEventRecord(ev[3], rx));
        }
    } else {
        rc = p25fe_shard_pass1_k1(s->h, owned, fmt, s->n, n_hist, s->n, abs0, st);              // (no halo to wait for: one launch)
        if (rc) return rc;
        rc = p25fe_shard_pass1_finish(s->h, owned, fmt, s->n, n_hist, s->n, abs0, res1, rx);
        if (rc) return rc;
        HCHK(hipMemcpyAsync(summ, res1, sizeof(p25fe_result_t), hipMemcpyDeviceToDevice, rx));
    }
    const bool exact = gather == P25FE_GATHER_ROOT_EXACT && multi;
    if (exact) {
        rc = exact_offsets_begin(s, summ, rx);
        if (rc) return rc;
    }
    if (ss.rx2 != ss.rx) {                                           // second stage: its stream takes over behind the summaries
        HCHK(hipEventRecord(s->e_stage, ss.rx));
        HCHK(hipStreamWaitEvent(ss.rx2, s->e_stage, 0));
    }
    // ---- 3. pass 2 with the combine inside it (carry-in anchor, dibit offsets).  Rank 0's shard starts at offset 0 of the
    // ordered stream: it slices straight into it as well (the loopback test ranks send to themselves instead)
    const bool to_root = gather == P25FE_GATHER_ROOT || gather == P25FE_GATHER_ROOT_EXACT;
    uint8_t* dup = (s->rank == 0 && !loopback && (to_root || !multi) && gather != P25FE_GATHER_NONE) ? s->d_stream : nullptr;
    rc = p25fe_shard_pass2_dev(s->h, summ, s->d_bb0, s->d_bbn, (size_t)s->world, (size_t)s->rank, s->d_anc, s->d_off, d_dibits, s->cap,
                               dup, d_result, ss.rx2);
    if (rc) return rc;
    s->gather_ran = multi ? gather : (gather == P25FE_GATHER_NONE ? P25FE_GATHER_NONE : P25FE_GATHER_ROOT);
    // ---- 4. the reduced dibit stream
    if (gather != P25FE_GATHER_NONE && multi) {
        size_t first_row = 1;                                        // rows the compaction pass has to move (ROOT / ALL)
        bool compact = false;
        if (exact) {
            // "the reduced dibit stream": exactly offsets[r + 1] - offsets[r] bytes per shard, received AT offsets[r] of the
            // ordered stream -- no padded rows on the wire, no compaction pass.  Costs the one host wait of the step (the
            // send / recv counts are host arguments); pass 2 is already enqueued and runs meanwhile.
            rc = exact_offsets_wait(s);
            if (rc) return rc;
            const uint64_t* off = s->h_off;
            if (s->staged) {
                HCHK(hipStreamSynchronize(ss.rx2));
                const size_t mine = (size_t)(off[s->rank + 1] - off[s->rank]);
                if (s->rank > 0 && mine) HCHK(hipMemcpy(s->shm.row(s->rank), d_dibits, mine, hipMemcpyDeviceToHost));
                s->shm.barrier();
                if (s->rank == 0)
                    for (int r = 1; r < s->world; ++r)
                        if (off[r + 1] > off[r]) HCHK(hipMemcpy(s->d_stream + off[r], s->shm.row(r), (size_t)(off[r + 1] - off[r]), hipMemcpyHostToDevice));
                s->shm.barrier();
            } else {
                if (timed) HCHK(hipEventRecord(ev[4], ss.rx2));
                NCHK(ncclGroupStart());
                if (s->rank == 0) {
                    for (int r = loopback ? 0 : 1; r < s->world; ++r)
                        if (off[r + 1] > off[r]) NCHK_G(s, ncclRecv(s->d_stream + off[r], (size_t)(off[r + 1] - off[r]), ncclUint8, r, s->comm, ss.rx2));
                }
                if ((s->rank > 0 || loopback) && off[s->rank + 1] > off[s->rank])
                    NCHK_G(s, ncclSend(d_dibits, (size_t)(off[s->rank + 1] - off[s->rank]), ncclUint8, 0, s->comm, ss.rx2));
                if (ncclGroupEnd() != ncclSuccess) { s->broken = true; return P25FE_ERR_HIP; }
                if (timed) HCHK(hipEventRecord(ev[5], ss.rx2));
            }
        } else if (s->staged) {
            HCHK(hipStreamSynchronize(ss.rx2));
            HCHK(hipMemcpy(s->shm.row(s->rank), d_dibits, s->cap, hipMemcpyDeviceToHost));
            s->shm.barrier();
            if (s->rank == 0 || gather == P25FE_GATHER_ALL) {
                first_row = gather == P25FE_GATHER_ALL ? 0 : 1;
                if ((size_t)s->world > first_row)
                    HCHK(hipMemcpy(s->d_gathered + first_row * s->cap, s->shm.row((int)first_row), ((size_t)s->world - first_row) * s->cap, hipMemcpyHostToDevice));
                compact = true;
            }
            s->shm.barrier();
        } else {
            if (timed) HCHK(hipEventRecord(ev[4], ss.rx2));
            if (gather == P25FE_GATHER_ALL) {
                NCHK(ncclAllGather(d_dibits, s->d_gathered, s->cap, ncclUint8, s->comm, ss.rx2));
                first_row = 0;
                compact = true;
            } else {
                // point-to-point to the root: every rank has its own xGMI link to it, the shards arrive in parallel (an
                // all-gather would move `world` times the bytes the one consumer needs around a per-link-bound ring)
                NCHK(ncclGroupStart());
                if (s->rank == 0)
                    for (int r = loopback ? 0 : 1; r < s->world; ++r) NCHK_G(s, ncclRecv(s->d_gathered + (size_t)r * s->cap, s->cap, ncclUint8, r, s->comm, ss.rx2));
                if (s->rank > 0 || loopback) NCHK_G(s, ncclSend(d_dibits, s->cap, ncclUint8, 0, s->comm, ss.rx2));
                if (ncclGroupEnd() != ncclSuccess) { s->broken = true; return P25FE_ERR_HIP; }
                if (s->rank == 0) { first_row = loopback ? 0 : 1; compact = true; }
            }
            if (timed) HCHK(hipEventRecord(ev[5], ss.rx2));
        }
        if (compact) {
            rc = p25fe_shard_compact_from_dev(s->h, s->d_gathered, s->cap, s->d_off, first_row, (size_t)s->world, s->d_stream,
                                              (size_t)s->world * s->cap, ss.rx2);
            if (rc) return rc;
        }
    }
    if (timed) ++s->timed;
    ++s->steps;
    return P25FE_OK;
}

extern "C" {

int p25fe_shard_step(p25fe_shard_t* s, void* d_buf, int fmt, uint8_t* d_dibits, p25fe_result_t* d_result, int gather, void* stream)
{
    if (!s) return P25FE_ERR_ARG;
    if (s->broken) return P25FE_ERR_HIP;
    HCHK(hipSetDevice(p25fe_device(s->h)));
    const StepStreams ss = {(hipStream_t)stream, (hipStream_t)stream, (hipStream_t)stream, false};
    s->coll_issued = false;
    const int rc = shard_step_impl(s, d_buf, fmt, d_dibits, d_result, gather, ss);
    if (rc && s->coll_issued) s->broken = true;                     // peers may already be inside an exchange this rank will not finish
    return rc;
}

int p25fe_shard_step_pipelined(p25fe_shard_t* s, void* d_buf, int fmt, uint8_t* d_dibits, p25fe_result_t* d_result, int gather, void* stream)
{
    if (!s) return P25FE_ERR_ARG;
    if (s->broken) return P25FE_ERR_HIP;
    HCHK(hipSetDevice(p25fe_device(s->h)));
    if (!step_args_ok(d_buf, fmt, d_dibits, d_result, gather)) return P25FE_ERR_ARG;      // before any state moves (scratch rotation, buffer swap)
    s->coll_issued = false;
    if (s->staged) {                                                 // the test hook synchronises the host between its phases: nothing to overlap
        const StepStreams ss = {(hipStream_t)stream, (hipStream_t)stream, (hipStream_t)stream, false};
        return shard_step_impl(s, d_buf, fmt, d_dibits, d_result, gather, ss);
    }
    void* rx = nullptr;
    int rc = p25fe_shard_pipe_begin(s->h, stream, &rx);
    if (rc) return rc;
    // the ordered stream of the PREVIOUS step stays readable (by work enqueued on `stream` before this call) while this step's
    // slicer / gather write the other buffer
    uint8_t* t = s->d_stream; s->d_stream = s->d_stream2; s->d_stream2 = t;
    // layout 2 needs the halo's own communicator (or no exchange at all); otherwise the step's own order on the receive stream
    const bool two_stage = s->pipe_layout == 2 && (s->comm_halo || !s->comm);
    const StepStreams ss = {(hipStream_t)stream, (hipStream_t)rx, two_stage ? s->cs : (hipStream_t)rx, two_stage};
    rc = shard_step_impl(s, d_buf, fmt, d_dibits, d_result, gather, ss);
    if (rc) {
        // a failure half-way: whatever has been enqueued stays joinable -- the stream the "done" event goes on first waits for the other
        // stage's (the event alone would not cover detection / scan / all-gather already on the receive stream)
        if (s->coll_issued) s->broken = true;
        if (ss.rx2 != ss.rx && hipEventRecord(s->e_stage, ss.rx) == hipSuccess) (void)hipStreamWaitEvent(ss.rx2, s->e_stage, 0);
    }
    const int erc = p25fe_shard_pipe_end(s->h, ss.rx2);
    return rc ? rc : erc;
}

int p25fe_shard_join(p25fe_shard_t* s, void* stream)
{
    if (!s) return P25FE_ERR_ARG;
    return p25fe_join_dev(s->h, stream);
}

int p25fe_shard_offsets(p25fe_shard_t* s, uint64_t* offsets)
{
    if (!s || !offsets) return P25FE_ERR_ARG;
    HCHK(hipSetDevice(p25fe_device(s->h)));
    HCHK(hipMemcpy(offsets, s->d_off, ((size_t)s->world + 1) * 8, hipMemcpyDeviceToHost));
    if (const int hrc = p25fe_shard_head_check(s->h)) { s->broken = true; return hrc; }     // a detection gave up waiting for the head: nothing of this step holds
    for (int r = 0; r < s->world; ++r)
        if (offsets[r + 1] - offsets[r] > s->cap) return P25FE_ERR_CAPACITY;     // the row was filled to the brim; the count is exact
    return P25FE_OK;
}

const uint8_t* p25fe_shard_stream_dev(const p25fe_shard_t* s) { return s ? s->d_stream : nullptr; }

int p25fe_shard_comm_ms(p25fe_shard_t* s, double ms[3], uint64_t* n_steps)
{
    if (!s || !ms) return P25FE_ERR_ARG;
    ms[0] = ms[1] = ms[2] = 0.0;
    HCHK(hipSetDevice(p25fe_device(s->h)));
    uint64_t from = s->read_from, cnt = 0;
    if (s->timed - from > (uint64_t)RING) from = s->timed - RING;
    if (s->comm && !s->staged) {
        for (uint64_t k = from; k < s->timed; ++k) {
            hipEvent_t* ev = s->ev[k % RING];
            for (int q = 0; q < 3; ++q) {
                float t = 0.f;
                if (hipEventSynchronize(ev[2 * q + 1]) == hipSuccess && hipEventElapsedTime(&t, ev[2 * q], ev[2 * q + 1]) == hipSuccess) ms[q] += t;
                else (void)hipGetLastError();                        // (a phase this step did not run, e.g. no gather)
            }
            ++cnt;
        }
    }
    for (int q = 0; q < 3; ++q) ms[q] = cnt ? ms[q] / (double)cnt : 0.0;
    if (n_steps) *n_steps = cnt;
    s->read_from = s->timed;
    return P25FE_OK;
}

}  // extern "C"
