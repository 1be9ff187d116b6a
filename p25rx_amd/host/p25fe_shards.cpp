// p25fe_shards -- one process per GPU over a time-sharded capture (BASELINE.json config 5), the launcher a Rust host would
// replace: forks RANKS children BEFORE anything touches the GPU; each child owns a contiguous range of a cf32 capture file,
// drives include/p25fe_rccl.h (halo by ncclSend / ncclRecv behind K1, summaries by ncclAllGather, device resolve, dibit
// rows to rank 0 + compaction) and rank 0 writes the ORDERED dibit stream -- byte for byte what `p25fe_replay cf32` writes.
//
//   p25fe_shards [-n RANKS] [-k STEPS] [-c 0|1] [-g exact|rows] [-t EVERY] [-p] [-s] [--shm] <in.cf32> <dibits.out>
//
//   -g      how the dibits reach rank 0 (include/p25fe_rccl.h): rows = whole rows + a compaction pass (default;
//           P25FE_GATHER_ROOT, no host wait in the step), exact = each shard's valid bytes, received at their offsets
//           (P25FE_GATHER_ROOT_EXACT: one host wait per step).  The JSON line reports the mode that RAN.
//   -t      HIP events around the exchanges on every EVERY-th step (p25fe_shard_comm_timing; 0 = never, default: the library's 16)
//   -c 1    the tracking symbol clock (docs/SPEC.md 3.8b; p25fe_config_t.symbol_clock)
//   --shm   TEST HOOK: all ranks on GPU 0, exchanges through a shared-memory segment instead of RCCL (a 1-GPU box)
//   -n 1    runs the same step through a ONE-rank RCCL communicator (self send / recv, all-gather of one)
// Prints one JSON line (rank 0): dibits, steps, ms per step, ms per step in each exchange.
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "p25fe_rccl.h"

static void die(const char* what, int rc)
{
    std::fprintf(stderr, "p25fe_shards: %s (%s, status %d)\n", what, p25fe_strerror(rc), rc);
    std::_Exit(1);
}

static int child(int rank, int world, int steps, bool shm, int clock, int gather, int timing, bool pipelined, bool own_stream, const char* in_path, const char* out_path, const std::string& key)
{
    FILE* f = std::fopen(in_path, "rb");
    if (!f) { std::fprintf(stderr, "unable to open %s\n", in_path); return 1; }
    struct stat sb;
    stat(in_path, &sb);
    const size_t total = (size_t)sb.st_size / 8;
    const size_t n = total / (size_t)world / 8 * 8;                  // shard cut points stay 16-byte aligned
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no HIP device\n"); return 1; }
    const int dev = shm ? 0 : rank % ndev;
    p25fe_config_t cfg;
    p25fe_default_config(&cfg);
    cfg.device = dev;
    cfg.symbol_clock = clock;
    p25fe_t* h = nullptr;
    int rc = p25fe_create(&cfg, &h);
    if (rc) die("unable to create the handle", rc);
    unsigned char id[P25FE_RCCL_ID_BYTES];
    const std::string idfile = std::string(out_path) + ".id";
    if (!shm) {                                                     // bootstrap: rank 0's id travels through a file
        if (rank == 0) {
            rc = p25fe_rccl_unique_id(id);
            if (rc) die("ncclGetUniqueId", rc);
            FILE* g = std::fopen((idfile + ".tmp").c_str(), "wb");
            std::fwrite(id, 1, sizeof id, g);
            std::fclose(g);
            std::rename((idfile + ".tmp").c_str(), idfile.c_str());
        } else {
            FILE* g = nullptr;
            for (int t = 0; t < 60000 && !(g = std::fopen(idfile.c_str(), "rb")); ++t) usleep(1000);
            if (!g || std::fread(id, 1, sizeof id, g) != sizeof id) { std::fprintf(stderr, "no communicator id\n"); return 1; }
            std::fclose(g);
        }
    } else {
        setenv("P25FE_SHARD_SHM", key.c_str(), 1);
    }
    p25fe_shard_t* s = nullptr;
    rc = p25fe_shard_create(h, rank, world, shm ? nullptr : id, n, &s);
    if (rc) die("unable to create the shard", rc);
    if (timing >= 0 && (rc = p25fe_shard_comm_timing(s, timing)) != 0) die("comm timing", rc);
    const size_t halo = p25fe_shard_halo(), cap = p25fe_shard_dibit_cap(s);
    // resident capture of this rank: [halo | owned]
    std::vector<float> host(2 * n);
    std::fseek(f, (long)((size_t)rank * n * 8), SEEK_SET);
    if (std::fread(host.data(), 8, n, f) != n) { std::fprintf(stderr, "short read\n"); return 1; }
    std::fclose(f);
    float* d_buf = nullptr;
    uint8_t* d_dib = nullptr;
    p25fe_result_t* d_res = nullptr;
    // The step goes to the NULL stream unless -s asks for a stream of this program's own.  (HIP multiplexes streams onto a few hardware
    // queues; before the library checked for it, one more user stream in the process put the step's side stream on K1's queue: 0.36 ms
    // instead of 0.28 pipelined / 0.355 instead of 0.337 plain -- profiles/r05_shard_pipelined.txt.  Now -s costs what the NULL stream costs.)
    hipStream_t st = nullptr;
    if (hipMalloc(&d_buf, (halo + n) * 8) != hipSuccess || hipMalloc(&d_dib, cap) != hipSuccess || hipMalloc(&d_res, sizeof *d_res) != hipSuccess ||
        (own_stream && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess))
        die("device buffers", P25FE_ERR_NOMEM);
    (void)hipMemset(d_buf, 0, halo * 8);
    (void)hipMemcpy(d_buf + 2 * halo, host.data(), n * 8, hipMemcpyHostToDevice);
    // once per stream the steps are given: the side stream must not share a hardware queue with it (no step checks: a step never blocks)
    if ((rc = p25fe_shard_prepare(s, st)) != 0) die("prepare", rc);
    // -p: p25fe_shard_step_pipelined -- only K1 on `st`, the rest of a step behind the next step's K1
    auto step = pipelined ? p25fe_shard_step_pipelined : p25fe_shard_step;
    for (int k = 0; k < 2; ++k) {                                     // warm-up (communicator set-up, scratch allocation)
        rc = step(s, d_buf, P25FE_FMT_CF32, d_dib, d_res, gather, st);
        if (rc) die("step", rc);
    }
    if ((rc = p25fe_shard_join(s, st)) != 0) die("join", rc);
    (void)hipStreamSynchronize(st);
    double cms[3];
    uint64_t cn = 0;
    (void)p25fe_shard_comm_ms(s, cms, &cn);
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; ++k) {
        rc = step(s, d_buf, P25FE_FMT_CF32, d_dib, d_res, gather, st);
        if (rc) die("step", rc);
    }
    if ((rc = p25fe_shard_join(s, st)) != 0) die("join", rc);
    const double enq_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;   // host time to ENQUEUE a step
    (void)hipStreamSynchronize(st);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
    (void)p25fe_shard_comm_ms(s, cms, &cn);
    std::vector<uint64_t> off((size_t)world + 1);
    rc = p25fe_shard_offsets(s, off.data());
    if (rc) die("offsets", rc);
    const int ran = p25fe_shard_gather_ran(s);
    p25fe_shard_info_t inf;
    if ((rc = p25fe_shard_info(s, &inf)) != 0) die("info", rc);
    // every rank says where it sits (the library's own evidence, not the launcher's arguments)
    std::fprintf(stderr, "{\"rank\":%d,\"rccl_ranks\":%d,\"rccl_rank\":%d,\"device\":%d,\"pci_bus_id\":\"%s\",\"comms\":%d,\"pipe_layout\":%d}\n",
                 inf.rank, inf.rccl_ranks, inf.rccl_rank, inf.device, inf.pci_bus_id, inf.comms, inf.pipe_layout);
    if (rank == 0) {
        std::vector<uint8_t> stream((size_t)off[(size_t)world]);
        (void)hipMemcpy(stream.data(), p25fe_shard_stream_dev(s), stream.size(), hipMemcpyDeviceToHost);
        FILE* g = std::fopen(out_path, "wb");
        std::fwrite(stream.data(), 1, stream.size(), g);
        std::fclose(g);
        std::printf("{\"ranks\":%d,\"samples_per_rank\":%zu,\"dibits\":%" PRIu64 ",\"steps\":%d,\"ms_per_step\":%.4f,\"host_enqueue_ms_per_step\":%.4f,"
                    "\"comm_ms_per_step\":{\"halo\":%.4f,\"summaries\":%.4f,\"dibit_gather\":%.4f,\"steps_averaged\":%" PRIu64 "},"
                    "\"exchange\":\"%s\",\"gather\":\"%s\",\"pipelined\":%s,\"rccl_ranks\":%d,\"comms\":%d,\"pipe_layout\":%d,\"pci_bus_id\":\"%s\"}\n",
                    world, n, off[(size_t)world], steps, ms, enq_ms, cms[0], cms[1], cms[2], cn, shm ? "TEST HOOK: shared memory, one GPU" : "RCCL",
                    ran == P25FE_GATHER_ROOT_EXACT ? "exact" : (ran == P25FE_GATHER_ROOT ? "rows" : "other"), pipelined ? "true" : "false",
                    inf.rccl_ranks, inf.comms, inf.pipe_layout, inf.pci_bus_id);
        std::remove(idfile.c_str());
        std::fflush(stdout);                                          // the child leaves through _Exit
    }
    p25fe_shard_destroy(s);
    p25fe_destroy(h);
    return 0;
}

int main(int argc, char** argv)
{
    int ranks = 1, steps = 3, clock = 0, a = 1, gather = P25FE_GATHER_ROOT, timing = -1;
    bool shm = false, pipelined = false, own_stream = false;
    for (; a < argc && argv[a][0] == '-'; ++a) {
        if (!std::strcmp(argv[a], "-n") && a + 1 < argc) ranks = std::atoi(argv[++a]);
        else if (!std::strcmp(argv[a], "-k") && a + 1 < argc) steps = std::atoi(argv[++a]);
        else if (!std::strcmp(argv[a], "-c") && a + 1 < argc) clock = std::atoi(argv[++a]);
        else if (!std::strcmp(argv[a], "-g") && a + 1 < argc) gather = !std::strcmp(argv[++a], "exact") ? P25FE_GATHER_ROOT_EXACT : P25FE_GATHER_ROOT;
        else if (!std::strcmp(argv[a], "-t") && a + 1 < argc) timing = std::atoi(argv[++a]);
        else if (!std::strcmp(argv[a], "--shm")) shm = true;
        else if (!std::strcmp(argv[a], "-p")) pipelined = true;
        else if (!std::strcmp(argv[a], "-s")) own_stream = true;
        else break;
    }
    if (argc - a != 2 || ranks < 1 || steps < 1 || (clock != 0 && clock != 1)) {
        std::fprintf(stderr, "usage: %s [-n RANKS] [-k STEPS] [-c 0|1] [-g exact|rows] [-t EVERY] [-p] [-s] [--shm] <in.cf32> <dibits.out>\n", argv[0]);
        return 2;
    }
    const std::string key = "/p25fe_shards_" + std::to_string((long)getpid());
    std::remove((std::string(argv[a + 1]) + ".id").c_str());
    // one process per rank, forked before the GPU is touched (a process that has initialised HIP must not fork workers)
    std::vector<pid_t> kids;
    for (int r = 0; r < ranks; ++r) {
        const pid_t p = fork();
        if (p == 0) std::_Exit(child(r, ranks, steps, shm, clock, gather, timing, pipelined, own_stream, argv[a], argv[a + 1], key));
        kids.push_back(p);
    }
    int bad = 0;
    for (pid_t p : kids) {
        int stt = 0;
        waitpid(p, &stt, 0);
        if (!WIFEXITED(stt) || WEXITSTATUS(stt) != 0) bad = 1;
    }
    if (shm) {
        const std::string path = "/dev/shm" + key;
        std::remove(path.c_str());
    }
    return bad;
}
