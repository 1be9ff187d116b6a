/*
 * p25fe_rccl.h -- the N > 1 step of a time-sharded capture behind the C ABI (BASELINE.json config 5: "Long IQ captures
 * ... shard across the 8 GPUs of one node with RCCL over xGMI carrying only filter-state overlap samples and the reduced
 * dibit stream").  libp25fe_rccl.so = libp25fe.so + librccl: a Rust (or C++) host drives one rank per GPU with these
 * calls and needs neither Python nor MPI.
 *
 * The reference has no distributed code: what the ranks hand to each other is the serial state
 * DemodTask / MessageReceiver carry from chunk to chunk (src/demod.rs:25-40, the receiver's lock), cut at shard boundaries.
 *
 *   rank r owns samples [r * n_per_rank, (r + 1) * n_per_rank) of ONE capture, resident in HBM as [halo | owned];
 *   one step (all enqueued; no host synchronisation unless P25FE_GATHER_ROOT_EXACT is asked for):
 *     1. on a side stream: the last p25fe_shard_halo() samples go to rank r + 1 (ncclSend / ncclRecv), then the shard's head
 *        segment (p25fe_shard_pass1_head) -- WHILE K1 runs over everything that does not touch the halo on the caller's
 *        stream (p25fe_shard_pass1_main); then sync detection + scan (p25fe_shard_pass1_finish);
 *     2. ncclAllGather of one p25fe_result_t per rank;
 *     3. pass 2 with the combine inside the slicer (p25fe_shard_pass2_dev: carry-in anchor and dibit offsets from the
 *        gathered summaries, no second scan, no resolve launch); rank 0 slices straight into the ordered stream as well;
 *     4. the other shards' dibit rows go to rank 0 point-to-point (xGMI is a full mesh) and are compacted there behind
 *        rank 0's own -- ONE ordered stream, what RecvTask feeds into MessageReceiver (src/recv.rs:148-150).
 *   Dependent launches behind K1 on rank 0: detection, scan, all-gather, slicer, receive, compaction.
 *   p25fe_shard_step_pipelined runs the same passes with the exchanges placed at the boundaries between consecutive steps' K1
 *   launches (the halo on a communicator of its own directly in front of K1; 2 and 3 beside the next step's K1, 4 beside the one
 *   after): see its declaration below.
 *
 * Bootstrap: rank 0 obtains a 128-byte id (p25fe_rccl_unique_id) and gives it to the other ranks by any means (file,
 * pipe, socket); every rank calls p25fe_shard_create with it.
 */
#ifndef P25FE_RCCL_H
#define P25FE_RCCL_H

#include "p25fe.h"

#ifdef __cplusplus
extern "C" {
#endif

#define P25FE_RCCL_ID_BYTES 128

typedef struct p25fe_shard p25fe_shard_t;

/* How the shards' dibits reach the consumer:
 *   ROOT        (the default of every driver in this repository) every rank sends its whole row (p25fe_shard_dibit_cap() bytes:
 *               the valid length + <= 0.02 % + 64 bytes of slack) to rank 0, which compacts the rows into the ordered stream; no
 *               host synchronisation anywhere in the step;
 *   ROOT_EXACT  every rank sends exactly its offsets[r + 1] - offsets[r] dibits, rank 0 receives them AT offsets[r] of the
 *               ordered stream (no padding on the wire, no compaction kernel).  CONTRACT CHANGE against ROOT: the counts are
 *               host arguments of ncclSend / ncclRecv, so p25fe_shard_step BLOCKS the calling thread once per step until the
 *               world + 1 offsets are on the host (a one-thread resolve + a 72-byte copy on the side stream, ready before
 *               pass 2 ends) -- and a rank that fails before that point leaves its peers waiting in their receives;
 *   ALL         all-gather of the rows, every rank compacts (diagnostic).
 * The one-rank communicator of the tests plays its own peer: it sends its row / its exact bytes to itself. */
enum { P25FE_GATHER_NONE = 0, P25FE_GATHER_ROOT = 1, P25FE_GATHER_ALL = 2, P25FE_GATHER_ROOT_EXACT = 3 };

/* rank 0: a fresh communicator id (ncclGetUniqueId) */
int p25fe_rccl_unique_id(void *id128);

/* h: a ONE-channel handle on this rank's GPU (it must outlive the shard object).  n_per_rank: owned samples per rank, a
 * multiple of 8.  id128: the id of p25fe_rccl_unique_id.
 * TEST HOOK: id128 == NULL selects a host-staged exchange through the POSIX shared-memory object named by the environment
 * variable P25FE_SHARD_SHM (same value in every rank), so that several ranks can share ONE GPU in tests; never the
 * product path. */
int p25fe_shard_create(p25fe_t *h, int rank, int world, const void *id128, size_t n_per_rank, p25fe_shard_t **out);
void p25fe_shard_destroy(p25fe_shard_t *s);

/* What the shard object itself knows about the job it is part of -- the evidence a record of an N > 1 run needs: did RCCL see N ranks, on
 * which GPU does this rank sit, which layout does the pipelined step run.  Every field is read from the library's own state (the
 * communicator's ncclCommCount / ncclCommUserRank, hipDeviceGetPCIBusId of the handle's device), none from the launcher's environment. */
typedef struct p25fe_shard_info {
    int32_t rank, world;                 /* as given to p25fe_shard_create */
    int32_t rccl_ranks, rccl_rank;       /* ncclCommCount / ncclCommUserRank of the communicator; 0 / -1: no communicator (world 1 without an id, the test hook) */
    int32_t device;                      /* HIP device ordinal of the handle */
    int32_t comms;                       /* communicators the steps use: 0 none; 1 one for everything; 3 halo / summaries / dibit rows each their own */
    int32_t pipe_layout;                 /* p25fe_shard_step_pipelined as AGREED by all ranks in p25fe_shard_create: 2 = two stages placed at K1
                                            boundaries (needs comms == 3 or 0), 1 = the step's own order on the receive stream */
    int32_t gather_ran;                  /* P25FE_GATHER_* of the last step */
    int32_t staged;                      /* 1: the shared-memory test hook, not RCCL */
    int32_t head_wait;                   /* plain step, how the detection waits for the head segment: 0 flag word (bounded spin), 1 stream event */
    int32_t broken;                      /* 1: a collective failed half-way; every later step fails */
    int32_t reserved;
    uint64_t steps;                      /* steps enqueued so far */
    char pci_bus_id[32];                 /* "0000:05:00.0" */
} p25fe_shard_info_t;
int p25fe_shard_info(const p25fe_shard_t *s, p25fe_shard_info_t *out);

/* Once per stream the steps will be given (optional, recommended for p25fe_shard_step_pipelined): makes sure the shard's side stream does not
 * share a HARDWARE queue with `stream` or with the handle's receive stream (p25fe_streams_share_queue; a side stream that does is
 * replaced).  SYNCHRONISES those streams (< 1 ms each) -- which is why it is a call of its own and no step does it: a step never
 * blocks the host (ROOT gather) and may be given a capturing stream.  p25fe_shard_create has already checked the side stream against
 * the handle's receive stream. */
int p25fe_shard_prepare(p25fe_shard_t *s, void *stream);

/* bytes of a per-rank dibit row (the gather granule): n / 50 plus proportional slack for the transmitter's symbol clock */
size_t p25fe_shard_dibit_cap(const p25fe_shard_t *s);

/* One pass of the hot path over this rank's shard.  d_buf: device buffer [p25fe_shard_halo() | n_per_rank] samples of
 * `fmt` (the halo part is overwritten by the neighbour's samples); d_dibits: device row of p25fe_shard_dibit_cap() bytes;
 * d_result: device record (after the step: this shard's final summary).  gather: P25FE_GATHER_*. */
int p25fe_shard_step(p25fe_shard_t *s, void *d_buf, int fmt, uint8_t *d_dibits, p25fe_result_t *d_result, int gather,
                     void *stream);

/* The same step for a host that feeds one capture after the other (p25fe_run_dev_pipelined's contract, applied to shards): only
 * K1's main launch is enqueued on `stream`; the halo / head run on the side stream as before and EVERYTHING behind K1 (sync
 * detection, scan, summary all-gather, slicer, dibit gather, compaction) on the handle's receive stream, so that the next call's
 * K1 starts while this call's ~80 us chain of short kernels and exchanges is still running.  Outputs (d_dibits, d_result, the
 * ordered stream, the offsets) are complete once a stream has been made to wait with p25fe_shard_join (or the device has been
 * synchronised).  Consecutive calls may name the same output buffers (d_dibits and d_result are written by pass 2 only, and the pass 2s
 * run in call order on one stream; pass 1's summaries live in the shard object, one slot per step in flight); the ordered
 * stream alternates between two buffers, p25fe_shard_stream_dev() names the LAST call's, and the previous call's stays untouched
 * until the call after this one; d_buf must stay unchanged until `stream` and the step's exchanges have passed it (join).
 * P25FE_GATHER_ROOT_EXACT keeps its one host wait per step, which ends the overlap for that mode; the shared-memory test hook runs
 * the plain step.  The halo and the summaries travel on communicators of their own (ncclCommSplit: RCCL serialises the operations of one
 * communicator across streams); without them the step keeps everything behind K1 in step order on the receive stream.
 * p25fe_shard_create makes two more communicators for this step (halo, summaries) and AGREES with the other ranks on whether all of them
 * have both (p25fe_shard_info.pipe_layout / .comms): if any rank lacks one, every rank runs the one-communicator layout.
 * STREAM: any; call p25fe_shard_prepare(s, stream) once before the first step on it ("Which stream to pass"). */
int p25fe_shard_step_pipelined(p25fe_shard_t *s, void *d_buf, int fmt, uint8_t *d_dibits, p25fe_result_t *d_result, int gather,
                               void *stream);
/* make `stream` wait for everything p25fe_shard_step_pipelined has enqueued so far */
int p25fe_shard_join(p25fe_shard_t *s, void *stream);

/* HIP events around the three exchanges (p25fe_shard_comm_ms) ride on every `every`-th step only (default 16; 0: never;
 * 1: every step): the four around the all-gather and the gather are packets between the kernels of the step's critical
 * path, ~4 us each. */
int p25fe_shard_comm_timing(p25fe_shard_t *s, int every);

/* P25FE_GATHER_* as the LAST step executed it (a shard without a communicator -- world 1, id128 NULL and no test hook --
 * has nothing to gather: ROOT). */
int p25fe_shard_gather_ran(const p25fe_shard_t *s);

/* After `stream` has been synchronised: the world + 1 dibit offsets of the capture (shard r holds
 * [offsets[r], offsets[r + 1]); returns P25FE_ERR_CAPACITY if a shard outgrew its row), and the device pointer of the
 * ordered stream (rank 0, or every rank with P25FE_GATHER_ALL; offsets[world] bytes). */
int p25fe_shard_offsets(p25fe_shard_t *s, uint64_t *offsets);
const uint8_t *p25fe_shard_stream_dev(const p25fe_shard_t *s);

/* Average milliseconds per TIMED step (p25fe_shard_comm_timing) spent in the three exchanges since the last call (HIP events
 * around them on their streams): ms[0] halo send / recv, ms[1] summary all-gather, ms[2] dibit gather; *n_steps = steps
 * averaged (at most the last 64 timed ones).  Reading synchronises those events. */
int p25fe_shard_comm_ms(p25fe_shard_t *s, double ms[3], uint64_t *n_steps);

#ifdef __cplusplus
}
#endif
#endif /* P25FE_RCCL_H */
