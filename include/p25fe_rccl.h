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
 * the plain step.  p25fe_shard_create makes two more communicators, for the halo and for the summaries (ncclCommSplit, a collective every
 * rank takes part in: RCCL serialises the operations of one communicator across streams); if that fails the step keeps everything behind
 * K1 in step order on the receive stream.
 * STREAM: any.  The first step that sees a caller's stream checks that the side stream does not share its hardware queue
 * (p25fe_streams_share_queue: synchronises both streams once) and replaces it if it does ("Which stream to pass"). */
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
