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
 *   one step (all enqueued, no host synchronisation):
 *     1. the last p25fe_shard_halo() samples go to rank r + 1 (ncclSend / ncclRecv on a side stream) WHILE K1 runs over
 *        everything that does not touch the halo (p25fe_shard_pass1_main); then the head + sync detection + scan;
 *     2. ncclAllGather of one p25fe_result_t per rank; p25fe_shard_resolve_dev: carry-in anchors, dibit offsets;
 *     3. pass 2 (scan with the carry-in + slicer); the shards' dibit rows go to rank 0 point-to-point (xGMI is a full mesh)
 *        and are compacted there into ONE ordered stream -- what RecvTask feeds into MessageReceiver (src/recv.rs:148-150).
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
 *   ROOT        every rank sends its whole row (p25fe_shard_dibit_cap() bytes: the valid length + <= 0.02 % + 64 bytes of slack)
 *               to rank 0, which compacts the rows into the ordered stream; no host synchronisation anywhere in the step;
 *   ROOT_EXACT  every rank sends exactly its offsets[r + 1] - offsets[r] dibits, rank 0 receives them AT offsets[r] of the
 *               ordered stream (no padding on the wire, no compaction kernel); the counts are host arguments of ncclSend /
 *               ncclRecv, so the step waits once for the world + 1 offsets (ready before pass 2 runs, which overlaps the wait);
 *   ALL         all-gather of the rows, every rank compacts (diagnostic). */
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

/* After `stream` has been synchronised: the world + 1 dibit offsets of the capture (shard r holds
 * [offsets[r], offsets[r + 1]); returns P25FE_ERR_CAPACITY if a shard outgrew its row), and the device pointer of the
 * ordered stream (rank 0, or every rank with P25FE_GATHER_ALL; offsets[world] bytes). */
int p25fe_shard_offsets(p25fe_shard_t *s, uint64_t *offsets);
const uint8_t *p25fe_shard_stream_dev(const p25fe_shard_t *s);

/* Average milliseconds per step spent in the three exchanges since the last call (HIP events around them on their
 * streams): ms[0] halo send / recv, ms[1] summary all-gather, ms[2] dibit gather; *n_steps = steps averaged.  Reading
 * synchronises those events. */
int p25fe_shard_comm_ms(p25fe_shard_t *s, double ms[3], uint64_t *n_steps);

#ifdef __cplusplus
}
#endif
#endif /* P25FE_RCCL_H */
