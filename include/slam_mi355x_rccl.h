/*
 * slam_mi355x_rccl.h -- the one exchange step of the multi-GPU path: the
 * per-GPU int32 hit/miss planes are summed over all ranks with a single RCCL
 * all-reduce over xGMI.  Integer sums are order
 * independent, so the merged counts are bit-identical to a single-GPU run
 * over the union of the scans.  The reference has no counterpart (it is a
 * single-process ROS node); a multi-GPU mapper calls this between
 * slam_grid_raycast_*() and slam_grid_finalize().
 *
 * Lives in its own shared object (libslam_mi355x_rccl.so) so that the core
 * library does not depend on RCCL.  One process per GPU; the communicator is
 * either created here from an ncclUniqueId the host exchanges out of band
 * (128 bytes from rank 0 to every rank), or adopted from an existing
 * ncclComm_t.
 */
#ifndef SLAM_MI355X_RCCL_H
#define SLAM_MI355X_RCCL_H

#include "slam_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slam_comm slam_comm_t;

#define SLAM_COMM_ID_BYTES 128

int  slam_comm_unique_id(char id[SLAM_COMM_ID_BYTES]);                       /* rank 0: ncclGetUniqueId */
int  slam_comm_create(const char id[SLAM_COMM_ID_BYTES], int rank, int n_ranks, slam_comm_t **out);
int  slam_comm_adopt(void *nccl_comm, slam_comm_t **out);                     /* wrap an ncclComm_t (not owned).  After SLAM_E_TIMEOUT /
                                                                                 * SLAM_E_COMM every entry point on it returns that code and
                                                                                 * enqueues nothing more; ncclCommAbort is its OWNER's call */
/* A communicator whose collectives go through the HOST: for ranks RCCL cannot connect (several processes on ONE GPU:
 * rehearsals of the N > 1 path on a one-GPU box) or for a transport of the caller's own (MPI, gloo).  Every entry point
 * below works the same way on it; the library stages the rows through pinned memory and calls `allreduce` on host
 * buffers of int32 (in place, over all ranks, blocking; op = SLAM_COMM_SUM or SLAM_COMM_MIN; returns 0 or non-zero
 * on failure).  slam_grid_merge_finish and the all-reduces synchronise `stream` on such a communicator. */
#define SLAM_COMM_SUM 0
#define SLAM_COMM_MIN 1
typedef int (*slam_host_allreduce_fn)(void *ctx, int32_t *buf, size_t count, int op);
int  slam_comm_create_host(int rank, int n_ranks, slam_host_allreduce_fn allreduce, void *ctx, slam_comm_t **out);
void slam_comm_destroy(slam_comm_t *comm);
int  slam_comm_info(slam_comm_t *comm, int *rank, int *n_ranks);
/* What the merges over this communicator cost since it was made (or since the last slam_comm_stats_reset): what a
 * multi-GPU run reports so that its scaling can be read (bench.py N > 1: `merge` in the JSON line). */
typedef struct slam_comm_stats {
    int       rank, n_ranks;   /* as the transport itself counts them (ncclCommUserRank / ncclCommCount) */
    int       transport;       /* 0 = RCCL, 1 = host-staged */
    int       rccl_version;    /* ncclGetVersion, e.g. 22203; 0 for the host-staged transport */
    long long merges;          /* slam_grid_merge_finish calls that merged */
    long long rows;            /* storage rows they summed, in total */
    long long bytes;           /* payload of those all-reduces per rank: 2 planes x 4 B x the cells of those rows */
    double    wait_ms;         /* host time spent inside slam_grid_merge_finish waiting for the united row range */
    double    allreduce_ms;    /* device time of the row all-reduces of the last `timed` merges (HIP events around them on */
    long long timed;           /* their stream: includes the time their kernels waited for a CU) */
    double    helper_wait_ms;  /* slam_grid_merge_async: the helper thread's waits for united ranges (wait_ms stays the CALLER's */
    long long async_merges;    /* thread: its waits in _finish, _ticket_wait and for a place in the helper's queue); merges it ran */
} slam_comm_stats;
int  slam_comm_get_stats(slam_comm_t *comm, slam_comm_stats *out); /* waits for the timed all-reduces to complete */
int  slam_comm_stats_reset(slam_comm_t *comm);

/* ncclAllReduce(planes, planes, 2*size_x*size_y, ncclInt32, ncclSum) on `stream` */
int  slam_grid_allreduce(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream);
/* the same for rows [row_lo, row_hi] of both planes only (one grouped pair of all-reduces) */
int  slam_grid_allreduce_rows(slam_grid_t *grid, slam_comm_t *comm, int row_lo, int row_hi, slam_stream_t stream);
/* Merge of the rows ANY rank touched, without the host knowing them in advance:
 *   begin : on `stream`, the ranks' device-tracked dirty ranges (slam_grid_dirty_rows_dev) are united with
 *           one 24-byte all-reduce and start travelling to the host; returns at once -- enqueue other work
 *           (the next batch's registration) before calling finish;
 *   finish: waits for that range on the host, then enqueues the all-reduce of those rows on `stream`;
 *           *row_lo / *row_hi (optional) receive the range (row_hi < row_lo: nothing to merge).
 * Storage rows mean the same world cells on every rank only while the ranks' windows sit on the same cells: the same
 * all-reduce carries every rank's window position (in cells, MLS::setPose mls.cpp:419-431) and finish fails with
 * SLAM_E_INVALID, merging nothing, when rolling grids were moved apart.
 * Up to four merges may be in flight on one communicator (a pipelined caller begins the merge of the next batch, into
 * another grid, before it finishes this one's); they finish in the order they were begun, every rank the same. */
int  slam_grid_merge_begin(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream);
int  slam_grid_merge_finish(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream, int *row_lo, int *row_hi);
/* The same merge with NO host wait on the caller's thread: begin, the wait for the united range, the rows' all-reduce and
 * what follows it are issued on `stream` by the communicator's helper thread (started on first use), whole merges one after
 * the other in the order posted -- the ranks post their merges in the same order, so their collectives meet in the same
 * order, and one thread at a time talks to RCCL.  Returns at once with a ticket.
 *   then : what the helper enqueues behind the rows' sum on `stream`:
 *          SLAM_MERGE_THEN_FINALIZE_RESET = slam_grid_finalize_reset (a batch per step: evidence + occupancy, counts zero),
 *          SLAM_MERGE_THEN_FOLD_FINALIZE  = slam_grid_fold of the united rows + slam_grid_finalize (a running map),
 *          SLAM_MERGE_THEN_NOTHING;
 *   done : (optional) an event recorded on `stream` behind all of that.  Other streams may wait for it only AFTER
 *          slam_comm_ticket_wait(ticket) has returned (an event not yet recorded waits for nothing).
 * Contract: what the merge covers is fixed when this call returns (the grid updates enqueued on `stream` so far); from
 * then until slam_comm_ticket_wait(ticket) returns, the caller enqueues nothing on `stream` and makes no call on `grid`
 * -- normally a wait that is long over when it is asked for (two grids in turn: the ticket of two steps ago).
 * slam_comm_ticket_wait returns the merge's status and united row range; slam_comm_drain waits for every merge posted
 * (before a device synchronisation that is meant to cover them).  The begin / finish pair above stays for callers that
 * want the range on their own thread; it first waits for the helper's queue to empty. */
#define SLAM_MERGE_THEN_NOTHING        0
#define SLAM_MERGE_THEN_FINALIZE_RESET 1
#define SLAM_MERGE_THEN_FOLD_FINALIZE  2
int  slam_grid_merge_async(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream, int then, slam_event_t done,
                           unsigned long long *ticket);
int  slam_comm_ticket_wait(slam_comm_t *comm, unsigned long long ticket, int *row_lo, int *row_hi);
int  slam_comm_drain(slam_comm_t *comm);
/* Fail fast.  A rank that dies leaves the others inside a collective that never completes; nothing in RCCL ends that within
 * a node (the peers' flags simply never change).  So every host wait of a merge is a POLL of its event: with the
 * communicator's health (ncclCommGetAsyncError) asked every millisecond, and a time-out (default 60 s) after which the
 * merge fails with SLAM_E_TIMEOUT.  On SLAM_E_COMM / SLAM_E_TIMEOUT the communicator is aborted (ncclCommAbort: this rank's
 * own enqueued collectives end, its streams drain) and every later call on it returns the same code; slam_last_error() names
 * the rank that noticed.  slam_comm_check asks the health without waiting for anything. */
int  slam_comm_set_timeout(slam_comm_t *comm, double seconds);
int  slam_comm_check(slam_comm_t *comm);
/* the streaming mapper's periodic merge (slam_mapper_params::merge_every) over this communicator */
int  slam_mapper_use_comm(slam_mapper_t *mapper, slam_comm_t *comm);

#ifdef __cplusplus
}
#endif
#endif
