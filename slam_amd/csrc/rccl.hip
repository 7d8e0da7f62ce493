// rccl.hip -- RCCL merge of the grid count planes (slam_mi355x_rccl.h).
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>

#include "common.hpp"
#include "slam_mi355x_rccl.h"

using namespace slam;

// What the ranks agree on before rows move: {lowest dirty row, -(highest), window cell x, -x, window cell y, -y}.  One
// MIN all-reduce unites the ranges (the union of {lo, -hi} is their minimum) and shows whether the windows sit on the same
// cells (min(x) == -min(-x) exactly when every rank holds the same x).
constexpr int kKeyInts = 6;

struct slam_comm {
    ncclComm_t comm = nullptr;      // null for a host-staged communicator
    bool       owned = false;
    int        rank = 0, n_ranks = 1;
    slam_host_allreduce_fn host_fn = nullptr; // host-staged transport (slam_comm_create_host)
    void      *host_ctx = nullptr;
    // merges in flight, oldest first: begin takes the next slot, finish the oldest (a pipelined caller begins the merge of
    // batch k + 1 -- on another grid -- before it finishes that of batch k)
    static constexpr int kSlots = 4;   // the caller's own merges in flight (slam_grid_merge_begin / _finish)
    static constexpr int kMaxPosted = 8; // merges handed to the helper thread and not yet through
    struct Slot {
        int       *d_key = nullptr;   // [kKeyInts] this rank's key
        int       *d_range = nullptr; // [kKeyInts] the minimum over the ranks
        int       *h_range = nullptr; // pinned copy
        hipEvent_t ev_range = nullptr;
        slam_grid_t *grid = nullptr;  // the grid the merge was begun on
    } slot[1 + kSlots];               // [0]: the helper thread's
    unsigned long long begun = 0, finished = 0;
    // ---- slam_grid_merge_async: a helper thread takes the host's one wait of a merge off the caller's thread.
    // Jobs run in the order posted; a ticket's outcome stays readable while fewer than kDone newer ones have completed.
    struct Job {
        unsigned long long ticket;
        slam_grid_t   *grid;
        hipStream_t    st;
        int            then;
        hipEvent_t     done;
        const int32_t *d_dirty;
        int            cell_x, cell_y;
    };
    struct Outcome {
        unsigned long long ticket = 0;
        int  rc = SLAM_OK, lo = 0, hi = -1;
        char err[256] = "";
    };
    static constexpr int kDone = 16;
    std::thread             worker;
    std::mutex              mu;
    std::condition_variable cv_job, cv_done;
    std::deque<Job>         jobs;
    unsigned long long      posted = 0, completed = 0;
    long long               async_merges = 0;     // merges the helper thread has issued since slam_comm_stats_reset
    Outcome                 outcome[kDone];
    bool                    stop = false;
    int                     device = 0;
    std::atomic<int>        failed{SLAM_OK};      // sticky: the first failure of a merge (a peer gone, a time-out)
    std::atomic<bool>       aborted{false};       // ncclCommAbort has run: `comm` is gone (read without the lock by usable / comm_health)
    char                    failed_msg[256] = "";
    std::atomic<double>     timeout_s{60.0};      // a united range that has not arrived by then is a lost rank (read by the helper thread's polls)
    double                  helper_wait_ms = 0.0; // the helper thread's waits for united ranges
    int32_t   *h_stage = nullptr;   // pinned staging of the host-staged transport
    size_t     cap_stage = 0;       // ints
    // statistics (slam_comm_get_stats)
    static constexpr int kTimed = 64;
    long long  merges = 0, rows = 0, bytes = 0;
    double     wait_ms = 0.0;
    hipEvent_t ev_ar[kTimed][2] = {};
    long long  ar_recorded = 0;     // row all-reduces whose events were recorded (the ring keeps the last kTimed)
    bool       in_merge = false;    // slam_grid_allreduce_rows called by slam_grid_merge_finish: time it
};

namespace {

__global__ void merge_key_kernel(const int *dirty, int cell_x, int cell_y, int *key)
{
    if (threadIdx.x || blockIdx.x) return;
    key[0] = dirty[0];
    key[1] = dirty[1];
    key[2] = cell_x;
    key[3] = -cell_x;
    key[4] = cell_y;
    key[5] = -cell_y;
}

bool usable(const slam_comm *c) { return c && !c->aborted && (c->comm || c->host_fn); }

int stage_reserve(slam_comm *c, size_t ints)
{
    if (ints <= c->cap_stage) return SLAM_OK;
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    c->cap_stage = 0;
    SLAM_HIP(hipHostMalloc((void **)&c->h_stage, ints * sizeof(int32_t), hipHostMallocDefault));
    c->cap_stage = ints;
    return SLAM_OK;
}

// host-staged sum of `count` ints at planes + first and planes + cells + first: both parts down, one all-reduce of the
// two together, both back.  Synchronises the stream (the caller's transport blocks anyway).
int host_sum_rows(slam_comm *c, int32_t *planes, size_t cells, size_t first, size_t count, hipStream_t st)
{
    SLAM_TRY(stage_reserve(c, 2 * count));
    SLAM_HIP(hipMemcpyAsync(c->h_stage, planes + first, count * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipMemcpyAsync(c->h_stage + count, planes + cells + first, count * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipStreamSynchronize(st));
    SLAM_REQUIRE(c->host_fn(c->host_ctx, c->h_stage, 2 * count, SLAM_COMM_SUM) == 0, SLAM_E_HIP,
                 "the host transport's all-reduce (sum of %zu ints) failed", 2 * count);
    SLAM_HIP(hipMemcpyAsync(planes + first, c->h_stage, count * sizeof(int32_t), hipMemcpyHostToDevice, st));
    SLAM_HIP(hipMemcpyAsync(planes + cells + first, c->h_stage + count, count * sizeof(int32_t), hipMemcpyHostToDevice, st));
    SLAM_HIP(hipStreamSynchronize(st)); // the staging buffer is free again
    return SLAM_OK;
}

// What RCCL says about the communicator without blocking: SLAM_OK, or SLAM_E_COMM with the text (a failed peer or
// transport shows up here; a peer that simply stopped shows up as the time-out of wait_range).
int comm_health(slam_comm *c)
{
    if (c->failed.load() != SLAM_OK) {
        set_error("%s", c->failed_msg);
        return c->failed.load();
    }
    if (!c->comm || c->aborted) return SLAM_OK;
    ncclResult_t st = ncclSuccess;
    const ncclResult_t r = ncclCommGetAsyncError(c->comm, &st);
    if (r == ncclSuccess && (st == ncclSuccess || st == ncclInProgress)) return SLAM_OK;
    const ncclResult_t bad = r != ncclSuccess ? r : st;
    set_error("rank %d of %d: RCCL reports an asynchronous error on the communicator: %d (%s)", c->rank, c->n_ranks, (int)bad,
              ncclGetErrorString(bad));
    return SLAM_E_COMM;
}

void mark_failed(slam_comm *c, int rc)
{
    if (c->failed.load() != SLAM_OK) return;
    snprintf(c->failed_msg, sizeof c->failed_msg, "%s", slam_last_error());
    c->failed.store(rc);
    // whatever this rank still has enqueued on the communicator would wait for the lost rank for ever: abort it, so that the
    // streams drain and the process can report and exit (the communicator is not usable after this: every later call on it
    // returns the same failure)
    if (c->comm && c->owned && !c->aborted) {
        c->aborted = true;
        (void)ncclCommAbort(c->comm);
    }
    // (an ADOPTED communicator is its owner's to abort: this library enqueues nothing more on it -- every entry point, the plain
    // all-reduces included, returns the failure from here on -- and slam_comm_check tells the owner, who calls ncclCommAbort)
}

// The host's one wait of a merge: the united range's event, polled -- so that a lost rank ends in an error code and not in a
// process asleep inside hipEventSynchronize.  Every millisecond the communicator's health is asked as well.
int wait_range(slam_comm *c, hipEvent_t ev, double *waited_ms)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto       next_health = t0 + std::chrono::milliseconds(1);
    int        rc = SLAM_OK;
    for (unsigned spin = 0;; ++spin) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) {
            rc = hip_fail(e, "hipEventQuery(united range)", __FILE__, __LINE__);
            break;
        }
        const auto now = clk::now();
        if (now >= next_health) {
            if ((rc = comm_health(c)) != SLAM_OK) break;
            next_health = now + std::chrono::milliseconds(1);
            const double s = std::chrono::duration<double>(now - t0).count();
            if (s > c->timeout_s.load()) {
                set_error("rank %d of %d: the united row range of a merge has not arrived after %.1f s: another rank has stopped "
                          "(or never began this merge)", c->rank, c->n_ranks, s);
                rc = SLAM_E_TIMEOUT;
                break;
            }
        }
        if (spin < 2000)
            std::this_thread::yield();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    if (waited_ms) *waited_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    return rc;
}

// every job posted so far has run (the caller's thread waits; a job's own wait is bounded by the time-out)
void drain(slam_comm *c)
{
    std::unique_lock<std::mutex> lk(c->mu);
    c->cv_done.wait(lk, [&] { return c->completed == c->posted; });
}

} // namespace

#define SLAM_NCCL(expr)                                                                   \
    do {                                                                                  \
        ncclResult_t r__ = (expr);                                                        \
        if (r__ != ncclSuccess) {                                                         \
            set_error("RCCL error %d (%s) in %s", (int)r__, ncclGetErrorString(r__), #expr); \
            return SLAM_E_HIP;                                                            \
        }                                                                                 \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= SLAM_COMM_ID_BYTES, "ncclUniqueId must fit the ABI buffer");

extern "C" {

int slam_comm_unique_id(char id[SLAM_COMM_ID_BYTES])
{
    SLAM_REQUIRE(id, SLAM_E_INVALID, "slam_comm_unique_id: null buffer");
    ncclUniqueId u;
    SLAM_NCCL(ncclGetUniqueId(&u));
    memset(id, 0, SLAM_COMM_ID_BYTES);
    memcpy(id, &u, sizeof u);
    return SLAM_OK;
}

int slam_comm_create(const char id[SLAM_COMM_ID_BYTES], int rank, int n_ranks, slam_comm_t **out)
{
    SLAM_REQUIRE(id && out && n_ranks >= 1 && rank >= 0 && rank < n_ranks, SLAM_E_INVALID,
                 "slam_comm_create: bad arguments");
    *out = nullptr;
    SLAM_TRY(require_device());
    slam_comm *c = new (std::nothrow) slam_comm();
    SLAM_REQUIRE(c, SLAM_E_NOMEM, "slam_comm_create: out of host memory");
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        set_error("RCCL error %d (%s) in ncclCommInitRank", (int)r, ncclGetErrorString(r));
        delete c;
        return SLAM_E_HIP;
    }
    c->owned = true;
    c->rank = rank;
    c->n_ranks = n_ranks;
    (void)hipGetDevice(&c->device); // the helper thread binds to the device the communicator was made on
    *out = c;
    return SLAM_OK;
}

int slam_comm_create_host(int rank, int n_ranks, slam_host_allreduce_fn allreduce, void *ctx, slam_comm_t **out)
{
    SLAM_REQUIRE(allreduce && out && n_ranks >= 1 && rank >= 0 && rank < n_ranks, SLAM_E_INVALID,
                 "slam_comm_create_host: bad arguments");
    *out = nullptr;
    SLAM_TRY(require_device());
    slam_comm *c = new (std::nothrow) slam_comm();
    SLAM_REQUIRE(c, SLAM_E_NOMEM, "slam_comm_create_host: out of host memory");
    c->host_fn = allreduce;
    c->host_ctx = ctx;
    c->rank = rank;
    c->n_ranks = n_ranks;
    (void)hipGetDevice(&c->device);
    *out = c;
    return SLAM_OK;
}

int slam_comm_adopt(void *nccl_comm, slam_comm_t **out)
{
    SLAM_REQUIRE(nccl_comm && out, SLAM_E_INVALID, "slam_comm_adopt: bad arguments");
    slam_comm *c = new (std::nothrow) slam_comm();
    SLAM_REQUIRE(c, SLAM_E_NOMEM, "slam_comm_adopt: out of host memory");
    c->comm = static_cast<ncclComm_t>(nccl_comm);
    c->owned = false;
    (void)ncclCommUserRank(c->comm, &c->rank);
    (void)ncclCommCount(c->comm, &c->n_ranks);
    if (ncclCommCuDevice(c->comm, &c->device) != ncclSuccess) (void)hipGetDevice(&c->device);
    *out = c;
    return SLAM_OK;
}

void slam_comm_destroy(slam_comm_t *comm)
{
    if (!comm) return;
    if (comm->worker.joinable()) { // the helper thread works its queue off (a job's wait is bounded by the time-out) and ends
        {
            std::lock_guard<std::mutex> lk(comm->mu);
            comm->stop = true;
        }
        comm->cv_job.notify_all();
        comm->worker.join();
    }
    for (auto &sl : comm->slot) {
        if (sl.d_range) (void)hipFree(sl.d_range);
        if (sl.d_key) (void)hipFree(sl.d_key);
        if (sl.h_range) (void)hipHostFree(sl.h_range);
        if (sl.ev_range) (void)hipEventDestroy(sl.ev_range);
    }
    if (comm->h_stage) (void)hipHostFree(comm->h_stage);
    for (auto &pr : comm->ev_ar)
        for (hipEvent_t e : pr)
            if (e) (void)hipEventDestroy(e);
    if (comm->owned && comm->comm && !comm->aborted) (void)ncclCommDestroy(comm->comm);
    delete comm;
}

int slam_comm_info(slam_comm_t *comm, int *rank, int *n_ranks)
{
    SLAM_REQUIRE(comm, SLAM_E_INVALID, "null communicator");
    if (rank) *rank = comm->rank;
    if (n_ranks) *n_ranks = comm->n_ranks;
    return SLAM_OK;
}

int slam_comm_get_stats(slam_comm_t *comm, slam_comm_stats *out)
{
    SLAM_REQUIRE(comm && out, SLAM_E_INVALID, "slam_comm_get_stats: bad arguments");
    drain(comm); // (the helper thread keeps these books while it has jobs)
    memset(out, 0, sizeof *out);
    out->rank = comm->rank;
    out->n_ranks = comm->n_ranks;
    out->transport = comm->comm ? 0 : 1;
    if (comm->comm && !comm->aborted) { // what the transport itself says, not what the caller passed
        (void)ncclCommUserRank(comm->comm, &out->rank);
        (void)ncclCommCount(comm->comm, &out->n_ranks);
        int v = 0;
        if (ncclGetVersion(&v) == ncclSuccess) out->rccl_version = v;
    }
    out->merges = comm->merges;
    out->rows = comm->rows;
    out->bytes = comm->bytes;
    out->wait_ms = comm->wait_ms;
    out->helper_wait_ms = comm->helper_wait_ms;
    {
        std::lock_guard<std::mutex> lk(comm->mu);
        out->async_merges = comm->async_merges;
    }
    const long long n = comm->ar_recorded < slam_comm::kTimed ? comm->ar_recorded : slam_comm::kTimed;
    for (long long i = 0; i < n; ++i) {
        hipEvent_t *ev = comm->ev_ar[i];
        float       ms = 0.f;
        SLAM_HIP(hipEventSynchronize(ev[1]));
        SLAM_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
        out->allreduce_ms += ms;
    }
    out->timed = n;
    return SLAM_OK;
}

int slam_comm_stats_reset(slam_comm_t *comm)
{
    SLAM_REQUIRE(comm, SLAM_E_INVALID, "null communicator");
    drain(comm);
    comm->merges = comm->rows = comm->bytes = 0;
    comm->wait_ms = comm->helper_wait_ms = 0.0;
    comm->ar_recorded = 0;
    {
        std::lock_guard<std::mutex> lk(comm->mu);
        comm->async_merges = 0;
    }
    return SLAM_OK;
}

#define SLAM_COMM_ALIVE(comm)                                  \
    do {                                                       \
        if ((comm) && (comm)->failed.load() != SLAM_OK) {      \
            set_error("%s", (comm)->failed_msg);               \
            return (comm)->failed.load();                      \
        }                                                      \
    } while (0)

int slam_grid_allreduce(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream)
{
    SLAM_COMM_ALIVE(comm); // (a communicator that has failed takes no more work: on an adopted one the collectives already enqueued never end)
    SLAM_REQUIRE(grid && usable(comm), SLAM_E_INVALID, "slam_grid_allreduce: bad arguments (or a communicator that has failed)");
    int32_t *planes = nullptr;
    size_t   n = 0;
    SLAM_TRY(slam_grid_counts_dev(grid, &planes, &n));
    if (comm->comm)
        SLAM_NCCL(ncclAllReduce(planes, planes, n, ncclInt32, ncclSum, comm->comm, as_stream(stream)));
    else
        SLAM_TRY(host_sum_rows(comm, planes, n / 2, 0, n / 2, as_stream(stream)));
    int sy = 0;
    SLAM_TRY(slam_grid_info(grid, nullptr, &sy, nullptr, nullptr, nullptr));
    return slam_grid_mark_rows(grid, 0, sy - 1, stream); // every row may hold another rank's counts now
}

int slam_grid_allreduce_rows(slam_grid_t *grid, slam_comm_t *comm, int row_lo, int row_hi, slam_stream_t stream)
{
    SLAM_COMM_ALIVE(comm);
    SLAM_REQUIRE(grid && usable(comm), SLAM_E_INVALID, "slam_grid_allreduce_rows: bad arguments");
    if (row_hi < row_lo) return SLAM_OK;
    int32_t *planes = nullptr;
    size_t   n = 0;
    int      sx = 0, sy = 0;
    SLAM_TRY(slam_grid_counts_dev(grid, &planes, &n));
    SLAM_TRY(slam_grid_info(grid, &sx, &sy, nullptr, nullptr, nullptr));
    SLAM_REQUIRE(row_lo >= 0 && row_hi < sy, SLAM_E_INVALID, "slam_grid_allreduce_rows: rows %d..%d outside the grid", row_lo, row_hi);
    const size_t cells = n / 2, first = (size_t)row_lo * sx, count = (size_t)(row_hi - row_lo + 1) * sx;
    hipEvent_t  *ev = nullptr;
    if (comm->in_merge) { // the merge's own all-reduce: bracketed by events on its stream (read in slam_comm_get_stats)
        ev = comm->ev_ar[comm->ar_recorded % slam_comm::kTimed];
        for (int k = 0; k < 2; ++k)
            if (!ev[k]) SLAM_HIP(hipEventCreate(&ev[k]));
        SLAM_HIP(hipEventRecord(ev[0], as_stream(stream)));
    }
    if (comm->comm) {
        SLAM_NCCL(ncclGroupStart());
        ncclResult_t r = ncclAllReduce(planes + first, planes + first, count, ncclInt32, ncclSum, comm->comm, as_stream(stream));
        if (r == ncclSuccess)
            r = ncclAllReduce(planes + cells + first, planes + cells + first, count, ncclInt32, ncclSum, comm->comm, as_stream(stream));
        const ncclResult_t e = ncclGroupEnd();
        SLAM_NCCL(r);
        SLAM_NCCL(e);
    } else {
        SLAM_TRY(host_sum_rows(comm, planes, cells, first, count, as_stream(stream)));
    }
    if (ev) {
        SLAM_HIP(hipEventRecord(ev[1], as_stream(stream)));
        ++comm->ar_recorded;
        ++comm->merges;
        comm->rows += row_hi - row_lo + 1;
        comm->bytes += (long long)(2 * count * sizeof(int32_t));
    }
    return slam_grid_mark_rows(grid, row_lo, row_hi, stream); // rows this rank did not touch hold the others' counts now
}

// ---- the two halves of a merge, on whichever thread issues the communicator's calls (the caller's, or the helper's)

// first half: this rank's key, the minimum over the ranks, on its way to the host
static int begin_body(slam_comm *comm, slam_comm::Slot &sl, slam_grid_t *grid, hipStream_t st, const int32_t *d_dirty, int cell_x, int cell_y)
{
    if (!sl.ev_range) { // (ev_range is made last: a slot whose allocation failed half way is completed, not used as it is)
        if (!sl.d_range) SLAM_HIP(hipMalloc((void **)&sl.d_range, kKeyInts * sizeof(int)));
        if (!sl.d_key) SLAM_HIP(hipMalloc((void **)&sl.d_key, kKeyInts * sizeof(int)));
        if (!sl.h_range) SLAM_HIP(hipHostMalloc((void **)&sl.h_range, kKeyInts * sizeof(int), hipHostMallocDefault));
        SLAM_HIP(hipEventCreateWithFlags(&sl.ev_range, hipEventDisableTiming));
    }
    hipLaunchKernelGGL(merge_key_kernel, dim3(1), dim3(64), 0, st, d_dirty, cell_x, cell_y, sl.d_key);
    SLAM_HIP(hipGetLastError());
    if (comm->comm) {
        SLAM_NCCL(ncclAllReduce(sl.d_key, sl.d_range, kKeyInts, ncclInt32, ncclMin, comm->comm, st));
        SLAM_HIP(hipMemcpyAsync(sl.h_range, sl.d_range, kKeyInts * sizeof(int), hipMemcpyDeviceToHost, st));
    } else { // host-staged: this rank's key travels to the host now, the minimum over the ranks is taken in the second half
        SLAM_HIP(hipMemcpyAsync(sl.h_range, sl.d_key, kKeyInts * sizeof(int), hipMemcpyDeviceToHost, st));
    }
    SLAM_HIP(hipEventRecord(sl.ev_range, st));
    sl.grid = grid;
    return SLAM_OK;
}

// second half: the host's wait for the united range, the consistency of the ranks' windows, the rows' all-reduce on `st`
static int finish_body(slam_comm *comm, slam_comm::Slot &sl, slam_grid_t *grid, hipStream_t st, double *waited_ms, int *row_lo, int *row_hi)
{
    *row_lo = 0;
    *row_hi = -1;
    SLAM_TRY(wait_range(comm, sl.ev_range, waited_ms));
    int *k = sl.h_range;
    if (!comm->comm)
        SLAM_REQUIRE(comm->host_fn(comm->host_ctx, k, kKeyInts, SLAM_COMM_MIN) == 0, SLAM_E_COMM,
                     "the host transport's all-reduce (minimum of %d ints) failed", kKeyInts);
    // every rank sees the same six numbers, so every rank takes the same branch: nobody is left waiting in a collective
    SLAM_REQUIRE(k[2] == -k[3] && k[4] == -k[5], SLAM_E_INVALID,
                 "slam_grid_merge_finish: the ranks' rolling windows sit on different cells (x %d..%d, y %d..%d): storage rows "
                 "do not mean the same world cells; move every rank's grid to the same pose before a merge",
                 k[2], -k[3], k[4], -k[5]);
    int sy = 0;
    SLAM_TRY(slam_grid_info(grid, nullptr, &sy, nullptr, nullptr, nullptr));
    const bool none = k[0] > sy;
    *row_lo = none ? 0 : k[0];
    *row_hi = none ? -1 : -k[1];
    comm->in_merge = true;
    const int rc = slam_grid_allreduce_rows(grid, comm, *row_lo, *row_hi, (slam_stream_t)st);
    comm->in_merge = false;
    return rc;
}

// The helper thread of slam_grid_merge_async.  While jobs are posted it is the ONLY thread that issues calls on the
// communicator, one whole merge after the other in the order posted: the ranks' collectives meet in the same order on every
// rank (RCCL matches them by order, and takes calls from one thread at a time).
static void worker_main(slam_comm *c)
{
    (void)hipSetDevice(c->device);
    for (;;) {
        slam_comm::Job job;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv_job.wait(lk, [&] { return c->stop || !c->jobs.empty(); });
            if (c->jobs.empty()) return; // (stop: only after the queue has been worked off)
            job = c->jobs.front();
            c->jobs.pop_front();
        }
        slam_comm::Slot &sl = c->slot[0]; // one merge at a time here: no ring needed
        int lo = 0, hi = -1;
        int rc = comm_health(c);
        if (rc == SLAM_OK) rc = begin_body(c, sl, job.grid, job.st, job.d_dirty, job.cell_x, job.cell_y);
        if (rc == SLAM_OK) rc = finish_body(c, sl, job.grid, job.st, &c->helper_wait_ms, &lo, &hi);
        // what the caller asked to have enqueued behind the rows' sum, on the same stream
        if (rc == SLAM_OK && job.then == SLAM_MERGE_THEN_FINALIZE_RESET) rc = slam_grid_finalize_reset(job.grid, (slam_stream_t)job.st);
        if (rc == SLAM_OK && job.then == SLAM_MERGE_THEN_FOLD_FINALIZE) {
            rc = slam_grid_fold(job.grid, lo, hi, (slam_stream_t)job.st);
            if (rc == SLAM_OK) rc = slam_grid_finalize(job.grid, (slam_stream_t)job.st);
        }
        if (rc == SLAM_OK && job.done) {
            const hipError_t e = hipEventRecord(job.done, job.st);
            if (e != hipSuccess) rc = hip_fail(e, "hipEventRecord(done)", __FILE__, __LINE__);
        }
        {
            std::lock_guard<std::mutex> lk(c->mu);
            if (rc != SLAM_OK && rc != SLAM_E_INVALID) mark_failed(c, rc);
            slam_comm::Outcome &o = c->outcome[job.ticket % slam_comm::kDone];
            o.ticket = job.ticket;
            o.rc = rc;
            o.lo = lo;
            o.hi = hi;
            snprintf(o.err, sizeof o.err, "%s", rc == SLAM_OK ? "" : slam_last_error());
            ++c->completed;
            ++c->async_merges;
        }
        c->cv_done.notify_all();
    }
}

int slam_grid_merge_begin(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream)
{
    SLAM_COMM_ALIVE(comm);
    SLAM_REQUIRE(grid && usable(comm), SLAM_E_INVALID, "slam_grid_merge_begin: bad arguments");
    drain(comm); // merges handed to the helper thread come first: one thread at a time issues the communicator's calls
    SLAM_COMM_ALIVE(comm);
    SLAM_REQUIRE(comm->begun - comm->finished < (unsigned long long)slam_comm::kSlots, SLAM_E_INVALID,
                 "slam_grid_merge_begin: %d merges are in flight already; finish the oldest first", slam_comm::kSlots);
    int32_t *d_dirty = nullptr;
    int      cell_x = 0, cell_y = 0;
    SLAM_TRY(slam_grid_dirty_rows_dev(grid, &d_dirty));
    SLAM_TRY(slam_grid_window_cell(grid, &cell_x, &cell_y)); // as of the updates enqueued so far (slam_grid_set_pose keeps it on the host)
    // (slot 0 is the helper thread's; the caller's own merges in flight go round the others)
    slam_comm::Slot &sl = comm->slot[1 + comm->begun % (slam_comm::kSlots)];
    SLAM_TRY(begin_body(comm, sl, grid, as_stream(stream), d_dirty, cell_x, cell_y));
    ++comm->begun;
    return SLAM_OK;
}

int slam_grid_merge_finish(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream, int *row_lo, int *row_hi)
{
    SLAM_COMM_ALIVE(comm);
    SLAM_REQUIRE(grid && usable(comm), SLAM_E_INVALID, "slam_grid_merge_finish: bad arguments");
    drain(comm);
    SLAM_COMM_ALIVE(comm);
    SLAM_REQUIRE(comm->begun > comm->finished, SLAM_E_INVALID, "slam_grid_merge_finish: no merge in flight");
    slam_comm::Slot &sl = comm->slot[1 + comm->finished % (slam_comm::kSlots)];
    SLAM_REQUIRE(sl.grid == grid, SLAM_E_INVALID,
                 "slam_grid_merge_finish: merges finish in the order they were begun, and the oldest one in flight is another grid's");
    int lo = 0, hi = -1;
    const int rc = finish_body(comm, sl, grid, as_stream(stream), &comm->wait_ms, &lo, &hi); // the one host wait of a merge
    ++comm->finished;
    if (row_lo) *row_lo = lo;
    if (row_hi) *row_hi = hi;
    if (rc != SLAM_OK && rc != SLAM_E_INVALID) {
        std::lock_guard<std::mutex> lk(comm->mu);
        mark_failed(comm, rc);
    }
    return rc;
}

int slam_grid_merge_async(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream, int then, slam_event_t done,
                          unsigned long long *ticket)
{
    SLAM_COMM_ALIVE(comm);
    SLAM_REQUIRE(grid && usable(comm) && ticket, SLAM_E_INVALID, "slam_grid_merge_async: bad arguments");
    SLAM_REQUIRE(then == SLAM_MERGE_THEN_NOTHING || then == SLAM_MERGE_THEN_FINALIZE_RESET || then == SLAM_MERGE_THEN_FOLD_FINALIZE,
                 SLAM_E_INVALID, "slam_grid_merge_async: unknown continuation %d", then);
    SLAM_REQUIRE(comm->begun == comm->finished, SLAM_E_INVALID,
                 "slam_grid_merge_async: finish the merges begun with slam_grid_merge_begin first (one order of collectives per communicator)");
    slam_comm::Job job{};
    job.grid = grid;
    job.st = as_stream(stream);
    job.then = then;
    job.done = reinterpret_cast<hipEvent_t>(done);
    // what the merge is about is fixed HERE, as of the updates the caller has enqueued so far: the dirty-range buffer in use
    // (slam_grid_finalize_reset alternates between two) and the window's cell
    int32_t *d_dirty = nullptr;
    SLAM_TRY(slam_grid_dirty_rows_dev(grid, &d_dirty));
    job.d_dirty = d_dirty;
    SLAM_TRY(slam_grid_window_cell(grid, &job.cell_x, &job.cell_y));
    std::unique_lock<std::mutex> lk(comm->mu);
    if (!comm->worker.joinable()) {
        try {
            comm->worker = std::thread(worker_main, comm);
        } catch (const std::exception &ex) { // (std::system_error: no thread to be had -- nothing may throw through the C boundary)
            set_error("slam_grid_merge_async: the helper thread could not be started (%s)", ex.what());
            return SLAM_E_NOMEM;
        }
    }
    if (comm->posted - comm->completed >= (unsigned long long)slam_comm::kMaxPosted) { // back-pressure on the caller's thread
        const auto t0 = std::chrono::steady_clock::now();
        comm->cv_done.wait(lk, [&] { return comm->posted - comm->completed < (unsigned long long)slam_comm::kMaxPosted; });
        comm->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    job.ticket = ++comm->posted;
    comm->jobs.push_back(job);
    *ticket = job.ticket;
    lk.unlock();
    comm->cv_job.notify_one();
    return SLAM_OK;
}

int slam_comm_ticket_wait(slam_comm_t *comm, unsigned long long ticket, int *row_lo, int *row_hi)
{
    SLAM_REQUIRE(comm, SLAM_E_INVALID, "null communicator");
    std::unique_lock<std::mutex> lk(comm->mu);
    SLAM_REQUIRE(ticket >= 1 && ticket <= comm->posted, SLAM_E_INVALID, "slam_comm_ticket_wait: no such ticket");
    if (comm->completed < ticket) {
        const auto t0 = std::chrono::steady_clock::now();
        comm->cv_done.wait(lk, [&] { return comm->completed >= ticket; });
        comm->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    const slam_comm::Outcome &o = comm->outcome[ticket % slam_comm::kDone];
    if (row_lo) *row_lo = 0;
    if (row_hi) *row_hi = -1;
    if (o.ticket != ticket) { // long gone: only the communicator's sticky state can still be told
        SLAM_COMM_ALIVE(comm);
        return SLAM_OK;
    }
    if (row_lo) *row_lo = o.lo;
    if (row_hi) *row_hi = o.hi;
    if (o.rc != SLAM_OK) set_error("%s", o.err);
    return o.rc;
}

int slam_comm_drain(slam_comm_t *comm)
{
    SLAM_REQUIRE(comm, SLAM_E_INVALID, "null communicator");
    drain(comm);
    SLAM_COMM_ALIVE(comm);
    return SLAM_OK;
}

int slam_comm_set_timeout(slam_comm_t *comm, double seconds)
{
    SLAM_REQUIRE(comm && seconds > 0, SLAM_E_INVALID, "slam_comm_set_timeout: bad arguments");
    comm->timeout_s.store(seconds);
    return SLAM_OK;
}

int slam_comm_check(slam_comm_t *comm)
{
    SLAM_REQUIRE(comm, SLAM_E_INVALID, "null communicator");
    std::lock_guard<std::mutex> lk(comm->mu);
    const int rc = comm_health(comm);
    if (rc != SLAM_OK) mark_failed(comm, rc);
    return rc;
}

static int mapper_merge_begin(void *ctx, slam_grid_t *grid, slam_stream_t stream, int *, int *)
{
    return slam_grid_merge_begin(grid, static_cast<slam_comm_t *>(ctx), stream);
}
static int mapper_merge_finish(void *ctx, slam_grid_t *grid, slam_stream_t stream, int *row_lo, int *row_hi)
{
    return slam_grid_merge_finish(grid, static_cast<slam_comm_t *>(ctx), stream, row_lo, row_hi);
}

int slam_mapper_use_comm(slam_mapper_t *mapper, slam_comm_t *comm)
{
    SLAM_REQUIRE(mapper && usable(comm), SLAM_E_INVALID, "slam_mapper_use_comm: bad arguments");
    return slam_mapper_set_merge(mapper, mapper_merge_begin, mapper_merge_finish, comm);
}

} // extern "C"
