// rccl.hip -- RCCL merge of the grid count planes (slam_mi355x_rccl.h).
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <new>

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
    static constexpr int kSlots = 4;
    struct Slot {
        int       *d_key = nullptr;   // [kKeyInts] this rank's key
        int       *d_range = nullptr; // [kKeyInts] the minimum over the ranks
        int       *h_range = nullptr; // pinned copy
        hipEvent_t ev_range = nullptr;
        slam_grid_t *grid = nullptr;  // the grid the merge was begun on
    } slot[kSlots];
    unsigned long long begun = 0, finished = 0;
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

bool usable(const slam_comm *c) { return c && (c->comm || c->host_fn); }

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
    *out = c;
    return SLAM_OK;
}

void slam_comm_destroy(slam_comm_t *comm)
{
    if (!comm) return;
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
    if (comm->owned && comm->comm) (void)ncclCommDestroy(comm->comm);
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
    memset(out, 0, sizeof *out);
    out->rank = comm->rank;
    out->n_ranks = comm->n_ranks;
    out->transport = comm->comm ? 0 : 1;
    if (comm->comm) { // what the transport itself says, not what the caller passed
        (void)ncclCommUserRank(comm->comm, &out->rank);
        (void)ncclCommCount(comm->comm, &out->n_ranks);
        int v = 0;
        if (ncclGetVersion(&v) == ncclSuccess) out->rccl_version = v;
    }
    out->merges = comm->merges;
    out->rows = comm->rows;
    out->bytes = comm->bytes;
    out->wait_ms = comm->wait_ms;
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
    comm->merges = comm->rows = comm->bytes = 0;
    comm->wait_ms = 0.0;
    comm->ar_recorded = 0;
    return SLAM_OK;
}

int slam_grid_allreduce(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream)
{
    SLAM_REQUIRE(grid && usable(comm), SLAM_E_INVALID, "slam_grid_allreduce: bad arguments");
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

int slam_grid_merge_begin(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream)
{
    SLAM_REQUIRE(grid && usable(comm), SLAM_E_INVALID, "slam_grid_merge_begin: bad arguments");
    SLAM_REQUIRE(comm->begun - comm->finished < (unsigned long long)slam_comm::kSlots, SLAM_E_INVALID,
                 "slam_grid_merge_begin: %d merges are in flight already; finish the oldest first", slam_comm::kSlots);
    slam_comm::Slot &sl = comm->slot[comm->begun % slam_comm::kSlots];
    if (!sl.d_range) {
        SLAM_HIP(hipMalloc((void **)&sl.d_range, kKeyInts * sizeof(int)));
        SLAM_HIP(hipMalloc((void **)&sl.d_key, kKeyInts * sizeof(int)));
        SLAM_HIP(hipHostMalloc((void **)&sl.h_range, kKeyInts * sizeof(int), hipHostMallocDefault));
        SLAM_HIP(hipEventCreateWithFlags(&sl.ev_range, hipEventDisableTiming));
    }
    int32_t *d_dirty = nullptr;
    int      cell_x = 0, cell_y = 0;
    SLAM_TRY(slam_grid_dirty_rows_dev(grid, &d_dirty));
    SLAM_TRY(slam_grid_window_cell(grid, &cell_x, &cell_y)); // as of the updates enqueued so far (slam_grid_set_pose keeps it on the host)
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(merge_key_kernel, dim3(1), dim3(64), 0, st, d_dirty, cell_x, cell_y, sl.d_key);
    SLAM_HIP(hipGetLastError());
    if (comm->comm) {
        SLAM_NCCL(ncclAllReduce(sl.d_key, sl.d_range, kKeyInts, ncclInt32, ncclMin, comm->comm, st));
        SLAM_HIP(hipMemcpyAsync(sl.h_range, sl.d_range, kKeyInts * sizeof(int), hipMemcpyDeviceToHost, st));
    } else { // host-staged: this rank's key travels to the host now, the minimum over the ranks is taken in finish
        SLAM_HIP(hipMemcpyAsync(sl.h_range, sl.d_key, kKeyInts * sizeof(int), hipMemcpyDeviceToHost, st));
    }
    SLAM_HIP(hipEventRecord(sl.ev_range, st));
    sl.grid = grid;
    ++comm->begun;
    return SLAM_OK;
}

int slam_grid_merge_finish(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream, int *row_lo, int *row_hi)
{
    SLAM_REQUIRE(grid && usable(comm) && comm->begun > comm->finished, SLAM_E_INVALID, "slam_grid_merge_finish: no merge in flight");
    slam_comm::Slot &sl = comm->slot[comm->finished % slam_comm::kSlots];
    SLAM_REQUIRE(sl.grid == grid, SLAM_E_INVALID,
                 "slam_grid_merge_finish: merges finish in the order they were begun, and the oldest one in flight is another grid's");
    ++comm->finished;
    {
        const auto t0 = std::chrono::steady_clock::now();
        const hipError_t e = hipEventSynchronize(sl.ev_range); // the one host wait of a merge
        comm->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        SLAM_HIP(e);
    }
    int *k = sl.h_range;
    if (!comm->comm)
        SLAM_REQUIRE(comm->host_fn(comm->host_ctx, k, kKeyInts, SLAM_COMM_MIN) == 0, SLAM_E_HIP,
                     "the host transport's all-reduce (minimum of %d ints) failed", kKeyInts);
    if (row_lo) *row_lo = 0;
    if (row_hi) *row_hi = -1;
    // every rank sees the same six numbers, so every rank takes the same branch: nobody is left waiting in a collective
    SLAM_REQUIRE(k[2] == -k[3] && k[4] == -k[5], SLAM_E_INVALID,
                 "slam_grid_merge_finish: the ranks' rolling windows sit on different cells (x %d..%d, y %d..%d): storage rows "
                 "do not mean the same world cells; move every rank's grid to the same pose before a merge",
                 k[2], -k[3], k[4], -k[5]);
    int sy = 0;
    SLAM_TRY(slam_grid_info(grid, nullptr, &sy, nullptr, nullptr, nullptr));
    const bool none = k[0] > sy;
    const int  lo = none ? 0 : k[0], hi = none ? -1 : -k[1];
    if (row_lo) *row_lo = lo;
    if (row_hi) *row_hi = hi;
    comm->in_merge = true;
    const int rc = slam_grid_allreduce_rows(grid, comm, lo, hi, stream);
    comm->in_merge = false;
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
