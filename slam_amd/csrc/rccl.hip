// rccl.hip -- RCCL merge of the grid count planes (slam_mi355x_rccl.h).
#include <rccl/rccl.h>

#include <cstring>
#include <new>

#include "common.hpp"
#include "slam_mi355x_rccl.h"

using namespace slam;

struct slam_comm {
    ncclComm_t comm = nullptr;
    bool       owned = false;
    int        rank = 0, n_ranks = 1;
    int       *d_range = nullptr;   // [2] united dirty range of a merge in flight
    int       *h_range = nullptr;   // pinned copy
    hipEvent_t ev_range = nullptr;
    bool       pending = false;
};

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
    if (comm->d_range) (void)hipFree(comm->d_range);
    if (comm->h_range) (void)hipHostFree(comm->h_range);
    if (comm->ev_range) (void)hipEventDestroy(comm->ev_range);
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

int slam_grid_allreduce(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream)
{
    SLAM_REQUIRE(grid && comm && comm->comm, SLAM_E_INVALID, "slam_grid_allreduce: bad arguments");
    int32_t *planes = nullptr;
    size_t   n = 0;
    SLAM_TRY(slam_grid_counts_dev(grid, &planes, &n));
    SLAM_NCCL(ncclAllReduce(planes, planes, n, ncclInt32, ncclSum, comm->comm, as_stream(stream)));
    int sy = 0;
    SLAM_TRY(slam_grid_info(grid, nullptr, &sy, nullptr, nullptr, nullptr));
    return slam_grid_mark_rows(grid, 0, sy - 1, stream); // every row may hold another rank's counts now
}

int slam_grid_allreduce_rows(slam_grid_t *grid, slam_comm_t *comm, int row_lo, int row_hi, slam_stream_t stream)
{
    SLAM_REQUIRE(grid && comm && comm->comm, SLAM_E_INVALID, "slam_grid_allreduce_rows: bad arguments");
    if (row_hi < row_lo) return SLAM_OK;
    int32_t *planes = nullptr;
    size_t   n = 0;
    int      sx = 0, sy = 0;
    SLAM_TRY(slam_grid_counts_dev(grid, &planes, &n));
    SLAM_TRY(slam_grid_info(grid, &sx, &sy, nullptr, nullptr, nullptr));
    SLAM_REQUIRE(row_lo >= 0 && row_hi < sy, SLAM_E_INVALID, "slam_grid_allreduce_rows: rows %d..%d outside the grid", row_lo, row_hi);
    const size_t cells = n / 2, first = (size_t)row_lo * sx, count = (size_t)(row_hi - row_lo + 1) * sx;
    SLAM_NCCL(ncclGroupStart());
    ncclResult_t r = ncclAllReduce(planes + first, planes + first, count, ncclInt32, ncclSum, comm->comm, as_stream(stream));
    if (r == ncclSuccess)
        r = ncclAllReduce(planes + cells + first, planes + cells + first, count, ncclInt32, ncclSum, comm->comm, as_stream(stream));
    const ncclResult_t e = ncclGroupEnd();
    SLAM_NCCL(r);
    SLAM_NCCL(e);
    return slam_grid_mark_rows(grid, row_lo, row_hi, stream); // rows this rank did not touch hold the others' counts now
}

int slam_grid_merge_begin(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream)
{
    SLAM_REQUIRE(grid && comm && comm->comm, SLAM_E_INVALID, "slam_grid_merge_begin: bad arguments");
    SLAM_REQUIRE(!comm->pending, SLAM_E_INVALID, "slam_grid_merge_begin: the previous merge was not finished");
    if (!comm->d_range) {
        SLAM_HIP(hipMalloc((void **)&comm->d_range, 2 * sizeof(int)));
        SLAM_HIP(hipHostMalloc((void **)&comm->h_range, 2 * sizeof(int), hipHostMallocDefault));
        SLAM_HIP(hipEventCreateWithFlags(&comm->ev_range, hipEventDisableTiming));
    }
    int32_t *d_dirty = nullptr;
    SLAM_TRY(slam_grid_dirty_rows_dev(grid, &d_dirty));
    // {lowest row, -(highest row)}: the union over the ranks is one minimum
    SLAM_NCCL(ncclAllReduce(d_dirty, comm->d_range, 2, ncclInt32, ncclMin, comm->comm, as_stream(stream)));
    SLAM_HIP(hipMemcpyAsync(comm->h_range, comm->d_range, 2 * sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
    SLAM_HIP(hipEventRecord(comm->ev_range, as_stream(stream)));
    comm->pending = true;
    return SLAM_OK;
}

int slam_grid_merge_finish(slam_grid_t *grid, slam_comm_t *comm, slam_stream_t stream, int *row_lo, int *row_hi)
{
    SLAM_REQUIRE(grid && comm && comm->comm && comm->pending, SLAM_E_INVALID, "slam_grid_merge_finish: no merge in flight");
    comm->pending = false;
    SLAM_HIP(hipEventSynchronize(comm->ev_range));
    int sy = 0;
    SLAM_TRY(slam_grid_info(grid, nullptr, &sy, nullptr, nullptr, nullptr));
    const bool none = comm->h_range[0] > sy;
    const int  lo = none ? 0 : comm->h_range[0], hi = none ? -1 : -comm->h_range[1];
    if (row_lo) *row_lo = lo;
    if (row_hi) *row_hi = hi;
    return slam_grid_allreduce_rows(grid, comm, lo, hi, stream);
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
    SLAM_REQUIRE(mapper && comm && comm->comm, SLAM_E_INVALID, "slam_mapper_use_comm: bad arguments");
    return slam_mapper_set_merge(mapper, mapper_merge_begin, mapper_merge_finish, comm);
}

} // extern "C"
