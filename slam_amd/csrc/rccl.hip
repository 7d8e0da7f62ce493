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
    return SLAM_OK;
}

} // extern "C"
