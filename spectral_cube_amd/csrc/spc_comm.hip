// RCCL stitch of the per-GPU row strips of a 2-D map.
// One process per GPU; the unique id travels through the host launcher.
#include "spc_common.h"
#include <rccl/rccl.h>

static_assert(sizeof(ncclUniqueId) <= SPC_COMM_ID_BYTES, "ncclUniqueId larger than SPC_COMM_ID_BYTES");

struct SpcComm {
    ncclComm_t comm;
    int device;
    int nranks;
    int rank;
};

#define SPC_NCCL(call)                                                        \
    do {                                                                      \
        ncclResult_t r_ = (call);                                             \
        if (r_ != ncclSuccess) {                                              \
            spc_set_error("%s failed: %s", #call, ncclGetErrorString(r_));    \
            return SPC_ERR_COMM;                                              \
        }                                                                     \
    } while (0)

extern "C" {

int spc_comm_unique_id(uint8_t id[SPC_COMM_ID_BYTES]) {
    SPC_REQUIRE(id, "id is NULL");
    ncclUniqueId u;
    SPC_NCCL(ncclGetUniqueId(&u));
    memset(id, 0, SPC_COMM_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return SPC_OK;
}

int spc_comm_init(int device, const uint8_t id[SPC_COMM_ID_BYTES], int nranks, int rank, void** comm) {
    SPC_REQUIRE(id && comm, "NULL pointer argument");
    SPC_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d / nranks %d", rank, nranks);
    SPC_HIP(hipSetDevice(device));     // RCCL binds the communicator to the current device
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    SpcComm* c = new SpcComm{nullptr, device, nranks, rank};
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        spc_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
        delete c;
        return SPC_ERR_COMM;
    }
    *comm = c;
    return SPC_OK;
}

int spc_comm_destroy(void* comm) {
    if (!comm) return SPC_OK;
    SpcComm* c = (SpcComm*)comm;
    ncclResult_t r = ncclCommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) {
        spc_set_error("ncclCommDestroy failed: %s", ncclGetErrorString(r));
        return SPC_ERR_COMM;
    }
    return SPC_OK;
}

int spc_allgather_rows(void* comm, void* stream, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    SPC_REQUIRE(comm && d_send && d_recv, "NULL pointer argument");
    SpcComm* c = (SpcComm*)comm;
    SPC_DEVICE(c->device);
    SPC_NCCL(ncclAllGather(d_send, d_recv, bytes_per_rank, ncclChar, c->comm, (hipStream_t)stream));
    return SPC_OK;
}

// n all-gathers as ONE grouped RCCL launch (ncclGroupStart / End): the three moment maps of one row chunk, each into
// its own destination map, without three launch latencies.
int spc_allgather_rows_batch(void* comm, void* stream, int n, const void* const* d_send, void* const* d_recv,
                             const size_t* bytes_per_rank) {
    SPC_REQUIRE(comm && d_send && d_recv && bytes_per_rank && n >= 1 && n <= 64, "bad argument");
    SpcComm* c = (SpcComm*)comm;
    SPC_DEVICE(c->device);
    SPC_NCCL(ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        ncclResult_t r = ncclAllGather(d_send[i], d_recv[i], bytes_per_rank[i], ncclChar, c->comm, (hipStream_t)stream);
        if (r != ncclSuccess) {
            (void)ncclGroupEnd();
            spc_set_error("ncclAllGather failed: %s", ncclGetErrorString(r));
            return SPC_ERR_COMM;
        }
    }
    SPC_NCCL(ncclGroupEnd());
    return SPC_OK;
}

}  // extern "C"
