// libspe_comm.so: RCCL collectives behind a C ABI (include/spe_comm.h).  One communicator per process (one process per
// GPU, reference util/misc.py:414-436); every call is asynchronous on the caller's stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include "../../../include/spe_comm.h"

static ncclComm_t g_comm = nullptr;
static int g_rank = -1, g_world = 0;

static_assert(sizeof(ncclUniqueId) == SPE_COMM_ID_BYTES, "ncclUniqueId size");

static int st(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }
static bool dt(int dtype, ncclDataType_t* out) {
    if (dtype == SPE_COMM_F32) { *out = ncclFloat32; return true; }
    if (dtype == SPE_COMM_BF16) { *out = ncclBfloat16; return true; }
    return false;
}

extern "C" int spe_comm_unique_id(void* id_out) {
    if (!id_out) return -2;
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return st(r);
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int spe_comm_init(int rank, int world, const void* id) {
    if (!id || world < 1 || rank < 0 || rank >= world) return -2;
    if (g_comm) return -2;                       // one communicator per process
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    const ncclResult_t r = ncclCommInitRank(&g_comm, world, uid, rank);
    if (r != ncclSuccess) { g_comm = nullptr; return st(r); }
    g_rank = rank; g_world = world;
    return 0;
}

extern "C" int spe_comm_world(int* rank, int* world) {
    if (!g_comm) return -1;
    if (rank) *rank = g_rank;
    if (world) *world = g_world;
    return 0;
}

extern "C" int spe_comm_allreduce(void* buf, long count, int dtype, spe_stream_t stream) {
    if (!g_comm) return -1;
    ncclDataType_t t;
    if (!buf || count < 0 || !dt(dtype, &t)) return -2;
    if (count == 0) return 0;
    return st(ncclAllReduce(buf, buf, (size_t)count, t, ncclSum, g_comm, (hipStream_t)stream));
}

extern "C" int spe_comm_broadcast(void* buf, long count, int dtype, int root, spe_stream_t stream) {
    if (!g_comm) return -1;
    ncclDataType_t t;
    if (!buf || count < 0 || root < 0 || root >= g_world || !dt(dtype, &t)) return -2;
    if (count == 0) return 0;
    return st(ncclBroadcast(buf, buf, (size_t)count, t, root, g_comm, (hipStream_t)stream));
}

extern "C" int spe_comm_destroy(void) {
    if (!g_comm) return 0;
    const ncclResult_t r = ncclCommDestroy(g_comm);
    g_comm = nullptr; g_rank = -1; g_world = 0;
    return st(r);
}
