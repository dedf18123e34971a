// nv_comm_*: the data-parallel exchange of the path (reference: DDP all-reduce behind
// tools/optims.py:52-54, task-id broadcast tasks/loaders.py:176-179) as C entry points over RCCL.
//
// One communicator per process (= per GPU).  RCCL is bound at run time (dlopen/dlsym) so that libnavillm_hip.so has
// no link-time dependency on a particular librccl: in a torch process the copy torch already loaded is reused,
// otherwise /opt/rocm's.  Collectives are in place, stream-ordered, and never allocate.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "nv_common.h"

namespace {

// the slice of rccl.h this file needs (ROCm 7.2 / RCCL 2.2x ABI)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1, ncclFloat32 = 7, ncclBfloat16 = 9 };
enum { ncclSum = 0, ncclAvg = 4 };

struct Rccl {
    int (*GetUniqueId)(ncclUniqueId*);
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    int (*CommDestroy)(ncclComm_t);
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
    int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
    bool ok;
};

Rccl* rccl() {
    static Rccl r = [] {
        Rccl q;
        memset(&q, 0, sizeof(q));
        void* h = nullptr;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);            // already mapped (torch's copy)?
        for (const char* n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return q;
        q.GetUniqueId = (int (*)(ncclUniqueId*))dlsym(h, "ncclGetUniqueId");
        q.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(h, "ncclCommInitRank");
        q.CommDestroy = (int (*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
        q.AllReduce = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
        q.Broadcast = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(h, "ncclBroadcast");
        q.ReduceScatter = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(h, "ncclReduceScatter");
        q.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllGather");
        q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.AllReduce && q.Broadcast && q.ReduceScatter && q.AllGather;
        return q;
    }();
    return &r;
}

}  // namespace

struct nv_ctx {
    ncclComm_t comm;
    int rank, world;
};

#define NV_ERR_COMM (-4)

extern "C" {

int nv_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

// rank 0 calls this and ships the 128 bytes to the other ranks out of band (file, TCP store, MPI, ...)
int nv_comm_unique_id(void* id_out) {
    if (!id_out) return NV_ERR_ARG;
    if (!rccl()->ok) return NV_ERR_COMM;
    ncclUniqueId id;
    if (rccl()->GetUniqueId(&id) != ncclSuccess) return NV_ERR_COMM;
    memcpy(id_out, &id, sizeof(id));
    return NV_OK;
}

// collective over all `world` ranks; uses the calling thread's current HIP device
int nv_comm_init(nv_ctx** out, const void* id, int rank, int world) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return NV_ERR_ARG;
    if (!rccl()->ok) return NV_ERR_COMM;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    nv_ctx* c = (nv_ctx*)malloc(sizeof(nv_ctx));
    if (!c) return NV_ERR_ARG;
    c->rank = rank; c->world = world; c->comm = nullptr;
    if (rccl()->CommInitRank(&c->comm, world, uid, rank) != ncclSuccess) { free(c); return NV_ERR_COMM; }
    *out = c;
    return NV_OK;
}

int nv_comm_rank(const nv_ctx* c) { return c ? c->rank : -1; }
int nv_comm_world(const nv_ctx* c) { return c ? c->world : -1; }

// in place; average != 0 -> mean over ranks (ncclAvg), else sum
int nv_comm_allreduce_bf16(nv_ctx* c, void* buf, long count, int average, void* stream) {
    if (!c || (!buf && count) || count < 0) return NV_ERR_ARG;
    if (count == 0) return NV_OK;
    return rccl()->AllReduce(buf, buf, (size_t)count, ncclBfloat16, average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream) == ncclSuccess
               ? NV_OK : NV_ERR_COMM;
}

int nv_comm_allreduce_f32(nv_ctx* c, void* buf, long count, int average, void* stream) {
    if (!c || (!buf && count) || count < 0) return NV_ERR_ARG;
    if (count == 0) return NV_OK;
    return rccl()->AllReduce(buf, buf, (size_t)count, ncclFloat32, average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream) == ncclSuccess
               ? NV_OK : NV_ERR_COMM;
}

// Two-phase form of the gradient mean for a fully connected xGMI node: in-place reduce-scatter (rank r
// ends up owning the reduced chunk buf[r*count/world .. (r+1)*count/world)) followed by an in-place all-gather of the
// chunks.  count must be a multiple of world (the flat gradient slices are multiples of 64 elements).
int nv_comm_reduce_scatter(nv_ctx* c, void* buf, long count, int is_bf16, int average, void* stream) {
    if (!c || (!buf && count) || count < 0 || count % c->world) return NV_ERR_ARG;
    if (count == 0) return NV_OK;
    const size_t chunk = (size_t)(count / c->world), esz = is_bf16 ? 2 : 4;
    char* mine = (char*)buf + (size_t)c->rank * chunk * esz;
    return rccl()->ReduceScatter(buf, mine, chunk, is_bf16 ? ncclBfloat16 : ncclFloat32, average ? ncclAvg : ncclSum, c->comm,
                                 (hipStream_t)stream) == ncclSuccess ? NV_OK : NV_ERR_COMM;
}

int nv_comm_all_gather(nv_ctx* c, void* buf, long bytes, void* stream) {
    if (!c || (!buf && bytes) || bytes < 0 || bytes % c->world) return NV_ERR_ARG;
    if (bytes == 0) return NV_OK;
    const size_t chunk = (size_t)(bytes / c->world);
    const char* mine = (const char*)buf + (size_t)c->rank * chunk;
    return rccl()->AllGather(mine, buf, chunk, ncclUint8, c->comm, (hipStream_t)stream) == ncclSuccess ? NV_OK : NV_ERR_COMM;
}

int nv_comm_broadcast(nv_ctx* c, void* buf, long bytes, int root, void* stream) {
    if (!c || (!buf && bytes) || bytes < 0 || root < 0 || root >= c->world) return NV_ERR_ARG;
    if (bytes == 0) return NV_OK;
    return rccl()->Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, c->comm, (hipStream_t)stream) == ncclSuccess ? NV_OK : NV_ERR_COMM;
}

int nv_comm_destroy(nv_ctx* c) {
    if (!c) return NV_OK;
    int rc = NV_OK;
    if (c->comm && rccl()->ok && rccl()->CommDestroy(c->comm) != ncclSuccess) rc = NV_ERR_COMM;
    free(c);
    return rc;
}

}  // extern "C"
