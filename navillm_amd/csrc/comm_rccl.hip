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
    int (*GetVersion)(int*);
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
        q.GetVersion = (int (*)(int*))dlsym(h, "ncclGetVersion");
        q.ok = q.GetVersion && q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.AllReduce && q.Broadcast && q.ReduceScatter && q.AllGather;
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

// NCCL_VERSION_CODE of the librccl bound at run time (22707 = 2.27.7 on ROCm 7.2), or a negative status
int nv_comm_rccl_version(void) {
    if (!rccl()->ok) return NV_ERR_COMM;
    int v = 0;
    if (rccl()->GetVersion(&v) != ncclSuccess) return NV_ERR_COMM;
    return v;
}

// The enum values above are restated from rccl.h (no link- or compile-time dependency on a particular librccl), so they are
// CHECKED against the library that was actually bound, once per communicator: a mean all-reduce of known bf16 and fp32
// vectors.  Rank r contributes {r+1, 3*4^(r&1)} (x1.5 in fp32): a library whose ncclAvg / ncclBfloat16 / ncclFloat32 had other
// values would return the sum / max / min / product, or the mean of the same bits read as another type (bf16 3 and 12 read
// as fp16 average to 2.375, not 7.5).  Collective: every rank runs it inside nv_comm_init.
static int comm_self_check(nv_ctx* c) {
    const int W = c->world, r = c->rank;
    struct { uint16_t b[8]; float f[8]; } h, o;
    memset(&h, 0, sizeof(h));
    auto bf = [](float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); };     // exact for these values
    const float v0 = (float)(r + 1), v1 = 3.f * ((r & 1) ? 4.f : 1.f);
    h.b[0] = bf(v0); h.b[1] = bf(v1);
    h.f[0] = 1.5f * v0; h.f[1] = 1.5f * v1;
    void* d = nullptr;
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return NV_ERR_COMM;
    int rc = NV_OK;
    if (hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) rc = NV_ERR_COMM;
    char* p = (char*)d;
    if (rc == NV_OK && rccl()->AllReduce(p, p, 8, ncclBfloat16, ncclAvg, c->comm, (hipStream_t)0) != ncclSuccess) rc = NV_ERR_COMM;
    if (rc == NV_OK && rccl()->AllReduce(p + 16, p + 16, 8, ncclFloat32, ncclAvg, c->comm, (hipStream_t)0) != ncclSuccess) rc = NV_ERR_COMM;
    if (rc == NV_OK && (hipStreamSynchronize((hipStream_t)0) != hipSuccess || hipMemcpy(&o, d, sizeof(o), hipMemcpyDeviceToHost) != hipSuccess))
        rc = NV_ERR_COMM;
    hipFree(d);
    if (rc != NV_OK) return rc;
    double e0 = 0, e1 = 0;
    for (int q = 0; q < W; ++q) { e0 += q + 1; e1 += 3.0 * ((q & 1) ? 4.0 : 1.0); }
    e0 /= W; e1 /= W;
    auto f_of = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; };
    auto close = [](double got, double want) { const double d_ = got - want; return (d_ < 0 ? -d_ : d_) <= 0.01 * want; };
    if (!close(f_of(o.b[0]), e0) || !close(f_of(o.b[1]), e1) || !close(o.f[0], 1.5 * e0) || !close(o.f[1], 1.5 * e1)) return NV_ERR_COMM;
    for (int i = 2; i < 8; ++i)
        if (o.b[i] != 0 || o.f[i] != 0.f) return NV_ERR_COMM;
    return NV_OK;
}

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
    // the hand-restated ABI slice is for NCCL major version 2 (rccl.h of ROCm 7.x: 2.2x); refuse anything else up front
    const int ver = nv_comm_rccl_version();
    if (ver < 20000 || ver >= 30000) { free(c); return NV_ERR_COMM; }
    if (rccl()->CommInitRank(&c->comm, world, uid, rank) != ncclSuccess) { free(c); return NV_ERR_COMM; }
    if (comm_self_check(c) != NV_OK) { rccl()->CommDestroy(c->comm); free(c); return NV_ERR_COMM; }
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
