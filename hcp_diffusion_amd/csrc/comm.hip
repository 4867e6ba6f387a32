// comm.hip — the data-parallel exchange of the hot path behind the C ABI: RCCL collectives over xGMI on flat buffers.
//
// Replaces what `accelerator.backward` / torch DDP do for the reference (train_ac.py:117-123,175,482: bucketed all-reduce of
// the trainable parameters' gradients; SURVEY.md §8(b) export list, §8(e)): ONE all-reduce per flat gradient bucket for the
// LoRA-sized buckets (12 MB: latency bound), reduce-scatter -> sharded fused AdamW -> all-gather for the full-fine-tune /
// ControlNet buckets (3.4 GB / 1.4 GB fp32: a ring all-reduce is per-link bound on point-to-point xGMI).
// One process per GPU; the communicator is an opaque handle owned by the caller (no process-global state); every call is
// stream-ordered on the caller's HIP stream, allocates nothing and never synchronises, so it can sit inside a captured
// hipGraph like any kernel launch.
//
// RCCL is bound at run time (dlopen, preferring the copy PyTorch-ROCm already mapped so the process holds one RCCL): the
// kernel library itself has no link-time dependency on it and single-GPU users never load it.  The unique id travels
// between ranks by whatever side channel the host has (the Python side uses the torch.distributed store).
// (The CPU interpreter build of the tests does not compile this file: tests/emu/hcp_emu_comm.cpp provides the same entry points for a
// world of one.)
#include "hcp_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>

namespace {

struct HcpComm {
    int rank, world;
    ncclComm_t nccl;
};

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// Resolved once per process (immutable afterwards; C++11 static initialisation is thread-safe).
const Rccl* rccl() {
    static const Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!x.so) x.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);   // already mapped (PyTorch's copy)?
        for (const char* n : names) if (!x.so) x.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!x.so) x.so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!x.so) return x;
#define HCP_SYM(field, name) x.field = (decltype(x.field))dlsym(x.so, name)
        HCP_SYM(GetUniqueId, "ncclGetUniqueId"); HCP_SYM(CommInitRank, "ncclCommInitRank"); HCP_SYM(CommDestroy, "ncclCommDestroy");
        HCP_SYM(AllReduce, "ncclAllReduce"); HCP_SYM(ReduceScatter, "ncclReduceScatter"); HCP_SYM(AllGather, "ncclAllGather");
        HCP_SYM(GetErrorString, "ncclGetErrorString");
#undef HCP_SYM
        return x;
    }();
    return &r;
}
int rccl_ready(const char* who) {
    const Rccl* r = rccl();
    if (!r->so) return hcp_set_error("%s: librccl.so not found (dlopen: %s)", who, dlerror());
    if (!r->GetUniqueId || !r->CommInitRank || !r->CommDestroy || !r->AllReduce || !r->ReduceScatter || !r->AllGather)
        return hcp_set_error("%s: librccl.so lacks a required symbol", who);
    return 0;
}
#define HCP_NCCL(call, who)                                                                                         \
    do {                                                                                                            \
        ncclResult_t r_ = (call);                                                                                   \
        if (r_ != ncclSuccess) return hcp_set_error("%s: RCCL error %d (%s)", who, (int)r_,                         \
                                                    rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "?");     \
    } while (0)

int to_nccl(int dtype, ncclDataType_t* out) {
    if (dtype == 0) { *out = ncclFloat32; return 0; }
    if (dtype == 1) { *out = ncclBfloat16; return 0; }
    return hcp_set_error("hcp comm: dtype %d unsupported (0 = fp32, 1 = bf16)", dtype);
}

}  // namespace

#define HCP_COMM_UNIQUE_ID_BYTES 128

// Rank 0 creates the rendezvous token (128 bytes) that every rank passes to hcp_comm_init.
HCP_API int hcp_comm_unique_id(void* out128) {
    HCP_REQUIRE(out128, "hcp_comm_unique_id: null pointer");
    if (int e = rccl_ready("hcp_comm_unique_id")) return e;
    static_assert(sizeof(ncclUniqueId) == HCP_COMM_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    HCP_NCCL(rccl()->GetUniqueId(&id), "hcp_comm_unique_id");
    memcpy(out128, &id, sizeof(id));
    return 0;
}

// Collective: all `world` ranks call it with the same token; the calling thread's current HIP device is the rank's GPU.
HCP_API int hcp_comm_init(int rank, int world, const void* unique_id128, void** comm_out) {
    HCP_REQUIRE(comm_out && unique_id128, "hcp_comm_init: null pointer");
    HCP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "hcp_comm_init: rank %d of %d", rank, world);
    HcpComm* c = (HcpComm*)calloc(1, sizeof(HcpComm));
    HCP_REQUIRE(c, "hcp_comm_init: out of host memory");
    c->rank = rank; c->world = world;
    if (int e = rccl_ready("hcp_comm_init")) { free(c); return e; }
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    ncclResult_t r = rccl()->CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        free(c);
        return hcp_set_error("hcp_comm_init: RCCL error %d (%s)", (int)r, rccl()->GetErrorString ? rccl()->GetErrorString(r) : "?");
    }
    *comm_out = c;
    return 0;
}

HCP_API int hcp_comm_destroy(void* comm) {
    if (!comm) return 0;
    HcpComm* c = (HcpComm*)comm;
    if (c->nccl) HCP_NCCL(rccl()->CommDestroy(c->nccl), "hcp_comm_destroy");
    free(c);
    return 0;
}

HCP_API int hcp_comm_rank(const void* comm) { return comm ? ((const HcpComm*)comm)->rank : -1; }
HCP_API int hcp_comm_world(const void* comm) { return comm ? ((const HcpComm*)comm)->world : -1; }

// buf[i] <- sum over ranks of buf[i], in place (DDP's gradient all-reduce; the 1/world factor lives in the optimizer kernel).
HCP_API int hcp_allreduce_flat(void* comm, void* buf, size_t count, int dtype, hipStream_t stream) {
    HCP_REQUIRE(comm && (buf || count == 0), "hcp_allreduce_flat: null pointer");
    HcpComm* c = (HcpComm*)comm;
    if (count == 0) return 0;
    ncclDataType_t dt;
    if (int e = to_nccl(dtype, &dt)) return e;
    HCP_NCCL(rccl()->AllReduce(buf, buf, count, dt, ncclSum, c->nccl, stream), "hcp_allreduce_flat");
    return 0;
}

// recv[0 .. recv_count) <- sum over ranks of send[rank*recv_count .. (rank+1)*recv_count); send holds world*recv_count elements.
HCP_API int hcp_reduce_scatter_flat(void* comm, const void* send, void* recv, size_t recv_count, int dtype, hipStream_t stream) {
    HCP_REQUIRE(comm && ((send && recv) || recv_count == 0), "hcp_reduce_scatter_flat: null pointer");
    HcpComm* c = (HcpComm*)comm;
    if (recv_count == 0) return 0;
    ncclDataType_t dt;
    if (int e = to_nccl(dtype, &dt)) return e;
    HCP_NCCL(rccl()->ReduceScatter(send, recv, recv_count, dt, ncclSum, c->nccl, stream), "hcp_reduce_scatter_flat");
    return 0;
}

// recv[r*send_count .. (r+1)*send_count) <- rank r's send[0 .. send_count); recv holds world*send_count elements
// (send may alias its own slot of recv: the in-place form RCCL recognises).
HCP_API int hcp_allgather_flat(void* comm, const void* send, void* recv, size_t send_count, int dtype, hipStream_t stream) {
    HCP_REQUIRE(comm && ((send && recv) || send_count == 0), "hcp_allgather_flat: null pointer");
    HcpComm* c = (HcpComm*)comm;
    if (send_count == 0) return 0;
    ncclDataType_t dt;
    if (int e = to_nccl(dtype, &dt)) return e;
    HCP_NCCL(rccl()->AllGather(send, recv, send_count, dt, c->nccl, stream), "hcp_allgather_flat");
    return 0;
}
