// TEST-ONLY: the exchange entry points (csrc/comm.hip) for the CPU interpreter build — a world of ONE rank, no RCCL: collectives are
// copies.  Multi-rank behaviour of the trainer is covered through torch.distributed/gloo (tests/test_dist.py); this file only lets the
// C-ABI exchange path (comm.AbiComm, --comm abi) run its world = 1 form on the interpreter.
#include "hcp_common.h"
#include <stdlib.h>

namespace {
struct HcpComm { int rank, world; };
size_t dtype_bytes(int dtype) { return dtype == 1 ? 2 : 4; }
}  // namespace

HCP_API int hcp_comm_unique_id(void* out128) {
    HCP_REQUIRE(out128, "hcp_comm_unique_id: null pointer");
    memset(out128, 0, 128);
    return 0;
}
HCP_API int hcp_comm_init(int rank, int world, const void* unique_id128, void** comm_out) {
    HCP_REQUIRE(comm_out && unique_id128, "hcp_comm_init: null pointer");
    HCP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "hcp_comm_init: rank %d of %d", rank, world);
    HCP_REQUIRE(world == 1, "hcp_comm_init: the interpreter build has no RCCL (world must be 1)");
    HcpComm* c = (HcpComm*)calloc(1, sizeof(HcpComm));
    HCP_REQUIRE(c, "hcp_comm_init: out of host memory");
    c->rank = rank; c->world = world;
    *comm_out = c;
    return 0;
}
HCP_API int hcp_comm_destroy(void* comm) { free(comm); return 0; }
HCP_API int hcp_comm_rank(const void* comm) { return comm ? ((const HcpComm*)comm)->rank : -1; }
HCP_API int hcp_comm_world(const void* comm) { return comm ? ((const HcpComm*)comm)->world : -1; }
HCP_API int hcp_allreduce_flat(void* comm, void* buf, size_t count, int, hipStream_t) {
    HCP_REQUIRE(comm && (buf || count == 0), "hcp_allreduce_flat: null pointer");
    return 0;
}
HCP_API int hcp_reduce_scatter_flat(void* comm, const void* send, void* recv, size_t recv_count, int dtype, hipStream_t stream) {
    HCP_REQUIRE(comm && ((send && recv) || recv_count == 0), "hcp_reduce_scatter_flat: null pointer");
    if (recv_count && send != recv) hcp_memcpy_async(recv, send, recv_count * dtype_bytes(dtype), stream);
    return 0;
}
HCP_API int hcp_allgather_flat(void* comm, const void* send, void* recv, size_t send_count, int dtype, hipStream_t stream) {
    HCP_REQUIRE(comm && ((send && recv) || send_count == 0), "hcp_allgather_flat: null pointer");
    if (send_count && send != recv) hcp_memcpy_async(recv, send, send_count * dtype_bytes(dtype), stream);
    return 0;
}
