// runtime.hip — error reporting and library identification for libhcp_mi355x.so.
#include "hcp_common.h"

static thread_local char g_err[512] = "";

extern "C" int hcp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

// Thread-local message of the last failing call on this thread ("" if none).
HCP_API const char* hcp_last_error(void) { return g_err; }

// 1 when this object was built by tests/emu (CPU interpreter), 0 for the gfx950 product library.
HCP_API int hcp_is_emulated(void) { return HCP_IS_EMULATED; }

HCP_API int hcp_abi_version(void) { return 3; }   // include/hcp_mi355x.h HCP_ABI_VERSION; _lib.py refuses any other value

// ---- device self-check of the fp32 atomic path (round 6) ----------------------------------------------------------------------------
// Every kernel of the library that still ADDS across workgroups (the loss and sum-of-squares scalars, bias / norm-affine column sums,
// the query-split dK / dV of the 77-key cross-attention) does so with global_atomic_add_f32.  Round 5 logged one box on which exactly
// those results were wrong by 2-70 % for the length of one process (profiles/r5_gpu_tests_run_with_9_failures.txt).  This entry point
// makes that condition detectable in a millisecond: `workgroups` x 256 threads add small INTEGERS (exact in fp32 in any order) into one
// 64-byte line (all workgroups on all XCDs hit the same 16 words) and into `nb` words `stride` floats apart (workgroup- and
// thread-dependent targets); the caller compares with the closed-form sums — __graft_entry__.smoke(), bench.py and the GPU tests call it
// through kernels.atomics_selfcheck() and stop on any difference.
namespace {
HCP_KERNEL(256) atomics_selfcheck_kernel(float* line, float* bucket, int nb, int stride) {
    const int tid = threadIdx.x;
    const float v = (float)((tid & 3) + 1);
    hcp_atomic_add(line + (tid & 15), v);
    const int idx = (int)(((long)blockIdx.x * 37 + (long)tid * 101) % nb);
    hcp_atomic_add(bucket + (size_t)idx * stride, v);
}
}  // namespace

// line: 16 floats; bucket: nb * stride floats; both are cleared here (fill kernel) and then receive workgroups * 256 adds each.
// Expected: line[j] = workgroups * 16 * ((j & 3) + 1); bucket[i * stride] = sum of ((t & 3) + 1) over (w, t) with (37 w + 101 t) % nb == i.
HCP_API int hcp_selfcheck_atomics(float* line, float* bucket, int nb, int stride, int workgroups, hipStream_t stream) {
    HCP_REQUIRE(line && bucket && nb > 0 && stride > 0 && workgroups > 0 && workgroups <= 4096, "hcp_selfcheck_atomics: bad arguments");
    if (hcp_memset_async(line, 0, 16 * sizeof(float), stream) != 0 || hcp_memset_async(bucket, 0, (size_t)nb * stride * sizeof(float), stream) != 0)
        return hcp_set_error("hcp_selfcheck_atomics: clearing the targets failed");
    HCP_LAUNCH(atomics_selfcheck_kernel, dim3(workgroups), dim3(256), 0, stream, line, bucket, nb, stride);
    HCP_LAUNCH_CHECK("atomics_selfcheck");
}
