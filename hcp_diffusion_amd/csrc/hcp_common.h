// hcp_common.h — C-ABI conventions shared by every translation unit of libhcp_mi355x.so.
// Every exported function returns int (0 ok, <0 error, message via hcp_last_error()),
// takes raw device pointers + explicit shapes + a hipStream_t, never allocates and never
// synchronises the stream (SURVEY.md §8(b) "What a C-ABI replacement must export").
#pragma once
#include "hcp_device.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#define HCP_API extern "C" __attribute__((visibility("default")))

// Tuning / ablation hooks (hcp_debug_*) are process-global knobs for tools and variant-coverage tests.  They exist only in
// builds made with -DHCP_TOOLS (libhcp_mi355x_tools.so, the interpreter build of tests/emu): in the product library every
// knob is a compile-time constant — no mutable global state behind the ABI, and the ablation branches fold away.
#if defined(HCP_TOOLS)
#define HCP_TUNABLE(type, name, value) type name = value
#else
#define HCP_TUNABLE(type, name, value) constexpr type name = value
#endif

extern "C" int hcp_set_error(const char* fmt, ...);

#define HCP_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) return hcp_set_error(__VA_ARGS__); \
    } while (0)

#if defined(HCP_EMU)
#define HCP_LAUNCH_CHECK(name) return 0
static inline int hcp_memset_async(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline int hcp_memcpy_async(void* d, const void* s, size_t n, hipStream_t) { memmove(d, s, n); return 0; }
#else
#define HCP_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) return hcp_set_error("%s: %s", name, hipGetErrorString(e_)); \
        return 0;                                                                     \
    } while (0)
static inline int hcp_memset_async(void* p, int v, size_t n, hipStream_t s) {
    return hipMemsetAsync(p, v, n, s) == hipSuccess ? 0 : -1;
}
static inline int hcp_memcpy_async(void* d, const void* src, size_t n, hipStream_t s) {
    return hipMemcpyAsync(d, src, n, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -1;
}
#endif

static inline int hcp_cdiv(int a, int b) { return (a + b - 1) / b; }
typedef unsigned short hcp_bf16;  // raw bf16 bits
