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

// (HCP_LAUNCH_CHECK, hcp_memset_async, hcp_memcpy_async: hcp_device.h — the interpreter build has its own in tests/emu/hcp_emu.h)

static inline int hcp_cdiv(int a, int b) { return (a + b - 1) / b; }
typedef unsigned short hcp_bf16;  // raw bf16 bits
