// Shared host-side helpers for libgshell_hip (error convention, launch geometry).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

namespace gs {

void set_error(const std::string& msg);

#define GS_HIP_CHECK(call)                                                              \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            ::gs::set_error(std::string(#call) + " failed: " + hipGetErrorString(_e) +  \
                            " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")");    \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

#define GS_REQUIRE(cond, msg)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ::gs::set_error(std::string(msg) + " [" #cond "]");                         \
            return 2;                                                                   \
        }                                                                               \
    } while (0)

#define GS_LAUNCH_CHECK() GS_HIP_CHECK(hipGetLastError())

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace gs
