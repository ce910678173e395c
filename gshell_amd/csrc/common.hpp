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

// ---- compile-time variants --------------------------------------------------------------------------------------------------------------------------
// Every tunable of a kernel file is declared as
//     #ifndef GS_FOO
//     #define GS_FOO 3
//     #endif
//     GS_TUNABLE(GS_FOO, 3)
// and registers itself at load time when the library was built with another value (tools/build_variant.sh -DGS_FOO=5): gs_build_flags() then
// lists "GS_FOO=5", and bench.py refuses to print a number for such a library unless told that a variant is being measured.  Switches that give
// WRONG results (timing-only ablations) compile only under -DGS_EXPERIMENT, which is itself reported.  The shipped target (csrc/Makefile) accepts
// no extra defines at all.
void report_flag(const char* name, long long value, long long dflt);
struct FlagReporter {
    FlagReporter(const char* name, long long value, long long dflt) { if (value != dflt) report_flag(name, value, dflt); }
};
#if defined(__HIP_DEVICE_COMPILE__)
#define GS_TUNABLE(name, dflt)
#define GS_TUNABLE_F(name, dflt)
#else
#define GS_TUNABLE(name, dflt) static const ::gs::FlagReporter gs_flag_reporter_##name(#name, (long long)(name), (long long)(dflt));
#define GS_TUNABLE_F(name, dflt) static const ::gs::FlagReporter gs_flag_reporter_##name(#name " x 1e6", (long long)((name) * 1e6), (long long)((dflt) * 1e6));
#endif
// ORACLE / ALTERNATE-DESIGN kernels: correct kernels that the shipped iteration never launches -- the exact-fp32 SDF forward of round 1 (csrc/mlp.hip), the
// register-resident one-product forward (k_h1r_fwd), the tangent-row eikonal instantiations (<EIK>) and the sampler-replay shading backward
// (k_shade_samples<true>).  They are what several tests compare the shipped kernels with, and they are the measured records of designs that were not
// adopted; they are NOT in libgshell_hip.so (one design per stage in the library bench.py certifies).  `make` builds a second library,
// lib/variants/oracles.so, from the same sources with -DGS_ORACLE_KERNELS=1 (reported by gs_build_flags(), refused by bench.py); tests reach it
// through gshell_amd._lib.use_variant("oracles").  In the shipped build their entry points exist and fail with this message.
#ifndef GS_ORACLE_KERNELS
#define GS_ORACLE_KERNELS 0
#endif
#define GS_ORACLE_ONLY(what) GS_REQUIRE(false, what ": oracle / alternate-design kernel, not in the shipped library -- use gshell_amd/lib/variants/oracles.so (gshell_amd._lib.use_variant)")

#ifdef GS_EXPERIMENT
#define GS_EXPERIMENT_ONLY(name)
#if !defined(__HIP_DEVICE_COMPILE__)
static const ::gs::FlagReporter gs_flag_reporter_experiment("GS_EXPERIMENT(" __BASE_FILE__ ")", 1, 0);      // one entry per .hip built that way (__BASE_FILE__: the translation unit, not this header)
#endif
#else
#define GS_EXPERIMENT_ONLY(name) static_assert(false, #name " gives wrong results (timing-only ablation): it compiles only with -DGS_EXPERIMENT");
#endif

}  // namespace gs
