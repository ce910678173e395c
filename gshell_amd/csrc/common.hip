#include "common.hpp"
#include "../../include/gshell_hip.h"

namespace gs {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace gs

namespace gs {
static std::string& flag_registry() { static std::string r; return r; }       // function-local: usable from other files' static initialisers
void report_flag(const char* name, long long value, long long) {
    std::string& r = flag_registry();
    if (!r.empty()) r += " ";
    r += std::string(name) + "=" + std::to_string(value);
}
}  // namespace gs

namespace gs { GS_TUNABLE(GS_ORACLE_KERNELS, 0) }

extern "C" const char* gs_last_error(void) { return gs::g_err.c_str(); }
extern "C" const char* gs_build_flags(void) { return gs::flag_registry().c_str(); }
extern "C" int gs_version(void) { return 100; }

extern "C" int gs_memcpy_d2d(void* dst, const void* src, int64_t bytes, gs_stream_t stream) {
    if (bytes <= 0) return 0;
    GS_REQUIRE(dst && src, "gs_memcpy_d2d: null pointer");
    GS_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
