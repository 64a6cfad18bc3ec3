// libngp_hip.so runtime glue: thread-local error message, ABI/arch queries.
#include "common.h"
#include <string>

namespace ngp {
static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace ngp

extern "C" const char* ngp_last_error(void) { return ngp::g_last_error.c_str(); }
extern "C" int ngp_abi_version(void) { return 10; }
extern "C" const char* ngp_target_arch(void) { return "gfx950"; }
