// thfhe_errors.cpp -- thread-local error message of the C ABI (include/thfhe_hip.h: thfhe_last_error).
#include <hip/hip_runtime.h>

#include <string>

#include "thfhe_common.h"

namespace {
thread_local std::string g_last_error;
}

namespace thfhe {
int thfhe_fail(int code, const char *msg) {
    g_last_error = msg ? msg : "";
    return code;
}
int thfhe_fail_hip(hipError_t e, const char *what) {
    g_last_error = std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ") in " + (what ? what : "?");
    return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? THFHE_E_NO_DEVICE : (e == hipErrorOutOfMemory ? THFHE_E_NOMEM : THFHE_E_HIP);
}
}  // namespace thfhe

extern "C" const char *thfhe_last_error(void) { return g_last_error.c_str(); }
