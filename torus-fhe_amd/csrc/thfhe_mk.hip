// thfhe_mk.hip -- 3-gen multi-key (Torus64 ring) path.  Placeholder: entry points exist so the ABI is complete,
// and report THFHE_E_UNSUPPORTED until the Torus64 kernels land.
#include <hip/hip_runtime.h>

#include "../../include/thfhe_hip.h"
#include "thfhe_common.h"

using namespace thfhe;

struct thfhe_mk_ctx {
    thfhe_params p;
};

extern "C" {
int thfhe_mk_ctx_create(const thfhe_params *, const int64_t *, const int32_t *, int, thfhe_mk_ctx **out) {
    if (out) *out = nullptr;
    return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key (Torus64) path not implemented yet");
}
void thfhe_mk_ctx_destroy(thfhe_mk_ctx *c) { delete c; }
int thfhe_mk_gates(thfhe_mk_ctx *, int, const int32_t *, const int32_t *, const int32_t *, int32_t *, size_t) { return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key path not implemented yet"); }
int thfhe_mk_bootstrap(thfhe_mk_ctx *, int64_t, const int32_t *, int32_t *, size_t) { return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key path not implemented yet"); }
void *thfhe_mk_dev_alloc(thfhe_mk_ctx *, size_t) { return nullptr; }
void thfhe_mk_dev_free(thfhe_mk_ctx *, void *) {}
int thfhe_mk_copy_h2d(thfhe_mk_ctx *, void *, const void *, size_t) { return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key path not implemented yet"); }
int thfhe_mk_copy_d2h(thfhe_mk_ctx *, void *, const void *, size_t) { return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key path not implemented yet"); }
int thfhe_mk_reserve(thfhe_mk_ctx *, size_t) { return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key path not implemented yet"); }
int thfhe_mk_gates_dev(thfhe_mk_ctx *, int, const int32_t *, const int32_t *, const int32_t *, int32_t *, size_t) { return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key path not implemented yet"); }
int thfhe_mk_sync(thfhe_mk_ctx *) { return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key path not implemented yet"); }
int thfhe_mk_set_profiling(thfhe_mk_ctx *, int) { return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key path not implemented yet"); }
int thfhe_mk_last_timings(thfhe_mk_ctx *, float *) { return thfhe_fail(THFHE_E_UNSUPPORTED, "multi-key path not implemented yet"); }
}
