// host_api_stubs.cpp -- TEST CODE: stand-ins for the GPU half of libthfhe_hip.so, so that csrc/tfhe_host.cpp (libtfhe's host API: parameters,
// seeded key generation, encryption, files) can be linked into a CPU-only sanitizer build.  The keygen / host modes of libtfhe_client never call them.
#include <cstdlib>

#include "../../include/tfhe_shim.h"
#include "../../include/thfhe_hip.h"

extern "C" {
void thfhe_tfhe_forget_key(const TFheGateBootstrappingCloudKeySet *) {}
const char *thfhe_last_error(void) { return "GPU half not linked (sanitizer build)"; }
int thfhe_poly_ctx_create(int, int, thfhe_poly_ctx **) { return THFHE_E_NO_DEVICE; }
int thfhe_partial_decrypt(thfhe_poly_ctx *, const int32_t *, const int32_t *, const int32_t *, int32_t *, size_t) { return THFHE_E_NO_DEVICE; }
#define STUB_GATE2(NAME) \
    void NAME(LweSample *, const LweSample *, const LweSample *, const TFheGateBootstrappingCloudKeySet *) { std::abort(); }
STUB_GATE2(bootsAND)
STUB_GATE2(bootsXOR)
STUB_GATE2(bootsOR)
void bootsCOPY(LweSample *, const LweSample *, const TFheGateBootstrappingCloudKeySet *) { std::abort(); }
}
