// Context, error reporting, version.
#include "common.hpp"

namespace spsp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace spsp

extern "C" {

const char* spsp_last_error(void) { return spsp::g_err; }
const char* spsp_version(void) { return "spsp 0.1 (gfx950)"; }

int spsp_init(int device, spsp_ctx** out) {
    SPSP_REQUIRE(out != nullptr, "spsp_init: out is NULL");
    int n = 0;
    SPSP_HIP_CHECK(hipGetDeviceCount(&n));
    SPSP_REQUIRE(device >= 0 && device < n, "spsp_init: device %d out of range (%d visible)", device, n);
    SPSP_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    SPSP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    spsp_ctx* c = new spsp_ctx;
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    *out = c;
    return SPSP_OK;
}

void spsp_destroy(spsp_ctx* ctx) { delete ctx; }

}  // extern "C"
