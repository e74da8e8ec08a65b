// capi_common.hip -- library-wide pieces of the C ABI: error slot, version, device probe.
#include "host_common.h"

namespace nnpops {
std::string& last_error_slot() {
    static thread_local std::string slot;
    return slot;
}
}  // namespace nnpops

extern "C" {

const char* nnpops_last_error(void) { return nnpops::last_error_slot().c_str(); }

// NNPOPS_SOURCE_HASH: sha256 (first 16 hex digits) of the sources this binary was compiled from, passed by
// nnpops_amd/build.py; the Python loader recomputes it from the tree and refuses a stale binary.
#ifndef NNPOPS_SOURCE_HASH
#define NNPOPS_SOURCE_HASH "unknown"
#endif
const char* nnpops_version(void) { return "nnpops_hip 0.2.0 gfx950 src:" NNPOPS_SOURCE_HASH; }

int nnpops_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return nnpops::fail(NNPOPS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

}  // extern "C"
