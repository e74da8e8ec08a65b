// HipStaging.h -- host/device pointer staging shared by the reference-side Hip* classes.
//
// The reference's core API (src/ani/ANISymmetryFunctions.h:66-86, src/schnet/CFConv.h:57-189) takes raw float
// pointers that its CUDA classes accept from EITHER memory space (every transfer is cudaMemcpyDefault,
// src/ani/CudaANISymmetryFunctions.cu:335-405); the C ABI of libnnpops_hip.so takes device pointers only.
// These helpers give the Hip* classes the reference's behaviour: device pointers are used in place, host
// pointers go through a device buffer owned by the object.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <stdexcept>
#include <string>

#include "nnpops_hip.h"

namespace nnpops_integration {

inline void hipTry(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
inline void abiTry(int code) {
    if (code != NNPOPS_OK) throw std::runtime_error(nnpops_last_error());
}

inline bool isDevicePointer(const void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();            // an unregistered host pointer: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice;
}

// A device buffer that stands in for a caller's array when that array lives on the host.
class DeviceMirror {
public:
    ~DeviceMirror() { if (buf) (void)hipFree(buf); }
    // device address holding the caller's data (copied in when `p` is a host pointer)
    const float* in(const float* p, size_t count) {
        if (p == nullptr || isDevicePointer(p)) return p;
        reserve(count);
        hipTry(hipMemcpy(buf, p, count * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy H2D");
        return buf;
    }
    // device address to compute into; call finish() afterwards
    float* out(float* p, size_t count) {
        host = nullptr;
        if (p == nullptr || isDevicePointer(p)) return p;
        reserve(count);
        host = p; host_count = count;
        return buf;
    }
    void finish() {
        if (host) hipTry(hipMemcpy(host, buf, host_count * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy D2H");
        host = nullptr;
    }
private:
    void reserve(size_t count) {
        if (count <= capacity) return;
        if (buf) (void)hipFree(buf);
        hipTry(hipMalloc((void**)&buf, count * sizeof(float)), "hipMalloc");
        capacity = count;
    }
    float* buf = nullptr;
    size_t capacity = 0;
    float* host = nullptr;
    size_t host_count = 0;
};

}  // namespace nnpops_integration
