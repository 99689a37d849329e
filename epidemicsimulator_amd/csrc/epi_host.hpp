// Host-side helpers shared by the translation units of the engine: error reporting, device selection and
// owning device / page-locked allocations.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../include/epi.h"

namespace epi {

// Records code and message as the calling thread's last error (read back by epi_last_error) and returns code.
int32_t fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(EPI_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline int32_t use_device(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(EPI_ERR_DEVICE, "no HIP device available: the engine has no CPU fallback");
    if (device < 0 || device >= count) return fail(EPI_ERR_BAD_ARGUMENT, "device %d out of range [0,%d)", device, count);
    HIP_TRY(hipSetDevice(device));
    return EPI_OK;
}

// Owning device allocation.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t want) {
        if (want <= bytes && p) return hipSuccess;
        release();
        if (want == 0) want = 16;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    template <class T>
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = ensure(v.size() * sizeof(T));
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

// Owning page-locked host allocation, mapped into the device's address space: kernels read and write it
// directly over PCIe through `dev` (staging for the per-step transfers of the batched worlds).
struct PinBuf {
    void* p = nullptr;
    void* dev = nullptr;
    size_t bytes = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    hipError_t ensure(size_t want) {
        if (want <= bytes && p) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = dev = nullptr; bytes = 0;
        if (want == 0) want = 16;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped);
        if (e != hipSuccess) return e;
        e = hipHostGetDevicePointer(&dev, p, 0);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
    template <class T>
    T* on_device() const { return static_cast<T*>(dev); }
};

}  // namespace epi
