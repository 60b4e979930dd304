// Shared host helpers: error reporting and a grow-only device buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/osfm_hip.h"

namespace osfm {

void set_error(const char *fmt, ...);

#define OSFM_HIP_CHECK(expr)                                                        \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            ::osfm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                __FILE__, __LINE__);                                                \
            return OSFM_E_DEVICE;                                                   \
        }                                                                           \
    } while (0)

#define OSFM_RETURN_IF(expr)      \
    do {                          \
        int _s = (expr);          \
        if (_s != OSFM_OK) return _s; \
    } while (0)

// Device allocation that only ever grows; contents are NOT preserved on growth.
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t need)
    {
        if (need <= bytes) return OSFM_OK;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; bytes = 0; }
        size_t want = need + need / 8 + 256;
        OSFM_HIP_CHECK(hipMalloc(&ptr, want));
        bytes = want;
        return OSFM_OK;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; bytes = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
};

}  // namespace osfm
