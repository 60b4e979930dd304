// Shared host helpers: error reporting and a grow-only device buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

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
// Owns its memory: freed by the destructor (or release()); movable, not copyable.
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    DeviceBuffer(DeviceBuffer &&o) noexcept : ptr(o.ptr), bytes(o.bytes) { o.ptr = nullptr; o.bytes = 0; }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept
    {
        if (this != &o) { release(); ptr = o.ptr; bytes = o.bytes; o.ptr = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DeviceBuffer() { release(); }
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

// A stream and its timing events, leased for the duration of one call.  Creating a
// stream costs about a millisecond -- a real share of a 15 ms solve that the
// incremental reconstruction repeats for every camera group -- so the sets live in
// a per-process free list: a call takes one for its device (concurrent callers get
// different ones) and hands it back, drained, on every exit path.
struct EventPair { hipEvent_t a = nullptr, b = nullptr; };
struct StreamSet {
    int device = -1;
    hipStream_t s = nullptr;
    EventPair ev[4];
    // page-locked host scratch (the LM state lands here, one slot per iteration) and events
    // created on demand; both live as long as the set and are reused by later calls
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    std::vector<hipEvent_t> events;
    int ensure_pinned(size_t bytes)
    {
        if (bytes <= pinned_bytes) return OSFM_OK;
        if (pinned) { (void)hipHostFree(pinned); pinned = nullptr; pinned_bytes = 0; }
        OSFM_HIP_CHECK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
        pinned_bytes = bytes;
        return OSFM_OK;
    }
    int ensure_events(size_t n)
    {
        while (events.size() < n) {
            hipEvent_t e = nullptr;
            OSFM_HIP_CHECK(hipEventCreate(&e));
            events.push_back(e);
        }
        return OSFM_OK;
    }
};
inline std::mutex g_stream_pool_mutex;
inline std::vector<StreamSet *> g_stream_pool;

struct StreamLease {
    hipStream_t s = nullptr;
    EventPair *ev = nullptr;
    StreamSet *set = nullptr;
    int acquire()
    {
        int device = 0;
        OSFM_HIP_CHECK(hipGetDevice(&device));
        {
            std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
            for (size_t i = 0; i < g_stream_pool.size(); ++i)
                if (g_stream_pool[i]->device == device) {
                    set = g_stream_pool[i];
                    g_stream_pool.erase(g_stream_pool.begin() + (long)i);
                    break;
                }
        }
        if (!set) {
            StreamSet *n = new StreamSet;
            n->device = device;
            hipError_t e = hipStreamCreateWithFlags(&n->s, hipStreamNonBlocking);
            for (auto &p : n->ev) {
                if (e == hipSuccess) e = hipEventCreate(&p.a);
                if (e == hipSuccess) e = hipEventCreate(&p.b);
            }
            if (e != hipSuccess) {
                for (auto &p : n->ev) { if (p.a) (void)hipEventDestroy(p.a); if (p.b) (void)hipEventDestroy(p.b); }
                if (n->s) (void)hipStreamDestroy(n->s);
                delete n;
                OSFM_HIP_CHECK(e);
            }
            set = n;
        }
        s = set->s; ev = set->ev;
        return OSFM_OK;
    }
    ~StreamLease()
    {
        if (!set) return;
        (void)hipStreamSynchronize(set->s);       // nothing of this call may still run when the next one reuses it
        std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
        g_stream_pool.push_back(set);
    }
};


}  // namespace osfm
