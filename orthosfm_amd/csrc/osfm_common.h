// Shared host helpers: error reporting and a grow-only device buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
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

// What the library itself holds (osfm_library_memory): every allocation path below books here, so that a handle
// that does not give its memory back shows as a number that stays up after its destroy -- whatever the runtime
// keeps in pools of its own (which the free-memory reading of hipMemGetInfo cannot tell apart from a leak).
struct MemoryLedger {
    std::atomic<long long> device_buffer_bytes{0};   // DeviceBuffer: grow-only device memory owned by live handles
    std::atomic<long long> pool_live_bytes{0};       // DevicePool blocks handed out (calls in flight)
    std::atomic<long long> pool_cached_bytes{0};     // DevicePool blocks kept for reuse
    std::atomic<long long> pinned_host_bytes{0};     // page-locked host staging
    std::atomic<int> live_matchers{0}, live_streams{0}, live_events{0};
};
inline MemoryLedger g_ledger;

inline hipError_t pinned_alloc(void **p, size_t bytes, unsigned flags)
{
    const hipError_t e = hipHostMalloc(p, bytes, flags);
    if (e == hipSuccess) g_ledger.pinned_host_bytes += (long long)bytes;
    return e;
}
inline void pinned_free(void *p, size_t bytes)
{
    if (!p) return;
    (void)hipHostFree(p);
    g_ledger.pinned_host_bytes -= (long long)bytes;
}
inline hipError_t stream_create(hipStream_t *s)
{
    const hipError_t e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    if (e == hipSuccess) g_ledger.live_streams++;
    return e;
}
inline void stream_destroy(hipStream_t s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    g_ledger.live_streams--;
}
inline hipError_t event_create(hipEvent_t *e, bool timing)
{
    const hipError_t r = timing ? hipEventCreate(e) : hipEventCreateWithFlags(e, hipEventDisableTiming);
    if (r == hipSuccess) g_ledger.live_events++;
    return r;
}
inline void event_destroy(hipEvent_t e)
{
    if (!e) return;
    (void)hipEventDestroy(e);
    g_ledger.live_events--;
}

struct DevicePool;
size_t device_pool_trim(int device);       // (defined behind DevicePool: DeviceBuffer retries through it)

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
        release();
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&ptr, want);
        if (e != hipSuccess) {
            // out of memory while the work-array cache of this device holds blocks nobody uses: hand them back, once
            (void)hipGetLastError();
            int device = 0;
            if (hipGetDevice(&device) == hipSuccess && device_pool_trim(device) > 0) e = hipMalloc(&ptr, want);
        }
        OSFM_HIP_CHECK(e);
        bytes = want;
        g_ledger.device_buffer_bytes += (long long)bytes;
        return OSFM_OK;
    }
    void release()
    {
        if (ptr) { (void)hipFree(ptr); g_ledger.device_buffer_bytes -= (long long)bytes; }
        ptr = nullptr; bytes = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
};

// Device memory of the short-lived work arrays of a call (bundle adjustment, triangulation,
// the filters: dozens of arrays per call, hundreds of calls per reconstruction).  hipMalloc /
// hipFree cost 50-100 us apiece -- hipFree also synchronises the device -- which was two
// thirds of a global-adjustment call inside the incremental reconstruction.  Blocks handed
// back are kept, by size class (steps of 1/8 of a power of two, at most 12.5 % slack), and
// given out again; osfm_trim_device_memory() returns them to the driver.  What is kept is
// bounded per device -- a sixth of the card, at most 48 GB (the work arrays of a 500-view global adjustment are
// ~6 GB; on a 288 GB card the bound is the 48) --; beyond it a block is freed at once.
struct DevicePool {
    static constexpr size_t kPoolKeepMax = (size_t)48 << 30;
    std::unordered_map<int, size_t> keep_limit;                             // per device, from hipMemGetInfo on first use
    size_t keep_bytes(int device)          // caller holds the mutex
    {
        auto it = keep_limit.find(device);
        if (it != keep_limit.end()) return it->second;
        size_t free_b = 0, total_b = 0, lim = kPoolKeepMax;
        int cur = 0;
        if (hipGetDevice(&cur) == hipSuccess && cur == device && hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b)
            lim = std::min(kPoolKeepMax, total_b / 6);
        keep_limit[device] = lim;
        return lim;
    }
    struct Block { void *ptr; size_t bytes; };
    std::mutex mutex;
    std::unordered_map<int, std::multimap<size_t, void *>> free_blocks;    // device -> size class -> blocks
    std::unordered_map<void *, std::pair<int, size_t>> live;               // block -> (device, size class)
    std::unordered_map<int, size_t> kept;                                   // bytes in free_blocks per device
    static size_t size_class(size_t bytes)
    {
        size_t b = bytes < 256 ? 256 : bytes;
        size_t p = 256;
        while (p < b) p <<= 1;                 // smallest power of two >= b
        const size_t step = p >> 4;             // sixteenths of it = eighths of the power below
        return step ? (b + step - 1) / step * step : p;
    }
    int alloc(void **out, size_t bytes)
    {
        int device = 0;
        OSFM_HIP_CHECK(hipGetDevice(&device));
        const size_t cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> lock(mutex);
            auto &fb = free_blocks[device];
            auto it = fb.find(cls);
            if (it != fb.end()) {
                *out = it->second;
                fb.erase(it);
                kept[device] -= cls;
                live[*out] = {device, cls};
                g_ledger.pool_cached_bytes -= (long long)cls; g_ledger.pool_live_bytes += (long long)cls;
                return OSFM_OK;
            }
        }
        hipError_t e = hipMalloc(out, cls);
        if (e != hipSuccess) {
            // out of memory with blocks of other sizes lying idle: hand them back and retry once
            (void)hipGetLastError();
            trim(device);
            e = hipMalloc(out, cls);
        }
        OSFM_HIP_CHECK(e);
        std::lock_guard<std::mutex> lock(mutex);
        live[*out] = {device, cls};
        g_ledger.pool_live_bytes += (long long)cls;
        return OSFM_OK;
    }
    void free(void *ptr)
    {
        if (!ptr) return;
        int device = -1;
        size_t cls = 0;
        {
            std::lock_guard<std::mutex> lock(mutex);
            auto it = live.find(ptr);
            if (it != live.end()) {
                device = it->second.first; cls = it->second.second;
                live.erase(it);
                g_ledger.pool_live_bytes -= (long long)cls;
                if (kept[device] + cls <= keep_bytes(device)) {
                    free_blocks[device].emplace(cls, ptr);
                    kept[device] += cls;
                    g_ledger.pool_cached_bytes += (long long)cls;
                    return;
                }
            }
        }
        (void)hipFree(ptr);
    }
    // frees what is kept for `device` (-1: every device); returns the bytes released
    size_t trim(int device)
    {
        std::vector<void *> victims;
        size_t bytes = 0;
        {
            std::lock_guard<std::mutex> lock(mutex);
            for (auto &dev : free_blocks) {
                if (device >= 0 && dev.first != device) continue;
                for (auto &b : dev.second) { victims.push_back(b.second); bytes += b.first; }
                dev.second.clear();
                kept[dev.first] = 0;
            }
        }
        for (void *p : victims) (void)hipFree(p);
        g_ledger.pool_cached_bytes -= (long long)bytes;
        return bytes;
    }
};
inline DevicePool g_device_pool;
inline size_t device_pool_trim(int device) { return g_device_pool.trim(device); }

// Blocks released while a stream lease is active on this thread may still be in use by
// work queued on that stream (an error path returns with kernels in flight); they go back
// to the pool only after the lease has drained its stream (StreamLease::~StreamLease).
struct StreamLease;
inline thread_local StreamLease *g_active_lease = nullptr;
inline void pool_release(void *ptr);

// DeviceBuffer's interface on pooled memory (grow-only, contents not preserved on growth)
struct PooledBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    PooledBuffer() = default;
    PooledBuffer(const PooledBuffer &) = delete;
    PooledBuffer &operator=(const PooledBuffer &) = delete;
    ~PooledBuffer() { release(); }
    int reserve(size_t need)
    {
        if (need <= bytes) return OSFM_OK;
        release();
        OSFM_RETURN_IF(g_device_pool.alloc(&ptr, need));
        bytes = DevicePool::size_class(need);
        return OSFM_OK;
    }
    void release()
    {
        if (ptr) pool_release(ptr);
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
        if (pinned) { pinned_free(pinned, pinned_bytes); pinned = nullptr; pinned_bytes = 0; }
        OSFM_HIP_CHECK(pinned_alloc(&pinned, bytes, hipHostMallocDefault));
        pinned_bytes = bytes;
        return OSFM_OK;
    }
    int ensure_events(size_t n)
    {
        while (events.size() < n) {
            hipEvent_t e = nullptr;
            OSFM_HIP_CHECK(event_create(&e, true));
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
    StreamLease *outer = nullptr;            // the lease this one nests in (same thread)
    std::vector<void *> pending_free;        // pool blocks released while this lease was active
    bool active = false;
    int acquire()
    {
        outer = g_active_lease; g_active_lease = this; active = true;
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
            hipError_t e = stream_create(&n->s);
            for (auto &p : n->ev) {
                if (e == hipSuccess) e = event_create(&p.a, true);
                if (e == hipSuccess) e = event_create(&p.b, true);
            }
            if (e != hipSuccess) {
                for (auto &p : n->ev) { event_destroy(p.a); event_destroy(p.b); }
                stream_destroy(n->s);
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
        if (set) (void)hipStreamSynchronize(set->s);  // nothing of this call may still run when the next one reuses it
        if (active) g_active_lease = outer;
        for (void *p : pending_free) g_device_pool.free(p);
        if (!set) return;
        std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
        g_stream_pool.push_back(set);
    }
};

inline void pool_release(void *ptr)
{
    if (g_active_lease) g_active_lease->pending_free.push_back(ptr);
    else g_device_pool.free(ptr);
}


}  // namespace osfm
