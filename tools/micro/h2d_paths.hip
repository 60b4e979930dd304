// Host-to-device copy of a pageable buffer: plain hipMemcpyAsync vs hipHostRegister + copy + unregister
// vs chunked copy through a persistent pinned staging buffer filled by several threads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 12 << 20;
    std::vector<char> host(n, 1);
    void *dev; (void)hipMalloc(&dev, n);
    hipStream_t s; (void)hipStreamCreate(&s);
    void *pin; (void)hipHostMalloc(&pin, n);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        (void)hipMemcpyAsync(dev, host.data(), n, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s);
        double t1 = now();
        (void)hipHostRegister(host.data(), n, hipHostRegisterDefault);
        double t1b = now();
        (void)hipMemcpyAsync(dev, host.data(), n, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s);
        double t1c = now();
        (void)hipHostUnregister(host.data());
        double t2 = now();
        // staged: 4 threads copy quarters into the pinned buffer, each quarter sent as soon as it is there
        const int T = 4; std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { memcpy((char *)pin + t * (n / T), host.data() + t * (n / T), n / T); });
        for (int t = 0; t < T; ++t) { th[t].join(); (void)hipMemcpyAsync((char *)dev + t * (n / T), (char *)pin + t * (n / T), n / T, hipMemcpyHostToDevice, s); }
        (void)hipStreamSynchronize(s);
        double t3 = now();
        memcpy(pin, host.data(), n);
        double t4 = now();
        (void)hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s);
        double t5 = now();
        printf("12 MB: pageable %.2f ms | register %.2f + copy %.2f + unregister %.2f | 4-thread staged %.2f | memcpy %.2f + pinned copy %.2f\n",
            t1 - t0, t1b - t1, t1c - t1b, t2 - t1c, t3 - t2, t4 - t3, t5 - t4);
    }
    return 0;
}
