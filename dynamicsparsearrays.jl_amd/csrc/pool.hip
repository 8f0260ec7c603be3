// csrc/pool.hip — a small caching allocator for HBM blocks (host code).
//
// hipMalloc / hipFree of the buffers a bulk build needs (hundreds of MB of scratch, the slot buffers of the new structure) cost
// milliseconds per build: more than the kernels (~4 ms of allocation beside 4.5 ms of K-build at 10 M triples).
// With 288 GB of HBM per MI355X, memory is not the scarce resource; launches and driver calls are.  Freed blocks are kept, by
// device and size class, and handed out again; beyond DSA_POOL_MAX_MB (default 16384) of idle blocks the largest are returned
// to the driver.  A block returned to the pool must not be referenced by work still in flight: callers synchronise the
// stream that used it first (the builders and pma_destroy do).
#include "dsa_dev.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace dsa {
namespace {

struct Pool {
    std::mutex mu;
    std::multimap<size_t, void*> idle[PerDeviceOnce::MAX_DEV];      // size class -> block
    std::unordered_map<void*, std::pair<int, size_t>> live;         // block -> (device, size class)
    size_t idle_bytes = 0;
    size_t max_idle = [] {
        const char* e = getenv("DSA_POOL_MAX_MB");
        return (size_t)(e ? atoll(e) : 16384) << 20;
    }();
};
Pool& pool() { static Pool* p = new Pool(); return *p; }      // never destroyed: blocks may be freed by late finalisers

// size classes: powers of two split in 8 steps (<= 12.5 % over-allocation), at least 4 KB
size_t size_class(size_t bytes) {
    size_t c = (size_t)4 << 10;
    while (c < bytes) c <<= 1;
    if (c == bytes || c <= ((size_t)64 << 10)) return c;
    const size_t half = c >> 1, step = half >> 3;
    return half + ((bytes - half + step - 1) / step) * step;
}

}  // namespace

hipError_t pool_alloc(void** out, size_t bytes) {
    *out = nullptr;
    if (bytes == 0) bytes = 1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= PerDeviceOnce::MAX_DEV) return hipMalloc(out, bytes);
    const size_t cls = size_class(bytes);
    Pool& P = pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.idle[dev].find(cls);
        if (it != P.idle[dev].end()) {
            *out = it->second;
            P.idle[dev].erase(it);
            P.idle_bytes -= cls;
            P.live[*out] = {dev, cls};
            return hipSuccess;
        }
    }
    e = hipMalloc(out, cls);
    if (e != hipSuccess) {
        // out of memory with idle blocks around: give them back and try once more
        pool_trim(0);
        (void)hipGetLastError();
        e = hipMalloc(out, cls);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(P.mu);
    P.live[*out] = {dev, cls};
    return hipSuccess;
}

void pool_free(void* p) {
    if (p == nullptr) return;
    Pool& P = pool();
    bool trim = false;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.live.find(p);
        if (it == P.live.end()) { (void)hipFree(p); return; }      // not ours (allocated before the pool was used)
        const int dev = it->second.first;
        const size_t cls = it->second.second;
        P.live.erase(it);
        P.idle[dev].insert({cls, p});
        P.idle_bytes += cls;
        trim = P.idle_bytes > P.max_idle;
    }
    if (trim) pool_trim(pool().max_idle / 2);
}

// returns idle blocks to the driver, largest first, until at most `keep_bytes` stay cached
void pool_trim(size_t keep_bytes) {
    Pool& P = pool();
    while (true) {
        void* victim = nullptr;
        int vdev = -1;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            if (P.idle_bytes <= keep_bytes) return;
            size_t best = 0;
            for (int d = 0; d < PerDeviceOnce::MAX_DEV; ++d)
                if (!P.idle[d].empty() && P.idle[d].rbegin()->first >= best) { best = P.idle[d].rbegin()->first; vdev = d; }
            if (vdev < 0) return;
            auto it = std::prev(P.idle[vdev].end());
            victim = it->second;
            P.idle_bytes -= it->first;
            P.idle[vdev].erase(it);
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != vdev) (void)hipSetDevice(vdev);
        (void)hipFree(victim);
        if (cur != vdev) (void)hipSetDevice(cur);
    }
}

// ---- pinned host blocks and streams: creating a handle (a control block mirror, a few words of landing area, a stream per
// orientation) cost ~0.5 ms in driver calls; both are kept when a handle dies ----------------------------------------------
namespace {
struct PinPool { std::mutex mu; std::multimap<size_t, void*> idle; std::unordered_map<void*, size_t> live; };
PinPool& pin_pool() { static PinPool* p = new PinPool(); return *p; }
struct StreamPool { std::mutex mu; std::vector<hipStream_t> idle[PerDeviceOnce::MAX_DEV]; };
StreamPool& stream_pool() { static StreamPool* p = new StreamPool(); return *p; }
}  // namespace

hipError_t pinned_alloc(void** out, size_t bytes) {
    const size_t cls = size_class(bytes ? bytes : 1);
    PinPool& P = pin_pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.idle.find(cls);
        if (it != P.idle.end()) { *out = it->second; P.idle.erase(it); P.live[*out] = cls; return hipSuccess; }
    }
    hipError_t e = hipHostMalloc(out, cls, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(P.mu);
    P.live[*out] = cls;
    return hipSuccess;
}
void pinned_free(void* p) {
    if (p == nullptr) return;
    PinPool& P = pin_pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.live.find(p);
        if (it != P.live.end()) {
            const size_t cls = it->second;
            P.live.erase(it);
            if (cls <= ((size_t)1 << 20) && P.idle.size() < 256) { P.idle.insert({cls, p}); return; }      // big staging areas go back to the driver
        }
    }
    (void)hipHostFree(p);
}
// a non-blocking stream of the current device; stream_put: the caller has synchronised it
hipError_t stream_get(hipStream_t* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < PerDeviceOnce::MAX_DEV) {
        StreamPool& S = stream_pool();
        std::lock_guard<std::mutex> lk(S.mu);
        if (!S.idle[dev].empty()) { *out = S.idle[dev].back(); S.idle[dev].pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void stream_put(hipStream_t s, int dev) {
    if (s == nullptr) return;
    if (dev >= 0 && dev < PerDeviceOnce::MAX_DEV) {
        StreamPool& S = stream_pool();
        std::lock_guard<std::mutex> lk(S.mu);
        if (S.idle[dev].size() < 64) { S.idle[dev].push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}

size_t pool_idle_bytes() {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    return P.idle_bytes;
}

}  // namespace dsa
