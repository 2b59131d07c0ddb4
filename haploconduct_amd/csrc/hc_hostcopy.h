// hc_hostcopy.h — a large device-to-host copy into pageable memory that may be untouched: the runtime's copy into fresh pages runs at
// 13 - 16 GB/s, most of it the pages' first touch; here a few threads touch the destination one 32 MiB stretch ahead of the copy
// (383 MB of find-next-overlaps text: 26 -> 10 ms).  Touching writes a zero at the start of every page of a stretch BEFORE that stretch
// is copied over, so what the destination held does not matter and what it holds afterwards is the copy.
//
// Order (round 4, ADVICE: the round-3 form counted touches in ONE sum, so threads running ahead could release a stretch a slow thread
// had not touched yet, and that thread's zeros then landed on copied bytes): every stretch has its OWN counter; a touching thread
// adds to stretch k's counter after its last write into k, the copier starts k only when that counter holds every thread.  No thread
// ever writes into a stretch after adding to its counter, so nothing is written behind the copy.
// `copy_touched_ahead` is free of HIP so that the ordering can be tested on the CPU with memcpy as the copy
// (tests/test_hostcopy.py, under taskset and ThreadSanitizer).
#pragma once
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <memory>
#include <thread>
#include <vector>

namespace hc {

// copy(dst_bytes, offset, len) -> 0 on success; called for consecutive stretches in order, from the calling thread.
template <class Copy>
inline int copy_touched_ahead(void* dst, uint64_t bytes, uint64_t chunk, unsigned T, Copy copy) {
    if (!bytes) return 0;
    char* h = (char*)dst;
    if (T < 1) T = 1;
    const uint64_t n_chunks = (bytes + chunk - 1) / chunk;
    std::unique_ptr<std::atomic<uint32_t>[]> done(new std::atomic<uint32_t>[n_chunks]);
    for (uint64_t k = 0; k < n_chunks; k++) done[k].store(0, std::memory_order_relaxed);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t] {
            for (uint64_t k = 0; k < n_chunks; k++) {
                const uint64_t b = k * chunk, len = std::min(bytes, b + chunk) - b;
                volatile char* p = h + b;
                const uint64_t pages = (len + 4095) / 4096;  // dealt to the threads whole: no byte has two writers
                for (uint64_t pg = pages * t / T; pg < pages * (t + 1) / T; pg++) p[pg * 4096] = 0;
                done[k].fetch_add(1, std::memory_order_release);  // this thread's last write into stretch k is behind it
            }
        });
    int err = 0;
    for (uint64_t k = 0; k < n_chunks && err == 0; k++) {
        while (done[k].load(std::memory_order_acquire) < T) std::this_thread::yield();
        const uint64_t b = k * chunk;
        err = copy(h + b, b, std::min(bytes, b + chunk) - b);
    }
    for (auto& x : th) x.join();
    return err;
}

}  // namespace hc

#ifndef HC_HOSTCOPY_NO_HIP
#include <hip/hip_runtime.h>

namespace hc {

inline hipError_t copy_to_pageable_host(void* dst, const void* src, uint64_t bytes) {
    const uint64_t chunk = (uint64_t)32 << 20;
    if (bytes < 2 * chunk) return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) : hipSuccess;
    {  // 2 MiB pages where the system grants them and the range is still untouched: a hint, harmless otherwise
        const uintptr_t huge = (uintptr_t)2 << 20, b = ((uintptr_t)dst + huge - 1) & ~(huge - 1), e = ((uintptr_t)dst + bytes) & ~(huge - 1);
        if (e > b) (void)madvise((void*)b, e - b, MADV_HUGEPAGE);
    }
    unsigned T = std::thread::hardware_concurrency();
    T = T > 8 ? 8 : (T ? T : 1);
    const char* d = (const char*)src;
    return (hipError_t)copy_touched_ahead(dst, bytes, chunk, T, [d](char* to, uint64_t off, uint64_t len) {
        return (int)hipMemcpy(to, d + off, len, hipMemcpyDeviceToHost);
    });
}

}  // namespace hc
#endif
