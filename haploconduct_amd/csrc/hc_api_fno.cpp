// hc_api_fno.cpp — host side of find-next-overlaps' device form (hc_fno_items.h): buffers, the launch sequence of
// hc_fno_kernels.hip, the four stable radix sorts, the copy of the text.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hcedge.h"
#include "hc_fno_device.h"
#include "hc_overlap_finder.h"
#include "host/Types.h"

namespace hc {
namespace {

struct DeviceBuffers {  // freed on every way out
    std::vector<void*> all;
    ~DeviceBuffers() {
        for (void* p : all) (void)hipFree(p);
    }
    template <typename T>
    T* get(uint64_t count) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<uint64_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) throw FatalError{HC_ERR_NOMEM, std::string("find-next-overlaps on the device: hipMalloc: ") + hipGetErrorString(e)};
        all.push_back(p);
        return (T*)p;
    }
};
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw FatalError{HC_ERR_HIP, std::string("find-next-overlaps on the device: ") + what + ": " + hipGetErrorString(e)};
}

}  // namespace

bool fno_device_wanted(uint64_t n_items) {
    const char* e = getenv("HC_FNO");
    if (e && strcmp(e, "host") == 0) return false;
    const bool forced = e && strcmp(e, "device") == 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        (void)hipGetLastError();
        if (forced) throw FatalError{HC_ERR_NO_DEVICE, "HC_FNO=device: no HIP device"};
        return false;
    }
    return forced || n_items >= 200000;
}

bool fno_lines_on_device(const FnoItem* items, uint64_t n, bool no_inclusions, const std::function<char*(uint64_t)>& text_of,
                         uint64_t counters[5], double* seconds) {
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    hipStream_t st = nullptr;  // the default stream of the calling thread's device
    DeviceBuffers d;
    FnoItem* d_items = d.get<FnoItem>(n);
    FnoRec* d_rec = d.get<FnoRec>(n);
    uint64_t* d_k[4];
    for (auto& k : d_k) k = d.get<uint64_t>(n);
    uint64_t *d_ka = d.get<uint64_t>(n + 1), *d_kb = d.get<uint64_t>(n + 1);
    uint32_t *d_pa = d.get<uint32_t>(n), *d_pb = d.get<uint32_t>(n);
    unsigned long long* d_counters = d.get<unsigned long long>(kFnoCounters);
    size_t tmp_bytes = 0, scan_bytes = 0;
    hip_check(sort_pairs_u64_u32(nullptr, tmp_bytes, d_ka, d_kb, d_pa, d_pb, (uint32_t)n, 64, st), "sort size");
    hip_check(finder_scan(nullptr, scan_bytes, d_ka, d_kb, n + 1, st), "scan size");
    tmp_bytes = std::max(tmp_bytes, scan_bytes);
    void* d_tmp = d.get<char>(tmp_bytes);
    hip_check(hipMemcpyAsync(d_items, items, n * sizeof(FnoItem), hipMemcpyHostToDevice, st), "copy of the combinations");
    hip_check(hipMemsetAsync(d_counters, 0, kFnoCounters * sizeof(unsigned long long), st), "memset");
    hip_check(fno_deduce(d_items, n, no_inclusions ? 1u : 0u, d_rec, d_k[0], d_k[1], d_k[2], d_k[3], d_pa, d_counters, st), "deduce");
    // least significant 64 bits first; every sort is stable
    uint32_t *perm = d_pa, *perm_next = d_pb;
    for (int c = 0; c < 4; c++) {
        const uint64_t* keys = d_k[c];
        if (c) {
            hip_check(fno_gather_keys(d_k[c], perm, n, d_ka, st), "gather");
            keys = d_ka;
        }
        size_t b = tmp_bytes;
        hip_check(sort_pairs_u64_u32(d_tmp, b, keys, d_kb, perm, perm_next, (uint32_t)n, 64, st), "sort");
        std::swap(perm, perm_next);
    }
    hip_check(fno_mark_lines(d_rec, perm, n, d_ka, d_counters, st), "mark");
    {
        size_t b = tmp_bytes;
        hip_check(finder_scan(d_tmp, b, d_ka, d_kb, n + 1, st), "scan");
    }
    unsigned long long h_counters[kFnoCounters];
    uint64_t total_bytes = 0;
    hip_check(hipMemcpyAsync(h_counters, d_counters, sizeof h_counters, hipMemcpyDeviceToHost, st), "counters");
    hip_check(hipMemcpyAsync(&total_bytes, d_kb + n, 8, hipMemcpyDeviceToHost, st), "size of the text");
    hip_check(hipStreamSynchronize(st), "synchronize");
    if (h_counters[4]) return false;
    const auto t1 = now();
    char* d_text = d.get<char>(total_bytes);
    hip_check(fno_format(d_rec, perm, d_ka, d_kb, n, d_text, st), "format");
    char* h_text = text_of(total_bytes);
    if (total_bytes) hip_check(hipMemcpy(h_text, d_text, total_bytes, hipMemcpyDeviceToHost), "copy of the text");
    for (int k = 0; k < 4; k++) counters[k] = h_counters[k];
    counters[4] = h_counters[5];
    if (seconds) {
        seconds[0] = std::chrono::duration<double>(t1 - t0).count();
        seconds[1] = std::chrono::duration<double>(now() - t1).count();
    }
    return true;
}

bool fno3_lines_on_device(const FnoItem* items, uint64_t n, bool no_inclusions, const std::function<char*(uint64_t)>& text_of, uint64_t* n_lines,
                          double* seconds) {
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    hipStream_t st = nullptr;
    DeviceBuffers d;
    FnoItem* d_items = d.get<FnoItem>(n);
    Fno3Rec* d_rec = d.get<Fno3Rec>(n);
    uint64_t *d_len = d.get<uint64_t>(n + 1), *d_off = d.get<uint64_t>(n + 1);
    unsigned long long* d_counters = d.get<unsigned long long>(kFnoCounters);
    size_t scan_bytes = 0;
    hip_check(finder_scan(nullptr, scan_bytes, d_len, d_off, n + 1, st), "scan size");
    void* d_tmp = d.get<char>(scan_bytes);
    hip_check(hipMemcpyAsync(d_items, items, n * sizeof(FnoItem), hipMemcpyHostToDevice, st), "copy of the candidate pairs");
    hip_check(hipMemsetAsync(d_counters, 0, kFnoCounters * sizeof(unsigned long long), st), "memset");
    hip_check(fno3_deduce(d_items, n, no_inclusions ? 1u : 0u, d_rec, d_len, d_counters, st), "deduce");
    hip_check(finder_scan(d_tmp, scan_bytes, d_len, d_off, n + 1, st), "scan");
    unsigned long long h_counters[kFnoCounters];
    uint64_t total_bytes = 0;
    hip_check(hipMemcpyAsync(h_counters, d_counters, sizeof h_counters, hipMemcpyDeviceToHost, st), "counters");
    hip_check(hipMemcpyAsync(&total_bytes, d_off + n, 8, hipMemcpyDeviceToHost, st), "size of the text");
    hip_check(hipStreamSynchronize(st), "synchronize");
    if (h_counters[4]) return false;
    const auto t1 = now();
    char* d_text = d.get<char>(total_bytes);
    hip_check(fno3_format(d_rec, d_len, d_off, n, d_text, st), "format");
    char* h_text = text_of(total_bytes);
    if (total_bytes) hip_check(hipMemcpy(h_text, d_text, total_bytes, hipMemcpyDeviceToHost), "copy of the text");
    if (n_lines) *n_lines = h_counters[5];
    if (seconds) {
        seconds[0] = std::chrono::duration<double>(t1 - t0).count();
        seconds[1] = std::chrono::duration<double>(now() - t1).count();
    }
    return true;
}

}  // namespace hc
