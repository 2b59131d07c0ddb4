// hc_api_prims.cpp — the device primitives of hc_prims.hip behind host-buffer entry points (include/hcedge.h, "device
// primitives"): what their unit tests call.  Synchronous; device 0 unless HC_DEVICE says otherwise.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#include "../../include/hcedge.h"
#include "hc_ctx.h"
#include "hc_prims.h"

static int fail(int status, const std::string& what) { return hc::set_last_error(status, what); }

namespace {
struct Dev {  // a device buffer freed on every return path
    void* p = nullptr;
    ~Dev() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
int pick_device() {
    const char* e = getenv("HC_DEVICE");
    return e ? atoi(e) : 0;
}
}  // namespace

extern "C" {

int hc_dev_radix_sort(uint32_t key_bytes, uint32_t val_bytes, const void* keys, const void* vals, void* keys_out, void* vals_out, uint64_t n,
                      int begin_bit, int end_bit) {
    const bool kv = (key_bytes == 4 && val_bytes == 4) || (key_bytes == 8 && (val_bytes == 4 || val_bytes == 8 || val_bytes == 0));
    if (!kv) return fail(HC_ERR_ARG, "hc_dev_radix_sort: supported shapes are (4, 4), (8, 4), (8, 8) and (8, 0) bytes of key and value");
    if (n && (!keys || !keys_out || (val_bytes && (!vals || !vals_out)))) return fail(HC_ERR_ARG, "hc_dev_radix_sort: null buffer");
    if (n == 0) return HC_OK;
    HC_HIP(hipSetDevice(pick_device()));
    Dev k0, k1, v0, v1, tmp;
    const size_t tb = hc::prims::sort_temp_bytes(n, key_bytes, val_bytes);
    HC_HIP(k0.alloc(n * key_bytes));
    HC_HIP(k1.alloc(n * key_bytes));
    HC_HIP(v0.alloc(n * val_bytes));
    HC_HIP(v1.alloc(n * val_bytes));
    HC_HIP(tmp.alloc(tb));
    HC_HIP(hipMemcpy(k0.p, keys, n * key_bytes, hipMemcpyHostToDevice));
    if (val_bytes) HC_HIP(hipMemcpy(v0.p, vals, n * val_bytes, hipMemcpyHostToDevice));
    hipError_t e;
    if (key_bytes == 4)
        e = hc::prims::sort_pairs(tmp.p, tb, (const uint32_t*)k0.p, (uint32_t*)k1.p, (const uint32_t*)v0.p, (uint32_t*)v1.p, n, begin_bit, end_bit, nullptr);
    else if (val_bytes == 4)
        e = hc::prims::sort_pairs(tmp.p, tb, (const uint64_t*)k0.p, (uint64_t*)k1.p, (const uint32_t*)v0.p, (uint32_t*)v1.p, n, begin_bit, end_bit, nullptr);
    else if (val_bytes == 8)
        e = hc::prims::sort_pairs(tmp.p, tb, (const uint64_t*)k0.p, (uint64_t*)k1.p, (const uint64_t*)v0.p, (uint64_t*)v1.p, n, begin_bit, end_bit, nullptr);
    else
        e = hc::prims::sort_keys(tmp.p, tb, (const uint64_t*)k0.p, (uint64_t*)k1.p, n, begin_bit, end_bit, nullptr);
    HC_HIP(e);
    HC_HIP(hipDeviceSynchronize());
    HC_HIP(hipMemcpy(keys_out, k1.p, n * key_bytes, hipMemcpyDeviceToHost));
    if (val_bytes) HC_HIP(hipMemcpy(vals_out, v1.p, n * val_bytes, hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_dev_exclusive_sum(uint32_t elem_bytes, const void* in, void* out, uint64_t n) {
    if (elem_bytes != 4 && elem_bytes != 8) return fail(HC_ERR_ARG, "hc_dev_exclusive_sum: elements of 4 or 8 bytes");
    if (n && (!in || !out)) return fail(HC_ERR_ARG, "hc_dev_exclusive_sum: null buffer");
    if (n == 0) return HC_OK;
    HC_HIP(hipSetDevice(pick_device()));
    Dev a, tmp;
    const size_t tb = hc::prims::scan_temp_bytes(n, elem_bytes);
    HC_HIP(a.alloc(n * elem_bytes));
    HC_HIP(tmp.alloc(tb));
    HC_HIP(hipMemcpy(a.p, in, n * elem_bytes, hipMemcpyHostToDevice));
    if (elem_bytes == 4) HC_HIP(hc::prims::exclusive_sum(tmp.p, tb, (const uint32_t*)a.p, (uint32_t*)a.p, n, nullptr));  // in place
    else HC_HIP(hc::prims::exclusive_sum(tmp.p, tb, (const uint64_t*)a.p, (uint64_t*)a.p, n, nullptr));
    HC_HIP(hipDeviceSynchronize());
    HC_HIP(hipMemcpy(out, a.p, n * elem_bytes, hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_dev_select_flagged(const uint8_t* flags, uint64_t n, uint32_t* idx_out, uint64_t* count) {
    if (!count || (n && (!flags || !idx_out))) return fail(HC_ERR_ARG, "hc_dev_select_flagged: null argument");
    *count = 0;
    if (n == 0) return HC_OK;
    HC_HIP(hipSetDevice(pick_device()));
    Dev f, idx, cnt, tmp;
    const size_t tb = hc::prims::select_temp_bytes(n);
    HC_HIP(f.alloc(n));
    HC_HIP(idx.alloc(n * 4));
    HC_HIP(cnt.alloc(8));
    HC_HIP(tmp.alloc(tb));
    HC_HIP(hipMemcpy(f.p, flags, n, hipMemcpyHostToDevice));
    HC_HIP(hc::prims::select_flagged(tmp.p, tb, (const uint8_t*)f.p, n, (uint32_t*)idx.p, (unsigned long long*)cnt.p, nullptr));
    HC_HIP(hipDeviceSynchronize());
    unsigned long long k = 0;
    HC_HIP(hipMemcpy(&k, cnt.p, 8, hipMemcpyDeviceToHost));
    *count = k;
    if (k) HC_HIP(hipMemcpy(idx_out, idx.p, k * 4, hipMemcpyDeviceToHost));
    return HC_OK;
}

int hc_dev_unique_u64(const uint64_t* in, uint64_t n, uint64_t* out, uint64_t* count) {
    if (!count || (n && (!in || !out))) return fail(HC_ERR_ARG, "hc_dev_unique_u64: null argument");
    *count = 0;
    if (n == 0) return HC_OK;
    HC_HIP(hipSetDevice(pick_device()));
    Dev a, b, cnt, tmp;
    const size_t tb = hc::prims::select_temp_bytes(n);
    HC_HIP(a.alloc(n * 8));
    HC_HIP(b.alloc(n * 8));
    HC_HIP(cnt.alloc(8));
    HC_HIP(tmp.alloc(tb));
    HC_HIP(hipMemcpy(a.p, in, n * 8, hipMemcpyHostToDevice));
    HC_HIP(hc::prims::unique(tmp.p, tb, (const uint64_t*)a.p, (uint64_t*)b.p, (unsigned long long*)cnt.p, n, nullptr));
    HC_HIP(hipDeviceSynchronize());
    unsigned long long k = 0;
    HC_HIP(hipMemcpy(&k, cnt.p, 8, hipMemcpyDeviceToHost));
    *count = k;
    if (k) HC_HIP(hipMemcpy(out, b.p, k * 8, hipMemcpyDeviceToHost));
    return HC_OK;
}

}  // extern "C"
