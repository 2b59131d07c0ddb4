// What does it cost to bring 300 MB back from the device: pageable destination, page-locked destination (and the
// price of page-locking it first), several pageable pieces from several threads.  hipcc -O2 -o pin_cost pin_cost.cpp
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 300u << 20;
    void* d;
    hipMalloc(&d, n);
    hipMemset(d, 1, n);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        char* p = (char*)malloc(n);
        double t = now();
        hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
        printf("pageable (fresh malloc) D2H: %.3f s\n", now() - t);
        t = now();
        hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
        printf("pageable (touched) D2H:      %.3f s\n", now() - t);
        t = now();
        hipMemcpy(d, p, n, hipMemcpyHostToDevice);
        printf("pageable H2D:                %.3f s\n", now() - t);
        free(p);
        void* h;
        t = now();
        hipHostMalloc(&h, n, hipHostMallocDefault);
        printf("hipHostMalloc 300 MB:        %.3f s\n", now() - t);
        t = now();
        hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
        printf("page-locked D2H:             %.3f s\n", now() - t);
        t = now();
        hipHostFree(h);
        printf("hipHostFree:                 %.3f s\n", now() - t);
        p = (char*)malloc(n);
        memset(p, 0, n);
        t = now();
        hipHostRegister(p, n, hipHostRegisterDefault);
        printf("hipHostRegister (touched):   %.3f s\n", now() - t);
        t = now();
        hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
        printf("registered D2H:              %.3f s\n", now() - t);
        hipHostUnregister(p);
        const int T = 4;
        std::vector<std::thread> th;
        t = now();
        for (int k = 0; k < T; k++)
            th.emplace_back([&, k] {
                hipStream_t s;
                hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                hipMemcpyAsync(p + n / T * k, (char*)d + n / T * k, n / T, hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                hipStreamDestroy(s);
            });
        for (auto& x : th) x.join();
        printf("pageable D2H, 4 threads:     %.3f s\n", now() - t);
        free(p);
    }
    return 0;
}
