// Can the stage skip its host copy?  Today the producer preads the overlaps file into page-locked block buffers and the
// runtime DMAs them to the device.  Alternative: mmap the file and page-lock the mapping itself, 16 MiB at a time
// (hipHostRegister), so the DMA reads the page cache directly.  What do registration and un-registration cost against the
// pread they replace?     hipcc -O2 -o register_cost register_cost.cpp && ./register_cost [/path/to/scratch/file]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/register_cost.bin";
    const size_t total = 2ull << 30, blk = 16u << 20, nblk = total / blk;
    {  // a 2 GiB file, left in the page cache
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
        std::vector<char> buf(blk, 'x');
        for (size_t i = 0; i < nblk; i++) {
            buf[0] = (char)i;
            if (write(fd, buf.data(), blk) != (ssize_t)blk) return 1;
        }
        close(fd);
    }
    int fd = open(path, O_RDONLY);
    void* d;
    CK(hipMalloc(&d, blk * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));

    // (1) today: pread into page-locked buffers (8 threads) + async H2D, 4 buffers in rotation
    {
        char* h[4];
        hipEvent_t ev[4];
        for (int k = 0; k < 4; k++) {
            CK(hipHostMalloc((void**)&h[k], blk, hipHostMallocDefault));
            CK(hipEventCreate(&ev[k]));
        }
        double t0 = now(), t_read = 0;
        for (size_t i = 0; i < nblk; i++) {
            const int k = i & 3;
            if (i >= 4) CK(hipEventSynchronize(ev[k]));
            double t = now();
            std::vector<std::thread> th;
            const int T = 8;
            for (int q = 0; q < T; q++)
                th.emplace_back([&, q] {
                    size_t a = blk * q / T, b = blk * (q + 1) / T;
                    if (pread(fd, h[k] + a, b - a, (off_t)(i * blk + a)) < 0) perror("pread");
                });
            for (auto& x : th) x.join();
            t_read += now() - t;
            CK(hipMemcpyAsync((char*)d + k * blk, h[k], blk, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[k], s));
        }
        CK(hipStreamSynchronize(s));
        printf("{\"form\": \"pread into page-locked buffers + H2D\", \"GiB\": 2, \"total_s\": %.3f, \"pread_s\": %.3f}\n", now() - t0, t_read);
        for (int k = 0; k < 4; k++) CK(hipHostFree(h[k]));
    }
    // (2) mmap + hipHostRegister per block + H2D from the mapping + unregister (4 blocks in flight)
    for (int flags_sel = 0; flags_sel < 2; flags_sel++) {
        char* m = (char*)mmap(nullptr, total, PROT_READ | (flags_sel ? 0 : PROT_WRITE), MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (m == MAP_FAILED) {
            perror("mmap");
            return 1;
        }
        const unsigned flags = flags_sel ? hipHostRegisterReadOnly : hipHostRegisterDefault;
        hipEvent_t ev[4];
        for (int k = 0; k < 4; k++) CK(hipEventCreate(&ev[k]));
        double t0 = now(), t_reg = 0, t_unreg = 0;
        bool ok = true;
        for (size_t i = 0; i < nblk && ok; i++) {
            const int k = i & 3;
            if (i >= 4) {
                CK(hipEventSynchronize(ev[k]));
                double t = now();
                CK(hipHostUnregister(m + (i - 4) * blk));
                t_unreg += now() - t;
            }
            double t = now();
            hipError_t e = hipHostRegister(m + i * blk, blk, flags);
            t_reg += now() - t;
            if (e != hipSuccess) {
                printf("{\"form\": \"mmap + hipHostRegister(%s)\", \"error\": \"%s\"}\n", flags_sel ? "ReadOnly, PROT_READ" : "Default, PROT_READ|WRITE private", hipGetErrorString(e));
                ok = false;
                break;
            }
            CK(hipMemcpyAsync((char*)d + k * blk, m + i * blk, blk, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[k], s));
        }
        if (ok) {
            CK(hipStreamSynchronize(s));
            for (size_t i = nblk - 4; i < nblk; i++) CK(hipHostUnregister(m + i * blk));
            printf("{\"form\": \"mmap + hipHostRegister(%s) + H2D + unregister\", \"GiB\": 2, \"total_s\": %.3f, \"register_s\": %.3f, \"unregister_s\": %.3f}\n",
                   flags_sel ? "ReadOnly, PROT_READ" : "Default, PROT_READ|WRITE private", now() - t0, t_reg, t_unreg);
        }
        munmap(m, total);
    }
    // (3) plain pageable H2D straight from the mapping (the runtime stages it): page tables populated up front, or not
    for (int populate = 1; populate >= 0; populate--) {
        double tm = now();
        char* m = (char*)mmap(nullptr, total, PROT_READ, MAP_PRIVATE | (populate ? MAP_POPULATE : 0), fd, 0);
        printf("{\"form\": \"mmap%s\", \"GiB\": 2, \"total_s\": %.3f}\n", populate ? " MAP_POPULATE" : "", now() - tm);
        double t0 = now();
        for (size_t i = 0; i < nblk; i++) CK(hipMemcpyAsync((char*)d + (i & 3) * blk, m + i * blk, blk, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        printf("{\"form\": \"pageable H2D from the mapping%s\", \"GiB\": 2, \"total_s\": %.3f}\n", populate ? "" : " (not populated)", now() - t0);
        munmap(m, total);
    }
    close(fd);
    unlink(path);
    return 0;
}
