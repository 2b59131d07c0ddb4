// How fast does pread fill a buffer from the page cache, by the kind of buffer?  2 GiB file, 16 MiB blocks, T threads.
//   hipcc -O2 -o pread_targets pread_targets.cpp -lpthread && ./pread_targets
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/pread_targets.bin";
    const size_t total = 2ull << 30, blk = 16u << 20, nblk = total / blk;
    {
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
        std::vector<char> buf(blk, 'x');
        for (size_t i = 0; i < nblk; i++)
            if (write(fd, buf.data(), blk) != (ssize_t)blk) return 1;
        close(fd);
    }
    int fd = open(path, O_RDONLY);
    for (int kind = 0; kind < 4; kind++) {
        char* h[4];
        for (int k = 0; k < 4; k++) {
            if (kind == 0) h[k] = (char*)aligned_alloc(4096, blk);
            else if (hipHostMalloc((void**)&h[k], blk, kind == 1 ? hipHostMallocDefault : (kind == 2 ? hipHostMallocNonCoherent : hipHostMallocWriteCombined)) != hipSuccess) return 2;
            for (size_t i = 0; i < blk; i += 4096) h[k][i] = 1;
        }
        for (int T : {8, 32}) {
            double t0 = now();
            for (size_t i = 0; i < nblk; i++) {
                std::vector<std::thread> th;
                for (int q = 0; q < T; q++)
                    th.emplace_back([&, q] {
                        size_t a = blk * q / T, b = blk * (q + 1) / T;
                        if (pread(fd, h[i & 3] + a, b - a, (off_t)(i * blk + a)) < 0) perror("pread");
                    });
                for (auto& x : th) x.join();
            }
            const double dt = now() - t0;
            printf("{\"buffer\": \"%s\", \"threads\": %d, \"GiB\": 2, \"s\": %.3f, \"GBps\": %.1f}\n",
                   kind == 0 ? "malloc" : (kind == 1 ? "hipHostMalloc default" : (kind == 2 ? "hipHostMalloc non-coherent" : "hipHostMalloc write-combined")), T, dt,
                   2.147 / dt);
        }
        for (int k = 0; k < 4; k++) kind == 0 ? free(h[k]) : (void)hipHostFree(h[k]);
    }
    close(fd);
    unlink(path);
    return 0;
}
