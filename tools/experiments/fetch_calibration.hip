// What does FETCH_SIZE count for the access shapes of the scoring kernel?  Every piece of a 4 GiB table (far beyond the
// 256 MiB Infinity Cache and the L2s) is read exactly once, in a scattered order, in pieces of
//   16 B  (one lane, one piece: the per-lane fetch),
//   64 B  (a quad reads one 64-byte row: the cooperative fetch),
//   128 B (eight lanes read one aligned 128-byte line),
//   and streamed (lane-linear, the guide's calibration case),
// so the bytes that must come from HBM are known: 4 GiB each (every 128-byte line is needed whole sooner or later, but a
// scattered order cannot merge the pieces of a line: they are fetched as often as the cache lets them fall out).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calibration fetch_calibration.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calibration
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

// piece i of the table -> scattered piece index: an odd multiplier is a bijection modulo a power of two
__device__ __forceinline__ uint64_t scatter(uint64_t i, uint64_t n_pow2) { return (i * 0x9E3779B97F4A7C15ull) & (n_pow2 - 1); }

template <int LANES_PER_PIECE /* 1, 4, 8; 0 = streamed */>
__global__ __launch_bounds__(256) void read_pieces(const uint4* __restrict__ table, uint64_t n16 /* 16-byte units, power of two */,
                                                   uint32_t* __restrict__ out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t u = gid; u < n16; u += stride) {
        uint64_t at;
        if (LANES_PER_PIECE == 0) {
            at = u;
        } else {
            const uint64_t piece = u / LANES_PER_PIECE, n_pieces = n16 / LANES_PER_PIECE;
            at = scatter(piece, n_pieces) * LANES_PER_PIECE + (u % LANES_PER_PIECE);
        }
        const uint4 v = table[at];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;  // keeps the loads
}

int main() {
    const uint64_t bytes = 4ull << 30, n16 = bytes / 16;
    uint4* d;
    uint32_t* d_out;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&d_out, 4));
    CK(hipMemset(d, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 16;
    for (int shape = 0; shape < 4; shape++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (shape == 0) read_pieces<0><<<grid, 256>>>(d, n16, d_out);
            else if (shape == 1) read_pieces<1><<<grid, 256>>>(d, n16, d_out);
            else if (shape == 2) read_pieces<4><<<grid, 256>>>(d, n16, d_out);
            else read_pieces<8><<<grid, 256>>>(d, n16, d_out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1)
                printf("{\"shape\": \"%s\", \"bytes_read_once\": %llu, \"ms\": %.3f, \"TBps_useful\": %.3f}\n",
                       shape == 0 ? "streamed" : (shape == 1 ? "16-byte pieces" : (shape == 2 ? "64-byte rows (quads)" : "128-byte lines (8 lanes)")),
                       (unsigned long long)bytes, ms, (double)bytes / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
