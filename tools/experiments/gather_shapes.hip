// How should the scoring kernel fetch its symbol windows?  Memory side only (no table, no sums): the C3 access pattern
// (2 x 150-symbol mates per read, one read shared by ~200 consecutive candidates, the partner a random read, windows of
// 75..150 symbols starting at `pos` in the A role and at 0 in the B role) read
//   v1: as the kernel does today — one lane per candidate, 16 bytes per lane and load, 64-symbol groups prefetched;
//   v2: cooperatively — four lanes fetch the four 16-byte pieces of ONE candidate's 64-symbol group (64 contiguous bytes
//       per quad), the pieces go through a per-wave LDS buffer (padded rows: conflict-free) to the lane that owns the candidate;
// each with the read slots as laid out today (16-byte aligned, 192-byte stride) and 128-byte aligned (256-byte stride).
//   hipcc --offload-arch=gfx950 -O3 -o gather_shapes gather_shapes.hip && ./gather_shapes
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

struct Cand {
    uint32_t seqA, seqB;  // first sequence (/1) of the A-role and of the B-role read; /2 is the next sequence
    uint32_t pos1, pos2;
};

constexpr uint32_t kLen = 150;

__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// v1: one lane per candidate
template <int G>
__global__ __launch_bounds__(256, 4) void gather_v1(const uint8_t* __restrict__ sym, uint32_t stride /* bytes per orientation slot */,
                                                   const Cand* __restrict__ cands, uint64_t n, uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Cand c = cands[i];
    uint32_t acc = 0;
    for (int sub = 0; sub < 2; sub++) {
        const uint32_t pos = sub ? c.pos2 : c.pos1;
        const uint8_t* a = sym + (uint64_t)(c.seqA + sub) * 2u * stride + pos;
        const uint8_t* b = sym + (uint64_t)(c.seqB + sub) * 2u * stride;
        const uint32_t L = kLen - pos, nch = (L + 15u) >> 4;
        uint4 na[G], nb[G];
#pragma unroll
        for (int q = 0; q < G; q++)
            if ((uint32_t)q < nch) {
                __builtin_memcpy(&na[q], a + 16 * q, 16);
                __builtin_memcpy(&nb[q], b + 16 * q, 16);
            }
        for (uint32_t c0 = 0; c0 < nch; c0 += G) {
            uint4 ca[G], cb[G];
#pragma unroll
            for (int q = 0; q < G; q++) {
                ca[q] = na[q];
                cb[q] = nb[q];
            }
#pragma unroll
            for (int q = 0; q < G; q++)
                if (c0 + G + q < nch) {
                    __builtin_memcpy(&na[q], a + 16 * (c0 + G + q), 16);
                    __builtin_memcpy(&nb[q], b + 16 * (c0 + G + q), 16);
                }
#pragma unroll
            for (int q = 0; q < G; q++)
                if (c0 + q < nch) acc += fold(ca[q]) * 3u + fold(cb[q]);
        }
    }
    out[i] = acc;
}

// v2: quads fetch one candidate's 64-byte group; LDS hands the pieces to the owner lane
constexpr uint32_t kRow = 80;  // 64 + 16: rows of 16 consecutive lanes fall into 16 different 16-byte bank groups
template <int DESC>
__global__ __launch_bounds__(256, 4) void gather_v2(const uint8_t* __restrict__ sym, uint32_t stride, const Cand* __restrict__ cands,
                                                   uint64_t n, uint32_t* __restrict__ out, const uint4* __restrict__ desc32,
                                                   const uint2* __restrict__ desc8) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[4][64 * kRow];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint8_t* buf = stage[wave];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    Cand c = {0, 0, 0, 0};
    if (i < n) c = cands[i];
    // where a read's sequences start: computed (fixed stride), or looked up per read as the scoring kernel does
    uint64_t baseA = (uint64_t)c.seqA * 2u * stride, baseB = (uint64_t)c.seqB * 2u * stride;
    if (DESC == 1 && i < n) {
        const uint4 a0 = desc32[(c.seqA >> 1) * 2], a1 = desc32[(c.seqA >> 1) * 2 + 1];
        const uint4 b0 = desc32[(c.seqB >> 1) * 2], b1 = desc32[(c.seqB >> 1) * 2 + 1];
        baseA = (((uint64_t)a0.y << 32) | a0.x) + (a1.z & 0u);
        baseB = (((uint64_t)b0.y << 32) | b0.x) + (b1.z & 0u);
    }
    if (DESC == 2 && i < n) {
        const uint2 a = desc8[c.seqA >> 1], b = desc8[c.seqB >> 1];
        baseA = (uint64_t)a.x * 16u + (a.y & 0u);
        baseB = (uint64_t)b.x * 16u + (b.y & 0u);
    }
    const uint32_t quad = lane >> 2, piece = lane & 3u;
    uint32_t acc = 0;
    for (int sub = 0; sub < 2; sub++) {
        const uint32_t pos = sub ? c.pos2 : c.pos1;
        const uint64_t offA = baseA + (uint64_t)sub * 2u * stride + pos;
        const uint64_t offB = baseB + (uint64_t)sub * 2u * stride;
        const uint32_t L = i < n ? kLen - pos : 0u;
        // what the loader lanes need of the four candidates they fetch for
        uint64_t la[4], lb[4];
        uint32_t ll[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int src = 16 * j + (int)quad;
            la[j] = ((uint64_t)__shfl((uint32_t)(offA >> 32), src) << 32) | __shfl((uint32_t)offA, src);
            lb[j] = ((uint64_t)__shfl((uint32_t)(offB >> 32), src) << 32) | __shfl((uint32_t)offB, src);
            ll[j] = __shfl(L, src);
        }
        uint32_t maxL = L;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(maxL, d);
            maxL = o > maxL ? o : maxL;
        }
        const uint32_t ngroups = (maxL + 63u) >> 6;
        uint4 na[4], nb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            na[j] = nb[j] = make_uint4(0, 0, 0, 0);
            if (16u * piece < ll[j]) {
                __builtin_memcpy(&na[j], sym + la[j] + 16u * piece, 16);
                __builtin_memcpy(&nb[j], sym + lb[j] + 16u * piece, 16);
            }
        }
        for (uint32_t g = 0; g < ngroups; g++) {
            uint4 ca[4], cb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) *(uint4*)(buf + (16 * j + quad) * kRow + 16 * piece) = na[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < 4; p++) ca[p] = *(const uint4*)(buf + lane * kRow + 16 * p);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; j++) *(uint4*)(buf + (16 * j + quad) * kRow + 16 * piece) = nb[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < 4; p++) cb[p] = *(const uint4*)(buf + lane * kRow + 16 * p);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t at = 64u * (g + 1u);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (at + 16u * piece < ll[j]) {
                    __builtin_memcpy(&na[j], sym + la[j] + at + 16u * piece, 16);
                    __builtin_memcpy(&nb[j], sym + lb[j] + at + 16u * piece, 16);
                }
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (64u * g + 16u * p < L) acc += fold(ca[p]) * 3u + fold(cb[p]);
        }
    }
    if (i < n) out[i] = acc;
}

int main(int argc, char** argv) {
    const uint32_t n_reads = 500000;
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 24);
    std::mt19937_64 rng(5);
    std::vector<Cand> h(n);
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t shared = (uint32_t)(i / 200) % n_reads, partner = (uint32_t)(rng() % n_reads);
        const bool a_shared = rng() & 1;
        h[i].seqA = 2u * (a_shared ? shared : partner);
        h[i].seqB = 2u * (a_shared ? partner : shared);
        h[i].pos1 = (uint32_t)(rng() % 76);
        h[i].pos2 = (uint32_t)(rng() % 76);
    }
    Cand* d_c;
    uint32_t* d_out;
    CK(hipMalloc(&d_c, n * sizeof(Cand)));
    CK(hipMalloc(&d_out, n * 4));
    CK(hipMemcpy(d_c, h.data(), n * sizeof(Cand), hipMemcpyHostToDevice));
    std::vector<uint32_t> ref(n), got(n);
    for (uint32_t stride : {192u, 256u}) {
        const size_t bytes = (size_t)n_reads * 2 * 2 * stride + 256;
        std::vector<uint8_t> hs(bytes);
        for (size_t k = 0; k < bytes; k++) hs[k] = (uint8_t)(rng() >> 11);
        uint8_t* d_s;
        CK(hipMalloc(&d_s, bytes));
        CK(hipMemcpy(d_s, hs.data(), bytes, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int grid = (int)((n + 255) / 256);
        // descriptors of the reads (pairs): 32 bytes as the kernel's ReadDesc, or 8 bytes (offset in 16-byte units, lengths)
        std::vector<uint4> hd32((size_t)n_reads * 2);
        std::vector<uint2> hd8(n_reads);
        for (uint32_t r = 0; r < n_reads; r++) {
            const uint64_t off = (uint64_t)r * 4u * stride;
            hd32[2 * r] = make_uint4((uint32_t)off, (uint32_t)(off >> 32), (uint32_t)(off + 2u * stride), (uint32_t)((off + 2u * stride) >> 32));
            hd32[2 * r + 1] = make_uint4(kLen, kLen, 1, 0);
            hd8[r] = make_uint2((uint32_t)(off / 16u), kLen | (kLen << 16));
        }
        uint4* d_d32;
        uint2* d_d8;
        CK(hipMalloc(&d_d32, hd32.size() * sizeof(uint4)));
        CK(hipMalloc(&d_d8, hd8.size() * sizeof(uint2)));
        CK(hipMemcpy(d_d32, hd32.data(), hd32.size() * sizeof(uint4), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_d8, hd8.data(), hd8.size() * sizeof(uint2), hipMemcpyHostToDevice));
        for (int variant = 0; variant < 5; variant++) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; rep++) {
                CK(hipEventRecord(e0));
                if (variant == 0) gather_v1<4><<<grid, 256>>>(d_s, stride, d_c, n, d_out);
                else if (variant == 1) gather_v1<2><<<grid, 256>>>(d_s, stride, d_c, n, d_out);
                else if (variant == 2) gather_v2<0><<<grid, 256>>>(d_s, stride, d_c, n, d_out, d_d32, d_d8);
                else if (variant == 3) gather_v2<1><<<grid, 256>>>(d_s, stride, d_c, n, d_out, d_d32, d_d8);
                else gather_v2<2><<<grid, 256>>>(d_s, stride, d_c, n, d_out, d_d32, d_d8);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            CK(hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost));
            if (variant == 0) ref = got;
            const bool same = ref == got;
            printf("{\"slot_stride\": %u, \"variant\": \"%s\", \"ms\": %.4f, \"cand_per_s\": %.3e, \"same_result\": %s}\n", stride,
                   variant == 0 ? "v1 lane-per-candidate G=4"
                                : (variant == 1 ? "v1 lane-per-candidate G=2"
                                                : (variant == 2 ? "v2 quad fetch via LDS, offsets computed"
                                                                : (variant == 3 ? "v2 + 32-byte read descriptors" : "v2 + 8-byte read descriptors"))),
                   best,
                   (double)n / (best * 1e-3), same ? "true" : "false");
        }
        CK(hipFree(d_s));
        CK(hipFree(d_d32));
        CK(hipFree(d_d8));
    }
    return 0;
}
