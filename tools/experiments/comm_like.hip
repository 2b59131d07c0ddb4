// comm_like.hip — stand-ins for a collective library's kernels, for ONE measurement (tools/coresident.py, VERDICT r4 next #1d): what
// does a communication kernel cost beside the scoring kernel, which holds one 1 024-lane workgroup (145 KiB of LDS, 4 x 120 of a
// SIMD's 512 registers) on every CU for the launch's whole duration?  A device-to-device copy of `bytes` by `blocks` workgroups of 256
// lanes, 16 bytes per lane and pass, in two shapes:
//   fat = 0   "thin": a few registers, no LDS — fits on a CU beside a scoring workgroup
//   fat = 1   "library-like": 128 registers per lane (clobbered, so that the allocation is that of a generic collective kernel) and
//             lds_bytes of LDS per workgroup — cannot share a CU with a scoring workgroup, needs a free one
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcommlike.so comm_like.hip   (tools/coresident.py does it)
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(256) void comm_thin_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" __global__ __launch_bounds__(256) void comm_fat_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, uint64_t n16) {
    extern __shared__ uint4 stage[];
    asm volatile("" ::: "v127");  // the register allocation of a generic collective kernel
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        stage[threadIdx.x] = src[i];  // through LDS, as a collective's staging does (the same lane reads it back: no barrier needed)
        dst[i] = stage[threadIdx.x];
    }
}

extern "C" int comm_like_launch(void* dst, const void* src, uint64_t bytes, int blocks, int lds_bytes, int fat, void* stream) {
    if (fat) {
        if (lds_bytes < 4096) lds_bytes = 4096;
        (void)hipFuncSetAttribute((const void*)comm_fat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(comm_fat_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (uint4*)dst, (const uint4*)src, bytes / 16);
    } else {
        hipLaunchKernelGGL(comm_thin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)dst, (const uint4*)src, bytes / 16);
    }
    return (int)hipGetLastError();
}
