// hc_lookback.h — a single-pass exclusive prefix over the tiles of a launch (decoupled look-back), for the stage's ordered compactions
// (the kept rows of a scored block, the line starts of a block of text): one launch and one pass over the data where round 4 ran count /
// one-workgroup scan / write triples.  A tile publishes its count and looks back over its predecessors' words until it meets an inclusive
// prefix; tiles take their numbers from a ticket counter, so a tile's predecessors have always started.
// State: one 64-bit word per tile, [flag:2 | epoch:30 | value:32].  The EPOCH (the launch's number on this state array, from the host) makes
// the words of earlier launches invalid without a memset; the state array is zeroed once, when it is allocated (epoch 0 is never used).
// The ticket counter lives behind the tiles' words; the last tile of a launch sets it back to 0.
// Memory order: RELAXED agent-scope atomics only.  The state word is all the tiles tell each other (flag, epoch and value travel in ONE
// 64-bit word), so nothing needs release / acquire — and on this chip those are not free: every XCD has its own L2, agent-scope release /
// acquire means L2 write-backs and invalidations (the first version, with them: 457 us per 16 MiB block for the line starts, and every
// kernel running beside it 3 - 10 x slower).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hc {

struct Lookback {
    unsigned long long* state;  // n_tiles words, then the ticket counter
    uint32_t* ticket;
    uint32_t epoch;             // 1 .. 2^30 - 1, a new one per launch
};

__device__ __forceinline__ unsigned long long lb_pack(uint32_t flag, uint32_t epoch, uint32_t value) {
    return ((unsigned long long)flag << 62) | ((unsigned long long)(epoch & 0x3FFFFFFFu) << 32) | value;
}

// Called by EVERY lane of the workgroup, `total` = the tile's count (the same value in every lane).  tile = the ticket taken by
// lb_take_ticket.  Returns the number of items in the tiles before this one.  lds2: two words of LDS scratch.
__device__ __forceinline__ uint32_t lb_take_ticket(const Lookback& lb, uint32_t* lds2) {
    if (threadIdx.x == 0) lds2[0] = atomicAdd(lb.ticket, 1u);
    __syncthreads();
    const uint32_t t = lds2[0];
    __syncthreads();
    return t;
}

__device__ __forceinline__ uint32_t lb_exclusive(const Lookback& lb, uint32_t tile, uint32_t total, uint32_t* lds2) {
    if (threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(&lb.state[tile], lb_pack(tile == 0 ? 2u : 1u, lb.epoch, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        int base = (int)tile - 1;
        while (base >= 0) {
            const int idx = base - (int)lane;
            unsigned long long s = lb_pack(2u, lb.epoch, 0u);  // in front of tile 0: an inclusive prefix of nothing
            if (idx >= 0) {
                for (;;) {
                    s = __hip_atomic_load(&lb.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)((s >> 32) & 0x3FFFFFFFu) == (lb.epoch & 0x3FFFFFFFu) && (s >> 62) != 0ull) break;
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            const unsigned long long incl = __ballot((s >> 62) == 2ull);
            uint32_t v = (uint32_t)s;
            if (incl) {
                const uint32_t first = (uint32_t)__builtin_ctzll(incl);  // the nearest predecessor that knows its inclusive prefix
                v = lane <= first ? v : 0u;
            }
            for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
            excl += v;
            if (incl) break;
            base -= 64;
        }
        if (lane == 0) {
            if (tile != 0) __hip_atomic_store(&lb.state[tile], lb_pack(2u, lb.epoch, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds2[1] = excl;
        }
    }
    __syncthreads();
    const uint32_t e = lds2[1];
    __syncthreads();
    return e;
}

}  // namespace hc
