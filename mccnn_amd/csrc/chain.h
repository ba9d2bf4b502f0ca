// Pieces of the single-pass (decoupled look-back) prefix sum shared by scan.hip and by the kernels that carry a prefix sum
// of their own instead of a launch between them (conv_rows.hip: the row-plan layout).
#pragma once
#include "common.h"

namespace mccnn {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

// Block-wide exclusive scan of one value per thread; returns exclusive prefix, total via ref.
__device__ __forceinline__ int block_excl_scan(int v, int& total, int* lds /*>=4 ints*/) {
    int incl = wave_incl_scan(v);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        int s = lds[w];
        if (w < wave) woff += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return woff + incl - v;
}

// Decoupled look-back, called by the FIRST WAVE of the workgroup that holds tile `tile` (its aggregate: tot): publishes
// the aggregate, looks back over the status words of the predecessors -- 64 at a time -- until it meets an inclusive
// prefix, publishes the tile's own inclusive prefix and returns its exclusive one. A status word is ONE aligned 8-byte
// agent-scope store: bits 63..62 = state (0 empty, 1 aggregate, 2 inclusive prefix), low 32 bits = value -- the data is
// the flag, no fences needed. The words must be zero on entry; tiles must be taken from a ticket (the word behind the
// last status word) or in dispatch order, so that a tile's predecessors are running when it waits for them.
__device__ __forceinline__ int chain_lookback(unsigned long long* status, int tile, int tot, int lane) {
    int excl = 0;
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(status, (2ull << 62) | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(status + tile, (1ull << 62) | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int hi = tile - 1;  // look back over tiles hi, hi - 1, ... (lane l reads tile hi - l)
    while (true) {
        const int p = hi - lane;
        unsigned long long w = 0;
        bool ready;
        do {  // every word of the window has to be published before the window can be summed
            w = (p >= 0) ? __hip_atomic_load(status + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
            ready = __all((w >> 62) != 0);
            if (!ready) __builtin_amdgcn_s_sleep(2);
        } while (!ready);
        const unsigned long long pref = __ballot((w >> 62) == 2);      // lanes holding an inclusive prefix
        const int first = (int)__builtin_ctzll(pref ? pref : 1ull << 63);
        int val = (pref == 0 || lane <= first) ? (int)(unsigned)w : 0;    // aggregates up to and incl. the prefix
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
        excl += val;
        if (pref) break;
        hi -= 64;
    }
    if (lane == 0) __hip_atomic_store(status + tile, (2ull << 62) | (unsigned)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

}  // namespace mccnn
