// One-shot side arguments of the next entry point called on this thread (include/uaps_hip.h, uaps_next_call_hints).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/uaps_hip.h"

namespace uaps {
// returns the pending hints (all-null when none) and clears them: every entry point that understands hints calls this
// first, so that a hint never outlives the call it was meant for
uaps_call_hints take_hints();

// block-wide max|v| of the calling threads' values -> atomic max on a device scalar (non-negative floats order like their
// bit patterns); every thread of the block must call it
__device__ __forceinline__ void block_amax_to(float* dst, float v, float* smem_16) {
    v = __builtin_fabsf(v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o, 64));
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) smem_16[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = smem_16[0];
        for (int i = 1; i < nw; ++i) m = __builtin_fmaxf(m, smem_16[i]);
        // thousands of blocks raise one scalar: only those that would change it issue the atomic (a stale, smaller value read
        // here merely costs a redundant atomic; same-address atomics serialise at the memory side)
        if (m == m && m > __builtin_bit_cast(float, __atomic_load_n(reinterpret_cast<unsigned*>(dst), __ATOMIC_RELAXED))) atomicMax(reinterpret_cast<unsigned*>(dst), __builtin_bit_cast(unsigned, m));
    }
}
}  // namespace uaps
