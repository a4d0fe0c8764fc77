// Philox4x32-10 (Salmon et al., SC'11): counter-based, stateless -- forward and backward regenerate
// the same draw from (seed, counter) instead of storing masks in HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace uaps {
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(uint64_t ctr, uint64_t key) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x55415053u /* "UAPS" */, c3 = 0;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }   // [0,1)
__device__ __forceinline__ uint32_t pick(const U4& r, int i) { return i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w; }
}  // namespace uaps
