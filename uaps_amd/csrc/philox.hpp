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
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair, both quarter rate
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
// Step state (uaps_set_step_state): a small device buffer the host refreshes before every step so that a CAPTURED step
// (hipGraph replay: kernel arguments are frozen) still draws new random numbers and sees the current schedule values.
//   word 0-1  uint64 added to every Philox key          word 2-9   float mixing weights w[8] (UAPS_train.py:251)
//   word 10   cw1    word 11  cw2  (UAPS_train.py:279-280)          word 12-13 Adam lr / bias-correction1, 1 / sqrt(bias-correction2)
// A null pointer (the default) means: use the by-value arguments.
constexpr int kStepW = 2, kStepCw1 = 10, kStepCw2 = 11, kStepAdam = 12, kStepWords = 16;
__device__ __forceinline__ uint64_t step_key(uint64_t seed, const uint32_t* st) {
    return st ? seed + (((uint64_t)st[1] << 32) | (uint64_t)st[0]) : seed;
}
__device__ __forceinline__ float step_f(const uint32_t* st, int word, float by_value) {      // NaN by value = read the step state
    return (st && by_value != by_value) ? __uint_as_float(st[word]) : by_value;
}
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }   // [0,1)
__device__ __forceinline__ uint32_t pick(const U4& r, int i) { return i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w; }
}  // namespace uaps
extern "C" const void* uaps_get_step_state(void);
