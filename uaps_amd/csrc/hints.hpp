// One-shot side arguments of the next entry point called on this thread (include/uaps_hip.h, uaps_next_call_hints).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "../../include/uaps_hip.h"

namespace uaps {
// returns the pending hints (all-null when none) and clears them: every entry point that understands hints calls this
// first, so that a hint never outlives the call it was meant for
uaps_call_hints take_hints();
// The explicit form (round 6: the *_h entry points, uaps_conv_ex): the caller's record of THIS call, copied size-versioned and validated
// exactly like uaps_next_call_hints does; `in` == NULL or struct_size == 0 = no hints.  Nothing thread-local is read or written.
int read_hints(const uaps_call_hints* in, uaps_call_hints& out);
#define UAPS_READ_HINTS(in, name)                              \
    uaps_call_hints name;                                       \
    {                                                           \
        const int rc_hints_ = uaps::read_hints(in, name);       \
        if (rc_hints_) return rc_hints_;                        \
    }

// uaps_next_launch_events (include/uaps_hip.h): a pair of events the calling thread's next MAIN kernel launch (the convolution /
// loss kernel of an entry point, not its packing, reduce or finalize launches) attaches to its dispatch, so that their
// elapsed time is that kernel's execution alone -- what rocprofv3's kernel trace reports -- instead of the event-to-event time
// of two extra packets on the stream.
// uaps_set_error_word: the sticky device error word the fp16-split kernels OR UAPS_ERR_* into, or nullptr
unsigned* error_word();
// uaps_account (include/uaps_hip.h): while switched on, every kernel entry point adds the ALGORITHMIC bytes of its launch -- each
// operand tensor read once, each result written once, fp32 / int64 as the reference holds them; workspaces, packed copies' re-reads,
// halo re-reads and partial sums are not counted -- to a process-wide tally (bench.py: roofline.step_algorithmic_bytes)
void account_bytes(double bytes);
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; bool armed = false, used = false; };
LaunchEvents& launch_events();
#define UAPS_LAUNCH_MAIN(kernel, grid, block, shmem, stream, ...)                                                            \
    do {                                                                                                                     \
        uaps::LaunchEvents& le_ = uaps::launch_events();                                                                     \
        if (le_.armed) {                                                                                                     \
            le_.armed = false; le_.used = true;                                                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, le_.start, le_.stop, 0, __VA_ARGS__);                  \
        } else {                                                                                                             \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                             \
        }                                                                                                                    \
    } while (0)

// A magnitude bound lives in UAPS_BOUND_SLOTS floats spaced UAPS_BOUND_STRIDE floats apart (include/uaps_hip.h); its value
// is the maximum over the slots.  Thousands of workgroups raising ONE address serialise at the memory side (measured: ~2.4 ns
// per atomic, +20 us on a 30 us kernel of 8192 blocks); spread over 16 lines they do not.
// Agent-scope loads: the slots are raised by memory-side atomics of the producing kernel, and a plain load may be served from a
// line this XCD's L2 still holds from before them -- the zero fill, or the previous replay's value at the same address (the same
// hazard as the FeatureDropout maximum in perturb.hip).  A stale bound is still a usable scale, so nothing breaks, but workgroups
// then scale by different powers of two from run to run and the results differ in the last bits (seen as rare bit mismatches
// between a replayed and an eager run of the same steps).
__device__ __forceinline__ float bound_max(const float* p) {
    float m = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 1; i < UAPS_BOUND_SLOTS; ++i)
        m = __builtin_fmaxf(m, __hip_atomic_load(p + i * UAPS_BOUND_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return m;
}

// block-wide max|v| of the calling threads' values -> atomic max on one slot of a bound (non-negative floats order like their
// bit patterns).  Only blocks that would change the slot issue the atomic (a stale, smaller value read here merely costs a
// redundant atomic).  Every thread of the block must call it; smem_16: 16 floats of LDS.
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
        unsigned* slot = reinterpret_cast<unsigned*>(dst) + ((blockIdx.x + blockIdx.y * 5u) % UAPS_BOUND_SLOTS) * UAPS_BOUND_STRIDE;
        if (m == m && m > __builtin_bit_cast(float, __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            atomicMax(slot, __builtin_bit_cast(unsigned, m));
    }
}
}  // namespace uaps
