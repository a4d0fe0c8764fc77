// In-kernel phase stamps, DIAGNOSTIC BUILD ONLY (make stamps -> uaps_amd/lib/libuaps_hip_stamps.so, -DUAPS_STAMPS; loaded through
// UAPS_HIP_LIB by tools/diag/stamp_table.py).  In the shipped library every macro below is empty: no stamp executes.
//
// A stamped wave keeps, in scalar registers, the shader-clock time (s_memtime) it spent in each of up to 12 phases of a kernel
// body and writes them with the kernel's first / last stamp and the 100 MHz wall clock (s_memrealtime) of both ends into a buffer
// of its own (uaps_debug_set_stamp_buffer; nothing else reads it, no output depends on it).  The in-kernel clock of a
// workgroup is (memtime delta) / (memrealtime delta) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
#pragma once
#include <hip/hip_runtime.h>

#ifdef UAPS_STAMPS
namespace uaps {
constexpr int kStampSlots = 20;                       // per wave: 12 phase sums, [12] first stamp, [13] last stamp, [14] wall start, [15] wall end, [16] XCC id, [17] first-MFMA stamp
static __device__ unsigned long long* g_stamp_buf = nullptr;
static __device__ unsigned long long g_stamp_waves = 0;
struct StampState {
    unsigned long long acc[12];
    unsigned long long t0, last, w0, first_mfma;
    __device__ __forceinline__ void init() {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 0;
        first_mfma = 0;
        w0 = __builtin_amdgcn_s_memrealtime();
        t0 = last = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void mark(int i) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        acc[i] += now - last; last = now;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void note_first_mfma() { if (first_mfma == 0) first_mfma = last; }
    __device__ __forceinline__ void flush() {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* buf = g_stamp_buf;
        if (buf != nullptr && (threadIdx.x & 63) == 0) {
            const unsigned long long wid = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
            if (wid < g_stamp_waves) {
                unsigned long long* p = buf + wid * kStampSlots;
#pragma unroll
                for (int i = 0; i < 12; ++i) p[i] = acc[i];
                p[12] = t0; p[13] = last; p[14] = w0; p[15] = w1;
                p[16] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID[3:0]
                p[17] = first_mfma;
            }
        }
    }
};
}  // namespace uaps
#define UAPS_STAMP_DECL uaps::StampState stamp_; stamp_.init()
#define UAPS_STAMP(i) stamp_.mark(i)
#define UAPS_STAMP_FIRST_MFMA() stamp_.note_first_mfma()
#define UAPS_STAMP_FLUSH() stamp_.flush()
#else
#define UAPS_STAMP_DECL do { } while (0)
#define UAPS_STAMP(i) do { } while (0)
#define UAPS_STAMP_FIRST_MFMA() do { } while (0)
#define UAPS_STAMP_FLUSH() do { } while (0)
#endif
