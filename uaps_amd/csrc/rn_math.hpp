// Separately rounded fp32 multiply / add.  hipcc's default -ffp-contract=fast-honor-pragmas fuses
// a*b+c into one v_fma_f32 (a single rounding), and HIP's __fmul_rn/__fadd_rn are plain operators
// that get fused too.  Where the reference's result is defined by two tensor ops (two roundings),
// e.g. `x.mul(noise) + x` (UAPS_unet.py:180) or `w0*p0 + w1*p1` (UAPS_train.py:252), use these.
#pragma once
#include <hip/hip_runtime.h>
namespace uaps {
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
}  // namespace uaps
