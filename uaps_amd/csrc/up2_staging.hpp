// Bilinear x2 up-sampling (align_corners = True) formed WHILE A ROW KERNEL STAGES ITS OPERAND (round 5): the up-sampled half of an
// UpBlock's concatenation (utilities/UAPS_unet.py:74-75, 83-85: `x1 = self.up(x1); x = torch.cat([x2, x1], dim=1)`) is never
// written -- the convolution reads the LOW-resolution tensor [B, C, H/2, W/2] and interpolates the rows it stages.
//
// A staging lane owns 4 consecutive output pixels x = 4 l .. 4 l + 3 of one row.  With r = (w - 1) / (W - 1), W = 2 w, the source
// column of output x is w0 = (int)(r x): m - 1 for x = 2 m (m >= 1), m for x = 2 m + 1, 0 for x = 0 (the launcher checks this against
// the fp32 arithmetic for the given width, up2_pattern_ok).  The lane's four outputs therefore draw on the low columns 2 l - 1 .. 2 l + 2
// at compile-time positions (0,1) (1,2) (1,2) (2,3) -- lane 0, which has no column -1, takes (1,2) for its first pixel.  Each lane
// FETCHES only columns 2 l and 2 l + 1 (one aligned 8-byte load: a wave-instruction is exactly one 512-byte low row of a channel) and gets
// column 2 l - 1 from lane l - 1 and column 2 l + 2 from lane l + 1 by two lane shifts; the column behind the row's end that the last
// lane would want carries weight exactly 0.  Arithmetic and association are resample.hip's bilerp (ATen's upsample_bilinear2d):
// bit-identical to the materialised tensor.
#pragma once
#include "rn_math.hpp"

namespace uaps {

typedef float up2_f32x4 __attribute__((ext_vector_type(4)));

struct Up2Lane { float lw0[4], lw1[4]; bool first; };

// lane_global: index of the lane's 4-pixel group in the row (x = 4 * lane_global)
__device__ __forceinline__ Up2Lane up2_lane(float rw, int lane_global) {
    Up2Lane L;
    // (rw as an opaque per-lane value, one scalar multiply per pixel: from the kernel-argument scalar pair the compiler packed the
    // four products into v_pk_mul_f32 with the scalar broadcast by op_sel -- a form tools/isa_lint.py keeps out of the library)
    float r = rw;
    asm volatile("" : "+v"(r));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float x = (float)(4 * lane_global + p);
        asm volatile("" : "+v"(x));
        const float sx = mul_rn(r, x);
        const int w0 = (int)sx;
        L.lw1[p] = sx - (float)w0; L.lw0[p] = 1.f - L.lw1[p];
    }
    L.first = lane_global == 0;
    return L;
}
__device__ __forceinline__ float up2_lerp2(float a0, float v0, float a1, float v1) { return __builtin_fmaf(a1, v1, mul_rn(a0, v0)); }

// lane i <- lane i - 1 (wave_shr:1; lane 0 <- 0) and lane i <- lane i + 1 (wave_shl:1; lane 63 <- 0): GFX9 whole-wave DPP shifts
__device__ __forceinline__ float up2_lane_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float up2_lane_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// a, b: the lane's 8-byte loads (low columns 2 l, 2 l + 1) of the two source rows h0, h1; (lh0, lh1): the row weights
typedef float up2_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ up2_f32x4 up2_row(const up2_f32x2& a, const up2_f32x2& b, const Up2Lane& L, float lh0, float lh1) {
    // columns 2 l - 1 (lane l - 1's second element; lane 0 gets 0, unused) and 2 l + 2 (lane l + 1's first; lane 63 gets 0, weight 0):
    // one v_mov_b32 with a whole-wave DPP shift each (the first build used __shfl_up / __shfl_down: 32 ds_bpermute round trips per
    // staged row pair made the up-sampling kernel 24 % slower than the plain one)
    const up2_f32x4 A{up2_lane_prev(a[1]), a[0], a[1], up2_lane_next(a[0])};
    const up2_f32x4 B{up2_lane_prev(b[1]), b[0], b[1], up2_lane_next(b[0])};
    up2_f32x4 v;
    const float a00 = L.first ? A[1] : A[0], a01 = L.first ? A[2] : A[1];
    const float b00 = L.first ? B[1] : B[0], b01 = L.first ? B[2] : B[1];
    v[0] = up2_lerp2(lh0, up2_lerp2(L.lw0[0], a00, L.lw1[0], a01), lh1, up2_lerp2(L.lw0[0], b00, L.lw1[0], b01));
    v[1] = up2_lerp2(lh0, up2_lerp2(L.lw0[1], A[1], L.lw1[1], A[2]), lh1, up2_lerp2(L.lw0[1], B[1], L.lw1[1], B[2]));
    v[2] = up2_lerp2(lh0, up2_lerp2(L.lw0[2], A[1], L.lw1[2], A[2]), lh1, up2_lerp2(L.lw0[2], B[1], L.lw1[2], B[2]));
    v[3] = up2_lerp2(lh0, up2_lerp2(L.lw0[3], A[2], L.lw1[3], A[3]), lh1, up2_lerp2(L.lw0[3], B[2], L.lw1[3], B[3]));
    return v;
}

// host: do the fp32 source columns of every output x of a W = 2 w row follow the static pattern above?
inline bool up2_pattern_ok(int w) {
    const int W = 2 * w;
    if (w < 2) return false;
    const float rw = (float)(w - 1) / (float)(W - 1);
    for (int x = 0; x < W; ++x) {
        volatile float sx = rw * (float)x;
        const int w0 = (int)sx, m = x / 2;
        const int want = (x % 2 == 0) ? (m > 0 ? m - 1 : 0) : m;
        if (w0 != want || w0 > w - 1) return false;
        if (w0 == w - 1 && sx - (float)w0 != 0.f) return false;      // the clamped neighbour must carry weight 0
    }
    return true;
}

}  // namespace uaps
