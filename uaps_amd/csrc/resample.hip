// UpBlock glue for gfx950: bilinear x2 up-sampling (align_corners=True) written straight into the
// channel-concatenated buffer the following ConvBlock reads, and its backward.
//
// Replaces utilities/UAPS_unet.py:83-85 (`x1 = self.up(x1); x = torch.cat([x2, x1], dim=1)`), i.e.
// torch's upsample_bilinear2d_out_frame (660 us per call on the full-resolution layer: it launches
// 1024 threads per *row block*, profiles/r01_baseline_miopen_kernel_stats.csv) + CatArrayBatchedCopy,
// and in the backward upsample_bilinear2d_backward + the slice copies.  Streaming, 16 B per lane.
//
// Interpolation arithmetic follows ATen's area_pixel_compute_scale / upsample_bilinear2d exactly:
//   r = (in-1)/(out-1) (fp32); src = r*o; i0 = (int)src; i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1
//   v = l0h*(l0w*v00 + l1w*v01) + l1h*(l0w*v10 + l1w*v11)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uaps_hip.h"
#include "hints.hpp"
#include "rn_math.hpp"

namespace {
using uaps::mul_rn;
constexpr int kThreads = 256;
inline int grid_for(long work, int cap = 4096) {
    long b = (work + kThreads - 1) / kThreads;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// a0 v0 + a1 v1 with the contraction spelled out (a rounded product, then one fma), so that every kernel that interpolates
// produces the same bits whatever the compiler would have fused
__device__ __forceinline__ float lerp2(float a0, float v0, float a1, float v1) { return __builtin_fmaf(a1, v1, mul_rn(a0, v0)); }

__device__ __forceinline__ float bilerp(const float* __restrict__ p, int w, int h0, int h1, float lh0, float lh1, int w0, int w1,
                                        float lw0, float lw1) {
    const float top = lerp2(lw0, p[(long)h0 * w + w0], lw1, p[(long)h0 * w + w1]);
    const float bot = lerp2(lw0, p[(long)h1 * w + w0], lw1, p[(long)h1 * w + w1]);
    return lerp2(lh0, top, lh1, bot);
}

// out[b, 0:Cs]      = skip[b]                      (copy)
// out[b, Cs:Cs+Cl]  = bilinear_x2(low[b])          (low is [B,Cl,h,w], out/skip are [.,.,2h,2w])
template <bool VEC>
__global__ __launch_bounds__(kThreads) void up_cat_fwd_kernel(const float* __restrict__ skip, const float* __restrict__ low,
                                                              float* __restrict__ out, int B, int Cs, int Cl, int h, int w,
                                                              float rh, float rw) {
    const int H = 2 * h, W = 2 * w;
    const int Ct = Cs + Cl;
    constexpr int V = VEC ? 4 : 1;
    const long Wg = W / V;                               // groups per row
    const long total = (long)B * Ct * H * Wg;
    for (long g = (long)blockIdx.x * kThreads + threadIdx.x; g < total; g += (long)gridDim.x * kThreads) {
        const int xg = (int)(g % Wg);
        long t = g / Wg;
        const int oy = (int)(t % H); t /= H;
        const int c = (int)(t % Ct);
        const int b = (int)(t / Ct);
        const long o = (((long)b * Ct + c) * H + oy) * W + (long)xg * V;
        if (c < Cs) {
            const long si = (((long)b * Cs + c) * H + oy) * W + (long)xg * V;
            if (VEC) *reinterpret_cast<float4*>(out + o) = *reinterpret_cast<const float4*>(skip + si);
            else out[o] = skip[si];
        } else {
            const float* p = low + ((long)b * Cl + (c - Cs)) * h * w;
            const float sy = mul_rn(rh, (float)oy);   // rounded product, as ATen: the fraction is taken from it
            const int h0 = (int)sy, h1 = h0 + (h0 < h - 1 ? 1 : 0);
            const float lh1 = sy - h0, lh0 = 1.f - lh1;
            float v[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int ox = xg * V + k;
                const float sx = mul_rn(rw, (float)ox);
                const int w0 = (int)sx, w1 = w0 + (w0 < w - 1 ? 1 : 0);
                const float lw1 = sx - w0, lw0 = 1.f - lw1;
                v[k] = bilerp(p, w, h0, h1, lh0, lh1, w0, w1, lw0, lw1);
            }
            if (VEC) *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
            else out[o] = v[0];
        }
    }
}

// The same interpolation for a plain up-sampling (no skip part), LDS-tiled: a block owns a (1024/TC) x TC output tile
// of one (b, c) plane, stages the <= 18 x 34 low-resolution patch it reads with coalesced loads, and every thread
// produces 4 consecutive outputs from LDS (16 ds_read_b32 instead of 16 scattered global loads per 16-byte store;
// the gather kernel above is load-instruction-bound at 2.3 TB/s).  Arithmetic and association are bilerp()'s.
template <int TC, int RPT>
__global__ __launch_bounds__(kThreads) void up2x_tiled_kernel(const float* __restrict__ low, float* __restrict__ out, int h, int w,
                                                              float rh, float rw, int tiles_x, int tiles_y, float* __restrict__ amax_out) {
    // RPT row groups per thread (round 3): 4 x fewer, longer workgroups -- the 1024-output form spent its time in launch,
    // barrier and max-reduction overhead (3.0 TB/s on the 256 x 256 maps)
    constexpr int TG = 1024 / TC, TR = TG * RPT, LR = TR / 2 + 2, LC = TC / 2 + 2, LCP = LC + 1;
    __shared__ float sL[LR * LCP];
    const int H = 2 * h, W = 2 * w;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const long plane = bid / tiles_y;
    const int oy0 = ty * TR, ox0 = tx * TC;
    const int sy0 = (int)mul_rn(rh, (float)oy0), sx0 = (int)mul_rn(rw, (float)ox0);
    const float* p = low + plane * h * w;
    for (int u = threadIdx.x; u < LR * LC; u += kThreads) {
        const int r = u / LC, cidx = u % LC;
        const int yy = sy0 + r, xx = sx0 + cidx;
        sL[r * LCP + cidx] = (yy < h && xx < w) ? p[(long)yy * w + xx] : 0.f;
    }
    __syncthreads();
    const int xg = threadIdx.x % (TC / 4), row = threadIdx.x / (TC / 4);
    const int ox = ox0 + xg * 4;
    float am = 0.f;                              // max|output| of this thread (uaps_call_hints::out_amax)
    int w0[4], w1[4];
    float lw0[4], lw1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sx = mul_rn(rw, (float)(ox + k));
        w0[k] = (int)sx; w1[k] = w0[k] + (w0[k] < w - 1 ? 1 : 0);
        lw1[k] = sx - w0[k]; lw0[k] = 1.f - lw1[k];
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
    const int oy = oy0 + row + i * TG;
    if (oy < H && ox < W) {
    const float sy = mul_rn(rh, (float)oy);
    const int h0 = (int)sy, h1 = h0 + (h0 < h - 1 ? 1 : 0);
    const float lh1 = sy - h0, lh0 = 1.f - lh1;
    const float* r0 = &sL[(h0 - sy0) * LCP - sx0];
    const float* r1 = &sL[(h1 - sy0) * LCP - sx0];
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float top = lerp2(lw0[k], r0[w0[k]], lw1[k], r0[w1[k]]);
        const float bot = lerp2(lw0[k], r1[w0[k]], lw1[k], r1[w1[k]]);
        v[k] = lerp2(lh0, top, lh1, bot);
    }
    *reinterpret_cast<float4*>(out + (plane * H + oy) * W + ox) = make_float4(v[0], v[1], v[2], v[3]);
    am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    }
    if (amax_out) {                              // uniform branch
        __shared__ float sm[16];
        uaps::block_amax_to(amax_out, am, sm);
    }
}

// d_skip = dout[:, :Cs] (copy) ;  d_low = transpose of the interpolation applied to dout[:, Cs:].
// The transpose is a gather: low-res pixel (iy, ix) collects from the output rows/cols whose source
// index touches it (at most 7 candidates per axis), so no atomics and a fixed summation order:
//   d_low[iy][ix] = sum_oy wy(oy, iy) * ( sum_ox wx(ox, ix) * dout[oy][ox] ).
// A block owns an 8 x 32 low-res tile of one (b, c) plane: it stages the 22 x 72 high-res patch in LDS with
// coalesced 16-byte loads, reduces it along x into T[22][32] (each thread's 7 column weights are computed
// once), then along y.  Same association as the plain double loop, 26 LDS reads per output instead of 49
// scattered global ones.
// Round 3: LY = 32 low-resolution rows per block where the map has them (70 patch rows for 64 own: 9 % halo instead of 37 %,
// four outputs per thread, a quarter of the workgroups); LY = 8 is the round-1 form.
constexpr int kLx = 32, kPc = 2 * kLx + 8;   // tile / patch columns (x origin 2*ix0 - 4)

__device__ __forceinline__ float tap_weight(float r, int o, int n_out, int n_in, int i) {
    if (o < 0 || o >= n_out) return 0.f;
    const float sx = mul_rn(r, (float)o);   // rounded product, as ATen: the fraction is taken from it
    const int i0 = (int)sx, i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    const float l1 = sx - i0, l0 = 1.f - l1;
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

template <int kLy>
__global__ __launch_bounds__(kThreads) void up_cat_bwd_low_kernel(const float* __restrict__ dout, float* __restrict__ dlow, int B,
                                                                  int Cs, int Cl, int h, int w, float rh, float rw,
                                                                  int tiles_x, int tiles_y) {
    constexpr int kPr = 2 * kLy + 6;             // patch rows
    __shared__ __attribute__((aligned(16))) float sA[kPr * kPc];
    __shared__ float sT[kPr * kLx];
    __shared__ float sWx[kLx][7], sWy[kLy][7];      // interpolation weights of the tile's columns / rows, computed once
    const int H = 2 * h, W = 2 * w, Ct = Cs + Cl;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int c = bid % Cl, b = bid / Cl;
    const int iy0 = ty * kLy, ix0 = tx * kLx;
    const int oy0 = 2 * iy0 - 3, ox0 = 2 * ix0 - 4;
    const float* p = dout + (((long)b * Ct + Cs + c) * H) * W;
    const bool vec = (W % 4) == 0;
    // ---- stage the patch (zero outside the image) ----
    for (int u = threadIdx.x; u < kPr * (kPc / 4); u += kThreads) {
        const int r = u / (kPc / 4), x4 = (u % (kPc / 4)) * 4;
        const int oy = oy0 + r, ox = ox0 + x4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oy >= 0 && oy < H) {
            if (vec && ox >= 0 && ox + 3 < W) {
                v = *reinterpret_cast<const float4*>(p + (long)oy * W + ox);
            } else {
                const float* q = p + (long)oy * W;
                if (ox >= 0 && ox < W) v.x = q[ox];
                if (ox + 1 >= 0 && ox + 1 < W) v.y = q[ox + 1];
                if (ox + 2 >= 0 && ox + 2 < W) v.z = q[ox + 2];
                if (ox + 3 >= 0 && ox + 3 < W) v.w = q[ox + 3];
            }
        }
        *reinterpret_cast<float4*>(&sA[r * kPc + x4]) = v;
    }
    // one weight per thread (the per-thread form spent ~300 VALU ops per output on 14 weights and was VALU-bound)
    if (threadIdx.x < kLx * 7) {
        const int l = threadIdx.x / 7, k = threadIdx.x % 7;
        sWx[l][k] = tap_weight(rw, 2 * (ix0 + l) - 3 + k, W, w, ix0 + l);
    }
    if (threadIdx.x < kLy * 7) {
        const int l = threadIdx.x / 7, k = threadIdx.x % 7;
        sWy[l][k] = tap_weight(rh, 2 * (iy0 + l) - 3 + k, H, h, iy0 + l);
    }
    const int lx = threadIdx.x % kLx, ly0 = threadIdx.x / kLx;
    const int ix = ix0 + lx;
    __syncthreads();
    float wx[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) wx[k] = sWx[lx][k];
    // ---- reduce along x: T[r][lx] = sum_k wx[k] * A[r][2*lx + 1 + k]   (2*ix - 3 + k - ox0 = 2*lx + 1 + k) ----
    for (int r = ly0; r < kPr; r += kThreads / kLx) {
        // columns 2*lx .. 2*lx + 7 as four 8-byte reads: lanes are 2 floats apart, so ds_read_b64 covers all 64 banks once
        // (seven ds_read_b32 at that stride are 2-way conflicted); a[k] = column 2*lx + 1 + k
        const float2* a2 = reinterpret_cast<const float2*>(&sA[r * kPc + 2 * lx]);
        const float2 p0 = a2[0], p1 = a2[1], p2 = a2[2], p3 = a2[3];
        const float a[7] = {p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
        float row = 0.f;
#pragma unroll
        for (int k = 0; k < 7; ++k) row += wx[k] * a[k];       // taps outside the footprint have weight 0 (patch is zero-filled: finite)
        sT[r * kLx + lx] = row;
    }
    __syncthreads();
    // ---- reduce along y: rows 2*iy - 3 + k - oy0 = 2*ly + k ----
#pragma unroll
    for (int j = 0; j < kLy / 8; ++j) {
        const int ly = ly0 + 8 * j, iy = iy0 + ly;
        if (ix < w && iy < h) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 7; ++k) acc += sWy[ly][k] * sT[(2 * ly + k) * kLx + lx];
            dlow[(((long)b * Cl + c) * h + iy) * w + ix] = acc;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void slice_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                                  int Ct, int c0, int Cn, long HW) {
    constexpr int V = VEC ? 4 : 1;
    const long per = HW / V;
    const long total = (long)B * Cn * per;
    for (long g = (long)blockIdx.x * kThreads + threadIdx.x; g < total; g += (long)gridDim.x * kThreads) {
        const long i = g % per;
        const long t = g / per;
        const int c = (int)(t % Cn);
        const int b = (int)(t / Cn);
        const long s = (((long)b * Ct + c0 + c) * HW) + i * V, d = (((long)b * Cn + c) * HW) + i * V;
        if (VEC) *reinterpret_cast<float4*>(dst + d) = *reinterpret_cast<const float4*>(src + s);
        else dst[d] = src[s];
    }
}
}  // namespace

static int up_cat_fwd_impl(float* amax_out, const float* skip, const float* low, float* out, int B, int Cs, int Cl, int h, int w, uaps_stream_t stream);
extern "C" int uaps_up_cat_fwd(const float* skip, const float* low, float* out, int B, int Cs, int Cl, int h, int w,
                               uaps_stream_t stream) {
    return up_cat_fwd_impl(uaps::take_hints().out_amax, skip, low, out, B, Cs, Cl, h, w, stream);
}
// (the *_h form: the hints of THIS call as the first argument, nothing thread-local -- see conv_fwd.hip)
extern "C" int uaps_up_cat_fwd_h(const uaps_call_hints* hints, const float* skip, const float* low, float* out, int B, int Cs, int Cl, int h,
                                 int w, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, hh);
    return up_cat_fwd_impl(hh.out_amax, skip, low, out, B, Cs, Cl, h, w, stream);
}
static int up_cat_fwd_impl(float* amax_out, const float* skip, const float* low, float* out, int B, int Cs, int Cl, int h, int w, uaps_stream_t stream) {
    if (!skip || !low || !out || B <= 0 || Cs < 0 || Cl <= 0 || h <= 0 || w <= 0) return UAPS_EINVAL;
    const int H = 2 * h, W = 2 * w;
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * (Cs + Cl) * H * W;
    uaps::account_bytes(4.0 * ((double)B * Cl * h * w + 2.0 * B * Cs * H * W + (double)B * Cl * H * W));      // low once, (skip copied,) the 4x larger output once
    if (amax_out && !(Cs == 0 && W % 4 == 0 && al16(out))) return UAPS_ERANGE;      // only the plain up-sampling form tracks max|out|
    if (Cs == 0 && W % 4 == 0 && al16(out)) {            // plain up-sampling: the LDS-tiled kernel
        const bool wide = W >= 64, tall = wide && H >= 64;      // 64 x 64 output tiles (4 row groups per thread) where the map holds one
        const int TC = wide ? 64 : 32, TR = (1024 / TC) * (tall ? 4 : 1);
        const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR;
        const long nblk = (long)B * Cl * tiles_x * tiles_y;
        if (nblk > 0x7fffffffL) return UAPS_ERANGE;
        if (tall) hipLaunchKernelGGL((up2x_tiled_kernel<64, 4>), dim3((unsigned)nblk), dim3(kThreads), 0, s, low, out, h, w, rh, rw, tiles_x, tiles_y, amax_out);
        else if (wide) hipLaunchKernelGGL((up2x_tiled_kernel<64, 1>), dim3((unsigned)nblk), dim3(kThreads), 0, s, low, out, h, w, rh, rw, tiles_x, tiles_y, amax_out);
        else hipLaunchKernelGGL((up2x_tiled_kernel<32, 1>), dim3((unsigned)nblk), dim3(kThreads), 0, s, low, out, h, w, rh, rw, tiles_x, tiles_y, amax_out);
        return (int)hipGetLastError();
    }
    if (W % 4 == 0 && al16(skip) && al16(out))
        hipLaunchKernelGGL(up_cat_fwd_kernel<true>, dim3(grid_for(total / 4)), dim3(kThreads), 0, s, skip, low, out, B, Cs, Cl, h, w, rh, rw);
    else
        hipLaunchKernelGGL(up_cat_fwd_kernel<false>, dim3(grid_for(total)), dim3(kThreads), 0, s, skip, low, out, B, Cs, Cl, h, w, rh, rw);
    return (int)hipGetLastError();
}

extern "C" int uaps_up_cat_bwd(const float* dout, float* dskip, float* dlow, int B, int Cs, int Cl, int h, int w,
                               uaps_stream_t stream) {
    if (!dout || !dlow || B <= 0 || Cs < 0 || Cl <= 0 || h <= 0 || w <= 0) return UAPS_EINVAL;
    const int H = 2 * h, W = 2 * w;
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    hipStream_t s = (hipStream_t)stream;
    uaps::account_bytes(4.0 * ((double)B * Cl * H * W + (double)B * Cl * h * w + (dskip ? 2.0 * B * Cs * H * W : 0.0)));
    if (dskip && Cs > 0) {
        const long HW = (long)H * W;
        if (HW % 4 == 0 && al16(dout) && al16(dskip))
            hipLaunchKernelGGL(slice_channels_kernel<true>, dim3(grid_for((long)B * Cs * HW / 4)), dim3(kThreads), 0, s, dout, dskip, B, Cs + Cl, 0, Cs, HW);
        else
            hipLaunchKernelGGL(slice_channels_kernel<false>, dim3(grid_for((long)B * Cs * HW)), dim3(kThreads), 0, s, dout, dskip, B, Cs + Cl, 0, Cs, HW);
    }
    const int ly = h >= 32 ? 32 : 8;
    const int tiles_x = (w + kLx - 1) / kLx, tiles_y = (h + ly - 1) / ly;
    const long nblk = (long)B * Cl * tiles_x * tiles_y;
    if (nblk > 0x7fffffffL) return UAPS_ERANGE;
    if (ly == 32) hipLaunchKernelGGL(up_cat_bwd_low_kernel<32>, dim3((unsigned)nblk), dim3(kThreads), 0, s, dout, dlow, B, Cs, Cl, h, w, rh, rw, tiles_x, tiles_y);
    else hipLaunchKernelGGL(up_cat_bwd_low_kernel<8>, dim3((unsigned)nblk), dim3(kThreads), 0, s, dout, dlow, B, Cs, Cl, h, w, rh, rw, tiles_x, tiles_y);
    return (int)hipGetLastError();
}
