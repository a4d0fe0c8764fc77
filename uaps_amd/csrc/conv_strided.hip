// General (strided) fp32 convolutions and the 3x3 / stride-2 max-pool of the ResNet stem for gfx950: the three strided layers
// of the reference's dilated ResNet-50 -- utilities/resnet.py:120 (7x7 stride 2 pad 3 stem), :124 (max-pool 3x3 stride 2 pad 1),
// :147 + :8-14 (layer2's 3x3 stride-2 convolution and its 1x1 stride-2 shortcut) -- in forward, input-gradient and
// weight-gradient form, so that no library convolution or pooling is left on the ResNet-encoder path.
//
// These layers are a few per cent of that network's work, so the kernels favour generality (any odd kernel size up to 7,
// stride 1 or 2, any padding) over the last factor of two: implicit GEMMs on the exact-f32 matrix instruction
// v_mfma_f32_16x16x4_f32 whose fragments come straight from global memory / L2 (no LDS staging) in the forward and
// input-gradient kernels -- with 16-32 accumulator registers per wave eight waves per SIMD hide the load latency -- and an
// LDS-staged pixel-split kernel for the weight gradient (fixed-order slab reduction, no float atomics).
//   forward        M = 16 consecutive output pixels of a row, N = output channels, K = (tap, input channel)
//   input gradient M = 16 input pixels of one row and one column-parity class (ix = ix0 + stride*j): all of them see the same
//                  set of valid taps, so a stride-2 layer does no work on the 3/4 of (pixel, tap) pairs that never meet
//   weight grad.   M = output channels, N = input channels, K = output pixels, one accumulator tile per tap of a tap group
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uaps_hip.h"
#include "conv_kernels.hpp"
#include "hints.hpp"
using namespace uaps;

namespace {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct SConvArgs {
    const float* in;     // forward: x [B,Cin,H,W]; input gradient: dy [B,Cout,OH,OW]
    const float* wp;     // forward: wf [taps][CinP4][CoutP16]; input gradient: wb [taps][CoutP4][CinP16] (taps NOT flipped)
    float* out;          // forward: y [B,Cout,OH,OW]; input gradient: dx [B,Cin,H,W]
    int B, Cin, Cout, H, W, OH, OW, ks, stride, pad, CinP, CoutP;
};

// ---- forward ----------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void convs_fwd_kernel(SConvArgs a, int tiles_x, int ncb, long nitems) {
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nitems) return;
    long t = wid;
    const int nb = (int)(t % ncb); t /= ncb;
    const int xt = (int)(t % tiles_x); t /= tiles_x;
    const int oy = (int)(t % a.OH);
    const int b = (int)(t / a.OH);
    const int ox0 = xt * 16, co0 = nb * 16 * NW;
    f32x4 acc[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xb = a.in + (size_t)b * a.Cin * a.H * a.W;
    const size_t HW = (size_t)a.H * a.W;
    for (int ky = 0; ky < a.ks; ++ky) {
        const int iy = oy * a.stride + ky - a.pad;
        if (iy < 0 || iy >= a.H) continue;                    // wave-uniform
        for (int kx = 0; kx < a.ks; ++kx) {
            const int ix = (ox0 + j) * a.stride + kx - a.pad;
            const bool okx = ix >= 0 && ix < a.W && ox0 + j < a.OW;
            const float* xp = xb + (size_t)iy * a.W + (okx ? ix : 0);
            const float* wrow = a.wp + ((size_t)(ky * a.ks + kx) * a.CinP) * a.CoutP + co0 + j;
            for (int c0 = 0; c0 < a.CinP; c0 += 4) {
                const int ci = c0 + kq;
                const float av = (okx && ci < a.Cin) ? xp[(size_t)ci * HW] : 0.f;
#pragma unroll
                for (int n = 0; n < NW; ++n)          // a channel block past CoutP (CoutP not a multiple of 16 NW) re-reads block 0: never stored
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wrow[(size_t)ci * a.CoutP + (co0 + n * 16 < a.CoutP ? n * 16 : 0)], acc[n], 0, 0, 0);
            }
        }
    }
    // lane (j, kq) holds pixels ox0 + 4 kq .. + 3 of channel co0 + 16 n + j
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int co = co0 + n * 16 + j;
        if (co >= a.Cout) continue;
        float* yp = a.out + (((size_t)b * a.Cout + co) * a.OH + oy) * a.OW + ox0 + kq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (ox0 + kq * 4 + r < a.OW) yp[r] = acc[n][r];
    }
}

// ---- input gradient -------------------------------------------------------------------------------------------------------
// work item = (image, input row, column-parity class, tile of 16 same-parity columns, block of 16*NW input channels)
template <int NW>
__global__ __launch_bounds__(256) void convs_bwd_data_kernel(SConvArgs a, int tiles_x, int ncb, long nitems) {
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nitems) return;
    long t = wid;
    const int nb = (int)(t % ncb); t /= ncb;
    const int xt = (int)(t % tiles_x); t /= tiles_x;
    const int px = (int)(t % a.stride); t /= a.stride;
    const int iy = (int)(t % a.H);
    const int b = (int)(t / a.H);
    const int s = a.stride, ix0 = px + xt * 16 * s, ci0 = nb * 16 * NW;
    f32x4 acc[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* dyb = a.in + (size_t)b * a.Cout * a.OH * a.OW;
    const size_t OHW = (size_t)a.OH * a.OW;
    for (int ky = 0; ky < a.ks; ++ky) {
        const int ny = iy + a.pad - ky;
        if (ny < 0 || ny % s) continue;                       // wave-uniform
        const int oy = ny / s;
        if (oy >= a.OH) continue;
        for (int kx = 0; kx < a.ks; ++kx) {
            const int nx0 = ix0 + a.pad - kx;                 // the same parity for every lane: ix = ix0 + s*j
            if (((nx0 % s) + s) % s) continue;                // wave-uniform
            const int ox = (nx0 >= 0 ? nx0 / s : -((-nx0) / s)) + j;       // nx0 is a multiple of s here
            const bool okx = ox >= 0 && ox < a.OW && ix0 + s * j < a.W;
            const float* dp = dyb + (size_t)oy * a.OW + (okx ? ox : 0);
            const float* wrow = a.wp + ((size_t)(ky * a.ks + kx) * a.CoutP) * a.CinP + ci0 + j;
            for (int c0 = 0; c0 < a.CoutP; c0 += 4) {
                const int co = c0 + kq;
                const float av = (okx && co < a.Cout) ? dp[(size_t)co * OHW] : 0.f;
#pragma unroll
                for (int n = 0; n < NW; ++n)          // as in the forward: blocks past CinP re-read block 0 and are never stored
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wrow[(size_t)co * a.CinP + (ci0 + n * 16 < a.CinP ? n * 16 : 0)], acc[n], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int ci = ci0 + n * 16 + j;
        if (ci >= a.Cin) continue;
        float* xp = a.out + (((size_t)b * a.Cin + ci) * a.H + iy) * a.W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ix = ix0 + s * (kq * 4 + r);
            if (ix < a.W) xp[ix] = acc[n][r];
        }
    }
}

// ---- weight packing for the two kernels above ----------------------------------------------------------------------------
__global__ void convs_pack_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb, int Cout, int Cin, int taps,
                                  int CinP4, int CoutP16, int CoutP4, int CinP16) {
    const long nf = (long)taps * CinP4 * CoutP16, nbk = (long)taps * CoutP4 * CinP16;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nf + nbk; e += (long)gridDim.x * blockDim.x) {
        if (e < nf) {
            const int co = (int)(e % CoutP16); const long r = e / CoutP16;
            const int ci = (int)(r % CinP4), t = (int)(r / CinP4);
            wf[e] = (co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * taps + t] : 0.f;
        } else {
            const long f = e - nf;
            const int ci = (int)(f % CinP16); const long r = f / CinP16;
            const int co = (int)(r % CoutP4), t = (int)(r / CoutP4);
            wb[f] = (co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * taps + t] : 0.f;
        }
    }
}

// ---- weight gradient --------------------------------------------------------------------------------------------------------
// dw[co][ci][ky][kx] = sum_{b,oy,ox} dy[b][co][oy][ox] * x[b][ci][oy*s + ky - p][ox*s + kx - p].
// A workgroup owns a 16 x 16 (co, ci) block, one tap group (GH x GW taps starting at (gy0, gx0): the whole 3x3 / 1x1 kernel,
// or one row of a 7x7 kernel) and every nsplit-th tile of 4 output rows x 32 output columns; its 4 waves take one row each.
struct SWrwArgs {
    const float* dy; const float* x; float* slab;    // slab [nsplit][taps][CoutS][CinS]
    int B, Cin, Cout, H, W, OH, OW, ks, stride, pad, CoutS, CinS, ncob, ncib, nsplit, tiles_x, tiles_y;
};
template <int GH, int GW, int S>
__global__ __launch_bounds__(256) void convs_wrw_kernel(SWrwArgs a) {
    constexpr int TH = 4, TW = 32, IH = (TH - 1) * S + GH, IW = (TW - 1) * S + GW, NT = GH * GW;
    constexpr int PSD = TH * TW + 2, PSI = IH * IW + ((IH * IW) % 2 ? 0 : 1);       // plane strides off the power-of-two banks
    __shared__ float sD[16 * PSD];
    __shared__ float sX[16 * PSI];
    __shared__ float red[3 * NT * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    int bid = blockIdx.x;
    const int cib = bid % a.ncib; bid /= a.ncib;
    const int cob = bid % a.ncob;
    const int split = bid / a.ncob;
    const int g = blockIdx.y;                                 // tap group
    const int groups_x = (a.ks + GW - 1) / GW;
    const int gy0 = (g / groups_x) * GH, gx0 = (g % groups_x) * GW;
    const int co0 = cob * 16, ci0 = cib * 16;
    const int tiles_per_img = a.tiles_x * a.tiles_y, ntiles = a.B * tiles_per_img;
    const int t_begin = (int)((long)ntiles * split / a.nsplit), t_end = (int)((long)ntiles * (split + 1) / a.nsplit);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / tiles_per_img, tt = tile % tiles_per_img;
        const int oy0 = (tt / a.tiles_x) * TH, ox0 = (tt % a.tiles_x) * TW;
        const int iy0 = oy0 * S + gy0 - a.pad, ix0 = ox0 * S + gx0 - a.pad;
        for (int e = tid; e < 16 * TH * TW; e += 256) {
            const int c = e / (TH * TW), r = (e / TW) % TH, col = e % TW;
            const int co = co0 + c, oy = oy0 + r, ox = ox0 + col;
            sD[c * PSD + r * TW + col] = (co < a.Cout && oy < a.OH && ox < a.OW) ? a.dy[(((size_t)b * a.Cout + co) * a.OH + oy) * a.OW + ox] : 0.f;
        }
        for (int e = tid; e < 16 * IH * IW; e += 256) {
            const int c = e / (IH * IW), r = (e / IW) % IH, col = e % IW;
            const int ci = ci0 + c, iy = iy0 + r, ix = ix0 + col;
            sX[c * PSI + r * IW + col] = (ci < a.Cin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? a.x[(((size_t)b * a.Cin + ci) * a.H + iy) * a.W + ix] : 0.f;
        }
        __syncthreads();
        const float* pa = sD + j * PSD + wave * TW + kq;
        const float* pb = sX + j * PSI + (wave * S) * IW + kq * S;
#pragma unroll
        for (int st = 0; st < TW / 4; ++st) {
            const float av = pa[st * 4];
#pragma unroll
            for (int ty = 0; ty < GH; ++ty)
#pragma unroll
                for (int tx = 0; tx < GW; ++tx)
                    acc[ty * GW + tx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, pb[ty * IW + st * 4 * S + tx], acc[ty * GW + tx], 0, 0, 0);
        }
        __syncthreads();
    }
    // sum the 4 row-waves in a fixed order through LDS
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave - 1) * NT + t) * 256 + r * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave != 0) return;
    float* slab = a.slab + (size_t)split * a.ks * a.ks * a.CoutS * a.CinS;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ky = gy0 + t / GW, kx = gx0 + t % GW;
        if (ky >= a.ks || kx >= a.ks) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[t][r];
#pragma unroll
            for (int w = 0; w < 3; ++w) v += red[(w * NT + t) * 256 + r * 64 + lane];
            // lane (j, kq), register r: co = co0 + kq*4 + r, ci = ci0 + j
            slab[((size_t)(ky * a.ks + kx) * a.CoutS + co0 + kq * 4 + r) * a.CinS + ci0 + j] = v;
        }
    }
}

// ---- weight gradient of the stem: 7x7 / stride 2 / padding 3 with <= 3 input channels (utilities/resnet.py:120) -------------
// GEMM view: M = output channels, N = (ci, ky, kx) = Cin * 49 <= 147 columns (ten 16-wide tiles), K = output pixels.  The
// general kernel above pads 3 input channels to a 16-channel block and restages the input for each of the 7 tap rows; here a
// workgroup stages the 13 x 69 x Cin input patch of a 4 x 32 output tile once, and wave w owns output channels [16w, 16w + 16)
// with all ten column tiles (40 accumulator registers), so no cross-wave sum is needed.  dy is read exactly once, straight
// from global memory as 16-byte pieces: lane (j, kq) holds pixels 4kq .. 4kq + 3 of channel j, and MFMA step s contracts
// pixel 4kq + s of every k-group (the order inside K is free as long as both operands agree), which the B operand follows by
// reading x[ci][2 oy + ky - 3][2 (4kq + s) + kx - 3] from LDS: per-lane column offsets (10 registers) plus immediates.
// Partials: slab [split][CoutS][160]; stem7_reduce_kernel sums the splits in a fixed order into dw [Cout][Cin][7][7].
struct StemWrwArgs {
    const float* dy; const float* x; float* slab;
    int B, Cin, Cout, H, W, OH, OW, CoutS, nsplit, tiles_x, tiles_y;
};
constexpr int kStemN = 160;
__global__ __launch_bounds__(256) void stem7_wrw_kernel(StemWrwArgs a) {
    constexpr int TH = 4, TW = 32, IH = 2 * (TH - 1) + 7, IW = 2 * (TW - 1) + 7, PS = IH * IW, NT = kStemN / 16;
    __shared__ float sX[3 * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const int split = blockIdx.x, co0 = blockIdx.y * 64 + wave * 16;
    int off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + j;
        const int ci = n / 49, t = n % 49;
        off[nt] = (n < a.Cin * 49 ? ci * PS + (t / 7) * IW + t % 7 : 0) + 8 * kq;      // columns past Cin * 49 are never written out
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tiles_per_img = a.tiles_x * a.tiles_y, ntiles = a.B * tiles_per_img;
    const int t_begin = (int)((long)ntiles * split / a.nsplit), t_end = (int)((long)ntiles * (split + 1) / a.nsplit);
    const int co = co0 + j;
    const bool co_ok = co < a.Cout;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / tiles_per_img, tt = tile % tiles_per_img;
        const int oy0 = (tt / a.tiles_x) * TH, ox0 = (tt % a.tiles_x) * TW;
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        __syncthreads();                              // the previous tile's fragments are read
        for (int e = tid; e < 3 * PS; e += 256) {
            const int c = e / PS, r = (e % PS) / IW, col = e % IW;
            const int iy = iy0 + r, ix = ix0 + col;
            sX[e] = (c < a.Cin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? a.x[(((size_t)b * a.Cin + c) * a.H + iy) * a.W + ix] : 0.f;
        }
        __syncthreads();
        const float* dyp = a.dy + ((size_t)b * a.Cout + (co_ok ? co : 0)) * a.OH * a.OW;
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            const int oy = oy0 + r;
#pragma unroll
            for (int g = 0; g < TW / 16; ++g) {
                const int px = ox0 + g * 16 + 4 * kq;          // OW % 4 == 0: the four pixels are all in or all out
                const f32x4 av = (co_ok && oy < a.OH && px < a.OW) ? *reinterpret_cast<const f32x4*>(dyp + (size_t)oy * a.OW + px)
                                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 4; ++st) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], sX[off[nt] + 2 * r * IW + 32 * g + 2 * st], acc[nt], 0, 0, 0);
                }
            }
        }
    }
    // lane (j, kq), register r: output channel co0 + 4 kq + r, column nt * 16 + j
    float* slab = a.slab + (size_t)split * a.CoutS * kStemN;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(size_t)(co0 + 4 * kq + r) * kStemN + nt * 16 + j] = acc[nt][r];
}
// dw[co][n] (n = ci * 49 + tap < Cin * 49) = sum over the splits, in split order: 32 outputs x 8 split groups per workgroup
__global__ __launch_bounds__(256) void stem7_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit, int Cout, int CoutS,
                                                           int ncol) {
    __shared__ float part[8][32];
    const int o = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    const bool ok = o < Cout * ncol;
    const int co = ok ? o / ncol : 0, n = ok ? o % ncol : 0;
    const int s0 = (int)((long)nsplit * grp / 8), s1 = (int)((long)nsplit * (grp + 1) / 8);
    float v = 0.f;
    for (int sp = s0; sp < s1; ++sp) v += slab[((size_t)sp * CoutS + co) * kStemN + n];
    part[grp][threadIdx.x & 31] = v;
    __syncthreads();
    if (grp == 0 && ok) {
        float t = part[0][threadIdx.x];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += part[g][threadIdx.x];
        dw[o] = t;
    }
}

// ---- 3x3 stride-2 pad-1 max-pool (utilities/resnet.py:124) ------------------------------------------------------------------
// forward: y and the position (0..8, row-major in the window, first maximum wins like torch) of the arg-max of each window;
// backward: every input pixel gathers the gradients of the <= 4 windows whose arg-max it is (fixed order, no atomics).
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                                             long planes, int H, int W, int OH, int OW) {
    const long n = planes * OH * OW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % OW), oy = (int)((e / OW) % OH);
        const long pl = e / ((long)OW * OH);
        const float* xp = x + pl * H * W;
        float best = -INFINITY;
        int bi = 0;
        bool any = false;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float v = xp[(long)iy * W + ix];
                if (!any || v > best || (v != v && best == best)) { best = v; bi = ky * 3 + kx; any = true; }     // NaN propagates like torch
            }
        y[e] = best;
        idx[e] = (uint8_t)bi;
    }
}
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                             long planes, int H, int W, int OH, int OW) {
    const long n = planes * H * W;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int ix = (int)(e % W), iy = (int)((e / W) % H);
        const long pl = e / ((long)W * H);
        float s = 0.f;
        // windows (oy, ox) with oy*2 - 1 <= iy <= oy*2 + 1
        for (int oy = (iy + 1) / 2 - ((iy + 1) % 2 == 0 ? 1 : 0); oy <= (iy + 1) / 2; ++oy) {
            if (oy < 0 || oy >= OH) continue;
            const int ky = iy - (oy * 2 - 1);
            if (ky < 0 || ky > 2) continue;
            for (int ox = (ix + 1) / 2 - ((ix + 1) % 2 == 0 ? 1 : 0); ox <= (ix + 1) / 2; ++ox) {
                if (ox < 0 || ox >= OW) continue;
                const int kx = ix - (ox * 2 - 1);
                if (kx < 0 || kx > 2) continue;
                const long o = (pl * OH + oy) * OW + ox;
                if (idx[o] == ky * 3 + kx) s += dy[o];
            }
        }
        dx[e] = s;
    }
}

int check_sconv(int B, int Cin, int Cout, int H, int W, int ks, int stride, int pad) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (ks < 1 || ks > 7 || !(ks & 1) || (stride != 1 && stride != 2) || pad < 0 || pad > ks / 2 + 3) return UAPS_ERANGE;
    if ((H + 2 * pad - ks) / stride + 1 <= 0 || (W + 2 * pad - ks) / stride + 1 <= 0) return UAPS_EINVAL;
    return UAPS_OK;
}

}  // namespace

extern "C" int uaps_convs_pack_floats(int Cout, int Cin, int ks, size_t* fwd_floats, size_t* bwd_floats) {
    if (Cout <= 0 || Cin <= 0 || ks < 1 || ks > 7 || !(ks & 1)) return UAPS_EINVAL;
    const size_t taps = (size_t)ks * ks;
    if (fwd_floats) *fwd_floats = taps * round_up(Cin, 4) * round_up(Cout, 16);
    if (bwd_floats) *bwd_floats = taps * round_up(Cout, 4) * round_up(Cin, 16);
    return UAPS_OK;
}

extern "C" int uaps_convs_pack_weights(const float* w, int Cout, int Cin, int ks, float* wf, float* wb, uaps_stream_t stream) {
    if (!w || !wf || !wb || Cout <= 0 || Cin <= 0 || ks < 1 || ks > 7 || !(ks & 1)) return UAPS_EINVAL;
    const int taps = ks * ks;
    const long n = (long)taps * ((long)round_up(Cin, 4) * round_up(Cout, 16) + (long)round_up(Cout, 4) * round_up(Cin, 16));
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(convs_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wf, wb, Cout, Cin, taps, round_up(Cin, 4),
                       round_up(Cout, 16), round_up(Cout, 4), round_up(Cin, 16));
    return (int)hipGetLastError();
}

extern "C" int uaps_convs_out_size(int H, int W, int ks, int stride, int pad, int* OH, int* OW) {
    if (!OH || !OW || H <= 0 || W <= 0 || ks < 1 || stride < 1) return UAPS_EINVAL;
    *OH = (H + 2 * pad - ks) / stride + 1;
    *OW = (W + 2 * pad - ks) / stride + 1;
    return UAPS_OK;
}

// y [B,Cout,OH,OW] = conv2d(x [B,Cin,H,W], w, stride, padding) with wf from uaps_convs_pack_weights
extern "C" int uaps_convs_fwd(const float* x, const float* wf, float* y, int B, int Cin, int Cout, int H, int W, int ks, int stride,
                              int pad, uaps_stream_t stream) {
    if (!x || !wf || !y) return UAPS_EINVAL;
    int rc = check_sconv(B, Cin, Cout, H, W, ks, stride, pad);
    if (rc) return rc;
    SConvArgs a{x, wf, y, B, Cin, Cout, H, W, (H + 2 * pad - ks) / stride + 1, (W + 2 * pad - ks) / stride + 1, ks, stride, pad,
                round_up(Cin, 4), round_up(Cout, 16)};
    const int tiles_x = (a.OW + 15) / 16;
    const int nw = a.CoutP >= 64 ? 4 : (a.CoutP >= 32 ? 2 : 1);
    const int ncb = (a.CoutP + 16 * nw - 1) / (16 * nw);
    const long nitems = (long)B * a.OH * tiles_x * ncb;
    const long grid = (nitems + 3) / 4;
    if (grid > 0x7fffffffL) return UAPS_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (nw == 4) UAPS_LAUNCH_MAIN((convs_fwd_kernel<4>), dim3((unsigned)grid), dim3(256), 0, s, a, tiles_x, ncb, nitems);
    else if (nw == 2) UAPS_LAUNCH_MAIN((convs_fwd_kernel<2>), dim3((unsigned)grid), dim3(256), 0, s, a, tiles_x, ncb, nitems);
    else UAPS_LAUNCH_MAIN((convs_fwd_kernel<1>), dim3((unsigned)grid), dim3(256), 0, s, a, tiles_x, ncb, nitems);
    return (int)hipGetLastError();
}

// dx [B,Cin,H,W] = gradient of that convolution w.r.t. its input, given dy [B,Cout,OH,OW] and wb from uaps_convs_pack_weights
extern "C" int uaps_convs_bwd_data(const float* dy, const float* wb, float* dx, int B, int Cin, int Cout, int H, int W, int ks, int stride,
                                   int pad, uaps_stream_t stream) {
    if (!dy || !wb || !dx) return UAPS_EINVAL;
    int rc = check_sconv(B, Cin, Cout, H, W, ks, stride, pad);
    if (rc) return rc;
    SConvArgs a{dy, wb, dx, B, Cin, Cout, H, W, (H + 2 * pad - ks) / stride + 1, (W + 2 * pad - ks) / stride + 1, ks, stride, pad,
                round_up(Cin, 16), round_up(Cout, 4)};
    const int cols = (W + stride - 1) / stride;                  // columns of one parity class (at most)
    const int tiles_x = (cols + 15) / 16;
    const int nw = a.CinP >= 64 ? 4 : (a.CinP >= 32 ? 2 : 1);
    const int ncb = (a.CinP + 16 * nw - 1) / (16 * nw);
    const long nitems = (long)B * H * stride * tiles_x * ncb;
    const long grid = (nitems + 3) / 4;
    if (grid > 0x7fffffffL) return UAPS_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (nw == 4) UAPS_LAUNCH_MAIN((convs_bwd_data_kernel<4>), dim3((unsigned)grid), dim3(256), 0, s, a, tiles_x, ncb, nitems);
    else if (nw == 2) UAPS_LAUNCH_MAIN((convs_bwd_data_kernel<2>), dim3((unsigned)grid), dim3(256), 0, s, a, tiles_x, ncb, nitems);
    else UAPS_LAUNCH_MAIN((convs_bwd_data_kernel<1>), dim3((unsigned)grid), dim3(256), 0, s, a, tiles_x, ncb, nitems);
    return (int)hipGetLastError();
}

namespace {
struct SWrwPlan { int CoutS, CinS, ncob, ncib, nsplit, tiles_x, tiles_y; bool stem; };
SWrwPlan plan_swrw(int B, int Cin, int Cout, int OH, int OW, int ks, int stride = 1, int pad = 0) {
    SWrwPlan p{};
    // the stem form (stem7_wrw_kernel): 7x7 / 2 / 3, <= 3 input channels, 16-byte rows of dy
    p.stem = ks == 7 && stride == 2 && pad == 3 && Cin <= 3 && OW % 4 == 0;
    if (p.stem) {
        p.ncob = (Cout + 63) / 64; p.ncib = 1;
        p.CoutS = p.ncob * 64; p.CinS = kStemN;
        p.tiles_x = (OW + 31) / 32; p.tiles_y = (OH + 3) / 4;
        const long tiles = (long)B * p.tiles_x * p.tiles_y;
        long want = 1024 / p.ncob;                    // ~4 workgroups per CU
        if (want > tiles) want = tiles;
        p.nsplit = (int)(want < 1 ? 1 : want);
        return p;
    }
    p.ncob = (Cout + 15) / 16; p.ncib = (Cin + 15) / 16;
    p.CoutS = p.ncob * 16; p.CinS = p.ncib * 16;
    p.tiles_x = (OW + 31) / 32; p.tiles_y = (OH + 3) / 4;
    const long tiles = (long)B * p.tiles_x * p.tiles_y, blocks = (long)p.ncob * p.ncib * (ks == 7 ? 7 : 1);
    long want = blocks >= 1024 ? 1 : 1024 / blocks;
    if (want > tiles) want = tiles;
    if (want < 1) want = 1;
    p.nsplit = (int)want;
    return p;
}
}  // namespace

extern "C" int uaps_convs_wrw_workspace_bytes(int B, int Cin, int Cout, int H, int W, int ks, int stride, int pad, size_t* out) {
    if (!out) return UAPS_EINVAL;
    int rc = check_sconv(B, Cin, Cout, H, W, ks, stride, pad);
    if (rc) return rc;
    const SWrwPlan p = plan_swrw(B, Cin, Cout, (H + 2 * pad - ks) / stride + 1, (W + 2 * pad - ks) / stride + 1, ks, stride, pad);
    *out = (size_t)p.nsplit * (p.stem ? 1 : ks * ks) * p.CoutS * p.CinS * sizeof(float);
    return UAPS_OK;
}

// dw [Cout,Cin,ks,ks] = gradient w.r.t. the weights
extern "C" int uaps_convs_bwd_weight(const float* dy, const float* x, float* dw, int B, int Cin, int Cout, int H, int W, int ks, int stride,
                                     int pad, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    if (!dy || !x || !dw || !ws) return UAPS_EINVAL;
    int rc = check_sconv(B, Cin, Cout, H, W, ks, stride, pad);
    if (rc) return rc;
    if (ks != 1 && ks != 3 && ks != 7) return UAPS_ERANGE;
    const int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    const SWrwPlan p = plan_swrw(B, Cin, Cout, OH, OW, ks, stride, pad);
    if (ws_bytes < (size_t)p.nsplit * (p.stem ? 1 : ks * ks) * p.CoutS * p.CinS * sizeof(float)) return UAPS_EWORKSPACE;
    if (p.stem && (uintptr_t)dy % 16 == 0) {
        StemWrwArgs sa{dy, x, (float*)ws, B, Cin, Cout, H, W, OH, OW, p.CoutS, p.nsplit, p.tiles_x, p.tiles_y};
        UAPS_LAUNCH_MAIN(stem7_wrw_kernel, dim3((unsigned)p.nsplit, (unsigned)p.ncob), dim3(256), 0, (hipStream_t)stream, sa);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return (int)e0;
        const int nout = Cout * Cin * 49;
        hipLaunchKernelGGL(stem7_reduce_kernel, dim3((unsigned)((nout + 31) / 32)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, dw, p.nsplit,
                           Cout, p.CoutS, Cin * 49);
        return (int)hipGetLastError();
    }
    SWrwArgs a{dy, x, (float*)ws, B, Cin, Cout, H, W, OH, OW, ks, stride, pad, p.CoutS, p.CinS, p.ncob, p.ncib, p.nsplit, p.tiles_x, p.tiles_y};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(p.nsplit * p.ncob * p.ncib), ks == 7 ? 7 : 1);
    if (ks == 1) {
        if (stride == 1) UAPS_LAUNCH_MAIN((convs_wrw_kernel<1, 1, 1>), grid, dim3(256), 0, s, a);
        else UAPS_LAUNCH_MAIN((convs_wrw_kernel<1, 1, 2>), grid, dim3(256), 0, s, a);
    } else if (ks == 3) {
        if (stride == 1) UAPS_LAUNCH_MAIN((convs_wrw_kernel<3, 3, 1>), grid, dim3(256), 0, s, a);
        else UAPS_LAUNCH_MAIN((convs_wrw_kernel<3, 3, 2>), grid, dim3(256), 0, s, a);
    } else {
        if (stride == 1) UAPS_LAUNCH_MAIN((convs_wrw_kernel<1, 7, 1>), grid, dim3(256), 0, s, a);
        else UAPS_LAUNCH_MAIN((convs_wrw_kernel<1, 7, 2>), grid, dim3(256), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int taps = ks * ks;
    const long n = (long)taps * p.CoutS * p.CinS;
    if (n < 32768)
        hipLaunchKernelGGL(conv_wrw_reduce_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, (const float*)ws, (const float*)nullptr, dw,
                           (float*)nullptr, p.nsplit, taps, Cout, Cin, p.CoutS, p.CinS);
    else
        hipLaunchKernelGGL(conv_wrw_reduce_kernel<64>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, (const float*)ws, (const float*)nullptr, dw,
                           (float*)nullptr, p.nsplit, taps, Cout, Cin, p.CoutS, p.CinS);
    return (int)hipGetLastError();
}

// Every second row and column of `planes` planes (the sampling a 1x1 / stride 2 convolution performs, utilities/resnet.py:13-14,
// 157-161: layer2's shortcut projection): y [planes, OH, OW] = x[:, ::2, ::2], and its adjoint dx = zeros with dy at the even
// positions.  The 1x1 convolution itself then runs at stride 1 on the GEMM-tiled kernels (conv_gemm1x1.hpp).
namespace {
__global__ __launch_bounds__(256) void subsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int H, int W,
                                                             int OH, int OW, int vec) {
    if (vec) {              // OW % 4 == 0 (so W % 8 == 0 or W == 2 * OW - 1 is excluded by the caller): 8 floats in, 4 out
        const int q = OW / 4;
        const long n = planes * OH * q;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
            const int g = (int)(e % q), oy = (int)((e / q) % OH);
            const long pl = e / ((long)q * OH);
            const f32x4* src = reinterpret_cast<const f32x4*>(x + (pl * H + 2 * oy) * W + 8 * g);
            const f32x4 a = src[0], b = src[1];
            *reinterpret_cast<f32x4*>(y + (pl * OH + oy) * OW + 4 * g) = f32x4{a.x, a.z, b.x, b.z};
        }
        return;
    }
    const long n = planes * OH * OW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % OW), oy = (int)((e / OW) % OH);
        const long pl = e / ((long)OW * OH);
        y[e] = x[(pl * H + 2 * oy) * W + 2 * ox];
    }
}
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long planes, int H, int W,
                                                             int OH, int OW, int vec) {
    if (vec) {              // W % 8 == 0: 4 floats in, 8 out; odd rows are zero
        const int q = W / 8;
        const long n = planes * H * q;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
            const int g = (int)(e % q), iy = (int)((e / q) % H);
            const long pl = e / ((long)q * H);
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
            if (!(iy & 1)) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(dy + (pl * OH + iy / 2) * OW + 4 * g);
                a = f32x4{v.x, 0.f, v.y, 0.f}; b = f32x4{v.z, 0.f, v.w, 0.f};
            }
            f32x4* dst = reinterpret_cast<f32x4*>(dx + (pl * H + iy) * W + 8 * g);
            dst[0] = a; dst[1] = b;
        }
        return;
    }
    const long n = planes * H * W;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int ix = (int)(e % W), iy = (int)((e / W) % H);
        const long pl = e / ((long)W * H);
        dx[e] = ((ix | iy) & 1) ? 0.f : dy[(pl * OH + iy / 2) * OW + ix / 2];
    }
}
}  // namespace

// x [B, C, H, W] <-> xs [B, 4, C, H/2, W/2] with xs[b][2 py + px][c][i][j] = x[b][c][2 i + py][2 j + px] (H, W even): the four
// sampling phases of a stride-2 convolution as extra channels.  A 3x3 / 2 / padding-1 convolution of x is a 3x3 / 1 /
// padding-1 convolution of xs whose weights hold the 9 taps at (phase, tap) = (1, 0), (0, 1), (1, 1) per axis for
// ky / kx = 0, 1, 2 and zeros elsewhere (uaps_amd/conv.py: conv3x3s2), which runs on the split kernels.
namespace {
template <bool INVERSE>
__global__ __launch_bounds__(256) void space_depth2_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W) {
    const int q = W / 8, OH = H / 2, OW = W / 2;
    const long n = (long)B * C * H * q;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int g = (int)(e % q), iy = (int)((e / q) % H);
        const long bc = e / ((long)q * H);
        const int c = (int)(bc % C);
        const long b = bc / C;
        const size_t full = ((size_t)bc * H + iy) * W + 8 * g;
        const size_t ph0 = ((((size_t)b * 4 + 2 * (iy & 1)) * C + c) * OH + iy / 2) * OW + 4 * g, ph1 = ph0 + (size_t)C * OH * OW;
        if constexpr (!INVERSE) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(src + full), v = *reinterpret_cast<const f32x4*>(src + full + 4);
            *reinterpret_cast<f32x4*>(dst + ph0) = f32x4{u.x, u.z, v.x, v.z};
            *reinterpret_cast<f32x4*>(dst + ph1) = f32x4{u.y, u.w, v.y, v.w};
        } else {
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(src + ph0), o0 = *reinterpret_cast<const f32x4*>(src + ph1);
            *reinterpret_cast<f32x4*>(dst + full) = f32x4{e0.x, o0.x, e0.y, o0.y};
            *reinterpret_cast<f32x4*>(dst + full + 4) = f32x4{e0.z, o0.z, e0.w, o0.w};
        }
    }
}
}  // namespace

extern "C" int uaps_space_to_depth2(const float* x, float* xs, int B, int C, int H, int W, int inverse, uaps_stream_t stream) {
    if (!x || !xs || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (H % 2 || W % 8 || ((uintptr_t)x | (uintptr_t)xs) % 16) return UAPS_ERANGE;
    const long n = (long)B * C * H * (W / 8);
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    if (inverse) hipLaunchKernelGGL(space_depth2_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, xs, B, C, H, W);
    else hipLaunchKernelGGL(space_depth2_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, xs, B, C, H, W);
    return (int)hipGetLastError();
}

extern "C" int uaps_subsample2_fwd(const float* x, float* y, long planes, int H, int W, uaps_stream_t stream) {
    if (!x || !y || planes <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int vec = W % 8 == 0 && ((uintptr_t)x | (uintptr_t)y) % 16 == 0;
    const long n = planes * OH * (vec ? OW / 4 : OW);
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipLaunchKernelGGL(subsample2_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W, OH, OW, vec);
    return (int)hipGetLastError();
}
extern "C" int uaps_subsample2_bwd(const float* dy, float* dx, long planes, int H, int W, uaps_stream_t stream) {
    if (!dy || !dx || planes <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int vec = W % 8 == 0 && ((uintptr_t)dy | (uintptr_t)dx) % 16 == 0;
    const long n = planes * H * (vec ? W / 8 : W);
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, dx, planes, H, W, OH, OW, vec);
    return (int)hipGetLastError();
}

// 3x3 / stride 2 / padding 1 max-pool over `planes` = B*C planes of H x W; idx: uint8 [planes, OH, OW] for the backward
extern "C" int uaps_maxpool3x3s2_fwd(const float* x, float* y, void* idx, long planes, int H, int W, uaps_stream_t stream) {
    if (!x || !y || !idx || planes <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long n = planes * OH * OW;
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, (uint8_t*)idx, planes, H, W, OH, OW);
    return (int)hipGetLastError();
}
extern "C" int uaps_maxpool3x3s2_bwd(const float* dy, const void* idx, float* dx, long planes, int H, int W, uaps_stream_t stream) {
    if (!dy || !idx || !dx || planes <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long n = planes * H * W;
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, (const uint8_t*)idx, dx, planes, H, W, OH, OW);
    return (int)hipGetLastError();
}
