// Exact-N kernels for the class dimension: the 3x3 convolution between 16 feature channels and the <= 4 class logits
// (Decoder.out_conv, utilities/UAPS_unet.py:138-139, 152; C = 2 and C = 4 are four of the five BASELINE configs) in all
// three directions (forward and weight gradient here; the input gradient, 4 -> 16 channels, stays on the fp32 matrix kernel,
// which measured faster than this scheme's mirror image).
//
// The matrix-core kernels pad the short side to 16: a 16 -> 4 convolution then does 4x the multiplies it needs and, at
// 256 x 256, runs at 2.5-3x its HBM time.  With <= 4 output (or contraction) channels the work per pixel is 576 multiply-adds,
// i.e. 18 us of packed fp32 VALU (v_pk_fma_f32: two pixels per lane and instruction) for 2.1 M pixels -- below the 27 us the
// tensors take to stream -- so these kernels use no matrix instruction at all: plain fp32 fma chains (exactly what
// F.conv2d's fp32 path computes, no operand splitting), inputs staged through LDS with their halo, weights copied once per
// workgroup from the packed exact layouts wf / wb (uaps_conv_pack_weights) into LDS and read as broadcasts.
//
// Packed-operand rule (tools/diag/pkfma_probe.hip): no v_pk_*_f32 may select the LOW half of its second
// source from the high register of a pair (`op_sel:[0,1,0]`: what hipcc emits for a scalar broadcast `f32x2{w, w}` whose w sits
// in an odd register).  On MI355X that form returned a wrong low half for lanes 48..63 whenever another wave of the SIMD was
// issuing v_mfma_f32_16x16x32 (another decoder's stream, another process): about one launch of out_conv in 100 came back
// with one wrong product in 16 of its outputs.  Broadcasts from the LOW register (`op_sel_hi:[1,0,1]`) are unaffected.  The
// forward kernel therefore keeps every weight TWICE in LDS and reads {w, w} as a natural pair; the weight gradient pins its
// broadcast values to low registers (bcast_lo, conv_kernels.hpp); tools/isa_lint.py checks the built code objects.
#pragma once
#include "conv_kernels.hpp"

namespace uaps {

// -------------------------------------------------------------------------------------------------
// Forward / input gradient.  KIN = contraction channels staged per chunk (8: 16 -> <= 4 forward; 4: <= 4 -> 16 input
// gradient), NOUT = output channels per thread (4 / 16).  Tile 16 x 64 pixels, thread = 4 consecutive pixels of one row.
// Weights: wp[(tap * KP + k) * NP + n] (KP = ConvFwdArgs::CinP, NP = CoutP), NOUT consecutive floats per (tap, k).
// -------------------------------------------------------------------------------------------------
template <int KIN, int NOUT, bool XF>
__device__ __forceinline__ void conv_small_body(const ConvFwdArgs& a) {
    constexpr int TH = 16, TW = 64, IH = TH + 2, IWP = TW + 8;      // LDS rows start 4 floats left of the tile (16-byte aligned)
    constexpr int UPR = IWP / 4, NUN = KIN * IH * UPR, NT = (NUN + kConvThreads - 1) / kConvThreads;
    constexpr int KMAX = NOUT <= 4 ? 32 : 4;          // contraction channels the launcher admits (plan_fwd)
    __shared__ __attribute__((aligned(16))) float sIn[KIN * IH * IWP];
    __shared__ __attribute__((aligned(16))) float sW[KMAX * 9 * NOUT * 2];  // [k][tap][n][2]: every weight TWICE (a natural pair {w, w}), 16-byte broadcast reads

    const int tid = threadIdx.x, row = tid >> 4, xg = tid & 15;
    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (bid >= a.B * a.tiles_x * a.tiles_y) return;
    // (scalar loads of the weights inside the fma loop were tried first: 576 SGPRs do not exist, the compiler serialised
    // ~110 s_load / s_waitcnt pairs per chunk and the kernel ran at 5x its VALU time)
    for (int e = tid; e < a.CinP * 9 * NOUT && e < KMAX * 9 * NOUT; e += kConvThreads) {
        const int n = e % NOUT, t = (e / NOUT) % 9, k = e / (9 * NOUT);
        const float w = a.wp[(size_t)(t * a.CinP + k) * a.CoutP + n];
        *reinterpret_cast<f32x2*>(&sW[2 * e]) = f32x2{w, w};
    }
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int HW = a.H * a.W;
    const uint32_t HW4 = (uint32_t)HW * 4u;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in + (size_t)b * a.Cin * HW, (uint32_t)a.Cin * HW4);
    const __amdgpu_buffer_rsrc_t rs_xf = XF ? make_rsrc(a.xf + (size_t)(b / (XF ? a.xf_Bg : 1)) * a.Cin, (uint32_t)a.Cin * 8u) : rs_in;

    // staging units: float4 pieces of the haloed tile, (channel, row, column unit)
    uint32_t goff[NT];
    int loff[NT], uch[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int u = tid + n * kConvThreads;
        const int c = u / (IH * UPR), r = (u % (IH * UPR)) / UPR, cu = u % UPR;
        const int gy = y0 - 1 + r, gx = x0 - 4 + cu * 4;
        const bool ok = u < NUN && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;      // W % 4 == 0: all in or all out
        goff[n] = ok ? (uint32_t)(c * HW + gy * a.W + gx) * 4u : kOob;
        loff[n] = u < NUN ? (c * IH + r) * IWP + cu * 4 : -1;
        uch[n] = c;
    }
    float rin[NT][4];
    f32x2 rxf[XF ? NT : 1];
    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bool ok = goff[n] != kOob && ci0 + uch[n] < a.Cin;
            buf_load<4>(rs_in, ok ? goff[n] + (uint32_t)ci0 * HW4 : kOob, rin[n]);
            if constexpr (XF)
                rxf[n] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_xf, ok ? (int)((uint32_t)(ci0 + uch[n]) * 8u) : (int)kOob, 0, 0));
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (loff[n] < 0) continue;
            f32x4 v = f32x4{rin[n][0], rin[n][1], rin[n][2], rin[n][3]};
            if constexpr (XF) {                       // leaky_relu(fma(y, scale, shift)); padding reads (0, 0) and stays zero
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z = __builtin_fmaf(v[k], rxf[n].x, rxf[n].y);
                    v[k] = __builtin_fmaxf(z, z * a.xf_slope);
                }
            }
            *reinterpret_cast<f32x4*>(&sIn[loff[n]]) = v;
        }
    };

    f32x2 acc[NOUT][2];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) { acc[n][0] = f32x2{0.f, 0.f}; acc[n][1] = f32x2{0.f, 0.f}; }

    const int nchunks = (a.Cin + KIN - 1) / KIN;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk((ch + 1) * KIN);
#pragma unroll 1
        for (int c = 0; c < KIN; ++c) {               // rolled: one channel's 9 x NOUT weights are live at a time (registers)
            const int k = ch * KIN + c;               // contraction channel (uniform); channels past Cin were staged as zeros
#pragma unroll(NOUT <= 4 ? 3 : 1)
            for (int ky = 0; ky < 3; ++ky) {
                const float* base = &sIn[(c * IH + row + ky) * IWP + 4 * xg + 3];
                const f32x4 mid = *reinterpret_cast<const f32x4*>(base + 1);
                const float seg[6] = {base[0], mid.x, mid.y, mid.z, mid.w, base[5]};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float* w = &sW[((k < a.CinP ? k : 0) * 9 + ky * 3 + kx) * NOUT * 2];  // same address in every lane: broadcast reads
                    const f32x2 s0 = f32x2{seg[kx], seg[kx + 1]}, s1 = f32x2{seg[kx + 2], seg[kx + 3]};
#pragma unroll
                    for (int n2 = 0; n2 < NOUT; n2 += 2) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + 2 * n2);             // {w[n2], w[n2], w[n2 + 1], w[n2 + 1]}
                        const f32x2 wa = f32x2{w4.x, w4.y}, wb = f32x2{w4.z, w4.w};               // aligned halves of the 16-byte read
                        acc[n2][0] = __builtin_elementwise_fma(s0, wa, acc[n2][0]);
                        acc[n2][1] = __builtin_elementwise_fma(s1, wa, acc[n2][1]);
                        acc[n2 + 1][0] = __builtin_elementwise_fma(s0, wb, acc[n2 + 1][0]);
                        acc[n2 + 1][1] = __builtin_elementwise_fma(s1, wb, acc[n2 + 1][1]);
                    }
                }
            }
        }
        __syncthreads();
        if (more) store_chunk();
        __syncthreads();
    }

    const int gy = y0 + row, gx = x0 + 4 * xg;
    if (gy < a.H && gx < a.W) {
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            if (n >= a.Cout) break;
            const float bv = a.bias ? a.bias[n] : 0.f;
            *reinterpret_cast<f32x4*>(a.out + ((size_t)b * a.Cout + n) * HW + (size_t)gy * a.W + gx) =
                f32x4{acc[n][0].x + bv, acc[n][0].y + bv, acc[n][1].x + bv, acc[n][1].y + bv};
        }
    }
}

// three workgroups per CU (LDS: 41.5 KB tile + 9.2 KB weights), i.e. 168 registers
template <int KIN, int NOUT>
__global__ __launch_bounds__(kConvThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))
void conv_small_kernel(ConvFwdArgs a) { conv_small_body<KIN, NOUT, false>(a); }
template <int KIN, int NOUT>
__global__ __launch_bounds__(kConvThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))
void conv_small_bn_kernel(ConvFwdArgs a) { conv_small_body<KIN, NOUT, true>(a); }

// -------------------------------------------------------------------------------------------------
// Weight gradient for <= 4 output channels and 16 input channels:
//   dw[co][ci][ky][kx] = sum_{b,y,x} dy[b][co][y][x] * in[b][ci][y + ky - 1][x + kx - 1]
// Lane = (input channel ci = lane & 15, pixel share rg = 16 shares of a tile of 8 x 64 pixels: one row half each); a thread
// keeps the 4 x 9 sums of its channel (+ the bias sums) in registers over its workgroup's whole run of tiles -- each as a PAIR
// of partial sums over the even and the odd pixels, added at the end: both operands of every packed fma are then natural pairs
// of adjacent pixels and nothing is broadcast (the packed-operand rule at the top of this file).  The 16 shares are summed
// through shuffles and LDS (fixed order) into slab[split][tap][4][16] for conv_wrw_reduce_kernel.
// -------------------------------------------------------------------------------------------------
constexpr int kSmallWrwS = 724;      // floats per staged input channel: 10 rows x 72 + 4 (== 20 mod 64: the 16 channel lanes hit disjoint banks)

template <bool XF>
__device__ __forceinline__ void conv_small_wrw_body(const ConvWrwArgs& a) {
    constexpr int TH = 8, TW = 64, IH = TH + 2, IWP = TW + 8, UPR = IWP / 4;
    constexpr int NXU = 16 * IH * UPR, NXT = (NXU + kConvThreads - 1) / kConvThreads;      // 2880 float4 units of the input tile
    constexpr int NDU = 4 * TH * (TW / 4), NDT = NDU / kConvThreads;                       // 512 float4 units of dy
    __shared__ __attribute__((aligned(16))) float sX[16 * kSmallWrwS];
    __shared__ __attribute__((aligned(16))) float sD[4 * TH * TW];
    __shared__ f32x2 sXf[XF ? kWrwMaxGroups * 16 + 1 : 1];      // (scale, shift) per (statistics group, channel); last slot (0, 0) for padding
    constexpr int XF_ZERO = kWrwMaxGroups * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = lane & 15, rg = (lane >> 4) + 4 * wave, row = rg >> 1, half = rg & 1;
    const int HW = a.H * a.W;
    const uint32_t HW4 = (uint32_t)HW * 4u;

    const int tpi = a.tiles_x * a.tiles_y, ntiles = a.B * tpi;
    const int split = xcd_swizzle(blockIdx.x, gridDim.x);
    if (split >= a.nsplit) return;
    const int t_begin = (int)((long)ntiles * split / a.nsplit), t_end = (int)((long)ntiles * (split + 1) / a.nsplit);

    if constexpr (XF) {
        const int G = a.B / a.xf_Bg;
        for (int i = tid; i < G * 16; i += kConvThreads) {
            const int g = i / 16, c = i % 16;
            f32x2 v = f32x2{0.f, 0.f};
            if (c < a.Cin) { const float2 t = a.xf[(size_t)g * a.Cin + c]; v = f32x2{t.x, t.y}; }
            sXf[i] = v;
        }
        if (tid == 0) sXf[XF_ZERO] = f32x2{0.f, 0.f};
        __syncthreads();
    }
    float rx[NXT][4], rd[NDT][4];
    int xfi[XF ? NXT : 1];
    auto load_tile = [&](int t) {
        const int b = t / tpi, tt = t - b * tpi, tx = tt / a.tiles_y, ty = tt - tx * a.tiles_y;      // down 64-pixel column strips
        const int y0 = ty * TH, x0 = tx * TW;
        const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.in + (size_t)b * a.Cin * HW, (uint32_t)a.Cin * HW4);
        const __amdgpu_buffer_rsrc_t rs_d = make_rsrc(a.dout + (size_t)b * a.Cout * HW, (uint32_t)a.Cout * HW4);
#pragma unroll
        for (int n = 0; n < NXT; ++n) {
            const int u = tid + n * kConvThreads;
            const int c = u / (IH * UPR), r = (u % (IH * UPR)) / UPR, cu = u % UPR;
            const int gy = y0 - 1 + r, gx = x0 - 4 + cu * 4;
            const bool ok = u < NXU && c < a.Cin && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            buf_load<4>(rs_x, ok ? (uint32_t)(c * HW + gy * a.W + gx) * 4u : kOob, rx[n]);
            if constexpr (XF) xfi[n] = ok ? (b / a.xf_Bg) * 16 + c : XF_ZERO;      // padding stays zero
        }
#pragma unroll
        for (int n = 0; n < NDT; ++n) {
            const int u = tid + n * kConvThreads;
            const int c = u / (TH * (TW / 4)), r = (u / (TW / 4)) % TH, cu = u % (TW / 4);
            const int gy = y0 + r, gx = x0 + cu * 4;
            const bool ok = c < a.Cout && gy < a.H && gx < a.W;
            buf_load<4>(rs_d, ok ? (uint32_t)(c * HW + gy * a.W + gx) * 4u : kOob, rd[n]);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int n = 0; n < NXT; ++n) {
            const int u = tid + n * kConvThreads;
            if (u >= NXU) continue;
            const int c = u / (IH * UPR), r = (u % (IH * UPR)) / UPR, cu = u % UPR;
            f32x4 v = f32x4{rx[n][0], rx[n][1], rx[n][2], rx[n][3]};
            if constexpr (XF) {
                const f32x2 cf = sXf[xfi[n]];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z = __builtin_fmaf(v[k], cf.x, cf.y);
                    v[k] = __builtin_fmaxf(z, z * a.xf_slope);
                }
            }
            *reinterpret_cast<f32x4*>(&sX[c * kSmallWrwS + r * IWP + cu * 4]) = v;
        }
#pragma unroll
        for (int n = 0; n < NDT; ++n) {
            const int u = tid + n * kConvThreads;
            *reinterpret_cast<f32x4*>(&sD[u * 4]) = f32x4{rd[n][0], rd[n][1], rd[n][2], rd[n][3]};
        }
    };

    f32x2 acc[4][9], accb[4];            // [output channel][tap] x (even pixels, odd pixels)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        accb[p] = f32x2{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[p][t] = f32x2{0.f, 0.f};
    }

    if (t_begin < t_end) { load_tile(t_begin); store_tile(); }
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const bool more = t + 1 < t_end;
        if (more) load_tile(t + 1);
#pragma unroll 2
        for (int g = 0; g < 8; ++g) {
            const int col = half * 32 + 4 * g;
            f32x4 d[4];
#pragma unroll
            for (int co = 0; co < 4; ++co) d[co] = *reinterpret_cast<const f32x4*>(&sD[(co * TH + row) * TW + col]);
            f32x2 dp[4][2];                   // dy pixel pairs (0, 1) and (2, 3) of every output channel: halves of the 16-byte reads
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                dp[co][0] = f32x2{d[co].x, d[co].y}; dp[co][1] = f32x2{d[co].z, d[co].w};
                accb[co] += dp[co][0]; accb[co] += dp[co][1];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* base = &sX[ci * kSmallWrwS + (row + ky) * IWP + col + 3];
                const f32x4 mid = *reinterpret_cast<const f32x4*>(base + 1);
                const float seg[6] = {base[0], mid.x, mid.y, mid.z, mid.w, base[5]};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    // input pixels (kx, kx + 1) meet dy pixels (0, 1), input pixels (kx + 2, kx + 3) meet (2, 3)
                    const f32x2 x01 = f32x2{seg[kx], seg[kx + 1]}, x23 = f32x2{seg[kx + 2], seg[kx + 3]};
#pragma unroll
                    for (int co = 0; co < 4; ++co) {
                        acc[co][ky * 3 + kx] = __builtin_elementwise_fma(dp[co][0], x01, acc[co][ky * 3 + kx]);
                        acc[co][ky * 3 + kx] = __builtin_elementwise_fma(dp[co][1], x23, acc[co][ky * 3 + kx]);
                    }
                }
            }
        }
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }

    // ---- sum the 16 pixel shares of every (co, tap, ci): the 4 shares of a wave by shuffles, the 4 waves through LDS ----
    float v[40];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int t = 0; t < 9; ++t) v[p * 9 + t] = acc[p][t].x + acc[p][t].y;
        v[36 + p] = accb[p].x + accb[p].y;
    }
#pragma unroll
    for (int i = 0; i < 40; ++i) {
        v[i] += __shfl_xor(v[i], 16, 64);
        v[i] += __shfl_xor(v[i], 32, 64);
    }
    float* red = sX;                         // [wave][40][16]
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 40; ++i) red[(wave * 40 + i) * 16 + ci] = v[i];
    }
    __syncthreads();
    float* slab = a.slab + (size_t)split * 9 * a.CoutS * a.CinS;      // [tap][CoutS = 4][CinS = 16]
    for (int e = tid; e < 40 * 16; e += kConvThreads) {
        const int i = e / 16, c = e % 16;
        const float s = (red[(0 * 40 + i) * 16 + c] + red[(1 * 40 + i) * 16 + c]) + (red[(2 * 40 + i) * 16 + c] + red[(3 * 40 + i) * 16 + c]);
        if (i < 36) slab[((size_t)(i % 9) * a.CoutS + i / 9) * a.CinS + c] = s;
        else if (a.bslab && c == 0) a.bslab[(size_t)split * a.CoutS + (i - 36)] = s;
    }
}

static __global__ __launch_bounds__(kConvThreads, 2) void conv_small_wrw_kernel(ConvWrwArgs a) { conv_small_wrw_body<false>(a); }
static __global__ __launch_bounds__(kConvThreads, 2) void conv_small_wrw_bn_kernel(ConvWrwArgs a) { conv_small_wrw_body<true>(a); }

}  // namespace uaps
