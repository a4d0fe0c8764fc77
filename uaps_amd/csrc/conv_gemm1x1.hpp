// 1x1 convolutions with many channels as what they are: GEMMs.  The bottleneck projections of the ResNet encoders
// (utilities/resnet.py:55-95: 64 ... 2048 channels, 43 % of the configs[4] step) have no tap reuse, so the 3x3 kernels' tiling
// (8 x 32 pixels x 32 output channels per workgroup) gives each staged input element only 32 multiply-adds.  Here a workgroup owns
// 128 consecutive pixels of one image x 128 (or 64) output channels and a wave a 64 x 64 (64 x 32) block of v_mfma_f32_32x32x16
// tiles -- the register blocking of a GEMM -- in the split arithmetic of conv_split.hpp (two fp16 pieces of the scaled operands, three
// products; or three bf16 pieces, six products, for operands without a magnitude bound).
//
//   forward / input gradient   y[b][co][p] = sum_ci w[co][ci] x[b][ci][p]       M = pixels, N = Cout, K = Cin, 32 channels per chunk
//   weight gradient            dw[co][ci]  = sum_{b,p} dy[b][co][p] x[b][ci][p] M = Cout,   N = Cin,  K = pixels, 32 pixels per chunk
//
// NCHW keeps the pixels of a channel contiguous: the forward's A operand (8 consecutive CHANNELS of one pixel per lane) is
// transposed while staging (a thread fetches 2 pixels x 8 channels, splits, and writes two 16-byte units), the weight gradient's
// operands (8 consecutive PIXELS of one channel per lane) are fragments as they lie in memory.  LDS images are [piece][k-group][row]
// of 16-byte units with `row` the MFMA row / column index: a fragment read is 512 contiguous bytes per k-group.
#pragma once
#include "conv_split.hpp"
#include "conv_split_wrw.hpp"

namespace uaps {

// Diagnostic builds only (make -C uaps_amd/csrc g1abl; tools/diag/g1_ablate.sh): UAPS_G1_ABLATE = 5 drops the staging behind the
// first chunk, 6 also the barriers, 7 also the LDS fragment reads; 1 .. 4 drop ONE part of the staging behind the first chunk
// (1 the activation loads, 2 the weight loads, 3 the split arithmetic, 4 the LDS stores) -- timing only, the results are meaningless.
#ifndef UAPS_G1_ABLATE
#define UAPS_G1_ABLATE 0
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// forward / input gradient.  BN = 128 or 64 output channels per workgroup; packed weights as conv_s32_body reads them
// ([piece][tap = 0][channel group][CoutP][8]); ConvFwdArgs::CinP = padded channel groups, tiles_x = pixel tiles per image.
// ---------------------------------------------------------------------------------------------------------------------------
template <int BN, bool H16>
__device__ __forceinline__ void conv_g1_body(const ConvFwdArgs& a) {
    constexpr int NP = H16 ? 2 : 3;
    constexpr int TM = 128, KG = 4;                   // pixels per tile; k-groups (8 channels) per chunk
    constexpr int NTW = BN / 64;                      // 32-channel N tiles per wave (waves: 2 pixel halves x 2 channel halves)
    constexpr int NWU = NP * KG * BN, NWT = NWU / kConvThreads;
    static_assert(NWU % kConvThreads == 0, "weight units per thread");

    __shared__ __attribute__((aligned(16))) u32x4 sA[NP * KG * TM];
    __shared__ __attribute__((aligned(16))) u32x4 sB[NP * KG * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (bid >= a.B * a.tiles_x * a.nblk) return;
    const int nb = bid % a.nblk; bid /= a.nblk;
    const int tx = bid % a.tiles_x;
    const int b = bid / a.tiles_x;
    const int HW = a.H * a.W, p0 = tx * TM, co0 = nb * BN;
    const uint32_t HW4 = (uint32_t)HW * 4u;

    float in_scale = 1.f, out_scale_a = 1.f, out_scale_w = 1.f;
    if constexpr (H16) {
        const f32x2 sc = h16_scale(bound_of(a.in_bound, a.in_mul));
        in_scale = sc.x; out_scale_a = sc.y; out_scale_w = a.wscale[1];
    }

    // ---- staging plan: thread = (k-group ug, pixel pair upp): 2 consecutive pixels x 8 channels ----
    const int ug = tid >> 6, upp = tid & 63;
    const bool uin = p0 + 2 * upp < HW;               // HW is even (W % 4 == 0): both pixels inside or both outside
    const uint32_t ugoff = (uint32_t)(ug * 8 * HW + p0 + 2 * upp) * 4u;
    const int uloff = ug * TM + 2 * upp;
    const int CGP = a.CinP;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in + (size_t)b * a.Cin * HW, (uint32_t)a.Cin * HW4);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.wp, (uint32_t)NP * CGP * a.CoutP * 16u);
    // weight unit e = tid + n * 256 -> (piece, k-group g, column); with BN = 256 the column is the thread and (piece, g) are the
    // same for the whole workgroup: one vector offset + scalar offsets
    constexpr bool WUNI = BN == kConvThreads;
    uint32_t wgoff[WUNI ? 1 : NWT];
    if constexpr (WUNI) wgoff[0] = (uint32_t)(co0 + tid) * 16u;
    else {
#pragma unroll
        for (int n = 0; n < NWT; ++n) {
            const int e = tid + n * kConvThreads;
            const int col = e % BN, g = (e / BN) % KG, piece = e / (BN * KG);
            wgoff[n] = (uint32_t)((piece * CGP + g) * a.CoutP + co0 + col) * 16u;
        }
    }

    f32x2 rin[8];
    u32x4 rw[NWT];
    u32x4 pk[NP][2];
    auto load_chunk = [&](int ci0) {
        if (UAPS_G1_ABLATE != 1 || ci0 == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c)                   // channels past Cin lie beyond the buffer's range: zeros
            rin[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, uin ? (int)(ugoff + (uint32_t)(ci0 + c) * HW4) : (int)kOob, 0, 0));
        }
        const uint32_t wbase = (uint32_t)(ci0 / 8) * a.CoutP * 16u;
        if (UAPS_G1_ABLATE != 2 || ci0 == 0) {
#pragma unroll
        for (int n = 0; n < NWT; ++n) {
            if constexpr (WUNI)
                rw[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)wgoff[0], (int)(wbase + (uint32_t)(((n / KG) * CGP + n % KG) * a.CoutP) * 16u), 0));
            else
                rw[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wgoff[n] + wbase), 0, 0));
        }
        }
    };
    bool first_split = true;
    auto split_chunk = [&]() {
        if (UAPS_G1_ABLATE == 3 && !first_split) {
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(rin[c]));      // the loads are still awaited
            return;
        }
        first_split = false;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                const float v0 = rin[2 * c2][p], v1 = rin[2 * c2 + 1][p];
                if constexpr (H16) {
                    unsigned q0, q1;
                    conv_split2h(v0 * in_scale, v1 * in_scale, q0, q1);
                    pk[0][p][c2] = q0; pk[1][p][c2] = q1;
                } else {
                    unsigned q0, q1, q2;
                    conv_split3(v0, v1, q0, q1, q2);
                    pk[0][p][c2] = q0; pk[1][p][c2] = q1; pk[2][p][c2] = q2;
                }
            }
    };
    bool first_store = true;
    auto store_chunk = [&]() {
        if (UAPS_G1_ABLATE == 4 && !first_store) {
#pragma unroll
            for (int q = 0; q < NP; ++q) { asm volatile("" : "+v"(pk[q][0])); asm volatile("" : "+v"(pk[q][1])); }
#pragma unroll
            for (int n = 0; n < NWT; ++n) asm volatile("" : "+v"(rw[n]));
            return;
        }
        first_store = false;
#pragma unroll
        for (int q = 0; q < NP; ++q) { sA[q * KG * TM + uloff] = pk[q][0]; sA[q * KG * TM + uloff + 1] = pk[q][1]; }
#pragma unroll
        for (int n = 0; n < NWT; ++n) sB[tid + n * kConvThreads] = rw[n];
    };

    f32x16 acc[2][NTW];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const int aoff = wm * 64 + r, boff = wn * (BN / 2) + r;
    const int nchunks = (a.Cin + 31) / 32;
    load_chunk(0);
    split_chunk();
    store_chunk();
    __syncthreads();
#if UAPS_G1_ABLATE >= 7
    bf16x8 af7[2][NP], bf7[NTW][NP];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < NP; ++p) af7[m][p] = __builtin_bit_cast(bf16x8, sA[(p * KG + h) * TM + aoff + m * 32]);
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int p = 0; p < NP; ++p) bf7[n][p] = __builtin_bit_cast(bf16x8, sB[(p * KG + h) * BN + boff + n * 32]);
#endif
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = UAPS_G1_ABLATE >= 5 ? false : ch + 1 < nchunks;
        if (more) load_chunk((ch + 1) * 32);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2][NP], bfr[NTW][NP];
#if UAPS_G1_ABLATE >= 7
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < NP; ++p) { af[m][p] = af7[m][p]; asm volatile("" : "+v"(af[m][p])); }
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int p = 0; p < NP; ++p) { bfr[n][p] = bf7[n][p]; asm volatile("" : "+v"(bfr[n][p])); }
#else
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < NP; ++p) af[m][p] = __builtin_bit_cast(bf16x8, sA[(p * KG + 2 * ks + h) * TM + aoff + m * 32]);
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int p = 0; p < NP; ++p) bfr[n][p] = __builtin_bit_cast(bf16x8, sB[(p * KG + 2 * ks + h) * BN + boff + n * 32]);
#endif
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m) {         // smallest partial products first
                    f32x16 c = acc[m][n];
                    if constexpr (H16) {
                        const auto H = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(af[m][1]), H(bfr[n][0]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(af[m][0]), H(bfr[n][1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(af[m][0]), H(bfr[n][0]), c, 0, 0, 0);
                    } else {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][2], bfr[n][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bfr[n][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][1], bfr[n][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][1], bfr[n][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bfr[n][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bfr[n][0], c, 0, 0, 0);
                    }
                    acc[m][n] = c;
                }
        }
        if (more) split_chunk();                     // behind the matrix phase: the fetched chunk has had its time to arrive
#if UAPS_G1_ABLATE < 6
        __syncthreads();
#endif
        if (more) store_chunk();
#if UAPS_G1_ABLATE < 6
        __syncthreads();
#endif
    }

    // ---- epilogue.  C/D of 32x32: lane (n = r, h) register i holds pixel 8 (i >> 2) + 4 h + (i & 3) of channel n ----
    float st_s[NTW], st_q[NTW];
    float chk = 0.f;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        st_s[n] = 0.f; st_q[n] = 0.f;
        const int co = co0 + wn * (BN / 2) + n * 32 + r;
        const bool co_ok = co < a.Cout;
        const float bv = (a.bias && co_ok) ? a.bias[co] : 0.f;
        const float sh = stats_shift(a, co, co_ok);
        float* out_c = a.out + ((size_t)b * a.Cout + (co_ok ? co : 0)) * HW;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int p = p0 + wm * 64 + m * 32 + 8 * g + 4 * h;
                f32x4 v = f32x4{acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                if constexpr (H16) { v *= out_scale_a; v *= out_scale_w; }      // exact: powers of two
                v += bv;
                if (co_ok && p < HW) {               // HW % 4 == 0: the 4 pixels are all inside or all outside
                    if constexpr (H16) note_nonfinite(chk, v);
                    *reinterpret_cast<f32x4*>(out_c + p) = v;
                    const f32x4 d = v - sh;
                    st_s[n] += (d.x + d.y) + (d.z + d.w);
                    st_q[n] += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
            }
    }
    if constexpr (H16) report_nonfinite(a.err, chk, UAPS_ERR_CONV_NONFINITE);
    if (a.stats != nullptr) {                    // per-tile BatchNorm partial sums: 4 partials (2 pixel halves x 2 lane halves) per channel, fixed order
        float* red = reinterpret_cast<float*>(sA);
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int cl = wn * (BN / 2) + n * 32 + r;
            red[((wm * 2 + h) * BN + cl) * 2 + 0] = st_s[n];
            red[((wm * 2 + h) * BN + cl) * 2 + 1] = st_q[n];
        }
        __syncthreads();
        if (tid < BN && co0 + tid < a.Cout) {
            float s0 = 0.f, q0 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s0 += red[(k * BN + tid) * 2]; q0 += red[(k * BN + tid) * 2 + 1]; }
            a.stats[((size_t)(co0 + tid) * a.B + b) * a.tiles_x + tx] = make_float2(s0, q0);
        }
    }
}

template <int BN>
__global__ __launch_bounds__(kConvThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))      // 168 registers (the compiler's free choice was 170)
void conv_g1h_kernel(ConvFwdArgs a) { conv_g1_body<BN, true>(a); }
// 256 output channels per workgroup (a wave: 64 pixels x 128 channels, 128 accumulator registers, two workgroups per CU): every
// staged activation element -- the expensive operand: fp32, fetched as 8-byte pieces and split here -- feeds twice the matrix work
static __global__ __launch_bounds__(kConvThreads, 2) void conv_g1h256_kernel(ConvFwdArgs a) { conv_g1_body<256, true>(a); }
template <int BN>
__global__ __launch_bounds__(kConvThreads, 2) void conv_g1s_kernel(ConvFwdArgs a) { conv_g1_body<BN, false>(a); }

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient.  Workgroup = 128 output x 128 input channels (wave: 64 x 64) x a contiguous run of 32-pixel chunks of the
// batch (HW % 32 == 0); partial sums to slab[split][CoutS][CinS] for conv_wrw_reduce_kernel (taps = 1), bias partials to bslab.
// ConvWrwArgs: tiles_x = chunks per image, ncob / ncib = 128-channel blocks.
// ---------------------------------------------------------------------------------------------------------------------------
template <bool H16>
__device__ __forceinline__ void conv_gw1_body(const ConvWrwArgs& a) {
    constexpr int NP = H16 ? 2 : 3;
    constexpr int BC = 128, KG = 4;                   // channels per block side; k-groups (8 pixels) per chunk
    constexpr int NU = BC * KG / kConvThreads;        // staging units (channel, k-group) per thread and operand: 2

    __shared__ __attribute__((aligned(16))) u32x4 sD[NP * KG * BC];
    __shared__ __attribute__((aligned(16))) u32x4 sX[NP * KG * BC];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wco = wave & 1, wci = wave >> 1;

    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (bid >= a.nsplit * a.ncob * a.ncib) return;
    const int cib = bid % a.ncib; bid /= a.ncib;
    const int cob = bid % a.ncob;
    const int split = bid / a.ncob;
    const int co0 = cob * BC, ci0 = cib * BC;
    const int HW = a.H * a.W;
    const int cpi = a.tiles_x, nchunks = a.B * cpi;
    const int c_begin = (int)((long)nchunks * split / a.nsplit), c_end = (int)((long)nchunks * (split + 1) / a.nsplit);
    const bool want_bias = a.bslab != nullptr && cib == 0;

    float sc_d = 1.f, sc_x = 1.f, inv_d = 1.f, inv_x = 1.f;
    if constexpr (H16) {
        const f32x2 sd = h16_scale(bound_of(a.dy_bound, a.dy_mul)), sx = h16_scale(bound_of(a.in_bound, a.in_mul));
        sc_d = sd.x; inv_d = sd.y; sc_x = sx.x; inv_x = sx.y;
    }

    // unit u = tid + n * 256 -> (channel u >> 2, k-group u & 3): the 4 k-groups of a channel are 128 contiguous bytes
    float rd[NU][8], rx[NU][8];
    auto load_chunk = [&](int c) {
        const int b = c / cpi, p0 = (c - b * cpi) * 32;
#pragma unroll
        for (int n = 0; n < NU; ++n) {
            const int u = tid + n * kConvThreads, chn = u >> 2, g = u & 3;
            const bool okd = co0 + chn < a.Cout, okx = ci0 + chn < a.Cin;
            const float* pd = a.dout + ((size_t)b * a.Cout + (okd ? co0 + chn : 0)) * HW + p0 + 8 * g;
            const float* px = a.in + ((size_t)b * a.Cin + (okx ? ci0 + chn : 0)) * HW + p0 + 8 * g;
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 d0 = okd ? *reinterpret_cast<const f32x4*>(pd) : z, d1 = okd ? *reinterpret_cast<const f32x4*>(pd + 4) : z;
            const f32x4 x0 = okx ? *reinterpret_cast<const f32x4*>(px) : z, x1 = okx ? *reinterpret_cast<const f32x4*>(px + 4) : z;
#pragma unroll
            for (int k = 0; k < 4; ++k) { rd[n][k] = d0[k]; rd[n][4 + k] = d1[k]; rx[n][k] = x0[k]; rx[n][4 + k] = x1[k]; }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int n = 0; n < NU; ++n) {
            const int u = tid + n * kConvThreads, chn = u >> 2, g = u & 3;
            const int o = g * BC + chn;
            if constexpr (H16) {
                u32x4 p0, p1;
                split8h(rd[n], sc_d, p0, p1);
                sD[o] = p0; sD[KG * BC + o] = p1;
                split8h(rx[n], sc_x, p0, p1);
                sX[o] = p0; sX[KG * BC + o] = p1;
            } else {
                u32x4 p0, p1, p2;
                split8(rd[n], p0, p1, p2);
                sD[o] = p0; sD[KG * BC + o] = p1; sD[2 * KG * BC + o] = p2;
                split8(rx[n], p0, p1, p2);
                sX[o] = p0; sX[KG * BC + o] = p1; sX[2 * KG * BC + o] = p2;
            }
        }
    };

    f32x16 acc[2][2];
    f32x16 accb[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[m][0][i] = 0.f; acc[m][1][i] = 0.f; accb[m][i] = 0.f; }
    }
    constexpr unsigned kOnes = H16 ? 0x3C003C00u : 0x3F803F80u;                          // two fp16 / bf16 1.0
    const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{kOnes, kOnes, kOnes, kOnes});

    const int aoff = wco * 64 + r, boff = wci * 64 + r;
    if (c_begin < c_end) { load_chunk(c_begin); store_chunk(); }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        if (more) load_chunk(c + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2][NP], bfr[2][NP];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    af[m][p] = __builtin_bit_cast(bf16x8, sD[(p * KG + 2 * ks + h) * BC + aoff + m * 32]);
                    bfr[m][p] = __builtin_bit_cast(bf16x8, sX[(p * KG + 2 * ks + h) * BC + boff + m * 32]);
                }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x16 cc = acc[m][n];
                    if constexpr (H16) {
                        const auto H = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(af[m][1]), H(bfr[n][0]), cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(af[m][0]), H(bfr[n][1]), cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H(af[m][0]), H(bfr[n][0]), cc, 0, 0, 0);
                    } else {
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][2], bfr[n][0], cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bfr[n][2], cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][1], bfr[n][1], cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][1], bfr[n][0], cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bfr[n][1], cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bfr[n][0], cc, 0, 0, 0);
                    }
                    acc[m][n] = cc;
                }
                if (want_bias && wci == 0) {          // every dy element once: row sums against a matrix of ones
                    if constexpr (H16) {
                        accb[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[m][1]), __builtin_bit_cast(f16x8, ones), accb[m], 0, 0, 0);
                        accb[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[m][0]), __builtin_bit_cast(f16x8, ones), accb[m], 0, 0, 0);
                    } else {
                        accb[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][2], ones, accb[m], 0, 0, 0);
                        accb[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][1], ones, accb[m], 0, 0, 0);
                        accb[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], ones, accb[m], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        if (more) store_chunk();
        __syncthreads();
    }

    // C/D of 32x32: lane (col = r, h) register i holds row 8 (i >> 2) + 4 h + (i & 3): rows = output channels, columns = input channels
    float* slab = a.slab + (size_t)split * a.CoutS * a.CinS;
    float chk = 0.f;
    const f32x2 id2 = bcast_lo(inv_d), ix2 = bcast_lo(inv_x);      // (packed-operand rule of conv_small.hpp: broadcasts from low registers)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 v = f32x2{acc[m][n][i], acc[m][n][i + 1]};
                if constexpr (H16) { v = (v * id2) * ix2; chk = __builtin_fmaf(v.x, 0.f, chk); chk = __builtin_fmaf(v.y, 0.f, chk); }      // exact: powers of two
                const int co = co0 + wco * 64 + m * 32 + 8 * (i >> 2) + 4 * h + (i & 3), ci = ci0 + wci * 64 + n * 32 + r;
                slab[(size_t)co * a.CinS + ci] = v.x;
                slab[(size_t)(co + 1) * a.CinS + ci] = v.y;
            }
        }
    if constexpr (H16) report_nonfinite(a.err, chk, UAPS_ERR_WRW_NONFINITE);
    if (want_bias && wci == 0 && r == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 v = f32x2{accb[m][i], accb[m][i + 1]};
                if constexpr (H16) v = v * id2;
                float* dst = a.bslab + (size_t)split * a.CoutS + co0 + wco * 64 + m * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                dst[0] = v.x; dst[1] = v.y;
            }
    }
}

static __global__ __launch_bounds__(kConvThreads, 2) void conv_gw1h_kernel(ConvWrwArgs a) { conv_gw1_body<true>(a); }
static __global__ __launch_bounds__(kConvThreads, 2) void conv_gw1s_kernel(ConvWrwArgs a) { conv_gw1_body<false>(a); }

}  // namespace uaps
