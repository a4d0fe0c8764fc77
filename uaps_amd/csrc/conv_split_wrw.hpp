// Weight gradient of the 3x3 convolutions on the bf16 matrix pipe with the exact three-way operand split of conv_split.hpp
// (six partial products per multiply, fp32 accumulation: the error of an fp32 fma chain).
//
//   dw[co][ci][ky][kx] = sum_{b,y,x} dy[b][co][y][x] * in[b][ci][y + ky - 1][x + kx - 1]
//
// GEMM view: M = 16 output channels, N = 16 input channels, K = pixels, one accumulator tile per tap.  v_mfma_f32_16x16x32_bf16
// takes 8 consecutive k per lane (k-group g = lane >> 4): a k-group is 8 consecutive pixels of a tile row, an MFMA covers one
// 32-pixel tile row.
//   * LDS holds the dy tile and the haloed input tile already split: [piece][channel][row][32 pixels] bf16, a fragment is one
//     ds_read_b128; channel planes are padded to stride == 32 (mod 64) bytes, which makes the four 16-lane groups the
//     hardware services together hit disjoint banks (lane (c, g) -> 16-byte slot 2c + g (mod 16));
//   * the three column taps need the input fragment shifted by -1 / 0 / +1 pixels = 16 bits of packed bf16: five
//     v_alignbit_b32 per piece build both shifted fragments from the aligned one and one "edge" dword per k-group that the
//     staging pass stores beside the row (high half = the pixel left of the group, low half = the pixel right of it);
//   * a wave owns one (16 co, 16 ci) block (and a share of the tile rows when the workgroup's block is smaller than 32 x 32):
//     every staged input row is turned into its 9 fragments (3 pieces x 3 shifts) once and used for the three row taps.
// Partials go to per-split slabs [split][tap][CoutS][CinS] like conv_wrw_kernel; conv_wrw_reduce_kernel sums them.
#pragma once
#include "conv_split.hpp"

namespace uaps {

// Diagnostic builds only (make -C uaps_amd/csrc wrwabl; tools/diag/wrw_ablate.sh): bits of UAPS_WRW_ABLATE remove ONE cost of
// conv_swrw_body each -- timing only, the results are meaningless.  1: the column taps use the aligned fragment (no v_perm / register
// assembly of the two shifted ones); 2: no edge dwords (their two 4-byte fetches per unit, their split, their LDS traffic); 4: every
// fetch unpredicated at clamped coordinates (no branch per staged unit); 8: the two halo rows of the input tile are neither fetched
// nor staged.  0 = the shipped kernel.
#ifndef UAPS_WRW_ABLATE
#define UAPS_WRW_ABLATE 0
#endif

template <int TH, int WCO, int WCI, int DIL = 1> struct SWrwCfg {
    static constexpr int TW = 32, WR = 4 / (WCO * WCI), RPW = TH / WR;          // rows of the tile per wave
    static constexpr int BCO = 16 * WCO, BCI = 16 * WCI;
    static constexpr int XG = DIL > 1 ? 6 : 4;                                   // 8-pixel groups per staged input row (dilated: one halo group either side)
    static constexpr int DPLU = TH * 4 + 2;                                      // dy plane stride in 16-byte units (== 2 mod 4)
    static constexpr int XPLU = (TH + 2) * XG + 2;                               // input plane stride in 16-byte units
    static constexpr int EPL = (TH + 2) * 4 + 1;                                 // edge dwords per input channel (odd stride)
    static constexpr int ND = (BCO * TH * 4 + kConvThreads - 1) / kConvThreads;  // dy staging units per thread (8 pixels each)
    static constexpr int NX = (BCI * (TH + 2) * XG + kConvThreads - 1) / kConvThreads;
    static_assert(WCO * WCI * WR == 4 && TH % WR == 0, "wave arrangement");
};

// 8 consecutive floats -> three 16-byte rows of packed bf16 pieces
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned a, b, c;
        conv_split3(v[2 * i], v[2 * i + 1], a, b, c);
        p0[i] = a; p1[i] = b; p2[i] = c;
    }
}

// 8 consecutive floats times the power-of-two scale sc -> two 16-byte rows of packed fp16 pieces.  The scale is pinned to the
// low register of a pair (bcast_lo): the packed multiply then broadcasts from the low half, never `op_sel` low-from-high
// (the packed-operand rule of conv_small.hpp; an SGPR-pair broadcast would be the first-source form, unproven either way).
__device__ __forceinline__ void split8h(const float (&v)[8], float sc, u32x4& p0, u32x4& p1) {
    const f32x2 sc2 = bcast_lo(sc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 t = f32x2{v[2 * i], v[2 * i + 1]} * sc2;
        unsigned a, b;
        conv_split2h(t.x, t.y, a, b);
        p0[i] = a; p1[i] = b;
    }
}

// H16: the two-piece fp16 form (three partial products); dy and the input are scaled by powers of two derived from the
// callers' bounds (ConvWrwArgs::dy_bound / in_bound / in2_bound), the slabs receive the unscaled partial sums
//
// DIL = 2 / 4 (the dilated 3x3 layers of utilities/resnet.py:8-10, 201-203; fp16 form only): a tile is TH rows DIL apart
// (tile row index = DIL * block + row phase), so that the three row taps still meet TH + 2 staged rows; the column taps are
// DIL pixels = DIL / 2 dwords of packed pieces away: the staged rows carry one 8-pixel halo group on either side and the two
// shifted fragments are register selections of the aligned fragment and its neighbours' dwords (no alignbit, no edge dwords).
template <int TH, int WCO, int WCI, bool XF, bool H16 = false, int DIL = 1>
__device__ __forceinline__ void conv_swrw_body(const ConvWrwArgs& a) {
    using Cfg = SWrwCfg<TH, WCO, WCI, DIL>;
    static_assert(DIL == 1 || ((DIL == 2 || DIL == 4) && H16 && !XF), "dilated form: fp16 pieces, no staging-time BatchNorm");
    constexpr int NP = H16 ? 2 : 3;
    constexpr int XG = Cfg::XG, XH = DIL > 1 ? 1 : 0;      // groups per staged row, halo groups on the left
    constexpr int TW = 32, WR = Cfg::WR, RPW = Cfg::RPW, BCO = Cfg::BCO, BCI = Cfg::BCI;
    constexpr int DPLU = Cfg::DPLU, XPLU = Cfg::XPLU, EPL = Cfg::EPL, ND = Cfg::ND, NX = Cfg::NX;
    constexpr int RED_FLOATS = WR > 1 ? (WR / 2) * WCO * WCI * 10 * 256 : 0;
    // EU (fp16 form, no dilation; round 6): the staged input rows carry a FIFTH unit per row that holds the two pixels outside the tile
    // (left neighbour of column 0 in the high half of dword 3, right neighbour of column 31 in the low half of dword 0); a lane's
    // shifted fragments then take their boundary dwords from the NEIGHBOURING units of the row already in LDS -- unit (kq + 4) % 5
    // dword 3 and unit kq + 1 dword 0 -- instead of an edge dword fetched, split and stored per staged unit (two 4-byte fetches, a
    // split and two LDS stores per unit: 9-10 % of the kernel, profiles/r06_wrw_ablation.txt).
    constexpr bool EU = H16 && DIL == 1;
    constexpr int XGS = EU ? 5 : XG;                                   // units per staged row in LDS
    constexpr int XPLS = EU ? (TH + 2) * 5 + ((TH + 2) * 5 % 4 == 2 ? 0 : (6 - (TH + 2) * 5 % 4) % 4) : XPLU;      // plane stride == 2 (mod 4)
    static_assert(XPLS % 4 == 2, "input plane stride");
    constexpr int NE = EU ? (BCI * (TH + 2) + kConvThreads - 1) / kConvThreads : 0;      // edge items (channel, row) per thread
    constexpr int STAGE_UNITS = NP * BCO * DPLU + NP * BCI * XPLS;
    constexpr int LDS_UNITS = STAGE_UNITS > RED_FLOATS / 4 ? STAGE_UNITS : RED_FLOATS / 4;

    __shared__ __attribute__((aligned(16))) u32x4 smem[LDS_UNITS];
    __shared__ unsigned sE[(DIL == 1 && !EU) ? NP * BCI * EPL : 1];
    __shared__ f32x2 sXf[XF ? kWrwMaxGroups * BCI + 1 : 1];
    u32x4* sD = smem;                        // [piece][co][row][4 groups] (+ 2 pad units per plane)
    u32x4* sX = smem + NP * BCO * DPLU;      // [piece][ci][row][4 groups] (+ 2 pad units per plane)
    constexpr int XF_ZERO = kWrwMaxGroups * BCI;

    float sc_d = 1.f, sc_x = 1.f, inv_d = 1.f, inv_x = 1.f;
    if constexpr (H16) {
        const f32x2 sd = h16_scale(bound_of(a.dy_bound, a.dy_mul));
        const f32x2 sx = h16_scale(__builtin_fmaxf(bound_of(a.in_bound, a.in_mul), bound_of(a.in2_bound, a.in2_mul)));
        sc_d = sd.x; inv_d = sd.y; sc_x = sx.x; inv_x = sx.y;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int wr = wave % WR, wci = (wave / WR) % WCI, wco = wave / (WR * WCI);

    int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (bid >= a.nsplit * a.ncob * a.ncib) return;
    stagger_by_wave_slot();
    const int cib = bid % a.ncib; bid /= a.ncib;
    const int cob = bid % a.ncob;
    const int split = bid / a.ncob;
    const int co0 = cob * BCO, ci0 = cib * BCI;
    const int HW = a.H * a.W;
    const int tiles_per_img = a.tiles_x * a.tiles_y, ntiles = a.B * tiles_per_img;
    const int t_begin = (int)((long)ntiles * split / a.nsplit), t_end = (int)((long)ntiles * (split + 1) / a.nsplit);
    const bool want_bias = a.bslab != nullptr && cib == 0;

    // ---- staging units: (channel, row, k-group) -> 8 pixels; tile-independent parts ----
    int dC[ND], dR[ND], dG[ND];
    int xC[NX], xR[NX], xG[NX];
#pragma unroll
    for (int n = 0; n < ND; ++n) {
        const int u = tid + n * kConvThreads;
        dC[n] = u < BCO * TH * 4 ? u / (TH * 4) : -1; dR[n] = (u / 4) % TH; dG[n] = u % 4;
    }
#pragma unroll
    for (int n = 0; n < NX; ++n) {
        const int u = tid + n * kConvThreads;
        xC[n] = u < BCI * (TH + 2) * XG ? u / ((TH + 2) * XG) : -1; xR[n] = (u / XG) % (TH + 2); xG[n] = u % XG;
    }
    if constexpr (XF) {
        const int G = a.B / a.xf_Bg;
        for (int i = tid; i < G * BCI; i += kConvThreads) {
            const int g = i / BCI, ch = ci0 + i % BCI;
            f32x2 v = f32x2{0.f, 0.f};
            if (ch < a.Cin) { const float2 t = a.xf[(size_t)g * a.Cin + ch]; v = f32x2{t.x, t.y}; }
            sXf[i] = v;
        }
        if (tid == 0) sXf[XF_ZERO] = f32x2{0.f, 0.f};
    }

    float rd[ND][8];
    float rx[NX][(DIL == 1 && !EU) ? 10 : 8];     // 8 pixels, then (bf16 form) the pixel left of the group and the pixel right of it
    float re[EU ? NE : 1][2];                     // EU: (left, right) outside pixels of this thread's edge items
    int eC[EU ? NE : 1], eR[EU ? NE : 1];
    if constexpr (EU) {
#pragma unroll
        for (int n = 0; n < NE; ++n) {
            const int i = tid + n * kConvThreads;
            eC[n] = i < BCI * (TH + 2) ? i / (TH + 2) : -1; eR[n] = i % (TH + 2);
        }
    }
    int xf_idx[XF ? NX : 1];
    int exf_idx[(XF && EU) ? NE : 1];

    auto load_tile = [&](int t) {
        const int b = t / tiles_per_img, tt = t % tiles_per_img;
        const int ry = a.col_major ? tt % a.tiles_y : tt / a.tiles_x, x0 = (a.col_major ? tt / a.tiles_y : tt % a.tiles_x) * TW;
        const int y0 = DIL == 1 ? ry * TH : ry % DIL + (ry / DIL) * (TH * DIL);      // dilated: rows y0 + DIL * r
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            const int c = co0 + dC[n], gy = y0 + dR[n] * DIL, gx = x0 + dG[n] * 8;
#if UAPS_WRW_ABLATE & 4
            {
                const int cc = c < 0 ? 0 : (c < a.Cout ? c : a.Cout - 1), yy = gy < a.H ? gy : a.H - 1, xx = gx + 8 <= a.W ? gx : a.W - 8;
                const float* q = a.dout + ((size_t)b * a.Cout + cc) * HW + yy * a.W + xx;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(q), w1 = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { rd[n][k] = w0[k]; rd[n][4 + k] = w1[k]; }
                continue;
            }
#endif
            const bool ok = dC[n] >= 0 && c < a.Cout && gy < a.H && gx < a.W;             // W % 4 == 0: float4 pieces are all in or all out
            const float* p = a.dout + ((size_t)b * a.Cout + (ok ? c : 0)) * HW + (ok ? gy * a.W + gx : 0);
            const f32x4 v0 = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 v1 = (ok && gx + 4 < a.W) ? *reinterpret_cast<const f32x4*>(p + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) { rd[n][k] = v0[k]; rd[n][4 + k] = v1[k]; }
        }
#pragma unroll
        for (int n = 0; n < NX; ++n) {
            const int c = ci0 + xC[n], gy = y0 + (xR[n] - 1) * DIL, gx = x0 + (xG[n] - XH) * 8;
#if UAPS_WRW_ABLATE & 8
            if (xR[n] == 0 || xR[n] == TH + 1) continue;
#endif
#if UAPS_WRW_ABLATE & 4
            {
                const int cc = c < 0 ? 0 : (c < a.Csplit ? c : a.Csplit - 1), yy = gy < 0 ? 0 : (gy < a.H ? gy : a.H - 1);
                const int xx = gx < 0 ? 0 : (gx + 8 <= a.W ? gx : a.W - 8);              // (timing layers: W % 8 == 0)
                const float* q = a.in + ((size_t)b * a.Csplit + cc) * HW + yy * a.W;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(q + xx), w1 = *reinterpret_cast<const f32x4*>(q + xx + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { rx[n][k] = w0[k]; rx[n][4 + k] = w1[k]; }
                if constexpr (DIL == 1 && !EU) {
#if !(UAPS_WRW_ABLATE & 2)
                    rx[n][8] = q[xx > 0 ? xx - 1 : 0]; rx[n][9] = q[xx + 8 < a.W ? xx + 8 : a.W - 1];
#else
                    rx[n][8] = 0.f; rx[n][9] = 0.f;
#endif
                }
                if constexpr (XF) xf_idx[n] = (b / a.xf_Bg) * BCI + (xC[n] < 0 ? 0 : xC[n]);
                continue;
            }
#endif
            const bool second = c >= a.Csplit;
            const float* src = second ? a.in2 + ((size_t)b * (a.Cin - a.Csplit) + (c - a.Csplit)) * HW : a.in + ((size_t)b * a.Csplit + c) * HW;
            const bool okc = xC[n] >= 0 && c < a.Cin && gy >= 0 && gy < a.H;
            const bool ok = okc && gx >= 0 && gx < a.W;
            const float* p = src + (okc ? gy * a.W : 0) + (ok ? gx : 0);
            const f32x4 v0 = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 v1 = (ok && gx + 4 < a.W) ? *reinterpret_cast<const f32x4*>(p + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) { rx[n][k] = v0[k]; rx[n][4 + k] = v1[k]; }
            if constexpr (DIL == 1 && !EU) {
#if UAPS_WRW_ABLATE & 2
                rx[n][8] = 0.f; rx[n][9] = 0.f;
#else
                rx[n][8] = (okc && gx - 1 >= 0 && gx - 1 < a.W) ? src[gy * a.W + gx - 1] : 0.f;
                rx[n][9] = (okc && gx + 8 < a.W) ? src[gy * a.W + gx + 8] : 0.f;
#endif
            }
            if constexpr (XF) xf_idx[n] = okc ? (b / a.xf_Bg) * BCI + xC[n] : XF_ZERO;     // padding rows / channels stay zero
        }
        if constexpr (EU) {                      // the pixel left of the tile's first column and the pixel right of its last, per (channel, row)
#pragma unroll
            for (int n = 0; n < NE; ++n) {
                const int c = ci0 + eC[n], gy = y0 + eR[n] - 1;
                const bool second = c >= a.Csplit;
                const float* src = second ? a.in2 + ((size_t)b * (a.Cin - a.Csplit) + (c - a.Csplit)) * HW : a.in + ((size_t)b * a.Csplit + c) * HW;
                const bool okc = eC[n] >= 0 && c < a.Cin && gy >= 0 && gy < a.H;
#if UAPS_WRW_ABLATE & 2
                re[n][0] = 0.f; re[n][1] = 0.f;
#else
                re[n][0] = (okc && x0 - 1 >= 0) ? src[gy * a.W + x0 - 1] : 0.f;
                re[n][1] = (okc && x0 + TW < a.W) ? src[gy * a.W + x0 + TW] : 0.f;
#endif
                if constexpr (XF) exf_idx[n] = okc ? (b / a.xf_Bg) * BCI + eC[n] : XF_ZERO;
            }
        }
    };
    auto store_tile = [&](int t) {
        const int tt = t % tiles_per_img;
        const int x0 = (a.col_major ? tt / a.tiles_y : tt % a.tiles_x) * TW;
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            if (dC[n] < 0) continue;
            const int u = dC[n] * DPLU + dR[n] * 4 + dG[n];
            if constexpr (H16) {
                u32x4 p0, p1;
                split8h(rd[n], sc_d, p0, p1);
                sD[u] = p0; sD[BCO * DPLU + u] = p1;
            } else {
                u32x4 p0, p1, p2;
                split8(rd[n], p0, p1, p2);
                sD[u] = p0; sD[BCO * DPLU + u] = p1; sD[2 * BCO * DPLU + u] = p2;
            }
        }
#pragma unroll
        for (int n = 0; n < NX; ++n) {
            if (xC[n] < 0) continue;
#if UAPS_WRW_ABLATE & 8
            if (xR[n] == 0 || xR[n] == TH + 1) continue;
#endif
            if constexpr (XF) {                  // leaky_relu(fma(y, scale, shift)) of the raw conv output; columns outside the image stay zero
                const f32x2 cf = sXf[xf_idx[n]];
                const int gx = x0 + xG[n] * 8;
#pragma unroll
                for (int k = 0; k < ((DIL == 1 && !EU) ? 10 : 8); ++k) {
                    const int col = k < 8 ? gx + k : (k == 8 ? gx - 1 : gx + 8);
                    const float z = __builtin_fmaf(rx[n][k], cf.x, cf.y);
                    rx[n][k] = (col >= 0 && col < a.W) ? __builtin_fmaxf(z, z * a.xf_slope) : 0.f;
                }
            }
            float v8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] = rx[n][k];
            const int u = xC[n] * XPLS + xR[n] * XGS + xG[n];
            const int eu = xC[n] * EPL + xR[n] * 4 + xG[n];
            if constexpr (DIL > 1) {
                u32x4 p0, p1;
                split8h(v8, sc_x, p0, p1);
                sX[u] = p0; sX[BCI * XPLS + u] = p1;
            } else if constexpr (EU) {
                u32x4 p0, p1;
                split8h(v8, sc_x, p0, p1);
                sX[u] = p0; sX[BCI * XPLS + u] = p1;
                (void)eu;
            } else if constexpr (H16) {
                u32x4 p0, p1;
                split8h(v8, sc_x, p0, p1);
                sX[u] = p0; sX[BCI * XPLS + u] = p1;
#if !(UAPS_WRW_ABLATE & 2)
                unsigned e0, e1;                 // (right neighbour, left neighbour) -> low / high half of the edge dword
                const f32x2 ev = f32x2{rx[n][9], rx[n][8]} * bcast_lo(sc_x);
                conv_split2h(ev.x, ev.y, e0, e1);
                sE[eu] = e0; sE[BCI * EPL + eu] = e1;
#else
                (void)eu;
#endif
            } else {
                u32x4 p0, p1, p2;
                split8(v8, p0, p1, p2);
                sX[u] = p0; sX[BCI * XPLS + u] = p1; sX[2 * BCI * XPLS + u] = p2;
                unsigned e0, e1, e2;             // (right neighbour, left neighbour) -> low / high half of the edge dword
                conv_split3(rx[n][9], rx[n][8], e0, e1, e2);
                sE[eu] = e0; sE[BCI * EPL + eu] = e1; sE[2 * BCI * EPL + eu] = e2;
            }
        }
        if constexpr (EU) {
#if !(UAPS_WRW_ABLATE & 2)
#pragma unroll
            for (int n = 0; n < NE; ++n) {
                if (eC[n] < 0) continue;
                float l = re[n][0], r = re[n][1];
                if constexpr (XF) {              // the same transform as the row's pixels; columns outside the image stay zero
                    const f32x2 cf = sXf[exf_idx[n]];
                    const float zl = __builtin_fmaf(l, cf.x, cf.y), zr = __builtin_fmaf(r, cf.x, cf.y);
                    l = x0 - 1 >= 0 ? __builtin_fmaxf(zl, zl * a.xf_slope) : 0.f;
                    r = x0 + TW < a.W ? __builtin_fmaxf(zr, zr * a.xf_slope) : 0.f;
                }
                unsigned e0, e1;                 // (right neighbour, left neighbour) -> low / high half: dword 0 serves the row's last unit, dword 3 its first
                const f32x2 ev = f32x2{r, l} * bcast_lo(sc_x);
                conv_split2h(ev.x, ev.y, e0, e1);
                const int ue = eC[n] * XPLS + eR[n] * XGS + 4;
                sX[ue] = u32x4{e0, e0, e0, e0}; sX[BCI * XPLS + ue] = u32x4{e1, e1, e1, e1};
            }
#endif
        }
    };

    f32x4 acc[9];
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr unsigned kOnes = H16 ? 0x3C003C00u : 0x3F803F80u;                          // two fp16 / bf16 1.0
    const u32x4 ones_u = u32x4{kOnes, kOnes, kOnes, kOnes};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

    const int aoff = (wco * 16 + j) * DPLU + kq;         // + row * 4 (+ piece plane)
    const int boff = (wci * 16 + j) * XPLS + kq + XH;    // + row * XGS
    const int eoff = (wci * 16 + j) * EPL + kq;
    // EU: dword indices (within a staged row of 5 units) of the dword left of this lane's k-group and of the dword right of it
    const int elidx = ((kq + 4) % 5) * 4 + 3, eridx = (kq + 1) * 4;

    if (t_begin < t_end) { load_tile(t_begin); }
    if constexpr (XF) __syncthreads();                   // sXf visible
    if (t_begin < t_end) store_tile(t_begin);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const bool more = t + 1 < t_end;
        if (more) load_tile(t + 1);
        // this wave's dy rows [wr*RPW, wr*RPW + RPW) meet the input rows [wr*RPW, wr*RPW + RPW + 2) (LDS row r = image row y0 - 1 + r)
#pragma unroll
        for (int rr = 0; rr < RPW + 2; ++rr) {
            const int r = wr * RPW + rr;
            bf16x8 bf[3][NP];                             // [shift kx][piece]
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const u32x4 c = sX[p * BCI * XPLS + boff + r * XGS];
                if constexpr (DIL > 1) {                  // neighbours' dwords: pixels x - DIL .. x - 1 and x + 8 .. x + 7 + DIL
                    const unsigned* cl = reinterpret_cast<const unsigned*>(&sX[p * BCI * XPLS + boff + r * XGS - 1]);
                    const unsigned* cr = reinterpret_cast<const unsigned*>(&sX[p * BCI * XPLS + boff + r * XGS + 1]);
                    if constexpr (DIL == 2) {
                        bf[0][p] = __builtin_bit_cast(bf16x8, u32x4{cl[3], c[0], c[1], c[2]});
                        bf[2][p] = __builtin_bit_cast(bf16x8, u32x4{c[1], c[2], c[3], cr[0]});
                    } else {
                        bf[0][p] = __builtin_bit_cast(bf16x8, u32x4{cl[2], cl[3], c[0], c[1]});
                        bf[2][p] = __builtin_bit_cast(bf16x8, u32x4{c[2], c[3], cr[0], cr[1]});
                    }
                    bf[1][p] = __builtin_bit_cast(bf16x8, c);
                    continue;
                }
#if UAPS_WRW_ABLATE & 1
                bf[0][p] = __builtin_bit_cast(bf16x8, c); bf[1][p] = __builtin_bit_cast(bf16x8, c); bf[2][p] = __builtin_bit_cast(bf16x8, c);
                continue;
#endif
                unsigned el, er;                          // high half of el = the pixel left of the group, low half of er = the pixel right of it
#if UAPS_WRW_ABLATE & 2
                el = er = 0u;
#else
                if constexpr (EU) {
                    const unsigned* rowd = reinterpret_cast<const unsigned*>(&sX[p * BCI * XPLS + (wci * 16 + j) * XPLS + r * XGS]);
                    el = rowd[elidx]; er = rowd[eridx];
                } else {
                    el = er = sE[p * BCI * EPL + eoff + r * 4];
                }
#endif
                const unsigned t01 = __builtin_amdgcn_alignbit(c[1], c[0], 16), t12 = __builtin_amdgcn_alignbit(c[2], c[1], 16);
                const unsigned t23 = __builtin_amdgcn_alignbit(c[3], c[2], 16);
                const unsigned tE0 = __builtin_amdgcn_alignbit(c[0], el, 16), t3E = __builtin_amdgcn_alignbit(er, c[3], 16);
                bf[0][p] = __builtin_bit_cast(bf16x8, u32x4{tE0, t01, t12, t23});          // pixels x - 1
                bf[1][p] = __builtin_bit_cast(bf16x8, c);
                bf[2][p] = __builtin_bit_cast(bf16x8, u32x4{t01, t12, t23, t3E});          // pixels x + 1
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = rr - ky;                   // dy row (relative to this wave's first) that meets input row r at row tap ky
                if (yy < 0 || yy >= RPW) continue;
                bf16x8 af[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) af[p] = __builtin_bit_cast(bf16x8, sD[p * BCO * DPLU + aoff + (wr * RPW + yy) * 4]);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    f32x4 c = acc[ky * 3 + kx];
                    if constexpr (H16) {
                        const auto H = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[1]), H(bf[kx][0]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[0]), H(bf[kx][1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(af[0]), H(bf[kx][0]), c, 0, 0, 0);
                    } else {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bf[kx][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[kx][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[kx][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[kx][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[kx][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[kx][0], c, 0, 0, 0);
                    }
                    acc[ky * 3 + kx] = c;
                }
                if (ky == 0 && want_bias && wci == 0) {   // every dy row exactly once
                    if constexpr (H16) {
                        accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[1]), __builtin_bit_cast(f16x8, ones), accb, 0, 0, 0);
                        accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, ones), accb, 0, 0, 0);
                    } else {
                        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], ones, accb, 0, 0, 0);
                        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], ones, accb, 0, 0, 0);
                        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], ones, accb, 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        if (more) store_tile(t + 1);
        __syncthreads();
    }

    // ---- sum the WR row-split partials of each (wco, wci) block through LDS (fixed order) ----
    if constexpr (WR > 1) {
        float* red = reinterpret_cast<float*>(smem);     // [slot][10][4][64]
#pragma unroll
        for (int s = WR / 2; s >= 1; s >>= 1) {
            if (wr >= s && wr < 2 * s) {
                float* p = red + (size_t)(((wr - s) * WCO + wco) * WCI + wci) * 10 * 256 + lane;
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[(t * 4 + r) * 64] = acc[t][r];
#pragma unroll
                for (int r = 0; r < 4; ++r) p[(36 + r) * 64] = accb[r];
            }
            __syncthreads();
            if (wr < s) {
                const float* p = red + (size_t)((wr * WCO + wco) * WCI + wci) * 10 * 256 + lane;
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] += p[(t * 4 + r) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) accb[r] += p[(36 + r) * 64];
            }
            __syncthreads();
        }
    }
    if (wr != 0) return;
    // lane (j, kq), register r: co = co0 + wco*16 + kq*4 + r, ci = ci0 + wci*16 + j
    float* slab = a.slab + (size_t)split * 9 * a.CoutS * a.CinS;
    float chk = 0.f;
    const f32x2 id2 = bcast_lo(inv_d), ix2 = bcast_lo(inv_x);      // (packed-operand rule of conv_small.hpp: broadcasts from low registers)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if constexpr (H16) note_nonfinite(chk, acc[t]);
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            f32x2 v = f32x2{acc[t][r], acc[t][r + 1]};
            if constexpr (H16) v = (v * id2) * ix2;                  // exact: powers of two
            const int co = co0 + wco * 16 + kq * 4 + r, ci = ci0 + wci * 16 + j;
            slab[((size_t)t * a.CoutS + co) * a.CinS + ci] = v.x;
            slab[((size_t)t * a.CoutS + co + 1) * a.CinS + ci] = v.y;
        }
    }
    if constexpr (H16) report_nonfinite(a.err, chk, UAPS_ERR_WRW_NONFINITE);
    if (want_bias && wci == 0 && j == 0) {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            f32x2 v = f32x2{accb[r], accb[r + 1]};
            if constexpr (H16) v = v * id2;
            a.bslab[(size_t)split * a.CoutS + co0 + wco * 16 + kq * 4 + r] = v.x;
            a.bslab[(size_t)split * a.CoutS + co0 + wco * 16 + kq * 4 + r + 1] = v.y;
        }
    }
}

template <int TH, int WCO, int WCI>
__global__ __launch_bounds__(kConvThreads, 2) void conv_swrw_kernel(ConvWrwArgs a) { conv_swrw_body<TH, WCO, WCI, false>(a); }
template <int TH, int WCO, int WCI>
__global__ __launch_bounds__(kConvThreads, 2) void conv_swrw_bn_kernel(ConvWrwArgs a) { conv_swrw_body<TH, WCO, WCI, true>(a); }
template <int TH, int WCO, int WCI>
__global__ __launch_bounds__(kConvThreads, 2) void conv_hwrw_kernel(ConvWrwArgs a) { conv_swrw_body<TH, WCO, WCI, false, true>(a); }
template <int TH, int WCO, int WCI>
__global__ __launch_bounds__(kConvThreads, 2) void conv_hwrw_bn_kernel(ConvWrwArgs a) { conv_swrw_body<TH, WCO, WCI, true, true>(a); }
template <int TH, int WCO, int WCI, int DIL>
__global__ __launch_bounds__(kConvThreads, 2) void conv_hwrw_d_kernel(ConvWrwArgs a) { conv_swrw_body<TH, WCO, WCI, false, true, DIL>(a); }

}  // namespace uaps
