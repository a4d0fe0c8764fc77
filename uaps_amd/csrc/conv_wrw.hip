// extern "C" entry points of the weight-gradient convolution (include/uaps_hip.h, "Convolutions").
#include "../../include/uaps_hip.h"
#include <stdio.h>
#include "conv_kernels.hpp"
#include "conv_split_wrw.hpp"
#include "conv_split_wrw_row.hpp"
#include "conv_small.hpp"
#include "conv_gemm1x1.hpp"
#include "hints.hpp"
using namespace uaps;

extern "C" int uaps_conv_get_mode(void);
extern "C" unsigned uaps_conv_get_tuning(void);

namespace {

// wave arrangement: (WCO, WCI) 16-channel blocks per workgroup, the remaining factor of 4 splits the tile rows
// split: the bf16-split kernels of conv_split_wrw.hpp (3x3, no dilation, 16-byte rows, >= 16 input channels)
// g1: the GEMM-tiled 1x1 kernel of conv_gemm1x1.hpp (128 x 128 channel blocks, 32-pixel chunks)
struct WrwPlan { int TH, TW, wco, wci, ncob, ncib, nsplit, CoutS, CinS, dil; long tiles; bool split, small, g1; };

WrwPlan plan_wrw(int B, int Cin, int Cout, int H, int W, int cfg, int ks = 3) {
    WrwPlan p{};
    p.dil = ((cfg >> 24) & 0xf) ? ((cfg >> 24) & 0xf) : 1;       // bits 24-27: dilation; bit 28: exact kernels; bit 29: split kernels; low bits: pixel splits override
    const bool force_exact = (cfg >> 28) & 1, force_split = (cfg >> 29) & 1;
    cfg &= 0xffffff;
    p.split = ks == 3 && p.dil == 1 && W % 4 == 0 && !force_exact && (force_split || (uaps_conv_get_mode() >= 1 && Cin >= 16 && W >= 16));      // 32-pixel row tiles: half empty on 16-wide maps, still 1.2x the fp32 kernel there (256 -> 256 @16^2, B = 32: 97 -> 79 us)
    if (uaps_conv_get_tuning() & UAPS_TUNE_NO_SPLIT_WRW) p.split = false;
    // <= 4 output channels x 16 input channels on a wide map: the exact-N VALU kernel (conv_small.hpp), slabs [tap][4][16]
    p.small = ks == 3 && p.dil == 1 && W % 4 == 0 && W >= 64 && Cout <= 4 && Cin == 16 && !force_exact && !force_split;
    // wide 1x1 projections (ResNet bottlenecks): the same rule as plan_fwd's big_1x1
    p.g1 = ks == 1 && uaps_conv_get_mode() >= 1 && !force_exact && !(uaps_conv_get_tuning() & UAPS_TUNE_NO_SPLIT_WRW) && cfg == 0 &&
           Cin >= 64 && (long)Cin * Cout >= 16384 && (long)H * W >= 1024 && ((long)H * W) % 32 == 0 && W % 4 == 0;
    if (p.g1) {
        p.split = p.small = false;
        p.wco = p.wci = 8; p.TH = 1; p.TW = 32;
        p.ncob = (Cout + 127) / 128; p.ncib = (Cin + 127) / 128;
        p.CoutS = p.ncob * 128; p.CinS = p.ncib * 128;
        p.tiles = (long)B * (((long)H * W) / 32);
        const long blocks = (long)p.ncob * p.ncib;
        long want = blocks >= 1024 ? 1 : (1024 + blocks - 1) / blocks;      // ~4 workgroups per CU in all
        if (want > p.tiles) want = p.tiles;
        p.nsplit = (int)(want < 1 ? 1 : want);
        return p;
    }
    if (p.small) {
        p.split = false;
        p.wco = p.wci = 1; p.TH = 8; p.TW = 64; p.ncob = p.ncib = 1; p.CoutS = 4; p.CinS = 16;
        p.tiles = (long)B * ((H + p.TH - 1) / p.TH) * ((W + p.TW - 1) / p.TW);
        long want = cfg > 0 ? cfg : 512;
        if (want > p.tiles) want = p.tiles;
        p.nsplit = (int)(want < 1 ? 1 : want);
        return p;
    }
    if (p.split) {
        p.wco = Cout > 16 ? 2 : 1; p.wci = Cin > 16 ? 2 : 1;
        p.TH = 4; p.TW = 32;
        p.ncob = (Cout + 16 * p.wco - 1) / (16 * p.wco);
        p.ncib = (Cin + 16 * p.wci - 1) / (16 * p.wci);
        p.CoutS = p.ncob * 16 * p.wco; p.CinS = p.ncib * 16 * p.wci;
        p.tiles = (long)B * ((H + p.TH - 1) / p.TH) * ((W + p.TW - 1) / p.TW);
        const long blocks = (long)p.ncob * p.ncib;
        long want = blocks >= 512 ? 1 : 512 / blocks;
        if (cfg > 0) want = cfg;
        if (want > p.tiles) want = p.tiles;
        if (want < 1) want = 1;
        p.nsplit = (int)want;
        return p;
    }
    p.wco = Cout > 16 ? 2 : 1;
    p.wci = Cin > 16 ? 2 : 1;
    if (p.dil > 1) p.wco = p.wci = 2;                             // dilated variants exist for 32 x 32 channel blocks only
    const bool wide = W >= 32 || p.dil > 1, big = p.wco * p.wci == 4;     // 32x32 channel blocks stage half-height tiles (LDS)
    p.TW = wide ? 32 : 16;
    p.TH = wide ? (big ? 4 : 8) : (big ? 8 : 16);
    p.ncob = (Cout + 16 * p.wco - 1) / (16 * p.wco);
    p.ncib = (Cin + 16 * p.wci - 1) / (16 * p.wci);
    p.CoutS = p.ncob * 16 * p.wco; p.CinS = p.ncib * 16 * p.wci;
    p.tiles = (long)B * ((H + p.TH - 1) / p.TH) * ((W + p.TW - 1) / p.TW);
    // pixel splits: the kernels run two workgroups per CU (register-limited), so aim at exactly one round of
    // 512 equally loaded workgroups; more, smaller ones leave a half-empty second round and grow the slabs
    const long blocks = (long)p.ncob * p.ncib;
    long want = blocks >= 512 ? 1 : 512 / blocks;
    if (cfg > 0) want = cfg;
    if (want > p.tiles) want = p.tiles;
    if (want < 1) want = 1;
    p.nsplit = (int)want;
    return p;
}

template <int TH, int WCO, int WCI>
int launch_swrw(const ConvWrwArgs& a, hipStream_t s) {
    const long grid = ((long)a.nsplit * a.ncob * a.ncib + 7) / 8 * 8;
    if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
    if (a.dy_bound) {                                 // the fp16 two-piece form
        if constexpr (WCO == 1 && TH == 4) {          // 16 output channels (the 256 x 256 level): 8-row tiles, halo rows 10/8 instead of 6/4 and
            // half the barriers -- measured 161 -> 150 us (32 -> 16 channels) and 103 -> 90 us (16 -> 16 with the staging-time
            // BatchNorm) at B = 32; the 32 x 16 channel block of the 128 x 128 level loses 10 % and keeps 4 rows
            const bool tall = !(uaps_conv_get_tuning() & UAPS_TUNE_WRW_SHORT_TILES);
            if (tall && a.H >= 8) {
                ConvWrwArgs b = a;
                b.tiles_y = (a.H + 7) / 8;
                if (b.xf) UAPS_LAUNCH_MAIN((conv_hwrw_bn_kernel<8, WCO, WCI>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, b);
                else UAPS_LAUNCH_MAIN((conv_hwrw_kernel<8, WCO, WCI>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, b);
                return (int)hipGetLastError();
            }
        }
        if (a.xf) UAPS_LAUNCH_MAIN((conv_hwrw_bn_kernel<TH, WCO, WCI>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_hwrw_kernel<TH, WCO, WCI>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        return (int)hipGetLastError();
    }
    if (a.xf) UAPS_LAUNCH_MAIN((conv_swrw_bn_kernel<TH, WCO, WCI>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    else UAPS_LAUNCH_MAIN((conv_swrw_kernel<TH, WCO, WCI>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    return (int)hipGetLastError();
}
int dispatch_swrw(const ConvWrwArgs& a, const WrwPlan& p, hipStream_t s) {
    if (p.wco == 2 && p.wci == 2) return launch_swrw<4, 2, 2>(a, s);
    if (p.wco == 2) return launch_swrw<4, 2, 1>(a, s);
    if (p.wci == 2) return launch_swrw<4, 1, 2>(a, s);
    return launch_swrw<4, 1, 1>(a, s);
}

size_t wrw_ws_floats(const WrwPlan& p, int taps) { return (size_t)p.nsplit * ((size_t)taps * p.CoutS * p.CinS + p.CoutS); }

template <int KS, int TH, int TW, int WCO, int WCI, int DIL = 1>
int launch_wrw(const ConvWrwArgs& a, bool vec, hipStream_t s) {
    const long grid = ((long)a.nsplit * a.ncob * a.ncib + 7) / 8 * 8;      // multiple of 8 for the XCD swizzle
    if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
    if (a.xf) {                  // BatchNorm + LeakyReLU of the input recomputed while staging: 16-byte form, no dilation
        if constexpr (DIL == 1) {
            if (!vec) return UAPS_ERANGE;
            UAPS_LAUNCH_MAIN((conv_wrw_bn_kernel<KS, TH, TW, WCO, WCI, 4, 1>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
            return (int)hipGetLastError();
        } else {
            return UAPS_ERANGE;
        }
    }
    if (vec) UAPS_LAUNCH_MAIN((conv_wrw_kernel<KS, TH, TW, WCO, WCI, 4, DIL>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    else UAPS_LAUNCH_MAIN((conv_wrw_kernel<KS, TH, TW, WCO, WCI, 1, DIL>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    return (int)hipGetLastError();
}

template <int KS>
int dispatch_wrw(const ConvWrwArgs& a, const WrwPlan& p, bool vec, hipStream_t s) {
    const bool wide = p.TW == 32;
    if constexpr (KS == 3) {
        if (p.dil == 2) return launch_wrw<3, 4, 32, 2, 2, 2>(a, vec, s);
        if (p.dil == 4) return launch_wrw<3, 4, 32, 2, 2, 4>(a, vec, s);
    }
    if (p.wco == 2 && p.wci == 2) return wide ? launch_wrw<KS, 4, 32, 2, 2>(a, vec, s) : launch_wrw<KS, 8, 16, 2, 2>(a, vec, s);
    if (p.wco == 2) return wide ? launch_wrw<KS, 8, 32, 2, 1>(a, vec, s) : launch_wrw<KS, 16, 16, 2, 1>(a, vec, s);
    if (p.wci == 2) return wide ? launch_wrw<KS, 8, 32, 1, 2>(a, vec, s) : launch_wrw<KS, 16, 16, 1, 2>(a, vec, s);
    return wide ? launch_wrw<KS, 8, 32, 1, 1>(a, vec, s) : launch_wrw<KS, 16, 16, 1, 1>(a, vec, s);
}

}  // namespace

extern "C" int uaps_conv_wrw_workspace_bytes(int B, int Cin, int Cout, int H, int W, int ks, int cfg, size_t* out) {
    if (!out || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (ks != 1 && ks != 3)) return UAPS_EINVAL;
    *out = wrw_ws_floats(plan_wrw(B, Cin, Cout, H, W, cfg, ks), ks * ks) * sizeof(float);
    return UAPS_OK;
}

// Step 1: per-split partial gradients into the workspace (the MFMA kernel).
static int wrw_partial_impl(const uaps_call_hints& hints, const float* dy, const float* x, const float* x2, int Csplit, int want_bias, int B, int Cin, int Cout,
                            int H, int W, int ks, int cfg, void* ws, size_t ws_bytes, uaps_stream_t stream,
                            const void* xf = nullptr, float xf_slope = 0.f, int groups = 1) {
    if (!dy || !x || !ws || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    // UAPS_CONV_X2_UP2 (include/uaps_hip.h): x2 is [B, Cin - Csplit, H / 2, W / 2], up-sampled x2 while staged
    const bool up2 = (cfg & UAPS_CONV_X2_UP2) != 0;
    cfg &= ~UAPS_CONV_X2_UP2;
    if (up2 && (!x2 || xf || H % 2 || W % 2)) return UAPS_EINVAL;
    if (xf && (x2 || groups < 1 || groups > kWrwMaxGroups || B % groups || (uintptr_t)xf % 8)) return UAPS_EINVAL;
    if (xf && !(xf_slope >= 0.f && xf_slope <= 1.f)) return UAPS_ERANGE;      // leaky_relu is evaluated as max(z, slope * z)
    if (!x2) Csplit = Cin;
    if (Csplit > Cin || (Csplit < Cin && (Csplit % 16 || Csplit == 0))) return UAPS_EINVAL;
    if (ks != 1 && ks != 3) return UAPS_ERANGE;
    if ((double)Cin * H * W * 4.0 >= 2147483648.0 || (double)Cout * H * W * 4.0 >= 2147483648.0) return UAPS_ERANGE;
    const int taps = ks * ks;
    WrwPlan p = plan_wrw(B, Cin, Cout, H, W, cfg, ks);
    const bool vec16 = (W % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0) && (!x2 || (uintptr_t)x2 % 16 == 0);
    if ((p.split || p.small) && !vec16) return UAPS_ERANGE;      // the split kernels stream 16-byte pieces (W % 4 == 0 is part of the plan; pass cfg bit 28 for odd pointers)
    if (p.dil != 1 && (ks != 3 || (p.dil != 2 && p.dil != 4))) return UAPS_ERANGE;
    if (ws_bytes < wrw_ws_floats(p, taps) * sizeof(float)) return UAPS_EWORKSPACE;
    // dy and x once each (+ the raw output read and the true dy written through when the call carries a pending BatchNorm transform)
    uaps::account_bytes(4.0 * B * H * W * ((double)Csplit + (Cin - Csplit) * (up2 ? 0.25 : 1.0) + Cout + (hints.dyt_y ? 2.0 * Cout : 0.0)));
    ConvWrwArgs a{};
    a.dout = dy; a.in = x; a.in2 = x2; a.Csplit = Csplit; a.slab = (float*)ws; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.CoutS = p.CoutS; a.CinS = p.CinS; a.tiles_x = (W + p.TW - 1) / p.TW; a.tiles_y = (H + p.TH - 1) / p.TH;
    a.col_major = (uaps_conv_get_tuning() & UAPS_TUNE_WRW_ROW_MAJOR) ? 0 : 1;      // column strips measured: 2.3x -> 1.07x of the algorithmic bytes at 32 -> 16 @ 256^2
    a.ncob = p.ncob; a.ncib = p.ncib; a.nsplit = p.nsplit;
    a.bslab = want_bias ? a.slab + (size_t)p.nsplit * taps * p.CoutS * p.CinS : nullptr;
    a.xf = (const float2*)xf; a.xf_slope = xf_slope; a.xf_Bg = xf ? B / groups : B;
    const bool vec = vec16;
    hipStream_t s = (hipStream_t)stream;
    // DT (uaps_call_hints::dyt_*): `dy` is d(activation) behind the BatchNorm that follows this convolution; only some kernels can
    // turn it into dy while staging -- every other path returns UAPS_ENOFORM before anything is launched
    const bool dyt = hints.dyt_y != nullptr;
    if (dyt) {
        if (!hints.dyt_coef || !hints.dyt_out || hints.dyt_groups < 1 || hints.dyt_groups > kWrwMaxGroups || B % hints.dyt_groups) return UAPS_EINVAL;
        if (!(hints.dyt_slope >= 0.f && hints.dyt_slope <= 1.f)) return UAPS_ERANGE;
        if (((uintptr_t)hints.dyt_y | (uintptr_t)hints.dyt_out) % 16) return UAPS_ENOFORM;      // the in-staging form streams 16-byte pieces
        a.dt_y = hints.dyt_y; a.dt_coef = hints.dyt_coef; a.dt_out = hints.dyt_out; a.dt_slope = hints.dyt_slope; a.dt_Bg = B / hints.dyt_groups;
    }
    // full-width-row kernels (conv_split_wrw_row.hpp): maps of 256 pixels width or a multiple (256-wide column strips), <= 16 output and 16 / 32 input channels, every operand bounded;
    // they write the slabs of the plan above (split or exact-N), so workspace and reduction do not change
    // (fewer than 16 input channels -- the first layer's 3 -- ride in the 16-channel form: the absent channels are never fetched)
    const bool force_exact = (cfg >> 28) & 1;
    if (ks == 3 && p.dil == 1 && !force_exact && vec16 && W % 256 == 0 && H % 16 == 0 && Cout <= 16 && Cin <= 32 &&
        p.CoutS <= 16 && p.CinS == (Cin > 16 ? 32 : 16) && p.ncob == 1 && p.ncib == 1 && uaps_conv_get_mode() == 2 && hints.bound[0] && hints.bound[1] &&
        (!x2 || Csplit >= Cin || (hints.bound[2] && Csplit % 4 == 0)) &&
        !(uaps_conv_get_tuning() & (UAPS_TUNE_NO_ROW_WRW | UAPS_TUNE_NO_SPLIT_WRW))) {
        a.dy_bound = hints.bound[0]; a.dy_mul = hints.mul[0];
        a.in_bound = hints.bound[1]; a.in_mul = hints.mul[1];
        if (x2 && Csplit < Cin) { a.in2_bound = hints.bound[2]; a.in2_mul = hints.mul[2]; }
        a.err = uaps::error_word();
        const unsigned grid = (unsigned)((p.nsplit + 7) / 8 * 8);
        if (up2) {                                     // up4's first convolution: 16 + 16 input channels, the second 16 from the low-resolution tensor
            if (!(Cin == 32 && Csplit == 16 && W == 256 && uaps::up2_pattern_ok(W / 2))) return UAPS_ENOFORM;
            a.up_rh = (float)(H / 2 - 1) / (float)(H - 1); a.up_rw = (float)(W / 2 - 1) / (float)(W - 1);      // as uaps_up_cat_fwd
            if (dyt) UAPS_LAUNCH_MAIN(conv_hrwrw_up_dt_kernel, dim3(grid), dim3(512), 0, s, a);
            else UAPS_LAUNCH_MAIN(conv_hrwrw_up_kernel, dim3(grid), dim3(512), 0, s, a);
            return (int)hipGetLastError();
        }
        if (dyt && W > 256) {                          // column strips
            if (Cin <= 16) {
                if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrww_bn_dt_kernel<1>), dim3(grid), dim3(256), 0, s, a);
                else UAPS_LAUNCH_MAIN((conv_hrwrww_dt_kernel<1>), dim3(grid), dim3(256), 0, s, a);
            } else {
                if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrww_bn_dt_kernel<2>), dim3(grid), dim3(512), 0, s, a);
                else UAPS_LAUNCH_MAIN((conv_hrwrww_dt_kernel<2>), dim3(grid), dim3(512), 0, s, a);
            }
            return (int)hipGetLastError();
        }
        if (dyt) {
            if (Cin <= 16) {
                if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrw_bn_dt_kernel<1>), dim3(grid), dim3(256), 0, s, a);
                else UAPS_LAUNCH_MAIN((conv_hrwrw_dt_kernel<1>), dim3(grid), dim3(256), 0, s, a);
            } else {
                if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrw_bn_dt_kernel<2>), dim3(grid), dim3(512), 0, s, a);
                else UAPS_LAUNCH_MAIN((conv_hrwrw_dt_kernel<2>), dim3(grid), dim3(512), 0, s, a);
            }
            return (int)hipGetLastError();
        }
        // diagnostic: two rows in flight ahead of the contraction (DEPTH 2 of conv_hrwrw_body; measured slower)
        if ((uaps_conv_get_tuning() & UAPS_TUNE_DEEP_ROWS) && W == 256) {
            if (Cin <= 16) {
                if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrw2_bn_kernel<1>), dim3(grid), dim3(256), 0, s, a);
                else UAPS_LAUNCH_MAIN((conv_hrwrw2_kernel<1>), dim3(grid), dim3(256), 0, s, a);
            } else {
                if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrw2_bn_kernel<2>), dim3(grid), dim3(512), 0, s, a);
                else UAPS_LAUNCH_MAIN((conv_hrwrw2_kernel<2>), dim3(grid), dim3(512), 0, s, a);
            }
            return (int)hipGetLastError();
        }
        if (W > 256) {                                 // 256-wide column strips
            if (Cin <= 16) {
                if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrww_bn_kernel<1>), dim3(grid), dim3(256), 0, s, a);
                else UAPS_LAUNCH_MAIN((conv_hrwrww_kernel<1>), dim3(grid), dim3(256), 0, s, a);
            } else {
                if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrww_bn_kernel<2>), dim3(grid), dim3(512), 0, s, a);
                else UAPS_LAUNCH_MAIN((conv_hrwrww_kernel<2>), dim3(grid), dim3(512), 0, s, a);
            }
        } else if (Cin <= 16) {
            if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrw_bn_kernel<1>), dim3(grid), dim3(256), 0, s, a);
            else UAPS_LAUNCH_MAIN((conv_hrwrw_kernel<1>), dim3(grid), dim3(256), 0, s, a);
        } else {
            if (a.xf) UAPS_LAUNCH_MAIN((conv_hrwrw_bn_kernel<2>), dim3(grid), dim3(512), 0, s, a);
            else UAPS_LAUNCH_MAIN((conv_hrwrw_kernel<2>), dim3(grid), dim3(512), 0, s, a);
        }
        return (int)hipGetLastError();
    }
    // (the tile kernels have no DT form: built and measured late in round 4 -- 32 x 32 channel blocks, bit-identical dy -- the extra
    // staging work cost them more than the stand-alone pass it replaced: 12 launches +314 us against 224 us saved)
    if (dyt || up2) return UAPS_ENOFORM;
    if (p.g1) {
        // single-tensor, 16-byte-aligned form only; workspace and reduce follow the same plan, so the caller chooses: cfg bit 28
        // (exact kernels) for a two-tensor / BatchNorm-in-staging / odd-pointer call of such a layer -- uaps_amd.conv.plan_cfg does
        if (x2 || xf || !vec16) return UAPS_ERANGE;
        a.tiles_x = (H * W) / 32; a.tiles_y = 1;
        const long grid = ((long)a.nsplit * a.ncob * a.ncib + 7) / 8 * 8;
        if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
        if (uaps_conv_get_mode() == 2 && hints.bound[0] && hints.bound[1]) {
            a.dy_bound = hints.bound[0]; a.dy_mul = hints.mul[0];
            a.in_bound = hints.bound[1]; a.in_mul = hints.mul[1];
            a.err = uaps::error_word();
            UAPS_LAUNCH_MAIN(conv_gw1h_kernel, dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        } else {
            UAPS_LAUNCH_MAIN(conv_gw1s_kernel, dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        }
        return (int)hipGetLastError();
    }
    if (p.small) {
        if (x2) return UAPS_EINVAL;
        const unsigned grid = (unsigned)((p.nsplit + 7) / 8 * 8);
        if (a.xf) UAPS_LAUNCH_MAIN(conv_small_wrw_bn_kernel, dim3(grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN(conv_small_wrw_kernel, dim3(grid), dim3(kConvThreads), 0, s, a);
        return (int)hipGetLastError();
    }
    // dilated 3x3 with both operands bounded: the dilated fp16 form (conv_hwrw_d_kernel); same 32 x 32 channel blocks, 4-row
    // tiles and slabs as the fp32 kernel the plan describes, so workspace size and reduction are unchanged
    if (p.dil > 1 && ks == 3 && !xf && !x2 && vec16 && W >= 32 && Cin >= 16 && !((cfg >> 28) & 1) && uaps_conv_get_mode() == 2 &&
        !(uaps_conv_get_tuning() & UAPS_TUNE_NO_SPLIT_WRW) && hints.bound[0] && hints.bound[1] && p.wco == 2 && p.wci == 2 && p.TH == 4) {
        a.dy_bound = hints.bound[0]; a.dy_mul = hints.mul[0];
        a.in_bound = hints.bound[1]; a.in_mul = hints.mul[1];
        a.err = uaps::error_word();
        a.tiles_y = p.dil * (((H + p.dil - 1) / p.dil + 3) / 4);      // per row phase: ceil(ceil(H / dil) / 4) tiles
        const long grid = ((long)a.nsplit * a.ncob * a.ncib + 7) / 8 * 8;
        if (grid <= 0 || grid > 0x7fffffffL || (long)a.B * a.tiles_x * a.tiles_y > 0x7fffffffL) return UAPS_EINVAL;
        if (p.dil == 2) UAPS_LAUNCH_MAIN((conv_hwrw_d_kernel<4, 2, 2, 2>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_hwrw_d_kernel<4, 2, 2, 4>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        return (int)hipGetLastError();
    }
    if (p.split) {
        // mode 2 and every tensor operand bounded (uaps_next_call_hints: 0 = dy, 1 = x, 2 = x2): the two-piece fp16 form
        if (uaps_conv_get_mode() == 2 && hints.bound[0] && hints.bound[1] && (!x2 || Csplit >= Cin || hints.bound[2])) {
            a.dy_bound = hints.bound[0]; a.dy_mul = hints.mul[0];
            a.in_bound = hints.bound[1]; a.in_mul = hints.mul[1];
            if (x2 && Csplit < Cin) { a.in2_bound = hints.bound[2]; a.in2_mul = hints.mul[2]; }
            a.err = uaps::error_word();
        }
        return dispatch_swrw(a, p, s);
    }
    return ks == 3 ? dispatch_wrw<3>(a, p, vec, s) : dispatch_wrw<1>(a, p, vec, s);
}

// (each entry point twice: the legacy form reads the thread's pending uaps_next_call_hints record, the *_h form takes the record as
// its first argument and reads nothing thread-local -- see conv_fwd.hip)
extern "C" int uaps_conv_bwd_weight_partial_h(const uaps_call_hints* hints, const float* dy, const float* x, int want_bias, int B, int Cin,
                                              int Cout, int H, int W, int ks, int cfg, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    return wrw_partial_impl(h, dy, x, nullptr, Cin, want_bias, B, Cin, Cout, H, W, ks, cfg, ws, ws_bytes, stream);
}
extern "C" int uaps_conv_bwd_weight_partial(const float* dy, const float* x, int want_bias, int B, int Cin, int Cout, int H, int W,
                                            int ks, int cfg, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    return wrw_partial_impl(uaps::take_hints(), dy, x, nullptr, Cin, want_bias, B, Cin, Cout, H, W, ks, cfg, ws, ws_bytes, stream);
}

// Weight gradient of a convolution whose input is leaky_relu(batch_norm_train(y)) of a previous conv's raw output y
// (ConvBlock, UAPS_unet.py:38-41): the activation is recomputed while staging from xf [groups][Cin] float2
// (scale, shift) as uaps_bn_finalize_train wrote it.  W % 4 == 0, 16-byte aligned tensors.
extern "C" int uaps_conv_bwd_weight_partial_bn_h(const uaps_call_hints* hints, const float* dy, const float* y, const void* xf, float slope,
                                                 int groups, int want_bias, int B, int Cin, int Cout, int H, int W, int ks, int cfg,
                                                 void* ws, size_t ws_bytes, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    if (!xf) return UAPS_EINVAL;
    return wrw_partial_impl(h, dy, y, nullptr, Cin, want_bias, B, Cin, Cout, H, W, ks, cfg, ws, ws_bytes, stream, xf, slope, groups);
}
extern "C" int uaps_conv_bwd_weight_partial_bn(const float* dy, const float* y, const void* xf, float slope, int groups,
                                               int want_bias, int B, int Cin, int Cout, int H, int W, int ks, int cfg, void* ws,
                                               size_t ws_bytes, uaps_stream_t stream) {
    const uaps_call_hints h = uaps::take_hints();
    if (!xf) return UAPS_EINVAL;
    return wrw_partial_impl(h, dy, y, nullptr, Cin, want_bias, B, Cin, Cout, H, W, ks, cfg, ws, ws_bytes, stream, xf, slope, groups);
}

// Weight gradient of a convolution whose input is the never-materialised concatenation of x1 [B,C1,H,W] and
// x2 [B,C2,H,W] (C1 % 16 == 0); follow with uaps_conv_bwd_weight_reduce(..., Cin = C1 + C2, ...).
extern "C" int uaps_conv_bwd_weight_partial_cat_h(const uaps_call_hints* hints, const float* dy, const float* x1, int C1, const float* x2, int C2,
                                                  int want_bias, int B, int Cout, int H, int W, int ks, int cfg, void* ws, size_t ws_bytes,
                                                  uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    if (!x2 || C1 <= 0 || C2 <= 0) return UAPS_EINVAL;
    return wrw_partial_impl(h, dy, x1, x2, C1, want_bias, B, C1 + C2, Cout, H, W, ks, cfg, ws, ws_bytes, stream);
}
extern "C" int uaps_conv_bwd_weight_partial_cat(const float* dy, const float* x1, int C1, const float* x2, int C2, int want_bias,
                                                int B, int Cout, int H, int W, int ks, int cfg, void* ws, size_t ws_bytes,
                                                uaps_stream_t stream) {
    const uaps_call_hints h = uaps::take_hints();
    if (!x2 || C1 <= 0 || C2 <= 0) return UAPS_EINVAL;
    return wrw_partial_impl(h, dy, x1, x2, C1, want_bias, B, C1 + C2, Cout, H, W, ks, cfg, ws, ws_bytes, stream);
}

// Step 2: fixed-order sum of the partials into dw (and dbias).
extern "C" int uaps_conv_bwd_weight_reduce(const void* ws, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W, int ks,
                                           int cfg, uaps_stream_t stream) {
    if (!ws || !dw || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (ks != 1 && ks != 3)) return UAPS_EINVAL;
    const int taps = ks * ks;
    const WrwPlan p = plan_wrw(B, Cin, Cout, H, W, cfg, ks);
    const float* slab = (const float*)ws;
    const float* bslab = dbias ? slab + (size_t)p.nsplit * taps * p.CoutS * p.CinS : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)taps * p.CoutS * p.CinS + (dbias ? p.CoutS : 0);
    if (n < 32768)
        hipLaunchKernelGGL(conv_wrw_reduce_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, slab, bslab, dw, dbias,
                           p.nsplit, taps, Cout, Cin, p.CoutS, p.CinS);
    else
        hipLaunchKernelGGL(conv_wrw_reduce_kernel<64>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, slab, bslab, dw, dbias,
                           p.nsplit, taps, Cout, Cin, p.CoutS, p.CinS);
    return (int)hipGetLastError();
}

// Step 2 for several weight gradients at once: item i is reduced exactly as uaps_conv_bwd_weight_reduce(items[i]...) would
// (bit-identical), by one launch per kReduceBatch items.  The workspaces must be distinct and stay untouched until this call.
extern "C" int uaps_conv_bwd_weight_reduce_batch(const uaps_wrw_reduce_item* items, int n, uaps_stream_t stream) {
    if (n < 0 || (n > 0 && !items)) return UAPS_EINVAL;
    for (int i = 0; i < n; ++i) {
        const uaps_wrw_reduce_item& it = items[i];
        if (!it.workspace || !it.dw || it.B <= 0 || it.Cin <= 0 || it.Cout <= 0 || it.H <= 0 || it.W <= 0 || (it.ks != 1 && it.ks != 3)) return UAPS_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    for (int i0 = 0; i0 < n; i0 += kReduceBatch) {
        ReduceBatch rb{};
        unsigned blocks = 0;
        const int m = n - i0 < kReduceBatch ? n - i0 : kReduceBatch;
        for (int j = 0; j < m; ++j) {
            const uaps_wrw_reduce_item& it = items[i0 + j];
            const int taps = it.ks * it.ks;
            const WrwPlan p = plan_wrw(it.B, it.Cin, it.Cout, it.H, it.W, it.cfg, it.ks);
            ReduceDesc& q = rb.d[j];
            q.slab = (const float*)it.workspace;
            q.bslab = it.dbias ? q.slab + (size_t)p.nsplit * taps * p.CoutS * p.CinS : nullptr;
            q.dw = it.dw; q.db = it.dbias;
            q.nsplit = p.nsplit; q.taps = taps; q.Cout = it.Cout; q.Cin = it.Cin; q.CoutS = p.CoutS; q.CinS = p.CinS;
            const long ne = (long)taps * p.CoutS * p.CinS + (it.dbias ? p.CoutS : 0);
            q.el = ne < 32768 ? 16 : 64;
            q.first = blocks;
            long nb = (ne + q.el - 1) / q.el;
            const long nw = (long)taps * p.CoutS * p.CinS;
            if (q.el == 64 && nw % 4 == 0 && (reinterpret_cast<uintptr_t>(q.slab) & 15) == 0) {      // the 16-byte form of conv_wrw_reduce_batch_kernel
                q.el = 256;
                nb = (nw + 255) / 256 + (it.dbias ? (p.CoutS + 63) / 64 : 0);
            }
            if (nb + blocks > 0x7fffffffL) return UAPS_ERANGE;
            blocks += (unsigned)nb;
        }
        rb.n = m; rb.blocks = blocks;
        hipLaunchKernelGGL(conv_wrw_reduce_batch_kernel, dim3(blocks), dim3(256), 0, s, rb);
    }
    return (int)hipGetLastError();
}

extern "C" int uaps_conv_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int B, int Cin, int Cout, int H,
                                    int W, int ks, int cfg, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    if (!dw) return UAPS_EINVAL;
    const int rc = uaps_conv_bwd_weight_partial(dy, x, dbias != nullptr, B, Cin, Cout, H, W, ks, cfg, ws, ws_bytes, stream);
    if (rc) return rc;
    return uaps_conv_bwd_weight_reduce(ws, dw, dbias, B, Cin, Cout, H, W, ks, cfg, stream);
}

extern "C" int uaps_conv_wrw_variant(int B, int Cin, int Cout, int H, int W, int ks, int cfg, char* buf, size_t buflen) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (ks != 1 && ks != 3) || !buf || buflen < 64) return UAPS_EINVAL;
    const WrwPlan p = plan_wrw(B, Cin, Cout, H, W, cfg, ks);
    if (p.small) snprintf(buf, buflen, "conv_small_wrw_kernel");
    else if (p.g1) snprintf(buf, buflen, "conv_gw1s_kernel");
    else if (p.split) snprintf(buf, buflen, "conv_swrw_kernel<%d, %d, %d>", p.TH, p.wco, p.wci);
    else snprintf(buf, buflen, "conv_wrw_kernel<%d, %d, %d, %d, %d, %d, %d>", ks, p.TH, p.TW, p.wco, p.wci, (W % 4 == 0) ? 4 : 1, p.dil);
    return UAPS_OK;
}
