// extern "C" entry points of the forward / input-gradient convolution and the weight packing
// (include/uaps_hip.h, section "Convolutions").
#include "../../include/uaps_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include "conv_kernels.hpp"
#include "conv_split.hpp"
#include "conv_split_n16.hpp"
#include "conv_split_row16.hpp"
#include "conv_small.hpp"
#include "conv_gemm1x1.hpp"
#include "conv_split_g.hpp"
#include "hints.hpp"
using namespace uaps;

// 0 = exact fp32 matrix instructions (v_mfma_f32_16x16x4_f32), 1 = exact 3-way bf16 split on the bf16 matrix pipe,
// 2 (default) = 1, and the two-piece fp16 split for operands with a known magnitude bound (conv_split.hpp).
// Set with uaps_conv_set_mode (the Python layer forwards UAPS_CONV_MODE once at load); the library itself reads no environment.
static int g_conv_mode = 2;
static unsigned g_conv_tuning = 0;      // UAPS_TUNE_* bits (uaps_conv_set_tuning): ablation / diagnosis switches of the planners
static int conv_mode() { return g_conv_mode; }
extern "C" int uaps_conv_set_mode(int mode) {
    if (mode != 0 && mode != 1 && mode != 2) return UAPS_EINVAL;
    g_conv_mode = mode;
    return UAPS_OK;
}
extern "C" int uaps_conv_get_mode(void) { return conv_mode(); }
extern "C" int uaps_conv_set_tuning(unsigned flags) { g_conv_tuning = flags; return UAPS_OK; }
extern "C" unsigned uaps_conv_get_tuning(void) { return g_conv_tuning; }

namespace {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int kdim_pad(int c, int ks) { return (ks == 3 && c <= 4) ? 4 : round_up(c, 8); }   // K (contraction) channels: chunk of 4 or 8
inline int ndim_pad(int c) { return round_up(c, 16); }              // N (output) channels: 16-wide MFMA tiles

template <int KS, int TH, int TW, int BN, int CK, int DIL = 1>
int launch_fwd(ConvFwdArgs a, bool vec, int extra_lds, hipStream_t s) {
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    a.nblk = a.CoutP / BN;
    const long grid = ((long)a.B * a.tiles_x * a.tiles_y * a.nblk + 7) / 8 * 8;   // multiple of 8 for the XCD swizzle
    if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
    // extra_lds: unused dynamic LDS, requested only to cap the number of co-resident workgroups per CU
    if (a.xf) {                  // BatchNorm + LeakyReLU of the input applied while staging: 16-byte form, 8-channel chunks, no dilation
        if constexpr (DIL == 1 && CK == 8) {
            if (!vec) return UAPS_ERANGE;
            UAPS_LAUNCH_MAIN((conv_fwd_bn_kernel<KS, TH, TW, BN, CK, 4, 1>), dim3((unsigned)grid), dim3(kConvThreads), extra_lds, s, a);
            return (int)hipGetLastError();
        } else {
            return UAPS_ERANGE;
        }
    }
    if (vec) UAPS_LAUNCH_MAIN((conv_fwd_kernel<KS, TH, TW, BN, CK, 4, DIL>), dim3((unsigned)grid), dim3(kConvThreads), extra_lds, s, a);
    else UAPS_LAUNCH_MAIN((conv_fwd_kernel<KS, TH, TW, BN, CK, 1, DIL>), dim3((unsigned)grid), dim3(kConvThreads), extra_lds, s, a);
    return (int)hipGetLastError();
}

template <int KS, int TH, int TW>
int dispatch_bn_ck(const ConvFwdArgs& a, int bn, int ck, bool vec, int xl, hipStream_t s) {
    if constexpr (KS == 3) {
        if (ck == 4) return launch_fwd<KS, TH, TW, 16, 4>(a, vec, xl, s);
    }
    if (bn == 16) return launch_fwd<KS, TH, TW, 16, 8>(a, vec, xl, s);
    if (bn == 32) return launch_fwd<KS, TH, TW, 32, 8>(a, vec, xl, s);
    return launch_fwd<KS, TH, TW, 64, 8>(a, vec, xl, s);
}

// dilated 3x3 (ResNet stages with stride replaced by dilation): 8x32 tiles, 8-channel chunks
template <int DIL>
int dispatch_dilated(const ConvFwdArgs& a, int bn, bool vec, int xl, hipStream_t s) {
    if (bn == 16) return launch_fwd<3, 8, 32, 16, 8, DIL>(a, vec, xl, s);
    if (bn == 32) return launch_fwd<3, 8, 32, 32, 8, DIL>(a, vec, xl, s);
    return launch_fwd<3, 8, 32, 64, 8, DIL>(a, vec, xl, s);
}

// ---- split-bf16 kernels (conv_split.hpp): 16-byte rows only, no dilation ----
template <int KS, int TH, int TW, int BN, int CK>
int launch_sfwd(ConvFwdArgs a, hipStream_t s) {
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    a.nblk = a.CoutP / BN;
    const long grid = ((long)a.B * a.tiles_x * a.tiles_y * a.nblk + 7) / 8 * 8;
    if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
    if (a.wscale) {                                   // the fp16 two-piece form (plan: h16)
        if (a.xf) UAPS_LAUNCH_MAIN((conv_hfwd_bn_kernel<KS, TH, TW, BN, CK>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_hfwd_kernel<KS, TH, TW, BN, CK>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        return (int)hipGetLastError();
    }
    if (a.xf) UAPS_LAUNCH_MAIN((conv_sfwd_bn_kernel<KS, TH, TW, BN, CK>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    else UAPS_LAUNCH_MAIN((conv_sfwd_kernel<KS, TH, TW, BN, CK>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    return (int)hipGetLastError();
}
template <int KS, int TH, int TW>
int dispatch_sfwd(const ConvFwdArgs& a, int bn, int ck, hipStream_t s) {
    if constexpr (KS == 3) {
        if (ck == 8) return bn == 16 ? launch_sfwd<3, TH, TW, 16, 8>(a, s) : launch_sfwd<3, TH, TW, 32, 8>(a, s);
        return bn == 16 ? launch_sfwd<3, TH, TW, 16, 16>(a, s) : launch_sfwd<3, TH, TW, 32, 16>(a, s);
    } else {
        return bn == 16 ? launch_sfwd<1, TH, TW, 16, 32>(a, s) : launch_sfwd<1, TH, TW, 32, 32>(a, s);
    }
}

// persistent 16-output-channel kernels of conv_split_n16.hpp (fp16 form only): each workgroup takes a run of tiles
int launch_hp16(ConvFwdArgs a, hipStream_t s) {
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 7) / 8;
    a.nblk = 1;
    const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffL) return UAPS_EINVAL;
    // one resident round: 3 (16 input channels) or 2 (32) workgroups fit a CU (registers), 256 CUs; forcing 4 spills and is slower
    const long want = a.Cin <= 16 ? 768 : 512;
    const unsigned grid = (unsigned)(((ntiles < want ? ntiles : want) + 7) / 8 * 8);
    if (a.Cin <= 16) {
        if (a.xf) UAPS_LAUNCH_MAIN((conv_hp16_bn_kernel<2>), dim3(grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_hp16_kernel<2>), dim3(grid), dim3(kConvThreads), 0, s, a);
    } else {
        if (a.xf) UAPS_LAUNCH_MAIN((conv_hp16_bn_kernel<4>), dim3(grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_hp16_kernel<4>), dim3(grid), dim3(kConvThreads), 0, s, a);
    }
    return (int)hipGetLastError();
}

// full-width-row kernels of conv_split_row16.hpp (fp16 form, maps of 256 pixels width -- or a multiple: 256-wide column strips --, H % 16 == 0): runs of 16 rows, 2 workgroups of
// 256 threads per CU for 16 input channels, 1 workgroup of 512 threads (135 KB of LDS) for 32
int launch_hr16(ConvFwdArgs a, hipStream_t s) {
    const long nruns = (long)a.B * (a.H / 16) * (a.W / 256);      // a run: 16 rows of one 256-wide column strip
    if (nruns <= 0 || nruns > 0x7fffffffL) return UAPS_EINVAL;
    const long want = a.Cin <= 16 ? 512 : 256;
    const unsigned grid = (unsigned)(((nruns < want ? nruns : want) + 7) / 8 * 8);
    const bool strips = a.W > 256;
    if (a.Cin <= 16) {
        if (strips) {
            if (a.xf) UAPS_LAUNCH_MAIN((conv_hr16w_bn_kernel<2>), dim3(grid), dim3(256), 0, s, a);
            else UAPS_LAUNCH_MAIN((conv_hr16w_kernel<2>), dim3(grid), dim3(256), 0, s, a);
        } else {
            if (a.xf) UAPS_LAUNCH_MAIN((conv_hr16_bn_kernel<2>), dim3(grid), dim3(256), 0, s, a);
            else UAPS_LAUNCH_MAIN((conv_hr16_kernel<2>), dim3(grid), dim3(256), 0, s, a);
        }
    } else {
        if (strips) {
            if (a.xf) UAPS_LAUNCH_MAIN((conv_hr16w_bn_kernel<4>), dim3(grid), dim3(512), 0, s, a);
            else UAPS_LAUNCH_MAIN((conv_hr16w_kernel<4>), dim3(grid), dim3(512), 0, s, a);
        } else {
            if (a.xf) UAPS_LAUNCH_MAIN((conv_hr16_bn_kernel<4>), dim3(grid), dim3(512), 0, s, a);
            else UAPS_LAUNCH_MAIN((conv_hr16_kernel<4>), dim3(grid), dim3(512), 0, s, a);
        }
    }
    return (int)hipGetLastError();
}

// 16 + 16 -> 16 channels, the second 16 up-sampled x2 from a low-resolution tensor while staging (conv_hr16_up_kernel; W == 256)
int launch_hr16_up(ConvFwdArgs a, hipStream_t s) {
    const long nruns = (long)a.B * (a.H / 16);
    if (nruns <= 0 || nruns > 0x7fffffffL) return UAPS_EINVAL;
    const int h = a.H / 2, w = a.W / 2;
    a.up_rh = (float)(h - 1) / (float)(a.H - 1); a.up_rw = (float)(w - 1) / (float)(a.W - 1);      // as uaps_up_cat_fwd
    const unsigned grid = (unsigned)(((nruns < 256 ? nruns : 256) + 7) / 8 * 8);
    UAPS_LAUNCH_MAIN(conv_hr16_up_kernel, dim3(grid), dim3(512), 0, s, a);
    return (int)hipGetLastError();
}

// 16 -> 16 channels, 256 wide: the input gradient that also forms the BatchNorm-backward sums of the layer in front (conv_hr16_bs_kernel)
int launch_hr16_bs(ConvFwdArgs a, hipStream_t s) {
    const long nruns = (long)a.B * (a.H / 16);
    if (nruns <= 0 || nruns > 0x7fffffffL) return UAPS_EINVAL;
    const unsigned grid = (unsigned)(((nruns < 512 ? nruns : 512) + 7) / 8 * 8);
    UAPS_LAUNCH_MAIN(conv_hr16_bs_kernel, dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int launch_hr16x2(ConvFwdArgs a, hipStream_t s) {
    const long nruns = (long)a.B * (a.H / 16) * (a.W / 256);
    if (nruns <= 0 || nruns > 0x7fffffffL) return UAPS_EINVAL;
    const unsigned grid = (unsigned)(((nruns < 512 ? nruns : 512) + 7) / 8 * 8);
    if (a.W > 256) UAPS_LAUNCH_MAIN(conv_hr16wx2_kernel, dim3(grid), dim3(256), 0, s, a);
    else UAPS_LAUNCH_MAIN(conv_hr16x2_kernel, dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

int launch_g1(ConvFwdArgs a, int bn, hipStream_t s) {
    a.tiles_x = (a.H * a.W + 127) / 128;
    a.tiles_y = 1;
    a.nblk = a.CoutP / bn;
    const long grid = ((long)a.B * a.tiles_x * a.nblk + 7) / 8 * 8;
    if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
    if (a.wscale) {
        if (bn == 256) UAPS_LAUNCH_MAIN(conv_g1h256_kernel, dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else if (bn == 128) UAPS_LAUNCH_MAIN((conv_g1h_kernel<128>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_g1h_kernel<64>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    } else {
        if (bn == 128) UAPS_LAUNCH_MAIN((conv_g1s_kernel<128>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_g1s_kernel<64>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    }
    return (int)hipGetLastError();
}

int launch_small(ConvFwdArgs a, int kind, hipStream_t s) {
    a.tiles_x = (a.W + 63) / 64;
    a.tiles_y = (a.H + 15) / 16;
    a.nblk = 1;
    const long grid = ((long)a.B * a.tiles_x * a.tiles_y + 7) / 8 * 8;
    if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
    (void)kind;
    if (a.xf) UAPS_LAUNCH_MAIN((conv_small_bn_kernel<8, 4>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    else UAPS_LAUNCH_MAIN((conv_small_kernel<8, 4>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    return (int)hipGetLastError();
}

template <int BN>
int launch_s32(ConvFwdArgs a, hipStream_t s) {
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 7) / 8;
    a.nblk = a.CoutP / BN;
    const long grid = ((long)a.B * a.tiles_x * a.tiles_y * a.nblk + 7) / 8 * 8;
    if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
    if (a.wscale) {                                   // the fp16 two-piece form (plan: h16)
        if (a.xf) UAPS_LAUNCH_MAIN((conv_h32_bn_kernel<BN>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_h32_kernel<BN>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        return (int)hipGetLastError();
    }
    if (a.xf) UAPS_LAUNCH_MAIN((conv_s32_bn_kernel<BN>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    else UAPS_LAUNCH_MAIN((conv_s32_kernel<BN>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    return (int)hipGetLastError();
}

// 16-row tiles, 32 output channels per workgroup (conv_s32t/h32t kernels)
int launch_s32t(ConvFwdArgs a, hipStream_t s) {
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 15) / 16;
    a.nblk = a.CoutP / 32;
    const long grid = ((long)a.B * a.tiles_x * a.tiles_y * a.nblk + 7) / 8 * 8;
    if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
    if (a.wscale) {
        if (a.xf) UAPS_LAUNCH_MAIN((conv_h32t_bn_kernel<32>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_h32t_kernel<32>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        return (int)hipGetLastError();
    }
    if (a.xf) UAPS_LAUNCH_MAIN((conv_s32t_bn_kernel<32>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    else UAPS_LAUNCH_MAIN((conv_s32t_kernel<32>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    return (int)hipGetLastError();
}

// split: one of the bf16-split kernels runs; s32: the 32x32x16 form (3x3, 8x32 tiles, >= 32 output channels)
// small: 1 = the exact-N VALU kernel for <= 4 output channels (conv_small.hpp)
// g1: the GEMM-tiled 1x1 kernels of conv_gemm1x1.hpp (128 consecutive pixels x 128 / 64 output channels per workgroup)
// s32t: 16-row tiles for the 32-channel blocks (the statistics parts stay per 8 rows)
struct FwdPlan { int ck, bn, th, tw; bool vec; int CinP, CoutP, extra_lds, dil; bool split, s32; int sck, sbn; int small; bool g1, s32t;
                 bool bounded, g128; };

int plan_fwd(const void* x, const void* y, int B, int Cin, int Cout, int H, int W, int ks, int cfg, FwdPlan* p) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (ks != 1 && ks != 3) return UAPS_ERANGE;
    if ((double)Cin * H * W * 4.0 >= 2147483648.0 || (double)Cout * H * W * 4.0 >= 2147483648.0) return UAPS_ERANGE;
    p->bounded = (cfg & UAPS_CONV_BOUNDED) != 0;      // the caller's promise of operand bounds (include/uaps_hip.h); not a planner setting
    cfg &= ~UAPS_CONV_BOUNDED;
    p->CinP = kdim_pad(Cin, ks); p->CoutP = ndim_pad(Cout);
    p->ck = (ks == 3 && Cin <= 4 && ((cfg >> 24) & 0xf) <= 1) ? 4 : 8;
    // 16-byte loads/stores need rows that start 16-byte aligned
    p->vec = (W % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    // pixel tile: 8 rows x 32 columns, or 16 x 16 for narrow maps (both 256 pixels = 16 M tiles)
    const bool wide = ((cfg >> 24) & 0xf) > 1 ? true : (((cfg >> 8) & 3) ? (((cfg >> 8) & 3) == 1) : (W >= 32));
    p->extra_lds = ((cfg >> 16) & 0xff) * 1024;
    p->dil = ((cfg >> 24) & 0xf) ? ((cfg >> 24) & 0xf) : 1;
    if (p->dil != 1 && (ks != 3 || (p->dil != 2 && p->dil != 4) || Cin <= 4)) return UAPS_ERANGE;
    p->th = wide ? 8 : 16; p->tw = wide ? 32 : 16;
    int bn = (p->CoutP % 64 == 0) ? 64 : (p->CoutP % 32 == 0 ? 32 : 16);
    const long tiles = (long)B * ((H + p->th - 1) / p->th) * ((W + p->tw - 1) / p->tw);
    while (bn > 16 && tiles * (p->CoutP / bn) < 512) bn /= 2;       // at least two workgroups per CU
    if (cfg & 0xff) { bn = cfg & 0xff; if ((bn != 16 && bn != 32 && bn != 64) || p->CoutP % bn) return UAPS_EINVAL; }
    if (p->ck == 4) bn = 16;
    p->bn = bn;
    p->small = 0;
    if (ks == 3 && p->dil == 1 && W % 4 == 0 && W >= 64 && H >= 8 && !(cfg & 0x70ffffff)) {
        if (Cout <= 4 && Cin % 8 == 0 && Cin <= 32) p->small = 1;      // conv_small_body: KMAX
        // (the mirror case, <= 4 contraction channels -> 16 outputs, measured 52 us against 47 us of the fp32 MFMA kernel at
        // 4 -> 16 @ 256 x 256, B = 32: not used)
    }
    // the split-bf16 form: 16-byte rows, no dilation, no forced tile / LDS settings; cfg bit 28 forces the exact kernels
    // cfg bits 29-30 select among them for tools/bench_modes.py: 1 = the 16x16x32 form, 2 = the 32x32x16 form (low byte: BN)
    // measured (tools/bench_modes.py, B = 32): the split forms win 1.2-1.7x on every 3x3 layer with more than 8 contraction
    // channels and lose on 1x1 convolutions (HBM-bound, the split only adds staging work) and on the <= 8-channel layers
    // (a chunk of 8 channels fills 9 of 12 k-groups); cfg bits 29-30 != 0 force a split form regardless
    // 1x1 convolutions: narrow ones, and wide ones on maps of fewer than 32 x 32 pixels (the U-Net's 256 -> 128 projection at a
    // 256^2 input), are HBM- or launch-bound and stay on the fp32 instruction; from 1024 pixels on (the same projection at a 512^2
    // input, and the bottleneck projections of the ResNet encoders: 64 ... 2048 channels, utilities/resnet.py:55-95) they are
    // compute-bound and take the split form.  plan_wrw (conv_wrw.hip) applies the same rule.
    const bool big_1x1 = ks == 1 && Cin >= 64 && (long)Cin * Cout >= 16384 && (long)H * W >= 1024;
    // dilated 3x3 (ResNet stages): the 32x32x16 split form only (>= 32 output channels, no forced settings)
    const bool dil_ok = p->dil == 1 || (ks == 3 && Cin > 8 && p->CoutP % 32 == 0 && !(cfg & 0x70ffffff));
    p->split = conv_mode() >= 1 && dil_ok && (W % 4 == 0) && !(cfg & 0x10ffff00) && ((ks == 3 && Cin > 8) || big_1x1 || ((cfg >> 29) & 3));
    if (g_conv_tuning & UAPS_TUNE_NO_SPLIT_FWD) p->split = false;
    p->g1 = p->split && big_1x1 && p->vec && p->CoutP % 64 == 0 && !(cfg & 0x7fffffff);
    const int sel = (cfg >> 29) & 3, bn_req = cfg & 0xff;
    p->sck = ks == 1 ? 32 : (Cin <= 8 ? 8 : 16);
    // 32 output channels per workgroup from 256 workgroups on (round 5, profiles/r05_mfma_shape_ab.txt: 256 -> 256 @ 16 x 16, B = 32:
    // 42.4 against 46.1 us with 16, 128 -> 256: 23.6 against 25.3; with 128 workgroups -- 256 -> 128 -- 16 stays faster, 34.0 against 40.4)
    p->sbn = (p->CoutP % 32 == 0 && tiles * (p->CoutP / 32) >= 256) ? 32 : 16;
    p->s32 = p->split && ks == 3 && p->tw == 32 && p->CoutP % 32 == 0 && sel != 1;
    // round 5 (profiles/r05_mfma_shape_ab.txt, same per-wave tile of 64 pixels x 32 channels): on maps of <= 32 x 32 pixels with >= 128
    // contraction channels and 128 output channels the 16x16x32 form with 16-channel chunks is 3-5 % faster than either 32x32x16 form
    // (128 -> 128: 33.1 against 34.3 us, 256 -> 128: 59.2 against 62.4 us at B = 32); everywhere else it loses 4-15 %
    // (measured in the fp16-split arithmetic only: mode 2)
    if (p->s32 && conv_mode() == 2 && sel == 0 && !bn_req && p->dil == 1 && (long)H * W <= 1024 && Cin >= 128 && p->CoutP == 128 &&
        tiles * (p->CoutP / 32) >= 512) { p->s32 = false; p->sbn = 32; p->sck = 16; }
    if (p->s32) {
        p->sck = 8;
        p->sbn = (p->CoutP % 64 == 0 && tiles * (p->CoutP / 64) >= 512) ? 64 : 32;
        if (bn_req == 32 || (bn_req == 64 && p->CoutP % 64 == 0)) p->sbn = bn_req;
        // 32-channel blocks: 16-row tiles while that still gives every CU two workgroups (UAPS_TUNE_NO_TALL_FWD keeps the 8-row tiles)
        const long tiles16 = (long)B * ((H + 15) / 16) * ((W + 31) / 32);
        p->s32t = p->sbn == 32 && p->dil == 1 && !(g_conv_tuning & UAPS_TUNE_NO_TALL_FWD) &&
                  tiles16 * (p->CoutP / 32) >= 512;
    } else if (p->split && bn_req) {
        if ((bn_req != 16 && bn_req != 32) || p->CoutP % bn_req) return UAPS_EINVAL;
        p->sbn = bn_req;
    }
    // round 6 (csrc/conv_split_g.hpp): 128-output-channel layers on 32-wide maps as ONE workgroup per 4-row band and 128 channels --
    // every input element staged once per layer, 32-channel chunks, weight fragments straight from L2; fp16-split arithmetic only
    // (exactly 128 output channels: with 256 -- the input gradient of up1's first convolution -- the form needs two rounds of one
    // workgroup per CU and stages the input twice: 100 us against 63.5 us of conv_h32_kernel<64>, profiles/r06_hg128_ab.txt)
    p->g128 = p->split && conv_mode() == 2 && ks == 3 && p->dil == 1 && W == 32 && H % 4 == 0 && Cin % 32 == 0 && Cout == 128 && p->vec &&
              !(cfg & 0x7fffffff) && !(g_conv_tuning & UAPS_TUNE_NO_G);
    return UAPS_OK;
}

int launch_hg128(ConvFwdArgs a, hipStream_t s) {
    a.tiles_x = 1;
    a.tiles_y = a.H / 4;
    a.nblk = a.Cout / 128;
    const long grid = ((long)a.B * a.tiles_y * a.nblk + 7) / 8 * 8;
    if (grid <= 0 || grid > 0x7fffffffL) return UAPS_EINVAL;
    static bool attr = false;
    if (!attr) {                                      // (idempotent; > 48 KB of dynamic LDS needs the attribute once per kernel)
        (void)hipFuncSetAttribute((const void*)conv_hg128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kHg128Lds);
        (void)hipFuncSetAttribute((const void*)conv_hg128_bn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kHg128Lds);
        (void)hipFuncSetAttribute((const void*)conv_hg128_deep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kHg128Lds);
        attr = true;
    }
    if (a.xf) UAPS_LAUNCH_MAIN(conv_hg128_bn_kernel, dim3((unsigned)grid), dim3(kHg128Threads), kHg128Lds, s, a);
    else if (g_conv_tuning & UAPS_TUNE_G_DEEP) UAPS_LAUNCH_MAIN(conv_hg128_deep_kernel, dim3((unsigned)grid), dim3(kHg128Threads), kHg128Lds, s, a);
    else UAPS_LAUNCH_MAIN(conv_hg128_kernel, dim3((unsigned)grid), dim3(kHg128Threads), kHg128Lds, s, a);
    return (int)hipGetLastError();
}

template <int BN>
int launch_s32d(ConvFwdArgs a, int dil, hipStream_t s) {
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 7) / 8;
    a.nblk = a.CoutP / BN;
    const long grid = ((long)a.B * a.tiles_x * a.tiles_y * a.nblk + 7) / 8 * 8;
    if (grid <= 0 || grid > 0x7fffffffL || (dil != 2 && dil != 4)) return UAPS_EINVAL;
    if (a.wscale) {
        if (dil == 2) UAPS_LAUNCH_MAIN((conv_h32d_kernel<BN, 2>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_h32d_kernel<BN, 4>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    } else {
        if (dil == 2) UAPS_LAUNCH_MAIN((conv_s32d_kernel<BN, 2>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
        else UAPS_LAUNCH_MAIN((conv_s32d_kernel<BN, 4>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    }
    return (int)hipGetLastError();
}

// x [B,Cin,H,W] * packed weights [taps][CinP][CoutP] -> y [B,Cout,H,W]
// x2 / Csplit: optional second input tensor holding channels [Csplit, Cin); y2 / Osplit likewise for the output
int conv_fwd_any(const uaps_call_hints& hints, const float* x, const float* wp, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int ks,
                 int cfg, hipStream_t s, float2* stats = nullptr, const float* x2 = nullptr, int Csplit = -1,
                 float* y2 = nullptr, int Osplit = -1, const void* xf = nullptr, float xf_slope = 0.f, int groups = 1) {
    if (!x || !wp || !y) return UAPS_EINVAL;
    // UAPS_CONV_X2_UP2: x2 is [B, Cin - Csplit, H / 2, W / 2] and is up-sampled x2 (bilinear, align_corners) while staged
    const bool up2 = (cfg & UAPS_CONV_X2_UP2) != 0;
    cfg &= ~UAPS_CONV_X2_UP2;
    if (up2 && (!x2 || H % 2 || W % 2)) return UAPS_EINVAL;
    if (xf && (x2 || groups < 1 || B % groups || (uintptr_t)xf % 8)) return UAPS_EINVAL;
    if (xf && !(xf_slope >= 0.f && xf_slope <= 1.f)) return UAPS_ERANGE;      // leaky_relu is evaluated as max(z, slope * z)
    if (Csplit < 0 || !x2) Csplit = Cin;
    if (Osplit < 0 || !y2) Osplit = Cout;
    if (Csplit > Cin || Osplit > Cout || (Csplit < Cin && (Csplit % 8 || Csplit == 0)) || (Osplit < Cout && Osplit == 0)) return UAPS_EINVAL;
    FwdPlan p{};
    const int rc = plan_fwd(x, y, B, Cin, Cout, H, W, ks, cfg, &p);
    if (rc) return rc;
    if ((x2 && !up2 && (uintptr_t)x2 % 16) || (y2 && (uintptr_t)y2 % 16)) p.vec = false;
    if (!p.vec || (Csplit < Cin && Csplit % p.sck)) p.split = false;      // unaligned tensors / odd concat split: exact kernels
    if (p.dil > 1 && (!p.s32 || x2 || y2 || xf)) p.split = false;        // the dilated split form: one tensor per side, no staging BatchNorm
    // uaps_conv_fwd_stats_parts reported the GEMM-tiled plan's parts (shape alone decides it); a call that cannot run that plan
    // must not write another layout.  The Python layer never gets here: conv.plan_cfg hands such calls cfg bit 28.
    if (stats) {
        FwdPlan p0{};
        if (plan_fwd(nullptr, nullptr, B, Cin, Cout, H, W, ks, cfg, &p0)) return UAPS_EINVAL;
        if (p0.g1 != (p.g1 && p.split && !x2 && !y2 && !xf)) return UAPS_ERANGE;
    }
    uaps::account_bytes(4.0 * B * H * W * ((double)Csplit + (Cin - Csplit) * (up2 ? 0.25 : 1.0) + Cout));      // every input and output element once
    ConvFwdArgs a{};
    a.in = x; a.in2 = x2; a.Csplit = Csplit; a.out2 = y2; a.Osplit = Osplit;
    a.wp = wp; a.bias = bias; a.out = y; a.stats = stats;
    if (stats) { a.stats_mean = hints.stats_mean; a.stats_bias = hints.stats_bias; } a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.CinP = p.CinP; a.CoutP = p.CoutP;
    a.xf = (const float2*)xf; a.xf_slope = xf_slope; a.xf_Bg = xf ? B / groups : B;
    // out_amax: raised to max|y| by the fp32-instruction kernels only (the U-Net's 1x1 projections, whose output has no BatchNorm
    // behind it to bound it); a plan that runs another kernel has no such form
    if (hints.out_amax) {
        if (p.split || p.small || p.g1 || (uintptr_t)hints.out_amax % 16) return UAPS_ENOFORM;
        a.amax = hints.out_amax;
    }
    const bool wide = p.tw == 32;
    // a side with <= 4 channels: the exact-N fp32 kernels (no BatchNorm statistics epilogue, one tensor per side)
    const bool no_small = (g_conv_tuning & UAPS_TUNE_NO_SMALL) != 0;
    // (on 256-wide maps with bounded operands the full-width-row kernel serves <= 4 output channels too: padded to one 16-channel
    // MFMA tile it streams at the HBM rate of a 16 -> 16 layer, which the 16 x 64-tile VALU kernel does not reach)
    const bool row16 = p.split && conv_mode() == 2 && hints.bound[0] && ks == 3 && p.dil == 1 && W % 256 == 0 && H % 16 == 0 && p.CoutP == 16 &&
                       Cin > 8 && Cin <= 32 && Cin % 8 == 0 && !y2 && (!x2 || Csplit >= Cin || hints.bound[1]) &&
                       !(g_conv_tuning & (UAPS_TUNE_NO_ROW16 | UAPS_TUNE_NO_HP16));
    // BatchNorm-backward sums in the epilogue (uaps_call_hints::bsum_*): ONE kernel, the 16 -> 16 input gradient on a 256-wide map
    const bool bsum = hints.bsum_y != nullptr;
    if (bsum) {
        if (!hints.bsum_mean || !hints.bsum_invstd || !hints.bsum_gamma || !hints.bsum_beta || !hints.bsum_partials || !hints.bsum_max ||
            hints.bsum_groups < 1 || B % hints.bsum_groups || !(hints.bsum_slope >= 0.f && hints.bsum_slope <= 1.f)) return UAPS_EINVAL;
        if (!(row16 && !x2 && !xf && !stats && !up2 && Cin == 16 && Cout == 16 && W == 256 && p.vec && (uintptr_t)hints.bsum_y % 16 == 0 &&
              (uintptr_t)hints.bsum_max % 16 == 0))
            return UAPS_ENOFORM;
        a.bs_y = hints.bsum_y; a.bs_mean = hints.bsum_mean; a.bs_invstd = hints.bsum_invstd; a.bs_gamma = hints.bsum_gamma;
        a.bs_beta = hints.bsum_beta; a.bs_slope = hints.bsum_slope; a.bs_Bg = B / hints.bsum_groups; a.bs_max = hints.bsum_max;
        a.stats = (float2*)hints.bsum_partials;
        uaps::account_bytes(4.0 * B * H * W * Cout);      // the BatchNorm's raw input, once more
    }
    // the up-sampling form exists in ONE kernel: up4's first convolution on a 256-wide map (16 + 16 -> 16 channels, bounded operands)
    // (every condition of the branch below that reaches launch_hr16_up is part of this test -- the 32-pixel-wide tile plan, no 32x32 /
    // GEMM-tiled plan -- so that no other kernel can ever be handed the low-resolution x2 as if it were a full-resolution tensor)
    if (up2 && !(row16 && wide && !p.s32 && !p.g1 && p.dil == 1 && x2 && Csplit == 16 && Cin == 32 && W == 256 && !xf && p.vec &&
                 up2_pattern_ok(W / 2)))
        return UAPS_ENOFORM;
    if (!no_small && !row16 && p.small && p.vec && !stats && !x2 && !y2 && (p.small == 1 || !xf)) return launch_small(a, p.small, s);
    if (p.split) {
        // the split weights follow the exact ones in the packed buffer (uaps_conv_pack_floats)
        a.wp = wp + (size_t)ks * ks * p.CinP * p.CoutP;
        a.CinP = split_cgroups(Cin);
        // mode 2 and every tensor operand bounded: the two-piece fp16 form; its weights follow the bf16 pieces behind a header
        if (conv_mode() == 2 && hints.bound[0] && (!x2 || Csplit >= Cin || hints.bound[1])) {
            const size_t piece = (size_t)ks * ks * a.CinP * p.CoutP * 4;
            a.wscale = a.wp + 3 * piece + 16;         // {s, 1 / s, max|w|} behind the 16 partial maxima
            a.wp = a.wp + 3 * piece + kH16Header;
            a.in_bound = hints.bound[0]; a.in_mul = hints.mul[0];
            if (x2 && Csplit < Cin) { a.in2_bound = hints.bound[1]; a.in2_mul = hints.mul[1]; }
            a.err = uaps::error_word();
        }
        // the whole-layer-width tile form (conv_split_g.hpp): with statistics only when the caller promised the bounds up front
        // (UAPS_CONV_BOUNDED: the partial-sum layout of this form is what uaps_conv_fwd_stats_parts reported for the same bit)
        // (not with the staging-time BatchNorm: its transform sits exposed in a one-wave-per-SIMD kernel -- 46.7 against 41.1 us of
        // conv_hfwd_bn_kernel -- so such calls keep the tile kernels AND their partial-sum layout: see uaps_conv_fwd_stats_parts)
        if (p.g128 && xf && stats && p.bounded) return UAPS_EINVAL;      // (the bit changes the statistics layout; this form has none)
        if (p.g128 && !xf && stats && p.bounded && !a.wscale) return UAPS_EINVAL;
        if (p.g128 && !xf && a.wscale && (!stats || p.bounded) && (Csplit == Cin || Csplit % 32 == 0) && (Osplit == Cout || Osplit % 128 == 0) &&
            !hints.out_amax && !bsum && !up2)
            return launch_hg128(a, s);
        if (p.g1) {
            // no two-tensor / BatchNorm-in-staging form of the GEMM-tiled kernels: the 3x3-style tiling (never with statistics, see above)
            if (!x2 && !y2 && !xf) {
                const bool wide = a.wscale && p.CoutP % 256 == 0 && !(g_conv_tuning & UAPS_TUNE_G1_NARROW);
                return launch_g1(a, wide ? 256 : (p.CoutP % 128 == 0 ? 128 : 64), s);
            }
        }
        // 16 -> 32 channels on a 256-wide map without statistics (the input gradient of up4's two-tensor convolution): two output
        // tiles of the full-width-row kernel, written as one or two 16-channel tensors
        if (a.wscale && ks == 3 && p.dil == 1 && Cin == 16 && Cout == 32 && !stats && !xf && !x2 && (Osplit == Cout || Osplit == 16) &&
            W % 256 == 0 && H % 16 == 0 && !(g_conv_tuning & (UAPS_TUNE_NO_ROW16 | UAPS_TUNE_NO_HP16))) return launch_hr16x2(a, s);
        if (p.dil > 1) return p.sbn == 64 ? launch_s32d<64>(a, p.dil, s) : launch_s32d<32>(a, p.dil, s);
        if (p.s32 && p.s32t) return launch_s32t(a, s);
        if (p.s32) return p.sbn == 64 ? launch_s32<64>(a, s) : launch_s32<32>(a, s);
        const bool no_hp16 = (g_conv_tuning & UAPS_TUNE_NO_HP16) != 0;
        if (!no_hp16 && a.wscale && ks == 3 && wide && p.CoutP == 16 && Cin > 8 && Cin <= 32 && Osplit == Cout) {
            if (up2) return launch_hr16_up(a, s);
            if (bsum) return launch_hr16_bs(a, s);
            if (W % 256 == 0 && H % 16 == 0 && !(g_conv_tuning & UAPS_TUNE_NO_ROW16)) return launch_hr16(a, s);
            return launch_hp16(a, s);
        }
        if (up2) return UAPS_ENOFORM;      // (unreachable by the test above; kept so that a future edit of either cannot launch a wrong form)
        if (ks == 3) return wide ? dispatch_sfwd<3, 8, 32>(a, p.sbn, p.sck, s) : dispatch_sfwd<3, 16, 16>(a, p.sbn, p.sck, s);
        return wide ? dispatch_sfwd<1, 8, 32>(a, p.sbn, p.sck, s) : dispatch_sfwd<1, 16, 16>(a, p.sbn, p.sck, s);
    }
    if (up2) return UAPS_ENOFORM;
    if (p.dil == 2) return dispatch_dilated<2>(a, p.bn, p.vec, p.extra_lds, s);
    if (p.dil == 4) return dispatch_dilated<4>(a, p.bn, p.vec, p.extra_lds, s);
    if (ks == 3) return wide ? dispatch_bn_ck<3, 8, 32>(a, p.bn, p.ck, p.vec, p.extra_lds, s) : dispatch_bn_ck<3, 16, 16>(a, p.bn, p.ck, p.vec, p.extra_lds, s);
    return wide ? dispatch_bn_ck<1, 8, 32>(a, p.bn, p.ck, p.vec, p.extra_lds, s) : dispatch_bn_ck<1, 16, 16>(a, p.bn, p.ck, p.vec, p.extra_lds, s);
}

}  // namespace

static PackDesc make_pack_desc(const float* w, float* wf, float* wb, int Cout, int Cin, int ks) {
    PackDesc q{};
    q.w = w; q.wf = wf; q.wb = wb; q.Cout = Cout; q.Cin = Cin; q.taps = ks * ks;
    q.CinP = kdim_pad(Cin, ks); q.CoutP = ndim_pad(Cout); q.CoutPk = kdim_pad(Cout, ks); q.CinPn = ndim_pad(Cin);
    q.CGf = split_cgroups(Cin); q.CGb = split_cgroups(Cout);
    return q;
}

// Sizes (in floats) of the two packed buffers of a convolution: each holds the exact fp32 layout followed by the
// three-piece bf16 layout of conv_split.hpp (3 * taps * groups * channels * 16 bytes), a 16-byte header and the
// two-piece fp16 layout (2 * taps * groups * channels * 16 bytes).
extern "C" int uaps_conv_pack_floats(int Cout, int Cin, int ks, size_t* fwd_floats, size_t* bwd_floats) {
    if (Cout <= 0 || Cin <= 0 || (ks != 1 && ks != 3)) return UAPS_EINVAL;
    const size_t taps = (size_t)ks * ks;
    const PackDesc q = make_pack_desc(nullptr, nullptr, nullptr, Cout, Cin, ks);
    (void)taps;
    if (fwd_floats) *fwd_floats = (size_t)pack_fwd_floats(q);
    if (bwd_floats) *bwd_floats = (size_t)pack_bwd_floats(q);
    return UAPS_OK;
}

extern "C" int uaps_conv_pack_weights(const float* w, int Cout, int Cin, int ks, float* wf, float* wb, uaps_stream_t stream) {
    if (!w || Cout <= 0 || Cin <= 0 || (ks != 1 && ks != 3) || (!wf && !wb)) return UAPS_EINVAL;
    if (((uintptr_t)wf | (uintptr_t)wb) % 16) return UAPS_EINVAL;
    const PackDesc q = make_pack_desc(w, wf, wb, Cout, Cin, ks);
    const long n = conv_pack_elems(q);
    uaps::account_bytes(4.0 * ((double)Cout * Cin * ks * ks + (double)n));      // the weight once, every packed layout once
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(conv_weight_scale_kernel, dim3(kH16Parts), dim3(256), 0, (hipStream_t)stream, q);
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, q);
    return (int)hipGetLastError();
}

// The same for n convolutions at once (host arrays of n entries each); any wf[i] / wb[i] may be null.
extern "C" int uaps_conv_pack_weights_batch(const float* const* w, float* const* wf, float* const* wb, const int* Cout,
                                            const int* Cin, const int* ks, int n, uaps_stream_t stream) {
    if (!w || !wf || !wb || !Cout || !Cin || !ks || n <= 0) return UAPS_EINVAL;
    for (int base = 0; base < n; base += kPackBatch) {
        const int m = n - base < kPackBatch ? n - base : kPackBatch;
        PackBatch pb{};
        long most = 0;
        for (int i = 0; i < m; ++i) {
            const int k = base + i;
            if (!w[k] || Cout[k] <= 0 || Cin[k] <= 0 || (ks[k] != 1 && ks[k] != 3)) return UAPS_EINVAL;
            if (((uintptr_t)wf[k] | (uintptr_t)wb[k]) % 16) return UAPS_EINVAL;
            pb.d[i] = make_pack_desc(w[k], wf[k], wb[k], Cout[k], Cin[k], ks[k]);
            const long e = conv_pack_elems(pb.d[i]);
            uaps::account_bytes(4.0 * ((double)Cout[k] * Cin[k] * ks[k] * ks[k] + (double)e));
            if (e > most) most = e;
        }
        const int bx = (int)((most + 255) / 256 < 256 ? (most + 255) / 256 : 256);
        hipLaunchKernelGGL(conv_weight_scale_batch_kernel, dim3(kH16Parts, m), dim3(256), 0, (hipStream_t)stream, pb);
        hipLaunchKernelGGL(conv_pack_weights_batch_kernel, dim3(bx, m), dim3(256), 0, (hipStream_t)stream, pb);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return UAPS_OK;
}

#ifdef UAPS_STAMPS
// diagnostic build only (stamps.hpp; not part of include/uaps_hip.h): where the stamped kernels of this file write, nwaves * 20 words
extern "C" int uaps_debug_set_stamp_buffer(unsigned long long* buf, unsigned long long nwaves) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(uaps::g_stamp_buf), &buf, sizeof buf);
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(uaps::g_stamp_waves), &nwaves, sizeof nwaves);
    return (int)e;
}
#endif

// Every entry point below exists twice: `uaps_X(args)` takes its optional operands (bounds, statistics shift, ...) from the calling
// thread's pending uaps_next_call_hints record (the legacy form), `uaps_X_h(hints, args)` takes them as its first argument (NULL =
// none) and reads nothing thread-local (round 6: the form the Python package calls).
extern "C" int uaps_conv_fwd_h(const uaps_call_hints* hints, const float* x, const float* wf, const float* bias, float* y, int B, int Cin, int Cout,
                               int H, int W, int ks, int cfg, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    return conv_fwd_any(h, x, wf, bias, y, B, Cin, Cout, H, W, ks, cfg, (hipStream_t)stream);
}
extern "C" int uaps_conv_fwd(const float* x, const float* wf, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                             int ks, int cfg, uaps_stream_t stream) {
    return conv_fwd_any(take_hints(), x, wf, bias, y, B, Cin, Cout, H, W, ks, cfg, (hipStream_t)stream);
}

// Forward convolution that also writes, per output channel, image and pixel tile, the (sum, sum of squares) of
// its output: the first pass of the train-mode BatchNorm that follows every 3x3 conv of a ConvBlock
// (UAPS_unet.py:37-38, 41-42), for uaps_bn_act_fwd_train_partials.  stats: float2 [Cout][B][parts_per_image].
extern "C" int uaps_conv_fwd_stats_h(const uaps_call_hints* hints, const float* x, const float* wf, const float* bias, float* y, void* stats,
                                     int B, int Cin, int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    if (!stats) return UAPS_EINVAL;
    return conv_fwd_any(h, x, wf, bias, y, B, Cin, Cout, H, W, ks, cfg, (hipStream_t)stream, (float2*)stats);
}
extern "C" int uaps_conv_fwd_stats(const float* x, const float* wf, const float* bias, float* y, void* stats, int B, int Cin,
                                   int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream) {
    const uaps_call_hints h = take_hints();
    if (!stats) return UAPS_EINVAL;
    return conv_fwd_any(h, x, wf, bias, y, B, Cin, Cout, H, W, ks, cfg, (hipStream_t)stream, (float2*)stats);
}

// y = conv(leaky_relu(batch_norm_train(x_raw))) where x_raw is a previous conv's raw output and the normalisation
// coefficients xf [groups][Cin] float2 (scale, shift) come from uaps_bn_finalize_train: the activated
// tensor between the two convs of a ConvBlock (UAPS_unet.py:38-41) is never written.  stats may be NULL.
extern "C" int uaps_conv_fwd_bn_h(const uaps_call_hints* hints, const float* x_raw, const void* xf, float slope, int groups, const float* wf,
                                  const float* bias, float* y, void* stats, int B, int Cin, int Cout, int H, int W, int ks, int cfg,
                                  uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    if (!xf) return UAPS_EINVAL;
    return conv_fwd_any(h, x_raw, wf, bias, y, B, Cin, Cout, H, W, ks, cfg, (hipStream_t)stream, (float2*)stats, nullptr, -1, nullptr, -1,
                        xf, slope, groups);
}
extern "C" int uaps_conv_fwd_bn(const float* x_raw, const void* xf, float slope, int groups, const float* wf, const float* bias,
                                float* y, void* stats, int B, int Cin, int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream) {
    const uaps_call_hints h = take_hints();
    if (!xf) return UAPS_EINVAL;
    return conv_fwd_any(h, x_raw, wf, bias, y, B, Cin, Cout, H, W, ks, cfg, (hipStream_t)stream, (float2*)stats, nullptr, -1, nullptr, -1,
                        xf, slope, groups);
}

extern "C" int uaps_conv_fwd_stats_parts(int B, int Cin, int Cout, int H, int W, int ks, int cfg, int* parts_per_image) {
    FwdPlan p{};
    const int rc = plan_fwd(nullptr, nullptr, B, Cin, Cout, H, W, ks, cfg, &p);
    if (rc) return rc;
    if (!parts_per_image) return UAPS_EINVAL;
    *parts_per_image = p.g1 ? (H * W + 127) / 128 : ((H + p.th - 1) / p.th) * ((W + p.tw - 1) / p.tw);
    if (p.g128 && p.bounded) *parts_per_image = H / 4;      // conv_split_g.hpp: one part per 4-row band
    return UAPS_OK;
}

// Convolution of the never-materialised concatenation torch.cat([x1, x2], dim=1) (UpBlock, UAPS_unet.py:84-85):
// x1 [B,C1,H,W], x2 [B,C2,H,W], weights packed for Cin = C1 + C2; C1 % 8 == 0.  stats may be NULL.
extern "C" int uaps_conv_fwd_cat_h(const uaps_call_hints* hints, const float* x1, int C1, const float* x2, int C2, const float* wf,
                                   const float* bias, float* y, void* stats, int B, int Cout, int H, int W, int ks, int cfg,
                                   uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    if (!x2 || C1 <= 0 || C2 <= 0) return UAPS_EINVAL;
    return conv_fwd_any(h, x1, wf, bias, y, B, C1 + C2, Cout, H, W, ks, cfg, (hipStream_t)stream, (float2*)stats, x2, C1);
}
extern "C" int uaps_conv_fwd_cat(const float* x1, int C1, const float* x2, int C2, const float* wf, const float* bias, float* y,
                                 void* stats, int B, int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream) {
    const uaps_call_hints h = take_hints();
    if (!x2 || C1 <= 0 || C2 <= 0) return UAPS_EINVAL;
    return conv_fwd_any(h, x1, wf, bias, y, B, C1 + C2, Cout, H, W, ks, cfg, (hipStream_t)stream, (float2*)stats, x2, C1);
}

// Input gradient of that convolution, written as two tensors: dx1 [B,C1,H,W] and dx2 [B,C2,H,W].
extern "C" int uaps_conv_bwd_data_cat_h(const uaps_call_hints* hints, const float* dy, const float* wb, float* dx1, int C1, float* dx2, int C2,
                                        int B, int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    if (!dx2 || C1 <= 0 || C2 <= 0) return UAPS_EINVAL;
    return conv_fwd_any(h, dy, wb, nullptr, dx1, B, Cout, C1 + C2, H, W, ks, cfg, (hipStream_t)stream, nullptr, nullptr, -1, dx2, C1);
}
extern "C" int uaps_conv_bwd_data_cat(const float* dy, const float* wb, float* dx1, int C1, float* dx2, int C2, int B, int Cout,
                                      int H, int W, int ks, int cfg, uaps_stream_t stream) {
    const uaps_call_hints h = take_hints();
    if (!dx2 || C1 <= 0 || C2 <= 0) return UAPS_EINVAL;
    return conv_fwd_any(h, dy, wb, nullptr, dx1, B, Cout, C1 + C2, H, W, ks, cfg, (hipStream_t)stream, nullptr, nullptr, -1, dx2, C1);
}

// dx = conv(dy, W^T flipped): the same kernel with the roles of the channel counts exchanged
extern "C" int uaps_conv_bwd_data_h(const uaps_call_hints* hints, const float* dy, const float* wb, float* dx, int B, int Cin, int Cout, int H,
                                    int W, int ks, int cfg, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    return conv_fwd_any(h, dy, wb, nullptr, dx, B, Cout, Cin, H, W, ks, cfg, (hipStream_t)stream);
}
extern "C" int uaps_conv_bwd_data(const float* dy, const float* wb, float* dx, int B, int Cin, int Cout, int H, int W, int ks,
                                  int cfg, uaps_stream_t stream) {
    return conv_fwd_any(take_hints(), dy, wb, nullptr, dx, B, Cout, Cin, H, W, ks, cfg, (hipStream_t)stream);
}

// Name of the kernel instantiation uaps_conv_fwd / uaps_conv_bwd_data launch for these dimensions, as
// rocprofv3 prints it (bench.py groups its HIP-event timings by it).  For bwd_data pass (Cout, Cin) swapped.
extern "C" int uaps_conv_fwd_variant(int B, int Cin, int Cout, int H, int W, int ks, int cfg, char* buf, size_t buflen) {
    FwdPlan p{};
    const int rc = plan_fwd(nullptr, nullptr, B, Cin, Cout, H, W, ks, cfg, &p);
    if (rc) return rc;
    if (!buf || buflen < 64) return UAPS_EINVAL;
    if (p.small) snprintf(buf, buflen, "conv_small_kernel<8, 4>");
    else if (p.g1) snprintf(buf, buflen, "conv_g1s_kernel<%d>", p.CoutP % 128 == 0 ? 128 : 64);
    else if (p.s32 && p.dil > 1) snprintf(buf, buflen, "conv_s32d_kernel<%d, %d>", p.sbn, p.dil);
    else if (p.s32 && p.s32t) snprintf(buf, buflen, "conv_s32t_kernel<32>");
    else if (p.s32) snprintf(buf, buflen, "conv_s32_kernel<%d>", p.sbn);
    else if (p.split) snprintf(buf, buflen, "conv_sfwd_kernel<%d, %d, %d, %d, %d>", ks, p.th, p.tw, p.sbn, p.sck);
    else snprintf(buf, buflen, "conv_fwd_kernel<%d, %d, %d, %d, %d, %d, %d>", ks, p.th, p.tw, p.bn, p.ck, p.vec ? 4 : 1, p.dil);
    return UAPS_OK;
}
