// extern "C" entry points of the forward / input-gradient convolution and the weight packing
// (include/uaps_hip.h, section "Convolutions").
#include "../../include/uaps_hip.h"
#include <stdio.h>
#include "conv_kernels.hpp"
using namespace uaps;

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
            hipLaunchKernelGGL((conv_fwd_bn_kernel<KS, TH, TW, BN, CK, 4, 1>), dim3((unsigned)grid), dim3(kConvThreads), extra_lds, s, a);
            return (int)hipGetLastError();
        } else {
            return UAPS_ERANGE;
        }
    }
    if (vec) hipLaunchKernelGGL((conv_fwd_kernel<KS, TH, TW, BN, CK, 4, DIL>), dim3((unsigned)grid), dim3(kConvThreads), extra_lds, s, a);
    else hipLaunchKernelGGL((conv_fwd_kernel<KS, TH, TW, BN, CK, 1, DIL>), dim3((unsigned)grid), dim3(kConvThreads), extra_lds, s, a);
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

struct FwdPlan { int ck, bn, th, tw; bool vec; int CinP, CoutP, extra_lds, dil; };

int plan_fwd(const void* x, const void* y, int B, int Cin, int Cout, int H, int W, int ks, int cfg, FwdPlan* p) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (ks != 1 && ks != 3) return UAPS_ERANGE;
    if ((double)Cin * H * W * 4.0 >= 2147483648.0 || (double)Cout * H * W * 4.0 >= 2147483648.0) return UAPS_ERANGE;
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
    return UAPS_OK;
}

// x [B,Cin,H,W] * packed weights [taps][CinP][CoutP] -> y [B,Cout,H,W]
// x2 / Csplit: optional second input tensor holding channels [Csplit, Cin); y2 / Osplit likewise for the output
int conv_fwd_any(const float* x, const float* wp, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int ks,
                 int cfg, hipStream_t s, float2* stats = nullptr, const float* x2 = nullptr, int Csplit = -1,
                 float* y2 = nullptr, int Osplit = -1, const void* xf = nullptr, float xf_slope = 0.f, int groups = 1) {
    if (!x || !wp || !y) return UAPS_EINVAL;
    if (xf && (x2 || groups < 1 || B % groups || (uintptr_t)xf % 8)) return UAPS_EINVAL;
    if (xf && !(xf_slope >= 0.f && xf_slope <= 1.f)) return UAPS_ERANGE;      // leaky_relu is evaluated as max(z, slope * z)
    if (Csplit < 0 || !x2) Csplit = Cin;
    if (Osplit < 0 || !y2) Osplit = Cout;
    if (Csplit > Cin || Osplit > Cout || (Csplit < Cin && (Csplit % 8 || Csplit == 0)) || (Osplit < Cout && Osplit == 0)) return UAPS_EINVAL;
    FwdPlan p{};
    const int rc = plan_fwd(x, y, B, Cin, Cout, H, W, ks, cfg, &p);
    if (rc) return rc;
    if ((x2 && (uintptr_t)x2 % 16) || (y2 && (uintptr_t)y2 % 16)) p.vec = false;
    ConvFwdArgs a{};
    a.in = x; a.in2 = x2; a.Csplit = Csplit; a.out2 = y2; a.Osplit = Osplit;
    a.wp = wp; a.bias = bias; a.out = y; a.stats = stats; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.CinP = p.CinP; a.CoutP = p.CoutP;
    a.xf = (const float2*)xf; a.xf_slope = xf_slope; a.xf_Bg = xf ? B / groups : B;
    const bool wide = p.tw == 32;
    if (p.dil == 2) return dispatch_dilated<2>(a, p.bn, p.vec, p.extra_lds, s);
    if (p.dil == 4) return dispatch_dilated<4>(a, p.bn, p.vec, p.extra_lds, s);
    if (ks == 3) return wide ? dispatch_bn_ck<3, 8, 32>(a, p.bn, p.ck, p.vec, p.extra_lds, s) : dispatch_bn_ck<3, 16, 16>(a, p.bn, p.ck, p.vec, p.extra_lds, s);
    return wide ? dispatch_bn_ck<1, 8, 32>(a, p.bn, p.ck, p.vec, p.extra_lds, s) : dispatch_bn_ck<1, 16, 16>(a, p.bn, p.ck, p.vec, p.extra_lds, s);
}

}  // namespace

extern "C" int uaps_conv_pack_floats(int Cout, int Cin, int ks, size_t* fwd_floats, size_t* bwd_floats) {
    if (Cout <= 0 || Cin <= 0 || (ks != 1 && ks != 3)) return UAPS_EINVAL;
    const size_t taps = (size_t)ks * ks;
    if (fwd_floats) *fwd_floats = taps * kdim_pad(Cin, ks) * ndim_pad(Cout);
    if (bwd_floats) *bwd_floats = taps * kdim_pad(Cout, ks) * ndim_pad(Cin);
    return UAPS_OK;
}

extern "C" int uaps_conv_pack_weights(const float* w, int Cout, int Cin, int ks, float* wf, float* wb, uaps_stream_t stream) {
    if (!w || Cout <= 0 || Cin <= 0 || (ks != 1 && ks != 3) || (!wf && !wb)) return UAPS_EINVAL;
    const int taps = ks * ks;
    const long n = (long)taps * (kdim_pad(Cin, ks) * ndim_pad(Cout) + kdim_pad(Cout, ks) * ndim_pad(Cin));
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wf, wb, Cout, Cin, taps,
                       kdim_pad(Cin, ks), ndim_pad(Cout), kdim_pad(Cout, ks), ndim_pad(Cin));
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
            PackDesc& q = pb.d[i];
            q.w = w[k]; q.wf = wf[k]; q.wb = wb[k]; q.Cout = Cout[k]; q.Cin = Cin[k]; q.taps = ks[k] * ks[k];
            q.CinP = kdim_pad(Cin[k], ks[k]); q.CoutP = ndim_pad(Cout[k]); q.CoutPk = kdim_pad(Cout[k], ks[k]); q.CinPn = ndim_pad(Cin[k]);
            const long e = (long)q.taps * ((long)q.CinP * q.CoutP + (long)q.CoutPk * q.CinPn);
            if (e > most) most = e;
        }
        const int bx = (int)((most + 255) / 256 < 256 ? (most + 255) / 256 : 256);
        hipLaunchKernelGGL(conv_pack_weights_batch_kernel, dim3(bx, m), dim3(256), 0, (hipStream_t)stream, pb);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return UAPS_OK;
}

extern "C" int uaps_conv_fwd(const float* x, const float* wf, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                             int ks, int cfg, uaps_stream_t stream) {
    return conv_fwd_any(x, wf, bias, y, B, Cin, Cout, H, W, ks, cfg, (hipStream_t)stream);
}

// Forward convolution that also writes, per output channel, image and pixel tile, the (sum, sum of squares) of
// its output: the first pass of the train-mode BatchNorm that follows every 3x3 conv of a ConvBlock
// (UAPS_unet.py:37-38, 41-42), for uaps_bn_act_fwd_train_partials.  stats: float2 [Cout][B][parts_per_image].
extern "C" int uaps_conv_fwd_stats(const float* x, const float* wf, const float* bias, float* y, void* stats, int B, int Cin,
                                   int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream) {
    if (!stats) return UAPS_EINVAL;
    return conv_fwd_any(x, wf, bias, y, B, Cin, Cout, H, W, ks, cfg, (hipStream_t)stream, (float2*)stats);
}

// y = conv(leaky_relu(batch_norm_train(x_raw))) where x_raw is a previous conv's raw output and the normalisation
// coefficients xf [groups][Cin] float2 (scale, shift) come from uaps_bn_finalize_train: the activated
// tensor between the two convs of a ConvBlock (UAPS_unet.py:38-41) is never written.  stats may be NULL.
extern "C" int uaps_conv_fwd_bn(const float* x_raw, const void* xf, float slope, int groups, const float* wf, const float* bias,
                                float* y, void* stats, int B, int Cin, int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream) {
    if (!xf) return UAPS_EINVAL;
    return conv_fwd_any(x_raw, wf, bias, y, B, Cin, Cout, H, W, ks, cfg, (hipStream_t)stream, (float2*)stats, nullptr, -1, nullptr, -1,
                        xf, slope, groups);
}

extern "C" int uaps_conv_fwd_stats_parts(int B, int Cin, int Cout, int H, int W, int ks, int cfg, int* parts_per_image) {
    FwdPlan p{};
    const int rc = plan_fwd(nullptr, nullptr, B, Cin, Cout, H, W, ks, cfg, &p);
    if (rc) return rc;
    if (!parts_per_image) return UAPS_EINVAL;
    *parts_per_image = ((H + p.th - 1) / p.th) * ((W + p.tw - 1) / p.tw);
    return UAPS_OK;
}

// Convolution of the never-materialised concatenation torch.cat([x1, x2], dim=1) (UpBlock, UAPS_unet.py:84-85):
// x1 [B,C1,H,W], x2 [B,C2,H,W], weights packed for Cin = C1 + C2; C1 % 8 == 0.  stats may be NULL.
extern "C" int uaps_conv_fwd_cat(const float* x1, int C1, const float* x2, int C2, const float* wf, const float* bias, float* y,
                                 void* stats, int B, int Cout, int H, int W, int ks, int cfg, uaps_stream_t stream) {
    if (!x2 || C1 <= 0 || C2 <= 0) return UAPS_EINVAL;
    return conv_fwd_any(x1, wf, bias, y, B, C1 + C2, Cout, H, W, ks, cfg, (hipStream_t)stream, (float2*)stats, x2, C1);
}

// Input gradient of that convolution, written as two tensors: dx1 [B,C1,H,W] and dx2 [B,C2,H,W].
extern "C" int uaps_conv_bwd_data_cat(const float* dy, const float* wb, float* dx1, int C1, float* dx2, int C2, int B, int Cout,
                                      int H, int W, int ks, int cfg, uaps_stream_t stream) {
    if (!dx2 || C1 <= 0 || C2 <= 0) return UAPS_EINVAL;
    return conv_fwd_any(dy, wb, nullptr, dx1, B, Cout, C1 + C2, H, W, ks, cfg, (hipStream_t)stream, nullptr, nullptr, -1, dx2, C1);
}

// dx = conv(dy, W^T flipped): the same kernel with the roles of the channel counts exchanged
extern "C" int uaps_conv_bwd_data(const float* dy, const float* wb, float* dx, int B, int Cin, int Cout, int H, int W, int ks,
                                  int cfg, uaps_stream_t stream) {
    return conv_fwd_any(dy, wb, nullptr, dx, B, Cout, Cin, H, W, ks, cfg, (hipStream_t)stream);
}

// Name of the kernel instantiation uaps_conv_fwd / uaps_conv_bwd_data launch for these dimensions, as
// rocprofv3 prints it (bench.py groups its HIP-event timings by it).  For bwd_data pass (Cout, Cin) swapped.
extern "C" int uaps_conv_fwd_variant(int B, int Cin, int Cout, int H, int W, int ks, int cfg, char* buf, size_t buflen) {
    FwdPlan p{};
    const int rc = plan_fwd(nullptr, nullptr, B, Cin, Cout, H, W, ks, cfg, &p);
    if (rc) return rc;
    if (!buf || buflen < 64) return UAPS_EINVAL;
    snprintf(buf, buflen, "conv_fwd_kernel<%d, %d, %d, %d, %d, %d, %d>", ks, p.th, p.tw, p.bn, p.ck, p.vec ? 4 : 1, p.dil);
    return UAPS_OK;
}
