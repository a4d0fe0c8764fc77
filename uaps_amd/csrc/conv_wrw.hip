// extern "C" entry points of the weight-gradient convolution (include/uaps_hip.h, "Convolutions").
#include "../../include/uaps_hip.h"
#include "conv_kernels.hpp"
using namespace uaps;

namespace {

struct WrwPlan { int TH, TW, mwc, nwc, ncob, ncib, nsplit, CoutS, CinS; long tiles; };

WrwPlan plan_wrw(int B, int Cin, int Cout, int H, int W, int cfg) {
    WrwPlan p{};
    const bool wide = W >= 32;
    p.TH = wide ? 8 : 16; p.TW = wide ? 32 : 16;
    p.mwc = Cout > 16 ? 2 : 1;
    p.nwc = Cin > 16 ? 2 : 1;
    p.ncob = (Cout + 16 * p.mwc - 1) / (16 * p.mwc);
    p.ncib = (Cin + 16 * p.nwc - 1) / (16 * p.nwc);
    p.CoutS = p.ncob * 16 * p.mwc; p.CinS = p.ncib * 16 * p.nwc;
    p.tiles = (long)B * ((H + p.TH - 1) / p.TH) * ((W + p.TW - 1) / p.TW);
    // pixel splits: fill the chip (two workgroups per CU) without making the slabs larger than needed
    long want = (512 + (long)p.ncob * p.ncib - 1) / ((long)p.ncob * p.ncib);
    if (cfg > 0) want = cfg;
    if (want > p.tiles) want = p.tiles;
    if (want < 1) want = 1;
    p.nsplit = (int)want;
    return p;
}

size_t wrw_ws_floats(const WrwPlan& p, int taps) { return (size_t)p.nsplit * ((size_t)taps * p.CoutS * p.CinS + p.CoutS); }

template <int KS, int TH, int TW, int MWC, int NWC>
int launch_wrw(ConvWrwArgs a, bool bias, hipStream_t s) {
    const long grid = (long)a.nsplit * a.ncob * a.ncib;
    if (bias) hipLaunchKernelGGL((conv_wrw_kernel<KS, TH, TW, MWC, NWC, true>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    else hipLaunchKernelGGL((conv_wrw_kernel<KS, TH, TW, MWC, NWC, false>), dim3((unsigned)grid), dim3(kConvThreads), 0, s, a);
    return (int)hipGetLastError();
}

template <int KS, int TH, int TW>
int dispatch_wrw(const ConvWrwArgs& a, int mwc, int nwc, bool bias, hipStream_t s) {
    if (mwc == 1 && nwc == 1) return launch_wrw<KS, TH, TW, 1, 1>(a, bias, s);
    if (mwc == 1 && nwc == 2) return launch_wrw<KS, TH, TW, 1, 2>(a, bias, s);
    if (mwc == 2 && nwc == 1) return launch_wrw<KS, TH, TW, 2, 1>(a, bias, s);
    return launch_wrw<KS, TH, TW, 2, 2>(a, bias, s);
}

}  // namespace

extern "C" int uaps_conv_wrw_workspace_bytes(int B, int Cin, int Cout, int H, int W, int ks, int cfg, size_t* out) {
    if (!out || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (ks != 1 && ks != 3)) return UAPS_EINVAL;
    *out = wrw_ws_floats(plan_wrw(B, Cin, Cout, H, W, cfg), ks * ks) * sizeof(float);
    return UAPS_OK;
}

extern "C" int uaps_conv_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int B, int Cin, int Cout, int H,
                                    int W, int ks, int cfg, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    if (!dy || !x || !dw || !ws || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (ks != 1 && ks != 3) return UAPS_ERANGE;
    const int taps = ks * ks;
    const WrwPlan p = plan_wrw(B, Cin, Cout, H, W, cfg);
    if (ws_bytes < wrw_ws_floats(p, taps) * sizeof(float)) return UAPS_EWORKSPACE;
    ConvWrwArgs a{};
    a.dout = dy; a.in = x; a.slab = (float*)ws; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.CoutS = p.CoutS; a.CinS = p.CinS; a.tiles_x = (W + p.TW - 1) / p.TW; a.tiles_y = (H + p.TH - 1) / p.TH;
    a.ncob = p.ncob; a.ncib = p.ncib; a.nsplit = p.nsplit;
    a.bslab = dbias ? a.slab + (size_t)p.nsplit * taps * p.CoutS * p.CinS : nullptr;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (ks == 3) rc = p.TW == 32 ? dispatch_wrw<3, 8, 32>(a, p.mwc, p.nwc, dbias != nullptr, s) : dispatch_wrw<3, 16, 16>(a, p.mwc, p.nwc, dbias != nullptr, s);
    else rc = p.TW == 32 ? dispatch_wrw<1, 8, 32>(a, p.mwc, p.nwc, dbias != nullptr, s) : dispatch_wrw<1, 16, 16>(a, p.mwc, p.nwc, dbias != nullptr, s);
    if (rc) return rc;
    const long n = (long)taps * p.CoutS * p.CinS + (dbias ? Cout : 0);
    hipLaunchKernelGGL(conv_wrw_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.slab, a.bslab, dw, dbias,
                       p.nsplit, taps, Cout, Cin, p.CoutS, p.CinS);
    return (int)hipGetLastError();
}
