// uaps_conv_ex (include/uaps_hip.h): the convolution entry points with every operand in one size-versioned struct.
#include "../../include/uaps_hip.h"
#include <string.h>
#include <stddef.h>

extern "C" int uaps_conv_ex(const uaps_conv_call* c) {
    if (!c) return UAPS_EINVAL;
    const unsigned n = c->struct_size;
    // everything up to and including `stream` is mandatory; the hints record at the tail is the part that grows (ABI 3: `stream` sits
    // in front of it, so a client built against a shorter uaps_call_hints still has its stream where this library reads it)
    if (n < offsetof(uaps_conv_call, hints) || n > sizeof(uaps_conv_call)) return UAPS_EINVAL;
    uaps_conv_call a;
    memset(&a, 0, sizeof a);
    memcpy(&a, c, n);
    // the struct's hints are THIS call's hints (none when hints.struct_size == 0): they travel as an argument into the *_h entry
    // points, nothing thread-local is read or written, and a record pending from uaps_next_call_hints is left alone
    const unsigned hmax = n - (unsigned)offsetof(uaps_conv_call, hints);
    if (a.hints.struct_size > hmax) return UAPS_EINVAL;          // a hints record that claims more bytes than the caller's struct holds
    const uaps_call_hints* h = a.hints.struct_size ? &a.hints : nullptr;
    const bool two = a.x2 != nullptr && a.C1 > 0 && a.C1 < a.Cin;
    switch (a.op) {
    case UAPS_CONV_FWD:
        if (a.xf) {
            if (two) return UAPS_ERANGE;      // no two-tensor form of the staging-time BatchNorm
            return uaps_conv_fwd_bn_h(h, a.x, a.xf, a.xf_slope, a.xf_groups > 0 ? a.xf_groups : 1, a.w_packed, a.bias, a.y, a.stats, a.B, a.Cin,
                                      a.Cout, a.H, a.W, a.ks, a.cfg, a.stream);
        }
        if (two) return uaps_conv_fwd_cat_h(h, a.x, a.C1, a.x2, a.Cin - a.C1, a.w_packed, a.bias, a.y, a.stats, a.B, a.Cout, a.H, a.W, a.ks, a.cfg, a.stream);
        if (a.stats) return uaps_conv_fwd_stats_h(h, a.x, a.w_packed, a.bias, a.y, a.stats, a.B, a.Cin, a.Cout, a.H, a.W, a.ks, a.cfg, a.stream);
        return uaps_conv_fwd_h(h, a.x, a.w_packed, a.bias, a.y, a.B, a.Cin, a.Cout, a.H, a.W, a.ks, a.cfg, a.stream);
    case UAPS_CONV_BWD_DATA:
        if (a.y2 && a.C1 > 0 && a.C1 < a.Cin)
            return uaps_conv_bwd_data_cat_h(h, a.x, a.w_packed, a.y, a.C1, a.y2, a.Cin - a.C1, a.B, a.Cout, a.H, a.W, a.ks, a.cfg, a.stream);
        return uaps_conv_bwd_data_h(h, a.x, a.w_packed, a.y, a.B, a.Cin, a.Cout, a.H, a.W, a.ks, a.cfg, a.stream);
    case UAPS_CONV_BWD_WEIGHT:
        if (a.xf) {
            if (two) return UAPS_ERANGE;
            return uaps_conv_bwd_weight_partial_bn_h(h, a.y_grad, a.x, a.xf, a.xf_slope, a.xf_groups > 0 ? a.xf_groups : 1, a.want_bias, a.B, a.Cin,
                                                     a.Cout, a.H, a.W, a.ks, a.cfg, a.workspace, a.workspace_bytes, a.stream);
        }
        if (two) return uaps_conv_bwd_weight_partial_cat_h(h, a.y_grad, a.x, a.C1, a.x2, a.Cin - a.C1, a.want_bias, a.B, a.Cout, a.H, a.W, a.ks,
                                                           a.cfg, a.workspace, a.workspace_bytes, a.stream);
        return uaps_conv_bwd_weight_partial_h(h, a.y_grad, a.x, a.want_bias, a.B, a.Cin, a.Cout, a.H, a.W, a.ks, a.cfg, a.workspace,
                                              a.workspace_bytes, a.stream);
    default:
        return UAPS_EINVAL;
    }
}
