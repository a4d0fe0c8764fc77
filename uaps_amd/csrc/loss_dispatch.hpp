// Host-side dispatch helpers shared by the loss translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uaps_hip.h"
#include "loss_kernels.hpp"

namespace uaps {

struct LossArgs {
    const float* const* logits; float* const* dlogits; const double* w;
    int D, B, C, H, W; float cw1, cw2, eps;
    int64_t* pseudo; const int64_t* labels; float* var; float* scalars; const float* cscalars;
    const float* gscale; float* partials; hipStream_t stream;
};

inline int check_dims(int D, int B, int C, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (D < 1 || D > UAPS_MAX_HEADS || C < 2 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    return UAPS_OK;
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// widest per-thread pixel vector the register budget allows for D*C live logits per pixel
template <int D, int C> constexpr int unsup_vec() { return D * C <= 16 ? 4 : (D * C <= 32 ? 2 : 1); }

// pixels per thread of the double-buffered unsupervised forward of the pair kernel (two register sets of D*C*VEC logits)
template <int D, int C> constexpr int unsup_vec_pf() { return D * C <= 16 ? 2 : 1; }

inline int grid_for(long ngroups) {
    long b = (ngroups + kThreads - 1) / kThreads;
    if (b > kMaxBlocks) b = kMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

template <int D> HeadPtrs<D> in_ptrs(const float* const* p) { HeadPtrs<D> h; for (int k = 0; k < D; ++k) h.p[k] = p[k]; return h; }
template <int D> HeadOutPtrs<D> out_ptrs(float* const* p) { HeadOutPtrs<D> h; for (int k = 0; k < D; ++k) h.p[k] = p[k]; return h; }

// can the D input (and optional output) planes be accessed VEC-wide?
inline bool vec_ok(const LossArgs& a, int vec, bool outs) {
    if (vec == 1) return true;
    const long HW = (long)a.H * a.W;
    if (HW % vec) return false;
    for (int k = 0; k < a.D; ++k) {
        if (!aligned_to(a.logits[k], 4 * vec)) return false;
        if (outs && !aligned_to(a.dlogits[k], 4 * vec)) return false;
    }
    if (a.pseudo && !aligned_to(a.pseudo, 16)) return false;
    if (a.labels && !aligned_to(a.labels, 16)) return false;
    if (a.var && !aligned_to(a.var, 4 * vec)) return false;
    return true;
}

#define UAPS_DISPATCH_C(FN, Dv, a)                 \
    switch ((a).C) {                               \
        case 2: return FN<Dv, 2>(a);               \
        case 3: return FN<Dv, 3>(a);               \
        case 4: return FN<Dv, 4>(a);               \
        case 5: return FN<Dv, 5>(a);               \
        case 6: return FN<Dv, 6>(a);               \
        case 7: return FN<Dv, 7>(a);               \
        case 8: return FN<Dv, 8>(a);               \
        default: return UAPS_ERANGE;               \
    }
#define UAPS_DISPATCH_DC(FN, a)                    \
    switch ((a).D) {                               \
        case 1: UAPS_DISPATCH_C(FN, 1, a)          \
        case 2: UAPS_DISPATCH_C(FN, 2, a)          \
        case 3: UAPS_DISPATCH_C(FN, 3, a)          \
        case 4: UAPS_DISPATCH_C(FN, 4, a)          \
        case 5: UAPS_DISPATCH_C(FN, 5, a)          \
        case 6: UAPS_DISPATCH_C(FN, 6, a)          \
        case 7: UAPS_DISPATCH_C(FN, 7, a)          \
        case 8: UAPS_DISPATCH_C(FN, 8, a)          \
        default: return UAPS_ERANGE;               \
    }

// both branches of a step (labelled logits `lab`, unlabelled logits `un`, D pointers each)
struct PairArgs {
    const float* const* lab; const float* const* un; float* const* dlab; float* const* dun; const double* w;
    int D, B, C, H, W; float ce_coef, dice_coef, cw1, cw2, eps;
    const int64_t* labels; int64_t* pseudo; const int64_t* cpseudo; float* var; float* sscal; float* uscal; double* sums;
    long Nloss; const float* gscale; float* partials; int cfg; hipStream_t stream;
    float* amax_out;      // backward: device scalar raised to max|gradient element| (uaps_call_hints::out_amax), or nullptr
};
// 16-byte access to every plane of both branches?
inline bool pair_vec_ok(const PairArgs& a, int vec, bool outs) {
    const long HW = (long)a.H * a.W;
    if (HW % vec) return false;
    for (int k = 0; k < a.D; ++k) {
        if (!aligned_to(a.lab[k], 16) || !aligned_to(a.un[k], 16)) return false;
        if (outs && (!aligned_to(a.dlab[k], 16) || !aligned_to(a.dun[k], 16))) return false;
    }
    if (!aligned_to(a.labels, 16)) return false;
    if (a.pseudo && !aligned_to(a.pseudo, 16)) return false;
    if (a.cpseudo && !aligned_to(a.cpseudo, 16)) return false;
    if (a.var && !aligned_to(a.var, 16)) return false;
    return true;
}
// blocks of one branch: persistent blocks, a few groups of pixels per thread (the forward prefetches the next group while it
// computes the current one); `cap` <= kMaxBlocks
inline int pair_grid(long ngroups, int cap) {
    int b = grid_for(ngroups);
    if (cap > 0 && b > cap) b = cap;
    return b;
}
// default split of the 2 x 256 resident blocks (two per CU at two waves per SIMD) between the branches, by their work;
// cfg = (supervised cap << 16) | unsupervised cap overrides it (tools/bench_loss.py)
inline int pair_cap_s(int cfg) { return (cfg >> 16) & 0xfff ? (cfg >> 16) & 0xfff : 192; }      // bits 28-30: experiment variants
inline int pair_cap_u(int cfg) { return cfg & 0xfff ? cfg & 0xfff : 320; }

int launch_pair_fwd(const PairArgs& a);
int launch_pair_bwd(const PairArgs& a);
int launch_unsup_fwd(const LossArgs& a);
int launch_unsup_bwd(const LossArgs& a);
int launch_sup_fwd(const LossArgs& a);
int launch_sup_bwd(const LossArgs& a);

}  // namespace uaps
