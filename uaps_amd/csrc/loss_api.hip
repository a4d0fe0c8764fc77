// extern "C" entry points of the loss block (include/uaps_hip.h).
#include "loss_dispatch.hpp"
using namespace uaps;

extern "C" int uaps_abi_version(void) { return 3; }      // 3 (round 6): the *_h entry points (explicit hints), uaps_conv_call::stream in front of the hints; 2: struct_size, UAPS_ENOFORM, uaps_conv_ex

// Process-wide pointer to the device-resident step state (philox.hpp); every launch wrapper that has per-step scalars or
// random draws passes it to its kernel.  NULL = by-value arguments only (the default; eager execution needs nothing else).
static const void* g_step_state = nullptr;
extern "C" int uaps_set_step_state(const void* device_ptr) {
    if ((uintptr_t)device_ptr % 8) return UAPS_EINVAL;
    g_step_state = device_ptr;
    return UAPS_OK;
}
extern "C" const void* uaps_get_step_state(void) { return g_step_state; }

extern "C" const char* uaps_error_string(int code) {
    switch (code) {
        case UAPS_OK: return "ok";
        case UAPS_EINVAL: return "invalid argument (null pointer or non-positive dimension)";
        case UAPS_ERANGE: return "argument outside the supported range (heads 1..8, classes 2..8; conv kernels 1x1 / 3x3, dilation 1 / 2 / 4; the fused BatchNorm forms need W % 4 == 0 and 16-byte aligned tensors)";
        case UAPS_EWORKSPACE: return "workspace too small";
        case UAPS_ENOFORM: return "the kernel this layer runs on has no form for the requested hint (uaps_call_hints::dyt_*); nothing was launched";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

static size_t loss_ws_bytes(int D, int C) {
    const size_t ns = (size_t)(D + 2 * D * C + C + 2 * D);
    return (size_t)kMaxBlocks * ns * sizeof(float);
}

extern "C" int uaps_loss_workspace_bytes(int D, int B, int C, int H, int W, size_t* out) {
    if (!out) return UAPS_EINVAL;
    int rc = check_dims(D, B, C, H, W);
    if (rc) return rc;
    *out = loss_ws_bytes(D, C);
    return UAPS_OK;
}

extern "C" int uaps_unsup_fwd(const float* const* logits, const double* w, int D, int B, int C, int H, int W, float cw1,
                              float cw2, float eps, int64_t* pseudo, float* var, float* scalars, void* ws,
                              size_t ws_bytes, uaps_stream_t stream) {
    int rc = check_dims(D, B, C, H, W);
    if (rc) return rc;
    if (!logits || !w || !pseudo || !scalars || !ws) return UAPS_EINVAL;
    for (int k = 0; k < D; ++k) if (!logits[k]) return UAPS_EINVAL;
    if (ws_bytes < loss_ws_bytes(D, C)) return UAPS_EWORKSPACE;
    LossArgs a{}; a.logits = logits; a.w = w; a.D = D; a.B = B; a.C = C; a.H = H; a.W = W; a.cw1 = cw1; a.cw2 = cw2;
    a.eps = eps; a.pseudo = pseudo; a.var = var; a.scalars = scalars; a.partials = (float*)ws; a.stream = (hipStream_t)stream;
    return launch_unsup_fwd(a);
}

extern "C" int uaps_unsup_bwd(const float* const* logits, const int64_t* pseudo, const float* scalars, float cw1, float cw2,
                              const float* gscale, int D, int B, int C, int H, int W, float* const* dlogits,
                              uaps_stream_t stream) {
    int rc = check_dims(D, B, C, H, W);
    if (rc) return rc;
    if (!logits || !pseudo || !scalars || !dlogits) return UAPS_EINVAL;
    for (int k = 0; k < D; ++k) if (!logits[k] || !dlogits[k]) return UAPS_EINVAL;
    LossArgs a{}; a.logits = logits; a.dlogits = dlogits; a.D = D; a.B = B; a.C = C; a.H = H; a.W = W; a.cw1 = cw1; a.cw2 = cw2;
    a.labels = pseudo; a.cscalars = scalars; a.gscale = gscale; a.stream = (hipStream_t)stream;
    return launch_unsup_bwd(a);
}

extern "C" int uaps_sup_fwd(const float* const* logits, const int64_t* labels, int D, int B, int C, int H, int W, float ce_coef,
                            float dice_coef, float eps, float* scalars, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    int rc = check_dims(D, B, C, H, W);
    if (rc) return rc;
    if (!logits || !labels || !scalars || !ws) return UAPS_EINVAL;
    for (int k = 0; k < D; ++k) if (!logits[k]) return UAPS_EINVAL;
    if (ws_bytes < loss_ws_bytes(D, C)) return UAPS_EWORKSPACE;
    LossArgs a{}; a.logits = logits; a.D = D; a.B = B; a.C = C; a.H = H; a.W = W; a.eps = eps; a.labels = labels;
    a.cw1 = ce_coef; a.cw2 = dice_coef; a.scalars = scalars; a.partials = (float*)ws; a.stream = (hipStream_t)stream;
    return launch_sup_fwd(a);
}

extern "C" int uaps_sup_bwd(const float* const* logits, const int64_t* labels, const float* scalars, float ce_coef,
                            float dice_coef, const float* gscale, int D, int B, int C, int H, int W, float* const* dlogits, uaps_stream_t stream) {
    int rc = check_dims(D, B, C, H, W);
    if (rc) return rc;
    if (!logits || !labels || !scalars || !dlogits) return UAPS_EINVAL;
    for (int k = 0; k < D; ++k) if (!logits[k] || !dlogits[k]) return UAPS_EINVAL;
    LossArgs a{}; a.logits = logits; a.dlogits = dlogits; a.D = D; a.B = B; a.C = C; a.H = H; a.W = W; a.labels = labels;
    a.cw1 = ce_coef; a.cw2 = dice_coef; a.cscalars = scalars; a.gscale = gscale; a.stream = (hipStream_t)stream;
    return launch_sup_bwd(a);
}

// ---------------------------------------------------------------------------------------------
// The whole loss block of a step (supervised branch on the labelled logits + unsupervised branch on the unlabelled ones)
// as one forward launch + one finalize + one backward launch.
// ---------------------------------------------------------------------------------------------
static size_t pairloss_ws_bytes(int D, int C) {
    return (size_t)kMaxBlocks * (size_t)(sup_nsums(D, C) + unsup_nsums(D, C)) * sizeof(float);
}

extern "C" int uaps_pairloss_workspace_bytes(int D, int C, size_t* out) {
    if (!out) return UAPS_EINVAL;
    if (D < 1 || D > UAPS_MAX_HEADS || C < 2 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    *out = pairloss_ws_bytes(D, C);
    return UAPS_OK;
}

extern "C" int uaps_pairloss_num_sums(int D, int C, int* out) {
    if (!out) return UAPS_EINVAL;
    if (D < 1 || D > UAPS_MAX_HEADS || C < 2 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    *out = sup_nsums(D, C) + unsup_nsums(D, C);
    return UAPS_OK;
}

static int check_heads(const float* const* p, int D) {
    if (!p) return UAPS_EINVAL;
    for (int k = 0; k < D; ++k) if (!p[k]) return UAPS_EINVAL;
    return UAPS_OK;
}

extern "C" int uaps_pairloss_fwd(const float* const* lab_logits, const float* const* un_logits, const int64_t* labels, const double* w,
                                 int D, int B, int C, int H, int W, float cw1, float cw2, float eps, int64_t* pseudo, float* var,
                                 float* sup_scalars, float* unsup_scalars, double* sums_out, void* ws, size_t ws_bytes, int cfg,
                                 uaps_stream_t stream) {
    int rc = check_dims(D, B, C, H, W);
    if (rc) return rc;
    if ((rc = check_heads(lab_logits, D)) || (rc = check_heads(un_logits, D))) return rc;
    if (!labels || !pseudo || !ws || (!sums_out && (!sup_scalars || !unsup_scalars))) return UAPS_EINVAL;
    if (!w && !g_step_state) return UAPS_EINVAL;          // NULL weights = the step state's (uaps_set_step_state)
    if (ws_bytes < pairloss_ws_bytes(D, C)) return UAPS_EWORKSPACE;
    PairArgs a{}; a.lab = lab_logits; a.un = un_logits; a.w = w; a.D = D; a.B = B; a.C = C; a.H = H; a.W = W;
    a.ce_coef = 0.5f / D; a.dice_coef = 0.5f / D; a.cw1 = cw1; a.cw2 = cw2; a.eps = eps; a.labels = labels; a.pseudo = pseudo;
    a.var = var; a.sscal = sup_scalars; a.uscal = unsup_scalars; a.sums = sums_out; a.partials = (float*)ws; a.cfg = cfg;
    a.stream = (hipStream_t)stream;
    // per pixel, labelled 4DC + 8 (labels), unlabelled 4DC + 8 (pseudo-label written) + 4D when the variance maps are stored
    uaps::account_bytes((double)B * H * W * (2.0 * (4.0 * D * C + 8.0) + (var ? 4.0 * D : 0.0)));
    return launch_pair_fwd(a);
}

extern "C" int uaps_pairloss_finalize_sums(const double* sums, int D, int C, long n_pixels, float cw1, float cw2, float eps,
                                           float* sup_scalars, float* unsup_scalars, uaps_stream_t stream) {
    if (D < 1 || D > UAPS_MAX_HEADS || C < 2 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    if (!sums || !sup_scalars || !unsup_scalars || n_pixels <= 0) return UAPS_EINVAL;
    hipLaunchKernelGGL(pair_finalize_sums_kernel, dim3(1), dim3(kFinalizeThreads), 0, (hipStream_t)stream, sums, D, C, n_pixels, 0.5f / D,
                       0.5f / D, cw1, cw2, eps, sup_scalars, unsup_scalars, (const uint32_t*)g_step_state);
    return (int)hipGetLastError();
}

static int pairloss_bwd_impl(float* amax_out, const float* const* lab_logits, const float* const* un_logits, const int64_t* labels,
                             const int64_t* pseudo, const float* sup_scalars, const float* unsup_scalars, float cw1, float cw2,
                             const float* gscale, int D, int B, int C, int H, int W, long n_pixels_loss, float* const* dlab,
                             float* const* dun, int cfg, uaps_stream_t stream);
extern "C" int uaps_pairloss_bwd(const float* const* lab_logits, const float* const* un_logits, const int64_t* labels,
                                 const int64_t* pseudo, const float* sup_scalars, const float* unsup_scalars, float cw1, float cw2,
                                 const float* gscale, int D, int B, int C, int H, int W, long n_pixels_loss, float* const* dlab,
                                 float* const* dun, int cfg, uaps_stream_t stream) {
    return pairloss_bwd_impl(uaps::take_hints().out_amax, lab_logits, un_logits, labels, pseudo, sup_scalars, unsup_scalars, cw1, cw2, gscale,
                             D, B, C, H, W, n_pixels_loss, dlab, dun, cfg, stream);
}
// (the *_h form: the hints of THIS call -- out_amax: raise this bound to max|d logits| -- as the first argument, nothing thread-local)
extern "C" int uaps_pairloss_bwd_h(const uaps_call_hints* hints, const float* const* lab_logits, const float* const* un_logits,
                                   const int64_t* labels, const int64_t* pseudo, const float* sup_scalars, const float* unsup_scalars,
                                   float cw1, float cw2, const float* gscale, int D, int B, int C, int H, int W, long n_pixels_loss,
                                   float* const* dlab, float* const* dun, int cfg, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    return pairloss_bwd_impl(h.out_amax, lab_logits, un_logits, labels, pseudo, sup_scalars, unsup_scalars, cw1, cw2, gscale,
                             D, B, C, H, W, n_pixels_loss, dlab, dun, cfg, stream);
}
static int pairloss_bwd_impl(float* amax_out, const float* const* lab_logits, const float* const* un_logits, const int64_t* labels,
                             const int64_t* pseudo, const float* sup_scalars, const float* unsup_scalars, float cw1, float cw2,
                             const float* gscale, int D, int B, int C, int H, int W, long n_pixels_loss, float* const* dlab,
                             float* const* dun, int cfg, uaps_stream_t stream) {
    int rc = check_dims(D, B, C, H, W);
    if (rc) return rc;
    if ((rc = check_heads(lab_logits, D)) || (rc = check_heads(un_logits, D))) return rc;
    if (!dlab || !dun || !labels || !pseudo || !sup_scalars || !unsup_scalars) return UAPS_EINVAL;
    for (int k = 0; k < D; ++k) if (!dlab[k] || !dun[k]) return UAPS_EINVAL;
    PairArgs a{}; a.lab = lab_logits; a.un = un_logits; a.dlab = dlab; a.dun = dun; a.D = D; a.B = B; a.C = C; a.H = H; a.W = W;
    a.ce_coef = 0.5f / D; a.dice_coef = 0.5f / D; a.cw1 = cw1; a.cw2 = cw2; a.labels = labels; a.cpseudo = pseudo;
    a.sscal = const_cast<float*>(sup_scalars); a.uscal = const_cast<float*>(unsup_scalars);
    a.Nloss = n_pixels_loss > 0 ? n_pixels_loss : (long)B * H * W; a.gscale = gscale; a.cfg = cfg; a.stream = (hipStream_t)stream;
    a.amax_out = amax_out;
    uaps::account_bytes((double)B * H * W * 2.0 * (8.0 * D * C + 8.0));      // both branches: logits + labels read, gradients written
    return launch_pair_bwd(a);
}
