// Two-tensor softmax consistency terms for gfx950: the sibling formulations of the UAPS uncertainty/consistency
// block that the reference ships next to it and its comparison methods use on the same [B,C,H,W] logits --
//   utilities/losses_1.py:9-26   softmax_mse_loss   (softmax(a) - softmax(b))^2, elementwise
//   utilities/losses_1.py:29-48  softmax_kl_loss    F.kl_div(log_softmax(a), softmax(b), reduction='mean')
//   utilities/losses_1.py:139-149 entropy_minmization / entropy_map   -sum_c p log(p + 1e-6)
//   utilities/losses_2.py:201-213 kl_loss           F.kl_div(log(pr), gt, reduction='mean')
//   UAPS-Testing.ipynb cell 24   test-time uncertainty map  sum_c KLDivLoss('none')(log_softmax(main), softmax(aux))
// One streaming pass each: a lane owns 4 horizontally adjacent pixels and all C classes of both tensors in
// registers (NCHW: a class plane is contiguous over pixels -> 16-byte coalesced loads), the class softmax is
// register-local.  Sums are block partials + a fixed-order double finalize (deterministic, no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uaps_hip.h"
#include "hints.hpp"

namespace {
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;

template <int C> __device__ __forceinline__ void softmax_c(const float (&z)[C], float (&p)[C], float (&lp)[C]) {
    float mx = z[0];
#pragma unroll
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = __expf(z[c] - mx); s += p[c]; }
    const float inv = __builtin_amdgcn_rcpf(s), lse = mx + __logf(s);
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] *= inv; lp[c] = z[c] - lse; }
}

__device__ __forceinline__ float block_sum(float v) {
    __shared__ float red[kThreads / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// mode bits: 1 = write mse map [B,C,H,W], 2 = write per-pixel KL map [B,H,W], 4 = KL block partials.
// PROBS: the inputs are already probabilities (kl_loss of losses_2.py: input log(pr), target gt).
template <int C, bool PROBS>
__global__ __launch_bounds__(kThreads) void pair_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long HW, long N,
                                                            float* __restrict__ mse, float* __restrict__ klmap,
                                                            float* __restrict__ partials) {
    float acc = 0.f;
    for (long n = (long)blockIdx.x * kThreads + threadIdx.x; n < N; n += (long)gridDim.x * kThreads) {
        const long img = n / HW, hw = n - img * HW, base = img * C * HW + hw;
        float za[C], zb[C], p[C], lp[C], t[C], lt[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { za[c] = a[base + (long)c * HW]; zb[c] = b[base + (long)c * HW]; }
        if (PROBS) {
#pragma unroll
            for (int c = 0; c < C; ++c) { p[c] = za[c]; lp[c] = __logf(za[c]); t[c] = zb[c]; }
        } else {
            softmax_c<C>(za, p, lp);
            softmax_c<C>(zb, t, lt);
        }
        float kl = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (mse) { const float d = p[c] - t[c]; mse[base + (long)c * HW] = d * d; }
            kl += (t[c] > 0.f ? t[c] * __logf(t[c]) : 0.f) - t[c] * lp[c];      // xlogy(t,t) - t*log p  (torch kl_div)
        }
        if (klmap) klmap[n] = kl;
        acc += kl;
    }
    if (partials) {
        const float s = block_sum(acc);
        if (threadIdx.x == 0) partials[blockIdx.x] = s;
    }
}

// out[0] = sum(partials) / denom
__global__ __launch_bounds__(64) void pair_finalize(const float* __restrict__ partials, int n, double denom, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += (double)partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[0] = (float)(s / denom);
}

// Gradient w.r.t. the input logits a (the target b gets none, as in the reference):
//   MSE : da_j = 2 p_j [ G_j (p_j - t_j) - sum_c G_c (p_c - t_c) p_c ]         G = upstream gradient of the map
//   KL  : da_j = g (p_j - t_j) / numel                                           g = upstream scalar, 'mean' reduction
template <int C, bool MSE>
__global__ __launch_bounds__(kThreads) void pair_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ gmap, const float* __restrict__ gscalar,
                                                            float scale, long HW, long N, float* __restrict__ da) {
    const float g = MSE ? 1.f : (gscalar ? gscalar[0] : 1.f) * scale;
    for (long n = (long)blockIdx.x * kThreads + threadIdx.x; n < N; n += (long)gridDim.x * kThreads) {
        const long img = n / HW, hw = n - img * HW, base = img * C * HW + hw;
        float za[C], zb[C], p[C], lp[C], t[C], lt[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { za[c] = a[base + (long)c * HW]; zb[c] = b[base + (long)c * HW]; }
        softmax_c<C>(za, p, lp);
        softmax_c<C>(zb, t, lt);
        if (MSE) {
            float G[C], dot = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) { G[c] = gmap[base + (long)c * HW] * (p[c] - t[c]); dot += G[c] * p[c]; }
#pragma unroll
            for (int c = 0; c < C; ++c) da[base + (long)c * HW] = 2.f * p[c] * (G[c] - dot);
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) da[base + (long)c * HW] = g * (p[c] - t[c]);
        }
    }
}

// Gradient of the per-pixel KL map v = sum_c KLDivLoss('none')(log_softmax(a), softmax(b)) w.r.t. BOTH logit tensors
// (UCC's uncertainty, UCC/UCC_train.py:213-217: neither head is detached).  With p = softmax(a), q = softmax(b) and G the
// upstream gradient of the map:   da_j = G (p_j - q_j)      db_j = G q_j ((log q_j - log p_j) - v)
template <int C>
__global__ __launch_bounds__(kThreads) void pair_klmap_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                  const float* __restrict__ gmap, long HW, long N,
                                                                  float* __restrict__ da, float* __restrict__ db) {
    for (long n = (long)blockIdx.x * kThreads + threadIdx.x; n < N; n += (long)gridDim.x * kThreads) {
        const long img = n / HW, hw = n - img * HW, base = img * C * HW + hw;
        float za[C], zb[C], p[C], lp[C], q[C], lq[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { za[c] = a[base + (long)c * HW]; zb[c] = b[base + (long)c * HW]; }
        softmax_c<C>(za, p, lp);
        softmax_c<C>(zb, q, lq);
        const float g = gmap[n];
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) v += q[c] * (lq[c] - lp[c]);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (da) da[base + (long)c * HW] = g * (p[c] - q[c]);
            if (db) db[base + (long)c * HW] = g * q[c] * ((lq[c] - lp[c]) - v);
        }
    }
}

// ent[b,hw] = -sum_c p log(p + 1e-6)
template <int C>
__global__ __launch_bounds__(kThreads) void entropy_kernel(const float* __restrict__ p, long HW, long N, float* __restrict__ ent,
                                                           float* __restrict__ partials) {
    float acc = 0.f;
    for (long n = (long)blockIdx.x * kThreads + threadIdx.x; n < N; n += (long)gridDim.x * kThreads) {
        const long img = n / HW, hw = n - img * HW, base = img * C * HW + hw;
        float e = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { const float v = p[base + (long)c * HW]; e -= v * __logf(v + 1e-6f); }
        if (ent) ent[n] = e;
        acc += e;
    }
    if (partials) {
        const float s = block_sum(acc);
        if (threadIdx.x == 0) partials[blockIdx.x] = s;
    }
}

inline int grid_for(long n) { long g = (n + kThreads - 1) / kThreads; return (int)(g > kMaxBlocks ? kMaxBlocks : (g < 1 ? 1 : g)); }

template <int C> int run_fwd(bool probs, const float* a, const float* b, long HW, long N, float* mse, float* klmap, float* partials,
                             int grid, hipStream_t s) {
    if (probs) hipLaunchKernelGGL((pair_fwd_kernel<C, true>), dim3(grid), dim3(kThreads), 0, s, a, b, HW, N, mse, klmap, partials);
    else hipLaunchKernelGGL((pair_fwd_kernel<C, false>), dim3(grid), dim3(kThreads), 0, s, a, b, HW, N, mse, klmap, partials);
    return (int)hipGetLastError();
}
template <int C> int run_bwd(bool mse, const float* a, const float* b, const float* gmap, const float* gs, float scale, long HW, long N,
                             float* da, int grid, hipStream_t s) {
    if (mse) hipLaunchKernelGGL((pair_bwd_kernel<C, true>), dim3(grid), dim3(kThreads), 0, s, a, b, gmap, gs, scale, HW, N, da);
    else hipLaunchKernelGGL((pair_bwd_kernel<C, false>), dim3(grid), dim3(kThreads), 0, s, a, b, gmap, gs, scale, HW, N, da);
    return (int)hipGetLastError();
}
template <int C> int run_klmap_bwd(const float* a, const float* b, const float* gmap, long HW, long N, float* da, float* db, int grid,
                                   hipStream_t s) {
    hipLaunchKernelGGL((pair_klmap_bwd_kernel<C>), dim3(grid), dim3(kThreads), 0, s, a, b, gmap, HW, N, da, db);
    return (int)hipGetLastError();
}
template <int C> int run_ent(const float* p, long HW, long N, float* ent, float* partials, int grid, hipStream_t s) {
    hipLaunchKernelGGL((entropy_kernel<C>), dim3(grid), dim3(kThreads), 0, s, p, HW, N, ent, partials);
    return (int)hipGetLastError();
}
#define UAPS_BY_C(C, CALL) \
    switch (C) { case 2: return CALL(2); case 3: return CALL(3); case 4: return CALL(4); case 5: return CALL(5); \
                 case 6: return CALL(6); case 7: return CALL(7); case 8: return CALL(8); default: return UAPS_ERANGE; }
}  // namespace

extern "C" int uaps_pair_workspace_bytes(size_t* out) {
    if (!out) return UAPS_EINVAL;
    *out = (size_t)kMaxBlocks * sizeof(float);
    return UAPS_OK;
}

static int pair_fwd_dispatch(int C, bool probs, const float* a, const float* b, long HW, long N, float* mse, float* klmap, float* partials,
                             int grid, hipStream_t s) {
#define CALL(K) run_fwd<K>(probs, a, b, HW, N, mse, klmap, partials, grid, s)
    UAPS_BY_C(C, CALL)
#undef CALL
}

extern "C" int uaps_softmax_pair_fwd(const float* a, const float* b, int probs, int B, int C, int H, int W, float* mse_map,
                                     float* kl_map, float* kl_mean, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    if (!a || !b || B <= 0 || H <= 0 || W <= 0 || (!mse_map && !kl_map && !kl_mean)) return UAPS_EINVAL;
    if (C < 2 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    if (kl_mean && (!ws || ws_bytes < (size_t)kMaxBlocks * sizeof(float))) return UAPS_EWORKSPACE;
    const long HW = (long)H * W, N = (long)B * HW;
    const int grid = grid_for(N);
    hipStream_t s = (hipStream_t)stream;
    const int rc = pair_fwd_dispatch(C, probs != 0, a, b, HW, N, mse_map, kl_map, kl_mean ? (float*)ws : nullptr, grid, s);
    if (rc || !kl_mean) return rc;
    hipLaunchKernelGGL(pair_finalize, dim3(1), dim3(64), 0, s, (const float*)ws, grid, (double)N * C, kl_mean);   // reduction='mean': all elements
    return (int)hipGetLastError();
}

static int pair_bwd_dispatch(int C, bool mse, const float* a, const float* b, const float* gmap, const float* gs, float scale, long HW,
                             long N, float* da, int grid, hipStream_t s) {
#define CALL(K) run_bwd<K>(mse, a, b, gmap, gs, scale, HW, N, da, grid, s)
    UAPS_BY_C(C, CALL)
#undef CALL
}

extern "C" int uaps_softmax_mse_bwd(const float* a, const float* b, const float* grad_map, int B, int C, int H, int W, float* da,
                                    uaps_stream_t stream) {
    if (!a || !b || !grad_map || !da || B <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (C < 2 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    const long HW = (long)H * W, N = (long)B * HW;
    return pair_bwd_dispatch(C, true, a, b, grad_map, nullptr, 1.f, HW, N, da, grid_for(N), (hipStream_t)stream);
}

extern "C" int uaps_softmax_kl_bwd(const float* a, const float* b, const float* gscalar, int B, int C, int H, int W, float* da,
                                   uaps_stream_t stream) {
    if (!a || !b || !da || B <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (C < 2 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    const long HW = (long)H * W, N = (long)B * HW;
    return pair_bwd_dispatch(C, false, a, b, nullptr, gscalar, (float)(1.0 / ((double)N * C)), HW, N, da, grid_for(N), (hipStream_t)stream);
}

static int klmap_bwd_dispatch(int C, const float* a, const float* b, const float* gmap, long HW, long N, float* da, float* db, int grid,
                              hipStream_t s) {
#define CALL(K) run_klmap_bwd<K>(a, b, gmap, HW, N, da, db, grid, s)
    UAPS_BY_C(C, CALL)
#undef CALL
}

extern "C" int uaps_softmax_klmap_bwd(const float* a, const float* b, const float* grad_map, int B, int C, int H, int W, float* da,
                                      float* db, uaps_stream_t stream) {
    if (!a || !b || !grad_map || (!da && !db) || B <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (C < 2 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    const long HW = (long)H * W, N = (long)B * HW;
    return klmap_bwd_dispatch(C, a, b, grad_map, HW, N, da, db, grid_for(N), (hipStream_t)stream);
}

static int ent_dispatch(int C, const float* p, long HW, long N, float* ent, float* partials, int grid, hipStream_t s) {
#define CALL(K) run_ent<K>(p, HW, N, ent, partials, grid, s)
    UAPS_BY_C(C, CALL)
#undef CALL
}

extern "C" int uaps_entropy_map(const float* p, int B, int C, int H, int W, float* ent_map, float* ent_mean, void* ws, size_t ws_bytes,
                                uaps_stream_t stream) {
    if (!p || B <= 0 || H <= 0 || W <= 0 || (!ent_map && !ent_mean)) return UAPS_EINVAL;
    if (C < 2 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    if (ent_mean && (!ws || ws_bytes < (size_t)kMaxBlocks * sizeof(float))) return UAPS_EWORKSPACE;
    const long HW = (long)H * W, N = (long)B * HW;
    const int grid = grid_for(N);
    hipStream_t s = (hipStream_t)stream;
    const int rc = ent_dispatch(C, p, HW, N, ent_map, ent_mean ? (float*)ws : nullptr, grid, s);
    if (rc || !ent_mean) return rc;
    hipLaunchKernelGGL(pair_finalize, dim3(1), dim3(64), 0, s, (const float*)ws, grid, (double)N, ent_mean);
    return (int)hipGetLastError();
}

// ---- residual join of the ResNet blocks (utilities/resnet.py:47-50, 88-91): out = relu(a + b), and its backward
// da = db = dout * (out > 0), each one 16-byte-per-lane streaming pass -------------------------------------------------
namespace {
// amax: optional bound (uaps_call_hints::out_amax) raised to max(out) -- the join's output feeds the next block's convolutions
__global__ __launch_bounds__(kThreads) void add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ out, long n4, long n, float* __restrict__ amax) {
    __shared__ float s16[16];
    float m = 0.f;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kThreads) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        const float4 r = make_float4(fmaxf(x.x + y.x, 0.f), fmaxf(x.y + y.y, 0.f), fmaxf(x.z + y.z, 0.f), fmaxf(x.w + y.w, 0.f));
        reinterpret_cast<float4*>(out)[i] = r;
        m = fmaxf(m, fmaxf(fmaxf(r.x, r.y), fmaxf(r.z, r.w)));
    }
    if (blockIdx.x == 0)
        for (long i = 4 * n4 + threadIdx.x; i < n; i += kThreads) { const float r = fmaxf(a[i] + b[i], 0.f); out[i] = r; m = fmaxf(m, r); }
    if (amax != nullptr) uaps::block_amax_to(amax, m, s16);
}
__global__ __launch_bounds__(kThreads) void relu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                            float* __restrict__ dx, long n4, long n) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kThreads) {
        const float4 g = reinterpret_cast<const float4*>(dout)[i], o = reinterpret_cast<const float4*>(out)[i];
        reinterpret_cast<float4*>(dx)[i] = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
    }
    if (blockIdx.x == 0)
        for (long i = 4 * n4 + threadIdx.x; i < n; i += kThreads) dx[i] = out[i] > 0.f ? dout[i] : 0.f;
}
// the join's output feeds K consumers (the next block's first convolution and its shortcut, a decoder): their K gradients are
// summed here, in argument order, instead of by K - 1 separate accumulation passes of the autograd engine
constexpr int kReluBwdMax = 4;
struct ReluBwdIn { const float* g[kReluBwdMax]; int k; };
__global__ __launch_bounds__(kThreads) void relu_bwd_sum_kernel(ReluBwdIn in, const float* __restrict__ out, float* __restrict__ dx, long n4, long n) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kThreads) {
        float4 g = reinterpret_cast<const float4*>(in.g[0])[i];
#pragma unroll
        for (int k = 1; k < kReluBwdMax; ++k)
            if (k < in.k) { const float4 t = reinterpret_cast<const float4*>(in.g[k])[i]; g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
        const float4 o = reinterpret_cast<const float4*>(out)[i];
        reinterpret_cast<float4*>(dx)[i] = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
    }
    if (blockIdx.x == 0)
        for (long i = 4 * n4 + threadIdx.x; i < n; i += kThreads) {
            float g = in.g[0][i];
            for (int k = 1; k < in.k; ++k) g += in.g[k][i];
            dx[i] = out[i] > 0.f ? g : 0.f;
        }
}
// out = [a ; b] along the batch axis (two tensors of n elements each), max|out| raised into the bound `amax` on the way
__global__ __launch_bounds__(kThreads) void cat2_amax_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                             long n4, long n, float* __restrict__ amax) {
    __shared__ float s16[16];
    float m = 0.f;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < 2 * n4; i += (long)gridDim.x * kThreads) {
        const bool hi = i >= n4;
        const float4 v = hi ? reinterpret_cast<const float4*>(b)[i - n4] : reinterpret_cast<const float4*>(a)[i];
        reinterpret_cast<float4*>(out + (hi ? n : 0))[hi ? i - n4 : i] = v;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0)
        for (long i = 4 * n4 + threadIdx.x; i < n; i += kThreads) {
            const float x = a[i], y = b[i];
            out[i] = x; out[n + i] = y;
            m = fmaxf(m, fmaxf(fabsf(x), fabsf(y)));
        }
    if (amax != nullptr) uaps::block_amax_to(amax, m, s16);
}
inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

// The two batches of a training step as one (UAPS_train.py:177 + :185 run as one pass, unet.UNet_UAPS.forward_pair): out [2n] =
// a [n] followed by b [n]; with uaps_call_hints::out_amax the bound of the result is raised to max|out| on the way, so the first
// convolution's weight gradient can take the fp16 form.
static int cat2_impl(float* amax, const float* a, const float* b, float* out, long n, uaps_stream_t stream);
extern "C" int uaps_cat2(const float* a, const float* b, float* out, long n, uaps_stream_t stream) {
    return cat2_impl(uaps::take_hints().out_amax, a, b, out, n, stream);
}
// (the *_h forms: the hints of THIS call as the first argument, nothing thread-local -- see conv_fwd.hip)
extern "C" int uaps_cat2_h(const uaps_call_hints* hints, const float* a, const float* b, float* out, long n, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    return cat2_impl(h.out_amax, a, b, out, n, stream);
}
static int cat2_impl(float* amax, const float* a, const float* b, float* out, long n, uaps_stream_t stream) {
    if (!a || !b || !out || n <= 0) return UAPS_EINVAL;
    uaps::account_bytes(16.0 * n);                        // both halves read, the joined batch written
    const long n4 = (al16p(a) && al16p(b) && al16p(out) && n % 4 == 0) ? n / 4 : 0;
    hipLaunchKernelGGL(cat2_amax_kernel, dim3(grid_for(n4 > 0 ? 2 * n4 : 1)), dim3(kThreads), 0, (hipStream_t)stream, a, b, out, n4, n, amax);
    return (int)hipGetLastError();
}

extern "C" int uaps_relu_bwd_sum(const float* const* dout_host, int k, const float* out, float* dx, long n, uaps_stream_t stream) {
    if (!dout_host || k < 1 || k > kReluBwdMax || !out || !dx || n <= 0) return UAPS_EINVAL;
    ReluBwdIn in{};
    in.k = k;
    bool al = al16p(out) && al16p(dx);
    for (int i = 0; i < k; ++i) { if (!dout_host[i]) return UAPS_EINVAL; in.g[i] = dout_host[i]; al = al && al16p(dout_host[i]); }
    const long n4 = al ? n / 4 : 0;
    hipLaunchKernelGGL(relu_bwd_sum_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kThreads), 0, (hipStream_t)stream, in, out, dx, n4, n);
    return (int)hipGetLastError();
}

static int add_relu_impl(float* amax, const float* a, const float* b, float* out, long n, uaps_stream_t stream);
extern "C" int uaps_add_relu(const float* a, const float* b, float* out, long n, uaps_stream_t stream) {
    return add_relu_impl(uaps::take_hints().out_amax, a, b, out, n, stream);
}
extern "C" int uaps_add_relu_h(const uaps_call_hints* hints, const float* a, const float* b, float* out, long n, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    return add_relu_impl(h.out_amax, a, b, out, n, stream);
}
static int add_relu_impl(float* amax, const float* a, const float* b, float* out, long n, uaps_stream_t stream) {
    if (!a || !b || !out || n <= 0) return UAPS_EINVAL;
    const long n4 = (al16p(a) && al16p(b) && al16p(out)) ? n / 4 : 0;
    hipLaunchKernelGGL(add_relu_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kThreads), 0, (hipStream_t)stream, a, b, out, n4, n, amax);
    return (int)hipGetLastError();
}

extern "C" int uaps_relu_bwd(const float* dout, const float* out, float* dx, long n, uaps_stream_t stream) {
    if (!dout || !out || !dx || n <= 0) return UAPS_EINVAL;
    const long n4 = (al16p(dout) && al16p(out) && al16p(dx)) ? n / 4 : 0;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kThreads), 0, (hipStream_t)stream, dout, out, dx, n4, n);
    return (int)hipGetLastError();
}
