// Fused BatchNorm2d(train) + LeakyReLU + Dropout, forward and backward, for gfx950.
//
// Replaces, per ConvBlock half (utilities/UAPS_unet.py:36-44): nn.BatchNorm2d -> nn.LeakyReLU ->
// nn.Dropout on the conv output, i.e. MIOpenBatchNormFwdTrainSpatial + a leaky_relu kernel + a
// fused_dropout kernel (forward) and their three backward kernels.  MIOpen's spatial BN launches
// one workgroup per channel, so the 16-channel full-resolution layers of this U-Net run on 16 of
// the chip's 256 CUs (measured 420 us per call, profiles/r01_baseline_miopen_kernel_stats.csv);
// here every (sample, channel) plane is cut into chunks, so all CUs stream:
//   forward : stats pass (1 read) -> per-channel finalize -> apply pass (1 read, 1 write)
//   backward: sums pass (2 reads) -> per-channel finalize -> dx pass (2 reads, 1 write)
// Reductions are block partials + fixed-order double finalize: bitwise reproducible, no float atomics.
// The conv bias (which train-mode BN cancels exactly) is folded in: it only shifts running_mean.
// Dropout masks come from Philox (philox.hpp) and are regenerated in the backward, never stored.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uaps_hip.h"
#include "philox.hpp"
#include "hints.hpp"

namespace {
using uaps::philox4x32_10; using uaps::u01; using uaps::U4; using uaps::pick;

constexpr int kThreads = 256;
constexpr int kChunk = 4096;       // floats per block: 4 float4 per thread

inline int nchunks_for(long HW) { return (int)((HW + kChunk - 1) / kChunk); }
inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int kMaxGroups = 8;
struct BnWs { float2* partials; float* coef; float* maxes; };   // coef: [groups][4][C] floats; maxes: two bounds (uaps_bn_act_bwd_prepare)
inline size_t bn_ws_bytes(int B, int C, long HW) {
    return align256((size_t)C * B * nchunks_for(HW) * sizeof(float2)) + align256((size_t)4 * C * kMaxGroups * sizeof(float)) +
           (size_t)2 * UAPS_BOUND_FLOATS * sizeof(float);
}
inline BnWs carve(void* ws, int B, int C, long HW) {
    BnWs w; w.partials = (float2*)ws;
    w.coef = (float*)((char*)ws + align256((size_t)C * B * nchunks_for(HW) * sizeof(float2)));
    w.maxes = (float*)((char*)w.coef + align256((size_t)4 * C * kMaxGroups * sizeof(float)));
    return w;
}

__device__ __forceinline__ float2 block_sum2(float a, float b) {
    __shared__ float2 red[kThreads / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(a, b);
    __syncthreads();
    float2 r = red[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) { r.x += red[w].x; r.y += red[w].y; }
    return r;
}

// ---- forward pass 1: per-chunk (sum, sum of squares) ------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const float* __restrict__ y, int C, long HW, int nchunks,
                                                            float2* __restrict__ partials, const float* __restrict__ shift_mean,
                                                            const float* __restrict__ shift_bias) {
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int b = plane / C, c = plane - b * C;
    const float* p = y + (long)plane * HW;
    const long lo = (long)chunk * kChunk, hi = min(HW, lo + (long)kChunk);
    // sums about the channel's shift (running mean minus conv bias): no cancellation in E[d^2] - E[d]^2 when |mean| >> std
    const float sh = (shift_mean ? shift_mean[c] : 0.f) - (shift_bias ? shift_bias[c] : 0.f);
    float s = 0.f, ss = 0.f;
    if (VEC) {
        for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * kThreads) {
            float4 v = *reinterpret_cast<const float4*>(p + i);
            v.x -= sh; v.y -= sh; v.z -= sh; v.w -= sh;
            s += (v.x + v.y) + (v.z + v.w);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += kThreads) { const float v = p[i] - sh; s += v; ss += v * v; }
    }
    const float2 r = block_sum2(s, ss);
    if (threadIdx.x == 0) partials[((long)c * gridDim.y / C + b) * nchunks + chunk] = r;
}

// ---- forward finalize: one wave per channel, looping over the G = B / Bg statistics groups in order -------
// (a group = the images of one reference forward call: the labelled and the unlabelled batch are normalised
// separately, UAPS_train.py:177,185, and update the running statistics one after the other)
// blockDim = kThreads * GP: GP statistics groups are reduced side by side (one 256-thread slice each), then thread 0
// finishes them in group order (the running statistics are updated group after group, like the reference's successive
// forward calls); groups beyond GP take further rounds.
__global__ __launch_bounds__(1024) void bn_finalize_fwd(const float2* __restrict__ partials, int B, int Bg, int nch, double HW,
                                                        const float* __restrict__ conv_bias, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                        float momentum, float eps, float* __restrict__ save_mean,
                                                        float* __restrict__ save_invstd, float* __restrict__ coef, int C,
                                                        float2* __restrict__ xf, const float* __restrict__ shift_mean,
                                                        const float* __restrict__ shift_bias) {
    __shared__ double red[4][2][kThreads / 64];
    const int c = blockIdx.x, GP = blockDim.x / kThreads;
    // the shift the partial sums were formed about; read by every thread before the first barrier, i.e. before thread 0
    // updates running_mean (shift_mean may be running_mean itself)
    const double sh = (double)(shift_mean ? shift_mean[c] : 0.f) - (double)(shift_bias ? shift_bias[c] : 0.f);
    const int slice = threadIdx.x / kThreads, t = threadIdx.x % kThreads, lane = t & 63, wave = t >> 6;
    const int G = B / Bg, nparts = Bg * nch;      // nparts reaches a few thousand when the partials come per conv tile
    const double M = (double)Bg * HW;
    for (int g0 = 0; g0 < G; g0 += GP) {
        const int g = g0 + slice;
        double s = 0.0, ss = 0.0;
        if (g < G) {
            const float2* pp = partials + ((long)c * B + (long)g * Bg) * nch;
            // 8 independent loads per trip (clamped index, masked add: the order of the additions is that of the plain loop);
            // the kernel is one short block per channel and all memory latency
            for (int i0 = t; i0 < nparts; i0 += 8 * kThreads) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * kThreads; v[u] = pp[i < nparts ? i : nparts - 1]; }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + u * kThreads < nparts) { s += v[u].x; ss += v[u].y; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
        __syncthreads();                           // red[] of the previous round has been consumed
        if (lane == 0) { red[slice][0][wave] = s; red[slice][1][wave] = ss; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int q = 0; q < GP && g0 + q < G; ++q) {
                const int gg = g0 + q;
                s = (red[q][0][0] + red[q][0][1]) + (red[q][0][2] + red[q][0][3]);
                ss = (red[q][1][0] + red[q][1][1]) + (red[q][1][2] + red[q][1][3]);
                const double dm = s / M;
                double var = ss / M - dm * dm;
                if (var < 0.0) var = 0.0;
                const double mean = sh + dm;
                const float invstd = (float)(1.0 / sqrt(var + (double)eps));
                save_mean[gg * C + c] = (float)mean;
                save_invstd[gg * C + c] = invstd;
                if (coef) {
                    coef[(gg * 4 + 0) * C + c] = gamma[c] * invstd;   // scale
                    coef[(gg * 4 + 1) * C + c] = beta[c];             // shift applied after (y - mean) * scale
                }
                if (xf) {                                             // for the convs that apply it as fma(y, scale, shift)
                    const float sc = gamma[c] * invstd;
                    xf[gg * C + c] = make_float2(sc, beta[c] - (float)mean * sc);
                }
                if (running_mean) {
                    const double bias = conv_bias ? (double)conv_bias[c] : 0.0;
                    const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
                    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * (mean + bias));
                    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
                }
            }
        }
    }
    if (nbt && c == 0 && threadIdx.x == 0) nbt[0] += G;
}

__global__ __launch_bounds__(64) void bn_coef_eval(const float* __restrict__ conv_bias, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ running_mean,
                                                   const float* __restrict__ running_var, float eps, float* __restrict__ coef,
                                                   float* __restrict__ mean_out, int C) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const float invstd = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
    coef[c] = gamma[c] * invstd;
    coef[C + c] = beta[c];
    mean_out[c] = running_mean[c] - (conv_bias ? conv_bias[c] : 0.f);     // (y + b - rm) = y - (rm - b)
}

// ---- forward pass 2: out = dropout(leaky_relu((y - mean) * scale + beta)) ---------------------------------
// res != nullptr (uaps_call_hints::residual, the residual joins of utilities/resnet.py:47-50, 88-91): out = relu(bn(y) + res), no
// slope, no dropout; amax != nullptr: the bound is raised to max|out|.
template <bool VEC, bool DROP>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const float* __restrict__ y, float* __restrict__ out, int C, long HW,
                                                            const float* __restrict__ mean, const float* __restrict__ coef,
                                                            float slope, float drop_p, float drop_scale, uint64_t seed_in,
                                                            uint64_t offset, int Bg, const uint32_t* __restrict__ st,
                                                            const float* __restrict__ res, float* __restrict__ amax) {
    const uint64_t seed = uaps::step_key(seed_in, st);
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int b = plane / C, c = plane - b * C, g = b / Bg;
    const float* cf = coef + (long)g * 4 * C;
    const float mu = mean[g * C + c], sc = cf[c], sh = cf[C + c];
    const long pbase = (long)plane * HW;
    const long lo = (long)chunk * kChunk, hi = min(HW, lo + (long)kChunk);
    float m = 0.f;
    if (VEC) {
        for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * kThreads) {
            const float4 v = *reinterpret_cast<const float4*>(y + pbase + i);
            float z[4] = {(v.x - mu) * sc + sh, (v.y - mu) * sc + sh, (v.z - mu) * sc + sh, (v.w - mu) * sc + sh};
            if (res != nullptr) {                        // uniform branch
                const float4 r = *reinterpret_cast<const float4*>(res + pbase + i);
                z[0] = fmaxf(z[0] + r.x, 0.f); z[1] = fmaxf(z[1] + r.y, 0.f); z[2] = fmaxf(z[2] + r.z, 0.f); z[3] = fmaxf(z[3] + r.w, 0.f);
            } else if (DROP) {
                const U4 r = philox4x32_10(offset + (uint64_t)((pbase + i) >> 2), seed);
#pragma unroll
                for (int k = 0; k < 4; ++k) { z[k] = z[k] > 0.f ? z[k] : z[k] * slope; z[k] = u01(pick(r, k)) >= drop_p ? z[k] * drop_scale : 0.f; }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) z[k] = z[k] > 0.f ? z[k] : z[k] * slope;
            }
            *reinterpret_cast<float4*>(out + pbase + i) = make_float4(z[0], z[1], z[2], z[3]);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(z[0]), fabsf(z[1])), fmaxf(fabsf(z[2]), fabsf(z[3]))));
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += kThreads) {
            float z = (y[pbase + i] - mu) * sc + sh;
            if (res != nullptr) {
                z = fmaxf(z + res[pbase + i], 0.f);
            } else {
                z = z > 0.f ? z : z * slope;
                if (DROP) {
                    const long e = pbase + i;
                    const U4 r = philox4x32_10(offset + (uint64_t)(e >> 2), seed);
                    z = u01(pick(r, (int)(e & 3))) >= drop_p ? z * drop_scale : 0.f;
                }
            }
            out[pbase + i] = z;
            m = fmaxf(m, fabsf(z));
        }
    }
    if (amax != nullptr) {                               // uniform branch
        __shared__ float sm[16];
        uaps::block_amax_to(amax, m, sm);
    }
}

// ---- backward helpers -------------------------------------------------------------------------------------
// d(pre-activation) from d(out): dropout mask/scale, then LeakyReLU slope chosen by the sign of the
// recomputed BN output z (torch: grad * (z > 0 ? 1 : slope)).
__device__ __forceinline__ float dpre(float dout, float yv, float mu, float sc, float sh, float slope) {
    const float z = (yv - mu) * sc + sh;
    return z > 0.f ? dout : dout * slope;
}

template <bool VEC, bool DROP>
__global__ __launch_bounds__(kThreads) void bn_bwd_sums_kernel(const float* __restrict__ dout, const float* __restrict__ y, int C,
                                                               long HW, int nchunks, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float slope, float drop_p,
                                                               float drop_scale, uint64_t seed_in, uint64_t offset,
                                                               float2* __restrict__ partials, int Bg, const uint32_t* __restrict__ st) {
    const uint64_t seed = uaps::step_key(seed_in, st);
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int b = plane / C, c = plane - b * C, g = b / Bg;
    const float mu = mean[g * C + c], is = invstd[g * C + c], sc = gamma[c] * is, sh = beta[c];
    const long pbase = (long)plane * HW;
    const long lo = (long)chunk * kChunk, hi = min(HW, lo + (long)kChunk);
    float s1 = 0.f, s2 = 0.f;
    if (VEC) {
        for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * kThreads) {
            const float4 g4 = *reinterpret_cast<const float4*>(dout + pbase + i);
            const float4 y4 = *reinterpret_cast<const float4*>(y + pbase + i);
            float g[4] = {g4.x, g4.y, g4.z, g4.w};
            const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
            if (DROP) {
                const U4 r = philox4x32_10(offset + (uint64_t)((pbase + i) >> 2), seed);
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = u01(pick(r, k)) >= drop_p ? g[k] * drop_scale : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = dpre(g[k], yy[k], mu, sc, sh, slope);
                s1 += d; s2 += d * ((yy[k] - mu) * is);
            }
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += kThreads) {
            float g = dout[pbase + i];
            const float yv = y[pbase + i];
            if (DROP) {
                const long e = pbase + i;
                const U4 r = philox4x32_10(offset + (uint64_t)(e >> 2), seed);
                g = u01(pick(r, (int)(e & 3))) >= drop_p ? g * drop_scale : 0.f;
            }
            const float d = dpre(g, yv, mu, sc, sh, slope);
            s1 += d; s2 += d * ((yv - mu) * is);
        }
    }
    const float2 r = block_sum2(s1, s2);
    if (threadIdx.x == 0) partials[((long)c * gridDim.y / C + b) * nchunks + chunk] = r;
}

// dy = gamma * invstd * (dpre - mean(dpre) - xhat * mean(dpre * xhat))
template <bool VEC, bool DROP>
__global__ __launch_bounds__(kThreads) void bn_bwd_dx_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                             float* __restrict__ dy, int C, long HW, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float2* __restrict__ partials,
                                                             int nch_p, int B, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dconv_bias,
                                                             float slope, float drop_p, float drop_scale, uint64_t seed_in,
                                                             uint64_t offset, int Bg, const uint32_t* __restrict__ st,
                                                             float* __restrict__ amax_out) {
    const uint64_t seed = uaps::step_key(seed_in, st);
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int b = plane / C, c = plane - b * C, g = b / Bg;
    float amax = 0.f;                            // max|dy| of this thread's elements (uaps_call_hints::out_amax)
    // The reduction's finalize runs here, not as a launch of its own: the first wave sums this channel's per-block
    // partials of the block's statistics group (a few hundred float2, L2-resident) in a fixed order -- every block of
    // the group computes the same two means bit for bit -- and the block of image 0, chunk 0 also adds up all groups
    // for dgamma / dbeta (and writes the zero gradient of the conv bias in front of the BatchNorm).
    __shared__ float sk[2];
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, nparts = Bg * nch_p;
        const double M = (double)Bg * (double)HW;
        const bool owner = b == 0 && chunk == 0;
        const int G = B / Bg;
        double t1 = 0.0, t2 = 0.0;
        for (int gg = owner ? 0 : g; gg < (owner ? G : g + 1); ++gg) {
            const float2* pp = partials + ((long)c * B + (long)gg * Bg) * nch_p;
            double s1 = 0.0, s2 = 0.0;
            for (int i = lane; i < nparts; i += 64) { const float2 v = pp[i]; s1 += v.x; s2 += v.y; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            if (gg == g && lane == 0) { sk[0] = (float)(s1 / M); sk[1] = (float)(s2 / M); }
            t1 += s1; t2 += s2;
        }
        if (owner && lane == 0) {
            dbeta[c] = (float)t1; dgamma[c] = (float)t2;
            if (dconv_bias) dconv_bias[c] = 0.f;     // a bias in front of a train-mode BatchNorm cancels: d(bias) = sum of dy = 0
        }
    }
    __syncthreads();
    const float mu = mean[g * C + c], is = invstd[g * C + c], sc = gamma[c] * is, sh = beta[c], k2 = sk[0], k3 = sk[1];
    const long pbase = (long)plane * HW;
    const long lo = (long)chunk * kChunk, hi = min(HW, lo + (long)kChunk);
    if (VEC) {
        for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * kThreads) {
            const float4 g4 = *reinterpret_cast<const float4*>(dout + pbase + i);
            const float4 y4 = *reinterpret_cast<const float4*>(y + pbase + i);
            float g[4] = {g4.x, g4.y, g4.z, g4.w};
            const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
            if (DROP) {
                const U4 r = philox4x32_10(offset + (uint64_t)((pbase + i) >> 2), seed);
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = u01(pick(r, k)) >= drop_p ? g[k] * drop_scale : 0.f;
            }
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = dpre(g[k], yy[k], mu, sc, sh, slope);
                o[k] = sc * (d - k2 - ((yy[k] - mu) * is) * k3);
            }
            *reinterpret_cast<float4*>(dy + pbase + i) = make_float4(o[0], o[1], o[2], o[3]);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += kThreads) {
            float g = dout[pbase + i];
            const float yv = y[pbase + i];
            if (DROP) {
                const long e = pbase + i;
                const U4 r = philox4x32_10(offset + (uint64_t)(e >> 2), seed);
                g = u01(pick(r, (int)(e & 3))) >= drop_p ? g * drop_scale : 0.f;
            }
            const float d = dpre(g, yv, mu, sc, sh, slope);
            const float o = sc * (d - k2 - ((yv - mu) * is) * k3);
            dy[pbase + i] = o;
            amax = fmaxf(amax, fabsf(o));
        }
    }
    if (amax_out) {                              // uniform branch
        __shared__ float sm[16];
        uaps::block_amax_to(amax_out, amax, sm);
    }
}

// ---- the backward in two halves (round 4): the reductions now, dy where it is consumed -----------------------------------
// uaps_bn_act_bwd_prepare = this sums pass (no dropout) + bn_bwd_finalize_kernel; the weight-gradient kernel of the convolution in
// front of the BatchNorm then forms dy = sc (d - k2 - x_hat k3) while it stages its dy operand and writes it through for the
// input-gradient kernel (uaps_call_hints::dyt_*), or bn_bwd_dx_coef_kernel does (uaps_bn_act_bwd_apply) where no such kernel runs.
// The sums pass also raises two bounds, max|d| and max|x_hat|, from which the finalize derives an upper bound of |dy| -- the
// fp16 form of the weight gradient needs its operand's bound before the operand exists.
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_sums_max_kernel(const float* __restrict__ dout, const float* __restrict__ y, int C,
                                                                   long HW, int nchunks, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float slope,
                                                                   float2* __restrict__ partials, int Bg, int B,
                                                                   float* __restrict__ gmax, float* __restrict__ xmax) {
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int b = plane / C, c = plane - b * C, g = b / Bg;
    const float mu = mean[g * C + c], is = invstd[g * C + c], sc = gamma[c] * is, sh = beta[c];
    const long pbase = (long)plane * HW;
    const long lo = (long)chunk * kChunk, hi = min(HW, lo + (long)kChunk);
    float s1 = 0.f, s2 = 0.f, md = 0.f, mx = 0.f;
    if (VEC) {
        for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * kThreads) {
            const float4 g4 = *reinterpret_cast<const float4*>(dout + pbase + i);
            const float4 y4 = *reinterpret_cast<const float4*>(y + pbase + i);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
            const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = dpre(gg[k], yy[k], mu, sc, sh, slope), xh = (yy[k] - mu) * is;
                s1 += d; s2 += d * xh;
                md = fmaxf(md, fabsf(d)); mx = fmaxf(mx, fabsf(xh));
            }
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += kThreads) {
            const float yv = y[pbase + i];
            const float d = dpre(dout[pbase + i], yv, mu, sc, sh, slope), xh = (yv - mu) * is;
            s1 += d; s2 += d * xh;
            md = fmaxf(md, fabsf(d)); mx = fmaxf(mx, fabsf(xh));
        }
    }
    const float2 r = block_sum2(s1, s2);
    if (threadIdx.x == 0) partials[((long)c * B + b) * nchunks + chunk] = r;
    __shared__ float sm[2][16];
    uaps::block_amax_to(gmax, md, sm[0]);
    uaps::block_amax_to(xmax, mx, sm[1]);
}

// one wave per channel: the two means of every statistics group (the arithmetic of bn_bwd_dx_kernel's first wave, bit for bit),
// coef [groups][C][8] = (mean, invstd, gamma invstd, beta, mean(d), mean(d x_hat), 0, 0), dgamma / dbeta over all groups, the zero
// gradient of a conv bias in front, and |dy| <= |gamma| invstd (max|d| + |k2| + max|x_hat| |k3|) raised into dy_bound
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float2* __restrict__ partials, int nch_p, int B, int Bg, int C, long HW,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ gmax, const float* __restrict__ xmax,
                                                             float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ dconv_bias, float* __restrict__ dy_bound) {
    const int c = blockIdx.x, lane = threadIdx.x, nparts = Bg * nch_p, G = B / Bg;
    const double M = (double)Bg * (double)HW;
    const float Gm = uaps::bound_max(gmax), Xm = uaps::bound_max(xmax);
    double t1 = 0.0, t2 = 0.0;
    float bnd = 0.f;
    for (int gg = 0; gg < G; ++gg) {
        const float2* pp = partials + ((long)c * B + (long)gg * Bg) * nch_p;
        double s1 = 0.0, s2 = 0.0;
        for (int i = lane; i < nparts; i += 64) { const float2 v = pp[i]; s1 += v.x; s2 += v.y; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        const float k2 = (float)(s1 / M), k3 = (float)(s2 / M);
        const float mu = mean[gg * C + c], is = invstd[gg * C + c], sc = gamma[c] * is;
        if (lane == 0) {
            float* q = coef + ((long)gg * C + c) * 8;
            q[0] = mu; q[1] = is; q[2] = sc; q[3] = beta[c]; q[4] = k2; q[5] = k3; q[6] = 0.f; q[7] = 0.f;
        }
        bnd = fmaxf(bnd, fabsf(sc) * (Gm + fabsf(k2) + Xm * fabsf(k3)));
        t1 += s1; t2 += s2;
    }
    if (lane == 0) {
        dbeta[c] = (float)t1; dgamma[c] = (float)t2;
        if (dconv_bias) dconv_bias[c] = 0.f;
        if (dy_bound && bnd == bnd) {
            unsigned* slot = reinterpret_cast<unsigned*>(dy_bound) + (c % UAPS_BOUND_SLOTS) * UAPS_BOUND_STRIDE;
            atomicMax(slot, __builtin_bit_cast(unsigned, bnd));
        }
    }
}

// dy from the coefficients of bn_bwd_finalize_kernel (the arithmetic of bn_bwd_dx_kernel)
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_dx_coef_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                                  float* __restrict__ dy, int C, long HW, const float* __restrict__ coef,
                                                                  float slope, int Bg, float* __restrict__ amax_out) {
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int b = plane / C, c = plane - b * C, g = b / Bg;
    const float* q = coef + ((long)g * C + c) * 8;
    const float mu = q[0], is = q[1], sc = q[2], sh = q[3], k2 = q[4], k3 = q[5];
    const long pbase = (long)plane * HW;
    const long lo = (long)chunk * kChunk, hi = min(HW, lo + (long)kChunk);
    float amax = 0.f;
    if (VEC) {
        for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * kThreads) {
            const float4 g4 = *reinterpret_cast<const float4*>(dout + pbase + i);
            const float4 y4 = *reinterpret_cast<const float4*>(y + pbase + i);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
            const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = dpre(gg[k], yy[k], mu, sc, sh, slope);
                o[k] = sc * (d - k2 - ((yy[k] - mu) * is) * k3);
            }
            *reinterpret_cast<float4*>(dy + pbase + i) = make_float4(o[0], o[1], o[2], o[3]);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += kThreads) {
            const float yv = y[pbase + i];
            const float d = dpre(dout[pbase + i], yv, mu, sc, sh, slope);
            const float o = sc * (d - k2 - ((yv - mu) * is) * k3);
            dy[pbase + i] = o;
            amax = fmaxf(amax, fabsf(o));
        }
    }
    if (amax_out) {                              // uniform branch
        __shared__ float sm[16];
        uaps::block_amax_to(amax_out, amax, sm);
    }
}

// eval-mode backward (running statistics are constants): dy = scale * dpre ; used only if someone
// differentiates through an eval() forward.
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_eval_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                               float* __restrict__ dy, int C, long HW, const float* __restrict__ mean,
                                                               const float* __restrict__ coef, float slope) {
    const int plane = blockIdx.y, chunk = blockIdx.x;
    const int c = plane % C;
    const float mu = mean[c], sc = coef[c], sh = coef[C + c];
    const long pbase = (long)plane * HW;
    const long lo = (long)chunk * kChunk, hi = min(HW, lo + (long)kChunk);
    const int step = VEC ? 4 : 1;
    for (long i = lo + step * threadIdx.x; i < hi; i += step * kThreads) {
        if (VEC) {
            const float4 g4 = *reinterpret_cast<const float4*>(dout + pbase + i);
            const float4 y4 = *reinterpret_cast<const float4*>(y + pbase + i);
            *reinterpret_cast<float4*>(dy + pbase + i) = make_float4(sc * dpre(g4.x, y4.x, mu, sc, sh, slope), sc * dpre(g4.y, y4.y, mu, sc, sh, slope),
                                                                     sc * dpre(g4.z, y4.z, mu, sc, sh, slope), sc * dpre(g4.w, y4.w, mu, sc, sh, slope));
        } else {
            dy[pbase + i] = sc * dpre(dout[pbase + i], y[pbase + i], mu, sc, sh, slope);
        }
    }
}

inline int check(const void* a, const void* b, int B, int C, int H, int W) {
    if (!a || !b || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if ((long)B * C > 65535) return UAPS_ERANGE;      // planes ride on gridDim.y
    return UAPS_OK;
}

}  // namespace

extern "C" int uaps_bn_workspace_bytes(int B, int C, int H, int W, size_t* out) {
    if (!out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    *out = bn_ws_bytes(B, C, (long)H * W);
    return UAPS_OK;
}

static int bn_fwd_train_impl(const uaps_call_hints& hints, const float2* given_partials, int given_parts_per_image, const float* y, const float* conv_bias, const float* gamma, const float* beta,
                                             float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                             float momentum, float eps, float slope, float drop_p, uint64_t seed, uint64_t offset,
                                             int B, int C, int H, int W, int groups, float* out, float* save_mean,
                                             float* save_invstd, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    // given partials: formed about the shift the caller names; own statistics pass: about running_mean - conv_bias
    const float* shm = given_partials ? hints.stats_mean : running_mean;
    const float* shb = given_partials ? hints.stats_bias : conv_bias;
    int rc = check(y, out, B, C, H, W);
    if (rc) return rc;
    if (!gamma || !beta || !save_mean || !save_invstd || !ws || !(drop_p >= 0.f && drop_p < 1.f)) return UAPS_EINVAL;
    if (groups < 1 || groups > kMaxGroups || B % groups) return UAPS_EINVAL;
    // y read and the activated tensor written (+ the residual read; + one more read of y when the statistics pass runs here)
    uaps::account_bytes(4.0 * B * C * H * W * (2.0 + (given_partials ? 0.0 : 1.0) + (hints.residual ? 1.0 : 0.0)));
    const long HW = (long)H * W;
    if (ws_bytes < bn_ws_bytes(B, C, HW)) return UAPS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const BnWs w = carve(ws, B, C, HW);
    const int nch = nchunks_for(HW), Bg = B / groups;
    const dim3 grid(nch, B * C);
    const bool vec = (HW % 4 == 0) && al16(y) && al16(out);
    if (!given_partials) {
        if (vec) hipLaunchKernelGGL(bn_stats_kernel<true>, grid, dim3(kThreads), 0, s, y, C, HW, nch, w.partials, shm, shb);
        else hipLaunchKernelGGL(bn_stats_kernel<false>, grid, dim3(kThreads), 0, s, y, C, HW, nch, w.partials, shm, shb);
    }
    hipLaunchKernelGGL(bn_finalize_fwd, dim3(C), dim3(kThreads * (groups < 4 ? groups : 4)), 0, s, given_partials ? given_partials : w.partials, B, Bg,
                       given_partials ? given_parts_per_image : nch, (double)HW, conv_bias, gamma, beta, running_mean,
                       running_var, num_batches_tracked, momentum, eps, save_mean, save_invstd, w.coef, C, (float2*)nullptr, shm, shb);
    const float dscale = 1.f / (1.f - drop_p);
    const float* res = hints.residual;
    if (res && (drop_p > 0.f || !al16(res))) return UAPS_EINVAL;      // the join has no dropout; the residual is a tensor like y
#define UAPS_APPLY(V, D) hipLaunchKernelGGL((bn_apply_kernel<V, D>), grid, dim3(kThreads), 0, s, y, out, C, HW, save_mean, w.coef, slope, drop_p, dscale, seed, offset, Bg, (const uint32_t*)uaps_get_step_state(), res, hints.out_amax)
    if (vec) { if (drop_p > 0.f) UAPS_APPLY(true, true); else UAPS_APPLY(true, false); }
    else { if (drop_p > 0.f) UAPS_APPLY(false, true); else UAPS_APPLY(false, false); }
#undef UAPS_APPLY
    return (int)hipGetLastError();
}

extern "C" int uaps_bn_act_fwd_train_grouped(const float* y, const float* conv_bias, const float* gamma, const float* beta,
                                             float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                             float momentum, float eps, float slope, float drop_p, uint64_t seed, uint64_t offset,
                                             int B, int C, int H, int W, int groups, float* out, float* save_mean,
                                             float* save_invstd, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    return bn_fwd_train_impl(uaps::take_hints(), nullptr, 0, y, conv_bias, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                             slope, drop_p, seed, offset, B, C, H, W, groups, out, save_mean, save_invstd, ws, ws_bytes, stream);
}

// The statistics pass is skipped: `partials` (float2 [C][B][parts_per_image], e.g. from uaps_conv_fwd_stats) already
// holds per-image partial (sum, sum of squares) of y.
#define UAPS_BN_PARTIALS_PARAMS const void* partials, int parts_per_image, const float* y, const float* conv_bias,                        \
                                              const float* gamma, const float* beta, float* running_mean, float* running_var,               \
                                              int64_t* num_batches_tracked, float momentum, float eps, float slope, float drop_p,           \
                                              uint64_t seed, uint64_t offset, int B, int C, int H, int W, int groups, float* out,           \
                                              float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, uaps_stream_t stream
#define UAPS_BN_PARTIALS_ARGS (const float2*)partials, parts_per_image, y, conv_bias, gamma, beta, running_mean, running_var,                \
                             num_batches_tracked, momentum, eps, slope, drop_p, seed, offset, B, C, H, W, groups, out, save_mean,            \
                             save_invstd, ws, ws_bytes, stream
extern "C" int uaps_bn_act_fwd_train_partials(UAPS_BN_PARTIALS_PARAMS) {
    const uaps_call_hints h = uaps::take_hints();
    if (!partials || parts_per_image <= 0) return UAPS_EINVAL;
    return bn_fwd_train_impl(h, UAPS_BN_PARTIALS_ARGS);
}
// (the *_h forms: the hints of THIS call as the first argument, nothing thread-local -- see conv_fwd.hip)
extern "C" int uaps_bn_act_fwd_train_partials_h(const uaps_call_hints* hints, UAPS_BN_PARTIALS_PARAMS) {
    UAPS_READ_HINTS(hints, h);
    if (!partials || parts_per_image <= 0) return UAPS_EINVAL;
    return bn_fwd_train_impl(h, UAPS_BN_PARTIALS_ARGS);
}
#undef UAPS_BN_PARTIALS_PARAMS
#undef UAPS_BN_PARTIALS_ARGS

// Statistics finalize only (no apply pass): per group and channel the batch mean / inverse std from the conv
// epilogue's partials, the running-statistics update, and xf [groups][C] float2 = (scale, shift) = (gamma*invstd, beta - mean*scale),
// the coefficients uaps_conv_fwd_bn / uaps_conv_bwd_weight_partial_bn apply while staging their input.
static int bn_finalize_train_impl(const uaps_call_hints& hints, const void* partials, int parts_per_image, const float* conv_bias, const float* gamma,
                                      const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                      float momentum, float eps, int B, int C, int H, int W, int groups, float* save_mean,
                                      float* save_invstd, void* xf, uaps_stream_t stream) {
    if (!partials || parts_per_image <= 0 || !gamma || !beta || !save_mean || !save_invstd || !xf || ((uintptr_t)xf % 8)) return UAPS_EINVAL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || groups < 1 || groups > kMaxGroups || B % groups) return UAPS_EINVAL;
    hipLaunchKernelGGL(bn_finalize_fwd, dim3(C), dim3(kThreads * (groups < 4 ? groups : 4)), 0, (hipStream_t)stream, (const float2*)partials, B, B / groups,
                       parts_per_image, (double)H * W, conv_bias, gamma, beta, running_mean, running_var, num_batches_tracked,
                       momentum, eps, save_mean, save_invstd, (float*)nullptr, C, (float2*)xf, hints.stats_mean, hints.stats_bias);
    return (int)hipGetLastError();
}
extern "C" int uaps_bn_finalize_train(const void* partials, int parts_per_image, const float* conv_bias, const float* gamma,
                                      const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                      float momentum, float eps, int B, int C, int H, int W, int groups, float* save_mean,
                                      float* save_invstd, void* xf, uaps_stream_t stream) {
    return bn_finalize_train_impl(uaps::take_hints(), partials, parts_per_image, conv_bias, gamma, beta, running_mean, running_var,
                                  num_batches_tracked, momentum, eps, B, C, H, W, groups, save_mean, save_invstd, xf, stream);
}
extern "C" int uaps_bn_finalize_train_h(const uaps_call_hints* hints, const void* partials, int parts_per_image, const float* conv_bias,
                                        const float* gamma, const float* beta, float* running_mean, float* running_var,
                                        int64_t* num_batches_tracked, float momentum, float eps, int B, int C, int H, int W, int groups,
                                        float* save_mean, float* save_invstd, void* xf, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    return bn_finalize_train_impl(h, partials, parts_per_image, conv_bias, gamma, beta, running_mean, running_var,
                                  num_batches_tracked, momentum, eps, B, C, H, W, groups, save_mean, save_invstd, xf, stream);
}

extern "C" int uaps_bn_act_fwd_train(const float* y, const float* conv_bias, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                     float eps, float slope, float drop_p, uint64_t seed, uint64_t offset, int B, int C, int H,
                                     int W, float* out, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                                     uaps_stream_t stream) {
    return uaps_bn_act_fwd_train_grouped(y, conv_bias, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                         slope, drop_p, seed, offset, B, C, H, W, 1, out, save_mean, save_invstd, ws, ws_bytes, stream);
}

extern "C" int uaps_bn_act_fwd_eval(const float* y, const float* conv_bias, const float* gamma, const float* beta,
                                    const float* running_mean, const float* running_var, float eps, float slope, int B, int C,
                                    int H, int W, float* out, float* save_mean, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    int rc = check(y, out, B, C, H, W);
    if (rc) return rc;
    if (!gamma || !beta || !running_mean || !running_var || !save_mean || !ws) return UAPS_EINVAL;
    const long HW = (long)H * W;
    if (ws_bytes < bn_ws_bytes(B, C, HW)) return UAPS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const BnWs w = carve(ws, B, C, HW);
    hipLaunchKernelGGL(bn_coef_eval, dim3((C + 63) / 64), dim3(64), 0, s, conv_bias, gamma, beta, running_mean, running_var, eps, w.coef, save_mean, C);
    const dim3 grid(nchunks_for(HW), B * C);
    if ((HW % 4 == 0) && al16(y) && al16(out))
        hipLaunchKernelGGL((bn_apply_kernel<true, false>), grid, dim3(kThreads), 0, s, y, out, C, HW, save_mean, w.coef, slope, 0.f, 1.f, 0ull, 0ull, B, (const uint32_t*)nullptr, (const float*)nullptr, (float*)nullptr);
    else
        hipLaunchKernelGGL((bn_apply_kernel<false, false>), grid, dim3(kThreads), 0, s, y, out, C, HW, save_mean, w.coef, slope, 0.f, 1.f, 0ull, 0ull, B, (const uint32_t*)nullptr, (const float*)nullptr, (float*)nullptr);
    return (int)hipGetLastError();
}

static int bn_bwd_impl(float* amax_out, const float* dout, const float* y, const float* gamma, const float* beta,
                                       const float* save_mean, const float* save_invstd, float slope, float drop_p, uint64_t seed,
                                       uint64_t offset, int B, int C, int H, int W, int groups, float* dy, float* dgamma,
                                       float* dbeta, float* dconv_bias, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    int rc = check(dout, dy, B, C, H, W);
    if (rc) return rc;
    if (!y || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !ws || !(drop_p >= 0.f && drop_p < 1.f)) return UAPS_EINVAL;
    if (groups < 1 || groups > kMaxGroups || B % groups) return UAPS_EINVAL;
    const long HW = (long)H * W;
    if (ws_bytes < bn_ws_bytes(B, C, HW)) return UAPS_EWORKSPACE;
    uaps::account_bytes(4.0 * B * C * HW * 5.0);      // the reductions need all of (d, y) before dx can start: 2 + 2 reads, 1 write
    hipStream_t s = (hipStream_t)stream;
    const BnWs w = carve(ws, B, C, HW);
    const int nch = nchunks_for(HW), Bg = B / groups;
    const dim3 grid(nch, B * C);
    const bool vec = (HW % 4 == 0) && al16(y) && al16(dout) && al16(dy);
    const float dscale = 1.f / (1.f - drop_p);
#define UAPS_SUMS(V, D) hipLaunchKernelGGL((bn_bwd_sums_kernel<V, D>), grid, dim3(kThreads), 0, s, dout, y, C, HW, nch, save_mean, save_invstd, gamma, beta, slope, drop_p, dscale, seed, offset, w.partials, Bg, (const uint32_t*)uaps_get_step_state())
#define UAPS_DX(V, D) hipLaunchKernelGGL((bn_bwd_dx_kernel<V, D>), grid, dim3(kThreads), 0, s, dout, y, dy, C, HW, save_mean, save_invstd, gamma, beta, (const float2*)w.partials, nch, B, dgamma, dbeta, dconv_bias, slope, drop_p, dscale, seed, offset, Bg, (const uint32_t*)uaps_get_step_state(), amax_out)
    if (vec) { if (drop_p > 0.f) UAPS_SUMS(true, true); else UAPS_SUMS(true, false); }
    else { if (drop_p > 0.f) UAPS_SUMS(false, true); else UAPS_SUMS(false, false); }
    if (vec) { if (drop_p > 0.f) UAPS_DX(true, true); else UAPS_DX(true, false); }
    else { if (drop_p > 0.f) UAPS_DX(false, true); else UAPS_DX(false, false); }
#undef UAPS_SUMS
#undef UAPS_DX
    return (int)hipGetLastError();
}

extern "C" int uaps_bn_act_bwd_grouped(const float* dout, const float* y, const float* gamma, const float* beta,
                                       const float* save_mean, const float* save_invstd, float slope, float drop_p, uint64_t seed,
                                       uint64_t offset, int B, int C, int H, int W, int groups, float* dy, float* dgamma,
                                       float* dbeta, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    return bn_bwd_impl(uaps::take_hints().out_amax, dout, y, gamma, beta, save_mean, save_invstd, slope, drop_p, seed, offset, B, C, H, W, groups, dy, dgamma,
                       dbeta, nullptr, ws, ws_bytes, stream);
}
extern "C" int uaps_bn_act_bwd_grouped_h(const uaps_call_hints* hints, const float* dout, const float* y, const float* gamma, const float* beta,
                                         const float* save_mean, const float* save_invstd, float slope, float drop_p, uint64_t seed,
                                         uint64_t offset, int B, int C, int H, int W, int groups, float* dy, float* dgamma,
                                         float* dbeta, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    return bn_bwd_impl(h.out_amax, dout, y, gamma, beta, save_mean, save_invstd, slope, drop_p, seed, offset, B, C, H, W, groups, dy, dgamma,
                       dbeta, nullptr, ws, ws_bytes, stream);
}

// The same, also writing the (identically zero) gradient of the bias of the convolution in front of the BatchNorm
// into dconv_bias [C], so that the host needs no separate fill launch for it.
extern "C" int uaps_bn_act_bwd_grouped_bias(const float* dout, const float* y, const float* gamma, const float* beta,
                                            const float* save_mean, const float* save_invstd, float slope, float drop_p,
                                            uint64_t seed, uint64_t offset, int B, int C, int H, int W, int groups, float* dy,
                                            float* dgamma, float* dbeta, float* dconv_bias, void* ws, size_t ws_bytes,
                                            uaps_stream_t stream) {
    float* amax_out = uaps::take_hints().out_amax;
    if (!dconv_bias) return UAPS_EINVAL;
    return bn_bwd_impl(amax_out, dout, y, gamma, beta, save_mean, save_invstd, slope, drop_p, seed, offset, B, C, H, W, groups, dy, dgamma,
                       dbeta, dconv_bias, ws, ws_bytes, stream);
}
extern "C" int uaps_bn_act_bwd_grouped_bias_h(const uaps_call_hints* hints, const float* dout, const float* y, const float* gamma,
                                              const float* beta, const float* save_mean, const float* save_invstd, float slope, float drop_p,
                                              uint64_t seed, uint64_t offset, int B, int C, int H, int W, int groups, float* dy,
                                              float* dgamma, float* dbeta, float* dconv_bias, void* ws, size_t ws_bytes,
                                              uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    if (!dconv_bias) return UAPS_EINVAL;
    return bn_bwd_impl(h.out_amax, dout, y, gamma, beta, save_mean, save_invstd, slope, drop_p, seed, offset, B, C, H, W, groups, dy, dgamma,
                       dbeta, dconv_bias, ws, ws_bytes, stream);
}

extern "C" int uaps_bn_act_bwd(const float* dout, const float* y, const float* gamma, const float* beta, const float* save_mean,
                               const float* save_invstd, float slope, float drop_p, uint64_t seed, uint64_t offset, int B, int C,
                               int H, int W, float* dy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                               uaps_stream_t stream) {
    return uaps_bn_act_bwd_grouped(dout, y, gamma, beta, save_mean, save_invstd, slope, drop_p, seed, offset, B, C, H, W, 1, dy,
                                   dgamma, dbeta, ws, ws_bytes, stream);
}

// The BatchNorm(train) + LeakyReLU backward in two halves (no dropout): `prepare` runs the reductions -- coef [groups][C][8] floats
// = (mean, invstd, gamma invstd, beta, mean(d), mean(d x_hat), 0, 0), dgamma, dbeta, the zero gradient of a conv bias in front
// (may be NULL) -- and raises the zeroed bound `dy_bound` to an upper bound of |dy|; dy itself is then formed either by the
// weight-gradient kernel of the convolution in front (uaps_call_hints::dyt_*: while it stages its dy operand, written through to
// dyt_out for the input-gradient kernel) or by `apply` (uaps_call_hints::out_amax honoured).
extern "C" int uaps_bn_act_bwd_prepare(const float* dout, const float* y, const float* gamma, const float* beta, const float* save_mean,
                                       const float* save_invstd, float slope, int B, int C, int H, int W, int groups, float* coef,
                                       float* dgamma, float* dbeta, float* dconv_bias, float* dy_bound, void* ws, size_t ws_bytes,
                                       uaps_stream_t stream) {
    if (!dout || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (!gamma || !beta || !save_mean || !save_invstd || !coef || !dgamma || !dbeta || !dy_bound || !ws) return UAPS_EINVAL;
    if (groups < 1 || groups > kMaxGroups || B % groups) return UAPS_EINVAL;
    const long HW = (long)H * W;
    if (ws_bytes < bn_ws_bytes(B, C, HW)) return UAPS_EWORKSPACE;
    uaps::account_bytes(4.0 * B * C * HW * 2.0);
    hipStream_t s = (hipStream_t)stream;
    const BnWs w = carve(ws, B, C, HW);
    const int nch = nchunks_for(HW), Bg = B / groups;
    int rc = uaps_zero_bounds(w.maxes, 2 * UAPS_BOUND_FLOATS, stream);
    if (rc) return rc;
    const dim3 grid(nch, B * C);
    float* gmax = w.maxes; float* xmax = w.maxes + UAPS_BOUND_FLOATS;
    if ((HW % 4 == 0) && al16(y) && al16(dout))
        hipLaunchKernelGGL(bn_bwd_sums_max_kernel<true>, grid, dim3(kThreads), 0, s, dout, y, C, HW, nch, save_mean, save_invstd, gamma, beta, slope, w.partials, Bg, B, gmax, xmax);
    else
        hipLaunchKernelGGL(bn_bwd_sums_max_kernel<false>, grid, dim3(kThreads), 0, s, dout, y, C, HW, nch, save_mean, save_invstd, gamma, beta, slope, w.partials, Bg, B, gmax, xmax);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, (const float2*)w.partials, nch, B, Bg, C, HW, save_mean, save_invstd, gamma, beta,
                       (const float*)gmax, (const float*)xmax, coef, dgamma, dbeta, dconv_bias, dy_bound);
    return (int)hipGetLastError();
}

// The second half of uaps_bn_act_bwd_prepare on its own: the sums were formed in the epilogue of the kernel that wrote the gradient
// (uaps_call_hints::bsum_* on uaps_conv_bwd_data) -- partials float2 [C][B][parts_per_image], maxes = the two raised bounds.
extern "C" int uaps_bn_act_bwd_finalize(const void* partials, int parts_per_image, const float* maxes, const float* gamma, const float* beta,
                                        const float* save_mean, const float* save_invstd, int B, int C, int H, int W, int groups,
                                        float* coef, float* dgamma, float* dbeta, float* dconv_bias, float* dy_bound, uaps_stream_t stream) {
    if (!partials || parts_per_image <= 0 || !maxes || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (!gamma || !beta || !save_mean || !save_invstd || !coef || !dgamma || !dbeta || !dy_bound) return UAPS_EINVAL;
    if (groups < 1 || groups > kMaxGroups || B % groups) return UAPS_EINVAL;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, (const float2*)partials, parts_per_image, B, B / groups, C,
                       (long)H * W, save_mean, save_invstd, gamma, beta, maxes, maxes + UAPS_BOUND_FLOATS, coef, dgamma, dbeta, dconv_bias, dy_bound);
    return (int)hipGetLastError();
}

static int bn_bwd_apply_impl(float* amax_out, const float* dout, const float* y, const float* coef, float slope, int B, int C, int H, int W,
                             int groups, float* dy, uaps_stream_t stream);
extern "C" int uaps_bn_act_bwd_apply(const float* dout, const float* y, const float* coef, float slope, int B, int C, int H, int W,
                                     int groups, float* dy, uaps_stream_t stream) {
    return bn_bwd_apply_impl(uaps::take_hints().out_amax, dout, y, coef, slope, B, C, H, W, groups, dy, stream);
}
extern "C" int uaps_bn_act_bwd_apply_h(const uaps_call_hints* hints, const float* dout, const float* y, const float* coef, float slope, int B,
                                       int C, int H, int W, int groups, float* dy, uaps_stream_t stream) {
    UAPS_READ_HINTS(hints, h);
    return bn_bwd_apply_impl(h.out_amax, dout, y, coef, slope, B, C, H, W, groups, dy, stream);
}
static int bn_bwd_apply_impl(float* amax_out, const float* dout, const float* y, const float* coef, float slope, int B, int C, int H, int W,
                             int groups, float* dy, uaps_stream_t stream) {
    int rc = check(dout, dy, B, C, H, W);
    if (rc) return rc;
    if (!y || !coef || groups < 1 || groups > kMaxGroups || B % groups) return UAPS_EINVAL;
    const long HW = (long)H * W;
    uaps::account_bytes(4.0 * B * C * HW * 3.0);
    const dim3 grid(nchunks_for(HW), B * C);
    if ((HW % 4 == 0) && al16(y) && al16(dout) && al16(dy))
        hipLaunchKernelGGL(bn_bwd_dx_coef_kernel<true>, grid, dim3(kThreads), 0, (hipStream_t)stream, dout, y, dy, C, HW, coef, slope, B / groups, amax_out);
    else
        hipLaunchKernelGGL(bn_bwd_dx_coef_kernel<false>, grid, dim3(kThreads), 0, (hipStream_t)stream, dout, y, dy, C, HW, coef, slope, B / groups, amax_out);
    return (int)hipGetLastError();
}

extern "C" int uaps_bn_act_bwd_eval(const float* dout, const float* y, const float* gamma, const float* beta,
                                    const float* mean_eff, const float* running_var, float eps, float slope, int B, int C, int H,
                                    int W, float* dy, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    int rc = check(dout, dy, B, C, H, W);
    if (rc) return rc;
    if (!y || !gamma || !beta || !mean_eff || !running_var || !ws) return UAPS_EINVAL;
    const long HW = (long)H * W;
    if (ws_bytes < bn_ws_bytes(B, C, HW)) return UAPS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const BnWs w = carve(ws, B, C, HW);
    // coef[0..C) = gamma / sqrt(rv + eps), coef[C..2C) = beta; the mean (running_mean - conv bias) is passed in
    hipLaunchKernelGGL(bn_coef_eval, dim3((C + 63) / 64), dim3(64), 0, s, (const float*)nullptr, gamma, beta, mean_eff, running_var, eps, w.coef, w.coef + 2 * C, C);
    const dim3 grid(nchunks_for(HW), B * C);
    if ((HW % 4 == 0) && al16(y) && al16(dout) && al16(dy))
        hipLaunchKernelGGL(bn_bwd_eval_kernel<true>, grid, dim3(kThreads), 0, s, dout, y, dy, C, HW, mean_eff, w.coef, slope);
    else
        hipLaunchKernelGGL(bn_bwd_eval_kernel<false>, grid, dim3(kThreads), 0, s, dout, y, dy, C, HW, mean_eff, w.coef, slope);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Magnitude bounds of train-mode BatchNorm outputs from the parameters alone: with batch statistics every normalised
// value obeys |x_hat| <= sqrt(n - 1) (Samuelson's inequality, n = elements per channel and statistics group), hence
//   |gamma * x_hat + beta| <= sqrt(n) * max_c(|gamma_c| + |beta_c|)        (n >= 1),
// and LeakyReLU (slope <= 1) only shrinks it.  Bound i (out + i * UAPS_BOUND_FLOATS) = max_c(|gamma_c| + |beta_c|) of layer i; the caller multiplies by
// sqrt(n) (and 1 / (1 - p) of a following dropout) when it passes the bound on (uaps_next_call_hints).  One launch for all
// layers of a model, once per optimizer step.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kBoundBatch = 64;
struct BoundBatch { const float* gamma[kBoundBatch]; const float* beta[kBoundBatch]; int C[kBoundBatch]; };
__global__ __launch_bounds__(64) void bn_param_bounds_kernel(BoundBatch bb, float* __restrict__ out) {
    const int i = blockIdx.x, C = bb.C[i];
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) m = fmaxf(m, fabsf(bb.gamma[i][c]) + fabsf(bb.beta[i][c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // a bound is UAPS_BOUND_SLOTS strided floats whose maximum counts (include/uaps_hip.h): slot 0 = m, the others 0
    // agent-scope stores: bounds are read with agent-scope loads (hints.hpp bound_max), every access to bound storage bypasses the XCD L2s
    if (threadIdx.x < UAPS_BOUND_SLOTS)
        __hip_atomic_store(out + (long)i * UAPS_BOUND_FLOATS + threadIdx.x * UAPS_BOUND_STRIDE, threadIdx.x == 0 ? m : 0.f, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

extern "C" int uaps_bn_param_bounds(const float* const* gamma, const float* const* beta, const int* C, int n, float* out,
                                    uaps_stream_t stream) {
    if (!gamma || !beta || !C || !out || n <= 0) return UAPS_EINVAL;
    for (int base = 0; base < n; base += kBoundBatch) {
        const int m = n - base < kBoundBatch ? n - base : kBoundBatch;
        BoundBatch bb{};
        for (int i = 0; i < m; ++i) {
            if (!gamma[base + i] || !beta[base + i] || C[base + i] <= 0) return UAPS_EINVAL;
            bb.gamma[i] = gamma[base + i]; bb.beta[i] = beta[base + i]; bb.C[i] = C[base + i];
        }
        hipLaunchKernelGGL(bn_param_bounds_kernel, dim3(m), dim3(64), 0, (hipStream_t)stream, bb, out + (size_t)base * UAPS_BOUND_FLOATS);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return UAPS_OK;
}
