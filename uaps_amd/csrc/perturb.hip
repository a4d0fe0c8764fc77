// Feature perturbations of the auxiliary decoders (utilities/UAPS_unet.py:156-185) and the
// confusion-matrix metric kernel (utilities/metrics.py:8-61) for gfx950.
// All of these are pure streaming kernels: 16 B per lane, grid-stride, HBM-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uaps_hip.h"
#include "rn_math.hpp"
#include "philox.hpp"
#include "hints.hpp"
using uaps::mul_rn; using uaps::add_rn; using uaps::U4; using uaps::philox4x32_10; using uaps::u01;

namespace {
// Zero-fill as a kernel of this library.  hipMemsetAsync in front of the atomics that accumulate into the same words is ordered
// with them in a stream, but under hipGraph replay the FeatureDropout maxima were observed to differ from the eager step from
// the third replay on (tools/diag/graph_val_debug.py): memset nodes and kernel nodes are different engines' work.
// Agent-scope stores: the words are then updated by memory-side atomics; a plain store may sit in one XCD's write-back L2 and be
// written back over them later (the same hazard as the magnitude bounds, uaps_zero_bounds in hints.hip).
__global__ void zero_words_kernel(uint32_t* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __hip_atomic_store(p + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;

inline int grid_for(long work) {
    long b = (work + kThreads - 1) / kThreads;
    if (b > kMaxBlocks) b = kMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- FeatureNoise: y[b,e] = x[b,e] * n[e] + x[b,e], n ~ U(-range, range), e over C*H*W ------------
__global__ __launch_bounds__(kThreads) void noise_rng_vec4(const float4* __restrict__ x, float4* __restrict__ y, int B,
                                                           long chw4, uint64_t seed, uint64_t offset, float range,
                                                           float4* __restrict__ noise_out) {
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < chw4; e += (long)gridDim.x * kThreads) {
        const U4 r = philox4x32_10(offset + (uint64_t)e, seed);
        float4 n;
        n.x = (2.f * u01(r.x) - 1.f) * range; n.y = (2.f * u01(r.y) - 1.f) * range;
        n.z = (2.f * u01(r.z) - 1.f) * range; n.w = (2.f * u01(r.w) - 1.f) * range;
        if (noise_out) noise_out[e] = n;
        for (int b = 0; b < B; ++b) {
            const float4 v = x[(long)b * chw4 + e];
            float4 o;   // x.mul(noise) + x : separate roundings as in UAPS_unet.py:180
            o.x = add_rn(mul_rn(v.x, n.x), v.x); o.y = add_rn(mul_rn(v.y, n.y), v.y);
            o.z = add_rn(mul_rn(v.z, n.z), v.z); o.w = add_rn(mul_rn(v.w, n.w), v.w);
            y[(long)b * chw4 + e] = o;
        }
    }
}
__global__ __launch_bounds__(kThreads) void noise_rng_scalar(const float* __restrict__ x, float* __restrict__ y, int B, long chw,
                                                             uint64_t seed, uint64_t offset, float range,
                                                             float* __restrict__ noise_out) {
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < chw; e += (long)gridDim.x * kThreads) {
        const U4 r = philox4x32_10(offset + (uint64_t)(e >> 2), seed);
        const uint32_t rr = (e & 3) == 0 ? r.x : (e & 3) == 1 ? r.y : (e & 3) == 2 ? r.z : r.w;
        const float n = (2.f * u01(rr) - 1.f) * range;
        if (noise_out) noise_out[e] = n;
        for (int b = 0; b < B; ++b) { const float v = x[(long)b * chw + e]; y[(long)b * chw + e] = add_rn(mul_rn(v, n), v); }
    }
}
__global__ __launch_bounds__(kThreads) void noise_apply_vec4(const float4* __restrict__ x, const float4* __restrict__ noise,
                                                             float4* __restrict__ y, int B, long chw4) {
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < chw4; e += (long)gridDim.x * kThreads) {
        const float4 n = noise[e];
        for (int b = 0; b < B; ++b) {
            const float4 v = x[(long)b * chw4 + e];
            float4 o;
            o.x = add_rn(mul_rn(v.x, n.x), v.x); o.y = add_rn(mul_rn(v.y, n.y), v.y);
            o.z = add_rn(mul_rn(v.z, n.z), v.z); o.w = add_rn(mul_rn(v.w, n.w), v.w);
            y[(long)b * chw4 + e] = o;
        }
    }
}
__global__ __launch_bounds__(kThreads) void noise_apply_scalar(const float* __restrict__ x, const float* __restrict__ noise,
                                                               float* __restrict__ y, int B, long chw) {
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < chw; e += (long)gridDim.x * kThreads) {
        const float n = noise[e];
        for (int b = 0; b < B; ++b) { const float v = x[(long)b * chw + e]; y[(long)b * chw + e] = add_rn(mul_rn(v, n), v); }
    }
}

// ---- Dropout(x, p): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) ------------------------------------
__global__ __launch_bounds__(kThreads) void bernoulli_vec4(const float4* __restrict__ x, float4* __restrict__ y, long n4,
                                                           uint64_t seed, uint64_t offset, float p, float scale,
                                                           uchar4* __restrict__ keep_out) {
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < n4; e += (long)gridDim.x * kThreads) {
        const U4 r = philox4x32_10(offset + (uint64_t)e, seed);
        const float4 v = x[e];
        const bool k0 = u01(r.x) >= p, k1 = u01(r.y) >= p, k2 = u01(r.z) >= p, k3 = u01(r.w) >= p;
        float4 o;
        o.x = k0 ? v.x * scale : 0.f; o.y = k1 ? v.y * scale : 0.f; o.z = k2 ? v.z * scale : 0.f; o.w = k3 ? v.w * scale : 0.f;
        y[e] = o;
        if (keep_out) keep_out[e] = make_uchar4(k0, k1, k2, k3);
    }
}
__global__ __launch_bounds__(kThreads) void bernoulli_scalar(const float* __restrict__ x, float* __restrict__ y, long start, long n,
                                                             uint64_t seed, uint64_t offset, float p, float scale,
                                                             uint8_t* __restrict__ keep_out) {
    for (long e = start + (long)blockIdx.x * kThreads + threadIdx.x; e < n; e += (long)gridDim.x * kThreads) {
        const U4 r = philox4x32_10(offset + (uint64_t)(e >> 2), seed);
        const uint32_t rr = (e & 3) == 0 ? r.x : (e & 3) == 1 ? r.y : (e & 3) == 2 ? r.z : r.w;
        const bool k = u01(rr) >= p;
        y[e] = k ? x[e] * scale : 0.f;
        if (keep_out) keep_out[e] = k;
    }
}
__global__ __launch_bounds__(kThreads) void mask_apply_kernel(const float* __restrict__ x, const uint8_t* __restrict__ keep,
                                                              float scale, float* __restrict__ y, long n) {
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < n; e += (long)gridDim.x * kThreads)
        y[e] = keep[e] ? x[e] * scale : 0.f;
}

// ---- FeatureDropout ----------------------------------------------------------------------------------
// order-preserving float -> uint key so that an integer atomicMax gives a deterministic float max
__device__ __forceinline__ uint32_t fkey(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// pass A: att[b,hw] = (sum_c x[b,c,hw]) / C, per-sample max via wave reduce + one atomicMax per wave.
// V pixels per lane (16-byte loads when V == 4); the channel sum keeps the c = 0..C-1 order.
template <int V>
__global__ __launch_bounds__(kThreads) void fdrop_attention(const float* __restrict__ x, int C, long HW, float* __restrict__ att,
                                                            uint32_t* __restrict__ maxkey) {
    const int b = blockIdx.y;
    const float* xb = x + (long)b * C * HW;
    uint32_t best = 0;
    for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * V; i < HW; i += (long)gridDim.x * kThreads * V) {
        float s[V];
        if constexpr (V == 4) {
            const float4 v = *reinterpret_cast<const float4*>(xb + i);
            s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w;
            int c = 1;
            for (; c + 14 < C; c += 15) {          // fifteen loads in flight (a 16-channel level is one batch), adds in channel order
                float4 t[15];
#pragma unroll
                for (int q = 0; q < 15; ++q) t[q] = *reinterpret_cast<const float4*>(xb + (long)(c + q) * HW + i);
#pragma unroll
                for (int q = 0; q < 15; ++q) {
                    s[0] = add_rn(s[0], t[q].x); s[1] = add_rn(s[1], t[q].y); s[2] = add_rn(s[2], t[q].z); s[3] = add_rn(s[3], t[q].w);
                }
            }
            for (; c + 3 < C; c += 4) {            // four loads in flight
                float4 t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const float4*>(xb + (long)(c + q) * HW + i);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s[0] = add_rn(s[0], t[q].x); s[1] = add_rn(s[1], t[q].y); s[2] = add_rn(s[2], t[q].z); s[3] = add_rn(s[3], t[q].w);
                }
            }
            for (; c < C; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(xb + (long)c * HW + i);
                s[0] = add_rn(s[0], t.x); s[1] = add_rn(s[1], t.y); s[2] = add_rn(s[2], t.z); s[3] = add_rn(s[3], t.w);
            }
        } else {
            s[0] = xb[i];
            for (int c = 1; c < C; ++c) s[0] = add_rn(s[0], xb[(long)c * HW + i]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            s[k] = s[k] / (float)C;
            const uint32_t key = fkey(s[k]);
            best = key > best ? key : best;
        }
        if constexpr (V == 4) *reinterpret_cast<float4*>(att + (long)b * HW + i) = make_float4(s[0], s[1], s[2], s[3]);
        else att[(long)b * HW + i] = s[0];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(best, o, 64); best = t > best ? t : best; }
    // one atomic per block, not per wave: a 256x256 level would otherwise queue 256 atomics per image on one address
    __shared__ uint32_t wbest[kThreads / 64];
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = max(max(wbest[0], wbest[1]), max(wbest[2], wbest[3]));
        if (best) atomicMax(maxkey + b, best);
    }
}
// The same for the deep encoder levels (many channels, few pixels): the four waves of a block own the same 256 pixels
// and a quarter of the channels each (in channel order), the quarter sums are combined ((q0 + q1) + q2) + q3 through
// LDS -- 4x the loads in flight where the plain kernel is latency-bound on a serial chain of C loads per lane.
__global__ __launch_bounds__(kThreads) void fdrop_attention_split(const float* __restrict__ x, int C, long HW, float* __restrict__ att,
                                                                  uint32_t* __restrict__ maxkey) {
    __shared__ float4 part[3][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = x + (long)b * C * HW;
    const long i = ((long)blockIdx.x * 64 + lane) * 4;
    const int c0 = wave * (C / 4), c1 = c0 + C / 4;           // C % 4 == 0 (host checks)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < HW) {
        s = *reinterpret_cast<const float4*>(xb + (long)c0 * HW + i);
        int c = c0 + 1;
        for (; c + 3 < c1; c += 4) {
            float4 t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const float4*>(xb + (long)(c + q) * HW + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) { s.x = add_rn(s.x, t[q].x); s.y = add_rn(s.y, t[q].y); s.z = add_rn(s.z, t[q].z); s.w = add_rn(s.w, t[q].w); }
        }
        for (; c < c1; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(xb + (long)c * HW + i);
            s.x = add_rn(s.x, t.x); s.y = add_rn(s.y, t.y); s.z = add_rn(s.z, t.z); s.w = add_rn(s.w, t.w);
        }
    }
    if (wave > 0) part[wave - 1][lane] = s;
    __syncthreads();
    if (wave != 0) return;
    uint32_t best = 0;
    if (i < HW) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { const float4 t = part[q][lane]; s.x = add_rn(s.x, t.x); s.y = add_rn(s.y, t.y); s.z = add_rn(s.z, t.z); s.w = add_rn(s.w, t.w); }
        s.x = s.x / (float)C; s.y = s.y / (float)C; s.z = s.z / (float)C; s.w = s.w / (float)C;
        *reinterpret_cast<float4*>(att + (long)b * HW + i) = s;
        const uint32_t k0 = fkey(s.x), k1 = fkey(s.y), k2 = fkey(s.z), k3 = fkey(s.w);
        best = max(max(k0, k1), max(k2, k3));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(best, o, 64); best = t > best ? t : best; }
    if (lane == 0 && best) atomicMax(maxkey + b, best);
}
// pass B: keep = att < max * u ; y = x * keep
template <int V>
__global__ __launch_bounds__(kThreads) void fdrop_apply(const float* __restrict__ x, float* __restrict__ y, int C, long HW,
                                                        const float* __restrict__ att, const uint32_t* __restrict__ maxkey,
                                                        float u, uint8_t* __restrict__ keep) {
    const int b = blockIdx.y;
    const float thr = mul_rn(fkey_inv(__hip_atomic_load(maxkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)), u);
    const float* xb = x + (long)b * C * HW;
    float* yb = y + (long)b * C * HW;
    for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * V; i < HW; i += (long)gridDim.x * kThreads * V) {
        if constexpr (V == 4) {
            const float4 a = *reinterpret_cast<const float4*>(att + (long)b * HW + i);
            const bool k0 = a.x < thr, k1 = a.y < thr, k2 = a.z < thr, k3 = a.w < thr;
            *reinterpret_cast<uchar4*>(keep + (long)b * HW + i) = make_uchar4(k0, k1, k2, k3);
#pragma unroll 4
            for (int c = 0; c < C; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(xb + (long)c * HW + i);
                *reinterpret_cast<float4*>(yb + (long)c * HW + i) = make_float4(k0 ? v.x : 0.f, k1 ? v.y : 0.f, k2 ? v.z : 0.f, k3 ? v.w : 0.f);
            }
        } else {
            const bool k = att[(long)b * HW + i] < thr;
            keep[(long)b * HW + i] = k;
            for (int c = 0; c < C; ++c) yb[(long)c * HW + i] = k ? xb[(long)c * HW + i] : 0.f;
        }
    }
}
template <int V>
__global__ __launch_bounds__(kThreads) void fdrop_bwd(const float* __restrict__ dy, const uint8_t* __restrict__ keep,
                                                      float* __restrict__ dx, int C, long HW) {
    const int b = blockIdx.y;
    for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * V; i < HW; i += (long)gridDim.x * kThreads * V) {
        if constexpr (V == 4) {
            const uchar4 k = *reinterpret_cast<const uchar4*>(keep + (long)b * HW + i);
#pragma unroll 4
            for (int c = 0; c < C; ++c) {
                const long o = ((long)b * C + c) * HW + i;
                const float4 v = *reinterpret_cast<const float4*>(dy + o);
                *reinterpret_cast<float4*>(dx + o) = make_float4(k.x ? v.x : 0.f, k.y ? v.y : 0.f, k.z ? v.z : 0.f, k.w ? v.w : 0.f);
            }
        } else {
            const bool k = keep[(long)b * HW + i];
            for (int c = 0; c < C; ++c) { const long o = ((long)b * C + c) * HW + i; dx[o] = k ? dy[o] : 0.f; }
        }
    }
}

// ---- sum of up to 4 equally shaped tensors (gradient fan-in of a feature map used by several decoders) ------
struct SumPtrs { const float* p[4]; };
__global__ __launch_bounds__(kThreads) void sum_n_kernel(SumPtrs in, int n, float* __restrict__ out, long n4, long total) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kThreads) {
        float4 a = *reinterpret_cast<const float4*>(in.p[0] + 4 * i);
        for (int k = 1; k < n; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(in.p[k] + 4 * i);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        *reinterpret_cast<float4*>(out + 4 * i) = a;
    }
    if (blockIdx.x == 0)
        for (long i = 4 * n4 + threadIdx.x; i < total; i += kThreads) {
            float a = in.p[0][i];
            for (int k = 1; k < n; ++k) a += in.p[k][i];
            out[i] = a;
        }
}

// ---- MaxPool2d(2) of DownBlock (UAPS_unet.py:55-58): out [P,H/2,W/2] and the arg-max position (dy*2+dx, first maximum in
// scan order wins as in ATen's kernel, NaN propagates) as uint8, for the scatter-free backward in the fan-in kernel ------
__global__ __launch_bounds__(kThreads) void maxpool2x2_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                  uint8_t* __restrict__ idx, long planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2, Wq = Wo / 4;                 // 4 outputs per thread: 2 x 8 inputs
    const long total = planes * Ho * Wq;
    for (long g = (long)blockIdx.x * kThreads + threadIdx.x; g < total; g += (long)gridDim.x * kThreads) {
        const int xq = (int)(g % Wq);
        const long t = g / Wq;
        const int yo = (int)(t % Ho);
        const long pl = t / Ho;
        const float* r0 = x + (pl * H + 2 * yo) * (long)W + xq * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(r0), a1 = *reinterpret_cast<const float4*>(r0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(r0 + W), b1 = *reinterpret_cast<const float4*>(r0 + W + 4);
        const float t0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, t1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[4]; uint8_t ix[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float m = t0[2 * k]; int w = 0;
            if (t0[2 * k + 1] > m || t0[2 * k + 1] != t0[2 * k + 1]) { m = t0[2 * k + 1]; w = 1; }
            if (t1[2 * k] > m || t1[2 * k] != t1[2 * k]) { m = t1[2 * k]; w = 2; }
            if (t1[2 * k + 1] > m || t1[2 * k + 1] != t1[2 * k + 1]) { m = t1[2 * k + 1]; w = 3; }
            o[k] = m; ix[k] = (uint8_t)w;
        }
        const long oo = (pl * Ho + yo) * (long)Wo + xq * 4;
        *reinterpret_cast<float4*>(out + oo) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uchar4*>(idx + oo) = make_uchar4(ix[0], ix[1], ix[2], ix[3]);
    }
}

// ---- gradient fan-in through the feature perturbations --------------------------------------------------
// An encoder feature map feeds the main decoder as is and every auxiliary decoder through a perturbation
// (UAPS_unet.py:226-232).  Its gradient is  g_main + sum_k P_k^T(g_k); the P_k are diagonal, so one pass can re-apply
// them to the incoming gradients and add, instead of one backward kernel per perturbation plus a sum:
//   mode 0 identity | 1 FeatureNoise: g*n + g, n from Philox(seed, off[group] + element/4) as in noise_rng_vec4
//   mode 2 Dropout: keep from Philox(seed, off[0] + element/4), g * keep / (1-p) | 3 FeatureDropout: g * keep[b,h,w]
//   mode 4 MaxPool2d(2) of the next encoder level: g is [B,C,H/2,W/2], routed to the arg-max position kept in `keep`
// Summation order = input order, the same association the separate kernels + uaps_sum_tensors produced.
// e = q * d + r for the flat element index of the fan kernels: a shift when d is a power of two (`sh` = log2 d, every level of the
// 256 x 256 / 512 x 512 configurations), one 32-bit division when both fit (the 640 x 640 ResNet maps), the 64-bit division otherwise.
// The 64-bit form alone cost the fan kernels a fifth of their time: they are VALU-co-limited by the two Philox calls per element.
__device__ __forceinline__ void fan_divmod(long e, long d, int sh, long& q, long& r) {
    if (sh >= 0) { q = e >> sh; r = e & (d - 1); }
    else if (((unsigned long)e | (unsigned long)d) >> 32 == 0) { const uint32_t q32 = (uint32_t)e / (uint32_t)d; q = q32; r = (uint32_t)e - q32 * (uint32_t)d; }
    else { q = e / d; r = e - q * d; }
}
inline int fan_log2(long d) { return d > 0 && (d & (d - 1)) == 0 ? __builtin_ctzl((unsigned long)d) : -1; }
constexpr int kFanMax = 8, kFanGroups = 4;
static_assert(kFanGroups == 4, "the fan kernels derive the statistics group of an image by three comparisons");
struct FanInArgs {
    const float4* g[kFanMax];
    const uchar4* keep[kFanMax];
    uint64_t off[kFanMax][kFanGroups];
    int mode[kFanMax];
    uint64_t seed;
    float range, p, scale;
    int n, B, Bg;
    long chw4, hw4;
    int W;
    int sh_chw4, sh_hw4, sh_w;      // log2 of chw4 / hw4 / W, or -1 (fan_divmod)
    int Ho;                         // H / 2 (mode 4)
    const uint32_t* st;             // step state (philox.hpp) or NULL
};
__global__ __launch_bounds__(kThreads) void fanin_perturbed_kernel(FanInArgs a, float4* __restrict__ out) {
    const long total = (long)a.B * a.chw4;
    const uint64_t seed = uaps::step_key(a.seed, a.st);
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
        long b, ce;
        fan_divmod(e, a.chw4, a.sh_chw4, b, ce);
        const int ib = (int)b, grp = (ib >= a.Bg) + (ib >= 2 * a.Bg) + (ib >= 3 * a.Bg);      // b / Bg for <= kFanGroups = 4 groups
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < kFanMax; ++k) {
            if (k >= a.n) break;
            float4 v;
            if (a.mode[k] == 4) {               // 4 consecutive pixels of row h <- 2 pooled pixels of row h/2
                long pl, pe4, hh, ww;
                fan_divmod(e, a.hw4, a.sh_hw4, pl, pe4);
                fan_divmod(pe4 * 4, a.W, a.sh_w, hh, ww);
                const int h = (int)hh, w = (int)ww, Wo = a.W / 2;
                const long po = (pl * a.Ho + h / 2) * (long)Wo + w / 2;
                const float2 gp = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(a.g[k]) + po);
                const uchar2 ix = *reinterpret_cast<const uchar2*>(reinterpret_cast<const uint8_t*>(a.keep[k]) + po);
                const int base = (h & 1) * 2;
                v.x = ix.x == base ? gp.x : 0.f; v.y = ix.x == base + 1 ? gp.x : 0.f;
                v.z = ix.y == base ? gp.y : 0.f; v.w = ix.y == base + 1 ? gp.y : 0.f;
            } else {
                v = a.g[k][e];
            }
            if (a.mode[k] == 1) {
                const U4 r = philox4x32_10(a.off[k][grp] + (uint64_t)ce, seed);
                const float n0 = (2.f * u01(r.x) - 1.f) * a.range, n1 = (2.f * u01(r.y) - 1.f) * a.range;
                const float n2 = (2.f * u01(r.z) - 1.f) * a.range, n3 = (2.f * u01(r.w) - 1.f) * a.range;
                v.x = add_rn(mul_rn(v.x, n0), v.x); v.y = add_rn(mul_rn(v.y, n1), v.y);
                v.z = add_rn(mul_rn(v.z, n2), v.z); v.w = add_rn(mul_rn(v.w, n3), v.w);
            } else if (a.mode[k] == 2) {
                const U4 r = philox4x32_10(a.off[k][0] + (uint64_t)e, seed);
                v.x = u01(r.x) >= a.p ? v.x * a.scale : 0.f; v.y = u01(r.y) >= a.p ? v.y * a.scale : 0.f;
                v.z = u01(r.z) >= a.p ? v.z * a.scale : 0.f; v.w = u01(r.w) >= a.p ? v.w * a.scale : 0.f;
            } else if (a.mode[k] == 3) {
                long cc, pix;
                fan_divmod(ce, a.hw4, a.sh_hw4, cc, pix);
                const uchar4 m = a.keep[k][b * a.hw4 + pix];
                v.x = m.x ? v.x : 0.f; v.y = m.y ? v.y : 0.f; v.z = m.z ? v.z : 0.f; v.w = m.w ? v.w : 0.f;
            }
            if (k == 0) acc = v;
            else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        out[e] = acc;
    }
}

// ---- forward counterpart: f -> P_1(f), ..., P_n(f) in ONE pass over f (each separate kernel re-read it) --------------
// The same per-mode arithmetic and Philox indexing as the stand-alone kernels above (and as fanin_perturbed_kernel
// re-applies in the backward): mode 1 FeatureNoise, 2 Dropout, 3 FeatureDropout with the keep mask derived here from
// the attention map and the per-image maximum (fdrop_attention ran before), written once per pixel for the backward.
struct FanOutArgs {
    float4* out[kFanMax];
    uchar4* keep[kFanMax];          // mode 3: keep mask output [B, H*W]
    uint64_t off[kFanMax][kFanGroups];
    int mode[kFanMax];
    float u[kFanGroups];            // mode 3: threshold factor per statistics group
    const float4* att;              // mode 3: [B, H*W] channel means
    const uint32_t* maxkey;         // mode 3: [B] order-preserving keys of the per-image maximum
    uint64_t seed;
    float range, p, scale;
    int n, B, Bg;
    long chw4, hw4;
    int sh_chw4, sh_hw4;            // log2 of chw4 / hw4, or -1 (fan_divmod)
    const uint32_t* st;             // step state (philox.hpp) or NULL
};
__global__ __launch_bounds__(kThreads) void fanout_perturbed_kernel(const float4* __restrict__ f, FanOutArgs a) {
    const long total = (long)a.B * a.chw4;
    const uint64_t seed = uaps::step_key(a.seed, a.st);
    for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
        long b, ce;
        fan_divmod(e, a.chw4, a.sh_chw4, b, ce);
        const int ib = (int)b, grp = (ib >= a.Bg) + (ib >= 2 * a.Bg) + (ib >= 3 * a.Bg);      // b / Bg for <= kFanGroups = 4 groups
        const float4 x = f[e];
#pragma unroll
        for (int k = 0; k < kFanMax; ++k) {
            if (k >= a.n) break;
            float4 v = x;
            if (a.mode[k] == 1) {
                const U4 r = philox4x32_10(a.off[k][grp] + (uint64_t)ce, seed);
                const float n0 = (2.f * u01(r.x) - 1.f) * a.range, n1 = (2.f * u01(r.y) - 1.f) * a.range;
                const float n2 = (2.f * u01(r.z) - 1.f) * a.range, n3 = (2.f * u01(r.w) - 1.f) * a.range;
                v.x = add_rn(mul_rn(v.x, n0), v.x); v.y = add_rn(mul_rn(v.y, n1), v.y);
                v.z = add_rn(mul_rn(v.z, n2), v.z); v.w = add_rn(mul_rn(v.w, n3), v.w);
            } else if (a.mode[k] == 2) {
                const U4 r = philox4x32_10(a.off[k][0] + (uint64_t)e, seed);
                v.x = u01(r.x) >= a.p ? v.x * a.scale : 0.f; v.y = u01(r.y) >= a.p ? v.y * a.scale : 0.f;
                v.z = u01(r.z) >= a.p ? v.z * a.scale : 0.f; v.w = u01(r.w) >= a.p ? v.w * a.scale : 0.f;
            } else if (a.mode[k] == 3) {
                long cc, pix;
                fan_divmod(ce, a.hw4, a.sh_hw4, cc, pix);
                // threshold factor U(0.7, 0.9) (UAPS_unet.py:164): the host's draw, or -- a negative value asks for it -- one
                // Philox draw per (call, statistics group) on the device (captured steps cannot take a new host number)
                const float uf = a.u[grp] >= 0.f ? a.u[grp] : 0.7f + 0.2f * u01(philox4x32_10(a.off[k][grp], seed).x);
                // agent-scope load: the maximum was raised by memory-side atomics of the previous kernel; a plain load may be served
                // from a line this XCD's L2 still holds from the memset before them (seen under hipGraph replay)
                const float thr = mul_rn(fkey_inv(__hip_atomic_load(a.maxkey + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)), uf);
                const float4 t = a.att[b * a.hw4 + pix];
                const bool k0 = t.x < thr, k1 = t.y < thr, k2 = t.z < thr, k3 = t.w < thr;
                if (ce < a.hw4) a.keep[k][b * a.hw4 + pix] = make_uchar4(k0, k1, k2, k3);      // channel 0 writes the mask
                v.x = k0 ? v.x : 0.f; v.y = k1 ? v.y : 0.f; v.z = k2 ? v.z : 0.f; v.w = k3 ? v.w : 0.f;
            }
            a.out[k][e] = v;
        }
    }
}

// ---- confusion matrix ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void confusion_kernel(const float* __restrict__ z, const int64_t* __restrict__ labels, int C,
                                                             long HW, long N, unsigned long long* __restrict__ counts) {
    __shared__ unsigned int local[UAPS_MAX_CLASSES * UAPS_MAX_CLASSES];
    for (int i = threadIdx.x; i < C * C; i += kThreads) local[i] = 0;
    __syncthreads();
    for (long n = (long)blockIdx.x * kThreads + threadIdx.x; n < N; n += (long)gridDim.x * kThreads) {
        const long b = n / HW, hw = n - b * HW;
        const float* p = z + b * C * HW + hw;
        float best = p[0]; int arg = 0;
        for (int c = 1; c < C; ++c) { const float v = p[(long)c * HW]; if (v > best) { best = v; arg = c; } }
        const long y = labels[n];
        if (y >= 0 && y < C) atomicAdd(&local[(int)y * C + arg], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += kThreads) if (local[i]) atomicAdd(counts + i, (unsigned long long)local[i]);
}

}  // namespace

extern "C" int uaps_feat_noise(const float* x, float* y, int B, int C, int H, int W, uint64_t seed, uint64_t offset,
                               float range, float* noise_out, uaps_stream_t stream) {
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    const long chw = (long)C * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (chw % 4 == 0 && al16(x) && al16(y) && (!noise_out || al16(noise_out)))
        hipLaunchKernelGGL(noise_rng_vec4, dim3(grid_for(chw / 4)), dim3(kThreads), 0, s, (const float4*)x, (float4*)y, B, chw / 4, seed, offset, range, (float4*)noise_out);
    else
        hipLaunchKernelGGL(noise_rng_scalar, dim3(grid_for(chw)), dim3(kThreads), 0, s, x, y, B, chw, seed, offset, range, noise_out);
    return (int)hipGetLastError();
}

extern "C" int uaps_feat_noise_apply(const float* x, const float* noise, float* y, int B, long chw, uaps_stream_t stream) {
    if (!x || !y || !noise || B <= 0 || chw <= 0) return UAPS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (chw % 4 == 0 && al16(x) && al16(y) && al16(noise))
        hipLaunchKernelGGL(noise_apply_vec4, dim3(grid_for(chw / 4)), dim3(kThreads), 0, s, (const float4*)x, (const float4*)noise, (float4*)y, B, chw / 4);
    else
        hipLaunchKernelGGL(noise_apply_scalar, dim3(grid_for(chw)), dim3(kThreads), 0, s, x, noise, y, B, chw);
    return (int)hipGetLastError();
}

extern "C" int uaps_feat_bernoulli(const float* x, float* y, long n, uint64_t seed, uint64_t offset, float p,
                                   uint8_t* keep_out, uaps_stream_t stream) {
    if (!x || !y || n <= 0 || !(p >= 0.f && p < 1.f)) return UAPS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.f / (1.f - p);
    long done = 0;
    if (al16(x) && al16(y) && (!keep_out || (reinterpret_cast<uintptr_t>(keep_out) & 3) == 0) && n >= 4) {
        const long n4 = n / 4;
        hipLaunchKernelGGL(bernoulli_vec4, dim3(grid_for(n4)), dim3(kThreads), 0, s, (const float4*)x, (float4*)y, n4, seed, offset, p, scale, (uchar4*)keep_out);
        done = n4 * 4;
    }
    if (done < n)
        hipLaunchKernelGGL(bernoulli_scalar, dim3(grid_for(n - done)), dim3(kThreads), 0, s, x, y, done, n, seed, offset, p, scale, keep_out);
    return (int)hipGetLastError();
}

extern "C" int uaps_feat_mask_apply(const float* x, const uint8_t* keep, float scale, float* y, long n, uaps_stream_t stream) {
    if (!x || !y || !keep || n <= 0) return UAPS_EINVAL;
    hipLaunchKernelGGL(mask_apply_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, x, keep, scale, y, n);
    return (int)hipGetLastError();
}

extern "C" int uaps_feat_dropout_workspace_bytes(int B, int C, int H, int W, size_t* out) {
    if (!out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    *out = (((size_t)B * 4 + 255) / 256) * 256 + (size_t)B * H * W * 4;
    return UAPS_OK;
}

extern "C" int uaps_feat_dropout_fwd(const float* x, float* y, int B, int C, int H, int W, float u, uint8_t* keep,
                                     void* ws, size_t ws_bytes, uaps_stream_t stream) {
    if (!x || !y || !keep || !ws || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    size_t need; uaps_feat_dropout_workspace_bytes(B, C, H, W, &need);
    if (ws_bytes < need) return UAPS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long HW = (long)H * W;
    uint32_t* maxkey = (uint32_t*)ws;
    float* att = (float*)((char*)ws + (((size_t)B * 4 + 255) / 256) * 256);
    uaps::account_bytes(4.0 * B * HW * (C + 1.0));       // x once, the attention map once
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, maxkey, (long)B);      // a kernel, not hipMemsetAsync: see zero_words_kernel
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const bool vec = (HW % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)keep % 4 == 0);
    int gx = grid_for(vec ? HW / 4 : HW); if ((long)gx * B > 4096) gx = (int)((4096 + B - 1) / B);
    if (vec) {
        hipLaunchKernelGGL(fdrop_attention<4>, dim3(gx, B), dim3(kThreads), 0, s, x, C, HW, att, maxkey);
        hipLaunchKernelGGL(fdrop_apply<4>, dim3(gx, B), dim3(kThreads), 0, s, x, y, C, HW, att, maxkey, u, keep);
    } else {
        hipLaunchKernelGGL(fdrop_attention<1>, dim3(gx, B), dim3(kThreads), 0, s, x, C, HW, att, maxkey);
        hipLaunchKernelGGL(fdrop_apply<1>, dim3(gx, B), dim3(kThreads), 0, s, x, y, C, HW, att, maxkey, u, keep);
    }
    return (int)hipGetLastError();
}

extern "C" int uaps_feat_dropout_bwd(const float* dy, const uint8_t* keep, float* dx, int B, int C, int H, int W,
                                     uaps_stream_t stream) {
    if (!dy || !dx || !keep || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    const long HW = (long)H * W;
    const bool vec = (HW % 4 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0) && ((uintptr_t)keep % 4 == 0);
    int gx = grid_for(vec ? HW / 4 : HW); if ((long)gx * B > 4096) gx = (int)((4096 + B - 1) / B);
    if (vec) hipLaunchKernelGGL(fdrop_bwd<4>, dim3(gx, B), dim3(kThreads), 0, (hipStream_t)stream, dy, keep, dx, C, HW);
    else hipLaunchKernelGGL(fdrop_bwd<1>, dim3(gx, B), dim3(kThreads), 0, (hipStream_t)stream, dy, keep, dx, C, HW);
    return (int)hipGetLastError();
}

// FeatureDropout statistics for a whole batch: att[b,hw] = mean_c x[b,c,hw] and the per-image maximum key, into the
// workspace of uaps_feat_dropout_workspace_bytes(B, ...) (what uaps_feat_dropout_fwd computes before it applies the mask).
extern "C" int uaps_feat_dropout_stats(const float* x, int B, int C, int H, int W, void* ws, size_t ws_bytes, uaps_stream_t stream) {
    if (!x || !ws || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    size_t need; uaps_feat_dropout_workspace_bytes(B, C, H, W, &need);
    if (ws_bytes < need) return UAPS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long HW = (long)H * W;
    uint32_t* maxkey = (uint32_t*)ws;
    float* att = (float*)((char*)ws + (((size_t)B * 4 + 255) / 256) * 256);
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, maxkey, (long)B);      // a kernel, not hipMemsetAsync: see zero_words_kernel
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const bool vec = (HW % 4 == 0) && ((uintptr_t)x % 16 == 0);
    if (vec && C >= 64 && C % 4 == 0 && HW <= 4096) {      // deep levels: split the channels over the waves of a block
        hipLaunchKernelGGL(fdrop_attention_split, dim3((unsigned)((HW / 4 + 63) / 64), B), dim3(kThreads), 0, s, x, C, HW, att, maxkey);
        return (int)hipGetLastError();
    }
    int gx = grid_for(vec ? HW / 4 : HW); if ((long)gx * B > 2048) gx = (int)((2048 + B - 1) / B);
    if (vec) hipLaunchKernelGGL(fdrop_attention<4>, dim3(gx, B), dim3(kThreads), 0, s, x, C, HW, att, maxkey);
    else hipLaunchKernelGGL(fdrop_attention<1>, dim3(gx, B), dim3(kThreads), 0, s, x, C, HW, att, maxkey);
    return (int)hipGetLastError();
}

// f [B,C,H,W] -> n perturbed copies in one pass (see fanout_perturbed_kernel): host arrays of n <= 8 entries: out (device
// pointers), mode (1 FeatureNoise, 2 Dropout, 3 FeatureDropout), keep (device uint8 [B,H,W] outputs for mode 3, else
// NULL), offsets [n][groups] (mode 1: one Philox offset per statistics group; mode 2: offsets[k*groups]); u [groups]: the
// FeatureDropout threshold factors; fdrop_ws: the workspace uaps_feat_dropout_stats filled for this f (NULL without
// mode 3).  Needs H*W % 4 == 0, 16-byte aligned tensors, groups <= 4 (else UAPS_EINVAL: use the separate kernels).
extern "C" int uaps_fanout_perturbed(const float* f, float* const* out, const int* mode, uint8_t* const* keep,
                                     const uint64_t* offsets, const float* u, const void* fdrop_ws, int n, int groups, uint64_t seed,
                                     float range, float p, int B, int C, int H, int W, uaps_stream_t stream) {
    if (!f || !out || !mode || n < 1 || n > kFanMax || groups < 1 || groups > kFanGroups || B <= 0 || C <= 0 || H <= 0 || W <= 0 ||
        B % groups)
        return UAPS_EINVAL;
    const long HW = (long)H * W;
    if (HW % 4 || !al16(f)) return UAPS_EINVAL;
    FanOutArgs a{};
    for (int k = 0; k < n; ++k) {
        if (!out[k] || !al16(out[k]) || mode[k] < 1 || mode[k] > 3) return UAPS_EINVAL;
        if (mode[k] == 3 && (!keep || !keep[k] || (reinterpret_cast<uintptr_t>(keep[k]) & 3) || !u || !fdrop_ws)) return UAPS_EINVAL;
        if (mode[k] != 3 && !offsets) return UAPS_EINVAL;
        a.out[k] = (float4*)out[k]; a.mode[k] = mode[k]; a.keep[k] = keep ? (uchar4*)keep[k] : nullptr;
        for (int q = 0; q < groups; ++q) a.off[k][q] = offsets ? offsets[(size_t)k * groups + q] : 0;
    }
    for (int q = 0; q < groups; ++q) a.u[q] = u ? u[q] : 0.f;
    if (fdrop_ws) {
        a.maxkey = (const uint32_t*)fdrop_ws;
        a.att = (const float4*)((const char*)fdrop_ws + (((size_t)B * 4 + 255) / 256) * 256);
    }
    a.seed = seed; a.range = range; a.p = p; a.scale = 1.f / (1.f - p); a.n = n; a.B = B; a.Bg = B / groups;
    a.st = (const uint32_t*)uaps_get_step_state();
    a.chw4 = (long)C * HW / 4; a.hw4 = HW / 4;
    a.sh_chw4 = fan_log2(a.chw4); a.sh_hw4 = fan_log2(a.hw4);
    {   // f once, every perturbed copy once (+ the FeatureDropout keep masks, one byte per pixel)
        double by = 4.0 * B * C * HW * (1.0 + n);
        for (int k = 0; k < n; ++k) if (mode[k] == 3) by += (double)B * HW;
        uaps::account_bytes(by);
    }
    hipLaunchKernelGGL(fanout_perturbed_kernel, dim3(grid_for((long)B * a.chw4)), dim3(kThreads), 0, (hipStream_t)stream, (const float4*)f, a);
    return (int)hipGetLastError();
}

// out = in[0] + in[1] + ... + in[n-1] (left to right), n in [1,4], `count` floats each; out may alias in[0].
extern "C" int uaps_sum_tensors(const float* const* in, int n, float* out, long count, uaps_stream_t stream) {
    if (!in || !out || n < 1 || n > 4 || count <= 0) return UAPS_EINVAL;
    SumPtrs sp{};
    bool al = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    for (int k = 0; k < n; ++k) { if (!in[k]) return UAPS_EINVAL; sp.p[k] = in[k]; al = al && (reinterpret_cast<uintptr_t>(in[k]) & 15) == 0; }
    const long n4 = al ? count / 4 : 0;
    hipLaunchKernelGGL(sum_n_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kThreads), 0, (hipStream_t)stream, sp, n, out, n4, count);
    return (int)hipGetLastError();
}

// out[B,C,H,W] = sum_k P_k(g_k), k < n <= 8 (see fanin_perturbed_kernel).  Host arrays of n entries: g (device
// pointers), mode, keep (device uint8 [B,H,W] for mode 3, else NULL), offsets (n x groups, row-major; mode 1 uses one
// Philox offset per statistics group of B/groups images, mode 2 uses offsets[k*groups]).  Needs H*W % 4 == 0 and
// 16-byte aligned tensors (returns UAPS_EINVAL otherwise: use the separate kernels), groups <= 4.
extern "C" int uaps_fanin_perturbed(const float* const* g, const int* mode, const uint8_t* const* keep, const uint64_t* offsets,
                                    int n, int groups, uint64_t seed, float range, float p, int B, int C, int H, int W, float* out,
                                    uaps_stream_t stream) {
    if (!g || !mode || !out || n < 1 || n > kFanMax || groups < 1 || groups > kFanGroups || B <= 0 || C <= 0 || H <= 0 || W <= 0 ||
        B % groups)
        return UAPS_EINVAL;
    const long HW = (long)H * W;
    if (HW % 4 || !al16(out)) return UAPS_EINVAL;
    FanInArgs a{};
    for (int k = 0; k < n; ++k) {
        if (!g[k] || !al16(g[k]) || mode[k] < 0 || mode[k] > 4) return UAPS_EINVAL;
        if ((mode[k] == 3 || mode[k] == 4) && (!keep || !keep[k] || (reinterpret_cast<uintptr_t>(keep[k]) & 3))) return UAPS_EINVAL;
        if (mode[k] == 4 && (H % 2 || W % 4)) return UAPS_EINVAL;
        if ((mode[k] == 1 || mode[k] == 2) && !offsets) return UAPS_EINVAL;
        a.g[k] = (const float4*)g[k]; a.mode[k] = mode[k]; a.keep[k] = keep ? (const uchar4*)keep[k] : nullptr;
        for (int q = 0; q < groups; ++q) a.off[k][q] = offsets ? offsets[(size_t)k * groups + q] : 0;
    }
    a.seed = seed; a.range = range; a.p = p; a.scale = 1.f / (1.f - p); a.n = n; a.B = B; a.Bg = B / groups;
    a.st = (const uint32_t*)uaps_get_step_state();
    a.chw4 = (long)C * HW / 4; a.hw4 = HW / 4; a.W = W; a.Ho = H / 2;
    a.sh_chw4 = fan_log2(a.chw4); a.sh_hw4 = fan_log2(a.hw4); a.sh_w = fan_log2(W);
    {   // every incoming gradient once (mode 4: the pooled gradient and its index bytes), the keep masks, the sum once
        double by = 4.0 * B * C * HW;
        for (int k = 0; k < n; ++k) by += mode[k] == 4 ? 5.0 * B * C * HW / 4 : 4.0 * B * C * HW + (mode[k] == 3 ? (double)B * HW : 0.0);
        uaps::account_bytes(by);
    }
    hipLaunchKernelGGL(fanin_perturbed_kernel, dim3(grid_for((long)B * a.chw4)), dim3(kThreads), 0, (hipStream_t)stream, a, (float4*)out);
    return (int)hipGetLastError();
}

// MaxPool2d(kernel 2, stride 2) on [B,C,H,W] (H even, W % 8 == 0, 16-byte aligned): out [B,C,H/2,W/2] and idx uint8 of the
// same shape (position dy*2+dx of the maximum; first maximum wins, NaN propagates, as torch.nn.MaxPool2d).  The backward is
// mode 4 of uaps_fanin_perturbed.
extern "C" int uaps_maxpool2x2_fwd(const float* x, int B, int C, int H, int W, float* out, uint8_t* idx, uaps_stream_t stream) {
    if (!x || !out || !idx || B <= 0 || C <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (H % 2 || W % 8 || !al16(x) || !al16(out) || (reinterpret_cast<uintptr_t>(idx) & 3)) return UAPS_EINVAL;
    const long planes = (long)B * C;
    uaps::account_bytes((double)planes * H * W * (4.0 + 5.0 / 4));      // x once, the pooled map and its index bytes once
    hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3(grid_for(planes * (H / 2) * (W / 8))), dim3(kThreads), 0, (hipStream_t)stream, x, out,
                       idx, planes, H, W);
    return (int)hipGetLastError();
}

extern "C" int uaps_seg_confusion(const float* logits, const int64_t* labels, int B, int C, int H, int W, int64_t* counts,
                                  uaps_stream_t stream) {
    if (!logits || !labels || !counts || B <= 0 || H <= 0 || W <= 0) return UAPS_EINVAL;
    if (C < 1 || C > UAPS_MAX_CLASSES) return UAPS_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    const long HW = (long)H * W, N = (long)B * HW;
    uaps::account_bytes((double)N * (4.0 * C + 8.0));
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((2 * C * C + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint32_t*>(counts), (long)2 * C * C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(N) > 512 ? 512 : grid_for(N)), dim3(kThreads), 0, s, logits, labels, C, HW, N,
                       (unsigned long long*)counts);
    return (int)hipGetLastError();
}
