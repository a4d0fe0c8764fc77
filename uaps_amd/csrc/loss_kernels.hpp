// Fused loss-block kernels of the UAPS step for gfx950 (wave64, HBM-bound streaming kernels).
//
// One thread owns VEC horizontally adjacent pixels and keeps all D*C logits of each in registers:
// with the reference's NCHW fp32 logits a (head, class) plane is contiguous over pixels, so a lane
// reads 16 B (VEC=4) from each of the D*C planes and a wave reads 1 KiB per plane, fully coalesced.
// The class softmax is therefore register-local (no cross-lane traffic); cross-lane work is only
// the block reduction of the per-thread partial sums (wave64 DPP/shuffle tree, then LDS across the
// 4 waves), written as one row of partials per block and reduced in a fixed order by a one-block
// finalize kernel (double accumulation) -- no float atomics, so results are bitwise reproducible
// run to run.
//
// Reference lines restated: UAPS_train.py:186-189, 194-218, 223-277, 282; pytorch_losses.py:81-89.
#pragma once
#include "hints.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "rn_math.hpp"
#include "philox.hpp"

namespace uaps {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;
constexpr int kFinalizeThreads = 1024;   // 16 waves: 3 of the <= 48 sums each

template <int D> struct HeadPtrs { const float* p[D]; };
template <int D> struct HeadOutPtrs { float* p[D]; };
template <int D> struct HeadWeights { float w[D]; };

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC> __device__ __forceinline__ void load_vec(const float* p, float (&out)[VEC]) {
    using V = typename VecT<VEC>::type;
    V v = *reinterpret_cast<const V*>(p);
    if constexpr (VEC == 1) { out[0] = v; }
    if constexpr (VEC == 2) { out[0] = v.x; out[1] = v.y; }
    if constexpr (VEC == 4) { out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
}
template <int VEC> __device__ __forceinline__ void store_vec(float* p, const float (&in)[VEC]) {
    using V = typename VecT<VEC>::type;
    V v;
    if constexpr (VEC == 1) { v = in[0]; }
    if constexpr (VEC == 2) { v.x = in[0]; v.y = in[1]; }
    if constexpr (VEC == 4) { v.x = in[0]; v.y = in[1]; v.z = in[2]; v.w = in[3]; }
    *reinterpret_cast<V*>(p) = v;
}
template <int VEC> __device__ __forceinline__ void load_labels(const int64_t* p, int (&out)[VEC]) {
    if constexpr (VEC == 1) { out[0] = (int)p[0]; }
    if constexpr (VEC == 2) { longlong2 v = *reinterpret_cast<const longlong2*>(p); out[0] = (int)v.x; out[1] = (int)v.y; }
    if constexpr (VEC == 4) {
        longlong2 a = *reinterpret_cast<const longlong2*>(p), b = *reinterpret_cast<const longlong2*>(p + 2);
        out[0] = (int)a.x; out[1] = (int)a.y; out[2] = (int)b.x; out[3] = (int)b.y;
    }
}
template <int VEC> __device__ __forceinline__ void store_labels(int64_t* p, const int (&in)[VEC]) {
    if constexpr (VEC == 1) { p[0] = in[0]; }
    if constexpr (VEC == 2) { longlong2 v; v.x = in[0]; v.y = in[1]; *reinterpret_cast<longlong2*>(p) = v; }
    if constexpr (VEC == 4) {
        longlong2 a, b; a.x = in[0]; a.y = in[1]; b.x = in[2]; b.y = in[3];
        *reinterpret_cast<longlong2*>(p) = a; *reinterpret_cast<longlong2*>(p + 2) = b;
    }
}

// v_exp_f32 / v_log_f32 / v_rcp_f32 based forms: ~1-2 ulp, far inside the 1e-4 parity budget and
// one quarter-rate instruction each instead of libm's ~15-instruction sequences (these kernels are
// VALU/HBM co-limited: ~30 transcendentals per pixel at D=C=4).
__device__ __forceinline__ float fexp(float x) { return __expf(x); }
__device__ __forceinline__ float flog(float x) { return __logf(x); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

// softmax over C register values: p, log p, all from one max/exp/sum pass.
template <int C> __device__ __forceinline__ void softmax_regs(const float (&z)[C], float (&p)[C], float (&lp)[C]) {
    float mx = z[0];
#pragma unroll
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = fexp(z[c] - mx); s += p[c]; }
    const float inv = frcp(s), lse = mx + flog(s);
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] *= inv; lp[c] = z[c] - lse; }
}

// Sum over the 64 lanes of a wave, result valid in every lane.  Within each row of 16 lanes the adds are DPP
// modifiers on v_add_f32 (quad_perm xor 1, xor 2, row_half_mirror, row_mirror): no LDS traffic and no waits, unlike
// __shfl_xor, which lowers to ds_bpermute_b32 + s_waitcnt per step (the 48 sums of the unsupervised forward cost 288
// dependent LDS round trips per thread that way and dominated the kernel).  The four row sums are then read with
// v_readlane and added in a fixed order.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);    // row_half_mirror
    v += dpp_f32<0x140>(v);    // row_mirror: every lane of a 16-lane row now holds the row sum
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// Block reduction of NS per-thread partial sums into column `blockIdx.x` of `partials` ([NS][gridDim.x], sum-major so
// that the finalize kernel reads each sum's block partials as one coalesced run).
template <int NS> __device__ __forceinline__ void block_reduce_store(float (&acc)[NS], float* partials, int bid, int nblk) {
    __shared__ float red[kThreads / 64][NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NS; i += kThreads) {
        float s = red[0][i];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) s += red[w][i];
        partials[(size_t)i * nblk + bid] = s;
    }
}

// --------------------------------------------------------------------------------------------------
// raw-sum layouts (floats per block row / doubles in the finalize)
//   unsup: CE[D] | I[D*C] | P[D*C] | cnt[C] | E[D] | V[D]
//   sup:   CE[D] | I[D*C] | P[D*C] | cnt[C] | bad
// --------------------------------------------------------------------------------------------------
template <int D, int C> struct UnsupLayout {
    static constexpr int CE = 0, I = D, P = D + D * C, CNT = D + 2 * D * C, E = CNT + C, V = E + D, NS = V + D;
};
template <int D, int C> struct SupLayout {
    static constexpr int CE = 0, I = D, P = D + D * C, CNT = D + 2 * D * C, BAD = CNT + C, NS = BAD + 1;
};

// --------------------------------------------------------------------------------------------------
// F1: unsupervised forward.  Algorithmic HBM bytes per pixel: 4DC (logits) + 8 (pseudo) + 4D (var).
// --------------------------------------------------------------------------------------------------
// (bid, nblk): this block's index among the nblk blocks that share the work -- blockIdx.x / gridDim.x for the stand-alone
// kernels, a sub-range of the grid for the pair kernels below
// the per-pixel arithmetic of one group of VEC pixels: zv -> pseudo-labels yv, variance maps varv, running sums acc
template <int D, int C, int VEC>
__device__ __forceinline__ void unsup_fwd_group(const float (&zv)[D][C][VEC], const HeadWeights<D>& w, float (&acc)[UnsupLayout<D, C>::NS],
                                                int (&yv)[VEC], float (&varv)[D][VEC]) {
    using L = UnsupLayout<D, C>;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float p[D][C], lp[D][C], m[C];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float zz[C];
#pragma unroll
            for (int c = 0; c < C; ++c) zz[c] = zv[k][c][v];
            softmax_regs<C>(zz, p[k], lp[k]);
        }
        // mean prediction (UAPS_train.py:223) and the mixture that feeds arg-max (:252-255):
        // separate multiply and add roundings, left to right, like the reference's tensor ops.
        float xm = 0.f;
        int y = 0;
        float best = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float s = p[0][c];
            float mix = mul_rn(w.w[0], p[0][c]);
#pragma unroll
            for (int k = 1; k < D; ++k) { s = add_rn(s, p[k][c]); mix = add_rn(mix, mul_rn(w.w[k], p[k][c])); }
            m[c] = s / (float)D;
            xm += (m[c] > 0.f) ? m[c] * flog(m[c]) : 0.f;           // xlogy(m, m)
            if (c == 0 || mix > best) { best = mix; y = c; }          // first maximum wins, as torch.argmax
        }
        yv[v] = y;
        float oh[C];                                 // one-hot of the pseudo-label: one compare + select per class, then plain
#pragma unroll                                       // multiply-adds (p * 1 and p * 0 are exact, so the sums are unchanged)
        for (int c = 0; c < C; ++c) { oh[c] = (y == c) ? 1.f : 0.f; acc[L::CNT + c] += oh[c]; }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float dot = 0.f, lpy = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dot += m[c] * lp[k][c];
                acc[L::P + k * C + c] += p[k][c];
                acc[L::I + k * C + c] = __builtin_fmaf(oh[c], p[k][c], acc[L::I + k * C + c]);
                lpy = __builtin_fmaf(oh[c], lp[k][c], lpy);
            }
            const float vk = xm - dot;                               // sum_c KL(m || p_k)  (:226)
            varv[k][v] = vk;
            acc[L::V + k] += vk;
            acc[L::E + k] += fexp(-vk);                              // :227
            acc[L::CE + k] -= lpy;
        }
    }
}

template <int D, int C, int VEC>
__device__ __forceinline__ void unsup_fwd_body(const HeadPtrs<D>& z, const HeadWeights<D>& w, int HW, long ngroups,
                                               long N, int64_t* __restrict__ pseudo,
                                               float* __restrict__ var, float* __restrict__ partials, int bid, int nblk) {
    using L = UnsupLayout<D, C>;
    float acc[L::NS];
#pragma unroll
    for (int i = 0; i < L::NS; ++i) acc[i] = 0.f;

    for (long g = (long)bid * kThreads + threadIdx.x; g < ngroups; g += (long)nblk * kThreads) {
        const long n0 = g * VEC;
        const long b = n0 / HW;
        const long hw = n0 - b * HW;
        const long base = b * C * (long)HW + hw;
        float zv[D][C][VEC];
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) load_vec<VEC>(z.p[k] + base + (long)c * HW, zv[k][c]);
        int yv[VEC];
        float varv[D][VEC];
        unsup_fwd_group<D, C, VEC>(zv, w, acc, yv, varv);
        store_labels<VEC>(pseudo + n0, yv);
        if (var != nullptr) {
#pragma unroll
            for (int k = 0; k < D; ++k) store_vec<VEC>(var + (long)k * N + n0, varv[k]);
        }
    }
    block_reduce_store<L::NS>(acc, partials, bid, nblk);
}

// The same with the logits of the thread's NEXT group fetched while the current group is computed (two register sets, the
// loop body written out for both): these kernels hold ~200 registers (48 running sums + the D*C*VEC logits), i.e. two waves
// per SIMD, and a thread that loads, waits, computes and only then loads again leaves HBM idle for the ~1700 instructions
// of a group.  For the persistent pair kernels (a few groups per thread).
template <int D, int C, int VEC>
__device__ __forceinline__ void unsup_fwd_body_pf(const HeadPtrs<D>& z, const HeadWeights<D>& w, int HW, long ngroups,
                                                  long N, int64_t* __restrict__ pseudo,
                                                  float* __restrict__ var, float* __restrict__ partials, int bid, int nblk) {
    using L = UnsupLayout<D, C>;
    float acc[L::NS];
#pragma unroll
    for (int i = 0; i < L::NS; ++i) acc[i] = 0.f;
    const long stride = (long)nblk * kThreads;
    auto load_group = [&](long g, float (&zv)[D][C][VEC]) {
        const long n0 = g * VEC, b = n0 / HW, hw = n0 - b * HW, base = b * C * (long)HW + hw;
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) load_vec<VEC>(z.p[k] + base + (long)c * HW, zv[k][c]);
    };
    auto finish_group = [&](long g, const float (&zv)[D][C][VEC]) {
        int yv[VEC];
        float varv[D][VEC];
        unsup_fwd_group<D, C, VEC>(zv, w, acc, yv, varv);
        const long n0 = g * VEC;
        store_labels<VEC>(pseudo + n0, yv);
        if (var != nullptr) {
#pragma unroll
            for (int k = 0; k < D; ++k) store_vec<VEC>(var + (long)k * N + n0, varv[k]);
        }
    };
    float za[D][C][VEC], zb[D][C][VEC];
    long g = (long)bid * kThreads + threadIdx.x;
    if (g < ngroups) load_group(g, za);
    while (g < ngroups) {
        if (g + stride < ngroups) load_group(g + stride, zb);
        finish_group(g, za);
        g += stride;
        if (g >= ngroups) break;
        if (g + stride < ngroups) load_group(g + stride, za);
        finish_group(g, zb);
        g += stride;
    }
    block_reduce_store<L::NS>(acc, partials, bid, nblk);
}
template <int D, int C, int VEC>
__global__ __launch_bounds__(kThreads) void unsup_fwd_kernel(HeadPtrs<D> z, HeadWeights<D> w, int HW, long ngroups,
                                                             long N, int64_t* __restrict__ pseudo,
                                                             float* __restrict__ var, float* __restrict__ partials) {
    unsup_fwd_body<D, C, VEC>(z, w, HW, ngroups, N, pseudo, var, partials, (int)blockIdx.x, (int)gridDim.x);
}

// --------------------------------------------------------------------------------------------------
// F3 forward: supervised branch.  Bytes per pixel: 4DC + 8.
// --------------------------------------------------------------------------------------------------
template <int D, int C, int VEC>
__device__ __forceinline__ void sup_fwd_body(const HeadPtrs<D>& z, int HW, long ngroups,
                                             const int64_t* __restrict__ labels,
                                             float* __restrict__ partials, int bid, int nblk) {
    using L = SupLayout<D, C>;
    float acc[L::NS];
#pragma unroll
    for (int i = 0; i < L::NS; ++i) acc[i] = 0.f;
    for (long g = (long)bid * kThreads + threadIdx.x; g < ngroups; g += (long)nblk * kThreads) {
        const long n0 = g * VEC;
        const long b = n0 / HW;
        const long hw = n0 - b * HW;
        const long base = b * C * (long)HW + hw;
        int yv[VEC];
        load_labels<VEC>(labels + n0, yv);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float zv[C][VEC];
#pragma unroll
            for (int c = 0; c < C; ++c) load_vec<VEC>(z.p[k] + base + (long)c * HW, zv[c]);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float zz[C], p[C], lp[C];
#pragma unroll
                for (int c = 0; c < C; ++c) zz[c] = zv[c][v];
                softmax_regs<C>(zz, p, lp);
                const int y = yv[v];
                float lpy = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    acc[L::P + k * C + c] += p[c];
                    acc[L::I + k * C + c] += (y == c) ? p[c] : 0.f;
                    lpy = (y == c) ? lp[c] : lpy;
                }
                acc[L::CE + k] -= lpy;
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int y = yv[v];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[L::CNT + c] += (y == c) ? 1.f : 0.f;
            acc[L::BAD] += (y < 0 || y >= C) ? 1.f : 0.f;
        }
    }
    block_reduce_store<L::NS>(acc, partials, bid, nblk);
}
// The supervised forward with the next head's logits (and, behind the last head, the next group's labels and first head)
// fetched while the current head is computed: the pair kernel runs this branch at the register budget of the
// unsupervised one (two waves per SIMD), where the plain loop above would wait out every load.
template <int D, int C, int VEC>
__device__ __forceinline__ void sup_fwd_body_pf(const HeadPtrs<D>& z, int HW, long ngroups,
                                                const int64_t* __restrict__ labels,
                                                float* __restrict__ partials, int bid, int nblk) {
    using L = SupLayout<D, C>;
    float acc[L::NS];
#pragma unroll
    for (int i = 0; i < L::NS; ++i) acc[i] = 0.f;
    const long stride = (long)nblk * kThreads;
    auto load_head = [&](long g, int k, float (&zv)[C][VEC]) {
        const long n0 = g * VEC, b = n0 / HW, hw = n0 - b * HW, base = b * C * (long)HW + hw;
#pragma unroll
        for (int c = 0; c < C; ++c) load_vec<VEC>(z.p[k] + base + (long)c * HW, zv[c]);
    };
    auto head = [&](const float (&zv)[C][VEC], const int (&yv)[VEC], int k) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float zz[C], p[C], lp[C];
#pragma unroll
            for (int c = 0; c < C; ++c) zz[c] = zv[c][v];
            softmax_regs<C>(zz, p, lp);
            const int y = yv[v];
            float lpy = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                acc[L::P + k * C + c] += p[c];
                acc[L::I + k * C + c] += (y == c) ? p[c] : 0.f;
                lpy = (y == c) ? lp[c] : lpy;
            }
            acc[L::CE + k] -= lpy;
        }
    };
    // buffers alternate head by head; PAR = which buffer holds head 0 of this group (flips per group when D is odd)
    float za[C][VEC], zb[C][VEC];
    int yv[VEC], yn[VEC];
    auto group = [&](long g, auto par) {
        constexpr int PAR = decltype(par)::value;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const bool cur_a = ((k + PAR) & 1) == 0;
            if (k + 1 < D) {
                if (cur_a) load_head(g, k + 1, zb); else load_head(g, k + 1, za);
            } else if (g + stride < ngroups) {
                load_labels<VEC>(labels + (g + stride) * VEC, yn);
                if (cur_a) load_head(g + stride, 0, zb); else load_head(g + stride, 0, za);
            }
            if (cur_a) head(za, yv, k); else head(zb, yv, k);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int y = yv[v];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[L::CNT + c] += (y == c) ? 1.f : 0.f;
            acc[L::BAD] += (y < 0 || y >= C) ? 1.f : 0.f;
            yv[v] = yn[v];
        }
    };
    long g = (long)bid * kThreads + threadIdx.x;
    if (g < ngroups) { load_labels<VEC>(labels + g * VEC, yv); load_head(g, 0, za); }
#pragma unroll
    for (int v = 0; v < VEC; ++v) yn[v] = 0;
    while (g < ngroups) {
        group(g, std::integral_constant<int, 0>{});
        g += stride;
        if (g >= ngroups) break;
        group(g, std::integral_constant<int, D & 1>{});      // odd D: head 0 of every other group sits in the second buffer
        g += stride;
    }
    block_reduce_store<L::NS>(acc, partials, bid, nblk);
}
template <int D, int C, int VEC>
__global__ __launch_bounds__(kThreads) void sup_fwd_kernel(HeadPtrs<D> z, int HW, long ngroups,
                                                           const int64_t* __restrict__ labels,
                                                           float* __restrict__ partials) {
    sup_fwd_body<D, C, VEC>(z, HW, ngroups, labels, partials, (int)blockIdx.x, (int)gridDim.x);
}

// --------------------------------------------------------------------------------------------------
// finalize: fixed-order double reduction of the block rows, then the scalar losses and the Dice
// gradient coefficients the backward kernels need.  One block.
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int kMaxSums = 4 * UAPS_MAX_HEADS + 2 * UAPS_MAX_HEADS * UAPS_MAX_CLASSES + UAPS_MAX_CLASSES + 1;

// fixed-order double reduction of the block partials [NS][nrows] into tot[NS]: kFinalizeThreads / 64 waves, one sum per wave at
// a time: coalesced reads of that sum's nrows block partials, lane-strided double accumulation, fixed-order wave reduction
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partials, int nrows, int NS, double* tot) {
    // 8 lanes per sum: lane j of the group adds rows j, j + 8, ... with four independent double chains (all loads of a thread
    // are independent: one trip through the memory pipeline instead of one per sum), then the 8 lanes are combined in a
    // fixed order.  1024 threads = 128 groups >= the 89 sums of D = C = 8.
    const int grp = threadIdx.x >> 3, j = threadIdx.x & 7;
    for (int i = grp; i < NS; i += kFinalizeThreads / 8) {
        const float* src = partials + (size_t)i * nrows;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int r = j;
        for (; r + 24 < nrows; r += 32) {
            s0 += (double)src[r]; s1 += (double)src[r + 8]; s2 += (double)src[r + 16]; s3 += (double)src[r + 24];
        }
        for (; r < nrows; r += 8) s0 += (double)src[r];
        double s = (s0 + s1) + (s2 + s3);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (j == 0) tot[i] = s;
    }
}

// tot[NS] (shared, already complete: call after a barrier) -> the scalar losses and the Dice gradient coefficients the
// backward kernels need.  N = number of pixels the sums run over (the local batch, or the gathered global batch when the
// sums were exchanged between ranks).  Ends with a barrier-free tail: only thread 0 writes the loss scalars.
template <bool UNSUP>
__device__ __forceinline__ void finalize_from_tot(const double* tot, double* dice_s, int D, int C, long N, float cw1, float cw2,
                                                  float eps, float* __restrict__ out) {
    const double *CE = tot, *I = tot + D, *P = I + D * C, *cnt = P + D * C;
    const int oA1 = UNSUP ? UAPS_U_A1(D, C) : UAPS_S_A1(D, C);
    const int oA2 = UNSUP ? UAPS_U_A2(D, C) : UAPS_S_A2(D, C);
    const int oI = UNSUP ? UAPS_U_I(D, C) : UAPS_S_I(D, C);
    const int oCard = UNSUP ? UAPS_U_CARD(D, C) : UAPS_S_CARD(D, C);
    const int oCnt = UNSUP ? UAPS_U_CNT(D, C) : UAPS_S_CNT(D, C);
    if ((int)threadIdx.x < D * C) {
        const int t = threadIdx.x, c = t % C;
        const double card = P[t] + cnt[c];
        const double den = card + (double)eps;
        out[oA1 + t] = (float)(-(2.0 / C) / den);
        out[oA2 + t] = (float)((2.0 / C) * I[t] / (den * den));
        out[oI + t] = (float)I[t];
        out[oCard + t] = (float)card;
    }
    if ((int)threadIdx.x < C) out[oCnt + threadIdx.x] = (float)cnt[threadIdx.x];
    if ((int)threadIdx.x < D) {
        const int k = threadIdx.x;
        double ds = 0.0;
        for (int c = 0; c < C; ++c) ds += 2.0 * I[k * C + c] / (P[k * C + c] + cnt[c] + (double)eps);
        dice_s[k] = 1.0 - ds / C;                                   // pytorch_losses.py:88-89
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (UNSUP) {
            const double *E = cnt + C, *V = E + D;
            double ps = 0.0, lu = 0.0;
            for (int k = 0; k < D; ++k) {
                const double ce = CE[k] / (double)N, s = 0.5 * (ce + dice_s[k]), Em = E[k] / (double)N;
                out[UAPS_U_CE(D, C) + k] = (float)ce;
                out[UAPS_U_DICE(D, C) + k] = (float)dice_s[k];
                out[UAPS_U_S(D, C) + k] = (float)s;
                out[UAPS_U_E(D, C) + k] = (float)Em;
                ps += s * Em;                                       // UAPS_train.py:265-268
                lu += V[k];
            }
            ps /= D;                                                // :277
            lu /= ((double)N * D);                                  // :241-243
            out[UAPS_U_PS(D, C)] = (float)ps;
            out[UAPS_U_LUN(D, C)] = (float)lu;
            out[UAPS_U_LOSS(D, C)] = (float)((double)cw1 * ps + (double)cw2 * lu);
            out[UAPS_U_LOSS(D, C) + 1] = 0.f;
        } else {
            double sup = 0.0;
            for (int k = 0; k < D; ++k) {
                const double ce = CE[k] / (double)N;
                out[UAPS_S_CE(D, C) + k] = (float)ce;
                out[UAPS_S_DICE(D, C) + k] = (float)dice_s[k];
                sup += (double)cw1 * ce + (double)cw2 * dice_s[k];  // UAPS_train.py:208-211 with cw1 = cw2 = 0.5/D
            }
            out[UAPS_S_SUP(D, C)] = (float)sup;                     // :218
            out[UAPS_S_BAD(D, C)] = (float)cnt[C];
        }
    }
}

__host__ __device__ constexpr int unsup_nsums(int D, int C) { return D + 2 * D * C + C + 2 * D; }
__host__ __device__ constexpr int sup_nsums(int D, int C) { return D + 2 * D * C + C + 1; }

// one block: reduce the block rows, then the scalars
template <bool UNSUP>
__global__ __launch_bounds__(kFinalizeThreads) void finalize_kernel(const float* __restrict__ partials, int nrows, int D, int C,
                                                            long N, float cw1, float cw2, float eps,
                                                            float* __restrict__ out) {
    __shared__ double tot[kMaxSums];
    __shared__ double dice_s[UAPS_MAX_HEADS];
    reduce_partials(partials, nrows, UNSUP ? unsup_nsums(D, C) : sup_nsums(D, C), tot);
    __syncthreads();
    finalize_from_tot<UNSUP>(tot, dice_s, D, C, N, cw1, cw2, eps, out);
}

// Both halves of a step's loss block (supervised partials, then unsupervised partials) by ONE block.  With `sums_out` the
// raw double sums [sup_nsums | unsup_nsums] are written instead and nothing is finalised: the caller exchanges them between
// ranks (sum) and runs pair_finalize_sums_kernel with the global pixel count -- the reference computes every mean and Dice
// sum over the gathered batch of all GPUs (UAPS_model.py:13 nn.DataParallel gather, UAPS_train.py:194-277).
static __global__ __launch_bounds__(kFinalizeThreads) void pair_finalize_kernel(const float* __restrict__ part_s, int nrows_s,
                                                                         const float* __restrict__ part_u, int nrows_u, int D, int C,
                                                                         long N, float ce_coef, float dice_coef, float cw1, float cw2,
                                                                         float eps, float* __restrict__ sscal, float* __restrict__ uscal,
                                                                         double* __restrict__ sums_out, const uint32_t* __restrict__ st) {
    cw1 = step_f(st, kStepCw1, cw1); cw2 = step_f(st, kStepCw2, cw2);
    __shared__ double tot_s[kMaxSums], tot_u[kMaxSums];
    __shared__ double dice_s[UAPS_MAX_HEADS];
    const int ns = sup_nsums(D, C), nu = unsup_nsums(D, C);
    reduce_partials(part_s, nrows_s, ns, tot_s);
    reduce_partials(part_u, nrows_u, nu, tot_u);
    __syncthreads();
    if (sums_out != nullptr) {
        for (int i = threadIdx.x; i < ns + nu; i += kFinalizeThreads) sums_out[i] = i < ns ? tot_s[i] : tot_u[i - ns];
        return;
    }
    finalize_from_tot<false>(tot_s, dice_s, D, C, N, ce_coef, dice_coef, eps, sscal);
    __syncthreads();
    finalize_from_tot<true>(tot_u, dice_s, D, C, N, cw1, cw2, eps, uscal);
}
static __global__ __launch_bounds__(kFinalizeThreads) void pair_finalize_sums_kernel(const double* __restrict__ sums, int D, int C, long N,
                                                                              float ce_coef, float dice_coef, float cw1, float cw2,
                                                                              float eps, float* __restrict__ sscal, float* __restrict__ uscal,
                                                                              const uint32_t* __restrict__ st) {
    cw1 = step_f(st, kStepCw1, cw1); cw2 = step_f(st, kStepCw2, cw2);
    __shared__ double tot_s[kMaxSums], tot_u[kMaxSums];
    __shared__ double dice_s[UAPS_MAX_HEADS];
    const int ns = sup_nsums(D, C), nu = unsup_nsums(D, C);
    for (int i = threadIdx.x; i < ns + nu; i += kFinalizeThreads) {
        if (i < ns) tot_s[i] = sums[i]; else tot_u[i - ns] = sums[i];
    }
    __syncthreads();
    finalize_from_tot<false>(tot_s, dice_s, D, C, N, ce_coef, dice_coef, eps, sscal);
    __syncthreads();
    finalize_from_tot<true>(tot_u, dice_s, D, C, N, cw1, cw2, eps, uscal);
}

// --------------------------------------------------------------------------------------------------
// F2: unsupervised backward.  Bytes per pixel: 4DC + 8 read, 4DC written.
// --------------------------------------------------------------------------------------------------
// N: the pixel count the loss was averaged over (local batch, or the gathered global batch)
template <int D, int C, int VEC, bool TRACK = false>
__device__ __forceinline__ void unsup_bwd_body(const HeadPtrs<D>& z, const HeadOutPtrs<D>& dz, int HW, long ngroups, long N,
                                               const int64_t* __restrict__ pseudo,
                                               const float* __restrict__ sc, float cw1, float cw2,
                                               const float* __restrict__ gscale, int bid, int nblk, float& am) {
    const float gs = gscale ? gscale[0] : 1.f;
    const float invN = 1.f / (float)N;
    // per-head constants: coefficient of exp(-v_k) in g_k, and of the (CE + Dice) term
    float ge[D], cs[D];
    const float gu = gs * cw2 * invN / (float)D;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        ge[k] = gs * (cw1 / (float)D) * sc[UAPS_U_S(D, C) + k] * invN;
        cs[k] = gs * (cw1 / (float)D) * sc[UAPS_U_E(D, C) + k] * 0.5f;
    }
    const float* __restrict__ A1 = sc + UAPS_U_A1(D, C);
    const float* __restrict__ A2 = sc + UAPS_U_A2(D, C);

    for (long g = (long)bid * kThreads + threadIdx.x; g < ngroups; g += (long)nblk * kThreads) {
        const long n0 = g * VEC;
        const long b = n0 / HW;
        const long hw = n0 - b * HW;
        const long base = b * C * (long)HW + hw;
        float zv[D][C][VEC];
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) load_vec<VEC>(z.p[k] + base + (long)c * HW, zv[k][c]);
        int yv[VEC];
        load_labels<VEC>(pseudo + n0, yv);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float p[D][C], lp[D][C], m[C], lm1[C], gk[D], h[C];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                float zz[C];
#pragma unroll
                for (int c = 0; c < C; ++c) zz[c] = zv[k][c][v];
                softmax_regs<C>(zz, p[k], lp[k]);
            }
            float xm = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float s = p[0][c];
#pragma unroll
                for (int k = 1; k < D; ++k) s = add_rn(s, p[k][c]);
                m[c] = s / (float)D;
                const float lm = (m[c] > 0.f) ? flog(m[c]) : 0.f;
                xm += m[c] * lm;
                lm1[c] = (m[c] > 0.f) ? lm + 1.f : 0.f;       // d xlogy(m,m)/dm, 0 at m == 0 (limit; reference NaNs)
                h[c] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                float dot = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) dot += m[c] * lp[k][c];
                gk[k] = gu - ge[k] * fexp(dot - xm);            // dL/dv_k ; exp(-v_k) = exp(dot - xm)
#pragma unroll
                for (int c = 0; c < C; ++c) h[c] += (m[c] > 0.f) ? gk[k] * (lm1[c] - lp[k][c]) : 0.f;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) h[c] *= (1.f / (float)D);
            const int y = yv[v];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float a[C], pa = 0.f, ph = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    a[c] = ((y == c) ? A1[j * C + c] : 0.f) + A2[j * C + c];
                    pa += p[j][c] * a[c];
                    ph += p[j][c] * h[c];
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float oh = (y == c) ? 1.f : 0.f;
                    const float gr = cs[j] * ((p[j][c] - oh) * invN + p[j][c] * (a[c] - pa))
                                   - gk[j] * (m[c] - p[j][c])
                                   + p[j][c] * (h[c] - ph);
                    zv[j][c][v] = gr;
                    if constexpr (TRACK) am = fmaxf(am, fabsf(gr));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) store_vec<VEC>(dz.p[k] + base + (long)c * HW, zv[k][c]);
    }
}
template <int D, int C, int VEC>
__global__ __launch_bounds__(kThreads) void unsup_bwd_kernel(HeadPtrs<D> z, HeadOutPtrs<D> dz, int HW, long ngroups, long N,
                                                             const int64_t* __restrict__ pseudo,
                                                             const float* __restrict__ sc, float cw1, float cw2,
                                                             const float* __restrict__ gscale) {
    float am = 0.f;
    unsup_bwd_body<D, C, VEC>(z, dz, HW, ngroups, N, pseudo, sc, cw1, cw2, gscale, (int)blockIdx.x, (int)gridDim.x, am);
}

// --------------------------------------------------------------------------------------------------
// F3 backward.  Bytes per pixel: 4DC + 8 read, 4DC written.
// --------------------------------------------------------------------------------------------------
template <int D, int C, int VEC, bool TRACK = false>
__device__ __forceinline__ void sup_bwd_body(const HeadPtrs<D>& z, const HeadOutPtrs<D>& dz, int HW, long ngroups, long N,
                                             const int64_t* __restrict__ labels,
                                             const float* __restrict__ sc, float ce_coef,
                                             float dice_coef, const float* __restrict__ gscale, int bid, int nblk, float& am) {
    const float gs0 = gscale ? gscale[0] : 1.f;
    const float gce = gs0 * ce_coef / (float)N, gdc = gs0 * dice_coef;
    const float* __restrict__ A1 = sc + UAPS_S_A1(D, C);
    const float* __restrict__ A2 = sc + UAPS_S_A2(D, C);
    for (long g = (long)bid * kThreads + threadIdx.x; g < ngroups; g += (long)nblk * kThreads) {
        const long n0 = g * VEC;
        const long b = n0 / HW;
        const long hw = n0 - b * HW;
        const long base = b * C * (long)HW + hw;
        int yv[VEC];
        load_labels<VEC>(labels + n0, yv);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float zv[C][VEC];
#pragma unroll
            for (int c = 0; c < C; ++c) load_vec<VEC>(z.p[k] + base + (long)c * HW, zv[c]);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float zz[C], p[C], lp[C], a[C], pa = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) zz[c] = zv[c][v];
                softmax_regs<C>(zz, p, lp);
                const int y = yv[v];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    a[c] = ((y == c) ? A1[k * C + c] : 0.f) + A2[k * C + c];
                    pa += p[c] * a[c];
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float oh = (y == c) ? 1.f : 0.f;
                    zv[c][v] = gce * (p[c] - oh) + gdc * p[c] * (a[c] - pa);
                    if constexpr (TRACK) am = fmaxf(am, fabsf(zv[c][v]));
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) store_vec<VEC>(dz.p[k] + base + (long)c * HW, zv[c]);
        }
    }
}
template <int D, int C, int VEC>
__global__ __launch_bounds__(kThreads) void sup_bwd_kernel(HeadPtrs<D> z, HeadOutPtrs<D> dz, int HW, long ngroups, long N,
                                                           const int64_t* __restrict__ labels,
                                                           const float* __restrict__ sc, float ce_coef,
                                                           float dice_coef, const float* __restrict__ gscale) {
    float am = 0.f;
    sup_bwd_body<D, C, VEC>(z, dz, HW, ngroups, N, labels, sc, ce_coef, dice_coef, gscale, (int)blockIdx.x, (int)gridDim.x, am);
}

// --------------------------------------------------------------------------------------------------
// The loss block of a training step as ONE forward and ONE backward launch: blocks [0, nb_s) run the supervised branch
// on the labelled logits, blocks [nb_s, gridDim.x) the unsupervised branch on the unlabelled logits (UAPS_train.py:
// 194-218 and 186-189, 223-277).  VS / VU: pixels per thread of the two branches.
// --------------------------------------------------------------------------------------------------
// PFS / PFU: the prefetching form of the supervised / unsupervised body; MINW: waves per SIMD the register allocation must allow
template <int D, int C, int VS, int VU, bool PFS, bool PFU, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void pair_fwd_kernel(HeadPtrs<D> zl, HeadPtrs<D> zu, HeadWeights<D> w, int HW, long N,
                                                                  const int64_t* __restrict__ labels, int64_t* __restrict__ pseudo,
                                                                  float* __restrict__ var, float* __restrict__ part_s,
                                                                  float* __restrict__ part_u, int nb_s, const uint32_t* __restrict__ st) {
    if (st != nullptr) {                 // captured step: the mixing weights of THIS step come from the step state (philox.hpp)
#pragma unroll
        for (int k = 0; k < D; ++k) w.w[k] = __uint_as_float(st[kStepW + k]);
    }
#ifdef UAPS_LOSS_STAGGER
    // two waves per SIMD running the same load -> compute loop from the same start stay in lockstep (both wait for memory,
    // then both compete for the VALU): delay the wave in the odd hardware slot by about half a group's period
    if (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 1) __builtin_amdgcn_s_sleep(UAPS_LOSS_STAGGER);
#endif
    if ((int)blockIdx.x < nb_s) {
        if constexpr (PFS) sup_fwd_body_pf<D, C, VS>(zl, HW, N / VS, labels, part_s, (int)blockIdx.x, nb_s);
        else sup_fwd_body<D, C, VS>(zl, HW, N / VS, labels, part_s, (int)blockIdx.x, nb_s);
    } else {
        if constexpr (PFU) unsup_fwd_body_pf<D, C, VU>(zu, w, HW, N / VU, N, pseudo, var, part_u, (int)blockIdx.x - nb_s, (int)gridDim.x - nb_s);
        else unsup_fwd_body<D, C, VU>(zu, w, HW, N / VU, N, pseudo, var, part_u, (int)blockIdx.x - nb_s, (int)gridDim.x - nb_s);
    }
}
// Nloss: the pixel count the scalars were finalised with (= N, or the global count after an exchange of the sums)
// TRACK: also raise *amax_out to max|gradient element| (uaps_call_hints::out_amax; costs ~17 us of VALU work at 16 + 16 images)
template <int D, int C, int VS, int VU, bool TRACK = false>
__global__ __launch_bounds__(kThreads) void pair_bwd_kernel(HeadPtrs<D> zl, HeadPtrs<D> zu, HeadOutPtrs<D> dl, HeadOutPtrs<D> du, int HW,
                                                            long N, long Nloss, const int64_t* __restrict__ labels,
                                                            const int64_t* __restrict__ pseudo, const float* __restrict__ sscal,
                                                            const float* __restrict__ uscal, float ce_coef, float dice_coef, float cw1,
                                                            float cw2, const float* __restrict__ gscale, int nb_s,
                                                            const uint32_t* __restrict__ st, float* __restrict__ amax_out) {
    cw1 = step_f(st, kStepCw1, cw1); cw2 = step_f(st, kStepCw2, cw2);
    float am = 0.f;                              // max|gradient element| written by this thread (uaps_call_hints::out_amax)
    if ((int)blockIdx.x < nb_s) sup_bwd_body<D, C, VS, TRACK>(zl, dl, HW, N / VS, Nloss, labels, sscal, ce_coef, dice_coef, gscale, (int)blockIdx.x, nb_s, am);
    else unsup_bwd_body<D, C, VU, TRACK>(zu, du, HW, N / VU, Nloss, pseudo, uscal, cw1, cw2, gscale, (int)blockIdx.x - nb_s, (int)gridDim.x - nb_s, am);
    if (TRACK && amax_out) {                              // uniform branch
        __shared__ float sm[16];
        block_amax_to(amax_out, am, sm);
    }
}

}  // namespace uaps
