// Fused loss-block kernels of the UAPS step for gfx950 (wave64, HBM-bound streaming kernels).
//
// One thread owns VEC horizontally adjacent pixels and keeps all D*C logits of each in registers:
// with the reference's NCHW fp32 logits a (head, class) plane is contiguous over pixels, so a lane
// reads 16 B (VEC=4) from each of the D*C planes and a wave reads 1 KiB per plane, fully coalesced.
// The class softmax is therefore register-local (no cross-lane traffic); cross-lane work is only
// the block reduction of the per-thread partial sums (through an LDS tile, block_reduce_store),
// written as one row of partials per block and reduced in a fixed order by a one-block
// finalize kernel (double accumulation) -- no float atomics, so results are bitwise reproducible
// run to run.  The forward bodies compute on pairs of adjacent pixels (v_pk_*_f32; "Lanes" below).
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
// one quarter-rate instruction (+ one multiply) each instead of libm's ~15-instruction sequences (these kernels are
// VALU/HBM co-limited: ~30 transcendentals per pixel at D=C=4).  __logf is NOT such a form on this toolchain: it expands to
// the denormal pre-scale (compare, select, ldexp) and a two-term log2 -> ln correction, 12 instructions per call, so the
// logarithm here is the instruction itself.  v_log_f32 reads denormal arguments as zero: callers pass softmax sums (>= 1)
// or probabilities tested with is_normal_pos().
__device__ __forceinline__ float fexp(float x) { return __expf(x); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ bool is_normal_pos(float x) { return x >= 1.17549435e-38f; }   // else xlogy(x, x) counts as 0 (|x log x| < 1e-36)
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

// --------------------------------------------------------------------------------------------------
// "Lanes": the forward bodies are written once over a value type T that is either one pixel (float) or two horizontally
// adjacent pixels (f2).  gfx950 issues v_pk_add/mul/fma_f32 on an aligned VGPR pair at the rate of the scalar forms, and the
// 16-byte logit loads put pixels (0,1) and (2,3) of a thread in aligned pairs already, so every multiply/add of the block
// costs half an instruction per pixel.  (Left to itself the SLP vectoriser also forms v_pk_* instructions, but across
// unrelated values: a quarter of the kernel was v_mov_b32 gathering operands into pairs.)  Transcendentals, max and
// compare/select have no packed form and are issued per pixel.
// --------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
template <class T> struct Lanes;
template <> struct Lanes<float> { static constexpr int N = 1; };
template <> struct Lanes<f2> { static constexpr int N = 2; };
template <class T> __device__ __forceinline__ T splat(float x);
template <> __device__ __forceinline__ float splat<float>(float x) { return x; }
template <> __device__ __forceinline__ f2 splat<f2>(float x) { return f2{x, x}; }
__device__ __forceinline__ float lane(float v, int) { return v; }
__device__ __forceinline__ float lane(f2 v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ float lane_sum(float v) { return v; }
__device__ __forceinline__ float lane_sum(f2 v) { return v.x + v.y; }
__device__ __forceinline__ float exp2_l(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ f2 exp2_l(f2 x) { return f2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ float log2_l(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ f2 log2_l(f2 x) { return f2{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)}; }
__device__ __forceinline__ float rcp_l(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ f2 rcp_l(f2 x) { return f2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ float max_l(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f2 max_l(f2 a, f2 b) { return f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ float fma_l(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f2 fma_l(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 mul_rn(f2 a, f2 b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ f2 add_rn(f2 a, f2 b) {
#pragma clang fp contract(off)
    return a + b;
}
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f, kFltMin = 1.17549435e-38f;
// one-hot of an integer label per lane: one compare + select per class and pixel
template <int C> __device__ __forceinline__ void one_hot(const int (&y)[1], float (&oh)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) oh[c] = (y[0] == c) ? 1.f : 0.f;
}
template <int C> __device__ __forceinline__ void one_hot(const int (&y)[2], f2 (&oh)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) oh[c] = f2{(y[0] == c) ? 1.f : 0.f, (y[1] == c) ? 1.f : 0.f};
}
// softmax over C values per lane: p and the log-sum-exp (log p_c = z_c - lse); same arithmetic as softmax_regs
template <int C, class T> __device__ __forceinline__ void softmax_lanes(const T (&z)[C], T (&p)[C], T& lse) {
    T mx = z[0];
#pragma unroll
    for (int c = 1; c < C; ++c) mx = max_l(mx, z[c]);
    T s = splat<T>(0.f);
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = exp2_l((z[c] - mx) * splat<T>(kLog2e)); s += p[c]; }
    const T inv = rcp_l(s);
    lse = mx + log2_l(s) * splat<T>(kLn2);
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
}
// VEC register values -> lanes of type T starting at pixel v
template <class T, int VEC> __device__ __forceinline__ T take(const float (&a)[VEC], int v) {
    if constexpr (Lanes<T>::N == 1) return a[v]; else return f2{a[v], a[v + 1]};
}
template <int VEC> using lanes_t = std::conditional_t<VEC % 2 == 0, f2, float>;
// Running sums are kept per lane (A = T: one packed add per pair of pixels, twice the registers) or per thread (A = float).
// The 48 sums of the unsupervised forward at D = C = 4 are 96 registers per lane-pair, on top of the 64 logits of a 4-pixel
// group: the per-lane form is for 2-pixel groups only.
template <int VEC> using acc_t = float;
__device__ __forceinline__ void accum(float& a, float x) { a += x; }
__device__ __forceinline__ void accum(f2& a, f2 x) { a += x; }
__device__ __forceinline__ void accum(float& a, f2 x) { a += x.x; a += x.y; }
__device__ __forceinline__ void accum_fma(float& a, float x, float y) { a = __builtin_fmaf(x, y, a); }
__device__ __forceinline__ void accum_fma(f2& a, f2 x, f2 y) { a = fma_l(x, y, a); }
__device__ __forceinline__ void accum_fma(float& a, f2 x, f2 y) { a = __builtin_fmaf(x.x, y.x, a); a = __builtin_fmaf(x.y, y.y, a); }
template <class A> __device__ __forceinline__ A zero_acc();
template <> __device__ __forceinline__ float zero_acc<float>() { return 0.f; }
template <> __device__ __forceinline__ f2 zero_acc<f2>() { return f2{0.f, 0.f}; }

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

// Block reduction of NS per-thread partial sums into column `bid` of `partials` ([NS][nblk], sum-major so that the
// finalize kernel reads each sum's block partials as one coalesced run).  Through LDS, 32 sums at a time: every thread
// parks its 32 values in a [32][kThreads + 1] tile (conflict-free: consecutive lanes, consecutive banks), then thread t adds
// the 32 columns [32 * (t / 32), +32) of row t % 32 (the row pitch of kThreads + 1 keeps the 64 lanes of a wave on 64
// different banks) and the eight column-block sums of a row are added in a fixed order.  ~110 instructions per 32 sums and
// thread.  The earlier form reduced every sum across the wave with DPP adds and v_readlane (23 instructions per sum, 1100
// for the 48 sums of the unsupervised forward -- as much as one group of pixels costs, and a persistent thread only
// processes three to five groups).
struct ReduceScratch { float tile[32][kThreads + 1]; float part[kThreads / 32][32]; };
__device__ __forceinline__ ReduceScratch& reduce_scratch() {     // one allocation per kernel, whatever the number of NS forms in it
    __shared__ ReduceScratch s;
    return s;
}
template <int NS> __device__ __forceinline__ void block_reduce_store(float (&acc)[NS], float* partials, int bid, int nblk) {
    constexpr int R = 32, SEG = kThreads / R;
    ReduceScratch& scratch = reduce_scratch();
    float (&tile)[R][kThreads + 1] = scratch.tile;
    float (&part)[SEG][R] = scratch.part;
    const int t = threadIdx.x, row = t % R, seg = t / R;
#pragma unroll
    for (int c0 = 0; c0 < NS; c0 += R) {
        const int rows = NS - c0 < R ? NS - c0 : R;
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (c0 + i < NS) tile[i][t] = acc[c0 + i];
        __syncthreads();
        if (row < rows) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int j = 0; j < kThreads / SEG; j += 4) {
                s0 += tile[row][seg * (kThreads / SEG) + j];
                s1 += tile[row][seg * (kThreads / SEG) + j + 1];
                s2 += tile[row][seg * (kThreads / SEG) + j + 2];
                s3 += tile[row][seg * (kThreads / SEG) + j + 3];
            }
            part[seg][row] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        if (t < rows) {
            float s = part[0][t];
#pragma unroll
            for (int g = 1; g < SEG; ++g) s += part[g][t];
            partials[(size_t)(c0 + t) * nblk + bid] = s;
        }
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
// the arithmetic of one group of VEC pixels: zv -> pseudo-labels yv, variance maps varv, running sums acc (per lane:
// the caller folds the lanes with fold_lanes before the block reduction)
template <int D, int C, int VEC, class A>
__device__ __forceinline__ void unsup_fwd_group(const float (&zv)[D][C][VEC], const HeadWeights<D>& w, A (&acc)[UnsupLayout<D, C>::NS],
                                                int (&yv)[VEC], float (&varv)[D][VEC]) {
    using L = UnsupLayout<D, C>;
    using T = lanes_t<VEC>;
    constexpr int LN = Lanes<T>::N;
#pragma unroll
    for (int v = 0; v < VEC; v += LN) {
        T z[D][C], p[D][C], lse[D], m[C];
#pragma unroll
        for (int k = 0; k < D; ++k) {
#pragma unroll
            for (int c = 0; c < C; ++c) z[k][c] = take<T>(zv[k][c], v);
            softmax_lanes<C>(z[k], p[k], lse[k]);
        }
        // mean prediction (UAPS_train.py:223) and the mixture that feeds arg-max (:252-255):
        // separate multiply and add roundings, left to right, like the reference's tensor ops.
        T xm = splat<T>(0.f);
        int y[LN];
        float best[LN];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            T sm = p[0][c];
            T mix = mul_rn(splat<T>(w.w[0]), p[0][c]);
#pragma unroll
            for (int k = 1; k < D; ++k) { sm = add_rn(sm, p[k][c]); mix = add_rn(mix, mul_rn(splat<T>(w.w[k]), p[k][c])); }
            m[c] = sm / splat<T>((float)D);
            // xlogy(m, m); v_log_f32 reads a denormal as zero, so the argument is held at FLT_MIN: m * log(FLT_MIN) is 0 for m = 0
            // and below 1e-36 in magnitude for denormal m
            xm = fma_l(m[c], log2_l(max_l(m[c], splat<T>(kFltMin))) * splat<T>(kLn2), xm);
#pragma unroll
            for (int i = 0; i < LN; ++i) {                          // first maximum wins, as torch.argmax
                const float mi = lane(mix, i);
                if (c == 0 || mi > best[i]) { best[i] = mi; y[i] = c; }
            }
        }
        T oh[C];                                     // one-hot of the pseudo-label: one compare + select per class, then plain
        one_hot<C>(y, oh);                           // multiply-adds (p * 1 and p * 0 are exact, so the sums are unchanged)
#pragma unroll
        for (int i = 0; i < LN; ++i) yv[v + i] = y[i];
#pragma unroll
        for (int c = 0; c < C; ++c) accum(acc[L::CNT + c], oh[c]);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            T dot = splat<T>(0.f), lpy = splat<T>(0.f);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const T lp = z[k][c] - lse[k];
                dot = fma_l(m[c], lp, dot);
                accum(acc[L::P + k * C + c], p[k][c]);
                accum_fma(acc[L::I + k * C + c], oh[c], p[k][c]);
                lpy = fma_l(oh[c], lp, lpy);
            }
            const T vk = xm - dot;                                   // sum_c KL(m || p_k)  (:226)
#pragma unroll
            for (int i = 0; i < LN; ++i) varv[k][v + i] = lane(vk, i);
            accum(acc[L::V + k], vk);
            accum(acc[L::E + k], exp2_l(vk * splat<T>(-kLog2e)));    // :227
            accum(acc[L::CE + k], -lpy);
        }
    }
}
template <int NS, class A> __device__ __forceinline__ void fold_lanes(const A (&acc)[NS], float (&out)[NS]) {
#pragma unroll
    for (int i = 0; i < NS; ++i) out[i] = lane_sum(acc[i]);
}

template <int D, int C, int VEC>
__device__ __forceinline__ void unsup_fwd_body(const HeadPtrs<D>& z, const HeadWeights<D>& w, int HW, long ngroups,
                                               long N, int64_t* __restrict__ pseudo,
                                               float* __restrict__ var, float* __restrict__ partials, int bid, int nblk) {
    using L = UnsupLayout<D, C>;
    using A = acc_t<VEC>;
    A acc[L::NS];
#pragma unroll
    for (int i = 0; i < L::NS; ++i) acc[i] = zero_acc<A>();

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
    float accf[L::NS];
    fold_lanes(acc, accf);
    block_reduce_store<L::NS>(accf, partials, bid, nblk);
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
    using A = acc_t<VEC>;
    A acc[L::NS];
#pragma unroll
    for (int i = 0; i < L::NS; ++i) acc[i] = zero_acc<A>();
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
    float accf[L::NS];
    fold_lanes(acc, accf);
    block_reduce_store<L::NS>(accf, partials, bid, nblk);
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
// one head of a group of VEC pixels: logits zv, labels yv with their one-hot oh (lanes_t<VEC> per class)
template <int D, int C, int VEC, class A, class T>
__device__ __forceinline__ void sup_fwd_head(const float (&zv)[C][VEC], const int (&yv)[VEC], const T (&oh)[VEC / Lanes<T>::N][C], int k,
                                             A (&acc)[SupLayout<D, C>::NS]) {
    using L = SupLayout<D, C>;
    constexpr int LN = Lanes<T>::N;
#pragma unroll
    for (int v = 0; v < VEC; v += LN) {
        T z[C], p[C], lse;
#pragma unroll
        for (int c = 0; c < C; ++c) z[c] = take<T>(zv[c], v);
        softmax_lanes<C>(z, p, lse);
        // log p of the label: a select chain over the logits (an out-of-range label selects nothing and contributes 0, as the
        // class compare of the scalar form did; the launch reports such labels through BAD)
        float zy[LN];
#pragma unroll
        for (int i = 0; i < LN; ++i) {
            zy[i] = lane(lse, i);
#pragma unroll
            for (int c = 0; c < C; ++c) zy[i] = (yv[v + i] == c) ? lane(z[c], i) : zy[i];
        }
        T zyl;
        if constexpr (LN == 1) zyl = zy[0]; else zyl = f2{zy[0], zy[1]};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            accum(acc[L::P + k * C + c], p[c]);
            accum_fma(acc[L::I + k * C + c], oh[v / LN][c], p[c]);
        }
        accum(acc[L::CE + k], lse - zyl);
    }
}
// labels of a group -> one-hot lanes, class counts and the out-of-range count
template <int D, int C, int VEC, class A, class T>
__device__ __forceinline__ void sup_fwd_labels(const int (&yv)[VEC], T (&oh)[VEC / Lanes<T>::N][C], A (&acc)[SupLayout<D, C>::NS]) {
    using L = SupLayout<D, C>;
    constexpr int LN = Lanes<T>::N;
#pragma unroll
    for (int v = 0; v < VEC; v += LN) {
        int y[LN];
#pragma unroll
        for (int i = 0; i < LN; ++i) y[i] = yv[v + i];
        one_hot<C>(y, oh[v / LN]);
#pragma unroll
        for (int c = 0; c < C; ++c) accum(acc[L::CNT + c], oh[v / LN][c]);
        float bad[LN];
#pragma unroll
        for (int i = 0; i < LN; ++i) bad[i] = (y[i] < 0 || y[i] >= C) ? 1.f : 0.f;
        if constexpr (LN == 1) accum(acc[L::BAD], bad[0]); else accum(acc[L::BAD], f2{bad[0], bad[1]});
    }
}
template <int D, int C, int VEC>
__device__ __forceinline__ void sup_fwd_body(const HeadPtrs<D>& z, int HW, long ngroups,
                                             const int64_t* __restrict__ labels,
                                             float* __restrict__ partials, int bid, int nblk) {
    using L = SupLayout<D, C>;
    using T = lanes_t<VEC>;
    using A = acc_t<VEC>;
    A acc[L::NS];
#pragma unroll
    for (int i = 0; i < L::NS; ++i) acc[i] = zero_acc<A>();
    for (long g = (long)bid * kThreads + threadIdx.x; g < ngroups; g += (long)nblk * kThreads) {
        const long n0 = g * VEC;
        const long b = n0 / HW;
        const long hw = n0 - b * HW;
        const long base = b * C * (long)HW + hw;
        int yv[VEC];
        load_labels<VEC>(labels + n0, yv);
        T oh[VEC / Lanes<T>::N][C];
        sup_fwd_labels<D, C, VEC>(yv, oh, acc);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float zv[C][VEC];
#pragma unroll
            for (int c = 0; c < C; ++c) load_vec<VEC>(z.p[k] + base + (long)c * HW, zv[c]);
            sup_fwd_head<D, C, VEC>(zv, yv, oh, k, acc);
        }
    }
    float accf[L::NS];
    fold_lanes(acc, accf);
    block_reduce_store<L::NS>(accf, partials, bid, nblk);
}
// The supervised forward with the next head's logits (and, behind the last head, the next group's labels and first head)
// fetched while the current head is computed: the pair kernel runs this branch at the register budget of the
// unsupervised one (two waves per SIMD), where the plain loop above would wait out every load.
template <int D, int C, int VEC>
__device__ __forceinline__ void sup_fwd_body_pf(const HeadPtrs<D>& z, int HW, long ngroups,
                                                const int64_t* __restrict__ labels,
                                                float* __restrict__ partials, int bid, int nblk) {
    using L = SupLayout<D, C>;
    using T = lanes_t<VEC>;
    using A = acc_t<VEC>;
    A acc[L::NS];
#pragma unroll
    for (int i = 0; i < L::NS; ++i) acc[i] = zero_acc<A>();
    const long stride = (long)nblk * kThreads;
    auto load_head = [&](long g, int k, float (&zv)[C][VEC]) {
        const long n0 = g * VEC, b = n0 / HW, hw = n0 - b * HW, base = b * C * (long)HW + hw;
#pragma unroll
        for (int c = 0; c < C; ++c) load_vec<VEC>(z.p[k] + base + (long)c * HW, zv[c]);
    };
    // buffers alternate head by head; PAR = which buffer holds head 0 of this group (flips per group when D is odd)
    float za[C][VEC], zb[C][VEC];
    int yv[VEC], yn[VEC];
    auto group = [&](long g, auto par) {
        constexpr int PAR = decltype(par)::value;
        T oh[VEC / Lanes<T>::N][C];
        sup_fwd_labels<D, C, VEC>(yv, oh, acc);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const bool cur_a = ((k + PAR) & 1) == 0;
            if (k + 1 < D) {
                if (cur_a) load_head(g, k + 1, zb); else load_head(g, k + 1, za);
            } else if (g + stride < ngroups) {
                load_labels<VEC>(labels + (g + stride) * VEC, yn);
                if (cur_a) load_head(g + stride, 0, zb); else load_head(g + stride, 0, za);
            }
            if (cur_a) sup_fwd_head<D, C, VEC>(za, yv, oh, k, acc); else sup_fwd_head<D, C, VEC>(zb, yv, oh, k, acc);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) yv[v] = yn[v];
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
    float accf[L::NS];
    fold_lanes(acc, accf);
    block_reduce_store<L::NS>(accf, partials, bid, nblk);
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

// fixed-order double reduction of two sets of block partials ([nsa][nra] -> ta, [nsb][nrb] -> tb; nsb may be 0).  8 lanes
// per sum, all sums of both sets at once (1024 threads = 128 groups >= the 89 + 80 sums of D = C = 8 in two trips): lane j
// adds rows j, j + 8, ... in four double chains, 16 independent loads per trip (clamped index, the value dropped beyond the
// last row), then the 8 lanes are combined in a fixed order.  The kernel is one block and all latency: at 192 + 320 rows
// a lane makes 2 + 3 trips through the memory pipeline; the first form (one set after the other, 4 loads per trip) made 16.
__device__ __forceinline__ void reduce_partials(const float* __restrict__ pa, int nra, int nsa, double* ta,
                                                const float* __restrict__ pb, int nrb, int nsb, double* tb) {
    const int grp = threadIdx.x >> 3, j = threadIdx.x & 7;
    for (int i = grp; i < nsa + nsb; i += kFinalizeThreads / 8) {
        const bool first = i < nsa;
        const int nrows = first ? nra : nrb;
        const float* src = first ? pa + (size_t)i * nra : pb + (size_t)(i - nsa) * nrb;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int r = j; r < nrows; r += 128) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int q = r + 8 * u;
                const float x = src[q < nrows ? q : nrows - 1];
                v[u] = q < nrows ? x : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; u += 4) { s0 += (double)v[u]; s1 += (double)v[u + 1]; s2 += (double)v[u + 2]; s3 += (double)v[u + 3]; }
        }
        double sum = (s0 + s1) + (s2 + s3);
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 4, 64);
        if (j == 0) { if (first) ta[i] = sum; else tb[i - nsa] = sum; }
    }
}

// tot[NS] (shared, already complete: call after a barrier) -> the scalar losses and the Dice gradient coefficients the
// backward kernels need.  N = number of pixels the sums run over (the local batch, or the gathered global batch when the
// sums were exchanged between ranks).  Two phases with a block barrier between them; `tid` is the thread's index within the
// group of threads that finalises this branch (negative: not a member), so that both branches run side by side.
template <bool UNSUP>
__device__ __forceinline__ void finalize_elems(int tid, const double* tot, double* dice_s, int D, int C, float eps, float* __restrict__ out) {
    const double *I = tot + D, *P = I + D * C, *cnt = P + D * C;
    const int oA1 = UNSUP ? UAPS_U_A1(D, C) : UAPS_S_A1(D, C);
    const int oA2 = UNSUP ? UAPS_U_A2(D, C) : UAPS_S_A2(D, C);
    const int oI = UNSUP ? UAPS_U_I(D, C) : UAPS_S_I(D, C);
    const int oCard = UNSUP ? UAPS_U_CARD(D, C) : UAPS_S_CARD(D, C);
    const int oCnt = UNSUP ? UAPS_U_CNT(D, C) : UAPS_S_CNT(D, C);
    if (tid >= 0 && tid < D * C) {
        const int t = tid, c = t % C;
        const double card = P[t] + cnt[c];
        const double den = card + (double)eps;
        out[oA1 + t] = (float)(-(2.0 / C) / den);
        out[oA2 + t] = (float)((2.0 / C) * I[t] / (den * den));
        out[oI + t] = (float)I[t];
        out[oCard + t] = (float)card;
    }
    if (tid >= 0 && tid < C) out[oCnt + tid] = (float)cnt[tid];
    // Dice per head from a second group of threads (pytorch_losses.py:88-89)
    const int k = tid - 64;
    if (k >= 0 && k < D) {
        double ds = 0.0;
        for (int c = 0; c < C; ++c) ds += 2.0 * I[k * C + c] / (P[k * C + c] + cnt[c] + (double)eps);
        dice_s[k] = 1.0 - ds / C;
    }
}
template <bool UNSUP>
__device__ __forceinline__ void finalize_scalars(int tid, const double* tot, const double* dice_s, int D, int C, long N, float cw1,
                                                 float cw2, float* __restrict__ out) {
    if (tid != 0) return;
    const double *CE = tot, *cnt = tot + D + 2 * D * C;
    const double invN = 1.0 / (double)N;
    if (UNSUP) {
        const double *E = cnt + C, *V = E + D;
        double ps = 0.0, lu = 0.0;
        for (int k = 0; k < D; ++k) {
            const double ce = CE[k] * invN, sk = 0.5 * (ce + dice_s[k]), Em = E[k] * invN;
            out[UAPS_U_CE(D, C) + k] = (float)ce;
            out[UAPS_U_DICE(D, C) + k] = (float)dice_s[k];
            out[UAPS_U_S(D, C) + k] = (float)sk;
            out[UAPS_U_E(D, C) + k] = (float)Em;
            ps += sk * Em;                                      // UAPS_train.py:265-268
            lu += V[k];
        }
        ps /= D;                                                // :277
        lu *= invN / D;                                         // :241-243
        out[UAPS_U_PS(D, C)] = (float)ps;
        out[UAPS_U_LUN(D, C)] = (float)lu;
        out[UAPS_U_LOSS(D, C)] = (float)((double)cw1 * ps + (double)cw2 * lu);
        out[UAPS_U_LOSS(D, C) + 1] = 0.f;
    } else {
        double sup = 0.0;
        for (int k = 0; k < D; ++k) {
            const double ce = CE[k] * invN;
            out[UAPS_S_CE(D, C) + k] = (float)ce;
            out[UAPS_S_DICE(D, C) + k] = (float)dice_s[k];
            sup += (double)cw1 * ce + (double)cw2 * dice_s[k];  // UAPS_train.py:208-211 with cw1 = cw2 = 0.5/D
        }
        out[UAPS_S_SUP(D, C)] = (float)sup;                     // :218
        out[UAPS_S_BAD(D, C)] = (float)cnt[C];
    }
}
// one branch by the whole block
template <bool UNSUP>
__device__ __forceinline__ void finalize_from_tot(const double* tot, double* dice_s, int D, int C, long N, float cw1, float cw2,
                                                  float eps, float* __restrict__ out) {
    finalize_elems<UNSUP>((int)threadIdx.x, tot, dice_s, D, C, eps, out);
    __syncthreads();
    finalize_scalars<UNSUP>((int)threadIdx.x, tot, dice_s, D, C, N, cw1, cw2, out);
}
// both branches side by side: supervised on threads [0, 128), unsupervised on [128, 256)
__device__ __forceinline__ void finalize_pair_from_tot(const double* tot_s, const double* tot_u, double* dice_s, double* dice_u, int D, int C,
                                                       long N, float ce_coef, float dice_coef, float cw1, float cw2, float eps,
                                                       float* __restrict__ sscal, float* __restrict__ uscal) {
    const int ts = threadIdx.x < 128 ? (int)threadIdx.x : -1000, tu = (int)threadIdx.x - 128 < 128 ? (int)threadIdx.x - 128 : -1000;
    finalize_elems<false>(ts, tot_s, dice_s, D, C, eps, sscal);
    finalize_elems<true>(tu, tot_u, dice_u, D, C, eps, uscal);
    __syncthreads();
    finalize_scalars<false>(ts, tot_s, dice_s, D, C, N, ce_coef, dice_coef, sscal);
    finalize_scalars<true>(tu, tot_u, dice_u, D, C, N, cw1, cw2, uscal);
    // the step's loss (UAPS_train.py:282): supervised + consistency terms, one fp32 add -- what the host used to launch an add kernel for
    __syncthreads();
    if (threadIdx.x == 0) uscal[UAPS_U_TOTAL(D, C)] = sscal[UAPS_S_SUP(D, C)] + uscal[UAPS_U_LOSS(D, C)];
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
    reduce_partials(partials, nrows, UNSUP ? unsup_nsums(D, C) : sup_nsums(D, C), tot, nullptr, 0, 0, nullptr);
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
    __shared__ double dice_s[UAPS_MAX_HEADS], dice_u[UAPS_MAX_HEADS];
    const int ns = sup_nsums(D, C), nu = unsup_nsums(D, C);
    reduce_partials(part_s, nrows_s, ns, tot_s, part_u, nrows_u, nu, tot_u);
    __syncthreads();
    if (sums_out != nullptr) {
        for (int i = threadIdx.x; i < ns + nu; i += kFinalizeThreads) sums_out[i] = i < ns ? tot_s[i] : tot_u[i - ns];
        return;
    }
    finalize_pair_from_tot(tot_s, tot_u, dice_s, dice_u, D, C, N, ce_coef, dice_coef, cw1, cw2, eps, sscal, uscal);
}
static __global__ __launch_bounds__(kFinalizeThreads) void pair_finalize_sums_kernel(const double* __restrict__ sums, int D, int C, long N,
                                                                              float ce_coef, float dice_coef, float cw1, float cw2,
                                                                              float eps, float* __restrict__ sscal, float* __restrict__ uscal,
                                                                              const uint32_t* __restrict__ st) {
    cw1 = step_f(st, kStepCw1, cw1); cw2 = step_f(st, kStepCw2, cw2);
    __shared__ double tot_s[kMaxSums], tot_u[kMaxSums];
    __shared__ double dice_s[UAPS_MAX_HEADS], dice_u[UAPS_MAX_HEADS];
    const int ns = sup_nsums(D, C), nu = unsup_nsums(D, C);
    for (int i = threadIdx.x; i < ns + nu; i += kFinalizeThreads) {
        if (i < ns) tot_s[i] = sums[i]; else tot_u[i - ns] = sums[i];
    }
    __syncthreads();
    finalize_pair_from_tot(tot_s, tot_u, dice_s, dice_u, D, C, N, ce_coef, dice_coef, cw1, cw2, eps, sscal, uscal);
}

// --------------------------------------------------------------------------------------------------
// F2: unsupervised backward.  Bytes per pixel: 4DC + 8 read, 4DC written.
// --------------------------------------------------------------------------------------------------
// N: the pixel count the loss was averaged over (local batch, or the gathered global batch)
// am = max(am, |x|, |y|) as ONE v_max3_f32 (round 5): `am = fmaxf(am, fabsf(g))` per gradient element compiled to a canonicalising
// multiply + v_max_f32 each, 32 VALU instructions per pixel at D = C = 4 -- the backward went from 54.6 to 67.8 us when the step
// began to track max|d logits| for out_conv's row weight gradient.  Pairs of elements per instruction: 8 per pixel.  NaN elements
// are ignored (IEEE maxnum), as with fmaxf: the bound stays finite and the convolution's own check reports the NaN.
__device__ __forceinline__ void amax3(float& am, float x, float y) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(am) : "v"(x), "v"(y));
}

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
                const float lm = is_normal_pos(m[c]) ? flog(m[c]) : 0.f;
                xm += m[c] * lm;
                lm1[c] = is_normal_pos(m[c]) ? lm + 1.f : 0.f;       // d xlogy(m,m)/dm, 0 at m == 0 (limit; reference NaNs)
                h[c] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                float dot = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) dot += m[c] * lp[k][c];
                gk[k] = gu - ge[k] * fexp(dot - xm);            // dL/dv_k ; exp(-v_k) = exp(dot - xm)
#pragma unroll
                for (int c = 0; c < C; ++c) h[c] += is_normal_pos(m[c]) ? gk[k] * (lm1[c] - lp[k][c]) : 0.f;
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
                    if constexpr (TRACK && (C % 2 == 0)) { if (c & 1) amax3(am, zv[j][c - 1][v], gr); }
                    else if constexpr (TRACK) am = fmaxf(am, fabsf(gr));
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
                    if constexpr (TRACK && (C % 2 == 0)) { if (c & 1) amax3(am, zv[c - 1][v], zv[c][v]); }
                    else if constexpr (TRACK) am = fmaxf(am, fabsf(zv[c][v]));
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
// The same work with every block taking an equal share of BOTH branches (grid-stride over all blocks in each): at the
// BASELINE sizes each thread then owns exactly two groups of either branch, where the split form above hands out 3.2 and
// 5.3 groups per thread and finishes with its slowest thread's 4 and 6.  Blocks alternate the order of the branches so that
// the loads of one half of the chip overlap the arithmetic of the other half.
template <int D, int C, int VS, int VU, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void pair_fwd_both_kernel(HeadPtrs<D> zl, HeadPtrs<D> zu, HeadWeights<D> w, int HW, long N,
                                                                       const int64_t* __restrict__ labels, int64_t* __restrict__ pseudo,
                                                                       float* __restrict__ var, float* __restrict__ part_s,
                                                                       float* __restrict__ part_u, const uint32_t* __restrict__ st) {
    if (st != nullptr) {
#pragma unroll
        for (int k = 0; k < D; ++k) w.w[k] = __uint_as_float(st[kStepW + k]);
    }
    const int bid = (int)blockIdx.x, nblk = (int)gridDim.x;
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
        if (((bid + phase) & 1) == 0) sup_fwd_body<D, C, VS>(zl, HW, N / VS, labels, part_s, bid, nblk);
        else unsup_fwd_body<D, C, VU>(zu, w, HW, N / VU, N, pseudo, var, part_u, bid, nblk);
    }
}
// Nloss: the pixel count the scalars were finalised with (= N, or the global count after an exchange of the sums)
// TRACK: also raise *amax_out to max|gradient element| (uaps_call_hints::out_amax), by amax3 above
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
