// The companion of tools/split_gemm_ceiling.hip for the WEIGHT-GRADIENT arithmetic of the default convolution mode.
//
// conv_hwrw_kernel<4,2,2> (uaps_amd/csrc/conv_split_wrw.hpp) is, on most boxes, the kernel with the largest
// share of the step, at 0.27-0.29 of 2500 / 3 TFLOP/s.  Its contraction is dw[co][ci][tap] = sum over pixels of dy[co][p] * x[ci][p + tap]:
// M = Cout, N = Cin x 9, K = pixels -- a short, fat GEMM with a very long K, split over the workgroups -- and BOTH operands are fp32
// activations in HBM that are scaled by a power of two and split into two fp16 pieces while they are staged (the forward's weights
// come pre-split).  This probe is that arithmetic with the geometry taken out: the nine taps use the SAME staged x fragment (no halo
// rows, no column shifts, no image borders), pixels are one contiguous run per channel, the per-split partial sums are written once
// (the fixed-order reduction over the splits is a separate launch in the product and is not part of either figure).  Forms: the
// shipped blocking (32 x 32 channels per workgroup, one 16 x 16 block per wave, two or more workgroups per CU) and the register-
// blocked one (64 x 64 per workgroup, 2 x 2 blocks per wave, one workgroup per CU), double-buffered LDS, one barrier per chunk.
//
//   hipcc --offload-arch=gfx950 -O3 tools/split_wrw_ceiling.hip -o tools/bin/split_wrw_ceiling && tools/bin/split_wrw_ceiling [out.json]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ f16x8 as_h(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, float sc, u32x4& hi, u32x4& lo) {
    f16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = (i < 4 ? a[i] : b[i - 4]) * sc;
        const _Float16 t = (_Float16)x;
        h[i] = t;
        l[i] = (_Float16)(x - (float)t);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

constexpr int KC = 128;                  // pixels per chunk: 16 k-groups of 8, four 32-pixel matrix steps
constexpr int PLU = KC / 8 + 1;          // units (16 B = 8 fp16 pixels) per channel row in LDS, + 1: 16 lanes on 16 different bank groups

// Workgroup: (WCO x WCI) waves, each MCO x MCI blocks of 16 x 16 channels x R taps.  D [Cout][P], X [Cin][P] fp32; the workgroup
// (cob, cib, split) sums pixels [split * P / nsplit, (split + 1) * P / nsplit) into slab[split][tap][Cout][Cin].
template <int MCO, int MCI, int WCO, int WCI, int R, int OCC>
__global__ __launch_bounds__(WCO * WCI * 64, OCC) void split_wrw(const float* __restrict__ D, const float* __restrict__ X, float* __restrict__ slab,
                                                                 int Cout, int Cin, int P, int nsplit, float sd, float sx, float out_scale) {
    constexpr int NTHR = WCO * WCI * 64, BCO = WCO * MCO * 16, BCI = WCI * MCI * 16;
    constexpr int UD = BCO * (KC / 8), UX = BCI * (KC / 8);            // staging units per chunk
    constexpr int ND = UD / NTHR, NX = UX / NTHR;
    static_assert(UD % NTHR == 0 && UX % NTHR == 0, "units per thread");
    extern __shared__ u32x4 lds[];                                       // [buf][dy hi | dy lo | x hi | x lo]
    constexpr int SD = BCO * PLU, SX = BCI * PLU, SBUF = 2 * SD + 2 * SX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wco = wave / WCI, wci = wave % WCI;
    const int j = lane & 15, kq = lane >> 4;
    const int ncib = Cin / BCI, ncob = Cout / BCO;
    int bid = blockIdx.x;
    const int cib = bid % ncib; bid /= ncib;
    const int cob = bid % ncob;
    const int split = bid / ncob;
    const int co0 = cob * BCO, ci0 = cib * BCI;
    const int nchunks = P / KC;
    const int c_begin = (int)((long)nchunks * split / nsplit), c_end = (int)((long)nchunks * (split + 1) / nsplit);

    f32x4 acc[MCO][MCI][R];
#pragma unroll
    for (int a = 0; a < MCO; ++a)
#pragma unroll
        for (int b = 0; b < MCI; ++b)
#pragma unroll
            for (int t = 0; t < R; ++t) acc[a][b][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // unit u -> (channel u / 16, k-group u % 16): the 16 k-groups of a channel are 512 contiguous bytes
    f32x4 rd[ND][2], rx[NX][2];
    auto fetch = [&](int c) {
        const size_t p0 = (size_t)c * KC;
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            const int u = tid + n * NTHR, ch = u >> 4, g = u & 15;
            const float* p = D + (size_t)(co0 + ch) * P + p0 + g * 8;
            rd[n][0] = *reinterpret_cast<const f32x4*>(p); rd[n][1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int n = 0; n < NX; ++n) {
            const int u = tid + n * NTHR, ch = u >> 4, g = u & 15;
            const float* p = X + (size_t)(ci0 + ch) * P + p0 + g * 8;
            rx[n][0] = *reinterpret_cast<const f32x4*>(p); rx[n][1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
    };
    auto stage = [&](int buf) {
        u32x4* base = lds + buf * SBUF;
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            const int u = tid + n * NTHR, ch = u >> 4, g = u & 15;
            u32x4 hi, lo;
            split8(rd[n][0], rd[n][1], sd, hi, lo);
            base[ch * PLU + g] = hi; base[SD + ch * PLU + g] = lo;
        }
#pragma unroll
        for (int n = 0; n < NX; ++n) {
            const int u = tid + n * NTHR, ch = u >> 4, g = u & 15;
            u32x4 hi, lo;
            split8(rx[n][0], rx[n][1], sx, hi, lo);
            base[2 * SD + ch * PLU + g] = hi; base[2 * SD + SX + ch * PLU + g] = lo;
        }
    };

    if (c_begin < c_end) { fetch(c_begin); stage(0); }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        const u32x4* base = lds + buf * SBUF;
        if (c + 1 < c_end) fetch(c + 1);
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
            u32x4 af[MCO][2], bf[MCI][2];
#pragma unroll
            for (int a = 0; a < MCO; ++a)
#pragma unroll
                for (int p = 0; p < 2; ++p) af[a][p] = base[p * SD + ((wco * MCO + a) * 16 + j) * PLU + ks * 4 + kq];
#pragma unroll
            for (int b = 0; b < MCI; ++b)
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[b][p] = base[2 * SD + p * SX + ((wci * MCI + b) * 16 + j) * PLU + ks * 4 + kq];
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int a = 0; a < MCO; ++a)
#pragma unroll
                    for (int b = 0; b < MCI; ++b) {
                        f32x4 v = acc[a][b][t];
                        v = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h(af[a][1]), as_h(bf[b][0]), v, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h(af[a][0]), as_h(bf[b][1]), v, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h(af[a][0]), as_h(bf[b][0]), v, 0, 0, 0);
                        acc[a][b][t] = v;
                    }
        }
        if (c + 1 < c_end) stage(buf ^ 1);
        __syncthreads();
    }
    // accumulator (a, b, t): lane holds rows (co) 4 kq + r, column (ci) j
    float* out = slab + (size_t)split * R * Cout * Cin;
#pragma unroll
    for (int a = 0; a < MCO; ++a)
#pragma unroll
        for (int b = 0; b < MCI; ++b)
#pragma unroll
            for (int t = 0; t < R; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + (wco * MCO + a) * 16 + kq * 4 + r, ci = ci0 + (wci * MCI + b) * 16 + j;
                    out[((size_t)t * Cout + co) * Cin + ci] = acc[a][b][t][r] * out_scale;
                }
}

static float pow2_scale(float bound) {
    int e;
    frexpf(bound, &e);
    return ldexpf(1.f, 15 - e);
}

struct Layer { const char* name; int P, Cin, Cout; };

template <int MCO, int MCI, int WCO, int WCI, int OCC>
static double run(const Layer& L, const char* form, int wg_target, bool check, FILE* js, bool first) {
    constexpr int R = 9, BCO = WCO * MCO * 16, BCI = WCI * MCI * 16, NTHR = WCO * WCI * 64;
    if (L.Cout % BCO || L.Cin % BCI || L.P % KC) { printf("  %-52s (shape does not tile)\n", form); return 0.0; }
    const int blocks = (L.Cout / BCO) * (L.Cin / BCI);
    int nsplit = wg_target / blocks;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > L.P / KC) nsplit = L.P / KC;
    std::vector<float> hD((size_t)L.Cout * L.P), hX((size_t)L.Cin * L.P);
    uint32_t s = 2463534242u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.f / 16777216.f)) * 2.f - 1.f; };
    for (auto& v : hX) { const float x = rnd(); v = x > 0.f ? x * 3.f : x * 0.03f; }      // LeakyReLU-shaped activations
    for (auto& v : hD) v = rnd() * 0.01f;                                                  // gradients
    const float sd = pow2_scale(0.01f), sx = pow2_scale(3.f);
    float *dD, *dX, *dS;
    const size_t slab = (size_t)nsplit * R * L.Cout * L.Cin;
    CK(hipMalloc(&dD, hD.size() * 4)); CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dS, slab * 4));
    CK(hipMemcpy(dD, hD.data(), hD.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    auto kern = split_wrw<MCO, MCI, WCO, WCI, R, OCC>;
    const size_t shmem = (size_t)2 * (2 * BCO + 2 * BCI) * PLU * 16;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const dim3 grid(blocks * nsplit), block(NTHR);
    const float os = 1.f / (sd * sx);
    hipLaunchKernelGGL(kern, grid, block, shmem, 0, dD, dX, dS, L.Cout, L.Cin, L.P, nsplit, sd, sx, os);
    CK(hipDeviceSynchronize());
    double maxerr = 0.0;
    if (check) {      // 300 sampled outputs (tap 0 and tap 8: all taps hold the same sum here) against float64
        std::vector<float> hS(slab);
        CK(hipMemcpy(hS.data(), dS, slab * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < 300; ++k) {
            const int co = (k * 7) % L.Cout, ci = (k * 13) % L.Cin, t = (k & 1) ? 8 : 0;
            double ref = 0.0, mag = 0.0, got = 0.0;
            for (int p = 0; p < L.P; ++p) { const double q = (double)hD[(size_t)co * L.P + p] * (double)hX[(size_t)ci * L.P + p]; ref += q; mag += fabs(q); }
            for (int sp = 0; sp < nsplit; ++sp) got += hS[(((size_t)sp * R + t) * L.Cout + co) * L.Cin + ci];
            maxerr = fmax(maxerr, fabs(ref - got) / mag);
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int rounds = 0; rounds < 200; ++rounds) {      // ~0.5 s under this load first: the clock settles
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(kern, grid, block, shmem, 0, dD, dX, dS, L.Cout, L.Cin, L.P, nsplit, sd, sx, os);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float w;
        CK(hipEventElapsedTime(&w, e0, e1));
        if (w > 500.f) break;
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, block, shmem, 0, dD, dX, dS, L.Cout, L.Cin, L.P, nsplit, sd, sx, os);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, flops = 2.0 * L.P * L.Cout * L.Cin * R, tf = flops / us / 1e6;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)kern));
    printf("  %-52s %4d wg (%3d splits) x %3d thr  %3d regs %5.1f KB LDS  %7.1f us  %6.1f TFLOP/s fp32-equivalent (= %.3f of 2500 / 3)",
           form, (int)grid.x, nsplit, NTHR, fa.numRegs, shmem / 1024.0, us, tf, tf / (2500.0 / 3));
    if (check) printf("  max err / sum|ab| %.2e", maxerr);
    printf("\n");
    if (js) fprintf(js, "%s{\"form\": \"%s\", \"workgroups\": %d, \"splits\": %d, \"threads\": %d, \"regs\": %d, \"us\": %.2f, \"tflops\": %.1f, \"max_rel_err\": %.3e}",
                    first ? "" : ", ", form, (int)grid.x, nsplit, NTHR, fa.numRegs, us, tf, maxerr);
    CK(hipFree(dD)); CK(hipFree(dX)); CK(hipFree(dS));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return tf;
}

int main(int argc, char** argv) {
    const char* jpath = argc > 1 ? argv[1] : nullptr;
    FILE* js = jpath ? fopen(jpath, "w") : nullptr;
    // the weight gradients of the step's four dominant layers at B = 32 (P = B * H * W pixels)
    const Layer layers[] = {{"64 -> 64 @64^2", 32 * 64 * 64, 64, 64}, {"128 -> 64 @64^2", 32 * 64 * 64, 128, 64},
                            {"128 -> 128 @32^2", 32 * 32 * 32, 128, 128}, {"256 -> 128 @32^2", 32 * 32 * 32, 256, 128}};
    if (js) fprintf(js, "{\"arithmetic\": \"weight gradient: BOTH operands fp32 in HBM, scaled by 2^k and split into two fp16 pieces in staging, three products "
                        "(hi*hi, hi*lo, lo*hi) on v_mfma_f32_16x16x32_f16, fp32 accumulation, nine taps on the same staged fragment, per-split partial "
                        "sums written once\", \"batch\": 32, \"taps\": 9, \"layers\": [");
    for (int li = 0; li < 4; ++li) {
        const Layer& L = layers[li];
        const double gf = 2.0 * L.P * L.Cout * L.Cin * 9 / 1e9;
        printf("%s, B = 32: M = %d, N = %d x 9 taps, K = %d pixels, %.2f GFLOP\n", L.name, L.Cout, L.Cin, L.P, gf);
        if (js) fprintf(js, "%s{\"layer\": \"%s\", \"gflop\": %.3f, \"forms\": [", li ? ", " : "", L.name, gf);
        double best = 0.0;
        bool first = true;
        auto note = [&](double tf) { best = fmax(best, tf); first = false; };
        note(run<1, 1, 2, 2, 2>(L, "32x32 per workgroup, 16x16 per wave, 2 wg / CU (shipped blocking)", 512, true, js, first));
        note(run<2, 2, 2, 2, 1>(L, "64x64 per workgroup, 32x32 per wave, 1 wg / CU", 256, true, js, first));
        note(run<2, 1, 2, 2, 2>(L, "64x32 per workgroup, 32x16 per wave, 2 wg / CU", 512, false, js, first));
        note(run<1, 2, 2, 2, 2>(L, "32x64 per workgroup, 16x32 per wave, 2 wg / CU", 512, false, js, first));
        note(run<2, 2, 1, 2, 2>(L, "32x64 per workgroup, 32x32 per wave, 2 waves, 2 wg / CU", 512, false, js, first));
        printf("  -> best %.1f TFLOP/s\n", best);
        if (js) fprintf(js, "], \"best_tflops\": %.1f}", best);
    }
    if (js) { fprintf(js, "]}\n"); fclose(js); }
    return 0;
}
