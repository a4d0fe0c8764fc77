// What does MI355X sustain for THE ARITHMETIC of the default convolution mode, with the convolution's geometry taken out?
//
// The fp16-split 3x3 kernels (uaps_amd/csrc/conv_split.hpp) evaluate an fp32 product as three fp16 partial
// products: the activation operand is fp32 in HBM, scaled by a power of two and split into two fp16 pieces WHILE IT IS STAGED; the
// weights are packed pre-split once per optimizer step; accumulation is fp32.  `roofline.frac` of conv_h32_kernel<64> has been 0.33 of
// 2500 / 3 TFLOP/s since round 2 and every restructuring left it there.  This probe is the same-arithmetic contraction on the
// programming guide's large-tile template -- few waves per CU (one or two per SIMD), a 64..128 x 64 register block per wave, software
// pipelined, one barrier per chunk -- for the M x N x K of the step's four dominant layers, so that the kernels can be held against
// something this chip has actually been seen to do with this arithmetic (VERDICT r5 "missing 3" / "next 2").
//
// Contraction: C[n][m] = sum over c < Cin, t < R of A[c][m] * B[c][t][n].  M = B*H*W pixels (contiguous per channel, NCHW), N = Cout,
// R = 9 taps.  R = 9 with the SAME rows of A for every tap is a 3x3 convolution without halo and without tap shifts: identical flops
// (2 M N Cin 9), identical HBM bytes (A once, C once, the small weight set from L2), identical staging work per flop (a staged
// activation chunk is used by all nine taps).  What it leaves out is what only the geometry costs: halo re-reads, the row / column
// shifts of the A fragments, image borders.  REUSE_A = true additionally keeps the A fragments in registers across the taps (an upper
// bound a real 3x3 kernel cannot fully reach: the kx taps need shifted rows); REUSE_A = false re-reads them from LDS per tap, as the
// convolution must.
//
//   hipcc --offload-arch=gfx950 -O3 tools/split_gemm_ceiling.hip -o tools/bin/split_gemm_ceiling && tools/bin/split_gemm_ceiling
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

// two fp16 pieces of 8 scaled fp32 values: hi = rn(x), lo = rn(x - hi)  (22 significant bits; conv_split2h of the product kernels)
__device__ __forceinline__ void split8(const float* x, u32x4& hi, u32x4& lo) {
    f16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 a = (_Float16)x[i];
        h[i] = a;
        l[i] = (_Float16)(x[i] - (float)a);
    }
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, l);
}

// MT x NT tiles of 16 x 16 per wave, WM x WN waves per workgroup, chunks of 32 channels, R taps per chunk.
// LDS: A image [2 buffers][2 pieces][4 k-groups][BM rows] of 16-byte units (8 consecutive channels of one row): a fragment read is one
// ds_read_b128 per lane, 16 lanes on 256 contiguous bytes.  B fragments come straight from global memory (L2-resident, packed in
// fragment order: one contiguous KiB per wave-instruction), prefetched one tap ahead.
template <int MT, int NT, int WM, int WN, int R, bool REUSE_A, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC) void split_gemm(const float* __restrict__ A, const u32x4* __restrict__ Bp, float* __restrict__ C,
                                                                int M, int N, int Cin, float a_scale, float out_scale) {
    constexpr int BM = WM * MT * 16, NTHR = WM * WN * 64;
    constexpr int QUADS = BM / 4;                 // staging items per k-group: 4 consecutive rows x 8 channels
    constexpr int ITEMS = (4 * QUADS + NTHR - 1) / NTHR;
    extern __shared__ u32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM;
    const int ntiles = N / 16, nt0 = (blockIdx.y * WN + wn) * NT;
    const int NC = Cin / 32;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 st[ITEMS][8];                            // the next chunk's activations on their way to LDS
    auto fetch = [&](int c) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = tid + it * NTHR;
            if (item < 4 * QUADS) {               // (wave-uniform: 4 * QUADS is a multiple of 64)
                const int kg = item / QUADS, q = item % QUADS;
                const float* p = A + (size_t)(c * 32 + kg * 8) * M + m0 + q * 4;
#pragma unroll
                for (int i = 0; i < 8; ++i) st[it][i] = *reinterpret_cast<const f32x4*>(p + (size_t)i * M);
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = tid + it * NTHR;
            if (item < 4 * QUADS) {
                const int kg = item / QUADS, q = item % QUADS;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = st[it][i][j] * a_scale;
                    u32x4 hi, lo;
                    split8(x, hi, lo);
                    lds[((buf * 2 + 0) * 4 + kg) * BM + q * 4 + j] = hi;
                    lds[((buf * 2 + 1) * 4 + kg) * BM + q * 4 + j] = lo;
                }
            }
        }
    };
    // B fragment (chunk c, tap t, piece p, wave's n-tile j): index ((((c R + t) 2 + p) ntiles + nt) 64 + lane)
    auto bload = [&](int c, int t, u32x4 (&b)[NT][2]) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[j][p] = Bp[((size_t)((c * R + t) * 2 + p) * ntiles + nt0 + j) * 64 + lane];
    };

    fetch(0);
    stage(0);
    __syncthreads();
    u32x4 bcur[NT][2], bnext[NT][2];
    bload(0, 0, bcur);
    for (int c = 0; c < NC; ++c) {
        const int buf = c & 1;
        if (c + 1 < NC) fetch(c + 1);
        u32x4 af[MT][2];
        const int abase = (lane >> 4) * BM + wm * MT * 16 + (lane & 15);
        if (REUSE_A) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) af[i][p] = lds[(buf * 2 + p) * 4 * BM + abase + i * 16];
        }
#pragma unroll
        for (int t = 0; t < R; ++t) {
            // the next tap's (or the next chunk's first) weight fragments while this tap's products run
            if (t + 1 < R) bload(c, t + 1, bnext);
            else if (c + 1 < NC) bload(c + 1, 0, bnext);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (!REUSE_A) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) af[i][p] = lds[(buf * 2 + p) * 4 * BM + abase + i * 16];
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    // activations are the matrix instruction's A operand (rows = pixels), weights its B operand (columns = output
                    // channels): a lane's accumulator then holds 4 consecutive pixels of one channel -> one 16-byte store
                    f32x4 v = acc[i][j];
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h(af[i][1]), as_h(bcur[j][0]), v, 0, 0, 0);      // lo x hi
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h(af[i][0]), as_h(bcur[j][1]), v, 0, 0, 0);      // hi x lo
                    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h(af[i][0]), as_h(bcur[j][0]), v, 0, 0, 0);      // hi x hi
                    acc[i][j] = v;
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) bcur[j][p] = bnext[j][p];
        }
        if (c + 1 < NC) stage(buf ^ 1);
        __syncthreads();
    }
    // accumulator of tile (i, j): lane l holds pixels 4 (l / 16) + r (r = 0..3) of channel l % 16  ->  64-byte runs per channel row
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = (nt0 + j) * 16 + (lane & 15);
            const int m = m0 + (wm * MT + i) * 16 + (lane >> 4) * 4;
            *reinterpret_cast<f32x4*>(C + (size_t)n * M + m) = acc[i][j] * out_scale;
        }
}

static float pow2_scale(float bound) {      // largest power of two s with s * bound < 2^15
    int e;
    frexpf(bound, &e);                       // bound = f * 2^e, f in [0.5, 1)
    return ldexpf(1.f, 15 - e);
}

struct Layer { const char* name; int B, H, W, Cin, Cout; };

template <int MT, int NT, int WM, int WN, int R, bool REUSE_A, int OCC>
static double run(const Layer& L, const char* form, bool check, FILE* js, bool first) {
    const int M = L.B * L.H * L.W, N = L.Cout, Cin = L.Cin;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    if (M % BM || N % BN || Cin % 32) { printf("  %-46s (shape does not tile)\n", form); return 0.0; }
    std::vector<float> hA((size_t)Cin * M), hB((size_t)Cin * R * N);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.f / 16777216.f)) * 2.f - 1.f; };
    for (auto& v : hA) { const float x = rnd(); v = x > 0.f ? x * 3.f : x * 0.03f; }      // LeakyReLU-shaped activations
    for (auto& v : hB) v = rnd() * 0.05f;
    const float sa = pow2_scale(3.f), sb = pow2_scale(0.05f);
    std::vector<_Float16> hBp((size_t)(Cin / 32) * R * 2 * (N / 16) * 64 * 8);
    for (int c = 0; c < Cin / 32; ++c)
        for (int t = 0; t < R; ++t)
            for (int nt = 0; nt < N / 16; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int ch = c * 32 + (lane / 16) * 8 + i, n = nt * 16 + lane % 16;
                        const float w = hB[((size_t)ch * R + t) * N + n] * sb;
                        const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
                        hBp[((((size_t)(c * R + t) * 2 + 0) * (N / 16) + nt) * 64 + lane) * 8 + i] = hi;
                        hBp[((((size_t)(c * R + t) * 2 + 1) * (N / 16) + nt) * 64 + lane) * 8 + i] = lo;
                    }
    float *dA, *dC;
    u32x4* dB;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dB, hBp.size() * 2));
    CK(hipMalloc(&dC, (size_t)N * M * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hBp.data(), hBp.size() * 2, hipMemcpyHostToDevice));
    auto kern = split_gemm<MT, NT, WM, WN, R, REUSE_A, OCC>;
    const size_t shmem = (size_t)2 * 2 * 4 * BM * 16;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const dim3 grid(M / BM, N / BN), block(WM * WN * 64);
    const float os = 1.f / (sa * sb);
    hipLaunchKernelGGL(kern, grid, block, shmem, 0, dA, dB, dC, M, N, Cin, sa, os);
    CK(hipDeviceSynchronize());
    double maxerr = 0.0, scale = 0.0;
    if (check) {      // 2000 sampled outputs against float64
        std::vector<float> hC((size_t)N * M);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < 2000; ++k) {
            const int m = (int)(((uint64_t)k * 2654435761u) % M), n = (k * 7) % N;
            double ref = 0.0, mag = 0.0;
            for (int c = 0; c < Cin; ++c)
                for (int t = 0; t < R; ++t) {
                    const double p = (double)hA[(size_t)c * M + m] * (double)hB[((size_t)c * R + t) * N + n];
                    ref += p; mag += fabs(p);
                }
            maxerr = fmax(maxerr, fabs(ref - hC[(size_t)n * M + m]) / mag);
            scale = fmax(scale, mag);
        }
    }
    // ~0.5 s of back-to-back launches first (the clock settles under the load), then 50 timed ones
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int rounds = 0; rounds < 200; ++rounds) {
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(kern, grid, block, shmem, 0, dA, dB, dC, M, N, Cin, sa, os);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float w;
        CK(hipEventElapsedTime(&w, e0, e1));
        if (w > 500.f) break;      // >= 0.5 s under this load
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, block, shmem, 0, dA, dB, dC, M, N, Cin, sa, os);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, flops = 2.0 * M * N * Cin * R, tf = flops / us / 1e6;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)kern));
    printf("  %-46s %5d wg x %3d thr  %3d regs %6.1f KB LDS  %7.1f us  %6.1f TFLOP/s fp32-equivalent (%.2f PF fp16 MFMA = %.3f of 2.5)",
           form, (int)(grid.x * grid.y), (int)block.x, fa.numRegs, shmem / 1024.0, us, tf, tf * 3 / 1e3, tf * 3 / 2500.0);
    if (check) printf("  max err / sum|ab| %.2e", maxerr);
    printf("\n");
    if (js) fprintf(js, "%s{\"form\": \"%s\", \"workgroups\": %d, \"threads\": %d, \"regs\": %d, \"us\": %.2f, \"tflops\": %.1f, \"max_rel_err\": %.3e}",
                    first ? "" : ", ", form, (int)(grid.x * grid.y), (int)block.x, fa.numRegs, us, tf, maxerr);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return tf;
}

int main(int argc, char** argv) {
    const char* jpath = argc > 1 ? argv[1] : nullptr;
    FILE* js = jpath ? fopen(jpath, "w") : nullptr;
    const Layer layers[] = {{"64 -> 64 @64^2", 32, 64, 64, 64, 64}, {"128 -> 64 @64^2", 32, 64, 64, 128, 64},
                            {"128 -> 128 @32^2", 32, 32, 32, 128, 128}, {"256 -> 128 @32^2", 32, 32, 32, 256, 128}};
    if (js) fprintf(js, "{\"arithmetic\": \"fp32 activations scaled by 2^k and split into two fp16 pieces in staging, pre-split fp16 weight pairs, three products "
                        "(hi*hi, hi*lo, lo*hi) on v_mfma_f32_16x16x32_f16, fp32 accumulation\", \"batch\": 32, \"taps\": 9, \"layers\": [");
    for (int li = 0; li < 4; ++li) {
        const Layer& L = layers[li];
        printf("%s, B = %d: M = %d pixels, N = %d, K = %d x 9 taps, %.2f GFLOP\n", L.name, L.B, L.B * L.H * L.W, L.Cout, L.Cin,
               2.0 * L.B * L.H * L.W * L.Cout * L.Cin * 9 / 1e9);
        if (js) fprintf(js, "%s{\"layer\": \"%s\", \"gflop\": %.3f, \"forms\": [", li ? ", " : "", L.name, 2.0 * L.B * L.H * L.W * L.Cout * L.Cin * 9 / 1e9);
        double best = 0.0, best_reread = 0.0;
        bool first = true;
        auto note = [&](double tf, bool reuse) { best = fmax(best, tf); if (!reuse) best_reread = fmax(best_reread, tf); first = false; };
        if (L.Cout == 64) {
            note(run<8, 4, 4, 1, 9, true, 1>(L, "128x64 per wave, 4 waves (1 / SIMD), A frags kept", true, js, first), true);
            note(run<8, 4, 4, 1, 9, false, 1>(L, "128x64 per wave, 4 waves (1 / SIMD), A re-read per tap", true, js, first), false);
            note(run<4, 4, 4, 1, 9, true, 2>(L, "64x64 per wave, 4 waves, 2 wg / CU, A frags kept", true, js, first), true);
            note(run<4, 4, 4, 1, 9, false, 2>(L, "64x64 per wave, 4 waves, 2 wg / CU, A re-read", true, js, first), false);
            note(run<4, 4, 8, 1, 9, false, 1>(L, "64x64 per wave, 8 waves (2 / SIMD), A re-read", false, js, first), false);
        } else {
            note(run<4, 4, 2, 2, 9, true, 1>(L, "64x64 per wave, 4 waves (1 / SIMD), A frags kept", true, js, first), true);
            note(run<4, 4, 2, 2, 9, false, 1>(L, "64x64 per wave, 4 waves (1 / SIMD), A re-read per tap", true, js, first), false);
            note(run<4, 4, 1, 2, 9, false, 2>(L, "64x64 per wave, 2 waves, 2 wg / CU, A re-read", false, js, first), false);
            note(run<8, 4, 2, 2, 9, true, 1>(L, "128x64 per wave, 4 waves, half the CUs, A kept", false, js, first), true);
            note(run<2, 4, 4, 2, 9, false, 2>(L, "32x64 per wave, 8 waves, 2 wg / CU, A re-read", false, js, first), false);
        }
        printf("  -> best %.1f TFLOP/s (A fragments kept), %.1f with A re-read per tap (what a 3x3 kernel must do)\n", best, best_reread);
        if (js) fprintf(js, "], \"best_tflops\": %.1f, \"best_tflops_reread\": %.1f}", best, best_reread);
    }
    if (js) { fprintf(js, "]}\n"); fclose(js); }
    // ---- the 32-output-channel layers of the 128 x 128 level (conv_h32t_kernel<32>: 11 launches per step at 165-222 TFLOP/s): not part
    // of the JSON's four layers (roofline.practical_peak is defined on those), printed for DESIGN.md section 3.2
    const Layer narrow[] = {{"32 -> 32 @128^2", 32, 128, 128, 32, 32}, {"64 -> 32 @128^2", 32, 128, 128, 64, 32}};
    for (const Layer& L : narrow) {
        printf("%s, B = %d: M = %d pixels, N = %d, K = %d x 9 taps, %.2f GFLOP\n", L.name, L.B, L.B * L.H * L.W, L.Cout, L.Cin,
               2.0 * L.B * L.H * L.W * L.Cout * L.Cin * 9 / 1e9);
        run<8, 2, 4, 1, 9, false, 1>(L, "128x32 per wave, 4 waves (1 / SIMD), A re-read per tap", true, nullptr, true);
        run<8, 2, 4, 1, 9, false, 2>(L, "128x32 per wave, 4 waves, 2 wg / CU, A re-read", false, nullptr, true);
        run<4, 2, 4, 1, 9, false, 2>(L, "64x32 per wave, 4 waves, 2 wg / CU, A re-read", false, nullptr, true);
        run<4, 2, 4, 1, 9, false, 4>(L, "64x32 per wave, 4 waves, 4 wg / CU, A re-read", false, nullptr, true);
        run<4, 2, 8, 1, 9, false, 2>(L, "64x32 per wave, 8 waves, 2 wg / CU, A re-read", false, nullptr, true);
        run<8, 2, 4, 1, 9, true, 1>(L, "128x32 per wave, 4 waves (1 / SIMD), A frags kept", false, nullptr, true);
    }
    // ---- the same kernel as a plain GEMM (R = 1): the ResNet-50 configuration's 1x1 bottleneck projections (utilities/resnet.py:
    // 55-95) at the configs[4] per-GPU batch, 8 + 8 images of 640 x 640 -> 80 x 80 maps.  Every staged activation element now serves
    // ONE tap: the split arithmetic of the staging is no longer amortised over nine.  Shipped conv_g1h256_kernel: 692-751 us on the
    // first two (profiles/r05_g1_wide_ab.txt, r05_g1_ablation.txt).
    const Layer pw[] = {{"1x1 2048 -> 512 @80^2, B = 16", 16, 80, 80, 2048, 512}, {"1x1 512 -> 2048 @80^2, B = 16", 16, 80, 80, 512, 2048},
                        {"1x1 1024 -> 512 @80^2, B = 16", 16, 80, 80, 1024, 512}};
    for (const Layer& L : pw) {
        printf("%s: M = %d, N = %d, K = %d, %.1f GFLOP\n", L.name, L.B * L.H * L.W, L.Cout, L.Cin, 2.0 * L.B * L.H * L.W * L.Cout * L.Cin / 1e9);
        run<8, 4, 2, 2, 1, true, 1>(L, "128x64 per wave, 4 waves (1 / SIMD), 256 x 128 tile", false, nullptr, true);
        run<8, 4, 4, 1, 1, true, 1>(L, "128x64 per wave, 4 waves (1 / SIMD), 512 x 64 tile", false, nullptr, true);
        run<4, 4, 2, 2, 1, true, 2>(L, "64x64 per wave, 4 waves, 2 wg / CU, 128 x 128 tile", false, nullptr, true);
        run<4, 4, 4, 2, 1, true, 1>(L, "64x64 per wave, 8 waves (2 / SIMD), 256 x 128 tile", false, nullptr, true);
    }
    return 0;
}
