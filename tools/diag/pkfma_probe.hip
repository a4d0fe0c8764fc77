// Does v_pk_fma_f32 with an op_sel operand swizzle compute what the ISA says while other waves keep the matrix pipe busy?
// Found in round 3: conv_small_bn_kernel<8, 4> (packed fp32 fma chains whose weight operand is selected with op_sel:[0,1,0])
// returned, in about 1 launch of 100 while a second process ran 16-bit MFMA kernels on the same card, one wrong product in
// the LOW half of one accumulator pair for lanes 48..63 of one wave.  This probe isolates the instruction:
//   pkfma_probe probe FORM LAUNCHES     every lane runs ITERS packed fmas (inline asm, the form under test) and the same
//                                       arithmetic as scalar v_fma_f32; any bit difference is counted and located
//   pkfma_probe hammer SECONDS          back-to-back v_mfma_f32_16x16x32_f16 on all CUs (run it in a second process, or pass
//                                       `both` to run it on a second stream of the probing process)
// FORM 0: no op_sel   1: the conv_small mix (op_sel_hi:[1,0,1] x2, op_sel:[0,1,0] x2)   2: all op_sel:[0,1,0]   3: all op_sel_hi:[1,0,1]
//      4: op_sel:[1,0,0] (src0)   5: v_pk_mul_f32 op_sel:[0,1]   6: v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]   7: op_sel:[0,0,1] (src2)
// hammer KIND (third argument): 0 = v_mfma_f32_16x16x32_f16, 1 = v_mfma_f32_32x32x16_bf16, 2 = v_mfma_f32_16x16x4_f32, 3 = v_fma_f32 only
//   hipcc --offload-arch=gfx950 -O3 tools/diag/pkfma_probe.hip -o tools/bin/pkfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define PK(form_str, d, a, b) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 " form_str : "+v"(d) : "v"(a), "v"(b))

template <int FORM>
__global__ __launch_bounds__(256) void probe_kernel(unsigned* bad, unsigned* where, int iters, unsigned seed) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    unsigned s = t * 2654435761u + seed;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f; };
    f32x2 d[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { d[i] = f32x2{0.f, 0.f}; r[i] = d[i]; }
    for (int it = 0; it < iters; ++it) {
        const f32x2 a = f32x2{rnd(), rnd()};
        const f32x2 b = f32x2{rnd(), rnd()};
        if constexpr (FORM == 0) {
            PK("", d[0], a, b); PK("", d[1], a, b); PK("", d[2], a, b); PK("", d[3], a, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) { r[i].x = __builtin_fmaf(a.x, b.x, r[i].x); r[i].y = __builtin_fmaf(a.y, b.y, r[i].y); }
        } else if constexpr (FORM == 1) {
            PK("op_sel_hi:[1,0,1]", d[0], a, b); PK("op_sel_hi:[1,0,1]", d[1], a, b);
            PK("op_sel:[0,1,0]", d[2], a, b); PK("op_sel:[0,1,0]", d[3], a, b);
#pragma unroll
            for (int i = 0; i < 2; ++i) { r[i].x = __builtin_fmaf(a.x, b.x, r[i].x); r[i].y = __builtin_fmaf(a.y, b.x, r[i].y); }
#pragma unroll
            for (int i = 2; i < 4; ++i) { r[i].x = __builtin_fmaf(a.x, b.y, r[i].x); r[i].y = __builtin_fmaf(a.y, b.y, r[i].y); }
        } else if constexpr (FORM == 2) {
            PK("op_sel:[0,1,0]", d[0], a, b); PK("op_sel:[0,1,0]", d[1], a, b); PK("op_sel:[0,1,0]", d[2], a, b); PK("op_sel:[0,1,0]", d[3], a, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) { r[i].x = __builtin_fmaf(a.x, b.y, r[i].x); r[i].y = __builtin_fmaf(a.y, b.y, r[i].y); }
        } else if constexpr (FORM == 3) {
            PK("op_sel_hi:[1,0,1]", d[0], a, b); PK("op_sel_hi:[1,0,1]", d[1], a, b); PK("op_sel_hi:[1,0,1]", d[2], a, b); PK("op_sel_hi:[1,0,1]", d[3], a, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) { r[i].x = __builtin_fmaf(a.x, b.x, r[i].x); r[i].y = __builtin_fmaf(a.y, b.x, r[i].y); }
        } else if constexpr (FORM == 4) {          // src0: low half from the high register
            PK("op_sel:[1,0,0]", d[0], a, b); PK("op_sel:[1,0,0]", d[1], a, b); PK("op_sel:[1,0,0]", d[2], a, b); PK("op_sel:[1,0,0]", d[3], a, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) { r[i].x = __builtin_fmaf(a.y, b.x, r[i].x); r[i].y = __builtin_fmaf(a.y, b.y, r[i].y); }
        } else if constexpr (FORM == 5) {          // v_pk_mul_f32, src1 low half from the high register, then a natural packed add
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 m;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(m) : "v"(a), "v"(b));
                d[i] += m;
                r[i].x += a.x * b.y; r[i].y += a.y * b.y;
            }
        } else if constexpr (FORM == 6) {          // v_pk_add_f32 with the halves of src1 swapped
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(d[i]) : "v"(b));
                r[i].x += b.y; r[i].y += b.x;
            }
        } else {                                   // FORM 7: the accumulator operand (src2) low half from the high register
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 o;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(o) : "v"(a), "v"(b), "v"(d[i]));
                d[i] = o;
                const float lo = __builtin_fmaf(a.x, b.x, r[i].y), hi = __builtin_fmaf(a.y, b.y, r[i].y);
                r[i].x = lo; r[i].y = hi;
            }
        }
    }
    unsigned n = 0, first = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (__builtin_bit_cast(unsigned, d[i].x) != __builtin_bit_cast(unsigned, r[i].x)) { ++n; if (first == 0xffffffffu) first = i * 2; }
        if (__builtin_bit_cast(unsigned, d[i].y) != __builtin_bit_cast(unsigned, r[i].y)) { ++n; if (first == 0xffffffffu) first = i * 2 + 1; }
    }
    if (n) {
        const unsigned k = atomicAdd(bad, n);
        if (k < 64) where[k] = (threadIdx.x & 63) | (first << 8) | ((threadIdx.x >> 6) << 16);      // lane, accumulator half, wave
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// KIND 0: v_mfma_f32_16x16x32_f16   1: v_mfma_f32_32x32x16_bf16   2: v_mfma_f32_16x16x4_f32   3: plain v_fma_f32 (no matrix instruction)
template <int KIND>
__global__ __launch_bounds__(256) void hammer2_kernel(float* out, int iters) {
    float s = 0.f;
    if constexpr (KIND == 1) {
        f32x16 acc[2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.37f * (float)((threadIdx.x * 7 + j * 3) % 13) - 2.f); b[j] = (__bf16)(0.21f * (float)((threadIdx.x * 5 + j) % 11) - 1.f); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    } else if constexpr (KIND == 2) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float a = 0.37f * (float)(threadIdx.x % 13) - 2.f, b = 0.21f * (float)(threadIdx.x % 11) - 1.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {
        float acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (float)i;
        const float a = 0.999f + 1e-6f * threadIdx.x, b = 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
        }
        for (int i = 0; i < 8; ++i) s += acc[i];
    }
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void hammer_kernel(float* out, int iters) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.37f * (float)((threadIdx.x * 7 + j * 3) % 13) - 2.f); b[j] = (_Float16)(0.21f * (float)((threadIdx.x * 5 + j) % 11) - 1.f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) out[0] = s;
}

template <int FORM> void launch_probe(unsigned* bad, unsigned* where, int blocks, int iters, unsigned seed, hipStream_t st) {
    hipLaunchKernelGGL(probe_kernel<FORM>, dim3(blocks), dim3(256), 0, st, bad, where, iters, seed);
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: pkfma_probe probe FORM LAUNCHES [both] | hammer SECONDS\n"); return 2; }
    float* out; (void)hipMalloc(&out, 4);
    if (!strcmp(argv[1], "hammer")) {
        const double secs = atof(argv[2]);
        const auto t0 = std::chrono::steady_clock::now();
        long n = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
            const int kind = argc > 3 ? atoi(argv[3]) : 0;
            for (int i = 0; i < 20; ++i) {
                if (kind == 0) hipLaunchKernelGGL(hammer_kernel, dim3(512), dim3(256), 0, 0, out, 2000);
                else if (kind == 1) hipLaunchKernelGGL(hammer2_kernel<1>, dim3(512), dim3(256), 0, 0, out, 2000);
                else if (kind == 2) hipLaunchKernelGGL(hammer2_kernel<2>, dim3(512), dim3(256), 0, 0, out, 2000);
                else hipLaunchKernelGGL(hammer2_kernel<3>, dim3(512), dim3(256), 0, 0, out, 8000);
            }
            (void)hipDeviceSynchronize(); n += 20;
        }
        printf("hammer: %ld launches in %.1f s\n", n, secs);
        return 0;
    }
    const int form = atoi(argv[2]), launches = argc > 3 ? atoi(argv[3]) : 2000;
    const bool both = argc > 4 && !strcmp(argv[4], "both");
    unsigned *bad, *where;
    (void)hipMalloc(&bad, 4); (void)hipMalloc(&where, 64 * 4);
    (void)hipMemset(bad, 0, 4); (void)hipMemset(where, 0xff, 256);
    hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    const int blocks = 512, iters = 1024;
    for (int l = 0; l < launches; ++l) {
        if (both && l % 4 == 0) hipLaunchKernelGGL(hammer_kernel, dim3(512), dim3(256), 0, s2, out, 4000);
        switch (form) {
            case 0: launch_probe<0>(bad, where, blocks, iters, (unsigned)l, s1); break;
            case 1: launch_probe<1>(bad, where, blocks, iters, (unsigned)l, s1); break;
            case 2: launch_probe<2>(bad, where, blocks, iters, (unsigned)l, s1); break;
            case 3: launch_probe<3>(bad, where, blocks, iters, (unsigned)l, s1); break;
            case 4: launch_probe<4>(bad, where, blocks, iters, (unsigned)l, s1); break;
            case 5: launch_probe<5>(bad, where, blocks, iters, (unsigned)l, s1); break;
            case 6: launch_probe<6>(bad, where, blocks, iters, (unsigned)l, s1); break;
            default: launch_probe<7>(bad, where, blocks, iters, (unsigned)l, s1); break;
        }
        if (l % 64 == 63) (void)hipDeviceSynchronize();
    }
    (void)hipDeviceSynchronize();
    unsigned hb, hw[64];
    (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hw, where, 256, hipMemcpyDeviceToHost);
    printf("form %d%s: %d launches x %d blocks x 256 lanes x %d iterations x 4 packed fmas: %u accumulator halves differ from the scalar chain\n",
           form, both ? " (+ in-process hammer stream)" : "", launches, blocks, iters, hb);
    for (unsigned i = 0; i < (hb < 24 ? hb : 24); ++i)
        if (hw[i] != 0xffffffffu) printf("   lane %u wave %u accumulator %u half %s\n", hw[i] & 63, (hw[i] >> 16) & 3, ((hw[i] >> 8) & 255) / 2, ((hw[i] >> 8) & 1) ? "hi" : "lo");
    return 0;
}
