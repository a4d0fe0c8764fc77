// Memory floor of the loss kernels' access pattern: NP planes of N floats each, a lane reads VEC consecutive floats from every plane
// (16 planes x 1 KiB per wave-instruction set), minimal arithmetic, optional int64 label stream and a 8-byte-per-pixel write.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
template <int NP, int VEC, int WORK>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ base, long plane_stride, long ngroups, const long long* __restrict__ lab,
                                             long long* __restrict__ out, float* __restrict__ sink) {
    float acc = 0.f;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (long)gridDim.x * 256) {
        float v[NP][VEC];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if constexpr (VEC == 4) { float4 t = *reinterpret_cast<const float4*>(base + p * plane_stride + g * 4); v[p][0] = t.x; v[p][1] = t.y; v[p][2] = t.z; v[p][3] = t.w; }
            else if constexpr (VEC == 2) { float2 t = *reinterpret_cast<const float2*>(base + p * plane_stride + g * 2); v[p][0] = t.x; v[p][1] = t.y; }
            else v[p][0] = base[p * plane_stride + g];
        }
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float x = v[p][k];
#pragma unroll
                for (int w = 0; w < WORK; ++w) x = __builtin_fmaf(x, 1.0001f, 0.5f);      // WORK VALU ops per element
                s += x;
            }
        acc += s;
        if (out) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) out[g * VEC + k] = (long long)(s > 0.f) + (lab ? lab[g * VEC + k] : 0);
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int NP, int VEC, int WORK> float run(const float* d, long N, int blocks, long long* lab, long long* out, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long ng = N / VEC;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<NP, VEC, WORK>), dim3(blocks), dim3(256), 0, 0, d, N, ng, lab, out, sink);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<NP, VEC, WORK>), dim3(blocks), dim3(256), 0, 0, d, N, ng, lab, out, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 10 * 1e3f;
}
int main() {
    const long N = 16L * 256 * 256;      // pixels of one branch at B = 16, 256 x 256
    float* d; long long *lab, *out; float* sink;
    hipMalloc(&d, 16 * N * 4 + 4096); hipMalloc(&lab, N * 8); hipMalloc(&out, N * 8); hipMalloc(&sink, 4);
    hipMemset(d, 0, 16 * N * 4); hipMemset(lab, 0, N * 8);
    const double rb = 16.0 * N * 4;
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        float t4 = run<16, 4, 0>(d, N, blocks, nullptr, nullptr, sink), t2 = run<16, 2, 0>(d, N, blocks, nullptr, nullptr, sink), t1 = run<16, 1, 0>(d, N, blocks, nullptr, nullptr, sink);
        printf("read-only 16 planes, blocks %4d: VEC4 %6.1f us %6.0f GB/s | VEC2 %6.1f us %6.0f GB/s | VEC1 %6.1f us %6.0f GB/s\n", blocks, t4, rb / t4 / 1e3, t2, rb / t2 / 1e3, t1, rb / t1 / 1e3);
    }
    for (int blocks : {512, 1024, 4096}) {
        float a = run<16, 4, 4>(d, N, blocks, nullptr, nullptr, sink), b = run<16, 4, 16>(d, N, blocks, nullptr, nullptr, sink), c = run<16, 4, 32>(d, N, blocks, lab, out, sink);
        printf("VEC4 blocks %4d: 4 ops/elt %6.1f us | 16 ops/elt %6.1f us | 32 ops/elt + labels + 8B store %6.1f us\n", blocks, a, b, c);
    }
    return 0;
}
