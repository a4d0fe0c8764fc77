// Sustained rate of back-to-back v_mfma_f32_16x16x4_f32 with all operands in registers (no memory traffic):
// the practical ceiling the conv kernels' MFMA fractions should be read against (clock under matrix load included).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0, long long* clk = nullptr) {
    const long long c0 = clock64(), w0 = wall_clock64();
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) out[0] = s;     // keep the loop alive
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int NACC>
void run(int waves_per_simd, int iters) {
    float* out; (void)hipMalloc(&out, 4);
    const int blocks = 256 * waves_per_simd;          // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.f, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    long long* clk; (void)hipMalloc(&clk, 16);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f, clk);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    const double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("acc=%d waves/SIMD=%d iters=%d: %.3f ms  %.1f TFLOP/s   clock64/wall_clock64 = %.3f (x100 MHz if clock64 is the shader clock)\n",
           NACC, waves_per_simd, iters, ms, flops / ms / 1e9, (double)h[0] / (double)h[1]);
    (void)hipFree(out);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<8>(1, 20000); run<8>(2, 20000); run<4>(2, 40000); run<8>(2, 200000);
        run<16>(2, 20000); run<8>(4, 20000); run<16>(4, 20000); run<8>(8, 20000);
    }
    return 0;
}
