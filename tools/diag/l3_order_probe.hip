// Does the order in which a consumer reads a tensor its producer has just written matter (Infinity Cache, 256 MiB, memory side)?
// A chain of streaming kernels like the step's 256-wide level: kernel k reads tensor k (T bytes, image by image) and writes tensor
// k + 1.  "same": every kernel walks images 0 .. B-1; "serpentine": odd kernels walk B-1 .. 0, so a kernel starts on the bytes its
// producer wrote LAST (distance = bytes moved between a line's store and its load; resident while < ~256 MiB: MI355X_MICROARCH.md).
// Prints us per kernel for T = 134 MB (the 256^2 level at B = 32) and 67 / 34 MB.   hipcc -O3 --offload-arch=gfx950 l3_order_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void stream_copy(const float4* __restrict__ in, float4* __restrict__ out, long per_image, int B, int reverse) {
    // persistent: 2048 workgroups walk the images in order; inside an image, workgroups interleave 4 KiB pieces
    for (int j = 0; j < B; ++j) {
        const int b = reverse ? B - 1 - j : j;
        const float4* s = in + (long)b * per_image;
        float4* d = out + (long)b * per_image;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_image; i += (long)gridDim.x * 256) {
            float4 v = s[i];
            v.x += 1.f;
            d[i] = v;
        }
    }
}

int main() {
    const int B = 32, NT = 6, REPS = 20;
    for (long T : {134217728L, 67108864L, 33554432L}) {
        float4* t[NT + 1];
        for (int k = 0; k <= NT; ++k) { (void)hipMalloc(&t[k], T); (void)hipMemset(t[k], 0, T); }
        const long per_image = T / 16 / B;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e30f, sum = 0.f;
            for (int rep = 0; rep < REPS + 2; ++rep) {
                (void)hipEventRecord(e0, 0);
                for (int k = 0; k < NT; ++k)
                    hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, 0, (const float4*)t[k], t[k + 1], per_image, B, mode == 1 ? (k & 1) : 0);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
            }
            printf("T = %4ld MB, %-10s: %7.1f us per kernel (mean), %7.1f (best)  -> %5.2f TB/s of read + write\n", T >> 20,
                   mode ? "serpentine" : "same", sum / REPS / NT * 1e3, best / NT * 1e3, 2.0 * T / (sum / REPS / NT * 1e-3) / 1e12);
        }
        for (int k = 0; k <= NT; ++k) (void)hipFree(t[k]);
    }
    return 0;
}
