// Calibration of rocprofv3's FETCH_SIZE on gfx950 against known byte counts (guide: "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Every kernel reads N distinct bytes of a 1 GiB buffer exactly once
// (nothing cache-resident), in the access shapes this library uses:
//   w16 / w8 / w4      contiguous 16 / 8 / 4 bytes per lane (row kernels, loss, BatchNorm: b128; conv_h32's half-units: b64)
//   piece160           row pieces of 160 bytes that start 16 bytes in front of a 128-byte line, 1 KiB apart (the 8 x 32-tile kernels'
//                      haloed rows: one full line and two 16-byte ends per piece), as 8-byte lanes like conv_s32_body stages them
// Run each under `rocprofv3 --pmc FETCH_SIZE`, `--pmc TCC_EA0_RDREQ_sum` and `--pmc TCC_EA0_RDREQ_32B_sum` (three passes) and
// compare with the bytes printed here: tools/diag/fetch_calib.sh.      hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <typename T>
__global__ __launch_bounds__(256) void calib_contig(const T* __restrict__ in, long n, float* __restrict__ sink) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const T v = in[i];
        acc += reinterpret_cast<const float*>(&v)[0];
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// pieces of 160 bytes: piece p starts at byte p * 1024 + 112 (16 bytes in front of the line at + 128); lane = 8 bytes, 20 lanes per piece
__global__ __launch_bounds__(256) void calib_piece160(const char* __restrict__ in, long npieces, float* __restrict__ sink) {
    float acc = 0.f;
    const int lane20 = threadIdx.x % 20, sub = threadIdx.x / 20;     // 12 pieces per workgroup pass (240 of 256 threads)
    for (long p = (long)blockIdx.x * 12 + sub; p < npieces && sub < 12; p += (long)gridDim.x * 12) {
        const float2 v = *reinterpret_cast<const float2*>(in + p * 1024 + 112 + lane20 * 8);
        acc += v.x;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const long bytes = 1L << 30;
    char* buf; float* sink;
    (void)hipMalloc(&buf, bytes); (void)hipMalloc(&sink, 4);
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_contig<float4>, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(calib_contig<float2>, dim3(4096), dim3(256), 0, 0, (const float2*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(calib_contig<float>, dim3(4096), dim3(256), 0, 0, (const float*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(calib_piece160, dim3(4096), dim3(256), 0, 0, (const char*)buf, bytes / 1024 - 1, sink);
    }
    (void)hipDeviceSynchronize();
    printf("known bytes per launch: contig<float4|float2|float> %ld each; piece160 used %ld (160 B of every KiB), lines touched %ld (3 x 128 B per piece)\n",
           bytes, (bytes / 1024 - 1) * 160, (bytes / 1024 - 1) * 384);
    return 0;
}
