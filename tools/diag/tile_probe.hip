// Memory floor of the 3x3 convolution's access pattern on a [B, C, 256, 256] fp32 tensor, as a function of the tile shape:
// a persistent workgroup walks tiles of TH x TW pixels (TH * TW = 256) down a column strip, reads the haloed tile of CIN channels
// ((TH + 2) rows x (TW + 8) floats per channel, 16 bytes per lane as conv_hp16_body stages it) and writes a COUT-channel tile
// (16 bytes per lane, rows of TW floats).  No arithmetic: what is measured is how fast HBM serves row pieces of (TW + 8) * 4 bytes
// 1 KiB apart, against the same bytes as full rows.      hipcc -O3 --offload-arch=gfx950 tile_probe.hip -o tile_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

template <int TH, int TW, int CIN, int COUT, bool WRITE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, float* __restrict__ sink) {
    constexpr int IH = TH + 2, UPR = (TW + 8) / 4, UPC = IH * UPR, NU = (CIN * UPC + 255) / 256;
    constexpr int OUPR = TW / 4, OUPC = TH * OUPR, NO = (COUT * OUPC + 255) / 256;
    const int tiles_x = W / TW, tiles_y = H / TH, tpi = tiles_x * tiles_y;
    const long ntiles = (long)B * tpi;
    const long t0 = ntiles * blockIdx.x / gridDim.x, t1 = ntiles * (blockIdx.x + 1) / gridDim.x;
    const long HW = (long)H * W;
    float acc = 0.f;
    float4 v[NU];
    for (long t = t0; t < t1; ++t) {
        const int b = (int)(t / tpi), tt = (int)(t % tpi), tx = tt / tiles_y, ty = tt % tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
        for (int n = 0; n < NU; ++n) {
            const int u = threadIdx.x + n * 256, c = u / UPC, r = (u % UPC) / UPR, cu = u % UPR;
            const int gy = y0 - 1 + r, gx = x0 - 4 + cu * 4;
            const bool ok = c < CIN && gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[n] = ok ? *reinterpret_cast<const float4*>(in + ((long)b * CIN + c) * HW + (long)gy * W + gx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int n = 0; n < NU; ++n) acc += v[n].x + v[n].w;
        if (WRITE) {
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const int u = threadIdx.x + n * 256, c = u / OUPC, r = (u % OUPC) / OUPR, cu = u % OUPR;
                if (c < COUT) *reinterpret_cast<float4*>(out + ((long)b * COUT + c) * HW + (long)(y0 + r) * W + x0 + cu * 4) = v[n % NU];
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// full-width rows with a ring: a workgroup owns ROWS consecutive rows of one image; per step it reads TWO new rows of every input
// channel (a wave-instruction = one 1-KiB row of one channel) and writes two rows of every output channel; 2 rows of warm-up per run
template <int CIN, int COUT, int ROWS>
__global__ __launch_bounds__(256) void ring_probe(const float* __restrict__ in, float* __restrict__ out, int B, float* __restrict__ sink) {
    constexpr int H = 256, W = 256, NL = CIN * 2 / 4;      // loads per thread and step: (2 rows x CIN channels) / 4 waves
    const long HW = (long)H * W;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int runs = B * (H / ROWS);
    float acc = 0.f;
    for (int run = blockIdx.x; run < runs; run += gridDim.x) {
        const int b = run / (H / ROWS), r0 = (run % (H / ROWS)) * ROWS;
        for (int y = r0 - 2; y < r0 + ROWS; y += 2) {      // stages rows y + 1, y + 2
            float4 v[NL];
#pragma unroll
            for (int n = 0; n < NL; ++n) {
                const int idx = n * 4 + wave, c = idx % CIN, rr = idx / CIN, gy = y + 1 + rr;
                v[n] = (gy >= 0 && gy < H) ? *reinterpret_cast<const float4*>(in + ((long)b * CIN + c) * HW + (long)gy * W + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int n = 0; n < NL; ++n) acc += v[n].x + v[n].w;
            if (y >= r0) {
#pragma unroll
                for (int n = 0; n < COUT * 2 / 4; ++n) {
                    const int idx = n * 4 + wave, c = idx % COUT, rr = idx / COUT;
                    *reinterpret_cast<float4*>(out + ((long)b * COUT + c) * HW + (long)(y + rr) * W + lane * 4) = v[n % NL];
                }
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int CIN, int COUT, int ROWS>
void run_ring(const float* in, float* out, int B, float* sink, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((ring_probe<CIN, COUT, ROWS>), dim3(blocks), dim3(256), 0, 0, in, out, B, sink);
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ring_probe<CIN, COUT, ROWS>), dim3(blocks), dim3(256), 0, 0, in, out, B, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double bytes = (double)B * 65536 * 4 * (CIN + COUT);
    printf("ring: full rows, %2d rows per run       cin %2d cout %2d  blocks %4d: %8.1f us  %7.1f GB/s (algorithmic bytes)\n", ROWS, CIN, COUT, blocks, ms * 1e3, bytes / ms / 1e6);
}

template <int TH, int TW, int CIN, int COUT, bool WRITE>
void run(const char* name, const float* in, float* out, int B, float* sink, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<TH, TW, CIN, COUT, WRITE>), dim3(blocks), dim3(256), 0, 0, in, out, B, 256, 256, sink);
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<TH, TW, CIN, COUT, WRITE>), dim3(blocks), dim3(256), 0, 0, in, out, B, 256, 256, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double bytes = (double)B * 65536 * 4 * (CIN + (WRITE ? COUT : 0));
    printf("%-34s tile %3d x %3d  cin %2d cout %2d  blocks %4d: %8.1f us  %7.1f GB/s (algorithmic bytes)\n", name, TH, TW, CIN, WRITE ? COUT : 0, blocks, ms * 1e3, bytes / ms / 1e6);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128;
    const size_t n_in = (size_t)B * 32 * 65536, n_out = (size_t)B * 16 * 65536;
    float *in, *out, *sink;
    hipMalloc(&in, n_in * 4); hipMalloc(&out, n_out * 4); hipMalloc(&sink, 4);
    hipMemset(in, 0, n_in * 4); hipMemset(out, 0, n_out * 4);
    for (int blocks : {256, 512, 768, 1024}) {
        run<8, 32, 16, 16, false>("read only", in, out, B, sink, blocks);
        run<4, 64, 16, 16, false>("read only", in, out, B, sink, blocks);
        run<2, 128, 16, 16, false>("read only", in, out, B, sink, blocks);
        run<1, 256, 16, 16, false>("read only", in, out, B, sink, blocks);
        run<8, 32, 16, 16, true>("read + write", in, out, B, sink, blocks);
        run<4, 64, 16, 16, true>("read + write", in, out, B, sink, blocks);
        run<2, 128, 16, 16, true>("read + write", in, out, B, sink, blocks);
        run<1, 256, 16, 16, true>("read + write", in, out, B, sink, blocks);
        run_ring<16, 16, 16>(in, out, B, sink, blocks);
        run_ring<16, 16, 32>(in, out, B, sink, blocks);
        run_ring<32, 16, 16>(in, out, B, sink, blocks);
        run_ring<32, 16, 32>(in, out, B, sink, blocks);
        run<8, 32, 32, 16, true>("read + write", in, out, B, sink, blocks);
        run<2, 128, 32, 16, true>("read + write", in, out, B, sink, blocks);
        run<1, 256, 32, 16, true>("read + write", in, out, B, sink, blocks);
    }
    return 0;
}
