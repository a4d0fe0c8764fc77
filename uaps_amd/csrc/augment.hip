// On-device input pipeline for gfx950: the per-sample work of the reference's
// training loader, utilities/dataloaders.py:60-119 --
//   A.Resize(256, 256, INTER_NEAREST), HorizontalFlip, VerticalFlip, RandomBrightnessContrast((0,0.5),(0,0.5)),
//   Blur (box 3/5/7), RandomRotate90, GaussNoise, then T.ToTensor() + T.Normalize(mean, std); the mask follows the
//   geometric steps only (nearest) and becomes int64 --
// as ONE gather kernel over a decoded uint8 batch resident in HBM: a thread owns one output pixel, walks the
// blur window in the resized frame (box filters commute with flips / quarter turns, so the geometry is a single
// coordinate map), applies the brightness/contrast table arithmetic to each tap, rounds as the uint8 pipeline of
// the loader does between its stages, adds the noise, normalises and writes the three channel planes.
// Which augmentations fire and their magnitudes are per-image parameters drawn on the host (uaps_amd/augment.py),
// because the reference's draws come from albumentations' Python RNG streams and cannot be reproduced.
// PARITY UNPINNED: cv2 and albumentations are not in the build image; the arithmetic below restates their
// documented uint8 behaviour (oracle/augment_oracle.py is the same restatement in numpy).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uaps_hip.h"
#include "philox.hpp"

namespace {
using uaps::U4; using uaps::philox4x32_10; using uaps::u01; using uaps::pick;
constexpr int kThreads = 256;

// cv2 BORDER_REFLECT_101: ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// params_i [B][8]: hflip, vflip, rot_k (np.rot90 count, 0..3), blur_k (0 = off, else 3 / 5 / 7), noise_on, 3 x reserved
// params_f [B][4]: alpha (contrast factor), beta (brightness shift as a fraction of 255), sigma (noise std in grey levels), reserved
__global__ __launch_bounds__(kThreads) void augment_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                           const int* __restrict__ pi, const float* __restrict__ pf,
                                                           const float* __restrict__ noise, uint64_t seed, int B, int Hs, int Ws,
                                                           int Ho, int Wo, float m0, float m1, float m2, float s0, float s1,
                                                           float s2, float* __restrict__ out, int64_t* __restrict__ out_mask) {
    const long n = (long)blockIdx.x * kThreads + threadIdx.x;
    const long per = (long)Ho * Wo;
    if (n >= (long)B * per) return;
    const int b = (int)(n / per);
    const int y = (int)((n - (long)b * per) / Wo), x = (int)(n % Wo);
    const int hflip = pi[b * 8 + 0], vflip = pi[b * 8 + 1], rot = pi[b * 8 + 2] & 3, bk = pi[b * 8 + 3], noise_on = pi[b * 8 + 4];
    const float alpha = pf[b * 4 + 0], beta = pf[b * 4 + 1], sigma = pf[b * 4 + 2];
    // out = np.rot90(img2, rot): the pixel of img2 (square Ho == Wo when rot is odd) that lands on (y, x)
    int ry, rx;
    if (rot == 0) { ry = y; rx = x; }
    else if (rot == 1) { ry = x; rx = Wo - 1 - y; }
    else if (rot == 2) { ry = Ho - 1 - y; rx = Wo - 1 - x; }
    else { ry = Ho - 1 - x; rx = y; }
    // img2 = vflip(hflip(resized)); resized[yy][xx] = src[floor(yy * Hs / Ho)][floor(xx * Ws / Wo)] (INTER_NEAREST)
    const uint8_t* src = img + (long)b * Hs * Ws * 3;
    const int r = bk / 2;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int dy = -r; dy <= r; ++dy) {
        int yy = reflect101(ry + dy, Ho);
        if (vflip) yy = Ho - 1 - yy;
        const int sy = min((int)(((long)yy * Hs) / Ho), Hs - 1);
        for (int dx = -r; dx <= r; ++dx) {
            int xx = reflect101(rx + dx, Wo);
            if (hflip) xx = Wo - 1 - xx;
            const int sx = min((int)(((long)xx * Ws) / Wo), Ws - 1);
            const uint8_t* p = src + ((long)sy * Ws + sx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // RandomBrightnessContrast on uint8: lut[v] = uint8(clip(v * alpha + beta * 255, 0, 255))  (truncation)
                float v = (float)p[c] * alpha + beta * 255.f;
                v = fminf(fmaxf(v, 0.f), 255.f);
                acc[c] += (float)(int)v;
            }
        }
    }
    const float inv = 1.f / (float)((2 * r + 1) * (2 * r + 1));
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    U4 rnd;
    if (noise_on && !noise) rnd = philox4x32_10((uint64_t)n, seed);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = bk ? rintf(acc[c] * inv) : acc[c];          // cv2.blur on uint8 rounds half to even
        if (noise_on) {
            float g;
            if (noise) {
                g = noise[((long)b * 3 + c) * per + (long)y * Wo + x];
            } else {                                          // Box-Muller on two Philox words per channel pair
                const float u1 = fmaxf(u01(pick(rnd, c & 1 ? 2 : 0)), 1e-7f), u2 = u01(pick(rnd, c & 1 ? 3 : 1));
                const float rad = sqrtf(-2.f * __logf(u1));
                g = (c == 2 ? rad * __sinf(6.2831853f * u2) : rad * __cosf(6.2831853f * u2)) * sigma;
            }
            v = (float)(int)fminf(fmaxf(v + g, 0.f), 255.f);  // float32 add, clip, astype(uint8) truncates
        }
        out[((long)b * 3 + c) * per + (long)y * Wo + x] = (v / 255.f - mean[c]) / stdv[c];
    }
    if (mask && out_mask) {
        int yy = vflip ? Ho - 1 - ry : ry, xx = hflip ? Wo - 1 - rx : rx;
        const int sy = min((int)(((long)yy * Hs) / Ho), Hs - 1), sx = min((int)(((long)xx * Ws) / Wo), Ws - 1);
        out_mask[n] = (int64_t)mask[((long)b * Hs + sy) * Ws + sx];
    }
}
}  // namespace

extern "C" int uaps_augment_batch(const uint8_t* images_hwc, const uint8_t* masks, const int* params_i, const float* params_f,
                                  const float* noise, uint64_t seed, int B, int Hs, int Ws, int Ho, int Wo, const float* mean3_host,
                                  const float* std3_host, float* out, int64_t* out_mask, uaps_stream_t stream) {
    if (!images_hwc || !params_i || !params_f || !out || !mean3_host || !std3_host) return UAPS_EINVAL;
    if (B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || (masks != nullptr) != (out_mask != nullptr)) return UAPS_EINVAL;
    const long total = (long)B * Ho * Wo;
    const long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 0x7fffffffL) return UAPS_ERANGE;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, images_hwc, masks, params_i, params_f,
                       noise, seed, B, Hs, Ws, Ho, Wo, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1],
                       std3_host[2], out, out_mask);
    return (int)hipGetLastError();
}
