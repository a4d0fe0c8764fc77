// Multi-tensor Adam step for gfx950: torch.optim.Adam(lr, betas, eps, weight_decay) as the reference uses it
// (UAPS_train.py:112, 285-292: Adam(model.parameters(), lr=base_lr), default betas / eps, no weight decay, no amsgrad)
// over all 208 parameter tensors of the U-Net in ceil(208 / 48) launches: blockIdx.y selects the tensor, a lane owns four
// elements of param / grad / exp_avg / exp_avg_sq (16-byte streams: 7 x 14.9 MB per step).
//   m = m + (g - m) (1 - b1)        v = b2 v + (1 - b2) g g       p = p - (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// with bc1 = 1 - b1^t, bc2 = 1 - b2^t: the operation order of torch's single-tensor Adam.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/uaps_hip.h"
#include "philox.hpp"
#include "hints.hpp"

namespace {
constexpr int kThreads = 256;
constexpr int kAdamBatch = 48;
struct AdamBatch {
    float* p[kAdamBatch]; const float* g[kAdamBatch]; float* m[kAdamBatch]; float* v[kAdamBatch];
    long n[kAdamBatch];
};
__global__ __launch_bounds__(kThreads) void adam_kernel(AdamBatch t, float one_minus_b1, float b2, float one_minus_b2, float step_size,
                                                        float inv_sqrt_bc2, float eps, float weight_decay, const uint32_t* __restrict__ st) {
    step_size = uaps::step_f(st, uaps::kStepAdam, step_size); inv_sqrt_bc2 = uaps::step_f(st, uaps::kStepAdam + 1, inv_sqrt_bc2);
    const int k = blockIdx.y;
    float* __restrict__ p = t.p[k]; const float* __restrict__ g = t.g[k]; float* __restrict__ m = t.m[k]; float* __restrict__ v = t.v[k];
    const long n = t.n[k];
    const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                      reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const long n4 = al ? n / 4 : 0;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kThreads) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float P[4] = {pp.x, pp.y, pp.z, pp.w}, M[4] = {mm.x, mm.y, mm.z, mm.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
        const float G0[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float G = G0[q] + weight_decay * P[q];
            M[q] = M[q] + (G - M[q]) * one_minus_b1;
            V[q] = V[q] * b2 + one_minus_b2 * G * G;
            P[q] = P[q] - step_size * (M[q] / (sqrtf(V[q]) * inv_sqrt_bc2 + eps));
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(P[0], P[1], P[2], P[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(M[0], M[1], M[2], M[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(V[0], V[1], V[2], V[3]);
    }
    if (blockIdx.x == 0)
        for (long i = 4 * n4 + threadIdx.x; i < n; i += kThreads) {
            const float G = g[i] + weight_decay * p[i];
            const float M = m[i] + (G - m[i]) * one_minus_b1, V = v[i] * b2 + one_minus_b2 * G * G;
            m[i] = M; v[i] = V;
            p[i] = p[i] - step_size * (M / (sqrtf(V) * inv_sqrt_bc2 + eps));
        }
}
}  // namespace

extern "C" int uaps_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                              const long* numel, int n, double lr, double beta1, double beta2, double eps, double weight_decay, long step,
                              uaps_stream_t stream) {
    // step < 1 with a step state set (uaps_set_step_state): lr / bias-correction1 and 1 / sqrt(bias-correction2) are read from it
    const void* st = uaps_get_step_state();
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || n <= 0 || (step < 1 && !st)) return UAPS_EINVAL;
    const double bc1 = 1.0 - pow(beta1, (double)(step < 1 ? 1 : step)), bc2 = 1.0 - pow(beta2, (double)(step < 1 ? 1 : step));
    const float step_size = step < 1 ? NAN : (float)(lr / bc1), inv_sqrt_bc2 = step < 1 ? NAN : (float)(1.0 / sqrt(bc2));
    for (int base = 0; base < n; base += kAdamBatch) {
        const int m = n - base < kAdamBatch ? n - base : kAdamBatch;
        AdamBatch t{};
        long most = 0;
        for (int i = 0; i < m; ++i) {
            const int k = base + i;
            if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || numel[k] <= 0) return UAPS_EINVAL;
            t.p[i] = params[k]; t.g[i] = grads[k]; t.m[i] = exp_avg[k]; t.v[i] = exp_avg_sq[k]; t.n[i] = numel[k];
            uaps::account_bytes(28.0 * numel[k]);          // p, g, m, v read; p, m, v written
            if (numel[k] > most) most = numel[k];
        }
        long bx = (most / 4 + kThreads - 1) / kThreads;
        if (bx < 1) bx = 1;
        if (bx > 256) bx = 256;
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)bx, m), dim3(kThreads), 0, (hipStream_t)stream, t, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), step_size, inv_sqrt_bc2, (float)eps, (float)weight_decay, (const uint32_t*)(step < 1 ? st : nullptr));
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return UAPS_OK;
}
