#include "loss_dispatch.hpp"
namespace uaps {
template <int D, int C> static int run_unsup_fwd(const LossArgs& a) {
    constexpr int V = unsup_vec<D, C>();
    const long HW = (long)a.H * a.W, N = (long)a.B * HW;
    HeadPtrs<D> z = in_ptrs<D>(a.logits);
    HeadWeights<D> w;
    for (int k = 0; k < D; ++k) w.w[k] = (float)a.w[k];   // float64 weights act as fp32 scalars (UAPS_train.py:252)
    int nrows;
    if (V > 1 && vec_ok(a, V, false)) {
        const long ng = N / V; nrows = grid_for(ng);
        hipLaunchKernelGGL((unsup_fwd_kernel<D, C, V>), dim3(nrows), dim3(kThreads), 0, a.stream, z, w, (int)HW, ng, N, a.pseudo, a.var, a.partials);
    } else {
        nrows = grid_for(N);
        hipLaunchKernelGGL((unsup_fwd_kernel<D, C, 1>), dim3(nrows), dim3(kThreads), 0, a.stream, z, w, (int)HW, N, N, a.pseudo, a.var, a.partials);
    }
    hipLaunchKernelGGL((finalize_kernel<true>), dim3(1), dim3(kFinalizeThreads), 0, a.stream, a.partials, nrows, D, C, N, a.cw1, a.cw2, a.eps, a.scalars);
    return (int)hipGetLastError();
}
int launch_unsup_fwd(const LossArgs& a) { UAPS_DISPATCH_DC(run_unsup_fwd, a) }
}  // namespace uaps
