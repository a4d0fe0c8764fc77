#include "loss_dispatch.hpp"
namespace uaps {
template <int D, int C> static int run_sup_fwd(const LossArgs& a) {
    constexpr int V = 4;
    const long HW = (long)a.H * a.W, N = (long)a.B * HW;
    HeadPtrs<D> z = in_ptrs<D>(a.logits);
    int nrows;
    if (vec_ok(a, V, false)) {
        const long ng = N / V; nrows = grid_for(ng);
        hipLaunchKernelGGL((sup_fwd_kernel<D, C, V>), dim3(nrows), dim3(kThreads), 0, a.stream, z, (int)HW, ng, a.labels, a.partials);
    } else {
        nrows = grid_for(N);
        hipLaunchKernelGGL((sup_fwd_kernel<D, C, 1>), dim3(nrows), dim3(kThreads), 0, a.stream, z, (int)HW, N, a.labels, a.partials);
    }
    hipLaunchKernelGGL((finalize_kernel<false>), dim3(1), dim3(kFinalizeThreads), 0, a.stream, a.partials, nrows, D, C, N, a.cw1, a.cw2, a.eps, a.scalars);
    return (int)hipGetLastError();
}
template <int D, int C> static int run_sup_bwd(const LossArgs& a) {
    constexpr int V = 4;
    const long HW = (long)a.H * a.W, N = (long)a.B * HW;
    HeadPtrs<D> z = in_ptrs<D>(a.logits);
    HeadOutPtrs<D> dz = out_ptrs<D>(a.dlogits);
    if (vec_ok(a, V, true)) {
        const long ng = N / V;
        hipLaunchKernelGGL((sup_bwd_kernel<D, C, V>), dim3(grid_for(ng)), dim3(kThreads), 0, a.stream, z, dz, (int)HW, ng, N, a.labels, a.cscalars, a.cw1, a.cw2, a.gscale);
    } else {
        hipLaunchKernelGGL((sup_bwd_kernel<D, C, 1>), dim3(grid_for(N)), dim3(kThreads), 0, a.stream, z, dz, (int)HW, N, N, a.labels, a.cscalars, a.cw1, a.cw2, a.gscale);
    }
    return (int)hipGetLastError();
}
int launch_sup_fwd(const LossArgs& a) { UAPS_DISPATCH_DC(run_sup_fwd, a) }
int launch_sup_bwd(const LossArgs& a) { UAPS_DISPATCH_DC(run_sup_bwd, a) }
}  // namespace uaps
