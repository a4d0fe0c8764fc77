#include "loss_dispatch.hpp"
namespace uaps {
template <int D, int C> static int run_unsup_bwd(const LossArgs& a) {
    constexpr int V = unsup_vec<D, C>();
    const long HW = (long)a.H * a.W, N = (long)a.B * HW;
    HeadPtrs<D> z = in_ptrs<D>(a.logits);
    HeadOutPtrs<D> dz = out_ptrs<D>(a.dlogits);
    if (V > 1 && vec_ok(a, V, true)) {
        const long ng = N / V;
        hipLaunchKernelGGL((unsup_bwd_kernel<D, C, V>), dim3(grid_for(ng)), dim3(kThreads), 0, a.stream, z, dz, (int)HW, ng, N, a.labels, a.cscalars, a.cw1, a.cw2, a.gscale);
    } else {
        hipLaunchKernelGGL((unsup_bwd_kernel<D, C, 1>), dim3(grid_for(N)), dim3(kThreads), 0, a.stream, z, dz, (int)HW, N, N, a.labels, a.cscalars, a.cw1, a.cw2, a.gscale);
    }
    return (int)hipGetLastError();
}
int launch_unsup_bwd(const LossArgs& a) { UAPS_DISPATCH_DC(run_unsup_bwd, a) }
}  // namespace uaps
