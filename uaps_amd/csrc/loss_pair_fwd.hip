// The loss block of a training step as one forward launch + one finalize (include/uaps_hip.h, "pair loss").
#include "loss_dispatch.hpp"
namespace uaps {
template <int D, int C> static int run_pair_fwd(const PairArgs& a) {
    constexpr int VU = unsup_vec<D, C>();
    const long HW = (long)a.H * a.W, N = (long)a.B * HW;
    HeadPtrs<D> zl = in_ptrs<D>(a.lab), zu = in_ptrs<D>(a.un);
    HeadWeights<D> w;
    for (int k = 0; k < D; ++k) w.w[k] = (float)a.w[k];   // float64 weights act as fp32 scalars (UAPS_train.py:252)
    float* part_s = a.partials;
    float* part_u = a.partials + (size_t)kMaxBlocks * sup_nsums(D, C);
    int nb_s, nb_u;
    if (pair_vec_ok(a, 4, false)) {
        nb_s = pair_grid(N / 4, a.cfg); nb_u = pair_grid(N / VU, a.cfg);
        hipLaunchKernelGGL((pair_fwd_kernel<D, C, 4, VU>), dim3(nb_s + nb_u), dim3(kThreads), 0, a.stream, zl, zu, w, (int)HW, N, a.labels,
                           a.pseudo, a.var, part_s, part_u, nb_s);
    } else {
        nb_s = nb_u = pair_grid(N, a.cfg);
        hipLaunchKernelGGL((pair_fwd_kernel<D, C, 1, 1>), dim3(nb_s + nb_u), dim3(kThreads), 0, a.stream, zl, zu, w, (int)HW, N, a.labels,
                           a.pseudo, a.var, part_s, part_u, nb_s);
    }
    hipLaunchKernelGGL(pair_finalize_kernel, dim3(1), dim3(kFinalizeThreads), 0, a.stream, part_s, nb_s, part_u, nb_u, D, C, N, a.ce_coef,
                       a.dice_coef, a.cw1, a.cw2, a.eps, a.sscal, a.uscal, a.sums);
    return (int)hipGetLastError();
}
int launch_pair_fwd(const PairArgs& a) { UAPS_DISPATCH_DC(run_pair_fwd, a) }
}  // namespace uaps
