// The loss block of a training step: one backward launch for both branches (include/uaps_hip.h, "pair loss").
#include "loss_dispatch.hpp"
namespace uaps {
template <int D, int C> static int run_pair_bwd(const PairArgs& a) {
    constexpr int VU = unsup_vec<D, C>();
    const long HW = (long)a.H * a.W, N = (long)a.B * HW;
    HeadPtrs<D> zl = in_ptrs<D>(a.lab), zu = in_ptrs<D>(a.un);
    HeadOutPtrs<D> dl = out_ptrs<D>(a.dlab), du = out_ptrs<D>(a.dun);
    if (pair_vec_ok(a, 4, true)) {
        const int nb_s = pair_grid(N / 4, pair_cap_s(a.cfg)), nb_u = pair_grid(N / VU, pair_cap_u(a.cfg));
        if (a.amax_out)
            UAPS_LAUNCH_MAIN((pair_bwd_kernel<D, C, 4, VU, true>), dim3(nb_s + nb_u), dim3(kThreads), 0, a.stream, zl, zu, dl, du, (int)HW, N, a.Nloss,
                               a.labels, a.cpseudo, a.sscal, a.uscal, a.ce_coef, a.dice_coef, a.cw1, a.cw2, a.gscale, nb_s, (const uint32_t*)uaps_get_step_state(), a.amax_out);
        else
            UAPS_LAUNCH_MAIN((pair_bwd_kernel<D, C, 4, VU>), dim3(nb_s + nb_u), dim3(kThreads), 0, a.stream, zl, zu, dl, du, (int)HW, N, a.Nloss,
                               a.labels, a.cpseudo, a.sscal, a.uscal, a.ce_coef, a.dice_coef, a.cw1, a.cw2, a.gscale, nb_s, (const uint32_t*)uaps_get_step_state(), a.amax_out);
    } else {
        const int nb_s = pair_grid(N, pair_cap_s(a.cfg)), nb_u = pair_grid(N, pair_cap_u(a.cfg));
        if (a.amax_out)
            UAPS_LAUNCH_MAIN((pair_bwd_kernel<D, C, 1, 1, true>), dim3(nb_s + nb_u), dim3(kThreads), 0, a.stream, zl, zu, dl, du, (int)HW, N, a.Nloss,
                               a.labels, a.cpseudo, a.sscal, a.uscal, a.ce_coef, a.dice_coef, a.cw1, a.cw2, a.gscale, nb_s, (const uint32_t*)uaps_get_step_state(), a.amax_out);
        else
            UAPS_LAUNCH_MAIN((pair_bwd_kernel<D, C, 1, 1>), dim3(nb_s + nb_u), dim3(kThreads), 0, a.stream, zl, zu, dl, du, (int)HW, N, a.Nloss,
                               a.labels, a.cpseudo, a.sscal, a.uscal, a.ce_coef, a.dice_coef, a.cw1, a.cw2, a.gscale, nb_s, (const uint32_t*)uaps_get_step_state(), a.amax_out);
    }
    return (int)hipGetLastError();
}
int launch_pair_bwd(const PairArgs& a) { UAPS_DISPATCH_DC(run_pair_bwd, a) }
}  // namespace uaps
