// The loss block of a training step as one forward launch + one finalize (include/uaps_hip.h, "pair loss").
#include "loss_dispatch.hpp"
namespace uaps {
template <int D, int C, int VS, int VU, bool PFS, bool PFU, int MINW>
static void launch_variant(const PairArgs& a, long N, int HW, const HeadPtrs<D>& zl, const HeadPtrs<D>& zu, const HeadWeights<D>& w,
                           float* part_s, float* part_u, int& nb_s, int& nb_u) {
    nb_s = pair_grid(N / VS, pair_cap_s(a.cfg)); nb_u = pair_grid(N / VU, pair_cap_u(a.cfg));
    UAPS_LAUNCH_MAIN((pair_fwd_kernel<D, C, VS, VU, PFS, PFU, MINW>), dim3(nb_s + nb_u), dim3(kThreads), 0, a.stream, zl, zu, w, HW, N, a.labels,
                       a.pseudo, a.var, part_s, part_u, nb_s, a.w ? (const uint32_t*)nullptr : (const uint32_t*)uaps_get_step_state());
}
template <int D, int C, int VS, int VU, int MINW>
static void launch_both(const PairArgs& a, long N, int HW, const HeadPtrs<D>& zl, const HeadPtrs<D>& zu, const HeadWeights<D>& w,
                        float* part_s, float* part_u, int& nb_s, int& nb_u) {
    const int cap = (a.cfg & 0xfff) ? (a.cfg & 0xfff) : 512;
    nb_s = nb_u = pair_grid(N / (VS > VU ? VS : VU), cap);
    UAPS_LAUNCH_MAIN((pair_fwd_both_kernel<D, C, VS, VU, MINW>), dim3(nb_s), dim3(kThreads), 0, a.stream, zl, zu, w, HW, N, a.labels,
                       a.pseudo, a.var, part_s, part_u, a.w ? (const uint32_t*)nullptr : (const uint32_t*)uaps_get_step_state());
}
template <int D, int C> static int run_pair_fwd(const PairArgs& a) {
    constexpr int VU = unsup_vec<D, C>();
    const long HW = (long)a.H * a.W, N = (long)a.B * HW;
    HeadPtrs<D> zl = in_ptrs<D>(a.lab), zu = in_ptrs<D>(a.un);
    HeadWeights<D> w;
    for (int k = 0; k < D; ++k) w.w[k] = a.w ? (float)a.w[k] : 0.f;   // float64 weights act as fp32 scalars (UAPS_train.py:252); NULL: step state
    float* part_s = a.partials;
    float* part_u = a.partials + (size_t)kMaxBlocks * sup_nsums(D, C);
    int nb_s, nb_u;
    if (pair_vec_ok(a, 4, false)) {
#ifdef UAPS_LOSS_EXPERIMENT      // tools/bench_loss.py: kernel variants selected by cfg bits 28-30 (built for D*C <= 16 only)
        if constexpr (D * C <= 16 && D >= 4) {
            switch ((a.cfg >> 28) & 7) {
                case 1: launch_both<D, C, 4, 4, 2>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u); break;
                case 2: launch_variant<D, C, 4, 1, false, false, 4>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u); break;
                case 3: launch_variant<D, C, 4, 2, false, false, 3>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u); break;
                case 4: launch_variant<D, C, 4, 4, true, false, 2>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u); break;
                case 5: launch_variant<D, C, 4, 2, true, true, 2>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u); break;
                case 6: launch_variant<D, C, 2, 1, false, false, 4>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u); break;
                case 7: launch_variant<D, C, 4, 2, false, false, 2>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u); break;
                default: launch_variant<D, C, 4, VU, false, false, 2>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u); break;
            }
        } else
#endif
        launch_variant<D, C, 4, VU, false, false, 2>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u);
    } else {
        launch_variant<D, C, 1, 1, false, false, 2>(a, N, (int)HW, zl, zu, w, part_s, part_u, nb_s, nb_u);
    }
    hipLaunchKernelGGL(pair_finalize_kernel, dim3(1), dim3(kFinalizeThreads), 0, a.stream, part_s, nb_s, part_u, nb_u, D, C, N, a.ce_coef,
                       a.dice_coef, a.cw1, a.cw2, a.eps, a.sscal, a.uscal, a.sums, (const uint32_t*)uaps_get_step_state());
    return (int)hipGetLastError();
}
int launch_pair_fwd(const PairArgs& a) { UAPS_DISPATCH_DC(run_pair_fwd, a) }
}  // namespace uaps
