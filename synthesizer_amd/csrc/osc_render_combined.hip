// osc_render_combined.hip -- every instantiation of k_render_combined (osc_render_kernels.hpp): lean and general lists, or the voice
// table directly, in one kernel -- single-group launches (small banks write the caller's bus themselves), banks without lean
// candidates, SYNTHHIP_NO_SPLIT.  Eighteen kernels that each carry the general code: a translation unit of its own, compiled beside
// osc_render.hip and osc_render_lean.hip (see there).
#include "osc_host.hpp"
#include "osc_render_kernels.hpp"

namespace shosc {

int launch_render_combined(int var, int mode, dim3 grid, hipStream_t st, const LaunchArgs& A, const NextArgs& N, const FoldIn& F, double2* parts, const BusOut& out) {
#define SH_LAUNCH_SHAPE(W_, F_, M_)                                                                                                      \
    do {                                                                                                                                 \
        if (mode == COMBINED_LEAN_HARM) hipLaunchKernelGGL((k_render_combined<W_, F_, M_, COMBINED_LEAN_HARM>), grid, dim3(W_ * 64), 0, st, A, N, F, parts, out); \
        else if (mode == COMBINED_LEAN_ALL) hipLaunchKernelGGL((k_render_combined<W_, F_, M_, COMBINED_LEAN_ALL>), grid, dim3(W_ * 64), 0, st, A, N, F, parts, out); \
        else hipLaunchKernelGGL((k_render_combined<W_, F_, M_, COMBINED_DIRECT>), grid, dim3(W_ * 64), 0, st, A, N, F, parts, out); \
    } while (0)
    switch (var) {
    case 4163: return sh::set_error(SH_ERR_INVALID, "sh_bank_render: shape 4163 needs a split launch of a Harmonics or an FM Sine bank");
    case 484: SH_LAUNCH_SHAPE(4, 8, 4); break;
    case 444: SH_LAUNCH_SHAPE(4, 4, 4); break;
    case 844: SH_LAUNCH_SHAPE(8, 4, 4); break;
    case 821: SH_LAUNCH_SHAPE(8, 2, 1); break;
    case 421: SH_LAUNCH_SHAPE(4, 2, 1); break;
    case 211: SH_LAUNCH_SHAPE(2, 1, 1); break;
    default: return sh::set_error(SH_ERR_INVALID, "sh_bank_render: SYNTHHIP_VARIANT %d is not one of 4163, 484, 444, 844, 821, 421, 211", var);
    }
#undef SH_LAUNCH_SHAPE
    SH_CHECK_LAUNCH("k_render_combined");
    return SH_OK;
}

}  // namespace shosc
