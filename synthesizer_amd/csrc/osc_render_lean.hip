// osc_render_lean.hip -- every instantiation of k_render_lean (osc_render_kernels.hpp), the lean lists of a split launch, and the one
// function that launches them.  A translation unit of its own so that the build compiles it beside osc_render.hip: the lean kernels
// are half of the render path's ISA, and one hipcc process per unit is the build's parallelism (no relocatable device code: a kernel
// lives in the unit that launches it).
#include "osc_host.hpp"
#include "osc_render_kernels.hpp"

namespace shosc {

int launch_render_lean(int var, int kinds, bool seg, dim3 grid, hipStream_t st, const LaunchArgs& A, const NextArgs& N, const FoldIn& F, double2* parts) {
#define SH_LEAN(W_, F_, M_, K_, S_) hipLaunchKernelGGL((k_render_lean<W_, F_, M_, K_, S_>), grid, dim3(W_ * 64), 0, st, A, N, F, parts)
#define SH_LEAN_KINDS(W_, F_, M_)                                    \
    do {                                                            \
        if (kinds == LEAN_K_HARM) SH_LEAN(W_, F_, M_, LEAN_K_HARM, false);  \
        else if (kinds == LEAN_K_FM) SH_LEAN(W_, F_, M_, LEAN_K_FM, false); \
        else SH_LEAN(W_, F_, M_, LEAN_K_ALL, false);                 \
    } while (0)
    if (seg) {                                           // a transition launch cut into segments: one shape, polynomial Harmonics or anything
        if (var != 484 || kinds == LEAN_K_FM) return sh::set_error(SH_ERR_INVALID, "launch_render_lean: segmented launches take shape 484");
        if (kinds == LEAN_K_HARM) SH_LEAN(4, 8, 4, LEAN_K_HARM, true); else SH_LEAN(4, 8, 4, LEAN_K_ALL, true);
    } else {
        switch (var) {
        case 4163:                                       // sixteen frames per lane: polynomial-Harmonics banks and FM Sine banks (bank_render)
            if (kinds == LEAN_K_HARM) SH_LEAN(4, 16, 3, LEAN_K_HARM, false);
            else if (kinds == LEAN_K_FM) SH_LEAN(4, 16, 3, LEAN_K_FM, false);
            else return sh::set_error(SH_ERR_INVALID, "sh_bank_render: shape 4163 needs a split launch of a Harmonics or an FM Sine bank");
            break;
        case 484: SH_LEAN_KINDS(4, 8, 4); break;
        case 444: SH_LEAN_KINDS(4, 4, 4); break;
        case 844: SH_LEAN_KINDS(8, 4, 4); break;
        case 821: SH_LEAN_KINDS(8, 2, 1); break;
        default: return sh::set_error(SH_ERR_INVALID, "sh_bank_render: no lean kernel of shape %d", var);
        }
    }
#undef SH_LEAN_KINDS
#undef SH_LEAN
    SH_CHECK_LAUNCH("k_render_lean");
    return SH_OK;
}

}  // namespace shosc
