// tu_sweepr.hip -- the instantiations of sweep_row_kernel (k_sweep_r.h: SCD sweep of the fp32-operand mode, row form), see tu_sweepq.h.
#include "tu_sweepq.h"
#include "k_sweep_r.h"

template <int CPL, int KR> static void launch_r(const SweepArgs &a, int nb, hipStream_t st)
{
    if (a.mask) sweep_row_kernel<CPL, true, KR, CPL><<<nb, 64 * SWEEPR_NW, 0, st>>>(a);
    else sweep_row_kernel<CPL, false, KR, CPL><<<nb, 64 * SWEEPR_NW, 0, st>>>(a);
}
// nb workgroups of 32 columns (eight wavefronts of four); a.k <= 50, the caller's rank padding KP = 16 ceil(k / 16)
void nnlm_tu_sweep_r(const SweepArgs &a, int nb, hipStream_t st)
{
    if (a.k <= 16) launch_r<1, 16>(a, nb, st);
    else if (a.k <= 32) launch_r<2, 32>(a, nb, st);
    else if (a.k <= 48) launch_r<3, 48>(a, nb, st);
    else launch_r<4, 50>(a, nb, st);
}
