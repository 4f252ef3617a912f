// tu_sweepr.hip -- the instantiations of sweep_row_kernel (k_sweep_r.h: SCD sweep of the fp32-operand mode, row form), see tu_sweepq.h.
#include "tu_sweepq.h"
#include "k_sweep_r.h"

template <int CPL, int KR, int NW> static void launch_r(const SweepArgs &a, int nb, hipStream_t st)
{
    if (a.mask) sweep_row_kernel<CPL, true, KR, CPL, NW><<<nb, 64 * NW, 0, st>>>(a);
    else sweep_row_kernel<CPL, false, KR, CPL, NW><<<nb, 64 * NW, 0, st>>>(a);
}
template <int NW> static void launch_k(const SweepArgs &a, int nb, hipStream_t st)
{
    if (a.k <= 16) launch_r<1, 16, NW>(a, nb, st);
    else if (a.k <= 32) launch_r<2, 32, NW>(a, nb, st);
    else if (a.k <= 48) launch_r<3, 48, NW>(a, nb, st);
    else launch_r<4, 50, NW>(a, nb, st);
}
// nb workgroups of NW = 4 or 8 wavefronts (4 columns each); a.k <= SWEEPR_KMAX, the caller's rank padding KP = 16 ceil(k / 16)
void nnlm_tu_sweep_r(const SweepArgs &a, int nb, int NW, hipStream_t st)
{
    if (NW == 8) launch_k<8>(a, nb, st);
    else launch_k<4>(a, nb, st);
}
