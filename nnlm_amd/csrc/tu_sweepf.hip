// tu_sweepf.hip -- the instantiations of sweep_scd_f_kernel (k_sweep_f.h: SCD sweep of the fp32-operand mode, fp32 chain), see tu_sweepq.h.
#include "tu_sweepq.h"
#include "k_sweep_f.h"

template <int NT, int NB, bool M, int NW> static hipError_t launch_k(const SweepArgs &a, int nb, hipStream_t st)
{
    const int lds = (int)sweepf_lds_bytes(16 * NT, NB, 16 * NW); // G' / x image + operand image (up to 98 KB at k = 64, 132 KB with 8 wavefronts)
    const hipError_t e = hipFuncSetAttribute((const void *)sweep_scd_f_kernel<NT, NB, M, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    sweep_scd_f_kernel<NT, NB, M, NW><<<nb, 64 * NW, lds, st>>>(a);
    return e;
}
template <int NT, int NB> static hipError_t launch_m(const SweepArgs &a, int nb, int NW, hipStream_t st)
{
    if (NW == 8) return a.mask ? launch_k<NT, NB, true, 8>(a, nb, st) : launch_k<NT, NB, false, 8>(a, nb, st);
    return a.mask ? launch_k<NT, NB, true, 4>(a, nb, st) : launch_k<NT, NB, false, 4>(a, nb, st);
}
hipError_t nnlm_tu_sweep_f(const SweepArgs &a, int nb, int NB, int NW, hipStream_t st)
{
    switch (NB) {
    case 1: return launch_m<1, 1>(a, nb, NW, st);
    case 2: return launch_m<1, 2>(a, nb, NW, st);
    case 3: return launch_m<1, 3>(a, nb, NW, st);
    case 4: return launch_m<1, 4>(a, nb, NW, st);
    case 5: return launch_m<2, 5>(a, nb, NW, st);
    case 6: return launch_m<2, 6>(a, nb, NW, st);
    case 7: return launch_m<2, 7>(a, nb, NW, st);
    case 8: return launch_m<2, 8>(a, nb, NW, st);
    case 9: return launch_m<3, 9>(a, nb, NW, st);
    case 10: return launch_m<3, 10>(a, nb, NW, st);
    case 11: return launch_m<3, 11>(a, nb, NW, st);
    case 12: return launch_m<3, 12>(a, nb, NW, st);
    case 13: return launch_m<4, 13>(a, nb, NW, st);
    case 14: return launch_m<4, 14>(a, nb, NW, st);
    case 15: return launch_m<4, 15>(a, nb, NW, st);
    default: return launch_m<4, 16>(a, nb, NW, st);
    }
}
