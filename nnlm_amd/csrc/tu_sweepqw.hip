// tu_sweepqw.hip -- the instantiations of sweep_scd_qw_kernel (k_sweep_q.h, the persistent form), see tu_sweepq.h.
#include "tu_sweepq.h"
#include "k_sweep_q.h"

template <int NT, int NB, bool M, bool S> static hipError_t launch_k(const SweepArgs &a, const double *img, int nb, int G, hipStream_t st)
{
    size_t lds = sweepqw_lds_bytes(16 * NT, NB, S, G);
    if (lds < (size_t)82 * 1024) lds = (size_t)82 * 1024; // (more than half a CU's LDS: one workgroup per CU, one wavefront per SIMD)
    const hipError_t e = hipFuncSetAttribute((const void *)sweep_scd_qw_kernel<NT, NB, M, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    sweep_scd_qw_kernel<NT, NB, M, S><<<nb, SWEEPQ_THREADS, lds, st>>>(a, img, G);
    return e;
}
template <int NT, int NB> static hipError_t launch_m(const SweepArgs &a, const double *img, int nb, bool strict, int G, hipStream_t st)
{
    (void)strict; // (the fp32-operand mode has its own kernel since round 6, k_sweep_f.h / tu_sweepf.hip: only the strict arithmetic is instantiated here)
    return a.mask ? launch_k<NT, NB, true, true>(a, img, nb, G, st) : launch_k<NT, NB, false, true>(a, img, nb, G, st);
}
hipError_t nnlm_tu_sweep_qw(const SweepArgs &a, const double *img, int nb, int NB, bool strict, int G, hipStream_t st)
{
    switch (NB) {
    case 1: return launch_m<1, 1>(a, img, nb, strict, G, st);
    case 2: return launch_m<1, 2>(a, img, nb, strict, G, st);
    case 3: return launch_m<1, 3>(a, img, nb, strict, G, st);
    case 4: return launch_m<1, 4>(a, img, nb, strict, G, st);
    case 5: return launch_m<2, 5>(a, img, nb, strict, G, st);
    case 6: return launch_m<2, 6>(a, img, nb, strict, G, st);
    case 7: return launch_m<2, 7>(a, img, nb, strict, G, st);
    case 8: return launch_m<2, 8>(a, img, nb, strict, G, st);
    case 9: return launch_m<3, 9>(a, img, nb, strict, G, st);
    case 10: return launch_m<3, 10>(a, img, nb, strict, G, st);
    case 11: return launch_m<3, 11>(a, img, nb, strict, G, st);
    case 12: return launch_m<3, 12>(a, img, nb, strict, G, st);
    case 13: return launch_m<4, 13>(a, img, nb, strict, G, st);
    case 14: return launch_m<4, 14>(a, img, nb, strict, G, st);
    case 15: return launch_m<4, 15>(a, img, nb, strict, G, st);
    default: return launch_m<4, 16>(a, img, nb, strict, G, st);
    }
}
