// tu_colsolve.hip -- the instantiations of colsolve_row_kernel (k_colsolve_row.h), see tu_sweepq.h.
#include "tu_sweepq.h"
#include "k_colsolve_row.h"

template <int CPL, int KR> static void launch_r(const SweepArgs &a, size_t g_stride, int nb, hipStream_t st)
{
    if (a.mask) colsolve_row_kernel<CPL, true, KR><<<nb, 256, 0, st>>>(a, g_stride);
    else colsolve_row_kernel<CPL, false, KR><<<nb, 256, 0, st>>>(a, g_stride);
}
// columns a.col0 .. a.ncols - 1, sixteen per workgroup (four per wavefront); a.k <= 64
void nnlm_tu_colsolve_row(const SweepArgs &a, size_t g_stride, hipStream_t st)
{
    const int nb = (a.ncols - a.col0 + 15) / 16;
    if (nb <= 0) return;
    if (a.k <= 16) launch_r<1, 16>(a, g_stride, nb, st);
    else if (a.k <= 32) launch_r<2, 32>(a, g_stride, nb, st);
    else if (a.k <= 48) launch_r<3, 48>(a, g_stride, nb, st);
    else if (a.k <= 50) launch_r<4, 50>(a, g_stride, nb, st); // 200 registers of G': two wavefronts per SIMD
    else launch_r<4, 64>(a, g_stride, nb, st);               // 256: one
}
