// tu_sweepq.h -- launch entries of the SCD sweep kernels of k_sweep_q.h.  Their instantiations (one per block count NB = 1 .. 16,
// mask, arithmetic mode: 64 + 64) dominate the build, so they are compiled as translation units of their own, in parallel with
// nnlm_mi355x.hip (nnlm_amd/build.py): tu_sweepq.hip (sweep_scd_q_kernel) and tu_sweepqw.hip (sweep_scd_qw_kernel, the persistent form).
// Both return what hipFuncSetAttribute(MaxDynamicSharedMemorySize) said; the launch itself is checked by the caller (LAUNCHCHK).
#pragma once
#include "k_sweep.h"

// nb workgroups of 64 columns (four wavefronts of 16); img = the operand image written by sweepq_pack_kernel
hipError_t nnlm_tu_sweep_q(const SweepArgs &a, const double *img, int nb, int NB, bool strict, hipStream_t st);
// nb workgroups of G column groups of 16 (one workgroup per CU)
hipError_t nnlm_tu_sweep_qw(const SweepArgs &a, const double *img, int nb, int NB, bool strict, int G, hipStream_t st);
// fp32-operand mode (k_sweep_f.h, tu_sweepf.hip): nb workgroups of NW = 4 or 8 wavefronts of 16 columns; reads a.Graw, no operand image
hipError_t nnlm_tu_sweep_f(const SweepArgs &a, int nb, int NB, int NW, hipStream_t st);
// fp32-operand mode, per-column Grams (k_colsolve_row.h, tu_colsolve.hip): columns a.col0 .. a.ncols - 1, four per wavefront
void nnlm_tu_colsolve_row(const SweepArgs &a, size_t g_stride, hipStream_t st);
// fp32-operand mode, row form (k_sweep_r.h, tu_sweepr.hip): nb workgroups of NW = 4 or 8 wavefronts of FOUR columns; a.k <= SWEEPR_KMAX; reads a.Graw
#define SWEEPR_KMAX 50
void nnlm_tu_sweep_r(const SweepArgs &a, int nb, int NW, hipStream_t st);
