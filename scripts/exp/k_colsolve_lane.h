// k_colsolve_lane.h -- colsolve_f32_kernel, the fp32-operand mode's per-column SCD solver until late in round 6 (one wavefront per column, lane =
// coordinate, lane q's delta through v_readlane / v_writelane), as it stood when colsolve_row_kernel (nnlm_amd/csrc/k_colsolve_row.h) replaced it:
// config 5 1.91 -> 1.83 ms per step, bit-identical results (profiles/r06_colsolve_row_ab.log, scripts/exp/lane_exp.hip).  Not compiled into the product.
#pragma once
#include "../../nnlm_amd/csrc/common.h"
#include "../../nnlm_amd/csrc/k_sweep.h"
__device__ static inline double readlane_f64(double v, int src)
{
    int2 p = __builtin_bit_cast(int2, v);
    p.x = __builtin_amdgcn_readlane(p.x, src);
    p.y = __builtin_amdgcn_readlane(p.y, src);
    return __builtin_bit_cast(double, p);
}
// ------------------------------------------------------------------------------------------------------------------
// colsolve_f32_kernel -- SCD-LS with a Gram of its own per column (or one shared Gram), fp32-operand mode (k <= 64).
//
// colsolve_ls_kernel above spends ~15 fp64 instructions per coordinate (four v_readlane pairs, reciprocal + Markstein quotient, compare,
// select): 2.9 ms per half-step at config 5.  This kernel runs the same recurrence in the arithmetic of the fp32-operand mode (rows of G
// divided by their diagonal, nu = mu / G[q][q], d = max(-x, -nu): ONE instruction) with one wavefront per column, lane = coordinate;
// lane q alone takes its coordinate's step under an execution mask of one lane, the delta reaches every lane's gradient through an SGPR.
// Rounds 2-5 ran the chain in fp64 (colsolve_fast_kernel, scripts/exp/csrc_r5/k_missing.h: five vector instructions per step, 104
// registers of Gram row, 4 wavefronts per SIMD -- 0.33 / 0.66 ms per half-step at config 5).  Round 6 (the mode's contract is 1e-4 on
// W, H; k_sweep_f.h): the chain on fp32 state.  The starting gradient nu0 = (G x - c + L1) / diag is still formed in fp64 (that is where the cancellation is) WHILE
// the scaled Gram row is read, so no fp64 copy of the row is ever held; the row lives in KR fp32 registers (52 instead of 104 at k = 50:
// 5 wavefronts per SIMD instead of 4); a step is four vector instructions instead of five (one v_readlane_b32), all of them fp32 (2.9
// against 5.1 cycles per instruction and SIMD, scripts/exp/valu_exp.hip).
#ifndef CSF_BATCH
#define CSF_BATCH 16
#endif
template <int NKQ, bool HAS_MASK, int KR = 16 * NKQ>
__global__ __launch_bounds__(256) void colsolve_f32_kernel(const SweepArgs a, size_t g_stride)
{
    const int lane = threadIdx.x & 63;
    const int col = a.col0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= a.ncols) return; // whole wavefront
    const int k = a.k;
    const bool lv = lane < k;
    const int lq = lv ? lane : 0;
    const double *G = a.Graw + (size_t)col * g_stride;
    unsigned long long mword = 0ull;
    if (HAS_MASK) mword = a.mask[col];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    const bool skip = HAS_MASK && ((mword & kmask) == kmask); // arma::all(mask.col(j)), src/update_with_missing.cpp:75-76

    double gd = 1.0; // edited G[lane][lane] (src/update_with_missing.cpp:98-103)
    if (lv) {
        gd = G[(size_t)lq * a.KPg + lq];
        if (a.r0 != a.r1) gd += a.r0 - a.r1;
        if (a.r1 != 0) gd += a.r1;
        gd += NNLM_TINY;
    }
    const double rgd = 1.0 / gd;
    const double x64 = lv ? a.X[(size_t)lq * a.ldx + col] : 0.0;
    double cv = 0.0;
    if (lv)
        for (int s = 0; s < a.nslabs; s++) cv += a.Cx[(size_t)s * a.slab_stride + (size_t)lq * a.ldc + col];
    // nu = (G x - c + L1) / G[lane][lane] in fp64, row `lane` of the scaled Gram (G is symmetric: G[lane][q] = G[q][lane], a coalesced read)
    // kept in fp32
    double nu64 = lv ? (((a.r2 != 0) ? a.r2 - cv : -cv) * rgd) : 0.0;
    float gs[KR];
#pragma unroll
    for (int q0 = 0; q0 < KR; q0 += CSF_BATCH) { // (CSF_BATCH fp64 loads in flight at a time: this is the kernel's register peak)
        double gv[CSF_BATCH];
#pragma unroll
        for (int e = 0; e < CSF_BATCH; e++) {
            const int q = q0 + e;
            // (g_upper: the per-column Gram holds its upper triangle only -- row `lane` of the symmetric matrix is column `lane` down to the
            //  diagonal, then row `lane`)
            gv[e] = (q < KR && q < k && lv) ? G[(a.g_upper && q > lane) ? (size_t)lane * a.KPg + q : (size_t)q * a.KPg + lane] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < CSF_BATCH; e++) {
            const int q = q0 + e;
            if (q < KR) {
                double v = 0.0;
                if (q < k && lv) {
                    v = gv[e];
                    if (q == lane && a.r0 != a.r1) v += a.r0 - a.r1;
                    if (a.r1 != 0) v += a.r1;
                    if (q == lane) v += NNLM_TINY;
                    v *= rgd;
                }
                float g32 = (float)v;
                asm volatile("" : "+v"(g32)); // (a register of its own: left to the allocator, the 52 floats sit in the low halves of 52 register PAIRS)
                gs[q] = g32;
                if (q < k) nu64 = __builtin_fma(readlane_f64(x64, q), v, nu64);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float x = (float)x64, nu = (float)nu64;

    unsigned t = 0;
    if (!skip) {
        const float tol = (float)a.rel_tol, tole = tol * (float)NNLM_TINY;
        bool more = true; // rel = 1 + rel_tol > rel_tol
        for (; t < a.max_iter && more; t++) {
            // a step:   v_min_f32 t = min(x, nu) (every lane: lane q's entry is MINUS the step's delta);  v_readlane_b32 e = t[q];
            //           v_writelane_b32 xd[q] = e;  v_fma_f32 nu -= e * Gs[q]        -- four vector instructions, NO scalar ones.
            // A coordinate moves once per sweep: lane q's x is still the sweep's starting value at ITS step, so x is brought up to date once,
            // behind the sweep (x = x0 - xd), and xd is ONE register through the sweep.  (Until round 6 the step switched the execution mask to
            // lane q around v_max / v_add: three to five s_mov per step on the ONE scalar unit the CU's four SIMDs share -- at five
            // wavefronts per SIMD as much scalar as vector issue time.  Same values to the last bit.)
            const float x0 = x;
            float xd = 0.0f;
            int kk = k;
            asm volatile("" : "+s"(kk)); // (opaque per sweep: otherwise 64 hoisted "q < k" masks spill into VGPR lanes)
            auto step = [&](const int q) {
                // (v_min, v_readlane and v_fma are the compiler's: it knows their wait states -- a vector result read by v_readlane, a scalar
                //  written by a vector instruction read by the next one: two on gfx940+.  v_writelane has no builtin in this toolchain; written
                //  out BEHIND the fused multiply-add -- tied to its result --, so that those wait states have passed for it too)
                const float tq = __builtin_fminf(x0, nu); // e = -delta = min(x, nu)  (max(x - nu, 0) - x = -min(x, nu))
                const int ei = __builtin_amdgcn_readlane(__builtin_bit_cast(int, tq), q);
                nu = __builtin_fmaf(-__builtin_bit_cast(float, ei), gs[q], nu);
                asm("v_writelane_b32 %0, %1, %2" : "+v"(xd) : "s"(ei), "n"(q), "v"(nu)); // (nu: an input only -- as an output the compiler canonicalises it before the next v_min)
            };
#pragma unroll
            for (int c = 0; c < NKQ; c++) {
                if (!HAS_MASK && 16 * c + 16 <= KR && 16 * c + 16 <= kk) { // a whole block of 16 coordinates: no per-step test
#pragma unroll
                    for (int e = 0; e < 16; e++) step(16 * c + e);
                } else if (16 * c < kk) {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        if (16 * c + e < KR) // (compile time)
                            if (16 * c + e < kk && !(HAS_MASK && ((mword >> (16 * c + e)) & 1ull))) step(16 * c + e); // wave-uniform
                }
            }
            x = x0 - xd;
            const bool big = 2.0f * __builtin_fabsf(xd) > __builtin_fmaf(tol, x + x0, tole); // src/base_algorithms.cpp:29-32 without the division
            more = __ballot(big && lv) != 0ull || 0.0f > tol;
        }
    }
    if (lv) {
        // (masked coordinates never took a step: their fp32 copy equals the rounded input -- hand the fp64 input back unchanged)
        const double xo = (HAS_MASK && ((mword >> lane) & 1ull)) || skip ? x64 : (double)x;
        a.Xout[(size_t)lane * a.ldo + (col - a.ocol0)] = xo;
        if (a.op_mode == 1) {
            if (a.op_f64) ((double *)a.op)[(size_t)lane * a.op_ld + col] = xo;
            else ((float *)a.op)[(size_t)lane * a.op_ld + col] = (float)xo;
        }
    }
    if (lane == 0 && t) atomicAdd(a.sweeps, (unsigned long long)t);
}

