// k_colsolve_row.h -- per-column SCD solver of the fp32-operand mode with four columns per wavefront (NA flow, ranks <= 64).
// Instantiated in tu_colsolve.hip only (a translation unit of its own: ten instantiations of a fully unrolled 50 x 50 recurrence).
#pragma once
#include "common.h"
#include "k_sweep.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------------------------
// colsolve_row_kernel -- SCD-LS with a Gram of its own per column (src/update_with_missing.cpp:86-111, src/base_algorithms.cpp:3-37) in the
// arithmetic of the fp32-operand mode: rows of the edited Gram divided by their diagonal, nu = mu / G[q][q], e = -delta = min(x, nu)
// (max(x - nu, 0) - x = -min(x, nu)); the starting gradient nu0 = (G x - c + L1) / diag in fp64 (that is where the cancellation is), the
// 2500-step chain on fp32 state.  FOUR columns per wavefront: a column is a row of 16 lanes, a lane owns CPL = ceil(k / 16) consecutive
// coordinates (q = CPL l + rr) and holds, for every step s, minus the scaled Gram entries G'[s][q] of its own coordinates (k CPL fp32
// registers: 200 at k = 50, two wavefronts per SIMD).
//
// Its predecessor (one wavefront per column, lane = coordinate: scripts/exp/k_colsolve_lane.h) took lane q's delta to the other lanes
// through v_readlane / v_writelane: ~9 of the step's ~14 cycles of SIMD time whatever the occupancy, the wait states nothing, the
// arithmetic 5 (scripts/exp/lane_exp.hip, profiles/r06_lane_exp.log).  Here the delta reaches the 16 lanes of ITS column through the DPP
// row broadcast gfx90a+ has (row_newbcast:L -- lane L of every row to its row), on the vector pipe, not through the scalar unit:
//     e = v_min(x[r], nu[r])   (every lane; lane L = s / CPL of each row holds its column's value)
//     xd[r] = v_cndmask(lanes L of the four rows, e, xd[r])         (the mask: a constant SGPR pair, held for all 16 L)
//     eb = v_mov_dpp row_newbcast:L(e);  nu[0..CPL-1] += eb * (-G'[s][..])   (packed FMAs)
// -- 5 vector instructions for the steps of FOUR columns at k = 50: 9.7 ns per step and SIMD against 4 x 6.1.  x is brought up to date once
// per sweep (a coordinate moves once per sweep).  Same arithmetic and order of operations per column as the predecessor: bit-identical
// results.  Masked coordinates cannot be skipped per column (the four columns of a wavefront carry different masks): their x enters the
// minimum as 0 and their gradient as +1e30 -- e = 0 for good, as in k_sweep_f.h --, their output is the fp64 input.  A column that has
// converged keeps computing with the wavefront; its values are taken at the sweep that ended it.
template <int I, int N, class F> __device__ __forceinline__ void csr_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        csr_for<I + 1, N>(f);
    }
}
#define CSR_NEWBCAST 0x150 // DPP control row_newbcast:0 (+ L)
template <int CPL, bool HAS_MASK, int KR>
__global__ __launch_bounds__(256, (KR * CPL > 208) ? 1 : 2) void colsolve_row_kernel(const SweepArgs a, size_t g_stride)
{
    static_assert(KR <= 16 * CPL && KR > 16 * (CPL - 1), "CPL = ceil(KR / 16)");
    const int lane = threadIdx.x & 63, l = lane & 15, rw = lane >> 4, wave = threadIdx.x >> 6;
    const int col = a.col0 + blockIdx.x * 16 + wave * 4 + rw;
    const bool in_range = col < a.ncols;
    const int cc = in_range ? col : a.col0;
    const int k = a.k;
    const double *G = a.Graw + (size_t)cc * g_stride;
    unsigned long long mword = 0ull;
    if (HAS_MASK) mword = a.mask[cc];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    const bool skip = HAS_MASK && ((mword & kmask) == kmask); // arma::all(mask.col(j)), src/update_with_missing.cpp:75-76

    bool lv[CPL], msk[CPL];
#pragma unroll
    for (int rr = 0; rr < CPL; rr++) {
        const int q = CPL * l + rr;
        lv[rr] = q < k;
        msk[rr] = HAS_MASK && lv[rr] && ((mword >> q) & 1ull);
    }
    // Set-up, one column at a time with lane = coordinate j (nu0 = (G x - c + L1) / diag in fp64 while row j of the scaled Gram is read --
    // upper triangle only, G[min][max]; x_q by an LDS broadcast read): every scaled entry goes
    // to the wavefront's LDS image as soon as it is rounded ([step q][coordinate j], fp32), and the row of 16 lanes that owns the column
    // picks up ITS registers from there: -G'[s][CPL l .. CPL l + CPL - 1] for every step (one read per step), nu0 and x.  (The same
    // set-up written directly in the row layout -- 200 fp64 loads per lane with per-entry selects, x_s by row broadcast -- compiled to
    // 7000 instructions and ~100-500 spilled registers in every arrangement tried.)
    __shared__ float img_all[4][KR * 64 + 128];
    __shared__ double xq_all[4][64];
    float *img = img_all[wave], *nuL = img + KR * 64, *xL = nuL + 64;
    double *xq = xq_all[wave]; // the column's x in fp64: x_q reaches all lanes as an LDS broadcast read (one instruction; v_readlane x 2 costs ~9 cycles)
    float gneg[KR][CPL], x[CPL], nu[CPL];
    // (every lane is in exactly one of the four rows: all of these are assigned below)
    const int colw = a.col0 + blockIdx.x * 16 + wave * 4; // first of the wavefront's four columns
    const bool e01 = a.r0 != a.r1, e1 = a.r1 != 0;        // (wave-uniform: the reference's edits of the Gram, src/update_with_missing.cpp:98-103)
    const double dall = e1 ? a.r1 : 0.0;
    csr_for<0, 4>([&](auto cq) {
        constexpr int c = decltype(cq)::value;
        const bool c_in = colw + c < a.ncols; // wave-uniform
        const int ccl = c_in ? colw + c : a.col0;
        const double *Gc = a.Graw + (size_t)ccl * g_stride;
        const bool lj = lane < k;
        const int lq = lj ? lane : 0;
        double gd = 1.0; // edited G[lane][lane] (src/update_with_missing.cpp:98-103)
        if (lj) {
            gd = Gc[(size_t)lq * a.KPg + lq];
            if (e01) gd += a.r0 - a.r1;
            if (e1) gd += a.r1;
            gd += NNLM_TINY;
        }
        const double rgdl = 1.0 / gd;
        const double rgde = lj ? rgdl : 0.0; // (lanes beyond k: every scaled entry 0)
        const double vdiag = lj ? gd * rgdl : 0.0; // the scaled diagonal entry: the edit sequence below applied to G[lane][lane] IS gd
        const double x64l = (lj && c_in) ? a.X[(size_t)lq * a.ldx + ccl] : 0.0;
        double cv = 0.0;
        if (lj)
            for (int sl = 0; sl < a.nslabs; sl++) cv += a.Cx[(size_t)sl * a.slab_stride + (size_t)lq * a.ldc + ccl];
        double nu64l = lj ? (((a.r2 != 0) ? a.r2 - cv : -cv) * rgdl) : 0.0;
        xq[lane] = x64l;
        // entry (q, lane) of the upper-triangle store: G[min][max]; 32-bit element offsets from the column's base
        const unsigned offr = (unsigned)lq * (unsigned)a.KPg, offc = (unsigned)lq;
        constexpr int QB = 8;
#pragma unroll
        for (int q0 = 0; q0 < KR; q0 += QB) {
            if (q0 < k) { // wave-uniform; steps q >= k of a partly valid batch: G is not read (v = 0 through the guard on the load and dall1 below)
                double gv[QB];
#pragma unroll
                for (int e = 0; e < QB; e++) {
                    const int q = q0 + e;
                    const unsigned off = (a.g_upper && q > lane) ? offr + (unsigned)q : (unsigned)q * (unsigned)a.KPg + offc;
                    gv[e] = (q < KR && q < k) ? Gc[off] : 0.0;
                }
#pragma unroll
                for (int e = 0; e < QB; e++) {
                    const int q = q0 + e;
                    if (q < KR) {
                        // src/update_with_missing.cpp:98-103 off the diagonal: + r1 (adding 0.0 when r1 == 0 changes nothing); rows divided by their diagonal
                        double v = (gv[e] + ((q < k) ? dall : 0.0)) * rgde;
                        v = (q == lane) ? ((q < k) ? vdiag : 0.0) : v;
                        img[q * 64 + lane] = -(float)v;
                        nu64l = __builtin_fma(xq[q < k ? q : 0], v, nu64l); // (v = 0 beyond k)
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < QB; e++)
                    if (q0 + e < KR) img[(q0 + e) * 64 + lane] = 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        nuL[lane] = (float)nu64l;
        xL[lane] = (float)x64l;
        if (rw == c) { // the 16 lanes that own this column
#pragma unroll
            for (int s = 0; s < KR; s++)
#pragma unroll
                for (int rr = 0; rr < CPL; rr++) gneg[s][rr] = img[s * 64 + CPL * l + rr];
#pragma unroll
            for (int rr = 0; rr < CPL; rr++) {
                nu[rr] = msk[rr] ? 1e30f : nuL[CPL * l + rr];
                x[rr] = xL[CPL * l + rr];
            }
        }
        asm volatile("" ::: "memory"); // (the next column's image is written behind these reads)
    });
    float xm[CPL], xfin[CPL];
#pragma unroll
    for (int rr = 0; rr < CPL; rr++) xfin[rr] = x[rr];
    // lanes L, 16 + L, 32 + L, 48 + L: the owner of a step in each of the four rows
    unsigned long long own[16];
#pragma unroll
    for (int L = 0; L < 16; L++) {
        own[L] = 0x0001000100010001ull << L;
        asm volatile("" : "+s"(own[L])); // (held in SGPRs: rebuilt per step it is two scalar instructions per step)
    }

    unsigned t_col = 0;
    bool act = in_range && !skip;
    const float tol = (float)a.rel_tol, tole = tol * (float)NNLM_TINY;
    for (unsigned t = 0; t < a.max_iter && __any(act); t++) {
        float x0[CPL], xd[CPL];
#pragma unroll
        for (int rr = 0; rr < CPL; rr++) x0[rr] = x[rr], xd[rr] = 0.0f, xm[rr] = msk[rr] ? 0.0f : x[rr];
        int kk = k;
        asm volatile("" : "+s"(kk)); // (opaque per sweep)
        auto step = [&](auto sc) {
            constexpr int s = decltype(sc)::value, L = s / CPL, r = s % CPL;
            const float e = __builtin_fminf(xm[r], nu[r]); // e = -delta = min(x, nu)
            asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(xd[r]) : "v"(e), "s"(own[L]));
            const float eb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, e), CSR_NEWBCAST + L, 0xF, 0xF, false));
#pragma unroll
            for (int rr = 0; rr < CPL; rr++) nu[rr] = __builtin_fmaf(eb, gneg[s][rr], nu[rr]);
        };
        // (coordinates in blocks of 16 steps: one wave-uniform test per block, none per step; steps s >= k of the last block are inert --
        //  x = 0, nu = 0, G' = 0 there: e = 0)
        csr_for<0, (KR + 15) / 16>([&](auto cc_) {
            constexpr int c = decltype(cc_)::value;
            if (16 * c < kk) csr_for<16 * c, (16 * c + 16 < KR ? 16 * c + 16 : KR)>(step);
        });
        bool big = false;
#pragma unroll
        for (int rr = 0; rr < CPL; rr++) {
            x[rr] = x0[rr] - xd[rr];
            big = big || (lv[rr] && 2.0f * __builtin_fabsf(xd[rr]) > __builtin_fmaf(tol, x[rr] + x0[rr], tole)); // src/base_algorithms.cpp:29-32 without the division
        }
        const unsigned rowbits = (unsigned)(__ballot(big) >> (16 * rw)) & 0xFFFFu;
        if (act) {
            t_col++;
            if (!(rowbits != 0u || 0.0f > tol)) act = false;
#pragma unroll
            for (int rr = 0; rr < CPL; rr++) xfin[rr] = x[rr]; // (the values of the last sweep this column ran)
        }
    }
    if (in_range) {
#pragma unroll
        for (int rr = 0; rr < CPL; rr++) {
            const int q = CPL * l + rr;
            if (lv[rr]) {
                const double xo = (msk[rr] || skip) ? a.X[(size_t)q * a.ldx + col] : (double)xfin[rr]; // (masked: the fp64 input, unchanged)
                a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xo;
                if (a.op_mode == 1) {
                    if (a.op_f64) ((double *)a.op)[(size_t)q * a.op_ld + col] = xo;
                    else ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xo;
                }
            }
        }
    }
    {
        const long long tot = wave_sum_ll((l == 0 && in_range) ? (long long)t_col : 0ll);
        if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
}

