// k_sweep_r.h -- SCD least-squares sweep of the fp32-operand mode, row form (round 6, last session): the recurrence of scd_ls_update
// (reference src/base_algorithms.cpp:3-37) for the DENSE half-step (one Gram for all columns) in the shape of colsolve_row_kernel
// (k_colsolve_row.h): FOUR columns per wavefront, a column is a row of 16 lanes, a lane owns CPL = ceil(k / 16) consecutive coordinates
// and holds -G'[s][q] of its coordinates for every step s (k CPL fp32 registers: 200 at k = 50, two wavefronts per SIMD); a step is
//     e = v_min(x[r], nu[r]);  xd[r] = v_cndmask(owner lanes, e, xd[r]);  eb = v_mov_dpp row_newbcast:L(e);  nu[..] += eb * (-G'[s][..])
// -- five vector instructions for the steps of four columns, 9.7 ns per step and SIMD at two wavefronts per SIMD
// (scripts/exp/lane_exp.hip, profiles/r06_lane_exp.log).
//
// When it is taken (launch_sweep_f): while the launch fits ONE round of its wavefronts -- at most two per SIMD, i.e. up to 32 columns per
// CU (8192 on the whole device).  k_sweep_f.h (one wavefront per 16 columns, the rank-4 update on the bf16 matrix pipe) spends 0.53
// instructions per column and step against 1.25 here, but its chain is 238 cycles per block of four steps for a lone wavefront whatever
// the other SIMDs do: ~0.1 ms per launch from 16 columns to 16384.  A lone wavefront of THIS form takes 12.8 ns per step (32 us per 50
// sweeps of 50 coordinates), a pair 19.5 ns each -- no matrix-instruction latency in the chain.  Beyond one round the form loses: its
// wavefronts hold 200 registers, a CU holds eight of them, and 20000 columns are three rounds of 49 us chains (0.25 against 0.144 ms,
// profiles/r06_sweep_row_ab.log).  So: the sweeps of mid-size problems and of every multi-GPU shard of the benchmark (<= 8192 columns).
// Set-up per workgroup (4 or 8 wavefronts, 16 or 32 columns): G' = edited Gram (src/update_with_missing.cpp:20-24) with the rows divided by their
// diagonal (diagonal exactly 1), built ONCE in LDS as fp64 [step][coordinate] and as -fp32; per column nu0 = ((L1 - c) + G x) / diag in
// fp64 in the row layout (x_s from an LDS image of the wavefront's four columns), rounded to fp32 once.  Masked coordinates as in k_sweep_f.h (x = 0, nu = +1e30: e = 0
// for good; output = the fp64 input).  Epilogue: sweepq_epilogue (factor outputs, max|x|, Gram partial sums of the workgroup's columns).
#pragma once
#include "common.h"
#include "k_sweep.h"
#include "k_sweep_q.h"
#include "k_colsolve_row.h"

// edited Gram entry E[r][c] (src/update_with_missing.cpp:20-24), the additions in the order of sweepq_img_put()
__device__ __forceinline__ double sweepr_edit(double g, bool diag, double r0, double r1)
{
    if (diag && r0 != r1) g += r0 - r1;
    if (r1 != 0) g += r1;
    if (diag) g += NNLM_TINY;
    return g;
}

// NW: wavefronts per workgroup -- 4 (16 columns, one wavefront per SIMD) while the launch has at most one workgroup per CU, 8 (32 columns) beyond
template <int CPL, bool HAS_MASK, int KR, int NT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void sweep_row_kernel(const SweepArgs a)
{
    static_assert(KR <= 16 * CPL && KR > 16 * (CPL - 1) && NT == CPL, "CPL = ceil(KR / 16) = rank padding / 16");
    constexpr int COLS = 4 * NW, KP = 16 * NT, XS = KP + 2, THREADS = 64 * NW;
    constexpr int G64_BYTES = KR * 64 * 8, XL_BYTES = COLS * XS * 8;
    __shared__ __attribute__((aligned(16))) unsigned char r0_all[G64_BYTES > XL_BYTES ? G64_BYTES : XL_BYTES]; // G' fp64 during the set-up, then the x image
    __shared__ __attribute__((aligned(16))) float img[KR * 64];                                                // -G' fp32 [step][coordinate]
    __shared__ double rinv[64];
    __shared__ double xq_all[NW][4][64]; // x of the wavefront's four columns in fp64 (set-up only)
    double *g64 = (double *)r0_all, *xl = (double *)r0_all;
    const int tid = threadIdx.x, lane = tid & 63, l = lane & 15, rw = lane >> 4, wave = tid >> 6;
    const int k = a.k;
    const int col_base = a.col0 + blockIdx.x * COLS;
    const int cl = 4 * wave + rw, col = col_base + cl;
    const bool in_range = col < a.ncols;
    const int cc = in_range ? col : a.col0;
    unsigned long long mword = 0ull;
    if (HAS_MASK) mword = a.mask[cc];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    const bool skip = HAS_MASK && ((mword & kmask) == kmask); // arma::all(mask.col(j)) -> column skipped

    // the column's own inputs, requested ahead of everything else
    bool lv[CPL], msk[CPL];
    double x64[CPL], cv[CPL];
#pragma unroll
    for (int rr = 0; rr < CPL; rr++) {
        const int q = CPL * l + rr;
        lv[rr] = q < k;
        msk[rr] = HAS_MASK && lv[rr] && ((mword >> q) & 1ull);
        x64[rr] = (lv[rr] && in_range) ? a.X[(size_t)(lv[rr] ? q : 0) * a.ldx + cc] : 0.0;
        cv[rr] = 0.0;
    }
    for (int s = 0; s < a.nslabs; s++) {
        const double *cs = a.Cx + (size_t)s * a.slab_stride + cc;
#pragma unroll
        for (int rr = 0; rr < CPL; rr++) cv[rr] += lv[rr] ? cs[(size_t)(CPL * l + rr) * a.ldc] : 0.0;
    }
    // ---- G' in LDS: g64[s][q] = E[q][s] / E[q][q] (q: the coordinate whose gradient the entry updates; diagonal exactly 1), img = -(float)
    if (tid < 64) rinv[tid] = (tid < k) ? 1.0 / sweepr_edit(a.Graw[(size_t)tid * a.KPg + tid], true, a.r0, a.r1) : 1.0;
    {
        constexpr int PER = (KR * 64 + THREADS - 1) / THREADS;
        double v[PER]; // (all loads of the thread in flight at once)
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int e = tid + i * THREADS, s = e >> 6, q = e & 63;
            v[i] = (e < KR * 64 && s < k && q < k) ? a.Graw[(size_t)q * a.KPg + s] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int e = tid + i * THREADS, s = e >> 6, q = e & 63;
            if (e < KR * 64) {
                const double g = (s < k && q < k) ? ((s == q) ? 1.0 : sweepr_edit(v[i], false, a.r0, a.r1) * rinv[q]) : 0.0;
                g64[e] = g;
                img[e] = -(float)g;
            }
        }
        __syncthreads();
    }
    // ---- starting gradients in fp64, row layout: nu0[q] = (L1 - c[q]) / diag + sum_s G'[q][s] x[s]
    double nu64[CPL];
#pragma unroll
    for (int rr = 0; rr < CPL; rr++) nu64[rr] = lv[rr] ? ((a.r2 != 0) ? a.r2 - cv[rr] : -cv[rr]) * rinv[CPL * l + rr] : 0.0;
    // (x_s of the lane's column from an LDS image of the wavefront's four columns -- a rolled loop: written with the DPP row broadcast, whose
    //  control is an immediate, the loop is unrolled and the masked instantiation spilled ~280 registers here)
#pragma unroll
    for (int rr = 0; rr < CPL; rr++) xq_all[wave][rw][CPL * l + rr] = x64[rr];
    {
        const double *xr = xq_all[wave][rw], *gr = g64 + CPL * l;
#pragma unroll 2
        for (int s = 0; s < k; s++) {
            const double xs = xr[s];
#pragma unroll
            for (int rr = 0; rr < CPL; rr++) nu64[rr] = __builtin_fma(xs, gr[s * 64 + rr], nu64[rr]);
        }
    }
    float gneg[KR][CPL], x[CPL], nu[CPL], xm[CPL], xfin[CPL];
#pragma unroll
    for (int rr = 0; rr < CPL; rr++) {
        x[rr] = xfin[rr] = (float)x64[rr];
        nu[rr] = msk[rr] ? 1e30f : (float)nu64[rr];
        asm volatile("" : "+v"(nu[rr])); // (rounded here: nothing fp64 lives on)
    }
#pragma unroll
    for (int s = 0; s < KR; s++)
#pragma unroll
        for (int rr = 0; rr < CPL; rr++) gneg[s][rr] = img[s * 64 + CPL * l + rr];
    __syncthreads(); // G' fp64 is not read beyond this point: its place becomes the x image

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
            const float e = __builtin_fminf(xm[r], nu[r]); // e = -delta = min(x, nu)   (max(x - nu, 0) - x = -min(x, nu))
            asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(xd[r]) : "v"(e), "s"(own[L]));
            const float eb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, e), CSR_NEWBCAST + L, 0xF, 0xF, false));
#pragma unroll
            for (int rr = 0; rr < CPL; rr++) nu[rr] = __builtin_fmaf(eb, gneg[s][rr], nu[rr]);
        };
        // (coordinates in blocks of 16 steps: one wave-uniform test per block, none per step; steps s >= k of the last block are inert)
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
            if (!(rowbits != 0u || 0.0f > tol)) act = false; // src/base_algorithms.cpp:35
#pragma unroll
            for (int rr = 0; rr < CPL; rr++) xfin[rr] = x[rr]; // (the values of the last sweep this column ran)
        }
    }
    // the column's final values -> x image (masked entries and skipped columns: the input; rows of out-of-range columns zero; coordinates >= k zero)
#pragma unroll
    for (int rr = 0; rr < CPL; rr++) {
        const int q = CPL * l + rr;
        double v = 0.0;
        if (in_range && lv[rr]) v = (msk[rr] || skip) ? a.X[(size_t)q * a.ldx + col] : (double)xfin[rr]; // (re-read: nothing fp64 is kept through the sweeps)
        xl[cl * XS + q] = v;
    }
    __syncthreads(); // x image final
    sweepq_epilogue<NT, COLS, NW>(a, xl, COLS, col_base, (int)blockIdx.x);
    {
        const long long tot = wave_sum_ll((l == 0 && in_range) ? (long long)t_col : 0ll);
        if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
}
