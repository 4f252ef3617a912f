// k_sweep.h -- per-column solvers for the square-loss methods, fp64.
//
// Reference: the body of update()'s column loop (src/update_with_missing.cpp:29-53) followed by
//   method 1: scd_ls_update (src/base_algorithms.cpp:3-37)
//   method 2: lee_ls_update (src/base_algorithms.cpp:40-68)
//
// The coordinate loop inside one column is loop-carried (Gauss-Seidel) and is NOT parallelised.
// Parallelism is over columns: one lane = one column, 64 columns per wavefront; the k coordinates
// of the column (x) and its gradient (mu) live in that lane's VGPRs as NCH vectors of 8 doubles, so
// every instruction of the sequential recurrence does useful work on 64 columns.  The coordinate
// index q is wave-uniform, so x[q]/mu[q] are reached with s_set_gpr_idx (uniform indirect VGPR
// addressing) inside a real loop -- no 4096-FMA straight-line code, no scratch.  The Gram matrix G
// (shared by all columns) sits in LDS and is read with wave-uniform addresses (broadcast).
//
// Prologue (per column j): c = sum of the split-K slabs of the cross product (fixed order),
//   method 1: mu = G x - c (+ L1)      (src/update_with_missing.cpp:39-41)
//   method 2: keeps c                   (src/update_with_missing.cpp:45)
// G gets the regularisation edits of src/update_with_missing.cpp:20-24 while it is copied to LDS.
//
// Differences from the reference's arithmetic (all below 1 ulp per operation):
//   mu += d*G[:,q] is one fused multiply-add per entry (the reference rounds the product first).
#pragma once
#include "common.h"

typedef double f64x8 __attribute__((ext_vector_type(8)));

struct SweepArgs {
    double *X;          // [KPx][ldx] master copy of the factor being solved (row q, column index fastest)
    int ldx;
    const double *Graw; // [KPg][KPg] Gram of the other factor (no edits applied yet)
    int KPg;
    const double *Cx;   // [nslabs][KPg][ldc] split-K partial cross products
    size_t slab_stride;
    int nslabs;
    int ldc;
    int ncols;          // columns to solve
    int k;              // true rank
    double r0, r1, r2;  // L2, angle, L1 of this half-step (beta in the reference's update())
    const unsigned long long *mask; // [ncols] bit q set <=> entry (q, col) is masked; NULL = no mask
    unsigned max_iter;
    double rel_tol;
    void *op;           // GEMM-operand copy of the updated factor
    int op_mode;        // 0: none, 1: [KP][op_ld] (same layout as X), 2: [col][op_ld] (kq fastest)
    int op_ld;
    int op_f64;         // element type of op: 0 float, 1 double
    unsigned long long *sweeps; // += sum of per-column sweep counts
};

// Load G into LDS with the regularisation edits of src/update_with_missing.cpp:20-24.
template <int KP8>
__device__ static inline void sweep_load_gram(double *Gs, const SweepArgs &a, int tid, int nthreads)
{
    const int k = a.k;
    for (int e = tid; e < KP8 * KP8; e += nthreads) {
        const int q = e / KP8, r = e % KP8;
        double g = 0.0;
        if (q < k && r < k) {
            g = a.Graw[(size_t)q * a.KPg + r];
            if (q == r && a.r0 != a.r1) g += a.r0 - a.r1;
            if (a.r1 != 0) g += a.r1;
            if (q == r) g += NNLM_TINY;
        }
        Gs[e] = g;
    }
}

template <int NCH, int METHOD>
__global__ __launch_bounds__(64) void sweep_ls_kernel(const SweepArgs a)
{
    constexpr int KP8 = 8 * NCH;
    __shared__ double Gs[KP8 * KP8]; // Gs[q*KP8 + r] = edited G[q][r] (symmetric)
    const int lane = threadIdx.x;
    const int col = blockIdx.x * 64 + lane;
    const int k = a.k;
    sweep_load_gram<KP8>(Gs, a, lane, 64);
    __syncthreads();

    const bool in_range = col < a.ncols;
    const int cc = in_range ? col : 0;
    unsigned long long mword = 0ull;
    if (a.mask) mword = a.mask[cc];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    // skip columns whose coordinates are all masked (arma::all(mask.col(j)), :33)
    bool act = in_range && !(a.mask && ((mword & kmask) == kmask));

    f64x8 x[NCH], v[NCH]; // v = mu (method 1) or c = Y*b (method 2)
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int q = 8 * c + e;
            double xv = 0.0, cv = 0.0;
            if (q < k) {
                xv = a.X[(size_t)q * a.ldx + cc];
                for (int s = 0; s < a.nslabs; s++) cv += a.Cx[(size_t)s * a.slab_stride + (size_t)q * a.ldc + cc];
            }
            x[c][e] = xv;
            v[c][e] = cv;
        }
    if (METHOD == 1) { // mu = G x - c (+ L1)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            for (int e = 0; e < 8; e++) {
                const int q = 8 * c + e;
                double s0 = 0.0;
#pragma unroll
                for (int c2 = 0; c2 < NCH; c2++)
#pragma unroll
                    for (int e2 = 0; e2 < 8; e2++) s0 = __builtin_fma(Gs[q * KP8 + 8 * c2 + e2], x[c2][e2], s0);
                double muq = s0 - v[c][e];
                if (a.r2 != 0) muq += a.r2;
                v[c][e] = (q < k) ? muq : 0.0;
            }
        }
    }

    int t_lane = 0;
    unsigned t = 0;
    const double tol = a.rel_tol;
    while (t < a.max_iter && __any(act)) {
        double rel = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int qend = (k - 8 * c) < 8 ? (k - 8 * c) : 8;
            for (int e = 0; e < qend; e++) {
                const int q = 8 * c + e;
                const bool free_q = act && !((mword >> q) & 1ull);
                const double xq = x[c][e];
                if (METHOD == 1) {
                    double tmp = xq - v[c][e] / Gs[q * KP8 + q];
                    if (tmp < 0) tmp = 0;
                    const bool upd = free_q && (tmp != xq);
                    const double d = upd ? tmp - xq : 0.0;
#pragma unroll
                    for (int c2 = 0; c2 < NCH; c2++)
#pragma unroll
                        for (int e2 = 0; e2 < 8; e2++) v[c2][e2] = __builtin_fma(d, Gs[q * KP8 + 8 * c2 + e2], v[c2][e2]);
                    if (upd) {
                        const double er = 2 * fabs(xq - tmp) / (tmp + xq + NNLM_TINY);
                        if (er > rel) rel = er;
                    }
                    x[c][e] = upd ? tmp : xq;
                } else {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int c2 = 0; c2 < NCH; c2++)
#pragma unroll
                        for (int e2 = 0; e2 < 8; e2 += 2) {
                            s0 = __builtin_fma(Gs[q * KP8 + 8 * c2 + e2], x[c2][e2], s0);
                            s1 = __builtin_fma(Gs[q * KP8 + 8 * c2 + e2 + 1], x[c2][e2 + 1], s1);
                        }
                    double tmp = (s0 + s1) + a.r2;
                    tmp = v[c][e] / (tmp + NNLM_TINY);
                    if (free_q) {
                        const double er = 2 * fabs(tmp - 1) / (tmp + 1);
                        if (er > rel) rel = er;
                    }
                    x[c][e] = free_q ? xq * tmp : xq;
                }
            }
        }
        if (act) {
            t_lane++;
            act = rel > tol;
        }
        t++;
    }

    if (in_range) {
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int q = 8 * c + e;
                if (q < k) {
                    const double xv = x[c][e];
                    a.X[(size_t)q * a.ldx + col] = xv;
                    if (a.op_mode == 1) {
                        if (a.op_f64) ((double *)a.op)[(size_t)q * a.op_ld + col] = xv;
                        else ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xv;
                    } else if (a.op_mode == 2) {
                        if (a.op_f64) ((double *)a.op)[(size_t)col * a.op_ld + q] = xv;
                        else ((float *)a.op)[(size_t)col * a.op_ld + q] = (float)xv;
                    }
                }
            }
    }
    long long tot = wave_sum_ll((long long)t_lane);
    if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
}
