// k_kl.h -- per-column solvers for the KL-divergence methods, fp64.
//
// Reference: scd_kl_update (src/base_algorithms.cpp:71-116) and lee_kl_update (src/base_algorithms.cpp:119-151) as
// called from update() (src/update_with_missing.cpp:47-49) and, with the contraction restricted to the finite
// entries of the column, from update_with_missing() (src/update_with_missing.cpp:118-133).
//
// One 512-thread block per column j of the factor being solved.  The length-p state vector of the column
// (y = Yt^T x, "Ajt"/"wh" in the reference) and the data column b stay in registers, EPT elements per thread;
// the k coordinates are visited sequentially (loop-carried, exactly as in the reference) and each visit is
//   one coalesced read of row q of the fixed factor  ->  per-thread partial sums  ->  block reduction
//   ->  the scalar update (computed redundantly by every thread)  ->  rank-1 refresh of y in registers.
// Missing entries (NA path) simply carry weight 0 (they are absent from Wt.cols(non_missing) in the
// reference); sumW is summed over the same index set inside the same reduction (src/update_with_missing.cpp:122,130).
// Arithmetic is n*m*k fp64 divides per sweep: VALU/transcendental bound, not HBM bound.
#pragma once
#include "common.h"

#define KL_THREADS 512
#define KL_MAX_P (KL_THREADS * 64)

struct KlArgs {
    const double *X; // [KP][ldx] master of the factor being solved, read
    double *Xout;    // same layout, written (may alias X)
    int ldx;
    const double *Y; // [KP][ldy] master of the fixed factor (contraction index fastest)
    int ldy;
    const void *A;   // resident matrix; element (contraction i, column c) at A[c*a_col_stride + i*a_i_stride]
    size_t a_col_stride, a_i_stride;
    const uint32_t *bits; // missing mask of column c over the contraction index: bits[c*words + i/32] >> (i%32); NULL = none
    int words;
    int p;           // contraction length
    int ncols, k;
    double r0, r1, r2;
    const unsigned long long *mask;
    unsigned max_iter;
    double rel_tol;
    void *op;
    int op_mode, op_ld, op_f64;
    unsigned long long *sweeps;
};

// block-wide sum of NV values, identical result in every thread; `red` is [2][NV][8] doubles, `par` alternates 0/1
template <int NV>
__device__ static inline void kl_block_sum(double (&v)[NV], double *red, int par)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NV; c++) {
        v[c] = wave_sum(v[c]);
        if (lane == 0) red[(par * NV + c) * 8 + wave] = v[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NV; c++) {
        const double *r = red + (par * NV + c) * 8;
        v[c] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    }
}

template <typename T, int EPT, int METHOD>
__global__ __launch_bounds__(KL_THREADS) void kl_update_kernel(const KlArgs a)
{
    __shared__ double xs[64];
    __shared__ double red[2 * 3 * 8];
    const int tid = threadIdx.x;
    const int col = blockIdx.x;
    const int k = a.k, p = a.p;

    unsigned long long mword = 0ull;
    if (a.mask) mword = a.mask[col];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    const bool skipcol = a.mask && ((mword & kmask) == kmask); // all coordinates masked: 0 sweeps, values copied through

    if (tid < 64) xs[tid] = (tid < k) ? a.X[(size_t)tid * a.ldx + col] : 0.0;
    __syncthreads();

    double y[EPT];
    T b[EPT];
    unsigned long long vbits = 0ull; // bit e: element e of this thread takes part
    const T *Acol = (const T *)a.A + (size_t)col * a.a_col_stride;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i = e * KL_THREADS + tid;
        bool valid = i < p;
        if (valid && a.bits) valid = !((a.bits[(size_t)col * a.words + (i >> 5)] >> (i & 31)) & 1u);
        b[e] = valid ? Acol[(size_t)i * a.a_i_stride] : (T)0;
        if (valid) vbits |= (1ull << e);
        y[e] = 0.0;
    }
    double S = 0.0;
    for (int q = 0; q < k; q++) { // y = Yt^T x, S = sum(x)
        const double xq = xs[q];
        S += xq;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int i = e * KL_THREADS + tid;
            const double w = ((vbits >> e) & 1ull) ? a.Y[(size_t)q * a.ldy + i] : 0.0;
            y[e] = __builtin_fma(w, xq, y[e]);
        }
    }

    double rel = 1.0 + a.rel_tol;
    unsigned t = 0;
    int par = 0;
    for (; !skipcol && t < a.max_iter && rel > a.rel_tol; t++) {
        rel = 0.0;
        for (int q = 0; q < k; q++) {
            if ((mword >> q) & 1ull) continue;
            const double xq = xs[q]; // read BEFORE the reduction's barrier: thread 0 rewrites xs[q] after it
            double w[EPT];
            double v[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int i = e * KL_THREADS + tid;
                w[e] = ((vbits >> e) & 1ull) ? a.Y[(size_t)q * a.ldy + i] : 0.0;
                if (METHOD == 4) {
                    v[0] += w[e] * ((double)b[e] / (y[e] + NNLM_TINY)); // Wt.row(k) * (Aj / (wh + eps)), :141
                } else {
                    const double u = w[e] / (y[e] + NNLM_TINY);          // mu, :97
                    v[0] += (double)b[e] * (u * u);                      // a, :98
                    v[1] += (double)b[e] * u;                            // b, :99
                }
                v[2] += w[e]; // sumW over the same index set
            }
            kl_block_sum<3>(v, red, par);
            par ^= 1;
            if (METHOD == 4) {
                double tmp = v[0] / (v[2] + a.r0 * xq + a.r1 * (S - xq) + a.r2); // :142
                const double c = (tmp - 1) * xq;                                  // :143
#pragma unroll
                for (int e = 0; e < EPT; e++) y[e] = __builtin_fma(c, w[e], y[e]);
                S += (tmp - 1) * xq; // :144
                if (tid == 0) xs[q] = xq * tmp; // :145
                tmp = 2 * fabs(tmp - 1) / (tmp + 1);
                if (tmp > rel) rel = tmp;
            } else {
                double aa = v[0], bb = v[1] - v[2];          // b = dot(Aj, mu) - sumW(k), :99
                aa += a.r0;                                  // :100
                bb += aa * xq - a.r2 - a.r1 * (S - xq);      // :101
                double tmp = bb / (aa + NNLM_TINY);          // :102
                if (tmp < 0) tmp = 0;
                if (tmp != xq) {
                    const double d = tmp - xq;
#pragma unroll
                    for (int e = 0; e < EPT; e++) y[e] = __builtin_fma(d, w[e], y[e]); // :106
                    const double er = 2 * fabs(xq - tmp) / (tmp + xq + NNLM_TINY);
                    if (er > rel) rel = er;
                    S += tmp - xq;
                    if (tid == 0) xs[q] = tmp;
                }
            }
        }
        __syncthreads(); // xs[] written by thread 0 during this sweep is read by everyone in the next
    }
    __syncthreads();
    if (tid < k) {
        const double xv = xs[tid];
        a.Xout[(size_t)tid * a.ldx + col] = xv;
        if (a.op_mode == 1) {
            if (a.op_f64) ((double *)a.op)[(size_t)tid * a.op_ld + col] = xv;
            else ((float *)a.op)[(size_t)tid * a.op_ld + col] = (float)xv;
        } else if (a.op_mode == 2) {
            if (a.op_f64) ((double *)a.op)[(size_t)col * a.op_ld + tid] = xv;
            else ((float *)a.op)[(size_t)col * a.op_ld + tid] = (float)xv;
        }
    }
    if (tid == 0 && t) atomicAdd(a.sweeps, (unsigned long long)t);
}
