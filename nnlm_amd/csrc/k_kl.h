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

// ------------------------------------------------------------------------------------------------------------------
// NNLM_PREC_F32 variant.  Same sequence of coordinate updates as above; what changes is the arithmetic of the O(p) part
// (the mode already stores A in fp32) and the amount of cached-operand traffic:
//   * the state vector y and the data column b are fp32 registers; the quotients b/(y+eps), w/(y+eps) use
//     v_rcp_f32 (1 ulp) instead of an fp64 division (~20 instructions at 8 issue cycles each);
//   * rows of the fixed factor are read from its fp32 GEMM-operand copy (Yf), half the bytes of the fp64 master;
//   * a block solves C = 2 adjacent columns, so that every row of the fixed factor fetched from L2 serves two
//     columns (the row traffic, ncols*k*p*4/C bytes per sweep, is what bounds this kernel);
//   * per-thread partial sums are fp32 over EPT terms, then fp64 across the block; the scalar coordinate update
//     (:128-150 above) stays fp64.
struct KlFastArgs {
    KlArgs a;
    const float *Yf; // [KP][ldyf] fp32 copy of the fixed factor, contraction index fastest
    int ldyf;
    const float *Yinit; // starting state vectors of all columns in the layout of A (wh_store_kernel); NULL: k passes over Yf
};

template <int EPT, int METHOD, int C>
__global__ __launch_bounds__(KL_THREADS) void kl_fast_kernel(const KlFastArgs fa)
{
    const KlArgs &a = fa.a;
    constexpr int NV = (METHOD == 4) ? 2 : 3; // sums per column: {num, sumW} or {a, b, sumW}
    __shared__ double xs[C][64];
    __shared__ double red[2 * NV * C * 8];
    const int tid = threadIdx.x;
    const int col0 = blockIdx.x * C;
    const int k = a.k, p = a.p;
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    const float tiny = (float)NNLM_TINY;

    unsigned long long mword[C];
    bool live[C]; // column exists and is not fully masked
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int col = col0 + c;
        mword[c] = (a.mask && col < a.ncols) ? a.mask[col] : 0ull;
        live[c] = col < a.ncols && !(a.mask && ((mword[c] & kmask) == kmask));
        if (tid < 64) xs[c][tid] = (tid < k && col < a.ncols) ? a.X[(size_t)tid * a.ldx + col] : 0.0;
    }
    __syncthreads();

    float y[C][EPT], b[C][EPT];
    unsigned long long vbits[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int col = (col0 + c < a.ncols) ? col0 + c : col0;
        const float *Acol = (const float *)a.A + (size_t)col * a.a_col_stride;
        vbits[c] = 0ull;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int i = e * KL_THREADS + tid;
            bool valid = i < p && col0 + c < a.ncols;
            if (valid && a.bits) valid = !((a.bits[(size_t)col * a.words + (i >> 5)] >> (i & 31)) & 1u);
            b[c][e] = valid ? Acol[(size_t)i * a.a_i_stride] : 0.0f;
            if (valid) vbits[c] |= (1ull << e);
            y[c][e] = 0.0f;
        }
    }
    double S[C];
#pragma unroll
    for (int c = 0; c < C; c++) S[c] = 0.0;
    if (fa.Yinit) { // y = Yt^T x was formed for all columns at once by wh_store_kernel
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int col = (col0 + c < a.ncols) ? col0 + c : col0;
            const float *Ycol = fa.Yinit + (size_t)col * a.a_col_stride;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int i = e * KL_THREADS + tid;
                y[c][e] = ((vbits[c] >> e) & 1ull) ? Ycol[(size_t)i * a.a_i_stride] : 0.0f;
            }
            for (int q = 0; q < k; q++) S[c] += xs[c][q];
        }
    } else
        for (int q = 0; q < k; q++) { // y = Yt^T x, S = sum(x)
            float w[EPT];
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int i = e * KL_THREADS + tid;
                w[e] = (i < p) ? fa.Yf[(size_t)q * fa.ldyf + i] : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < C; c++) {
                const double xq = xs[c][q];
                S[c] += xq;
                const float xf = (float)xq;
#pragma unroll
                for (int e = 0; e < EPT; e++) y[c][e] = __builtin_fmaf(((vbits[c] >> e) & 1ull) ? w[e] : 0.0f, xf, y[c][e]);
            }
        }

    double rel[C];
    unsigned tdone[C];
    bool run[C];
#pragma unroll
    for (int c = 0; c < C; c++) rel[c] = 1.0 + a.rel_tol, tdone[c] = 0, run[c] = live[c] && a.max_iter > 0 && rel[c] > a.rel_tol;
    int par = 0;
    bool any = false;
#pragma unroll
    for (int c = 0; c < C; c++) any = any || run[c];
    while (any) { // block-uniform: all state that decides it is computed redundantly by every thread
#pragma unroll
        for (int c = 0; c < C; c++)
            if (run[c]) rel[c] = 0.0;
        for (int q = 0; q < k; q++) {
            bool doq[C];
            bool anyq = false;
#pragma unroll
            for (int c = 0; c < C; c++) doq[c] = run[c] && !((mword[c] >> q) & 1ull), anyq = anyq || doq[c];
            if (!anyq) continue;
            double xq[C];
#pragma unroll
            for (int c = 0; c < C; c++) xq[c] = xs[c][q]; // read BEFORE the reduction's barrier: thread 0 rewrites it after
            float w[EPT];
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int i = e * KL_THREADS + tid;
                w[e] = (i < p) ? fa.Yf[(size_t)q * fa.ldyf + i] : 0.0f;
            }
            double v[NV * C];
#pragma unroll
            for (int c = 0; c < C; c++) {
                float s0 = 0.0f, s1 = 0.0f, sw = 0.0f;
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const float we = ((vbits[c] >> e) & 1ull) ? w[e] : 0.0f;
                    const float r = __builtin_amdgcn_rcpf(y[c][e] + tiny);
                    if (METHOD == 4) {
                        s0 = __builtin_fmaf(we, b[c][e] * r, s0); // Wt.row(k) * (Aj / (wh + eps)), :141
                    } else {
                        const float u = we * r;                   // mu, :97
                        s0 = __builtin_fmaf(b[c][e] * u, u, s0);  // a, :98
                        s1 = __builtin_fmaf(b[c][e], u, s1);      // b, :99
                    }
                    sw += we; // sumW over the same index set
                }
                v[NV * c] = (double)s0;
                if (METHOD != 4) v[NV * c + 1] = (double)s1;
                v[NV * c + NV - 1] = (double)sw;
            }
            kl_block_sum<NV * C>(v, red, par);
            par ^= 1;
#pragma unroll
            for (int c = 0; c < C; c++) {
                if (!doq[c]) continue; // block-uniform
                float coef = 0.0f;
                if (METHOD == 4) {
                    double tmp = v[NV * c] / (v[NV * c + 1] + a.r0 * xq[c] + a.r1 * (S[c] - xq[c]) + a.r2); // :142
                    coef = (float)((tmp - 1) * xq[c]);                                                      // :143
                    S[c] += (tmp - 1) * xq[c];                                                              // :144
                    if (tid == 0) xs[c][q] = xq[c] * tmp;                                                   // :145
                    tmp = 2 * fabs(tmp - 1) / (tmp + 1);
                    if (tmp > rel[c]) rel[c] = tmp;
                } else {
                    double aa = v[NV * c], bb = v[NV * c + 1] - v[NV * c + 2]; // b = dot(Aj, mu) - sumW(k), :99
                    aa += a.r0;                                                // :100
                    bb += aa * xq[c] - a.r2 - a.r1 * (S[c] - xq[c]);           // :101
                    double tmp = bb / (aa + NNLM_TINY);                        // :102
                    if (tmp < 0) tmp = 0;
                    if (tmp != xq[c]) {
                        coef = (float)(tmp - xq[c]);
                        const double er = 2 * fabs(xq[c] - tmp) / (tmp + xq[c] + NNLM_TINY);
                        if (er > rel[c]) rel[c] = er;
                        S[c] += tmp - xq[c];
                        if (tid == 0) xs[c][q] = tmp;
                    }
                }
                if (coef != 0.0f) {
#pragma unroll
                    for (int e = 0; e < EPT; e++) y[c][e] = __builtin_fmaf(coef, ((vbits[c] >> e) & 1ull) ? w[e] : 0.0f, y[c][e]); // :106, :143
                }
            }
        }
        __syncthreads(); // xs[] written by thread 0 during this sweep is read by everyone in the next
        any = false;
#pragma unroll
        for (int c = 0; c < C; c++) {
            if (run[c]) {
                tdone[c]++;
                run[c] = tdone[c] < a.max_iter && rel[c] > a.rel_tol;
            }
            any = any || run[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int col = col0 + c;
        if (col < a.ncols && tid < k) {
            const double xv = xs[c][tid];
            a.Xout[(size_t)tid * a.ldx + col] = xv;
            if (a.op_mode == 1) ((float *)a.op)[(size_t)tid * a.op_ld + col] = (float)xv;
            else if (a.op_mode == 2) ((float *)a.op)[(size_t)col * a.op_ld + tid] = (float)xv;
        }
    }
    if (tid == 0) {
        unsigned long long tot = 0;
#pragma unroll
        for (int c = 0; c < C; c++) tot += tdone[c];
        if (tot) atomicAdd(a.sweeps, tot);
    }
}
