// k_generic.h -- the square-loss half-step without a rank limit.
//
// The reference has no limit on the rank: nnmf(k = 80), nnmf(k = 50) plus 15 known profiles (K = k + ncol(W0) + nrow(H0),
// R/misc.R:84) and nnlm(x, y) with any number of predictors (src/nnlm.cpp:44-47 runs update() with "rank" = ncol(x)) all
// go through the same update() / update_with_missing() (src/update_with_missing.cpp:3-139).  The tuned kernels of this
// library hold a column's coordinates in one 64-lane wavefront / four 16-wide MFMA tiles; beyond 64 coordinates a half-step
// is assembled from the pieces below instead.  They are written for generality (any K that fits the device), in the
// reference's arithmetic (fp64, true divisions), not for speed:
//   gram_partial_generic_kernel   G = Y Y^T, one upper 16x16 tile pair per blockIdx.y, fp64 MFMA, slabs folded by gram_reduce_kernel
//   (cross products)              the A-streaming MFMA kernels of k_xprod.h / k_xprod16.h, launched once per 64 rows of the factor
//   na_gram_generic_kernel        per-column Gram over the row lists of k_missing.h (update_with_missing, :90)
//   sweep_generic_kernel          scd_ls_update / lee_ls_update (src/base_algorithms.cpp:3-68): one wavefront per column, the
//                                 column's x and mu in LDS, a Gram shared by all columns in LDS when it fits, else read from L2
#pragma once
#include "common.h"
#include "k_sweep.h"

// grid (column blocks of 256, NT*(NT+1)/2 upper tile pairs); slabs [gridDim.x][KP*KP] (upper tiles), KP = 16*NT
__global__ __launch_bounds__(256) void gram_partial_generic_kernel(const double *__restrict__ X, int ld, int c_begin, int c_end, int NT,
                                                                   double *__restrict__ slabs)
{
    __shared__ double red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    int ta = 0, tb = blockIdx.y; // pair index -> (ta, tb), ta <= tb
    while (tb >= NT - ta) {
        tb -= NT - ta;
        ta++;
    }
    tb += ta;
    const int KP = 16 * NT;
    const int c0 = c_begin + blockIdx.x * 256 + wave * 64;
    f64x4 acc = {0, 0, 0, 0};
    for (int c = c0; c < c0 + 64 && c < c_end; c += 4) {
        const int cc = c + lg;
        const double xa = (cc < c_end) ? X[(size_t)(16 * ta + l15) * ld + cc] : 0.0;
        const double xb = (cc < c_end) ? X[(size_t)(16 * tb + l15) * ld + cc] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xb, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) red[wave][(lg + 4 * r) * 16 + l15] = acc[r];
    __syncthreads();
    const int e = threadIdx.x; // entry (e / 16, e % 16) of the tile
    const double s = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    slabs[(size_t)blockIdx.x * KP * KP + (size_t)(16 * ta + e / 16) * KP + 16 * tb + (e % 16)] = s;
}

// Per-column Gram over the listed rows (k_missing.h: ptr / meta / idx; the listed rows are the missing ones when meta's
// top bit is set: G_j = G_full - sum).  One 256-thread block per column; entries of the KP x KP result are dealt to the
// threads in passes of 256 x 8; the listed rows pass through LDS 16 at a time.
__global__ __launch_bounds__(256) void na_gram_generic_kernel(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ meta, const int *__restrict__ idx,
                                                              const double *__restrict__ Yrow, int KP, const double *__restrict__ Gfull,
                                                              double *__restrict__ Gcols, int col0 = 0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char nag_smem[];
    double *rows = (double *)nag_smem; // [16][KP]
    const int col = col0 + blockIdx.x, tid = threadIdx.x;
    const uint32_t mt = meta[col];
    const int len = (int)(mt & 0x7FFFFFFFu);
    const bool complement = (mt >> 31) != 0;
    const int *list = idx + ptr[col];
    const int total = KP * KP;
    double *out = Gcols + (size_t)col * total;
    for (int e0 = 0; e0 < total; e0 += 256 * 8) {
        double acc[8];
        int ea[8], eb[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc[u] = 0.0;
            const int e = e0 + u * 256 + tid;
            ea[u] = (e < total) ? e / KP : 0;
            eb[u] = (e < total) ? e % KP : 0;
        }
        for (int r0 = 0; r0 < len; r0 += 16) {
            const int nb = (len - r0 < 16) ? len - r0 : 16;
            __syncthreads();
            for (int t = tid; t < nb * KP; t += 256) rows[t] = Yrow[(size_t)list[r0 + t / KP] * KP + (t % KP)];
            __syncthreads();
            for (int r = 0; r < nb; r++) {
#pragma unroll
                for (int u = 0; u < 8; u++) acc[u] = __builtin_fma(rows[r * KP + ea[u]], rows[r * KP + eb[u]], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * 256 + tid;
            if (e < total) out[e] = complement ? Gfull[e] - acc[u] : acc[u];
        }
    }
}

// One wavefront per column, four columns per 256-thread block.  Dynamic LDS: [the edited G, KP*KP doubles, when all columns
// share one Gram (g_stride = 0) and g_in_lds] + per wavefront x[k], mu[k] (method 1) or c[k] (method 2), diag[k].
// Per-column Grams (g_stride = KP*KP: missing values) and Grams too large for LDS are read from global memory (L2) at
// every coordinate.  mask: mw 64-bit words per column.  A wavefront only synchronises with itself (its own LDS arrays):
// LDS instructions of one wavefront execute in order, the fence keeps the compiler from reordering them.
__host__ __device__ static inline size_t sweep_generic_lds_bytes(int k, int KP, bool g_in_lds) { return (size_t)4 * 3 * k * 8 + (g_in_lds ? (size_t)KP * KP * 8 : 0); }
#define SG_WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

template <int METHOD>
__global__ __launch_bounds__(256) void sweep_generic_kernel(const SweepArgs a, size_t g_stride, int mw, int g_in_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = a.k, KP = a.KPg;
    const int col = a.col0 + blockIdx.x * 4 + wave;
    double *Gs = (double *)sg_smem; // [KP][KP] edited G (g_in_lds)
    double *xs = Gs + (g_in_lds ? (size_t)KP * KP : 0) + (size_t)wave * 3 * k; // [k]
    double *vs = xs + k;                                                       // [k]  mu (method 1) / c (method 2)
    double *gd = vs + k;                                                       // [k]  edited diagonal
    const bool in_range = col < a.ncols;
    const double *Graw = a.Graw + (size_t)(in_range ? col : a.col0) * g_stride;
    auto edited = [&](int r, int c) -> double { // src/update_with_missing.cpp:20-24 / :98-103
        double g = Graw[(size_t)r * KP + c];
        if (r == c && a.r0 != a.r1) g += a.r0 - a.r1;
        if (a.r1 != 0) g += a.r1;
        if (r == c) g += NNLM_TINY;
        return g;
    };
    if (g_in_lds) {
        for (int e = threadIdx.x; e < KP * KP; e += 256) {
            const int r = e / KP, c = e % KP;
            Gs[e] = (r < k && c < k) ? edited(r, c) : 0.0;
        }
        __syncthreads();
    }
    if (!in_range) return; // whole wavefront
    const unsigned long long *mrow = a.mask ? a.mask + (size_t)col * mw : nullptr;
    bool skip = false;
    if (mrow) { // arma::all(mask.col(j)), src/update_with_missing.cpp:33
        skip = true;
        for (int w = 0; w < mw; w++) {
            const int nb = (k - 64 * w >= 64) ? 64 : k - 64 * w;
            const unsigned long long km = (nb >= 64) ? ~0ull : ((1ull << nb) - 1ull);
            skip = skip && ((mrow[w] & km) == km);
        }
    }
    for (int q = lane; q < k; q += 64) {
        xs[q] = a.X[(size_t)q * a.ldx + col];
        gd[q] = edited(q, q);
        double cv = 0.0;
        for (int s = 0; s < a.nslabs; s++) cv += a.Cx[(size_t)s * a.slab_stride + (size_t)q * a.ldc + col];
        vs[q] = cv;
    }
    SG_WAVE_SYNC();
    auto grow = [&](int q, int c) -> double { return g_in_lds ? Gs[q * KP + c] : edited(q, c); };
    if (METHOD == 1) { // mu = G x - c (+ L1), src/update_with_missing.cpp:39-41
        for (int c = lane; c < k; c += 64) {
            double s = 0.0;
            for (int q = 0; q < k; q++) s = __builtin_fma(grow(q, c), xs[q], s); // G symmetric: column c = row c
            s -= vs[c];
            if (a.r2 != 0) s += a.r2;
            vs[c] = s;
        }
        SG_WAVE_SYNC();
    }
    unsigned t = 0;
    if (!skip) {
        double rel = 1.0 + a.rel_tol;
        for (; t < a.max_iter && rel > a.rel_tol; t++) {
            rel = 0.0;
            for (int q = 0; q < k; q++) {
                if (mrow && ((mrow[q >> 6] >> (q & 63)) & 1ull)) continue;
                const double xq = xs[q];
                if (METHOD == 1) { // src/base_algorithms.cpp:21-34
                    double tmp = xq - vs[q] / gd[q];
                    if (tmp < 0) tmp = 0;
                    if (tmp != xq) { // wave-uniform
                        const double d = tmp - xq;
                        SG_WAVE_SYNC();
                        for (int c = lane; c < k; c += 64) vs[c] = __builtin_fma(d, grow(q, c), vs[c]);
                        const double e = 2 * fabs(xq - tmp) / (tmp + xq + NNLM_TINY);
                        if (e > rel) rel = e;
                        if (lane == 0) xs[q] = tmp;
                        SG_WAVE_SYNC();
                    }
                } else { // src/base_algorithms.cpp:57-65
                    double part = 0.0;
                    for (int c = lane; c < k; c += 64) part = __builtin_fma(grow(q, c), xs[c], part);
                    const double dot = wave_sum(part);
                    const double tmp = vs[q] / (dot + a.r2 + NNLM_TINY);
                    SG_WAVE_SYNC();
                    if (lane == 0) xs[q] = xq * tmp;
                    const double e = 2 * fabs(tmp - 1) / (tmp + 1);
                    if (e > rel) rel = e;
                    SG_WAVE_SYNC();
                }
            }
        }
    }
    SG_WAVE_SYNC();
    for (int q = lane; q < k; q += 64) {
        const double xv = xs[q];
        a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xv;
        if (a.op_mode == 1) {
            if (a.op_f64) ((double *)a.op)[(size_t)q * a.op_ld + col] = xv;
            else ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xv;
        }
    }
    if (lane == 0 && t) atomicAdd(a.sweeps, (unsigned long long)t);
}

// T[c][r] = S[r][c] for element type T ([rows][lds] -> [cols][ldt]; rows, cols multiples of 64): the contraction-contiguous
// copy of A that lets the W half-step run the TN cross-product kernel (made once per matrix, on first use)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T *__restrict__ S, int lds_, T *__restrict__ D, int ldt)
{
    __shared__ T tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = S[(size_t)(r0 + r) * lds_ + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 64; c += 4) D[(size_t)(c0 + c) * ldt + r0 + tx] = tile[tx][c];
}
