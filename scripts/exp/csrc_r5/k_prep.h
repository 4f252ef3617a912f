// k_prep.h -- one streaming pass over the uploaded fp64 matrix: convert to the resident element
// type and padded layout, find the non-finite (missing) entries and sum the constant KL part.
// Restates the prologue of c_nnmf, reference src/nnmf.cpp:65-73:
//   any_missing = !A.is_finite(); non_missing = find_finite(A); N_non_missing;
//   mkl_const = mean((A+eps) % log(A+eps) - A) over finite entries.
// HBM-bound: reads 8 B, writes sizeof(T) B (+1 bit) per element.
#pragma once
#include "common.h"

#define PREP_BLOCK 256
#define PREP_GRID_Y 64

// src: chunk of `cols` columns of the caller's matrix (column-major, leading dimension n).
// dst: A_dev (column j0+jj of the chunk lands at (j0+jj)*npad).  miss: bit i%32 of word [j][i/32] set <=> A[i,j] is not finite.
// partial: [gridDim.y*gridDim.x][3] = {finite count, sum((a+eps)log(a+eps)-a), finite entries beyond the range of T} per block.
// (T = float: a finite |a| > FLT_MAX would become +-Inf in the resident copy -- a "missing" value the bit matrix does not know.  The
//  reference's fp64 takes such a matrix; the fp32-operand mode refuses it, nnlm_set_matrix.)
template <typename T>
__global__ __launch_bounds__(PREP_BLOCK) void prep_convert_kernel(const double *__restrict__ src, int n, int cols, int j0,
                                                                  T *__restrict__ dst, int npad,
                                                                  uint32_t *__restrict__ miss,
                                                                  double *__restrict__ partial)
{
    const int i = blockIdx.x * PREP_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double cnt = 0.0, klc = 0.0, over = 0.0;
    const int words = npad >> 5;
    for (int jj = blockIdx.y; jj < cols; jj += gridDim.y) {
        double v = 0.0;
        bool fin = true; // padding rows count as "not missing" and are excluded from the sums below
        if (i < n) {
            v = src[(size_t)jj * n + i];
            fin = isfinite(v);
            if (fin) {
                cnt += 1.0;
                klc += (v + NNLM_TINY) * log(v + NNLM_TINY) - v;
                if (sizeof(T) == 4 && fabs(v) > 3.4028234663852886e38) over += 1.0;
            }
        }
        const size_t col = (size_t)(j0 + jj);
        dst[col * npad + i] = fin ? (T)v : (T)0; // i < npad always: grid.x covers npad exactly
        const unsigned long long b = __ballot(!fin);
        if (lane == 0) {
            const int w = i >> 5;
            miss[col * words + w] = (uint32_t)b;
            miss[col * words + w + 1] = (uint32_t)(b >> 32);
        }
    }
    // block reduction (fixed order -> deterministic)
    __shared__ double red[3][PREP_BLOCK / 64];
    cnt = wave_sum(cnt);
    klc = wave_sum(klc);
    over = wave_sum(over);
    if (lane == 0) {
        red[0][threadIdx.x >> 6] = cnt;
        red[1][threadIdx.x >> 6] = klc;
        red[2][threadIdx.x >> 6] = over;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0, s = 0, o = 0;
        for (int w = 0; w < PREP_BLOCK / 64; w++) {
            c += red[0][w];
            s += red[1][w];
            o += red[2][w];
        }
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[3 * blk] = c;
        partial[3 * blk + 1] = s;
        partial[3 * blk + 2] = o;
    }
}
