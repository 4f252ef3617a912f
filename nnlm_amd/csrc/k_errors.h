// k_errors.h -- the error/trace block of c_nnmf and the sums add_penalty() needs.
//
// Reference (src/nnmf.cpp:121-126,135-140): Ahat = W.t()*H is materialised (n x m), then
//   mse = mean((A-Ahat)^2),  mkl += mean(-(A+eps) % log(Ahat+eps) + Ahat)   over finite entries of A.
// Here Ahat is never stored: each block forms a 64 x 64 tile of W H with MFMA (contraction k), reads the
// matching tile of the resident A once, and reduces both sums in fp64.  One pass over A (HBM bound at
// large k-independent cost n*m*sizeof(T) bytes) + n*m logs.
#pragma once
#include "common.h"

#define ERR_TILE 64

// partial: [gridDim.y*gridDim.x][2] = {sum (a-ahat)^2, sum -(a+eps)log(ahat+eps)+ahat} over valid entries
template <typename T>
__global__ __launch_bounds__(256) void errors_kernel(const T *__restrict__ A, int lda, const uint32_t *__restrict__ miss,
                                                     const double *__restrict__ W64, int ldw,
                                                     const double *__restrict__ H64, int ldh, int n, int m, int k4,
                                                     double *__restrict__ partial)
{
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int ib = blockIdx.x * ERR_TILE + 32 * (wave & 1);
    const int jb = blockIdx.y * ERR_TILE + 32 * (wave >> 1);

    acc_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = acc_t{0, 0, 0, 0};

    for (int kq = lg; kq < k4; kq += 4) {
        T wa[2], hb[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            wa[t] = (T)W64[(size_t)kq * ldw + ib + 16 * t + l15];
            hb[t] = (T)H64[(size_t)kq * ldh + jb + 16 * t + l15];
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = M::mma(wa[a], hb[b], acc[a][b]);
    }

    double s2 = 0.0, skl = 0.0;
    const int words = lda >> 5;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int j = jb + 16 * b + l15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = ib + 16 * a + M::row_of(lane, r);
                bool valid = (i < n) && (j < m);
                if (miss && valid) valid = !((miss[(size_t)j * words + (i >> 5)] >> (i & 31)) & 1u);
                const T av = A[(size_t)j * lda + i];
                const T ah = acc[a][b][r];
                if (valid) {
                    const T d = av - ah;
                    s2 += (double)d * (double)d;
                    T lg_;
                    if constexpr (sizeof(T) == 4) lg_ = logf(ah + (T)NNLM_TINY);
                    else lg_ = log(ah + (T)NNLM_TINY);
                    skl += (double)(-(av + (T)NNLM_TINY) * lg_ + ah);
                }
            }
        }
    __shared__ double red[2][4];
    s2 = wave_sum(s2);
    skl = wave_sum(skl);
    if (lane == 0) {
        red[0][wave] = s2;
        red[1][wave] = skl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        partial[2 * blk + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
}

// Penalty sums for one factor X [KP][ld] (src/nnmf.cpp:224-240):
//   partial[blk] = {sum x^2, sum_col (sum_q x[q,col])^2, sum x} over the block's 256 columns.
__global__ __launch_bounds__(256) void penalty_kernel(const double *__restrict__ X, int ld, int ncols, int k,
                                                      double *__restrict__ partial)
{
    const int col = blockIdx.x * 256 + threadIdx.x;
    double sq = 0.0, cs = 0.0;
    if (col < ncols)
        for (int q = 0; q < k; q++) {
            const double v = X[(size_t)q * ld + col];
            sq += v * v;
            cs += v;
        }
    double v3[3] = {sq, cs * cs, cs};
    __shared__ double red[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        v3[c] = wave_sum(v3[c]);
        if (lane == 0) red[c][wave] = v3[c];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        partial[3 * (size_t)blockIdx.x + c] = ((red[c][0] + red[c][1]) + red[c][2]) + red[c][3];
    }
}

// out[c] = sum_b partial[b*width + c], fixed order (one block of 256 threads).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ partial, size_t nblocks, int width,
                                                              double *__restrict__ out)
{
    __shared__ double red[256];
    for (int c = 0; c < width; c++) {
        double s = 0.0;
        for (size_t b = threadIdx.x; b < nblocks; b += 256) s += partial[b * width + c];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[c] = red[0];
        __syncthreads();
    }
}
