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

// ln x for a positive, finite, NORMAL double (the error sums take ln(ahat + 1e-16): never zero, never denormal) in ~28 fp64
// instructions: x = m 2^e with m in [sqrt(1/2), sqrt 2), ln m = 2 atanh(s), s = (m - 1) / (m + 1) (|s| <= 0.1716: ten terms of the
// odd series leave 6e-19), the quotient correctly rounded (reciprocal, two Newton steps, one residual correction).  libm's log()
// compiles to ~55 instructions here and was two thirds of errors_kernel<double>'s arithmetic.  Error < 2 ulp of the result.
__device__ static inline double nnlm_log_pos(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x); // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752;
    m = lo ? 2.0 * m : m;
    e = lo ? e - 1 : e;
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    double sq = num * r;
    sq = __builtin_fma(__builtin_fma(-den, sq, num), r, sq);
    const double z = sq * sq;
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, z, 1.0 / 19.0);
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    // ln m = 2 s + 2 s z p;  e ln 2 in two pieces (the high one has 11 trailing zero bits: e * LN2_HI is exact for |e| < 2048)
    const double ed = (double)e;
    const double t = __builtin_fma(2.0 * sq * z, p, ed * 1.9082149292705877e-10);
    return __builtin_fma(ed, 0.693147180369123816490, 2.0 * sq + t);
}

// partial: [gridDim.y*gridDim.x][2] = {sum (a-ahat)^2, sum -(a+eps)log(ahat+eps)+ahat} over valid entries
template <typename T>
__global__ __launch_bounds__(256) void errors_kernel(const T *__restrict__ A, int lda, const uint32_t *__restrict__ miss,
                                                     const double *__restrict__ W64, int ldw,
                                                     const double *__restrict__ H64, int ldh, int n, int m, int k4,
                                                     double *__restrict__ partial, int jt0)
{
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int ib = blockIdx.x * ERR_TILE + 32 * (wave & 1);
    const int jb = (blockIdx.y + jt0) * ERR_TILE + 32 * (wave >> 1); // jt0: first j-tile of this rank's share (multi-GPU)

    acc_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = acc_t{0, 0, 0, 0};
    // the tile of A first: its 16 loads stay in flight during the MFMA phase.  Accumulator layout: column index N = l15 <-> 16
    // CONSECUTIVE rows i of one column j of A per lane group: 128-byte (fp64) / 64-byte segments.  (Rounds 1-2 had the roles the
    // other way round -- 16 lanes on 16 different columns, 32-byte pieces of 16 cache lines per load -- and loaded after the MFMAs.)
    T av[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                av[a][b][r] = A[(size_t)(jb + 16 * b + M::row_of(lane, r)) * lda + ib + 16 * a + l15];

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
            for (int b = 0; b < 2; b++) acc[a][b] = M::mma(hb[b], wa[a], acc[a][b]); // M = column j, N = row i
    }

    double s2 = 0.0, skl = 0.0;
    const int words = lda >> 5;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int i = ib + 16 * a + l15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = jb + 16 * b + M::row_of(lane, r);
                bool valid = (i < n) && (j < m);
                if (miss && valid) valid = !((miss[(size_t)j * words + (i >> 5)] >> (i & 31)) & 1u);
                const T ah = acc[a][b][r];
                if (valid) {
                    const T d = av[a][b][r] - ah;
                    s2 += (double)d * (double)d;
                    T lg_;
                    if constexpr (sizeof(T) == 4) lg_ = logf(ah + (T)NNLM_TINY);
                    else lg_ = nnlm_log_pos(ah + (T)NNLM_TINY);
                    skl += (double)(-(av[a][b][r] + (T)NNLM_TINY) * lg_ + ah);
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

// ------------------------------------------------------------------------------------------------
// F32 fast path of the error block.  128 x 128 tile of W H per 256-thread block:
//   * the k x 128 slices of the fp32 [kq][col] copies of W and H go to LDS with global_load_lds ([kq][128]), issued FIRST;
//     the wavefront's 64 x 64 block of A (64 dword loads per lane) follows and stays in flight during the MFMA phase:
//     vmcnt retires in order, so waiting until only the A loads are outstanding is waiting for the operands;
//   * each wavefront forms a 64 x 64 block as 2 x 2 v_mfma_f32_32x32x2_f32 tiles with M = j and N = i, so that in the
//     accumulator layout (column = lane & 31) the 32 lanes of a half-wave hold 32 CONSECUTIVE rows i of one column j:
//     the matching A entries are read as full 128-byte lines (the generic kernel above reads 64-byte pieces);
//   * missing entries (HAS_MISS) come from the TRANSPOSED bit matrix missT[i][j/32]: the 32 columns j of an accumulator
//     tile are the 32 bits of ONE word of the lane's row i -- four word loads per lane instead of one per entry; the
//     bounds of an edge tile are folded into the same words, so the sums have one masked and one unmasked form;
//   * blocks are numbered so that an XCD (block id mod 8) always works on the same eighth of the i-tiles: its L2 keeps
//     those W slices (0.5 MB) and the H slice of the current j-tile -- the W/H slices were 0.35 GB of L2 misses per
//     launch against 0.8 GB of A (r01 PMC: 1.44x the algorithmic bytes);
//   * per tile the two sums are accumulated in fp32 over the lane's 16 entries, then folded into fp64.
// partial: [jtiles * nx][2] as above.  Grid: 1-D, 8 * ceil(nx / 8) * jtiles blocks.
// ------------------------------------------------------------------------------------------------
#define ERRF_TILE 128
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool HAS_MISS>
__global__ __launch_bounds__(256) void errors_f32_kernel(const float *__restrict__ A, int lda, const uint32_t *__restrict__ missT, int wordsT,
                                                         const float *__restrict__ Wf, int ldw,
                                                         const float *__restrict__ Hf, int ldh, int n, int m, int k2,
                                                         double *__restrict__ partial, int jt0, int nx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_err[];
    float *Ws = (float *)smem_err;               // [k2][128]
    float *Hs = Ws + (size_t)k2 * ERRF_TILE;      // [k2][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nx + 7) >> 3;                // i-tiles per XCD
    const int bid = blockIdx.x, it = ((bid >> 3) % per) * 8 + (bid & 7), jt = (bid >> 3) / per;
    if (it >= nx) return;
    const int i0 = it * ERRF_TILE, j0 = (jt + jt0) * ERRF_TILE; // jt0: this rank's first j-tile
    const int l31 = lane & 31, lh = lane >> 5;
    const int ib = 64 * (wave & 1), jb = 64 * (wave >> 1);

    // 1. k x 128 slices of W and H (fp32 [kq][col] copies) straight into LDS with global_load_lds: one instruction
    //    moves two 512-byte rows; all of them are in flight at once and use no VGPRs
    {
        const int nrow2 = k2 / 2; // instructions per matrix
        for (int t = wave; t < 2 * nrow2; t += 4) {
            const bool isw = t < nrow2;
            const int tt = isw ? t : t - nrow2;
            const int row = 2 * tt + (lane >> 5);
            const float *src = isw ? (Wf + (size_t)row * ldw + i0 + 4 * (lane & 31)) : (Hf + (size_t)row * ldh + j0 + 4 * (lane & 31));
            glds16(src, (unsigned char *)(isw ? Ws : Hs) + (size_t)tt * 1024);
        }
    }
    asm volatile("" ::: "memory");

    // 2. the HBM reads of this wavefront's 64 x 64 block of A: the first half before the barrier, the second behind it; both
    //    stay in flight during the MFMA phase.  D layout 32x32: row M = (r&3) + 8*(r>>2) + 4*(lane>>5), column N = lane & 31.
    float av[2][2][16];
    uint32_t mw[2][2];
    const bool edge = (i0 + ERRF_TILE > n) || (j0 + ERRF_TILE > m);
    const bool masked = HAS_MISS || edge; // (block uniform)
#pragma unroll
    for (int a = 0; a < 2; a++) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int i = i0 + ib + 32 * b + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int j = j0 + jb + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
                av[a][b][r] = A[(size_t)j * lda + i];
            }
            uint32_t wd = 0;
            if (HAS_MISS) wd = missT[(size_t)i * wordsT + ((j0 + jb + 32 * a) >> 5)];
            if (edge) {
                const int jbase = j0 + jb + 32 * a;
                if (i >= n || jbase >= m) wd = ~0u;
                else if (jbase + 32 > m) wd |= ~0u << (m - jbase);
            }
            mw[a][b] = wd >> (4 * lh); // bit of entry r: (r&3) + 8*(r>>2)
        }
        if (a == 0) {
            // everything issued so far except the 32 loads of A (+ 2 mask words) has landed: the LDS image is complete
            if (HAS_MISS) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }

    // 3. W H for the block: 2 x 2 tiles of 32 x 32, contraction k
    f32x16 acc[2][2]; // [tj][ti]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    for (int q = lh; q < k2; q += 2) {
        float hj[2], wi[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            hj[t] = Hs[(size_t)q * ERRF_TILE + jb + 32 * t + l31];
            wi[t] = Ws[(size_t)q * ERRF_TILE + ib + 32 * t + l31];
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(hj[a], wi[b], acc[a][b], 0, 0, 0);
    }

    // 4. the two sums
    double s2 = 0.0, skl = 0.0;
    if (!masked) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                float p2 = 0.f, pk = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float ah = acc[a][b][r], aa = av[a][b][r];
                    const float d = aa - ah;
                    const float l2 = log2_native(ah + (float)NNLM_TINY);
                    const float cf = __builtin_fmaf(aa, -NNLM_LN2F, -(float)NNLM_TINY * NNLM_LN2F); // -(a + eps) ln 2
                    p2 = __builtin_fmaf(d, d, p2);
                    pk += __builtin_fmaf(cf, l2, ah);
                }
                s2 += (double)p2;
                skl += (double)pk;
            }
    } else {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                float p2 = 0.f, pk = 0.f;
                const uint32_t wd = mw[a][b];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float ah = acc[a][b][r], aa = av[a][b][r];
                    const float d = aa - ah;
                    const float l2 = log2_native(ah + (float)NNLM_TINY);
                    const float cf = __builtin_fmaf(aa, -NNLM_LN2F, -(float)NNLM_TINY * NNLM_LN2F);
                    const bool off = (wd >> ((r & 3) + 8 * (r >> 2))) & 1u;
                    p2 += off ? 0.f : d * d;
                    pk += off ? 0.f : __builtin_fmaf(cf, l2, ah);
                }
                s2 += (double)p2;
                skl += (double)pk;
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
        const size_t blk = (size_t)jt * nx + it;
        partial[2 * blk] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        partial[2 * blk + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
}

// What[j][i] = sum_q W[q][i] H[q][j] in fp32, stored in the layout of A ([mpad][lda]): the starting state vectors
// y = Yt^T x of ALL columns of a KL half-step (src/base_algorithms.cpp:81, :129) as one GEMM instead of k passes over the
// fixed factor per column (k_kl.h).  Same tiling and MFMA phase as errors_f32_kernel.
// Grid: 1-D, 8 * ceil(nx / 8) * ny blocks in the XCD-aware order of errors_f32_kernel (an XCD keeps its eighth of the W slices in L2).
__global__ __launch_bounds__(256) void wh_store_kernel(const float *__restrict__ Wf, int ldw, const float *__restrict__ Hf, int ldh, int k2,
                                                       float *__restrict__ What, int lda, int nx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_err[];
    float *Ws = (float *)smem_err;          // [k2][128]
    float *Hs = Ws + (size_t)k2 * ERRF_TILE; // [k2][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nx + 7) >> 3;
    const int bid = blockIdx.x, it = ((bid >> 3) % per) * 8 + (bid & 7), jt = (bid >> 3) / per;
    if (it >= nx) return;
    const int i0 = it * ERRF_TILE, j0 = jt * ERRF_TILE;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ib = 64 * (wave & 1), jb = 64 * (wave >> 1);
    {
        const int nrow2 = k2 / 2;
        for (int t = wave; t < 2 * nrow2; t += 4) {
            const bool isw = t < nrow2;
            const int tt = isw ? t : t - nrow2;
            const int row = 2 * tt + (lane >> 5);
            const float *src = isw ? (Wf + (size_t)row * ldw + i0 + 4 * (lane & 31)) : (Hf + (size_t)row * ldh + j0 + 4 * (lane & 31));
            glds16(src, (unsigned char *)(isw ? Ws : Hs) + (size_t)tt * 1024);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 acc[2][2]; // [tj][ti]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    for (int q = lh; q < k2; q += 2) {
        float hj[2], wi[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            hj[t] = Hs[(size_t)q * ERRF_TILE + jb + 32 * t + l31];
            wi[t] = Ws[(size_t)q * ERRF_TILE + ib + 32 * t + l31];
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(hj[a], wi[b], acc[a][b], 0, 0, 0);
    }
    // D layout 32x32: row M = (r&3) + 8*(r>>2) + 4*(lane>>5) (a column j), column N = lane & 31 (a row i): 128-byte stores
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int i = i0 + ib + 32 * b + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int j = j0 + jb + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
                What[(size_t)j * lda + i] = acc[a][b][r];
            }
        }
}

// Xf[q][c] = (float) X[q][c]: fp32 [kq][col] copy of a factor for errors_f32_kernel.
__global__ __launch_bounds__(256) void factor_to_f32_kernel(const double *__restrict__ X, size_t count, float *__restrict__ Xf)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < count) Xf[e] = (float)X[e];
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
