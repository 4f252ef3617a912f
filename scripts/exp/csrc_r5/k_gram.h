// k_gram.h -- k x k Gram matrix of the fixed factor of a half-step, in fp64.
// Reference: `WtW = Wt * Wt.t()` at the top of update(), src/update_with_missing.cpp:19
// (the regularisation edits of :20-24 are applied where G is consumed, see k_sweep.h).
//
//   gram_partial : block b contracts 256 columns of X ([KP][ld], fp64 master copy, zero padded)
//                  with v_mfma_f64_16x16x4_f64 and writes one KP x KP slab (upper tiles only);
//   gram_reduce  : sums the slabs in a fixed order and mirrors the lower tiles.
// Work is O(k^2 n) = 1e8 flops at config 2 -- off the roofline; these kernels only have to be short.
#pragma once
#include "common.h"
#include "k_sweep.h"

#define GRAM_COLS_PER_BLOCK 256

// maxbits != NULL: also atomicMax the bit pattern of max|x| (as float) over everything the block reads -- the scale of the
// split-fp16 copy of the same factor (k_xprod16.h) then needs no pass of its own.
template <int NKQ>
__global__ __launch_bounds__(256) void gram_partial_kernel(const double *__restrict__ X, int ld, int c_begin, int c_end,
                                                           double *__restrict__ slabs, unsigned *__restrict__ maxbits = nullptr)
{
    constexpr int KP = 16 * NKQ;
    __shared__ double red[KP * KP]; // waves 3,2,1 fold their tiles here in turn (fixed order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int c0 = c_begin + blockIdx.x * GRAM_COLS_PER_BLOCK + wave * 64;

    f64x4 acc[NKQ][NKQ];
#pragma unroll
    for (int a = 0; a < NKQ; a++)
#pragma unroll
        for (int b = 0; b < NKQ; b++) acc[a][b] = f64x4{0, 0, 0, 0};

    float mx = 0.0f;
    // 8 columns per step: lane holds X[16t + l15][c + 2*lg + e], e = 0,1
    for (int c = c0; c < c0 + 64 && c < c_end; c += 8) {
        f64x2 x[NKQ];
#pragma unroll
        for (int t = 0; t < NKQ; t++) x[t] = *(const f64x2 *)(X + (size_t)(16 * t + l15) * ld + c + 2 * lg);
        if (maxbits) {
#pragma unroll
            for (int t = 0; t < NKQ; t++) mx = fmaxf(mx, fmaxf(fabsf((float)x[t][0]), fabsf((float)x[t][1])));
        }
        if (c + 8 > c_end) { // ragged end of a slab range (multi-GPU split): drop columns >= c_end
#pragma unroll
            for (int t = 0; t < NKQ; t++) {
                if (c + 2 * lg >= c_end) x[t][0] = 0.0;
                if (c + 2 * lg + 1 >= c_end) x[t][1] = 0.0;
            }
        }
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int a = 0; a < NKQ; a++)
#pragma unroll
                for (int b = a; b < NKQ; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[a][e], x[b][e], acc[a][b], 0, 0, 0);
    }
    if (maxbits) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0 && mx > 0.0f) atomicMax(maxbits, __float_as_uint(mx));
    }
    // f64 C/D layout: reg r -> row lg + 4r, col l15
    for (int w = 3; w >= 1; --w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < NKQ; a++)
#pragma unroll
                for (int b = a; b < NKQ; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int idx = (16 * a + lg + 4 * r) * KP + 16 * b + l15;
                        red[idx] = (w == 3) ? acc[a][b][r] : red[idx] + acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    if (wave == 0) {
        double *out = slabs + (size_t)blockIdx.x * KP * KP;
#pragma unroll
        for (int a = 0; a < NKQ; a++)
#pragma unroll
            for (int b = a; b < NKQ; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int idx = (16 * a + lg + 4 * r) * KP + 16 * b + l15;
                    out[idx] = red[idx] + acc[a][b][r];
                }
    }
}

// G = sum of MANY slabs (one per workgroup of the fast sweep kernel, k_sweep_q.h: hundreds), no fences, fixed order:
// a block owns 64 consecutive entries, its 16 wavefronts add every 16th slab (coalesced 512-byte reads), LDS folds the
// 16 partial sums in index order.  Launch with KP*KP/64 blocks of 1024 threads.  Lower tiles are written as mirrors.
__device__ static inline void gram_fold_body(const double *__restrict__ slabs, int nslabs, int KP, double *__restrict__ G, int blk,
                                             const SweepImg &im)
{
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blk * 64 + lane;
    const int a = e / KP, b = e % KP;
    const bool upper = (a >> 4) <= (b >> 4); // the slabs hold upper tiles only (lanes of lower tiles idle)
    double s = 0.0;
    if (upper) {
#pragma unroll 4
        for (int i = w; i < nslabs; i += 16) s += slabs[(size_t)i * KP * KP + e];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && upper) {
        double t = part[0][lane];
#pragma unroll
        for (int j = 1; j < 16; j++) t += part[j][lane];
        G[e] = t;
        if ((a >> 4) < (b >> 4)) G[(size_t)b * KP + a] = t;
        if (im.img) {
            sweepq_img_put(im, a, b, t);
            if ((a >> 4) < (b >> 4)) sweepq_img_put(im, b, a, t);
        }
    }
}
// im.img != NULL: also the SCD sweep's operand image (no sweepq_pack_kernel launch)
__global__ __launch_bounds__(1024) void gram_fold_kernel(const double *__restrict__ slabs, int nslabs, int KP, double *__restrict__ G, const SweepImg im)
{
    gram_fold_body(slabs, nslabs, KP, G, blockIdx.x, im);
}

// The same fold for the Gram partial sums that travel behind a rank's packed slab (multi-GPU, dense SCD, column form), plus the
// rank's max|x| -- left by its sweep as the bit pattern of a float in *maxword -- as ONE more double behind them (G[KP * KP]);
// the word is cleared for the next sweep.  The unpack on every rank then knows max|factor| BEFORE it reads a single entry and
// writes the split-fp16 copy itself (shard_unpack_kernel).
__global__ __launch_bounds__(1024) void gram_fold_tail_kernel(const double *__restrict__ slabs, int nslabs, int KP, double *__restrict__ G,
                                                              unsigned *__restrict__ maxword)
{
    gram_fold_body(slabs, nslabs, KP, G, blockIdx.x, SweepImg{});
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        G[(size_t)KP * KP] = (double)__uint_as_float(*maxword);
        *maxword = 0u;
    }
}

// G[a][b] = sum over slabs (fixed order); entries of lower tiles are read from the mirrored upper tile.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double *__restrict__ slabs, int nslabs, int KP,
                                                          double *__restrict__ G)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= KP * KP) return;
    int a = idx / KP, b = idx % KP;
    if ((a >> 4) > (b >> 4)) {
        const int t = a;
        a = b;
        b = t;
    }
    const size_t src = (size_t)a * KP + b;
    // eight loads in flight, summed in slab order (the same sum as the one-load-at-a-time loop, ~4x sooner: the loop is load latency)
    const size_t st = (size_t)KP * KP;
    const double *q = slabs + src;
    double s = 0.0;
    int i = 0;
    for (; i + 8 <= nslabs; i += 8, q += 8 * st) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = q[u * st];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; i < nslabs; i++, q += st) s += *q;
    G[idx] = s;
}

// out[e] = sum_s slabs[s][e] (fixed order): folds the split-K slabs of the cross product into the contiguous
// buffer that one RCCL all-reduce sums over ranks.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const double *__restrict__ slabs, int nslabs, size_t cnt,
                                                          double *__restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= cnt) return;
    double s = 0.0;
    for (int i = 0; i < nslabs; i++) s += slabs[(size_t)i * cnt + e];
    out[e] = s;
}

// Multi-GPU: scatter the all-gathered per-rank slabs of the updated factor back into the resident layouts.
// packed: per rank a [KPt][cpr] slab (KPt = k: the padding rows of a slab are not gathered) + its tail, rank_stride doubles apart;
// rank rr holds columns rr*cpr .. of the factor; X [KP][ldx] master; op = GEMM operand copy
// (op_mode 1: [KP][op_ld] same layout as X, 0: none), element type float or double.
// The grid covers KProws x (nranks * cpr) entries: KProws = k without the split copy below, the padded rank 16 NKQ with it.
// maxw (optional): receives max |x| over the factor as the bit pattern of a float (what absmax_f64_kernel computes), for the
// split-fp16 copies of the NEXT half-step, whose fixed factor this is.
//   tail_max_off == (size_t)-1: found here while the entries are read (atomicMax; the caller zeroed the word);
//   otherwise packed[r * rank_stride + tail_max_off] holds rank r's own max (gram_fold_tail_kernel): max|factor| is known up
//   front, and with Y16 != NULL the kernel also writes the split-fp16 copy of the factor [KProws][plen/64][2][64] (k_xprod16.h),
//   zero padded, and its exponent -- the next half-step then starts with its cross product (no absmax pass, no factor16_kernel).
__device__ static inline int unpack_split16_exponent(float maxabs) // = split16_exponent (k_xprod16.h includes this file)
{
    if (!(maxabs > 0.0f) || maxabs > 3.0e38f) return 0;
    int ex;
    (void)frexpf(maxabs, &ex);
    return 15 - ex;
}
__global__ __launch_bounds__(256) void shard_unpack_kernel(const double *__restrict__ packed, int nranks, int KProws, int cpr, int k,
                                                           int ncols, double *__restrict__ X, int ldx, void *__restrict__ op,
                                                           int op_mode, int op_ld, int op_f64, unsigned *__restrict__ maxw, size_t rank_stride,
                                                           size_t tail_max_off, uint32_t *__restrict__ Y16, int plen, int *__restrict__ exp_out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_rank = (size_t)KProws * cpr;
    bool in_grid = e < per_rank * nranks;
    bool live = in_grid;
    int q = 0, col = 0, rr = 0, c = 0;
    if (in_grid) {
        rr = (int)(e / per_rank);
        q = (int)((e % per_rank) / cpr);
        c = (int)(e % cpr);
        col = rr * cpr + c;
        live = q < k && col < ncols;
    }
    const double v = live ? packed[(size_t)rr * rank_stride + (size_t)q * cpr + c] : 0.0;
    if (tail_max_off != (size_t)-1) {
        float mx = 0.0f;
        for (int r = 0; r < nranks; r++) mx = fmaxf(mx, (float)packed[(size_t)r * rank_stride + tail_max_off]);
        if (e == 0) {
            if (maxw) *maxw = __float_as_uint(mx);
            if (exp_out) *exp_out = unpack_split16_exponent(mx);
        }
        if (Y16 && in_grid && col < plen) {
            const float x = (float)v * ldexpf(1.0f, unpack_split16_exponent(mx));
            const _Float16 hi = (_Float16)x, lo = (_Float16)((x - (float)hi) * 2048.0f); // split16()
            _Float16 *row = (_Float16 *)(Y16 + (size_t)q * plen + (size_t)(col >> 6) * 64);
            row[col & 63] = hi;
            row[64 + (col & 63)] = lo;
        }
    } else if (maxw) { // (whole workgroups stay together for the reduction: ONE atomic per workgroup -- one per wavefront, 15625 of
                       //  them on one word for the 20000 x 50 factor, took 0.14 ms of a 0.16 ms launch)
        __shared__ float wmx[4];
        float mx = fabsf((float)v);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            mx = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
            // (the word only grows: a workgroup whose maximum is not above what is already there has nothing to add -- after the first
            //  few workgroups hardly any atomic is left)
            if (mx > 0.0f && __float_as_uint(mx) > __hip_atomic_load(maxw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxw, __float_as_uint(mx));
        }
    }
    if (!live) return;
    X[(size_t)q * ldx + col] = v;
    if (op_mode == 1) {
        if (op_f64) ((double *)op)[(size_t)q * op_ld + col] = v;
        else ((float *)op)[(size_t)q * op_ld + col] = (float)v;
    }
}

// Multi-GPU, dense SCD: every rank's sweep leaves the Gram partial sums of ITS columns behind (k_sweep_q.h's epilogue, folded into
// one KP x KP matrix behind its packed slab); after the all-gather the Gram of the whole factor -- the fixed factor of the next
// half-step -- is their sum in rank order: G[e] = sum_r packed[r * rank_stride + tail_off + e].  Identical on every rank.
__global__ __launch_bounds__(256) void shard_gram_sum_kernel(const double *__restrict__ packed, int nranks, size_t rank_stride, size_t tail_off,
                                                             int cnt, double *__restrict__ G)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= cnt) return;
    double s = 0.0;
    for (int r = 0; r < nranks; r++) s += packed[(size_t)r * rank_stride + tail_off + e];
    G[e] = s;
}
