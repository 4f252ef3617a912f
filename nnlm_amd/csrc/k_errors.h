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
// (The series coefficients live in constant memory, NOT as literals: gfx950's VOP3 encoding has no 64-bit literal operand, so each
// literal costs a VGPR pair -- 30 registers of a kernel that needs them for occupancy; loaded from a non-const __constant__ array
// they arrive by s_load and stay in SGPR pairs, one scalar source per fma.)
static __constant__ double NNLM_LOGC[19] = {1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0,
                                            1.0 / 5.0,  1.0 / 3.0,  1.9082149292705877e-10, 0.693147180369123816490,
                                            1.0 / 7.0, -1.0 / 6.0, 1.0 / 5.0, -1.0 / 4.0, 1.0 / 3.0, -0.5, 0.69314718055994530942}; // [12..18]: nnlm_log_tab
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
    double p = NNLM_LOGC[0];
#pragma unroll
    for (int c = 1; c < 10; c++) p = __builtin_fma(p, z, NNLM_LOGC[c]);
    // ln m = 2 s + 2 s z p;  e ln 2 in two pieces (the high one has 11 trailing zero bits: e * LN2_HI is exact for |e| < 2048)
    const double ed = (double)e;
    const double t = __builtin_fma(2.0 * sq * z, p, ed * NNLM_LOGC[10]);
    return __builtin_fma(ed, NNLM_LOGC[11], 2.0 * sq + t);
}

// ln x for a positive, finite, NORMAL double from a 64-entry table: x = m 2^e, m in [1, 2), bin i = the top six mantissa bits,
// c_i = 1 + (i + 1/2) / 64, tab[i] = {r_i = fl(1 / c_i), -ln r_i};  t = m r_i - 1 (one fma, exact to rounding, |t| < 2^-7),
// ln m = -ln r_i + log1p(t) with seven terms of the series (t^8 / 8 < 2e-18).  16 vector instructions + one 16-byte LDS read
// against 34 for nnlm_log_pos (no quotient, three series terms less).  ABSOLUTE error ~2e-16 (the table value and e ln 2 are single
// doubles): what a sum of O(1) terms needs -- not the relative accuracy of nnlm_log_pos near x = 1.
__device__ static inline void nnlm_log_tab_fill(f64x2 *tab, int tid)
{
    if (tid < 64) {
        const double r = 1.0 / (1.0 + (tid + 0.5) * (1.0 / 64.0));
        tab[tid] = f64x2{r, -nnlm_log_pos(r)};
    }
}
__device__ static inline double nnlm_log_tab(double x, const f64x2 *tab)
{
    const int2 xb = __builtin_bit_cast(int2, x);
    const int e = (int)(((unsigned)xb.y >> 20) & 0x7ffu) - 1023;
    const int i = (int)(((unsigned)xb.y >> 14) & 63u);
    const double m = __builtin_bit_cast(double, int2{xb.x, (int)(((unsigned)xb.y & 0x000fffffu) | 0x3ff00000u)});
    const f64x2 rl = tab[i];
    const double t = __builtin_fma(m, rl[0], -1.0);
    double p = NNLM_LOGC[12];
    p = __builtin_fma(p, t, NNLM_LOGC[13]);
    p = __builtin_fma(p, t, NNLM_LOGC[14]);
    p = __builtin_fma(p, t, NNLM_LOGC[15]);
    p = __builtin_fma(p, t, NNLM_LOGC[16]);
    p = __builtin_fma(p, t, NNLM_LOGC[17]);
    const double l1p = __builtin_fma(t * t, p, t); // t - t^2/2 + t^3/3 - ... + t^7/7
    return __builtin_fma((double)e, NNLM_LOGC[18], l1p + rl[1]);
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
// Strict fp64 mode, ranks <= 64: the same two sums with the factor slices staged in LDS and the per-entry arithmetic branch free.
// (errors_kernel<double> above: 0.90 ms at config 2 -- its lanes fetched every MFMA operand from global memory, 3.2x the bytes of
// the A tile, and the `if (valid)` regions made the compiler rematerialise the log's 20 constant registers per entry:
// 1880 VALU instructions per wavefront for 16 entries per lane.)
//   * a block of 4 wavefronts owns one 64-row i-tile and walks a chunk of j-tiles: the k x 64 slice of W goes to LDS once, the
//     k x 64 slice of H once per tile (global_load_lds, two 512-byte rows per instruction; the two 128-byte halves of odd rows
//     are exchanged on the SOURCE side so that a half-wave's operand read -- rows kq, kq+1, sixteen consecutive columns each --
//     covers all 64 banks);
//   * per tile: [H slice + A tile landed, barrier] 13 x 4 v_mfma_f64_16x16x4 per wavefront (32 x 32 entries) [barrier] the next
//     H slice and the next A tile are requested, and stay in flight while the 16 entries per lane of this tile are summed;
//   * two (k <= 52: three) blocks per CU drift out of phase: one block's matrix phase runs beside the other's logarithms;
//   * edge tiles and missing entries (MASKED, block-uniform choice per tile) select 0 instead of branching.
// Bound: 2 n m k fp64 matrix flops (0.254 ms at 78.6 TF for config 2) + ~40 fp64 VALU slots per entry (0.2 ms) on the same
// pipes, against 1.6 GB of A (0.2 ms).
// partial: [gridDim.x][2].  Grid: nit * nchunks blocks, i-tile fastest (neighbouring blocks share the H slices in L2).
// ------------------------------------------------------------------------------------------------
#define ERR64_THREADS 256
#ifndef ERR64_WPS
#define ERR64_WPS 2
#endif
#ifndef ERR64_UNROLL
#define ERR64_UNROLL 2
#endif
#define ERR64_HR 8 // 16-byte pieces of a slice per lane: ranks up to 64
// (Ablations of errors64_kernel -- no sums, no matrix phase, no A loads, no barrier / slice loads: scripts/exp/csrc_r5/k_errors.h with
// scripts/exp/err64_exp.hip.  The product kernel carries no switch.)
__host__ __device__ static inline int errors64_lds_bytes(int k4) { return 3 * k4 * 512; }

template <bool MASKED>
__device__ __forceinline__ void errors64_sums(const f64x4 (&acc)[2][2], const double (&av)[2][2][4], const uint32_t (&mw)[2][4], int ivalid,
                                              int jvalid, int l15, int lg, double &s2, double &skl, const f64x2 *ltab)
{
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double ah = acc[a][b][r], aa = av[a][b][r];
                const double d = aa - ah;
                // (the masked form keeps the table-free logarithm: with the table reads in flight next to the selects it spilled ~900 bytes)
                const double lg_ = MASKED ? nnlm_log_pos(ah + NNLM_TINY) : nnlm_log_tab(ah + NNLM_TINY, ltab);
                const double term = __builtin_fma(-(aa + NNLM_TINY), lg_, ah);
                if constexpr (MASKED) {
                    // (lane values against scalars, constant shifts: per-entry lane constants -- lg | 4 r, 1 << (16 a + l15) -- were hoisted
                    //  out of the tile loop and spilled)
                    const bool valid = (l15 < ivalid - 16 * a) && (lg < jvalid - (16 * b + 4 * r)) && !(((mw[b][r] >> l15) >> (16 * a)) & 1u);
                    s2 += valid ? d * d : 0.0;
                    skl += valid ? term : 0.0;
                } else {
                    s2 = __builtin_fma(d, d, s2);
                    skl += term;
                }
                // (four entries at a time: with all sixteen table reads hoisted the masked form spilled 928 bytes)
                if (r == 3) __builtin_amdgcn_sched_barrier(0);
            }
}

template <bool HAS_MISS>
__global__ __launch_bounds__(ERR64_THREADS, ERR64_WPS) void errors64_kernel(const double *__restrict__ A, int lda, const uint32_t *__restrict__ miss,
                                                                    const double *__restrict__ W64, int ldw, const double *__restrict__ H64,
                                                                    int ldh, int n, int m, int k4, double *__restrict__ partial, int jt0,
                                                                    int jcnt, int chunk, int nit)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_e64[];
    const int slice_bytes = k4 * 512;
    unsigned char *Ws = smem_e64, *Hs = smem_e64 + slice_bytes; // Hs: two buffers
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int it = blockIdx.x % nit, ch = blockIdx.x / nit;
    const int i0 = it * ERR_TILE;
    const int jt_begin = jt0 + ch * chunk;
    const int jt_end = (jt_begin + chunk < jt0 + jcnt) ? jt_begin + chunk : jt0 + jcnt;
    const int ib = 32 * (wave & 1), jb = 32 * (wave >> 1);
    const int nch = k4 >> 1; // 1 KiB pieces (two rows) per slice; piece t belongs to wavefront t & 3
    const int words = lda >> 5;

    // k4 x 64 slice of a factor, columns [c0, c0 + 64), through registers (ERR64_HR x 16 bytes per lane).  Every request of this
    // kernel is an ordinary load the compiler's wait-count pass can see and count: with LDS-DMA in the queue (invisible when
    // written out, a full wait in front of every LDS read when not) its counted waits for the A tile forced part of the NEXT
    // tile to land.  Piece = rows 2t (lanes 0..31) and 2t+1 (lanes 32..63), 16 bytes per lane, odd rows with their 128-byte halves
    // exchanged (see above).
    auto slice_load = [&](const double *X, int ld, int c0, f32x4 (&hr)[ERR64_HR], int ln) {
        const int half = ln >> 5;
        const int lbyte = ((ln & 31) * 16) ^ (half << 7);
#pragma unroll
        for (int u = 0; u < ERR64_HR; u++) {
            const int t = (wave + 4 * u < nch) ? wave + 4 * u : nch - 1; // (clamped, not skipped: a conditional load is sunk to its use)
            hr[u] = *(const f32x4 *)((const unsigned char *)(X + (size_t)(2 * t + half) * ld + c0) + lbyte);
        }
    };
    auto slice_store = [&](unsigned char *dst, const f32x4 (&hr)[ERR64_HR], int ln) {
#pragma unroll
        for (int u = 0; u < ERR64_HR; u++) {
            const int t = wave + 4 * u;
            if (t < nch) *(f32x4 *)(dst + t * 1024 + ln * 16) = hr[u];
        }
    };
    auto tile_a = [&](int jt, double (&av)[2][2][4], uint32_t (&mw)[2][4], int ln) {
        const int l15 = ln & 15, lg = ln >> 4;
        // (addresses rebuilt from an opaque copy of the tile index: as induction variables of the tile loop the 24 of them were kept in
        //  registers -- 48 VGPRs -- and spilled)
        asm volatile("" : "+s"(jt));
        if constexpr (HAS_MISS) {
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) mw[b][r] = miss[(size_t)(jt * ERR_TILE + jb + 16 * b + lg + 4 * r) * words + ((i0 + ib) >> 5)];
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    av[a][b][r] = A[(size_t)(jt * ERR_TILE + jb + 16 * b + lg + 4 * r) * lda + i0 + ib + 16 * a + l15];
    };

    __shared__ f64x2 ltab[64]; // nnlm_log_tab
    nnlm_log_tab_fill(ltab, threadIdx.x);
    double s2 = 0.0, skl = 0.0;
    double av0[2][2][4], av1[2][2][4];
    uint32_t mw0[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, mw1[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 hr[ERR64_HR];
#pragma unroll
    for (int u = 0; u < ERR64_HR; u++) hr[u] = f32x4{0, 0, 0, 0};
    if (jt_begin < jt_end) {
        f32x4 wr[ERR64_HR];
#pragma unroll
        for (int u = 0; u < ERR64_HR; u++) wr[u] = f32x4{0, 0, 0, 0};
        slice_load(W64, ldw, i0, wr, lane);
        slice_load(H64, ldh, jt_begin * ERR_TILE, hr, lane);
        tile_a(jt_begin, av0, mw0, lane);
        slice_store(Ws, wr, lane);
        slice_store(Hs, hr, lane);
    }
    const bool iedge = i0 + ERR_TILE > n;

    // One tile: [the slices written at the end of the last tile are visible] barrier [H slice of the next tile requested into
    // registers] matrix phase [A of the next tile requested] sums [the H registers go to the other LDS buffer].  The loads of A get
    // the sums of one tile and the matrix phase of the next (~6500 cycles) to arrive, the H slice a whole tile.
    auto tile = [&](int jt, int buf, double (&av)[2][2][4], uint32_t (&mw)[2][4], double (&avn)[2][2][4], uint32_t (&mwn)[2][4]) {
        const int jn = (jt + 1 < jt_end) ? jt + 1 : jt; // (unconditional requests: a conditional load is a phi, and the phi a copy behind a full wait)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // With missing entries the lane number is taken afresh in every tile (two instructions) and everything that depends on it --
        // slice / tile / LDS offsets -- is rebuilt from it: as loop invariants those ~20 registers were spilled next to the masked sums'
        // working set, and every reload of a spilled register is a full vmcnt wait on gfx950.
        int ln = lane;
        if constexpr (HAS_MISS) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int l15 = ln & 15, lg = ln >> 4;
        const int swz = (lg & 1) << 7;
        const unsigned char *wrow = Ws + lg * 512;
        const int wo0 = ((ib + l15) * 8) ^ swz, wo1 = ((ib + 16 + l15) * 8) ^ swz;
        const int ho0 = ((jb + l15) * 8) ^ swz, ho1 = ((jb + 16 + l15) * 8) ^ swz;
        slice_load(H64, ldh, jn * ERR_TILE, hr, ln);
        const unsigned char *hrow = Hs + buf * slice_bytes + lg * 512;
        f64x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = f64x4{0, 0, 0, 0};
        // Matrix phase, operands two sets deep: the four LDS reads of step s+1 are in flight while the four MFMAs of step s
        // issue.  Reads and waits are written out and pinned (left to the compiler, the loop is rotated back into read -> wait ->
        // use, one LDS latency per step: 116 instead of 64 cycles per MFMA); the two sets keep their names (no copies of
        // registers a read is still in flight to).
        const unsigned wad = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char *)wrow;
        const unsigned had = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char *)hrow;
        const unsigned aw0 = wad + wo0, aw1 = wad + wo1, ah0 = had + ho0, ah1 = had + ho1;
        double pw0, pw1, ph0, ph1, qw0, qw1, qh0, qh1;
#define E64_READ(w0_, w1_, h0_, h1_, step)                                                                                              \
    {                                                                                                                                 \
        const unsigned o_ = (unsigned)(step) * 2048u;                                                                                 \
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7"                          \
                     : "=&v"(w0_), "=&v"(w1_), "=&v"(h0_), "=&v"(h1_)                                                                   \
                     : "v"(aw0 + o_), "v"(aw1 + o_), "v"(ah0 + o_), "v"(ah1 + o_)                                                      \
                     : "memory");                                                                                                     \
    }
#define E64_MMA(w0_, w1_, h0_, h1_, cnt)                                                                                               \
    {                                                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(w0_), "+v"(w1_), "+v"(h0_), "+v"(h1_)::"memory");                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                           \
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(h0_, w0_, acc[0][0], 0, 0, 0); /* M = column j, N = row i */                  \
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(h0_, w1_, acc[1][0], 0, 0, 0);                                               \
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(h1_, w0_, acc[0][1], 0, 0, 0);                                               \
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(h1_, w1_, acc[1][1], 0, 0, 0);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                           \
    }
        const int ns = k4 >> 2; // k steps of 4
        const int nl = ns - 1;
        double rw0, rw1, rh0, rh1;
        E64_READ(pw0, pw1, ph0, ph1, 0);
        E64_READ(qw0, qw1, qh0, qh1, (1 < nl) ? 1 : nl);
        int st = 0;
        for (; st + 3 <= ns; st += 3) { // (three sets: a read into a set is two MFMA groups behind the group that used it)
            E64_READ(rw0, rw1, rh0, rh1, (st + 2 < nl) ? st + 2 : nl);
            E64_MMA(pw0, pw1, ph0, ph1, 8);
            E64_READ(pw0, pw1, ph0, ph1, (st + 3 < nl) ? st + 3 : nl);
            E64_MMA(qw0, qw1, qh0, qh1, 8);
            E64_READ(qw0, qw1, qh0, qh1, (st + 4 < nl) ? st + 4 : nl);
            E64_MMA(rw0, rw1, rh0, rh1, 8);
        }
        if (st < ns) E64_MMA(pw0, pw1, ph0, ph1, 4);
        if (st + 1 < ns) E64_MMA(qw0, qw1, qh0, qh1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (redundant last reads)
#undef E64_READ
#undef E64_MMA
        if constexpr (!HAS_MISS) tile_a(jn, avn, mwn, ln); // (with missing entries: ONE register set, requested behind the sums -- see below)
        // (with missing entries the H slice of the next tile leaves its registers BEFORE the sums: it was requested a whole matrix phase
        //  ago, the other buffer has been free since this tile's barrier, and the masked sums need the 32 registers -- 25 spilled)
        if constexpr (HAS_MISS) slice_store(Hs + (1 - buf) * slice_bytes, hr, ln);
        const bool masked = HAS_MISS || iedge || ((jt + 1) * ERR_TILE > m); // block uniform
        if (masked) errors64_sums<true>(acc, av, mw, n - i0 - ib, m - jt * ERR_TILE - jb, l15, lg, s2, skl, ltab);
        else errors64_sums<false>(acc, av, mw, 0, 0, l15, lg, s2, skl, ltab);
        // (the masked sums + the mask words + two sets of A spilled 136 .. 928 bytes: with missing entries the next tile is requested
        //  into the SAME set once this tile's sums are done; the next matrix phase covers most of its latency)
        if constexpr (HAS_MISS) tile_a(jn, avn, mwn, ln);
        else slice_store(Hs + (1 - buf) * slice_bytes, hr, ln);
    };
    int jt = jt_begin;
    for (; jt + 1 < jt_end; jt += 2) { // (pairs: the two register sets of A keep their names, no copies)
        if constexpr (HAS_MISS) {
            tile(jt, 0, av0, mw0, av0, mw0);
            tile(jt + 1, 1, av0, mw0, av0, mw0);
        } else {
            tile(jt, 0, av0, mw0, av1, mw1);
            tile(jt + 1, 1, av1, mw1, av0, mw0);
        }
    }
    if (jt < jt_end) {
        if constexpr (HAS_MISS) tile(jt, 0, av0, mw0, av0, mw0);
        else tile(jt, 0, av0, mw0, av1, mw1);
    }

    __shared__ double red64[2][4];
    s2 = wave_sum(s2);
    skl = wave_sum(skl);
    if (lane == 0) {
        red64[0][wave] = s2;
        red64[1][wave] = skl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * (size_t)blockIdx.x] = ((red64[0][0] + red64[0][1]) + red64[0][2]) + red64[0][3];
        partial[2 * (size_t)blockIdx.x + 1] = ((red64[1][0] + red64[1][1]) + red64[1][2]) + red64[1][3];
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

// out[c] = sum_b partial[b*width + c], fixed order (one block of REDUCE_THREADS threads, four loads in flight per thread: the
// error kernels leave up to 5e4 partial pairs behind; one 256-thread block with one load at a time took 0.29 ms for them).
#define REDUCE_THREADS 1024
__global__ __launch_bounds__(REDUCE_THREADS) void reduce_partials_kernel(const double *__restrict__ partial, size_t nblocks, int width,
                                                                         double *__restrict__ out)
{
    __shared__ double red[REDUCE_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = 0; c < width; c++) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        size_t b = threadIdx.x;
        for (; b + 3 * REDUCE_THREADS < nblocks; b += 4 * REDUCE_THREADS) {
            s0 += partial[b * width + c];
            s1 += partial[(b + REDUCE_THREADS) * width + c];
            s2 += partial[(b + 2 * REDUCE_THREADS) * width + c];
            s3 += partial[(b + 3 * REDUCE_THREADS) * width + c];
        }
        for (; b < nblocks; b += REDUCE_THREADS) s0 += partial[b * width + c];
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < REDUCE_THREADS / 64; w++) t += red[w];
            out[c] = t;
        }
        __syncthreads();
    }
}
