// k_xprod16.h -- the A-streaming cross product with split-fp16 operands on the fp16 matrix cores.
//
// Same job as xprod_tn_kernel (k_xprod.h):  C[kq, c] = sum_i Y[kq, i] * A[i, c]  for the columns c of one resident
// matrix, contraction index i contiguous.  The fp32 kernel is bound by its MFMA phase (v_mfma_f32_16x16x4_f32 runs at
// 1/16 of the fp16 rate) while HBM could deliver A in ~0.6 of its time.  Here every operand value v (pre-scaled by a
// power of two so that max|v| lands in [2^14, 2^15)) is stored as TWO fp16 numbers in the SAME 4 bytes per element,
//      hi = fp16(v),      lo = fp16((v - hi) * 2^11)            v = hi + lo * 2^-11  to 2^-22 relative
// and the product is accumulated in fp32 by three v_mfma_f32_16x16x32_f16 per 32 contraction elements:
//      main += hi_a * hi_y          cross += hi_a * lo_y + lo_a * hi_y          C = main + cross * 2^-11
// (lo_a * lo_y, 2^-22 relative, is dropped).  fp32 partial sums are folded into fp64 every 256 contraction elements
// exactly like the fp32 kernel.  HBM traffic and the LDS images have the same size as in the fp32 kernel; the MFMA
// phase is 4x shorter, so the kernel runs at the speed the loads arrive.
//
// Layout ("split rows"): a row of the contraction index is cut into chunks of 64 elements; a chunk is 256 bytes =
// [64 x hi | 64 x lo].  A16 [cols][plen/64][2][64], Y16 [KP][plen/64][2][64] (plen = padded contraction length).
// The W half-step uses the same kernel on a transposed split copy of A (A16T), so there is no "NT" variant.
// Scaling: A by 2^eA once per matrix, the factor by 2^eY per half-step (absmax_kernel + factor16_kernel); the epilogue
// multiplies by 2^-(eA+eY), read from device memory.
#pragma once
#include "common.h"
#include "k_xprod.h"
#include "k_gram.h"
#include <type_traits>

typedef _Float16 xh8 __attribute__((ext_vector_type(8)));
#define XPROD16_LO_SCALE 2048.0f // 2^11

// exponent e such that max * 2^e lies in [2^14, 2^15) (0 for max = 0 / non-finite)
__host__ __device__ static inline int split16_exponent(float maxabs)
{
    if (!(maxabs > 0.0f) || maxabs > 3.0e38f) return 0;
    int ex;
    (void)frexpf(maxabs, &ex); // maxabs = f * 2^ex, f in [0.5, 1)
    return 15 - ex;
}

__device__ static inline void split16(float v, float scale, _Float16 &hi, _Float16 &lo)
{
    const float x = v * scale;
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * XPROD16_LO_SCALE);
}

// scal_exp[0] = eA (written by the host), scal_exp[1] = eY (written by factor16_kernel's companion absmax pass)
// Ring of XPROD_NBUF = 3 LDS stage buffers: two stages in flight while one is consumed, 144 KB at KP = 64 -- the block owns its CU.
// (A two-buffer, 96 KB form that leaves room for a workgroup of the SCD sweep on the same CU was measured in round 4 -- DESIGN.md
// section 6: co-residency hides 0.07 of 0.26 ms -- and is not kept.)
template <int NKQ, int EXP = 0>
__global__ __launch_bounds__(XPROD_THREADS) void xprod16_tn_kernel(const uint32_t *__restrict__ A16, int lda,   // lda: elements per column
                                                                   const uint32_t *__restrict__ Y16, int ldy,   // ldy: elements per row
                                                                   double *__restrict__ Cx, int ldc, size_t slab_stride,
                                                                   int stage_begin, int stage_end, int stages_per_split,
                                                                   const int *__restrict__ scal_exp)
{
    constexpr int KP = 16 * NKQ;
    constexpr int BUF = XPROD_A_IMG_BYTES + KP * XPROD_ROWB;
    constexpr int FL = XPROD_FLUSH_ELEMS / 64;
    constexpr int YI = KP / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    // (EXP bit 4, scripts/exp/xerr_exp.hip: tiles, slabs and stages in descending order -- does a pass find the end of the previous one in the infinity cache?)
    const int j0 = ((EXP & 16) ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * XPROD_TN_BJ;
    int st0 = stage_begin + ((EXP & 16) ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y) * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;

    f32x4 accm[NKQ], accx[NKQ];
    f64x4 acc64[NKQ];
#pragma unroll
    for (int b = 0; b < NKQ; b++) {
        accm[b] = f32x4{0, 0, 0, 0};
        accx[b] = f32x4{0, 0, 0, 0};
        acc64[b] = f64x4{0, 0, 0, 0};
    }

    // a stage = one 256-byte chunk ([64 hi | 64 lo]) of 128 columns of A and of KP rows of the factor; the sixteen
    // 16-byte slots of a row are XOR-swizzled with the row index through the global source address (as in k_xprod.h).
    // Piece t = wave + 8 i of an image = rows 4 t + lg: (row & 15) = (4 wave + lg) & 15 for every i, so ONE per-lane byte offset
    // per image serves all of a wavefront's requests; piece and stage go into the scalar base.
    const int rw = 4 * wave + lg, sw = l15 ^ (rw & 15);
    const unsigned voffA = (unsigned)(((size_t)rw * lda + sw * 4) * 4), voffY = (unsigned)(((size_t)rw * ldy + sw * 4) * 4);
    const unsigned long long baseA = xp_uniform64(A16 + (size_t)j0 * lda), baseY = xp_uniform64(Y16);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    auto issue = [&](int st, int bi) {
        const int sta = (EXP & 16) ? st1 - 1 - (st - st0) : st;
        const unsigned long long c0b = (unsigned long long)((EXP & 4) && st > st0 + 1 ? st0 : sta) * 256ull; // byte offset of the chunk
        const unsigned dst = lds0 + (unsigned)bi * (unsigned)BUF + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < XPROD_A_IMG_BYTES / 1024 / XPROD_WAVES; i++)
            glds16_s(voffA, baseA + c0b + (unsigned long long)i * 32ull * (unsigned long long)lda * 4ull, dst + (unsigned)i * 8192u);
#pragma unroll
        for (int i = 0; i < (YI + XPROD_WAVES - 1) / XPROD_WAVES; i++)
            if (wave + XPROD_WAVES * i < YI)
                glds16_s(voffY, baseY + c0b + (unsigned long long)i * 32ull * (unsigned long long)ldy * 4ull,
                         dst + (unsigned)XPROD_A_IMG_BYTES + (unsigned)i * 8192u);
    };
    const int per_stage = XPROD_A_IMG_BYTES / 1024 / XPROD_WAVES + strided_count(wave, YI);
    if (st0 < st1) issue(st0, 0);
    constexpr int NBUF = XPROD_NBUF;
    static_assert(NBUF == 3, "two stages in flight while one is consumed");
    if (st0 + 1 < st1) issue(st0 + 1, 1);
    int since_flush = 0;
    for (int st = st0; st < st1; ++st) {
        unsigned char *buf = smem + ((st - st0) % NBUF) * BUF;
        wait_vmcnt((st + 1 < st1) ? per_stage : 0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool late = (EXP & 8) && wave >= XPROD_WAVES / 2; // (experiment: half of the wavefronts issue behind their MFMA phase)
        if (st + 2 < st1 && !late) issue(st + 2, (st + 2 - st0) % NBUF);
        if (EXP & 2) continue;
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) { // two K = 32 chunks per stage; lane (l15, lg) holds elements 32*c2 + 8*lg .. +7
            const int arow = 16 * wave + l15;
            const int sh = (4 * c2 + lg), sl = 8 + 4 * c2 + lg; // logical 16-byte slots of the hi / lo halves
            const xh8 ah = *(const xh8 *)(buf + arow * XPROD_ROWB + ((sh ^ l15) * 16));
            const xh8 al = *(const xh8 *)(buf + arow * XPROD_ROWB + ((sl ^ l15) * 16));
            xh8 yh[NKQ], yl[NKQ];
#pragma unroll
            for (int nt = 0; nt < NKQ; nt++) {
                const unsigned char *yrow = buf + XPROD_A_IMG_BYTES + (16 * nt + l15) * XPROD_ROWB;
                yh[nt] = *(const xh8 *)(yrow + ((sh ^ l15) * 16));
                yl[nt] = *(const xh8 *)(yrow + ((sl ^ l15) * 16));
            }
#pragma unroll
            for (int nt = 0; nt < NKQ; nt++) {
                accm[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yh[nt], accm[nt], 0, 0, 0);
                accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yl[nt], accx[nt], 0, 0, 0);
                accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, yh[nt], accx[nt], 0, 0, 0);
            }
        }
        if (st + 2 < st1 && late) issue(st + 2, (st + 2 - st0) % NBUF);
        if (++since_flush == FL) {
            since_flush = 0;
#pragma unroll
            for (int b = 0; b < NKQ; b++) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc64[b][r] += (double)accm[b][r] + (double)accx[b][r] * (1.0 / XPROD16_LO_SCALE);
                accm[b] = f32x4{0, 0, 0, 0};
                accx[b] = f32x4{0, 0, 0, 0};
            }
        }
    }
    // epilogue: D[M = column, N = kq], undo the two power-of-two scalings (exact)
    const double unscale = ldexp(1.0, -(scal_exp[0] + scal_exp[1]));
    double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
    for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int kq = 16 * nt + l15;
            const int j = j0 + 16 * wave + 4 * lg + r;
            const double v = acc64[nt][r] + (double)accm[nt][r] + (double)accx[nt][r] * (1.0 / XPROD16_LO_SCALE);
            out[(size_t)kq * ldc + j] = v * unscale;
        }
}

// ---- operand preparation -------------------------------------------------------------------------------------------
// maxbits = max over all elements of the bit pattern of |x| as float (non-negative floats order like unsigned ints)
__global__ __launch_bounds__(256) void absmax_f64_kernel(const double *__restrict__ X, int ld, int ncols, int k, unsigned *__restrict__ maxbits)
{
    const int col = blockIdx.x * 256 + threadIdx.x;
    float mx = 0.0f;
    if (col < ncols)
        for (int q = 0; q < k; q++) mx = fmaxf(mx, fabsf((float)X[(size_t)q * ld + col]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.0f) atomicMax(maxbits, __float_as_uint(mx));
}

__global__ __launch_bounds__(256) void absmax_f32_kernel(const float *__restrict__ X, size_t count, unsigned *__restrict__ maxbits)
{
    float mx = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) mx = fmaxf(mx, fabsf(X[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.0f) atomicMax(maxbits, __float_as_uint(mx));
}

// Y16 [KP][plen/64][2][64] from the fp64 master X [KP][ld]; exp_out = the exponent used (read by the cross product).
// One thread per element; maxbits comes from absmax_f64_kernel (zeroed by the host before that pass).
__global__ __launch_bounds__(256) void factor16_kernel(const double *__restrict__ X, int ld, int ncols, int k, int KP, int plen,
                                                       unsigned *__restrict__ maxbits, int *__restrict__ exp_out, uint32_t *__restrict__ Y16,
                                                       unsigned *__restrict__ zero_word = nullptr)
{
    const int e = split16_exponent(__uint_as_float(*maxbits));
    const float scale = ldexpf(1.0f, e);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; // over KP * plen
    if (idx < (size_t)KP * plen) {
        const int q = (int)(idx / plen), i = (int)(idx % plen);
        const float v = (q < k && i < ncols) ? (float)X[(size_t)q * ld + i] : 0.0f;
        _Float16 hi, lo;
        split16(v, scale, hi, lo);
        _Float16 *row = (_Float16 *)(Y16 + (size_t)q * plen + (size_t)(i >> 6) * 64);
        row[i & 63] = hi;
        row[64 + (i & 63)] = lo;
    }
    if (idx == 0) {
        *exp_out = e;
        if (zero_word) *zero_word = 0u; // the word the sweep of THIS half-step accumulates its max into
    }
}

// factor16_kernel and gram_fold_kernel (k_gram.h) in ONE launch of 1024-thread blocks -- the two are independent and each is
// as long as a launch is (4.5 us): blocks [0, KP*KP/64) fold the Gram slabs, the rest convert 1024 entries of the factor each.
__global__ __launch_bounds__(1024) void factor16_fold_kernel(const double *__restrict__ X, int ld, int ncols, int k, int KP, int plen,
                                                             unsigned *__restrict__ maxbits, int *__restrict__ exp_out, uint32_t *__restrict__ Y16,
                                                             unsigned *__restrict__ zero_word, const double *__restrict__ slabs, int nslabs,
                                                             double *__restrict__ G, const SweepImg im)
{
    const int nfold = KP * KP / 64;
    if ((int)blockIdx.x < nfold) {
        gram_fold_body(slabs, nslabs, KP, G, blockIdx.x, im); // (+ the operand image of this half-step's sweep: no pack launch)
        return;
    }
    const int e = split16_exponent(__uint_as_float(*maxbits));
    const float scale = ldexpf(1.0f, e);
    const size_t idx = (size_t)(blockIdx.x - nfold) * 1024 + threadIdx.x; // over KP * plen
    if (idx < (size_t)KP * plen) {
        const int q = (int)(idx / plen), i = (int)(idx % plen);
        const float v = (q < k && i < ncols) ? (float)X[(size_t)q * ld + i] : 0.0f;
        _Float16 hi, lo;
        split16(v, scale, hi, lo);
        _Float16 *row = (_Float16 *)(Y16 + (size_t)q * plen + (size_t)(i >> 6) * 64);
        row[i & 63] = hi;
        row[64 + (i & 63)] = lo;
    }
    if (idx == 0) {
        *exp_out = e;
        if (zero_word) *zero_word = 0u;
    }
}

// A16 [cols][plen/64][2][64] from the resident fp32 A [cols][lda] (same column-major geometry, plen = lda)
__global__ __launch_bounds__(256) void a16_convert_kernel(const float *__restrict__ A, size_t count, float scale, uint32_t *__restrict__ A16)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= count) return;
    _Float16 hi, lo;
    split16(A[idx], scale, hi, lo);
    _Float16 *row = (_Float16 *)(A16 + (idx & ~(size_t)63));
    row[idx & 63] = hi;
    row[64 + (idx & 63)] = lo;
}

// A16T [npad][mpad/64][2][64]: split copy of A transposed (row i of A becomes a "column"), 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void a16_transpose_kernel(const float *__restrict__ A, int lda, int npad, int mpad, float scale,
                                                            uint32_t *__restrict__ A16T)
{
    __shared__ float tile[64][65];
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int jj = e / 64, ii = e % 64;
        tile[jj][ii] = A[(size_t)(j0 + jj) * lda + i0 + ii];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int ii = e / 64, jj = e % 64;
        _Float16 hi, lo;
        split16(tile[jj][ii], scale, hi, lo);
        _Float16 *row = (_Float16 *)(A16T + (size_t)(i0 + ii) * mpad + (size_t)j0);
        row[jj] = hi;
        row[64 + jj] = lo;
    }
}

// ---- cross product of the W half-step + the error block of the PREVIOUS iteration in one pass over A --------------------
// On a trace iteration the reference evaluates  mse = mean((A - W H)^2)  and the KL term  mean(WH - (A+eps) log(WH+eps))
// (src/nnmf.cpp:135-140) -- one more pass over A.  The W half-step that follows streams A anyway (as A16T: 128 rows i per
// block, 64 columns j per stage) with H as its fixed factor, and W is not touched until its sweep: so  W H  for the tile in
// LDS costs 3 more fp16 MFMAs per 16 x 16 x 32 (as cheap as the cross product's own) and the two sums ride along.
//   * W rows of the block: kq-contiguous split copy W16c [npad][2][64], held in registers (A operand, M = i, K = kq);
//   * H columns of the stage: kq-contiguous split copy H16c [mpad][2][64], a third LDS image (B operand, N = j);
//   * a(i, j) is rebuilt from the hi/lo halves already in the A image (hi + lo * 2^-11: 22 bits), moved into the accumulator
//     layout of W H by a one-hot MFMA (exact: one product with 1.0 and one with 2^-11 per element);
//   * the two sums: fp32 over the 16 elements of a lane per stage, fp64 across stages, one pair per block in `partial`.
// No missing values, no masks on A (host falls back to errors_f32_kernel).
//
// Wavefront-specialised (round 5): a wavefront owns 32 rows of the block and ONE of the two jobs.
//   * wavefronts 0..3 ("E", rows 32 rg ..): W H and the error arithmetic.  W fragments of two M-tiles stay in registers, every H
//     fragment is read once per 32 rows, and the arithmetic of column tile t runs in the shadow of the MFMAs of tile t + 1;
//   * wavefronts 4..7 ("X", same rows): the cross product (every Y fragment read once per 32 rows) and ALL of the block's requests.
//     A wavefront that hands 16 KB of requests to the memory pipeline sits in the issue queue until most of the previous stage has
//     been delivered (~1100 cycles at config 2); while it does, the E wavefront of its SIMD has the matrix and vector pipes.
// The rounds 1-4 form (eight wavefronts x 16 rows, each doing both jobs and its share of the requests: scripts/exp/k_xerr.h xerr0) spent
// 2700 of its 5800 cycles per stage with all eight wavefronts in that queue and read 352 KB of fragments per stage; this one reads 192 KB
// and takes 4000 (0.293 -> 0.228 ms alone at config 2, scripts/exp/xerr_exp.hip).  Row tile 2 rg + mt of this kernel = wavefront 2 rg + mt
// of the old one, with the same accumulation orders: the cross product is bit-identical to it, the error sums agree to 3e-11 relative
// (where the compiler contracts the fp32 partial sums of a stage into fmas differs between the two code shapes).
// LDS: ring of THREE A images (the HBM stream: two stages in flight, counted wait) + two pairs of factor images (L2 hits: one stage in
// flight) = 96 + 2 x (4 NKQ + 16) KB = 160 KB at NKQ = 4; the block's reduction scratch aliases the ring.
__host__ __device__ static inline int xprod16_err_lds_bytes(int NKQ) { return 3 * XPROD_A_IMG_BYTES + 2 * (16 * NKQ + 64) * XPROD_ROWB; }
// HAS_MISS: entries of A that are missing (stored as 0 in the split copy, so the cross product is already right) are left out of the two
// sums -- bit j % 32 of missT[i][j / 32], the transposed bit matrix the W half-step of the NA flow uses; an E wavefront fetches the two
// words of each of its 32 rows per stage (8 eight-byte loads per lane, 4 distinct addresses per request).
template <int NKQ, bool HAS_MISS = false>
__global__ __launch_bounds__(XPROD_THREADS) void xprod16_err_kernel(const uint32_t *__restrict__ A16, int lda, const uint32_t *__restrict__ Y16, int ldy,
                                                                    const uint32_t *__restrict__ H16c, const uint32_t *__restrict__ W16c,
                                                                    double *__restrict__ Cx, int ldc, size_t slab_stride, int stage_begin,
                                                                    int stage_end, int stages_per_split, const int *__restrict__ scal_exp,
                                                                    const int *__restrict__ w_exp, int n_rows, int n_cols,
                                                                    double *__restrict__ partial, unsigned *__restrict__ zero_word,
                                                                    const uint32_t *__restrict__ missT = nullptr, int wordsT = 0)
{
    constexpr int FL = XPROD_FLUSH_ELEMS / 64;
    constexpr int NC2 = NKQ > 2 ? 2 : 1;                                   // 32-wide chunks of kq that can be non-zero
    constexpr int HOFF = 16 * NKQ * XPROD_ROWB, FBUF = HOFF + 64 * XPROD_ROWB; // a pair of factor images: [Y: 16 NKQ rows | H: 64 columns]
    constexpr int FOFF = 3 * XPROD_A_IMG_BYTES;
    constexpr int A_REQ = XPROD_A_IMG_BYTES / 1024 / 4;                    // requests per X wavefront and A image
    static_assert(A_REQ == 8, "wait_vmcnt(8) below");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int rg = wave & 3; // rows 32 rg .. 32 rg + 31 of the block
    const int i0 = blockIdx.x * XPROD_TN_BJ;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;
    double *red = (double *)smem; // [2][XPROD_WAVES], after the last fragment read
    // (the word this half-step's sweep accumulates max|x| into: factor16_fold_err_kernel has read max|W| from it and could not clear it)
    if (zero_word && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *zero_word = 0u;
    // byte offsets of a lane's fragment inside an image row: 16-byte slot (4 c2 + lg) of the hi half, (8 + 4 c2 + lg) of the lo half
    const int oh0 = ((0 + lg) ^ l15) * 16, oh1 = ((4 + lg) ^ l15) * 16, ol0 = ((8 + lg) ^ l15) * 16, ol1 = ((12 + lg) ^ l15) * 16;

    if (wave >= XPROD_WAVES / 2) {
        // ---------------------------------------------------------------- X: cross product + requests
        f32x4 accm[2][NKQ], accx[2][NKQ];
        f64x4 acc64[2][NKQ];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int b = 0; b < NKQ; b++) {
                accm[mt][b] = f32x4{0, 0, 0, 0};
                accx[mt][b] = f32x4{0, 0, 0, 0};
                acc64[mt][b] = f64x4{0, 0, 0, 0};
            }
        // piece t = rg + 4 i of an image = rows 4 t + lg = rw + 16 i: (row & 15) = rw & 15 for every i, so one per-lane byte offset
        // per image serves all of a wavefront's requests (slots XOR-swizzled through the global source address, as in k_xprod.h)
        const int rw = 4 * rg + lg, sw = l15 ^ (rw & 15);
        const unsigned voffA = (unsigned)(((size_t)rw * lda + sw * 4) * 4), voffY = (unsigned)(((size_t)rw * ldy + sw * 4) * 4);
        const unsigned voffH = (unsigned)((rw * 64 + sw * 4) * 4);
        const unsigned long long baseA = xp_uniform64(A16 + (size_t)i0 * lda), baseY = xp_uniform64(Y16), baseH = xp_uniform64(H16c);
        const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
        auto issue_a = [&](int st) {
            const unsigned long long c0b = (unsigned long long)st * 256ull;
            const unsigned dst = lds0 + (unsigned)((st - st0) % 3) * (unsigned)XPROD_A_IMG_BYTES + (unsigned)rg * 1024u;
#pragma unroll
            for (int i = 0; i < A_REQ; i++) glds16_s(voffA, baseA + c0b + (unsigned long long)i * 16ull * (unsigned long long)lda * 4ull, dst + (unsigned)i * 4096u);
        };
        auto issue_f = [&](int st) { // factor rows (Y, 64 columns j of the stage each) and factor columns (H, 64 kq each)
            const unsigned long long c0b = (unsigned long long)st * 256ull;
            const unsigned dst = lds0 + (unsigned)FOFF + (unsigned)((st - st0) & 1) * (unsigned)FBUF + (unsigned)rg * 1024u;
#pragma unroll
            for (int i = 0; i < NKQ; i++) glds16_s(voffY, baseY + c0b + (unsigned long long)i * 16ull * (unsigned long long)ldy * 4ull, dst + (unsigned)i * 4096u);
#pragma unroll
            for (int i = 0; i < 4; i++)
                glds16_s(voffH, baseH + (unsigned long long)st * 64ull * 256ull + (unsigned long long)i * 16ull * 256ull, dst + (unsigned)HOFF + (unsigned)i * 4096u);
        };
        if (st0 < st1) {
            issue_f(st0);
            issue_a(st0);
        }
        if (st0 + 1 < st1) issue_a(st0 + 1);
        int since_flush = 0;
        for (int st = st0; st < st1; ++st) {
            const unsigned char *abuf = smem + ((st - st0) % 3) * XPROD_A_IMG_BYTES, *fbuf = smem + FOFF + ((st - st0) & 1) * FBUF;
            // requests complete in issue order: behind this stage's images only the A image of the next stage has been issued
            if (st + 1 < st1) wait_vmcnt(A_REQ);
            else wait_vmcnt(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (st + 1 < st1) issue_f(st + 1);
            if (st + 2 < st1) issue_a(st + 2);
            const unsigned char *arow0 = abuf + (32 * rg + l15) * XPROD_ROWB, *arow1 = arow0 + 16 * XPROD_ROWB;
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++) { // two K = 32 chunks of the stage's 64 columns
                const int oh = c2 ? oh1 : oh0, ol = c2 ? ol1 : ol0;
                const xh8 ah0 = *(const xh8 *)(arow0 + oh), al0 = *(const xh8 *)(arow0 + ol);
                const xh8 ah1 = *(const xh8 *)(arow1 + oh), al1 = *(const xh8 *)(arow1 + ol);
#pragma unroll
                for (int nt = 0; nt < NKQ; nt++) {
                    const unsigned char *yrow = fbuf + (16 * nt + l15) * XPROD_ROWB;
                    const xh8 yh = *(const xh8 *)(yrow + oh);
                    const xh8 yl = *(const xh8 *)(yrow + ol);
                    accm[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, yh, accm[0][nt], 0, 0, 0);
                    accx[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, yl, accx[0][nt], 0, 0, 0);
                    accx[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, yh, accx[0][nt], 0, 0, 0);
                    accm[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, yh, accm[1][nt], 0, 0, 0);
                    accx[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, yl, accx[1][nt], 0, 0, 0);
                    accx[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, yh, accx[1][nt], 0, 0, 0);
                }
            }
            if (++since_flush == FL) {
                since_flush = 0;
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int b = 0; b < NKQ; b++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) acc64[mt][b][r] += (double)accm[mt][b][r] + (double)accx[mt][b][r] * (1.0 / XPROD16_LO_SCALE);
                        accm[mt][b] = f32x4{0, 0, 0, 0};
                        accx[mt][b] = f32x4{0, 0, 0, 0};
                    }
            }
        }
        const double unscale = ldexp(1.0, -(scal_exp[0] + scal_exp[1]));
        double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kq = 16 * nt + l15;
                    const int j = i0 + 32 * rg + 16 * mt + 4 * lg + r;
                    const double v = acc64[mt][nt][r] + (double)accm[mt][nt][r] + (double)accx[mt][nt][r] * (1.0 / XPROD16_LO_SCALE);
                    out[(size_t)kq * ldc + j] = v * unscale;
                }
        __syncthreads(); // (pairs with the E wavefronts' barrier in front of their use of the ring as scratch)
    } else {
        // ---------------------------------------------------------------- E: W H and the two error sums
        // this wavefront's 32 rows of W, kq-contiguous: lane (l15 = row, lg) holds kq = 32c + 8lg .. +7 of both halves
        xh8 wh[2][NC2], wl[2][NC2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const _Float16 *wrow = (const _Float16 *)(W16c + (size_t)(i0 + 32 * rg + 16 * mt + l15) * 64);
#pragma unroll
            for (int c = 0; c < NC2; c++) {
                wh[mt][c] = *(const xh8 *)(wrow + 32 * c + 8 * lg);
                wl[mt][c] = *(const xh8 *)(wrow + 64 + 32 * c + 8 * lg);
            }
        }
        // a(i, j) of a 16 x 16 tile in the accumulator layout of W H by ONE MFMA: its A operand holds the hi halves of the tile's 16 columns
        // in K slots 0..15 and their lo halves in slots 16..31 (lane group lg reads slot 2 t + lg of the hi half of the image row for
        // lg < 2, slot 2 t + lg - 2 of the lo half otherwise), the B operand is 1 at slot n and 2^-11 at slot 16 + n for the lane's column
        // n = l15: D = hi + lo * 2^-11, exact in fp32 (22 bits).
        xh8 identm;
#pragma unroll
        for (int e = 0; e < 8; e++)
            identm[e] = (8 * lg + e == l15) ? (_Float16)1.0f : (8 * lg + e == 16 + l15) ? (_Float16)(1.0f / XPROD16_LO_SCALE) : (_Float16)0.0f;
        const int amslot = (lg < 2 ? 0 : 8) + (lg & 1);
        const float ca = ldexpf(1.0f, -scal_exp[0]);               // a      = (hi + lo/2048) * ca
        const float cwh = ldexpf(1.0f, -(w_exp[0] + scal_exp[1])); // (W H)  = (main + cross/2048) * cwh
        const float il = 1.0f / XPROD16_LO_SCALE, tiny = (float)NNLM_TINY;
        double s2[2] = {0.0, 0.0}, skl[2] = {0.0, 0.0}; // per M-tile, as the two 16-row wavefronts of the old form kept them
        const unsigned lanebit[2] = {1u << l15, 1u << (16 + l15)}; // column 16 t + l15 of the stage = bit 16 (t & 1) + l15 of word t / 2
        const uint32_t *mrow = HAS_MISS ? missT + (size_t)(i0 + 32 * rg + 4 * lg) * wordsT : nullptr;
        for (int st = st0; st < st1; ++st) {
            const unsigned char *abuf = smem + ((st - st0) % 3) * XPROD_A_IMG_BYTES, *fbuf = smem + FOFF + ((st - st0) & 1) * FBUF;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const bool interior = (i0 + XPROD_TN_BJ <= n_rows) && (st * 64 + 64 <= n_cols);
            // two copies of the stage: the interior one has no edge tests and is a single basic block (MFMAs and arithmetic interleave)
            auto stage_body = [&](auto interior_c) {
                constexpr bool INTERIOR = decltype(interior_c)::value;
                const unsigned char *arow0 = abuf + (32 * rg + l15) * XPROD_ROWB, *arow1 = arow0 + 16 * XPROD_ROWB;
                uint2 mw[2][4]; // [M-tile][r]: the stage's 64 mask bits of row 32 rg + 16 mt + 4 lg + r
                if constexpr (HAS_MISS) {
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
#pragma unroll
                        for (int r = 0; r < 4; r++) mw[mt][r] = *(const uint2 *)(mrow + (size_t)(16 * mt + r) * wordsT + 2 * st);
                }
                xh8 am[2][2]; // [buffer][M-tile]: hi | lo halves of a(i, 16 t .. 16 t + 15), column tiles t and t + 1
                auto read_a = [&](int t, int b) {
                    am[b][0] = *(const xh8 *)(arow0 + (((amslot + 2 * t) ^ l15) * 16));
                    am[b][1] = *(const xh8 *)(arow1 + (((amslot + 2 * t) ^ l15) * 16));
                };
                read_a(0, 0);
                xh8 hh[2][NC2], hl[2][NC2]; // [buffer][K chunk of kq]: H fragments of column tile t and t + 1
                auto read_h = [&](int t, int b) {
                    const unsigned char *hrow = fbuf + HOFF + (16 * t + l15) * XPROD_ROWB;
                    hh[b][0] = *(const xh8 *)(hrow + oh0), hl[b][0] = *(const xh8 *)(hrow + ol0);
                    if constexpr (NC2 > 1) hh[b][NC2 - 1] = *(const xh8 *)(hrow + oh1), hl[b][NC2 - 1] = *(const xh8 *)(hrow + ol1);
                };
                read_h(0, 0);
                f32x4 em[2][2], ex[2][2], da[2][2]; // [buffer][M-tile]: W H (main, cross) and a of a 16 x 16 tile, all still scaled
                f32x4 p2[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, pk[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
                // the two sums over a 16 x 16 tile: lane (l15 = column within tile t, lg) holds rows 4*lg + r (vector arithmetic over r)
                auto sums = [&](int t, int b) {
#pragma unroll
                    for (int mt = 0; mt < 2; mt++) {
                        const f32x4 aa = da[b][mt] * ca;
                        const f32x4 ah2 = (em[b][mt] + ex[b][mt] * il) * cwh;
                        const f32x4 d = aa - ah2;
                        f32x4 lg4;
#pragma unroll
                        for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny); // (log2: ln 2 goes into the coefficient)
                        f32x4 t2 = d * d;
                        f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
                        if constexpr (!INTERIOR || HAS_MISS) {
                            // (selects, no short-circuit: written as `if (!(inside && !missing)) t2 = tk = 0` hipcc 7.2 turned the chain into
                            //  exec-masked regions and lost the zeroing of one of the four entries -- found by the NA golden test)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                unsigned drop = 0u;
                                if constexpr (!INTERIOR)
                                    drop = (unsigned)(i0 + 32 * rg + 16 * mt + 4 * lg + r >= n_rows) | (unsigned)(st * 64 + 16 * t + l15 >= n_cols);
                                if constexpr (HAS_MISS) drop |= ((t < 2) ? mw[mt][r].x : mw[mt][r].y) & lanebit[t & 1];
                                t2[r] = drop ? 0.f : t2[r];
                                tk[r] = drop ? 0.f : tk[r];
                            }
                        }
                        p2[mt] += t2;
                        pk[mt] += tk;
                    }
                };
#pragma unroll
                for (int t = 0; t < 4; t++) { // column tile t of the stage
                    const int b = t & 1;
                    if (t < 3) read_h(t + 1, b ^ 1), read_a(t + 1, b ^ 1);
#pragma unroll
                    for (int mt = 0; mt < 2; mt++) da[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[b][mt], identm, f32x4{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; mt++) {
                        em[b][mt] = f32x4{0, 0, 0, 0}, ex[b][mt] = f32x4{0, 0, 0, 0};
#pragma unroll
                        for (int c2 = 0; c2 < NC2; c2++) {
                            em[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt][c2], hh[b][c2], em[b][mt], 0, 0, 0);
                            ex[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt][c2], hl[b][c2], ex[b][mt], 0, 0, 0);
                            ex[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[mt][c2], hh[b][c2], ex[b][mt], 0, 0, 0);
                        }
                    }
                    if (t > 0) sums(t - 1, b ^ 1); // (in the shadow of the MFMAs just issued)
                    __builtin_amdgcn_sched_barrier(0);
                }
                sums(3, 1);
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    s2[mt] += (double)((p2[mt][0] + p2[mt][1]) + (p2[mt][2] + p2[mt][3]));
                    skl[mt] += (double)((pk[mt][0] + pk[mt][1]) + (pk[mt][2] + pk[mt][3]));
                }
            };
            if (interior) stage_body(std::true_type{});
            else stage_body(std::false_type{});
        }
        __syncthreads(); // every fragment read of the block is done: the ring becomes the reduction scratch
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const double a = wave_sum(s2[mt]), b = wave_sum(skl[mt]);
            if (lane == 0) red[2 * rg + mt] = a, red[XPROD_WAVES + 2 * rg + mt] = b;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double a2 = 0.0, ak = 0.0;
        for (int w = 0; w < XPROD_WAVES; w++) a2 += red[w], ak += red[XPROD_WAVES + w]; // (row tiles in the old form's wavefront order)
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = a2;
        partial[2 * blk + 1] = ak;
    }
}

// X16c [cols][2][64]: kq-contiguous split copy of the fp64 master X [KP][ld]; one block per 64 columns, transposed through LDS
// so that both the reads (along the columns) and the writes (256 bytes per column) are contiguous
__global__ __launch_bounds__(256) void factor16c_kernel(const double *__restrict__ X, int ld, int ncols, int k, const unsigned *__restrict__ maxbits,
                                                        int *__restrict__ exp_out, uint32_t *__restrict__ X16c)
{
    __shared__ float tile[64][65]; // [kq][column]
    const int e = split16_exponent(__uint_as_float(*maxbits));
    const float scale = ldexpf(1.0f, e);
    const int c0 = blockIdx.x * 64;
    for (int t = threadIdx.x; t < 64 * 64; t += 256) {
        const int q = t >> 6, c = t & 63;
        tile[q][c] = (q < k && c0 + c < ncols) ? (float)X[(size_t)q * ld + c0 + c] : 0.0f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 64 * 64; t += 256) {
        const int c = t >> 6, q = t & 63;
        _Float16 hi, lo;
        split16(tile[q][c], scale, hi, lo);
        _Float16 *row = (_Float16 *)(X16c + (size_t)(c0 + c) * 64);
        row[q] = hi;
        row[64 + q] = lo;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && exp_out) *exp_out = e;
}

// factor16c_kernel's tile with 1024 threads (device function: blk = 64-column tile)
__device__ static inline void factor16c_body(const double *__restrict__ X, int ld, int ncols, int k, float scale, uint32_t *__restrict__ X16c, int blk,
                                             float (*tile)[65])
{
    const int c0 = blk * 64;
    for (int t = threadIdx.x; t < 64 * 64; t += 1024) {
        const int q = t >> 6, c = t & 63;
        tile[q][c] = (q < k && c0 + c < ncols) ? (float)X[(size_t)q * ld + c0 + c] : 0.0f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 64 * 64; t += 1024) {
        const int c = t >> 6, q = t & 63;
        _Float16 hi, lo;
        split16(tile[q][c], scale, hi, lo);
        _Float16 *row = (_Float16 *)(X16c + (size_t)(c0 + c) * 64);
        row[q] = hi;
        row[64 + q] = lo;
    }
}

// Everything between a sweep and the fused cross product / error block of a trace iteration in ONE launch of 1024-thread blocks
// (prepare_factor16's three kernels + gram_fold_kernel were 4 launches of 5-9 us each, none of which depends on another):
//   blocks [0, nfold)            fold the sweep's Gram slabs (+ the operand image of this half-step's sweep)
//   blocks [nfold, +nf16)        Y16: split rows of the fixed factor H, 1024 entries each                 (exponent -> exp_out[0])
//   blocks [.., +mpad/64)        H16c: kq-contiguous split copy of H, same exponent
//   blocks [.., +npad/64)        W16c: kq-contiguous split copy of W, scaled by max|W| that W's sweep left in *wmax (exponent -> w_exp_out)
// *wmax is the word this half-step's sweep accumulates into: it is cleared by xprod16_err_kernel, which runs between the two.
__global__ __launch_bounds__(1024) void factor16_fold_err_kernel(const double *__restrict__ Hm, int ldh, int m, const double *__restrict__ Wm, int ldw, int n, int k,
                                                                 int KP, const unsigned *__restrict__ hmax, const unsigned *__restrict__ wmax,
                                                                 int *__restrict__ exp_out, int *__restrict__ w_exp_out, uint32_t *__restrict__ Y16,
                                                                 uint32_t *__restrict__ H16c, uint32_t *__restrict__ W16c, const double *__restrict__ slabs,
                                                                 int nslabs, double *__restrict__ G, const SweepImg im)
{
    __shared__ float tile[64][65];
    const int nfold = KP * KP / 64, nf16 = (int)(((size_t)KP * ldh + 1023) / 1024), nhc = ldh / 64;
    int b = blockIdx.x;
    if (b < nfold) {
        gram_fold_body(slabs, nslabs, KP, G, b, im);
        return;
    }
    b -= nfold;
    const int eh = split16_exponent(__uint_as_float(*hmax));
    if (b < nf16) {
        const float scale = ldexpf(1.0f, eh);
        const size_t idx = (size_t)b * 1024 + threadIdx.x; // over KP * ldh
        if (idx < (size_t)KP * ldh) {
            const int q = (int)(idx / ldh), i = (int)(idx % ldh);
            const float v = (q < k && i < m) ? (float)Hm[(size_t)q * ldh + i] : 0.0f;
            _Float16 hi, lo;
            split16(v, scale, hi, lo);
            _Float16 *row = (_Float16 *)(Y16 + (size_t)q * ldh + (size_t)(i >> 6) * 64);
            row[i & 63] = hi;
            row[64 + (i & 63)] = lo;
        }
        if (idx == 0) *exp_out = eh;
        return;
    }
    b -= nf16;
    if (b < nhc) {
        factor16c_body(Hm, ldh, m, k, ldexpf(1.0f, eh), H16c, b, tile);
        return;
    }
    b -= nhc;
    const int ew = split16_exponent(__uint_as_float(*wmax));
    factor16c_body(Wm, ldw, n, k, ldexpf(1.0f, ew), W16c, b, tile);
    if (b == 0 && threadIdx.x == 0) *w_exp_out = ew;
}

