// k_sweep_q.h -- SCD least-squares sweep of the fp32-operand mode, ONE wavefront per 16 columns, no exchange between
// wavefronts at all: the whole recurrence of scd_ls_update (reference src/base_algorithms.cpp:3-37) on
// v_mfma_f64_4x4x4_4b_f64.
//
// What the workgroup-specialised kernels of rounds 1-2 (one chain wavefront + three update wavefronts on v_mfma_f64_16x16x4) paid per
// block of 4 coordinates -- one s_barrier and two LDS round trips per role, ~500 of ~690 cycles -- came from the operand layouts: the chain wanted
// lane = column, the 16x16x4 matrix instruction wants lane = (coordinate, column).  The 4x4x4 instruction with 4 blocks
// (probed on the box, scripts/exp/mfma44_exp.hip: A lane = 16 k + 4 blk + i, B lane = 16 k + 4 blk + j,
// D lane = 16 i + 4 blk + j; 16.6 cycles back to back; cbsz / abid are ignored, scripts/exp/mfma44_cbsz.hip) has the SAME
// lane map for its B operand and its result: lane (i, col) with col = 4 blk + j.  So with
//      block beta = coordinates 4 beta .. 4 beta + 3,   accumulator acc[beta] at lane (i, col) = nu[4 beta + i][col]
// (nu = mu / G[q][q], rows of G divided by their diagonal) the deltas of a block, computed by the lanes
// that hold its gradients, ARE the B operand of the rank-4 update of every other block -- nothing moves between lanes.
//   * The four dependent coordinate steps of a block run on the matrix core as well: with L = the strictly lower part of the
//     block's own 4x4 piece of G',   c <- max(-x, -(m0 + L c))   three times makes rows 0..s of c final after pass s
//     (row s only needs rows < s), so after three passes c is the block's delta vector: 4 v_max_f64 + 3 small MFMAs, no
//     cross-lane traffic, no per-coordinate FMAs.  Candidates of rows that are not final yet are finite and multiplied by zeros.
//   * A operands: lane (k, blk, i) of an operand holds G'[4 b + i][4 beta + k] (the same 16 numbers in all four blk groups;
//     the instruction cannot broadcast them).  All NB (NB + 1) operands are 364 registers at k = 50 -- too many next to a
//     second wavefront on the SIMD -- so they live in an LDS image (23 KB) and a step fetches the NB + 1 it needs with
//     ds_read_b128 (two operands each) one step ahead; the operands of a block are used by its urgent product and by the
//     next step's lazy products, so two register sets alternate.
//   * Per step: 3 chain + NB - 1 lazy + 1 urgent MFMAs (the next block's accumulator) and 5 VALU instructions (+ 3 while the
//     rel-change tests are on); the lazy products of the previous block's deltas fill the issue slots between the dependent
//     chain instructions.  No barrier, no scalar loads inside the sweep.
//   * Columns that are done (rel_err <= rel_tol) are written to the x image in LDS at that moment and keep being computed
//     (their lanes cost nothing); masked coordinates carry x = 0, nu = 1e300 in the loop (delta = -0) and take their value
//     from the input when a column is written.
// STRICT = false (fp32-operand mode): rows of G divided by their diagonal, delta = max(-x, -nu) -- one instruction per chain pass.
// STRICT = true (strict fp64 mode): the reference's arithmetic -- unscaled G, tmp = max(x - mu / G[q][q], 0) with the correctly
// rounded quotient (reciprocal + one Markstein correction), delta = tmp - x, x = tmp; six instructions per chain pass, the
// block's 1 / G[q][q] and G[q][q] arrive with its chain operand.  Both differ from the reference only in the order in which a
// block's four deltas are added to a gradient (the matrix core's).  The epilogue leaves the factor outputs, max|x| and the Gram
// partial sums for the next half-step behind.
#pragma once
#include "common.h"
#include "k_sweep.h"
#include <type_traits>

#define SWEEPQ_THREADS 256
#define SWEEPQ_COLS 64 // columns per workgroup, 16 per wavefront

__host__ __device__ static inline int sweepq_np(int NB, bool strict) { return (NB + (strict ? 4 : 2)) / 2; } // operand PAIRS per step
__host__ __device__ static inline size_t sweepq_img_doubles(int NB, bool strict) { return (size_t)NB * sweepq_np(NB, strict) * 32 + 4 * NB; }

// Operand image in memory ("raw": edited, not yet scaled), img[((beta * NP + p) * 16 + li) * 2 + e], li = 4 kA + iA, entry s = 2 p + e:
//   s < NB  : E[4 s + iA][4 beta + kA]                        (operand of accumulator s for the deltas of block beta)
//   s == NB : strictly lower part of E[4 bn + iA][4 bn + kA], bn = (beta + 1) % NB   (chain operand of the NEXT block)
//   strict only: s == NB + 1, NB + 2: unused here (filled in LDS: 1 / E[q][q] and E[q][q] of the next block's coordinates q = 4 bn + kA)
// followed by the diagonal E[q][q], q < 4 NB.  E = edited G (src/update_with_missing.cpp:20-24); coordinates >= k are inert (identity).
// Written by sweepq_img_put() (k_gram.h) -- from the fold of the Gram partial sums (gram_fold_kernel / factor16_fold_kernel: no launch
// of its own in the steady state of the dense flows) or by sweepq_pack_kernel below.  The sweep kernels turn it into the image they
// keep in LDS while they copy it (sweepq_load_image): fp32-operand mode: row r divided by its diagonal, G' = E[r][c] * (1 / E[r][r]),
// diagonal exactly 1; strict: E itself plus the per-coordinate constants.
static __global__ __launch_bounds__(256) void sweepq_pack_kernel(const double *__restrict__ Graw, int KPg, int k, double r0, double r1, int NB,
                                                          double *__restrict__ img, int strict)
{
    SweepImg im;
    im.img = img, im.NB = NB, im.NP = sweepq_np(NB, strict != 0), im.k = k, im.r0 = r0, im.r1 = r1;
    const int n = 4 * NB;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n; e += gridDim.x * 256) {
        const int r = e / n, c = e % n;
        sweepq_img_put(im, r, c, (r < k && c < k) ? Graw[(size_t)r * KPg + c] : 0.0);
    }
}

// memory image -> the workgroup's LDS image opl[NB * NP * 32] and rinv_l[4 NB] = 1 / E[q][q] (all threads; ends with a barrier)
template <int NB, bool STRICT> __device__ __forceinline__ void sweepq_load_image(const double *__restrict__ img, double *opl, double *rinv_l)
{
    constexpr int NP = (NB + (STRICT ? 4 : 2)) / 2;
    const int tid = threadIdx.x;
    const double *diag = img + (size_t)NB * NP * 32;
    if (tid < 4 * NB) rinv_l[tid] = 1.0 / diag[tid];
    __syncthreads();
    for (int e = tid; e < NB * NP * 16; e += SWEEPQ_THREADS) {
        f64x2 v = ((const f64x2 *)img)[e];
        const int li = e & 15, p = (e >> 4) % NP, beta = (e >> 4) / NP, kA = li >> 2, iA = li & 3, bn = (beta + 1) % NB;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int s = 2 * p + h;
            if (s < NB) {
                if (!STRICT) v[h] = (4 * s + iA == 4 * beta + kA) ? 1.0 : v[h] * rinv_l[4 * s + iA];
            } else if (s == NB) {
                if (!STRICT) v[h] = v[h] * rinv_l[4 * bn + iA]; // (strictly lower part: never the diagonal)
            } else if (STRICT && s == NB + 1) v[h] = rinv_l[4 * bn + kA];
            else if (STRICT && s == NB + 2) v[h] = diag[4 * bn + kA];
            else v[h] = 0.0;
        }
        ((f64x2 *)opl)[e] = v;
    }
    __syncthreads();
}

template <int I, int N, class F> __device__ __forceinline__ void sq_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sq_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ double sq_mfma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double sq_delta(double x, double m)
{
    double d; // max(x - m, 0) - x = max(-x, -m): one instruction (fmax() costs a canonicalisation and a negation)
    asm volatile("v_max_f64 %0, -%1, -%2" : "=v"(d) : "v"(x), "v"(m));
    return d;
}
template <int N> __device__ __forceinline__ void sq_nop()
{
    if constexpr (N > 0) asm volatile("s_nop %0" ::"n"(N - 1)); // N wait states
}

// The NL = NB - 1 lazy products of a step, dealt to its four stages: post[s] behind the stage's dependent MFMA (two cover the six
// wait states the next v_max needs), pre[s] between the v_max and that MFMA (one covers its two); off[] = first product of each slot.
struct SqSched {
    int pre[4], post[4], off[8];
};
constexpr SqSched sq_sched(int NL)
{
    SqSched s{};
    int left = NL;
    for (int i = 0; i < 4; i++) {
        s.post[i] = left < 2 ? left : 2;
        left -= s.post[i];
    }
    for (int i = 0; i < 4; i++) {
        s.pre[i] = left < 1 ? left : 1;
        left -= s.pre[i];
    }
    for (int i = 0; left > 0; i++, left--) s.post[i % 4]++;
    int o = 0;
    for (int i = 0; i < 4; i++) { // slot order pre[0] post[0] pre[1] ...: product 0 (the next block's accumulator) comes first
        s.off[2 * i] = o;
        o += s.pre[i];
        s.off[2 * i + 1] = o;
        o += s.post[i];
    }
    return s;
}

// What every sweep workgroup leaves behind from its final x image xl[columns][KP + 2] (columns col_base .. col_base + columns - 1): the
// factor outputs, max|x| and the Gram partial sums of its columns (slab `slab_idx`) for the next half-step.  COLS > 0: that many
// columns, known at compile time (plain form: 64); COLS = 0: `ncl` of them (persistent form: 16 G, a multiple of 16).
template <int NT, int COLS> __device__ __forceinline__ void sweepq_epilogue(const SweepArgs &a, const double *xl, int ncl, int col_base, int slab_idx)
{
    constexpr int KP = 16 * NT, XS = KP + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = a.k;
    const int ncols_wg = COLS ? COLS : ncl;
    float xmax = 0.0f;
    for (int e = tid; e < ncols_wg * KP; e += SWEEPQ_THREADS) {
        const int q = e / ncols_wg, c = e % ncols_wg, ecol = col_base + c;
        if (q < k && ecol < a.ncols) {
            const double xv = xl[c * XS + q];
            xmax = fmaxf(xmax, fabsf((float)xv));
            a.Xout[(size_t)q * a.ldo + (ecol - a.ocol0)] = xv;
            if (a.op_mode == 1) {
                if (a.op_f64) ((double *)a.op)[(size_t)q * a.op_ld + ecol] = xv;
                else ((float *)a.op)[(size_t)q * a.op_ld + ecol] = (float)xv;
            }
        }
    }
    if (a.maxbits) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
        if (lane == 0 && xmax > 0.0f) atomicMax(a.maxbits, __float_as_uint(xmax));
    }
    if (a.gram_slabs) {
        // Gram partial sums of this workgroup's columns, X X^T over the COLS columns with v_mfma_f64_16x16x4_f64, upper tiles dealt
        // to the four wavefronts, slab layout of gram_partial_kernel (k_gram.h); folded by factor16_fold_kernel (fence-free)
        const int l15 = lane & 15, lg = lane >> 4;
        double *slab = a.gram_slabs + (size_t)slab_idx * KP * KP;
        int tix = 0;
#pragma unroll
        for (int ta = 0; ta < NT; ta++)
#pragma unroll
            for (int tb = ta; tb < NT; tb++) {
                if ((tix++ & 3) != wave) continue;
                f64x4 g = f64x4{0, 0, 0, 0};
#pragma unroll(COLS ? COLS / 4 : 1)
                for (int s4 = 0; s4 < ncols_wg / 4; s4++) {
                    const double *xr = xl + (4 * s4 + lg) * XS + l15;
                    g = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[16 * ta], xr[16 * tb], g, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) slab[(16 * ta + lg + 4 * r) * KP + 16 * tb + l15] = g[r];
            }
    }
}

// ---- one block of the sweep: the step both kernel forms take (sweepq16_body, sweep_scd_qw_kernel) -----------------------------------
// Expands to a generic lambda (block B, rel-change tests on / off) inside a scope that holds: As[2][2 NP] (operand sets by block
// parity), acc[NB], x[NB], d_pend, Lc, rinvc, gdc, flag, tol, tolh, tolhe, fetch(), the template parameters NB, NP, STRICT -- and, with
// SWEEPQ_DEBUG, the harness's dump hook.  The step is four stages  [v_max]  pre  [dependent MFMA]  post : three chain passes and the
// urgent product.  `pre` / `post` are lazy products of the PREVIOUS block's deltas (independent of the chain): they fill the issue slots
// the dependent instructions would otherwise wait through.  The v_max is inline asm, which the compiler's hazard recogniser does not
// look into, so the wait states are provided here: a VALU result needs 2 wait states before a DGEMM reads it (one 4-pass MFMA in
// between = 4), a 4x4x4 DGEMM result 6 before a VALU reads it (two MFMAs = 8); where a block count leaves a slot without lazy products,
// s_nop stands in.  (Found the hard way: without them v_max reads the accumulator's OLD value -- every x doubled per sweep.)  The
// scheduling barriers pin the order.
#define SQ_IC(v) std::integral_constant<int, (v)> {}
#define SQ_STAGE(s_, val_in, dep_expr)                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if constexpr (STRICT) {                                                                                             \
        const double q0 = (val_in) * rinvc; /* mu / G[q][q], correctly rounded: reciprocal + one Markstein correction */ \
        const double qq = __builtin_fma(__builtin_fma(-q0, gdc, (val_in)), rinvc, q0);                                  \
        tmpx = __builtin_fmax(xb - qq, 0.0); /* src/base_algorithms.cpp:23-24 */                                         \
        c = tmpx - xb;                                                                                                  \
    } else                                                                                                              \
        c = sq_delta(xb, val_in);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    sq_nop<(S.pre[s_] == 0 ? 2 : 0)>();                                                                                 \
    lazies(SQ_IC(S.off[2 * (s_)]), SQ_IC(S.off[2 * (s_)] + S.pre[s_]));                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    dep_expr;                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    lazies(SQ_IC(S.off[2 * (s_) + 1]), SQ_IC(S.off[2 * (s_) + 1] + S.post[s_]));                                          \
    sq_nop<(S.post[s_] == 0 ? 6 : (S.post[s_] == 1 ? 2 : 0))>();                                                        \
    __builtin_amdgcn_sched_barrier(0);
#ifdef SWEEPQ_ABL_NOFETCH /* (harness ablation: timing without the operand fetches) */
#define SQ_FETCH_AC()
#else
#define SQ_FETCH_AC() fetch(bc, Ac) /* (the previous step's lazy products were the last readers of this set) */
#endif
#define SWEEPQ_STEP_LAMBDA(DBG_HOOK)                                                                                                   \
    [&](auto bc, auto tc) {                                                                                                            \
        constexpr int B = decltype(bc)::value, BN = (B + 1) % NB;                                                                      \
        constexpr bool TEST = decltype(tc)::value;                                                                                     \
        constexpr SqSched S = sq_sched(NB - 1);                                                                                        \
        double(&Ap)[2 * NP] = As[(B + 1) & 1]; /* operands of the previous block (lazy products) */                                     \
        double(&Ac)[2 * NP] = As[B & 1];       /* operands of this block: fetched now, first used by the urgent product */              \
        const double m0 = acc[B], xb = x[B];                                                                                           \
        auto lazies = [&](auto fromc, auto toc) { /* lazy products number from .. to - 1, the next block's accumulator first */         \
            sq_for<decltype(fromc)::value, decltype(toc)::value>([&](auto oc) {                                                        \
                constexpr int T = (B + 1 + decltype(oc)::value) % NB;                                                                  \
                acc[T] = sq_mfma(Ap[T], d_pend, acc[T]);                                                                               \
            });                                                                                                                        \
        };                                                                                                                             \
        double c, m, tmpx = 0.0;                                                                                                       \
        SQ_STAGE(0, m0, m = sq_mfma(Lc, c, m0))                                                                                        \
        SQ_FETCH_AC();                                                                                                                 \
        SQ_STAGE(1, m, m = sq_mfma(Lc, c, m0))                                                                                         \
        SQ_STAGE(2, m, m = sq_mfma(Lc, c, m0))                                                                                         \
        /* the fourth stage's dependent product is the urgent one: the next block's gradient (its lazy product came first above) */    \
        SQ_STAGE(3, m, acc[BN] = sq_mfma(Ac[BN], c, acc[BN]))                                                                          \
        const double d = c;                                                                                                            \
        /* rel-change test (src/base_algorithms.cpp:29-32): fp32-operand mode division-free, 2|d| > tol (x + d + x + eps); strict mode \
           the rounded quotient's decision (rel_change_exceeds, common.h: the division only where the two sides agree to ~2 ulp) */    \
        if (TEST) {                                                                                                                    \
            if constexpr (STRICT) flag |= rel_change_exceeds(2.0 * fabs(d), tmpx + xb + NNLM_TINY, tol); /* (d = 0 when tmp == Hj(k): the reference's `continue`) */ \
            else flag |= fabs(d) > __builtin_fma(tolh, __builtin_fma(2.0, xb, d), tolhe);                                              \
        }                                                                                                                              \
        x[B] = STRICT ? tmpx : xb + d; /* (strict: Hj(k) = tmp itself, src/base_algorithms.cpp:33) */                                   \
        DBG_HOOK                                                                                                                       \
        d_pend = d;                                                                                                                    \
        Lc = Ac[NB];                                                                                                                   \
        if constexpr (STRICT) rinvc = Ac[NB + 1], gdc = Ac[NB + 2];                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
    }
#ifdef SWEEPQ_DEBUG
#define SQ_DBG_DUMP                                                                                                                    \
    if (a.op_mode == 99 && t == 0 && blockIdx.x == 0 && wave == 0) {                                                                   \
        double *dbg = (double *)a.op + 1024 + B * 6 * 64;                                                                              \
        dbg[lane] = xb, dbg[64 + lane] = m0, dbg[128 + lane] = m, dbg[192 + lane] = d, dbg[256 + lane] = x[B], dbg[320 + lane] = d_pend; \
    }
#else
#define SQ_DBG_DUMP
#endif

// LDS of one 16-column-per-wavefront workgroup: the x image and the operand image
__host__ __device__ static inline size_t sweepq_lds_bytes(int KP, int NB, bool strict) { return ((size_t)SWEEPQ_COLS * (KP + 2) + (size_t)NB * sweepq_np(NB, strict) * 32 + 4 * NB) * 8; }

// NT: the caller's rank padding KP = 16 NT (layout of the outputs and of the Gram slabs); NB = ceil(k / 4) blocks.
// Workgroup blockIdx.x of the launch.  (A device function + sweepq_epilogue: scripts/exp/k_sweep_q4.h mixes it with a second
// workgroup shape in one launch -- measured, not taken.)
template <int NT, int NB, bool HAS_MASK, bool STRICT>
__device__ __forceinline__ void sweepq16_body(const SweepArgs &a, const double *__restrict__ img, unsigned char *smem)
{
    constexpr int KP = 16 * NT, NP = (NB + (STRICT ? 4 : 2)) / 2, XS = KP + 2;
    static_assert(NB <= 4 * NT && NB > 4 * (NT - 1) && NB >= 1, "NB = ceil(k / 4)");
    double *xl = (double *)smem;       // [SWEEPQ_COLS][XS]: x[column][coordinate], final values
    double *opl = xl + SWEEPQ_COLS * XS; // [NB * NP * 32]: the operand image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ri = lane >> 4, c16 = lane & 15; // row of the lane's coordinates inside their blocks; column inside the wavefront
    const int k = a.k;
    const int col_base = a.col0 + blockIdx.x * SWEEPQ_COLS;
    const int cl = 16 * wave + c16, col = col_base + cl;
    const bool in_range = col < a.ncols;
    const int cc = in_range ? col : a.col0;

    double *rinv = opl + NB * NP * 32; // [4 NB]: 1 / E[q][q]
    sweepq_load_image<NB, STRICT>(img, opl, rinv);
    unsigned long long mword = 0ull;
    if (HAS_MASK) mword = a.mask[cc];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    bool act = in_range && !(HAS_MASK && ((mword & kmask) == kmask)); // arma::all(mask.col(j)) -> column skipped
    double acc[NB], x[NB];
    // nu = ((L1 - c) + G x) / diag   (src/update_with_missing.cpp:39-41).  All NB loads of a slab are issued together (a loop over the
    // slabs INSIDE the loop over the blocks would be NB x nslabs dependent round trips to L2: 20 us of a 130 us kernel)
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int q = 4 * b + ri;
        acc[b] = 0.0;
        x[b] = (q < k && in_range) ? a.X[(size_t)q * a.ldx + col] : 0.0;
    }
    for (int s = 0; s < a.nslabs; s++) {
        const double *cs = a.Cx + (size_t)s * a.slab_stride + cc;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int q = 4 * b + ri;
            acc[b] += (q < k) ? cs[(size_t)q * a.ldc] : 0.0;
        }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int q = 4 * b + ri;
        acc[b] = (q < k) ? ((a.r2 != 0) ? a.r2 - acc[b] : -acc[b]) * (STRICT ? 1.0 : rinv[q]) : 0.0;
    }
    // this lane's operand values: lane (kA, blk, iA) reads entry li = 4 kA + iA of every operand (all four blk groups the same)
    const f64x2 *opv = (const f64x2 *)opl + (4 * (lane >> 4) + (lane & 3));
    // pairs p = 0 .. NP - 1 of block B: entries 2 p, 2 p + 1 of fetch(B) -> set[2 p], set[2 p + 1]
    auto fetch = [&](auto bc, double(&set)[2 * NP]) {
        constexpr int B = decltype(bc)::value;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const f64x2 v = opv[(B * NP + p) * 16];
            set[2 * p] = v[0];
            set[2 * p + 1] = v[1];
        }
    };
    double As[2][2 * NP]; // operand sets by block parity: As[B & 1][s], s < NB: accumulator s <- deltas of block B; [NB]: chain operand of block B + 1
    sq_for<0, NB>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        fetch(bc, As[0]);
#pragma unroll
        for (int T = 0; T < NB; T++) acc[T] = sq_mfma(As[0][T], x[B], acc[T]);
    });
    if (HAS_MASK) {
#pragma unroll
        for (int b = 0; b < NB; b++)
            if ((mword >> (4 * b + ri)) & 1ull) x[b] = 0.0, acc[b] = 1e150; // delta = max(-0, -1e150) = -0 (strict: max(0 - huge, 0) - 0) for good
    }
#ifdef SWEEPQ_DEBUG
    if (a.op_mode == 99 && blockIdx.x == 0 && wave == 0) { // harness: the initial (scaled) gradients
#pragma unroll
        for (int b = 0; b < NB; b++) ((double *)a.op)[(4 * b + ri) * 16 + c16] = acc[b];
    }
#endif
    // the column's final values -> x image (masked entries from the input; rows of out-of-range columns zero)
    auto write_col = [&]() {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int q = 4 * b + ri;
            double v = x[b];
            if (HAS_MASK && in_range && q < k && ((mword >> q) & 1ull)) v = a.X[(size_t)q * a.ldx + col];
            xl[cl * XS + q] = in_range ? v : 0.0;
        }
    };
    if constexpr (KP > 4 * NB) { // coordinates beyond the last block
        constexpr int REST = KP - 4 * NB;
        for (int e = tid; e < SWEEPQ_COLS * REST; e += SWEEPQ_THREADS) xl[(e / REST) * XS + 4 * NB + e % REST] = 0.0;
    }
    if (!act) write_col();

    const double tol = a.rel_tol, tolh = 0.5 * tol, tolhe = 0.5 * tol * NNLM_TINY;
    unsigned t = 0;
    int t_lane = 0;
    bool go = a.max_iter > 0 && __any(act);
    double d_pend = 0.0; // deltas of the previous block, still owed to every accumulator but the current block's
    bool flag = false;
    // entering step 0: As[1] = operands of block NB - 1 (lazy products of d_pend = 0: any finite values), Lc = chain operand of block 0
    fetch(std::integral_constant<int, NB - 1>{}, As[1]);
    double Lc = As[1][NB];
    double rinvc = STRICT ? As[1][NB + 1] : 0.0, gdc = STRICT ? As[1][NB + 2] : 0.0; // strict: 1 / G[q][q] and G[q][q] of the current block's coordinates
    sq_nop<8>(); // (the initial gradients come out of MFMAs; the first v_max below is inline asm)

    // One block: SWEEPQ_STEP_LAMBDA (above), shared with the persistent form
    auto step = SWEEPQ_STEP_LAMBDA(SQ_DBG_DUMP);
    // some live column of the wavefront has no coordinate yet that moved by more than rel_tol
    auto tests_needed = [&]() -> bool {
        const unsigned long long bal = __ballot(flag);
        const unsigned cf = (unsigned)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull);
        return __any(act && !((cf >> c16) & 1u));
    };
    while (go) {
        flag = 0.0 > tol; // rel_err starts each sweep at 0: a negative rel_tol never stops
        step(std::integral_constant<int, 0>{}, std::true_type{});
        if (tests_needed()) {
            sq_for<1, NB>([&](auto bc) { step(bc, std::true_type{}); });
        } else {
            sq_for<1, NB>([&](auto bc) { step(bc, std::false_type{}); });
        }
        if constexpr (NB & 1) { // the last block's operands sit in set 0; step 0 reads its lazy operands from set 1
#pragma unroll
            for (int s = 0; s < 2 * NP; s++) As[1][s] = As[0][s];
        }
        // end of a sweep (src/base_algorithms.cpp:35: stop when rel_err <= rel_tol)
        const unsigned long long bal = __ballot(flag);
        const unsigned cf = (unsigned)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull);
        if (act) {
            t_lane++;
            if (!((cf >> c16) & 1u)) {
                write_col(); // done: these are the column's final values, whatever its lanes go on computing
                act = false;
            }
        }
        t++;
        go = t < a.max_iter && __any(act);
    }
    if (act) write_col();
    __syncthreads(); // x image final

    sweepq_epilogue<NT, SWEEPQ_COLS>(a, xl, SWEEPQ_COLS, col_base, (int)blockIdx.x);
    {
        const long long tot = wave_sum_ll((ri == 0) ? (long long)t_lane : 0ll);
        if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
}

// (masked with k > 56 -- strict mode: k > 52 --: one wavefront per SIMD rather than spills)
template <int NT, int NB, bool HAS_MASK, bool STRICT>
__global__ __launch_bounds__(SWEEPQ_THREADS, ((HAS_MASK && (NB >= 15 || (STRICT && NB >= 14))) ? 1 : 2)) void sweep_scd_q_kernel(const SweepArgs a, const double *__restrict__ img)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sq_smem[]; // sweepq_lds_bytes(KP, NB, STRICT)
    sweepq16_body<NT, NB, HAS_MASK, STRICT>(a, img, sq_smem);
}

// =====================================================================================================================================
// Persistent form ("wrap"): the W half-step of the benchmark has 20000 columns = 1250 wavefronts of 16 for 1024 SIMDs.  A SIMD runs ONE
// wavefront of this kernel at full speed -- a second one adds its whole time --, so the plain launch takes two
// rounds on the 226 SIMDs that carry two wavefronts (0.205 ms) while 798 SIMDs idle through the second.  A column's sweeps are a
// sequential chain, but the chain need not stay on one wavefront: with G column groups (of 16) per workgroup and its four wavefronts
// as four machines, McNaughton's wrap-around rule for preemptive scheduling on identical machines -- lay the G chains of S sweeps end to
// end on one time line, cut it into four pieces of T = ceil(G S / 4) -- gives every wavefront T sweeps of work instead of 2 S: a group cut
// by a piece boundary has its FIRST sweeps run by the wavefront whose piece starts inside it (at the start of its time) and its LAST
// sweeps by the wavefront whose piece ends inside it (at the end of its time); in between the group's state (x, gradients, the deltas
// still owed, activity and sweep counts: 2 NB + 1 doubles per lane) waits in LDS.  The two parts never overlap in time (T >= S), the
// consumer finds the state ready unless the producer was delayed (it then spins on an LDS flag), and no wavefront of a piece waits for
// a wavefront that can wait for it: producers are first in their wavefront's order.  One workgroup per CU (the launch pads its LDS
// request so that two cannot share one), 256 threads, one wavefront per SIMD at any time: G = 5 at the benchmark's W half-step, T = 63
// sweeps instead of 100.  Same arithmetic, same order of operations per column as the plain form: results are bit-identical.
// =====================================================================================================================================
#define SWEEPQ_WRAP_MAXG 10 // (x image of 16 G columns + operand image + three hand-over slots: 154 KB at k = 64 in the strict mode)
__host__ __device__ static inline size_t sweepqw_slot_doubles(int NB) { return (size_t)(2 * NB + 1) * 64 + 64; } // per lane: x, gradients, owed deltas; (act, sweeps) as ints
__host__ __device__ static inline size_t sweepqw_lds_bytes(int KP, int NB, bool strict, int G)
{
    return ((size_t)16 * G * (KP + 2) + (size_t)NB * sweepq_np(NB, strict) * 32 + 4 * NB + 3 * sweepqw_slot_doubles(NB)) * 8 + 64;
}

template <int NT, int NB, bool HAS_MASK, bool STRICT>
__global__ __launch_bounds__(SWEEPQ_THREADS, 1) void sweep_scd_qw_kernel(const SweepArgs a, const double *__restrict__ img, int G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sq_smem[]; // sweepqw_lds_bytes(KP, NB, STRICT, G) (or more: see the launch)
    constexpr int KP = 16 * NT, NP = (NB + (STRICT ? 4 : 2)) / 2, XS = KP + 2;
    static_assert(NB <= 4 * NT && NB > 4 * (NT - 1) && NB >= 1, "NB = ceil(k / 4)");
    const int SLOT = (int)sweepqw_slot_doubles(NB);
    double *xl = (double *)sq_smem;             // [16 G][XS]: x[column][coordinate], final values
    double *opl = xl + (size_t)16 * G * XS;     // [NB * NP * 32]: the operand image
    double *rinv = opl + NB * NP * 32;          // [4 NB]: 1 / E[q][q]
    double *hand = rinv + 4 * NB;               // [3][SLOT]: slot w = state on its way from wavefront w + 1 to wavefront w
    int *ready = (int *)(hand + 3 * SLOT);      // [3]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ri = lane >> 4, c16 = lane & 15;
    const int k = a.k;
    const int col_wg = a.col0 + 16 * G * (int)blockIdx.x; // first column of the workgroup

    if constexpr (KP > 4 * NB) { // coordinates beyond the last block
        constexpr int REST = KP - 4 * NB;
        for (int e = tid; e < 16 * G * REST; e += SWEEPQ_THREADS) xl[(e / REST) * XS + 4 * NB + e % REST] = 0.0;
    }
    if (tid < 3) ready[tid] = 0;
    sweepq_load_image<NB, STRICT>(img, opl, rinv); // (ends with a barrier: operand image complete, flags clear)
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    const f64x2 *opv = (const f64x2 *)opl + (4 * (lane >> 4) + (lane & 3));
    auto fetch = [&](auto bc, double(&set)[2 * NP]) {
        constexpr int B = decltype(bc)::value;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const f64x2 v = opv[(B * NP + p) * 16];
            set[2 * p] = v[0];
            set[2 * p + 1] = v[1];
        }
    };
    double As[2][2 * NP];
    // ---- state of the group in hand -------------------------------------------------------------------------------------------
    double acc[NB], x[NB];
    double d_pend = 0.0;
    bool act = false, in_range = false;
    int t_lane = 0, col = 0, cl = 0;
    unsigned long long mword = 0ull;
    long long counted = 0; // sweeps of the columns this wavefront finished

    auto write_col = [&]() {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int q = 4 * b + ri;
            double v = x[b];
            if (HAS_MASK && in_range && q < k && ((mword >> q) & 1ull)) v = a.X[(size_t)q * a.ldx + col];
            xl[cl * XS + q] = in_range ? v : 0.0;
        }
    };
    auto select_group = [&](int gl) {
        cl = 16 * gl + c16;
        col = col_wg + cl;
        in_range = col < a.ncols;
        mword = 0ull;
        if (HAS_MASK) mword = a.mask[in_range ? col : a.col0];
    };
    // the prologue of the plain form: x, nu = ((L1 - c) + G x) / diag, masks
    auto init_fresh = [&]() {
        const int cc = in_range ? col : a.col0;
        act = in_range && !(HAS_MASK && ((mword & kmask) == kmask));
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int q = 4 * b + ri;
            acc[b] = 0.0;
            x[b] = (q < k && in_range) ? a.X[(size_t)q * a.ldx + col] : 0.0;
        }
        for (int s = 0; s < a.nslabs; s++) {
            const double *cs = a.Cx + (size_t)s * a.slab_stride + cc;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int q = 4 * b + ri;
                acc[b] += (q < k) ? cs[(size_t)q * a.ldc] : 0.0;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int q = 4 * b + ri;
            acc[b] = (q < k) ? ((a.r2 != 0) ? a.r2 - acc[b] : -acc[b]) * (STRICT ? 1.0 : rinv[q]) : 0.0;
        }
        sq_for<0, NB>([&](auto bc) {
            constexpr int B = decltype(bc)::value;
            fetch(bc, As[0]);
#pragma unroll
            for (int T = 0; T < NB; T++) acc[T] = sq_mfma(As[0][T], x[B], acc[T]);
        });
        if (HAS_MASK) {
#pragma unroll
            for (int b = 0; b < NB; b++)
                if ((mword >> (4 * b + ri)) & 1ull) x[b] = 0.0, acc[b] = 1e150;
        }
        d_pend = 0.0;
        t_lane = 0;
        if (!act) write_col();
    };
    auto store_state = [&](int slot) {
        double *hv = hand + (size_t)slot * SLOT;
#pragma unroll
        for (int b = 0; b < NB; b++) hv[b * 64 + lane] = x[b], hv[(NB + b) * 64 + lane] = acc[b];
        hv[2 * NB * 64 + lane] = d_pend;
        int *hi = (int *)(hv + (2 * NB + 1) * 64);
        hi[lane] = act ? 1 : 0;
        hi[64 + lane] = t_lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&ready[slot], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto load_state = [&](int slot) {
        while (__hip_atomic_load(&ready[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const double *hv = hand + (size_t)slot * SLOT;
#pragma unroll
        for (int b = 0; b < NB; b++) x[b] = hv[b * 64 + lane], acc[b] = hv[(NB + b) * 64 + lane];
        d_pend = hv[2 * NB * 64 + lane];
        const int *hi = (const int *)(hv + (2 * NB + 1) * 64);
        act = hi[lane] != 0;
        t_lane = hi[64 + lane];
    };

    const double tol = a.rel_tol, tolh = 0.5 * tol, tolhe = 0.5 * tol * NNLM_TINY;
    bool flag = false;
    double Lc = 0.0, rinvc = 0.0, gdc = 0.0;
    // One block: the step of the plain form (SWEEPQ_STEP_LAMBDA above)
    auto step = SWEEPQ_STEP_LAMBDA();
    auto tests_needed = [&]() -> bool {
        const unsigned long long bal = __ballot(flag);
        const unsigned cf = (unsigned)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull);
        return __any(act && !((cf >> c16) & 1u));
    };
    // sweeps [t0, t1) of the group in hand
    auto run_sweeps = [&](unsigned t0, unsigned t1) {
        unsigned t = t0;
        bool go = t < t1 && __any(act);
        fetch(std::integral_constant<int, NB - 1>{}, As[1]); // entering step 0: operands of block NB - 1, chain operand of block 0
        Lc = As[1][NB];
        rinvc = STRICT ? As[1][NB + 1] : 0.0, gdc = STRICT ? As[1][NB + 2] : 0.0;
        sq_nop<8>();
        while (go) {
            flag = 0.0 > tol;
            step(std::integral_constant<int, 0>{}, std::true_type{});
            if (tests_needed()) {
                sq_for<1, NB>([&](auto bc) { step(bc, std::true_type{}); });
            } else {
                sq_for<1, NB>([&](auto bc) { step(bc, std::false_type{}); });
            }
            if constexpr (NB & 1) {
#pragma unroll
                for (int s = 0; s < 2 * NP; s++) As[1][s] = As[0][s];
            }
            const unsigned long long bal = __ballot(flag);
            const unsigned cf = (unsigned)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull);
            if (act) {
                t_lane++;
                if (!((cf >> c16) & 1u)) {
                    write_col();
                    act = false;
                }
            }
            t++;
            go = t < t1 && __any(act);
        }
    };
    auto finish_group = [&]() { // the column's last sweeps were run here
        if (act) write_col();
        counted += (ri == 0) ? (long long)t_lane : 0ll;
    };

    // ---- this wavefront's piece [lo, hi) of the time line of G groups x S sweeps ------------------------------------------------
    const long long S = (long long)a.max_iter, T = (S * G + 3) / 4;
    const long long lo = (long long)wave * T, hi_raw = lo + T, tot = S * G;
    const long long hi = hi_raw < tot ? hi_raw : tot;
    // (ONE loop over the wavefront's parts of groups with ONE call of the sweep loop: each call site would be another inlined copy of it)
    for (long long pos = lo; pos < hi;) {
        const int gl = (int)(pos / S);
        const long long gbeg = (long long)gl * S, gend = gbeg + S;
        const bool head = pos != gbeg;            // the piece starts inside the group: its FIRST sweeps, then the state goes to the wavefront below
        const long long end = gend < hi ? gend : hi;
        const bool tail = !head && end != gend;   // the piece ends inside the group: its LAST sweeps, on the state the wavefront above leaves behind
        select_group(gl);
        unsigned t0 = 0u, t1 = (unsigned)S;
        if (tail) {
            load_state(wave);
            t0 = (unsigned)(S - (end - pos));
        } else {
            init_fresh();
            if (head) t1 = (unsigned)(gend - pos);
        }
        run_sweeps(t0, t1);
        if (head) store_state(wave - 1);
        else finish_group();
        pos = head ? gend : end;
    }
    __syncthreads(); // x image final

    sweepq_epilogue<NT, 0>(a, xl, 16 * G, col_wg, (int)blockIdx.x);
    {
        const long long totc = wave_sum_ll(counted);
        if (lane == 0 && totc) atomicAdd(a.sweeps, (unsigned long long)totc);
    }
}
