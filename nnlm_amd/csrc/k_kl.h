// k_kl.h -- per-column solvers for the KL-divergence methods: kl_reg64_kernel (strict fp64 mode), kl_tile_kernel (fp32-operand
// mode), kl_stream_kernel (no size limits).
//
// Reference: scd_kl_update (src/base_algorithms.cpp:71-116) and lee_kl_update (src/base_algorithms.cpp:119-151) as
// called from update() (src/update_with_missing.cpp:47-49) and, with the contraction restricted to the finite
// entries of the column, from update_with_missing() (src/update_with_missing.cpp:118-133).
//
// A 512-thread block owns one or a few columns j of the factor being solved.  The length-p state vector of a column
// (y = Yt^T x, "Ajt"/"wh" in the reference) and the data column b stay in registers; the k coordinates are visited
// sequentially (loop-carried, exactly as in the reference) and each visit is
//   one read of row q of the fixed factor  ->  per-thread partial sums  ->  block reduction
//   ->  the scalar update  ->  rank-1 refresh of y in registers.
// Missing entries (NA path) simply carry weight 0 (they are absent from Wt.cols(non_missing) in the
// reference); sumW is summed over the same index set inside the same reduction (src/update_with_missing.cpp:122,130).
// Arithmetic is n*m*k fp64 divides per sweep: VALU/transcendental bound, not HBM bound.
#pragma once
#include "common.h"
#include <type_traits>


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
    int ncols, k;    // ncols: END of the column range solved (exclusive)
    double r0, r1, r2;
    const unsigned long long *mask;
    unsigned max_iter;
    double rel_tol;
    void *op;
    int op_mode, op_ld, op_f64;
    unsigned long long *sweeps;
    // multi-GPU column shards: this launch solves columns col0 .. ncols-1 and writes entry (q, col) to Xout[q*ldo + col - ocol0]
    // (a packed per-rank slab for the all-gather); single GPU: col0 = ocol0 = 0, ldo = ldx
    int col0 = 0, ldo = 0, ocol0 = 0;
};

// ==================================================================================================================
// kl_reg64_kernel -- the KL solvers of the strict fp64 mode (replaces kl_update_kernel<double, EPT, METHOD>, which kept the row
// of the fixed factor in registers between its two passes: 256 VGPRs + up to 1152 bytes of scratch per lane, 52 ms per half-step
// at config 3).  Same iteration, the reference's arithmetic (fp64 state, correctly rounded quotients), organised like
// kl_tile_kernel:
//   * a 512-thread block owns C columns; the state vector y and the data column b of each column stay in registers as double2
//     chunks (thread t owns chunks t, t + 512, ...: 16-byte loads from contraction-contiguous layouts -- A / What64 for the H
//     half-step, the transposed copy AT / What64^T for the W half-step); C * EPT2 <= 20 chunks = 160 state registers;
//   * the starting states y = Yt^T x of ALL columns come from one fp64 GEMM (wh_store64_kernel), not from k passes per column;
//   * row q of the fixed factor is read ONCE per step from L2 (the master, fp64) in pass A, parked in LDS (16 bytes per slot,
//     every thread reads back only its own slots: no barrier) and picked up again by pass B -- the row is 160 KB at p = 20000,
//     the LDS holds the first ELDS chunks (144 KB), the last ones stay in registers;
//   * b / (y + eps) = b r (1 + e + e^2 ...) from v_rcp_f64 + two Newton steps + one residual correction: the correctly rounded
//     quotient for these operands (y + eps >= 1e-16, no scaling cases) in 8 instructions instead of the 14 of the IEEE sequence;
//   * the row sums of the fixed factor come precomputed (kl_sumw_kernel / kl_sumw_cols_kernel), missing entries carry b = 0.
// fp64 VALU bound: ~12 instructions per element and coordinate = 3.7 ms per half-step at config 3 at the full fp64 rate.
struct Kl64Args {
    const double *Adata; // column c at Adata + c * lda, contraction index contiguous
    size_t lda;          // (even)
    const double *Yinit; // same layout: starting state vectors (wh_store64_kernel)
    const double *Y;     // [k][ldy] master of the fixed factor, contraction index contiguous
    int ldy;
    int p, ncols, k;
    const double *X;
    double *Xout;
    int ldx;
    int colbase, ldo, ocol0;
    const double *sumw;      // [k]
    const double *sumw_cols; // [ncols][ldsw] or NULL
    int ldsw;
    double r0, r1, r2;
    const unsigned long long *mask; // [ncols] (rank <= 64) or NULL
    unsigned max_iter;
    double rel_tol;
    void *op; // fp64 operand copy [KP][op_ld] (op_mode 1) or none
    int op_mode, op_ld;
    unsigned long long *sweeps;
};
#define KL64_THREADS 512
#define KL64_ELDS_MAX 19 // chunks of a row the LDS holds (19 * 8 KB = 152 KB; the 20th stays in registers)
__host__ __device__ static inline size_t kl64_lds_bytes(int EPT2, int C, int k)
{
    const int elds = EPT2 < KL64_ELDS_MAX ? EPT2 : KL64_ELDS_MAX;
    return (size_t)elds * KL64_THREADS * 16 + (size_t)2 * C * k * 8 + (size_t)2 * 2 * C * 8 * 8;
}
// a block-uniform double as a scalar (SGPR pair): the per-column bookkeeping is identical in every lane, and 160 state registers
// leave no room for per-lane copies of it
__device__ static inline double kl64_uni(double v)
{
    int2 p = __builtin_bit_cast(int2, v);
    p.x = __builtin_amdgcn_readfirstlane(p.x);
    p.y = __builtin_amdgcn_readfirstlane(p.y);
    return __builtin_bit_cast(double, p);
}
// correctly rounded a / d for d >= 1e-16 finite, |a| moderate: reciprocal, two Newton steps, one residual correction
__device__ static inline double kl64_div(double a, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-d, q, a), r, q);
}

template <int I, int N, class F> __device__ __forceinline__ void klq_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        klq_for<I + 1, N>(f);
    }
}

template <int EPT2, int C, int METHOD>
__global__ __launch_bounds__(KL64_THREADS) void kl_reg64_kernel(const Kl64Args a)
{
    constexpr int NV = (METHOD == 4) ? 1 : 2;
    constexpr int ELDS = EPT2 < KL64_ELDS_MAX ? EPT2 : KL64_ELDS_MAX, EREG = EPT2 - ELDS;
    // SCD (two sums per column, their products) with two columns on 160 state registers does not fit 256: the allocator kept eleven
    // chunks of b in scratch and reloaded them in every step, each reload a full vmcnt wait.  There the LAST chunks of the data columns
    // (e >= EB) are not resident: pass A requests chunk e + BD from L2 while it works on chunk e (the block's re-read part of A stays
    // in its XCD's L2).  (One column of 20 chunks: streaming made the allocator's choices worse -- see the pinned LDS reads below.)
    constexpr bool BSTREAM = (METHOD == 3 && C == 2 && EPT2 >= 9);
    constexpr int EB = BSTREAM ? EPT2 - 5 : EPT2;                        // first streamed chunk
    constexpr int BD = 2;                                                // chunks ahead
    static_assert(!BSTREAM || (EB >= BD && EPT2 - EB >= BD), "streamed chunks of b");
    extern __shared__ __attribute__((aligned(16))) unsigned char kl64_smem[];
    f64x2 *rowl = (f64x2 *)kl64_smem;                                  // [ELDS][512] this step's row, parked between the passes
    double *xs = (double *)(kl64_smem + (size_t)ELDS * KL64_THREADS * 16); // [C][k]
    double *sws = xs + C * a.k;                                        // [C][k]
    double *red = sws + C * a.k;                                       // [2][NV * C][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave-uniform: scalar address arithmetic)
    const int k = a.k, p = a.p, P2 = (p + 1) / 2; // double2 chunks of a row that hold data
    const int col0 = a.colbase + blockIdx.x * C;

    for (int e = tid; e < C * k; e += KL64_THREADS) {
        const int c = e / k, q = e - c * k, col = col0 + c;
        xs[e] = (col < a.ncols) ? a.X[(size_t)q * a.ldx + col] : 0.0;
        sws[e] = (col < a.ncols) ? (a.sumw_cols ? a.sumw_cols[(size_t)col * a.ldsw + q] : a.sumw[q]) : 1.0;
    }
    f64x2 y[C][EPT2], b[C][EB];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int col = (col0 + c < a.ncols) ? col0 + c : col0;
        const f64x2 *Ac = (const f64x2 *)(a.Adata + (size_t)col * a.lda), *Yc = (const f64x2 *)(a.Yinit + (size_t)col * a.lda);
#pragma unroll
        for (int e = 0; e < EPT2; e++) {
            const int i2 = e * KL64_THREADS + tid;
            const bool valid = i2 < P2 && col0 + c < a.ncols;
            if (e < EB) b[c][e < EB ? e : 0] = valid ? Ac[i2] : f64x2{0.0, 0.0};
            y[c][e] = valid ? Yc[i2] : f64x2{1.0, 1.0};
            if (valid && 2 * i2 + 1 >= p) {                                   // (odd contraction length: the pad element)
                if (e < EB) b[c][e < EB ? e : 0][1] = 0.0;
                y[c][e][1] = 1.0;
            }
        }
    }
    // per-column bookkeeping, identical in every thread (block-uniform)
    bool live[C];
    unsigned long long mword[C];
    double S[C];
    unsigned tdone[C];
    bool run[C], flag[C];
    __syncthreads();
    bool any = false;
#pragma unroll
    for (int c = 0; c < C; c++) {
        live[c] = col0 + c < a.ncols;
        mword[c] = (a.mask && live[c]) ? a.mask[col0 + c] : 0ull;
        const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
        if (a.mask && (mword[c] & kmask) == kmask) live[c] = false; // all coordinates masked: 0 sweeps (src/update_with_missing.cpp:33)
        S[c] = 0.0;
        for (int q = 0; q < k; q++) S[c] += xs[c * k + q];
        S[c] = kl64_uni(S[c]);
        tdone[c] = 0;
        run[c] = live[c] && a.max_iter > 0 && (1.0 + a.rel_tol) > a.rel_tol;
        flag[c] = false;
        any = any || run[c];
    }
    // The row of the fixed factor comes through LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, so all ELDS chunks of
    // the NEXT row are in flight while pass B and the reduction of this step run -- with register-destination loads two chunks
    // ahead the kernel sat at 6.5 TB/s of L2 reads, latency x bytes in flight).  One row buffer: chunk e of the next row is requested
    // into its slot right after pass B has read chunk e of this row back; every thread reads only slots its own wavefront loaded, so
    // `s_waitcnt vmcnt(0)` in front of pass A is all the synchronisation the rows need.  The instruction is written out (scalar base,
    // one per-lane offset register, LDS base in M0): the compiler's wait-count pass would otherwise put a vmcnt(0) in front of every
    // LDS read that follows it.  Chunks beyond the LDS (EREG of them) are plain loads at the top of pass A, used at its end.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)kl64_smem + (unsigned)wave * 1024u;
    const int voff = lane * 16, L2c = a.ldy / 2; // (chunks at or beyond L2c lie outside the row: never loaded, their slots stay zero)
    for (int i = tid; i < ELDS * KL64_THREADS; i += KL64_THREADS)
        if (i >= L2c) rowl[i] = f64x2{0.0, 0.0};
    auto issue_chunk = [&](int qq, int e) {
        const unsigned char *src = (const unsigned char *)(a.Y + (size_t)qq * a.ldy) + (size_t)e * (KL64_THREADS * 16) + (size_t)wave * 1024; // wave-uniform
        const unsigned long long sp = (unsigned long long)src;
        const unsigned long long su = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sp >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)sp);
        const unsigned du = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)e * (KL64_THREADS * 16)));
        if (e * KL64_THREADS + tid < L2c) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(su), "s"(du) : "memory");
    };
    auto issue_row = [&](int qq) {
#pragma unroll
        for (int e = 0; e < ELDS; e++) issue_chunk(qq, e);
    };
    __syncthreads(); // (the zeroed slots)
    int par = 0;
    if (any) issue_row(0);
    while (any) {
#pragma unroll
        for (int c = 0; c < C; c++) flag[c] = 0.0 > a.rel_tol; // rel_err starts each sweep at 0 (src/base_algorithms.cpp:93,137)
        for (int q = 0; q < k; q++) {
            const int qn = (q + 1 < k) ? q + 1 : 0; // next row (row 0 again for a sweep that may follow)
            bool doq[C], anyq = false;
#pragma unroll
            for (int c = 0; c < C; c++) {
                doq[c] = run[c] && !((mword[c] >> q) & 1ull);
                anyq = anyq || doq[c];
            }
            if (!anyq) { // no column of the block visits this coordinate: straight to the next row
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (row q has landed: its slots can be requested again)
                issue_row(qn);
                continue;
            }
            const f64x2 *Yq = (const f64x2 *)(a.Y + (size_t)q * a.ldy);
            // (the LDS slot addresses are recomputed from this opaque copy in every step: as loop invariants the allocator kept one
            //  address register per 64 KB window alive next to 160 state registers and spilled them -- and a scratch reload behind
            //  the LDS-DMAs of pass B is a vmcnt(0), i.e. a wait for the whole row)
            int tl = tid;
            asm volatile("" : "+v"(tl));
            double xq[C];
#pragma unroll
            for (int c = 0; c < C; c++) xq[c] = kl64_uni(xs[c * k + q]); // read BEFORE the barrier: thread 0 rewrites it after
            double v[C][NV];
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int u = 0; u < NV; u++) v[c][u] = 0.0;
            f64x2 wreg[EREG > 0 ? EREG : 1];
#pragma unroll
            for (int e = ELDS; e < EPT2; e++) {
                const int i2 = e * KL64_THREADS + tid;
                wreg[e - ELDS] = (i2 < L2c) ? Yq[i2] : f64x2{0.0, 0.0};
            }
            // pass A.  The quotient is a chain of eight dependent fp64 instructions (~10 cycles each): four of them -- two (chunk,
            // column) pairs x the two elements of a chunk -- are carried through the chain together, step by step, and the sums go to
            // two partial accumulators per column (element 0 / element 1), or two wavefronts per SIMD spend the pass waiting on latency
            // (first version: one quotient after the other into one accumulator, 10.9 ms per half-step at config 3 against a 3.7 ms
            // issue bound).
            double v2[C][NV]; // second partial accumulators (element 1 of every chunk)
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int u = 0; u < NV; u++) v2[c][u] = 0.0;
            // chunk e of row q was requested e-th of ELDS LDS-DMAs (in order, during the previous step's pass B), the EREG plain loads
            // above come behind them: a COUNTED wait lets pass A start on chunk 0 while the rest of the row is still on its way
            // (one vmcnt(0) in front of the pass left the whole row fetch -- 160 KB per block from L2 -- exposed: 6.2 us per step)
            // (streamed chunks of b: requests issued before chunk e's row wait are younger than every row request and stay outstanding)
            auto wfetch = [&](auto ec) -> f64x2 {
                constexpr int e = decltype(ec)::value;
                constexpr int nbs = BSTREAM ? C * ((e < EPT2 - BD ? e : EPT2 - BD) > EB - BD ? (e < EPT2 - BD ? e : EPT2 - BD) - (EB - BD) : 0) : 0;
                if constexpr (e < ELDS) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ELDS - 1 - e + EREG + nbs) : "memory");
                    return rowl[e * KL64_THREADS + tl];
                } else
                    return wreg[e - ELDS];
            };
            f64x2 bs[C][BD + 1];
            auto brequest = [&](auto ec) { // chunk e + BD of the data column(s), unconditionally (clamped: a conditional load is sunk to its use)
                constexpr int en = decltype(ec)::value + BD;
                if constexpr (BSTREAM && en >= EB && en < EPT2) {
                    const int i2 = en * KL64_THREADS + tl, i2c = i2 < P2 ? i2 : P2 - 1;
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const int col = (col0 + c < a.ncols) ? col0 + c : col0;
                        bs[c][en % (BD + 1)] = ((const f64x2 *)(a.Adata + (size_t)col * a.lda))[i2c];
                    }
                }
            };
            auto bget = [&](int c, auto ec) -> f64x2 {
                constexpr int e = decltype(ec)::value;
                if constexpr (e < EB) return b[c][e];
                else {
                    const int i2 = e * KL64_THREADS + tl;
                    f64x2 r = bs[c][e % (BD + 1)];
                    const bool valid = i2 < P2 && col0 + c < a.ncols;
                    r[0] = valid ? r[0] : 0.0;
                    r[1] = (valid && 2 * i2 + 1 < p) ? r[1] : 0.0;
                    return r;
                }
            };
            auto pair4 = [&](int ea, int ca, const f64x2 &wa, const f64x2 &ba, int eb, int cb, const f64x2 &wb_, const f64x2 &bb_) {
                double num[4], den[4], r[4], t[4], qv[4];
                const double wv[4] = {wa[0], wa[1], wb_[0], wb_[1]};
                const double bv[4] = {ba[0], ba[1], bb_[0], bb_[1]};
                den[0] = y[ca][ea][0] + NNLM_TINY, den[1] = y[ca][ea][1] + NNLM_TINY, den[2] = y[cb][eb][0] + NNLM_TINY, den[3] = y[cb][eb][1] + NNLM_TINY;
#pragma unroll
                for (int i = 0; i < 4; i++) num[i] = (METHOD == 4) ? bv[i] : wv[i]; // Lee: Aj / (wh + eps), :141; SCD: mu = w / (Ajt + eps), :97
#pragma unroll
                for (int i = 0; i < 4; i++) r[i] = __builtin_amdgcn_rcp(den[i]);
#pragma unroll
                for (int i = 0; i < 4; i++) t[i] = __builtin_fma(-den[i], r[i], 1.0);
#pragma unroll
                for (int i = 0; i < 4; i++) r[i] = __builtin_fma(t[i], r[i], r[i]);
#pragma unroll
                for (int i = 0; i < 4; i++) t[i] = __builtin_fma(-den[i], r[i], 1.0);
#pragma unroll
                for (int i = 0; i < 4; i++) r[i] = __builtin_fma(t[i], r[i], r[i]);
#pragma unroll
                for (int i = 0; i < 4; i++) qv[i] = num[i] * r[i];
#pragma unroll
                for (int i = 0; i < 4; i++) t[i] = __builtin_fma(-den[i], qv[i], num[i]);
#pragma unroll
                for (int i = 0; i < 4; i++) qv[i] = __builtin_fma(t[i], r[i], qv[i]); // the correctly rounded quotient
                if constexpr (METHOD == 4) {
                    v[ca][0] = __builtin_fma(wv[0], qv[0], v[ca][0]);
                    v2[ca][0] = __builtin_fma(wv[1], qv[1], v2[ca][0]);
                    v[cb][0] = __builtin_fma(wv[2], qv[2], v[cb][0]);
                    v2[cb][0] = __builtin_fma(wv[3], qv[3], v2[cb][0]);
                } else {
                    double bu[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) bu[i] = bv[i] * qv[i];
                    v[ca][0] = __builtin_fma(bu[0], qv[0], v[ca][0]), v[ca][1] += bu[0];     // a, :98; b, :99
                    v2[ca][0] = __builtin_fma(bu[1], qv[1], v2[ca][0]), v2[ca][1] += bu[1];
                    v[cb][0] = __builtin_fma(bu[2], qv[2], v[cb][0]), v[cb][1] += bu[2];
                    v2[cb][0] = __builtin_fma(bu[3], qv[3], v2[cb][0]), v2[cb][1] += bu[3];
                }
            };
            if constexpr (C == 1 && EPT2 >= 18) { // 160 state registers: one chunk (two chains) at a time, or the allocator spills
                klq_for<0, EPT2>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    const f64x2 w = wfetch(ec);
                    brequest(ec);
                    const f64x2 be = bget(0, ec);
                    double den[2] = {y[0][e][0] + NNLM_TINY, y[0][e][1] + NNLM_TINY}, num[2], r[2], t[2], qv[2];
#pragma unroll
                    for (int i = 0; i < 2; i++) num[i] = (METHOD == 4) ? be[i] : w[i];
#pragma unroll
                    for (int i = 0; i < 2; i++) r[i] = __builtin_amdgcn_rcp(den[i]);
#pragma unroll
                    for (int i = 0; i < 2; i++) t[i] = __builtin_fma(-den[i], r[i], 1.0);
#pragma unroll
                    for (int i = 0; i < 2; i++) r[i] = __builtin_fma(t[i], r[i], r[i]);
#pragma unroll
                    for (int i = 0; i < 2; i++) t[i] = __builtin_fma(-den[i], r[i], 1.0);
#pragma unroll
                    for (int i = 0; i < 2; i++) r[i] = __builtin_fma(t[i], r[i], r[i]);
#pragma unroll
                    for (int i = 0; i < 2; i++) qv[i] = num[i] * r[i];
#pragma unroll
                    for (int i = 0; i < 2; i++) t[i] = __builtin_fma(-den[i], qv[i], num[i]);
#pragma unroll
                    for (int i = 0; i < 2; i++) qv[i] = __builtin_fma(t[i], r[i], qv[i]);
                    if constexpr (METHOD == 4) {
                        v[0][0] = __builtin_fma(w[0], qv[0], v[0][0]);
                        v2[0][0] = __builtin_fma(w[1], qv[1], v2[0][0]);
                    } else {
                        const double bu0 = be[0] * qv[0], bu1 = be[1] * qv[1];
                        v[0][0] = __builtin_fma(bu0, qv[0], v[0][0]), v[0][1] += bu0;
                        v2[0][0] = __builtin_fma(bu1, qv[1], v2[0][0]), v2[0][1] += bu1;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else if constexpr (C == 1) {
                static_assert(C != 1 || EPT2 % 2 == 0, "C = 1 instantiations take the chunks in pairs");
                klq_for<0, EPT2 / 2>([&](auto gc) {
                    constexpr int e = 2 * decltype(gc)::value;
                    const f64x2 wa = wfetch(std::integral_constant<int, e>{}), wb_ = wfetch(std::integral_constant<int, e + 1>{});
                    pair4(e, 0, wa, b[0][e], e + 1, 0, wb_, b[0][e + 1]); // (EPT2 < 18: every chunk of b resident)
                    __builtin_amdgcn_sched_barrier(0); // (keeps the compiler from hoisting later chunks' LDS reads: 160 state registers leave no room)
                });
            } else {
                klq_for<0, EPT2>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    const f64x2 w = wfetch(ec);
                    brequest(ec);
#pragma unroll
                    for (int c = 0; c < C; c += 2) pair4(e, c, w, bget(c, ec), e, c + 1, w, bget(c + 1, ec));
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int u = 0; u < NV; u++) v[c][u] += v2[c][u];
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int u = 0; u < NV; u++) {
                    const double t = wave_sum(v[c][u]);
                    if (lane == 0) red[((par * C + c) * NV + u) * 8 + wave] = t;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // (LDS traffic only: __syncthreads() would also wait for nothing here, but keeps the habit of kl_tile_kernel)
            asm volatile("" ::: "memory");
            double coef[C];
#pragma unroll
            for (int c = 0; c < C; c++) {
                double sv[NV];
#pragma unroll
                for (int u = 0; u < NV; u++) {
                    const double *rr = red + ((par * C + c) * NV + u) * 8;
                    sv[u] = ((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]));
                }
                const double sw = sws[c * k + q];
                coef[c] = 0.0;
                if (METHOD == 4) {
                    double tmp = kl64_uni(sv[0] / (sw + a.r0 * xq[c] + a.r1 * (S[c] - xq[c]) + a.r2)); // :142
                    if (doq[c]) {
                        coef[c] = (tmp - 1) * xq[c];                                         // :143
                        S[c] += (tmp - 1) * xq[c];                                           // :144
                        if (tid == 0) xs[c * k + q] = xq[c] * tmp;                           // :145
                        tmp = 2 * fabs(tmp - 1) / (tmp + 1);                                 // :146
                        flag[c] = flag[c] || tmp > a.rel_tol;
                    }
                } else {
                    double aa = sv[0], bb = sv[NV - 1] - sw; // b = dot(Aj, mu) - sumW(k), :99
                    aa += a.r0;                              // :100
                    bb += aa * xq[c] - a.r2 - a.r1 * (S[c] - xq[c]); // :101
                    double tmp = kl64_uni(bb / (aa + NNLM_TINY)); // :102
                    if (tmp < 0) tmp = 0;
                    if (doq[c] && tmp != xq[c]) {
                        coef[c] = tmp - xq[c];
                        const double er = 2 * fabs(xq[c] - tmp) / (tmp + xq[c] + NNLM_TINY); // :107
                        flag[c] = flag[c] || er > a.rel_tol;
                        S[c] += tmp - xq[c];
                        if (tid == 0) xs[c * k + q] = tmp;
                    }
                }
            }
            par ^= 1;
            // pass B: y += coef * w (:106, :143); the row from LDS in groups of two chunks, each group's slots refilled with the
            // next row as soon as it has been read back
            constexpr int GB = (C == 1 && EPT2 >= 18) ? 1 : 2; // chunks per group (one where 160 state registers leave no room for two)
#pragma unroll
            for (int e0 = 0; e0 < EPT2; e0 += GB) {
                f64x2 wb[GB];
#pragma unroll
                for (int u = 0; u < GB; u++) {
                    const int e = e0 + u;
                    if (e < EPT2) wb[u] = (e < ELDS) ? rowl[e * KL64_THREADS + tl] : wreg[e < ELDS ? 0 : e - ELDS];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < GB; u++)
                    if (e0 + u < ELDS) issue_chunk(qn, e0 + u);
#pragma unroll
                for (int u = 0; u < GB; u++) {
                    const int e = e0 + u;
                    if (e < EPT2) {
#pragma unroll
                        for (int c = 0; c < C; c++) {
                            y[c][e][0] = __builtin_fma(coef[c], wb[u][0], y[c][e][0]);
                            y[c][e][1] = __builtin_fma(coef[c], wb[u][1], y[c][e][1]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads(); // xs[] written by thread 0 during this sweep is read by everyone in the next
        any = false;
#pragma unroll
        for (int c = 0; c < C; c++) {
            if (run[c]) {
                tdone[c]++;
                run[c] = tdone[c] < a.max_iter && flag[c];
            }
            any = any || run[c];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the row requested for a sweep that did not follow
    __syncthreads();
    for (int e = tid; e < C * k; e += KL64_THREADS) {
        const int c = e / k, q = e - c * k, col = col0 + c;
        if (col < a.ncols) {
            const double xv = xs[e];
            a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xv;
            if (a.op_mode == 1) ((double *)a.op)[(size_t)q * a.op_ld + col] = xv;
        }
    }
    if (tid == 0) {
        unsigned long long tot = 0;
#pragma unroll
        for (int c = 0; c < C; c++) tot += tdone[c];
        if (tot) atomicAdd(a.sweeps, tot);
    }
}

// What64[j][i] = sum_q W[q][i] H[q][j] in fp64, stored in the layout of A ([cols][lda], contraction index contiguous): the starting
// state vectors of a strict-mode KL half-step (src/base_algorithms.cpp:81, :129) as one GEMM.  64 x 64 tile per block,
// v_mfma_f64_16x16x4_f64 with M = column j, N = row i: a lane group stores 16 consecutive i (128 bytes).  For the W half-step the
// caller swaps the roles of the factors (output rows = rows of A).
__global__ __launch_bounds__(256) void wh_store64_kernel(const double *__restrict__ Wm, int ldw, const double *__restrict__ Hm, int ldh, int k4,
                                                         double *__restrict__ What, size_t lda, int ni, int nj)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int ib = blockIdx.x * 64 + 32 * (wave & 1), jb = blockIdx.y * 64 + 32 * (wave >> 1);
    f64x4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int z = 0; z < 2; z++) acc[x][z] = f64x4{0, 0, 0, 0};
    for (int kq = lg; kq < k4; kq += 4) {
        double wa[2], hb[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            wa[t] = Wm[(size_t)kq * ldw + ib + 16 * t + l15];
            hb[t] = Hm[(size_t)kq * ldh + jb + 16 * t + l15];
        }
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
            for (int z = 0; z < 2; z++) acc[x][z] = __builtin_amdgcn_mfma_f64_16x16x4f64(hb[z], wa[x], acc[x][z], 0, 0, 0);
    }
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int z = 0; z < 2; z++) {
            const int i = ib + 16 * x + l15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = jb + 16 * z + lg + 4 * r;
                if (i < ni && j < nj) What[(size_t)j * lda + i] = acc[x][z][r];
            }
        }
}

// ==================================================================================================================
// kl_tile_kernel -- the KL solvers of the fp32-operand mode (replaces kl_fast_kernel).
//
// What bounded kl_fast_kernel (config 3, 34 ms per iteration): 40 row elements + 160 state registers per thread spilled,
// every row of the fixed factor was fetched from L2 by plain per-lane loads that could not be issued ahead, the data
// column and the state of the W half-step were read with a stride, and every coordinate step reduced NV*C fp64 values
// with 6 two-dword shuffles each.  Here:
//   * a 512-thread block owns C columns; the state vector y (+eps folded in once) and the data column b of each column
//     stay in registers as float4 chunks (thread t owns float4s t, t+512, ...: coalesced 16-byte loads, both from
//     contraction-contiguous layouts: A / What for the H half-step, their transposed fp32 copies for the W half-step);
//   * row q of the fixed factor is staged in LDS by global_load_lds one coordinate step AHEAD (two row buffers; every
//     thread reads back exactly the slots its own wavefront loaded, so a counted s_waitcnt vmcnt is the only
//     synchronisation the rows need) and serves all C columns and both passes of the step; a row is padded to whole
//     64 x 16-byte wavefront pieces so that "does this piece exist" is a scalar branch, never a per-lane select;
//   * pass A: r = v_rcp_f32(y), s += w * (b * r) (Lee) on float4 lanes, 4 partial sums per column; the wave sum is 6 DPP
//     adds in fp32, the 8 wave totals are added in fp64; ONE barrier per coordinate step; pass B: y += coef * w;
//   * the row sums of the fixed factor ("sumW") do not depend on the state: one k-vector per half-step (dense) or one
//     per column over its non-missing entries (kl_sumw_cols_kernel) -- the missing entries themselves need no predicate
//     in the loop (b = 0 there, so they add nothing to the sums the reference restricts to non_missing);
//   * rank is not limited by a 64-bit word: coordinates live in LDS, masks are mw words per column.
// |y| in the quotient (a free source modifier): a state entry that rounding pushed below zero cannot flip a sign.
// VALU bound: 3 plain + 1 transcendental instruction per element and coordinate = 14.9-17.7 cycles per 64 elements per
// SIMD measured in isolation (scripts/exp/valu_exp.hip), i.e. ~1.0 ms per half-step at config 3.
// Arithmetic differences from kl_update_kernel: fp32 state and quotients as kl_fast_kernel; tmp = num / den through
// v_rcp_f64 + one Newton step; the rel-change test 2|d|/(..) > tol without the division.
#ifndef KLT_TIMING
#define KLT_TIMING 0 // 1: wavefront 0 of every workgroup accumulates the cycles it spends in each part of a step (scripts/exp/klt_exp.hip)
#endif
#ifndef KLT_V
#define KLT_V 7 // step-loop arrangement of kl_tile_kernel (bits; 0 = the arrangement of rounds 2-4): 1 Lee's reciprocal denominator between
                // the first chunks of pass A, row sum read at the top of the step; 2 wave totals of all columns side by side; 4 pass B's
                // first row elements requested in front of the scalar part
#endif
struct KlTileArgs {
    const float *Adata; // column c at Adata + c * lda, contraction index contiguous
    size_t lda;         // (= ldyf, a multiple of 4)
    const float *Yinit; // same layout: starting state vectors y = Yt^T x of all columns (wh_store_kernel), or NULL (two row buffers only):
                        // the block forms them itself from the rows of the fixed factor before its first step
    const float *Yf;    // [k][ldyf] fp32 fixed factor, contraction index contiguous, zero beyond p
    int ldyf;
    int p, ncols, k;
    const double *X;
    double *Xout;
    int ldx;
    int colbase, ldo, ocol0; // first column of this launch; entry (q, col) goes to Xout[q*ldo + col - ocol0] (KlArgs)
    const double *sumw;      // [k]: sum over the contraction index of row q of the fixed factor, or
    const double *sumw_cols; // [ncols][ldsw]: the same restricted to the non-missing entries of each column (NULL: dense)
    int ldsw;
    double r0, r1, r2;
    const unsigned long long *mask; // [ncols][mw] or NULL
    int mw;
    unsigned max_iter;
    double rel_tol;
    void *op;
    int op_mode, op_ld;
    unsigned long long *sweeps;
#if KLT_TIMING
    unsigned long long *tim = nullptr; // [blocks][8] cycle counts of wavefront 0 (experiments only)
#endif
};

template <int CTRL, int RMASK> __device__ static inline float kl_dpp(float oldv, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv), __builtin_bit_cast(int, v), CTRL, RMASK, 0xF, false));
}
// sum over the wavefront, valid in lane 63
__device__ static inline float kl_wave_total(float v)
{
    v += kl_dpp<0xB1, 0xF>(0.f, v);  // quad_perm [1,0,3,2]
    v += kl_dpp<0x4E, 0xF>(0.f, v);  // quad_perm [2,3,0,1]
    v += kl_dpp<0x141, 0xF>(0.f, v); // row_half_mirror
    v += kl_dpp<0x140, 0xF>(0.f, v); // row_mirror: every lane of a row holds the row sum
    v += kl_dpp<0x142, 0xA>(0.f, v); // row_bcast15 into rows 1 and 3
    v += kl_dpp<0x143, 0xC>(0.f, v); // row_bcast31 into rows 2 and 3
    return v;
}
#define KLT_THREADS 512
#ifndef KLT_PBD1
#define KLT_PBD1 1 // the same in the one-buffer form (160 state registers at 20 pieces)
#endif
#ifndef KLT_PBD
#define KLT_PBD 5 // pass B: row elements requested this many chunks ahead
#endif
#ifndef KLT_PIN
#define KLT_PIN 1 // 1: keep every chunk's arithmetic between its own row request and the next one (see pass A)
#endif
// (Ablations of kl_tile_kernel -- no barrier / scalar part, no row requests: scripts/exp/csrc_r5/k_kl.h with scripts/exp/klt_exp.hip.  The
// product kernel carries no switch.)
// workgroup barrier that orders LDS traffic only: __syncthreads() would also drain the vector-memory queue, i.e. wait for the
// row prefetch (LDS-DMA) at every coordinate step.  Each wavefront reads back only LDS slots its own LDS-DMA wrote, after its
// own s_waitcnt vmcnt, so the barrier has nothing to order there.
#define KLT_BARRIER()                                        \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_s_barrier();                        \
        asm volatile("" ::: "memory");                       \
    } while (0)
// dynamic LDS of kl_tile_kernel: two row buffers, coordinates and row sums of the C columns, reduction scratch
// (a staged row is a whole number of 64 x 16-byte wavefront pieces: which pieces exist is then wave-uniform)
__host__ __device__ static inline int kl_tile_p4(int p) { return ((p + 3) / 4 + 63) / 64 * 64; }
__host__ __device__ static inline size_t kl_tile_lds_bytes(int p, int k, int C, int mw_masked = 0, int nbuf = 2, int nw = 8)
{
    return nbuf * (size_t)kl_tile_p4(p) * 16 + (size_t)2 * C * k * 8 + 2 * 2 * C * (nw > 8 ? nw : 8) * 4 + (size_t)C * mw_masked * 8 +
           (nbuf == 1 ? (size_t)64 * nw * 4 : 0); // (one row buffer: + the per-thread sweep counters)
}
// s_waitcnt vmcnt(n), n wave-uniform at run time (0 .. 19: the pieces of a row a wavefront may have in flight)
__device__ static inline void klt_wait_vm(int n)
{
    switch (n) {
#define KLT_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        KLT_W(1) KLT_W(2) KLT_W(3) KLT_W(4) KLT_W(5) KLT_W(6) KLT_W(7) KLT_W(8) KLT_W(9) KLT_W(10) KLT_W(11) KLT_W(12) KLT_W(13) KLT_W(14)
        KLT_W(15) KLT_W(16) KLT_W(17) KLT_W(18) KLT_W(19)
#undef KLT_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// ONEBUF (contractions of 20481 .. ~40400: a row is up to 158 KB, one buffer is all the LDS holds): piece e of the NEXT row is
// requested into its slot right after pass B has read piece e of this row back, and pass A waits for piece e with a counted
// s_waitcnt -- the scheme of kl_reg64_kernel.  (Two buffers otherwise: the next row is requested during pass A.)
// NT = threads of the workgroup.  The library instantiates 512 only (one workgroup per CU, two of its wavefronts per SIMD).  256 (TWO
// independent workgroups per CU, one wavefront of each per SIMD, one row buffer: LDS <= 80 KB) and 1024 (four wavefronts per SIMD at
// 128 registers) are the forms scripts/exp/klt_exp.hip measured in round 5 -- same throughput / 40 % slower.
template <int EPT4, int C, int METHOD, bool ONEBUF = false, int NT = KLT_THREADS>
__global__ __launch_bounds__(NT, (NT == 256 ? 2 : (NT == 1024 ? 4 : 1))) void kl_tile_kernel(const KlTileArgs a)
{
    constexpr int NV = (METHOD == 4) ? 1 : 2; // sums per column and step: {num} or {a, b}
    constexpr int NROWBUF = ONEBUF ? 1 : 2;
    constexpr int NW = NT / 64;               // wavefronts of the workgroup
    constexpr int KV = (ONEBUF && EPT4 >= 19) ? (KLT_V & 2) : KLT_V; // (160 state registers at 20 pieces: nothing more may stay live across a pass)
    constexpr int RS = NW > 8 ? NW : 8;       // wave totals per sum in the reduction scratch
    extern __shared__ __attribute__((aligned(16))) unsigned char kl_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave-uniform: scalar branches)
    const int k = a.k, P4 = kl_tile_p4(a.p); // float4 slots per row, padded to whole wavefront pieces (the arrays are padded further)
    const int rowb = P4 * 16;
    double *xs = (double *)(kl_smem + NROWBUF * (size_t)rowb); // [C][k]
    double *sws = xs + C * k;                            // [C][k]
    float *red = (float *)(sws + C * k);                 // [2][NV * C][8]
    unsigned long long *mks = (unsigned long long *)(red + 2 * 2 * C * RS); // [C][mw] mask words of the block's columns (a.mask only)
    const int col0 = a.colbase + blockIdx.x * C;
    const float tiny = (float)NNLM_TINY;

    // Pieces e = 0 .. EPT4-1 of a row (float4 slots e * 512 + 64 * wave + lane) belong to this wavefront: it loads them, reads
    // them back and owns the matching chunks of the state.  EPT4 = ceil(P4 / 512) exactly (the host picks the instantiation), so
    // every piece but the last exists for every wavefront and only `last` is a run-time (wave-uniform) condition.
    const bool last = (EPT4 - 1) * NT + wave * 64 < P4;
#define KLT_HAS(e_) ((e_) + 1 < EPT4 || last)
    // Slots at or beyond L4 = ld / 4 (only in a row's last piece; the arrays end there) are never loaded: they are zeroed here
    // once in both buffers, their state entries are b = 0, y = 1, so they add nothing and never change.
    const int voff = lane * 16, L4 = (int)(a.lda >> 2);
    for (int i = L4 + tid; i < P4; i += NT) {
        *(f32x4 *)(kl_smem + (size_t)i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!ONEBUF) *(f32x4 *)(kl_smem + (size_t)rowb + (size_t)i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // piece e of row q -> buffer bufsel.  An LDS-DMA instruction costs its wavefront ~100 issue cycles, so the pieces of the NEXT
    // row are requested one per chunk of pass A (their issue hides behind the other wavefront's arithmetic), not in a burst at
    // the top of the step where every wavefront of the block would be issuing loads at the same time.
    // (written as the instruction itself: scalar base address + one per-lane offset register for all pieces, LDS base in M0 --
    //  the builtin form spent ~10 scalar and vector instructions per piece on 64-bit per-lane addresses)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)kl_smem + (unsigned)wave * 1024u;
    auto issue_piece = [&](int q, int bufsel, int e) {
        const unsigned char *src = (const unsigned char *)(a.Yf + (size_t)q * a.ldyf) + (size_t)wave * 1024 + (size_t)e * (NT * 16); // wave-uniform
        const unsigned dst = lds0 + (unsigned)bufsel * (unsigned)rowb + (unsigned)e * (NT * 16);
        const unsigned long long sp = (unsigned long long)src;
        const unsigned long long su = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sp >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)sp);
        const unsigned du = (unsigned)__builtin_amdgcn_readfirstlane((int)dst);
        // (only slots of the LAST piece can lie beyond the end of the arrays: L4 >= 512 (EPT4 - 1) + 64 wave whenever that piece exists)
        if (e + 1 < EPT4 || e * NT + wave * 64 + lane < L4)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(su), "s"(du) : "memory");
    };
    auto issue = [&](int q, int bufsel) {
#pragma unroll
        for (int e = 0; e < EPT4; e++)
            if (KLT_HAS(e)) issue_piece(q, bufsel, e);
    };

    // The scalar part of a coordinate step (sums of the eight wave totals, quotient, new coordinate, rel-change test) is the same
    // ~50 fp64 instructions for every column: lane c of every wavefront does it for column c (all C columns in ONE instruction
    // stream, redundantly per wavefront -- no second barrier), the per-column coefficients come back with v_readlane.
    const int lc = (lane < C) ? lane : 0; // this lane's column of the block
    bool live_l = col0 + lc < a.ncols;
    if (live_l && a.mask) { // all coordinates masked: 0 sweeps, values copied through (src/update_with_missing.cpp:33)
        bool all = true;
        for (int w = 0; w < a.mw; w++) {
            const int bits = (k - 64 * w >= 64) ? 64 : k - 64 * w;
            const unsigned long long km = (bits >= 64) ? ~0ull : ((1ull << bits) - 1ull);
            all = all && ((a.mask[(size_t)(col0 + lc) * a.mw + w] & km) == km);
        }
        live_l = !all;
    }
    if (a.mask)
        for (int e = tid; e < C * a.mw; e += NT) mks[e] = (col0 + e / a.mw < a.ncols) ? a.mask[(size_t)col0 * a.mw + e] : ~0ull;
    for (int e = tid; e < C * k; e += NT) {
        const int c = e / k, q = e - c * k, col = col0 + c;
        xs[e] = (col < a.ncols) ? a.X[(size_t)q * a.ldx + col] : 0.0;
        sws[e] = (col < a.ncols) ? (a.sumw_cols ? a.sumw_cols[(size_t)col * a.ldsw + q] : a.sumw[q]) : 1.0;
    }

    const bool own_init = !ONEBUF && a.Yinit == nullptr; // block-uniform
    f32x4 y[C][EPT4], b[C][EPT4];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int col = (col0 + c < a.ncols) ? col0 + c : col0;
        const f32x4 *Ac = (const f32x4 *)(a.Adata + (size_t)col * a.lda);
        const f32x4 *Yc = own_init ? Ac : (const f32x4 *)(a.Yinit + (size_t)col * a.lda);
#pragma unroll
        for (int e = 0; e < EPT4; e++) {
            const int idx4 = e * NT + tid;
            const bool valid = KLT_HAS(e) && idx4 < L4 && col0 + c < a.ncols;
            b[c][e] = valid ? Ac[idx4] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (ONEBUF || !own_init) y[c][e] = valid ? Yc[idx4] : f32x4{1.f, 1.f, 1.f, 1.f};
            else y[c][e] = valid ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{1.f, 1.f, 1.f, 1.f};
        }
    }
    // All 4 C EPT4 loads are in flight together; a register use of every one of them HERE (once): otherwise the compiler puts
    // the wait for these loads at their first use inside the step loop, where it would also drain the row prefetch every step.
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
        for (int e = 0; e < EPT4; e++) asm volatile("" : "+v"(b[c][e]), "+v"(y[c][e]));
    __syncthreads();
    int bufsel = 0;
    if constexpr (!ONEBUF) {
        // Starting states formed here, y = sum_q x[q] * (row q of the fixed factor) -- the k rows come through the two row buffers
        // exactly as in the step loop (the last step requests row 0 for the first coordinate step): one fused multiply-add per element and
        // coordinate, against a GEMM kernel that writes a matrix-sized buffer this kernel then reads back (wh_store_kernel).
        if (own_init) {
            issue(0, 0);
            for (int q = 0; q < k; q++) {
                const int qn = (q + 1 < k) ? q + 1 : 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned char *rowi = kl_smem + (size_t)bufsel * rowb;
                const int nbuf = bufsel ^ 1;
                bufsel ^= 1;
                float xc[C];
#pragma unroll
                for (int c = 0; c < C; c++) xc[c] = (float)xs[c * k + q];
#pragma unroll
                for (int e = 0; e < EPT4; e++) {
                    if (KLT_HAS(e)) {
                        const f32x4 w = *(const f32x4 *)(rowi + (size_t)(e * NT + tid) * 16);
                        issue_piece(qn, nbuf, e);
#pragma unroll
                        for (int c = 0; c < C; c++) y[c][e] = __builtin_elementwise_fma(f32x4{xc[c], xc[c], xc[c], xc[c]}, w, y[c][e]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
        for (int e = 0; e < EPT4; e++) y[c][e] = y[c][e] + tiny;
#if KLT_TIMING
    // T[i] += cycles since the previous mark: 0 prologue / loop overhead, 1 top of the step (row wait, LDS reads of the coordinate), 2 pass A,
    // 3 wave totals, 4 barrier, 5 scalar part, 6 pass B, 7 epilogue
    unsigned long long tT[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#define KLT_T(i_)                                                                             \
    do {                                                                                      \
        unsigned long long tn_;                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");          \
        tT[i_] += tn_ - tprev;                                                                \
        tprev = tn_;                                                                          \
    } while (0)
#else
#define KLT_T(i_) do { } while (0)
#endif
    const unsigned long long cmask = (C >= 64) ? ~0ull : ((1ull << C) - 1ull); // lanes that own a column
    double S_l = 0.0;
    for (int q = 0; q < k; q++) S_l += xs[lc * k + q];
    unsigned tdone_l = 0;
    // (SCD only: in Lee's instantiation the same change moved a chunk of the state into scratch INSIDE the step loop -- it keeps its
    //  20 bytes of per-sweep spills)
    constexpr bool TDLDS = ONEBUF && METHOD == 3;
    unsigned *tdl = (unsigned *)(mks + (a.mask ? C * a.mw : 0)); // ONEBUF: [NT] sweep counters (kl_tile_lds_bytes reserves them)
    auto tid_fresh = [&]() -> int { // threadIdx.x from the lane count and the (scalar) wavefront number
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        return TDLDS ? 64 * wave + ln : tid;
    };
    if constexpr (TDLDS) tdl[tid] = 0u;
    bool run_l = live_l && a.max_iter > 0 && (1.0 + a.rel_tol) > a.rel_tol, flag_l = false;
    bool any = (__ballot(run_l) & cmask) != 0ull;
    int par = 0;
    if (any && !own_init) issue(0, 0); // (own_init: row 0 is already on its way into buffer `bufsel`)
    while (any) { // block-uniform: every wavefront holds the same per-lane column state
        flag_l = 0.0 > a.rel_tol; // rel_err starts each sweep at 0 (src/base_algorithms.cpp:93,137)
        for (int q = 0; q < k; q++) {
            KLT_T(0);
            const int qn = (q + 1 < k) ? q + 1 : 0; // next row (row 0 again for a sweep that may follow)
            if (!ONEBUF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wavefront's pieces of row q (requested during step q - 1) have landed
            const unsigned char *rowp = kl_smem + (ONEBUF ? 0 : (size_t)bufsel * rowb);
            const int nbuf = ONEBUF ? 0 : bufsel ^ 1; // free since pass B of step q - 1
            bufsel ^= 1;
            const int npm1 = (last ? EPT4 : EPT4 - 1) - 1; // ONEBUF: pieces of row q this wavefront requested (in order) - 1
            bool m_l = false;
            if (a.mask) m_l = (mks[lc * a.mw + (q >> 6)] >> (q & 63)) & 1ull; // (LDS copy: a global load here would drain the row prefetch)
            const bool doq_l = run_l && !m_l;
            // (a coordinate that no column of the block visits -- masks only -- takes the step with coefficients 0: a `continue` here
            //  is a second path into the loop latch, and the compiler then writes every updated state chunk to a NEW register and
            //  copies 80 registers back per step)
            const double xq_l = xs[lc * k + q]; // read BEFORE the barrier: wavefront 0 rewrites it after
            const double sw_l = sws[lc * k + q];
            // Lee: the denominator (src/base_algorithms.cpp:142) does not depend on this step's sums: den, its reciprocal and the Newton step
            // are three short dependent fp64 chains, issued one each behind the first chunks of pass A (where both wavefronts of a SIMD have
            // vector work to cover them) instead of between the pass and the barrier (where every wavefront of the block waited on them)
            double den4 = 1.0, rd4 = 0.0;
            auto rd_stage = [&](int st) {
                if (METHOD != 4) return;
                if (st == 0) den4 = sw_l + a.r0 * xq_l + a.r1 * (S_l - xq_l) + a.r2; // :142
                else if (st == 1) rd4 = __builtin_amdgcn_rcp(den4);
                else rd4 = __builtin_fma(__builtin_fma(-den4, rd4, 1.0), rd4, rd4);
                asm volatile("" : "+v"(den4), "+v"(rd4)); // (stays where it is written)
            };
            KLT_T(1);
            f32x4 acc[C][NV];
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int v = 0; v < NV; v++) acc[c][v] = f32x4{0.f, 0.f, 0.f, 0.f};
            // the row element of chunk e + 1 is fetched from LDS while chunk e is processed; the scheduling barrier keeps the
            // compiler from hoisting all EPT4 fetches (and their registers) to the top of the unrolled loop
            auto wload = [&](int e) -> f32x4 { return *(const f32x4 *)(rowp + (size_t)(e * NT + tid) * 16); };
            f32x4 wq[2]; // (two named registers by the parity of e: no copy per chunk)
            if (ONEBUF) klt_wait_vm(npm1); // piece 0 has landed once at most the npm1 younger ones are outstanding
            wq[0] = wload(0);
#pragma unroll
            for (int e = 0; e < EPT4; e++) {
                if (KLT_HAS(e)) { // wave-uniform (compile-time true for all but the last piece)
                    if (e + 1 < EPT4 && KLT_HAS(e + 1)) {
                        if (ONEBUF) klt_wait_vm(npm1 - (e + 1));
                        wq[(e + 1) & 1] = wload(e + 1);
                    }
                    if (!ONEBUF) issue_piece(qn, nbuf, e);
                    const f32x4 w = wq[e & 1];
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        f32x4 r;
                        r[0] = __builtin_amdgcn_rcpf(__builtin_fabsf(y[c][e][0]));
                        r[1] = __builtin_amdgcn_rcpf(__builtin_fabsf(y[c][e][1]));
                        r[2] = __builtin_amdgcn_rcpf(__builtin_fabsf(y[c][e][2]));
                        r[3] = __builtin_amdgcn_rcpf(__builtin_fabsf(y[c][e][3]));
                        if constexpr (METHOD == 4) {
                            acc[c][0] = __builtin_elementwise_fma(w, b[c][e] * r, acc[c][0]); // Wt.row(k) * (Aj / (wh + eps)), :141
                        } else {
                            const f32x4 u = w * r;                                             // mu, :97
                            const f32x4 bu = b[c][e] * u;
                            acc[c][0] = __builtin_elementwise_fma(bu, u, acc[c][0]);           // a, :98
                            acc[c][1] = acc[c][1] + bu;                                        // b, :99
                        }
                    }
                }
                if ((KV & 1) && e < 3) rd_stage(e);
#if KLT_PIN
                // (the last piece's run-time test splits the unrolled loop into basic blocks, and the optimizer then SINKS the arithmetic of
                //  every chunk -- pure values, used only by the reduction -- into the last block: all row requests and LDS reads came out in
                //  one burst ahead of the whole pass.  An opaque use of the sums keeps a chunk's instructions where they are written.)
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int v = 0; v < NV; v++) asm volatile("" : "+v"(acc[c][v]));
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int st = ((KV & 1) ? (EPT4 < 3 ? EPT4 : 3) : 0); st < 3; st++) rd_stage(st); // (what found no chunk to ride on)
            KLT_T(2);
            if (KV & 2) { // the C * NV wave totals side by side: six dependent DPP steps for all of them, not six per sum one after the other
                float t[C][NV];
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int v = 0; v < NV; v++) t[c][v] = (acc[c][v][0] + acc[c][v][1]) + (acc[c][v][2] + acc[c][v][3]);
#define KLT_RSTEP(CTRL, RM)                                    \
    _Pragma("unroll") for (int c = 0; c < C; c++)              \
        _Pragma("unroll") for (int v = 0; v < NV; v++) t[c][v] += kl_dpp<CTRL, RM>(0.f, t[c][v]);
                KLT_RSTEP(0xB1, 0xF)
                KLT_RSTEP(0x4E, 0xF)
                KLT_RSTEP(0x141, 0xF)
                KLT_RSTEP(0x140, 0xF)
                KLT_RSTEP(0x142, 0xA)
                KLT_RSTEP(0x143, 0xC)
#undef KLT_RSTEP
                if (lane == 63) {
#pragma unroll
                    for (int c = 0; c < C; c++)
#pragma unroll
                        for (int v = 0; v < NV; v++) red[((par * C + c) * NV + v) * RS + wave] = t[c][v];
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        const float t = kl_wave_total((acc[c][v][0] + acc[c][v][1]) + (acc[c][v][2] + acc[c][v][3]));
                        if (lane == 63) red[((par * C + c) * NV + v) * RS + wave] = t;
                    }
            }
            KLT_T(3);
            KLT_BARRIER();
            KLT_T(4);
            // pass B's first row elements are requested here, behind the reads of the wave totals and in front of the scalar part: they
            // arrive while it runs (KV & 4; otherwise in front of the pass itself)
            constexpr int PDW = (NT == 1024) ? 2 : KLT_PBD; // (four wavefronts per SIMD cover an LDS round trip with less of it in flight, and have half the registers)
            constexpr int PD = ONEBUF ? KLT_PBD1 : ((EPT4 < PDW) ? EPT4 : PDW);
            f32x4 wb[PD + 1];
            float coef_l = 0.f;
            {
                f32x4 rrv[NV][NW / 4]; // the wave totals of the lane's column (fp32, like the totals themselves)
#pragma unroll
                for (int v = 0; v < NV; v++)
#pragma unroll
                    for (int u = 0; u < NW / 4; u++) rrv[v][u] = *(const f32x4 *)(red + ((par * C + lc) * NV + v) * RS + 4 * u);
                if (KV & 4) {
#pragma unroll
                    for (int e = 0; e < PD; e++)
                        if (KLT_HAS(e)) wb[e] = wload(e);
                }
                double sv[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    float t4 = (rrv[v][0][0] + rrv[v][0][1]) + (rrv[v][0][2] + rrv[v][0][3]);
                    if (NW >= 8) t4 = t4 + ((rrv[v][1][0] + rrv[v][1][1]) + (rrv[v][1][2] + rrv[v][1][3]));
                    if (NW == 16)
                        t4 = t4 + (((rrv[v][NW / 4 - 2][0] + rrv[v][NW / 4 - 2][1]) + (rrv[v][NW / 4 - 2][2] + rrv[v][NW / 4 - 2][3])) +
                                   ((rrv[v][NW / 4 - 1][0] + rrv[v][NW / 4 - 1][1]) + (rrv[v][NW / 4 - 1][2] + rrv[v][NW / 4 - 1][3])));
                    sv[v] = (double)t4;
                }
                const double sw = sw_l;
                if (METHOD == 4) {
                    const double tmp = sv[0] * rd4;
                    const double d = (tmp - 1) * xq_l; // :143
                    if (doq_l) {
                        coef_l = (float)d;
                        S_l += d;                                                          // :144
                        if (wave == 0 && lane < C) xs[lc * k + q] = xq_l * tmp;            // :145
                        flag_l = flag_l || (2 * fabs(tmp - 1) > a.rel_tol * (tmp + 1));    // :146-147 without the division
                    }
                } else {
                    const double aa = sv[0] + a.r0;                                             // :98,100
                    const double bb = (sv[NV - 1] - sw) + aa * xq_l - a.r2 - a.r1 * (S_l - xq_l); // :99,101
                    const double den = aa + NNLM_TINY;
                    double rd = __builtin_amdgcn_rcp(den);
                    rd = __builtin_fma(__builtin_fma(-den, rd, 1.0), rd, rd);
                    double tmp = bb * rd; // :102
                    if (!(tmp > 0)) tmp = 0;
                    if (doq_l && tmp != xq_l) {
                        const double d = tmp - xq_l;
                        coef_l = (float)d;
                        flag_l = flag_l || (2 * fabs(d) > a.rel_tol * (tmp + xq_l + NNLM_TINY)); // :107-108
                        S_l += d;
                        if (wave == 0 && lane < C) xs[lc * k + q] = tmp;
                    }
                }
            }
            float coef[C];
#pragma unroll
            for (int c = 0; c < C; c++) coef[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, coef_l), c));
            par ^= 1;
            KLT_T(5);
            { // pass B: y += coef * w for every column of the block, unconditionally (coef = 0 leaves a state as it is: a branch around
              // the pass makes the compiler copy all state registers where the two paths meet)
                // (two fused multiply-adds per chunk and column do not cover an LDS round trip: with the row element one chunk ahead, as
                //  in pass A, the pass waited ~100 cycles per chunk -- 0.8-1.0 of a half-step's 2.3 ms at config 3; KLT_PBD chunks ahead)
                if (!(KV & 4)) {
#pragma unroll
                    for (int e = 0; e < PD; e++)
                        if (KLT_HAS(e)) wb[e] = wload(e);
                }
#pragma unroll
                for (int e = 0; e < EPT4; e++) {
                    if (KLT_HAS(e)) { // wave-uniform
                        if (e + PD < EPT4 && KLT_HAS(e + PD)) wb[(e + PD) % (PD + 1)] = wload(e + PD);
                        const f32x4 w = wb[e % (PD + 1)];
                        if (ONEBUF) { // slot e has been read back: the next row's piece goes into it
                            asm volatile("" : : "v"(w) : "memory");
                            issue_piece(qn, 0, e);
                        }
#pragma unroll
                        for (int c = 0; c < C; c++) y[c][e] = __builtin_elementwise_fma(f32x4{coef[c], coef[c], coef[c], coef[c]}, w, y[c][e]); // :106, :143
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            KLT_T(6);
        }
        KLT_BARRIER(); // xs[] written by thread 0 during this sweep is read by everyone in the next
        if constexpr (TDLDS) {
            // (160 state registers: the sweep counter -- touched once per sweep -- lives in LDS, not in a register the allocator would park
            //  in scratch; each lane its own word of the reduction scratch's tail)
            const int tq = tid_fresh(); // (the thread index rebuilt from the lane count: its address in LDS is not kept through the sweep either)
            unsigned td = tdl[tq];
            if (run_l) {
                td++;
                run_l = td < a.max_iter && flag_l;
            }
            tdl[tq] = td;
        } else if (run_l) {
            tdone_l++;
            run_l = tdone_l < a.max_iter && flag_l;
        }
        any = (__ballot(run_l) & cmask) != 0ull;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the row requested for a sweep that did not follow
    __syncthreads();
    const int tid_e = tid_fresh();
    if constexpr (TDLDS) tdone_l = tdl[tid_e];
    for (int e = tid_e; e < C * k; e += NT) {
        const int c = e / k, q = e - c * k, col = col0 + c;
        if (col < a.ncols) {
            const double xv = xs[e];
            a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xv;
            if (a.op_mode == 1) ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xv;
        }
    }
    if (tid_e < 64) { // (wavefront 0; the thread index afresh: its copies in `tid`, `lane` need not survive the sweeps)
        const long long tot = wave_sum_ll((tid_e < C) ? (long long)tdone_l : 0ll);
        if (tid_e == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
#if KLT_TIMING
    KLT_T(7);
    if (a.tim && tid == 0)
        for (int i = 0; i < 8; i++) a.tim[(size_t)blockIdx.x * 8 + i] = tT[i];
#endif
#undef KLT_T
#undef KLT_HAS
}

// sumw[q] = sum_i Y[q][i], i < p (fp64, from the master): sumW of src/update_with_missing.cpp:27
__global__ __launch_bounds__(256) void kl_sumw_kernel(const double *__restrict__ Y, int ldy, int p, double *__restrict__ sumw)
{
    __shared__ double part[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < p; i += 256) s += Y[(size_t)q * ldy + i];
    s = wave_sum(s);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) sumw[q] = (part[0] + part[1]) + (part[2] + part[3]);
}

// Per-column row sums over the non-missing entries (src/update_with_missing.cpp:122,130) from the CSR row lists of
// k_missing.h (the listed rows are the missing ones when meta's top bit is set: sum = full - listed).  One wavefront per
// column, lane = coordinate (chunks of 64), Yrow [p][KP] row-major.
__global__ __launch_bounds__(256) void kl_sumw_cols_kernel(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ meta, const int *__restrict__ idx,
                                                           const double *__restrict__ Yrow, int KP, int k, const double *__restrict__ sumw_full,
                                                           double *__restrict__ out, int ldsw, int ncols)
{
    const int lane = threadIdx.x & 63, col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= ncols) return;
    const uint32_t mt = meta[col];
    const int len = (int)(mt & 0x7FFFFFFFu);
    const bool complement = (mt >> 31) != 0;
    const int *rows = idx + ptr[col];
    for (int q0 = 0; q0 < k; q0 += 64) {
        const int q = q0 + lane;
        double s = 0.0;
        if (q < k)
            for (int r = 0; r < len; r++) s += Yrow[(size_t)rows[r] * KP + q];
        if (q < k) out[(size_t)col * ldsw + q] = complement ? sumw_full[q] - s : s;
    }
}

// ==================================================================================================================
// kl_stream_kernel -- the KL solvers without size limits (any contraction length, any rank, both precisions): what runs
// when the register-resident kernels do not fit (contraction longer than they hold, rank beyond a 64-bit mask word, LDS).
// One 256-thread block per column; the state vector y and a contiguous copy of the data column live in a global scratch
// buffer St [ncols][2][ldst] (y, then b) and are streamed once per coordinate step: the pending rank-1 refresh of step
// q-1 is applied while the sums of step q are formed.  Same arithmetic as kl_update_kernel (T = double: the reference's).
template <typename T, int METHOD>
__global__ __launch_bounds__(256) void kl_stream_kernel(const KlArgs a, int mw, T *__restrict__ St, size_t ldst)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kls_smem[];
    double *xs = (double *)kls_smem; // [k]
    double *red = xs + a.k;          // [2][3][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = a.col0 + blockIdx.x, k = a.k, p = a.p;
    T *ys = St + (size_t)col * 2 * ldst, *bs = ys + ldst;
    const uint32_t *bits = a.bits ? a.bits + (size_t)col * a.words : nullptr;

    bool skipcol = false;
    if (a.mask) {
        skipcol = true;
        for (int w = 0; w < mw; w++) {
            const int nb = (k - 64 * w >= 64) ? 64 : k - 64 * w;
            const unsigned long long km = (nb >= 64) ? ~0ull : ((1ull << nb) - 1ull);
            skipcol = skipcol && ((a.mask[(size_t)col * mw + w] & km) == km);
        }
    }
    for (int q = tid; q < k; q += 256) xs[q] = a.X[(size_t)q * a.ldx + col];
    __syncthreads();
    double S = 0.0;
    for (int q = 0; q < k; q++) S += xs[q];
    auto valid_at = [&](int i) { return !(bits && ((bits[i >> 5] >> (i & 31)) & 1u)); };
    if (!skipcol && a.max_iter > 0) { // y = Yt^T x over the finite entries; contiguous copy of the data column
        const T *Acol = (const T *)a.A + (size_t)col * a.a_col_stride;
        for (int i = tid; i < p; i += 256) {
            const bool v = valid_at(i);
            double yv = 0.0;
            if (v)
                for (int q = 0; q < k; q++) yv = __builtin_fma(a.Y[(size_t)q * a.ldy + i], xs[q], yv);
            ys[i] = (T)yv;
            bs[i] = v ? Acol[(size_t)i * a.a_i_stride] : (T)0;
        }
    }
    double rel = 1.0 + a.rel_tol;
    unsigned t = 0;
    int par = 0, qprev = -1;
    double cprev = 0.0;
    for (; !skipcol && t < a.max_iter && rel > a.rel_tol; t++) {
        rel = 0.0;
        for (int q = 0; q < k; q++) {
            if (a.mask && ((a.mask[(size_t)col * mw + (q >> 6)] >> (q & 63)) & 1ull)) continue;
            const double xq = xs[q];
            double v[3] = {0.0, 0.0, 0.0};
            for (int i = tid; i < p; i += 256) {
                const bool ok = valid_at(i);
                double yv = (double)ys[i];
                if (cprev != 0.0) { // the refresh step qprev owes this entry (:106, :143)
                    if (ok) yv = __builtin_fma(cprev, a.Y[(size_t)qprev * a.ldy + i], yv);
                    if (sizeof(T) == 4) yv = (double)(float)yv;
                    ys[i] = (T)yv;
                }
                const double w = ok ? a.Y[(size_t)q * a.ldy + i] : 0.0;
                const double bv = (double)bs[i];
                if (METHOD == 4) {
                    v[0] += w * (bv / (yv + NNLM_TINY));
                } else {
                    const double u = w / (yv + NNLM_TINY);
                    v[0] += bv * (u * u);
                    v[1] += bv * u;
                }
                v[2] += w;
            }
            cprev = 0.0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                v[c] = wave_sum(v[c]);
                if (lane == 0) red[(par * 3 + c) * 4 + wave] = v[c];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double *r = red + (par * 3 + c) * 4;
                v[c] = (r[0] + r[1]) + (r[2] + r[3]);
            }
            par ^= 1;
            if (METHOD == 4) {
                double tmp = v[0] / (v[2] + a.r0 * xq + a.r1 * (S - xq) + a.r2);
                cprev = (tmp - 1) * xq;
                S += (tmp - 1) * xq;
                if (tid == 0) xs[q] = xq * tmp;
                tmp = 2 * fabs(tmp - 1) / (tmp + 1);
                if (tmp > rel) rel = tmp;
            } else {
                double aa = v[0], bb = v[1] - v[2];
                aa += a.r0;
                bb += aa * xq - a.r2 - a.r1 * (S - xq);
                double tmp = bb / (aa + NNLM_TINY);
                if (tmp < 0) tmp = 0;
                if (tmp != xq) {
                    cprev = tmp - xq;
                    const double er = 2 * fabs(xq - tmp) / (tmp + xq + NNLM_TINY);
                    if (er > rel) rel = er;
                    S += tmp - xq;
                    if (tid == 0) xs[q] = tmp;
                }
            }
            qprev = q;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int q = tid; q < k; q += 256) {
        const double xv = xs[q];
        a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xv;
        if (a.op_mode == 1) {
            if (a.op_f64) ((double *)a.op)[(size_t)q * a.op_ld + col] = xv;
            else ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xv;
        }
    }
    if (tid == 0 && t) atomicAdd(a.sweeps, (unsigned long long)t);
}
