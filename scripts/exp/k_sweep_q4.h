// k_sweep_q4.h -- EXPERIMENT (scripts/exp/sweepq_exp.hip, Q4=split / Q4=all), NOT part of the product: the SCD sweep with FOUR columns
// per wavefront, and a launch that mixes both workgroup shapes.
//
// Idea: k_sweep_q.h gives a wavefront 16 columns (the four blk groups of v_mfma_f64_4x4x4_4b are four column groups; a step is
// NB + 3 MFMAs).  20000 columns are 1250 such wavefronts: 228 SIMDs carry two (0.21 ms where 16384 columns take 0.12), and a
// multi-GPU column shard of 2500 columns still runs 650 dependent steps per wavefront.  Here the four blk groups are four
// coordinate BLOCKS of the same four columns: register r at lane (i, g, j) = gradient of coordinate 4 (4 r + g) + i of column j; one
// bulk MFMA updates four blocks, the chain passes run on the register holding the current block (chain operand zero outside its
// blk group), the block's deltas are copied to the other groups by three row_ror DPP moves per dword.  ceil(NB / 4) + 3 = 7 MFMAs
// per step at k = 50 instead of 16.
//
// Measured (MI355X, k = 50, 50 sweeps; correct to 1e-13 with equal sweep counts in every mode, strict included):
//   * a 4-column wavefront alone on its SIMD takes 119 us, a 16-column one 110: the step is bound by the LATENCY of its four
//     dependent stages (v_max -> 2 wait states -> MFMA -> 6 wait states: 82 cycles each, scripts/exp/chain_overlap.hip), not by the
//     number of MFMAs -- so small column shards gain nothing from the shape;
//   * 256 workgroups of 64 columns + 226 of 16 (every SIMD one large wavefront, 904 of them a small one too; placement traced with
//     HW_ID): the small wavefront, launched later, is starved under oldest-first arbitration (207 us, the large one 124); with
//     s_setprio 3 it takes 185 us and the large one 143: 0.198 ms against 0.2105 for 313 workgroups of 64 columns.  Two waves of the
//     SAME shape overlap well (two 4-column wavefronts per SIMD: 148 us), two different instruction streams do not.
// 6 % of the W half-step (1.6 % of an iteration) for a second operand image, a second kernel body and a launch split: not taken.
#pragma once
#include "csrc_r5/k_sweep_q.h"

#define SWEEPQ4_COLS 16 // columns per workgroup, 4 per wavefront
#ifdef SWEEPQ_TRACE
__device__ unsigned long long sweepq_trace[4 * 4 * 4096];
#endif

__host__ __device__ static inline int sweepq4_nab(int NB) { return (NB + 3) / 4; }                                         // accumulator registers
__host__ __device__ static inline int sweepq4_np(int NB, bool strict) { return (sweepq4_nab(NB) + (strict ? 3 : 1) + 1) / 2; } // operand PAIRS per step
__host__ __device__ static inline size_t sweepq4_img_doubles(int NB, bool strict) { return (size_t)NB * sweepq4_np(NB, strict) * 128 + 4 * NB; }
__host__ __device__ static inline size_t sweepq4_lds_bytes(int KP, int NB, bool strict)
{
    return ((size_t)SWEEPQ4_COLS * (KP + 2) + (size_t)NB * sweepq4_np(NB, strict) * 128) * 8;
}

// Operand image of the 4-column shape: img4[((beta * NP + p) * 64 + lane) * 2 + e], lane = 16 kA + 4 g + iA, entry s = 2 p + e:
//   s < NAB   : G'[4 (4 s + g) + iA][4 beta + kA]                    (operand of register s for the deltas of block beta)
//   s == NAB  : strictly lower part of G'[4 bn + iA][4 bn + kA] in blk group bn % 4, zero elsewhere, bn = (beta + 1) % NB
//   strict only: s == NAB + 1 : 1 / G[q][q], s == NAB + 2 : G[q][q], q = 4 bn + kA   (per-coordinate constants of the next block)
// followed by rinv[q] = 1 / G[q][q], q < 4 NB.  G' as in sweepq_pack_kernel.  Grid: any; blocks [first, gridDim.x) of the launch work here.
__device__ static inline void sweepq4_pack(const double *__restrict__ Graw, int KPg, int k, double r0, double r1, int NB, double *__restrict__ img4,
                                           int strict, int first)
{
    const int NAB = sweepq4_nab(NB), NP = sweepq4_np(NB, strict != 0);
    auto edited = [&](int c, int kc) -> double {
        if (c >= k || kc >= k) return (c == kc) ? 1.0 : 0.0;
        double g = Graw[(size_t)c * KPg + kc];
        if (c == kc && r0 != r1) g += r0 - r1;
        if (r1 != 0) g += r1;
        if (c == kc) g += NNLM_TINY;
        return g;
    };
    auto scaled = [&](int r, int c) -> double { return strict ? edited(r, c) : ((r == c) ? 1.0 : edited(r, c) * (1.0 / edited(r, r))); };
    const int total = NB * NP * 128;
    for (int e = ((int)blockIdx.x - first) * 256 + (int)threadIdx.x; e < total + 4 * NB; e += ((int)gridDim.x - first) * 256) {
        if (e >= total) {
            img4[e] = 1.0 / edited(e - total, e - total);
            continue;
        }
        const int ee = e & 1, lane = (e >> 1) & 63, p = (e >> 7) % NP, beta = (e >> 7) / NP;
        const int kA = lane >> 4, g = (lane >> 2) & 3, iA = lane & 3, s = 2 * p + ee;
        double v = 0.0;
        if (s < NAB) v = scaled(4 * (4 * s + g) + iA, 4 * beta + kA);
        else if (s <= NAB + 2) {
            const int bn = (beta + 1) % NB;
            if (s == NAB) v = (g == (bn & 3) && iA > kA) ? scaled(4 * bn + iA, 4 * bn + kA) : 0.0;
            else if (strict) v = (s == NAB + 1) ? 1.0 / edited(4 * bn + kA, 4 * bn + kA) : edited(4 * bn + kA, 4 * bn + kA);
        }
        img4[e] = v;
    }
}
// both images in one launch: blocks [0, 8) the 16-column image, blocks [8, gridDim.x) the 4-column one
__global__ __launch_bounds__(256) void sweepq_pack2_kernel(const double *__restrict__ Graw, int KPg, int k, double r0, double r1, int NB,
                                                           double *__restrict__ img, double *__restrict__ img4, int strict)
{
    if (blockIdx.x >= 8) {
        sweepq4_pack(Graw, KPg, k, r0, r1, NB, img4, strict, 8);
        return;
    }
    const int NP = sweepq_np(NB, strict != 0);
    auto edited = [&](int c, int kc) -> double {
        if (c >= k || kc >= k) return (c == kc) ? 1.0 : 0.0;
        double g = Graw[(size_t)c * KPg + kc];
        if (c == kc && r0 != r1) g += r0 - r1;
        if (r1 != 0) g += r1;
        if (c == kc) g += NNLM_TINY;
        return g;
    };
    auto scaled = [&](int r, int c) -> double { return strict ? edited(r, c) : ((r == c) ? 1.0 : edited(r, c) * (1.0 / edited(r, r))); };
    const int total = NB * NP * 32;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total + 4 * NB; e += 8 * 256) {
        if (e >= total) {
            img[e] = 1.0 / edited(e - total, e - total);
            continue;
        }
        const int s = 2 * ((e >> 5) % NP) + (e & 1), beta = (e >> 5) / NP, li = (e >> 1) & 15, kA = li >> 2, iA = li & 3;
        double v = 0.0;
        if (s < NB) v = scaled(4 * s + iA, 4 * beta + kA);
        else if (s <= NB + 2) {
            const int bn = (beta + 1) % NB;
            if (s == NB) v = (iA > kA) ? scaled(4 * bn + iA, 4 * bn + kA) : 0.0;
            else if (strict) v = (s == NB + 1) ? 1.0 / edited(4 * bn + kA, 4 * bn + kA) : edited(4 * bn + kA, 4 * bn + kA);
        }
        img[e] = v;
    }
}

// value of blk group G (lanes 4 G .. 4 G + 3 of every row of 16) in all four groups: row_ror by 4 t writes group (G + t) % 4
template <int G> __device__ __forceinline__ double sq4_replicate(double v)
{
    int2 p = __builtin_bit_cast(int2, v);
    p.x = __builtin_amdgcn_update_dpp(p.x, p.x, 0x124, 0xF, 1 << ((G + 1) & 3), false);
    p.y = __builtin_amdgcn_update_dpp(p.y, p.y, 0x124, 0xF, 1 << ((G + 1) & 3), false);
    p.x = __builtin_amdgcn_update_dpp(p.x, p.x, 0x128, 0xF, 1 << ((G + 2) & 3), false);
    p.y = __builtin_amdgcn_update_dpp(p.y, p.y, 0x128, 0xF, 1 << ((G + 2) & 3), false);
    p.x = __builtin_amdgcn_update_dpp(p.x, p.x, 0x12C, 0xF, 1 << ((G + 3) & 3), false);
    p.y = __builtin_amdgcn_update_dpp(p.y, p.y, 0x12C, 0xF, 1 << ((G + 3) & 3), false);
    return __builtin_bit_cast(double, p);
}

// workgroup `wg` of the 16-column workgroups of a launch: columns col0 + 64 n16 + 16 wg ..., Gram slab n16 + wg
template <int NT, int NB, bool STRICT>
__device__ __forceinline__ void sweepq4_body(const SweepArgs &a, const double *__restrict__ img4, unsigned char *smem, int wg, int n16)
{
    constexpr int KP = 16 * NT, XS = KP + 2, NAB = (NB + 3) / 4, NP = (NAB + (STRICT ? 3 : 1) + 1) / 2;
    static_assert(NB <= 4 * NT && NB > 4 * (NT - 1) && NB >= 1, "NB = ceil(k / 4)");
    double *xl = (double *)smem;          // [SWEEPQ4_COLS][XS]: x[column][coordinate], final values
    double *opl = xl + SWEEPQ4_COLS * XS; // [NB * NP * 128]: the operand image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ri = lane >> 4, gB = (lane >> 2) & 3, jB = lane & 3; // row inside the blocks; blk group (= block inside a register); column
    const int k = a.k;
    const int col_base = a.col0 + SWEEPQ_COLS * n16 + wg * SWEEPQ4_COLS;
    const int cl = 4 * wave + jB, col = col_base + cl;
    const bool in_range = col < a.ncols;
    const int cc = in_range ? col : a.col0;

    for (int e = tid; e < NB * NP * 64; e += SWEEPQ_THREADS) ((f64x2 *)opl)[e] = ((const f64x2 *)img4)[e];
    const double *rinv = img4 + (size_t)NB * NP * 128;
    bool act = in_range;
    double acc[NAB], x[NAB];
    // nu = ((L1 - c) + G x) / diag   (src/update_with_missing.cpp:39-41); all loads of a slab in flight together
#pragma unroll
    for (int r = 0; r < NAB; r++) {
        const int q = 4 * (4 * r + gB) + ri;
        acc[r] = 0.0;
        x[r] = (q < k && in_range) ? a.X[(size_t)q * a.ldx + col] : 0.0;
    }
    for (int s = 0; s < a.nslabs; s++) {
        const double *cs = a.Cx + (size_t)s * a.slab_stride + cc;
#pragma unroll
        for (int r = 0; r < NAB; r++) {
            const int q = 4 * (4 * r + gB) + ri;
            acc[r] += (q < k) ? cs[(size_t)q * a.ldc] : 0.0;
        }
    }
#pragma unroll
    for (int r = 0; r < NAB; r++) {
        const int q = 4 * (4 * r + gB) + ri;
        acc[r] = (q < k) ? ((a.r2 != 0) ? a.r2 - acc[r] : -acc[r]) * (STRICT ? 1.0 : rinv[q]) : 0.0;
    }
    __syncthreads(); // operand image complete
    // A 4-column wavefront shares its SIMD with a 16-column one that was launched first and issues 16 MFMAs per step: under the
    // default oldest-first arbitration the small wavefront got the matrix pipe for 126 of its 650 steps while the large one ran
    // (traced: 207 us against 124), and then ran on alone.  With priority its 7 MFMAs per step go first and the large wavefront,
    // whose step is latency bound (266 of ~380 cycles of MFMA issue), takes the slots that are left.
    __builtin_amdgcn_s_setprio(3);
    const f64x2 *opv = (const f64x2 *)opl + lane;
    auto fetch = [&](auto bc, double(&set)[2 * NP]) {
        constexpr int B = decltype(bc)::value;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const f64x2 v = opv[(B * NP + p) * 64];
            set[2 * p] = v[0];
            set[2 * p + 1] = v[1];
        }
    };
    double As[2][2 * NP]; // operand sets by block parity: As[B & 1][r], r < NAB: register r <- deltas of block B; [NAB]: chain operand of block B + 1
    sq_for<0, NB>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        fetch(bc, As[0]);
        const double xr = sq4_replicate<B % 4>(x[B / 4]); // block B sits in register B / 4, blk group B % 4
#pragma unroll
        for (int r = 0; r < NAB; r++) acc[r] = sq_mfma(As[0][r], xr, acc[r]);
    });
    auto write_col = [&]() {
#pragma unroll
        for (int r = 0; r < NAB; r++)
            if (4 * r + gB < NB) xl[cl * XS + 4 * (4 * r + gB) + ri] = in_range ? x[r] : 0.0;
    };
    if constexpr (KP > 4 * NB) { // coordinates beyond the last block
        constexpr int REST = KP - 4 * NB;
        for (int e = tid; e < SWEEPQ4_COLS * REST; e += SWEEPQ_THREADS) xl[(e / REST) * XS + 4 * NB + e % REST] = 0.0;
    }
    if (!act) write_col();

    const double tol = a.rel_tol, tolh = 0.5 * tol, tolhe = 0.5 * tol * NNLM_TINY;
    unsigned t = 0;
    int t_lane = 0;
    bool go = a.max_iter > 0 && __any(act);
    double d_pend = 0.0; // deltas of the previous block in all four blk groups, still owed to every register but the current block's
    bool flag = false;
    // entering step 0: As[1] = operands of block NB - 1 (lazy products of d_pend = 0: any finite values), Lc = chain operand of block 0
    fetch(std::integral_constant<int, NB - 1>{}, As[1]);
    double Lc = As[1][NAB];
    double rinvc = STRICT ? As[1][NAB + 1] : 0.0, gdc = STRICT ? As[1][NAB + 2] : 0.0;
    sq_nop<8>(); // (the initial gradients come out of MFMAs; the first v_max below is inline asm)

    // One block: the four stages of k_sweep_q.h -- [v_max]  pre  [dependent MFMA]  post -- on the register RB that holds block B; in
    // the fourth stage the deltas (valid in blk group GB) are copied to all groups between the v_max and the urgent product, which
    // updates the register of the NEXT block (often RB itself).  Lazy products: the previous block's deltas to every other register.
    auto step = [&](auto bc, auto tc) {
        constexpr int B = decltype(bc)::value, BN = (B + 1) % NB;
        constexpr bool TEST = decltype(tc)::value;
        constexpr int RB = B / 4, GB = B % 4, RN = BN / 4;
        constexpr SqSched S = sq_sched(NAB - 1);
        double(&Ap)[2 * NP] = As[(B + 1) & 1]; // operands of the previous block (lazy products)
        double(&Ac)[2 * NP] = As[B & 1];       // operands of this block: fetched now, first used by the urgent product
        const double m0 = acc[RB], xb = x[RB];
        const bool inG = gB == GB; // this lane holds block B
        auto lazies = [&](auto fromc, auto toc) { // lazy products number from .. to - 1: the registers behind RB, RB itself excluded
            sq_for<decltype(fromc)::value, decltype(toc)::value>([&](auto oc) {
                constexpr int R = (RB + 1 + decltype(oc)::value) % NAB;
                acc[R] = sq_mfma(Ap[R], d_pend, acc[R]);
            });
        };
#define SQ_IC(v) std::integral_constant<int, (v)> {}
#define SQ4_STAGE(s_, val_in, dep_expr)                                                                                 \
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
        double c, m, tmpx = 0.0, dr = 0.0;
        SQ4_STAGE(0, m0, m = sq_mfma(Lc, c, m0))
        fetch(bc, Ac); // (the previous step's lazy products were the last readers of this set)
        SQ4_STAGE(1, m, m = sq_mfma(Lc, c, m0))
        SQ4_STAGE(2, m, m = sq_mfma(Lc, c, m0))
        SQ4_STAGE(3, m, (dr = sq4_replicate<GB>(c), __builtin_amdgcn_sched_barrier(0), acc[RN] = sq_mfma(Ac[RN], dr, acc[RN])))
#undef SQ4_STAGE
#undef SQ_IC
        // rel-change test (src/base_algorithms.cpp:29-32), division-free, in the lanes that hold the block
        if (TEST) {
            if constexpr (STRICT) flag |= inG && (2.0 * fabs(c) > tol * (tmpx + xb + NNLM_TINY));
            else flag |= inG && (fabs(c) > __builtin_fma(tolh, __builtin_fma(2.0, xb, c), tolhe));
        }
        x[RB] = inG ? (STRICT ? tmpx : xb + c) : xb;
        d_pend = dr;
        Lc = Ac[NAB];
        if constexpr (STRICT) rinvc = Ac[NAB + 1], gdc = Ac[NAB + 2];
        __builtin_amdgcn_sched_barrier(0);
    };
    // column flags = OR over the lanes with the same (lane & 3)
    auto colflags = [&](bool f) -> unsigned {
        unsigned long long b = __ballot(f);
        b |= b >> 32;
        b |= b >> 16;
        b |= b >> 8;
        b |= b >> 4;
        return (unsigned)(b & 0xFull);
    };
    while (go) {
        flag = 0.0 > tol; // rel_err starts each sweep at 0: a negative rel_tol never stops
        step(std::integral_constant<int, 0>{}, std::true_type{});
        if (__any(act && !((colflags(flag) >> jB) & 1u))) { // some live column has no coordinate yet that moved by more than rel_tol
            sq_for<1, NB>([&](auto bc) { step(bc, std::true_type{}); });
        } else {
            sq_for<1, NB>([&](auto bc) { step(bc, std::false_type{}); });
        }
        if constexpr (NB & 1) { // the last block's operands sit in set 0; step 0 reads its lazy operands from set 1
#pragma unroll
            for (int s = 0; s < 2 * NP; s++) As[1][s] = As[0][s];
        }
        // end of a sweep (src/base_algorithms.cpp:35: stop when rel_err <= rel_tol)
        const unsigned cf = colflags(flag);
        if (act) {
            t_lane++;
            if (!((cf >> jB) & 1u)) {
                write_col(); // done: these are the column's final values, whatever its lanes go on computing
                act = false;
            }
        }
        t++;
        go = t < a.max_iter && __any(act);
    }
    if (act) write_col();
    __syncthreads(); // x image final
    sweepq_epilogue<NT, SWEEPQ4_COLS>(a, xl, col_base, n16 + wg);
    {
        const long long tot = wave_sum_ll((ri == 0 && gB == 0) ? (long long)t_lane : 0ll); // one lane per column
        if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
}

// The sweep launch: workgroups [0, n16) solve 64 columns each (k_sweep_q.h), workgroups [n16, gridDim.x) 16 columns each, behind them.
// (masked with k > 56: one wavefront per SIMD rather than spills)
template <int NT, int NB, bool HAS_MASK, bool STRICT>
__global__ __launch_bounds__(SWEEPQ_THREADS, ((HAS_MASK && NB >= 15) ? 1 : 2)) void sweep_scd_qmix_kernel(const SweepArgs a, const double *__restrict__ img,
                                                                                                             const double *__restrict__ img4, int n16)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sq_smem[];
#ifdef SWEEPQ_TRACE // harness only: where and when every wavefront ran
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (HAS_MASK) {
        sweepq16_body<NT, NB, true, STRICT>(a, img, sq_smem);
    } else {
        if ((int)blockIdx.x < n16) sweepq16_body<NT, NB, false, STRICT>(a, img, sq_smem);
        else sweepq4_body<NT, NB, STRICT>(a, img4, sq_smem, (int)blockIdx.x - n16, n16);
    }
#ifdef SWEEPQ_TRACE
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *tr = sweepq_trace + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        tr[0] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | (unsigned)__builtin_amdgcn_s_getreg(0xF804);
        tr[1] = tr_t0;
        tr[2] = __builtin_amdgcn_s_memrealtime();
        tr[3] = ((int)blockIdx.x < n16) ? 16 : 4;
    }
#endif
}

// How a launch of `ncols` columns is split: full rounds of 64-column workgroups (one wavefront per SIMD and round: 16384 columns),
// the rest as 16-column workgroups if that is at most one more wavefront per SIMD (4096 columns), else as 64-column ones.
// (per step and SIMD: a 16-column wavefront alone ~380 cycles, two 760, one + a 4-column one ~620, a 4-column one alone ~240)
static inline void sweepq_split(int ncols, bool allow4, int *n16, int *n4)
{
    const int round = 1024 * 16, small = 1024 * 4;
    const int full = allow4 ? (ncols / round) * round : ncols;
    int rest = ncols - full;
    *n16 = full / SWEEPQ_COLS;
    *n4 = 0;
    if (rest > 0) {
        if (allow4 && rest <= small) *n4 = (rest + SWEEPQ4_COLS - 1) / SWEEPQ4_COLS;
        else *n16 += (rest + SWEEPQ_COLS - 1) / SWEEPQ_COLS;
    }
    if (!allow4) *n16 = (ncols + SWEEPQ_COLS - 1) / SWEEPQ_COLS;
}
