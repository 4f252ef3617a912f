// k_sweep_q20.h -- EXPERIMENT (scripts/exp/sweepq_exp.hip, Q20=1), NOT part of the product: sweep_scd_q_kernel with TWENTY columns per
// wavefront.  Correct (1e-13 against the CPU restatement, sweep counts equal) but no faster: 0.2125 ms for 20000 columns in ONE
// round of 1000 wavefronts against 0.2155 ms for the two rounds of the 16-column kernel -- a wavefront's step time is the SUM of its
// instructions' issue times (23 MFMAs x 16.6 + ~35 VALU / DPP / LDS instructions ~ 710 cycles): the second chain does not hide in
// the first one's shadow, it queues behind it.
//
// The 4x4x4 matrix instruction works on 16 columns and a SIMD takes one wavefront of the sweep at full speed -- a second one adds
// its whole time (scripts/exp/mfma44_exp.hip).  The W half-step of the benchmark has 20000 columns = 19.5 per SIMD: 1250
// wavefronts, 226 SIMDs carry two, 0.205 ms where 16384 columns take 0.12.  Here a wavefront owns
//   set A: 16 columns, exactly as in k_sweep_q.h   (acc[beta] at lane (i, col16) = gradient of coordinate 4 beta + i), and
//   set B:  4 columns with FOUR coordinate blocks packed per accumulator register: the instruction's four blk groups, which
//           are four column groups in set A, are four BLOCKS here -- accB[rho] at lane (i, g, j) = gradient of coordinate
//           4 (4 rho + g) + i of column j.  One bulk MFMA updates four blocks: ceil(NB / 4) bulk products per step instead of
//           NB, 3 chain passes on the register that holds the current block (chain operand zero outside its blk group), and the
//           block's four deltas replicated to the other three groups by three row_ror DPP moves per dword (they are the B operand
//           of every group's bulk product).
// 23 MFMAs per step for 20 columns instead of 2 x 16 for 2 x 16, and the two chains are independent: each one's dependent
// instructions sit in the other's shadow.  20000 columns are 1000 wavefronts: one round.
// fp32-operand mode without masks only (the launch picks it when it saves a round: launch_sweep_q); same arithmetic, same
// epilogue, same operand image as k_sweep_q.h plus a second image with the packed operands of set B.
#pragma once
#include "csrc_r5/k_sweep_q.h"

#define SWEEPQ20_COLS 80 // columns per workgroup: 20 per wavefront (16 + 4)

__host__ __device__ static inline int sweepq20_nab(int NB) { return (NB + 3) / 4; }           // accumulator registers of set B
__host__ __device__ static inline int sweepq20_npb(int NB) { return (sweepq20_nab(NB) + 1) / 2; } // operand pairs per step
__host__ __device__ static inline size_t sweepq20_img_doubles(int NB) { return (size_t)NB * sweepq20_npb(NB) * 128; }

// Operand image of set B: img[((beta * NPB + p) * 64 + lane) * 2 + e], lane = 16 kA + 4 g + iA, register rho = 2 p + e:
//   G'[4 (4 rho + g) + iA][4 beta + kA]   (row-scaled edited G, diagonal exactly 1; blocks >= NB / coordinates >= k: identity)
__global__ __launch_bounds__(256) void sweepq20_pack_kernel(const double *__restrict__ Graw, int KPg, int k, double r0, double r1, int NB,
                                                            double *__restrict__ img)
{
    const int NPB = sweepq20_npb(NB);
    auto edited = [&](int c, int kc) -> double {
        if (c >= k || kc >= k) return (c == kc) ? 1.0 : 0.0;
        double g = Graw[(size_t)c * KPg + kc];
        if (c == kc && r0 != r1) g += r0 - r1;
        if (r1 != 0) g += r1;
        if (c == kc) g += NNLM_TINY;
        return g;
    };
    const int total = NB * NPB * 128;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int ee = e & 1, lane = (e >> 1) & 63, p = (e >> 7) % NPB, beta = (e >> 7) / NPB;
        const int kA = lane >> 4, g = (lane >> 2) & 3, iA = lane & 3, rho = 2 * p + ee;
        const int r = 4 * (4 * rho + g) + iA, c = 4 * beta + kA;
        img[e] = (r == c) ? 1.0 : edited(r, c) * (1.0 / edited(r, r));
    }
}

// value of blk group G (lanes 4 G .. 4 G + 3 of every row of 16) in all four groups: row_ror by 4 t writes group (G + t) % 4
template <int G> __device__ __forceinline__ double sq20_replicate(double v)
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

template <int NT, int NB>
__global__ __launch_bounds__(SWEEPQ_THREADS, 1) void sweep_scd_q20_kernel(const SweepArgs a, const double *__restrict__ img, const double *__restrict__ imgB)
{
    constexpr int KP = 16 * NT, NP = (NB + 2) / 2, XS = KP + 2, NAB = (NB + 3) / 4, NPB = (NAB + 1) / 2;
    static_assert(NB <= 4 * NT && NB > 4 * (NT - 1) && NB >= 1, "NB = ceil(k / 4)");
    __shared__ __attribute__((aligned(16))) double xl[SWEEPQ20_COLS * XS]; // x[column][coordinate], final values
    __shared__ __attribute__((aligned(16))) double opl[NB * NP * 32];      // operand image of set A (k_sweep_q.h)
    __shared__ __attribute__((aligned(16))) double oplB[NB * NPB * 128];   // operand image of set B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ri = lane >> 4, c16 = lane & 15;   // set A: row of the lane's coordinates inside their blocks; column inside the wavefront
    const int gB = (lane >> 2) & 3, jB = lane & 3; // set B: the lane's blk group (= block inside a register); column
    const int k = a.k;
    const int col_base = a.col0 + blockIdx.x * SWEEPQ20_COLS;
    const int clA = 20 * wave + c16, colA = col_base + clA;
    const int clB = 20 * wave + 16 + jB, colB = col_base + clB;
    const bool inA = colA < a.ncols, inB = colB < a.ncols;
    const int ccA = inA ? colA : a.col0, ccB = inB ? colB : a.col0;

    for (int e = tid; e < NB * NP * 16; e += SWEEPQ_THREADS) ((f64x2 *)opl)[e] = ((const f64x2 *)img)[e];
    for (int e = tid; e < NB * NPB * 64; e += SWEEPQ_THREADS) ((f64x2 *)oplB)[e] = ((const f64x2 *)imgB)[e];
    const double *rinv = img + (size_t)NB * NP * 32;
    bool actA = inA, actB = inB;
    double acc[NB], x[NB], accB[NAB], xB[NAB];
    // nu = ((L1 - c) + G x) / diag   (src/update_with_missing.cpp:39-41); all loads of a slab in flight together
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int q = 4 * b + ri;
        acc[b] = 0.0;
        x[b] = (q < k && inA) ? a.X[(size_t)q * a.ldx + colA] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < NAB; r++) {
        const int q = 4 * (4 * r + gB) + ri;
        accB[r] = 0.0;
        xB[r] = (q < k && inB) ? a.X[(size_t)q * a.ldx + colB] : 0.0;
    }
    for (int s = 0; s < a.nslabs; s++) {
        const double *cs = a.Cx + (size_t)s * a.slab_stride;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int q = 4 * b + ri;
            acc[b] += (q < k) ? cs[(size_t)q * a.ldc + ccA] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < NAB; r++) {
            const int q = 4 * (4 * r + gB) + ri;
            accB[r] += (q < k) ? cs[(size_t)q * a.ldc + ccB] : 0.0;
        }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int q = 4 * b + ri;
        acc[b] = (q < k) ? ((a.r2 != 0) ? a.r2 - acc[b] : -acc[b]) * rinv[q] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < NAB; r++) {
        const int q = 4 * (4 * r + gB) + ri;
        accB[r] = (q < k) ? ((a.r2 != 0) ? a.r2 - accB[r] : -accB[r]) * rinv[q] : 0.0;
    }
    __syncthreads(); // operand images complete
    const f64x2 *opv = (const f64x2 *)opl + (4 * (lane >> 4) + (lane & 3));
    const f64x2 *opvB = (const f64x2 *)oplB + lane;
    auto fetch = [&](auto bc, double(&set)[2 * NP]) {
        constexpr int B = decltype(bc)::value;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const f64x2 v = opv[(B * NP + p) * 16];
            set[2 * p] = v[0];
            set[2 * p + 1] = v[1];
        }
    };
    auto fetchB = [&](auto bc, double(&set)[2 * NPB]) {
        constexpr int B = decltype(bc)::value;
#pragma unroll
        for (int p = 0; p < NPB; p++) {
            const f64x2 v = opvB[(B * NPB + p) * 64];
            set[2 * p] = v[0];
            set[2 * p + 1] = v[1];
        }
    };
    double As[2][2 * NP], AsB[2][2 * NPB];
    sq_for<0, NB>([&](auto bc) {
        constexpr int B = decltype(bc)::value;
        fetch(bc, As[0]);
        fetchB(bc, AsB[0]);
#pragma unroll
        for (int T = 0; T < NB; T++) acc[T] = sq_mfma(As[0][T], x[B], acc[T]);
        const double xr = sq20_replicate<B % 4>(xB[B / 4]); // block B of set B sits in register B / 4, blk group B % 4
#pragma unroll
        for (int r = 0; r < NAB; r++) accB[r] = sq_mfma(AsB[0][r], xr, accB[r]);
    });
    auto write_colA = [&]() {
#pragma unroll
        for (int b = 0; b < NB; b++) xl[clA * XS + 4 * b + ri] = inA ? x[b] : 0.0;
    };
    auto write_colB = [&]() {
#pragma unroll
        for (int r = 0; r < NAB; r++)
            if (4 * r + gB < NB) xl[clB * XS + 4 * (4 * r + gB) + ri] = inB ? xB[r] : 0.0;
    };
    if constexpr (KP > 4 * NB) { // coordinates beyond the last block
        constexpr int REST = KP - 4 * NB;
        for (int e = tid; e < SWEEPQ20_COLS * REST; e += SWEEPQ_THREADS) xl[(e / REST) * XS + 4 * NB + e % REST] = 0.0;
    }
    if (!actA) write_colA();
    if (!actB) write_colB();

    const double tol = a.rel_tol, tolh = 0.5 * tol, tolhe = 0.5 * tol * NNLM_TINY;
    unsigned t = 0;
    int t_laneA = 0, t_laneB = 0;
    bool go = a.max_iter > 0 && __any(actA || actB);
    double d_pend = 0.0, d_pendB = 0.0; // deltas of the previous block (set B: replicated to all blk groups), still owed to the lazy products
    bool flagA = false, flagB = false;
    fetch(std::integral_constant<int, NB - 1>{}, As[1]);
    fetchB(std::integral_constant<int, NB - 1>{}, AsB[1]);
    double Lc = As[1][NB];
    sq_nop<8>();

    // One block of both sets.  Stages as in k_sweep_q.h -- [v_max A, v_max B]  pre  [dependent MFMA A, dependent MFMA B]  post --
    // with the lazy products of BOTH sets (NB - 1 of set A, NAB - 1 of set B) dealt to the pre / post slots; in the fourth stage
    // the deltas of set B are replicated across the blk groups (6 DPP moves) between the v_max and the urgent products.
    auto step = [&](auto bc, auto tc) {
        constexpr int B = decltype(bc)::value, BN = (B + 1) % NB;
        constexpr bool TEST = decltype(tc)::value;
        constexpr int RB = B / 4, GB = B % 4, RN = BN / 4;         // set B: register and blk group of this block, register of the next
        constexpr int NLA = NB - 1, NLB = NAB - 1, NL = NLA + NLB; // lazy products: set A, set B
        constexpr SqSched S = sq_sched(NL);
        double(&Ap)[2 * NP] = As[(B + 1) & 1];
        double(&Ac)[2 * NP] = As[B & 1];
        double(&ApB)[2 * NPB] = AsB[(B + 1) & 1];
        double(&AcB)[2 * NPB] = AsB[B & 1];
        const double m0 = acc[B], xb = x[B], m0B = accB[RB], xbB = xB[RB];
        const bool inG = gB == GB;                // this lane holds block B of set B
        const double LcB = inG ? Lc : 0.0;        // chain operand of set B: zero outside the block's blk group
        // lazy product n: 0 = set A's next block; 1 .. NLB = set B (the register of the next block first, if it is not this block's:
        // it receives the urgent product at the end of the step); then the rest of set A
        auto lazy = [&](auto nc) {
            constexpr int n = decltype(nc)::value;
            if constexpr (n == 0 || n > NLB) {
                constexpr int o = (n == 0) ? 0 : n - NLB;
                constexpr int T = (B + 1 + o) % NB;
                acc[T] = sq_mfma(Ap[T], d_pend, acc[T]);
            } else {
                constexpr int o = n - 1;                       // registers other than RB, starting behind it
                constexpr int R = (RB + 1 + o) % NAB;
                accB[R] = sq_mfma(ApB[R], d_pendB, accB[R]);
            }
        };
        auto lazies = [&](auto fromc, auto toc) { sq_for<decltype(fromc)::value, decltype(toc)::value>([&](auto oc) { lazy(oc); }); };
#define SQ_IC(v) std::integral_constant<int, (v)> {}
#define SQ20_STAGE(s_, inA_, inB_)                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    c = sq_delta(xb, inA_);                                                                                             \
    cB = sq_delta(xbB, inB_);                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    sq_nop<(S.pre[s_] == 0 ? 2 : 0)>();                                                                                 \
    lazies(SQ_IC(S.off[2 * (s_)]), SQ_IC(S.off[2 * (s_)] + S.pre[s_]));                                                  \
    __builtin_amdgcn_sched_barrier(0);
#define SQ20_POST(s_)                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    lazies(SQ_IC(S.off[2 * (s_) + 1]), SQ_IC(S.off[2 * (s_) + 1] + S.post[s_]));                                          \
    sq_nop<(S.post[s_] == 0 ? 6 : (S.post[s_] == 1 ? 2 : 0))>();                                                        \
    __builtin_amdgcn_sched_barrier(0);
        double c, cB, m, mB;
        SQ20_STAGE(0, m0, m0B)
        m = sq_mfma(Lc, c, m0);
        mB = sq_mfma(LcB, cB, m0B);
        SQ20_POST(0)
        fetch(bc, Ac);
        fetchB(bc, AcB);
        SQ20_STAGE(1, m, mB)
        m = sq_mfma(Lc, c, m0);
        mB = sq_mfma(LcB, cB, m0B);
        SQ20_POST(1)
        SQ20_STAGE(2, m, mB)
        m = sq_mfma(Lc, c, m0);
        mB = sq_mfma(LcB, cB, m0B);
        SQ20_POST(2)
        SQ20_STAGE(3, m, mB)
        // (cB comes out of inline asm: the DPP moves below need two wait states behind it -- the pre slot's product or its s_nop)
        const double dBr = sq20_replicate<GB>(cB);
        __builtin_amdgcn_sched_barrier(0);
        acc[BN] = sq_mfma(Ac[BN], c, acc[BN]);      // urgent, set A: the next block's gradient
        accB[RN] = sq_mfma(AcB[RN], dBr, accB[RN]); // urgent, set B: the register of the next block (often this block's own)
        SQ20_POST(3)
#undef SQ20_STAGE
#undef SQ20_POST
#undef SQ_IC
        const double d = c;
        if (TEST) {
            flagA |= fabs(d) > __builtin_fma(tolh, __builtin_fma(2.0, xb, d), tolhe);
            flagB |= inG && (fabs(cB) > __builtin_fma(tolh, __builtin_fma(2.0, xbB, cB), tolhe));
        }
        x[B] = xb + d;
        xB[RB] = inG ? xbB + cB : xbB;
        d_pend = d;
        d_pendB = dBr;
        Lc = Ac[NB];
        __builtin_amdgcn_sched_barrier(0);
    };
    // set B: column flags = OR over the lanes with the same (lane & 3)
    auto colflagsB = [&](bool f) -> unsigned {
        unsigned long long b = __ballot(f);
        b |= b >> 32;
        b |= b >> 16;
        b |= b >> 8;
        b |= b >> 4;
        return (unsigned)(b & 0xFull);
    };
    auto colflagsA = [&](bool f) -> unsigned {
        const unsigned long long b = __ballot(f);
        return (unsigned)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xFFFFull);
    };
    auto tests_needed = [&]() -> bool {
        const unsigned cfA = colflagsA(flagA), cfB = colflagsB(flagB);
        return __any((actA && !((cfA >> c16) & 1u)) || (actB && !((cfB >> jB) & 1u)));
    };
    while (go) {
        flagA = flagB = 0.0 > tol; // rel_err starts each sweep at 0: a negative rel_tol never stops
        step(std::integral_constant<int, 0>{}, std::true_type{});
        if (tests_needed()) {
            sq_for<1, NB>([&](auto bc) { step(bc, std::true_type{}); });
        } else {
            sq_for<1, NB>([&](auto bc) { step(bc, std::false_type{}); });
        }
        if constexpr (NB & 1) { // the last block's operands sit in set 0; step 0 reads its lazy operands from set 1
#pragma unroll
            for (int s = 0; s < 2 * NP; s++) As[1][s] = As[0][s];
#pragma unroll
            for (int s = 0; s < 2 * NPB; s++) AsB[1][s] = AsB[0][s];
        }
        // end of a sweep (src/base_algorithms.cpp:35: stop when rel_err <= rel_tol)
        const unsigned cfA = colflagsA(flagA), cfB = colflagsB(flagB);
        if (actA) {
            t_laneA++;
            if (!((cfA >> c16) & 1u)) {
                write_colA();
                actA = false;
            }
        }
        if (actB) {
            t_laneB++;
            if (!((cfB >> jB) & 1u)) {
                write_colB();
                actB = false;
            }
        }
        t++;
        go = t < a.max_iter && __any(actA || actB);
    }
    if (actA) write_colA();
    if (actB) write_colB();
    __syncthreads(); // x image final

    float xmax = 0.0f;
    for (int e = tid; e < SWEEPQ20_COLS * KP; e += SWEEPQ_THREADS) {
        const int q = e / SWEEPQ20_COLS, c = e % SWEEPQ20_COLS, ecol = col_base + c;
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
    if (a.gram_slabs) { // Gram partial sums of this workgroup's 80 columns (k_sweep_q.h)
        const int l15 = lane & 15, lg = lane >> 4;
        double *slab = a.gram_slabs + (size_t)blockIdx.x * KP * KP;
        int tix = 0;
#pragma unroll
        for (int ta = 0; ta < NT; ta++)
#pragma unroll
            for (int tb = ta; tb < NT; tb++) {
                if ((tix++ & 3) != wave) continue;
                f64x4 g = f64x4{0, 0, 0, 0};
#pragma unroll
                for (int s4 = 0; s4 < SWEEPQ20_COLS / 4; s4++) {
                    const double *xr = xl + (4 * s4 + lg) * XS + l15;
                    g = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[16 * ta], xr[16 * tb], g, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) slab[(16 * ta + lg + 4 * r) * KP + 16 * tb + l15] = g[r];
            }
    }
    {
        // sweeps per column: set A counted by the lanes of row 0, set B by the lanes of row 0 in blk group 0
        const long long tot = wave_sum_ll(((ri == 0) ? (long long)t_laneA : 0ll) + ((ri == 0 && gB == 0) ? (long long)t_laneB : 0ll));
        if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
}
