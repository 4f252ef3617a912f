// k_sweep_f.h -- SCD least-squares sweep of the fp32-operand mode (round 6): the recurrence of scd_ls_update (reference
// src/base_algorithms.cpp:3-37) with its starting gradients in fp64 and the 2500-step chain on fp32 state, ONE wavefront per 16 columns.
//
// The fp64 form (k_sweep_q.h, the strict mode's kernel) spends 16 v_mfma_f64_4x4x4 (16.6 cycles each) and four dependent
// v_max_f64 -> MFMA stages per block of four coordinates: 330-360 cycles.  Nothing in the fp32-operand mode's contract (W, H within
// 1e-4 of the reference; average.epochs tolerance-only) asks for an fp64 chain -- only  mu0 = G x - c  cancels, and that is computed
// in fp64 here as before.  What bounds a sweep wavefront on this chip is measured in scripts/exp/issue_exp.hip: a LONE wavefront on a
// SIMD issues one instruction per 6.5 cycles (9.5 when it depends on the previous one) whatever the instruction, v_mfma_f32_16x16x4_f32
// (exact fp32, 32 cycles) blocks the VALU while it runs -- it IS the VALU rate --, and v_mfma_f32_16x16x32_bf16 (16-23 cycles) runs
// beside the VALU.  So the step is built to be SHORT in instructions, with the rank-4 update on the bf16 matrix pipe:
//   * accumulator T (f32x4) holds the scaled gradients nu = mu / G[q][q] of coordinates 16 T .. 16 T + 15 of the wavefront's 16
//     columns in the matrix instruction's own result layout: lane (g, j) = 16 g + j, register r  <->  coordinate 16 T + 4 g + r of
//     column j.  A block of four consecutive coordinates therefore sits in the four registers of ONE lane: its four coordinate steps
//         e_r = min(x_r, nu_r - sum_{s<r} G'[r][s] e_s)      (e = -delta:  max(x - nu, 0) - x = -min(x, nu))
//     are plain in-lane VALU work -- v_med3, v_fma, v_pk_fma, v_min, v_pk_fma, v_min, v_fma, v_min -- with the block's six
//     strictly-lower entries of G' in registers (every lane group carries the constants of ITS block of the accumulator; all four
//     groups execute the chain, the group whose turn it is holds the valid one).
//   * the rank-4 update of all gradients,  nu[16 T ..][cols] += G'[16 T .., 4 b .. 4 b + 3] d,  is ONE matrix instruction per
//     accumulator.  Its B operand wants lane (kk, j) = delta of the block's coordinate kk for column j; the chain leaves them in lane
//     group g, registers 0..3: a 4 x 4 transposition between register index and lane group, done with gfx950's
//     v_permlane16_swap / v_permlane32_swap -- three instructions, depth two.
//   * the product itself is fp32-exact on the bf16 pipe: every fp32 number is EXACTLY the sum of three bf16 numbers (truncations:
//     24 significant bits = 8 + 8 + 8), v_mfma_f32_16x16x32_bf16 has eight K slots per lane group, and the six products
//     d1 g1 + d1 g2 + d1 g3 + d2 g1 + d2 g2 + d3 g1 (everything above 2^-24 of d g) fill six of them: B = [d1 d1 | d1 d2 | d2 d3 | 0 0]
//     from the transposed delta (seven VALU instructions), A = [g1 g2 | g3 g1 | g2 g1 | 0 0] split once per half-step (A = -G', rows
//     of G divided by their diagonal).  The A operands of the first two accumulators stay in registers, the others come from an LDS
//     image one step ahead (ds_read_b128).  Measured against the same step with v_mfma_f32_16x16x4_f32 (scripts/exp/k_sweep_f_v2.h,
//     profiles/r06_sweepf_*.log): 240 instead of 283 cycles per block, same results to the last digits printed.
//   * x (fp32, same layout as the gradients) is updated off the chain with a per-group 0/-1 multiplier; the relative-change tests go
//     through wave ballots masked to the block's lane group (SALU), on the first block of a sweep and then only while some live
//     column has not moved yet, as in the fp64 form.
// Prologue: G' = edited Gram (src/update_with_missing.cpp:20-24) with rows divided by their diagonal, built in LDS as fp64 from
// a.Graw; nu0 = ((L1 - c) + G x) / diag (src/update_with_missing.cpp:39-41) on v_mfma_f64_16x16x4_f64 with the operand rows permuted
// so that its result layout (row = lane group + 4 register) IS the fp32 layout above; then rounded to fp32 once.  Epilogue: the
// fp64 form's (sweepq_epilogue: factor outputs, max|x|, Gram partial sums of the workgroup's columns).
// Workgroups are NW = 4 wavefronts (64 columns; one wavefront per SIMD) or NW = 8 (128 columns, two per SIMD: their instruction
// streams interleave at 3.3 cycles per instruction, the bf16 matrix pipe is far from full -- twice the columns in the same time, which
// is what the persistent form of the fp64 kernel buys with its hand-over machinery).
#pragma once
#include "common.h"
#include "k_sweep.h"
#include "k_sweep_q.h"

// accumulators whose A operands stay in registers: what 512 registers per lane (four wavefronts per workgroup, one per SIMD) or 256
// (eight, two per SIMD) hold without a spill; masked launches carry the mask words and one more code path -- one fewer
#define SWEEPF_RES(has_mask, nw) (((nw) == 8 ? 1 : 2) - ((has_mask) ? 1 : 0))

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS of one workgroup (KP = 16 NA, cols = 16 NW): region 0 = G' [KP][KP + 1] + 1 / diag [KP] (doubles) during the prologue, then the
// x image [cols][KP + 2] (doubles); region 1 = the operand image, NB x NA operands of 64 lanes x 16 bytes.
__host__ __device__ static inline size_t sweepf_r0_bytes(int KP, int cols)
{
    const size_t g = ((size_t)KP * (KP + 1) + KP) * 8, xi = (size_t)cols * (KP + 2) * 8;
    return ((g > xi ? g : xi) + 15) / 16 * 16;
}
__host__ __device__ static inline size_t sweepf_lds_bytes(int KP, int NB, int cols) { return sweepf_r0_bytes(KP, cols) + (size_t)NB * (KP / 16) * 1024; }

// edited Gram entry E[r][c] (r, c < k), the additions in the order of sweepq_img_put()
__device__ __forceinline__ double sweepf_edit(double g, bool diag, double r0, double r1)
{
    if (diag && r0 != r1) g += r0 - r1;
    if (r1 != 0) g += r1;
    if (diag) g += NNLM_TINY;
    return g;
}

// G' -> gl[KP][KP + 1] (fp64: E[r][c] / E[r][r], diagonal exactly 1; coordinates >= k inert), rinv[KP] = 1 / E[q][q]; all threads, ends with a barrier
template <int KP, int THREADS> __device__ __forceinline__ void sweepf_build_gprime(const SweepArgs &a, double *gl, double *rinv)
{
    constexpr int GP = KP + 1, PER = (KP * KP + THREADS - 1) / THREADS;
    const int tid = threadIdx.x, k = a.k;
    double v[PER]; // (all loads of the thread in flight at once: a loop over them is PER round trips to L2)
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int e = tid + i * THREADS, r = e / KP, c = e % KP;
        v[i] = (e < KP * KP && r < k && c < k) ? a.Graw[(size_t)r * a.KPg + c] : 0.0;
    }
    if (tid < KP) rinv[tid] = (tid < k) ? 1.0 / sweepf_edit(a.Graw[(size_t)tid * a.KPg + tid], true, a.r0, a.r1) : 1.0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int e = tid + i * THREADS, r = e / KP, c = e % KP;
        if (e < KP * KP) gl[r * GP + c] = (r == c) ? 1.0 : ((r < k && c < k) ? sweepf_edit(v[i], false, a.r0, a.r1) * rinv[r] : 0.0);
    }
    __syncthreads();
}

__device__ __forceinline__ f32x4 sf_mfma3(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// lane group g's registers (v0, v1, v2, v3)  ->  one register whose lane group kk holds v_kk of group g (the B operand of the update)
template <int g> __device__ __forceinline__ float sf_transpose(float v0, float v1, float v2, float v3)
{
    // v_permlane16_swap a, b: rows (of 16 lanes) 1, 3 of a <-> rows 0, 2 of b;   v_permlane32_swap a, b: lanes 32..63 of a <-> lanes 0..31 of b
    // (probed on the box: scripts/exp/sweepf_exp.hip PROBE=1)
    const auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
    const auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
    const auto q = __builtin_amdgcn_permlane32_swap(p01[g & 1], p23[g & 1], false, false);
    return __uint_as_float(q[(g >> 1) & 1]);
}

// fp32 = three bf16 pieces, exactly (truncation: 24 significant bits = 8 + 8 + 8): v = hi16(v) + hi16(r1) + hi16(r2)
#define SF_HI 0xFFFF0000u
#define SF_PERM_HH 0x07060302u // v_perm_b32(s0, s1): low half <- high half of s1, high half <- high half of s0
// A-side operand of the bf16 x 3 product (slots [g1 g2 | g3 g1 | g2 g1 | 0 0]) and B-side ([d1 d1 | d1 d2 | d2 d3 | 0 0]):
// sum over the slots = d1 g1 + d1 g2 + d1 g3 + d2 g1 + d2 g2 + d3 g1 -- every product above 2^-24 of d g
__device__ __forceinline__ u32x4 sf_split_a(float v)
{
    const unsigned u = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u & SF_HI);
    const unsigned u1 = __float_as_uint(r1);
    const unsigned u2 = __float_as_uint(r1 - __uint_as_float(u1 & SF_HI));
    return u32x4{__builtin_amdgcn_perm(u1, u, SF_PERM_HH), __builtin_amdgcn_perm(u, u2, SF_PERM_HH), __builtin_amdgcn_perm(u, u1, SF_PERM_HH), 0u};
}
__device__ __forceinline__ u32x4 sf_split_b(float v)
{
    const unsigned u = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u & SF_HI);
    const unsigned u1 = __float_as_uint(r1);
    const unsigned u2 = __float_as_uint(r1 - __uint_as_float(u1 & SF_HI));
    return u32x4{__builtin_amdgcn_perm(u, u, 0x07060706u), __builtin_amdgcn_perm(u1, u, SF_PERM_HH), __builtin_amdgcn_perm(u2, u1, SF_PERM_HH), 0u};
}

// ---- one block of the sweep.  Expands to a generic lambda (block B, rel-change tests on / off) inside a scope that holds:
// nu[NA], x[NA] (f32x4), Ares[RES][NB], Aset[2][NA] and fetch(), Ls[NA][2], Lp[NA][2] (= -L10, -L32; (-L20, -L30), (-L21, -L31)),
// nsel[4] (-1 in the lane group's own lanes, else 0), neg_huge, flagmask, tolh, tolhe and the template parameters NB, NA, RES.
// The chain: e_r = min(x_r, nu_r - sum_{s<r} L_rs e_s), the constants negated; rows 2, 3 take the deltas of rows 0, 1 as packed
// FMAs.  The first minimum reads a matrix-instruction result: v_med3_f32 with an opaque -huge third operand is min() without the
// canonicalising v_max the compiler puts in front of a v_min of values it did not produce with arithmetic (the instruction must stay
// the compiler's: as inline asm it loses the wait states of the matrix result and reads the OLD gradient -- measured, 6e-3).
// x + d comes before the swaps (they consume the e's in place: behind them it costs four register copies), its first half in the
// slot the second packed FMA's result needs anyway.
#define SWEEPF_STEP_LAMBDA()                                                                                                           \
    [&](auto bc, auto tc) {                                                                                                            \
        constexpr int B = decltype(bc)::value, T = B / 4, g = B % 4, TN = ((B + 1) % NB) / 4;                                          \
        constexpr bool TEST = decltype(tc)::value;                                                                                     \
        const f32x4 n = nu[T], xo = x[T];                                                                                              \
        const float e0 = __builtin_amdgcn_fmed3f(xo[0], n[0], neg_huge);                                                               \
        const float n1 = __builtin_fmaf(Ls[T][0], e0, n[1]);                                                                           \
        f32x2 n23 = __builtin_elementwise_fma(Lp[T][0], f32x2{e0, e0}, f32x2{n[2], n[3]});                                             \
        const float e1 = __builtin_fminf(xo[1], n1);                                                                                   \
        n23 = __builtin_elementwise_fma(Lp[T][1], f32x2{e1, e1}, n23);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        const f32x2 x01 = __builtin_elementwise_fma(f32x2{e0, e1}, f32x2{nsel[g], nsel[g]}, f32x2{xo[0], xo[1]});                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        const float e2 = __builtin_fminf(xo[2], n23[0]);                                                                               \
        const float n3 = __builtin_fmaf(Ls[T][1], e2, n23[1]);                                                                         \
        const float e3 = __builtin_fminf(xo[3], n3);                                                                                   \
        const f32x2 x23 = __builtin_elementwise_fma(f32x2{e2, e3}, f32x2{nsel[g], nsel[g]}, f32x2{xo[2], xo[3]});                      \
        if (TEST) { /* 2 |d| > tol (x + d + x + eps)  (src/base_algorithms.cpp:29-32), division-free */                                \
            const float ev[4] = {e0, e1, e2, e3};                                                                                      \
            unsigned long long mv = 0ull;                                                                                              \
            _Pragma("unroll") for (int r = 0; r < 4; r++)                                                                              \
                mv |= __ballot(__builtin_fabsf(ev[r]) > __builtin_fmaf(tolh, __builtin_fmaf(2.0f, xo[r], -ev[r]), tolhe));             \
            flagmask |= mv & (0xFFFFull << (16 * g));                                                                                  \
        }                                                                                                                              \
        x[T] = f32x4{x01[0], x01[1], x23[0], x23[1]};                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        const float bv = sf_transpose<g>(e0, e1, e2, e3);                                                                              \
        /* operands of the NEXT block: requested now, used one step later (two register sets by block parity) */                        \
        if constexpr (B + 1 < NB) fetch(std::integral_constant<int, B + 1>{}, Aset[(B + 1) & 1]);                                      \
        else fetch(std::integral_constant<int, 0>{}, Aset[NB & 1]);                                                                    \
        const u32x4 b3 = sf_split_b(bv);                                                                                               \
        nu[TN] = sf_mfma3(TN < RES ? Ares[TN < RES ? TN : 0][B] : Aset[B & 1][TN], b3, nu[TN]); /* the next block's accumulator first */ \
        _Pragma("unroll") for (int o = 1; o < NA; o++) {                                                                               \
            const int To = (TN + o) % NA;                                                                                              \
            nu[To] = sf_mfma3(To < RES ? Ares[To < RES ? To : 0][B] : Aset[B & 1][To], b3, nu[To]);                                    \
        }                                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
    }

// NT: the caller's rank padding KP = 16 NT; NB = ceil(k / 4) blocks; NA = NT accumulators; NW wavefronts (16 columns each) per
// workgroup.  Workgroup blockIdx.x of the launch.
template <int NT, int NB, bool HAS_MASK, int NW>
__device__ __forceinline__ void sweepf16_body(const SweepArgs &a, unsigned char *smem)
{
    constexpr int KP = 16 * NT, NA = NT, GP = KP + 1, XS = KP + 2, COLS = 16 * NW, THREADS = 64 * NW;
    constexpr int RES = SWEEPF_RES(HAS_MASK, NW) < NA ? SWEEPF_RES(HAS_MASK, NW) : NA;
    static_assert(NB <= 4 * NT && NB > 4 * (NT - 1) && NB >= 1, "NB = ceil(k / 4)");
    double *xl = (double *)smem;          // [COLS][XS]: x[column][coordinate], final values -- AFTER the prologue, in the place of
    double *gl = (double *)smem;          // [KP][GP]: G'
    double *rinv = gl + KP * GP;          // [KP]: 1 / E[q][q]
    u32x4 *opl = (u32x4 *)(smem + sweepf_r0_bytes(KP, COLS)); // [NB][NA][64]: operand image of the bf16 x 3 update
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, c16 = lane & 15; // lane group (block inside the accumulator); column inside the wavefront
    const int k = a.k;
    const int col_base = a.col0 + blockIdx.x * COLS;
    const int cl = 16 * wave + c16, col = col_base + cl;
    const bool in_range = col < a.ncols;
    const int cc = in_range ? col : a.col0;

#ifdef SWEEPF_TIMING
    const long long tm0 = __builtin_readcyclecounter();
#endif
    unsigned long long mword = 0ull;
    if (HAS_MASK) mword = a.mask[cc];
    // the column's own inputs, requested ahead of everything else: cross-product slabs (summed in slab order) and the factor entries
    // in the two layouts that read them (fp32 state: coordinate 16 T + 4 g + r; B operand of G' x: coordinate 4 c + g)
    f64x4 a64[NA];
    f32x4 nu[NA], x[NA];
    double xb[NB];
#pragma unroll
    for (int T = 0; T < NA; T++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = 16 * T + 4 * g4 + r;
            a64[T][r] = 0.0;
            x[T][r] = (q < k && in_range) ? (float)a.X[(size_t)q * a.ldx + col] : 0.0f;
        }
    for (int s = 0; s < a.nslabs; s++) {
        const double *cs = a.Cx + (size_t)s * a.slab_stride + cc;
#pragma unroll
        for (int T = 0; T < NA; T++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = 16 * T + 4 * g4 + r;
                a64[T][r] += (q < k) ? cs[(size_t)q * a.ldc] : 0.0;
            }
    }
    sweepf_build_gprime<KP, THREADS>(a, gl, rinv);
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    bool act = in_range && !(HAS_MASK && ((mword & kmask) == kmask)); // arma::all(mask.col(j)) -> column skipped

#pragma unroll
    for (int c = 0; c < NB; c++) { // (requested here: in flight while the operand image is built)
        const int q = 4 * c + g4;
        xb[c] = (q < k && in_range) ? a.X[(size_t)q * a.ldx + col] : 0.0;
    }
    // ---- operand image: A = -G', lane (kk, i) of operand (T, b) = -G'[16 T + i][4 b + kk], split into three bf16 pieces
    for (int e = tid; e < NB * NA * 64; e += THREADS) {
        const int l = e & 63, T = (e >> 6) % NA, b = (e >> 6) / NA;
        opl[e] = sf_split_a(-(float)gl[(16 * T + (l & 15)) * GP + 4 * b + (l >> 4)]);
    }
    // ---- starting gradients in fp64: nu0 = ((L1 - c) + G x) / diag, result layout lane (g, j), register r <-> coordinate 16 T + 4 g + r
    {
#pragma unroll
        for (int T = 0; T < NA; T++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = 16 * T + 4 * g4 + r;
                a64[T][r] = (q < k) ? ((a.r2 != 0) ? a.r2 - a64[T][r] : -a64[T][r]) * rinv[q] : 0.0;
            }
        // G' x: B operand lane (kk, j) = x[4 c + kk][column j]; A operand lane (kk, i) = G'[16 T + 4 (i & 3) + (i >> 2)][4 c + kk]
        // (the f64 instruction's result row is lane group + 4 register: operand row i = g + 4 r carries coordinate 4 g + r)
        const int arow = 4 * (c16 & 3) + (c16 >> 2);
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int q = 4 * c + g4;
#pragma unroll
            for (int T = 0; T < NA; T++) a64[T] = __builtin_amdgcn_mfma_f64_16x16x4f64(gl[(16 * T + arow) * GP + q], xb[c], a64[T], 0, 0, 0);
        }
#pragma unroll
        for (int T = 0; T < NA; T++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = 16 * T + 4 * g4 + r;
                nu[T][r] = (float)a64[T][r];
                if (HAS_MASK && ((mword >> q) & 1ull)) x[T][r] = 0.0f, nu[T][r] = 1e30f; // e = min(0, 1e30) = 0 for good
            }
    }
    // ---- the chain constants of the lane group's blocks
    float Ls[NA][2], nsel[4];
    f32x2 Lp[NA][2];
    float neg_huge = -3.0e38f;
    asm volatile("" : "+v"(neg_huge)); // (opaque: keeps v_med3_f32 a v_med3_f32)
#pragma unroll
    for (int T = 0; T < NA; T++) {
        const double *gb = gl + (16 * T + 4 * g4) * GP + 16 * T + 4 * g4; // the block's own 4 x 4 piece
        Ls[T][0] = -(float)gb[1 * GP + 0], Ls[T][1] = -(float)gb[3 * GP + 2];
        Lp[T][0] = f32x2{-(float)gb[2 * GP + 0], -(float)gb[3 * GP + 0]};
        Lp[T][1] = f32x2{-(float)gb[2 * GP + 1], -(float)gb[3 * GP + 1]};
    }
#pragma unroll
    for (int g = 0; g < 4; g++) nsel[g] = (g4 == g) ? -1.0f : 0.0f;
    __syncthreads(); // operand image complete; G' is not read beyond this point: its place becomes the x image
    const u32x4 *opv = opl + lane;
    u32x4 Ares[RES ? RES : 1][NB]; // the operands of the first RES accumulators stay in registers, the others are fetched one step ahead
#pragma unroll
    for (int T = 0; T < RES; T++)
#pragma unroll
        for (int b = 0; b < NB; b++) Ares[T][b] = opv[(b * NA + T) * 64];
    auto fetch = [&](auto bc, u32x4(&set)[NA]) {
        constexpr int Bf = decltype(bc)::value;
#pragma unroll
        for (int T = RES; T < NA; T++) set[T] = opv[(Bf * NA + T) * 64];
    };
    u32x4 Aset[2][NA];
    fetch(std::integral_constant<int, 0>{}, Aset[0]);

    // the column's final values -> x image (masked entries from the input; rows of out-of-range columns zero)
    auto write_col = [&]() {
#pragma unroll
        for (int T = 0; T < NA; T++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = 16 * T + 4 * g4 + r;
                double v = (double)x[T][r];
                if (HAS_MASK && in_range && q < k && ((mword >> q) & 1ull)) v = a.X[(size_t)q * a.ldx + col];
                xl[cl * XS + q] = in_range ? v : 0.0;
            }
    };
    if (!act) write_col();

    const float tol = (float)a.rel_tol, tolh = 0.5f * tol, tolhe = 0.5f * tol * (float)NNLM_TINY;
    unsigned t = 0;
    int t_lane = 0;
    bool go = a.max_iter > 0 && __any(act);
    unsigned long long flagmask = 0ull;
    auto step = SWEEPF_STEP_LAMBDA();
    // columns (bit c16) some coordinate of which moved by more than rel_tol in this sweep
    auto moved = [&]() -> unsigned { return (unsigned)((flagmask | (flagmask >> 16) | (flagmask >> 32) | (flagmask >> 48)) & 0xFFFFull); };
#ifdef SWEEPF_TIMING
    const long long tm1 = __builtin_readcyclecounter();
#endif
    while (go) {
        flagmask = (0.0f > tol) ? ~0ull : 0ull; // rel_err starts each sweep at 0: a negative rel_tol never stops
        step(std::integral_constant<int, 0>{}, std::true_type{});
        if (__any(act && !((moved() >> c16) & 1u))) {
            sq_for<1, NB>([&](auto bc) { step(bc, std::true_type{}); });
        } else {
            sq_for<1, NB>([&](auto bc) { step(bc, std::false_type{}); });
        }
        if constexpr (NB & 1) { // block 0's operands were requested into set 1 by the last block; block 0 reads set 0
#pragma unroll
            for (int T = RES; T < NA; T++) Aset[0][T] = Aset[1][T];
        }
        // end of a sweep (src/base_algorithms.cpp:35: stop when rel_err <= rel_tol)
        if (act) {
            t_lane++;
            if (!((moved() >> c16) & 1u)) {
                write_col(); // done: these are the column's final values, whatever its lanes go on computing
                act = false;
            }
        }
        t++;
        go = t < a.max_iter && __any(act);
    }
#ifdef SWEEPF_TIMING
    const long long tm2 = __builtin_readcyclecounter();
#endif
    if (act) write_col();
    __syncthreads(); // x image final

    sweepq_epilogue<NT, COLS, NW>(a, xl, COLS, col_base, (int)blockIdx.x);
#ifdef SWEEPF_TIMING
    if (a.op_mode == 98 && tid == 0 && blockIdx.x == 0) {
        long long *tmo = (long long *)a.op;
        tmo[0] = tm1 - tm0, tmo[1] = tm2 - tm1, tmo[2] = __builtin_readcyclecounter() - tm2, tmo[3] = t;
    }
#endif
    {
        const long long tot = wave_sum_ll((g4 == 0) ? (long long)t_lane : 0ll);
        if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
}

template <int NT, int NB, bool HAS_MASK, int NW>
__global__ __launch_bounds__(64 * NW, 1) void sweep_scd_f_kernel(const SweepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[]; // sweepf_lds_bytes(KP, NB, 16 NW)
    sweepf16_body<NT, NB, HAS_MASK, NW>(a, sf_smem);
}
