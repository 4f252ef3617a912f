// k_sweep_wgf.h -- SCD least-squares sweep of the fp32-operand mode: the workgroup-specialised kernel of k_sweep_wg.h
// (one chain wave + three update waves, one barrier per block of 4 coordinates) with the roles cut down to what the
// measured bottlenecks allow.  Same iteration as scd_ls_update (reference src/base_algorithms.cpp:3-37); what changes
// is who does what and in which arithmetic form:
//
//   * every row of G is divided by its diagonal (sweep_wg_const, common.h): the update waves carry nu = mu / G[q][q],
//     the chain wave needs no quotient and a coordinate step is delta = max(-x, -nu) -- one instruction -- followed by
//     the rank-1 update of the block's remaining nu.  The chain wave is issue bound (one VALU instruction per ~8
//     cycles per wave, any type): 41 instead of 71 instructions per block.
//   * x lives in LDS and is kept by the UPDATE waves (x += delta, one read-modify-write per lane and step); the chain
//     wave only reads it.  Columns that are done keep being computed (their lanes cost nothing) but their x is not
//     stored: the ballot of live columns travels with the "another sweep follows" word.
//   * update waves: the product that feeds `far` of the next block (urgent) is the only MFMA between the arrival of the
//     deltas and the store of far; the other tiles get the same deltas one step later (lazy), issued while the next
//     deltas are still on their way from LDS, and -- NT = 4 -- one of them behind the store of far.  The fp64 MFMA
//     blocks every later instruction of its wave for 64 cycles, so the ORDER of the four MFMAs of a step is what the
//     step time of this role consists of.  The step loop is fully unrolled: every accumulator element is a static
//     register (a register-indexed read next to the MFMAs makes the allocator copy whole accumulators).
//   * the rel-change flags are lane masks in SGPRs and the test is three instructions per coordinate.
//
// Results differ from k_sweep_wg.h by rounding only (1e-13 relative in the harness, scripts/exp/sweepwg_exp.hip);
// the strict fp64 mode keeps k_sweep_wg.h.  Needs k > 8 (three blocks: x of a block is rewritten one step after it
// was used and read again one step before it is used).
#pragma once
#include "common.h"
#include "k_sweep.h"
#include "k_sweep_wg.h"

// slope experiments (scripts/exp/sweepwg_exp.hip only): N extra independent fp64 FMAs at a point of the chain wave / extra MFMAs
#ifndef SWEEP_WG_XA
#define SWEEP_WG_XA 0
#endif
#ifndef SWEEP_WG_XB
#define SWEEP_WG_XB 0
#endif
#ifndef SWEEP_WG_XU
#define SWEEP_WG_XU 0
#endif
#define SWG_EXTRA(N, v)                                                                                                 \
    _Pragma("unroll") for (int xi = 0; xi < (N); xi++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(v));
// SWEEP_WG_MARKS (harness): time stamps inside a step, accumulated per role: mk[i] += stamp(i+1) - stamp(i)
#ifdef SWEEP_WG_MARKS
#define SWG_MARK(i) asm volatile("s_memtime %0" : "=s"(swg_mk[i]));
#define SWG_MARK_ACC()                                                                                                  \
    {                                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        _Pragma("unroll") for (int mi = 0; mi < 7; mi++) swg_acc[mi] += swg_mk[mi + 1] - swg_mk[mi];                     \
    }
#else
#define SWG_MARK(i)
#define SWG_MARK_ACC()
#endif
__host__ __device__ static inline int sweep_wgf_lds_bytes(int NT) { return sweep_wg_lds_bytes(NT) + 16; } // + the ballot words

// TAIL (k = 16 (NT - 1) + 1 or + 2): the one or two coordinates beyond a multiple of 16 form a block of their own that no
// update wave holds.  Their gradients live in two registers of the chain wave, which brings them up to date with every
// block's deltas itself (8 FMAs per step, always current: no far / near for that block), and the update waves carry
// NTU = NT - 1 tiles -- 3 instead of 4 MFMAs per step at k = 50, on the role that bounds the step.
template <int NT, bool HAS_MASK, bool TAIL = false>
__global__ __launch_bounds__(SWEEP_WG_THREADS) void sweep_scd_wgf_kernel(const SweepArgs a, const double *__restrict__ consts_g)
{
    constexpr int KP = 16 * NT, NB = 4 * NT;
    constexpr int NTU = TAIL ? NT - 1 : NT; // tiles of an update wave
    constexpr int XS = KP + 2; // row stride of the x image: 16-byte aligned rows, b128 reads of 16 lanes hit 16 distinct slots
    constexpr int CW = 0;      // the chain wave
    // Gz[b][t][g][l] = edited G[coord(t, l)][4b + g];  coordinate of (tile t, accumulator row M) = 4*((M/4)*NT + t) + M%4
    extern __shared__ __attribute__((aligned(16))) unsigned char sweep_wg_smem[]; // sweep_wg_lds_bytes(NT): > 64 KB at NT = 4
    double *Gz = (double *)sweep_wg_smem;                        // [NB * NT * 64]
    double *xl = Gz + NB * NT * 64;                              // [SWEEP_WG_LCOLS * XS]  x[column][coordinate]
    double(*dbuf)[SWEEP_WG_LCOLS * 4] = (double(*)[SWEEP_WG_LCOLS * 4])(xl + SWEEP_WG_LCOLS * XS); // [parity][column][g] deltas of a block
    double(*fbuf)[SWEEP_WG_LCOLS * 4] = dbuf + 2;                // [parity][column][s] far gradient of a block
    int *ctrl = (int *)(fbuf + 2);                               // [2]  "another sweep follows", by parity of its last step
    unsigned long long *actw = (unsigned long long *)(ctrl + 2); // [2]  ballot of the columns still being swept
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k;
    const int nbk = (k + 3) / 4;
    const int col_base = a.col0 + blockIdx.x * SWEEP_WG_COLS;

    auto edited = [&](int c, int kc) -> double { // regularisation edits of src/update_with_missing.cpp:20-24
        double g = a.Graw[(size_t)c * a.KPg + kc];
        if (c == kc && a.r0 != a.r1) g += a.r0 - a.r1;
        if (a.r1 != 0) g += a.r1;
        if (c == kc) g += NNLM_TINY;
        return g;
    };
    for (int e = tid; e < NB * NTU * 64; e += SWEEP_WG_THREADS) {
        const int b = e / (NTU * 64), rem = e % (NTU * 64), t = rem / 64, g = (rem % 64) / 16, l = rem % 16;
        const int c = 4 * ((l >> 2) * NTU + t) + (l & 3), kc = 4 * b + g;
        double gv = (c < k && kc < k) ? edited(c, kc) : 0.0;
        if (c < k) gv *= consts_g[(c >> 2) * SWEEP_WG_CONSTS + (c & 3)]; // row c / G[c][c]
        Gz[e] = gv;
    }
    for (int e = tid; e < SWEEP_WG_LCOLS * KP; e += SWEEP_WG_THREADS) {
        const int q = e / SWEEP_WG_LCOLS, c = e % SWEEP_WG_LCOLS, col = col_base + c;
        xl[c * XS + q] = (q < k && c < SWEEP_WG_COLS && col < a.ncols) ? a.X[(size_t)q * a.ldx + col] : 0.0;
    }
    for (int e = tid; e < 4 * SWEEP_WG_LCOLS * 4; e += SWEEP_WG_THREADS) (&dbuf[0][0])[e] = 0.0; // dbuf and fbuf
    __syncthreads();

    int t_lane = 0; // chain wave: sweeps done by this lane's column
#ifdef SWEEP_WG_TIMING
    unsigned long long swg_work = 0, swg_wait = 0;
#endif
#ifdef SWEEP_WG_MARKS
    unsigned long long swg_mk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, swg_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif

    if (wave != CW) {
        // ---------------------------------------------------------------- update wave: 16 columns, all coordinates
        const int u = (wave < CW) ? wave : wave - 1;
        const int l15 = lane & 15, lg = lane >> 4;
        const int cl = 16 * u + l15; // column inside the workgroup
        const int col = col_base + cl;
        const int cc = (col < a.ncols) ? col : a.col0;
        // element e = 4t + r of mu <-> block b = r*NT + t <-> coordinate 4b + lg (fp64 accumulator layout: row = lg + 4r)
        f64x16 mu;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int b = (e & 3) * NTU + (e >> 2); // meaningful for e < 4*NTU
            const int q = 4 * b + lg;
            double cv = 0.0;
            if (e < 4 * NTU && q < k)
                for (int s = 0; s < a.nslabs; s++) cv += a.Cx[(size_t)s * a.slab_stride + (size_t)q * a.ldc + cc];
            mu[e] = (e < 4 * NTU && q < k) ? ((a.r2 != 0) ? a.r2 - cv : -cv) : 0.0;
            if (e < 4 * NTU && q < k) mu[e] *= consts_g[(q >> 2) * SWEEP_WG_CONSTS + (q & 3)];
        }
        const double *gzl = Gz + lane; // + (b*NT + t)*64
#define SWEEP_WG_RANK4(bidx, coef)                                                                                       \
    _Pragma("unroll") for (int t2 = 0; t2 < NTU; t2++)                                                                  \
    {                                                                                                                   \
        f64x4 tile = f64x4{mu[4 * t2], mu[4 * t2 + 1], mu[4 * t2 + 2], mu[4 * t2 + 3]};                                  \
        tile = __builtin_amdgcn_mfma_f64_16x16x4f64(gzl[((bidx) * NTU + t2) * 64], (coef), tile, 0, 0, 0);                \
        mu[4 * t2] = tile[0];                                                                                           \
        mu[4 * t2 + 1] = tile[1];                                                                                       \
        mu[4 * t2 + 2] = tile[2];                                                                                       \
        mu[4 * t2 + 3] = tile[3];                                                                                       \
    }
        // mu = (L1 - c) + G x   (src/update_with_missing.cpp:39-41)
        for (int kb = 0; kb < nbk; kb++) {
            const double xb = xl[cl * XS + 4 * kb + lg];
            SWEEP_WG_RANK4(kb, xb)
        }
        fbuf[0][cl * 4 + lg] = mu[0]; // far of block 0 (element 0), read by the chain wave in step 0
        __syncthreads();

        // Step b (the chain wave works on block b, its deltas d_b do not exist yet):
        //   urgent  mu[tile of the next block] += G[:, b-1] d_{b-1}   -> `far` of the next block, published before the barrier
        //   lazy    mu[every tile but block b's] += G[:, b-2] d_{b-2}  -> issued FIRST, while d_{b-1} is still on its way
        //           from LDS (block b's own tile got d_{b-2} as the urgent product of step b-1)
        // so every delta reaches every tile once, one step later for the tiles nobody is waiting for, and only one MFMA
        // stands between the arrival of d_{b-1} and `far`.
        int par = 0, pb = nbk - 1; // step parity; block whose deltas arrive in this step (all zero in step 0)
        bool go = true;
        // x is kept by the update waves, x[block pb] += d, one LDS read-modify-write per lane and step (the chain wave
        // only reads x).  Columns that are done must keep their x: the ballot of live columns of the sweep the deltas
        // belong to (block pb = the last block <=> the previous sweep) comes from the chain wave with `ctrl`.
        // mine_cur: this lane's column was live in the sweep the arriving deltas belong to (switches after step 0)
        bool mine_next = ((actw[0] >> cl) & 1ull) != 0, mine_cur = mine_next; // (written before the barrier above)
        double *xcell = xl + cl * XS + lg; // + 4 * pb
        // A operands, fetched before the barrier of the previous step: gzl_[t] = G[tile t, block of d_prev]; gzu = G[tile of the
        // next block, block pb] for the urgent product.  When the next block is block 0 (tile 0) the urgent product is
        // issued on the static tile (t0 + 1) % NT with a ZERO operand and on tile 0 with the real one (gzw) -- a uniform
        // branch around an accumulator update makes the register allocator copy whole accumulators.
        double gzl_[NTU], gzu = 0.0, gzw = 0.0; // (first step: all deltas are zero)
        // deltas of this step and of the previous one in two registers that swap roles from step to step (static for even
        // NT): with `d_prev = d` the compiler gives both one register and the load of d has to wait for the lazy products
        constexpr bool ALT = (NTU % 2) == 0 && !TAIL;
        double dq[2] = {0.0, 0.0};
        // NT = 4: the third lazy product is issued LATE, behind the store of `far` (it runs while that store drains); its
        // operand has a register of its own (two, swapping like dq: the next one is requested before this one is used)
        #ifdef SWEEP_WG_NOLATE
        constexpr bool LATE = false;
#else
        constexpr bool LATE = NTU == 4;
#endif
        double gzlate[2] = {0.0, 0.0};
#pragma unroll
        for (int t2 = 0; t2 < NTU; t2++) gzl_[t2] = 0.0;
#define SWG_TILE_FMA(t2, A, B)                                                                                          \
    if (!(SWEEP_WG_ABL & 16)) {                                                                                                                   \
        f64x4 tile = f64x4{mu[4 * (t2)], mu[4 * (t2) + 1], mu[4 * (t2) + 2], mu[4 * (t2) + 3]};                          \
        tile = __builtin_amdgcn_mfma_f64_16x16x4f64((A), (B), tile, 0, 0, 0);                                            \
        mu[4 * (t2)] = tile[0];                                                                                         \
        mu[4 * (t2) + 1] = tile[1];                                                                                     \
        mu[4 * (t2) + 2] = tile[2];                                                                                     \
        mu[4 * (t2) + 3] = tile[3];                                                                                     \
    }
        while (go) {
#pragma unroll
            for (int r0 = 0; r0 < 4; r0++) { // fully unrolled: every accumulator element a step touches is a static register
#pragma unroll
                for (int t0 = 0; t0 < NTU; t0++) {
                    const int b = r0 * NTU + t0; // consecutive blocks, consecutive tiles
                    if (b >= (TAIL ? 4 * NTU : nbk)) continue; // wave-uniform (TAIL: the tail block has its own step below)
                    SWG_T0()
                    // d_{b-1} (and x of its block) are requested BEFORE the lazy products and awaited after them; written as
                    // instructions because the compiler sinks the loads below the MFMAs (d would share d_prev's register)
                    SWG_MARK(0)
                    double &d = dq[ALT ? (t0 & 1) : 0], &d_prev = dq[ALT ? ((t0 & 1) ^ 1) : 1];
                    double xold = 0.0;
                    asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"((unsigned)(size_t)&dbuf[par ^ 1][cl * 4 + lg]));
                    asm volatile("ds_read_b64 %0, %1" : "=v"(xold) : "v"((unsigned)(size_t)&xcell[4 * pb]));
                    const bool wrap = !(b + 1 < nbk); // the next block is block 0: tile 0, register 0
                    const int tn = (t0 + 1) % NTU;
                    const int rn = (t0 == NTU - 1) ? r0 + 1 : r0;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int uu = 0; uu < NTU - 1 - (LATE ? 1 : 0); uu++) { // lazy products; the tile of the next block first
                        const int t2 = (t0 + 1 + uu) % NTU;
                        SWG_TILE_FMA(t2, gzl_[t2], d_prev)
                    }
                    if (SWEEP_WG_XU) {
                        f64x4 xt = f64x4{0, 0, 0, 0};
#pragma unroll
                        for (int xi = 0; xi < SWEEP_WG_XU; xi++) xt = __builtin_amdgcn_mfma_f64_16x16x4f64(gzu, d, xt, 0, 0, 0);
                        asm volatile("" ::"v"(xt));
                    }
                    __builtin_amdgcn_sched_barrier(0); // keep the wait behind the lazy products
                    SWG_MARK(1)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d), "+v"(xold));
                    SWG_MARK(2)
                    // operands of the next step, requested as soon as their registers are free: lazy = G[:, pb] (with this d) ...
#pragma unroll
                    for (int t2 = 0; t2 < NTU; t2++) gzl_[t2] = gzl[(pb * NTU + t2) * 64];
                    if (LATE) gzlate[(t0 & 1) ^ 1] = gzl[(pb * NTU + ((wrap ? 0 : t0 + 1) + NTU - 1) % NTU) * 64];
                    // x before the urgent product (the fp64 MFMA holds up every VALU instruction behind it)
                    if (mine_cur) xcell[4 * pb] = xold + d;
                    SWG_MARK(3)
                    SWG_TILE_FMA(tn, gzu, d)
                    if (tn != 0 && wrap) SWG_TILE_FMA(0, gzw, d)
                    SWG_MARK(4)
                    { // ... urgent = G[tile of the block after the next, b]
                        const int nb_ = wrap ? 0 : b + 1;
                        const bool nwrap = !(nb_ + 1 < nbk);
                        const int ntn = ((nb_ % NTU) + 1) % NTU; // = the static tn of the next step
                        const double gu = gzl[(b * NTU + (nwrap ? 0 : ntn)) * 64];
                        gzu = (nwrap && ntn != 0) ? 0.0 : gu;
                        gzw = gu;
                    }
                    // far of the next block: STATIC tile, register picked with selects (a register-indexed read right behind
                    // the MFMAs is not covered by the compiler's MFMA->VALU hazard handling)
                    const double far = wrap ? mu[0] : mu[4 * tn + (rn & 3)];
                    fbuf[par ^ 1][cl * 4 + lg] = far;
                    SWG_MARK(5)
                    if (LATE) {
                        __builtin_amdgcn_sched_barrier(0); // behind the store of far, not in front of it
                        SWG_TILE_FMA((t0 + NTU - 1) % NTU, gzlate[t0 & 1], d_prev)
                    }
                    if (!ALT) d_prev = d;
                    else if (wrap && (t0 & 1) == 0) dq[1] = dq[0], gzlate[0] = gzlate[1]; // the next step is block 0, an even step again
#ifdef SWEEP_WG_MARKS
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    SWG_MARK(6)
#endif
                    SWG_SYNC(swg_work, swg_wait)
                    SWG_MARK(7)
                    SWG_MARK_ACC()
                    pb = b;
                    if (b == nbk - 1) {
                        go = ctrl[par] != 0;
                        mine_next = ((actw[par] >> cl) & 1ull) != 0;
                    }
                    if (b == 0) mine_cur = mine_next;
                    par ^= 1;
                }
            }
            if (TAIL) {
                // The chain wave is on the tail block (b = nbk - 1, no tile).  The step before applied d_{b-2} to tile 0 as if it were
                // urgent (its `far` is not read), so the lazy products of this step go to the other tiles; the deltas of block
                // b - 1 go to tile 0, whose first block is next.
                const int b = nbk - 1;
                SWG_T0()
                double &d = dq[0], &d_prev = dq[1];
                double xold = 0.0;
                asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"((unsigned)(size_t)&dbuf[par ^ 1][cl * 4 + lg]));
                asm volatile("ds_read_b64 %0, %1" : "=v"(xold) : "v"((unsigned)(size_t)&xcell[4 * pb]));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t2 = 1; t2 < NTU; t2++) SWG_TILE_FMA(t2, gzl_[t2], d_prev)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d), "+v"(xold));
#pragma unroll
                for (int t2 = 0; t2 < NTU; t2++) gzl_[t2] = gzl[(pb * NTU + t2) * 64];
                if (mine_cur) xcell[4 * pb] = xold + d;
                SWG_TILE_FMA(0, gzw, d)
                gzu = gzl[(b * NTU + 1 % NTU) * 64]; // step 0: the tail block's deltas to the tile of block 1
                gzw = gzu;
                fbuf[par ^ 1][cl * 4 + lg] = mu[0];
                d_prev = d;
                SWG_SYNC(swg_work, swg_wait)
                pb = b;
                go = ctrl[par] != 0;
                mine_next = ((actw[par] >> cl) & 1ull) != 0;
                par ^= 1;
            }
        }
        if (mine_cur) xcell[4 * pb] += dbuf[par ^ 1][cl * 4 + lg]; // deltas of the very last step
#undef SWG_TILE_FMA
#undef SWEEP_WG_RANK4
    } else {
        // ---------------------------------------------------------------- chain wave: lane = column
        const int col = col_base + lane;
        const bool in_range = lane < SWEEP_WG_COLS && col < a.ncols;
        unsigned long long mword = 0ull;
        if (HAS_MASK) mword = a.mask[in_range ? col : a.col0];
        const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
        bool act = in_range && !(HAS_MASK && ((mword & kmask) == kmask)); // arma::all(mask.col(j)) -> column skipped
        const auto *cdat = (const __attribute__((address_space(4))) double *)(unsigned long long)consts_g; // uniform reads -> s_load
        double *xrow = xl + lane * XS;
        const double tol = a.rel_tol, tolh = 0.5 * tol, tolhe = 0.5 * tol * NNLM_TINY;
        unsigned t = 0;
        int par = 0;
        bool go = a.max_iter > 0 && __any(act);
        // constants of a block: fetched through the scalar cache one step AHEAD (before the barrier of the previous step)
        struct Consts {
            double gl[6]; // scaled G[4b+s2][4b+s], s2 > s
            double gt[8]; // TAIL: scaled G[tail coordinate s][4b+g] at 4s + g
        };
        auto load_chain = [&](int b, Consts &c) {
            const auto *cb = cdat + b * SWEEP_WG_CONSTS;
#pragma unroll
            for (int i = 0; i < 6; i++) c.gl[i] = cb[8 + i];
            if (TAIL) {
#pragma unroll
                for (int i = 0; i < 8; i++) c.gt[i] = cb[32 + i];
            }
        };
        Consts cc;
        load_chain(0, cc);
        double gn[16]; // G[this block][previous block]: the near part, record of the previous block
#pragma unroll
        for (int i = 0; i < 16; i++) gn[i] = 0.0;
        double dd[4] = {0, 0, 0, 0}; // deltas of the previous step
        double xdummy = 0.5;
        (void)xdummy;
        f64x2 x01 = *(const f64x2 *)&xrow[0], x23 = *(const f64x2 *)&xrow[2]; // x of the next block
        // TAIL: the (scaled) gradients of the tail coordinates, (L1 - c) / G[q][q] + sum_j G'[q][j] x_j  (update_with_missing.cpp:39-41)
        double mut[2] = {0.0, 0.0};
        if (TAIL) {
            const int tc = 4 * (nbk - 1), ccol = in_range ? col : a.col0;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (tc + s >= k) continue; // wave-uniform
                double cv = 0.0;
                for (int sl = 0; sl < a.nslabs; sl++) cv += a.Cx[(size_t)sl * a.slab_stride + (size_t)(tc + s) * a.ldc + ccol];
                mut[s] = ((a.r2 != 0) ? a.r2 - cv : -cv) * cdat[(nbk - 1) * SWEEP_WG_CONSTS + s];
            }
            for (int bb = 0; bb < nbk; bb++) {
                const auto *cb = cdat + bb * SWEEP_WG_CONSTS + 32;
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int g = 0; g < 4; g++) mut[s] = __builtin_fma(cb[4 * s + g], xrow[4 * bb + g], mut[s]);
            }
        }
        {
            const unsigned long long bal = __ballot(act);
            if (lane == 0) actw[0] = bal;
        }
        __syncthreads(); // far of block 0 is in fbuf[0]
        if (!go) { // nothing to do: release the update waves through the normal protocol (one full sweep of idle steps)
            for (int b = 0; b < nbk; b++) {
                if (b == nbk - 1 && lane == 0) ctrl[par] = 0, actw[par] = 0ull;
                __syncthreads();
                par ^= 1;
            }
        }
        while (go) {
            bool flag = 0.0 > tol; // rel_err starts each sweep at 0: a negative rel_tol never stops (a lane mask in SGPRs)
            bool tests_on = !(0.0 > tol);   // wave-uniform: some live column has not moved by more than rel_tol yet
            for (int b = 0; b < nbk; b++) {
                SWG_T0()
                SWG_MARK(0)
                const f64x2 f01 = *(const f64x2 *)&fbuf[par][lane * 4], f23 = *(const f64x2 *)&fbuf[par][lane * 4 + 2];
                // the part of the previous block's gradient update this block cannot wait for (hides the LDS latency of far)
                double near[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (SWEEP_WG_ABL & 4) {
                        near[s] = dd[s];
                        continue;
                    }
                    double acc = dd[0] * gn[4 * s];
                    acc = __builtin_fma(dd[1], gn[4 * s + 1], acc);
                    acc = __builtin_fma(dd[2], gn[4 * s + 2], acc);
                    acc = __builtin_fma(dd[3], gn[4 * s + 3], acc);
                    near[s] = acc;
                }
                SWG_EXTRA(SWEEP_WG_XA, xdummy)
                SWG_MARK(1)
                double m[4] = {f01[0] + near[0], f01[1] + near[1], f23[0] + near[2], f23[1] + near[3]};
                if (TAIL && b == nbk - 1) m[0] = mut[0], m[1] = mut[1], m[2] = 0.0, m[3] = 0.0; // (always current: no far, no near)
#ifdef SWEEP_WG_MARKS
                asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]));
#endif
                SWG_MARK(2)
                SWG_EXTRA(SWEEP_WG_XB, xdummy)
                const double xs[4] = {x01[0], x01[1], x23[0], x23[1]};
                const double gl[4][4] = {{0, 0, 0, 0}, {cc.gl[0], 0, 0, 0}, {cc.gl[1], cc.gl[2], 0, 0}, {cc.gl[3], cc.gl[4], cc.gl[5], 0}};
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (SWEEP_WG_ABL & 8) {
                        dd[s] = act ? m[s] * 1e-30 : 0.0;
                        continue;
                    }
                    // Padded coordinates (q >= k) are inert: x = mu = 0, G = identity.
                    // Columns that are done (act == false) keep being computed -- their lanes cost nothing -- but the
                    // update waves do not store their x, so they stay exactly as the reference leaves them; their
                    // deltas only reach their own columns of the update waves.
                    // m = mu / G[q][q]: delta = max(x - m, 0) - x = max(-x, -m); written as the instruction because
                    // fmax() makes the compiler canonicalise x first and negate the result afterwards (3 instructions)
                    asm("v_max_f64 %0, -%1, -%2" : "=v"(dd[s]) : "v"(xs[s]), "v"(m[s]));
                    if (HAS_MASK && ((mword >> (4 * b + s)) & 1ull)) dd[s] = 0.0;
#pragma unroll
                    for (int s2 = s + 1; s2 < 4; s2++) m[s2] = __builtin_fma(dd[s], gl[s2][s], m[s2]);
                }
                if (TAIL) { // the tail gradients see every block's deltas at once (the tail block's own included: G'[q][q] = 1)
#pragma unroll
                    for (int s = 0; s < 2; s++)
#pragma unroll
                        for (int g = 0; g < 4; g++) mut[s] = __builtin_fma(cc.gt[4 * s + g], dd[g], mut[s]);
                }
#ifdef SWEEP_WG_MARKS
                asm volatile("" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]));
#endif
                SWG_MARK(3)
                if (!(SWEEP_WG_ABL & 2)) {
                    *(f64x2 *)&dbuf[par][lane * 4] = f64x2{dd[0], dd[1]};
                    *(f64x2 *)&dbuf[par][lane * 4 + 2] = f64x2{dd[2], dd[3]};
                }
                // rel-change tests (src/base_algorithms.cpp:29-32), division-free.  Only "did ANY coordinate of the sweep move
                // by more than rel_tol" matters, so once every column of the wave has its flag the tests of the remaining
                // blocks of this sweep are skipped (wave-uniform branch; same decisions, ~20 fp64 instructions less)
                SWG_MARK(4)
                if (tests_on) {
#pragma unroll
                    for (int s = 0; s < 4; s++) // 2|d| > tol (x + d + x + eps) in three instructions: |d| > (tol/2)(2x + d) + tol eps/2
                        flag |= fabs(dd[s]) > __builtin_fma(tolh, __builtin_fma(2.0, xs[s], dd[s]), tolhe); // (|=: no short-circuit branches)
                    tests_on = !__all(flag | !act);
                }
                const int nb = (b + 1 < nbk) ? b + 1 : 0;
                if (b == nbk - 1) { // end of a sweep (src/base_algorithms.cpp:35: stop when rel_err <= rel_tol)
                    if (act) {
                        t_lane++;
                        act = flag;
                    }
                    t++;
                    go = t < a.max_iter && __any(act);
                    if (lane == 0) ctrl[par] = go ? 1 : 0;
                    {
                        const unsigned long long bal = __ballot(act);
                        if (lane == 0) actw[par] = bal;
                    }
                    flag = 0.0 > tol;
                    tests_on = !(0.0 > tol);
                }
                // next step's operands: constants through the scalar cache, x from this wave's own LDS rows
                if (!(SWEEP_WG_ABL & 1)) {
                    const auto *cb = cdat + b * SWEEP_WG_CONSTS + 16;
#pragma unroll
                    for (int i = 0; i < 16; i++) gn[i] = cb[i];
                    load_chain(nb, cc);
                }
                x01 = *(const f64x2 *)&xrow[4 * nb];
                x23 = *(const f64x2 *)&xrow[4 * nb + 2];
                SWG_MARK(5)
#ifdef SWEEP_WG_MARKS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                SWG_MARK(6)
#endif
                SWG_SYNC(swg_work, swg_wait)
                SWG_MARK(7)
                SWG_MARK_ACC()
                par ^= 1;
            }
        }
    }
    __syncthreads(); // x image final

    // The epilogue reads the launch arguments AGAIN from the kernarg segment: held in SGPRs across the step loops they push
    // the chain wave's block constants out (8 v_readlane per step to get the spilled pointer of the constants image back).
    const SweepArgs *ap = (const SweepArgs *)__builtin_amdgcn_kernarg_segment_ptr(); // first kernel parameter
    asm volatile("" : "+s"(ap));
    const SweepArgs &ea = *ap;

    float xmax = 0.0f;
    for (int e = tid; e < SWEEP_WG_COLS * KP; e += SWEEP_WG_THREADS) {
        const int q = e / SWEEP_WG_COLS, c = e % SWEEP_WG_COLS, col = col_base + c;
        if (q < k && col < ea.ncols) {
            const double xv = xl[c * XS + q];
            xmax = fmaxf(xmax, fabsf((float)xv));
            ea.Xout[(size_t)q * ea.ldo + (col - ea.ocol0)] = xv;
            if (ea.op_mode == 1) {
                if (ea.op_f64) ((double *)ea.op)[(size_t)q * ea.op_ld + col] = xv;
                else ((float *)ea.op)[(size_t)q * ea.op_ld + col] = (float)xv;
            }
        }
    }
    if (ea.op_mode == 2) { // [col][op_ld], kq fastest: consecutive threads write consecutive kq of one column
        for (int e = tid; e < SWEEP_WG_COLS * KP; e += SWEEP_WG_THREADS) {
            const int c = e / KP, q = e % KP, col = col_base + c;
            if (q < k && col < ea.ncols) {
                const double xv = xl[c * XS + q];
                if (ea.op_f64) ((double *)ea.op)[(size_t)col * ea.op_ld + q] = xv;
                else ((float *)ea.op)[(size_t)col * ea.op_ld + q] = (float)xv;
            }
        }
    }
    if (ea.maxbits) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
        if (lane == 0 && xmax > 0.0f) atomicMax(ea.maxbits, __float_as_uint(xmax));
    }
    if (ea.gram_slabs) {
        // Gram partial sums of this workgroup's columns (what gram_partial_kernel, k_gram.h, would compute after reading the
        // factor back; rows of padded / out-of-range columns of the x image are zero): X X^T over the 48 columns with
        // v_mfma_f64_16x16x4_f64, upper tiles dealt to the four wavefronts, same slab layout as gram_partial_kernel.
        // ~1 us per workgroup.  The slabs are folded by gram_fold_kernel (k_gram.h) -- NOT here: a "last workgroup of a
        // group adds them" step needs __threadfence(), and two of those per workgroup cost 65 us of the kernel's tail.
        const int l15 = lane & 15, lg = lane >> 4;
        double *slab = ea.gram_slabs + (size_t)blockIdx.x * KP * KP;
        int tix = 0;
#pragma unroll
        for (int ta = 0; ta < NT; ta++)
#pragma unroll
            for (int tb = ta; tb < NT; tb++) {
                if ((tix++ & 3) != wave) continue;
                f64x4 acc = f64x4{0, 0, 0, 0};
#pragma unroll
                for (int s4 = 0; s4 < SWEEP_WG_COLS / 4; s4++) {
                    const double *xr = xl + (4 * s4 + lg) * XS + l15;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[16 * ta], xr[16 * tb], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) slab[(16 * ta + lg + 4 * r) * KP + 16 * tb + l15] = acc[r];
            }
    }
#ifdef SWEEP_WG_TIMING
    if (ea.op && blockIdx.x == 0 && lane == 0 && (wave == CW || wave == 1)) {
        unsigned long long *dbg = (unsigned long long *)ea.op; // harness: [role][work, wait]
        dbg[(wave == CW ? 0 : 2)] = swg_work;
        dbg[(wave == CW ? 0 : 2) + 1] = swg_wait;
#ifdef SWEEP_WG_MARKS
        for (int mi = 0; mi < 7; mi++) dbg[4 + (wave == CW ? 0 : 7) + mi] = swg_acc[mi];
#endif
    }
#endif
    if (wave == CW) {
        long long tot = wave_sum_ll((long long)t_lane);
        if (lane == 0 && tot) atomicAdd(ea.sweeps, (unsigned long long)tot);
    }
}
