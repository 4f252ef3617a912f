// k_sweep_wg.h -- SCD least-squares sweep, workgroup-specialised: one "chain" wavefront + four "update" wavefronts.
//
// Same iteration as scd_ls_update (reference src/base_algorithms.cpp:3-37): coordinates strictly in order 0..k-1, each
// step sees every earlier update -- in the REFERENCE'S arithmetic (reciprocal + Markstein-corrected quotient = the correctly
// rounded mu / G[q][q], max, delta): the SCD sweep of the strict fp64 mode and of ranks below 9.  (The fp32-operand mode runs
// k_sweep_q.h: one wavefront per 16 columns, rows of G divided by their diagonal.)  The work of a block of 4 coordinates is
// split by ROLE across the wavefronts of a workgroup that owns 48 columns, synchronised by one s_barrier per block:
//
//   chain wave (1):  lane = column.  Per block b: m = far[b] + near, the 4 dependent coordinate steps, x_new, the rel-change
//     tests, then `near` = G[next block, b] * d_b (16 FMAs) -- the one part of the gradient update the NEXT block cannot wait
//     for.  Block constants are wave-uniform and come through the scalar cache (s_load from a small image written by
//     sweep_consts_kernel) -- no LDS traffic, no VGPRs.  x lives in LDS ([column][coordinate], padded rows).
//   update waves (3, 16 columns each):  hold the gradient mu of ALL coordinates in fp64 MFMA accumulators
//     (v_mfma_f64_16x16x4_f64: coordinate 16 t + 4 r + lg in register r of tile t, lane (lg, column)).  During block b they apply
//     the deltas of block b-1, mu += G[:, b-1] * d_{b-1}, and publish `far` = mu[block b+1] -- which therefore holds every
//     update except d_b, the one the chain wave adds itself as `near`.  They have a whole block time of slack.
//
// Exchange through LDS, double-buffered by step parity: dbuf (deltas, chain -> update), fbuf (far, update -> chain).
// Differences from the reference's arithmetic stay sub-ulp: a coordinate's gradient is assembled as far + near instead of
// one running sum (and far accumulates in the matrix core's order).
#pragma once
#include "common.h"
#include "k_sweep.h"

#define SWEEP_WG_THREADS 256 // 4 wavefronts = one per SIMD: the chain wave + 3 update waves
#define SWEEP_WG_COLS 48     // columns per workgroup (16 per update wave; lanes 48..63 of the chain wave idle)
#define SWEEP_WG_LCOLS 64    // rows of the LDS images (one per chain-wave lane)
// SWEEP_WG_TIMING (scripts/exp/sweepwg_exp.hip only): cycles spent working / waiting at the step barrier, per role
#ifndef SWEEP_WG_ABL
#define SWEEP_WG_ABL 0 // ablation bits (scripts/exp/sweepwg_exp.hip only): 1 no constant reloads, 2 no LDS writes, 4 no near, 8 no chain
#endif
#ifdef SWEEP_WG_TIMING
#define SWG_T0() const unsigned long long swg_t0 = __builtin_readcyclecounter();
#define SWG_SYNC(work, wait)                                                                                            \
    {                                                                                                                   \
        const unsigned long long swg_t1 = __builtin_readcyclecounter();                                                 \
        __syncthreads();                                                                                                \
        const unsigned long long swg_t2 = __builtin_readcyclecounter();                                                 \
        work += swg_t1 - swg_t0;                                                                                        \
        wait += swg_t2 - swg_t1;                                                                                        \
    }
#else
#define SWG_T0()
#define SWG_SYNC(work, wait) __syncthreads();
#endif
__host__ __device__ static inline int sweep_wg_lds_bytes(int NT)
{
    const int KP = 16 * NT, NB = 4 * NT;
    return (NB * NT * 64 + SWEEP_WG_LCOLS * (KP + 2) + 4 * SWEEP_WG_LCOLS * 4) * 8 + 16;
}

// Constants image for the chain wave, one record per block b (nb = the block visited after b, cyclic over the
// blocks that hold real coordinates):
//   [0..3]   1 / G[4b+s][4b+s]          [4..7]  G[4b+s][4b+s]
//   [8..13]  G[4b+s2][4b+s], s2 > s, in the order (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)        [14,15] unused
//   [16..31] G[4nb+s][4b+g] at 16 + 4s + g
// with the regularisation edits of src/update_with_missing.cpp:20-24; padded coordinates: diagonal 1, rest 0.
// (records are produced by sweep_wg_const, common.h -- shared with gram_reduce_consts_kernel)
__global__ __launch_bounds__(256) void sweep_consts_kernel(const double *__restrict__ Graw, int KPg, int k, double r0, double r1,
                                                           double *__restrict__ consts)
{
    const int nbk = (k + 3) / 4;
    auto edited = [&](int c, int kc) -> double {
        if (c >= k || kc >= k) return (c == kc) ? 1.0 : 0.0;
        double g = Graw[(size_t)c * KPg + kc];
        if (c == kc && r0 != r1) g += r0 - r1;
        if (r1 != 0) g += r1;
        if (c == kc) g += NNLM_TINY;
        return g;
    };
    for (int e = threadIdx.x; e < nbk * SWEEP_WG_CONSTS; e += blockDim.x)
        consts[e] = sweep_wg_const(edited, k, nbk, e / SWEEP_WG_CONSTS, e % SWEEP_WG_CONSTS);
}

template <int NT, bool HAS_MASK>
__global__ __launch_bounds__(SWEEP_WG_THREADS) void sweep_scd_wg_kernel(const SweepArgs a, const double *__restrict__ consts_g)
{
    constexpr int KP = 16 * NT, NB = 4 * NT;
    constexpr int XS = KP + 2; // row stride of the x image: 16-byte aligned rows, b128 reads of 16 lanes hit 16 distinct slots
    constexpr int CW = 0;      // the chain wave
    // Gz[b][t][g][l] = edited G[coord(t, l)][4b + g];  coordinate of (tile t, accumulator row M) = 4*((M/4)*NT + t) + M%4
    extern __shared__ __attribute__((aligned(16))) unsigned char sweep_wg_smem[]; // sweep_wg_lds_bytes(NT): > 64 KB at NT = 4
    double *Gz = (double *)sweep_wg_smem;                        // [NB * NT * 64]
    double *xl = Gz + NB * NT * 64;                              // [SWEEP_WG_LCOLS * XS]  x[column][coordinate]
    double(*dbuf)[SWEEP_WG_LCOLS * 4] = (double(*)[SWEEP_WG_LCOLS * 4])(xl + SWEEP_WG_LCOLS * XS); // [parity][column][g] deltas of a block
    double(*fbuf)[SWEEP_WG_LCOLS * 4] = dbuf + 2;                // [parity][column][s] far gradient of a block
    int *ctrl = (int *)(fbuf + 2);                               // [2]
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
    for (int e = tid; e < NB * NT * 64; e += SWEEP_WG_THREADS) {
        const int b = e / (NT * 64), rem = e % (NT * 64), t = rem / 64, g = (rem % 64) / 16, l = rem % 16;
        const int c = 4 * ((l >> 2) * NT + t) + (l & 3), kc = 4 * b + g;
        Gz[e] = (c < k && kc < k) ? edited(c, kc) : 0.0;
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
            const int b = (e & 3) * NT + (e >> 2); // meaningful for e < 4*NT
            const int q = 4 * b + lg;
            double cv = 0.0;
            if (e < NB && q < k)
                for (int s = 0; s < a.nslabs; s++) cv += a.Cx[(size_t)s * a.slab_stride + (size_t)q * a.ldc + cc];
            mu[e] = (e < NB && q < k) ? ((a.r2 != 0) ? a.r2 - cv : -cv) : 0.0;
        }
        const double *gzl = Gz + lane; // + (b*NT + t)*64
#define SWEEP_WG_RANK4(bidx, coef)                                                                                       \
    _Pragma("unroll") for (int t2 = 0; t2 < NT; t2++)                                                                   \
    {                                                                                                                   \
        f64x4 tile = f64x4{mu[4 * t2], mu[4 * t2 + 1], mu[4 * t2 + 2], mu[4 * t2 + 3]};                                  \
        tile = __builtin_amdgcn_mfma_f64_16x16x4f64(gzl[((bidx) * NT + t2) * 64], (coef), tile, 0, 0, 0);                 \
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

        int par = 0, pb = nbk - 1; // step parity; block whose deltas are applied in this step (all zero in step 0)
        bool go = true;
        // A operands of the first step (block pb); afterwards they are fetched before the barrier of the previous step
        double gz[NT];
#pragma unroll
        for (int t2 = 0; t2 < NT; t2++) gz[t2] = gzl[(pb * NT + t2) * 64];
        while (go) {
#pragma nounroll
            for (int r0 = 0; r0 < 4; r0++) {
#pragma unroll
                for (int t0 = 0; t0 < NT; t0++) {
                    const int b = r0 * NT + t0; // consecutive blocks, consecutive tiles (tile index static, register index r0)
                    if (b >= nbk) continue;     // wave-uniform
                    SWG_T0()
                    const double d = dbuf[par ^ 1][cl * 4 + lg];
                    // the tile of the NEXT block first: its far value is what the chain wave waits for
#pragma unroll
                    for (int uu = 0; uu < NT; uu++) {
                        const int t2 = (t0 + 1 + uu) % NT;
                        f64x4 tile = f64x4{mu[4 * t2], mu[4 * t2 + 1], mu[4 * t2 + 2], mu[4 * t2 + 3]};
                        tile = __builtin_amdgcn_mfma_f64_16x16x4f64(gz[t2], d, tile, 0, 0, 0);
                        mu[4 * t2] = tile[0];
                        mu[4 * t2 + 1] = tile[1];
                        mu[4 * t2 + 2] = tile[2];
                        mu[4 * t2 + 3] = tile[3];
                    }
                    // far of the next block: STATIC tile, register picked with selects (a register-indexed read right
                    // behind the MFMAs is not covered by the compiler's MFMA->VALU hazard handling)
                    double far;
                    if (b + 1 < nbk) {
                        const int tn = (t0 + 1) % NT;
                        const int rn = (t0 == NT - 1) ? r0 + 1 : r0;
                        far = (rn == 0) ? mu[4 * tn] : (rn == 1) ? mu[4 * tn + 1] : (rn == 2) ? mu[4 * tn + 2] : mu[4 * tn + 3];
                    } else
                        far = mu[0];
                    fbuf[par ^ 1][cl * 4 + lg] = far;
#pragma unroll
                    for (int t2 = 0; t2 < NT; t2++) gz[t2] = gzl[(b * NT + t2) * 64]; // next step applies block b
                    SWG_SYNC(swg_work, swg_wait)
                    pb = b;
                    if (b == nbk - 1) go = ctrl[par] != 0;
                    par ^= 1;
                }
            }
        }
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
        const double tol = a.rel_tol;
        unsigned t = 0;
        int par = 0;
        bool go = a.max_iter > 0 && __any(act);
        // constants of a block: fetched through the scalar cache one step AHEAD (before the barrier of the previous step)
        struct Consts {
            double rg[4], gd[4], gl[6];
        };
        auto load_chain = [&](int b, Consts &c) {
            const auto *cb = cdat + b * SWEEP_WG_CONSTS;
#pragma unroll
            for (int i = 0; i < 4; i++) c.rg[i] = cb[i], c.gd[i] = cb[4 + i];
#pragma unroll
            for (int i = 0; i < 6; i++) c.gl[i] = cb[8 + i];
        };
        Consts cc;
        load_chain(0, cc);
        double gn[16]; // G[this block][previous block]: the near part, record of the previous block
#pragma unroll
        for (int i = 0; i < 16; i++) gn[i] = 0.0;
        double dd[4] = {0, 0, 0, 0}; // deltas of the previous step
        f64x2 x01 = *(const f64x2 *)&xrow[0], x23 = *(const f64x2 *)&xrow[2]; // x of the next block: only this wave writes x
        __syncthreads(); // far of block 0 is in fbuf[0]
        if (!go) { // nothing to do: release the update waves through the normal protocol (one full sweep of idle steps)
            for (int b = 0; b < nbk; b++) {
                if (b == nbk - 1 && lane == 0) ctrl[par] = 0;
                __syncthreads();
                par ^= 1;
            }
        }
        while (go) {
            int flag = (0.0 > tol) ? 1 : 0; // rel_err starts each sweep at 0: a negative rel_tol never stops
            for (int b = 0; b < nbk; b++) {
                SWG_T0()
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
                double m[4] = {f01[0] + near[0], f01[1] + near[1], f23[0] + near[2], f23[1] + near[3]};
                const double xs[4] = {x01[0], x01[1], x23[0], x23[1]};
                const double gl[4][4] = {{0, 0, 0, 0}, {cc.gl[0], 0, 0, 0}, {cc.gl[1], cc.gl[2], 0, 0}, {cc.gl[3], cc.gl[4], cc.gl[5], 0}};
                double xn[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (SWEEP_WG_ABL & 8) {
                        dd[s] = act ? m[s] * 1e-30 : 0.0;
                        xn[s] = xs[s];
                        continue;
                    }
                    const double q0 = m[s] * cc.rg[s];
                    const double rr = __builtin_fma(-q0, cc.gd[s], m[s]);
                    const double quo = __builtin_fma(rr, cc.rg[s], q0); // = mu / G[q][q], correctly rounded
                    const double tmp = fmax(xs[s] - quo, 0.0);
                    bool upd = act;
                    if (HAS_MASK) upd = upd && !((mword >> (4 * b + s)) & 1ull);
                    // padded coordinates (q >= k) are inert by construction: x = mu = 0, G = identity there
                    dd[s] = upd ? tmp - xs[s] : 0.0;
                    xn[s] = upd ? tmp : xs[s];
#pragma unroll
                    for (int s2 = s + 1; s2 < 4; s2++) m[s2] = __builtin_fma(dd[s], gl[s2][s], m[s2]);
                }
                if (!(SWEEP_WG_ABL & 2)) {
                    *(f64x2 *)&dbuf[par][lane * 4] = f64x2{dd[0], dd[1]};
                    *(f64x2 *)&dbuf[par][lane * 4 + 2] = f64x2{dd[2], dd[3]};
                    *(f64x2 *)&xrow[4 * b] = f64x2{xn[0], xn[1]};
                    *(f64x2 *)&xrow[4 * b + 2] = f64x2{xn[2], xn[3]};
                }
                // rel-change tests (src/base_algorithms.cpp:29-32), division-free.  Only "did ANY coordinate of the sweep move
                // by more than rel_tol" matters, so once every column of the wave has its flag the tests of the remaining
                // blocks of this sweep are skipped (wave-uniform branch; same decisions, ~20 fp64 instructions less)
                if (!__all(flag != 0 || !act)) {
#pragma unroll
                    for (int s = 0; s < 4; s++) flag |= ((2 * fabs(dd[s])) > tol * (xn[s] + xs[s] + NNLM_TINY)) ? 1 : 0;
                }
                const int nb = (b + 1 < nbk) ? b + 1 : 0;
                if (b == nbk - 1) { // end of a sweep (src/base_algorithms.cpp:35: stop when rel_err <= rel_tol)
                    if (act) {
                        t_lane++;
                        act = flag != 0;
                    }
                    t++;
                    go = t < a.max_iter && __any(act);
                    if (lane == 0) ctrl[par] = go ? 1 : 0;
                    flag = (0.0 > tol) ? 1 : 0;
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
                SWG_SYNC(swg_work, swg_wait)
                par ^= 1;
            }
        }
    }
    __syncthreads(); // x image final

    for (int e = tid; e < SWEEP_WG_COLS * KP; e += SWEEP_WG_THREADS) {
        const int q = e / SWEEP_WG_COLS, c = e % SWEEP_WG_COLS, col = col_base + c;
        if (q < k && col < a.ncols) {
            const double xv = xl[c * XS + q];
            a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xv;
            if (a.op_mode == 1) {
                if (a.op_f64) ((double *)a.op)[(size_t)q * a.op_ld + col] = xv;
                else ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xv;
            }
        }
    }
#ifdef SWEEP_WG_TIMING
    if (a.op && blockIdx.x == 0 && lane == 0 && (wave == CW || wave == 1)) {
        unsigned long long *dbg = (unsigned long long *)a.op; // harness: [role][work, wait]
        dbg[(wave == CW ? 0 : 2)] = swg_work;
        dbg[(wave == CW ? 0 : 2) + 1] = swg_wait;
    }
#endif
    if (wave == CW) {
        long long tot = wave_sum_ll((long long)t_lane);
        if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
}
