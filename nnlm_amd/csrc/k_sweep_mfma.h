// k_sweep_mfma.h -- SCD least-squares sweep with the gradient update on the matrix cores.
//
// Same arithmetic as scd_ls_update (reference src/base_algorithms.cpp:3-37) and as sweep_ls_kernel (k_sweep.h):
// coordinates are visited strictly in order 0..k-1, each one sees every earlier update (Gauss-Seidel).  What changes
// is WHERE the k FMAs of `mu += (tmp - Hj(k)) * WtW.col(k)` (src/base_algorithms.cpp:27) run.  fp64 VALU issues one
// instruction per 8 cycles per wavefront on gfx950 (measured), so the VALU version is bound by the 14 FMAs + row
// fetches per coordinate.  Here coordinates are processed in blocks of 4:
//   * inside a block the 4 dependent steps only need mu of those 4 coordinates and the 4x4 diagonal block of G:
//     every lane runs that short chain redundantly for its column on values all-gathered across the wavefront's four
//     16-lane rows with v_permlane32_swap / v_permlane16_swap (3 VALU ops per dword, no LDS);
//   * the rank-4 update of all the OTHER coordinates, mu[others] += G[others, block] * d[block], is one
//     v_mfma_f64_16x16x4_f64 per 16 coordinates: A = 16 coordinates x 4 block columns of G (diagonal block zeroed),
//     B = the 4 deltas x 16 columns of the factor, C/D = mu.  In the f64 accumulator layout (row = (lane>>4) + 4*reg,
//     column = lane & 15) lane (g, c) holds mu of coordinates {16t + 4r + g} of column c -- i.e. exactly one
//     coordinate of every block -- so the lane's own delta IS its B operand and no broadcast is needed for the MFMA.
// A wavefront solves 16 columns; mu and x are 4*NT doubles per lane.
//
// The MFMA accumulates the four products of a block in its own order and the block's own coordinates are updated
// with 4 sequential FMAs: same sub-ulp freedom as documented in k_sweep.h.
#pragma once
#include "common.h"
#include "k_sweep.h"

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#ifndef SWEEP_CH
typedef double f64x16 __attribute__((ext_vector_type(16)));
#endif

// out[g] = the value held by lane (lane&15) + 16*g, for g = 0..3, in every lane
__device__ static inline void rows_allgather_u32(unsigned v, unsigned (&out)[4])
{
    const u32x2_t a = __builtin_amdgcn_permlane32_swap(v, v, false, false); // a[0] = rows (0,1,0,1), a[1] = rows (2,3,2,3)
    const u32x2_t b = __builtin_amdgcn_permlane16_swap(a[0], a[0], false, false);
    const u32x2_t c = __builtin_amdgcn_permlane16_swap(a[1], a[1], false, false);
    out[0] = b[0];
    out[1] = b[1];
    out[2] = c[0];
    out[3] = c[1];
}
__device__ static inline void rows_allgather(double v, double (&out)[4])
{
    const uint2 p = __builtin_bit_cast(uint2, v);
    unsigned lo[4], hi[4];
    rows_allgather_u32(p.x, lo);
    rows_allgather_u32(p.y, hi);
#pragma unroll
    for (int g = 0; g < 4; g++) out[g] = __builtin_bit_cast(double, uint2{lo[g], hi[g]});
}

// HAS_MASK = false: no per-entry mask (a.mask == NULL); the only per-column predicate left is `act`, applied once
// per block instead of once per coordinate.
template <int NT, bool HAS_MASK>
__global__ __launch_bounds__(256) void sweep_scd_mfma_kernel(const SweepArgs a)
{
    constexpr int KP = 16 * NT, NB = 4 * NT;
    // Blocks are dealt to the NT accumulator tiles round-robin: block b lives in tile b % NT, accumulator register b / NT,
    // so consecutive blocks sit in different tiles and the next block's chain only waits for ONE MFMA.
    // coordinate of (tile t, accumulator row M) = 4*((M/4)*NT + t) + M%4.
    // Gz[b][t][g][l] = edited G[coord(t, l)][4b + g], with the 4x4 diagonal blocks zeroed  (MFMA A operand of block b, tile t)
    __shared__ __attribute__((aligned(16))) double Gz[NB * NT * 64];
    __shared__ __attribute__((aligned(16))) double G4[NB * 16]; // [b][s'][s] = edited G[4b + s'][4b + s]
    __shared__ __attribute__((aligned(16))) double rG[KP];      // 1 / G[q][q]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int k = a.k;

    auto edited = [&](int c, int kc) -> double { // regularisation edits of src/update_with_missing.cpp:20-24
        double g = a.Graw[(size_t)c * a.KPg + kc];
        if (c == kc && a.r0 != a.r1) g += a.r0 - a.r1;
        if (a.r1 != 0) g += a.r1;
        if (c == kc) g += NNLM_TINY;
        return g;
    };
    for (int e = tid; e < NB * NT * 64; e += 256) {
        const int b = e / (NT * 64), rem = e % (NT * 64), t = rem / 64, g = (rem % 64) / 16, l = rem % 16;
        const int c = 4 * ((l >> 2) * NT + t) + (l & 3), kc = 4 * b + g;
        Gz[e] = (c < k && kc < k && (c >> 2) != b) ? edited(c, kc) : 0.0;
    }
    for (int e = tid; e < NB * 16; e += 256) {
        const int b = e / 16, c = 4 * b + (e % 16) / 4, kc = 4 * b + (e % 4);
        G4[e] = (c < k && kc < k) ? edited(c, kc) : ((c == kc) ? 1.0 : 0.0); // padding: inert, finite reciprocal
    }
    for (int q = tid; q < KP; q += 256) rG[q] = 1.0 / ((q < k) ? edited(q, q) : 1.0);
    __syncthreads();

    const int col = a.col0 + (blockIdx.x * 4 + wave) * 16 + l15;
    const bool in_range = col < a.ncols;
    const int cc = in_range ? col : 0;
    unsigned long long mword = 0ull;
    if (a.mask) mword = a.mask[cc];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    bool act = in_range && !(a.mask && ((mword & kmask) == kmask)); // arma::all(mask.col(j)) -> column skipped

    // element e = 4t + r of this lane's vectors <-> block b = r*NT + t <-> coordinate 4b + lg; MFMA tile t = elements
    // 4t..4t+3 (f64 accumulator layout: reg r of tile t is row lg + 4r).  One 16-element vector per quantity so that the
    // wave-uniform register index r can address it (s_set_gpr_idx).
    f64x16 x, mu;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int b = (e & 3) * NT + (e >> 2); // meaningful for e < 4*NT
        const int q = 4 * b + lg;
        double xv = 0.0, cv = 0.0;
        if (e < NB && q < k) {
            xv = a.X[(size_t)q * a.ldx + cc];
            for (int s = 0; s < a.nslabs; s++) cv += a.Cx[(size_t)s * a.slab_stride + (size_t)q * a.ldc + cc];
        }
        x[e] = xv;
        mu[e] = (e < NB && q < k) ? ((a.r2 != 0) ? a.r2 - cv : -cv) : 0.0;
    }
    const double *gzl = Gz + lane;                 // + (b*NT + t)*64
    const f64x2 *g4row = (const f64x2 *)(G4 + 4 * lg); // + 8*b : row lg of diagonal block b (this lane's coordinate)
    const int nbk = (k + 3) / 4;                   // blocks that hold real coordinates

// mu[tile t2] += Gz[b][t2] (16 coordinates x 4 block columns) * coef (4 block rows x 16 columns), all tiles
#define SWEEP_MFMA_RANK4(bidx, coef)                                                                                   \
    _Pragma("unroll") for (int t2 = 0; t2 < NT; t2++)                                                                   \
    {                                                                                                                   \
        f64x4 tile = f64x4{mu[4 * t2], mu[4 * t2 + 1], mu[4 * t2 + 2], mu[4 * t2 + 3]};                                  \
        tile = __builtin_amdgcn_mfma_f64_16x16x4f64(gzl[((bidx) * NT + t2) * 64], (coef), tile, 0, 0, 0);                 \
        mu[4 * t2] = tile[0];                                                                                           \
        mu[4 * t2 + 1] = tile[1];                                                                                       \
        mu[4 * t2 + 2] = tile[2];                                                                                       \
        mu[4 * t2 + 3] = tile[3];                                                                                       \
    }

    // mu = (L1 - c) + G x : off-diagonal blocks on the matrix cores, diagonal blocks with 4 FMAs
#pragma nounroll
    for (int r0 = 0; r0 < 4; r0++) {
#pragma unroll
        for (int t0 = 0; t0 < NT; t0++) {
            const int kb = r0 * NT + t0;
            if (kb < nbk) { // wave-uniform
                const double xb = x[4 * t0 + r0];
                SWEEP_MFMA_RANK4(kb, xb)
                double xs[4];
                rows_allgather(xb, xs);
                const f64x2 ga = g4row[8 * kb], gb = g4row[8 * kb + 1];
                double add = mu[4 * t0 + r0];
                add = __builtin_fma(ga[0], xs[0], add);
                add = __builtin_fma(ga[1], xs[1], add);
                add = __builtin_fma(gb[0], xs[2], add);
                add = __builtin_fma(gb[1], xs[3], add);
                mu[4 * t0 + r0] = add;
            }
        }
    }

    int t_lane = 0;
    unsigned t = 0;
    const double tol = a.rel_tol;
    while (t < a.max_iter && __any(act)) {
        int flag = (0.0 > tol) ? 1 : 0; // rel_err starts each sweep at 0: a negative rel_tol never stops
#pragma nounroll
        for (int r0 = 0; r0 < 4; r0++) {
#pragma unroll
          for (int t0 = 0; t0 < NT; t0++) {
            const int b = r0 * NT + t0; // consecutive blocks, consecutive tiles
            if (b >= nbk) continue;     // wave-uniform
            const double mu_own = mu[4 * t0 + r0], x_own = x[4 * t0 + r0];
            double m[4], xs[4];
            rows_allgather(mu_own, m);
            rows_allgather(x_own, xs);
            const f64x2 *g4 = (const f64x2 *)(G4 + 16 * b); // uniform: whole diagonal block, row-major
            const f64x2 g00 = g4[0], g10 = g4[2], g20 = g4[4], g21 = g4[5], g30 = g4[6], g31 = g4[7];
            const f64x2 rg0 = *(const f64x2 *)(rG + 4 * b), rg1 = *(const f64x2 *)(rG + 4 * b + 2);
            const double gd[4] = {g00[0], g10[1], g21[0], g31[1]};
            const double rg[4] = {rg0[0], rg0[1], rg1[0], rg1[1]};
            // column s of the strictly lower triangle: G[4b+s2][4b+s], s2 > s
            const double gl[4][4] = {{0, 0, 0, 0}, {g10[0], 0, 0, 0}, {g20[0], g20[1], 0, 0}, {g30[0], g30[1], g31[0], 0}};
            double dd[4], xn[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double q0 = m[s] * rg[s];
                const double rr = __builtin_fma(-q0, gd[s], m[s]);
                const double quo = __builtin_fma(rr, rg[s], q0); // = mu / G[q][q], correctly rounded
                const double tmp = fmax(xs[s] - quo, 0.0);
                if (HAS_MASK) {
                    const int q = 4 * b + s;
                    const bool free_q = !((mword >> q) & 1ull);
                    dd[s] = free_q ? tmp - xs[s] : 0.0;
                    xn[s] = free_q ? tmp : xs[s];
                } else { // padded coordinates (q >= k) are inert by construction: x = mu = 0, G = identity there
                    dd[s] = tmp - xs[s];
                    xn[s] = tmp;
                }
#pragma unroll
                for (int s2 = s + 1; s2 < 4; s2++) m[s2] = __builtin_fma(dd[s], gl[s2][s], m[s2]);
            }
            // this lane's coordinate of the block is 4b + lg; converged / out-of-range columns keep their values
            double x_new = (lg == 0) ? xn[0] : (lg == 1) ? xn[1] : (lg == 2) ? xn[2] : xn[3];
            x_new = act ? x_new : x_own;
            const double d_own = x_new - x_own; // == dd[lg] (tmp - x), exactly 0 when nothing moved or the column is idle
            // every other coordinate: rank-4 update on the matrix cores (rows of this block are zero in Gz); the tile that
            // holds the next block's coordinate goes first
#pragma unroll
            for (int u = 0; u < NT; u++) {
                const int t2 = (t0 + 1 + u) % NT; // the tile of the NEXT block first: its chain waits for one MFMA only
                f64x4 tile = f64x4{mu[4 * t2], mu[4 * t2 + 1], mu[4 * t2 + 2], mu[4 * t2 + 3]};
                tile = __builtin_amdgcn_mfma_f64_16x16x4f64(gzl[(b * NT + t2) * 64], d_own, tile, 0, 0, 0);
                mu[4 * t2] = tile[0];
                mu[4 * t2 + 1] = tile[1];
                mu[4 * t2 + 2] = tile[2];
                mu[4 * t2 + 3] = tile[3];
            }
            // rel-change test (src/base_algorithms.cpp:29-32), each row tests its own coordinate; OR-ed after the sweep
            flag |= ((2 * fabs(d_own)) > tol * (x_new + x_own + NNLM_TINY)) ? 1 : 0;
            // own coordinate: the four updates of this block in order (idle columns: all four deltas are masked to 0)
            const f64x2 ga = g4row[8 * b], gb = g4row[8 * b + 1];
            double mo = mu_own;
            mo = __builtin_fma(dd[0], ga[0], mo);
            mo = __builtin_fma(dd[1], ga[1], mo);
            mo = __builtin_fma(dd[2], gb[0], mo);
            mo = __builtin_fma(dd[3], gb[1], mo);
            mo = act ? mo : mu_own;
            mu[4 * t0 + r0] = mo;
            x[4 * t0 + r0] = x_new;
          }
        }
        unsigned fl[4];
        rows_allgather_u32((unsigned)flag, fl);
        flag = (int)(fl[0] | fl[1] | fl[2] | fl[3]);
        if (act) {
            t_lane++;
            act = flag != 0;
        }
        t++;
    }
#undef SWEEP_MFMA_RANK4

    if (in_range) {
#pragma unroll
        for (int e = 0; e < NB; e++) {
            {
                const int q = 4 * ((e & 3) * NT + (e >> 2)) + lg;
                if (q < k) {
                    const double xv = x[e];
                    a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xv;
                    if (a.op_mode == 1) {
                        if (a.op_f64) ((double *)a.op)[(size_t)q * a.op_ld + col] = xv;
                        else ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xv;
                    } else if (a.op_mode == 2) {
                        if (a.op_f64) ((double *)a.op)[(size_t)col * a.op_ld + q] = xv;
                        else ((float *)a.op)[(size_t)col * a.op_ld + q] = (float)xv;
                    }
                }
            }
        }
    }
    long long tot = wave_sum_ll((lg == 0) ? (long long)t_lane : 0ll);
    if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
}
