// k_sweep.h -- per-column solvers for the square-loss methods, fp64.
//
// Reference: the body of update()'s column loop (src/update_with_missing.cpp:29-53) followed by
//   method 1: scd_ls_update (src/base_algorithms.cpp:3-37)
//   method 2: lee_ls_update (src/base_algorithms.cpp:40-68)
//
// The coordinate loop inside one column is loop-carried (Gauss-Seidel) and is NOT parallelised: 2500 dependent
// coordinate steps per column per half-step at config 2.  Parallelism is over columns.  A column is owned by L
// adjacent lanes (L = 1, 2 or 4; 64/L columns per wavefront); coordinate q = L*r + s lives in register r of
// sub-lane s, so the k-long gradient update  mu += d * G[:,q]  costs R = ceil(k/L) FMAs per lane and every
// instruction works on 64/L columns at once.  L trades FMA work per step against the number of wavefronts
// (columns*L/64): the host picks the largest L that still leaves at most one wavefront per SIMD.
//   * x and mu are NCH vectors of CH doubles per lane; the chunk index is unrolled, the element index is a real
//     loop whose wave-uniform counter addresses the registers through s_set_gpr_idx (no scratch, small code);
//   * the delta of the coordinate's owner reaches its L-1 neighbours with a DPP quad_perm move;
//   * G (shared by all columns, with the regularisation edits of src/update_with_missing.cpp:20-24) sits in LDS
//     permuted as Gp[q][s][r] so that each lane reads ITS R entries of row q as contiguous 16-byte words
//     (the 64/L lanes with equal s read the same address -> broadcast, L distinct addresses, no bank conflict);
//   * x - mu/G[q][q] uses the reciprocal of the diagonal (computed once per half-step with a true division)
//     followed by one Newton/Markstein correction, which returns the correctly rounded quotient;
//   * the test  2|d|/(tmp+x+eps) > rel_tol  is done without a division (2|d| > tol*(tmp+x+eps)); it can decide
//     differently from the rounded quotient only when both sides agree to ~2 ulp;
//   * row q+1 of G is fetched from LDS while coordinate q is processed (two register buffers).
//
// Prologue (per column j): c = sum of the split-K slabs of the cross product (fixed order),
//   method 1: mu = G x - c (+ L1)      (src/update_with_missing.cpp:39-41)
//   method 2: keeps c                   (src/update_with_missing.cpp:45)
// Differences from the reference's arithmetic (all below 1 ulp per operation):
//   mu += d*G[:,q] is one fused multiply-add per entry (the reference rounds the product first).
#pragma once
#include "common.h"

typedef double f64x8 __attribute__((ext_vector_type(8)));

struct SweepArgs {
    const double *X;    // [KPx][ldx] master copy of the factor being solved (row q, column index fastest), read
    double *Xout;       // written: entry (q, col) goes to Xout[q*ldo + (col - ocol0)]  (may alias X; a different buffer
                        // lets a half-step run speculatively; a packed per-rank slab feeds the multi-GPU all-gather)
    int ldx;
    int ldo, ocol0;
    int col0;           // first column this launch solves (columns col0 .. ncols-1)
    const double *Graw; // [KPg][KPg] Gram of the other factor (no edits applied yet)
    int KPg;
    const double *Cx;   // [nslabs][KPg][ldc] split-K partial cross products
    size_t slab_stride;
    int nslabs;
    int ldc;
    int ncols;          // END of the column range to solve (exclusive); all of X has at least this many columns
    int k;              // true rank
    double r0, r1, r2;  // L2, angle, L1 of this half-step (beta in the reference's update())
    const unsigned long long *mask; // [ncols] bit q set <=> entry (q, col) is masked; NULL = no mask
    unsigned max_iter;
    double rel_tol;
    void *op;           // GEMM-operand copy of the updated factor
    int op_mode;        // 0: none, 1: [KP][op_ld] (same layout as X)
    int op_ld;
    int op_f64;         // element type of op: 0 float, 1 double
    unsigned long long *sweeps; // += sum of per-column sweep counts
    // k_sweep_q.h only (NULL elsewhere): what the NEXT half-step needs from the factor solved here, produced from the
    // final x image while it is still in LDS -- its Gram partial sums and max|x| (the scale of its split-fp16 copy)
    unsigned *maxbits = nullptr;      // atomicMax of the float bit pattern of max|x| over the solved columns
    int g_upper = 0;                  // colsolve_*_kernel: the per-column Grams at Graw hold their upper triangle only (na_gram_*_kernel, upper_only)
    double *gram_slabs = nullptr;     // [workgroups][KP*KP]  Gram of each workgroup's 64 (persistent form: 16 G) columns (upper tiles)
};

// The SCD sweep's operand image (k_sweep_q.h), written straight from the fold: SweepImg describes it, sweepq_img_put() stores the
// edited Gram entry E[r][c] (src/update_with_missing.cpp:20-24: + r0 - r1 on the diagonal, + r1 everywhere, + 1e-16 on the diagonal;
// coordinates >= k inert) into the slot(s) of the image that hold it.  Layout: k_sweep_q.h.  img == NULL: no image.
struct SweepImg {
    double *img = nullptr;
    int NB = 0, NP = 0, k = 0;
    double r0 = 0.0, r1 = 0.0;
};
__device__ static inline void sweepq_img_put(const SweepImg &im, int r, int c, double graw)
{
    const int NB = im.NB, NP = im.NP;
    if (r >= 4 * NB || c >= 4 * NB) return;
    double v;
    if (r >= im.k || c >= im.k) v = (r == c) ? 1.0 : 0.0;
    else {
        v = graw;
        if (r == c && im.r0 != im.r1) v += im.r0 - im.r1;
        if (im.r1 != 0) v += im.r1;
        if (r == c) v += NNLM_TINY;
    }
    const int s = r >> 2, iA = r & 3, beta = c >> 2, kA = c & 3;
    im.img[(((size_t)beta * NP + (s >> 1)) * 16 + 4 * kA + iA) * 2 + (s & 1)] = v;         // operand of accumulator s for the deltas of block beta
    if (s == beta) {                                                                       // the block's own 4 x 4 piece: chain operand (strictly lower part)
        const int bprev = (s + NB - 1) % NB;                                               // stored with the block that precedes it
        im.img[(((size_t)bprev * NP + (NB >> 1)) * 16 + 4 * kA + iA) * 2 + (NB & 1)] = (iA > kA) ? v : 0.0;
        if (r == c) im.img[(size_t)NB * NP * 32 + r] = v;                                  // the diagonal, behind the operands
    }
}


typedef double f64x16 __attribute__((ext_vector_type(16)));
#define SWEEP_CH 16 // registers per indexable chunk (32 VGPRs: the largest s_set_gpr_idx-addressable vector)

// quad_perm DPP broadcast of sub-lane S (of each group of L lanes) to the whole group
template <int L, int S> __device__ static inline double dpp_bcast(double v)
{
    if constexpr (L == 1) return v;
    constexpr int ctrl = (L == 4) ? (S | (S << 2) | (S << 4) | (S << 6)) : (S | (S << 2) | ((2 + S) << 4) | ((2 + S) << 6));
    int2 p = __builtin_bit_cast(int2, v);
    p.x = __builtin_amdgcn_update_dpp(p.x, p.x, ctrl, 0xF, 0xF, false);
    p.y = __builtin_amdgcn_update_dpp(p.y, p.y, ctrl, 0xF, 0xF, false);
    return __builtin_bit_cast(double, p);
}
// sum over the L lanes of a group (result in every lane of the group)
template <int L> __device__ static inline double dpp_group_sum(double v)
{
    if constexpr (L >= 2) {
        int2 p = __builtin_bit_cast(int2, v);
        p.x = __builtin_amdgcn_update_dpp(p.x, p.x, 0xB1, 0xF, 0xF, false); // quad_perm [1,0,3,2]
        p.y = __builtin_amdgcn_update_dpp(p.y, p.y, 0xB1, 0xF, 0xF, false);
        v += __builtin_bit_cast(double, p);
    }
    if constexpr (L >= 4) {
        int2 p = __builtin_bit_cast(int2, v);
        p.x = __builtin_amdgcn_update_dpp(p.x, p.x, 0x4E, 0xF, 0xF, false); // quad_perm [2,3,0,1]
        p.y = __builtin_amdgcn_update_dpp(p.y, p.y, 0x4E, 0xF, 0xF, false);
        v += __builtin_bit_cast(double, p);
    }
    return v;
}
template <int L> __device__ static inline int dpp_group_or(int v)
{
    if constexpr (L >= 2) v |= __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);
    if constexpr (L >= 4) v |= __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);
    return v;
}

// LDS image of G for lane groups of L: Gp[(q*L + s)*R + r] = edited G[q][L*r + s]; Gd[q] = {G[q][q], 1/G[q][q]}
template <int R, int L>
__device__ static inline void sweep_load_gram(double *Gp, f64x2 *Gd, const SweepArgs &a, int tid, int nthreads)
{
    constexpr int KPs = R * L;
    const int k = a.k;
    for (int e = tid; e < KPs * KPs; e += nthreads) {
        const int q = e / KPs, rem = e % KPs, s = rem / R, r = rem % R;
        const int c = L * r + s;
        double g = 0.0;
        if (q < k && c < k) {
            g = a.Graw[(size_t)q * a.KPg + c];
            if (q == c && a.r0 != a.r1) g += a.r0 - a.r1; // src/update_with_missing.cpp:20-21
            if (a.r1 != 0) g += a.r1;                      // :22-23
            if (q == c) g += NNLM_TINY;                    // :24
        }
        Gp[e] = g;
    }
    for (int q = tid; q < KPs; q += nthreads) {
        double g = 1.0;
        if (q < k) {
            g = a.Graw[(size_t)q * a.KPg + q];
            if (a.r0 != a.r1) g += a.r0 - a.r1;
            if (a.r1 != 0) g += a.r1;
            g += NNLM_TINY;
        }
        Gd[q] = f64x2{g, 1.0 / g};
    }
}

// (Ablations -- no AXPY FMAs, no LDS traffic in the loop, no rel-change test, chain cut: scripts/exp/csrc_r5/k_sweep.h with
// scripts/exp/sweep_exp.hip.  The product kernel carries no switch.)
template <int R, int L, int METHOD>
__global__ __launch_bounds__(64) void sweep_ls_kernel(const SweepArgs a)
{
    constexpr int CH = SWEEP_CH;
    constexpr int NCH = (R + CH - 1) / CH; // chunks per lane
    constexpr int KPs = R * L;             // padded rank of this instantiation (>= k); R is even
    __shared__ __attribute__((aligned(16))) double Gp[KPs * KPs];
    __shared__ __attribute__((aligned(16))) f64x2 Gd[KPs];
    const int lane = threadIdx.x;
    const int sub = lane % L;
    const int col = a.col0 + blockIdx.x * (64 / L) + lane / L;
    const int k = a.k;
    sweep_load_gram<R, L>(Gp, Gd, a, lane, 64);
    __syncthreads();

    const bool in_range = col < a.ncols;
    const int cc = in_range ? col : 0;
    unsigned long long mword = 0ull;
    if (a.mask) mword = a.mask[cc];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    // skip columns whose coordinates are all masked (arma::all(mask.col(j)), src/update_with_missing.cpp:33)
    bool act = in_range && !(a.mask && ((mword & kmask) == kmask));

    f64x16 x[NCH], v[NCH]; // v = mu (method 1) or c = Y*b (method 2); register r = CH*c + e holds coordinate L*r + sub
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int e = 0; e < CH; e++) {
            const int r = CH * c + e;
            const int q = L * r + sub;
            double xv = 0.0, cv = 0.0;
            if (r < R && q < k) {
                xv = a.X[(size_t)q * a.ldx + cc];
                for (int s = 0; s < a.nslabs; s++) cv += a.Cx[(size_t)s * a.slab_stride + (size_t)q * a.ldc + cc];
            }
            x[c][e] = xv;
            v[c][e] = (METHOD == 1) ? ((r < R && q < k) ? ((a.r2 != 0) ? a.r2 - cv : -cv) : 0.0) : cv;
        }
    const f64x2 *grow = (const f64x2 *)(Gp + (size_t)sub * R); // + q*L*R/2 selects row q for this sub-lane

// v[:] += (coef) * G[q][own coordinates], R FMAs with static register indices
#define SWEEP_AXPY(coef, gq)                                                              \
    _Pragma("unroll") for (int r2_ = 0; r2_ < R; r2_ += 2)                                 \
    {                                                                                     \
        const f64x2 g2_ = (gq)[r2_ / 2];                                                  \
        v[r2_ / CH][r2_ % CH] = __builtin_fma((coef), g2_[0], v[r2_ / CH][r2_ % CH]);      \
        v[(r2_ + 1) / CH][(r2_ + 1) % CH] = __builtin_fma((coef), g2_[1], v[(r2_ + 1) / CH][(r2_ + 1) % CH]); \
    }

    if (METHOD == 1) { // mu = (L1 - c) + sum_q x[q] * G[q][:]   (rank-1 accumulation, q ascending)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int eend = (R - CH * c) < CH ? (R - CH * c) : CH;
#pragma nounroll
            for (int e = 0; e < eend; e++) {
                const double xown = x[c][e];
#pragma unroll
                for (int s = 0; s < L; s++) {
                    const int q = L * (CH * c + e) + s;
                    const double xq = (s == 0) ? dpp_bcast<L, 0>(xown) : (s == 1) ? dpp_bcast<L, (L > 1 ? 1 : 0)>(xown)
                                      : (s == 2) ? dpp_bcast<L, (L > 2 ? 2 : 0)>(xown) : dpp_bcast<L, (L > 3 ? 3 : 0)>(xown);
                    const f64x2 *gq = grow + (size_t)q * (L * R / 2);
                    SWEEP_AXPY(xq, gq)
                }
            }
        }
    }

    int t_lane = 0;
    unsigned t = 0;
    const double tol = a.rel_tol;

    // Row q of G for this sub-lane, fetched one coordinate ahead of its use (LDS latency off the dependent chain):
    // R/2 16-byte words of the row, the {G[q][q], 1/G[q][q]} pair and G[q][this lane's coordinate of the block].
    struct GRow {
        f64x2 g[R / 2];
        f64x2 gd;
    };
    auto fetch = [&](GRow &o, int q) {
        const f64x2 *gq = grow + (size_t)q * (L * R / 2);
#pragma unroll
        for (int i = 0; i < R / 2; i++) o.g[i] = gq[i];
        o.gd = Gd[q];
    };
    GRow rowA, rowB;
    if (METHOD == 1) fetch(rowA, 0);

    while (t < a.max_iter && __any(act)) {
        int flag = (METHOD == 1 && 0.0 > tol) ? 1 : 0; // rel_err starts each sweep at 0 (src/base_algorithms.cpp:20):
                                                       // a negative rel_tol never stops
        double relmax = 0.0; // method 2 only
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int eend = (R - CH * c) < CH ? (R - CH * c) : CH;
#pragma nounroll
            for (int e = 0; e < eend; e++) {
                if (L * (CH * c + e) >= k) break; // wave-uniform
                const double xown = x[c][e], vown = v[c][e];
                double xnew = xown, vcur = vown;
                const double *gown = Gp + (size_t)sub * R + (CH * c + e); // + q*L*R: G[q][this lane's coordinate]
#pragma unroll
                for (int s = 0; s < L; s++) {
                    // coordinates q >= k of the last block are inert: x = mu = 0, G row/column 0, Gd = {1, 1}
                    const int q = L * (CH * c + e) + s;
                    const bool free_q = act && !((mword >> q) & 1ull) && (q < k);
                    const bool owner = (sub == s);
                    if (METHOD == 1) {
                        // L is even or 1: the two row buffers alternate with the parity of s (static)
                        GRow &cur = ((s & 1) == 0 || L == 1) ? rowA : rowB;
                        GRow &nxt = ((s & 1) == 0 || L == 1) ? rowB : rowA;
                        const double gownq = gown[(size_t)q * (L * R)];
                        int qn = q + 1;
                        qn = (qn >= k) ? 0 : qn; // after the last coordinate the next sweep starts at 0
                        if (L > 1) fetch(nxt, qn);
                        // vcur = this lane's mu for ITS coordinate of the block.  Every lane runs the chain on its own values;
                        // the DPP broadcast below picks the owner sub-lane's result, so no owner test sits on the chain, and
                        // tmp - x is exactly 0 when nothing changes, so neither does the reference's `tmp != Hj(k)` test.
                        const double q0 = vcur * cur.gd[1];
                        const double rr = __builtin_fma(-q0, cur.gd[0], vcur);
                        const double quo = __builtin_fma(rr, cur.gd[1], q0); // = mu / G[q][q], correctly rounded
                        const double tmp = fmax(xown - quo, 0.0);           // NaN -> 0 (the reference would keep the NaN)
                        const double dd = free_q ? tmp - xown : 0.0;         // free_q does not depend on the chain
                        const double d = (s == 0) ? dpp_bcast<L, 0>(dd) : (s == 1) ? dpp_bcast<L, (L > 1 ? 1 : 0)>(dd)
                                         : (s == 2) ? dpp_bcast<L, (L > 2 ? 2 : 0)>(dd) : dpp_bcast<L, (L > 3 ? 3 : 0)>(dd);
                        if (L > 1) vcur = __builtin_fma(d, gownq, vcur); // keeps the chain out of the indexed registers
#pragma unroll
                        for (int r2 = 0; r2 < R; r2 += 2) {
                            v[r2 / CH][r2 % CH] = __builtin_fma(d, cur.g[r2 / 2][0], v[r2 / CH][r2 % CH]);
                            v[(r2 + 1) / CH][(r2 + 1) % CH] = __builtin_fma(d, cur.g[r2 / 2][1], v[(r2 + 1) / CH][(r2 + 1) % CH]);
                        }
                        // rel-change test of src/base_algorithms.cpp:29-32 without the division:
                        //   2|d| / (tmp + x + eps) > tol   <=>   2|d| > tol * (tmp + x + eps)
                        // (the rounded quotient's decision: the division is formed only where the two sides agree to ~2 ulp, common.h)
                        const bool big = rel_change_exceeds(2 * fabs(dd), tmp + xown + NNLM_TINY, tol);
                        flag |= (owner && big) ? 1 : 0;
                        xnew = (free_q && owner) ? tmp : xnew;
                        if (L == 1) fetch(rowA, qn);
                    } else {
                        const f64x2 *gq = grow + (size_t)q * (L * R / 2);
                        double part = 0.0, part2 = 0.0;
#pragma unroll
                        for (int r2 = 0; r2 < R; r2 += 2) {
                            const f64x2 g2 = gq[r2 / 2];
                            part = __builtin_fma(g2[0], x[r2 / CH][r2 % CH], part);
                            part2 = __builtin_fma(g2[1], x[(r2 + 1) / CH][(r2 + 1) % CH], part2);
                        }
                        const double dot = dpp_group_sum<L>(part + part2);
                        double tmp = dot + a.r2;
                        tmp = vown / (tmp + NNLM_TINY);
                        const bool app = free_q && owner;
                        const double er = 2 * fabs(tmp - 1) / (tmp + 1);
                        relmax = (app && er > relmax) ? er : relmax;
                        xnew = app ? xown * tmp : xnew;
                        if (L > 1 && s + 1 < L) x[c][e] = xnew; // later sub-lanes' dots must see this coordinate's new value
                    }
                }
                x[c][e] = xnew;
            }
        }
        if (METHOD == 2) flag = relmax > tol;
        flag = dpp_group_or<L>(flag);
        if (act) {
            t_lane++;
            act = flag != 0;
        }
        t++;
    }
#undef SWEEP_AXPY

    if (in_range) {
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int e = 0; e < CH; e++) {
                const int r = CH * c + e;
                const int q = L * r + sub;
                if (r < R && q < k) {
                    const double xv = x[c][e];
                    a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xv;
                    if (a.op_mode == 1) {
                        if (a.op_f64) ((double *)a.op)[(size_t)q * a.op_ld + col] = xv;
                        else ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xv;
                    }
                }
            }
    }
    long long tot = wave_sum_ll((sub == 0) ? (long long)t_lane : 0ll);
    if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
}
