// k_missing.h -- square-loss half-step when A has missing entries.
//
// Reference: update_with_missing(), src/update_with_missing.cpp:58-139, methods 1 and 2.  Per column j the
// reference restricts the contraction to non_missing = find_finite(A.col(j)) (:80-83) and forms a PER-COLUMN
// Gram  WtW_j = Wt[:,nm] Wt[:,nm]^T  and cross product  Wt[:,nm] A[nm,j]  (:90-91), then applies the same
// regularisation edits (:98-103) and the same per-column solvers.
//
// Here:
//   * the cross product needs nothing new: the resident A holds 0 at missing positions, so the dense
//     A-streaming kernels (k_xprod.h) already sum over finite rows only;
//   * the per-column Gram G_j is formed on the matrix cores, one wavefront per column, as either the direct sum over the
//     finite rows or  G_full - sum over missing rows  (complement), whichever touches fewer rows, from row lists made once per
//     matrix: na_gram_f16_kernel (fp32-operand mode: split-fp16 rows, v_mfma_f32_16x16x32_f16) / na_gram_lds_kernel<double>
//     (strict mode: fp64 rows gathered by LDS-DMA, v_mfma_f64_16x16x4_f64);
//   * solves with that column's own G: colsolve_row_kernel (k_colsolve_row.h: SCD of the fp32-operand mode, four columns per wavefront,
//     rows of G divided by their diagonal); one wavefront per column, lane = coordinate: colsolve_strict_kernel (SCD in the reference's
//     arithmetic), colsolve_ls_kernel
//     (Lee's multiplicative updates: lane r keeps column r of G_j in VGPRs, G[q][r] is an indirect VGPR read, x[q] a v_readlane).
//
// The missing-entry index sets are 1-bit-per-entry masks built by the prep pass (exact, integer):
//   miss  [mpad][npad/32]  bit (i%32) of word [j][i/32]   -- used by the H half-step (column j of A)
//   missT [npad][mpad/32]  bit (j%32) of word [i][j/32]   -- used by the W half-step (row i of A)
#pragma once
#include "common.h"
#include "k_sweep.h"

// missT[i][j/32] bit j%32 = miss[j][i/32] bit i%32.  One thread per output word.
__global__ __launch_bounds__(256) void miss_transpose_kernel(const uint32_t *__restrict__ miss, int npad, int mpad,
                                                             uint32_t *__restrict__ missT)
{
    const int wj = blockIdx.x * 256 + threadIdx.x; // word index along j
    const int i = blockIdx.y;
    const int words_i = npad >> 5, words_j = mpad >> 5;
    if (wj >= words_j) return;
    uint32_t out = 0;
    for (int b = 0; b < 32; b++) {
        const int j = wj * 32 + b;
        out |= ((miss[(size_t)j * words_i + (i >> 5)] >> (i & 31)) & 1u) << b;
    }
    missT[(size_t)i * words_j + wj] = out;
}

// Yrow[c][q] = Y[q][c] for q < KP, so that a row of the fixed factor is one contiguous KP-element read (T = double, or
// float for the fp32-operand mode's per-column Grams).
template <typename T = double>
__global__ __launch_bounds__(256) void factor_rows_kernel(const double *__restrict__ Y, int ld, int ncols, int KP, T *__restrict__ Yrow)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    for (int q = 0; q < KP; q++) Yrow[(size_t)c * KP + q] = (T)Y[(size_t)q * ld + c];
}

// One wavefront per column, lane = coordinate (k <= 64).  a.Graw is either one shared Gram (g_stride = 0) or the
// per-column Grams (g_stride = KPg*KPg).  Same arithmetic as sweep_ls_kernel (k_sweep.h).
// value of lane `src` (wave-uniform) in every lane: two v_readlane_b32 instead of a cross-lane LDS permute
__device__ static inline double readlane_f64(double v, int src)
{
    int2 p = __builtin_bit_cast(int2, v);
    p.x = __builtin_amdgcn_readlane(p.x, src);
    p.y = __builtin_amdgcn_readlane(p.y, src);
    return __builtin_bit_cast(double, p);
}

template <int NKQ, int METHOD>
__global__ __launch_bounds__(256) void colsolve_ls_kernel(const SweepArgs a, size_t g_stride)
{
    constexpr int NCH = 2 * NKQ; // chunks of 8 coordinates
    const int lane = threadIdx.x & 63;
    const int col = a.col0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= a.ncols) return; // whole wave
    const int k = a.k;
    const bool lv = lane < k;
    const int lq = lv ? lane : 0;
    const double *G = a.Graw + (size_t)col * g_stride;

    unsigned long long mword = 0ull;
    if (a.mask) mword = a.mask[col];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    const bool skip = a.mask && ((mword & kmask) == kmask); // arma::all(mask.col(j)), :75-76

    // column `lane` of the edited G (= row `lane`, G is symmetric): g[q] = G[q][lane], as NCH vectors of 8 so that the
    // wave-uniform index q can address it with s_set_gpr_idx
    f64x8 g[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int q = 8 * c + e;
            double v = 0.0;
            if (q < k && lv) {
                v = G[(a.g_upper && q > lane) ? (size_t)lane * a.KPg + q : (size_t)q * a.KPg + lane];
                if (q == lane && a.r0 != a.r1) v += a.r0 - a.r1; // :98-99
                if (a.r1 != 0) v += a.r1;                          // :100-101
                if (q == lane) v += NNLM_TINY;                     // :103
            }
            g[c][e] = v;
        }
    double gd = 1.0; // edited G[lane][lane]
    if (lv) {
        gd = G[(size_t)lq * a.KPg + lq];
        if (a.r0 != a.r1) gd += a.r0 - a.r1;
        if (a.r1 != 0) gd += a.r1;
        gd += NNLM_TINY;
    }
    const double rgd = 1.0 / gd; // x - mu/G[q][q] through the reciprocal + one Markstein correction (correctly rounded, k_sweep.h)
    double x = lv ? a.X[(size_t)lq * a.ldx + col] : 0.0;
    double cv = 0.0;
    if (lv)
        for (int s = 0; s < a.nslabs; s++) cv += a.Cx[(size_t)s * a.slab_stride + (size_t)lq * a.ldc + col];
    double v;
    if (METHOD == 1) { // mu = G x - c (+ L1): lane r accumulates sum_q G[r][q] x[q] = sum_q g[q] * x_q
        double s0 = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int qend = (k - 8 * c) < 8 ? (k - 8 * c) : 8;
            for (int e = 0; e < qend; e++) s0 = __builtin_fma(g[c][e], __shfl(x, 8 * c + e, 64), s0);
        }
        v = s0 - cv;
        if (a.r2 != 0) v += a.r2;
        if (!lv) v = 0.0;
    } else
        v = cv;

    int t = 0;
    if (!skip) {
        double rel = 1.0 + a.rel_tol;
        for (; (unsigned)t < a.max_iter && rel > a.rel_tol; t++) {
            rel = 0.0;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int qend = (k - 8 * c) < 8 ? (k - 8 * c) : 8;
                for (int e = 0; e < qend; e++) {
                    const int q = 8 * c + e;
                    if ((mword >> q) & 1ull) continue; // wave uniform
                    if (METHOD == 1) {
                        const double xq = readlane_f64(x, q), muq = readlane_f64(v, q), gqq = readlane_f64(gd, q), rq = readlane_f64(rgd, q);
                        const double q0 = muq * rq;
                        const double rr = __builtin_fma(-q0, gqq, muq);
                        const double quo = __builtin_fma(rr, rq, q0); // = mu / G[q][q], correctly rounded
                        const double tmp = fmax(xq - quo, 0.0);
                        if (tmp != xq) { // uniform
                            const double d = tmp - xq;
                            v = __builtin_fma(d, g[c][e], v);
                            // rel only matters through "rel > rel_tol": the quotient's decision, division only at the boundary (common.h)
                            if (rel_change_exceeds(2 * fabs(d), tmp + xq + NNLM_TINY, a.rel_tol)) rel = 1.0 + fabs(a.rel_tol);
                            if (lane == q) x = tmp;
                        }
                    } else {
                        const double xq = __shfl(x, q, 64);
                        (void)xq;
                        const double dot = wave_sum(lv ? g[c][e] * x : 0.0);
                        double tmp = dot + a.r2;
                        tmp = __shfl(v, q, 64) / (tmp + NNLM_TINY);
                        if (lane == q) x *= tmp;
                        const double er = 2 * fabs(tmp - 1) / (tmp + 1);
                        if (er > rel) rel = er;
                    }
                }
            }
        }
    }
    if (lv) {
        a.Xout[(size_t)lane * a.ldo + (col - a.ocol0)] = x;
        if (a.op_mode == 1) {
            if (a.op_f64) ((double *)a.op)[(size_t)lane * a.op_ld + col] = x;
            else ((float *)a.op)[(size_t)lane * a.op_ld + col] = (float)x;
        }
    }
    if (lane == 0 && t) atomicAdd(a.sweeps, (unsigned long long)t);
}

// ------------------------------------------------------------------------------------------------------------------
// Per-column Grams over PRECOMPUTED row lists.
// The missing pattern of A does not change between iterations, so the rows a column's Gram sums over (the missing rows
// when at most half are missing -- G_j = G_full - sum, otherwise the present rows) are compacted once per matrix into
// CSR lists (na_count_kernel / na_fill_kernel); the Gram kernels below then run one wavefront per column over its list:
// one MFMA per upper tile pair accumulates  sum_r y_r y_r^T  (the same register is the A operand of tile a and the B
// operand of tile b).
__global__ __launch_bounds__(256) void na_count_kernel(const uint32_t *__restrict__ bits_all, int words, int p, int ncols, uint32_t *__restrict__ cnt)
{
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (col >= ncols) return;
    const uint32_t *bits = bits_all + (size_t)col * words;
    int c = 0;
    for (int w = lane; w < (p + 31) / 32; w += 64) {
        uint32_t v = bits[w];
        if ((w + 1) * 32 > p) v &= (p & 31) ? ((1u << (p & 31)) - 1u) : 0xFFFFFFFFu;
        c += __popc(v);
    }
    c = (int)wave_sum_ll(c);
    if (lane == 0) cnt[col] = (uint32_t)c;
}

// idx[ptr[col] .. ptr[col] + len) = the listed rows of column col in increasing order; meta[col] = len | complement << 31
__global__ __launch_bounds__(256) void na_fill_kernel(const uint32_t *__restrict__ bits_all, int words, int p, const uint32_t *__restrict__ ptr,
                                                      const uint32_t *__restrict__ meta, int *__restrict__ idx)
{
    __shared__ int wcnt[4];
    __shared__ int base_s;
    const int col = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *bits = bits_all + (size_t)col * words;
    const uint32_t want = (meta[col] >> 31) ? 1u : 0u; // complement: list the missing rows
    int *out = idx + ptr[col];
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < p; b0 += 256) {
        const int i = b0 + tid;
        bool take = false;
        if (i < p) take = ((bits[i >> 5] >> (i & 31)) & 1u) == want;
        const unsigned long long bal = __ballot(take);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; w++) off += wcnt[w];
        if (take) out[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) base_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
}

// Per-column Gram of the strict mode (instantiated with T = double) with the listed rows gathered by
// LDS-DMA (global_load_lds_dwordx4): a group of four rows is ONE instruction per wavefront (two for fp64 rows of 64) (lane = 16-byte chunk: row lane / (KP/4), chunk lane % (KP/4)) that lands in one of four stage buffers
// of the wavefront and occupies no VGPR.  Three groups are in flight while one is multiplied (s_waitcnt vmcnt(3), written by hand:
// the instruction is issued as inline asm, so the compiler's wait-count pass neither sees it nor serialises the LDS reads behind
// it).  Register-destination gathers leave the depth of the pipeline to the register
// allocator: a rotated copy or a branch around a gather ends in "wait for everything".  The four row indices of a group are
// scalar loads (lgkmcnt), two groups ahead.  Operands come from LDS in the natural layout (tile t, position l15 = coordinate
// 16 t + l15).
// TAIL: k = 16 NT + 1 or 16 NT + 2 (the benchmark's k = 50 = 48 + 2).  The padded form spends 4 of its 10 tile products (NKQ = 4) on
// the tile that holds the two coordinates beyond 48, and its MFMA pipe was busy 46 % of the kernel (PMC,
// profiles/r02_v8_cfg5_pmc_summary.txt).  Here the matrix cores get the NT full tiles only (6 instead of 10 products per four
// rows) and the two tail coordinates are plain FMAs on the lanes that already hold the row: ta[t] += y[c0] y[q], tb[t] += y[c1] y[q]
// for the lane's NT coordinates q, + the 2 x 2 corner -- 2 NT + 3 fp32 FMAs per four rows, issued under the MFMAs.  A lane sums
// ITS row of each group of four; the four lane groups are added at the end.  The fp64 images of the tail sums are touched once
// per 256 rows and live in LDS.  Entries beyond k are never read by the solvers and are not written.
__device__ static inline float na_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ static inline double na_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T, int NT, bool TAIL>
__global__ __launch_bounds__(256) void na_gram_lds_kernel(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ meta, const int *__restrict__ idx,
                                                          const T *__restrict__ Yrow, const double *__restrict__ Gfull, double *__restrict__ Gcols,
                                                          int ncols, int col0, int k, int upper_only)
{
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int KP = 16 * (NT + (TAIL ? 1 : 0));
    constexpr int NP = NT * (NT + 1) / 2;
    constexpr int C0 = 16 * NT, C1 = 16 * NT + 1;
    constexpr int CH = KP * (int)sizeof(T) / 16;  // 16-byte chunks per row
    constexpr int NI = (4 * CH + 63) / 64;        // gather instructions per group of four rows (1 for fp32 rows, 2 for fp64 rows of 64)
    constexpr int STG = 4;                        // stage buffers per wavefront
    constexpr int SF = 4 * KP;                    // elements per stage: four rows
    constexpr int NTL = TAIL ? 2 * NT + 3 : 1;
    __shared__ __attribute__((aligned(16))) T stage_all[4][STG][SF];
    __shared__ double tail64[F32 ? 4 : 1][NTL][64]; // (fp32 rows: fp64 images of the tail sums)
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4, wave = threadIdx.x >> 6;
    const int col = col0 + blockIdx.x * 4 + wave;
    if (col >= ncols) return; // whole wave
    const uint32_t mt = meta[col];
    const int ulen = __builtin_amdgcn_readfirstlane((int)(mt & 0x7FFFFFFFu));
    const bool complement = (mt >> 31) != 0;
    const int base = __builtin_amdgcn_readfirstlane((int)ptr[col]);
    T *stage = &stage_all[wave][0][0];
    double(*t64)[64] = tail64[F32 ? wave : 0];

    acc_t acc[NP];
    f64x4 acc64[F32 ? NP : 1];
    T ta[NT], tb[NT], tt[3] = {(T)0, (T)0, (T)0};
#pragma unroll
    for (int i = 0; i < NP; i++) acc[i] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < (F32 ? NP : 1); i++) acc64[i] = f64x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; t++) ta[t] = tb[t] = (T)0;
    if (TAIL && F32) {
#pragma unroll
        for (int e = 0; e < NTL; e++) t64[e][lane] = 0.0;
    }

    struct Idx4 {
        int a, b, c, d;
    };
    const int ngt = (ulen + 3) >> 2, ng = ulen >> 2; // groups, full groups
    auto load_idx = [&](int g, Idx4 &ri) {            // (the list array carries slack behind its end)
        const int *src = idx + base + 4 * g;
        ri.a = src[0], ri.b = src[1], ri.c = src[2], ri.d = src[3];
    };
    const unsigned long long yp = (unsigned long long)Yrow;
    const unsigned long long ybase = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(yp >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)yp);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) T *)stage);
    // DMA role of the lane in instruction ii: chunk 64 ii + lane = row c / CH, chunk c % CH (loop-invariant lane constants; lane masks,
    // not a ?: chain over the indices: that becomes a scratch array here)
    int drow_[NI], m1_[NI], m2_[NI], m3_[NI];
    unsigned choff_[NI];
    bool dact_[NI];
#pragma unroll
    for (int ii = 0; ii < NI; ii++) {
        const int c = 64 * ii + lane;
        drow_[ii] = c / CH;
        choff_[ii] = (unsigned)(c % CH) * 16u;
        dact_[ii] = c < 4 * CH;
        m1_[ii] = (drow_[ii] == 1) ? -1 : 0, m2_[ii] = (drow_[ii] == 2) ? -1 : 0, m3_[ii] = (drow_[ii] == 3) ? -1 : 0;
    }
    auto issue = [&](int g, const Idx4 &ri) { // group g (wave-uniform) into stage g % STG; rows past the end of the list repeat the group's first
#pragma unroll
        for (int ii = 0; ii < NI; ii++) {
            int row = ri.a ^ ((ri.a ^ ri.b) & m1_[ii]) ^ ((ri.a ^ ri.c) & m2_[ii]) ^ ((ri.a ^ ri.d) & m3_[ii]);
            if (4 * g + drow_[ii] >= ulen) row = ri.a;
            const unsigned voff = (unsigned)row * (unsigned)(KP * sizeof(T)) + choff_[ii];
            const unsigned dst = lds0 + (unsigned)(g & (STG - 1)) * (unsigned)(SF * sizeof(T)) + (unsigned)ii * 1024u;
            if (dact_[ii]) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ybase), "s"(dst) : "memory");
        }
    };
    struct Row {
        T x[NT];
        T tl[2];
    };
    auto fetch = [&](int g, Row &r) { // operands of group g from its stage buffer
        const T *sb = stage + (g & (STG - 1)) * SF + lg * KP;
#pragma unroll
        for (int t = 0; t < NT; t++) r.x[t] = sb[16 * t + l15];
        if (TAIL) {
            r.tl[0] = sb[C0];
            r.tl[1] = sb[C1];
        } else {
            r.tl[0] = r.tl[1] = (T)0;
        }
    };
    auto mult = [&](const Row &x) {
        int pi = 0;
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int b = a; b < NT; b++, pi++) acc[pi] = M::mma(x.x[a], x.x[b], acc[pi]);
        if (TAIL) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                ta[t] = na_fma(x.tl[0], x.x[t], ta[t]);
                tb[t] = na_fma(x.tl[1], x.x[t], tb[t]);
            }
            tt[0] = na_fma(x.tl[0], x.tl[0], tt[0]);
            tt[1] = na_fma(x.tl[0], x.tl[1], tt[1]);
            tt[2] = na_fma(x.tl[1], x.tl[1], tt[2]);
        }
    };
    if (ngt > 0) {
        // prologue: groups 0, 1, 2 (clamped) in flight, indices of group 3 requested
        Idx4 ri;
        const int last = ngt - 1;
        load_idx(0, ri);
        issue(0, ri);
        load_idx(last < 1 ? last : 1, ri);
        issue(last < 1 ? last : 1, ri); // (a clamped group lands in the stage of its own number: never the one being read, see below)
        load_idx(last < 2 ? last : 2, ri);
        issue(last < 2 ? last : 2, ri);
        load_idx(last < 3 ? last : 3, ri);
        int since = 0;
        for (int g = 0; g < ng; g++) {
            // group g + 3 goes into the stage group g - 1 was read from (its operands were in registers before its MFMAs issued).
            // Past the end the LAST group is fetched again, into its own stage and with its own data: harmless whenever it lands.
            const int gi = (g + 3 < last) ? g + 3 : last;
            issue(gi, ri);
            load_idx((g + 4 < last) ? g + 4 : last, ri);
            if (NI == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); // groups g + 1 .. g + 3 may still be on their way
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            Row x;
            fetch(g, x);
            mult(x);
            if (F32 && ++since == 64) { // 256 rows: fp32 partial sums into their fp64 images
                since = 0;
#pragma unroll
                for (int i = 0; i < NP; i++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) acc64[F32 ? i : 0][r] += (double)acc[i][r];
                    acc[i] = acc_t{0, 0, 0, 0};
                }
                if (TAIL) {
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        t64[t][lane] += (double)ta[t], t64[NT + t][lane] += (double)tb[t];
                        ta[t] = tb[t] = (T)0;
                    }
#pragma unroll
                    for (int e = 0; e < 3; e++) t64[2 * NT + e][lane] += (double)tt[e], tt[e] = (T)0;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (nothing may land in LDS after the wavefront has gone)
        if (ulen & 3) {                                   // the last one to three rows: lane groups beyond the end multiply zeros
            Row x;
            fetch(ng, x);
            if (4 * ng + lg >= ulen) {
#pragma unroll
                for (int t = 0; t < NT; t++) x.x[t] = (T)0;
                x.tl[0] = x.tl[1] = (T)0;
            }
            mult(x);
        }
    }
    double *out = Gcols + (size_t)col * KP * KP;
    auto put = [&](int i, int j, double sum) { // both triangles, or (upper_only: the solvers read G[min][max]) the upper one
        const double v = complement ? Gfull[i * KP + j] - sum : sum;
        if (!upper_only || i <= j) out[i * KP + j] = v;
        if (i != j && (!upper_only || j < i)) out[j * KP + i] = v;
    };
    int pi = 0;
#pragma unroll
    for (int a = 0; a < NT; a++)
#pragma unroll
        for (int b = a; b < NT; b++, pi++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 16 * a + M::row_of(lane, r), j = 16 * b + l15;
                double sum = (double)acc[pi][r];
                if (F32) sum += acc64[F32 ? pi : 0][r];
                const double v = complement ? Gfull[i * KP + j] - sum : sum;
                if ((TAIL || (i < k && j < k)) && !(upper_only && i > j)) { // (entries beyond k are never read by the solvers)
                    out[i * KP + j] = v;
                    if (a != b && !upper_only) out[j * KP + i] = v;
                }
            }
    if (TAIL) {
        // tail rows: add the four lane groups (each summed its own row of every group of four); lane group 0 stores row C0, lane
        // group 1 row C1, lane 32 the corner
#pragma unroll
        for (int t = 0; t < NT; t++) {
            double va = (double)ta[t], vb = (double)tb[t];
            if (F32) va += t64[t][lane], vb += t64[NT + t][lane];
            va += __shfl_xor(va, 16, 64);
            va += __shfl_xor(va, 32, 64);
            vb += __shfl_xor(vb, 16, 64);
            vb += __shfl_xor(vb, 32, 64);
            const int q = 16 * t + l15;
            if (lg == 0) put(C0, q, va);
            if (lg == 1 && C1 < k) put(C1, q, vb);
        }
        double c[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            c[e] = (double)tt[e];
            if (F32) c[e] += t64[2 * NT + e][lane];
            c[e] += __shfl_xor(c[e], 16, 64);
            c[e] += __shfl_xor(c[e], 32, 64);
        }
        if (lane == 32) {
            put(C0, C0, c[0]);
            if (C1 < k) {
                put(C0, C1, c[1]);
                put(C1, C1, c[2]);
            }
        }
    }
}

// Per-column Gram of the fp32-operand mode on the fp16 matrix cores: the listed rows come as SPLIT fp16 pairs (hi + lo 2^-11, 22
// bits, scaled by a power of two: the cross products' representation, factor16c_kernel: row = 64 hi halves | 64 lo halves = 256 B),
// 32 rows per step, and  sum_r y_r y_r^T  over a step is  Hi Hi^T + (Hi Lo^T + Lo Hi^T) 2^-11  on v_mfma_f32_16x16x32_f16:
// 3 products per upper tile pair, 16 cycles each for 32 rows -- 60 cycles of matrix pipe per four rows against 192 (tail form) /
// 320 (padded) for v_mfma_f32_16x16x4_f32, and one set of index / address / LDS instructions per 32 rows instead of per 4.
//   * gather: 8 global_load_lds_dwordx4 per step, FOUR WHOLE ROWS each (lane l: a 16-byte chunk of row 4 j + l / 16: a request
//     touches 8 cache lines; one [32 rows][16 halves] subtile per request touched 32 and ran into the L1's tag stage) into a
//     row-major [32 rows][256 B] stage buffer, chunks swizzled through the GLOBAL address so that ds_read_b64_tr_b16 (MFMA operand
//     = 8 halves along k for the lane's coordinate: two transpose reads) is conflict free -- details at the mapping below;
//   * row indices: one global_load_lds_dword per step into a ring of four 256-byte slots, read back with ds_read_b32 -- no VGPR
//     destination, no compiler-visible VMEM: a step is "s_waitcnt vmcnt(9); read this step's operands; issue the gathers of the
//     step after the next one into the buffer just read out; multiply this step" with two 8 KB stage buffers per wavefront;
//   * rows past the end of the list read row p of the split copy, which factor16c_kernel leaves zero;
//   * fp32 accumulators folded into fp64 every 8 steps (256 rows), unscaled by 2^-2e at the end (e: the split copy's exponent).
// The correction is ~the missing fraction of G, so 22-bit products keep G_j at ~1e-8 relative, as the fp32 form did.
typedef _Float16 gh8 __attribute__((ext_vector_type(8)));
typedef _Float16 gh4 __attribute__((ext_vector_type(4)));
template <int NKQ>
__global__ __launch_bounds__(256) void na_gram_f16_kernel(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ meta, const int *__restrict__ idx,
                                                          const uint32_t *__restrict__ Y16rows, int zero_row, const int *__restrict__ exp_in,
                                                          const double *__restrict__ Gfull, double *__restrict__ Gcols, int ncols, int col0, int k, int upper_only)
{
    constexpr int KP = 16 * NKQ;          // row stride of Gfull / Gcols
    constexpr int NP = NKQ * (NKQ + 1) / 2;
    constexpr int STAGE = 8 * 1024;        // bytes: 2 planes x 4 coordinate tiles x [32][16] halves (the split copy always has 64 coordinates)
    __shared__ __attribute__((aligned(1024))) unsigned char stage_all[4][2][STAGE];
    __shared__ int ibuf_all[4][4][64]; // ring of four index slots per wavefront (the indices run four steps ahead)
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4, wave = threadIdx.x >> 6;
    const int col = col0 + blockIdx.x * 4 + wave;
    if (col >= ncols) return; // whole wave
    const uint32_t mt = meta[col];
    const int ulen = __builtin_amdgcn_readfirstlane((int)(mt & 0x7FFFFFFFu));
    const bool complement = (mt >> 31) != 0;
    const int base = __builtin_amdgcn_readfirstlane((int)ptr[col]);
    unsigned char *stage = &stage_all[wave][0][0];
    int *ibuf = &ibuf_all[wave][0][0];

    f32x4 accm[NP], accx[NP];
    f64x4 acc64[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) accm[i] = accx[i] = f32x4{0, 0, 0, 0}, acc64[i] = f64x4{0, 0, 0, 0};

    const int nst = (ulen + 31) >> 5; // steps of 32 rows
    auto sbase = [&](const void *q) { // 64-bit pointer as an SGPR pair
        const unsigned long long v = (unsigned long long)q;
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    };
    const unsigned long long ybase = sbase(Y16rows), ibase = sbase(idx + base);
    const unsigned lds_stage = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)stage);
    const unsigned lds_ibuf = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) int *)ibuf);
    // gather mapping: instruction j of a step fetches rows 4 j .. 4 j + 3 WHOLE (lane l: row 4 j + l / 16, one 16-byte chunk of its
    // 256 bytes), so the 64 lanes touch 8 cache lines instead of 32 (two rows' 32-byte pieces per quad made the L1's tag stage the bound).
    // The stage buffer is row-major [32 rows][256 B]; the DMA puts lane l's 16 bytes at slot l & 15 of its row, so the swizzle that
    // keeps ds_read_b64_tr_b16 conflict free is applied on the GLOBAL side: slot s of row r holds chunk (((s >> 1) ^ f(r)) << 1) | (s & 1),
    // f(r) = (r & 3) | ((r >> 1) & 4) -- the eight rows one half-wavefront's transpose read touches have eight different f.
    const int r4 = lane >> 4;
    const unsigned goff0 = (unsigned)((((lane >> 1) & 7) ^ r4) << 5) | (unsigned)((lane & 1) << 4); // rows with bit 3 clear (j & 2 == 0)
    auto issue_idx = [&](int g) { // entries 32 g + (lane & 31) of the list -> ring slot g & 3 (past the end: the last step's again)
        const int ge = g < nst ? g : nst - 1;
        const unsigned voff = (unsigned)(32 * ge + (lane & 31)) * 4u;
        const unsigned dst = lds_ibuf + (unsigned)(g & 3) * 256u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(ibase), "s"(dst) : "memory");
    };
    auto issue_rows = [&](int g) { // the 32 rows of step g into stage buffer g & 1, four rows per instruction
        const int rem = ulen - 32 * g - r4; // rows 4 j + r4 >= ulen - 32 g read the zero row
        int rows[8];
#pragma unroll
        for (int j = 0; j < 8; j++) rows[j] = ibuf[(g & 3) * 64 + 4 * j + r4];
        const unsigned dst = lds_stage + (unsigned)(g & 1) * (unsigned)STAGE;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int row = (4 * j >= rem) ? zero_row : rows[j];
            const unsigned voff = (unsigned)row * 256u + (goff0 ^ ((j & 2) ? 128u : 0u));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ybase), "s"(dst + (unsigned)j * 1024u) : "memory");
        }
    };
    if (nst > 0) {
        // TWO steps in flight with two stage buffers: a step's operands are in registers before its products start, so its buffer is
        // refilled (step g + 2) in front of the products, not behind them.  Every step issues 1 index request (four steps ahead) + 8
        // row requests, in this order -- "rows of step g have landed" is vmcnt(9), and the indices of step g + 2 are older than those rows.
        issue_idx(0);
        issue_idx(1);
        issue_idx(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue_rows(0);
        issue_idx(3);
        issue_rows(1);
        // operand (plane pl, tile t) of the lane: halves k = 8 lg .. 8 lg + 7 of coordinate 16 t + l15: two transpose reads of
        // [4 rows][16 halves] blocks, lane (4 j + i) of a 16-lane group addressing row 8 lg + j (+ 4), halves 4 i .. 4 i + 3 of chunk
        // pair m = 4 pl + t, which sits at pair slot m ^ f(row): bits 5..7 of the lane's base hold f, so the address is base ^ (m << 5)
        const unsigned tr_lane = (unsigned)(8 * lg + (l15 >> 2)) * 256u + (unsigned)(((l15 >> 2) | ((lg & 1) << 2)) << 5) + (unsigned)(l15 & 3) * 8u;
        int since = 0;
        for (int g = 0; g < nst; g++) {
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); // rows of step g (and indices up to step g + 2) have landed
            const unsigned sb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)stage + (unsigned)(g & 1) * (unsigned)STAGE + tr_lane;
            gh8 hi[NKQ], lo[NKQ];
#pragma unroll
            for (int t = 0; t < NKQ; t++) {
                gh4 a0, a1, b0, b1;
                const unsigned sh = sb ^ (unsigned)(t << 5), sl = sb ^ (unsigned)((4 + t) << 5);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a0) : "v"(sh));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(a1) : "v"(sh));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0) : "v"(sl));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(b1) : "v"(sl));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                hi[t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                lo[t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            issue_idx(g + 4);  // (slot g & 3: its entries were consumed two steps ago)
            issue_rows(g + 2); // buffer g & 1 has been read out (past the end: zero rows, never multiplied)
            __builtin_amdgcn_sched_barrier(0);
            int pi = 0;
#pragma unroll
            for (int a = 0; a < NKQ; a++)
#pragma unroll
                for (int b = a; b < NKQ; b++, pi++) {
                    accm[pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi[a], hi[b], accm[pi], 0, 0, 0);
                    accx[pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi[a], lo[b], accx[pi], 0, 0, 0);
                    accx[pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo[a], hi[b], accx[pi], 0, 0, 0);
                }
            if (++since == 8) { // 256 rows
                since = 0;
#pragma unroll
                for (int i = 0; i < NP; i++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) acc64[i][r] += (double)accm[i][r] + (double)accx[i][r] * (1.0 / 2048.0);
                    accm[i] = accx[i] = f32x4{0, 0, 0, 0};
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (nothing may land in LDS after the wavefront has gone)
    }
    const double unscale = ldexp(1.0, -2 * exp_in[0]);
    double *out = Gcols + (size_t)col * KP * KP;
    int pi = 0;
#pragma unroll
    for (int a = 0; a < NKQ; a++)
#pragma unroll
        for (int b = a; b < NKQ; b++, pi++) {
            // C/D layout of the 16x16 fp32 tile: row i = 16 a + 4 (lane >> 4) + r, column j = 16 b + (lane & 15)
            const int i0 = 16 * a + 4 * lg, j = 16 * b + l15;
            f64x4 v4;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double sum = (acc64[pi][r] + (double)accm[pi][r] + (double)accx[pi][r] * (1.0 / 2048.0)) * unscale;
                v4[r] = complement ? Gfull[(i0 + r) * KP + j] - sum : sum;
            }
            // upper_only (the one-column-per-wavefront solvers read G[min][max]): nothing below the diagonal is written.  Otherwise
            // (colsolve_row_kernel reads whole rows) the mirrored tile goes out as ONE 32-byte store per lane -- the lane's four rows i are
            // four consecutive entries of row j, the four lane groups fill a 128-byte line -- not as four 8-byte stores 512 bytes apart
            // (those were 2.4 % of a config-5 iteration).  Entries beyond k are never read by the solvers (the buffer has KP x KP of them).
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = i0 + r;
                if (i < k && j < k && !(upper_only && i > j)) out[i * KP + j] = v4[r];
            }
            if (a != b && !upper_only && j < k) *(f64x4 *)(out + j * KP + i0) = v4;
        }
}

// colsolve_strict_kernel -- SCD-LS per column in the REFERENCE's arithmetic (strict fp64 mode) with the structure of
// the fp32-operand mode's former solver: one wavefront per column, lane = coordinate, row `lane` of the edited Gram in registers, the coordinate loop
// fully unrolled.  Every lane evaluates the step of ITS coordinate from its own x, mu, G[lane][lane] -- tmp = max(x - mu / G, 0) with
// the correctly rounded quotient (reciprocal + Markstein correction, k_sweep.h), d = tmp - x -- and lane q's d is the step's delta:
// 2 v_readlane_b32 + 1 v_fma_f64 bring mu up to date (d = 0 when the reference skips the coordinate: adds nothing).  Lane q keeps
// tmp itself (x + d is not always tmp) through two v_cndmask_b32 under a literal lane mask.  The rel-change test runs once per sweep
// on all lanes (a coordinate moves once per sweep: same maximum), division free as in colsolve_ls_kernel.  11 VALU instructions per
// coordinate against ~15 + a rolled loop with register-indexed Gram reads and eight v_readlane_b32 in colsolve_ls_kernel.
template <int NKQ, bool HAS_MASK, int KR = 16 * NKQ>
__global__ __launch_bounds__(256) void colsolve_strict_kernel(const SweepArgs a, size_t g_stride)
{
    const int lane = threadIdx.x & 63;
    const int col = a.col0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= a.ncols) return; // whole wavefront
    const int k = a.k;
    const bool lv = lane < k;
    const int lq = lv ? lane : 0;
    const double *G = a.Graw + (size_t)col * g_stride;
    unsigned long long mword = 0ull;
    if (HAS_MASK) mword = a.mask[col];
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
    const bool skip = HAS_MASK && ((mword & kmask) == kmask); // arma::all(mask.col(j)), src/update_with_missing.cpp:75-76

    double gd = 1.0; // edited G[lane][lane] (src/update_with_missing.cpp:98-103)
    if (lv) {
        gd = G[(size_t)lq * a.KPg + lq];
        if (a.r0 != a.r1) gd += a.r0 - a.r1;
        if (a.r1 != 0) gd += a.r1;
        gd += NNLM_TINY;
    }
    const double rgd = 1.0 / gd;
    double g[KR]; // row `lane` of the edited Gram (G is symmetric: G[lane][q] = G[q][lane], a coalesced read)
#pragma unroll
    for (int q = 0; q < KR; q++) {
        double v = 0.0;
        if (q < k && lv) {
            v = G[(a.g_upper && q > lane) ? (size_t)lane * a.KPg + q : (size_t)q * a.KPg + lane];
            if (q == lane && a.r0 != a.r1) v += a.r0 - a.r1;
            if (a.r1 != 0) v += a.r1;
            if (q == lane) v += NNLM_TINY;
        }
        g[q] = v;
    }
    double x = lv ? a.X[(size_t)lq * a.ldx + col] : 0.0;
    double cv = 0.0;
    if (lv)
        for (int s = 0; s < a.nslabs; s++) cv += a.Cx[(size_t)s * a.slab_stride + (size_t)lq * a.ldc + col];
    // mu = G x - c (+ L1), summed in the order of colsolve_ls_kernel
    double mu = 0.0;
#pragma unroll
    for (int q = 0; q < KR; q++)
        if (q < k) mu = __builtin_fma(g[q], readlane_f64(x, q), mu);
    mu -= cv;
    if (a.r2 != 0) mu += a.r2;
    if (!lv) mu = 0.0;

    unsigned t = 0;
    if (!skip) {
        bool more = true; // rel = 1 + rel_tol > rel_tol
        for (; t < a.max_iter && more; t++) {
            double xn = x; // the sweep's new coordinates (lane q's changes at step q; x keeps the sweep's start for the test below)
            int kk = k;
            asm volatile("" : "+s"(kk)); // (opaque per sweep: otherwise 64 hoisted "q < k" masks spill into VGPR lanes)
            auto step = [&](const int q) {
                const double q0 = mu * rgd;
                const double rr = __builtin_fma(-q0, gd, mu);
                const double quo = __builtin_fma(rr, rgd, q0); // = mu / G[lane][lane], correctly rounded
                double tmp;
                asm("v_max_f64 %0, %1, 0" : "=v"(tmp) : "v"(x - quo)); // tmp = x - mu / G; if (tmp < 0) tmp = 0   (base_algorithms.cpp:23-24)
                const double dd = tmp - x;
                int2 dp = __builtin_bit_cast(int2, dd);
                const int dlo = __builtin_amdgcn_readlane(dp.x, q), dhi = __builtin_amdgcn_readlane(dp.y, q);
                {
                    // mu += (tmp - x) * G.col(q) (:26) as a product and a sum, like the reference's build (no FMA contraction on x86-64): with
                    // nothing observed in a column and no regularisation G = NNLM_TINY I, mu = fl(TINY x), and the step to 0 must leave
                    // mu = fl(TINY x) - fl(x TINY) = 0 exactly -- a fused multiply-add leaves the product's rounding error, which the next
                    // sweep turns into 1e-17 of dust and, half of the time, one more counted sweep than the reference runs
#pragma clang fp contract(off)
                    const double prod = __builtin_bit_cast(double, int2{dlo, dhi}) * g[q];
                    mu = mu + prod;
                }
                int2 xp = __builtin_bit_cast(int2, xn), tp = __builtin_bit_cast(int2, tmp);
                asm volatile("s_mov_b32 vcc_lo, %4\n\ts_mov_b32 vcc_hi, %5\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc"
                             : "+v"(xp.x), "+v"(xp.y)
                             : "v"(tp.x), "v"(tp.y), "n"(q < 32 ? (int)(1u << (q & 31)) : 0), "n"(q >= 32 ? (int)(1u << (q & 31)) : 0)
                             : "vcc");
                xn = __builtin_bit_cast(double, xp);
            };
#pragma unroll
            for (int c = 0; c < NKQ; c++) {
                if (!HAS_MASK && 16 * c + 16 <= KR && 16 * c + 16 <= kk) { // a whole block of 16 coordinates: no per-step test
#pragma unroll
                    for (int e = 0; e < 16; e++) step(16 * c + e);
                } else if (16 * c < kk) {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        if (16 * c + e < KR) // (compile time)
                            if (16 * c + e < kk && !(HAS_MASK && ((mword >> (16 * c + e)) & 1ull))) step(16 * c + e); // wave-uniform
                }
            }
            const bool big = rel_change_exceeds(2 * fabs(x - xn), xn + x + NNLM_TINY, a.rel_tol); // src/base_algorithms.cpp:29-32, the quotient's decision
            x = xn;
            more = __ballot(big && lv) != 0ull || 0.0 > a.rel_tol;
        }
    }
    if (lv) {
        a.Xout[(size_t)lane * a.ldo + (col - a.ocol0)] = x;
        if (a.op_mode == 1) {
            if (a.op_f64) ((double *)a.op)[(size_t)lane * a.op_ld + col] = x;
            else ((float *)a.op)[(size_t)lane * a.op_ld + col] = (float)x;
        }
    }
    if (lane == 0 && t) atomicAdd(a.sweeps, (unsigned long long)t);
}
