// k_xerr.h -- experimental forms of the fused cross product / error block (k_xprod16.h xprod16_err_kernel), timed by xerr_exp.hip.
// Not part of the product.
//
// xerr3_kernel<NKQ = 4, SPREAD, TIMING>: same images, same arithmetic and the same accumulation orders as xprod16_err_kernel (the cross
// product is bit-identical, the error sums too), different schedule of a stage:
//   * requests of the next stage with a scalar base (glds16_s, as xprod16_tn_kernel), SPREAD = 1: handed out over the steps of the stage;
//   * W H tile by tile: step t issues the 6 MFMAs of tile t and the 6 cross-product MFMAs of N-tile t, the fragment reads of step t + 1
//     and -- in the shadow of those MFMAs -- the error arithmetic of tile t - 1 (the product form: all 56 MFMAs, then all of it);
//   * interior stages (no edge tests) are one basic block; edge stages take the plain form.
#pragma once
#include "csrc_r5/k_xprod16.h"
#include <type_traits>

#define XE_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// xerr0_kernel: the form of rounds 1-4 (eight wavefronts x 16 rows, each doing the cross product, W H, the error arithmetic and its share
// of the requests; ring of two 64 KB stages) -- the reference every other form is checked against bit for bit.
#define XPROD16_ERR_BUF (XPROD_A_IMG_BYTES + 64 * XPROD_ROWB + 64 * XPROD_ROWB)
template <int NKQ>
__global__ __launch_bounds__(XPROD_THREADS) void xerr0_kernel(const uint32_t *__restrict__ A16, int lda, const uint32_t *__restrict__ Y16, int ldy,
                                                                    const uint32_t *__restrict__ H16c, const uint32_t *__restrict__ W16c,
                                                                    double *__restrict__ Cx, int ldc, size_t slab_stride, int stage_begin,
                                                                    int stage_end, int stages_per_split, const int *__restrict__ scal_exp,
                                                                    const int *__restrict__ w_exp, int n_rows, int n_cols,
                                                                    double *__restrict__ partial)
{
    constexpr int KP = 16 * NKQ;
    constexpr int FL = XPROD_FLUSH_ELEMS / 64;
    constexpr int YOFF = XPROD_A_IMG_BYTES, HOFF = XPROD_A_IMG_BYTES + 64 * XPROD_ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[2][XPROD_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int i0 = blockIdx.x * XPROD_TN_BJ;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;

    f32x4 accm[NKQ], accx[NKQ];
    f64x4 acc64[NKQ];
#pragma unroll
    for (int b = 0; b < NKQ; b++) {
        accm[b] = f32x4{0, 0, 0, 0};
        accx[b] = f32x4{0, 0, 0, 0};
        acc64[b] = f64x4{0, 0, 0, 0};
    }
    // this wave's 16 rows of W, kq-contiguous: lane (l15 = row, lg) holds kq = 32c + 8lg .. +7 of both halves
    xh8 wh[2], wl[2];
    {
        const _Float16 *wrow = (const _Float16 *)(W16c + (size_t)(i0 + 16 * wave + l15) * 64);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            wh[c] = *(const xh8 *)(wrow + 32 * c + 8 * lg);
            wl[c] = *(const xh8 *)(wrow + 64 + 32 * c + 8 * lg);
        }
    }
    // one-hot B operands that move the A fragment (M = row, K = 32 columns) into the accumulator layout of W H:
    // ident[u][k] = 1 iff column k of the K chunk is column 16u + n of the lane's 16-column tile (n = l15)
    xh8 ident[2];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int e = 0; e < 8; e++) ident[u][e] = (8 * lg + e == 16 * u + l15) ? (_Float16)1.0f : (_Float16)0.0f;
    const float ca = ldexpf(1.0f, -scal_exp[0]);                    // a      = (hi + lo/2048) * ca
    const float cwh = ldexpf(1.0f, -(w_exp[0] + scal_exp[1]));      // (W H)  = (main + cross/2048) * cwh
    double s2 = 0.0, skl = 0.0;

    auto issue = [&](int st, unsigned char *buf) {
        const size_t c0 = (size_t)st * 64;
#pragma unroll
        for (int t = wave; t < XPROD_A_IMG_BYTES / 1024; t += XPROD_WAVES) {
            const int row = 4 * t + lg;
            const int s = l15 ^ (row & 15);
            glds16(A16 + (size_t)(i0 + row) * lda + c0 + s * 4, buf + t * 1024);
        }
#pragma unroll
        for (int t = wave; t < KP / 4; t += XPROD_WAVES) {
            const int row = 4 * t + lg;
            const int s = l15 ^ (row & 15);
            glds16(Y16 + (size_t)row * ldy + c0 + s * 4, buf + YOFF + t * 1024);
        }
#pragma unroll
        for (int t = wave; t < 16; t += XPROD_WAVES) { // 64 columns j of the stage, 256 bytes (64 hi | 64 lo over kq) each
            const int row = 4 * t + lg;
            const int s = l15 ^ (row & 15);
            glds16(H16c + (c0 + row) * 64 + s * 4, buf + HOFF + t * 1024);
        }
    };
    if (st0 < st1) issue(st0, smem);
    int since_flush = 0;
    for (int st = st0; st < st1; ++st) {
        unsigned char *buf = smem + ((st - st0) & 1) * XPROD16_ERR_BUF;
        wait_vmcnt(0);
        __builtin_amdgcn_s_barrier();
        if (st + 1 < st1) issue(st + 1, smem + ((st + 1 - st0) & 1) * XPROD16_ERR_BUF);
        f32x4 em[4], ex[4], dh[4], dl[4];
#pragma unroll
        for (int t = 0; t < 4; t++) em[t] = f32x4{0, 0, 0, 0}, ex[t] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
            const int arow = 16 * wave + l15;
            const int sh = (4 * c2 + lg), sl = 8 + 4 * c2 + lg;
            const xh8 ah = *(const xh8 *)(buf + arow * XPROD_ROWB + ((sh ^ l15) * 16));
            const xh8 al = *(const xh8 *)(buf + arow * XPROD_ROWB + ((sl ^ l15) * 16));
#pragma unroll
            for (int nt = 0; nt < NKQ; nt++) {
                const unsigned char *yrow = buf + YOFF + (16 * nt + l15) * XPROD_ROWB;
                const xh8 yh = *(const xh8 *)(yrow + ((sh ^ l15) * 16));
                const xh8 yl = *(const xh8 *)(yrow + ((sl ^ l15) * 16));
                accm[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yh, accm[nt], 0, 0, 0);
                accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yl, accx[nt], 0, 0, 0);
                accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, yh, accx[nt], 0, 0, 0);
            }
            // a(i, j) of this wave's rows in the accumulator layout (exact: one product with 1.0 per element)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                dh[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
                dl[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
            }
            // W H for this wave's 16 rows x the stage's 64 columns (contraction over kq = 32*c2 ..)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const unsigned char *hrow = buf + HOFF + (16 * t + l15) * XPROD_ROWB;
                const xh8 hh = *(const xh8 *)(hrow + ((sh ^ l15) * 16));
                const xh8 hl = *(const xh8 *)(hrow + ((sl ^ l15) * 16));
                em[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hh, em[t], 0, 0, 0);
                ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hl, ex[t], 0, 0, 0);
                ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c2], hh, ex[t], 0, 0, 0);
            }
        }
        // the two sums over the 16 x 64 tile: lane (l15 = column within tile t, lg) holds rows 4*lg + r (vector arithmetic over
        // r so that the multiplies and adds pair up into v_pk_*_f32)
        {
            f32x4 p2 = f32x4{0, 0, 0, 0}, pk = f32x4{0, 0, 0, 0};
            const bool interior = (i0 + XPROD_TN_BJ <= n_rows) && (st * 64 + 64 <= n_cols);
            const float il = 1.0f / XPROD16_LO_SCALE, tiny = (float)NNLM_TINY;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const f32x4 aa = (dh[t] + dl[t] * il) * ca;
                const f32x4 ah2 = (em[t] + ex[t] * il) * cwh;
                const f32x4 d = aa - ah2;
                f32x4 lg4;
#pragma unroll
                for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny); // (log2: ln 2 goes into the coefficient)
                f32x4 t2 = d * d;
                f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
                if (!interior) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const bool valid = (i0 + 16 * wave + 4 * lg + r < n_rows) && (st * 64 + 16 * t + l15 < n_cols);
                        if (!valid) t2[r] = 0.f, tk[r] = 0.f;
                    }
                }
                p2 += t2;
                pk += tk;
            }
            s2 += (double)((p2[0] + p2[1]) + (p2[2] + p2[3]));
            skl += (double)((pk[0] + pk[1]) + (pk[2] + pk[3]));
        }
        if (++since_flush == FL) {
            since_flush = 0;
#pragma unroll
            for (int b = 0; b < NKQ; b++) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc64[b][r] += (double)accm[b][r] + (double)accx[b][r] * (1.0 / XPROD16_LO_SCALE);
                accm[b] = f32x4{0, 0, 0, 0};
                accx[b] = f32x4{0, 0, 0, 0};
            }
        }
    }
    const double unscale = ldexp(1.0, -(scal_exp[0] + scal_exp[1]));
    double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
    for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int kq = 16 * nt + l15;
            const int j = i0 + 16 * wave + 4 * lg + r;
            const double v = acc64[nt][r] + (double)accm[nt][r] + (double)accx[nt][r] * (1.0 / XPROD16_LO_SCALE);
            out[(size_t)kq * ldc + j] = v * unscale;
        }
    s2 = wave_sum(s2);
    skl = wave_sum(skl);
    if (lane == 0) red[0][wave] = s2, red[1][wave] = skl;
    __syncthreads();
    if (tid == 0) {
        double a2 = 0.0, ak = 0.0;
        for (int w = 0; w < XPROD_WAVES; w++) a2 += red[0][w], ak += red[1][w];
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = a2;
        partial[2 * blk + 1] = ak;
    }
}


template <int NKQ, int SPREAD, int TIMING, int PAT>
__global__ __launch_bounds__(XPROD_THREADS) void xerr3_kernel(const uint32_t *__restrict__ A16, int lda, const uint32_t *__restrict__ Y16, int ldy,
                                                              const uint32_t *__restrict__ H16c, const uint32_t *__restrict__ W16c,
                                                              double *__restrict__ Cx, int ldc, size_t slab_stride, int stage_begin, int stage_end,
                                                              int stages_per_split, const int *__restrict__ scal_exp, const int *__restrict__ w_exp,
                                                              int n_rows, int n_cols, double *__restrict__ partial, unsigned long long *__restrict__ tim)
{
    static_assert(NKQ == 4, "experiment: rank 49..64");
    constexpr int FL = XPROD_FLUSH_ELEMS / 64;
    constexpr int YOFF = XPROD_A_IMG_BYTES, HOFF = XPROD_A_IMG_BYTES + 64 * XPROD_ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[2][XPROD_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int i0 = blockIdx.x * XPROD_TN_BJ;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;

    f32x4 accm[NKQ], accx[NKQ];
    f64x4 acc64[NKQ];
#pragma unroll
    for (int b = 0; b < NKQ; b++) {
        accm[b] = f32x4{0, 0, 0, 0};
        accx[b] = f32x4{0, 0, 0, 0};
        acc64[b] = f64x4{0, 0, 0, 0};
    }
    xh8 wh[2], wl[2];
    {
        const _Float16 *wrow = (const _Float16 *)(W16c + (size_t)(i0 + 16 * wave + l15) * 64);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            wh[c] = *(const xh8 *)(wrow + 32 * c + 8 * lg);
            wl[c] = *(const xh8 *)(wrow + 64 + 32 * c + 8 * lg);
        }
    }
    xh8 ident[2];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int e = 0; e < 8; e++) ident[u][e] = (8 * lg + e == 16 * u + l15) ? (_Float16)1.0f : (_Float16)0.0f;
    const float ca = ldexpf(1.0f, -scal_exp[0]);
    const float cwh = ldexpf(1.0f, -(w_exp[0] + scal_exp[1]));
    double s2 = 0.0, skl = 0.0;

    // requests: piece t = wave + 8 i of an image = rows 4 t + lg = rw + 32 i; one per-lane byte offset per image, the rest scalar
    const int rw = 4 * wave + lg, sw = l15 ^ (rw & 15);
    const unsigned voffA = (unsigned)(((size_t)rw * lda + sw * 4) * 4), voffY = (unsigned)(((size_t)rw * ldy + sw * 4) * 4);
    const unsigned voffH = (unsigned)((rw * 64 + sw * 4) * 4);
    const unsigned long long baseA = xp_uniform64(A16 + (size_t)i0 * lda), baseY = xp_uniform64(Y16), baseH = xp_uniform64(H16c);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    // request r of a wavefront's eight per stage: 0..3 A, 4..5 factor rows (Y), 6..7 factor columns (H)
    auto issue1 = [&](int st, int bi, int r) {
        const unsigned long long c0b = (unsigned long long)st * 256ull;
        const unsigned dst = lds0 + (unsigned)bi * (unsigned)XPROD16_ERR_BUF + (unsigned)wave * 1024u;
        if (r < 4)
            glds16_s(voffA, baseA + c0b + (unsigned long long)r * 32ull * (unsigned long long)lda * 4ull, dst + (unsigned)r * 8192u);
        else if (r < 6)
            glds16_s(voffY, baseY + c0b + (unsigned long long)(r - 4) * 32ull * (unsigned long long)ldy * 4ull, dst + (unsigned)YOFF + (unsigned)(r - 4) * 8192u);
        else
            glds16_s(voffH, baseH + (unsigned long long)st * 64ull * 256ull + (unsigned long long)(r - 6) * 32ull * 256ull, dst + (unsigned)HOFF + (unsigned)(r - 6) * 8192u);
    };
    auto issue_all = [&](int st, int bi) {
#pragma unroll
        for (int r = 0; r < 8; r++) issue1(st, bi, r);
    };
    // per-lane byte offsets of the fragment reads inside an image row
    const int oh0 = ((0 + lg) ^ l15) * 16, oh1 = ((4 + lg) ^ l15) * 16, ol0 = ((8 + lg) ^ l15) * 16, ol1 = ((12 + lg) ^ l15) * 16;
    const float il = 1.0f / XPROD16_LO_SCALE, tiny = (float)NNLM_TINY;

    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int i) {
        if constexpr (TIMING) {
            unsigned long long tn_;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");
            tacc[i] += tn_ - tprev;
            tprev = tn_;
        }
    };
    if constexpr (TIMING) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");

    if (st0 < st1) issue_all(st0, 0);
    int since_flush = 0;
    for (int st = st0; st < st1; ++st) {
        unsigned char *buf = smem + ((st - st0) & 1) * XPROD16_ERR_BUF;
        const int nbi = (st + 1 - st0) & 1;
        const bool more = st + 1 < st1;
        const int stn = more ? st + 1 : st;
        wait_vmcnt(0);
        mark(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        mark(1);
        const bool interior = (i0 + XPROD_TN_BJ <= n_rows) && (st * 64 + 64 <= n_cols);
        if (interior && PAT != 2) {
            // (the last stage of a block repeats its own requests into the idle buffer: no branch inside the stage)
            if (SPREAD == 0) issue_all(stn, nbi);
            else { issue1(stn, nbi, 0); issue1(stn, nbi, 1); }
            const unsigned char *arowp = buf + (16 * wave + l15) * XPROD_ROWB;
            xh8 ah[2], al[2];
            ah[0] = *(const xh8 *)(arowp + oh0), al[0] = *(const xh8 *)(arowp + ol0);
            ah[1] = *(const xh8 *)(arowp + oh1), al[1] = *(const xh8 *)(arowp + ol1);
            xh8 hh[2][2], hl[2][2], yh[2][2], yl[2][2];
            auto read_hy = [&](int t, int b) {
                const unsigned char *hrow = buf + HOFF + (16 * t + l15) * XPROD_ROWB;
                hh[b][0] = *(const xh8 *)(hrow + oh0), hl[b][0] = *(const xh8 *)(hrow + ol0);
                hh[b][1] = *(const xh8 *)(hrow + oh1), hl[b][1] = *(const xh8 *)(hrow + ol1);
                const unsigned char *yrow = buf + YOFF + (16 * t + l15) * XPROD_ROWB;
                yh[b][0] = *(const xh8 *)(yrow + oh0), yl[b][0] = *(const xh8 *)(yrow + ol0);
                yh[b][1] = *(const xh8 *)(yrow + oh1), yl[b][1] = *(const xh8 *)(yrow + ol1);
            };
            read_hy(0, 0);
            f32x4 dh[4], dl[4];
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    dh[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c2], ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
                    dl[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c2], ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
                }
            if (SPREAD) { issue1(stn, nbi, 2); issue1(stn, nbi, 3); }
            __builtin_amdgcn_sched_barrier(0);
            mark(2);
            f32x4 em[2], ex[2];
            f32x4 p2 = f32x4{0, 0, 0, 0}, pk = f32x4{0, 0, 0, 0};
            auto epi = [&](int t, const f32x4 &em_, const f32x4 &ex_) {
                const f32x4 aa = (dh[t] + dl[t] * il) * ca;
                const f32x4 ah2 = (em_ + ex_ * il) * cwh;
                const f32x4 d = aa - ah2;
                f32x4 lg4;
#pragma unroll
                for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny);
                f32x4 t2 = d * d;
                f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
                p2 += t2;
                pk += tk;
            };
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int b = t & 1;
                if (t < 3) read_hy(t + 1, b ^ 1);
                if (SPREAD && t < 2) { issue1(stn, nbi, 4 + 2 * t); issue1(stn, nbi, 5 + 2 * t); }
                em[b] = f32x4{0, 0, 0, 0}, ex[b] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int c2 = 0; c2 < 2; c2++) {
                    em[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hh[b][c2], em[b], 0, 0, 0);
                    ex[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hl[b][c2], ex[b], 0, 0, 0);
                    ex[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c2], hh[b][c2], ex[b], 0, 0, 0);
                }
#pragma unroll
                for (int c2 = 0; c2 < 2; c2++) {
                    accm[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c2], yh[b][c2], accm[t], 0, 0, 0);
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c2], yl[b][c2], accx[t], 0, 0, 0);
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c2], yh[b][c2], accx[t], 0, 0, 0);
                }
                if (t > 0) epi(t - 1, em[b ^ 1], ex[b ^ 1]);
                if (PAT == 1) {
                    if (t < 3) XE_SGB(0x100, 8);
#pragma unroll
                    for (int i = 0; i < 12; i++) {
                        XE_SGB(0x8, 1);
                        if (t > 0) XE_SGB(0x402, 2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            mark(3);
            epi(3, em[1], ex[1]);
            s2 += (double)((p2[0] + p2[1]) + (p2[2] + p2[3]));
            skl += (double)((pk[0] + pk[1]) + (pk[2] + pk[3]));
            mark(4);
        } else {
            if (more) issue_all(st + 1, nbi);
            f32x4 em[4], ex[4], dh[4], dl[4];
#pragma unroll
            for (int t = 0; t < 4; t++) em[t] = f32x4{0, 0, 0, 0}, ex[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++) {
                const int arow = 16 * wave + l15;
                const int sh = (4 * c2 + lg), sl = 8 + 4 * c2 + lg;
                const xh8 ah = *(const xh8 *)(buf + arow * XPROD_ROWB + ((sh ^ l15) * 16));
                const xh8 al = *(const xh8 *)(buf + arow * XPROD_ROWB + ((sl ^ l15) * 16));
#pragma unroll
                for (int nt = 0; nt < NKQ; nt++) {
                    const unsigned char *yrow = buf + YOFF + (16 * nt + l15) * XPROD_ROWB;
                    const xh8 yh = *(const xh8 *)(yrow + ((sh ^ l15) * 16));
                    const xh8 yl = *(const xh8 *)(yrow + ((sl ^ l15) * 16));
                    accm[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yh, accm[nt], 0, 0, 0);
                    accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yl, accx[nt], 0, 0, 0);
                    accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, yh, accx[nt], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    dh[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
                    dl[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const unsigned char *hrow = buf + HOFF + (16 * t + l15) * XPROD_ROWB;
                    const xh8 hh = *(const xh8 *)(hrow + ((sh ^ l15) * 16));
                    const xh8 hl = *(const xh8 *)(hrow + ((sl ^ l15) * 16));
                    em[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hh, em[t], 0, 0, 0);
                    ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hl, ex[t], 0, 0, 0);
                    ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c2], hh, ex[t], 0, 0, 0);
                }
            }
            f32x4 p2 = f32x4{0, 0, 0, 0}, pk = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const f32x4 aa = (dh[t] + dl[t] * il) * ca;
                const f32x4 ah2 = (em[t] + ex[t] * il) * cwh;
                const f32x4 d = aa - ah2;
                f32x4 lg4;
#pragma unroll
                for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny);
                f32x4 t2 = d * d;
                f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool valid = (i0 + 16 * wave + 4 * lg + r < n_rows) && (st * 64 + 16 * t + l15 < n_cols);
                    if (!valid) t2[r] = 0.f, tk[r] = 0.f;
                }
                p2 += t2;
                pk += tk;
            }
            s2 += (double)((p2[0] + p2[1]) + (p2[2] + p2[3]));
            skl += (double)((pk[0] + pk[1]) + (pk[2] + pk[3]));
        }
        if (++since_flush == FL) {
            since_flush = 0;
#pragma unroll
            for (int b = 0; b < NKQ; b++) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc64[b][r] += (double)accm[b][r] + (double)accx[b][r] * (1.0 / XPROD16_LO_SCALE);
                accm[b] = f32x4{0, 0, 0, 0};
                accx[b] = f32x4{0, 0, 0, 0};
            }
        }
        mark(5);
    }
    wait_vmcnt(0);
    const double unscale = ldexp(1.0, -(scal_exp[0] + scal_exp[1]));
    double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
    for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int kq = 16 * nt + l15;
            const int j = i0 + 16 * wave + 4 * lg + r;
            const double v = acc64[nt][r] + (double)accm[nt][r] + (double)accx[nt][r] * (1.0 / XPROD16_LO_SCALE);
            out[(size_t)kq * ldc + j] = v * unscale;
        }
    s2 = wave_sum(s2);
    skl = wave_sum(skl);
    if (lane == 0) red[0][wave] = s2, red[1][wave] = skl;
    __syncthreads();
    if (tid == 0) {
        double a2 = 0.0, ak = 0.0;
        for (int w = 0; w < XPROD_WAVES; w++) a2 += red[0][w], ak += red[1][w];
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = a2;
        partial[2 * blk + 1] = ak;
    }
    if constexpr (TIMING) {
        if (blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 40) && lane == 0) {
            unsigned long long *r = tim + ((blockIdx.x == 0 ? 0 : 40) + wave) * 16;
            for (int i = 0; i < 6; i++) r[i] = tacc[i];
            r[15] = (unsigned long long)(st1 - st0);
        }
    }
}

// xerr4_kernel: the product's stage body (compiler-scheduled), but the two halves of a block's wavefronts issue their requests at
// different times so that one wavefront of a SIMD computes while the other waits for the memory pipeline to take its requests:
//   * wavefronts 0..3 ("GA") own the even 16-row tiles of every image, wavefronts 4..7 ("GB") the odd ones (piece t = wave + 8 i);
//   * GA issues stage s + 1 right behind the barrier of stage s (two slots of 32 KB), GB issues stage s + 2 at the END of stage s (three
//     slots of 32 KB): 64 + 96 = 160 KB of LDS, the block's reduction scratch aliases the ring.
#define XE4_SLOT 32768
#define XE4_YOFF 16384
#define XE4_HOFF 24576
template <int NKQ, int SHIFT, int TIMING>
__global__ __launch_bounds__(XPROD_THREADS) void xerr4_kernel(const uint32_t *__restrict__ A16, int lda, const uint32_t *__restrict__ Y16, int ldy,
                                                              const uint32_t *__restrict__ H16c, const uint32_t *__restrict__ W16c,
                                                              double *__restrict__ Cx, int ldc, size_t slab_stride, int stage_begin, int stage_end,
                                                              int stages_per_split, const int *__restrict__ scal_exp, const int *__restrict__ w_exp,
                                                              int n_rows, int n_cols, double *__restrict__ partial, unsigned long long *__restrict__ tim)
{
    static_assert(NKQ == 4, "experiment: rank 49..64");
    constexpr int FL = XPROD_FLUSH_ELEMS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const bool gb = wave >= 4;
    const int i0 = blockIdx.x * XPROD_TN_BJ;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;

    f32x4 accm[NKQ], accx[NKQ];
    f64x4 acc64[NKQ];
#pragma unroll
    for (int b = 0; b < NKQ; b++) {
        accm[b] = f32x4{0, 0, 0, 0};
        accx[b] = f32x4{0, 0, 0, 0};
        acc64[b] = f64x4{0, 0, 0, 0};
    }
    xh8 wh[2], wl[2];
    {
        const _Float16 *wrow = (const _Float16 *)(W16c + (size_t)(i0 + 16 * wave + l15) * 64);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            wh[c] = *(const xh8 *)(wrow + 32 * c + 8 * lg);
            wl[c] = *(const xh8 *)(wrow + 64 + 32 * c + 8 * lg);
        }
    }
    xh8 ident[2];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int e = 0; e < 8; e++) ident[u][e] = (8 * lg + e == 16 * u + l15) ? (_Float16)1.0f : (_Float16)0.0f;
    const float ca = ldexpf(1.0f, -scal_exp[0]);
    const float cwh = ldexpf(1.0f, -(w_exp[0] + scal_exp[1]));
    double s2 = 0.0, skl = 0.0;

    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int i) {
        if constexpr (TIMING) {
            unsigned long long tn_;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");
            tacc[i] += tn_ - tprev;
            tprev = tn_;
        }
    };
    if constexpr (TIMING) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");

    // slot of stage number rel (counted from st0) for a group
    auto slot_of = [&](int rel, bool b) -> unsigned char * { return b ? smem + 2 * XE4_SLOT + (rel % 3) * XE4_SLOT : smem + (rel & 1) * XE4_SLOT; };
    // this wavefront's eight requests of a stage: piece t = wave + 8 i = rows 4 t + lg of the image = row 4 (wave & 3) + lg of local tile i
    auto issue = [&](int st) {
        unsigned char *dst = slot_of(st - st0, gb) + (wave & 3) * 1024;
        const size_t c0 = (size_t)st * 64;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = 4 * (wave + 8 * i) + lg;
            const int s = l15 ^ (row & 15);
            glds16(A16 + (size_t)(i0 + row) * lda + c0 + s * 4, dst + i * 4096);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = 4 * (wave + 8 * i) + lg;
            const int s = l15 ^ (row & 15);
            glds16(Y16 + (size_t)row * ldy + c0 + s * 4, dst + XE4_YOFF + i * 4096);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = 4 * (wave + 8 * i) + lg;
            const int s = l15 ^ (row & 15);
            glds16(H16c + (c0 + row) * 64 + s * 4, dst + XE4_HOFF + i * 4096);
        }
    };
    if (st0 < st1) issue(st0);
    if (SHIFT && gb && st0 + 1 < st1) issue(st0 + 1);
    int since_flush = 0;
    for (int st = st0; st < st1; ++st) {
        const int rel = st - st0;
        const unsigned char *bufe = slot_of(rel, false), *bufo = slot_of(rel, true);
        if (SHIFT && gb && st + 1 < st1) wait_vmcnt(8);
        else wait_vmcnt(0);
        mark(0);
        __builtin_amdgcn_s_barrier();
        mark(1);
        if (st + 1 < st1 && (!SHIFT || !gb)) issue(st + 1);
        // tile T of an image: even tiles in GA's slot, odd tiles in GB's
        auto tile = [&](int T, int xoff) -> const unsigned char * { return ((T & 1) ? bufo : bufe) + xoff + (T >> 1) * 4096 + l15 * XPROD_ROWB; };
        f32x4 em[4], ex[4], dh[4], dl[4];
#pragma unroll
        for (int t = 0; t < 4; t++) em[t] = f32x4{0, 0, 0, 0}, ex[t] = f32x4{0, 0, 0, 0};
        const unsigned char *arowp = ((wave & 1) ? bufo : bufe) + (wave >> 1) * 4096 + l15 * XPROD_ROWB;
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
            const int sh = (4 * c2 + lg), sl = 8 + 4 * c2 + lg;
            const xh8 ah = *(const xh8 *)(arowp + ((sh ^ l15) * 16));
            const xh8 al = *(const xh8 *)(arowp + ((sl ^ l15) * 16));
#pragma unroll
            for (int nt = 0; nt < NKQ; nt++) {
                const unsigned char *yrow = tile(nt, XE4_YOFF);
                const xh8 yh = *(const xh8 *)(yrow + ((sh ^ l15) * 16));
                const xh8 yl = *(const xh8 *)(yrow + ((sl ^ l15) * 16));
                accm[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yh, accm[nt], 0, 0, 0);
                accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yl, accx[nt], 0, 0, 0);
                accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, yh, accx[nt], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                dh[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
                dl[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const unsigned char *hrow = tile(t, XE4_HOFF);
                const xh8 hh = *(const xh8 *)(hrow + ((sh ^ l15) * 16));
                const xh8 hl = *(const xh8 *)(hrow + ((sl ^ l15) * 16));
                em[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hh, em[t], 0, 0, 0);
                ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hl, ex[t], 0, 0, 0);
                ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c2], hh, ex[t], 0, 0, 0);
            }
        }
        {
            f32x4 p2 = f32x4{0, 0, 0, 0}, pk = f32x4{0, 0, 0, 0};
            const bool interior = (i0 + XPROD_TN_BJ <= n_rows) && (st * 64 + 64 <= n_cols);
            const float il = 1.0f / XPROD16_LO_SCALE, tiny = (float)NNLM_TINY;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const f32x4 aa = (dh[t] + dl[t] * il) * ca;
                const f32x4 ah2 = (em[t] + ex[t] * il) * cwh;
                const f32x4 d = aa - ah2;
                f32x4 lg4;
#pragma unroll
                for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny);
                f32x4 t2 = d * d;
                f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
                if (!interior) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const bool valid = (i0 + 16 * wave + 4 * lg + r < n_rows) && (st * 64 + 16 * t + l15 < n_cols);
                        if (!valid) t2[r] = 0.f, tk[r] = 0.f;
                    }
                }
                p2 += t2;
                pk += tk;
            }
            s2 += (double)((p2[0] + p2[1]) + (p2[2] + p2[3]));
            skl += (double)((pk[0] + pk[1]) + (pk[2] + pk[3]));
        }
        mark(2);
        if (SHIFT && gb && st + 2 < st1) issue(st + 2);
        mark(3);
        if (++since_flush == FL) {
            since_flush = 0;
#pragma unroll
            for (int b = 0; b < NKQ; b++) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc64[b][r] += (double)accm[b][r] + (double)accx[b][r] * (1.0 / XPROD16_LO_SCALE);
                accm[b] = f32x4{0, 0, 0, 0};
                accx[b] = f32x4{0, 0, 0, 0};
            }
        }
    }
    const double unscale = ldexp(1.0, -(scal_exp[0] + scal_exp[1]));
    double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
    for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int kq = 16 * nt + l15;
            const int j = i0 + 16 * wave + 4 * lg + r;
            const double v = acc64[nt][r] + (double)accm[nt][r] + (double)accx[nt][r] * (1.0 / XPROD16_LO_SCALE);
            out[(size_t)kq * ldc + j] = v * unscale;
        }
    s2 = wave_sum(s2);
    skl = wave_sum(skl);
    __syncthreads(); // (every fragment read of the last stage is done: the ring becomes the reduction scratch)
    double *red = (double *)smem;
    if (lane == 0) red[wave] = s2, red[XPROD_WAVES + wave] = skl;
    __syncthreads();
    if (tid == 0) {
        double a2 = 0.0, ak = 0.0;
        for (int w = 0; w < XPROD_WAVES; w++) a2 += red[w], ak += red[XPROD_WAVES + w];
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = a2;
        partial[2 * blk + 1] = ak;
    }
    if constexpr (TIMING) {
        if (blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 40) && lane == 0) {
            unsigned long long *r = tim + ((blockIdx.x == 0 ? 0 : 40) + wave) * 16;
            for (int i = 0; i < 6; i++) r[i] = tacc[i];
            r[15] = (unsigned long long)(st1 - st0);
        }
    }
}

template <int SHIFT, int TIMING>
static int xerr4_launch(dim3 grid, const uint32_t *A16T, int mpad, const uint32_t *Y16, const uint32_t *H16c, const uint32_t *W16c, double *C, int npad, size_t slab,
                        int stages, int sps, const int *scal, int n, int m, double *P, unsigned long long *tim)
{
    const int lds = 5 * XE4_SLOT;
    if (hipFuncSetAttribute((const void *)xerr4_kernel<4, SHIFT, TIMING>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { printf("LDS attribute refused\n"); return 1; }
    xerr4_kernel<4, SHIFT, TIMING><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, H16c, W16c, C, npad, slab, 0, stages, sps, scal, scal + 2, n, m, P, tim);
    return hipGetLastError() != hipSuccess;
}

template <int NKQ, int SHIFT, int TIMING, int PAT>
__global__ __launch_bounds__(XPROD_THREADS) void xerr5_kernel(const uint32_t *__restrict__ A16, int lda, const uint32_t *__restrict__ Y16, int ldy,
                                                              const uint32_t *__restrict__ H16c, const uint32_t *__restrict__ W16c,
                                                              double *__restrict__ Cx, int ldc, size_t slab_stride, int stage_begin, int stage_end,
                                                              int stages_per_split, const int *__restrict__ scal_exp, const int *__restrict__ w_exp,
                                                              int n_rows, int n_cols, double *__restrict__ partial, unsigned long long *__restrict__ tim)
{
    static_assert(NKQ == 4, "experiment: rank 49..64");
    constexpr int FL = XPROD_FLUSH_ELEMS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int i0 = blockIdx.x * XPROD_TN_BJ;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;

    f32x4 accm[NKQ], accx[NKQ];
    f64x4 acc64[NKQ];
#pragma unroll
    for (int b = 0; b < NKQ; b++) {
        accm[b] = f32x4{0, 0, 0, 0};
        accx[b] = f32x4{0, 0, 0, 0};
        acc64[b] = f64x4{0, 0, 0, 0};
    }
    xh8 wh[2], wl[2];
    {
        const _Float16 *wrow = (const _Float16 *)(W16c + (size_t)(i0 + 16 * wave + l15) * 64);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            wh[c] = *(const xh8 *)(wrow + 32 * c + 8 * lg);
            wl[c] = *(const xh8 *)(wrow + 64 + 32 * c + 8 * lg);
        }
    }
    xh8 ident[2];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int e = 0; e < 8; e++) ident[u][e] = (8 * lg + e == 16 * u + l15) ? (_Float16)1.0f : (_Float16)0.0f;
    const float ca = ldexpf(1.0f, -scal_exp[0]);
    const float cwh = ldexpf(1.0f, -(w_exp[0] + scal_exp[1]));
    double s2 = 0.0, skl = 0.0;

    const bool gb = wave >= 4;
    auto slot_of = [&](int rel, bool b) -> unsigned char * { return b ? smem + 2 * XE4_SLOT + (rel % 3) * XE4_SLOT : smem + (rel & 1) * XE4_SLOT; };
    // piece t = wave + 8 i = rows 4 t + lg = rw + 32 i of an image: one per-lane byte offset per image, the rest scalar
    const int rw = 4 * wave + lg, sw = l15 ^ (rw & 15);
    const unsigned voffA = (unsigned)(((size_t)rw * lda + sw * 4) * 4), voffY = (unsigned)(((size_t)rw * ldy + sw * 4) * 4);
    const unsigned voffH = (unsigned)((rw * 64 + sw * 4) * 4);
    const unsigned long long baseA = xp_uniform64(A16 + (size_t)i0 * lda), baseY = xp_uniform64(Y16), baseH = xp_uniform64(H16c);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    auto issue = [&](int st) {
        const int rel = st - st0;
        const unsigned dst = lds0 + (gb ? 2u * XE4_SLOT + (unsigned)(rel % 3) * XE4_SLOT : (unsigned)(rel & 1) * XE4_SLOT) + (unsigned)(wave & 3) * 1024u;
        const unsigned long long c0b = (unsigned long long)st * 256ull;
#pragma unroll
        for (int i = 0; i < 4; i++) glds16_s(voffA, baseA + c0b + (unsigned long long)i * 32ull * (unsigned long long)lda * 4ull, dst + (unsigned)i * 4096u);
#pragma unroll
        for (int i = 0; i < 2; i++)
            glds16_s(voffY, baseY + c0b + (unsigned long long)i * 32ull * (unsigned long long)ldy * 4ull, dst + (unsigned)XE4_YOFF + (unsigned)i * 4096u);
#pragma unroll
        for (int i = 0; i < 2; i++)
            glds16_s(voffH, baseH + (unsigned long long)st * 64ull * 256ull + (unsigned long long)i * 32ull * 256ull, dst + (unsigned)XE4_HOFF + (unsigned)i * 4096u);
    };
    // per-lane byte offsets of the fragment reads inside an image row
    const int oh0 = ((0 + lg) ^ l15) * 16, oh1 = ((4 + lg) ^ l15) * 16, ol0 = ((8 + lg) ^ l15) * 16, ol1 = ((12 + lg) ^ l15) * 16;
    const float il = 1.0f / XPROD16_LO_SCALE, tiny = (float)NNLM_TINY;

    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int i) {
        if constexpr (TIMING) {
            unsigned long long tn_;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");
            tacc[i] += tn_ - tprev;
            tprev = tn_;
        }
    };
    if constexpr (TIMING) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");

    if (st0 < st1) issue(st0);
    if (SHIFT && gb && st0 + 1 < st1) issue(st0 + 1);
    int since_flush = 0;
    for (int st = st0; st < st1; ++st) {
        const int rel = st - st0;
        const unsigned char *bufe = slot_of(rel, false), *bufo = slot_of(rel, true);
        if (SHIFT && gb && st + 1 < st1) wait_vmcnt(8);
        else wait_vmcnt(0);
        mark(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        mark(1);
        if (st + 1 < st1 && (!SHIFT || !gb)) issue(st + 1);
        auto tile = [&](int T, int xoff) -> const unsigned char * { return ((T & 1) ? bufo : bufe) + xoff + (T >> 1) * 4096 + l15 * XPROD_ROWB; };
        const unsigned char *arowp = ((wave & 1) ? bufo : bufe) + (wave >> 1) * 4096 + l15 * XPROD_ROWB;
        const bool interior = (i0 + XPROD_TN_BJ <= n_rows) && (st * 64 + 64 <= n_cols);
        if (interior && PAT != 2) {
            xh8 ah[2], al[2];
            ah[0] = *(const xh8 *)(arowp + oh0), al[0] = *(const xh8 *)(arowp + ol0);
            ah[1] = *(const xh8 *)(arowp + oh1), al[1] = *(const xh8 *)(arowp + ol1);
            xh8 hh[2][2], hl[2][2], yh[2][2], yl[2][2];
            auto read_hy = [&](int t, int b) {
                const unsigned char *hrow = tile(t, XE4_HOFF);
                hh[b][0] = *(const xh8 *)(hrow + oh0), hl[b][0] = *(const xh8 *)(hrow + ol0);
                hh[b][1] = *(const xh8 *)(hrow + oh1), hl[b][1] = *(const xh8 *)(hrow + ol1);
                const unsigned char *yrow = tile(t, XE4_YOFF);
                yh[b][0] = *(const xh8 *)(yrow + oh0), yl[b][0] = *(const xh8 *)(yrow + ol0);
                yh[b][1] = *(const xh8 *)(yrow + oh1), yl[b][1] = *(const xh8 *)(yrow + ol1);
            };
            read_hy(0, 0);
            f32x4 dh[2], dl[2]; // (of the tile in work and of the one before it)
            __builtin_amdgcn_sched_barrier(0);
            mark(2);
            f32x4 em[2], ex[2];
            f32x4 p2 = f32x4{0, 0, 0, 0}, pk = f32x4{0, 0, 0, 0};
            auto epi = [&](int t, const f32x4 &em_, const f32x4 &ex_) {
                const f32x4 aa = (dh[t & 1] + dl[t & 1] * il) * ca;
                const f32x4 ah2 = (em_ + ex_ * il) * cwh;
                const f32x4 d = aa - ah2;
                f32x4 lg4;
#pragma unroll
                for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny);
                f32x4 t2 = d * d;
                f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
                p2 += t2;
                pk += tk;
            };
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int b = t & 1;
                if (t < 3) read_hy(t + 1, b ^ 1);
                if (t > 0) epi(t - 1, em[b ^ 1], ex[b ^ 1]);
                dh[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t >> 1], ident[t & 1], f32x4{0, 0, 0, 0}, 0, 0, 0);
                dl[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t >> 1], ident[t & 1], f32x4{0, 0, 0, 0}, 0, 0, 0);
                em[b] = f32x4{0, 0, 0, 0}, ex[b] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int c2 = 0; c2 < 2; c2++) {
                    em[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hh[b][c2], em[b], 0, 0, 0);
                    ex[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hl[b][c2], ex[b], 0, 0, 0);
                    ex[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c2], hh[b][c2], ex[b], 0, 0, 0);
                }
#pragma unroll
                for (int c2 = 0; c2 < 2; c2++) {
                    accm[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c2], yh[b][c2], accm[t], 0, 0, 0);
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c2], yl[b][c2], accx[t], 0, 0, 0);
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c2], yh[b][c2], accx[t], 0, 0, 0);
                }
                if (PAT == 1) {
                    if (t < 3) XE_SGB(0x100, 8);
#pragma unroll
                    for (int i = 0; i < 12; i++) {
                        XE_SGB(0x8, 1);
                        if (t > 0) XE_SGB(0x402, 2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            mark(3);
            epi(3, em[1], ex[1]);
            s2 += (double)((p2[0] + p2[1]) + (p2[2] + p2[3]));
            skl += (double)((pk[0] + pk[1]) + (pk[2] + pk[3]));
            mark(4);
        } else {
            f32x4 em[4], ex[4], dh[4], dl[4];
#pragma unroll
            for (int t = 0; t < 4; t++) em[t] = f32x4{0, 0, 0, 0}, ex[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++) {
                const int sh = (4 * c2 + lg), sl = 8 + 4 * c2 + lg;
                const xh8 ah = *(const xh8 *)(arowp + ((sh ^ l15) * 16));
                const xh8 al = *(const xh8 *)(arowp + ((sl ^ l15) * 16));
#pragma unroll
                for (int nt = 0; nt < NKQ; nt++) {
                    const unsigned char *yrow = tile(nt, XE4_YOFF);
                    const xh8 yh = *(const xh8 *)(yrow + ((sh ^ l15) * 16));
                    const xh8 yl = *(const xh8 *)(yrow + ((sl ^ l15) * 16));
                    accm[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yh, accm[nt], 0, 0, 0);
                    accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yl, accx[nt], 0, 0, 0);
                    accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, yh, accx[nt], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    dh[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
                    dl[2 * c2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ident[u], f32x4{0, 0, 0, 0}, 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const unsigned char *hrow = tile(t, XE4_HOFF);
                    const xh8 hh = *(const xh8 *)(hrow + ((sh ^ l15) * 16));
                    const xh8 hl = *(const xh8 *)(hrow + ((sl ^ l15) * 16));
                    em[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hh, em[t], 0, 0, 0);
                    ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hl, ex[t], 0, 0, 0);
                    ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c2], hh, ex[t], 0, 0, 0);
                }
            }
            f32x4 p2 = f32x4{0, 0, 0, 0}, pk = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const f32x4 aa = (dh[t] + dl[t] * il) * ca;
                const f32x4 ah2 = (em[t] + ex[t] * il) * cwh;
                const f32x4 d = aa - ah2;
                f32x4 lg4;
#pragma unroll
                for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny);
                f32x4 t2 = d * d;
                f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool valid = (i0 + 16 * wave + 4 * lg + r < n_rows) && (st * 64 + 16 * t + l15 < n_cols);
                    if (!valid) t2[r] = 0.f, tk[r] = 0.f;
                }
                p2 += t2;
                pk += tk;
            }
            s2 += (double)((p2[0] + p2[1]) + (p2[2] + p2[3]));
            skl += (double)((pk[0] + pk[1]) + (pk[2] + pk[3]));
        }
        if (SHIFT && gb && st + 2 < st1) issue(st + 2);
        if (++since_flush == FL) {
            since_flush = 0;
#pragma unroll
            for (int b = 0; b < NKQ; b++) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc64[b][r] += (double)accm[b][r] + (double)accx[b][r] * (1.0 / XPROD16_LO_SCALE);
                accm[b] = f32x4{0, 0, 0, 0};
                accx[b] = f32x4{0, 0, 0, 0};
            }
        }
        mark(5);
    }
    const double unscale = ldexp(1.0, -(scal_exp[0] + scal_exp[1]));
    double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
    for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int kq = 16 * nt + l15;
            const int j = i0 + 16 * wave + 4 * lg + r;
            const double v = acc64[nt][r] + (double)accm[nt][r] + (double)accx[nt][r] * (1.0 / XPROD16_LO_SCALE);
            out[(size_t)kq * ldc + j] = v * unscale;
        }
    s2 = wave_sum(s2);
    skl = wave_sum(skl);
    __syncthreads();
    double *red = (double *)smem;
    if (lane == 0) red[wave] = s2, red[XPROD_WAVES + wave] = skl;
    __syncthreads();
    if (tid == 0) {
        double a2 = 0.0, ak = 0.0;
        for (int w = 0; w < XPROD_WAVES; w++) a2 += red[w], ak += red[XPROD_WAVES + w];
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = a2;
        partial[2 * blk + 1] = ak;
    }
    if constexpr (TIMING) {
        if (blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 40) && lane == 0) {
            unsigned long long *r = tim + ((blockIdx.x == 0 ? 0 : 40) + wave) * 16;
            for (int i = 0; i < 6; i++) r[i] = tacc[i];
            r[15] = (unsigned long long)(st1 - st0);
        }
    }
}

template <int SHIFT, int TIMING, int PAT>
static int xerr5_launch(dim3 grid, const uint32_t *A16T, int mpad, const uint32_t *Y16, const uint32_t *H16c, const uint32_t *W16c, double *C, int npad, size_t slab,
                        int stages, int sps, const int *scal, int n, int m, double *P, unsigned long long *tim)
{
    const int lds = 5 * XE4_SLOT;
    if (hipFuncSetAttribute((const void *)xerr5_kernel<4, SHIFT, TIMING, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { printf("LDS attribute refused\n"); return 1; }
    xerr5_kernel<4, SHIFT, TIMING, PAT><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, H16c, W16c, C, npad, slab, 0, stages, sps, scal, scal + 2, n, m, P, tim);
    return hipGetLastError() != hipSuccess;
}

// xerr6_kernel: wavefront-specialised form.  Same images and ring as the product (2 x 64 KB), same arithmetic and accumulation orders
// (cross product and error sums bit-identical), but a wavefront owns 32 rows and ONE of the two jobs:
//   * wavefronts 0..3 ("E"): W H and the error arithmetic of rows 32 rg .. 32 rg + 31 x the stage's 64 columns -- W fragments of two
//     M-tiles in registers, every H fragment read once per 32 rows (the product: once per 16), the arithmetic of tile t in the shadow of
//     the MFMAs of tile t + 1;
//   * wavefronts 4..7 ("X"): the cross product of the same rows (every Y fragment read once per 32 rows) and ALL of the block's requests:
//     while they sit in the memory pipeline's queue, the E wavefront of their SIMD computes.
// LDS fragment reads per stage and CU: 4 x (8 + 16) + 4 x (8 + 16) = 192 KB (the product: 8 x 44 = 352 KB).
template <int NKQ, int ISPLIT, int TIMING>
__global__ __launch_bounds__(XPROD_THREADS) void xerr6_kernel(const uint32_t *__restrict__ A16, int lda, const uint32_t *__restrict__ Y16, int ldy,
                                                              const uint32_t *__restrict__ H16c, const uint32_t *__restrict__ W16c,
                                                              double *__restrict__ Cx, int ldc, size_t slab_stride, int stage_begin, int stage_end,
                                                              int stages_per_split, const int *__restrict__ scal_exp, const int *__restrict__ w_exp,
                                                              int n_rows, int n_cols, double *__restrict__ partial, unsigned long long *__restrict__ tim)
{
    constexpr int FL = XPROD_FLUSH_ELEMS / 64;
    constexpr int YOFF = XPROD_A_IMG_BYTES, HOFF = XPROD_A_IMG_BYTES + 64 * XPROD_ROWB;
    constexpr int YI = 4 * NKQ; // 4-row pieces of the factor image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[2][XPROD_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int rg = wave & 3; // rows 32 rg .. 32 rg + 31 of the block = the product's wavefronts (row tiles) 2 rg and 2 rg + 1
    const bool xrole = wave >= 4;
    const int i0 = blockIdx.x * XPROD_TN_BJ;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;
    const int oh0 = ((0 + lg) ^ l15) * 16, oh1 = ((4 + lg) ^ l15) * 16, ol0 = ((8 + lg) ^ l15) * 16, ol1 = ((12 + lg) ^ l15) * 16;

    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int i) {
        if constexpr (TIMING) {
            unsigned long long tn_;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");
            tacc[i] += tn_ - tprev;
            tprev = tn_;
        }
    };
    if constexpr (TIMING) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");

    if (xrole) {
        // ---------------------------------------------------------------- cross product + requests
        f32x4 accm[2][NKQ], accx[2][NKQ];
        f64x4 acc64[2][NKQ];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int b = 0; b < NKQ; b++) {
                accm[mt][b] = f32x4{0, 0, 0, 0};
                accx[mt][b] = f32x4{0, 0, 0, 0};
                acc64[mt][b] = f64x4{0, 0, 0, 0};
            }
        // piece t = rg + 4 i of an image = rows 4 t + lg = rw + 16 i: (row & 15) = rw & 15 for every i
        const int rw = 4 * rg + lg, sw = l15 ^ (rw & 15);
        const unsigned voffA = (unsigned)(((size_t)rw * lda + sw * 4) * 4), voffY = (unsigned)(((size_t)rw * ldy + sw * 4) * 4);
        const unsigned voffH = (unsigned)((rw * 64 + sw * 4) * 4);
        const unsigned long long baseA = xp_uniform64(A16 + (size_t)i0 * lda), baseY = xp_uniform64(Y16), baseH = xp_uniform64(H16c);
        const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
        auto issue_a = [&](int st, int bi, int ia, int ib) { // A pieces i in [ia, ib) of this wavefront's eight
            const unsigned long long c0b = (unsigned long long)st * 256ull;
            const unsigned dst = lds0 + (unsigned)bi * (unsigned)XPROD16_ERR_BUF + (unsigned)rg * 1024u;
#pragma unroll
            for (int i = ia; i < ib; i++) glds16_s(voffA, baseA + c0b + (unsigned long long)i * 16ull * (unsigned long long)lda * 4ull, dst + (unsigned)i * 4096u);
        };
        auto issue_f = [&](int st, int bi) { // the factor images: Y rows and H columns
            const unsigned long long c0b = (unsigned long long)st * 256ull;
            const unsigned dst = lds0 + (unsigned)bi * (unsigned)XPROD16_ERR_BUF + (unsigned)rg * 1024u;
#pragma unroll
            for (int i = 0; i < NKQ; i++)
                glds16_s(voffY, baseY + c0b + (unsigned long long)i * 16ull * (unsigned long long)ldy * 4ull, dst + (unsigned)YOFF + (unsigned)i * 4096u);
#pragma unroll
            for (int i = 0; i < 4; i++)
                glds16_s(voffH, baseH + (unsigned long long)st * 64ull * 256ull + (unsigned long long)i * 16ull * 256ull, dst + (unsigned)HOFF + (unsigned)i * 4096u);
        };
        if (st0 < st1) { issue_f(st0, 0); issue_a(st0, 0, 0, 8); }
        int since_flush = 0;
        for (int st = st0; st < st1; ++st) {
            const unsigned char *buf = smem + ((st - st0) & 1) * XPROD16_ERR_BUF;
            const int nbi = (st + 1 - st0) & 1;
            const bool more = st + 1 < st1;
            wait_vmcnt(0);
            mark(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            mark(1);
            if (more) {
                issue_f(st + 1, nbi);
                issue_a(st + 1, nbi, 0, ISPLIT ? 4 : 8);
            }
            mark(2);
            const unsigned char *arow0 = buf + (32 * rg + l15) * XPROD_ROWB, *arow1 = arow0 + 16 * XPROD_ROWB;
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++) {
                const int oh = c2 ? oh1 : oh0, ol = c2 ? ol1 : ol0;
                const xh8 ah0 = *(const xh8 *)(arow0 + oh), al0 = *(const xh8 *)(arow0 + ol);
                const xh8 ah1 = *(const xh8 *)(arow1 + oh), al1 = *(const xh8 *)(arow1 + ol);
#pragma unroll
                for (int nt = 0; nt < NKQ; nt++) {
                    const unsigned char *yrow = buf + YOFF + (16 * nt + l15) * XPROD_ROWB;
                    const xh8 yh = *(const xh8 *)(yrow + oh);
                    const xh8 yl = *(const xh8 *)(yrow + ol);
                    accm[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, yh, accm[0][nt], 0, 0, 0);
                    accx[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, yl, accx[0][nt], 0, 0, 0);
                    accx[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, yh, accx[0][nt], 0, 0, 0);
                    accm[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, yh, accm[1][nt], 0, 0, 0);
                    accx[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, yl, accx[1][nt], 0, 0, 0);
                    accx[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, yh, accx[1][nt], 0, 0, 0);
                }
                if (ISPLIT && c2 == 0 && more) issue_a(st + 1, nbi, 4, 8);
            }
            mark(3);
            if (++since_flush == FL) {
                since_flush = 0;
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int b = 0; b < NKQ; b++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) acc64[mt][b][r] += (double)accm[mt][b][r] + (double)accx[mt][b][r] * (1.0 / XPROD16_LO_SCALE);
                        accm[mt][b] = f32x4{0, 0, 0, 0};
                        accx[mt][b] = f32x4{0, 0, 0, 0};
                    }
            }
            mark(4);
        }
        const double unscale = ldexp(1.0, -(scal_exp[0] + scal_exp[1]));
        double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kq = 16 * nt + l15;
                    const int j = i0 + 32 * rg + 16 * mt + 4 * lg + r;
                    const double v = acc64[mt][nt][r] + (double)accm[mt][nt][r] + (double)accx[mt][nt][r] * (1.0 / XPROD16_LO_SCALE);
                    out[(size_t)kq * ldc + j] = v * unscale;
                }
    } else {
        // ---------------------------------------------------------------- W H and the two error sums
        xh8 wh[2][2], wl[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const _Float16 *wrow = (const _Float16 *)(W16c + (size_t)(i0 + 32 * rg + 16 * mt + l15) * 64);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                wh[mt][c] = *(const xh8 *)(wrow + 32 * c + 8 * lg);
                wl[mt][c] = *(const xh8 *)(wrow + 64 + 32 * c + 8 * lg);
            }
        }
        xh8 ident[2];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int e = 0; e < 8; e++) ident[u][e] = (8 * lg + e == 16 * u + l15) ? (_Float16)1.0f : (_Float16)0.0f;
        const float ca = ldexpf(1.0f, -scal_exp[0]);
        const float cwh = ldexpf(1.0f, -(w_exp[0] + scal_exp[1]));
        const float il = 1.0f / XPROD16_LO_SCALE, tiny = (float)NNLM_TINY;
        double s2[2] = {0.0, 0.0}, skl[2] = {0.0, 0.0};
        for (int st = st0; st < st1; ++st) {
            const unsigned char *buf = smem + ((st - st0) & 1) * XPROD16_ERR_BUF;
            mark(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            mark(1);
            const bool interior = (i0 + XPROD_TN_BJ <= n_rows) && (st * 64 + 64 <= n_cols);
            auto stage_body = [&](auto interior_c) { // (two copies: the interior one is a single basic block)
            constexpr bool INTERIOR = decltype(interior_c)::value;
            const unsigned char *arow0 = buf + (32 * rg + l15) * XPROD_ROWB, *arow1 = arow0 + 16 * XPROD_ROWB;
            xh8 ah[2][2], al[2][2]; // [M-tile][K chunk]
            ah[0][0] = *(const xh8 *)(arow0 + oh0), al[0][0] = *(const xh8 *)(arow0 + ol0);
            ah[1][0] = *(const xh8 *)(arow1 + oh0), al[1][0] = *(const xh8 *)(arow1 + ol0);
            ah[0][1] = *(const xh8 *)(arow0 + oh1), al[0][1] = *(const xh8 *)(arow0 + ol1);
            ah[1][1] = *(const xh8 *)(arow1 + oh1), al[1][1] = *(const xh8 *)(arow1 + ol1);
            xh8 hh[2][2], hl[2][2]; // [buffer][K chunk]
            auto read_h = [&](int t, int b) {
                const unsigned char *hrow = buf + HOFF + (16 * t + l15) * XPROD_ROWB;
                hh[b][0] = *(const xh8 *)(hrow + oh0), hl[b][0] = *(const xh8 *)(hrow + ol0);
                hh[b][1] = *(const xh8 *)(hrow + oh1), hl[b][1] = *(const xh8 *)(hrow + ol1);
            };
            read_h(0, 0);
            f32x4 em[2][2], ex[2][2], dh[2][2], dl[2][2]; // [buffer][M-tile]
            f32x4 p2[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, pk[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
            auto epi = [&](int t, int b) {
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    const f32x4 aa = (dh[b][mt] + dl[b][mt] * il) * ca;
                    const f32x4 ah2 = (em[b][mt] + ex[b][mt] * il) * cwh;
                    const f32x4 d = aa - ah2;
                    f32x4 lg4;
#pragma unroll
                    for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny);
                    f32x4 t2 = d * d;
                    f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
                    if constexpr (!INTERIOR) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const bool valid = (i0 + 32 * rg + 16 * mt + 4 * lg + r < n_rows) && (st * 64 + 16 * t + l15 < n_cols);
                            if (!valid) t2[r] = 0.f, tk[r] = 0.f;
                        }
                    }
                    p2[mt] += t2;
                    pk[mt] += tk;
                }
            };
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int b = t & 1;
                if (t < 3) read_h(t + 1, b ^ 1);
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    dh[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][t >> 1], ident[t & 1], f32x4{0, 0, 0, 0}, 0, 0, 0);
                    dl[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt][t >> 1], ident[t & 1], f32x4{0, 0, 0, 0}, 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    em[b][mt] = f32x4{0, 0, 0, 0}, ex[b][mt] = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int c2 = 0; c2 < 2; c2++) {
                        em[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt][c2], hh[b][c2], em[b][mt], 0, 0, 0);
                        ex[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt][c2], hl[b][c2], ex[b][mt], 0, 0, 0);
                        ex[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[mt][c2], hh[b][c2], ex[b][mt], 0, 0, 0);
                    }
                }
                if (t > 0) epi(t - 1, b ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            mark(2);
            epi(3, 1);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                s2[mt] += (double)((p2[mt][0] + p2[mt][1]) + (p2[mt][2] + p2[mt][3]));
                skl[mt] += (double)((pk[mt][0] + pk[mt][1]) + (pk[mt][2] + pk[mt][3]));
            }
            };
            if (interior) stage_body(std::true_type{});
            else stage_body(std::false_type{});
            mark(3);
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const double a = wave_sum(s2[mt]), b = wave_sum(skl[mt]);
            if (lane == 0) red[0][2 * rg + mt] = a, red[1][2 * rg + mt] = b;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double a2 = 0.0, ak = 0.0;
        for (int w = 0; w < XPROD_WAVES; w++) a2 += red[0][w], ak += red[1][w];
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = a2;
        partial[2 * blk + 1] = ak;
    }
    if constexpr (TIMING) {
        if (blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 40) && lane == 0) {
            unsigned long long *r = tim + ((blockIdx.x == 0 ? 0 : 40) + wave) * 16;
            for (int i = 0; i < 6; i++) r[i] = tacc[i];
            r[15] = (unsigned long long)(st1 - st0);
        }
    }
}

template <int ISPLIT, int TIMING>
static int xerr6_launch(dim3 grid, const uint32_t *A16T, int mpad, const uint32_t *Y16, const uint32_t *H16c, const uint32_t *W16c, double *C, int npad, size_t slab,
                        int stages, int sps, const int *scal, int n, int m, double *P, unsigned long long *tim)
{
    const int lds = 2 * XPROD16_ERR_BUF;
    if (hipFuncSetAttribute((const void *)xerr6_kernel<4, ISPLIT, TIMING>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { printf("LDS attribute refused\n"); return 1; }
    xerr6_kernel<4, ISPLIT, TIMING><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, H16c, W16c, C, npad, slab, 0, stages, sps, scal, scal + 2, n, m, P, tim);
    return hipGetLastError() != hipSuccess;
}

// xerr7_kernel: xerr6 with a ring of THREE A images (the HBM stream: two stages in flight) and two factor-image pairs -- 96 + 64 = 160 KB,
// the reduction scratch aliases the ring.  (xerr6:) wavefront-specialised form.  Same images and ring as the product (2 x 64 KB), same arithmetic and accumulation orders
// (cross product and error sums bit-identical), but a wavefront owns 32 rows and ONE of the two jobs:
//   * wavefronts 0..3 ("E"): W H and the error arithmetic of rows 32 rg .. 32 rg + 31 x the stage's 64 columns -- W fragments of two
//     M-tiles in registers, every H fragment read once per 32 rows (the product: once per 16), the arithmetic of tile t in the shadow of
//     the MFMAs of tile t + 1;
//   * wavefronts 4..7 ("X"): the cross product of the same rows (every Y fragment read once per 32 rows) and ALL of the block's requests:
//     while they sit in the memory pipeline's queue, the E wavefront of their SIMD computes.
// LDS fragment reads per stage and CU: 4 x (8 + 16) + 4 x (8 + 16) = 192 KB (the product: 8 x 44 = 352 KB).
template <int NKQ, int ABL, int TIMING>
__global__ __launch_bounds__(XPROD_THREADS) void xerr7_kernel(const uint32_t *__restrict__ A16, int lda, const uint32_t *__restrict__ Y16, int ldy,
                                                              const uint32_t *__restrict__ H16c, const uint32_t *__restrict__ W16c,
                                                              double *__restrict__ Cx, int ldc, size_t slab_stride, int stage_begin, int stage_end,
                                                              int stages_per_split, const int *__restrict__ scal_exp, const int *__restrict__ w_exp,
                                                              int n_rows, int n_cols, double *__restrict__ partial, unsigned long long *__restrict__ tim)
{
    constexpr int FL = XPROD_FLUSH_ELEMS / 64;
    constexpr int FOFF = 3 * XPROD_A_IMG_BYTES, FBUF = 128 * XPROD_ROWB; // factor images behind the three A images: [2][Y 16 KB | H 16 KB]
    constexpr int YOFF = 0, HOFF = 64 * XPROD_ROWB;
    constexpr int YI = 4 * NKQ; // 4-row pieces of the factor image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int rg = wave & 3; // rows 32 rg .. 32 rg + 31 of the block = the product's wavefronts (row tiles) 2 rg and 2 rg + 1
    const bool xrole = (ABL & 16) ? wave < 4 : wave >= 4;
    const int i0 = blockIdx.x * XPROD_TN_BJ;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;
    double *red = (double *)smem;
    const int oh0 = ((0 + lg) ^ l15) * 16, oh1 = ((4 + lg) ^ l15) * 16, ol0 = ((8 + lg) ^ l15) * 16, ol1 = ((12 + lg) ^ l15) * 16;

    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int i) {
        if constexpr (TIMING) {
            unsigned long long tn_;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");
            tacc[i] += tn_ - tprev;
            tprev = tn_;
        }
    };
    if constexpr (TIMING) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");

    if (xrole) {
        // ---------------------------------------------------------------- cross product + requests
        f32x4 accm[2][NKQ], accx[2][NKQ];
        f64x4 acc64[2][NKQ];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int b = 0; b < NKQ; b++) {
                accm[mt][b] = f32x4{0, 0, 0, 0};
                accx[mt][b] = f32x4{0, 0, 0, 0};
                acc64[mt][b] = f64x4{0, 0, 0, 0};
            }
        // piece t = rg + 4 i of an image = rows 4 t + lg = rw + 16 i: (row & 15) = rw & 15 for every i
        const int rw = 4 * rg + lg, sw = l15 ^ (rw & 15);
        const unsigned voffA = (unsigned)(((size_t)rw * lda + sw * 4) * 4), voffY = (unsigned)(((size_t)rw * ldy + sw * 4) * 4);
        const unsigned voffH = (unsigned)((rw * 64 + sw * 4) * 4);
        const unsigned long long baseA = xp_uniform64(A16 + (size_t)i0 * lda), baseY = xp_uniform64(Y16), baseH = xp_uniform64(H16c);
        const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
        auto issue_a = [&](int st, int ia, int ib) { // A pieces i in [ia, ib) of this wavefront's eight
            const unsigned long long c0b = (unsigned long long)st * 256ull;
            const unsigned dst = lds0 + (unsigned)((st - st0) % 3) * (unsigned)XPROD_A_IMG_BYTES + (unsigned)rg * 1024u;
#pragma unroll
            for (int i = ia; i < ib; i++) glds16_s(voffA, baseA + c0b + (unsigned long long)i * 16ull * (unsigned long long)lda * 4ull, dst + (unsigned)i * 4096u);
        };
        auto issue_f = [&](int st) { // the factor images: Y rows and H columns
            const unsigned long long c0b = (unsigned long long)st * 256ull;
            const unsigned dst = lds0 + (unsigned)FOFF + (unsigned)((st - st0) & 1) * (unsigned)FBUF + (unsigned)rg * 1024u;
#pragma unroll
            for (int i = 0; i < NKQ; i++)
                glds16_s(voffY, baseY + c0b + (unsigned long long)i * 16ull * (unsigned long long)ldy * 4ull, dst + (unsigned)YOFF + (unsigned)i * 4096u);
#pragma unroll
            for (int i = 0; i < 4; i++)
                glds16_s(voffH, baseH + (unsigned long long)st * 64ull * 256ull + (unsigned long long)i * 16ull * 256ull, dst + (unsigned)HOFF + (unsigned)i * 4096u);
        };
        if (st0 < st1) { issue_f(st0); issue_a(st0, 0, 8); }
        if (st0 + 1 < st1) issue_a(st0 + 1, 0, 8);
        int since_flush = 0;
        for (int st = st0; st < st1; ++st) {
            const unsigned char *abuf = smem + ((st - st0) % 3) * XPROD_A_IMG_BYTES, *fbuf = smem + FOFF + ((st - st0) & 1) * FBUF;
            const bool more = st + 1 < st1;
            if (st + 1 < st1) wait_vmcnt(8); // (issued since this stage's images: the A image of the next stage)
            else wait_vmcnt(0);
            mark(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            mark(1);
            if (more) issue_f(st + 1);
            if (st + 2 < st1) issue_a(st + 2, 0, 8);
            else if (more) { // keep the count of requests per stage: repeat the factor images (idempotent)
                issue_f(st + 1);
            }
            mark(2);
            const unsigned char *arow0 = abuf + (32 * rg + l15) * XPROD_ROWB, *arow1 = arow0 + 16 * XPROD_ROWB;
#pragma unroll
            for (int c2 = 0; c2 < ((ABL & 1) ? 0 : 2); c2++) {
                const int oh = c2 ? oh1 : oh0, ol = c2 ? ol1 : ol0;
                const xh8 ah0 = *(const xh8 *)(arow0 + oh), al0 = *(const xh8 *)(arow0 + ol);
                const xh8 ah1 = *(const xh8 *)(arow1 + oh), al1 = *(const xh8 *)(arow1 + ol);
#pragma unroll
                for (int nt = 0; nt < NKQ; nt++) {
                    const unsigned char *yrow = fbuf + YOFF + (16 * nt + l15) * XPROD_ROWB;
                    const xh8 yh = *(const xh8 *)(yrow + oh);
                    const xh8 yl = *(const xh8 *)(yrow + ol);
                    accm[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, yh, accm[0][nt], 0, 0, 0);
                    accx[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, yl, accx[0][nt], 0, 0, 0);
                    accx[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, yh, accx[0][nt], 0, 0, 0);
                    accm[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, yh, accm[1][nt], 0, 0, 0);
                    accx[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, yl, accx[1][nt], 0, 0, 0);
                    accx[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, yh, accx[1][nt], 0, 0, 0);
                }
            }
            mark(3);
            if (++since_flush == FL) {
                since_flush = 0;
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int b = 0; b < NKQ; b++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) acc64[mt][b][r] += (double)accm[mt][b][r] + (double)accx[mt][b][r] * (1.0 / XPROD16_LO_SCALE);
                        accm[mt][b] = f32x4{0, 0, 0, 0};
                        accx[mt][b] = f32x4{0, 0, 0, 0};
                    }
            }
            mark(4);
        }
        const double unscale = ldexp(1.0, -(scal_exp[0] + scal_exp[1]));
        double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kq = 16 * nt + l15;
                    const int j = i0 + 32 * rg + 16 * mt + 4 * lg + r;
                    const double v = acc64[mt][nt][r] + (double)accm[mt][nt][r] + (double)accx[mt][nt][r] * (1.0 / XPROD16_LO_SCALE);
                    out[(size_t)kq * ldc + j] = v * unscale;
                }
        __syncthreads();
    } else {
        // ---------------------------------------------------------------- W H and the two error sums
        xh8 wh[2][2], wl[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const _Float16 *wrow = (const _Float16 *)(W16c + (size_t)(i0 + 32 * rg + 16 * mt + l15) * 64);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                wh[mt][c] = *(const xh8 *)(wrow + 32 * c + 8 * lg);
                wl[mt][c] = *(const xh8 *)(wrow + 64 + 32 * c + 8 * lg);
            }
        }
        xh8 ident[2];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int e = 0; e < 8; e++) ident[u][e] = (8 * lg + e == 16 * u + l15) ? (_Float16)1.0f : (_Float16)0.0f;
        xh8 identm; // K slot 8 lg + e: 1 at slot n (hi half of column n of the tile), 2^-11 at slot 16 + n (its lo half), n = l15
#pragma unroll
        for (int e = 0; e < 8; e++) identm[e] = (8 * lg + e == l15) ? (_Float16)1.0f : (8 * lg + e == 16 + l15) ? (_Float16)(1.0f / XPROD16_LO_SCALE) : (_Float16)0.0f;
        const float ca = ldexpf(1.0f, -scal_exp[0]);
        const float cwh = ldexpf(1.0f, -(w_exp[0] + scal_exp[1]));
        const float il = 1.0f / XPROD16_LO_SCALE, tiny = (float)NNLM_TINY;
        double s2[2] = {0.0, 0.0}, skl[2] = {0.0, 0.0};
        for (int st = st0; st < st1; ++st) {
            const unsigned char *abuf = smem + ((st - st0) % 3) * XPROD_A_IMG_BYTES, *fbuf = smem + FOFF + ((st - st0) & 1) * FBUF;
            mark(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            mark(1);
            const bool interior = (i0 + XPROD_TN_BJ <= n_rows) && (st * 64 + 64 <= n_cols);
            auto stage_body = [&](auto interior_c) { // (two copies: the interior one is a single basic block)
            constexpr bool INTERIOR = decltype(interior_c)::value;
            const unsigned char *arow0 = abuf + (32 * rg + l15) * XPROD_ROWB, *arow1 = arow0 + 16 * XPROD_ROWB;
            xh8 ah[2][2], al[2][2]; // [M-tile][K chunk]
            xh8 am[2][4];           // (ABL 32) [M-tile][column tile]: K slots 0..15 = hi halves of the tile's 16 columns, 16..31 = their lo halves
            if (ABL & 32) {
                const int bs = (lg < 2 ? 0 : 8) + (lg & 1);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    am[0][t] = *(const xh8 *)(arow0 + (((bs + 2 * t) ^ l15) * 16));
                    am[1][t] = *(const xh8 *)(arow1 + (((bs + 2 * t) ^ l15) * 16));
                }
            } else {
            ah[0][0] = *(const xh8 *)(arow0 + oh0), al[0][0] = *(const xh8 *)(arow0 + ol0);
            ah[1][0] = *(const xh8 *)(arow1 + oh0), al[1][0] = *(const xh8 *)(arow1 + ol0);
            ah[0][1] = *(const xh8 *)(arow0 + oh1), al[0][1] = *(const xh8 *)(arow0 + ol1);
            ah[1][1] = *(const xh8 *)(arow1 + oh1), al[1][1] = *(const xh8 *)(arow1 + ol1);
            }
            xh8 hh[2][2], hl[2][2]; // [buffer][K chunk]
            auto read_h = [&](int t, int b) {
                const unsigned char *hrow = fbuf + HOFF + (16 * t + l15) * XPROD_ROWB;
                hh[b][0] = *(const xh8 *)(hrow + oh0), hl[b][0] = *(const xh8 *)(hrow + ol0);
                hh[b][1] = *(const xh8 *)(hrow + oh1), hl[b][1] = *(const xh8 *)(hrow + ol1);
            };
            read_h(0, 0);
            f32x4 em[2][2], ex[2][2], dh[2][2], dl[2][2]; // [buffer][M-tile]
            f32x4 p2[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, pk[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
            auto epi = [&](int t, int b) {
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    const f32x4 aa = (ABL & 32) ? dh[b][mt] * ca : (dh[b][mt] + dl[b][mt] * il) * ca;
                    const f32x4 ah2 = (em[b][mt] + ex[b][mt] * il) * cwh;
                    const f32x4 d = aa - ah2;
                    f32x4 lg4;
#pragma unroll
                    for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny);
                    f32x4 t2 = d * d;
                    f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
                    if constexpr (!INTERIOR) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const bool valid = (i0 + 32 * rg + 16 * mt + 4 * lg + r < n_rows) && (st * 64 + 16 * t + l15 < n_cols);
                            if (!valid) t2[r] = 0.f, tk[r] = 0.f;
                        }
                    }
                    p2[mt] += t2;
                    pk[mt] += tk;
                }
            };
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int b = t & 1;
                if (t < 3) read_h(t + 1, b ^ 1);
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    if (ABL & 32) dh[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[mt][t], identm, f32x4{0, 0, 0, 0}, 0, 0, 0);
                    else {
                    dh[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][t >> 1], ident[t & 1], f32x4{0, 0, 0, 0}, 0, 0, 0);
                    dl[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt][t >> 1], ident[t & 1], f32x4{0, 0, 0, 0}, 0, 0, 0);
                    }
                }
                if (ABL & 8) { // the two M-tiles alternate: no MFMA directly behind one on the same accumulator
#pragma unroll
                    for (int mt = 0; mt < 2; mt++) em[b][mt] = f32x4{0, 0, 0, 0}, ex[b][mt] = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int c2 = 0; c2 < 2; c2++) {
                        em[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[0][c2], hh[b][c2], em[b][0], 0, 0, 0);
                        em[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[1][c2], hh[b][c2], em[b][1], 0, 0, 0);
                        ex[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[0][c2], hl[b][c2], ex[b][0], 0, 0, 0);
                        ex[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[1][c2], hl[b][c2], ex[b][1], 0, 0, 0);
                        ex[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[0][c2], hh[b][c2], ex[b][0], 0, 0, 0);
                        ex[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[1][c2], hh[b][c2], ex[b][1], 0, 0, 0);
                    }
                } else {
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    em[b][mt] = f32x4{0, 0, 0, 0}, ex[b][mt] = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int c2 = 0; c2 < ((ABL & 4) ? 0 : 2); c2++) {
                        em[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt][c2], hh[b][c2], em[b][mt], 0, 0, 0);
                        ex[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt][c2], hl[b][c2], ex[b][mt], 0, 0, 0);
                        ex[b][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[mt][c2], hh[b][c2], ex[b][mt], 0, 0, 0);
                    }
                }
                }
                if (t > 0) { if (ABL & 2) { p2[0] += em[b ^ 1][0] + ex[b ^ 1][0] + dh[b ^ 1][0] + dl[b ^ 1][0]; p2[1] += em[b ^ 1][1] + ex[b ^ 1][1] + dh[b ^ 1][1] + dl[b ^ 1][1]; } else epi(t - 1, b ^ 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            mark(2);
            if (ABL & 2) { p2[0] += em[1][0] + ex[1][0] + dh[1][0] + dl[1][0]; p2[1] += em[1][1] + ex[1][1] + dh[1][1] + dl[1][1]; } else epi(3, 1);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                s2[mt] += (double)((p2[mt][0] + p2[mt][1]) + (p2[mt][2] + p2[mt][3]));
                skl[mt] += (double)((pk[mt][0] + pk[mt][1]) + (pk[mt][2] + pk[mt][3]));
            }
            };
            if (interior) stage_body(std::true_type{});
            else stage_body(std::false_type{});
            mark(3);
        }
        __syncthreads(); // (every fragment read is done: the ring becomes the reduction scratch)
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const double a = wave_sum(s2[mt]), b = wave_sum(skl[mt]);
            if (lane == 0) red[2 * rg + mt] = a, red[XPROD_WAVES + 2 * rg + mt] = b;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double a2 = 0.0, ak = 0.0;
        for (int w = 0; w < XPROD_WAVES; w++) a2 += red[w], ak += red[XPROD_WAVES + w];
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = a2;
        partial[2 * blk + 1] = ak;
    }
    if constexpr (TIMING) {
        if (blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == 40) && lane == 0) {
            unsigned long long *r = tim + ((blockIdx.x == 0 ? 0 : 40) + wave) * 16;
            for (int i = 0; i < 6; i++) r[i] = tacc[i];
            r[15] = (unsigned long long)(st1 - st0);
        }
    }
}

template <int ABL, int TIMING>
static int xerr7_launch(dim3 grid, const uint32_t *A16T, int mpad, const uint32_t *Y16, const uint32_t *H16c, const uint32_t *W16c, double *C, int npad, size_t slab,
                        int stages, int sps, const int *scal, int n, int m, double *P, unsigned long long *tim)
{
    const int lds = 3 * XPROD_A_IMG_BYTES + 2 * 128 * XPROD_ROWB;
    if (hipFuncSetAttribute((const void *)xerr7_kernel<4, ABL, TIMING>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { printf("LDS attribute refused\n"); return 1; }
    xerr7_kernel<4, ABL, TIMING><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, H16c, W16c, C, npad, slab, 0, stages, sps, scal, scal + 2, n, m, P, tim);
    return hipGetLastError() != hipSuccess;
}

template <int SPREAD, int TIMING, int PAT>
static int xerr3_launch(dim3 grid, const uint32_t *A16T, int mpad, const uint32_t *Y16, const uint32_t *H16c, const uint32_t *W16c, double *C, int npad, size_t slab,
                        int stages, int sps, const int *scal, int n, int m, double *P, unsigned long long *tim)
{
    const int lds = 2 * XPROD16_ERR_BUF;
    if (hipFuncSetAttribute((const void *)xerr3_kernel<4, SPREAD, TIMING, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
    xerr3_kernel<4, SPREAD, TIMING, PAT><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, H16c, W16c, C, npad, slab, 0, stages, sps, scal, scal + 2, n, m, P, tim);
    return hipGetLastError() != hipSuccess;
}

// variants: 3 = all requests at the top, plain order; 4 = spread; 5 = spread + interleave pattern; 13 / 14 / 15: the same with time marks
static int xerr_launch(int v, dim3 grid, const uint32_t *A16T, int mpad, const uint32_t *Y16, const uint32_t *H16c, const uint32_t *W16c, double *C, int npad,
                       size_t slab, int stages, int sps, const int *scal, int n, int m, double *P, unsigned long long *tim)
{
    switch (v) {
    case 3: return xerr3_launch<0, 0, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 4: return xerr3_launch<1, 0, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 5: return xerr3_launch<1, 0, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 6: return xerr3_launch<0, 0, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 13: return xerr3_launch<0, 1, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 14: return xerr3_launch<1, 1, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 15: return xerr3_launch<1, 1, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 12: return xerr3_launch<0, 1, 2>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 16: return xerr3_launch<0, 1, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 7: return xerr4_launch<0, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 8: return xerr4_launch<1, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 17: return xerr4_launch<0, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 18: return xerr4_launch<1, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 9: return xerr5_launch<1, 0, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 10: return xerr5_launch<0, 0, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 19: return xerr5_launch<1, 1, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 20: return xerr6_launch<0, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 21: return xerr6_launch<1, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 30: return xerr6_launch<0, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 31: return xerr6_launch<1, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 22: return xerr7_launch<0, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 32: return xerr7_launch<0, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 41: return xerr7_launch<1, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 42: return xerr7_launch<2, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 43: return xerr7_launch<3, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 44: return xerr7_launch<4, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 46: return xerr7_launch<6, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 23: return xerr7_launch<8, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 24: return xerr7_launch<16, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 25: return xerr7_launch<24, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 26: return xerr7_launch<32, 0>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    case 36: return xerr7_launch<32, 1>(grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
    default: return 1;
    }
}
