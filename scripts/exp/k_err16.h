// k_err16.h -- err16_kernel: the fp32-operand mode's error block as a kernel of its own on the fp16 matrix cores, meant to run BESIDE the
// sweep of the speculative W half-step instead of riding in its cross product (experiment of round 5, measured and NOT taken).
// Wired into nnlm_run behind a temporary switch (errors_launch: split copies W16c / H16c from the fused path's preparation, the plain cross
// product on the main stream, this kernel on the error stream after ev_xdone, grid (npad / 64, S); not kept), bench.py at config 2, same final mse:
//     fused cross product (xprod16_err_kernel, production)            0.686 ms per step   1395 it/s
//     err16_kernel on the error stream after the cross product, S = 4 0.737               1301        (S = 2: 0.757, 8: 0.740, 16: 0.743)
// The kernel needs ~0.3 ms beside the sweep (the host waits for it: 0.36 ms from the cross product's end to the sums) where the fused
// kernel adds 0.107: next to a 272-register wavefront of the persistent sweep only ONE of its wavefronts fits on a SIMD (234 VGPRs; at
// 168 it spills 236 bytes), and one wavefront cannot overlap its own load wait, barrier, 24 MFMAs and ~770 cycles of vector work per stage.
#pragma once
#include "csrc_r5/k_xprod16.h"

// ---- the error block as a kernel of its own on the fp16 matrix cores (round 5) ---------------------------------------------------
// xprod16_err_kernel above lets the error sums of a trace iteration ride in the speculative W half-step's cross product: +0.107 ms on
// a kernel that is on the critical path.  The SCD sweep that follows that cross product is bound by dependent latency, streams nothing
// from HBM and leaves most of every CU's issue slots empty -- so the same sums can be formed BESIDE it, by a kernel that reads the
// resident fp32 A once (0.8 GB, HBM idle otherwise) and forms W H on the fp16 matrix cores exactly as the fused kernel does:
//   * a block of four wavefronts owns 64 rows i (a wavefront 16 of them: its rows of W, kq-contiguous split copy W16c, in registers)
//     and walks the columns j in stages of 64 (H16c [mpad][2][64] -> a 16 KB LDS image per stage, XOR-swizzled, two buffers);
//   * a(i, j) comes straight from global memory IN the accumulator layout of the 16 x 16 tiles (lane (l15 = column, lg): rows
//     4 lg .. 4 lg + 3 of column 16 t + l15 = one 16-byte load per tile) -- no A image in LDS, no one-hot layout products -- requested
//     TWO stages ahead (three register sets: one wavefront per SIMD has to keep 5 TB/s of loads in flight);
//   * W H: 3 v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 (hi hi, hi lo, lo hi), 24 per stage and wavefront; the two sums as in the
//     fused kernel (fp32 over a stage, fp64 across stages), one pair per block in `partial`;
//   * HAS_MISS: the missing bits of the lane's four rows come from the column-major bit matrix (one word per tile and lane).
// 256 threads, one wavefront per SIMD at < 240 VGPRs: it fits on a SIMD beside a wavefront of the persistent sweep (272 registers).
#define ERR16_THREADS 256
#define ERR16_WAVES 4
#define ERR16_ROWS 64
#define ERR16_LDS_BYTES (2 * 64 * XPROD_ROWB)
template <bool HAS_MISS>
__global__ __launch_bounds__(ERR16_THREADS, 2) void err16_kernel(const float *__restrict__ A, int lda, const uint32_t *__restrict__ miss, int words,
                                                                 const uint32_t *__restrict__ H16c, const uint32_t *__restrict__ W16c,
                                                                 const int *__restrict__ h_exp, const int *__restrict__ w_exp, int n_rows, int n_cols,
                                                                 int stage_begin, int stage_end, int stages_per_split, double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red[2][ERR16_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int i0 = blockIdx.x * ERR16_ROWS;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;
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
    const float cwh = ldexpf(1.0f, -(w_exp[0] + h_exp[0])); // (W H) = (main + cross / 2048) * cwh
    const int irow = i0 + 16 * wave + 4 * lg;              // the lane's four rows (accumulator layout: rows 4 lg + r of the wave's 16)
    const float *arow = A + irow;
    const uint32_t *mrow = HAS_MISS ? miss + (irow >> 5) : nullptr;
    const int mshift = irow & 31;
    double s2 = 0.0, skl = 0.0;

    auto issue_h = [&](int st, unsigned char *buf) { // 64 columns j of the stage, 256 bytes (64 hi | 64 lo over kq) each
        const size_t c0 = (size_t)st * 64;
#pragma unroll
        for (int t = wave; t < 16; t += ERR16_WAVES) {
            const int row = 4 * t + lg;
            const int sx = l15 ^ (row & 15);
            glds16(H16c + (c0 + row) * 64 + sx * 4, buf + t * 1024);
        }
    };
    auto load_a = [&](int st, f32x4 (&av)[4], uint32_t (&mw)[4]) { // (past the last stage: the last one again -- unconditional requests keep the counts constant)
        const int sc = st < st1 ? st : st1 - 1;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const size_t j = (size_t)sc * 64 + 16 * t + l15;
            av[t] = *(const f32x4 *)(arow + j * lda);
            if (HAS_MISS) mw[t] = mrow[j * words];
        }
    };
    constexpr int NLD = HAS_MISS ? 8 : 4; // register-destination loads per stage and lane
    f32x4 a0[4], a1[4], a2[4];
    uint32_t m0[4], m1[4], m2[4];
    if (st0 >= st1) return; // (whole block)
    issue_h(st0, smem);
    load_a(st0, a0, m0);
    load_a(st0 + 1, a1, m1);
    // stage st: its H image was requested one stage ago, its a values two stages ago; behind them in the queue: the a values of stage
    // st + 1 (NLD loads) -- waiting until only those are outstanding is waiting for everything this stage needs
    auto stage = [&](int st, f32x4 (&av)[4], uint32_t (&mw)[4], f32x4 (&avn)[4], uint32_t (&mwn)[4]) {
        unsigned char *buf = smem + ((st - st0) & 1) * (64 * XPROD_ROWB);
        wait_vmcnt(NLD);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 1 < st1) issue_h(st + 1, smem + ((st + 1 - st0) & 1) * (64 * XPROD_ROWB)); // (the other buffer: every wavefront has left stage st - 1)
        load_a(st + 2, avn, mwn);
        f32x4 em[4], ex[4];
#pragma unroll
        for (int t = 0; t < 4; t++) em[t] = f32x4{0, 0, 0, 0}, ex[t] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
            const int sh = (4 * c2 + lg), sl = 8 + 4 * c2 + lg; // logical 16-byte slots of the hi / lo halves
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const unsigned char *hrow = buf + (16 * t + l15) * XPROD_ROWB;
                const xh8 hh = *(const xh8 *)(hrow + ((sh ^ l15) * 16));
                const xh8 hl = *(const xh8 *)(hrow + ((sl ^ l15) * 16));
                em[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hh, em[t], 0, 0, 0);
                ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c2], hl, ex[t], 0, 0, 0);
                ex[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c2], hh, ex[t], 0, 0, 0);
            }
        }
        f32x4 p2 = f32x4{0, 0, 0, 0}, pk = f32x4{0, 0, 0, 0};
        const bool interior = (i0 + ERR16_ROWS <= n_rows) && (st * 64 + 64 <= n_cols);
        const float il = 1.0f / XPROD16_LO_SCALE, tiny = (float)NNLM_TINY;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const f32x4 aa = av[t];
            const f32x4 ah2 = (em[t] + ex[t] * il) * cwh;
            const f32x4 d = aa - ah2;
            f32x4 lg4;
#pragma unroll
            for (int r = 0; r < 4; r++) lg4[r] = log2_native(ah2[r] + tiny); // (log2: ln 2 goes into the coefficient)
            f32x4 t2 = d * d;
            f32x4 tk = ah2 - (aa * NNLM_LN2F + tiny * NNLM_LN2F) * lg4;
            if (HAS_MISS || !interior) {
                const uint32_t mbits = HAS_MISS ? (mw[t] >> mshift) : 0u;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool valid = (irow + r < n_rows) && (st * 64 + 16 * t + l15 < n_cols) && !((mbits >> r) & 1u);
                    if (!valid) t2[r] = 0.f, tk[r] = 0.f;
                }
            }
            p2 += t2;
            pk += tk;
        }
        s2 += (double)((p2[0] + p2[1]) + (p2[2] + p2[3]));
        skl += (double)((pk[0] + pk[1]) + (pk[2] + pk[3]));
    };
    int st = st0;
    for (; st + 2 < st1; st += 3) { // (triples: the three register sets of A keep their names, no copies)
        stage(st, a0, m0, a2, m2);
        stage(st + 1, a1, m1, a0, m0);
        stage(st + 2, a2, m2, a1, m1);
    }
    if (st < st1) stage(st, a0, m0, a2, m2);
    if (st + 1 < st1) stage(st + 1, a1, m1, a0, m0);
    wait_vmcnt(0); // (the redundant requests past the last stage)
    s2 = wave_sum(s2);
    skl = wave_sum(skl);
    if (lane == 0) red[0][wave] = s2, red[1][wave] = skl;
    __syncthreads();
    if (tid == 0) {
        double a2s = 0.0, aks = 0.0;
        for (int w = 0; w < ERR16_WAVES; w++) a2s += red[0][w], aks += red[1][w];
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = a2s;
        partial[2 * blk + 1] = aks;
    }
}

