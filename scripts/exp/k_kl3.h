// k_kl3.h -- kl_tile3_kernel: kl_tile_kernel with ONE vector pass per coordinate step (experiment of round 5).
//
// kl_tile_kernel: pass A (sums over w_q b / y) -> wave totals -> barrier -> scalar part -> pass B (y += coef_q w_q).  Pass B is two
// fused multiply-adds per chunk and column behind an LDS read of its own: LDS-latency / LDS-issue bound (~590 of ~4450 cycles per step
// for the first wavefront of a block), with the vector ALUs mostly idle.  Here the update a step owes the state is applied by the NEXT
// step's pass, chunk by chunk, right before the chunk is used:
//     step q, chunk e:   y[e] += coef_{q-1} * w_{q-1}[e];   r = 1 / y[e];   sums += w_q[e] * (b[e] * r)
// Same operations on the same numbers in the same order per element (the update still precedes the next use): bit-identical results.
// Rows: two LDS buffers; step q reads w_{q-1} from one and w_q from the other, and slot e of the first is handed to row q + 1 the
// moment chunk e has been read back (it lands a whole step before step q + 1 needs it).  With E pieces requested per row, piece e of
// row q has exactly E - 1 younger requests behind it whenever it is needed (E - 1 - e of its own row, e of row q + 1): ONE constant
// s_waitcnt vmcnt(E - 1) per chunk.  The last update of the kernel is never applied (the state is not an output).
#pragma once
#include "csrc_r5/k_kl.h"

template <int EPT4, int C, int METHOD>
__global__ __launch_bounds__(KLT_THREADS) void kl_tile3_kernel(const KlTileArgs a)
{
    constexpr int NV = (METHOD == 4) ? 1 : 2;
    constexpr int NT = KLT_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char kl_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = a.k, P4 = kl_tile_p4(a.p);
    const int rowb = P4 * 16;
    double *xs = (double *)(kl_smem + 2 * (size_t)rowb); // [C][k]
    double *sws = xs + C * k;                            // [C][k]
    float *red = (float *)(sws + C * k);                 // [2][NV * C][8]
    unsigned long long *mks = (unsigned long long *)(red + 2 * 2 * C * 8);
    const int col0 = a.colbase + blockIdx.x * C;
    const float tiny = (float)NNLM_TINY;
    const bool last = (EPT4 - 1) * NT + wave * 64 < P4;
#define KLT_HAS(e_) ((e_) + 1 < EPT4 || last)
    const int voff = lane * 16, L4 = (int)(a.lda >> 2);
    // buffer 0: slots beyond the arrays' end zeroed (never loaded); buffer 1: ALL slots zeroed -- step 0 reads it as "row -1" (times coefficient 0)
    for (int i = L4 + tid; i < P4; i += NT) *(f32x4 *)(kl_smem + (size_t)i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < P4; i += NT) *(f32x4 *)(kl_smem + (size_t)rowb + (size_t)i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)kl_smem + (unsigned)wave * 1024u;
    auto issue_piece = [&](int q, int bufsel, int e) {
        const unsigned char *src = (const unsigned char *)(a.Yf + (size_t)q * a.ldyf) + (size_t)wave * 1024 + (size_t)e * (NT * 16);
        const unsigned dst = lds0 + (unsigned)bufsel * (unsigned)rowb + (unsigned)e * (NT * 16);
        const unsigned long long sp = (unsigned long long)src;
        const unsigned long long su = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sp >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)sp);
        const unsigned du = (unsigned)__builtin_amdgcn_readfirstlane((int)dst);
        if (e + 1 < EPT4 || e * NT + wave * 64 + lane < L4)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(su), "s"(du) : "memory");
    };
    const int lc = (lane < C) ? lane : 0;
    bool live_l = col0 + lc < a.ncols;
    if (live_l && a.mask) {
        bool all = true;
        for (int w = 0; w < a.mw; w++) {
            const int bits = (k - 64 * w >= 64) ? 64 : k - 64 * w;
            const unsigned long long km = (bits >= 64) ? ~0ull : ((1ull << bits) - 1ull);
            all = all && ((a.mask[(size_t)(col0 + lc) * a.mw + w] & km) == km);
        }
        live_l = !all;
    }
    if (a.mask)
        for (int e = tid; e < C * a.mw; e += NT) mks[e] = (col0 + e / a.mw < a.ncols) ? a.mask[(size_t)col0 * a.mw + e] : ~0ull;
    for (int e = tid; e < C * k; e += NT) {
        const int c = e / k, q = e - c * k, col = col0 + c;
        xs[e] = (col < a.ncols) ? a.X[(size_t)q * a.ldx + col] : 0.0;
        sws[e] = (col < a.ncols) ? (a.sumw_cols ? a.sumw_cols[(size_t)col * a.ldsw + q] : a.sumw[q]) : 1.0;
    }
    f32x4 y[C][EPT4], b[C][EPT4];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int col = (col0 + c < a.ncols) ? col0 + c : col0;
        const f32x4 *Ac = (const f32x4 *)(a.Adata + (size_t)col * a.lda), *Yc = (const f32x4 *)(a.Yinit + (size_t)col * a.lda);
#pragma unroll
        for (int e = 0; e < EPT4; e++) {
            const int idx4 = e * NT + tid;
            const bool valid = KLT_HAS(e) && idx4 < L4 && col0 + c < a.ncols;
            b[c][e] = valid ? Ac[idx4] : f32x4{0.f, 0.f, 0.f, 0.f};
            y[c][e] = valid ? Yc[idx4] : f32x4{1.f, 1.f, 1.f, 1.f};
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
        for (int e = 0; e < EPT4; e++) {
            asm volatile("" : "+v"(b[c][e]), "+v"(y[c][e]));
            y[c][e] = y[c][e] + tiny;
        }
    __syncthreads();
    const unsigned long long cmask = (C >= 64) ? ~0ull : ((1ull << C) - 1ull);
    double S_l = 0.0;
    for (int q = 0; q < k; q++) S_l += xs[lc * k + q];
    unsigned tdone_l = 0;
    bool run_l = live_l && a.max_iter > 0 && (1.0 + a.rel_tol) > a.rel_tol, flag_l = false;
    bool any = (__ballot(run_l) & cmask) != 0ull;
    int gb = 0; // buffer of the current step's row
    float coefp[C]; // what the previous step owes the state
#pragma unroll
    for (int c = 0; c < C; c++) coefp[c] = 0.f;
    if (any) {
#pragma unroll
        for (int e = 0; e < EPT4; e++)
            if (KLT_HAS(e)) issue_piece(0, 0, e);
    }
    while (any) {
        flag_l = 0.0 > a.rel_tol;
        for (int q = 0; q < k; q++) {
            const int qn = (q + 1 < k) ? q + 1 : 0;
            const unsigned char *rown = kl_smem + (size_t)gb * rowb, *rowo = kl_smem + (size_t)(gb ^ 1) * rowb;
            bool m_l = false;
            if (a.mask) m_l = (mks[lc * a.mw + (q >> 6)] >> (q & 63)) & 1ull;
            const bool doq_l = run_l && !m_l;
            const double xq_l = xs[lc * k + q];
            const double sw_l = sws[lc * k + q];
            double den4 = 1.0, rd4 = 0.0;
            auto rd_stage = [&](int st) {
                if (METHOD != 4) return;
                if (st == 0) den4 = sw_l + a.r0 * xq_l + a.r1 * (S_l - xq_l) + a.r2;
                else if (st == 1) rd4 = __builtin_amdgcn_rcp(den4);
                else rd4 = __builtin_fma(__builtin_fma(-den4, rd4, 1.0), rd4, rd4);
                asm volatile("" : "+v"(den4), "+v"(rd4));
            };
            f32x4 acc[C][NV];
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int v = 0; v < NV; v++) acc[c][v] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto wload = [&](const unsigned char *rp, int e) -> f32x4 { return *(const f32x4 *)(rp + (size_t)(e * NT + tid) * 16); };
            auto wait_new = [&]() { // piece e of this step's row has E - 1 younger requests behind it (E = pieces this wavefront requests per row)
                if (last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EPT4 - 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EPT4 > 1 ? EPT4 - 2 : 0) : "memory");
            };
            f32x4 wn[2], wo[2]; // this step's and the previous step's row elements, one chunk ahead
            wait_new();
            wn[0] = wload(rown, 0);
            wo[0] = wload(rowo, 0);
#pragma unroll
            for (int e = 0; e < EPT4; e++) {
                if (KLT_HAS(e)) {
                    const f32x4 won = wo[e & 1], wnn = wn[e & 1];
                    // slot e of the older row's buffer has been read back: it goes to row q + 1
                    asm volatile("" : : "v"(won) : "memory");
                    issue_piece(qn, gb ^ 1, e);
                    if (e + 1 < EPT4 && KLT_HAS(e + 1)) {
                        wait_new();
                        wn[(e + 1) & 1] = wload(rown, e + 1);
                        wo[(e + 1) & 1] = wload(rowo, e + 1);
                    }
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        f32x4 &yy = y[c][e];
                        yy = __builtin_elementwise_fma(f32x4{coefp[c], coefp[c], coefp[c], coefp[c]}, won, yy); // :106, :143 (owed by step q - 1)
                        f32x4 r;
                        r[0] = __builtin_amdgcn_rcpf(__builtin_fabsf(yy[0]));
                        r[1] = __builtin_amdgcn_rcpf(__builtin_fabsf(yy[1]));
                        r[2] = __builtin_amdgcn_rcpf(__builtin_fabsf(yy[2]));
                        r[3] = __builtin_amdgcn_rcpf(__builtin_fabsf(yy[3]));
                        if constexpr (METHOD == 4) {
                            acc[c][0] = __builtin_elementwise_fma(wnn, b[c][e] * r, acc[c][0]);
                        } else {
                            const f32x4 u = wnn * r;
                            const f32x4 bu = b[c][e] * u;
                            acc[c][0] = __builtin_elementwise_fma(bu, u, acc[c][0]);
                            acc[c][1] = acc[c][1] + bu;
                        }
                    }
                }
                if (e < 3) rd_stage(e);
#pragma unroll
                for (int c = 0; c < C; c++) {
#pragma unroll
                    for (int v = 0; v < NV; v++) asm volatile("" : "+v"(acc[c][v]));
                    asm volatile("" : "+v"(y[c][e]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int st = (EPT4 < 3 ? EPT4 : 3); st < 3; st++) rd_stage(st);
            {
                float t[C][NV];
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int v = 0; v < NV; v++) t[c][v] = (acc[c][v][0] + acc[c][v][1]) + (acc[c][v][2] + acc[c][v][3]);
#define KLT3_RSTEP(CTRL, RM)                                   \
    _Pragma("unroll") for (int c = 0; c < C; c++)              \
        _Pragma("unroll") for (int v = 0; v < NV; v++) t[c][v] += kl_dpp<CTRL, RM>(0.f, t[c][v]);
                KLT3_RSTEP(0xB1, 0xF)
                KLT3_RSTEP(0x4E, 0xF)
                KLT3_RSTEP(0x141, 0xF)
                KLT3_RSTEP(0x140, 0xF)
                KLT3_RSTEP(0x142, 0xA)
                KLT3_RSTEP(0x143, 0xC)
#undef KLT3_RSTEP
                if (lane == 63) {
#pragma unroll
                    for (int c = 0; c < C; c++)
#pragma unroll
                        for (int v = 0; v < NV; v++) red[((gb * C + c) * NV + v) * 8 + wave] = t[c][v];
                }
            }
            KLT_BARRIER();
            float coef_l = 0.f;
            {
                f32x4 rrv[NV][2];
#pragma unroll
                for (int v = 0; v < NV; v++)
#pragma unroll
                    for (int u = 0; u < 2; u++) rrv[v][u] = *(const f32x4 *)(red + ((gb * C + lc) * NV + v) * 8 + 4 * u);
                double sv[NV];
#pragma unroll
                for (int v = 0; v < NV; v++)
                    sv[v] = (double)(((rrv[v][0][0] + rrv[v][0][1]) + (rrv[v][0][2] + rrv[v][0][3])) + ((rrv[v][1][0] + rrv[v][1][1]) + (rrv[v][1][2] + rrv[v][1][3])));
                if (METHOD == 4) {
                    const double tmp = sv[0] * rd4;
                    const double d = (tmp - 1) * xq_l;
                    if (doq_l) {
                        coef_l = (float)d;
                        S_l += d;
                        if (wave == 0 && lane < C) xs[lc * k + q] = xq_l * tmp;
                        flag_l = flag_l || (2 * fabs(tmp - 1) > a.rel_tol * (tmp + 1));
                    }
                } else {
                    const double aa = sv[0] + a.r0;
                    const double bb = (sv[NV - 1] - sw_l) + aa * xq_l - a.r2 - a.r1 * (S_l - xq_l);
                    const double den = aa + NNLM_TINY;
                    double rd = __builtin_amdgcn_rcp(den);
                    rd = __builtin_fma(__builtin_fma(-den, rd, 1.0), rd, rd);
                    double tmp = bb * rd;
                    if (!(tmp > 0)) tmp = 0;
                    if (doq_l && tmp != xq_l) {
                        const double d = tmp - xq_l;
                        coef_l = (float)d;
                        flag_l = flag_l || (2 * fabs(d) > a.rel_tol * (tmp + xq_l + NNLM_TINY));
                        S_l += d;
                        if (wave == 0 && lane < C) xs[lc * k + q] = tmp;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < C; c++) coefp[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, coef_l), c));
            gb ^= 1;
        }
        KLT_BARRIER();
        if (run_l) {
            tdone_l++;
            run_l = tdone_l < a.max_iter && flag_l;
        }
        any = (__ballot(run_l) & cmask) != 0ull;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = tid; e < C * k; e += NT) {
        const int c = e / k, q = e - c * k, col = col0 + c;
        if (col < a.ncols) {
            const double xv = xs[e];
            a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xv;
            if (a.op_mode == 1) ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xv;
        }
    }
    if (wave == 0) {
        const long long tot = wave_sum_ll((lane < C) ? (long long)tdone_l : 0ll);
        if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
#undef KLT_HAS
}
