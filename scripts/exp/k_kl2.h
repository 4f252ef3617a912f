// k_kl2.h -- kl_tile2_kernel: the KL solvers of the fp32-operand mode with the block's columns in TWO GROUPS that run half a
// coordinate step apart (experiment of round 5; moves to nnlm_amd/csrc/k_kl.h when it wins).
//
// kl_tile_kernel takes a coordinate step as  pass A (all columns) -> wave totals -> barrier -> scalar part -> pass B (all columns):
// between the passes every wavefront of the block executes the same dependent chain at the same time (two DPP reductions, an LDS
// round trip, ~15 dependent fp64 instructions, v_readlane) and the vector ALUs idle -- ~1000-1500 of ~4900 cycles per step.
// Here the columns of a block are two groups a, b; group b lags group a by half a step, and every dependent chain of one group is
// issued between the chunks of a vector pass of the OTHER group:
//     phase 1 of step q:   [ scalar part a(q)  ||  pass A of b(q) ]   wave totals b(q)   pass B of a(q)   barrier
//     phase 2 of step q:   [ scalar part b(q)  ||  pass A of a(q+1) ] wave totals a(q+1) pass B of b(q)   barrier
// A barrier is followed by the other group's vector pass, never by a wait.  Same arithmetic per column as kl_tile_kernel (same
// chunk order, same reduction tree, same scalar formulas): results are bit-identical to it.
// Rows of the fixed factor: two LDS buffers as before; row q is last read by pass B of b(q) at the end of phase 2, which hands
// each slot to row q + 2 the moment it has read it back (the ONEBUF scheme of kl_tile_kernel); pass A of a(q+2), half a step
// later, waits for piece e with a counted s_waitcnt.
#pragma once
#include "csrc_r5/k_kl.h"

__host__ __device__ static inline size_t kl_tile2_lds_bytes(int p, int k, int C, int mw_masked = 0)
{
    return 2 * (size_t)kl_tile_p4(p) * 16 + (size_t)2 * C * k * 8 + (size_t)2 * C * 8 * 4 + (size_t)C * mw_masked * 8;
}

template <int EPT4, int HC, int METHOD>
__global__ __launch_bounds__(KLT_THREADS) void kl_tile2_kernel(const KlTileArgs a)
{
    constexpr int C = 2 * HC, NV = (METHOD == 4) ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char kl_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = a.k, P4 = kl_tile_p4(a.p);
    const int rowb = P4 * 16;
    double *xs = (double *)(kl_smem + 2 * (size_t)rowb); // [C][k]
    double *sws = xs + C * k;                            // [C][k]
    float *red = (float *)(sws + C * k);                 // [2 groups][HC * NV][8]
    unsigned long long *mks = (unsigned long long *)(red + 2 * HC * NV * 8); // [C][mw]
    const int col0 = a.colbase + blockIdx.x * C;
    const float tiny = (float)NNLM_TINY;
    const bool last = (EPT4 - 1) * KLT_THREADS + wave * 64 < P4;
#define KLT_HAS(e_) ((e_) + 1 < EPT4 || last)
    const int voff = lane * 16, L4 = (int)(a.lda >> 2);
    for (int i = L4 + tid; i < P4; i += KLT_THREADS) {
        *(f32x4 *)(kl_smem + (size_t)i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        *(f32x4 *)(kl_smem + (size_t)rowb + (size_t)i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)kl_smem + (unsigned)wave * 1024u;
    auto issue_piece = [&](int q, int bufsel, int e) {
        const unsigned char *src = (const unsigned char *)(a.Yf + (size_t)q * a.ldyf) + (size_t)wave * 1024 + (size_t)e * (KLT_THREADS * 16);
        const unsigned dst = lds0 + (unsigned)bufsel * (unsigned)rowb + (unsigned)e * (KLT_THREADS * 16);
        const unsigned long long sp = (unsigned long long)src;
        const unsigned long long su = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sp >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)sp);
        const unsigned du = (unsigned)__builtin_amdgcn_readfirstlane((int)dst);
        if (e + 1 < EPT4 || e * KLT_THREADS + wave * 64 + lane < L4)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(su), "s"(du) : "memory");
    };
    auto issue = [&](int q, int bufsel) {
#pragma unroll
        for (int e = 0; e < EPT4; e++)
            if (KLT_HAS(e)) issue_piece(q, bufsel, e);
    };

    // lane l < HC does the scalar part of column g * HC + l when group g's turn comes (every wavefront redundantly, as before)
    const int lh = (lane < HC) ? lane : 0;
    if (a.mask)
        for (int e = tid; e < C * a.mw; e += KLT_THREADS) mks[e] = (col0 + e / a.mw < a.ncols) ? a.mask[(size_t)col0 * a.mw + e] : ~0ull;
    for (int e = tid; e < C * k; e += KLT_THREADS) {
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
            const int idx4 = e * KLT_THREADS + tid;
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

    // per-group lane state (lane l < HC: column g * HC + l)
    double S_l[2];
    unsigned tdone_l[2] = {0u, 0u};
    bool run_l[2], flag_l[2] = {false, false};
    double xq_l[2] = {0.0, 0.0}, rd4_l[2] = {0.0, 0.0}, sw_l[2] = {0.0, 0.0};
    bool doq_l[2] = {false, false};
    float coef_l[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int cl = g * HC + lh;
        bool live = col0 + cl < a.ncols;
        if (live && a.mask) {
            bool all = true;
            for (int w = 0; w < a.mw; w++) {
                const int bits = (k - 64 * w >= 64) ? 64 : k - 64 * w;
                const unsigned long long km = (bits >= 64) ? ~0ull : ((1ull << bits) - 1ull);
                all = all && ((a.mask[(size_t)(col0 + cl) * a.mw + w] & km) == km);
            }
            live = !all;
        }
        double s = 0.0;
        for (int q = 0; q < k; q++) s += xs[cl * k + q];
        S_l[g] = s;
        run_l[g] = live && a.max_iter > 0 && (1.0 + a.rel_tol) > a.rel_tol;
    }
    const unsigned long long hmask = (1ull << HC) - 1ull;
    bool any = ((__ballot(run_l[0]) | __ballot(run_l[1])) & hmask) != 0ull;

    auto wload = [&](const unsigned char *rowp, int e) -> f32x4 { return *(const f32x4 *)(rowp + (size_t)(e * KLT_THREADS + tid) * 16); };

    // ---- the scalar part of group G's step q, cut into stages that are issued between the chunks of the other group's pass A ----
    // A value that comes from LDS is first USED one stage (= one chunk of vector work) after its read was issued; what a stage computes
    // is pinned where it is written (an opaque use), values still in flight are not (a pin on them would be a wait).
    // pre-part (does not depend on the step's sums): coordinate, row sum, mask bit -> Lee's reciprocal denominator
    auto pre_load = [&](auto Gc, int q) {
        constexpr int G = decltype(Gc)::value;
        const int cl = G * HC + lh;
        bool m = false;
        if (a.mask) m = (mks[cl * a.mw + (q >> 6)] >> (q & 63)) & 1ull;
        doq_l[G] = run_l[G] && !m;
        xq_l[G] = xs[cl * k + q];
        sw_l[G] = sws[cl * k + q];
    };
    double den_[2] = {1.0, 1.0};
    auto pre_den = [&](auto Gc) {
        constexpr int G = decltype(Gc)::value;
        if (METHOD == 4) {
            den_[G] = sw_l[G] + a.r0 * xq_l[G] + a.r1 * (S_l[G] - xq_l[G]) + a.r2; // :142
            double &d0 = den_[G];
            asm volatile("" : "+v"(d0));
        }
    };
    auto pre_rcp = [&](auto Gc, int part) {
        constexpr int G = decltype(Gc)::value;
        if (METHOD == 4) {
            if (part == 0) rd4_l[G] = __builtin_amdgcn_rcp(den_[G]);
            else rd4_l[G] = __builtin_fma(__builtin_fma(-den_[G], rd4_l[G], 1.0), rd4_l[G], rd4_l[G]);
            double &d0 = rd4_l[G];
            asm volatile("" : "+v"(d0));
        }
    };
    auto pre = [&](auto Gc, int q) { // (all at once: head of a sweep)
        pre_load(Gc, q);
        pre_den(Gc);
        pre_rcp(Gc, 0);
        pre_rcp(Gc, 1);
    };
    f32x4 rr_[2][NV]; // the eight wave totals of the lane's column, as read from LDS (stage 0)
    double sv_[NV];
    constexpr int NST = 7;
    auto stage = [&](auto Gc, auto Sc, int q) {
        constexpr int G = decltype(Gc)::value, ST = decltype(Sc)::value;
        const int cl = G * HC + lh;
        if constexpr (ST == 0) {
#pragma unroll
            for (int v = 0; v < NV; v++) {
                const f32x4 *rp = (const f32x4 *)(red + ((G * HC + lh) * NV + v) * 8);
                rr_[0][v] = rp[0];
                rr_[1][v] = rp[1];
            }
        } else if constexpr (ST == 1) {
#pragma unroll
            for (int v = 0; v < NV; v++) {
                sv_[v] = (double)(((rr_[0][v][0] + rr_[0][v][1]) + (rr_[0][v][2] + rr_[0][v][3])) + ((rr_[1][v][0] + rr_[1][v][1]) + (rr_[1][v][2] + rr_[1][v][3])));
                double &s0 = sv_[v];
                asm volatile("" : "+v"(s0));
            }
        } else if constexpr (ST == 2) {
            coef_l[G] = 0.f;
            if (METHOD == 4) {
                const double tmp = sv_[0] * rd4_l[G];
                const double d = (tmp - 1) * xq_l[G]; // :143
                if (doq_l[G]) {
                    coef_l[G] = (float)d;
                    S_l[G] += d;                                                               // :144
                    if (wave == 0 && lane < HC) xs[cl * k + q] = xq_l[G] * tmp;                // :145
                    flag_l[G] = flag_l[G] || (2 * fabs(tmp - 1) > a.rel_tol * (tmp + 1));      // :146-147 without the division
                }
            } else {
                const double aa = sv_[0] + a.r0;                                                        // :98,100
                const double bb = (sv_[NV - 1] - sw_l[G]) + aa * xq_l[G] - a.r2 - a.r1 * (S_l[G] - xq_l[G]); // :99,101
                const double den = aa + NNLM_TINY;
                double rd = __builtin_amdgcn_rcp(den);
                rd = __builtin_fma(__builtin_fma(-den, rd, 1.0), rd, rd);
                double tmp = bb * rd; // :102
                if (!(tmp > 0)) tmp = 0;
                if (doq_l[G] && tmp != xq_l[G]) {
                    const double d = tmp - xq_l[G];
                    coef_l[G] = (float)d;
                    flag_l[G] = flag_l[G] || (2 * fabs(d) > a.rel_tol * (tmp + xq_l[G] + NNLM_TINY)); // :107-108
                    S_l[G] += d;
                    if (wave == 0 && lane < HC) xs[cl * k + q] = tmp;
                }
            }
            float &cf = coef_l[G];
            double &d0 = S_l[G];
            asm volatile("" : "+v"(cf), "+v"(d0));
        } else if constexpr (ST == 3) {
            // the NEXT step's pre-part of this group (its coordinate is read now: wavefront 0 rewrites it a whole step from here)
            pre_load(Gc, (q + 1 < k) ? q + 1 : 0);
        } else if constexpr (ST == 4) {
            pre_den(Gc);
        } else if constexpr (ST == 5) {
            pre_rcp(Gc, 0);
        } else if constexpr (ST == 6) {
            pre_rcp(Gc, 1);
        }
    };

    // ---- pass A of group G on the row at rowp, the stages of the other group's scalar part between its chunks; ends with the wave
    // totals in red.  WAIT: pieces of this row may still be in flight (counted waits).  OG_ACTIVE: the other group has a step to finish.
    auto passA = [&](auto Gc, const unsigned char *rowp, bool wait_pieces, bool og_active, int og_q) {
        constexpr int G = decltype(Gc)::value, OG = 1 - G;
        f32x4 acc[HC][NV];
#pragma unroll
        for (int c = 0; c < HC; c++)
#pragma unroll
            for (int v = 0; v < NV; v++) acc[c][v] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (og_active) stage(std::integral_constant<int, OG>{}, std::integral_constant<int, 0>{}, og_q);
        f32x4 wq[3];
        // piece e of the row has landed once at most (pieces requested - 1 - e) younger requests are outstanding; the wavefront requested
        // EPT4 pieces (last) or EPT4 - 1
        auto wait_piece = [&](auto ec) {
            constexpr int e = decltype(ec)::value;
            if (last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EPT4 - 1 - e) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EPT4 - 2 - e > 0 ? EPT4 - 2 - e : 0) : "memory");
        };
        // (the row element of chunk e + 2 is requested while chunk e is processed: a one-column chunk is eight vector instructions, less
        //  than an LDS round trip)
        if (wait_pieces) wait_piece(std::integral_constant<int, 0>{});
        wq[0] = wload(rowp, 0);
        if (EPT4 > 1 && KLT_HAS(1)) {
            if (wait_pieces) wait_piece(std::integral_constant<int, (EPT4 > 1 ? 1 : 0)>{});
            wq[1] = wload(rowp, 1);
        }
        klq_for<0, EPT4>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            if (KLT_HAS(e)) {
                if (e + 2 < EPT4 && KLT_HAS(e + 2)) {
                    if (wait_pieces) wait_piece(std::integral_constant<int, (e + 2 < EPT4 ? e + 2 : 0)>{});
                    wq[(e + 2) % 3] = wload(rowp, e + 2);
                }
                const f32x4 w = wq[e % 3];
#pragma unroll
                for (int c = 0; c < HC; c++) {
                    const f32x4 &yy = y[G * HC + c][e], &bq = b[G * HC + c][e];
                    f32x4 r;
                    r[0] = __builtin_amdgcn_rcpf(__builtin_fabsf(yy[0]));
                    r[1] = __builtin_amdgcn_rcpf(__builtin_fabsf(yy[1]));
                    r[2] = __builtin_amdgcn_rcpf(__builtin_fabsf(yy[2]));
                    r[3] = __builtin_amdgcn_rcpf(__builtin_fabsf(yy[3]));
                    if constexpr (METHOD == 4) {
                        acc[c][0] = __builtin_elementwise_fma(w, bq * r, acc[c][0]);
                    } else {
                        const f32x4 u = w * r;
                        const f32x4 bu = bq * u;
                        acc[c][0] = __builtin_elementwise_fma(bu, u, acc[c][0]);
                        acc[c][1] = acc[c][1] + bu;
                    }
                }
            }
            // (opaque use: without it the optimizer sinks every chunk's arithmetic into the last basic block of the pass)
#pragma unroll
            for (int c = 0; c < HC; c++)
#pragma unroll
                for (int v = 0; v < NV; v++) asm volatile("" : "+v"(acc[c][v]));
            __builtin_amdgcn_sched_barrier(0);
            // one stage of the other group's scalar chain per chunk (stage s + 1 after chunk s: an LDS round trip / a few dependent
            // fp64 instructions are covered by a chunk of both wavefronts' vector work)
            if constexpr (e + 1 < NST) {
                if (og_active) stage(std::integral_constant<int, OG>{}, std::integral_constant<int, e + 1>{}, og_q);
                        __builtin_amdgcn_sched_barrier(0);
            }
        });
        if (og_active) // (short rows: the stages that found no chunk to ride on)
            klq_for<(EPT4 + 1 < NST ? EPT4 + 1 : NST), NST>([&](auto sc) { stage(std::integral_constant<int, OG>{}, sc, og_q); });
        // wave totals of group G (the chains of its HC columns step by step side by side)
        float t[HC][NV];
#pragma unroll
        for (int c = 0; c < HC; c++)
#pragma unroll
            for (int v = 0; v < NV; v++) t[c][v] = (acc[c][v][0] + acc[c][v][1]) + (acc[c][v][2] + acc[c][v][3]);
#define KLT2_STEP(CTRL, RM)                                       \
    _Pragma("unroll") for (int c = 0; c < HC; c++)                \
        _Pragma("unroll") for (int v = 0; v < NV; v++) t[c][v] += kl_dpp<CTRL, RM>(0.f, t[c][v]);
        KLT2_STEP(0xB1, 0xF)
        KLT2_STEP(0x4E, 0xF)
        KLT2_STEP(0x141, 0xF)
        KLT2_STEP(0x140, 0xF)
        KLT2_STEP(0x142, 0xA)
        KLT2_STEP(0x143, 0xC)
#undef KLT2_STEP
        if (lane == 63) {
#pragma unroll
            for (int c = 0; c < HC; c++)
#pragma unroll
                for (int v = 0; v < NV; v++) red[((G * HC + c) * NV + v) * 8 + wave] = t[c][v];
        }
    };
    // ---- pass B of group G: y += coef * w on the row at rowp; HANDOVER: slot e of that buffer goes to piece e of row qnext as soon as
    // it has been read back
    auto passB = [&](auto Gc, const unsigned char *rowp, bool handover, int qnext, int bufh) {
        constexpr int G = decltype(Gc)::value;
        float coef[HC];
#pragma unroll
        for (int c = 0; c < HC; c++) coef[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, coef_l[G]), c));
        constexpr int PD = (EPT4 < KLT_PBD) ? EPT4 : KLT_PBD;
        f32x4 wb[PD + 1];
#pragma unroll
        for (int e = 0; e < PD; e++)
            if (KLT_HAS(e)) wb[e] = wload(rowp, e);
#pragma unroll
        for (int e = 0; e < EPT4; e++) {
            if (KLT_HAS(e)) {
                if (e + PD < EPT4 && KLT_HAS(e + PD)) wb[(e + PD) % (PD + 1)] = wload(rowp, e + PD);
                const f32x4 w = wb[e % (PD + 1)];
                if (handover) {
                    asm volatile("" : : "v"(w) : "memory");
                    issue_piece(qnext, bufh, e);
                }
#pragma unroll
                for (int c = 0; c < HC; c++) {
                    f32x4 &yy = y[G * HC + c][e];
                    yy = __builtin_elementwise_fma(f32x4{coef[c], coef[c], coef[c], coef[c]}, w, yy);
                    asm volatile("" : "+v"(yy)); // (keeps the chunk's FMAs between its row request and the next one)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using G0 = std::integral_constant<int, 0>;
    using G1 = std::integral_constant<int, 1>;

    int gb = 0; // buffer of the row of the step in progress (rows alternate, continuing over sweeps)
    if (any) {
        issue(0, 0);
        if (k > 1) issue(1, 1);
        else issue(0, 1); // (rank 1: row 0 again for a sweep that may follow)
    }
    bool first = true;
    while (any) {
        flag_l[0] = flag_l[1] = 0.0 > a.rel_tol; // rel_err starts each sweep at 0 (src/base_algorithms.cpp:93,137)
        // head of the sweep: group a takes the first half of step 0 alone
        pre(G0{}, 0);
        pre(G1{}, 0);
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (rows 0 and 1; later sweeps: row 0 was waited for like any other)
        else klt_wait_vm(0);
        first = false;
        passA(G0{}, kl_smem + (size_t)gb * rowb, false, false, 0);
        KLT_BARRIER();
        for (int q = 0; q < k; q++) {
            const unsigned char *rowq = kl_smem + (size_t)gb * rowb, *rown = kl_smem + (size_t)(gb ^ 1) * rowb;
            const int qn1 = (q + 1 < k) ? q + 1 : 0;
            int qn2 = q + 2;
            if (qn2 >= k) qn2 -= k;
            if (qn2 >= k) qn2 = 0; // (k = 1)
            // phase 1: scalar part of a(q) inside pass A of b(q); pass B of a(q)
            passA(G1{}, rowq, false, true, q);
            passB(G0{}, rowq, false, 0, 0);
            KLT_BARRIER();
            // phase 2: scalar part of b(q) inside pass A of a(q + 1) (the last step of a sweep: alone); pass B of b(q) hands the row's
            // buffer over to row q + 2
            if (q + 1 < k) passA(G0{}, rown, true, true, q);
            else klq_for<0, NST>([&](auto sc) { stage(G1{}, sc, q); });
            passB(G1{}, rowq, true, qn2, gb);
            if (q + 1 < k) KLT_BARRIER();
            gb ^= 1;
            (void)qn1;
        }
        KLT_BARRIER(); // xs[] written during this sweep is read by everyone in the next
#pragma unroll
        for (int g = 0; g < 2; g++)
            if (run_l[g]) {
                tdone_l[g]++;
                run_l[g] = tdone_l[g] < a.max_iter && flag_l[g];
            }
        any = ((__ballot(run_l[0]) | __ballot(run_l[1])) & hmask) != 0ull;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = tid; e < C * k; e += KLT_THREADS) {
        const int c = e / k, q = e - c * k, col = col0 + c;
        if (col < a.ncols) {
            const double xv = xs[e];
            a.Xout[(size_t)q * a.ldo + (col - a.ocol0)] = xv;
            if (a.op_mode == 1) ((float *)a.op)[(size_t)q * a.op_ld + col] = (float)xv;
        }
    }
    if (wave == 0) {
        const long long tot = wave_sum_ll((lane < HC) ? (long long)tdone_l[0] + (long long)tdone_l[1] : 0ll);
        if (lane == 0 && tot) atomicAdd(a.sweeps, (unsigned long long)tot);
    }
#undef KLT_HAS
}
