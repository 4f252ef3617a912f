// Standalone check + timing of sweep_scd_q_kernel (k_sweep_q.h) against a CPU restatement of the SCD recurrence (not part of the product).
// SLABS=3: the cross product arrives as three split-K slabs, as in the iteration loop; GRAM=1: the epilogue leaves max|x| and Gram slabs; OP=1: fp32 operand copy
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o sweepq_exp sweepq_exp.hip ; ./sweepq_exp [ncols] [k] [max_iter]
// #define SWEEPQ_DEBUG 1  (debug dumps: DUMP=col)
#include "csrc_r5/k_sweep.h"
#include "csrc_r5/k_sweep_q.h"
#include "k_sweep_q4.h" // (experiment: four columns per wavefront, mixed launches)
#include "k_sweep_q20.h" // (experiment only, not part of the product: 16 + 4 columns per wavefront -- measured, no gain)
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT, int NB> static int run(int ncols, int k, int max_iter)
{
    const int KP = 16 * NT;
    const int ld = (ncols + 255) / 256 * 256;
    const double r0 = 0.02, r1 = 0.01, r2 = 0.03;
    const double tol = getenv("REL_TOL") ? atof(getenv("REL_TOL")) : 1e-9;
    const bool masked = getenv("MASK") != nullptr, q20 = getenv("Q20") != nullptr;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0, 1);
    const int nsl = getenv("SLABS") ? atoi(getenv("SLABS")) : 1;
    std::vector<double> G(KP * KP, 0.0), X((size_t)KP * ld, 0.0), C((size_t)KP * ld, 0.0), W((size_t)k * 500);
    std::vector<unsigned long long> M(ld, 0ull);
    for (auto &w : W) w = U(rng);
    for (int q = 0; q < k; q++) for (int r = 0; r < k; r++) { double s = 0; for (int i = 0; i < 500; i++) s += W[q * 500 + i] * W[r * 500 + i]; G[q * KP + r] = s; }
    for (int q = 0; q < k; q++) for (int c = 0; c < ncols; c++) { X[(size_t)q * ld + c] = U(rng); C[(size_t)q * ld + c] = 125 * U(rng); }
    if (masked)
        for (int c = 0; c < ncols; c++) {
            for (int q = 0; q < k; q++) if (U(rng) < 0.15) M[c] |= 1ull << q;
            if (c % 97 == 5) M[c] = ~0ull;
        }
    const bool strict = getenv("STRICT") != nullptr;
    int n16 = 0, n4 = 0; // Q4=all: every column in 16-column workgroups of four 4-column wavefronts; Q4=0: none; default: the product's split
    if (getenv("Q4") && !strcmp(getenv("Q4"), "all")) n4 = (ncols + SWEEPQ4_COLS - 1) / SWEEPQ4_COLS;
    else sweepq_split(ncols, getenv("Q4") && !strcmp(getenv("Q4"), "split") && !masked, &n16, &n4);
    double *dG, *dX, *dC, *dO1, *dO2, *dI, *dIB, *dI4; unsigned long long *dS, *dM;
    CK(hipMalloc(&dG, G.size() * 8)); CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dC, C.size() * 8 * nsl)); CK(hipMalloc(&dS, 16)); CK(hipMalloc(&dM, M.size() * 8));
    CK(hipMalloc(&dO1, X.size() * 8)); CK(hipMalloc(&dO2, X.size() * 8)); CK(hipMalloc(&dI, sweepq_img_doubles(NB, true) * 8)); CK(hipMalloc(&dI4, sweepq4_img_doubles(NB, true) * 8)); CK(hipMalloc(&dIB, sweepq20_img_doubles(NB) * 8));
    CK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    {
        std::vector<double> Cs(C.size());
        for (int sl = 0; sl < nsl; sl++) { // slab sl = C * w_sl with weights that sum to one (powers of two: exact)
            const double w = (nsl == 1) ? 1.0 : (sl == 0 ? 0.5 : 0.5 / (nsl - 1));
            for (size_t i = 0; i < C.size(); i++) Cs[i] = C[i] * w;
            CK(hipMemcpy(dC + (size_t)sl * C.size(), Cs.data(), C.size() * 8, hipMemcpyHostToDevice));
        }
    } CK(hipMemset(dS, 0, 16)); CK(hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dO1, 0, X.size() * 8)); CK(hipMemset(dO2, 0, X.size() * 8));
    SweepArgs a{};
    a.X = dX; a.ldx = ld; a.ldo = ld; a.ocol0 = 0; a.col0 = 0; a.Graw = dG; a.KPg = KP; a.Cx = dC; a.slab_stride = (size_t)KP * ld; a.nslabs = nsl; a.ldc = ld;
    a.ncols = ncols; a.k = k; a.r0 = r0; a.r1 = r1; a.r2 = r2; a.mask = masked ? dM : nullptr; a.max_iter = max_iter; a.rel_tol = tol; a.op = nullptr; a.op_mode = 0; a.sweeps = dS;
    float *dOp = nullptr;
    if (getenv("OP")) { CK(hipMalloc(&dOp, (size_t)ld * KP * 4)); a.op = dOp; a.op_mode = 2; a.op_ld = KP; a.op_f64 = 0; }
    if (getenv("GRAM")) {
        const int nwg = n16 + n4 + 1;
        CK(hipMalloc(&a.gram_slabs, (size_t)nwg * KP * KP * 8));
        CK(hipMalloc(&a.maxbits, 4)); CK(hipMemset(a.maxbits, 0, 4));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms1 = 0, ms2 = 0, msp = 0;
    const bool run_old = false;
    for (int rep = 0; rep < 3; rep++) {
        a.Xout = dO2;
        CK(hipMemset(dS, 0, 16));
        hipEventRecord(e0);
        sweepq_pack2_kernel<<<16, 256>>>(dG, KP, k, a.r0, a.r1, NB, dI, dI4, strict ? 1 : 0);
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&msp, e0, e1);
        if (q20) sweepq20_pack_kernel<<<8, 256>>>(dG, KP, k, a.r0, a.r1, NB, dIB);
                size_t lds = 0;
        if (n16) lds = sweepq_lds_bytes(KP, NB, strict);
        if (n4 && sweepq4_lds_bytes(KP, NB, strict) > lds) lds = sweepq4_lds_bytes(KP, NB, strict);
        hipEventRecord(e0);
#define LAUNCH(M_, S_) { if (n4) { hipFuncSetAttribute((const void *)sweep_scd_qmix_kernel<NT, NB, M_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                   sweep_scd_qmix_kernel<NT, NB, M_, S_><<<n16 + n4, SWEEPQ_THREADS, lds>>>(a, dI, dI4, n16); } \
                         else { hipFuncSetAttribute((const void *)sweep_scd_q_kernel<NT, NB, M_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                sweep_scd_q_kernel<NT, NB, M_, S_><<<n16, SWEEPQ_THREADS, lds>>>(a, dI); } }
        if (q20) sweep_scd_q20_kernel<NT, NB><<<(ncols + SWEEPQ20_COLS - 1) / SWEEPQ20_COLS, SWEEPQ_THREADS>>>(a, dI, dIB);
        else if (masked && strict) LAUNCH(true, true)
        else if (masked) LAUNCH(true, false)
        else if (strict) LAUNCH(false, true)
        else LAUNCH(false, false)
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms2, e0, e1);
    }
    CK(hipGetLastError());
#ifdef SWEEPQ_TRACE
    {
        std::vector<unsigned long long> tr((size_t)(n16 + n4) * 16);
        CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(sweepq_trace), tr.size() * 8));
        // per SIMD: which shapes ran there, total span
        struct Slot { int n16 = 0, n4 = 0; unsigned long long t0 = ~0ull, t1 = 0; };
        std::vector<Slot> slots(8 * 16 * 16 * 4);
        unsigned long long tmin = ~0ull;
        for (int w = 0; w < (n16 + n4) * 4; w++) tmin = std::min(tmin, tr[w * 4 + 1]);
        double dur16 = 0, dur4 = 0; int c16 = 0, c4 = 0;
        for (int w = 0; w < (n16 + n4) * 4; w++) {
            const unsigned hw = (unsigned)tr[w * 4], xcc = (unsigned)(tr[w * 4] >> 32) & 15;
            const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            Slot &sl = slots[(((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd];
            if (tr[w * 4 + 3] == 16) sl.n16++, dur16 += (double)(tr[w * 4 + 2] - tr[w * 4 + 1]), c16++; else sl.n4++, dur4 += (double)(tr[w * 4 + 2] - tr[w * 4 + 1]), c4++;
            sl.t0 = std::min(sl.t0, tr[w * 4 + 1]); sl.t1 = std::max(sl.t1, tr[w * 4 + 2]);
        }
        int hist[4][8] = {};
        for (auto &sl : slots) if (sl.n16 + sl.n4) hist[std::min(sl.n16, 3)][std::min(sl.n4, 7)]++;
        printf("  trace (100 MHz ticks): mean wave duration 16-col %.0f (%d waves), 4-col %.0f (%d waves); SIMDs by (16-col waves, 4-col waves):", c16 ? dur16 / c16 : 0, c16, c4 ? dur4 / c4 : 0, c4);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) if (hist[i][j]) printf(" (%d,%d):%d", i, j, hist[i][j]);
        printf("\n");
    }
#endif
    std::vector<double> O1(X.size()), O2(X.size());
    unsigned long long S[2];
    CK(hipMemcpy(O1.data(), dO1, X.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(O2.data(), dO2, X.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(S, dS, 16, hipMemcpyDeviceToHost));
    // CPU restatement (reference arithmetic, src/base_algorithms.cpp:3-37) on a sample of the columns
    auto edited = [&](int c, int kc) { double g = G[c * KP + kc]; if (c == kc && r0 != r1) g += r0 - r1; if (r1 != 0) g += r1; if (c == kc) g += 1e-16; return g; };
    std::vector<double> Ge(k * k);
    for (int q = 0; q < k; q++) for (int r = 0; r < k; r++) Ge[q * k + r] = edited(q, r);
    double worst = 0, worst_old = 0, xm = 0; int wq = -1, wc = -1; long long sweeps_ref = 0; int nchk = 0;
    const int stride = ncols > 4000 ? 37 : 1;
    std::vector<char> checked(ncols, 0);
    for (int c = 0; c < ncols; c += stride) {
        std::vector<double> x(k), mu(k);
        for (int q = 0; q < k; q++) x[q] = X[(size_t)q * ld + c];
        for (int q = 0; q < k; q++) { double s = r2 - C[(size_t)q * ld + c]; for (int r = 0; r < k; r++) s += Ge[q * k + r] * x[r]; mu[q] = s; }
        const unsigned long long mw = masked ? M[c] : 0ull, km = (k >= 64) ? ~0ull : ((1ull << k) - 1);
        int t = 0; double rel = 1 + tol;
        if (!(masked && (mw & km) == km))
            for (; t < max_iter && rel > tol; t++) {
                rel = 0;
                for (int q = 0; q < k; q++) {
                    if ((mw >> q) & 1) continue;
                    double tmp = x[q] - mu[q] / Ge[q * k + q]; if (tmp < 0) tmp = 0;
                    if (tmp != x[q]) { const double d = tmp - x[q]; for (int r = 0; r < k; r++) mu[r] += d * Ge[r * k + q]; } else continue;
                    const double e = 2 * fabs(x[q] - tmp) / (tmp + x[q] + 1e-16); if (e > rel) rel = e;
                    x[q] = tmp;
                }
            }
        sweeps_ref += t; nchk++; checked[c] = 1;
        for (int q = 0; q < k; q++) {
            const double d = fabs(O2[(size_t)q * ld + c] - x[q]);
            if (d > worst) { worst = d; wq = q; wc = c; }
            if (run_old) worst_old = fmax(worst_old, fabs(O1[(size_t)q * ld + c] - x[q]));
            xm = fmax(xm, fabs(x[q]));
        }
    }
    printf("NT=%d NB=%d ncols=%d k=%d max_iter=%d tol=%g mask=%d strict=%d wg64=%d wg16=%d: (unused %.4f) q %.4f ms (pack %.4f ms); max|x| %.3g; q vs CPU max |diff| %.3e at (q=%d, col=%d); (unused %.3e)\n",
           NT, NB, ncols, k, max_iter, tol, (int)masked, (int)strict, n16, n4, ms1, ms2, msp, xm, worst, wq, wc, worst_old);
    if (getenv("DUMP")) {
        const int c = atoi(getenv("DUMP"));
        std::vector<double> x(k), mu(k);
        for (int q = 0; q < k; q++) x[q] = X[(size_t)q * ld + c];
        for (int q = 0; q < k; q++) { double s = r2 - C[(size_t)q * ld + c]; for (int r = 0; r < k; r++) s += Ge[q * k + r] * x[r]; mu[q] = s; }
        for (int t = 0; t < max_iter; t++)
            for (int q = 0; q < k; q++) { double tmp = x[q] - mu[q] / Ge[q * k + q]; if (tmp < 0) tmp = 0; const double d = tmp - x[q]; for (int r = 0; r < k; r++) mu[r] += d * Ge[r * k + q]; x[q] = tmp; }
#ifdef SWEEPQ_DEBUG
        {
            double *dD; std::vector<double> D(64 * 16 + 16 * 6 * 64);
            CK(hipMalloc(&dD, D.size() * 8)); CK(hipMemset(dD, 0, D.size() * 8));
            SweepArgs b2 = a; b2.op = dD; b2.op_mode = 99; b2.Xout = dO2; b2.mask = nullptr;
            sweep_scd_q_kernel<NT, NB, false, false><<<1, SWEEPQ_THREADS, sweepq_lds_bytes(KP, NB, false)>>>(b2, dI);
            CK(hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> x0(k);
            for (int q = 0; q < k; q++) x0[q] = X[(size_t)q * ld + c];
            printf("  initial nu of column %d (q: cpu gpu):\n", c);
            for (int q = 0; q < k; q++) { double s = r2 - C[(size_t)q * ld + c]; for (int r = 0; r < k; r++) s += Ge[q * k + r] * x0[r]; printf("   %2d: %.6f %.6f\n", q, s / Ge[q * k + q], D[q * 16 + c]); }
            for (int B = 0; B < 2; B++) {
                printf("  step %d, column %d, rows 0..3: (xb m0 m3 d xnew d_pend)\n", B, c);
                for (int i = 0; i < 4; i++) { printf("   "); for (int v = 0; v < 6; v++) printf(" %.6f", D[1024 + B * 384 + v * 64 + 16 * i + c]); printf("\n"); }
            }
        }
#endif
        printf("  column %d after %d sweeps (q: cpu gpu x0):\n", c, max_iter);
        for (int q = 0; q < k; q++) printf("   %2d: %.6f %.6f  (x0 %.6f)\n", q, x[q], O2[(size_t)q * ld + c], X[(size_t)q * ld + c]);
    }
    if (stride == 1) printf("  sweeps: GPU %llu, CPU %lld %s\n", S[0], sweeps_ref, (long long)S[0] == sweeps_ref ? "(equal)" : "(DIFFERENT)");
    return 0;
}
int main(int argc, char **argv)
{
    const int ncols = argc > 1 ? atoi(argv[1]) : 10000, k = argc > 2 ? atoi(argv[2]) : 50, it = argc > 3 ? atoi(argv[3]) : 50;
    const int NB = (k + 3) / 4;
    switch (NB) {
    case 1: return run<1, 1>(ncols, k, it);
    case 3: return run<1, 3>(ncols, k, it);
    case 4: return run<1, 4>(ncols, k, it);
    case 5: return run<2, 5>(ncols, k, it);
    case 8: return run<2, 8>(ncols, k, it);
    case 12: return run<3, 12>(ncols, k, it);
    case 13: return run<4, 13>(ncols, k, it);
    case 16: return run<4, 16>(ncols, k, it);
    default: printf("k not instantiated in the harness\n"); return 1;
    }
}
