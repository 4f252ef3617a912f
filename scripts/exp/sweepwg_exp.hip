// Standalone check + timing of sweep_scd_wg_kernel against sweep_scd_mfma_kernel (not part of the product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o sweepwg_exp sweepwg_exp.hip ; ./sweepwg_exp [ncols] [k] [max_iter]
#ifndef SWEEP_NO_TIMING
#define SWEEP_WG_TIMING 1
#endif
#ifndef FASTV
#define FASTV true
#endif
#include "csrc_r5/k_sweep_mfma.h"
#include "csrc_r5/k_sweep_wg.h"
#include "csrc_r5/k_sweep_wgf.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#ifndef TAILV
#define TAILV false
#endif
#if FASTV
#define WGK(NT_) sweep_scd_wgf_kernel<NT_, false, (TAILV && NT_ >= 2)>
#else
#define WGK(NT_) sweep_scd_wg_kernel<NT_, false>
#endif
template <int NT> static int run(int ncols, int k, int max_iter)
{
    const int KP = 16 * NT;
    const int ld = (ncols + 255) / 256 * 256;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0, 1);
    std::vector<double> G(KP * KP, 0.0), X((size_t)KP * ld, 0.0), C((size_t)KP * ld, 0.0), W((size_t)k * 500);
    for (auto &w : W) w = U(rng);
    for (int q = 0; q < k; q++) for (int r = 0; r < k; r++) { double s = 0; for (int i = 0; i < 500; i++) s += W[q * 500 + i] * W[r * 500 + i]; G[q * KP + r] = s; }
    for (int q = 0; q < k; q++) for (int c = 0; c < ncols; c++) { X[(size_t)q * ld + c] = U(rng); C[(size_t)q * ld + c] = 125 * U(rng); }
    double *dG, *dX, *dC, *dO1, *dO2, *dK; unsigned long long *dS;
    CK(hipMalloc(&dG, G.size() * 8)); CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dC, C.size() * 8)); CK(hipMalloc(&dS, 16));
    CK(hipMalloc(&dO1, X.size() * 8)); CK(hipMalloc(&dO2, X.size() * 8)); CK(hipMalloc(&dK, 16 * SWEEP_WG_CONSTS * 8));
    CK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(dS, 0, 16));
    CK(hipMemset(dO1, 0, X.size() * 8)); CK(hipMemset(dO2, 0, X.size() * 8));
    SweepArgs a{};
    a.X = dX; a.ldx = ld; a.ldo = ld; a.ocol0 = 0; a.col0 = 0; a.Graw = dG; a.KPg = KP; a.Cx = dC; a.slab_stride = (size_t)KP * ld; a.nslabs = 1; a.ldc = ld;
    a.ncols = ncols; a.k = k; a.r0 = 0.02; a.r1 = 0.01; a.r2 = 0.03; a.mask = nullptr; a.max_iter = max_iter; a.rel_tol = getenv("REL_TOL") ? atof(getenv("REL_TOL")) : 1e-9; a.op = nullptr; a.op_mode = 0; a.sweeps = dS;
#if FASTV
    if (getenv("GRAM")) { // the epilogue that leaves max|x| and Gram partial sums behind
        const int nwg = (ncols + SWEEP_WG_COLS - 1) / SWEEP_WG_COLS;
        CK(hipMalloc(&a.gram_slabs, (size_t)nwg * KP * KP * 8));
        CK(hipMalloc(&a.maxbits, 4)); CK(hipMemset(a.maxbits, 0, 4));
    }
#endif
    const int lds = FASTV ? sweep_wgf_lds_bytes(NT) : sweep_wg_lds_bytes(NT);
    CK(hipFuncSetAttribute((const void *)WGK(NT), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms1 = 0, ms2 = 0;
    for (int rep = 0; rep < 3; rep++) {
        a.Xout = dO1;
        hipEventRecord(e0);
        sweep_scd_mfma_kernel<NT, false><<<(ncols + 63) / 64, 256>>>(a);
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms1, e0, e1);
        a.Xout = dO2;
        hipEventRecord(e0);
        sweep_consts_kernel<<<1, 256>>>(dG, KP, k, a.r0, a.r1, dK, FASTV ? 1 : 0);
        WGK(NT)<<<(ncols + SWEEP_WG_COLS - 1) / SWEEP_WG_COLS, SWEEP_WG_THREADS, lds>>>(a, dK);
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms2, e0, e1);
    }
    CK(hipGetLastError());
    {
        unsigned long long *dT, T[18];
        CK(hipMalloc(&dT, 144)); CK(hipMemset(dT, 0, 144));
        SweepArgs b2 = a; b2.op = dT; b2.op_mode = 0; b2.Xout = dO2;
        WGK(NT)<<<(ncols + SWEEP_WG_COLS - 1) / SWEEP_WG_COLS, SWEEP_WG_THREADS, lds>>>(b2, dK);
        CK(hipMemcpy(T, dT, 144, hipMemcpyDeviceToHost));
        const double steps = (double)max_iter * ((k + 3) / 4);
        printf("  per step (100 MHz ticks x 24 ~ cycles): chain wave work %.1f wait %.1f | update wave work %.1f wait %.1f  [raw counter units]\n",
               T[0] / steps, T[1] / steps, T[2] / steps, T[3] / steps);
#ifdef SWEEP_WG_MARKS
        printf("  chain marks :"); for (int i = 0; i < 7; i++) printf(" %.0f", T[4 + i] / steps); printf("\n");
        printf("  update marks:"); for (int i = 0; i < 7; i++) printf(" %.0f", T[11 + i] / steps); printf("\n");
#endif
        hipFree(dT);
    }
    std::vector<double> O1(X.size()), O2(X.size()), K(16 * SWEEP_WG_CONSTS);
    CK(hipMemcpy(O1.data(), dO1, X.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(O2.data(), dO2, X.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(K.data(), dK, K.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0; int wq = -1, wc = -1;
    std::vector<double> perq(k, 0.0);
    for (int q = 0; q < k; q++) for (int c = 0; c < ncols; c++) {
        const double d = fabs(O1[(size_t)q * ld + c] - O2[(size_t)q * ld + c]);
        if (d > perq[q]) perq[q] = d;
        if (d > worst) { worst = d; wq = q; wc = c; }
    }
    printf("NT=%d ncols=%d k=%d max_iter=%d: one-wave %.3f ms, workgroup %.3f ms, max |diff| %.3e at (q=%d, col=%d)\n", NT, ncols, k, max_iter, ms1, ms2, worst, wq, wc);
    if (worst > 1e-9) { printf("  per-coordinate max diff:"); for (int q = 0; q < k; q++) printf(" %.1e", perq[q]); printf("\n"); }
    if (getenv("DUMP_CONSTS")) {
        const int nbk = (k + 3) / 4, b = nbk - 1;
        printf("  consts record %d:", b); for (int i = 0; i < 32; i++) printf(" %.4g", K[b * 32 + i]); printf("\n");
        printf("  G[4b..][4b..] diag: %.4g %.4g\n", G[(4 * b) * KP + 4 * b] + 0.02 - 0.01 + 0.01 + 1e-16, 1.0 / (G[(4 * b) * KP + 4 * b] + 0.02));
    }
    hipFree(dG); hipFree(dX); hipFree(dC); hipFree(dO1); hipFree(dO2); hipFree(dK); hipFree(dS);
    return 0;
}
int main(int argc, char **argv)
{
    const int ncols = argc > 1 ? atoi(argv[1]) : 10000, k = argc > 2 ? atoi(argv[2]) : 50, it = argc > 3 ? atoi(argv[3]) : 50;
    const int NT = (k + 15) / 16;
    switch (NT) { case 1: return run<1>(ncols, k, it); case 2: return run<2>(ncols, k, it); case 3: return run<3>(ncols, k, it); default: return run<4>(ncols, k, it); }
}
