// Standalone check + timing of sweep_scd_f_kernel (k_sweep_f.h, fp32 chain) next to sweep_scd_q_kernel (k_sweep_q.h, fp64 chain)
// against a CPU restatement of the SCD recurrence in fp64 (not part of the product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o sweepf_exp sweepf_exp.hip ; ./sweepf_exp [ncols] [k] [max_iter]
// env: REL_TOL, MASK=1, SLABS=n, GRAM=1, NW=4|8 (wavefronts per workgroup), WARM=1 (start from the solution of a 1 % different right-hand side,
// as consecutive iterations do), TIMING=1 (s_memtime brackets), PROBE=1 (print the lane maps of the permlane swaps).
// (The variants measured on the way -- v_mfma_f32_16x16x4_f32 update, gradient as hi + lo, error feedback of x -- are in k_sweep_f_v2.h; their logs: profiles/r06_sweepf_*.log)
#define SWEEPF_TIMING 1
#include "../../nnlm_amd/csrc/k_sweep.h"
#include "../../nnlm_amd/csrc/k_sweep_q.h"
#include "../../nnlm_amd/csrc/k_sweep_f.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe_kernel(unsigned *out)
{
    const unsigned lane = threadIdx.x;
    auto p = __builtin_amdgcn_permlane16_swap(lane, 100u + lane, false, false);
    out[lane] = p[0], out[64 + lane] = p[1];
    auto q = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
    out[128 + lane] = q[0], out[192 + lane] = q[1];
    out[256 + lane] = __float_as_uint(sf_transpose<0>((float)lane, 100.f + lane, 200.f + lane, 300.f + lane));
    out[320 + lane] = __float_as_uint(sf_transpose<1>((float)lane, 100.f + lane, 200.f + lane, 300.f + lane));
    out[384 + lane] = __float_as_uint(sf_transpose<2>((float)lane, 100.f + lane, 200.f + lane, 300.f + lane));
    out[448 + lane] = __float_as_uint(sf_transpose<3>((float)lane, 100.f + lane, 200.f + lane, 300.f + lane));
}

template <int NT, int NB> static int run(int ncols, int k, int max_iter)
{
    const int KP = 16 * NT;
    const int ld = (ncols + 255) / 256 * 256;
    const double r0 = 0.02, r1 = 0.01, r2 = 0.03;
    const double tol = getenv("REL_TOL") ? atof(getenv("REL_TOL")) : 1e-9;
    const bool masked = getenv("MASK") != nullptr;
    const int NW = getenv("NW") ? atoi(getenv("NW")) : 4;
    const bool warm = getenv("WARM") != nullptr;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0, 1);
    const int nsl = getenv("SLABS") ? atoi(getenv("SLABS")) : 1;
    std::vector<double> G(KP * KP, 0.0), X((size_t)KP * ld, 0.0), C((size_t)KP * ld, 0.0), W((size_t)k * 500);
    std::vector<unsigned long long> M(ld, 0ull);
    for (auto &w : W) w = U(rng);
    for (int q = 0; q < k; q++) for (int r = 0; r < k; r++) { double s = 0; for (int i = 0; i < 500; i++) s += W[q * 500 + i] * W[r * 500 + i]; G[q * KP + r] = s; }
    for (int q = 0; q < k; q++) for (int c = 0; c < ncols; c++) { X[(size_t)q * ld + c] = U(rng); C[(size_t)q * ld + c] = 125 * U(rng); }
    if (warm) { // X := 100 exact sweeps for this right-hand side, then the right-hand side moves by 1 %
        std::vector<double> Ge0(k * k);
        for (int q = 0; q < k; q++) for (int r = 0; r < k; r++) { double g = G[q * KP + r]; if (q == r && r0 != r1) g += r0 - r1; if (r1 != 0) g += r1; if (q == r) g += 1e-16; Ge0[q * k + r] = g; }
        for (int c = 0; c < ncols; c++) {
            std::vector<double> x(k), mu(k);
            for (int q = 0; q < k; q++) x[q] = X[(size_t)q * ld + c];
            for (int q = 0; q < k; q++) { double s = r2 - C[(size_t)q * ld + c]; for (int r = 0; r < k; r++) s += Ge0[q * k + r] * x[r]; mu[q] = s; }
            for (int t = 0; t < 100; t++)
                for (int q = 0; q < k; q++) {
                    double tmp = x[q] - mu[q] / Ge0[q * k + q]; if (tmp < 0) tmp = 0;
                    const double d = tmp - x[q];
                    if (d != 0) for (int r = 0; r < k; r++) mu[r] += d * Ge0[r * k + q];
                    x[q] = tmp;
                }
            for (int q = 0; q < k; q++) { X[(size_t)q * ld + c] = x[q]; C[(size_t)q * ld + c] *= 1.0 + 0.01 * (U(rng) - 0.5); }
        }
    }
    if (masked)
        for (int c = 0; c < ncols; c++) {
            for (int q = 0; q < k; q++) if (U(rng) < 0.15) M[c] |= 1ull << q;
            if (c % 97 == 5) M[c] = ~0ull;
        }
    double *dG, *dX, *dC, *dO1, *dO2, *dI; unsigned long long *dS, *dM;
    CK(hipMalloc(&dG, G.size() * 8)); CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dC, C.size() * 8 * nsl)); CK(hipMalloc(&dS, 16)); CK(hipMalloc(&dM, M.size() * 8));
    CK(hipMalloc(&dO1, X.size() * 8)); CK(hipMalloc(&dO2, X.size() * 8)); CK(hipMalloc(&dI, sweepq_img_doubles(NB, true) * 8));
    CK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    {
        std::vector<double> Cs(C.size());
        for (int sl = 0; sl < nsl; sl++) { // slab sl = C * w_sl with weights that sum to one (powers of two: exact)
            const double w = (nsl == 1) ? 1.0 : (sl == 0 ? 0.5 : 0.5 / (nsl - 1));
            for (size_t i = 0; i < C.size(); i++) Cs[i] = C[i] * w;
            CK(hipMemcpy(dC + (size_t)sl * C.size(), Cs.data(), C.size() * 8, hipMemcpyHostToDevice));
        }
    }
    CK(hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dO1, 0, X.size() * 8)); CK(hipMemset(dO2, 0, X.size() * 8));
    SweepArgs a{};
    a.X = dX; a.ldx = ld; a.ldo = ld; a.ocol0 = 0; a.col0 = 0; a.Graw = dG; a.KPg = KP; a.Cx = dC; a.slab_stride = (size_t)KP * ld; a.nslabs = nsl; a.ldc = ld;
    a.ncols = ncols; a.k = k; a.r0 = r0; a.r1 = r1; a.r2 = r2; a.mask = masked ? dM : nullptr; a.max_iter = max_iter; a.rel_tol = tol; a.op = nullptr; a.op_mode = 0; a.sweeps = dS;
    const int nwg = (ncols + 63) / 64, nwgf = (ncols + 16 * NW - 1) / (16 * NW);
    if (getenv("GRAM")) {
        CK(hipMalloc(&a.gram_slabs, (size_t)(nwg + 1) * KP * KP * 8));
        CK(hipMalloc(&a.maxbits, 4)); CK(hipMemset(a.maxbits, 0, 4));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float msq = 0, msf = 0;
    unsigned long long S[2] = {0, 0};
    const size_t ldsq = sweepq_lds_bytes(KP, NB, false), ldsf = sweepf_lds_bytes(KP, NB, 16 * NW);
    for (int rep = 0; rep < 3; rep++) {
        // fp64 chain (round 3-5 kernel)
        a.Xout = dO1;
        CK(hipMemset(dS, 0, 16));
        sweepq_pack_kernel<<<8, 256>>>(dG, KP, k, a.r0, a.r1, NB, dI, 0);
        hipEventRecord(e0);
        if (masked) { hipFuncSetAttribute((const void *)sweep_scd_q_kernel<NT, NB, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq);
                      sweep_scd_q_kernel<NT, NB, true, false><<<nwg, SWEEPQ_THREADS, ldsq>>>(a, dI); }
        else { hipFuncSetAttribute((const void *)sweep_scd_q_kernel<NT, NB, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq);
               sweep_scd_q_kernel<NT, NB, false, false><<<nwg, SWEEPQ_THREADS, ldsq>>>(a, dI); }
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&msq, e0, e1);
        CK(hipMemcpy(&S[0], dS, 8, hipMemcpyDeviceToHost));
        // fp32 chain
        a.Xout = dO2;
        CK(hipMemset(dS, 0, 16));
        hipEventRecord(e0);
#define LF(M_, W_) { hipFuncSetAttribute((const void *)sweep_scd_f_kernel<NT, NB, M_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf); \
                     sweep_scd_f_kernel<NT, NB, M_, W_><<<nwgf, 64 * W_, ldsf>>>(a); }
        if (NW == 8) { if (masked) LF(true, 8) else LF(false, 8) } else { if (masked) LF(true, 4) else LF(false, 4) }
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&msf, e0, e1);
        CK(hipMemcpy(&S[1], dS, 8, hipMemcpyDeviceToHost));
    }
    CK(hipGetLastError());
    if (getenv("TIMING")) { // s_memtime brackets of workgroup 0, wavefront 0: prologue / sweeps / epilogue
        long long *dT, hT[4];
        CK(hipMalloc(&dT, 32)); CK(hipMemset(dT, 0, 32));
        SweepArgs b2 = a; b2.op = dT; b2.op_mode = 98; b2.Xout = dO2;
        if (NW == 8) { hipFuncSetAttribute((const void *)sweep_scd_f_kernel<NT, NB, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf); sweep_scd_f_kernel<NT, NB, false, 8><<<nwgf, 512, ldsf>>>(b2); }
        else { hipFuncSetAttribute((const void *)sweep_scd_f_kernel<NT, NB, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf); sweep_scd_f_kernel<NT, NB, false, 4><<<nwgf, 256, ldsf>>>(b2); }
        CK(hipMemcpy(hT, dT, 32, hipMemcpyDeviceToHost));
        printf("  counter ticks: prologue %lld, %lld sweeps %lld (%.1f per sweep, %.1f per block step), epilogue %lld\n", hT[0], hT[3], hT[1], (double)hT[1] / hT[3], (double)hT[1] / hT[3] / NB, hT[2]);
    }
    std::vector<double> O1(X.size()), O2(X.size());
    CK(hipMemcpy(O1.data(), dO1, X.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(O2.data(), dO2, X.size() * 8, hipMemcpyDeviceToHost));
    // CPU restatement (reference arithmetic, src/base_algorithms.cpp:3-37) on a sample of the columns
    auto edited = [&](int c, int kc) { double g = G[c * KP + kc]; if (c == kc && r0 != r1) g += r0 - r1; if (r1 != 0) g += r1; if (c == kc) g += 1e-16; return g; };
    std::vector<double> Ge(k * k);
    for (int q = 0; q < k; q++) for (int r = 0; r < k; r++) Ge[q * k + r] = edited(q, r);
    double worst_q = 0, worst_f = 0, xm = 0, sq_q = 0, sq_f = 0, sq_x = 0; int wq = -1, wc = -1; long long sweeps_ref = 0;
    const int stride = ncols > 4000 ? 37 : 1;
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
        sweeps_ref += t;
        for (int q = 0; q < k; q++) {
            const double df = fabs(O2[(size_t)q * ld + c] - x[q]), dq = fabs(O1[(size_t)q * ld + c] - x[q]);
            if (df > worst_f) { worst_f = df; wq = q; wc = c; }
            worst_q = fmax(worst_q, dq);
            sq_q += dq * dq, sq_f += df * df, sq_x += x[q] * x[q];
            xm = fmax(xm, fabs(x[q]));
        }
    }
    printf("NT=%d NB=%d ncols=%d k=%d max_iter=%d tol=%g mask=%d NW=%d slabs=%d: fp64 chain %.4f ms, fp32 chain %.4f ms; max|x| %.3g; vs CPU fp64: fp64 chain max %.3e relF %.3e | fp32 chain max %.3e (q=%d col=%d) relF %.3e\n",
           NT, NB, ncols, k, max_iter, tol, (int)masked, NW, nsl, msq, msf, xm, worst_q, sqrt(sq_q / sq_x), worst_f, wq, wc, sqrt(sq_f / sq_x));
    if (stride == 1) printf("  sweeps: fp64 chain %llu, fp32 chain %llu, CPU %lld\n", S[0], S[1], sweeps_ref);
    if (getenv("DUMP")) {
        const int c = atoi(getenv("DUMP"));
        printf("  column %d (q: x0 fp64-chain fp32-chain):\n", c);
        for (int q = 0; q < k; q++) printf("   %2d: %.7f %.7f %.7f\n", q, X[(size_t)q * ld + c], O1[(size_t)q * ld + c], O2[(size_t)q * ld + c]);
    }
    if (a.gram_slabs) { // Gram slabs of the two forms agree to fp32 accuracy of x
        std::vector<double> s1((size_t)KP * KP);
        CK(hipMemcpy(s1.data(), a.gram_slabs, s1.size() * 8, hipMemcpyDeviceToHost));
        double ref = 0; for (int cc = 0; cc < 64 && cc < ncols; cc++) ref += O2[cc] * O2[cc];
        printf("  gram slab 0 [0][0] = %.9g, sum x0^2 over the workgroup's columns = %.9g\n", s1[0], ref);
    }
    return 0;
}
int main(int argc, char **argv)
{
    if (getenv("PROBE")) {
        unsigned *d; std::vector<unsigned> h(512);
        CK(hipMalloc(&d, 512 * 4));
        probe_kernel<<<1, 64>>>(d);
        CK(hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost));
        const char *names[4] = {"permlane16_swap(lane, 100+lane)[0]", "[1]", "permlane32_swap(lane, 100+lane)[0]", "[1]"};
        for (int v = 0; v < 4; v++) { printf("%s:", names[v]); for (int l = 0; l < 64; l++) printf(" %u", h[v * 64 + l]); printf("\n"); }
        for (int v = 0; v < 4; v++) { printf("transpose<%d>:", v); for (int l = 0; l < 64; l++) printf(" %g", __builtin_bit_cast(float, h[256 + v * 64 + l])); printf("\n"); }
    }
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
