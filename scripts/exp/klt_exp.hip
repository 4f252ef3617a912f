// Standalone timing + check of the KL tile kernels (k_kl.h kl_tile_kernel, scripts/exp/k_kl2.h kl_tile2_kernel) -- not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DKLT_EXP=n] -o klt_exp klt_exp.hip
//   ./klt_exp p ncols k method variant [reps] [sweeps]      variant 0: kl_tile_kernel, 2: kl_tile2_kernel
// Inputs: A = U(0,1) fp32 [ncols][lda], fixed factor Yf = U(0,1) [k][lda], X = 0.5 + U(0,1) [64][ldx]; the starting states WtH come from a
// plain device kernel.  Check: the first and last columns against an fp64 host restatement of lee_kl_update / scd_kl_update.
#include "csrc_r5/k_kl.h"
#include "k_kl2.h"
#include "k_kl3.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void wh_naive(const float *Yf, int ldyf, const double *X, int ldx, int k, int p, int ncols, float *What, size_t lda)
{
    const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (i >= p) return;
    double s = 0.0;
    for (int q = 0; q < k; q++) s += (double)Yf[(size_t)q * ldyf + i] * X[(size_t)q * ldx + c];
    What[(size_t)c * lda + i] = (float)s;
}

template <int EPT4, int C, int METHOD> static int launch0(const KlTileArgs &ta, size_t lds, int nb)
{
    CK(hipFuncSetAttribute((const void *)kl_tile_kernel<EPT4, C, METHOD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kl_tile_kernel<EPT4, C, METHOD, false><<<nb, KLT_THREADS, lds>>>(ta);
    return 0;
}
template <int EPT4, int C, int METHOD> static int launch3(const KlTileArgs &ta, size_t lds, int nb) // two 256-thread workgroups per CU, one row buffer
{
    CK(hipFuncSetAttribute((const void *)kl_tile_kernel<EPT4, C, METHOD, true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kl_tile_kernel<EPT4, C, METHOD, true, 256><<<nb, 256, lds>>>(ta);
    return 0;
}
template <int EPT4, int C, int METHOD> static int launch4(const KlTileArgs &ta, size_t lds, int nb) // 1024-thread workgroups: four wavefronts per SIMD
{
    CK(hipFuncSetAttribute((const void *)kl_tile_kernel<EPT4, C, METHOD, false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kl_tile_kernel<EPT4, C, METHOD, false, 1024><<<nb, 1024, lds>>>(ta);
    return 0;
}
template <int EPT4, int C, int METHOD> static int launch5(const KlTileArgs &ta, size_t lds, int nb) // one vector pass per step (k_kl3.h)
{
    CK(hipFuncSetAttribute((const void *)kl_tile3_kernel<EPT4, C, METHOD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kl_tile3_kernel<EPT4, C, METHOD><<<nb, KLT_THREADS, lds>>>(ta);
    return 0;
}
template <int EPT4, int HC, int METHOD> static int launch2(const KlTileArgs &ta, size_t lds, int nb)
{
    CK(hipFuncSetAttribute((const void *)kl_tile2_kernel<EPT4, HC, METHOD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kl_tile2_kernel<EPT4, HC, METHOD><<<nb, KLT_THREADS, lds>>>(ta);
    return 0;
}

int main(int argc, char **argv)
{
    const int p = argc > 1 ? atoi(argv[1]) : 20000, ncols = argc > 2 ? atoi(argv[2]) : 10000, k = argc > 3 ? atoi(argv[3]) : 50;
    const int method = argc > 4 ? atoi(argv[4]) : 4, variant = argc > 5 ? atoi(argv[5]) : 0, reps = argc > 6 ? atoi(argv[6]) : 5;
    const unsigned sweeps = argc > 7 ? atoi(argv[7]) : 1;
    const size_t lda = (size_t)(p + 255) / 256 * 256;
    const int ldx = (ncols + 127) / 128 * 128, KP = 64;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> A((size_t)ncols * lda, 0.f), Yf((size_t)k * lda, 0.f);
    std::vector<double> X((size_t)KP * ldx, 0.0), sumw(k, 0.0);
    for (int c = 0; c < ncols; c++) for (int i = 0; i < p; i++) A[(size_t)c * lda + i] = U(rng);
    for (int q = 0; q < k; q++) for (int i = 0; i < p; i++) { Yf[(size_t)q * lda + i] = U(rng); sumw[q] += Yf[(size_t)q * lda + i]; }
    for (int q = 0; q < k; q++) for (int c = 0; c < ncols; c++) X[(size_t)q * ldx + c] = 0.01 * (0.5 + U(rng));
    float *dA, *dY, *dWh; double *dX, *dXo, *dS; unsigned long long *dSw;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dY, Yf.size() * 4)); CK(hipMalloc(&dWh, A.size() * 4)); CK(hipMalloc(&dX, X.size() * 8));
    CK(hipMalloc(&dXo, X.size() * 8)); CK(hipMalloc(&dS, k * 8)); CK(hipMalloc(&dSw, 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dY, Yf.data(), Yf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dS, sumw.data(), k * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dWh, 0, A.size() * 4)); CK(hipMemset(dXo, 0, X.size() * 8)); CK(hipMemset(dSw, 0, 8));
    wh_naive<<<dim3((p + 255) / 256, ncols), 256>>>(dY, (int)lda, dX, ldx, k, p, ncols, dWh, lda);
    CK(hipDeviceSynchronize());
    KlTileArgs ta{};
    ta.Adata = dA; ta.lda = lda; ta.Yinit = dWh; ta.Yf = dY; ta.ldyf = (int)lda; ta.p = p; ta.ncols = ncols; ta.k = k; ta.X = dX; ta.Xout = dXo; ta.ldx = ldx;
    ta.colbase = 0; ta.ldo = ldx; ta.ocol0 = 0; ta.sumw = dS; ta.sumw_cols = nullptr; ta.ldsw = 0; ta.r0 = 0.0; ta.r1 = 0.0; ta.r2 = 0.0; ta.mask = nullptr; ta.mw = 1;
    ta.max_iter = sweeps; ta.rel_tol = -1.0; ta.op = nullptr; ta.op_mode = 0; ta.op_ld = 0; ta.sweeps = dSw;
    const int nt = variant == 3 ? 256 : (variant == 4 ? 1024 : KLT_THREADS);
    const int e = (kl_tile_p4(p) + nt - 1) / nt;
    const int C = variant == 4 ? 2 : (e <= 2 ? 8 : (e <= 5 ? 4 : (e <= 10 ? 2 : 1)));
    const int nb = (ncols + C - 1) / C;
    const size_t lds = variant == 2 ? kl_tile2_lds_bytes(p, k, C) : kl_tile_lds_bytes(p, k, C, 0, variant == 3 ? 1 : 2, nt / 64);
#if KLT_TIMING
    unsigned long long *dT; CK(hipMalloc(&dT, (size_t)nb * 8 * 8)); CK(hipMemset(dT, 0, (size_t)nb * 8 * 8));
    ta.tim = dT;
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, tot = 0.f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0);
        int rc = 1;
#define L0(E_, C_) if (e == E_) rc = (method == 4) ? launch0<E_, C_, 4>(ta, lds, nb) : launch0<E_, C_, 3>(ta, lds, nb);
#define L2(E_, H_) if (e == E_) rc = (method == 4) ? launch2<E_, H_, 4>(ta, lds, nb) : launch2<E_, H_, 3>(ta, lds, nb);
#define L3(E_, C_) if (e == E_) rc = (method == 4) ? launch3<E_, C_, 4>(ta, lds, nb) : launch3<E_, C_, 3>(ta, lds, nb);
        if (variant == 0) { L0(10, 2) L0(5, 4) L0(3, 4) L0(8, 2) }
        else if (variant == 3) { L3(10, 2) L3(5, 4) L3(20, 1) }
#define L5(E_, C_) if (e == E_) rc = (method == 4) ? launch5<E_, C_, 4>(ta, lds, nb) : launch5<E_, C_, 3>(ta, lds, nb);
        else if (variant == 5) { L5(10, 2) L5(5, 4) L5(3, 4) L5(8, 2) }
        else if (variant == 4) { if (e == 5) rc = launch4<5, 2, 4>(ta, lds, nb); if (e == 3) rc = launch4<3, 2, 4>(ta, lds, nb); }
        else { L2(10, 1) L2(5, 2) L2(3, 2) L2(8, 1) }
        if (rc) { printf("no instantiation for EPT4 = %d\n", e); return 1; }
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 || reps == 1) { tot += ms; if (ms < best) best = ms; }
    }
    CK(hipGetLastError());
    // check a few columns in fp64 on the host
    std::vector<double> Xo(X.size());
    CK(hipMemcpy(Xo.data(), dXo, X.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    const int cols[6] = {0, 1, 2, 3, ncols / 2 + 1, ncols - 1};
    for (int ci = 0; ci < 6; ci++) {
        const int c = cols[ci];
        if (c >= ncols) continue;
        std::vector<double> y(p, 0.0), x(k);
        for (int q = 0; q < k; q++) x[q] = X[(size_t)q * ldx + c];
        for (int i = 0; i < p; i++) { double s = 0.0; for (int q = 0; q < k; q++) s += (double)Yf[(size_t)q * lda + i] * x[q]; y[i] = (double)(float)s; }
        double S = 0.0; for (int q = 0; q < k; q++) S += x[q];
        for (unsigned t = 0; t < sweeps; t++)
            for (int q = 0; q < k; q++) {
                const float *w = &Yf[(size_t)q * lda];
                if (method == 4) {
                    double num = 0.0;
                    for (int i = 0; i < p; i++) num += w[i] * ((double)A[(size_t)c * lda + i] / (y[i] + 1e-16));
                    const double tmp = num / (sumw[q]);
                    for (int i = 0; i < p; i++) y[i] += (tmp - 1) * x[q] * w[i];
                    S += (tmp - 1) * x[q];
                    x[q] *= tmp;
                } else {
                    double aa = 0.0, bb = 0.0;
                    for (int i = 0; i < p; i++) { const double u = w[i] / (y[i] + 1e-16), bv = A[(size_t)c * lda + i]; aa += bv * u * u; bb += bv * u; }
                    bb = bb - sumw[q] + aa * x[q];
                    double tmp = bb / (aa + 1e-16);
                    if (tmp < 0) tmp = 0;
                    if (tmp != x[q]) { for (int i = 0; i < p; i++) y[i] += (tmp - x[q]) * w[i]; S += tmp - x[q]; x[q] = tmp; }
                }
            }
        double nd = 0.0, nn = 0.0;
        for (int q = 0; q < k; q++) { const double d = Xo[(size_t)q * ldx + c] - x[q]; nd += d * d; nn += x[q] * x[q]; }
        const double rel = sqrt(nd / nn);
        if (rel > worst) worst = rel;
    }
    unsigned long long sw = 0; CK(hipMemcpy(&sw, dSw, 8, hipMemcpyDeviceToHost));
    // checksum of the whole output (bit-identity between variants)
    double cs = 0.0; for (int q = 0; q < k; q++) for (int c = 0; c < ncols; c++) cs += Xo[(size_t)q * ldx + c] * (1.0 + 1e-3 * ((q * 31 + c) % 97));
#if KLT_TIMING
    {
        std::vector<unsigned long long> T((size_t)nb * 8);
        CK(hipMemcpy(T.data(), dT, T.size() * 8, hipMemcpyDeviceToHost));
        double av[8] = {0};
        for (int bq = 0; bq < nb; bq++) for (int i = 0; i < 8; i++) av[i] += (double)T[(size_t)bq * 8 + i];
        double tot8 = 0; for (int i = 0; i < 8; i++) { av[i] /= nb; tot8 += av[i]; }
        printf("   cycles per block (wavefront 0, last launch): loop/prologue %.0f | top %.0f | passA %.0f | totals %.0f | barrier %.0f | scalar %.0f | passB %.0f | epilogue %.0f | sum %.0f (%.1f per step)\n",
               av[0], av[1], av[2], av[3], av[4], av[5], av[6], av[7], tot8, tot8 / (k * sweeps));
    }
#endif
    printf("p %d ncols %d k %d method %d variant %d exp %d EPT4 %d C %d lds %zu : best %.4f ms  mean %.4f ms  rel-err(6 cols) %.2e  sweeps %llu  checksum %.15e\n", p, ncols, k,
           method, variant, (int)KLT_EXP + 100 * (int)KLT_V, e, C, lds, best, tot / (reps > 1 ? reps - 1 : 1), worst, sw, cs);
    return 0;
}
