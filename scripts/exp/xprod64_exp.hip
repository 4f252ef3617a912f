// Ablation micro-benchmark of xprod_tn_kernel<double> in both half-step geometries (not part of the product).
#include "../../nnlm_amd/csrc/k_xprod.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int EXP, int NKQ = 3, int KT = 2> static float run(const double *A, int lda, const double *Y, int ldy, double *Cx, int ldc, int tiles, int S, int sps, int stages, int reps)
{
    constexpr int KP = 16 * (NKQ + (KT > 0 ? 1 : 0));
    const int lds = xprod_tn_lds_bytes(KP);
    hipFuncSetAttribute((const void *)xprod_tn_kernel<double, NKQ, KT, EXP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid(tiles, S);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    xprod_tn_kernel<double, NKQ, KT, EXP><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)KP * ldc, 0, stages, sps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) xprod_tn_kernel<double, NKQ, KT, EXP><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)KP * ldc, 0, stages, sps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main(int argc, char **argv)
{
    const int npad = 20096, mpad = 10112; // the product's paddings at config 2 (multiples of 128)
    double *A, *Y, *Cx;
    CK(hipMalloc(&A, (size_t)npad * mpad * 8)); CK(hipMalloc(&Y, (size_t)64 * npad * 8)); CK(hipMalloc(&Cx, (size_t)16 * 64 * npad * 8));
    CK(hipMemset(A, 0x3c, (size_t)npad * mpad * 8)); CK(hipMemset(Y, 0x3c, (size_t)64 * npad * 8));
    for (int geo = 0; geo < 2; geo++) {
        // geo 0: H half-step (columns j = mpad, contraction npad); geo 1: W half-step on the transposed copy
        const int cols = geo == 0 ? mpad : npad, con = geo == 0 ? npad : mpad;
        const int stages = con / 32, tiles = cols / 128;
        for (int S : {1, 2, 3, 6}) {
            const int sps = (stages + S - 1) / S;
            printf("geo %d S=%d (%d blocks x %d stages): full %.3f | no-MFMA %.3f | cached A+Y %.3f | MFMA only %.3f | LDS+MFMA %.3f | KT=0 NKQ=3 full %.3f | NKQ=4 full %.3f\n", geo, S, tiles * S, sps,
                   run<0>(A, con, Y, con, Cx, cols, tiles, S, sps, stages, 5), run<2>(A, con, Y, con, Cx, cols, tiles, S, sps, stages, 5),
                   run<5>(A, con, Y, con, Cx, cols, tiles, S, sps, stages, 5), run<24>(A, con, Y, con, Cx, cols, tiles, S, sps, stages, 5),
                   run<16>(A, con, Y, con, Cx, cols, tiles, S, sps, stages, 5), run<0, 3, 0>(A, con, Y, con, Cx, cols, tiles, S, sps, stages, 5),
                   run<0, 4, 0>(A, con, Y, con, Cx, cols, tiles, S, sps, stages, 5));
        }
    }
    return 0;
}
