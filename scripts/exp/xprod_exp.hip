// Ablation micro-benchmark of xprod_tn_kernel (not part of the product).
#include "csrc_r5/k_xprod.h"
#include "csrc_r5/k_xprod16.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int EXP, int NKQ = 4, int KT = 0> static float run(const float *A, int lda, const float *Y, int ldy, double *Cx, int ldc, int tiles, int S, int sps, int stages, int reps)
{
    const int lds = xprod_tn_lds_bytes(64);
    hipFuncSetAttribute((const void *)xprod_tn_kernel<float, NKQ, KT, EXP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid(tiles, S);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    xprod_tn_kernel<float, NKQ, KT, EXP><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)64 * ldc, 0, stages, sps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) xprod_tn_kernel<float, NKQ, KT, EXP><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)64 * ldc, 0, stages, sps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
template <int EXP> static float run16(const uint32_t *A, int lda, const uint32_t *Y, int ldy, double *Cx, int ldc, int tiles, int S, int sps, int stages, int reps, const int *sc)
{
    const int lds = xprod_tn_lds_bytes(64);
    hipFuncSetAttribute((const void *)xprod16_tn_kernel<4, EXP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid(tiles, S);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    xprod16_tn_kernel<4, EXP><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)64 * ldc, 0, stages, sps, sc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) xprod16_tn_kernel<4, EXP><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)64 * ldc, 0, stages, sps, sc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main()
{
    const int npad = 20224, mpad = 10112;
    float *A, *Y; double *Cx;
    CK(hipMalloc(&A, (size_t)npad * mpad * 4)); CK(hipMalloc(&Y, (size_t)64 * npad * 4)); CK(hipMalloc(&Cx, (size_t)8 * 64 * mpad * 8));
    CK(hipMemset(A, 0x3c, (size_t)npad * mpad * 4)); CK(hipMemset(Y, 0x3c, (size_t)64 * npad * 4));
    const int stages = npad / 64, tiles = mpad / 128;
    int *sc; CK(hipMalloc(&sc, 8)); CK(hipMemset(sc, 0, 8));
    for (int S : {3, 6}) {
        const int sps = (stages + S - 1) / S;
        printf("S=%d (%d blocks): full %.3f | no-Y %.3f | no-MFMA %.3f | cached A %.3f | cached A+Y %.3f | loads only %.3f | cached, no LDS reads %.3f | MFMA only (no loads/barriers/LDS) %.3f | LDS+MFMA no loads/barriers %.3f ms\n", S, tiles * S,
               run<0>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<1>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10),
               run<2>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<4>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10),
               run<5>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<3>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10),
               run<13>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<24>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10),
               run<16>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10));
        printf("   tail NKQ=3 KT=2: full %.3f | cached A+Y %.3f | MFMA only %.3f | LDS+MFMA %.3f | cached no LDS reads %.3f | NKQ=3 KT=0 (k=48): full %.3f\n",
               run<0, 3, 2>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<5, 3, 2>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10),
               run<24, 3, 2>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<16, 3, 2>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10),
               run<13, 3, 2>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<0, 3, 0>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10));
        printf("   split-fp16 (4 tiles): full %.3f | no MFMA %.3f | cached A %.3f\n", run16<0>((const uint32_t *)A, npad, (const uint32_t *)Y, npad, Cx, mpad, tiles, S, sps, stages, 10, sc),
               run16<2>((const uint32_t *)A, npad, (const uint32_t *)Y, npad, Cx, mpad, tiles, S, sps, stages, 10, sc),
               run16<4>((const uint32_t *)A, npad, (const uint32_t *)Y, npad, Cx, mpad, tiles, S, sps, stages, 10, sc));
    }
    return 0;
}
