// Ablation micro-benchmark of xprod_tn_kernel (not part of the product).
#include "../../nnlm_amd/csrc/k_xprod.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int EXP> static float run(const float *A, int lda, const float *Y, int ldy, double *Cx, int ldc, int tiles, int S, int sps, int stages, int reps)
{
    const int lds = xprod_tn_lds_bytes(64);
    hipFuncSetAttribute((const void *)xprod_tn_kernel<float, 4, EXP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid(tiles, S);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    xprod_tn_kernel<float, 4, EXP><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)64 * ldc, 0, stages, sps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) xprod_tn_kernel<float, 4, EXP><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)64 * ldc, 0, stages, sps);
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
    for (int S : {3, 6}) {
        const int sps = (stages + S - 1) / S;
        printf("S=%d (%d blocks): full %.3f | no-Y %.3f | no-MFMA %.3f | no-A(cached) %.3f | no-A no-Y %.3f | loads only(no MFMA, no Y) %.3f ms\n", S, tiles * S,
               run<0>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<1>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10),
               run<2>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<4>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10),
               run<5>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10), run<3>(A, npad, Y, npad, Cx, mpad, tiles, S, sps, stages, 10));
    }
    return 0;
}
