// Ablation micro-benchmark of xprod_tn_kernel<double> in both half-step geometries (not part of the product).
#include "csrc_r5/k_xprod.h"
#include "k_xprod64_rs.h"
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
template <int NKQ = 3, int KT = 2> static float run_rs(const double *A, int lda, const double *Y, int ldy, double *Cx, int ldc, int tiles, int S, int sps, int stages, int reps)
{
    constexpr int KP = 16 * (NKQ + (KT > 0 ? 1 : 0));
    const int lds = xprod_tn_lds_bytes(KP);
    hipFuncSetAttribute((const void *)xprod64_rs_kernel<NKQ, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid(tiles, S);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    xprod64_rs_kernel<NKQ, KT><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)KP * ldc, 0, stages, sps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) xprod64_rs_kernel<NKQ, KT><<<grid, XPROD_THREADS, lds>>>(A, lda, Y, ldy, Cx, ldc, (size_t)KP * ldc, 0, stages, sps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
__global__ void fill_rand(double *X, size_t n, unsigned seed)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    X[i] = (double)(x >> 8) * (1.0 / 16777216.0) + (double)(x & 255) * 1e-9;
}
int main(int argc, char **argv)
{
    const int npad = 20096, mpad = 10112; // the product's paddings at config 2 (multiples of 128)
    double *A, *Y, *Cx;
    CK(hipMalloc(&A, (size_t)npad * mpad * 8)); CK(hipMalloc(&Y, (size_t)64 * npad * 8)); CK(hipMalloc(&Cx, (size_t)16 * 64 * npad * 8));
    CK(hipMemset(A, 0x3c, (size_t)npad * mpad * 8)); CK(hipMemset(Y, 0x3c, (size_t)64 * npad * 8));
    if (argc > 1) { // the product's forms against the one where every wavefront issues in front of its MFMAs; outputs compared bit for bit
        double *Cx2; CK(hipMalloc(&Cx2, (size_t)16 * 64 * npad * 8));
        fill_rand<<<(unsigned)(((size_t)npad * mpad + 255) / 256), 256>>>(A, (size_t)npad * mpad, 1u);
        fill_rand<<<(unsigned)(((size_t)64 * npad + 255) / 256), 256>>>(Y, (size_t)64 * npad, 2u);
        CK(hipDeviceSynchronize());
        for (int geo = 0; geo < 2; geo++) {
            const int cols = geo == 0 ? mpad : npad, con = geo == 0 ? npad : mpad;
            const int stages = con / 32, tiles = cols / 128, S = 3, sps = (stages + S - 1) / S;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(Cx, 0, (size_t)16 * 64 * npad * 8)); CK(hipMemset(Cx2, 0, (size_t)16 * 64 * npad * 8));
                const float t32 = run<32>(A, con, Y, con, Cx, cols, tiles, S, sps, stages, 20), t0 = run<0>(A, con, Y, con, Cx2, cols, tiles, S, sps, stages, 20);
                std::vector<double> c0((size_t)S * 64 * cols), c1(c0.size());
                CK(hipMemcpy(c0.data(), Cx, c0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), Cx2, c1.size() * 8, hipMemcpyDeviceToHost));
                size_t bad0 = 0; for (size_t i = 0; i < c0.size(); i++) bad0 += c0[i] != c1[i];
                CK(hipMemset(Cx2, 0, (size_t)16 * 64 * npad * 8));
                const float trs = run_rs(A, con, Y, con, Cx2, cols, tiles, S, sps, stages, 20);
                CK(hipMemcpy(c1.data(), Cx2, c1.size() * 8, hipMemcpyDeviceToHost));
                size_t bad1 = 0; for (size_t i = 0; i < c0.size(); i++) bad1 += c0[i] != c1[i];
                printf("geo %d S=3: every wavefront issues in front of its MFMAs %.4f | wavefronts 4..7 behind %.4f (%zu differ) | roles %.4f (%zu differ; c[5] = %.17g)\n", geo, t32, t0, bad0, trs, bad1, c1[5]);
            }
        }
        return 0;
    }
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
