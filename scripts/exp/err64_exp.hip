// Stand-alone timing of the strict-mode error kernels (not part of the product).  EXP bits via -DERR64_EXP=...:
//   1 no logarithm / sums, 2 no MFMA phase, 4 no A loads
#include "csrc_r5/k_errors.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(double *p, size_t cnt, double scale, unsigned seed)
{
    size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= cnt) return;
    unsigned long long x = (e + 1) * 6364136223846793005ull + seed * 1442695040888963407ull;
    x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
    p[e] = scale * (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
int main(int argc, char **argv)
{
    const int n = 20000, m = 10000, k = 50, npad = 20096, mpad = 10112, KP = 64, k4 = 52;
    double *A, *W, *H, *part, *out;
    CK(hipMalloc(&A, (size_t)npad * mpad * 8)); CK(hipMalloc(&W, (size_t)KP * npad * 8)); CK(hipMalloc(&H, (size_t)KP * mpad * 8));
    CK(hipMalloc(&part, (size_t)2 * 60000 * 8)); CK(hipMalloc(&out, 64));
    fill<<<(unsigned)(((size_t)npad * mpad + 255) / 256), 256>>>(A, (size_t)npad * mpad, 1.0, 1);
    CK(hipMemset(W, 0, (size_t)KP * npad * 8)); CK(hipMemset(H, 0, (size_t)KP * mpad * 8));
    fill<<<(unsigned)(((size_t)k * npad + 255) / 256), 256>>>(W, (size_t)k * npad, 0.2, 2);
    fill<<<(unsigned)(((size_t)k * mpad + 255) / 256), 256>>>(H, (size_t)k * mpad, 0.2, 3);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double res[2][2];
    for (int which = 0; which < 2; which++) {
        const int nit = npad / ERR_TILE, jcnt = mpad / ERR_TILE;
        int nchunks = argc > 1 ? atoi(argv[1]) : (8 * 512 + nit / 2) / nit;
        const int chunk = (jcnt + nchunks - 1) / nchunks;
        nchunks = (jcnt + chunk - 1) / chunk;
        size_t nb;
        const int lds = errors64_lds_bytes(k4);
        hipFuncSetAttribute((const void *)errors64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int occ = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)errors64_kernel<false>, ERR64_THREADS, lds);
        if (which == 1) printf("errors64_kernel: %d resident blocks per CU (LDS %d B per block)\n", occ, lds);
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            if (which == 0) {
                dim3 grid(npad / ERR_TILE, jcnt);
                nb = (size_t)grid.x * grid.y;
                errors_kernel<double><<<grid, 256>>>(A, npad, nullptr, W, npad, H, mpad, n, m, k4, part, 0);
            } else {
                nb = (size_t)nit * nchunks;
                errors64_kernel<false><<<(unsigned)nb, ERR64_THREADS, lds>>>(A, npad, nullptr, W, npad, H, mpad, n, m, k4, part, 0, jcnt, chunk, nit);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        hipEventRecord(e0);
        reduce_partials_kernel<<<1, REDUCE_THREADS>>>(part, nb, 2, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float msr; hipEventElapsedTime(&msr, e0, e1);
        CK(hipMemcpy(res[which], out, 16, hipMemcpyDeviceToHost));
        printf("%s: %.3f ms (%zu blocks, chunk %d)  reduce %.3f ms  sums %.15e %.15e\n", which ? "errors64_kernel" : "errors_kernel<double>", best, nb, chunk, msr, res[which][0], res[which][1]);
    }
    printf("rel diff %.2e %.2e\n", (res[1][0] - res[0][0]) / res[0][0], (res[1][1] - res[0][1]) / res[0][1]);
    return 0;
}
