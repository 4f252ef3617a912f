// Stand-alone timing of errors64_kernel<true> (strict mode, missing entries): the round-5 kernel (104 bytes of scratch per lane;
// default) against the product's (-DNEW: lane constants rebuilt per tile, H slice stored in front of the sums: no scratch).  Same
// inputs, same launch shape as nnlm_errors; prints the two sums (must agree to the last bit: the arithmetic did not change).
#ifdef NEW
#include "../../nnlm_amd/csrc/common.h"
#include "../../nnlm_amd/csrc/k_errors.h"
#else
#include "csrc_r5/k_errors.h"
#endif
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(double *p, size_t cnt, double scale, unsigned seed)
{
    size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= cnt) return;
    unsigned long long x = (e + 1) * 6364136223846793005ull + seed * 1442695040888963407ull;
    x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
    p[e] = scale * (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
__global__ void fill_bits(uint32_t *p, size_t cnt) // ~10 % of the bits set
{
    size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= cnt) return;
    uint32_t w = 0;
    for (int b = 0; b < 32; b++) {
        unsigned long long x = (e * 32 + b + 1) * 6364136223846793005ull + 99 * 1442695040888963407ull;
        x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
        if ((x >> 11) % 10 == 0) w |= 1u << b;
    }
    p[e] = w;
}
int main(int argc, char **argv)
{
    const int n = 20000, m = 10000, k = 50, npad = 20096, mpad = 10112, KP = 64, k4 = 52;
    double *A, *W, *H, *part, *out;
    uint32_t *miss;
    CK(hipMalloc(&A, (size_t)npad * mpad * 8)); CK(hipMalloc(&W, (size_t)KP * npad * 8)); CK(hipMalloc(&H, (size_t)KP * mpad * 8));
    CK(hipMalloc(&part, (size_t)2 * 60000 * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&miss, (size_t)mpad * (npad / 32) * 4));
    fill<<<(unsigned)(((size_t)npad * mpad + 255) / 256), 256>>>(A, (size_t)npad * mpad, 1.0, 1);
    fill_bits<<<(unsigned)(((size_t)mpad * (npad / 32) + 255) / 256), 256>>>(miss, (size_t)mpad * (npad / 32));
    CK(hipMemset(W, 0, (size_t)KP * npad * 8)); CK(hipMemset(H, 0, (size_t)KP * mpad * 8));
    fill<<<(unsigned)(((size_t)k * npad + 255) / 256), 256>>>(W, (size_t)k * npad, 0.2, 2);
    fill<<<(unsigned)(((size_t)k * mpad + 255) / 256), 256>>>(H, (size_t)k * mpad, 0.2, 3);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nit = npad / ERR_TILE, jcnt = mpad / ERR_TILE;
    int nchunks = argc > 1 ? atoi(argv[1]) : (8 * 512 + nit / 2) / nit;
    const int chunk = (jcnt + nchunks - 1) / nchunks;
    nchunks = (jcnt + chunk - 1) / chunk;
    const int lds = errors64_lds_bytes(k4);
    for (int miss_on = 0; miss_on < 2; miss_on++) {
        const size_t nb = (size_t)nit * nchunks;
        float best = 1e9f;
        for (int rep = 0; rep < 8; rep++) {
            hipEventRecord(e0);
            if (miss_on) {
                hipFuncSetAttribute((const void *)errors64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                errors64_kernel<true><<<(unsigned)nb, ERR64_THREADS, lds>>>(A, npad, miss, W, npad, H, mpad, n, m, k4, part, 0, jcnt, chunk, nit);
            } else {
                hipFuncSetAttribute((const void *)errors64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                errors64_kernel<false><<<(unsigned)nb, ERR64_THREADS, lds>>>(A, npad, nullptr, W, npad, H, mpad, n, m, k4, part, 0, jcnt, chunk, nit);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        CK(hipGetLastError());
        reduce_partials_kernel<<<1, REDUCE_THREADS>>>(part, nb, 2, out);
        double res[2];
        CK(hipMemcpy(res, out, 16, hipMemcpyDeviceToHost));
        printf("errors64_kernel<%s>: %.3f ms (%zu blocks, chunk %d)  sums %.17e %.17e\n", miss_on ? "true" : "false", best, nb, chunk, res[0], res[1]);
    }
    return 0;
}
