// Dependent-launch cost of an (almost) empty kernel as a function of its launch shape (not part of the product):
// does the ~10 us gap in front of / behind the big kernels of the iteration come with the LDS size or the grid?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int *p, int sleep) { extern __shared__ int s[]; if (sleep) for (int i = 0; i < sleep; i++) __builtin_amdgcn_s_sleep(64); if (threadIdx.x == 9999) p[0] = s[0]; }
__global__ void k_write(double *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0; }
static float run(int grid, int threads, int lds, int sleep, int reps, double *buf, size_t wr)
{
    hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int *p; hipMalloc(&p, 4);
    k_empty<<<grid, threads, lds>>>(p, sleep); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) { if (wr) k_write<<<1024, 256>>>(buf, wr); k_empty<<<grid, threads, lds>>>(p, sleep); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(p);
    return ms / reps * 1000.f;
}
int main()
{
    double *buf; hipMalloc(&buf, (size_t)32 << 20);
    printf("us per launch (back-to-back, same stream):\n");
    printf("  1 block x 64, no LDS            : %.2f\n", run(1, 64, 0, 0, 400, buf, 0));
    printf("  237 blocks x 512, no LDS        : %.2f\n", run(237, 512, 0, 0, 400, buf, 0));
    printf("  237 blocks x 512, 144 KB LDS    : %.2f\n", run(237, 512, 144 * 1024, 0, 400, buf, 0));
    printf("  209 blocks x 256, 74 KB LDS     : %.2f\n", run(209, 256, 74 * 1024, 0, 400, buf, 0));
    printf("  417 blocks x 256, 74 KB LDS     : %.2f\n", run(417, 256, 74 * 1024, 0, 400, buf, 0));
    printf("  4096 blocks x 256, no LDS       : %.2f\n", run(4096, 256, 0, 0, 400, buf, 0));
    printf("  237 x 512, 144 KB, ~20 us body  : %.2f\n", run(237, 512, 144 * 1024, 12, 200, buf, 0));
    printf("  writer (16 MB) + 237 x 512 144KB: %.2f\n", run(237, 512, 144 * 1024, 0, 200, buf, (size_t)2 << 20));
    printf("  writer (16 MB) + 1 block        : %.2f\n", run(1, 64, 0, 0, 200, buf, (size_t)2 << 20));
    return 0;
}
