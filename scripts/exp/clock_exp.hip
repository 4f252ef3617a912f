// Measures shader clock and fp64 FMA issue/latency on the box (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_dep(double *out, long long *cyc, int n)
{
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) { a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); }
    long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
__global__ void k_ind(double *out, long long *cyc, int n)
{
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0000001, c = 1e-9;
    long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
        a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
    }
    long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
int main()
{
    double *d; long long *c, h[2];
    hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8); hipMalloc(&c, 16);
    for (int grid : {1, 1024, 4096}) {
        for (int rep = 0; rep < 3; rep++) {
            const int n = 200000;
            k_dep<<<grid, 64>>>(d, c, n); hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            printf("grid %4d dependent  : %.2f shader-ticks/FMA, %.2f ns/FMA (wall 100MHz) -> tick rate %.0f MHz\n", grid, (double)h[0] / (4.0 * n), 10.0 * h[1] / (4.0 * n), (double)h[0] / (h[1] * 0.01));
            k_ind<<<grid, 64>>>(d, c, n); hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            printf("grid %4d independent: %.2f shader-ticks/FMA, %.2f ns/FMA -> tick rate %.0f MHz\n", grid, (double)h[0] / (8.0 * n), 10.0 * h[1] / (8.0 * n), (double)h[0] / (h[1] * 0.01));
        }
    }
    return 0;
}
