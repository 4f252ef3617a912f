// Do the dependent chains of two wavefronts on one SIMD overlap?  (not part of the product)
// chain: v_max_f64 (asm) -> s_nop 1 -> v_mfma_f64_4x4x4 (dependent) -> s_nop 5 -> v_max ...   (the sweep's chain stage)
// MODE 0: the chain alone; MODE 1: + 3 independent MFMAs per stage (the 16-column step's lazies); MODE 2: chain with VALU only (fma instead of MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(256) void chain(const double *src, double *out, int iters)
{
    const int lane = threadIdx.x & 63;
    double x = src[lane], L = src[64 + lane], m0 = src[128 + lane], c = 0, m = m0;
    double a0 = src[192 + lane], l0 = 0, l1 = 0, l2 = 0;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_max_f64 %0, -%1, -%2" : "=v"(c) : "v"(x), "v"(m));
        asm volatile("s_nop 1");
        if (MODE == 2) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(m) : "v"(L), "v"(c), "v"(m0));
        else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3" : "=v"(m) : "v"(L), "v"(c), "v"(m0));
        if (MODE == 1) {
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(l0) : "v"(a0), "v"(x));
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(l1) : "v"(a0), "v"(x));
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(l2) : "v"(a0), "v"(x));
        } else asm volatile("s_nop 5");
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = c + m + l0 + l1 + l2;
}
template <int MODE> static void run(const char *name, const double *src, double *out, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s", name);
    for (int wps = 1; wps <= 4; wps++) {
        chain<MODE><<<256 * wps, 256>>>(src, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        chain<MODE><<<256 * wps, 256>>>(src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %d waves/SIMD: %.3f ms (%.1f cycles per stage at 2.4 GHz)", wps, ms, ms * 1e-3 * 2.4e9 / iters);
    }
    printf("\n");
}
int main()
{
    double *src, *out;
    hipMalloc(&src, 4096); hipMalloc(&out, (size_t)256 * 4 * 256 * 8);
    double h[512]; for (int i = 0; i < 512; i++) h[i] = 0.001 * (i % 13);
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    run<0>("chain (v_max, MFMA)", src, out, 200000);
    run<1>("chain + 3 lazy MFMAs", src, out, 200000);
    run<2>("chain (v_max, v_fma)", src, out, 200000);
    return 0;
}
