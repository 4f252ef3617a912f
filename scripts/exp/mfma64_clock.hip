// fp64 MFMA 16x16x4 issue rate and shader clock under load, by operand data (not part of the product).
//   ./mfma64_clock          -> for data in {zeros, small ints, random}, waves/SIMD in {1, 2}: ms, cycles per MFMA by s_memtime, clock by s_memtime / s_memrealtime
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define M16(c, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
template <int MODE> // 10..15: 16x16x4 operand / accumulator patterns; 0: 16x16x4 f64 MFMA x4 accumulators; 1: fp64 FMA chains (8 independent); 2: 4x4x4 MFMA x4
__global__ __launch_bounds__(256) void burn(const double *src, double *out, int iters, unsigned long long *clk)
{
    const int lane = threadIdx.x & 63;
    double a0 = src[lane], a1 = src[64 + lane], b0 = src[128 + lane], b1 = src[192 + lane];
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double f[8];
    for (int u = 0; u < 8; u++) f[u] = src[lane + u];
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            M16(c0, a0, b0);
            M16(c1, a0, b1);
            M16(c2, a1, b0);
            M16(c3, a1, b1);
        } else if (MODE == 10) { // one accumulator, dependent
            M16(c0, a0, b0);
            M16(c0, a0, b0);
            M16(c0, a0, b0);
            M16(c0, a0, b0);
        } else if (MODE == 11) { // two accumulators
            M16(c0, a0, b0);
            M16(c1, a0, b1);
            M16(c0, a1, b0);
            M16(c1, a1, b1);
        } else if (MODE == 12) { // four accumulators, same operands
            M16(c0, a0, b0);
            M16(c1, a0, b0);
            M16(c2, a0, b0);
            M16(c3, a0, b0);
        } else if (MODE == 13) { // three accumulators, A shared (the cross-product kernel's pattern) + one more
            M16(c0, a0, b0);
            M16(c1, a0, b1);
            M16(c2, a0, a1);
            M16(c0, a1, b0);
        } else if (MODE == 14) { // four accumulators, B shared by pairs
            M16(c0, a0, b0);
            M16(c1, a1, b0);
            M16(c2, a0, b1);
            M16(c3, a1, b1);
        } else if (MODE == 2) {
            double d0 = c0[0], d1 = c1[0], d2 = c2[0], d3 = c3[0];
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d0) : "v"(a0), "v"(b0));
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d1) : "v"(a0), "v"(b1));
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d2) : "v"(a1), "v"(b0));
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d3) : "v"(a1), "v"(b1));
            c0[0] = d0; c1[0] = d1; c2[0] = d2; c3[0] = d3;
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) f[u] = __builtin_fma(f[u], a0, b0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int r = 0; r < 4; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int u = 0; u < 8; u++) s += f[u];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int MODE> static void run(const char *name, const double *src, double *out, unsigned long long *clk, int wps, int iters, int per_iter)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    burn<MODE><<<256 * wps, 256>>>(src, out, 100, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    burn<MODE><<<256 * wps, 256>>>(src, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)c[0] / (double)c[1] * 100.0;
    printf("  %-10s waves/SIMD %d: %.3f ms  counter/instr %.1f  counter rate %.0f MHz (if realtime = 100 MHz)  => wall ns per instr per SIMD %.2f\n", name, wps, ms,
           (double)c[0] / ((double)iters * per_iter), mhz, ms * 1e6 / ((double)iters * per_iter * wps));
}
int main()
{
    double *src, *out; unsigned long long *clk;
    hipMalloc(&src, 4096); hipMalloc(&out, (size_t)256 * 8 * 256 * 8); hipMalloc(&clk, 16);
    std::vector<double> h(512);
    for (int data = 0; data < 3; data++) {
        for (int i = 0; i < 512; i++) h[i] = data == 0 ? 0.0 : data == 1 ? (double)(i % 7) : (double)rand() / RAND_MAX * 1.3e-3 + 1e-7 * rand();
        hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
        printf("data %s\n", data == 0 ? "zeros" : data == 1 ? "small ints" : "random");
        for (int wps = 1; wps <= 4; wps++) {
            run<0>("mfma16x16", src, out, clk, wps, 20000, 4);
            if (data == 2) {
                run<10>("16 1acc", src, out, clk, wps, 20000, 4);
                run<11>("16 2acc", src, out, clk, wps, 20000, 4);
                run<12>("16 4acc same", src, out, clk, wps, 20000, 4);
                run<13>("16 3acc", src, out, clk, wps, 20000, 4);
                run<14>("16 4acc Bsh", src, out, clk, wps, 20000, 4);
            }
            run<2>("mfma4x4x4", src, out, clk, wps, 40000, 4);
            run<1>("fma_f64", src, out, clk, wps, 40000, 8);
        }
    }
    return 0;
}
