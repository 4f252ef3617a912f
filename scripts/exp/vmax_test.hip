#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *o, const double *in)
{
    double x = in[0], m = in[1], d, d2;
    asm("v_max_f64 %0, -%1, -%2" : "=v"(d) : "v"(x), "v"(m));
    double acc = __builtin_amdgcn_mfma_f64_4x4x4f64(0.0, 0.0, m, 0, 0, 0);
    asm("v_max_f64 %0, -%1, -%2" : "=v"(d2) : "v"(x), "v"(acc));
    o[threadIdx.x] = d; o[64 + threadIdx.x] = d2; o[128 + threadIdx.x] = x + d2;
}
int main()
{
    double *d, h[192], in[2] = {0.185, 3.4};
    hipMalloc(&d, 192 * 8 + 16); hipMemcpy(d + 192, in, 16, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, d + 192); hipMemcpy(h, d, 192 * 8, hipMemcpyDeviceToHost);
    printf("asm max(-0.185,-3.4) = %g ; after mfma: %g ; x + d = %g\n", h[0], h[64], h[128]);
    return 0;
}
