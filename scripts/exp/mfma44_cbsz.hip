// which A lane does v_mfma_f64_4x4x4_4b_f64 read under cbsz / abid?  A[l] = l + 1, B one-hot at lb: D[lane] - 1 = source lane of A
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CBSZ, int ABID> __global__ void k(double *out, int lb)
{
    const int lane = threadIdx.x;
    double a = lane + 1, b = (lane == lb) ? 1.0 : 0.0, c = 0.0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, CBSZ, ABID, 0);
    out[lane] = c;
}
template <int CBSZ, int ABID> void run(double *d)
{
    double h[64];
    for (int lb : {0, 5, 22, 63}) {
        k<CBSZ, ABID><<<1, 64>>>(d, lb);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("cbsz %d abid %d, B one-hot at lane %2d (k=%d blk=%d j=%d): ", CBSZ, ABID, lb, lb >> 4, (lb >> 2) & 3, lb & 3);
        for (int l = 0; l < 64; l++) if (h[l] != 0.0) printf(" D[%d]<-A[%d]", l, (int)h[l] - 1);
        printf("\n");
    }
}
int main()
{
    double *d; hipMalloc(&d, 512);
    run<0, 0>(d); run<2, 0>(d); run<2, 1>(d); run<2, 3>(d); run<1, 0>(d); run<1, 1>(d);
    return 0;
}
