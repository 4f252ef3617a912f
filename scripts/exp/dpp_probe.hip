// which lane does v_mov_b32_dpp row_ror:n read?  (decides the quad replication of k_sweep_q20.h)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK> __global__ void k(int *o)
{
    const int lane = threadIdx.x;
    o[lane] = __builtin_amdgcn_update_dpp(-1, lane, CTRL, 0xF, BANK, false);
}
template <int CTRL, int BANK> void run(int *d, const char *name)
{
    int h[64];
    k<CTRL, BANK><<<1, 64>>>(d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%-28s:", name);
    for (int l = 0; l < 20; l++) printf(" %d", h[l]);
    printf("\n");
}
int main()
{
    int *d; hipMalloc(&d, 256);
    run<0x124, 0xF>(d, "row_ror:4 bank 0xF");
    run<0x128, 0xF>(d, "row_ror:8 bank 0xF");
    run<0x12C, 0xF>(d, "row_ror:12 bank 0xF");
    run<0x124, 0x2>(d, "row_ror:4 bank 0x2 (old -1)");
    run<0x114, 0xF>(d, "row_shr:4 bank 0xF");
    return 0;
}
