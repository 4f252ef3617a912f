// What bounds colsolve_f32_kernel's coordinate step at 5-7 wavefronts per SIMD?  The step as the compiler emits it --
//   v_min_f32 t, x, nu ; s_nop 0 ; v_readlane_b32 s, t, q ; s_nop 1 ; v_fma_f32 nu, -s, g, nu ; v_writelane_b32 xd, s, q
// -- against (1) the same without the wait-state fillers (WRONG results on hardware that needs them: timing only), (2) two independent
// chains interleaved so that every wait state is filled by the other chain's instruction, (3) the chain alone (v_min + v_fma with a VGPR
// broadcast stand-in: no lane instructions).  One launch of 256-thread blocks, WPS wavefronts per SIMD resident, 2500 x 16 steps each.
// hipcc --offload-arch=gfx950 -O3 -o lane_exp lane_exp.hip ; ./lane_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define STEP(X, NU, XD, S, G, Q, NOP0, NOP1) \
    "v_min_f32 %[t" #X "], %[x" #X "], %[nu" #X "]\n\t" NOP0 "v_readlane_b32 %[s" #X "], %[t" #X "], " #Q "\n\t" NOP1 \
    "v_fma_f32 %[nu" #X "], -%[s" #X "], %[g" #X "], %[nu" #X "]\n\tv_writelane_b32 %[xd" #X "], %[s" #X "], " #Q "\n\t"
template <int S> __device__ __forceinline__ void row_step(float x, float &xd, float (&nu)[4], const float (&g)[4], unsigned long long msk)
{
    const float e = __builtin_fminf(x, nu[S & 3]);
    asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(xd) : "v"(e), "s"(msk));
    const float eb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, e), 0x150 + (S >> 2), 0xF, 0xF, false));
#pragma unroll
    for (int rr = 0; rr < 4; rr++) nu[rr] = __builtin_fmaf(eb, g[rr], nu[rr]);
}
template <int S = 0> __device__ __forceinline__ void row16(float x, float &xd, float (&nu)[4], const float (&g)[4], unsigned long long msk)
{
    if constexpr (S < 16) {
        row_step<S>(x, xd, nu, g, msk);
        row16<S + 1>(x, xd, nu, g, msk);
    }
}
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float nu4[4] = {0.5f, 0.6f, 0.7f, 0.8f};
    const float g4[4] = {1e-3f, 2e-3f, 3e-3f, 4e-3f};
    unsigned long long msk = 0x0001000100010001ull;
    asm volatile("" : "+s"(msk));
    float xa = 1.0f + threadIdx.x * 1e-3f, nua = 0.5f, xda = 0.f, ga = 1e-3f, ta = 0.f;
    float xb = 2.0f + threadIdx.x * 1e-3f, nub = 0.7f, xdb = 0.f, gb = 2e-3f, tb = 0.f;
    int sa = 0, sb = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { // the product's step, 16 per iteration
#define S16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define M0(Q) "v_min_f32 %[ta], %[xa], %[nua]\n\ts_nop 0\n\tv_readlane_b32 %[sa], %[ta], " #Q "\n\ts_nop 1\n\tv_fma_f32 %[nua], -%[sa], %[ga], %[nua]\n\tv_writelane_b32 %[xda], %[sa], " #Q "\n\t"
            asm volatile(S16(M0) : [ta] "+v"(ta), [nua] "+v"(nua), [xda] "+v"(xda), [sa] "+s"(sa) : [xa] "v"(xa), [ga] "v"(ga));
        } else if (MODE == 1) { // no fillers (timing only)
#define M1(Q) "v_min_f32 %[ta], %[xa], %[nua]\n\tv_readlane_b32 %[sa], %[ta], " #Q "\n\tv_fma_f32 %[nua], -%[sa], %[ga], %[nua]\n\tv_writelane_b32 %[xda], %[sa], " #Q "\n\t"
            asm volatile(S16(M1) : [ta] "+v"(ta), [nua] "+v"(nua), [xda] "+v"(xda), [sa] "+s"(sa) : [xa] "v"(xa), [ga] "v"(ga));
        } else if (MODE == 2) { // two chains interleaved: min A, min B, readlane A, readlane B, writelane A' / B' fill the gap, fma A, fma B  (8 steps of each = 16 steps)
#define S8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define M2(Q) "v_min_f32 %[ta], %[xa], %[nua]\n\tv_min_f32 %[tb], %[xb], %[nub]\n\tv_readlane_b32 %[sa], %[ta], " #Q "\n\tv_readlane_b32 %[sb], %[tb], " #Q "\n\t" \
              "v_writelane_b32 %[xda], %[sa], " #Q "\n\tv_fma_f32 %[nua], -%[sa], %[ga], %[nua]\n\tv_writelane_b32 %[xdb], %[sb], " #Q "\n\tv_fma_f32 %[nub], -%[sb], %[gb], %[nub]\n\t"
            asm volatile(S8(M2) : [ta] "+v"(ta), [nua] "+v"(nua), [xda] "+v"(xda), [sa] "+s"(sa), [tb] "+v"(tb), [nub] "+v"(nub), [xdb] "+v"(xdb), [sb] "+s"(sb)
                         : [xa] "v"(xa), [ga] "v"(ga), [xb] "v"(xb), [gb] "v"(gb));
        } else if (MODE == 4) { // the row form (colsolve_row_kernel's step as the compiler emits it): four columns per wavefront, delta by DPP row broadcast
            // one "step" here = the steps of FOUR columns: v_min, v_cndmask, s_nop 1, v_mov_b32_dpp row_newbcast, 2 v_pk_fma_f32
            row16(xa, xda, nu4, g4, msk);
        } else { // the arithmetic alone: v_min + v_fma per step (no lane instructions)
#define M3(Q) "v_min_f32 %[ta], %[xa], %[nua]\n\tv_fma_f32 %[nua], -%[ta], %[ga], %[nua]\n\t"
            asm volatile(S16(M3) : [ta] "+v"(ta), [nua] "+v"(nua) : [xa] "v"(xa), [ga] "v"(ga));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = nua + xda + nub + xdb + ta + tb + nu4[0] + nu4[1] + nu4[2] + nu4[3];
}
template <int MODE> static int run(const char *name, int wps)
{
    float *out;
    const int blocks = 256 * wps; // wps blocks of 4 wavefronts per CU
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2500;
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    CK(hipGetLastError());
    // steps per SIMD = wps wavefronts x iters x 16
    printf("%-46s %d wavefronts/SIMD: %.3f ms = %.2f ns per step and SIMD (%.1f cycles at 2.1 GHz)\n", name, wps, best, 1e6 * best / ((double)wps * iters * 16),
           2.1e3 * best / ((double)wps * iters * 16) * 1e3 / 1e3);
    hipFree(out);
    return 0;
}
int main()
{
    for (int wps : {1, 2, 5, 7}) {
        run<0>("product step (min, nop, readlane, nop1, fma, writelane)", wps);
        run<1>("same without the wait-state fillers", wps);
        run<2>("two chains interleaved, no fillers", wps);
        run<3>("arithmetic alone (min, fma)", wps);
        if (wps <= 2) run<4>("row form: min, cndmask, nop1, dpp bcast, 2 pk_fma (4 columns)", wps);
    }
    return 0;
}
