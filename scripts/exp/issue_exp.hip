// Issue-rate probes for the fp32 sweep (k_sweep_f.h): what ONE wavefront per SIMD can issue on gfx950 (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o issue_exp issue_exp.hip ; ./issue_exp
// Every case: 256 workgroups x 256 threads (one wavefront per SIMD) -- or WPS wavefronts per SIMD -- run ITER iterations of a body;
// reported: shader cycles per iteration from s_memtime deltas of wavefront 0 (and wall time / 2.4 GHz for comparison).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)

// CASE: 0 = 16 independent v_fma_f32; 1 = 16 dependent v_fma_f32; 2 = 16 dependent v_min_f32/v_fma_f32 alternating;
//       3 = 4 MFMA (independent accumulators); 4 = 4 MFMA + NV independent VALU spread between them; 5 = MFMA -> v_min -> MFMA (same accumulator) dependent pair x 4
//       6 = 3 permlane swaps (transposition) dependent on a VALU; 7 = 16 independent v_pk_fma_f32; 8 = 4 MFMA each followed by 4 DEPENDENT VALU (chain across the iteration)
template <int CASE, int NV> __global__ __launch_bounds__(256) void k_issue(float *out, long long *cyc, int iters)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = out[threadIdx.x] + i;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = f32x4{a[i], a[i + 1], a[i + 2], a[i + 3]};
    const float fb = 1.0000001f, fc = 1e-9f;
    float ch = a[0];
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if constexpr (CASE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(fb), "v"(fc));
        } else if constexpr (CASE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ch) : "v"(fb), "v"(fc));
        } else if constexpr (CASE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(ch) : "v"(a[1]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ch) : "v"(fb), "v"(fc));
            }
        } else if constexpr (CASE == 3) {
            MF(acc[0], a[4], a[5]); MF(acc[1], a[4], a[5]); MF(acc[2], a[4], a[5]); MF(acc[3], a[4], a[5]);
        } else if constexpr (CASE == 4) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                SB();
                MF(acc[m], a[4], a[5]);
                SB();
#pragma unroll
                for (int i = 0; i < NV; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[6 + (i & 7)]) : "v"(fb), "v"(fc));
            }
        } else if constexpr (CASE == 5) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                MF(acc[0], a[4], ch);
                ch = __builtin_fminf(a[7], acc[0][0]);
            }
        } else if constexpr (CASE == 6) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const float v0 = ch * fb, v1 = ch + fc, v2 = ch - fc, v3 = ch * 0.5f;
                const auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
                const auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
                const auto q = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
                ch = __uint_as_float(q[0]);
            }
        } else if constexpr (CASE == 7) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 8; i++) {
                f32x2 p = {a[2 * i], a[2 * i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(f32x2{fb, fb}), "v"(f32x2{fc, fc}));
                a[2 * i] = p[0], a[2 * i + 1] = p[1];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                f32x2 p = {a[2 * i], a[2 * i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(f32x2{fb, fb}), "v"(f32x2{fc, fc}));
                a[2 * i] = p[0], a[2 * i + 1] = p[1];
            }
        } else if constexpr (CASE == 9) { // 4 x (bf16 16x16x32 MFMA, NV independent VALU)
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8 va, vb;
#pragma unroll
            for (int i = 0; i < 8; i++) va[i] = (__bf16)a[i], vb[i] = (__bf16)a[8 + i];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                SB();
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[m], 0, 0, 0);
                SB();
#pragma unroll
                for (int i = 0; i < NV; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[6 + (i & 7)]) : "v"(fb), "v"(fc));
            }
        } else if constexpr (CASE == 10) { // 4 x (bf16 16x16x32 MFMA, NV VALU of one dependent chain)
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8 va, vb;
#pragma unroll
            for (int i = 0; i < 8; i++) va[i] = (__bf16)a[i], vb[i] = (__bf16)a[8 + i];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                SB();
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[m], 0, 0, 0);
                SB();
#pragma unroll
                for (int i = 0; i < NV; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ch) : "v"(fb), "v"(fc));
            }
        } else if constexpr (CASE == 11) { // 4 x (f64 4x4x4 MFMA x 3, NV dependent f32 VALU)
            double da = a[4], db = a[5];
            static double dacc[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                SB();
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(*(double *)&acc[m]) : "v"(da), "v"(db));
                SB();
#pragma unroll
                for (int i = 0; i < NV; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ch) : "v"(fb), "v"(fc));
            }
        } else if constexpr (CASE == 8) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                SB();
                MF(acc[m], a[4], a[5]);
                SB();
#pragma unroll
                for (int i = 0; i < NV; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ch) : "v"(fb), "v"(fc));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = ch;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CASE, int NV> static int run(const char *what, int wps)
{
    float *out; long long *cyc;
    const int nwg = 256 * wps, iters = 20000;
    CK(hipMalloc(&out, (size_t)nwg * 256 * 4)); CK(hipMalloc(&cyc, 8));
    CK(hipMemset(out, 0, (size_t)nwg * 256 * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k_issue<CASE, NV><<<nwg, 256>>>(out, cyc, iters);
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    }
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-78s wps=%d: %7.1f counter ticks / iteration, wall %7.1f cycles @2.4GHz / iteration\n", what, wps, (double)c / iters, ms * 1e-3 * 2.4e9 / iters);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main()
{
    for (int wps = 1; wps <= 2; wps++) {
        run<0, 0>("16 independent v_fma_f32", wps);
        run<1, 0>("16 dependent v_fma_f32", wps);
        run<2, 0>("8 x (v_min_f32, v_fma_f32) dependent", wps);
        run<7, 0>("16 independent v_pk_fma_f32", wps);
        run<3, 0>("4 MFMA 16x16x4 f32, independent accumulators", wps);
        run<4, 1>("4 x (MFMA, 1 independent VALU)", wps);
        run<4, 2>("4 x (MFMA, 2 independent VALU)", wps);
        run<4, 3>("4 x (MFMA, 3 independent VALU)", wps);
        run<4, 4>("4 x (MFMA, 4 independent VALU)", wps);
        run<4, 5>("4 x (MFMA, 5 independent VALU)", wps);
        run<4, 6>("4 x (MFMA, 6 independent VALU)", wps);
        run<4, 8>("4 x (MFMA, 8 independent VALU)", wps);
        run<8, 2>("4 x (MFMA, 2 VALU of one dependent chain)", wps);
        run<8, 3>("4 x (MFMA, 3 VALU of one dependent chain)", wps);
        run<8, 4>("4 x (MFMA, 4 VALU of one dependent chain)", wps);
        run<8, 5>("4 x (MFMA, 5 VALU of one dependent chain)", wps);
        run<8, 6>("4 x (MFMA, 6 VALU of one dependent chain)", wps);
        run<9, 0>("4 bf16 16x16x32 MFMA, independent accumulators", wps);
        run<9, 2>("4 x (bf16 MFMA, 2 independent VALU)", wps);
        run<9, 4>("4 x (bf16 MFMA, 4 independent VALU)", wps);
        run<9, 6>("4 x (bf16 MFMA, 6 independent VALU)", wps);
        run<9, 8>("4 x (bf16 MFMA, 8 independent VALU)", wps);
        run<10, 2>("4 x (bf16 MFMA, 2 VALU of one dependent chain)", wps);
        run<10, 4>("4 x (bf16 MFMA, 4 VALU of one dependent chain)", wps);
        run<10, 6>("4 x (bf16 MFMA, 6 VALU of one dependent chain)", wps);
        run<11, 0>("4 f64 4x4x4 MFMA, independent accumulators", wps);
        run<11, 2>("4 x (f64 4x4x4 MFMA, 2 dependent f32 VALU)", wps);
        run<11, 4>("4 x (f64 4x4x4 MFMA, 4 dependent f32 VALU)", wps);
        run<5, 0>("4 x (MFMA -> v_min on its result -> B operand of the next MFMA, same accumulator)", wps);
        run<6, 0>("4 x (4 VALU -> permlane16_swap x2 -> permlane32_swap) dependent", wps);
    }
    return 0;
}
