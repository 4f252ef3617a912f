// How do fp64 VALU work and fp64 MFMA work of DIFFERENT wavefronts share a SIMD?  (not part of the product)
// One workgroup of W wavefronts (wave w sits on SIMD w % 4); role per wave chosen by a bit pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// role 0: idle, 1: independent fp64 FMAs, 2: fp64 MFMA 16x16x4 (independent accumulators), 3: fp32 FMAs, 4: dependent fp64 chain
__global__ void k_mix(double *out, long long *cyc, int n, unsigned long long roles)
{
    const int wave = threadIdx.x >> 6;
    const int role = (roles >> (4 * wave)) & 15;
    double b = 1.0000001, c = 1e-9;
    double a0 = out[threadIdx.x & 63], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float g0 = (float)a0, g1 = g0 + 1, g2 = g0 + 2, g3 = g0 + 3, g4 = g0 + 4, g5 = g0 + 5, g6 = g0 + 6, g7 = g0 + 7, fb = 1.0000001f, fc = 1e-9f;
    f64x4 t0 = {a0, a1, a2, a3}, t1 = t0, t2 = t0, t3 = t0;
    __syncthreads();
    long long w0 = wall_clock64(), s0 = __builtin_readcyclecounter();
    if (role == 1) {
        for (int i = 0; i < n; i++) {
            a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
            a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
        }
    } else if (role == 2) {
        for (int i = 0; i < n; i++) {
            t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, t0, 0, 0, 0); t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, t2, 0, 0, 0); t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, t3, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, t0, 0, 0, 0); t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, t2, 0, 0, 0); t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, t3, 0, 0, 0);
        }
    } else if (role == 3) {
        for (int i = 0; i < n; i++) {
            g0 = __builtin_fmaf(g0, fb, fc); g1 = __builtin_fmaf(g1, fb, fc); g2 = __builtin_fmaf(g2, fb, fc); g3 = __builtin_fmaf(g3, fb, fc);
            g4 = __builtin_fmaf(g4, fb, fc); g5 = __builtin_fmaf(g5, fb, fc); g6 = __builtin_fmaf(g6, fb, fc); g7 = __builtin_fmaf(g7, fb, fc);
        }
    } else if (role == 4) {
        for (int i = 0; i < n; i++) {
            a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c);
            a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c);
        }
    }
    long long w1 = wall_clock64(), s1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7 + t0[0] + t1[1] + t2[2] + t3[3];
    if ((threadIdx.x & 63) == 0) { cyc[2 * wave] = w1 - w0; cyc[2 * wave + 1] = s1 - s0; }
}

int main()
{
    double *d; long long *c, h[32];
    hipMalloc(&d, 1024 * 8); hipMemset(d, 0, 1024 * 8); hipMalloc(&c, 32 * 8);
    const int n = 100000;
    struct Case { const char *name; int waves; unsigned long long roles; } cases[] = {
        {"1 wave fp64 FMA", 1, 0x1ull},
        {"1 wave fp32 FMA", 1, 0x3ull},
        {"1 wave fp64 dependent chain", 1, 0x4ull},
        {"1 wave fp64 MFMA", 1, 0x2ull},
        {"4 waves (1/SIMD) fp64 FMA", 4, 0x1111ull},
        {"8 waves (2/SIMD) fp64 FMA", 8, 0x11111111ull},
        {"2 waves SAME SIMD (0,4) fp64 FMA", 5, 0x10001ull},
        {"2 waves SAME SIMD fp32 FMA", 5, 0x30003ull},
        {"2 waves SAME SIMD fp64 dependent chains", 5, 0x40004ull},
        {"2 waves SAME SIMD fp64 MFMA", 5, 0x20002ull},
        {"same SIMD: fp64 FMA (w0) + fp64 MFMA (w4)", 5, 0x20001ull},
        {"same SIMD: dependent chain (w0) + fp64 MFMA (w4)", 5, 0x20004ull},
        {"same SIMD: dependent chain (w0) + fp64 FMA (w4)", 5, 0x10004ull},
        {"same SIMD: chain (w0) + 2 MFMA waves (w4, w8)", 9, 0x200020004ull},
        {"different SIMDs: fp64 FMA (w0) + fp64 MFMA (w1)", 2, 0x21ull},
    };
    for (auto &cs : cases) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(c, 0, 32 * 8);
            k_mix<<<1, 64 * cs.waves>>>(d, c, n, cs.roles);
            hipMemcpy(h, c, 32 * 8, hipMemcpyDeviceToHost);
            if (rep == 0) continue;
            printf("%-52s", cs.name);
            for (int w = 0; w < cs.waves; w++) {
                const int role = (cs.roles >> (4 * w)) & 15;
                if (!role) continue;
                printf(" | w%d role%d %.2f ns/op (%.1f ticks)", w, role, 10.0 * h[2 * w] / (8.0 * n), (double)h[2 * w + 1] / (8.0 * n));
            }
            printf("\n");
        }
    }
    return 0;
}
