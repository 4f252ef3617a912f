// VALU issue rates on gfx950 as a function of wavefronts per SIMD (not part of the product): what bounds the KL solvers.
// Every case runs 256 x wps workgroups of 256 threads (one wavefront per SIMD each), NI instructions of one kind per
// iteration on 8-16 independent registers; reported: shader cycles per wave-instruction per SIMD (2.4 GHz assumed) and
// the aggregate rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ __launch_bounds__(256) void k_rate(float *out, int n)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = out[threadIdx.x] + i;
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = a[i];
    const float fb = 1.0000001f, fc = 1e-9f;
    const f32x2 pb = {fb, fb}, pc = {fc, fc};
    const double db = 1.0000001, dc = 1e-9;
    for (int it = 0; it < n; it++) {
        if (OP == 0) { // v_fma_f32 x 16
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(fb), "v"(fc));
        } else if (OP == 1) { // v_pk_fma_f32 x 8 (16 FMAs per lane)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        } else if (OP == 2) { // v_rcp_f32 x 16
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if (OP == 3) { // v_fma_f64 x 8
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
        } else if (OP == 4) { // v_rcp_f64 x 8
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        } else if (OP == 5) { // Lee-KL element body, scalar: y = fma(c, wp, y); r = rcp(y); t = b r; s = fma(w, t, s)   (8 elements)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float r, t;
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(fc), "v"(fb));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(a[i]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(fb), "v"(r));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[8 + i]) : "v"(fc), "v"(t));
            }
        } else if (OP == 6) { // the same with packed fma / mul / fma (8 elements = 4 pairs) and scalar rcp
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x2 r, t;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pc), "v"(pb));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(r.x) : "v"(p[i].x));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(r.y) : "v"(p[i].y));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(pb), "v"(r));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[4 + i]) : "v"(pc), "v"(t));
            }
        } else if (OP == 7) { // v_mul_f32 x 16 (plain, for the price of a non-fma op)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fb));
        } else if (OP == 8) { // v_pk_mul_f32 x 8
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y + (float)d[i];
    out[threadIdx.x] = s;
}

// LDS read rates beside VALU: ds_read_b128 x 8 per iteration
template <int W> __global__ __launch_bounds__(256) void k_lds(float *out, int n)
{
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = i;
    __syncthreads();
    float acc = 0;
    const float4 *s4 = (const float4 *)sm;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float4 v = s4[(threadIdx.x + 256 * i + it) & 2047];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    out[threadIdx.x] = acc;
}

template <int OP> static void run(const char *name, int per_iter, float *d)
{
    const int n = 20000;
    for (int wps : {1, 2, 4, 8}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k_rate<OP><<<256 * wps, 256>>>(d, 100);
        hipEventRecord(e0);
        k_rate<OP><<<256 * wps, 256>>>(d, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)n * per_iter * wps);
        printf("%-34s wps %d: %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", name, wps, ms, cyc);
    }
}

int main()
{
    float *d;
    hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    run<0>("v_fma_f32", 16, d);
    run<1>("v_pk_fma_f32", 8, d);
    run<7>("v_mul_f32", 16, d);
    run<8>("v_pk_mul_f32", 8, d);
    run<2>("v_rcp_f32", 16, d);
    run<3>("v_fma_f64", 8, d);
    run<4>("v_rcp_f64", 8, d);
    run<5>("KL body scalar (per element)", 8, d);
    run<6>("KL body packed (per element)", 8, d);
    for (int wps : {1, 2, 4}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int n = 20000;
        k_lds<0><<<256 * wps, 256, 32768>>>(d, 100);
        hipEventRecord(e0);
        k_lds<0><<<256 * wps, 256, 32768>>>(d, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("ds_read_b128 + 4 adds, wps %d: %.3f ms, %.2f cycles per read per SIMD, %.1f B/clk/CU\n", wps, ms,
               ms * 1e-3 * 2.4e9 / ((double)n * 8 * wps), 4.0 * wps * 8 * n * 1024.0 / (ms * 1e-3 * 2.4e9));
    }
    return 0;
}
