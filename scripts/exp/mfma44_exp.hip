// v_mfma_f64_4x4x4_4b_f64 on gfx950: operand layout (found by one-hot probing) and cost next to VALU / DPP work of the
// same wavefront.  (not part of the product; decides the design of the one-wavefront SCD sweep, k_sweep_q.h)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void k_layout(int *dl)
{
    // for every (la, lb): lane of D that receives A[la] * B[lb] (or -1)
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0, c = 0.0;
            c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
            unsigned long long bal = __ballot(c != 0.0);
            if (lane == 0) dl[la * 64 + lb] = bal ? (__builtin_popcountll(bal) == 1 ? __builtin_ctzll(bal) : 100 + __builtin_popcountll(bal)) : -1;
        }
}

// mode 0: 13 independent MFMAs per iteration; 1: one dependent MFMA chain; 2: 13 MFMAs + 1 VALU op on accumulator 0 (dependent);
// 3: 16 independent v_mov_b32_dpp; 4: 8 dependent (v_max_f64 -> dpp mov pair -> v_fma_f64) rounds; 5: 16 independent fp64 FMAs;
// 6: 13 MFMAs + 20 independent VALU FMAs interleaved by the compiler; 7: 13 MFMAs then 20 VALU
__global__ __launch_bounds__(256, 2) void k_time(double *out, long long *cyc, int n, int mode)
{
    const int lane = threadIdx.x & 63;
    double acc[13], a = out[lane] + 1.0, b = 1e-9;
#pragma unroll
    for (int i = 0; i < 13; i++) acc[i] = out[lane] + i;
    double v[20];
#pragma unroll
    for (int i = 0; i < 20; i++) v[i] = out[lane] + 0.5 * i;
    __syncthreads();
    long long s0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if (mode == 0) {
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 13; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
    } else if (mode == 1) {
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 13; i++) acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[0], 0, 0, 0);
        }
    } else if (mode == 2) {
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 13; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
            asm volatile("v_max_f64 %0, -%1, -%2" : "=v"(a) : "v"(acc[0]), "v"(a));
        }
    } else if (mode == 3) {
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = lane + i;
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_amdgcn_update_dpp(x[i], x[(i + 5) & 15], 0x114, 0xf, 0x2, false);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) a += x[i];
    } else if (mode == 4) {
        double x = out[lane], m = a;
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                double d;
                asm volatile("v_max_f64 %0, -%1, -%2" : "=v"(d) : "v"(x), "v"(m));
                int lo = __builtin_amdgcn_update_dpp(__double2loint(d), __double2loint(d), 0x114, 0xf, 0xe, false);
                int hi = __builtin_amdgcn_update_dpp(__double2hiint(d), __double2hiint(d), 0x114, 0xf, 0xe, false);
                m = __builtin_fma(__hiloint2double(hi, lo), b, m);
            }
        }
        a = m;
    } else if (mode == 5) {
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = __builtin_fma(v[i], a, b);
        }
    } else if (mode == 6) {
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 13; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 20; i++) v[i] = __builtin_fma(v[i], a, b);
        }
    } else if (mode == 7) {
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 13; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 20; i++) v[i] = __builtin_fma(v[i], a, b);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (mode == 8) { // 16x16x4 for comparison: 4 independent
        typedef double f64x4 __attribute__((ext_vector_type(4)));
        f64x4 t[4];
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = f64x4{acc[i], acc[i + 4], acc[i + 8], a};
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, t[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) a += t[i][0] + t[i][3];
    }
    else if (mode == 9) { // 13 independent MFMAs, 13 distinct A operands, one B
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 13; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[i], b, acc[i], 0, 0, 0);
        }
    } else if (mode == 10) { // distinct A and B
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 13; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[i], v[19 - (i % 7)], acc[i], 0, 0, 0);
        }
    } else if (mode == 11) { // out of place: D != C
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 13; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[i], b, acc[(i + 1) % 13], 0, 0, 0);
        }
    } else if (mode == 12) { // the step: 4 x (v_max, MFMA, dependent MFMA, MFMA, MFMA)
        double x = out[lane] + 0.25, m = a;
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                double c;
                asm volatile("v_max_f64 %0, -%1, -%2" : "=v"(c) : "v"(x), "v"(m));
                __builtin_amdgcn_sched_barrier(0);
                acc[3 * s] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[3 * s], b, acc[3 * s], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                m = __builtin_amdgcn_mfma_f64_4x4x4f64(v[12], c, acc[12], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc[3 * s + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[3 * s + 1], b, acc[3 * s + 1], 0, 0, 0);
                acc[3 * s + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[3 * s + 2], b, acc[3 * s + 2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        a = m;
    }
    else if (mode == 13) { // strict KL element body x 4: d = y + eps, quotient (rcp, 2 Newton, residual), fma into the sum, refresh
        double num[4] = {v[0], v[1], v[2], v[3]}, yv[4] = {v[4] + 1, v[5] + 1, v[6] + 1, v[7] + 1}, acc4[2] = {0, 0};
        for (int it = 0; it < n; it++) {
            double den[4], r[4], t[4], q[4];
#pragma unroll
            for (int i = 0; i < 4; i++) den[i] = yv[i] + 1e-16;
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = __builtin_amdgcn_rcp(den[i]);
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = __builtin_fma(-den[i], r[i], 1.0);
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = __builtin_fma(t[i], r[i], r[i]);
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = __builtin_fma(-den[i], r[i], 1.0);
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = __builtin_fma(t[i], r[i], r[i]);
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = num[i] * r[i];
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = __builtin_fma(-den[i], q[i], num[i]);
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = __builtin_fma(t[i], r[i], q[i]);
            acc4[0] = __builtin_fma(b, q[0], acc4[0]); acc4[1] = __builtin_fma(b, q[1], acc4[1]);
            acc4[0] = __builtin_fma(b, q[2], acc4[0]); acc4[1] = __builtin_fma(b, q[3], acc4[1]);
#pragma unroll
            for (int i = 0; i < 4; i++) yv[i] = __builtin_fma(b, num[i], yv[i]);
        }
        a = acc4[0] + acc4[1] + yv[0] + yv[1] + yv[2] + yv[3];
    } else if (mode == 14) { // 16 independent v_rcp_f64
        for (int it = 0; it < n; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = __builtin_amdgcn_rcp(v[i]);
        }
    }
    long long s1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = a;
#pragma unroll
    for (int i = 0; i < 13; i++) s += acc[i];
#pragma unroll
    for (int i = 0; i < 20; i++) s += v[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = s1 - s0; cyc[1] = w1 - w0; }
}

int main()
{
    int *dl, hl[4096];
    hipMalloc(&dl, 4096 * 4);
    k_layout<<<1, 64>>>(dl);
    hipMemcpy(hl, dl, 4096 * 4, hipMemcpyDeviceToHost);
    // hypotheses: lane = 16*blk + 4*x + y
    for (int ha = 0; ha < 2; ha++)
        for (int hb = 0; hb < 2; hb++)
            for (int hd = 0; hd < 2; hd++) {
                int ok = 1;
                for (int la = 0; la < 64 && ok; la++)
                    for (int lb = 0; lb < 64 && ok; lb++) {
                        const int ba = la >> 4, xa = (la >> 2) & 3, ya = la & 3, bb = lb >> 4, xb = (lb >> 2) & 3, yb = lb & 3;
                        const int i = ha ? xa : ya, ka = ha ? ya : xa, kb = hb ? yb : xb, j = hb ? xb : yb;
                        int want = -1;
                        if (ba == bb && ka == kb) want = 16 * ba + (hd ? 4 * j + i : 4 * i + j);
                        if (hl[la * 64 + lb] != want) ok = 0;
                    }
                if (ok)
                    printf("LAYOUT: blk = lane>>4;  A: %s;  B: %s;  D: lane&15 = %s\n", ha ? "i=(l>>2)&3 k=l&3" : "i=l&3 k=(l>>2)&3",
                           hb ? "k=l&3 j=(l>>2)&3" : "k=(l>>2)&3 j=l&3", hd ? "4j+i" : "4i+j");
            }
    printf("raw (la: lb->ld) for la = 0, 1, 4, 5, 16:\n");
    for (int la : {0, 1, 4, 5, 16}) {
        printf(" la %2d:", la);
        for (int lb = 0; lb < 64; lb++)
            if (hl[la * 64 + lb] >= 0) printf(" %d->%d", lb, hl[la * 64 + lb]);
        printf("\n");
    }
    double *d;
    long long *c, h[2];
    hipMalloc(&d, 4096 * 64 * 8);
    hipMemset(d, 0, 4096 * 64 * 8);
    hipMalloc(&c, 16);
    const char *names[] = {"13 indep MFMA 4x4x4", "13 dependent MFMA 4x4x4", "13 MFMA + dependent v_max", "16 indep dpp mov", "8 x (max->2 dpp->fma) dependent",
                           "16 indep fp64 FMA", "13 MFMA + 20 FMA (free order)", "13 MFMA then 20 FMA", "4 indep MFMA 16x16x4", "13 indep MFMA, distinct A", "13 indep MFMA, distinct A and B", "13 MFMA out of place", "step: 4 x (max, L, chain, L, L)", "strict KL body x 4 elements", "16 indep v_rcp_f64"};
    const int per[] = {13, 13, 14, 16, 8, 16, 33, 33, 4, 13, 13, 13, 20, 4, 16};
    for (int wpb : {1, 2}) // waves per SIMD on the measured CU: blocks of 256 x wpb threads
        for (int mode = 0; mode < 15; mode++) {
            const int n = 20000;
            for (int rep = 0; rep < 2; rep++) {
                k_time<<<1, 64 * (wpb == 1 ? 1 : 5)>>>(d, c, n, mode); // 5 waves: wave 0 and wave 4 share SIMD 0
                hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            }
            printf("[%s] %-34s: %.1f ticks/iter, %.2f ticks/op  (%.1f ns/iter)\n", wpb == 1 ? "lone wave " : "5 waves/CU", names[mode], (double)h[0] / n,
                   (double)h[0] / n / per[mode], 10.0 * h[1] / n);
        }
    // chip-wide: the step pattern (mode 12) and plain MFMAs (mode 0) on every SIMD, 1 / 2 / 3 wavefronts per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode : {13, 14})
        for (int grid : {1, 64, 256, 512, 768}) {
            const int n = 20000; float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                k_time<<<grid, 256>>>(d, c, n, mode);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            printf("[grid %3d x 256] %-34s: kernel %.3f ms = %.1f ns/iter; block 0 wave 0: %.1f ticks/iter\n", grid, names[mode], ms, 1e6 * ms / n, (double)h[0] / n);
        }
    return 0;
}
