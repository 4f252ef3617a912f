// k_xprod64_rs.h -- a tried form of the fp64 cross product (scripts/exp/xprod64_exp.hip), not part of the product: measured 0.44-0.48 ms
// at config 2 with random data against 0.42-0.45 for xprod_tn_kernel<double> with its late issue (one computing wavefront per SIMD does
// not keep the fp64 matrix pipe as busy as two that take turns).
#pragma once
#include "csrc_r5/k_xprod.h"

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in 0..12
__device__ static inline void xp_wait_vmcnt_rt(int n)
{
    switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// ------------------------------------------------------------------------------------------------
// The fp64 cross product with wavefront roles (round 5; same images, ring and results as xprod_tn_kernel<double>):
//   * wavefronts 0..3 ("C") own 32 columns each and issue every MFMA of the block -- one wavefront per SIMD keeps the fp64 matrix
//     pipe busy (a v_mfma_f64_16x16x4 occupies it for 64 cycles), and a factor fragment is read once per 32 columns;
//   * wavefronts 4..7 ("L") issue every request of the block and nothing else: a wavefront that hands requests to the memory pipeline
//     sits in the issue queue until they are taken -- for most of a stage when HBM is the bound -- and issues nothing meanwhile.
// Column tile 2 rg + mt of this kernel = wavefront 2 rg + mt of xprod_tn_kernel, same order of accumulation: bit-identical output.
// ------------------------------------------------------------------------------------------------
template <int NKQ, int KT = 0>
__global__ __launch_bounds__(XPROD_THREADS) void xprod64_rs_kernel(const double *__restrict__ A, int lda, const double *__restrict__ Yop, int ldy,
                                                                   double *__restrict__ Cx, int ldc, size_t slab_stride, int stage_begin, int stage_end,
                                                                   int stages_per_split)
{
    constexpr int KP = 16 * (NKQ + (KT > 0 ? 1 : 0));
    constexpr int BUF = XPROD_A_IMG_BYTES + KP * XPROD_ROWB;
    constexpr int YI = (16 * NKQ + KT + 3) / 4; // 4-row pieces of the factor image that are actually used
    constexpr int LW = XPROD_WAVES / 2;          // wavefronts per role
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int rg = wave & (LW - 1);
    const int j0 = blockIdx.x * XPROD_TN_BJ;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;

    if (wave >= LW) {
        // ---------------------------------------------------------------- L: requests.  Piece t = rg + 4 i of an image = rows 4 t + lg
        const int rw = 4 * rg + lg, sw = l15 ^ (rw & 15);
        const unsigned voffA = (unsigned)(((size_t)rw * lda + sw * 2) * 8), voffY = (unsigned)(((size_t)rw * ldy + sw * 2) * 8);
        const unsigned long long baseA = xp_uniform64(A + (size_t)j0 * lda), baseY = xp_uniform64(Yop);
        const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
        constexpr int A_REQ = XPROD_A_IMG_BYTES / 1024 / LW;
        const int y_req = (rg < YI) ? (YI - rg + LW - 1) / LW : 0;
        auto issue = [&](int st) {
            const unsigned long long b0 = (unsigned long long)st * XPROD_ROWB;
            const unsigned dst = lds0 + (unsigned)((st - st0) % XPROD_NBUF) * (unsigned)BUF + (unsigned)rg * 1024u;
#pragma unroll
            for (int i = 0; i < A_REQ; i++) glds16_s(voffA, baseA + b0 + (unsigned long long)i * 16ull * (unsigned long long)lda * 8ull, dst + (unsigned)i * 4096u);
#pragma unroll
            for (int i = 0; i < (YI + LW - 1) / LW; i++)
                if (rg + LW * i < YI)
                    glds16_s(voffY, baseY + b0 + (unsigned long long)i * 16ull * (unsigned long long)ldy * 8ull, dst + (unsigned)XPROD_A_IMG_BYTES + (unsigned)i * 4096u);
        };
        const int per_stage = A_REQ + y_req;
        if (st0 < st1) issue(st0);
        if (st0 + 1 < st1) issue(st0 + 1);
        for (int st = st0; st < st1; ++st) {
            xp_wait_vmcnt_rt((st + 1 < st1) ? per_stage : 0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (st + 2 < st1) issue(st + 2);
        }
        return;
    }
    // -------------------------------------------------------------------- C: MFMAs of column tiles 2 rg, 2 rg + 1
    f64x4 acc[2][NKQ];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int b = 0; b < NKQ; b++) acc[mt][b] = f64x4{0, 0, 0, 0};
    xp_f64x2 tacc[2][KT > 0 ? KT : 1]; // even / odd contraction elements of the tail rows
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int u = 0; u < (KT > 0 ? KT : 1); u++) tacc[mt][u] = xp_f64x2{0, 0};
    for (int st = st0; st < st1; ++st) {
        const unsigned char *buf = smem + ((st - st0) % XPROD_NBUF) * BUF;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int phys = ((lg + 4 * kk) ^ l15) * 16;
            double a[2][2], b[NKQ][2];
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                const f32x4 raw = *(const f32x4 *)(buf + (32 * rg + 16 * mt + l15) * XPROD_ROWB + phys);
                __builtin_memcpy(a[mt], &raw, 16);
            }
#pragma unroll
            for (int nt = 0; nt < NKQ; nt++) {
                const f32x4 raw = *(const f32x4 *)(buf + XPROD_A_IMG_BYTES + (16 * nt + l15) * XPROD_ROWB + phys);
                __builtin_memcpy(b[nt], &raw, 16);
            }
            double w[KT > 0 ? KT : 1][2];
            if constexpr (KT > 0) {
#pragma unroll
                for (int u = 0; u < KT; u++) {
                    const f32x4 raw = *(const f32x4 *)(buf + XPROD_A_IMG_BYTES + (16 * NKQ + u) * XPROD_ROWB + (((lg + 4 * kk) ^ u) * 16));
                    __builtin_memcpy(w[u], &raw, 16);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int nt = 0; nt < NKQ; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
                if constexpr (KT > 0) {
#pragma unroll
                    for (int u = 0; u < KT; u++) tacc[mt][u] = __builtin_elementwise_fma(xp_f64x2{a[mt][0], a[mt][1]}, xp_f64x2{w[u][0], w[u][1]}, tacc[mt][u]);
                }
            }
        }
    }
    double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
        for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) out[(size_t)(16 * nt + l15) * ldc + j0 + 32 * rg + 16 * mt + Mfma<double>::row_of(lane, r)] = acc[mt][nt][r];
        if constexpr (KT > 0) { // lane (l15, lg) holds the partial of column 32 rg + 16 mt + l15 over its quarter of the contraction
#pragma unroll
            for (int u = 0; u < KT; u++) {
                double v = tacc[mt][u][0] + tacc[mt][u][1];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (lg == 0) out[(size_t)(16 * NKQ + u) * ldc + j0 + 32 * rg + 16 * mt + l15] = v;
            }
        }
    }
}
