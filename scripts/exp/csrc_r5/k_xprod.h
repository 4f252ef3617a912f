// k_xprod.h -- the A-streaming skinny GEMM of a half-step in the strict fp64 mode (the fp32-operand mode runs k_xprod16.h).
//
// Reference: the per-column gemv `Wt * A.col(j)` inside update()'s OpenMP loop
// (src/update_with_missing.cpp:39,45) and, for the W half-step, the same on the materialised
// `A.t()` (src/nnmf.cpp:131 -- in EVERY iteration; here the transposed copy is made once per matrix):
//
//   xprod_tn : C[kq, j] = sum_i Y[kq, i] * A[i, j]   (contraction index contiguous in both operands)
//
// H half-step: A as stored, Y = W^T; W half-step: the transposed copy of A, Y = H.  (Round 1-2 ran the W half-step on
// the untransposed matrix with a strided-contraction "NT" kernel: 1.0 ms against 0.47 ms for the same bytes.)
// Split-K: block (x, s) contracts stage range s of tile x and writes an fp64 slab Cx[s][KP][ldc]; the consumer (sweep
// kernel, or the slab reduce ahead of the RCCL all-reduce) sums the slabs in a fixed order -> deterministic.
//
// MFMA: 16x16x4 (f32 or f64 inputs), one operand element per lane.  Tiles of A and of the factor
// go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip) through a ring of XPROD_NBUF stage
// buffers: two stages (64 KiB of A per CU) stay in flight while one is consumed, ordered by counted
// `s_waitcnt vmcnt(N)` + a raw s_barrier (a __syncthreads() would drain the queue).  A block is 8
// wavefronts = 2 per SIMD, each owning one 16-row M-tile of the stage: while one wave of a SIMD issues
// its LDS fragment reads / DMA, the other keeps the matrix pipe busy.  Fragments leave LDS as
// ds_read_b128.  With T = float partial sums are kept in f32 for at most
// XPROD_FLUSH_ELEMS contraction elements and then folded into fp64 accumulators (SURVEY.md section 7
// "precision ladder").
//
// All operands are zero padded to full tiles, so there are no bounds checks in the hot loop.
#pragma once
#include "common.h"

#define XPROD_THREADS 512
#define XPROD_WAVES 8
#define XPROD_FLUSH_ELEMS 256
#define XPROD_ROWB 256            // bytes per LDS row of the TN images (one 16-lane group per row)
#define XPROD_TN_BJ 128           // columns j per block (TN) = 8 waves x 16
#define XPROD_A_IMG_BYTES 32768   // A image per stage
#define XPROD_NBUF 3              // LDS stage buffers: one being consumed, two in flight from HBM

__host__ __device__ static inline int xprod_tn_lds_bytes(int KP) { return XPROD_NBUF * (XPROD_A_IMG_BYTES + KP * XPROD_ROWB); }

// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n: waits until at most n of this wavefront's vector-memory
// operations are outstanding.  global_load_lds completes in issue order, so "n = loads of the newest stage" means
// "every older stage has landed".
__device__ static inline void wait_vmcnt(int n)
{
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// global_load_lds_dwordx4 written out: scalar base (wave-uniform), one 32-bit per-lane byte offset, LDS base in M0.  The builtin
// form (glds16) takes a 64-bit per-lane address: 5-6 vector instructions per request to rebuild it (v_lshl_add_u64 x2,
// v_readfirstlane for M0, moves) -- a third of what a wavefront of the cross-product kernels issues per stage.
__device__ __forceinline__ unsigned long long xp_uniform64(const void *p)
{
    const unsigned long long v = (unsigned long long)p;
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void glds16_s(unsigned voff, unsigned long long sbase, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// number of indices u in {wave, wave + 8, ...} below cnt
__device__ static inline int strided_count(int wave, int cnt) { return (wave < cnt) ? (cnt - wave + XPROD_WAVES - 1) / XPROD_WAVES : 0; }

// ------------------------------------------------------------------------------------------------
// TN: contraction along i (contiguous in memory).
//   A      [mpad][lda]  (column j at j*lda), Yop [KP][ldy] (row kq, i fastest)
//   Cx     [S][KP][ldc] fp64, ldc >= mpad
//   grid   (mpad/128, S); stage = 256 bytes of contraction per row (64 f32 / 32 f64)
// LDS image per stage: rows of 256 B; row r holds its sixteen 16-byte slots XOR-swizzled
// (physical slot = logical slot ^ (r & 15)) so that the fragment reads (16 lanes = 16 different
// rows, same logical slot) are bank-conflict free.  global_load_lds writes LDS linearly, so the
// swizzle is applied to the per-lane GLOBAL source address (cdna_hip_programming.md rule 21).
// Wave w owns image rows (columns j) [16w, 16w+16): one M-tile x NKQ N-tiles.
// ------------------------------------------------------------------------------------------------
// EXP: ablation switches for scripts/exp/xprod_exp.hip only (0 in the product): bit0 load the factor image only for the
// first two stages, bit1 skip the MFMA phase, bit2 load the A image only for the first two stages, bit5 every wavefront issues its requests in
// front of its MFMA phase (the form before round 5).
typedef float xp_f32x2 __attribute__((ext_vector_type(2)));
typedef double xp_f64x2 __attribute__((ext_vector_type(2)));
template <typename T> struct XpVec2;
template <> struct XpVec2<float> { using type = xp_f32x2; };  // one v_pk_fma_f32 per pair
template <> struct XpVec2<double> { using type = xp_f64x2; };
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// KT > 0: the rank is 16*NKQ + (1..KT) -- the last KT (2 or 4) rows of the factor are NOT padded to a fourth 16-wide MFMA tile
// (k = 50 would issue 64/50 = 28 % more MFMAs); their dot products run as KT*EPV plain FMAs per fragment in the shadow of
// the MFMAs, on the A fragment the lane already holds, and are summed across the four lane groups in the epilogue.
template <typename T, int NKQ, int KT = 0, int EXP = 0>
__global__ __launch_bounds__(XPROD_THREADS) void xprod_tn_kernel(const T *__restrict__ A, int lda,
                                                                 const T *__restrict__ Yop, int ldy,
                                                                 double *__restrict__ Cx, int ldc, size_t slab_stride,
                                                                 int stage_begin, int stage_end, int stages_per_split)
{
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    constexpr int EPV = M::EPV;
    constexpr int KP = 16 * (NKQ + (KT > 0 ? 1 : 0)); // rows of the factor image (= the handle's KP)
    constexpr int CE = XPROD_ROWB / (int)sizeof(T); // contraction elements per stage
    constexpr int BUF = XPROD_A_IMG_BYTES + KP * XPROD_ROWB;
    constexpr int FL = XPROD_FLUSH_ELEMS / CE;
    constexpr int YI = (16 * NKQ + KT + 3) / 4;      // 4-row pieces of the factor image that are actually used
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int j0 = blockIdx.x * XPROD_TN_BJ;
    int st0 = stage_begin + blockIdx.y * stages_per_split;
    int st1 = st0 + stages_per_split;
    if (st1 > stage_end) st1 = stage_end;

    acc_t acc[NKQ];
    f64x4 acc64[NKQ];
#pragma unroll
    for (int b = 0; b < NKQ; b++) {
        acc[b] = acc_t{0, 0, 0, 0};
        acc64[b] = f64x4{0, 0, 0, 0};
    }
    using v2_t = typename XpVec2<T>::type;
    v2_t tacc[KT > 0 ? KT : 1]; // even / odd contraction elements
    double tacc64[KT > 0 ? KT : 1];
#pragma unroll
    for (int u = 0; u < (KT > 0 ? KT : 1); u++) {
        tacc[u] = v2_t{0, 0};
        tacc64[u] = 0.0;
    }

    // Piece t = wave + 8 i of an image = rows 4 t + lg: (row & 15) = (4 wave + lg) & 15 for every i, so ONE per-lane byte offset per
    // image serves all of a wavefront's requests; piece and stage go into the scalar base (glds16_s).
    const int rw = 4 * wave + lg, sw = l15 ^ (rw & 15);
    const unsigned voffA = (unsigned)(((size_t)rw * lda + sw * EPV) * sizeof(T)), voffY = (unsigned)(((size_t)rw * ldy + sw * EPV) * sizeof(T));
    const unsigned long long baseA = xp_uniform64(A + (size_t)j0 * lda), baseY = xp_uniform64(Yop);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    auto issue = [&](int st, unsigned char *buf) {
        const unsigned long long bA = (unsigned long long)(((EXP & 4) && st > st0 + 1) ? st0 : st) * XPROD_ROWB;        // EXP: re-read a cached stage
        const unsigned long long bY = (unsigned long long)(((EXP & 5) == 5 && st > st0 + 1) ? st0 : st) * XPROD_ROWB;
        const unsigned dst = lds0 + (unsigned)(buf - smem) + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < XPROD_A_IMG_BYTES / 1024 / XPROD_WAVES; i++)
            glds16_s(voffA, baseA + bA + (unsigned long long)i * 32ull * (unsigned long long)lda * sizeof(T), dst + (unsigned)i * 8192u);
        if ((EXP & 1) && st > st0 + 1) return;
#pragma unroll
        for (int i = 0; i < (YI + XPROD_WAVES - 1) / XPROD_WAVES; i++)
            if (wave + XPROD_WAVES * i < YI)
                glds16_s(voffY, baseY + bY + (unsigned long long)i * 32ull * (unsigned long long)ldy * sizeof(T),
                         dst + (unsigned)XPROD_A_IMG_BYTES + (unsigned)i * 8192u);
    };

    // loads THIS wavefront issues per stage (A image: 32 instructions over 8 waves; factor image: KP/4 instructions)
    int per_stage = XPROD_A_IMG_BYTES / 1024 / XPROD_WAVES + strided_count(wave, YI);
    if (EXP & 1) per_stage = XPROD_A_IMG_BYTES / 1024 / XPROD_WAVES;
    if (st0 < st1) issue(st0, smem);
    if (st0 + 1 < st1) issue(st0 + 1, smem + BUF);
    // fp64 (round 5): wavefronts 4..7 hand their requests of stage st + 2 to the memory pipeline BEHIND their MFMA phase, wavefronts 0..3
    // in front of it.  A wavefront sits in the issue queue until the pipeline takes its 5-6 KB (most of a stage's HBM time when all eight
    // ask at once) and issues no MFMA meanwhile; with the two wavefronts of a SIMD asking at different times one of them computes --
    // 0.392 -> 0.377 ms (H) / 0.399 -> 0.385 (W) at config 2 (scripts/exp/xprod64_exp.hip).  The split-fp16 kernel's MFMA phase is a
    // quarter of this one's and gains nothing (k_xprod16.h, EXP bit 3).  Same buffers, same counted waits: requests complete in issue
    // order per wavefront, and stage st + 2's buffer is free from the barrier of stage st on.
    const bool late = sizeof(T) == 8 && !(EXP & (32 | 16)) && wave >= XPROD_WAVES / 2;
    int since_flush = 0;
    for (int st = st0; st < st1; ++st) {
        unsigned char *buf = smem + ((st - st0) % XPROD_NBUF) * BUF;
        // stage `st` has landed once at most the newer stage's loads are outstanding; the barrier then also tells
        // every wave that stage st-1 has been consumed, so its buffer may be refilled with stage st+2
        if constexpr ((EXP & 16) == 0) {
            wait_vmcnt((st + 1 < st1) ? per_stage : 0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (st + 2 < st1 && !late) issue(st + 2, smem + ((st + 2 - st0) % XPROD_NBUF) * BUF);
        }
        if (EXP & 2) continue;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int phys = ((lg + 4 * kk) ^ l15) * 16;
            T a[EPV], b[NKQ][EPV];
            if constexpr ((EXP & 8) != 0) { // EXP: fragments from registers, no LDS reads
#pragma unroll
                for (int e = 0; e < EPV; e++) {
                    a[e] = (T)(lane + e + kk);
#pragma unroll
                    for (int nt = 0; nt < NKQ; nt++) b[nt][e] = (T)(lane - e + nt);
                }
            } else {
            {
                const int row = 16 * wave + l15;
                const f32x4 raw = *(const f32x4 *)(buf + row * XPROD_ROWB + phys);
                __builtin_memcpy(a, &raw, 16);
            }
#pragma unroll
            for (int nt = 0; nt < NKQ; nt++) {
                const int row = 16 * nt + l15;
                const f32x4 raw = *(const f32x4 *)(buf + XPROD_A_IMG_BYTES + row * XPROD_ROWB + phys);
                __builtin_memcpy(b[nt], &raw, 16);
            }
            }
            T w[KT > 0 ? KT : 1][EPV];
            if constexpr (KT > 0) { // tail rows 16*NKQ + u: same 16-byte slot of the image row, uniform over the 16 lanes of a group
#pragma unroll
                for (int u = 0; u < KT; u++) {
                    const f32x4 raw = *(const f32x4 *)(buf + XPROD_A_IMG_BYTES + (16 * NKQ + u) * XPROD_ROWB + (((lg + 4 * kk) ^ u) * 16));
                    __builtin_memcpy(w[u], &raw, 16);
                }
            }
#pragma unroll
            for (int e = 0; e < EPV; e++)
#pragma unroll
                for (int nt = 0; nt < NKQ; nt++) acc[nt] = M::mma(a[e], b[nt][e], acc[nt]);
            if constexpr (KT > 0) {
#pragma unroll
                for (int e = 0; e < EPV; e += 2)
#pragma unroll
                    for (int u = 0; u < KT; u++)
                        tacc[u] = __builtin_elementwise_fma(v2_t{a[e], a[e + 1]}, v2_t{w[u][e], w[u][e + 1]}, tacc[u]);
            }
        }
        if (late && st + 2 < st1) issue(st + 2, smem + ((st + 2 - st0) % XPROD_NBUF) * BUF);
        if constexpr (sizeof(T) == 4) {
            if (++since_flush == FL) {
                since_flush = 0;
#pragma unroll
                for (int b = 0; b < NKQ; b++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) acc64[b][r] += (double)acc[b][r];
                    acc[b] = acc_t{0, 0, 0, 0};
                }
                if constexpr (KT > 0) {
#pragma unroll
                    for (int u = 0; u < KT; u++) {
                        tacc64[u] += (double)tacc[u][0] + (double)tacc[u][1];
                        tacc[u] = v2_t{0, 0};
                    }
                }
            }
        }
    }
    // epilogue: D[M = j-row, N = kq]
    double *out = Cx + (size_t)blockIdx.y * slab_stride;
#pragma unroll
    for (int nt = 0; nt < NKQ; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int kq = 16 * nt + l15;
            const int j = j0 + 16 * wave + M::row_of(lane, r);
            double v;
            if constexpr (sizeof(T) == 4) v = acc64[nt][r] + (double)acc[nt][r];
            else v = acc[nt][r];
            out[(size_t)kq * ldc + j] = v;
        }
    if constexpr (KT > 0) { // lane (l15, lg) holds the partial of row j = 16*wave + l15 over its quarter of the contraction
#pragma unroll
        for (int u = 0; u < KT; u++) {
            double v = tacc64[u] + ((double)tacc[u][0] + (double)tacc[u][1]);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lg == 0) out[(size_t)(16 * NKQ + u) * ldc + j0 + 16 * wave + l15] = v;
        }
    }
}
