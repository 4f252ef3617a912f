// Standalone timing + check of the fused cross product / error block (k_xprod16.h xprod16_err_kernel) and of the experimental forms in
// k_xerr.h -- not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o xerr_exp xerr_exp.hip
//   ./xerr_exp n m variant [reps] [S]  variant 0: xerr0_kernel<4> (rounds 1-4), 50: xprod16_err_kernel<4> (the product), 1: xprod16_tn_kernel<4>,
//                                      2: the same with half of the wavefronts issuing late, 3..: k_xerr.h (xerr_launch)
// Inputs are synthetic split-fp16 images written by a device kernel (values as the product's: hi in [2^14, 2^15) at most).  Check: the
// cross product of every variant against variant 0 bit for bit, the two error sums to 1e-12 relative.
#include "csrc_r5/k_xprod16.h"
#include "k_xerr.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ static inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// split rows [rows][len/64][2][64] of uniform values in (0, vmax) (pre-scaled), entries beyond (rows_true, len_true) zero
__global__ void fill_split(uint32_t *X, size_t rows, size_t len, size_t rows_true, size_t len_true, float vmax, unsigned seed)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * len) return;
    const size_t r = idx / len, i = idx % len;
    const float v = (r < rows_true && i < len_true) ? vmax * (float)(hash32((unsigned)idx * 2654435761u + seed) >> 8) * (1.0f / 16777216.0f) : 0.f;
    _Float16 hi, lo;
    split16(v, 1.0f, hi, lo);
    _Float16 *row = (_Float16 *)(X + r * len + (i >> 6) * 64);
    row[i & 63] = hi;
    row[64 + (i & 63)] = lo;
}
// kq-contiguous copy [cols][2][64] of a split-row factor [64][len/64][2][64]
__global__ void to_kqc(const uint32_t *Y16, size_t len, uint32_t *Xc)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; // over len * 64
    if (idx >= len * 64) return;
    const size_t c = idx / 64, q = idx % 64;
    const _Float16 *row = (const _Float16 *)(Y16 + q * len + (c >> 6) * 64);
    _Float16 *dst = (_Float16 *)(Xc + c * 64);
    dst[q] = row[c & 63];
    dst[64 + q] = row[64 + (c & 63)];
}

// reads (and drops) the first `xst` stages of every block of a (tiles x S) plan of the cross product: rows of 256 bytes per stage
// and image row, so that they are in the memory-side cache when the cross product starts
__global__ __launch_bounds__(256) void mall_prefetch_kernel(const uint32_t *__restrict__ A16, int lda, int tiles, int S, int sps, int stages, int xst, int nblk, unsigned *sink)
{
    // one request = 16 bytes per lane; a wavefront covers 4 rows x 256 bytes
    const size_t total = (size_t)nblk * xst * 32; // 1 KB pieces: 32 per block and stage
    unsigned acc = 0;
    for (size_t pc = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); pc < total; pc += (size_t)gridDim.x * 4) {
        const int piece = (int)(pc % 32), st = (int)((pc / 32) % xst), blk = (int)(pc / 32 / xst);
        const int tile = blk % tiles, slab = blk / tiles;
        const int stage = slab * sps + st;
        if (stage >= stages) continue;
        const int lane = threadIdx.x & 63, row = tile * 128 + piece * 4 + (lane >> 4);
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 v = *(const u4 *)(A16 + (size_t)row * lda + (size_t)stage * 64 + (lane & 15) * 4);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 20000, m = argc > 2 ? atoi(argv[2]) : 10000, variant = argc > 3 ? atoi(argv[3]) : 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 10;
    const int npad = (n + 127) / 128 * 128, mpad = (m + 63) / 64 * 64, KP = 64, k = 50;
    const int tiles_x = npad / 128, stages = mpad / 64;
    int S = argc > 5 ? atoi(argv[5]) : 3;
    const int sps = (stages + S - 1) / S;
    S = (stages + sps - 1) / sps;
    uint32_t *A16T, *Y16, *H16c, *W16, *W16c; double *Cx, *Cx0, *partial, *partial0; int *scal; unsigned long long *tim;
    CK(hipMalloc(&A16T, (size_t)npad * mpad * 4)); CK(hipMalloc(&Y16, (size_t)KP * mpad * 4)); CK(hipMalloc(&H16c, (size_t)mpad * 64 * 4));
    CK(hipMalloc(&W16, (size_t)KP * npad * 4)); CK(hipMalloc(&W16c, (size_t)npad * 64 * 4));
    CK(hipMalloc(&Cx, (size_t)S * KP * npad * 8)); CK(hipMalloc(&Cx0, (size_t)S * KP * npad * 8));
    CK(hipMalloc(&partial, (size_t)2 * tiles_x * S * 8)); CK(hipMalloc(&partial0, (size_t)2 * tiles_x * S * 8)); CK(hipMalloc(&scal, 16));
    CK(hipMalloc(&tim, 64 * 8 * 16 * 8)); CK(hipMemset(tim, 0, 64 * 8 * 16 * 8));
    const int sc[4] = {15, 18, 17, 0}; // A = U(0,1) * 2^15; H = U(0, 1/8) * 2^18; W = U(0, 1/4) * 2^17  (W H of order 0.4)
    CK(hipMemcpy(scal, sc, 16, hipMemcpyHostToDevice));
    fill_split<<<(unsigned)(((size_t)npad * mpad + 255) / 256), 256>>>(A16T, npad, mpad, n, m, 32767.f, 1u);
    fill_split<<<(unsigned)(((size_t)KP * mpad + 255) / 256), 256>>>(Y16, KP, mpad, k, m, 32767.f, 2u);
    fill_split<<<(unsigned)(((size_t)KP * npad + 255) / 256), 256>>>(W16, KP, npad, k, n, 32767.f, 3u);
    to_kqc<<<(unsigned)(((size_t)mpad * 64 + 255) / 256), 256>>>(Y16, mpad, H16c);
    to_kqc<<<(unsigned)(((size_t)npad * 64 + 255) / 256), 256>>>(W16, npad, W16c);
    CK(hipDeviceSynchronize());
    dim3 grid(tiles_x, S);
    hipEvent_t eA, eB; CK(hipEventCreate(&eA)); CK(hipEventCreate(&eB));
    const size_t slab = (size_t)KP * npad;
    auto launch = [&](int v, double *C, double *P) -> int {
        if (v == 0) {
            const int lds = 2 * XPROD16_ERR_BUF;
            CK(hipFuncSetAttribute((const void *)xerr0_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            xerr0_kernel<4><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, H16c, W16c, C, npad, slab, 0, stages, sps, scal, scal + 2, n, m, P);
        } else if (v == 50) { // the product's kernel
            const int lds = xprod16_err_lds_bytes(4);
            CK(hipFuncSetAttribute((const void *)xprod16_err_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            xprod16_err_kernel<4><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, H16c, W16c, C, npad, slab, 0, stages, sps, scal, scal + 2, n, m, P, nullptr);
        } else if (v == 1) {
            const int lds = xprod_tn_lds_bytes(KP);
            CK(hipFuncSetAttribute((const void *)xprod16_tn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            xprod16_tn_kernel<4><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, C, npad, slab, 0, stages, sps, scal);
        } else if (v >= 70 && v <= 73) { // the memory-side cache as a prefetch target: 70 pass over another buffer, then the timed pass; 71-73: + prefetch of 16 / 24 / 32 stages
            static uint32_t *A2 = nullptr; static unsigned *sink = nullptr;
            if (!A2) { CK(hipMalloc(&A2, (size_t)npad * mpad * 4)); CK(hipMemset(A2, 0, (size_t)npad * mpad * 4)); CK(hipMalloc(&sink, 4)); }
            const int lds = xprod_tn_lds_bytes(KP);
            CK(hipFuncSetAttribute((const void *)xprod16_tn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            xprod16_tn_kernel<4><<<grid, XPROD_THREADS, lds>>>(A2, mpad, Y16, mpad, C, npad, slab, 0, stages, sps, scal);
            if (v > 70) mall_prefetch_kernel<<<1024, 256>>>(A16T, mpad, tiles_x, S, sps, stages, 8 + 8 * (v - 70), 256, sink);
            CK(hipEventRecord(eA));
            xprod16_tn_kernel<4><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, C, npad, slab, 0, stages, sps, scal);
            CK(hipEventRecord(eB));
        } else if (v == 60 || v == 61) { // 60: ascending and descending passes alternate over the same buffer; 61: two ascending passes (control)
            const int lds = xprod_tn_lds_bytes(KP);
            CK(hipFuncSetAttribute((const void *)xprod16_tn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            CK(hipFuncSetAttribute((const void *)xprod16_tn_kernel<4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            xprod16_tn_kernel<4><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, C, npad, slab, 0, stages, sps, scal);
            if (v == 60) xprod16_tn_kernel<4, 16><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, C, npad, slab, 0, stages, sps, scal);
            else xprod16_tn_kernel<4><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, C, npad, slab, 0, stages, sps, scal);
        } else if (v == 2) {
            const int lds = xprod_tn_lds_bytes(KP);
            CK(hipFuncSetAttribute((const void *)xprod16_tn_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            xprod16_tn_kernel<4, 8><<<grid, XPROD_THREADS, lds>>>(A16T, mpad, Y16, mpad, C, npad, slab, 0, stages, sps, scal);
        } else {
            return xerr_launch(v, grid, A16T, mpad, Y16, H16c, W16c, C, npad, slab, stages, sps, scal, n, m, P, tim);
        }
        return 0;
    };
    CK(hipMemset(Cx0, 0, (size_t)S * KP * npad * 8)); CK(hipMemset(Cx, 0, (size_t)S * KP * npad * 8));
    if (launch(0, Cx0, partial0)) return 1;
    CK(hipDeviceSynchronize());
    if (launch(variant, Cx, partial)) return 1;
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 10; r++) if (launch(variant, Cx, partial)) return 1; // warm-up (clocks)
    std::vector<float> tms;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0));
        if (launch(variant, Cx, partial)) return 1;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (variant >= 70 && variant <= 73) CK(hipEventElapsedTime(&ms, eA, eB)); // (the second pass alone)
        tms.push_back(ms);
    }
    std::sort(tms.begin(), tms.end());
    printf("variant %d n %d m %d grid %d x %d sps %d: best %.4f ms  median %.4f ms\n", variant, n, m, tiles_x, S, sps, tms[0], tms[tms.size() / 2]);
    std::vector<double> c0((size_t)S * KP * npad), c1(c0.size()), p0((size_t)2 * tiles_x * S), p1(p0.size());
    CK(hipMemcpy(c0.data(), Cx0, c0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), Cx, c1.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p0.data(), partial0, p0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), partial, p1.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < c0.size(); i++) bad += (c0[i] != c1[i]);
    double s0[2] = {0, 0}, s1[2] = {0, 0};
    for (size_t i = 0; i < p0.size(); i++) s0[i & 1] += p0[i], s1[i & 1] += p1[i];
    printf("  cross product: %zu of %zu differ from variant 0; sums %.15g %.15g vs %.15g %.15g (rel %.2e %.2e)\n", bad, c0.size(), s1[0], s1[1], s0[0], s0[1],
           fabs(s1[0] - s0[0]) / fabs(s0[0]), fabs(s1[1] - s0[1]) / fabs(s0[1]));
    if (variant >= 3) {
        std::vector<unsigned long long> t(64 * 8 * 16);
        CK(hipMemcpy(t.data(), tim, t.size() * 8, hipMemcpyDeviceToHost));
        if (t[15]) {
            printf("  cycles per stage (s_memtime, 100 MHz ticks x 1 -- see header), block 0 / wave 0..7 and block 200 / wave 0:\n");
            for (int w = 0; w < 9; w++) {
                const unsigned long long *r = &t[(w < 8 ? w : 8 * 5) * 16];
                printf("   ");
                for (int i = 0; i < 8; i++) printf(" %8.1f", (double)r[i] / (double)(r[15] ? r[15] : 1));
                printf("   stages %llu\n", r[15]);
            }
        }
    }
    return 0;
}
