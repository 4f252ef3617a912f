// Ablation micro-benchmark of sweep_ls_kernel (not part of the product).  hipcc --offload-arch=gfx950 -O3 -o sweep_exp sweep_exp.hip
#include "csrc_r5/k_sweep.h"
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R, int L, int EXP>
static float run(const SweepArgs &a, int reps)
{
    const int cpw = 64 / L, nb = (a.ncols + cpw - 1) / cpw;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    sweep_ls_kernel<R, L, 1, EXP><<<nb, 64>>>(a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) sweep_ls_kernel<R, L, 1, EXP><<<nb, 64>>>(a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int k = 50, KP = 64;
    const int ncols = argc > 1 ? atoi(argv[1]) : 10000;
    const int ld = (ncols + 255) / 256 * 256;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0, 1);
    std::vector<double> G(KP * KP, 0.0), X((size_t)KP * ld, 0.0), C((size_t)KP * ld, 0.0), W((size_t)k * 2000);
    for (auto &w : W) w = U(rng);
    for (int q = 0; q < k; q++) for (int r = 0; r < k; r++) { double s = 0; for (int i = 0; i < 2000; i++) s += W[q * 2000 + i] * W[r * 2000 + i]; G[q * KP + r] = s; }
    for (int q = 0; q < k; q++) for (int c = 0; c < ncols; c++) { X[(size_t)q * ld + c] = U(rng); C[(size_t)q * ld + c] = 500 * U(rng); }
    double *dG, *dX, *dC; unsigned long long *dS;
    CK(hipMalloc(&dG, G.size() * 8)); CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dC, C.size() * 8)); CK(hipMalloc(&dS, 8));
    CK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(dS, 0, 8));
    SweepArgs a{};
    a.X = dX; a.ldx = ld; a.Graw = dG; a.KPg = KP; a.Cx = dC; a.slab_stride = (size_t)KP * ld; a.nslabs = 1; a.ldc = ld;
    a.ncols = ncols; a.k = k; a.r0 = a.r1 = a.r2 = 0; a.mask = nullptr; a.max_iter = 50; a.rel_tol = -1.0; a.op = nullptr; a.op_mode = 0; a.sweeps = dS;
    printf("ncols=%d k=%d 50 sweeps\n", ncols, k);
#define RUN(R, L, E) printf("  R=%2d L=%d EXP=%2d : %.3f ms\n", R, L, E, run<R, L, E>(a, 5));
    RUN(14, 4, 0) RUN(14, 4, 1) RUN(14, 4, 2) RUN(14, 4, 3) RUN(14, 4, 4) RUN(14, 4, 8) RUN(14, 4, 15) RUN(14, 4, 7)
    RUN(28, 2, 0) RUN(28, 2, 1) RUN(28, 2, 15)
    a.max_iter = 1;
    printf(" 1 sweep:\n");
    RUN(14, 4, 0)
    a.max_iter = 0;
    printf(" 0 sweeps:\n");
    RUN(14, 4, 0)
    return 0;
}
