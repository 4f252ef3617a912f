// common.h -- shared device helpers for libnnlm_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NNLM_TINY 1e-16 // TINY_NUM, reference src/nnlm.h:17

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Geometry of the resident layouts (see DESIGN.md "Data layout in HBM").
//   A_dev  : [mpad][npad] elements of T, column j of A starts at j*npad (zero padded)
//   npad   : n rounded up to NNLM_PAD_N (tile of the A*H^T kernel along i)
//   mpad   : m rounded up to NNLM_PAD_M (tile of the W^T*A kernel along j)
#define NNLM_PAD_N 256
#define NNLM_PAD_M 128
#define NNLM_KQ_MAX 64 // fast path: rank padded to KP = 16*NKQ <= 64

static inline int round_up_i(int x, int q) { return (x + q - 1) / q * q; }

// ---- MFMA wrappers: 16x16x4, one element of A and B per lane ---------------------------------
//   A operand: lane l holds A[M = l&15][K = l>>4];  B operand: lane l holds B[K = l>>4][N = l&15]
//   f32 C/D: lane l, reg r -> D[M = 4*(l>>4)+r][N = l&15]
//   f64 C/D: lane l, reg r -> D[M = (l>>4)+4*r][N = l&15]   (cdna_hip_programming.md section 3)
template <typename T> struct Mfma;
template <> struct Mfma<float> {
    typedef f32x4 acc_t;
    static constexpr int EPV = 4; // elements per 16-byte LDS read
    __device__ static inline acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static inline int row_of(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <> struct Mfma<double> {
    typedef f64x4 acc_t;
    static constexpr int EPV = 2;
    __device__ static inline acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static inline int row_of(int lane, int r) { return (lane >> 4) + 4 * r; }
};

// 16-byte direct global->LDS copy (global_load_lds_dwordx4): per-lane global source, LDS
// destination = wave-uniform base + lane*16.
__device__ static inline void glds16(const void *gsrc, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// ln x for a NORMAL positive fp32 x: v_log_f32 (log2, 1 ulp, one quarter-rate instruction) times ln 2.  logf() / __logf() compile
// to a ~20-instruction sequence here (denormal rescaling, two-term ln 2 product, special cases) -- measured as half of the error
// block's time.  The error sums take ln(ahat + 1e-16): never denormal; a relative error of 1e-7 per term averages out over 2e8.
#define NNLM_LN2F 0.69314718055994531f
__device__ static inline float log2_native(float x) { return __builtin_amdgcn_logf(x); }

// The reference's relative-change test  2|x_new - x| / (x_new + x + eps) > rel_tol  (src/base_algorithms.cpp:29-35), decided exactly
// like its rounded quotient without a division on the common path: away from the boundary  2|d| > tol * s  is the same decision (both
// roundings are below 2 ulp); inside a band of 1e-15 relative the IEEE quotient is formed and compared, as the reference does.
__device__ static inline bool rel_change_exceeds(double d2, double s, double tol)
{
    const double rhs = tol * s;
    if (__builtin_expect(fabs(d2 - rhs) <= 1e-15 * fabs(rhs), 0)) return d2 / s > tol;
    return d2 > rhs;
}

__device__ static inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ static inline long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
