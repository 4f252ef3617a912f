/*
 * nnlm_mi355x.h -- C ABI of libnnlm_mi355x.so: the MI355X (gfx950) implementation of the hot
 * path of the R package linxihui/NNLM (alternating nnmf() loop + single-solve nnlm()).
 *
 * This header is the drop-in boundary.  The reference's FFI for this path is the pair of
 * registered .Call routines `_NNLM_c_nnmf` (17 SEXP args) and `_NNLM_c_nnlm` (9 SEXP args)
 * (reference src/RcppExports.cpp:10-27, :29-54, :56-65; called from R/RcppExports.R:4-10).
 * `nnlm_c_nnmf()` / `nnlm_c_nnlm()` below take exactly those arguments as plain pointers and
 * sizes and return exactly the members of the reference's named result lists
 * (src/nnmf.cpp:211-219, src/nnlm.cpp:49-52).  pkg/src/r_glue.c shows the Rinternals-only
 * .Call stub a maintainer adds on the R side; nnlm_amd/_lib.py is the ctypes binding used here.
 *
 * Conventions (all taken from the reference):
 *   - every matrix is column-major (R / Armadillo), fp64 at the boundary;
 *   - logical masks are `int` arrays (R LGLSXP), non-zero = masked (entry is never updated) -- NA_LOGICAL (INT_MIN) included:
 *     the reference converts the matrix to arma::umat and tests `mask(k) > 0` (src/RcppExports.cpp:38-39, src/base_algorithms.cpp:21);
 *   - missing entries of A / y are any non-finite value (NA, NaN, +-Inf), src/nnmf.cpp:65-68;
 *   - method: 1 scd+mse, 2 lee+mse, 3 scd+mkl, 4 lee+mkl (R/misc.R:28-35);
 *   - alpha/beta: [L2, angle, L1] (src/nnmf.cpp:19-20).
 * Inputs are never written.  No exceptions cross this ABI: every function returns NNLM_OK or an
 * error code and leaves a message retrievable with nnlm_last_error().
 */
#ifndef NNLM_MI355X_H
#define NNLM_MI355X_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNLM_ABI_VERSION 1

/* return codes */
#define NNLM_OK 0
#define NNLM_ERR_ARG 1       /* invalid argument (the reference would throw from Armadillo / R stop()) */
#define NNLM_ERR_HIP 2       /* HIP runtime failure (no device, out of memory, launch failure) */
#define NNLM_ERR_INTERRUPT 3 /* the check_interrupt callback asked to stop (Rcpp::checkUserInterrupt) */
#define NNLM_ERR_COMM 4      /* RCCL failure */
#define NNLM_ERR_UNSUPPORTED 5

/* arithmetic modes: what A is stored as in HBM and which MFMA the cross-products run on.
 * Gram matrices, mu = G*x - c, the coordinate sweeps and all reductions are fp64 in both modes. */
#define NNLM_PREC_F32 0 /* A fp32 in HBM (4 bytes per element); the A-streaming cross products take their operands as split-fp16 pairs
                         * (hi + lo * 2^-11: 22 significant bits) on v_mfma_f32_16x16x32_f16, fp32 partial sums flushed into fp64
                         * every 256 elements; KL solvers keep their state in fp32 */
#define NNLM_PREC_F64 1 /* A and GEMM operands fp64, v_mfma_f64_16x16x4_f64 (strict-parity mode; the default of the one-shot entries) */

/*
 * Host callbacks = the R API points the reference touches from its main thread.
 * Any member (or the whole struct pointer) may be NULL.
 */
typedef struct nnlm_callbacks {
    void *ctx;
    int (*check_interrupt)(void *ctx);                        /* src/nnmf.cpp:111; non-zero aborts the run */
    void (*progress)(void *ctx, unsigned done, unsigned total); /* RcppProgress increment, src/nnmf.cpp:60,112 (verbose==1) */
    void (*print)(void *ctx, const char *text);               /* Rprintf, src/nnmf.cpp:100-104,155-156,188-189,194-198 (verbose==2) */
    void (*warning)(void *ctx, const char *text);             /* Rcpp::warning, src/nnmf.cpp:208-209 */
    double (*unif_rand)(void *ctx);                           /* R's RNG behind arma::randu, src/nnmf.cpp:84,94; src/nnlm.cpp:39 */
} nnlm_callbacks;

/* ------------------------------------------------------------------------------------------
 * One-shot entries (what `.Call("_NNLM_c_nnmf", ...)` / `.Call("_NNLM_c_nnlm", ...)` bind to)
 * ---------------------------------------------------------------------------------------- */

/* Length the four trace vectors must have: ceil(max_iter/trace)+1 (src/nnmf.cpp:53-54). */
unsigned nnlm_trace_capacity(unsigned max_iter, unsigned trace);

/*
 * Replaces c_nnmf (reference src/nnmf.cpp:4-220; signature src/RcppExports.cpp:29-51).
 *   A        n x m, const, may contain non-finite = missing
 *   k        rank K (already includes known-profile columns, R/misc.R:84)
 *   W_init   n x k initial W, or NULL for the default 0.01*U(0,1) init (src/nnmf.cpp:82-88)
 *   H_init   k x m initial H, or NULL (src/nnmf.cpp:92-98)
 *   Wm, Hm   n x k / k x m logical masks, or NULL when empty (src/nnmf.cpp:75-80)
 *   n_threads accepted for signature compatibility; the GPU path ignores it
 * Outputs (caller-allocated): W_out n x k, H_out k x m, four traces of nnlm_trace_capacity()
 * doubles each with *n_trace entries used (src/nnmf.cpp:200-206), *n_iteration (src/nnmf.cpp:218),
 * *warned = 1 iff the reference would have raised "Target tolerance not reached. Try a larger
 * max.iter." (src/nnmf.cpp:208-209; the text is also passed to cb->warning).
 */
int nnlm_c_nnmf(const double *A, int n, int m, unsigned k,
                const double *W_init, const double *H_init, const int *Wm, const int *Hm,
                const double alpha[3], const double beta[3],
                unsigned max_iter, double rel_tol, int n_threads, int verbose, int show_warning,
                unsigned inner_max_iter, double inner_rel_tol, int method, unsigned trace,
                double *W_out, double *H_out,
                double *mse_error, double *mkl_error, double *target_error, double *average_epoch,
                int *n_trace, unsigned *n_iteration, int *warned,
                const nnlm_callbacks *cb);

/*
 * Replaces c_nnlm (reference src/nnlm.cpp:4-53; signature src/RcppExports.cpp:10-27).
 *   x n x p, y n x q (may contain missing), mask p x q or NULL, beta0 p x q or NULL (-> U(0,1) init).
 * Outputs: coefficient p x q, *n_iteration = summed per-column sweeps (src/nnlm.cpp:44-51).
 */
int nnlm_c_nnlm(const double *x, const double *y, int n, int p, int q,
                const double alpha[3], const int *mask, const double *beta0,
                unsigned max_iter, double rel_tol, int n_threads, int method,
                double *coefficient, int *n_iteration, const nnlm_callbacks *cb);

/* ------------------------------------------------------------------------------------------
 * Resident API: the same path with A kept in HBM across calls.  The one-shot entries are thin
 * wrappers over it; bench.py and the parity tests of single half-steps use it directly.
 * ---------------------------------------------------------------------------------------- */
typedef struct nnlm_handle nnlm_handle;

/* device = HIP device ordinal; precision = NNLM_PREC_*.  Fails loudly when no gfx950 device exists. */
int nnlm_create(nnlm_handle **out, int device, int precision);
void nnlm_destroy(nnlm_handle *h);
/* Process-wide caches: the streams / events / small buffers of the last destroyed handle wait for the next nnlm_create on the same
 * device, and nnlm_set_matrix keeps its two pinned bounce buffers (up to 2 x 64 MB of pinned host memory).  Both are released at exit;
 * an embedder that unloads the library earlier (the R package's .onUnload, pkg/src/r_glue.c) or wants the memory back calls this.
 * Handles in use are not affected. */
int nnlm_release_caches(void);
const char *nnlm_last_error(const nnlm_handle *h); /* h may be NULL: error of the last failed nnlm_create / one-shot call */
int nnlm_abi_version(void);

/* Upload A (n x m, fp64, column-major; never written).  One pass on the device converts to the resident layout,
 * finds the non-finite entries -- NA, NaN, +Inf and -Inf alike are "missing" (src/nnmf.cpp:65-69) -- and sums the constant
 * KL part (src/nnmf.cpp:70,73).  The matrix is streamed through pinned staging buffers filled by a few host threads.
 * NNLM_PREC_F32 only: NNLM_ERR_UNSUPPORTED when A holds a finite entry beyond FLT_MAX or its largest entry is below 2^-100
 * (the 4-byte resident copy cannot represent it; NNLM_PREC_F64 takes such a matrix, as the reference does). */
int nnlm_set_matrix(nnlm_handle *h, const double *A, int n, int m);
/* Number of finite entries of A (N_non_missing, src/nnmf.cpp:51,69) and the any_missing flag. */
int nnlm_matrix_info(nnlm_handle *h, double *n_non_missing, int *any_missing, double *kl_const);

/* Set rank, factors (W n x k, H k x m; NULL = zeros) and masks (NULL = none). */
int nnlm_set_factors(nnlm_handle *h, unsigned k, const double *W, const double *H, const int *Wm, const int *Hm);
int nnlm_get_factors(nnlm_handle *h, double *W, double *H);

/*
 * One half-step = update()/update_with_missing() (reference src/update_with_missing.cpp:3-55, :58-139).
 * which = 0 updates W (solves A^T ~ H^T W^T with `reg` = alpha), 1 updates H (`reg` = beta).
 * Asynchronous on the handle's stream; the integer sweep count is accumulated on the device.
 */
int nnlm_half_step(nnlm_handle *h, int which, const double reg[3], unsigned inner_max_iter,
                   double inner_rel_tol, int method);
/* n_iter outer iterations (W half-step then H half-step, src/nnmf.cpp:114-133), asynchronous. */
int nnlm_iterate(nnlm_handle *h, unsigned n_iter, const double alpha[3], const double beta[3],
                 unsigned inner_max_iter, double inner_rel_tol, int method);
/* The alternating loop of c_nnmf (reference src/nnmf.cpp:100-209) on the resident matrix and factors: same arguments,
 * traces, stopping rule, warning and callbacks as nnlm_c_nnmf, without the upload.  nnlm_c_nnmf is
 * create + set_matrix + set_factors + nnlm_run + get_factors; bench.py times this call. */
int nnlm_run(nnlm_handle *h, const double alpha[3], const double beta[3], unsigned max_iter, double rel_tol, int verbose,
             int show_warning, unsigned inner_max_iter, double inner_rel_tol, int method, unsigned trace,
             double *mse_error, double *mkl_error, double *target_error, double *average_epoch, int *n_trace,
             unsigned *n_iteration, int *warned, const nnlm_callbacks *cb);
/* Summed per-column sweeps since the last reset (total_raw_iter, src/nnmf.cpp:106,158); synchronises. */
int nnlm_take_sweeps(nnlm_handle *h, long long *sweeps, int reset);
/* Error block (src/nnmf.cpp:121-126,135-140): mse = mean((A-WH)^2), mkl_var = mean(-(A+eps)log(WH+eps)+WH)
 * over finite entries, plus the penalty sums add_penalty() needs (src/nnmf.cpp:224-240):
 * pen[0..2] = sum(W^2), sum over i of (sum_q W[i,q])^2, sum(W); pen[3..5] the same for H.  Synchronises. */
int nnlm_errors(nnlm_handle *h, double *mse, double *mkl_var, double pen[6]);
int nnlm_sync(nnlm_handle *h);

/* Per-kernel device timing (HIP events on the handle's stream) for bench.py's roofline block.
 * names: "xprod_h" (A-streaming W^T A), "xprod_w" (A H^T), "xprod_w_err" (the same with the fused error sums), "gram", "sweep_h",
 * "sweep_w", "errors" (a separate pass over A), "err_reduce" (reduction of the fused error sums), "allgather", "allreduce", "unpack". */
int nnlm_profile_enable(nnlm_handle *h, int on);
int nnlm_profile_get(nnlm_handle *h, const char *name, double *total_ms, long long *launches);
int nnlm_profile_reset(nnlm_handle *h);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU (one process per GPU, RCCL over xGMI).  A is replicated.  The column of the factor being
 * solved is the unit: per half-step a rank forms the cross product of ITS 1/N of the columns over the
 * whole contraction (1/N of A from HBM), the Gram of the fixed factor (dense: one shared k x k Gram,
 * replicated; missing values: one per column), solves its columns into a packed slab, and ONE
 * ncclAllGather returns the updated factor to every rank (NNLM_FORM_COLS, the default).  Dense square loss also
 * has the form north_star words (NNLM_FORM_REDUCE, chosen with nnlm_comm_set_form): each rank contracts
 * its slab of rows (H half-step) / columns (W half-step), ONE ncclAllReduce sums the partial
 * [Gram | cross-product] buffer, then the column-sharded sweep and the all-gather -- same HBM bytes and
 * kernel time per rank (profiles/r02_shard_times.json), one more collective of (KP^2 + KP cols) doubles.
 * Error block: each rank reduces its share of A, two doubles are all-reduced.
 * ---------------------------------------------------------------------------------------- */
#define NNLM_COMM_ID_BYTES 128
#define NNLM_FORM_COLS 0   /* column-sharded half-steps, one all-gather each (every method) */
#define NNLM_FORM_REDUCE 1 /* dense square loss, rank <= 64: contraction-sharded [G | C] + all-reduce, then sweep + all-gather */
int nnlm_comm_unique_id(char id[NNLM_COMM_ID_BYTES]); /* rank 0 creates, the host layer broadcasts */
int nnlm_comm_init(nnlm_handle *h, const char id[NNLM_COMM_ID_BYTES], int rank, int nranks);
/* id == NULL makes a "virtual rank": the handle only computes rank's slab of an nranks-way split and leaves the
 * partial sums un-reduced (used with nnlm_debug_partial to test the shard arithmetic on one device). */
/* Form of the dense square-loss half-step across ranks (NNLM_FORM_*); nnlm_comm_init resets it to NNLM_FORM_COLS.  The other
 * half-steps (missing values, KL, rank > 64) are column-sharded whatever is set here. */
int nnlm_comm_set_form(nnlm_handle *h, int form);
int nnlm_comm_info(nnlm_handle *h, int *rank, int *nranks);
/* Contraction range [begin, end) owned by `rank` of `nranks`: rows i of A for the H half-step (which = 1), columns j
 * for the W half-step (which = 0).  Pure function of the sizes (no device needed). */
int nnlm_shard_range(int n, int m, int precision, int which, int rank, int nranks, int *begin, int *end);
/* Columns [col0, col1) of the factor being solved (m columns of H for which = 1, n rows of A = columns of W^T for which = 0) that
 * `rank` solves, and the width cpr of the packed slab [k][cpr] every rank contributes to the all-gather.  Pure function. */
int nnlm_shard_cols(int ncols, int rank, int nranks, int *cpr, int *col0, int *col1);
/* Test hook: partial [Gram k x k | cross product k x cols] of this (virtual) rank's slab, column-major, before the
 * all-reduce and before the regularisation edits of src/update_with_missing.cpp:20-24. */
int nnlm_debug_partial(nnlm_handle *h, int which, double *G_out, double *C_out);

/* Test hooks for virtual ranks (several handles in one process, nnlm_comm_init(h, NULL, r, P)): nnlm_debug_phase runs ONE
 * phase of a sharded half-step -- 1: cross product + Gram of the rank's contraction slab folded into [G | C];
 * 2: sweep of the rank's columns into its packed slab; 3: unpack of the gathered slabs -- and nnlm_debug_exchange does,
 * through the host, what ncclAllReduce (stage 1) / ncclAllGather (stage 2) do between them on a real node. */
int nnlm_debug_phase(nnlm_handle *h, int which, int phase, const double reg[3], unsigned inner_max_iter,
                     double inner_rel_tol, int method);
int nnlm_debug_exchange(nnlm_handle **handles, int nranks, int which, int stage);
/* The nnlm_debug_* entries are TEST HOOKS, not part of the production surface: process-global (atomic) settings read once per
 * nnlm_create / allocation, never to be changed while another thread creates handles.
 * Test hook: handles created from now on plan their launches as if the device had `cus` compute units (0 = the device's own
 * count) -- small problems then take the launch forms large ones take on the real device (persistent SCD sweep). */
int nnlm_debug_set_cus(int cus);
/* Test hook: the matrix-sized workspaces of the KL solvers (starting states of all columns, transposed copy of A, streaming scratch)
 * "do not fit" when they exceed `bytes` (0 = no limit): the half-step then takes its smaller-footprint path -- the streaming kernel
 * over column chunks -- exactly as it does when hipMalloc itself says no. */
int nnlm_debug_alloc_limit(size_t bytes);
/* Facts about the handle's last launches, for bench.py's kernel attribution: key = "cus" (compute units the launch policy
 * counts), "sweep_form_w" / "sweep_form_h" (SCD sweep of the last W / H half-step: 0 plain sweep_scd_q_kernel, 1 persistent
 * sweep_scd_qw_kernel -- both strict fp64 --, 2 sweep_scd_f_kernel, 3 sweep_row_kernel (fp32-operand mode: 3 while the launch is one
 * round of four-column wavefronts, at most 32 columns per CU), -1 none yet), "sweep_groups_w" /
 * "sweep_groups_h" (column groups -- form 2: wavefronts -- per workgroup of that launch), "kl_form_w" / "kl_form_h" (KL solver of the
 * last W / H half-step: 0 kl_tile_kernel on the starting states of the wh_store GEMM, 1 kl_tile_kernel forming its own starting states
 * -- no room for the matrix-sized buffer --, 2 kl_reg64_kernel (strict), 3 kl_stream_kernel over column chunks, -1 none yet). */
int nnlm_get_info(nnlm_handle *h, const char *key, double *value);

#ifdef __cplusplus
}
#endif
#endif /* NNLM_MI355X_H */
