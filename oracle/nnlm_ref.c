/*
 * TEST INFRASTRUCTURE ONLY -- plain-C (C99 + OpenMP) fp64 restatement of the hot path of
 * linxihui/NNLM v0.4.4: c_nnmf(), c_nnlm(), update(), update_with_missing() and the four
 * per-column solvers.  It is the fast CPU oracle behind tests/, __graft_entry__.smoke() and
 * the `cpu_baseline` leg of bench.py.  The product (nnlm_amd/, libnnlm_mi355x.so) never
 * links, loads or calls it.
 *
 * Parity pin: reproduces the reference's known-answer vectors
 * (tests/testthat/test-nnlm.R:6-15,19-26,29-43 -> tests/golden/nnlm_kat.json) and agrees with
 * the independent numpy restatement oracle/nnlm_oracle.py (tests/test_oracle.py).  The
 * reference itself needs R, Rcpp, RcppArmadillo, RcppProgress and R's BLAS and is unbuildable in
 * this image; Armadillo/BLAS summation order is therefore unpinned (value-level parity only).
 *
 * The *cost structure* of the reference is kept on purpose so that timing this file stands in
 * for "the reference's OpenMP path": A.t() is materialised every outer iteration
 * (src/nnmf.cpp:131), every column does its own gemv Wt*A[:,j] inside an
 * `omp parallel for schedule(dynamic)` (src/update_with_missing.cpp:29-53), and the error
 * block forms the full n x m product (src/nnmf.cpp:135-140).  Armadillo->BLAS calls are
 * replaced by hand-written loops (no BLAS in the image).
 *
 * Storage: column-major everywhere, as in Armadillo.  Wt is k x n, H is k x m, A is n x m.
 * Citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TINY_NUM 1e-16 /* src/nnlm.h:17 */

/* ---------------------------------------------------------------------------------------
 * per-column solvers, src/base_algorithms.cpp
 * mask: pointer to the k mask words of this column, or NULL when the mask matrix is empty
 * ------------------------------------------------------------------------------------- */

/* src/base_algorithms.cpp:3-37 */
int ref_scd_ls_update(double *Hj, const double *WtW, double *mu, const int *mask, int k,
                      unsigned max_iter, double rel_tol)
{
    double rel_err = 1 + rel_tol;
    unsigned t = 0;
    for (; t < max_iter && rel_err > rel_tol; t++) {
        rel_err = 0;
        for (int q = 0; q < k; q++) {
            if (mask && mask[q] != 0) continue; /* `mask(k) > 0` on the reference's UNSIGNED matrix: any non-zero (R's NA_LOGICAL = INT_MIN converts to a huge uword) */
            double tmp = Hj[q] - mu[q] / WtW[q + (size_t)q * k];
            if (tmp < 0) tmp = 0;
            if (tmp != Hj[q]) {
                const double d = tmp - Hj[q];
                const double *col = WtW + (size_t)q * k;
                for (int r = 0; r < k; r++) mu[r] += d * col[r];
            } else
                continue;
            double etmp = 2 * fabs(Hj[q] - tmp) / (tmp + Hj[q] + TINY_NUM);
            if (etmp > rel_err) rel_err = etmp;
            Hj[q] = tmp;
        }
    }
    return (int)t;
}

/* src/base_algorithms.cpp:40-68 */
int ref_lee_ls_update(double *Hj, const double *WtW, const double *WtAj, double beta3,
                      const int *mask, int k, unsigned max_iter, double rel_tol)
{
    double rel_err = rel_tol + 1;
    unsigned t = 0;
    for (; t < max_iter && rel_err > rel_tol; t++) {
        rel_err = 0;
        for (int q = 0; q < k; q++) {
            if (mask && mask[q] != 0) continue; /* `mask(k) > 0` on the reference's UNSIGNED matrix: any non-zero (R's NA_LOGICAL = INT_MIN converts to a huge uword) */
            const double *col = WtW + (size_t)q * k;
            double tmp = 0;
            for (int r = 0; r < k; r++) tmp += col[r] * Hj[r];
            tmp += beta3;
            tmp = WtAj[q] / (tmp + TINY_NUM);
            Hj[q] *= tmp;
            tmp = 2 * fabs(tmp - 1) / (tmp + 1);
            if (tmp > rel_err) rel_err = tmp;
        }
    }
    return (int)t;
}

/* src/base_algorithms.cpp:71-116.  Wt is k x p with leading dimension ldw (row q = Wt[q + i*ldw]);
 * Aj has p entries; scratch holds p doubles (Ajt). */
int ref_scd_kl_update(double *Hj, const double *Wt, int ldw, const double *Aj, int p,
                      const double *sumW, const int *mask, const double *beta, int k,
                      unsigned max_iter, double rel_tol, double *scratch)
{
    double sumHj = 0;
    for (int q = 0; q < k; q++) sumHj += Hj[q];
    double *Ajt = scratch;
    for (int i = 0; i < p; i++) {
        double s = 0;
        for (int q = 0; q < k; q++) s += Wt[q + (size_t)i * ldw] * Hj[q];
        Ajt[i] = s;
    }
    double rel_err = 1 + rel_tol;
    unsigned t = 0;
    for (; t < max_iter && rel_err > rel_tol; t++) {
        rel_err = 0;
        for (int q = 0; q < k; q++) {
            if (mask && mask[q] != 0) continue; /* `mask(k) > 0` on the reference's UNSIGNED matrix: any non-zero (R's NA_LOGICAL = INT_MIN converts to a huge uword) */
            double a = 0, b = 0;
            for (int i = 0; i < p; i++) {
                double mu = Wt[q + (size_t)i * ldw] / (Ajt[i] + TINY_NUM);
                a += Aj[i] * (mu * mu);
                b += Aj[i] * mu;
            }
            b -= sumW[q];
            a += beta[0];
            b += a * Hj[q] - beta[2] - beta[1] * (sumHj - Hj[q]);
            double tmp = b / (a + TINY_NUM);
            if (tmp < 0) tmp = 0;
            if (tmp != Hj[q]) {
                const double d = tmp - Hj[q];
                for (int i = 0; i < p; i++) Ajt[i] += d * Wt[q + (size_t)i * ldw];
                double etmp = 2 * fabs(Hj[q] - tmp) / (tmp + Hj[q] + TINY_NUM);
                if (etmp > rel_err) rel_err = etmp;
                sumHj += tmp - Hj[q];
                Hj[q] = tmp;
            }
        }
    }
    return (int)t;
}

/* src/base_algorithms.cpp:119-151 */
int ref_lee_kl_update(double *Hj, const double *Wt, int ldw, const double *Aj, int p,
                      const double *sumW, const int *mask, const double *beta, int k,
                      unsigned max_iter, double rel_tol, double *scratch)
{
    double sumHj = 0;
    for (int q = 0; q < k; q++) sumHj += Hj[q];
    double *wh = scratch;
    for (int i = 0; i < p; i++) {
        double s = 0;
        for (int q = 0; q < k; q++) s += Wt[q + (size_t)i * ldw] * Hj[q];
        wh[i] = s;
    }
    double rel_err = rel_tol + 1;
    unsigned t = 0;
    for (; t < max_iter && rel_err > rel_tol; t++) {
        rel_err = 0;
        for (int q = 0; q < k; q++) {
            if (mask && mask[q] != 0) continue; /* `mask(k) > 0` on the reference's UNSIGNED matrix: any non-zero (R's NA_LOGICAL = INT_MIN converts to a huge uword) */
            double tmp = 0;
            for (int i = 0; i < p; i++) tmp += Wt[q + (size_t)i * ldw] * (Aj[i] / (wh[i] + TINY_NUM));
            tmp /= (sumW[q] + beta[0] * Hj[q] + beta[1] * (sumHj - Hj[q]) + beta[2]);
            const double c = (tmp - 1) * Hj[q];
            for (int i = 0; i < p; i++) wh[i] += c * Wt[q + (size_t)i * ldw];
            sumHj += (tmp - 1) * Hj[q];
            Hj[q] *= tmp;
            tmp = 2 * fabs(tmp - 1) / (tmp + 1);
            if (tmp > rel_err) rel_err = tmp;
        }
    }
    return (int)t;
}

/* ---------------------------------------------------------------------------------------
 * half-steps, src/update_with_missing.cpp
 * ------------------------------------------------------------------------------------- */
static void gram_full(const double *Wt, int k, int n, double *G)
{ /* G = Wt * Wt^T, src/update_with_missing.cpp:19 */
    memset(G, 0, sizeof(double) * (size_t)k * k);
    for (int i = 0; i < n; i++) {
        const double *w = Wt + (size_t)i * k;
        for (int r = 0; r < k; r++) {
            const double wr = w[r];
            double *g = G + (size_t)r * k;
            for (int q = 0; q < k; q++) g[q] += w[q] * wr;
        }
    }
}

static void gram_edits(double *G, int k, const double *beta)
{ /* src/update_with_missing.cpp:20-24 and :98-103 */
    if (beta[0] != beta[1])
        for (int q = 0; q < k; q++) G[q + (size_t)q * k] += beta[0] - beta[1];
    if (beta[1] != 0)
        for (size_t e = 0; e < (size_t)k * k; e++) G[e] += beta[1];
    for (int q = 0; q < k; q++) G[q + (size_t)q * k] += TINY_NUM;
}

static int all_masked(const int *mcol, int k)
{ /* arma::all(mask.col(j)), src/update_with_missing.cpp:33 */
    for (int q = 0; q < k; q++)
        if (mcol[q] == 0) return 0;
    return 1;
}

/* Dense half-step, src/update_with_missing.cpp:3-55.
 * H k x m (in/out), Wt k x n, A n x m, mask k x m or NULL, beta[3]. Returns total sweeps. */
int ref_update(double *H, const double *Wt, const double *A, const int *mask, const double *beta,
               int k, int n, int m, unsigned max_iter, double rel_tol, int n_threads, int method)
{
    int total_raw_iter = 0;
    if (n_threads < 0) n_threads = 0;
    double *WtW = NULL, *sumW = NULL;
    if (method == 1 || method == 2) {
        WtW = (double *)malloc(sizeof(double) * (size_t)k * k);
        gram_full(Wt, k, n, WtW);
        gram_edits(WtW, k, beta);
    } else {
        sumW = (double *)calloc((size_t)k, sizeof(double));
        for (int i = 0; i < n; i++)
            for (int q = 0; q < k; q++) sumW[q] += Wt[q + (size_t)i * k];
    }
#ifdef _OPENMP
    int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#pragma omp parallel num_threads(nt)
#endif
    {
        double *mu = (double *)malloc(sizeof(double) * (size_t)k);
        double *scratch = (method >= 3) ? (double *)malloc(sizeof(double) * (size_t)n) : NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
        for (int j = 0; j < m; j++) {
            const int *mj = mask ? mask + (size_t)j * k : NULL;
            if (mj && all_masked(mj, k)) continue;
            double *Hj = H + (size_t)j * k;
            const double *Aj = A + (size_t)j * n;
            int iter = 0;
            if (method == 1 || method == 2) {
                /* WtAj = Wt * A[:,j]  (per-column gemv, src/update_with_missing.cpp:39,45) */
                for (int q = 0; q < k; q++) mu[q] = 0;
                for (int i = 0; i < n; i++) {
                    const double a = Aj[i];
                    const double *w = Wt + (size_t)i * k;
                    for (int q = 0; q < k; q++) mu[q] += w[q] * a;
                }
                if (method == 1) {
                    for (int q = 0; q < k; q++) {
                        double s = 0;
                        for (int r = 0; r < k; r++) s += WtW[q + (size_t)r * k] * Hj[r];
                        mu[q] = s - mu[q];
                    }
                    if (beta[2] != 0)
                        for (int q = 0; q < k; q++) mu[q] += beta[2];
                    iter = ref_scd_ls_update(Hj, WtW, mu, mj, k, max_iter, rel_tol);
                } else
                    iter = ref_lee_ls_update(Hj, WtW, mu, beta[2], mj, k, max_iter, rel_tol);
            } else if (method == 3)
                iter = ref_scd_kl_update(Hj, Wt, k, Aj, n, sumW, mj, beta, k, max_iter, rel_tol, scratch);
            else if (method == 4)
                iter = ref_lee_kl_update(Hj, Wt, k, Aj, n, sumW, mj, beta, k, max_iter, rel_tol, scratch);
#ifdef _OPENMP
#pragma omp critical
#endif
            total_raw_iter += iter;
        }
        free(mu);
        free(scratch);
    }
    free(WtW);
    free(sumW);
    return total_raw_iter;
}

/* Half-step with NA in A, src/update_with_missing.cpp:58-139. */
int ref_update_with_missing(double *H, const double *Wt, const double *A, const int *mask,
                            const double *beta, int k, int n, int m, unsigned max_iter,
                            double rel_tol, int n_threads, int method)
{
    unsigned total_raw_iter = 0;
    if (n_threads < 0) n_threads = 0;
#ifdef _OPENMP
    int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#pragma omp parallel num_threads(nt)
#endif
    {
        double *WtW = (double *)malloc(sizeof(double) * (size_t)k * k);
        double *mu = (double *)malloc(sizeof(double) * (size_t)k);
        double *sumW = (double *)malloc(sizeof(double) * (size_t)k);
        double *Wsub = (double *)malloc(sizeof(double) * (size_t)k * n); /* Wt.cols(non_missing) */
        double *Asub = (double *)malloc(sizeof(double) * (size_t)n);     /* A.elem(j*n + non_missing) */
        double *scratch = (double *)malloc(sizeof(double) * (size_t)(n > k ? n : k)); /* (holds k gradients below: k may exceed n) */
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
        for (int j = 0; j < m; j++) {
            const int *mj = mask ? mask + (size_t)j * k : NULL;
            if (mj && all_masked(mj, k)) continue;
            double *Hj = H + (size_t)j * k;
            const double *Aj = A + (size_t)j * n;
            /* non_missing = find_finite(A.col(j)), :80-83 */
            int p = 0;
            for (int i = 0; i < n; i++)
                if (isfinite(Aj[i])) {
                    memcpy(Wsub + (size_t)p * k, Wt + (size_t)i * k, sizeof(double) * (size_t)k);
                    Asub[p] = Aj[i];
                    p++;
                }
            /* (when nothing is missing p == n and Wsub/Asub equal Wt/A[:,j]) */
            if (method == 1 || method == 2) {
                gram_full(Wsub, k, p, WtW);
                for (int q = 0; q < k; q++) mu[q] = 0;
                for (int i = 0; i < p; i++) {
                    const double a = Asub[i];
                    const double *w = Wsub + (size_t)i * k;
                    for (int q = 0; q < k; q++) mu[q] += w[q] * a;
                }
                gram_edits(WtW, k, beta);
            }
            int iter = 0;
            if (method == 1) {
                for (int q = 0; q < k; q++) { /* mu = WtW*H.col(j) - mu, :109 */
                    double s = 0;
                    for (int r = 0; r < k; r++) s += WtW[q + (size_t)r * k] * Hj[r];
                    scratch[q] = s - mu[q];
                }
                for (int q = 0; q < k; q++) mu[q] = scratch[q];
                if (beta[2] != 0)
                    for (int q = 0; q < k; q++) mu[q] += beta[2];
                iter = ref_scd_ls_update(Hj, WtW, mu, mj, k, max_iter, rel_tol);
            } else if (method == 2)
                iter = ref_lee_ls_update(Hj, WtW, mu, beta[2], mj, k, max_iter, rel_tol);
            else if (method == 3 || method == 4) {
                for (int q = 0; q < k; q++) sumW[q] = 0;
                for (int i = 0; i < p; i++)
                    for (int q = 0; q < k; q++) sumW[q] += Wsub[q + (size_t)i * k];
                if (method == 3)
                    iter = ref_scd_kl_update(Hj, Wsub, k, Asub, p, sumW, mj, beta, k, max_iter, rel_tol, scratch);
                else
                    iter = ref_lee_kl_update(Hj, Wsub, k, Asub, p, sumW, mj, beta, k, max_iter, rel_tol, scratch);
            }
#ifdef _OPENMP
#pragma omp critical
#endif
            total_raw_iter += (unsigned)iter;
        }
        free(WtW); free(mu); free(sumW); free(Wsub); free(Asub); free(scratch);
    }
    return (int)total_raw_iter;
}

/* ---------------------------------------------------------------------------------------
 * drivers
 * ------------------------------------------------------------------------------------- */
static void transpose(const double *X, int r, int c, double *Xt, int n_threads)
{ /* Xt (c x r) = X (r x c)^T, blocked; stands in for arma's A.t() */
    const int B = 32;
    (void)n_threads;
#ifdef _OPENMP
    int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nt)
#endif
    for (int jb = 0; jb < c; jb += B)
        for (int ib = 0; ib < r; ib += B) {
            int je = jb + B < c ? jb + B : c, ie = ib + B < r ? ib + B : r;
            for (int j = jb; j < je; j++)
                for (int i = ib; i < ie; i++) Xt[j + (size_t)i * c] = X[i + (size_t)j * r];
        }
}

/* src/nnmf.cpp:224-240; Wt is k x n, H is k x m */
static double add_penalty(double terr, const double *Wt, const double *H, int k, int n, int m,
                          double N, const double *alpha, const double *beta)
{
    const double *X[2] = {Wt, H};
    const double *r[2] = {alpha, beta};
    const int len[2] = {n, m};
    /* the reference adds the six terms in the order W-L2, H-L2, W-angle, H-angle, W-L1, H-L1 */
    for (int s = 0; s < 2; s++)
        if (r[s][0] != r[s][1]) {
            double a = 0;
            for (size_t e = 0; e < (size_t)k * len[s]; e++) a += X[s][e] * X[s][e];
            terr += 0.5 * (r[s][0] - r[s][1]) * a / N;
        }
    for (int s = 0; s < 2; s++)
        if (r[s][1] != 0) { /* accu(X * X.t()) = sum_i (sum_q X[q,i])^2 */
            double a = 0;
            for (int i = 0; i < len[s]; i++) {
                double c = 0;
                for (int q = 0; q < k; q++) c += X[s][q + (size_t)i * k];
                a += c * c;
            }
            terr += 0.5 * r[s][1] * a / N;
        }
    for (int s = 0; s < 2; s++)
        if (r[s][2] != 0) {
            double a = 0;
            for (size_t e = 0; e < (size_t)k * len[s]; e++) a += X[s][e];
            terr += r[s][2] * a / N;
        }
    return terr;
}

/* MSE and the variable part of MKL over finite entries of A, src/nnmf.cpp:121-126,135-140 */
static void errors(const double *A, const double *Wt, const double *H, int k, int n, int m,
                   int n_threads, double N, double *mse, double *kl)
{
    double s2 = 0, sk = 0;
    (void)n_threads;
#ifdef _OPENMP
    int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#pragma omp parallel for schedule(static) reduction(+ : s2, sk) num_threads(nt)
#endif
    for (int j = 0; j < m; j++) {
        const double *h = H + (size_t)j * k;
        double c2 = 0, ck = 0;
        for (int i = 0; i < n; i++) {
            const double a = A[i + (size_t)j * n];
            if (!isfinite(a)) continue;
            const double *w = Wt + (size_t)i * k;
            double ah = 0;
            for (int q = 0; q < k; q++) ah += w[q] * h[q];
            const double d = a - ah;
            c2 += d * d;
            ck += -(a + TINY_NUM) * log(ah + TINY_NUM) + ah;
        }
        s2 += c2;
        sk += ck;
    }
    *mse = s2 / N;
    *kl = sk / N;
}

/*
 * c_nnmf, src/nnmf.cpp:4-220 (17-argument .Call signature, src/RcppExports.cpp:29-51).
 *   A n x m (may hold NaN/Inf = missing); W n x k in/out (W_given=0: default init from unif());
 *   H k x m in/out; Wm n x k / Hm k x m logical (int) or NULL; traces have capacity
 *   ceil(max_iter/trace)+1; *n_err receives their used length; *warn the warning flag.
 *   unif: stand-in for R's unif_rand() (NULL -> a fixed LCG; only the reference's scale is kept).
 */
int ref_c_nnmf(const double *A, int n, int m, unsigned k_, double *W, int W_given, double *H,
               int H_given, const int *Wm, const int *Hm, const double *alpha, const double *beta,
               unsigned max_iter, double rel_tol, int n_threads, int verbose, int show_warning,
               unsigned inner_max_iter, double inner_rel_tol, int method, unsigned trace,
               double *mse_err, double *mkl_err, double *terr, double *ave_epoch, int *n_err,
               unsigned *n_iteration, int *warn, double (*unif)(void))
{
    const int k = (int)k_;
    (void)verbose;
    if (trace < 1) trace = 1;
    unsigned err_len = (unsigned)ceil((double)max_iter / (double)trace) + 1;
    double N_non_missing = (double)n * (double)m;
    double rel_err = rel_tol + 1, terr_last = 1e99;

    /* any_missing / constant KL part, :65-73 */
    int any_missing = 0;
    {
        double cnt = 0, s = 0;
        for (size_t e = 0; e < (size_t)n * m; e++) {
            const double a = A[e];
            if (!isfinite(a)) { any_missing = 1; continue; }
            cnt += 1;
            s += (a + TINY_NUM) * log(a + TINY_NUM) - a;
        }
        if (any_missing) N_non_missing = cnt;
        for (unsigned e = 0; e < err_len; e++) mkl_err[e] = s / N_non_missing;
    }

    double *Wt = (double *)malloc(sizeof(double) * (size_t)k * n);
    double *At = (double *)malloc(sizeof(double) * (size_t)n * m);
    int *Wmt = NULL;
    if (Wm) { /* inplace_trans(Wm), :78 */
        Wmt = (int *)malloc(sizeof(int) * (size_t)k * n);
        for (int i = 0; i < n; i++)
            for (int q = 0; q < k; q++) Wmt[q + (size_t)i * k] = Wm[i + (size_t)q * n];
    }
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
#define DRAW() (unif ? unif() : ((lcg = lcg * 6364136223846793005ull + 1442695040888963407ull), (double)(lcg >> 11) * (1.0 / 9007199254740992.0)))
    if (!W_given) { /* :82-88 */
        for (size_t e = 0; e < (size_t)k * n; e++) Wt[e] = DRAW() * 0.01;
        if (Wmt)
            for (size_t e = 0; e < (size_t)k * n; e++)
                if (Wmt[e] != 0) Wt[e] = 0.0; /* find(Wm > 0) on arma::umat */
    } else
        transpose(W, n, k, Wt, n_threads);
    if (!H_given) { /* :92-98 */
        for (size_t e = 0; e < (size_t)k * m; e++) H[e] = DRAW() * 0.01;
        if (Hm)
            for (size_t e = 0; e < (size_t)k * m; e++)
                if (Hm[e] != 0) H[e] = 0.0;
    }
#undef DRAW

    int total_raw_iter = 0;
    unsigned i = 0, i_e = 0;
    for (; i < max_iter && fabs(rel_err) > rel_tol; i++) {
        transpose(A, n, m, At, n_threads); /* A.t() materialised each iteration, :117/:131 */
        if (any_missing) {
            total_raw_iter += ref_update_with_missing(Wt, H, At, Wmt, alpha, k, m, n, inner_max_iter, inner_rel_tol, n_threads, method);
            total_raw_iter += ref_update_with_missing(H, Wt, A, Hm, beta, k, n, m, inner_max_iter, inner_rel_tol, n_threads, method);
        } else {
            total_raw_iter += ref_update(Wt, H, At, Wmt, alpha, k, m, n, inner_max_iter, inner_rel_tol, n_threads, method);
            total_raw_iter += ref_update(H, Wt, A, Hm, beta, k, n, m, inner_max_iter, inner_rel_tol, n_threads, method);
        }
        if (i % trace == 0) {
            double mse, kl;
            errors(A, Wt, H, k, n, m, n_threads, N_non_missing, &mse, &kl);
            mse_err[i_e] = mse;
            mkl_err[i_e] += kl;
            ave_epoch[i_e] = (double)total_raw_iter / (n + m);
            terr[i_e] = (method < 3) ? 0.5 * mse_err[i_e] : mkl_err[i_e];
            terr[i_e] = add_penalty(terr[i_e], Wt, H, k, n, m, N_non_missing, alpha, beta);
            rel_err = 2 * (terr_last - terr[i_e]) / (terr_last + terr[i_e] + TINY_NUM);
            terr_last = terr[i_e];
            total_raw_iter = 0;
            ++i_e;
        }
    }
    if ((unsigned)(i - 1) % trace != 0) { /* :164 (unsigned arithmetic) */
        double mse, kl;
        errors(A, Wt, H, k, n, m, n_threads, N_non_missing, &mse, &kl);
        mse_err[i_e] = mse;
        mkl_err[i_e] += kl;
        ave_epoch[i_e] = (double)total_raw_iter / (n + m);
        terr[i_e] = (method < 3) ? 0.5 * mse_err[i_e] : mkl_err[i_e];
        terr[i_e] = add_penalty(terr[i_e], Wt, H, k, n, m, N_non_missing, alpha, beta);
        rel_err = 2 * (terr_last - terr[i_e]) / (terr_last + terr[i_e] + TINY_NUM);
        terr_last = terr[i_e];
        ++i_e;
    }
    *n_err = (int)i_e;
    *n_iteration = i;
    *warn = (show_warning && rel_err > rel_tol) ? 1 : 0; /* :208 */
    transpose(Wt, k, n, W, n_threads);                    /* W = W.t(), :212 */
    free(Wt); free(At); free(Wmt);
    return 0;
}

/* c_nnlm, src/nnlm.cpp:4-53.  x n x p, y n x q, beta p x q (in: beta0 when beta_given). */
int ref_c_nnlm(const double *x, const double *y, int n, int p, int q, const double *alpha,
               const int *mask, double *beta, int beta_given, unsigned max_iter, double rel_tol,
               int n_threads, int method, double (*unif)(void))
{
    int any_missing = 0;
    for (size_t e = 0; e < (size_t)n * q; e++)
        if (!isfinite(y[e])) { any_missing = 1; break; }
    if (!beta_given) {
        uint64_t lcg = 0x9E3779B97F4A7C15ull;
        for (size_t e = 0; e < (size_t)p * q; e++) {
            if (unif) beta[e] = unif();
            else { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; beta[e] = (double)(lcg >> 11) * (1.0 / 9007199254740992.0); }
        }
    }
    double *xt = (double *)malloc(sizeof(double) * (size_t)n * p);
    transpose(x, n, p, xt, n_threads);
    int nstep = any_missing ? ref_update_with_missing(beta, xt, y, mask, alpha, p, n, q, max_iter, rel_tol, n_threads, method)
                            : ref_update(beta, xt, y, mask, alpha, p, n, q, max_iter, rel_tol, n_threads, method);
    free(xt);
    return nstep;
}
