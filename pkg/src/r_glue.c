/*
 * r_glue.c -- the R side of the drop-in boundary: Rinternals-only `.Call` entry points with the EXACT names,
 * arities and named-list results of the reference's generated glue (reference src/RcppExports.cpp:10-27, :29-54,
 * registration :56-65), forwarding to the C ABI of libnnlm_mi355x.so (include/nnlm_mi355x.h).
 *
 * With this file compiled into the package's shared object (and libnnlm_mi355x.so linked), the reference's R layer
 * (R/nnmf.R, R/nnlm.R, R/nnmf_methods.R, R/misc.R, R/RcppExports.R) runs UNCHANGED on the MI355X:
 *     R/RcppExports.R:4-10   .Call(`_NNLM_c_nnlm`, ...9 args)  /  .Call(`_NNLM_c_nnmf`, ...17 args)
 *     NAMESPACE:11           useDynLib(NNLM, .registration = TRUE)
 * No Rcpp, RcppArmadillo or RcppProgress is needed any more.
 *
 * This image has no R (no R.h / Rinternals.h), so the file is compiled only where R exists:
 *     R CMD SHLIB -o NNLM.so r_glue.c -I<repo>/include -L<repo>/nnlm_amd -lnnlm_mi355x
 * (see INTEGRATION.md).  It is deliberately plain C against the stable R API.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <R_ext/Utils.h>
#include <R_ext/Random.h>

#include "nnlm_mi355x.h"

/* Re-raises the user interrupt the library reported (NNLM_ERR_INTERRUPT) -- after the device handle has been released, so the jump
 * crosses no C++ frame.  The poll inside the library runs R_CheckUserInterrupt() under R_ToplevelExec(), which CONSUMES the pending
 * interrupt; R's own way of raising it again, Rf_onintr() (what Rcpp's END_RCPP calls), is not part of the API `R CMD check` accepts.
 * This does what onintr() does through API entry points only: a condition of class c("interrupt", "condition") is signalled to the
 * handler stack (tryCatch(interrupt = ) unwinds to its handler, withCallingHandlers() sees it) and, if nothing took it, control
 * returns to top level through the "abort" restart. */
static void raise_interrupt(void)
{
    SEXP cls = PROTECT(Rf_allocVector(STRSXP, 2));
    SEXP cond = PROTECT(Rf_allocVector(VECSXP, 0));
    SEXP call, arg;
    SET_STRING_ELT(cls, 0, Rf_mkChar("interrupt"));
    SET_STRING_ELT(cls, 1, Rf_mkChar("condition"));
    Rf_setAttrib(cond, R_ClassSymbol, cls);
    call = PROTECT(Rf_lang2(Rf_install("signalCondition"), cond));
    Rf_eval(call, R_BaseEnv);
    arg = PROTECT(Rf_mkString("abort"));
    call = PROTECT(Rf_lang2(Rf_install("invokeRestart"), arg));
    Rf_eval(call, R_BaseEnv); /* does not return where an "abort" restart is on the stack (every R front-end's top level) */
    UNPROTECT(5);
    /* an embedded front-end without that restart: never hand a NULL result back for an interrupted call */
    Rf_error("nnmf: interrupted by the user");
}

/* ---- callbacks = the R API points the reference touches (include/nnlm_mi355x.h nnlm_callbacks) ---------------- */
static void chk_intr_body(void *dummy) { (void)dummy; R_CheckUserInterrupt(); }
static int cb_check_interrupt(void *ctx)
{ /* Rcpp::checkUserInterrupt(), reference src/nnmf.cpp:111: report instead of long-jumping out of device code */
    (void)ctx;
    return R_ToplevelExec(chk_intr_body, NULL) == FALSE;
}
static void cb_progress(void *ctx, unsigned done, unsigned total)
{ /* RcppProgress bar (verbose == 1), reference src/nnmf.cpp:60,112 */
    (void)ctx;
    if (done == 1) Rprintf("0%%   10   20   30   40   50   60   70   80   90   100%%\n[----|----|----|----|----|----|----|----|----|----|\n");
    {
        unsigned before = (unsigned)(50.0 * (done - 1) / total), now = (unsigned)(50.0 * done / total);
        for (; before < now; before++) Rprintf("*");
        if (done == total) Rprintf("|\n");
    }
}
static void cb_print(void *ctx, const char *text) { (void)ctx; Rprintf("%s", text); }                 /* Rprintf */
static double cb_unif(void *ctx) { (void)ctx; return unif_rand(); }                                    /* arma::randu -> R RNG */

/* The warning callback stays NULL: Rf_warning() may long-jump (options(warn = 2), calling handlers) and must not do so
 * through the library's C++ frames while the device handle is alive.  nnlm_c_nnmf() reports `warned`; the warning of
 * reference src/nnmf.cpp:208-209 is raised below, after the library has returned and released the device. */
static const nnlm_callbacks k_callbacks = {NULL, cb_check_interrupt, cb_progress, cb_print, NULL, cb_unif};

static SEXP named_list(int n, const char **names)
{
    SEXP out = PROTECT(Rf_allocVector(VECSXP, n)), nm = PROTECT(Rf_allocVector(STRSXP, n));
    for (int i = 0; i < n; i++) SET_STRING_ELT(nm, i, Rf_mkChar(names[i]));
    Rf_setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(2);
    return out;
}

static const int *lgl_or_null(SEXP m) { return (XLENGTH(m) == 0) ? NULL : LOGICAL(m); }
static const double *dbl_or_null(SEXP m) { return (XLENGTH(m) == 0) ? NULL : REAL(m); }

/* c_nnmf: reference src/RcppExports.cpp:29-54 (17 arguments), result list of src/nnmf.cpp:211-219 */
SEXP _NNLM_c_nnmf(SEXP ASEXP, SEXP kSEXP, SEXP WSEXP, SEXP HSEXP, SEXP WmSEXP, SEXP HmSEXP, SEXP alphaSEXP, SEXP betaSEXP,
                  SEXP max_iterSEXP, SEXP rel_tolSEXP, SEXP n_threadsSEXP, SEXP verboseSEXP, SEXP show_warningSEXP,
                  SEXP inner_max_iterSEXP, SEXP inner_rel_tolSEXP, SEXP methodSEXP, SEXP traceSEXP)
{
    const int n = Rf_nrows(ASEXP), m = Rf_ncols(ASEXP);
    const unsigned k = (unsigned)Rf_asInteger(kSEXP);
    const unsigned max_iter = (unsigned)Rf_asInteger(max_iterSEXP);
    unsigned trace = (unsigned)Rf_asInteger(traceSEXP);
    const unsigned cap = nnlm_trace_capacity(max_iter, trace);
    const char *names[] = {"W", "H", "mse_error", "mkl_error", "target_error", "average_epoch", "n_iteration"};
    SEXP out = PROTECT(named_list(7, names));
    SEXP W = PROTECT(Rf_allocMatrix(REALSXP, n, (int)k)), H = PROTECT(Rf_allocMatrix(REALSXP, (int)k, m));
    SEXP mse = PROTECT(Rf_allocVector(REALSXP, cap)), mkl = PROTECT(Rf_allocVector(REALSXP, cap));
    SEXP terr = PROTECT(Rf_allocVector(REALSXP, cap)), ep = PROTECT(Rf_allocVector(REALSXP, cap));
    int n_trace = 0, warned = 0;
    unsigned n_iteration = 0;
    int rc;

    GetRNGstate(); /* Rcpp::RNGScope, reference src/RcppExports.cpp:33 */
    rc = nnlm_c_nnmf(REAL(ASEXP), n, m, k, dbl_or_null(WSEXP), dbl_or_null(HSEXP), lgl_or_null(WmSEXP), lgl_or_null(HmSEXP),
                     REAL(alphaSEXP), REAL(betaSEXP), max_iter, Rf_asReal(rel_tolSEXP), Rf_asInteger(n_threadsSEXP),
                     Rf_asInteger(verboseSEXP), Rf_asLogical(show_warningSEXP), (unsigned)Rf_asInteger(inner_max_iterSEXP),
                     Rf_asReal(inner_rel_tolSEXP), Rf_asInteger(methodSEXP), trace, REAL(W), REAL(H), REAL(mse), REAL(mkl),
                     REAL(terr), REAL(ep), &n_trace, &n_iteration, &warned, &k_callbacks);
    PutRNGstate();
    if (rc == NNLM_ERR_INTERRUPT) { UNPROTECT(7); raise_interrupt(); return R_NilValue; }
    if (rc != NNLM_OK) { UNPROTECT(7); Rf_error("%s", nnlm_last_error(NULL)); } /* BEGIN_RCPP/END_RCPP: C++ exception -> R error */

    SET_VECTOR_ELT(out, 0, W);
    SET_VECTOR_ELT(out, 1, H);
    SET_VECTOR_ELT(out, 2, Rf_xlengthgets(mse, n_trace));  /* traces truncated to i_e, reference src/nnmf.cpp:200-206 */
    SET_VECTOR_ELT(out, 3, Rf_xlengthgets(mkl, n_trace));
    SET_VECTOR_ELT(out, 4, Rf_xlengthgets(terr, n_trace));
    SET_VECTOR_ELT(out, 5, Rf_xlengthgets(ep, n_trace));
    SET_VECTOR_ELT(out, 6, Rf_ScalarInteger((int)n_iteration));
    if (warned) Rf_warning("Target tolerance not reached. Try a larger max.iter."); /* Rcpp::warning, src/nnmf.cpp:208-209 */
    UNPROTECT(7);
    return out;
}

/* c_nnlm: reference src/RcppExports.cpp:10-27 (9 arguments), result list of src/nnlm.cpp:49-52 */
SEXP _NNLM_c_nnlm(SEXP xSEXP, SEXP ySEXP, SEXP alphaSEXP, SEXP maskSEXP, SEXP beta0SEXP, SEXP max_iterSEXP, SEXP rel_tolSEXP,
                  SEXP n_threadsSEXP, SEXP methodSEXP)
{
    const int n = Rf_nrows(xSEXP), p = Rf_ncols(xSEXP), q = Rf_ncols(ySEXP);
    const char *names[] = {"coefficient", "n_iteration"};
    SEXP out = PROTECT(named_list(2, names));
    SEXP coef = PROTECT(Rf_allocMatrix(REALSXP, p, q));
    int nit = 0, rc;
    if (Rf_nrows(ySEXP) != n) { UNPROTECT(2); Rf_error("Dimensions of x and y do not match."); }
    GetRNGstate();
    rc = nnlm_c_nnlm(REAL(xSEXP), REAL(ySEXP), n, p, q, REAL(alphaSEXP), lgl_or_null(maskSEXP), dbl_or_null(beta0SEXP),
                     (unsigned)Rf_asInteger(max_iterSEXP), Rf_asReal(rel_tolSEXP), Rf_asInteger(n_threadsSEXP),
                     Rf_asInteger(methodSEXP), REAL(coef), &nit, &k_callbacks);
    PutRNGstate();
    if (rc != NNLM_OK) { UNPROTECT(2); Rf_error("%s", nnlm_last_error(NULL)); }
    SET_VECTOR_ELT(out, 0, coef);
    SET_VECTOR_ELT(out, 1, Rf_ScalarInteger(nit));
    UNPROTECT(2);
    return out;
}

/* registration, reference src/RcppExports.cpp:56-65 */
static const R_CallMethodDef CallEntries[] = {
    {"_NNLM_c_nnlm", (DL_FUNC)&_NNLM_c_nnlm, 9},
    {"_NNLM_c_nnmf", (DL_FUNC)&_NNLM_c_nnmf, 17},
    {NULL, NULL, 0}};

void R_init_NNLM(DllInfo *dll)
{
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

/* R calls this when the package's shared object is unloaded (library.dynam.unload / detach(unload = TRUE)): the device library keeps
 * streams, events and pinned bounce buffers between calls (nnlm_release_caches, include/nnlm_mi355x.h) -- they go with the package. */
void R_unload_NNLM(DllInfo *dll)
{
    (void)dll;
    (void)nnlm_release_caches();
}
