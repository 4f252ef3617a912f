/* TEST-ONLY mock of the R API surface declared in this directory: just enough of SEXP, protection, vectors, lists, errors
 * (longjmp, as R does), warnings, the RNG hooks and routine registration to call the .Call entry points of
 * pkg/src/r_glue.c from tests/test_gpu_boundary.py through ctypes.  Memory is never freed (a test process). */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "Rinternals.h"
#include "R.h"
#include "R_ext/Rdynload.h"
#include "R_ext/Random.h"
#include "R_ext/Utils.h"

struct SEXPREC {
    unsigned type;
    R_xlen_t len;
    int nrow, ncol;
    void *data;   /* double* / int* / SEXP* / char* */
    SEXP names;
    SEXP klass;   /* class attribute (STRSXP) or NULL */
};
static struct SEXPREC nil_rec = {0, 0, 0, 0, NULL, NULL, NULL}, names_rec = {0, 0, 0, 0, NULL, NULL, NULL}, class_rec = {0, 0, 0, 0, NULL, NULL, NULL},
                      base_rec = {0, 0, 0, 0, NULL, NULL, NULL};
SEXP R_NilValue = &nil_rec, R_NamesSymbol = &names_rec, R_ClassSymbol = &class_rec, R_BaseEnv = &base_rec;

static jmp_buf mock_jmp;
static int mock_jmp_armed = 0, mock_protect_depth = 0, mock_interrupt_after = -1, mock_interrupt_polls = 0, mock_rng_depth = 0;
static char mock_error[512], mock_warning_text[512], mock_output[8192];
static int mock_warnings = 0, mock_registered = 0, mock_rng_calls = 0, mock_onintr = 0;
static unsigned long long mock_rng_state = 12345;
static R_CallMethodDef mock_calls[8];

SEXP Rf_protect(SEXP s) { mock_protect_depth++; return s; }
void Rf_unprotect(int n) { mock_protect_depth -= n; }
SEXP Rf_allocVector(unsigned type, R_xlen_t n)
{
    SEXP s = (SEXP)calloc(1, sizeof *s);
    s->type = type; s->len = n; s->nrow = (int)n; s->ncol = 1; s->names = R_NilValue;
    const size_t es = (type == REALSXP) ? 8 : (type == VECSXP || type == STRSXP) ? sizeof(SEXP) : 4;
    s->data = calloc((size_t)(n > 0 ? n : 1), es);
    return s;
}
SEXP Rf_allocMatrix(unsigned type, int nrow, int ncol)
{
    SEXP s = Rf_allocVector(type, (R_xlen_t)nrow * ncol);
    s->nrow = nrow; s->ncol = ncol;
    return s;
}
SEXP Rf_mkChar(const char *c)
{
    SEXP s = (SEXP)calloc(1, sizeof *s);
    s->type = CHARSXP; s->len = (R_xlen_t)strlen(c); s->data = strdup(c); s->names = R_NilValue;
    return s;
}
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP *)x->data)[i] = v; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP *)x->data)[i] = v; return v; }
SEXP Rf_setAttrib(SEXP vec, SEXP name, SEXP val) { if (name == R_NamesSymbol) vec->names = val; else if (name == R_ClassSymbol) vec->klass = val; return val; }
SEXP Rf_mkString(const char *c) { SEXP s = Rf_allocVector(STRSXP, 1); SET_STRING_ELT(s, 0, Rf_mkChar(c)); return s; }
SEXP Rf_install(const char *c) { SEXP s = Rf_mkChar(c); s->type = SYMSXP; return s; }
SEXP Rf_lang2(SEXP fn, SEXP arg) { SEXP s = Rf_allocVector(VECSXP, 2); s->type = LANGSXP; ((SEXP *)s->data)[0] = fn; ((SEXP *)s->data)[1] = arg; return s; }
R_xlen_t XLENGTH(SEXP x) { return x->len; }
int *LOGICAL(SEXP x) { return (int *)x->data; }
double *REAL(SEXP x) { return (double *)x->data; }
int Rf_nrows(SEXP x) { return x->nrow; }
int Rf_ncols(SEXP x) { return x->ncol; }
int Rf_asInteger(SEXP x) { return x->type == REALSXP ? (int)((double *)x->data)[0] : ((int *)x->data)[0]; }
double Rf_asReal(SEXP x) { return x->type == REALSXP ? ((double *)x->data)[0] : (double)((int *)x->data)[0]; }
int Rf_asLogical(SEXP x) { return Rf_asInteger(x) != 0; }
SEXP Rf_xlengthgets(SEXP x, R_xlen_t n)
{
    SEXP s = Rf_allocVector(x->type, n);
    const size_t es = (x->type == REALSXP) ? 8 : 4;
    memcpy(s->data, x->data, (size_t)(n < x->len ? n : x->len) * es);
    return s;
}
SEXP Rf_ScalarInteger(int v) { SEXP s = Rf_allocVector(INTSXP, 1); ((int *)s->data)[0] = v; return s; }
void Rf_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(mock_error, sizeof mock_error, fmt, ap); va_end(ap);
    if (mock_jmp_armed) longjmp(mock_jmp, 1);
    fprintf(stderr, "mock R: error outside a guarded call: %s\n", mock_error); abort();
}
void Rf_warning(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(mock_warning_text, sizeof mock_warning_text, fmt, ap); va_end(ap);
    mock_warnings++;
}
/* The two calls r_glue.c evaluates to raise an interrupt: signalCondition(<condition of class "interrupt">) finds no handler in this
 * mock and returns; invokeRestart("abort") jumps to top level. */
static int mock_interrupt_signalled = 0;
SEXP Rf_eval(SEXP call, SEXP env)
{
    (void)env;
    if (call->type != LANGSXP) return call;
    {
        const char *fn = (const char *)((SEXP *)call->data)[0]->data;
        SEXP arg = ((SEXP *)call->data)[1];
        if (strcmp(fn, "signalCondition") == 0) {
            int is_intr = 0;
            if (arg->klass)
                for (R_xlen_t i = 0; i < arg->klass->len; i++)
                    if (strcmp((const char *)((SEXP *)arg->klass->data)[i]->data, "interrupt") == 0) is_intr = 1;
            if (is_intr) mock_interrupt_signalled = 1;
            return R_NilValue;
        }
        if (strcmp(fn, "invokeRestart") == 0 && strcmp((const char *)((SEXP *)arg->data)[0]->data, "abort") == 0) {
            if (mock_interrupt_signalled) mock_onintr++;
            mock_interrupt_signalled = 0;
            if (mock_jmp_armed) { snprintf(mock_error, sizeof mock_error, "interrupt"); longjmp(mock_jmp, 2); }
            return R_NilValue;
        }
    }
    snprintf(mock_error, sizeof mock_error, "mock R cannot evaluate this call");
    if (mock_jmp_armed) longjmp(mock_jmp, 1);
    abort();
}
void Rprintf(const char *fmt, ...)
{
    const size_t used = strlen(mock_output);
    va_list ap; va_start(ap, fmt); vsnprintf(mock_output + used, sizeof mock_output - used, fmt, ap); va_end(ap);
}
static int mock_pending_interrupt(void) { return mock_interrupt_after >= 0 && mock_interrupt_polls++ >= mock_interrupt_after; }
void R_CheckUserInterrupt(void) { /* R long-jumps to top level; under R_ToplevelExec that surfaces as FALSE */ }
Rboolean R_ToplevelExec(void (*fun)(void *), void *data)
{
    if (mock_pending_interrupt()) return FALSE;
    fun(data);
    return TRUE;
}
void GetRNGstate(void) { mock_rng_depth++; }
void PutRNGstate(void) { mock_rng_depth--; }
double unif_rand(void)
{
    mock_rng_calls++;
    mock_rng_state = mock_rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(mock_rng_state >> 11) * (1.0 / 9007199254740992.0);
}
int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e)
{
    (void)info; (void)c; (void)f; (void)e;
    mock_registered = 0;
    for (; call && call->name && mock_registered < 8; call++) mock_calls[mock_registered++] = *call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value) { (void)info; return value; }

/* ---- helpers for the Python side ------------------------------------------------------------------------------ */
SEXP mock_real(const double *v, int nrow, int ncol)
{
    SEXP s = Rf_allocMatrix(REALSXP, nrow, ncol);
    if (v) memcpy(s->data, v, (size_t)nrow * ncol * 8);
    return s;
}
SEXP mock_lgl(const int *v, int nrow, int ncol)
{
    SEXP s = Rf_allocMatrix(LGLSXP, nrow, ncol);
    if (v) memcpy(s->data, v, (size_t)nrow * ncol * 4);
    return s;
}
SEXP mock_int(int v) { return Rf_ScalarInteger(v); }
SEXP mock_dbl(double v) { SEXP s = Rf_allocVector(REALSXP, 1); ((double *)s->data)[0] = v; return s; }
int mock_type(SEXP s) { return (int)s->type; }
long mock_length(SEXP s) { return (long)s->len; }
int mock_nrow(SEXP s) { return s->nrow; }
int mock_ncol(SEXP s) { return s->ncol; }
SEXP mock_elt(SEXP s, int i) { return ((SEXP *)s->data)[i]; }
const char *mock_name(SEXP s, int i) { return (s->names == R_NilValue) ? "" : (const char *)((SEXP *)s->names->data)[i]->data; }
double *mock_real_ptr(SEXP s) { return (double *)s->data; }
int mock_int_value(SEXP s) { return ((int *)s->data)[0]; }
const char *mock_last_error(void) { return mock_error; }
const char *mock_last_warning(void) { return mock_warning_text; }
const char *mock_printed(void) { return mock_output; }
int mock_warning_count(void) { return mock_warnings; }
int mock_protect_balance(void) { return mock_protect_depth; }
int mock_rng_balance(void) { return mock_rng_depth; }
int mock_rng_draws(void) { return mock_rng_calls; }
int mock_onintr_count(void) { return mock_onintr; }
void mock_reset(int interrupt_after)
{
    mock_error[0] = mock_warning_text[0] = mock_output[0] = 0;
    mock_warnings = mock_rng_calls = mock_onintr = 0;
    mock_protect_depth = mock_rng_depth = 0;
    mock_interrupt_after = interrupt_after;
    mock_interrupt_polls = 0;
    mock_rng_state = 12345;
}
int mock_registered_count(void) { return mock_registered; }
const char *mock_registered_name(int i) { return mock_calls[i].name; }
int mock_registered_arity(int i) { return mock_calls[i].numArgs; }

void R_init_NNLM(DllInfo *dll);
void mock_load_package(void) { R_init_NNLM(NULL); }

/* .Call through the registration table, guarded like R's top level: returns NULL after Rf_error / an interrupt raised through the "abort" restart */
typedef SEXP (*call9_t)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*call17_t)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP mock_dotcall(const char *name, int nargs, SEXP *a)
{
    for (int i = 0; i < mock_registered; i++) {
        if (strcmp(mock_calls[i].name, name) != 0) continue;
        if (mock_calls[i].numArgs != nargs) { snprintf(mock_error, sizeof mock_error, "Incorrect number of arguments (%d), expecting %d for '%s'", nargs, mock_calls[i].numArgs, name); return NULL; }
        SEXP out = NULL;
        mock_jmp_armed = 1;
        if (setjmp(mock_jmp) == 0) {
            if (nargs == 9) out = ((call9_t)mock_calls[i].fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]);
            else if (nargs == 17) out = ((call17_t)mock_calls[i].fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16]);
        } else {
            mock_protect_depth = 0; /* R unwinds the protection stack on error */
            mock_rng_depth = 0;
        }
        mock_jmp_armed = 0;
        return out;
    }
    snprintf(mock_error, sizeof mock_error, "\"%s\" not available for .Call() for package \"NNLM\"", name);
    return NULL;
}
