/* TEST-ONLY declarations of exactly the R API entries pkg/src/r_glue.c uses (this image has no R).  Not part of the
 * product and not a substitute for R's headers: tests/r_stub/mock_r.c implements them just far enough to drive the two
 * .Call entry points from a test. */
#ifndef NNLM_TEST_RINTERNALS_H
#define NNLM_TEST_RINTERNALS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE } Rboolean;
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define CHARSXP 9
extern SEXP R_NilValue, R_NamesSymbol, R_ClassSymbol, R_BaseEnv;
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_allocVector(unsigned int type, R_xlen_t n);
SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol);
SEXP Rf_mkChar(const char *);
SEXP Rf_mkString(const char *);
SEXP Rf_install(const char *);
SEXP Rf_lang2(SEXP, SEXP);
SEXP Rf_eval(SEXP, SEXP);
#define SYMSXP 1
#define LANGSXP 6
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP Rf_setAttrib(SEXP vec, SEXP name, SEXP val);
R_xlen_t XLENGTH(SEXP x);
int *LOGICAL(SEXP x);
double *REAL(SEXP x);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
int Rf_asLogical(SEXP);
SEXP Rf_xlengthgets(SEXP, R_xlen_t);
SEXP Rf_ScalarInteger(int);
void Rf_error(const char *, ...) __attribute__((noreturn));
void Rf_warning(const char *, ...);
#ifdef __cplusplus
}
#endif
#endif
