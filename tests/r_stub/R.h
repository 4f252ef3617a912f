/* TEST-ONLY stand-in, see Rinternals.h in this directory. */
#ifndef NNLM_TEST_R_H
#define NNLM_TEST_R_H
#include "Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
void Rprintf(const char *, ...);
#ifdef __cplusplus
}
#endif
#endif
