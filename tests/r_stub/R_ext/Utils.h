/* TEST-ONLY stand-in, see ../Rinternals.h. */
#ifndef NNLM_TEST_RUTILS_H
#define NNLM_TEST_RUTILS_H
#include "../Rinternals.h"
void R_CheckUserInterrupt(void);
Rboolean R_ToplevelExec(void (*fun)(void *), void *data);
#endif
