/* TEST-ONLY stand-in, see ../Rinternals.h. */
#ifndef NNLM_TEST_RDYNLOAD_H
#define NNLM_TEST_RDYNLOAD_H
#include "../Rinternals.h"
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);
#endif
