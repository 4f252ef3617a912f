/* TEST-ONLY stand-in, see ../Rinternals.h. */
#ifndef NNLM_TEST_RRANDOM_H
#define NNLM_TEST_RRANDOM_H
void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);
#endif
