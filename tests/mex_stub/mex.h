/* TEST INFRASTRUCTURE, not MATLAB's header: declarations of the handful of documented MEX C-API functions that
 * matlab/mex/sbtv_mex.c uses, so that `gcc -fsyntax-only` can type-check the gateway's calls into include/sbtv.h on a
 * machine without MATLAB (tests/test_matlab_shims.py).  Nothing is linked or run against it. */
#ifndef SBTV_TEST_MEX_STUB_H
#define SBTV_TEST_MEX_STUB_H
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef size_t mwSize;
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
int mexAtExit(void (*fn)(void));
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
int mxIsEmpty(const mxArray *a);
double *mxGetPr(const mxArray *a);
double mxGetScalar(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
void mxSetN(mxArray *a, mwSize n);
size_t mxGetNumberOfElements(const mxArray *a);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxDuplicateArray(const mxArray *a);
void mxDestroyArray(mxArray *a);
#endif
