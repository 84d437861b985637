/* MEX gateway over libsbtv.so for MATLAB installations where loadlibrary is unavailable
 * (no supported C compiler for the thunk file).  Build on a machine with MATLAB:
 *     mex -I../../../include sbtv_mex.c -L../../lib -lsbtv
 * It is NOT built by this repository (mex.h ships with MATLAB only); the calllib shims one directory
 * up are the primary route.  All images are MATLAB doubles, i.e. already the column-major M x N
 * layout the C-ABI expects, so nothing is copied on the host side.
 *
 *   [f, px, py, k, err] = sbtv_mex('chambolle', g, lambda, maxiter, tol, tau [, px0, py0])
 *         chambolle_prox_TV_stop(g,'lambda',..,'maxiter',..,'tol',..,'tau',..,'dualvars',[px0 py0])
 *   t   = sbtv_mex('TVnorm', x)
 *   out = sbtv_mex('A_wrapper', taps, x, mode [, mu])          mode 1 A, 2 AT, 9 invLS
 *   [x, numA, numAt, objective, distance, times, mses] =
 *         sbtv_mex('SALSA_v2', y, taps, tau, mu, [stop maxiter TViters init tolA], true_x_or_[], xinit_or_[])
 */
#include <string.h>

#include "mex.h"
#include "sbtv.h"

static sbtv_ctx *g_ctx = NULL;

static void at_exit(void) {
    if (g_ctx) sbtv_ctx_destroy(g_ctx);
    g_ctx = NULL;
}

static void need_ctx(void) {
    if (g_ctx) return;
    if (sbtv_ctx_create(0, &g_ctx) != 0) mexErrMsgIdAndTxt("sbtv:ctx", "%s", sbtv_last_error(NULL));
    mexAtExit(at_exit);
}

static void check(int rc, const char *who) {
    if (rc != 0) mexErrMsgIdAndTxt("sbtv:call", "%s: %s", who, sbtv_last_error(g_ctx));
}

static const double *opt_image(const mxArray *a) { return (a && !mxIsEmpty(a)) ? mxGetPr(a) : NULL; }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char cmd[32];
    if (nrhs < 2 || mxGetString(prhs[0], cmd, sizeof cmd) != 0) mexErrMsgIdAndTxt("sbtv:usage", "sbtv_mex(command, ...)");
    need_ctx();
    if (strcmp(cmd, "chambolle") == 0) {
        if (nrhs < 6) mexErrMsgIdAndTxt("sbtv:usage", "chambolle: g, lambda, maxiter, tol, tau");
        const int M = (int)mxGetM(prhs[1]), N = (int)mxGetN(prhs[1]);
        const double lambda = mxGetScalar(prhs[2]);
        const int warm = (nrhs >= 8 && !mxIsEmpty(prhs[6]) && !mxIsEmpty(prhs[7]));
        int k = 0;
        double err = 0.0;
        plhs[0] = mxCreateDoubleMatrix(M, N, mxREAL);
        mxArray *px = warm ? mxDuplicateArray(prhs[6]) : mxCreateDoubleMatrix(M, N, mxREAL);
        mxArray *py = warm ? mxDuplicateArray(prhs[7]) : mxCreateDoubleMatrix(M, N, mxREAL);
        check(sbtv_chambolle_prox_TV_stop(g_ctx, mxGetPr(prhs[1]), M, N, 1, &lambda, (int)mxGetScalar(prhs[3]),
                                          mxGetScalar(prhs[4]), mxGetScalar(prhs[5]), warm, mxGetPr(px), mxGetPr(py),
                                          mxGetPr(plhs[0]), &k, &err, SBTV_HOST_PTRS), "chambolle_prox_TV_stop");
        if (nlhs > 1) plhs[1] = px; else mxDestroyArray(px);
        if (nlhs > 2) plhs[2] = py; else mxDestroyArray(py);
        if (nlhs > 3) plhs[3] = mxCreateDoubleScalar((double)k);
        if (nlhs > 4) plhs[4] = mxCreateDoubleScalar(err);
    } else if (strcmp(cmd, "TVnorm") == 0) {
        double t = 0.0;
        check(sbtv_TVnorm(g_ctx, mxGetPr(prhs[1]), (int)mxGetM(prhs[1]), (int)mxGetN(prhs[1]), 1, &t, SBTV_HOST_PTRS), "TVnorm");
        plhs[0] = mxCreateDoubleScalar(t);
    } else if (strcmp(cmd, "A_wrapper") == 0) {
        if (nrhs < 4) mexErrMsgIdAndTxt("sbtv:usage", "A_wrapper: taps, x, mode [, mu]");
        const int M = (int)mxGetM(prhs[2]), N = (int)mxGetN(prhs[2]);
        double mu = (nrhs > 4) ? mxGetScalar(prhs[4]) : 0.0;
        plhs[0] = mxCreateDoubleMatrix(M, N, mxREAL);
        check(sbtv_A_wrapper(g_ctx, mxGetPr(prhs[1]), (int)mxGetM(prhs[1]), (nrhs > 4) ? &mu : NULL, mxGetPr(prhs[2]),
                             mxGetPr(plhs[0]), M, N, 1, (int)mxGetScalar(prhs[3]), SBTV_HOST_PTRS), "A_wrapper");
    } else if (strcmp(cmd, "SALSA_v2") == 0) {
        if (nrhs < 6 || mxGetNumberOfElements(prhs[5]) < 5) mexErrMsgIdAndTxt("sbtv:usage", "SALSA_v2: y, taps, tau, mu, [stop maxiter TViters init tolA], true_x, xinit");
        const int M = (int)mxGetM(prhs[1]), N = (int)mxGetN(prhs[1]);
        const double tau = mxGetScalar(prhs[3]), mu = mxGetScalar(prhs[4]);
        const double *ov = mxGetPr(prhs[5]);
        sbtv_salsa_opts o;
        sbtv_salsa_opts_default(&o);
        o.stopcriterion = (int)ov[0];
        o.maxiter = (int)ov[1];
        o.TViters = (int)ov[2];
        o.initialization = (int)ov[3];
        o.tolA = ov[4];
        const double *tx = (nrhs > 6) ? opt_image(prhs[6]) : NULL, *xi = (nrhs > 7) ? opt_image(prhs[7]) : NULL;
        o.compute_mse = tx != NULL;
        if (xi) o.initialization = 33333;
        int numA = 0, numAt = 0, n = 0;
        mxArray *obj = mxCreateDoubleMatrix(1, o.maxiter + 1, mxREAL), *dist = mxCreateDoubleMatrix(1, o.maxiter, mxREAL);
        mxArray *tim = mxCreateDoubleMatrix(1, o.maxiter + 1, mxREAL), *mse = mxCreateDoubleMatrix(1, o.maxiter + 1, mxREAL);
        plhs[0] = mxCreateDoubleMatrix(M, N, mxREAL);
        check(sbtv_SALSA_v2(g_ctx, mxGetPr(prhs[1]), M, N, 1, mxGetPr(prhs[2]), (int)mxGetM(prhs[2]), &tau, &mu, &o, tx, xi,
                            mxGetPr(plhs[0]), mxGetPr(obj), mxGetPr(dist), mxGetPr(tim), tx ? mxGetPr(mse) : NULL, &numA,
                            &numAt, &n, SBTV_HOST_PTRS), "SALSA_v2");
        mxSetN(obj, n + 1);                        /* trim the traces to the iterations actually run */
        mxSetN(dist, n);
        mxSetN(tim, n + 1);
        mxSetN(mse, tx ? n + 1 : 0);
        mxArray *outs[7] = {plhs[0], mxCreateDoubleScalar(numA), mxCreateDoubleScalar(numAt), obj, dist, tim, mse};
        for (int q = 1; q < 7; ++q) {
            if (q < nlhs) plhs[q] = outs[q]; else mxDestroyArray(outs[q]);
        }
    } else {
        mexErrMsgIdAndTxt("sbtv:usage", "unknown command '%s'", cmd);
    }
}
