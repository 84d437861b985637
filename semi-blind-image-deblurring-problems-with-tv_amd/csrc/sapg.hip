// FISTA, power iteration and the SAPG (MYULA) parameter-estimation loop.
#include <chrono>
#include <cmath>

#include "sbtv_internal.h"

using namespace sbtv;

extern "C" {

int sbtv_fista_tv(sbtv_ctx *ctx, const double *b, int M, int N, int batch, const double *taps, int taille,
                  const double *tau, double L, int prox_iters, int stopcriterion, double tolerance, int maxiters,
                  int zero_start, const double *true_x, double *x_out, double *objective, double *mses, int *n_iter,
                  int flags) {
    return fail(ctx, SBTV_ERR_BADARG, "fista_tv: not implemented yet");
}

int sbtv_SAPG_algorithm(sbtv_ctx *ctx, const double *y, int M, int N, int batch, const sbtv_sapg_opts *op,
                        const double *x0, const double *noise, double *thetas, double *ps, double *sigmas,
                        double *logpi, double *logpi_wu, double *gx, double *grads, double *eb, double *x_last,
                        sbtv_allreduce_fn reduce_fn, void *reduce_user, int flags) {
    return fail(ctx, SBTV_ERR_BADARG, "SAPG_algorithm: not implemented yet");
}

int sbtv_max_eigenval(sbtv_ctx *ctx, const double *taps, int taille, const double *x0, int M, int N, double tol,
                      int max_iter, double *val, int *iters, int flags) {
    return fail(ctx, SBTV_ERR_BADARG, "max_eigenval: not implemented yet");
}

}  // extern "C"
