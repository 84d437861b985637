/* A plain-C host of libsbtv.so (no Python, no torch): the binding a MATLAB MEX file, a Julia ccall or any other
 * FFI would make.  Builds with   gcc -std=c99 -Iinclude examples/c_host.c -L<pkg>/lib -lsbtv -lm -o c_host
 * and runs on an MI355X:  TV-denoise a synthetic image with chambolle_prox_TV_stop, blur it with the 7x7 Gaussian
 * PSF of the demos, deblur it with SALSA_v2, print the checks.  Exit code 0 = every check passed.            */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "sbtv.h"

#define M 128
#define N 64

static int fail(sbtv_ctx *ctx, const char *what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, sbtv_last_error(ctx));
    return 1;
}

int main(void) {
    sbtv_ctx *ctx = NULL;
    int rc = sbtv_ctx_create(0, &ctx);
    if (rc != 0) return fail(NULL, "sbtv_ctx_create", rc);
    printf("libsbtv version %d\n", sbtv_version());

    /* column-major image (i,j) -> j*M + i, as MATLAB stores it: a bright square on a ramp */
    double *x = malloc(sizeof(double) * M * N), *y = malloc(sizeof(double) * M * N), *f = malloc(sizeof(double) * M * N);
    double *px = calloc(M * N, sizeof(double)), *py = calloc(M * N, sizeof(double)), *xm = malloc(sizeof(double) * M * N);
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i)
            x[j * M + i] = 40.0 + 0.5 * i + ((i > 40 && i < 90 && j > 16 && j < 48) ? 120.0 : 0.0);

    /* [f,px,py] = chambolle_prox_TV_stop(x,'lambda',8,'maxiter',25) */
    double lambda = 8.0, err = 0.0, tv_x = 0.0, tv_f = 0.0;
    int k = 0;
    rc = sbtv_chambolle_prox_TV_stop(ctx, x, M, N, 1, &lambda, 25, 1e-3, 0.249, 0, px, py, f, &k, &err, SBTV_HOST_PTRS);
    if (rc != 0) return fail(ctx, "sbtv_chambolle_prox_TV_stop", rc);
    if ((rc = sbtv_TVnorm(ctx, x, M, N, 1, &tv_x, SBTV_HOST_PTRS)) != 0) return fail(ctx, "sbtv_TVnorm", rc);
    if ((rc = sbtv_TVnorm(ctx, f, M, N, 1, &tv_f, SBTV_HOST_PTRS)) != 0) return fail(ctx, "sbtv_TVnorm", rc);
    printf("prox: k = %d, err = %.6g, TV %.1f -> %.1f\n", k, err, tv_x, tv_f);
    int ok = (k == 25) && (tv_f < tv_x);

    /* y = A x with Gaussian_psf(7, 0.4, 0.3, 0);  then x_map = SALSA_v2(y, A, tau, 'MU', mu, ...) */
    double p[3] = {0.4, 0.3, 0.0}, taps[49];
    if ((rc = sbtv_psf_taps(SBTV_PSF_GAUSSIAN, 7, p, taps, NULL, NULL)) != 0) return fail(ctx, "sbtv_psf_taps", rc);
    if ((rc = sbtv_A_wrapper(ctx, taps, 7, NULL, x, y, M, N, 1, 1, SBTV_HOST_PTRS)) != 0) return fail(ctx, "sbtv_A_wrapper", rc);
    sbtv_salsa_opts o;
    sbtv_salsa_opts_default(&o);
    o.maxiter = 200;
    o.TViters = 10;
    o.tolA = 1e-6;
    double tau = 0.05, mu = 0.005, objective[201], distance[200], times[201], mses[201];
    int numA = 0, numAt = 0, n_outer = 0;
    rc = sbtv_SALSA_v2(ctx, y, M, N, 1, taps, 7, &tau, &mu, &o, x, NULL, xm, objective, distance, times, mses, &numA, &numAt,
                       &n_outer, SBTV_HOST_PTRS);
    if (rc != 0) return fail(ctx, "sbtv_SALSA_v2", rc);
    double psnr_y = 0.0, psnr_x = 0.0;
    if ((rc = sbtv_PSNR(ctx, x, y, M, N, 1, &psnr_y, SBTV_HOST_PTRS)) != 0) return fail(ctx, "sbtv_PSNR", rc);
    if ((rc = sbtv_PSNR(ctx, x, xm, M, N, 1, &psnr_x, SBTV_HOST_PTRS)) != 0) return fail(ctx, "sbtv_PSNR", rc);
    printf("SALSA_v2: %d outer iterations, numA = %d, numAt = %d, PSNR blurred %.2f dB -> deblurred %.2f dB\n", n_outer,
           numA, numAt, psnr_y, psnr_x);
    ok = ok && (n_outer >= 2) && (psnr_x > psnr_y + 3.0) && (numA == n_outer + 1) && (mses[n_outer] < mses[0]);

    /* the reference's error() sites come back as negative status codes with the same message */
    rc = sbtv_A_wrapper(ctx, taps, 7, NULL, x, y, M, N, 1, 4, SBTV_HOST_PTRS);
    printf("mode 4 -> %d: %s\n", rc, sbtv_last_error(ctx));
    ok = ok && (rc == SBTV_ERR_MODE);

    /* a batch of three observations in ONE call: dealt to the context's two lanes (two internal streams; include/sbtv.h
     * sbtv_ctx_set_lanes) - every image must come out bit for bit as with the lanes switched off, and image 0 as above */
    {
        enum { B = 3 };
        const size_t P = (size_t)M * N;
        double *yb = malloc(sizeof(double) * P * B), *xb = malloc(sizeof(double) * P * B), *x1 = malloc(sizeof(double) * P * B),
               *x2 = malloc(sizeof(double) * P * B);
        double tapsb[B * 49], taub[B], mub[B], ob1[B * 201], ob2[B * 201];
        int nA[B], nAt[B], nout1[B], nout2[B];
        for (int b = 0; b < B; ++b) {
            for (size_t q = 0; q < P; ++q) {
                yb[b * P + q] = y[q] * (1.0 - 0.1 * b) + 3.0 * b;
                xb[b * P + q] = x[q] * (1.0 - 0.1 * b) + 3.0 * b;
            }
            for (int q = 0; q < 49; ++q) tapsb[b * 49 + q] = taps[q];
            taub[b] = tau * (1.0 + 0.5 * b);
            mub[b] = mu;
        }
        rc = sbtv_SALSA_v2(ctx, yb, M, N, B, tapsb, 7, taub, mub, &o, xb, NULL, x2, ob2, NULL, NULL, NULL, nA, nAt, nout2, SBTV_HOST_PTRS);
        if (rc != 0) return fail(ctx, "sbtv_SALSA_v2 (batch, two lanes)", rc);
        if ((rc = sbtv_ctx_set_lanes(ctx, 1)) != 0) return fail(ctx, "sbtv_ctx_set_lanes", rc);
        rc = sbtv_SALSA_v2(ctx, yb, M, N, B, tapsb, 7, taub, mub, &o, xb, NULL, x1, ob1, NULL, NULL, NULL, nA, nAt, nout1, SBTV_HOST_PTRS);
        if (rc != 0) return fail(ctx, "sbtv_SALSA_v2 (batch, one stream)", rc);
        int same = 1;
        for (size_t q = 0; q < P * B; ++q) same = same && (x1[q] == x2[q]);
        for (int b = 0; b < B; ++b) same = same && (nout1[b] == nout2[b]) && (ob1[b * 201 + nout1[b]] == ob2[b * 201 + nout2[b]]);
        for (size_t q = 0; q < P; ++q) same = same && (x2[q] == xm[q]);         /* image 0 of the batch = the single solve */
        printf("batch of %d in two lanes vs one stream: %s (outer iterations %d / %d / %d)\n", B, same ? "bit-equal" : "DIFFERENT",
               nout2[0], nout2[1], nout2[2]);
        ok = ok && same && (nout2[0] == n_outer);
        /* row-major hosts: the layout helper turns `rows x cols` row-major images into the column-major ones used above */
        double rm[6] = {1, 2, 3, 4, 5, 6}, cm[6];
        if ((rc = sbtv_host_transpose(rm, cm, 1, 2, 3)) != 0) return fail(ctx, "sbtv_host_transpose", rc);
        ok = ok && cm[0] == 1 && cm[1] == 4 && cm[2] == 2 && cm[3] == 5 && cm[4] == 3 && cm[5] == 6;
        free(yb); free(xb); free(x1); free(x2);
    }

    sbtv_ctx_destroy(ctx);
    free(x); free(y); free(f); free(px); free(py); free(xm);
    printf(ok ? "C host: all checks passed\n" : "C host: CHECK FAILED\n");
    return ok ? 0 : 2;
}
