/* A plain-C, single-process host that uses SEVERAL GPUs through sbtv_group (the reference's host is one MATLAB process:
 * run_Gaussian_demo.m:199,229).    gcc -std=c99 -Iinclude examples/c_host_multi.c -L<pkg>/lib -lsbtv -lm -o c_host_multi
 *     ./c_host_multi            every visible GPU once;   ./c_host_multi 0 0 0   three virtual shards on GPU 0
 * Deblurs NIMG independent images with sbtv_SALSA_v2_sharded and checks the result against the same images solved one
 * batch on one context (must be bit-equal: image k of a batch is computed like image k alone).  Exit code 0 = ok.   */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sbtv.h"

#define M 128
#define N 96
#define NIMG 5
#define MAXIT 60

int main(int argc, char **argv) {
    int devices[64], nd = 0;
    for (int a = 1; a < argc && nd < 64; ++a) devices[nd++] = atoi(argv[a]);
    if (nd == 0) {                       /* default: device 0 twice (works on a one-GPU box) */
        devices[nd++] = 0;
        devices[nd++] = 0;
    }
    sbtv_group *g = NULL;
    int rc = sbtv_group_create(devices, nd, &g);
    if (rc != 0) {
        fprintf(stderr, "sbtv_group_create failed (%d): %s\n", rc, sbtv_last_error(NULL));
        return 1;
    }
    printf("group of %d shard(s)\n", sbtv_group_size(g));
    const size_t P = (size_t)M * N;
    double *x = malloc(sizeof(double) * P * NIMG), *y = malloc(sizeof(double) * P * NIMG);
    double *xs = malloc(sizeof(double) * P * NIMG), *x1 = malloc(sizeof(double) * P * NIMG);
    double taps[49 * NIMG], tau[NIMG], mu[NIMG], p[3] = {0.4, 0.3, 0.0};
    for (int b = 0; b < NIMG; ++b) {
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i)
                x[b * P + (size_t)j * M + i] = 30.0 + 10.0 * b + 0.4 * i + ((i > 30 + 5 * b && i < 90 && j > 20 && j < 70 - 4 * b) ? 110.0 : 0.0);
        if ((rc = sbtv_psf_taps(SBTV_PSF_GAUSSIAN, 7, p, taps + 49 * b, NULL, NULL)) != 0) return 2;
        tau[b] = 0.05 + 0.01 * b;
        mu[b] = 0.005;
    }
    sbtv_ctx *c0 = sbtv_group_ctx(g, 0);
    if ((rc = sbtv_A_wrapper(c0, taps, 7, NULL, x, y, M, N, NIMG, 1, SBTV_HOST_PTRS)) != 0) {
        fprintf(stderr, "A_wrapper failed: %s\n", sbtv_last_error(c0));
        return 3;
    }
    sbtv_salsa_opts o;
    sbtv_salsa_opts_default(&o);
    o.maxiter = MAXIT;
    o.TViters = 10;
    o.tolA = 1e-5;
    static double obj_s[NIMG * (MAXIT + 1)], obj_1[NIMG * (MAXIT + 1)];
    int nout_s[NIMG], nout_1[NIMG];
    rc = sbtv_SALSA_v2_sharded(g, y, M, N, NIMG, taps, 7, tau, mu, &o, x, NULL, xs, obj_s, NULL, NULL, NULL, NULL, NULL, nout_s);
    if (rc != 0) {
        fprintf(stderr, "sbtv_SALSA_v2_sharded failed (%d): %s\n", rc, sbtv_group_last_error(g));
        return 4;
    }
    rc = sbtv_SALSA_v2(c0, y, M, N, NIMG, taps, 7, tau, mu, &o, x, NULL, x1, obj_1, NULL, NULL, NULL, NULL, NULL, nout_1, SBTV_HOST_PTRS);
    if (rc != 0) {
        fprintf(stderr, "sbtv_SALSA_v2 failed (%d): %s\n", rc, sbtv_last_error(c0));
        return 5;
    }
    int ok = 1;
    for (int b = 0; b < NIMG; ++b) {
        int shard = -1, first = 0, count = 0;
        sbtv_group_shard_of(g, NIMG, b, &shard, &first, &count);
        const int same = nout_s[b] == nout_1[b] && memcmp(xs + b * P, x1 + b * P, sizeof(double) * P) == 0 &&
                         memcmp(obj_s + b * (MAXIT + 1), obj_1 + b * (MAXIT + 1), sizeof(double) * (nout_1[b] + 1)) == 0;
        printf("image %d on shard %d (block %d..%d): %d outer iterations, objective %.6e %s\n", b, shard, first,
               first + count - 1, nout_s[b], obj_s[b * (MAXIT + 1) + nout_s[b]], same ? "== single context" : "DIFFERS");
        ok = ok && same && nout_s[b] > 1;
    }
    sbtv_group_destroy(g);
    free(x); free(y); free(xs); free(x1);
    printf(ok ? "all checks passed\n" : "CHECK FAILED\n");
    return ok ? 0 : 6;
}
