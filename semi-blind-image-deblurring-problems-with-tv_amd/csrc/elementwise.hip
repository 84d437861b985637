// Fused element-wise passes (K4) of the FISTA / MYULA iterations (SALSA's pass is fused into the inverse
// column FFT, fft.hip), the
// scalar collectors and the metrics.  Every pass is 16-byte-per-lane
// vectorised (images have an even number of elements) and every reduction is a
// fixed-order two-level sum (per-block partials -> one block per quantity).
#include "sbtv_internal.h"

#pragma clang fp contract(off)

namespace sbtv {

constexpr int EWB = 256;

__device__ __forceinline__ double ew_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block reduce NQ quantities; thread 0 gets the totals
template <int NQ>
__device__ __forceinline__ void ew_block_sum(double (&v)[NQ], double *red /*[NQ*4]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
        const double a = ew_wave_sum(v[c]);
        if (lane == 0) red[c * 4 + w] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < NQ; ++c) v[c] = (red[c * 4] + red[c * 4 + 1]) + (red[c * 4 + 2] + red[c * 4 + 3]);
    }
}

int ew_blocks(size_t P) {
    size_t nb = (P / 2 + EWB - 1) / EWB;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// SALSA from the zero start (x = u = bu = 0, SALSA_v2.m:366-393): one pass clears the three images and leaves the
// partial sums of the initial objective and mse (:399-401): resid = y - A(0) = y, so ||resid||^2 = sum y^2 (stored in
// the units of the row-pass partials, i.e. of the unnormalised spectrum: x M N), TVnorm(0) = 0, mse(1) = sum true^2 / P.
// accp [batch][3][G] (slot 0 is read), postp [batch][6][G] (slot 0 = mse numerator, slot 5 = TV, the others zero);
// the collector reduces them in the usual fixed order.
__global__ __launch_bounds__(EWB) void salsa_zero_start_kernel(const double *__restrict__ y, const double *__restrict__ tru,
                                                                double *__restrict__ x, double *__restrict__ u,
                                                                double *__restrict__ bu, double *__restrict__ accp,
                                                                double *__restrict__ postp, size_t P, double scale) {
    const int b = blockIdx.y;
    __shared__ double red[2 * 4];
    const size_t base = (size_t)b * P;
    double acc[2] = {0, 0};
    for (size_t q = (size_t)blockIdx.x * EWB + threadIdx.x; q < P; q += (size_t)gridDim.x * EWB) {
        const double yv = y[base + q];
        const double tv = tru ? tru[base + q] : 0.0;
        acc[0] += yv * yv;
        acc[1] += tv * tv;
        x[base + q] = 0.0;
        u[base + q] = 0.0;
        bu[base + q] = 0.0;
    }
    ew_block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        const size_t G = gridDim.x;
        for (int q = 0; q < 3; ++q) accp[((size_t)b * 3 + q) * G + blockIdx.x] = (q == 0) ? acc[0] * scale : 0.0;
        for (int q = 0; q < 6; ++q) postp[((size_t)b * 6 + q) * G + blockIdx.x] = (q == 0) ? acc[1] : 0.0;
    }
}

// generic two-array sums: partials[b][4][nb] = sum (a-b)^2, sum a^2, sum b^2, max a
__global__ __launch_bounds__(EWB) void pair_sums_kernel(const double *__restrict__ a, const double *__restrict__ c,
                                                         double *__restrict__ partials, size_t P) {
    const int b = blockIdx.y;
    __shared__ double red[3 * 4];
    __shared__ double redm[4];
    const size_t base = (size_t)b * P;
    double acc[3] = {0, 0, 0};
    double mx = -1.0e308;
    for (size_t q = (size_t)blockIdx.x * EWB + threadIdx.x; q < P; q += (size_t)gridDim.x * EWB) {
        const double av = a[base + q];
        const double cv = c ? c[base + q] : 0.0;
        const double d = av - cv;
        acc[0] += d * d;
        acc[1] += av * av;
        acc[2] += cv * cv;
        mx = fmax(mx, av);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
    ew_block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        for (int q = 0; q < 3; ++q) partials[((size_t)b * 4 + q) * gridDim.x + blockIdx.x] = acc[q];
        partials[((size_t)b * 4 + 3) * gridDim.x + blockIdx.x] = fmax(fmax(redm[0], redm[1]), fmax(redm[2], redm[3]));
    }
}

__global__ __launch_bounds__(256) void reduce_max_kernel(const double *__restrict__ partials, int n,
                                                          double *__restrict__ out) {
    __shared__ double redm[4];
    const double *p = partials + (size_t)blockIdx.x * n;
    double mx = -1.0e308;
    for (int q = threadIdx.x; q < n; q += 256) mx = fmax(mx, p[q]);
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmax(fmax(redm[0], redm[1]), fmax(redm[2], redm[3]));
}

// ---- FISTA (SALSA/my_fista.m:25,29)
//   grad step : y = y - (1/L) * grad
__global__ __launch_bounds__(EWB) void axpy_kernel(double *__restrict__ y, const double *__restrict__ gr, double a,
                                                    size_t Ptot, ProxArm arm, int batch) {
    if (arm.ctrl && blockIdx.x == 0 && (int)threadIdx.x < batch) {
        const int b = threadIdx.x;         // prox_reset(keep_cur = false) for the cold-start prox that follows
        ProxCtrl c = arm.ctrl[b];
        c.k = 0;
        c.done = (arm.frozen && arm.frozen[b]) ? 1 : 0;
        c.cur = 0;
        c.maxiter = arm.maxiter;
        c.redo = 0;
        c.f_valid = 0;
        c.err = 0.0;
        c.lambda = arm.lambda[b];
        c.tol = arm.tol;
        c.tau = arm.tau;
        arm.ctrl[b] = c;
    }
    for (size_t q = (size_t)blockIdx.x * EWB + threadIdx.x; q < Ptot / 2; q += (size_t)gridDim.x * EWB) {
        double2 yv = *reinterpret_cast<double2 *>(y + 2 * q);
        const double2 gv = *reinterpret_cast<const double2 *>(gr + 2 * q);
        yv.x = yv.x - a * gv.x;
        yv.y = yv.y - a * gv.y;
        *reinterpret_cast<double2 *>(y + 2 * q) = yv;
    }
}
//   momentum  : y = x + c (x - x_old) ; sums (x-true)^2, (x-x_old)^2, x^2   partials [batch][3][nb]
//   (x_old = x of my_fista.m:30 is not a copy: the caller double-buffers x, the previous iterate IS the other buffer)
__global__ __launch_bounds__(EWB) void fista_momentum_kernel(const double *__restrict__ x, const double *__restrict__ xold,
                                                              double *__restrict__ y, const double *__restrict__ tru,
                                                              double c, double *__restrict__ partials, size_t P,
                                                              const int *__restrict__ frozen) {
    const int b = blockIdx.y;
    if (frozen && frozen[b]) return;
    __shared__ double red[3 * 4];
    const size_t base = (size_t)b * P;
    double acc[3] = {0, 0, 0};
    for (size_t q = (size_t)blockIdx.x * EWB + threadIdx.x; q < P / 2; q += (size_t)gridDim.x * EWB) {
        const size_t o = base + 2 * q;
        const double2 xv = *reinterpret_cast<const double2 *>(x + o);
        const double2 xo = *reinterpret_cast<const double2 *>(xold + o);
        const double d0 = xv.x - xo.x, d1 = xv.y - xo.y;
        *reinterpret_cast<double2 *>(y + o) = make_double2(xv.x + c * d0, xv.y + c * d1);
        if (tru) {
            const double2 tv = *reinterpret_cast<const double2 *>(tru + o);
            const double e0 = xv.x - tv.x, e1 = xv.y - tv.y;
            acc[0] += e0 * e0 + e1 * e1;
        }
        acc[1] += d0 * d0 + d1 * d1;
        acc[2] += xv.x * xv.x + xv.y * xv.y;
    }
    ew_block_sum<3>(acc, red);
    if (threadIdx.x == 0)
        for (int q = 0; q < 3; ++q) partials[((size_t)b * 3 + q) * gridDim.x + blockIdx.x] = acc[q];
}

// ---- MYULA step (SAPG/SAPG_algorithm_Guassian.m:80-81,161):
//   X = abs( X + gam*(prox - X)/lamb - gam*gradF + sqrt(2 gam) * Z ),  gradF = grad / sigma2
__global__ __launch_bounds__(EWB) void myula_step_kernel(double *__restrict__ X, const double *__restrict__ prox,
                                                          const double *__restrict__ grad,
                                                          const double *__restrict__ Z,
                                                          const double *__restrict__ sigma2, double gam, double lamb,
                                                          double sq2g, size_t P, RngArgs rng, ProxArm arm) {
    const int b = blockIdx.y;
    if (arm.ctrl && blockIdx.x == 0 && threadIdx.x == 0) {
        ProxCtrl c = arm.ctrl[b];          // prox_reset(keep_cur = false) for the cold-start prox that follows
        c.k = 0;
        c.done = 0;
        c.cur = 0;
        c.maxiter = arm.maxiter;
        c.redo = 0;
        c.f_valid = 0;
        c.err = 0.0;
        c.lambda = arm.lambda[b];
        c.tol = arm.tol;
        c.tau = arm.tau;
        arm.ctrl[b] = c;
    }
    const size_t base = (size_t)b * P;
    const double s2 = sigma2[b];
    const unsigned step = rng.step_dev ? (unsigned)rng.step_dev[0] : rng.step;
    for (size_t q = (size_t)blockIdx.x * EWB + threadIdx.x; q < P / 2; q += (size_t)gridDim.x * EWB) {
        const size_t o = base + 2 * q;
        const double2 xv = *reinterpret_cast<const double2 *>(X + o);
        const double2 pv = *reinterpret_cast<const double2 *>(prox + o);
        const double2 gv = *reinterpret_cast<const double2 *>(grad + o);
        const double2 zv = Z ? *reinterpret_cast<const double2 *>(Z + o)
                             : philox_normal_pair(q, step, rng.chain0 + (unsigned)b, rng.seed);
        double2 r;
        r.x = myula_nocontract(xv.x, pv.x, gv.x, 1.0, zv.x, gam, lamb, s2, sq2g);
        r.y = myula_nocontract(xv.y, pv.y, gv.y, 1.0, zv.y, gam, lamb, s2, sq2g);
        *reinterpret_cast<double2 *>(X + o) = r;
    }
}

// ---- plain MYULA chain (SALSA/myula.m:16): no abs(), the reference's own grouping of the terms
//   x = (1 - gam/lamb) x - gam (grad/sigma2 - prox/lamb) + sqrt(2 gam) z
__global__ __launch_bounds__(EWB) void myula_plain_kernel(double *__restrict__ X, const double *__restrict__ prox,
                                                           const double *__restrict__ grad,
                                                           const double *__restrict__ Z,
                                                           const double *__restrict__ sigma2, double gam, double lamb,
                                                           double sq2g, size_t P, RngArgs rng, ProxArm arm) {
    const int b = blockIdx.y;
    if (arm.ctrl && blockIdx.x == 0 && threadIdx.x == 0) {
        ProxCtrl c = arm.ctrl[b];
        c.k = 0;
        c.done = 0;
        c.cur = 0;
        c.maxiter = arm.maxiter;
        c.redo = 0;
        c.f_valid = 0;
        c.err = 0.0;
        c.lambda = arm.lambda[b];
        c.tol = arm.tol;
        c.tau = arm.tau;
        arm.ctrl[b] = c;
    }
    const size_t base = (size_t)b * P;
    const double s2 = sigma2[b], keep = 1.0 - gam / lamb;
    for (size_t q = (size_t)blockIdx.x * EWB + threadIdx.x; q < P / 2; q += (size_t)gridDim.x * EWB) {
        const size_t o = base + 2 * q;
        const double2 xv = *reinterpret_cast<const double2 *>(X + o);
        const double2 pv = *reinterpret_cast<const double2 *>(prox + o);
        const double2 gv = *reinterpret_cast<const double2 *>(grad + o);
        const double2 zv = Z ? *reinterpret_cast<const double2 *>(Z + o)
                             : philox_normal_pair(q, rng.step, rng.chain0 + (unsigned)b, rng.seed);
        double2 r;
        r.x = (keep * xv.x - gam * (gv.x / s2 - pv.x / lamb)) + sq2g * zv.x;
        r.y = (keep * xv.y - gam * (gv.y / s2 - pv.y / lamb)) + sq2g * zv.y;
        *reinterpret_cast<double2 *>(X + o) = r;
    }
}

// --------------------------------------------------------------------------
// host wrappers
// --------------------------------------------------------------------------
int myula_plain_step(sbtv_ctx *ctx, double *X, const double *prox, const double *grad, const double *Z,
                     const double *sigma2_dev, double gam, double lamb, size_t P, int batch, const RngArgs *rng,
                     const ProxArm *arm) {
    if (!Z && !rng) return fail(ctx, SBTV_ERR_BADARG, "myula_plain_step: neither a noise array nor generator arguments");
    const RngArgs r = rng ? *rng : RngArgs{0ull, 0u, 0u, nullptr};
    const ProxArm pa = arm ? *arm : ProxArm{nullptr, nullptr, 0, 0.0, 0.0, nullptr};
    hipLaunchKernelGGL(myula_plain_kernel, dim3(ew_blocks(P), batch), dim3(EWB), 0, ctx->stream, X, prox, grad, Z,
                       sigma2_dev, gam, lamb, sqrt(2 * gam), P, r, pa);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int pair_sums(sbtv_ctx *ctx, const double *a, const double *c, size_t P, int batch, double *out4_dev) {
    const int nb = ew_blocks(P);
    double *partials = nullptr;
    SBTV_TRY(ws_get_t(ctx, "ew.pair.partials", (size_t)batch * 4 * nb, &partials));
    hipLaunchKernelGGL(pair_sums_kernel, dim3(nb, batch), dim3(EWB), 0, ctx->stream, a, c, partials, P);
    SBTV_HIP(ctx, hipGetLastError());
    // sums: vectors (b,0..2); max: vector (b,3)
    for (int b = 0; b < batch; ++b) {
        SBTV_TRY(reduce_partials(ctx, partials + (size_t)b * 4 * nb, 3, nb, out4_dev + (size_t)b * 4));
        hipLaunchKernelGGL(reduce_max_kernel, dim3(1), dim3(256), 0, ctx->stream, partials + ((size_t)b * 4 + 3) * nb,
                           nb, out4_dev + (size_t)b * 4 + 3);
    }
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int salsa_zero_start(sbtv_ctx *ctx, const double *y, const double *tru, double *x, double *u, double *bu, size_t P, int batch,
                     double scale, double **accp, double **postp, int *nblk) {
    const int G = ew_blocks(2 * P);
    SBTV_TRY(ws_get_t(ctx, "ew.zero.acc", (size_t)batch * 3 * G, accp));
    SBTV_TRY(ws_get_t(ctx, "ew.zero.post", (size_t)batch * 6 * G, postp));
    hipLaunchKernelGGL(salsa_zero_start_kernel, dim3(G, batch), dim3(EWB), 0, ctx->stream, y, tru, x, u, bu, *accp, *postp, P,
                       scale);
    SBTV_HIP(ctx, hipGetLastError());
    *nblk = G;
    return 0;
}

int axpy(sbtv_ctx *ctx, double *y, const double *gr, double a, size_t Ptot, const ProxArm *arm, int batch) {
    if (arm && batch > EWB) return fail(ctx, SBTV_ERR_BADARG, "axpy: cannot arm more than 256 control blocks");
    const ProxArm pa = arm ? *arm : ProxArm{nullptr, nullptr, 0, 0.0, 0.0, nullptr};
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(Ptot)), dim3(EWB), 0, ctx->stream, y, gr, a, Ptot, pa, batch);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int fista_momentum(sbtv_ctx *ctx, const double *x, const double *xold, double *y, const double *tru, double coef,
                   double *partials, size_t P, int batch, const int *frozen) {
    hipLaunchKernelGGL(fista_momentum_kernel, dim3(ew_blocks(P), batch), dim3(EWB), 0, ctx->stream, x, xold, y, tru,
                       coef, partials, P, frozen);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int myula_step(sbtv_ctx *ctx, double *X, const double *prox, const double *grad, const double *Z,
               const double *sigma2_dev, double gam, double lamb, size_t P, int batch, const RngArgs *rng,
               const ProxArm *arm) {
    if (!Z && !rng) return fail(ctx, SBTV_ERR_BADARG, "myula_step: neither a noise array nor generator arguments");
    const RngArgs r = rng ? *rng : RngArgs{0ull, 0u, 0u, nullptr};
    const ProxArm pa = arm ? *arm : ProxArm{nullptr, nullptr, 0, 0.0, 0.0, nullptr};
    hipLaunchKernelGGL(myula_step_kernel, dim3(ew_blocks(P), batch), dim3(EWB), 0, ctx->stream, X, prox, grad, Z,
                       sigma2_dev, gam, lamb, sqrt(2 * gam), P, r, pa);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace sbtv

using namespace sbtv;

extern "C" {

static int metric_common(sbtv_ctx *ctx, const double *x_true, const double *x, int M, int N, int batch, double *out,
                         int flags, bool psnr) {
    if (!ctx) return SBTV_ERR_BADARG;
    if (!x_true || !x || !out || M < 1 || N < 1 || batch < 1) return fail(ctx, SBTV_ERR_BADARG, "metric: bad arguments");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)M * N;
    const double *a = nullptr, *c = nullptr;
    SBTV_TRY(stage_in(ctx, "metric.a", x_true, P * batch, flags, &a));
    SBTV_TRY(stage_in(ctx, "metric.c", x, P * batch, flags, &c));
    double *o4 = nullptr;
    SBTV_TRY(ws_get_t(ctx, "metric.out", (size_t)batch * 4, &o4));
    SBTV_TRY(pair_sums(ctx, a, c, P, batch, o4));
    std::vector<double> h((size_t)batch * 4);
    SBTV_HIP(ctx, hipMemcpyAsync(h.data(), o4, sizeof(double) * batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < batch; ++b) {
        const double se = h[(size_t)b * 4], mx = h[(size_t)b * 4 + 3];
        if (psnr)
            out[b] = 10 * log10(mx * mx) - 10 * log10(se / (double)P);   // utils/PSNR.m:3-4
        else
            out[b] = 10 * log10(se / (double)P);                          // utils/MSE.m:3
    }
    return canary_epilogue(ctx, 0);
}

int sbtv_PSNR(sbtv_ctx *ctx, const double *x_true, const double *x, int M, int N, int batch, double *out, int flags) {
    return metric_common(ctx, x_true, x, M, N, batch, out, flags, true);
}
int sbtv_MSE(sbtv_ctx *ctx, const double *x_true, const double *x, int M, int N, int batch, double *out, int flags) {
    return metric_common(ctx, x_true, x, M, N, batch, out, flags, false);
}

}  // extern "C"
