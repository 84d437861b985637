// FISTA (SALSA/my_fista.m, my_deblur_fista.m), the power iteration
// (utils/max_eigenval_*.m) and the SAPG / MYULA parameter-estimation loop
// (SAPG/SAPG_algorithm_Guassian.m, _moffat.m, _laplace.m) as device-resident
// loops over the TV-prox and spectral-operator kernels.
#include <chrono>
#include <cmath>
#include <cstring>

#include "sbtv_internal.h"

namespace sbtv {

__global__ __launch_bounds__(256) void scale_kernel(double *__restrict__ x, double a, size_t n2) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n2; q += (size_t)gridDim.x * 256) {
        double2 v = reinterpret_cast<double2 *>(x)[q];
        v.x *= a;
        v.y *= a;
        reinterpret_cast<double2 *>(x)[q] = v;
    }
}

static int launch_scale(sbtv_ctx *ctx, double *x, double a, size_t n) {
    hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks(n)), dim3(256), 0, ctx->stream, x, a, n / 2);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

// SAPG scalars of one iteration in one launch: block (q, b) reduces, in a fixed order, the rows-kernel accumulator
// q < 3 of image b (||AX-y||^2 and the two <dA_p X, AX-y> sums, [batch][3][nrb]) or, for q = 3, the periodic-TV
// partials ([batch][ntv]) and writes the total where the host reads it: out[b*3 + q] resp. out[3*batch + b].
// `out` is the device view of pinned host memory, so no copy kernel follows.
__global__ __launch_bounds__(256) void sapg_collect_kernel(const double *__restrict__ acc, int nrb,
                                                           const double *__restrict__ tvp, int ntv,
                                                           double *__restrict__ out, int batch) {
    __shared__ double red[4];
    const int q = blockIdx.x, b = blockIdx.y;
    const double *p = (q < 3) ? acc + ((size_t)b * 3 + q) * nrb : tvp + (size_t)b * ntv;
    const int n = (q < 3) ? nrb : ntv;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(q < 3) ? (size_t)b * 3 + q : 3 * (size_t)batch + b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// FISTA scalars of one iteration in one launch: block (q, b): q < 3 rows-kernel accumulators [batch][3][nrb] ->
// out[b*3+q]; q = 3..5 momentum-kernel sums [batch][3][npb] (may be null) -> out[3*batch + b*3 + (q-3)];
// q = 6 periodic-TV partials [batch][ntv] -> out[6*batch + b].  `out` is the device view of pinned host memory.
__global__ __launch_bounds__(256) void fista_collect_kernel(const double *__restrict__ acc, int nrb,
                                                            const double *__restrict__ mom, int npb,
                                                            const double *__restrict__ tvp, int ntv,
                                                            double *__restrict__ out, int batch) {
    __shared__ double red[4];
    const int q = blockIdx.x, b = blockIdx.y;
    const double *p;
    int n;
    size_t o;
    if (q < 3) {
        p = acc + ((size_t)b * 3 + q) * nrb;
        n = nrb;
        o = (size_t)b * 3 + q;
    } else if (q < 6) {
        p = mom ? mom + ((size_t)b * 3 + (q - 3)) * npb : nullptr;
        n = npb;
        o = 3 * (size_t)batch + (size_t)b * 3 + (q - 3);
    } else {
        p = tvp + (size_t)b * ntv;
        n = ntv;
        o = 6 * (size_t)batch + b;
    }
    double s = 0.0;
    if (p)
        for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[o] = (red[0] + red[1]) + (red[2] + red[3]);
}

// sums of the rows-kernel accumulators: out[b*3 + c]
static int reduce_acc(sbtv_ctx *ctx, const double *acc, int batch, int nrb, double *out_dev) {
    return reduce_partials(ctx, acc, batch * 3, nrb, out_dev);
}

}  // namespace sbtv

using namespace sbtv;

extern "C" {

// ---------------------------------------------------------------------------
// a-9: power iteration on A'A
// ---------------------------------------------------------------------------
int sbtv_max_eigenval(sbtv_ctx *ctx, const double *taps, int taille, const double *x0, int M, int N, double tol,
                      int max_iter, double *val_out, int *iters, int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    if (!taps || !x0 || !val_out) return fail(ctx, SBTV_ERR_BADARG, "max_eigenval: bad arguments");
    if (taille < 1 || taille > 15 || taille > M || taille > N) return fail(ctx, SBTV_ERR_PSF, "Mask does not fit inside array");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    if (((size_t)M * N) & 1)
        return fail(ctx, SBTV_ERR_SIZE, "this entry point needs an even number of pixels (its element-wise passes move two per lane)");
    FftPlan fp;
    SBTV_TRY(fft_plan(ctx, M, N, 1, &fp));
    const size_t P = (size_t)M * N;
    const double *x0d = nullptr;
    SBTV_TRY(stage_in(ctx, "ev.x0", x0, P, flags, &x0d));
    double *x = nullptr, *taps_d = nullptr, *o4 = nullptr;
    double2 *S = nullptr, *Hs = nullptr;
    SBTV_TRY(ws_get_t(ctx, "ev.x", P, &x));
    SBTV_TRY(ws_get_t(ctx, "ev.taps", (size_t)taille * taille, &taps_d));
    SBTV_TRY(ws_get_t(ctx, "ev.o4", 4, &o4));
    SBTV_TRY(ws_get_t(ctx, "ev.S", fp.s_img, &S));
    SBTV_TRY(ws_get_t(ctx, "ev.H", fp.u_img, &Hs));
    SBTV_HIP(ctx, hipMemcpyAsync(taps_d, taps, sizeof(double) * taille * taille, hipMemcpyHostToDevice, ctx->stream));
    SBTV_HIP(ctx, hipMemcpyAsync(x, x0d, sizeof(double) * P, hipMemcpyDeviceToDevice, ctx->stream));
    SBTV_TRY(psf_spectrum(ctx, fp, taps_d, taille, Hs));
    double h4[4];
    auto norm_x = [&](double *nrm) -> int {
        SBTV_TRY(pair_sums(ctx, x, nullptr, P, 1, o4));
        SBTV_HIP(ctx, hipMemcpyAsync(h4, o4, sizeof(h4), hipMemcpyDeviceToHost, ctx->stream));
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *nrm = sqrt(h4[1]);
        return 0;
    };
    double nrm = 0.0;
    SBTV_TRY(norm_x(&nrm));
    SBTV_TRY(launch_scale(ctx, x, 1.0 / nrm, P));              // x = x / norm(x(:))          (:5)
    double init_val = 1.0, val = 1.0;
    int k = 0;
    const double inv_scale = 1.0 / ((double)fp.n1 * N);
    for (k = 1; k <= max_iter; ++k) {
        RowsArgs a{};
        a.dir_fwd = 1;
        a.dir_inv = 1;
        a.op = OP_ATA;                                          // y = A(x); x = At(y)        (:9-10)
        a.H = Hs;
        SBTV_TRY(fft_cols_fwd(ctx, fp, x, nullptr, S));
        SBTV_TRY(fft_rows(ctx, fp, S, S, a));
        SBTV_TRY(fft_cols_inv(ctx, fp, S, x, inv_scale));
        SBTV_TRY(norm_x(&val));                                 // val = norm(x(:))            (:11)
        const double rel_var = fabs(val - init_val) / init_val;
        if (rel_var < tol) break;                               //                            (:16-18)
        init_val = val;
        SBTV_TRY(launch_scale(ctx, x, 1.0 / val, P));          // x = x / val                (:20)
    }
    *val_out = val;
    if (iters) *iters = (k > max_iter) ? max_iter : k;
    return canary_epilogue(ctx, 0);
}

// ---------------------------------------------------------------------------
// a-8: FISTA with Psi = cold-start Chambolle, Phi = periodic TVnorm
// ---------------------------------------------------------------------------
int sbtv_fista_tv(sbtv_ctx *ctx, const double *bimg, int M, int N, int batch, const double *taps, int taille,
                  const double *tau, double L, int prox_iters, int stopcriterion, double tolerance, int maxiters,
                  int zero_start, const double *true_x, double *x_out, double *objective, double *mses, int *n_iter,
                  int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    if (!bimg || !taps || !tau || !true_x || batch < 1 || maxiters < 1 || !(L > 0.0))
        return fail(ctx, SBTV_ERR_BADARG, "fista_tv: bad arguments (b, taps, tau, true are required)");
    if (stopcriterion < 1 || stopcriterion > 3) return fail(ctx, SBTV_ERR_STOPCRITERION, "Invalid stopping criterion!");
    if (prox_iters <= 0) return fail(ctx, SBTV_ERR_MAXITER, "fista_tv: prox_iters must be positive");
    if (taille < 1 || taille > 15 || taille > M || taille > N) return fail(ctx, SBTV_ERR_PSF, "Mask does not fit inside array");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    if (((size_t)M * N) & 1)
        return fail(ctx, SBTV_ERR_SIZE, "this entry point needs an even number of pixels (its element-wise passes move two per lane)");
    FftPlan fp;
    SBTV_TRY(fft_plan(ctx, M, N, batch, &fp));
    ProxPlan pp;
    SBTV_TRY(prox_plan(ctx, M, N, batch, &pp));
    const size_t P = (size_t)M * N, cnt = P * batch;
    const double *bd = nullptr, *td = nullptr;
    SBTV_TRY(stage_in(ctx, "fista.b", bimg, cnt, flags, &bd));
    SBTV_TRY(stage_in(ctx, "fista.true", true_x, cnt, flags, &td));
    double *x = nullptr, *xold = nullptr, *y = nullptr, *grad = nullptr, *xfinal = nullptr;
    SBTV_TRY(ws_get_t(ctx, "fista.x", cnt, &x));
    SBTV_TRY(ws_get_t(ctx, "fista.xold", cnt, &xold));
    SBTV_TRY(ws_get_t(ctx, "fista.y", cnt, &y));
    SBTV_TRY(ws_get_t(ctx, "fista.grad", cnt, &grad));
    SBTV_TRY(stage_out_buf(ctx, "fista.xfinal", x_out, cnt, flags, &xfinal));
    double2 *S = nullptr, *Hs = nullptr, *Bs = nullptr;
    SBTV_TRY(ws_get_t(ctx, "fista.S", (size_t)batch * fp.s_img, &S));
    SBTV_TRY(ws_get_t(ctx, "fista.H", (size_t)batch * fp.u_img, &Hs));
    SBTV_TRY(ws_get_t(ctx, "fista.B", (size_t)batch * fp.u_img, &Bs));
    const size_t npar = (size_t)batch * taille * taille + 2 * (size_t)batch;
    double *par = nullptr;
    SBTV_TRY(ws_get_t(ctx, "fista.par", npar, &par));
    double *taps_d = par, *lam_d = par + (size_t)batch * taille * taille;
    std::vector<double> hpar(npar);
    for (size_t q = 0; q < (size_t)batch * taille * taille; ++q) hpar[q] = taps[q];
    for (int b = 0; b < batch; ++b) {
        hpar[(size_t)batch * taille * taille + b] = tau[b] / L;     // Psi(y, tau/L)   (my_fista.m:26)
        hpar[(size_t)batch * taille * taille + batch + b] = 0.0;
    }
    SBTV_HIP(ctx, hipMemcpyAsync(par, hpar.data(), sizeof(double) * npar, hipMemcpyHostToDevice, ctx->stream));
    int *frozen_d = nullptr;
    SBTV_TRY(ws_get_t(ctx, "fista.frozen", (size_t)batch, &frozen_d));
    SBTV_HIP(ctx, hipMemsetAsync(frozen_d, 0, sizeof(int) * batch, ctx->stream));
    const int nrb = fft_rows_blocks(fp), npb = ew_blocks(P);
    double *acc = nullptr, *momp = nullptr, *o4 = nullptr;
    SBTV_TRY(ws_get_t(ctx, "fista.acc", (size_t)batch * 3 * nrb, &acc));
    SBTV_TRY(ws_get_t(ctx, "fista.momp", (size_t)batch * 3 * npb, &momp));
    SBTV_TRY(ws_get_t(ctx, "fista.o4", (size_t)batch * 4, &o4));
    double *scal_h = nullptr, *scal_hd = nullptr;      // pinned [acc3 | mom3 | tv | pad] per image, host / device view
    {
        void *pz = nullptr, *dp = nullptr;
        SBTV_TRY(pinned_get(ctx, sizeof(double) * 8 * batch, &pz));
        scal_h = static_cast<double *>(pz);
        SBTV_HIP(ctx, hipHostGetDevicePointer(&dp, pz, 0));
        scal_hd = static_cast<double *>(dp);
    }
    const double inv_scale = 1.0 / ((double)fp.n1 * N), parseval = 1.0 / ((double)M * N);
    SBTV_TRY(psf_spectrum(ctx, fp, taps_d, taille, Hs));
    {
        RowsArgs a{};
        a.dir_fwd = 1;
        SBTV_TRY(fft_cols_fwd(ctx, fp, bd, nullptr, S));
        SBTV_TRY(fft_rows(ctx, fp, S, S, a));
        SBTV_TRY(spec_unpack(ctx, fp, S, Bs));
    }
    // x = AT(b) (my_fista.m:7) or zeros (my_deblur_fista.m:21)
    if (zero_start) {
        SBTV_HIP(ctx, hipMemsetAsync(x, 0, sizeof(double) * cnt, ctx->stream));
    } else {
        RowsArgs a{};
        a.dir_fwd = 1;
        a.dir_inv = 1;
        a.op = OP_MUL_HC;
        a.H = Hs;
        SBTV_TRY(fft_cols_fwd(ctx, fp, bd, nullptr, S));
        SBTV_TRY(fft_rows(ctx, fp, S, S, a));
        SBTV_TRY(fft_cols_inv(ctx, fp, S, x, inv_scale));
    }
    SBTV_HIP(ctx, hipMemcpyAsync(y, x, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
    SBTV_HIP(ctx, hipMemcpyAsync(xold, x, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));

    // objective(k) = 0.5*||A x - b||^2 + tau*Phi(x) ; mses(k)   (:14-15, :31-33)
    // residual energy (Parseval) and TV partials of x, then ONE collector launch that reduces them (and the
    // momentum-kernel sums when given) straight into pinned host memory
    auto objective_of_x = [&](const int *frozen, const double *mom_partials) -> int {
        RowsArgs a{};
        a.dir_fwd = 1;
        a.op = OP_RESID;
        a.H = Hs;
        a.Y = Bs;
        a.acc = acc;
        a.frozen = frozen;
        SBTV_TRY(fft_cols_fwd_f(ctx, fp, x, nullptr, S, frozen));
        SBTV_TRY(fft_rows(ctx, fp, S, nullptr, a));
        double *tvp = nullptr;
        int ntv = 0;
        SBTV_TRY(tvnorm_partials(ctx, x, M, N, batch, &tvp, &ntv));
        hipLaunchKernelGGL(fista_collect_kernel, dim3(7, batch), dim3(256), 0, ctx->stream, (const double *)acc, nrb,
                           mom_partials, npb, (const double *)tvp, ntv, scal_hd, batch);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    };
    auto fetch = [&]() -> int {
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    };
    std::vector<double> obj_prev(batch, 0.0);
    std::vector<int> frozen(batch, 0), h_niter(batch, 1);
    SBTV_TRY(objective_of_x(nullptr, nullptr));
    SBTV_TRY(pair_sums(ctx, x, td, P, batch, o4));
    {
        std::vector<double> h4((size_t)batch * 4);
        SBTV_HIP(ctx, hipMemcpyAsync(h4.data(), o4, sizeof(double) * 4 * batch, hipMemcpyDeviceToHost, ctx->stream));
        SBTV_TRY(fetch());
        for (int b = 0; b < batch; ++b) {
            const double f0 = 0.5 * (scal_h[(size_t)b * 3] * parseval) + tau[b] * scal_h[6 * (size_t)batch + b];
            obj_prev[b] = f0;
            if (objective) objective[(size_t)b * maxiters] = f0;
            if (mses) mses[(size_t)b * maxiters] = h4[(size_t)b * 4] / (double)P;
        }
    }
    double t = 1.0;
    int active = batch;
    for (int k = 2; k <= maxiters && active > 0; ++k) {
        const double t_old = t;
        // y = y - (1/L) * AT(A(y) - b)                                   (:25)
        {
            RowsArgs a{};
            a.dir_fwd = 1;
            a.dir_inv = 1;
            a.op = OP_GRADF;
            a.H = Hs;
            a.Y = Bs;
            a.acc = acc;
            a.frozen = frozen_d;
            SBTV_TRY(fft_cols_fwd_f(ctx, fp, y, nullptr, S, frozen_d));
            SBTV_TRY(fft_rows(ctx, fp, S, S, a));
            SBTV_TRY(fft_cols_inv_f(ctx, fp, S, grad, inv_scale, frozen_d));
            // the gradient-step kernel also re-arms the control blocks of the cold-start prox that follows
            const ProxArm arm{pp.ctrl, lam_d, prox_iters, 1e-3, 0.249, frozen_d};
            if (batch <= 256) {
                SBTV_TRY(axpy(ctx, y, grad, 1.0 / L, cnt, &arm, batch));
            } else {
                SBTV_TRY(axpy(ctx, y, grad, 1.0 / L, cnt));
                SBTV_TRY(prox_reset(ctx, pp, lam_d, 1.0, prox_iters, 1e-3, 0.249, false, frozen_d));
            }
        }
        // x = Psi(y, tau/L): cold-start Chambolle                        (:26 ; run_moffat_demo.m:181-182)
        SBTV_TRY(prox_iterate(ctx, pp, y, prox_iters, x, true));
        t = 0.5 * (1 + sqrt(1 + 4 * t_old * t_old));                     // :28
        SBTV_TRY(fista_momentum(ctx, x, xold, y, td, (t_old - 1) / t, momp, P, batch, frozen_d));   // :29
        SBTV_TRY(objective_of_x(frozen_d, momp));
        SBTV_TRY(fetch());
        bool changed = false;
        for (int b = 0; b < batch; ++b) {
            if (frozen[b]) continue;
            const double f = 0.5 * (scal_h[(size_t)b * 3] * parseval) + tau[b] * scal_h[6 * (size_t)batch + b];
            const double *mom = scal_h + 3 * (size_t)batch + (size_t)b * 3;
            if (objective) objective[(size_t)b * maxiters + (k - 1)] = f;
            if (mses) mses[(size_t)b * maxiters + (k - 1)] = mom[0] / (double)P;
            h_niter[b] = k;
            double crit;
            if (stopcriterion == 1)
                crit = fabs(f - obj_prev[b]) / f;                        // :38 (divides by objective(k))
            else if (stopcriterion == 2)
                crit = sqrt(mom[1]) / sqrt(mom[2]);                      // :40
            else
                crit = f;                                                // :42
            obj_prev[b] = f;
            if (crit < tolerance) {                                      // :51
                frozen[b] = 1;
                --active;
                changed = true;
                SBTV_HIP(ctx, hipMemcpyAsync(xfinal + (size_t)b * P, x + (size_t)b * P, sizeof(double) * P,
                                             hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        if (changed && active > 0)
            SBTV_HIP(ctx, hipMemcpyAsync(frozen_d, frozen.data(), sizeof(int) * batch, hipMemcpyHostToDevice, ctx->stream));
        if (changed) SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    for (int b = 0; b < batch; ++b)
        if (!frozen[b])
            SBTV_HIP(ctx, hipMemcpyAsync(xfinal + (size_t)b * P, x + (size_t)b * P, sizeof(double) * P,
                                         hipMemcpyDeviceToDevice, ctx->stream));
    SBTV_TRY(stage_out_copy(ctx, x_out, xfinal, cnt, flags));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_iter)
        for (int b = 0; b < batch; ++b) n_iter[b] = h_niter[b];
    return canary_epilogue(ctx, 0);
}

// ---------------------------------------------------------------------------
// a-6: plain MYULA chain at fixed parameters (SALSA/myula.m:1-22)
// ---------------------------------------------------------------------------
int sbtv_myula(sbtv_ctx *ctx, const double *y, int M, int N, int batch, const double *taps, int taille, double lambda,
               double gamma, const double *theta, const double *sigma2, int samples, int chambolleit,
               unsigned long long seed, int chain_offset, const double *noise, double *x_out, int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    if (!y || !taps || !theta || !sigma2 || !x_out || batch < 1 || samples < 2 || !(lambda > 0.0) || !(gamma > 0.0) ||
        chain_offset < 0)
        return fail(ctx, SBTV_ERR_BADARG, "myula: bad arguments");
    if (chambolleit <= 0) return fail(ctx, SBTV_ERR_MAXITER, "myula: chambolleit must be positive");
    if (taille < 1 || taille > 15 || taille > M || taille > N) return fail(ctx, SBTV_ERR_PSF, "Mask does not fit inside array");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    if (((size_t)M * N) & 1)
        return fail(ctx, SBTV_ERR_SIZE, "this entry point needs an even number of pixels (its element-wise passes move two per lane)");
    FftPlan fp;
    SBTV_TRY(fft_plan(ctx, M, N, batch, &fp));
    ProxPlan pp;
    SBTV_TRY(prox_plan(ctx, M, N, batch, &pp));
    const size_t P = (size_t)M * N, cnt = P * batch, spec = fp.u_img;
    const double *yd = nullptr;
    SBTV_TRY(stage_in(ctx, "myula.y", y, cnt, flags, &yd));
    const bool noise_host = noise && !(flags & SBTV_DEVICE_PTRS);
    double *X = nullptr, *prox = nullptr, *grad = nullptr, *Z = nullptr, *par = nullptr, *acc = nullptr;
    double2 *S = nullptr, *Hs = nullptr, *Ys = nullptr;
    SBTV_TRY(stage_out_buf(ctx, "myula.X", x_out, cnt, flags, &X));
    SBTV_TRY(ws_get_t(ctx, "myula.prox", cnt, &prox));
    SBTV_TRY(ws_get_t(ctx, "myula.grad", cnt, &grad));
    if (noise_host) SBTV_TRY(ws_get_t(ctx, "myula.Z", cnt, &Z));
    SBTV_TRY(ws_get_t(ctx, "myula.S", (size_t)batch * fp.s_img, &S));
    SBTV_TRY(ws_get_t(ctx, "myula.H", spec * batch, &Hs));
    SBTV_TRY(ws_get_t(ctx, "myula.Y", spec * batch, &Ys));
    SBTV_TRY(ws_get_t(ctx, "myula.acc", (size_t)batch * 3 * fft_rows_blocks(fp), &acc));
    const size_t t2 = (size_t)taille * taille, npar = t2 * batch + 2 * (size_t)batch;
    SBTV_TRY(ws_get_t(ctx, "myula.par", npar, &par));      // [taps | lambda*theta | sigma2]
    double *taps_d = par, *lam_d = par + t2 * batch, *sig_d = lam_d + batch;
    std::vector<double> hpar(npar);
    for (size_t q = 0; q < t2 * batch; ++q) hpar[q] = taps[q];
    for (int b = 0; b < batch; ++b) {
        hpar[t2 * batch + b] = lambda * theta[b];           // proxG(x, lambda, theta)   (run_deblur_tv.m:126)
        hpar[t2 * batch + batch + b] = sigma2[b];
    }
    SBTV_HIP(ctx, hipMemcpyAsync(par, hpar.data(), sizeof(double) * npar, hipMemcpyHostToDevice, ctx->stream));
    SBTV_TRY(psf_spectrum(ctx, fp, taps_d, taille, Hs));
    {
        RowsArgs a{};
        a.dir_fwd = 1;
        SBTV_TRY(fft_cols_fwd(ctx, fp, yd, nullptr, S));
        SBTV_TRY(fft_rows(ctx, fp, S, S, a));
        SBTV_TRY(spec_unpack(ctx, fp, S, Ys));
    }
    SBTV_HIP(ctx, hipMemcpyAsync(X, yd, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));   // x = op.y  (:3,11)
    const double inv_scale = 1.0 / ((double)fp.n1 * N);
    SBTV_TRY(prox_reset(ctx, pp, lam_d, 1.0, chambolleit, 1e-3, 0.249, false, nullptr));
    const ProxArm arm{pp.ctrl, lam_d, chambolleit, 1e-3, 0.249, nullptr};
    for (int ii = 2; ii <= samples - 1; ++ii) {             // :13
        const size_t step = (size_t)(ii - 2);
        SBTV_TRY(prox_iterate(ctx, pp, X, chambolleit, prox, true));                         // :15
        RowsArgs a{};                                       // gradF = AT(A x - y) / sigma2   (run_deblur_tv.m:131)
        a.dir_fwd = 1;
        a.dir_inv = 1;
        a.op = OP_GRADF;
        a.H = Hs;
        a.Y = Ys;
        a.acc = acc;
        SBTV_TRY(fft_cols_fwd(ctx, fp, X, nullptr, S));
        SBTV_TRY(fft_rows(ctx, fp, S, S, a));
        SBTV_TRY(fft_cols_inv(ctx, fp, S, grad, inv_scale));
        const double *zd = nullptr;
        if (noise_host) {
            SBTV_HIP(ctx, hipMemcpyAsync(Z, noise + step * cnt, sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream));
            zd = Z;
        } else if (noise) {
            zd = noise + step * cnt;
        }
        const RngArgs r{seed, (unsigned)step, (unsigned)chain_offset, nullptr};
        SBTV_TRY(myula_plain_step(ctx, X, prox, grad, zd, sig_d, gamma, lambda, P, batch, &r, &arm));   // :16
    }
    SBTV_TRY(stage_out_copy(ctx, x_out, X, cnt, flags));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return canary_epilogue(ctx, 0);
}

// ---------------------------------------------------------------------------
// a-5 / a-6: SAPG with a MYULA kernel
// ---------------------------------------------------------------------------
int sbtv_SAPG_algorithm(sbtv_ctx *ctx, const double *y, int M, int N, int batch, const sbtv_sapg_opts *op,
                        const double *x0, const double *noise, double *thetas, double *ps, double *sigmas,
                        double *logpi, double *logpi_wu, double *gx, double *grads, double *eb, double *x_last,
                        sbtv_allreduce_fn reduce_fn, void *reduce_user, int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    if (!y || !op || batch < 1) return fail(ctx, SBTV_ERR_BADARG, "SAPG_algorithm: bad arguments");
    if (op->kind < 0 || op->kind > 2) return fail(ctx, SBTV_ERR_PSF, "SAPG_algorithm: unknown PSF kind");
    const int taille = op->psf_size;
    if (taille < 1 || taille > 15 || taille > M || taille > N) return fail(ctx, SBTV_ERR_PSF, "Mask does not fit inside array");
    if (op->samples < 2 || op->warmup < 0 || op->burnIn < 1 || op->burnIn > op->samples)
        return fail(ctx, SBTV_ERR_BADARG, "SAPG_algorithm: need samples >= 2, 1 <= burnIn <= samples");
    if (op->chambolleit <= 0) return fail(ctx, SBTV_ERR_MAXITER, "SAPG_algorithm: chambolleit must be positive");
    if (op->chain_offset < 0) return fail(ctx, SBTV_ERR_BADARG, "SAPG_algorithm: chain_offset must be >= 0");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    const int npar = (op->kind == SBTV_PSF_LAPLACE) ? 1 : 2;
    const int shared = op->share_gradients ? 1 : 0;
    const int nspec = shared ? 1 : batch;              // spectra sets (H, D1, D2, Y)
    if (((size_t)M * N) & 1)
        return fail(ctx, SBTV_ERR_SIZE, "this entry point needs an even number of pixels (its element-wise passes move two per lane)");
    FftPlan fp, fps;
    SBTV_TRY(fft_plan(ctx, M, N, batch, &fp));
    SBTV_TRY(fft_plan(ctx, M, N, nspec, &fps));
    ProxPlan pp;
    SBTV_TRY(prox_plan(ctx, M, N, batch, &pp));
    const size_t P = (size_t)M * N, cnt = P * batch;
    const double dimX = (double)P;
    const int samples = op->samples, warmup = op->warmup;
    const size_t nsteps_noise = (size_t)(warmup > 0 ? warmup - 1 : 0) + (size_t)(samples - 1);

    // ---- buffers
    const double *yd = nullptr, *x0d = nullptr;
    SBTV_TRY(stage_in(ctx, "sapg.y", y, shared ? P : cnt, flags, &yd));
    SBTV_TRY(stage_in(ctx, "sapg.x0", x0, cnt, flags, &x0d));
    const double *noise_d = nullptr;
    const bool noise_host = noise && !(flags & SBTV_DEVICE_PTRS);
    if (noise && !noise_host) noise_d = noise;
    double *X = nullptr, *prox = nullptr, *grad = nullptr, *Z = nullptr;
    SBTV_TRY(ws_get_t(ctx, "sapg.X", cnt, &X));
    SBTV_TRY(ws_get_t(ctx, "sapg.prox", cnt, &prox));
    SBTV_TRY(ws_get_t(ctx, "sapg.grad", cnt, &grad));
    SBTV_TRY(ws_get_t(ctx, "sapg.Z", cnt, &Z));
    double2 *S = nullptr, *Hs = nullptr, *D1s = nullptr, *D2s = nullptr, *Ys = nullptr, *S1 = nullptr;
    const size_t spec = fp.u_img;
    SBTV_TRY(ws_get_t(ctx, "sapg.S", (size_t)batch * fp.s_img, &S));
    SBTV_TRY(ws_get_t(ctx, "sapg.H", spec * nspec, &Hs));
    SBTV_TRY(ws_get_t(ctx, "sapg.D1", spec * nspec, &D1s));
    SBTV_TRY(ws_get_t(ctx, "sapg.D2", spec * nspec, &D2s));
    SBTV_TRY(ws_get_t(ctx, "sapg.Y", spec * nspec, &Ys));
    SBTV_TRY(ws_get_t(ctx, "sapg.S1", (size_t)nspec * fp.s_img, &S1));
    const size_t t2 = (size_t)taille * taille;
    double *par = nullptr;       // [taps | d0 | d1] per spectrum set, then lam[batch], sigma2[batch], noise step
    const size_t npar_all = 3 * t2 * nspec + 2 * (size_t)batch + 1;
    SBTV_TRY(ws_get_t(ctx, "sapg.par", npar_all, &par));
    double *taps_d = par, *d0_d = par + t2 * nspec, *d1_d = par + 2 * t2 * nspec, *lam_d = par + 3 * t2 * nspec,
           *sig_d = lam_d + batch, *step_d = sig_d + batch;
    const int nrb = fft_rows_blocks(fp);
    double *acc = nullptr;
    SBTV_TRY(ws_get_t(ctx, "sapg.acc", (size_t)batch * 3 * nrb, &acc));
    double *scal_h = nullptr;
    {
        void *pz = nullptr;
        SBTV_TRY(pinned_get(ctx, sizeof(double) * (4 * batch + 3 * t2 * nspec + 2 * batch + 1), &pz));
        scal_h = static_cast<double *>(pz);
    }
    double *scal_hd = nullptr;                        // the same pinned scalars as the device sees them
    {
        void *dp = nullptr;
        SBTV_HIP(ctx, hipHostGetDevicePointer(&dp, scal_h, 0));
        scal_hd = static_cast<double *>(dp);
    }
    double *par_h = scal_h + 4 * (size_t)batch;       // pinned staging for the per-iteration parameter upload
    const double inv_scale = 1.0 / ((double)fp.n1 * N), parseval = 1.0 / ((double)M * N);
    const double lamb = op->lambda, gam = op->gamma;

    // ---- chain state (host scalars)
    std::vector<double> theta(batch, op->th_init), sig2(batch, op->sigma2_init);
    std::vector<double> p0(batch, op->p_init[0]), p1(batch, npar > 1 ? op->p_init[1] : 0.0);
    std::vector<double> th_hist, s_hist, p_hist;      // running sums for the EB means
    std::vector<double> sum_th(batch, 0.0), sum_s(batch, 0.0), sum_p0(batch, 0.0), sum_p1(batch, 0.0);

    // upload taps/derivative taps for the current parameters and rebuild the spectra
    std::vector<double> last_p0(nspec, NAN), last_p1(nspec, NAN);
    auto refresh_spectra = [&]() -> int {
        bool dirty = false;
        for (int s = 0; s < nspec; ++s)
            if (!(p0[s] == last_p0[s]) || !(p1[s] == last_p1[s])) dirty = true;
        if (!dirty) return 0;
        for (int s = 0; s < nspec; ++s) {
            double pv[3] = {p0[s], (op->kind == SBTV_PSF_GAUSSIAN || op->kind == SBTV_PSF_MOFFAT) ? p1[s] : 0.0,
                            op->kind == SBTV_PSF_GAUSSIAN ? op->phi : 0.0};
            int rc = sbtv_psf_taps(op->kind, taille, pv, par_h + s * t2, par_h + t2 * nspec + s * t2,
                                   par_h + 2 * t2 * nspec + s * t2);
            if (rc != 0) return fail(ctx, rc, "SAPG_algorithm: PSF parameters out of range");
            last_p0[s] = p0[s];
            last_p1[s] = p1[s];
        }
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));     // par_h may still be in flight
        SBTV_HIP(ctx, hipMemcpyAsync(par, par_h, sizeof(double) * 3 * t2 * nspec, hipMemcpyHostToDevice, ctx->stream));
        SBTV_TRY(psf_spectrum(ctx, fps, taps_d, taille, Hs));
        SBTV_TRY(psf_spectrum(ctx, fps, d0_d, taille, D1s));
        if (npar > 1) SBTV_TRY(psf_spectrum(ctx, fps, d1_d, taille, D2s));
        else SBTV_TRY(psf_spectrum(ctx, fps, d0_d, taille, D2s));
        return 0;
    };
    auto upload_lam_sigma = [&](const std::vector<double> &th) -> int {
        double *stage = par_h + 3 * t2 * nspec;
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int b = 0; b < batch; ++b) {
            stage[b] = lamb * th[b];                  // proxG(x, theta): 'lambda', op.lambda*theta  (run_Gaussian_demo.m:191)
            stage[batch + b] = sig2[b];
        }
        SBTV_HIP(ctx, hipMemcpyAsync(lam_d, stage, sizeof(double) * 2 * batch, hipMemcpyHostToDevice, ctx->stream));
        return 0;
    };
    // spectral pass over X with the CURRENT spectra: accumulates ||AX-y||^2 and <dA_q X, AX-y>, and (if want_grad)
    // leaves grad = AT(AX - y) (unscaled by sigma^2)
    auto operator_pass = [&](bool want_grad) -> int {
        RowsArgs a{};
        a.dir_fwd = 1;
        a.dir_inv = want_grad ? 1 : 0;
        a.op = OP_GRAD;
        a.H = Hs;
        a.Y = Ys;
        a.D1 = D1s;
        a.D2 = D2s;
        a.acc = acc;
        a.shared_spec = shared;
        SBTV_TRY(fft_cols_fwd(ctx, fp, X, nullptr, S));
        SBTV_TRY(fft_rows(ctx, fp, S, want_grad ? S : nullptr, a));
        if (want_grad) SBTV_TRY(fft_cols_inv(ctx, fp, S, grad, inv_scale));
        return 0;
    };
    // TVnorm(X) partials + ONE collector launch that reduces them together with the accumulators of the last
    // operator pass straight into pinned host memory (no separate reductions, no copy kernel)
    auto collect_scalars = [&]() -> int {
        double *tvp = nullptr;
        int ntv = 0;
        SBTV_TRY(tvnorm_partials(ctx, X, M, N, batch, &tvp, &ntv));
        hipLaunchKernelGGL(sapg_collect_kernel, dim3(4, batch), dim3(256), 0, ctx->stream, (const double *)acc, nrb,
                           (const double *)tvp, ntv, scal_hd, batch);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    };
    auto fetch_scalars = [&]() -> int {
        SBTV_TRY(collect_scalars());
        SBTV_TRY(wait_stream(ctx));
        return 0;
    };
    size_t noise_step = 0;
    auto next_noise = [&]() -> int {          // injected noise only (the device generator runs inside the MYULA step)
        if (noise_d) {
            SBTV_HIP(ctx, hipMemcpyAsync(Z, noise_d + noise_step * cnt, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
        } else if (noise_host) {
            SBTV_HIP(ctx, hipMemcpyAsync(Z, noise + noise_step * cnt, sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream));
        }
        ++noise_step;
        return 0;
    };
    // X <- |X + gam (prox - X)/lamb - gam gradF + sqrt(2 gam) Z|  (:80-81,160-161).  With the device generator the
    // normals are drawn inside the step kernel (no Z array is written or read); injected noise goes through Z.
    // The step kernel also re-arms the prox control blocks for the cold-start prox that always follows it.
    const ProxArm arm{pp.ctrl, lam_d, op->chambolleit, 1e-3, 0.249, nullptr};
    auto myula = [&](bool in_graph) -> int {
        if (noise) {
            SBTV_TRY(next_noise());
            return myula_step(ctx, X, prox, grad, Z, sig_d, gam, lamb, P, batch, nullptr, &arm);
        }
        const RngArgs r{op->seed, (unsigned)noise_step, (unsigned)op->chain_offset, in_graph ? step_d : nullptr};
        if (!in_graph) ++noise_step;
        return myula_step(ctx, X, prox, grad, nullptr, sig_d, gam, lamb, P, batch, &r, &arm);
    };
    // prox = chambolle(X, lambda*theta, cold start); armed: the MYULA step before it has reset the control blocks
    auto do_prox = [&](bool armed) -> int {
        if (!armed) SBTV_TRY(prox_reset(ctx, pp, lam_d, 1.0, op->chambolleit, 1e-3, 0.249, false, nullptr));
        SBTV_TRY(prox_iterate(ctx, pp, X, op->chambolleit, prox, true));
        return 0;
    };

    // ---- hipGraph replay (small images: ~25 launches of a few microseconds per iteration make the loop
    // launch-bound).  The graph body reads every per-iteration value (taps, lambda*theta, sigma^2, noise step)
    // from the device parameter block, which its first node refreshes from the pinned staging block.
    const bool params_move = !(op->fix_p[0] && (npar < 2 || op->fix_p[1]));
    bool use_graph = (noise == nullptr) && graph_wanted(cnt);
    hipGraphExec_t g_warm = nullptr, g_main = nullptr;
    struct GraphGuard {
        hipGraphExec_t *a, *b;
        ~GraphGuard() {
            if (*a) (void)hipGraphExecDestroy(*a);
            if (*b) (void)hipGraphExecDestroy(*b);
        }
    } graph_guard{&g_warm, &g_main};
    auto graph_body = [&](bool main_loop) -> int {
        SBTV_HIP(ctx, hipMemcpyAsync(par, par_h, sizeof(double) * npar_all, hipMemcpyHostToDevice, ctx->stream));
        if (main_loop && params_move) {
            // spectra of the parameters moved by the previous iteration, then grad = AT(AX - y) with them
            SBTV_TRY(psf_spectrum(ctx, fps, taps_d, taille, Hs));
            SBTV_TRY(psf_spectrum(ctx, fps, d0_d, taille, D1s));
            SBTV_TRY(psf_spectrum(ctx, fps, npar > 1 ? d1_d : d0_d, taille, D2s));
            RowsArgs a{};
            a.dir_fwd = 1;
            a.dir_inv = 1;
            a.op = OP_GRADF;
            a.H = Hs;
            a.Y = Ys;
            a.acc = acc;
            a.shared_spec = shared;
            SBTV_TRY(fft_cols_fwd(ctx, fp, X, nullptr, S));
            SBTV_TRY(fft_rows(ctx, fp, S, S, a));
            SBTV_TRY(fft_cols_inv(ctx, fp, S, grad, inv_scale));
        }
        SBTV_TRY(myula(true));
        SBTV_TRY(do_prox(true));
        SBTV_TRY(operator_pass(main_loop ? !params_move : true));
        SBTV_TRY(collect_scalars());
        return 0;
    };
    // host side of one replayed iteration: stage the parameters, launch, wait for the scalars
    auto graph_iteration = [&](hipGraphExec_t *exec, bool main_loop, const std::vector<double> &th, bool *replayed) -> int {
        *replayed = false;
        if (!*exec) {
            if (graph_begin(ctx) != 0 || graph_end(ctx, graph_body(main_loop), exec) != 0) {
                *exec = nullptr;
                use_graph = false;                 // capture unavailable: the caller keeps launching eagerly
                return 0;
            }
        }
        if (main_loop && params_move) {
            for (int s = 0; s < nspec; ++s) {
                double pv[3] = {p0[s], (op->kind == SBTV_PSF_GAUSSIAN || op->kind == SBTV_PSF_MOFFAT) ? p1[s] : 0.0,
                                op->kind == SBTV_PSF_GAUSSIAN ? op->phi : 0.0};
                int rc = sbtv_psf_taps(op->kind, taille, pv, par_h + s * t2, par_h + t2 * nspec + s * t2,
                                       par_h + 2 * t2 * nspec + s * t2);
                if (rc != 0) return fail(ctx, rc, "SAPG_algorithm: PSF parameters out of range");
                last_p0[s] = p0[s];
                last_p1[s] = p1[s];
            }
        }
        double *stage = par_h + 3 * t2 * nspec;
        for (int b = 0; b < batch; ++b) {
            stage[b] = lamb * th[b];
            stage[batch + b] = sig2[b];
        }
        stage[2 * batch] = (double)noise_step;
        ++noise_step;
        SBTV_HIP(ctx, hipGraphLaunch(*exec, ctx->stream));
        SBTV_TRY(wait_stream(ctx));
        *replayed = true;
        return 0;
    };

    // ---- Y spectrum (one per spectrum set)
    {
        RowsArgs a{};
        a.dir_fwd = 1;
        SBTV_TRY(fft_cols_fwd(ctx, fps, yd, nullptr, S1));
        SBTV_TRY(fft_rows(ctx, fps, S1, S1, a));
        SBTV_TRY(spec_unpack(ctx, fps, S1, Ys));
    }
    // X0 = y by default (SAPG_algorithm_Guassian.m:10-12)
    if (x0d) {
        SBTV_HIP(ctx, hipMemcpyAsync(X, x0d, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        for (int b = 0; b < batch; ++b)
            SBTV_HIP(ctx, hipMemcpyAsync(X + (size_t)b * P, yd + (shared ? 0 : (size_t)b * P), sizeof(double) * P,
                                         hipMemcpyDeviceToDevice, ctx->stream));
    }

    // logPi = -||y-AX||^2/(2 sigma2) - theta*TVnorm(X)   (run_Gaussian_demo.m:171,195)
    auto log_pi = [&](int b, double th, double s2) -> double {
        const double resid2 = scal_h[(size_t)b * 3] * parseval;
        return -resid2 / (2 * s2) - th * scal_h[3 * (size_t)batch + b];
    };

    // =========================== warm-up (:66-93) ===========================
    SBTV_TRY(refresh_spectra());
    if (warmup > 0) {
        SBTV_TRY(upload_lam_sigma(theta));
        SBTV_TRY(do_prox(false));
        SBTV_TRY(operator_pass(true));                       // grad for the first step
        for (int ii = 2; ii <= warmup; ++ii) {
            bool replayed = false;
            if (use_graph && ii >= 3) SBTV_TRY(graph_iteration(&g_warm, false, theta, &replayed));
            if (!replayed) {
                SBTV_TRY(myula(false));                                                           // :80-81
                SBTV_TRY(do_prox(true));                                                               // :82
                SBTV_TRY(operator_pass(true));
                SBTV_TRY(fetch_scalars());
            }
            if (logpi_wu)
                for (int b = 0; b < batch; ++b) logpi_wu[(size_t)b * warmup + (ii - 1)] = log_pi(b, theta[b], sig2[b]);   // :85
        }
    } else {
        SBTV_TRY(upload_lam_sigma(theta));
        SBTV_TRY(operator_pass(true));
    }

    // =========================== SAPG loop (:98-248) ===========================
    // slot 0 of the traces (ii = 1)
    SBTV_TRY(fetch_scalars());
    for (int b = 0; b < batch; ++b) {
        if (thetas) thetas[(size_t)b * samples] = theta[b];
        if (sigmas) sigmas[(size_t)b * samples] = sig2[b];
        if (ps) {
            ps[((size_t)b * 2 + 0) * samples] = p0[b];
            ps[((size_t)b * 2 + 1) * samples] = p1[b];
        }
        if (logpi) logpi[(size_t)b * samples] = log_pi(b, theta[b], sig2[b]);   // :131
        if (op->burnIn == 1) { sum_th[b] += theta[b]; sum_s[b] += sig2[b]; sum_p0[b] += p0[b]; sum_p1[b] += p1[b]; }
    }
    SBTV_TRY(upload_lam_sigma(theta));
    SBTV_TRY(do_prox(false));                                      // proxGX = proxG(X, thetas(1))   (:134)
    // device work of SAPG iteration ii up to the scalars it needs on the host.  In the shared-gradient mode a failure
    // here must not return at once: the peer ranks are about to enter this iteration's all-reduce and would wait
    // for ever, so the status travels with the gradients (sixth reduced element) and every rank leaves together.
    auto iterate_device = [&](int ii) -> int {
        bool replayed = false;
        if (use_graph && ii >= 3) SBTV_TRY(graph_iteration(&g_main, true, theta, &replayed));
        // gradF(X, p(ii-1), sigma(ii-1)): `grad` already holds AT(AX-y) for the current spectra unless
        // the PSF parameters moved at the end of the previous iteration
        if (!replayed && params_move && ii > 2) {
            SBTV_TRY(refresh_spectra());
            RowsArgs a{};
            a.dir_fwd = 1;
            a.dir_inv = 1;
            a.op = OP_GRADF;
            a.H = Hs;
            a.Y = Ys;
            a.acc = acc;
            a.shared_spec = shared;
            SBTV_TRY(fft_cols_fwd(ctx, fp, X, nullptr, S));
            SBTV_TRY(fft_rows(ctx, fp, S, S, a));
            SBTV_TRY(fft_cols_inv(ctx, fp, S, grad, inv_scale));
        }
        if (!replayed) {
            SBTV_TRY(upload_lam_sigma(theta));                                                    // theta(ii-1), sigma(ii-1)
            SBTV_TRY(myula(false));                                                                // :160-161
            SBTV_TRY(do_prox(true));                                                                   // :162
            SBTV_TRY(operator_pass(!params_move));                                                 // G_w*, G_s, f  (:170-188)
            SBTV_TRY(fetch_scalars());                                                             // incl. g(X)  (:165)
        }
        return 0;
    };
    for (int ii = 2; ii <= samples; ++ii) {
        const int i0 = ii - 1;
        const int rc_dev = iterate_device(ii);
        if (rc_dev != 0 && !(shared && reduce_fn)) return rc_dev;
        const double delta = op->d_scale * (pow((double)ii, -op->d_exp) / dimX);              // :55
        // per-chain gradients
        std::vector<double> Gt(batch), Gp0(batch), Gp1(batch), Gs(batch);
        for (int b = 0; b < batch && rc_dev == 0; ++b) {
            const double resid2 = scal_h[(size_t)b * 3] * parseval;
            const double tv = scal_h[3 * (size_t)batch + b];
            Gt[b] = dimX / theta[b] - tv;                                                      // :165
            Gp0[b] = (scal_h[(size_t)b * 3 + 1] * parseval) / sig2[b];                         // :170
            Gp1[b] = (scal_h[(size_t)b * 3 + 2] * parseval) / sig2[b];                         // :179
            Gs[b] = resid2 / (2 * sig2[b] * sig2[b]) - dimX / (2 * sig2[b]);                   // :188
            if (logpi) logpi[(size_t)b * samples + i0] = log_pi(b, theta[b], sig2[b]);         // :207
            if (gx) gx[(size_t)b * samples + (i0 - 1)] = tv;                                   // :208
        }
        if (shared) {
            // all chains sample the same posterior: average their gradients (the reference's
            // `for jj=1:1 ... G = mean(g_*)`, SAPG_algorithm_moffat.m:158-173), across ranks too
            // [sum G_theta, sum G_p0, sum G_p1, sum G_sigma, chains, ranks that failed in this iteration]
            double buf[6] = {0, 0, 0, 0, (double)batch, rc_dev != 0 ? 1.0 : 0.0};
            for (int b = 0; b < batch && rc_dev == 0; ++b) { buf[0] += Gt[b]; buf[1] += Gp0[b]; buf[2] += Gp1[b]; buf[3] += Gs[b]; }
            if (reduce_fn) {
                int rc = reduce_fn(reduce_user, buf, 6);
                if (rc_dev != 0) return rc_dev;            // the local error, after the peers have been told
                if (rc != 0) return fail(ctx, SBTV_ERR_BADARG, "SAPG_algorithm: reduce_fn failed");
                if (buf[5] != 0.0)
                    return fail(ctx, SBTV_ERR_PEER, "SAPG_algorithm: a peer rank of the shared-gradient chains failed in this iteration");
            }
            for (int b = 0; b < batch; ++b) { Gt[b] = buf[0] / buf[4]; Gp0[b] = buf[1] / buf[4]; Gp1[b] = buf[2] / buf[4]; Gs[b] = buf[3] / buf[4]; }
        }
        for (int b = 0; b < batch; ++b) {
            const double th_new = fmin(fmax(theta[b] + op->c_theta * delta * Gt[b], op->min_th), op->max_th);   // :166-167
            double q0 = op->fix_p[0] ? op->p_true[0] : p0[b] - op->c_p[0] * delta * Gp0[b];                     // :171-176
            q0 = fmin(fmax(q0, op->p_min[0]), op->p_max[0]);
            double q1 = p1[b];
            if (npar > 1) {
                q1 = op->fix_p[1] ? op->p_true[1] : p1[b] - op->c_p[1] * delta * Gp1[b];                         // :180-185
                q1 = fmin(fmax(q1, op->p_min[1]), op->p_max[1]);
            }
            double s_new = op->fix_sigma ? op->sigma2_init : sig2[b] + op->c_sigma * delta * Gs[b];              // :189-194
            s_new = fmin(fmax(s_new, fmin(op->sigma2_min, op->sigma2_max)), fmax(op->sigma2_min, op->sigma2_max));
            if (grads) {
                grads[((size_t)b * 4 + 0) * samples + i0] = Gt[b];
                grads[((size_t)b * 4 + 1) * samples + i0] = Gp0[b];
                grads[((size_t)b * 4 + 2) * samples + i0] = Gp1[b];
                grads[((size_t)b * 4 + 3) * samples + i0] = Gs[b];
            }
            theta[b] = th_new;
            p0[b] = q0;
            p1[b] = q1;
            sig2[b] = s_new;
            if (thetas) thetas[(size_t)b * samples + i0] = th_new;
            if (sigmas) sigmas[(size_t)b * samples + i0] = s_new;
            if (ps) {
                ps[((size_t)b * 2 + 0) * samples + i0] = q0;
                ps[((size_t)b * 2 + 1) * samples + i0] = q1;
            }
            if (ii >= op->burnIn) { sum_th[b] += th_new; sum_s[b] += s_new; sum_p0[b] += q0; sum_p1[b] += q1; }
        }
    }
    // EB estimates: mean over burnIn..samples (:258-284)
    if (eb) {
        const double cntm = (double)(samples - op->burnIn + 1);
        for (int b = 0; b < batch; ++b) {
            eb[(size_t)b * 4 + 0] = sum_th[b] / cntm;
            eb[(size_t)b * 4 + 1] = sum_p0[b] / cntm;
            eb[(size_t)b * 4 + 2] = sum_p1[b] / cntm;
            eb[(size_t)b * 4 + 3] = sum_s[b] / cntm;
        }
    }
    if (x_last) {
        if (flags & SBTV_DEVICE_PTRS)
            SBTV_HIP(ctx, hipMemcpyAsync(x_last, X, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
        else
            SBTV_HIP(ctx, hipMemcpyAsync(x_last, X, sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    }
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)nsteps_noise;
    return canary_epilogue(ctx, 0);
}

}  // extern "C"
