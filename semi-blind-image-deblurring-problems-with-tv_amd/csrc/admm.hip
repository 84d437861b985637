// The two other ADMM front-ends of the SALSA toolbox that call the Chambolle TV prox, built from
// the same device kernels as SALSA_v2:
//   C-SALSA  (SALSA/CSALSA_v2.m:160-561)  min TV(x)  s.t. ||Ax - y|| <= epsilon
//   CoRAL    (SALSA/CoRAL_v2.m:2-476)     min 0.5||Ax-y||^2 + tau1 TV(x) + tau2 TV(x)   (two split copies)
// Neither is called by the reference's demos, so they are not fused to the last pass; what they share with the SALSA
// loop since round 3 is its stream discipline: the Chambolle launches of an outer iteration are OPTIMISTIC (all TViters
// iterations back to back, no stop-rule kernels, no redo pass; the host applies chambolle_prox_TV_stop.m:131 to the step
// sums it reads with the iteration's scalars and, should the rule ever fire before the last step, repeats the solve with
// exact launches: SBTV_ADMM_EXACT=1 forces those), and the host evaluates the outer stop rule one iteration late while
// the next iteration already runs (x is double-buffered, so the result of the stopping iteration is intact).  The
// images of a batch are solved one after another.
#include <chrono>
#include <cmath>

#include "sbtv_internal.h"

#pragma clang fp contract(off)

namespace sbtv {

namespace {

constexpr int AB = 256;

__device__ __forceinline__ double ad_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-reduce NQ running sums and store them as partials[q * gridDim.x + blockIdx.x]
template <int NQ>
__device__ __forceinline__ void ad_store_partials(double (&acc)[NQ], double *__restrict__ partials) {
    __shared__ double red[NQ * 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
        const double a = ad_wave_sum(acc[c]);
        if (lane == 0) red[c * 4 + w] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < NQ; ++c)
            partials[(size_t)c * gridDim.x + blockIdx.x] = (red[c * 4] + red[c * 4 + 1]) + (red[c * 4 + 2] + red[c * 4 + 3]);
    }
}

#define AD_LOOP(q, P) for (size_t q = (size_t)blockIdx.x * AB + threadIdx.x; q < (P) / 2; q += (size_t)gridDim.x * AB)
#define AD_LD(p, q) (*reinterpret_cast<const double2 *>((p) + 2 * (q)))
#define AD_ST(p, q, v) (*reinterpret_cast<double2 *>((p) + 2 * (q)) = (v))

// g = x - b
__global__ __launch_bounds__(AB) void ad_sub_kernel(const double *__restrict__ x, const double *__restrict__ b,
                                                     double *__restrict__ g, size_t P) {
    AD_LOOP(q, P) {
        const double2 xv = AD_LD(x, q), bv = AD_LD(b, q);
        AD_ST(g, q, make_double2(xv.x - bv.x, xv.y - bv.y));
    }
}

// ---- C-SALSA -----------------------------------------------------------------------------------
// w = mu2 * (y + v + bv)                                                   (CSALSA_v2.m:467)
__global__ __launch_bounds__(AB) void csalsa_w_kernel(const double *__restrict__ y, const double *__restrict__ v,
                                                       const double *__restrict__ bv, const double *__restrict__ par,
                                                       double *__restrict__ w, size_t P) {
    const double mu2 = par[1];
    AD_LOOP(q, P) {
        const double2 a = AD_LD(y, q), b = AD_LD(v, q), c = AD_LD(bv, q);
        AD_ST(w, q, make_double2(mu2 * ((a.x + b.x) + c.x), mu2 * ((a.y + b.y) + c.y)));
    }
}

// partials[nb] = sum (Ax - y - bv)^2                                        (:483-484)
__global__ __launch_bounds__(AB) void csalsa_ve_kernel(const double *__restrict__ Ax, const double *__restrict__ y,
                                                        const double *__restrict__ bv, double *__restrict__ partials,
                                                        size_t P) {
    double acc[1] = {0.0};
    AD_LOOP(q, P) {
        const double2 a = AD_LD(Ax, q), b = AD_LD(y, q), c = AD_LD(bv, q);
        const double e0 = (a.x - b.x) - c.x, e1 = (a.y - b.y) - c.y;
        acc[0] += e0 * e0 + e1 * e1;
    }
    ad_store_partials<1>(acc, partials);
}

// projection of ve on the epsilon ball, multiplier updates and the sums of the traces (:483-501,534)
//   partials [6][nb]: (Ax-y)^2, (Ax-y-v)^2, (x-u)^2, (x-true)^2, (x-xprev)^2, x^2
__global__ __launch_bounds__(AB) void csalsa_update_kernel(const double *__restrict__ Ax, const double *__restrict__ y,
                                                            const double *__restrict__ x, const double *__restrict__ u,
                                                            double *__restrict__ v, double *__restrict__ bv,
                                                            double *__restrict__ bu, const double *__restrict__ tru,
                                                            const double *__restrict__ xprev,
                                                            const double *__restrict__ nve2, double eps,
                                                            double *__restrict__ partials, size_t P) {
    const double n_ve = sqrt(nve2[0]);
    const bool inside = n_ve <= eps;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    AD_LOOP(q, P) {
        const double2 a = AD_LD(Ax, q), yy = AD_LD(y, q), c = AD_LD(bv, q);
        const double t0 = a.x - yy.x, t1 = a.y - yy.y;
        const double e0 = t0 - c.x, e1 = t1 - c.y;
        const double v0 = inside ? e0 : e0 / n_ve * eps, v1 = inside ? e1 : e1 / n_ve * eps;
        AD_ST(v, q, make_double2(v0, v1));
        const double r0 = t0 - v0, r1 = t1 - v1;
        AD_ST(bv, q, make_double2(c.x - r0, c.y - r1));
        const double2 xv = AD_LD(x, q), uv = AD_LD(u, q), bb = AD_LD(bu, q);
        const double d0 = xv.x - uv.x, d1 = xv.y - uv.y;
        AD_ST(bu, q, make_double2(bb.x - d0, bb.y - d1));
        acc[0] += t0 * t0 + t1 * t1;
        acc[1] += r0 * r0 + r1 * r1;
        acc[2] += d0 * d0 + d1 * d1;
        if (tru) {
            const double2 tv = AD_LD(tru, q);
            const double m0 = xv.x - tv.x, m1 = xv.y - tv.y;
            acc[3] += m0 * m0 + m1 * m1;
        }
        if (xprev) {
            const double2 xo = AD_LD(xprev, q);
            const double m0 = xv.x - xo.x, m1 = xv.y - xo.y;
            acc[4] += m0 * m0 + m1 * m1;
        }
        acc[5] += xv.x * xv.x + xv.y * xv.y;
    }
    ad_store_partials<6>(acc, partials);
}

// ---- C-SALSA with the constraint split kept as a spectrum (default; see csalsa_one) ------------------------------
// After the row pass: the three spectral sums of the iteration -> the projection factor s = min(1, eps / ||ve||), the
// coefficients of the next row pass and the two traces that depend on Ax.  st: [0] mu2, [1] c_e = mu2 (2s-1),
// [2] c_keep = 1-s, [3] sum |E|^2 of the previous iteration (unscaled), [4] 1 - s of the previous iteration.
//   out[0] = ||Ax-y||^2, out[1] = ||Ax-y-v||^2 = ||(1-s) E' - (1-s_prev) E||^2 from |E'|^2, |E|^2 and |E'-E|^2:
//   (a-c)(a A1 - c A1p) + a c A2 (no difference of large numbers once the iteration settles: a -> c, E' -> E)
__global__ __launch_bounds__(256) void csalsa_scal_kernel(const double *__restrict__ accp, int nrb, double *__restrict__ st,
                                                          double eps, double pv, double *__restrict__ out) {
    // the three sums of the row pass, partials [3][nrb], in the order of reduce_partials_kernel
    __shared__ double red[4 * 3], tot[3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nrb; i += 256) s += accp[(size_t)c * nrb + i];
        s = ad_wave_sum(s);
        if (lane == 0) red[c * 4 + w] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) tot[threadIdx.x] = (red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1]) + (red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3]);
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double A0 = tot[0], A1 = tot[1], A2 = tot[2];
    const double n_ve = sqrt(A1 * pv);
    const double sfac = (n_ve <= eps) ? 1.0 : eps / n_ve;
    const double a = 1.0 - sfac, c = st[4], A1p = st[3];
    const double d1 = ((a - c) * (a * A1 - c * A1p) + a * c * A2) * pv;
    out[0] = A0 * pv;
    out[1] = d1 > 0.0 ? d1 : 0.0;
    st[1] = st[0] * (2.0 * sfac - 1.0);
    st[2] = a;
    st[3] = A1;
    st[4] = a;
}

// after the prox: bu <- bu - (x - u) (:492) and the sums of the traces that live in the image domain
//   partials [5][nb]: (x-u)^2, (x-true)^2, (x-xprev)^2, x^2, periodic TV(x) (utils/TVnorm.m:2)
__global__ __launch_bounds__(AB) void csalsa_post_kernel(const double *__restrict__ x, const double *__restrict__ u,
                                                          double *__restrict__ bu, const double *__restrict__ tru,
                                                          const double *__restrict__ xprev, double *__restrict__ partials,
                                                          unsigned M, unsigned N, size_t P) {
    double acc[5] = {0, 0, 0, 0, 0};
    AD_LOOP(q, P) {
        const double2 xv = AD_LD(x, q), uv = AD_LD(u, q), bb = AD_LD(bu, q);
        const double d0 = xv.x - uv.x, d1 = xv.y - uv.y;
        AD_ST(bu, q, make_double2(bb.x - d0, bb.y - d1));
        acc[0] += d0 * d0 + d1 * d1;
        if (tru) {
            const double2 tv = AD_LD(tru, q);
            const double m0 = xv.x - tv.x, m1 = xv.y - tv.y;
            acc[1] += m0 * m0 + m1 * m1;
        }
        if (xprev) {
            const double2 xo = AD_LD(xprev, q);
            const double m0 = xv.x - xo.x, m1 = xv.y - xo.y;
            acc[2] += m0 * m0 + m1 * m1;
        }
        acc[3] += xv.x * xv.x + xv.y * xv.y;
        // element 2q = (i, j), i even (M is even on this path): left neighbours in column j-1, upper neighbour of row i
        const unsigned idx = (unsigned)(2 * q), j = idx / M, i = idx - j * M;
        const size_t lcol = (size_t)(j > 0 ? j - 1 : N - 1) * M + i;
        const double2 xl = *reinterpret_cast<const double2 *>(x + lcol);
        const double xu = x[(size_t)j * M + (i > 0 ? i - 1 : M - 1)];
        const double h0 = xv.x - xl.x, v0 = xv.x - xu, h1 = xv.y - xl.y, v1 = xv.y - xv.x;
        acc[4] += sqrt(h0 * h0 + v0 * v0) + sqrt(h1 * h1 + v1 * v1);
    }
    ad_store_partials<5>(acc, partials);
}

// ---- CoRAL -------------------------------------------------------------------------------------
// s = (mu1 (u+bu) + mu2 (v+bv)) / mu_ls                                     (CoRAL_v2.m:411, ATy enters spectrally)
// TVS: also the periodic TV norms of u and v (utils/TVnorm.m:2) for the objective (:425), partials [2][nb]: the two
// arrays are in flight here anyway (M even; element 2q = (i, j) with i even)
template <bool TVS>
__global__ __launch_bounds__(AB) void coral_s_kernel(const double *__restrict__ u, const double *__restrict__ bu,
                                                      const double *__restrict__ v, const double *__restrict__ bv,
                                                      double mu1, double mu2, double mu_ls, double *__restrict__ s,
                                                      double *__restrict__ partials, unsigned M, unsigned N, size_t P) {
    double acc[2] = {0, 0};
    AD_LOOP(q, P) {
        const double2 a = AD_LD(u, q), b = AD_LD(bu, q), c = AD_LD(v, q), d = AD_LD(bv, q);
        AD_ST(s, q, make_double2((mu1 * (a.x + b.x) + mu2 * (c.x + d.x)) / mu_ls,
                                 (mu1 * (a.y + b.y) + mu2 * (c.y + d.y)) / mu_ls));
        if constexpr (TVS) {
            const unsigned idx = (unsigned)(2 * q), j = idx / M, i = idx - j * M;
            const size_t lcol = (size_t)(j > 0 ? j - 1 : N - 1) * M + i, up = (size_t)j * M + (i > 0 ? i - 1 : M - 1);
            const double2 ul = *reinterpret_cast<const double2 *>(u + lcol), vl = *reinterpret_cast<const double2 *>(v + lcol);
            const double uu = u[up], vu = v[up];
            {
                const double h0 = a.x - ul.x, v0 = a.x - uu, h1 = a.y - ul.y, v1 = a.y - a.x;
                acc[0] += sqrt(h0 * h0 + v0 * v0) + sqrt(h1 * h1 + v1 * v1);
            }
            {
                const double h0 = c.x - vl.x, v0 = c.x - vu, h1 = c.y - vl.y, v1 = c.y - c.x;
                acc[1] += sqrt(h0 * h0 + v0 * v0) + sqrt(h1 * h1 + v1 * v1);
            }
        }
    }
    if constexpr (TVS) ad_store_partials<2>(acc, partials);
}

// bu += u - x ; bv += v - x ; g1 = x - bu ; g2 = x - bv  (:420-421 and the next prox inputs :401,406)
//   partials [7][nb]: (x-true)^2, (x-u)^2, (x-v)^2, x^2, u^2, v^2, (x-xprev)^2
__global__ __launch_bounds__(AB) void coral_post_kernel(const double *__restrict__ x, const double *__restrict__ xprev,
                                                         const double *__restrict__ u, const double *__restrict__ v,
                                                         double *__restrict__ bu, double *__restrict__ bv,
                                                         double *__restrict__ g1, double *__restrict__ g2,
                                                         const double *__restrict__ tru, double *__restrict__ partials,
                                                         size_t P) {
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    AD_LOOP(q, P) {
        const double2 xv = AD_LD(x, q), uv = AD_LD(u, q), vv = AD_LD(v, q);
        double2 b1 = AD_LD(bu, q), b2 = AD_LD(bv, q);
        b1.x = b1.x + (uv.x - xv.x);
        b1.y = b1.y + (uv.y - xv.y);
        b2.x = b2.x + (vv.x - xv.x);
        b2.y = b2.y + (vv.y - xv.y);
        AD_ST(bu, q, b1);
        AD_ST(bv, q, b2);
        AD_ST(g1, q, make_double2(xv.x - b1.x, xv.y - b1.y));
        AD_ST(g2, q, make_double2(xv.x - b2.x, xv.y - b2.y));
        if (tru) {
            const double2 tv = AD_LD(tru, q);
            const double m0 = xv.x - tv.x, m1 = xv.y - tv.y;
            acc[0] += m0 * m0 + m1 * m1;
        }
        {
            const double d0 = xv.x - uv.x, d1 = xv.y - uv.y;
            acc[1] += d0 * d0 + d1 * d1;
            const double e0 = xv.x - vv.x, e1 = xv.y - vv.y;
            acc[2] += e0 * e0 + e1 * e1;
        }
        acc[3] += xv.x * xv.x + xv.y * xv.y;
        acc[4] += uv.x * uv.x + uv.y * uv.y;
        acc[5] += vv.x * vv.x + vv.y * vv.y;
        if (xprev) {
            const double2 xo = AD_LD(xprev, q);
            const double m0 = xv.x - xo.x, m1 = xv.y - xo.y;
            acc[6] += m0 * m0 + m1 * m1;
        }
    }
    ad_store_partials<7>(acc, partials);
}

inline int ad_blocks(size_t P) {
    size_t nb = (P / 2 + AB - 1) / AB;
    if (nb > 1024) nb = 1024;
    return nb < 1 ? 1 : (int)nb;
}

// everything one single-image solve shares: plans, spectra of the PSF and of y
struct AdmmCommon {
    FftPlan fp;
    size_t P;
    double2 *S, *Hs, *Ys;
    double *taps_d;
    double inv_scale, parseval;
    bool spectral = false;
};

int admm_common(sbtv_ctx *ctx, int M, int N, const double *taps, int taille, const double *yd, AdmmCommon *c) {
    if (((size_t)M * N) & 1)
        return fail(ctx, SBTV_ERR_SIZE, "this entry point needs an even number of pixels (its element-wise passes move two per lane)");
    SBTV_TRY(fft_plan(ctx, M, N, 1, &c->fp));
    c->P = (size_t)M * N;
    SBTV_TRY(ws_get_t(ctx, "admm.S", c->fp.s_img, &c->S));
    SBTV_TRY(ws_get_t(ctx, "admm.H", c->fp.u_img, &c->Hs));
    SBTV_TRY(ws_get_t(ctx, "admm.Y", c->fp.u_img, &c->Ys));
    SBTV_TRY(ws_get_t(ctx, "admm.taps", (size_t)taille * taille, &c->taps_d));
    SBTV_HIP(ctx, hipMemcpyAsync(c->taps_d, taps, sizeof(double) * taille * taille, hipMemcpyHostToDevice, ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    c->inv_scale = 1.0 / ((double)c->fp.n1 * N);
    c->parseval = 1.0 / ((double)M * N);
    SBTV_TRY(psf_spectrum(ctx, c->fp, c->taps_d, taille, c->Hs));
    if (yd) {
        RowsArgs a{};
        a.dir_fwd = 1;
        a.op = OP_NONE;
        SBTV_TRY(fft_cols_fwd(ctx, c->fp, yd, nullptr, c->S));
        SBTV_TRY(fft_rows(ctx, c->fp, c->S, c->S, a));
        SBTV_TRY(spec_unpack(ctx, c->fp, c->S, c->Ys));
    }
    return 0;
}

// out = real(ifft2(op(H) .* fft2(in)))
int admm_apply(sbtv_ctx *ctx, const AdmmCommon &c, int op, const double *in, double *out) {
    RowsArgs a{};
    a.dir_fwd = 1;
    a.dir_inv = 1;
    a.op = op;
    a.H = c.Hs;
    SBTV_TRY(fft_cols_fwd(ctx, c.fp, in, nullptr, c.S));
    SBTV_TRY(fft_rows(ctx, c.fp, c.S, c.S, a));
    return fft_cols_inv(ctx, c.fp, c.S, out, c.inv_scale);
}

int admm_init_x(sbtv_ctx *ctx, const AdmmCommon &c, int initialization, const double *yd, const double *xi, double *x) {
    if (initialization == 0) {
        SBTV_HIP(ctx, hipMemsetAsync(x, 0, sizeof(double) * c.P, ctx->stream));   // AT(zeros) == 0
    } else if (initialization == 2) {
        SBTV_TRY(admm_apply(ctx, c, OP_MUL_HC, yd, x));                            // x = ATy
    } else {
        SBTV_HIP(ctx, hipMemcpyAsync(x, xi, sizeof(double) * c.P, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return 0;
}

int check_common(sbtv_ctx *ctx, const char *who, const double *y, const double *taps, const sbtv_salsa_opts *opts,
                 const double *x_init, int M, int N, int batch, int taille) {
    if (!y || !opts || batch < 1) return fail(ctx, SBTV_ERR_BADARG, std::string(who) + ": missing required argument");
    if (!taps) return fail(ctx, SBTV_ERR_MISSING_AT, "The function handle for transpose of A is missing");
    if (opts->stopcriterion < 1 || opts->stopcriterion > 3) return fail(ctx, SBTV_ERR_STOPCRITERION, "Unknown stopping criterion");
    if (opts->initialization != 0 && opts->initialization != 2 && opts->initialization != 33333)
        return fail(ctx, SBTV_ERR_INIT, "Unknown 'Initialization' option");
    if (opts->initialization == 33333 && !x_init) return fail(ctx, SBTV_ERR_INIT, "Initialization = array but x_init is NULL");
    if (opts->TViters <= 0) return fail(ctx, SBTV_ERR_MAXITER, std::string(who) + ": TViters must be positive");
    if (opts->maxiter < 1) return fail(ctx, SBTV_ERR_MAXITER, std::string(who) + ": maxiter must be positive");
    if (taille < 1 || taille > 15 || taille > M || taille > N) return fail(ctx, SBTV_ERR_PSF, "Mask does not fit inside array");
    return 0;
}

constexpr int ADMM_RESTART_EXACT = 12345;      // internal status: the optimistic prox met its stop rule early

inline bool admm_exact_forced() {
    static const bool on = [] {
        const char *e = getenv("SBTV_ADMM_EXACT");
        return e && e[0] == '1';
    }();
    return on;
}

// C-SALSA: the constraint split (v, bv) as ONE spectrum and a scalar (default) or as images (SBTV_CSALSA_SPECTRAL=0)
inline bool csalsa_spectral_wanted() {
    static const bool on = [] {
        const char *e = getenv("SBTV_CSALSA_SPECTRAL");
        return !(e && e[0] == '0');
    }();
    return on;
}

// One warm-started TV prox of an outer iteration.  Optimistic: `nlaunch` counts the optimistic launches made on this plan
// so far (the kernels read dual buffer cur ^ parity); the step sums of the K iterations go to stepsums[0..K).
inline int admm_prox(sbtv_ctx *ctx, const ProxPlan &pp, const double *g, int K, double *f, bool spec, long long *nlaunch,
                     double *stepsums, const double *lam_dev, double tol, double tau) {
    if (!spec) {
        SBTV_TRY(prox_reset(ctx, pp, lam_dev, 1.0, K, tol, tau, true, nullptr));
        return prox_iterate(ctx, pp, g, K, f);
    }
    SBTV_TRY(prox_iterate(ctx, pp, g, K, f, false, 1, (int)(*nlaunch & 1), nullptr));
    *nlaunch += prox_launches(pp, K);
    return reduce_partials(ctx, pp.partials, K, pp.fnblk, stepsums);
}
// cont = (k < MaxIter) & (err > tol) over the K step sums: did the rule stop before the last step?
inline bool admm_fired_early(const double *stepsums, int K, double tol) {
    for (int k = 1; k < K; ++k)
        if (!(sqrt(stepsums[k - 1]) > tol * SPEC_TOL_GUARD)) return true;
    return false;
}
// the step sums of launches that sum a subset of the pixels are lower bounds (ProxPlan::esub_off): are they still far from tol^2?
inline bool admm_sums_near_tol(const double *stepsums, int K, double tol) {
    for (int k = 1; k <= K; ++k)
        if (!(stepsums[k - 1] > ESUB_MARGIN * tol * tol)) return true;
    return false;
}

// Per-iteration scalars travel through two pinned slots; an event per slot tells the host (which runs one iteration
// ahead) when slot `outer & 1` holds the scalars of iteration `outer`.
struct AdmmSlots {
    hipEvent_t ev[2] = {nullptr, nullptr};
    ~AdmmSlots() {
        for (auto e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

// ------------------------------------------------------------------------------------------------
int csalsa_one(sbtv_ctx *ctx, const double *yd, int M, int N, const double *taps, int taille, double mu1, double mu2,
               double sigma, double epsilon, double delta, const sbtv_salsa_opts *opts, const double *td,
               const double *xi, double *x_out_dev, double *objective, double *distance1, double *distance2,
               double *criterion, double *times, double *mses, int *numA, int *numAt, int *n_outer, bool spec_wanted) {
    AdmmCommon c;
    {
        FftPlan chk;
        SBTV_TRY(fft_plan(ctx, M, N, 1, &chk));
        // spectral form of the constraint split (see `enqueue_spectral` below): needs fixed mu1 / mu2, an even M (pairs of
        // one column per lane) and the row kernels that carry OP_CSALSA
        const bool sp = delta == 1.0 && !admm_exact_forced() && csalsa_spectral_wanted() && fft_rows_csalsa_ok(chk) && !(M & 1) &&
                        (size_t)M * N < ((size_t)1 << 31);
        SBTV_TRY(admm_common(ctx, M, N, taps, taille, sp ? yd : nullptr, &c));
        c.spectral = sp;
    }
    const bool spectral = c.spectral;
    ProxPlan pp;
    SBTV_TRY(prox_plan(ctx, M, N, 1, &pp));
    const size_t P = c.P;
    const int nb = ad_blocks(P);
    double *xbuf[2], *u, *bu, *v, *bv, *w, *Ax, *g, *par, *partials, *sums, *acc;
    double2 *Ws;
    SBTV_TRY(ws_get_t(ctx, "admm.acc", (size_t)3 * fft_rows_blocks(c.fp), &acc));
    SBTV_TRY(ws_get_t(ctx, "admm.x0", P, &xbuf[0]));
    SBTV_TRY(ws_get_t(ctx, "admm.x1", P, &xbuf[1]));
    SBTV_TRY(ws_get_t(ctx, "admm.u", P, &u));
    SBTV_TRY(ws_get_t(ctx, "admm.bu", P, &bu));
    SBTV_TRY(ws_get_t(ctx, "admm.v", P, &v));
    SBTV_TRY(ws_get_t(ctx, "admm.bv", P, &bv));
    SBTV_TRY(ws_get_t(ctx, "admm.w", P, &w));
    SBTV_TRY(ws_get_t(ctx, "admm.Ax", P, &Ax));
    SBTV_TRY(ws_get_t(ctx, "admm.g", P, &g));
    SBTV_TRY(ws_get_t(ctx, "admm.par", (size_t)4, &par));               // mu1, mu2, 1/mu1
    SBTV_TRY(ws_get_t(ctx, "admm.partials", (size_t)8 * nb, &partials));
    SBTV_TRY(ws_get_t(ctx, "admm.sums", (size_t)64, &sums));            // [0..5] update sums, [6] TV(x), [8] n_ve^2, [16..] prox step sums
    SBTV_TRY(ws_get_t(ctx, "admm.W", c.fp.u_img, &Ws));
    double2 *Es = Ws;                                                   // spectral form: the state spectrum E (W is not formed)
    double *cst = nullptr, *rs = nullptr;
    SBTV_TRY(ws_get_t(ctx, "admm.cs", (size_t)8, &cst));
    SBTV_TRY(ws_get_t(ctx, "admm.rs", (size_t)4, &rs));
    double *hs = nullptr;
    {
        void *pz = nullptr;
        SBTV_TRY(pinned_get(ctx, sizeof(double) * 128, &pz));
        hs = static_cast<double *>(pz);
    }
    const int maxiter = opts->maxiter;
    if (!(epsilon != 0.0)) epsilon = sqrt((double)P + 8 * sqrt((double)P)) * sigma;              // :413
    auto upload_par = [&]() -> int {
        const double h[4] = {mu1, mu2, 1.0 / mu1, 0.0};
        SBTV_HIP(ctx, hipMemcpyAsync(par, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    };
    SBTV_TRY(upload_par());
    int h_numA = 0, h_numAt = 1;                                         // ATy (:301)
    ctx->calls += 2;                                                     // AT(y), invLS(ATy, mu1) (:301,310)
    double *x = xbuf[0];
    SBTV_TRY(admm_init_x(ctx, c, opts->initialization, yd, xi, x));
    if (opts->initialization == 0) ctx->calls += 1;
    if (opts->initialization == 2) h_numAt += 1;                         // :385
    SBTV_HIP(ctx, hipMemsetAsync(u, 0, sizeof(double) * P, ctx->stream));   // :404-408
    SBTV_HIP(ctx, hipMemsetAsync(bu, 0, sizeof(double) * P, ctx->stream));
    SBTV_HIP(ctx, hipMemsetAsync(v, 0, sizeof(double) * P, ctx->stream));
    SBTV_HIP(ctx, hipMemsetAsync(bv, 0, sizeof(double) * P, ctx->stream));
    SBTV_TRY(prox_zero_duals(ctx, pp));                                  // :440-441
    SBTV_TRY(prox_reset(ctx, pp, par + 2, 1.0, opts->TViters, opts->chambolle_tol, opts->chambolle_tau, false, nullptr));

    // state before the loop (:416-448): Ax, criterion(1), objective(1) = TV(x), mses(1), distances
    SBTV_TRY(admm_apply(ctx, c, OP_MUL_H, x, Ax));
    h_numA += 2;                                                         // :421,445
    ctx->calls += 2;
    {
        // v = 0, bv = 0, u = 0: the update kernel on scratch multipliers gives the three norms without side effects
        double *sv, *sbv, *sbu;
        SBTV_TRY(ws_get_t(ctx, "admm.scratch0", P, &sv));
        SBTV_TRY(ws_get_t(ctx, "admm.scratch1", P, &sbv));
        SBTV_TRY(ws_get_t(ctx, "admm.scratch2", P, &sbu));
        SBTV_HIP(ctx, hipMemsetAsync(sbv, 0, sizeof(double) * P, ctx->stream));
        SBTV_HIP(ctx, hipMemsetAsync(sbu, 0, sizeof(double) * P, ctx->stream));
        SBTV_HIP(ctx, hipMemsetAsync(sums, 0, sizeof(double) * 16, ctx->stream));   // n_ve^2 = 0 -> inside the ball, v = ve
        hipLaunchKernelGGL(csalsa_update_kernel, dim3(nb), dim3(AB), 0, ctx->stream, Ax, yd, x, u, sv, sbv, sbu, td,
                           (const double *)nullptr, sums + 8, 1.0, partials, P);
        SBTV_TRY(reduce_partials(ctx, partials, 6, nb, sums));
        SBTV_TRY(tvnorm_dev(ctx, x, M, N, 1, sums + 6));
        SBTV_HIP(ctx, hipMemcpyAsync(hs, sums, sizeof(double) * 8, hipMemcpyDeviceToHost, ctx->stream));
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // before the loop v = 0, so distance1(1) = ||Ax-y-v|| = ||Ax-y|| (:447); hs[1] is not that (v=ve there)
        if (criterion) criterion[0] = sqrt(hs[0]);
        if (distance1) distance1[0] = sqrt(hs[0]);
        if (distance2) distance2[0] = sqrt(hs[2]);                       // ||x - u||, u = 0 (:448)
        if (objective) objective[0] = hs[6];                             // phi(x) = TVnorm(x) (:423)
        if (mses && td) mses[0] = hs[3] / (double)P;                     // :436
        if (times) times[0] = 0.0;
    }
    double crit_prev = sqrt(hs[0]), obj_prev = hs[6];
    const auto t0 = std::chrono::steady_clock::now();
    SBTV_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    // optimistic prox launches need a fixed threshold (the control block is armed once) and the tile kernels
    const int K = opts->TViters;
    const bool spec = spec_wanted && delta == 1.0 && !(opts->speculate & 2) && !admm_exact_forced() && prox_spec_ok(pp, g, u, K);
    const int lag = (spec && (opts->speculate & 1)) ? 1 : 0;
    long long nlaunch = 0, prox_iters = 0;
    AdmmSlots slots;
    for (auto &e : slots.ev) SBTV_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    double *hslot[2] = {hs, hs + 64};
    const bool tv_fused = fft_cols_tv_ok(c.fp);
    const int ntv = tv_fused ? fft_cols_blocks(c.fp) : 0;
    double *tvp = nullptr;
    if (tv_fused) SBTV_TRY(ws_get_t(ctx, "admm.tvc", (size_t)ntv, &tvp));
    if (spectral) {
        // v = bv = 0 before the loop: E = 0, s_prev = 1
        const double h[8] = {mu2, mu2, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        SBTV_HIP(ctx, hipMemsetAsync(Es, 0, sizeof(double2) * c.fp.u_img, ctx->stream));
        SBTV_HIP(ctx, hipMemcpyAsync(cst, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // Spectral form of an outer iteration.  The projection on the epsilon ball scales ve = Ax - y - bv by ONE scalar
    // s = min(1, eps/||ve||) (:485-489), so v = s ve, bv' = bv - (Ax - y - v) = -(1 - s) ve, and the next right-hand side
    // y + v + bv' = y + (2s - 1) ve: the split (v, bv) is the spectrum E of ve and the scalar s.  One row pass then does
    // what took three transforms: W = mu2 (Y + (2 s_prev - 1) E), X = (conj(H) W + mu1 S)/(|H|^2 + mu1), T = H X - Y
    // (= fft2(Ax - y)), E' = T + (1 - s_prev) E, with ||Ax-y||^2, ||ve||^2 and ||E' - E||^2 by Parseval (the norms the
    // traces and the next s need).  Image-domain work left: u + bu into the forward pass, x - bu for the prox, and one
    // pass for bu, TV(x) and the sums over x.
    auto enqueue_spectral = [&](int outer) -> int {
        const int k = outer - 1;
        double *xn = xbuf[k & 1];
        const double *xprev = xbuf[(k & 1) ^ 1];
        RowsArgs a{};
        a.dir_fwd = 1;
        a.dir_inv = 1;
        a.op = OP_CSALSA;
        a.H = c.Hs;
        a.Y = c.Ys;
        a.E = Es;
        a.cs = cst;
        a.mu = par;                                                      // mu1
        a.acc = acc;
        SBTV_TRY(fft_cols_fwd(ctx, c.fp, u, bu, c.S));
        SBTV_TRY(fft_rows(ctx, c.fp, c.S, c.S, a));
        if (fft_cols_inv_step_ok(c.fp)) {
            SBTV_TRY(fft_cols_inv_sub(ctx, c.fp, c.S, xn, c.inv_scale, bu, g));       // x and the prox input x - bu
        } else {
            SBTV_TRY(fft_cols_inv(ctx, c.fp, c.S, xn, c.inv_scale));
            hipLaunchKernelGGL(ad_sub_kernel, dim3(nb), dim3(AB), 0, ctx->stream, xn, bu, g, P);
        }
        hipLaunchKernelGGL(csalsa_scal_kernel, dim3(1), dim3(256), 0, ctx->stream, acc, fft_rows_blocks(c.fp), cst, epsilon,
                           c.parseval, sums);
        RedJobs jb;
        if (!spec) {
            SBTV_TRY(admm_prox(ctx, pp, g, K, u, false, &nlaunch, sums + 16, par + 2, opts->chambolle_tol, opts->chambolle_tau));
        } else {
            // (admm_prox without its reduction: the step sums are reduced together with the sums of the last pass)
            SBTV_TRY(prox_iterate(ctx, pp, g, K, u, false, 1, (int)(nlaunch & 1), nullptr));
            nlaunch += prox_launches(pp, K);
            jb.add(pp.partials, K, pp.fnblk, sums + 16);
        }
        hipLaunchKernelGGL(csalsa_post_kernel, dim3(nb), dim3(AB), 0, ctx->stream, xn, u, bu, td,
                           (opts->stopcriterion == 2) ? xprev : (const double *)nullptr, partials, (unsigned)M, (unsigned)N, P);
        jb.add(partials, 5, nb, sums + 2);
        SBTV_TRY(reduce_jobs(ctx, jb));
        SBTV_HIP(ctx, hipGetLastError());
        double *hsl = hslot[outer & 1];
        SBTV_HIP(ctx, hipMemcpyAsync(hsl, sums, sizeof(double) * (16 + (spec ? K : 0)), hipMemcpyDeviceToHost, ctx->stream));
        if (!spec) SBTV_HIP(ctx, hipMemcpyAsync(hsl + 8, pp.ctrl, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SBTV_HIP(ctx, hipEventRecord(slots.ev[outer & 1], ctx->stream));
        return 0;
    };
    auto enqueue = [&](int outer) -> int {                               // :461
        if (spectral) return enqueue_spectral(outer);
        const int k = outer - 1;
        double *xn = xbuf[k & 1];
        const double *xprev = xbuf[(k & 1) ^ 1];
        // r = mu1 (u+bu) + mu2 AT(y+v+bv) ; x = invLS(r, mu1)   (:467-471), all in one spectral pass:
        //   X = (conj(H) W + mu1 S) / (|H|^2 + mu1),  W = fft2(mu2 (y+v+bv)),  S = fft2(u+bu)
        hipLaunchKernelGGL(csalsa_w_kernel, dim3(nb), dim3(AB), 0, ctx->stream, yd, v, bv, par, w, P);
        {
            RowsArgs a{};
            a.dir_fwd = 1;
            a.op = OP_NONE;
            SBTV_TRY(fft_cols_fwd(ctx, c.fp, w, nullptr, c.S));
            SBTV_TRY(fft_rows(ctx, c.fp, c.S, c.S, a));
            SBTV_TRY(spec_unpack(ctx, c.fp, c.S, Ws));
        }
        {
            RowsArgs a{};
            a.dir_fwd = 1;
            a.dir_inv = 1;
            a.op = OP_SALSA;
            a.H = c.Hs;
            a.Y = Ws;
            a.mu = par;                                                  // mu1
            a.acc = acc;                                                 // residual sum of OP_SALSA, not used here
            SBTV_TRY(fft_cols_fwd(ctx, c.fp, u, bu, c.S));
            SBTV_TRY(fft_rows(ctx, c.fp, c.S, c.S, a));
            SBTV_TRY(fft_cols_inv(ctx, c.fp, c.S, xn, c.inv_scale));
        }
        // u = prox_{TV/mu1}(x - bu), warm-started duals (:476)
        hipLaunchKernelGGL(ad_sub_kernel, dim3(nb), dim3(AB), 0, ctx->stream, xn, bu, g, P);
        SBTV_TRY(admm_prox(ctx, pp, g, K, u, spec, &nlaunch, sums + 16, par + 2, opts->chambolle_tol, opts->chambolle_tau));
        // Ax (the forward column pass of x also leaves the periodic-TV partials of phi(x) = TVnorm(x), :498), projection on
        // the epsilon ball, multipliers, traces (:481-501)
        {
            RowsArgs a{};
            a.dir_fwd = 1;
            a.dir_inv = 1;
            a.op = OP_MUL_H;
            a.H = c.Hs;
            SBTV_TRY(fft_cols_fwd_f(ctx, c.fp, xn, nullptr, c.S, nullptr, tvp));
            SBTV_TRY(fft_rows(ctx, c.fp, c.S, c.S, a));
            SBTV_TRY(fft_cols_inv(ctx, c.fp, c.S, Ax, c.inv_scale));
        }
        hipLaunchKernelGGL(csalsa_ve_kernel, dim3(nb), dim3(AB), 0, ctx->stream, Ax, yd, bv, partials, P);
        SBTV_TRY(reduce_partials(ctx, partials, 1, nb, sums + 8));
        hipLaunchKernelGGL(csalsa_update_kernel, dim3(nb), dim3(AB), 0, ctx->stream, Ax, yd, xn, u, v, bv, bu, td,
                           (opts->stopcriterion == 2) ? xprev : (const double *)nullptr, sums + 8, epsilon, partials, P);
        SBTV_TRY(reduce_partials(ctx, partials, 6, nb, sums));
        if (tv_fused) SBTV_TRY(reduce_partials(ctx, tvp, 1, ntv, sums + 6));
        else SBTV_TRY(tvnorm_dev(ctx, xn, M, N, 1, sums + 6));
        SBTV_HIP(ctx, hipGetLastError());
        double *hsl = hslot[outer & 1];
        SBTV_HIP(ctx, hipMemcpyAsync(hsl, sums, sizeof(double) * (16 + (spec ? K : 0)), hipMemcpyDeviceToHost, ctx->stream));
        if (!spec) SBTV_HIP(ctx, hipMemcpyAsync(hsl + 8, pp.ctrl, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SBTV_HIP(ctx, hipEventRecord(slots.ev[outer & 1], ctx->stream));
        return 0;
    };
    bool stop = false, fired = false;
    // host side of iteration `outer`: traces (:494-501) and the stop rule (:517-541)
    auto process = [&](int outer) -> int {
        const int k = outer - 1;
        const double *hsl = hslot[outer & 1];
        SBTV_HIP(ctx, hipEventSynchronize(slots.ev[outer & 1]));
        if (spec && admm_fired_early(hsl + 16, K, opts->chambolle_tol)) {
            fired = true;
            return 0;
        }
        if (spec && !pp.esub_off && admm_sums_near_tol(hsl + 16, K, opts->chambolle_tol)) {
            pp.esub_off = 1;
            ctx->solve_stats[1] += 1;
        }
        prox_iters += spec ? K : *reinterpret_cast<const int *>(hsl + 8);
        h_numAt += 1;
        h_numA += 1;
        ctx->calls += 3;                                                 // AT, invLS, A
        const double crit = sqrt(hsl[0]), obj = hsl[6];
        if (criterion) criterion[k] = crit;                              // :494
        if (distance1) distance1[k] = sqrt(hsl[1]);                      // :495
        if (distance2) distance2[k] = sqrt(hsl[2]);                      // :497
        if (objective) objective[k] = obj;                               // :498
        if (mses && td) mses[k] = hsl[3] / (double)P;                    // :501
        if (times) times[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (delta != 1.0) {                                              // :517-518
            mu1 *= delta;
            mu2 *= delta;
            SBTV_TRY(upload_par());
        }
        double sc;
        if (opts->stopcriterion == 1)
            sc = fabs(obj - obj_prev) / obj;                             // :527
        else if (opts->stopcriterion == 2)
            sc = fabs(sqrt(hsl[4]) / sqrt(hsl[5]));                      // :534
        else
            sc = fabs(crit - crit_prev) / crit;                          // :539
        obj_prev = obj;
        crit_prev = crit;
        stop = (sc < opts->tolA && crit <= epsilon);                     // :529,535,541
        return 0;
    };
    int last = 1, enq = 1, done = 1;
    while (!stop && done < maxiter) {
        while (enq < maxiter && enq - done <= lag) SBTV_TRY(enqueue(++enq));
        SBTV_TRY(process(++done));
        if (fired) {
            SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return ADMM_RESTART_EXACT;
        }
        last = done;
    }
    SBTV_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    {
        float ms = 0.f;
        SBTV_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ctx->timing[0] = ms;
        ctx->timing[1] = 0.0;
        ctx->timing[2] = (double)prox_iters;
        ctx->timing[3] = 40.0 * (double)P * (double)prox_iters;
    }
    SBTV_HIP(ctx, hipMemcpyAsync(x_out_dev, xbuf[(last - 1) & 1], sizeof(double) * P, hipMemcpyDeviceToDevice, ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (numA) *numA = h_numA;
    if (numAt) *numAt = h_numAt;
    if (n_outer) *n_outer = last;
    return 0;
}

// ------------------------------------------------------------------------------------------------
int coral_one(sbtv_ctx *ctx, const double *yd, int M, int N, const double *taps, int taille, double tau1, double tau2,
              double mu1, double mu2, double mu_ls, int TViters2, const sbtv_salsa_opts *opts, const double *td,
              const double *xi, double *x_out_dev, double *objective, double *distance, double *times, double *mses,
              int *numA, int *numAt, int *n_outer, bool spec_wanted) {
    AdmmCommon c;
    SBTV_TRY(admm_common(ctx, M, N, taps, taille, yd, &c));
    // The two TV proxes of an iteration are independent (:401,406).  With the same iteration count they run as ONE batch of
    // two images (own threshold, own duals, own stop rule per image: the batch semantics of the prox): two Chambolle
    // launches with twice the tiles per outer iteration instead of four (SBTV_CORAL_BATCH=0: one plan per prox)
    static const bool env_batch = [] {
        const char *e = getenv("SBTV_CORAL_BATCH");
        return !(e && e[0] == '0');
    }();
    const bool batched = env_batch && opts->TViters == TViters2;
    ProxPlan pu, pv;
    SBTV_TRY(prox_plan(ctx, M, N, batched ? 2 : 1, &pu, batched ? "prox.pair" : "prox"));
    if (!batched) SBTV_TRY(prox_plan(ctx, M, N, 1, &pv, "prox2"));
    const size_t P = c.P;
    const int nb = ad_blocks(P), nrb = fft_rows_blocks(c.fp);
    const bool tv_in_s = !(M & 1) && P < ((size_t)1 << 31);             // TV(u), TV(v) ride in the pass that forms s
    double *xbuf[2], *u, *bu, *v, *bv, *s, *g1, *g2, *par, *partials, *sums, *acc, *tvpart;
    SBTV_TRY(ws_get_t(ctx, "admm.x0", P, &xbuf[0]));
    SBTV_TRY(ws_get_t(ctx, "admm.x1", P, &xbuf[1]));
    SBTV_TRY(ws_get_t(ctx, "admm.uv", 2 * P, &u));                      // [u | v] and [g1 | g2]: the images of the batch
    v = u + P;
    SBTV_TRY(ws_get_t(ctx, "admm.bu", P, &bu));
    SBTV_TRY(ws_get_t(ctx, "admm.bv", P, &bv));
    SBTV_TRY(ws_get_t(ctx, "admm.w", P, &s));
    SBTV_TRY(ws_get_t(ctx, "admm.g12", 2 * P, &g1));
    g2 = g1 + P;
    SBTV_TRY(ws_get_t(ctx, "admm.tvpart", (size_t)2 * nb, &tvpart));
    SBTV_TRY(ws_get_t(ctx, "admm.par", (size_t)4, &par));               // mu_ls, thr1, thr2
    SBTV_TRY(ws_get_t(ctx, "admm.partials", (size_t)8 * nb, &partials));
    SBTV_TRY(ws_get_t(ctx, "admm.sums", (size_t)96, &sums));            // [0..6] post sums, [8] resid2, [9] TV(u), [10] TV(v), [16..], [48..] step sums
    SBTV_TRY(ws_get_t(ctx, "admm.acc", (size_t)3 * nrb, &acc));
    double *hs = nullptr;
    {
        void *pz = nullptr;
        SBTV_TRY(pinned_get(ctx, sizeof(double) * 192, &pz));
        hs = static_cast<double *>(pz);
    }
    {
        const double h[4] = {mu_ls, tau1 / mu1, tau2 / mu2, 0.0};        // thresholds :356,361
        SBTV_HIP(ctx, hipMemcpyAsync(par, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const int maxiter = opts->maxiter;
    int h_numA = 0, h_numAt = 2;                                         // ATy twice (:189,219)
    ctx->calls += 3;                                                     // AT(y), invLS(ATy), AT(y)
    double *x = xbuf[0];
    SBTV_TRY(admm_init_x(ctx, c, opts->initialization, yd, xi, x));
    if (opts->initialization == 0) ctx->calls += 1;
    // u = x ; bu = u ; v = x ; bv = v  (:353-359)  ->  first prox inputs x - bu = x - bv = 0
    for (double *dst : {u, bu, v, bv})
        SBTV_HIP(ctx, hipMemcpyAsync(dst, x, sizeof(double) * P, hipMemcpyDeviceToDevice, ctx->stream));
    SBTV_HIP(ctx, hipMemsetAsync(g1, 0, sizeof(double) * P, ctx->stream));
    SBTV_HIP(ctx, hipMemsetAsync(g2, 0, sizeof(double) * P, ctx->stream));
    SBTV_TRY(prox_zero_duals(ctx, pu));                                  // :384-392
    SBTV_TRY(prox_reset(ctx, pu, par + 1, 1.0, opts->TViters, opts->chambolle_tol, opts->chambolle_tau, false, nullptr));
    if (!batched) {
        SBTV_TRY(prox_zero_duals(ctx, pv));
        SBTV_TRY(prox_reset(ctx, pv, par + 2, 1.0, TViters2, opts->chambolle_tol, opts->chambolle_tau, false, nullptr));
    }

    // initial objective (:366-368)
    {
        RowsArgs a{};
        a.dir_fwd = 1;
        a.op = OP_RESID;
        a.H = c.Hs;
        a.Y = c.Ys;
        a.acc = acc;
        SBTV_TRY(fft_cols_fwd(ctx, c.fp, x, nullptr, c.S));
        SBTV_TRY(fft_rows(ctx, c.fp, c.S, nullptr, a));
        SBTV_TRY(reduce_partials(ctx, acc, 1, nrb, sums + 8));
        SBTV_TRY(tvnorm_dev(ctx, u, M, N, 1, sums + 9));
        SBTV_HIP(ctx, hipMemsetAsync(sums, 0, sizeof(double) * 8, ctx->stream));
        if (td) {
            double *o4 = nullptr;
            SBTV_TRY(ws_get_t(ctx, "admm.o4", (size_t)4, &o4));
            SBTV_TRY(pair_sums(ctx, x, td, P, 1, o4));
            SBTV_HIP(ctx, hipMemcpyAsync(sums, o4, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        }
        SBTV_HIP(ctx, hipMemcpyAsync(hs, sums, sizeof(double) * 16, hipMemcpyDeviceToHost, ctx->stream));
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        h_numA += 1;
        ctx->calls += 1;
        const double f0 = 0.5 * (hs[8] * c.parseval) + tau1 * hs[9] + tau2 * hs[9];   // u = v = x
        if (objective) objective[0] = f0;
        if (times) times[0] = 0.0;
        if (mses && td) mses[0] = hs[0] / (double)P;                     // :381
        hs[15] = f0;
    }
    double obj_prev = hs[15];
    const auto t0 = std::chrono::steady_clock::now();
    SBTV_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    const int K1 = opts->TViters, K2 = TViters2;
    const bool spec = spec_wanted && !(opts->speculate & 2) && !admm_exact_forced() && prox_spec_ok(pu, g1, u, K1) &&
                      (batched || prox_spec_ok(pv, g2, v, K2));
    const int lag = (spec && (opts->speculate & 1)) ? 1 : 0;
    long long nl_u = 0, nl_v = 0, prox_iters = 0;
    AdmmSlots slots;
    for (auto &e : slots.ev) SBTV_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    double *hslot[2] = {hs, hs + 96};
    auto enqueue = [&](int outer) -> int {                               // :394
        double *xn = xbuf[outer & 1];
        const double *xprev = xbuf[(outer & 1) ^ 1];
        // the two TV proxes with their own warm-started duals (:401,406).  The first outer iteration always runs exactly:
        // its prox inputs x - bu = x - bv are zero and the rule stops at k = 1; its control blocks are re-armed afterwards
        const bool sp = spec && outer >= 2;
        RedJobs jb;
        if (spec && outer == 2) {
            SBTV_TRY(prox_reset(ctx, pu, par + 1, 1.0, K1, opts->chambolle_tol, opts->chambolle_tau, true, nullptr));
            if (!batched) SBTV_TRY(prox_reset(ctx, pv, par + 2, 1.0, K2, opts->chambolle_tol, opts->chambolle_tau, true, nullptr));
        }
        if (batched) {
            // both images in one plan: the step sums of image b land at sums + 16 + b FSTRIDE (= sums + 48 for v)
            static_assert(FSTRIDE == 32, "the step sums of the second prox are read at sums + 48");
            if (!sp) {
                SBTV_TRY(prox_reset(ctx, pu, par + 1, 1.0, K1, opts->chambolle_tol, opts->chambolle_tau, true, nullptr));
                SBTV_TRY(prox_iterate(ctx, pu, g1, K1, u));
            } else {
                SBTV_TRY(prox_iterate(ctx, pu, g1, K1, u, false, 1, (int)(nl_u & 1), nullptr));
                nl_u += prox_launches(pu, K1);
                jb.add(pu.partials, FSTRIDE + K1, pu.fnblk, sums + 16);      // reduced at the end of the iteration
            }
        } else {
            SBTV_TRY(admm_prox(ctx, pu, g1, K1, u, sp, &nl_u, sums + 16, par + 1, opts->chambolle_tol, opts->chambolle_tau));
            SBTV_TRY(admm_prox(ctx, pv, g2, K2, v, sp, &nl_v, sums + 48, par + 2, opts->chambolle_tol, opts->chambolle_tau));
        }
        // r = ATy + mu1 (u+bu) + mu2 (v+bv) ; x = invLS(r)   (:411-413)  + residual energy (:423, Parseval)
        if (tv_in_s) {
            hipLaunchKernelGGL(coral_s_kernel<true>, dim3(nb), dim3(AB), 0, ctx->stream, u, bu, v, bv, mu1, mu2, mu_ls, s, tvpart,
                               (unsigned)M, (unsigned)N, P);
            jb.add(tvpart, 2, nb, sums + 9);
        } else {
            hipLaunchKernelGGL(coral_s_kernel<false>, dim3(nb), dim3(AB), 0, ctx->stream, u, bu, v, bv, mu1, mu2, mu_ls, s, tvpart,
                               (unsigned)M, (unsigned)N, P);
        }
        {
            RowsArgs a{};
            a.dir_fwd = 1;
            a.dir_inv = 1;
            a.op = OP_SALSA;
            a.H = c.Hs;
            a.Y = c.Ys;
            a.mu = par;                                                  // mu_ls
            a.acc = acc;
            SBTV_TRY(fft_cols_fwd(ctx, c.fp, s, nullptr, c.S));
            SBTV_TRY(fft_rows(ctx, c.fp, c.S, c.S, a));
            SBTV_TRY(fft_cols_inv(ctx, c.fp, c.S, xn, c.inv_scale));
        }
        hipLaunchKernelGGL(coral_post_kernel, dim3(nb), dim3(AB), 0, ctx->stream, xn,
                           (opts->stopcriterion == 2) ? xprev : (const double *)nullptr, u, v, bu, bv, g1, g2, td, partials, P);
        jb.add(partials, 7, nb, sums);
        jb.add(acc, 1, nrb, sums + 8);
        SBTV_TRY(reduce_jobs(ctx, jb));
        if (!tv_in_s) {
            SBTV_TRY(tvnorm_dev(ctx, u, M, N, 1, sums + 9));
            SBTV_TRY(tvnorm_dev(ctx, v, M, N, 1, sums + 10));
        }
        SBTV_HIP(ctx, hipGetLastError());
        double *hsl = hslot[outer & 1];
        SBTV_HIP(ctx, hipMemcpyAsync(hsl, sums, sizeof(double) * (sp ? 80 : 12), hipMemcpyDeviceToHost, ctx->stream));
        if (!sp) {
            SBTV_HIP(ctx, hipMemcpyAsync(hsl + 12, pu.ctrl, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            SBTV_HIP(ctx, hipMemcpyAsync(hsl + 13, batched ? pu.ctrl + 1 : pv.ctrl, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        }
        SBTV_HIP(ctx, hipEventRecord(slots.ev[outer & 1], ctx->stream));
        return 0;
    };
    bool stop = false, fired = false;
    auto process = [&](int outer) -> int {
        const double *hsl = hslot[outer & 1];
        const bool sp = spec && outer >= 2;
        SBTV_HIP(ctx, hipEventSynchronize(slots.ev[outer & 1]));
        if (sp && (admm_fired_early(hsl + 16, K1, opts->chambolle_tol) || admm_fired_early(hsl + 48, K2, opts->chambolle_tol))) {
            fired = true;
            return 0;
        }
        if (sp && (admm_sums_near_tol(hsl + 16, K1, opts->chambolle_tol) || admm_sums_near_tol(hsl + 48, K2, opts->chambolle_tol)))
            pu.esub_off = pv.esub_off = 1;
        prox_iters += sp ? (long long)(K1 + K2)
                         : (long long)(*reinterpret_cast<const int *>(hsl + 12) + *reinterpret_cast<const int *>(hsl + 13));
        h_numA += 1;
        ctx->calls += 2;                                                 // invLS, A
        const double f = 0.5 * (hsl[8] * c.parseval) + tau1 * hsl[9] + tau2 * hsl[10];      // :425
        if (objective) objective[outer] = f;
        if (mses && td) mses[outer] = hsl[0] / (double)P;                // :428-429
        if (distance) {
            distance[(size_t)(outer - 1) * 2] = sqrt(hsl[1]) / sqrt(hsl[3] + hsl[4]);       // :432
            distance[(size_t)(outer - 1) * 2 + 1] = sqrt(hsl[2]) / sqrt(hsl[3] + hsl[5]);   // :433
        }
        if (times) times[outer] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (outer > 1) {                                                 // :435
            double crit;
            if (opts->stopcriterion == 1)
                crit = fabs(f - obj_prev) / obj_prev;                    // :441
            else if (opts->stopcriterion == 2)
                crit = fabs(sqrt(hsl[6]) / sqrt(hsl[3]));                // :445
            else
                crit = f;                                                // :448
            stop = crit < opts->tolA;                                    // :453
        }
        obj_prev = f;
        return 0;
    };
    int last = 0, enq = 0, done = 0;
    while (!stop && done < maxiter) {
        while (enq < maxiter && enq - done <= lag) SBTV_TRY(enqueue(++enq));
        SBTV_TRY(process(++done));
        if (fired) {
            SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return ADMM_RESTART_EXACT;
        }
        last = done;
    }
    SBTV_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    {
        float ms = 0.f;
        SBTV_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ctx->timing[0] = ms;
        ctx->timing[1] = 0.0;
        ctx->timing[2] = (double)prox_iters;
        ctx->timing[3] = 40.0 * (double)P * (double)prox_iters;
    }
    SBTV_HIP(ctx, hipMemcpyAsync(x_out_dev, xbuf[last & 1], sizeof(double) * P, hipMemcpyDeviceToDevice, ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (numA) *numA = h_numA;
    if (numAt) *numAt = h_numAt;
    if (n_outer) *n_outer = last;
    return 0;
}

}  // namespace
}  // namespace sbtv

using namespace sbtv;

extern "C" {

int sbtv_CSALSA_v2(sbtv_ctx *ctx, const double *y, int M, int N, int batch, const double *taps, int taille,
                   const double *mu1, const double *mu2, const double *sigma, const double *epsilon,
                   double continuationfactor, const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                   double *x_out, double *objective, double *distance1, double *distance2, double *criterion,
                   double *times, double *mses, int *numA, int *numAt, int *n_outer, int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    SBTV_TRY(check_common(ctx, "CSALSA_v2", y, taps, opts, x_init, M, N, batch, taille));
    if (!mu1 || !mu2 || !sigma) return fail(ctx, SBTV_ERR_BADARG, "CSALSA_v2: missing required argument");
    for (int b = 0; b < batch; ++b)
        if (!(mu1[b] > 0.0) || !(mu2[b] > 0.0))
            return fail(ctx, SBTV_ERR_MISSING_LS, "(A^T A + mu I)^(-1) must be specified: mu1, mu2 must be > 0");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    { FftPlan chk; SBTV_TRY(fft_plan(ctx, M, N, 1, &chk)); }
    if (sbtv_group *lg = lanes_group(ctx, batch, false)) {       // independent images: two lanes of this context (group.hip)
        LaneCall lc(ctx, lg);
        return lc.done(csalsa_sharded(lg, y, M, N, batch, taps, taille, mu1, mu2, sigma, epsilon, continuationfactor, opts,
                                      true_x, x_init, x_out, objective, distance1, distance2, criterion, times, mses, numA,
                                      numAt, n_outer, flags), batch);
    }
    const size_t P = (size_t)M * N, cnt = P * batch;
    const double *yd = nullptr, *td = nullptr, *xi = nullptr;
    SBTV_TRY(stage_in(ctx, "admm.in.y", y, cnt, flags, &yd));
    SBTV_TRY(stage_in(ctx, "admm.in.true", true_x, cnt, flags, &td));
    SBTV_TRY(stage_in(ctx, "admm.in.xinit", x_init, cnt, flags, &xi));
    double *xo = nullptr;
    SBTV_TRY(stage_out_buf(ctx, "admm.out.x", x_out, cnt, flags, &xo));
    const int mi = opts->maxiter;
    for (int b = 0; b < batch; ++b) {
        const size_t o = (size_t)b * P, r = (size_t)b * mi;
        const long long calls0 = ctx->calls;
        for (int attempt = 0; attempt < 2; ++attempt) {       // optimistic prox launches first; exact ones if the rule fired early
            const int rc = csalsa_one(ctx, yd + o, M, N, taps + (size_t)b * taille * taille, taille, mu1[b], mu2[b], sigma[b],
                                      epsilon ? epsilon[b] : 0.0, continuationfactor, opts, td ? td + o : nullptr,
                                      xi ? xi + o : nullptr, xo + o, objective ? objective + r : nullptr,
                                      distance1 ? distance1 + r : nullptr, distance2 ? distance2 + r : nullptr,
                                      criterion ? criterion + r : nullptr, times ? times + r : nullptr,
                                      mses ? mses + r : nullptr, numA ? numA + b : nullptr, numAt ? numAt + b : nullptr,
                                      n_outer ? n_outer + b : nullptr, attempt == 0);
            if (rc == ADMM_RESTART_EXACT && attempt == 0) {
                ctx->calls = calls0;
                ctx->solve_stats[0] += 1;
                continue;
            }
            SBTV_TRY(rc);
            break;
        }
    }
    SBTV_TRY(stage_out_copy(ctx, x_out, xo, cnt, flags));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return canary_epilogue(ctx, 0);
}

int sbtv_CoRAL_v2(sbtv_ctx *ctx, const double *y, int M, int N, int batch, const double *taps, int taille,
                  const double *tau1, const double *tau2, const double *mu1, const double *mu2, const double *mu_ls,
                  int TViters2, const sbtv_salsa_opts *opts, const double *true_x, const double *x_init, double *x_out,
                  double *objective, double *distance, double *times, double *mses, int *numA, int *numAt, int *n_outer,
                  int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    SBTV_TRY(check_common(ctx, "CoRAL_v2", y, taps, opts, x_init, M, N, batch, taille));
    if (!tau1 || !tau2 || !mu1 || !mu2) return fail(ctx, SBTV_ERR_BADARG, "CoRAL_v2: missing required argument");
    if (TViters2 <= 0) return fail(ctx, SBTV_ERR_MAXITER, "CoRAL_v2: TViters2 must be positive");
    for (int b = 0; b < batch; ++b)
        if (!(mu1[b] > 0.0) || !(mu2[b] > 0.0) || (mu_ls && !(mu_ls[b] > 0.0)))
            return fail(ctx, SBTV_ERR_MISSING_LS, "(A^T A + mu I)^(-1) must be specified: mu1, mu2 must be > 0");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    { FftPlan chk; SBTV_TRY(fft_plan(ctx, M, N, 1, &chk)); }
    if (sbtv_group *lg = lanes_group(ctx, batch, false)) {       // independent images: two lanes of this context (group.hip)
        LaneCall lc(ctx, lg);
        return lc.done(coral_sharded(lg, y, M, N, batch, taps, taille, tau1, tau2, mu1, mu2, mu_ls, TViters2, opts, true_x,
                                     x_init, x_out, objective, distance, times, mses, numA, numAt, n_outer, flags), batch);
    }
    const size_t P = (size_t)M * N, cnt = P * batch;
    const double *yd = nullptr, *td = nullptr, *xi = nullptr;
    SBTV_TRY(stage_in(ctx, "admm.in.y", y, cnt, flags, &yd));
    SBTV_TRY(stage_in(ctx, "admm.in.true", true_x, cnt, flags, &td));
    SBTV_TRY(stage_in(ctx, "admm.in.xinit", x_init, cnt, flags, &xi));
    double *xo = nullptr;
    SBTV_TRY(stage_out_buf(ctx, "admm.out.x", x_out, cnt, flags, &xo));
    const int mi = opts->maxiter;
    for (int b = 0; b < batch; ++b) {
        const size_t o = (size_t)b * P;
        const long long calls0 = ctx->calls;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const int rc = coral_one(ctx, yd + o, M, N, taps + (size_t)b * taille * taille, taille, tau1[b], tau2[b], mu1[b],
                                     mu2[b], mu_ls ? mu_ls[b] : mu1[b] + mu2[b], TViters2, opts, td ? td + o : nullptr,
                                     xi ? xi + o : nullptr, xo + o, objective ? objective + (size_t)b * (mi + 1) : nullptr,
                                     distance ? distance + (size_t)b * mi * 2 : nullptr,
                                     times ? times + (size_t)b * (mi + 1) : nullptr,
                                     mses ? mses + (size_t)b * (mi + 1) : nullptr, numA ? numA + b : nullptr,
                                     numAt ? numAt + b : nullptr, n_outer ? n_outer + b : nullptr, attempt == 0);
            if (rc == ADMM_RESTART_EXACT && attempt == 0) {
                ctx->calls = calls0;
                ctx->solve_stats[0] += 1;
                continue;
            }
            SBTV_TRY(rc);
            break;
        }
    }
    SBTV_TRY(stage_out_copy(ctx, x_out, xo, cnt, flags));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return canary_epilogue(ctx, 0);
}

}  // extern "C"
