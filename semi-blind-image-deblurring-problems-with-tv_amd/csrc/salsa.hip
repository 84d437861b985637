// Device-resident SALSA_v2 (ADMM, SALSA/SALSA_v2.m:389-494) over the TV prox and the spectral
// blur operator.  The host only sees eight scalars per image and outer iteration (objective
// terms, mse, distance) and applies the stopping rule; images never leave HBM.
//
// One SALSA outer iteration (fused form):
//   u      = prox_{(tau/mu) TV}(x - bu)   warm-started duals       (:429)
//   S      = rfft2(u + bu)                                        (:434, FFT of mu*(u+bu) up to the factor)
//   Xh     = (conj(H) Yh + mu S) / (|H|^2 + mu)                   (:434-436; ATy enters as conj(H) Yh)
//   resid2 = sum |Yh - H Xh|^2 / (M N)          (Parseval for  :442-444, no second FFT pair)
//   x      = irfft2(Xh)
//   bu    += u - x ; g = x - bu ; sums for mse / distance / TVnorm(u)  (:440,444,446-451)
//
// Speculative pipelining: the host enqueues outer iteration k+1 before it has seen the scalars of
// iteration k, so the GPU never waits for the stop-rule round trip.  x is double-buffered, hence
// when iteration k turns out to be the last one its x is still intact (one wasted iteration).
#include <algorithm>
#include <chrono>
#include <cmath>

#include <sys/resource.h>
#include <time.h>

#include "sbtv_internal.h"

namespace sbtv {

// images the host has frozen: their prox control block is parked (done = 1)
__global__ void prox_park_kernel(ProxCtrl *__restrict__ ctrl, const int *__restrict__ frozen, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch && frozen[b]) ctrl[b].done = 1;
}

// start of a call: per-image parameters from the pinned staging block into device memory, frozen flags cleared
__global__ void salsa_setup_kernel(const double *__restrict__ src, double *__restrict__ par, int npar, int *__restrict__ frozen,
                                   int batch) {
    for (int i = threadIdx.x; i < npar; i += blockDim.x) par[i] = src[i];
    for (int b = threadIdx.x; b < batch; b += blockDim.x) frozen[b] = 0;
}

// x = g + bu = (x - bu) + bu of an image's last iteration (NOX mode of sbtv_SALSA_v2: x is not stored in the loop)
__global__ __launch_bounds__(256) void salsa_recover_x_kernel(const double *__restrict__ g, const double *__restrict__ bu,
                                                               double *__restrict__ x, size_t P) {
#pragma clang fp contract(off)
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < P / 2; q += (size_t)gridDim.x * 256) {
        const double2 a = *reinterpret_cast<const double2 *>(g + 2 * q), b = *reinterpret_cast<const double2 *>(bu + 2 * q);
        *reinterpret_cast<double2 *>(x + 2 * q) = make_double2(a.x + b.x, a.y + b.y);
    }
}

// grid (7 [+ psteps], batch): see collect.inc (salsa_collect_block); the same blocks can ride on the first Chambolle
// launch of the next outer iteration instead (SideJob, tv_fused.inc)
__global__ __launch_bounds__(256) void salsa_collect_kernel(Collect c, ProxCtrl *__restrict__ ctrl,
                                                             unsigned long long out_addr,
                                                             const int *__restrict__ frozen, int rearm,
                                                             unsigned long long tags_addr, double seq) {
    __shared__ double red[4];
    salsa_collect_block(c, ctrl, out_addr, frozen, rearm, tags_addr, seq, blockIdx.x, blockIdx.y, red);
}

}  // namespace sbtv

using namespace sbtv;

extern "C" {

void sbtv_salsa_opts_default(sbtv_salsa_opts *o) {
    if (!o) return;
    o->stopcriterion = 1;      // SALSA_v2.m:171
    o->maxiter = 10000;        // :173
    o->TViters = 5;            // :181
    o->initialization = 0;     // :174
    o->compute_mse = 0;        // :172
    o->speculate = 1;
    o->tolA = 0.001;           // :178
    o->chambolle_tol = 1e-3;   // chambolle_prox_TV_stop.m:78
    o->chambolle_tau = 0.249;  // chambolle_prox_TV_stop.m:77
}

int sbtv_SALSA_v2(sbtv_ctx *ctx, const double *y, int M, int N, int batch, const double *taps, int taille,
                  const double *tau, const double *mu, const sbtv_salsa_opts *opts, const double *true_x,
                  const double *x_init, double *x_out, double *objective, double *distance, double *times,
                  double *mses, int *numA, int *numAt, int *n_outer, int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    if (!y || !tau || !mu || !opts || batch < 1) return fail(ctx, SBTV_ERR_BADARG, "SALSA_v2: missing required argument");
    if (!taps) return fail(ctx, SBTV_ERR_MISSING_AT, "The function handle for transpose of A is missing");
    if (opts->stopcriterion < 1 || opts->stopcriterion > 3) return fail(ctx, SBTV_ERR_STOPCRITERION, "Unknown stopping criterion");
    if (opts->initialization != 0 && opts->initialization != 2 && opts->initialization != 33333)
        return fail(ctx, SBTV_ERR_INIT, "Unknown 'Initialization' option");
    if (opts->initialization == 33333 && !x_init) return fail(ctx, SBTV_ERR_INIT, "Initialization = array but x_init is NULL");
    if (opts->TViters <= 0) return fail(ctx, SBTV_ERR_MAXITER, "SALSA_v2: TViters must be positive");
    if (taille < 1 || taille > 15 || taille > M || taille > N) return fail(ctx, SBTV_ERR_PSF, "Mask does not fit inside array");
    for (int b = 0; b < batch; ++b)
        if (!(mu[b] > 0.0)) return fail(ctx, SBTV_ERR_MISSING_LS, "(A^T A + mu I)^(-1) must be specified: mu must be > 0");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    // a batch: the images are independent (each stops by its own rule) -> two lanes of this context, group.hip
    if (sbtv_group *lg = lanes_group(ctx, batch, false)) {
        LaneCall lc(ctx, lg);
        return lc.done(salsa_sharded(lg, y, M, N, batch, taps, taille, tau, mu, opts, true_x, x_init, x_out, objective,
                                     distance, times, mses, numA, numAt, n_outer, flags), batch);
    }
    FftPlan fp;
    SBTV_TRY(fft_plan(ctx, M, N, batch, &fp));
    ProxPlan pp;
    SBTV_TRY(prox_plan(ctx, M, N, batch, &pp));
    const size_t P = (size_t)M * N, cnt = P * batch;
    const int maxiter = opts->maxiter;
    const bool want_mse = (true_x != nullptr);
    const bool crit2 = (opts->stopcriterion == 2);
    const int lag = (opts->speculate & 1) ? 1 : 0;
    const long long calls_at_entry = ctx->calls;

    // start of the device-side clock of the call (sbtv_last_timing[0]): recorded while the stream is still idle - an event
    // record between two kernels would cost the stream 5-6 us
    SBTV_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));

    // ---- stage inputs
    const double *yd = nullptr, *td = nullptr, *xi = nullptr;
    SBTV_TRY(stage_in(ctx, "salsa.y", y, cnt, flags, &yd));
    SBTV_TRY(stage_in(ctx, "salsa.true", true_x, cnt, flags, &td));
    SBTV_TRY(stage_in(ctx, "salsa.xinit", x_init, cnt, flags, &xi));
    double *xbuf[2] = {nullptr, nullptr}, *u = nullptr, *bu = nullptr, *g = nullptr;
    SBTV_TRY(ws_get_t(ctx, "salsa.x0", cnt, &xbuf[0]));
    SBTV_TRY(ws_get_t(ctx, "salsa.x1", cnt, &xbuf[1]));
    // The last possible iteration (outer == maxiter) writes its x straight into a device-resident x_out: a solve that
    // runs to MAXITERA then ends without the copy out of the double buffer (images that stop earlier are copied as
    // before, after that write).  Not when x_out overlaps an input (an optimistic solve that has to be repeated reads
    // them again) and not with captured iterations (their arguments are frozen).
    auto overlaps_out = [&](const double *p) { return p && x_out && (p < x_out + cnt) && (x_out < p + cnt); };
    const bool direct_last_ok = x_out && (flags & SBTV_DEVICE_PTRS) && !graph_wanted(cnt) && !overlaps_out(yd) &&
                                !overlaps_out(td) && !overlaps_out(xi);
    SBTV_TRY(ws_get_t(ctx, "salsa.u", cnt, &u));
    SBTV_TRY(ws_get_t(ctx, "salsa.bu", cnt, &bu));
    SBTV_TRY(ws_get_t(ctx, "salsa.g", cnt, &g));
    // NOX (sizes of the wave-granular column pass, stop criteria 1 and 3, eager launches): the inverse column pass does not
    // STORE x (one of the eight array passes of the bookkeeping kernel: -2.5 % of an outer iteration at 2048^2).  Nothing in
    // the loop reads x again; the final x of an image is recovered as g + bu = (x - bu) + bu of ITS last iteration, so g
    // and bu alternate between two buffers by iteration parity (the host learns one iteration late which one was the
    // last; x was double-buffered for the same reason).  The recovered x differs from the transform's output by the
    // rounding of that sum (<= 1 ulp of |x| + |bu|, ~3e-14 on 0..255 data).  SBTV_SALSA_NOX=0: store x.
    static const bool env_nox = [] {
        const char *e = getenv("SBTV_SALSA_NOX");
        return !(e && e[0] == '0');
    }();
    const bool nox = env_nox && fft_cols_inv_step_ok(fp) && !crit2 && !graph_wanted(cnt) && x_out != nullptr;
    double *gb[2] = {g, g}, *bub[2] = {bu, bu};
    if (nox) {
        SBTV_TRY(ws_get_t(ctx, "salsa.bu1", cnt, &bub[1]));
        SBTV_TRY(ws_get_t(ctx, "salsa.g1", cnt, &gb[1]));
    }
    const bool direct_last = direct_last_ok && !nox;
    double2 *S = nullptr, *Hs = nullptr, *Ys = nullptr;
    SBTV_TRY(ws_get_t(ctx, "salsa.S", (size_t)batch * fp.s_img, &S));
    SBTV_TRY(ws_get_t(ctx, "salsa.H", (size_t)batch * fp.u_img, &Hs));
    SBTV_TRY(ws_get_t(ctx, "salsa.Y", (size_t)batch * fp.u_img, &Ys));
    // small per-image parameter arrays: [taps | mu | thr]
    double *par = nullptr;
    const size_t npar = (size_t)batch * taille * taille + 2 * (size_t)batch;
    SBTV_TRY(ws_get_t(ctx, "salsa.par", npar, &par));
    double *taps_d = par, *mu_d = par + (size_t)batch * taille * taille, *thr_d = mu_d + batch;
    int *frozen_d = nullptr;
    SBTV_TRY(ws_get_t(ctx, "salsa.frozen", (size_t)batch, &frozen_d));
    const int nrb = fft_rows_blocks(fp);
    const int npb = fft_cols_blocks(fp);      // partial sums per image of the fused column/bookkeeping pass
    double *acc = nullptr, *postp = nullptr;
    SBTV_TRY(ws_get_t(ctx, "salsa.acc", (size_t)batch * 3 * nrb, &acc));
    SBTV_TRY(ws_get_t(ctx, "salsa.post", (size_t)batch * 6 * npb, &postp));
    SalsaScal *scal_d = nullptr;       // [2][batch]
    SBTV_TRY(ws_get_t(ctx, "salsa.scal", 2 * (size_t)batch, &scal_d));
    SalsaScal *scal_h = nullptr;       // pinned [2][batch] + frozen staging
    SalsaScal *scal_hd = nullptr;      // the same pinned block as seen from the device
    int *frozen_h = nullptr;
    {
        void *pz = nullptr;
        SBTV_TRY(pinned_get(ctx, sizeof(SalsaScal) * 3 * batch + sizeof(double) * batch * (2 * FSTRIDE + 3 * SALSA_TAGS) + sizeof(int) * batch +
                                     sizeof(double) * npar, &pz));
        scal_h = static_cast<SalsaScal *>(pz);
        void *dp = nullptr;
        SBTV_HIP(ctx, hipHostGetDevicePointer(&dp, pz, 0));
        scal_hd = static_cast<SalsaScal *>(dp);
        frozen_h = reinterpret_cast<int *>(reinterpret_cast<double *>(scal_h + 3 * (size_t)batch) + (size_t)batch * (2 * FSTRIDE + 3 * SALSA_TAGS) + npar);
        for (int b = 0; b < batch; ++b) frozen_h[b] = 0;
    }
    // completion tags [2][batch][8] behind the scalars (same pinned block): tag q of slot s = outer iteration whose
    // scalar q is in scal_h[s]
    // behind them: the prox step sums [2][batch][FSTRIDE] of the optimistic launches
    SalsaScal *init_h = scal_h + 2 * (size_t)batch, *init_hd = scal_hd + 2 * (size_t)batch;   // initial objective (own slot)
    double *psum_h = reinterpret_cast<double *>(scal_h + 3 * (size_t)batch);
    double *psum_hd = reinterpret_cast<double *>(scal_hd + 3 * (size_t)batch);
    double *tags_h = psum_h + 2 * (size_t)batch * FSTRIDE;
    double *tags_hd = psum_hd + 2 * (size_t)batch * FSTRIDE;
    double *init_tags_h = tags_h + 2 * (size_t)batch * SALSA_TAGS, *init_tags_hd = tags_hd + 2 * (size_t)batch * SALSA_TAGS;
    for (size_t i = 0; i < (size_t)3 * batch * SALSA_TAGS; ++i) tags_h[i] = 0.0;
    double *par_stage = init_tags_h + (size_t)batch * SALSA_TAGS;                             // parameter upload staging
    {
        // taps, mu and the prox threshold go up through pinned memory: no synchronisation (every call ends with one)
        for (size_t q = 0; q < (size_t)batch * taille * taille; ++q) par_stage[q] = taps[q];
        for (int b = 0; b < batch; ++b) {
            par_stage[(size_t)batch * taille * taille + b] = mu[b];
            par_stage[(size_t)batch * taille * taille + batch + b] = tau[b] / mu[b];   // threshold = tau/mu (:394)
        }
        // one small kernel reads them from the pinned block and clears the frozen flags (a memset and a copy are two blit
        // launches with 10-20 us of idle stream around each)
        const double *par_stage_hd = reinterpret_cast<const double *>(scal_hd) + (par_stage - reinterpret_cast<double *>(scal_h));
        hipLaunchKernelGGL(salsa_setup_kernel, dim3(1), dim3(256), 0, ctx->stream, par_stage_hd, par, (int)npar, frozen_d, batch);
        SBTV_HIP(ctx, hipGetLastError());
    }

    const double inv_scale = 1.0 / ((double)fp.n1 * N);
    const double parseval = 1.0 / ((double)M * N);

    // ---- operator spectra: H from the taps (resize.m), Yh = fft2(y)
    // H depends on the taps and the plan only: a call with the taps of the previous call on this context (the same problem
    // solved again, a sweep over tau / mu, the exact repeat of an optimistic solve) finds it in its workspace
    {
        const long long dims[6] = {M, N, batch, taille, fp.u_tiled, (long long)fp.u_img};
        const size_t nt = (size_t)batch * taille * taille;
        const bool same = ctx->salsa_h_ptr == (const void *)Hs && std::equal(dims, dims + 6, ctx->salsa_h_dims) &&
                          ctx->salsa_h_taps.size() == nt && std::equal(taps, taps + nt, ctx->salsa_h_taps.begin());
        if (!same) {
            ctx->salsa_h_ptr = nullptr;
            SBTV_TRY(psf_spectrum(ctx, fp, taps_d, taille, Hs));
            ctx->salsa_h_taps.assign(taps, taps + nt);
            std::copy(dims, dims + 6, ctx->salsa_h_dims);
            ctx->salsa_h_ptr = (const void *)Hs;
        }
    }
    {
        RowsArgs a{};
        a.dir_fwd = 1;
        a.op = OP_NONE;
        SBTV_TRY(fft_cols_fwd(ctx, fp, yd, nullptr, S));
        SBTV_TRY(fft_rows(ctx, fp, S, S, a));
        SBTV_TRY(spec_unpack(ctx, fp, S, Ys));
    }
    std::vector<int> h_numA(batch, 0), h_numAt(batch, 1);   // ATy = AT(y) (:288-289)
    ctx->calls += 2LL * batch;                               // AT(y), invLS(ATy) size check (:298)

    // ---- initialisation (:366-381)
    double *x = xbuf[0];
    if (opts->initialization == 0) {
        ctx->calls += batch;                                                      // AT(zeros) == 0: cleared below
    } else if (opts->initialization == 2) {
        // x = ATy = real(ifft2(conj(H) .* fft2(y)))
        RowsArgs a{};
        a.dir_fwd = 1;
        a.dir_inv = 1;
        a.op = OP_MUL_HC;
        a.H = Hs;
        SBTV_TRY(fft_cols_fwd(ctx, fp, yd, nullptr, S));
        SBTV_TRY(fft_rows(ctx, fp, S, S, a));
        SBTV_TRY(fft_cols_inv(ctx, fp, S, x, inv_scale));
    } else {
        SBTV_HIP(ctx, hipMemcpyAsync(x, xi, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // u = x ; bu = 0 ; g = x - bu = x (:392-393) ; duals zero (:418-421).  From the zero start x = u = 0, g is not read
    // before the first bookkeeping pass writes it (the first prox is not launched, see the loop), and neither start
    // clears the duals: the first prox that is launched treats them as zero (cold start flag)
    const bool zero_init = (opts->initialization == 0);
    if (!zero_init) {
        SBTV_HIP(ctx, hipMemcpyAsync(u, x, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
        SBTV_HIP(ctx, hipMemcpyAsync(g, x, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
        SBTV_HIP(ctx, hipMemsetAsync(bu, 0, sizeof(double) * cnt, ctx->stream));
    }
    SBTV_TRY(prox_reset(ctx, pp, thr_d, 1.0, opts->TViters, opts->chambolle_tol, opts->chambolle_tau, false, nullptr));

    // ---- initial objective (:399-401): resid = y - A(x).  Its scalars go to their own pinned slot with completion tags
    // and are read when the first outer iteration is processed: the loop is enqueued without waiting for them.
    std::vector<double> obj_prev(batch);
    double *o4 = nullptr;
    SBTV_TRY(ws_get_t(ctx, "salsa.o4", (size_t)batch * 4, &o4));
    {
        Collect c{};
        if (zero_init) {
            // x = u = bu = 0: resid = y, TV(u) = 0, mse(1) = sum true^2 / P.  One pass clears the three images and leaves
            // the partial sums (no transform of a zero image, no separate clears)
            double *accz = nullptr, *postz = nullptr;
            int nz = 0;
            SBTV_TRY(salsa_zero_start(ctx, yd, want_mse ? td : nullptr, x, u, bu, P, batch, (double)M * (double)N, &accz, &postz,
                                      &nz));
            c = Collect{accz, nz, nullptr, 0, postz, nz, nullptr, 0, 0, 0ull, nullptr};
        } else {
            double *tvp = nullptr;
            int ntv = 0;
            RowsArgs a{};
            a.dir_fwd = 1;
            a.op = OP_RESID;
            a.H = Hs;
            a.Y = Ys;
            a.acc = acc;
            SBTV_TRY(fft_cols_fwd(ctx, fp, x, nullptr, S));
            SBTV_TRY(tvnorm_partials(ctx, u, M, N, batch, &tvp, &ntv));
            SBTV_TRY(fft_rows(ctx, fp, S, nullptr, a));
            if (want_mse) SBTV_TRY(pair_sums(ctx, x, td, P, batch, o4));
            c = Collect{acc, nrb, tvp, ntv, nullptr, 0, nullptr, 0, 0, 0ull, want_mse ? o4 : nullptr};
        }
        hipLaunchKernelGGL(salsa_collect_kernel, dim3(7, batch), dim3(256), 0, ctx->stream, c, (ProxCtrl *)nullptr,
                           (unsigned long long)(uintptr_t)init_hd, (const int *)nullptr, 0,
                           (unsigned long long)(uintptr_t)init_tags_hd, 1.0);
        SBTV_HIP(ctx, hipGetLastError());
        ctx->calls += batch;
    }
    bool init_read = false;
    auto read_initial = [&]() -> int {
        if (init_read) return 0;
        init_read = true;
        volatile const double *tg = init_tags_h;
        for (unsigned spin = 0;; ++spin) {
            bool ready = true;
            for (int b = 0; b < batch && ready; ++b)
                for (int i = 0; i < 8 && ready; ++i) ready = (tg[(size_t)b * SALSA_TAGS + i] == 1.0);
            if (ready) break;
            if (spin > 2000) {                                 // not there after a while: let the stream finish
                SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
                break;
            }
            struct timespec ts = {0, 5000};
            nanosleep(&ts, nullptr);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        for (int b = 0; b < batch; ++b) {
            h_numA[b] += 1;
            const double f0 = 0.5 * (init_h[b].resid2 * parseval) + tau[b] * init_h[b].tv_u;
            obj_prev[b] = f0;
            if (objective) objective[(size_t)b * (maxiter + 1)] = f0;
            if (times) times[(size_t)b * (maxiter + 1)] = 0.0;
            if (mses && want_mse) mses[(size_t)b * (maxiter + 1)] = init_h[b].mse_num / (double)P;
        }
        return 0;
    };

    std::vector<int> frozen(batch, 0), h_nouter(batch, 0);
    int active = batch;
    double ms_prox = 0.0;
    long long prox_iters_run = 0;
    // captured graphs of this call: released on EVERY return path.  The six events of the loop belong to the context
    // (created on first use: creating and destroying them cost every call ~20 us of host time before its first launch)
    struct LoopResources {
        hipGraphExec_t gexec[2] = {nullptr, nullptr};
        ~LoopResources() {
            for (int s = 0; s < 2; ++s)
                if (gexec[s]) (void)hipGraphExecDestroy(gexec[s]);
        }
    } res;
    for (auto &e : ctx->loop_ev)
        if (!e) SBTV_HIP(ctx, hipEventCreate(&e));
    hipEvent_t ev_done[2] = {ctx->loop_ev[0], ctx->loop_ev[1]}, ev_p0[2] = {ctx->loop_ev[2], ctx->loop_ev[3]},
               ev_p1[2] = {ctx->loop_ev[4], ctx->loop_ev[5]};
    hipGraphExec_t(&gexec)[2] = res.gexec;
    const auto t0 = std::chrono::steady_clock::now();

    // enqueue the kernels of outer iteration `outer` (reads x = xbuf[(outer-1)&1] through g, writes
    // xbuf[outer&1]); `timed` brackets the prox with events (not inside a graph capture)
    bool prox_timed[2] = {false, false};
    // Optimistic prox launches: the Chambolle launches of an outer iteration run all TViters iterations without
    // stop-rule kernels and without the redo pass (three launches less per outer iteration); the collector applies the
    // rule over all steps at the end of the iteration.  Natural images never meet err <= tol inside a prox
    // (SURVEY.md section 8 a-1); if one does, the solve is repeated from the start with exact launches (bit 1 of
    // `speculate`), so the result is always that of the exact rule.
    // The first outer iteration always runs exactly: from the zero start its prox input is flat and the rule stops at k = 1.
    const bool spec_ok = !(opts->speculate & 2) && !graph_wanted(cnt) && prox_spec_ok(pp, g, u, opts->TViters);
    static const bool piggyback = [] {
        const char *e = getenv("SBTV_COLLECT_RIDE");
        return !(e && e[0] == '0');
    }();
    bool fired_early = false;
    bool slot_tagged[2] = {false, false}, slot_spec[2] = {false, false};
    long long prox_iters_timed = 0;
    // The collector of an optimistic iteration does not get a launch of its own when another iteration follows: its
    // blocks ride on the first Chambolle launch of that next iteration (SideJob; a launch costs ~5 us however little
    // it does).  `pend` = the collector still to be placed.  Error partials alternate between two sets by iteration
    // parity, so the riding collector reads one set while its host launch already writes the other.
    struct PendingCollect {
        bool valid = false;
        int outer = 0;
        bool spec = false;
    } pend;
    const int nl_prox = prox_launches(pp, opts->TViters);
    const bool zero_start = (opts->initialization == 0);
    auto make_collect = [&](int outer, bool spec) -> Collect {
        const int slot = outer & 1;
        return Collect{acc, nrb, nullptr, 0, postp, npb, spec ? pp.partials + (size_t)slot * pp.part_stride : nullptr, pp.fnblk,
                       opts->TViters, (unsigned long long)(uintptr_t)(psum_hd + (size_t)slot * batch * FSTRIDE), nullptr};
    };
    auto launch_collect = [&](int outer, bool spec, bool tagged) -> int {
        const int slot = outer & 1;
        const Collect c = make_collect(outer, spec);
        // the collector writes the eight scalars straight into pinned host memory (no copy kernel).  Eager launches:
        // it tags them with the iteration number (the host polls the tags); inside a captured graph the arguments are
        // frozen, so replay keeps the event
        hipLaunchKernelGGL(salsa_collect_kernel, dim3(spec ? 7 + opts->TViters : 7, batch), dim3(256), 0, ctx->stream, c, pp.ctrl,
                           (unsigned long long)(uintptr_t)(scal_hd + (size_t)slot * batch), (const int *)frozen_d, 1,
                           tagged ? (unsigned long long)(uintptr_t)(tags_hd + (size_t)slot * batch * SALSA_TAGS) : 0ull, (double)outer);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    };
    auto flush_pending = [&]() -> int {
        if (!pend.valid) return 0;
        pend.valid = false;
        return launch_collect(pend.outer, pend.spec, true);
    };
    auto enqueue_body = [&](int outer, bool eager) -> int {
        const int slot = outer & 1;
        const bool tagged = eager;
        const bool spec = spec_ok && outer >= 2;
        // the prox is bracketed by events on every 32nd iteration only: an event record leaves the stream idle for
        // 5-6 us; sbtv_last_timing scales the sampled time to all iterations
        // (4, 12, 36, 68, ...: warm-started proxes; the first one launched starts cold.  Iteration 12 as well, so that a
        // 20-step call - what the driver times - rests on two samples, not one)
        const bool timed = eager && ((outer & 31) == 4 || outer == 12);
        slot_tagged[slot] = tagged;
        slot_spec[slot] = spec;
        double *xn = (direct_last && outer == maxiter) ? x_out : xbuf[slot];
        const double *xprev = xbuf[slot ^ 1];
        // the previous iteration's collector: rides on this iteration's first (optimistic) launch, or gets its own
        SideJob side{};
        if (pend.valid && spec) {
            const int ps = pend.outer & 1;
            side.c = make_collect(pend.outer, pend.spec);
            side.out_addr = (unsigned long long)(uintptr_t)(scal_hd + (size_t)ps * batch);
            side.tags_addr = (unsigned long long)(uintptr_t)(tags_hd + (size_t)ps * batch * SALSA_TAGS);
            side.frozen = frozen_d;
            side.seq = (double)pend.outer;
            side.nblocks = 7 + (pend.spec ? opts->TViters : 0);
            pend.valid = false;
        } else {
            SBTV_TRY(flush_pending());
        }
        // (1) TV prox with warm-started duals (:429); the control block was re-armed by the first (exact) iteration's
        //     collector (or by prox_reset before the loop); optimistic launches leave it alone
        if (timed) SBTV_HIP(ctx, hipEventRecord(ev_p0[slot], ctx->stream));
        // u = g - lambda div p written by the last launch.  Optimistic mode: no stop-rule kernels, no redo pass; the
        // host applies the rule over all TViters steps when it reads the iteration's scalars
        ProxPlan pps = pp;
        pps.partials = pp.partials + (size_t)slot * pp.part_stride;
        // From the zero start (INITIALIZATION 0: x = 0, bu = 0) the first prox input is g = 0: u = div p - g/lambda = 0,
        // err = 0 <= tol, so chambolle_prox_TV_stop.m:131 stops at k = 1 with p = 0 and f = 0 - everything is already
        // in place (u = 0, duals = 0) and no launch is needed; the host books the one iteration.
        // (the duals are never cleared: the first prox that is launched starts cold)
        if (!(zero_start && outer == 1))
            SBTV_TRY(prox_iterate(ctx, pps, gb[(outer - 1) & 1], opts->TViters, u, outer == (zero_start ? 2 : 1), spec,
                                  spec ? ((outer - 2) * nl_prox) & 1 : 0, side.nblocks ? &side : nullptr));
        if (timed) SBTV_HIP(ctx, hipEventRecord(ev_p1[slot], ctx->stream));
        prox_timed[slot] = timed;
        // (2) LS step in the spectral domain + residual energy (:434-444)
        RowsArgs a{};
        a.dir_fwd = 1;
        a.dir_inv = 1;
        a.op = OP_SALSA;
        a.H = Hs;
        a.Y = Ys;
        a.mu = mu_d;
        a.acc = acc;
        a.frozen = frozen_d;
        SBTV_TRY(fft_cols_fwd_f(ctx, fp, u, bub[(outer - 1) & 1], S, frozen_d));
        SBTV_TRY(fft_rows(ctx, fp, S, S, a));
        // (3) inverse column pass fused with the bu update, the next prox input and the sums
        //     incl. TVnorm(u) (:440-451), while x is still in registers
        ColsPost cp;
        cp.u = u;
        cp.bu = bub[outer & 1];
        cp.g = gb[outer & 1];
        if (nox) {
            cp.bu_in = bub[(outer - 1) & 1];
            cp.skip_x = 1;
        }
        cp.tru = td;
        cp.xprev = crit2 ? xprev : nullptr;
        cp.partials = postp;
        SBTV_TRY(fft_cols_inv_post(ctx, fp, S, xn, inv_scale, frozen_d, cp));
        // (4) the collector: deferred to the next iteration's first launch when that iteration will exist and be
        //     optimistic too (the host is then one iteration ahead anyway), else launched here
        if (spec && eager && lag == 1 && outer < maxiter && piggyback) {
            pend.valid = true;
            pend.outer = outer;
            pend.spec = spec;
            return 0;
        }
        return launch_collect(outer, spec, tagged);
    };
    // Small problems are launch-bound (about a dozen kernels of a few microseconds): from the third outer
    // iteration on, the body of each x-buffer slot is captured once and replayed with one hipGraphLaunch.
    bool use_graph = graph_wanted(cnt);
    auto enqueue = [&](int outer) -> int {
        const int slot = outer & 1;
        if (use_graph && outer >= 3) {
            if (!gexec[slot]) {
                if (graph_begin(ctx) != 0 || graph_end(ctx, enqueue_body(outer, false), &gexec[slot]) != 0) {
                    use_graph = false;                    // capture unavailable: keep launching eagerly
                    gexec[slot] = nullptr;
                }
            }
            if (gexec[slot]) {
                SBTV_HIP(ctx, hipGraphLaunch(gexec[slot], ctx->stream));
                SBTV_HIP(ctx, hipEventRecord(ev_done[slot], ctx->stream));
                return 0;
            }
        }
        SBTV_TRY(enqueue_body(outer, true));
        return 0;
    };
    // wait until the collector of iteration `outer` has delivered all eight scalars of every image: poll the tags (the
    // host is normally one iteration ahead), yielding the core between polls.  No HIP call in the normal case: a
    // stream query makes the runtime append a marker packet, which costs the stream 5-6 us before the next kernel.
    // Only after 50 ms without the tags is the stream asked, so that a failed launch cannot leave the host waiting.
    // The wait has three phases: (1) spin on the tags for up to `spin_us` microseconds (default 150; SBTV_TAG_SPIN_US):
    // an outer iteration of a small image takes 50 us and only ONE more iteration is queued behind it, while a
    // nanosleep of 5 us returns after 55-60 us (the kernel's default timer slack is 50 us) or much later when the core
    // went into a deep idle state - a host that sleeps there lets the queue run dry and a 512^2 solve then runs at a
    // third of its speed (the "slow mode" of round 2, `sbtv_last_host_stats`); (2) sleep between polls - a 2048^2
    // iteration takes 240 us, the spin would burn a core for nothing; (3) after 50 ms ask the stream.
    static const double spin_us = [] {
        const char *e = getenv("SBTV_TAG_SPIN_US");
        return e ? atof(e) : 150.0;
    }();
    ctx->hstat = HostStats{};
    struct rusage ru0 {};
    (void)getrusage(RUSAGE_THREAD, &ru0);
    auto wait_tags = [&](int slot, int outer) -> int {
        volatile const double *tg = tags_h + (size_t)slot * batch * SALSA_TAGS;
        const double want = (double)outer;
        const int ntag = 8 + (slot_spec[slot] ? opts->TViters : 0);
        const auto t_begin = std::chrono::steady_clock::now();
        auto t_query = t_begin;
        bool slept = false;
        HostStats &hs = ctx->hstat;
        hs.waits += 1;
        for (unsigned spin = 0;; ++spin) {
            bool ready = true;
            for (int b = 0; b < batch && ready; ++b)
                for (int i = 0; i < ntag && ready; ++i) ready = (tg[(size_t)b * SALSA_TAGS + i] == want);
            if (ready) {
                if (spin == 0) hs.ready_at_once += 1;
                break;
            }
            __builtin_ia32_pause();
            if ((spin & 15) != 15) continue;                       // look at the clock every 16th poll only
            const auto now = std::chrono::steady_clock::now();
            if (std::chrono::duration<double, std::micro>(now - t_begin).count() < spin_us) continue;
            struct timespec ts = {0, 5000};
            nanosleep(&ts, nullptr);
            hs.sleeps += 1;
            slept = true;
            if (now - t_query > std::chrono::milliseconds(50)) {
                hs.stream_queries += 1;
                const hipError_t e = hipStreamQuery(ctx->stream);
                if (e == hipSuccess) {
                    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));     // everything has run: the scalars are there
                    break;
                }
                if (e != hipErrorNotReady) return fail_hip(ctx, e, "hipStreamQuery", __FILE__, __LINE__);
                t_query = std::chrono::steady_clock::now();
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const double w = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        hs.wait_s += w;
        if (w > hs.wait_max_s) {
            hs.wait_max_s = w;
            hs.wait_max_outer = (double)outer;
        }
        if (slept) hs.waits_slept += 1;
        return 0;
    };
    // host side of outer iteration `outer`: traces + stopping rule (:444-482)
    auto process = [&](int outer) -> int {
        const int slot = outer & 1;
        SBTV_TRY(read_initial());                                            // objective(1), mses(1): long there by now
        if (pend.valid && pend.outer == outer) SBTV_TRY(flush_pending());     // no later iteration took it along
        if (slot_tagged[slot]) SBTV_TRY(wait_tags(slot, outer));
        else SBTV_TRY(wait_event(ctx, ev_done[slot]));
        if (prox_timed[slot]) {
            float ms = 0.f;
            SBTV_HIP(ctx, hipEventSynchronize(ev_p1[slot]));
            SBTV_HIP(ctx, hipEventElapsedTime(&ms, ev_p0[slot], ev_p1[slot]));
            ms_prox += ms;
            for (int b = 0; b < batch; ++b)
                if (!frozen[b]) prox_iters_timed += slot_spec[slot] ? (long long)opts->TViters : (long long)scal_h[(size_t)slot * batch + b].pad;
            prox_timed[slot] = false;
        }
        const double tnow = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        bool changed = false;
        for (int b = 0; b < batch; ++b) {
            if (frozen[b]) continue;
            const SalsaScal &s = scal_h[(size_t)slot * batch + b];
            h_numA[b] += 1;
            h_nouter[b] = outer;
            double prox_k = s.pad;                   // exact launches: iterations booked by the stop-rule kernels
            if (zero_start && outer == 1) prox_k = 1.0;   // the prox of a zero image: one iteration, not launched
            if (slot_spec[slot]) {
                // optimistic launches: cont = (k < MaxIter) & (err > tol)  (chambolle_prox_TV_stop.m:131) over the steps
                const double *ps = psum_h + ((size_t)slot * batch + b) * FSTRIDE;
                prox_k = (double)opts->TViters;
                for (int k = 1; k < opts->TViters; ++k)
                    if (!(sqrt(ps[k - 1]) > opts->chambolle_tol * SPEC_TOL_GUARD)) {
                        fired_early = true;          // the rule stopped before the last step: start over, exactly
                        break;
                    }
                if (fired_early) break;
                // the sums of launches that sum a subset of the pixels are lower bounds: long before they can come near
                // tol^2 the launches of this solve go back to the full sums (ProxPlan::esub_off)
                for (int k = 1; k <= opts->TViters && !pp.esub_off; ++k)
                    if (!(ps[k - 1] > ESUB_MARGIN * opts->chambolle_tol * opts->chambolle_tol)) {
                        pp.esub_off = 1;
                        ctx->solve_stats[1] += 1;
                    }
            }
            prox_iters_run += (long long)prox_k;
            ctx->calls += 2;   // invLS + A (callcounter)
            const double f = 0.5 * (s.resid2 * parseval) + tau[b] * s.tv_u;                  // :444
            if (objective) objective[(size_t)b * (maxiter + 1) + outer] = f;
            if (mses && want_mse) mses[(size_t)b * (maxiter + 1) + outer] = s.mse_num / (double)P;   // :446-449
            if (distance) distance[(size_t)b * maxiter + (outer - 1)] = sqrt(s.dist_num) / sqrt(s.x2 + s.u2);   // :451
            if (times) times[(size_t)b * (maxiter + 1) + outer] = tnow;
            bool stop = false;
            if (outer > 1) {                                                                  // :453
                double crit;
                if (opts->stopcriterion == 1)
                    crit = fabs(f - obj_prev[b]) / obj_prev[b];                               // :458
                else if (opts->stopcriterion == 2)
                    crit = fabs(sqrt(s.dx2) / sqrt(s.x2));                                    // :462
                else
                    crit = f;                                                                 // :465
                stop = crit < opts->tolA;                                                     // :472
            }
            obj_prev[b] = f;
            if (stop) {
                frozen[b] = 1;
                frozen_h[b] = 1;
                --active;
                changed = true;
            }
        }
        if (fired_early) return 0;
        if (changed && active > 0) {
            SBTV_HIP(ctx, hipMemcpyAsync(frozen_d, frozen_h, sizeof(int) * batch, hipMemcpyHostToDevice, ctx->stream));
            // park the frozen images' prox as well (optimistic launches do not re-arm the control blocks)
            hipLaunchKernelGGL(prox_park_kernel, dim3((batch + 63) / 64), dim3(64), 0, ctx->stream, pp.ctrl, (const int *)frozen_d, batch);
            SBTV_HIP(ctx, hipGetLastError());
        }
        return 0;
    };

    int rc = 0, enq = 0, done = 0;
    while (active > 0 && done < maxiter) {
        // keep up to 1 + lag iterations in flight
        while (rc == 0 && enq < maxiter && enq - done <= lag && active > 0) {
            const auto te = std::chrono::steady_clock::now();
            rc = enqueue(++enq);
            const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - te).count();
            ctx->hstat.enqueue_s += d;
            if (d > ctx->hstat.enqueue_max_s) ctx->hstat.enqueue_max_s = d;
        }
        if (rc != 0) break;
        rc = process(++done);
        if (rc != 0 || fired_early) break;
    }
    if (rc != 0) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    if (fired_early) {
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->calls = calls_at_entry;
        ctx->solve_stats[0] += 1;
        sbtv_salsa_opts exact = *opts;
        exact.speculate = (opts->speculate & 1) | 2;
        return sbtv_SALSA_v2(ctx, y, M, N, batch, taps, taille, tau, mu, &exact, true_x, x_init, x_out, objective, distance,
                             times, mses, numA, numAt, n_outer, flags);
    }
    SBTV_TRY(read_initial());
    {
        struct rusage ru1 {};
        (void)getrusage(RUSAGE_THREAD, &ru1);
        ctx->hstat.nvcsw = (double)(ru1.ru_nvcsw - ru0.ru_nvcsw);
        ctx->hstat.nivcsw = (double)(ru1.ru_nivcsw - ru0.ru_nivcsw);
        ctx->hstat.minflt = (double)(ru1.ru_minflt - ru0.ru_minflt);
        ctx->hstat.majflt = (double)(ru1.ru_majflt - ru0.ru_majflt);
    }
    SBTV_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (x_out && nox) {
        // image b's result is g + bu of ITS last processed iteration (x itself was never stored)
        for (int b = 0; b < batch; ++b) {
            const int par = h_nouter[b] & 1;
            const bool dev_direct = direct_last_ok;      // device-resident x_out that overlaps no input
            double *dst = dev_direct ? x_out + (size_t)b * P : xbuf[0] + (size_t)b * P;
            const int nblk = (int)std::min<size_t>((P / 2 + 255) / 256, 2048);
            if (h_nouter[b] == 0) {          // no iteration ran (maxiter = 0): the start image
                SBTV_HIP(ctx, hipMemcpyAsync(dst, x + (size_t)b * P, sizeof(double) * P, hipMemcpyDeviceToDevice, ctx->stream));
            } else {
                hipLaunchKernelGGL(salsa_recover_x_kernel, dim3(nblk), dim3(256), 0, ctx->stream,
                                   (const double *)(gb[par] + (size_t)b * P), (const double *)(bub[par] + (size_t)b * P), dst, P);
                SBTV_HIP(ctx, hipGetLastError());
            }
            if (!dev_direct && (flags & SBTV_DEVICE_PTRS))
                SBTV_HIP(ctx, hipMemcpyAsync(x_out + (size_t)b * P, dst, sizeof(double) * P, hipMemcpyDeviceToDevice, ctx->stream));
            else if (!dev_direct)
                SBTV_TRY(stage_out_copy(ctx, x_out + (size_t)b * P, dst, P, flags));        // host: through the copy lanes
        }
    } else if (x_out) {
        // image b's result is the x written by ITS last processed iteration
        for (int b = 0; b < batch; ++b) {
            if (direct_last && h_nouter[b] == maxiter) continue;        // already there
            const double *src = xbuf[h_nouter[b] & 1] + (size_t)b * P;
            if (flags & SBTV_DEVICE_PTRS)
                SBTV_HIP(ctx, hipMemcpyAsync(x_out + (size_t)b * P, src, sizeof(double) * P, hipMemcpyDeviceToDevice, ctx->stream));
            else
                SBTV_TRY(stage_out_copy(ctx, x_out + (size_t)b * P, src, P, flags));        // host: through the copy lanes
        }
    }
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the one synchronisation of the call
    {
        float ms = 0.f;
        SBTV_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ctx->timing[0] = ms;
        // time inside the Chambolle launches: measured on the sampled iterations, scaled to all of them
        ctx->timing[1] = prox_iters_timed > 0 ? ms_prox * ((double)prox_iters_run / (double)prox_iters_timed) : 0.0;
        ctx->timing[2] = (double)prox_iters_run / batch;      // Chambolle iterations (image-averaged)
        ctx->timing[3] = 40.0 * (double)P * (double)prox_iters_run;
    }
    for (int b = 0; b < batch; ++b) {
        if (numA) numA[b] = h_numA[b];
        if (numAt) numAt[b] = h_numAt[b];
        if (n_outer) n_outer[b] = h_nouter[b];
    }
    return canary_epilogue(ctx, 0);
}

}  // extern "C"
