/*
 * sbtv.h — C-ABI of libsbtv.so: the MI355X (gfx950) implementation of the
 * FFT-convolution + TV-proximal inner loop of SALSA / FISTA / SAPG(MYULA).
 *
 * The reference (charles-kmc/Semi-blind-image-deblurring-problems-with-TV) is
 * pure MATLAB and has no FFI layer: its "operator API" is function handles
 * and name/value option lists.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference checkout).  A host
 * binds these with MATLAB loadlibrary/calllib (this header is plain C, no
 * mex.h), a MEX gateway, or Python ctypes — see INTEGRATION.md.
 *
 * Conventions
 *   - all image buffers are IEEE double, column-major (MATLAB layout):
 *     element (i,j) of image b lives at  buf[b*M*N + j*M + i],  M rows, N cols.
 *   - image sizes: 2 <= M, N <= 4096 for every entry point that applies the blur
 *     operator (SBTV_ERR_SIZE otherwise).  Powers of two from 16 take the tuned radix-2^k
 *     real-FFT kernels; any other size runs like the reference's fft2 closures do
 *     (utils/resize.m:1-12) through a chirp-z (Bluestein) complex transform, several
 *     times slower.  FISTA, SAPG / MYULA, max_eigenval, C-SALSA and CoRAL additionally
 *     need an even number of pixels.  The TV entry points (prox, TVnorm) accept any
 *     M >= 2, N >= 2 (even M takes the fused 16-byte-per-lane kernels, odd M a scalar
 *     one-iteration kernel).
 *   - `flags & SBTV_DEVICE_PTRS`: image buffers are device pointers on the
 *     context's GPU (no PCIe copies; asynchronous on the context stream).
 *     Otherwise they are host pointers and the call copies in/out and returns
 *     after the results are on the host.  Small option/result arrays
 *     (scalars per image, traces) are ALWAYS host pointers.
 *   - return value: 0 = ok; < 0 argument errors (mirror the reference's
 *     error() sites); > 0 HIP runtime errors.  sbtv_last_error() gives text.
 *     Nothing throws across the boundary.
 *   - one context = one GPU = one host thread at a time; sbtv_group (below) bundles one context per GPU behind
 *     one call for hosts that are a single process.
 */
#ifndef SBTV_H
#define SBTV_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBTV_VERSION 100

/* flags */
#define SBTV_HOST_PTRS   0
#define SBTV_DEVICE_PTRS 1
/* sbtv_SAPG_algorithm only: */
#define SBTV_REDUCE_DEVICE  2   /* reduce_fn is an sbtv_allreduce_dev_fn (in-stream collective on a device buffer)  */
#define SBTV_SAPG_HOST_LOOP 4   /* parameter updates on the host, one synchronisation per iteration (the round-1 loop) */
/* sbtv_fista_tv only: */
#define SBTV_FISTA_EXACT_PROX 2 /* stop-rule kernel after every Chambolle launch (default: the launches of a prox run all
                                 * prox_iters iterations, the host applies the rule of chambolle_prox_TV_stop.m:131 over the
                                 * steps afterwards and repeats the call with this flag if it stopped early) */

/* status codes */
#define SBTV_OK                   0
#define SBTV_ERR_BADARG          -1   /* generic bad argument                                  */
#define SBTV_ERR_SIZE            -2   /* unsupported image size (outside 2..4096, or an odd pixel count where noted) */
#define SBTV_ERR_MAXITER         -3   /* chambolle: 'maxiter' missing  (chambolle_prox_TV_stop.m:95,131, quirk Q1) */
#define SBTV_ERR_DUALVARS        -4   /* 'Wrong size of the dual variables' (chambolle_prox_TV_stop.m:103)          */
#define SBTV_ERR_MODE            -5   /* 'The value of parameter mode must be 1 or 2' (A_wrapper.m:15)              */
#define SBTV_ERR_STOPCRITERION   -6   /* 'Unknown stopping criterion' (SALSA_v2.m:246; my_fista.m:45)               */
#define SBTV_ERR_INIT            -7   /* "Unknown 'Initialization' option" (SALSA_v2.m:382)                          */
#define SBTV_ERR_MISSING_AT      -8   /* 'The function handle for transpose of A is missing' (SALSA_v2.m:262)        */
#define SBTV_ERR_MISSING_LS      -9   /* '(A^T A + mu I)^(-1) must be specified' (SALSA_v2.m:296)                    */
#define SBTV_ERR_PSF            -10   /* bad PSF size / mask does not fit (conv2c.m:15)                               */
#define SBTV_ERR_NOMEM          -11
#define SBTV_ERR_NODEVICE       -12   /* no usable GPU: the library has NO CPU fallback                               */
#define SBTV_ERR_PEER           -14   /* shared-gradient chains: another rank reported an error through reduce_fn     */
#define SBTV_ERR_CANARY         -13   /* SBTV_CANARY=1: a kernel wrote outside its workspace (guard band damaged)     */

typedef struct sbtv_ctx sbtv_ctx;

/* ---- context ---------------------------------------------------------- */
int         sbtv_version(void);
/* Create a context on GPU `device`.  Fails with SBTV_ERR_NODEVICE when no
 * gfx950-capable device is visible (there is deliberately no CPU path). */
int         sbtv_ctx_create(int device, sbtv_ctx **out);
int         sbtv_ctx_destroy(sbtv_ctx *ctx);
const char *sbtv_last_error(const sbtv_ctx *ctx);   /* ctx may be NULL: global message */
/* Use an externally owned hipStream_t (e.g. torch's current stream). NULL = own stream. */
int         sbtv_ctx_set_stream(sbtv_ctx *ctx, void *hip_stream);
int         sbtv_ctx_sync(sbtv_ctx *ctx);
/* Lanes (no counterpart in the reference): a call with batch >= 2 independent items (the images of sbtv_SALSA_v2 /
 * sbtv_fista_tv / sbtv_CSALSA_v2 / sbtv_CoRAL_v2, the independent chains of sbtv_SAPG_algorithm) is dealt in two
 * contiguous halves to two internal contexts on the same GPU - own stream, workspaces and host thread each - so that the
 * launch tails and memory-bound passes of one half run under the Chambolle launches of the other (+15..20 % image-
 * iterations/s).  Results do not change: a batched call computes image k bit for bit like image k alone.
 *   mode 0 (default) independent items only;  1 never (one stream);  2 also shared-gradient chains (share_gradients = 1:
 *   the two halves then exchange their six gradient sums in-stream once per SAPG iteration, results equal to rounding).
 * The environment variable SBTV_LANES = 0 | 1 | 2 overrides the mode of every context.  With a caller-owned stream
 * (sbtv_ctx_set_stream) the call first waits for that stream, then runs on the lanes' own streams. */
int         sbtv_ctx_set_lanes(sbtv_ctx *ctx, int mode);
/* Operator-call counters: the reference's global `calls` (SALSA/callcounter.m:8-15). */
int         sbtv_callcounter_get(const sbtv_ctx *ctx, long long *calls);
int         sbtv_callcounter_reset(sbtv_ctx *ctx);
/* Timing of the most recent solver call, measured with HIP events on the
 * context stream: [0] total ms of the call on the device (set-up and iteration loop), [1] ms inside the
 * Chambolle iteration kernels, [2] number of Chambolle iteration launches,
 * [3] algorithmic bytes those launches moved (40 B/pixel/iteration). */
int         sbtv_last_timing(const sbtv_ctx *ctx, double out[4]);

/* raw device memory helpers so hosts without a GPU array type (MATLAB) can
 * keep buffers resident between calls */
int         sbtv_malloc(sbtv_ctx *ctx, size_t bytes, void **dptr);
int         sbtv_free(sbtv_ctx *ctx, void *dptr);
int         sbtv_memcpy_h2d(sbtv_ctx *ctx, void *dst, const void *src, size_t bytes);
int         sbtv_memcpy_d2h(sbtv_ctx *ctx, void *dst, const void *src, size_t bytes);

/* ---- a-1: TV proximal operator ----------------------------------------
 * Replaces  [f,px,py] = chambolle_prox_TV_stop(g,'lambda',L,'maxiter',K,
 *                        'tol',t,'tau',T,'dualvars',[px py])
 * (utils/chambolle_prox_TV_stop.m:1-166), batched over `batch` images.
 *   lambda[batch]   regularisation weight per image
 *   maxiter         REQUIRED, > 0 (quirk Q1) else SBTV_ERR_MAXITER
 *   tol, tau        reference defaults 1e-3, 0.249 (:77-78)
 *   warm_start      0: px=py=0 (:68-69); 1: px,py hold the dual variables on entry (:99-107)
 *   px,py,f         M*N*batch each; px,py always written; f may be NULL
 *   k_out[batch]    iterations actually run;  err_out[batch] last `err` (:128)
 */
int sbtv_chambolle_prox_TV_stop(sbtv_ctx *ctx, const double *g, int M, int N, int batch,
                                const double *lambda, int maxiter, double tol, double tau,
                                int warm_start, double *px, double *py, double *f,
                                int *k_out, double *err_out, int flags);

/* ---- a-2: periodic isotropic TV ----------------------------------------
 * Replaces  TVnorm(x)  (utils/TVnorm.m:2; SALSA/diffh.m, diffv.m, conv2c.m). out[batch]. */
int sbtv_TVnorm(sbtv_ctx *ctx, const double *x, int M, int N, int batch, double *out, int flags);

/* ---- a-3/a-4: circular blur operator from PSF taps ----------------------
 * The reference builds A, AT, dA/dp, invLS as FFT closures over
 * resize(h) = fft2 of the taille x taille taps zero-padded into the TOP-LEFT
 * corner (utils/resize.m:1-12; run_Gaussian_demo.m:126-139,224-225).  Here a
 * PSF is its taps (column-major taille x taille, taille <= 15, per image).
 *
 * sbtv_A_wrapper replaces A_wrapper(A,AT,x,M1,N1,M2,N2,mode) (SALSA/A_wrapper.m:5-17)
 * for the closures of the demos:
 *   mode 1: A x   = real(ifft2( H        .* fft2(x)))      run_Gaussian_demo.m:136
 *   mode 2: AT x  = real(ifft2( conj(H)  .* fft2(x)))      run_Gaussian_demo.m:137
 *   mode 3: dA x  = same as mode 1 (pass the derivative taps)   :138-139
 *   mode 9: invLS = real(ifft2( fft2(x) ./ (abs(H).^2 + mu)))   :224-225
 * other modes -> SBTV_ERR_MODE.  Each call bumps the call counter by `batch`.
 */
int sbtv_A_wrapper(sbtv_ctx *ctx, const double *taps, int taille, const double *mu,
                   const double *x, double *out, int M, int N, int batch, int mode, int flags);

/* PSF tap builders (host arithmetic, exactly the reference formulas).
 * kind 0 gaussian  p = {w1, w2, phi}  utils/Gaussian_psf.m:2-19, Sum_gauss_psf.m, diff_fftgaus_w1/w2.m
 * kind 1 moffat    p = {alpha, beta}  utils/psf_moffat.m:2-20, sum_mof_psf.m, diff_moffat_alpha/beta.m
 * kind 2 laplace   p = {b}            utils/psf_laplace.m:1-13, sum_lap_psf.m, diff_laplace_b.m
 * taps, d0, d1: taille*taille doubles (column-major); d0/d1 may be NULL. */
#define SBTV_PSF_GAUSSIAN 0
#define SBTV_PSF_MOFFAT   1
#define SBTV_PSF_LAPLACE  2
int sbtv_psf_taps(int kind, int taille, const double *p, double *taps, double *d0, double *d1);

/* The PSF-tracking trace results.err_psf of the SAPG loops (SAPG_algorithm_Guassian.m:146,203-204, _moffat.m:204-205,
 * _laplace.m:136,190-191):  out[i] = l2(psf(params(i)), psf(p_true)) with utils/l2.m = norm(.)^2 of the MATRIX
 * (spectral norm, quirk Q9).  ps = [first-parameter trace (n) | second-parameter trace (n)], host arrays; the
 * Gaussian pairs w1s(i) with w2s(i-1) (quirk Q8), the Moffat's first entry stays 0 as in the reference. */
int sbtv_err_psf(int kind, int taille, const double *ps, int n, const double *p_true, double phi, double *out);

/* Packed half-spectrum of a real image (debug / test entry for the FFT
 * kernels): out is (M/2) x N complex, column-major, interleaved re/im; row 0
 * holds X[0,l] + i*X[M/2,l].  inverse=1 maps it back (scaled like ifft2). */
int sbtv_rfft2_packed(sbtv_ctx *ctx, const double *in, double *out, int M, int N, int batch,
                      int inverse, int flags);

/* ---- a-7: SALSA_v2 -------------------------------------------------------
 * Replaces  [x,numA,numAt,objective,distance,times,mses] = SALSA_v2(y,A,tau,
 *   'MU',mu,'AT',AT,'StopCriterion',c,'True_x',x,'ToleranceA',tol,'MAXITERA',n,
 *   'TVINITIALIZATION',1,'TViters',k,'LS',invLS,...)   (SALSA/SALSA_v2.m:156-494)
 * with A/AT/invLS the FFT closures defined by `taps` (run_Gaussian_demo.m:215-242).
 * Only the TV path ('TVINITIALIZATION' = 1) exists: like the reference (:318-320,
 * quirk Q7) user Psi/Phi are ignored in that mode.
 */
typedef struct sbtv_salsa_opts {
    int    stopcriterion;    /* 1,2,3  (SALSA_v2.m:245-247,456-469)                 */
    int    maxiter;          /* 'MAXITERA'  default 10000 (:173)                    */
    int    TViters;          /* 'TVITERS'   default 5 (:181)                        */
    int    initialization;   /* 0 zeros (:369), 2 AT*y (:373), 33333 x_init given   */
    int    compute_mse;      /* 1 when 'TRUE_X' given (:227-229)                    */
    int    speculate;        /* bit 0 (default 1): the host evaluates the outer stop rule one iteration late while the
                              * next iteration already runs (0: it waits for every iteration).
                              * bit 1: never launch the TV prox optimistically.  By default the Chambolle launches of an
                              * outer iteration run all TViters iterations without stop-rule kernels and the rule
                              * (chambolle_prox_TV_stop.m:131) is applied over those steps at the end of the iteration;
                              * if it fired before the last one the whole solve is repeated with exact launches, so the
                              * result is always that of the exact rule (TViters <= 10, even M). */
    double tolA;             /* 'TOLERANCEA' default 1e-3 (:178)                    */
    double chambolle_tol;    /* 1e-3  (chambolle_prox_TV_stop.m:78)                 */
    double chambolle_tau;    /* 0.249 (chambolle_prox_TV_stop.m:77)                 */
} sbtv_salsa_opts;

void sbtv_salsa_opts_default(sbtv_salsa_opts *o);

/*   y, true_x, x_init, x_out : M*N*batch images (true_x / x_init may be NULL)
 *   taps[batch*taille^2], tau[batch], mu[batch]            (host arrays)
 *   objective[batch*(maxiter+1)], distance[batch*maxiter], times[batch*(maxiter+1)],
 *   mses[batch*(maxiter+1)]  (host arrays, any may be NULL; row b starts at b*(maxiter+1)
 *   resp. b*maxiter);  numA, numAt, n_outer: [batch] (host, may be NULL)
 *   Images in a batch iterate in lock-step; an image that met its stop rule is
 *   frozen (its x no longer changes) while the others continue.
 *   x_out may overlap an input (then it is written once, at the end); a device-resident
 *   x_out that overlaps none is also written by the iteration numbered maxiter, so its
 *   contents are undefined until the call returns. */
int sbtv_SALSA_v2(sbtv_ctx *ctx, const double *y, int M, int N, int batch,
                  const double *taps, int taille, const double *tau, const double *mu,
                  const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                  double *x_out, double *objective, double *distance, double *times, double *mses,
                  int *numA, int *numAt, int *n_outer, int flags);

/* ---- f-3: the other ADMM front-ends over the same kernels ----------------
 * (never called by the reference's demos)  Only the TV paths exist
 * ('TVINITIALIZATION*' = 1, P/PT = identity).  Images of a batch are solved one after another.
 *
 * sbtv_CSALSA_v2 replaces
 *   [x,numA,numAt,objective,distance1,distance2,criterion,times,mses] = csalsa(y,A,mu1,mu2,sigma,
 *      'AT',AT,'LS',invLS,'TVINITIALIZATION',1,'TVITERS',k,'STOPCRITERION',c,'TOLERANCEA',tol,
 *      'MAXITERA',n,'CONTINUATIONFACTOR',delta,'EPSILON',eps,...)   (SALSA/CSALSA_v2.m:160-561)
 * with invLS(r,mu) = real(ifft2(fft2(r)./(|H|^2+mu))) (:116 of its help, called as invLS(r,mu1) :471).
 * opts: stopcriterion 1..3 (default of the reference: 3), maxiter, TViters, initialization, tolA and the
 * Chambolle knobs are used; `speculate` as for sbtv_SALSA_v2 (bit 0: the host evaluates the stop rule one iteration late,
 * bit 1: never launch the TV prox optimistically); a continuation factor != 1 selects exact launches and no lag.  Traces are 1-based like the reference: entry 0 is the
 * state before the loop, the loop runs outer = 2..maxiter (:461).  All trace rows have maxiter entries.
 * epsilon[b] = 0 selects sqrt(numel(y)+8*sqrt(numel(y)))*sigma[b] (:413).  n_outer[b] = last `outer`. */
int sbtv_CSALSA_v2(sbtv_ctx *ctx, const double *y, int M, int N, int batch,
                   const double *taps, int taille, const double *mu1, const double *mu2,
                   const double *sigma, const double *epsilon, double continuationfactor,
                   const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                   double *x_out, double *objective, double *distance1, double *distance2,
                   double *criterion, double *times, double *mses,
                   int *numA, int *numAt, int *n_outer, int flags);

/* sbtv_CoRAL_v2 replaces
 *   [x,numA,numAt,objective,distance,times,mses] = CoRAL(y,A,tau1,tau2,'MU1',mu1,'MU2',mu2,'AT',AT,
 *      'LS',invLS,'TVINITIALIZATION1',1,'TVITERS1',k1,'TVINITIALIZATION2',1,'TVITERS2',k2,...)
 *   (SALSA/CoRAL_v2.m:2-476), invLS(r) = real(ifft2(fft2(r)./(|H|^2+mu_ls))), mu_ls = mu1+mu2 (:137)
 * when mu_ls is NULL.  opts->TViters is TViters1; TViters2 is a separate argument.
 * objective/times/mses: [batch*(maxiter+1)]; distance: [batch*maxiter*2], entry (outer-1)*2+{0,1}. */
int sbtv_CoRAL_v2(sbtv_ctx *ctx, const double *y, int M, int N, int batch,
                  const double *taps, int taille, const double *tau1, const double *tau2,
                  const double *mu1, const double *mu2, const double *mu_ls, int TViters2,
                  const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                  double *x_out, double *objective, double *distance, double *times, double *mses,
                  int *numA, int *numAt, int *n_outer, int flags);

/* ---- a-8: FISTA with the TV prox ----------------------------------------
 * Replaces my_fista(b,A,AT,tau,L,Phi,Psi,stopcriterion,tolerance,maxiters,true,verbose)
 * (SALSA/my_fista.m:5-56) with Psi = cold-start Chambolle(prox_iters) and Phi = TVnorm
 * (run_moffat_demo.m:181-182), and my_deblur_fista (SALSA/my_deblur_fista.m:5-68) when
 * zero_start = 1 and L = 1.  objective/mses: [batch*maxiters]; n_iter[batch]. */
int sbtv_fista_tv(sbtv_ctx *ctx, const double *b, int M, int N, int batch,
                  const double *taps, int taille, const double *tau, double L,
                  int prox_iters, int stopcriterion, double tolerance, int maxiters,
                  int zero_start, const double *true_x, double *x_out,
                  double *objective, double *mses, int *n_iter, int flags);

/* ---- a-5/a-6: SAPG (MYULA) parameter estimation ---------------------------
 * Replaces SAPG_algorithm_Guassian / _moffat / _laplace (SAPG/ directory) for `batch`
 * independent chains.  One chain = one image y_b with its own state.  */
typedef struct sbtv_sapg_opts {
    int    kind;              /* SBTV_PSF_*                                              */
    int    psf_size;          /* 7                                                       */
    int    samples;           /* op.samples  (total_iter)                                */
    int    warmup;            /* op.warmup                                               */
    int    burnIn;            /* op.burnIn (1-based like the reference)                  */
    int    chambolleit;       /* 25 (run_Gaussian_demo.m:188)                            */
    int    fix_p[2];          /* op.fix_w1/op.fix_w2 (alpha/beta, b)                     */
    int    fix_sigma;
    int    share_gradients;   /* 0: independent chains. 1: all chains sample one image and
                                 average their gradients (the reference's vestigial
                                 `for jj=1:1 ... mean(g_*)`, SAPG_algorithm_moffat.m:158-173) */
    double lambda, gamma;     /* c.lam*op.lambda, c.gam*op.gamma                         */
    double th_init, min_th, max_th;
    double p_init[2], p_min[2], p_max[2], p_true[2], phi;
    double sigma2_init, sigma2_min, sigma2_max, sigma2_true;
    double d_scale, d_exp;    /* delta(i) = d_scale * i^-d_exp / dimX                    */
    double c_theta, c_p[2], c_sigma;
    unsigned long long seed;  /* Philox seed when noise == NULL                          */
    int    chain_offset;      /* index of this call's first chain among ALL chains: chain b draws the
                                 Philox stream chain_offset + b, so chains spread over several processes
                                 (sbtv.dist.split_chains) never repeat a stream                     */
    int    iter_offset;       /* resume (no counterpart in the reference; SURVEY.md section 5 checkpoint / resume): SAPG
                                 iteration ii of this call is iteration ii + iter_offset of the chain, i.e. its step is
                                 delta(ii + iter_offset).  Continue a chain of S samples with x0 = its last sample,
                                 th_init / p_init / sigma2_init = its last values, warmup = 0, iter_offset = S - 1.  The
                                 Philox steps of a call always count from 0: give a resumed segment its own seed.
                                 0 = the reference's loop.                                          */
} sbtv_sapg_opts;

/*   y: M*N*batch;  x0: start images (NULL -> y, SAPG_algorithm_Guassian.m:10-12)
 *   noise: NULL (device Philox randn) or host/device array with
 *          (warmup-1 + samples-1) * batch * M*N doubles, step-major, consumed
 *          in the reference's order (warm-up first)
 *   traces (host, may be NULL): thetas, sigmas [batch*samples]; ps [batch*2*samples];
 *          logpi [batch*samples]; logpi_wu [batch*warmup]; gx [batch*samples]; grads [batch*4*samples]
 *   eb[batch*4] : theta_EB, p0_EB, p1_EB, sigma2_EB ;  x_last: last sample (may be NULL)
 *   reduce_fn: when share_gradients=1 and the chains are spread over several
 *          processes, called once per iteration with (user, buf, n = 6) to SUM buf[n]
 *          across processes in place (e.g. an RCCL all-reduce); may be NULL.
 *          buf = {sum G_theta, sum G_p0, sum G_p1, sum G_sigma, chains, failed ranks}: a rank whose iteration
 *          failed locally still calls reduce_fn (with failed = 1) before it returns its error, and every other
 *          rank then returns SBTV_ERR_PEER, so no rank is left waiting inside the collective.  Every rank must
 *          run at least one chain (batch >= 1).
 *   The loop itself (SAPG_algorithm_Guassian.m:98-248) is device-resident: the gradients G_theta, G_p, G_sigma
 *          (:165-188), the projected updates (:166-194), the PSF taps of the new parameters and the traces are
 *          computed by a one-workgroup kernel at the end of every iteration, so the host enqueues iterations without
 *          waiting for them (one synchronisation per 1024 iterations and at the end).  A host `reduce_fn` needs the
 *          scalars on the host and therefore selects the host-side loop (as does SBTV_SAPG_HOST_LOOP): identical
 *          arithmetic for theta / p / sigma; the PSF taps then come from the host's libm instead of the device's
 *          (differences in the last bit of exp / pow).  With SBTV_REDUCE_DEVICE `reduce_fn` must be an
 *          sbtv_allreduce_dev_fn: it is called once per iteration with the DEVICE address of the same 6 doubles and
 *          the library's stream and must enqueue an in-place SUM over the processes that is ordered after the work
 *          already in that stream and before work enqueued later (e.g. ncclAllReduce / torch.distributed.all_reduce
 *          on that stream); it must not wait for the GPU.  In that mode the host runs ahead of the device, so a rank whose
 *          iteration fails locally keeps calling reduce_fn once per remaining iteration with {0, 0, 0, 0, 0 chains,
 *          1 failed} before it returns its error; its peers latch the flag on the device and return SBTV_ERR_PEER at
 *          their next synchronisation (every 1024 iterations and at the end).  Only a failure of reduce_fn itself makes
 *          a rank return at once. */
typedef int (*sbtv_allreduce_fn)(void *user, double *buf, int n);
typedef int (*sbtv_allreduce_dev_fn)(void *user, double *dev_buf, int n, void *hip_stream);
int sbtv_SAPG_algorithm(sbtv_ctx *ctx, const double *y, int M, int N, int batch,
                        const sbtv_sapg_opts *op, const double *x0, const double *noise,
                        double *thetas, double *ps, double *sigmas, double *logpi,
                        double *logpi_wu, double *gx, double *grads, double *eb,
                        double *x_last, sbtv_allreduce_fn reduce_fn, void *reduce_user, int flags);

/* Plain MYULA chain at fixed parameters: replaces  xMAP = myula(op, im)  (SALSA/myula.m:1-22) with the closures
 * of SALSA/run_deblur_tv.m:126,131:  proxG(x,lambda,theta) = chambolle_prox_TV_stop(x,'lambda',lambda*theta,
 * 'maxiter',chambolleit),  gradF(x) = AT(A x - y)/sigma2.  x starts at y; samples-2 steps
 *   x = (1 - gamma/lambda) x - gamma (gradF(x) - prox/lambda) + sqrt(2 gamma) z        (:16, no abs())
 * theta[batch], sigma2[batch] host arrays; noise: NULL (device Philox, stream chain_offset + b) or
 * (samples-2)*batch*M*N doubles, step-major.  x_out: the last sample of every chain. */
int sbtv_myula(sbtv_ctx *ctx, const double *y, int M, int N, int batch, const double *taps, int taille,
               double lambda, double gamma, const double *theta, const double *sigma2, int samples,
               int chambolleit, unsigned long long seed, int chain_offset, const double *noise,
               double *x_out, int flags);

/* ---- a-9: largest eigenvalue of A'A by power iteration --------------------
 * Replaces max_eigenval(A,At,params,im_size,tol,max_iter,verbose)
 * (utils/max_eigenval_Gaussian_Moffat.m:1-27, max_eigenval_Laplace.m:1-28).
 * x0: start vector (the reference draws randn; MATLAB's stream is unpinned). */
int sbtv_max_eigenval(sbtv_ctx *ctx, const double *taps, int taille, const double *x0,
                      int M, int N, double tol, int max_iter, double *val, int *iters, int flags);

/* ---- a-10: metrics ----------------------------------------------------------
 * sbtv_PSNR: utils/PSNR.m:2-4 ; sbtv_MSE: utils/MSE.m:1-4 (dB). out[batch]. */
int sbtv_PSNR(sbtv_ctx *ctx, const double *x_true, const double *x, int M, int N, int batch, double *out, int flags);
int sbtv_MSE(sbtv_ctx *ctx, const double *x_true, const double *x, int M, int N, int batch, double *out, int flags);

/* ---- several GPUs behind ONE host process ---------------------------------------------
 * The reference's host is a single MATLAB process (run_Gaussian_demo.m:199 calls the SAPG loop, :229-242 SALSA_v2);
 * a group gives such a host all the GPUs of a node without a second process: one context and one host thread per
 * entry of `devices` (an ordinal may repeat: "virtual shards" on one GPU).  Items are dealt to the shards in
 * contiguous blocks (sbtv_group_shard_of); only min(n, n_items) shards take part in a call.  All pointers are HOST
 * pointers, laid out exactly as for the single-context entry points with batch = n_items.
 *   sbtv_fista_tv_sharded, sbtv_CSALSA_v2_sharded, sbtv_CoRAL_v2_sharded: independent images, no exchange
 *                                (SALSA/my_fista.m:5, SALSA/CSALSA_v2.m:160, SALSA/CoRAL_v2.m:2); arguments as for the
 *                                single-context entry points.
 *   sbtv_SALSA_v2_sharded        independent images: no exchange; image k is computed bit for bit as by sbtv_SALSA_v2.
 *   sbtv_SAPG_algorithm_sharded  op->share_gradients = 0: independent images / chains (chain i draws the Philox stream
 *                                op->chain_offset + i whatever the sharding).  share_gradients = 1: n_items MYULA chains
 *                                on ONE image y (SAPG_algorithm_moffat.m:143-173, `G = mean(g_*)`), the six gradient sums
 *                                added up across the shards once per iteration by an in-process, in-stream exchange
 *                                (pinned peer-visible slots + events; no GPU ever waits for its host, no RCCL).  A shard
 *                                that fails keeps the exchange in step until the end of the loop; the others return
 *                                SBTV_ERR_PEER and the call returns the failing shard's status (sbtv_group_last_error).
 */
typedef struct sbtv_group sbtv_group;
int         sbtv_group_create(const int *devices, int n, sbtv_group **out);
int         sbtv_group_destroy(sbtv_group *g);
int         sbtv_group_size(const sbtv_group *g);
sbtv_ctx   *sbtv_group_ctx(sbtv_group *g, int i);            /* context of shard i (e.g. for sbtv_last_timing) */
const char *sbtv_group_last_error(const sbtv_group *g);
/* which shard computes `item` of n_items, and that shard's block [first, first + count) */
int         sbtv_group_shard_of(const sbtv_group *g, int n_items, int item, int *shard, int *first, int *count);
int sbtv_SALSA_v2_sharded(sbtv_group *g, const double *y, int M, int N, int n_items,
                          const double *taps, int taille, const double *tau, const double *mu,
                          const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                          double *x_out, double *objective, double *distance, double *times, double *mses,
                          int *numA, int *numAt, int *n_outer);
/* device-resident variant: y[r] / true_x[r] / x_init[r] / x_out[r] (r < min(n, n_items)) are DEVICE pointers on shard r's
 * device to that shard's block of images (sbtv_group_shard_of: first, count), laid out as for sbtv_SALSA_v2 with
 * SBTV_DEVICE_PTRS; true_x and x_init may be NULL.  Nothing is copied between host and devices. */
int sbtv_SALSA_v2_sharded_dev(sbtv_group *g, const double *const *y, int M, int N, int n_items,
                              const double *taps, int taille, const double *tau, const double *mu,
                              const sbtv_salsa_opts *opts, const double *const *true_x,
                              const double *const *x_init, double *const *x_out, double *objective,
                              double *distance, double *times, double *mses, int *numA, int *numAt, int *n_outer);
int sbtv_SAPG_algorithm_sharded(sbtv_group *g, const double *y, int M, int N, int n_items,
                                const sbtv_sapg_opts *op, const double *x0, const double *noise,
                                double *thetas, double *ps, double *sigmas, double *logpi,
                                double *logpi_wu, double *gx, double *grads, double *eb, double *x_last);
int sbtv_fista_tv_sharded(sbtv_group *g, const double *b, int M, int N, int n_items,
                          const double *taps, int taille, const double *tau, double L,
                          int prox_iters, int stopcriterion, double tolerance, int maxiters,
                          int zero_start, const double *true_x, double *x_out,
                          double *objective, double *mses, int *n_iter);
int sbtv_CSALSA_v2_sharded(sbtv_group *g, const double *y, int M, int N, int n_items,
                           const double *taps, int taille, const double *mu1, const double *mu2,
                           const double *sigma, const double *epsilon, double continuationfactor,
                           const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                           double *x_out, double *objective, double *distance1, double *distance2,
                           double *criterion, double *times, double *mses,
                           int *numA, int *numAt, int *n_outer);
int sbtv_CoRAL_v2_sharded(sbtv_group *g, const double *y, int M, int N, int n_items,
                          const double *taps, int taille, const double *tau1, const double *tau2,
                          const double *mu1, const double *mu2, const double *mu_ls, int TViters2,
                          const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                          double *x_out, double *objective, double *distance, double *times, double *mses,
                          int *numA, int *numAt, int *n_outer);

/* ---- diagnostics (no counterpart in the reference; SURVEY.md §5 sanitizer / tracing rows) ----
 * sbtv_diag_canary: with SBTV_CANARY=1 in the environment when the context was created, every device workspace of
 *   the context carries a 256-byte guard band on both sides and every entry point above ends by verifying all of
 *   them (SBTV_ERR_CANARY on damage).  This call verifies on demand: *enabled, number of guarded workspaces, damaged
 *   bytes.  poke = 1 first overwrites the rear guard of one workspace (self-test of the detector) and repairs it.
 * sbtv_diag_solve_stats: cumulative, this context and its lanes: out = {solves repeated with exact Chambolle launches because
 *   the stop rule (chambolle_prox_TV_stop.m:131) fired inside an optimistic prox, switches of a solve from subset error sums
 *   back to full sums, 0, 0}.
 * sbtv_diag_stage_stats: cumulative staging of large pageable host arrays by this context (and its lanes): out = {bytes
 *   host -> device, seconds, bytes device -> host, seconds}.  Arrays of >= 4 MB passed with SBTV_HOST_PTRS move through
 *   four copy lanes (pinned chunks, own streams, SBTV_STAGE_THREADS = 0..4); smaller ones through a plain hipMemcpyAsync.
 * sbtv_diag_prox_variant: which TV-prox kernel a (M, N, batch) problem takes: out = {columns per wave, waves per
 *   workgroup, waves per SIMD requested, rows per lane, tiles per image, 2 = streaming pipeline kernel / 1 = temporally
 *   fused tile kernel / 0 = the one-iteration kernels (odd M, SBTV_SINGLE_STEP)} — lets a parity test assert which kernel it exercised.
 * sbtv_diag_time_pass: times ONE pass of the hot path on scratch data of the given shape with HIP events on the
 *   context stream (`reps` launches after two untimed ones) -> average ms per launch and the algorithmic bytes of one
 *   launch.  pass: 0 forward column FFT of u+bu; 1 row pass with the SALSA spectral solve (forward FFT, operator,
 *   inverse FFT); 2 inverse column FFT fused with the SALSA bookkeeping; 3 plain inverse column FFT; 4 forward row
 *   FFT; 5 / 6 row pass with the SAPG gradient operators (with / without the PSF-parameter sums); 7 warm-started TV
 *   prox of 10 iterations incl. f (one SALSA outer iteration's share); 8 cold TV prox of 25 iterations incl. f;
 *   11..15 ONE fused Chambolle launch of 1..5 warm-started iterations without f and without control kernels (separates
 *   the fixed cost of a launch from the cost of an iteration).
 * sbtv_last_host_stats: how the HOST side of the most recent sbtv_SALSA_v2 call waited for the device (the loop keeps one
 *   iteration queued ahead and polls completion tags in pinned memory): out = {waits, waits that found the scalars at
 *   the first look, waits that went past the spin window and slept, nanosleep calls, stream queries (the 50 ms
 *   fallback), seconds inside the waits, longest wait, seconds spent enqueueing, longest enqueue of one iteration,
 *   the outer iteration the longest wait was for, and what the operating system did to the calling thread during the
 *   loop (getrusage(RUSAGE_THREAD) deltas): voluntary / involuntary context switches, minor / major page faults}.
 * sbtv_diag_workspace: device address / capacity of a named internal workspace of the context, e.g. "salsa.u",
 *   "salsa.bu", "salsa.g": the state the most recent sbtv_SALSA_v2 call left behind (parity tests of intermediate arrays;
 *   with the default one-iteration lag they belong to the last iteration ENQUEUED, which is the stopping iteration
 *   only when the solve ran to MAXITERA).
 * sbtv_diag_switches: the SBTV_* environment switches that are set in this process, as "NAME=value ..." (returns their
 *   number; 0 and an empty string = the default kernels).  They are tuning / A-B hooks, read once per process. */
/* Layout helper for row-major hosts (NumPy, C; MATLAB needs none): dst[b][c][r] = src[b][r][c] for `batch` images of
 * rows x cols doubles - row-major images -> the column-major images every entry point takes, and back with rows / cols swapped. */
int sbtv_host_transpose(const double *src, double *dst, int batch, int rows, int cols);
int sbtv_diag_solve_stats(const sbtv_ctx *ctx, double out[4]);
int sbtv_diag_stage_stats(const sbtv_ctx *ctx, double out[4]);
int sbtv_diag_canary(sbtv_ctx *ctx, int poke, int *enabled, int *nbuf, int *nbad);
int sbtv_last_host_stats(const sbtv_ctx *ctx, double out[14]);
int sbtv_diag_workspace(sbtv_ctx *ctx, const char *name, void **dptr, size_t *bytes);
int sbtv_diag_switches(char *buf, size_t cap);
int sbtv_diag_time_pass(sbtv_ctx *ctx, int pass, int M, int N, int batch, int reps, double *ms_avg, double *alg_bytes);
int sbtv_diag_prox_variant(sbtv_ctx *ctx, int M, int N, int batch, int out[6]);

#ifdef __cplusplus
}
#endif
#endif /* SBTV_H */
