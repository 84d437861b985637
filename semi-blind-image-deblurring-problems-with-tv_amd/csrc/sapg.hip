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
// Blocks q >= 4 (deferred stop rule of the multi-buffer prox, device-resident loop): block 4 + s totals the error
// partials of Chambolle step s into out[4*batch + b*FSTRIDE + s]; the parameter-update kernel applies the rule.
__global__ __launch_bounds__(256) void sapg_collect_kernel(const double *__restrict__ acc, int nrb,
                                                           const double *__restrict__ tvp, int ntv,
                                                           double *__restrict__ out, int batch,
                                                           const double *__restrict__ ppart, int pnblk) {
    __shared__ double red[4];
    const int q = blockIdx.x, b = blockIdx.y;
    if (q >= 4) {
        if (threadIdx.x >= 64) return;
        const int st = q - 4;
        const double tot = mb_step_sum(ppart + ((size_t)b * FSTRIDE + st) * pnblk, pnblk, threadIdx.x);
        if (threadIdx.x == 0) out[4 * (size_t)batch + (size_t)b * FSTRIDE + st] = tot;
        return;
    }
    const double *p = (q < 3) ? acc + ((size_t)b * 3 + q) * nrb : tvp + (size_t)b * ntv;
    const int n = (q < 3) ? nrb : ntv;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(q < 3) ? (size_t)b * 3 + q : 3 * (size_t)batch + b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// images the host has frozen: their prox control block is parked (done = 1)
__global__ void fista_park_kernel(ProxCtrl *__restrict__ ctrl, const int *__restrict__ frozen, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch && frozen[b]) ctrl[b].done = 1;
}

// FISTA scalars of one iteration in one launch: block (q, b): q < 3 rows-kernel accumulators [batch][3][nrb] ->
// out[b*3+q]; q = 3..5 momentum-kernel sums [batch][3][npb] (may be null) -> out[3*batch + b*3 + (q-3)];
// q = 6 periodic-TV partials [batch][ntv] -> out[6*batch + b].  `out` is the device view of pinned host memory.
__global__ __launch_bounds__(256) void fista_collect_kernel(const double *__restrict__ acc, int nrb,
                                                            const double *__restrict__ mom, int npb,
                                                            const double *__restrict__ tvp, int ntv,
                                                            double *__restrict__ out, int batch,
                                                            const double *__restrict__ ppart, int pnblk,
                                                            unsigned long long tags_addr, double seq) {
    // tags [batch][8 + FSTRIDE] (pinned host memory, passed as an integer like the SALSA collector's): tag q (or 8 + s
    // for the step sums) = the iteration whose value `out` now holds; the host polls them instead of synchronising
    double *__restrict__ tags = reinterpret_cast<double *>(tags_addr);
    __shared__ double red[4];
    const int q = blockIdx.x, b = blockIdx.y;
    if (q >= 7) {
        // optimistic prox launches (prox_iterate, spec): block 7 + s totals the error partials of Chambolle step s into
        // out[8*batch + b*FSTRIDE + s]; the host applies the stop rule of chambolle_prox_TV_stop.m:131 over the steps
        const int st = q - 7;
        const double *pp = ppart + ((size_t)b * FSTRIDE + st) * pnblk;
        double a = 0.0;
        constexpr int NB = 8;
        for (int base = 0; base < pnblk; base += 256 * NB) {
            double v[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int i = base + r * 256 + (int)threadIdx.x;
                v[r] = (i < pnblk) ? __hip_atomic_load(pp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) a += v[r];
        }
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            out[8 * (size_t)batch + (size_t)b * FSTRIDE + st] = (red[0] + red[1]) + (red[2] + red[3]);
            if (tags) {
                __threadfence_system();
                __hip_atomic_store(&tags[(size_t)b * (8 + FSTRIDE) + 8 + st], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
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
    if (threadIdx.x == 0) {
        out[o] = (red[0] + red[1]) + (red[2] + red[3]);
        if (tags) {
            __threadfence_system();
            __hip_atomic_store(&tags[(size_t)b * (8 + FSTRIDE) + q], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- device-resident SAPG parameter update (SAPG_algorithm_Guassian.m:165-248 and twins) ------------------------
// State of one chain and the constants of a run; the kernel below is the host-side arithmetic of the round-1 loop moved
// to the device, operation for operation (no FMA contraction: theta / p / sigma traces are bit-identical to the host
// loop for the same scalars), so that no iteration has to wait for the host.
struct SapgChain {
    double theta, p0, p1, sig2, sum_th, sum_p0, sum_p1, sum_s;
};
struct SapgDev {
    int kind, taille, npar, nspec, batch, shared, samples, warmup, burnIn, params_move, fix_p0, fix_p1, fix_sigma;
    double dimX, parseval, lamb, c_theta, c_p0, c_p1, c_sigma, min_th, max_th, p_min0, p_max0, p_min1, p_max1, p_true0,
        p_true1, s_lo, s_hi, sigma2_init, phi, step_base;
    const double *scal;        // [4*batch] totals of the collector: ||AX-y||^2, <dA_p0 X, r>, <dA_p1 X, r> per chain, TV
    SapgChain *chain;          // [batch]
    double *par;               // [taps | d0 | d1] per spectrum set, lam[batch], sigma2[batch], noise step
    const double *delta;       // [samples + 1]: delta(ii) of :55, tabulated by the host (pow)
    int *it;                   // [0] ii of the SAPG iteration in flight, [1] ii of the warm-up iteration in flight
    double *red;               // [6] shared-gradient sums (the all-reduce buffer)
    double *G;                 // [batch*4] per-chain gradients
    double *tr_theta, *tr_p, *tr_sigma, *tr_logpi, *tr_gx, *tr_grads, *tr_wu;   // device traces, layouts of sbtv.h
    // deferred stop rule of the multi-buffer prox (null: the prox applies it itself): control blocks, the step sums the
    // collector left behind the scalars, and the launch geometry (prox_k steps, launch l ran prox_base + (l < prox_extra))
    ProxCtrl *pctrl;
    int prox_k, prox_base, prox_extra;
};
enum { SAPG_PH_GRADS = 1, SAPG_PH_UPDATE = 2, SAPG_PH_WARMUP = 4 };

__global__ __launch_bounds__(256) void sapg_update_kernel(SapgDev u, int phase) {
#pragma clang fp contract(off)
    __shared__ double sf[256], se0[256], se1[256], ssum[3];
    const int B = u.batch, tid = threadIdx.x, t2 = u.taille * u.taille;
    double *lam_d = u.par + (size_t)3 * t2 * u.nspec, *sig_d = lam_d + B, *step_d = sig_d + B;
    if (u.pctrl && (phase & (SAPG_PH_WARMUP | SAPG_PH_GRADS))) {
        // the stop rule of the prox this iteration ran optimistically (chambolle_prox_TV_stop.m:131): one thread per chain
        // books k / err and, if it stopped early, what the redo launch (which follows this kernel) has to repeat
        for (int b = tid; b < B; b += 256) {
            ProxCtrl c = u.pctrl[b];
            if (!c.done) {
                mb_apply_rule(c, u.scal + 4 * (size_t)B + (size_t)b * FSTRIDE, u.prox_k, u.prox_base, u.prox_extra);
                u.pctrl[b] = c;
            }
        }
    }
    if (phase & SAPG_PH_WARMUP) {
        // logPiTrace_WU(ii) of the warm-up iteration that just finished (:85); theta and sigma do not move here
        const int ii = u.it[1];
        for (int b = tid; b < B; b += 256) {
            const SapgChain c = u.chain[b];
            const double resid2 = u.scal[(size_t)b * 3] * u.parseval;
            u.tr_wu[(size_t)b * u.warmup + (ii - 1)] = -resid2 / (2 * c.sig2) - c.theta * u.scal[3 * (size_t)B + b];
        }
        __syncthreads();
        if (tid == 0) {
            u.it[1] = ii + 1;
            *step_d = (double)(ii - 1);          // noise step of warm-up iteration ii+1 (steps count from 0 at ii = 2)
        }
        return;
    }
    const int ii = u.it[0], i0 = ii - 1;
    if (phase & SAPG_PH_GRADS) {
        for (int b = tid; b < B; b += 256) {
            const SapgChain c = u.chain[b];
            const double resid2 = u.scal[(size_t)b * 3] * u.parseval;
            const double tv = u.scal[3 * (size_t)B + b];
            u.G[b * 4 + 0] = u.dimX / c.theta - tv;                                                   // :165
            u.G[b * 4 + 1] = (u.scal[(size_t)b * 3 + 1] * u.parseval) / c.sig2;                       // :170
            u.G[b * 4 + 2] = (u.scal[(size_t)b * 3 + 2] * u.parseval) / c.sig2;                       // :179
            u.G[b * 4 + 3] = resid2 / (2 * c.sig2 * c.sig2) - u.dimX / (2 * c.sig2);                  // :188
            u.tr_logpi[(size_t)b * u.samples + i0] = -resid2 / (2 * c.sig2) - c.theta * tv;           // :207
            u.tr_gx[(size_t)b * u.samples + (i0 - 1)] = tv;                                           // :208
        }
        __threadfence_block();
        __syncthreads();
        if (u.shared && tid == 0) {
            // the chains sample one posterior: G = mean over the chains (SAPG_algorithm_moffat.m:158-173), summed
            // here in chain order; the sums of the other ranks are added by the all-reduce between the two phases
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int b = 0; b < B; ++b) {
                a0 += u.G[b * 4 + 0];
                a1 += u.G[b * 4 + 1];
                a2 += u.G[b * 4 + 2];
                a3 += u.G[b * 4 + 3];
            }
            u.red[0] = a0;
            u.red[1] = a1;
            u.red[2] = a2;
            u.red[3] = a3;
            u.red[4] = (double)B;
            u.red[5] = 0.0;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (!(phase & SAPG_PH_UPDATE)) return;
    // a rank whose iteration failed locally contributes {0, 0, 0, 0, 0 chains, 1}: the flag is latched (red[6] is outside
    // the six reduced doubles) and the host returns SBTV_ERR_PEER when it next looks at the device
    if (u.shared && tid == 0 && u.red[5] != 0.0) u.red[6] = 1.0;
    const double delta = u.delta[ii];
    for (int b = tid; b < B; b += 256) {
        SapgChain c = u.chain[b];
        double Gt, Gp0, Gp1, Gs;
        if (u.shared) {
            Gt = u.red[0] / u.red[4];
            Gp0 = u.red[1] / u.red[4];
            Gp1 = u.red[2] / u.red[4];
            Gs = u.red[3] / u.red[4];
        } else {
            Gt = u.G[b * 4 + 0];
            Gp0 = u.G[b * 4 + 1];
            Gp1 = u.G[b * 4 + 2];
            Gs = u.G[b * 4 + 3];
        }
        const double th_new = fmin(fmax(c.theta + u.c_theta * delta * Gt, u.min_th), u.max_th);       // :166-167
        double q0 = u.fix_p0 ? u.p_true0 : c.p0 - u.c_p0 * delta * Gp0;                               // :171-176
        q0 = fmin(fmax(q0, u.p_min0), u.p_max0);
        double q1 = c.p1;
        if (u.npar > 1) {
            q1 = u.fix_p1 ? u.p_true1 : c.p1 - u.c_p1 * delta * Gp1;                                  // :180-185
            q1 = fmin(fmax(q1, u.p_min1), u.p_max1);
        }
        double s_new = u.fix_sigma ? u.sigma2_init : c.sig2 + u.c_sigma * delta * Gs;                 // :189-194
        s_new = fmin(fmax(s_new, u.s_lo), u.s_hi);
        u.tr_grads[((size_t)b * 4 + 0) * u.samples + i0] = Gt;
        u.tr_grads[((size_t)b * 4 + 1) * u.samples + i0] = Gp0;
        u.tr_grads[((size_t)b * 4 + 2) * u.samples + i0] = Gp1;
        u.tr_grads[((size_t)b * 4 + 3) * u.samples + i0] = Gs;
        u.tr_theta[(size_t)b * u.samples + i0] = th_new;
        u.tr_sigma[(size_t)b * u.samples + i0] = s_new;
        u.tr_p[((size_t)b * 2 + 0) * u.samples + i0] = q0;
        u.tr_p[((size_t)b * 2 + 1) * u.samples + i0] = q1;
        c.theta = th_new;
        c.p0 = q0;
        c.p1 = q1;
        c.sig2 = s_new;
        if (ii >= u.burnIn) {
            c.sum_th += th_new;
            c.sum_s += s_new;
            c.sum_p0 += q0;
            c.sum_p1 += q1;
        }
        u.chain[b] = c;
        lam_d[b] = u.lamb * th_new;          // proxG(x, theta): 'lambda', op.lambda*theta  (run_Gaussian_demo.m:191)
        sig_d[b] = s_new;
    }
    __threadfence_block();
    __syncthreads();
    if (u.params_move && t2 <= 64) {
        // taps and derivative taps of the new PSF parameters (sbtv_psf_taps): one WAVE per spectrum set, four sets at a time
        // (a 7 x 7 mask is 49 lanes; the sums run over the taps in MATLAB's column-major order on lane 0, as below)
        const int w = tid >> 6, lane = tid & 63;
        for (int sset = w; sset < u.nspec; sset += 4) {
            const SapgChain c = u.chain[sset];
            const double pv[3] = {c.p0, u.kind == 2 ? 0.0 : c.p1, u.kind == 0 ? u.phi : 0.0};
            double f = 0.0, e0 = 0.0, e1 = 0.0;
            if (lane < t2) psf_taps_point(u.kind, u.taille, pv, lane, &f, &e0, &e1);
            sf[tid] = f;
            se0[tid] = e0;
            se1[tid] = e1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double a = 0, a0 = 0, a1 = 0;
            if (lane == 0) {
                for (int q = 0; q < t2; ++q) {
                    a += sf[w * 64 + q];
                    a0 += se0[w * 64 + q];
                    a1 += se1[w * 64 + q];
                }
            }
            a = __shfl(a, 0, 64);
            a0 = __shfl(a0, 0, 64);
            a1 = __shfl(a1, 0, 64);
            if (lane < t2) {
                u.par[(size_t)sset * t2 + lane] = f / a;
                u.par[(size_t)t2 * u.nspec + (size_t)sset * t2 + lane] = (e0 * a - f * a0) / (a * a);
                u.par[(size_t)2 * t2 * u.nspec + (size_t)sset * t2 + lane] = (e1 * a - f * a1) / (a * a);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    } else if (u.params_move) {
        // masks above 8 x 8: one spectrum set after the other on the whole workgroup
        for (int sset = 0; sset < u.nspec; ++sset) {
            const SapgChain c = u.chain[sset];
            const double pv[3] = {c.p0, u.kind == 2 ? 0.0 : c.p1, u.kind == 0 ? u.phi : 0.0};
            if (tid < t2) psf_taps_point(u.kind, u.taille, pv, tid, &sf[tid], &se0[tid], &se1[tid]);
            __syncthreads();
            if (tid == 0) {
                double a = 0, a0 = 0, a1 = 0;     // MATLAB sum(k(:)): column-major order
                for (int q = 0; q < t2; ++q) {
                    a += sf[q];
                    a0 += se0[q];
                    a1 += se1[q];
                }
                ssum[0] = a;
                ssum[1] = a0;
                ssum[2] = a1;
            }
            __syncthreads();
            if (tid < t2) {
                const double a = ssum[0];
                u.par[(size_t)sset * t2 + tid] = sf[tid] / a;
                u.par[(size_t)t2 * u.nspec + (size_t)sset * t2 + tid] = (se0[tid] * a - sf[tid] * ssum[1]) / (a * a);
                u.par[(size_t)2 * t2 * u.nspec + (size_t)sset * t2 + tid] = (se1[tid] * a - sf[tid] * ssum[2]) / (a * a);
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        u.it[0] = ii + 1;
        *step_d = u.step_base + (double)(ii - 1);    // noise step of SAPG iteration ii+1
    }
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
    if (sbtv_group *lg = lanes_group(ctx, batch, false)) {       // independent images: two lanes of this context (group.hip)
        LaneCall lc(ctx, lg);
        return lc.done(fista_sharded(lg, bimg, M, N, batch, taps, taille, tau, L, prox_iters, stopcriterion, tolerance, maxiters,
                                     zero_start, true_x, x_out, objective, mses, n_iter, flags), batch);
    }
    FftPlan fp;
    SBTV_TRY(fft_plan(ctx, M, N, batch, &fp));
    ProxPlan pp;
    SBTV_TRY(prox_plan(ctx, M, N, batch, &pp));
    const size_t P = (size_t)M * N, cnt = P * batch;
    const double *bd = nullptr, *td = nullptr;
    SBTV_TRY(stage_in(ctx, "fista.b", bimg, cnt, flags, &bd));
    SBTV_TRY(stage_in(ctx, "fista.true", true_x, cnt, flags, &td));
    // x is double-buffered by iteration parity: the host evaluates the stopping rule one iteration late while the next
    // iteration already runs, and the iterate of a stopping iteration must still be intact then
    double *xb[2] = {nullptr, nullptr}, *y = nullptr, *grad = nullptr, *xfinal = nullptr;
    SBTV_TRY(ws_get_t(ctx, "fista.x", cnt, &xb[1]));
    SBTV_TRY(ws_get_t(ctx, "fista.x2", cnt, &xb[0]));
    double *x = xb[1];                                  // iterate 1 (the start)
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
    // pinned: two slots (iteration parity) of [acc3 | mom3 | tv | pad] per image + the step sums of an optimistic prox,
    // then their completion tags, then the frozen flags for upload; host / device view
    constexpr int FT = 8 + FSTRIDE;
    const size_t slot_n = (size_t)FT * batch;
    double *scal_base_h = nullptr, *scal_base_hd = nullptr;
    int *frozen_h = nullptr;
    {
        void *pz = nullptr, *dp = nullptr;
        SBTV_TRY(pinned_get(ctx, sizeof(double) * 4 * slot_n + sizeof(int) * batch, &pz));
        scal_base_h = static_cast<double *>(pz);
        SBTV_HIP(ctx, hipHostGetDevicePointer(&dp, pz, 0));
        scal_base_hd = static_cast<double *>(dp);
        for (size_t i = 0; i < 2 * slot_n; ++i) scal_base_h[2 * slot_n + i] = 0.0;      // tags: no iteration yet
        frozen_h = reinterpret_cast<int *>(scal_base_h + 4 * slot_n);
        for (int b = 0; b < batch; ++b) frozen_h[b] = 0;
    }
    double *tags_base_h = scal_base_h + 2 * slot_n, *tags_base_hd = scal_base_hd + 2 * slot_n;
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

    // objective(k) = 0.5*||A x - b||^2 + tau*Phi(x) ; mses(k)   (:14-15, :31-33)
    // residual energy (Parseval) and TV partials of x, then ONE collector launch that reduces them (and the
    // momentum-kernel sums when given) straight into pinned host memory
    // Optimistic prox launches (no stop-rule kernels, no redo pass: 6 launches less per iteration); the host applies the
    // rule over the prox_iters step sums when it reads the iteration's scalars and, should it have stopped early, repeats
    // the whole call with exact launches (flag SBTV_FISTA_EXACT_PROX), so the result is always that of the exact rule.
    // (not when a device-resident x_out overlaps an input: frozen images are copied into x_out while the loop runs, and a
    // repeated call would then start from damaged inputs - such a call takes the exact launches from the start)
    auto overlaps_out = [&](const double *p) { return p && x_out && (p < x_out + cnt) && (x_out < p + cnt); };
    const bool out_aliases_input = (flags & SBTV_DEVICE_PTRS) && (overlaps_out(bimg) || overlaps_out(true_x));
    const bool prox_spec = !(flags & SBTV_FISTA_EXACT_PROX) && !out_aliases_input && prox_spec_ok(pp, y, x, prox_iters);
    bool prox_was_spec = false;
    // objective / sums of iterate `xk` of iteration k -> pinned slot k & 1, tagged with k
    auto objective_of_x = [&](const double *xk, int k, const int *frozen, const double *mom_partials) -> int {
        RowsArgs a{};
        a.dir_fwd = 1;
        a.op = OP_RESID;
        a.H = Hs;
        a.Y = Bs;
        a.acc = acc;
        a.frozen = frozen;
        // TVnorm(x) rides on the forward column pass over the same image (no TV launch of its own)
        double *tvp = nullptr;
        int ntv = 0;
        if (fft_cols_tv_ok(fp)) {
            ntv = fft_cols_blocks(fp);
            SBTV_TRY(ws_get_t(ctx, "fista.tvc", (size_t)batch * ntv, &tvp));
        }
        SBTV_TRY(fft_cols_fwd_f(ctx, fp, xk, nullptr, S, frozen, tvp));
        SBTV_TRY(fft_rows(ctx, fp, S, nullptr, a));
        if (!tvp) SBTV_TRY(tvnorm_partials(ctx, xk, M, N, batch, &tvp, &ntv));
        const int slot = k & 1;
        hipLaunchKernelGGL(fista_collect_kernel, dim3(prox_was_spec ? 7 + prox_iters : 7, batch), dim3(256), 0, ctx->stream,
                           (const double *)acc, nrb, mom_partials, npb, (const double *)tvp, ntv, scal_base_hd + slot * slot_n,
                           batch, (const double *)pp.partials, pp.fnblk,
                           (unsigned long long)(uintptr_t)(tags_base_hd + slot * slot_n), (double)k);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    };
    // wait until the collector of iteration k has delivered every scalar: poll the tags in pinned memory (no HIP call in
    // the normal case: a stream query would cost the stream a marker packet); ask the stream only after 50 ms
    auto wait_tags = [&](int k, bool spec) -> int {
        volatile const double *tg = tags_base_h + (size_t)(k & 1) * slot_n;
        const double want = (double)k;
        auto t_begin = std::chrono::steady_clock::now();
        for (unsigned spin = 0;; ++spin) {
            bool ready = true;
            for (int b = 0; b < batch && ready; ++b) {
                for (int i = 0; i < 7 && ready; ++i) ready = (tg[(size_t)b * FT + i] == want);
                for (int i = 0; spec && i < prox_iters && ready; ++i) ready = (tg[(size_t)b * FT + 8 + i] == want);
            }
            if (ready) break;
            if (spin < 200) {
                __builtin_ia32_pause();
                continue;
            }
            struct timespec ts = {0, 5000};
            nanosleep(&ts, nullptr);
            if ((spin & 255) == 0 && std::chrono::steady_clock::now() - t_begin > std::chrono::milliseconds(50)) {
                const hipError_t e = hipStreamQuery(ctx->stream);
                if (e == hipSuccess) {
                    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    break;
                }
                if (e != hipErrorNotReady) return fail_hip(ctx, e, "hipStreamQuery", __FILE__, __LINE__);
                t_begin = std::chrono::steady_clock::now();
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return 0;
    };
    std::vector<double> obj_prev(batch, 0.0);
    std::vector<int> frozen(batch, 0), h_niter(batch, 1);
    SBTV_TRY(objective_of_x(x, 1, nullptr, nullptr));
    SBTV_TRY(pair_sums(ctx, x, td, P, batch, o4));
    {
        std::vector<double> h4((size_t)batch * 4);
        SBTV_HIP(ctx, hipMemcpyAsync(h4.data(), o4, sizeof(double) * 4 * batch, hipMemcpyDeviceToHost, ctx->stream));
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const double *sc = scal_base_h + slot_n;       // slot of iteration 1
        for (int b = 0; b < batch; ++b) {
            const double f0 = 0.5 * (sc[(size_t)b * 3] * parseval) + tau[b] * sc[6 * (size_t)batch + b];
            obj_prev[b] = f0;
            if (objective) objective[(size_t)b * maxiters] = f0;
            if (mses) mses[(size_t)b * maxiters] = h4[(size_t)b * 4] / (double)P;
        }
    }
    // The loop keeps one iteration in flight beyond the one the host is looking at (SBTV_FISTA_LAG=0: none): iteration
    // k + 1 is enqueued before the scalars of iteration k are read, so the GPU never waits for the host.  If iteration k
    // turns out to be an image's last, its iterate is still intact in its half of the double buffer (iteration k + 1
    // wrote the other half), and whatever k + 1 did to that image is ignored.
    static const int lag = [] {
        const char *e = getenv("SBTV_FISTA_LAG");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    double t_enq = 1.0;
    int active = batch;
    bool slot_spec[2] = {false, false};
    // fused gradient step (SBTV_FISTA_FUSED_STEP=0: the two-pass form, for A/B runs)
    static const bool fused_wanted = [] {
        const char *e = getenv("SBTV_FISTA_FUSED_STEP");
        return !(e && e[0] == '0');
    }();
    const bool fused_step = fused_wanted && prox_spec && fft_cols_inv_step_ok(fp);
    if (fused_step) SBTV_TRY(prox_reset(ctx, pp, lam_d, 1.0, prox_iters, 1e-3, 0.249, false, frozen_d));
    auto enqueue = [&](int k) -> int {
        const double t_old = t_enq;
        double *xk = xb[k & 1];
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
            if (fused_step) {
                // the gradient never reaches memory: the inverse column pass applies the step to y from its registers
                // (optimistic prox launches do not consult the control blocks: armed once before the loop)
                SBTV_TRY(fft_cols_inv_step(ctx, fp, S, y, inv_scale, 1.0 / L, frozen_d));
            } else {
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
        }
        // x = Psi(y, tau/L): cold-start Chambolle                        (:26 ; run_moffat_demo.m:181-182)
        prox_was_spec = prox_spec;
        slot_spec[k & 1] = prox_spec;
        SBTV_TRY(prox_iterate(ctx, pp, y, prox_iters, xk, true, prox_spec));
        t_enq = 0.5 * (1 + sqrt(1 + 4 * t_old * t_old));                 // :28
        SBTV_TRY(fista_momentum(ctx, xk, xb[(k - 1) & 1], y, td, (t_old - 1) / t_enq, momp, P, batch, frozen_d));   // :29-30
        SBTV_TRY(objective_of_x(xk, k, frozen_d, momp));
        return 0;
    };
    bool fired = false;
    auto process = [&](int k) -> int {
        SBTV_TRY(wait_tags(k, slot_spec[k & 1]));
        const double *sc = scal_base_h + (size_t)(k & 1) * slot_n;
        if (slot_spec[k & 1]) {
            // cont = (k < MaxIter) & (err > tol)  (chambolle_prox_TV_stop.m:131; tol 1e-3 as armed above)
            for (int b = 0; b < batch && !fired; ++b) {
                if (frozen[b]) continue;
                const double *ps = sc + 8 * (size_t)batch + (size_t)b * FSTRIDE;
                for (int kk = 1; kk < prox_iters && !fired; ++kk) fired = !(sqrt(ps[kk - 1]) > 1e-3 * SPEC_TOL_GUARD);
                // (lower-bound sums of the subset launches: back to the full sums long before they can reach tol^2)
                for (int kk = 1; kk <= prox_iters && !pp.esub_off; ++kk)
                    if (!(ps[kk - 1] > ESUB_MARGIN * 1e-6)) {
                        pp.esub_off = 1;
                        ctx->solve_stats[1] += 1;
                    }
            }
            if (fired) return 0;
        }
        bool changed = false;
        for (int b = 0; b < batch; ++b) {
            if (frozen[b]) continue;
            const double f = 0.5 * (sc[(size_t)b * 3] * parseval) + tau[b] * sc[6 * (size_t)batch + b];
            const double *mom = sc + 3 * (size_t)batch + (size_t)b * 3;
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
                frozen_h[b] = 1;
                --active;
                changed = true;
                SBTV_HIP(ctx, hipMemcpyAsync(xfinal + (size_t)b * P, xb[k & 1] + (size_t)b * P, sizeof(double) * P,
                                             hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        if (changed && active > 0) {
            SBTV_HIP(ctx, hipMemcpyAsync(frozen_d, frozen_h, sizeof(int) * batch, hipMemcpyHostToDevice, ctx->stream));
            if (fused_step) {      // nothing re-arms the control blocks per iteration: park the frozen images' prox
                hipLaunchKernelGGL(fista_park_kernel, dim3((batch + 63) / 64), dim3(64), 0, ctx->stream, pp.ctrl,
                                   (const int *)frozen_d, batch);
                SBTV_HIP(ctx, hipGetLastError());
            }
        }
        return 0;
    };
    {
        int rc = 0, enq = 1, done = 1;
        while (active > 0 && done < maxiters) {
            while (rc == 0 && enq < maxiters && enq - done <= lag && active > 0) rc = enqueue(++enq);
            if (rc != 0) break;
            rc = process(++done);
            if (rc != 0 || fired) break;
        }
        if (rc != 0) {
            (void)hipStreamSynchronize(ctx->stream);
            return rc;
        }
        if (fired) {
            // the stop rule fired inside an optimistic prox: repeat the whole call with exact launches
            SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return sbtv_fista_tv(ctx, bimg, M, N, batch, taps, taille, tau, L, prox_iters, stopcriterion, tolerance, maxiters,
                                 zero_start, true_x, x_out, objective, mses, n_iter, flags | SBTV_FISTA_EXACT_PROX);
        }
    }
    for (int b = 0; b < batch; ++b)
        if (!frozen[b])
            SBTV_HIP(ctx, hipMemcpyAsync(xfinal + (size_t)b * P, xb[h_niter[b] & 1] + (size_t)b * P, sizeof(double) * P,
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
    if (op->iter_offset < 0) return fail(ctx, SBTV_ERR_BADARG, "SAPG_algorithm: iter_offset must be >= 0");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    const int npar = (op->kind == SBTV_PSF_LAPLACE) ? 1 : 2;
    const int shared = op->share_gradients ? 1 : 0;
    const int nspec = shared ? 1 : batch;              // spectra sets (H, D1, D2, Y)
    if (((size_t)M * N) & 1)
        return fail(ctx, SBTV_ERR_SIZE, "this entry point needs an even number of pixels (its element-wise passes move two per lane)");
    // independent chains (SAPG_algorithm_moffat.m:143-173: every chain has its own state) -> two lanes of this context;
    // shared-gradient chains only in lanes_mode 2 (their per-iteration exchange couples the two streams).  Not when the
    // caller reduces across processes itself, forces the host loop, or injects device-resident noise (re-packed per lane).
    if (!reduce_fn && !(flags & (SBTV_SAPG_HOST_LOOP | SBTV_REDUCE_DEVICE)) && !(noise && (flags & SBTV_DEVICE_PTRS))) {
        if (sbtv_group *lg = lanes_group(ctx, batch, shared != 0)) {
            LaneCall lc(ctx, lg);
            return lc.done(sapg_sharded(lg, y, M, N, batch, op, x0, noise, thetas, ps, sigmas, logpi, logpi_wu, gx, grads, eb,
                                        x_last, flags), batch);
        }
    }
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
    // a one-parameter PSF (Laplace) has one derivative spectrum: the second one the gradient pass reads IS the first
    // (same memory: no third spectrum to compute, and its lines are already in the cache when the row pass asks again)
    if (npar > 1) SBTV_TRY(ws_get_t(ctx, "sapg.D2", spec * nspec, &D2s));
    else D2s = D1s;
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

    // ---- where the parameter updates run.  Default: on the device (sapg_update_kernel), the host only enqueues.
    // A host reduce_fn needs the gradients on the host every iteration, which selects the host-side loop below.
    static const bool env_host_loop = [] {
        const char *e = getenv("SBTV_SAPG_HOST");
        return e && e[0] == '1';
    }();
    const bool reduce_dev = reduce_fn && (flags & SBTV_REDUCE_DEVICE);
    const bool dev_loop = !(flags & SBTV_SAPG_HOST_LOOP) && !env_host_loop && (!reduce_fn || reduce_dev);
    if (reduce_dev && !dev_loop)
        return fail(ctx, SBTV_ERR_BADARG, "SAPG_algorithm: SBTV_REDUCE_DEVICE needs the device-resident loop (no SBTV_SAPG_HOST_LOOP)");
    double *scal_d = nullptr, *delta_d = nullptr, *red_d = nullptr, *G_d = nullptr, *tr_d = nullptr;
    SapgChain *chain_d = nullptr;
    int *it_d = nullptr;
    const size_t ntr = (size_t)batch * samples * 10 + (size_t)batch * (warmup > 0 ? warmup : 1);
    if (dev_loop) {
        SBTV_TRY(ws_get_t(ctx, "sapg.scal", (4 + (size_t)FSTRIDE) * batch, &scal_d));      // + the prox step sums
        SBTV_TRY(ws_get_t(ctx, "sapg.delta", (size_t)samples + 1, &delta_d));
        SBTV_TRY(ws_get_t(ctx, "sapg.red", 8, &red_d));
        SBTV_TRY(ws_get_t(ctx, "sapg.G", 4 * (size_t)batch, &G_d));
        SBTV_TRY(ws_get_t(ctx, "sapg.traces", ntr, &tr_d));
        SBTV_TRY(ws_get_t(ctx, "sapg.chain", (size_t)batch, &chain_d));
        SBTV_TRY(ws_get_t(ctx, "sapg.it", 2, &it_d));
    }
    double *scal_out = dev_loop ? scal_d : scal_hd;   // where the collector leaves the scalars of an iteration

    // ---- chain state (host scalars)
    std::vector<double> theta(batch, op->th_init), sig2(batch, op->sigma2_init);
    std::vector<double> p0(batch, op->p_init[0]), p1(batch, npar > 1 ? op->p_init[1] : 0.0);
    std::vector<double> th_hist, s_hist, p_hist;      // running sums for the EB means
    std::vector<double> sum_th(batch, 0.0), sum_s(batch, 0.0), sum_p0(batch, 0.0), sum_p1(batch, 0.0);

    // spectra of the taps and of their parameter derivatives, ONE launch for the two or three sets
    auto spectra = [&]() -> int {
        const double *tp[3] = {taps_d, d0_d, d1_d};
        double2 *up[3] = {Hs, D1s, D2s};
        return psf_spectrum_sets(ctx, fps, tp, taille, up, npar > 1 ? 3 : 2);
    };
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
        SBTV_TRY(spectra());
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
    double *tvc = nullptr;                    // TV partials of X from the forward column pass [batch][fft_cols_blocks]
    const int ntvc = fft_cols_tv_ok(fp) ? fft_cols_blocks(fp) : 0;
    if (ntvc) SBTV_TRY(ws_get_t(ctx, "sapg.tvc", (size_t)batch * ntvc, &tvc));
    // The gradient's only reader is the MYULA step.  On the sizes of the wave-granular column pass its spectrum stays in S
    // and the inverse column pass that would store it runs as part of that step instead (fft_cols_inv_myula: no gradient
    // array, one launch less); SBTV_SAPG_FUSED_MYULA=0 or any other size: inverse pass here, element-wise step later.
    static const bool fuse_wanted = [] {
        const char *e = getenv("SBTV_SAPG_FUSED_MYULA");
        return !(e && e[0] == '0');
    }();
    const bool fuse_myula = fuse_wanted && fft_cols_inv_step_ok(fp);
    bool grad_in_S = false;               // S holds the spectrum of the gradient the next MYULA step needs
    auto gradient_from_S = [&]() -> int {
        if (fuse_myula) {
            grad_in_S = true;
            return 0;
        }
        return fft_cols_inv(ctx, fp, S, grad, inv_scale);
    };
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
        // TVnorm(X) (needed by the collector that follows every operator pass) rides on this column pass
        SBTV_TRY(fft_cols_fwd_f(ctx, fp, X, nullptr, S, nullptr, tvc));
        SBTV_TRY(fft_rows(ctx, fp, S, want_grad ? S : nullptr, a));
        if (want_grad) SBTV_TRY(gradient_from_S());
        return 0;
    };
    // TVnorm(X) partials + ONE collector launch that reduces them together with the accumulators of the last
    // operator pass straight into pinned host memory (no separate reductions, no copy kernel)
    int collect_steps = 0;      // > 0: the collector also totals the step sums of the prox just run (deferred stop rule)
    auto collect_scalars = [&]() -> int {
        double *tvp = tvc;
        int ntv = ntvc;
        if (!tvp) SBTV_TRY(tvnorm_partials(ctx, X, M, N, batch, &tvp, &ntv));      // arbitrary-size path
        hipLaunchKernelGGL(sapg_collect_kernel, dim3(4 + collect_steps, batch), dim3(256), 0, ctx->stream, (const double *)acc, nrb,
                           (const double *)tvp, ntv, scal_out, batch, (const double *)pp.partials, pp.fnblk);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    };
    auto fetch_scalars = [&]() -> int {
        SBTV_TRY(collect_scalars());
        if (dev_loop) SBTV_HIP(ctx, hipMemcpyAsync(scal_h, scal_d, sizeof(double) * 4 * batch, hipMemcpyDeviceToHost, ctx->stream));
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
        const bool fused = grad_in_S;
        grad_in_S = false;
        if (noise) {
            SBTV_TRY(next_noise());
            if (fused) return fft_cols_inv_myula(ctx, fp, S, inv_scale, X, prox, Z, sig_d, gam, lamb, nullptr, &arm);
            return myula_step(ctx, X, prox, grad, Z, sig_d, gam, lamb, P, batch, nullptr, &arm);
        }
        const RngArgs r{op->seed, (unsigned)noise_step, (unsigned)op->chain_offset, in_graph ? step_d : nullptr};
        if (!in_graph) ++noise_step;
        if (fused) return fft_cols_inv_myula(ctx, fp, S, inv_scale, X, prox, nullptr, sig_d, gam, lamb, &r, &arm);
        return myula_step(ctx, X, prox, grad, nullptr, sig_d, gam, lamb, P, batch, &r, &arm);
    };
    // prox = chambolle(X, lambda*theta, cold start); armed: the MYULA step before it has reset the control blocks
    // Inside the device-resident loop the prox runs in the multi-buffer optimistic mode: its Chambolle launches go back to
    // back without stop-rule work in between (in-kernel or as separate kernels that costs 4-5 us per launch), every
    // launch boundary keeps its duals, one small kernel applies the rule over all steps and a (normally empty) redo
    // launch re-runs the steps up to an early stop.  Early stops DO happen here (small lambda*theta: err falls below
    // 1e-3 within the 25 iterations), so unlike SALSA / FISTA this path must handle them in place.
    const bool prox_mb = dev_loop && prox_spec_ok(pp, X, prox, op->chambolleit);
    if (prox_mb) SBTV_TRY(prox_reserve_pairs(ctx, &pp, prox_launches(pp, op->chambolleit) + 2));
    // Deferred rule (default in the device-resident loop; SBTV_SAPG_DEFER=0: rule kernel + redo right after the launches):
    // the prox output is not needed before the NEXT iteration's MYULA step, so the step sums are totalled by blocks of
    // the collector, the rule is applied by the parameter-update kernel (both launched anyway) and only the (normally
    // empty) redo launch follows them - one launch of 7-8 us less per iteration.
    static const bool defer_wanted = [] {
        const char *e = getenv("SBTV_SAPG_DEFER");
        return !(e && e[0] == '0');
    }();
    const bool defer_rule = prox_mb && defer_wanted;
    auto do_prox = [&](bool armed) -> int {
        if (!armed) SBTV_TRY(prox_reset(ctx, pp, lam_d, 1.0, op->chambolleit, 1e-3, 0.249, false, nullptr));
        SBTV_TRY(prox_iterate(ctx, pp, X, op->chambolleit, prox, true, (armed && prox_mb) ? (defer_rule ? 3 : 2) : 0));
        return 0;
    };
    auto prox_redo = [&]() -> int { return prox_iterate(ctx, pp, X, op->chambolleit, prox, true, 4); };

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
            SBTV_TRY(spectra());
            RowsArgs a{};
            a.dir_fwd = 1;
            a.dir_inv = 1;
            a.op = OP_GRADF;
            a.H = Hs;
            a.Y = Ys;
            a.acc = acc;
            a.shared_spec = shared;
            // S still holds colFFT(X): the gradient-sums pass that ended the previous iteration wrote no spectrum, and X
            // has not moved since - no second forward column pass
            SBTV_TRY(fft_rows(ctx, fp, S, S, a));
            SBTV_TRY(gradient_from_S());
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

    if (dev_loop) {
        // ================= device-resident loop: warm-up (:66-93), SAPG iterations (:98-248) =================
        SapgDev u{};
        u.kind = op->kind; u.taille = taille; u.npar = npar; u.nspec = nspec; u.batch = batch; u.shared = shared;
        u.samples = samples; u.warmup = warmup > 0 ? warmup : 1; u.burnIn = op->burnIn; u.params_move = params_move ? 1 : 0;
        u.fix_p0 = op->fix_p[0]; u.fix_p1 = op->fix_p[1]; u.fix_sigma = op->fix_sigma;
        u.dimX = dimX; u.parseval = parseval; u.lamb = lamb; u.c_theta = op->c_theta; u.c_p0 = op->c_p[0]; u.c_p1 = op->c_p[1];
        u.c_sigma = op->c_sigma; u.min_th = op->min_th; u.max_th = op->max_th; u.p_min0 = op->p_min[0]; u.p_max0 = op->p_max[0];
        u.p_min1 = op->p_min[1]; u.p_max1 = op->p_max[1]; u.p_true0 = op->p_true[0]; u.p_true1 = op->p_true[1];
        u.s_lo = fmin(op->sigma2_min, op->sigma2_max); u.s_hi = fmax(op->sigma2_min, op->sigma2_max);
        u.sigma2_init = op->sigma2_init; u.phi = op->phi; u.step_base = (double)(warmup > 0 ? warmup - 1 : 0);
        {
            const int nlp = prox_launches(pp, op->chambolleit);
            u.pctrl = defer_rule ? pp.ctrl : nullptr;
            u.prox_k = op->chambolleit; u.prox_base = op->chambolleit / nlp; u.prox_extra = op->chambolleit % nlp;
        }
        u.scal = scal_d; u.chain = chain_d; u.par = par; u.delta = delta_d; u.it = it_d; u.red = red_d; u.G = G_d;
        const size_t bs = (size_t)batch * samples;
        u.tr_theta = tr_d; u.tr_sigma = tr_d + bs; u.tr_logpi = tr_d + 2 * bs; u.tr_gx = tr_d + 3 * bs;
        u.tr_p = tr_d + 4 * bs; u.tr_grads = tr_d + 6 * bs; u.tr_wu = tr_d + 10 * bs;
        // In-stream collective (SBTV_REDUCE_DEVICE): the host runs up to 1024 iterations ahead of the device, so a rank
        // that fails locally cannot simply return - its peers have already enqueued, or will enqueue, one all-reduce per
        // remaining iteration and would wait for it inside the collective.  Such a rank keeps calling reduce_fn once per
        // remaining iteration with {0, 0, 0, 0, 0 chains, 1 failed} and only then returns its error; the peers latch the
        // flag on the device (red[6]) and return SBTV_ERR_PEER at their next synchronisation.  Only when reduce_fn
        // itself fails does a rank return at once.
        bool collective_done = false, reduce_broken = false;
        int local_rc = 0;
        std::string local_err;
        auto update = [&](int phase) -> int {
            hipLaunchKernelGGL(sapg_update_kernel, dim3(1), dim3(256), 0, ctx->stream, u, phase);
            SBTV_HIP(ctx, hipGetLastError());
            return 0;
        };
        // one iteration's device work; main = SAPG iteration (else warm-up), respec = new PSF spectra + gradF first
        auto enqueue_iteration = [&](bool main, bool respec, bool in_graph) -> int {
            if (respec) {
                SBTV_TRY(spectra());
                RowsArgs a{};
                a.dir_fwd = 1;
                a.dir_inv = 1;
                a.op = OP_GRADF;
                a.H = Hs;
                a.Y = Ys;
                a.acc = acc;
                a.shared_spec = shared;
                // S still holds colFFT(X): the gradient-sums pass that ended the previous iteration wrote no spectrum, and X
                // has not moved since - no second forward column pass
                SBTV_TRY(fft_rows(ctx, fp, S, S, a));
                SBTV_TRY(gradient_from_S());
            }
            SBTV_TRY(myula(in_graph));                                                             // :80-81 / :160-161
            SBTV_TRY(do_prox(true));                                                               // :82 / :162
            SBTV_TRY(operator_pass(main ? !params_move : true));                                   // G_w*, G_s, f  (:170-188)
            collect_steps = defer_rule ? op->chambolleit : 0;
            SBTV_TRY(collect_scalars());                                                           // incl. g(X)  (:165)
            collect_steps = 0;
            if (!main) {
                SBTV_TRY(update(SAPG_PH_WARMUP));
            } else if (!(shared && reduce_dev)) {
                SBTV_TRY(update(SAPG_PH_GRADS | SAPG_PH_UPDATE));
            } else {
                SBTV_TRY(update(SAPG_PH_GRADS));
                if (reinterpret_cast<sbtv_allreduce_dev_fn>(reduce_fn)(reduce_user, red_d, 6, (void *)ctx->stream) != 0) {
                    reduce_broken = true;              // the collective itself failed: nothing left to keep in step with
                    return fail(ctx, SBTV_ERR_BADARG, "SAPG_algorithm: reduce_fn failed");
                }
                collective_done = true;
                SBTV_TRY(update(SAPG_PH_UPDATE));
            }
            // the prox's stop rule has been applied by the update kernel: re-run the steps up to an early stop
            if (defer_rule) SBTV_TRY(prox_redo());
            return 0;
        };
        // graph replay (opt-in, small images): the body takes everything from device memory, so launches need no
        // staging; a run with an in-stream collective launches eagerly (the collective is enqueued by the caller's code)
        bool dev_graph = use_graph && !(shared && reduce_dev);
        auto replay = [&](hipGraphExec_t *exec, bool main, bool *replayed) -> int {
            *replayed = false;
            if (!dev_graph) return 0;
            if (!*exec) {
                if (graph_begin(ctx) != 0 || graph_end(ctx, enqueue_iteration(main, main && params_move, true), exec) != 0) {
                    *exec = nullptr;
                    dev_graph = false;                 // capture unavailable: keep launching eagerly
                    return 0;
                }
            }
            ++noise_step;                              // the device counter advances by itself; keep the host's in step
            SBTV_HIP(ctx, hipGraphLaunch(*exec, ctx->stream));
            *replayed = true;
            return 0;
        };
        // constants and initial chain state
        {
            std::vector<double> dl((size_t)samples + 1, 0.0);
            for (int ii = 2; ii <= samples; ++ii) dl[ii] = op->d_scale * (pow((double)ii + op->iter_offset, -op->d_exp) / dimX);      // :55
            std::vector<SapgChain> ch(batch);
            for (int b = 0; b < batch; ++b) {
                ch[b] = SapgChain{theta[b], p0[b], p1[b], sig2[b], 0.0, 0.0, 0.0, 0.0};
                if (op->burnIn == 1) { ch[b].sum_th = theta[b]; ch[b].sum_s = sig2[b]; ch[b].sum_p0 = p0[b]; ch[b].sum_p1 = p1[b]; }
            }
            const int it0[2] = {2, 2};
            SBTV_HIP(ctx, hipMemcpyAsync(delta_d, dl.data(), sizeof(double) * dl.size(), hipMemcpyHostToDevice, ctx->stream));
            SBTV_HIP(ctx, hipMemcpyAsync(chain_d, ch.data(), sizeof(SapgChain) * batch, hipMemcpyHostToDevice, ctx->stream));
            SBTV_HIP(ctx, hipMemcpyAsync(it_d, it0, sizeof(it0), hipMemcpyHostToDevice, ctx->stream));
            SBTV_HIP(ctx, hipMemsetAsync(tr_d, 0, sizeof(double) * ntr, ctx->stream));
            SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the staging vectors go out of scope
        }
        SBTV_TRY(refresh_spectra());
        SBTV_TRY(upload_lam_sigma(theta));
        if (warmup > 0) {
            SBTV_TRY(do_prox(false));
            SBTV_TRY(operator_pass(true));                       // grad for the first step
            for (int ii = 2; ii <= warmup; ++ii) {
                bool replayed = false;
                if (ii >= 3) SBTV_TRY(replay(&g_warm, false, &replayed));
                if (!replayed) SBTV_TRY(enqueue_iteration(false, false, false));
                if ((ii & 1023) == 0) SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
        } else {
            SBTV_TRY(operator_pass(true));
        }
        // slot 0 of the traces (ii = 1): the one place where the host looks at the scalars
        SBTV_TRY(fetch_scalars());
        std::vector<double> logpi0(batch);
        for (int b = 0; b < batch; ++b) logpi0[b] = log_pi(b, theta[b], sig2[b]);                  // :131
        SBTV_TRY(do_prox(false));                                      // proxGX = proxG(X, thetas(1))   (:134)
        const bool in_stream = shared && reduce_dev;
        if (in_stream) SBTV_HIP(ctx, hipMemsetAsync(red_d, 0, sizeof(double) * 8, ctx->stream));
        auto peer_failed = [&]() -> int {          // after a synchronisation: has any rank reported a failure?
            if (!in_stream) return 0;
            double latch = 0.0;
            SBTV_HIP(ctx, hipMemcpy(&latch, red_d + 6, sizeof(double), hipMemcpyDeviceToHost));
            return latch != 0.0 ? fail(ctx, SBTV_ERR_PEER, "SAPG_algorithm: another rank reported an error through reduce_fn") : 0;
        };
        // test hook: SBTV_TEST_FAIL_SAPG="ii:chain_offset" makes the call whose first chain is `chain_offset` fail locally
        // at SAPG iteration ii (how tests exercise the failure protocol of the in-stream collective)
        int inject_ii = -1;
        if (const char *e = getenv("SBTV_TEST_FAIL_SAPG")) {
            int a = 0, b = 0;
            if (sscanf(e, "%d:%d", &a, &b) == 2 && b == op->chain_offset) inject_ii = a;
        }
        for (int ii = 2; ii <= samples; ++ii) {
            collective_done = false;
            if (local_rc == 0) {
                bool replayed = false;
                int rc = (ii == inject_ii) ? fail(ctx, SBTV_ERR_NOMEM, "SAPG_algorithm: injected failure (SBTV_TEST_FAIL_SAPG)") : 0;
                if (rc == 0 && ii >= 3) rc = replay(&g_main, true, &replayed);
                if (rc == 0 && !replayed) rc = enqueue_iteration(true, params_move && ii > 2, false);
                if (rc != 0) {
                    if (!in_stream || reduce_broken) return rc;
                    local_rc = rc;
                    local_err = ctx->err;
                }
            }
            if (local_rc != 0 && !collective_done) {
                static const double failed_vec[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 1.0};
                if (hipMemcpyAsync(red_d, failed_vec, sizeof(failed_vec), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                    reinterpret_cast<sbtv_allreduce_dev_fn>(reduce_fn)(reduce_user, red_d, 6, (void *)ctx->stream) != 0)
                    break;
            }
            if ((ii & 1023) == 0) {
                SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (local_rc == 0) SBTV_TRY(peer_failed());
            }
        }
        if (local_rc != 0) {
            (void)hipStreamSynchronize(ctx->stream);
            return fail(ctx, local_rc, local_err);
        }
        if (in_stream) {
            SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            SBTV_TRY(peer_failed());
        }
        // ---- traces, EB means (:258-284), last sample
        std::vector<double> tr(ntr);
        std::vector<SapgChain> ch(batch);
        SBTV_HIP(ctx, hipMemcpyAsync(tr.data(), tr_d, sizeof(double) * ntr, hipMemcpyDeviceToHost, ctx->stream));
        SBTV_HIP(ctx, hipMemcpyAsync(ch.data(), chain_d, sizeof(SapgChain) * batch, hipMemcpyDeviceToHost, ctx->stream));
        if (x_last) {
            if (flags & SBTV_DEVICE_PTRS)
                SBTV_HIP(ctx, hipMemcpyAsync(x_last, X, sizeof(double) * cnt, hipMemcpyDeviceToDevice, ctx->stream));
            else
                SBTV_TRY(stage_out_copy(ctx, x_last, X, cnt, flags));
        }
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const double *h_theta = tr.data(), *h_sigma = h_theta + bs, *h_logpi = h_theta + 2 * bs, *h_gx = h_theta + 3 * bs,
                     *h_p = h_theta + 4 * bs, *h_grads = h_theta + 6 * bs, *h_wu = h_theta + 10 * bs;
        for (int b = 0; b < batch; ++b) {
            const size_t o = (size_t)b * samples;
            if (thetas) { memcpy(thetas + o, h_theta + o, sizeof(double) * samples); thetas[o] = op->th_init; }
            if (sigmas) { memcpy(sigmas + o, h_sigma + o, sizeof(double) * samples); sigmas[o] = op->sigma2_init; }
            if (logpi) { memcpy(logpi + o, h_logpi + o, sizeof(double) * samples); logpi[o] = logpi0[b]; }
            if (gx) memcpy(gx + o, h_gx + o, sizeof(double) * samples);
            if (ps) {
                memcpy(ps + 2 * o, h_p + 2 * o, sizeof(double) * 2 * samples);
                ps[2 * o] = op->p_init[0];
                ps[2 * o + samples] = npar > 1 ? op->p_init[1] : 0.0;
            }
            if (grads) memcpy(grads + 4 * o, h_grads + 4 * o, sizeof(double) * 4 * samples);
            if (logpi_wu && warmup > 0) memcpy(logpi_wu + (size_t)b * warmup, h_wu + (size_t)b * warmup, sizeof(double) * warmup);
            if (eb) {
                const double cntm = (double)(samples - op->burnIn + 1);
                eb[(size_t)b * 4 + 0] = ch[b].sum_th / cntm;
                eb[(size_t)b * 4 + 1] = ch[b].sum_p0 / cntm;
                eb[(size_t)b * 4 + 2] = ch[b].sum_p1 / cntm;
                eb[(size_t)b * 4 + 3] = ch[b].sum_s / cntm;
            }
        }
        return canary_epilogue(ctx, 0);
    }

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
            // S still holds colFFT(X): the gradient-sums pass that ended the previous iteration wrote no spectrum, and X
            // has not moved since - no second forward column pass
            SBTV_TRY(fft_rows(ctx, fp, S, S, a));
            SBTV_TRY(gradient_from_S());
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
        const double delta = op->d_scale * (pow((double)ii + op->iter_offset, -op->d_exp) / dimX);     // :55
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
            SBTV_TRY(stage_out_copy(ctx, x_last, X, cnt, flags));
    }
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)nsteps_noise;
    return canary_epilogue(ctx, 0);
}

}  // extern "C"
