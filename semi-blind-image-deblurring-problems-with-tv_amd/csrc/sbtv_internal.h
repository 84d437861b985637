// Internal declarations shared by the libsbtv.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/sbtv.h"

namespace sbtv {

constexpr int kWave = 64;  // CDNA wavefront width

// ---------------------------------------------------------------------------
// per-image control block of one TV prox (device resident)
// ---------------------------------------------------------------------------
struct ProxCtrl {
    int k;         // iterations run so far
    int done;      // 1: stop rule met -> iteration kernels return at once
    int cur;       // which dual ping-pong buffer holds the current (px,py)
    int maxiter;
    int redo;      // >0: the last fused launch over-ran the stop rule; re-run that many steps from `cur`
    int f_valid;   // 1: a fused launch already stored the final f = g - lambda div p
    double err;    // last err (chambolle_prox_TV_stop.m:128)
    double lambda;
    double tol;
    double tau;
};

// ---- stop rule of a multi-buffer optimistic prox (prox_iterate modes 2-4): shared by chambolle_mb_ctrl_kernel (tv.hip)
// and by the SAPG collector / parameter-update kernels, which can apply it on their way (sapg.hip).
// Sum of the error partials of one step by ONE wave, in the fixed order of the control kernels (lane-strided over
// chunks of 1024 tiles, then the xor tree).
__device__ __forceinline__ double mb_step_sum(const double *__restrict__ p, int nblk, int lane) {
    double acc = 0.0;
    for (int q0 = 0; q0 < nblk; q0 += 64 * 16) {
        double v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + r * 64 + lane;
            v[r] = (q < nblk) ? p[q] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += v[r];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    return acc;
}
// cont = (k < MaxIter) & (err > tol)  (chambolle_prox_TV_stop.m:131) over the `total` step sums of a prox whose launch l
// ran base + (l < extra) steps.  Stopped at the last step: the f the last launch wrote stands (f_valid).  Stopped at an
// earlier k: `cur` = the dual buffer of the launch boundary before k, `redo` = the steps from there to k (the redo
// launch re-runs them and rewrites f).
__device__ __forceinline__ void mb_apply_rule(ProxCtrl &c, const double *tots, int total, int base, int extra) {
    int k = total;
    for (int st = 0; st < total; ++st) {
        if (!((st + 1 < total) && (sqrt(tots[st]) > c.tol))) {
            k = st + 1;
            break;
        }
    }
    c.k = k;
    c.err = sqrt(tots[k - 1]);
    c.done = 1;
    if (k == total) {
        c.redo = 0;
        c.f_valid = 1;
    } else {
        int l = 0, start = 0;
        for (;; ++l) {
            const int len = base + (l < extra ? 1 : 0);
            if (k <= start + len) break;
            start += len;
        }
        c.cur = l;                              // dual buffer at the launch boundary before k (0 = the cold start)
        c.redo = k - start;
        c.f_valid = 0;
    }
}

// Scalars one SALSA outer iteration hands back to the host (per image).
struct SalsaScal {
    double resid2;   // ||y - A x||^2 (Parseval)
    double tv_u;     // TVnorm(u)
    double mse_num;  // sum (x-true)^2
    double dist_num; // sum (x-u)^2
    double x2;       // sum x^2
    double u2;       // sum u^2
    double dx2;      // sum (x-xprev)^2  (criterion 2)
    double pad;
};

// How the host side of the most recent solver loop waited for the device (sbtv_last_host_stats)
struct HostStats {
    double waits = 0;           // waits for the scalars of an iteration
    double ready_at_once = 0;   // ... that found them at the first look
    double waits_slept = 0;     // ... that went past the spin window and slept at least once
    double sleeps = 0;          // nanosleep calls
    double stream_queries = 0;  // the 50 ms fallback: the stream was asked
    double wait_s = 0, wait_max_s = 0;         // host time inside the waits: total, longest
    double enqueue_s = 0, enqueue_max_s = 0;   // host time spent enqueueing iterations: total, longest
    double wait_max_outer = 0;  // outer iteration whose scalars the longest wait was for
    // what the operating system did to the calling thread during the loop (getrusage(RUSAGE_THREAD) deltas): a host
    // thread that is taken off its core, or that blocks in a page fault, lets the one-deep queue of a small image run dry
    double nvcsw = 0, nivcsw = 0, minflt = 0, majflt = 0;
};

struct DevBuf {
    void *p = nullptr;       // what the kernels see
    void *base = nullptr;    // what hipMalloc returned (p - guard band in canary mode, else p)
    size_t bytes = 0;        // capacity behind p
};

}  // namespace sbtv

struct sbtv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string err;
    long long calls = 0;
    double timing[4] = {0, 0, 0, 0};
    sbtv::HostStats hstat;
    std::map<std::string, sbtv::DevBuf> ws;   // named device workspaces (grow-only)
    // what the tap spectrum in workspace "salsa.H" was built from (sbtv_SALSA_v2 keeps it across calls with the same taps)
    std::vector<double> salsa_h_taps;
    long long salsa_h_dims[6] = {0, 0, 0, 0, 0, 0};
    const void *salsa_h_ptr = nullptr;
    std::map<int, double2 *> twiddles;         // n -> exp(-2 pi i k / n), k < n
    std::map<int, double2 *> any_axes;         // n -> Bluestein tables of the arbitrary-size path (fft_any.inc)
    bool any_attr_done = false;                // its kernel's dynamic-LDS limit has been raised on this context's device
    void *pinned = nullptr;                    // pinned host staging for scalar read-back
    size_t pinned_bytes = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t loop_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // events of the SALSA loop (created on first use)
    int cu_count = 256;
    // SBTV_CANARY=1 (read when the context is created): every workspace gets a guard band on both sides,
    // verified at the end of each C-ABI call (canary_epilogue)
    // Host <-> device staging of large pageable arrays (ctx.hip `stage_copy`): kStageThreads copy lanes, each with its
    // own stream, two pinned chunks and an event per chunk (created on first use)
    struct StageLane {
        hipStream_t s = nullptr;
        void *pin[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
    } stage[4];
    bool stage_ready = false;
    double stage_stats[4] = {0, 0, 0, 0};     // cumulative: bytes in, seconds in, bytes out, seconds out (sbtv_diag_stage_stats)
    // cumulative (sbtv_diag_solve_stats): solves repeated with exact Chambolle launches because the stop rule fired inside
    // an optimistic prox; switches of a solve from subset error sums back to full sums (ProxPlan::esub_off)
    double solve_stats[4] = {0, 0, 0, 0};
    // Lanes: a batch of independent items (images, chains) is dealt to two internal contexts on this device, one host
    // thread and one stream each, so that the launch tails and the memory-bound passes of one half run under the
    // issue-bound Chambolle launches of the other (csrc/group.hip, `lanes_group`).  Image k is computed bit for bit as
    // without lanes (a batched call computes every image like that image alone).
    struct sbtv_group *lanes = nullptr;        // created on first use, destroyed with the context
    int lanes_mode = 0;                        // 0: independent items only (default), 1: never, 2: shared-gradient chains too
    bool is_lane = false;                      // this context IS a lane: never split again
    bool canary = false;
    bool canary_dirty = true;                  // the descriptor table below is stale
    void *canary_desc = nullptr;               // device: {lo guard, hi guard} per workspace
    int *canary_bad = nullptr;                 // device: bad bytes per workspace
    int canary_n = 0, canary_cap = 0;
};

namespace sbtv {

int fail(sbtv_ctx *ctx, int code, const std::string &msg);
int fail_hip(sbtv_ctx *ctx, hipError_t e, const char *what, const char *file, int line);
void set_global_error(const std::string &msg);

#define SBTV_HIP(ctx, expr)                                                     \
    do {                                                                        \
        hipError_t e__ = (expr);                                                \
        if (e__ != hipSuccess) return sbtv::fail_hip((ctx), e__, #expr, __FILE__, __LINE__); \
    } while (0)

#define SBTV_TRY(expr)             \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != 0) return rc__; \
    } while (0)

// grow-only named workspace on the context's device
int ws_get(sbtv_ctx *ctx, const char *name, size_t bytes, void **out);
template <typename T>
inline int ws_get_t(sbtv_ctx *ctx, const char *name, size_t count, T **out) {
    void *p = nullptr;
    int rc = ws_get(ctx, name, count * sizeof(T), &p);
    *out = static_cast<T *>(p);
    return rc;
}
int pinned_get(sbtv_ctx *ctx, size_t bytes, void **out);
// end of a C-ABI call: with SBTV_CANARY=1 verify every guard band (one tiny kernel + a sync), else return rc
int canary_epilogue(sbtv_ctx *ctx, int rc);
int canary_verify(sbtv_ctx *ctx, int *nbuf, int *nbad, std::string *first_bad);
int twiddle_get(sbtv_ctx *ctx, int n, const double2 **out);

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Stage an image argument: returns a device pointer for `p` (copying from the
// host into workspace `name` when the caller passed host pointers).
int stage_in(sbtv_ctx *ctx, const char *name, const double *p, size_t count, int flags, const double **dev);
// Output staging: device pointer to write into; stage_out copies back if host.
int stage_out_buf(sbtv_ctx *ctx, const char *name, double *p, size_t count, int flags, double **dev);
int stage_out_copy(sbtv_ctx *ctx, double *host, const double *dev, size_t count, int flags);

// 1/d for a divisor of ordinary magnitude (the spectral solve: d = |H|^2 + mu in [mu, 1 + mu]): v_rcp_f64 and one cubic
// correction r0 (1 + e + e^2), e = 1 - d r0, three FMAs, within ~1 ulp of the IEEE quotient (the IEEE division expands
// to ~20 instructions with scaling steps that such divisors never need); the form validated in the fused Chambolle kernel
__device__ __forceinline__ double fast_rcp(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}

// ---- K9: Philox4x32-10 counter-based generator + Box-Muller -> standard normals.
// counter = (pixel-pair index lo, hi, step, chain) ; key = seed.  One call yields 128 random bits =
// two 53-bit uniforms = two normals = one double2 of Z.  (Statistical parity only: MATLAB's
// randn('state',1) stream cannot be reproduced)
__device__ __forceinline__ void philox_round(unsigned &c0, unsigned &c1, unsigned &c2, unsigned &c3, unsigned k0,
                                             unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
    c0 = h1 ^ c1 ^ k0;
    c1 = l1;
    c2 = h0 ^ c3 ^ k1;
    c3 = l0;
}
__device__ __forceinline__ double2 philox_normal_pair(size_t q, unsigned step, unsigned chain, unsigned long long seed) {
#pragma clang fp contract(off)      // the same bits from every translation unit that draws these numbers
    unsigned c0 = (unsigned)q, c1 = (unsigned)(q >> 32), c2 = step, c3 = chain;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const unsigned long long a = ((unsigned long long)c0 << 32) | c1, bb = ((unsigned long long)c2 << 32) | c3;
    const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);    // (0,1)
    const double u2 = ((double)(bb >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    return make_double2(r * c, r * s);
}
// one element of the MYULA step  X = | X + gam (prox - X)/lamb - gam grad/sigma2 + sqrt(2 gam) Z |
// (SAPG_algorithm_Guassian.m:80-81,161), grad = scale * v; ONE definition for the element-wise kernel (scale = 1) and for
// the epilogue of the inverse column pass, no contraction anywhere in it
__device__ __forceinline__ double myula_nocontract(double x, double p, double v, double scale, double z, double gam,
                                                   double lamb, double s2, double sq2g) {
#pragma clang fp contract(off)
    const double g = v * scale;
    return fabs(((x + gam * (p - x) / lamb) - gam * (g / s2)) + sq2g * z);
}
// where the in-kernel generator of the MYULA step takes its counters from
struct RngArgs {
    unsigned long long seed;
    unsigned step, chain0;
    const double *step_dev;     // non-null: the step counter lives in device memory (graph replay)
};

// lets the MYULA step kernel (which runs right before the cold-start prox) re-arm the prox control blocks, i.e. do
// what prox_reset does with keep_cur = false and no frozen images, and so save that launch
struct ProxArm {
    ProxCtrl *ctrl;             // [batch]; nullptr: nothing to arm
    const double *lambda;       // [batch] (device)
    int maxiter;
    double tol, tau;
    const int *frozen;          // optional per-image flag: armed as done (FISTA batches)
};

// ---- lanes (group.hip): the two-stream split of a batch inside ONE context
// the internal group to deal `n_items` independent items (shared = chains that exchange their gradients) to, or nullptr
// when the call runs on ctx itself (one item, lanes switched off, ctx is a lane, or the group cannot be created)
::sbtv_group *lanes_group(sbtv_ctx *ctx, int n_items, bool shared);
// brackets a call that was handed to the lanes: books calls / timings / the error message on the parent context
struct LaneCall {
    sbtv_ctx *ctx;
    ::sbtv_group *g;
    long long calls0[8];
    LaneCall(sbtv_ctx *c, ::sbtv_group *grp);
    int done(int rc, int n_items);
};
int salsa_sharded(::sbtv_group *g, const double *y, int M, int N, int n_items, const double *taps, int taille,
                  const double *tau, const double *mu, const sbtv_salsa_opts *opts, const double *true_x,
                  const double *x_init, double *x_out, double *objective, double *distance, double *times, double *mses,
                  int *numA, int *numAt, int *n_outer, int flags);
int sapg_sharded(::sbtv_group *g, const double *y, int M, int N, int n_items, const sbtv_sapg_opts *op,
                 const double *x0, const double *noise, double *thetas, double *ps, double *sigmas, double *logpi,
                 double *logpi_wu, double *gx, double *grads, double *eb, double *x_last, int flags);
int fista_sharded(::sbtv_group *g, const double *b, int M, int N, int n_items, const double *taps, int taille,
                  const double *tau, double L, int prox_iters, int stopcriterion, double tolerance, int maxiters,
                  int zero_start, const double *true_x, double *x_out, double *objective, double *mses, int *n_iter,
                  int flags);
int csalsa_sharded(::sbtv_group *g, const double *y, int M, int N, int n_items, const double *taps, int taille,
                   const double *mu1, const double *mu2, const double *sigma, const double *epsilon,
                   double continuationfactor, const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                   double *x_out, double *objective, double *distance1, double *distance2, double *criterion,
                   double *times, double *mses, int *numA, int *numAt, int *n_outer, int flags);
int coral_sharded(::sbtv_group *g, const double *y, int M, int N, int n_items, const double *taps, int taille,
                  const double *tau1, const double *tau2, const double *mu1, const double *mu2, const double *mu_ls,
                  int TViters2, const sbtv_salsa_opts *opts, const double *true_x, const double *x_init, double *x_out,
                  double *objective, double *distance, double *times, double *mses, int *numA, int *numAt, int *n_outer,
                  int flags);

// low-latency host waits of the solver loops (poll, then block; ctx.hip)
int wait_event(sbtv_ctx *ctx, hipEvent_t ev);
int wait_stream(sbtv_ctx *ctx);

// hipGraph replay of launch-bound iteration bodies (ctx.hip)
bool graph_wanted(size_t total_px);
int graph_begin(sbtv_ctx *ctx);
int graph_end(sbtv_ctx *ctx, int body_rc, hipGraphExec_t *exec);

// ----------------------------- TV kernels (tv.hip) --------------------------
struct ProxPlan {
    int M, N, batch;
    int tiles_i, tiles_j, nblk;   // blocks per image (one-iteration kernels)
    int ftiles_i, ftiles_j, fnblk; // blocks per image (temporally fused kernel)
    int cj, nw, minw, rpl;        // fused-kernel variant of this plan (columns per wave, waves, waves/SIMD, rows per lane)
    int pipe, nbands, nseg, seglen;   // pipe = 1: streaming pipeline kernel (tv_pipe.inc): bands x column segments
    ProxCtrl *ctrl;               // [batch]
    double *pbuf;                 // [pairs][2 (px,py)][batch][M*N]; pairs = 2 (ping-pong) unless prox_reserve_pairs() asked for more
    int pairs;
    std::string tag;              // workspace name prefix
    double *partials;             // [2 sets][batch][FSTRIDE][fnblk] (or [batch][nblk] for the one-iteration kernels)
    size_t part_stride;           // doubles per set
    unsigned *counters;           // [batch] arrival tickets (in-kernel stop-rule path)
    const int *order;             // [fnblk] workgroup -> tile of the 128-row tile kernel (null: the arithmetic XCD-chunk order)
    // mixed tiling (mix_nfull > 0): tile ids below mix_nfull are 128-row tiles on a grid of mix_nfi tile rows, the others
    // 64-row one-row-per-lane tiles on mix_nhi tile rows starting at image row mix_row0; dispatched last (tv.hip)
    int mix_nfull, mix_nfi, mix_nhi, mix_row0;
    // optimistic launches (prox_iterate spec == 1) sum their error terms over a subset of the pixels (a lower bound of the
    // step sums: enough to PROVE that the stop rule cannot have fired, tv_fused.inc ESUB).  A solver loop sets this to 1 once
    // the step sums it reads come within ESUB_MARGIN of tol^2: from then on its launches sum every pixel again, so that a
    // solve whose prox converges does not take the exact path because of the looser bound
    int esub_off;
};
int prox_plan(sbtv_ctx *ctx, int M, int N, int batch, ProxPlan *plan, const char *tag = "prox");
// (Re)initialise the control blocks: k=0, done=0 and the per-image lambda from a
// DEVICE array `lambda_dev` scaled by `lambda_scale`; keep_cur keeps the
// ping-pong index (warm start from the duals of the previous call).
int prox_reset(sbtv_ctx *ctx, const ProxPlan &pl, const double *lambda_dev, double lambda_scale,
               int maxiter, double tol, double tau, bool keep_cur, const int *frozen);
int prox_zero_duals(sbtv_ctx *ctx, const ProxPlan &pl);
int prox_set_duals(sbtv_ctx *ctx, const ProxPlan &pl, const double *px, const double *py);   // device ptrs
int prox_get_duals(sbtv_ctx *ctx, const ProxPlan &pl, double *px, double *py);               // device ptrs
// Run up to `maxiter` iterations (device-side early exit), no host sync.
// If f_out is given it receives f = g - lambda div p (fused into the last launch where possible).
// cold = true: start from px = py = 0 without reading (or requiring the caller to clear) the dual buffer; the
// control block must have been reset with keep_cur = false.
constexpr int FSMAX = 10;     // most iterations any fused Chambolle kernel runs per LAUNCH
// The after-the-fact check of an optimistic prox compares step sums that were added up in another order than the exact
// launches' (and, with ESUB, over a subset of the pixels) with tol: a sum within this relative band above tol counts as
// "fired" too and takes the exact path, so that the certificate "the rule did not fire" holds in floating point as well
constexpr double SPEC_TOL_GUARD = 1.0 + 1e-12;
constexpr double ESUB_MARGIN = 32.0;    // see ProxPlan::esub_off (step sums move by a few per cent per outer iteration, the host is one behind)
constexpr int FSTRIDE = 32;   // step slots of the error partials [batch][FSTRIDE][nblk]: most iterations of an optimistic prox
// ---- SALSA collector: what it reduces (salsa.hip launches it; the Chambolle kernels can host its blocks, tv_fused.inc)
constexpr int SALSA_TAGS = 8 + FSTRIDE;   // completion tags per image: 8 scalars + FSTRIDE prox step sums
struct Collect {
    const double *acc;      // rows kernel partials: [batch][3][nrb]           -> resid2
    int nrb;
    const double *tvp;      // tvnorm partials [batch][ntv] (initial objective) or null -> post slot 5
    int ntv;
    const double *post;     // post kernel partials [batch][6][npb]
    int npb;
    // optimistic prox launches: error partials [batch][FSTRIDE][pnblk] of the psteps iterations the launches ran
    // without a stop rule (null: the prox ran exactly); blocks 7.. total one step each for the host's rule
    const double *ppart;
    int pnblk, psteps;
    unsigned long long psum_addr;   // [batch][FSTRIDE] step sums (pinned host memory, as an integer like `out`)
    const double *mse0;     // initial objective only: pair_sums output [batch][4], element 0 -> mse_num (else null)
};
// A collector that rides on another kernel's launch: the first `nblocks` workgroups of that grid run its blocks
// (a launch costs ~5 us however little it does; these blocks finish long before the host kernel's own work)
struct SideJob {
    Collect c;
    unsigned long long out_addr, tags_addr;
    const int *frozen;
    double seq;
    int nblocks;            // 0: no side job
};
#ifdef __HIPCC__
#include "collect.inc"
#endif

int prox_iterate(sbtv_ctx *ctx, const ProxPlan &pl, const double *g, int maxiter, double *f_out = nullptr,
                 bool cold = false, int spec = 0, int spec_parity = 0, const SideJob *side = nullptr);
int prox_reserve_pairs(sbtv_ctx *ctx, ProxPlan *pl, int pairs);
int prox_launches(const ProxPlan &pl, int maxiter);
bool prox_spec_ok(const ProxPlan &pl, const double *g, const double *f_out, int maxiter);
// f = g - lambda * div(p)
int prox_finish(sbtv_ctx *ctx, const ProxPlan &pl, const double *g, double *f);
// periodic TV norm of x -> out_dev[batch] (device)
int tvnorm_dev(sbtv_ctx *ctx, const double *x, int M, int N, int batch, double *out_dev);
// same without the final reduction: per-block partials [batch][nblk] (workspace owned)
int tvnorm_partials(sbtv_ctx *ctx, const double *x, int M, int N, int batch, double **partials, int *nblk);

// ----------------------------- FFT / operator (fft.hip) ---------------------
enum SpecOp {
    OP_NONE = 0,      // plain transform
    OP_MUL_H = 1,     // X *= H
    OP_MUL_HC = 2,    // X *= conj(H)
    OP_INVLS = 3,     // X /= (|H|^2 + mu)
    OP_SALSA = 4,     // X = (conj(H) Y + mu S) / (|H|^2+mu);  acc += w |Y - H X|^2
    OP_RESID = 5,     // acc += w |H X - Y|^2 (no write)
    OP_GRAD = 6,      // X = conj(H) (H X - Y) ; acc0 += w|HX-Y|^2 ; acc1/2 += w Re(D1/2 X conj(HX-Y))
    OP_ATA = 7,       // X *= |H|^2
    OP_GRADF = 8,     // X = conj(H) (H X - Y) ; acc0 += w|HX-Y|^2
    OP_CSALSA = 9,    // C-SALSA with its constraint state kept as a spectrum E (admm.hip): W = cs[0] Y + cs[1] E;
                      // X = (conj(H) W + mu S) / (|H|^2+mu); T = H X - Y; E' = T + cs[2] E (stored in place);
                      // acc0 += w|T|^2; acc1 += w|E'|^2; acc2 += w|E' - E|^2
};
struct FftPlan {
    int M, N, batch;
    int n1;                 // M/2  (column transform length, complex)
    int wave;               // 1: wave-granular kernels + tiled spectrum layout (fft_wave.inc), chosen by the image size
    int u_tiled;            // wave mode, workgroup / pipelined row kernels: operator spectra tiled like S, U[(l/4)][k][l%4], k = 0..M/2
    int u_ld;               // wave mode: leading dimension of the row-major operator spectra U[k][l] (N + pad: a
                            // power-of-two row stride would put every workgroup's row on the same memory channels)
    size_t u_img;           // complex elements of ONE operator spectrum (H, Y, D1, D2) per image
    size_t s_img;           // complex elements of the spectrum S of one image (M/2 x N packed; M x N in generic mode)
    int generic;            // 1: arbitrary-size path (fft_any.inc): full complex spectra, n1 = M
    const double2 *tw_n1;   // twiddles of length n1
    const double2 *tw_M;    // twiddles of length M (split step)
    const double2 *tw_N;    // twiddles of length N
};
int fft_plan(sbtv_ctx *ctx, int M, int N, int batch, FftPlan *plan);
// real M x N (x batch) -> packed half spectrum S (M/2 x N complex), column pass only + row pass
// `add` (optional) is added to `x` on load (x + add).
int fft_cols_fwd(sbtv_ctx *ctx, const FftPlan &pl, const double *x, const double *add, double2 *S);
int fft_cols_inv(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale);
// inverse column pass fused with the SALSA bookkeeping pass (bu, g, reductions, TV(u)); see fft.hip
struct ColsPost {
    const double *u = nullptr;
    double *bu = nullptr;
    double *g = nullptr;
    // wave-granular kernel only (sizes of fft_cols_inv_step_ok): bu is READ from bu_in (null: from `bu`, in place) and
    // written to `bu`; skip_x: the transform's output x is used for the bookkeeping but not stored (the SALSA loop can
    // recover its final x as g + bu, see salsa.hip)
    const double *bu_in = nullptr;
    int skip_x = 0;
    const double *tru = nullptr;
    const double *xprev = nullptr;
    double *partials = nullptr;      // [batch][6][fft_cols_blocks]
    // gradient-step mode (fft_cols_inv_step): ystep <- ystep - alpha * x, x itself is not stored
    double *ystep = nullptr;
    double alpha = 0.0;
    // MYULA mode (fft_cols_inv_myula): the transform's output is the gradient AT(AX - y); the chain state X = ystep is
    // advanced in place (SAPG_algorithm_Guassian.m:80-81,161), the gradient is not stored
    const double *mprox = nullptr, *msig2 = nullptr, *mZ = nullptr;
    double mgam = 0.0, mlamb = 0.0, msq2g = 0.0;
    RngArgs mrng = {0ull, 0u, 0u, nullptr};
    ProxArm marm = {nullptr, nullptr, 0, 0.0, 0.0, nullptr};
};
// X <- | X + gam (prox - X)/lamb - gam (scale colIFFT(S))/sigma2 + sqrt(2 gam) Z |  with the gradient still in the
// registers of the inverse column pass (same bits as fft_cols_inv followed by myula_step); Z: injected noise or null
// (then the Philox generator, `rng`); `arm` re-arms the control blocks of the cold prox that follows, as myula_step
// does.  Sizes of the wave-granular column pass only (fft_cols_inv_step_ok).
int fft_cols_inv_myula(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double scale, double *X, const double *prox,
                       const double *Z, const double *sigma2_dev, double gam, double lamb, const RngArgs *rng,
                       const ProxArm *arm);
int fft_cols_blocks(const FftPlan &pl);
// y <- y - alpha * (scale * colIFFT(S)) with the transform's output still in registers (my_fista.m:25: the gradient is
// never written); only for the sizes of the wave-granular column pass (fft_cols_inv_step_ok), same bits as
// fft_cols_inv_f followed by axpy
bool fft_cols_inv_step_ok(const FftPlan &pl);
int fft_cols_inv_step(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *y, double scale, double alpha,
                      const int *frozen);
// x = scale * colIFFT(S) and g = x - b from the column still in registers (sizes of fft_cols_inv_step_ok)
int fft_cols_inv_sub(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale, const double *b, double *g);
int fft_cols_inv_post(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale, const int *frozen,
                      const ColsPost &post);
struct RowsArgs {
    int dir_fwd;            // 1: apply forward row FFT first
    int dir_inv;            // 1: apply inverse row FFT last
    int op;                 // SpecOp
    const double2 *H;       // (M/2+1) x N per image (unpacked rows 0..M/2)
    const double2 *Y;       // same layout (OP_SALSA / OP_RESID / OP_GRAD)
    const double2 *D1, *D2; // derivative spectra (OP_GRAD), may be null
    double2 *E;             // OP_CSALSA: state spectrum, layout of H, read and written in place
    const double *cs;       // OP_CSALSA: three coefficients (device), see SpecOp
    const double *mu;       // per image (device)
    double *acc;            // partial sums [batch][3][nblk_rows]
    const int *frozen;      // optional per-image flag: skip image when set
    int shared_spec;        // 1: one spectrum set shared by the whole batch (parallel chains on one image)
};
int fft_rows(sbtv_ctx *ctx, const FftPlan &pl, const double2 *Sin, double2 *Sout, const RowsArgs &a);
int fft_rows_blocks(const FftPlan &pl);   // number of row blocks per image (acc stride)
bool fft_rows_csalsa_ok(const FftPlan &pl);   // OP_CSALSA exists in the row kernels this plan uses
// unpack packed spectrum S -> full rows 0..M/2 layout ((M/2+1) x N)
int spec_unpack(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double2 *U);
// direct evaluation of the tap spectrum: U[(M/2+1) x N] per image from taps[batch][taille^2] (device)
int psf_spectrum(sbtv_ctx *ctx, const FftPlan &pl, const double *taps_dev, int taille, double2 *U);
int psf_spectrum_sets(sbtv_ctx *ctx, const FftPlan &pl, const double *const *taps_dev, int taille, double2 *const *U, int nsets);

// ----------------------------- elementwise (elementwise.hip) ----------------
// generic deterministic final reduction: out[b*nout + q] = sum_i partials[(b*nout+q)*n + i]
int reduce_partials(sbtv_ctx *ctx, const double *partials, int nvec, int n, double *out);
// up to four such reductions in one launch: dst[q][v] = sum_i src[q][v*n[q] + i], v < nvec[q] (unused jobs: nvec = 0)
struct RedJobs {
    const double *src[4] = {nullptr, nullptr, nullptr, nullptr};
    double *dst[4] = {nullptr, nullptr, nullptr, nullptr};
    int nvec[4] = {0, 0, 0, 0};
    int n[4] = {0, 0, 0, 0};
    int dropped = 0;               // jobs that found all four slots taken: reduce_jobs refuses such a set
    void add(const double *s, int nv, int nn, double *d) {
        for (int q = 0; q < 4; ++q)
            if (nvec[q] == 0) {
                src[q] = s;
                dst[q] = d;
                nvec[q] = nv;
                n[q] = nn;
                return;
            }
        ++dropped;
    }
};
int reduce_jobs(sbtv_ctx *ctx, const RedJobs &jb);
int ew_blocks(size_t P);
// out4_dev[b*4 + {0,1,2,3}] = sum (a-c)^2, sum a^2, sum c^2, max a
int pair_sums(sbtv_ctx *ctx, const double *a, const double *c, size_t P, int batch, double *out4_dev);
int salsa_zero_start(sbtv_ctx *ctx, const double *y, const double *tru, double *x, double *u, double *bu, size_t P, int batch,
                     double scale, double **accp, double **postp, int *nblk);
int axpy(sbtv_ctx *ctx, double *y, const double *gr, double a, size_t Ptot, const ProxArm *arm = nullptr, int batch = 0);
int fista_momentum(sbtv_ctx *ctx, const double *x, const double *xold, double *y, const double *tru, double coef,
                   double *partials, size_t P, int batch, const int *frozen);
// Z == nullptr: the normals are drawn in the kernel from `rng` (the numbers randn_kernel would have stored)
int myula_step(sbtv_ctx *ctx, double *X, const double *prox, const double *grad, const double *Z,
               const double *sigma2_dev, double gam, double lamb, size_t P, int batch, const RngArgs *rng = nullptr,
               const ProxArm *arm = nullptr);
int myula_plain_step(sbtv_ctx *ctx, double *X, const double *prox, const double *grad, const double *Z,
                     const double *sigma2_dev, double gam, double lamb, size_t P, int batch, const RngArgs *rng,
                     const ProxArm *arm);
int fft_cols_fwd_f(sbtv_ctx *ctx, const FftPlan &pl, const double *x, const double *add, double2 *S, const int *frozen,
                   double *tvp = nullptr);
bool fft_cols_tv_ok(const FftPlan &pl);
int fft_cols_inv_f(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale, const int *frozen);

#include "psf_taps.inc"   // psf_taps_point(): PSF formulas shared by host and device

}  // namespace sbtv
