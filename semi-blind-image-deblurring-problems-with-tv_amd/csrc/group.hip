// Several GPUs behind ONE host process (SURVEY.md section 8b `sbtv_ctx_create(devices[], n)`, section 8e).
//
// The reference's host is a single MATLAB process: run_Gaussian_demo.m:199 calls the SAPG loop and :229-242 SALSA_v2, and
// its only in-process seam for more than one sample per update is the vestigial `for jj = 1:1 ... G = mean(g_*)` loop
// (SAPG/SAPG_algorithm_moffat.m:143-173).  A group is n contexts (one per entry of `devices`; entries may repeat, which
// gives "virtual shards" on one GPU) with one host thread each:
//   * independent items (images of a batch, BASELINE configs[3]) are dealt to the shards in contiguous blocks and every
//     shard runs the ordinary single-context entry point on its block - no exchange of any kind, and because a batched
//     call computes image k bit for bit like image k alone, the sharded result equals the single-context one exactly;
//   * MYULA chains on one image with averaged gradients (configs[4]) are dealt the same way and exchange six doubles per
//     SAPG iteration through an IN-PROCESS all-reduce that never synchronises a GPU with its host: every shard's stream
//     publishes its sums into a pinned, peer-visible slot (one tiny kernel), records an event, waits (hipStreamWaitEvent)
//     for the events of the other shards and adds the slots up in fixed shard order (a second tiny kernel).  The host
//     threads only meet at a barrier so that every event has been RECORDED before anybody waits for it; they run ahead of
//     their GPUs like the single-context loop does.  No RCCL is involved (48 bytes per iteration).
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

#include "sbtv_internal.h"

namespace sbtv {

// ---- abortable rendezvous of the shard threads ---------------------------------------------------------------
struct Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, waiting = 0;
    unsigned long gen = 0;
    bool aborted = false;
    void reset(int n_) {
        std::lock_guard<std::mutex> lk(m);
        n = n_;
        waiting = 0;
        aborted = false;
    }
    // true: everybody arrived; false: somebody gave up (abort) - the caller must not wait for its peers any more
    bool arrive() {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        const unsigned long g = gen;
        if (++waiting == n) {
            waiting = 0;
            ++gen;
            cv.notify_all();
            return true;
        }
        cv.wait(lk, [&] { return gen != g || aborted; });
        return gen != g;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(m);
        aborted = true;
        cv.notify_all();
    }
};

constexpr int kRed = 8;       // doubles per slot (6 used: the all-reduce buffer of sbtv_SAPG_algorithm)

__global__ void group_publish_kernel(const double *__restrict__ buf, double *__restrict__ slot, int n) {
    if ((int)threadIdx.x < n) slot[threadIdx.x] = buf[threadIdx.x];
    __threadfence_system();
}
// buf[i] = sum over the shards in shard order 0 .. nshards-1 (every shard computes the same bits)
__global__ void group_sum_kernel(double *__restrict__ buf, const double *__restrict__ slots, int nshards, int n) {
#pragma clang fp contract(off)
    const int i = threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int r = 0; r < nshards; ++r) s += slots[(size_t)r * kRed + i];
    buf[i] = s;
}

}  // namespace sbtv

struct sbtv_group {
    std::vector<int> devices;
    std::vector<sbtv_ctx *> ctxs;
    std::string err;
    // in-process all-reduce state
    double *slots_h = nullptr;                      // pinned [2 parities][n][kRed], portable + coherent
    std::vector<double *> slots_d;                  // the same block as each shard's device sees it
    std::vector<hipEvent_t> ev;                     // [2][n]
    std::vector<long> iter;                         // collectives made by each shard in the current call
    sbtv::Rendezvous rv;
    int active = 0;                                 // shards that take part in the current call
};

namespace sbtv {

struct ShardUser {
    sbtv_group *g;
    int rank;
};

// sbtv_allreduce_dev_fn of a shard: enqueue "sum dev_buf over the active shards" on `stream`, ordered after what is
// already in it; never waits for a GPU
static int group_allreduce(void *user, double *dev_buf, int n, void *stream_v) {
    ShardUser *su = static_cast<ShardUser *>(user);
    sbtv_group *g = su->g;
    const int r = su->rank, ns = g->active;
    hipStream_t st = static_cast<hipStream_t>(stream_v);
    if (n > kRed) return 1;
    if (ns == 1) return 0;
    const int par = (int)(g->iter[r] & 1);
    ++g->iter[r];
    double *mine = g->slots_d[r] + ((size_t)par * g->devices.size() + r) * kRed;
    hipLaunchKernelGGL(group_publish_kernel, dim3(1), dim3(64), 0, st, (const double *)dev_buf, mine, n);
    if (hipGetLastError() != hipSuccess || hipEventRecord(g->ev[(size_t)par * g->devices.size() + r], st) != hipSuccess) {
        g->rv.abort();
        return 1;
    }
    // every shard has recorded its event of this iteration (a wait for an event that was never recorded is a no-op)
    if (!g->rv.arrive()) return 1;
    for (int q = 0; q < ns; ++q)
        if (q != r && hipStreamWaitEvent(st, g->ev[(size_t)par * g->devices.size() + q], 0) != hipSuccess) {
            g->rv.abort();
            return 1;
        }
    hipLaunchKernelGGL(group_sum_kernel, dim3(1), dim3(64), 0, st, dev_buf,
                       (const double *)(g->slots_d[r] + (size_t)par * g->devices.size() * kRed), ns, n);
    if (hipGetLastError() != hipSuccess) {
        g->rv.abort();
        return 1;
    }
    // Slot / event reuse two iterations later is safe without another rendezvous: shard q overwrites its slot of this
    // parity only after its own sum of the NEXT iteration, which waited for every shard's publish of that iteration,
    // which each shard enqueued behind its sum of THIS iteration; and q re-records this event only after the next
    // rendezvous, by which time every shard has enqueued its waits of this one.
    return 0;
}

static int gfail(sbtv_group *g, int code, const std::string &msg) {
    if (g) g->err = msg;
    set_global_error(msg);
    return code;
}

// contiguous blocks: shard r of `ns` gets items [lo, hi)
static inline void block_of(int n_items, int ns, int r, int *lo, int *hi) {
    const int base = n_items / ns, extra = n_items % ns;
    *lo = r * base + (r < extra ? r : extra);
    *hi = *lo + base + (r < extra ? 1 : 0);
}

// run fn(rank) on one host thread per active shard; the first non-zero status (lowest rank) is returned with its message
template <class F>
static int run_shards(sbtv_group *g, int ns, F fn) {
    std::vector<int> rc(ns, 0);
    std::vector<std::thread> th;
    th.reserve(ns);
    bool spawn_failed = false;
    for (int r = 0; r < ns; ++r) {
        try {
            th.emplace_back([&, r] {
                rc[r] = (hipSetDevice(g->devices[r]) == hipSuccess) ? fn(r) : (int)hipErrorInvalidDevice;
                if (rc[r] != 0) g->rv.abort();          // nobody may wait for this shard any more
            });
        } catch (...) {                                 // no thread to be had: nothing may cross the C-ABI as an exception
            spawn_failed = true;
            g->rv.abort();                              // the shards already running must not wait for the missing ones
            break;
        }
    }
    for (auto &t : th) t.join();
    if (spawn_failed) return gfail(g, SBTV_ERR_NOMEM, "sbtv_group: cannot start a host thread per shard");
    // a shard that failed on its own is reported before one that only saw its peer fail
    for (int pass = 0; pass < 2; ++pass)
        for (int r = 0; r < ns; ++r)
            if (rc[r] != 0 && (pass == 1 || rc[r] != SBTV_ERR_PEER)) {
                g->err = "shard " + std::to_string(r) + " (device " + std::to_string(g->devices[r]) + "): " + g->ctxs[r]->err;
                set_global_error(g->err);
                return rc[r];
            }
    return 0;
}

}  // namespace sbtv

using namespace sbtv;

extern "C" {

int sbtv_group_create(const int *devices, int n, sbtv_group **out) {
    if (!out) return gfail(nullptr, SBTV_ERR_BADARG, "sbtv_group_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return gfail(nullptr, SBTV_ERR_BADARG, "sbtv_group_create: 1 <= n <= 64 devices");
    sbtv_group *g = new sbtv_group();
    g->devices.assign(devices, devices + n);
    g->ctxs.assign(n, nullptr);
    g->slots_d.assign(n, nullptr);
    g->ev.assign((size_t)2 * n, nullptr);
    g->iter.assign(n, 0);
    auto init = [&]() -> int {
        for (int r = 0; r < n; ++r) {
            const int rc = sbtv_ctx_create(devices[r], &g->ctxs[r]);
            if (rc != 0) return gfail(g, rc, std::string("sbtv_group_create: ") + sbtv_last_error(nullptr));
        }
        if (hipSetDevice(devices[0]) != hipSuccess ||
            hipHostMalloc((void **)&g->slots_h, sizeof(double) * 2 * n * kRed,
                          hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
            return gfail(g, SBTV_ERR_NOMEM, "sbtv_group_create: cannot allocate the pinned exchange block");
        memset(g->slots_h, 0, sizeof(double) * 2 * n * kRed);
        for (int r = 0; r < n; ++r) {
            void *dp = nullptr;
            if (hipSetDevice(devices[r]) != hipSuccess || hipHostGetDevicePointer(&dp, g->slots_h, 0) != hipSuccess)
                return gfail(g, SBTV_ERR_NODEVICE, "sbtv_group_create: the exchange block is not visible from every device");
            g->slots_d[r] = static_cast<double *>(dp);
            for (int p = 0; p < 2; ++p)
                if (hipEventCreateWithFlags(&g->ev[(size_t)p * n + r], hipEventDisableTiming) != hipSuccess)
                    return gfail(g, SBTV_ERR_NOMEM, "sbtv_group_create: hipEventCreate failed");
        }
        return 0;
    };
    const int rc = init();
    if (rc != 0) {
        const std::string msg = g->err;
        (void)sbtv_group_destroy(g);
        return gfail(nullptr, rc, msg);
    }
    *out = g;
    return 0;
}

int sbtv_group_destroy(sbtv_group *g) {
    if (!g) return 0;
    for (size_t r = 0; r < g->ctxs.size(); ++r) {
        (void)hipSetDevice(g->devices[r]);
        if (g->ctxs[r]) (void)sbtv_ctx_destroy(g->ctxs[r]);
        for (int p = 0; p < 2; ++p)
            if (g->ev[(size_t)p * g->ctxs.size() + r]) (void)hipEventDestroy(g->ev[(size_t)p * g->ctxs.size() + r]);
    }
    if (g->slots_h) (void)hipHostFree(g->slots_h);
    delete g;
    return 0;
}

int sbtv_group_size(const sbtv_group *g) { return g ? (int)g->ctxs.size() : 0; }

sbtv_ctx *sbtv_group_ctx(sbtv_group *g, int i) { return (g && i >= 0 && i < (int)g->ctxs.size()) ? g->ctxs[i] : nullptr; }

const char *sbtv_group_last_error(const sbtv_group *g) { return g ? g->err.c_str() : sbtv_last_error(nullptr); }

int sbtv_group_shard_of(const sbtv_group *g, int n_items, int item, int *shard, int *first, int *count) {
    if (!g || n_items < 1 || item < 0 || item >= n_items) return SBTV_ERR_BADARG;
    const int ns = std::min((int)g->ctxs.size(), n_items);
    for (int r = 0; r < ns; ++r) {
        int lo, hi;
        block_of(n_items, ns, r, &lo, &hi);
        if (item >= lo && item < hi) {
            if (shard) *shard = r;
            if (first) *first = lo;
            if (count) *count = hi - lo;
            return 0;
        }
    }
    return SBTV_ERR_BADARG;
}

int sbtv_SALSA_v2_sharded(sbtv_group *g, const double *y, int M, int N, int n_items, const double *taps, int taille,
                          const double *tau, const double *mu, const sbtv_salsa_opts *opts, const double *true_x,
                          const double *x_init, double *x_out, double *objective, double *distance, double *times,
                          double *mses, int *numA, int *numAt, int *n_outer) {
    return salsa_sharded(g, y, M, N, n_items, taps, taille, tau, mu, opts, true_x, x_init, x_out, objective, distance, times,
                         mses, numA, numAt, n_outer, SBTV_HOST_PTRS);
}

int sbtv_SAPG_algorithm_sharded(sbtv_group *g, const double *y, int M, int N, int n_items, const sbtv_sapg_opts *op,
                                const double *x0, const double *noise, double *thetas, double *ps, double *sigmas,
                                double *logpi, double *logpi_wu, double *gx, double *grads, double *eb, double *x_last) {
    return sapg_sharded(g, y, M, N, n_items, op, x0, noise, thetas, ps, sigmas, logpi, logpi_wu, gx, grads, eb, x_last,
                        SBTV_HOST_PTRS);
}

int sbtv_fista_tv_sharded(sbtv_group *g, const double *b, int M, int N, int n_items, const double *taps, int taille,
                          const double *tau, double L, int prox_iters, int stopcriterion, double tolerance, int maxiters,
                          int zero_start, const double *true_x, double *x_out, double *objective, double *mses, int *n_iter) {
    return fista_sharded(g, b, M, N, n_items, taps, taille, tau, L, prox_iters, stopcriterion, tolerance, maxiters, zero_start,
                         true_x, x_out, objective, mses, n_iter, SBTV_HOST_PTRS);
}

int sbtv_CSALSA_v2_sharded(sbtv_group *g, const double *y, int M, int N, int n_items, const double *taps, int taille,
                           const double *mu1, const double *mu2, const double *sigma, const double *epsilon,
                           double continuationfactor, const sbtv_salsa_opts *opts, const double *true_x,
                           const double *x_init, double *x_out, double *objective, double *distance1, double *distance2,
                           double *criterion, double *times, double *mses, int *numA, int *numAt, int *n_outer) {
    return csalsa_sharded(g, y, M, N, n_items, taps, taille, mu1, mu2, sigma, epsilon, continuationfactor, opts, true_x, x_init,
                          x_out, objective, distance1, distance2, criterion, times, mses, numA, numAt, n_outer, SBTV_HOST_PTRS);
}

int sbtv_CoRAL_v2_sharded(sbtv_group *g, const double *y, int M, int N, int n_items, const double *taps, int taille,
                          const double *tau1, const double *tau2, const double *mu1, const double *mu2, const double *mu_ls,
                          int TViters2, const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                          double *x_out, double *objective, double *distance, double *times, double *mses, int *numA,
                          int *numAt, int *n_outer) {
    return coral_sharded(g, y, M, N, n_items, taps, taille, tau1, tau2, mu1, mu2, mu_ls, TViters2, opts, true_x, x_init, x_out,
                         objective, distance, times, mses, numA, numAt, n_outer, SBTV_HOST_PTRS);
}

// Device-resident images on several GPUs: y[r] (true_x[r], x_init[r], x_out[r]) is a DEVICE pointer on shard r's device to
// that shard's block of images (sbtv_group_shard_of gives first / count), column-major, one after the other.
int sbtv_SALSA_v2_sharded_dev(sbtv_group *g, const double *const *y, int M, int N, int n_items, const double *taps,
                              int taille, const double *tau, const double *mu, const sbtv_salsa_opts *opts,
                              const double *const *true_x, const double *const *x_init, double *const *x_out,
                              double *objective, double *distance, double *times, double *mses, int *numA, int *numAt,
                              int *n_outer) {
    if (!g) return SBTV_ERR_BADARG;
    if (!y || !tau || !mu || !opts || n_items < 1 || M < 2 || N < 2)
        return gfail(g, SBTV_ERR_BADARG, "SALSA_v2_sharded_dev: missing required argument");
    if (!taps) return gfail(g, SBTV_ERR_MISSING_AT, "The function handle for transpose of A is missing");
    const int ns = std::min((int)g->ctxs.size(), n_items);
    g->active = ns;
    g->rv.reset(ns);
    for (int r = 0; r < ns; ++r)
        if (!y[r] || (x_out && !x_out[r])) return gfail(g, SBTV_ERR_BADARG, "SALSA_v2_sharded_dev: a shard's image pointer is NULL");
    const size_t t2 = (size_t)taille * taille, K = (size_t)(opts->maxiter > 0 ? opts->maxiter : 0);
    auto offp = [](auto *p, size_t o) { return p ? p + o : p; };
    return run_shards(g, ns, [&](int r) -> int {
        int lo, hi;
        block_of(n_items, ns, r, &lo, &hi);
        const size_t b = (size_t)lo;
        return sbtv_SALSA_v2(g->ctxs[r], y[r], M, N, hi - lo, taps + b * t2, taille, tau + b, mu + b, opts,
                             true_x ? true_x[r] : nullptr, x_init ? x_init[r] : nullptr, x_out ? x_out[r] : nullptr,
                             offp(objective, b * (K + 1)), offp(distance, b * K), offp(times, b * (K + 1)), offp(mses, b * (K + 1)),
                             offp(numA, b), offp(numAt, b), offp(n_outer, b), SBTV_DEVICE_PTRS);
    });
}

int sbtv_ctx_set_lanes(sbtv_ctx *ctx, int mode) {
    if (!ctx || mode < 0 || mode > 2) return SBTV_ERR_BADARG;
    ctx->lanes_mode = mode;
    return 0;
}

}  // extern "C"

// ================================ the sharded drivers (host or device pointers) ================================
namespace sbtv {

static inline int active_shards(sbtv_group *g, int n_items) {
    const int ns = std::min((int)g->ctxs.size(), n_items);
    g->active = ns;
    g->rv.reset(ns);
    return ns;
}
template <class T>
static inline T *off(T *p, size_t o) { return p ? p + o : p; }

int salsa_sharded(sbtv_group *g, const double *y, int M, int N, int n_items, const double *taps, int taille,
                  const double *tau, const double *mu, const sbtv_salsa_opts *opts, const double *true_x,
                  const double *x_init, double *x_out, double *objective, double *distance, double *times, double *mses,
                  int *numA, int *numAt, int *n_outer, int flags) {
    if (!g) return SBTV_ERR_BADARG;
    if (!y || !tau || !mu || !opts || n_items < 1 || M < 2 || N < 2)
        return gfail(g, SBTV_ERR_BADARG, "SALSA_v2_sharded: missing required argument");
    if (!taps) return gfail(g, SBTV_ERR_MISSING_AT, "The function handle for transpose of A is missing");
    const int ns = active_shards(g, n_items);
    const size_t P = (size_t)M * N, t2 = (size_t)taille * taille, K = (size_t)(opts->maxiter > 0 ? opts->maxiter : 0);
    return run_shards(g, ns, [&](int r) -> int {
        int lo, hi;
        block_of(n_items, ns, r, &lo, &hi);
        const size_t b = (size_t)lo;
        return sbtv_SALSA_v2(g->ctxs[r], y + b * P, M, N, hi - lo, taps + b * t2, taille, tau + b, mu + b, opts,
                             off(true_x, b * P), off(x_init, b * P), off(x_out, b * P), off(objective, b * (K + 1)),
                             off(distance, b * K), off(times, b * (K + 1)), off(mses, b * (K + 1)), off(numA, b),
                             off(numAt, b), off(n_outer, b), flags);
    });
}

int sapg_sharded(sbtv_group *g, const double *y, int M, int N, int n_items, const sbtv_sapg_opts *op, const double *x0,
                 const double *noise, double *thetas, double *ps, double *sigmas, double *logpi, double *logpi_wu,
                 double *gx, double *grads, double *eb, double *x_last, int flags) {
    if (!g) return SBTV_ERR_BADARG;
    if (!y || !op || n_items < 1 || M < 2 || N < 2) return gfail(g, SBTV_ERR_BADARG, "SAPG_algorithm_sharded: missing required argument");
    const int ns = active_shards(g, n_items);
    if (noise && ns > 1 && (flags & SBTV_DEVICE_PTRS))
        return gfail(g, SBTV_ERR_BADARG, "SAPG_algorithm_sharded: injected noise must be a host array (it is re-packed per shard)");
    for (auto &it : g->iter) it = 0;
    const bool shared = op->share_gradients != 0;
    const size_t P = (size_t)M * N, S = (size_t)(op->samples > 0 ? op->samples : 0), W = (size_t)(op->warmup > 0 ? op->warmup : 1);
    const size_t steps = (size_t)(op->warmup > 1 ? op->warmup - 1 : 0) + (S > 0 ? S - 1 : 0);
    std::vector<ShardUser> users(ns);
    return run_shards(g, ns, [&](int r) -> int {
        int lo, hi;
        block_of(n_items, ns, r, &lo, &hi);
        const size_t b = (size_t)lo, nb = (size_t)(hi - lo);
        sbtv_sapg_opts o = *op;
        o.chain_offset = op->chain_offset + lo;          // chain b of a call draws the Philox stream chain_offset + b
        // injected noise is step-major over ALL chains: [step][chain][P] -> this shard's [step][local chain][P]
        std::vector<double> nz;
        const double *nzp = nullptr;
        if (noise) {
            if (ns == 1) {
                nzp = noise;
            } else {
                nz.resize(steps * nb * P);
                for (size_t s = 0; s < steps; ++s)
                    memcpy(nz.data() + s * nb * P, noise + (s * (size_t)n_items + b) * P, sizeof(double) * nb * P);
                nzp = nz.data();
            }
        }
        users[r] = ShardUser{g, r};
        const bool coll = shared && ns > 1;
        return sbtv_SAPG_algorithm(g->ctxs[r], shared ? y : y + b * P, M, N, (int)nb, &o, off(x0, b * P), nzp,
                                   off(thetas, b * S), off(ps, b * 2 * S), off(sigmas, b * S), off(logpi, b * S),
                                   off(logpi_wu, b * W), off(gx, b * S), off(grads, b * 4 * S), off(eb, b * 4),
                                   off(x_last, b * P),
                                   coll ? reinterpret_cast<sbtv_allreduce_fn>(&group_allreduce) : nullptr,
                                   coll ? &users[r] : nullptr, flags | (coll ? SBTV_REDUCE_DEVICE : 0));
    });
}

// my_fista over independent images (SALSA/my_fista.m:5): no exchange
int fista_sharded(sbtv_group *g, const double *bimg, int M, int N, int n_items, const double *taps, int taille,
                  const double *tau, double L, int prox_iters, int stopcriterion, double tolerance, int maxiters,
                  int zero_start, const double *true_x, double *x_out, double *objective, double *mses, int *n_iter,
                  int flags) {
    if (!g) return SBTV_ERR_BADARG;
    if (!bimg || !taps || !tau || !true_x || n_items < 1 || maxiters < 1 || M < 2 || N < 2)
        return gfail(g, SBTV_ERR_BADARG, "fista_tv_sharded: bad arguments (b, taps, tau, true are required)");
    const int ns = active_shards(g, n_items);
    const size_t P = (size_t)M * N, t2 = (size_t)taille * taille, K = (size_t)maxiters;
    return run_shards(g, ns, [&](int r) -> int {
        int lo, hi;
        block_of(n_items, ns, r, &lo, &hi);
        const size_t b = (size_t)lo;
        return sbtv_fista_tv(g->ctxs[r], bimg + b * P, M, N, hi - lo, taps + b * t2, taille, tau + b, L, prox_iters,
                             stopcriterion, tolerance, maxiters, zero_start, true_x + b * P, off(x_out, b * P),
                             off(objective, b * K), off(mses, b * K), off(n_iter, b), flags);
    });
}

// C-SALSA over independent images (SALSA/CSALSA_v2.m:160): no exchange; every trace row has maxiter entries
int csalsa_sharded(sbtv_group *g, const double *y, int M, int N, int n_items, const double *taps, int taille,
                   const double *mu1, const double *mu2, const double *sigma, const double *epsilon,
                   double continuationfactor, const sbtv_salsa_opts *opts, const double *true_x, const double *x_init,
                   double *x_out, double *objective, double *distance1, double *distance2, double *criterion,
                   double *times, double *mses, int *numA, int *numAt, int *n_outer, int flags) {
    if (!g) return SBTV_ERR_BADARG;
    if (!y || !taps || !mu1 || !mu2 || !sigma || !opts || n_items < 1 || M < 2 || N < 2)
        return gfail(g, SBTV_ERR_BADARG, "CSALSA_v2_sharded: missing required argument");
    const int ns = active_shards(g, n_items);
    const size_t P = (size_t)M * N, t2 = (size_t)taille * taille, K = (size_t)(opts->maxiter > 0 ? opts->maxiter : 0);
    return run_shards(g, ns, [&](int r) -> int {
        int lo, hi;
        block_of(n_items, ns, r, &lo, &hi);
        const size_t b = (size_t)lo;
        return sbtv_CSALSA_v2(g->ctxs[r], y + b * P, M, N, hi - lo, taps + b * t2, taille, mu1 + b, mu2 + b, sigma + b,
                              off(epsilon, b), continuationfactor, opts, off(true_x, b * P), off(x_init, b * P),
                              off(x_out, b * P), off(objective, b * K), off(distance1, b * K), off(distance2, b * K),
                              off(criterion, b * K), off(times, b * K), off(mses, b * K), off(numA, b), off(numAt, b),
                              off(n_outer, b), flags);
    });
}

// CoRAL over independent images (SALSA/CoRAL_v2.m:2): no exchange
int coral_sharded(sbtv_group *g, const double *y, int M, int N, int n_items, const double *taps, int taille,
                  const double *tau1, const double *tau2, const double *mu1, const double *mu2, const double *mu_ls,
                  int TViters2, const sbtv_salsa_opts *opts, const double *true_x, const double *x_init, double *x_out,
                  double *objective, double *distance, double *times, double *mses, int *numA, int *numAt, int *n_outer,
                  int flags) {
    if (!g) return SBTV_ERR_BADARG;
    if (!y || !taps || !tau1 || !tau2 || !mu1 || !mu2 || !opts || n_items < 1 || M < 2 || N < 2)
        return gfail(g, SBTV_ERR_BADARG, "CoRAL_v2_sharded: missing required argument");
    const int ns = active_shards(g, n_items);
    const size_t P = (size_t)M * N, t2 = (size_t)taille * taille, K = (size_t)(opts->maxiter > 0 ? opts->maxiter : 0);
    return run_shards(g, ns, [&](int r) -> int {
        int lo, hi;
        block_of(n_items, ns, r, &lo, &hi);
        const size_t b = (size_t)lo;
        return sbtv_CoRAL_v2(g->ctxs[r], y + b * P, M, N, hi - lo, taps + b * t2, taille, tau1 + b, tau2 + b, mu1 + b,
                             mu2 + b, off(mu_ls, b), TViters2, opts, off(true_x, b * P), off(x_init, b * P),
                             off(x_out, b * P), off(objective, b * (K + 1)), off(distance, b * K * 2),
                             off(times, b * (K + 1)), off(mses, b * (K + 1)), off(numA, b), off(numAt, b),
                             off(n_outer, b), flags);
    });
}

// ================================ lanes: two streams behind ONE context ================================
// Same-box table profiles/r04_lanes.md: four 2048^2 images in one call 4 988 -> 5 232 image-iterations/s, sixteen 512^2 images
// 51 500 -> 59 000, SAPG Laplace 8 x 1024^2 6 523 -> 6 794: the launch tails and memory-bound passes of one half run under the
// Chambolle launches of the other; three or four lanes buy nothing more.  Independent items by construction
// (SAPG_algorithm_moffat.m:143-173: chains; the images of a batch), so the split needs no exchange; shared-gradient chains use
// the in-process exchange above (lanes_mode 2 only: one exchange per SAPG iteration couples the two streams; no gain).
::sbtv_group *lanes_group(sbtv_ctx *ctx, int n_items, bool shared) {
    static const int env_mode = [] {
        const char *e = getenv("SBTV_LANES");
        return (e && e[0] >= '0' && e[0] <= '2' && e[1] == '\0') ? e[0] - '0' : -1;
    }();
    if (!ctx || ctx->is_lane || n_items < 2) return nullptr;
    const int mode = env_mode >= 0 ? env_mode : ctx->lanes_mode;
    if (mode == 1 || (shared && mode != 2)) return nullptr;
    if (!ctx->lanes) {
        // two lanes; SBTV_LANE_COUNT = 2..8 for experiments (profiles/r04_lanes.md: what more lanes buy)
        static const int n_lanes = [] {
            const char *e = getenv("SBTV_LANE_COUNT");
            const int v = e ? atoi(e) : 2;
            return v >= 2 && v <= 8 ? v : 2;
        }();
        int dev[8];
        for (int &d : dev) d = ctx->device;
        sbtv_group *g = nullptr;
        if (sbtv_group_create(dev, n_lanes, &g) != 0) return nullptr;       // no lanes: the call runs on ctx itself
        for (sbtv_ctx *c : g->ctxs) c->is_lane = true;
        ctx->lanes = g;
    }
    return ctx->lanes;
}

LaneCall::LaneCall(sbtv_ctx *c, ::sbtv_group *grp) : ctx(c), g(grp) {
    // a stream the caller gave us (sbtv_ctx_set_stream) may still hold the work that produces the inputs
    if (!ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
    for (size_t r = 0; r < g->ctxs.size() && r < 8; ++r) calls0[r] = g->ctxs[r]->calls;
}

int LaneCall::done(int rc, int n_items) {
    const int ns = std::min((int)g->ctxs.size(), n_items);
    double t[4] = {0, 0, 0, 0};
    for (int r = 0; r < ns; ++r) {
        const sbtv_ctx *c = g->ctxs[r];
        int lo, hi;
        block_of(n_items, ns, r, &lo, &hi);
        ctx->calls += c->calls - calls0[r];                  // callcounter.m: one count per operator application and image
        t[0] = std::max(t[0], c->timing[0]);                 // device time of the loop: the longer lane
        t[1] = std::max(t[1], c->timing[1]);
        t[2] += c->timing[2] * (double)(hi - lo) / n_items;  // Chambolle iterations, image-averaged
        t[3] += c->timing[3];
    }
    for (int i = 0; i < 4; ++i) ctx->timing[i] = t[i];
    ctx->hstat = g->ctxs[0]->hstat;
    if (rc != 0) return fail(ctx, rc, g->err);
    return 0;
}

}  // namespace sbtv
