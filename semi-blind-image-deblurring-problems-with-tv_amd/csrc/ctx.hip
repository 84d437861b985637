// Context, workspaces, staging, twiddle tables and the host-side PSF tap builders.
#include <chrono>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <thread>
#include <vector>

#include "sbtv_internal.h"

namespace sbtv {

static std::string g_error;
static std::mutex g_error_mu;

void set_global_error(const std::string &msg) {
    std::lock_guard<std::mutex> lk(g_error_mu);
    g_error = msg;
}

int fail(sbtv_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    set_global_error(msg);
    return code;
}

int fail_hip(sbtv_ctx *ctx, hipError_t e, const char *what, const char *file, int line) {
    std::string m = std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ") at " +
                    file + ":" + std::to_string(line) + " in " + what;
    fail(ctx, (int)e > 0 ? (int)e : 999, m);
    return (int)e > 0 ? (int)e : 999;
}

// ---- workspace guard bands (sanitizer row) -------------------------------------------
// GPU AddressSanitizer is not available on the target pool, so an out-of-bounds store of a kernel is caught
// the poor man's way: with SBTV_CANARY=1 every named workspace is allocated with kGuard pattern bytes in
// front of and behind it, and every C-ABI call ends with one small kernel that checks all of them.
constexpr size_t kGuard = 256;
constexpr unsigned char kGuardByte = 0xA5;
struct CanaryDesc { const unsigned char *lo, *hi; };

__global__ __launch_bounds__(256) void canary_check_kernel(const CanaryDesc *__restrict__ d, int *__restrict__ bad) {
    const CanaryDesc c = d[blockIdx.x];
    const int t = threadIdx.x;      // kGuard == blockDim.x
    const int n = (c.lo[t] != kGuardByte ? 1 : 0) + (c.hi[t] != kGuardByte ? 1 : 0);
    if (n) atomicAdd(&bad[blockIdx.x], n);
}

int canary_verify(sbtv_ctx *ctx, int *nbuf, int *nbad, std::string *first_bad) {
    *nbuf = 0;
    *nbad = 0;
    if (!ctx->canary) return 0;
    std::vector<CanaryDesc> h;
    std::vector<const std::string *> names;
    for (auto &kv : ctx->ws)
        if (kv.second.base) {
            const unsigned char *b = static_cast<const unsigned char *>(kv.second.base);
            h.push_back({b, b + kGuard + kv.second.bytes});
            names.push_back(&kv.first);
        }
    const int n = (int)h.size();
    *nbuf = n;
    if (n == 0) return 0;
    if (n > ctx->canary_cap) {
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->canary_desc) SBTV_HIP(ctx, hipFree(ctx->canary_desc));
        if (ctx->canary_bad) SBTV_HIP(ctx, hipFree(ctx->canary_bad));
        ctx->canary_cap = n + 32;
        SBTV_HIP(ctx, hipMalloc(&ctx->canary_desc, sizeof(CanaryDesc) * ctx->canary_cap));
        SBTV_HIP(ctx, hipMalloc((void **)&ctx->canary_bad, sizeof(int) * ctx->canary_cap));
        ctx->canary_dirty = true;
    }
    if (ctx->canary_dirty || n != ctx->canary_n) {
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SBTV_HIP(ctx, hipMemcpy(ctx->canary_desc, h.data(), sizeof(CanaryDesc) * n, hipMemcpyHostToDevice));
        ctx->canary_n = n;
        ctx->canary_dirty = false;
    }
    SBTV_HIP(ctx, hipMemsetAsync(ctx->canary_bad, 0, sizeof(int) * n, ctx->stream));
    hipLaunchKernelGGL(canary_check_kernel, dim3(n), dim3((unsigned)kGuard), 0, ctx->stream,
                       static_cast<const CanaryDesc *>(ctx->canary_desc), ctx->canary_bad);
    SBTV_HIP(ctx, hipGetLastError());
    std::vector<int> bad(n);
    SBTV_HIP(ctx, hipMemcpyAsync(bad.data(), ctx->canary_bad, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < n; ++q)
        if (bad[q]) {
            if (*nbad == 0 && first_bad) *first_bad = *names[q];
            *nbad += bad[q];
        }
    return 0;
}

int canary_epilogue(sbtv_ctx *ctx, int rc) {
    if (!ctx || !ctx->canary || rc != 0) return rc;
    int nbuf = 0, nbad = 0;
    std::string name;
    SBTV_TRY(canary_verify(ctx, &nbuf, &nbad, &name));
    if (nbad)
        return fail(ctx, SBTV_ERR_CANARY, "guard band of workspace '" + name + "' was overwritten (" +
                                              std::to_string(nbad) + " bytes in all workspaces)");
    return 0;
}

int ws_get(sbtv_ctx *ctx, const char *name, size_t bytes, void **out) {
    DevBuf &b = ctx->ws[name];
    if (b.bytes < bytes || !b.base) {
        if (b.base) {
            SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->canary) {                      // the evidence would be freed with the buffer
                int nbuf = 0, nbad = 0;
                std::string bad;
                SBTV_TRY(canary_verify(ctx, &nbuf, &nbad, &bad));
                if (nbad) return fail(ctx, SBTV_ERR_CANARY, "guard band of workspace '" + bad + "' was overwritten");
            }
            SBTV_HIP(ctx, hipFree(b.base));
            b.p = b.base = nullptr;
            b.bytes = 0;
        }
        // canary mode sizes the buffer tightly (16-byte granule) so that the rear guard starts right behind it
        const size_t want = ctx->canary ? ((bytes + 15) & ~size_t(15)) : ((bytes + 255) & ~size_t(255));
        const size_t guard = ctx->canary ? kGuard : 0;
        SBTV_HIP(ctx, hipMalloc(&b.base, want + 2 * guard));
        b.p = static_cast<unsigned char *>(b.base) + guard;
        b.bytes = want;
        if (ctx->canary) {
            SBTV_HIP(ctx, hipMemsetAsync(b.base, kGuardByte, kGuard, ctx->stream));
            SBTV_HIP(ctx, hipMemsetAsync(static_cast<unsigned char *>(b.p) + want, kGuardByte, kGuard, ctx->stream));
            ctx->canary_dirty = true;
        }
    }
    *out = b.p;
    return 0;
}

int pinned_get(sbtv_ctx *ctx, size_t bytes, void **out) {
    if (ctx->pinned_bytes < bytes) {
        if (ctx->pinned) {
            SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
            SBTV_HIP(ctx, hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_bytes = 0;
        }
        size_t want = (bytes + 4095) & ~size_t(4095);
        // coherent (fine-grained): device writes are visible to the host while the stream is still running (the solver
        // loops poll completion tags in this block)
        SBTV_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocCoherent));
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return 0;
}

// exp(-2 pi i k / n) with octant symmetry so that the exactly representable
// values (1, -i, -1, i, and the +-sqrt(1/2) pairs) are exact / symmetric.
static void unit_root(int k, int n, double *re, double *im) {
    // angle = -2 pi k / n, reduce k into [0, n)
    k %= n;
    if (k < 0) k += n;
    // use long double evaluation on the first octant
    auto eval = [&](long long kk, long double *c, long double *s) {
        // 0 <= kk <= n/8
        long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)kk / (long double)n;
        *c = cosl(a);
        *s = sinl(a);
    };
    long long k8 = (long long)k * 8;
    int oct = (int)(k8 / n);          // octant 0..7 (angle = k/n turns)
    long double c, s;
    long long kk = k;
    switch (oct) {
        case 0: eval(kk, &c, &s); break;                                                  // a
        case 1: { long double cc, ss; eval((long long)n / 4 - kk, &cc, &ss); c = ss; s = cc; } break;      // pi/2 - a'
        case 2: { long double cc, ss; eval(kk - (long long)n / 4, &cc, &ss); c = -ss; s = cc; } break;     // pi/2 + a'
        case 3: { long double cc, ss; eval((long long)n / 2 - kk, &cc, &ss); c = -cc; s = ss; } break;     // pi - a'
        case 4: { long double cc, ss; eval(kk - (long long)n / 2, &cc, &ss); c = -cc; s = -ss; } break;    // pi + a'
        case 5: { long double cc, ss; eval(3 * (long long)n / 4 - kk, &cc, &ss); c = -ss; s = -cc; } break; // 3pi/2 - a'
        case 6: { long double cc, ss; eval(kk - 3 * (long long)n / 4, &cc, &ss); c = ss; s = -cc; } break; // 3pi/2 + a'
        default: { long double cc, ss; eval((long long)n - kk, &cc, &ss); c = cc; s = -ss; } break;        // 2pi - a'
    }
    *re = (double)c;
    *im = (double)(-s);   // exp(-i a) = cos a - i sin a
}

int twiddle_get(sbtv_ctx *ctx, int n, const double2 **out) {
    auto it = ctx->twiddles.find(n);
    if (it != ctx->twiddles.end()) {
        *out = it->second;
        return 0;
    }
    std::vector<double2> h(n);
    for (int k = 0; k < n; ++k) {
        if (n >= 8 && n % 8 == 0) {
            unit_root(k, n, &h[k].x, &h[k].y);
        } else {
            long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
            h[k].x = (double)cosl(a);
            h[k].y = (double)(-sinl(a));
        }
    }
    double2 *d = nullptr;
    SBTV_HIP(ctx, hipMalloc((void **)&d, sizeof(double2) * (size_t)n));
    SBTV_HIP(ctx, hipMemcpy(d, h.data(), sizeof(double2) * (size_t)n, hipMemcpyHostToDevice));
    ctx->twiddles[n] = d;
    *out = d;
    return 0;
}

// ---- staging of PAGEABLE host arrays (what a MATLAB or NumPy host hands over) -------------------------------------
// The runtime's own path for pageable memory pins the caller's pages: 56 GB/s on pages it has pinned before, but 14-37 ms
// for the 403 MB of four FRESHLY allocated 2048^2 images in and out - and a MATLAB host passes fresh arrays every call
// (profiles/r04_hostcall.md).  Arrays of at least one chunk therefore go through up to kStageMaxThreads copy lanes: thread t
// takes chunks t, t + T, ... of kStageChunk bytes and copies each into one of its two pinned chunks while the DMA of the
// other is in flight (own stream, one event per chunk): 9-12 ms for the same arrays, fresh or not (51 GB/s with four lanes,
// 36 / 22 GB/s in / out with one: the host-side memcpy is the bound, so it is spread over several cores).
// SBTV_STAGE_THREADS = 0 restores the plain copy.
constexpr size_t kStageChunk = (size_t)4 << 20;
constexpr int kStageMaxThreads = 4;
static int stage_threads() {
    static const int n = [] {
        const char *e = getenv("SBTV_STAGE_THREADS");
        const int v = e ? atoi(e) : kStageMaxThreads;
        return v < 0 ? 0 : (v > kStageMaxThreads ? kStageMaxThreads : v);
    }();
    return n;
}
static int stage_init(sbtv_ctx *ctx) {
    if (ctx->stage_ready) return 0;
    for (auto &l : ctx->stage) {                       // (a call that failed half-way left what it had made: nothing twice)
        if (!l.s) SBTV_HIP(ctx, hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            if (!l.pin[k]) SBTV_HIP(ctx, hipHostMalloc(&l.pin[k], kStageChunk, hipHostMallocDefault));
            if (!l.ev[k]) SBTV_HIP(ctx, hipEventCreateWithFlags(&l.ev[k], hipEventDisableTiming));
        }
    }
    ctx->stage_ready = true;
    return 0;
}
// dst <- src for `bytes` bytes, to_device: src is pageable host memory and dst device memory, else the reverse.  Returns
// when the data has arrived (the caller's stream is not involved: for a device source it must have been synchronised).
static int stage_copy(sbtv_ctx *ctx, void *dst, const void *src, size_t bytes, bool to_device) {
    // a lane stages its half of a batch while its twin stages the other half: two copy lanes each (eight memcpy threads
    // on the two halves measured slower than four, tools/bench_hostcall.py)
    const int T = ctx->is_lane ? std::min(2, stage_threads()) : stage_threads();
    const auto t0 = std::chrono::steady_clock::now();
    SBTV_TRY(stage_init(ctx));
    const size_t nchunks = (bytes + kStageChunk - 1) / kStageChunk;
    hipError_t errs[kStageMaxThreads];
    for (auto &e : errs) e = hipSuccess;
    auto lane = [&](int t) {
        sbtv_ctx::StageLane &l = ctx->stage[t];
        hipError_t e = hipSetDevice(ctx->device);
        bool used[2] = {false, false};
        auto span = [&](size_t c, size_t *off, size_t *len) {
            *off = c * kStageChunk;
            *len = std::min(kStageChunk, bytes - *off);
        };
        if (to_device) {
            for (size_t c = (size_t)t, i = 0; c < nchunks && e == hipSuccess; c += (size_t)T, ++i) {
                const int k = (int)(i & 1);
                size_t off, len;
                span(c, &off, &len);
                if (used[k]) e = hipEventSynchronize(l.ev[k]);            // the DMA out of this pinned chunk has finished
                if (e != hipSuccess) break;
                memcpy(l.pin[k], (const char *)src + off, len);
                e = hipMemcpyAsync((char *)dst + off, l.pin[k], len, hipMemcpyHostToDevice, l.s);
                if (e == hipSuccess) e = hipEventRecord(l.ev[k], l.s);
                used[k] = true;
            }
        } else {
            // device -> pinned of chunk i+1 is in flight while chunk i moves pinned -> pageable
            size_t off, len;
            if ((size_t)t < nchunks) {
                span((size_t)t, &off, &len);
                e = hipMemcpyAsync(l.pin[0], (const char *)src + off, len, hipMemcpyDeviceToHost, l.s);
                if (e == hipSuccess) e = hipEventRecord(l.ev[0], l.s);
            }
            for (size_t c = (size_t)t, i = 0; c < nchunks && e == hipSuccess; c += (size_t)T, ++i) {
                const int k = (int)(i & 1);
                const size_t cn = c + (size_t)T;
                if (cn < nchunks) {
                    size_t o2, l2;
                    span(cn, &o2, &l2);
                    e = hipMemcpyAsync(l.pin[k ^ 1], (const char *)src + o2, l2, hipMemcpyDeviceToHost, l.s);
                    if (e == hipSuccess) e = hipEventRecord(l.ev[k ^ 1], l.s);
                    if (e != hipSuccess) break;
                }
                e = hipEventSynchronize(l.ev[k]);
                if (e != hipSuccess) break;
                span(c, &off, &len);
                memcpy((char *)dst + off, l.pin[k], len);
            }
        }
        const hipError_t e2 = hipStreamSynchronize(l.s);
        errs[t] = (e != hipSuccess) ? e : e2;
    };
    {
        std::vector<std::thread> th;
        const int nt = (int)std::min<size_t>((size_t)T, nchunks);
        int started = 1;
        try {
            for (int t = 1; t < nt; ++t, ++started) th.emplace_back(lane, t);
        } catch (...) {                                                // no thread to be had: the caller's thread does the rest
        }
        lane(0);                                                       // the calling thread is lane 0
        for (auto &x : th) x.join();
        for (int t = started; t < nt; ++t) lane(t);                    // chunks of the lanes that could not be started
    }
    for (int t = 0; t < kStageMaxThreads; ++t)
        if (errs[t] != hipSuccess) return fail_hip(ctx, errs[t], "stage_copy", __FILE__, __LINE__);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ctx->stage_stats[to_device ? 0 : 2] += (double)bytes;
    ctx->stage_stats[to_device ? 1 : 3] += dt;
    return 0;
}
static inline bool stage_threaded(size_t bytes) { return stage_threads() > 0 && bytes >= kStageChunk; }

int stage_in(sbtv_ctx *ctx, const char *name, const double *p, size_t count, int flags, const double **dev) {
    if (p == nullptr) {
        *dev = nullptr;
        return 0;
    }
    if (flags & SBTV_DEVICE_PTRS) {
        *dev = p;
        return 0;
    }
    double *d = nullptr;
    SBTV_TRY(ws_get_t(ctx, name, count, &d));
    if (stage_threaded(count * sizeof(double))) {
        // the workspace may still be read by work of an earlier call in the stream (entry points end synchronised, so
        // this is a formality) - then the copy lanes write it
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SBTV_TRY(stage_copy(ctx, d, p, count * sizeof(double), true));
    } else {
        SBTV_HIP(ctx, hipMemcpyAsync(d, p, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    *dev = d;
    return 0;
}

int stage_out_buf(sbtv_ctx *ctx, const char *name, double *p, size_t count, int flags, double **dev) {
    if (p != nullptr && (flags & SBTV_DEVICE_PTRS)) {
        *dev = p;
        return 0;
    }
    // host destination (or no destination at all): internal buffer
    return ws_get_t(ctx, name, count, dev);
}

int stage_out_copy(sbtv_ctx *ctx, double *host, const double *dev, size_t count, int flags) {
    if (host == nullptr || (flags & SBTV_DEVICE_PTRS)) return 0;
    if (stage_threaded(count * sizeof(double))) {
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the result is complete
        return stage_copy(ctx, host, dev, count * sizeof(double), false);
    }
    SBTV_HIP(ctx, hipMemcpyAsync(host, dev, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

// ---- host waits inside the solver loops -----------------------------------------------------------------
// One outer iteration of a small image takes tens of microseconds; a host thread that blocks in
// hipEventSynchronize / hipStreamSynchronize is woken much later than that and the GPU queue runs dry.  These
// helpers can poll (hipEventQuery / hipStreamQuery) for up to 2 ms before they fall back to the blocking call.
// Opt-in (SBTV_SPIN=1): measured on MI355X it helps at 512^2 (8 500 vs 6 500 SALSA iterations/s) and hurts at 256^2
// and 1024^2 (profiles/r02_small_sizes.md) - the polling calls compete with the launches of the same thread.
static inline bool spin_enabled() {
    static const bool on = [] {
        const char *e = getenv("SBTV_SPIN");
        return e && e[0] == '1';
    }();
    return on;
}
template <class Q>
static inline hipError_t poll_2ms(Q query) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0;; ++it) {
        const hipError_t e = query();
        if (e != hipErrorNotReady) return e;
        if ((it & 63) == 63 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2e-3)
            return hipErrorNotReady;
    }
}
int wait_event(sbtv_ctx *ctx, hipEvent_t ev) {
    if (spin_enabled()) {
        const hipError_t e = poll_2ms([&] { return hipEventQuery(ev); });
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail_hip(ctx, e, "hipEventQuery", __FILE__, __LINE__);
    }
    SBTV_HIP(ctx, hipEventSynchronize(ev));
    return 0;
}
int wait_stream(sbtv_ctx *ctx) {
    if (spin_enabled()) {
        const hipError_t e = poll_2ms([&] { return hipStreamQuery(ctx->stream); });
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail_hip(ctx, e, "hipStreamQuery", __FILE__, __LINE__);
    }
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- hipGraph replay of launch-bound iteration bodies -------------------------------------------
// Small images make the solver loops latency-bound (20-odd dependent kernels of a few microseconds per
// iteration).  With SBTV_GRAPH=1 the iteration body is captured once from the stream and replayed with one
// hipGraphLaunch: the host thread then issues one call per iteration instead of ~25.  Measured on MI355X
// (512^2 demo): 0.168 vs 0.172 ms per SAPG iteration - the loop is bound by the dependent-kernel latency on
// the GPU, not by host launch cost - so replay is opt-in (it mainly frees the host core when 8 ranks share
// a node).  Results are bit-identical either way (tests/test_gpu_modes.py).
bool graph_wanted(size_t total_px) {
    static const char *e = getenv("SBTV_GRAPH");
    (void)total_px;
    return e && e[0] == '1';
}

int graph_begin(sbtv_ctx *ctx) {
    SBTV_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    return 0;
}

// Ends the capture started by graph_begin (always, so the stream leaves capture mode even when the body
// failed) and instantiates the graph.  body_rc is the status of the captured enqueue code.
int graph_end(sbtv_ctx *ctx, int body_rc, hipGraphExec_t *exec) {
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    *exec = nullptr;
    if (body_rc != 0) {
        if (g) (void)hipGraphDestroy(g);
        return body_rc;
    }
    if (e != hipSuccess || !g) return fail_hip(ctx, e, "hipStreamEndCapture", __FILE__, __LINE__);
    const hipError_t ei = hipGraphInstantiate(exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) {
        *exec = nullptr;
        return fail_hip(ctx, ei, "hipGraphInstantiate", __FILE__, __LINE__);
    }
    return 0;
}

}  // namespace sbtv

using namespace sbtv;

extern "C" {

int sbtv_version(void) { return SBTV_VERSION; }

const char *sbtv_last_error(const sbtv_ctx *ctx) {
    if (ctx) return ctx->err.c_str();
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_error_mu);
    copy = g_error;
    return copy.c_str();
}

int sbtv_ctx_create(int device, sbtv_ctx **out) {
    if (!out) return fail(nullptr, SBTV_ERR_BADARG, "sbtv_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SBTV_ERR_NODEVICE,
                    "sbtv_ctx_create: no HIP device visible; libsbtv has no CPU fallback");
    if (device < 0 || device >= ndev)
        return fail(nullptr, SBTV_ERR_NODEVICE, "sbtv_ctx_create: device ordinal out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, SBTV_ERR_NODEVICE, "sbtv_ctx_create: cannot query device");
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(nullptr, SBTV_ERR_NODEVICE,
                    std::string("sbtv_ctx_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    sbtv_ctx *ctx = new sbtv_ctx();
    ctx->device = device;
    ctx->cu_count = prop.multiProcessorCount;
    if (const char *cn = getenv("SBTV_CANARY")) ctx->canary = (cn[0] != '\0' && cn[0] != '0');
    auto init = [&]() -> int {
        SBTV_HIP(ctx, hipSetDevice(device));
        SBTV_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        for (auto &ev : ctx->ev) SBTV_HIP(ctx, hipEventCreate(&ev));
        return 0;
    };
    const int rc = init();
    if (rc != 0) {
        const std::string msg = ctx->err;       // the message outlives the half-built context
        (void)sbtv_ctx_destroy(ctx);
        return fail(nullptr, rc, msg);
    }
    *out = ctx;
    return 0;
}

int sbtv_ctx_destroy(sbtv_ctx *ctx) {
    if (!ctx) return 0;
    // teardown: release everything even if one call fails (nothing useful can be done about it here)
    if (ctx->lanes) (void)sbtv_group_destroy(ctx->lanes);
    ctx->lanes = nullptr;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->ws)
        if (kv.second.base) (void)hipFree(kv.second.base);
    if (ctx->canary_desc) (void)hipFree(ctx->canary_desc);
    if (ctx->canary_bad) (void)hipFree(ctx->canary_bad);
    for (auto &kv : ctx->twiddles) (void)hipFree(kv.second);
    for (auto &kv : ctx->any_axes) (void)hipFree(kv.second);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (auto &l : ctx->stage) {
        for (int k = 0; k < 2; ++k) {
            if (l.pin[k]) (void)hipHostFree(l.pin[k]);
            if (l.ev[k]) (void)hipEventDestroy(l.ev[k]);
        }
        if (l.s) (void)hipStreamDestroy(l.s);
    }
    for (auto &ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : ctx->loop_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int sbtv_ctx_set_stream(sbtv_ctx *ctx, void *hip_stream) {
    if (!ctx) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hip_stream == nullptr) {
        if (!ctx->own_stream) {
            SBTV_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
            ctx->own_stream = true;
        }
        return 0;
    }
    if (ctx->own_stream && ctx->stream) SBTV_HIP(ctx, hipStreamDestroy(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return 0;
}

int sbtv_ctx_sync(sbtv_ctx *ctx) {
    if (!ctx) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sbtv_callcounter_get(const sbtv_ctx *ctx, long long *calls) {
    if (!ctx || !calls) return SBTV_ERR_BADARG;
    *calls = ctx->calls;
    return 0;
}

int sbtv_callcounter_reset(sbtv_ctx *ctx) {
    if (!ctx) return SBTV_ERR_BADARG;
    ctx->calls = 0;
    return 0;
}

int sbtv_last_timing(const sbtv_ctx *ctx, double out[4]) {
    if (!ctx || !out) return SBTV_ERR_BADARG;
    for (int i = 0; i < 4; ++i) out[i] = ctx->timing[i];
    return 0;
}

int sbtv_last_host_stats(const sbtv_ctx *ctx, double out[14]) {
    if (!ctx || !out) return SBTV_ERR_BADARG;
    const HostStats &h = ctx->hstat;
    const double v[14] = {h.waits, h.ready_at_once, h.waits_slept, h.sleeps, h.stream_queries,
                          h.wait_s, h.wait_max_s, h.enqueue_s, h.enqueue_max_s, h.wait_max_outer,
                          h.nvcsw, h.nivcsw, h.minflt, h.majflt};
    for (int i = 0; i < 14; ++i) out[i] = v[i];
    return 0;
}

// Host-side layout helper for row-major hosts (NumPy, C): dst[b][c][r] = src[b][r][c], i.e. `batch` row-major rows x cols
// images -> the column-major images the C-ABI takes (and back, with rows and cols swapped).  64 x 64 blocks, up to four
// threads over the block rows.  (Four 2048^2 images into a warm destination: 58 ms here against 0.8 s for NumPy's strided copy.)
int sbtv_host_transpose(const double *src, double *dst, int batch, int rows, int cols) {
    if (!src || !dst || batch < 0 || rows < 0 || cols < 0) return SBTV_ERR_BADARG;
    constexpr int BS = 64;
    const long long nbr = ((long long)rows + BS - 1) / BS;
    const long long jobs = nbr * batch;
    auto work = [&](int t, int T) {
        for (long long job = t; job < jobs; job += T) {
            const int b = (int)(job / nbr), r0 = (int)(job % nbr) * BS, r1 = std::min(rows, r0 + BS);
            const double *s = src + (size_t)b * rows * cols;
            double *d = dst + (size_t)b * rows * cols;
            for (int c0 = 0; c0 < cols; c0 += BS) {
                const int c1 = std::min(cols, c0 + BS);
                for (int c = c0; c < c1; ++c)
                    for (int r = r0; r < r1; ++r) d[(size_t)c * rows + r] = s[(size_t)r * cols + c];
            }
        }
    };
    const size_t bytes = (size_t)batch * rows * cols * sizeof(double);
    const int T = bytes >= ((size_t)4 << 20) ? 4 : 1;
    std::vector<std::thread> th;
    int started = 1;
    try {
        for (int t = 1; t < T; ++t, ++started) th.emplace_back(work, t, T);
    } catch (...) {                                                    // no thread to be had: the caller's thread does the rest
    }
    work(0, T);
    for (auto &x : th) x.join();
    for (int t = started; t < T; ++t) work(t, T);
    return 0;
}

int sbtv_diag_solve_stats(const sbtv_ctx *ctx, double out[4]) {
    if (!ctx || !out) return SBTV_ERR_BADARG;
    for (int i = 0; i < 4; ++i) out[i] = ctx->solve_stats[i];
    if (ctx->lanes)
        for (int r = 0; r < sbtv_group_size(ctx->lanes); ++r)
            for (int i = 0; i < 4; ++i) out[i] += sbtv_group_ctx(ctx->lanes, r)->solve_stats[i];
    return 0;
}

int sbtv_diag_stage_stats(const sbtv_ctx *ctx, double out[4]) {
    if (!ctx || !out) return SBTV_ERR_BADARG;
    for (int i = 0; i < 4; ++i) out[i] = ctx->stage_stats[i];
    if (ctx->lanes)
        for (int r = 0; r < sbtv_group_size(ctx->lanes); ++r)
            for (int i = 0; i < 4; ++i) out[i] += sbtv_group_ctx(ctx->lanes, r)->stage_stats[i];
    return 0;
}

int sbtv_diag_workspace(sbtv_ctx *ctx, const char *name, void **dptr, size_t *bytes) {
    if (!ctx || !name || !dptr) return SBTV_ERR_BADARG;
    auto it = ctx->ws.find(name);
    if (it == ctx->ws.end() || !it->second.p) return fail(ctx, SBTV_ERR_BADARG, std::string("no workspace named '") + name + "'");
    *dptr = it->second.p;
    if (bytes) *bytes = it->second.bytes;
    return 0;
}

// Every SBTV_* environment switch the library reads (each is read ONCE per process, at first use).  A stray variable
// silently changes which kernels run, so hosts that measure (bench.py) print this next to their numbers.
int sbtv_diag_switches(char *buf, size_t cap) {
    static const char *const names[] = {
        "SBTV_CANARY", "SBTV_COLLECT_RIDE", "SBTV_EXACT", "SBTV_FFT_WAVE", "SBTV_FISTA_FUSED_STEP", "SBTV_FISTA_LAG",
        "SBTV_FUSED_VARIANT", "SBTV_GRAPH", "SBTV_INLINE_CTRL", "SBTV_PROX_SPEC", "SBTV_SAPG_DEFER",
        "SBTV_SAPG_FUSED_MYULA", "SBTV_SAPG_HOST", "SBTV_SINGLE_STEP", "SBTV_SPIN", "SBTV_TAG_SPIN_US", "SBTV_TILE_ORDER", "SBTV_ADMM_EXACT", "SBTV_SALSA_NOX", "SBTV_CSALSA_SPECTRAL",
        "SBTV_CORAL_BATCH", "SBTV_ROWS_FOLD", "SBTV_ERR_SUBSET", "SBTV_LANES", "SBTV_LANE_COUNT", "SBTV_STAGE_THREADS", "SBTV_TEST_FAIL_SAPG", "SBTV_FUSED_STAGGER"};
    // variants that lost their measurements: only the lab build (make lab, -DSBTV_LAB) carries the kernels and reads these
    static const char *const lab_names[] = {"SBTV_PROX_PIPE", "SBTV_ROWS_KERNEL", "SBTV_ROWS_PIPE", "SBTV_ROWS_RK", "SBTV_ROWS_SUB", "SBTV_TAIL_HALF", "SBTV_TAIL_ROWS",
                                            "SBTV_ROWS_V", "SBTV_U_TILED"};
    if (!buf || cap == 0) return SBTV_ERR_BADARG;
    std::string s;
    int nset = 0;
    auto add = [&](const char *n, const char *e, const char *note) {
        s += (nset++ ? " " : "");
        s += n;
        s += "=";
        s += e;
        s += note;
    };
    for (const char *n : names)
        if (const char *e = getenv(n)) add(n, e, "");
    for (const char *n : lab_names)
        if (const char *e = getenv(n)) {
#ifdef SBTV_LAB
            add(n, e, "");
#else
            add(n, e, "(ignored:lab-build-only)");
            --nset;                                   // does not change which kernels run
#endif
        }
#ifdef SBTV_LAB
    s += (s.empty() ? "" : " ");
    s += "[build: SBTV_LAB]";
#endif
    if (s.size() + 1 > cap) s.resize(cap - 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return nset;
}

int sbtv_diag_canary(sbtv_ctx *ctx, int poke, int *enabled, int *nbuf, int *nbad) {
    if (!ctx) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    if (enabled) *enabled = ctx->canary ? 1 : 0;
    int nb = 0, bad = 0;
    if (ctx->canary && poke) {
        // simulate a kernel writing one element past the end of the first workspace
        for (auto &kv : ctx->ws)
            if (kv.second.base) {
                SBTV_HIP(ctx, hipMemsetAsync(static_cast<unsigned char *>(kv.second.p) + kv.second.bytes, 0, 8, ctx->stream));
                break;
            }
    }
    std::string name;
    SBTV_TRY(canary_verify(ctx, &nb, &bad, &name));
    if (ctx->canary && poke && bad) {
        // repair the poked guard so that the context stays usable
        for (auto &kv : ctx->ws)
            if (kv.second.base) {
                SBTV_HIP(ctx, hipMemsetAsync(static_cast<unsigned char *>(kv.second.p) + kv.second.bytes, kGuardByte, kGuard, ctx->stream));
                break;
            }
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (nbuf) *nbuf = nb;
    if (nbad) *nbad = bad;
    return 0;
}

int sbtv_malloc(sbtv_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    SBTV_HIP(ctx, hipMalloc(dptr, bytes));
    return 0;
}

int sbtv_free(sbtv_ctx *ctx, void *dptr) {
    if (!ctx) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SBTV_HIP(ctx, hipFree(dptr));
    return 0;
}

int sbtv_memcpy_h2d(sbtv_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    if (stage_threaded(bytes)) {
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return stage_copy(ctx, dst, src, bytes, true);
    }
    SBTV_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sbtv_memcpy_d2h(sbtv_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    if (stage_threaded(bytes)) {
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return stage_copy(ctx, dst, src, bytes, false);
    }
    SBTV_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------
// results.err_psf of the SAPG loops: l2(psf(params(ii)), psf_true) with utils/l2.m = norm(x-y)^2, where norm() of a
// MATRIX is the spectral norm (quirk Q9).  Largest eigenvalue of D'D by cyclic Jacobi sweeps (D is at most 15 x 15).
// ---------------------------------------------------------------------------
static double spectral_norm_sq(const double *D, int t) {
    double B[15 * 15];
    for (int i = 0; i < t; ++i)
        for (int j = 0; j < t; ++j) {
            double s = 0.0;
            for (int k = 0; k < t; ++k) s += D[i * t + k] * D[j * t + k];   // D is column-major: (D'D)(i,j)
            B[i * t + j] = s;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < t; ++i)
            for (int j = 0; j < t; ++j) (i == j ? diag : off) += B[i * t + j] * B[i * t + j];
        if (off <= 1e-34 * diag || off == 0.0) break;
        for (int p = 0; p < t - 1; ++p)
            for (int q = p + 1; q < t; ++q) {
                const double apq = B[p * t + q];
                if (apq == 0.0) continue;
                const double theta = (B[q * t + q] - B[p * t + p]) / (2.0 * apq);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < t; ++k) {       // rotate columns p, q
                    const double bkp = B[k * t + p], bkq = B[k * t + q];
                    B[k * t + p] = cs * bkp - sn * bkq;
                    B[k * t + q] = sn * bkp + cs * bkq;
                }
                for (int k = 0; k < t; ++k) {       // rotate rows p, q
                    const double bpk = B[p * t + k], bqk = B[q * t + k];
                    B[p * t + k] = cs * bpk - sn * bqk;
                    B[q * t + k] = sn * bpk + cs * bqk;
                }
            }
    }
    double mx = 0.0;
    for (int i = 0; i < t; ++i) mx = fmax(mx, B[i * t + i]);
    return mx;
}

int sbtv_err_psf(int kind, int taille, const double *ps, int n, const double *p_true, double phi, double *out) {
    if (kind < 0 || kind > 2 || taille < 1 || taille > 15 || !ps || !p_true || !out || n < 1)
        return fail(nullptr, SBTV_ERR_PSF, "sbtv_err_psf: bad arguments");
    const int t2 = taille * taille;
    double truth[225], cur[225], D[225];
    double pt[3] = {p_true[0], kind == SBTV_PSF_LAPLACE ? 0.0 : p_true[1], kind == SBTV_PSF_GAUSSIAN ? phi : 0.0};
    int rc = sbtv_psf_taps(kind, taille, pt, truth, nullptr, nullptr);
    if (rc != 0) return rc;
    for (int i = 0; i < n; ++i) {
        if (kind == SBTV_PSF_MOFFAT && i == 0) {          // kept under another name by the reference (:156)
            out[0] = 0.0;
            continue;
        }
        // gaussian: w1s(ii) with w2s(ii-1) (SAPG_algorithm_Guassian.m:203, quirk Q8); ps = [p0 trace | p1 trace]
        double p[3] = {ps[i], 0.0, kind == SBTV_PSF_GAUSSIAN ? phi : 0.0};
        if (kind == SBTV_PSF_GAUSSIAN) p[1] = ps[n + (i > 0 ? i - 1 : 0)];
        if (kind == SBTV_PSF_MOFFAT) p[1] = ps[n + i];
        rc = sbtv_psf_taps(kind, taille, p, cur, nullptr, nullptr);
        if (rc != 0) return rc;
        for (int q = 0; q < t2; ++q) D[q] = cur[q] - truth[q];
        out[i] = spectral_norm_sq(D, taille);
    }
    return 0;
}

// ---------------------------------------------------------------------------
// PSF taps (host, double) — formulas of utils/*.m, column-major taille x taille.
// Element (ii,jj) (0-based) at taps[jj*taille + ii].
// ---------------------------------------------------------------------------
int sbtv_psf_taps(int kind, int taille, const double *p, double *taps, double *d0, double *d1) {
    if (taille < 1 || taille > 15 || !p || !taps)
        return fail(nullptr, SBTV_ERR_PSF, "sbtv_psf_taps: bad arguments (1 <= taille <= 15)");
    const int t = taille;
    if (kind != SBTV_PSF_GAUSSIAN && kind != SBTV_PSF_MOFFAT && kind != SBTV_PSF_LAPLACE)
        return fail(nullptr, SBTV_ERR_PSF, "sbtv_psf_taps: unknown PSF kind");
    std::vector<double> f(t * t), e0(t * t, 0.0), e1(t * t, 0.0);
    double s = 0, s0 = 0, s1 = 0;
    for (int q = 0; q < t * t; ++q) psf_taps_point(kind, t, p, q, &f[q], &e0[q], &e1[q]);   // psf_taps.inc
    // MATLAB sum(sum(.)) / sum(k(:)) : column-major accumulation order
    for (int q = 0; q < t * t; ++q) { s += f[q]; s0 += e0[q]; s1 += e1[q]; }
    for (int q = 0; q < t * t; ++q) {
        taps[q] = f[q] / s;
        if (d0) d0[q] = (e0[q] * s - f[q] * s0) / (s * s);
        if (d1) d1[q] = (e1[q] * s - f[q] * s1) / (s * s);
    }
    return 0;
}

}  // extern "C"
