// Host-only entry points of libsbtv.so under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5, row 2:
// "host runtime built under ASan/UBSan").  Built and run on the CPU box by tests/test_host_sanitize.py against
// lib/libsbtv_asan.so (make -C csrc sanitize: the HOST side of every translation unit instrumented; GPU sanitizers are not
// available on the target pool).  No GPU is needed: everything here is host arithmetic, argument validation or the
// no-device error path.  Exit code 0 and no sanitizer report = pass.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "sbtv.h"

static int fails = 0;
#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); \
            ++fails;                                                     \
        }                                                                \
    } while (0)

int main() {
    // ---- PSF tap builders: every family, smallest and largest size, derivative outputs present / absent
    for (int kind = 0; kind < 3; ++kind)
        for (int t : {1, 2, 7, 15}) {
            std::vector<double> taps((size_t)t * t), d0((size_t)t * t), d1((size_t)t * t);
            const double p[3] = {kind == 1 ? 0.4 : 0.5, kind == 1 ? 3.5 : 0.3, 0.1};
            CHECK(sbtv_psf_taps(kind, t, p, taps.data(), d0.data(), d1.data()) == 0);
            CHECK(sbtv_psf_taps(kind, t, p, taps.data(), nullptr, nullptr) == 0);
            double s = 0;
            for (double v : taps) s += v;
            CHECK(std::fabs(s - 1.0) < 1e-12);
        }
    {
        double p[3] = {0.4, 0.3, 0.0}, taps[225];
        CHECK(sbtv_psf_taps(3, 7, p, taps, nullptr, nullptr) == SBTV_ERR_PSF);
        CHECK(sbtv_psf_taps(0, 0, p, taps, nullptr, nullptr) == SBTV_ERR_PSF);
        CHECK(sbtv_psf_taps(0, 16, p, taps, nullptr, nullptr) == SBTV_ERR_PSF);
        CHECK(sbtv_psf_taps(0, 7, nullptr, taps, nullptr, nullptr) == SBTV_ERR_PSF);
        CHECK(sbtv_psf_taps(0, 7, p, nullptr, nullptr, nullptr) == SBTV_ERR_PSF);
        CHECK(std::strlen(sbtv_last_error(nullptr)) > 0);
    }
    // ---- results.err_psf traces (Jacobi spectral norm on the stack)
    for (int kind = 0; kind < 3; ++kind)
        for (int t : {1, 7, 15}) {
            const int n = 9;
            std::vector<double> ps(2 * n), out(n, -1.0);
            for (int i = 0; i < n; ++i) {
                ps[i] = 0.2 + 0.05 * i;
                ps[n + i] = kind == 1 ? 2.0 + 0.3 * i : 0.3 + 0.02 * i;
            }
            const double pt[2] = {0.4, kind == 1 ? 3.5 : 0.3};
            CHECK(sbtv_err_psf(kind, t, ps.data(), n, pt, 0.0, out.data()) == 0);
            for (int i = 0; i < n; ++i) CHECK(out[i] >= 0.0 && std::isfinite(out[i]));
            if (kind == 1) CHECK(out[0] == 0.0);
        }
    {
        double ps[2] = {0.4, 0.3}, pt[2] = {0.4, 0.3}, out[1];
        CHECK(sbtv_err_psf(0, 7, ps, 0, pt, 0.0, out) == SBTV_ERR_PSF);
        CHECK(sbtv_err_psf(5, 7, ps, 1, pt, 0.0, out) == SBTV_ERR_PSF);
        CHECK(sbtv_err_psf(0, 7, nullptr, 1, pt, 0.0, out) == SBTV_ERR_PSF);
    }
    // ---- option defaults
    {
        sbtv_salsa_opts o;
        std::memset(&o, 0xff, sizeof o);
        sbtv_salsa_opts_default(&o);
        CHECK(o.stopcriterion == 1 && o.maxiter == 10000 && o.TViters == 5 && o.tolA == 0.001 && o.chambolle_tau == 0.249);
        sbtv_salsa_opts_default(nullptr);
    }
    // ---- argument validation without a context, and the no-device path (this box has no GPU: no CPU fallback)
    {
        CHECK(sbtv_version() == SBTV_VERSION);
        sbtv_ctx *ctx = nullptr;
        const int rc = sbtv_ctx_create(0, &ctx);
        if (rc == 0) {                       // (a GPU box: the context path is covered by the GPU tests)
            CHECK(ctx != nullptr);
            CHECK(sbtv_ctx_destroy(ctx) == 0);
        } else {
            CHECK(rc == SBTV_ERR_NODEVICE && ctx == nullptr);
            CHECK(std::strstr(sbtv_last_error(nullptr), "no") != nullptr);
        }
        CHECK(sbtv_ctx_create(0, nullptr) == SBTV_ERR_BADARG);
        CHECK(sbtv_ctx_destroy(nullptr) == 0);
        double y[4] = {1, 2, 3, 4}, lam = 1.0, out[4];
        int k = 0;
        CHECK(sbtv_chambolle_prox_TV_stop(nullptr, y, 2, 2, 1, &lam, 5, 1e-3, 0.249, 0, nullptr, nullptr, out, &k, nullptr, 0) == SBTV_ERR_BADARG);
        CHECK(sbtv_TVnorm(nullptr, y, 2, 2, 1, out, 0) == SBTV_ERR_BADARG);
        sbtv_salsa_opts o;
        sbtv_salsa_opts_default(&o);
        CHECK(sbtv_SALSA_v2(nullptr, y, 2, 2, 1, nullptr, 7, &lam, &lam, &o, nullptr, nullptr, out, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr, 0) == SBTV_ERR_BADARG);
        double hs[14], tm[4];
        long long calls = 0;
        CHECK(sbtv_last_host_stats(nullptr, hs) == SBTV_ERR_BADARG);
        CHECK(sbtv_last_timing(nullptr, tm) == SBTV_ERR_BADARG);
        CHECK(sbtv_callcounter_get(nullptr, &calls) == SBTV_ERR_BADARG);
        CHECK(sbtv_ctx_sync(nullptr) == SBTV_ERR_BADARG);
    }
    // ---- switches report into buffers of every size (truncation must stay inside the buffer)
    for (size_t cap : {(size_t)1, (size_t)2, (size_t)17, (size_t)4096}) {
        std::vector<char> buf(cap, 'x');
        CHECK(sbtv_diag_switches(buf.data(), cap) >= 0);
        CHECK(std::memchr(buf.data(), 0, cap) != nullptr);
    }
    CHECK(sbtv_diag_switches(nullptr, 10) == SBTV_ERR_BADARG);
    // ---- group: creation fails cleanly without a device (every half-built piece released), bad arguments
    {
        sbtv_group *g = nullptr;
        int devs[3] = {0, 0, 0};
        const int rc = sbtv_group_create(devs, 3, &g);
        if (rc == 0) {
            int shard = -1, first = -1, count = -1;
            CHECK(sbtv_group_size(g) == 3);
            CHECK(sbtv_group_shard_of(g, 5, 4, &shard, &first, &count) == 0 && shard == 2 && first == 4 && count == 1);
            CHECK(sbtv_group_shard_of(g, 5, 5, &shard, &first, &count) == SBTV_ERR_BADARG);
            CHECK(sbtv_group_destroy(g) == 0);
        } else {
            CHECK(g == nullptr && rc == SBTV_ERR_NODEVICE);
        }
        CHECK(sbtv_group_create(nullptr, 3, &g) == SBTV_ERR_BADARG);
        CHECK(sbtv_group_create(devs, 0, &g) == SBTV_ERR_BADARG);
        CHECK(sbtv_group_create(devs, 3, nullptr) == SBTV_ERR_BADARG);
        CHECK(sbtv_group_size(nullptr) == 0 && sbtv_group_ctx(nullptr, 0) == nullptr);
        CHECK(sbtv_group_destroy(nullptr) == 0);
        CHECK(sbtv_SALSA_v2_sharded(nullptr, nullptr, 2, 2, 1, nullptr, 7, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == SBTV_ERR_BADARG);
    }
    // ---- round 4: the threaded layout helper on ragged shapes (block edges, one row / one column, threads vs no threads),
    // the new sharded entry points and diagnostics without a group / context
    {
        const int shapes[][3] = {{1, 1, 1}, {2, 1, 9}, {3, 65, 1}, {2, 64, 64}, {1, 130, 67}, {4, 700, 300}};
        for (const auto &sh : shapes) {
            const int B = sh[0], R = sh[1], Cc = sh[2];
            std::vector<double> a((size_t)B * R * Cc), t(a.size(), -1.0), back(a.size(), -2.0);
            for (size_t i = 0; i < a.size(); ++i) a[i] = (double)i * 0.5 - 3.0;
            CHECK(sbtv_host_transpose(a.data(), t.data(), B, R, Cc) == 0);
            bool ok = true;
            for (int b = 0; b < B && ok; ++b)
                for (int r = 0; r < R && ok; ++r)
                    for (int c = 0; c < Cc; ++c)
                        if (t[((size_t)b * Cc + c) * R + r] != a[((size_t)b * R + r) * Cc + c]) { ok = false; break; }
            CHECK(ok);
            CHECK(sbtv_host_transpose(t.data(), back.data(), B, Cc, R) == 0);
            CHECK(back == a);
        }
        double d = 0.0;
        CHECK(sbtv_host_transpose(nullptr, &d, 1, 1, 1) == SBTV_ERR_BADARG);
        CHECK(sbtv_host_transpose(&d, &d, 1, -1, 1) == SBTV_ERR_BADARG);
        CHECK(sbtv_host_transpose(&d, &d, 0, 5, 5) == 0);
        double st4[4];
        CHECK(sbtv_diag_stage_stats(nullptr, st4) == SBTV_ERR_BADARG && sbtv_diag_solve_stats(nullptr, st4) == SBTV_ERR_BADARG);
        CHECK(sbtv_ctx_set_lanes(nullptr, 0) == SBTV_ERR_BADARG);
        CHECK(sbtv_fista_tv_sharded(nullptr, nullptr, 2, 2, 1, nullptr, 7, nullptr, 1.0, 25, 1, 0.0, 1, 0, nullptr, nullptr, nullptr,
                                    nullptr, nullptr) == SBTV_ERR_BADARG);
        CHECK(sbtv_CSALSA_v2_sharded(nullptr, nullptr, 2, 2, 1, nullptr, 7, nullptr, nullptr, nullptr, nullptr, 1.0, nullptr, nullptr,
                                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                     nullptr) == SBTV_ERR_BADARG);
        CHECK(sbtv_CoRAL_v2_sharded(nullptr, nullptr, 2, 2, 1, nullptr, 7, nullptr, nullptr, nullptr, nullptr, nullptr, 5, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    nullptr) == SBTV_ERR_BADARG);
        CHECK(sbtv_SALSA_v2_sharded_dev(nullptr, nullptr, 2, 2, 1, nullptr, 7, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == SBTV_ERR_BADARG);
    }
    std::printf(fails ? "%d host checks FAILED\n" : "host sanitizer run: all checks passed\n", fails);
    return fails ? 1 : 0;
}
