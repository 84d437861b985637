#!/usr/bin/env python3
"""Fixed cost of one SALSA_v2 call at 2048^2: wall time of calls with K = 1 ... 100 outer iterations (best of 5) and the
time the library's own events measure for the call on the device (sbtv_last_timing[0]); fixed cost = call(K) - K x the
step time between K = 20 and K = 100."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import torch, sbtv, bench
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = sbtv.default_context(0)
x, y, sigma, noise = bench.make_problem(1, size)
yd, xd = sbtv.to_device(y), sbtv.to_device(x)
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, 0.4, 0.3), ctx=ctx)
mu, tau = 0.003, 0.03 * sigma ** 2
def solve(k):
    return sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "ToleranceA", -1.0, "MAXITERA", k,
                         "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
solve(30); torch.cuda.synchronize()
res = {}
for k in (1, 2, 5, 20, 100):
    best = 1e9; dev = 0
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); solve(k); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if dt < best: best, dev = dt, ctx.last_timing()["loop_ms"]
    res[k] = (best, dev)
step = (res[100][0] - res[20][0]) / 80
for k, (best, dev) in res.items():
    print(f"K={k:4d}: call {1e3*best:7.3f} ms (device events {dev:7.3f} ms), fixed cost {1e3*(best-k*step):6.3f} ms at {1e3*step:.4f} ms per step")
