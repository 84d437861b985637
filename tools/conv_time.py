#!/usr/bin/env python3
"""SALSA wall time per solve at 2048^2 and 512^2: a converging solve and fixed-length ones (see tools/README.md)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import bench, numpy as np, torch, sbtv
ctx = sbtv.Context(0)
for size in (2048, 512):
    x, y, sigma, noise = bench.make_problem(1, size)
    yd, xd = sbtv.to_device(y, "cuda:0"), sbtv.to_device(x, "cuda:0")
    A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, *bench.W_TRUE), ctx=ctx)
    mu, tau = bench.THETA / 10, bench.THETA * sigma ** 2
    def solve(K, tol):
        return sbtv.SALSA_v2(yd, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xd, "StopCriterion", 1, "ToleranceA", tol,
                             "MAXITERA", K, "TVINITIALIZATION", 1, "TViters", 10, ctx=ctx)
    solve(100, -1.0)
    for K, tol in ((500, 1e-5), (60, -1.0), (150, -1.0), (400, -1.0)):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); out = solve(K, tol); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        n = len(out[3]) - 1
        print(size, "K", K, "tol", tol, "iters", n, "ms", round(1e3 * min(ts), 3), "it/s", round(n / min(ts)), "launches", ctx.last_timing().get("chambolle_launches"), flush=True)
