#!/usr/bin/env python3
"""Aggregate throughput of independent SALSA solves sharing ONE MI355X: T host threads, each with its own
sbtv context (own HIP stream and workspaces), versus one solve of a batch of T images in one context.

  python tools/bench_concurrent.py [--threads 2] [--size 2048] [--steps 200]

Independent images are the natural sharding unit.  A single solve alternates between a
VALU-bound kernel (fused Chambolle) and memory-bound ones (FFT passes); two solves on two streams can overlap
those phases, a batch in one context runs them in lock-step.  Prints image-iterations/s for each arrangement.
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import numpy as np
import sbtv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    man = np.load(os.path.join(ROOT, "tests", "golden", "man_512.npy")).astype(np.float64)
    r = max(1, a.size // 512)
    x = np.tile(man, (r, r))[:a.size, :a.size]
    rng = np.random.default_rng(1)
    st = sbtv.demo_setup("gaussian", x, rng.standard_normal(x.shape), evMax=1.0)
    theta, mu = 0.03, 0.003
    tau = theta * st["sigma"] ** 2
    taps = sbtv.Gaussian_psf(7, 0.4, 0.3)

    def solve(ctx, y, xt, steps):
        A = sbtv.BlurOperator(taps, ctx=ctx)
        return sbtv.SALSA_v2(y, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xt, "ToleranceA", -1.0,
                             "MAXITERA", steps, "TVINITIALIZATION", 1, "TViters", 10, "VERBOSE", 0, ctx=ctx)

    T = a.threads
    ctxs = [sbtv.Context(0) for _ in range(T)]
    ys = [sbtv.to_device(st["y"]) for _ in range(T)]
    xs = [sbtv.to_device(x) for _ in range(T)]
    for c, y, xt in zip(ctxs, ys, xs):
        solve(c, y, xt, 5)                                   # workspaces, twiddles
    # (1) one context, one image
    t0 = time.perf_counter()
    solve(ctxs[0], ys[0], xs[0], a.steps)
    t1 = time.perf_counter() - t0
    print(f"1 context, 1 image            : {a.steps / t1:9.1f} image-iterations/s")
    # (2) T threads x 1 image, one context each
    th = [threading.Thread(target=solve, args=(c, y, xt, a.steps)) for c, y, xt in zip(ctxs, ys, xs)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    tT = time.perf_counter() - t0
    print(f"{T} contexts (threads), 1 image : {T * a.steps / tT:9.1f} image-iterations/s")
    # (3) one context, batch of T images
    yb = sbtv.to_device(np.stack([st["y"]] * T))
    xb = sbtv.to_device(np.stack([x] * T))
    solve(ctxs[0], yb, xb, 5)
    t0 = time.perf_counter()
    solve(ctxs[0], yb, xb, a.steps)
    tb = time.perf_counter() - t0
    print(f"1 context, batch of {T}         : {T * a.steps / tb:9.1f} image-iterations/s")


if __name__ == "__main__":
    main()
