#!/usr/bin/env python3
"""Throughput of the single-process multi-device path (sbtv_group, sbtv_SALSA_v2_sharded) on ONE GPU with virtual shards:
4 images of the headline problem, host buffers (as a MATLAB host passes them), groups of 1 / 2 / 4 contexts on device 0 and
calls of 20 / 200 / 800 outer iterations (the fixed part of a call = host <-> device copies of the images), next to the
single-context batch with host and with device buffers."""
import os, sys, time
os.environ.setdefault("OPENBLAS_NUM_THREADS","1"); os.environ.setdefault("OMP_NUM_THREADS","1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import numpy as np, torch, sbtv, bench
x1, y1, s1, _ = bench.make_problem(seed=1)
ctx = sbtv.Context(0)
A = sbtv.BlurOperator(sbtv.Gaussian_psf(7, *bench.W_TRUE), ctx=ctx)
mu, tau = bench.THETA / 10, bench.THETA * s1 ** 2
ys, xs = np.stack([y1] * 4), np.stack([x1] * 4)
def run(c, n, yy=ys, xx=xs):
    return sbtv.SALSA_v2(yy, A, tau, "MU", mu, "AT", A.T, "LS", A.LS(mu), "True_x", xx, "StopCriterion", 1, "ToleranceA", -1.0, "MAXITERA", n, "TVINITIALIZATION", 1, "TViters", 10, ctx=c)
for nsh in (1, 2, 4):
    g = sbtv.Group([0] * nsh)
    run(g, 5)
    for n in (20, 200, 800):
        t = time.perf_counter(); run(g, n); e = time.perf_counter() - t
        print(f"group of {nsh}: n={n}: {1e3*e:.1f} ms -> {4*n/e:.0f} image-it/s", flush=True)
    g.close()
run(ctx, 5)
for n in (20, 200, 800):
    t = time.perf_counter(); run(ctx, n); e = time.perf_counter() - t
    print(f"one context, host buffers, batch 4: n={n}: {1e3*e:.1f} ms -> {4*n/e:.0f} image-it/s", flush=True)
yd, xd = sbtv.to_device(ys), sbtv.to_device(xs)
run(ctx, 5, yd, xd)
for n in (20, 200):
    torch.cuda.synchronize(); t = time.perf_counter(); run(ctx, n, yd, xd); torch.cuda.synchronize(); e = time.perf_counter() - t
    print(f"one context, device buffers, batch 4: n={n}: {1e3*e:.1f} ms -> {4*n/e:.0f} image-it/s", flush=True)
